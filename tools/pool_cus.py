"""Per-CU streaming rate of the pooling kernel alone: pool_rows2_kernel restricted to W resident workgroups (grid-stride,
one per CU through the padding LDS).  usage (GPU box): INFV_LTM_LIBRARY=exp INFV_PR_WGS=<W> python tools/pool_cus.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infinite_video_amd.engine import LTMEngine
dev = torch.device("cuda:0")
T, P, D = 256, 32, 768
eng = LTMEngine(256, 12, 64, D, P, tau=.75, sticky=True, device=dev)
k = torch.randn(126, T * P, D, device=dev)
eng.pool_rows(k); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): eng.pool_rows(k)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
gb = 126 * (255 * P * D * 4 + 64 * D * 4) / 1e9
w = int(os.environ.get("INFV_PR_WGS", "0"))
print(f"WGS={w or 'all'} pad={os.environ.get('INFV_PR_PAD','84K')}: {ms:.3f} ms, {gb/ms*1e3:.0f} GB/s" + (f", {gb/ms*1e3/w:.1f} GB/s per workgroup" if w else ""))
