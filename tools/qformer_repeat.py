import sys, os
sys.path.insert(0, os.getcwd())
import torch
from infinite_video_amd import synth
from infinite_video_amd.video_qformer import InfVideoEncoder
dev = torch.device("cuda:0")
m = InfVideoEncoder(num_basis=256, tau=0.75, alpha=0.9, sticky=True)
m.load_reference_state_dict(synth.video_qformer_weights())
m = m.to(dev)
frames = torch.randn(64, 256 * 32, 768, device=dev)
u = torch.from_numpy(synth.gibbs_uniforms(64, 2)).to(dev)
first = None; bad = 0
for i in range(30):
    out = m.encode_frames_batch(frames, new_video=True, u=u)
    out = out[0] if isinstance(out, (tuple, list)) else out
    torch.cuda.synchronize()
    if first is None: first = out.clone()
    elif not torch.equal(first, out): bad += 1
print("layer-major video Q-former, 30 repeats of 64 chunks:", bad, "differ from the first; finite:", bool(torch.isfinite(first).all()))
del m
