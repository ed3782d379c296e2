#!/usr/bin/env python3
"""Registers / scratch / static LDS of every kernel in a built library (no GPU needed):
   python tools/kernel_regs.py [libinfv_ltm.so|libinfv_ltm_exp.so] [name fragment ...]"""
import os, re, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else "libinfv_ltm.so"
frags = [a for a in sys.argv[1:] if not a.endswith(".so")]
with tempfile.TemporaryDirectory() as d:
    shutil.copy(os.path.join(ROOT, "infinite-video_amd", lib), os.path.join(d, lib))
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", lib], cwd=d, capture_output=True, check=True)
    for f in sorted(os.listdir(d)):
        if "amdgcn" not in f:
            continue
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", f], cwd=d, capture_output=True, text=True, check=True).stdout
        cur = {}
        rows = []
        # the metadata keys of a kernel come in alphabetical order (.agpr_count ... .wavefront_size): a record ends at .wavefront_size
        for line in notes.split("\n"):
            m = re.search(r"\.(name|vgpr_count|sgpr_count|agpr_count|private_segment_fixed_size|group_segment_fixed_size|vgpr_spill_count|wavefront_size):\s+(\S+)", line)
            if m:
                if m.group(1) == "wavefront_size":
                    if "name" in cur:
                        rows.append(cur)
                    cur = {}
                else:
                    cur[m.group(1)] = m.group(2)
        for r in rows:
            n = subprocess.run(["c++filt", r.get("name", "?")], capture_output=True, text=True).stdout.strip()
            if frags and not any(x in n for x in frags):
                continue
            print(f"{n[:110]:110s} vgpr {r.get('vgpr_count','?'):>4} agpr {r.get('agpr_count','0'):>4} sgpr {r.get('sgpr_count','?'):>4} "
                  f"scratch {r.get('private_segment_fixed_size','0'):>5} lds {r.get('group_segment_fixed_size','0'):>6} spill {r.get('vgpr_spill_count','0')}")
