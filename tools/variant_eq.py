#!/usr/bin/env python3
"""Bit-for-bit comparison of two settings of the library on the variant child of tests/test_timed_path_gpu.py (70-chunk call +
5-chunk continuation).  usage: python tools/variant_eq.py "ENV=V ..." "ENV=V ..."   (INFV_LTM_LIBRARY may be among them)"""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.test_timed_path_gpu import _VARIANT_CHILD
outs = []
for spec in sys.argv[1:3]:
    env = dict(os.environ)
    for kv in spec.split():
        k, v = kv.split("=", 1); env[k] = v
    path = tempfile.mktemp(suffix=".npz")
    subprocess.run([sys.executable, "-c", _VARIANT_CHILD, path], check=True, env=env, cwd=ROOT, timeout=900)
    outs.append({k: v for k, v in np.load(path).items()})
bad = 0
for key in outs[0]:
    same = np.array_equal(outs[0][key], outs[1][key])
    d = float(np.abs(outs[0][key].astype(np.float64) - outs[1][key].astype(np.float64)).max())
    print(f"{key}: {'identical' if same else 'DIFFERENT'}  max |diff| {d:.3e}  finite {bool(np.isfinite(outs[0][key].astype(np.float64)).all())}")
    bad += not same
print("VARIANT_EQ_OK" if bad == 0 else "VARIANT_EQ_FAIL")
