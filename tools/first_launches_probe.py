#!/usr/bin/env python3
"""[developer check script — TEST INFRASTRUCTURE like tests/] The first three full launches of a process do not get all 1024 workgroups of a 512 + 512 pool resident (finding 4).  Do three
SHORT real launches of the same shape (512 tiny frames) cure that before the first long one?  Run with the library of commit 3d677b3 (pools use every slot, re-warm in front of every launch).
usage: IMCVT_HEVC_LIB=.../libimcvt_hevc_fullpool.so python tools/first_launches_probe.py [0|1]   (1: with the short launches)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, imcvt_amd
from imcvt_amd import synth
short = len(sys.argv) > 1 and sys.argv[1] == "1"
imgs = [torch.from_numpy(synth.syn(1920, 1080, s)).cuda() for s in range(512)]
enc = imcvt_amd.DeviceEncoder()
b = enc.make_batch(imgs, 0)
out = {"short_launches_first": short}
if short:
    tiny = [torch.from_numpy(synth.syn(96, 64, s)).cuda() for s in range(512)]
    bt = enc.make_batch(tiny, 0)
    v = []
    for _ in range(3):
        enc.encode(bt); torch.cuda.synchronize(); v.append((round(enc.last_kernel_ms(), 2), enc.last_shape(), enc.last_resident(), enc.last_start_spread_us()))
    out["short"] = v
v = []
for _ in range(4):
    enc.encode(b); torch.cuda.synchronize(); v.append((round(enc.last_kernel_ms(), 1), enc.last_shape(), enc.last_resident(), enc.last_start_spread_us()))
out["long"] = v
print(json.dumps(out), flush=True)
