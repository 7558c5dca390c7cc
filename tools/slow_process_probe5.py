#!/usr/bin/env python3
"""[developer check script — TEST INFRASTRUCTURE like tests/] Do the launches of wide / pipe-wave workgroups have a slow state of their own after launches of another kind?  64 frames, one frame, 200 frames,
each fresh and after a full 192-thread launch / a launch of the other kinds.   usage: python tools/slow_process_probe5.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, imcvt_amd
from imcvt_amd import synth
imgs = [torch.from_numpy(synth.syn(1920, 1080, s)).cuda() for s in range(512)]
enc = imcvt_amd.DeviceEncoder()
B = {k: enc.make_batch(imgs[:k], 0) for k in (1, 64, 200, 512)}
out = {}
def run(k, tag, reps=2):
    v = []
    for _ in range(reps):
        enc.encode(B[k]); torch.cuda.synchronize(); v.append(round(enc.last_kernel_ms(), 1))
    out[f"{tag}: {k} frames"] = v; print(tag, k, v, flush=True)
run(64, "1 fresh", 3); run(1, "1 fresh (after the 64-frame launches)", 2)
run(512, "2 one full 192-thread launch", 1)
run(64, "3 after the full launch", 3); run(1, "3 after the full launch", 2)
run(200, "4 pipe-wave pool", 2)
run(64, "5 after the pipe-wave pool", 3); run(1, "5 after the pipe-wave pool", 2)
run(512, "6 full launch again", 1)
run(200, "7 pipe-wave pool after the full launch", 2)
e2 = imcvt_amd.DeviceEncoder(); e2.close()
run(64, "8 after creating a context", 3); run(1, "8 after creating a context", 2); run(200, "8 after creating a context", 2)
print(json.dumps({"probe": "slow_process5", **out}))
