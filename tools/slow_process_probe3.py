#!/usr/bin/env python3
"""[developer check script — TEST INFRASTRUCTURE like tests/] A 256-thread pool launch puts a process's full 192-thread launches into a slow regime (5.15 s against 4.81 s).  Does the context's pre-warm (empty
full-grid launches, run when a context is created) bring the fast regime back?   usage: python tools/slow_process_probe3.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, imcvt_amd
from imcvt_amd import synth
n = 512
imgs = [torch.from_numpy(synth.syn(1920, 1080, s)).cuda() for s in range(n)]
enc = imcvt_amd.DeviceEncoder()
b = enc.make_batch(imgs, 0)
b200 = enc.make_batch(imgs[:200], 0)
enc.encode(b); torch.cuda.synchronize()
out = {}
def run(tag, k=3):
    v = []
    for _ in range(k):
        enc.encode(b); torch.cuda.synchronize(); v.append((round(enc.last_kernel_ms(), 1), enc.last_resident(), enc.last_start_spread_us()))
    out[tag] = v; print(tag, v, flush=True)
run("1_fresh")
enc.encode(b200); torch.cuda.synchronize()
run("2_after_a_256_thread_pool_launch")
e2 = imcvt_amd.DeviceEncoder(); e2.close()
run("3_after_creating_a_context_(pre-warm)")
enc.encode(b200); torch.cuda.synchronize()
run("4_after_another_256_thread_launch")
for _ in range(3): enc.lib.imcvt_hevc_debug_census(enc.ctx, 1024)
torch.cuda.synchronize()
run("5_after_three_census_launches_of_1024_x_192")
print(json.dumps({"probe": "slow_process3", **out}))
