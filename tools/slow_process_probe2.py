#!/usr/bin/env python3
"""[developer check script — TEST INFRASTRUCTURE like tests/] Is it the first launch of wide workgroups that puts a process's full launches into the slow mode?  Resident launches only.
usage: python tools/slow_process_probe2.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, imcvt_amd
from imcvt_amd import synth
n = 512
imgs = [torch.from_numpy(synth.syn(1920, 1080, s)).cuda() for s in range(n)]
enc = imcvt_amd.DeviceEncoder()
b = enc.make_batch(imgs, 0)
small = {k: enc.make_batch(imgs[:k], 0) for k in (1, 32, 200)}
enc.encode(b); torch.cuda.synchronize()
out = {}
def run(tag, k=3):
    v = []
    for _ in range(k):
        enc.encode(b); torch.cuda.synchronize(); v.append(round(enc.last_kernel_ms(), 1))
    out[tag] = v; print(tag, v, flush=True)
run("1_fresh", 4)
enc.encode(small[200]); torch.cuda.synchronize(); print("200 frames:", enc.last_shape(), enc.last_pipe(), enc.last_wide(), flush=True)
run("2_after_a_256_thread_pool_launch")
enc.encode(small[32]); torch.cuda.synchronize(); print("32 frames:", enc.last_shape(), enc.last_pipe(), enc.last_wide(), enc.last_partners(), flush=True)
run("3_after_a_wide_launch_with_partners")
enc.set_partners(0); enc.encode(small[1]); torch.cuda.synchronize(); enc.set_partners(1)
run("4_after_a_wide_launch_without_partners")
st = torch.cuda.Stream()
enc.encode(small[32], stream=st); torch.cuda.synchronize()
run("5_after_a_wide_launch_on_another_stream")
print(json.dumps({"probe": "slow_process2", **out}))
