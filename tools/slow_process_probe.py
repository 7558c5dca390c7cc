#!/usr/bin/env python3
"""[developer check script — TEST INFRASTRUCTURE like tests/] Every launch of a process that has used the host-pointer path is ~4 % slower (5.0 s against 4.8 s for the bench batch).  What does it: extra streams,
pinned buffers, a second context, the host path itself?  Resident launches of the 512 bench frames after each step.   usage: python tools/slow_process_probe.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, imcvt_amd
from imcvt_amd import synth, hevc
n = 512
host = [synth.syn(1920, 1080, s) for s in range(n)]
enc = imcvt_amd.DeviceEncoder()
b = enc.make_batch([torch.from_numpy(a).cuda() for a in host], 0)
enc.encode(b); torch.cuda.synchronize()
out = {}
def run(tag, k=2):
    v = []
    for _ in range(k):
        enc.encode(b); torch.cuda.synchronize(); v.append(round(enc.last_kernel_ms(), 1))
    out[tag] = v; print(tag, v, flush=True)
run("1_fresh", 3)
streams = [torch.cuda.Stream() for _ in range(4)]
for s in streams:
    with torch.cuda.stream(s): torch.zeros(1024, device="cuda").add_(1)
torch.cuda.synchronize()
run("2_after_4_extra_streams")
pins = [torch.empty(8 << 20, dtype=torch.uint8, pin_memory=True) for _ in range(7)]
d = torch.empty(8 << 20, dtype=torch.uint8, device="cuda")
for p in pins: d.copy_(p, non_blocking=True)
torch.cuda.synchronize()
run("3_after_pinned_buffers")
enc2 = imcvt_amd.DeviceEncoder()
run("4_after_second_context")
enc2.close()
run("5_second_context_closed")
hevc.HEVCImageEncoderBatch(host[:32], 0)
run("6_after_small_host_batch")
r = hevc.HEVCImageEncoderBatch(host, 0, copy=False); out["host_kernel_ms"] = round(hevc.transfer_stats()["kernel_ms"], 1); del r
run("7_after_full_host_batch")
imcvt_amd.load_library().imcvt_hevc_shutdown()
run("8_after_shutdown")
print(json.dumps({"probe": "slow_process", **out}))
