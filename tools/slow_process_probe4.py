#!/usr/bin/env python3
"""[developer check script — TEST INFRASTRUCTURE like tests/] The re-warm before a full 192-thread launch that follows a launch of another kind (hevc_hip.hip imcvt_hevc_encode_device): with it and
without it (IMCVT_HEVC_NO_REWARM=1); resident launches and the host-pointer path.   usage: python tools/slow_process_probe4.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, imcvt_amd
from imcvt_amd import synth, hevc
n = 512
host = [synth.syn(1920, 1080, s) for s in range(n)]
imgs = [torch.from_numpy(a).cuda() for a in host]
enc = imcvt_amd.DeviceEncoder()
b = enc.make_batch(imgs, 0)
small = {k: enc.make_batch(imgs[:k], 0) for k in (1, 64, 200)}
enc.encode(b); torch.cuda.synchronize()
out = {}
def run(tag, k=3):
    v = []
    for _ in range(k):
        enc.encode(b); torch.cuda.synchronize(); v.append(round(enc.last_kernel_ms(), 1))
    out[tag] = v; print(tag, v, flush=True)
run("1_fresh")
for k in (200, 64, 1):
    enc.encode(small[k]); torch.cuda.synchronize()
    run(f"2_after_a_{k}_frame_launch_(re-warm_on)", 2)
os.environ["IMCVT_HEVC_NO_REWARM"] = "1"
enc.encode(small[200]); torch.cuda.synchronize()
run("3_after_a_200_frame_launch_(re-warm_off)", 2)
del os.environ["IMCVT_HEVC_NO_REWARM"]
enc.encode(small[200]); torch.cuda.synchronize()
run("4_after_a_200_frame_launch_(re-warm_on_again)", 2)
hevc.HEVCImageEncoderBatch(host[:32], 0)
hk = []
for _ in range(3):
    t0 = time.perf_counter(); r = hevc.HEVCImageEncoderBatch(host, 0, copy=False); dt = (time.perf_counter() - t0) * 1e3
    hk.append((round(hevc.transfer_stats()["kernel_ms"], 1), round(dt, 1))); del r
out["5_host_pointer_batches_(kernel_ms, wall_ms)"] = hk; print("host", hk, flush=True)
run("6_resident_after_the_host_batches", 2)
print(json.dumps({"probe": "rewarm", **out}))
