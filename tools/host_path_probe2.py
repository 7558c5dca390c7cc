#!/usr/bin/env python3
"""[developer check script — TEST INFRASTRUCTURE like tests/] The host-pointer path's launch takes 5.0 - 5.3 s where the resident one takes 4.8 s: with plain copies (IMCVT_HEVC_PLAIN_COPIES=1, the round-5 path:
copies, launch, copies on one stream) too?  512 bench frames; host path (staged / plain), then the same frames resident, interleaved.   usage: python tools/host_path_probe2.py [reps]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, imcvt_amd
from imcvt_amd import synth, hevc
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = 512
host = [synth.syn(1920, 1080, s) for s in range(n)]
enc = imcvt_amd.DeviceEncoder()
b = enc.make_batch([torch.from_numpy(a).cuda() for a in host], 0)
enc.encode(b); torch.cuda.synchronize()
hevc.HEVCImageEncoderBatch(host[:32], 0)
res = {}
def rec(k, v): res.setdefault(k, []).append(round(v, 1)); print(k, round(v, 1), flush=True)
for r in range(reps):
    for mode in ("staged", "plain"):
        if mode == "plain": os.environ["IMCVT_HEVC_PLAIN_COPIES"] = "1"
        else: os.environ.pop("IMCVT_HEVC_PLAIN_COPIES", None)
        t0 = time.perf_counter(); out = hevc.HEVCImageEncoderBatch(host, 0, copy=False); dt = (time.perf_counter() - t0) * 1e3
        rec(f"host_{mode}_kernel_ms", hevc.transfer_stats()["kernel_ms"]); rec(f"host_{mode}_wall_ms", dt)
        del out
        enc.encode(b); torch.cuda.synchronize(); rec("resident_kernel_ms", enc.last_kernel_ms())
print(json.dumps({"probe": "host_path_plain_vs_staged", **res}))
