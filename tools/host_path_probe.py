#!/usr/bin/env python3
"""[developer check script — TEST INFRASTRUCTURE like tests/] Why is the kernel of the host-pointer path 3-4 % slower than the resident one?  512 bench frames, kernel ms (HIP events),
interleaved on one box: (A) torch allocations, current stream; (B) the host path's slab layout ([img | out | rcon] per frame in one allocation), current stream; (C) torch allocations on a
non-blocking side stream; (E) a second context created later; (D) HEVCImageEncoderBatch itself.   usage: python tools/host_path_probe.py [frames] [reps]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import imcvt_amd
from imcvt_amd import synth, hevc
from imcvt_amd.hevc import imcvt_hevc_frame, stream_bound, padded

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda", 0)
W, H = 1920, 1080
host = [synth.syn(W, H, s) for s in range(n)]
big = torch.stack([torch.from_numpy(h) for h in host]).to(dev)
enc = imcvt_amd.DeviceEncoder()
bA = enc.make_batch([big[i] for i in range(n)], 0)

def a256(v): return (v + 255) // 256 * 256
per = a256(W * H) + a256(stream_bound(H, W)) + a256(padded(H) * padded(W))
slab = torch.empty(per * n + 4 * n + 256, dtype=torch.uint8, device=dev)
base = slab.data_ptr()
arr = (imcvt_hevc_frame * n)()
for i in range(n):
    o = base + i * per
    slab[i * per:i * per + W * H].copy_(big[i].reshape(-1))
    arr[i] = imcvt_hevc_frame(o, o + a256(W * H), o + a256(W * H) + a256(stream_bound(H, W)), base + per * n + 4 * i, H, W, 0)
bB = dict(n=n, frames=arr)
side = torch.cuda.Stream()
enc2 = imcvt_amd.DeviceEncoder()
res = {}
def rec(k, v): res.setdefault(k, []).append(round(v, 1)); print(k, round(v, 1), flush=True)
enc.encode(bA); torch.cuda.synchronize()
hevc.HEVCImageEncoderBatch(host[:64], 0)
for r in range(reps):
    enc.encode(bA); torch.cuda.synchronize(); rec("A_torch_layout", enc.last_kernel_ms())
    enc.encode(bB); torch.cuda.synchronize(); rec("B_slab_layout", enc.last_kernel_ms())
    enc.encode(bA, stream=side); torch.cuda.synchronize(); rec("C_side_stream", enc.last_kernel_ms())
    enc2.encode(bA); torch.cuda.synchronize(); rec("E_second_ctx", enc2.last_kernel_ms())
    for mode in ("0", "3"):
        os.environ["IMCVT_HEVC_FOLLOW"] = mode
        t0 = time.perf_counter()
        hevc.HEVCImageEncoderBatch(host, 0, copy=False)
        wall = (time.perf_counter() - t0) * 1e3
        rec("D_host_follow" + mode + "_kernel", hevc.transfer_stats()["kernel_ms"])
        rec("D_host_follow" + mode + "_wall", wall)
print(json.dumps({"probe": "host_path", "frames": n, **res}))
