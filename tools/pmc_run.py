#!/usr/bin/env python3
"""Small fixed workload for counter collection: n frames of w x h, one launch (plus one warm-up), followed by a
calibration pass of known HBM byte count (a 1 GiB device copy: 1 GiB read + 1 GiB written, far beyond L2 + MALL)
so that FETCH_SIZE / WRITE_SIZE can be put on an absolute scale as the MI355X guide asks."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, imcvt_amd
from imcvt_amd import synth
w, h, n, q = (int(a) for a in (sys.argv[1:5] + ["256", "128", "768", "0"][len(sys.argv) - 1:]))
enc = imcvt_amd.DeviceEncoder()
frames = [torch.from_numpy(synth.syn(w, h, s % 16)).cuda() for s in range(n)]
batch = enc.make_batch(frames, q)
enc.encode(batch); torch.cuda.synchronize()
enc.encode(batch); torch.cuda.synchronize()
print("kernel ms", enc.last_kernel_ms(), "ctus", ((w + 31) // 32) * ((h + 31) // 32) * n)
src = torch.empty(1 << 28, dtype=torch.int32, device="cuda").fill_(7)      # 1 GiB
dst = torch.empty_like(src)
torch.cuda.synchronize()
dst.copy_(src); torch.cuda.synchronize()
