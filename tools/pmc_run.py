#!/usr/bin/env python3
"""Small fixed workload for counter collection: n frames of w x h, one launch (plus one warm-up)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, imcvt_amd
from oracle import synth
w, h, n, q = (int(a) for a in (sys.argv[1:5] + ["256", "128", "768", "0"][len(sys.argv) - 1:]))
enc = imcvt_amd.DeviceEncoder()
frames = [torch.from_numpy(synth.syn(w, h, s % 16)).cuda() for s in range(n)]
batch = enc.make_batch(frames, q)
enc.encode(batch); torch.cuda.synchronize()
enc.encode(batch); torch.cuda.synchronize()
print("kernel ms", enc.last_kernel_ms(), "ctus", ((w + 31) // 32) * ((h + 31) // 32) * n)
