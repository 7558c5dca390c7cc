#!/usr/bin/env python3
"""[developer check script — TEST INFRASTRUCTURE like tests/] Counter workload for the slow state of the full 192-thread launch: the bench batch once in the fast state, then a 200-frame
launch with IMCVT_HEVC_NO_REWARM=1, then the bench batch twice more (slow).  Run under rocprofv3 --pmc; tools/rocpd_pmc.py prints the dispatches in order."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, imcvt_amd
from imcvt_amd import synth
os.environ["IMCVT_HEVC_NO_REWARM"] = "1"
imgs = [torch.from_numpy(synth.syn(1920, 1080, s)).cuda() for s in range(512)]
enc = imcvt_amd.DeviceEncoder()
b = enc.make_batch(imgs, 0); b200 = enc.make_batch(imgs[:200], 0)
for tag in ("fast", "fast"):
    enc.encode(b); torch.cuda.synchronize(); print(tag, enc.last_kernel_ms(), flush=True)
enc.encode(b200); torch.cuda.synchronize(); print("200 frames", enc.last_kernel_ms(), flush=True)
for tag in ("slow", "slow"):
    enc.encode(b); torch.cuda.synchronize(); print(tag, enc.last_kernel_ms(), flush=True)
