#!/usr/bin/env python3
"""[developer check script — TEST INFRASTRUCTURE like tests/: uses the CPU checker under oracle/ to verify what it times] Quick throughput probe: n frames of w x h (default 1024 x 512x256, q0), 3 launches; first frames checked against the CPU checker."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, imcvt_amd
from oracle import oracle, synth
w, h, n, q = (int(a) for a in (sys.argv[1:5] + ["512", "256", "1024", "0"][len(sys.argv) - 1:]))
enc = imcvt_amd.DeviceEncoder()
imgs = [synth.syn(w, h, s % 16) for s in range(min(n, 16))]
frames = [torch.from_numpy(imgs[s % 16]).cuda() for s in range(n)]
batch = enc.make_batch(frames, q)
ms = []
for _ in range(int(os.environ.get("QB_LAUNCHES", "3"))):
    enc.encode(batch); torch.cuda.synchronize(); ms.append(enc.last_kernel_ms())
res = enc.results(batch)
bad = 0
for i in range(min(n, 2)):
    want, wr, _ = oracle.cpu_encode(imgs[i], q)
    bad += not (res[i][0] == want and (res[i][1] == wr).all())
best = min(ms)
print(f"team {enc.last_team()}", end="  ")
print(f"{n} x {w}x{h} q{q}: kernel ms {[round(m, 1) for m in ms]}  best {w * h * n / best / 1e3:.2f} Mpx/s  parity {'OK' if not bad else 'MISMATCH'}")
sys.exit(1 if bad else 0)
