#!/usr/bin/env python3
"""[developer check script] Hand-off stress for teams: many frames of very different sizes and content (uneven load on the mailboxes),
several launches, team of 3 / team of 2 against frame-per-workgroup launches of the same batch — every stream and reconstruction
must be identical.  usage: tools/team_stress.py [frames] [launches]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, imcvt_amd
n, reps = (int(a) for a in (sys.argv[1:3] + ["300", "3"][len(sys.argv) - 1:]))
rng = np.random.default_rng(2024)
imgs = []
for i in range(n):
    h, w = int(rng.integers(1, 420)), int(rng.integers(1, 640))
    k = i % 4
    a = (rng.integers(0, 256, (h, w)) if k == 0 else np.clip(rng.normal(128, 40, (h, w)), 0, 255) if k == 1
         else (np.add.outer(np.arange(h) * 2, np.arange(w) * 3) % 256) if k == 2 else np.full((h, w), int(rng.integers(0, 256))))
    imgs.append(torch.from_numpy(a.astype(np.uint8)).cuda())
qs = [i % 5 for i in range(n)]
enc = imcvt_amd.DeviceEncoder()
batch = enc.make_batch(imgs, qs)
enc.set_team(1); enc.encode(batch); ref = enc.results(batch)
bad = 0
for team in (3, 2):
    enc.set_team(team)
    for r in range(reps):
        enc.encode(batch); got = enc.results(batch)
        assert enc.last_team()[0] == team
        d = sum(1 for (s, rc), (s2, rc2) in zip(got, ref) if s != s2 or not (rc == rc2).all())
        bad += d
        print(f"team {team} launch {r}: {d} of {n} frames differ, kernel {enc.last_kernel_ms():.0f} ms", flush=True)
print("FAILURES:", bad)
sys.exit(1 if bad else 0)
