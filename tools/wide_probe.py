#!/usr/bin/env python3
"""[developer check script — TEST INFRASTRUCTURE like tests/] Wide-workgroup probe: n frames of w x h at qpd6 q, launched with
256-thread pipe-wave workgroups and with 512-thread wide workgroups (imcvt_hevc_set_wide), interleaved; kernel ms per launch, digests compared.
usage: wide_probe.py w h q n [n ...]     (WP_LAUNCHES launches per setting, default 2)"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, imcvt_amd
from imcvt_amd import synth
w, h, q = (int(a) for a in sys.argv[1:4])
enc = imcvt_amd.DeviceEncoder()
for n in (int(a) for a in sys.argv[4:]):
    frames = [torch.from_numpy(synth.syn(w, h, s)).cuda() for s in range(n)]
    batch = enc.make_batch(frames, q)
    modes = (0, 1)
    ms = {m: [] for m in modes}; dig = {}; used = {}
    for _ in range(int(os.environ.get("WP_LAUNCHES", "2"))):
        for m in modes:
            enc.set_wide(-1 if m else 0)
            enc.encode(batch); torch.cuda.synchronize(); ms[m].append(enc.last_kernel_ms())
            used[m] = (enc.last_pipe(), enc.last_wide())
            dig[m] = hashlib.sha256(b"".join(s + r.tobytes() for s, r in enc.results(batch))).hexdigest()[:16]
    for m in modes:
        print(f"{n} x {w}x{h} q{q} wide {m} (ran with pipe wave / wide: {used[m]}) shape {enc.last_shape()}: kernel ms {[round(v, 1) for v in ms[m]]}  "
              f"{w * h * n / min(ms[m]) / 1e3:7.2f} Mpx/s  digest {dig[m]} {'same' if dig[m] == dig[modes[0]] else 'DIFFERENT'}", flush=True)
    del batch, frames
