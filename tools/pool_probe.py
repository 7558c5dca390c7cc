#!/usr/bin/env python3
"""[developer check script — TEST INFRASTRUCTURE like tests/] Launch-shape probe: n frames of w x h at qpd6 q under a list of
(mains, helpers) shapes; kernel ms per shape, digests compared between shapes (and with the first shape's).
usage: pool_probe.py w h n q  m:h [m:h ...]      (0:0 = frames per workgroup, a:a = automatic)"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, imcvt_amd
from imcvt_amd import synth
w, h, n, q = (int(a) for a in sys.argv[1:5])
enc = imcvt_amd.DeviceEncoder()
frames = [torch.from_numpy(synth.syn(w, h, s)).cuda() for s in range(n)]
batch = enc.make_batch(frames, q)
ref = None
for sh in sys.argv[5:]:
    if sh == "a:a":
        enc.set_shape(0, 0); enc.set_team(0)
    else:
        m, hp = (int(v) for v in sh.split(":"))
        if m == 0:
            enc.set_shape(0, 0); enc.set_team(1)
        else:
            enc.set_team(0); enc.set_shape(m, hp)
    ms = []
    for _ in range(int(os.environ.get("PP_LAUNCHES", "2"))):
        enc.encode(batch); torch.cuda.synchronize(); ms.append(enc.last_kernel_ms())
    dig = hashlib.sha256(b"".join(s for s, _ in enc.results(batch))).hexdigest()[:16]
    if ref is None:
        ref = dig
    print(f"{n} x {w}x{h} q{q} shape {sh:>9s} -> {enc.last_shape()}: kernel ms {[round(v, 1) for v in ms]}  {w * h * n / min(ms) / 1e3:7.2f} Mpx/s  digest {dig} {'same' if dig == ref else 'DIFFERENT'}", flush=True)
