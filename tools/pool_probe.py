#!/usr/bin/env python3
"""[developer check script — TEST INFRASTRUCTURE like tests/] Launch-shape probe: n frames of w x h at qpd6 q under a list of
(mains, helpers) shapes; kernel ms per shape, digests compared between shapes (and with the first shape's).
usage: pool_probe.py w h n q  m:h[:lim16:lim32:prio] ...      (0:0 = frames per workgroup, a:a = automatic)"""
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
        f = [int(v) for v in sh.split(":")]
        m, hp = f[0], f[1]
        enc.set_pool_tuning(*(f[2:5] if len(f) >= 5 else (-1, -1, -1)))
        if m == 0:
            enc.set_shape(0, 0); enc.set_team(1)
        else:
            enc.set_team(0); enc.set_shape(m, hp)
    ms = []; resid = []
    for _ in range(int(os.environ.get("PP_LAUNCHES", "2"))):
        enc.encode(batch); torch.cuda.synchronize(); ms.append(enc.last_kernel_ms()); resid.append(enc.last_resident())
    dig = hashlib.sha256(b"".join(s for s, _ in enc.results(batch))).hexdigest()[:16]
    if ref is None:
        ref = dig
    print(f"{n} x {w}x{h} q{q} shape {sh:>18s} -> {enc.last_shape()}: kernel ms {[round(v, 1) for v in ms]} resident {resid}  {w * h * n / min(ms) / 1e3:7.2f} Mpx/s  digest {dig} {'same' if dig == ref else 'DIFFERENT'}", flush=True)
