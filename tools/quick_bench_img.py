#!/usr/bin/env python3
"""[developer check script — TEST INFRASTRUCTURE like tests/] Throughput probe on a natural picture: n copies of a PGM file (default
tests/golden/p5_gray.pgm, 300x263) in one device batch, 3 launches; the first stream is checked against the committed golden one when there is one.
usage: python tools/quick_bench_img.py [n q file]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, imcvt_amd
n, q = (int(a) for a in (sys.argv[1:3] + ["1024", "0"][len(sys.argv) - 1:]))
path = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "tests", "golden", "p5_gray.pgm")
raw = open(path, "rb").read()
parts = raw.split(b"\n", 3)
w, h = (int(v) for v in parts[1].split())
img = np.frombuffer(parts[3], np.uint8, w * h).reshape(h, w)
enc = imcvt_amd.DeviceEncoder()
batch = enc.make_batch([torch.from_numpy(img.copy()).cuda() for _ in range(n)], q)
ms = []
for _ in range(3):
    enc.encode(batch); torch.cuda.synchronize(); ms.append(enc.last_kernel_ms())
res = enc.results(batch)
print(f"{n} x {w}x{h} q{q} ({os.path.basename(path)}): kernel ms {[round(m, 1) for m in ms]}  best {w * h * n / min(ms) / 1e3:.2f} Mpx/s  stream {len(res[0][0])} B  all equal {all(r[0] == res[0][0] for r in res)}")
