#!/usr/bin/env python3
"""[developer check script — TEST INFRASTRUCTURE like tests/] Decision trace of one frame (8 ints per CU: y, x, size, kind, mode(s), cost, 0, 0) from the
library named by IMCVT_HEVC_LIB, next to the CPU checker's result.  usage: trace_dump.py kind w h arg q out.npy   (kind: syn | noise | flat)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, imcvt_amd
from oracle import oracle, synth
kind, w, h, arg, q, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
img = getattr(synth, kind)(w, h, arg)
enc = imcvt_amd.DeviceEncoder()
enc.set_team(1)
cap = 8 * 37 * ((w + 31) // 32) * ((h + 31) // 32) + 64
tr = torch.zeros(cap, dtype=torch.int32, device="cuda")
enc.lib.imcvt_hevc_set_trace(enc.ctx, tr.data_ptr(), cap)
batch = enc.make_batch([torch.from_numpy(img).cuda()], q)
enc.encode(batch); torch.cuda.synchronize()
(s, r), = enc.results(batch)
ws, wr, _ = oracle.cpu_encode(img, q)
t = tr.cpu().numpy().reshape(-1, 8)
t = t[t[:, 2] != 0]
np.save(out, t)
print(f"{os.environ.get('IMCVT_HEVC_LIB', 'default')}: {len(s)} bytes (checker {len(ws)}) {'OK' if s == ws and (r == wr).all() else 'MISMATCH'}; {len(t)} CUs traced")
