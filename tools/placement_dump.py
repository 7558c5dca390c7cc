#!/usr/bin/env python3
"""[developer check script — TEST INFRASTRUCTURE like tests/] Where the workgroups of the bench launch land: per workgroup the compute unit, the SIMD of each wavefront, the arrival index, the role; per
frame the end time and the CTUs its main workgroup ran at raised priority — in the fast state and in the slow one (after a 200-frame launch with IMCVT_HEVC_NO_REWARM=1).
usage: python tools/placement_dump.py out.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, imcvt_amd
from imcvt_amd import synth
n = 512
imgs = [torch.from_numpy(synth.syn(1920, 1080, s)).cuda() for s in range(n)]
enc = imcvt_amd.DeviceEncoder()
b = enc.make_batch(imgs, 0); b200 = enc.make_batch(imgs[:200], 0)
fclk = torch.zeros(4 * n + 4 * 1024, dtype=torch.int64, device="cuda")
enc.lib.imcvt_hevc_set_frame_clock(enc.ctx, fclk.data_ptr())
enc.encode(b); torch.cuda.synchronize()
dump = []
def launch(tag):
    fclk.zero_(); enc.encode(b); torch.cuda.synchronize()
    a = fclk.cpu().numpy(); fc = a[:4 * n].reshape(n, 4); wg = a[4 * n:].reshape(1024, 4); t0 = int(fc[:, 0].min())
    frames = [dict(f=i, blk=int(fc[i, 2] & 0xFFFFFFFF), cu=int((fc[i, 2] >> 32) & 0xFFFF), raised=int((fc[i, 2] >> 48) & 0xFFFF), end_ms=(int(fc[i, 1]) - t0) / 1e5, kept=int(fc[i, 3] & 0xFFFF)) for i in range(n)]
    wgs = [dict(blk=bk, cu=int(wg[bk, 2]) & 0xFFFF, main=int(wg[bk, 2] >> 32) & 1, simd=[(int(wg[bk, 1]) >> (8 * k)) & 3 for k in range(3)], arrival=(int(wg[bk, 1]) >> 56) & 0x7F, served=int(wg[bk, 0]) & 0xFFFFFFFF) for bk in range(1024) if wg[bk, 3] != 0]
    dump.append(dict(tag=tag, kernel_ms=round(enc.last_kernel_ms(), 1), frames=frames, wgs=wgs)); print(tag, dump[-1]["kernel_ms"], flush=True)
for i in range(3): launch("fast")
os.environ["IMCVT_HEVC_NO_REWARM"] = "1"
enc.encode(b200); torch.cuda.synchronize()
for i in range(3): launch("slow")
json.dump(dump, open(sys.argv[1], "w"))
