#!/usr/bin/env python3
"""Rate-distortion table of the H.265 path on the synthetic inputs (GPU box): bytes, bits/pixel and PSNR of the returned
reconstruction for qpd6 = 0..4.  usage: python tools/rd_table.py [w h seeds] > profiles/rNN_rd_table.json"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imcvt_amd as amd
from imcvt_amd import quality, synth

w, h, seeds = (int(v) for v in (sys.argv[1:4] + ["1920", "1080", "4"][len(sys.argv) - 1:]))
rows = []
for q in range(5):
    imgs = [synth.syn(w, h, s) for s in range(seeds)]
    res = amd.HEVCImageEncoderBatch(imgs, q)
    pts = [quality.rd_point(img, s, r) for img, (s, r, _) in zip(imgs, res)]
    rows.append({"qpd6": q, "qp": 4 + 6 * q, "frames": seeds, "bytes_per_frame": sum(p["bytes"] for p in pts) / seeds,
                 "bpp": sum(p["bpp"] for p in pts) / seeds, "psnr_db": sum(p["psnr_db"] for p in pts) / seeds})
print(json.dumps({"input": "syn(%d,%d,0..%d)" % (w, h, seeds - 1), "points": rows}, indent=1))
