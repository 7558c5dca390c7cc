#!/usr/bin/env python3
"""[developer check script — TEST INFRASTRUCTURE like tests/] 64 frames (one GPU's share of BASELINE configs[3] at N = 8): the planned wide pool (64 main + 64 partner + 112 helper workgroups, a sixteenth of
the compute units spare) against forced shapes that use every compute unit; kernel ms, interleaved.   usage: python tools/shape64_probe.py [reps]"""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, imcvt_amd
from imcvt_amd import synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
gold = json.load(open(os.path.join(ROOT, "tests", "golden", "bench512_kat.json")))["frames"]
enc = imcvt_amd.DeviceEncoder()
for n in (64, 60):
    imgs = [torch.from_numpy(synth.syn(1920, 1080, s)).cuda() for s in range(n)]
    b = enc.make_batch(imgs, 0)
    res = {}
    for r in range(reps):
        for name, shape in (("planned", (0, 0)), ("forced_128_helpers", (n, 2 * n)), ("forced_120_helpers", (n, 120)), ("forced_96_helpers", (n, 96))):
            enc.set_shape(*shape)
            enc.encode(b); torch.cuda.synchronize()
            res.setdefault(name, []).append((round(enc.last_kernel_ms(), 1), enc.last_shape(), enc.last_partners(), enc.last_resident()))
            lens = b["lens"].cpu().tolist()
            for i in (0, n - 1):
                assert hashlib.sha256(b["outs"][i][:lens[i]].cpu().numpy().tobytes()).hexdigest() == gold[str(i)]["sha256"], (name, i)
    enc.set_shape(0, 0)
    print(json.dumps({"probe": "shape64", "frames": n, **{k: {"ms": [v[0] for v in vs], "shape": list(vs[0][1]), "partners": vs[0][2], "resident": [v[3] for v in vs]} for k, vs in res.items()}}), flush=True)
    del b, imgs
