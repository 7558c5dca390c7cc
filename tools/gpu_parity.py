#!/usr/bin/env python3
"""[developer check script — TEST INFRASTRUCTURE like tests/: uses the CPU checker under oracle/ to verify what it times] Quick GPU parity run: HIP path vs the CPU checker on small seeded inputs, then golden digests for a 1080p frame."""
import hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import imcvt_amd
from oracle import oracle, synth

print(imcvt_amd.load_library().imcvt_hevc_version().decode())
bad = 0
cases = [("flat32", synth.flat(32, 32, 0)), ("syn33", synth.syn(33, 31, 2)), ("noise64", synth.noise(64, 64, 1)), ("syn100", synth.syn(100, 70, 3)),
         ("syn256", synth.syn(256, 128, 0))]
for name, img in cases:
    for q in (0, 2, 4):
        t = time.time(); b, r, _ = imcvt_amd.HEVCImageEncoder(img, q); tg = time.time() - t
        b2, r2, _ = oracle.cpu_encode(img, q)
        ok = (b == b2) and bool((r == r2).all())
        bad += not ok
        print(f"{name} q{q} gpu {len(b)} cpu {len(b2)} {'OK' if ok else 'MISMATCH'} {tg*1e3:.1f} ms", flush=True)
# batch of different frames
imgs = [synth.syn(64 + 8 * i, 48 + 4 * i, i) for i in range(12)]
res = imcvt_amd.HEVCImageEncoderBatch(imgs, 1)
for i, (b, r, _) in enumerate(res):
    b2, r2, _ = oracle.cpu_encode(imgs[i], 1)
    ok = (b == b2) and bool((r == r2).all()); bad += not ok
print("batch12", "OK" if bad == 0 else "MISMATCH")
if "--big" in sys.argv:
    kat = {(e["input"].get("kind"), e["input"].get("w"), e["input"].get("arg"), e["qpd6"]): e for e in json.load(open(os.path.join(ROOT, "tests/golden/hevc_kat.json")))}
    img = synth.syn(1920, 1080, 0)
    t = time.time(); b, r, _ = imcvt_amd.HEVCImageEncoder(img, 0); tg = time.time() - t
    e = kat[("syn", 1920, 0, 0)]
    ok = hashlib.sha256(b).hexdigest() == e["sha256"] and hashlib.sha256(r.tobytes()).hexdigest() == e["rcon_sha256"]
    bad += not ok
    print(f"1080p q0: {len(b)} bytes {'OK' if ok else 'MISMATCH'} in {tg:.2f} s -> {1920*1080/tg/1e6:.3f} Mpx/s", flush=True)
print("FAILURES:", bad)
sys.exit(1 if bad else 0)
