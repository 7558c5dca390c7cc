#!/usr/bin/env python3
"""Generates tests/golden/jls_kat.json from the REFERENCE JPEG-LS encoder (oracle/_ref/libref_jls.so, compiled from
/root/reference/src/imageio_jls.c where it lies): for seeded inputs (oracle/synth.py generators, gray and RGB) and the
committed sample images, NEAR = 0..4: output length and SHA-256, plus two whole small streams.  Dev container only."""
import base64, ctypes as C, hashlib, json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import synth
lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_jls.so"))
lib.writeJLSImageFile.restype = C.c_int
lib.writeJLSImageFile.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_uint32, C.c_uint32, C.c_int]

def ref_encode(img, near):
    rgb = img.ndim == 3
    h, w = img.shape[:2]
    with tempfile.NamedTemporaryFile(suffix=".jls", delete=False) as f:
        name = f.name
    assert lib.writeJLSImageFile(name.encode(), np.ascontiguousarray(img).tobytes(), int(rgb), h, w, near) == 0
    data = open(name, "rb").read(); os.unlink(name)
    return data

def pgm(name):
    d = open(os.path.join(ROOT, "tests", "golden", name), "rb").read().split(b"\n", 3)
    w, h = map(int, d[1].split())
    return np.frombuffer(d[3], np.uint8, w * h).reshape(h, w)

def rgb_syn(w, h, seed):
    return np.stack([synth.syn(w, h, seed), synth.syn(w, h, seed + 1)[::-1].copy(), synth.noise(w, h, seed + 2) >> 2], axis=-1)

cases = []
for kind, w, h, arg in (("flat", 64, 64, 128), ("flat", 32, 32, 0), ("flat", 17, 5, 255), ("noise", 64, 64, 1), ("syn", 256, 128, 0), ("syn", 33, 31, 2),
                        ("syn", 100, 70, 3), ("syn", 1, 1, 4), ("syn", 1, 9, 5), ("syn", 9, 1, 6), ("syn", 2, 2, 7)):
    cases.append((dict(kind=kind, w=w, h=h, arg=arg), getattr(synth, kind)(w, h, arg), range(5)))
for f in ("p4_gray.pgm", "p5_gray.pgm"):
    cases.append((dict(kind="file", file=f), pgm(f), range(5)))
cases.append((dict(kind="rgb_syn", w=48, h=40, arg=9), rgb_syn(48, 40, 9), range(5)))
cases.append((dict(kind="syn", w=1920, h=1080, arg=0), synth.syn(1920, 1080, 0), (0, 2)))
cases.append((dict(kind="syn", w=3840, h=2160, arg=0), synth.syn(3840, 2160, 0), (0,)))
out = []
for spec, img, nears in cases:
    for near in nears:
        data = ref_encode(img, near)
        e = dict(input=spec, near=near, bytes=len(data), sha256=hashlib.sha256(data).hexdigest())
        if len(data) < 800 and spec.get("kind") != "flat":
            e["stream_b64"] = base64.b64encode(data).decode()
        out.append(e)
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "jls_kat.json"), "w"), indent=0)
print(len(out), "vectors")
for e in out:
    if e["input"].get("w") in (1920, 3840) or e["input"].get("file") == "p5_gray.pgm": print(e["input"], e["near"], e["bytes"], e["sha256"][:16])
