#!/usr/bin/env python3
"""Regenerate tests/golden/ from the REAL reference (oracle/_ref, built by oracle/Makefile from
/root/reference).  Runs only in the dev container; the outputs are committed data:

  p4_gray.pgm / p5_gray.pgm / p6_green.pgm   pixel arrays the reference's own PNM loader
                                             (src/imageio_pnm.c:73) returns for image/P4,P5,P6.pnm
                                             (P6: the green channel, src/imageio_hevc.c:21-27)
  hevc_kat.json                              (input, qpd6) -> stream length + SHA-256 + recon SHA-256
  *.h265                                     a few whole reference streams for byte-level diffs

usage:  python tests/golden/make_golden.py [--large]     (--large adds the 1080p x8 and 4K digests, ~5 min on 8 cores)
"""
import ctypes as C
import hashlib
import json
import os
import sys
from multiprocessing import Pool

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle, synth  # noqa: E402

REF_IMG = "/root/reference/image"


def load_with_reference_pnm(path):
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_pnm.so"))
    lib.loadPNMImageFile.restype = C.POINTER(C.c_ubyte)
    lib.loadPNMImageFile.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    rgb, h, w = C.c_int(), C.c_uint32(), C.c_uint32()
    p = lib.loadPNMImageFile(path.encode(), C.byref(rgb), C.byref(h), C.byref(w))
    assert p
    n = h.value * w.value * (3 if rgb.value else 1)
    a = np.ctypeslib.as_array(p, shape=(n,)).copy()
    return a.reshape(h.value, w.value, 3) if rgb.value else a.reshape(h.value, w.value)


def gen_input(spec):
    k = spec["kind"]
    if k == "file":
        with open(os.path.join(HERE, spec["file"]), "rb") as f:
            data = f.read()
        # our fixtures are always "P5\n<w> <h>\n255\n" + pixels
        hdr = data.split(b"\n", 3)
        w, h = map(int, hdr[1].split())
        return np.frombuffer(hdr[3], dtype=np.uint8, count=w * h).reshape(h, w)
    return getattr(synth, k)(spec["w"], spec["h"], spec["arg"])


def run_one(job):
    spec, q = job
    img = gen_input(spec)
    stream, rcon, _ = oracle.ref_encode(img, q)
    return dict(input=spec, qpd6=q, pixels_sha256=hashlib.sha256(img.tobytes()).hexdigest(), bytes=len(stream),
                sha256=hashlib.sha256(stream).hexdigest(), rcon_sha256=hashlib.sha256(rcon.tobytes()).hexdigest())


def main():
    large = "--large" in sys.argv
    oracle.build("ref")
    # pixel fixtures from the reference's own sample images
    p4 = load_with_reference_pnm(f"{REF_IMG}/P4.pnm")
    p5 = load_with_reference_pnm(f"{REF_IMG}/P5.pnm")
    p6 = load_with_reference_pnm(f"{REF_IMG}/P6.pnm")[:, :, 1]
    for name, a in (("p4_gray.pgm", p4), ("p5_gray.pgm", p5), ("p6_green.pgm", p6)):
        with open(os.path.join(HERE, name), "wb") as f:
            f.write(synth.pgm_bytes(np.ascontiguousarray(a)))

    small = [dict(kind="file", file="p4_gray.pgm"), dict(kind="file", file="p5_gray.pgm"),
             dict(kind="flat", w=64, h=64, arg=128), dict(kind="flat", w=32, h=32, arg=0),
             dict(kind="noise", w=64, h=64, arg=1), dict(kind="syn", w=256, h=128, arg=0),
             dict(kind="syn", w=33, h=31, arg=2), dict(kind="syn", w=100, h=70, arg=3),
             dict(kind="noise", w=40, h=72, arg=9), dict(kind="syn", w=64, h=32, arg=5)]
    jobs = [(s, q) for s in small for q in range(5)] + [(dict(kind="file", file="p6_green.pgm"), 0)]
    if large:
        jobs += [(dict(kind="syn", w=1920, h=1080, arg=s), 0) for s in range(8)]
        jobs += [(dict(kind="syn", w=1920, h=1080, arg=0), 4), (dict(kind="syn", w=3840, h=2160, arg=0), 0)]
    with Pool(8) as pool:
        kat = pool.map(run_one, jobs, chunksize=1)
    path = os.path.join(HERE, "hevc_kat.json")
    if not large and os.path.exists(path):      # keep previously generated large entries
        old = [e for e in json.load(open(path)) if e["input"].get("w", 0) >= 1920]
        kat += old
    json.dump(kat, open(path, "w"), indent=1)

    # whole streams for byte-level diffing
    for spec, q, name in ((small[0], 0, "p4_q0.h265"), (small[1], 4, "p5_q4.h265"), (small[6], 0, "syn33x31s2_q0.h265"),
                          (small[4], 2, "noise64s1_q2.h265")):
        stream, _, _ = oracle.ref_encode(gen_input(spec), q)
        open(os.path.join(HERE, name), "wb").write(stream)
    print("wrote", len(kat), "known-answer entries")


if __name__ == "__main__":
    main()
