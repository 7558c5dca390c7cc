"""CPU tests of the checker itself: the plain-C restatement (oracle/hevc_oracle.c) against the committed golden
vectors (generated from the real reference by tests/golden/make_golden.py) and, where oracle/_ref exists, against the
reference directly.  Reference behaviour under test: src/HEVCe/HEVCe.c:1569 HEVCImageEncoder."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

from conftest import ROOT, kat_entries, kat_id, kat_input
from oracle import oracle, synth

SMALL = [e for e in kat_entries() if e["input"].get("w", 0) < 1920]


def test_generators_pinned():
    # vectorised syn() == literal definition, and pixel digests of the golden inputs
    for (w, h, s) in [(33, 31, 2), (70, 40, 5)]:
        assert (synth.syn(w, h, s) == synth.syn_py(w, h, s)).all()
    assert hashlib.sha256(synth.syn(256, 128, 0).tobytes()).hexdigest().startswith("bedce3b48edac207")
    assert hashlib.sha256(synth.noise(64, 64, 1).tobytes()).hexdigest().startswith("b650bb4639e9ff5f")


@pytest.mark.parametrize("e", SMALL, ids=kat_id)
def test_port_matches_golden(built, e):
    img = kat_input(e["input"])
    assert hashlib.sha256(img.tobytes()).hexdigest() == e["pixels_sha256"]
    stream, rcon, _ = oracle.port_encode(img, e["qpd6"])
    assert len(stream) == e["bytes"]
    assert hashlib.sha256(stream).hexdigest() == e["sha256"]
    assert hashlib.sha256(rcon.tobytes()).hexdigest() == e["rcon_sha256"]


@pytest.mark.parametrize("name,spec,q", [("p4_q0.h265", dict(kind="file", file="p4_gray.pgm"), 0),
                                         ("p5_q4.h265", dict(kind="file", file="p5_gray.pgm"), 4),
                                         ("syn33x31s2_q0.h265", dict(kind="syn", w=33, h=31, arg=2), 0),
                                         ("noise64s1_q2.h265", dict(kind="noise", w=64, h=64, arg=1), 2)])
def test_port_matches_golden_streams(built, name, spec, q):
    want = open(os.path.join(ROOT, "tests", "golden", name), "rb").read()
    got, _, _ = oracle.port_encode(kat_input(spec), q)
    assert got == want


def test_header_bytes(built):
    # first bytes of P4 q0: VPS | SPS | dims | PPS | slice header
    got, _, _ = oracle.port_encode(kat_input(dict(kind="file", file="p4_gray.pgm")), 0)
    vps = bytes.fromhex("00 00 01 40 01 0c 01 ff ff 03 10 00 00 03 00 00 03 00 00 03 00 00 03 00 b4 f0 24")
    assert got[:27] == vps
    assert got[27:31] == bytes.fromhex("00 00 01 42") and got[48] == 0xB4
    assert got[49:58] == bytes.fromhex("a0 42 08 59 7e e4 68 1e d1")          # 32x32 dims + fixed flags
    assert got[58:69] == bytes.fromhex("00 00 01 44 01 c0 90 91 81 d9 20")    # PPS
    assert got[69:77] == bytes.fromhex("00 00 01 26 01 ac 16 de")             # slice header, qpd6=0


@pytest.mark.skipif(not oracle.have_ref(), reason="oracle/_ref not built (no /root/reference here)")
@pytest.mark.parametrize("seed", range(6))
def test_port_matches_reference_random(built, seed):
    rng = np.random.default_rng(seed)
    h, w = int(rng.integers(1, 80)), int(rng.integers(1, 80))
    kind = seed % 3
    img = (rng.integers(0, 256, (h, w)) if kind == 0 else
           np.clip(rng.normal(128, 20, (h, w)), 0, 255) if kind == 1 else
           (np.add.outer(np.arange(h) * 3, np.arange(w) * 2) % 256)).astype(np.uint8)
    q = seed % 5
    a, ra, _ = oracle.port_encode(img, q)
    b, rb, _ = oracle.ref_encode(img, q)
    assert a == b and (ra == rb).all()


@pytest.mark.skipif(not oracle.have_ref(), reason="oracle/_ref not built")
def test_port_tables_match_reference(built):
    """Generated transform matrices / CABAC tables equal the reference's exported tables (:391-464, :700-714)."""
    ref, port = oracle.ref_lib(), oracle.port_lib()
    port.oracle_transform_entry.restype = C.c_int
    for s, (n, sym) in enumerate([(4, "DST4_MAT"), (8, "DCT8_MAT"), (16, "DCT16_MAT"), (32, "DCT32_MAT")]):
        m = np.ctypeslib.as_array((C.c_int * (n * 32)).in_dll(ref, sym)).reshape(n, 32)
        for i in range(n):
            for k in range(n):
                assert port.oracle_transform_entry(s, i, k) == m[i, k]


def test_edge_sizes(built):
    # 1x1, ragged, exactly-32, one-over: padded dims and determinism of output w.r.t. rcon pre-fill
    for (h, w) in [(1, 1), (17, 21), (32, 32), (33, 65)]:
        img = synth.noise(w, h, h * w)
        a, r, (hp, wp) = oracle.port_encode(img, 1)
        assert (hp, wp) == ((h + 31) // 32 * 32, (w + 31) // 32 * 32)
        assert r.shape == (hp, wp) and len(a) > 80


def test_bench_workload_digests_are_complete_and_consistent():
    """tests/golden/bench512_kat.json pins every frame of BASELINE configs[3] (syn(1920,1080,0..511), qpd6 0) to the REAL
    reference (make_bench_golden.py).  Complete, and equal to the independently generated 1080p entries of hevc_kat.json."""
    import json
    from conftest import ROOT, kat_entries
    d = json.load(open(os.path.join(ROOT, "tests", "golden", "bench512_kat.json")))
    assert d["qpd6"] == 0 and d["input"] == {"kind": "syn", "w": 1920, "h": 1080}
    assert sorted(int(k) for k in d["frames"]) == list(range(512))
    seen = 0
    for e in kat_entries():
        i = e["input"]
        if i.get("kind") == "syn" and (i.get("w"), i.get("h")) == (1920, 1080) and e["qpd6"] == 0:
            f = d["frames"][str(i["arg"])]
            assert (f["bytes"], f["sha256"], f["rcon_sha256"]) == (e["bytes"], e["sha256"], e["rcon_sha256"])
            seen += 1
    assert seen >= 8
    assert len({f["sha256"] for f in d["frames"].values()}) == 512
