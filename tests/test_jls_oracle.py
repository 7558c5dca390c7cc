"""CPU tests of the JPEG-LS checker (oracle/jls_oracle.c): against the golden vectors generated from the compiled
reference (tests/golden/jls_kat.json <- tests/golden/make_jls_golden.py), and against the reference itself on seeded
inputs wherever oracle/_ref exists."""
import base64
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import ROOT, kat_input

KAT = json.load(open(os.path.join(ROOT, "tests", "golden", "jls_kat.json")))


def jls_input(spec):
    from oracle import synth
    if spec["kind"] == "rgb_syn":
        w, h, s = spec["w"], spec["h"], spec["arg"]
        return np.stack([synth.syn(w, h, s), synth.syn(w, h, s + 1)[::-1].copy(), synth.noise(w, h, s + 2) >> 2], axis=-1)
    return kat_input(spec)


def jls_id(e):
    s = e["input"]
    return (s.get("file") or f"{s['kind']}{s['w']}x{s['h']}a{s['arg']}") + f"-near{e['near']}"


SMALL = [e for e in KAT if e["input"].get("w", 0) < 1920]
LARGE = [e for e in KAT if e["input"].get("w", 0) >= 1920]


@pytest.mark.parametrize("e", SMALL + [e for e in LARGE if e["input"]["w"] == 1920 and e["near"] == 0], ids=jls_id)
def test_port_matches_reference_vectors(built, e):
    from oracle import oracle
    got = oracle.jls_port_encode(jls_input(e["input"]), e["near"])
    assert len(got) == e["bytes"] and hashlib.sha256(got).hexdigest() == e["sha256"]
    if "stream_b64" in e:
        assert got == base64.b64decode(e["stream_b64"])


def test_port_vs_compiled_reference_seeded(built):
    from oracle import oracle
    if not oracle.jls_have_ref():
        pytest.skip("compiled reference only exists in the dev container")
    rng = np.random.default_rng(77)
    for i in range(40):
        h, w = int(rng.integers(1, 70)), int(rng.integers(1, 70))
        kind = i % 5
        shape = (h, w, 3) if kind == 4 else (h, w)
        img = (rng.integers(0, 256, shape) if kind in (0, 4) else np.clip(rng.normal(128, 6, shape), 0, 255) if kind == 1
               else np.full(shape, int(rng.integers(0, 256))) if kind == 2 else (rng.integers(0, 2, shape) * 255)).astype(np.uint8)
        near = int(rng.integers(0, 5))
        assert oracle.jls_port_encode(img, near) == oracle.jls_ref_encode(img, near), (i, shape, near)
