"""Decoder-free quality tooling: PSNR of the reconstruction the encoder hands back."""
import math

import numpy as np
import pytest

from imcvt_amd import quality


def test_psnr_known_values():
    img = np.full((5, 7), 100, np.uint8)
    rc = np.full((32, 32), 100, np.uint8)                       # padded reconstruction
    assert quality.psnr(img, rc) == math.inf
    rc2 = rc.copy(); rc2[:5, :7] = 101
    assert quality.sse(img, rc2) == 35
    assert abs(quality.psnr(img, rc2) - 10 * math.log10(255 ** 2)) < 1e-12
    rc3 = rc.copy(); rc3[10:, 10:] = 0                          # differences in the padding do not count
    assert quality.psnr(img, rc3) == math.inf
    p = quality.rd_point(img, b"x" * 35, rc2)
    assert p["bytes"] == 35 and p["bpp"] == 8.0


def test_psnr_of_the_oracle_reconstruction_falls_with_qp():
    from oracle import oracle, synth
    img = synth.syn(64, 48, 3)
    vals = []
    for q in range(5):
        s, r, _ = oracle.cpu_encode(img, q)
        vals.append(quality.rd_point(img, s, r))
    assert all(a["psnr_db"] > b["psnr_db"] and a["bytes"] > b["bytes"] for a, b in zip(vals, vals[1:]))
    assert vals[0]["psnr_db"] > 45 and vals[4]["psnr_db"] > 25


@pytest.mark.gpu
def test_device_rd_points_equal_the_oracles(built):
    import imcvt_amd as amd
    from oracle import oracle, synth
    imgs = [synth.syn(96, 64, s) for s in range(3)]
    for q in (0, 2, 4):
        for img, (s, r, _) in zip(imgs, amd.HEVCImageEncoderBatch(imgs, q)):
            so, ro, _ = oracle.cpu_encode(img, q)
            assert quality.rd_point(img, s, r) == quality.rd_point(img, so, ro)
