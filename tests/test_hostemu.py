"""Logic tests of the DEVICE SOURCE on CPU: imcvt_amd/csrc/hevc_{core,frame}.h compiled for the host by
tests/hostemu (a wavefront = a serial loop over 64 lanes).  This is a test harness only — it is not shipped and is
not a fallback; what it buys is that every decision branch of the kernel is checked against the golden vectors
without a GPU, so the -m gpu tests only have to expose synchronisation bugs."""
import ctypes as C
import hashlib

import numpy as np
import pytest

from conftest import kat_entries, kat_id, kat_input

u8p = C.POINTER(C.c_ubyte)
# P5 at q0 and q4 together exercise every branch of the CU search; keep the CPU suite short
PICK = [e for e in kat_entries() if e["input"].get("w", 0) < 1920
        and not (e["input"].get("file") == "p5_gray.pgm" and e["qpd6"] in (1, 2, 3))
        and not (e["input"].get("file") == "p6_green.pgm")
        and not (e["input"].get("w") == 256 and e["qpd6"] in (1, 2, 3))]


def emu_encode(lib, img, q):
    h, w = img.shape
    hp, wp = (h + 31) // 32 * 32, (w + 31) // 32 * 32
    out = np.zeros(2 * (w + 32) * (h + 32) + 65536, np.uint8)
    rc = np.zeros(hp * wp, np.uint8)
    ys, xs = C.c_int(h), C.c_int(w)
    n = lib.hostemu_HEVCImageEncoder(out.ctypes.data_as(u8p), np.ascontiguousarray(img).ctypes.data_as(u8p), rc.ctypes.data_as(u8p),
                                     C.byref(ys), C.byref(xs), q, None, 0)
    return out[:n].tobytes(), rc.reshape(hp, wp)


@pytest.mark.parametrize("e", PICK, ids=kat_id)
def test_device_source_matches_golden(hostemu, e):
    stream, rcon = emu_encode(hostemu, kat_input(e["input"]), e["qpd6"])
    assert len(stream) == e["bytes"]
    assert hashlib.sha256(stream).hexdigest() == e["sha256"]
    assert hashlib.sha256(rcon.tobytes()).hexdigest() == e["rcon_sha256"]


OVF = [e for e in PICK if e["input"].get("w", 999) <= 100 or e["input"].get("file") == "p4_gray.pgm"]


@pytest.mark.parametrize("e", OVF, ids=kat_id)
def test_ring_overflow_path_matches_golden(hostemu_ovf, e):
    # trial coders whose byte ring overflows are repeated on the safe path (hevc_core.h run_trial): same streams
    stream, rcon = emu_encode(hostemu_ovf, kat_input(e["input"]), e["qpd6"])
    assert hashlib.sha256(stream).hexdigest() == e["sha256"]
    assert hashlib.sha256(rcon.tobytes()).hexdigest() == e["rcon_sha256"]


@pytest.mark.parametrize("e", OVF, ids=kat_id)
def test_out_of_line_escape_path_matches_golden(hostemu_esc, e):
    # remaining-level code words longer than the inline slots take tok_escape (hevc_core.h tokg_b): same streams
    stream, rcon = emu_encode(hostemu_esc, kat_input(e["input"]), e["qpd6"])
    assert hashlib.sha256(stream).hexdigest() == e["sha256"]
    assert hashlib.sha256(rcon.tobytes()).hexdigest() == e["rcon_sha256"]


@pytest.mark.parametrize("e", OVF, ids=kat_id)
def test_token_row_overflow_path_matches_golden(hostemu_row, e):
    # a pass whose group tokens do not fit the lanes' LDS rows counts and writes them the plain way (hevc_core.h p1_run_t step 3)
    stream, rcon = emu_encode(hostemu_row, kat_input(e["input"]), e["qpd6"])
    assert hashlib.sha256(stream).hexdigest() == e["sha256"]
    assert hashlib.sha256(rcon.tobytes()).hexdigest() == e["rcon_sha256"]


def test_sizes_beyond_8192_are_cropped_like_the_reference(hostemu):
    # :1580-1581 pads min(dim, 8192) while the source keeps its own stride (:1621): one CTU row / column of 257 CTUs' worth
    from oracle import oracle, synth
    for h, w in ((3, 8200), (8200, 3)):
        img = synth.noise(w, h, 5)
        stream, rcon = emu_encode(hostemu, img, 3) if max(h, w) <= 8192 else _emu_big(hostemu, img, 3)
        ws, wr, dims = oracle.cpu_encode(img, 3)
        assert stream == ws and (rcon == wr).all() and rcon.shape == dims


def _emu_big(lib, img, q):
    h, w = img.shape
    hp, wp = (min(h, 8192) + 31) // 32 * 32, (min(w, 8192) + 31) // 32 * 32
    out = np.zeros(2 * (w + 32) * (h + 32) + 65536, np.uint8)
    rc = np.zeros(hp * wp, np.uint8)
    ys, xs = C.c_int(h), C.c_int(w)
    n = lib.hostemu_HEVCImageEncoder(out.ctypes.data_as(u8p), np.ascontiguousarray(img).ctypes.data_as(u8p), rc.ctypes.data_as(u8p),
                                     C.byref(ys), C.byref(xs), q, None, 0)
    assert (ys.value, xs.value) == (hp, wp)
    return out[:n].tobytes(), rc.reshape(hp, wp)


def test_lds_budget(hostemu):
    # two workgroups per CU need <= 80 KiB each (160 KiB LDS per CU)
    assert hostemu.hostemu_shm_bytes() <= 80 * 1024
