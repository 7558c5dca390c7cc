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


def emu_encode_team(lib, imgs, q, team_size, nteams):
    n = len(imgs)
    imgs = [np.ascontiguousarray(a) for a in imgs]
    outs = [np.zeros(2 * (a.shape[1] + 32) * (a.shape[0] + 32) + 65536, np.uint8) for a in imgs]
    rcs = [np.zeros(((a.shape[0] + 31) // 32 * 32) * ((a.shape[1] + 31) // 32 * 32), np.uint8) for a in imgs]
    P = u8p * n
    ys = (C.c_int * n)(*[a.shape[0] for a in imgs]); xs = (C.c_int * n)(*[a.shape[1] for a in imgs]); lens = (C.c_int * n)()
    lib.hostemu_HEVCImageEncoderTeam.restype = C.c_int
    assert lib.hostemu_HEVCImageEncoderTeam(n, P(*[o.ctypes.data_as(u8p) for o in outs]), P(*[a.ctypes.data_as(u8p) for a in imgs]),
                                            P(*[r.ctypes.data_as(u8p) for r in rcs]), ys, xs, q, lens, team_size, nteams) == 0
    return [(outs[i][:lens[i]].tobytes(), rcs[i]) for i in range(n)]


@pytest.mark.parametrize("team_size,nteams", [(3, 1), (2, 1), (3, 2), (2, 3)])
@pytest.mark.parametrize("q", [0, 4])
def test_team_modes_match_golden(hostemu, q, team_size, nteams):
    """A frame encoded by a TEAM of workgroups (hevc_frame.h: the main workgroup walks the 8x8 CUs, helper workgroups evaluate
    the 16x16 / 32x32 candidate sets from the posted entry states) gives the reference's bytes; several teams pull frames
    from one queue.  The emulated workgroups run concurrently (fibers), so the request / result hand-offs are real."""
    es = [e for e in OVF if e["qpd6"] == q]
    res = emu_encode_team(hostemu, [kat_input(e["input"]) for e in es], q, team_size, nteams)
    for e, (stream, rcon) in zip(es, res):
        assert hashlib.sha256(stream).hexdigest() == e["sha256"], kat_id(e)
        assert hashlib.sha256(rcon.tobytes()).hexdigest() == e["rcon_sha256"], kat_id(e)


def test_team_mode_natural_image(hostemu):
    # 10 x 9 CTUs of the reference's own sample picture: helpers read their borders from the reconstruction plane across CTU rows
    e = next(e for e in kat_entries() if e["input"].get("file") == "p5_gray.pgm" and e["qpd6"] == 4)
    (stream, rcon), = emu_encode_team(hostemu, [kat_input(e["input"])], 4, 3, 1)
    assert hashlib.sha256(stream).hexdigest() == e["sha256"] and hashlib.sha256(rcon.tobytes()).hexdigest() == e["rcon_sha256"]


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
