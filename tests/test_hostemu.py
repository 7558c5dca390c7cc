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
    # the frame's progress record (hevc_frame.h publish_progress: what the host-pointer path follows while a launch runs) ends on "all CTU rows, done, n bytes"
    lib.hostemu_prog.restype = C.c_uint
    assert lib.hostemu_prog(0, 0) == (min(hp, 8192) // 32 | 0x80000000) and lib.hostemu_prog(0, 1) == n
    return out[:n].tobytes(), rc.reshape(hp, wp)


@pytest.mark.parametrize("e", PICK, ids=kat_id)
def test_device_source_matches_golden(hostemu, e):
    stream, rcon = emu_encode(hostemu, kat_input(e["input"]), e["qpd6"])
    assert len(stream) == e["bytes"]
    assert hashlib.sha256(stream).hexdigest() == e["sha256"]
    assert hashlib.sha256(rcon.tobytes()).hexdigest() == e["rcon_sha256"]


OVF = [e for e in PICK if e["input"].get("w", 999) <= 100 or e["input"].get("file") == "p4_gray.pgm"]


@pytest.mark.parametrize("e", OVF, ids=kat_id)
def test_lead_sink_exact_paths_match_golden(hostemu_ovf, e):
    # every trial's byte-level state from the real logic over its lead list, every winner's bytes by lane 0's walk (hevc_core.h leads_exact, resolve_leads): same streams
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


@pytest.mark.parametrize("e", OVF, ids=kat_id)
def test_vector_transform_build_matches_golden(hostemu_vec, e):
    # the N = 16 / 32 transforms as vector code (-DP1_MFMA=0, the A/B build): same streams as the matrix-instruction path
    stream, rcon = emu_encode(hostemu_vec, kat_input(e["input"]), e["qpd6"])
    assert hashlib.sha256(stream).hexdigest() == e["sha256"]
    assert hashlib.sha256(rcon.tobytes()).hexdigest() == e["rcon_sha256"]


# the small vectors, one natural picture and one 8 x 4-CTU synthetic one (the full list runs on the 192-thread build above and, with the pipe wave, on the GPU)
PIPE_PICK = OVF + [e for e in PICK if e not in OVF and ((e["input"].get("file") == "p5_gray.pgm" and e["qpd6"] == 4) or (e["input"].get("w") == 256 and e["qpd6"] == 0))]


@pytest.mark.parametrize("e", PIPE_PICK, ids=kat_id)
def test_pipe_wave_matches_golden(hostemu_pipe, e):
    # 256-thread workgroups: the NxN trial of every 8x8 CU runs on the pipe wave, 35 guesses of the last PU's mode ahead of the
    # PU wave (hevc_frame.h nxn_pipe); the emulated wavefronts are concurrent fibers, so the flag hand-offs between them are real
    stream, rcon = emu_encode(hostemu_pipe, kat_input(e["input"]), e["qpd6"])
    assert len(stream) == e["bytes"]
    assert hashlib.sha256(stream).hexdigest() == e["sha256"]
    assert hashlib.sha256(rcon.tobytes()).hexdigest() == e["rcon_sha256"]


@pytest.mark.parametrize("e", OVF, ids=kat_id)
def test_pipe_wave_lead_sink_exact_paths_match_golden(hostemu_pipe_ovf, e):
    # the same with a pipe wave: the lane that holds the NxN result counts header + four PU segments exactly
    stream, rcon = emu_encode(hostemu_pipe_ovf, kat_input(e["input"]), e["qpd6"])
    assert hashlib.sha256(stream).hexdigest() == e["sha256"]
    assert hashlib.sha256(rcon.tobytes()).hexdigest() == e["rcon_sha256"]


@pytest.mark.parametrize("e", PIPE_PICK, ids=kat_id)
def test_wide_workgroup_matches_golden(hostemu_wide, e):
    # 512-thread workgroups: every trial coder of an 8x8 CU (one-TU and four-TU candidate sets, PU pricing, the pipe wave's NxN
    # streams) runs as a range half and a byte half on two wavefronts joined by a record queue in LDS — concurrent fibers here
    stream, rcon = emu_encode(hostemu_wide, kat_input(e["input"]), e["qpd6"])
    assert len(stream) == e["bytes"]
    assert hashlib.sha256(stream).hexdigest() == e["sha256"]
    assert hashlib.sha256(rcon.tobytes()).hexdigest() == e["rcon_sha256"]


@pytest.mark.parametrize("e", OVF, ids=kat_id)
def test_wide_workgroup_lead_sink_exact_paths_match_golden(hostemu_wide_ovf, e):
    # the same in wide workgroups: the byte halves on partner wavefronts, PU pricing
    stream, rcon = emu_encode(hostemu_wide_ovf, kat_input(e["input"]), e["qpd6"])
    assert hashlib.sha256(stream).hexdigest() == e["sha256"]
    assert hashlib.sha256(rcon.tobytes()).hexdigest() == e["rcon_sha256"]


@pytest.mark.parametrize("e", OVF, ids=kat_id)
def test_wide_workgroup_ep_guard_safe_path_matches_golden(hostemu_wide_ep, e):
    # a PU candidate whose stream may hold two zero bytes in a row is priced again by the plain coder (hevc_frame.h pu_price); this build widens the guard
    stream, rcon = emu_encode(hostemu_wide_ep, kat_input(e["input"]), e["qpd6"])
    assert hashlib.sha256(stream).hexdigest() == e["sha256"]
    assert hashlib.sha256(rcon.tobytes()).hexdigest() == e["rcon_sha256"]


def _extreme_pictures():
    rng = np.random.default_rng(1)
    pics = {"bw_noise": rng.integers(0, 2, (32, 64), dtype=np.uint8) * 255, "noise": rng.integers(0, 256, (32, 64), dtype=np.uint8)}
    cb = np.zeros((32, 32), np.uint8); cb[::2, 1::2] = 255; cb[1::2, ::2] = 255; pics["checker"] = cb
    st = np.zeros((32, 64), np.uint8); st[:, 32:] = 255; st[16:, :32] = 255; pics["steps"] = st
    sp = np.zeros((64, 32), np.uint8); sp[:, ::2] = 255; pics["stripes"] = sp
    wh = np.full((32, 32), 255, np.uint8); wh[0, 0] = 0; pics["white"] = wh
    return pics


@pytest.mark.parametrize("name", sorted(_extreme_pictures()))
@pytest.mark.parametrize("q", [0, 4])
def test_matrix_transforms_on_extreme_residuals(hostemu, name, q):
    """The matrix-instruction transforms cut their wider operands into i8 limbs (hevc_core.h mx_mm): full-swing residuals drive
    the intermediate of the forward transform to its 17 bits and the dequantised levels into the 16-bit clip (:511-515, :613) —
    compared with the CPU checker, and the run must really have issued both matrix instructions."""
    from oracle import oracle
    hostemu.hostemu_mfma_calls.restype = C.c_long
    before = [hostemu.hostemu_mfma_calls(k) for k in (0, 1)]
    img = np.ascontiguousarray(_extreme_pictures()[name])
    stream, rcon = emu_encode(hostemu, img, q)
    want, wr, _ = oracle.cpu_encode(img, q)
    assert stream == want and (rcon == wr).all()
    assert all(hostemu.hostemu_mfma_calls(k) > before[k] for k in (0, 1))


@pytest.mark.parametrize("q", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("sz", [4, 8, 16, 32])
def test_rdoq_thresholds_match_reference_loop(hostemu, q, sz):
    """The device's RDOQ decides by thresholds on the rounding remainder (hevc_core.h rdoq_group) where the reference prices three
    levels per coefficient (src/HEVCe/HEVCe.c:540-594).  Exhaustive over every |coefficient| the 17-bit clamp distinguishes
    (0 .. 0x20001), both signs, for this (qpd6, size): the levels must be the ones the reference's quantize() returns — the real
    reference where oracle/_ref is built (dev container), its pinned restatement otherwise.  Fifteen test values and one
    saturating filler per 4x4 group keep the weak-group rule out of the way; a second pass without fillers exercises that rule."""
    from oracle import oracle
    n = sz * sz
    ip = C.POINTER(C.c_int)
    if oracle.have_ref():
        ref = oracle.ref_lib()
        def want(block):                                            # quantize(qpd6, sz, src[][32], dst[][32])
            src = np.zeros((32, 32), np.int32); dst = np.zeros((32, 32), np.int32)
            src[:sz, :sz] = block.reshape(sz, sz)
            ref.quantize(q, sz, src.ctypes.data_as(ip), dst.ctypes.data_as(ip))
            return dst[:sz, :sz].reshape(-1).copy()
    else:
        port = oracle.port_lib()
        def want(block):
            dst = np.zeros(n, np.int32)
            port.oracle_rdoq(q, sz, block.ctypes.data_as(ip), dst.ctypes.data_as(ip))
            return dst
    def got(block):
        dst = np.zeros(n, np.int32)
        assert hostemu.hostemu_rdoq_block(q, sz, block.ctypes.data_as(ip), dst.ctypes.data_as(ip)) == 0
        return dst
    # positions of a block in (group, in-group) order; slot 15 of every group is the filler
    gy, gx = np.meshgrid(np.arange(0, sz, 4), np.arange(0, sz, 4), indexing="ij")
    pos = np.stack([((gy.reshape(-1, 1) + np.arange(16) // 4) * sz + gx.reshape(-1, 1) + np.arange(16) % 4)], 0)[0]   # [groups][16]
    test_pos = pos[:, :15].reshape(-1)
    vals = np.arange(0, 0x20002, dtype=np.int64)
    vals = np.concatenate([vals, -vals]).astype(np.int32)
    per = test_pos.size
    for filler in (0x1ffff, None):
        for i in range(0, vals.size, per):
            chunk = vals[i:i + per]
            block = np.zeros(n, np.int32)
            block[test_pos[:chunk.size]] = chunk
            if filler is not None:
                block[pos[:, 15]] = filler
            elif i > 40 * per:
                break                                              # without fillers only small values meet the weak-group rule
            w, g = want(block), got(block)
            assert (w == g).all(), (q, sz, filler, i, block[np.nonzero(w != g)[0][:4]], w[np.nonzero(w != g)[0][:4]], g[np.nonzero(w != g)[0][:4]])


def test_rdoq_never_class_is_really_never(hostemu):
    """Classes 0 and 9 of the threshold table (level 0; levels >= 8 whose escape length does not grow) carry 'never'."""
    hostemu.hostemu_rdoq_threshold.restype = C.c_int
    for q in range(5):
        for s in range(4):
            assert hostemu.hostemu_rdoq_threshold(q, s, 0) < -(1 << 20) and hostemu.hostemu_rdoq_threshold(q, s, 9) < -(1 << 20)
            assert all(hostemu.hostemu_rdoq_threshold(q, s, c) < 0 for c in range(10))


def emu_encode_pool(lib, imgs, q, nmains, nhelp, npart=0):
    n = len(imgs)
    imgs = [np.ascontiguousarray(a) for a in imgs]
    outs = [np.zeros(2 * (a.shape[1] + 32) * (a.shape[0] + 32) + 65536, np.uint8) for a in imgs]
    rcs = [np.zeros(((a.shape[0] + 31) // 32 * 32) * ((a.shape[1] + 31) // 32 * 32), np.uint8) for a in imgs]
    P = u8p * n
    ys = (C.c_int * n)(*[a.shape[0] for a in imgs]); xs = (C.c_int * n)(*[a.shape[1] for a in imgs]); lens = (C.c_int * n)()
    lib.hostemu_HEVCImageEncoderPool3.restype = C.c_int
    assert lib.hostemu_HEVCImageEncoderPool3(n, P(*[o.ctypes.data_as(u8p) for o in outs]), P(*[a.ctypes.data_as(u8p) for a in imgs]),
                                             P(*[r.ctypes.data_as(u8p) for r in rcs]), ys, xs, q, lens, nmains, nhelp, npart) == 0
    lib.hostemu_prog.restype = C.c_uint
    for i in range(n):
        assert lib.hostemu_prog(i, 0) == (ys[i] // 32 | 0x80000000) and lib.hostemu_prog(i, 1) == lens[i], i
    return [(outs[i][:lens[i]].tobytes(), rcs[i]) for i in range(n)]


@pytest.mark.parametrize("nmains,nhelp", [(1, 2), (1, 1), (2, 4), (3, 3), (4, 1), (1, 5)])
@pytest.mark.parametrize("q", [0, 4])
def test_pool_modes_match_golden(hostemu, q, nmains, nhelp):
    """Frames encoded by main workgroups and a POOL of helper workgroups (hevc_frame.h: a main workgroup walks the 8x8 CUs of its
    frame, any helper evaluates the 16x16 / 32x32 candidate sets from the entry states it posts) give the reference's bytes,
    whatever the ratio of the two kinds; the main workgroups pull frames from one queue.  The emulated workgroups run
    concurrently (fibers), so the ticket queue and the request / result hand-offs are real."""
    es = [e for e in OVF if e["qpd6"] == q]
    res = emu_encode_pool(hostemu, [kat_input(e["input"]) for e in es], q, nmains, nhelp)
    for e, (stream, rcon) in zip(es, res):
        assert hashlib.sha256(stream).hexdigest() == e["sha256"], kat_id(e)
        assert hashlib.sha256(rcon.tobytes()).hexdigest() == e["rcon_sha256"], kat_id(e)


@pytest.mark.parametrize("nmains,nhelp", [(1, 2), (3, 2), (2, 4)])
def test_pool_as_two_launches_matches_golden(hostemu, nmains, nhelp):
    """A pool spread over two cooperating launches (hevc_hip.hip launch of role 1 / role 2 workgroups: the main workgroups in one, the
    helpers in the other): roles come from the launch, not from where a workgroup landed; same bytes."""
    es = [e for e in OVF if e["qpd6"] == 0]
    hostemu.hostemu_set_role_split(1)
    try:
        res = emu_encode_pool(hostemu, [kat_input(e["input"]) for e in es], 0, nmains, nhelp)
    finally:
        hostemu.hostemu_set_role_split(0)
    for e, (stream, rcon) in zip(es, res):
        assert hashlib.sha256(stream).hexdigest() == e["sha256"], kat_id(e)
        assert hashlib.sha256(rcon.tobytes()).hexdigest() == e["rcon_sha256"], kat_id(e)


@pytest.mark.parametrize("nmains,nhelp", [(1, 2), (2, 3), (1, 0), (3, 0)])
@pytest.mark.parametrize("q", [0, 4])
def test_pipe_wave_with_and_without_helpers(hostemu_pipe, q, nmains, nhelp):
    """Pipe-wave workgroups as main workgroups of a pool (their helpers' fourth wavefront only keeps the barrier count) and as
    plain frame-per-workgroup launches pulling several frames each: the reference's bytes."""
    es = [e for e in OVF if e["qpd6"] == q]
    res = emu_encode_pool(hostemu_pipe, [kat_input(e["input"]) for e in es], q, nmains, nhelp)
    for e, (stream, rcon) in zip(es, res):
        assert hashlib.sha256(stream).hexdigest() == e["sha256"], kat_id(e)
        assert hashlib.sha256(rcon.tobytes()).hexdigest() == e["rcon_sha256"], kat_id(e)


@pytest.mark.parametrize("nmains,nhelp", [(1, 2), (2, 1), (2, 0)])
@pytest.mark.parametrize("q", [0, 4])
def test_wide_workgroups_with_and_without_helpers(hostemu_wide, q, nmains, nhelp):
    """Wide workgroups as main and helper workgroups of a pool (the helpers' 16x16 / 32x32 trial coders run split over two wavefronts
    too) and as plain frame-per-workgroup launches pulling several frames each: the reference's bytes."""
    es = [e for e in OVF if e["qpd6"] == q]
    res = emu_encode_pool(hostemu_wide, [kat_input(e["input"]) for e in es], q, nmains, nhelp)
    for e, (stream, rcon) in zip(es, res):
        assert hashlib.sha256(stream).hexdigest() == e["sha256"], kat_id(e)
        assert hashlib.sha256(rcon.tobytes()).hexdigest() == e["rcon_sha256"], kat_id(e)


@pytest.mark.parametrize("nmains,nhelp,npart", [(1, 1, 1), (1, 2, 1), (2, 2, 2), (2, 3, 1)])
@pytest.mark.parametrize("q", [0, 4])
def test_wide_partner_workgroups_match_golden(hostemu_wide, q, nmains, nhelp, npart):
    """Wide pools with PARTNER workgroups (hevc_frame.h "8x8 CUs with a partner workgroup"): main workgroup i posts the entry state of every 8x8 CU to
    partner i, which evaluates the CU's two 2Nx2N candidate sets (TU 0 of the four-TU set made there, two lenders for the one-TU set) while the main
    workgroup walks the NxN chain alone; the answer — the last minimum of the 70 — meets the NxN cost under the reference's order and tie rule
    (:1439, :1475, :1545).  One partner for two main workgroups too (the second keeps its sets): the reference's bytes."""
    es = [e for e in OVF if e["qpd6"] == q]
    st = (C.c_long * 3)()
    hostemu_wide.hostemu_remote8_stats(st, 1)
    res = emu_encode_pool(hostemu_wide, [kat_input(e["input"]) for e in es], q, nmains, nhelp, npart)
    hostemu_wide.hostemu_remote8_stats(st, 1)
    ctus = sum(((e["input"].get("h", 32) + 31) // 32) * ((e["input"].get("w", 32) + 31) // 32) for e in es)
    assert st[0] >= 16 * ctus * npart // (2 * nmains) and 0 < st[1] < st[0] and st[2] == 0, (list(st), ctus)      # the partners did serve, both kinds of winner occurred, nothing was abandoned
    for e, (stream, rcon) in zip(es, res):
        assert hashlib.sha256(stream).hexdigest() == e["sha256"], kat_id(e)
        assert hashlib.sha256(rcon.tobytes()).hexdigest() == e["rcon_sha256"], kat_id(e)


def test_wide_partner_workgroups_natural_image(hostemu_wide):
    # 10 x 9 CTUs of the reference's own sample picture at both ends of the quantiser range (every decision branch)
    for q in (0, 4):
        e = next(e for e in kat_entries() if e["input"].get("file") == "p5_gray.pgm" and e["qpd6"] == q)
        (stream, rcon), = emu_encode_pool(hostemu_wide, [kat_input(e["input"])], q, 1, 1, 1)
        assert hashlib.sha256(stream).hexdigest() == e["sha256"] and hashlib.sha256(rcon.tobytes()).hexdigest() == e["rcon_sha256"], q


def test_wide_partner_workgroups_exact_lead_paths(hostemu_wide_ovf):
    es = [e for e in OVF if e["qpd6"] == 0]
    res = emu_encode_pool(hostemu_wide_ovf, [kat_input(e["input"]) for e in es], 0, 1, 1, 1)
    for e, (stream, rcon) in zip(es, res):
        assert hashlib.sha256(stream).hexdigest() == e["sha256"] and hashlib.sha256(rcon.tobytes()).hexdigest() == e["rcon_sha256"], kat_id(e)


@pytest.mark.parametrize("q", [0, 3])
def test_abandoned_partner_answers_are_recomputed(hostemu_abn, q):
    """... and a main workgroup that stops waiting for its partner (three polls here) walks the CU again itself — all three candidate sets, same bytes —
    and posts nothing until the late answer has landed."""
    es = [e for e in OVF if e["qpd6"] == q] or [e for e in OVF][:2]
    hostemu_abn.hostemu_set_threads(512)
    st = (C.c_long * 3)()
    hostemu_abn.hostemu_remote8_stats(st, 1)
    try:
        res = emu_encode_pool(hostemu_abn, [kat_input(e["input"]) for e in es], es[0]["qpd6"], 1, 1, 1)
    finally:
        hostemu_abn.hostemu_set_threads(192)
    hostemu_abn.hostemu_remote8_stats(st, 1)
    assert st[2] > 0, list(st)              # answers were abandoned
    for e, (stream, rcon) in zip(es, res):
        assert hashlib.sha256(stream).hexdigest() == e["sha256"], kat_id(e)
        assert hashlib.sha256(rcon.tobytes()).hexdigest() == e["rcon_sha256"], kat_id(e)


@pytest.mark.parametrize("nmains,nhelp", [(1, 1), (2, 2), (4, 1)])
@pytest.mark.parametrize("q", [0, 3])
def test_abandoned_answers_are_recomputed_by_the_main_workgroup(hostemu_abn, q, nmains, nhelp):
    """A main workgroup that does not get an answer in time evaluates the CU itself and does not reuse the mailbox before the late
    answer has landed (on the device: helpers held up by wave preemption).  Built to give up after three polls, so that most
    requests are abandoned: same bytes."""
    es = [e for e in OVF if e["qpd6"] == q] or [e for e in OVF][:2]
    res = emu_encode_pool(hostemu_abn, [kat_input(e["input"]) for e in es], es[0]["qpd6"], nmains, nhelp)
    for e, (stream, rcon) in zip(es, res):
        assert hashlib.sha256(stream).hexdigest() == e["sha256"], kat_id(e)
        assert hashlib.sha256(rcon.tobytes()).hexdigest() == e["rcon_sha256"], kat_id(e)


def test_idle_helpers_take_over_as_main_workgroups(hostemu, monkeypatch):
    """No workgroup starts as a main one (quota 0, as if every compute unit had filled up before this launch's workgroups arrived):
    idle helpers take the free main indices, encode the frames and are served by the remaining helpers; same bytes."""
    monkeypatch.setenv("HOSTEMU_QUOTA", "0")
    es = [e for e in OVF if e["qpd6"] == 0]
    res = emu_encode_pool(hostemu, [kat_input(e["input"]) for e in es], 0, 2, 3)
    for e, (stream, rcon) in zip(es, res):
        assert hashlib.sha256(stream).hexdigest() == e["sha256"], kat_id(e)
        assert hashlib.sha256(rcon.tobytes()).hexdigest() == e["rcon_sha256"], kat_id(e)


def test_busy_helpers_take_over_late_main_indices(hostemu, monkeypatch):
    """... and a main index that is still free when the launch is milliseconds old (its workgroup is held back by the dispatcher: launches that fill every slot)
    is taken by a running helper at once, busy or idle (hevc_core.h late_main_due; here: the helpers of odd queue shards take that path, the others the idle one)."""
    monkeypatch.setenv("HOSTEMU_QUOTA", "0")
    monkeypatch.setenv("HOSTEMU_LATE_MAIN", "1")
    es = [e for e in OVF if e["qpd6"] == 0]
    for nmains, nhelp in ((2, 3), (1, 3)):
        res = emu_encode_pool(hostemu, [kat_input(e["input"]) for e in es], 0, nmains, nhelp)
        for e, (stream, rcon) in zip(es, res):
            assert hashlib.sha256(stream).hexdigest() == e["sha256"], kat_id(e)
            assert hashlib.sha256(rcon.tobytes()).hexdigest() == e["rcon_sha256"], kat_id(e)


def test_pool_mode_natural_image(hostemu):
    # 10 x 9 CTUs of the reference's own sample picture: helpers read their borders from the reconstruction plane across CTU rows
    e = next(e for e in kat_entries() if e["input"].get("file") == "p5_gray.pgm" and e["qpd6"] == 4)
    (stream, rcon), = emu_encode_pool(hostemu, [kat_input(e["input"])], 4, 1, 2)
    assert hashlib.sha256(stream).hexdigest() == e["sha256"] and hashlib.sha256(rcon.tobytes()).hexdigest() == e["rcon_sha256"]


def test_sizes_beyond_8192_are_cropped_like_the_reference(hostemu):
    # :1580-1581 pads min(dim, 8192) while the source keeps its own stride (:1621): one CTU row / column of 257 CTUs' worth
    from oracle import oracle, synth
    for h, w in ((3, 8200),):          # (the 8200 x 3 column runs on the GPU, tests/test_gpu_parity.py: in the emulation each orientation takes most of a minute)
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
    # four 192-thread workgroups per CU, or three 256-thread ones with the pipe wave's dynamic slice
    assert 4 * hostemu.hostemu_shm_bytes() <= 160 * 1024
    assert 3 * (hostemu.hostemu_shm_bytes() + hostemu.hostemu_pipe_lds_bytes()) <= 160 * 1024
    # one 512-thread wide workgroup per CU with the partner wavefronts' record queues
    assert hostemu.hostemu_shm_bytes() + hostemu.hostemu_wide_lds_bytes() <= 160 * 1024


def _byte_logic(leads, nbytes, buf, zeros, cnt, out):
    """The byte-level logic of the coder (:863-878, :820-831) over a list of 9-bit leads; bytes go to out[cnt ..]."""
    def emit(v):
        nonlocal zeros, cnt
        v &= 0xFF
        if zeros >= 2 and v <= 3:
            out[cnt] = 3; cnt += 1; zeros = 0
        out[cnt] = v; cnt += 1
        zeros = 0 if v else zeros + 1
    for lead in leads:
        if lead == 0xFF:
            nbytes += 1
        elif nbytes > 0:
            carry = lead >> 8
            v = buf + carry
            buf = lead & 0xFF
            emit(v)
            v = (0xFF + carry) & 0xFF
            while nbytes > 1:
                emit(v); nbytes -= 1
        else:
            nbytes, buf = 1, lead
    return nbytes, buf, zeros, cnt


def test_winner_leads_to_bytes_matches_the_byte_level_logic(hostemu):
    """hevc_core.h resolve_leads (the winner's leads -> bytes, by one wavefront: carry look-ahead over ballots, emulation prevention checked on the
    result, lane 0's walk when it would strike) against the byte-level logic itself: random lists, lists full of 0xFF runs and carries, lists that
    force emulation prevention, lists longer than a wavefront, every kind of entry state."""
    import random
    rng = random.Random(11)
    lib = hostemu
    lib.hostemu_resolve_leads.restype = C.c_int
    lib.hostemu_resolve_leads.argtypes = [C.POINTER(C.c_ushort), C.c_int, C.POINTER(C.c_int), u8p]
    pools = [list(range(512)), [0xFF] * 6 + [0x1FF, 0x100, 0x00, 0x01, 0xFE, 0x1FE, 0x80, 0x17F], [0x00, 0x100, 0xFF, 0x01, 0x02, 0x03, 0x04, 0x1FF, 0x55]]
    ep_runs = 0
    for it in range(1500):
        pool = pools[it % 3]
        n = rng.choice([0, 1, 2, 5, 30, 63, 64, 65, 130, 200]) if it % 5 else rng.randint(0, 300)
        leads = [rng.choice(pool) for _ in range(n)]
        fresh = it % 7 == 0
        st = (0, 0xFF, 0, 0) if fresh else (rng.choice([1, 1, 1, 2, 3, 5]), rng.choice([0, 0xFF, 0x7F, 3, 0xFE]), rng.choice([0, 0, 1, 2, 3]), rng.randint(0, 40))
        # a carry can only be absorbed by a byte below 0xFF somewhere in the buffer: the coder guarantees it, random lists must too
        chk, ok = ([] if fresh else [st[1]] + [0xFF] * (st[0] - 1)), True
        for i, lead in enumerate(leads):
            v = lead & 0xFF
            if lead >> 8:
                if fresh and i == 0:
                    pass                                    # (the first lead's carry is dropped)
                else:
                    j = len(chk) - 1
                    while j >= 0 and chk[j] == 0xFF:
                        j -= 1
                    if j < 0 and not (fresh and i == 0):
                        ok = False; break
                    chk[j] += 1
                    for t in range(j + 1, len(chk)):
                        chk[t] = 0
            chk.append(v)
        if not ok:
            continue
        want = np.zeros(800, np.uint8)
        wst = _byte_logic(leads, st[0], st[1], st[2], st[3], want)
        ep_runs += wst[0] + wst[3] != st[0] + st[3] + len(leads)
        got = np.zeros(800, np.uint8)
        arr = (C.c_ushort * max(n, 1))(*leads)
        gst = (C.c_int * 4)(*st)
        lib.hostemu_resolve_leads(arr, n, gst, got.ctypes.data_as(u8p))
        g = tuple(gst)
        assert (g[0], g[1] & 0xFF, min(g[2], 2), g[3]) == (wst[0], wst[1] & 0xFF, min(wst[2], 2), wst[3]), (it, st, leads[:20], g, wst)
        assert bytes(got[st[3]:wst[3]]) == bytes(want[st[3]:wst[3]]), (it, st, leads[:20])
    assert ep_runs > 20                                     # emulation prevention did strike in some of them (lane 0's walk)


def test_lead_sink_guard_no_hit_means_bytes_equal_leads(hostemu):
    """hevc_core.h lsink_begin / lsink_flush8: a trial's byte count is taken as its lead count unless the flushes see a lead that may come out
    at most 3 behind two bytes that come out zero (local carry look-ahead, the carry into a flush's last lead taken as set).  A model of that
    guard against the byte-level logic: whenever it stays quiet, emitted + buffered bytes have grown by exactly the number of leads."""
    import random
    rng = random.Random(7)
    lib = hostemu
    lib.hostemu_lsink_guard.restype = C.c_int
    lib.hostemu_lsink_guard.argtypes = [C.POINTER(C.c_ushort), C.c_int, C.POINTER(C.c_int)]
    O, Fb, Cb, D = 1, 2, 4, 8

    def lh(lead):
        v = lead & 0xFF
        return (O if v == 0 else 0) | (Fb if v == 0xFF else 0) | ((lead >> 8 & 1) << 2)

    def brev32(x):
        return int(format(x & 0xFFFFFFFF, "032b")[::-1], 2)

    def guard(leads, nbytes, buf, zeros):
        d1, d2 = (D if zeros >= 1 else 0), (D if zeros >= 2 else 0)
        r, vb = nbytes - 1, buf & 0xFF
        b = lh(vb)
        hb = int(d2 != 0 and (vb <= 3 or vb == 0xFF))
        if nbytes < 1: hist, hit = d1 | d2 << 4, 0
        elif r == 0: hist, hit = b | d1 << 4, hb
        elif r == 1: hist, hit = Fb | b << 4, hb | int(d1 != 0 and (b & (O | Fb)) != 0)
        else: hist, hit = Fb | Fb << 4, 1
        fl = 0
        while fl < len(leads):
            grp = leads[fl:fl + 8]; valid = len(grp); fl += 8
            Om = (2 if hist & O else 0) | (1 if hist >> 4 & O else 0); Fm = (2 if hist & Fb else 0) | (1 if hist >> 4 & Fb else 0)
            Cm = 2 if hist & Cb else 0; Dm = (2 if hist & D else 0) | (1 if hist >> 4 & D else 0); S3 = 0
            l = [lh(x) for x in grp]
            for j, x in enumerate(grp):
                Om |= (1 if l[j] & O else 0) << (j + 2); Fm |= (1 if l[j] & Fb else 0) << (j + 2); Cm |= (1 if l[j] & Cb else 0) << (j + 2)
                S3 |= (1 if (x & 0xFC) == 0 else 0) << (j + 2)
            top = valid + 1
            keep = (2 << top) - 1
            Cr, Fr = brev32(Cm & keep) >> (31 - top), brev32(Fm & keep) >> (31 - top)
            A = Cr | Fr; S = (A + Cr + 1) & 0xFFFFFFFF
            CI = brev32(((S ^ A ^ Cr) << (31 - top)) & 0xFFFFFFFF) & keep
            Z = Dm | (Om & ~CI) | (Fm & CI); T = S3 | (Fm & CI)
            hit |= int((T & (Z << 1) & (Z << 2) & keep & ~3) != 0)
            l1 = l[valid - 1] if valid >= 1 else hist & 15
            l2 = l[valid - 2] if valid >= 2 else (hist & 15)
            hist = l1 | l2 << 4
        return hit

    pools = [list(range(512)), [0x00, 0xFF, 0x100, 0x1FF, 0x01, 0x02, 0x03, 0x04, 0xFE, 0x101, 0x55, 0x1AA], [0x00, 0xFF, 0x100, 0x1FF, 0x03, 0x103]]
    quiet = fired = inserted = 0
    for it in range(40000):
        pool = pools[it % 3]
        leads = [rng.choice(pool) for _ in range(rng.randint(1, 40))]
        fresh = it % 5 == 0
        st = (0, 0xFF, 0, 0) if fresh else (rng.choice([1, 1, 1, 2, 3]), rng.choice([0, 0xFF, 0x7F, 3, 2, 0xFE]), rng.choice([0, 0, 1, 2, 3]), 0)
        # (lists in which a carry meets nothing but 0xFF are not coder output: skip them, as the resolve test does)
        chk, ok = ([] if fresh else [st[1]] + [0xFF] * (st[0] - 1)), True
        for i, lead in enumerate(leads):
            if lead >> 8 and not (fresh and i == 0):
                j = len(chk) - 1
                while j >= 0 and chk[j] == 0xFF:
                    j -= 1
                if j < 0:
                    ok = False; break
                chk[j] += 1
                for t in range(j + 1, len(chk)):
                    chk[t] = 0
            chk.append(lead & 0xFF)
        if not ok:
            continue
        out = np.zeros(200, np.uint8)
        nb, _, _, cnt = _byte_logic(leads, st[0], st[1], st[2], st[3], out)
        grown = nb + cnt - st[0] - st[3]
        arr = (C.c_ushort * len(leads))(*leads)
        dev = lib.hostemu_lsink_guard(arr, len(leads), (C.c_int * 3)(st[0], st[1], st[2]))
        assert dev == guard(leads, st[0], st[1], st[2]), (st, leads)      # the device source's sink and the model agree
        if dev:
            fired += 1; inserted += grown != len(leads)
        else:
            quiet += 1
            assert grown == len(leads), (st, leads)
    assert quiet > 2000 and fired > 2000 and inserted > 200
