"""GPU parity tests (run on MI355X with -m gpu): the HIP path, called through the C ABI, against
  * the committed golden vectors generated from the real reference (tests/golden/hevc_kat.json),
  * the CPU checker on seeded inputs (oracle/_ref when it travelled, else the pinned restatement),
  * size-independent properties at the BASELINE sizes (golden digests of whole 1080p frames, batch == single,
    reconstruction consistent with the stream's own decisions).
Bar: bit-exact streams AND bit-exact reconstructions."""
import hashlib
import os

import numpy as np
import pytest

from conftest import kat_entries, kat_id, kat_input

pytestmark = pytest.mark.gpu

KAT = kat_entries()
SMALL = [e for e in KAT if e["input"].get("w", 0) < 1920]
LARGE = [e for e in KAT if e["input"].get("w", 0) == 1920]
UHD = [e for e in KAT if e["input"].get("w", 0) == 3840]


@pytest.fixture(scope="module")
def amd(built):
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import imcvt_amd
    imcvt_amd.load_library()      # raises if the HIP extension is missing: no fallback
    return imcvt_amd


def test_small_golden_vectors_one_batch(amd):
    """All small golden vectors (P4/P5/P6 sample images, synthetic x qpd6 0..4) in ONE device batch."""
    imgs = [kat_input(e["input"]) for e in SMALL]
    res = amd.HEVCImageEncoderBatch(imgs, [e["qpd6"] for e in SMALL])
    bad = []
    for e, (stream, rcon, dims) in zip(SMALL, res):
        if (len(stream) != e["bytes"] or hashlib.sha256(stream).hexdigest() != e["sha256"]
                or hashlib.sha256(rcon.tobytes()).hexdigest() != e["rcon_sha256"]):
            bad.append(kat_id(e))
    assert not bad, bad


@pytest.mark.parametrize("name,spec,q", [("p4_q0.h265", dict(kind="file", file="p4_gray.pgm"), 0),
                                         ("p5_q4.h265", dict(kind="file", file="p5_gray.pgm"), 4),
                                         ("syn33x31s2_q0.h265", dict(kind="syn", w=33, h=31, arg=2), 0)])
def test_single_call_matches_reference_stream_bytes(amd, name, spec, q):
    from conftest import ROOT
    want = open(os.path.join(ROOT, "tests", "golden", name), "rb").read()
    got, _, _ = amd.HEVCImageEncoder(kat_input(spec), q)
    assert got == want


def test_seeded_random_vs_cpu_checker(amd):
    from oracle import oracle
    rng = np.random.default_rng(1234)
    imgs, qs = [], []
    for i in range(24):
        h, w = int(rng.integers(1, 130)), int(rng.integers(1, 130))
        kind = i % 4
        a = (rng.integers(0, 256, (h, w)) if kind == 0 else np.clip(rng.normal(120, 25, (h, w)), 0, 255) if kind == 1
             else (np.add.outer(np.arange(h) * 5, np.arange(w) * 3) % 256) if kind == 2 else np.full((h, w), int(rng.integers(0, 256))))
        imgs.append(a.astype(np.uint8)); qs.append(i % 5)
    res = amd.HEVCImageEncoderBatch(imgs, qs)
    for img, q, (s, r, _) in zip(imgs, qs, res):
        ws, wr, _ = oracle.cpu_encode(img, q)
        assert s == ws and (r == wr).all(), (img.shape, q)


def test_edge_cases(amd):
    from oracle import oracle
    # 1x1, single row/column, exact CTU multiples, one past a multiple, extreme values
    imgs = [np.array([[7]], np.uint8), np.arange(70, dtype=np.uint8).reshape(1, 70), np.arange(45, dtype=np.uint8).reshape(45, 1),
            np.zeros((32, 64), np.uint8), np.full((64, 32), 255, np.uint8), (np.indices((33, 65)).sum(0) % 2 * 255).astype(np.uint8),
            (np.arange(3 * 8200) % 251).astype(np.uint8).reshape(3, 8200), (np.arange(3 * 8200) % 241).astype(np.uint8).reshape(8200, 3)]   # beyond 8192: cropped (:1580-1581)
    for q in (0, 4):
        for img, (s, r, dims) in zip(imgs, amd.HEVCImageEncoderBatch(imgs, q)):
            ws, wr, wd = oracle.cpu_encode(img, q)
            assert dims == wd and s == ws and (r == wr).all(), (img.shape, q)
    assert amd.HEVCImageEncoderBatch([], 0) == []


def test_bad_arguments_are_rejected(amd):
    with pytest.raises(RuntimeError, match="bad argument"):
        amd.HEVCImageEncoder(np.zeros((8, 8), np.uint8), 5)


def test_full_hd_frames_golden_digest(amd):
    """BASELINE config 2/4 size: whole 1920x1080 frames against the reference's digests (two seeds + qpd6=4)."""
    from oracle import synth
    pick = [e for e in LARGE if (e["input"]["arg"], e["qpd6"]) in ((0, 0), (3, 0), (0, 4))]
    imgs = [synth.syn(1920, 1080, e["input"]["arg"]) for e in pick]
    res = amd.HEVCImageEncoderBatch(imgs, [e["qpd6"] for e in pick])
    for e, (s, r, dims) in zip(pick, res):
        assert dims == (1088, 1920)
        assert len(s) == e["bytes"] and hashlib.sha256(s).hexdigest() == e["sha256"], kat_id(e)
        assert hashlib.sha256(r.tobytes()).hexdigest() == e["rcon_sha256"], kat_id(e)


def test_4k_frame_golden_digest(amd):
    """BASELINE config 3: one 3840x2160 frame (8160 CTUs, one workgroup walks them serially) against the reference's digest."""
    from oracle import synth
    e = UHD[0]
    s, r, dims = amd.HEVCImageEncoder(synth.syn(3840, 2160, e["input"]["arg"]), e["qpd6"])
    assert dims == (2176, 3840)
    assert len(s) == e["bytes"] and hashlib.sha256(s).hexdigest() == e["sha256"]
    assert hashlib.sha256(r.tobytes()).hexdigest() == e["rcon_sha256"]


def test_batch_equals_single_and_is_order_independent(amd):
    """Size-independent property: frames are independent — any batching/ordering gives the same bytes."""
    from oracle import synth
    imgs = [synth.syn(160, 96, s) for s in range(6)]
    a = amd.HEVCImageEncoderBatch(imgs, 2)
    b = amd.HEVCImageEncoderBatch(imgs[::-1], 2)[::-1]
    for i, img in enumerate(imgs):
        s, r, _ = amd.HEVCImageEncoder(img, 2)
        assert s == a[i][0] == b[i][0] and (r == a[i][1]).all() and (r == b[i][1]).all()


def test_device_resident_batch_matches_host_abi(amd):
    import torch
    from oracle import synth
    enc = amd.DeviceEncoder()
    frames = [synth.syn(96 + 16 * s, 64, s) for s in range(5)]
    batch = enc.make_batch([torch.from_numpy(f).cuda() for f in frames], [0, 1, 2, 3, 4])
    enc.encode(batch)
    got = enc.results(batch)
    assert enc.last_kernel_ms() > 0
    for f, q, (s, r) in zip(frames, range(5), got):
        ws, wr, _ = amd.HEVCImageEncoder(f, q)
        assert s == ws and (r == wr).all()
    enc.close()


@pytest.mark.parametrize("flag", ["-DIMCVT_FORCE_OVF", "-DROWCAP=6", "-DP1_MFMA=0", "-DLATE_MAIN_TICKS=0"], ids=["lead-sink-exact-paths", "token-row-overflow-path", "vector-transforms", "late-main-take-over"])
def test_rare_paths_on_the_device(amd, flag, tmp_path):
    """The paths that practically never run — a trial whose leads show the emulation-prevention pattern gets its byte count from the real
    logic over its list, a winner whose bytes would need emulation prevention is turned into bytes by lane 0's walk; a pass
    whose group tokens do not fit the lanes' LDS rows counts and writes them the plain way — forced by a build flag, on the
    real hardware (the host emulation checks their logic; lock-step execution is what only the GPU has).  Third variant: the
    N = 16 / 32 transforms as vector code instead of matrix instructions (the A/B build of hevc_core.h p1_run_t) — same bytes.  Fourth: helpers take
    free main-workgroup indices as soon as they see one (hevc_core.h late_main_due with no age limit: the take-over that otherwise waits until a launch is 2 ms old races the
    main workgroups' own start here) — roles move between workgroups, bytes do not."""
    import subprocess, sys
    from conftest import ROOT
    so = str(tmp_path / "libimcvt_hevc_variant.so")
    src = os.path.join(ROOT, "imcvt_amd", "csrc", "hevc_hip.hip")
    subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", "-mllvm", "-disable-machine-licm",
                    flag, src, "-o", so], check=True)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_parity.py")], env=dict(os.environ, IMCVT_HEVC_LIB=so),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "FAILURES: 0" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


INL = "__device__ __forceinline__"


@pytest.mark.parametrize("flags", [["-DHDN_BORDER=" + INL], ["-DHDN_EVAL=" + INL], ["-DHDN_BORDER=" + INL, "-DHDN_EVAL=" + INL]],
                         ids=["borders-inlined", "candidate-sets-inlined", "both-inlined"])
def test_results_do_not_depend_on_inlining(amd, flags, tmp_path):
    """The border and candidate-set functions are out of line for code size only: built inline (either or both), the kernel
    gives the same bytes.  (In round 2 inlining the border functions changed results: their wave-uniform arguments arrived in
    vector registers and everything derived from them — branch conditions around the wave-level syncs included — was vector
    code; the arguments are now moved to scalar registers on entry, uni_i / uni_p in hevc_core.h, and the branches are scalar.)"""
    import subprocess, sys
    from conftest import ROOT
    so = str(tmp_path / "libimcvt_hevc_variant.so")
    src = os.path.join(ROOT, "imcvt_amd", "csrc", "hevc_hip.hip")
    subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", "-mllvm", "-disable-machine-licm",
                    *flags, src, "-o", so], check=True)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_parity.py")], env=dict(os.environ, IMCVT_HEVC_LIB=so),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "FAILURES: 0" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_more_frames_than_workgroup_slots(amd):
    """A batch larger than the 1024 resident workgroups: the persistent workgroups keep pulling frames from the queue."""
    from oracle import oracle, synth
    imgs = [synth.syn(32 + (i % 3), 32, i % 7) for i in range(1100)]
    res = amd.HEVCImageEncoderBatch(imgs, 2)
    want = {}
    for i in (0, 1, 2, 3, 4, 5, 6, 1023, 1024, 1099):
        key = (imgs[i].shape, i % 7)
        if key not in want:
            want[key] = oracle.cpu_encode(imgs[i], 2)
        s, r, _ = res[i]
        assert s == want[key][0] and (r == want[key][1]).all(), i
    # (width, seed) repeats with period lcm(3, 7) = 21: 21 distinct frames, each repeated — also across the 1024-slot boundary
    assert len({bytes(res[i][0]) for i in range(21)}) == 21
    assert all(res[i][0] == res[i % 21][0] and (res[i][1] == res[i % 21][1]).all() for i in range(1100))


def test_bench_batch_device_resident(amd):
    """BASELINE configs[3] shape on one GPU: a device-resident batch of 512 independent frames in one launch (small frames so
    the test stays short): every stream and reconstruction equals the single-frame call, frames are pulled in any order."""
    import torch
    from oracle import oracle, synth
    uniq = [synth.syn(96, 64, s) for s in range(16)]
    enc = amd.DeviceEncoder()
    batch = enc.make_batch([torch.from_numpy(uniq[i % 16]).cuda() for i in range(512)], 0)
    enc.encode(batch)
    got = enc.results(batch)
    enc.encode(batch)                                   # a second launch on the same context reuses every buffer
    again = enc.results(batch)
    enc.close()
    want = [oracle.cpu_encode(u, 0) for u in uniq]
    for i in range(512):
        assert got[i][0] == want[i % 16][0] and (got[i][1] == want[i % 16][1]).all(), i
        assert again[i][0] == got[i][0]


@pytest.mark.parametrize("team", [1, 2, 3])
def test_team_sizes_give_identical_streams(amd, team):
    """One frame per workgroup, or per team of 2 / 3 cooperating workgroups (requests and results handed over through
    global memory between compute units): identical bytes.  Frames of many sizes, more frames than teams of 3 fit at once
    would need only with 350+ frames, so the queue is exercised with a forced small team count by the batch itself."""
    import torch
    from oracle import oracle
    rng = np.random.default_rng(99)
    imgs = [kat_input(dict(kind="file", file="p5_gray.pgm"))]
    for i in range(40):
        h, w = int(rng.integers(1, 200)), int(rng.integers(1, 200))
        imgs.append((rng.integers(0, 256, (h, w)) if i % 3 == 0 else np.clip(rng.normal(128, 30, (h, w)), 0, 255) if i % 3 == 1
                     else np.add.outer(np.arange(h) * 3, np.arange(w) * 7) % 256).astype(np.uint8))
    qs = [i % 5 for i in range(len(imgs))]
    enc = amd.DeviceEncoder()
    enc.set_team(team)
    batch = enc.make_batch([torch.from_numpy(a).cuda() for a in imgs], qs)
    for rep in range(2):                                # twice: sequence numbers and mailboxes restart with every launch
        enc.encode(batch)
        got = enc.results(batch)
        assert enc.last_team()[0] == team
        for a, q, (s, r) in zip(imgs, qs, got):
            ws, wr, _ = oracle.cpu_encode(a, q)
            assert s == ws and (r == wr).all(), (a.shape, q, team, rep)
    enc.close()


def test_auto_team_choice_and_full_hd_team(amd):
    """A single 1080p frame is given to a team automatically; its stream still has the reference's digest."""
    import torch
    from oracle import synth
    e = next(e for e in LARGE if (e["input"]["arg"], e["qpd6"]) == (1, 0))
    enc = amd.DeviceEncoder()
    batch = enc.make_batch([torch.from_numpy(synth.syn(1920, 1080, 1)).cuda()], 0)
    enc.encode(batch)
    (s, r), = enc.results(batch)
    assert enc.last_team()[0] == 3
    assert len(s) == e["bytes"] and hashlib.sha256(s).hexdigest() == e["sha256"] and hashlib.sha256(r.tobytes()).hexdigest() == e["rcon_sha256"]
    enc.close()


@pytest.mark.parametrize("team", [1, 3])
@pytest.mark.parametrize("pipe", [0, 1, 2])
def test_pipe_wave_on_and_off(amd, pipe, team):
    """256-thread workgroups whose fourth wavefront prices the NxN candidate of the 8x8 CUs ahead of the PU wave (hevc_frame.h
    nxn_pipe) against 192-thread ones, alone and as main workgroups of a pool: all small golden vectors, twice each.
    pipe 2: 512-thread wide workgroups — pipe wave + four partner wavefronts, every trial coder split over two wavefronts, the PU
    steps over three (hevc_core.h stream_seg_R / stream_seg_L, hevc_frame.h pu_step_wide) — as main workgroups, helpers and alone."""
    import torch
    enc = amd.DeviceEncoder()
    enc.set_pipe(min(pipe, 1)); enc.set_wide(-1 if pipe == 2 else 0); enc.set_team(team)
    batch = enc.make_batch([torch.from_numpy(np.ascontiguousarray(kat_input(e["input"]))).cuda() for e in SMALL], [e["qpd6"] for e in SMALL])
    for rep in range(2):
        enc.encode(batch)
        got = enc.results(batch)
        assert enc.last_pipe() == bool(pipe) and enc.last_wide() == (pipe == 2) and enc.last_team()[0] == team
        bad = [kat_id(e) for e, (s, r) in zip(SMALL, got)
               if len(s) != e["bytes"] or hashlib.sha256(s).hexdigest() != e["sha256"] or hashlib.sha256(r.tobytes()).hexdigest() != e["rcon_sha256"]]
        assert not bad, (pipe, team, rep, bad)
    enc.close()


def test_pipe_wave_full_hd_frame_and_launches_too_large_for_it(amd):
    """One 1080p frame with the pipe wave (the default for a launch that leaves room) has the reference's digest; a launch that
    needs four workgroups per compute unit runs without it, same results as with it forced off."""
    import torch
    from oracle import synth
    e = next(e for e in LARGE if (e["input"]["arg"], e["qpd6"]) == (2, 0))
    enc = amd.DeviceEncoder()
    batch = enc.make_batch([torch.from_numpy(synth.syn(1920, 1080, 2)).cuda()], 0)
    enc.encode(batch)
    (s, r), = enc.results(batch)
    assert enc.last_pipe() and enc.last_team()[0] == 3
    assert len(s) == e["bytes"] and hashlib.sha256(s).hexdigest() == e["sha256"] and hashlib.sha256(r.tobytes()).hexdigest() == e["rcon_sha256"]
    imgs = [torch.from_numpy(synth.syn(96, 64, i)).cuda() for i in range(800)]
    big = enc.make_batch(imgs, 1)
    enc.encode(big); a = enc.results(big)
    assert not enc.last_pipe()
    enc.set_pipe(1); enc.set_team(1)
    sub = enc.make_batch(imgs[:300], 1)
    enc.encode(sub); b = enc.results(sub)
    assert enc.last_pipe()
    assert all(x[0] == y[0] and (x[1] == y[1]).all() for x, y in zip(a, b))
    enc.close()


def test_wide_workgroups_full_hd_frame_and_launches_too_large_for_them(amd):
    """One 1080p frame runs wide workgroups by default (one main + two helper workgroups of 512 threads, a compute unit each) and has
    the reference's digest; so has the same frame with them switched off; 64 frames (64 + 128 workgroups) run wide, 128 frames too (the
    main workgroups take half of the compute units, the helpers are cut to the other half), 200 frames do not — same bytes either way."""
    import torch
    from oracle import synth
    e = next(e for e in LARGE if (e["input"]["arg"], e["qpd6"]) == (3, 0))
    enc = amd.DeviceEncoder()
    batch = enc.make_batch([torch.from_numpy(synth.syn(1920, 1080, 3)).cuda()], 0)
    for wide in (-1, 0):
        enc.set_wide(wide)
        enc.encode(batch)
        (s, r), = enc.results(batch)
        assert enc.last_pipe() and enc.last_wide() == (wide != 0) and enc.last_team()[0] == 3
        assert len(s) == e["bytes"] and hashlib.sha256(s).hexdigest() == e["sha256"] and hashlib.sha256(r.tobytes()).hexdigest() == e["rcon_sha256"]
    imgs = [torch.from_numpy(synth.syn(160, 96, i)).cuda() for i in range(200)]
    enc.set_wide(-1)
    b64 = enc.make_batch(imgs[:64], 0)
    enc.encode(b64); a = enc.results(b64)
    assert enc.last_wide() and enc.last_shape() == (64, 112) and enc.last_partners() == 64      # (round 6: a partner workgroup per main workgroup, the helpers cut to what is left of 15/16 of the compute units)
    enc.set_partners(0)
    enc.encode(b64); a_p = enc.results(b64)
    assert enc.last_wide() and enc.last_shape() == (64, 128) and enc.last_partners() == 0
    assert all(x[0] == y[0] and (x[1] == y[1]).all() for x, y in zip(a, a_p))
    enc.set_partners(-1)
    b128 = enc.make_batch(imgs[:128], 0)
    enc.encode(b128); b = enc.results(b128)
    cus = enc.residency()["cus"]
    assert enc.last_wide() and enc.last_split() and enc.last_shape() == (128, 3 * (cus - cus // 2))      # (round 6: two cooperating launches, wide main workgroups on one half of the compute units, 192-thread helpers on the other)
    enc.set_split(0)
    enc.encode(b128); b_one = enc.results(b128)
    assert enc.last_wide() and not enc.last_split() and enc.last_shape() == (128, cus - 128)
    assert all(x[0] == y[0] and (x[1] == y[1]).all() for x, y in zip(b, b_one))
    enc.set_split(-1)
    b200 = enc.make_batch(imgs, 0)
    enc.encode(b200); c200 = enc.results(b200)
    assert not enc.last_wide()
    assert all(x[0] == y[0] and (x[1] == y[1]).all() for x, y in zip(a, b)) and all(x[0] == y[0] and (x[1] == y[1]).all() for x, y in zip(b, c200))
    enc.close()


def test_host_batch_uses_the_visible_devices(amd):
    """HEVCImageEncoderBatch fans frames out over min(n, devices) devices; one device on this box, results unchanged."""
    import torch
    from oracle import synth
    imgs = [synth.syn(64, 64, s) for s in range(5)]
    a = amd.HEVCImageEncoderBatch(imgs, 1)
    lib = amd.load_library()
    assert lib.imcvt_hevc_batch_devices() == min(len(imgs), torch.cuda.device_count()) >= 1
    lib.imcvt_hevc_shutdown()                           # contexts are re-created on the next call
    b = amd.HEVCImageEncoderBatch(imgs, 1)
    assert [x[0] for x in a] == [x[0] for x in b]


def test_team_handoffs_under_uneven_load(amd):
    """Mailbox hand-offs between workgroups under uneven load (frames of very different sizes and content, more frames than
    teams so that every team serves several, repeated launches): teams of 3 and of 2 give what frame-per-workgroup launches give."""
    import torch
    rng = np.random.default_rng(77)
    imgs = []
    for i in range(140):
        h, w = int(rng.integers(1, 300)), int(rng.integers(1, 400))
        k = i % 4
        a = (rng.integers(0, 256, (h, w)) if k == 0 else np.clip(rng.normal(128, 40, (h, w)), 0, 255) if k == 1
             else (np.add.outer(np.arange(h) * 2, np.arange(w) * 3) % 256) if k == 2 else np.full((h, w), int(rng.integers(0, 256))))
        imgs.append(torch.from_numpy(a.astype(np.uint8)).cuda())
    qs = [i % 5 for i in range(len(imgs))]
    enc = amd.DeviceEncoder(max_workgroups=96)          # a small device share: 24 teams of 3 (32 of 2) serve the 140 frames
    batch = enc.make_batch(imgs, qs)
    enc.set_team(1); enc.encode(batch); ref = enc.results(batch)
    for team in (3, 2):
        enc.set_team(team)
        for rep in range(2):
            enc.encode(batch); got = enc.results(batch)
            assert enc.last_team()[0] == team
            for i, ((s, r), (s2, r2)) in enumerate(zip(got, ref)):
                assert s == s2 and (r == r2).all(), (team, rep, i)
    enc.close()


def test_pool_shapes_give_identical_streams(amd):
    """Any ratio of main to helper workgroups gives the bytes of frame-per-workgroup launches: many mains on one helper (the
    mains queue up for it), one main with many helpers (most of them idle), more mains than frames."""
    import torch
    rng = np.random.default_rng(5)
    imgs = [torch.from_numpy(rng.integers(0, 256, (int(rng.integers(20, 150)), int(rng.integers(20, 200)))).astype(np.uint8) if i % 2
                             else (np.add.outer(np.arange(90 + i) * 5, np.arange(70 + 3 * i) * 3) % 256).astype(np.uint8)).cuda() for i in range(24)]
    qs = [i % 5 for i in range(len(imgs))]
    enc = amd.DeviceEncoder()
    batch = enc.make_batch(imgs, qs)
    enc.set_team(1); enc.encode(batch); ref = enc.results(batch)
    for shape in ((8, 1), (1, 9), (3, 5), (40, 40), (24, 1)):
        enc.set_shape(*shape)
        enc.encode(batch); got = enc.results(batch)
        assert enc.last_shape() == (min(shape[0], len(imgs)), shape[1])
        for i, ((s, r), (s2, r2)) in enumerate(zip(got, ref)):
            assert s == s2 and (r == r2).all(), (shape, i)
    enc.close()


@pytest.mark.slow
def test_bench_frames_sample_against_reference_digests(amd):
    """BASELINE configs[3] at FULL size: 16 of the 512 bench frames (seeds spread over the range) in one device batch against
    the digests the REAL reference produced for them (tests/golden/bench512_kat.json, make_bench_golden.py)."""
    import json
    import torch
    from conftest import ROOT
    from oracle import synth
    kat = json.load(open(os.path.join(ROOT, "tests", "golden", "bench512_kat.json")))
    assert kat["input"] == {"kind": "syn", "w": 1920, "h": 1080} and kat["qpd6"] == 0
    seeds = [8 + 33 * i for i in range(15)] + [511]
    enc = amd.DeviceEncoder()
    batch = enc.make_batch([torch.from_numpy(synth.syn(1920, 1080, s)).cuda() for s in seeds], 0)
    enc.encode(batch)
    for s, (stream, rcon) in zip(seeds, enc.results(batch)):
        e = kat["frames"][str(s)]
        assert len(stream) == e["bytes"] and hashlib.sha256(stream).hexdigest() == e["sha256"], s
        assert hashlib.sha256(rcon.tobytes()).hexdigest() == e["rcon_sha256"], s      # the reconstruction too (the table holds both)
    enc.close()


def test_teams_fill_exactly_the_workgroups_they_are_given(amd):
    """Forward progress of teams: members of a team are launched in one grid and wait for each other, so every member must be
    resident.  The library sizes team launches from max_workgroups (never more than 3 T workgroups for T teams); here the context
    is told the device holds exactly 24 workgroups and 40 frames are queued: 8 teams of 3 serve them in five rounds."""
    import torch
    from oracle import oracle, synth
    imgs = [synth.syn(40 + i, 33 + (i % 5), i) for i in range(40)]
    enc = amd.DeviceEncoder(max_workgroups=24)
    enc.set_team(3)
    batch = enc.make_batch([torch.from_numpy(a).cuda() for a in imgs], 1)
    enc.encode(batch)
    team, nteams = enc.last_team()
    assert team == 3 and 3 * nteams <= 24
    for a, (s, r) in zip(imgs, enc.results(batch)):
        ws, wr, _ = oracle.cpu_encode(a, 1)
        assert s == ws and (r == wr).all()
    enc.close()


def test_residency_is_planned_from_the_occupancy_api_and_a_census(amd):
    """Every pool shape is planned against the workgroups that are resident at once: that number comes from the HIP occupancy API for
    the real block sizes and dynamic LDS (192 threads; 256 threads + the pipe wave's slice) and is confirmed by a census launch when
    the context is created — no hard-coded '4 per compute unit'."""
    enc = amd.DeviceEncoder()
    r = enc.residency()
    assert r["cus"] >= 1 and r["occ_per_cu"] >= 1 and r["occ_pipe_per_cu"] >= 1
    assert r["max_wg"] <= r["occ_per_cu"] * r["cus"] and r["pipe_wg"] <= r["occ_pipe_per_cu"] * r["cus"]
    assert r["census_wg"] == r["max_wg"] and r["census_pipe"] == r["pipe_wg"], r      # what was measured is what is planned with
    lib = amd.load_library()
    assert lib.imcvt_hevc_debug_census(enc.ctx, r["max_wg"]) == r["max_wg"]
    enc.close()


def test_forced_shape_beyond_the_context_is_an_error_before_anything_is_written(amd):
    """imcvt_hevc_set_shape(max_wg - 1, 1) asks for more main workgroups than there are mailboxes: IMCVT_ERR_ARG, and nothing was
    zeroed past the mailbox array on the way (the context keeps working, results unchanged)."""
    import torch
    from oracle import oracle, synth
    enc = amd.DeviceEncoder()
    r = enc.residency()
    imgs = [synth.syn(48 + i, 40, i) for i in range(r["max_wg"])]           # enough frames that the forced main count is not clipped to the batch
    batch = enc.make_batch([torch.from_numpy(a).cuda() for a in imgs], 1)
    enc.encode(batch); ref = enc.results(batch)
    for shape in ((r["max_wg"] - 1, 1), (r["max_wg"], r["max_wg"])):
        enc.set_shape(*shape)
        with pytest.raises(RuntimeError, match="bad argument"):
            enc.encode(batch)
    enc.set_shape(7, 0)                                  # half a shape is no shape: back to the automatic choice
    enc.encode(batch); got = enc.results(batch)
    assert all(a[0] == b[0] and (a[1] == b[1]).all() for a, b in zip(got, ref))
    for i in (0, 1, len(imgs) - 1):
        ws, wr, _ = oracle.cpu_encode(imgs[i], 1)
        assert got[i][0] == ws and (got[i][1] == wr).all()
    enc.close()


def test_sixteen_threads_share_one_device_batch(amd):
    """The reference's HEVCImageEncoder is re-entrant; here 16 host threads calling it at once are merged into one device batch
    (submission queue, hevc_hip.hip) instead of waiting for each other: byte-identical outputs, a handful of batches, and about one
    call's time rather than sixteen."""
    import threading, time
    from oracle import synth
    imgs = [synth.syn(512, 256, s) for s in range(16)]
    amd.HEVCImageEncoder(imgs[0], 0)                     # contexts exist, code is loaded
    t0 = time.time(); one = amd.HEVCImageEncoder(imgs[0], 0); t_one = time.time() - t0
    amd.hevc.coalesce_stats(reset=True)
    out = {}
    th = [threading.Thread(target=lambda i=i: out.__setitem__(i, amd.HEVCImageEncoder(imgs[i], 0))) for i in range(16)]
    t0 = time.time()
    for t in th: t.start()
    for t in th: t.join()
    t_all = time.time() - t0
    calls, batches, biggest = amd.hevc.coalesce_stats()
    assert calls == 16 and batches <= 4 and biggest >= 8, (calls, batches, biggest)
    assert out[0][0] == one[0]
    for i in range(16):
        s, r, d = amd.HEVCImageEncoderBatch([imgs[i]], 0)[0]
        assert out[i][0] == s and (out[i][1] == r).all() and out[i][2] == d, i
    assert t_all < 4 * t_one, (t_one, t_all)             # serialised it would be 16 x


def test_host_fan_out_over_four_logical_devices(amd):
    """HEVCImageEncoderBatch's device fan-out (frame i -> device i mod D, per-device context / stream / slab, collection in frame
    order) with D = 4 on a box with one GPU: IMCVT_HEVC_FAKE_DEVICES maps four logical devices onto device 0.  Same bytes as D = 1."""
    from oracle import synth
    lib = amd.load_library()
    imgs = [synth.syn(64 + 8 * (s % 5), 48 + 4 * (s % 3), s) for s in range(11)]
    qs = [s % 5 for s in range(11)]
    a = amd.HEVCImageEncoderBatch(imgs, qs)
    assert lib.imcvt_hevc_batch_devices() == 1
    lib.imcvt_hevc_shutdown()
    os.environ["IMCVT_HEVC_FAKE_DEVICES"] = "4"
    try:
        b = amd.HEVCImageEncoderBatch(imgs, qs)
        assert lib.imcvt_hevc_batch_devices() == 4
        c = amd.HEVCImageEncoderBatch(imgs[:3], qs[:3])   # fewer frames than devices: D = 3
        assert lib.imcvt_hevc_batch_devices() == 3
        one = amd.HEVCImageEncoder(imgs[5], qs[5])
    finally:
        del os.environ["IMCVT_HEVC_FAKE_DEVICES"]
        lib.imcvt_hevc_shutdown()
    for i in range(11):
        assert a[i][0] == b[i][0] and (a[i][1] == b[i][1]).all() and a[i][2] == b[i][2], i
    assert all(a[i][0] == c[i][0] for i in range(3)) and one[0] == a[5][0]
    assert amd.HEVCImageEncoderBatch(imgs[:2], qs[:2])[1][0] == a[1][0] and lib.imcvt_hevc_batch_devices() == 1


def test_pool_launch_beside_a_co_tenant_kernel(amd):
    """A pool launch planned for the whole device (512 main + 448 helper workgroups) while another kernel holds compute units: fewer
    of its workgroups are resident than planned.  Nothing in the pool depends on a workgroup that is not running (roles by placement,
    requests only once a helper reported in, late answers abandoned) — ten launches, identical results, no watchdog, no outlier."""
    import torch
    from oracle import synth
    lib = amd.load_library()
    enc = amd.DeviceEncoder()
    r = enc.residency()
    if r["max_wg"] < 1024:
        pytest.skip("planned for a device that holds 1024 workgroups")
    uniq = [torch.from_numpy(synth.syn(160, 96, s)).cuda() for s in range(32)]
    batch = enc.make_batch([uniq[i % 32] for i in range(512)], 0)
    enc.set_shape(512, 448)
    enc.encode(batch); ref = enc.results(batch); base_ms = enc.last_kernel_ms()
    side = torch.cuda.Stream()
    ms = []
    for rep in range(10):
        # 96 co-tenant workgroups of 256 threads that hold 100 KB of LDS each: 96 compute units take at most one encoder workgroup instead of four
        assert lib.imcvt_hevc_debug_filler(96, 100 * 1024, 400, side.cuda_stream) == 0
        enc.encode(batch); got = enc.results(batch)
        ms.append(enc.last_kernel_ms())
        assert enc.last_shape() == (512, 448)
        assert all(a[0] == b[0] and (a[1] == b[1]).all() for a, b in zip(got, ref)), rep
        side.synchronize()
    assert max(ms) < 6 * base_ms + 50, (base_ms, ms)
    from conftest import ROOT
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        open(os.path.join(out, "cotenant_test.log"), "w").write(f"512+448 workgroups, 512 x 160x96 q0: alone {base_ms:.1f} ms; beside 96 co-tenant workgroups (100 KB LDS each, 400 ms): {[round(v, 1) for v in ms]} ms; resident {enc.last_resident()}\n")
    enc.close()


def test_pool_as_two_launches_gives_identical_streams(amd):
    """A pool spread over TWO cooperating launches (hevc_hip.hip imcvt_hevc_plan_split: wide main workgroups on their own compute units, 192-thread
    helper workgroups on the others, streams with disjoint compute-unit masks): 120 frames with two, three and four helpers per compute unit give what
    one launch gives, and the CPU checker's bytes.  (The halves are fixed: 120 main workgroups on 128 compute units.)"""
    import torch
    from oracle import oracle, synth
    imgs = [synth.syn(64 + 32 * (s % 3), 64 + 32 * (s % 2), 100 + s) for s in range(120)]
    enc = amd.DeviceEncoder()
    batch = enc.make_batch([torch.from_numpy(a).cuda() for a in imgs], 0)
    enc.set_split(0)
    enc.encode(batch); ref = enc.results(batch)
    assert not enc.last_split()
    for i in (0, 37, 119):
        ws, wr, _ = oracle.cpu_encode(imgs[i], 0)
        assert ref[i][0] == ws and (ref[i][1] == wr).all(), i
    for hpc in (2, 3, 4):
        enc.set_split(1, hpc)
        enc.encode(batch); got = enc.results(batch)
        assert enc.last_split() and enc.last_wide() and enc.last_shape() == (120, min(128 * hpc, 480)), (hpc, enc.last_shape())
        for i, ((s, r), (s2, r2)) in enumerate(zip(got, ref)):
            assert s == s2 and (r == r2).all(), (hpc, i)
    enc.set_split(-1)
    enc.encode(batch)
    assert enc.last_split()                       # the default: where imcvt_hevc_plan_split says
    enc.close()


def test_wide_kernel_instantiations_agree(amd, monkeypatch):
    """Wide launches run hevc_wide.hip's instantiation of the kernel (256 registers per wavefront, loop-invariant code motion on); with
    IMCVT_HEVC_WIDE_KERNEL=0 the common instantiation.  Same bytes: a natural picture alone (1 main + 2 helper workgroups, all wide)."""
    import torch
    e = next(e for e in SMALL if e["input"].get("file") == "p5_gray.pgm" and e["qpd6"] == 0)
    img = torch.from_numpy(kat_input(e["input"])).cuda()
    out = []
    for knob in ("1", "0"):
        monkeypatch.setenv("IMCVT_HEVC_WIDE_KERNEL", knob)
        enc = amd.DeviceEncoder()
        b = enc.make_batch([img], 0)
        enc.encode(b)
        (s, r), = enc.results(b)
        assert enc.last_wide()
        out.append((s, r))
        enc.close()
    assert out[0][0] == out[1][0] and (out[0][1] == out[1][1]).all()
    assert hashlib.sha256(out[0][0]).hexdigest() == e["sha256"] and hashlib.sha256(out[0][1].tobytes()).hexdigest() == e["rcon_sha256"]


def test_host_batch_follows_the_running_launch(amd, monkeypatch):
    """The host-pointer entry point with a batch large enough for its round-6 transfers: inputs through pinned staging buffers (>= 32 MB), finished CTU
    rows of the reconstructions and finished stream bytes copied to the caller's buffers WHILE the launch runs (per-frame progress records,
    hevc_frame.h publish_progress).  Same streams and reconstructions as the plain path (copies, launch, copies), most bytes moved during the launch."""
    from imcvt_amd import hevc
    from oracle import oracle, synth
    lib = amd.load_library()
    imgs = [synth.syn(1024, 1024 - 8 * (s % 3), 300 + s) for s in range(36)]
    lib.imcvt_hevc_shutdown()
    a = amd.HEVCImageEncoderBatch(imgs, 1)
    xs = hevc.transfer_stats()
    assert xs["bytes_during"] > 0.5 * (xs["bytes_during"] + xs["bytes_after"]), xs
    total = sum(len(s) + r.size for s, r, _ in a)
    assert xs["bytes_during"] + xs["bytes_after"] == total, (xs, total)
    monkeypatch.setenv("IMCVT_HEVC_PLAIN_COPIES", "1")
    b = amd.HEVCImageEncoderBatch(imgs, 1)
    xs2 = hevc.transfer_stats()
    assert xs2["bytes_during"] == 0 and xs2["bytes_after"] == total
    monkeypatch.delenv("IMCVT_HEVC_PLAIN_COPIES")
    for i in range(len(imgs)):
        assert a[i][0] == b[i][0] and (a[i][1] == b[i][1]).all() and a[i][2] == b[i][2], i
    ws, wr, dims = oracle.cpu_encode(imgs[7], 1)
    assert a[7][0] == ws and (a[7][1] == wr).all() and a[7][2] == dims
    lib.imcvt_hevc_shutdown()


def test_partner_workgroups_give_identical_streams(amd):
    """Wide pools with PARTNER workgroups (hevc_frame.h "8x8 CUs with a partner workgroup": the two 2Nx2N candidate sets of every 8x8 CU on a second compute unit, the
    main workgroup walking the NxN chain alone): the golden vectors one frame at a time (1 main + 1 partner + 2 helpers) and all in one launch, partners on
    and off — the reference's streams and reconstructions either way."""
    import torch
    es = [e for e in SMALL if e["qpd6"] in (0, 4)]
    enc = amd.DeviceEncoder()
    for e in es[:6] + [e for e in es if e["input"].get("file") == "p5_gray.pgm"]:
        b = enc.make_batch([torch.from_numpy(kat_input(e["input"]).copy()).cuda()], e["qpd6"])
        enc.encode(b)
        (s, r), = enc.results(b)
        assert enc.last_wide() and enc.last_partners() == 1 and enc.last_shape() == (1, 2), kat_id(e)
        assert len(s) == e["bytes"] and hashlib.sha256(s).hexdigest() == e["sha256"] and hashlib.sha256(r.tobytes()).hexdigest() == e["rcon_sha256"], kat_id(e)
    for q in (0, 4):
        qs = [e for e in es if e["qpd6"] == q]
        batch = enc.make_batch([torch.from_numpy(kat_input(e["input"]).copy()).cuda() for e in qs], q)
        for mode in (1, 0):
            enc.set_partners(mode)
            enc.encode(batch)
            assert enc.last_wide() and enc.last_partners() == (len(qs) if mode else 0), (q, mode, enc.last_shape())
            for e, (s, r) in zip(qs, enc.results(batch)):
                assert len(s) == e["bytes"] and hashlib.sha256(s).hexdigest() == e["sha256"] and hashlib.sha256(r.tobytes()).hexdigest() == e["rcon_sha256"], (kat_id(e), mode)
    enc.close()


@pytest.mark.parametrize("seed", [3, 4])
def test_partner_workgroups_1080p_digest(amd, seed):
    """One 1080p frame (BASELINE config 2's shape) with a partner workgroup — 1 main + 1 partner + 2 helper workgroups, a compute unit each — against the
    reference's digests, stream and reconstruction."""
    import torch
    from oracle import synth
    e = next(e for e in LARGE if e["input"].get("arg") == seed and e["qpd6"] == 0)
    enc = amd.DeviceEncoder()
    b = enc.make_batch([torch.from_numpy(synth.syn(1920, 1080, seed)).cuda()], 0)
    enc.encode(b)
    (s, r), = enc.results(b)
    assert enc.last_wide() and enc.last_partners() == 1
    assert len(s) == e["bytes"] and hashlib.sha256(s).hexdigest() == e["sha256"] and hashlib.sha256(r.tobytes()).hexdigest() == e["rcon_sha256"]
    enc.close()


@pytest.mark.slow
def test_two_launch_pool_full_size_against_reference_digests(amd):
    """BASELINE configs[3]'s share of one GPU at N = 4 — the first 128 bench frames at full size — as the library runs it by default: two cooperating launches, 128 wide main
    workgroups on one half of the compute units, 192-thread helpers on the other.  Streams and reconstructions against the digests the REAL reference produced
    (tests/golden/bench512_kat.json)."""
    import json
    import torch
    from conftest import ROOT
    from oracle import synth
    kat = json.load(open(os.path.join(ROOT, "tests", "golden", "bench512_kat.json")))["frames"]
    enc = amd.DeviceEncoder()
    batch = enc.make_batch([torch.from_numpy(synth.syn(1920, 1080, s)).cuda() for s in range(128)], 0)
    enc.encode(batch)
    assert enc.last_split() and enc.last_wide() and enc.last_shape()[0] == 128
    for s, (stream, rcon) in enumerate(enc.results(batch)):
        e = kat[str(s)]
        assert len(stream) == e["bytes"] and hashlib.sha256(stream).hexdigest() == e["sha256"], s
        assert hashlib.sha256(rcon.tobytes()).hexdigest() == e["rcon_sha256"], s
    enc.close()
