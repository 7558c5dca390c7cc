"""CPU tests of the boundary: the C-ABI library loads, exports every symbol include/imcvt_hevc.h declares, and
fails LOUDLY without a device (no CPU fallback).  No compute calls are made here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "imcvt_hevc.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set(re.findall(r"\b([A-Za-z_]\w*)\s*\(", src))
    return sorted(n for n in names if n.startswith(("imcvt_hevc_", "HEVCImage", "writeHEVC")) and n != "imcvt_hevc_frame")


def test_library_exports_every_declared_symbol(built):
    import imcvt_amd
    lib = imcvt_amd.load_library()
    syms = declared_symbols()
    assert set(syms) == set(imcvt_amd.hevc.EXPORTS), (syms, imcvt_amd.hevc.EXPORTS)
    for s in syms:
        assert hasattr(lib, s), s
    assert b"gfx950" in lib.imcvt_hevc_version()


def test_geometry_helpers(built):
    import imcvt_amd
    lib = imcvt_amd.load_library()
    for v, want in [(1, 32), (32, 32), (33, 64), (1080, 1088), (8192, 8192), (9000, 8192)]:     # reference :1580-1581
        assert lib.imcvt_hevc_padded(v) == want == imcvt_amd.padded(v)
    assert lib.imcvt_hevc_stream_bound(1080, 1920) == 2 * (1920 + 32) * (1080 + 32) + 65536   # src/imageio_hevc.c:14


def test_frame_descriptor_layout(built):
    import imcvt_amd
    f = imcvt_amd.hevc.imcvt_hevc_frame
    assert C.sizeof(f) == 48 and f.h.offset == 32 and f.qpd6.offset == 40


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_no_device_fails_loudly(built, capfd):
    import imcvt_amd
    with pytest.raises(RuntimeError, match="no HIP device"):
        imcvt_amd.HEVCImageEncoder(np.zeros((32, 32), np.uint8), 0)
    with pytest.raises(RuntimeError):
        imcvt_amd.DeviceEncoder()
    assert imcvt_amd.writeHEVCImageFile("/tmp/_never.h265", np.zeros((8, 8), np.uint8), False, 8, 8, 0) == 1   # reference: 1 = failed
    assert "no CPU fallback" in capfd.readouterr().err


def test_product_never_imports_oracle():
    """The product package must not reference the checker."""
    for root, _, files in os.walk(os.path.join(ROOT, "imcvt_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(root, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "liboracle" not in txt and "libref_" not in txt, f


def test_shipped_kernel_uses_the_matrix_cores(built, tmp_path):
    """The gfx950 code object inside libimcvt_hevc.so really contains the two i8 matrix instructions the N = 16 / 32 transforms are
    written for (hevc_core.h mx_mm) — checked on the bundled device code, no GPU needed."""
    import shutil, subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump of the ROCm toolchain not found")
    so = tmp_path / "libimcvt_hevc.so"
    shutil.copy(os.path.join(ROOT, "imcvt_amd", "csrc", "libimcvt_hevc.so"), so)
    subprocess.run([objdump, "--offloading", str(so)], check=True, capture_output=True)
    co = sorted(f for f in os.listdir(tmp_path) if "gfx950" in f)
    assert len(co) == 2, os.listdir(tmp_path)          # hevc_hip.hip's (every launch shape) and hevc_wide.hip's (wide launches: 256 registers per wavefront)
    asms = [subprocess.run([objdump, "-d", "--mcpu=gfx950", str(tmp_path / f)], check=True, capture_output=True, text=True).stdout for f in co]
    assert sum("hevc_encode_frames_wide" in a for a in asms) == 1
    for asm in asms:
        _check_code_object(asm)


def _check_code_object(asm):
    assert asm.count("v_mfma_i32_32x32x32_i8") >= 9 and asm.count("v_mfma_i32_16x16x32_i8") >= 36, (asm.count("v_mfma_i32_32x32x32_i8"), asm.count("v_mfma_i32_16x16x32_i8"))
    # The candidate-set functions are out of line and use every vector register; none of their callers keeps anything in a
    # callee-saved one, and the backend drops the saves once no call site carries LLVM's `tail` marker (HDN, hevc_core.h).
    # A toolchain that brings them back (64 stores at the top of eval_2Nx2N, one per callee-saved register v40..v159) costs
    # ~2 MB of scratch traffic per CTU each way: look at the first instructions of the function.
    import re
    m = re.search(r"<_Z10eval_2Nx2Niiiiii>:\n((?:.*\n){60})", asm)
    assert m, "eval_2Nx2N is expected to be an out-of-line function of the code object"
    head = m.group(1)
    assert head.count("scratch_store_dword") <= 8, head


REF_SRC = "/root/reference/src"


@pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="the reference checkout exists only in the dev container")
def test_reference_main_links_against_the_library_unchanged(built, tmp_path):
    """the reference's own main.c + format modules, minus HEVCe.c and imageio_hevc.c, link against
    libimcvt_hevc.so with no source change; writeHEVCImageFile resolves to the library, and the binary fails loudly
    (reference error line, exit code 1) on a machine without a device."""
    import subprocess
    csrc = os.path.join(ROOT, "imcvt_amd", "csrc")
    inc = tmp_path / "inc"
    inc.mkdir()
    os.symlink(os.path.join(REF_SRC, "uPNG", "uPNG.h"), inc / "upng.h")      # the reference's own case bug, not ours
    exe = str(tmp_path / "ImCvt")
    srcs = [os.path.join(REF_SRC, f) for f in ("main.c", "imageio_pnm.c", "imageio_png.c", "imageio_bmp.c", "imageio_qoi.c", "imageio_jls.c", "uPNG/uPNG.c")]
    subprocess.run(["gcc", *srcs, "-O1", "-w", "-I", str(inc), "-o", exe, "-L" + csrc, "-limcvt_hevc", "-Wl,-rpath," + csrc], check=True)
    nm = subprocess.run(["nm", "-D", "--undefined-only", exe], capture_output=True, text=True, check=True).stdout
    assert "writeHEVCImageFile" in nm and "HEVCImageEncoder" not in nm.replace("writeHEVCImageFile", "")
    ldd = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "libimcvt_hevc.so" in ldd and "not found" not in ldd
    if not os.path.exists("/dev/kfd"):
        src = tmp_path / "in.pgm"
        src.write_bytes(b"P5\n8 8\n255\n" + bytes(64))
        r = subprocess.run([exe, str(src), "-o", str(tmp_path / "out.h265")], capture_output=True, text=True)
        assert r.returncode == 1 and "***ERROR" in r.stdout and "no CPU fallback" in r.stderr
        assert not (tmp_path / "out.h265").exists()


def test_launch_shape_policy(built):
    """imcvt_hevc_plan is pure (no device): main workgroups plus a pool of helper workgroups while the batch leaves the device
    room (at most two helpers per main workgroup — it has two requests outstanding at most — and never more workgroups than are
    resident), frames per workgroup when the batch fills the device."""
    import imcvt_amd
    lib = imcvt_amd.load_library()

    def plan(n, wg=1024, force=0):
        m, h = C.c_int(-1), C.c_int(-1)
        return lib.imcvt_hevc_plan(n, wg, force, C.byref(m), C.byref(h)), m.value, h.value

    assert plan(1) == (2, 1, 2) and plan(64) == (2, 64, 128) and plan(320) == (2, 320, 640) and plan(342) == (2, 342, 682)
    assert plan(512) == (2, 512, 512) and plan(640) == (2, 512, 512)      # the mains pull the remaining frames as they finish
    assert plan(641) == (1, 641, 0) and plan(1000) == (1, 1000, 0) and plan(5000) == (1, 1024, 0)
    for n in range(1, 1400, 7):
        mode, m, h = plan(n)
        assert mode in (1, 2) and 1 <= m <= min(n, 1024) and m + h <= 1024
        assert (h == 0) if mode == 1 else (0 < h <= 2 * m and m <= 512 and m + h <= 1024)
    assert plan(100, force=1) == (1, 100, 0) and plan(100, force=2) == (2, 100, 100) and plan(100, force=3) == (2, 100, 200)
    assert plan(2000, force=3) == (2, 341, 682) and plan(2000, force=2) == (2, 512, 512)
    assert plan(10, wg=16) == (2, 8, 8) and plan(11, wg=16) == (1, 11, 0) and plan(5, wg=16) == (2, 5, 10)
    assert plan(0) == (1, 0, 0)

    # pipe wave (256-thread workgroups, three per CU): pure too — used when the planned launch fits 15/16 of 768 slots; a pool that just
    # misses that gives up helpers for it, down to 1.5 per main workgroup
    def pipe(n, wg=1024, forced=0):
        mode, m, h = plan(n, wg)
        mm, hh = C.c_int(m), C.c_int(h)
        return lib.imcvt_hevc_plan_pipe(mode, wg, forced, C.byref(mm), C.byref(hh)), mm.value, hh.value

    assert pipe(1) == (1, 1, 2) and pipe(64) == (1, 64, 128) and pipe(240) == (1, 240, 480)
    assert pipe(256) == (1, 256, 464) and pipe(288) == (1, 288, 432)        # helpers cut to 720 - n
    assert pipe(289) == (0, 289, 578) and pipe(512) == (0, 512, 512)        # (less than 1.5 helpers per main would be left: no pipe wave)
    assert pipe(700) == (1, 700, 0) and pipe(720) == (1, 720, 0) and pipe(721) == (0, 721, 0)      # a frame per workgroup
    assert pipe(5, wg=16) == (0, 5, 10) and pipe(3, wg=16) == (1, 3, 6)     # 12 slots of 256 threads, 11 usable... 9 workgroups fit
    for n in range(1, 1400, 5):
        mode, m, h = plan(n)
        use, mm, hh = pipe(n)
        assert mm == m and hh <= h and (hh == h or (use and mm + hh == 720 and 2 * hh >= 3 * mm))
        assert use == (mm + hh <= 720)
    m, h = C.c_int(512), C.c_int(256)
    assert lib.imcvt_hevc_plan_pipe(2, 1024, 1, C.byref(m), C.byref(h)) == 1 and (m.value, h.value) == (512, 256)      # a forced shape may fill the last slot
    # wide workgroups (512 threads, one per compute unit): whenever a pipe-wave launch leaves every workgroup a compute unit of its own, a sixteenth to spare
    assert lib.imcvt_hevc_plan_wide(1, 3, 256, 0) == 1 and lib.imcvt_hevc_plan_wide(1, 192, 256, 0) == 1 and lib.imcvt_hevc_plan_wide(1, 240, 256, 0) == 1
    assert lib.imcvt_hevc_plan_wide(1, 241, 256, 0) == 0 and lib.imcvt_hevc_plan_wide(1, 256, 256, 1) == 1 and lib.imcvt_hevc_plan_wide(1, 257, 256, 1) == 0
    assert lib.imcvt_hevc_plan_wide(0, 3, 256, 0) == 0 and lib.imcvt_hevc_plan_wide(1, 3, 0, 0) == 0
    def wide_pool(mm, hh, mode=2, forced=0, wg=256):
        m, h = C.c_int(mm), C.c_int(hh)
        return lib.imcvt_hevc_plan_wide_pool(1, mode, forced, wg, C.byref(m), C.byref(h)), h.value
    assert wide_pool(64, 128) == (1, 128)            # fits as planned
    assert wide_pool(128, 256) == (1, 128)           # the main workgroups take half of the compute units: the helpers get the other half
    assert wide_pool(100, 200) == (1, 156)
    assert wide_pool(129, 258) == (0, 258)           # more main workgroups than that: 256-thread workgroups as planned
    assert wide_pool(128, 256, forced=1) == (0, 256) # a forced shape is never changed
    assert wide_pool(81, 162) == (1, 162)            # 243 workgroups miss the 15/16 cap but fit the compute units: helpers are cut, never raised (advisor, round 5)
    for mm in range(1, 129):
        use, hh = wide_pool(mm, 2 * mm)
        assert use == 1 and hh <= 2 * mm and mm + hh <= 256 and hh >= min(2 * mm, 256 - mm)


def test_split_launch_policy(built):
    """imcvt_hevc_plan_split is pure: a pool runs as two cooperating launches (wide main workgroups on their own compute units, 192-thread helpers on
    the others) only where the one-launch wide shape would leave a main workgroup fewer than two helpers and the helpers' compute units can keep up."""
    import imcvt_amd
    lib = imcvt_amd.load_library()

    def split(m, cus=256, wide_wg=256, occ=4, hpc=0, mode=2):
        h = C.c_int(-1)
        return lib.imcvt_hevc_plan_split(mode, m, cus, wide_wg, occ, hpc, C.byref(h)), h.value

    assert split(64) == (0, -1) and split(80) == (0, -1) and split(96) == (0, -1) and split(106) == (0, -1)      # one launch of wide workgroups leaves 1.4 helpers per main workgroup or more
    assert split(107) == (1, 384) and split(112) == (1, 384) and split(128) == (1, 384)      # half of the compute units x 3 helpers, at most four per main workgroup
    assert split(128, hpc=2) == (1, 256) and split(128, hpc=4) == (1, 512) and split(128, hpc=9) == (1, 512)
    assert split(129) == (0, -1)                                       # beyond half of the compute units: as planned
    assert split(120, mode=1) == (0, -1) and split(120, wide_wg=200) == (0, -1) and split(120, wide_wg=250)[0] == 1      # no pool / fewer wide workgroups resident than compute units
    for m in range(1, 300):
        use, h = split(m)
        assert use in (0, 1) and (not use or (m <= h <= 4 * m and 106 < m <= 128))


def test_partner_workgroup_policy(built):
    """imcvt_hevc_plan_partners is pure: a wide pool gets one partner workgroup per main workgroup (the 2Nx2N sets of its 8x8 CUs on a second compute unit) when the
    launch has room beside the planned helpers, or with the helpers cut to what is left as long as 1.5 per main workgroup remain."""
    import imcvt_amd
    lib = imcvt_amd.load_library()

    def part(m, h, wg=256, forced=0):
        hh = C.c_int(h)
        return lib.imcvt_hevc_plan_partners(m, C.byref(hh), wg, forced), hh.value

    assert part(1, 2) == (1, 2) and part(48, 96) == (48, 96) and part(60, 120) == (60, 120)       # 4 x 60 = 240 = 15/16 of the compute units
    assert part(61, 122) == (61, 118) and part(64, 128) == (64, 112) and part(68, 136) == (68, 104)   # helpers cut to the rest, 1.5 per main workgroup at least
    assert part(69, 138) == (0, 138) and part(80, 160) == (0, 160)                                  # no room: as planned
    assert part(64, 128, forced=1) == (64, 128) and part(65, 130, forced=1) == (0, 130)           # a forced shape may fill the last compute unit, and is never changed
    for m in range(1, 130):
        n, h = part(m, 2 * m)
        assert n in (0, m) and (n == 0 or (2 * m + h <= 240 and 2 * h >= 3 * m)) and (n != 0 or h == 2 * m)


def test_submission_queue_merges_concurrent_callers(built):
    """The reference's HEVCImageEncoder is re-entrant (src/HEVCe/HEVCe.c:1569, no mutable globals).  Here concurrent calls are merged
    into one device batch by a submission queue (hevc_hip.hip): exercised without a GPU through a stand-in for the device batch that
    records what it is handed.  16 threads with a frame each -> far fewer batches than calls, every caller gets ITS result back, and
    a call that arrives while a batch runs joins the next one."""
    import threading, time
    import imcvt_amd
    lib = imcvt_amd.load_library()
    u8p = C.POINTER(C.c_ubyte); ip = C.POINTER(C.c_int)
    seen = []

    @C.CFUNCTYPE(C.c_int, C.c_int, C.POINTER(u8p), C.POINTER(u8p), C.POINTER(u8p), ip, ip, ip, ip)
    def backend(n, pbuffers, imgs, rcons, ysz, xsz, qpd6, out_len):
        seen.append(n)
        time.sleep(0.05)                                   # "the kernel runs": later callers pile up meanwhile
        for i in range(n):
            tag = imgs[i][0]                               # the caller's first pixel
            pbuffers[i][0] = tag; rcons[i][0] = tag ^ 0xFF
            out_len[i] = 100 + tag + qpd6[i]; ysz[i] = ysz[i] + 1000; xsz[i] = xsz[i] + 2000
        return 0

    lib.imcvt_hevc_debug_set_backend(C.cast(backend, C.c_void_p))
    try:
        imcvt_amd.hevc.coalesce_stats(reset=True)
        results = {}

        def call(t):
            img = np.full(64, t, np.uint8); out = np.zeros(16, np.uint8); rc = np.zeros(16, np.uint8)
            ys, xs = C.c_int(8), C.c_int(8)
            n = lib.HEVCImageEncoder(out.ctypes.data_as(u8p), img.ctypes.data_as(u8p), rc.ctypes.data_as(u8p), C.byref(ys), C.byref(xs), t % 5)
            results[t] = (n, int(out[0]), int(rc[0]), ys.value, xs.value)

        th = [threading.Thread(target=call, args=(t,)) for t in range(16)]
        for t in th: t.start()
        for t in th: t.join()
        for t in range(16):
            assert results[t] == (100 + t + t % 5, t, t ^ 0xFF, 1008, 2008), (t, results[t])
        calls, batches, biggest = imcvt_amd.hevc.coalesce_stats()
        assert calls == 16 and sum(seen) == 16 and batches == len(seen)
        assert batches <= 4 and biggest >= 8, (seen, batches, biggest)      # one or two rounds in practice: the first few callers, then everyone who arrived meanwhile
        # an error of the batch reaches every caller of that batch; bad arguments never reach the queue
        assert lib.HEVCImageEncoder(None, None, None, None, None, 0) == -3
        assert imcvt_amd.hevc.coalesce_stats()[0] == 16
        # a multi-frame call is one submission
        imgs = [np.full(64, 40 + i, np.uint8) for i in range(3)]; outs = [np.zeros(16, np.uint8) for _ in range(3)]; rcs = [np.zeros(16, np.uint8) for _ in range(3)]
        P = u8p * 3
        ys = (C.c_int * 3)(8, 8, 8); xs = (C.c_int * 3)(8, 8, 8); qv = (C.c_int * 3)(0, 1, 2); lens = (C.c_int * 3)()
        assert lib.HEVCImageEncoderBatch(3, P(*[o.ctypes.data_as(u8p) for o in outs]), P(*[a.ctypes.data_as(u8p) for a in imgs]), P(*[r.ctypes.data_as(u8p) for r in rcs]), ys, xs, qv, lens) == 0
        assert list(lens) == [140, 142, 144] and seen[-1] == 3
        # a merged batch that fails on its ARGUMENTS (-3) is retried submission by submission: only the caller whose frame cannot be encoded sees the error
        failing = {"on": True}

        @C.CFUNCTYPE(C.c_int, C.c_int, C.POINTER(u8p), C.POINTER(u8p), C.POINTER(u8p), ip, ip, ip, ip)
        def picky(n, pbuffers, imgs, rcons, ysz, xsz, qpd6, out_len):
            time.sleep(0.02)
            if any(imgs[i][0] == 7 for i in range(n)):
                return -3
            for i in range(n):
                out_len[i] = 200 + imgs[i][0]
            return 0

        lib.imcvt_hevc_debug_set_backend(C.cast(picky, C.c_void_p))
        res2 = {}

        def call2(t):
            img = np.full(64, t, np.uint8); out = np.zeros(16, np.uint8); rc = np.zeros(16, np.uint8)
            ys, xs = C.c_int(8), C.c_int(8)
            res2[t] = lib.HEVCImageEncoder(out.ctypes.data_as(u8p), img.ctypes.data_as(u8p), rc.ctypes.data_as(u8p), C.byref(ys), C.byref(xs), 0)

        th = [threading.Thread(target=call2, args=(t,)) for t in range(12)]
        for t in th: t.start()
        for t in th: t.join()
        assert res2[7] == -3 and all(res2[t] == 200 + t for t in range(12) if t != 7), res2
        # a device-wide failure (HIP error, watchdog, no device) reaches every caller of the round as it is: no relaunch per caller on a broken device
        broken_calls = []

        @C.CFUNCTYPE(C.c_int, C.c_int, C.POINTER(u8p), C.POINTER(u8p), C.POINTER(u8p), ip, ip, ip, ip)
        def broken(n, pbuffers, imgs, rcons, ysz, xsz, qpd6, out_len):
            broken_calls.append(n)
            time.sleep(0.05)
            return -2

        lib.imcvt_hevc_debug_set_backend(C.cast(broken, C.c_void_p))
        res3 = {}

        def call3(t):
            img = np.full(64, t, np.uint8); out = np.zeros(16, np.uint8); rc = np.zeros(16, np.uint8)
            ys, xs = C.c_int(8), C.c_int(8)
            res3[t] = lib.HEVCImageEncoder(out.ctypes.data_as(u8p), img.ctypes.data_as(u8p), rc.ctypes.data_as(u8p), C.byref(ys), C.byref(xs), 0)

        th = [threading.Thread(target=call3, args=(t,)) for t in range(12)]
        for t in th: t.start()
        for t in th: t.join()
        assert all(res3[t] == -2 for t in range(12)), res3
        assert sum(broken_calls) == 12 and len(broken_calls) <= 4, broken_calls      # every frame handed to the device once, in a few merged rounds
    finally:
        lib.imcvt_hevc_debug_set_backend(None)
