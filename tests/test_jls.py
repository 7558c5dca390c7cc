"""JPEG-LS path (BASELINE config 5).
CPU: the DEVICE source (imcvt_amd/csrc/jls_core.h) compiled for the host by tests/hostemu/jls_hostemu.cpp — the serial walk
is ordinary C++, so this is the kernel's own logic — against the golden vectors generated from the compiled reference.
GPU (-m gpu): the HIP path through the C ABI against the same vectors, the CPU checker on seeded inputs, the 1080p / 4K
golden digests, a device-resident batch, and the reference's file-writer entry point."""
import base64
import ctypes as C
import hashlib
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from test_jls_oracle import KAT, SMALL, LARGE, jls_id, jls_input

u8p = C.POINTER(C.c_ubyte)


@pytest.fixture(scope="module")
def jls_emu(built):
    d = os.path.join(ROOT, "tests", "hostemu")
    so, src = os.path.join(d, "libjls_hostemu.so"), os.path.join(d, "jls_hostemu.cpp")
    deps = [src] + [os.path.join(ROOT, "imcvt_amd", "csrc", f) for f in ("jls_core.h", "jls_par.h")]
    if not os.path.exists(so) or max(os.path.getmtime(f) for f in deps) > os.path.getmtime(so):
        tmp = f"{so}.tmp.{os.getpid()}"            # private file + rename: parallel test workers never load a half-written library
        subprocess.run(["g++", "-O2", "-fPIC", "-shared", "-o", tmp, src], check=True)
        os.replace(tmp, so)
    lib = C.CDLL(so)
    lib.jls_hostemu_encode.restype = C.c_longlong
    lib.jls_hostemu_encode.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, u8p]
    lib.jls_hostemu_encode_par.restype = C.c_longlong
    lib.jls_hostemu_encode_par.argtypes = [u8p, C.c_int, C.c_int, C.c_int, u8p]

    def enc(img, near, par=False):
        img = np.ascontiguousarray(img)
        h, w = img.shape[:2]
        out = np.zeros((8 * w * h + 65536) * (3 if img.ndim == 3 else 1), np.uint8)
        if par:
            assert near == 0
            n = lib.jls_hostemu_encode_par(img.ctypes.data_as(u8p), int(img.ndim == 3), h, w, out.ctypes.data_as(u8p))
        else:
            n = lib.jls_hostemu_encode(img.ctypes.data_as(u8p), int(img.ndim == 3), h, w, near, out.ctypes.data_as(u8p))
        return out[:n].tobytes()
    return enc


def _check(got, e):
    assert len(got) == e["bytes"] and hashlib.sha256(got).hexdigest() == e["sha256"]
    if "stream_b64" in e:
        assert got == base64.b64decode(e["stream_b64"])


@pytest.mark.parametrize("e", SMALL + [e for e in LARGE if e["input"]["w"] == 1920], ids=jls_id)
def test_device_source_matches_reference_vectors(jls_emu, e):
    _check(jls_emu(jls_input(e["input"]), e["near"]), e)


@pytest.mark.parametrize("e", [e for e in SMALL if e["near"] == 0] + [e for e in LARGE if e["input"]["w"] == 1920 and e["near"] == 0], ids=jls_id)
def test_plane_parallel_source_matches_reference_vectors(jls_emu, e):
    """The plane-parallel lossless path (csrc/jls_par.h: classification, per-context chains, prefix sums of code lengths, chunked
    bit stuffing) with every grid run as a loop on the host: the reference's bytes."""
    _check(jls_emu(jls_input(e["input"]), 0, par=True), e)


def test_plane_parallel_source_vs_cpu_checker_seeded(jls_emu):
    # runs, run interruptions at row ends, binary pictures (maximal 0xFF stuffing), single rows / columns, RGB
    from oracle import oracle
    rng = np.random.default_rng(5)
    for i in range(48):
        h, w = int(rng.integers(1, 90)), int(rng.integers(1, 90))
        kind = i % 6
        shape = (h, w, 3) if kind == 4 else (h, w)
        img = (rng.integers(0, 256, shape) if kind in (0, 4) else np.clip(rng.normal(128, 3, shape), 0, 255) if kind == 1
               else np.full(shape, int(rng.integers(0, 256))) if kind == 2 else (rng.integers(0, 2, shape) * 255) if kind == 3
               else np.repeat(rng.integers(0, 256, (h, 1)), w, axis=1)).astype(np.uint8)
        assert jls_emu(img, 0, par=True) == oracle.jls_cpu_encode(img, 0), (i, shape, kind)


# (seed, h, w, unstuffed bits mod 16384) of noise pictures whose scan ends 1..8 bits past a 16 Kbit stuffing-chunk edge — the
# hand-over between the last two chunks (a byte of chunk c that ends beyond the end of the scan) that round 2 got wrong in
# one picture out of ~5000.  Found by search with the serial walker as the judge; the test re-checks the remainder.
CHUNK_EDGE_CASES = [(3694, 49, 72, 1), (22544, 42, 85, 2), (23965, 32, 52, 2), (9236, 32, 53, 3), (37995, 50, 70, 3), (38401, 44, 80, 3), (17247, 56, 63, 3),
                    (6373, 58, 61, 4), (34627, 41, 87, 4), (19589, 34, 49, 5), (37468, 44, 81, 5), (28385, 41, 86, 6), (12697, 56, 63, 7), (10014, 36, 46, 8)]


def _edge_img(seed, h, w):
    rng = np.random.default_rng(seed)
    assert (int(rng.integers(30, 60)), int(rng.integers(40, 90))) == (h, w)      # (the search drew the size from the same generator)
    return rng.integers(0, 256, (h, w), dtype=np.uint8)


@pytest.fixture(scope="module")
def jls_emu_libs(built):
    """The host build of the plane-parallel source twice: the product's 16 Kbit stuffing chunks, and 64-bit chunks (a chunk edge
    every few pixels, so every picture exercises every hand-over state)."""
    d = os.path.join(ROOT, "tests", "hostemu")
    src = os.path.join(d, "jls_hostemu.cpp")
    deps = [src] + [os.path.join(ROOT, "imcvt_amd", "csrc", f) for f in ("jls_core.h", "jls_par.h")]
    libs = {}
    for name, flags in (("libjls_hostemu.so", []), ("libjls_hostemu_c64.so", ["-DJLS_CHUNK_BITS=64"])):
        so = os.path.join(d, name)
        if not os.path.exists(so) or max(os.path.getmtime(f) for f in deps) > os.path.getmtime(so):
            tmp = f"{so}.tmp.{os.getpid()}"
            subprocess.run(["g++", "-O2", "-fPIC", "-shared", *flags, "-o", tmp, src], check=True)
            os.replace(tmp, so)
        lib = C.CDLL(so)
        lib.jls_hostemu_encode.restype = C.c_longlong
        lib.jls_hostemu_encode.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, u8p]
        lib.jls_hostemu_encode_par.restype = C.c_longlong
        lib.jls_hostemu_encode_par.argtypes = [u8p, C.c_int, C.c_int, C.c_int, u8p]
        lib.jls_hostemu_last_bits.restype = C.c_longlong
        libs[name] = lib
    assert libs["libjls_hostemu.so"].jls_hostemu_chunk_bits() == 16384 and libs["libjls_hostemu_c64.so"].jls_hostemu_chunk_bits() == 64
    return libs["libjls_hostemu.so"], libs["libjls_hostemu_c64.so"]


def _both(lib, img):
    h, w = img.shape
    o1 = np.zeros(8 * w * h + 65536, np.uint8); o2 = np.zeros_like(o1)
    a = lib.jls_hostemu_encode(img.ctypes.data_as(u8p), 0, h, w, 0, o1.ctypes.data_as(u8p))
    b = lib.jls_hostemu_encode_par(img.ctypes.data_as(u8p), 0, h, w, o2.ctypes.data_as(u8p))
    return o1[:a].tobytes(), o2[:b].tobytes(), int(lib.jls_hostemu_last_bits())


def test_plane_parallel_stuffing_at_chunk_edges(jls_emu_libs):
    """Scans that end 1..8 bits past a 16 Kbit chunk edge (the product's chunk size): serial walker == plane-parallel == CPU checker."""
    from oracle import oracle
    lib, _ = jls_emu_libs
    seen = set()
    for seed, h, w, rem in CHUNK_EDGE_CASES:
        img = _edge_img(seed, h, w)
        ser, par, bits = _both(lib, img)
        assert bits % 16384 == rem, "the case no longer lands where it was aimed"
        assert par == ser == oracle.jls_cpu_encode(img, 0), (seed, h, w, rem)
        seen.add(rem)
    assert seen == set(range(1, 9))


def test_plane_parallel_stuffing_small_chunks_fuzz(jls_emu_libs):
    """The same source built with 64-bit stuffing chunks: every picture crosses hundreds of chunk edges in every entry state
    (offset 0..7, after-0xFF or not), including scans that end 0..8 bits past an edge and 0xFF-heavy binary pictures."""
    _, lib = jls_emu_libs
    rng = np.random.default_rng(77)
    rems = set()
    for i in range(400):
        h, w = int(rng.integers(1, 40)), int(rng.integers(1, 48))
        kind = i % 4
        img = (rng.integers(0, 256, (h, w)) if kind == 0 else rng.integers(0, 2, (h, w)) * 255 if kind == 1
               else np.clip(rng.normal(128, 2, (h, w)), 0, 255) if kind == 2 else rng.integers(250, 256, (h, w))).astype(np.uint8)
        ser, par, bits = _both(lib, img)
        assert par == ser, (i, h, w, kind, bits % 64)
        rems.add(bits % 64)
    assert set(range(0, 9)) <= rems


def test_reciprocal_quantiser_is_exact():
    # jls_core.h quant_err: n / quant as (n * ceil(2^20 / quant)) >> 20 for every quant = 2*near+1 the ABI admits and every n it can see
    for q in range(1, 512, 2):
        r = ((1 << 20) + q - 1) // q
        n = np.arange(1024, dtype=np.int64)
        assert ((n * r) >> 20 == n // q).all() and 1023 * r < 2 ** 31


def test_library_exports_the_declared_symbols(built):
    lib = C.CDLL(os.path.join(ROOT, "imcvt_amd", "csrc", "libimcvt_jls.so"))
    for name in ("writeJLSImageFile", "imcvt_jls_encode", "imcvt_jls_stream_bound", "imcvt_jls_encode_device", "imcvt_jls_last_kernel_ms", "imcvt_jls_version", "imcvt_jls_last_path"):
        assert hasattr(lib, name), name
    lib.imcvt_jls_stream_bound.restype = C.c_longlong
    assert lib.imcvt_jls_stream_bound(1080, 1920) == 8 * 1920 * 1080 + 65536


# ---- GPU ---------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def jls_gpu(built):
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from imcvt_amd import jls
    jls.load_jls_library()
    return jls


@pytest.mark.gpu
@pytest.mark.parametrize("e", SMALL, ids=jls_id)
def test_gpu_matches_reference_vectors(jls_gpu, e):
    _check(jls_gpu.JLSencode(jls_input(e["input"]), e["near"]), e)


@pytest.mark.gpu
def test_gpu_large_frames_golden_digests_in_one_batch(jls_gpu):
    big = [e for e in LARGE if e["near"] == 0]
    got = jls_gpu.JLSencodeBatch([jls_input(e["input"]) for e in big], 0)
    for g, e in zip(got, big):
        _check(g, e)
    e2 = [e for e in LARGE if e["near"] == 2][0]
    _check(jls_gpu.JLSencode(jls_input(e2["input"]), 2), e2)


@pytest.mark.gpu
def test_gpu_both_lossless_paths_agree(jls_gpu, monkeypatch):
    """Lossless planes in small batches are spread over the device (jls_par.h); IMCVT_JLS_PAR=0 forces the walker path:
    the same bytes, equal to the reference's digests (1080p and 4K) and to the CPU checker on run-heavy inputs."""
    import torch
    from oracle import oracle, synth
    big = [e for e in LARGE if e["near"] == 0]
    rng = np.random.default_rng(11)
    extra = [synth.flat(300, 200, 7), (rng.integers(0, 2, (97, 131)) * 255).astype(np.uint8), np.repeat(rng.integers(0, 256, (64, 1)), 500, axis=1).astype(np.uint8),
             np.clip(rng.normal(128, 1.2, (150, 333)), 0, 255).astype(np.uint8), synth.noise(1, 77, 3), synth.noise(77, 1, 3)]
    imgs = [jls_input(e["input"]) for e in big] + extra
    d = jls_gpu.DevicePlanes([torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in imgs], 0)
    d.encode(); par = d.results()
    assert d.last_path() == 1
    monkeypatch.setenv("IMCVT_JLS_PAR", "0")
    d.encode(); walk = d.results()
    assert d.last_path() == 0
    for g, e in zip(par, big):
        _check(g, e)
    for a, g, wv in zip(imgs, par, walk):
        assert g == wv, a.shape
    for a, g in zip(extra, par[len(big):]):
        assert g == oracle.jls_cpu_encode(a, 0), a.shape


@pytest.mark.gpu
def test_gpu_seeded_random_vs_cpu_checker(jls_gpu):
    from oracle import oracle
    rng = np.random.default_rng(2024)
    for i in range(30):
        h, w = int(rng.integers(1, 150)), int(rng.integers(1, 150))
        shape = (h, w, 3) if i % 4 == 3 else (h, w)
        img = (rng.integers(0, 256, shape) if i % 3 == 0 else np.clip(rng.normal(100, 4, shape), 0, 255) if i % 3 == 1
               else rng.integers(0, 2, shape) * 255).astype(np.uint8)
        near = int(rng.integers(0, 5))
        assert jls_gpu.JLSencode(img, near) == oracle.jls_cpu_encode(img, near), (i, shape, near)


@pytest.mark.gpu
def test_gpu_batch_of_ragged_planes_and_file_writer(jls_gpu, tmp_path):
    from oracle import oracle, synth
    imgs = [synth.syn(40 + 13 * i, 30 + 7 * i, i) for i in range(20)] + [synth.flat(5, 3, 9), synth.noise(1, 200, 3), synth.noise(300, 1, 4)]
    for near in (0, 3):
        got = jls_gpu.JLSencodeBatch(imgs, near)
        for g, im in zip(got, imgs):
            assert g == oracle.jls_cpu_encode(im, near)
    rgb = np.stack([imgs[3], imgs[3][::-1].copy(), 255 - imgs[3]], axis=-1)
    assert jls_gpu.writeJLSImageFile(str(tmp_path / "o.jls"), rgb, 1) == 0
    assert (tmp_path / "o.jls").read_bytes() == oracle.jls_cpu_encode(rgb, 1)
    assert jls_gpu.writeJLSImageFile(str(tmp_path / "nodir" / "o.jls"), rgb, 1) == 1


@pytest.mark.gpu
def test_cli_writes_reference_jls(built, tmp_path):
    from oracle import oracle, synth
    img = synth.syn(100, 70, 3)
    (tmp_path / "a.pgm").write_bytes(b"P5\n100 70\n255\n" + img.tobytes())
    exe = os.path.join(ROOT, "imcvt_amd", "csrc", "imcvt")
    r = subprocess.run([exe, "-2", "a.pgm", "-o", "a.jls"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout == "(1/1)  a.pgm -> a.jls\n"
    assert (tmp_path / "a.jls").read_bytes() == oracle.jls_cpu_encode(img, 2)


@pytest.mark.gpu
def test_gpu_stuffing_at_chunk_edges(jls_gpu):
    """The chunk-edge pictures (scan ends 1..8 bits past a 16 Kbit stuffing chunk) through the plane-parallel kernels on the device."""
    from oracle import oracle
    imgs = [_edge_img(seed, h, w) for seed, h, w, _ in CHUNK_EDGE_CASES]
    got = jls_gpu.JLSencodeBatch(imgs, 0)
    assert jls_gpu.load_jls_library().imcvt_jls_last_path() == 1
    for img, g in zip(imgs, got):
        assert g == oracle.jls_cpu_encode(img, 0)
