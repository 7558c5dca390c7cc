"""Integer-only synthetic gray8 inputs: workload generation for bench.py, the tools and the tests.

syn(w,h,seed)   gradient + 16x16 block texture + 4-bit xorshift32 noise
noise(w,h,seed) uniform 8-bit xorshift32 noise
flat(w,h,v)     constant

The xorshift stream is advanced once per pixel in raster order, so a frame is reproducible from
(w, h, seed) alone in C and in Python.  numpy is used for speed; `syn_py` is the literal definition and
is kept for the unit test that pins the vectorised form to it.
"""
import numpy as np

M32 = 0xFFFFFFFF


def seed0(seed: int) -> int:
    s = (seed * 2654435761 + 0x9E3779B9) & M32
    return s if s else 1


def _xs(s: int) -> int:
    s ^= (s << 13) & M32
    s ^= s >> 17
    s ^= (s << 5) & M32
    return s


_JUMP = {}       # log2(steps) -> the `steps`-step linear map of xorshift32 as 32 basis images


def _apply_map(basis: np.ndarray, src: np.ndarray) -> np.ndarray:
    acc = np.zeros(src.shape, dtype=np.uint32)
    for b in range(32):
        acc ^= basis[b] * ((src >> np.uint32(b)) & np.uint32(1))
    return acc


def _jump_map(k: int) -> np.ndarray:
    """Basis images of A^(2^k) (A = one xorshift32 step, linear over GF(2)), by repeated squaring."""
    if k not in _JUMP:
        if k == 0:
            _JUMP[0] = np.array([_xs(1 << b) for b in range(32)], dtype=np.uint32)
        else:
            prev = _jump_map(k - 1)
            _JUMP[k] = _apply_map(prev, prev)
    return _JUMP[k]


def _xorshift_stream(seed: int, n: int) -> np.ndarray:
    """n successive xorshift32 states after seed0(seed) (state AFTER each step).
    state_k = A^k state_0, so the second half of a 2^(j+1)-state prefix is the 2^j-step map applied to the first half:
    the stream is built by doubling, 32 vector passes per doubling."""
    out = np.empty(max(n, 1), dtype=np.uint32)
    out[0] = _xs(seed0(seed))
    have, k = 1, 0
    while have < n:
        m = min(have, n - have)
        out[have:have + m] = _apply_map(_jump_map(k), out[:m])
        have += m
        k += 1
    return out[:n]


def syn(w: int, h: int, seed: int) -> np.ndarray:
    s = _xorshift_stream(seed, w * h).reshape(h, w)
    y = np.arange(h, dtype=np.int32)[:, None]
    x = np.arange(w, dtype=np.int32)[None, :]
    g = ((3 * x + 5 * y) >> 4) & 0xFF
    t = (((x >> 4) ^ (y >> 4)) * 37) & 0x3F
    v = (g >> 1) + t + 32 + (s & np.uint32(0x0F)).astype(np.int32)
    return np.minimum(255, v).astype(np.uint8)


def noise(w: int, h: int, seed: int) -> np.ndarray:
    s = _xorshift_stream(seed, w * h).reshape(h, w)
    return (s & 0xFF).astype(np.uint8)


def flat(w: int, h: int, v: int) -> np.ndarray:
    return np.full((h, w), v, dtype=np.uint8)


def syn_py(w: int, h: int, seed: int) -> np.ndarray:
    """Literal (slow) definition of syn(), used to pin the vectorised version."""
    out = np.empty((h, w), dtype=np.uint8)
    s = seed0(seed)
    for y in range(h):
        for x in range(w):
            s = _xs(s)
            g = ((3 * x + 5 * y) >> 4) & 0xFF
            t = (((x >> 4) ^ (y >> 4)) * 37) & 0x3F
            n = s & 0x0F
            out[y, x] = min(255, (g >> 1) + t + n + 32)
    return out


def pgm_bytes(img: np.ndarray) -> bytes:
    """P5 file image; single '\\n' after maxval (reference loader quirk, src/imageio_pnm.c:96-98)."""
    h, w = img.shape
    return b"P5\n%d %d\n255\n" % (w, h) + img.tobytes()
