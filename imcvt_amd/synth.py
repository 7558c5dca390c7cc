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


def _xorshift_stream(seed: int, n: int) -> np.ndarray:
    """n successive xorshift32 states after seed0(seed) (state AFTER each step)."""
    # xorshift32 is linear over GF(2): state_k = A^k state_0.  Vectorise with a jump table: generate the
    # first B states serially, then advance whole blocks with the B-step matrix applied via bit tricks.
    # For the sizes used here (<= 8.3 M pixels) a chunked serial generator in numpy-uint32 is enough:
    out = np.empty(n, dtype=np.uint32)
    s = seed0(seed)
    B = 4096
    # serial for first block
    first = min(B, n)
    for i in range(first):
        s = _xs(s)
        out[i] = s
    if n <= B:
        return out
    # Build the B-step linear map as 32 basis images, then apply it column-wise to previous block.
    basis = np.empty(32, dtype=np.uint32)
    for b in range(32):
        v = 1 << b
        for _ in range(B):
            v = _xs(v)
        basis[b] = v
    pos = B
    prev = out[:B]
    while pos < n:
        m = min(B, n - pos)
        src = prev[:m]
        acc = np.zeros(m, dtype=np.uint32)
        for b in range(32):
            mask = ((src >> np.uint32(b)) & np.uint32(1)).astype(bool)
            acc[mask] ^= basis[b]
        out[pos:pos + m] = acc
        prev = out[pos:pos + m] if m == B else prev
        pos += m
    return out


def syn(w: int, h: int, seed: int) -> np.ndarray:
    s = _xorshift_stream(seed, w * h).reshape(h, w)
    y, x = np.mgrid[0:h, 0:w]
    g = ((3 * x + 5 * y) >> 4) & 0xFF
    t = (((x >> 4) ^ (y >> 4)) * 37) & 0x3F
    n = (s & 0x0F).astype(np.int64)
    return np.minimum(255, (g >> 1) + t + n + 32).astype(np.uint8)


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
