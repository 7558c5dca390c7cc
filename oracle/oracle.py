"""ctypes front-ends for the CPU checkers.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
package (imcvt_amd) never does.

  port_*  : oracle/_build/liboracle_*.so  — our plain-C restatement (oracle/hevc_oracle.c, jls_oracle.c)
  ref_*   : oracle/_ref/libref_*.so       — the real reference compiled by oracle/Makefile from
                                            /root/reference (exists only where that build has run)
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_u8p = C.POINTER(C.c_ubyte)


def build(target: str = "all") -> None:
    subprocess.run(["make", "-s", "-C", HERE, target], check=True)


def _load(path):
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    return C.CDLL(path)


def _bind_encoder(lib, name="HEVCImageEncoder"):
    fn = getattr(lib, name)
    fn.restype = C.c_int
    fn.argtypes = [_u8p, _u8p, _u8p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
    return fn


_cache = {}


def port_lib():
    if "port" not in _cache:
        p = os.path.join(HERE, "_build", "liboracle_hevc.so")
        if not os.path.exists(p):
            build("port")
        _cache["port"] = _load(p)
    return _cache["port"]


def ref_lib():
    if "ref" not in _cache:
        _cache["ref"] = _load(os.path.join(HERE, "_ref", "libref_hevce.so"))
    return _cache["ref"]


def have_ref() -> bool:
    return os.path.exists(os.path.join(HERE, "_ref", "libref_hevce.so"))


def _encode(fn, img: np.ndarray, qpd6: int):
    """Call an encoder with the reference C signature (src/HEVCe/HEVCe.h:5-12).

    Returns (stream bytes, recon[yszn, xszn] uint8, (yszn, xszn))."""
    assert img.dtype == np.uint8 and img.ndim == 2
    img = np.ascontiguousarray(img)
    h, w = img.shape
    hp, wp = (min(h, 8192) + 31) // 32 * 32, (min(w, 8192) + 31) // 32 * 32
    out = np.zeros(2 * (w + 32) * (h + 32) + 65536, dtype=np.uint8)
    rcon = np.zeros(hp * wp, dtype=np.uint8)
    ys, xs = C.c_int(h), C.c_int(w)
    n = fn(out.ctypes.data_as(_u8p), img.ctypes.data_as(_u8p), rcon.ctypes.data_as(_u8p),
           C.byref(ys), C.byref(xs), int(qpd6))
    assert (ys.value, xs.value) == (hp, wp)
    return out[:n].tobytes(), rcon.reshape(hp, wp), (hp, wp)


def port_encode(img, qpd6=0):
    return _encode(_bind_encoder(port_lib(), "oracle_HEVCImageEncoder"), img, qpd6)


def ref_encode(img, qpd6=0):
    return _encode(_bind_encoder(ref_lib()), img, qpd6)


def cpu_encode(img, qpd6=0):
    """Checker used on the GPU box: the real reference when its prebuilt .so travelled, else the port."""
    return ref_encode(img, qpd6) if have_ref() else port_encode(img, qpd6)


# ---- JPEG-LS (BASELINE config 5; reference src/imageio_jls.c) ------------------------------------------------------
def jls_port_encode(img: np.ndarray, near: int = 0) -> bytes:
    """Our C restatement (oracle/jls_oracle.c), in memory.  img: [h, w] gray8 or [h, w, 3] RGB24."""
    p = os.path.join(HERE, "_build", "liboracle_jls.so")
    if not os.path.exists(p):
        build("port")
    lib = _cache.setdefault("jls_port", _load(p))
    lib.jls_oracle_encode.restype = C.c_longlong
    lib.jls_oracle_encode.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, _u8p]
    img = np.ascontiguousarray(img)
    h, w = img.shape[:2]
    out = np.zeros(8 * w * h + 65536, dtype=np.uint8)
    n = lib.jls_oracle_encode(img.ctypes.data_as(_u8p), int(img.ndim == 3), h, w, int(near), out.ctypes.data_as(_u8p))
    assert n > 0
    return out[:n].tobytes()


def jls_have_ref() -> bool:
    return os.path.exists(os.path.join(HERE, "_ref", "libref_jls.so"))


def jls_ref_encode(img: np.ndarray, near: int = 0) -> bytes:
    """The real reference (file based: writeJLSImageFile, src/imageio.h:20) through a temporary file."""
    import tempfile
    lib = _cache.setdefault("jls_ref", _load(os.path.join(HERE, "_ref", "libref_jls.so")))
    lib.writeJLSImageFile.restype = C.c_int
    lib.writeJLSImageFile.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_uint32, C.c_uint32, C.c_int]
    img = np.ascontiguousarray(img)
    h, w = img.shape[:2]
    with tempfile.NamedTemporaryFile(suffix=".jls", delete=False) as f:
        name = f.name
    try:
        assert lib.writeJLSImageFile(name.encode(), img.tobytes(), int(img.ndim == 3), h, w, int(near)) == 0
        return open(name, "rb").read()
    finally:
        os.unlink(name)


def jls_cpu_encode(img, near=0) -> bytes:
    return jls_ref_encode(img, near) if jls_have_ref() else jls_port_encode(img, near)
