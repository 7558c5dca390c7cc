"""Host mirror of the reference's JPEG-LS interface (src/imageio.h:20, src/imageio_jls.c:428) over libimcvt_jls.so.

    writeJLSImageFile(path, img, near)        -> 0 / 1, the reference's file writer
    JLSencode(img, near)                      -> bytes, the same stream in memory
    JLSencodeBatch(list of gray planes, near) -> list of bytes, all planes concurrently (device-resident): up to 64 lossless
                                                 planes are each spread over the device (context chains, csrc/jls_par.h),
                                                 otherwise one wavefront walks each plane

img: numpy uint8 [h, w] (gray) or [h, w, 3] (RGB).  No CPU fallback: the library needs a gfx950 device."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_u8p = C.POINTER(C.c_ubyte)
_lib = None


class _Plane(C.Structure):
    _fields_ = [("d_img", C.c_void_p), ("d_out", C.c_void_p), ("d_len", C.c_void_p), ("h", C.c_int), ("w", C.c_int), ("near", C.c_int)]


def load_jls_library():
    global _lib
    if _lib is None:
        path = os.environ.get("IMCVT_JLS_LIB", os.path.join(_HERE, "csrc", "libimcvt_jls.so"))
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` (there is no CPU fallback)")
        lib = C.CDLL(path)
        lib.imcvt_jls_encode.restype = C.c_longlong
        lib.imcvt_jls_encode.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, _u8p]
        lib.imcvt_jls_stream_bound.restype = C.c_longlong
        lib.imcvt_jls_stream_bound.argtypes = [C.c_int, C.c_int]
        lib.writeJLSImageFile.restype = C.c_int
        lib.writeJLSImageFile.argtypes = [C.c_char_p, _u8p, C.c_int, C.c_uint32, C.c_uint32, C.c_int]
        lib.imcvt_jls_encode_device.restype = C.c_int
        lib.imcvt_jls_encode_device.argtypes = [C.c_int, C.POINTER(_Plane), C.c_void_p]
        lib.imcvt_jls_last_kernel_ms.restype = C.c_float
        lib.imcvt_jls_version.restype = C.c_char_p
        lib.imcvt_jls_last_path.restype = C.c_int
        _lib = lib
    return _lib


def _check(img):
    img = np.ascontiguousarray(img)
    if img.dtype != np.uint8 or img.ndim not in (2, 3) or (img.ndim == 3 and img.shape[2] != 3):
        raise ValueError("img must be uint8 [h, w] or [h, w, 3]")
    return img


def JLSencode(img, near=0) -> bytes:
    lib = load_jls_library()
    img = _check(img)
    h, w = img.shape[:2]
    out = np.empty(int(lib.imcvt_jls_stream_bound(h, w)) * (3 if img.ndim == 3 else 1), np.uint8)
    n = lib.imcvt_jls_encode(img.ctypes.data_as(_u8p), int(img.ndim == 3), h, w, int(near), out.ctypes.data_as(_u8p))
    if n <= 0:
        raise RuntimeError(f"imcvt_jls_encode failed ({n})")
    return out[:n].tobytes()


def writeJLSImageFile(path, img, near=0) -> int:
    lib = load_jls_library()
    img = _check(img)
    h, w = img.shape[:2]
    return lib.writeJLSImageFile(os.fsencode(path), img.ctypes.data_as(_u8p), int(img.ndim == 3), h, w, int(near))


class DevicePlanes:
    """Device-resident batch of gray planes (torch tensors on the GPU): what tools/jls_bench.py times."""

    def __init__(self, planes_dev, near=0):
        import torch
        self.lib = load_jls_library()
        self.planes = [p.contiguous() for p in planes_dev]
        self.near = int(near)
        self.n = len(self.planes)
        bound = [int(self.lib.imcvt_jls_stream_bound(p.shape[0], p.shape[1])) for p in self.planes]
        self.outs = [torch.empty(b, dtype=torch.uint8, device=p.device) for b, p in zip(bound, self.planes)]
        self.lens = torch.zeros(self.n, dtype=torch.int64, device=self.planes[0].device)
        self.desc = (_Plane * self.n)()
        for i, p in enumerate(self.planes):
            self.desc[i] = _Plane(p.data_ptr(), self.outs[i].data_ptr(), self.lens.data_ptr() + 8 * i, p.shape[0], p.shape[1], self.near)

    def encode(self):
        import torch
        rc = self.lib.imcvt_jls_encode_device(self.n, self.desc, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        if rc != 0:
            raise RuntimeError(f"imcvt_jls_encode_device failed ({rc})")

    def last_kernel_ms(self):
        return float(self.lib.imcvt_jls_last_kernel_ms())

    def last_path(self):
        """1: planes spread over the device (jls_par.h), 0: one walker per plane."""
        return int(self.lib.imcvt_jls_last_path())

    def results(self):
        import torch
        torch.cuda.synchronize()
        lens = self.lens.cpu().numpy()
        return [self.outs[i][:int(lens[i])].cpu().numpy().tobytes() for i in range(self.n)]


def JLSencodeBatch(imgs, near=0):
    import torch
    d = DevicePlanes([torch.from_numpy(_check(i)).cuda() for i in imgs], near)
    d.encode()
    return d.results()
