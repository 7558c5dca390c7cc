"""Host-side mirror of the reference's H.265 interface over libimcvt_hevc.so (C ABI: include/imcvt_hevc.h).

Names, argument meaning and error behaviour follow the reference:

  HEVCImageEncoder(img, qpd6)            <- src/HEVCe/HEVCe.h:5-12  (returns stream, reconstruction, padded size)
  writeHEVCImageFile(path, buf, ...)     <- src/imageio.h:21 / src/imageio_hevc.c:9 (0 = ok, 1 = failed)
  HEVCImageEncoderBatch(imgs, qpd6)      <- new: the batch seam at src/main.c:162

The compute path is the HIP library only.  If it is missing, or no gfx950 device is visible, these
functions raise — there is no CPU fallback and nothing here touches oracle/.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IMCVT_HEVC_LIB") or os.path.join(_HERE, "csrc", "libimcvt_hevc.so")   # override: dev/profiling builds only

_u8p = C.POINTER(C.c_ubyte)
_ip = C.POINTER(C.c_int)

ERRORS = {-1: "no HIP device visible (no CPU fallback)", -2: "HIP runtime error", -3: "bad argument", -4: "device watchdog: a wait between workgroups gave up, results invalid"}

# every symbol include/imcvt_hevc.h declares
EXPORTS = ("HEVCImageEncoder", "writeHEVCImageFile", "HEVCImageEncoderBatch", "imcvt_hevc_create", "imcvt_hevc_destroy",
           "imcvt_hevc_stream_bound", "imcvt_hevc_padded", "imcvt_hevc_encode_device", "imcvt_hevc_last_kernel_ms",
           "imcvt_hevc_set_trace", "imcvt_hevc_debug_prof", "imcvt_hevc_debug_occupancy", "imcvt_hevc_version",
           "imcvt_hevc_set_team", "imcvt_hevc_set_pipe", "imcvt_hevc_last_pipe", "imcvt_hevc_set_wide", "imcvt_hevc_last_wide", "imcvt_hevc_plan_wide", "imcvt_hevc_plan_wide_pool", "imcvt_hevc_last_team", "imcvt_hevc_last_shape", "imcvt_hevc_set_shape", "imcvt_hevc_set_pool_tuning", "imcvt_hevc_set_pool_split", "imcvt_hevc_batch_devices", "imcvt_hevc_shutdown", "imcvt_hevc_debug_census", "imcvt_hevc_last_resident", "imcvt_hevc_last_status", "imcvt_hevc_set_frame_clock", "imcvt_hevc_last_start_spread_us", "imcvt_hevc_plan", "imcvt_hevc_plan_pipe",
           "imcvt_hevc_residency", "imcvt_hevc_debug_filler", "imcvt_hevc_debug_set_backend", "imcvt_hevc_coalesce_stats",
           "imcvt_hevc_set_progress", "imcvt_hevc_batch_transfer_stats", "imcvt_hevc_batch_kernel_ms", "imcvt_hevc_set_split", "imcvt_hevc_last_split", "imcvt_hevc_plan_split",
           "imcvt_hevc_set_partners", "imcvt_hevc_last_partners", "imcvt_hevc_plan_partners")


class imcvt_hevc_frame(C.Structure):
    _fields_ = [("d_img", C.c_void_p), ("d_out", C.c_void_p), ("d_rcon", C.c_void_p), ("d_len", C.c_void_p),
                ("h", C.c_int), ("w", C.c_int), ("qpd6", C.c_int)]


_lib = None


def load_library():
    """Load libimcvt_hevc.so (built by __graft_entry__.build() / imcvt_amd.build).  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    try:                     # one HIP runtime per process: if torch is around, let it load its runtime first
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(the HIP extension is mandatory; there is no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    lib.HEVCImageEncoder.restype = C.c_int
    lib.HEVCImageEncoder.argtypes = [_u8p, _u8p, _u8p, _ip, _ip, C.c_int]
    lib.writeHEVCImageFile.restype = C.c_int
    lib.writeHEVCImageFile.argtypes = [C.c_char_p, _u8p, C.c_int, C.c_uint32, C.c_uint32, C.c_int]
    lib.HEVCImageEncoderBatch.restype = C.c_int
    lib.HEVCImageEncoderBatch.argtypes = [C.c_int, C.POINTER(_u8p), C.POINTER(_u8p), C.POINTER(_u8p), _ip, _ip, _ip, _ip]
    lib.imcvt_hevc_create.restype = C.c_void_p
    lib.imcvt_hevc_create.argtypes = [C.c_int]
    lib.imcvt_hevc_destroy.restype = None
    lib.imcvt_hevc_destroy.argtypes = [C.c_void_p]
    lib.imcvt_hevc_stream_bound.restype = C.c_longlong
    lib.imcvt_hevc_stream_bound.argtypes = [C.c_int, C.c_int]
    lib.imcvt_hevc_padded.restype = C.c_int
    lib.imcvt_hevc_padded.argtypes = [C.c_int]
    lib.imcvt_hevc_encode_device.restype = C.c_int
    lib.imcvt_hevc_encode_device.argtypes = [C.c_void_p, C.c_int, C.POINTER(imcvt_hevc_frame), C.c_void_p]
    lib.imcvt_hevc_last_kernel_ms.restype = C.c_float
    lib.imcvt_hevc_last_kernel_ms.argtypes = [C.c_void_p]
    lib.imcvt_hevc_set_trace.restype = None
    lib.imcvt_hevc_set_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lib.imcvt_hevc_debug_prof.restype = C.c_int
    lib.imcvt_hevc_debug_prof.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int, C.c_int]
    lib.imcvt_hevc_debug_occupancy.restype = C.c_int
    lib.imcvt_hevc_debug_occupancy.argtypes = [_ip, _ip, _ip, _ip]
    lib.imcvt_hevc_version.restype = C.c_char_p
    lib.imcvt_hevc_version.argtypes = []
    lib.imcvt_hevc_set_team.restype = None
    lib.imcvt_hevc_set_team.argtypes = [C.c_void_p, C.c_int]
    lib.imcvt_hevc_plan_pipe.restype = C.c_int
    lib.imcvt_hevc_plan_pipe.argtypes = [C.c_int, C.c_int, C.c_int, _ip, _ip]
    lib.imcvt_hevc_set_pipe.restype = None
    lib.imcvt_hevc_set_pipe.argtypes = [C.c_void_p, C.c_int]
    lib.imcvt_hevc_last_pipe.restype = C.c_int
    lib.imcvt_hevc_last_pipe.argtypes = [C.c_void_p]
    if hasattr(lib, "imcvt_hevc_set_wide"):             # (absent only from older builds loaded through IMCVT_HEVC_LIB for A/B runs)
        lib.imcvt_hevc_set_wide.restype = None
        lib.imcvt_hevc_set_wide.argtypes = [C.c_void_p, C.c_int]
        lib.imcvt_hevc_last_wide.restype = C.c_int
        lib.imcvt_hevc_last_wide.argtypes = [C.c_void_p]
        lib.imcvt_hevc_plan_wide.restype = C.c_int
        lib.imcvt_hevc_plan_wide.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
        lib.imcvt_hevc_plan_wide_pool.restype = C.c_int
        lib.imcvt_hevc_plan_wide_pool.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _ip, _ip]
    lib.imcvt_hevc_last_team.restype = C.c_int
    lib.imcvt_hevc_last_team.argtypes = [C.c_void_p, _ip]
    lib.imcvt_hevc_batch_devices.restype = C.c_int
    lib.imcvt_hevc_batch_devices.argtypes = []
    lib.imcvt_hevc_debug_census.restype = C.c_int
    lib.imcvt_hevc_debug_census.argtypes = [C.c_void_p, C.c_int]
    lib.imcvt_hevc_last_resident.restype = C.c_int
    lib.imcvt_hevc_last_resident.argtypes = [C.c_void_p]
    lib.imcvt_hevc_last_status.restype = C.c_int
    lib.imcvt_hevc_last_status.argtypes = [C.c_void_p]
    lib.imcvt_hevc_set_frame_clock.restype = None
    lib.imcvt_hevc_set_frame_clock.argtypes = [C.c_void_p, C.c_void_p]
    lib.imcvt_hevc_last_start_spread_us.restype = C.c_longlong
    lib.imcvt_hevc_last_start_spread_us.argtypes = [C.c_void_p]
    lib.imcvt_hevc_plan.restype = C.c_int
    lib.imcvt_hevc_plan.argtypes = [C.c_int, C.c_int, C.c_int, _ip, _ip]
    lib.imcvt_hevc_last_shape.restype = C.c_int
    lib.imcvt_hevc_last_shape.argtypes = [C.c_void_p, _ip, _ip]
    lib.imcvt_hevc_set_shape.restype = None
    lib.imcvt_hevc_set_shape.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.imcvt_hevc_set_pool_tuning.restype = None
    lib.imcvt_hevc_set_pool_tuning.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    lib.imcvt_hevc_set_pool_split.restype = None
    lib.imcvt_hevc_set_pool_split.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.imcvt_hevc_shutdown.restype = None
    lib.imcvt_hevc_shutdown.argtypes = []
    if hasattr(lib, "imcvt_hevc_residency"):             # (absent only from older builds loaded through IMCVT_HEVC_LIB for A/B runs)
        lib.imcvt_hevc_residency.restype = C.c_int
        lib.imcvt_hevc_residency.argtypes = [C.c_void_p, _ip, _ip, _ip, _ip, _ip, _ip]
        lib.imcvt_hevc_debug_filler.restype = C.c_int
        lib.imcvt_hevc_debug_filler.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p]
        lib.imcvt_hevc_debug_set_backend.restype = None
        lib.imcvt_hevc_debug_set_backend.argtypes = [C.c_void_p]
        lib.imcvt_hevc_coalesce_stats.restype = None
        lib.imcvt_hevc_coalesce_stats.argtypes = [C.POINTER(C.c_long), C.POINTER(C.c_long), C.POINTER(C.c_long), C.c_int]
    if hasattr(lib, "imcvt_hevc_set_progress"):          # (round 6 on)
        lib.imcvt_hevc_set_progress.restype = None
        lib.imcvt_hevc_set_progress.argtypes = [C.c_void_p, C.c_void_p]
        lib.imcvt_hevc_batch_transfer_stats.restype = None
        lib.imcvt_hevc_batch_transfer_stats.argtypes = [C.POINTER(C.c_double)] * 5
        lib.imcvt_hevc_batch_kernel_ms.restype = C.c_double
        lib.imcvt_hevc_batch_kernel_ms.argtypes = []
        lib.imcvt_hevc_set_split.restype = None
        lib.imcvt_hevc_set_split.argtypes = [C.c_void_p, C.c_int, C.c_int]
        lib.imcvt_hevc_last_split.restype = C.c_int
        lib.imcvt_hevc_last_split.argtypes = [C.c_void_p]
        lib.imcvt_hevc_set_partners.restype = None
        lib.imcvt_hevc_set_partners.argtypes = [C.c_void_p, C.c_int]
        lib.imcvt_hevc_last_partners.restype = C.c_int
        lib.imcvt_hevc_last_partners.argtypes = [C.c_void_p]
        lib.imcvt_hevc_plan_partners.restype = C.c_int
        lib.imcvt_hevc_plan_partners.argtypes = [C.c_int, _ip, C.c_int, C.c_int]
        lib.imcvt_hevc_plan_split.restype = C.c_int
        lib.imcvt_hevc_plan_split.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _ip]
    _lib = lib
    return lib


def coalesce_stats(reset=False):
    """(calls submitted, device batches run, frames in the largest batch) of the host-pointer entry points' submission queue."""
    lib = load_library()
    a, b, c = C.c_long(0), C.c_long(0), C.c_long(0)
    lib.imcvt_hevc_coalesce_stats(C.byref(a), C.byref(b), C.byref(c), int(reset))
    return a.value, b.value, c.value


def _check(rc, what):
    if rc < 0:
        raise RuntimeError(f"{what} failed: {ERRORS.get(rc, rc)}")
    return rc


def padded(v: int) -> int:
    return (min(v, 8192) + 31) // 32 * 32


def stream_bound(h: int, w: int) -> int:
    return 2 * (w + 32) * (h + 32) + 65536


def HEVCImageEncoder(img: np.ndarray, qpd6: int = 0):
    """Encode one gray8 frame.  Returns (stream bytes, reconstruction[yszn, xszn] uint8, (yszn, xszn))."""
    lib = load_library()
    assert img.dtype == np.uint8 and img.ndim == 2
    img = np.ascontiguousarray(img)
    h, w = img.shape
    hp, wp = padded(h), padded(w)
    out = np.empty(stream_bound(h, w), dtype=np.uint8)
    rcon = np.empty(hp * wp, dtype=np.uint8)
    ys, xs = C.c_int(h), C.c_int(w)
    n = _check(lib.HEVCImageEncoder(out.ctypes.data_as(_u8p), img.ctypes.data_as(_u8p), rcon.ctypes.data_as(_u8p),
                                    C.byref(ys), C.byref(xs), int(qpd6)), "HEVCImageEncoder")
    return out[:n].tobytes(), rcon.reshape(ys.value, xs.value), (ys.value, xs.value)


def transfer_stats():
    """How the last host-pointer batch moved its data: dict(upload_s, follow_s, tail_s, bytes_during, bytes_after) — see include/imcvt_hevc.h."""
    lib = load_library()
    v = [C.c_double(0) for _ in range(5)]
    lib.imcvt_hevc_batch_transfer_stats(*[C.byref(x) for x in v])
    return dict(zip(("upload_s", "follow_s", "tail_s", "bytes_during", "bytes_after"), (x.value for x in v)), kernel_ms=float(lib.imcvt_hevc_batch_kernel_ms()))


def HEVCImageEncoderBatch(imgs, qpd6=0, copy=True):
    """Encode independent gray8 frames concurrently (one workgroup per frame).  qpd6: int or per-frame list.
    Returns a list of (stream bytes, reconstruction, (yszn, xszn)); with copy=False the streams are uint8 views of the buffers the
    library wrote into (as a C caller sees them: it owns every buffer, src/imageio_hevc.c:14-16) instead of bytes objects."""
    lib = load_library()
    n = len(imgs)
    if n == 0:
        return []
    qs = [int(qpd6)] * n if np.isscalar(qpd6) else [int(q) for q in qpd6]
    imgs = [np.ascontiguousarray(a) for a in imgs]
    for a in imgs:
        assert a.dtype == np.uint8 and a.ndim == 2
    outs = [np.empty(stream_bound(*a.shape), dtype=np.uint8) for a in imgs]
    rcons = [np.empty(padded(a.shape[0]) * padded(a.shape[1]), dtype=np.uint8) for a in imgs]
    P = _u8p * n
    ys = (C.c_int * n)(*[a.shape[0] for a in imgs])
    xs = (C.c_int * n)(*[a.shape[1] for a in imgs])
    qv = (C.c_int * n)(*qs)
    lens = (C.c_int * n)()
    _check(lib.HEVCImageEncoderBatch(n, P(*[o.ctypes.data_as(_u8p) for o in outs]), P(*[a.ctypes.data_as(_u8p) for a in imgs]),
                                     P(*[r.ctypes.data_as(_u8p) for r in rcons]), ys, xs, qv, lens), "HEVCImageEncoderBatch")
    return [(outs[i][:lens[i]].tobytes() if copy else outs[i][:lens[i]], rcons[i].reshape(ys[i], xs[i]), (ys[i], xs[i])) for i in range(n)]


def writeHEVCImageFile(filename: str, buf: np.ndarray, is_rgb: bool, height: int, width: int, qpd6: int = 0) -> int:
    """Reference semantics: 0 = success, 1 = failed (src/imageio_hevc.c:9-53)."""
    lib = load_library()
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    return lib.writeHEVCImageFile(filename.encode(), buf.ctypes.data_as(_u8p), int(bool(is_rgb)), height, width, int(qpd6))


class DeviceEncoder:
    """Device-resident batch encoder: inputs/outputs stay in HBM (torch tensors supply the memory)."""

    def __init__(self, max_workgroups: int = 0):
        self.lib = load_library()
        self.ctx = self.lib.imcvt_hevc_create(int(max_workgroups))
        if not self.ctx:
            raise RuntimeError("imcvt_hevc_create failed: " + ERRORS[-1])

    def close(self):
        if self.ctx:
            self.lib.imcvt_hevc_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_team(self, team_size: int):
        """Helper workgroups: 0 = chosen per launch, 1 = none (a frame per workgroup), 2 / 3 = one / two per main workgroup (same results)."""
        self.lib.imcvt_hevc_set_team(self.ctx, int(team_size))

    def set_pipe(self, mode: int):
        """Pipe wave (256-thread workgroups, the NxN trial of the 8x8 CUs off the PU chain): < 0 / 1 = whenever the launch fits three workgroups per CU, 0 = never (same results)."""
        self.lib.imcvt_hevc_set_pipe(self.ctx, int(mode))

    def set_wide(self, mode: int):
        """Wide workgroups (512 threads, split trial coders): -1 automatic, 0 never, 1 wherever they fit.  Results are identical."""
        if hasattr(self.lib, "imcvt_hevc_set_wide"):
            self.lib.imcvt_hevc_set_wide(self.ctx, int(mode))

    def set_partners(self, mode: int):
        """Partner workgroups (wide pools: the 2Nx2N sets of a main workgroup's 8x8 CUs on a second compute unit): -1 / 1 wherever they fit, 0 never."""
        self.lib.imcvt_hevc_set_partners(self.ctx, int(mode))

    def last_partners(self) -> int:
        """Partner workgroups of the last launch."""
        return int(self.lib.imcvt_hevc_last_partners(self.ctx))

    def set_split(self, mode: int, helpers_per_cu: int = 0):
        """A pool as two cooperating launches (wide main workgroups + 192-thread helpers on disjoint compute units): 0 never, 1 / -1 where planned."""
        self.lib.imcvt_hevc_set_split(self.ctx, int(mode), int(helpers_per_cu))

    def last_split(self) -> bool:
        """True if the last launch was such a pair of launches."""
        return int(self.lib.imcvt_hevc_last_split(self.ctx)) == 1

    def last_wide(self) -> bool:
        """True if the last launch ran wide workgroups."""
        return hasattr(self.lib, "imcvt_hevc_last_wide") and int(self.lib.imcvt_hevc_last_wide(self.ctx)) == 1

    def last_pipe(self) -> bool:
        """True if the last launch ran with the pipe wave."""
        return int(self.lib.imcvt_hevc_last_pipe(self.ctx)) == 1

    def last_team(self):
        """(1 / 2 / 3 = no / fewer than two / two helpers per main workgroup, main workgroups of a launch with helpers) of the last launch."""
        nt = C.c_int(0)
        return int(self.lib.imcvt_hevc_last_team(self.ctx, C.byref(nt))), nt.value

    def set_shape(self, nmains: int, nhelp: int):
        """Debug / tuning: exactly this many main and helper workgroups for the next launches; (0, 0) = automatic again."""
        self.lib.imcvt_hevc_set_shape(self.ctx, int(nmains), int(nhelp))

    def set_pool_tuning(self, lim16: int = -1, lim32: int = -1, prio: int = -1):
        """Debug / tuning: posting limits per queue shard and main-workgroup priority of launches with helpers (< 0: defaults)."""
        self.lib.imcvt_hevc_set_pool_tuning(self.ctx, int(lim16), int(lim32), int(prio))

    def set_pool_split(self, post16: int = -1, post32: int = -1):
        """Debug / tuning: per mille of the 16x16 / 32x32 CUs offered to the helpers (< 0: from the launch shape)."""
        self.lib.imcvt_hevc_set_pool_split(self.ctx, int(post16), int(post32))

    def residency(self):
        """What the context plans its launches against: dict(cus, max_wg, pipe_wg, occ_per_cu, occ_pipe_per_cu, census_wg, census_pipe)."""
        v = [C.c_int(0) for _ in range(6)]
        cus = int(self.lib.imcvt_hevc_residency(self.ctx, *[C.byref(x) for x in v]))
        return dict(zip(("max_wg", "pipe_wg", "occ_per_cu", "occ_pipe_per_cu", "census_wg", "census_pipe"), (x.value for x in v)), cus=cus)

    def last_resident(self):
        """Most workgroups of the last launch that ran at the same time."""
        return int(self.lib.imcvt_hevc_last_resident(self.ctx))

    def last_start_spread_us(self):
        """Microseconds between the start of the first and of the last workgroup of the last launch."""
        return int(self.lib.imcvt_hevc_last_start_spread_us(self.ctx))

    def last_shape(self):
        """(main workgroups, helper workgroups) of the last launch."""
        a, b = C.c_int(0), C.c_int(0)
        self.lib.imcvt_hevc_last_shape(self.ctx, C.byref(a), C.byref(b))
        return a.value, b.value

    def make_batch(self, imgs_dev, qpd6=0):
        """imgs_dev: list of 2-D uint8 CUDA tensors.  Allocates outputs (torch) and the descriptor array."""
        import torch
        n = len(imgs_dev)
        qs = [int(qpd6)] * n if np.isscalar(qpd6) else [int(q) for q in qpd6]
        dev = imgs_dev[0].device
        outs, rcons = [], []
        lens = torch.zeros(n, dtype=torch.int32, device=dev)
        arr = (imcvt_hevc_frame * n)()
        for i, t in enumerate(imgs_dev):
            assert t.dtype == torch.uint8 and t.dim() == 2 and t.is_contiguous() and t.is_cuda
            h, w = t.shape
            o = torch.empty(stream_bound(h, w), dtype=torch.uint8, device=dev)
            r = torch.empty((padded(h), padded(w)), dtype=torch.uint8, device=dev)
            outs.append(o)
            rcons.append(r)
            arr[i] = imcvt_hevc_frame(t.data_ptr(), o.data_ptr(), r.data_ptr(), lens.data_ptr() + 4 * i, h, w, qs[i])
        return dict(n=n, frames=arr, imgs=imgs_dev, outs=outs, rcons=rcons, lens=lens)

    def encode(self, batch, stream=None):
        """Asynchronous launch on `stream` (a torch.cuda.Stream or None for the current stream)."""
        import torch
        s = stream if stream is not None else torch.cuda.current_stream()
        _check(self.lib.imcvt_hevc_encode_device(self.ctx, batch["n"], batch["frames"], C.c_void_p(s.cuda_stream)),
               "imcvt_hevc_encode_device")

    PROF_CATS = ("border", "p1_32", "p1_16", "p1_8", "p1_4", "p2_32", "p2_16", "p2_8", "p2_pu", "p2_nxn", "sync", "wait_help", "recon", "idle", "t_setup", "t_hdr", "passA", "passB", "passC", "n_cg", "x1", "x2", "x3")

    def debug_prof(self, reset=True):
        """Per-wave cycle totals by phase (only non-zero for -DIMCVT_PROF builds): rows = 3 waves of the main role (or of
        frame-per-workgroup launches), then 3 waves of each helper role."""
        buf = (C.c_ulonglong * 512)()
        n = _check(self.lib.imcvt_hevc_debug_prof(self.ctx, buf, 512, int(reset)), "imcvt_hevc_debug_prof")
        k = len(self.PROF_CATS)
        self._regions = [int(buf[9 * k + i]) for i in range(max(0, min(n, 512) - 9 * k))]      # (-DIMCVT_REGCNT builds: executions of the marked regions, tools/valu_dyn_mix.py)
        return [[int(buf[w * k + i]) for i in range(k)] for w in range(min(n // k, 9))]

    def debug_regions(self, reset=True):
        """Region execution counters of a -DIMCVT_REGCNT build (hevc_core.h RCNT), summed over the launches since the last reset."""
        self.debug_prof(reset)
        return self._regions

    def last_kernel_ms(self) -> float:
        ms = float(self.lib.imcvt_hevc_last_kernel_ms(self.ctx))
        self.status()
        return ms

    def status(self):
        """Waits for the last launch; raises if its watchdog fired."""
        _check(self.lib.imcvt_hevc_last_status(self.ctx), "imcvt_hevc_last_status")

    def results(self, batch):
        import torch
        torch.cuda.synchronize()
        self.status()
        lens = batch["lens"].cpu().tolist()
        return [(batch["outs"][i][:lens[i]].cpu().numpy().tobytes(), batch["rcons"][i].cpu().numpy()) for i in range(batch["n"])]
