import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


@pytest.fixture(scope="session")
def built():
    """Build everything once: oracle port (+ reference where /root/reference exists) and the HIP library."""
    import __graft_entry__ as g
    g.build()
    return True


def _hostemu_lib(name, flags):
    import ctypes as C
    d = os.path.join(ROOT, "tests", "hostemu")
    so = os.path.join(d, name)
    srcs = [os.path.join(d, "hostemu.cpp")] + [os.path.join(ROOT, "imcvt_amd", "csrc", f) for f in ("hevc_core.h", "hevc_frame.h", "hevc_tables.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        tmp = f"{so}.tmp.{os.getpid()}"            # parallel workers (pytest -n) build into private files and rename: nobody loads a half-written library
        subprocess.run(["g++", "-O2", "-fPIC", "-shared", *flags, "-o", tmp, srcs[0]], check=True)
        os.replace(tmp, so)
    lib = C.CDLL(so)
    u8p = C.POINTER(C.c_ubyte)
    lib.hostemu_HEVCImageEncoder.restype = C.c_int
    lib.hostemu_HEVCImageEncoder.argtypes = [u8p, u8p, u8p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int), C.c_int]
    return lib


@pytest.fixture(scope="session")
def hostemu(built):
    """TEST-ONLY host emulation of the device source (tests/hostemu)."""
    return _hostemu_lib("libhostemu.so", [])


@pytest.fixture(scope="session")
def hostemu_ovf(built):
    """Same, built so that every trial coder's lead sink reports the emulation-prevention pattern and every winner's leads are turned into bytes by
    lane 0's walk (-DIMCVT_FORCE_OVF): the exact paths of hevc_core.h leads_exact / resolve_leads on every candidate."""
    return _hostemu_lib("libhostemu_ovf.so", ["-DIMCVT_FORCE_OVF"])


@pytest.fixture(scope="session")
def hostemu_esc(built):
    """Same, with every escape code word of the token generator taking the out-of-line path (tokg_b's rare branch)."""
    return _hostemu_lib("libhostemu_esc.so", ["-DTOKB_INLINE_BINS=14"])


@pytest.fixture(scope="session")
def hostemu_row(built):
    """Same, with 6-token lane rows: nearly every pass overflows its token rows and takes the count-then-write path."""
    return _hostemu_lib("libhostemu_row.so", ["-DROWCAP=6"])


@pytest.fixture(scope="session")
def hostemu_vec(built):
    """Same, with the N = 16 / 32 transforms as vector code instead of (emulated) matrix instructions: the A/B build."""
    return _hostemu_lib("libhostemu_vec.so", ["-DP1_MFMA=0"])


@pytest.fixture(scope="session")
def hostemu_pipe(built):
    """Same, with 256-thread workgroups: the fourth wavefront is the pipe wave (hevc_frame.h nxn_pipe), which prices the NxN
    candidate of every 8x8 CU for all 35 possible modes of the last PU while the PU wave is still working on it."""
    return _hostemu_lib("libhostemu_pipe.so", ["-DEMU_DEFAULT_PIPE"])


@pytest.fixture(scope="session")
def hostemu_pipe_ovf(built):
    """Pipe wave + the exact paths of the lead sinks on every candidate (the pipe wave's three-segment stream included)."""
    return _hostemu_lib("libhostemu_pipe_ovf.so", ["-DEMU_DEFAULT_PIPE", "-DIMCVT_FORCE_OVF"])


@pytest.fixture(scope="session")
def hostemu_wide(built):
    """Same, with 512-thread workgroups: pipe wave + four partner wavefronts, each running the byte half of the trial coders whose
    range half its owner runs (hevc_core.h stream_seg_R / stream_seg_L; hevc_frame.h partner_trial / partner_pu / partner_pipe)."""
    return _hostemu_lib("libhostemu_wide.so", ["-DEMU_DEFAULT_WIDE"])


@pytest.fixture(scope="session")
def hostemu_wide_ovf(built):
    """Wide workgroups + the exact paths of the lead sinks on every candidate (the partners' byte halves, PU pricing)."""
    return _hostemu_lib("libhostemu_wide_ovf.so", ["-DEMU_DEFAULT_WIDE", "-DIMCVT_FORCE_OVF"])


@pytest.fixture(scope="session")
def hostemu_wide_ep(built):
    """Wide workgroups with a wide net for the trial coders' emulation-prevention guard (low bytes up to 0x1F count as zero bytes): many
    candidates get their byte count on the exact path (hevc_core.h leads_exact)."""
    return _hostemu_lib("libhostemu_wide_ep.so", ["-DEMU_DEFAULT_WIDE", "-DEP_GUARD_WIDE"])


@pytest.fixture(scope="session")
def hostemu_abn(built):
    """Same, with main workgroups that stop waiting for a helper's answer after three polls: exercises the path on which a late
    answer is abandoned, the CU evaluated by the main workgroup itself and the mailbox left alone until the answer has arrived."""
    return _hostemu_lib("libhostemu_abn.so", ["-DABANDON_POLLS=3"])


def kat_entries():
    import json
    return json.load(open(os.path.join(ROOT, "tests", "golden", "hevc_kat.json")))


def kat_input(spec):
    """Materialise a golden input (committed pixel file or seeded generator)."""
    import numpy as np
    from oracle import synth
    if spec["kind"] == "file":
        data = open(os.path.join(ROOT, "tests", "golden", spec["file"]), "rb").read()
        hdr = data.split(b"\n", 3)
        w, h = map(int, hdr[1].split())
        return np.frombuffer(hdr[3], dtype=np.uint8, count=w * h).reshape(h, w)
    return getattr(synth, spec["kind"])(spec["w"], spec["h"], spec["arg"])


def kat_id(e):
    s = e["input"]
    name = s.get("file") or f"{s['kind']}{s['w']}x{s['h']}a{s['arg']}"
    return f"{name}-q{e['qpd6']}"
