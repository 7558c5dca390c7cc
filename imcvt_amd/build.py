"""Build libimcvt_hevc.so for gfx950 in-tree with hipcc (cross-compiles without a GPU)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SRC = ["hevc_hip.hip", "hevc_wide.hip"]      # one object each (different code generation flags, below), linked into one library
DEPS = ["hevc_hip.hip", "hevc_wide.hip", "hevc_core.h", "hevc_frame.h", "hevc_tables.h", os.path.join("..", "..", "include", "imcvt_hevc.h")]
OUT = os.path.join(CSRC, "libimcvt_hevc.so")


def _run_to(cmd, out):
    """Run a compile/link command whose output path is `out` through a private temporary file and rename it into place:
    concurrent builders (pytest -n) never see, execute or overwrite a half-written binary."""
    tmp = f"{out}.tmp.{os.getpid()}"
    try:
        subprocess.run([*cmd, "-o", tmp], check=True)
        os.replace(tmp, out)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)


# Code generation of the encoder kernel.  Machine LICM hoists ~80 loop-invariant values per candidate-set call out of the pass loops; at
# 168 registers they are spilled at once and reloaded inside the loops.  Without it: private segment 1008 -> 720 B per lane, HBM traffic
# 5.8 -> 4.5 MB per CTU, same speed (A/B on one box, profiles/r03w2_licm_ab.log).
KERNEL_FLAGS = ["-mllvm", "-disable-machine-licm"]
# The instantiation for wide launches (hevc_wide.hip: 512-thread workgroups, 256 registers per wavefront) keeps machine LICM: what it hoists fits.
WIDE_FLAGS = [f for f in os.environ.get("IMCVT_WIDE_FLAGS", "").split() if f]
SRC_FLAGS = {"hevc_hip.hip": KERNEL_FLAGS, "hevc_wide.hip": WIDE_FLAGS}
BASE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]


def _hipcc() -> str:
    return os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _source_hash(deps, flags) -> str:
    """sha256 over the sources a library is built from and the flags it is built with."""
    import hashlib
    h = hashlib.sha256(" ".join([_hipcc(), *BASE_FLAGS, "-shared", *flags]).encode())      # the whole command line, compiler path included
    for d in deps:
        with open(os.path.join(CSRC, d), "rb") as f:
            h.update(d.encode() + b"\0" + f.read())
    return h.hexdigest()


def _stamp(out: str) -> str:
    return out + ".srchash"          # travels with the library (git-ignored like it)


def _is_current(out, deps, flags) -> bool:
    """The library exists and was built from exactly these sources with exactly these flags (content, not modification times:
    a prebuilt library that travelled to another machine is rebuilt there only if what it was built from differs)."""
    try:
        return os.path.exists(out) and open(_stamp(out)).read().strip() == _source_hash(deps, flags)
    except OSError:
        return False


def _write_stamp(out, digest):
    """`digest` = _source_hash() taken BEFORE the compiler ran: a source edited during the compile leaves a stamp that no longer matches."""
    with open(_stamp(out), "w") as f:
        f.write(digest + "\n")


def needs_build() -> bool:
    return not _is_current(OUT, DEPS, ALL_FLAGS)


ALL_FLAGS = [*KERNEL_FLAGS, "|", *WIDE_FLAGS]      # what the library's stamp covers: both objects' flags


JLS_OUT = os.path.join(CSRC, "libimcvt_jls.so")   # JPEG-LS (BASELINE config 5)
JLS_DEPS = ["jls_hip.hip", "jls_core.h", "jls_par.h", os.path.join("..", "..", "include", "imcvt_jls.h")]


def build_jls(force: bool = False) -> str:
    if force or not _is_current(JLS_OUT, JLS_DEPS, []):
        digest = _source_hash(JLS_DEPS, [])
        _run_to([_hipcc(), *BASE_FLAGS, "-shared", os.path.join(CSRC, "jls_hip.hip")], JLS_OUT)
        _write_stamp(JLS_OUT, digest)
    return JLS_OUT


HOST = os.path.join(CSRC, "host")
CLI = os.path.join(CSRC, "imcvt")                  # the drop-in converter binary (reference: src/main.c)
PNM_SO = os.path.join(CSRC, "libimcvt_pnm.so")     # the image-file readers / writers alone (PNM, PNG, BMP, QOI), for the host-side tests
IO_SRC = ["pnm_io.cpp", "png_io.cpp", "bmp_io.cpp", "qoi_io.cpp"]


def build_host(force: bool = False) -> str:
    """Host glue of the drop-in binary: image-file I/O + CLI, linked against libimcvt_hevc.so (plain g++, no device code)."""
    srcs = [os.path.join(HOST, "imcvt_cli.cpp")] + [os.path.join(HOST, f) for f in IO_SRC]
    build_jls()
    deps = srcs + [OUT, JLS_OUT, os.path.join(CSRC, "..", "..", "include", "imcvt_hevc.h")]
    stale = lambda o: force or not os.path.exists(o) or any(os.path.getmtime(d) > os.path.getmtime(o) for d in deps)
    if stale(PNM_SO):
        _run_to(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", *srcs[1:]], PNM_SO)
    if stale(CLI):
        _run_to(["g++", "-O2", "-std=c++17", "-DIMCVT_WITH_JLS", *srcs, "-L" + CSRC, "-limcvt_hevc", "-limcvt_jls", "-Wl,-rpath,$ORIGIN"], CLI)
    return CLI


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        build_host()
        return OUT
    digest = _source_hash(DEPS, ALL_FLAGS)
    objs = []
    try:
        procs = []
        for src in SRC:                                  # the two objects compile side by side (half a minute each)
            obj = os.path.join(CSRC, f"{os.path.splitext(src)[0]}.tmp.{os.getpid()}.o")
            objs.append(obj)
            cmd = [_hipcc(), *BASE_FLAGS, *SRC_FLAGS[src], "-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
            procs.append((cmd, subprocess.Popen(cmd)))
        for cmd, pr in procs:
            if pr.wait() != 0:
                raise subprocess.CalledProcessError(pr.returncode, cmd)
        _run_to([_hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared", *objs], OUT)
    finally:
        for o in objs:
            if os.path.exists(o):
                os.remove(o)
    _write_stamp(OUT, digest)
    build_host(force=True)
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))
