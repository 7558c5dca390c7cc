"""Build libimcvt_hevc.so for gfx950 in-tree with hipcc (cross-compiles without a GPU)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SRC = ["hevc_hip.hip"]
DEPS = ["hevc_hip.hip", "hevc_core.h", "hevc_frame.h", "hevc_tables.h", os.path.join("..", "..", "include", "imcvt_hevc.h")]
OUT = os.path.join(CSRC, "libimcvt_hevc.so")


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
           *[os.path.join(CSRC, s) for s in SRC], "-o", OUT]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))
