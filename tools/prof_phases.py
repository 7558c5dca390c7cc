#!/usr/bin/env python3
"""Phase breakdown of the encoder kernel (cycle counters per wave).  Needs a -DIMCVT_PROF build:
   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DIMCVT_PROF imcvt_amd/csrc/hevc_hip.hip -o gpurun_out/libimcvt_hevc_prof.so
   IMCVT_HEVC_LIB=gpurun_out/libimcvt_hevc_prof.so python tools/prof_phases.py [w h nframes q]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import imcvt_amd
from imcvt_amd import synth
w, h, n, q = (int(a) for a in (sys.argv[1:5] + ["512", "256", "1", "0"][len(sys.argv) - 1:]))
enc = imcvt_amd.DeviceEncoder()
frames = [torch.from_numpy(synth.syn(w, h, s)).cuda() for s in range(n)]
batch = enc.make_batch(frames, q)
enc.encode(batch); torch.cuda.synchronize(); enc.debug_prof(True)
t = time.time(); enc.encode(batch); torch.cuda.synchronize(); dt = time.time() - t
ms = enc.last_kernel_ms()
prof = enc.debug_prof(True)
nctu = ((w + 31) // 32) * ((h + 31) // 32) * n
print(f"team {enc.last_team()}  ", end="")
print(f"{n} x {w}x{h} q{q}: kernel {ms:.1f} ms  ({ms * 1e3 / nctu * n:.1f} us per CTU per frame, {w*h*n/ms/1e3:.3f} Mpx/s)")
cats = enc.PROF_CATS
print("cycles per CTU (per wave):")
print("wave " + " ".join(f"{c:>8s}" for c in cats) + "    total")
for wv, row in enumerate(prof):
    if not any(row):
        continue
    print(f"{'MHh'[wv // 3]}{wv % 3:3d} " + " ".join(f"{v / nctu:8.0f}" for v in row) + f" {sum(row) / nctu:9.0f}")
