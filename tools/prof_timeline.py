#!/usr/bin/env python3
"""Timeline of the 8x8 CUs of a wide workgroup (average time of each event since the CU was entered, in shader-clock cycles).  Needs a
-DIMCVT_PROF -DIMCVT_PROF_TL build (hevc_core.h tl_mark):
   IMCVT_HEVC_LIB=.../libimcvt_hevc_tl.so python tools/prof_timeline.py [w h q]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import imcvt_amd
from imcvt_amd import synth
w, h, q = (int(a) for a in (sys.argv[1:4] + ["512", "256", "0"][len(sys.argv) - 1:]))
enc = imcvt_amd.DeviceEncoder()
enc.set_wide(1)
batch = enc.make_batch([torch.from_numpy(synth.syn(w, h, 0)).cuda()], q)
enc.encode(batch); torch.cuda.synchronize(); enc.debug_prof(True)
enc.encode(batch); torch.cuda.synchronize()
ms = enc.last_kernel_ms()
prof = enc.debug_prof(True)
flat = [v for row in prof[:3] for v in row]
part = [v for row in prof[6:9] for v in row] if len(prof) >= 9 else []      # role 2: the partner workgroup (8x8 CUs' 2Nx2N sets on a second compute unit)
n = flat[0]
names = {9: "all waves through the candidate sets (barrier)", 10: "winner committed"}
for wv, what in enumerate(["wave 0 (one-TU set: passes, range half)", "wave 1 (four-TU set: passes)", "wave 2 (PU chain)", "wave 3 (pipe)", "wave 4 (four-TU set: coders)",
                           "wave 5 (one-TU set: a pass, byte half)", "wave 6 (pipe: byte half)", "wave 7 (PU rows, PU 3 pricing)"]):
    names[1 + wv] = what + " done"
for k in range(4):
    names[16 + k] = f"PU {k}: levels published"; names[24 + k] = f"PU {k}: first part of the tokens made"; names[20 + k] = f"PU {k}: range half through"
    names[28 + k] = f"PU {k}: remaining-level rows made"; names[32 + k] = f"PU {k}: reconstructions + SSE made"
    names[44 + k] = f"PU {k}: byte half through the first part"; names[48 + k] = f"PU {k}: byte half through the rows"; names[12 + k] = f"PU {k}: priced"
    names[52 + k] = f"PU {k}: decided and kept"; names[36 + k] = f"four-TU set: TU {k} passed"
names[56] = "PU 1: step begins"; names[57] = "PU 1: borders made"; names[58] = "PU 1: pricing: guard checked"; names[59] = "PU 1: pricing: reconstructions seen"; names[60] = "PU 1: pricing: costs stored"; names[61] = "PU 1: mode picked"
names[62] = "PU 1: predicted"; names[63] = "PU 1: transformed"; names[64] = "PU 1: quantised"
names[11] = "request to the partner workgroup is out"; names[67] = "the partner's answer: flag seen"
names[40] = "wave 0: first pass item done"; names[41] = "wave 0: second pass item done"; names[42] = "one-TU set: tokens complete"
if flat[11]:          # (CUs whose 2Nx2N sets went to a partner workgroup: slots 36 .. 42 time the pipe wave and its byte half there)
    for ev in (36, 37, 38, 39, 40, 41, 42):
        names.pop(ev, None)
    names[36] = "pipe wave: 35 headers made"; names[37] = "pipe wave: range half through the headers"; names[38] = "pipe wave: range half through the winners of PUs 0..2"
    names[40] = "pipe byte half: through the headers"; names[41] = "pipe byte half: through the winners of PUs 0..2"; names[42] = "pipe byte half: through PU 3's winner"
print(f"trials finished {flat[66]}, with lanes on the exact path {flat[65]}")
print(f"1 x {w}x{h} q{q}: kernel {ms:.1f} ms, wide {enc.last_wide()}, {n} 8x8 CUs; average cycles since the CU was entered:")
for ev, t in sorted(((ev, flat[ev] / max(n, 1)) for ev in names if flat[ev]), key=lambda x: x[1]):
    print(f"  {t:9.0f}  {names[ev]}")

if part and part[0]:
    pn = part[0]
    pnames = {11: "inputs staged", 1: "wave 0 (one-TU set: pass 0..15, range half) done", 2: "wave 1 (four-TU set: passes, last range half) done", 3: "wave 2 (one-TU set: pass 16..31, byte half) done",
              4: "wave 3 (four-TU set: coders) done", 8: "wave 7 (one-TU set: pass 32..34) done", 9: "all wavefronts through (barrier)", 12: "winner picked", 13: "winner rebuilt", 14: "winner's leads are bytes", 10: "answered",
              42: "one-TU set: tokens complete", 40: "wave 0: first pass item done"}
    for k in range(4):
        pnames[36 + k] = f"four-TU set: TU {k} passed"
    print(f"partner workgroup, {pn} requests; average cycles since the request was seen:")
    for ev, t in sorted(((ev, part[ev] / pn) for ev in pnames if part[ev]), key=lambda x: x[1]):
        print(f"  {t:9.0f}  {pnames[ev]}")
