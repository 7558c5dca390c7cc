#!/usr/bin/env python3
"""Static instruction counts of the pipeline's steps: compiles the device source with -DIMCVT_MARK (comment markers at the
step boundaries of p1_run_t, hevc_core.h) and counts the gfx950 instructions between consecutive markers, per function and
per template instance.  Runs without a GPU.  The MAC stages of N=32 are loops (static != dynamic); the rest is straight-line.
usage: python tools/isa_regions.py > profiles/rNN_isa_regions.txt"""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "imcvt_amd", "csrc", "hevc_hip.hip")
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "k.s")
    subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-mllvm", "-disable-machine-licm", "-DIMCVT_MARK",
                    "-S", "--cuda-device-only", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
fn, inst, c = "?", collections.Counter(), collections.Counter()
print("%-22s %-4s %-22s %6s %6s %5s %5s %5s" % ("function", "inst", "region (ends at)", "total", "valu", "salu", "lds", "vmem"))
for ln in lines:
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        fn = m.group(1)[:22]; c.clear(); continue
    m = re.search(r"; MARK (\w+)", ln)
    if m:
        if m.group(1) == "pass_setup":
            inst[fn] += 1
        print("%-22s %-4d %-22s %6d %6d %5d %5d %5d" % (fn, inst[fn], m.group(1), sum(c.values()), c["v"], c["s"], c["l"], c["m"]))
        c.clear(); continue
    t = ln.strip()
    if t.startswith("v_"): c["v"] += 1
    elif t.startswith("s_"): c["s"] += 1
    elif t.startswith("ds_"): c["l"] += 1
    elif t.startswith(("global_", "scratch_", "buffer_", "flat_")): c["m"] += 1
