#!/usr/bin/env python3
"""[developer measurement tool] The encoder kernel's DYNAMIC VALU opcode mix, and the issue peak that follows from it.

Round 5 weighted the per-opcode issue costs (tools/valu_mix_probe.py: cycles per wave64 instruction per SIMD, four wavefronts per SIMD on all 256 compute
units) with STATIC opcode counts of the kernel's ISA.  What a SIMD sees is the dynamic mix: the token step of the stream coders and the 4x4 / 8x8 passes
run hundreds of times per CTU, the decision code once per CU.  This tool

  1. compiles the device source with -DIMCVT_MARK (comment markers at the ends of the marked regions: the steps of the pipeline passes p1_run_t<LG> per
     transform size, of the 4x4 pass p1_run_4, the stream coder's token-block loop, the wide PU step) and takes every region's static opcode histogram
     (the code between the previous marker and the region's marker; averaged over the inlined copies of a region);
  2. runs a -DIMCVT_REGCNT build (hevc_core.h RCNT: every region's end marker counts its wave executions) on bench frames in the bench's launch shape and
     reads the counters (GPU; `--counts file.json` replays a saved run);
  3. dynamic histogram = sum over regions of (static histogram x executions); the rest of the kernel (borders, headers, decisions, pool code: everything
     outside the marked regions) enters with the kernel's static mix outside the regions, scaled to what is missing from the measured VALU instructions
     per CTU (profiles/pmc_issue.json, SQ_INSTS_VALU of the same workload);
  4. weights profiles/valu_mix.json's per-opcode cycles with it.

usage (GPU box): python tools/valu_dyn_mix.py [--frames 64] [--out profiles/valu_dyn_mix.json]
       (anywhere): python tools/valu_dyn_mix.py --counts profiles/r06_region_counts.json"""
import argparse, collections, json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CSRC = os.path.join(ROOT, "imcvt_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-mllvm", "-disable-machine-licm"]
P4 = ["b4_setup", "b4_predict", "b4_residual_dst", "b4_rdoq", "b4_tokens", "b4_inverse_recon_sse"]
PT = ["pass_setup", "mx_predict", "mx_forward", "predict", "residual", "fwd_stage1", "fwd_stage2", "rdoq", "scan_dequant_cfg", "group_tokens", "tokens_to_stream",
      "dequant_store", "mx_inverse", "inv_stage1", "inv_stage2_recon_sse"]
IDS = {(n, 0): i for i, n in enumerate(P4)}
IDS.update({(n, s): 8 + 4 * r + s for r, n in enumerate(PT) for s in (1, 2, 3)})
IDS.update({("p2_ring_sync", 0): 72, ("p2_eight_tokens", 0): 73, ("a4_stage1", 0): 74, ("a4_partA", 0): 75})

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=64)
ap.add_argument("--counts", default=None, help="replay saved region counters instead of running the -DIMCVT_REGCNT build")
ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "valu_dyn_mix.json"))
ap.add_argument("--save-counts", default=os.path.join(ROOT, "profiles", "r06_region_counts.json"))
a = ap.parse_args()


def static_regions():
    """(name, s) -> (mean opcode Counter over the copies, copies); and the opcode Counter of everything outside the regions."""
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run([HIPCC, *FLAGS, "-DIMCVT_MARK", "-S", "--cuda-device-only", os.path.join(CSRC, "hevc_hip.hip"), "-o", out], check=True, stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    copies = collections.defaultdict(list)
    outside, cur = collections.Counter(), collections.Counter()
    for ln in lines:
        if re.match(r"^_Z\w+:", ln):                        # a new function: what was collected since the last marker lies outside the regions
            outside.update(cur); cur = collections.Counter(); continue
        m = re.search(r"; MARK (\w+) (s(\d)|begin)", ln)
        if m:
            if m.group(2) == "begin":
                outside.update(cur)
            else:
                copies[(m.group(1), int(m.group(3)))].append(cur)
            cur = collections.Counter(); continue
        t = ln.strip().split()
        if t and t[0].startswith("v_") and not t[0].startswith("v_mfma"):
            cur[t[0]] += 1
    outside.update(cur)
    mean = {}
    for k, cs in copies.items():
        tot = collections.Counter()
        for c in cs:
            tot.update(c)
        mean[k] = (collections.Counter({op: n / len(cs) for op, n in tot.items()}), len(cs))
    return mean, outside


def region_counts():
    if a.counts:
        return json.load(open(a.counts))
    import torch
    os.environ["IMCVT_HEVC_LIB"] = os.path.join(CSRC, "variants", "libimcvt_hevc_regcnt.so")
    if not os.path.exists(os.environ["IMCVT_HEVC_LIB"]):
        subprocess.run([HIPCC, *FLAGS, "-fPIC", "-shared", "-DIMCVT_REGCNT", os.path.join(CSRC, "hevc_hip.hip"), "-o", os.environ["IMCVT_HEVC_LIB"]], check=True)
    import imcvt_amd
    from imcvt_amd import synth
    enc = imcvt_amd.DeviceEncoder()
    n = a.frames
    b = enc.make_batch([torch.from_numpy(synth.syn(1920, 1080, s)).cuda() for s in range(n)], 0)
    # the bench's launch shape in proportion (512 main + 448 helper workgroups of 192 threads): mains : helpers = 8 : 7, no pipe wave, no wide workgroups
    enc.set_pipe(0); enc.set_wide(0); enc.set_shape(n, max(1, n * 7 // 8))
    enc.debug_regions(True)
    enc.encode(b); torch.cuda.synchronize()
    cnt = enc.debug_regions(True)
    ctus = n * 34 * 60
    rec = {"frames": n, "ctus": ctus, "shape": list(enc.last_shape()), "kernel_ms": enc.last_kernel_ms(), "counts": cnt}
    enc.close()
    json.dump(rec, open(a.save_counts, "w"))
    return rec


mean, outside = static_regions()
rec = region_counts()
cnt, ctus = rec["counts"], rec["ctus"]
dyn = collections.Counter()
rows = []
for key, (hist, ncopies) in sorted(mean.items(), key=lambda kv: IDS.get(kv[0], 999)):
    rid = IDS.get(key)
    if rid is None or rid >= len(cnt):
        continue
    ex = cnt[rid] / ctus
    sv = sum(hist.values())
    rows.append({"region": key[0], "s": key[1], "copies": ncopies, "static_valu": round(sv, 1), "executions_per_ctu": round(ex, 2), "valu_per_ctu": round(sv * ex)})
    for op, n in hist.items():
        dyn[op] += n * ex
marked = sum(dyn.values())
pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_issue.json")))
measured = pm["valu_wave_insts_per_ctu"]
rest = max(0.0, measured - marked)
so = sum(outside.values())
for op, n in outside.items():
    dyn[op] += n / so * rest
mv = json.load(open(os.path.join(ROOT, "profiles", "valu_mix.json")))
per = mv["per_opcode"]
tot = sum(dyn.values())
cov = sum(n for op, n in dyn.items() if op in per)
simd = sum(n * per[op]["cycles_simd_4_waves"] for op, n in dyn.items() if op in per) / cov
lone = sum(n * per[op]["cycles_lone_wave"] for op, n in dyn.items() if op in per) / cov
static_simd = mv["mix_weighted_cycles_simd"]
out = {"what": "dynamic VALU opcode mix of hevc_encode_frames at the bench's launch shape: static opcode histograms of the marked regions x their measured executions per CTU "
               "(-DIMCVT_REGCNT build), the rest of the kernel with its static mix scaled to the measured SQ_INSTS_VALU per CTU; per-opcode issue costs from profiles/valu_mix.json",
       "frames": rec["frames"], "ctus": ctus, "launch_shape": rec.get("shape"), "valu_per_ctu_measured": measured, "valu_per_ctu_in_marked_regions": round(marked),
       "share_in_marked_regions": round(marked / measured, 4), "share_covered_by_opcode_table": round(cov / tot, 4),
       "mix_weighted_cycles_simd": round(simd, 3), "mix_weighted_cycles_lone_wave": round(lone, 3), "static_mix_weighted_cycles_simd": static_simd, "clock_ghz": mv["clock_ghz"],
       "issue_peak_G_wave_inst_per_s": round(256 * 4 * mv["clock_ghz"] / simd, 1),
       "top_opcodes": [{"op": op, "dynamic_share": round(n / tot, 4), "cycles_simd_4_waves": per.get(op, {}).get("cycles_simd_4_waves")} for op, n in dyn.most_common(24)],
       "regions": rows}
json.dump(out, open(a.out, "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k not in ("regions", "top_opcodes")}, indent=1))
if len(cnt) > 76 and cnt[76]:
    print(f"lead-sink flushes (wave executions of lsink_flush8): {cnt[76] / ctus:.1f} per CTU against {cnt[72] / ctus:.1f} token blocks of the stream coders")
for r in sorted(rows, key=lambda r: -r["valu_per_ctu"])[:40]:
    print("%-22s s%d  copies %d  static %7.1f  x %9.2f per CTU = %9d" % (r["region"], r["s"], r["copies"], r["static_valu"], r["executions_per_ctu"], r["valu_per_ctu"]))
