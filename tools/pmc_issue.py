#!/usr/bin/env python3
"""Issue-side summary of the SQ counter passes (tools/gpu_pmc.sh -> *_pmc_sq.txt) as profiles/pmc_issue.json (bench.py's roofline_issue reads it).
Per counter the LAST dispatch of hevc_encode_frames is taken (the first full launch of the process is the warm-up).
usage: python tools/pmc_issue.py gpurun_out/TAG_pmc_sq.txt frames w h q "description of the launch shape" > profiles/pmc_issue.json"""
import json, sys
vals = {}
for ln in open(sys.argv[1]):
    f = ln.split()
    if len(f) >= 4 and f[1] == "dispatch":
        vals[f[0]] = float(f[3])                     # later dispatches overwrite earlier ones
frames, w, h, q = (int(v) for v in sys.argv[2:6])
ctus = frames * ((w + 31) // 32) * ((h + 31) // 32)
g = lambda k: vals.get(k, 0.0)
cycles_per_xcc = g("GRBM_GUI_ACTIVE") / 8.0          # the counter is summed over the 8 XCCs
simds = 1024
out = {"source": f"{sys.argv[1].replace('gpurun_out/', 'profiles/')} (rocprofv3 --pmc, {frames} x {w}x{h} frames, qpd6={q}, {sys.argv[6] if len(sys.argv) > 6 else ''}; one launch = {ctus} CTUs)",
       "valu_wave_insts_per_ctu": round(g("SQ_INSTS_VALU") / ctus, 1), "salu_wave_insts_per_ctu": round(g("SQ_INSTS_SALU") / ctus, 1),
       "lds_wave_insts_per_ctu": round(g("SQ_INSTS_LDS") / ctus, 1), "vmem_wave_insts_per_ctu": round((g("SQ_INSTS_VMEM_RD") + g("SQ_INSTS_VMEM_WR")) / ctus, 1),
       "valu_lane_activity": round(g("SQ_THREAD_CYCLES_VALU") / (64.0 * g("SQ_INSTS_VALU")), 4) if g("SQ_INSTS_VALU") else None,
       "waves_per_simd": round(g("SQ_WAVE_CYCLES") * 4 / (cycles_per_xcc * simds), 3) if cycles_per_xcc else None,
       "valu_busy_frac": round(g("SQ_ACTIVE_INST_VALU") * 4 / (cycles_per_xcc * simds), 4) if cycles_per_xcc else None,
       "wave_wait_frac": round(g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), 4) if g("SQ_WAVE_CYCLES") else None,
       "lds_bank_conflict_frac": round(g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE"), 4) if g("SQ_LDS_IDX_ACTIVE") else None,
       "mfma_i8_wave_insts_per_ctu": round(g("SQ_INSTS_VALU_MFMA_I8") / ctus, 1) if "SQ_INSTS_VALU_MFMA_I8" in vals else None,
       "mfma_busy_frac": round(g("SQ_VALU_MFMA_BUSY_CYCLES") / (cycles_per_xcc * 256), 5) if ("SQ_VALU_MFMA_BUSY_CYCLES" in vals and cycles_per_xcc) else None,
       "units": "SQ_ACTIVE_* / SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles (x4 = cycles); GRBM_GUI_ACTIVE is summed over the 8 XCCs; 1024 SIMDs",
       "frames": frames, "w": w, "h": h, "qpd6": q}
import os
try:      # the build these counters were taken on (bench.py marks the numbers stale when the timed library differs)
    out["lib_srchash"] = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "imcvt_amd", "csrc", "libimcvt_hevc.so.srchash")).read().strip()
except OSError:
    out["lib_srchash"] = None
print(json.dumps(out, indent=1))
