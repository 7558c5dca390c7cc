#!/usr/bin/env python3
"""HBM traffic of hevc_encode_frames from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as the
MI355X guide prescribes), put on an absolute scale with the 1 GiB device copy tools/pmc_run.py issues afterwards
(known: 2^30 bytes read, 2^30 bytes written).
usage: python tools/pmc_traffic.py <fetch.db> <write.db> <frames> <qpd6> [w h] > profiles/pmc_traffic.json"""
import json, sqlite3, sys

def per_kernel(dbfile, counter):
    cur = sqlite3.connect(dbfile).cursor()
    rows = list(cur.execute("select kernel_name, dispatch_id, sum(value) from counters_collection where counter_name=? group by dispatch_id order by dispatch_id", (counter,)))
    ours = [v for k, _, v in rows if k.startswith("hevc_encode_frames")]
    # the calibration copy is the last dispatch of the process; the fill before it writes 1 GiB and reads nothing
    calib = rows[-1][2] if rows and not rows[-1][0].startswith("hevc_encode_frames") else None
    return ours, calib, [(k[:48], v) for k, _, v in rows if not k.startswith("hevc_encode_frames")][-3:]

fetch, fcal, ftail = per_kernel(sys.argv[1], "FETCH_SIZE")
write, wcal, wtail = per_kernel(sys.argv[2], "WRITE_SIZE")
f_raw = fetch[-1]; w_raw = write[-1]                       # the timed launch (the first one is the warm-up)
GIB = float(1 << 30)
f_scale = GIB / fcal if fcal else None                      # bytes per counter unit, measured
w_scale = GIB / wcal if wcal else None
out = {"frames": int(sys.argv[3]), "qpd6": int(sys.argv[4]), "launches_profiled": [len(fetch), len(write)],
       "fetch_raw": f_raw, "write_raw": w_raw,
       "calibration": {"what": "1 GiB int32 device copy (2^30 B read, 2^30 B written)", "fetch_raw": fcal, "write_raw": wcal,
                       "bytes_per_fetch_unit": f_scale, "bytes_per_write_unit": w_scale, "other_kernels_fetch": ftail, "other_kernels_write": wtail},
       "hbm_read_bytes_per_launch": int(f_raw * f_scale) if f_scale else None,
       "hbm_write_bytes_per_launch": int(w_raw * w_scale) if w_scale else None}
if f_scale and w_scale:
    out["hbm_bytes_per_launch"] = out["hbm_read_bytes_per_launch"] + out["hbm_write_bytes_per_launch"]
if len(sys.argv) > 6:
    out["w"], out["h"] = int(sys.argv[5]), int(sys.argv[6])
    out["ctus"] = out["frames"] * ((out["w"] + 31) // 32) * ((out["h"] + 31) // 32)
import os
try:      # the build these counters were taken on (bench.py marks the numbers stale when the timed library differs)
    out["lib_srchash"] = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "imcvt_amd", "csrc", "libimcvt_hevc.so.srchash")).read().strip()
except OSError:
    out["lib_srchash"] = None
print(json.dumps(out, indent=1))
