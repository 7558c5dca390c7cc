#!/usr/bin/env python3
"""HBM traffic of hevc_encode_frames from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as the
MI355X guide prescribes).  FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE tallies 128-byte
requests at 64 B, so wide coalesced reads are under-reported by 2x — both the raw and the x2-corrected figures are
kept (this kernel's reads are mostly 16- and 32-byte segments, so the correction is an upper bound).
usage: python tools/pmc_traffic.py <fetch.db> <write.db> <frames> <qpd6> > profiles/pmc_traffic.json"""
import json, sqlite3, sys

def per_launch(dbfile, counter):
    cur = sqlite3.connect(dbfile).cursor()
    rows = list(cur.execute("select dispatch_id, sum(value) from counters_collection where kernel_name like 'hevc_encode_frames%' and counter_name=? group by dispatch_id order by dispatch_id", (counter,)))
    vals = [v for _, v in rows]
    return vals

fetch = per_launch(sys.argv[1], "FETCH_SIZE")
write = per_launch(sys.argv[2], "WRITE_SIZE")
f_kib = sum(fetch[1:]) / max(1, len(fetch) - 1) if len(fetch) > 1 else fetch[0]      # skip the warm-up launch
w_kib = sum(write[1:]) / max(1, len(write) - 1) if len(write) > 1 else write[0]
out = {"frames": int(sys.argv[3]), "qpd6": int(sys.argv[4]), "launches_profiled": [len(fetch), len(write)],
       "fetch_kib_raw": f_kib, "write_kib_raw": w_kib,
       "hbm_bytes_per_launch_raw": int((f_kib + w_kib) * 1024),
       "hbm_bytes_per_launch": int((2 * f_kib + w_kib) * 1024),
       "note": "hbm_bytes_per_launch = 2*FETCH_SIZE + WRITE_SIZE (gfx950 fetch correction, upper bound for this access pattern)"}
print(json.dumps(out, indent=1))
