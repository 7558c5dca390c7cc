#!/usr/bin/env python3
"""Dump PMC counter values per dispatch of hevc_encode_frames from a rocprofv3 rocpd database.
   usage: rocpd_pmc.py results.db [n_ctus]   (n_ctus: also print per-CTU values)"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
nctu = float(sys.argv[2]) if len(sys.argv) > 2 else None
rows = list(cur.execute("select counter_name, dispatch_id, sum(value), max(scratch_size), max(vgpr_count), max(lds_block_size) from counters_collection "
                        "where kernel_name like 'hevc_encode_frames%' group by counter_name, dispatch_id order by counter_name, dispatch_id"))
for name, disp, val, scr, vg, lds in rows:
    per = f"  per-CTU {val / nctu:12.1f}" if nctu else ""
    print(f"{name:28s} dispatch {disp:3d}  {val:18.0f}{per}   (scratch {scr} vgpr {vg} lds {lds})")
