#!/usr/bin/env python3
"""Dump PMC counter values per dispatch of hevc_encode_frames from a rocprofv3 rocpd database."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
v = [t for t in tabs if t in ("counters_collection", "pmc_events", "counters")]
print("views:", [t for t in tabs if "count" in t.lower() or "pmc" in t.lower()])
for t in v:
    cols = [d[1] for d in cur.execute(f"pragma table_info({t})")]
    print(t, cols)
try:
    rows = list(cur.execute("select counter_name, sum(value), count(*) from counters_collection where name like 'hevc_encode_frames%' group by counter_name"))
    for r in rows: print(r)
except Exception as e:
    print("query failed:", e)
