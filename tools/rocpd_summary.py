#!/usr/bin/env python3
"""Text summary of a rocprofv3 (ROCm 7.2 `rocpd` SQLite) result: per-kernel stats + every dispatch of our kernel.
usage: python tools/rocpd_summary.py gpurun_out/rocprof_r01/r01_results.db > profiles/r01_kernel_trace_stats.txt"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
print("# rocprofv3 --kernel-trace --stats summary (from", sys.argv[1].split("/")[-1] + ")")
print("# kernel | calls | total_ms | avg_ms | pct")
for name, calls, tot, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print(f"{name.split('(')[0][:60]:60s} | {calls:5d} | {tot/1e3:12.3f} | {avg/1e3:12.3f} | {pct:8.4f}")
print("\n# dispatches of hevc_encode_frames: duration_ms grid workgroup lds_bytes scratch_bytes vgpr accum_vgpr sgpr")
for r in cur.execute("select (end-start)/1e6, grid_x, workgroup_x, lds_size, scratch_size, vgpr_count, accum_vgpr_count, sgpr_count from kernels where name like 'hevc_encode_frames%' order by start"):
    print("%.3f %d %d %d %d %d %d %d" % r)
