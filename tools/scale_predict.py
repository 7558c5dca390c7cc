#!/usr/bin/env python3
"""Strong-scaling prediction of BASELINE configs[3] (512 x 1080p frames over N GPUs) from ONE GPU: the share a rank would own at
N = 1, 2, 4, 8 (512 / N frames) is run through bench.py's own distributed path (IMCVT_BENCH_FORCE_DIST=1: RCCL process group,
stream gather, digest checks against the reference), one rank on this GPU.  The job time at N GPUs is the time of one share
(ranks run concurrently, the gather to rank 0 moves <= 0.7 GB over xGMI); what one GPU cannot show is the N-1 incoming
point-to-point receives on rank 0.  Writes profiles/r03_scale_prediction.json.
usage: python tools/scale_predict.py [--out profiles/r03_scale_prediction.json] [--ns 1,2,4,8]"""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_scale_prediction.json"))
ap.add_argument("--ns", default="1,2,4,8")
ap.add_argument("--total", type=int, default=512)
a = ap.parse_args()
rows = []
for n in [int(v) for v in a.ns.split(",")]:
    share = a.total // n
    env = dict(os.environ, IMCVT_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29611 + n), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--total-frames", str(share), "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-latency-view"], env=env, capture_output=True, text=True)
    line = next((l for l in r.stdout.splitlines()[::-1] if l.startswith("{")), None)
    if r.returncode != 0 or line is None:
        print(r.stdout[-2000:], r.stderr[-2000:]); raise SystemExit(f"bench.py failed for the share of N={n}")
    j = json.loads(line)
    rows.append({"n_gpus": n, "frames_per_gpu": share, "ms_per_step_of_one_share": j["ms_per_step"], "kernel_ms": j["roofline"]["kernel_ms"],
                 "verified": j["config"]["verified"], "parallelism": j["config"]["parallelism"]})
    print(rows[-1], flush=True)
W, H = 1920, 1080
t1 = rows[0]["ms_per_step_of_one_share"] * rows[0]["n_gpus"] / 1.0 if rows[0]["n_gpus"] == 1 else None
for r in rows:
    r["predicted_job_mpx_s"] = round(a.total * W * H / r["ms_per_step_of_one_share"] / 1e3, 2)
    if t1:
        r["predicted_speedup_vs_1gpu"] = round(t1 / r["ms_per_step_of_one_share"], 3)
        r["predicted_efficiency"] = round(t1 / r["ms_per_step_of_one_share"] / r["n_gpus"], 3)
out = {"what": "per-GPU shares of the 512-frame job run on ONE MI355X through bench.py's RCCL path (IMCVT_BENCH_FORCE_DIST=1); job time at N GPUs = time of one share",
       "total_frames": a.total, "rows": rows}
json.dump(out, open(a.out, "w"), indent=1)
print(json.dumps(out))
