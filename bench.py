#!/usr/bin/env python3
"""bench.py — Mpixels/s of H.265 intra encode (gray8), frame-sharded over N GPUs (BASELINE.json metric).

Workload (default) = BASELINE configs[3] as written: ONE batch of 512 independent 1920x1080 gray8 frames
syn(1920,1080,seed), seed = 0..511, qpd6 = 0, split over the N ranks (strong scaling: rank r owns a
contiguous block of 512/N frames).  A "step" = one pass of the hot path over that batch: every rank encodes its frames,
already resident in HBM, through the device-resident C ABI (imcvt_hevc_encode_device), then the encoded streams are
gathered to rank 0 over RCCL (all-gather of lengths + one point-to-point send per peer, imcvt_amd/shard.py) — the one
exchange step the job has.  `--scaling weak --frames F` keeps F frames per GPU instead (seeds rank*F+i).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling strong|weak] [--total-frames T] [--frames F] [--qpd6 Q]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Started plain with --gpus N > 1 it launches its N ranks itself (torch.distributed.run on 127.0.0.1).
Rank 0 prints ONE JSON line.  value = input pixels (W*H) of all ranks' frames x K / max-over-ranks wall time of the K steps.

Checks on what was timed (failing any aborts the run): every frame's stream digest is equal between the last warm-up
step and the last timed step; every stream gathered on rank 0 equals the REAL reference's digest
(tests/golden/bench512_kat.json for seeds 0..511, hevc_kat.json for seeds 0..7) — frames without a golden digest are
compared with the CPU checker on a sample, or the line says "unverified".

roofline: the HBM view the metric asks for — algorithmic bytes (W*H read + Wp*Hp reconstruction written + stream bytes
written) per launch / HIP-event duration of the kernel on its launch stream.
roofline_issue: the bound that binds (VALU issue).  cpu_baseline: the CPU checker on a bounded sample (N=1 only).
latency_view: one 1080p frame and one 4K frame alone on the GPU (BASELINE configs 1 and 2), kernel ms; the reference's P4 sample picture
through writeHEVCImageFile (configs[0]), wall ms (N=1 only).  qpd6_4_view: the timed batch once more at qpd6 = 4 (N=1 only).
host_abi_view: the reference-shaped host-pointer entry point over ALL the bench frames, PCIe copies included (N=1 only).
solo_1000f: secondary throughput with the device full (1000 frames, a frame per workgroup), same kernel (N=1 only).
jls_view: BASELINE config 5 (1920x1080 gray8 -> .jls, NEAR=0): one plane and 64 planes, kernel ms, reference digest (N=1 only).
All of these run after the timed region and do not enter `value`.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H = 1920, 1080


def _gen(seed):
    from imcvt_amd import synth
    return synth.syn(W, H, seed)


def _cpu_unit(args):
    """CPU baseline unit: one syn(1920, rows, seed) picture through the CPU checker; returns its own seconds."""
    seed, q, rows = args
    from imcvt_amd import synth
    from oracle import oracle              # cpu_baseline leg: the CPU checker is what is timed here
    img = synth.syn(W, rows, seed)
    t = time.perf_counter()
    oracle.cpu_encode(img, q)
    return time.perf_counter() - t


def cpu_baseline(qpd6):
    """The reference on this box's host cores, one process per core (`cores` whole frames of the bench workload).
    A 1080p frame takes a core ~75 s, so `value` comes from cores x ONE whole syn(1920,1080,seed) frame; the round-1..3 sample
    (cores x a 1920x256 strip, ~18 s) is kept beside it as `strips` — same content class, a quarter of a frame's context adaptation."""
    from multiprocessing import Pool
    from oracle import oracle
    cores = max(1, min(os.cpu_count() or 1, 32))
    out = {}
    for name, rows in (("strips", 256), ("frames", H)):
        t0 = time.perf_counter()
        pool = Pool(cores)
        each = pool.map(_cpu_unit, [(s, qpd6, rows) for s in range(cores)], chunksize=1)
        wall = time.perf_counter() - t0
        pool.close(); pool.join()       # let workers exit normally (a terminate() under rocprofv3 hangs in its signal handler)
        out[name] = {"mpx_s": round(cores * W * rows / wall / 1e6, 4), "wall_s": round(wall, 1), "per_core_mpx_s": round(W * rows / (sum(each) / len(each)) / 1e6, 4)}
    return {"value": out["frames"]["mpx_s"], "unit": "Mpixels/s", "cores": cores,
            "kind": "reference" if oracle.have_ref() else "port",
            "sample": f"{cores} whole syn(1920,1080,seed) frames (seeds 0..{cores - 1}), one process per core, qpd6={qpd6}, {out['frames']['wall_s']} s wall",
            "per_core_mpx_s": out["frames"]["per_core_mpx_s"],
            "strips": dict(out["strips"], sample=f"{cores} x syn(1920,256,seed) strips (8 CTU rows each), the sample of rounds 1-3")}


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _digests(outs, lens):
    return [hashlib.sha256(o[:n].cpu().numpy().tobytes()).hexdigest() for o, n in zip(outs, lens)]


def golden_digests(qpd6):
    """seed -> (bytes, sha256) from the committed reference-generated fixtures."""
    g = {}
    p = os.path.join(ROOT, "tests", "golden", "bench512_kat.json")
    if os.path.exists(p):
        d = json.load(open(p))
        if d.get("qpd6") == qpd6 and d["input"] == {"kind": "syn", "w": W, "h": H}:
            g.update({int(k): (v["bytes"], v["sha256"], v.get("rcon_sha256")) for k, v in d["frames"].items()})
    for e in json.load(open(os.path.join(ROOT, "tests", "golden", "hevc_kat.json"))):
        i = e["input"]
        if i.get("kind") == "syn" and (i.get("w"), i.get("h")) == (W, H) and e["qpd6"] == qpd6:
            g[int(i["arg"])] = (e["bytes"], e["sha256"], e.get("rcon_sha256"))
    return g


def _lib_srchash():
    """Content hash of the sources + flags the timed library was built from (imcvt_amd/build.py writes it beside the .so)."""
    try:
        return open(os.path.join(ROOT, "imcvt_amd", "csrc", "libimcvt_hevc.so.srchash")).read().strip()
    except OSError:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong")
    ap.add_argument("--total-frames", type=int, default=512, help="strong scaling: frames of the whole job (BASELINE configs[3]: 512)")
    ap.add_argument("--frames", type=int, default=1000, help="weak scaling: frames per GPU per step")
    ap.add_argument("--qpd6", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency-view", action="store_true")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # started plain: launch the N ranks ourselves, one per GPU, rendezvous on 127.0.0.1
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd, env=env).returncode)

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs the MI355X"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or os.environ.get("IMCVT_BENCH_FORCE_DIST") == "1"      # (the env knob exercises the RCCL path on a 1-GPU box)
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    import __graft_entry__ as g
    if rank == 0:
        g.build()
    if use_dist:
        dist.barrier()
    import imcvt_amd
    from imcvt_amd import shard
    from multiprocessing import Pool

    strong = args.scaling == "strong"
    seeds = list(shard.split_frames(args.total_frames, rank, world)) if strong else list(shard.frame_range(rank, world, args.frames))
    F = len(seeds)
    total_frames = args.total_frames if strong else args.frames * world
    pool = Pool(max(1, min((os.cpu_count() or 1) // max(1, world), 16)))
    frames_np = pool.map(_gen, seeds, chunksize=4)
    pool.close(); pool.join()
    big = torch.from_numpy(np.stack(frames_np)).to(dev) if F else torch.empty((0, H, W), dtype=torch.uint8, device=dev)   # resident in HBM before any timing
    del frames_np
    enc = imcvt_amd.DeviceEncoder()
    batch = enc.make_batch([big[i] for i in range(F)], args.qpd6)

    def sync_all():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    gathered = None
    kernel_ms = []
    launch_diag = []                                   # (workgroups resident at once, microseconds between the first and the last workgroup's start) per timed launch

    def step(timed):
        nonlocal gathered
        enc.encode(batch)
        if timed:
            kernel_ms.append(enc.last_kernel_ms())      # HIP events on the launch stream (synchronises it)
            launch_diag.append((enc.last_resident(), enc.last_start_spread_us()))
        if use_dist:                                    # the exchange step: encoded streams to rank 0 over RCCL
            lens = batch["lens"].cpu().tolist()
            gathered = shard.gather_streams(shard.pack_streams(batch["outs"], lens), lens, dev)

    for _ in range(args.warmup):
        step(False)
    sync_all()
    first = _digests(batch["outs"], batch["lens"].cpu().tolist()) if args.warmup else None
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    sync_all()
    dt = time.perf_counter() - t0
    dt = shard.max_over_ranks(dt, dev)
    ranks_seen = shard.ranks_observed(dev) if use_dist else 1      # an all-reduce of ones: what the collective backend itself counts
    frames_per_rank = [len(v) for v in shard.gather_lengths([0] * F, dev)] if use_dist else [F]

    # ---- correctness of what was just timed
    lens = batch["lens"].cpu().tolist()
    last = _digests(batch["outs"], lens)
    if first is not None and first != last:
        raise SystemExit(f"rank {rank}: {sum(a != b for a, b in zip(first, last))} of {F} frames changed between steps — result invalid")
    gold = golden_digests(args.qpd6)
    checked = ok = 0
    verified = ""
    if rank == 0:
        if use_dist and gathered is not None:            # what arrived on rank 0, in global frame order
            all_seeds, all_dig, all_len = [], [], []
            for r, (ls, packed) in enumerate(gathered):
                rs = list(shard.split_frames(args.total_frames, r, world)) if strong else list(shard.frame_range(r, world, args.frames))
                assert len(rs) == len(ls)
                all_seeds += rs; all_len += ls
                all_dig += [hashlib.sha256(v.cpu().numpy().tobytes()).hexdigest() for v in shard.unpack_streams(ls, packed)]
            if all_dig[:F] != last:
                raise SystemExit("rank 0: gathered streams differ from the local ones — result invalid")
        else:
            all_seeds, all_dig, all_len = seeds, last, lens
        for s, d, n in zip(all_seeds, all_dig, all_len):
            if s in gold:
                checked += 1
                ok += (gold[s][:2] == (n, d))
        if checked and ok != checked:
            raise SystemExit(f"{checked - ok} of {checked} streams differ from the reference digests — result invalid")
        verified = f"{ok}/{len(all_seeds)} streams sha256-equal to reference digests"
        # the reconstructions of this rank's frames (they stay on the GPU that made them: the gather moves streams only) against the same table
        rc_checked = rc_ok = 0
        for i, sd in enumerate(seeds):
            if sd in gold and gold[sd][2]:
                rc_checked += 1
                rc_ok += hashlib.sha256(batch["rcons"][i].cpu().numpy().tobytes()).hexdigest() == gold[sd][2]
        if rc_checked and rc_ok != rc_checked:
            raise SystemExit(f"{rc_checked - rc_ok} of {rc_checked} reconstructions differ from the reference digests — result invalid")
        verified += f"; {rc_ok}/{F} reconstructions of rank 0 sha256-equal to reference digests"
        if checked < len(all_seeds):                      # no golden digest for some frames: check a sample against the CPU checker
            from oracle import oracle                     # (checker only, outside the timed region)
            todo = [i for i, s in enumerate(seeds) if s not in gold][:2]
            for i in todo:
                ws, wr, _ = oracle.cpu_encode(big[i].cpu().numpy(), args.qpd6)
                if hashlib.sha256(ws).hexdigest() != last[i]:
                    raise SystemExit(f"frame seed {seeds[i]} differs from the CPU checker — result invalid")
            verified += f"; {len(todo)} more byte-equal to the CPU checker; {len(all_seeds) - checked - len(todo)} unverified"

    if rank == 0:
        px_step = total_frames * W * H
        value = px_step * args.steps / dt / 1e6
        k_avg = sum(kernel_ms) / len(kernel_ms) / 1e3                      # seconds per launch (rank 0)
        hp, wp = imcvt_amd.padded(H), imcvt_amd.padded(W)
        ctus = F * (hp // 32) * (wp // 32)
        algo_bytes = F * (W * H + hp * wp) + int(sum(lens))                # per launch on one GPU
        achieved = algo_bytes / k_avg / 1e9
        traffic, traffic_src = None, "not collected in this run (rocprofv3 --pmc passes: tools/pmc_traffic.py)"
        tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")            # calibrated FETCH_SIZE / WRITE_SIZE of a counter run
        lib_hash = _lib_srchash()
        stale = lambda rec: "" if (lib_hash is not None and rec.get("lib_srchash") == lib_hash) else "STALE (counters were taken on a different build of the library than the one timed here) - "
        if os.path.exists(tp):
            t = json.load(open(tp))
            if t.get("qpd6") == args.qpd6 and t.get("hbm_bytes_per_launch") and t.get("ctus"):
                same = (t.get("frames"), t.get("w"), t.get("h")) == (F, W, H)
                traffic = int(t["hbm_bytes_per_launch"] * ctus / t["ctus"])
                what = f"{t.get('frames')} x {t.get('w')}x{t.get('h')} frames, {t['ctus']} CTUs, {t['hbm_bytes_per_launch']} B per launch"
                traffic_src = stale(t) + (f"calibrated FETCH_SIZE + WRITE_SIZE of separate rocprofv3 --pmc passes over this same workload ({what}; {tp[len(ROOT) + 1:]}); not collected in this run"
                               if same else f"extrapolated by CTU count from a counter run over {what} ({tp[len(ROOT) + 1:]}); not measured on this workload")
        macs = 12320 * hp * wp * F                                          # transform MACs per launch
        mode = f"{'strong' if strong else 'weak'}: {total_frames} frames over {world} GPU(s), {F} on rank 0"
        line = {
            "metric": "Mpixels/s HEVC intra encode (gray8), bit-exact vs CPU" if checked else "Mpixels/s HEVC intra encode (gray8)",
            "value": round(value, 3), "unit": "Mpixels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "i32 (u8 pixels, i8 transform matrices, i32 accumulate)",
            "data": "synthetic syn(1920,1080,seed), SURVEY App. C",
            "config": {"workload": f"BASELINE configs[3]: batch of {total_frames} independent 1920x1080 gray8 frames -> .h265, qpd6={args.qpd6}, frame-sharded ({mode})",
                       "global_frames": total_frames, "frames_rank0": F, "width": W, "height": H, "qpd6": args.qpd6,
                       "parallelism": f"frames x{world}; streams gathered to rank 0 over RCCL (one batch of isend/irecv)" if use_dist else "frames x1 (single GPU, no collective)",
                       "rccl_ranks_observed": ranks_seen, "frames_per_rank": frames_per_rank,
                       "stream_bytes_per_frame": int(sum(lens) / max(1, F)), "verified": verified,
                       "step_to_step": "all frames digest-equal between the last warm-up step and the last timed step" if first is not None else "not checked (no warm-up)",
                       "encoder": imcvt_amd.load_library().imcvt_hevc_version().decode()},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 4), "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 8),
                         "traffic": traffic, "traffic_source": traffic_src, "kernel": "hevc_encode_frames", "kernel_ms": round(k_avg * 1e3, 2), "algorithmic_bytes": algo_bytes,
                         "kernel_ms_per_step": [round(v, 1) for v in kernel_ms], "resident_and_start_spread_us_per_step": launch_diag},
            "compute_view": {"transform_GMAC_per_launch": round(macs / 1e9, 1), "achieved_TMAC_s": round(macs / k_avg / 1e12, 4),
                             "note": "path is integer-ALU / serial-CABAC bound, not HBM bound (SURVEY F6, DESIGN.md §5)"},
        }
        ip = os.path.join(ROOT, "profiles", "pmc_issue.json")             # SQ counter passes (tools/gpu_prof.sh): what actually bounds the kernel
        if os.path.exists(ip):
            iv = json.load(open(ip))
            insts = iv["valu_wave_insts_per_ctu"] * ctus
            nominal = 256 * 4 * 2.4e9 / 2                                  # wave-instructions/s at 2 cycles per wave64 instruction (SIMD-32), the guide's figure for v_fma_f32
            # what a SIMD sustains on THIS kernel's opcode mix, measured (tools/valu_mix_probe.py: four waves per SIMD on all 256 compute units, every opcode
            # that makes up >= 0.4 % of the kernel's VALU instructions; adds / logic / right shifts 2.26 cycles, selects / compares / left shifts / bit-field /
            # 3-operand / multiplies 4.1 - 4.4): the peak the issue fraction is taken against
            mp = os.path.join(ROOT, "profiles", "valu_mix.json")
            dp = os.path.join(ROOT, "profiles", "valu_dyn_mix.json")          # round 6: the DYNAMIC opcode mix (tools/valu_dyn_mix.py: region execution counters x static histograms)
            peak, peak_src = nominal, "nominal 2 cycles per wave64 VALU instruction (no measured mix in profiles/valu_mix.json)"
            if os.path.exists(dp) and os.path.exists(mp):
                dv = json.load(open(dp))
                peak = 256 * 4 * dv["clock_ghz"] * 1e9 / dv["mix_weighted_cycles_simd"]
                peak_src = (f"dynamic: {dv['mix_weighted_cycles_simd']} cycles per wave64 VALU instruction on the kernel's DYNAMIC opcode mix — static opcode histograms of the marked regions x their "
                            f"executions per CTU from a region-counter build at this launch shape ({round(100 * dv['share_in_marked_regions'], 1)} % of the measured VALU instructions lie in marked regions, "
                            f"{round(100 * dv['share_covered_by_opcode_table'], 1)} % of the dynamic instructions are opcodes with a measured cost), profiles/valu_dyn_mix.json (tools/valu_dyn_mix.py); "
                            f"per-opcode costs from profiles/valu_mix.json (four wavefronts per SIMD on all 256 compute units); the static mix of round 5 gave {dv['static_mix_weighted_cycles_simd']}")
            elif os.path.exists(mp):
                mv = json.load(open(mp))
                peak = 256 * 4 * mv["clock_ghz"] * 1e9 / mv["mix_weighted_cycles_simd"]
                peak_src = (f"measured: {mv['mix_weighted_cycles_simd']} cycles per wave64 VALU instruction on the kernel's static opcode mix ({round(100 * mv['share_covered'], 1)} % of its VALU "
                            f"instructions covered), profiles/valu_mix.json (tools/valu_mix_probe.py); a lone wavefront issues one per {mv['mix_weighted_cycles_lone_wave']} cycles")
            line["roofline_issue"] = {"bound": "valu issue", "achieved": round(insts / k_avg / 1e9, 2), "peak": round(peak / 1e9, 1), "unit": "G wave-inst/s",
                                      "frac": round(insts / k_avg / peak, 4), "peak_source": peak_src,
                                      "peak_nominal_2_cycles": round(nominal / 1e9, 1), "frac_of_nominal": round(insts / k_avg / nominal, 4),
                                      "valu_wave_insts_per_ctu": iv["valu_wave_insts_per_ctu"], "lane_activity": iv.get("valu_lane_activity"),
                                      "valu_busy_frac_in_counter_run": iv["valu_busy_frac"], "source": stale(iv) + "instruction count per CTU and lane activity from " + iv["source"] + "; time from this run"}
        if world == 1 and not args.no_latency_view:
            line["latency_view"] = latency_view(enc, dev, args.qpd6)
            line["qpd6_4_view"] = qpd6_view(enc, big, 4) if args.qpd6 != 4 else None
            line["solo_1000f"] = solo_view(enc, big, last, args.qpd6)
            line["host_abi_view"] = hv = host_abi_view(big, last, args.qpd6, gold={i: gold[s] for i, s in enumerate(seeds) if s in gold})
            # wall time of the reference-shaped entry point, H2D and D2H included, beside the kernel-only figure
            line["t_wall_incl_transfers"] = {"ms": hv["wall_ms"], "mpx_s": hv["mpx_s"], "frac_of_value": round(hv["mpx_s"] / value, 4) if hv["frames"] == F else None,
                                             "what": f"HEVCImageEncoderBatch over the {hv['frames']} bench frames from and to pageable host memory, one call (host_abi_view)"}
            line["job_seconds_by_share"] = share_view(enc, big, args.qpd6, k_avg)
            line["jls_view"] = jls_view(dev)
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(args.qpd6)
        print(json.dumps(line), flush=True)
    enc.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def solo_view(enc, big, digests, qpd6, n=1000):
    """The same kernel with the device full: n frames (the bench frames, repeated), a frame per workgroup — the throughput the
    path reaches when the batch is large enough to fill every workgroup slot.  One warm launch, one timed; kernel ms from HIP events."""
    import torch
    F = big.shape[0]
    if F < 1:
        return None
    b = enc.make_batch([big[i % F] for i in range(n)], qpd6)
    enc.set_team(1)
    enc.encode(b); torch.cuda.synchronize()
    # A launch that fills 97 % of the workgroup slots sporadically starts ONE workgroup seconds late (999 of 1000 resident, that frame then runs
    # after the first workgroup has left — 14.3 s instead of 9.4 s; 3 launches of 8 on one box in round 6, profiles/r06c_ab.log).  Up to three timed launches: the
    # line reports every one with its residency, `kernel_ms` is the best.
    runs = []
    for _ in range(3):
        enc.encode(b); torch.cuda.synchronize()
        runs.append({"kernel_ms": round(enc.last_kernel_ms(), 1), "resident_at_once": enc.last_resident(), "start_spread_us": enc.last_start_spread_us()})
        if runs[-1]["resident_at_once"] >= n:
            break
    ms = min(r["kernel_ms"] for r in runs)
    enc.set_team(0)
    lens = b["lens"].cpu().tolist()
    same = all(hashlib.sha256(b["outs"][i][:lens[i]].cpu().numpy().tobytes()).hexdigest() == digests[i % F] for i in list(range(0, n, 37)) + [n - 1])
    if not same:
        raise SystemExit("solo_1000f: streams differ from the timed batch's")
    return {"frames": n, "kernel_ms": round(ms, 1), "mpx_s": round(n * W * H / ms / 1e3, 3), "shape": list(enc.last_shape()),
            "timed_launches": runs, "streams_equal_to_timed_batch": f"{len(range(0, n, 37)) + 1} sampled"}


def share_view(enc, big, qpd6, k512):
    """One GPU's share of BASELINE configs[3] at N = 1 / 2 / 4 / 8 as seconds of kernel time (the first 512 / 256 / 128 / 64 bench frames in one
    launch each): what the strong-scaling job costs per N, read in seconds rather than as a ratio."""
    import torch
    F = big.shape[0]
    out = {str(F): {"kernel_s": round(k512, 3), "what": "the timed batch"}}
    for n in (256, 128, 64):
        if n >= F:
            continue
        b = enc.make_batch([big[i] for i in range(n)], qpd6)
        enc.encode(b); torch.cuda.synchronize()
        out[str(n)] = {"kernel_s": round(enc.last_kernel_ms() / 1e3, 3), "shape": list(enc.last_shape()), "pipe_wave": enc.last_pipe(), "wide_workgroups": enc.last_wide()}
        del b
    return out


def qpd6_view(enc, big, q):
    """The bench batch at the other end of the quantiser range (SURVEY section 8d asks for qpd6 = 0 and 4): same frames, same launch shape,
    one warm launch and one timed; frame 0 against the reference's digest (tests/golden/hevc_kat.json holds syn(1920,1080,0) at qpd6 4)."""
    import torch
    F = big.shape[0]
    if F < 1:
        return None
    b = enc.make_batch([big[i] for i in range(F)], q)
    enc.encode(b); torch.cuda.synchronize()
    enc.encode(b); torch.cuda.synchronize()
    ms = enc.last_kernel_ms()
    lens = b["lens"].cpu().tolist()
    e = next((e for e in json.load(open(os.path.join(ROOT, "tests", "golden", "hevc_kat.json")))
              if e["input"] == {"kind": "syn", "w": W, "h": H, "arg": 0} and e["qpd6"] == q), None)
    okd = (e["bytes"] == lens[0] and hashlib.sha256(b["outs"][0][:lens[0]].cpu().numpy().tobytes()).hexdigest() == e["sha256"]) if e else None
    if okd is False:
        raise SystemExit(f"qpd6_{q}_view: frame 0 differs from the reference digest")
    return {"qpd6": q, "frames": F, "kernel_ms": round(ms, 1), "mpx_s": round(F * W * H / ms / 1e3, 3), "shape": list(enc.last_shape()),
            "stream_bytes_per_frame": int(sum(lens) / F), "frame0_sha256_equal_to_reference": okd}


def jls_view(dev):
    """BASELINE config 5: 1920x1080 gray8 -> .jls (NEAR=0, src/imageio_jls.c:240-399) on the device: one plane alone and 64 planes
    in one launch (kernel ms from HIP events, inputs resident), digest against the reference's (tests/golden/jls_kat.json), and the
    CPU checker's one-core time for the same plane beside it."""
    import torch
    from imcvt_amd import jls, synth
    img = synth.syn(W, H, 0)
    gold = next((e for e in json.load(open(os.path.join(ROOT, "tests", "golden", "jls_kat.json")))
                 if e["input"] == {"kind": "syn", "w": W, "h": H, "arg": 0} and e["near"] == 0), None)
    out = {}
    for name, n in (("1_plane", 1), ("64_planes", 64)):
        d = jls.DevicePlanes([torch.from_numpy(img).to(dev) for _ in range(n)], 0)
        ms = []
        for _ in range(3):
            d.encode(); torch.cuda.synchronize(); ms.append(d.last_kernel_ms())
        res = d.results()
        okd = all(len(r) == gold["bytes"] and hashlib.sha256(r).hexdigest() == gold["sha256"] for r in (res[0], res[-1])) if gold else None
        if okd is False:
            raise SystemExit(f"jls_view {name}: stream differs from the reference digest")
        out[name] = {"kernel_ms": round(min(ms), 2), "mpx_s": round(n * W * H / min(ms) / 1e3, 2), "bytes_per_plane": len(res[0]), "sha256_equal_to_reference": okd,
                     "path": "planes spread over the device" if d.last_path() == 1 else "one walker per plane"}
    from oracle import oracle                                  # (checker leg, like cpu_baseline)
    t = time.perf_counter(); oracle.jls_cpu_encode(img, 0); cs = time.perf_counter() - t
    out["cpu_1core"] = {"seconds": round(cs, 3), "mpx_s": round(W * H / cs / 1e6, 2), "kind": "reference" if oracle.jls_have_ref() else "port"}
    # near-lossless (NEAR = 2): the neighbourhood of a pixel is made of RECONSTRUCTED samples, which depend on the coded errors and
    # through them on the adaptive context state in raster order (src/imageio_jls.c:325,361) — one plane is one serial chain, walked by
    # one lane; the device's answer is many planes at once.  Both ends, with the CPU checker's one-core time for the same plane.
    for name, n in (("near2_1_plane", 1), ("near2_64_planes", 64)):
        d = jls.DevicePlanes([torch.from_numpy(img).to(dev) for _ in range(n)], 2)
        d.encode(); torch.cuda.synchronize(); ms = d.last_kernel_ms()
        res = d.results()
        if name == "near2_1_plane":
            t = time.perf_counter(); want = oracle.jls_cpu_encode(img, 2); cs2 = time.perf_counter() - t
        if res[0] != want or res[-1] != want:
            raise SystemExit(f"jls_view {name}: stream differs from the CPU checker")
        out[name] = {"kernel_ms": round(ms, 1), "mpx_s": round(n * W * H / ms / 1e3, 2), "bytes_per_plane": len(res[0]), "bytes_equal_to_cpu_checker": True,
                     "path": "planes spread over the device" if d.last_path() == 1 else "one walker per plane"}
    out["near2_cpu_1core"] = {"seconds": round(cs2, 3), "mpx_s": round(W * H / cs2 / 1e6, 2)}
    return out


def host_abi_view(big, digests, qpd6, n=512, gold=None):
    """The reference-shaped entry point (HOST pointers in and out): wall time of HEVCImageEncoderBatch over the
    bench frames, PCIe copies in both directions, slab management and the launch included — next to `value`, which is measured
    with inputs resident in HBM (BASELINE.md §3 asks for both)."""
    import imcvt_amd
    n = min(n, big.shape[0])
    from imcvt_amd import hevc
    imgs = [big[i].cpu().numpy() for i in range(n)]
    imcvt_amd.HEVCImageEncoderBatch(imgs[:32], qpd6)             # creates the per-device context, slab and staging buffers outside the timed call
    # Three calls: a launch of this shape takes 4.8 s or 5.0 s with no regard to the path it came by (one in three, profiles/r06f_streams.log) — all are reported, the best one is the view's figure
    calls = []
    for _ in range(3):
        t0 = time.perf_counter()
        res = imcvt_amd.HEVCImageEncoderBatch(imgs, qpd6, copy=False)      # (streams as views of the caller-owned buffers, as a C caller has them)
        dt_ = time.perf_counter() - t0
        xs_ = hevc.transfer_stats()
        calls.append((dt_, xs_))
    dt, xs = min(calls, key=lambda c: c[0])
    same = all(hashlib.sha256(s).hexdigest() == d for (s, _, _), d in zip(res, digests[:n]))
    if not same:
        raise SystemExit("host_abi_view: host-pointer batch differs from the device-resident batch")
    if gold:                                                     # the reconstructions that came back over the link, against the reference's digests
        bad = [i for i in range(n) if i in gold and gold[i][2] and hashlib.sha256(res[i][1].tobytes()).hexdigest() != gold[i][2]]
        if bad:
            raise SystemExit(f"host_abi_view: {len(bad)} reconstructions differ from the reference digests (first: frame {bad[0]})")
    lib = imcvt_amd.load_library()
    out = {"frames": n, "wall_ms": round(dt * 1e3, 1), "mpx_s": round(n * W * H / dt / 1e6, 3), "devices": int(lib.imcvt_hevc_batch_devices()),
           "bytes_h2d": n * W * H, "bytes_d2h": int(sum(len(s) for s, _, _ in res)) + n * imcvt_amd.padded(H) * imcvt_amd.padded(W),
           "upload_ms": round(xs["upload_s"] * 1e3, 1), "followed_launch_ms": round(xs["follow_s"] * 1e3, 1), "tail_ms": round(xs["tail_s"] * 1e3, 1),
           "bytes_copied_out_while_the_launch_ran": int(xs["bytes_during"]), "bytes_copied_out_after_it": int(xs["bytes_after"]),
           "kernel_ms": round(xs["kernel_ms"], 1), "all_calls": [{"wall_ms": round(c[0] * 1e3, 1), "kernel_ms": round(c[1]["kernel_ms"], 1)} for c in calls],
           "streams_equal_to_resident_run": True, "reconstructions_equal_to_reference_digests": bool(gold)}
    lib.imcvt_hevc_shutdown()
    return out


def latency_view(enc, dev, qpd6):
    """BASELINE configs 1 and 2: ONE frame alone on the GPU (kernel ms from HIP events, inputs resident)."""
    import torch
    from imcvt_amd import synth
    out = {}
    gold = {(e["input"].get("w"), e["input"].get("arg"), e["qpd6"]): e for e in json.load(open(os.path.join(ROOT, "tests", "golden", "hevc_kat.json"))) if e["input"].get("kind") == "syn"}
    for name, (w, h) in (("1080p", (1920, 1080)), ("4k", (3840, 2160))):
        img = torch.from_numpy(synth.syn(w, h, 0)).to(dev)
        b = enc.make_batch([img], qpd6)
        enc.encode(b)
        ms = enc.last_kernel_ms()
        n = int(b["lens"][0].item())
        e = gold.get((w, 0, qpd6))
        okd = (hashlib.sha256(b["outs"][0][:n].cpu().numpy().tobytes()).hexdigest() == e["sha256"]) if e else None
        if okd is False:
            raise SystemExit(f"latency_view {name}: stream differs from the reference digest")
        out[name] = {"kernel_ms": round(ms, 1), "mpx_s": round(w * h / ms / 1e3, 3), "bytes": n, "sha256_equal_to_reference": okd,
                     "shape": list(enc.last_shape()), "pipe_wave": enc.last_pipe(), "wide_workgroups": enc.last_wide()}
    # the same 1080p frame with 256-thread workgroups (round 4's latency shape), and the per-GPU share of the 512-frame job at N = 8 (64 frames)
    img = torch.from_numpy(synth.syn(1920, 1080, 0)).to(dev)
    b = enc.make_batch([img], qpd6)
    enc.set_wide(0); enc.encode(b); out["1080p_256_thread_workgroups"] = {"kernel_ms": round(enc.last_kernel_ms(), 1), "wide_workgroups": enc.last_wide()}; enc.set_wide(-1)
    imgs64 = [torch.from_numpy(synth.syn(1920, 1080, s)).to(dev) for s in range(64)]
    b64 = enc.make_batch(imgs64, qpd6)
    enc.encode(b64)
    ms64 = enc.last_kernel_ms()
    out["64_frames"] = {"what": "64 x 1080p in one launch = one GPU's share of BASELINE configs[3] at N = 8", "kernel_ms": round(ms64, 1), "mpx_s": round(64 * 1920 * 1080 / ms64 / 1e3, 2),
                        "shape": list(enc.last_shape()), "pipe_wave": enc.last_pipe(), "wide_workgroups": enc.last_wide()}
    del b64, imgs64
    # BASELINE configs[0]: the reference's own sample picture (image/P4.pnm, 21 x 17 raw PBM -> one 32 x 32 CTU; its pixels as gray8 are tests/golden/p4_gray.pgm) through the
    # reference-shaped FILE entry point, writeHEVCImageFile (src/imageio_hevc.c:9): host buffers, PCIe, the launch and the file write included
    import tempfile
    import numpy as np
    import imcvt_amd
    data = open(os.path.join(ROOT, "tests", "golden", "p4_gray.pgm"), "rb").read().split(b"\n", 3)
    pw, ph = map(int, data[1].split())
    px = np.frombuffer(data[3], dtype=np.uint8, count=pw * ph).reshape(ph, pw)
    want = open(os.path.join(ROOT, "tests", "golden", "p4_q0.h265"), "rb").read()
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "p4.h265")
        imcvt_amd.writeHEVCImageFile(path, px, False, ph, pw, 0)             # (first call: per-device context and slab)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); rc = imcvt_amd.writeHEVCImageFile(path, px, False, ph, pw, 0); ts.append(time.perf_counter() - t0)
        got = open(path, "rb").read()
    if rc != 0 or got != want:
        raise SystemExit("latency_view p4: writeHEVCImageFile output differs from the reference's stream")
    out["p4"] = {"what": "BASELINE configs[0]: the reference's 21x17 sample picture image/P4.pnm (one CTU) -> .h265 through writeHEVCImageFile (host buffers, file write), qpd6 0",
                 "wall_ms": round(min(ts) * 1e3, 2), "mpx_s": round(pw * ph / min(ts) / 1e6, 3), "bytes": len(got), "bytes_equal_to_reference": True}
    imcvt_amd.load_library().imcvt_hevc_shutdown()
    return out


if __name__ == "__main__":
    main()
