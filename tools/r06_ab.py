#!/usr/bin/env python3
"""Round-6 A/B probe on one box (interleaved launches, kernel ms from HIP events):
  wide   one 1080p frame and 64 frames: wide launches on hevc_wide.hip's instantiation (256 registers per wavefront) vs the common one
  split  128 frames: one launch (128 + 128 wide workgroups) vs two cooperating launches (128 wide mains + 192-thread helpers, 2 / 3 / 4 per compute unit)
usage: python tools/r06_ab.py [wide] [split] [--reps 3]"""
import hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import imcvt_amd
from imcvt_amd import synth

reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 3
what = [a for a in sys.argv[1:] if not a.startswith("-") and not a.isdigit()] or ["wide", "split", "partners", "follow", "solo"]
dev = torch.device("cuda", 0)
gold = json.load(open(os.path.join(ROOT, "tests", "golden", "bench512_kat.json")))["frames"]


def frames(n):
    return [torch.from_numpy(synth.syn(1920, 1080, s)).to(dev) for s in range(n)]


def check(b, n):
    lens = b["lens"].cpu().tolist()
    for i in (0, n // 2, n - 1):
        e = gold[str(i)]
        assert lens[i] == e["bytes"] and hashlib.sha256(b["outs"][i][:lens[i]].cpu().numpy().tobytes()).hexdigest() == e["sha256"], i


if "wide" in what:
    encs = {}
    for knob in ("1", "0"):
        os.environ["IMCVT_HEVC_WIDE_KERNEL"] = knob
        encs[knob] = imcvt_amd.DeviceEncoder()
    del os.environ["IMCVT_HEVC_WIDE_KERNEL"]
    for n in (1, 64):
        imgs = frames(n)
        bs = {k: e.make_batch(imgs, 0) for k, e in encs.items()}
        res = {k: [] for k in encs}
        for r in range(reps):
            for k, e in encs.items():
                e.encode(bs[k]); torch.cuda.synchronize(); res[k].append(round(e.last_kernel_ms(), 1))
        for k in encs:
            check(bs[k], n)
        print(json.dumps({"probe": "wide_kernel", "frames": n, "wide_instantiation_ms": res["1"], "common_instantiation_ms": res["0"], "shape": list(encs["1"].last_shape()), "wide": encs["1"].last_wide()}), flush=True)
        del bs, imgs
    for e in encs.values():
        e.close()

if "partners" in what:
    enc = imcvt_amd.DeviceEncoder()
    for n in (1, 64):
        imgs = frames(n)
        b = enc.make_batch(imgs, 0)
        res = {}
        for r in range(reps):
            for mode in (1, 0):
                enc.set_partners(mode)
                enc.encode(b); torch.cuda.synchronize()
                res.setdefault(mode, []).append((round(enc.last_kernel_ms(), 1), enc.last_partners(), enc.last_shape()))
                check(b, n)
        print(json.dumps({"probe": "partner_workgroups", "frames": n, "with_partners": {"ms": [v[0] for v in res[1]], "partners": res[1][0][1], "shape": list(res[1][0][2])},
                          "without": {"ms": [v[0] for v in res[0]], "partners": res[0][0][1], "shape": list(res[0][0][2])}}), flush=True)
        del b, imgs
    enc.set_partners(1)
    img4k = [torch.from_numpy(synth.syn(3840, 2160, 0)).to(dev)]
    b = enc.make_batch(img4k, 0)
    enc.encode(b); torch.cuda.synchronize()
    print(json.dumps({"probe": "partner_workgroups", "frames": "one 4K frame", "ms": round(enc.last_kernel_ms(), 1), "partners": enc.last_partners()}), flush=True)
    enc.close()

if "follow" in what:
    from imcvt_amd import hevc
    n = 512
    imgs = [synth.syn(1920, 1080, s) for s in range(n)]
    hevc.HEVCImageEncoderBatch(imgs[:32], 0)
    for mode in ("3", "1", "2", "0", "3"):
        os.environ["IMCVT_HEVC_FOLLOW"] = mode
        t0 = time.perf_counter()
        res = hevc.HEVCImageEncoderBatch(imgs, 0, copy=False)
        dt = time.perf_counter() - t0
        xs = hevc.transfer_stats()
        e = gold["511"]
        assert len(res[511][0]) == e["bytes"] and hashlib.sha256(res[511][0]).hexdigest() == e["sha256"] and hashlib.sha256(res[511][1].tobytes()).hexdigest() == e["rcon_sha256"]
        print(json.dumps({"probe": "host_follow", "IMCVT_HEVC_FOLLOW": mode, "wall_ms": round(dt * 1e3, 1), **{k: (round(v * 1e3, 1) if k.endswith("_s") else round(v, 1) if k == "kernel_ms" else int(v)) for k, v in xs.items()}}), flush=True)
        del res
    del os.environ["IMCVT_HEVC_FOLLOW"]
    imcvt_amd.load_library().imcvt_hevc_shutdown()
    # the same frames resident, same process: the kernel alone
    enc = imcvt_amd.DeviceEncoder()
    b = enc.make_batch([torch.from_numpy(a).to(dev) for a in imgs], 0)
    ks = []
    for r in range(3):
        enc.encode(b); torch.cuda.synchronize(); ks.append(round(enc.last_kernel_ms(), 1))
    print(json.dumps({"probe": "host_follow", "resident_kernel_ms_same_process": ks}), flush=True)
    enc.close(); del b

if "streams" in what:
    # the same 512 frames: resident on the null stream, resident on a stream of the caller's, and through the host-pointer path (its own stream), interleaved
    from imcvt_amd import hevc
    n = 512
    host_imgs = [synth.syn(1920, 1080, s) for s in range(n)]
    hevc.HEVCImageEncoderBatch(host_imgs[:32], 0)
    enc = imcvt_amd.DeviceEncoder()
    b = enc.make_batch([torch.from_numpy(a).to(dev) for a in host_imgs], 0)
    st = torch.cuda.Stream()
    res = {"resident_null_stream": [], "resident_user_stream": [], "host_path_kernel": [], "host_path_wall": []}
    for r in range(reps):
        enc.encode(b); torch.cuda.synchronize(); res["resident_null_stream"].append(round(enc.last_kernel_ms(), 1))
        enc.encode(b, stream=st); torch.cuda.synchronize(); res["resident_user_stream"].append(round(enc.last_kernel_ms(), 1))
        t0 = time.perf_counter(); out = hevc.HEVCImageEncoderBatch(host_imgs, 0, copy=False); dt = time.perf_counter() - t0
        res["host_path_kernel"].append(round(hevc.transfer_stats()["kernel_ms"], 1)); res["host_path_wall"].append(round(dt * 1e3, 1))
        del out
    print(json.dumps({"probe": "streams", "frames": n, **res}), flush=True)
    enc.close(); imcvt_amd.load_library().imcvt_hevc_shutdown(); del b, host_imgs

if "layout" in what:
    # does the device-side layout of a batch matter?  the same 512 frames resident as (a) separate torch tensors (what bench.py times), (b) ONE slab, frame by frame
    # [img | out | rcon] (what the host-pointer path builds), (c) one slab, grouped [all imgs | all outs | all rcons]; interleaved launches
    from imcvt_amd.hevc import imcvt_hevc_frame, stream_bound, padded
    n = 512
    imgs = frames(n)
    enc = imcvt_amd.DeviceEncoder()
    ba = enc.make_batch(imgs, 0)
    al = lambda v: (v + 255) & ~255
    hp, wp, bound = padded(1080), padded(1920), stream_bound(1080, 1920)
    def slab_batch(grouped):
        per = al(1920 * 1080) + al(bound) + al(hp * wp)
        slab = torch.empty(per * n + 4096, dtype=torch.uint8, device=dev)
        lens = torch.zeros(n, dtype=torch.int32, device=dev)
        base = slab.data_ptr()
        arr = (imcvt_hevc_frame * n)()
        outs = []
        for i in range(n):
            if grouped:
                o_img = al(1920 * 1080) * i; o_out = al(1920 * 1080) * n + al(bound) * i; o_rc = (al(1920 * 1080) + al(bound)) * n + al(hp * wp) * i
            else:
                o_img = per * i; o_out = o_img + al(1920 * 1080); o_rc = o_out + al(bound)
            slab[o_img:o_img + 1920 * 1080] = imgs[i].reshape(-1)
            arr[i] = imcvt_hevc_frame(base + o_img, base + o_out, base + o_rc, lens.data_ptr() + 4 * i, 1080, 1920, 0)
            outs.append(slab[o_out:o_out + bound])
        return dict(n=n, frames=arr, imgs=imgs, outs=outs, rcons=[], lens=lens, slab=slab)
    bb, bc = slab_batch(False), slab_batch(True)
    res = {"separate_tensors": [], "slab_frame_by_frame": [], "slab_grouped": []}
    for r in range(reps):
        for name, b in (("separate_tensors", ba), ("slab_frame_by_frame", bb), ("slab_grouped", bc)):
            enc.encode(b); torch.cuda.synchronize(); res[name].append(round(enc.last_kernel_ms(), 1))
    for b in (ba, bb, bc):
        check(b, n)
    print(json.dumps({"probe": "device_layout", "frames": n, **res}), flush=True)
    enc.close(); del ba, bb, bc, imgs

if "solo" in what:
    for knob in ("0", "1", "0", "1"):
        os.environ["IMCVT_HEVC_WIDE_KERNEL"] = knob
        enc = imcvt_amd.DeviceEncoder()
        imgs = frames(40)
        b = enc.make_batch([imgs[i % 40] for i in range(1000)], 0)
        enc.set_team(1)
        out = []
        for r in range(2):
            enc.encode(b); torch.cuda.synchronize()
            out.append((round(enc.last_kernel_ms(), 1), enc.last_resident(), enc.last_start_spread_us()))
        print(json.dumps({"probe": "solo_1000f", "IMCVT_HEVC_WIDE_KERNEL": knob, "ms_resident_spread_us": out}), flush=True)
        enc.close(); del b, imgs
    del os.environ["IMCVT_HEVC_WIDE_KERNEL"]

if "split" in what:
    enc = imcvt_amd.DeviceEncoder()
    for n in (128, 112):
        imgs = frames(n)
        b = enc.make_batch(imgs, 0)
        res = {}
        for r in range(reps):
            for hpc in (0, 2, 3, 4):
                enc.set_split(1 if hpc else 0, hpc)
                enc.encode(b); torch.cuda.synchronize()
                res.setdefault(hpc, []).append((round(enc.last_kernel_ms(), 1), enc.last_split(), enc.last_shape()))
                check(b, n)
        print(json.dumps({"probe": "split_launch", "frames": n, **{("one_launch" if h == 0 else f"split_{h}_per_cu"): {"ms": [v[0] for v in res[h]], "split": res[h][0][1], "shape": list(res[h][0][2])} for h in res}}), flush=True)
        del b, imgs
    enc.close()
