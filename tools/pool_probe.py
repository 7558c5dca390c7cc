#!/usr/bin/env python3
"""[developer check script — TEST INFRASTRUCTURE like tests/] Launch-shape probe: n frames of w x h at qpd6 q under a list of
(mains, helpers) shapes; kernel ms per shape, digests compared between shapes (and with the first shape's).
usage: pool_probe.py w h n q  m:h[:lim16:lim32:prio[:post16:post32]] ...      (0:0 = frames per workgroup, a:a = automatic)"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, imcvt_amd
from imcvt_amd import synth
w, h, n, q = (int(a) for a in sys.argv[1:5])
enc = imcvt_amd.DeviceEncoder()
frames = [torch.from_numpy(synth.syn(w, h, s)).cuda() for s in range(n)]
batch = enc.make_batch(frames, q)
fclk = torch.zeros(4 * n + 4 * 1024, dtype=torch.int64, device="cuda")
enc.lib.imcvt_hevc_set_frame_clock(enc.ctx, fclk.data_ptr())
ref = None
for sh in sys.argv[5:]:
    if sh == "a:a":
        enc.set_shape(0, 0); enc.set_team(0)
    else:
        f = [int(v) for v in sh.split(":")]
        m, hp = f[0], f[1]
        enc.set_pool_tuning(*(f[2:5] if len(f) >= 5 else (-1, -1, -1)))
        enc.set_pool_split(*(f[5:7] if len(f) >= 7 else (-1, -1)))
        if m == 0:
            enc.set_shape(0, 0); enc.set_team(1)
        else:
            enc.set_team(0); enc.set_shape(m, hp)
    ms = []; resid = []
    for _ in range(int(os.environ.get("PP_LAUNCHES", "2"))):
        enc.encode(batch); torch.cuda.synchronize(); ms.append(enc.last_kernel_ms()); resid.append((enc.last_resident(), enc.last_start_spread_us()))
        if os.environ.get("PP_VERBOSE"):
            allc = fclk.cpu().numpy(); fc = allc[:4 * n].reshape(n, 4).copy(); wg = allc[4 * n:].reshape(1024, 4); t0 = fc[:, 0].min()
            gaps = [(int(wg[b, 0]) / 1e5, (int(wg[b, 1]) - int(t0)) / 1e5, int(wg[b, 2]) & 0xFFFF, int(wg[b, 2]) >> 32, b) for b in range(1024) if wg[b, 0] > 2e6]
            if gaps: print('      workgroups with a heartbeat gap > 20 ms (gap ms, began at ms, cu key, main?, block): ' + ' '.join(f'({g[0]:.0f},{g[1]:.0f},{g[2]:#x},{g[3]},{g[4]})' for g in sorted(gaps, reverse=True)[:24]), flush=True)
            end = (fc[:, 1] - t0) / 1e5; start = (fc[:, 0] - t0) / 1e5; late = int(end.argmax())
            import collections
            keys = collections.Counter(((fc[:, 2] >> 32) & 0xFFFF).tolist())
            print(f"      main workgroups ran on {len(keys)} compute units; frames per compute unit: {sorted(collections.Counter(keys.values()).items())}", flush=True)
            cuk = (fc[:, 2] >> 32) & 0xFFFF
            raised = (fc[:, 2] >> 48) & 0xFFFF
            if os.environ.get("PP_OUTLIER"):
                import collections as _c3
                hs = [(int(wg[b, 0]) & 0xFFFFFFFF, (int(wg[b, 1]) >> 56) & 0x7F, b) for b in range(1024) if wg[b, 3] != 0 and not (int(wg[b, 2]) >> 32) & 1 and (int(wg[b, 1]) >> 56) & 0x80]
                by = _c3.defaultdict(list)
                for sv, arr, b in hs: by[arr].append(sv)
                print("      helpers: requests served by arrival index on the compute unit: " + "; ".join(f"arrival {a}: {len(v)} helpers, mean {sum(v) / len(v):.0f} min {min(v)} max {max(v)}" for a, v in sorted(by.items())), flush=True)
                by2 = _c3.defaultdict(list)
                for sv, arr, b in hs: by2[b // 64].append(sv)
                print("      helpers: mean requests served by block / 64: " + " ".join(f"{k}:{sum(v) / len(v):.0f}" for k, v in sorted(by2.items())), flush=True)
                mr = _c3.defaultdict(list)
                for i in range(n): mr[int(fc[i, 2] & 0xFFFFFFFF) // 64].append((float(end[i]) if False else 0, int(raised[i])))
                print("      mains: CTUs at raised priority, mean by block / 64: " + " ".join(f"{k}:{sum(r for _, r in v) / len(v):.0f}" for k, v in sorted(mr.items())), flush=True)
            if os.environ.get("PP_OUTLIER") and ms[-1] > 1.12 * min(ms + [float(os.environ.get("PP_BASE_MS", "1e9"))]):
                odd = [k for k, v in keys.items() if v != 2]
                print("      OUTLIER: compute units whose number of frames is not 2: " + "; ".join(f"cu {k:#x}: " + ", ".join(f"f{i} blk {int(fc[i, 2] & 0xFFFFFFFF)} start {(fc[i, 0] - t0) / 1e5:.1f} end {(fc[i, 1] - t0) / 1e5:.0f} kept {int(fc[i, 3] & 0xFFFF)}" for i in range(n) if cuk[i] == k) for k in odd))
                med = float(sorted(end)[n // 2])
                slow = [i for i in range(n) if end[i] > 1.1 * med]
                print(f"      OUTLIER: {len(slow)} frames end later than 1.1 x median: " + "; ".join(f"f{i} cu {int(cuk[i]):#x} blk {int(fc[i, 2] & 0xFFFFFFFF)} start {start[i]:.1f} end {end[i]:.0f} kept {int(fc[i, 3] & 0xFFFF)}" for i in slow[:24]))
                simd = lambda b: "".join(str((int(wg[b, 1]) >> (8 * k)) & 3) for k in range(3) if (int(wg[b, 1]) >> (8 * k)) & 0x80) + (f" arrival {(int(wg[b, 1]) >> 56) & 0x7F}" if (int(wg[b, 1]) >> 56) & 0x80 else "")
                print("      OUTLIER: SIMDs of the wavefronts of the slow frames' workgroups: " + "; ".join(f"f{i} blk {int(fc[i, 2] & 0xFFFFFFFF)} end {end[i]:.0f} raised {int(raised[i])}: {simd(int(fc[i, 2] & 0xFFFFFFFF))}" for i in sorted(slow, key=lambda i: -end[i])[:16]))
                import collections as _c2
                allm = _c2.Counter(simd(int(fc[i, 2] & 0xFFFFFFFF))[3:] for i in range(n))
                print("      OUTLIER: arrival index on the compute unit, all main workgroups: " + str(sorted(allm.items(), key=lambda kv: -kv[1])[:12]))
                slowp = _c2.Counter(simd(int(fc[i, 2] & 0xFFFFFFFF))[3:] for i in slow)
                print("      OUTLIER: ... of the slow ones: " + str(sorted(slowp.items(), key=lambda kv: -kv[1])[:12]))
                wgs = [(int(wg[b, 2]) & 0xFFFF, int(wg[b, 2]) >> 32, b, (int(wg[b, 3]) - int(t0)) / 1e5) for b in range(1024) if wg[b, 3] != 0]
                import collections as _c
                per = _c.Counter(k for k, _, _, _ in wgs)
                print("      OUTLIER: workgroups per compute unit (count: units) " + str(sorted(_c.Counter(per.values()).items())) + "; units with < 3: " + "; ".join(f"cu {k:#x}: " + ", ".join(f"blk {b} main {m} left {e:.0f}" for kk, m, b, e in wgs if kk == k) for k, v in per.items() if v < 3), flush=True)
            fc[:, 2] &= 0xFFFFFFFF
            waited = ((fc[:, 3] >> 16) & 0xFFFFFF) / 100.0; wmax = (fc[:, 3] >> 40) / 100.0; fc[:, 3] &= 0xFFFF
            top = sorted(range(n), key=lambda i: -((fc[i, 1] - t0)))[:4]
            print("      slowest frames: " + "; ".join(f"f{i} blk {fc[i, 2]} end {(fc[i, 1] - t0) / 1e5:.0f} ms waited {waited[i]:.0f} ms (longest {wmax[i]:.1f})" for i in top) + f" | median waited {sorted(waited)[n // 2]:.0f} ms, longest wait anywhere {wmax.max():.1f} ms", flush=True)
            print(f"      launch {ms[-1]:.1f} ms resident/start-spread-us {resid[-1]}  frame ends ms: median {float(sorted(end)[n // 2]):.0f} max {end.max():.0f} (frame {late}: block {fc[late, 2]}, started {start[late]:.0f}, kept {fc[late, 3]}); started late (>100 ms): {int((start > 100).sum())}; kept per frame median {sorted(fc[:, 3])[n // 2]} max {fc[:, 3].max()}", flush=True)
        if os.environ.get("PP_PROF"):                      # -DIMCVT_PROF build: G-cycles per role summed over the launch's waves
            pr = enc.debug_prof(True); cats = enc.PROF_CATS
            pick = ("sync", "wait_help", "idle", "p1_4", "p2_8", "p2_nxn")
            print("      prof " + "  ".join(f"{'MH'[r]}:" + ",".join(f"{c}={sum(pr[3 * r + w][cats.index(c)] for w in range(3)) / 1e9:.1f}" for c in pick) for r in range(2)), flush=True)
    dig = hashlib.sha256(b"".join(s for s, _ in enc.results(batch))).hexdigest()[:16]
    if ref is None:
        ref = dig
    print(f"{n} x {w}x{h} q{q} shape {sh:>18s} -> {enc.last_shape()}{' + pipe wave' if enc.last_pipe() else ''}: kernel ms {[round(v, 1) for v in ms]} resident {resid}  {w * h * n / min(ms) / 1e3:7.2f} Mpx/s  digest {dig} {'same' if dig == ref else 'DIFFERENT'}", flush=True)
