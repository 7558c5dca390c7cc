#!/usr/bin/env python3
"""[developer check script — TEST INFRASTRUCTURE like tests/: uses the CPU checker under oracle/ to verify what it times] JPEG-LS throughput probe (BASELINE config 5): n gray planes syn(w,h,seed), NEAR=near, device-resident, one launch.
Prints kernel times, Mpx/s, and checks the first planes against the CPU checker / golden digest."""
import hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from imcvt_amd import jls
from oracle import oracle, synth
w, h, n, near = (int(a) for a in (sys.argv[1:5] + ["1920", "1080", "1024", "0"][len(sys.argv) - 1:]))
imgs = [synth.syn(w, h, s) for s in range(min(n, 8))]
planes = [torch.from_numpy(imgs[s % len(imgs)]).cuda() for s in range(n)]
d = jls.DevicePlanes(planes, near)
ms = []
for _ in range(3):
    d.encode(); torch.cuda.synchronize(); ms.append(d.last_kernel_ms())
res = d.results()
t = time.perf_counter(); want = oracle.jls_cpu_encode(imgs[0], near); cpu_s = time.perf_counter() - t
ok = res[0] == want and (n < 2 or res[1] == oracle.jls_cpu_encode(imgs[1 % len(imgs)], near))
best = min(ms)
print(json.dumps({"workload": f"{n} x {w}x{h} gray8 -> .jls NEAR={near}", "kernel_ms": [round(m, 2) for m in ms], "Mpx_s": round(w * h * n / best / 1e3, 2),
                  "bytes_plane0": len(res[0]), "sha256_plane0": hashlib.sha256(res[0]).hexdigest(), "parity": "OK" if ok else "MISMATCH",
                  "cpu_checker_1_plane_s": round(cpu_s, 3), "cpu_checker_Mpx_s_1core": round(w * h / cpu_s / 1e6, 2)}))
sys.exit(0 if ok else 1)
