#!/usr/bin/env python3
"""[developer check script] n frames of 1080p at q0 with forced launch shapes (mains, helpers), wide workgroups: kernel ms per launch.
usage: shape_probe.py n mains:helpers[:post16] ..."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, imcvt_amd
from imcvt_amd import synth
n = int(sys.argv[1])
enc = imcvt_amd.DeviceEncoder()
frames = [torch.from_numpy(synth.syn(1920, 1080, s)).cuda() for s in range(n)]
batch = enc.make_batch(frames, 0)
for rep in range(2):
    for spec in sys.argv[2:]:
        p = [int(x) for x in spec.split(":")]
        enc.set_shape(p[0], p[1]); enc.set_wide(-1)
        enc.set_pool_split(p[2] if len(p) > 2 else -1, -1)
        enc.encode(batch); torch.cuda.synchronize()
        dig = hashlib.sha256(b"".join(s + r.tobytes() for s, r in enc.results(batch))).hexdigest()[:16]
        print(f"{n} frames shape {spec} -> ran {enc.last_shape()} wide {enc.last_wide()}: {enc.last_kernel_ms():.1f} ms  digest {dig}", flush=True)
