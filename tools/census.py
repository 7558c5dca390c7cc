#!/usr/bin/env python3
"""Residency census of the encoder kernel: how many of `grid` workgroups are on the device at the same time."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import imcvt_amd
enc = imcvt_amd.DeviceEncoder()
for grid in [int(a) for a in sys.argv[1:]] or [256, 512, 768, 896, 960, 1008, 1024, 1100]:
    print(grid, [enc.lib.imcvt_hevc_debug_census(enc.ctx, grid) for _ in range(3)], flush=True)
