#!/usr/bin/env python3
"""First differing CU decisions between two traces written by tools/trace_dump.py."""
import sys, numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
n = min(len(a), len(b)); shown = 0
for i in range(n):
    if (a[i] != b[i]).any():
        print(i, "A", a[i][:6].tolist(), "B", b[i][:6].tolist()); shown += 1
        if shown >= 8: break
print("traces", len(a), len(b), "differing rows shown", shown)
