#!/usr/bin/env python3
"""Digests of BASELINE configs[3]'s whole workload from the REAL reference: syn(1920,1080,s), qpd6=0, s = 0..511
(the frames bench.py encodes).  Dev container only (needs oracle/_ref, built from /root/reference by oracle/Makefile);
the output tests/golden/bench512_kat.json is committed data: seed -> stream length, stream SHA-256, reconstruction SHA-256.

~55 s of one core per frame: `nice python tests/golden/make_bench_golden.py 5` takes about 1.6 h on 5 cores.
Resumable: seeds already in the JSON are skipped; the file is rewritten every 16 frames.
"""
import hashlib
import json
import os
import sys
from multiprocessing import Pool

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
OUT = os.path.join(HERE, "bench512_kat.json")
W, H, Q, N = 1920, 1080, 0, 512


def run_one(seed):
    from oracle import oracle, synth
    img = synth.syn(W, H, seed)
    stream, rcon, _ = oracle.ref_encode(img, Q)
    return seed, dict(bytes=len(stream), sha256=hashlib.sha256(stream).hexdigest(), rcon_sha256=hashlib.sha256(rcon.tobytes()).hexdigest())


def save(done):
    doc = dict(generator="tests/golden/make_bench_golden.py", source="oracle/_ref/libref_hevce.so (reference src/HEVCe/HEVCe.c compiled unmodified)",
               input=dict(kind="syn", w=W, h=H), qpd6=Q, frames={str(k): done[k] for k in sorted(done)})
    with open(OUT + ".tmp", "w") as f:
        json.dump(doc, f, indent=0)
    os.replace(OUT + ".tmp", OUT)


def main():
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else max(1, (os.cpu_count() or 2) - 2)
    done = {}
    if os.path.exists(OUT):
        done = {int(k): v for k, v in json.load(open(OUT))["frames"].items()}
    todo = [s for s in range(N) if s not in done]
    print(f"{len(done)} done, {len(todo)} to do on {workers} workers", flush=True)
    with Pool(workers) as pool:
        for i, (seed, rec) in enumerate(pool.imap_unordered(run_one, todo, chunksize=1)):
            done[seed] = rec
            if (i + 1) % 16 == 0 or i + 1 == len(todo):
                save(done)
                print(f"{len(done)}/{N}", flush=True)
    save(done)


if __name__ == "__main__":
    main()
