#!/usr/bin/env python3
"""Generates tests/golden/formats_kat.json: small PNG / BMP / QOI files and what the REFERENCE converter
(oracle/_ref/ImCvt_ref, compiled from /root/reference/src where it lies) makes of them — its exit code, its stdout and the
file it writes — plus the PNG / BMP / QOI files it writes from PNM inputs.  Run in the dev container only; the JSON travels.
Inputs are built here from first principles (zlib + struct), never copied from the reference tree."""
import base64, hashlib, json, os, struct, subprocess, sys, tempfile, zlib

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from formats_cases import load_cases, write_cases          # the same case builders the test uses for live comparison

REF = os.path.join(ROOT, "oracle", "_ref", "ImCvt_ref")


def run_ref(data, src_name, dst_name):
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, src_name), "wb").write(data)
        r = subprocess.run([REF, src_name, "-o", dst_name], cwd=d, capture_output=True, text=True, timeout=120)
        out = os.path.join(d, dst_name)
        return r.returncode, r.stdout, open(out, "rb").read() if os.path.exists(out) else None


def entry(name, data, src_name, dst_name):
    rc, stdout, out = run_ref(data, src_name, dst_name)
    e = {"name": name, "src": src_name, "dst": dst_name, "file_b64": base64.b64encode(zlib.compress(data, 9)).decode(), "rc": rc, "stdout": stdout}
    if out is not None:
        e["out_len"] = len(out)
        e["out_sha256"] = hashlib.sha256(out).hexdigest()
        if len(out) <= 4096:
            e["out_b64"] = base64.b64encode(out).decode()
    return e


out = [entry(n, d, "in.bin", "out.pnm") for n, d in load_cases()]
out += [entry(n, d, "in.pnm", dst) for n, d, dst in write_cases()]
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "formats_kat.json"), "w"), indent=0)
print(len(out), "cases;", sum(e["rc"] == 0 for e in out), "convert")
