#!/usr/bin/env python3
"""Generates tests/golden/pnm_kat.json: small PNM files (P1..P6, with the header / padding quirks SURVEY.md §8f lists) and
what the REFERENCE loader (oracle/_ref/libref_pnm.so, compiled from /root/reference/src/imageio_pnm.c where it lies)
returns for each: is_rgb, height, width and the pixel bytes.  Run in the dev container only; the JSON travels."""
import base64, ctypes as C, json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_pnm.so"))
lib.loadPNMImageFile.restype = C.c_void_p
lib.loadPNMImageFile.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
libc = C.CDLL(None); libc.free.argtypes = [C.c_void_p]

def cases():
    px = bytes((7 * i + 3) & 255 for i in range(5 * 4))
    yield "p5_plain_header", b"P5\n5 4\n255\n" + px
    yield "p5_comment_crlf", b"P5 # a comment\r\n5 # width\n4\n255\n" + px
    yield "p5_maxval_15_not_rescaled", b"P5\n5 4\n15\n" + bytes(b & 15 for b in px)
    yield "p5_trailing_junk_on_maxval_line", b"P5\n5 4\n255 junk here\n" + px
    yield "p5_short", b"P5\n5 4\n255\n" + px[:-1]
    yield "p5_maxval_300", b"P5\n5 4\n300\n" + px
    yield "p6_rgb", b"P6\n3 2\n255\n" + bytes(range(18))
    yield "p4_padded_rows", b"P4\n10 3\n" + bytes([0xA5, 0xC0, 0xFF, 0x00, 0x12, 0x40])
    yield "p4_short", b"P4\n10 3\n" + bytes([0xA5, 0xC0, 0xFF])
    yield "p4_no_maxval_line_skips_to_lf", b"P4\n9 2 trailing\n" + bytes([0x80, 0x80, 0x01, 0x00])
    yield "p1_plain", b"P1\n4 3\n1 0 1 0\n0 0 1 1\n1 1 1 0\n"
    yield "p1_packed_digits", b"P1\n4 2\n1010 0 7 0\n0 1 1 0 1\n"
    yield "p2_plain", b"P2\n# c\n3 2\n255\n0 17 255\n 300 4 5\n"
    yield "p2_short", b"P2\n3 2\n255\n0 17 255 1\n"
    yield "p3_plain", b"P3\n2 1\n255\n1 2 3 4 5 6\n"
    yield "bad_magic", b"Q5\n5 4\n255\n" + px
    yield "p7", b"P7\n5 4\n255\n" + px
    yield "zero_width", b"P5\n0 4\n255\n"

out = []
for name, data in cases():
    with tempfile.NamedTemporaryFile(delete=False) as f:
        f.write(data)
    rgb, h, w = C.c_int(-1), C.c_uint32(0), C.c_uint32(0)
    p = lib.loadPNMImageFile(f.name.encode(), C.byref(rgb), C.byref(h), C.byref(w))
    e = {"name": name, "file_b64": base64.b64encode(data).decode(), "ok": bool(p)}
    if p:
        n = (3 if rgb.value else 1) * h.value * w.value
        e.update(is_rgb=rgb.value, h=h.value, w=w.value, pixels_b64=base64.b64encode(C.string_at(p, n)).decode())
        libc.free(p)
    os.unlink(f.name)
    out.append(e)
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "pnm_kat.json"), "w"), indent=0)
print(len(out), "cases;", sum(e["ok"] for e in out), "load")
