"""Input files for the PNG / BMP / QOI reader and writer tests (tests/test_host_formats.py) and for the generator of
their golden vectors (tests/golden/make_formats_golden.py).  Everything is built from first principles with zlib and
struct; pixel content is seeded integer noise over gradients so that every PNG filter, deflate block type and QOI chunk
type occurs."""
import struct
import zlib

import numpy as np


def picture(h, w, ch, seed):
    """h x w x ch uint8: gradient + blocks + sparse noise (runs, small deltas and jumps all occur)."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    base = (2 * x + 3 * y) & 255
    img = np.stack([(base + 40 * c + ((x // 8 + y // 8) & 1) * 9 * (c + 1)) & 255 for c in range(ch)], axis=-1)
    noise = rng.integers(0, 256, img.shape)
    mask = rng.integers(0, 16, (h, w, 1)) == 0
    img = np.where(mask, noise, img)
    img[h // 3: h // 3 + 2, :, :] = 77                          # long runs
    return img.astype(np.uint8)


# ---------------------------------------------------------------------------------------------------- PNG
def chunk(name, data):
    return struct.pack(">I", len(data)) + name + data + struct.pack(">I", zlib.crc32(name + data))


def paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else b if pb <= pc else c


def filter_rows(rows, bpp, filters):
    """rows: list of bytes (one scanline each); filters: filter type per row (cycled)."""
    out = bytearray()
    prev = bytes(len(rows[0]))
    for i, row in enumerate(rows):
        ft = filters[i % len(filters)]
        out.append(ft)
        for k, v in enumerate(row):
            a = row[k - bpp] if k >= bpp else 0
            b = prev[k]
            c = prev[k - bpp] if k >= bpp else 0
            pred = 0 if ft == 0 else a if ft == 1 else b if ft == 2 else (a + b) // 2 if ft == 3 else paeth(a, b, c) if ft == 4 else 0
            out.append((v - pred) & 255)
        prev = row
    return bytes(out)


def deflate(data, level=9, strategy=zlib.Z_DEFAULT_STRATEGY):
    c = zlib.compressobj(level, zlib.DEFLATED, 15, 9, strategy)
    return c.compress(data) + c.flush()


def png(img, color_type, depth=8, filters=(0,), level=9, strategy=zlib.Z_DEFAULT_STRATEGY, interlace=0, idat_split=0, extra=(), raw_rows=None, z=None):
    """img: h x w x channels uint8 (samples as stored).  extra: chunks inserted before IDAT."""
    h, w = img.shape[:2]
    rows = raw_rows if raw_rows is not None else [img[y].tobytes() for y in range(h)]
    bpp = max(1, len(rows[0]) // w) if depth >= 8 else 1
    zdata = z if z is not None else deflate(filter_rows(rows, bpp, filters), level, strategy)
    f = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color_type, 0, 0, interlace))
    for c in extra:
        f += c
    if idat_split:
        for i in range(0, len(zdata), idat_split):
            f += chunk(b"IDAT", zdata[i:i + idat_split])
    else:
        f += chunk(b"IDAT", zdata)
    return f + chunk(b"IEND", b"")


def png_cases():
    g = picture(13, 17, 1, 1)
    c3 = picture(9, 11, 3, 2)
    c4 = picture(7, 5, 4, 3)
    yield "png_gray8_all_filters_dynamic", png(g, 0, filters=(0, 1, 2, 3, 4))
    yield "png_gray8_stored", png(g, 0, level=0)
    yield "png_gray8_fixed_huffman", png(g, 0, filters=(4,), strategy=zlib.Z_FIXED)
    yield "png_gray8_level1", png(g, 0, filters=(1, 3), level=1)
    yield "png_rgb8_all_filters", png(c3, 2, filters=(4, 3, 2, 1, 0))
    yield "png_rgb8_split_idat_and_ancillary", png(c3, 2, filters=(2, 4), idat_split=7, extra=(chunk(b"gAMA", struct.pack(">I", 45455)), chunk(b"tEXt", b"k\0v")))
    yield "png_rgba8_alpha_dropped", png(c4, 6, filters=(3, 4))
    yield "png_1x1", png(picture(1, 1, 1, 4), 0)
    big = picture(120, 200, 3, 5)
    yield "png_rgb8_120x200_long_matches", png(big, 2, filters=(4, 1, 2))
    yield "png_gray8_300x250_sub", png(picture(250, 300, 1, 6), 0, filters=(1,), level=6)
    # rejections
    g16 = picture(4, 6, 2, 7)
    yield "png_gray16_rejected", png(g16, 0, depth=16)
    yield "png_rgb16_rejected", png(picture(3, 2, 6, 8), 2, depth=16)
    yield "png_rgba16_rejected", png(picture(3, 2, 8, 8), 6, depth=16)
    yield "png_gray_alpha8_rejected", png(g16, 4)
    for d in (1, 2, 4):                                           # 8 pixels a row = d bytes a row
        yield "png_gray%d_rejected" % d, png(np.zeros((3, 8, 1), np.uint8), 0, depth=d, raw_rows=[bytes([0xA5] * d)] * 3)
    yield "png_palette_rejected_code7", png(g, 3, extra=(chunk(b"PLTE", bytes(range(48))),))
    yield "png_depth3_rejected_code7", png(g, 0, depth=3)
    yield "png_interlaced_rejected_code6", png(g, 0, interlace=1)
    yield "png_unknown_critical_chunk_code5", png(g, 0, extra=(chunk(b"ABCD", b"xyz"),))
    yield "png_rgb_with_plte_code5", png(c3, 2, extra=(chunk(b"PLTE", bytes(range(48))),))
    yield "png_bad_filter_type", png(g, 0, filters=(5,))
    yield "png_bad_signature", b"\x89PNG\r\n\x1a\r" + png(g, 0)[8:]
    yield "png_first_chunk_not_ihdr", b"\x89PNG\r\n\x1a\n" + chunk(b"gAMA", struct.pack(">I", 1)) + png(g, 0)[8:]
    yield "png_chunk_runs_past_end", png(g, 0)[:-20]
    yield "png_zlib_bad_header", png(g, 0, z=b"\x78\x02" + deflate(b"abc")[2:])
    yield "png_zlib_preset_dictionary", png(g, 0, z=b"\x78\x20" + deflate(b"abc")[2:])
    yield "png_deflate_reserved_block_type", png(g, 0, z=b"\x78\x01\x07")
    yield "png_output_larger_than_image", png(g, 0, z=deflate(bytes(13 * 18 + 400)))
    yield "png_stored_len_mismatch", png(g, 0, z=b"\x78\x01\x01\x05\x00\x00\x00abcde")
    yield "png_compression_method_1", png(g, 0)[:26] + b"\x01" + png(g, 0)[27:]
    yield "png_too_short", b"\x89PNG\r\n\x1a\n" + b"\0" * 12


# ---------------------------------------------------------------------------------------------------- BMP
def bmp(w, h, bpp, rows, palette=b"", ncolors=None, dib=40, dib_extra=b"", gap=b"", magic=b"BM", compression=0, offset=None):
    """rows: bottom-up list of unpadded row bytes."""
    pad = (-(len(rows[0]))) % 4 if rows else 0
    body = b"".join(r + bytes(pad) for r in rows)
    off = 14 + dib + len(palette) + len(gap) if offset is None else offset
    n = len(palette) // 4 if ncolors is None else ncolors
    head = magic + struct.pack("<IHHI", off + len(body), 0, 0, off)
    info = struct.pack("<IiiHHIIiiII", dib, w, h, 1, bpp, compression, 0, 3780, 3780, n, 0) + dib_extra
    return head + info + palette + gap + body


def bmp_cases():
    rng = np.random.default_rng(11)
    gray_pal = b"".join(bytes([i, i, i, 0]) for i in range(256))
    col_pal = b"".join(bytes([i, 255 - i, (3 * i) & 255, 0]) for i in range(256))
    for w in (1, 2, 3, 4, 5):
        rows = [rng.integers(0, 256, 3 * w, dtype=np.uint8).tobytes() for _ in range(3)]
        yield "bmp_24bpp_w%d_row_padding" % w, bmp(w, 3, 24, rows)
    rows8 = [rng.integers(0, 256, 7, dtype=np.uint8).tobytes() for _ in range(4)]
    yield "bmp_8bpp_gray_palette", bmp(7, 4, 8, rows8, gray_pal)
    yield "bmp_8bpp_colour_palette_becomes_rgb", bmp(7, 4, 8, rows8, col_pal)
    yield "bmp_8bpp_zero_palette_entries_all_black", bmp(7, 4, 8, rows8, b"", ncolors=0)
    yield "bmp_8bpp_16_entries", bmp(7, 4, 8, [bytes(b & 15 for b in r) for r in rows8], gray_pal[:64])
    yield "bmp_32bpp", bmp(3, 2, 32, [rng.integers(0, 256, 12, dtype=np.uint8).tobytes() for _ in range(2)])
    yield "bmp_v5_header_8bpp", bmp(7, 4, 8, rows8, gray_pal, dib=124, dib_extra=bytes(84))
    yield "bmp_gap_before_pixels", bmp(2, 2, 24, [bytes(range(6)), bytes(range(6, 12))], gap=b"\xAA" * 10)
    yield "bmp_truncated_reads_ff", bmp(5, 3, 24, [rng.integers(0, 256, 15, dtype=np.uint8).tobytes() for _ in range(3)])[:-20]
    yield "bmp_header_only", bmp(2, 2, 24, [bytes(6), bytes(6)])[:54]
    yield "bmp_cut_inside_header", bmp(2, 2, 24, [bytes(6), bytes(6)])[:30]
    yield "bmp_bad_magic", bmp(2, 2, 24, [bytes(6), bytes(6)], magic=b"BN")
    yield "bmp_rle_rejected", bmp(7, 4, 8, rows8, gray_pal, compression=1)
    yield "bmp_16bpp_rejected", bmp(2, 2, 16, [bytes(4), bytes(4)])
    yield "bmp_1bpp_rejected", bmp(8, 2, 1, [b"\xA5", b"\x5A"], gray_pal[:8])
    yield "bmp_offset_below_54", bmp(2, 2, 24, [bytes(6), bytes(6)], offset=40)
    yield "bmp_core_header_rejected", bmp(2, 2, 24, [bytes(6), bytes(6)], dib=12)
    yield "bmp_too_many_colours", bmp(7, 4, 8, rows8, gray_pal, ncolors=257)
    yield "bmp_zero_width", bmp(0, 2, 24, [b"", b""])


# ---------------------------------------------------------------------------------------------------- QOI
def qoi_header(w, h, channels=3, cs=0):
    return b"qoif" + struct.pack(">IIBB", w, h, channels, cs)


def qoi_cases():
    yield "qoi_all_chunk_types_rgba_stream", qoi_header(4, 3, 4) + bytes([
        0xFE, 10, 20, 30,            # rgb
        0xC1,                        # run of 2
        0x40 | 3 << 4 | 1 << 2 | 2,  # diff +1 -1 0
        0x80 | 40, 0x9A,             # luma dg=+8
        0xFF, 1, 2, 3, 4,            # rgba
        0x00 | 21,                   # index (whatever is there)
        0xC3,                        # run of 4
        0xFE, 200, 100, 50,
    ]) + bytes(7) + b"\x01"
    yield "qoi_starts_with_run_of_initial_black", qoi_header(3, 2) + bytes([0xC2, 0xFE, 9, 8, 7, 0xC1])
    yield "qoi_run_longer_than_picture", qoi_header(2, 2) + bytes([0xFE, 1, 2, 3, 0xFD])
    yield "qoi_diff_wraps_mod_256", qoi_header(3, 1) + bytes([0x40, 0x7F, 0x6A])
    yield "qoi_channels_2_rejected", qoi_header(2, 2, 2) + bytes([0xFE, 1, 2, 3, 0xFD])
    yield "qoi_channels_5_rejected", qoi_header(2, 2, 5) + bytes([0xFE, 1, 2, 3, 0xFD])
    yield "qoi_zero_height_rejected", qoi_header(2, 0) + bytes([0xFE, 1, 2, 3])
    yield "qoi_header_only_rejected", qoi_header(2, 2)
    yield "qoi_short_header_rejected", qoi_header(2, 2)[:11]
    yield "qoi_bad_magic", b"qoig" + qoi_header(2, 2)[4:] + bytes([0xFD])


def load_cases():
    yield from png_cases()
    yield from bmp_cases()
    yield from qoi_cases()


# ---------------------------------------------------------------------------------------------------- writers
def pnm(img):
    h, w, ch = img.shape
    return (b"P6" if ch == 3 else b"P5") + b"\n%d %d\n255\n" % (w, h) + img.tobytes()


def write_cases():
    """(name, PNM input, output file name)"""
    shapes = [("gray_5x4", picture(4, 5, 1, 21)), ("rgb_3x2", picture(2, 3, 3, 22)), ("rgb_7x5", picture(5, 7, 3, 23)), ("gray_1x1", picture(1, 1, 1, 24)),
              ("gray_64x64", picture(64, 64, 1, 25)), ("rgb_64x48", picture(48, 64, 3, 26)),
              ("gray_300x250_two_stored_blocks", picture(250, 300, 1, 27)),         # 75 250 filtered bytes > 65 535
              ("gray_255x257_exactly_one_block", picture(257, 254, 1, 28)),         # (254 + 1) * 257 = 65 535
              ("rgb_black_runs", np.zeros((9, 40, 3), np.uint8))]
    flat = picture(20, 70, 3, 29)
    flat[:, :, :] = flat[:1, :1, :]                                                 # one colour: QOI runs of 62
    shapes.append(("rgb_flat_70x20", flat))
    for name, img in shapes:
        for ext in ("png", "bmp", "qoi"):
            yield "write_%s_%s" % (ext, name), pnm(img), "out." + ext
    yield "write_png_upper_case_suffix", pnm(shapes[0][1]), "out.PNG"
