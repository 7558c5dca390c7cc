// qoi_io.cpp — QOI reader / writer with the reference's observable behaviour (src/imageio_qoi.c:7-120 writer,
// :125-255 loader; interface src/imageio.h:11,17).  Host-only code of the drop-in converter.
//
// Behaviours kept on purpose (each checked against vectors from the compiled reference, tests/test_host_formats.py):
//   * writer: 14-byte header with channels = 3, colorspace = 0, then the chunks — and NO 8-byte end marker (:99-116)
//   * chunk choice order: run (flushed at 62), index, diff, luma, rgb (:43-85); alpha is constant 255, so the RGBA
//     chunk never appears; a gray image is coded as r = g = b
//   * the writer's colour table is updated for every pixel, also inside runs (:88-91)
//   * loader: channels must be 3 or 4, the result is always RGB (:151,:186); decoding stops when either the pixel
//     buffer is full or the chunk bytes are used up (:241-248) — pixels a short file never reaches are left as
//     they were allocated (zero here, unspecified in the reference)
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace {
struct Px { uint8_t r, g, b, a; };
inline bool same(const Px &x, const Px &y) { return x.r == y.r && x.g == y.g && x.b == y.b && x.a == y.a; }
inline int slot(const Px &p) { return (3 * p.r + 5 * p.g + 7 * p.b + 11 * p.a) & 63; }
}  // namespace

extern "C" int writeQOIImageFile(const char *p_filename, const uint8_t *p_buf, int is_rgb, uint32_t height, uint32_t width) {
    if (width < 1 || height < 1) return 1;
    std::vector<uint8_t> o;
    const size_t npx = (size_t)width * height;
    o.reserve(npx + 64);
    const uint8_t hdr[14] = { 'q', 'o', 'i', 'f', (uint8_t)(width >> 24), (uint8_t)(width >> 16), (uint8_t)(width >> 8), (uint8_t)width,
                              (uint8_t)(height >> 24), (uint8_t)(height >> 16), (uint8_t)(height >> 8), (uint8_t)height, 3, 0 };
    o.insert(o.end(), hdr, hdr + 14);
    Px table[64] = {}, prev = { 0, 0, 0, 255 };
    int run = 0;
    for (size_t i = 0; i < npx; i++) {
        Px c;
        if (is_rgb) { c.r = p_buf[3 * i]; c.g = p_buf[3 * i + 1]; c.b = p_buf[3 * i + 2]; } else c.r = c.g = c.b = p_buf[i];
        c.a = 255;
        const int k = slot(c);
        if (same(c, prev)) {
            if (++run >= 62) { o.push_back((uint8_t)(0xC0 | (run - 1))); run = 0; }
        } else {
            if (run > 0) { o.push_back((uint8_t)(0xC0 | (run - 1))); run = 0; }
            if (same(c, table[k])) o.push_back((uint8_t)k);
            else {
                const uint8_t dr = (uint8_t)(c.r - prev.r + 2), dg = (uint8_t)(c.g - prev.g + 2), db = (uint8_t)(c.b - prev.b + 2);
                if (dr < 4 && dg < 4 && db < 4) o.push_back((uint8_t)(0x40 | dr << 4 | dg << 2 | db));
                else {
                    const uint8_t lr = (uint8_t)(dr - dg + 8), lb = (uint8_t)(db - dg + 8), lg = (uint8_t)(dg + 30);     // mod-256 like :70-72
                    if (lr < 16 && lg < 64 && lb < 16) { o.push_back((uint8_t)(0x80 | lg)); o.push_back((uint8_t)(lr << 4 | lb)); }
                    else { o.push_back(0xFE); o.push_back(c.r); o.push_back(c.g); o.push_back(c.b); }
                }
            }
        }
        table[k] = prev = c;
    }
    if (run > 0) o.push_back((uint8_t)(0xC0 | (run - 1)));
    FILE *fp = fopen(p_filename, "wb");
    if (!fp) return 1;
    const int failed = fwrite(o.data(), 1, o.size(), fp) != o.size();
    fclose(fp);
    return failed;
}

extern "C" uint8_t *loadQOIImageFile(const char *p_filename, int *p_is_rgb, uint32_t *p_height, uint32_t *p_width) {
    FILE *fp = fopen(p_filename, "rb");
    if (!fp) return NULL;
    std::vector<uint8_t> d;
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, fp)) > 0) d.insert(d.end(), buf, buf + n);
    fclose(fp);
    if (d.size() < 4 || memcmp(d.data(), "qoif", 4) != 0) return NULL;
    // header fields read byte-wise with fgetc in the reference: a byte past the end is EOF (-1) added in (:139-148)
    auto at = [&](size_t i) -> uint32_t { return i < d.size() ? d[i] : (uint32_t)-1; };
    uint32_t w = at(4), h = at(8);
    for (int i = 1; i < 4; i++) { w = (w << 8) + at(4 + i); h = (h << 8) + at(8 + i); }
    *p_width = w; *p_height = h;
    const uint8_t channels = (uint8_t)at(12);
    if (w < 1 || h < 1 || channels < 3 || channels > 4) return NULL;
    if (d.size() <= 14) return NULL;                                          // no chunk bytes at all (:164-167)
    const size_t npx3 = (size_t)3 * w * h;
    uint8_t *px = (uint8_t *)calloc(npx3 + 16, 1);
    if (!px) return NULL;
    *p_is_rgb = 1;
    d.resize(d.size() + 16, 0);                                               // a chunk cut off by the end of the file reads on (into slack)
    const size_t end = d.size() - 16;
    size_t q = 14, o = 0;
    Px table[64] = {}, c = { 0, 0, 0, 255 };
    for (;;) {
        const uint8_t tag = d[q++], lo = tag & 63;
        int run = 1;
        switch (tag >> 6) {
        case 0: c = table[lo]; break;
        case 1: c.r += (lo >> 4) - 2; c.g += ((lo >> 2) & 3) - 2; c.b += (lo & 3) - 2; break;
        case 2: { const uint8_t x = d[q++]; c.g += lo - 32; c.r += lo - 40 + (x >> 4); c.b += lo - 40 + (x & 15); break; }
        default:
            if (lo < 62) run = 1 + lo;
            else { c.r = d[q++]; c.g = d[q++]; c.b = d[q++]; if (lo == 63) c.a = d[q++]; }
        }
        table[slot(c)] = c;
        for (; run > 0; run--) {
            px[o] = c.r; px[o + 1] = c.g; px[o + 2] = c.b; o += 3;
            if (o >= npx3) return px;
        }
        if (q >= end) return px;
    }
}
