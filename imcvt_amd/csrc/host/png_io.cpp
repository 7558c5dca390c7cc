// png_io.cpp — PNG reader / writer with the reference's observable behaviour (src/imageio_png.c:35-105 writer,
// :114-188 loader, which sits on the vendored uPNG decoder src/uPNG/uPNG.c; interface src/imageio.h:9,15).
// Host-only code of the drop-in converter.
//
// Writer (:35-105): 8-bit gray (colour type 0) or RGB (type 2), every scanline with filter 0, the zlib stream made of
// STORED deflate blocks of at most 65535 bytes (header 78 01), one IDAT chunk, no ancillary chunks.
//
// Loader: what the reference accepts is what uPNG decodes AND imageio_png.c keeps (:148): non-interlaced LUMA8, RGB8
// and RGBA8 (alpha dropped with the reference's warning line, :164).  Rejections print the reference's lines:
//   * 16-bit, sub-byte gray, gray+alpha            -> "only support LUMA8, RGB8, and RGBA8. But this PNG is <name>" (:149)
//   * palette images / bad depth (error 7), interlaced (6), unknown critical chunk (5)
//                                                   -> "this PNG format is not-yet supported, error code = N" (:142)
//   * anything malformed fails silently (uPNG.c:898-1116).  Like uPNG, chunk CRCs and the Adler-32 are not verified.
// The inflate below is an independent table-driven implementation of RFC 1951 (canonical Huffman decoding by
// code-length counts), not uPNG's tree walker; for well-formed streams the two cannot differ.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <new>
#include <stdexcept>

namespace {

// ---- writer side
uint32_t crc32_update(uint32_t crc, const uint8_t *p, size_t n) {
    static uint32_t table[256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; table[i] = c; }
        ready = true;
    }
    for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
    return crc;
}
void put_be32(std::vector<uint8_t> &o, uint32_t v) { o.push_back((uint8_t)(v >> 24)); o.push_back((uint8_t)(v >> 16)); o.push_back((uint8_t)(v >> 8)); o.push_back((uint8_t)v); }
void put_chunk(std::vector<uint8_t> &o, const char *name, const std::vector<uint8_t> &data) {
    put_be32(o, (uint32_t)data.size());
    const size_t at = o.size();
    o.insert(o.end(), name, name + 4);
    o.insert(o.end(), data.begin(), data.end());
    put_be32(o, ~crc32_update(0xFFFFFFFFu, o.data() + at, o.size() - at));
}

// ---- loader side: RFC 1951 inflate
struct BitReader {
    const uint8_t *p; size_t n, pos = 0; uint32_t acc = 0; int cnt = 0; bool over = false;
    BitReader(const uint8_t *p_, size_t n_) : p(p_), n(n_) {}
    uint32_t bits(int k) {
        while (cnt < k) { if (pos >= n) { over = true; return 0; } acc |= (uint32_t)p[pos++] << cnt; cnt += 8; }
        const uint32_t v = acc & ((1u << k) - 1);
        acc >>= k; cnt -= k;
        return v;
    }
    void align() { acc = 0; cnt = 0; }
};
struct Huffman {
    uint16_t count[16], symbol[288];
    // canonical code from the code lengths; false for an over-subscribed set
    bool build(const uint8_t *len, int n) {
        memset(count, 0, sizeof count);
        for (int i = 0; i < n; i++) count[len[i]]++;
        int left = 1;
        for (int l = 1; l < 16; l++) { left = 2 * left - count[l]; if (left < 0) return false; }
        uint16_t offs[16]; offs[1] = 0;
        for (int l = 1; l < 15; l++) offs[l + 1] = offs[l] + count[l];
        for (int i = 0; i < n; i++) if (len[i]) symbol[offs[len[i]]++] = (uint16_t)i;
        return true;
    }
    int decode(BitReader &br) const {
        int code = 0, first = 0, index = 0;
        for (int l = 1; l < 16; l++) {
            code |= (int)br.bits(1);
            if (br.over) return -1;
            const int c = count[l];
            if (code - c < first) return symbol[index + (code - first)];
            index += c; first = (first + c) << 1; code <<= 1;
        }
        return -1;
    }
};
const uint16_t kLenBase[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
const uint8_t kLenExtra[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
const uint16_t kDistBase[30] = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577 };
const uint8_t kDistExtra[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };

// zlib stream -> out (at most cap bytes; more is an error, fewer is fine — uPNG.c:1071-1079 sizes the buffer generously)
bool zlib_inflate(const uint8_t *in, size_t n, std::vector<uint8_t> &out, size_t cap) {
    if (n < 2) return false;
    if ((in[0] * 256 + in[1]) % 31 != 0 || (in[0] & 15) != 8 || (in[0] >> 4) > 7 || (in[1] & 32)) return false;   // uPNG.c:645-671
    BitReader br(in + 2, n - 2);
    for (int last = 0; !last;) {
        last = (int)br.bits(1);
        const int type = (int)br.bits(2);
        if (br.over || type == 3) return false;
        if (type == 0) {
            br.align();
            if (br.pos + 4 > br.n) return false;
            const uint32_t len = br.p[br.pos] | br.p[br.pos + 1] << 8, nlen = br.p[br.pos + 2] | br.p[br.pos + 3] << 8;
            br.pos += 4;
            if ((len ^ nlen) != 0xFFFF || br.pos + len > br.n || out.size() + len > cap) return false;
            out.insert(out.end(), br.p + br.pos, br.p + br.pos + len);
            br.pos += len;
            continue;
        }
        Huffman lit, dist;
        uint8_t lens[320];
        if (type == 1) {
            for (int i = 0; i < 288; i++) lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
            lit.build(lens, 288);
            for (int i = 0; i < 30; i++) lens[i] = 5;
            dist.build(lens, 30);
        } else {
            const int nlit = (int)br.bits(5) + 257, ndist = (int)br.bits(5) + 1, ncl = (int)br.bits(4) + 4;
            if (br.over || nlit > 286 || ndist > 30) return false;
            static const uint8_t order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
            uint8_t cl[19] = {};
            for (int i = 0; i < ncl; i++) cl[order[i]] = (uint8_t)br.bits(3);
            Huffman clh;
            if (br.over || !clh.build(cl, 19)) return false;
            for (int i = 0; i < nlit + ndist;) {
                const int s = clh.decode(br);
                if (s < 0) return false;
                if (s < 16) { lens[i++] = (uint8_t)s; continue; }
                int rep, v = 0;
                if (s == 16) { if (i == 0) return false; v = lens[i - 1]; rep = 3 + (int)br.bits(2); }
                else if (s == 17) rep = 3 + (int)br.bits(3);
                else rep = 11 + (int)br.bits(7);
                if (br.over || i + rep > nlit + ndist) return false;
                while (rep--) lens[i++] = (uint8_t)v;
            }
            if (lens[256] == 0 || !lit.build(lens, nlit) || !dist.build(lens + nlit, ndist)) return false;
        }
        for (;;) {
            const int s = lit.decode(br);
            if (s < 0) return false;
            if (s < 256) { if (out.size() >= cap) return false; out.push_back((uint8_t)s); continue; }
            if (s == 256) break;
            if (s > 285) return false;
            const int len = kLenBase[s - 257] + (int)br.bits(kLenExtra[s - 257]);
            const int ds = dist.decode(br);
            if (ds < 0 || ds > 29) return false;
            const size_t d = kDistBase[ds] + br.bits(kDistExtra[ds]);
            if (br.over || d > out.size() || out.size() + len > cap) return false;
            for (int k = 0; k < len; k++) out.push_back(out[out.size() - d]);
        }
    }
    return true;
}

int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc) ? b : c;
}
uint32_t be32(const uint8_t *p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }

enum { kOk = 0, kNotPng = 3, kMalformed = 4, kUnsupported = 5, kInterlaced = 6, kFormat = 7 };     // upng_error values, uPNG.h:30-40

// names imageio_png.c:119-133 prints for the formats uPNG decodes but the converter does not take
const char *format_name(int color_type, int depth) {
    static const char *lum[] = { "LUMA1", "LUMA2", "LUMA4", "LUMA8" }, *luma[] = { "LUMA_ALPHA1", "LUMA_ALPHA2", "LUMA_ALPHA4", "LUMA_ALPHA8" };
    const int k = depth == 1 ? 0 : depth == 2 ? 1 : depth == 4 ? 2 : depth == 8 ? 3 : -1;
    if (color_type == 0) return k >= 0 ? lum[k] : NULL;
    if (color_type == 4) return k >= 0 ? luma[k] : NULL;
    if (color_type == 2) return depth == 8 ? "RGB8" : depth == 16 ? "RGB16" : NULL;
    if (color_type == 6) return depth == 8 ? "RGBA8" : depth == 16 ? "RGBA16" : NULL;
    return NULL;
}

// decodes into `img` (samples as stored: 1, 3 or 4 bytes per pixel for the accepted formats); returns an upng_error value
int decode_unguarded(const std::vector<uint8_t> &f, uint32_t &w, uint32_t &h, int &color_type, int &depth, std::vector<uint8_t> &img);
// A header may declare any size: a picture whose buffers cannot be allocated is reported like the reference's failed
// malloc (uPNG.c:1071-1079 -> "open failed"), not by an exception leaving an extern "C" loader.
int decode(const std::vector<uint8_t> &f, uint32_t &w, uint32_t &h, int &color_type, int &depth, std::vector<uint8_t> &img) {
    try { return decode_unguarded(f, w, h, color_type, depth, img); }
    catch (const std::bad_alloc &) { return kMalformed; }
    catch (const std::length_error &) { return kMalformed; }
}
int decode_unguarded(const std::vector<uint8_t> &f, uint32_t &w, uint32_t &h, int &color_type, int &depth, std::vector<uint8_t> &img) {
    static const uint8_t sig[8] = { 137, 80, 78, 71, 13, 10, 26, 10 };
    if (f.size() < 29 || memcmp(f.data(), sig, 8) != 0) return kNotPng;
    if (memcmp(f.data() + 12, "IHDR", 4) != 0) return kMalformed;
    w = be32(&f[16]); h = be32(&f[20]); depth = f[24]; color_type = f[25];
    if (!format_name(color_type, depth)) return kFormat;
    if (f[26] != 0 || f[27] != 0) return kMalformed;
    if (f[28] != 0) return kInterlaced;
    std::vector<uint8_t> z;
    for (size_t at = 33; at < f.size();) {                                    // chunk walk, uPNG.c:1001-1041
        if (at + 12 > f.size()) return kMalformed;
        const uint32_t len = be32(&f[at]);
        if (len > 0x7FFFFFFFu || at + len + 12 > f.size()) return kMalformed;
        if (memcmp(&f[at + 4], "IDAT", 4) == 0) z.insert(z.end(), f.begin() + at + 8, f.begin() + at + 8 + len);
        else if (memcmp(&f[at + 4], "IEND", 4) == 0) break;
        else if (!(f[at + 4] & 32)) return kUnsupported;                      // critical chunk we do not know
        at += (size_t)len + 12;
    }
    const int channels = color_type == 0 ? 1 : color_type == 2 ? 3 : color_type == 4 ? 2 : 4;
    const size_t bpp = (size_t)channels * depth, line = ((size_t)w * bpp + 7) / 8, bytes_pp = (bpp + 7) / 8;
    std::vector<uint8_t> raw;
    if (w == 0 || h == 0 || line > ((size_t)1 << 40) / h) return kMalformed;   // (sizes whose products would wrap)
    const size_t cap = ((size_t)w * ((size_t)h * bpp + 7)) / 8 + h;           // uPNG.c:1071
    if (!zlib_inflate(z.data(), z.size(), raw, cap)) return kMalformed;
    raw.resize((line + 1) * (size_t)h > raw.size() ? (line + 1) * (size_t)h : raw.size(), 0);
    img.assign(line * (size_t)h, 0);
    for (uint32_t y = 0; y < h; y++) {                                        // unfilter, uPNG.c:693-783
        const uint8_t *s = &raw[(line + 1) * (size_t)y + 1];
        uint8_t *r = &img[line * (size_t)y];
        const uint8_t *up = y ? r - line : NULL;
        const int ft = s[-1];
        if (ft > 4) return kMalformed;
        for (size_t i = 0; i < line; i++) {
            const int a = i >= bytes_pp ? r[i - bytes_pp] : 0, b = up ? up[i] : 0, c = (up && i >= bytes_pp) ? up[i - bytes_pp] : 0;
            const int pred = ft == 0 ? 0 : ft == 1 ? a : ft == 2 ? b : ft == 3 ? (a + b) / 2 : paeth(a, b, c);
            r[i] = (uint8_t)(s[i] + pred);
        }
    }
    return kOk;
}

}  // namespace

extern "C" int writePNGImageFile(const char *p_filename, const uint8_t *p_buf, int is_rgb, uint32_t height, uint32_t width) {
    if (width < 1 || height < 1) return 1;
    const size_t line = (size_t)(is_rgb ? 3 : 1) * width + 1, total = line * height;
    std::vector<uint8_t> o, ihdr, idat;
    static const uint8_t sig[8] = { 137, 80, 78, 71, 13, 10, 26, 10 };
    o.insert(o.end(), sig, sig + 8);
    put_be32(ihdr, width); put_be32(ihdr, height);
    ihdr.push_back(8); ihdr.push_back(is_rgb ? 2 : 0); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    put_chunk(o, "IHDR", ihdr);
    idat.reserve(total + total / 65535 * 5 + 16);
    idat.push_back(0x78); idat.push_back(0x01);
    uint32_t a = 1, b = 0;
    for (size_t i = 0; i < total; i++) {
        if (i % 0xFFFF == 0) {                                                // a stored block per 65535 bytes (:72-79, :89-94)
            const size_t left = total - i, n = left < 0xFFFF ? left : 0xFFFF;
            idat.push_back(left <= 0xFFFF ? 1 : 0);
            idat.push_back((uint8_t)n); idat.push_back((uint8_t)(n >> 8)); idat.push_back((uint8_t)~n); idat.push_back((uint8_t)(~n >> 8));
        }
        const uint8_t v = (i % line == 0) ? 0 : *p_buf++;                     // filter byte 0, then the row
        idat.push_back(v);
        a = (a + v) % 65521; b = (b + a) % 65521;
    }
    put_be32(idat, b << 16 | a);
    put_chunk(o, "IDAT", idat);
    put_chunk(o, "IEND", std::vector<uint8_t>());
    FILE *fp = fopen(p_filename, "wb");
    if (!fp) return 1;
    fwrite(o.data(), 1, o.size(), fp);
    fclose(fp);
    return 0;                                                                 // like the reference, a short write is not reported (:101-104)
}

// quiet != 0 suppresses the reference's stdout lines (used when the converter pre-loads inputs for its device batch)
extern "C" uint8_t *imcvt_load_png(const char *p_filename, int *p_is_rgb, uint32_t *p_height, uint32_t *p_width, int quiet) {
    FILE *fp = fopen(p_filename, "rb");
    if (!fp) return NULL;
    std::vector<uint8_t> f;
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, fp)) > 0) f.insert(f.end(), buf, buf + n);
    fclose(fp);
    uint32_t w = 0, h = 0; int ct = 0, depth = 0;
    std::vector<uint8_t> img;
    const int err = decode(f, w, h, ct, depth, img);
    if (err != kOk) {
        if (!quiet && (err == kUnsupported || err == kInterlaced || err == kFormat)) printf("   ***ERROR: this PNG format is not-yet supported, error code = %d\n", err);
        return NULL;
    }
    if (depth != 8 || ct == 4) {
        if (!quiet) printf("   ***ERROR: only support LUMA8, RGB8, and RGBA8. But this PNG is %s\n", format_name(ct, depth));
        return NULL;
    }
    *p_is_rgb = ct != 0; *p_height = h; *p_width = w;
    const size_t npx = (size_t)w * h, bytes = (ct ? 3 : 1) * npx;
    uint8_t *px = (uint8_t *)malloc(bytes ? bytes : 1);
    if (!px) return NULL;
    if (ct == 6) {
        if (!quiet) printf("   *warning: disard alpha channel of this PNG\n");
        for (size_t i = 0; i < npx; i++) { px[3 * i] = img[4 * i]; px[3 * i + 1] = img[4 * i + 1]; px[3 * i + 2] = img[4 * i + 2]; }
    } else memcpy(px, img.data(), bytes);
    return px;
}
extern "C" uint8_t *loadPNGImageFile(const char *p_filename, int *p_is_rgb, uint32_t *p_height, uint32_t *p_width) {
    return imcvt_load_png(p_filename, p_is_rgb, p_height, p_width, 0);
}
