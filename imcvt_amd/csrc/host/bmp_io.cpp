// bmp_io.cpp — BMP reader / writer with the reference's observable behaviour (src/imageio_bmp.c:23-90 writer,
// :95-177 loader; interface src/imageio.h:10,16).  Host-only code of the drop-in converter.
//
// Behaviours kept on purpose (each checked against vectors from the compiled reference, tests/test_host_formats.py):
//   * writer: 14 + 40 byte headers, 0xEC4 pixels per metre both ways, gray images get a 256-entry palette
//     (b,g,r,0xFF) and 8 bpp, RGB goes out as 24 bpp BGR; rows bottom-up, zero-padded to 4 bytes (:41-82)
//   * loader accepts 8 / 24 / 32 bpp, BI_RGB only, data offset >= 54, DIB header >= 40, at most 256 palette entries
//     (:119); an 8 bpp file is gray unless some palette entry has unequal components, then it becomes RGB (:131-139)
//   * the palette count is taken literally: 0 means NO entries (every index then maps to 0), not 2^bpp (:133)
//   * nothing checks for a short file: bytes past the end read as 0xFF (fgetc's EOF truncated to a byte, :151-166)
//   * a negative (top-down) height is a huge unsigned one and ends in a failed allocation (:146-148)
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

namespace {

// whole file in memory; reads past the end give EOF (-1) like fgetc, seeks never fail like fseek on a regular file
struct Bytes {
    std::vector<uint8_t> d;
    size_t pos = 0;
    int get() { const int v = pos < d.size() ? d[pos] : -1; if (pos < d.size()) pos++; return v; }
    void skip(size_t n) { pos = (n > d.size() - pos) ? d.size() : pos + n; }
    uint32_t le(int n) { uint32_t v = 0; for (int i = 0; i < n; i++) v |= (uint32_t)get() << (8 * i); return v; }   // EOF ors ones in, as :17
};

bool slurp(const char *name, std::vector<uint8_t> &out) {
    FILE *fp = fopen(name, "rb");
    if (!fp) return false;
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, fp)) > 0) out.insert(out.end(), buf, buf + n);
    fclose(fp);
    return true;
}

void put_le(std::vector<uint8_t> &o, uint32_t v, int n) { for (; n > 0; n--, v >>= 8) o.push_back((uint8_t)v); }

}  // namespace

extern "C" int writeBMPImageFile(const char *p_filename, const uint8_t *p_buf, int is_rgb, uint32_t height, uint32_t width) {
    if (width < 1 || height < 1) return 1;
    const size_t row = (size_t)(is_rgb ? 3 : 1) * width, row_padded = (row + 3) / 4 * 4;
    const uint32_t ncolors = is_rgb ? 0 : 256;
    const size_t data_at = 14 + 40 + 4 * (size_t)ncolors, total = data_at + (size_t)height * row_padded;
    FILE *fp = fopen(p_filename, "wb");
    if (!fp) return 1;
    std::vector<uint8_t> o;
    o.reserve(data_at + row_padded);
    put_le(o, 0x4D42, 2); put_le(o, (uint32_t)total, 4); put_le(o, 0, 4); put_le(o, (uint32_t)data_at, 4);          // file header
    put_le(o, 40, 4); put_le(o, width, 4); put_le(o, height, 4); put_le(o, 1, 2); put_le(o, is_rgb ? 24 : 8, 2);   // DIB header
    put_le(o, 0, 4); put_le(o, 0, 4); put_le(o, 0xEC4, 4); put_le(o, 0xEC4, 4); put_le(o, ncolors, 4); put_le(o, 0, 4);
    for (uint32_t i = 0; i < ncolors; i++) { o.push_back((uint8_t)i); o.push_back((uint8_t)i); o.push_back((uint8_t)i); o.push_back(0xFF); }
    size_t written = fwrite(o.data(), 1, o.size(), fp);
    for (uint32_t y = height; y-- > 0;) {                                     // bottom row first
        const uint8_t *src = p_buf + (size_t)y * row;
        o.assign(row_padded, 0);
        if (is_rgb) for (uint32_t x = 0; x < width; x++) { o[3 * x] = src[3 * x + 2]; o[3 * x + 1] = src[3 * x + 1]; o[3 * x + 2] = src[3 * x]; }
        else for (size_t x = 0; x < row; x++) o[x] = src[x];
        written += fwrite(o.data(), 1, row_padded, fp);
    }
    fclose(fp);
    return written != total;
}

extern "C" uint8_t *loadBMPImageFile(const char *p_filename, int *p_is_rgb, uint32_t *p_height, uint32_t *p_width) {
    Bytes f;
    if (!slurp(p_filename, f.d)) return NULL;
    const uint32_t magic = f.le(2);
    f.skip(8);
    const uint32_t data_at = f.le(4), dib = f.le(4);
    *p_width = f.le(4);
    *p_height = f.le(4);
    f.skip(2);
    const uint32_t bpp = f.le(2), compression = f.le(4);
    f.skip(12);
    const uint32_t ncolors = f.le(4);
    f.skip(4);
    if (magic != 0x4D42 || data_at < 54 || dib < 40 || *p_width < 1 || *p_height < 1 || (bpp != 8 && bpp != 24 && bpp != 32) ||
        compression != 0 || ncolors > 256) return NULL;
    const uint32_t bytes_pp = bpp / 8, w = *p_width, h = *p_height;
    uint8_t pal[256][3] = {};                                                  // r, g, b
    if (bytes_pp > 1) *p_is_rgb = 1;
    else {
        *p_is_rgb = 0;
        f.skip(dib - 40);
        for (uint32_t i = 0; i < ncolors; i++) {
            pal[i][2] = (uint8_t)f.get(); pal[i][1] = (uint8_t)f.get(); pal[i][0] = (uint8_t)f.get(); f.get();
            if (pal[i][0] != pal[i][1] || pal[i][1] != pal[i][2]) *p_is_rgb = 1;
        }
    }
    f.pos = data_at < f.d.size() ? data_at : f.d.size();
    const size_t spp = *p_is_rgb ? 3 : 1;
    uint8_t *px = (uint8_t *)malloc(spp * w * h);
    if (!px) return NULL;
    const uint32_t pad = (bytes_pp * w + 3) / 4 * 4 - bytes_pp * w;
    for (uint32_t i = 0; i < h; i++) {
        uint8_t *row = px + spp * (size_t)(h - 1 - i) * w;
        if (bytes_pp > 1) {
            for (uint32_t x = 0; x < w; x++, row += 3) {
                row[2] = (uint8_t)f.get(); row[1] = (uint8_t)f.get(); row[0] = (uint8_t)f.get();
                if (bytes_pp == 4) f.get();
            }
        } else {
            for (uint32_t x = 0; x < w; x++) {
                const uint8_t v = (uint8_t)f.get();
                *row++ = pal[v][0];
                if (*p_is_rgb) { *row++ = pal[v][1]; *row++ = pal[v][2]; }
            }
        }
        f.skip(pad);
    }
    return px;
}
