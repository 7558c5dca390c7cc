// tests/hostemu/jls_hostemu.cpp — TEST-ONLY: compiles the device JPEG-LS source (imcvt_amd/csrc/jls_core.h) for the host and
// drives it exactly as the kernel does (row buffers rotated per row, one walker per plane), so the kernel's logic is
// checked against the golden vectors on a machine without a GPU.  Not shipped, not loaded by imcvt_amd, not a fallback.
#define IMCVT_JLS_HOST 1
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../imcvt_amd/csrc/jls_core.h"

static long long encode_plane(const uint8_t *src0, int stride, int h, int w, int near, uint8_t *out) {
    std::vector<jls::PCtx> cx(364);
    const int rs = (w + 1 + 15) & ~15;
    std::vector<uint8_t> buf(3 * (size_t)rs);
    uint8_t *src = buf.data(), *rec = src + rs, *prev = rec + rs;            // near == 0: the row is its own reconstruction, as in the kernel
    jls::Plane S;
    jls::plane_begin(S, cx.data(), w, near, out);
    for (int y = 0; y < h; y++) {
        uint8_t *t = prev; prev = rec; rec = t;
        uint8_t *in = near ? src : rec;
        for (int x = 0; x < w; x++) in[x] = src0[((size_t)y * w + x) * stride];
        jls::plane_row(S, cx.data(), y, in, rec, prev);
    }
    return jls::plane_end(S);
}
extern "C" long long jls_hostemu_encode(const uint8_t *img, int is_rgb, int h, int w, int near, uint8_t *out) {
    const int planes = is_rgb ? 3 : 1;
    int at = jls::frame_header(out, planes, h, w);
    for (int c = 0; c < planes; c++) {
        at = jls::scan_header(out, at, c + 1, near);
        at += (int)encode_plane(img + c, planes, h, w, near, out + at);
    }
    return jls::put_be(out, at, 0xFFD9u, 2);
}
