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

// ---- the parallel (one plane over the whole GPU) path of jls_par.h, with every grid as a loop -------------------------
#include "../../imcvt_amd/csrc/jls_par.h"
static long long g_last_bits;
static long long encode_plane_par(const uint8_t *src0, int stride, int h, int w, uint8_t *out) {
    jls::ParPlane P;
    P.src = src0; P.stride = stride; P.h = h; P.w = w; P.out = out; P.hdr = 0;
    std::vector<uint8_t> ws(jls::par_workspace(h, w) + 256, 0xA5);          // dirty workspace: nothing may rely on zeros but the bit buffer
    jls::par_carve(P, (uint8_t *)(((uintptr_t)ws.data() + 255) & ~(uintptr_t)255));
    const long npx = (long)h * w, cmax = (long)jls::par_chunks_max((size_t)npx);
    for (long t = 0; t < npx; t++) jls::k1_classify(P, t);
    for (long y = 0; y < h; y++) jls::k2_rows(P, y, P.rowcnt + (size_t)y * 364);
    for (long t = 0; t < 365; t++) jls::k3_cells(P, t);
    jls::k3_bases(P);
    for (long t = 0; t < npx; t++) jls::k4_scatter(P, t);
    for (long t = 0; t < 365; t++) jls::k5_chain(P, t);
    for (long t = 0; t < npx; t++) jls::k6_len(P, t);
    for (long y = 0; y < h; y++) jls::k6_rowscan(P, y);
    jls::k6_rows(P);
    memset(P.bits, 0, jls::par_bits_bytes(h, w));
    for (long t = 0; t < npx; t++) jls::k7_pack(P, t);
    for (long t = 0; t < cmax * 16; t++) jls::k8_simulate(P, t);
    jls::k8_chain(P, cmax);
    for (long c = 0; c < cmax; c++) jls::k8_write(P, c, cmax);
    g_last_bits = (long long)P.total[0];
    return (long long)P.total[1];
}
extern "C" long long jls_hostemu_last_bits(void) { return g_last_bits; }     // unstuffed bits of the plane coded last (tests aim at chunk edges with it)
extern "C" int jls_hostemu_chunk_bits(void) { return (int)jls::CHUNK_BITS; }
extern "C" long long jls_hostemu_encode_par(const uint8_t *img, int is_rgb, int h, int w, uint8_t *out) {
    const int planes = is_rgb ? 3 : 1;
    int at = jls::frame_header(out, planes, h, w);
    for (int c = 0; c < planes; c++) {
        at = jls::scan_header(out, at, c + 1, 0);
        at += (int)encode_plane_par(img + c, planes, h, w, out + at);
    }
    return jls::put_be(out, at, 0xFFD9u, 2);
}
