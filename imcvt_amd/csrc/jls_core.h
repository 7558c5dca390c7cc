// jls_core.h — JPEG-LS (ITU-T T.87 baseline as the reference implements it, src/imageio_jls.c) for CDNA4.
//
// Why this shape: the adaptive state of JPEG-LS — A,B,C,N of 364 contexts, the two run-interruption contexts, the
// run index and the bit position of the stuffed output — is carried from pixel to pixel in raster order through the
// whole plane, so one plane is one serial chain.  A "row-wavefront" schedule (BASELINE config 5's
// wording) would change the context statistics and therefore the bits.  The exploitable parallelism is across planes:
// one wavefront per plane (three for an RGB picture, one per colour scan), hundreds of planes in flight on the GPU.
// Inside a wavefront 64 lanes stream the rows through LDS (coalesced 16-byte loads), one lane walks the chain with the
// pixel neighbourhood in registers and the contexts in LDS, and the lanes hand the finished bytes over in 16-byte runs.
//
// Compiled by hipcc for gfx950 (jls_hip.hip) and by g++ with -DIMCVT_JLS_HOST for the CPU logic test (tests/hostemu):
// the serial walk is ordinary C++, so that test runs the very same source.
#pragma once
#include <stdint.h>
#include <stddef.h>

#ifdef IMCVT_JLS_HOST
#define JD static inline
#define JHD static inline
#define JLS_LDS
#define JLS_GLB
#else
// Row buffers and contexts live in LDS, the output in global memory: saying so keeps the walker on ds_* / global_*
// instructions (a generic pointer would make every access a FLAT operation that waits on both memory counters).
#define JLS_LDS __attribute__((address_space(3)))
#define JLS_GLB __attribute__((address_space(1)))
#define JD __device__ __forceinline__
#define JHD __host__ __device__ __forceinline__        // file framing is also done by the host shim (RGB: three scans, one frame)
#endif

namespace jls {

struct Ctx { int a, b, c, n; };                         // :241-247
// A context as it is kept in LDS: A stays below 2^15 (|error| <= 128 per sample, halved every 32..64 samples), the stored B lies in
// (-N, 0], C in [-128, 127], N <= 64 — eight bytes, one 64-bit LDS access each way.
struct PCtx { uint16_t a; int8_t b, c; uint8_t n, pad_[3]; };
struct Par { int near, alpha, t1, t2, t3, quant, qbeta, qbpp, limit, a_init, half, recip; };
JD Par make_par(int near) {                             // :26-38 for 8-bit samples
    Par p; p.near = near; p.alpha = 256;
    p.t1 = 3 + 3 * near; p.t2 = 7 + 5 * near; p.t3 = 21 + 7 * near;
    p.quant = 2 * near + 1; p.qbeta = (256 + 4 * near) / p.quant;
    p.qbpp = 1; while ((1 << p.qbpp) < p.qbeta) p.qbpp++;
    p.limit = 32 - p.qbpp - 1;
    p.a_init = (p.qbeta + 32) / 64; if (p.a_init < 2) p.a_init = 2;
    p.half = (p.qbeta + 1) / 2;
    p.recip = ((1 << 20) + p.quant - 1) / p.quant;         // n / quant == (n * recip) >> 20 for 0 <= n < 1024 and odd quant <= 511 (exhaustively checked, tests/test_jls.py)
    return p;
}
JD int iabs(int v) { return v < 0 ? -v : v; }
JD int imin(int a, int b) { return a < b ? a : b; }
JD int imax(int a, int b) { return a > b ? a : b; }
JD int clampi(int v, int lo, int hi) { return imin(imax(v, lo), hi); }
JD int jtab(int i) {                                    // :14 {0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,5,5,6,6,7,7,8..15}
    return i < 16 ? i >> 2 : i < 24 ? 4 + ((i - 16) >> 1) : i - 16;
}
// error quantisation (:97-102) without a division: |e| + near <= 255 + 255 here
JD int quant_err(const Par &p, int e) { const int m = ((iabs(e) + p.near) * p.recip) >> 20; return e < 0 ? -m : m; }
JD int grad(const Par &p, int v) {                      // :67-76
    const int m = iabs(v);
    const int g = m >= p.t3 ? 4 : m >= p.t2 ? 3 : m >= p.t1 ? 2 : m > p.near ? 1 : 0;
    return v < 0 ? -g : g;
}

// MSB-first bit packer with the JPEG-LS stuffing rule: the byte after a 0xFF carries 7 bits (:162-174)
struct Bits {
    JLS_GLB uint8_t *out; int len;
    unsigned acc; int cnt, cap;                        // cnt pending bits in acc (low end, stale bits above), cap = bits the next byte takes
};
JD void bits_init(Bits &w, uint8_t *out) { w.out = (JLS_GLB uint8_t *)out; w.len = 0; w.acc = 0; w.cnt = 0; w.cap = 8; }
JD void bits_drain(Bits &w) {
    while (w.cnt >= w.cap) {
        const unsigned v = (w.acc >> (w.cnt - w.cap)) & ((1u << w.cap) - 1u);
        w.out[w.len++] = (uint8_t)v;
        w.cnt -= w.cap;
        w.cap = (v == 0xFFu) ? 7 : 8;
    }
}
JD void put_bits24(Bits &w, unsigned v, int n) {        // n <= 24: with at most 7 bits pending the 32-bit accumulator holds them
    w.acc = (w.acc << n) | v; w.cnt += n;
    bits_drain(w);
}
JD void put_bits(Bits &w, unsigned v, int n) {          // n <= 32; v has no bits above n
    if (n > 24) { put_bits24(w, v >> 24, n - 24); v &= 0xFFFFFFu; n = 24; }
    put_bits24(w, v, n);
}
JD void bits_flush(Bits &w) {                           // :183-190 — also the empty 7-bit byte after a 0xFF
    if (w.cnt > 0 || w.cap == 7) {
        const unsigned v = (w.acc << (w.cap - w.cnt)) & ((1u << w.cap) - 1u);
        w.out[w.len++] = (uint8_t)v;
        w.cnt = 0; w.cap = 8;
    }
}
JD void golomb(Bits &w, const Par &p, int limit, int v, int k) {   // :193-203
    const int q = v >> k;
    if (q < limit) { put_bits(w, 1u, q + 1); if (k) put_bits(w, (unsigned)v & ((1u << k) - 1u), k); }
    else { put_bits(w, 1u, limit + 1); put_bits(w, (unsigned)(v - 1) & ((1u << p.qbpp) - 1u), p.qbpp); }
}
// smallest k with (n << k) >= a (:114-121), without the loop: the bit lengths give it to within one
JD int bitlen(unsigned v) { return v ? 32 - __builtin_clz(v) : 0; }
JD int golomb_k(int a, int n) {
    if (a <= n) return 0;
    const int k0 = bitlen((unsigned)(a - 1)) - bitlen((unsigned)n);         // 2^(k0-1) n < ... : (n << k0) may still be short of a by one doubling
    const int k = k0 < 0 ? 0 : k0;
    return ((n << k) < a) ? k + 1 : k;
}

// The serial walk of one plane.  State that crosses rows lives here; the row buffers are the caller's:
//   src  [w]      row y of the source plane
//   rec  [w]      row y of the reconstruction, written here (== src values when near == 0)
//   prev [w + 1]  row y-1 of the reconstruction (unused for y == 0)
struct Plane {
    Par p; Bits bw; Ctx ri[2];
    int run_idx, w, prev2_first;                        // prev2_first: reconstruction of (y-2, 0)
};
typedef JLS_LDS PCtx *CtxMem;
typedef JLS_LDS uint8_t *RowMem;
typedef const JLS_LDS uint8_t *CRowMem;
JD void ctx_store(CtxMem p, const Ctx &c) {
    typedef unsigned long long JLS_LDS *W64;
    *(W64)p = (unsigned long long)(uint32_t)(((uint32_t)c.a & 0xFFFFu) | ((uint32_t)c.b & 0xFFu) << 16 | ((uint32_t)c.c & 0xFFu) << 24) | (unsigned long long)((uint32_t)c.n & 0xFFu) << 32;
}
JD Ctx ctx_load(CtxMem p) {
    typedef const unsigned long long JLS_LDS *W64;
    const unsigned long long v = *(W64)p; const uint32_t lo = (uint32_t)v;
    Ctx c; c.a = (int)(lo & 0xFFFFu); c.b = (int)(int8_t)(lo >> 16); c.c = (int)(int8_t)(lo >> 24); c.n = (int)((uint32_t)(v >> 32) & 0xFFu); return c;
}
JD void plane_begin(Plane &S, CtxMem cx, int w, int near, uint8_t *out) {
    S.p = make_par(near); S.w = w; S.run_idx = 0; S.prev2_first = 0;
    bits_init(S.bw, out);
    for (int i = 0; i < 364; i++) { Ctx c; c.a = S.p.a_init; c.b = 0; c.c = 0; c.n = 1; ctx_store(cx + i, c); }
    for (int i = 0; i < 2; i++) { S.ri[i].a = S.p.a_init; S.ri[i].b = 0; S.ri[i].c = 0; S.ri[i].n = 1; }
}
JD void plane_row(Plane &S, CtxMem cx, int y, CRowMem src, RowMem rec, CRowMem prev) {
    const Par &p = S.p;
    const int w = S.w, near = p.near;
    int in_run = 0, run_len = 0;
    // neighbourhood registers (:46-65): a left, b above, c above-left, d above-right
    int b = (y > 0) ? prev[0] : 0;
    int a = b, c = (y > 1) ? S.prev2_first : 0;
    int d = (y > 0) ? ((1 < w) ? prev[1] : b) : 0;
    int v = src[0];
    for (int x = 0; x < w; x++) {
        // next pixel's loads first: they do not depend on the coder state
        const int v_next = (x + 1 < w) ? src[x + 1] : 0;
        const int d_next = (y > 0) ? ((x + 2 < w) ? prev[x + 2] : d) : 0;
        int q = 81 * grad(p, d - b) + 9 * grad(p, b - c) + grad(p, c - a);       // :79-84
        int sgn = q < 0 ? -1 : 1; q = iabs(q);
        int rx;
        if (q == 0) in_run = 1;
        if (in_run && iabs(v - a) <= near) {                                       // run continues (:291-303)
            rx = a;
            if (++run_len >= (1 << jtab(S.run_idx))) { put_bits(S.bw, 1u, 1); run_len -= 1 << jtab(S.run_idx); if (S.run_idx < 31) S.run_idx++; }
            if (x == w - 1 && run_len > 0) put_bits(S.bw, 1u, 1);
        } else if (in_run) {                                                       // run interruption (:305-344)
            const int jr = jtab(S.run_idx), glimit = p.limit - 1 - jr;
            in_run = 0;
            put_bits(S.bw, (unsigned)run_len, jr + 1);
            run_len = 0; if (S.run_idx > 0) S.run_idx--;
            const int t = iabs(a - b) <= near;
            sgn = (a > b + near) ? -1 : 1;
            const int pred = t ? a : b;
            int e = sgn * (v - pred);
            e = quant_err(p, e);
            rx = near ? clampi(pred + sgn * p.quant * e, 0, 255) : v;
            if (e < 0) e += p.qbeta; if (e >= p.half) e -= p.qbeta;
            Ctx r = S.ri[t];
            const int k = golomb_k(r.a + (t ? (r.n >> 1) : 0), r.n);
            const int map = (e != 0) && ((e > 0) == (k == 0 && 2 * r.b < r.n));
            const int me = 2 * iabs(e) - t - map;
            golomb(S.bw, p, glimit, me, k);
            if (e < 0) r.b++;
            r.a += (me + 1 - t) >> 1;
            if (r.n >= 64) { r.a >>= 1; r.b >>= 1; r.n >>= 1; }
            r.n++;
            S.ri[t] = r;
        } else {                                                                   // regular mode (:346-394)
            Ctx r = ctx_load(cx + (q - 1));
            run_len = 0;
            const int lo = imin(a, b), hi = imax(a, b);
            const int med = c >= hi ? lo : c <= lo ? hi : a + b - c;               // :87-94
            const int pred = clampi(med + sgn * r.c, 0, 255);
            int e = sgn * (v - pred);
            e = quant_err(p, e);
            rx = near ? clampi(pred + sgn * p.quant * e, 0, 255) : v;
            if (e < 0) e += p.qbeta; if (e >= p.half) e -= p.qbeta;
            const int k = golomb_k(r.a, r.n);
            const int map = (k == 0) && (2 * r.b <= -r.n) && (near == 0);
            int me = 2 * iabs(e);
            if (e < 0) me -= map + 1; else me += map;
            golomb(S.bw, p, p.limit, me, k);
            r.b += e * p.quant; r.a += iabs(e);
            if (r.n >= 64) { r.a >>= 1; r.b >>= 1; r.n >>= 1; }
            r.n++;
            if (r.b <= -r.n) { r.b = imax(r.b + r.n, -r.n + 1); r.c--; }
            else if (r.b > 0) { r.b = imin(r.b - r.n, 0); r.c++; }
            r.c = clampi(r.c, -128, 127);
            ctx_store(cx + (q - 1), r);
        }
        rec[x] = (uint8_t)rx;
        c = b; b = d; d = d_next; a = rx; v = v_next;
    }
    if (y > 0) S.prev2_first = prev[0];
}
JD long long plane_end(Plane &S) { bits_flush(S.bw); return S.bw.len; }

// File framing (:206-237).  Gray: SOI, SOF55 (11 bytes), one scan header; RGB: SOF55 with three components.
JHD int put_be(uint8_t *o, int at, unsigned v, int nbytes) { while (nbytes-- > 0) o[at++] = (uint8_t)(v >> (8 * nbytes)); return at; }
JHD int frame_header(uint8_t *o, int planes, int h, int w) {
    int at = put_be(o, 0, 0xFFD8u, 2);
    at = put_be(o, at, planes == 3 ? 0xFFF70011u : 0xFFF7000Bu, 4);
    at = put_be(o, at, 8, 1); at = put_be(o, at, (unsigned)h, 2); at = put_be(o, at, (unsigned)w, 2); at = put_be(o, at, (unsigned)planes, 1);
    for (int c = 1; c <= planes; c++) at = put_be(o, at, ((unsigned)c << 16) | 0x1100u, 3);
    return at;
}
JHD int scan_header(uint8_t *o, int at, int comp, int near) {
    at = put_be(o, at, 0xFFDAu, 2); at = put_be(o, at, 8, 2); at = put_be(o, at, 1, 1); at = put_be(o, at, (unsigned)comp, 1);
    at = put_be(o, at, 0, 1); at = put_be(o, at, (unsigned)near, 1); at = put_be(o, at, 0, 2);
    return at;
}
enum { FRAME_HDR_GRAY = 15, FRAME_HDR_RGB = 21, SCAN_HDR = 10 };

}  // namespace jls
