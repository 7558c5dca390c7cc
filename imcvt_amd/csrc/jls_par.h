// jls_par.h — ONE lossless JPEG-LS plane spread over the whole GPU (BASELINE config 5; reference src/imageio_jls.c:240-399).
//
// The reference walks the plane pixel by pixel because every pixel updates adaptive state.  With NEAR = 0 the
// reconstruction IS the input (:325, :361), so everything that only reads the neighbourhood is data-parallel, and the
// adaptive state falls apart into independent chains:
//
//   k1  per pixel   neighbourhood a,b,c,d (:46-65), context q and sign (:79-84), MED prediction (:87-94), x == a
//   k2  per row     run / run-interruption / regular classification — `running` is a scan along the row, reset per row (:270,
//                   :284-290); stable rank of every regular pixel inside its (row, context) cell; run segments -> events
//   k3  per context exclusive sums of the cell counts over the rows: where each context's pixels start in its list
//   k4  per pixel   scatter: every context's pixels, in raster order, as a dense list of (x, prediction, sign); events
//   k5  per chain   364 regular contexts (:346-394), each walking ITS list with A,B,C,N in registers, and one run chain
//                   (run index :250,:293-303,:306-311 and the two run-interruption contexts :313-343) walking the events;
//                   output = Golomb code word + length per list element (:187-197)
//   k6  per pixel / per row   code lengths in raster order, prefix sums -> bit position of every pixel's code words
//   k7  per pixel   code words into an (unstuffed) bit stream at their positions
//   k8  chunks      bit stuffing — after a 0xFF byte the next byte carries 7 bits (:156-168): every chunk of the unstuffed
//                   stream is simulated from each of its 16 possible entry states, a short serial pass picks the real ones,
//                   then every chunk writes its bytes
//
// Every step is one thread per item with no intra-workgroup communication, so the same source compiles for the host
// (-DIMCVT_JLS_HOST, tests/hostemu/jls_hostemu.cpp: every grid is a loop) and the bits are checked on the CPU against the
// golden vectors.  Compiled for gfx950 by jls_hip.hip.  NEAR > 0 keeps the walker path (jls_core.h): there the
// reconstruction depends on the coded errors and the neighbourhood is no longer known in advance.
#pragma once
#include "jls_core.h"

#ifdef IMCVT_JLS_HOST
#define JLS_UNROLL
#else
#define JLS_UNROLL _Pragma("unroll")
#endif

namespace jls {

struct ParPlane {
    const uint8_t *src; int stride;      // first sample, bytes between samples (1 gray, 3 interleaved RGB)
    int h, w;
    // per pixel
    uint16_t *qs;        // q (0..364) | sign<0 << 9 | (q == 0) << 10 | (x == a) << 11
    uint8_t  *med;       // MED prediction
    uint8_t  *cls;       // CL_*
    uint16_t *rank;      // regular: rank inside its (row, context) cell; segment end: event number inside the row
    uint16_t *aux;       // segment end: run length
    uint32_t *pos;       // regular: position in the context lists; segment end: event index
    uint8_t  *len;       // bits this pixel emits
    uint32_t *bitpos;    // bit position of the pixel's first code word inside its row
    // per (row, context)
    uint32_t *rowcnt;    // [h][364] counts, turned into exclusive row sums by k3
    uint32_t *rowev;     // [h] events per row, turned into exclusive sums by k3
    uint32_t *binbase;   // [365] start of each context's list (k3); [364] = number of regular pixels
    unsigned long long *rowbits;   // [h] bits per row -> exclusive sums
    // lists
    uint32_t *list;      // [npx] x | (MED prediction + s) << 8 | s << 17 with s = 1 for the negative sign: context-major, raster order inside a context
    uint32_t *code;      // [2 npx] what each sample's code word is made from (regular_step below), same order
    uint32_t *ev;        // [2 * nev] run length | type << 16, x | a << 8 | b << 16
    uint32_t *evout;     // [3 * nev] ones, run-count bits (value | len << 24), Golomb code (value | len << 24)
    uint32_t *bits;      // unstuffed bit stream, MSB-first 32-bit words, zeroed before k7
    // stuffing
    uint32_t *chunk;     // [nchunk][16] exit state | bytes << 8 of every (chunk, entry state); then [nchunk] entry state | byte offset
    unsigned long long *total;     // [0] bits of the scan, [1] bytes of the stuffed scan
    uint8_t *out;        // the scan's bytes go to out + hdr
    int hdr;
};
enum { CL_REG = 0, CL_RUN = 1, CL_RUNEND = 2, CL_EOL = 3 };      // CL_EOL: last pixel of a row, inside a run
#ifndef JLS_CHUNK_BITS
#define JLS_CHUNK_BITS 16384                                        // unstuffed bits per stuffing chunk (tests also build the host form with 64: a chunk edge every few pixels)
#endif
enum { CHUNK_BITS = JLS_CHUNK_BITS };

// Pointers that come out of a struct are generic to the compiler: FLAT loads / stores, which wait on both memory counters
// (a load issued ahead of its use would be waited for at the next store).  G() says "global memory" at the point of use.
#define G(type, ptr) ((JLS_GLB type *)(ptr))
// A context chain is ONE dependent program.  Run by a whole wavefront on wave-uniform values it is executed by the scalar
// unit (s_* instructions, one issue per cycle or two) instead of as 64-lane vector instructions of which one lane matters:
// UNI() tells the compiler a loaded value is the same in every lane.
#if defined(IMCVT_JLS_HOST) || defined(JLS_CHAIN_VECTOR)
#define UNI(x) (x)
#else
#define UNI(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
#endif
JD int px_at(const ParPlane &P, int y, int x) { return G(const uint8_t, P.src)[((size_t)y * P.w + x) * P.stride]; }

// ---- k1: one thread per pixel
JD void k1_classify(const ParPlane &P, long t) {
    const int y = (int)(t / P.w), x = (int)(t % P.w);
    const int v = px_at(P, y, x);
    int a, b = 0, c = 0, d = 0;                                                     // :46-65
    if (y > 0) { b = px_at(P, y - 1, x); d = (x + 1 < P.w) ? px_at(P, y - 1, x + 1) : b; }
    if (x == 0) { a = b; if (y > 1) c = px_at(P, y - 2, 0); }
    else { a = px_at(P, y, x - 1); if (y > 0) c = px_at(P, y - 1, x - 1); }
    const Par p = make_par(0);
    int q = 81 * grad(p, d - b) + 9 * grad(p, b - c) + grad(p, c - a);              // :79-84
    const int neg = q < 0; q = iabs(q);
    const int lo = imin(a, b), hi = imax(a, b);
    P.med[t] = (uint8_t)(c >= hi ? lo : c <= lo ? hi : a + b - c);                  // :87-94
    P.qs[t] = (uint16_t)(q | neg << 9 | (q == 0) << 10 | (v == a) << 11);
}

// ---- k2: one thread per row.  cnt = the row's 364 cell counters (device: an LDS row, copied out by the caller; host: the cell array itself)
template <class CntPtr>
JD void k2_rows(const ParPlane &P, long row, CntPtr cnt) {
    const size_t r0 = (size_t)row * P.w;
    JLS_GLB const uint16_t *qs = G(const uint16_t, P.qs) + r0;
    JLS_GLB uint8_t *cls = G(uint8_t, P.cls) + r0;
    JLS_GLB uint16_t *aux = G(uint16_t, P.aux) + r0, *rank = G(uint16_t, P.rank) + r0;
    for (int i = 0; i < 364; i++) cnt[i] = 0;
    int running = 0, run = 0, nev = 0;
    for (int x0 = 0; x0 < P.w; x0 += 8) {
        uint16_t sv[8];
        for (int i = 0; i < 8; i++) sv[i] = qs[imin(x0 + i, P.w - 1)];             // loads first: they do not depend on the scan state
        for (int i = 0; i < 8 && x0 + i < P.w; i++) {
            const int x = x0 + i, s = sv[i], q = s & 511;
            if (running | ((s >> 10) & 1)) {                                        // :284-290
                if ((s >> 11) & 1) {
                    running = 1; run++;
                    if (x == P.w - 1) { cls[x] = CL_EOL; aux[x] = (uint16_t)run; rank[x] = (uint16_t)nev++; }
                    else cls[x] = CL_RUN;
                } else {
                    cls[x] = CL_RUNEND; aux[x] = (uint16_t)run; rank[x] = (uint16_t)nev++;
                    running = 0; run = 0;
                }
            } else {
                cls[x] = CL_REG;
                rank[x] = (uint16_t)cnt[q - 1]++;
            }
        }
    }
    G(uint32_t, P.rowev)[row] = (uint32_t)nev;
}

// ---- k3: one thread per context (t < 364: exclusive sums over the rows; t == 364: events per row), then ONE thread for the bases
JD void k3_cells(const ParPlane &P, long t) {
    uint32_t s = 0;
    if (t < 364) {
        for (int y0 = 0; y0 < P.h; y0 += 8) {
            uint32_t v[8];
            for (int i = 0; i < 8; i++) v[i] = (y0 + i < P.h) ? G(uint32_t, P.rowcnt)[(size_t)(y0 + i) * 364 + t] : 0;
            for (int i = 0; i < 8 && y0 + i < P.h; i++) { G(uint32_t, P.rowcnt)[(size_t)(y0 + i) * 364 + t] = s; s += v[i]; }
        }
        P.binbase[t] = s;                                                           // (the total, for now)
    } else {
        for (int y = 0; y < P.h; y++) { const uint32_t v = P.rowev[y]; P.rowev[y] = s; s += v; }
        P.binbase[365] = s;                                                         // number of events
    }
}
JD void k3_bases(const ParPlane &P) {
    uint32_t s = 0;
    for (int i = 0; i < 364; i++) { const uint32_t v = P.binbase[i]; P.binbase[i] = s; s += v; }
    P.binbase[364] = s;
}

// ---- k4: one thread per pixel
JD void k4_scatter(const ParPlane &P, long t) {
    const int y = (int)(t / P.w), x = (int)(t % P.w);
    const int s = P.qs[t], q = s & 511, cl = P.cls[t];
    if (cl == CL_REG) {
        const uint32_t at = P.binbase[q - 1] + P.rowcnt[(size_t)y * 364 + (q - 1)] + P.rank[t];
        P.pos[t] = at;
        const uint32_t sg = (uint32_t)((s >> 9) & 1);
        P.list[at] = (uint32_t)px_at(P, y, x) | ((uint32_t)P.med[t] + sg) << 8 | sg << 17;
    } else if (cl == CL_RUN) P.pos[t] = 0xFFFFFFFFu;
    else {
        const uint32_t e = P.rowev[y] + P.rank[t];
        P.pos[t] = e;
        int a, b = 0;
        if (y > 0) b = px_at(P, y - 1, x);
        a = (x == 0) ? b : px_at(P, y, x - 1);
        P.ev[2 * e] = (uint32_t)P.aux[t] | (uint32_t)cl << 16;
        P.ev[2 * e + 1] = (uint32_t)px_at(P, y, x) | (uint32_t)a << 8 | (uint32_t)b << 16;
    }
}

// Golomb code word (:187-197) as value | length << 24: `zeros` zero bits, a one, then `k` (or qbpp) low bits
JD uint32_t golomb_word(const Par &p, int limit, int v, int k) {
    const int zeros = v >> k;
    if (zeros < limit) return ((1u << k) | ((uint32_t)v & ((1u << k) - 1u))) | (uint32_t)(zeros + 1 + k) << 24;
    return ((1u << p.qbpp) | ((uint32_t)(v - 1) & ((1u << p.qbpp) - 1u))) | (uint32_t)(limit + 1 + p.qbpp) << 24;
}

// one regular-mode sample (:346-394 with near == 0): code word out, context updated.  Written without branches: the chain is
// one lane's dependent program (a taken branch costs it more than the few instructions it skips), and with qbeta = 256 the
// range reduction of :105-111 is ((e + 128) & 255) - 128.
JD int golomb_k_nb(int a, int n) {                        // golomb_k without its early exit (a <= n gives k0 <= 0)
    const int k = imax(bitlen((unsigned)imax(a - 1, 0)) - bitlen((unsigned)n), 0);
    return k + ((n << k) < a);
}
// clamp on wave-uniform values: the instruction selector turns the max/min pattern into the vector unit's v_med3 even when
// everything around it is scalar, which costs a round trip through v_readfirstlane
#if defined(IMCVT_JLS_HOST) || defined(JLS_CHAIN_VECTOR)
JD int uclamp(int v, int lo, int hi) { return clampi(v, lo, hi); }
#else
JD int uclamp(int v, int lo, int hi) {
    int t;
    asm("s_max_i32 %0, %1, %2" : "=s"(t) : "s"(v), "s"(lo) : "scc");
    asm("s_min_i32 %0, %1, %2" : "=s"(t) : "s"(t), "s"(hi) : "scc");
    return t;
}
#endif
// The chain itself only carries the context: it hands out, per sample, what the code word is made from — A, N and B before
// the update (A < 2^15, N <= 64, -64 < B <= 0) and the 8-bit error — as two words of two 16-bit halves (one s_pack each); the
// code word (regular_word below) is computed where it is consumed, by the per-pixel kernels k6 / k7, off the chain.
//     w0 = A | N << 16        w1 = (e & 0xFFFF) | B << 16
// The chain is a lone wave's scalar program and costs its instruction COUNT (one issue per ~4 cycles whatever the instruction), so
// the list element arrives pre-digested (x, MED + s and s at fixed bit fields: med + sign * C = (MED + s) + (C ^ -s)) and the
// B / C update of :383-392 is written on the scalar condition code directly: compare, select, add-with-carry.
#if defined(IMCVT_JLS_HOST) || defined(JLS_CHAIN_VECTOR)
JD uint32_t pack16(int lo, int hi) { return ((uint32_t)lo & 0xFFFFu) | (uint32_t)hi << 16; }
// B beyond (-N, 0] is pulled back by N (at most to the edge) and C steps (:383-392); nn = -N
JD void bias_step(int &B, int &c, int N, int nn) {
    const int lo = B <= nn, hi = B > 0;
    const int b_lo = imax(B + N, 1 + nn), b_hi = imin(B - N, 0);
    B = lo ? b_lo : hi ? b_hi : B;
    c += hi - lo;
}
#else
JD uint32_t pack16(int lo, int hi) { uint32_t t; asm("s_pack_ll_b32_b16 %0, %1, %2" : "=s"(t) : "s"(lo), "s"(hi)); return t; }
JD void bias_step(int &B, int &c, int N, int nn) {
    const int b_lo = imax(B + N, 1 + nn), b_hi = imin(B - N, 0);
    int Bn = B, cn = c;
    asm("s_cmp_gt_i32 %2, 0\n\ts_cselect_b32 %0, %3, %2\n\ts_addc_u32 %1, %1, 0\n\t"
        "s_cmp_le_i32 %2, %5\n\ts_cselect_b32 %0, %4, %0\n\ts_subb_u32 %1, %1, 0"
        : "=&s"(Bn), "+s"(cn) : "s"(B), "s"(b_hi), "s"(b_lo), "s"(nn) : "scc");
    B = Bn; c = cn;
}
#endif
JD void regular_step(const Par &p, Ctx &r, uint32_t el, uint32_t &w0, uint32_t &w1) {
    (void)p;
    const int m = (int)(el << 14) >> 31;                                                           // -1 for the negative sign
    const int pred = uclamp((int)((el >> 8) & 511u) + (r.c ^ m), 0, 255);                          // med + sign * C (:349-350)
    int e = (int)el - pred;                                                                        // (only the low byte counts from here on)
    e = (e ^ m) - m;                                                                               // sign * (x - px)
    e = (int)(int8_t)e;                                                                            // modRange with qbeta = 256 (:105-111) is the sign extension of the low byte
    const int ae = iabs(e);
    w0 = pack16(r.a, r.n); w1 = pack16(e, r.b);
    const int rs = r.n >> 6;                                                                       // N >= 64 (N never exceeds 64): :376-381
    int B = (r.b + e) >> rs; const int N = (r.n >> rs) + 1;
    r.a = (r.a + ae) >> rs;
    bias_step(B, r.c, N, -N);
    r.c = uclamp(r.c, -128, 127);
    r.b = B; r.n = N;
}
// code word (value | length << 24) of a regular-mode sample from its two state words (:363-375, :187-197 with qbpp = 8)
JD uint32_t regular_word(const Par &p, uint32_t w0, uint32_t w1) {
    const int a = (int)(w0 & 0xFFFFu), n = (int)(w0 >> 16), e = (int)(int16_t)(w1 & 0xFFFFu), b = (int)w1 >> 16, nb = 2 * b <= -n;
    const int ae = iabs(e), neg = (int)((uint32_t)e >> 31);
    const int k = golomb_k_nb(a, n);
    const int map = (k == 0) & nb;                                                                 // :366
    const int me = 2 * ae + map - neg * (2 * map + 1);                                             // :367-372: e < 0 ? 2|e| - map - 1 : 2|e| + map
    const int zeros = me >> k, esc = zeros >= p.limit;
    const uint32_t val_n = (1u << k) | ((uint32_t)me & ((1u << k) - 1u)), val_e = 256u | ((uint32_t)(me - 1) & 255u);
    const uint32_t word_n = val_n | (uint32_t)(zeros + 1 + k) << 24, word_e = val_e | (uint32_t)(p.limit + 1 + 8) << 24;
    return esc ? word_e : word_n;
}

// ---- k5: one thread per chain.  t < 364: regular context t; t == 364: the run chain
JD void k5_chain(const ParPlane &P, long t) {
    const Par p = make_par(0);
    if (t < 364) {
        const uint32_t base = UNI(G(const uint32_t, P.binbase)[t]), n = UNI(G(const uint32_t, P.binbase)[t + 1]) - base;
        Ctx r; r.a = p.a_init; r.b = 0; r.c = 0; r.n = 1;
        // Whole blocks of eight samples: the list is read TWO blocks ahead of its use (stores and loads share one in-order
        // completion counter on this target, so a wait for a block's loads also waits for the code words stored before them:
        // with two blocks of distance both have had a block's time to complete).  Constant indices into fully unrolled
        // loops keep the blocks in registers.  The list is padded, reading past a chain's end is harmless.
        JLS_GLB const uint32_t *list = G(const uint32_t, P.list) + base;
        JLS_GLB uint32_t *code = G(uint32_t, P.code) + 2 * (size_t)base;
        uint32_t b0[8], b1[8];
        JLS_UNROLL for (int j = 0; j < 8; j++) { b0[j] = UNI(list[j]); b1[j] = UNI(list[8 + j]); }
        uint32_t i0 = 0;
        for (; i0 + 8 <= n; i0 += 8) {
            uint32_t cur[8], wa[8], wb[8];
            JLS_UNROLL for (int j = 0; j < 8; j++) { cur[j] = b0[j]; b0[j] = b1[j]; b1[j] = UNI(list[i0 + 16 + j]); }
            JLS_UNROLL for (int j = 0; j < 8; j++) regular_step(p, r, cur[j], wa[j], wb[j]);
            JLS_UNROLL for (int j = 0; j < 8; j++) { code[2 * (i0 + j)] = wa[j]; code[2 * (i0 + j) + 1] = wb[j]; }
        }
        for (; i0 < n; i0++) { uint32_t wa, wb; regular_step(p, r, UNI(list[i0]), wa, wb); code[2 * i0] = wa; code[2 * i0 + 1] = wb; }
    } else {
        const uint32_t nev = P.binbase[365];
        Ctx ri[2];
        for (int i = 0; i < 2; i++) { ri[i].a = p.a_init; ri[i].b = 0; ri[i].c = 0; ri[i].n = 1; }
        int run_idx = 0;
        for (uint32_t ei = 0; ei < nev; ei++) {
            const uint32_t e0 = G(const uint32_t, P.ev)[2 * ei], e1 = G(const uint32_t, P.ev)[2 * ei + 1];
            int rl = (int)(e0 & 0xFFFF), ones = 0;
            const int type = (int)(e0 >> 16);
            while (rl >= (1 << jtab(run_idx))) { rl -= 1 << jtab(run_idx); ones++; if (run_idx < 31) run_idx++; }      // :293-300, one run at a time
            uint32_t w1 = 0, w2 = 0;
            if (type == CL_EOL) { if (rl > 0) ones++; }                             // :302-303
            else {                                                                  // :305-344
                const int jr = jtab(run_idx), glimit = p.limit - 1 - jr;
                w1 = (uint32_t)rl | (uint32_t)(jr + 1) << 24;
                if (run_idx > 0) run_idx--;
                const int v = (int)(e1 & 255), a = (int)((e1 >> 8) & 255), b = (int)((e1 >> 16) & 255);
                const int tt = a == b, sgn = (a > b) ? -1 : 1, pred = tt ? a : b;
                int e = sgn * (v - pred);
                if (e < 0) e += p.qbeta; if (e >= p.half) e -= p.qbeta;
                Ctx r = ri[tt];
                const int k = golomb_k(r.a + (tt ? (r.n >> 1) : 0), r.n);
                const int map = (e != 0) && ((e > 0) == (k == 0 && 2 * r.b < r.n));
                const int me = 2 * iabs(e) - tt - map;
                w2 = golomb_word(p, glimit, me, k);
                if (e < 0) r.b++;
                r.a += (me + 1 - tt) >> 1;
                if (r.n >= 64) { r.a >>= 1; r.b >>= 1; r.n >>= 1; }
                r.n++;
                ri[tt] = r;
            }
            G(uint32_t, P.evout)[3 * ei] = (uint32_t)ones; G(uint32_t, P.evout)[3 * ei + 1] = w1; G(uint32_t, P.evout)[3 * ei + 2] = w2;
        }
    }
}

// ---- k6a: one thread per pixel — bits it emits
JD void k6_len(const ParPlane &P, long t) {
    const int cl = P.cls[t];
    int n = 0;
    if (cl == CL_REG) n = (int)(regular_word(make_par(0), P.code[2 * (size_t)P.pos[t]], P.code[2 * (size_t)P.pos[t] + 1]) >> 24);
    else if (cl != CL_RUN) { const uint32_t e = P.pos[t]; n = (int)P.evout[3 * e] + (int)(P.evout[3 * e + 1] >> 24) + (int)(P.evout[3 * e + 2] >> 24); }
    P.len[t] = (uint8_t)n;
}
// ---- k6b: one thread per row — prefix sums inside the row; k6c: ONE thread — prefix sums over the rows
JD void k6_rowscan(const ParPlane &P, long row) {
    const size_t r0 = (size_t)row * P.w;
    uint32_t s = 0;
    for (int x0 = 0; x0 < P.w; x0 += 16) {
        uint8_t lv[16];
        for (int i = 0; i < 16; i++) lv[i] = G(const uint8_t, P.len)[r0 + imin(x0 + i, P.w - 1)];
        for (int i = 0; i < 16 && x0 + i < P.w; i++) { G(uint32_t, P.bitpos)[r0 + x0 + i] = s; s += lv[i]; }
    }
    P.rowbits[row] = s;
}
JD void k6_rows(const ParPlane &P) {
    unsigned long long s = 0;
    for (int y = 0; y < P.h; y++) { const unsigned long long v = P.rowbits[y]; P.rowbits[y] = s; s += v; }
    P.total[0] = s;
}

// n <= 32 bits of v (no bits above n) at bit position at of the MSB-first stream
#ifdef IMCVT_JLS_HOST
JD void or32(uint32_t *p, uint32_t v) { *p |= v; }
#else
JD void or32(uint32_t *p, uint32_t v) { if (v) atomicOr(p, v); }
#endif
JD void put_at(const ParPlane &P, unsigned long long at, uint32_t v, int n) {
    if (n == 0) return;
    const int off = (int)(at & 31);
    const unsigned long long w = (unsigned long long)v << (64 - n - off);          // off + n <= 63
    or32(P.bits + (at >> 5), (uint32_t)(w >> 32));
    if ((uint32_t)w) or32(P.bits + (at >> 5) + 1, (uint32_t)w);
}
// ---- k7: one thread per pixel
JD void k7_pack(const ParPlane &P, long t) {
    const int cl = P.cls[t];
    if (cl == CL_RUN || P.len[t] == 0) return;
    unsigned long long at = P.rowbits[t / P.w] + P.bitpos[t];
    if (cl == CL_REG) { const uint32_t c = regular_word(make_par(0), P.code[2 * (size_t)P.pos[t]], P.code[2 * (size_t)P.pos[t] + 1]); put_at(P, at, c & 0xFFFFFFu, (int)(c >> 24)); return; }
    const uint32_t e = P.pos[t];
    const int ones = (int)P.evout[3 * e];
    put_at(P, at, ones >= 32 ? 0xFFFFFFFFu : (1u << ones) - 1u, ones); at += ones;
    const uint32_t w1 = P.evout[3 * e + 1], w2 = P.evout[3 * e + 2];
    put_at(P, at, w1 & 0xFFFFFFu, (int)(w1 >> 24)); at += w1 >> 24;
    put_at(P, at, w2 & 0xFFFFFFu, (int)(w2 >> 24));
}

// ---- k8: bit stuffing.  Byte i of the scan takes cap_i bits of the unstuffed stream at position pos_i: cap = 7 after a
// 0xFF byte, else 8 (:156-168).  Chunk c owns the bytes that START in [c * CHUNK_BITS, (c+1) * CHUNK_BITS); its first byte
// starts 0..7 bits into the chunk with cap 7 or 8 — 16 entry states.
JD unsigned long long imin64(unsigned long long a, unsigned long long b) { return a < b ? a : b; }
JD uint32_t bits_at(const ParPlane &P, unsigned long long at, int n) {             // n <= 8 bits at `at` (zeros beyond the end)
    const unsigned long long w = ((unsigned long long)G(const uint32_t, P.bits)[at >> 5] << 32) | G(const uint32_t, P.bits)[(at >> 5) + 1];
    return (uint32_t)(w >> (64 - n - (int)(at & 31))) & ((1u << n) - 1u);
}
JD long chunks_used(const ParPlane &P) { return (long)((P.total[0] + CHUNK_BITS - 1) / CHUNK_BITS); }
// walks one chunk from entry state st (offset | (cap == 7) << 3); writes bytes when dst != null; returns exit state | bytes << 8
JD uint32_t stuff_chunk(const ParPlane &P, long c, int st, uint8_t *dst) {
    const unsigned long long T = P.total[0], end = imin64((unsigned long long)(c + 1) * CHUNK_BITS, T);
    unsigned long long at = (unsigned long long)c * CHUNK_BITS + (st & 7);
    int cap = (st & 8) ? 7 : 8;
    uint32_t n = 0;
    while (at < end) {
        const uint32_t v = bits_at(P, at, cap);
        if (dst) G(uint8_t, dst)[n] = (uint8_t)v;
        n++;
        at += cap;
        cap = (v == 0xFFu) ? 7 : 8;
    }
    // Only the LAST chunk has no successor to hand an offset to.  A byte of an earlier chunk may end past T (when T lies 1..7 bits
    // beyond the chunk edge): its successor then starts at or past T, writes nothing and passes the pending cap on.
    const int out_off = (end >= T) ? 0 : (int)(at - (unsigned long long)(c + 1) * CHUNK_BITS);
    return (uint32_t)(out_off | (cap == 7) << 3) | n << 8;
}
// one thread per (chunk, entry state)
JD void k8_simulate(const ParPlane &P, long t) { if ((t >> 4) < chunks_used(P)) P.chunk[t] = stuff_chunk(P, t >> 4, (int)(t & 15), (uint8_t *)0); }
// ONE thread: the entry state and byte offset every chunk really has; the scan's length incl. the flush rule (:170-176).
// cmax = chunks the launch was sized for (the stream's bound); the stream itself has ceil(total bits / CHUNK_BITS) of them.
JD void k8_chain(const ParPlane &P, long cmax) {
    int st = 0; unsigned long long bytes = 0;
    const long nchunk = chunks_used(P);
    for (long c = 0; c < nchunk; c++) {
        const uint32_t r = P.chunk[c * 16 + st];
        P.chunk[cmax * 16 + 2 * c] = (uint32_t)st; P.chunk[cmax * 16 + 2 * c + 1] = (uint32_t)bytes;
        bytes += r >> 8; st = (int)(r & 15);
    }
    // flushBits: a partial byte was already counted (its missing bits are zeros); a pending 7-bit byte after a final 0xFF is written empty
    if (st & 8) { P.out[P.hdr + bytes] = 0; bytes++; }
    P.total[1] = bytes;
}
// one thread per chunk
JD void k8_write(const ParPlane &P, long c, long cmax) {
    if (c < chunks_used(P)) stuff_chunk(P, c, (int)P.chunk[cmax * 16 + 2 * c], P.out + P.hdr + P.chunk[cmax * 16 + 2 * c + 1]);
}
// sizes of the work arrays of one plane (bytes), in the order of carve() below
JHD size_t par_chunks_max(size_t npx) { return (npx * 64 + CHUNK_BITS - 1) / CHUNK_BITS + 1; }     // 8 bytes per pixel: the reference's own output bound (:440)
JHD size_t par_align(size_t v) { return (v + 255) & ~(size_t)255; }
JHD size_t par_workspace(int h, int w) {
    const size_t n = (size_t)h * w;
    return par_align(2 * n) + par_align(n) + par_align(n) + par_align(2 * n) + par_align(2 * n) + par_align(4 * n) + par_align(n) + par_align(4 * n)
         + par_align(4 * (size_t)h * 364) + par_align(4 * (size_t)h) + par_align(4 * 366) + par_align(8 * (size_t)h)
         + par_align(4 * n + 128) + par_align(8 * n) + par_align(8 * n) + par_align(12 * n) + par_align(8 * n + 16) + par_align(4 * 18 * par_chunks_max(n)) + par_align(16);
}
JHD void par_carve(ParPlane &P, uint8_t *b) {
    const size_t n = (size_t)P.h * P.w;
    P.qs = (uint16_t *)b; b += par_align(2 * n);   P.med = b; b += par_align(n);   P.cls = b; b += par_align(n);
    P.rank = (uint16_t *)b; b += par_align(2 * n); P.aux = (uint16_t *)b; b += par_align(2 * n);
    P.pos = (uint32_t *)b; b += par_align(4 * n);  P.len = b; b += par_align(n);   P.bitpos = (uint32_t *)b; b += par_align(4 * n);
    P.rowcnt = (uint32_t *)b; b += par_align(4 * (size_t)P.h * 364); P.rowev = (uint32_t *)b; b += par_align(4 * (size_t)P.h);
    P.binbase = (uint32_t *)b; b += par_align(4 * 366); P.rowbits = (unsigned long long *)b; b += par_align(8 * (size_t)P.h);
    P.list = (uint32_t *)b; b += par_align(4 * n + 128); P.code = (uint32_t *)b; b += par_align(8 * n);
    P.ev = (uint32_t *)b; b += par_align(8 * n);   P.evout = (uint32_t *)b; b += par_align(12 * n);
    P.bits = (uint32_t *)b; b += par_align(8 * n + 16);
    P.chunk = (uint32_t *)b; b += par_align(4 * 18 * par_chunks_max(n)); P.total = (unsigned long long *)b;
}
JHD size_t par_bits_bytes(int h, int w) { return par_align(8 * (size_t)h * w + 16); }

}  // namespace jls
