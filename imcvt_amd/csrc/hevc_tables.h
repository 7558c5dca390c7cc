// hevc_tables.h — host-side construction of the constant tables (struct Tables) and of the stream headers.
// Everything here is generated from the standard's defining data rather than stored as big literals:
//   * transform matrices from the 32 first-column values of the 32-point core transform (reference :431-464);
//   * scan orders (reference :1126-1150) from the up-right-diagonal / horizontal / vertical rules;
//   * CABAC probability tables (reference :700-714) from H.265 rangeTabLps / transIdxLps.
#pragma once
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include "hevc_core.h"

namespace imcvt {

static const i8 kCos32[32] = { 64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67,
                               64, 61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4 };
static const i8 kDst4[16] = { 29, 55, 74, 84, 74, 74, 0, -74, 84, -29, -74, 55, 55, -84, 74, -29 };
static const u8 kRangeLps[256] = {
    128,176,208,240, 128,167,197,227, 128,158,187,216, 123,150,178,205, 116,142,169,195, 111,135,160,185, 105,128,152,175, 100,122,144,166,
     95,116,137,158,  90,110,130,150,  85,104,123,142,  81, 99,117,135,  77, 94,111,128,  73, 89,105,122,  69, 85,100,116,  66, 80, 95,110,
     62, 76, 90,104,  59, 72, 86, 99,  56, 69, 81, 94,  53, 65, 77, 89,  51, 62, 73, 85,  48, 59, 69, 80,  46, 56, 66, 76,  43, 53, 63, 72,
     41, 50, 59, 69,  39, 48, 56, 65,  37, 45, 54, 62,  35, 43, 51, 59,  33, 41, 48, 56,  32, 39, 46, 53,  30, 37, 43, 50,  29, 35, 41, 48,
     27, 33, 39, 45,  26, 31, 37, 43,  24, 30, 35, 41,  23, 28, 33, 39,  22, 27, 32, 37,  21, 26, 30, 35,  20, 24, 29, 33,  19, 23, 27, 31,
     18, 22, 26, 30,  17, 21, 25, 28,  16, 20, 23, 27,  15, 19, 22, 25,  14, 18, 21, 24,  14, 17, 20, 23,  13, 16, 19, 22,  12, 15, 18, 21,
     12, 14, 17, 20,  11, 14, 16, 19,  11, 13, 15, 18,  10, 12, 15, 17,  10, 12, 14, 16,   9, 11, 13, 15,   9, 11, 12, 14,   8, 10, 12, 14,
      8,  9, 11, 13,   7,  9, 11, 12,   7,  9, 10, 12,   7,  8, 10, 11,   6,  8,  9, 11,   6,  7,  9, 10,   6,  7,  8,  9,   2,  2,  2,  2 };
static const u8 kTransLps[64] = { 0, 0, 1, 2, 2, 4, 4, 5, 6, 7, 8, 9, 9, 11, 11, 12, 13, 13, 15, 15, 16, 16, 18, 18, 19, 19, 21, 21, 22, 22,
    23, 24, 24, 25, 26, 26, 27, 27, 28, 29, 29, 30, 30, 30, 31, 32, 32, 33, 33, 33, 34, 34, 35, 35, 35, 36, 36, 36, 37, 37, 37, 38, 38, 63 };
static const u8 kCtxInit[NCTX] = {   // I-slice initValues in the CX_* order of hevc_core.h (reference :762-776)
    139, 141, 157,  184,  184,  63,  153, 138, 138,  111, 141,  94,
    110, 110, 124, 125, 140, 153, 125, 127, 140, 109, 111, 143, 127, 111, 79,
    110, 110, 124, 125, 140, 153, 125, 127, 140, 109, 111, 143, 127, 111, 79,
    91, 171,
    111, 111, 125, 110, 110, 94, 124, 108, 124, 107, 125, 141, 179, 153, 125, 107, 125, 141, 179, 153, 125, 107, 125, 141, 179, 153, 125,
    140, 92, 137, 138, 140, 152, 138, 139, 153, 74, 149, 92, 139, 107, 122, 152,
    138, 153, 136, 167 };
static const i8 kAng[35] = { 0, 0, 32, 26, 21, 17, 13, 9, 5, 2, 0, -2, -5, -9, -13, -17, -21, -26, -32,
                             -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32 };
static const u16 kInvAng[35] = { 0, 0, 256, 315, 390, 482, 630, 910, 1638, 4096, 0, 4096, 1638, 910, 630, 482, 390, 315, 256,
                                 315, 390, 482, 630, 910, 1638, 4096, 0, 4096, 1638, 910, 630, 482, 390, 315, 256 };

inline int dct_entry(int n, int i, int j) {
    if (i == 0) return 64;
    int m = (i * (32 / n) * (2 * j + 1)) & 127;
    if (m > 64) m = 128 - m;
    return (m > 32) ? -kCos32[64 - m] : kCos32[m];
}

// The reference's per-coefficient RDOQ decision (:555-582), spelled out as it is written there: levels l0, l0-1, l0-2 priced by
// calcRDcost(dist, estimateCoeffRate(level)) (:177-185, :526-535), strictly smaller cost replaces.  Host only: the device code
// uses the thresholds derived from it below.
inline int ref_level_rate(int level) {
    static const int t[6] = { 0, 70000, 90000, 92000, 157536, 190304 };
    if (level < 6) return t[level];
    int i; level -= 6;
    for (i = 0; (1 << i) <= level; i++) level -= 1 << i;
    return 92000 + ((3 + i * 2 + 1) << 15);
}
inline int ref_rd_cost(int q, int dist, int bits) {
    static const int wd[5] = { 11, 11, 11, 5, 1 }, wb[5] = { 1, 4, 16, 29, 23 };
    const int c1 = (I32MAX / wd[q] <= dist) ? I32MAX : wd[q] * dist, c2 = (I32MAX / wb[q] <= bits) ? I32MAX : wb[q] * bits;
    return (I32MAX - c1 <= c2) ? I32MAX : c1 + c2;
}
inline int ref_rdoq_pick(int q, int s, int absval) {             // s = log2(TU size) - 2
    const int dist_sft = 8 - s, sft = 19 - s + q, add = 1 << sft >> 1, max_dlevel = I32MAX - add;
    const long long sh14 = (long long)(absval & 0x1ffff) << 14;
    const int dlevel = (absval > 0x1ffff) ? max_dlevel : (int)(sh14 < max_dlevel ? sh14 : max_dlevel);
    int level = (int)(((long long)dlevel + add) >> sft); level = level > 32767 ? 32767 : level;
    const int min_level = level - 2 > 0 ? level - 2 : 0;
    int best = I32MAX, pick = 0;
    for (; level >= min_level; level--) {
        long long e = (long long)dlevel - ((long long)level << sft); if (e < 0) e = -e;
        const int dist1 = (int)(e >> dist_sft);
        const int dist = ((dist1 < 46340) ? dist1 * dist1 : I32MAX) >> 7;
        const int cost = ref_rd_cost(q, dist, ref_level_rate(level));
        if (cost < best) { best = cost; pick = level; }
    }
    return pick;
}

inline void build_tables(Tables &T, ColdTables &K) {
    memset(&T, 0, sizeof(T)); memset(&K, 0, sizeof(K));
    for (int s = 0; s < 4; s++) {
        const int n = 4 << s, off = s == 0 ? 0 : s == 1 ? 16 : s == 2 ? 80 : 336;
        for (int i = 0; i < n; i++) for (int k = 0; k < n; k++) {
            const int v = (s == 0) ? kDst4[i * 4 + k] : dct_entry(n, i, k);
            T.C[off + i * n + k] = (i8)v;
        }
    }
    // in-group 4x4 patterns: up-right diagonal, horizontal, vertical
    int n = 0;
    for (int d = 0; d < 7; d++) for (int y = (d < 3 ? d : 3); y >= 0; y--) { const int x = d - y; if (x > 3) continue; T.incg[0][n++] = (u8)((y << 2) | x); }
    for (int k = 0; k < 16; k++) { T.incg[1][k] = (u8)(((k >> 2) << 2) | (k & 3)); T.incg[2][k] = (u8)(((k & 3) << 2) | (k >> 2)); }
    for (int t = 0; t < 3; t++) for (int k = 0; k < 16; k++) T.incg_rank[t][T.incg[t][k]] = (u8)k;
    // group orders: diagonal for every size; horizontal / vertical only exist for 8x8 TUs (2x2 groups)
    for (int s = 0; s < 4; s++) {
        const int ncg = 1 << s; int m = 0;
        for (int d = 0; d < 2 * ncg - 1; d++) for (int y = (d < ncg - 1 ? d : ncg - 1); y >= 0; y--) { const int x = d - y; if (x >= ncg) continue; T.cgpos_d[s][m++] = (u8)((y << 3) | x); }
        for (int g = 0; g < m; g++) T.cgrank_d[s][T.cgpos_d[s][g]] = (u8)g;     // index (gy<<3)|gx == gy*8+gx
    }
    { int m = 0; for (int y = 0; y < 2; y++) for (int x = 0; x < 2; x++) T.cgpos_hv[0][m++] = (u8)((y << 3) | x); }
    { int m = 0; for (int x = 0; x < 2; x++) for (int y = 0; y < 2; y++) T.cgpos_hv[1][m++] = (u8)((y << 3) | x); }
    for (int t = 0; t < 2; t++) for (int g = 0; g < 4; g++) T.cgrank_hv[t][T.cgpos_hv[t][g]] = (u8)g;
    // CABAC
    for (int p = 0; p < 128; p++) {
        const int st = p >> 1, mps = p & 1;
        const u32 next_lps = (st == 0) ? (u32)(1 - mps) : (u32)((kTransLps[st] << 1) | mps);
        T.pst[p].x = (u32)kRangeLps[st * 4] | (u32)kRangeLps[st * 4 + 1] << 8 | (u32)kRangeLps[st * 4 + 2] << 16 | (u32)kRangeLps[st * 4 + 3] << 24;
        T.pst[p].y = next_lps | (u32)((p < 124) ? p + 2 : p) << 8 | (u32)p << 16;      // (bits 16..22: the packed state itself — a context copy that holds table ENTRIES, hevc_core.h block_C8e, reads its MPS there; every other user takes bytes 0 / 1)
    }
    // sig_coeff_flag context increments per in-group scan position (reference :1115-1120)
    for (int pat = 0; pat < 4; pat++) for (int t = 0; t < 3; t++) {
        u32 w = 0;
        for (int k = 0; k < 16; k++) {
            const int yi = T.incg[t][k] >> 2, xi = T.incg[t][k] & 3; int v;
            if (pat == 0) { const int sum = yi + xi; v = sum == 0 ? 2 : sum < 3 ? 1 : 0; }
            else if (pat == 1) v = yi == 0 ? 2 : yi == 1 ? 1 : 0;
            else if (pat == 2) v = xi == 0 ? 2 : xi == 1 ? 1 : 0;
            else v = 2;
            w |= (u32)v << (2 * k);
        }
        T.posadd[pat][t] = w;
    }
    static const u8 c4[16] = { 0, 1, 4, 5, 2, 3, 4, 5, 6, 6, 8, 8, 7, 7, 8, 8 };     // reference :1092
    for (int t = 0; t < 3; t++) { u64 w = 0; for (int k = 0; k < 16; k++) w |= (u64)c4[T.incg[t][k]] << (4 * k); T.c4tab[t] = w; }
    // scan positions above n that share n's sig_coeff context in a 4x4 TU (tokg_a_fast<0, true>); position 15 is never coded as a flag
    for (int t = 0; t < 3; t++) for (int k = 0; k < 16; k++) {
        int prev[3] = { 0, 0, 0 }, np = 0;
        for (int m = k + 1; m < 15; m++) if (c4[T.incg[t][m]] == c4[T.incg[t][k]]) { if (np < 3) prev[np] = m; np++; }
        if (np > 2) { fprintf(stderr, "imcvt_hevc: a sig_coeff context of a 4x4 TU serves more than three positions\n"); abort(); }
        T.c4prev[t][k >> 2] |= (u32)(prev[0] | prev[1] << 4) << (8 * (k & 3));
    }
    for (int q = 0; q < 5; q++) {
        const int qp = q * 6 + 4;
        for (int i = 0; i < NCTX; i++) {
            const int v = kCtxInit[i];
            int st = ((((v >> 4) * 5 - 45) * qp) >> 4) + ((v & 15) << 3) - 16;
            st = st < 1 ? 1 : st > 126 ? 126 : st;
            K.ctx_init[q][i] = (st >= 64) ? (u8)(((st - 64) << 1) | 1) : (u8)((63 - st) << 1);
        }
    }
    for (int m = 0; m < 35; m++) { T.ang[m] = (u8)(kAng[m] + 32); T.iang[m] = kInvAng[m]; }
    // state hints of the 4x4 PU candidates: what a FRESH context's state is after the bins coded on it earlier in the same TU
    for (int q = 0; q < 5; q++) {
        auto next = [&](int p, int bin) { return (int)(((bin ^ p) & 1) ? (T.pst[p].y & 0xFF) : ((T.pst[p].y >> 8) & 0xFF)); };
        for (int f = 0; f < 9; f++) {
            const int p0 = K.ctx_init[q][CX_SIG + f];
            K.pu_sig[q][8 * f] = (u8)p0;
            for (int b1 = 0; b1 < 2; b1++) K.pu_sig[q][8 * f + 1 + b1] = (u8)next(p0, b1);
            for (int b2 = 0; b2 < 2; b2++) for (int b1 = 0; b1 < 2; b1++) K.pu_sig[q][8 * f + 3 + 2 * b2 + b1] = (u8)next(next(p0, b2), b1);      // b2 is coded first
        }
        K.pu_gt[q][0] = K.ctx_init[q][CX_GT1 + 1]; K.pu_gt[q][1] = K.ctx_init[q][CX_GT1 + 2];
        { int p = K.ctx_init[q][CX_GT1 + 3]; for (int k = 0; k < 6; k++) { K.pu_gt[q][2 + k] = (u8)p; p = next(p, 0); } }
        for (int n = 0; n <= 6; n++) for (int pat = 0; pat < (1 << n); pat++) {
            int p = K.ctx_init[q][CX_GT1 + 0];
            for (int i = n - 1; i >= 0; i--) p = next(p, (pat >> i) & 1);               // the first bin is the pattern's top bit
            K.pu_gt[q][8 + (1 << n) - 1 + pat] = (u8)p;
        }
    }
    // RDOQ thresholds (rdoq_group, hevc_core.h): for one level of every class, the largest remainder (in units of 2^14) at which the
    // reference's loop still prefers l0 - 1; "never" (below every remainder) where it does not occur.
    for (int q = 0; q < 5; q++) for (int s = 0; s < 4; s++) {
        const int sft = 19 - s + q, per = 1 << (sft - 14);            // |coef| values per level
        for (int c = 0; c < RQ_CLASSES; c++) {
            int thr = -(1 << 30);
            if (c >= 1 && c <= 8) {
                const int l0 = c < 8 ? c : 9;
                for (int x = -1; x >= -per / 2; x--) if (ref_rdoq_pick(q, s, l0 * per + x) < l0) { thr = x; break; }
            }
            K.rthr[q][s][c] = thr;
        }
    }
}

// VPS | SPS(+dims) | PPS | slice header (reference :664-690).  Returns the number of bytes written (<= 96).
inline int build_headers(u8 *out, int q, int hp, int wp) {
    static const u8 vps[27] = { 0, 0, 1, 0x40, 1, 0x0C, 1, 0xFF, 0xFF, 3, 0x10, 0, 0, 3, 0, 0, 3, 0, 0, 3, 0, 0, 3, 0, 0xB4, 0xF0, 0x24 };
    static const u8 sps[22] = { 0, 0, 1, 0x42, 1, 1, 3, 0x10, 0, 0, 3, 0, 0, 3, 0, 0, 3, 0, 0, 3, 0, 0xB4 };
    static const u8 pps[11] = { 0, 0, 1, 0x44, 1, 0xC0, 0x90, 0x91, 0x81, 0xD9, 0x20 };
    static const u8 slice_qp[5][2] = { {0x16, 0xDE}, {0x10, 0xDE}, {0x2B, 0x78}, {0x4D, 0xE0}, {0x97, 0x80} };
    static const u8 slice[6] = { 0, 0, 1, 0x26, 1, 0xAC };
    u8 *p = out;
    memcpy(p, vps, 27); p += 27; memcpy(p, sps, 22); p += 22;
    // MSB-first bit writer
    u64 acc = 0; int nb = 0;
    auto put = [&](u32 v, int n) { for (int i = n - 1; i >= 0; i--) { acc = (acc << 1) | ((v >> i) & 1); if (++nb == 8) { *p++ = (u8)acc; acc = 0; nb = 0; } } };
    auto ue_like = [&](int v) {             // the reference's ue(v) variant, length from v+2 (:641-647)
        int len = 1; v++;
        for (int t = v + 1; t != 1; t >>= 1) len += 2;
        put((u32)(v & ((1 << ((len + 1) >> 1)) - 1)), (len >> 1) + ((len + 1) >> 1));
    };
    put(0xA, 4); ue_like(wp); ue_like(hp); put(0x197EE4, 22); put(0x681ED1, 24);
    if (nb) { *p++ = (u8)(acc << (8 - nb)); }
    memcpy(p, pps, 11); p += 11; memcpy(p, slice, 6); p += 6;
    *p++ = slice_qp[q][0]; *p++ = slice_qp[q][1];
    return (int)(p - out);
}

}  // namespace imcvt
