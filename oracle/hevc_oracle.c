/*
 * oracle/hevc_oracle.c — CPU restatement of the ImCvt H.265 intra encoder hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker for the HIP path: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call it.  The product
 * (imcvt_amd/, libimcvt_hevc.so) never links or calls anything under oracle/.
 *
 * Parity status: PINNED.  oracle_HEVCImageEncoder() is byte-identical to the real reference
 * (oracle/_ref/libref_hevce.so, built by oracle/Makefile from /root/reference/src/HEVCe/HEVCe.c) on
 * every vector of tests/golden/hevc_kat.json (P4/P5/P6 sample images, six synthetic
 * inputs x qpd6 0..4, 1080p/4K digests) — see tests/test_oracle.py.
 *
 * It is a restatement, not a copy: same arithmetic, own structure (generated transform / scan tables,
 * compact 91-entry luma context set, byte-stack trial coders, frame-wide neighbour maps).  Each block
 * cites the reference lines (src/HEVCe/HEVCe.c unless noted) whose behaviour it reproduces.
 */
#include <stdint.h>
#include <string.h>
#include <stdlib.h>

#define CTU 32
#define NMODE 35
#define I32MAX 0x7fffffff

static inline int iabs(int v) { return v < 0 ? -v : v; }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int clip3(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int clip16(int v) { return clip3(v, -32768, 32767); }
static inline int lg2(int sz) { return sz == 4 ? 2 : sz == 8 ? 3 : sz == 16 ? 4 : 5; }

/* ------------------------------------------------------------------------------------------------
 * Tables.  Transform matrices are generated from the 32 first-column values of the HEVC 32-point
 * core transform (values as in :431-464, column 0); the 16- and 8-point matrices are its even-row
 * decimations (:399-428); the 4x4 DST is literal (:391-396).  Scan orders are generated (:1126-1150).
 * ------------------------------------------------------------------------------------------------ */
static const int8_t COS32[32] = { 64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67,
                                  64, 61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4 };
static const int8_t DST4[4][4] = { {29, 55, 74, 84}, {74, 74, 0, -74}, {84, -29, -74, 55}, {55, -84, 74, -29} };

static int TM[4][32][32];     /* TM[log2-2][i][k]  forward matrix C          */
static int TMT[4][32][32];    /* TMT[log2-2][k][i] its transpose             */
static uint8_t SCAN[3][4][1024][2];  /* [type: 0 diag,1 hor,2 ver][log2-2][pos] = {y,x} */
static int tables_ready = 0;

static int dct_entry(int n, int i, int j) {   /* n-point DCT-II integer matrix entry via the 32-pt base */
    int step = 32 / n;                         /* row i of the n-pt matrix = row i*step of the 32-pt one */
    int m = (i * step * (2 * j + 1)) & 127;    /* angle in units of pi/64 */
    if (i == 0) return 64;
    if (m > 64) m = 128 - m;
    return (m > 32) ? -COS32[64 - m] : COS32[m];
}

static void gen_scan(void) {
    /* 4x4 in-group patterns */
    int g[3][16][2], n, d, y, x, t, s, k;
    n = 0;
    for (d = 0; d < 7; d++) for (y = imin(d, 3); y >= 0; y--) { x = d - y; if (x > 3) continue; g[0][n][0] = y; g[0][n][1] = x; n++; }
    for (k = 0; k < 16; k++) { g[1][k][0] = k >> 2; g[1][k][1] = k & 3; g[2][k][0] = k & 3; g[2][k][1] = k >> 2; }
    for (t = 0; t < 3; t++) for (s = 0; s < 4; s++) {
        int ncg = (4 << s) >> 2, pos = 0, cg[64][2], m = 0;
        if (t == 0) { for (d = 0; d < 2 * ncg - 1; d++) for (y = imin(d, ncg - 1); y >= 0; y--) { x = d - y; if (x >= ncg) continue; cg[m][0] = y; cg[m][1] = x; m++; } }
        else if (t == 1) { for (y = 0; y < ncg; y++) for (x = 0; x < ncg; x++) { cg[m][0] = y; cg[m][1] = x; m++; } }
        else { for (x = 0; x < ncg; x++) for (y = 0; y < ncg; y++) { cg[m][0] = y; cg[m][1] = x; m++; } }
        for (k = 0; k < m; k++) for (n = 0; n < 16; n++) {
            SCAN[t][s][pos][0] = (uint8_t)(cg[k][0] * 4 + g[t][n][0]);
            SCAN[t][s][pos][1] = (uint8_t)(cg[k][1] * 4 + g[t][n][1]);
            pos++;
        }
    }
}

static void init_tables(void) {
    int s, i, k;
    if (tables_ready) return;
    for (s = 0; s < 4; s++) {
        int n = 4 << s;
        for (i = 0; i < n; i++) for (k = 0; k < n; k++) {
            int v = (s == 0) ? DST4[i][k] : dct_entry(n, i, k);
            TM[s][i][k] = v; TMT[s][k][i] = v;
        }
    }
    gen_scan();
    tables_ready = 1;
}

/* ------------------------------------------------------------------------------------------------
 * RD cost (:177-185) and the coefficient rate model (:526-535)
 * ------------------------------------------------------------------------------------------------ */
static const int W_DIST[5] = { 11, 11, 11, 5, 1 };
static const int W_BITS[5] = { 1, 4, 16, 29, 23 };

static int rd_cost(int q, int dist, int bits) {
    int wd = W_DIST[q], wb = W_BITS[q];
    int c1 = (I32MAX / wd <= dist) ? I32MAX : wd * dist;
    int c2 = (I32MAX / wb <= bits) ? I32MAX : wb * bits;
    return (I32MAX - c1 <= c2) ? I32MAX : c1 + c2;
}

static int level_rate(int level) {
    static const int small[6] = { 0, 70000, 90000, 92000, 157536, 190304 };
    int i;
    if (level < 6) return small[level];
    level -= 6;
    for (i = 0; (1 << i) <= level; i++) level -= 1 << i;
    return 92000 + ((3 + i * 2 + 1) << 15);
}

/* ------------------------------------------------------------------------------------------------
 * Prediction borders (:196-257) and the 35 predictors (:262-381)
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    uint8_t ucorner, fcorner;
    uint8_t uleft[64], uabove[64];   /* unfiltered: left+below-left, above+above-right */
    uint8_t fleft[64], fabove[64];   /* [1 2 1]/4 filtered                              */
} Border;

/* rec points at the block's top-left inside a plane of stride rs; [-1] row/col are the neighbours. */
static void fetch_border(Border *b, const uint8_t *rec, int rs, int sz, int has_l, int has_bl, int has_a, int has_ar) {
    int i, n = 2 * sz;
    if (has_l && has_a) b->ucorner = rec[-rs - 1];
    else if (has_l)     b->ucorner = rec[-1];
    else if (has_a)     b->ucorner = rec[-rs];
    else                b->ucorner = 128;
    for (i = 0; i < sz; i++)  b->uleft[i]  = has_l  ? rec[i * rs - 1] : b->ucorner;
    for (i = sz; i < n; i++)  b->uleft[i]  = has_bl ? rec[i * rs - 1] : b->uleft[sz - 1];
    for (i = 0; i < sz; i++)  b->uabove[i] = has_a  ? rec[-rs + i]    : b->ucorner;
    for (i = sz; i < n; i++)  b->uabove[i] = has_ar ? rec[-rs + i]    : b->uabove[sz - 1];

    b->fcorner   = (uint8_t)((2 + b->uleft[0] + b->uabove[0] + 2 * b->ucorner) >> 2);
    b->fleft[0]  = (uint8_t)((2 + 2 * b->uleft[0]  + b->uleft[1]  + b->ucorner) >> 2);
    b->fabove[0] = (uint8_t)((2 + 2 * b->uabove[0] + b->uabove[1] + b->ucorner) >> 2);
    for (i = 1; i < n - 1; i++) {
        b->fleft[i]  = (uint8_t)((2 + 2 * b->uleft[i]  + b->uleft[i - 1]  + b->uleft[i + 1])  >> 2);
        b->fabove[i] = (uint8_t)((2 + 2 * b->uabove[i] + b->uabove[i - 1] + b->uabove[i + 1]) >> 2);
    }
    b->fleft[n - 1] = b->uleft[n - 1];
    b->fabove[n - 1] = b->uabove[n - 1];
}

static const int8_t  ANG[35]  = { 0, 0, 32, 26, 21, 17, 13, 9, 5, 2, 0, -2, -5, -9, -13, -17, -21, -26, -32,
                                  -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32 };
static const int16_t IANG[35] = { 0, 0, 256, 315, 390, 482, 630, 910, 1638, 4096, 0, 4096, 1638, 910, 630, 482, 390, 315, 256,
                                  315, 390, 482, 630, 910, 1638, 4096, 0, 4096, 1638, 910, 630, 482, 390, 315, 256 };

/* which modes use the smoothed border, per block size (:274-280): a distance-from-H/V threshold */
static int uses_filtered(int sz, int mode) {
    int dist;
    if (sz == 4 || mode == 1) return 0;
    if (mode == 0) return 1;
    dist = imin(iabs(mode - 10), iabs(mode - 26));
    return dist > (sz == 8 ? 7 : sz == 16 ? 1 : 0);
}

static void predict_block(uint8_t *dst, int ds, int sz, int mode, const Border *b) {
    const int filt = uses_filtered(sz, mode);
    const int edge = sz <= 16;
    const int corner = filt ? b->fcorner : b->ucorner;
    const uint8_t *left = filt ? b->fleft : b->uleft;
    const uint8_t *above = filt ? b->fabove : b->uabove;
    int y, x;

    if (mode == 0) {
        for (y = 0; y < sz; y++) for (x = 0; x < sz; x++) {
            int h = (sz - 1 - x) * left[y] + (x + 1) * above[sz];
            int v = (sz - 1 - y) * above[x] + (y + 1) * left[sz];
            dst[y * ds + x] = (uint8_t)((sz + h + v) / (2 * sz));
        }
    } else if (mode == 1) {
        int dc = sz;
        for (x = 0; x < sz; x++) dc += left[x] + above[x];
        dc /= 2 * sz;
        for (y = 0; y < sz; y++) for (x = 0; x < sz; x++) dst[y * ds + x] = (uint8_t)dc;
        if (edge) {
            dst[0] = (uint8_t)((2 + 2 * dc + left[0] + above[0]) >> 2);
            for (x = 1; x < sz; x++) {
                dst[x] = (uint8_t)((2 + 3 * dc + above[x]) >> 2);
                dst[x * ds] = (uint8_t)((2 + 3 * dc + left[x]) >> 2);
            }
        }
    } else if (mode == 10) {
        for (y = 0; y < sz; y++) for (x = 0; x < sz; x++) dst[y * ds + x] = left[y];
        if (edge) for (x = 0; x < sz; x++) dst[x] = (uint8_t)clip3(((above[x] - corner) >> 1) + left[0], 0, 255);
    } else if (mode == 26) {
        for (y = 0; y < sz; y++) for (x = 0; x < sz; x++) dst[y * ds + x] = above[x];
        if (edge) for (y = 0; y < sz; y++) dst[y * ds] = (uint8_t)clip3(((left[y] - corner) >> 1) + above[0], 0, 255);
    } else {
        const int horiz = mode < 18;
        const int ang = ANG[mode], iang = IANG[mode];
        const uint8_t *mainb = horiz ? left : above, *sideb = horiz ? above : left;
        uint8_t line0[160], *line = line0 + 72;           /* line[0] = corner, line[1+i] = main, line[-k] = projected side */
        int i, j, last = (sz * ang) >> 5;
        line[0] = (uint8_t)corner;
        for (i = -1; i > last; i--) {
            int k = (128 - iang * i) >> 8;                /* 1-based index into the side border */
            line[i] = sideb[k - 1];
        }
        for (i = 0; i < 2 * sz; i++) line[1 + i] = mainb[i];
        line[1 + 2 * sz] = 0;                             /* read only with weight 0 (:372) */
        for (i = 0; i < sz; i++) {
            int off = ang * (i + 1), oi = off >> 5, of = off & 31;
            for (j = 0; j < sz; j++) {
                int p = ((32 - of) * line[oi + j + 1] + of * line[oi + j + 2] + 16) >> 5;
                if (horiz) dst[j * ds + i] = (uint8_t)p; else dst[i * ds + j] = (uint8_t)p;
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * Transforms (:469-516).  Blocks are int[sz*sz], row-major, stride sz.
 * ------------------------------------------------------------------------------------------------ */
static void fwd_transform(int sz, const int *src, int *dst) {
    const int s = lg2(sz) - 2, a = s + 1, b = a + 7;
    const int ra = 1 << a >> 1, rb = 1 << b >> 1;
    int tmp[32 * 32], i, j, k;
    for (i = 0; i < sz; i++) for (j = 0; j < sz; j++) tmp[i * sz + j] = ra;
    for (i = 0; i < sz; i++) for (k = 0; k < sz; k++) { int c = TM[s][i][k]; for (j = 0; j < sz; j++) tmp[i * sz + j] += c * src[k * sz + j]; }
    for (i = 0; i < sz * sz; i++) tmp[i] >>= a;
    for (i = 0; i < sz; i++) for (j = 0; j < sz; j++) {
        int acc = rb; for (k = 0; k < sz; k++) acc += tmp[i * sz + k] * TM[s][j][k];
        dst[i * sz + j] = acc >> b;
    }
}

static void inv_transform(int sz, const int *src, int *dst) {
    const int s = lg2(sz) - 2;
    int tmp[32 * 32], i, j, k;
    for (i = 0; i < sz; i++) for (j = 0; j < sz; j++) tmp[i * sz + j] = 64;
    for (i = 0; i < sz; i++) for (k = 0; k < sz; k++) { int c = TMT[s][i][k]; for (j = 0; j < sz; j++) tmp[i * sz + j] += c * src[k * sz + j]; }
    for (i = 0; i < sz * sz; i++) tmp[i] = clip16(tmp[i] >> 7);
    for (i = 0; i < sz; i++) for (j = 0; j < sz; j++) {
        int acc = 2048; for (k = 0; k < sz; k++) acc += tmp[i * sz + k] * TMT[s][j][k];
        dst[i * sz + j] = clip16(acc >> 12);
    }
}

/* ------------------------------------------------------------------------------------------------
 * Simplified RDOQ (:540-594) and dequantisation (:599-614)
 * ------------------------------------------------------------------------------------------------ */
static void rdoq(int q, int sz, const int *coef, int *lvl) {
    const int s = lg2(sz) - 2;
    const int dsh = 8 - s, sh = 19 - s + q, add = 1 << sh >> 1;
    const int dmax = I32MAX - add, thr = 9 << sh >> 2;
    int gy, gx, y, x;
    for (gy = 0; gy < sz; gy += 4) for (gx = 0; gx < sz; gx += 4) {
        int sum = 0;
        for (y = gy; y < gy + 4; y++) for (x = gx; x < gx + 4; x++) {
            int c = coef[y * sz + x], a = iabs(c);
            int d = (a > 0x1ffff) ? dmax : imin((a & 0x1ffff) << 14, dmax);
            int l = clip16((d + add) >> sh), lo = imax(0, l - 2), best = I32MAX, pick = 0;
            for (; l >= lo; l--) {
                int e = iabs(d - (l << sh)) >> dsh;
                int dist = ((e < 46340) ? e * e : I32MAX) >> 7;
                int cost = rd_cost(q, dist, level_rate(l));
                if (cost < best) { best = cost; pick = l; }
            }
            lvl[y * sz + x] = (c < 0) ? -pick : pick;
            sum += imin(d, thr);
        }
        if (sum < thr) for (y = gy; y < gy + 4; y++) for (x = gx; x < gx + 4; x++) lvl[y * sz + x] = 0;
    }
}

static void dequant(int q, int sz, const int *lvl, int *out) {
    const int mul = 1 << (7 - lg2(sz) + q);      /* shift 5,4,3,2 (+q) for 4,8,16,32 */
    int i;
    for (i = 0; i < sz * sz; i++) out[i] = clip16(lvl[i] * mul);
}

/* one candidate pipeline: predict -> residual -> T -> RDOQ -> deQ -> T^-1 -> recon; returns SSE (:1425-1436) */
static int run_candidate(int q, int sz, int mode, const Border *b, const uint8_t *org, int os,
                         int *lvl, uint8_t *rec, int rs) {
    uint8_t pred[32 * 32];
    int res[32 * 32], y, x, sse = 0;
    predict_block(pred, sz, sz, mode, b);
    for (y = 0; y < sz; y++) for (x = 0; x < sz; x++) res[y * sz + x] = (int)org[y * os + x] - pred[y * sz + x];
    fwd_transform(sz, res, res);
    rdoq(q, sz, res, lvl);
    dequant(q, sz, lvl, res);
    inv_transform(sz, res, res);
    for (y = 0; y < sz; y++) for (x = 0; x < sz; x++) {
        int r = clip3(res[y * sz + x] + pred[y * sz + x], 0, 255), d = (int)org[y * os + x] - r;
        rec[y * rs + x] = (uint8_t)r;
        sse += d * d;
    }
    return sse;
}

/* ------------------------------------------------------------------------------------------------
 * CABAC engine (:700-932).  Probability state is packed (state<<1 | mps) as in the reference.
 * Tables are the standard H.265 rangeTabLps / transIdxLps.
 * ------------------------------------------------------------------------------------------------ */
static const uint8_t RANGE_LPS[64 * 4] = {
    128,176,208,240, 128,167,197,227, 128,158,187,216, 123,150,178,205, 116,142,169,195, 111,135,160,185, 105,128,152,175, 100,122,144,166,
     95,116,137,158,  90,110,130,150,  85,104,123,142,  81, 99,117,135,  77, 94,111,128,  73, 89,105,122,  69, 85,100,116,  66, 80, 95,110,
     62, 76, 90,104,  59, 72, 86, 99,  56, 69, 81, 94,  53, 65, 77, 89,  51, 62, 73, 85,  48, 59, 69, 80,  46, 56, 66, 76,  43, 53, 63, 72,
     41, 50, 59, 69,  39, 48, 56, 65,  37, 45, 54, 62,  35, 43, 51, 59,  33, 41, 48, 56,  32, 39, 46, 53,  30, 37, 43, 50,  29, 35, 41, 48,
     27, 33, 39, 45,  26, 31, 37, 43,  24, 30, 35, 41,  23, 28, 33, 39,  22, 27, 32, 37,  21, 26, 30, 35,  20, 24, 29, 33,  19, 23, 27, 31,
     18, 22, 26, 30,  17, 21, 25, 28,  16, 20, 23, 27,  15, 19, 22, 25,  14, 18, 21, 24,  14, 17, 20, 23,  13, 16, 19, 22,  12, 15, 18, 21,
     12, 14, 17, 20,  11, 14, 16, 19,  11, 13, 15, 18,  10, 12, 15, 17,  10, 12, 14, 16,   9, 11, 13, 15,   9, 11, 12, 14,   8, 10, 12, 14,
      8,  9, 11, 13,   7,  9, 11, 12,   7,  9, 10, 12,   7,  8, 10, 11,   6,  8,  9, 11,   6,  7,  9, 10,   6,  7,  8,  9,   2,  2,  2,  2 };
static const uint8_t TRANS_LPS[64] = { 0, 0, 1, 2, 2, 4, 4, 5, 6, 7, 8, 9, 9, 11, 11, 12, 13, 13, 15, 15, 16, 16, 18, 18, 19, 19, 21, 21, 22, 22,
    23, 24, 24, 25, 26, 26, 27, 27, 28, 29, 29, 30, 30, 30, 31, 32, 32, 33, 33, 33, 34, 34, 35, 35, 35, 36, 36, 36, 37, 37, 37, 38, 38, 63 };

static inline uint8_t next_lps(uint8_t p) { int st = p >> 1, mps = p & 1; return st == 0 ? (uint8_t)(1 - mps) : (uint8_t)((TRANS_LPS[st] << 1) | mps); }
static inline uint8_t next_mps(uint8_t p) { return (p >> 1) < 62 ? (uint8_t)(p + 2) : p; }

/* luma-only context layout (the reference's 142-byte ContextSet :744-758 minus entries this path never touches) */
enum { CX_SPLIT_CU = 0, CX_PART = 3, CX_PREV_INTRA = 4, CX_CHROMA_PRED = 5, CX_SPLIT_TU = 6, CX_CBF_LUMA = 9, CX_CBF_CHROMA = 11,
       CX_LAST_X = 12, CX_LAST_Y = 27, CX_CSBF = 42, CX_SIG = 44, CX_GT1 = 71, CX_GT2 = 87, NCTX = 91 };
static const uint8_t CTX_INIT[NCTX] = {   /* I-slice initValues (:762-776) */
    139, 141, 157,  184,  184,  63,  153, 138, 138,  111, 141,  94,
    110, 110, 124, 125, 140, 153, 125, 127, 140, 109, 111, 143, 127, 111, 79,
    110, 110, 124, 125, 140, 153, 125, 127, 140, 109, 111, 143, 127, 111, 79,
    91, 171,
    111, 111, 125, 110, 110, 94, 124, 108, 124, 107, 125, 141, 179, 153, 125, 107, 125, 141, 179, 153, 125, 107, 125, 141, 179, 153, 125,
    140, 92, 137, 138, 140, 152, 138, 139, 153, 74, 149, 92, 139, 107, 122, 152,
    138, 153, 136, 167 };
static const uint8_t LAST_CTX_BASE[4] = { 0, 3, 6, 10 };   /* per block size 4,8,16,32 (:1050) */

typedef struct { int range, low, nbits, nbytes, bufbyte, zeros, cnt; } Arith;
typedef struct { Arith a; uint8_t cx[NCTX]; uint8_t *buf; } Coder;   /* buf: byte stack of the current CTU */

static void ctx_reset(uint8_t *cx, int q) {       /* :726-734 */
    int i, qp = q * 6 + 4;
    for (i = 0; i < NCTX; i++) {
        int v = CTX_INIT[i], st = ((((v >> 4) * 5 - 45) * qp) >> 4) + ((v & 15) << 3) - 16;
        st = clip3(st, 1, 126);
        cx[i] = (st >= 64) ? (uint8_t)(((st - 64) << 1) | 1) : (uint8_t)((63 - st) << 1);
    }
}
static void arith_reset(Arith *a) { a->range = 510; a->low = 0; a->nbits = 23; a->nbytes = 0; a->bufbyte = 0xFF; a->zeros = 0; a->cnt = 0; }
static inline int arith_len(const Arith *a) { return 8 * (a->cnt + a->nbytes) + 23 - a->nbits; }   /* :834 */

static void emit_byte(Coder *c, int v) {          /* with emulation prevention (:820-831) */
    v &= 0xFF;
    if (c->a.zeros >= 2 && v <= 3) { c->buf[c->a.cnt++] = 3; c->a.zeros = 0; }
    c->buf[c->a.cnt++] = (uint8_t)v;
    c->a.zeros = v ? 0 : c->a.zeros + 1;
}
static void carry_out(Coder *c) {                 /* :858-878 */
    Arith *a = &c->a;
    if (a->nbits >= 12) return;
    {
        int lead = a->low >> (24 - a->nbits);
        a->nbits += 8;
        a->low &= (int)(0xFFFFFFFFu >> a->nbits);
        if (lead == 0xFF) a->nbytes++;
        else if (a->nbytes > 0) {
            int carry = lead >> 8, v = a->bufbyte + carry;
            a->bufbyte = lead & 0xFF;
            emit_byte(c, v);
            v = (0xFF + carry) & 0xFF;
            for (; a->nbytes > 1; a->nbytes--) emit_byte(c, v);
        } else { a->nbytes = 1; a->bufbyte = lead; }
    }
}
static void put_bin(Coder *c, int bin, int ci) {  /* :913-932 */
    Arith *a = &c->a;
    uint8_t p = c->cx[ci];
    int lps = RANGE_LPS[(p >> 1) * 4 + ((a->range >> 6) & 3)];
    a->range -= lps;
    if ((bin != 0) != (p & 1)) {
        int sh = 6, t = lps >> 3;                 /* renorm count table (:714) == 8 - floor(log2 lps), capped at 6 */
        while (t) { sh--; t >>= 1; }
        if (lps < 8) sh = 6;
        c->cx[ci] = next_lps(p);
        a->low = (a->low + a->range) << sh;
        a->range = lps << sh;
        a->nbits -= sh;
    } else {
        c->cx[ci] = next_mps(p);
        if (a->range < 256) { a->low <<= 1; a->range <<= 1; a->nbits--; }
    }
    carry_out(c);
}
static void put_bypass(Coder *c, int bits, int len) {   /* :898-910, <=8 bins per renormalisation */
    Arith *a = &c->a;
    bits &= (1 << len) - 1;
    while (len > 0) {
        int n = imin(len, 8);
        len -= n;
        a->low = (a->low << n) + a->range * ((bits >> len) & ((1 << n) - 1));
        a->nbits -= n;
        carry_out(c);
    }
}
static void put_terminate(Coder *c, int bin) {    /* :881-895 */
    Arith *a = &c->a;
    a->range -= 2;
    if (bin) { a->low = (a->low + a->range) << 7; a->range = 256; a->nbits -= 7; }
    else if (a->range < 256) { a->low <<= 1; a->range <<= 1; a->nbits--; }
    carry_out(c);
}
static void arith_finish(Coder *c) {              /* :839-855 */
    Arith *a = &c->a;
    int fill = 0, t;
    if ((a->low >> (32 - a->nbits)) > 0) { emit_byte(c, a->bufbyte + 1); a->low -= 1 << (32 - a->nbits); }
    else { if (a->nbytes > 0) emit_byte(c, a->bufbyte); fill = 0xFF; }
    for (; a->nbytes > 1; a->nbytes--) emit_byte(c, fill);
    t = (a->low >> 8) << a->nbits;
    emit_byte(c, t >> 16); emit_byte(c, t >> 8); emit_byte(c, t);
}

/* ------------------------------------------------------------------------------------------------
 * Syntax (:942-1339)
 * ------------------------------------------------------------------------------------------------ */
static void mpm_list(int l, int a, int *m) {      /* :957-976 */
    if (l != a) { m[0] = l; m[1] = a; m[2] = (l != 0 && a != 0) ? 0 : (l + a < 2) ? 26 : 1; }
    else if (l > 1) { m[0] = l; m[1] = ((l + 29) % 32) + 2; m[2] = ((l - 1) % 32) + 2; }
    else { m[0] = 0; m[1] = 1; m[2] = 26; }
}

/* prev_intra_luma_pred_flag for all PUs first, then mpm_idx / rem_intra_luma_pred_mode (:984-1017) */
static void put_luma_modes(Coder *c, int n, const int *mode, const int *ml, const int *ma) {
    int mpm[4][3], hit[4], i, j;
    for (i = 0; i < n; i++) {
        mpm_list(ml[i], ma[i], mpm[i]);
        hit[i] = -1;
        for (j = 0; j < 3; j++) if (mpm[i][j] == mode[i]) hit[i] = j;
        put_bin(c, hit[i] >= 0, CX_PREV_INTRA);
    }
    for (i = 0; i < n; i++) {
        if (hit[i] >= 0) {
            put_bypass(c, hit[i] > 0, 1);
            if (hit[i] > 0) put_bypass(c, hit[i] - 1, 1);
        } else {
            int r = mode[i], t, *p = mpm[i];
            if (p[0] < p[1]) { t = p[0]; p[0] = p[1]; p[1] = t; }
            if (p[1] < p[2]) { t = p[1]; p[1] = p[2]; p[2] = t; }
            if (p[0] < p[1]) { t = p[0]; p[0] = p[1]; p[1] = t; }
            for (j = 0; j < 3; j++) if (r > p[j]) r--;
            put_bypass(c, r, 5);
        }
    }
}

static int scan_type_of(int sz, int mode) {       /* :1133-1141 */
    if (sz <= 8) { if (iabs(mode - 26) <= 4) return 1; if (iabs(mode - 10) <= 4) return 2; }
    return 0;
}

static void put_last_pos(Coder *c, int sz, int st, int y, int x) {     /* :1045-1086 */
    static const uint8_t grp[32] = { 0, 1, 2, 3, 4, 4, 5, 5, 6, 6, 6, 6, 7, 7, 7, 7, 8, 8, 8, 8, 8, 8, 8, 8, 9, 9, 9, 9, 9, 9, 9, 9 };
    static const uint8_t gmin[10] = { 0, 1, 2, 3, 4, 6, 8, 12, 16, 24 };
    const int s = lg2(sz) - 2, base = LAST_CTX_BASE[s], sh = (s == 0) ? 0 : 1, gmax = grp[sz - 1];
    int ty = (st == 2) ? x : y, tx = (st == 2) ? y : x, gy = grp[ty], gx = grp[tx], i;
    for (i = 0; i < gx; i++) put_bin(c, 1, CX_LAST_X + base + (i >> sh));
    if (gx < gmax) put_bin(c, 0, CX_LAST_X + base + (gx >> sh));
    for (i = 0; i < gy; i++) put_bin(c, 1, CX_LAST_Y + base + (i >> sh));
    if (gy < gmax) put_bin(c, 0, CX_LAST_Y + base + (gy >> sh));
    if (gx > 3) { tx -= gmin[gx]; for (i = ((gx - 2) >> 1) - 1; i >= 0; i--) put_bypass(c, (tx >> i) & 1, 1); }
    if (gy > 3) { ty -= gmin[gy]; for (i = ((gy - 2) >> 1) - 1; i >= 0; i--) put_bypass(c, (ty >> i) & 1, 1); }
}

static int sig_ctx_index(int sz, int st, int y, int x, int pat) {      /* luma branch of :1091-1121 */
    static const uint8_t c4[16] = { 0, 1, 4, 5, 2, 3, 4, 5, 6, 6, 8, 8, 7, 7, 8, 8 };
    int ci, yi = y & 3, xi = x & 3, v;
    if (y == 0 && x == 0) return 0;
    if (sz == 4) return c4[y * 4 + x];
    ci = 9;
    if (sz >= 16) ci += 12;
    if (sz == 8 && st != 0) ci += 6;
    if ((y >> 2) || (x >> 2)) ci += 3;
    switch (pat) {
        case 0: v = yi + xi; return ci + (v == 0 ? 2 : v < 3 ? 1 : 0);
        case 1: return ci + (yi == 0 ? 2 : yi == 1 ? 1 : 0);
        case 2: return ci + (xi == 0 ? 2 : xi == 1 ? 1 : 0);
        default: return ci + 2;
    }
}

static void put_remaining(Coder *c, int v, int k) {   /* coeff_abs_level_remaining (:1153-1168) */
    if (v < (3 << k)) {
        int p = v >> k;
        put_bypass(c, (1 << (p + 1)) - 2, p + 1);
        put_bypass(c, v & ((1 << k) - 1), k);
    } else {
        int n = k, t;
        v -= 3 << k;
        for (; v >= (1 << n); n++) v -= 1 << n;
        t = 4 + n - k;
        put_bypass(c, (1 << t) - 2, t);
        put_bypass(c, v, n);
    }
}

/* residual_coding of one TU (:1172-1268); lvl is int[sz*sz] */
static void put_residual(Coder *c, int sz, int mode, const int *lvl) {
    const int st = scan_type_of(sz, mode), s = lg2(sz) - 2, ncg = sz >> 2;
    const uint8_t (*sc)[2] = SCAN[st][s];
    uint8_t cgsig[8][8];
    int i, last = 0, nnz = 0, signs = 0, pat = 0, c1 = 1, mag[16];
    memset(cgsig, 0, sizeof cgsig);
    for (i = 0; i < sz * sz; i++) if (lvl[sc[i][0] * sz + sc[i][1]]) { cgsig[sc[i][0] >> 2][sc[i][1] >> 2] = 1; last = i; }
    put_last_pos(c, sz, st, sc[last][0], sc[last][1]);

    for (i = last; i >= 0; i--) {
        const int y = sc[i][0], x = sc[i][1], gy = y >> 2, gx = x >> 2, v = lvl[y * sz + x];
        const int dc_group = (gy | gx) == 0, is_last = (i == last), lowest = (i & 15) == 0;
        const int coded_group = cgsig[gy][gx];
        if ((i & 15) == 15 || is_last) {                     /* entering a coefficient group */
            int right = gx < ncg - 1 && cgsig[gy][gx + 1], below = gy < ncg - 1 && cgsig[gy + 1][gx];
            pat = (below << 1) | right; nnz = 0; signs = 0;
            if (!dc_group && !is_last) put_bin(c, coded_group, CX_CSBF + (pat != 0));
        }
        if (!is_last && (dc_group || (coded_group && (!lowest || nnz > 0))))
            put_bin(c, v != 0, CX_SIG + sig_ctx_index(sz, st, y, x, pat));
        if (v) { mag[nnz++] = iabs(v); signs = (signs << 1) | (v < 0); }

        if (lowest && nnz > 0) {                             /* levels of this group */
            const int set = (dc_group ? 0 : 2) + (c1 == 0);
            int j, esc = nnz > 8, g2 = -1;
            c1 = 1;
            for (j = 0; j < 8 && j < nnz; j++) {
                put_bin(c, mag[j] > 1, CX_GT1 + 4 * set + c1);
                if (mag[j] > 1) { c1 = 0; if (g2 < 0) g2 = mag[j] > 2; else esc = 1; }
                else if (c1 > 0 && c1 < 3) c1++;
            }
            if (c1 == 0 && g2 >= 0) { put_bin(c, g2, CX_GT2 + set); esc |= g2; }
            put_bypass(c, signs, nnz);
            if (esc) {
                int base2 = 3, rice = 0;
                for (j = 0; j < nnz; j++) {
                    int r = mag[j] - (j < 8 ? base2 : 1);
                    if (r >= 0) { put_remaining(c, r, rice); if (mag[j] > (3 << rice)) rice = imin(rice + 1, 4); }
                    if (mag[j] >= 2) base2 = 2;
                }
            }
        }
    }
}

static int any_nonzero(const int *lvl, int n) { int i; for (i = 0; i < n; i++) if (lvl[i]) return 1; return 0; }

static void put_split_cu(Coder *c, int sz, int flag, int big_l, int big_a) { if (sz >= 16) put_bin(c, flag, CX_SPLIT_CU + big_l + big_a); }

/* coding_unit() for the three shapes (:1271-1339).  shape 0: 2Nx2N one TU, 1: 2Nx2N four TUs, 2: NxN.
 * lv[k] are int[(sz or sz/2)^2] level blocks. */
static void put_cu(Coder *c, int sz, int shape, const int *mode, const int *ml, const int *ma, int *const *lv) {
    int k, h = sz / 2;
    if (sz == 8) put_bin(c, shape != 2, CX_PART);
    put_luma_modes(c, shape == 2 ? 4 : 1, mode, ml, ma);
    put_bin(c, 0, CX_CHROMA_PRED);
    if (shape != 2) put_bin(c, shape == 1, CX_SPLIT_TU + (5 - lg2(sz)));
    put_bin(c, 0, CX_CBF_CHROMA); put_bin(c, 0, CX_CBF_CHROMA);
    if (shape == 0) {
        int cbf = any_nonzero(lv[0], sz * sz);
        put_bin(c, cbf, CX_CBF_LUMA + 1);
        if (cbf) put_residual(c, sz, mode[0], lv[0]);
    } else for (k = 0; k < 4; k++) {
        int cbf = any_nonzero(lv[k], h * h);
        put_bin(c, cbf, CX_CBF_LUMA + 0);
        if (cbf) put_residual(c, h, mode[shape == 2 ? k : 0], lv[k]);
    }
}

/* ------------------------------------------------------------------------------------------------
 * Encoder state and the exhaustive CU search (:1349-1559)
 * ------------------------------------------------------------------------------------------------ */
#define RS 65   /* recon tile stride: 1 border column + 64 (own 32 + above-right 32) */

typedef struct {
    int q, wp, hp;                       /* qpd6, padded size */
    uint8_t org[CTU][CTU];
    uint8_t rec0[(CTU + 1) * RS];        /* rec0[(y+1)*RS + (x+1)] */
    uint8_t *mapsz, *mapmode; int ms;    /* frame-wide 4x4-unit maps with a 1-cell top/left apron, stride ms */
    int cy, cx;                          /* CTU origin in 4x4 units */
    Coder live;
    uint8_t stack[4096];                 /* bytes of the current CTU (TMPBUF_LEN is 3200, :794) */
    uint8_t trial_bytes[4096];
    /* optional decision trace */
    int *trace; int trace_cap, trace_n;
} Enc;

static inline uint8_t *rec_at(Enc *e, int y, int x) { return e->rec0 + (y + 1) * RS + (x + 1); }

static void trace_cu(Enc *e, int y0, int x0, int sz, int kind, int m0, int cost) {
    if (e->trace && e->trace_n + 8 <= e->trace_cap) {
        int *t = e->trace + e->trace_n;
        t[0] = e->cy * 4 + y0; t[1] = e->cx * 4 + x0; t[2] = sz; t[3] = kind; t[4] = m0; t[5] = cost;
        t[6] = arith_len(&e->live.a); t[7] = e->live.a.range;
        e->trace_n += 8;
    }
}

/* neighbour context: above across a CTU-row boundary keeps the CU size but forgets the mode (DC), :1633-1636 */
static int nb_size(Enc *e, int uy, int ux) { return e->mapsz[(uy + 1) * e->ms + ux + 1]; }
static int nb_mode(Enc *e, int uy, int ux) { return (uy < e->cy) ? 1 : e->mapmode[(uy + 1) * e->ms + ux + 1]; }
static void fill_map(Enc *e, uint8_t *map, int y0, int x0, int sz, int v) {
    int i, j, n = sz / 4, uy = e->cy + y0 / 4, ux = e->cx + x0 / 4;
    for (i = 0; i < n; i++) for (j = 0; j < n; j++) map[(uy + i + 1) * e->ms + ux + j + 1] = (uint8_t)v;
}

static void commit(Enc *e, const Coder *t, int entry_cnt) {
    memcpy(e->stack + entry_cnt, t->buf + entry_cnt, (size_t)(t->a.cnt - entry_cnt));
    e->live.a = t->a;
    memcpy(e->live.cx, t->cx, NCTX);
}

static void search_cu(Enc *e, int y0, int x0, int sz, int has_l, int has_bl, int has_a, int has_ar) {
    const int q = e->q, h = sz / 2;
    const int uy = e->cy + y0 / 4, ux = e->cx + x0 / 4;
    const Coder entry = e->live;
    const int len0 = arith_len(&entry.a), cnt0 = entry.a.cnt;
    const int big_l = sz > nb_size(e, uy, ux - 1), big_a = sz > nb_size(e, uy - 1, ux);
    const int ml = nb_mode(e, uy, ux - 1), ma = nb_mode(e, uy - 1, ux);
    /* Z-order availability of the four quadrants (:1375-1378) */
    const int sl[4] = { has_l, 1, has_l, 1 }, sbl[4] = { has_l, 0, has_bl, 0 };
    const int sa[4] = { has_a, has_a, 1, 1 }, sar[4] = { has_a, has_ar, 1, 0 };
    const int qy[4] = { y0, y0, y0 + h, y0 + h }, qx[4] = { x0, x0 + h, x0, x0 + h };
    uint8_t *const rec = rec_at(e, y0, x0);
    const uint8_t *const org = &e->org[y0][x0];
    uint8_t keep[32 * 32], cand[32 * 32];
    int lvl_local[4][32 * 32];     /* level blocks of the candidate being priced */
    int *lp[4];
    Border b;
    Coder t;
    int best = I32MAX, k, m, y, x, sse, cost;
    for (k = 0; k < 4; k++) lp[k] = lvl_local[k];
    t.buf = e->trial_bytes;

    if (sz > 8) {                                              /* candidate 0: split into four CUs, on the live coder */
        put_split_cu(&e->live, sz, 1, big_l, big_a);
        for (k = 0; k < 4; k++) search_cu(e, qy[k], qx[k], h, sl[k], sbl[k], sa[k], sar[k]);
        sse = 0;
        for (y = 0; y < sz; y++) for (x = 0; x < sz; x++) { int d = (int)org[y * CTU + x] - rec[y * RS + x]; sse += d * d; keep[y * sz + x] = rec[y * RS + x]; }
        best = rd_cost(q, sse, arith_len(&e->live.a) - len0);
    }

    fetch_border(&b, rec, RS, sz, has_l, has_bl, has_a, has_ar);
    for (m = 0; m < NMODE; m++) {                              /* 2Nx2N, one TU */
        sse = run_candidate(q, sz, m, &b, org, CTU, lp[0], cand, sz);
        t.a = entry.a; memcpy(t.cx, entry.cx, NCTX);
        put_split_cu(&t, sz, 0, big_l, big_a);
        put_cu(&t, sz, 0, &m, &ml, &ma, lp);
        cost = rd_cost(q, sse, arith_len(&t.a) - len0);
        if (best >= cost) {
            best = cost; commit(e, &t, cnt0); memcpy(keep, cand, (size_t)sz * sz);
            fill_map(e, e->mapsz, y0, x0, sz, sz); fill_map(e, e->mapmode, y0, x0, sz, m);
        }
    }

    for (m = 0; m < NMODE; m++) {                              /* 2Nx2N, four TUs reconstructed in place */
        for (k = 0; k < 4; k++) {
            uint8_t *r = rec_at(e, qy[k], qx[k]);
            fetch_border(&b, r, RS, h, sl[k], sbl[k], sa[k], sar[k]);
            run_candidate(q, h, m, &b, &e->org[qy[k]][qx[k]], CTU, lp[k], r, RS);
        }
        t.a = entry.a; memcpy(t.cx, entry.cx, NCTX);
        put_split_cu(&t, sz, 0, big_l, big_a);
        put_cu(&t, sz, 1, &m, &ml, &ma, lp);
        sse = 0;
        for (y = 0; y < sz; y++) for (x = 0; x < sz; x++) { int d = (int)org[y * CTU + x] - rec[y * RS + x]; sse += d * d; }
        cost = rd_cost(q, sse, arith_len(&t.a) - len0);
        if (best >= cost) {
            best = cost; commit(e, &t, cnt0);
            for (y = 0; y < sz; y++) memcpy(keep + y * sz, rec + y * RS, (size_t)sz);
            fill_map(e, e->mapsz, y0, x0, sz, sz); fill_map(e, e->mapmode, y0, x0, sz, m);
        }
    }

    if (sz == 8) {                                             /* NxN: PU modes picked on a fresh coder, residual bits only */
        int pm[4], pl[4], pa[4], tl[16];
        uint8_t fresh_bytes[256];
        for (k = 0; k < 4; k++) {
            uint8_t *r = rec_at(e, qy[k], qx[k]);
            int pbest = I32MAX;
            fetch_border(&b, r, RS, 4, sl[k], sbl[k], sa[k], sar[k]);
            for (m = 0; m < NMODE; m++) {
                Coder f;
                f.buf = fresh_bytes; arith_reset(&f.a); ctx_reset(f.cx, q);
                sse = run_candidate(q, 4, m, &b, &e->org[qy[k]][qx[k]], CTU, tl, cand, 4);
                put_residual(&f, 4, m, tl);
                cost = rd_cost(q, sse, arith_len(&f.a));
                if (pbest >= cost) {
                    pbest = cost; pm[k] = m; memcpy(lp[k], tl, sizeof tl);
                    for (y = 0; y < 4; y++) memcpy(r + y * RS, cand + y * 4, 4);
                }
            }
        }
        pl[0] = ml;                       pa[0] = ma;
        pl[1] = pm[0];                    pa[1] = nb_mode(e, uy - 1, ux + 1);
        pl[2] = nb_mode(e, uy + 1, ux - 1); pa[2] = pm[0];
        pl[3] = pm[2];                    pa[3] = pm[1];
        t.a = entry.a; memcpy(t.cx, entry.cx, NCTX);
        put_split_cu(&t, sz, 0, big_l, big_a);
        put_cu(&t, sz, 2, pm, pl, pa, lp);
        sse = 0;
        for (y = 0; y < sz; y++) for (x = 0; x < sz; x++) { int d = (int)org[y * CTU + x] - rec[y * RS + x]; sse += d * d; }
        cost = rd_cost(q, sse, arith_len(&t.a) - len0);
        if (best >= cost) {
            commit(e, &t, cnt0);
            fill_map(e, e->mapsz, y0, x0, sz, sz);
            for (k = 0; k < 4; k++) fill_map(e, e->mapmode, qy[k], qx[k], 4, pm[k]);
            trace_cu(e, y0, x0, sz, 3, pm[0] | pm[1] << 8 | pm[2] << 16 | pm[3] << 24, cost);
            return;                                            /* the in-place PU reconstructions are the CU's (:1554) */
        }
    }
    for (y = 0; y < sz; y++) memcpy(rec + y * RS, keep + y * sz, (size_t)sz);
    trace_cu(e, y0, x0, sz, nb_size(e, uy, ux) == sz ? 1 : 0, nb_mode(e, uy, ux), best);
}

/* ------------------------------------------------------------------------------------------------
 * Stream headers (:624-690)
 * ------------------------------------------------------------------------------------------------ */
typedef struct { uint8_t *p; int bit; } BitW;
static void bw_put(BitW *w, int v, int n) {
    for (n--; n >= 0; n--) {
        if ((v >> n) & 1) *w->p |= (uint8_t)(1 << w->bit); else *w->p &= (uint8_t)~(1 << w->bit);
        if (w->bit > 0) w->bit--; else { w->bit = 7; w->p++; }
    }
}
static void bw_ue_like(BitW *w, int v) {       /* the reference's ue(v) variant (:641-647), replicated literally */
    int t, len = 1;
    v++;
    for (t = v + 1; t != 1; t >>= 1) len += 2;
    bw_put(w, v & ((1 << ((len + 1) >> 1)) - 1), (len >> 1) + ((len + 1) >> 1));
}
static int write_headers(uint8_t *out, int q, int hp, int wp) {
    static const uint8_t vps[27] = { 0, 0, 1, 0x40, 1, 0x0C, 1, 0xFF, 0xFF, 3, 0x10, 0, 0, 3, 0, 0, 3, 0, 0, 3, 0, 0, 3, 0, 0xB4, 0xF0, 0x24 };
    static const uint8_t sps[22] = { 0, 0, 1, 0x42, 1, 1, 3, 0x10, 0, 0, 3, 0, 0, 3, 0, 0, 3, 0, 0, 3, 0, 0xB4 };
    static const uint8_t pps[11] = { 0, 0, 1, 0x44, 1, 0xC0, 0x90, 0x91, 0x81, 0xD9, 0x20 };
    static const uint8_t slice_qp[5][2] = { {0x16, 0xDE}, {0x10, 0xDE}, {0x2B, 0x78}, {0x4D, 0xE0}, {0x97, 0x80} };
    static const uint8_t slice[6] = { 0, 0, 1, 0x26, 1, 0xAC };
    BitW w;
    uint8_t *p = out;
    memcpy(p, vps, 27); p += 27;
    memcpy(p, sps, 22); p += 22;
    w.p = p; w.bit = 7;
    bw_put(&w, 0xA, 4); bw_ue_like(&w, wp); bw_ue_like(&w, hp);
    bw_put(&w, 0x197EE4, 22); bw_put(&w, 0x681ED1, 24);
    if (w.bit < 7) { *w.p &= (uint8_t)(0xFE << w.bit); w.p++; }
    p = w.p;
    memcpy(p, pps, 11); p += 11;
    memcpy(p, slice, 6); p += 6;
    *p++ = slice_qp[q][0]; *p++ = slice_qp[q][1];
    return (int)(p - out);
}

/* ------------------------------------------------------------------------------------------------
 * Frame driver (:1569-1646).  Same C signature as the reference's HEVCImageEncoder (src/HEVCe/HEVCe.h:5-12).
 * ------------------------------------------------------------------------------------------------ */
static int *g_trace; static int g_trace_cap; static int g_trace_n;
void oracle_set_trace(int *buf, int cap) { g_trace = buf; g_trace_cap = cap; g_trace_n = 0; }
int oracle_trace_len(void) { return g_trace_n; }

int oracle_HEVCImageEncoder(unsigned char *pbuffer, const unsigned char *img, unsigned char *img_rcon,
                            int *ysz, int *xsz, const int qpd6) {
    const int h = *ysz, w = *xsz;
    const int hp = (imin(h, 8192) + 31) / 32 * 32, wp = (imin(w, 8192) + 31) / 32 * 32;
    Enc *e = (Enc *)calloc(1, sizeof(Enc));
    uint8_t *out = pbuffer;
    int cy, cx, i, j;
    init_tables();
    e->q = qpd6; e->hp = hp; e->wp = wp;
    e->ms = wp / 4 + 2;
    e->mapsz = (uint8_t *)malloc((size_t)e->ms * (hp / 4 + 2));
    e->mapmode = (uint8_t *)malloc((size_t)e->ms * (hp / 4 + 2));
    memset(e->mapsz, 32, (size_t)e->ms * (hp / 4 + 2));
    memset(e->mapmode, 1, (size_t)e->ms * (hp / 4 + 2));
    e->trace = g_trace; e->trace_cap = g_trace_cap; e->trace_n = 0;
    e->live.buf = e->stack; arith_reset(&e->live.a); ctx_reset(e->live.cx, qpd6);
    out += write_headers(out, qpd6, hp, wp);

    for (cy = 0; cy < hp; cy += CTU) for (cx = 0; cx < wp; cx += CTU) {
        const int has_l = cx > 0, has_a = cy > 0, has_ar = has_a && (cx + CTU < wp);
        e->cy = cy / 4; e->cx = cx / 4;
        for (i = 0; i < CTU; i++)            /* neighbours come from the padded reconstruction, clamped (:1613-1617) */
            *rec_at(e, i, -1) = img_rcon[(size_t)clip3(cy + i, 0, hp - 1) * wp + clip3(cx - 1, 0, wp - 1)];
        for (j = -1; j < 2 * CTU; j++)
            *rec_at(e, -1, j) = img_rcon[(size_t)clip3(cy - 1, 0, hp - 1) * wp + clip3(cx + j, 0, wp - 1)];
        for (i = 0; i < CTU; i++) for (j = 0; j < CTU; j++)   /* source pixels replicate the original edges (:1621) */
            e->org[i][j] = img[(size_t)clip3(cy + i, 0, h - 1) * w + clip3(cx + j, 0, w - 1)];
        search_cu(e, 0, 0, CTU, has_l, 0, has_a, has_ar);
        for (i = 0; i < CTU; i++) memcpy(img_rcon + (size_t)(cy + i) * wp + cx, rec_at(e, i, 0), CTU);
        put_terminate(&e->live, cy + CTU >= hp && cx + CTU >= wp);
        memcpy(out, e->stack, (size_t)e->live.a.cnt); out += e->live.a.cnt; e->live.a.cnt = 0;
    }
    arith_finish(&e->live);
    memcpy(out, e->stack, (size_t)e->live.a.cnt); out += e->live.a.cnt;
    g_trace_n = e->trace_n;
    free(e->mapsz); free(e->mapmode); free(e);
    *ysz = hp; *xsz = wp;
    return (int)(out - pbuffer);
}

/* small stage-level exports for differential tests against the reference's own (non-static) functions */
void oracle_predict(int sz, int mode, const unsigned char *left_above_corner /* [1+64+64]: corner, left[64], above[64] (unfiltered) */,
                    unsigned char *dst /* sz*sz */) {
    Border b; int i, n = 2 * sz;
    init_tables();
    b.ucorner = left_above_corner[0];
    memcpy(b.uleft, left_above_corner + 1, 64); memcpy(b.uabove, left_above_corner + 65, 64);
    b.fcorner   = (uint8_t)((2 + b.uleft[0] + b.uabove[0] + 2 * b.ucorner) >> 2);
    b.fleft[0]  = (uint8_t)((2 + 2 * b.uleft[0]  + b.uleft[1]  + b.ucorner) >> 2);
    b.fabove[0] = (uint8_t)((2 + 2 * b.uabove[0] + b.uabove[1] + b.ucorner) >> 2);
    for (i = 1; i < n - 1; i++) {
        b.fleft[i]  = (uint8_t)((2 + 2 * b.uleft[i]  + b.uleft[i - 1]  + b.uleft[i + 1])  >> 2);
        b.fabove[i] = (uint8_t)((2 + 2 * b.uabove[i] + b.uabove[i - 1] + b.uabove[i + 1]) >> 2);
    }
    b.fleft[n - 1] = b.uleft[n - 1]; b.fabove[n - 1] = b.uabove[n - 1];
    predict_block(dst, sz, sz, mode, &b);
}
void oracle_fwd_transform(int sz, const int *src, int *dst) { init_tables(); fwd_transform(sz, src, dst); }
void oracle_inv_transform(int sz, const int *src, int *dst) { init_tables(); inv_transform(sz, src, dst); }
void oracle_rdoq(int q, int sz, const int *coef, int *lvl) { init_tables(); rdoq(q, sz, coef, lvl); }
void oracle_dequant(int q, int sz, const int *lvl, int *out) { dequant(q, sz, lvl, out); }
int  oracle_transform_entry(int s, int i, int k) { init_tables(); return TM[s][i][k]; }
void oracle_scan(int type, int s, unsigned char *yx) { init_tables(); memcpy(yx, SCAN[type][s], (size_t)2 * (16 << (2 * s))); }
