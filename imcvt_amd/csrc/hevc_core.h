// hevc_core.h — the H.265 intra encode hot path of ImCvt (reference: src/HEVCe/HEVCe.c), written for
// CDNA4: one workgroup per frame, one 64-lane wavefront per candidate set of a CU (35 modes of one
// shape), prediction borders / transform tiles / CABAC contexts staged in LDS, the CABAC coder INSIDE the
// on-device decision loop (the RD rate is the live coder's byte position, reference :1363-1364,:1437).
//
// The file is compiled two ways:
//   * by hipcc for gfx950 (imcvt_amd/csrc/hevc_hip.hip) — the product;
//   * by g++ with -DIMCVT_HOSTEMU (tests/hostemu) — a TEST-ONLY harness in which a wavefront is a serial
//     loop over 64 lanes, used to debug bit-exactness on a machine without a GPU.  It is never shipped,
//     never loaded by imcvt_amd, and is not a fallback.
//
// Conventions: code outside LANES(){} is wave-uniform; every value that crosses a LANES block lives in LDS.
#pragma once
#include <stdint.h>

typedef uint8_t u8;  typedef uint16_t u16; typedef uint32_t u32; typedef uint64_t u64;
typedef int8_t i8;   typedef int16_t i16;  typedef int32_t i32;

#define NWAVES 3
#define WG_THREADS (NWAVES * 64)
#define NMODE 35
#define I32MAX 0x7fffffff

#ifdef IMCVT_HOSTEMU
  #define HD static inline
  #define HDN static
  #define LANES(l) for (int l = 0; l < 64; ++l)
  #define WAVES(w) for (int w = 0; w < NWAVES; ++w)
  HD void wave_sync() {}
  HD void wg_sync() {}
  HD i32 lds_add(i32 *p, i32 v) { i32 o = *p; *p += v; return o; }
  HD i32 lds_max(i32 *p, i32 v) { i32 o = *p; if (v > o) *p = v; return o; }
  HD u32 lds_or(u32 *p, u32 v) { u32 o = *p; *p |= v; return o; }
  HD int clz32(u32 v) { return v ? __builtin_clz(v) : 32; }
#else
  #define HD __device__ __forceinline__
  #define HDN __device__ __noinline__
  #define LANES(l) for (int l = (int)(threadIdx.x & 63u), l##_once = 1; l##_once; l##_once = 0)
  #define WAVES(w) for (int w = (int)(threadIdx.x >> 6), w##_once = 1; w##_once; w##_once = 0)
  // LDS traffic of one wavefront is in program order; the fence only stops the compiler (and drains
  // global stores, which the trial coders read back from other lanes of the same wave).
  HD void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); }
  HD void wg_sync() { __syncthreads(); }
  HD i32 lds_add(i32 *p, i32 v) { return atomicAdd(p, v); }
  HD i32 lds_max(i32 *p, i32 v) { return atomicMax(p, v); }
  HD u32 lds_or(u32 *p, u32 v) { return atomicOr(p, v); }
  HD int clz32(u32 v) { return __clz((int)v); }
#endif

HD int iabs(int v) { return v < 0 ? -v : v; }
HD int imin(int a, int b) { return a < b ? a : b; }
HD int imax(int a, int b) { return a > b ? a : b; }
HD int clip3(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
HD int clip16(int v) { return clip3(v, -32768, 32767); }

// ---------------------------------------------------------------------------------------------------
// Context layout (luma path only; the reference's 142-byte ContextSet :744-758 minus never-used entries)
// ---------------------------------------------------------------------------------------------------
enum { CX_SPLIT_CU = 0, CX_PART = 3, CX_PREV_INTRA = 4, CX_CHROMA_PRED = 5, CX_SPLIT_TU = 6, CX_CBF_LUMA = 9, CX_CBF_CHROMA = 11,
       CX_LAST_X = 12, CX_LAST_Y = 27, CX_CSBF = 42, CX_SIG = 44, CX_GT1 = 71, CX_GT2 = 87, NCTX = 91, CTX_STRIDE = 92 };

// ---------------------------------------------------------------------------------------------------
// Constant tables, built once on the host (hevc_tables.h) and staged into LDS by every workgroup.
// ---------------------------------------------------------------------------------------------------
struct Tables {
    i8  C[1360];           // forward matrices, row-major [i][k]; offsets {0,16,80,336} for N=4,8,16,32   (:391-464)
    i8  CT[1360];          // their transposes [k][i]
    u8  cgpos[3][4][64];   // [scan type][log2N-2][g] -> (gy<<3)|gx, coefficient-group scan order        (:1126-1150)
    u8  cgrank[3][4][64];  // inverse: [type][s][gy*8+gx] -> g
    u8  incg[3][16];       // [type][n] -> (yi<<2)|xi inside a 4x4 group
    u8  incg_rank[3][16];  // inverse: [type][yi*4+xi] -> n
    u32 lps4[64];          // rangeTabLps rows packed little-endian                                       (:703-712)
    u8  nextlps[128];      // packed-state transition on LPS                                              (:701)
    u32 posadd[4][3];      // sig_coeff ctx increment per in-group scan position, 2 bits each [pattern][type]  (:1115-1120)
    u64 c4tab[3];          // 4x4-TU sig_coeff ctx per scan position, 4 bits each [type]                  (:1092)
    u8  ctx_init[5][CTX_STRIDE];   // initial context states per qpd6                                    (:726-784)
    u8  ang[36];           // intraPredAngle + 32                                                         (:282)
    u16 iang[36];          // |invAngle|                                                                  (:283)
};
HD int mat_off(int s) { return s == 0 ? 0 : s == 1 ? 16 : s == 2 ? 80 : 336; }

// ---------------------------------------------------------------------------------------------------
// Arithmetic coder state (:796-805) — `cnt` counts bytes of the current CTU already pushed.
// ---------------------------------------------------------------------------------------------------
struct Arith { i32 range, low, nbits, nbytes, bufbyte, zeros, cnt; };
HD void arith_reset(Arith &a) { a.range = 510; a.low = 0; a.nbits = 23; a.nbytes = 0; a.bufbyte = 0xFF; a.zeros = 0; a.cnt = 0; }
HD int arith_len(const Arith &a) { return 8 * (a.cnt + a.nbytes) + 23 - a.nbits; }      // :834

// sink[a.cnt] is where the next byte goes
HD void emit_byte(Arith &a, u8 *sink, int v) {                                             // :820-831
    v &= 0xFF;
    if (a.zeros >= 2 && v <= 3) { sink[a.cnt++] = 3; a.zeros = 0; }
    sink[a.cnt++] = (u8)v;
    a.zeros = v ? 0 : a.zeros + 1;
}
HD void carry_out(Arith &a, u8 *sink) {                                                    // :858-878
    if (a.nbits < 12) {
        int lead = (int)((u32)a.low >> (24 - a.nbits));
        a.nbits += 8;
        a.low &= (i32)(0xFFFFFFFFu >> a.nbits);
        if (lead == 0xFF) a.nbytes++;
        else if (a.nbytes > 0) {
            int carry = lead >> 8, v = a.bufbyte + carry;
            a.bufbyte = lead & 0xFF;
            emit_byte(a, sink, v);
            v = (0xFF + carry) & 0xFF;
            for (; a.nbytes > 1; a.nbytes--) emit_byte(a, sink, v);
        } else { a.nbytes = 1; a.bufbyte = lead; }
    }
}
HD void code_bin(Arith &a, u8 *cx, const Tables &T, u8 *sink, int ci, int bin) {          // :913-932
    const int p = cx[ci];
    const int lps = (int)((T.lps4[p >> 1] >> (((a.range >> 6) & 3) * 8)) & 0xFF);
    a.range -= lps;
    if (bin != (p & 1)) {
        const int sh = imin(6, clz32((u32)lps) - 23);     // renorm table :714 == 8 - floor(log2 lps), capped at 6
        cx[ci] = T.nextlps[p];
        a.low = (a.low + a.range) << sh;
        a.range = lps << sh;
        a.nbits -= sh;
    } else {
        cx[ci] = (u8)((p < 124) ? p + 2 : p);
        if (a.range < 256) { a.low <<= 1; a.range <<= 1; a.nbits--; }
    }
    carry_out(a, sink);
}
HD void code_bypass_chunk(Arith &a, u8 *sink, int v, int n) {                             // one <=8-bin step of :898-910
    a.low = (a.low << n) + a.range * v;
    a.nbits -= n;
    carry_out(a, sink);
}
HD void code_terminate(Arith &a, u8 *sink, int bin) {                                      // :881-895
    a.range -= 2;
    if (bin) { a.low = (a.low + a.range) << 7; a.range = 256; a.nbits -= 7; }
    else if (a.range < 256) { a.low <<= 1; a.range <<= 1; a.nbits--; }
    carry_out(a, sink);
}
HD void arith_finish(Arith &a, u8 *sink) {                                                 // :839-855
    int fill = 0, t;
    if ((a.low >> (32 - a.nbits)) > 0) { emit_byte(a, sink, a.bufbyte + 1); a.low -= 1 << (32 - a.nbits); }
    else { if (a.nbytes > 0) emit_byte(a, sink, a.bufbyte); fill = 0xFF; }
    for (; a.nbytes > 1; a.nbytes--) emit_byte(a, sink, fill);
    t = (a.low >> 8) << a.nbits;
    emit_byte(a, sink, t >> 16); emit_byte(a, sink, t >> 8); emit_byte(a, sink, t);
}

// ---------------------------------------------------------------------------------------------------
// RD cost (:177-185), coefficient rate model (:526-535)
// ---------------------------------------------------------------------------------------------------
HD int w_dist(int q) { return q < 3 ? 11 : q == 3 ? 5 : 1; }
HD int w_bits(int q) { return q == 0 ? 1 : q == 1 ? 4 : q == 2 ? 16 : q == 3 ? 29 : 23; }
struct RdW { int wd, wb, td, tb; };     // weights and their saturation thresholds I32MAX/w
HD RdW rd_weights(int q) { RdW r; r.wd = w_dist(q); r.wb = w_bits(q); r.td = I32MAX / r.wd; r.tb = I32MAX / r.wb; return r; }
HD int rd_cost(const RdW &w, int dist, int bits) {
    int c1 = (w.td <= dist) ? I32MAX : w.wd * dist;
    int c2 = (w.tb <= bits) ? I32MAX : w.wb * bits;
    return (I32MAX - c1 <= c2) ? I32MAX : c1 + c2;
}
HD int level_rate(int level) {
    if (level < 6) return level == 0 ? 0 : level == 1 ? 70000 : level == 2 ? 90000 : level == 3 ? 92000 : level == 4 ? 157536 : 190304;
    int i = 31 - clz32((u32)(level - 5));           // number of exp-Golomb escape doublings
    return 92000 + ((4 + 2 * i) << 15);
}

// ---------------------------------------------------------------------------------------------------
// Workgroup memory
// ---------------------------------------------------------------------------------------------------
#define RS 68            // reconstruction tile stride: 1 border column + 64, padded
#define FIFO_CAP 64
#define FIFO_STRIDE 66   // u16 units per lane (odd dword stride: conflict-free when lanes read the same slot)

struct Border {          // prediction references of one block (:196-257): unfiltered / [1 2 1]-filtered
    u8 uc, fc; i16 dc;
    u8 ul[68], ua[68], fl[68], fa[68];
};
struct BorderS {         // same for blocks <= 16 (the four-TU shape keeps one per mode)
    u8 uc, fc; i16 dc;
    u8 ul[36], ua[36], fl[36], fa[36];
};

struct WaveMem {
    union {
        struct { u8 pred[1024]; i16 res[1024]; i32 tmp[1024]; } p1;                 // one pipeline pass
        struct { u8 cx[NMODE][CTX_STRIDE]; u16 fifo[NMODE][FIFO_STRIDE]; i16 mag[NMODE][16]; } p2;   // trial coders
    } u;
    Border  bsh;                 // border shared by all modes of a block
    BorderS bc[NMODE];           // per-mode borders (four-TU shape, TUs 1..3)
    u8  t3row[NMODE][4][16];     // per mode: bottom row / right column of each reconstructed TU
    u8  t3col[NMODE][4][16];
    u8  rec4[NMODE][16];         // 4x4 PU candidates' reconstructions
    i32 last[4][NMODE];          // per TU: last significant scan position (-1: none)
    u32 cgm[4][NMODE][2];        // per TU: significant-group bitmap, bit gy*8+gx
    i32 sse[NMODE];
    i32 cost[NMODE];
    Arith fin[NMODE];            // coder state each trial ended in
    // NxN bookkeeping (PU wave)
    i32 pu_mode[4], pu_sse[4], pu_last[4];
    i16 pu_lv[4][16];
    i32 nxn_cost;
};

struct Shm {
    Tables T;
    u8  org[32][32];
    u8  rec[33][RS];             // rec[y+1][x+1]; row 0 / column 0 are the neighbours
    u8  cx[CTX_STRIDE];          // live contexts
    Arith live;
    Arith entry_a[3];            // coder + contexts on entry to the CU of depth 0/1/2
    u8  entry_cx[3][CTX_STRIDE];
    u8  mapsz[10][12], mapmode[10][12];   // 4x4-unit neighbour maps of this CTU with a 1-cell apron (:1591-1599)
    i32 split_cost[3];
    i32 win_kind, win_mode;      // decision broadcast
    i32 red[NWAVES];             // small reductions
    WaveMem W[NWAVES];
};

// Per-frame job and per-workgroup scratch (global memory)
struct FrameJob {
    const u8 *img;   // h*w gray8
    u8 *out;         // stream buffer
    u8 *rcon;        // hp*wp reconstruction
    i32 h, w, hp, wp, q;
    i32 hdr_len;     // header bytes already placed at out[0..hdr_len)
    i32 *out_len;    // result
};
struct Scratch {
    i16 *lv;         // [NWAVES][NMODE*1024] quantised levels in scan order
    u8  *bytes;      // [NWAVES][NMODE][TRIAL_BYTES] bytes emitted by trial coders
    u8  *above_sz;   // [wp/4] CU sizes of the CTU row above (:1633-1636)
    i32 *trace;      // optional decision trace (8 ints per CU), or null
    i32 trace_cap;
};
#define TRIAL_BYTES 3584
#define LV_PER_WAVE (NMODE * 1024)

// ---------------------------------------------------------------------------------------------------
// Prediction (:262-381), evaluated per pixel
// ---------------------------------------------------------------------------------------------------
HD int uses_filtered(int N, int mode) {            // :274-280 as a distance-to-H/V threshold
    if (N == 4 || mode == 1) return 0;
    if (mode == 0) return 1;
    int d = imin(iabs(mode - 10), iabs(mode - 26));
    return d > (N == 8 ? 7 : N == 16 ? 1 : 0);
}

template <class B>
HD int pred_px(const Tables &T, const B &b, int N, int lg, int mode, int y, int x) {
    const int f = uses_filtered(N, mode);
    const u8 *L = f ? b.fl : b.ul, *A = f ? b.fa : b.ua;
    const int corner = f ? b.fc : b.uc;
    if (mode == 0)
        return ((N - 1 - x) * L[y] + (x + 1) * A[N] + (N - 1 - y) * A[x] + (y + 1) * L[N] + N) >> (lg + 1);
    if (mode == 1) {
        const int dc = b.dc;
        if (N <= 16) {
            if (y == 0 && x == 0) return (2 + 2 * dc + L[0] + A[0]) >> 2;
            if (y == 0) return (2 + 3 * dc + A[x]) >> 2;
            if (x == 0) return (2 + 3 * dc + L[y]) >> 2;
        }
        return dc;
    }
    if (mode == 10) return (N <= 16 && y == 0) ? clip3(((A[x] - corner) >> 1) + L[0], 0, 255) : L[y];
    if (mode == 26) return (N <= 16 && x == 0) ? clip3(((L[y] - corner) >> 1) + A[0], 0, 255) : A[x];
    {
        const int horiz = mode < 18;
        const int ang = (int)T.ang[mode] - 32, iang = T.iang[mode];
        const u8 *M = horiz ? L : A, *Sd = horiz ? A : L;
        const int i = horiz ? x : y, j = horiz ? y : x;
        const int off = ang * (i + 1), oi = off >> 5, of = off & 31;
        const int t1 = oi + j + 1, t2 = t1 + 1;
        // reference line: t==0 corner, t>0 main[t-1], t<0 projected side sample (:353-364)
        int p1 = t1 == 0 ? corner : t1 > 0 ? M[t1 - 1] : Sd[((128 - iang * t1) >> 8) - 1];
        int p2 = t2 == 0 ? corner : t2 > 0 ? M[t2 - 1] : Sd[((128 - iang * t2) >> 8) - 1];   // M[2N] is read only with of==0
        return ((32 - of) * p1 + of * p2 + 16) >> 5;
    }
}

// ---------------------------------------------------------------------------------------------------
// Border assembly
// ---------------------------------------------------------------------------------------------------
// Shared border of the block at (y0,x0) from the reconstruction tile.  Wave-uniform call.
HD void border_from_tile(Shm &S, WaveMem &W, int N, int y0, int x0, int hl, int hbl, int ha, int har) {
    Border &b = W.bsh;
    const int n2 = 2 * N;
    LANES(l) {
        const u8 *t = &S.rec[y0 + 1][x0 + 1];
        int uc = (hl && ha) ? t[-RS - 1] : hl ? t[-1] : ha ? t[-RS] : 128;
        if (l < n2) {
            int lv_, av_;
            if (l < N) { lv_ = hl ? t[l * RS - 1] : uc; av_ = ha ? t[-RS + l] : uc; }
            else {
                lv_ = hbl ? t[l * RS - 1] : (hl ? t[(N - 1) * RS - 1] : uc);
                av_ = har ? t[-RS + l] : (ha ? t[-RS + N - 1] : uc);
            }
            b.ul[l] = (u8)lv_; b.ua[l] = (u8)av_;
        }
        if (l == 0) { b.uc = (u8)uc; b.ul[n2] = 0; b.ua[n2] = 0; b.fl[n2] = 0; b.fa[n2] = 0; }
    }
    wave_sync();
    LANES(l) {
        if (l < n2) {
            int fl_, fa_;
            if (l == 0) { fl_ = (2 + 2 * b.ul[0] + b.ul[1] + b.uc) >> 2; fa_ = (2 + 2 * b.ua[0] + b.ua[1] + b.uc) >> 2; }
            else if (l == n2 - 1) { fl_ = b.ul[l]; fa_ = b.ua[l]; }
            else { fl_ = (2 + 2 * b.ul[l] + b.ul[l - 1] + b.ul[l + 1]) >> 2; fa_ = (2 + 2 * b.ua[l] + b.ua[l - 1] + b.ua[l + 1]) >> 2; }
            b.fl[l] = (u8)fl_; b.fa[l] = (u8)fa_;
        }
        if (l == 0) {
            int dc = N;
            for (int i = 0; i < N; i++) dc += b.ul[i] + b.ua[i];
            b.dc = (i16)(dc / (2 * N));
            b.fc = (u8)((2 + b.ul[0] + b.ua[0] + 2 * b.uc) >> 2);
        }
    }
    wave_sync();
}

// Per-mode borders of TU k (1..3) of the four-TU shape of the CU at (y0,x0,N): samples inside the CU come
// from that mode's own reconstruction of TUs < k (:1459,1466), the rest from the tile.
HD void border_tu_split(Shm &S, WaveMem &W, int N, int y0, int x0, int k, int hl, int hbl, int ha, int har) {
    const int h = N / 2, n2 = N;   // 2*h entries per side
    LANES(l) {
        for (int e = l; e < NMODE * n2; e += 64) {
            const int c = e / n2, i = e - c * n2;
            BorderS &b = W.bc[c];
            int uc, lv_, av_;
            if (k == 1) {
                const u8 *col0 = W.t3col[c][0];
                const u8 *ab = &S.rec[y0][x0 + h + 1];                     // row y0-1, starting at column x0+h
                uc = ha ? ab[-1] : col0[0];
                lv_ = (i < h) ? col0[i] : col0[h - 1];
                av_ = (i < h) ? (ha ? ab[i] : uc) : (har ? ab[i] : (ha ? ab[h - 1] : uc));
            } else if (k == 2) {
                const u8 *lf = &S.rec[y0 + h + 1][x0];                     // column x0-1, starting at row y0+h
                const u8 *row0 = W.t3row[c][0], *row1 = W.t3row[c][1];
                uc = hl ? lf[-RS] : row0[0];
                lv_ = (i < h) ? (hl ? lf[i * RS] : uc) : (hbl ? lf[i * RS] : (hl ? lf[(h - 1) * RS] : uc));
                av_ = (i < h) ? row0[i] : row1[i - h];
            } else {
                const u8 *col2 = W.t3col[c][2], *row1 = W.t3row[c][1];
                uc = W.t3row[c][0][h - 1];
                lv_ = (i < h) ? col2[i] : col2[h - 1];
                av_ = (i < h) ? row1[i] : row1[h - 1];
            }
            b.ul[i] = (u8)lv_; b.ua[i] = (u8)av_;
            if (i == 0) { b.uc = (u8)uc; b.ul[n2] = 0; b.ua[n2] = 0; b.fl[n2] = 0; b.fa[n2] = 0; }
        }
    }
    wave_sync();
    LANES(l) {
        for (int e = l; e < NMODE * n2; e += 64) {
            const int c = e / n2, i = e - c * n2;
            BorderS &b = W.bc[c];
            int fl_, fa_;
            if (i == 0) { fl_ = (2 + 2 * b.ul[0] + b.ul[1] + b.uc) >> 2; fa_ = (2 + 2 * b.ua[0] + b.ua[1] + b.uc) >> 2; }
            else if (i == n2 - 1) { fl_ = b.ul[i]; fa_ = b.ua[i]; }
            else { fl_ = (2 + 2 * b.ul[i] + b.ul[i - 1] + b.ul[i + 1]) >> 2; fa_ = (2 + 2 * b.ua[i] + b.ua[i - 1] + b.ua[i + 1]) >> 2; }
            b.fl[i] = (u8)fl_; b.fa[i] = (u8)fa_;
        }
        if (l < NMODE) {
            BorderS &b = W.bc[l];
            int dc = h;
            for (int i = 0; i < h; i++) dc += b.ul[i] + b.ua[i];
            b.dc = (i16)(dc / (2 * h));
            b.fc = (u8)((2 + b.ul[0] + b.ua[0] + 2 * b.uc) >> 2);
        }
    }
    wave_sync();
}

// ---------------------------------------------------------------------------------------------------
// The candidate pipeline: predict -> residual -> T -> RDOQ -> deQ -> T^-1 -> recon -> SSE  (:1425-1436)
// One lane owns one 4x4 output block (= one coefficient group) of one candidate; a pass covers
// 64/(N/4)^2 candidates.  All intermediates of a pass live in the wave's LDS slice.
// ---------------------------------------------------------------------------------------------------
enum { OUT_NONE = 0, OUT_REC4 = 1, OUT_T3SIDE = 2, OUT_TILE = 3 };

struct P1Args {
    int N, y0, x0;       // block inside the CTU
    int k;               // TU slot for last/cgm/levels
    int per_mode_border; // 0: W.bsh, 1: W.bc[c]
    int out_kind;        // what to keep of the reconstruction
    int only_mode;       // -1: all 35 modes, else just this one (winner reconstruction)
    i16 *lv;             // global levels base for this TU: [c][N*N], scan order (may be null for only_mode)
    int q;
};

HD void p1_run(Shm &S, WaveMem &W, const P1Args &P) {
    const Tables &T = S.T;
    const int N = P.N, lg = (N == 4) ? 2 : (N == 8) ? 3 : (N == 16) ? 4 : 5, s = lg - 2;
    const int nb = N >> 2, lpc = nb * nb, G = 64 / lpc, NN = N * N;
    const i8 *C = T.C + mat_off(s), *CT = T.CT + mat_off(s);
    const int ncand = (P.only_mode >= 0) ? 1 : NMODE;
    const int q = P.q;
    // quantiser constants (:546-554)
    const int a1 = s + 1, b1 = a1 + 7, ra = 1 << a1 >> 1, rb = 1 << b1 >> 1;
    const int dsh = 8 - s, sh = 19 - s + q, add = 1 << sh >> 1, dmax = I32MAX - add, thr = 9 << sh >> 2;
    const int dq = 1 << (5 - s + q);
    const RdW rw = rd_weights(q);

    for (int c0 = 0; c0 < ncand; c0 += G) {
        // ---- step 1: prediction and residual
        LANES(l) {
            const int sl = l / lpc, blk = l - sl * lpc, by = blk / nb, bx = blk - by * nb, c = c0 + sl;
            if (c < ncand) {
                const int mode = (P.only_mode >= 0) ? P.only_mode : c;
                u8 *pp = W.u.p1.pred + sl * NN; i16 *rp = W.u.p1.res + sl * NN;
                for (int yi = 0; yi < 4; yi++) for (int xi = 0; xi < 4; xi++) {
                    const int y = by * 4 + yi, x = bx * 4 + xi;
                    const int p = P.per_mode_border ? pred_px(T, W.bc[c], N, lg, mode, y, x) : pred_px(T, W.bsh, N, lg, mode, y, x);
                    pp[y * N + x] = (u8)p;
                    rp[y * N + x] = (i16)((int)S.org[P.y0 + y][P.x0 + x] - p);
                }
            }
        }
        wave_sync();
        // ---- step 2: tmp = (C * res + ra) >> a                                              (:514 forward)
        LANES(l) {
            const int sl = l / lpc, blk = l - sl * lpc, by = blk / nb, bx = blk - by * nb, c = c0 + sl;
            if (c < ncand) {
                const i16 *rp = W.u.p1.res + sl * NN; i32 *tp = W.u.p1.tmp + sl * NN;
                int acc[4][4];
                for (int r = 0; r < 4; r++) for (int cc = 0; cc < 4; cc++) acc[r][cc] = ra;
                for (int k0 = 0; k0 < N; k0 += 4) {
                    int xv[4][4];
                    for (int kk = 0; kk < 4; kk++) for (int cc = 0; cc < 4; cc++) xv[kk][cc] = rp[(k0 + kk) * N + bx * 4 + cc];
                    for (int r = 0; r < 4; r++) {
                        const i8 *cr = C + (by * 4 + r) * N + k0;
                        for (int kk = 0; kk < 4; kk++) { const int cv = cr[kk]; for (int cc = 0; cc < 4; cc++) acc[r][cc] += cv * xv[kk][cc]; }
                    }
                }
                for (int r = 0; r < 4; r++) for (int cc = 0; cc < 4; cc++) tp[(by * 4 + r) * N + bx * 4 + cc] = acc[r][cc] >> a1;
            }
        }
        wave_sync();
        // ---- step 3: coef = (tmp * C^T + rb) >> b ; RDOQ ; levels out ; dequantise           (:515, :540-614)
        LANES(l) {
            const int sl = l / lpc, blk = l - sl * lpc, by = blk / nb, bx = blk - by * nb, c = c0 + sl;
            if (c < ncand) {
                const int mode = (P.only_mode >= 0) ? P.only_mode : c;
                const i32 *tp = W.u.p1.tmp + sl * NN; i16 *dp = W.u.p1.res + sl * NN;
                int acc[4][4];
                for (int r = 0; r < 4; r++) for (int cc = 0; cc < 4; cc++) acc[r][cc] = rb;
                for (int k0 = 0; k0 < N; k0 += 4) {
                    int tv[4][4];
                    for (int r = 0; r < 4; r++) for (int kk = 0; kk < 4; kk++) tv[r][kk] = tp[(by * 4 + r) * N + k0 + kk];
                    for (int cc = 0; cc < 4; cc++) {
                        const i8 *cr = C + (bx * 4 + cc) * N + k0;
                        for (int kk = 0; kk < 4; kk++) { const int cv = cr[kk]; for (int r = 0; r < 4; r++) acc[r][cc] += tv[r][kk] * cv; }
                    }
                }
                int lvl[4][4], sum = 0;
                for (int r = 0; r < 4; r++) for (int cc = 0; cc < 4; cc++) {
                    const int cf = acc[r][cc] >> b1, av = iabs(cf);
                    const int d = (av > 0x1ffff) ? dmax : imin((av & 0x1ffff) << 14, dmax);
                    int lq = clip16((int)(((u32)d + (u32)add) >> sh));
                    const int lo = imax(0, lq - 2);
                    int best = I32MAX, pick = 0;
                    for (; lq >= lo; lq--) {
                        const int e = iabs(d - (lq << sh)) >> dsh;
                        const int dist = ((e < 46340) ? e * e : I32MAX) >> 7;
                        const int cost = rd_cost(rw, dist, level_rate(lq));
                        if (cost < best) { best = cost; pick = lq; }
                    }
                    lvl[r][cc] = (cf < 0) ? -pick : pick;
                    sum += imin(d, thr);
                }
                const int zero_out = sum < thr;
                // scan bookkeeping: this lane's block is coefficient group (by,bx)
                const int st = (N <= 8) ? ((iabs(mode - 26) <= 4) ? 1 : (iabs(mode - 10) <= 4) ? 2 : 0) : 0;   // :1133-1141
                const int g = T.cgrank[st][s][by * 8 + bx];
                int hi = -1, any = 0;
                for (int r = 0; r < 4; r++) for (int cc = 0; cc < 4; cc++) { if (zero_out) lvl[r][cc] = 0; any |= lvl[r][cc]; }
                i16 *lvg = (P.lv && any) ? P.lv + (size_t)c * NN + g * 16 : (i16 *)0;   // only coded groups are ever read back
                for (int r = 0; r < 4; r++) for (int cc = 0; cc < 4; cc++) {
                    const int v = lvl[r][cc];
                    const int n = T.incg_rank[st][r * 4 + cc];
                    if (v) hi = imax(hi, n);
                    if (lvg) lvg[n] = (i16)v;
                    dp[(by * 4 + r) * N + bx * 4 + cc] = (i16)clip16(v * dq);
                }
                if (hi >= 0 && P.only_mode < 0) {
                    lds_max(&W.last[P.k][c], g * 16 + hi);
                    const int bit = by * 8 + bx;
                    lds_or(&W.cgm[P.k][c][bit >> 5], 1u << (bit & 31));
                }
            }
        }
        wave_sync();
        // ---- step 4: itmp = clip16((C^T * deq + 64) >> 7)                                    (:514 inverse)
        LANES(l) {
            const int sl = l / lpc, blk = l - sl * lpc, by = blk / nb, bx = blk - by * nb, c = c0 + sl;
            if (c < ncand) {
                const i16 *dp = W.u.p1.res + sl * NN; i16 *ip = (i16 *)W.u.p1.tmp + sl * NN;
                int acc[4][4];
                for (int r = 0; r < 4; r++) for (int cc = 0; cc < 4; cc++) acc[r][cc] = 64;
                for (int k0 = 0; k0 < N; k0 += 4) {
                    int xv[4][4];
                    for (int kk = 0; kk < 4; kk++) for (int cc = 0; cc < 4; cc++) xv[kk][cc] = dp[(k0 + kk) * N + bx * 4 + cc];
                    for (int r = 0; r < 4; r++) {
                        const i8 *cr = CT + (by * 4 + r) * N + k0;
                        for (int kk = 0; kk < 4; kk++) { const int cv = cr[kk]; for (int cc = 0; cc < 4; cc++) acc[r][cc] += cv * xv[kk][cc]; }
                    }
                }
                // all lanes of this candidate must have finished reading tmp before it is overwritten as i16:
                // they have — step 3 ended with a wave_sync and this step only reads `res`.
                for (int r = 0; r < 4; r++) for (int cc = 0; cc < 4; cc++) ip[(by * 4 + r) * N + bx * 4 + cc] = (i16)clip16(acc[r][cc] >> 7);
            }
        }
        wave_sync();
        // ---- step 5: rec = clip8(clip16((itmp * C + 2048) >> 12) + pred) ; SSE                (:515 inverse, :146,:165)
        LANES(l) {
            const int sl = l / lpc, blk = l - sl * lpc, by = blk / nb, bx = blk - by * nb, c = c0 + sl;
            if (c < ncand) {
                const i16 *ip = (const i16 *)W.u.p1.tmp + sl * NN; const u8 *pp = W.u.p1.pred + sl * NN;
                int acc[4][4];
                for (int r = 0; r < 4; r++) for (int cc = 0; cc < 4; cc++) acc[r][cc] = 2048;
                for (int k0 = 0; k0 < N; k0 += 4) {
                    int tv[4][4];
                    for (int r = 0; r < 4; r++) for (int kk = 0; kk < 4; kk++) tv[r][kk] = ip[(by * 4 + r) * N + k0 + kk];
                    for (int cc = 0; cc < 4; cc++) {
                        const i8 *cr = CT + (bx * 4 + cc) * N + k0;
                        for (int kk = 0; kk < 4; kk++) { const int cv = cr[kk]; for (int r = 0; r < 4; r++) acc[r][cc] += tv[r][kk] * cv; }
                    }
                }
                int part = 0;
                for (int r = 0; r < 4; r++) for (int cc = 0; cc < 4; cc++) {
                    const int y = by * 4 + r, x = bx * 4 + cc;
                    const int rc = clip3(clip16(acc[r][cc] >> 12) + pp[y * N + x], 0, 255);
                    const int d = (int)S.org[P.y0 + y][P.x0 + x] - rc;
                    part += d * d;
                    if (P.out_kind == OUT_REC4) W.rec4[c][y * 4 + x] = (u8)rc;
                    else if (P.out_kind == OUT_TILE) S.rec[P.y0 + y + 1][P.x0 + x + 1] = (u8)rc;
                    else if (P.out_kind == OUT_T3SIDE) {
                        if (y == N - 1) W.t3row[c][P.k][x] = (u8)rc;
                        if (x == N - 1) W.t3col[c][P.k][y] = (u8)rc;
                    }
                }
                if (P.only_mode < 0) lds_add(&W.sse[c], part);
            }
        }
        wave_sync();
    }
}

// ---------------------------------------------------------------------------------------------------
// Trial coding.  Each lane binarises its candidate one unit at a time into a private token FIFO and
// drains it through the arithmetic coder; the SIMT loop keeps the drain convergent across lanes.
//   token: bit15=0 -> context bin  (ci<<1 | bin);  bit15=1 -> bypass chunk (n<<8 | value), n<=8
// ---------------------------------------------------------------------------------------------------
struct Fifo { u16 *buf; int n; };
HD void tk_bin(Fifo &f, int ci, int bin) { f.buf[f.n++] = (u16)((ci << 1) | (bin & 1)); }
HD void tk_bypass(Fifo &f, int v, int len) {                  // chunking of :898-910
    v &= (1 << len) - 1;
    while (len > 0) { const int n = imin(len, 8); len -= n; f.buf[f.n++] = (u16)(0x8000 | (n << 8) | ((v >> len) & ((1 << n) - 1))); }
}
HD void fifo_drain(Fifo &f, Arith &a, u8 *cx, const Tables &T, u8 *sink) {
    for (int i = 0; i < f.n; i++) {
        const int t = f.buf[i];
        if (t & 0x8000) code_bypass_chunk(a, sink, t & 0xFF, (t >> 8) & 0xF);
        else code_bin(a, cx, T, sink, t >> 1, t & 1);
    }
    f.n = 0;
}

HD void mpm_list(int l, int a, int *m) {                       // :957-976
    if (l != a) { m[0] = l; m[1] = a; m[2] = (l != 0 && a != 0) ? 0 : (l + a < 2) ? 26 : 1; }
    else if (l > 1) { m[0] = l; m[1] = ((l + 29) & 31) + 2; m[2] = ((l - 1) & 31) + 2; }
    else { m[0] = 0; m[1] = 1; m[2] = 26; }
}
HD int mpm_hit(const int *m, int mode) { int h = -1; for (int j = 0; j < 3; j++) if (m[j] == mode) h = j; return h; }
HD void tk_mode_rest(Fifo &f, int *m, int hit, int mode) {    // second half of :984-1017
    if (hit >= 0) { tk_bypass(f, hit > 0, 1); if (hit > 0) tk_bypass(f, hit - 1, 1); }
    else {
        int t, r = mode;
        if (m[0] < m[1]) { t = m[0]; m[0] = m[1]; m[1] = t; }
        if (m[1] < m[2]) { t = m[1]; m[1] = m[2]; m[2] = t; }
        if (m[0] < m[1]) { t = m[0]; m[0] = m[1]; m[1] = t; }
        for (int j = 0; j < 3; j++) if (r > m[j]) r--;
        tk_bypass(f, r, 5);
    }
}

HD int scan_type_of(int N, int mode) { return (N <= 8) ? ((iabs(mode - 26) <= 4) ? 1 : (iabs(mode - 10) <= 4) ? 2 : 0) : 0; }

HD void tk_last_pos(Fifo &f, int N, int s, int st, int y, int x) {          // :1045-1086
    const int base = (s == 0) ? 0 : (s == 1) ? 3 : (s == 2) ? 6 : 10, shf = (s == 0) ? 0 : 1;
    int ty = (st == 2) ? x : y, tx = (st == 2) ? y : x;
    // group index of a coordinate: 0,1,2,3,4,4,5,5,6,6,6,6,7,7,7,7,8*8,9*8
    const int gx = tx < 4 ? tx : (2 * (31 - clz32((u32)tx)) + ((tx >> (30 - clz32((u32)tx))) & 1));
    const int gy = ty < 4 ? ty : (2 * (31 - clz32((u32)ty)) + ((ty >> (30 - clz32((u32)ty))) & 1));
    const int gmax = 2 * (s + 2) - 1;                                         // group of N-1
    for (int i = 0; i < gx; i++) tk_bin(f, CX_LAST_X + base + (i >> shf), 1);
    if (gx < gmax) tk_bin(f, CX_LAST_X + base + (gx >> shf), 0);
    for (int i = 0; i < gy; i++) tk_bin(f, CX_LAST_Y + base + (i >> shf), 1);
    if (gy < gmax) tk_bin(f, CX_LAST_Y + base + (gy >> shf), 0);
    if (gx > 3) { const int nb_ = (gx - 2) >> 1, mn = (2 + (gx & 1)) << (nb_); for (int i = nb_ - 1; i >= 0; i--) tk_bypass(f, ((tx - mn) >> i) & 1, 1); }
    if (gy > 3) { const int nb_ = (gy - 2) >> 1, mn = (2 + (gy & 1)) << (nb_); for (int i = nb_ - 1; i >= 0; i--) tk_bypass(f, ((ty - mn) >> i) & 1, 1); }
}

HD void tk_remaining(Fifo &f, int v, int k) {                                // :1153-1168
    if (v < (3 << k)) { const int p = v >> k; tk_bypass(f, (1 << (p + 1)) - 2, p + 1); tk_bypass(f, v & ((1 << k) - 1), k); }
    else {
        int n = k; v -= 3 << k;
        for (; v >= (1 << n); n++) v -= 1 << n;
        const int t = 4 + n - k;
        tk_bypass(f, (1 << t) - 2, t); tk_bypass(f, v, n);
    }
}

// Description of what one lane has to code
struct TrialJob {
    int N;              // CU size
    int shape;          // 0: 2Nx2N one TU, 1: 2Nx2N four TUs, 2: NxN, 3: residual of one 4x4 TU only (PU pricing, :1515)
    int ctx_split;      // context of split_cu_flag=0, or -1 when the flag is absent
    int mode[4], ml[4], ma[4];
    const i16 *lv[4];   // scan-ordered levels per TU
    int last[4];        // last significant scan position per TU (-1: all zero)
    u32 cg0[4], cg1[4]; // significant-group bitmaps
};

// Code the whole job on (a, cx).  mag: 16 x i16 lane-private scratch.
HD void trial_run(const Tables &T, const TrialJob &J, Arith &a, u8 *cx, u8 *sink, Fifo &f, i16 *mag) {
    const int ntu = (J.shape == 0 || J.shape == 3) ? 1 : 4;
    const int Ntu = (J.shape == 0) ? J.N : (J.shape == 3) ? 4 : J.N / 2;
    const int s = (Ntu == 4) ? 0 : (Ntu == 8) ? 1 : (Ntu == 16) ? 2 : 3, ncg = Ntu >> 2;
    // ---- coding_unit header (:1271-1339)
    if (J.shape != 3) {
        const int np = (J.shape == 2) ? 4 : 1;
        int mp[4][3], hit[4];
        if (J.ctx_split >= 0) tk_bin(f, J.ctx_split, 0);
        if (J.N == 8) tk_bin(f, CX_PART, J.shape != 2);
        for (int i = 0; i < np; i++) { mpm_list(J.ml[i], J.ma[i], mp[i]); hit[i] = mpm_hit(mp[i], J.mode[i]); tk_bin(f, CX_PREV_INTRA, hit[i] >= 0); }
        for (int i = 0; i < np; i++) tk_mode_rest(f, mp[i], hit[i], J.mode[i]);
        tk_bin(f, CX_CHROMA_PRED, 0);
        if (J.shape != 2) tk_bin(f, CX_SPLIT_TU + (J.N == 32 ? 0 : J.N == 16 ? 1 : 2), J.shape == 1);
        tk_bin(f, CX_CBF_CHROMA, 0); tk_bin(f, CX_CBF_CHROMA, 0);
        fifo_drain(f, a, cx, T, sink);
    }
    for (int k = 0; k < ntu; k++) {
        const int mode = J.mode[(J.shape == 2) ? k : 0];
        const int st = scan_type_of(Ntu, mode);
        const int cbf = J.last[k] >= 0;
        if (J.shape != 3) tk_bin(f, CX_CBF_LUMA + (J.shape == 0 ? 1 : 0), cbf);
        if (!cbf && J.shape != 3) continue;
        // ---- residual_coding (:1172-1268)
        const int last = imax(J.last[k], 0);
        const u32 m0 = J.cg0[k], m1 = J.cg1[k];
        const int glast = last >> 4;
        {
            const int gp = T.cgpos[st][s][glast], in = T.incg[st][last & 15];
            tk_last_pos(f, Ntu, s, st, (gp >> 3) * 4 + (in >> 2), (gp & 7) * 4 + (in & 3));
        }
        fifo_drain(f, a, cx, T, sink);
        int c1 = 1;
        for (int g = glast; g >= 0; g--) {
            const int gp = T.cgpos[st][s][g], gy = gp >> 3, gx = gp & 7, bit = gy * 8 + gx;
            const int coded = (int)(((bit < 32 ? m0 >> bit : m1 >> (bit - 32))) & 1);
            const int rbit = bit + 1, bbit = bit + 8;
            const int right = (gx < ncg - 1) ? (int)(((rbit < 32 ? m0 >> rbit : m1 >> (rbit - 32))) & 1) : 0;
            const int below = (gy < ncg - 1) ? (int)(((bbit < 32 ? m0 >> bbit : m1 >> (bbit - 32))) & 1) : 0;
            const int pat = (below << 1) | right, dcg = (bit == 0), has_last = (g == glast);
            if (!dcg && !has_last) tk_bin(f, CX_CSBF + (pat != 0), coded);
            if (coded || dcg) {
                // 16 levels of this group, scan order
                int v[16];
                if (coded) { const i16 *p = J.lv[k] + g * 16; for (int n = 0; n < 16; n++) v[n] = p[n]; }
                else for (int n = 0; n < 16; n++) v[n] = 0;
                int sbase = 0; u32 padd = 0; u64 c4 = 0;
                if (Ntu == 4) c4 = T.c4tab[st];
                else { sbase = 9 + (Ntu >= 16 ? 12 : 0) + ((Ntu == 8 && st != 0) ? 6 : 0) + (dcg ? 0 : 3); padd = T.posadd[pat][st]; }
                int nnz = 0, signs = 0;
                const int nstart = has_last ? (last & 15) : 15;
                for (int n = 15; n >= 0; n--) {
                    if (n > nstart) continue;
                    const int is_last = has_last && n == nstart;
                    if (!is_last && (dcg || n != 0 || nnz > 0)) {
                        int ci;
                        if (dcg && n == 0) ci = 0;
                        else if (Ntu == 4) ci = (int)((c4 >> (4 * n)) & 15);
                        else ci = sbase + (int)((padd >> (2 * n)) & 3);
                        tk_bin(f, CX_SIG + ci, v[n] != 0);
                    }
                    if (v[n]) { mag[nnz++] = (i16)iabs(v[n]); signs = (signs << 1) | (v[n] < 0); }
                }
                if (nnz > 0) {
                    const int set = (dcg ? 0 : 2) + (c1 == 0);
                    int esc = nnz > 8, g2 = -1;
                    c1 = 1;
                    for (int j = 0; j < 8 && j < nnz; j++) {
                        const int big = mag[j] > 1;
                        tk_bin(f, CX_GT1 + 4 * set + c1, big);
                        if (big) { c1 = 0; if (g2 < 0) g2 = mag[j] > 2; else esc = 1; }
                        else if (c1 > 0 && c1 < 3) c1++;
                    }
                    if (c1 == 0 && g2 >= 0) { tk_bin(f, CX_GT2 + set, g2); esc |= g2; }
                    tk_bypass(f, signs, nnz);
                    if (esc) {
                        int base2 = 3, rice = 0;
                        for (int j = 0; j < nnz; j++) {
                            const int m = mag[j], r = m - (j < 8 ? base2 : 1);
                            if (f.n > FIFO_CAP - 6) fifo_drain(f, a, cx, T, sink);
                            if (r >= 0) { tk_remaining(f, r, rice); if (m > (3 << rice)) rice = imin(rice + 1, 4); }
                            if (m >= 2) base2 = 2;
                        }
                    }
                }
            }
            if (f.n > FIFO_CAP - 36 || coded || g == 0) fifo_drain(f, a, cx, T, sink);
        }
    }
    fifo_drain(f, a, cx, T, sink);
}
