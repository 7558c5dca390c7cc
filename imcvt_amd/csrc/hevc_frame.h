// hevc_frame.h — CU decision tree, CTU loop and frame driver on top of hevc_core.h.
// Reference behaviour: processCURecurs (src/HEVCe/HEVCe.c:1349-1559) and HEVCImageEncoder (:1569-1646).
//
// Schedule (round 1): the CUs of a CTU are visited in the reference's Z-order.  For one CU the three
// candidate sets run concurrently, one wavefront each:
//     wave 0: 2Nx2N / one TU   (35 modes)          — pipeline passes, then 35 trial coders
//     wave 1: 2Nx2N / four TUs (35 modes, TUs chained through that mode's own reconstruction)
//     wave 2: (8x8 only) the NxN chain: four PUs x 35 modes priced on a fresh coder, then the NxN trial
// followed by one decision (exact cost order and "last minimum wins" rule of :1439,:1475,:1520,:1545).
#pragma once
#include "hevc_core.h"

struct Avail { int l, bl, a, ar; };
HD Avail child_avail(const Avail &p, int k) {       // Z-order availability of quadrant k (:1375-1378)
    Avail c;
    c.l  = (k & 1) ? 1 : p.l;
    c.bl = (k == 0) ? p.l : (k == 2) ? p.bl : 0;
    c.a  = (k & 2) ? 1 : p.a;
    c.ar = (k == 0) ? p.a : (k == 1) ? p.ar : (k == 2) ? 1 : 0;
    return c;
}

struct FrameCtx {
    FrameJob job;
    Scratch sc;
    int out_pos;        // bytes of finished CTUs (incl. headers)
    int ctu_y, ctu_x;   // pixel origin of the current CTU
    int trace_n;
};

HD i16 *wave_lv(const Scratch &sc, int wave) { return sc.lv + (size_t)wave * LV_PER_WAVE; }
HD u8 *lane_bytes(const Scratch &sc, int wave, int lane) { return sc.bytes + ((size_t)wave * NMODE + lane) * TRIAL_BYTES; }

// neighbour context in 4x4 units relative to the CTU (apron row/column = neighbours outside the CTU)
HD int nb_size(const Shm &S, int uy, int ux) { return S.mapsz[uy + 1][ux + 1]; }
HD int nb_mode(const Shm &S, int uy, int ux) { return S.mapmode[uy + 1][ux + 1]; }

// ---- one candidate set: 2Nx2N with one TU (shape 0) or four TUs (shape 1); wave-uniform call --------------------
HD void eval_2Nx2N(Shm &S, const FrameCtx &F, int wave, int depth, int shape, int N, int y0, int x0, const Avail &av) {
    WaveMem &W = S.W[wave];
    const int q = F.job.q, h = N / 2;
    i16 *lv = wave_lv(F.sc, wave);
    LANES(l) {
        if (l < NMODE) {
            W.sse[l] = 0;
            for (int k = 0; k < 4; k++) { W.last[k][l] = -1; W.cgm[k][l][0] = 0; W.cgm[k][l][1] = 0; }
        }
    }
    wave_sync();
    P1Args P;
    P.q = q; P.only_mode = -1;
    if (shape == 0) {
        border_from_tile(S, W, N, y0, x0, av.l, av.bl, av.a, av.ar);
        P.N = N; P.y0 = y0; P.x0 = x0; P.k = 0; P.per_mode_border = 0; P.out_kind = OUT_NONE; P.lv = lv;
        p1_run(S, W, P);
    } else {
        for (int k = 0; k < 4; k++) {
            const Avail ca = child_avail(av, k);
            const int yk = y0 + (k >> 1) * h, xk = x0 + (k & 1) * h;
            if (k == 0) border_from_tile(S, W, h, yk, xk, ca.l, ca.bl, ca.a, ca.ar);
            else border_tu_split(S, W, N, y0, x0, k, av.l, av.bl, av.a, av.ar);
            P.N = h; P.y0 = yk; P.x0 = xk; P.k = k; P.per_mode_border = (k != 0); P.out_kind = OUT_T3SIDE;
            P.lv = lv + (size_t)k * NMODE * h * h;
            p1_run(S, W, P);
        }
    }
    // trial coders: lane m prices mode m from the CU's entry state
    const int uy = y0 >> 2, ux = x0 >> 2;
    const int big_l = N > nb_size(S, uy, ux - 1), big_a = N > nb_size(S, uy - 1, ux);
    const int ml = nb_mode(S, uy, ux - 1), ma = nb_mode(S, uy - 1, ux);
    const RdW rw = rd_weights(q);
    LANES(l) {
        if (l < NMODE) {
            TrialJob J;
            J.N = N; J.shape = shape; J.ctx_split = (N >= 16) ? CX_SPLIT_CU + big_l + big_a : -1;
            J.mode[0] = l; J.ml[0] = ml; J.ma[0] = ma;
            const int nt = shape ? 4 : 1, tn = shape ? h * h : N * N;
            for (int k = 0; k < nt; k++) {
                J.lv[k] = lv + (size_t)k * NMODE * tn + (size_t)l * tn;
                J.last[k] = W.last[k][l]; J.cg0[k] = W.cgm[k][l][0]; J.cg1[k] = W.cgm[k][l][1];
            }
            u8 *cx = W.u.p2.cx[l];
            for (int i = 0; i < CTX_STRIDE; i++) cx[i] = S.entry_cx[depth][i];
            Arith a = S.entry_a[depth];
            const int len0 = arith_len(a);
            Fifo f; f.buf = W.u.p2.fifo[l]; f.n = 0;
            trial_run(S.T, J, a, cx, lane_bytes(F.sc, wave, l) - a.cnt, f, W.u.p2.mag[l]);
            W.fin[l] = a;
            W.cost[l] = rd_cost(rw, W.sse[l], arith_len(a) - len0);
        }
    }
    wave_sync();
}

// ---- the NxN chain of an 8x8 CU (:1490-1543); wave-uniform call ---------------------------------------------------
HD void eval_NxN(Shm &S, const FrameCtx &F, int wave, int y0, int x0, const Avail &av) {
    WaveMem &W = S.W[wave];
    const int q = F.job.q;
    i16 *lv = wave_lv(F.sc, wave);
    const RdW rw = rd_weights(q);
    for (int k = 0; k < 4; k++) {
        const Avail ca = child_avail(av, k);
        const int yk = y0 + (k >> 1) * 4, xk = x0 + (k & 1) * 4;
        LANES(l) { if (l < NMODE) { W.sse[l] = 0; W.last[0][l] = -1; W.cgm[0][l][0] = 0; W.cgm[0][l][1] = 0; } }
        wave_sync();
        border_from_tile(S, W, 4, yk, xk, ca.l, ca.bl, ca.a, ca.ar);
        P1Args P;
        P.q = q; P.only_mode = -1; P.N = 4; P.y0 = yk; P.x0 = xk; P.k = 0; P.per_mode_border = 0; P.out_kind = OUT_REC4; P.lv = lv;
        p1_run(S, W, P);
        LANES(l) {                                      // residual bits on a fresh coder and fresh contexts (:1504-1518)
            if (l < NMODE) {
                TrialJob J;
                J.N = 8; J.shape = 3; J.ctx_split = -1; J.mode[0] = l; J.ml[0] = 0; J.ma[0] = 0;
                J.lv[0] = lv + l * 16; J.last[0] = W.last[0][l]; J.cg0[0] = W.cgm[0][l][0]; J.cg1[0] = 0;
                u8 *cx = W.u.p2.cx[l];
                for (int i = 0; i < CTX_STRIDE; i++) cx[i] = S.T.ctx_init[q][i];
                Arith a; arith_reset(a);
                Fifo f; f.buf = W.u.p2.fifo[l]; f.n = 0;
                trial_run(S.T, J, a, cx, lane_bytes(F.sc, wave, l), f, W.u.p2.mag[l]);
                W.cost[l] = rd_cost(rw, W.sse[l], arith_len(a));
            }
        }
        wave_sync();
        LANES(l) {                                      // pick the PU mode: later mode wins ties (:1520)
            if (l == 0) {
                int best = I32MAX, bm = 0;
                for (int m = 0; m < NMODE; m++) if (best >= W.cost[m]) { best = W.cost[m]; bm = m; }
                W.pu_mode[k] = bm; W.pu_sse[k] = W.sse[bm]; W.pu_last[k] = W.last[0][bm];
            }
        }
        wave_sync();
        LANES(l) {                                      // keep its levels and put its reconstruction in place (:1523-1524)
            if (l < 16) {
                const int bm = W.pu_mode[k];
                W.pu_lv[k][l] = (W.pu_last[k] >= 0) ? lv[bm * 16 + l] : (i16)0;
                S.rec[yk + (l >> 2) + 1][xk + (l & 3) + 1] = W.rec4[bm][l];
            }
        }
        wave_sync();
    }
    // price the whole NxN CU from the entry state (:1530-1543)
    const int uy = y0 >> 2, ux = x0 >> 2;
    LANES(l) {
        if (l == 0) {
            TrialJob J;
            J.N = 8; J.shape = 2; J.ctx_split = -1;
            for (int k = 0; k < 4; k++) { J.mode[k] = W.pu_mode[k]; J.lv[k] = W.pu_lv[k]; J.last[k] = W.pu_last[k]; J.cg0[k] = W.pu_last[k] >= 0; J.cg1[k] = 0; }
            J.ml[0] = nb_mode(S, uy, ux - 1);     J.ma[0] = nb_mode(S, uy - 1, ux);
            J.ml[1] = J.mode[0];                  J.ma[1] = nb_mode(S, uy - 1, ux + 1);
            J.ml[2] = nb_mode(S, uy + 1, ux - 1); J.ma[2] = J.mode[0];
            J.ml[3] = J.mode[2];                  J.ma[3] = J.mode[1];
            u8 *cx = W.u.p2.cx[0];
            for (int i = 0; i < CTX_STRIDE; i++) cx[i] = S.entry_cx[2][i];
            Arith a = S.entry_a[2];
            const int len0 = arith_len(a);
            Fifo f; f.buf = W.u.p2.fifo[0]; f.n = 0;
            trial_run(S.T, J, a, cx, lane_bytes(F.sc, wave, 0) - a.cnt, f, W.u.p2.mag[0]);
            W.fin[0] = a;
            W.nxn_cost = rd_cost(rw, W.pu_sse[0] + W.pu_sse[1] + W.pu_sse[2] + W.pu_sse[3], arith_len(a) - len0);
        }
    }
    wave_sync();
}

// ---- one CU after its children (if any) are done: evaluate the unsplit shapes, decide, commit ---------------------
// All waves call this with identical arguments.
HD void decide_cu(Shm &S, FrameCtx &F, int depth, int N, int y0, int x0, const Avail &av) {
    u8 *live_sink = F.job.out + F.out_pos;
    WAVES(w) {
        if (w == 0) eval_2Nx2N(S, F, 0, depth, 0, N, y0, x0, av);
        else if (w == 1) eval_2Nx2N(S, F, 1, depth, 1, N, y0, x0, av);
        else if (w == 2 && N == 8) eval_NxN(S, F, 2, y0, x0, av);
    }
    wg_sync();
    WAVES(w) LANES(l) {
        if (w == 0 && l == 0) {
            int best = (N > 8) ? S.split_cost[depth] : I32MAX, kind = 0, mode = 0;
            for (int m = 0; m < NMODE; m++) if (best >= S.W[0].cost[m]) { best = S.W[0].cost[m]; kind = 1; mode = m; }
            for (int m = 0; m < NMODE; m++) if (best >= S.W[1].cost[m]) { best = S.W[1].cost[m]; kind = 2; mode = m; }
            if (N == 8 && best >= S.W[2].nxn_cost) { best = S.W[2].nxn_cost; kind = 3; }
            S.win_kind = kind; S.win_mode = mode;
            if (F.sc.trace && F.trace_n + 8 <= F.sc.trace_cap) {
                i32 *t = F.sc.trace + F.trace_n;
                t[0] = F.ctu_y + y0; t[1] = F.ctu_x + x0; t[2] = N; t[3] = kind; t[4] = (kind == 3) ? (S.W[2].pu_mode[0] | S.W[2].pu_mode[1] << 8 | S.W[2].pu_mode[2] << 16 | S.W[2].pu_mode[3] << 24) : mode;
                t[5] = best; t[6] = 0; t[7] = 0;
            }
        }
    }
    if (F.sc.trace) F.trace_n += 8;
    wg_sync();
    const int kind = S.win_kind, mode = S.win_mode;
    if (kind != 0) {
        const int ww = (kind == 3) ? 2 : kind - 1, wl = (kind == 3) ? 0 : mode;
        const int cnt0 = S.entry_a[depth].cnt, cnt1 = S.W[ww].fin[wl].cnt;
        const u8 *src = lane_bytes(F.sc, ww, wl);
        WAVES(w) LANES(l) {
            const int tid = w * 64 + l;
            for (int i = tid; i < cnt1 - cnt0; i += WG_THREADS) live_sink[cnt0 + i] = src[i];
            if (tid < CTX_STRIDE) S.cx[tid] = S.W[ww].u.p2.cx[wl][tid];
            if (tid == 64) S.live = S.W[ww].fin[wl];
            if (tid >= 128 && tid < 128 + 64) {             // neighbour maps (:1444-1445, :1549-1553)
                const int n = N >> 2, i = (tid - 128) >> 3, j = (tid - 128) & 7;
                if (i < n && j < n) {
                    const int uy = (y0 >> 2) + i, ux = (x0 >> 2) + j;
                    S.mapsz[uy + 1][ux + 1] = (u8)N;
                    S.mapmode[uy + 1][ux + 1] = (u8)((kind == 3) ? S.W[2].pu_mode[i * 2 + j] : mode);
                }
            }
        }
        wg_sync();
        if (kind != 3) {                                    // rebuild the winner's reconstruction in the tile
            WAVES(w) {
                if (w == 0) {
                    WaveMem &W = S.W[0];
                    P1Args P;
                    P.q = F.job.q; P.only_mode = mode; P.k = 0; P.per_mode_border = 0; P.out_kind = OUT_TILE; P.lv = (i16 *)0;
                    if (kind == 1) {
                        border_from_tile(S, W, N, y0, x0, av.l, av.bl, av.a, av.ar);
                        P.N = N; P.y0 = y0; P.x0 = x0;
                        p1_run(S, W, P);
                    } else {
                        const int h = N / 2;
                        for (int k = 0; k < 4; k++) {
                            const Avail ca = child_avail(av, k);
                            P.N = h; P.y0 = y0 + (k >> 1) * h; P.x0 = x0 + (k & 1) * h;
                            border_from_tile(S, W, h, P.y0, P.x0, ca.l, ca.bl, ca.a, ca.ar);
                            p1_run(S, W, P);
                        }
                    }
                }
            }
        }
        wg_sync();
    }
}

// snapshot the live coder as the entry state of `depth`, optionally after coding split_cu_flag=1 (:1363-1364, :1403)
HD void enter_cu(Shm &S, FrameCtx &F, int depth, int N, int y0, int x0, int code_split) {
    u8 *live_sink = F.job.out + F.out_pos;
    WAVES(w) LANES(l) {
        const int tid = w * 64 + l;
        if (tid < CTX_STRIDE) S.entry_cx[depth][tid] = S.cx[tid];
        if (tid == 64) S.entry_a[depth] = S.live;
    }
    wg_sync();
    if (code_split) {
        WAVES(w) LANES(l) {
            if (w == 0 && l == 0) {
                const int uy = y0 >> 2, ux = x0 >> 2;
                const int big_l = N > nb_size(S, uy, ux - 1), big_a = N > nb_size(S, uy - 1, ux);
                Arith a = S.live;
                code_bin(a, S.cx, S.T, live_sink, CX_SPLIT_CU + big_l + big_a, 1);
                S.live = a;
            }
        }
        wg_sync();
    }
}

// cost of keeping the split (:1408-1409): SSE of the children's reconstruction + bits spent since entry
HD void price_split(Shm &S, FrameCtx &F, int depth, int N, int y0, int x0) {
    WAVES(w) LANES(l) { if (w == 0 && l == 0) S.red[0] = 0; }
    wg_sync();
    WAVES(w) LANES(l) {
        const int tid = w * 64 + l;
        int part = 0;
        for (int i = tid; i < N * N; i += WG_THREADS) {
            const int y = y0 + i / N, x = x0 + i % N;
            const int d = (int)S.org[y][x] - S.rec[y + 1][x + 1];
            part += d * d;
        }
        if (part) lds_add(&S.red[0], part);
    }
    wg_sync();
    WAVES(w) LANES(l) {
        if (w == 0 && l == 0) {
            const RdW rw = rd_weights(F.job.q);
            S.split_cost[depth] = rd_cost(rw, S.red[0], arith_len(S.live) - arith_len(S.entry_a[depth]));
        }
    }
    wg_sync();
}

HD void encode_ctu(Shm &S, FrameCtx &F) {
    const FrameJob &J = F.job;
    const int cy = F.ctu_y, cx = F.ctu_x;
    Avail a32; a32.l = cx > 0; a32.bl = 0; a32.a = cy > 0; a32.ar = (cy > 0) && (cx + 32 < J.wp);
    // ---- load the CTU: source pixels replicate the original edges, neighbours come from the padded reconstruction (:1613-1621)
    WAVES(w) LANES(l) {
        const int tid = w * 64 + l;
        for (int i = tid; i < 1024; i += WG_THREADS) {
            const int y = i >> 5, x = i & 31;
            S.org[y][x] = J.img[(size_t)clip3(cy + y, 0, J.h - 1) * J.w + clip3(cx + x, 0, J.w - 1)];
        }
        if (tid < 32) S.rec[tid + 1][0] = J.rcon[(size_t)clip3(cy + tid, 0, J.hp - 1) * J.wp + clip3(cx - 1, 0, J.wp - 1)];
        if (tid >= 32 && tid < 32 + 65) {
            const int j = tid - 32 - 1;
            S.rec[0][j + 1] = J.rcon[(size_t)clip3(cy - 1, 0, J.hp - 1) * J.wp + clip3(cx + j, 0, J.wp - 1)];
        }
        // neighbour-map aprons: above row keeps sizes but forgets modes (:1633-1636); left column comes from the previous CTU
        if (tid >= 100 && tid < 100 + 10) {
            const int j = tid - 100;      // apron column index 0..9 <-> unit x = j-1
            S.mapsz[0][j] = (u8)((cy > 0 && j >= 1 && j <= 8) ? F.sc.above_sz[(cx >> 2) + j - 1] : 32);
            S.mapmode[0][j] = 1;
            if (cx == 0 && j < 9) { S.mapsz[j + 1][0] = 32; S.mapmode[j + 1][0] = 1; }
        }
    }
    wg_sync();

    enter_cu(S, F, 0, 32, 0, 0, 1);
    for (int i16_ = 0; i16_ < 4; i16_++) {
        const int y16 = (i16_ >> 1) * 16, x16 = (i16_ & 1) * 16;
        const Avail a16 = child_avail(a32, i16_);
        enter_cu(S, F, 1, 16, y16, x16, 1);
        for (int i8_ = 0; i8_ < 4; i8_++) {
            const int y8 = y16 + (i8_ >> 1) * 8, x8 = x16 + (i8_ & 1) * 8;
            const Avail a8 = child_avail(a16, i8_);
            enter_cu(S, F, 2, 8, y8, x8, 0);
            decide_cu(S, F, 2, 8, y8, x8, a8);
        }
        price_split(S, F, 1, 16, y16, x16);
        decide_cu(S, F, 1, 16, y16, x16, a16);
    }
    price_split(S, F, 0, 32, 0, 0);
    decide_cu(S, F, 0, 32, 0, 0, a32);

    // ---- store the reconstruction, end_of_slice_segment_flag, hand the CTU's bytes over (:1625-1630)
    u8 *live_sink = J.out + F.out_pos;
    WAVES(w) LANES(l) {
        const int tid = w * 64 + l;
        for (int i = tid; i < 1024; i += WG_THREADS) {
            const int y = i >> 5, x = i & 31;
            J.rcon[(size_t)(cy + y) * J.wp + cx + x] = S.rec[y + 1][x + 1];
        }
        if (tid < 8) {
            F.sc.above_sz[(cx >> 2) + tid] = S.mapsz[8][tid + 1];
            S.mapsz[tid + 1][0] = S.mapsz[tid + 1][8];          // right column becomes the next CTU's left apron
            S.mapmode[tid + 1][0] = S.mapmode[tid + 1][8];
        }
        if (tid == 64) {
            Arith a = S.live;
            code_terminate(a, live_sink, (cy + 32 >= J.hp) && (cx + 32 >= J.wp));
            S.live = a;
        }
    }
    wg_sync();
    F.out_pos += S.live.cnt;
    wg_sync();
    WAVES(w) LANES(l) { if (w == 0 && l == 0) S.live.cnt = 0; }
    wg_sync();
}

// Encode one frame with one workgroup.  `hdr` = the stream headers, prepared on the host (:664-690).
HD void encode_frame(Shm &S, const Tables *gT, const FrameJob &job, const Scratch &sc, const u8 *hdr) {
    FrameCtx F;
    F.job = job; F.sc = sc; F.out_pos = job.hdr_len; F.trace_n = 0;
    WAVES(w) LANES(l) {
        const int tid = w * 64 + l;
        const u32 *src = (const u32 *)gT; u32 *dst = (u32 *)&S.T;
        for (int i = tid; i < (int)(sizeof(Tables) / 4); i += WG_THREADS) dst[i] = src[i];
        for (int i = tid; i < job.hdr_len; i += WG_THREADS) job.out[i] = hdr[i];
    }
    wg_sync();
    WAVES(w) LANES(l) {
        const int tid = w * 64 + l;
        if (tid < CTX_STRIDE) S.cx[tid] = S.T.ctx_init[job.q][tid];
        if (tid == 64) arith_reset(S.live);
    }
    wg_sync();
    for (int cy = 0; cy < job.hp; cy += 32)
        for (int cx = 0; cx < job.wp; cx += 32) { F.ctu_y = cy; F.ctu_x = cx; encode_ctu(S, F); }
    WAVES(w) LANES(l) {
        if (w == 0 && l == 0) {
            Arith a = S.live;
            arith_finish(a, job.out + F.out_pos);                                      // :1639-1640
            *job.out_len = F.out_pos + a.cnt;
        }
    }
    wg_sync();
}
