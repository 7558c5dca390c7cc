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

HD int slot_of(int N) { return N == 32 ? SLOT_32 : SLOT_16; }
struct Avail { int l, bl, a, ar; };
HD Avail unpack_avail(int m) { Avail a; a.l = m & 1; a.bl = (m >> 1) & 1; a.a = (m >> 2) & 1; a.ar = (m >> 3) & 1; return a; }
HD int pack_avail(const Avail &a) { return a.l | a.bl << 1 | a.a << 2 | a.ar << 3; }
HD Avail child_avail(const Avail &p, int k) {       // Z-order availability of quadrant k (:1375-1378)
    Avail c;
    c.l  = (k & 1) ? 1 : p.l;
    c.bl = (k == 0) ? p.l : (k == 2) ? p.bl : 0;
    c.a  = (k & 2) ? 1 : p.a;
    c.ar = (k == 0) ? p.a : (k == 1) ? p.ar : (k == 2) ? 1 : 0;
    return c;
}
#define F (SM.F)
// debug heartbeat (thread 0; -DIMCVT_HB builds — the three 64-bit words are LDS the shipped image does not have to spare): the longest interval between two
// beats of this workgroup and when it began
#ifdef IMCVT_HB
HD void hb_beat() {
    const unsigned long long now = wd_now();
    if (F.hb_last && now - F.hb_last > F.hb_gap) { F.hb_gap = now - F.hb_last; F.hb_when = F.hb_last; }
    F.hb_last = now;
}
HD void hb_set(unsigned long long v) { F.hb_last = v; }
#define HB_GAP F.hb_gap
#define HB_WHEN F.hb_when
#else
HD void hb_beat() {}
HD void hb_set(unsigned long long) {}
#define HB_GAP ((unsigned long long)(u32)F.kept)      // (debug buffer: a helper's requests served — helper_loop leaves the count there — or what the main workgroup's last frame kept)
#define HB_WHEN (dbg[4 * blk + 1])      // (left as it is: the wavefronts' SIMDs, kernel_main)
#endif
#ifndef NXN_PRIO_SOLO
#define NXN_PRIO_SOLO 1
#endif
#ifndef ROLE8_PERM
// nibble w_: the role physical wavefront w_ runs in the 8x8 CUs of a wide workgroup (decide_cu).  Roles 5 (a pass + byte half of the one-TU set) and 6 (byte half of the
// pipe wave's streams) swapped against round 5: the PU chain's SIMD-mate is the one-TU set's light partner, the four-TU passes share theirs with the wavefront that idles
// until PU 2 is decided — 64 frames 2.30 -> 2.27 s, one frame unchanged; the other placements tried were 2 - 5 % slower (profiles/r06i_role_perm_ab.log)
#define ROLE8_PERM 0x75643210u
#endif
#define HDR_MAX 96       // bytes reserved per frame for the stream headers the host prepares (:664-690)

HD u16 *wave_tok(const Scratch &sc, int wave) { return sc.tok + (size_t)wave * TOK_SLOTS * TOK_CAP; }
HD u8 *lane_bytes(const Scratch &sc, int wave, int lane) { return sc.bytes + ((size_t)wave * NMODE + lane) * TRIAL_BYTES; }

// neighbour context in 4x4 units relative to the CTU (apron row/column = neighbours outside the CTU)
HD int nb_size(int uy, int ux) { return SM.mapsz[uy + 1][ux + 1]; }
HD int nb_mode(const int uy, int ux) { return SM.mapmode[uy + 1][ux + 1]; }

// ---- the 2Nx2N candidate sets of a CU: one TU (shape 0, owned by wave 0) and four TUs (shape 1, owned by wave 1) -------------
// Wave-uniform call.  An 8x8 CU is evaluated by the two owners alone (wave 2 runs the NxN chain meanwhile).  For a 16x16 /
// 32x32 CU all three waves call this: wave 2, which has no candidate set of its own there, runs the pipeline passes of the
// upper modes of BOTH sets (modes are independent; a four-TU mode's TU chain stays inside one wave), writing tokens, SSE
// and token counts straight into the owners' arrays; the owners then price all 35 of their candidates.
struct P1Item { int own, shape, lo, hi; };
// TU 0 of the four-TU shape of an 8x8 CU is the same 4x4 block, predicted from the same samples in the same 35 modes, as PU 0 of
// the NxN chain (:1459-1466 with isub = 0 against :1497-1512 with isub = 0): same levels, same tokens (an all-zero block apart,
// whose residual syntax only PU pricing codes, :1515), same reconstruction.  The PU wave runs that pass anyway and first; the
// four-TU wave takes its results instead of repeating it: lane c appends candidate c's tokens to its own stream and keeps the
// bottom row / right column of its reconstruction for TUs 1..3.
#ifndef TU0_SHARE
#define TU0_SHARE 1
#endif
#ifndef NXN_UNI
#define NXN_UNI 1         // launches without a pipe wave: the NxN trial of an 8x8 CU runs on wave-uniform values, i.e. on the scalar unit (hevc_core.h stream_run_uni)
#endif
#ifndef PU_HINTS
#define PU_HINTS 1        // launches with a pipe wave: the PU candidates of an 8x8 CU carry state hints and are priced without context copies
#endif
HDN void tu0_from_pu0(int wave_, u16 *tok1_) {          // (out of line: inlined it costs eval_2Nx2N's passes registers)
    const int wave = uni_i(wave_); u16 *const tok1 = uni_p(tok1_);
    WaveMem &W = WM(wave);
    const WaveMem &W2 = WM(2);
    const u16 *tok2 = wave_tok(F.sc, 2);
    while (lds_ld_i32(&SM.pu0_ready) == 0) pipe_pause();
    wave_sync();
    LANES(l) {
        if (l < NMODE) {
            const int c = l, nz = W2.tnz[c], cnt = nz ? W2.tokn[c] - 7 : 1;
            const u16 *src = tok2 + (size_t)c * TOK_CAP + 7;        // [7] cbf_luma, [8..] the rest: 16-byte blocks from token 8 on
            LaneStream ls;
            TokW w = ls_begin(ls, W, c, lane_row(W, l), tok1 + (size_t)c * TOK_CAP);
            tk_put(w, (int)(u16)g_ld16((const i16 *)src));
            NOUNROLL
            for (int i = 1; i < cnt; i += 8) {                      // (a PU candidate's stream ends on a block boundary, padded with idle tokens)
                const U4 b = g_ld128(src + i);
                for (int j = 0; j < 8; j++) to_put(w.o, w.n + j, (int)tok_of(b, j));
                w.n += imin(8, cnt - i);
                if (w.n >= LCAP - 16) ls_flush(ls, w);
            }
            ls_end(ls, w, W, c);
            W.tnz[c] = (u8)nz;
            W.sse[c] += W2.sse[c];
            for (int i = 0; i < 4; i++) { SM.X.t3row[c][0][i] = W2.u.w2.rec4[c][12 + i]; SM.X.t3col[c][0][i] = W2.u.w2.rec4[c][4 * i + 3]; }
        }
    }
    wave_sync_lds();
    LANES(l) { if (l == 0) lds_st_i32(&SM.pu0_taken, 1); }
}
// ---- wide workgroups: the two halves of a trial coder (hevc_core.h stream_seg_R / stream_seg_L) --------------------------------
// owner: counters zeroed, then the partner is told to start (its generation counter)
HD void split_start(SplitQ &q) {
    LANES(l) { split_reset(q, l); }
    wave_sync_lds();
    LANES(l) { if (l == 0) lds_st_i32(&q.go, lds_ld_i32(&q.go) + 1); }
}
HD void split_flag(i32 *flag, SplitQ &q) {              // owner: everything this wavefront stored to LDS so far is visible to whoever sees the flag
    wave_sync_lds();
    LANES(l) { if (l == 0) lds_st_i32(flag, lds_ld_i32(&q.go)); }
}
HD void split_await(const i32 *flag, const SplitQ &q) { // returns once *flag names the running generation
    while (lds_ld_i32(flag) != lds_ld_i32(&q.go)) pipe_pause();
    wave_sync();
}
HD void ctx_copy(u8 *dst, const u8 *src) { for (int i = 0; i < CTX_STRIDE; i += 4) *(u32a *)(dst + i) = *(const u32a *)(src + i); }
// partner of wave `own` (0: one TU, 1: four TUs) of an 8x8 CU: the byte half of its 35 trial coders; writes their final coder states and costs
HDN void partner_trial(int own_, int depth_) {
    const int own = uni_i(own_); const int depth = uni_i(depth_);
    PartnerMem &X = XM(own); SplitQ &q = X.q;
    WaveMem &W = WM(own);
    while (lds_ld_i32(&q.go) == lds_ld_i32(&q.done)) pipe_pause();
    wave_sync();                                        // token counts are final, the streams are in memory
    const RdW rw = rd_weights(F.job.q);
    u8 *const ubytes = uniform_ptr(F.sc.bytes);
    LANES(l) {
        const int on = l < NMODE, ll = on ? l : 0;
        Arith a = SM.entry_a[depth];
        const Arith a0 = a;
        const int len0 = arith_len(a), n = on ? W.tokn[ll] : 0;
        u8 *gbuf = ubytes + (size_t)(own * NMODE + ll) * TRIAL_BYTES;
        LeadSink sink; lsink_begin(sink, a0, X.lm[ll].ring, gbuf);
        int blk = 0, qn = 0;
        stream_seg_L(a, sink, qn, q, l, blk, n);
        trial_finish(a, a0, sink, qn, on);
        split_await(&q.rdone, q);
        if (on) {
            a.range = q.range_out[l];
            W.fin[l] = pack_arith(a);
            W.cost[l] = rd_cost(rw, W.sse[l], arith_len(a) - len0);
        }
    }
    split_flag(&q.done, q);
}
// sequence number of PU k of the 8x8 CU being walked (wide workgroups)
HD u32 pu_seq_of(int k) { return 4u * (u32)(lds_ld_i32(&WCTL.cu8) - 1) + (u32)k + 1u; }
// The partners' shares of PU k (wide workgroups, pu_step_wide).  pu_recon_k: reconstructions and SSE of the 35 candidates.  pu_price: the byte
// half of their pricing (fresh coder, :1504-1518) — over the range half's records for the first part of the tokens, over the remaining-level rows
// themselves (bypass chunks: no range half needed) for the rest — and the costs.
HD void pu_recon_k(int y0, int x0, int k) {
    P1Args P;
    P.q = F.job.q; P.only_mode = -1; P.shape = 3; P.tok = (u16 *)0; P.N = 4; P.y0 = y0 + (k >> 1) * 4; P.x0 = x0 + (k & 1) * 4; P.k = 0; P.per_mode_border = 0; P.out_kind = OUT_REC4;
    P.own = 2; P.c_lo = 0; P.c_hi = NMODE; P.hint = 1; P.rec8 = nullptr;
    const long long tq = prof_now();                    // (IMCVT_PROF builds: the partners' times go to columns wave 2's row leaves empty — p2_32 reconstructions, p2_16 waiting for the first counts, p2_8 byte half over the first part, p2_nxn waiting for rows and final range, p1_16 byte half over the rows, x3 costs; idle: the remaining-level rows)
    pu_recon(2, P, pu_seq_of(k));
    prof_add_row(2, PF_P2_32, tq);
}
HD void pu_price(int k) {
    PartnerMem &X = XM(2); SplitQ &q = X.q;
    WaveMem &W = WM(2);
    PuX &U = PUX;
    const RdW rw = rd_weights(F.job.q);
    const u32 seq = pu_seq_of(k);
    u8 *const ubytes = uniform_ptr(F.sc.bytes);
    long long tq = prof_now();
    while ((u32)lds_ld_i32((const i32 *)&U.pu_seq) != seq) pipe_pause();      // (the step's generation has started: q.go is this step's)
    while (lds_ld_i32(&q.go) == lds_ld_i32(&q.done)) pipe_pause();
    split_await(&q.mid, q);                             // the first parts' token counts are in place
    prof_add_row(2, PF_P2_16, tq); tq = prof_now();
    LANES(l) {
        const int on = l < NMODE, ll = on ? l : 0;
        Arith a; arith_reset(a);
        const int na = on ? U.na[ll] : 0;
        const Arith a0 = a;
        LeadSink sink; lsink_begin(sink, a0, X.lm[ll].ring, ubytes + (size_t)(2 * NMODE + ll) * TRIAL_BYTES);
        int blk = 0, qn = 0;
        stream_seg_L1(a, sink, qn, q, l, blk, na);
        tl_mark(44 + k);                                 // 44 .. 47: byte half through the first part of PU k
        prof_add_row(2, PF_P2_8, tq); tq = prof_now();
        while ((u32)lds_ld_i32((const i32 *)&U.b_seq) != seq) pipe_pause();
        wave_sync_lds();
        const int nb = on ? U.bcnt[ll] : 0;
        split_await(&q.rdone, q);                                       // the range half is through the first part: the range it ended on (the rest is bypass chunks, which this half takes from the rows itself)
        prof_add_row(2, PF_P2_NXN, tq); tq = prof_now();
        stream_seg_L1_byp(a, sink, qn, U.brow[ll], nb, on ? q.range_out[ll] : 510);
        tl_mark(48 + k);                                 // 48 .. 51: ... and through its rows
        prof_add_row(2, PF_P1_16, tq); tq = prof_now();
        trial_finish(a, a0, sink, qn, on);                              // (the bytes themselves are never read, :1518)
        const int len = arith_len(a);
        if (k == 1) tl_mark(58);                         // 58: PU 1: guard checked
        while ((u32)lds_ld_i32((const i32 *)&U.r_seq) != seq) pipe_pause();      // SSE and reconstructions of this PU's candidates are in place (long since)
        wave_sync_lds();
        if (k == 1) tl_mark(59);                         // 59: PU 1: reconstructions seen
        if (on) W.cost[l] = rd_cost(rw, W.sse[l], len);
        if (k == 1) tl_mark(60);                         // 60: PU 1: costs stored
        prof_add_row(2, PF_X3, tq);
    }
    split_flag(&q.done, q);
}
// Who does what for PU k of an 8x8 CU (wide workgroups), next to the PU wave (pu_step_wide):
//   k = 0..2   wave 7: remaining-level rows      pipe wave (3): reconstructions + SSE, then the pricing's byte half (it has nothing of its own to do until PU 2 is decided)
//   k = 3      wave 7: remaining-level rows, then the pricing's byte half      the PU wave itself: reconstructions + SSE, once its range half is through
//              (nobody else is free by then: the pipe wave and wave 6 are coding PUs 0..2, waves 0 / 5 and 1 / 4 are still in the 2Nx2N sets' trial coders —
//              with the rows on wave 5 or 4 the PU wave waited 11 k cycles per step for them, profiles/r05ae, r05af)
HDN void partner_pu(int y0_, int x0_) {
    (void)y0_; (void)x0_;
#ifndef IMCVT_HOSTEMU
    if (F.prio_base) SETPRIO(3); else SETPRIO(NXN_PRIO_SOLO);      // (part of the PU chain, the longest of an 8x8 CU: as eval_NxN)
#endif
    for (int k = 0; k < 4; k++) pu_part_b(pu_seq_of(k));
    pu_price(3);
#ifndef IMCVT_HOSTEMU
    if (F.prio_base) SETPRIO(2); else SETPRIO(0);
#endif
}
// the pipe wave before PU 2 is decided: reconstructions and pricing of PUs 0..2
HDN void partner_pu_early(int y0_, int x0_) {
    const int y0 = uni_i(y0_); const int x0 = uni_i(x0_);
#ifndef IMCVT_HOSTEMU
    if (F.prio_base) SETPRIO(3); else SETPRIO(NXN_PRIO_SOLO);      // (part of the PU chain)
#endif
    for (int k = 0; k < 3; k++) { pu_recon_k(y0, x0, k); pu_price(k); }
}
// owner of a 2Nx2N candidate set in a wide workgroup: the range half of its 35 trial coders (final coder states and costs are the partner's to
// write, before the barrier that follows the candidate sets).  Out of line, so that the other launch shapes' eval_2Nx2N is not charged its registers.
HDN void coder_range_half(int wave_, int depth_) {
    const int wave = uni_i(wave_); const int depth = uni_i(depth_);
    WaveMem &W = WM(wave);
    SplitQ &q = XM(wave).q;
    const u16 *tok = wave_tok(F.sc, wave);
    split_start(q);
    LANES(l) {
        const int on = l < NMODE, ll = on ? l : 0;
        int range = SM.entry_a[depth].range, blk = 0;
        if (on) ctx_copy(W.u.p2.cx[ll], SM.entry_cx[depth]);
        stream_seg_R<false>(range, W.u.p2.cx[ll], q, l, blk, tok + (size_t)ll * TOK_CAP, on ? W.tokn[ll] : 0);
        if (on) q.range_out[l] = range;
    }
    split_flag(&q.rdone, q);
}
HD void seg_close(WaveMem &W, int k);
// ---- partner workgroups (8x8 CUs' 2Nx2N sets on a second compute unit, see "8x8 CUs with a partner workgroup" below) ---------------------------------------
// every candidate keeps its reconstruction (64 bytes, raster order): the winner's is answered without running its shape once more.  The store lies in the LDS slice of a
// lender wavefront the partner workgroup does not use (wave 5's: 2 x 35 x 64 bytes of 8960).
HD u8 *rec8_of(int shape) { return wave_mem_ptr(WAVE_A_PARTNER) + shape * (NMODE * 64); }
static_assert(2 * NMODE * 64 <= (int)sizeof(WaveMem), "the candidates' reconstructions fit a lender's slice");
// The four-TU set of a partner workgroup is its longest chain: four 4x4 passes in a row (TU k + 1 of a mode predicts from TU k's reconstruction of that mode, :1459, :1466),
// 29 k cycles each with their tokens — which are 55 % of a pass and which nothing on the chain waits for.  So wave 1 runs predict / DST / RDOQ, publishes the levels (PuX::lev,
// the PU steps' hand-over: a partner workgroup walks no PUs), goes on with the inverse, the reconstruction and the edges the next TU predicts from — and a second wavefront
// (fourtu_tokens_solo) makes the TU's tokens from the levels, appends them to the candidates' streams and closes the segment for the coder wavefront (partner_fourtu).
// Same levels, same tokens in the same order, same reconstruction as p1_run_4.
HDN_EVAL void fourtu_passes_solo(int y0_, int x0_, int avm_) {
    const int y0 = uni_i(y0_); const int x0 = uni_i(x0_); const int avm = uni_i(avm_);
    const Avail av = unpack_avail(avm);
    WaveMem &W = WM(1);
    WideCtl &C = WCTL;
    P1Args P;
    P.q = F.job.q; P.only_mode = -1; P.hint = 0; P.own = 1; P.c_lo = 0; P.c_hi = NMODE; P.shape = 1; P.tok = wave_tok(F.sc, 1); P.rec8 = rec8_of(1);
    const QConst Q = qconst<0>(P.q);
    for (int k = 0; k < 4; k++) {
        const Avail ca = child_avail(av, k);
        const int yk = y0 + (k >> 1) * 4, xk = x0 + (k & 1) * 4;
        if (k == 0) border_from_tile(1, 4, yk, xk, ca.l, ca.bl, ca.a, ca.ar);
        else border_tu_split(8, y0, x0, k, av.l, av.bl, av.a, av.ar, 0, NMODE);
        P.N = 4; P.y0 = yk; P.x0 = xk; P.k = k; P.per_mode_border = (k != 0); P.out_kind = OUT_T3SIDE;
        while (lds_ld_i32(&C.lv_taken) != lds_ld_i32(&C.lv_seq)) pipe_pause();      // the token wavefront has taken the TU before
        if (k == 0) wave_sync(); else wave_sync_lds();                              // (TU 0: the candidates' headers are in memory before anybody is told to go on from them)
        LANES(l) {
            const int c = l, live = c < NMODE;
            int x[4][4], pr[4][4], t[4][4];
            int any = 0;
            if (live) any = pu_stage1(W, P, c, x, pr);
            wave_sync_lds();
            if (l == 0) lds_st_i32(&C.lv_seq, lds_ld_i32(&C.lv_seq) + 1);          // levels published
            if (live) {
                int part = 0;
                if (any) {
                    for (int r = 0; r < 4; r++) for (int cc = 0; cc < 4; cc++) x[r][cc] = clip16(x[r][cc] * (1 << Q.dqs));
                    for (int j = 0; j < 4; j++) {
                        const int a = x[0][j], b = x[1][j], cc_ = x[2][j], d = x[3][j];
                        t[0][j] = clip16((29 * a + 74 * b + 84 * cc_ + 55 * d + 64) >> 7);
                        t[1][j] = clip16((55 * a + 74 * b - 29 * cc_ - 84 * d + 64) >> 7);
                        t[2][j] = clip16((74 * (a - cc_ + d) + 64) >> 7);
                        t[3][j] = clip16((84 * a - 74 * b + 55 * cc_ - 29 * d + 64) >> 7);
                    }
                    for (int i = 0; i < 4; i++) {
                        const int a = t[i][0], b = t[i][1], cc_ = t[i][2], d = t[i][3];
                        x[i][0] = clip16((29 * a + 74 * b + 84 * cc_ + 55 * d + 2048) >> 12);
                        x[i][1] = clip16((55 * a + 74 * b - 29 * cc_ - 84 * d + 2048) >> 12);
                        x[i][2] = clip16((74 * (a - cc_ + d) + 2048) >> 12);
                        x[i][3] = clip16((84 * a - 74 * b + 55 * cc_ - 29 * d + 2048) >> 12);
                    }
                } else {
                    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) x[i][j] = 0;
                }
                u32 oww[4];
                for (int yi = 0; yi < 4; yi++) oww[yi] = *(const u32a *)&SM.org[yk + yi][xk];
                for (int yi = 0; yi < 4; yi++) {
                    const u32 ow = oww[yi];
                    u32 rw4 = 0;
                    for (int xi = 0; xi < 4; xi++) {
                        const int rc = clip3(x[yi][xi] + pr[yi][xi], 0, 255);
                        const int d = (int)((ow >> (8 * xi)) & 255) - rc;
                        part += d * d;
                        rw4 |= (u32)rc << (8 * xi);
                        if (k < 3 && yi == 3) SM.X.t3row[c][k][xi] = (u8)rc;
                        if (k < 3 && xi == 3) SM.X.t3col[c][k][yi] = (u8)rc;
                    }
                    *(u32a *)(P.rec8 + c * 64 + ((k >> 1) * 4 + yi) * 8 + (k & 1) * 4) = rw4;
                }
                W.sse[c] += part;
            }
        }
        wave_sync_lds();
        tl_mark(36 + k);                                 // 36 .. 39: the four-TU set's TU k passed (without its tokens)
    }
}
// ... and the wavefront that makes those tokens (wave 6 of a partner workgroup): TU by TU behind wave 1, into the candidates' streams (wave 1's rows, counts and streams:
// wave 1 does not touch them after the headers), a segment closed per TU.  The token part of p1_run_4 for shape 1, no hints.
HDN void fourtu_tokens_solo() {
    WaveMem &W = WM(1);
    WideCtl &C = WCTL;
    const Tables &T = SM.T;
    u16 *const tok = wave_tok(F.sc, 1);
    for (int k = 0; k < 4; k++) {
        while (lds_ld_i32(&C.lv_seq) == lds_ld_i32(&C.lv_taken)) pipe_pause();      // the next TU's levels are published
        wave_sync_lds();
        LANES(l) {
            const int c = l, live = c < NMODE;
            int x[4][4], pr[4][4];
            if (live) pu_levels(c, x, pr);
            wave_sync_lds();
            if (l == 0) lds_st_i32(&C.lv_taken, lds_ld_i32(&C.lv_taken) + 1);      // taken: wave 1 may publish the TU after
            if (live) {
                int any = 0;
                for (int r = 0; r < 4; r++) for (int cc = 0; cc < 4; cc++) any |= x[r][cc];
                const int st = scan_type_of(4, c);
                Lv16 L; u32 nzm = 0, mcode = 0;
                if (any) nzm = scan_levels(L, x, st, 0, &mcode);
                LaneStream ls;
                TokW w = ls_begin(ls, W, c, lane_row(W, l), tok + (size_t)c * TOK_CAP);
                tk_bin(w, CX_CBF_LUMA, nzm != 0);
                if (nzm != 0) {
                    const int in = T.incg[st][hibit(nzm)];
                    const LastPos lp = last_pos_prep(0, st, in >> 2, in & 3);
                    w.n = last_pos_emit<0, true>(w.o, w.n, lp);
                    TgB B;
                    const int cfg4 = TG_DC | TG_LAST | st << TG_ST;
                    w.n = tokg_a_fast<0>(w.o.tb, w.n, L, nzm, mcode, cfg4, B) & 0xFFFF;
                    if (B.esc) {
                        ls_flush(ls, w);
                        w.n = tokg_b<true, true, 15, 8>(w.o, w.n, L, B);
                        if (w.n > 14) ls_flush(ls, w);
                        w.n = tokg_b<true, true, 7, 0>(w.o, w.n, L, B);
                    }
                    w.n = tokg_end<true, true>(w.o, w.n, B);
                }
                ls_end(ls, w, W, c);
                W.tnz[c] = (u8)(nzm != 0);
            }
        }
        wave_sync_lds();
        seg_close(W, k);
    }
}
// A lender wavefront (wide workgroups): candidates lo .. hi-1 of the one-TU set of the 8x8 CU at (y0, x0), exactly as wave 0 runs its own
// (eval_2Nx2N: same border, same pass, tokens / counts / SSE into wave 0's arrays and streams) on this wavefront's own slice.
HDN void lend_passes(int wave_, int li_, int lo_, int hi_, int y0_, int x0_, int avm_) {
    const int wave = uni_i(wave_); const int li = uni_i(li_); const int lo = uni_i(lo_); const int hi = uni_i(hi_); const int y0 = uni_i(y0_); const int x0 = uni_i(x0_); const int avm = uni_i(avm_);
    WideCtl &C = WCTL;
    while (lds_ld_i32(&C.a_go) == lds_ld_i32(&C.lend_done[li])) pipe_pause();
    wave_sync();                                        // the candidates' headers are in memory, their counts in place
    const Avail av = unpack_avail(avm);
    border_from_tile(wave, 8, y0, x0, av.l, av.bl, av.a, av.ar);
    P1Args P;
    P.q = F.job.q; P.only_mode = -1; P.hint = 0; P.own = 0; P.c_lo = lo; P.c_hi = hi; P.shape = 0; P.tok = wave_tok(F.sc, 0);
    P.N = 8; P.y0 = y0; P.x0 = x0; P.k = 0; P.per_mode_border = 0; P.out_kind = OUT_NONE;
    P.rec8 = lds_ld_i32(&C.solo2n) != 0 ? rec8_of(0) : nullptr;      // (partner workgroup: every candidate keeps its reconstruction)
    p1_run_t<3>(wave, P);
    wave_sync();                                        // the tokens are in memory
    LANES(l) { if (l == 0) lds_st_i32(&C.lend_done[li], lds_ld_i32(&C.a_go)); }
}
// four-TU set of an 8x8 CU in a wide workgroup: a token segment (header + TU 0, then one per TU) is complete — note where it ends in every
// candidate's stream, start the next one on a token-block boundary (the gap holds idle tokens, which leave coder and contexts untouched),
// and tell the coder wavefront (partner_fourtu)
HD void seg_close(WaveMem &W, int k) {
    WideCtl &C = WCTL;
    LANES(l) {
        if (l < NMODE) {
            C.seg_end[k][l] = W.tokn[l];
            if (k < 3) { W.tokn[l] = (W.tokn[l] + 7) & ~7; blk_idle((u32a *)W.pend[l]); }
        }
    }
    wave_sync();                                        // the segment is in memory
    LANES(l) { if (l == 0) lds_st_i32(&C.b_seg, lds_ld_i32(&C.b_seg) + 1); }
}
// ... whose 35 trial coders run on wave 5 WHILE wave 1 still works on the later TUs: segment by segment on one byte sink each (as the pipe
// wave codes its stream, stream_seg), contexts in this wavefront's own scratch (wave 1's pass buffer is busy), copied over at the end.
HDN void partner_fourtu(int depth_) {
    const int depth = uni_i(depth_);
    PartnerMem &X = XM(1);
    WideCtl &C = WCTL;
    WaveMem &W = WM(1);
    const RdW rw = rd_weights(F.job.q);
    u8 *const ubytes = uniform_ptr(F.sc.bytes);
    const u16 *tok = wave_tok(F.sc, 1);
    const int base = lds_ld_i32(&C.b_cons);
    LANES(l) {
        const int on = l < NMODE, ll = on ? l : 0;
        Arith a = SM.entry_a[depth];
        const int len0 = arith_len(a);
        u8 *cx = X.cx[ll]; LaneMem *lm = &X.lm[ll];
        u8 *gbuf = ubytes + (size_t)(1 * NMODE + ll) * TRIAL_BYTES;
        const u16 *ts = tok + (size_t)ll * TOK_CAP;
        if (on) ctx_copy(cx, SM.entry_cx[depth]);
        const Arith a0 = a;
        LeadSink sink; lsink_begin(sink, a0, lm->ring, gbuf);
        int from = 0, qn = 0;
        for (int k = 0; k < 3; k++) {
            while (lds_ld_i32(&C.b_seg) - base <= k) pipe_pause();
            wave_sync();
            const int end = C.seg_end[k][ll];
            stream_seg(a, cx, sink, qn, ts + from, on ? end - from : 0);
            from = (end + 7) & ~7;
        }
        {   // the last segment: wave 1 is through with its passes by the time it is complete and takes the range half (fourtu_last_range_half); the byte half stays here
            SplitQ &q = X.q;
            if (on) q.range_out[l] = a.range;
            wave_sync_lds();
            if (l == 0) lds_st_i32(&C.b_hand, lds_ld_i32(&C.cu8));
            while (lds_ld_i32(&q.go) == lds_ld_i32(&q.done)) pipe_pause();
            wave_sync();
            const int end = C.seg_end[3][ll];
            int blk = 0;
            stream_seg_L(a, sink, qn, q, l, blk, on ? end - from : 0);
            split_await(&q.rdone, q);
            if (on) a.range = q.range_out[l];
        }
        trial_finish(a, a0, sink, qn, on);
        if (on) {
            W.fin[l] = pack_arith(a);
            W.cost[l] = rd_cost(rw, W.sse[l], arith_len(a) - len0);
            ctx_copy(W.u.p2.cx[l], cx);                 // (wave 1 finished its last pass before it released the last segment: its pass buffer is free)
        }
    }
    split_flag(&X.q.done, X.q);
    LANES(l) { if (l == 0) lds_st_i32(&C.b_cons, base + 4); }
}
// wave 1, its four passes done: the range half of the four-TU set's last segment, from the range wave 4's coders reached and on their contexts
HDN void fourtu_last_range_half() {
    PartnerMem &X = XM(1); SplitQ &q = X.q;
    WideCtl &C = WCTL;
    const u16 *tok = wave_tok(F.sc, 1);
    while (lds_ld_i32(&C.b_hand) != lds_ld_i32(&C.cu8)) pipe_pause();
    while (lds_ld_i32(&C.b_seg) - lds_ld_i32(&C.b_cons) < 4) pipe_pause();      // (partner workgroups: the last segment is closed by the token wavefront, fourtu_tokens_solo)
    wave_sync();
    split_start(q);
    LANES(l) {
        const int on = l < NMODE, ll = on ? l : 0;
        int range = on ? q.range_out[ll] : 510, blk = 0;
        const int from = (C.seg_end[2][ll] + 7) & ~7, end = C.seg_end[3][ll];
        stream_seg_R<false>(range, X.cx[ll], q, l, blk, tok + (size_t)ll * TOK_CAP + from, on ? end - from : 0);
        if (on) q.range_out[l] = range;
    }
    split_flag(&q.rdone, q);
}
#ifndef SPL32_0
#define SPL32_0 23
#define SPL32_1 24
#define SPL16_0 24
#define SPL16_1 32
#endif
HD int split_mode(int N, int shape) { return N == 32 ? (shape == 0 ? SPL32_0 : SPL32_1) : (shape == 0 ? SPL16_0 : SPL16_1); }   // passes come out balanced over the three waves
HDN_EVAL void eval_2Nx2N(int wave_, int depth_, int N_, int y0_, int x0_, int avm_) {
    const int wave = uni_i(wave_); const int depth = uni_i(depth_); const int N = uni_i(N_); const int y0 = uni_i(y0_); const int x0 = uni_i(x0_); const int avm = uni_i(avm_);
    const int lender = F.wide && N >= 16 && (wave == WAVE_PIPE_PARTNER || wave == WAVE_PU_PARTNER);      // wide workgroups, 16x16 / 32x32 CUs: waves 6 and 7 take pipeline passes too
    if (wave >= NWAVES && !lender) {                    // the pipe wave has no share in these sets: it only keeps the workgroup's barrier count
        if (N >= 16) {
            wg_sync_p(); wg_sync_p();
            if (F.wide && (wave == WAVE_A_PARTNER || wave == WAVE_B_CODER)) partner_trial(wave == WAVE_A_PARTNER ? 0 : 1, depth);      // wide workgroups: the byte half of the trial coders of waves 0 / 1 (on the other one's SIMD)
        }
        return;
    }
    const Avail av = unpack_avail(avm);
    WaveMem &W = WM(wave);
    const int q = F.job.q, h = N / 2, big = N >= 16;
    u8 *const ubytes = uniform_ptr(F.sc.bytes);
    const int uy = y0 >> 2, ux = x0 >> 2;
    const int big_l = N > nb_size(uy, ux - 1), big_a = N > nb_size(uy - 1, ux);
    const int ml = nb_mode(uy, ux - 1), ma = nb_mode(uy - 1, ux);
    if (wave < 2) {
        u16 *tok = wave_tok(F.sc, wave);
        LANES(l) {
            if (l < NMODE) {                            // every candidate's stream opens with its coding_unit header
                W.sse[l] = 0; W.tokn[l] = 0;
                blk_idle((u32a *)W.pend[l]);
                CuHdr J;
                J.N = N; J.shape = wave; J.ctx_split = (N >= 16) ? CX_SPLIT_CU + big_l + big_a : -1;
                J.mode[0] = l; J.ml[0] = ml; J.ma[0] = ma;
                LaneStream ls;
                TokW w = ls_begin(ls, W, l, lane_row(W, l), tok + (size_t)l * TOK_CAP);
                tk_cu_header(w, J);
                ls_end(ls, w, W, l);
            }
        }
    }
    const int wide8 = F.wide && !big;                   // 8x8 CU of a wide workgroup: two partner wavefronts take passes of the one-TU set, the four-TU set is coded segment by segment by wave 5
    const int solo = wide8 && lds_ld_i32(&WCTL.solo2n) != 0;      // ... of a PARTNER workgroup: no NxN chain beside the sets — TU 0 is made here, a second lender takes candidates 32 .. 34
    if (big) wg_sync_p(); else if (wide8 && wave == 0) { wave_sync(); LANES(l) { if (l == 0) lds_st_i32(&WCTL.a_go, lds_ld_i32(&WCTL.a_go) + 1); } } else wave_sync_lds();         // wave 2 / the lenders start from the owners' header counts
    P1Item it[2]; int nit = 1;
    if (!big) { it[0].own = wave; it[0].shape = wave; it[0].lo = 0; it[0].hi = (wide8 && wave == 0) ? 16 : NMODE;
                if (wide8 && wave == 0 && !solo) { nit = 2; it[1].own = 0; it[1].shape = 0; it[1].lo = 32; it[1].hi = NMODE; } }      // (candidates 16..31: wave 5, lend_passes)
    else if (F.wide) {                                  // five wavefronts: the one-TU set on waves 0 and 7, the four-TU set's mode chains on waves 1, 2 and 6
        const int a0 = N == 32 ? 18 : 20, b0 = N == 32 ? 12 : 16, b1 = N == 32 ? 24 : 32;
        const int sh = (wave == 0 || wave == WAVE_PU_PARTNER) ? 0 : 1;
        it[0].own = sh; it[0].shape = sh;
        it[0].lo = wave == 0 ? 0 : wave == WAVE_PU_PARTNER ? a0 : wave == 1 ? 0 : wave == 2 ? b0 : b1;
        it[0].hi = wave == 0 ? a0 : wave == WAVE_PU_PARTNER ? NMODE : wave == 1 ? b0 : wave == 2 ? b1 : NMODE;
    }
    else if (wave < 2) { it[0].own = wave; it[0].shape = wave; it[0].lo = 0; it[0].hi = split_mode(N, wave); }
    else { nit = 2; for (int i = 0; i < 2; i++) { it[i].own = i; it[i].shape = i; it[i].lo = split_mode(N, i); it[i].hi = NMODE; } }
    P1Args P;
    P.q = q; P.only_mode = -1; P.hint = 0; P.rec8 = nullptr;
    long long pt = prof_now();
    for (int ii = 0; ii < nit; ii++) {
        const int shape = it[ii].shape;
        P.own = it[ii].own; P.c_lo = it[ii].lo; P.c_hi = it[ii].hi; P.shape = shape; P.tok = wave_tok(F.sc, it[ii].own);
        P.rec8 = solo ? rec8_of(shape) : nullptr;
        if (solo && shape == 1) { fourtu_passes_solo(y0, x0, avm); continue; }      // partner workgroup: the four TU passes without their tokens (a second wavefront makes them: fourtu_tokens_solo)
        const int ntu = (shape == 0) ? 1 : 4;
        int k_first = 0;
        if (TU0_SHARE && N == 8 && shape == 1 && !solo) { tu0_from_pu0(wave, P.tok); k_first = 1; if (wide8) seg_close(W, 0); }      // TU 0 = the PU wave's PU 0
        for (int k = k_first; k < ntu; k++) {
            if (shape == 0) {
                border_from_tile(wave, N, y0, x0, av.l, av.bl, av.a, av.ar);
                P.N = N; P.y0 = y0; P.x0 = x0; P.k = 0; P.per_mode_border = 0; P.out_kind = OUT_NONE;
            } else {
                const Avail ca = child_avail(av, k);
                const int yk = y0 + (k >> 1) * h, xk = x0 + (k & 1) * h;
                if (k == 0) border_from_tile(wave, h, yk, xk, ca.l, ca.bl, ca.a, ca.ar);
                else border_tu_split(N, y0, x0, k, av.l, av.bl, av.a, av.ar, P.c_lo, P.c_hi);
                P.N = h; P.y0 = yk; P.x0 = xk; P.k = k; P.per_mode_border = (k != 0); P.out_kind = OUT_T3SIDE;
            }
            p1_run(wave, P);
            if (wide8 && shape == 1) seg_close(W, k);
            if (wide8) tl_mark(shape == 1 ? 36 + k : 40 + ii);       // 36 .. 39: the four-TU set's TU k passed; 40, 41: wave 0's passes
        }
    }
    if (wide8 && wave == 0) {                           // the lenders' candidates are done: their tokens are in memory, counts and SSE in this wavefront's arrays
        while (lds_ld_i32(&WCTL.lend_done[0]) != lds_ld_i32(&WCTL.a_go)) pipe_pause();
        if (solo) while (lds_ld_i32(&WCTL.lend_done[1]) != lds_ld_i32(&WCTL.a_go)) pipe_pause();
        wave_sync();
    }
    if (wide8 && wave == 1) { prof_add(PF_P1_4, pt); fourtu_last_range_half(); return; }      // (the trial coders of this set have been running on wave 5 all along: partner_fourtu)
    prof_add(wave == 2 ? PF_P1_4 : wave == 0 ? (N == 32 ? PF_P1_32 : N == 16 ? PF_P1_16 : PF_P1_8) : (N == 32 ? PF_P1_16 : N == 16 ? PF_P1_8 : PF_P1_4), pt);
    if (big) wg_sync_p(); else wave_sync();             // the tokens are in memory
    if (wave >= 2) return;
    // trial coders: lane m prices mode m from the CU's entry state
    u16 *tok = wave_tok(F.sc, wave);
    const RdW rw = rd_weights(q);
    pt = prof_now();
#ifndef IMCVT_HOSTEMU
    if (big) { if (F.prio_base) SETPRIO(3); else SETPRIO(1); }   // the third wave of the workgroup waits for these two: they are its critical path (1024 frames in flight: +4 %)
#endif
    if (wide8) tl_mark(42);                             // 42: the one-TU set's tokens complete (lenders included)
    if (F.wide) coder_range_half(wave, depth);          // wide workgroup: this wavefront runs the range half of its 35 coders, a partner wavefront the byte half (partner_trial)
    else
    LANES(l) {
        const int on = l < NMODE, ll = on ? l : 0;
        Arith a = SM.entry_a[depth];
        const int len0 = arith_len(a);
        run_trial(a, SM.entry_cx[depth], W.u.p2.cx[ll], &W.u.p2.lm[ll], ubytes + (size_t)(wave * NMODE + ll) * TRIAL_BYTES, tok + (size_t)ll * TOK_CAP, W.tokn[ll], on);
        if (on) {
            W.fin[l] = pack_arith(a);
            W.cost[l] = rd_cost(rw, W.sse[l], arith_len(a) - len0);
#ifdef IMCVT_TOKSTAT
            fprintf(stderr, "TS %d %d %d %d %d %d %d %d\n", N, wave, l, W.tokn[l], W.cost[l], W.sse[l], arith_len(a) - len0, (N > 8) ? SM.split_cost[depth] : -1);
#endif
        }
    }
    wave_sync();
#ifndef IMCVT_HOSTEMU
    if (big) { if (F.prio_base) SETPRIO(2); else SETPRIO(0); }
#endif

    prof_add(N == 32 ? PF_P2_32 : N == 16 ? PF_P2_16 : PF_P2_8, pt);
}

// ---- the NxN chain of an 8x8 CU (:1490-1543); wave-uniform call ---------------------------------------------------
// Token layout of a PU candidate (slot c): [7] cbf_luma, [8..] last position + group (the part PU pricing codes, :1515).
// Slot NMODE holds the NxN stream ([0..) header, then the four winners' tokens) and, from NXN_KEEP on, the winners' copies.
#define NXN_KEEP 1024
// One PU step of the NxN chain in a wide workgroup, on the PU wave: pass and pricing of the 35 candidates of PU k (hevc_core.h "A PU step ...").
//   here      prediction, DST, RDOQ -> levels published; first part of the tokens into the lane rows; range half of the pricing over them from LDS
//             (the remaining-level rows are bypass chunks: the byte half prices them alone); PU 0 only: the complete streams to memory (the four-TU wave's TU 0)
//   wave 7    remaining-level rows (pu_part_b)          pipe wave (wave 7 for PU 3)    reconstructions and SSE (pu_recon_k; PU 3: this wave itself), byte half of the pricing, costs (pu_price)
// Same tokens in the same order as p1_run_4 writes, same coder arithmetic as run_trial_r.
HDN_EVAL void pu_step_wide(int wave_, int yk_, int xk_, int k_) {      // (out of line: inlined, its registers would be eval_NxN's — and the 192- / 256-thread shapes' — to pay for)
    const int wave = uni_i(wave_); const int k = uni_i(k_);
    P1Args P;
    P.q = F.job.q; P.only_mode = -1; P.shape = 3; P.tok = wave_tok(F.sc, wave); P.N = 4; P.y0 = uni_i(yk_); P.x0 = uni_i(xk_); P.k = 0; P.per_mode_border = 0; P.out_kind = OUT_REC4;
    P.own = wave; P.c_lo = 0; P.c_hi = NMODE; P.hint = 1; P.rec8 = nullptr;
    WaveMem &W = WM(wave);
    const Tables &T = SM.T;
    PuX &U = PUX; SplitQ &q = XM(2).q;
    const u32 seq = pu_seq_of(k);
    const int share0 = TU0_SHARE && k == 0 && lds_ld_i32(&WCTL.remote8) == 0;      // PU 0's streams are the four-TU wave's TU 0 — unless that wave runs on another compute unit
    LANES(l) { split_reset(q, l); }
    LANES(l) {
        const int c = l, live = c < NMODE;
        int x[4][4];
        int any = 0;
        long long t4 = prof_now();
        MARKB("a4");
        if (live) any = pu_stage1(W, P, c, x);
        wave_sync_lds();
        if (l == 0) { lds_st_i32((i32 *)&U.pu_seq, (i32)seq); lds_st_i32(&q.go, lds_ld_i32(&q.go) + 1); }      // levels published (waves 7 and 6 start), queue counters zeroed
        tl_mark(16 + k);                                 // 16 .. 19: PU k's levels published
        MARKQ("a4_stage1", 74);
        prof_add(PF_T_HDR, t4); t4 = prof_now();        // (IMCVT_PROF builds: t_hdr = predict + DST + RDOQ, passA = first part of the tokens, passB = range half over it, passC = waiting for the remaining-level rows, n_cg = range half over them)
        const int st = scan_type_of(4, live ? c : 0);
        Lv16 L; u32 nzm = 0, mcode = 0;
        u16 *const row = lane_row(W, live ? l : 0);
        TokW w; w.n = 7; w.wr = 1; w.o.tb = row; w.o.pos = 0; w.o.cap = LCAP; w.o.glob = 0;      // (a PU candidate's stream starts at token 7, cbf_luma: the part that is priced, from token 8 on, starts a token block)
        if (live) {
            if (any) nzm = scan_levels(L, x, st, 0, &mcode);
            tk_bin(w, CX_CBF_LUMA + (P.shape == 0 ? 1 : 0), nzm != 0);
            TgB B;
            if (nzm != 0) {
                const int in = T.incg[st][hibit(nzm)];
                const LastPos lp = last_pos_prep(0, st, in >> 2, in & 3);
                w.n = last_pos_emit<0, true, true>(w.o, w.n, lp);
                w.n = tokg_a_fast<0, true>(w.o.tb, w.n, L, nzm, mcode, TG_DC | TG_LAST | st << TG_ST, B) & 0xFFFF;
            } else w.n = last_pos_emit<0, true, true>(w.o, w.n, last_pos_prep(0, st, 0, 0));      // PU pricing codes the residual syntax of an all-zero block (:1515)
            for (int i = 0; i < 8; i++) to_put(w.o, w.n + i, (int)TOK_IDLE);     // (at most 8 + 34 tokens so far)
            U.na[c] = w.n - 8;
        }
        const int na = live ? w.n - 8 : 0;
        wave_sync_lds();
        if (l == 0) lds_st_i32(&q.mid, lds_ld_i32(&q.go));                     // the counts are in place: the byte half may start
        tl_mark(24 + k);                                 // 24 .. 27: first part of PU k's tokens made
        MARKQ("a4_partA", 75);
        prof_add(PF_T_GEN, t4); t4 = prof_now();
        int range = 510, blk = 0;
        stream_seg_R_lds(range, q, l, blk, row + 8, na);
        if (live) q.range_out[l] = range;
        split_flag(&q.rdone, q);                                                 // (the remaining-level rows are bypass chunks only: the byte half goes on alone)
        tl_mark(20 + k);                                 // 20 .. 23: range half of PU k through
        prof_add(PF_T_DRAIN, t4); t4 = prof_now();
        while ((u32)lds_ld_i32((const i32 *)&U.b_seq) != seq) pipe_pause();      // (every lane waits here, outside lane-divergent code)
        wave_sync_lds();
        prof_add(PF_T_NDRAIN, t4); t4 = prof_now();
        const int nb = live ? U.bcnt[c] : 0;
        prof_add(PF_T_NTOK, t4);
        if (live) { W.tokn[c] = 8 + na + nb; W.tnz[c] = (u8)(nzm != 0); }
        if (share0 && live) {                           // PU 0: the four-TU wave takes TU 0 from these streams (tu0_from_pu0) — to memory, the remaining-level part behind the first, idle tokens up to the block boundary (the rows stay as they are)
            u16 *const g = P.tok + (size_t)c * TOK_CAP;
            g_st16((i16 *)(g + 7), (int)row[7]);
            row_to_stream(row + 8, g, 8, na);
            row_to_stream(U.brow[c], g, 8 + na, nb);
            const int end = 8 + na + nb;
            for (int i = 0; i < 7; i++) if (((end + i) >> 3) == (end >> 3) && (end & 7) != 0) g_st16((i16 *)(g + end + i), (int)TOK_IDLE);
        }
    }
    long long t5 = prof_now();
    if (share0) {
        wave_sync();                                    // the streams are in memory
        while ((u32)lds_ld_i32((const i32 *)&U.r_seq) != seq) pipe_pause();      // ... and SSE / reconstructions in this wave's slice (long since)
        wave_sync_lds();
        LANES(l) { if (l == 0) lds_st_i32(&SM.pu0_ready, 1); }
    }
    prof_add(PF_X1, t5); t5 = prof_now();
    if (k == 3) pu_recon(2, P, seq);                    // the last PU: nobody else has the time (who does what: above partner_pu)
    split_await(&q.done, q);                            // the costs
    tl_mark(12 + k);                                    // 12 .. 15: PU k priced
    prof_add(PF_X2, t5);
}
HDN_EVAL void eval_NxN(int wave_, int y0_, int x0_, int avm_) {
    const int wave = uni_i(wave_); const int y0 = uni_i(y0_); const int x0 = uni_i(x0_); const int avm = uni_i(avm_);
#ifndef IMCVT_HOSTEMU
    // the NxN chain is the longest of an 8x8 CU's three candidate sets: its wave wins the VALU arbitration of the SIMD it shares
    // with waves of other workgroups (1024 frames in flight: +3 %)
    if (F.prio_base) SETPRIO(3); else SETPRIO(NXN_PRIO_SOLO);
#endif
    const Avail av = unpack_avail(avm);
    WaveMem &W = WM(wave);
    const int q = F.job.q;
    const int pipe = F.pipe;                            // a pipe wave prices the NxN CU (nxn_pipe below): this wave only walks the PU chain
    const int hint = PU_HINTS && pipe;                  // latency-bound launches: PU candidates are priced on resolved tokens (fewer instructions on the PU chain, more in its passes)
    u16 *tok = wave_tok(F.sc, wave);
    u16 *nxn = tok + (size_t)NMODE * TOK_CAP;
    u8 *const ubytes = uniform_ptr(F.sc.bytes);
    const RdW rw = rd_weights(q);
    for (int k = 0; k < 4; k++) {
        const Avail ca = child_avail(av, k);
        const int yk = y0 + (k >> 1) * 4, xk = x0 + (k & 1) * 4;
        if (TU0_SHARE && k == 1 && !(F.wide && lds_ld_i32(&WCTL.remote8) != 0)) {      // the four-TU wave has its copy of PU 0's pass (long ago: it takes it while this wave prices PU 0)
            while (lds_ld_i32(&SM.pu0_taken) == 0) pipe_pause();
            wave_sync_lds();
            LANES(l) { if (l == 0) { lds_st_i32(&SM.pu0_ready, 0); lds_st_i32(&SM.pu0_taken, 0); } }
        }
        LANES(l) { if (l < NMODE) { W.sse[l] = 0; W.tokn[l] = 7; blk_idle((u32a *)W.pend[l]); } }
        wave_sync_lds();
        long long pt = prof_now();
        if (F.wide && k == 1) tl_mark(56);              // 56: PU 1's step begins
        if (F.wide) border4_from_tile(W, yk, xk, ca.l, ca.bl, ca.a, ca.ar); else border_from_tile(wave, 4, yk, xk, ca.l, ca.bl, ca.a, ca.ar);
        if (F.wide && k == 1) tl_mark(57);              // 57: its borders made
        prof_add(PF_P1_32, pt); pt = prof_now();            // (NxN chain, IMCVT_PROF builds: p1_32 = borders, p1_16 = store drain before pricing, p1_8 = pick + keep,
        P1Args P;                                           //  p2_32 = NxN header + stream assembly, p2_16 = the NxN trial itself)
        P.q = q; P.only_mode = -1; P.shape = 3; P.tok = tok; P.N = 4; P.y0 = yk; P.x0 = xk; P.k = 0; P.per_mode_border = 0; P.out_kind = OUT_REC4;
        P.own = wave; P.c_lo = 0; P.c_hi = NMODE; P.hint = hint; P.rec8 = nullptr;
        if (F.wide) {                                   // wide workgroup: pass and pricing shared with waves 7 and 6
            pu_step_wide(wave, yk, xk, k);
            prof_add(PF_P1_4, pt); pt = prof_now();
        } else {
        p1_run(wave, P);
        prof_add(PF_P1_4, pt); pt = prof_now();
        wave_sync();
        if (TU0_SHARE && k == 0) { LANES(l) { if (l == 0) lds_st_i32(&SM.pu0_ready, 1); } }      // tokens in memory, SSE / reconstructions in this wave's slice: the four-TU wave's TU 0
        prof_add(PF_P1_16, pt); pt = prof_now();
        LANES(l) {                                      // residual bits on a fresh coder and fresh contexts (:1504-1518)
            const int on = l < NMODE, ll = on ? l : 0;
            Arith a; arith_reset(a);
            u8 *gb = ubytes + (size_t)(wave * NMODE + ll) * TRIAL_BYTES; const u16 *ts = tok + (size_t)ll * TOK_CAP + 8;
            if (hint) run_trial_r(a, SM.cx0, W.u.p2.cx[ll], &W.u.p2.lm[ll], gb, ts, W.tokn[ll] - 8, on);      // (tokens with state hints: p1_run_4)
            else run_trial(a, SM.cx0, W.u.p2.cx[ll], &W.u.p2.lm[ll], gb, ts, W.tokn[ll] - 8, on);
            if (on) W.cost[l] = rd_cost(rw, W.sse[l], arith_len(a));
#ifdef IMCVT_TOKSTAT
            if (on) fprintf(stderr, "TS %d %d %d %d %d %d %d %d\n", 4, wave, l, W.tokn[l] - 8, W.cost[l], W.sse[l], arith_len(a), -1);
#endif
        }
        }
        wave_sync_lds();
        prof_add(PF_P2_PU, pt); pt = prof_now();
        LANES(l) {                                      // pick the PU mode: later mode wins ties (:1520)
            int mn;
            const int bm = wave_last_min(l < NMODE ? W.cost[l] : 0, l < NMODE, l, &mn);
            if (l == 0) {
                W.pu_mode[k] = bm; W.pu_sse[k] = W.sse[bm];
                W.pu_cnt[k] = W.tnz[bm] ? W.tokn[bm] - 7 : 1;                            // an all-zero PU is cbf_luma = 0 inside the CU
            }
        }
        wave_sync_lds();
        if (F.wide && k == 1) tl_mark(61);              // 61: PU 1's mode picked
        LANES(l) {                                      // keep its tokens and put its reconstruction in place (:1523-1524)
            const int bm = W.pu_mode[k], cnt = W.pu_cnt[k];
            const u16 *src = tok + (size_t)bm * TOK_CAP + 7;
            // the pipe wave codes the winners of PUs 0..2 as ONE stream segment (kept back to back) and PU 3's as another: idle tokens up to
            // the block boundary after PU 2 and after PU 3
            const int at = !pipe ? k * NXN_KEEP_STRIDE : k == 3 ? 3 * NXN_KEEP_STRIDE : (k >= 1 ? W.pu_cnt[0] : 0) + (k >= 2 ? W.pu_cnt[1] : 0);
            const int end = at + cnt;
            if (F.wide) {                               // wide workgroup: the winner's tokens lie in its lane row (cbf_luma + first part) and in wave 7's row (remaining levels) and are kept in LDS (PuX::kept): the pipe wave's coders read them there
                u16 *dst = PUX.kept + at;
                const int na1 = 1 + PUX.na[bm];
                const u16 *ra = lane_row(W, bm) + 7, *rb = PUX.brow[bm];
                for (int i = l; i < cnt; i += 64) dst[i] = i < na1 ? ra[i] : rb[i - na1];
                if (k >= 2 && l < 8 && ((end + l) >> 3) == (end >> 3) && (end & 7) != 0) PUX.kept[end + l] = (u16)TOK_IDLE;
            } else {
            u16 *dst = nxn + NXN_KEEP + at;
            for (int i = l; i < cnt; i += 64) g_st16((i16 *)(dst + i), g_ld16((const i16 *)(src + i)));
            if (pipe && k >= 2 && l < 8 && ((end + l) >> 3) == (end >> 3) && (end & 7) != 0) g_st16((i16 *)(nxn + NXN_KEEP + end + l), (int)TOK_IDLE);
            }
            if (l < 16) SM.rec[yk + (l >> 2) + 1][xk + (l & 3) + 1] = W.u.w2.rec4[bm][l];
        }
        wave_sync_lds();
        if (pipe && k >= 2) {                           // PUs 0..2 are enough for the pipe wave to start (it tries all 35 modes of PU 3 in the header), PU 3 lets it finish
            if (!F.wide) wave_sync();                   // the kept tokens are in memory (wide workgroups keep them in LDS)
            LANES(l) { if (l == 0) lds_st_i32(k == 2 ? &SM.pipe_a : &SM.pipe_b, 1); }
        }
        if (F.wide) tl_mark(52 + k);                     // 52 .. 55: PU k decided and kept
        prof_add(PF_P1_8, pt);
    }
    if (pipe) {
#ifndef IMCVT_HOSTEMU
        if (F.prio_base) SETPRIO(2); else SETPRIO(0);
#endif
        return;
    }
    // price the whole NxN CU from the entry state (:1530-1543)
    const int uy = y0 >> 2, ux = x0 >> 2;
    const long long ptn = prof_now();
    LANES(l) {
        if (l == 0) {
            CuHdr J;
            J.N = 8; J.shape = 2; J.ctx_split = -1;
            for (int k = 0; k < 4; k++) J.mode[k] = W.pu_mode[k];
            J.ml[0] = nb_mode(uy, ux - 1);     J.ma[0] = nb_mode(uy - 1, ux);
            J.ml[1] = J.mode[0];                  J.ma[1] = nb_mode(uy - 1, ux + 1);
            J.ml[2] = nb_mode(uy + 1, ux - 1); J.ma[2] = J.mode[0];
            J.ml[3] = J.mode[2];                  J.ma[3] = J.mode[1];
            W.tokn[NMODE] = 0;
            blk_idle((u32a *)W.pend[NMODE]);
            LaneStream ls;
            TokW w = ls_begin(ls, W, NMODE, lane_row(W, 0), nxn);
            tk_cu_header(w, J);
            ls_end(ls, w, W, NMODE);
        }
    }
    wave_sync();                                        // header and kept tokens are in memory
    LANES(l) {
        int pos = W.tokn[NMODE];
        for (int k = 0; k < 4; k++) {
            const int cnt = W.pu_cnt[k];
            const u16 *src = nxn + NXN_KEEP + k * NXN_KEEP_STRIDE;
            for (int i = l; i < cnt; i += 64) g_st16((i16 *)(nxn + pos + i), g_ld16((const i16 *)(src + i)));
            pos += cnt;
        }
        if (l < 8 && ((pos + l) >> 3) == (pos >> 3) && (pos & 7) != 0) g_st16((i16 *)(nxn + pos + l), (int)TOK_IDLE);   // idle tokens up to the block boundary
        wave_sync();                                    // the stream is in memory
        prof_add(PF_P2_32, ptn);
        const long long ptt = prof_now();
#if NXN_UNI
        // the trial on wave-uniform values (stream_run_uni): the scalar unit walks the stream, lane 0 stores
        if (l < CTX_STRIDE / 4) *(u32a *)(W.u.p2.cx[0] + 4 * l) = *(const u32a *)(SM.entry_cx[2] + 4 * l);
        wave_sync_lds();
        if (UNI_RUN(l)) {
            Arith a;
            a.range = UNI(SM.entry_a[2].range); a.low = UNI(SM.entry_a[2].low); a.nbits = UNI(SM.entry_a[2].nbits); a.nbytes = UNI(SM.entry_a[2].nbytes);
            a.bufbyte = UNI(SM.entry_a[2].bufbyte); a.zeros = UNI(SM.entry_a[2].zeros); a.cnt = UNI(SM.entry_a[2].cnt);
            const int len0 = arith_len(a);
            stream_run_uni(a, W.u.p2.cx[0], ubytes + (size_t)(wave * NMODE) * TRIAL_BYTES, nxn, UNI(pos), l);
            if (l == 0) {
                W.fin[0] = pack_arith(a);
                W.nxn_cost = rd_cost(rw, W.pu_sse[0] + W.pu_sse[1] + W.pu_sse[2] + W.pu_sse[3], arith_len(a) - len0);
            }
        }
#else
        const int on = l == 0;
        Arith a = SM.entry_a[2];
        const int len0 = arith_len(a);
        run_trial(a, SM.entry_cx[2], W.u.p2.cx[0], &W.u.p2.lm[0], ubytes + (size_t)(wave * NMODE) * TRIAL_BYTES, nxn, pos, on);
        if (on) {
            W.fin[0] = pack_arith(a);
            W.nxn_cost = rd_cost(rw, W.pu_sse[0] + W.pu_sse[1] + W.pu_sse[2] + W.pu_sse[3], arith_len(a) - len0);
        }
#endif
        prof_add(PF_P2_16, ptt);
    }
    wave_sync();
    prof_add(PF_P2_NXN, ptn);
#ifndef IMCVT_HOSTEMU
    if (F.prio_base) SETPRIO(2); else SETPRIO(0);
#endif
}

#define PIPE_HDR_OFF 2048     // the 35 header streams: token PIPE_HDR_OFF.. of the PU wave's candidate slots (a PU candidate uses < 200)
// the pipe wave's coders in a wide workgroup: range half of the 35 speculative NxN streams (header, winners of PUs 0..2, PU 3's winner on the lane that guessed its mode)
// the context stage of the pipe wave's streams (a main workgroup whose 2Nx2N sets are with its partner: wave 1 has nothing else to do) — hevc_core.h CtxQ
#define CX8_STRIDE (CTX_STRIDE + 1)        // entries per lane (odd: the lanes' copies start in different banks)
HD uint2 *cx8_of(int lane) { return (uint2 *)wave_mem_ptr(WAVE_A_PARTNER) + lane * CX8_STRIDE; }      // the lenders' three slices are idle while a CU's 2Nx2N sets are with the partner
static_assert((size_t)NMODE * CX8_STRIDE * 8 <= NLEND * sizeof(WaveMem), "the context copies of table entries fit the lenders' slices");
HDN void pipe_ctx_stage() {
    WaveMem &W = PM;
    const WaveMem &W2 = WM(2);
    const u16 *tok2 = wave_tok(F.sc, 2);
    CtxQ &cq = CTXQ;
    // the copies first — a table ENTRY per context (block_C8e) from the CU's entry states, long before anybody needs them (this wavefront has nothing else to do until PU 2 is decided)
    LANES(l) {
        if (l < NMODE) {
            uint2 *c8 = cx8_of(l);
            for (int i = 0; i < CTX_STRIDE; i++) c8[i] = SM.T.pst[SM.entry_cx[2][i] & 127];
        }
    }
    wave_sync_lds();
    while (lds_ld_i32(&cq.go) == lds_ld_i32(&cq.done)) pipe_pause();      // the headers are made (and in memory), the rings' counters zeroed
    wave_sync();
    LANES(l) {
        const int on = l < NMODE, ll = on ? l : 0;
        const int nh = W.tokn[ll];
        u8 *cx = (u8 *)cx8_of(ll);
        const u16 *hdr = tok2 + (size_t)ll * TOK_CAP + PIPE_HDR_OFF;
        int cblk = 0;
        stream_seg_C<false, true>(cx, cq, l, cblk, hdr, on ? nh : 0);
        const int n012 = W2.pu_cnt[0] + W2.pu_cnt[1] + W2.pu_cnt[2];
        stream_seg_C<true, true>(cx, cq, l, cblk, PUX.kept, on ? n012 : 0);
        split_await(&XM(PIPE_WAVE).q.mid, XM(PIPE_WAVE).q);      // PU 3 is decided (the range wavefront passes flag B on as this generation's `mid`, as to the byte half)
        const int mine = on & (l == W2.pu_mode[3]);
        stream_seg_C<true, true>(cx, cq, l, cblk, PUX.kept + 3 * NXN_KEEP_STRIDE, mine ? W2.pu_cnt[3] : 0);
    }
    wave_sync_lds();
    LANES(l) {                                              // the states of the lane that guessed PU 3's mode, where the decision looks for the NxN trial's contexts
        const uint2 *c8 = cx8_of(W2.pu_mode[3]);
        u8 *dst = W.u.p2.cx[W2.pu_mode[3]];
        for (int i = l; i < CTX_STRIDE; i += 64) dst[i] = (u8)((c8[i].y >> 16) & 127u);
    }
    wave_sync_lds();
    LANES(l) { if (l == 0) lds_st_i32(&cq.done, lds_ld_i32(&cq.go)); }
}
HDN void nxn_pipe_wide() {
    WaveMem &W = PM;
    const WaveMem &W2 = WM(2);
    u16 *tok2 = wave_tok(F.sc, 2);
        SplitQ &q = XM(PIPE_WAVE).q;
        const int three = lds_ld_i32(&WCTL.remote8) != 0;      // the 2Nx2N sets are with the partner workgroup: a third wavefront runs the context side of these coders (pipe_ctx_stage)
        split_start(q);
        if (three) {
            CtxQ &cq = CTXQ;
            LANES(l) { if (l < NMODE) { cq.prod[l] = 0; cq.cons[l] = 0; } }
            wave_sync_lds();
            LANES(l) { if (l == 0) lds_st_i32(&cq.go, lds_ld_i32(&cq.go) + 1); }
        }
        LANES(l) {
            const int on = l < NMODE, ll = on ? l : 0;
            const int nh = W.tokn[ll];
            u8 *cx = W.u.p2.cx[ll];
            const u16 *hdr = tok2 + (size_t)ll * TOK_CAP + PIPE_HDR_OFF;
            int range = SM.entry_a[2].range, blk = 0, cblk = 0;
            const int n012 = W2.pu_cnt[0] + W2.pu_cnt[1] + W2.pu_cnt[2];
            if (three) stream_seg_Rq(range, CTXQ, q, l, cblk, blk, on ? nh : 0);
            else { if (on) ctx_copy(cx, SM.entry_cx[2]); stream_seg_R<false>(range, cx, q, l, blk, hdr, on ? nh : 0); }
            if (three) tl_mark(37);                             // 37: range half through the headers
            if (three) stream_seg_Rq(range, CTXQ, q, l, cblk, blk, on ? n012 : 0);
            else stream_seg_R_ldsrc(range, cx, q, l, blk, PUX.kept, on ? n012 : 0);
            if (three) tl_mark(38);                             // 38: ... through the winners of PUs 0 .. 2
            while (lds_ld_i32(&SM.pipe_b) == 0) pipe_pause();
            wave_sync_lds();
            if (l == 0) lds_st_i32(&q.mid, lds_ld_i32(&q.go));      // PU 3 is decided: the partner may go on too
            const int mine = on & (l == W2.pu_mode[3]);
            if (three) stream_seg_Rq(range, CTXQ, q, l, cblk, blk, mine ? W2.pu_cnt[3] : 0);
            else stream_seg_R_ldsrc(range, cx, q, l, blk, PUX.kept + 3 * NXN_KEEP_STRIDE, mine ? W2.pu_cnt[3] : 0);
            if (mine) { q.range_out[l] = range; SM.nxn_lane = l; }
            if (l == 0) { lds_st_i32(&SM.pipe_a, 0); lds_st_i32(&SM.pipe_b, 0); }
        }
        split_flag(&q.rdone, q);
#ifndef IMCVT_HOSTEMU
        if (F.prio_base) SETPRIO(2); else SETPRIO(0);
#endif
}
// ---- the NxN trial of an 8x8 CU on the pipe wave (256-thread launches) -------------------------------------------------
// The NxN stream is header, then the four PU winners' residuals (:1530-1543), and the header names all four PU modes — so the
// trial cannot start before PU 3 is decided, and on the PU wave it is a serial tail of ~190 tokens coded by one lane.  Here
// lane m of the pipe wave assumes PU 3 = mode m: as soon as PUs 0..2 are decided, 35 lanes code the 35 possible headers and
// the winners of PUs 0..2 from the CU's entry state, while the PU wave is busy with PU 3.  When PU 3 is decided the lane that
// guessed its mode codes PU 3's winner and holds the trial's result; what is left of the tail is that one segment.
// The stream is coded in three segments (header; PUs 0..2, kept back to back; PU 3), each padded to a token block with idle tokens,
// which leave the coder untouched: same bins in the same order as :1530-1543.
HDN_EVAL void nxn_pipe(int y0_, int x0_) {
    const int y0 = uni_i(y0_); const int x0 = uni_i(x0_);
#ifndef IMCVT_HOSTEMU
    if (F.prio_base) SETPRIO(3); else SETPRIO(NXN_PRIO_SOLO);
#endif
    WaveMem &W = PM;
    const WaveMem &W2 = WM(2);
    u16 *tok2 = wave_tok(F.sc, 2);
    const u16 *kept = tok2 + (size_t)NMODE * TOK_CAP + NXN_KEEP;
    u8 *const ubytes = uniform_ptr(F.sc.bytes);
    const RdW rw = rd_weights(F.job.q);
    const int uy = y0 >> 2, ux = x0 >> 2;
    while (lds_ld_i32(&SM.pipe_a) == 0) pipe_pause();
    wave_sync();
    LANES(l) {
        if (l < NMODE) {
            CuHdr J;
            J.N = 8; J.shape = 2; J.ctx_split = -1;
            for (int k = 0; k < 3; k++) J.mode[k] = W2.pu_mode[k];
            J.mode[3] = l;
            J.ml[0] = nb_mode(uy, ux - 1);     J.ma[0] = nb_mode(uy - 1, ux);
            J.ml[1] = J.mode[0];               J.ma[1] = nb_mode(uy - 1, ux + 1);
            J.ml[2] = nb_mode(uy + 1, ux - 1); J.ma[2] = J.mode[0];
            J.ml[3] = J.mode[2];               J.ma[3] = J.mode[1];
            W.tokn[l] = 0;
            blk_idle((u32a *)W.pend[l]);
            LaneStream ls;
            TokW w = ls_begin(ls, W, l, lane_row(W, l), tok2 + (size_t)l * TOK_CAP + PIPE_HDR_OFF);
            tk_cu_header(w, J);
            ls_end(ls, w, W, l);
        }
    }
    wave_sync();                                        // the headers are in memory; the token rows make room for the coders' contexts
    if (F.wide && lds_ld_i32(&WCTL.remote8) != 0) tl_mark(36);      // (timeline builds, CUs whose 2Nx2N sets are out: 36 the pipe wave's 35 headers made)
    if (F.wide) { nxn_pipe_wide(); return; }           // wide workgroup: the range half of the 35 streams here, the byte half on the partner wavefront (partner_pipe)
    LANES(l) {
        const int on = l < NMODE, ll = on ? l : 0;
        const int nh = W.tokn[ll];
        u8 *cx = W.u.p2.cx[ll]; LaneMem *lm = &W.u.p2.lm[ll];
        u8 *gbuf = ubytes + (size_t)(PIPE_WAVE * NMODE + ll) * TRIAL_BYTES;
        const u16 *hdr = tok2 + (size_t)ll * TOK_CAP + PIPE_HDR_OFF;
        Arith a = SM.entry_a[2];
        const int len0 = arith_len(a);
        if (on) for (int i = 0; i < CTX_STRIDE; i += 4) *(u32a *)(cx + i) = *(const u32a *)(SM.entry_cx[2] + i);
        const Arith a0 = a;
        LeadSink sink; lsink_begin(sink, a0, lm->ring, gbuf);
        int qn = 0;
        stream_seg(a, cx, sink, qn, hdr, on ? nh : 0);
        const int n012 = W2.pu_cnt[0] + W2.pu_cnt[1] + W2.pu_cnt[2];
        stream_seg(a, cx, sink, qn, kept, on ? n012 : 0);
        while (lds_ld_i32(&SM.pipe_b) == 0) pipe_pause();
        wave_sync();
        const int mine = on & (l == W2.pu_mode[3]);
        stream_seg(a, cx, sink, qn, kept + 3 * NXN_KEEP_STRIDE, mine ? W2.pu_cnt[3] : 0);
        trial_finish(a, a0, sink, qn, mine);
        if (mine) {
            W.fin[0] = pack_arith(a);
            WM(2).nxn_cost = rd_cost(rw, W2.pu_sse[0] + W2.pu_sse[1] + W2.pu_sse[2] + W2.pu_sse[3], arith_len(a) - len0);
            SM.nxn_lane = l;
        }
        if (l == 0) { lds_st_i32(&SM.pipe_a, 0); lds_st_i32(&SM.pipe_b, 0); }      // (the PU wave sets them again only after the workgroup barrier that ends this CU)
    }
    wave_sync();
#ifndef IMCVT_HOSTEMU
    if (F.prio_base) SETPRIO(2); else SETPRIO(0);
#endif
}

// partner of the pipe wave: the byte half of the 35 speculative NxN streams; the lane that guessed PU 3's mode holds the trial's result
HDN void partner_pipe() {
    PartnerMem &X = XM(PIPE_WAVE); SplitQ &q = X.q;
    WaveMem &W = PM;
    const WaveMem &W2 = WM(2);
    const u16 *tok2 = wave_tok(F.sc, 2);
    const u16 *kept = tok2 + (size_t)NMODE * TOK_CAP + NXN_KEEP;
    u8 *const ubytes = uniform_ptr(F.sc.bytes);
    const RdW rw = rd_weights(F.job.q);
    while (lds_ld_i32(&q.go) == lds_ld_i32(&q.done)) pipe_pause();
    wave_sync();
    LANES(l) {
        const int on = l < NMODE, ll = on ? l : 0;
        const int nh = W.tokn[ll];
        u8 *gbuf = ubytes + (size_t)(PIPE_WAVE * NMODE + ll) * TRIAL_BYTES;
        const u16 *hdr = tok2 + (size_t)ll * TOK_CAP + PIPE_HDR_OFF;
        Arith a = SM.entry_a[2];
        const int len0 = arith_len(a);
        const Arith a0 = a;
        LeadSink sink; lsink_begin(sink, a0, X.lm[ll].ring, gbuf);
        int blk = 0, qn = 0;
        stream_seg_L(a, sink, qn, q, l, blk, on ? nh : 0);
        if (lds_ld_i32(&WCTL.remote8) != 0) tl_mark(40);      // 40: byte half through the headers
        const int n012 = W2.pu_cnt[0] + W2.pu_cnt[1] + W2.pu_cnt[2];
        stream_seg_L(a, sink, qn, q, l, blk, on ? n012 : 0);
        if (lds_ld_i32(&WCTL.remote8) != 0) tl_mark(41);      // 41: ... through the winners of PUs 0 .. 2
        split_await(&q.mid, q);                         // PU 3 is decided (its winner's tokens are in memory)
        const int mine = on & (l == W2.pu_mode[3]);
        stream_seg_L(a, sink, qn, q, l, blk, mine ? W2.pu_cnt[3] : 0);
        if (lds_ld_i32(&WCTL.remote8) != 0) tl_mark(42);      // 42: ... through PU 3's winner
        trial_finish(a, a0, sink, qn, mine);
        split_await(&q.rdone, q);
        if (mine) {
            a.range = q.range_out[l];
            W.fin[0] = pack_arith(a);
            WM(2).nxn_cost = rd_cost(rw, W2.pu_sse[0] + W2.pu_sse[1] + W2.pu_sse[2] + W2.pu_sse[3], arith_len(a) - len0);
        }
    }
    split_flag(&q.done, q);
}

// ---- the winner's reconstruction (only the winner's is ever needed, so candidates do not store theirs): the winning
// 2Nx2N shape is run once more, writing the tile.  All waves call this; wave 0 works.
HDN void rebuild_winner(int kind_, int mode_, int N_, int y0_, int x0_, int avm_) {
    const int kind = uni_i(kind_); const int mode = uni_i(mode_); const int N = uni_i(N_); const int y0 = uni_i(y0_); const int x0 = uni_i(x0_); const int avm = uni_i(avm_);
    const Avail av = unpack_avail(avm);
    WAVES(w) {
        if (w == 0) {
            const int wave = 0;
            P1Args P;
            P.q = F.job.q; P.only_mode = mode; P.shape = 0; P.k = 0; P.per_mode_border = 0; P.out_kind = OUT_TILE; P.tok = (u16 *)0;
            P.own = 0; P.c_lo = 0; P.c_hi = 1; P.rec8 = nullptr;
            if (kind == 1) {
                border_from_tile(wave, N, y0, x0, av.l, av.bl, av.a, av.ar);
                P.N = N; P.y0 = y0; P.x0 = x0;
                p1_run_cold(wave, P);
            } else {
                const int h = N / 2;
                for (int k = 0; k < 4; k++) {
                    const Avail ca = child_avail(av, k);
                    P.N = h; P.y0 = y0 + (k >> 1) * h; P.x0 = x0 + (k & 1) * h;
                    border_from_tile(wave, h, P.y0, P.x0, ca.l, ca.bl, ca.a, ca.ar);
                    p1_run_cold(wave, P);
                }
            }
        }
    }
}

// ---- one CU after its children (if any) are done: evaluate the unsplit shapes, decide, commit ---------------------
// All waves call this with identical arguments.
HDN void decide_cu(int depth_, int N_, int y0_, int x0_, int avm_) {
    const int depth = uni_i(depth_); const int N = uni_i(N_); const int y0 = uni_i(y0_); const int x0 = uni_i(x0_); const int avm = uni_i(avm_);
    u8 *live_sink = F.job.out + F.out_pos;
    const int tl = N == 8 && F.wide;                     // (-DIMCVT_PROF_TL builds: timeline of a wide workgroup's 8x8 CUs)
    if (tl) tl_start();
    WAVES_ALL(w_) {
        // 8x8 CU of a wide workgroup: physical wavefront w_ runs ROLE w (roles are named by the wavefront that runs them under the identity: memory, flags and queues go
        // by role).  Which two roles share a SIMD — wavefronts w and w + 4 do — is the placement; ROLE8_PERM permutes it (A/B builds, profiles/r06i_role_perm_ab.log).
        const int w = (F.wide && N < 16) ? (int)(((u32)ROLE8_PERM >> (4 * w_)) & 7u) : w_;
        if (w >= NWAVES && N < 16) {
            if (w == PIPE_WAVE) { if (F.wide) partner_pu_early(y0, x0); nxn_pipe(y0, x0); }      // (wide workgroups: until PU 2 is decided the pipe wave has nothing of its own to do — reconstructions and byte half of the pricing of PUs 0..2, on a SIMD the PU wave does not run on)
            else if (!F.wide) { }
            else if (w == WAVE_B_CODER) partner_fourtu(depth);                                                   // partner wavefronts of a wide workgroup (hevc_core.h, "who is whose partner"): 4 the trial coders of the four-TU set, segment by segment behind wave 1's passes
            else if (w == WAVE_A_PARTNER) { lend_passes(w, 0, 16, 32, y0, x0, avm); partner_trial(0, depth); }   // 5 a pass of the one-TU set, then the byte half of its trial coders
            else if (w == WAVE_PIPE_PARTNER) partner_pipe();                                                     // 6 the byte half of the pipe wave's streams (the PU wave's SIMD-mate: idle while the first three PUs are walked)
            else partner_pu(y0, x0);                                                                             // 7 the PU chain's partner: remaining-level tokens, reconstructions, byte half of the pricing
        }
        else if (w != 2 || N >= 16) eval_2Nx2N(w, depth, N, y0, x0, avm);
        else eval_NxN(2, y0, x0, avm);
        if (tl) tl_mark(1 + w_);                         // 1 .. 8: wave w_ is through with its part of the candidate sets
    }
    wg_sync_p();
    if (tl) WAVES(w) { if (w == 0) tl_mark(9); }         // 9: every wave is
    WAVES(w) LANES(l) {
        // split cost first, then modes 0..34 with one TU, 0..34 with four TUs, then NxN, each accepted with `best >= cost`: the last
        // minimum wins.  Wave 0 scans: the minimum of the 70, a four-TU candidate that holds it beats every one-TU candidate.
        int m1 = 0, m2 = 0, l1 = 0, l2 = 0;
        if (w == 0) {
            l1 = wave_last_min(l < NMODE ? WM(0).cost[l] : 0, l < NMODE, l, &m1);
            l2 = wave_last_min(l < NMODE ? WM(1).cost[l] : 0, l < NMODE, l, &m2);
        }
        if (w == 0 && l == 0) {
            int best = (N > 8) ? SM.split_cost[depth] : I32MAX, kind = 0, mode = 0;
            if (best >= m1) { best = m1; kind = 1; mode = l1; }
            if (best >= m2) { best = m2; kind = 2; mode = l2; }
            if (N == 8 && best >= WM(2).nxn_cost) { best = WM(2).nxn_cost; kind = 3; }
            SM.win_kind = kind; SM.win_mode = mode;
            if (F.sc.trace && F.trace_n + 8 <= F.sc.trace_cap) {
                i32 *t = F.sc.trace + F.trace_n;
                t[0] = F.ctu_y + y0; t[1] = F.ctu_x + x0; t[2] = N; t[3] = kind; t[4] = (kind == 3) ? (WM(2).pu_mode[0] | WM(2).pu_mode[1] << 8 | WM(2).pu_mode[2] << 16 | WM(2).pu_mode[3] << 24) : mode;
                t[5] = best; t[6] = m1; t[7] = m2;             // (the minima of the one-TU and the four-TU candidate sets)
                F.trace_n += 8;
                if (N == 8 && F.trace_n + 8 <= F.sc.trace_cap) {      // 8x8 CUs: a second row (size 4) for the NxN candidate
                    t += 8;
                    t[0] = F.ctu_y + y0; t[1] = F.ctu_x + x0; t[2] = 4; t[3] = 3; t[4] = WM(2).pu_mode[0] | WM(2).pu_mode[1] << 8 | WM(2).pu_mode[2] << 16 | WM(2).pu_mode[3] << 24;
                    t[5] = WM(2).nxn_cost; t[6] = WM(2).pu_sse[0] + WM(2).pu_sse[1]; t[7] = WM(2).pu_sse[2] + WM(2).pu_sse[3];
                    F.trace_n += 8;
                }
            }
        }
    }
    wg_sync_p();
    const int kind = SM.win_kind, mode = SM.win_mode;
    if (kind != 0) {
        const int pk = (kind == 3) && F.pipe;             // the NxN trial ran on the pipe wave: its result sits in that wave's slice, lane nxn_lane
        const int ww = pk ? PIPE_WAVE : (kind == 3) ? 2 : kind - 1, wl = pk ? SM.nxn_lane : (kind == 3) ? 0 : mode, fl = pk ? 0 : wl;
        const WaveMem &WW = pk ? PM : WM(ww);
        const u8 *src = lane_bytes(F.sc, ww, wl);
        const FinState fe = WW.fin[fl];                     // (the coder state the winner's trial ended in: read before the rebuild below reuses wave 0's buffers)
        const int as_bytes = NXN_UNI && kind == 3 && !pk;      // (the NxN trial on the scalar unit, stream_run_uni, leaves bytes; every other trial leaves the leads of its bytes)
        WAVES(w) LANES(l) {
            const int tid = w * 64 + l;
            if (tid < CTX_STRIDE) SM.cx[tid] = WW.u.p2.cx[wl][tid];
            if (as_bytes) {
                const int cnt0 = SM.entry_a[depth].cnt, cnt1 = (int)(WW.fin[fl].w2 >> 16);
                for (int i = tid; i < cnt1 - cnt0; i += WG_THREADS) g_st8(live_sink + cnt0 + i, g_ld8(src + i));
                if (tid == 64) SM.live = unpack_arith(WW.fin[fl]);
            }
            if (tid >= 128 && tid < 128 + 64) {             // neighbour maps (:1444-1445, :1549-1553)
                const int n = N >> 2, i = (tid - 128) >> 3, j = (tid - 128) & 7;
                if (i < n && j < n) {
                    const int uy = (y0 >> 2) + i, ux = (x0 >> 2) + j;
                    SM.mapsz[uy + 1][ux + 1] = (u8)N;
                    SM.mapmode[uy + 1][ux + 1] = (u8)((kind == 3) ? WM(2).pu_mode[i * 2 + j] : mode);
                }
            }
        }
        wg_sync_p();
        const long long ptr_ = prof_now();
        if (kind != 3) rebuild_winner(kind, mode, N, y0, x0, avm);     // the winner's reconstruction into the tile (wave 0)
        if (!as_bytes) {                                    // meanwhile wave 1: the winner's leads -> bytes, straight into the frame's stream, and the byte-level state they leave (hevc_core.h resolve_leads)
            WAVES(w) LANES(l) {
                if (w == 1) {
                    const Arith e = unpack_arith(fe);
                    Arith a = SM.entry_a[depth];
                    a.low = e.low; a.range = e.range; a.nbits = e.nbits;
                    resolve_leads(a, src, (int)g_ld32(src + TRIAL_BYTES - 4), live_sink);
                    if (l == 0) SM.live = a;
                }
            }
        }
        prof_add(PF_RECON, ptr_);
        wg_sync_p();
    }
    if (tl) WAVES(w) { if (w == 0) tl_mark(10); }        // 10: winner committed
}

// snapshot the live coder as the entry state of `depth`, optionally after coding split_cu_flag=1 (:1363-1364, :1403)
HDN void post_request(int depth, int N, int y0, int x0, int avm);
HDN void enter_cu(int depth_, int N_, int y0_, int x0_, int code_split_, int avm_) {
    const int depth = uni_i(depth_); const int N = uni_i(N_); const int y0 = uni_i(y0_); const int x0 = uni_i(x0_); const int code_split = uni_i(code_split_); const int avm = uni_i(avm_);
    u8 *live_sink = F.job.out + F.out_pos;
    WAVES(w) LANES(l) {
        const int tid = w * 64 + l;
        if (tid < CTX_STRIDE) SM.entry_cx[depth][tid] = SM.cx[tid];
        if (tid == 64) SM.entry_a[depth] = SM.live;
        if (tid == 65 && N == 8 && F.wide) {
            WideCtl &C = WCTL;
            C.cu8++;
            int r8 = 0;                                 // this CU's 2Nx2N sets go to the partner workgroup (decide_cu8_remote) — if there is one, and no answer of an abandoned request is still on its way
            if (F.mail && C.part_ok) {
                if (C.stale8 && (i32)m_ld32(&F.mail->s[SLOT_8].res_flag) == C.stale8) C.stale8 = 0;
                r8 = C.stale8 == 0;
            }
            C.remote8 = r8;
        }
    }
    wg_sync();
    if (F.mail && N >= 16) {     // pool: a helper starts on this CU's 70 unsplit candidates now — unless requests already wait unclaimed (every
        WAVES(w) LANES(l) {      // helper is busy): then this workgroup evaluates the CU itself when it comes back from the children
            if (w == 0 && l == 0) {
                const int slot = slot_of(N);
                const PoolShard *q = &F.pq->sh[F.main_id % POOL_SHARDS];
                // a fixed share of the CUs is offered (the launch shape's balance between the main workgroups' own work and what
                // the pool can take, hevc_hip.hip pool_split), evenly spread; an offered CU is still kept while requests wait unclaimed
                hb_beat();
                F.post_acc[slot] += F.post_pm[slot];
                const int offer = F.post_acc[slot] >= 1000;
                if (offer) F.post_acc[slot] -= 1000;
                if (F.stale[slot] && (i32)m_ld32(&F.mail->s[slot].res_flag) == F.stale[slot]) F.stale[slot] = 0;     // the answer this workgroup stopped waiting for has arrived: the mailbox is free again
                F.posted[depth] = offer && !F.stale[slot] && (i32)(m_ld32(&q->tail[slot]) - m_ld32(&q->head[slot])) < F.lim[slot] && m_ld32(&F.pq->alive) != 0u;
                F.kept += !F.posted[depth];
            }
        }
        wg_sync();
        if (F.posted[depth]) post_request(depth, N, y0, x0, avm);
    }
    if (code_split) {
        WAVES(w) LANES(l) {
            if (w == 0 && l == 0) {
                const int uy = y0 >> 2, ux = x0 >> 2;
                const int big_l = N > nb_size(uy, ux - 1), big_a = N > nb_size(uy - 1, ux);
                Arith a = SM.live;
                Sink ls; ls.base = live_sink; ls.off = 0;
                code_bin(a, SM.cx, ls, CX_SPLIT_CU + big_l + big_a, 1);
                SM.live = a;
            }
        }
        wg_sync();
    }
}

// cost of keeping the split (:1408-1409): SSE of the children's reconstruction + bits spent since entry
HDN void price_split(int depth_, int N_, int y0_, int x0_) {
    const int depth = uni_i(depth_); const int N = uni_i(N_); const int y0 = uni_i(y0_); const int x0 = uni_i(x0_);
    WAVES(w) LANES(l) { if (w == 0 && l == 0) SM.red[0] = 0; }
    wg_sync();
    WAVES(w) LANES(l) {
        const int tid = w * 64 + l;
        int part = 0;
        const int lgn = hibit((u32)N);
        NOUNROLL
        for (int i = tid; i < N * N; i += WG_THREADS) {
            const int y = y0 + (i >> lgn), x = x0 + (i & (N - 1));
            const int d = (int)SM.org[y][x] - SM.rec[y + 1][x + 1];
            part += d * d;
        }
        if (part) lds_add(&SM.red[0], part);
    }
    wg_sync();
    WAVES(w) LANES(l) {
        if (w == 0 && l == 0) {
            const RdW rw = rd_weights(F.job.q);
            SM.split_cost[depth] = rd_cost(rw, SM.red[0], arith_len(SM.live) - arith_len(SM.entry_a[depth]));
        }
    }
    wg_sync();
}


// =====================================================================================================================
// Teams: one frame, several workgroups.
// The 70 unsplit candidates of a 16x16 / 32x32 CU start from the coder state at the CU's entry and predict from samples
// outside the CU, so nothing in them depends on the CU's children (reference :1363-1364, :1419-1483).  In a team the main
// workgroup posts that entry state when it enters the CU, walks the children itself, and picks up the helper's answer —
// the last minimum among the 70, with its coder state, contexts, bytes and reconstruction — when it has priced the split
// (:1408-1409).  The decision is the reference's: split cost first, then the candidates in order, each accepted with
// `best >= cost`, i.e. the last minimum of the 70 wins iff it does not exceed the split cost.
// =====================================================================================================================
// All threads call these.  publish: every mail word this workgroup stored so far is visible to whoever then sees flag == v
// (`v` is read from thread 0 only).
HD void team_publish(i32 *flag, i32 v) {
    drain_stores();
    wg_sync();
    WAVES(w) LANES(l) { if (w == 0 && l == 0) m_st32(flag, (u32)v); }
}
// the same for a request: the mail words are visible to the helper that claims the ticket and finds this workgroup's index in it
// A ring entry names its ticket: (ticket mod 2^20) << 12 | main workgroup index + 1.  A claimer waits for ITS ticket's entry; an entry of a
// later lap (the claimer was held up for a whole lap of the ring while other workgroups kept posting and being served) tells it that
// its ticket is gone — the owner has long stopped waiting (ABANDON_TICKS) — and it goes back to polling instead of waiting for a
// value that will never come.
HD u32 ring_entry(u32 ticket, int main_id) { return (ticket & 0xFFFFFu) << 12 | ((u32)main_id + 1u); }
HD void pool_push(int slot) {
    drain_stores();
    wg_sync();
    WAVES(w) LANES(l) {
        if (w == 0 && l == 0) { PoolShard *q = &F.pq->sh[F.main_id % POOL_SHARDS]; const u32 t = m_add32(&q->tail[slot], 1u); m_st32(&q->ring[slot][t % POOL_QCAP], ring_entry(t, F.main_id)); }
    }
}
// await: returns once flag == v; mail loads issued afterwards see what the publisher stored before publishing
// Waits between workgroups carry a watchdog: a wait that lasts WD_TICKS (nothing legitimate comes near) records what it was
// waiting for, raises the launch's abort flag and gives up; every other wait sees the flag and gives up too, every workgroup leaves
// at its next CTU / request, and the host reports IMCVT_ERR_WATCHDOG instead of hanging.  `code`, a, b: for the record.
HD int wd_poll(PoolQ *pq, unsigned long long t0, int n, int code, int a, int b) {      // thread 0, every few polls; returns non-zero to give up
    if ((n & 15) != 15) return 0;
    if (m_ld32(&pq->abort) != 0u) return 1;
    if (wd_now() - t0 < WD_TICKS) return 0;
    if (m_cas32(&pq->abort, 0u, (u32)code)) { m_st32(&pq->dbg[0], (u32)code); m_st32(&pq->dbg[1], (u32)a); m_st32(&pq->dbg[2], (u32)b); m_st32(&pq->dbg[3], (u32)F.main_id); m_st32(&pq->dbg[4], (u32)F.frame);
        if (code == 1) { const PoolShard *q = &pq->sh[F.main_id % POOL_SHARDS]; m_st32(&pq->dbg[5], m_ld32(&q->head[a])); m_st32(&pq->dbg[6], m_ld32(&q->tail[a])); m_st32(&pq->dbg[7], m_ld32(&F.mail->s[a].req_flag));
            for (int i = 0; i < 4; i++) m_st32(&pq->pad_[i], m_ld32(&F.mail->s[a].pad0_[i])); m_st32(&pq->pad_[4], (u32)wd_now()); m_st32(&pq->pad_[5], (u32)t0); } }
    return 1;
}
HD void team_await(i32 *flag, i32 v, int slot) {
    WAVES(w) LANES(l) {
        if (w == 0 && l == 0) {
            const unsigned long long t0 = wd_now();
            for (int n = 0; (i32)m_ld32(flag) != v; n++) {
                if (wd_poll(F.pq, t0, n, 1, slot, v)) { F.aborted = 1; break; }
#ifdef IMCVT_HOSTEMU
                if (n >= ABANDON_POLLS) { F.gaveup = 1; break; }
#else
                if ((n & 3) == 3 && wd_now() - t0 > ABANDON_TICKS) { F.gaveup = 1; break; }
#endif
                mail_poll_pause();
            }
            const u32 dt = (u32)(wd_now() - t0); F.waited += dt; if (dt > F.waited_max) F.waited_max = dt;
            hb_set(wd_now());                   // (waiting is not a gap)
        }
    }
    wg_sync();
}
HD int ld_i(const i32 *p) { return (i32)m_ld32(p); }

// main: post the entry state of the CU at (y0,x0,N) — call right after enter_cu's snapshot, before the split flag is coded
HDN void post_request(int depth_, int N_, int y0_, int x0_, int avm_) {
    const int depth = uni_i(depth_); const int N = uni_i(N_); const int y0 = uni_i(y0_); const int x0 = uni_i(x0_); const int avm = uni_i(avm_);
    const int slot = slot_of(N);
    MailSlot *m = &F.mail->s[slot];
    const int uy = y0 >> 2, ux = x0 >> 2;
    WAVES(w) LANES(l) {
        const int tid = w * 64 + l;
        if (tid < CTX_STRIDE / 4) m_st32(m->req.ctx + 4 * tid, *(const u32a *)&SM.entry_cx[depth][4 * tid]);
        if (tid >= 32 && tid < 32 + 17) {                 // row above: tile row y0, columns x0 .. x0+2N (the tile row holds 65 samples)
            const int i = tid - 32;
            const u8 *r = &SM.rec[y0][imin(x0 + 4 * i, 64)];
            if (4 * i <= 2 * N) m_st32(m->req.above + 4 * i, (u32)r[0] | (u32)r[1] << 8 | (u32)r[2] << 16 | (u32)r[3] << 24);
        }
        if (tid >= 64 && tid < 64 + 16) {                 // column to the left: tile column x0, rows y0+1 .. y0+2N (rows beyond the tile are never available)
            const int i = tid - 64;
            if (4 * i < 2 * N) {
                u32 v = 0;
                for (int k = 0; k < 4; k++) v |= (u32)SM.rec[imin(y0 + 1 + 4 * i + k, 32)][x0] << (8 * k);
                m_st32(m->req.left + 4 * i, v);
            }
        }
        if (tid == 128) {
            i32 *r = (i32 *)&m->req;
            const Arith a = SM.entry_a[depth];
            const i32 v[20] = { OP_WORK, F.frame, F.ctu_y, F.ctu_x, N, y0, x0, avm, nb_size(uy, ux - 1), nb_size(uy - 1, ux), nb_mode(uy, ux - 1), nb_mode(uy - 1, ux),
                                a.range, a.low, a.nbits, a.nbytes, a.bufbyte, a.zeros, a.cnt, F.seq[slot] + 1 };
            for (int i = 0; i < 20; i++) m_st32(r + i, (u32)v[i]);
        }
    }
    wg_sync();
    WAVES(w) LANES(l) { if (w == 0 && l == 0) { F.seq[slot]++; m_st32(&m->req_flag, 0u); } }      // (req_flag: debug marker, set by the helper that takes the request)
    pool_push(slot);
}

// main: the CU's children are done and the split is priced — take the helper's answer and decide (:1439, :1475).  Returns non-zero
// when the answer did not come in time: the caller then evaluates the CU itself (same result), and the mailbox stays out of use
// until the late answer has arrived (enter_cu).
HDN int decide_remote(int depth_, int N_, int y0_, int x0_) {
    const int depth = uni_i(depth_); const int N = uni_i(N_); const int y0 = uni_i(y0_); const int x0 = uni_i(x0_);
    const int slot = slot_of(N);
    MailSlot *m = &F.mail->s[slot];
    u8 *live_sink = F.job.out + F.out_pos;
    const long long t0 = prof_now();
    team_await(&m->res_flag, F.seq[slot], slot);
    prof_add(PF_DECIDE, t0);                                // (booked as "wait_help": waiting for the helper's answer)
    if (F.aborted) return 0;                                // (watchdog: there is no answer to read)
    if (F.gaveup) {
        wg_sync();
        WAVES(w) LANES(l) { if (w == 0 && l == 0) { F.stale[slot] = F.seq[slot]; F.gaveup = 0; F.kept++; } }
        wg_sync();
        return 1;
    }
    const HelpRes *R = &m->res;
    WAVES(w) LANES(l) {
        if (w == 0 && l == 0) {
            const int cost = ld_i(&R->cost), take = SM.split_cost[depth] >= cost;
            SM.win_kind = take ? ld_i(&R->kind) : 0; SM.win_mode = ld_i(&R->mode);
            if (F.sc.trace && F.trace_n + 8 <= F.sc.trace_cap) {
                i32 *t = F.sc.trace + F.trace_n;
                t[0] = F.ctu_y + y0; t[1] = F.ctu_x + x0; t[2] = N; t[3] = SM.win_kind; t[4] = SM.win_mode;
                t[5] = take ? cost : SM.split_cost[depth]; t[6] = 0; t[7] = 0;
                F.trace_n += 8;
            }
        }
    }
    wg_sync();
    const int kind = SM.win_kind, mode = SM.win_mode;
    if (kind != 0) {
        const int cnt0 = SM.entry_a[depth].cnt, nbytes = ld_i(&R->nbytes);
        WAVES(w) LANES(l) {
            const int tid = w * 64 + l;
            for (int i = tid; i < (nbytes + 3) / 4; i += WG_THREADS) {      // (the stream buffer has slack beyond the CTU's bytes; whole dwords are copied)
                const u32 v = m_ld32(R->bytes + 4 * i);
                u8 *d = live_sink + cnt0 + 4 * i;
                g_st8(d, (int)(v & 255)); g_st8(d + 1, (int)((v >> 8) & 255)); g_st8(d + 2, (int)((v >> 16) & 255)); g_st8(d + 3, (int)(v >> 24));
            }
            if (tid < CTX_STRIDE / 4) *(u32a *)&SM.cx[4 * tid] = m_ld32(R->ctx + 4 * tid);
            if (tid == 64) { FinState f; f.w0 = m_ld32(&R->fin.w0); f.w1 = m_ld32(&R->fin.w1); f.w2 = m_ld32(&R->fin.w2); SM.live = unpack_arith(f); }
            if (tid >= 128 && tid < 128 + 64) {             // neighbour maps (:1444-1445)
                const int n = N >> 2, i = (tid - 128) >> 3, j = (tid - 128) & 7;
                if (i < n && j < n) {
                    const int uy = (y0 >> 2) + i, ux = (x0 >> 2) + j;
                    SM.mapsz[uy + 1][ux + 1] = (u8)N; SM.mapmode[uy + 1][ux + 1] = (u8)mode;
                }
            }
            for (int i = tid; i < N * N / 4; i += WG_THREADS) {      // its reconstruction replaces the children's (:1441, :1477)
                const int y = i / (N / 4), x4 = (i % (N / 4)) * 4;
                const u32 v = m_ld32(R->rec + y * N + x4);
                u8 *d = &SM.rec[y0 + y + 1][x0 + x4 + 1];
                d[0] = (u8)v; d[1] = (u8)(v >> 8); d[2] = (u8)(v >> 16); d[3] = (u8)(v >> 24);
            }
        }
        wg_sync();
    }
    return 0;
}

// helper: serve one request — stage what the candidate sets read, evaluate the 70 candidates, answer with the last minimum
HDN void serve_request(const ColdTables *gK_, const FrameJob *jobs_, MailSlot *m_) {
    const ColdTables *const gK = uni_p(gK_); const FrameJob *const jobs = uni_p(jobs_); MailSlot *const m = uni_p(m_);
    const HelpReq *Q = &m->req;
    const int seq = ld_i(&Q->seq);
    HelpRes *R = &m->res;
    const int frame = ld_i(&Q->frame), cy = ld_i(&Q->cy), cx = ld_i(&Q->cx);
    const int N = ld_i(&Q->N), y0 = ld_i(&Q->y0), x0 = ld_i(&Q->x0), avm = ld_i(&Q->avm);
    const int depth = (N == 32) ? 0 : 1;
    WAVES(w) LANES(l) {
        if (w == 0 && l == 0) { F.job = jobs[frame]; F.ctu_y = cy; F.ctu_x = cx; F.out_pos = 0; F.trace_n = 0; }
    }
    wg_sync();
    const FrameJob J = F.job;
    WAVES(w) LANES(l) {
        const int tid = w * 64 + l;
        NOUNROLL
        for (int i = tid; i < N * N; i += WG_THREADS) {     // source pixels of the CU (:1621)
            const int y = y0 + i / N, x = x0 + i % N;
            SM.org[y][x] = g_ld8(J.img + (size_t)clip3(cy + y, 0, J.h - 1) * J.w + clip3(cx + x, 0, J.w - 1));
        }
        if (tid < 17 && 4 * tid <= 2 * N) {                   // the samples it predicts from, placed where the main workgroup's tile has them
            const u32 v = m_ld32(Q->above + 4 * tid);
            u8 *d = &SM.rec[y0][imin(x0 + 4 * tid, 64)];
            d[0] = (u8)v; d[1] = (u8)(v >> 8); d[2] = (u8)(v >> 16); d[3] = (u8)(v >> 24);
        }
        if (tid >= 64 && tid < 64 + 16 && 4 * (tid - 64) < 2 * N) {
            const int i = tid - 64;
            const u32 v = m_ld32(Q->left + 4 * i);
            for (int k = 0; k < 4; k++) if (y0 + 1 + 4 * i + k <= 32) SM.rec[y0 + 1 + 4 * i + k][x0] = (u8)(v >> (8 * k));
        }
        if (tid >= 128 && tid < 128 + CTX_STRIDE / 4) *(u32a *)&SM.entry_cx[depth][4 * (tid - 128)] = m_ld32(Q->ctx + 4 * (tid - 128));
        if (tid >= 152 && tid < 152 + 4 * RQ_CLASSES) (&SM.rthr[0][0])[tid - 152] = (i32)g_ld32(&gK->rthr[J.q][0][0] + (tid - 152));
        if (tid == 32) {
            const i32 *r = (const i32 *)&Q->a;
            Arith a; a.range = ld_i(r); a.low = ld_i(r + 1); a.nbits = ld_i(r + 2); a.nbytes = ld_i(r + 3);
            a.bufbyte = ld_i(r + 4); a.zeros = ld_i(r + 5); a.cnt = ld_i(r + 6);
            SM.entry_a[depth] = a;
            const int uy = y0 >> 2, ux = x0 >> 2;                 // the two neighbour cells the CU header reads (:942-946, :957-976)
            SM.mapsz[uy + 1][ux] = (u8)ld_i(&Q->szl); SM.mapsz[uy][ux + 1] = (u8)ld_i(&Q->sza);
            SM.mapmode[uy + 1][ux] = (u8)ld_i(&Q->ml); SM.mapmode[uy][ux + 1] = (u8)ld_i(&Q->ma);
        }
    }
    wg_sync();
    WAVES(w) LANES(l) { if (w == 0 && l == 0) { m_st32(&m->pad0_[1], (u32)wd_now()); hb_beat(); } }      // (debug stamps: staged / evaluated / about to publish)
    WAVES_ALL(w) eval_2Nx2N(w, depth, N, y0, x0, avm);
    wg_sync();
    WAVES(w) LANES(l) { if (w == 0 && l == 0) { m_st32(&m->pad0_[2], (u32)wd_now()); hb_beat(); } }
    WAVES(w) LANES(l) {
        if (w == 0) {
            int m1, m2;
            const int l1 = wave_last_min(l < NMODE ? WM(0).cost[l] : 0, l < NMODE, l, &m1);
            const int l2 = wave_last_min(l < NMODE ? WM(1).cost[l] : 0, l < NMODE, l, &m2);
            if (l == 0) { const int four = m1 >= m2; SM.win_kind = four ? 2 : 1; SM.win_mode = four ? l2 : l1; SM.red[0] = four ? m2 : m1; }
        }
    }
    wg_sync();
    const int kind = SM.win_kind, mode = SM.win_mode;
    {   // the winner's contexts (the rebuild below reuses wave 0's pass buffer, where they live); then its reconstruction (wave 0) and, meanwhile, its
        // leads -> bytes (wave 1, hevc_core.h resolve_leads: into the buffer of a candidate that lost); then bytes, coder state and reconstruction to the mailbox
        const int ww = kind - 1;
        const u8 *src = lane_bytes(F.sc, ww, mode);
        u8 *tmp = lane_bytes(F.sc, ww ^ 1, 0);
        const int cnt0 = SM.entry_a[depth].cnt;
        WAVES(w) LANES(l) {
            const int tid = w * 64 + l;
            if (tid < CTX_STRIDE / 4) m_st32(R->ctx + 4 * tid, *(const u32a *)&WM(ww).u.p2.cx[mode][4 * tid]);
        }
        const FinState fe = WM(ww).fin[mode];
        wg_sync();
        rebuild_winner(kind, mode, N, y0, x0, avm);
        WAVES(w) LANES(l) {
            if (w == 1) {
                const Arith e = unpack_arith(fe);
                Arith a = SM.entry_a[depth];
                a.low = e.low; a.range = e.range; a.nbits = e.nbits;
                resolve_leads(a, src, (int)g_ld32(src + TRIAL_BYTES - 4), tmp - cnt0);
                if (l == 0) SM.live = a;
            }
        }
        wg_sync();
        const FinState fin = pack_arith(SM.live);
        const int nbytes = SM.live.cnt - cnt0;
        WAVES(w) LANES(l) {
            const int tid = w * 64 + l;
            for (int i = tid; i < (nbytes + 3) / 4; i += WG_THREADS) m_st32(R->bytes + 4 * i, g_ld32(tmp + 4 * i));
            if (tid == 64) {
                m_st32(&R->cost, (u32)SM.red[0]); m_st32(&R->kind, (u32)kind); m_st32(&R->mode, (u32)mode); m_st32(&R->nbytes, (u32)nbytes);
                m_st32(&R->fin.w0, fin.w0); m_st32(&R->fin.w1, fin.w1); m_st32(&R->fin.w2, fin.w2);
            }
            for (int i = tid; i < N * N / 4; i += WG_THREADS) {
                const int y = i / (N / 4), x4 = (i % (N / 4)) * 4;
                const u8 *sp = &SM.rec[y0 + y + 1][x0 + x4 + 1];
                m_st32(R->rec + y * N + x4, (u32)sp[0] | (u32)sp[1] << 8 | (u32)sp[2] << 16 | (u32)sp[3] << 24);
            }
        }
    }
    WAVES(w) LANES(l) { if (w == 0 && l == 0) { m_st32(&m->pad0_[3], (u32)wd_now()); hb_beat(); } }
    team_publish(&m->res_flag, seq);
}

// =====================================================================================================================
// 8x8 CUs with a partner workgroup (round 6; wide launches).  One frame alone is the chain of its 8x8 CUs, and an 8x8 CU of a wide workgroup is
// work-bound on its compute unit: eight wavefronts, two per SIMD, all busy in its second half.  The CU's two 2Nx2N candidate
// sets start from the coder state at the CU's entry and predict from samples outside it, exactly like those of a 16x16 / 32x32 CU — nothing in
// them depends on the NxN chain that runs beside them (reference :1419-1483 against :1490-1543).  So a main workgroup that has a PARTNER workgroup
// (kernel_main: partner i serves main workgroup i; its own mailbox slot, no queue) posts every 8x8 CU's entry state, walks the NxN chain alone —
// four wavefronts, a SIMD each — and takes the partner's answer (the last minimum of the 70, as a helper gives it) when the chain is through:
//     main       wave 0: request out, answer in (staged in LDS while the chain runs)      wave 2: PU chain      wave 3: pipe wave
//                wave 4: byte half of the pipe wave's streams                           wave 5: the PU chain's partner (rows, byte half of PU 3's pricing)
//     partner    wave 0 + wave 2 (candidates 16..31, byte half) + wave 7 (candidates 32..34): one-TU set      wave 1 + wave 3 (coders): four-TU set, TU 0 included
// The decision is the reference's: modes 0..34 one TU, 0..34 four TUs (the partner's last minimum), then NxN, each accepted with `best >= cost` (:1439, :1475, :1545).
// =====================================================================================================================
HD void stage_tables(const Tables *gT);
struct Ans8 { i32 cost, kind, mode, nbytes; FinState fin; i32 pad_; alignas(4) u8 ctx[CTX_STRIDE]; alignas(4) u8 rec[64]; alignas(4) u8 bytes[TRIAL_OUT_BYTES]; };
#define ANS8 (*(Ans8 *)WM(0).u.raw)      // wave 0's pass buffer: idle while the CU's 2Nx2N sets are out
static_assert(sizeof(Ans8) <= 7168, "the staged answer lives in a wave's pass buffer");
#ifndef PART_POLL_SLEEP
#define PART_POLL_SLEEP 4                // x 64 cycles between polls of the partner mailbox: one wavefront per compute unit polls, the answer sits on the frame's chain
#endif
#ifdef IMCVT_HOSTEMU
HD void part_poll_pause() { emu_yield(); }
static long g_remote8[3];                // (test builds: 8x8 CUs decided with the partner's answer, of those won by the partner's candidate, CUs whose answer was abandoned)
#define R8CNT(i) (g_remote8[i]++)
#else
#define R8CNT(i) ((void)0)
HD void part_poll_pause() { __builtin_amdgcn_s_sleep(PART_POLL_SLEEP); }
#endif
// main workgroup, wave 0 alone (the other wavefronts are walking the NxN chain): the request, then the answer into LDS
HDN void remote8_wave0(int y0_, int x0_, int avm_) {
    const int y0 = uni_i(y0_); const int x0 = uni_i(x0_); const int avm = uni_i(avm_);
    MailSlot *m = &F.mail->s[SLOT_8];
    WideCtl &C = WCTL;
    const int uy = y0 >> 2, ux = x0 >> 2;
    const int seq = lds_ld_i32(&C.seq8) + 1;
    LANES(l) {
        if (l < CTX_STRIDE / 4) m_st32(m->req.ctx + 4 * l, *(const u32a *)&SM.entry_cx[2][4 * l]);
        if (l >= 32 && l < 32 + 5) {                      // row above: tile row y0, columns x0 .. x0 + 16 (corner first)
            const int i = l - 32;
            const u8 *r = &SM.rec[y0][imin(x0 + 4 * i, 64)];
            m_st32(m->req.above + 4 * i, (u32)r[0] | (u32)r[1] << 8 | (u32)r[2] << 16 | (u32)r[3] << 24);
        }
        if (l >= 40 && l < 40 + 4) {                      // column to the left: tile column x0, rows y0 + 1 .. y0 + 16
            const int i = l - 40;
            u32 v = 0;
            for (int k = 0; k < 4; k++) v |= (u32)SM.rec[imin(y0 + 1 + 4 * i + k, 32)][x0] << (8 * k);
            m_st32(m->req.left + 4 * i, v);
        }
        if (l == 48) {
            i32 *r = (i32 *)&m->req;
            const Arith a = SM.entry_a[2];
            const i32 v[20] = { OP_WORK, F.frame, F.ctu_y, F.ctu_x, 8, y0, x0, avm, nb_size(uy, ux - 1), nb_size(uy - 1, ux), nb_mode(uy, ux - 1), nb_mode(uy - 1, ux),
                                a.range, a.low, a.nbits, a.nbytes, a.bufbyte, a.zeros, a.cnt, seq };
            for (int i = 0; i < 20; i++) m_st32(r + i, (u32)v[i]);
        }
    }
    drain_stores();
    wave_sync();
    LANES(l) { if (l == 0) { lds_st_i32(&C.seq8, seq); m_st32(&m->req_flag, (u32)seq); } }
    tl_mark(11);                                               // 11: the request is out
    // the answer: lane 0 polls, then the wavefront stages it
    LANES(l) {
        if (l == 0) {
            const unsigned long long t0 = wd_now();
            int got = seq;
            for (int n = 0; (i32)m_ld32(&m->res_flag) != seq; n++) {
                if (wd_poll(F.pq, t0, n, 3, SLOT_8, seq)) { F.aborted = 1; got = -1; break; }
#ifdef IMCVT_HOSTEMU
                if (n >= ABANDON_POLLS) { got = -1; break; }
#else
                if ((n & 15) == 15 && wd_now() - t0 > ABANDON_TICKS) { got = -1; break; }
#endif
                part_poll_pause();
            }
            lds_st_i32(&C.ans_seq, got);
        }
    }
    wave_sync_lds();
    tl_mark(67);                                               // 67: the answer's flag seen
    if (lds_ld_i32(&C.ans_seq) != seq) return;
    const HelpRes *R = &m->res;
    Ans8 &A8 = ANS8;
    LANES(l) {
        if (l == 0) { A8.cost = ld_i(&R->cost); A8.kind = ld_i(&R->kind); A8.mode = ld_i(&R->mode); A8.fin.w0 = m_ld32(&R->fin.w0); A8.fin.w1 = m_ld32(&R->fin.w1); A8.fin.w2 = m_ld32(&R->fin.w2); }
        const int nbytes = ld_i(&R->nbytes);
        if (l == 1) A8.nbytes = nbytes;
        if (l < CTX_STRIDE / 4) *(u32a *)&A8.ctx[4 * l] = m_ld32(R->ctx + 4 * l);
        if (l >= 32 && l < 48) *(u32a *)&A8.rec[4 * (l - 32)] = m_ld32(R->rec + 4 * (l - 32));
        for (int i = l; i < (nbytes + 3) / 4; i += 64) *(u32a *)&A8.bytes[4 * i] = m_ld32(R->bytes + 4 * i);
    }
    wave_sync_lds();
}
// All waves call this with identical arguments.  Returns non-zero when the partner's answer did not come in time: the caller evaluates the CU itself
// (decide_cu: same result — the PU chain starts again from the same samples), and nothing more is posted until the late answer has landed.
HDN int decide_cu8_remote(int y0_, int x0_, int avm_) {
    const int y0 = uni_i(y0_); const int x0 = uni_i(x0_); const int avm = uni_i(avm_);
    u8 *live_sink = F.job.out + F.out_pos;
    WideCtl &C = WCTL;
    tl_start();                                                // (-DIMCVT_PROF_TL builds: timeline of the CU, tools/prof_timeline.py)
    WAVES_ALL(w) {
        if (w == 0) remote8_wave0(y0, x0, avm);
        else if (w == 2) eval_NxN(2, y0, x0, avm);
        else if (w == PIPE_WAVE) { partner_pu_early(y0, x0); nxn_pipe(y0, x0); }
        else if (w == WAVE_B_CODER) partner_pipe();            // (wave 4: a SIMD of its own here — the one-TU set's wavefront that shares it elsewhere is idle)
        else if (w == WAVE_A_PARTNER) partner_pu(y0, x0);      // (wave 5: likewise)
        else if (w == 1) pipe_ctx_stage();                     // (wave 1: the context stage of the pipe wave's streams)
        if (w <= 5) tl_mark(1 + w);      // 1: the partner's answer is staged; 3 .. 6: the NxN chain's wavefronts are through
    }
    wg_sync_p();
    WAVES(w) { if (w == 0) tl_mark(9); }
    if (lds_ld_i32(&C.ans_seq) != lds_ld_i32(&C.seq8)) {       // gave up (or the watchdog fired): this mailbox is out of use until the late answer has arrived
        WAVES(w) LANES(l) { if (w == 0 && l == 0) { lds_st_i32(&C.stale8, lds_ld_i32(&C.seq8)); lds_st_i32(&C.remote8, 0); C.cu8++; F.kept++; R8CNT(2); } }      // (cu8: the PU steps of the second walk get sequence numbers of their own)
        wg_sync();
        return 1;
    }
    const Ans8 &A8 = ANS8;
    WAVES(w) LANES(l) {
        if (w == 0 && l == 0) {
            int best = A8.cost, kind = A8.kind;               // (I32MAX >= cost: the last minimum of the 70 is always taken first)
            const int mode = A8.mode;
            if (best >= WM(2).nxn_cost) { best = WM(2).nxn_cost; kind = 3; }
            SM.win_kind = kind; SM.win_mode = mode;
            R8CNT(0); if (kind != 3) R8CNT(1);
            if (F.sc.trace && F.trace_n + 16 <= F.sc.trace_cap) {
                i32 *t = F.sc.trace + F.trace_n;
                t[0] = F.ctu_y + y0; t[1] = F.ctu_x + x0; t[2] = 8; t[3] = kind; t[4] = (kind == 3) ? (WM(2).pu_mode[0] | WM(2).pu_mode[1] << 8 | WM(2).pu_mode[2] << 16 | WM(2).pu_mode[3] << 24) : mode;
                t[5] = best; t[6] = A8.kind == 1 ? A8.cost : 0; t[7] = A8.kind == 2 ? A8.cost : 0;      // (only the better of the two sets' minima travels)
                t += 8;
                t[0] = F.ctu_y + y0; t[1] = F.ctu_x + x0; t[2] = 4; t[3] = 3; t[4] = WM(2).pu_mode[0] | WM(2).pu_mode[1] << 8 | WM(2).pu_mode[2] << 16 | WM(2).pu_mode[3] << 24;
                t[5] = WM(2).nxn_cost; t[6] = WM(2).pu_sse[0] + WM(2).pu_sse[1]; t[7] = WM(2).pu_sse[2] + WM(2).pu_sse[3];
                F.trace_n += 16;
            }
        }
    }
    wg_sync_p();
    const int kind = SM.win_kind, mode = SM.win_mode;
    if (kind == 3) {                                          // the NxN trial's result sits in the pipe wave's slice, lane nxn_lane (as decide_cu)
        const int wl = SM.nxn_lane;
        const u8 *src = lane_bytes(F.sc, PIPE_WAVE, wl);
        const FinState fe = PM.fin[0];
        WAVES(w) LANES(l) {
            const int tid = w * 64 + l;
            if (tid < CTX_STRIDE) SM.cx[tid] = PM.u.p2.cx[wl][tid];
            if (tid >= 128 && tid < 128 + 4) {
                const int i = (tid - 128) >> 1, j = (tid - 128) & 1, uy = (y0 >> 2) + i, ux = (x0 >> 2) + j;
                SM.mapsz[uy + 1][ux + 1] = 8; SM.mapmode[uy + 1][ux + 1] = (u8)WM(2).pu_mode[i * 2 + j];
            }
        }
        wg_sync_p();
        WAVES(w) LANES(l) {
            if (w == 1) {
                const Arith e = unpack_arith(fe);
                Arith a = SM.entry_a[2];
                a.low = e.low; a.range = e.range; a.nbits = e.nbits;
                resolve_leads(a, src, (int)g_ld32(src + TRIAL_BYTES - 4), live_sink);
                if (l == 0) SM.live = a;
            }
        }
        wg_sync_p();
    } else {                                                  // the partner's candidate: bytes, contexts, coder state, maps, reconstruction (:1441-1445, :1477-1481)
        const int cnt0 = SM.entry_a[2].cnt, nbytes = A8.nbytes;
        WAVES(w) LANES(l) {
            const int tid = w * 64 + l;
            for (int i = tid; i < nbytes; i += WG_THREADS) g_st8(live_sink + cnt0 + i, (int)A8.bytes[i]);
            if (tid < CTX_STRIDE / 4) *(u32a *)&SM.cx[4 * tid] = *(const u32a *)&A8.ctx[4 * tid];
            if (tid == 64) SM.live = unpack_arith(A8.fin);
            if (tid >= 128 && tid < 128 + 4) {
                const int i = (tid - 128) >> 1, j = (tid - 128) & 1, uy = (y0 >> 2) + i, ux = (x0 >> 2) + j;
                SM.mapsz[uy + 1][ux + 1] = 8; SM.mapmode[uy + 1][ux + 1] = (u8)mode;
            }
            if (tid >= 96 && tid < 96 + 16) {
                const int i = tid - 96, y = i >> 1, x4 = (i & 1) * 4;
                const u32 v = *(const u32a *)&A8.rec[y * 8 + x4];
                u8 *d = &SM.rec[y0 + y + 1][x0 + x4 + 1];
                d[0] = (u8)v; d[1] = (u8)(v >> 8); d[2] = (u8)(v >> 16); d[3] = (u8)(v >> 24);
            }
        }
        wg_sync_p();
    }
    WAVES(w) { if (w == 0) tl_mark(10); }
    return 0;
}

// partner workgroup: one request — the two 2Nx2N sets of the 8x8 CU it describes, answered with the last minimum of the 70 (as serve_request, N = 8)
HDN void serve_request8(const ColdTables *gK_, const FrameJob *jobs_, MailSlot *m_) {
    const ColdTables *const gK = uni_p(gK_); const FrameJob *const jobs = uni_p(jobs_); MailSlot *const m = uni_p(m_);
    const HelpReq *Q = &m->req;
    const int seq = ld_i(&Q->seq);
    HelpRes *R = &m->res;
    const int frame = ld_i(&Q->frame), cy = ld_i(&Q->cy), cx = ld_i(&Q->cx);
    const int y0 = ld_i(&Q->y0), x0 = ld_i(&Q->x0), avm = ld_i(&Q->avm);
    const int depth = 2, N = 8;
    const int newframe = frame != F.frame;                      // (the job and the per-quantiser tables change with the frame, not with the CU)
    tl_start();                                                 // (timeline builds: the request has been seen)
    WAVES(w) LANES(l) {
        if (w == 0 && l == 0) { if (newframe) F.job = jobs[frame]; F.ctu_y = cy; F.ctu_x = cx; F.out_pos = 0; F.trace_n = 0; WCTL.cu8++; }
    }
    wg_sync();
    const FrameJob J = F.job;
    WAVES(w) LANES(l) {
        const int tid = w * 64 + l;
        if (tid < 64) {                                         // source pixels of the CU (:1621)
            const int y = y0 + (tid >> 3), x = x0 + (tid & 7);
            SM.org[y][x] = g_ld8(J.img + (size_t)clip3(cy + y, 0, J.h - 1) * J.w + clip3(cx + x, 0, J.w - 1));
        }
        if (tid >= 64 && tid < 64 + 5) {                        // the samples it predicts from, placed where the main workgroup's tile has them
            const int i = tid - 64;
            const u32 v = m_ld32(Q->above + 4 * i);
            u8 *d = &SM.rec[y0][imin(x0 + 4 * i, 64)];
            d[0] = (u8)v; d[1] = (u8)(v >> 8); d[2] = (u8)(v >> 16); d[3] = (u8)(v >> 24);
        }
        if (tid >= 72 && tid < 72 + 4) {
            const int i = tid - 72;
            const u32 v = m_ld32(Q->left + 4 * i);
            for (int k = 0; k < 4; k++) if (y0 + 1 + 4 * i + k <= 32) SM.rec[y0 + 1 + 4 * i + k][x0] = (u8)(v >> (8 * k));
        }
        if (tid >= 128 && tid < 128 + CTX_STRIDE / 4) *(u32a *)&SM.entry_cx[depth][4 * (tid - 128)] = m_ld32(Q->ctx + 4 * (tid - 128));
        if (newframe && tid >= 152 && tid < 152 + 4 * RQ_CLASSES) (&SM.rthr[0][0])[tid - 152] = (i32)g_ld32(&gK->rthr[J.q][0][0] + (tid - 152));
        if (tid == 32) {
            const i32 *r = (const i32 *)&Q->a;
            Arith a; a.range = ld_i(r); a.low = ld_i(r + 1); a.nbits = ld_i(r + 2); a.nbytes = ld_i(r + 3);
            a.bufbyte = ld_i(r + 4); a.zeros = ld_i(r + 5); a.cnt = ld_i(r + 6);
            SM.entry_a[depth] = a;
            const int uy = y0 >> 2, ux = x0 >> 2;                 // the two neighbour cells the CU header reads (:957-976)
            SM.mapsz[uy + 1][ux] = (u8)ld_i(&Q->szl); SM.mapsz[uy][ux + 1] = (u8)ld_i(&Q->sza);
            SM.mapmode[uy + 1][ux] = (u8)ld_i(&Q->ml); SM.mapmode[uy][ux + 1] = (u8)ld_i(&Q->ma);
            F.frame = frame;
        }
    }
    wg_sync();
    WAVES(w) { if (w == 0) tl_mark(11); }                       // 11: inputs staged
    WAVES_ALL(w) {
        if (w < 2) eval_2Nx2N(w, depth, N, y0, x0, avm);
        else if (w == 2) { lend_passes(2, 0, 16, 32, y0, x0, avm); partner_trial(0, depth); }      // candidates 16 .. 31 of the one-TU set, then the byte half of its coders (SIMD c)
        else if (w == PIPE_WAVE) partner_fourtu(depth);                                             // the four-TU set's coders, segment by segment behind its tokens (SIMD d)
        else if (w == WAVE_PIPE_PARTNER) fourtu_tokens_solo();                                      // the four-TU set's tokens, TU by TU behind wave 1's passes (wave 6)
        else if (w == WAVE_PU_PARTNER) lend_passes(w, 1, 32, NMODE, y0, x0, avm);                   // candidates 32 .. 34 (wave 7)
        if (w <= PIPE_WAVE || w == WAVE_PU_PARTNER) tl_mark(1 + w);      // 1 .. 4, 8: the wavefronts are through the candidate sets
    }
    wg_sync();
    WAVES(w) { if (w == 0) tl_mark(9); }
    WAVES(w) LANES(l) {
        if (w == 0) {
            int m1, m2;
            const int l1 = wave_last_min(l < NMODE ? WM(0).cost[l] : 0, l < NMODE, l, &m1);
            const int l2 = wave_last_min(l < NMODE ? WM(1).cost[l] : 0, l < NMODE, l, &m2);
            if (l == 0) { const int four = m1 >= m2; SM.win_kind = four ? 2 : 1; SM.win_mode = four ? l2 : l1; SM.red[0] = four ? m2 : m1; }
        }
    }
    wg_sync();
    const int kind = SM.win_kind, mode = SM.win_mode;
    {
        const int ww = kind - 1;
        const u8 *src = lane_bytes(F.sc, ww, mode);
        u8 *tmp = lane_bytes(F.sc, ww ^ 1, 0);
        const int cnt0 = SM.entry_a[depth].cnt;
        WAVES(w) LANES(l) {
            const int tid = w * 64 + l;
            if (tid < CTX_STRIDE / 4) m_st32(R->ctx + 4 * tid, *(const u32a *)&WM(ww).u.p2.cx[mode][4 * tid]);
        }
        const FinState fe = WM(ww).fin[mode];
        wg_sync();
        WAVES(w) { if (w == 0) tl_mark(12); }                   // 12: winner picked, its contexts out (its reconstruction was kept by its pass: rec8_of)
        WAVES(w) LANES(l) {
            if (w == 1) {
                const Arith e = unpack_arith(fe);
                Arith a = SM.entry_a[depth];
                a.low = e.low; a.range = e.range; a.nbits = e.nbits;
                resolve_leads(a, src, (int)g_ld32(src + TRIAL_BYTES - 4), tmp - cnt0);
                if (l == 0) SM.live = a;
            }
        }
        WAVES(w) { if (w == 1) tl_mark(14); }                   // 14: its leads are bytes
        wg_sync();
        const FinState fin = pack_arith(SM.live);
        const int nbytes = SM.live.cnt - cnt0;
        WAVES(w) LANES(l) {
            const int tid = w * 64 + l;
            for (int i = tid; i < (nbytes + 3) / 4; i += WG_THREADS) m_st32(R->bytes + 4 * i, g_ld32(tmp + 4 * i));
            if (tid == 64) {
                m_st32(&R->cost, (u32)SM.red[0]); m_st32(&R->kind, (u32)kind); m_st32(&R->mode, (u32)mode); m_st32(&R->nbytes, (u32)nbytes);
                m_st32(&R->fin.w0, fin.w0); m_st32(&R->fin.w1, fin.w1); m_st32(&R->fin.w2, fin.w2);
            }
            if (tid >= 96 && tid < 96 + 16) m_st32(R->rec + 4 * (tid - 96), *(const u32a *)(rec8_of(ww) + mode * 64 + 4 * (tid - 96)));      // the winner's reconstruction as its pass left it
        }
    }
    team_publish(&m->res_flag, seq);
    WAVES(w) { if (w == 0) tl_mark(10); }                       // 10: answered
}
// partner workgroup `pid`: serves main workgroup pid's 8x8 requests until every frame is finished
HDN void partner8_loop(const Tables *gT_, const ColdTables *gK_, const FrameJob *jobs_, const Scratch sc, TeamMail *mail_, PoolQ *pq_, int pid_, int njobs_) {
    const Tables *const gT = uni_p(gT_); const ColdTables *const gK = uni_p(gK_); const FrameJob *const jobs = uni_p(jobs_); TeamMail *const mail = uni_p(mail_); PoolQ *const pq = uni_p(pq_);
    const int pid = uni_i(pid_); const int njobs = uni_i(njobs_);
    MailSlot *m = &mail[pid].s[SLOT_8];
    stage_tables(gT);
#if defined(IMCVT_PROF) && !defined(IMCVT_HOSTEMU)
    if (threadIdx.x < WG_THREADS && (threadIdx.x & 63u) < PF_N) SM.prof[threadIdx.x >> 6][threadIdx.x & 63u] = 0;
#endif
    WAVES(w) LANES(l) { if (w == 0 && l == 0) { F.sc = sc; F.mail = (TeamMail *)0; F.pq = pq; F.frame = -1; F.aborted = 0; WCTL.solo2n = 1; m_st32(&m->pad0_[8], 1u); } }      // (pad0_[8]: "the partner has reported in")
    wg_sync();
    int served = 0;
    for (;;) {
        WAVES(w) LANES(l) {
            if (w == 0 && l == 0) {
                int go = 0;
                for (int n = 0; ; n++) {
                    if ((i32)m_ld32(&m->req_flag) == served + 1) { go = 1; break; }
                    if ((n & 63) == 63 && (m_ld32(&pq->frames_done) == (u32)njobs || m_ld32(&pq->abort) != 0u)) break;      // every frame is finished (or the launch is being abandoned)
                    part_poll_pause();
                }
                SM.red[1] = go;
            }
        }
        wg_sync();
        const int go = SM.red[1];
        wg_sync();
        if (!go) break;
        serve_request8(gK, jobs, m);
        served++;
        wg_sync();
    }
#if defined(IMCVT_PROF) && !defined(IMCVT_HOSTEMU)
    if (sc.prof && threadIdx.x < WG_THREADS && (threadIdx.x & 63u) < PF_N) atomicAdd(&sc.prof[(2 * NWAVES + (threadIdx.x >> 6)) * PF_N + (threadIdx.x & 63u)], SM.prof[threadIdx.x >> 6][threadIdx.x & 63u]);      // role 2's slice
#endif
}


HD void encode_ctu() {
    const FrameJob J = F.job;
    const int cy = F.ctu_y, cx = F.ctu_x;
    Avail a32; a32.l = cx > 0; a32.bl = 0; a32.a = cy > 0; a32.ar = (cy > 0) && (cx + 32 < J.wp);
    // ---- load the CTU: source pixels replicate the original edges, neighbours come from the padded reconstruction (:1613-1621)
    WAVES(w) LANES(l) {
        const int tid = w * 64 + l;
        if (cx + 32 <= J.w && (((size_t)J.img | (size_t)J.w) & 3) == 0) {      // CTU columns inside the picture, rows 4-byte aligned: coalesced dword loads
            NOUNROLL
            for (int i = tid; i < 256; i += WG_THREADS) {
                const int y = i >> 3, x4 = (i & 7) * 4;
                *(u32a *)&SM.org[y][x4] = g_ld32(J.img + (size_t)clip3(cy + y, 0, J.h - 1) * J.w + cx + x4);
            }
        } else {
            NOUNROLL
            for (int i = tid; i < 1024; i += WG_THREADS) {
                const int y = i >> 5, x = i & 31;
                SM.org[y][x] = g_ld8(J.img + (size_t)clip3(cy + y, 0, J.h - 1) * J.w + clip3(cx + x, 0, J.w - 1));
            }
        }
        if (tid < 32) SM.rec[tid + 1][0] = g_ld8(J.rcon + (size_t)clip3(cy + tid, 0, J.hp - 1) * J.wp + clip3(cx - 1, 0, J.wp - 1));
        if (tid >= 32 && tid < 32 + 65) {
            const int j = tid - 32 - 1;
            SM.rec[0][j + 1] = g_ld8(J.rcon + (size_t)clip3(cy - 1, 0, J.hp - 1) * J.wp + clip3(cx + j, 0, J.wp - 1));
        }
        // neighbour-map aprons: above row keeps sizes but forgets modes (:1633-1636); left column comes from the previous CTU
        if (tid >= 100 && tid < 100 + 10) {
            const int j = tid - 100;      // apron column index 0..9 <-> unit x = j-1
            SM.mapsz[0][j] = (u8)((cy > 0 && j >= 1 && j <= 8) ? g_ld8(F.sc.above_sz + (cx >> 2) + j - 1) : 32);
            SM.mapmode[0][j] = 1;
            if (cx == 0 && j < 9) { SM.mapsz[j + 1][0] = 32; SM.mapmode[j + 1][0] = 1; }
        }
    }
    wg_sync();

    const int team = F.mail != nullptr;
    enter_cu(0, 32, 0, 0, 1, pack_avail(a32));
    for (int i16_ = 0; i16_ < 4; i16_++) {
        const int y16 = (i16_ >> 1) * 16, x16 = (i16_ & 1) * 16;
        const Avail a16 = child_avail(a32, i16_);
        enter_cu(1, 16, y16, x16, 1, pack_avail(a16));
        for (int i8_ = 0; i8_ < 4; i8_++) {
            const int y8 = y16 + (i8_ >> 1) * 8, x8 = x16 + (i8_ & 1) * 8;
            const Avail a8 = child_avail(a16, i8_);
            enter_cu(2, 8, y8, x8, 0, 0);
            if (!(F.wide && lds_ld_i32(&WCTL.remote8) != 0 && !decide_cu8_remote(y8, x8, pack_avail(a8)))) decide_cu(2, 8, y8, x8, pack_avail(a8));
        }
        price_split(1, 16, y16, x16);
        if (!(team && F.posted[1]) || decide_remote(1, 16, y16, x16)) decide_cu(1, 16, y16, x16, pack_avail(a16));
    }
    price_split(0, 32, 0, 0);
    if (!(team && F.posted[0]) || decide_remote(0, 32, 0, 0)) decide_cu(0, 32, 0, 0, pack_avail(a32));

    // ---- store the reconstruction, end_of_slice_segment_flag, hand the CTU's bytes over (:1625-1630)
    u8 *live_sink = J.out + F.out_pos;
    WAVES(w) LANES(l) {
        const int tid = w * 64 + l;
        if (((size_t)J.rcon & 3) == 0) {                    // the padded plane's rows are multiples of 32: dword stores
            NOUNROLL
            for (int i = tid; i < 256; i += WG_THREADS) {
                const int y = i >> 3, x4 = (i & 7) * 4;
                const u8 *r = &SM.rec[y + 1][x4 + 1];
                g_st32(J.rcon + (size_t)(cy + y) * J.wp + cx + x4, (u32)r[0] | (u32)r[1] << 8 | (u32)r[2] << 16 | (u32)r[3] << 24);
            }
        } else {
            NOUNROLL
            for (int i = tid; i < 1024; i += WG_THREADS) {
                const int y = i >> 5, x = i & 31;
                g_st8(J.rcon + (size_t)(cy + y) * J.wp + cx + x, SM.rec[y + 1][x + 1]);
            }
        }
        if (tid < 8) {
            g_st8(F.sc.above_sz + (cx >> 2) + tid, SM.mapsz[8][tid + 1]);
            SM.mapsz[tid + 1][0] = SM.mapsz[tid + 1][8];          // right column becomes the next CTU's left apron
            SM.mapmode[tid + 1][0] = SM.mapmode[tid + 1][8];
        }
        if (tid == 64) {
            Arith a = SM.live;
            Sink ls; ls.base = live_sink; ls.off = 0;
            code_terminate(a, ls, (cy + 32 >= J.hp) && (cx + 32 >= J.wp));
            SM.live = a;
        }
    }
    wg_sync();
    WAVES(w) LANES(l) { if (w == 0 && l == 0) { F.out_pos += SM.live.cnt; SM.live.cnt = 0; } }
    wg_sync();
}

// A frame whose progress the host follows (FrameJob::prog; the host-pointer entry points copy finished CTU rows and stream bytes out WHILE the
// launch runs): every wave's stores have reached L2, the workgroup's L2 is written back, then the two words go out at system scope.  All threads call this.
HD void publish_progress(u32 *prog, u32 rows) {
    drain_stores();
    wg_sync();
    WAVES(w) LANES(l) { if (w == 0 && l == 0) { sys_release(); sys_st32(prog + 1, (u32)F.out_pos); sys_st32(prog, rows); } }
}

// Encode one frame with one workgroup.  `hdr` = the stream headers, prepared on the host (:664-690).
HDN void encode_frame(const Tables *gT, const ColdTables *gK, const FrameJob job, const Scratch sc, const u8 *hdr) {
    u32 *const prog = uni_p(job.prog);                  // (wave-uniform: the branches on it hold workgroup barriers)
    WAVES(w) LANES(l) {
        const int tid = w * 64 + l;
        const u32 *src = (const u32 *)gT; u32 *dst = (u32 *)&SM.T;
        NOUNROLL
        for (int i = tid; i < (int)(sizeof(Tables) / 4); i += WG_THREADS) dst[i] = src[i];
        for (int i = tid; i < job.hdr_len; i += WG_THREADS) g_st8(job.out + i, g_ld8(hdr + i));
        if (tid == 0) { F.job = job; F.sc = sc; F.out_pos = job.hdr_len; F.trace_n = 0; F.ctu_y = 0; F.ctu_x = 0; F.pace_inc = 65536 / ((job.hp / 32) * (job.wp / 32)); }      // (pace_mine runs on over this workgroup's frames, like the launch's progress sum)
#if defined(IMCVT_PROF) && !defined(IMCVT_HOSTEMU)
        if (l < PF_N) SM.prof[w][l] = 0;
#endif
    }
    wg_sync();
    WAVES(w) LANES(l) {
        const int tid = w * 64 + l;
        if (tid < CTX_STRIDE) { const u8 v = g_ld8(&gK->ctx_init[job.q][tid]); SM.cx[tid] = v; SM.cx0[tid] = v; }
        if (tid >= 128 && tid < 128 + 4 * RQ_CLASSES) (&SM.rthr[0][0])[tid - 128] = (i32)g_ld32(&gK->rthr[job.q][0][0] + (tid - 128));
        for (int i = tid; i < (PU_SIG_N + PU_GT_N) / 4; i += WG_THREADS) { if (i < PU_SIG_N / 4) *(u32a *)&SM.pu_sig[4 * i] = g_ld32(&gK->pu_sig[job.q][4 * i]); else *(u32a *)&SM.pu_gt[4 * (i - PU_SIG_N / 4)] = g_ld32(&gK->pu_gt[job.q][4 * (i - PU_SIG_N / 4)]); }
        if (tid == 64) arith_reset(SM.live);
    }
    wg_sync();
    for (int cy = 0; cy < job.hp; cy += 32)
        for (int cx = 0; cx < job.wp; cx += 32) {
            WAVES(w) LANES(l) {
                if (w == 0 && l == 0) {
                    F.ctu_y = cy; F.ctu_x = cx;
                    if (F.mail) {
                        if (m_ld32(&F.pq->abort) != 0u) F.aborted = 1;
                        if (F.wide) WCTL.part_ok = m_ld32(&F.mail->s[SLOT_8].pad0_[8]) != 0u;      // the partner workgroup (8x8 CUs' 2Nx2N sets) has reported in
                        // Pace control.  The launch ends with its slowest frame, and on a full device a workgroup's speed depends on
                        // the age of its waves (the older wave of a SIMD wins the issue arbitration, the guide's two-waves-per-SIMD
                        // section: workgroups dispatched later run their frames up to 1.5x slower).  So every main workgroup compares the
                        // share of its frame it has finished with the average over all of them, and one that lags more than a CTU and
                        // a half runs at raised wave priority until it has caught up to within half a CTU (priority outranks age).
                        if ((cy | cx) != 0) { F.pace_mine += F.pace_inc; m_add32(&F.pq->progress, (u32)F.pace_inc); }
                        const u32 taken_ = m_ld32(&F.pq->mains_taken), running = taken_ < (u32)F.pace_n ? taken_ : (u32)F.pace_n;      // main workgroups that have actually started (the counter is bumped past the planned number by late claimers)
                        const int avg = (int)(m_ld32(&F.pq->progress) / (running ? running : 1u)), lag = avg - F.pace_mine;
                        if (lag * 2 > 3 * F.pace_inc) F.prio_base = 2; else if (lag * 2 < F.pace_inc) F.prio_base = F.pace_base;
                        F.raised += F.prio_base != F.pace_base;
                    }
                }
            }
            wg_sync();
            if (F.aborted) return;                                    // (watchdog: the launch is being abandoned)
#ifndef IMCVT_HOSTEMU
            if (F.mail) { if (F.prio_base) SETPRIO(2); else SETPRIO(0); }
#endif
            encode_ctu();
            if (prog && cx + 32 >= job.wp && cy + 32 < job.hp) publish_progress(prog, (u32)(cy / 32 + 1));      // a CTU row is complete (the last one goes out with the frame, below)
        }
#if defined(IMCVT_PROF) && !defined(IMCVT_HOSTEMU)
    if (sc.prof && threadIdx.x < WG_THREADS && (threadIdx.x & 63u) < PF_N) atomicAdd(&sc.prof[(threadIdx.x >> 6) * PF_N + (threadIdx.x & 63u)], SM.prof[threadIdx.x >> 6][threadIdx.x & 63u]);   // role 0's slice
#endif
    WAVES(w) LANES(l) {
        if (w == 0 && l == 0) {
            Arith a = SM.live;
            Sink ls; ls.base = job.out + F.out_pos; ls.off = 0;
            arith_finish(a, ls);                                      // :1639-1640
            *job.out_len = F.out_pos + a.cnt;
            F.out_pos += a.cnt;
        }
    }
    wg_sync();
    if (prog) publish_progress(prog, (u32)(job.hp / 32) | PROG_DONE);
}

// ---- helper workgroup: serve requests until every slot it listens on has been closed ----------------------------------
HD void stage_tables(const Tables *gT) {
    WAVES(w) LANES(l) {
        const int tid = w * 64 + l;
        const u32 *src = (const u32 *)gT; u32 *dst = (u32 *)&SM.T;
        NOUNROLL
        for (int i = tid; i < (int)(sizeof(Tables) / 4); i += WG_THREADS) dst[i] = src[i];
    }
}
// Returns -1 when every frame is finished, or the main-workgroup index this workgroup took over: an idle helper becomes a main
// workgroup when frames wait in the queue and indices are left (fewer workgroups started as mains than there are frames to
// encode at once — e.g. compute units that hold fewer workgroups of this launch than expected).
HDN int helper_loop(const Tables *gT_, const ColdTables *gK_, const FrameJob *jobs_, const Scratch sc, TeamMail *mail_, PoolQ *pq_, int nmains_, int home_, int role_, int *counter_, int njobs_, int early_) {
    const Tables *const gT = uni_p(gT_); const ColdTables *const gK = uni_p(gK_); const FrameJob *const jobs = uni_p(jobs_); TeamMail *const mail = uni_p(mail_); PoolQ *const pq = uni_p(pq_); const int nmains = uni_i(nmains_); const int home = uni_i(home_); const int role = uni_i(role_);
    int *const counter = uni_p(counter_); const int njobs = uni_i(njobs_);
    const int early = uni_i(early_);                     // this workgroup was the first of its compute unit that did not become a main workgroup: the one whose wavefronts are oldest
    int taken = -1, served = 0;
    const int home_blk = home;
    stage_tables(gT);
#if defined(IMCVT_PROF) && !defined(IMCVT_HOSTEMU)
    if (threadIdx.x < WG_THREADS && (threadIdx.x & 63u) < PF_N) SM.prof[threadIdx.x >> 6][threadIdx.x & 63u] = 0;
#endif
    WAVES(w) LANES(l) { if (w == 0 && l == 0) { F.sc = sc; F.mail = (TeamMail *)0; F.pq = pq; F.seq[0] = 0; F.seq[1] = 0; F.frame = -1; m_add32(&pq->alive, 1u); } }
    wg_sync();
    for (;;) {
        const long long tidle = prof_now();
        WAVES(w) LANES(l) {
            if (w == 0 && l == 0) {
                int pick = -1, id = -1, round = 0, shard = home; u32 ticket = 0;
                hb_beat(); hb_set(0);                   // (idle time is not a gap)
                // Home shard first, then one other shard per round.  16x16 requests go before 32x32 ones (their main workgroups need the
                // answers sooner) three times out of four: requests arrive in that ratio (four 16x16 CUs — three quarters of them
                // offered — per 32x32 CU), and with a strict order a busy pool never got to the 32x32 queue: main workgroups were
                // seen waiting seconds for an answer that takes a millisecond (profiles/r03l_wd_probe.log).
                const int first = (served & 3) == 3 ? SLOT_32 : SLOT_16;
                // A main-workgroup index that is still free although the launch is 1.5 s old (LATE_MAIN_TICKS): its workgroup is being held back by the dispatcher (launches that fill every
                // workgroup slot: one or two of a thousand workgroups start only when another has left — seconds late, and a frame with it).  A running helper takes the index at
                // once, busy or not (the idle take-over below waits for the pool to drain); the late workgroup finds every index taken and becomes a helper.
                // One helper per compute unit at most (bits 16.. of the unit's arrival counter; a workgroup that arrives there later sees a count beyond every quota and becomes
                // a helper): a compute unit with a third main workgroup stretches three frames by a fifth (profiles/r06t_full_pool_verbose.log), one with a fourth doubles them.
                if (early && late_main_due(counter, home) && (i32)m_ld32(counter) < njobs && m_ld32(&pq->mains_taken) < (u32)nmains
                    && (m_add32(&pq->cu_count[pool_cu_key(home)], 0x10000u) >> 16) == 0u) {
                    if (m_add32(&pq->alive, (u32)-1) <= 1u) m_add32(&pq->alive, 1u);
                    else {
                        const u32 m = m_add32(&pq->mains_taken, 1u);
                        if (m < (u32)nmains) { id = (int)m; pick = -2; }
                        else m_add32(&pq->alive, 1u);
                    }
                }
                if (pick != -2)
                for (;;) {
                    for (int pass = 0; pass < 2 && pick < 0; pass++) {
                        shard = pass == 0 ? home : (home + 1 + round % (POOL_SHARDS - 1)) % POOL_SHARDS;
                        PoolShard *q = &pq->sh[shard];
                        for (int k_ = 0; k_ < MAIL_SLOTS && pick < 0; k_++) {
                            const int s_ = k_ == 0 ? first : (first ^ 1);
                            const u32 h = m_ld32(&q->head[s_]), t = m_ld32(&q->tail[s_]);
                            if ((i32)(t - h) > 0) { if (m_cas32(&q->head[s_], h, h + 1u)) { pick = s_; ticket = h; } else k_--; }   // lost the race for ticket h: look again
                        }
                    }
                    if (pick >= 0) break;
                    if ((round & 3) == 3) {
                        if (m_ld32(&pq->frames_done) == (u32)njobs || m_ld32(&pq->abort) != 0u) break;   // every frame is finished: no request can follow (or the watchdog fired)
                        if (round >= 127 && (i32)m_ld32(counter) < njobs && m_ld32(&pq->mains_taken) < (u32)nmains) {
                            // frames wait and a main-workgroup index is free: take it — unless this is the last helper (requests that
                            // were posted while a helper was alive must find one)
                            if (m_add32(&pq->alive, (u32)-1) <= 1u) m_add32(&pq->alive, 1u);
                            else {
                                const u32 m = m_add32(&pq->mains_taken, 1u);
                                if (m < (u32)nmains) { id = (int)m; pick = -2; break; }
                                m_add32(&pq->alive, 1u);
                            }
                        }
                    }
                    mail_idle_pause(round++);
                }
                if (pick >= 0) {                            // the ticket's owner publishes its index right after taking the ticket
                    const u32 *e = &pq->sh[shard].ring[pick][ticket % POOL_QCAP]; u32 v; int got = 0;
                    const unsigned long long t0 = wd_now();
                    for (int n = 0; ; n++) {
                        v = m_ld32(e);
                        const u32 lap = ((v >> 12) - ticket) & 0xFFFFFu;                  // 0: this ticket's entry (or nothing published in a fresh ring: v == 0)
                        if (v != 0u && lap == 0u) { got = 1; break; }
                        if (v != 0u && lap < 0x80000u) break;                              // a later lap's entry: the ticket is gone
                        if ((n & 15) == 15 && m_ld32(&pq->frames_done) == (u32)njobs) { pick = -1; break; }      // nothing can be outstanding any more
                        if (wd_poll(pq, t0, n, 2, shard, (int)ticket)) { pick = -1; break; }
                        mail_poll_pause();
                    }
                    if (got) id = (int)(v & 0xFFFu) - 1;
                    else if (pick >= 0) pick = -3;          // poll again
                }
                hb_set(wd_now());
                SM.red[1] = pick; SM.red[2] = id;
            }
        }
        wg_sync();
        prof_add(PF_CTUIO, tidle);                          // (booked as "idle": waiting for a request)
        const int slot = SM.red[1], id = SM.red[2];
        if (slot == -3) { wg_sync(); continue; }            // (every wave has read red[1..2] before thread 0 polls again and overwrites them)
        if (slot < 0) { taken = slot == -2 ? id : -1; break; }
        WAVES(w) LANES(l) { if (w == 0 && l == 0) { m_st32(&mail[id].s[slot].req_flag, 0x10000u | (u32)home_blk); m_st32(&mail[id].s[slot].pad0_[0], (u32)wd_now()); } }      // (debug: who took the request, when)
        serve_request(gK, jobs, &mail[id].s[slot]);
        served++;
        wg_sync();
    }
#if defined(IMCVT_PROF) && !defined(IMCVT_HOSTEMU)
    if (sc.prof && threadIdx.x < WG_THREADS && (threadIdx.x & 63u) < PF_N) atomicAdd(&sc.prof[(role * NWAVES + (threadIdx.x >> 6)) * PF_N + (threadIdx.x & 63u)], SM.prof[threadIdx.x >> 6][threadIdx.x & 63u]);
#endif
    WAVES(w) LANES(l) { if (w == 0 && l == 0) F.kept = served; }      // (debug buffer, HB_GAP)
    (void)role;
    return taken;
}

// ---- kernel body (shared by the gfx950 kernel and the host emulation) -------------------------------------------------
struct KArgs {
    const Tables *gT; const ColdTables *gK; const FrameJob *jobs; const u8 *hdrs; int njobs;
    const Scratch *scr; int *counter; i32 *trace; int trace_cap; unsigned long long *prof;
    TeamMail *mail; PoolQ *pq;
    unsigned long long *fclk;                   // optional debug buffer, 4 words per frame: start clock, end clock (100 MHz), block, requests kept local
    int post16, post32;                         // per mille of the 16x16 / 32x32 CUs a main workgroup offers to the helpers
    int lim16, lim32, prio, quota;              // quota: workgroups per compute unit that start as main workgroups                     // pool tuning: unclaimed requests per shard beyond which a main workgroup keeps a CU (16x16 / 32x32); wave priority of the main workgroups
    int team_size, nteams, nhelp;               // team_size 1: every workgroup encodes whole frames alone; > 1: `nteams` main workgroups + a pool of `nhelp` helper workgroups
    int npart;                                  // wide pool launches: partner workgroups (partner i takes the 2Nx2N sets of main workgroup i's 8x8 CUs, "8x8 CUs with a partner workgroup")
    int role;                                   // pool launches: 0 roles by placement (one launch holds main and helper workgroups); 1 / 2: this launch holds the main / the helper workgroups of a
                                                // pool that is spread over two cooperating launches (different workgroup sizes on disjoint sets of compute units, hevc_hip.hip launch_split)
};
#ifdef IMCVT_HOSTEMU
HD int next_job(int *counter) { return (*counter)++; }
#else
HD int next_job(int *counter) { return atomicAdd(counter, 1); }
#endif
HD void kernel_main(const KArgs &A, int block) {
    // The main workgroups are the first blocks of the grid, the helpers follow: if fewer workgroups are resident than were
    // launched, the ones that are missing are helpers.  Nothing depends on a workgroup that is not running: a main workgroup
    // posts requests only once a helper has reported in (helpers stay until every main workgroup has left, so what is posted
    // is served), evaluates the CUs itself until then, and frames are pulled by whichever main workgroups run.
#ifndef IMCVT_HOSTEMU
    if (A.team_size < 0) {      // residency census (debug): how many workgroups of this launch are on the device at the same time
        if (threadIdx.x == 0) {
            atomicAdd(A.counter, 1);
            const long long t0 = clock64();
            while (clock64() - t0 < 2000000) __builtin_amdgcn_s_sleep(32);
            atomicMax(A.counter + 1, (int)m_ld32(A.counter));
            atomicAdd(A.counter, -1);
        }
        return;
    }
#endif
#ifndef IMCVT_HOSTEMU
    // residency diagnostic: counter[2] = workgroups of this launch running now, counter[3] = the most there ever were (imcvt_hevc_last_resident)
    // (debug buffer: which SIMD each wavefront of this workgroup runs on — a byte each, 0x80 | HW_ID.simd_id — in word 1 of the workgroup's record; tools/pool_probe.py PP_OUTLIER)
    if (A.fclk && (threadIdx.x & 63u) == 0u && (threadIdx.x >> 6) < 8u) ((volatile u8 *)(A.fclk + 4 * A.njobs + 4 * block + 1))[threadIdx.x >> 6] = (u8)(0x80u | ((__builtin_amdgcn_s_getreg((31 << 11) | 4) >> 4) & 3u));
    if (threadIdx.x == 0) {
        atomicMax(A.counter + 3, atomicAdd(A.counter + 2, 1) + 1);
        const unsigned long long now = wall_clock64();                  // 100 MHz: when the first and the last workgroup of the launch started
        atomicMin((unsigned long long *)(A.counter + 4), now); atomicMax((unsigned long long *)(A.counter + 6), now);
    }
#ifdef IMCVT_REGCNT
    if (threadIdx.x < REG_N) SM.regcnt[threadIdx.x] = 0;
    __syncthreads();
    struct LeaveR { unsigned long long *prof; __device__ ~LeaveR() { __syncthreads(); if (prof && threadIdx.x < REG_N) atomicAdd(prof + 3 * NWAVES * PF_N + threadIdx.x, (unsigned long long)SM.regcnt[threadIdx.x]); } } leave_r_{ A.prof };
#endif
    struct Leave { int *c; unsigned long long *dbg; int blk; __device__ ~Leave() { if (threadIdx.x == 0) { atomicAdd(c + 2, -1);
        if (dbg) { dbg[4 * blk] = HB_GAP; dbg[4 * blk + 1] = HB_WHEN; dbg[4 * blk + 2] = (unsigned long long)hw_cu_key() | (unsigned long long)(F.mail ? 1 : 0) << 32; dbg[4 * blk + 3] = wall_clock64(); } } } } leave_{ A.counter, A.fclk ? A.fclk + 4 * A.njobs : nullptr, block };
#ifdef IMCVT_HB
    if (threadIdx.x == 0) { F.hb_last = 0; F.hb_gap = 0; F.hb_when = 0; }
#endif
#endif
    WAVES(w) LANES(l) { if (w == 0 && l == 0 && wg_is_wide()) { for (int i = 0; i < XWAVES; i++) { SplitQ &q = XM(i).q; q.go = 0; q.mid = 0; q.rdone = 0; q.done = 0; } WCTL.a_go = 0; for (int i = 0; i < NLEND; i++) WCTL.lend_done[i] = 0; WCTL.b_seg = 0; WCTL.b_cons = 0; WCTL.b_hand = 0; WCTL.cu8 = 0; WCTL.part_ok = 0; WCTL.seq8 = 0; WCTL.stale8 = 0; WCTL.remote8 = 0; WCTL.ans_seq = 0; WCTL.solo2n = 0; WCTL.lv_seq = 0; WCTL.lv_taken = 0; CTXQ.go = 0; CTXQ.done = 0; PUX.pu_seq = 0; PUX.b_seq = 0; PUX.r_seq = 0; } }
    WAVES(w) LANES(l) { if (w == 0 && l == 0) { F.pipe = wg_has_pipe_wave(); F.wide = wg_is_wide() && PU_HINTS; SM.pipe_a = 0; SM.pipe_b = 0; SM.nxn_lane = 0; SM.pu0_ready = 0; SM.pu0_taken = 0; } }      // (read after the barriers below)
    const int pool = A.team_size > 1 && A.nhelp > 0;
    const int nm = A.nteams > 0 ? A.nteams : 1, tot = nm + A.nhelp;
    (void)tot;
    // Roles are chosen when a workgroup starts, by where it landed: of the first blocks of the launch (they are placed first) the first `quota` on a compute unit
    // become main workgroups, what is left goes to the first `quota` arrivals of a compute unit among the others — see below; every compute unit carries the same
    // mix whatever order the dispatcher fills them in, and if fewer workgroups are resident than were launched it is helpers that are missing.  Nothing depends on a workgroup that is not running: a main workgroup posts requests only once a helper has
    // reported in (helpers stay until every frame is finished, so what is posted is served) and evaluates the CUs itself until
    // then; frames are pulled by whichever main workgroups run, and an idle helper takes a free main index when frames wait.
    int team = block;
    Scratch sc = A.scr[block];
    sc.trace_cap = A.trace_cap; sc.prof = A.prof;
    if (pool) {
        WAVES(w) LANES(l) {
            if (w == 0 && l == 0) {
#ifdef IMCVT_HOSTEMU
                const int key = block % POOL_CU_KEYS;
#else
                const int key = hw_cu_key() % POOL_CU_KEYS;
#endif
                int mid = -1;
                // (a launch of main workgroups only: every workgroup takes an index while they last; of helpers only: none does — an idle helper may still take one later, helper_loop)
                const int first_blocks = block < nm && A.quota > 0;       // one of the first nm blocks of the launch (counted separately in bits 8 .. 15 of the compute unit's arrival counter)
                const u32 seen = (A.role == 1 || A.role == 2) ? 0u : m_add32(&A.pq->cu_count[key], first_blocks ? 0x101u : 1u);
                const int arrival = (int)(seen & 0xFFu), arrival_first = (int)(seen >> 8 & 0xFFu);      // which workgroup of this launch on its compute unit (0, 1, ...), and which of the first blocks
                // Who becomes a main workgroup (round 6).  Of the FIRST nm blocks of the launch, the first `quota` that land on a compute unit; then, a moment later — when those have
                // made their claims — whatever indices are left go to later blocks, the first `quota` arrivals of a compute unit as in rounds 3 - 5.  Why the first blocks first: on
                // some boxes a workgroup that the dispatcher places late on its compute unit runs a frame 1.1 - 1.9 x slower than one placed early, and "the first two arrivals of
                // every compute unit" alone let late blocks that happened to start before a compute unit's second early block become main workgroups — the bench launch took 4.85 or
                // 5.05 s by that luck, 4.80 s twenty times out of twenty with the first blocks as main workgroups (profiles/r06zb_roles_by_block.log).  Why not simply the first nm
                // blocks: on other boxes the dispatcher puts three of them on some compute units and one on others in nearly every launch, and three main workgroups on a compute
                // unit cost 4 - 6 % (profiles/r06ze_roles_default.log) where the first two arrivals ran 4.81 s flat.  (quota < 0: the rule of rounds 3 - 5 alone, A/B.)
                const int q_ = A.quota < 0 ? -A.quota : A.quota;
                int want;
                if (A.role == 1) want = 1;
                else if (A.role == 2) want = 0;
                else if (A.quota < 0) want = arrival < q_;
                else if (first_blocks) want = arrival_first < q_;
                else {
#ifndef IMCVT_HOSTEMU
                    const unsigned long long t_in = wall_clock64();
                    while (wall_clock64() - t_in < ROLE_GRACE_TICKS) __builtin_amdgcn_s_sleep(8);
#endif
                    want = arrival < q_;
                }
                SM.next_frame = arrival;                 // (a free word until the frame loop)
#ifndef IMCVT_HOSTEMU
                if (A.fclk) ((volatile u8 *)(A.fclk + 4 * A.njobs + 4 * block + 1))[7] = (u8)(0x80u | (u32)arrival);      // (debug buffer, beside the wavefronts' SIMDs)
#endif
                if (want && m_ld32(&A.pq->mains_taken) < (u32)nm) { const u32 m = m_add32(&A.pq->mains_taken, 1u); if (m < (u32)nm) mid = (int)m; }
                // wide launches with partner workgroups: the next `npart` workgroups to start serve the 8x8 CUs of main workgroups 0 .. npart - 1 (partner8_loop)
                int pid = -1;
                if (mid < 0 && A.npart > 0 && wg_is_wide() && m_ld32(&A.pq->parts_taken) < (u32)A.npart) { const u32 p_ = m_add32(&A.pq->parts_taken, 1u); if (p_ < (u32)A.npart) pid = (int)p_; }
                SM.red[0] = mid; SM.red[1] = pid;
            }
        }
        wg_sync();
        team = SM.red[0];
        const int pid = SM.red[1];
        const int arrival = SM.next_frame;
        wg_sync();
        if (team < 0 && pid >= 0) { sc.trace = (i32 *)0; partner8_loop(A.gT, A.gK, A.jobs, sc, A.mail, A.pq, pid, A.njobs); return; }
        if (team < 0) {
            sc.trace = (i32 *)0;
            team = helper_loop(A.gT, A.gK, A.jobs, sc, A.mail, A.pq, nm, block % POOL_SHARDS, 1, A.counter, A.njobs, arrival <= (A.quota < 0 ? -A.quota : A.quota));
            if (team < 0) return;
        }
    }
#ifndef IMCVT_HOSTEMU
    // the main workgroup of a team carries the frame's critical path while its helpers have ~40 % slack: it wins the VALU
    // arbitration of the SIMDs it shares with them (measured: 320 teams 5.75 s -> 4.95 s, 256 teams 4.78 s -> 4.31 s)
    if (pool && A.prio >= 2) SETPRIO(2);
#endif
    WAVES(w) LANES(l) { if (w == 0 && l == 0) { F.prio_base = (pool && A.prio >= 2) ? 2 : 0; F.pace_base = F.prio_base; F.pace_n = nm; F.pace_mine = 0; } }
    WAVES(w) LANES(l) { if (w == 0 && l == 0) { F.mail = pool ? A.mail + team : (TeamMail *)0; F.pq = A.pq; F.main_id = team; F.lim[SLOT_16] = A.lim16; F.lim[SLOT_32] = A.lim32; F.post_pm[SLOT_16] = A.post16; F.post_pm[SLOT_32] = A.post32; F.post_acc[SLOT_16] = 500; F.post_acc[SLOT_32] = 500; F.posted[0] = 0; F.posted[1] = 0; F.stale[0] = 0; F.stale[1] = 0; F.gaveup = 0; F.seq[0] = 0; F.seq[1] = 0; F.aborted = 0; } }
    // (this barrier is load-bearing: without it hipcc threads the `thread 0` branch above into the one inside the loop, and the
    // other lanes of wave 0 then reach the loop's first barrier BEFORE thread 0 has stored next_frame — seen as a memory fault)
    wg_sync();
    for (;;) {
        WAVES(w) LANES(l) { if (w == 0 && l == 0) SM.next_frame = next_job(A.counter); }
        wg_sync();
        const int f = SM.next_frame;
        wg_sync();
        if (f >= A.njobs) break;
        sc.trace = (f == 0) ? A.trace : (i32 *)0;
        WAVES(w) LANES(l) { if (w == 0 && l == 0) { F.frame = f; F.kept = 0; F.raised = 0; F.waited = 0; F.waited_max = 0; } }
        if (F.aborted) break;
#ifndef IMCVT_HOSTEMU
        if (A.fclk && threadIdx.x == 0) A.fclk[4 * f] = wall_clock64();
#endif
        encode_frame(A.gT, A.gK, A.jobs[f], sc, A.hdrs + (size_t)HDR_MAX * f);
        if (pool) { WAVES(w) LANES(l) { if (w == 0 && l == 0) m_add32(&A.pq->frames_done, 1u); } }      // (every request of the frame has been answered)
#ifndef IMCVT_HOSTEMU
        if (A.fclk && threadIdx.x == 0) { A.fclk[4 * f + 1] = wall_clock64(); A.fclk[4 * f + 2] = (unsigned long long)block | (unsigned long long)hw_cu_key() << 32 | (unsigned long long)(F.raised & 0xFFFF) << 48; A.fclk[4 * f + 3] = (unsigned long long)F.kept | (unsigned long long)(F.waited / 1000u) << 16 | (unsigned long long)(F.waited_max / 1000u) << 40; }
#endif
    }
}
#undef F
