// tests/hostemu/hostemu.cpp — TEST-ONLY harness: compiles the device encoder source (imcvt_amd/csrc/hevc_core.h,
// hevc_frame.h) for the host.  Every lane of every 192-thread (256 with a pipe wave) workgroup is a cooperative fiber with its own stack, so
// the kernel runs under real SIMT semantics: registers survive barriers, divergent lanes make independent progress,
// wave_sync()/wg_sync() are true barriers and ballots/shuffles are collectives.  It exists so the bit-exactness of
// the kernel LOGIC can be checked against the oracle on a machine without a GPU; it is not part of the product,
// is never loaded by imcvt_amd, and is not a fallback.
#define IMCVT_HOSTEMU 1
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <stdint.h>

// ---- fibers -----------------------------------------------------------------------------------------------
extern "C" void emu_ctx_switch(void **save_sp, void *load_sp);
asm(".text\n.globl emu_ctx_switch\n.type emu_ctx_switch,@function\nemu_ctx_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp,(%rdi)\n  movq %rsi,%rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n  ret\n");

#define EMU_WG_THREADS_MAX 512                              // 192, 256 when the workgroups carry a pipe wave (hostemu_set_pipe), 512 when they are wide (pipe wave + four partner wavefronts)
#if defined(EMU_DEFAULT_WIDE)
static int g_wg_threads = 512;
#elif defined(EMU_DEFAULT_PIPE)
static int g_wg_threads = 256;
#else
static int g_wg_threads = 192;
#endif
#define EMU_WG_THREADS g_wg_threads
#define EMU_MAX_WG 6                                        // main + helper workgroups of one emulated launch
#define EMU_THREADS (EMU_WG_THREADS_MAX * EMU_MAX_WG)
#define EMU_STACK (512 * 1024)
struct EmuFiber {
    void *sp; char *stack; int done;
    volatile unsigned *wait_gen; unsigned wait_val;        // blocked until *wait_gen != wait_val
};
static EmuFiber g_fib[EMU_THREADS];
static void *g_main_sp;
static int g_cur;                                           // running fiber (= blockIdx.x * 192 + threadIdx.x)
static int g_nfib = 192;                                    // fibers of this run (192 or 256 per workgroup)
static unsigned g_wave_gen[8 * EMU_MAX_WG], g_wave_arr[8 * EMU_MAX_WG], g_wg_gen[EMU_MAX_WG], g_wg_arr[EMU_MAX_WG];
static uint64_t g_xchg[EMU_THREADS];                        // collective exchange slots
static void (*g_entry)(void);
static unsigned g_yield_gen;                                // a yielding fiber is runnable again at once

static int emu_lane() { return g_cur & 63; }
static int emu_wave() { return (g_cur % EMU_WG_THREADS) >> 6; }
static int emu_block() { return g_cur / EMU_WG_THREADS; }
static void emu_wait(volatile unsigned *gen, unsigned val) {
    EmuFiber &f = g_fib[g_cur];
    f.wait_gen = gen; f.wait_val = val;
    emu_ctx_switch(&f.sp, g_main_sp);
}
static void emu_wave_sync() {
    const int w = g_cur >> 6;                               // global wave index
    if (++g_wave_arr[w] == 64) { g_wave_arr[w] = 0; g_wave_gen[w]++; }
    else emu_wait(&g_wave_gen[w], g_wave_gen[w]);
}
static void emu_wg_sync() {
    const int b = emu_block();
    if (++g_wg_arr[b] == (unsigned)EMU_WG_THREADS) { g_wg_arr[b] = 0; g_wg_gen[b]++; }
    else emu_wait(&g_wg_gen[b], g_wg_gen[b]);
}
static void emu_yield() {                                   // spin-wait on another workgroup's flag: let everyone else run
    EmuFiber &f = g_fib[g_cur];
    f.wait_gen = &g_yield_gen; f.wait_val = g_yield_gen - 1;
    emu_ctx_switch(&f.sp, g_main_sp);
}
static uint64_t emu_ballot(int p) {
    g_xchg[g_cur] = p ? 1 : 0;
    emu_wave_sync();
    uint64_t m = 0;
    for (int i = 0; i < 64; i++) m |= g_xchg[(g_cur & ~63) + i] << i;
    emu_wave_sync();
    return m;
}
static int emu_shfl(int v, int src_lane) {                  // value of `v` held by lane src_lane (own value if out of range)
    g_xchg[g_cur] = (uint64_t)(uint32_t)v;
    emu_wave_sync();
    const int r = (src_lane >= 0 && src_lane < 64) ? (int)(uint32_t)g_xchg[(g_cur & ~63) + src_lane] : v;
    emu_wave_sync();
    return r;
}
// Matrix instructions (wave collectives): every lane deposits its operand registers, then computes the result registers
// the hardware would hand it.  Layouts as verified on the device by tools/mfma_probe.hip.
static uint32_t g_mx[8 * EMU_MAX_WG][64][8];                // per lane: a[0..3], b[0..3]
static long g_mfma_calls[2];                                // wave-level matrix instructions issued so far (32x32x32, 16x16x32)
static int emu_sx8(uint32_t w, int k) { return (int)(int8_t)(w >> (8 * k)); }
// v_mfma_i32_32x32x32_i8: lane l holds A[l%32][16*(l/32) .. +15], B[16*(l/32) .. +15][l%32]; acc r: D[8*(r/4) + 4*(l/32) + r%4][l%32]
static void emu_mfma32(const uint32_t *a, const uint32_t *b, int *acc) {
    const int w = g_cur >> 6, l = g_cur & 63, i = l & 31, h = l >> 5;
    g_mfma_calls[0] += l == 0;
    for (int d = 0; d < 4; d++) { g_mx[w][l][d] = a[d]; g_mx[w][l][4 + d] = b[d]; }
    emu_wave_sync();
    for (int r = 0; r < 16; r++) {
        const int m = 8 * (r / 4) + 4 * h + r % 4;
        int s = 0;
        for (int hh = 0; hh < 2; hh++) for (int e = 0; e < 16; e++)
            s += emu_sx8(g_mx[w][m + 32 * hh][e / 4], e % 4) * emu_sx8(g_mx[w][i + 32 * hh][4 + e / 4], e % 4);
        acc[r] += s;
    }
    emu_wave_sync();
}
// v_mfma_i32_16x16x32_i8: lane l holds A[l%16][8*(l/16) .. +7], B[8*(l/16) .. +7][l%16]; acc r: D[4*(l/16) + r][l%16]
static void emu_mfma16(const uint32_t *a, const uint32_t *b, int *acc) {
    const int w = g_cur >> 6, l = g_cur & 63, i = l & 15, h = l >> 4;
    g_mfma_calls[1] += l == 0;
    for (int d = 0; d < 2; d++) { g_mx[w][l][d] = a[d]; g_mx[w][l][4 + d] = b[d]; }
    emu_wave_sync();
    for (int r = 0; r < 4; r++) {
        const int m = 4 * h + r;
        int s = 0;
        for (int hh = 0; hh < 4; hh++) for (int e = 0; e < 8; e++)
            s += emu_sx8(g_mx[w][m + 16 * hh][e / 4], e % 4) * emu_sx8(g_mx[w][i + 16 * hh][4 + e / 4], e % 4);
        acc[r] += s;
    }
    emu_wave_sync();
}
struct Shm;
static Shm *g_shm_of[EMU_MAX_WG];                          // each emulated workgroup's LDS image
static unsigned char *g_pipe_of[EMU_MAX_WG];               // ... and its dynamic part (the pipe wave's slice)
static int emu_pipe_on() { return g_wg_threads > 192; }
static int emu_wide_on() { return g_wg_threads >= 512; }
static int emu_late_main() { return getenv("HOSTEMU_LATE_MAIN") != nullptr; }      // (hevc_core.h late_main_due)
static long g_spins;
static void emu_set_shm(int wg);                            // (defined below, next to the device source's LDS pointer)
static void emu_trampoline() {
    g_entry();
    g_fib[g_cur].done = 1;
    for (;;) emu_ctx_switch(&g_fib[g_cur].sp, g_main_sp);
}
static void emu_run(void (*entry)(void)) {
    g_entry = entry;
    memset(g_wave_gen, 0, sizeof g_wave_gen); memset(g_wave_arr, 0, sizeof g_wave_arr); memset(g_wg_gen, 0, sizeof g_wg_gen); memset(g_wg_arr, 0, sizeof g_wg_arr);
    for (int i = 0; i < g_nfib; i++) {
        EmuFiber &f = g_fib[i];
        if (!f.stack) f.stack = (char *)malloc(EMU_STACK);
        f.done = 0; f.wait_gen = nullptr;
        // initial frame: six callee-saved registers, then the return address; rsp % 16 == 8 on entry to the trampoline
        uintptr_t top = ((uintptr_t)f.stack + EMU_STACK) & ~(uintptr_t)15;
        void **sp = (void **)(top - 8);
        *--sp = (void *)emu_trampoline;
        for (int k = 0; k < 6; k++) *--sp = nullptr;
        f.sp = sp;
    }
    for (;;) {
        int live = 0, ran = 0, worked = 0;
        for (int i = 0; i < g_nfib; i++) {
            EmuFiber &f = g_fib[i];
            if (f.done) continue;
            live++;
            if (f.wait_gen && *f.wait_gen == f.wait_val) continue;
            worked += f.wait_gen != &g_yield_gen;           // a fiber that comes back from a yield has done nothing yet
            f.wait_gen = nullptr; g_cur = i; ran++;
            emu_set_shm(i / EMU_WG_THREADS);
            emu_ctx_switch(&g_main_sp, f.sp);
        }
        if (!live) break;
        if (!ran) { fprintf(stderr, "hostemu: deadlock (divergent barrier)\n"); abort(); }
        if (!worked && ++g_spins > 1000000) { fprintf(stderr, "hostemu: deadlock (workgroups wait for each other)\n"); abort(); } else if (worked) g_spins = 0;
    }
}

#include "../../imcvt_amd/csrc/hevc_frame.h"
#include "../../imcvt_amd/csrc/hevc_tables.h"

static void emu_set_shm(int wg) { g_shm_host = g_shm_of[wg]; g_pipe_host = g_pipe_of[wg]; }
static KArgs g_args;
static u32 *g_prog; static int g_prog_n;
// the progress record frame i of the last emulated launch ended with: word 0 (CTU rows | PROG_DONE), word 1 (stream bytes)
extern "C" unsigned hostemu_prog(int i, int k) { return (g_prog && i >= 0 && i < g_prog_n && (k == 0 || k == 1)) ? g_prog[2 * i + k] : 0u; }
static int g_role_split;      // 1: the emulated pool is "two launches" — blocks below nteams are a launch of main workgroups (role 1), the others a launch of helpers (role 2)
extern "C" void hostemu_set_role_split(int on) { g_role_split = on; }
static void emu_entry() { KArgs a = g_args; if (g_role_split && a.nhelp > 0) a.role = emu_block() < a.nteams ? 1 : 2; kernel_main(a, emu_block()); }

// nhelp 0: `nmains` workgroups encode the frames alone (frames pulled one after the other); > 0: they hand the 16x16 / 32x32 candidate
// sets to a pool of `nhelp` helper workgroups (hevc_frame.h)
static int emu_encode(int n, unsigned char *const *pbuffers, const unsigned char *const *imgs, unsigned char *const *rcons,
                      int *ysz, int *xsz, int qpd6, int *out_len, int *trace, int trace_cap, int nmains, int nhelp, int npart = 0) {
    static Tables T; static ColdTables K; static int ready = 0;
    if (!ready) { imcvt::build_tables(T, K); ready = 1; }
    if (nmains < 1) nmains = 1;
    if (nhelp < 0) nhelp = 0;
    if (npart < 0 || nhelp < 1) npart = 0;
    const int nteams = nmains, nwg = nmains + nhelp + npart;
    if (nwg > EMU_MAX_WG) return -1;
    FrameJob *jobs = (FrameJob *)calloc(n, sizeof(FrameJob));
    u8 *hdrs = (u8 *)calloc(n, HDR_MAX);
    free(g_prog); g_prog = (u32 *)calloc((size_t)n, 2 * sizeof(u32)); g_prog_n = n;      // every emulated frame reports its progress (hevc_frame.h publish_progress)
    for (int i = 0; i < n; i++) {
        const int h = ysz[i], w = xsz[i];
        FrameJob &job = jobs[i];
        job.img = imgs[i]; job.out = pbuffers[i]; job.rcon = rcons[i]; job.h = h; job.w = w; job.q = qpd6;
        job.hp = ((h < 8192 ? h : 8192) + 31) / 32 * 32; job.wp = ((w < 8192 ? w : 8192) + 31) / 32 * 32;
        job.hdr_len = imcvt::build_headers(hdrs + (size_t)HDR_MAX * i, qpd6, job.hp, job.wp);
        out_len[i] = 0; job.out_len = &out_len[i];
        job.prog = g_prog + 2 * i;
        ysz[i] = job.hp; xsz[i] = job.wp;
    }
    Scratch sc[EMU_MAX_WG]; void *pool[EMU_MAX_WG];
    for (int b = 0; b < nwg; b++) {
        g_shm_of[b] = (Shm *)calloc(1, sizeof(Shm));
        g_pipe_of[b] = (unsigned char *)aligned_alloc(16, WIDE_LDS_BYTES); memset(g_pipe_of[b], 0xA5, WIDE_LDS_BYTES);      // (LDS is not zeroed on the device either)
        pool[b] = calloc(1, scratch_bytes_per_wg());
        scratch_carve(sc[b], (u8 *)pool[b]);
    }
    TeamMail *mail = (TeamMail *)aligned_alloc(256, sizeof(TeamMail) * nteams);
    memset(mail, 0, sizeof(TeamMail) * nteams);
    PoolQ *pq = (PoolQ *)aligned_alloc(256, sizeof(PoolQ));
    memset(pq, 0, sizeof(PoolQ));
    int counter[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    g_args.gT = &T; g_args.gK = &K; g_args.jobs = jobs; g_args.hdrs = hdrs; g_args.njobs = n; g_args.scr = sc; g_args.counter = counter;
    g_args.trace = trace; g_args.trace_cap = trace_cap; g_args.prof = nullptr; g_args.mail = mail; g_args.pq = pq; g_args.team_size = nhelp > 0 ? 2 : 1; g_args.nteams = nteams; g_args.nhelp = nhelp; g_args.post16 = 750; g_args.post32 = 1000; g_args.lim16 = 1; g_args.lim32 = 1; g_args.prio = 0; g_args.quota = getenv("HOSTEMU_QUOTA") ? atoi(getenv("HOSTEMU_QUOTA")) : 1; g_args.fclk = nullptr; g_args.role = 0; g_args.npart = npart;      // (quota 0: no workgroup starts as a main one — idle helpers take the roles)
    g_nfib = nwg * EMU_WG_THREADS; g_spins = 0;
    emu_run(emu_entry);
    for (int b = 0; b < nwg; b++) { free(pool[b]); free(g_shm_of[b]); free(g_pipe_of[b]); }
    free(mail); free(pq); free(jobs); free(hdrs);
    return 0;
}
extern "C" int hostemu_HEVCImageEncoder(unsigned char *pbuffer, const unsigned char *img, unsigned char *img_rcon,
                                        int *ysz, int *xsz, int qpd6, int *trace, int trace_cap) {
    int len = 0;
    unsigned char *pb[1] = { pbuffer }; const unsigned char *im[1] = { img }; unsigned char *rc[1] = { img_rcon };
    if (emu_encode(1, pb, im, rc, ysz, xsz, qpd6, &len, trace, trace_cap, 1, 0) < 0) return -1;
    return len;
}
// n frames by `nmains` main workgroups and a pool of `nhelp` helper workgroups (the frames are pulled from one queue, as on the device)
extern "C" int hostemu_HEVCImageEncoderPool(int n, unsigned char *const *pbuffers, const unsigned char *const *imgs, unsigned char *const *rcons,
                                            int *ysz, int *xsz, int qpd6, int *out_len, int nmains, int nhelp) {
    return emu_encode(n, pbuffers, imgs, rcons, ysz, xsz, qpd6, out_len, nullptr, 0, nmains, nhelp);
}
// ... and `npart` partner workgroups (wide workgroups only: partner i evaluates the 2Nx2N sets of main workgroup i's 8x8 CUs)
extern "C" int hostemu_HEVCImageEncoderPool3(int n, unsigned char *const *pbuffers, const unsigned char *const *imgs, unsigned char *const *rcons,
                                             int *ysz, int *xsz, int qpd6, int *out_len, int nmains, int nhelp, int npart) {
    return emu_encode(n, pbuffers, imgs, rcons, ysz, xsz, qpd6, out_len, nullptr, 0, nmains, nhelp, npart);
}
extern "C" void hostemu_remote8_stats(long *o, int reset) { for (int i = 0; i < 3; i++) { o[i] = g_remote8[i]; if (reset) g_remote8[i] = 0; } }
// 1: the emulated workgroups have 256 threads, the fourth wavefront being the pipe wave (hevc_frame.h nxn_pipe); 0: 192 threads
extern "C" void hostemu_set_pipe(int on) { g_wg_threads = on ? 256 : 192; }
// 2: wide workgroups (512 threads: pipe wave + four partner wavefronts, the trial coders of the 8x8 CUs split over two wavefronts each)
extern "C" void hostemu_set_threads(int n) { g_wg_threads = n >= 512 ? 512 : n >= 256 ? 256 : 192; }
extern "C" int hostemu_wide_lds_bytes(void) { return (int)WIDE_LDS_BYTES; }
extern "C" long hostemu_mfma_calls(int kind) { return g_mfma_calls[kind & 1]; }
extern "C" int hostemu_shm_bytes(void) { return (int)sizeof(Shm); }
extern "C" int hostemu_pipe_lds_bytes(void) { return (int)PIPE_LDS_BYTES; }

// The device's lead sink (hevc_core.h lsink_begin / lsink_flush8) fed a list of leads the way a trial coder feeds it (ring of 16, a flush per
// eight leads, the rest at the end); returns its `hit` flag.  st = { nbytes, bufbyte, zeros } of the entry state.
extern "C" int hostemu_lsink_guard(const unsigned short *leads, int n, const int *st) {
    static unsigned char gbuf[TRIAL_BYTES];
    alignas(4) unsigned short ring[LRING];
    Arith a; arith_reset(a); a.nbytes = st[0]; a.bufbyte = st[1]; a.zeros = st[2];
    LeadSink s; lsink_begin(s, a, ring, gbuf);
    int qn = 0;
    for (int i = 0; i < n; i++) {
        if ((i & 7) == 0) lsink_sync(s, qn);                 // (as between token blocks: at most eight leads join between two syncs)
        ring[qn & (LRING - 1)] = leads[i]; qn++;
    }
    lsink_finish(s, qn);
    for (int i = 0; i < n; i++) if (((unsigned short *)gbuf)[i] != leads[i]) return -1;      // (the list in memory is the list)
    return s.hit;
}
#ifdef IMCVT_DBGCNT
extern "C" void hostemu_dbg(long *o) { for (int i = 0; i < 8; i++) o[i] = g_dbg[i]; }
#endif
// The device's resolve_leads (hevc_core.h) on one emulated wavefront: leads -> bytes at dst[st[3] ..), byte-level state in / out in st = { nbytes, bufbyte, zeros, cnt }
static struct { const unsigned char *list; int n; Arith a; unsigned char *dst; } g_rl;
static void emu_entry_resolve() { Arith a = g_rl.a; resolve_leads(a, g_rl.list, g_rl.n, g_rl.dst); if (emu_lane() == 0) g_rl.a = a; }
extern "C" int hostemu_resolve_leads(const unsigned short *leads, int n, int *st, unsigned char *dst) {
    static Shm *shm = (Shm *)calloc(1, sizeof(Shm));
    g_shm_of[0] = shm; g_pipe_of[0] = nullptr;
    Arith a; arith_reset(a); a.nbytes = st[0]; a.bufbyte = st[1]; a.zeros = st[2]; a.cnt = st[3];
    g_rl.list = (const unsigned char *)leads; g_rl.n = n; g_rl.a = a; g_rl.dst = dst;
    g_nfib = 64; g_spins = 0;
    emu_run(emu_entry_resolve);
    st[0] = g_rl.a.nbytes; st[1] = g_rl.a.bufbyte; st[2] = g_rl.a.zeros; st[3] = g_rl.a.cnt;
    return 0;
}

// The device's RDOQ (rdoq_group, hevc_core.h) on a sz x sz block of transform coefficients, group by group, with the thresholds
// the host derives for qpd6 (hevc_tables.h) staged as a frame stages them.  dst: signed levels; groups the weak-group test
// clears come back as zeros, like the reference's quantize().
template <int S>
static void emu_rdoq_groups(int q, int sz, const int *src, int *dst) {
    const QConst Q = qconst<S>(q);
    for (int gy = 0; gy < sz; gy += 4) for (int gx = 0; gx < sz; gx += 4) {
        int acc[4][4];
        for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) acc[r][c] = src[(gy + r) * sz + gx + c] * (1 << (S + 8));
        const int any = rdoq_group<S>(acc, Q);
        for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) dst[(gy + r) * sz + gx + c] = any ? acc[r][c] : 0;
    }
}
extern "C" int hostemu_rdoq_block(int q, int sz, const int *src, int *dst) {
    static Tables T; static ColdTables K; static int ready = 0;
    if (!ready) { imcvt::build_tables(T, K); ready = 1; }
    static Shm *shm = (Shm *)calloc(1, sizeof(Shm));
    g_shm_host = shm;
    memcpy(shm->rthr, K.rthr[q], sizeof(shm->rthr));
    if (sz == 4) emu_rdoq_groups<0>(q, sz, src, dst); else if (sz == 8) emu_rdoq_groups<1>(q, sz, src, dst);
    else if (sz == 16) emu_rdoq_groups<2>(q, sz, src, dst); else if (sz == 32) emu_rdoq_groups<3>(q, sz, src, dst); else return -1;
    return 0;
}
extern "C" int hostemu_rdoq_threshold(int q, int s, int cls) {
    static Tables T; static ColdTables K; static int ready = 0;
    if (!ready) { imcvt::build_tables(T, K); ready = 1; }
    return K.rthr[q][s][cls];
}
