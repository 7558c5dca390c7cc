// tests/hostemu/hostemu.cpp — TEST-ONLY harness: compiles the device encoder source (imcvt_amd/csrc/hevc_core.h,
// hevc_frame.h) for the host.  Every lane of the 192-thread workgroup is a cooperative fiber with its own stack, so
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

#define EMU_THREADS 192
#define EMU_STACK (512 * 1024)
struct EmuFiber {
    void *sp; char *stack; int done;
    volatile unsigned *wait_gen; unsigned wait_val;        // blocked until *wait_gen != wait_val
};
static EmuFiber g_fib[EMU_THREADS];
static void *g_main_sp;
static int g_cur;                                           // running fiber (= threadIdx.x)
static unsigned g_wave_gen[3], g_wave_arr[3], g_wg_gen, g_wg_arr;
static uint64_t g_xchg[EMU_THREADS];                        // collective exchange slots
static void (*g_entry)(void);

static int emu_lane() { return g_cur & 63; }
static int emu_wave() { return g_cur >> 6; }
static void emu_block(volatile unsigned *gen, unsigned val) {
    EmuFiber &f = g_fib[g_cur];
    f.wait_gen = gen; f.wait_val = val;
    emu_ctx_switch(&f.sp, g_main_sp);
}
static void emu_wave_sync() {
    const int w = emu_wave();
    if (++g_wave_arr[w] == 64) { g_wave_arr[w] = 0; g_wave_gen[w]++; }
    else emu_block(&g_wave_gen[w], g_wave_gen[w]);
}
static void emu_wg_sync() {
    if (++g_wg_arr == EMU_THREADS) { g_wg_arr = 0; g_wg_gen++; }
    else emu_block(&g_wg_gen, g_wg_gen);
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
static void emu_trampoline() {
    g_entry();
    g_fib[g_cur].done = 1;
    for (;;) emu_ctx_switch(&g_fib[g_cur].sp, g_main_sp);
}
static void emu_run(void (*entry)(void)) {
    g_entry = entry;
    memset(g_wave_gen, 0, sizeof g_wave_gen); memset(g_wave_arr, 0, sizeof g_wave_arr); g_wg_gen = g_wg_arr = 0;
    for (int i = 0; i < EMU_THREADS; i++) {
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
        int live = 0, ran = 0;
        for (int i = 0; i < EMU_THREADS; i++) {
            EmuFiber &f = g_fib[i];
            if (f.done) continue;
            live++;
            if (f.wait_gen && *f.wait_gen == f.wait_val) continue;
            f.wait_gen = nullptr; g_cur = i; ran++;
            emu_ctx_switch(&g_main_sp, f.sp);
        }
        if (!live) break;
        if (!ran) { fprintf(stderr, "hostemu: deadlock (divergent barrier)\n"); abort(); }
    }
}

#include "../../imcvt_amd/csrc/hevc_frame.h"
#include "../../imcvt_amd/csrc/hevc_tables.h"

static struct { const Tables *T; const ColdTables *K; FrameJob job; Scratch sc; const u8 *hdr; } g_args;
static void emu_entry() { encode_frame(g_args.T, g_args.K, g_args.job, g_args.sc, g_args.hdr); }

extern "C" int hostemu_HEVCImageEncoder(unsigned char *pbuffer, const unsigned char *img, unsigned char *img_rcon,
                                        int *ysz, int *xsz, int qpd6, int *trace, int trace_cap) {
    static Tables T; static ColdTables K; static int ready = 0;
    if (!ready) { imcvt::build_tables(T, K); ready = 1; }
    const int h = *ysz, w = *xsz;
    const int hp = ((h < 8192 ? h : 8192) + 31) / 32 * 32, wp = ((w < 8192 ? w : 8192) + 31) / 32 * 32;
    Shm *S = (Shm *)calloc(1, sizeof(Shm));
    Scratch sc;
    void *pool = calloc(1, scratch_bytes_per_wg());
    scratch_carve(sc, (u8 *)pool);
    sc.trace = trace; sc.trace_cap = trace_cap;
    u8 hdr[96];
    FrameJob job;
    int out_len = 0;
    job.img = img; job.out = pbuffer; job.rcon = img_rcon; job.h = h; job.w = w; job.hp = hp; job.wp = wp; job.q = qpd6;
    job.hdr_len = imcvt::build_headers(hdr, qpd6, hp, wp);
    job.out_len = &out_len;
    g_shm_host = S; sc.prof = nullptr;
    g_args.T = &T; g_args.K = &K; g_args.job = job; g_args.sc = sc; g_args.hdr = hdr;
    emu_run(emu_entry);
    free(pool); free(S);
    *ysz = hp; *xsz = wp;
    return out_len;
}
extern "C" int hostemu_shm_bytes(void) { return (int)sizeof(Shm); }
