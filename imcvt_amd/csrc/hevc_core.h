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
#include <stddef.h>

typedef uint8_t u8;  typedef uint16_t u16; typedef uint32_t u32; typedef uint64_t u64;
typedef int8_t i8;   typedef int16_t i16;  typedef int32_t i32;
typedef u32 __attribute__((may_alias)) u32a;      // a dword view of data that is also accessed as bytes / 16-bit tokens

#define NWAVES 3
#define WG_THREADS (NWAVES * 64)
// Optional fourth wavefront ("pipe wave", hevc_frame.h nxn_pipe): launches of 256 threads per workgroup carry one; it takes the
// NxN trial of every 8x8 CU off the wave that walks the PU chain.  It joins the workgroup barriers and skips everything else.
#define PIPE_WAVE NWAVES
#define WG_THREADS_PIPE ((NWAVES + 1) * 64)
// Wide workgroups (round 5; launches that leave every main workgroup a compute unit of its own): 512 threads.  Wavefronts 4..7 are the
// PARTNERS of wavefronts 0..3 — wave 4 + i runs the byte half of every trial coder whose range half wave i runs (stream_seg_R / stream_seg_L
// below): a lone wavefront issues one instruction per ~4.6 cycles whatever the instruction (tools/valu_rate_probe.hip), so a chain gets
// shorter only by putting fewer instructions on each wavefront.  Like the pipe wave they join every workgroup barrier and skip the rest.
#define XWAVES 4
#define WG_THREADS_WIDE ((NWAVES + 1 + XWAVES) * 64)
#define NMODE 35
#define I32MAX 0x7fffffff
#define REG_N 96           // region counters of -DIMCVT_REGCNT builds (RCNT below)

#ifdef IMCVT_HOSTEMU
  struct uint2 { uint32_t x, y; }; struct int4 { int32_t x, y, z, w; };
  #define HD static inline
  #define HDN static
  // every lane is a cooperative fiber (tests/hostemu/hostemu.cpp): real SIMT semantics, real barriers
  static int emu_lane(); static int emu_wave(); static void emu_wave_sync(); static void emu_wg_sync();
  static uint64_t emu_ballot(int p); static int emu_shfl(int v, int src_lane);
  #define LANES(l) for (int l = emu_lane(), l##_once = 1; l##_once; l##_once = 0)
  #define WAVES(w) for (int w = emu_wave(), w##_once = (w < NWAVES); w##_once; w##_once = 0)      // the pipe wave sits these out
  #define WAVES_ALL(w) for (int w = emu_wave(), w##_once = 1; w##_once; w##_once = 0)
  HD void wave_sync() { emu_wave_sync(); }
  HD void wave_sync_lds() { emu_wave_sync(); }
  HD void wg_sync() { emu_wg_sync(); }
  HD i32 lds_add(i32 *p, i32 v) { i32 o = *p; *p += v; return o; }
  HD i32 lds_max(i32 *p, i32 v) { i32 o = *p; if (v > o) *p = v; return o; }
  HD u32 lds_or(u32 *p, u32 v) { u32 o = *p; *p |= v; return o; }
  HD int clz32(u32 v) { return v ? __builtin_clz(v) : 32; }
  HD u64 wave_ballot(int p) { return emu_ballot(p); }
  HD int wave_shfl(int v, int src_lane) { return emu_shfl(v, src_lane); }
  // cross-workgroup mail (teams): the emulated workgroups are fibers of one thread, so plain accesses are coherent
  static void emu_yield();
  HD u32 m_ld32(const void *p) { return *(const volatile u32 *)p; }
  HD void m_st32(void *p, u32 v) { *(volatile u32 *)p = v; }
  HD u32 m_add32(void *p, u32 v) { const u32 o = *(volatile u32 *)p; *(volatile u32 *)p = o + v; return o; }
  HD int m_cas32(void *p, u32 expect, u32 desired) { if (*(volatile u32 *)p != expect) return 0; *(volatile u32 *)p = desired; return 1; }
  HD void mail_poll_pause() { emu_yield(); }
  HD void mail_idle_pause(int) { emu_yield(); }
  // flags between the wavefronts of one workgroup (pipe wave): LDS words, polled
  HD i32 lds_ld_i32(const i32 *p) { return *(const volatile i32 *)p; }
  HD void lds_st_i32(i32 *p, i32 v) { *(volatile i32 *)p = v; }
  HD void pipe_pause() { emu_yield(); }
  static int emu_pipe_on(); static int emu_wide_on(); static int emu_late_main();
  HD int wg_has_pipe_wave() { return emu_pipe_on(); }
  HD int wg_is_wide() { return emu_wide_on(); }
  HD unsigned long long wd_now() { return 0; }      // (the emulation has its own deadlock detector)
  HD void drain_stores() {}
  HD void sys_release() {}
  HD void sys_st32(void *p, u32 v) { *(volatile u32 *)p = v; }
#else
  #define HD __device__ __forceinline__
  // Out-of-line device functions.  not_tail_called keeps LLVM's `tail` marker off their call sites; with the marker on any call site
  // the AMDGPU backend does not apply its no-callee-saved-registers optimisation to an internal function, and the candidate-set
  // functions (168 registers each, nothing live in their callers) then open with 67 scratch stores and close with 67 loads per call.
  #ifndef HDN
  #define HDN __device__ __noinline__ __attribute__((not_tail_called))
  #endif
  #define LANES(l) for (int l = (int)(threadIdx.x & 63u), l##_once = 1; l##_once; l##_once = 0)
  #define WAVES(w) for (int w = (int)(threadIdx.x >> 6), w##_once = (w < NWAVES); w##_once; w##_once = 0)      // the pipe wave sits these out
  #define WAVES_ALL(w) for (int w = (int)(threadIdx.x >> 6), w##_once = 1; w##_once; w##_once = 0)
  // LDS traffic of one wavefront is in program order; the fence only stops the compiler (and drains
  // global stores, which the trial coders read back from other lanes of the same wave).
  HD void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); }
  // same, but only LDS traffic is ordered (global stores stay in flight)
  HD void wave_sync_lds() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup", "local"); __builtin_amdgcn_wave_barrier(); }
  HD void wg_sync() { __syncthreads(); }
  HD u64 wave_ballot(int p) { return __builtin_amdgcn_ballot_w64(p != 0); }
  HD int wave_shfl(int v, int src_lane) { return __shfl(v, src_lane, 64); }
  HD i32 lds_add(i32 *p, i32 v) { return atomicAdd(p, v); }
  HD i32 lds_max(i32 *p, i32 v) { return atomicMax(p, v); }
  HD u32 lds_or(u32 *p, u32 v) { return atomicOr(p, v); }
  HD int clz32(u32 v) { return __clz((int)v); }
  // Cross-workgroup mail (teams, hevc_frame.h).  Per-XCD L2s are not coherent with each other for ordinary accesses and a
  // CU's L1 is never refreshed by another CU's stores, so everything that crosses workgroups — payload and flags alike — is
  // written and read with agent-scope accesses (global_store / global_load ... sc1: write-through, L1 bypassed, coherent
  // across the device).  Order: the producer's payload stores are drained (vmcnt(0) in every wave, then the workgroup
  // barrier) before one lane stores the flag; the consumer sees the flag, passes a barrier, and only then loads the
  // payload.  No L2 write-back / L1 invalidate is involved: with hundreds of workgroups per XCD keeping megabytes of
  // dirty token scratch in L2, an agent-scope release (buffer_wbl2) per hand-off was measured to halve the throughput.
#ifndef MAIL_POLL_SLEEP
#define MAIL_POLL_SLEEP 32      // x 64 cycles between polls (~0.9 us): a hop is microseconds, a CTU milliseconds; hundreds of waves polling faster load the fabric
#endif
  HD u32 m_ld32(const void *p) { return __hip_atomic_load((const u32 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  HD void m_st32(void *p, u32 v) { __hip_atomic_store((u32 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  HD u32 m_add32(void *p, u32 v) { return __hip_atomic_fetch_add((u32 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  HD int m_cas32(void *p, u32 expect, u32 desired) { return __hip_atomic_compare_exchange_strong((u32 *)p, &expect, desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  HD void mail_poll_pause() { __builtin_amdgcn_s_sleep(MAIL_POLL_SLEEP); }
  // an idle helper backs off (round r of an unsuccessful poll): ~0.9 us doubling to ~7 us, so that hundreds of idle helpers do not hammer the queue words
  HD void mail_idle_pause(int r) { const int n = r < 3 ? 1 << r : 8; for (int i = 0; i < n; i++) __builtin_amdgcn_s_sleep(MAIL_POLL_SLEEP); }
  HD void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  // what this compute unit's waves stored so far becomes visible OUTSIDE the device's caches (a copy engine reading HBM while the kernel runs): a
  // system-scope release writes the XCD's L2 back.  Used once per CTU row of a frame whose progress the host follows, never on the mail path.
  HD void sys_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, ""); }
  HD void sys_st32(void *p, u32 v) { __hip_atomic_store((u32 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
  HD unsigned long long wd_now() { return wall_clock64(); }
  // flags between the wavefronts of one workgroup (pipe wave): LDS words, polled.  The wavefronts of a workgroup share a compute
  // unit and its vector L1, so a producer's global stores are visible to the consumer once they have been issued and waited for
  // (wave_sync() before the flag store), exactly as across a workgroup barrier.
  HD i32 lds_ld_i32(const i32 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
  HD void lds_st_i32(i32 *p, i32 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
  HD void pipe_pause() { __builtin_amdgcn_s_sleep(2); }
  HD int wg_has_pipe_wave() { return blockDim.x > (unsigned)WG_THREADS; }
  HD int wg_is_wide() { return blockDim.x >= (unsigned)WG_THREADS_WIDE; }
#endif

#if defined(IMCVT_HOSTEMU)
#define SCHED_FENCE() do {} while (0)
#define NOUNROLL
#else
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#ifdef IMCVT_NO_PRIO
#define SETPRIO(x) do {} while (0)
#else
#define SETPRIO(x) __builtin_amdgcn_s_setprio(x)
#endif
#define NOUNROLL _Pragma("unroll 1")
#endif
struct alignas(16) U4 { u32 x, y, z, w; };
// Global-memory accessors.  Pointers that come out of structs are generic ("flat") to the compiler, and FLAT
// instructions count against lgkmcnt — every LDS wait would then also wait for a global round trip.
// These force address space 1 (global_* instructions, vmcnt only).
#ifdef IMCVT_HOSTEMU
HD void g_st8(u8 *p, int v) { *p = (u8)v; }
HD void g_st16(i16 *p, int v) { *p = (i16)v; }
HD void g_st32(void *p, u32 v) { *(u32a *)p = v; }
HD u8 g_ld8(const u8 *p) { return *p; }
HD u32 g_ld32(const void *p) { return *(const u32a *)p; }
HD i16 g_ld16(const i16 *p) { return *p; }
HD U4 g_ld128(const void *p) { return *(const U4 *)p; }
HD void g_st128(void *p, const U4 &v) { *(U4 *)p = v; }
HD u8 *uniform_ptr(u8 *p) { return p; }
HD int uni_i(int v) { return v; }
template <class T_> HD T_ *uni_p(T_ *p) { return p; }
HD int hibit(u32 v) { return 31 - __builtin_clz(v); }
HD int clz_nz(u32 v) { return v ? __builtin_clz(v) : 32; }
HD int popc32(u32 v) { return __builtin_popcount(v); }
HD int ctz64(u64 v) { return __builtin_ctzll(v); }
#else
#define GAS __attribute__((address_space(1)))
HD void g_st8(u8 *p, int v) { *(GAS u8 *)p = (u8)v; }
HD void g_st16(i16 *p, int v) { *(GAS i16 *)p = (i16)v; }
HD void g_st32(void *p, u32 v) { *(GAS u32 *)p = v; }
HD u8 g_ld8(const u8 *p) { return *(const GAS u8 *)p; }
HD u32 g_ld32(const void *p) { return *(const GAS u32 *)p; }
HD i16 g_ld16(const i16 *p) { return *(const GAS i16 *)p; }
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
HD U4 g_ld128(const void *p) { const u32x4 v = *(const GAS u32x4 *)p; U4 r; r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w; return r; }
HD void g_st128(void *p, const U4 &v) { u32x4 t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w; *(GAS u32x4 *)p = t; }
#endif
#ifndef IMCVT_HOSTEMU
// a pointer every lane agrees on, moved to SGPRs so that stores can use the scalar-base + 32-bit-offset form
HD u8 *uniform_ptr(u8 *p) {
    const u64 v = (u64)p;
    const u32 lo = __builtin_amdgcn_readfirstlane((u32)v), hi = __builtin_amdgcn_readfirstlane((u32)(v >> 32));
    return (u8 *)(((u64)hi << 32) | lo);
}
// arguments of out-of-line functions arrive in vector registers; these are wave-uniform by construction: moved to scalar registers,
// so that what is computed from them stays scalar and branches on them are scalar branches
HD int uni_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <class T_> HD T_ *uni_p(T_ *p) {
    const u64 v = (u64)p;
    const u32 lo = __builtin_amdgcn_readfirstlane((u32)v), hi = __builtin_amdgcn_readfirstlane((u32)(v >> 32));
    return (T_ *)(((u64)hi << 32) | lo);
}
// the compute unit this wave runs on, as a key below POOL_CU_KEYS: XCC_ID[3:0] | HW_ID.se_id[15:13] | sh_id[12] | cu_id[11:8]
HD int hw_cu_key() {
    const u32 hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // hwreg(HW_REG_HW_ID), hwreg(HW_REG_XCC_ID)
    return (int)(((xcc & 15u) << 8) | ((hw >> 8) & 255u));
}
HD int hibit(u32 v) { return 31 - __clz((int)v); }
HD int clz_nz(u32 v) { return __builtin_clz(v); }          // v != 0 where the result is used
HD int popc32(u32 v) { return __popc(v); }
HD int ctz64(u64 v) { return __builtin_ctzll(v); }
#endif
// 24-bit multiplies are full rate on the VALU; v_mul_lo_u32 is not.  Only used where both operands provably fit.
#ifdef IMCVT_HOSTEMU
HD int mul24(int a, int b) { return a * b; }
HD int umul24(int a, int b) { return a * b; }
#else
HD int mul24(int a, int b) { return __mul24(a, b); }
HD int umul24(int a, int b) { return (int)__umul24((u32)a, (u32)b); }      // both operands in [0, 2^24)
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
    u8  cgpos_d[4][64];    // diagonal: [log2N-2][g] -> (gy<<3)|gx, coefficient-group scan order           (:1126-1150)
    u8  cgrank_d[4][64];   // inverse: [s][gy*8+gx] -> g
    u8  cgpos_hv[2][4];    // horizontal / vertical group order of an 8x8 TU (4 groups); 4x4 TUs have one group
    u8  cgrank_hv[2][16];  // inverse, indexed gy*8+gx (only 0,1,8,9 used)
    u8  incg[3][16];       // [type][n] -> (yi<<2)|xi inside a 4x4 group
    u8  incg_rank[3][16];  // inverse: [type][yi*4+xi] -> n
    uint2 pst[128];        // per packed state p: .x = the 4 LPS ranges, .y = nextLPS | nextMPS<<8 | p<<16        (:700-712)
    u32 posadd[4][3];      // sig_coeff ctx increment per in-group scan position, 2 bits each [pattern][type]  (:1115-1120)
    u64 c4tab[3];          // 4x4-TU sig_coeff ctx per scan position, 4 bits each [type]                  (:1092)
    u32 c4prev[3][4];      // [type]: per scan position n of a 4x4 TU (a byte each) the scan positions above n that share its sig_coeff context: nearest | next << 4 (0: none)
    u8  ang[36];           // intraPredAngle + 32                                                         (:282)
    u16 iang[36];          // |invAngle|                                                                  (:283)
};
// tables that stay in global memory (read once per frame)
#define RQ_CLASSES 10
#define PU_SIG_N 72
#define PU_GT_N 136
struct ColdTables {
    u8 ctx_init[5][CTX_STRIDE];            // initial context states per qpd6 (:726-784)
    i32 rthr[5][4][RQ_CLASSES];            // RDOQ decision thresholds per qpd6 and TU size (rdoq_group below); a frame stages its qpd6's 40 words into LDS
    u8 pu_sig[5][PU_SIG_N];                // state hints of the 4x4 PU candidates (tokg_a_fast<0, true>), per qpd6: sig_coeff context f after no / one / two earlier bins: [8 f + idx]
    u8 pu_gt[5][PU_GT_N];                  // greater-1 contexts: [j] the j-th flag while no level above 1 was seen; [8 + (1 << n) - 1 + pattern] context 0 after n earlier bins
};
HD int mat_off(int s) { return s == 0 ? 0 : s == 1 ? 16 : s == 2 ? 80 : 336; }

// ---------------------------------------------------------------------------------------------------
// Arithmetic coder state (:796-805) — `cnt` counts bytes of the current CTU already pushed.
// ---------------------------------------------------------------------------------------------------
enum { PF_BORDER = 0, PF_P1_32, PF_P1_16, PF_P1_8, PF_P1_4, PF_P2_32, PF_P2_16, PF_P2_8, PF_P2_PU, PF_P2_NXN, PF_SYNC, PF_DECIDE, PF_RECON, PF_CTUIO, PF_T_SETUP, PF_T_HDR, PF_T_GEN, PF_T_DRAIN, PF_T_NDRAIN, PF_T_NTOK, PF_X1, PF_X2, PF_X3, PF_N };
struct Arith { i32 range, low, nbits, nbytes, bufbyte, zeros, cnt; };
HD void arith_reset(Arith &a) { a.range = 510; a.low = 0; a.nbits = 23; a.nbytes = 0; a.bufbyte = 0xFF; a.zeros = 0; a.cnt = 0; }
HD int arith_len(const Arith &a) { return 8 * (a.cnt + a.nbytes) + 23 - a.nbits; }      // :834

// ---------------------------------------------------------------------------------------------------
// Workgroup memory
// ---------------------------------------------------------------------------------------------------
#define RS 68            // reconstruction tile stride: 1 border column + 64, padded

struct Border {          // prediction references of one block (:196-257): unfiltered / [1 2 1]-filtered
    u8 uc, fc; i16 dc;
    u8 ul[68], ua[68], fl[68], fa[68];
};
struct BorderS {         // per-mode border of a block <= 16 (four-TU shape): only the variant (filtered or not) that mode uses
    u8 c, pad_; i16 dc;
    u8 l[36], a[36];
};

struct FinState { u32 w0, w1, w2; };   // packed Arith: low | range,nbits,zeros(sat 2),bufbyte | nbytes,cnt
HD FinState pack_arith(const Arith &a) {
    FinState f; f.w0 = (u32)a.low;
    f.w1 = (u32)a.range | (u32)a.nbits << 10 | (u32)imin(a.zeros, 2) << 16 | (u32)a.bufbyte << 18;
    f.w2 = (u32)a.nbytes | (u32)a.cnt << 16; return f;
}
HD Arith unpack_arith(const FinState &f) {
    Arith a; a.low = (i32)f.w0; a.range = (i32)(f.w1 & 1023); a.nbits = (i32)((f.w1 >> 10) & 63); a.zeros = (i32)((f.w1 >> 16) & 3);
    a.bufbyte = (i32)(f.w1 >> 18); a.nbytes = (i32)(f.w2 & 0xFFFF); a.cnt = (i32)(f.w2 >> 16); return a;
}

#define LRING 16           // per-lane ring of byte leads of the trial coders (LeadSink): at most 8 leads join between two flushes of 8
struct LaneMem { u16 ring[LRING]; u16 pad_[10]; };   // 52 bytes = 13 dwords: odd stride, lanes hit different LDS banks
#define LSTRIDE_DW 33       // dwords per lane row of the lane-private token staging (LSTRIDE below)
#define P1_RES_BYTES 2304   // 16 tiles of 8x8 + 4 or 4 tiles of 16x16 + 16 i16 (padded against LDS bank conflicts), 16-byte multiple
#define W2_PAD (((NMODE * CTX_STRIDE + 15) & ~15) + ((NMODE * (int)sizeof(LaneMem) + 15) & ~15))      // p2's extent
struct alignas(16) WaveMem {
    Border  bsh;                 // border shared by all modes of a block
    i32 tokn[NMODE + 1];         // tokens written so far to each candidate's stream (slot NMODE: the NxN stream)
    u8  tnz[NMODE + 1];          // the TU tokenised last has a non-zero level
    i32 sse[NMODE];
    i32 cost[NMODE];
    FinState fin[NMODE];         // coder state each trial ended in
    i32 pu_mode[4], pu_sse[4], pu_cnt[4];   // NxN bookkeeping (PU wave)
    i32 nxn_cost;
    alignas(16) u16 pend[NMODE + 1][8];     // the partial last 8-token block of each candidate's stream (rest: idle tokens)
    union alignas(16) {          // MUST stay last: the 4x4-only wave's slice is truncated after `w2`
        struct { i16 res[P1_RES_BYTES / 2]; i32 tmp[(7168 - P1_RES_BYTES) / 4]; } p1;                    // one pipeline pass: residual / dequantised tiles, stage outputs (tile strides: p1_run_t)
        u32 raw[1792];                                                                                // per-lane token staging (4x4 blocks, CU headers): lane l at raw + 33 l
        struct { u8 cx[NMODE][CTX_STRIDE]; alignas(16) LaneMem lm[NMODE]; } p2;                          // trial coders: context copies, lead rings
        struct { u8 pad_[W2_PAD]; u8 rec4[NMODE][16]; } w2;                                             // 4x4 PU candidates' reconstructions (beside p2)
    } u;
};
#define WAVE2_BYTES (sizeof(WaveMem) - 7168 + W2_PAD + NMODE * 16)

#define TRIAL_OUT_BYTES 3584      // bytes one trial may emit: the reference's own per-CTU coder buffer is TMPBUF_LEN = 3200 (:794), unchecked there too
static_assert(TRIAL_OUT_BYTES >= 3200 + 256, "a trial's byte buffer must cover the reference's per-CTU coder buffer plus the 16-byte flush granularity");
#define TRIAL_BYTES (2 * TRIAL_OUT_BYTES)      // a trial coder's buffer: it leaves the 9-bit LEAD of every byte (u16, LeadSink), not the byte; its last word holds the number of leads
// Per-frame job and per-workgroup scratch (global memory)
struct FrameJob {
    const u8 *img;   // h*w gray8
    u8 *out;         // stream buffer
    u8 *rcon;        // hp*wp reconstruction
    i32 h, w, hp, wp, q;
    i32 hdr_len;     // header bytes already placed at out[0..hdr_len)
    i32 *out_len;    // result
    u32 *prog;       // optional progress record of this frame, two words the host may read WHILE the launch runs (imcvt_hevc_set_progress; pinned host or device memory), or null:
                     // [0] CTU rows whose reconstruction is final in rcon (| PROG_DONE once the frame is finished), [1] stream bytes that are final in out
};
#define PROG_DONE 0x80000000u
#define TOK_CAP 7040             // u16 per candidate stream: 18 (CU header) + 4 x 25 (cbf + last position per TU) + 64 groups x 108, rounded to 16 bytes
#define TOK_SLOTS (NMODE + 1)
struct Scratch {
    u16 *tok;        // [NWAVES][TOK_SLOTS][TOK_CAP] bin tokens of the candidates being priced
    u8  *bytes;      // [NWAVES + 1][NMODE][TRIAL_BYTES] bytes emitted by trial coders (the last block: the pipe wave's)
    u8  *above_sz;   // [wp/4] CU sizes of the CTU row above (:1633-1636)
    i32 *trace;      // optional decision trace (8 ints per CU), or null
    unsigned long long *prof;   // optional [NWAVES][PF_N] cycle totals (IMCVT_PROF builds), or null
    i32 trace_cap;
};
// host side: size and carving of one workgroup's scratch slab
static inline size_t scratch_align(size_t v) { return (v + 255) & ~(size_t)255; }
static inline size_t scratch_tok_bytes() { return scratch_align((size_t)NWAVES * TOK_SLOTS * TOK_CAP * sizeof(u16) + 64); }
static inline size_t scratch_bytes_per_wg() { return scratch_tok_bytes() + scratch_align((size_t)(NWAVES + 1) * NMODE * TRIAL_BYTES) + scratch_align(8192 / 4 + 64); }
static inline void scratch_carve(Scratch &sc, u8 *base) {
    sc.tok = (u16 *)base; base += scratch_tok_bytes();
    sc.bytes = base;      base += scratch_align((size_t)(NWAVES + 1) * NMODE * TRIAL_BYTES);
    sc.above_sz = base;
    sc.trace = nullptr; sc.trace_cap = 0; sc.prof = nullptr;
}

// ---------------------------------------------------------------------------------------------------
// Teams: a frame encoded by several cooperating workgroups (hevc_frame.h).  The candidate sets of a 16x16 / 32x32 CU
// start from the coder state at the CU's ENTRY (:1363-1364, :1422, :1455) and predict from samples outside the CU
// (:196-257), so they do not depend on the CU's children: helper workgroups evaluate them while the main workgroup walks
// the 8x8 CUs.  Requests and results travel through one mailbox per request kind in global memory.
// ---------------------------------------------------------------------------------------------------
// SLOT_16 / SLOT_32: requests that go through the pool's queues (POOL_KINDS of them).  SLOT_8 (round 6, wide launches): the mailbox between a main workgroup and
// its PARTNER workgroup — the two 2Nx2N candidate sets of every 8x8 CU on a second compute unit (hevc_frame.h "8x8 CUs with a partner workgroup"); no queue, the
// partner serves one main workgroup.
enum { SLOT_16 = 0, SLOT_32 = 1, POOL_KINDS = 2, SLOT_8 = 2, MAIL_SLOTS = 3 };
enum { OP_WORK = 0, OP_EXIT = 1 };
struct alignas(16) HelpReq {                     // main -> helper: everything the candidate sets of one CU start from
    i32 op, frame, cy, cx;                       // job index, CTU origin
    i32 N, y0, x0, avm;                          // the CU inside the CTU, neighbour availability
    i32 szl, sza, ml, ma;                        // CU size / mode of the left and above neighbour cells (split flag context, MPM)
    Arith a; i32 seq;                            // coder state at the CU's entry; sequence number the answer is published under
    alignas(4) u8 ctx[CTX_STRIDE];               // contexts at the CU's entry
    alignas(4) u8 above[68];                     // reconstructed samples the candidates predict from (:196-257): the row above the CU from
    alignas(4) u8 left[64];                      // x0-1 to x0+2N-1 (corner first), and the column left of it from y0 to y0+2N-1
};
struct alignas(16) HelpRes {                     // helper -> main: the best unsplit candidate ("last minimum" of the 70)
    i32 cost, kind, mode, nbytes;                // kind 1: one TU, 2: four TUs; nbytes: bytes the winning trial emitted
    FinState fin; i32 pad_;                      // coder state the winning trial ended in
    alignas(4) u8 ctx[CTX_STRIDE];               // its contexts
    alignas(16) u8 rec[1024];                    // its reconstruction, N x N row-major
    alignas(16) u8 bytes[TRIAL_OUT_BYTES];       // its bytes
};
struct alignas(256) MailSlot {
    i32 req_flag; i32 pad0_[63];                 // sequence number of the request in `req` (flags on their own lines)
    i32 res_flag; i32 pad1_[63];                 // sequence number of the result in `res`
    HelpReq req;
    HelpRes res;
};
struct TeamMail { MailSlot s[MAIL_SLOTS]; };
// Request queues of a launch: the helper workgroups form ONE pool that serves the requests of every main workgroup.  A queue
// is a ticket ring per request kind: a main workgroup takes ticket t = tail++ and publishes (t, its index) in ring[t]; a helper
// claims ticket h = head++ (compare-and-swap while head < tail) and waits for ring[h] to name ticket h.  A main workgroup has at
// most one request of a kind outstanding, so a ring of POOL_QCAP entries only wraps onto an entry whose claimer has been held up
// for a whole lap — that claimer sees the later ticket in its slot and goes back to polling (hevc_frame.h ring_entry).  The queue is cut into
// POOL_SHARDS shards on their own cache lines (main workgroup i posts to shard i mod POOL_SHARDS; a helper looks at its home
// shard first and then at one other shard per poll), so that hundreds of polling workgroups do not meet on one line.
#define POOL_SHARDS 16
// A main workgroup waits this long for an answer (100 MHz ticks; answers take 0.3 - 2 ms, queueing included), then evaluates the CU
// itself and leaves the request to arrive whenever it does: a helper can be held up for seconds by things outside this code
// (wave preemption on a full device, profiles/r03q_hb_probe.log).  The host emulation has no clock and counts polls instead.
#ifndef ABANDON_TICKS
#define ABANDON_TICKS 1000000ull
#endif
#ifndef ABANDON_POLLS
#define ABANDON_POLLS (1 << 30)
#endif
#ifndef WD_TICKS
#define WD_TICKS 2000000000ull   // 20 s of the 100 MHz clock: no wait between workgroups comes near (a request is served in ~1 ms)
#endif
#define POOL_CU_KEYS 4096     // XCC (4 bits) | shader engine (3) | shader array (1) | CU (4)
#define POOL_QCAP 256        // per shard and kind: >= 2 x the main workgroups that share a shard (POOL_SHARDS x POOL_QCAP / 2 = 2048 mains)
struct alignas(256) PoolShard {
    u32 head[POOL_KINDS], tail[POOL_KINDS];      // tickets claimed / issued, per request kind
    u32 pad_[60];
    u32 ring[POOL_KINDS][POOL_QCAP];             // ticket -> (ticket mod 2^20) << 12 | main workgroup index + 1 (0: nothing published yet)
};
struct alignas(256) PoolQ {
    u32 frames_done;                             // frames finished: helpers leave when all are (no request can follow)
    u32 alive;                                   // helper workgroups that have started: requests are only posted once there is one to serve them
    u32 mains_taken;                             // main-workgroup indices handed out so far
    u32 progress;                                // sum over the main workgroups of the share of their frames they have finished, in 1/65536 frames (pace control)
    u32 abort;                                   // watchdog: a wait between workgroups exceeded WD_TICKS — every wait gives up, every workgroup leaves (host: IMCVT_ERR_WATCHDOG)
    u32 dbg[8];                                  // what the wait that gave up was waiting for
    u32 pad_[8];                                 // (the watchdog's clocks, hevc_frame.h wd_poll)
    u32 parts_taken;                             // partner-workgroup indices handed out so far (partner i serves main workgroup i)
    u32 pad2_[42];
    PoolShard sh[POOL_SHARDS];
    u32 cu_count[POOL_CU_KEYS];                  // workgroups of this launch that have started on each compute unit (role choice, hevc_frame.h kernel_main)
};
// A pool launch is so old that a main-workgroup index which is still free belongs to a workgroup the dispatcher will hold back until another one leaves — seconds (launches that
// fill every slot) — and a running helper takes it (hevc_frame.h helper_loop; `start_lo`: low word of the launch's earliest start, 100 MHz).  Not sooner: the first full launch on
// a fresh queue starts a good part of its workgroups up to a second late (the private-segment ring grows), and a helper that takes over is a workgroup from the end of the
// dispatch order, which runs a frame 1.1 - 1.9 x slower than the workgroup it stands in for would — with a limit of 2 ms the first host-pointer batch of a
// process took 8.6 s instead of 5.6 s (profiles/r06fin2_bench_512f.json).
#ifndef ROLE_GRACE_TICKS
#define ROLE_GRACE_TICKS 30000u          // 300 us (100 MHz): how long a later block waits before it claims a main-workgroup index the first blocks have left (hevc_frame.h kernel_main) — the shader engines hand their blocks out independently over ~40 us; 30 us let late blocks of a fast engine claim before early blocks of a slow one (profiles/r06zd_roles_default.log)
#endif
#ifndef LATE_MAIN_TICKS
#define LATE_MAIN_TICKS 150000000u     // 1.5 s
#endif
#ifdef IMCVT_HOSTEMU
HD int pool_cu_key(int home) { return home; }            // (the emulated workgroups have no compute units: a queue shard stands in — two helpers of one shard do not both take the path)
HD int late_main_due(const int *, int home) { return emu_late_main() && (home & 1); }      // (tests: HOSTEMU_LATE_MAIN=1 — the helpers of odd shards take the new path, the others the idle one)
#else
HD int pool_cu_key(int) { return hw_cu_key() % POOL_CU_KEYS; }
HD int late_main_due(const int *counter, int) { const u32 t0 = m_ld32(counter + 4); return t0 != 0xFFFFFFFFu && (u32)((u32)wd_now() - t0) > LATE_MAIN_TICKS; }
#endif
struct FrameCtx {
    FrameJob job;
    Scratch sc;
    i32 out_pos;        // bytes of finished CTUs (incl. headers)
    i32 ctu_y, ctu_x;   // pixel origin of the current CTU
    i32 trace_n;
    i32 frame;          // index of the job being encoded
    TeamMail *mail;     // this main workgroup's mailboxes (null: the workgroup encodes its frames alone)
    PoolQ *pq;          // the launch's request queues
    i32 main_id;        // index of this main workgroup (= of its mailboxes)
    i32 prio_base;      // wave priority of this workgroup outside its critical sections (2: main workgroup of a team, 0 otherwise)
    i32 lim[POOL_KINDS];   // a request of this kind is only posted while fewer than this many wait unclaimed in the workgroup's shard (else the CU is evaluated here)
    i32 post_pm[POOL_KINDS], post_acc[POOL_KINDS];   // share of the CUs of a kind that is offered to the helpers at all, per mille, and its running remainder
    i32 posted[3];      // the CU of depth 0 / 1 being walked has a request out
    i32 stale[POOL_KINDS], gaveup;   // sequence number of a request this workgroup stopped waiting for (its mailbox is not reused before that answer has arrived); the last wait gave up
    i32 kept;           // CUs of this frame evaluated here because the helpers were busy (debug statistic)
    i32 aborted;        // the launch's watchdog has fired (read once per CTU and after every wait)
#ifdef IMCVT_HB
    unsigned long long hb_last, hb_gap, hb_when;   // heartbeat (debug, -DIMCVT_HB builds): clock of the last beat, longest gap between two beats and when it began
#endif
    u32 waited, waited_max;   // 100 MHz ticks this frame's main workgroup spent waiting for answers, and the longest single wait (debug statistics)
    i32 raised;         // CTUs of this frame that ran at raised wave priority (pace control; debug statistic)
    i32 pace_inc, pace_mine, pace_n, pace_base;   // pace control: 65536 / CTUs of this frame, this workgroup's share done, main workgroups of the launch, configured base priority
    i32 seq[POOL_KINDS];   // requests posted (main) / served (helper) so far, per slot
    i32 pipe;           // this launch's workgroups carry a pipe wave (256 threads or more)
    i32 wide;           // ... and four partner wavefronts (512 threads): the trial coders of the 8x8 CUs run split over two wavefronts each
};

struct FourTU {                  // state of the four-TU shape (one wave evaluates it at a time)
    BorderS bc[NMODE];           // per-mode borders of TUs 1..3
    u8  t3row[NMODE][3][16];     // per mode: bottom row / right column of reconstructed TUs 0..2 (TU 3 has no successor)
    u8  t3col[NMODE][3][16];
};

struct alignas(16) Shm {
    alignas(16) Tables T;
    FrameCtx F;                  // per-frame context (kept in LDS so that callees read it with ds_* ops)
    alignas(16) u8 org[32][32];
    alignas(16) u8 rec[33][RS];             // rec[y+1][x+1]; row 0 / column 0 are the neighbours
    alignas(4) u8 cx[CTX_STRIDE];   // live contexts
    Arith live;
    Arith entry_a[3];            // coder + contexts on entry to the CU of depth 0/1/2
    alignas(4) u8 entry_cx[3][CTX_STRIDE];
    u8  mapsz[10][12], mapmode[10][12];   // 4x4-unit neighbour maps of this CTU with a 1-cell apron (:1591-1599)
    i32 split_cost[3];
    i32 win_kind, win_mode;      // decision broadcast
    i32 red[NWAVES];             // small reductions
    i32 next_frame;              // job index pulled from the queue
    i32 pipe_a, pipe_b;          // PU wave -> pipe wave: the winners of PUs 0..2 / of PU 3 are in place (cleared by the pipe wave)
    i32 nxn_lane;                // pipe wave: the lane that holds the NxN trial's result (= PU 3's mode)
    i32 pu0_ready, pu0_taken;    // 8x8 CU: the PU wave's pass over PU 0 is complete / the four-TU wave has taken its copy (hevc_frame.h tu0_from_pu0)
#ifdef IMCVT_REGCNT
    i32 regcnt[REG_N];           // (-DIMCVT_REGCNT: executions of the marked regions by this workgroup's wavefronts)
#endif
#ifdef IMCVT_PROF
    unsigned long long prof[NWAVES][PF_N];
    unsigned long long tl_t0;    // (-DIMCVT_PROF_TL: when the 8x8 CU being walked was entered)
#endif
    FourTU X;
    alignas(4) u8 cx0[CTX_STRIDE];       // fresh context states of this frame's qpd6 (:1505)
    i32 rthr[4][RQ_CLASSES];             // RDOQ thresholds of this frame's qpd6
    alignas(4) u8 pu_sig[PU_SIG_N]; alignas(4) u8 pu_gt[PU_GT_N];      // state hints of this frame's qpd6 (ColdTables)
    alignas(16) u8 wraw[NWAVES * sizeof(WaveMem)];   // wave slices (wave 2 runs full pipeline passes for the 16x16 / 32x32 CUs too)
};

#if !defined(IMCVT_PROF) && !defined(IMCVT_REGCNT)
static_assert(sizeof(Shm) <= 40960, "four workgroups per compute unit need an LDS image of at most 160 KB / 4");
#endif
// The workgroup's LDS image is one file-scope object, so non-inlined callees still address it with ds_* ops.
#ifdef IMCVT_HOSTEMU
static Shm *g_shm_host;
#define SM (*g_shm_host)
#else
__shared__ Shm g_shm;
#define SM g_shm
#endif
// some lane of the wave (in host emulation: this lane) needs the rare path
#ifdef IMCVT_HOSTEMU
#define WAVE_ANY(c) (c)
#else
#define WAVE_ANY(c) (__builtin_amdgcn_ballot_w64(c) != 0)
#endif
// region counters of -DIMCVT_REGCNT builds (the marked regions: MARKR / MARKQ below)
#if defined(IMCVT_REGCNT) && !defined(IMCVT_HOSTEMU)
#define RCNT(id) do { if ((threadIdx.x & 63u) == 0u) atomicAdd(&SM.regcnt[id], 1); } while (0)
#define RCNT_ANY(id) do { const int me_ = (int)(threadIdx.x & 63u); if (__builtin_amdgcn_readfirstlane(me_) == me_) atomicAdd(&SM.regcnt[id], 1); } while (0)      // inside lane-divergent code: one count per wave execution
#else
#define RCNT(id) do {} while (0)
#define RCNT_ANY(id) do {} while (0)
#endif
#define WM(w) (*(WaveMem *)wave_mem_ptr(w))
// The pipe wave's slice (a WaveMem cut off after the trial coders' extent) is dynamic LDS: only 256-thread launches pay for it.
#define PIPE_UNION_BYTES 5120
#define PIPE_LDS_BYTES ((sizeof(WaveMem) - 7168 + PIPE_UNION_BYTES + 15) & ~(size_t)15)
static_assert(NMODE * LSTRIDE_DW * 4 <= PIPE_UNION_BYTES && W2_PAD <= PIPE_UNION_BYTES, "the pipe wave's slice holds 35 token rows / the trial coders");
static_assert(W2_PAD + NMODE * 16 <= 7168, "the PU candidates' reconstructions lie beside the trial coders in the pass buffer");
#ifdef IMCVT_HOSTEMU
static u8 *g_pipe_host;
#define PM (*(WaveMem *)g_pipe_host)
#else
extern __shared__ __attribute__((aligned(16))) u8 g_pipe_lds[];
#define PM (*(WaveMem *)g_pipe_lds)
#endif
// ---- partner wavefronts (wide workgroups): record queue between the two halves of a trial coder, and the byte half's own lane memory.
// The range half (stream_seg_R: contexts, range) leaves one 32-bit record per token — what the byte half (stream_seg_L: low, bit
// position, byte output) needs of it — in a ring of QDEPTH token blocks per lane; prod / cons count the blocks written / read, per lane
// (the lanes of a wavefront move in lock-step on the device, so one counter would do there; the host emulation's lanes are fibers).
#define QDEPTH 4
#define QSTRIDE (QDEPTH * 8 + 4)                 // dwords per lane row: 16-byte aligned rows, 36 l mod 64 spreads the lanes over 16 banks
struct alignas(16) SplitQ {
    u32 rec[NMODE][QSTRIDE];
    i32 prod[NMODE], cons[NMODE];
    i32 range_out[NMODE];                        // the range each lane's coder ended with
    i32 go, mid, rdone, done;                    // generation started by the owner / whose last segment may start (pipe wave) / whose ranges are final / finished by the partner
};
struct alignas(16) PartnerMem {
    SplitQ q;
    alignas(16) LaneMem lm[NMODE];               // lead rings of the byte half
    alignas(4) u8 cx[NMODE][CTX_STRIDE];         // context copies of coders that run on the partner wavefront itself (the four-TU set's, hevc_frame.h partner_fourtu)
};
// Control words of a wide workgroup's 8x8 CUs, and the LDS slices of the two partner wavefronts that LEND themselves for pipeline passes
// (hevc_frame.h lend_passes: the one-TU candidate set's three passes of an 8x8 CU run on three wavefronts at once).
#define NLEND 3
struct alignas(16) WideCtl {
    i32 cu8;                                     // 8x8 CUs entered so far (enter_cu): the sequence numbers of their four PU steps follow from it
    i32 a_go, lend_done[NLEND];                  // one-TU set: generation whose headers are in place / finished by each lender
    i32 b_seg, b_cons;                           // four-TU set: token segments (header + TU 0, TU 1, TU 2, TU 3) complete so far / coded so far, counted over the frame
    i32 b_hand;                                  // ... and 8x8 CUs whose last segment's range half wave 4 has handed to wave 1 (= cu8 once the current CU's is)
    i32 seg_end[4][NMODE];                       // ... and where each candidate's segment ends in its stream (a segment starts on a token-block boundary)
    // 8x8 CUs with a partner workgroup (hevc_frame.h): main workgroup
    i32 part_ok;                                 // this main workgroup's partner has reported in (read from its mailbox once per CTU)
    i32 seq8, stale8;                            // 8x8 requests posted so far / sequence number of one this workgroup stopped waiting for (nothing is posted until that answer has landed)
    i32 remote8;                                 // the 8x8 CU being walked has its 2Nx2N sets out: the PU chain shares nothing with a four-TU wave here
    i32 ans_seq;                                 // sequence number of the answer wave 0 has staged in LDS (Ans8), or -1: it stopped waiting
    // partner workgroup
    i32 lv_seq, lv_taken;                        // four-TU set: TUs whose levels wave 1 has published (PuX::lev) / the token wavefront has taken, counted over the workgroup's requests
    i32 solo2n;                                  // this workgroup evaluates 2Nx2N sets of 8x8 CUs with no NxN chain beside them: the four-TU wave makes TU 0 itself, the one-TU set has two lenders
};
// A PU step of an 8x8 CU in a wide workgroup (hevc_frame.h pu_step_wide): the PU wave predicts, transforms and quantises the 35 candidates,
// leaves levels and prediction here, makes the FIRST part of every candidate's tokens (cbf, last position, significance / greater-1 /
// greater-2 flags, full sign chunks) in its lane rows and runs the range half of the pricing over them straight from LDS; one partner
// (pu_part_b) makes the REST (remaining levels: rows `brow` — bypass chunks only, which the byte half of the pricing takes from the rows itself);
// another (pu_recon_k, pu_price) makes the reconstructions and SSE and runs the byte half of the pricing.  Tokens reach global memory only
// where someone needs them there.
#define NXN_KEEP_STRIDE 160                      // tokens reserved per kept PU winner (a PU candidate's stream is at most 1 + 34 + 72)
#define BROW_CAP 72                              // tokens of a partner row: 16 levels x 32 bins at most + 7 pending sign bins = 519 bins <= 65 chunks
#define BROW_STRIDE (BROW_CAP + 10)              // u16 per lane (41 dwords: odd): room for the 8 idle tokens that pad the last token block
struct alignas(16) PuX {
    u32 lev[NMODE][13];                          // per candidate: 16 levels (i16, raster order, zero when the weak-group test cleared the block), then the prediction (16 u8)
    alignas(4) u16 brow[NMODE][BROW_STRIDE];
    i32 bcnt[NMODE];                             // tokens in the partner's row
    i32 na[NMODE];                               // tokens of the first part behind cbf_luma (lane row slots 8 ..)
    u32 pu_seq, b_seq, r_seq;                    // sequence number of the PU whose levels are published / whose rows are complete / whose reconstructions and SSE are in place
    alignas(16) u16 kept[4 * NXN_KEEP_STRIDE];   // the four PU winners' tokens for the pipe wave (round 6: LDS instead of the candidate slot in memory — no store drain on the PU chain, no load latency on its tail): PUs 0 .. 2 back to back, PU 3 from 3 x NXN_KEEP_STRIDE, each run padded to a token block with idle tokens
};
// Third stage of a split trial coder (round 6; the pipe wave's streams of a main workgroup whose 2Nx2N sets are with its partner workgroup — four of its wavefronts are idle):
// the CONTEXT side of the range half on a wavefront of its own.  The context recurrence (state, bin -> next state, :913-920) involves neither the range nor low, so a
// wavefront can run ahead with it and leave, per token, the four LPS ranges of the state the bin met and the token itself with an LPS / MPS bit (two dwords); the range
// wavefront then does the range arithmetic alone (stream_seg_Rq) and the byte wavefront what it always did.  Rings of QDEPTH token blocks per lane, like SplitQ.
#define CQSTRIDE (QDEPTH * 16 + 4)               // dwords per lane row
struct alignas(16) CtxQ {
    u32 rec[NMODE][CQSTRIDE];
    i32 prod[NMODE], cons[NMODE];
    i32 go, done;                                // generation started by the range wavefront / finished by the context wavefront
};
#define WIDE_LDS_BYTES (PIPE_LDS_BYTES + XWAVES * sizeof(PartnerMem) + sizeof(WideCtl) + NLEND * sizeof(WaveMem) + sizeof(PuX) + sizeof(CtxQ))
#define CTXQ (*(CtxQ *)(DYN_LDS + PIPE_LDS_BYTES + XWAVES * sizeof(PartnerMem) + sizeof(WideCtl) + NLEND * sizeof(WaveMem) + sizeof(PuX)))
#define PUX (*(PuX *)(DYN_LDS + PIPE_LDS_BYTES + XWAVES * sizeof(PartnerMem) + sizeof(WideCtl) + NLEND * sizeof(WaveMem)))
#ifdef IMCVT_HOSTEMU
#define DYN_LDS g_pipe_host
#else
#define DYN_LDS g_pipe_lds
#endif
#define XM(i) (*(PartnerMem *)(DYN_LDS + PIPE_LDS_BYTES + (i) * sizeof(PartnerMem)))
#define WCTL (*(WideCtl *)(DYN_LDS + PIPE_LDS_BYTES + XWAVES * sizeof(PartnerMem)))
// Who is whose partner follows from where wavefronts run: wavefronts w and w + 4 of a workgroup share a SIMD (tools/simd_map_probe.hip,
// profiles/r05_simd_map.log), and a SIMD serves ONE wavefront's stream of shifts / bit-field / select instructions at full speed, not two
// (tools/valu_rate_probe.hip) — so the two halves of a chain sit on different SIMDs, and the PU chain's SIMD-mates are the wavefronts with the least to do:
//     SIMD a: wave 0 (one-TU set)   + wave 4 (coders of the four-TU set)            SIMD c: wave 2 (PU chain)  + wave 6 (byte half of the pipe wave: idle until PU 2 is decided)
//     SIMD b: wave 1 (four-TU set)  + wave 5 (a pass + byte half of the one-TU set) SIMD d: wave 3 (pipe wave; before PU 2 is decided: reconstructions + byte half of the PU pricing) + wave 7 (remaining-level tokens of the PU chain)
// (The names are ROLES.  Round 6: in the 8x8 CUs roles 5 and 6 run on each other's wavefront — hevc_frame.h ROLE8_PERM — so that the PU chain's SIMD-mate is the one-TU set's
// light partner and the four-TU passes share theirs with the role that idles until PU 2 is decided: 64 frames 2.30 -> 2.27 s, profiles/r06i_role_perm_ab.log.)
#define WAVE_B_CODER (PIPE_WAVE + 1)
#define WAVE_A_PARTNER (PIPE_WAVE + 2)
#define WAVE_PIPE_PARTNER (PIPE_WAVE + 3)
#define WAVE_PU_PARTNER (PIPE_WAVE + 4)
HD u8 *wave_mem_ptr(int w) {                     // a wavefront's WaveMem: waves 0..2 in the static image, the lenders' (waves 5, 6, 7) in the dynamic part (wide workgroups only)
    return w < NWAVES ? SM.wraw + w * sizeof(WaveMem) : (u8 *)DYN_LDS + PIPE_LDS_BYTES + XWAVES * sizeof(PartnerMem) + sizeof(WideCtl) + (w - WAVE_A_PARTNER) * sizeof(WaveMem);
}
HD int cg_pos(int st, int s, int g) { return st == 0 ? SM.T.cgpos_d[s][g] : (s == 0 ? 0 : SM.T.cgpos_hv[st - 1][g]); }
HD int cg_rank(int st, int s, int bit) { return st == 0 ? SM.T.cgrank_d[s][bit] : (s == 0 ? 0 : SM.T.cgrank_hv[st - 1][bit]); }

// Optional cycle accounting per wave (build with -DIMCVT_PROF): category -> accumulated shader clocks.
// -DIMCVT_PROF -DIMCVT_PROF_TL instead records a TIMELINE of the 8x8 CUs of a wide workgroup: tl_mark(ev) adds the time since the CU was entered
// (decide_cu) to slot ev of the same array (slot 0 counts the CUs), from whichever wavefront reaches the event; tools/prof_timeline.py names the slots.
#if defined(IMCVT_PROF) && defined(IMCVT_PROF_TL) && !defined(IMCVT_HOSTEMU)
HD long long prof_now() { return 0; }
HD int threadIdx_wave() { return (int)(threadIdx.x >> 6); }
HD void prof_add(int, long long) {}
HD void prof_add_row(int, int, long long) {}
HD void prof_cnt(int, int) {}
HD void tl_start() { if (threadIdx.x == 0) { SM.tl_t0 = (unsigned long long)clock64(); ((unsigned long long *)SM.prof)[0] += 1; } }
HD void tl_mark(int ev) { if ((threadIdx.x & 63u) == 0) ((unsigned long long *)SM.prof)[ev] += (unsigned long long)clock64() - SM.tl_t0; }
HD void tl_count(int ev) { if ((threadIdx.x & 63u) == 0) ((unsigned long long *)SM.prof)[ev] += 1; }
#elif defined(IMCVT_PROF) && !defined(IMCVT_HOSTEMU)
HD long long prof_now() { return clock64(); }
HD int threadIdx_wave() { return (int)(threadIdx.x >> 6); }
HD void prof_add(int cat, long long t0) { if ((threadIdx.x & 63u) == 0 && threadIdx.x < WG_THREADS) SM.prof[threadIdx.x >> 6][cat] += (unsigned long long)(clock64() - t0); }
HD void prof_add_row(int row, int cat, long long t0) { if ((threadIdx.x & 63u) == 0) SM.prof[row][cat] += (unsigned long long)(clock64() - t0); }      // a partner wavefront's time, booked in a column its owner's row does not use
HD void prof_cnt(int cat, int n) { if ((threadIdx.x & 63u) == 0 && threadIdx.x < WG_THREADS) SM.prof[threadIdx.x >> 6][cat] += (unsigned long long)n; }
HD void tl_start() {}
HD void tl_mark(int) {}
HD void tl_count(int) {}
#else
HD long long prof_now() { return 0; }
HD int threadIdx_wave() { return 0; }
HD void prof_add(int, long long) {}
HD void prof_add_row(int, int, long long) {}
HD void prof_cnt(int, int) {}
HD void tl_start() {}
HD void tl_mark(int) {}
HD void tl_count(int) {}
#endif
// Wave collectives for the decisions: the reference's "last minimum wins" scan (`best >= cost` accepts, :1439, :1475, :1520)
// over a wave's lanes = the minimum over the valid lanes, then the highest valid lane that holds it.
#ifdef IMCVT_HOSTEMU
HD int wave_min_i32(int v, int l) {
    for (int d = 32; d >= 1; d >>= 1) v = imin(v, wave_shfl(v, l ^ d));
    return v;
}
#else
// minimum over the wavefront by data-parallel-primitive moves (no LDS round trips: six of them were ~400 cycles on the PU chain's pick):
// within quads, half rows and rows of 16, then row 15 -> rows 1 / 3 and lane 31 -> the upper half; lane 63 holds the minimum
HD int wave_min_i32(int v, int) {
    v = imin(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false));       // quad_perm [1,0,3,2]
    v = imin(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false));       // quad_perm [2,3,0,1]
    v = imin(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xF, 0xF, false));      // row_half_mirror
    v = imin(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xF, 0xF, false));      // row_mirror
    v = imin(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xA, 0xF, false));      // row_bcast:15 into rows 1 and 3
    v = imin(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xC, 0xF, false));      // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}
#endif
HD int hibit64(u64 m) { return (m >> 32) ? 32 + hibit((u32)(m >> 32)) : hibit((u32)m); }
// returns the winning lane (valid lanes only; at least one lane must be valid); *mn receives the minimum
HD int wave_last_min(int cost, int valid, int l, int *mn) {
    const int m = wave_min_i32(valid ? cost : I32MAX, l);
    *mn = m;
    return hibit64(wave_ballot(valid && cost == m));
}

// workgroup barrier whose wait time is booked under PF_SYNC
HD void wg_sync_p() { const long long t = prof_now(); wg_sync(); prof_add(PF_SYNC, t); }

// Byte sinks.  sink[a.cnt] is where the next byte goes.
//   Sink     — straight to global memory (the live coder: split flags, terminate bins, finish; the safe trial path).
//   RingSink — the trial coders: bytes collect in a 64-byte LDS ring per lane and leave as aligned 16-byte global stores
//              between token blocks, so the hot inner loop issues no VMEM instruction at all (a token load then only
//              has to be waited for where it is used, one block later).  A burst the ring cannot take (a run of >= 30
//              pending 0xFF bytes resolving at once) raises `ovf`; the caller then repeats the trial on the safe path.
struct Sink { u8 *base; u32 off; };          // byte i of the lane's run lives at base[off + i]; base is wave-uniform
HD void sink_put(Sink &s, int i, int v) { g_st8(s.base + (u32)(s.off + (u32)i), v); }
struct CountSinkT { int dummy; };             // bytes are counted (a.cnt), not kept
HD void sink_put(CountSinkT &, int, int) {}
// LeadSink — the trial coders.  A trial's BYTES are wanted only if it wins; its cost needs their NUMBER.  So a trial coder does not run the
//   byte-level logic (:863-878, :820-831: carry into the buffered byte, runs of 0xFF, emulation prevention) at all: it leaves the 9-bit lead
//   (carry + byte) of every byte that leaves `low` (:858-862) in a 16-entry LDS ring per lane, flushed as aligned 16-byte global stores
//   between token blocks (the hot loop issues no VMEM instruction).  Every lead becomes exactly one byte — buffered, part of a run of
//   0xFF, or emitted — so bytes emitted + buffered grow by the number of leads, plus one per emulation-prevention byte inserted; and such a
//   byte takes two emitted zero bytes in a row and then a byte of 3 or less, i.e. (a byte is its lead's low byte plus a carry) two leads
//   with low byte 0x00 / 0xFF and then one with 0xFF / 0x00..0x03.  The flush looks for that pattern (`hit`; the bytes still buffered on
//   entry count as leads: lsink_begin); a lane that shows it gets its byte-level state by the real logic over its list (leads_exact) —
//   a few lanes per frame.  The winner's list is turned into bytes once, by a whole wavefront (resolve_leads).
#ifdef IMCVT_HOSTEMU
HD u32 brev32(u32 x) { u32 r = 0; for (int i = 0; i < 32; i++) r |= ((x >> i) & 1u) << (31 - i); return r; }
HD u32 lperm(u32 hi, u32 lo, u32 sel) { const u64 v = (u64)hi << 32 | lo; u32 r = 0; for (int k = 0; k < 4; k++) r |= (u32)((v >> (8 * ((sel >> (8 * k)) & 7))) & 255) << (8 * k); return r; }
HD u32 udot4(u32 a, u32 b, u32 c) { for (int k = 0; k < 4; k++) c += ((a >> (8 * k)) & 255u) * ((b >> (8 * k)) & 255u); return c; }
#else
HD u32 brev32(u32 x) { return __builtin_bitreverse32(x); }
HD u32 lperm(u32 hi, u32 lo, u32 sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
HD u32 udot4(u32 a, u32 b, u32 c) { return __builtin_amdgcn_udot4(a, b, c, false); }      // v_dot4_u32_u8: four byte products and an addend
#endif
// What the flush looks at in a lead: its low byte is 0x00 (O) / 0xFF (F) / at most 3 (S), it carries into the lead before it (C); D: an emitted
// zero byte.  With the carry INTO a byte known — from the lead behind it, or through a run of 0xFF behind it: CI — the byte is zero iff
// O & !CI | F & CI (| D), and at most 3 only if S or F & CI.  A flush knows the eight leads it takes and the two before them; the carry into
// its last lead is the only unknown (it comes with the next flush) and is taken as set.  (Tests build with -DEP_GUARD_WIDE: every low byte up
// to 0x1F counts as zero, which puts many lanes on the exact path.)
struct LeadSink { u16 *ring; u8 *gbuf; int fl; u32 hist; int hit; };     // lead i: ring[i % 16] until flushed (fl leads, a multiple of 8), then ((u16 *)gbuf)[i]; hist: O | F << 2 | C << 4 | D << 6 of the last two leads (bit 1 of a field: the last, bit 0: the one before it)
#ifdef EP_GUARD_WIDE
#define LEAD_O_MASK 0xE0E0E0E0u
#else
#define LEAD_O_MASK 0xFFFFFFFFu
#endif
HD u32 bytes_nz(u32 v) { return (((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v) >> 7 & 0x01010101u; }      // byte k: 1 if byte k of v is not zero
HD u32 bytes_mask8(u32 f0, u32 f1) { return udot4(f1, 0x80402010u, udot4(f0, 0x08040201u, 0u)); }  // eight 0 / 1 bytes -> eight bits (byte k of f0: bit k, of f1: bit 4 + k)
#if defined(IMCVT_HOSTEMU) && defined(IMCVT_DBGCNT)
static long g_dbg[8];      // (test builds: leads flushed, flushes with the pattern, seeds with it, sinks opened)
#define DBGCNT(i, n) (g_dbg[i] += (n))
#else
#define DBGCNT(i, n) ((void)0)
#endif
HD void lsink_begin(LeadSink &s, const Arith &a0, u16 *ring, u8 *gbuf) {
    s.ring = ring; s.gbuf = gbuf; s.fl = 0;
    const u32 d1 = a0.zeros >= 1, d2 = a0.zeros >= 2;                              // emitted zero bytes before the buffered ones
    const int r = a0.nbytes - 1;                                                   // 0xFF bytes buffered behind bufbyte
    const u32 vb = (u32)a0.bufbyte & 0xFFu, bo = (vb & (LEAD_O_MASK & 0xFFu)) == 0u, bf = vb == 0xFFu;
    const int hb = (int)d2 & (vb <= 3u || bf);                                     // two zeros emitted and a buffered byte that may come out small
    if (a0.nbytes < 1) { s.hist = (d1 << 1 | d2) << 6; s.hit = 0; }
    else if (r == 0) { s.hist = bo << 1 | bf << 3 | d1 << 6; s.hit = hb; }
    else if (r == 1) { s.hist = bo | (2u | bf) << 2; s.hit = hb | (int)(d1 & (bo | bf)); }      // (... or a zero, a buffered byte that may come out zero, and the 0xFF behind it)
    else { s.hist = 3u << 2; s.hit = 1; }                                          // (a longer run of 0xFF buffered on entry: practically never — the exact path)
#ifdef IMCVT_FORCE_OVF          // (test builds: every lane takes the exact path)
    s.hit = 1;
#endif
    DBGCNT(2, s.hit); DBGCNT(3, 1);
}
HD void lsink_flush8(LeadSink &s, int valid) {           // the ring is only 4-byte aligned (odd dword stride between lanes); `valid` of the 8 leads are real
    const u32a *r = (const u32a *)(s.ring + (s.fl & (LRING - 1)));
    U4 b; b.x = r[0]; b.y = r[1]; b.z = r[2]; b.w = r[3];
    g_st128(s.gbuf + 2 * s.fl, b); s.fl += 8;
    RCNT_ANY(76);                                         // (region-counter builds: wave executions of a flush)
    // the eight low bytes as two dwords, the eight carry bits as two more; then one 8-bit mask per predicate (bit j: lead j)
    const u32 L0 = lperm(b.y, b.x, 0x06040200u), L1 = lperm(b.w, b.z, 0x06040200u);
#ifndef LSINK_NO_QUIET
    // Round 6: the quiet flush.  The pattern below needs a lead that may come out ZERO — low byte 0x00 (as LEAD_O_MASK sees it), or 0xFF with a carry into it —
    // among these eight, the two before them, or a zero byte already emitted (D).  Without any of those Z is empty, nothing hits, and what the next flush
    // inherits is the carry bits of leads 6 and 7 alone.  A lead's low byte is 0x00 or 0xFF with probability 1 / 128: 94 % of the flushes are quiet, and a
    // flush is a wave execution for the one or two lanes whose ring has filled — ~120 vector instructions became ~35 (the guard was a sixth of the
    // kernel's vector instructions at the bench shape, profiles/r06e_valu_dyn_mix.log: p2_ring_sync).
    {
        const u32 nz = bytes_nz(L0 & LEAD_O_MASK) & bytes_nz(L1 & LEAD_O_MASK) & bytes_nz(~L0) & bytes_nz(~L1);
        const int loud = !(valid >= 8 && nz == 0x01010101u && (s.hist & 0xCFu) == 0u);
        if (!WAVE_ANY(loud)) {                            // every lane of this wave execution is quiet (a loud one takes the quiet ones along: same result)
            RCNT_ANY(77);                                 // (region-counter builds: ... of which quiet)
            DBGCNT(0, 8);
            s.hist = ((b.w >> 8 & 1u) | (b.w >> 23 & 2u)) << 4;
            return;
        }
    }
#endif
    const u32 H0 = lperm(b.y, b.x, 0x07050301u), H1 = lperm(b.w, b.z, 0x07050301u);
    const int top = (valid >= 8 ? 8 : valid) + 1;         // bit of the last real lead once the two leads before the eight sit in bits 1, 0
    const u32 keep = (2u << top) - 1u, live8 = keep >> 2;
    const u32 O8 = ~bytes_mask8(bytes_nz(L0 & LEAD_O_MASK), bytes_nz(L1 & LEAD_O_MASK)) & live8, F8 = ~bytes_mask8(bytes_nz(~L0), bytes_nz(~L1)) & live8;
    const u32 S8 = ~bytes_mask8(bytes_nz(L0 & 0xFCFCFCFCu), bytes_nz(L1 & 0xFCFCFCFCu)) & live8, C8 = bytes_mask8(H0 & 0x01010101u, H1 & 0x01010101u) & live8;
    const u32 O = O8 << 2 | (s.hist & 3u), Fm = F8 << 2 | (s.hist >> 2 & 3u), Cm = C8 << 2 | (s.hist >> 4 & 3u), D = s.hist >> 6 & 3u, S3 = S8 << 2;
    // CO_j = C_j | F_j & CO_{j+1} (CO above the last lead: 1), CI_j = CO_{j+1}: a carry look-ahead, by an addition over the bit-reversed masks
    const u32 Cr = brev32(Cm) >> (31 - top), Fr = brev32(Fm) >> (31 - top);
    const u32 A = Cr | Fr, S = A + Cr + 1u;
    const u32 CI = brev32((S ^ A ^ Cr) << (31 - top)) & keep;
    const u32 Z = D | (O & ~CI) | (Fm & CI), T = S3 | (Fm & CI);
    const int hitn = (T & (Z << 1) & (Z << 2) & keep & ~3u) != 0u;      // a lead of this flush that may come out at most 3 behind two bytes that come out zero
    DBGCNT(0, valid >= 8 ? 8 : valid); DBGCNT(1, hitn);
    s.hit |= hitn;
    s.hist = (O >> (top - 1) & 3u) | (Fm >> (top - 1) & 3u) << 2 | (Cm >> (top - 1) & 3u) << 4 | (D >> (top - 1) & 3u) << 6;
}
HD void lsink_sync(LeadSink &s, int qn) { NOUNROLL while (qn - s.fl >= 8) lsink_flush8(s, 8); }             // between token blocks: < 8 leads stay pending
HD void lsink_finish(LeadSink &s, int qn) { NOUNROLL while (qn > s.fl) lsink_flush8(s, qn - s.fl); }       // tail: the leads beyond qn are never read
template <class S>
HD void emit_byte(Arith &a, S &sink, int v) {                                                     // :820-831
    v &= 0xFF;
    if (a.zeros >= 2 && v <= 3) { sink_put(sink, a.cnt, 3); a.cnt++; a.zeros = 0; }
    sink_put(sink, a.cnt, v);
    a.cnt++;
    a.zeros = v ? 0 : a.zeros + 1;
}
// the rare branches of :858-878: a run of 0xFF bytes grows or resolves, emulation prevention, the very first byte
template <class S>
HD void carry_rare(Arith &a, S &sink, int lead) {
    if (lead == 0xFF) a.nbytes++;
    else if (a.nbytes > 0) {
        int carry = lead >> 8, v = a.bufbyte + carry;
        a.bufbyte = lead & 0xFF;
        emit_byte(a, sink, v);
        v = (0xFF + carry) & 0xFF;
        NOUNROLL
        for (; a.nbytes > 1; a.nbytes--) emit_byte(a, sink, v);
    } else { a.nbytes = 1; a.bufbyte = lead; }
}
template <class S>
HD void carry_out(Arith &a, S &sink) {                                                            // :858-878
    if (a.nbits < 12) {
        const int lead = (int)((u32)a.low >> (24 - a.nbits));
        a.nbits += 8;
        a.low &= (i32)(0xFFFFFFFFu >> a.nbits);
        const int v1 = (a.bufbyte + (lead >> 8)) & 0xFF;
        if (a.nbytes == 1 && lead != 0xFF && !(a.zeros >= 2 && v1 <= 3)) {
            // common case: exactly one byte is buffered, no run of 0xFF, no emulation prevention
            sink_put(sink, a.cnt++, v1);
            a.zeros = v1 ? 0 : a.zeros + 1;
            a.bufbyte = lead & 0xFF;
        } else carry_rare(a, sink, lead);
    }
}
HD void code_bin(Arith &a, u8 *cx, Sink &sink, int ci, int bin) {                     // :913-932
    const int p = cx[ci];
    const uint2 e = SM.T.pst[p];
    const int lps = (int)((e.x >> (((a.range >> 6) & 3) * 8)) & 0xFF);
    const int rm = a.range - lps;
    const int is_lps = (bin ^ p) & 1;
    const int sh = is_lps ? imin(6, clz32((u32)lps) - 23) : (rm < 256);     // renorm table :714 == 8 - floor(log2 lps), capped at 6
    cx[ci] = (u8)(is_lps ? e.y : e.y >> 8);
    a.low = (a.low + (is_lps ? rm : 0)) << sh;
    a.range = (is_lps ? lps : rm) << sh;
    a.nbits -= sh;
    carry_out(a, sink);
}
HD void code_terminate(Arith &a, Sink &sink, int bin) {                                      // :881-895
    a.range -= 2;
    if (bin) { a.low = (a.low + a.range) << 7; a.range = 256; a.nbits -= 7; }
    else if (a.range < 256) { a.low <<= 1; a.range <<= 1; a.nbits--; }
    carry_out(a, sink);
}
HD void arith_finish(Arith &a, Sink &sink) {                                                 // :839-855
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
// Prediction (:262-381), evaluated per pixel
// ---------------------------------------------------------------------------------------------------
HD int uses_filtered(int N, int mode) {            // :274-280 as a distance-to-H/V threshold
    if (N == 4 || mode == 1) return 0;
    if (mode == 0) return 1;
    int d = imin(iabs(mode - 10), iabs(mode - 26));
    return d > (N == 8 ? 7 : N == 16 ? 1 : 0);
}

struct BorderRef { const u8 *ul, *ua, *fl, *fa; int uc, fc, dc; };
HD int pred_px(const Tables &T, const BorderRef &b, int N, int lg, int mode, int y, int x) {
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

// Prediction of one 4x4 block at (y0,x0) of an N x N predictor (same arithmetic as pred_px, :262-381), with the per-row
// angle terms hoisted and neighbouring reference samples shared between pixels.
HD int ref_line(const u8 *M, const u8 *Sd, int corner, int iang, int t) {     // t==0 corner, t>0 main[t-1], t<0 projected side (:353-364)
    return t == 0 ? corner : t > 0 ? M[t - 1] : Sd[((128 - mul24(iang, t)) >> 8) - 1];      // iang <= 4096, |t| <= 65
}
HD void pred_block4(const Tables &T, const BorderRef &b, int N, int lg, int mode, int y0, int x0, int out[4][4]) {
    const int f = uses_filtered(N, mode);
    const u8 *L = f ? b.fl : b.ul, *A = f ? b.fa : b.ua;
    const int corner = f ? b.fc : b.uc;
    if (mode == 0) {
        const int an = A[N], ln = L[N];
        int lv[4], av[4];
        for (int i = 0; i < 4; i++) { lv[i] = L[y0 + i]; av[i] = A[x0 + i]; }
        for (int yi = 0; yi < 4; yi++) for (int xi = 0; xi < 4; xi++) {
            const int y = y0 + yi, x = x0 + xi;
            out[yi][xi] = (mul24(N - 1 - x, lv[yi]) + mul24(x + 1, an) + mul24(N - 1 - y, av[xi]) + mul24(y + 1, ln) + N) >> (lg + 1);
        }
    } else if (mode == 1) {
        const int dc = b.dc;
        for (int yi = 0; yi < 4; yi++) for (int xi = 0; xi < 4; xi++) out[yi][xi] = dc;
        if (N <= 16) {
            if (y0 == 0) for (int xi = 0; xi < 4; xi++) out[0][xi] = (2 + 3 * dc + A[x0 + xi]) >> 2;
            if (x0 == 0) for (int yi = 0; yi < 4; yi++) out[yi][0] = (2 + 3 * dc + L[y0 + yi]) >> 2;
            if (y0 == 0 && x0 == 0) out[0][0] = (2 + 2 * dc + L[0] + A[0]) >> 2;
        }
    } else if (mode == 10) {
        for (int yi = 0; yi < 4; yi++) { const int v = L[y0 + yi]; for (int xi = 0; xi < 4; xi++) out[yi][xi] = v; }
        if (N <= 16 && y0 == 0) for (int xi = 0; xi < 4; xi++) out[0][xi] = clip3(((A[x0 + xi] - corner) >> 1) + L[0], 0, 255);
    } else if (mode == 26) {
        for (int xi = 0; xi < 4; xi++) { const int v = A[x0 + xi]; for (int yi = 0; yi < 4; yi++) out[yi][xi] = v; }
        if (N <= 16 && x0 == 0) for (int yi = 0; yi < 4; yi++) out[yi][0] = clip3(((L[y0 + yi] - corner) >> 1) + A[0], 0, 255);
    } else {
        const int horiz = mode < 18;
        const int ang = (int)T.ang[mode] - 32, iang = T.iang[mode];
        const u8 *M = horiz ? L : A, *Sd = horiz ? A : L;
        const int i0 = horiz ? x0 : y0, j0 = horiz ? y0 : x0;           // i runs along the prediction direction
        for (int ii = 0; ii < 4; ii++) {
            const int off = mul24(ang, i0 + ii + 1), oi = off >> 5, of = off & 31, t0 = oi + j0 + 1;
            int p[5];
            for (int k = 0; k < 5; k++) p[k] = ref_line(M, Sd, corner, iang, t0 + k);      // M[2N] is read only with of==0
            for (int jj = 0; jj < 4; jj++) {
                const int v = (mul24(32 - of, p[jj]) + mul24(of, p[jj + 1]) + 16) >> 5;
                if (horiz) out[jj][ii] = v; else out[ii][jj] = v;
            }
        }
    }
}

// Prediction of the vertical strip (y0 .. y0+3, x) of an N x N predictor, packed one sample per byte (row y0 + t in byte t):
// the layout of the matrix-core passes (a lane owns a column and four-row strips of it).  `mode` is wave-uniform there.
HD u32 pred_strip4(const Tables &T, const BorderRef &b, int N, int lg, int mode, int y0, int x) {
    const int f = uses_filtered(N, mode);
    const u8 *L = f ? b.fl : b.ul, *A = f ? b.fa : b.ua;
    const int corner = f ? b.fc : b.uc;
    int v[4];
    if (mode == 0) {
        const int ax = A[x], ln = L[N], base = mul24(x + 1, A[N]) + N;
        for (int t = 0; t < 4; t++) { const int y = y0 + t; v[t] = (mul24(N - 1 - x, L[y]) + base + mul24(N - 1 - y, ax) + mul24(y + 1, ln)) >> (lg + 1); }
    } else if (mode == 1) {
        const int dc = b.dc;
        for (int t = 0; t < 4; t++) v[t] = dc;
        if (N <= 16) {
            if (x == 0) for (int t = 0; t < 4; t++) v[t] = (2 + 3 * dc + L[y0 + t]) >> 2;
            if (y0 == 0) v[0] = (x == 0) ? (2 + 2 * dc + L[0] + A[0]) >> 2 : (2 + 3 * dc + A[x]) >> 2;
        }
    } else if (mode == 10) {
        for (int t = 0; t < 4; t++) v[t] = L[y0 + t];
        if (N <= 16 && y0 == 0) v[0] = clip3(((A[x] - corner) >> 1) + L[0], 0, 255);
    } else if (mode == 26) {
        const int ax = A[x];
        for (int t = 0; t < 4; t++) v[t] = ax;
        if (N <= 16 && x == 0) for (int t = 0; t < 4; t++) v[t] = clip3(((L[y0 + t] - corner) >> 1) + A[0], 0, 255);
    } else {
        const int ang = (int)T.ang[mode] - 32, iang = T.iang[mode];
        if (mode < 18) {                                          // horizontal family: the angle term belongs to the column
            const int off = mul24(ang, x + 1), oi = off >> 5, of = off & 31, t0 = oi + y0 + 1;
            int p[5];
            for (int k = 0; k < 5; k++) p[k] = ref_line(L, A, corner, iang, t0 + k);
            for (int t = 0; t < 4; t++) v[t] = (mul24(32 - of, p[t]) + mul24(of, p[t + 1]) + 16) >> 5;
        } else {                                                  // vertical family: one angle term per row
            for (int t = 0; t < 4; t++) {
                const int off = mul24(ang, y0 + t + 1), oi = off >> 5, of = off & 31, t1 = oi + x + 1;
                v[t] = (mul24(32 - of, ref_line(A, L, corner, iang, t1)) + mul24(of, ref_line(A, L, corner, iang, t1 + 1)) + 16) >> 5;
            }
        }
    }
    return (u32)v[0] | (u32)v[1] << 8 | (u32)v[2] << 16 | (u32)v[3] << 24;
}

// ---------------------------------------------------------------------------------------------------
// Border assembly
// ---------------------------------------------------------------------------------------------------
// Shared border of the block at (y0,x0) from the reconstruction tile.  Wave-uniform call.
#ifndef HDN_BORDER
#define HDN_BORDER HDN
#endif
#ifndef HDN_BFT
#define HDN_BFT HDN_BORDER
#endif
#ifndef HDN_BTS
#define HDN_BTS HDN_BORDER
#endif
#ifndef HDN_EVAL
#define HDN_EVAL HDN
#endif
HDN_BFT void border_from_tile(int wave_, int N_, int y0_, int x0_, int hl_, int hbl_, int ha_, int har_) {
    const int wave = uni_i(wave_); const int N = uni_i(N_); const int y0 = uni_i(y0_); const int x0 = uni_i(x0_); const int hl = uni_i(hl_); const int hbl = uni_i(hbl_); const int ha = uni_i(ha_); const int har = uni_i(har_);
    WaveMem &W = WM(wave);
    Border &b = W.bsh;
    const int n2 = 2 * N;
    LANES(l) {
        const u8 *t = &SM.rec[y0 + 1][x0 + 1];
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
            b.dc = (i16)(dc >> (hibit((u32)N) + 1));                   // dc / 2N (N is a power of two, dc is not negative; a shift, not a division sequence)
            b.fc = (u8)((2 + b.ul[0] + b.ua[0] + 2 * b.uc) >> 2);
        }
    }
    wave_sync();
}

// The same for a 4x4 block, inline and in one step (the PU chain of an 8x8 CU in a wide workgroup walks it four times per CU): 4x4 blocks are
// predicted from the unfiltered samples only (:274-280), so the [1 2 1] variants are not made, and the DC value is summed from the tile directly.
HD void border4_from_tile(WaveMem &W, int y0, int x0, int hl, int hbl, int ha, int har) {
    Border &b = W.bsh;
    LANES(l) {
        const u8 *t = &SM.rec[y0 + 1][x0 + 1];
        const int uc = (hl && ha) ? t[-RS - 1] : hl ? t[-1] : ha ? t[-RS] : 128;
        if (l < 8) {
            int lv_, av_;
            if (l < 4) { lv_ = hl ? t[l * RS - 1] : uc; av_ = ha ? t[-RS + l] : uc; }
            else {
                lv_ = hbl ? t[l * RS - 1] : (hl ? t[3 * RS - 1] : uc);
                av_ = har ? t[-RS + l] : (ha ? t[-RS + 3] : uc);
            }
            b.ul[l] = (u8)lv_; b.ua[l] = (u8)av_;
        }
        if (l == 8) {
            int dc = 4;
            for (int i = 0; i < 4; i++) dc += (hl ? t[i * RS - 1] : uc) + (ha ? t[-RS + i] : uc);
            b.dc = (i16)(dc >> 3);
            b.uc = (u8)uc; b.ul[8] = 0; b.ua[8] = 0;
        }
    }
    wave_sync_lds();
}

// Per-mode borders of TU k (1..3) of the four-TU shape of the CU at (y0,x0,N): samples inside the CU come
// from that mode's own reconstruction of TUs < k (:1459,1466), the rest from the tile.  Only the variant the mode
// predicts from is stored (mode c of size h uses either the unfiltered or the [1 2 1]-filtered border, :274-289).
// Written without lane-divergent branches: which source a sample comes from is an index / pointer select, the [1 2 1]
// filter is always evaluated and selected by the mode's flag (availability flags and K are wave-uniform).
struct TuSrc { const u8 *col0, *col2, *row0, *row1, *ab, *lf; int h, hl, hbl, ha, har, uc; };
template <int K>
HD int tu_left(const TuSrc &s, int i) {          // unfiltered left / below-left sample i (0..2h-1)
    const int h = s.h;
    if (K == 1) return s.col0[imin(i, h - 1)];
    if (K == 3) return s.col2[imin(i, h - 1)];
    const int inside = i < h;
    const int idx = (inside | s.hbl) ? i : h - 1, use = inside ? s.hl : (s.hbl | s.hl);      // rows beyond the tile are never addressed: idx stays below h without below-left
    const int v = s.lf[idx * RS];
    return use ? v : s.uc;
}
template <int K>
HD int tu_above(const TuSrc &s, int i) {         // unfiltered above / above-right sample i
    const int h = s.h;
    if (K == 2) { const u8 *q = (i < h) ? s.row0 + i : s.row1 + (i - h); return *q; }
    if (K == 3) return s.row1[imin(i, h - 1)];
    const int inside = i < h;
    const int idx = (inside | s.har) ? i : h - 1, use = inside ? s.ha : (s.har | s.ha);
    const int v = s.ab[idx];
    return use ? v : s.uc;
}
template <int K>
HD void border_tu_split_k(int N, int y0, int x0, int hl, int hbl, int ha, int har, int c_lo, int c_hi) {
    const int h = N / 2, n2 = N, lg2 = hibit((u32)N);   // 2*h entries per side
    LANES(l) {
        NOUNROLL
        for (int e0 = c_lo * n2; e0 < c_hi * n2; e0 += 64) {
            const int e = e0 + l;
            if (e < c_hi * n2) {
                const int c = e >> lg2, i = e & (n2 - 1);              // (n2 = 8 or 16)
                BorderS &b = SM.X.bc[c];
                TuSrc s;
                s.h = h; s.hl = hl; s.hbl = hbl; s.ha = ha; s.har = har;
                s.col0 = SM.X.t3col[c][0]; s.col2 = SM.X.t3col[c][2]; s.row0 = SM.X.t3row[c][0]; s.row1 = SM.X.t3row[c][1];
                s.ab = &SM.rec[y0][x0 + h + 1];                       // row y0-1, starting at column x0+h
                s.lf = &SM.rec[y0 + h + 1][x0];                       // column x0-1, starting at row y0+h
                s.uc = (K == 1) ? (ha ? s.ab[-1] : s.col0[0]) : (K == 2) ? (hl ? s.lf[-RS] : s.row0[0]) : s.row0[h - 1];
                const int filt = uses_filtered(h, c);
                const int lc = tu_left<K>(s, i), ac = tu_above<K>(s, i);
                const int lp = (i == 0) ? s.uc : tu_left<K>(s, imax(i - 1, 0)), ap = (i == 0) ? s.uc : tu_above<K>(s, imax(i - 1, 0));
                const int ln = tu_left<K>(s, imin(i + 1, n2 - 1)), an = tu_above<K>(s, imin(i + 1, n2 - 1));
                const int fsel = filt & (i != n2 - 1);                 // [1 2 1]/4, ends unfiltered (:246-256)
                b.l[i] = (u8)(fsel ? (2 + 2 * lc + lp + ln) >> 2 : lc);
                b.a[i] = (u8)(fsel ? (2 + 2 * ac + ap + an) >> 2 : ac);
                if (i == 0) {
                    b.c = (u8)(filt ? (2 + lc + ac + 2 * s.uc) >> 2 : s.uc);
                    b.l[n2] = 0; b.a[n2] = 0;
                    if (c == 1) {                                       // only the DC mode reads dc (unfiltered border)
                        int dc = h;
                        for (int j = 0; j < h; j++) dc += tu_left<K>(s, j) + tu_above<K>(s, j);
                        b.dc = (i16)(dc >> lg2);                         // dc / 2h
                    }
                }
            }
        }
    }
    wave_sync();
}
HDN_BTS void border_tu_split(int N_, int y0_, int x0_, int k_, int hl_, int hbl_, int ha_, int har_, int c_lo_, int c_hi_) {
    const int N = uni_i(N_); const int y0 = uni_i(y0_); const int x0 = uni_i(x0_); const int k = uni_i(k_); const int hl = uni_i(hl_); const int hbl = uni_i(hbl_); const int ha = uni_i(ha_); const int har = uni_i(har_); const int c_lo = uni_i(c_lo_); const int c_hi = uni_i(c_hi_);      // modes c_lo .. c_hi-1
    if (k == 1) border_tu_split_k<1>(N, y0, x0, hl, hbl, ha, har, c_lo, c_hi);
    else if (k == 2) border_tu_split_k<2>(N, y0, x0, hl, hbl, ha, har, c_lo, c_hi);
    else border_tu_split_k<3>(N, y0, x0, hl, hbl, ha, har, c_lo, c_hi);
}

// ---------------------------------------------------------------------------------------------------
// The candidate pipeline: predict -> residual -> T -> RDOQ -> deQ -> T^-1 -> recon -> SSE  (:1425-1436)
// One lane owns one 4x4 output block (= one coefficient group) of one candidate; a pass covers
// 64/(N/4)^2 candidates.  All intermediates of a pass live in the wave's LDS slice.
// ---------------------------------------------------------------------------------------------------
enum { OUT_NONE = 0, OUT_REC4 = 1, OUT_T3SIDE = 2, OUT_TILE = 3 };

struct P1Args {
    int N, y0, x0;       // block inside the CTU
    int k;               // TU slot (four-TU shape)
    int per_mode_border; // 0: W.bsh, 1: SM.X.bc[c]
    int out_kind;        // what to keep of the reconstruction
    int only_mode;       // -1: all 35 modes, else just this one (winner reconstruction)
    int shape;           // CU shape the TU belongs to (0: one TU, 1: four TUs, 2: NxN, 3: PU pricing) — picks the cbf_luma context
    u16 *tok;            // the OWNER's token streams ([TOK_SLOTS][TOK_CAP]); null: no tokens (winner reconstruction)
    int q;
    int own;             // wave whose candidate set this is (its tokn / tnz / sse arrays and token streams); the executing wave lends lanes and its pass buffer
    int c_lo, c_hi;      // candidates (modes) handled by this call
    int hint;            // 4x4 PU candidates (shape 3): the tokens carry state hints (tokg_a_fast<0, true>) — they are priced by code_token_r
    u8 *rec8;            // partner workgroups (8x8 CUs, hevc_frame.h): every candidate keeps its reconstruction here, 64 bytes per candidate in raster order (LDS), so that the winner's need not be rebuilt; else null
};

// sign-extended byte kk of a packed word / i16 halves of a packed word
HD int sx8(u32 w, int kk) { return (int)(i8)(w >> (8 * kk)); }
HD int lo16(u32 w) { return (int)(i16)(w & 0xFFFFu); }
HD int hi16(u32 w) { return (int)w >> 16; }

// acc[r][c] += sum_kk M[row0+r][k0+kk] * X[k0+kk][col0+c]      (M: i8 matrix rows, X: i16 rows of stride N)
// TR: M is the TRANSPOSE of the stored matrix C (M[i][k] = C[k][i]) — the same 4x4 byte block of C is fetched as four row
// words and indexed the other way round, so no transposed copy of the matrices is kept in LDS.
template <int N, bool TR>
HD void mac_MX(int acc[4][4], const i8 *C, const i16 *X, int row0, int col0) {
    NOUNROLL
    for (int k0 = 0; k0 < N; k0 += 4) {
        u32 mw[4];
        for (int r = 0; r < 4; r++) mw[r] = TR ? *(const u32a *)(C + (k0 + r) * N + row0) : *(const u32a *)(C + (row0 + r) * N + k0);
        for (int kk = 0; kk < 4; kk++) {
            const uint2 xw = *(const uint2 *)(X + (k0 + kk) * N + col0);
            const int x0 = lo16(xw.x), x1 = hi16(xw.x), x2 = lo16(xw.y), x3 = hi16(xw.y);
            for (int r = 0; r < 4; r++) {
                const int m = TR ? sx8(mw[kk], r) : sx8(mw[r], kk);
                acc[r][0] += m * x0; acc[r][1] += m * x1; acc[r][2] += m * x2; acc[r][3] += m * x3;
            }
        }
    }
}
// acc[r][c] += sum_kk Y[row0+r][k0+kk] * M[col0+c][k0+kk]      (Y: i32 rows of stride N)
// (sw: the column swizzle of Y's rows row0..row0+3, see p1_run_t — 0 for unswizzled tiles)
template <int N>
HD void mac_YM32(int acc[4][4], const i32 *Y, const i8 *M, int row0, int col0, int sw) {
    NOUNROLL
    for (int k0 = 0; k0 < N; k0 += 4) {
        u32 mw[4];
        for (int c = 0; c < 4; c++) mw[c] = *(const u32a *)(M + (col0 + c) * N + k0);
        for (int r = 0; r < 4; r++) {
            const int4 y = *(const int4 *)(Y + (row0 + r) * N + (k0 ^ sw));
            for (int c = 0; c < 4; c++)
                acc[r][c] += mul24(y.x, sx8(mw[c], 0)) + mul24(y.y, sx8(mw[c], 1)) + mul24(y.z, sx8(mw[c], 2)) + mul24(y.w, sx8(mw[c], 3));   // |tmp| <= 45900 (17 bits)
        }
    }
}
// acc[r][c] += sum_kk Y[row0+r][k0+kk] * C[k0+kk][col0+c]      (Y: i16 rows; the multiplier is C itself, i.e. the transpose of mac_YM32's)
template <int N>
HD void mac_YM16(int acc[4][4], const i16 *Y, const i8 *C, int row0, int col0, int sw) {
    NOUNROLL
    for (int k0 = 0; k0 < N; k0 += 4) {
        u32 mw[4];
        for (int kk = 0; kk < 4; kk++) mw[kk] = *(const u32a *)(C + (k0 + kk) * N + col0);
        for (int r = 0; r < 4; r++) {
            const uint2 yw = *(const uint2 *)(Y + (row0 + r) * N + (k0 ^ sw));
            const int y0 = lo16(yw.x), y1 = hi16(yw.x), y2 = lo16(yw.y), y3 = hi16(yw.y);
            for (int c = 0; c < 4; c++)
                acc[r][c] += y0 * sx8(mw[0], c) + y1 * sx8(mw[1], c) + y2 * sx8(mw[2], c) + y3 * sx8(mw[3], c);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// The N = 16 / 32 transforms on the matrix cores (:469-516, matrices :391-464).
// i8 x i8 -> i32 matrix instructions are exact, so every stage is an integer product whose wider operand is cut into i8
// limbs (layouts verified on the device by tools/mfma_probe.hip):
//   forward 1   tmp  = (C (O - P) + r) >> a       O - P = (O - 128) + (127 - P) + 1: two products with i8 operands
//                                                 (O ^ 0x80, P ^ 0x7F) and the row sums of C (64 N in row 0, else 0)
//   forward 2   coef = tmp C^T + r                tmp (17 bits + sign) = l0 + 256 l1 + 65536 l2, limbs signed: the bytes of
//                                                 tmp + 0x8080 with the low two XOR 0x80 (the bias rides in stage 1's rounding constant)
//   inverse 1/2 X (i16) = 256 hi + (lo - 128) + 128: hi = the high byte as it is, lo ^ 0x80, and 128 x the column sums of C
// Limbs are chained Horner-wise through the accumulator (acc = (acc << 8) + constant between the products).
// A lane (i, h) of the wave owns column / row i and the reduction slots rows(h): for N = 32 (v_mfma_i32_32x32x32_i8, i = l % 32,
// h = l / 32) the rows 8 j + 4 h + t — register 4 j + t, byte t of operand dword j; for N = 16 (v_mfma_i32_16x16x32_i8 with the
// upper half of the reduction zero, i = l % 16, h = l / 16) the rows 4 h + t of each of the pass's four candidates — register
// 4 slot + t, operand dword `slot`.  The results of one stage are the operand slots of the next, so tmp and the inverse's
// intermediate never leave the registers; prediction, reconstruction and SSE are computed in the same layout (column i, the
// lane's rows).  Only the coefficients travel through LDS (to the lanes that own the 4x4 coefficient groups: RDOQ, tokens)
// and the dequantised levels back (as byte limbs, column-major).
// ---------------------------------------------------------------------------------------------------
#ifndef P1_MFMA
#define P1_MFMA 1            // 0: the N = 16 / 32 transforms as vector code like N = 8 (A/B measurements, tests)
#endif
#ifdef IMCVT_HOSTEMU
static void emu_mfma32(const u32 *a, const u32 *b, int *acc);
static void emu_mfma16(const u32 *a, const u32 *b, int *acc);
HD void mfma32(int acc[16], const u32 a[4], const u32 b[4]) { emu_mfma32(a, b, acc); }
HD void mfma16(int acc[4], u32 a, u32 b) { const u32 aa[2] = { a, 0 }, bb[2] = { b, 0 }; emu_mfma16(aa, bb, acc); }
HD u32 perm_b32(u32 hi, u32 lo, u32 sel) {          // v_perm_b32 with selectors 0..7
    const u64 v = (u64)hi << 32 | lo; u32 r = 0;
    for (int k = 0; k < 4; k++) r |= (u32)((v >> (8 * ((sel >> (8 * k)) & 7))) & 255) << (8 * k);
    return r;
}
#else
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v16i_t __attribute__((ext_vector_type(16)));
HD void mfma32(int acc[16], const u32 a[4], const u32 b[4]) {
    v4i_t A, B; v16i_t Cc;
    for (int d = 0; d < 4; d++) { A[d] = (int)a[d]; B[d] = (int)b[d]; }
    for (int r = 0; r < 16; r++) Cc[r] = acc[r];
    Cc = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B, Cc, 0, 0, 0);
    for (int r = 0; r < 16; r++) acc[r] = Cc[r];
}
HD void mfma16(int acc[4], u32 a, u32 b) {
    v4i_t Cc;
    for (int r = 0; r < 4; r++) Cc[r] = acc[r];
    Cc = __builtin_amdgcn_mfma_i32_16x16x32_i8((long)(u64)a, (long)(u64)b, Cc, 0, 0, 0);
    for (int r = 0; r < 4; r++) acc[r] = Cc[r];
}
HD u32 perm_b32(u32 hi, u32 lo, u32 sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
#endif
// acc (16 registers) += a x b: one 32x32x32 product, or the 16x16 products of the pass's live candidates (slots 0 .. nlive-1)
template <int LG>
HD void mx_mm(int acc[16], const u32 a[4], const u32 b[4], int nlive) {
    if constexpr (LG == 5) mfma32(acc, a, b);
    else for (int sl = 0; sl < 4; sl++) if (sl < nlive) mfma16(acc + 4 * sl, a[sl], b[sl]);
}
// byte k of the four values of a register group -> one operand dword per limb
#define SEL_PAIR 0x05010400u     // perm(v1, v0): v0.b0 v1.b0 v0.b1 v1.b1
#define SEL_PAIR2 0x05010602u    // perm(v1, v0): v0.b2 v1.b2 (then bytes that are not used)
#define SEL_LO 0x05040100u       // perm(Y, X): X.b0 X.b1 Y.b0 Y.b1
#define SEL_HI 0x07060302u       // perm(Y, X): X.b2 X.b3 Y.b2 Y.b3
// v = x + 0x8080 with x in 17 bits + sign: x = l0 + 256 l1 + 65536 l2
HD void mx_limbs3(const int v[16], u32 l0[4], u32 l1[4], u32 l2[4]) {
    for (int d = 0; d < 4; d++) {
        const u32 X = perm_b32((u32)v[4 * d + 1], (u32)v[4 * d], SEL_PAIR), Y = perm_b32((u32)v[4 * d + 3], (u32)v[4 * d + 2], SEL_PAIR);
        l0[d] = perm_b32(Y, X, SEL_LO) ^ 0x80808080u; l1[d] = perm_b32(Y, X, SEL_HI) ^ 0x80808080u;
        const u32 X2 = perm_b32((u32)v[4 * d + 1], (u32)v[4 * d], SEL_PAIR2), Y2 = perm_b32((u32)v[4 * d + 3], (u32)v[4 * d + 2], SEL_PAIR2);
        l2[d] = perm_b32(Y2, X2, SEL_LO);
    }
}
// four i16 values (in i32 registers): w = 256 hi + (lo - 128) + 128
HD void mx_limbs2(int w0, int w1, int w2, int w3, u32 &lo, u32 &hi) {
    const u32 X = perm_b32((u32)w1, (u32)w0, SEL_PAIR), Y = perm_b32((u32)w3, (u32)w2, SEL_PAIR);
    lo = perm_b32(Y, X, SEL_LO) ^ 0x80808080u; hi = perm_b32(Y, X, SEL_HI);
}
HD int sum_bytes_i8(u32 w) { return sx8(w, 0) + sx8(w, 1) + sx8(w, 2) + sx8(w, 3); }

// quantiser constants of one (size, qpd6) pair (:546-554, :606-608)
struct QConst { int sh, add, dmax, thr, dq, dqs; RdW rw; };
template <int S>
HD QConst qconst(int q) { QConst Q; Q.sh = 19 - S + q; Q.add = 1 << Q.sh >> 1; Q.dmax = I32MAX - Q.add; Q.thr = 9 << Q.sh >> 2; Q.dqs = 5 - S + q; Q.dq = 1 << Q.dqs; Q.rw = rd_weights(q); return Q; }

// Simplified RDOQ (:540-594).  The reference prices the levels l0 = round(|coef| / step), l0 - 1 and l0 - 2 of every coefficient
// and keeps the cheapest (the larger level on ties).  What it decides depends on |coef| alone (for a given qpd6 and TU size), and
// it has this shape — established by exhaustive comparison with the reference's loop over every |coef| the transforms can
// produce, for all 5 x 4 (qpd6, size) pairs, in tests/test_oracle.py::test_rdoq_thresholds_match_reference_loop:
//   * l0 - 2 never wins: its extra distortion (>= (1.5 step)^2 scaled) outweighs any rate it can save;
//   * l0 - 1 wins iff the remainder x = |coef| 2^14 - l0 step (in [-step/2, step/2)) is at or below a threshold that depends only
//     on the rate step rate(l0) - rate(l0 - 1) (:526-535): one value per l0 in 1..7, one for the l0 >= 8 where the exp-Golomb
//     escape grows (l0 - 5 a power of two), and "never" for every other l0 (no rate to save, more distortion).
// The thresholds (in units of 2^14, i.e. of |coef|) are derived on the host from the reference's own cost loop (hevc_tables.h).
// class of a rounded level: 0..7 itself, 8: l0 >= 8 with l0 - 5 a power of two, 9: the other l0 >= 8
HD int rdoq_class(int l0) { return l0 < 8 ? l0 : (((l0 - 5) & (l0 - 6)) == 0 ? 8 : 9); }
// in: acc = forward-transform sums before the final shift; out: acc = signed levels.  Returns non-zero when the group keeps
// any level after the weak-group test (:588-591).
template <int S>
HD int rdoq_group(int acc[4][4], const QConst &Q) {
    constexpr int b1 = S + 8;
    const i32 *th = SM.rthr[S];
    const int xs = Q.sh - 14;
    int sum = 0, any = 0;
    for (int r = 0; r < 4; r++) for (int cc = 0; cc < 4; cc++) {
        const int cf = acc[r][cc] >> b1, av = imin(iabs(cf), 0x20000);
        const u32 dd = (u32)av << 14;                                             // :556-558 (2^31 when |coef| exceeds 17 bits: clamped next)
        const int d = (int)(dd < (u32)Q.dmax ? dd : (u32)Q.dmax);
        const int l0 = imin((int)(((u32)d + (u32)Q.add) >> Q.sh), 32767);     // the 16-bit clip of :560 (the value is not negative)
        const int xa = av - (l0 << xs);                                           // remainder in units of 2^14 (a clamped |coef| lands in class 9)
        const int pick = l0 - (xa <= th[rdoq_class(l0)]);
        acc[r][cc] = (cf < 0) ? -pick : pick;
        any |= pick;
        sum += imin(d, Q.thr);
    }
    return (sum < Q.thr) ? 0 : any;                     // weak group: cleared (:588-591)
}

HD int scan_type_of(int N, int mode) { return (N <= 8) ? ((iabs(mode - 26) <= 4) ? 1 : (iabs(mode - 10) <= 4) ? 2 : 0) : 0; }   // :1133-1141

HD void fill_border_ref(BorderRef &br, const WaveMem &W, int per_mode, int c) {
    if (per_mode) { const BorderS &b = SM.X.bc[c]; br.ul = b.l; br.ua = b.a; br.fl = b.l; br.fa = b.a; br.uc = b.c; br.fc = b.c; br.dc = b.dc; }
    else { const Border &b = W.bsh; br.ul = b.ul; br.ua = b.ua; br.fl = b.fl; br.fa = b.fa; br.uc = b.uc; br.fc = b.fc; br.dc = b.dc; }
}

// ---------------------------------------------------------------------------------------------------
// Bin tokens.  The syntax of a candidate (:1172-1339) is flattened, by the lanes that own its coefficient
// groups, into a stream of 16-bit tokens in coding order:
//     context-coded bin :  context index << 8 | hint << 1 | bin             (< 0x5B00)
//     bypass chunk      :  0x8000 | nbins << 8 | value   (1..8 bins, the reference's own chunking :898-910)
// `hint` (7 bits) is the packed state the context is in when the bin is coded — filled in only where the generator can know it
// (the PU candidates of an 8x8 CU, priced from fresh contexts: tokg_a_fast<0, true>; the NxN trial's pre-resolved segments), zero
// elsewhere.  A coder that keeps context copies (code_token_q) ignores it; one that runs on resolved tokens (code_token_r) needs
// nothing else.
// Tokens do not depend on the coder or context STATE, so they are produced in parallel (one lane per 4x4
// group) while the arithmetic coding itself — the only truly serial part — becomes a tight loop over a
// linear stream (stream_run below), one lane per candidate.
// ---------------------------------------------------------------------------------------------------
// Where a writer's tokens go: token k lands in tb[pos + k] (LDS) when 0 <= pos + k < cap, else in the dump slot tb[cap].
// Tokens are staged in LDS and leave for global memory as whole 16-byte blocks (8 tokens), never as scattered 2-byte stores.
#define TOK_IDLE 0x8000u
struct TokOut { u16 *tb; int pos; int cap; int glob; };   // glob: tb is the stream in global memory, written in place (no window, no dump slot)
HD void to_put(const TokOut &o, int k, int tok) {
    if (o.glob) { g_st16((i16 *)(o.tb + (o.pos + k)), tok); return; }
    const u32 i = (u32)(o.pos + k); o.tb[i < (u32)o.cap ? i : (u32)o.cap] = (u16)tok;
}
HD void to_put_if(const TokOut &o, int k, int tok, int pred) {
    if (o.glob) { if (pred) g_st16((i16 *)(o.tb + (o.pos + k)), tok); return; }
    const u32 i = pred ? (u32)(o.pos + k) : 0xFFFFFFFFu; o.tb[i < (u32)o.cap ? i : (u32)o.cap] = (u16)tok;
}
struct TokW { TokOut o; int n; int wr; };       // wr == 0: count only
HD void tk_put(TokW &w, int t) { if (w.wr) to_put(w.o, w.n, t); w.n++; }
#define TK(ci, bin) (((ci) << 8) | (bin))
HD void tk_bin(TokW &w, int ci, int bin) { tk_put(w, TK(ci, bin)); }
HD void tk_chunk(TokW &w, int v, int n) { tk_put(w, 0x8000 | (n << 8) | v); }
HD void tk_bypass(TokW &w, int v, int len) {                                                // :898-910
    v &= (1 << len) - 1;
    while (len > 0) { const int n = imin(len, 8); len -= n; tk_chunk(w, (v >> len) & ((1 << n) - 1), n); }
}

HD void mpm_list(int l, int a, int *m) {                       // :957-976
    if (l != a) { m[0] = l; m[1] = a; m[2] = (l != 0 && a != 0) ? 0 : (l + a < 2) ? 26 : 1; }
    else if (l > 1) { m[0] = l; m[1] = ((l + 29) & 31) + 2; m[2] = ((l - 1) & 31) + 2; }
    else { m[0] = 0; m[1] = 1; m[2] = 26; }
}
// prev_intra_luma_pred_flag of one PU; returns the hit index (or -1) and the candidate list in m[]
HD int tk_luma_flag(TokW &w, int ml, int ma, int mode, int *m) {
    mpm_list(ml, ma, m);
    const int hit = (m[2] == mode) ? 2 : (m[1] == mode) ? 1 : (m[0] == mode) ? 0 : -1;   // later entries win, as :992-994
    tk_bin(w, CX_PREV_INTRA, hit >= 0);
    return hit;
}
HD void tk_luma_rest(TokW &w, int *m, int hit, int mode) {     // mpm_idx / rem_intra_luma_pred_mode (:998-1016)
    if (hit >= 0) { tk_bypass(w, hit > 0, 1); if (hit > 0) tk_bypass(w, hit - 1, 1); }
    else {
        int t, r = mode;
        if (m[0] < m[1]) { t = m[0]; m[0] = m[1]; m[1] = t; }
        if (m[1] < m[2]) { t = m[1]; m[1] = m[2]; m[2] = t; }
        if (m[0] < m[1]) { t = m[0]; m[0] = m[1]; m[1] = t; }
        r -= (r > m[0]); r -= (r > m[1]); r -= (r > m[2]);
        tk_bypass(w, r, 5);
    }
}
// coding_unit header up to (not including) the first cbf_luma (:1271-1339).  shape 0: 2Nx2N one TU, 1: 2Nx2N four TUs, 2: NxN
struct CuHdr { int N, shape, ctx_split; int mode[4], ml[4], ma[4]; };
HD void tk_cu_header(TokW &w, const CuHdr &J) {
    if (J.ctx_split >= 0) tk_bin(w, J.ctx_split, 0);
    if (J.N == 8) tk_bin(w, CX_PART, J.shape != 2);
    if (J.shape == 2) {
        int m0[3], m1[3], m2[3], m3[3];
        const int h0 = tk_luma_flag(w, J.ml[0], J.ma[0], J.mode[0], m0), h1 = tk_luma_flag(w, J.ml[1], J.ma[1], J.mode[1], m1);
        const int h2 = tk_luma_flag(w, J.ml[2], J.ma[2], J.mode[2], m2), h3 = tk_luma_flag(w, J.ml[3], J.ma[3], J.mode[3], m3);
        tk_luma_rest(w, m0, h0, J.mode[0]); tk_luma_rest(w, m1, h1, J.mode[1]);
        tk_luma_rest(w, m2, h2, J.mode[2]); tk_luma_rest(w, m3, h3, J.mode[3]);
    } else {
        int m0[3];
        const int h0 = tk_luma_flag(w, J.ml[0], J.ma[0], J.mode[0], m0);
        tk_luma_rest(w, m0, h0, J.mode[0]);
    }
    tk_bin(w, CX_CHROMA_PRED, 0);
    if (J.shape != 2) tk_bin(w, CX_SPLIT_TU + (J.N == 32 ? 0 : J.N == 16 ? 1 : 2), J.shape == 1);
    tk_bin(w, CX_CBF_CHROMA, 0); tk_bin(w, CX_CBF_CHROMA, 0);
}
// last_sig_coeff_{x,y}_prefix / _suffix of a TU (:1045-1086) as straight-line code: the prefix bins of a coordinate are
// "group index" ones and a closing zero (none when the group is the largest), at most 2 log2(N) - 1 of them, so they are
// emitted by a fixed-length predicated loop; the two suffixes are adjacent bypass bins and leave as ONE chunk (<= 6 bins;
// see BitRun below for why bypass bins may be re-chunked).
struct LastPos { int gx, gy, gmax, cbase, shf, nsuf, suf; };
HD int lp_group(int t) { return t < 4 ? t : (2 * (31 - clz32((u32)t)) + ((t >> (30 - clz32((u32)t))) & 1)); }   // 0,1,2,3,4,4,5,5,6,6,6,6,7,7,7,7,8*8,9*8
HD LastPos last_pos_prep(int s, int st, int y, int x) {
    LastPos p;
    p.cbase = (s == 0) ? 0 : (s == 1) ? 3 : (s == 2) ? 6 : 10; p.shf = (s == 0) ? 0 : 1;
    const int ty = (st == 2) ? x : y, tx = (st == 2) ? y : x;       // x / y swapped for the vertical scan (:1049-1053)
    p.gx = lp_group(tx); p.gy = lp_group(ty);
    p.gmax = 2 * (s + 2) - 1;                                        // group of N-1
    const int nbx = p.gx > 3 ? (p.gx - 2) >> 1 : 0, nby = p.gy > 3 ? (p.gy - 2) >> 1 : 0;
    const int sx = tx - ((2 + (p.gx & 1)) << nbx), sy = ty - ((2 + (p.gy & 1)) << nby);
    p.nsuf = nbx + nby;
    p.suf = ((nbx ? sx : 0) << nby) | (nby ? sy : 0);
    return p;
}
HD int last_pos_count(const LastPos &p) { return p.gx + (p.gx < p.gmax) + p.gy + (p.gy < p.gmax) + (p.nsuf > 0); }
// emits tokens cnt.. ; S = log2(TU size) - 2 fixes the loop length.  PRIV as for tokg_a below.
template <int S, bool PRIV, bool HINT = false>      // HINT: fresh contexts, each used once per TU (4x4: one context per prefix bin) — the state hint is the initial state
HD int last_pos_emit(const TokOut &o, int cnt, const LastPos &p) {
    constexpr int gmax = 2 * (S + 2) - 1, shf = (S == 0) ? 0 : 1;
    static_assert(!HINT || S == 0, "state hints exist for the 4x4 PU candidates only");
    int hx[gmax], hy[gmax];                              // (HINT: the table reads before the first store, see tokg_a_fast)
    for (int i = 0; i < gmax; i++) { hx[i] = HINT ? SM.cx0[CX_LAST_X + p.cbase + (i >> shf)] : 0; hy[i] = HINT ? SM.cx0[CX_LAST_Y + p.cbase + (i >> shf)] : 0; }
    for (int i = 0; i < gmax; i++) {
        const int ci = CX_LAST_X + p.cbase + (i >> shf);
        const int pr = i <= p.gx, tok = TK(ci, i < p.gx) | hx[i] << 1;
        if (PRIV) to_put(o, cnt, tok); else to_put_if(o, cnt, tok, pr);
        cnt += pr;
    }
    for (int i = 0; i < gmax; i++) {
        const int ci = CX_LAST_Y + p.cbase + (i >> shf);
        const int pr = i <= p.gy, tok = TK(ci, i < p.gy) | hy[i] << 1;
        if (PRIV) to_put(o, cnt, tok); else to_put_if(o, cnt, tok, pr);
        cnt += pr;
    }
    { const int pr = p.nsuf > 0, tok = 0x8000 | (p.nsuf << 8) | p.suf; if (PRIV) to_put(o, cnt, tok); else to_put_if(o, cnt, tok, pr); cnt += pr; }
    return cnt;
}

// Tokens of one coefficient group (the body of the group loop of :1172-1268), as straight-line code over the group's
// 16 levels held in registers in scan order: every potential token is one LDS store to a computed slot (the dump slot
// when the token does not exist), the greater-1 context and Rice parameter recurrences are select chains.  Only
// remaining-level code words longer than 16 bins branch (tokg_b).
//   cfg : bit1 DC group | bit2 group holds the last significant coefficient | bit3 greater-1 context set carry (previous
//         coded group ended with c1 == 0) | bits4-5 neighbour pattern (below << 1 | right) | bits6-7 scan type
//         | bits8-9 log2(TU size) - 2
//   WR   : false only counts.
//   PRIV : the writer owns tb beyond its current position (lane-private staging) -> a token that does not exist may be
//          written in place and overwritten by the next one; otherwise (shared pass buffer) it goes to the dump slot.
// Part A: coded_sub_block_flag, significance flags, greater-1 / greater-2 flags, signs.  Part B: remaining levels.
enum { TG_DC = 2, TG_LAST = 4, TG_C1Z = 8, TG_PAT = 4, TG_ST = 6, TG_S = 8 };
struct Lv16 { int v[16]; };
#ifdef IMCVT_HOSTEMU
#define UNROLL_FULL
#else
#define UNROLL_FULL _Pragma("unroll")
#endif
HD int tk_chunk_word(int v, int n) { return 0x8000 | (n << 8) | v; }
// Bypass bins are coded as one run per coefficient group: sign bins and all coeff_abs_level_remaining bins of the group
// follow each other without a context-coded bin in between (:1228-1262), and the state a bypass run leaves the coder in —
// low, the bit position, the bytes shifted out — does not depend on how the run is cut into <= 8-bin CABACputBins calls
// (:898-910): every call adds range * pattern below the bits already there, and CABACupdate moves out one byte whenever
// fewer than 12 bits of headroom remain, at byte positions fixed by the total bin count.  So the run is cut into full
// 8-bin chunks (plus one remainder) instead of the reference's per-syntax-element chunks: ~35 % fewer tokens per group,
// the same bytes.  (Checked bit-for-bit against the golden vectors; the RD decisions read the coder only at trial ends.)
struct BitRun { u32 acc; int nb; };             // nb (< 8 between appends) pending bins, right-aligned in acc
// coeff_abs_level_remaining beyond the prefix-3 range: EG(k+1) escape (:1160-1167), out of line.  Appends the escape's bins to
// the run, emitting full chunks as tokens k0.. of the writer; returns the number of tokens emitted.
struct EscRet { int ntok; u32 acc; int nb; };
HDN EscRet tok_escape(TokOut o, int k0, int wr, int v, int k, u32 acc, int nb) {
    TokW w; w.o = o; w.n = k0; w.wr = wr;
    int n = k; v -= 3 << k;
    for (; v >= (1 << n); n++) v -= 1 << n;
    const int t = 4 + n - k;
    for (int part = 0; part < 2; part++) {
        int len = part ? n : t; const u32 val = part ? (u32)v : ((1u << t) - 2u);
        while (len > 0) {
            const int take = imin(len, 8 - nb); len -= take;
            acc = (acc << take) | ((val >> len) & ((1u << take) - 1u)); nb += take;
            if (nb == 8) { tk_chunk(w, (int)(acc & 0xFFu), 8); acc = 0; nb = 0; }
        }
    }
    EscRet e; e.ntok = w.n - k0; e.acc = acc; e.nb = nb;
    return e;
}
struct TgB { int esc, base2, rice, j; BitRun run; };       // state handed from part A to part B
#define TK_EMIT(pred, tok) do { const int p_ = (pred); if (WR) { if (PRIV) to_put(o, cnt, (tok)); else to_put_if(o, cnt, (tok), p_); } cnt += p_; } while (0)
// returns the token count so far | (this group ends with c1 == 0) << 16
template <bool WR, bool PRIV>
HD int tokg_a(const TokOut &o, int cnt, const Lv16 &L, u32 nzm, int cfg, TgB &B) {
    const Tables &T = SM.T;
    const int dcg = (cfg & TG_DC) != 0, has_last = (cfg & TG_LAST) != 0, pat = (cfg >> TG_PAT) & 3, st = (cfg >> TG_ST) & 3, s = (cfg >> TG_S) & 3;
    B.esc = 0; B.base2 = 3; B.rice = 0; B.j = 0; B.run.acc = 0; B.run.nb = 0;
    TK_EMIT(!dcg && !has_last, TK(CX_CSBF + (pat != 0), nzm != 0));
    if (nzm == 0 && !dcg) return cnt;
    {   // significance flags, scan positions nstart..0.  Context of position n: base + field n of a packed table (2-bit fields; 4-bit for 4x4 TUs)
        const int nstart = has_last ? hibit(nzm) : 15;
        u32 tlo, thi = 0; int base;
        if (s == 0) { const u64 t = T.c4tab[st]; tlo = (u32)t; thi = (u32)(t >> 32); base = 0; }
        else { tlo = T.posadd[pat][st]; base = 9 + (s >= 2 ? 12 : 0) + ((s == 1 && st != 0) ? 6 : 0) + (dcg ? 0 : 3); }
        UNROLL_FULL
        for (int n = 15; n >= 0; n--) {
            const int code = (n <= nstart) & !(has_last & (n == nstart)) & (dcg | (n != 0) | ((nzm >> (n + 1)) != 0));
            const int f4 = (int)(((n < 8 ? tlo : thi) >> (4 * (n & 7))) & 15), f2 = (int)((tlo >> (2 * n)) & 3);
            const int ci = (dcg && n == 0) ? 0 : base + (s == 0 ? f4 : f2);
            TK_EMIT(code, TK(CX_SIG + ci, (int)((nzm >> n) & 1)));
        }
    }
    if (nzm == 0) return cnt;
    const int nnz = popc32(nzm);
    const int set = (dcg ? 0 : 2) + ((cfg & TG_C1Z) != 0);
    int c1 = 1, g2 = -1, esc = nnz > 8, seen = 0, signs = 0;
    UNROLL_FULL
    for (int n = 15; n >= 0; n--) {                    // greater-1 flags of the first 8 non-zero levels, scan-reverse order
        const int v = L.v[n], mg = iabs(v), isnz = v != 0;
        signs = isnz ? ((signs << 1) | (v < 0)) : signs;
        const int act = isnz & (seen < 8), big = mg > 1;
        TK_EMIT(act, TK(CX_GT1 + 4 * set + c1, big));
        const int ab = act & big, an = act & !big;
        esc |= ab & (g2 >= 0);
        g2 = (ab & (g2 < 0)) ? (mg > 2) : g2;
        c1 = ab ? 0 : ((an & (c1 > 0) & (c1 < 3)) ? c1 + 1 : c1);
        seen += isnz;
    }
    TK_EMIT(g2 >= 0, TK(CX_GT2 + set, g2 > 0));
    esc |= (g2 > 0);
    {   // sign bins open the group's bypass run: full chunks leave at once, the rest waits for the remaining-level bins
        int nb = nnz;
        TK_EMIT(nb >= 8, tk_chunk_word((signs >> (nb & 7)) >> (nb >= 16 ? 8 : 0) & 0xFF, 8));
        TK_EMIT(nb >= 16, tk_chunk_word(signs & 0xFF, 8));
        nb &= 7;                                        // nnz <= 16: at most 7 bins stay pending (nnz == 16 leaves none)
        B.run.nb = nb; B.run.acc = (u32)signs & ((1u << nb) - 1u);
    }
    B.esc = esc;
    return cnt | ((g2 >= 0) ? 1 << 16 : 0);
}
// magnitude codes of a group: min(|level|, 3) of scan position n in bits 2n, 2n+1 (built by scan_levels)
#define MC_LO 0x55555555u
HD u32 mc_nz(u32 P) { return (P | (P >> 1)) & MC_LO; }        // bit 2n: level n is non-zero
HD u32 mc_big(u32 P) { return (P >> 1) & MC_LO; }             //          exceeds 1
HD u32 mc_g2(u32 P) { return P & (P >> 1) & MC_LO; }          //          exceeds 2
HD u32 mc_of(const Lv16 &L) { u32 P = 0; for (int n = 0; n < 16; n++) P |= (u32)imin(iabs(L.v[n]), 3) << (2 * n); return P; }
// Part A for a lane-private row (the hot path: pass rows of p1_run_t, lane rows of 4x4 TUs), same tokens as tokg_a<true, true>.
// The row has room for all of part A (at most 1 + 16 + 8 + 1 + 2 tokens after at most 7 staged ones), so nothing is clamped;
// a token that does not exist is written where the next existing one will land.  Positions follow from counts instead of a
// running predicate: the significance flags of scan positions top..0 are consecutive, the greater-1 flags belong to the first
// min(nnz, 8) set bits of the non-zero mask, walked with clz; contexts of the latter are min(1 + j, 3) until a level above 1
// was seen and 0 afterwards (:1218-1227).  P = the group's magnitude codes (scan_levels).
// HINT (4x4 PU candidates, priced from fresh contexts: S == 0, one DC group that holds the last position): every context-coded token
// also carries the state its context is in when the bin is coded.  From fresh contexts that state is a function of the bins coded
// earlier in THIS group on the same context, read from per-frame tables (Shm::pu_sig, pu_gt; built on the host, hevc_tables.h):
//   significance flags: a context serves at most three scan positions of a 4x4 TU (:1092) — the (at most two) coded positions above
//                       n that share its context are listed in T.c4prev, their flags index the table;
//   greater-1 flags   : contexts 1, 2, 3, 3, ... until a level above 1 was seen (every earlier bin on context 3 was a 0), context 0
//                       afterwards (any earlier bins: indexed by their count and pattern);
//   greater-2 flag    : first use.
template <int S, bool HINT = false>
HD int tokg_a_fast(u16 *tb, int cnt, const Lv16 &L, u32 nzm, u32 P, int cfg, TgB &B) {
    static_assert(!HINT || S == 0, "state hints exist for the 4x4 PU candidates only");
    const Tables &T = SM.T;
    const int dcg = (cfg & TG_DC) != 0, has_last = (cfg & TG_LAST) != 0, pat = (cfg >> TG_PAT) & 3, st = (cfg >> TG_ST) & 3;
    B.esc = 0; B.base2 = 3; B.rice = 0; B.j = 0; B.run.acc = 0; B.run.nb = 0;
    const int gt2_hint = HINT ? SM.cx0[CX_GT2] : 0;      // (HINT: one DC group, greater-1 context set 0)
    tb[cnt] = (u16)TK(CX_CSBF + (pat != 0), nzm != 0);
    cnt += (!dcg && !has_last);
    if (nzm == 0 && !dcg) return cnt;
    {   // significance flags of scan positions top..0 (the last significant position itself is not coded; position 0 is inferred
        // when nothing else of a group known to be coded is set)
        const int top = has_last ? hibit(nzm) - 1 : 15;
        u32 tlo, thi = 0; int base;
        if (S == 0) { const u64 t = T.c4tab[st]; tlo = (u32)t; thi = (u32)(t >> 32); base = 0; }
        else { tlo = T.posadd[pat][st]; base = 9 + (S >= 2 ? 12 : 0) + ((S == 1 && st != 0) ? 6 : 0) + (dcg ? 0 : 3); }
        // (all table reads of the hints come before the first token store: the compiler cannot move an LDS read above an LDS write it
        // cannot tell apart, and a read per token between the stores would be a wait per token)
        int hints[16];
        if (HINT) {
            u32 pv[4];
            for (int i = 0; i < 4; i++) pv[i] = T.c4prev[st][i];
            const u32 coded = nzm & ((2u << (top & 31)) - 1u) & (top >= 0 ? 0xFFFFu : 0u);      // the flags that are coded at all: positions 0 .. top
            const u32 inrange = (top >= 0) ? ((2u << top) - 1u) : 0u;
            UNROLL_FULL
            for (int n = 15; n >= 0; n--) {
                const int f = (int)(((n < 8 ? tlo : thi) >> (4 * (n & 7))) & 15);
                const int pp = (int)((pv[n >> 2] >> (8 * (n & 3))) & 255), p1 = pp & 15, p2 = pp >> 4;      // coded earlier on the same context: p1 (nearest), p2 > p1; 0: none
                const int k1 = (int)((inrange >> p1) & 1u) & (p1 != 0), k2 = (int)((inrange >> p2) & 1u) & (p2 != 0);
                const int b1 = (int)((coded >> p1) & 1u) & k1, b2 = (int)((coded >> p2) & 1u) & k2;
                hints[n] = SM.pu_sig[8 * f + k1 + b1 + 2 * (k2 + b2)];                               // 0 | 1 + b1 | 3 + 2 b2 + b1
            }
        }
        u16 *const lo = tb + cnt, *const hi = lo + top;
        UNROLL_FULL
        for (int n = 15; n >= 0; n--) {
            const int f = S == 0 ? (int)(((n < 8 ? tlo : thi) >> (4 * (n & 7))) & 15) : (int)((tlo >> (2 * n)) & 3);
            const int ci = (n == 0 && dcg) ? 0 : base + f;
            u16 *p = hi - n;
            p = p < lo ? lo : p;
            *p = (u16)(TK(CX_SIG + ci, (int)((nzm >> n) & 1)) | (HINT ? hints[n] << 1 : 0));
        }
        cnt += top + 1 - ((top >= 0) & !(dcg | ((nzm >> 1) != 0)));
    }
    if (nzm == 0) return cnt;
    const u32 nzs = mc_nz(P), bigs = mc_big(P), g2s = mc_g2(P);
    const int nnz = popc32(nzm), m8 = imin(nnz, 8);
    const int set = (dcg ? 0 : 2) + ((cfg & TG_C1Z) != 0);
    u32 rem = nzs;
    {   // greater-1 flags of the first 8 non-zero levels
        const int K = TK(CX_GT1 + 4 * set, 0);
        u16 *const g0 = tb + cnt, *const gend = g0 + m8;
        int ghints[8];
        if (HINT) {                                         // (as above: the eight table reads first)
            u32 rem2 = nzs; int seen2 = 0, hidx = 8;        // hidx: 8 + (1 << bins coded on context 0 so far) - 1 + their pattern
            UNROLL_FULL
            for (int j = 0; j < 8; j++) {
                const int p2 = 31 - clz_nz(rem2 | 1u);
                const int bigj = (int)((bigs >> p2) & 1u);
                ghints[j] = SM.pu_gt[seen2 ? hidx : j];
                hidx = seen2 ? 2 * hidx - 7 + bigj : hidx;  // (8 + m) -> 8 + 2 m + 1 + bin
                seen2 |= bigj;
                rem2 &= (1u << p2) - 1u;
            }
        }
        int seenbig = 0;
        UNROLL_FULL
        for (int j = 0; j < 8; j++) {
            const int p2 = 31 - clz_nz(rem | 1u);
            const int bigj = (int)((bigs >> p2) & 1u);
            u16 *p = g0 + j;
            p = p > gend ? gend : p;
            *p = (u16)(K + ((seenbig ? 0 : (j < 2 ? j + 1 : 3)) << 8) + (HINT ? ghints[j] << 1 : 0) + bigj);
            seenbig |= bigj;
            rem &= (1u << p2) - 1u;
        }
        cnt += m8;
    }
    const u32 big8 = bigs & (nzs ^ rem);                  // levels above 1 among the first 8 non-zero ones
    const int anybig = big8 != 0, fb = 31 - clz_nz(big8 | 1u);
    const int g2 = (int)((g2s >> fb) & 1u) & anybig;     // greater-2 flag of the first of them (:1232-1238)
    tb[cnt] = (u16)(TK(CX_GT2 + set, g2) | (HINT ? gt2_hint << 1 : 0));
    cnt += anybig;
    int signs = 0;
    UNROLL_FULL
    for (int n = 15; n >= 0; n--) { const int v = L.v[n]; signs = (signs << (int)((nzs >> (2 * n)) & 1u)) | (int)((u32)v >> 31); }
    {   // sign bins open the group's bypass run: full chunks leave at once, the rest waits for the remaining-level bins
        int nb = nnz;
        tb[cnt] = (u16)tk_chunk_word((signs >> (nb & 7)) >> (nb >= 16 ? 8 : 0) & 0xFF, 8); cnt += nb >= 8;
        tb[cnt] = (u16)tk_chunk_word(signs & 0xFF, 8); cnt += nb >= 16;
        nb &= 7;
        B.run.nb = nb; B.run.acc = (u32)signs & ((1u << nb) - 1u);
    }
    B.esc = (nnz > 8) | (popc32(big8) > 1) | g2;
    return cnt | (anybig << 16);
}
// remaining absolute levels of scan positions hi..lo (:1243-1262), appended to the group's bypass run; returns the token count so far.
// Both binarisations are "l1 ones, a zero, the low m bits of val" (:1150-1167):
//     r <  3 << rice :  l1 = r >> rice,         m = rice, val = r                                  (at most 7 bins)
//     r >= 3 << rice :  l1 = 3 + n - rice,      m = n,    val = r - (2 << rice),  n = floor(log2 val)   (the EG(rice+1) escape)
// so one straight-line append serves both; with at most 7 bins pending, a code word of up to 16 bins leaves as (up to) two
// full chunks here.  Longer ones (levels beyond ~100 at Rice parameter 0; measured faster than a third chunk slot on the
// synthetic and the natural test pictures) take the out-of-line path.
#ifndef TOKB_INLINE_BINS
#define TOKB_INLINE_BINS 23     // pending + code word bins handled inline: two full chunks + 7 pending (tests build with 14: every escape out of line)
#endif
template <bool WR, bool PRIV, int HI, int LO>
HD int tokg_b(const TokOut &o, int cnt, const Lv16 &L, TgB &B) {
    if (B.esc) {
        int base2 = B.base2, rice = B.rice, j = B.j;
        u32 acc = B.run.acc; int nb = B.run.nb;
        UNROLL_FULL
        for (int n = HI; n >= LO; n--) {
            const int mg = iabs(L.v[n]), isnz = mg != 0;
            const int r = mg - (j < 8 ? base2 : 1);
            const int doit = isnz & (r >= 0), small = r < (3 << rice);
            const int v2 = r - (2 << rice), ne = 31 - clz_nz((u32)v2 | 1u);      // (ne is only used when r >= 3 << rice, where v2 >= 1 << rice)
            const int m = small ? rice : ne, val = small ? r : v2;
            const int l1 = small ? (r >> rice) : 3 + ne - rice;
            const int len = doit ? l1 + m + 1 : 0;
            const int wide = nb + len > TOKB_INLINE_BINS;                          // never for the short form (at most 7 + 7 bins)
            const int inl = (len != 0) & !wide;
            const int s1 = inl ? l1 : 0, s2 = inl ? m + 1 : 0;
            acc = (acc << s1) | ((1u << s1) - 1u);                                 // bits above the nb pending ones are stale (already emitted) and never looked at
            acc = (acc << s2) | (inl ? (u32)val & ((1u << m) - 1u) : 0u);
            nb += s1 + s2;
            TK_EMIT(nb >= 8, tk_chunk_word((int)(acc >> ((nb - 8) & 31)) & 0xFF, 8));
            TK_EMIT(nb >= 16, tk_chunk_word((int)(acc >> ((nb - 16) & 31)) & 0xFF, 8));
            nb &= 7;
            if (WAVE_ANY(wide)) { if (wide) { const EscRet e = tok_escape(o, cnt, WR, r, rice, acc, nb); cnt += e.ntok; acc = e.acc; nb = e.nb; } }
            rice = (doit & (mg > (3 << rice))) ? imin(rice + 1, 4) : rice;
            base2 = (isnz & (mg >= 2)) ? 2 : base2;
            j += isnz;
        }
        B.base2 = base2; B.rice = rice; B.j = j; B.run.acc = acc; B.run.nb = nb;
    }
    return cnt;
}
// the run's last, partial chunk
template <bool WR, bool PRIV>
HD int tokg_end(const TokOut &o, int cnt, TgB &B) {
    TK_EMIT(B.run.nb > 0, tk_chunk_word((int)(B.run.acc & ((1u << B.run.nb) - 1u)), B.run.nb));
    return cnt;
}
#undef TK_EMIT
// one group into the shared pass buffer / count only: tokens k0.. of the writer; returns the count | c1-zero flag << 16
#if defined(IMCVT_MARK) && !defined(IMCVT_HOSTEMU)
#define HD_RARE HDN      // (-DIMCVT_MARK compiles: the two fallbacks of an overflowing token row — 0.003 % of the groups — stay out of the regions' static counts)
#else
#define HD_RARE HD
#endif
HD_RARE int tok_count(const Lv16 &L, u32 nzm, int cfg) {
    TokOut o; o.tb = (u16 *)0; o.pos = 0; o.cap = 0; o.glob = 0; TgB B;
    const int ra = tokg_a<false, false>(o, 0, L, nzm, cfg, B);
    return tokg_end<false, false>(o, tokg_b<false, false, 15, 0>(o, ra & 0xFFFF, L, B), B) | (ra & ~0xFFFF);
}
HD_RARE int tok_write(u16 *p, int k0, const Lv16 &L, u32 nzm, int cfg) {      // straight into the candidate's stream in global memory
    TokOut o; o.tb = p; o.pos = 0; o.cap = 0; o.glob = 1; TgB B;
    const int ra = tokg_a<true, false>(o, k0, L, nzm, cfg, B);
    return tokg_end<true, false>(o, tokg_b<true, false, 15, 0>(o, ra & 0xFFFF, L, B), B) | (ra & ~0xFFFF);
}

// ---- lane-private token streams (CU headers, 4x4 TUs): the lane stages up to LCAP tokens in its own LDS row and hands
// whole blocks to its candidate's stream in global memory.
#define LCAP 64
#define LSTRIDE LSTRIDE_DW  // dwords per lane row: 64 tokens + the dump slot; odd -> rows start in different banks
struct LaneStream { u16 *buf; u16 *g; int blk0; };      // buf[0] is token 8 * blk0 of the stream g
HD u16 *lane_row(WaveMem &W, int lane) { return (u16 *)(W.u.raw + lane * LSTRIDE); }
HD void blk_copy(u32a *d, const u32a *s_) { d[0] = s_[0]; d[1] = s_[1]; d[2] = s_[2]; d[3] = s_[3]; }
HD void blk_idle(u32a *d) { const u32 w2 = TOK_IDLE | TOK_IDLE << 16; d[0] = w2; d[1] = w2; d[2] = w2; d[3] = w2; }
HD TokW ls_begin(LaneStream &s, WaveMem &W, int c, u16 *buf, u16 *g) {
    const int n0 = W.tokn[c];
    s.buf = buf; s.g = g; s.blk0 = n0 >> 3;
    blk_copy((u32a *)buf, (const u32a *)W.pend[c]);
    TokW w; w.o.tb = buf; w.o.pos = 0; w.o.cap = LCAP; w.o.glob = 0; w.n = n0 & 7; w.wr = 1;
    return w;
}
HD void ls_out(const LaneStream &s, int nb) {
    NOUNROLL
    for (int b = 0; b < nb; b++) { const u32a *r = (const u32a *)(s.buf + 8 * b); U4 v; v.x = r[0]; v.y = r[1]; v.z = r[2]; v.w = r[3]; g_st128(s.g + 8 * (s.blk0 + b), v); }
}
HD void ls_flush(LaneStream &s, TokW &w) {              // full blocks leave, the partial one moves to the front
    const int nf = w.n >> 3;
    ls_out(s, nf);
    if (nf) blk_copy((u32a *)s.buf, (const u32a *)(s.buf + 8 * nf));
    s.blk0 += nf; w.n &= 7;
}
HD void ls_end(LaneStream &s, TokW &w, WaveMem &W, int c) {
    for (int i = 0; i < 8; i++) to_put(w.o, w.n + i, (int)TOK_IDLE);       // idle tokens up to (and beyond) the block boundary; needs w.n + 7 < LCAP
    ls_out(s, (w.n + 7) >> 3);
    blk_copy((u32a *)W.pend[c], (const u32a *)(s.buf + 8 * (w.n >> 3)));     // the partial block, or eight idle tokens
    W.tokn[c] = 8 * s.blk0 + w.n;
}

// the 16 levels of a 4x4 group (raster x[r][c]) in scan order `st`, and their non-zero mask
// *mc receives min(|level|, 3) of scan position n in bits 2n, 2n+1 ("magnitude codes": the flags of part A are bit tricks on them)
HD u32 scan_levels(Lv16 &L, const int x[4][4], int st, int fixed_diag, u32 *mc) {
    const int diag[16] = { 0, 4, 1, 8, 5, 2, 12, 9, 6, 3, 13, 10, 7, 14, 11, 15 };      // T.incg[0]
    u32 nzm = 0, P = 0;
    for (int n = 0; n < 16; n++) {
        const int d = x[diag[n] >> 2][diag[n] & 3];
        const int v = fixed_diag ? d : (st == 0 ? d : st == 1 ? x[n >> 2][n & 3] : x[n & 3][n >> 2]);
        L.v[n] = v; nzm |= (u32)(v != 0) << n;
        P |= (u32)imin(iabs(v), 3) << (2 * n);
    }
    *mc = P;
    return nzm;
}

// -DIMCVT_MARK: comment markers in the ISA at the pipeline's step boundaries (tools/isa_regions.py counts the instructions
// between them); nothing otherwise
#if defined(IMCVT_MARK) && !defined(IMCVT_HOSTEMU)
#define MARK(x) asm volatile("; MARK " x ::: "memory")
#else
#define MARK(x)
#endif
// Region counters (-DIMCVT_REGCNT builds, tools/valu_dyn_mix.py): how often each marked region of the pipeline, the token generators and the stream coders is
// executed — wave executions, counted by lane 0 at the region's END marker.  With the static opcode histogram of the region (a -DIMCVT_MARK compile) this gives the
// DYNAMIC opcode mix of the kernel.  MARKR(name, r): region r of p1_run_t<LG> (one counter per transform size s = LG - 2); MARKQ(name, id): a region with one counter.
#if defined(IMCVT_MARK) && !defined(IMCVT_HOSTEMU)
#define MARKB(x) asm volatile("; MARK " x " begin" ::: "memory")      // where a run of regions begins: what lies before it in the text is not theirs
#define MARKR(x, r) asm volatile("; MARK " x " s%c0" : : "n"(s) : "memory")
#define MARKQ(x, id) asm volatile("; MARK " x " s0" ::: "memory")
#else
#define MARKB(x) do {} while (0)
#define MARKR(x, r) RCNT(8 + 4 * (r) + s)
#define MARKQ(x, id) RCNT(id)
#endif
// ---- 4x4 blocks: one lane owns the whole block, so the pipeline runs entirely in registers (DST constants are
// immediates, no LDS intermediates, no wave syncs between the stages) and the lane writes the TU's tokens itself.
HD void p1_run_4(int wave, const P1Args &P) {
    WaveMem &W = WM(wave);
    const Tables &T = SM.T;
    const int ncand = (P.only_mode >= 0) ? 1 : NMODE;
    const QConst Q = qconst<0>(P.q);
    LANES(l) {
        const int c = l;
        if (c < ncand) {
            const int mode = (P.only_mode >= 0) ? P.only_mode : c;
            MARKB("b4");
            BorderRef br; fill_border_ref(br, W, P.per_mode_border, c);
            int pr[4][4], x[4][4], t[4][4];
            MARKQ("b4_setup", 0);
            long long t4 = prof_now();
            pred_block4(T, br, 4, 2, mode, 0, 0, pr);
            MARKQ("b4_predict", 1);
            for (int yi = 0; yi < 4; yi++) {
                const u32 ow = *(const u32a *)&SM.org[P.y0 + yi][P.x0];
                for (int xi = 0; xi < 4; xi++) x[yi][xi] = (int)((ow >> (8 * xi)) & 255) - pr[yi][xi];
            }
            // forward DST (:391-396): t = (D*x + 1) >> 1 ; coef sums = t*D^T + 128 (shift by 8 inside rdoq_group)
            for (int j = 0; j < 4; j++) {
                const int a = x[0][j], b = x[1][j], cc_ = x[2][j], d = x[3][j];
                t[0][j] = (29 * a + 55 * b + 74 * cc_ + 84 * d + 1) >> 1;
                t[1][j] = (74 * (a + b - d) + 1) >> 1;
                t[2][j] = (84 * a - 29 * b - 74 * cc_ + 55 * d + 1) >> 1;
                t[3][j] = (55 * a - 84 * b + 74 * cc_ - 29 * d + 1) >> 1;
            }
            for (int i = 0; i < 4; i++) {
                const int a = t[i][0], b = t[i][1], cc_ = t[i][2], d = t[i][3];
                x[i][0] = 29 * a + 55 * b + 74 * cc_ + 84 * d + 128;
                x[i][1] = 74 * (a + b - d) + 128;
                x[i][2] = 84 * a - 29 * b - 74 * cc_ + 55 * d + 128;
                x[i][3] = 55 * a - 84 * b + 74 * cc_ - 29 * d + 128;
            }
            MARKQ("b4_residual_dst", 2);
            prof_add(PF_T_HDR, t4); t4 = prof_now();        // (4x4 pipeline, IMCVT_PROF builds: t_hdr = predict + DST, passA = RDOQ, p2_8 on wave 2 = tokens, recon = inverse + SSE)
            const int any = rdoq_group<0>(x, Q);
            MARKQ("b4_rdoq", 3);
            prof_add(PF_T_GEN, t4); t4 = prof_now();
            const int st = scan_type_of(4, mode);
            if (P.tok) {                                    // the TU's tokens: cbf_luma, last position, the one group
                Lv16 L; u32 nzm = 0, mcode = 0;
                if (any) nzm = scan_levels(L, x, st, 0, &mcode);
                LaneStream ls;
                TokW w = ls_begin(ls, W, c, lane_row(W, l), P.tok + (size_t)c * TOK_CAP);
                tk_bin(w, CX_CBF_LUMA + (P.shape == 0 ? 1 : 0), nzm != 0);
                if (nzm != 0) {
                    const int in = T.incg[st][hibit(nzm)];
                    const LastPos lp = last_pos_prep(0, st, in >> 2, in & 3);
                    w.n = P.hint ? last_pos_emit<0, true, true>(w.o, w.n, lp) : last_pos_emit<0, true>(w.o, w.n, lp);      // PU candidates: tokens with state hints (priced by code_token_r)
                    TgB B;                                  // (at most 7 + 1 + 7 slots are in use here and part A touches 28 more: the row holds 64)
                    const int cfg4 = TG_DC | TG_LAST | st << TG_ST;
                    w.n = (P.hint ? tokg_a_fast<0, true>(w.o.tb, w.n, L, nzm, mcode, cfg4, B) : tokg_a_fast<0>(w.o.tb, w.n, L, nzm, mcode, cfg4, B)) & 0xFFFF;
                    if (B.esc) {
                        ls_flush(ls, w);                    // <= 7 tokens stay staged; eight levels add at most 8 x 5 chunks (escape code words of 32 bins)
                        w.n = tokg_b<true, true, 15, 8>(w.o, w.n, L, B);
                        if (w.n > 14) ls_flush(ls, w);      // (rarely: 14 + 8 x 5 + the closing chunk + 8 idle tokens still fit the row)
                        w.n = tokg_b<true, true, 7, 0>(w.o, w.n, L, B);
                    }
                    w.n = tokg_end<true, true>(w.o, w.n, B);
                } else if (P.shape == 3) w.n = P.hint ? last_pos_emit<0, true, true>(w.o, w.n, last_pos_prep(0, st, 0, 0)) : last_pos_emit<0, true>(w.o, w.n, last_pos_prep(0, st, 0, 0));   // PU pricing codes the residual syntax of an all-zero block (:1515)
                ls_end(ls, w, W, c);
                W.tnz[c] = (u8)(nzm != 0);
            }
            MARKQ("b4_tokens", 4);
            prof_add(threadIdx_wave() == 2 ? PF_P2_8 : PF_T_NTOK, t4); t4 = prof_now();
            int part = 0;
            if (any) {
                for (int r = 0; r < 4; r++) for (int cc = 0; cc < 4; cc++) x[r][cc] = clip16(x[r][cc] * (1 << Q.dqs));
                // inverse DST with the 16-bit clips (:511-515): t = clip16((D^T*x + 64) >> 7) ; r = clip16((t*D + 2048) >> 12)
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
                for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) x[i][j] = 0;   // all-zero levels reconstruct to the prediction
            }
            u32 oww[4];                                     // (the source rows are read before the first store below: an LDS read behind an LDS write waits for it)
            for (int yi = 0; yi < 4; yi++) oww[yi] = *(const u32a *)&SM.org[P.y0 + yi][P.x0];
            for (int yi = 0; yi < 4; yi++) {
                const u32 ow = oww[yi];
                u32 rw4 = 0;
                for (int xi = 0; xi < 4; xi++) {
                    const int rc = clip3(x[yi][xi] + pr[yi][xi], 0, 255);
                    const int d = (int)((ow >> (8 * xi)) & 255) - rc;
                    part += d * d;
                    rw4 |= (u32)rc << (8 * xi);
                    if (xi == 3 && P.rec8) *(u32a *)(P.rec8 + c * 64 + ((P.k >> 1) * 4 + yi) * 8 + (P.k & 1) * 4) = rw4;
                    if (P.out_kind == OUT_REC4) { if (xi == 3) *(u32a *)&W.u.w2.rec4[c][yi * 4] = rw4; }
                    else if (P.out_kind == OUT_TILE) SM.rec[P.y0 + yi + 1][P.x0 + xi + 1] = (u8)rc;
                    else if (P.out_kind == OUT_T3SIDE) {
                        if (P.k < 3 && yi == 3) SM.X.t3row[c][P.k][xi] = (u8)rc;
                        if (P.k < 3 && xi == 3) SM.X.t3col[c][P.k][yi] = (u8)rc;
                    }
                }
            }
            if (P.only_mode < 0) W.sse[c] += part;         // this lane is the only writer of sse[c] in this pass
            MARKQ("b4_inverse_recon_sse", 5);
            prof_add(threadIdx_wave() == 2 ? PF_RECON : PF_T_NDRAIN, t4);
        }
    }
    wave_sync_lds();
}

// ---- the same pass split over two wavefronts (wide workgroups, the PU candidates of an 8x8 CU: shape 3, OUT_REC4, state hints) --------
// what part A of a group's tokens hands to part B (tokg_a_fast's B), from the levels alone
HD TgB tokg_b_state(const Lv16 &L, u32 nzm, u32 P) {
    TgB B; B.base2 = 3; B.rice = 0; B.j = 0;
    const u32 nzs = mc_nz(P), bigs = mc_big(P), g2s = mc_g2(P);
    const int nnz = popc32(nzm);
    u32 rem = nzs;
    UNROLL_FULL
    for (int j = 0; j < 8; j++) { const int p2 = 31 - clz_nz(rem | 1u); rem &= (1u << p2) - 1u; }
    const u32 big8 = bigs & (nzs ^ rem);
    const int anybig = big8 != 0, fb = 31 - clz_nz(big8 | 1u);
    const int g2 = (int)((g2s >> fb) & 1u) & anybig;
    int signs = 0;
    UNROLL_FULL
    for (int n = 15; n >= 0; n--) { const int v = L.v[n]; signs = (signs << (int)((nzs >> (2 * n)) & 1u)) | (int)((u32)v >> 31); }
    const int nb = nnz & 7;
    B.run.nb = nb; B.run.acc = (u32)signs & ((1u << nb) - 1u);
    B.esc = (nnz > 8) | (popc32(big8) > 1) | g2;
    return B;
}
// PU wave, first half of the pass: prediction, residual, DST, RDOQ of candidate `c` (= mode); levels (raster) in x, prediction published with them
HD int pu_stage1(const WaveMem &W, const P1Args &P, int c, int x[4][4], int (*pr_out)[4] = nullptr) {
    const Tables &T = SM.T;
    const QConst Q = qconst<0>(P.q);
    BorderRef br; fill_border_ref(br, W, P.per_mode_border, c);
    int pr[4][4], t[4][4];
#if defined(IMCVT_PROF_TL)
    const int tl1 = ((PUX.pu_seq - 1u) & 3u) == 0u;      // (timeline builds: PU 1 is being walked)
#else
    const int tl1 = 0;
#endif
    pred_block4(T, br, 4, 2, c, 0, 0, pr);
    if (tl1) tl_mark(62);                                // 62: PU 1: predicted
    for (int yi = 0; yi < 4; yi++) {
        const u32 ow = *(const u32a *)&SM.org[P.y0 + yi][P.x0];
        for (int xi = 0; xi < 4; xi++) x[yi][xi] = (int)((ow >> (8 * xi)) & 255) - pr[yi][xi];
    }
    for (int j = 0; j < 4; j++) {
        const int a = x[0][j], b = x[1][j], cc_ = x[2][j], d = x[3][j];
        t[0][j] = (29 * a + 55 * b + 74 * cc_ + 84 * d + 1) >> 1;
        t[1][j] = (74 * (a + b - d) + 1) >> 1;
        t[2][j] = (84 * a - 29 * b - 74 * cc_ + 55 * d + 1) >> 1;
        t[3][j] = (55 * a - 84 * b + 74 * cc_ - 29 * d + 1) >> 1;
    }
    for (int i = 0; i < 4; i++) {
        const int a = t[i][0], b = t[i][1], cc_ = t[i][2], d = t[i][3];
        x[i][0] = 29 * a + 55 * b + 74 * cc_ + 84 * d + 128;
        x[i][1] = 74 * (a + b - d) + 128;
        x[i][2] = 84 * a - 29 * b - 74 * cc_ + 55 * d + 128;
        x[i][3] = 55 * a - 84 * b + 74 * cc_ - 29 * d + 128;
    }
    if (tl1) tl_mark(63);                                // 63: PU 1: transformed
    const int any = rdoq_group<0>(x, Q);
    if (tl1) tl_mark(64);                                // 64: PU 1: quantised
    u32 *pv = PUX.lev[c];
    for (int r = 0; r < 4; r++) {
        pv[2 * r] = any ? ((u32)(x[r][0] & 0xFFFF) | (u32)x[r][1] << 16) : 0u;
        pv[2 * r + 1] = any ? ((u32)(x[r][2] & 0xFFFF) | (u32)x[r][3] << 16) : 0u;
        pv[8 + r] = (u32)pr[r][0] | (u32)pr[r][1] << 8 | (u32)pr[r][2] << 16 | (u32)pr[r][3] << 24;
    }
    if (pr_out) for (int r = 0; r < 4; r++) for (int cc = 0; cc < 4; cc++) pr_out[r][cc] = pr[r][cc];
    return any;
}
HD void pu_levels(int c, int x[4][4], int pr[4][4]) {           // a partner reads what pu_stage1 published
    const u32 *pv = PUX.lev[c];
    for (int r = 0; r < 4; r++) {
        const u32 a = pv[2 * r], b = pv[2 * r + 1], pw = pv[8 + r];
        x[r][0] = lo16(a); x[r][1] = hi16(a); x[r][2] = lo16(b); x[r][3] = hi16(b);
        for (int xi = 0; xi < 4; xi++) pr[r][xi] = (int)((pw >> (8 * xi)) & 255);
    }
}
// partner (wave 7): the remaining-level tokens of every candidate of the PU with sequence number `seq` into its row, padded to a token block with idle tokens
HD void pu_part_b(u32 seq) {
    PuX &U = PUX;
    while ((u32)lds_ld_i32((const i32 *)&U.pu_seq) != seq) pipe_pause();
    wave_sync_lds();
    const long long tb = prof_now();                    // (IMCVT_PROF builds: booked in wave 2's row, column `idle`)
    LANES(l) {
        const int c = l;
        if (c < NMODE) {
            int x[4][4], pr[4][4];
            pu_levels(c, x, pr);
            const int st = scan_type_of(4, c);
            Lv16 L; u32 mcode = 0;
            const u32 nzm = scan_levels(L, x, st, 0, &mcode);
            TokOut o; o.tb = U.brow[c]; o.pos = 0; o.cap = BROW_CAP + 8; o.glob = 0;
            int cnt = 0;
            if (nzm != 0) {
                TgB B = tokg_b_state(L, nzm, mcode);
                cnt = tokg_end<true, true>(o, tokg_b<true, true, 15, 0>(o, 0, L, B), B);
            }
            for (int i = 0; i < 8; i++) to_put(o, cnt + i, (int)TOK_IDLE);
            U.bcnt[c] = cnt;
        }
    }
    wave_sync_lds();
    prof_add_row(2, PF_CTUIO, tb);
    tl_mark(28 + (int)((seq - 1u) & 3u));               // 28 .. 31: PU k's remaining-level rows made
    LANES(l) { if (l == 0) lds_st_i32((i32 *)&U.b_seq, (i32)seq); }
}
// partner (wave 6, wave 7 for the last PU): dequantisation, inverse DST, reconstruction and SSE of every candidate (p1_run_4's tail) into the PU wave's arrays
HD void pu_recon(int own, const P1Args &P, u32 seq) {
    WaveMem &W = WM(own);
    const QConst Q = qconst<0>(P.q);
    PuX &U = PUX;
    while ((u32)lds_ld_i32((const i32 *)&U.pu_seq) != seq) pipe_pause();
    wave_sync_lds();
    LANES(l) {
        const int c = l;
        if (c < NMODE) {
            int x[4][4], pr[4][4], t[4][4], part = 0, any = 0;
            pu_levels(c, x, pr);
            for (int r = 0; r < 4; r++) for (int cc = 0; cc < 4; cc++) any |= x[r][cc];
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
            }
            u32 oww[4];
            for (int yi = 0; yi < 4; yi++) oww[yi] = *(const u32a *)&SM.org[P.y0 + yi][P.x0];
            for (int yi = 0; yi < 4; yi++) {
                const u32 ow = oww[yi];
                u32 rw4 = 0;
                for (int xi = 0; xi < 4; xi++) {
                    const int rc = clip3(x[yi][xi] + pr[yi][xi], 0, 255);
                    const int d = (int)((ow >> (8 * xi)) & 255) - rc;
                    part += d * d;
                    rw4 |= (u32)rc << (8 * xi);
                }
                *(u32a *)&W.u.w2.rec4[c][yi * 4] = rw4;
            }
            W.sse[c] += part;
        }
    }
    wave_sync_lds();
    tl_mark(32 + (int)((seq - 1u) & 3u));               // 32 .. 35: PU k's reconstructions and SSE made
    LANES(l) { if (l == 0) lds_st_i32((i32 *)&U.r_seq, (i32)seq); }
}

// some level among the first 8 non-zero ones (coding order: scan positions 15 .. 0) exceeds 1 — the flag tokg_a reports as
// "this group ends with c1 == 0"; the next coded group's greater-1 context set depends on it (:1218-1221)
// = the highest-placed level above 1 has fewer than 8 non-zero levels before it (in coding order)
HD int group_big(u32 P) {
    const u32 bigs = mc_big(P);
    const int fb = 31 - clz_nz(bigs | 1u);
    return (bigs != 0) & (popc32(mc_nz(P) >> fb) <= 8);
}
// A lane's staged group tokens row[0..c) to tokens o.. of a stream (dst = stream base, 16-byte aligned): dword stores, a
// 16-bit store at an odd start and for an odd tail.
#ifndef ROWCAP
#define ROWCAP 53           // tokens per lane row (measured on syn q0: 0.003 % of the groups need more); slot 53 is the dump slot
#endif
#define ROWSTRIDE 27         // dwords: odd, so the lanes' rows start in different LDS banks; 64 rows = 6912 bytes of the 7168-byte pass buffer
// slot header rows (cbf_luma + 2 x (2 log2 N - 1) prefix bins + one suffix chunk + dump slot) live in W.pend, which only
// the lane-private streams of 4x4 TUs use: 16 x 7, 4 x 9 or 1 x 11 dwords <= 144 dwords
#define HDRCAP_S(S) (2 + 2 * (2 * ((S) + 2) - 1))
#define HDRSTRIDE_S(S) ((HDRCAP_S(S) + 1 + 1) / 2 | 1)
HD void row_to_stream(const u16 *row, u16 *dst, int o, int c) {
    const u32a *rw = (const u32a *)row;
    const int odd = o & 1;
    if (odd && c > 0) g_st16((i16 *)(dst + o), row[0]);
    const int nd = (c - odd) >> 1;                      // whole dwords after the odd head
    u32a *d32 = (u32a *)(dst + o + odd);
    u32 lo = rw[0];
    NOUNROLL
    for (int j = 0; j < nd; j++) {
        const u32 hi = rw[j + 1];                       // (the row has a slot to spare: reading one dword ahead stays inside it)
        g_st32(d32 + j, odd ? (lo >> 16) | (hi << 16) : lo);
        lo = hi;
    }
    if (c > odd && ((c - odd) & 1)) g_st16((i16 *)(dst + o + c - 1), row[c - 1]);
}
// exclusive suffix sum of v over the lanes of this lane's segment of `lpc` consecutive lanes (wave collective)
HD int seg_suffix_sum(int v, int l, int lpc, int *total) {
    int inc = v;
    const int r = l % lpc;
    for (int d = 1; d < lpc; d <<= 1) { const int t = wave_shfl(inc, l + d); if (r + d < lpc) inc += t; }
    *total = wave_shfl(inc, l - r);
    return inc - v;
}

template <int LG>
HD void p1_run_t(int wave, const P1Args &P) {
    constexpr int N = 1 << LG, s = LG - 2, nb = N >> 2, lpc = nb * nb, G = 64 / lpc, NN = N * N;
    WaveMem &W = WM(wave);
    const Tables &T = SM.T;
    const i8 *C = T.C + mat_off(s);
    WaveMem &WO = WM(P.own);
    const int ncand = P.c_hi;
    constexpr int a1 = s + 1, ra = 1 << a1 >> 1, rb = 1 << (a1 + 7) >> 1;
    const QConst Q = qconst<s>(P.q);
    // LDS layout of a pass, chosen against bank conflicts (wave64 b64 / b128 reads are served 32 / 16 lanes at a time over 64 banks):
    // the candidates' tiles are a few dwords apart from a multiple of 64 (8x8: 16 tiles, otherwise all on the same banks), and
    // the rows of the stage outputs of 16x16 / 32x32 tiles — whose lanes read block-rows 64 .. 512 dwords apart — are stored with
    // the 4-element column index XOR-ed by the block-row (writers and readers of a row agree on its block-row).
    constexpr int TR = NN + (N == 8 ? 4 : N == 16 ? 16 : 0), TT = NN + (N == 8 ? 4 : 0), TI = NN + (N == 8 ? 4 : N == 16 ? 4 : 0);
    static_assert(G * TR * 2 <= P1_RES_BYTES && G * TT * 4 <= 7168 - P1_RES_BYTES && G * TI * 2 <= G * TT * 4, "pass buffer");
    // matrix-core passes (N = 16, 32; see mx_mm above).  Pass buffer: the coefficients as i32 rows of CST dwords per candidate
    // (stage 2 -> group lanes), later the dequantised levels as two byte-limb tiles per candidate, column-major, DCS bytes per
    // column (group lanes -> inverse stage 1).  The strides keep the 16-lane groups of a b128 access / the 32 lanes of a b32 access on distinct banks.
    constexpr bool MX = P1_MFMA && LG >= 4;
    constexpr int MI = MX ? N - 1 : 63, MHS = MX ? LG : 6;                    // lane -> (i, h) of the matrix layout
    constexpr int CST = N + 4, CTILE = N * CST, DCS = N + 4, DTILE = N * DCS;
    static_assert(!MX || (G * CTILE * 4 <= 7168 && G * 2 * DTILE <= 7168), "pass buffer (matrix-core layout)");

    LANES(l) {
      // loop-invariant operands of the matrix-core passes: this lane's strips of the source column (as O - 128), column i of C
      // restricted to the lane's rows (the inverse stages' operand) and 128 x the sum of that whole column (limb bias)
      const int mi = l & MI, mh = l >> MHS;
      u32 oq[4] = { 0, 0, 0, 0 }, ccol[4] = { 0, 0, 0, 0 };
      int cs128 = 0;
      if constexpr (MX) {
          for (int d = 0; d < (LG == 5 ? 4 : 1); d++) {
              const int yb = (LG == 5 ? 8 * d : 0) + 4 * mh;
              u32 ow = 0, cw = 0;
              for (int t = 0; t < 4; t++) { ow |= (u32)SM.org[P.y0 + yb + t][P.x0 + mi] << (8 * t); cw |= (u32)(u8)C[(yb + t) * N + mi] << (8 * t); }
              oq[d] = ow ^ 0x80808080u; ccol[d] = cw; cs128 += sum_bytes_i8(cw);
          }
          if constexpr (LG == 4) { oq[1] = oq[2] = oq[3] = oq[0]; ccol[1] = ccol[2] = ccol[3] = ccol[0]; cs128 += wave_shfl(cs128, l ^ 16); }
          cs128 += wave_shfl(cs128, l ^ 32);
          cs128 *= 128;
      }
    NOUNROLL
    for (int c0 = P.c_lo; c0 < ncand; c0 += G) {
      {
        MARKB("pass");
        // lane <-> coefficient group: slot sl of this pass, group of scan rank r of that candidate's TU
        const int sl = l / lpc, r = l % lpc, c = c0 + sl, live = c < ncand;
        const int mode = (P.only_mode >= 0) ? P.only_mode : (live ? c : 0);
        const int st = scan_type_of(N, mode);
        const int gp = cg_pos(st, s, r), by = gp >> 3, bx = gp & 7;
        const int sw = (N >= 16) ? (by & (nb - 1)) << 2 : 0;
        i16 *const rt = W.u.p1.res + sl * TR; i32 *const tt = W.u.p1.tmp + sl * TT; i16 *const it = (i16 *)W.u.p1.tmp + sl * TI;
        const int tokn0 = (P.tok && live) ? WO.tokn[c] : 0;
        u32 predw[4] = { 0, 0, 0, 0 };                  // this lane's 4x4 block of the prediction, a packed row per dword (kept in registers until step 5)
        MARKR("pass_setup", 0);
        u32 pq[4] = { 0, 0, 0, 0 };                     // matrix-core passes: this lane's strips of the prediction (a sample per byte)
        if constexpr (MX) {
            // ---- steps 1-3a on the matrix cores: prediction in the operand layout, tmp^T = res^T C^T, coef^T = C tmp^T (limbs)
            const int nlive = imin(G, ncand - c0);
            u32 crow[4], pn[4];
            for (int d = 0; d < 4; d++) {
                const int sc_ = (LG == 5) ? 0 : d, cm = c0 + sc_;
                crow[d] = *(const u32a *)(C + mi * N + (LG == 5 ? 8 * d : 0) + 4 * mh);
                if (sc_ < nlive) {
                    const int md = (P.only_mode >= 0) ? P.only_mode : cm;
                    BorderRef br; fill_border_ref(br, W, P.per_mode_border, cm);
                    pq[d] = pred_strip4(T, br, N, LG, md, (LG == 5 ? 8 * d : 0) + 4 * mh, mi);
                }
                pn[d] = pq[d] ^ 0x7F7F7F7Fu;
            }
            MARKR("mx_predict", 1);
            int acc[16];
            const int i1 = ra + (0x8080 << a1) + (mi == 0 ? 64 * N : 0);
            for (int r4 = 0; r4 < 16; r4++) acc[r4] = i1;
            mx_mm<LG>(acc, oq, crow, nlive);
            mx_mm<LG>(acc, pn, crow, nlive);
            for (int r4 = 0; r4 < 16; r4++) acc[r4] >>= a1;                              // tmp + 0x8080
            u32 l0[4], l1[4], l2[4];
            mx_limbs3(acc, l0, l1, l2);
            for (int r4 = 0; r4 < 16; r4++) acc[r4] = 0;
            mx_mm<LG>(acc, crow, l2, nlive);
            for (int r4 = 0; r4 < 16; r4++) acc[r4] = (int)((u32)acc[r4] << 8);
            mx_mm<LG>(acc, crow, l1, nlive);
            for (int r4 = 0; r4 < 16; r4++) acc[r4] = (int)(((u32)acc[r4] << 8) + (u32)rb);
            mx_mm<LG>(acc, crow, l0, nlive);
            i32 *const ct = (i32 *)W.u.raw;
            for (int d = 0; d < 4; d++) {
                int4 o; o.x = acc[4 * d]; o.y = acc[4 * d + 1]; o.z = acc[4 * d + 2]; o.w = acc[4 * d + 3];
                if (LG == 5) *(int4 *)(ct + mi * CST + 8 * d + 4 * mh) = o;
                else if (d < nlive) *(int4 *)(ct + d * CTILE + mi * CST + 4 * mh) = o;
            }
            MARKR("mx_forward", 2);
        } else {
        // ---- step 1: prediction and residual
        if (live) {
            i16 *rp = rt;
            BorderRef br; fill_border_ref(br, W, P.per_mode_border, c);
            int pr[4][4];
            pred_block4(T, br, N, LG, mode, by * 4, bx * 4, pr);
            MARKR("predict", 3);
            for (int yi = 0; yi < 4; yi++) {
                const int y = by * 4 + yi;
                const u32 ow = *(const u32a *)&SM.org[P.y0 + y][P.x0 + bx * 4];
                predw[yi] = (u32)pr[yi][0] | (u32)pr[yi][1] << 8 | (u32)pr[yi][2] << 16 | (u32)pr[yi][3] << 24;
                uint2 rw_;
                rw_.x = (u32)(((int)(ow & 255) - pr[yi][0]) & 0xFFFF) | (u32)((int)((ow >> 8) & 255) - pr[yi][1]) << 16;
                rw_.y = (u32)(((int)((ow >> 16) & 255) - pr[yi][2]) & 0xFFFF) | (u32)((int)(ow >> 24) - pr[yi][3]) << 16;
                *(uint2 *)(rp + y * N + bx * 4) = rw_;
            }
        }
        wave_sync_lds();
        MARKR("residual", 4);
        // ---- step 2: tmp = (C * res + ra) >> a                                              (:514 forward)
        if (live) {
            int acc[4][4];
            for (int r4 = 0; r4 < 4; r4++) for (int cc = 0; cc < 4; cc++) acc[r4][cc] = ra;
            mac_MX<N, false>(acc, C, rt, by * 4, bx * 4);
            i32 *tp = tt;
            for (int r4 = 0; r4 < 4; r4++) {
                int4 o; o.x = acc[r4][0] >> a1; o.y = acc[r4][1] >> a1; o.z = acc[r4][2] >> a1; o.w = acc[r4][3] >> a1;
                *(int4 *)(tp + (by * 4 + r4) * N + ((bx * 4) ^ sw)) = o;
            }
        }
        }
        wave_sync_lds();
        MARKR("fwd_stage1", 5);
        // ---- step 3: coef = (tmp * C^T + rb) >> b ; RDOQ ; tokens ; dequantise                 (:515, :540-614, :1172-1268)
        {
            int acc[4][4];
            int any = 0;
            if (live) {
                if constexpr (MX) {
                    const i32 *ct = (const i32 *)W.u.raw + sl * CTILE + (by * 4) * CST + bx * 4;
                    for (int r4 = 0; r4 < 4; r4++) { const int4 o = *(const int4 *)(ct + r4 * CST); acc[r4][0] = o.x; acc[r4][1] = o.y; acc[r4][2] = o.z; acc[r4][3] = o.w; }
                } else {
                for (int r4 = 0; r4 < 4; r4++) for (int cc = 0; cc < 4; cc++) acc[r4][cc] = rb;
                mac_YM32<N>(acc, tt, C, by * 4, bx * 4, sw);
                }
                MARKR("fwd_stage2", 6);
                any = rdoq_group<s>(acc, Q);
                MARKR("rdoq", 7);
            }
            Lv16 L; u32 nzm = 0, mcode = 0;
            if (P.tok) { if (live && any) nzm = scan_levels(L, acc, st, N >= 16, &mcode); else for (int n = 0; n < 16; n++) L.v[n] = 0; }
            const u64 cm = P.tok ? wave_ballot(any) : 0;
            uint2 dq[4];                                    // dequantised levels, packed; stored once the token buffer (which lives in res/tmp) is done with
            for (int r4 = 0; r4 < 4; r4++)
                for (int cc = 0; cc < 4; cc++) acc[r4][cc] = any ? clip16(acc[r4][cc] * (1 << Q.dqs)) : 0;      // :613 (a shift; levels may be negative)
            if constexpr (MX) { for (int cc = 0; cc < 4; cc++) mx_limbs2(acc[0][cc], acc[1][cc], acc[2][cc], acc[3][cc], dq[cc].x, dq[cc].y); }      // per column: low / high byte limbs of its four rows
            else for (int r4 = 0; r4 < 4; r4++) { dq[r4].x = (u32)(acc[r4][0] & 0xFFFF) | (u32)acc[r4][1] << 16; dq[r4].y = (u32)(acc[r4][2] & 0xFFFF) | (u32)acc[r4][3] << 16; }
            if (P.tok) {
                const int sb = sl * lpc;
                const u64 seg = (lpc == 64) ? cm : ((cm >> sb) & ((1ull << (lpc & 63)) - 1));
                const u64 above = (r == 63) ? 0ull : (seg >> (r + 1));
                const int has_last = any && above == 0;
                // tokens exist for the groups up to the last coded one; an all-zero TU is just cbf_luma = 0, written by the DC lane
                const int talk = live && (any || above != 0 || r == 0);
                int cfg = st << TG_ST | s << TG_S | (r == 0 ? TG_DC : 0) | (has_last ? TG_LAST : 0);
                if (talk && seg != 0) {
                    const int right = (bx < nb - 1) ? (int)((seg >> cg_rank(st, s, by * 8 + bx + 1)) & 1) : 0;
                    const int below = (by < nb - 1) ? (int)((seg >> cg_rank(st, s, (by + 1) * 8 + bx)) & 1) : 0;
                    cfg |= (below << 1 | right) << TG_PAT;
                }
                u16 *base = P.tok + (size_t)c * TOK_CAP + tokn0;
                const int cbf_ctx = CX_CBF_LUMA + (P.shape == 0 ? 1 : 0);
                MARKR("scan_dequant_cfg", 8);
                const long long ptk0 = prof_now();
                // greater-1 context set carry (:1218-1221): needs only the levels, not the tokens
                const int big = (talk && nzm != 0) ? group_big(mcode) : 0;
                const u64 bmask = wave_ballot(big);
                if (above != 0 && ((bmask >> (sb + r + 1 + ctz64(above))) & 1)) cfg |= TG_C1Z;
                // TU header (cbf_luma, last position) is written by the lane of the last coded group, first in coding order
                const int lead = talk && seg != 0 && has_last;
                const int lin = lead ? T.incg[st][hibit(nzm)] : 0;
                const LastPos lp = last_pos_prep(s, st, by * 4 + (lin >> 2), bx * 4 + (lin & 3));
                const int hdr = lead ? 1 + last_pos_count(lp) : 0;
                // ONE pass: every lane writes its group's tokens into its own LDS row (res/tmp are dead here) and learns their
                // number; a suffix sum over the candidate's lanes gives the place in the stream; rows leave as dword stores.
                u16 *row = (u16 *)(W.u.raw + l * ROWSTRIDE);          // the whole pass buffer is dead here: the prediction sits in registers
                wave_sync_lds();                                            // every lane is done reading tmp
                int cg = 0;
                if (talk) {
                    TokOut o; o.tb = row; o.pos = 0; o.cap = ROWCAP; o.glob = 0;
                    if (seg == 0) { to_put(o, 0, TK(cbf_ctx, 0)); cg = 1; }
                    else { TgB B; const int ra = tokg_a_fast<s>(row, 0, L, nzm, mcode, cfg, B); cg = tokg_end<true, true>(o, tokg_b<true, true, 15, 0>(o, ra & 0xFFFF, L, B), B); }
                }
                MARKR("group_tokens", 9);
                prof_add(PF_T_SETUP, ptk0);
                const long long ptk1 = prof_now();
                const int fits = wave_ballot(cg > ROWCAP) == 0;
                int cnt = hdr + cg;
                if (!fits) {                                                // a row overflowed (escape-heavy group): count the plain way
                    cnt = 0;
                    if (talk) cnt = (seg == 0) ? 1 : hdr + (tok_count(L, nzm, cfg) & 0xFFFF);
                }
                int total;
                const int off = seg_suffix_sum(cnt, l, lpc, &total);
                prof_add(PF_T_GEN, ptk1);
                const long long ptk2 = prof_now();
                if (talk) {
                    TokW w; w.o.tb = base + off; w.o.pos = 0; w.o.cap = 0; w.o.glob = 1; w.n = 0; w.wr = 1;
                    if (fits) {
                        if (lead) {                                         // header tokens: staged in the slot's header row, like the group rows
                            TokOut ho; ho.tb = (u16 *)((u32a *)&W.pend[0][0] + sl * HDRSTRIDE_S(s)); ho.pos = 0; ho.cap = HDRCAP_S(s); ho.glob = 0;
                            to_put(ho, 0, TK(cbf_ctx, 1));
                            last_pos_emit<s, true>(ho, 1, lp);
                            row_to_stream(ho.tb, P.tok + (size_t)c * TOK_CAP, tokn0 + off, hdr);
                        }
                        row_to_stream(row, P.tok + (size_t)c * TOK_CAP, tokn0 + off + hdr, cg); w.n = hdr + cg;
                    } else if (seg == 0) tk_bin(w, cbf_ctx, 0);
                    else {
                        if (lead) { tk_bin(w, cbf_ctx, 1); w.n = last_pos_emit<s, false>(w.o, w.n, lp); }
                        w.n = tok_write(w.o.tb, w.n, L, nzm, cfg) & 0xFFFF;
                    }
                    if (r == 0) {                                           // the DC lane, last in coding order, pads the final block with idle tokens
                        const int e7 = (tokn0 + total) & 7;
                        for (int i = 0; i < 7; i++) to_put_if(w.o, w.n + i, (int)TOK_IDLE, e7 != 0 && e7 + i < 8);
                        WO.tokn[c] = tokn0 + total; WO.tnz[c] = (u8)(seg != 0);
                    }
                }
                MARKR("tokens_to_stream", 10);
                prof_add(PF_T_HDR, ptk2);
                wave_sync_lds();                                            // rows are done with before res is written again
            }
            if constexpr (MX) {
                if (!P.tok) wave_sync_lds();                // the limb tiles lie over the coefficient tiles: every lane has read its group
                if (live) {
                    u8 *dt = (u8 *)W.u.raw + sl * 2 * DTILE + by * 4;
                    for (int cc = 0; cc < 4; cc++) { *(u32a *)(dt + (bx * 4 + cc) * DCS) = dq[cc].x; *(u32a *)(dt + DTILE + (bx * 4 + cc) * DCS) = dq[cc].y; }
                }
            } else
            if (live) {
                i16 *dp = rt;
                for (int r4 = 0; r4 < 4; r4++) *(uint2 *)(dp + (by * 4 + r4) * N + bx * 4) = dq[r4];
            }
        }
        wave_sync_lds();
        MARKR("dequant_store", 11);
        if constexpr (MX) {
            // ---- steps 4-5 on the matrix cores: itmp = clip16((deq^T C + 64) >> 7) as [y][v], rec = clip16((itmp C + 2048) >> 12) as [y][x = i]
            const int nlive = imin(G, ncand - c0);
            u32 dl[4], dh[4];
            for (int d = 0; d < 4; d++) {
                const u8 *dt = (const u8 *)W.u.raw + (LG == 5 ? 0 : d * 2 * DTILE) + mi * DCS + (LG == 5 ? 8 * d : 0) + 4 * mh;
                const int on = (LG == 5) || d < nlive;
                dl[d] = on ? *(const u32a *)dt : 0u; dh[d] = on ? *(const u32a *)(dt + DTILE) : 0u;
            }
            int acc[16];
            for (int r4 = 0; r4 < 16; r4++) acc[r4] = 0;
            mx_mm<LG>(acc, dh, ccol, nlive);
            for (int r4 = 0; r4 < 16; r4++) acc[r4] = (int)(((u32)acc[r4] << 8) + (u32)(64 + cs128));
            mx_mm<LG>(acc, dl, ccol, nlive);
            u32 il[4], ih[4];
            for (int d = 0; d < 4; d++) mx_limbs2(clip16(acc[4 * d] >> 7), clip16(acc[4 * d + 1] >> 7), clip16(acc[4 * d + 2] >> 7), clip16(acc[4 * d + 3] >> 7), il[d], ih[d]);
            for (int r4 = 0; r4 < 16; r4++) acc[r4] = 0;
            mx_mm<LG>(acc, ih, ccol, nlive);
            for (int r4 = 0; r4 < 16; r4++) acc[r4] = (int)(((u32)acc[r4] << 8) + (u32)(2048 + cs128));
            mx_mm<LG>(acc, il, ccol, nlive);
            MARKR("mx_inverse", 12);
            int part = 0;
            for (int d = 0; d < 4; d++) {
                const int sc_ = (LG == 5) ? 0 : d, cm = c0 + sc_;
                if (sc_ < nlive) {
                    const u32 ow = oq[d] ^ 0x80808080u, pw = pq[d];
                    if (LG == 4) part = 0;
                    for (int t = 0; t < 4; t++) {
                        const int y = (LG == 5 ? 8 * d : 0) + 4 * mh + t, x = mi;
                        const int rc = clip3(clip16(acc[4 * d + t] >> 12) + (int)((pw >> (8 * t)) & 255), 0, 255);
                        const int dd = (int)((ow >> (8 * t)) & 255) - rc;
                        part += dd * dd;
                        if (P.out_kind == OUT_TILE) SM.rec[P.y0 + y + 1][P.x0 + x + 1] = (u8)rc;
                        else if (P.out_kind == OUT_T3SIDE) {
                            if (P.k < 3 && y == N - 1) SM.X.t3row[cm][P.k][x] = (u8)rc;
                            if (P.k < 3 && x == N - 1) SM.X.t3col[cm][P.k][y] = (u8)rc;
                        }
                    }
                    if (LG == 4 && P.only_mode < 0) lds_add(&WO.sse[cm], part);
                }
            }
            if (LG == 5 && P.only_mode < 0) lds_add(&WO.sse[c0], part);
        } else {
        // ---- step 4: itmp = clip16((C^T * deq + 64) >> 7)                                    (:514 inverse)
        if (live) {
            int acc[4][4];
            for (int r4 = 0; r4 < 4; r4++) for (int cc = 0; cc < 4; cc++) acc[r4][cc] = 64;
            mac_MX<N, true>(acc, C, rt, by * 4, bx * 4);
            i16 *ip = it;                                // tmp (i32) was last read in step 3; reuse it as i16
            for (int r4 = 0; r4 < 4; r4++) {
                uint2 o;
                o.x = (u32)(clip16(acc[r4][0] >> 7) & 0xFFFF) | (u32)clip16(acc[r4][1] >> 7) << 16;
                o.y = (u32)(clip16(acc[r4][2] >> 7) & 0xFFFF) | (u32)clip16(acc[r4][3] >> 7) << 16;
                *(uint2 *)(ip + (by * 4 + r4) * N + ((bx * 4) ^ sw)) = o;
            }
        }
        wave_sync_lds();
        MARKR("inv_stage1", 13);
        // ---- step 5: rec = clip8(clip16((itmp * C + 2048) >> 12) + pred) ; SSE                (:515 inverse, :146,:165)
        if (live) {
            int acc[4][4];
            for (int r4 = 0; r4 < 4; r4++) for (int cc = 0; cc < 4; cc++) acc[r4][cc] = 2048;
            mac_YM16<N>(acc, it, C, by * 4, bx * 4, sw);
            int part = 0;
            for (int r4 = 0; r4 < 4; r4++) {
                const int y = by * 4 + r4;
                const u32 pw = predw[r4];
                const u32 ow = *(const u32a *)&SM.org[P.y0 + y][P.x0 + bx * 4];
                u32 rw4 = 0;
                for (int cc = 0; cc < 4; cc++) {
                    const int x = bx * 4 + cc;
                    const int rc = clip3(clip16(acc[r4][cc] >> 12) + (int)((pw >> (8 * cc)) & 255), 0, 255);
                    const int d = (int)((ow >> (8 * cc)) & 255) - rc;
                    part += d * d;
                    rw4 |= (u32)rc << (8 * cc);
                    if (P.out_kind == OUT_TILE) SM.rec[P.y0 + y + 1][P.x0 + x + 1] = (u8)rc;
                    else if (P.out_kind == OUT_T3SIDE) {
                        if (P.k < 3 && y == N - 1) SM.X.t3row[c][P.k][x] = (u8)rc;
                        if (P.k < 3 && x == N - 1) SM.X.t3col[c][P.k][y] = (u8)rc;
                    }
                }
                if (N == 8 && P.rec8) *(u32a *)(P.rec8 + c * 64 + y * 8 + bx * 4) = rw4;
            }
            if (P.only_mode < 0) lds_add(&WO.sse[c], part);
        }
        }
        MARKR("inv_stage2_recon_sse", 14);
        wave_sync_lds();
      }
    }
    }
}

HD void p1_run(int wave, const P1Args &P) {
    if (P.N == 32) p1_run_t<5>(wave, P);
    else if (P.N == 16) p1_run_t<4>(wave, P);
    else if (P.N == 8) p1_run_t<3>(wave, P);
    else p1_run_4(wave, P);
}
HDN void p1_run_cold(int wave_, const P1Args P_) {
    P1Args P;
    P.N = uni_i(P_.N); P.y0 = uni_i(P_.y0); P.x0 = uni_i(P_.x0); P.k = uni_i(P_.k); P.per_mode_border = uni_i(P_.per_mode_border); P.out_kind = uni_i(P_.out_kind);
    P.only_mode = uni_i(P_.only_mode); P.shape = uni_i(P_.shape); P.tok = uni_p(P_.tok); P.q = uni_i(P_.q); P.own = uni_i(P_.own); P.c_lo = uni_i(P_.c_lo); P.c_hi = uni_i(P_.c_hi); P.hint = 0; P.rec8 = nullptr;
    p1_run(uni_i(wave_), P);
}     // the winner's reconstruction: once per CU, kept out of line

// ---------------------------------------------------------------------------------------------------
// Stream coding.  One lane codes one candidate's token stream with its own arithmetic coder and context copy;
// every wave step is exactly one token per live lane.  Bit-exactness: same bins, same order, same <=8-bin
// bypass chunking as :898-1268 (the tokens), same coder arithmetic as :858-932.
// ---------------------------------------------------------------------------------------------------

// one token on the lane's coder
template <class S>
HD void code_token(Arith &a, u8 *cx, S &sink, u32 tok) {
    if (tok & 0x8000u) {                                                            // bypass chunk, :898-910
        const int nb_ = (int)((tok >> 8) & 15u);
        a.low = (a.low << nb_) + mul24(a.range, (int)(tok & 255u));                 // range <= 510, value <= 255
        a.nbits -= nb_;
    } else {                                                                        // context-coded bin, :913-932
        const int ci = (int)(tok >> 8), bin = (int)(tok & 1u);
        const int pz = cx[ci];
        const uint2 e = SM.T.pst[pz];
        const int lps = (int)((e.x >> (((a.range >> 6) & 3) * 8)) & 0xFF);
        const int rm = a.range - lps;
        const int is_lps = (bin ^ pz) & 1;
        const int sh = is_lps ? imin(6, clz32((u32)lps) - 23) : (rm < 256);     // renorm table :714 == 8 - floor(log2 lps), capped at 6
        cx[ci] = (u8)(is_lps ? e.y : e.y >> 8);
        a.low = (a.low + (is_lps ? rm : 0)) << sh;
        a.range = (is_lps ? lps : rm) << sh;
        a.nbits -= sh;
    }
    carry_out(a, sink);
}
// The same for the trial coders, as straight-line code (no divergent branch): a bypass chunk is a "context" bin on the
// row's pad byte with its own shift and an idle lane's token is a 0-bin chunk.  The arithmetic step does not touch the
// byte-level state at all: when a byte leaves `low` (:858-862) its 9-bit lead (carry + byte) is queued in LDS, and
// the carry / 0xFF-run / emulation-prevention logic of :863-878,:820-831 runs once per queued lead after the block
// (lead_step below) — about one byte per eight tokens, so that logic costs an eighth of what it did per token.
#define CX_PAD (CTX_STRIDE - 1)
HD void code_token_q(Arith &a, u8 *cx, u16 *lq, int &qn, u32 tok) {
    const int byp = tok >= 0x8000u;
    const u32 cim = tok >> 8;
    const int ci = (int)(cim < (u32)CX_PAD ? cim : (u32)CX_PAD);
    const int pz = cx[ci];                                                        // < 128 everywhere: states are 7 bits, and the pad byte starts as 0 (ctx_init) and only ever receives next-state bytes
    const uint2 e = SM.T.pst[pz];
    const int lps = (int)((e.x >> ((a.range >> 3) & 24)) & 0xFF);                 // :917-918
    const int rm = a.range - lps;
    const int is_lps = (int)(tok ^ (u32)pz) & 1;
    const int r2 = is_lps ? lps : rm;
    const int sh = clz_nz((u32)r2) - 23;                                          // renorm table :714 (lps >= 6) and the MPS shift :926 (rm >= 128) in one
    cx[ci] = (u8)(is_lps ? e.y : e.y >> 8);
    const int nb_ = byp ? (int)((tok >> 8) & 15u) : sh;
    const int add = (is_lps & !byp) ? rm : 0;
    a.low = ((a.low + add) << nb_) + mul24(a.range, byp ? (int)(tok & 255u) : 0);      // :898-910 / :921-930
    a.range = byp ? a.range : (r2 << sh);
    a.nbits -= nb_;
    const int need = a.nbits < 12;                                                // :858-862
    lq[qn & (LRING - 1)] = (u16)((u32)a.low >> ((24 - a.nbits) & 31));                          // always written; only kept when `need`
    qn += need;
    a.nbits += need ? 8 : 0;
    a.low = need ? (i32)((u32)a.low & (0xFFFFFFFFu >> a.nbits)) : a.low;
}
// The same on RESOLVED tokens (context-coded tokens that carry the state their context is in, see "Bin tokens"): no context
// copy is read or written; `lw` = the four LPS ranges of the token's state (T.pst[(tok >> 1) & 127].x), whose address depends on
// the token alone — the caller loads the words of a whole token block ahead of its steps, so nothing on the step's dependent
// chain waits for LDS.
HD void code_token_r(Arith &a, u16 *lq, int &qn, u32 tok, u32 lw) {
    const int byp = tok >= 0x8000u;
    const int lps = (int)((lw >> ((a.range >> 3) & 24)) & 0xFF);                    // :917-918
    const int rm = a.range - lps;
    const int is_lps = (int)(tok ^ (tok >> 1)) & 1;                                // bin ^ MPS of the hinted state
    const int r2 = is_lps ? lps : rm;
    const int sh = clz_nz((u32)r2) - 23;
    const int nb_ = byp ? (int)((tok >> 8) & 15u) : sh;
    const int add = (is_lps & !byp) ? rm : 0;
    a.low = ((a.low + add) << nb_) + mul24(a.range, byp ? (int)(tok & 255u) : 0);
    a.range = byp ? a.range : (r2 << sh);
    a.nbits -= nb_;
    const int need = a.nbits < 12;
    lq[qn & (LRING - 1)] = (u16)((u32)a.low >> ((24 - a.nbits) & 31));
    qn += need;
    a.nbits += need ? 8 : 0;
    a.low = need ? (i32)((u32)a.low & (0xFFFFFFFFu >> a.nbits)) : a.low;
}
template <class S>
HD void lead_step(Arith &a, S &sink, int lead) {                                  // :863-878
    const int v1 = (a.bufbyte + (lead >> 8)) & 0xFF;
    if (a.nbytes == 1 && lead != 0xFF && !(a.zeros >= 2 && v1 <= 3)) {
        sink_put(sink, a.cnt++, v1);
        a.zeros = v1 ? 0 : a.zeros + 1;
        a.bufbyte = lead & 0xFF;
    } else carry_rare(a, sink, lead);
}
HD u32 tok_of(const U4 &b, int j) { const u32 w = (j < 2) ? b.x : (j < 4) ? b.y : (j < 6) ? b.z : b.w; return (j & 1) ? (w >> 16) : (w & 0xFFFFu); }

// Code tokens p[0..n) (global memory, 16-byte aligned); the leads of the bytes that leave `low` go to the lane's lead sink (LeadSink), `qn` counts them.
// Wave collective: every lane calls it, idle lanes with n == 0.  All lanes are in the same phase of their streams, so
// the token loads (one 16-byte block per lane per 8 steps, issued one block ahead) and the lead flushes are wave-synchronous.
// (stream_seg: one segment of a stream on a sink that outlives it — the pipe wave codes a CU's stream in pieces as they become known)
// RES: the tokens are resolved (code_token_r); cx is not used
template <bool RES>
HD void stream_seg_t(Arith &a, u8 *cx, LeadSink &sink, int &qn, const u16 *p, int n) {
    // token blocks are loaded unconditionally (index clamped to the stream's last block; p is always a valid address),
    // so the loop carries no conditional load and the only wait for a block is where it is first used, one round later
    const int last_blk = imax((n - 1) >> 3, 0);
    const long long tx2 = prof_now();
    U4 cur = g_ld128(p);
#if defined(IMCVT_PROF) && !defined(IMCVT_HOSTEMU)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(cur.x), "+v"(cur.y), "+v"(cur.z), "+v"(cur.w) : : "memory");
#endif
    prof_add(PF_X2, tx2);
    NOUNROLL
    for (int k0 = 0; WAVE_ANY(k0 < n); k0 += 8) {           // wave-uniform: one 16-byte block of tokens per lane per round
        MARKB("p2");
        const u16 *pn = p + 8 * imin((k0 >> 3) + 1, last_blk);
#ifdef IMCVT_HOSTEMU
        const U4 nxt = g_ld128(pn);
#else
        // issued and waited for by hand: the compiler's own wait placement would drain this load before the token steps
        u32x4 nv;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(nv) : "v"(pn) : "memory");
#endif
        if (k0 < n) {                                       // lanes without a stream (or past its end) sit out: their cx / ring rows belong to lane 0
            const long long tp0 = prof_now();
            lsink_sync(sink, qn);                           // full 16-byte runs of leads leave the ring
            MARKQ("p2_ring_sync", 72);
            prof_add(PF_T_NDRAIN, tp0);
            const long long tp1 = prof_now();
            if constexpr (RES) {
                u32 lw[8];
                UNROLL_FULL
                for (int j = 0; j < 8; j++) lw[j] = SM.T.pst[(tok_of(cur, j) >> 1) & 127u].x;
                UNROLL_FULL
                for (int j = 0; j < 8; j++) code_token_r(a, sink.ring, qn, tok_of(cur, j), lw[j]);
            } else {
            UNROLL_FULL
            for (int j = 0; j < 8; j++)                     // no VMEM instruction in here; the stream's last block is padded with idle tokens
                code_token_q(a, cx, sink.ring, qn, tok_of(cur, j));
            }
            MARKQ("p2_eight_tokens", 73);
            prof_add(PF_T_NTOK, tp1); prof_cnt(PF_BORDER, 1);
        }
#ifdef IMCVT_HOSTEMU
        cur = nxt;
#else
        const long long tp3 = prof_now();
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(nv) : : "memory");
        prof_add(PF_T_SETUP, tp3);
        cur.x = nv.x; cur.y = nv.y; cur.z = nv.z; cur.w = nv.w;
#endif
    }
}
HD void stream_seg(Arith &a, u8 *cx, LeadSink &sink, int &qn, const u16 *p, int n) { stream_seg_t<false>(a, cx, sink, qn, p, n); }

// ---------------------------------------------------------------------------------------------------
// The trial coder split over two wavefronts (wide workgroups).  A token step (code_token_q) is two recurrences: the RANGE side —
// context state, LPS range, renormalisation shift, new range (:913-926) — which needs nothing of `low`, and the BYTE side — low,
// the bit position, bytes leaving (:921-930 for low, :858-878) — which needs of the range side only the addend (range - LPS for
// an LPS bin), the shift and, for a bypass chunk, the range it multiplies (:898-910).  stream_seg_R runs the first on the owner's
// wavefront and leaves those numbers as one record per token; stream_seg_L, on the partner wavefront, one token block behind, runs
// the second on the records.  Same arithmetic, same order, per lane: low = ((low + add) << nb) + range * value, exactly the line of
// code_token_q — each wavefront just issues half of the instructions.
//     record = add (9 bits) | range << 9 | nb << 18 | value << 22        (context bin: value = 0; bypass chunk: add = 0)
// ---------------------------------------------------------------------------------------------------
HD u32 token_R(int &range, u8 *cx, u32 tok) {
    const int byp = tok >= 0x8000u;
    const u32 cim = tok >> 8;
    const int ci = (int)(cim < (u32)CX_PAD ? cim : (u32)CX_PAD);
    const int pz = cx[ci];
    const uint2 e = SM.T.pst[pz];
    const int lps = (int)((e.x >> ((range >> 3) & 24)) & 0xFF);
    const int rm = range - lps;
    const int is_lps = (int)(tok ^ (u32)pz) & 1;
    const int r2 = is_lps ? lps : rm;
    const int sh = clz_nz((u32)r2) - 23;
    cx[ci] = (u8)(is_lps ? e.y : e.y >> 8);
    const int nb_ = byp ? (int)((tok >> 8) & 15u) : sh;
    const int add = (is_lps & !byp) ? rm : 0;
    const u32 rec = (u32)add | (u32)range << 9 | (u32)nb_ << 18 | (byp ? (tok & 255u) << 22 : 0u);
    range = byp ? range : (r2 << sh);
    return rec;
}
// The same over a block of eight tokens with the context side run ONE TOKEN AHEAD: the context recurrence (state -> next state, :913-920) does
// not involve the range, so the state of token j + 1 and its table entry are read while token j's range arithmetic runs — BEFORE token j's
// state is written back, hence corrected when both tokens use the same context (then the state is token j's next state, known at once, and its
// table entry was read at the start of token j as well).  Same states, same bins, same order as token_R eight times; a lone wavefront just no
// longer waits two LDS round trips per token.
HD void block_R8(int &range, u8 *cx, const U4 &cur, u32 rec[8]) {
    u32 tok = tok_of(cur, 0);
    u32 cim = tok >> 8;
    int ci = (int)(cim < (u32)CX_PAD ? cim : (u32)CX_PAD);
    int pz = cx[ci];
    uint2 e = SM.T.pst[pz];
    UNROLL_FULL
    for (int j = 0; j < 8; j++) {
        const int is_lps = (int)(tok ^ (u32)pz) & 1;
        const int nx = (int)((is_lps ? e.y : e.y >> 8) & 255u);          // the state this token leaves its context in
        u32 tokn = 0; int cin = 0, pzn = 0; uint2 en, ef; en.x = en.y = ef.x = ef.y = 0;
        if (j < 7) {                                                     // next token's reads, ahead of this token's write
            tokn = tok_of(cur, j + 1);
            const u32 cimn = tokn >> 8;
            cin = (int)(cimn < (u32)CX_PAD ? cimn : (u32)CX_PAD);
            pzn = cx[cin];
            ef = SM.T.pst[nx];                                           // (its entry if it meets this token's context again)
            en = SM.T.pst[pzn];
        }
        const int byp = tok >= 0x8000u;
        const int lps = (int)((e.x >> ((range >> 3) & 24)) & 0xFF);
        const int rm = range - lps;
        const int r2 = is_lps ? lps : rm;
        const int sh = clz_nz((u32)r2) - 23;
        cx[ci] = (u8)nx;
        const int nb_ = byp ? (int)((tok >> 8) & 15u) : sh;
        const int add = (is_lps & !byp) ? rm : 0;
        rec[j] = (u32)add | (u32)range << 9 | (u32)nb_ << 18 | (byp ? (tok & 255u) << 22 : 0u);
        range = byp ? range : (r2 << sh);
        if (j < 7) {
            const int same = cin == ci;
            pz = same ? nx : pzn;
            e.x = same ? ef.x : en.x; e.y = same ? ef.y : en.y;
            tok = tokn; ci = cin;
        }
    }
}
HD u32 token_R_res(int &range, u32 tok, u32 lw) {                // resolved tokens (code_token_r): no context copy
    const int byp = tok >= 0x8000u;
    const int lps = (int)((lw >> ((range >> 3) & 24)) & 0xFF);
    const int rm = range - lps;
    const int is_lps = (int)(tok ^ (tok >> 1)) & 1;
    const int r2 = is_lps ? lps : rm;
    const int sh = clz_nz((u32)r2) - 23;
    const int nb_ = byp ? (int)((tok >> 8) & 15u) : sh;
    const int add = (is_lps & !byp) ? rm : 0;
    const u32 rec = (u32)add | (u32)range << 9 | (u32)nb_ << 18 | (byp ? (tok & 255u) << 22 : 0u);
    range = byp ? range : (r2 << sh);
    return rec;
}
// The pricing of the PU candidates (stream_seg_R_lds -> stream_seg_L1) uses a leaner record: v | nb << 17 with v = add << nb for a context
// bin, range * value for a bypass chunk — low = (low << nb) + v either way, one instruction on the byte side.
HD u32 token_R_res2(int &range, u32 tok, u32 lw) {
    // Branch-free, by masks: a select with the product in one arm is compiled into exec-mask control flow — four or five scalar
    // instructions per token, each a pipeline turn-around on a serial chain (profiles/r05_valu_sgpr.log).  A bypass chunk goes through the
    // same arithmetic with an LPS range of zero: range - 0, shift clz(range) - 23 = 0, nothing added — the range stays.
    const int bm = (int)(tok << 16) >> 31;                                      // all ones: bypass chunk
    const int lps = (int)(((lw & ~(u32)bm) >> ((range >> 3) & 24)) & 0xFF);
    const int rm = range - lps;
    const int lm = -(int)((tok ^ (tok >> 1)) & 1u) & ~bm;                        // all ones: context bin that takes the LPS path
    const int r2 = (lps & lm) | (rm & ~lm);
    const int sh = clz_nz((u32)r2) - 23;
    const int nb_ = sh + ((int)(tok >> 8) & 15 & bm);
    const int v = mul24(range, (int)(tok & 255u) & bm) + ((rm & lm) << sh);      // (rm << sh < 2^15, range * value < 2^17)
    range = r2 << sh;
    return (u32)v | (u32)nb_ << 17;
}
HD void token_L(Arith &a, u16 *lq, int &qn, u32 rec) {
    const int add = (int)(rec & 511u), rg = (int)((rec >> 9) & 511u), nb_ = (int)((rec >> 18) & 15u), val = (int)(rec >> 22);
    a.low = ((a.low + add) << nb_) + mul24(rg, val);
    a.nbits -= nb_;
    const int need = a.nbits < 12;                                                // :858-862
    lq[qn & (LRING - 1)] = (u16)((u32)a.low >> ((24 - a.nbits) & 31));
    qn += need;
    a.nbits += need ? 8 : 0;
    a.low = need ? (i32)((u32)a.low & (0xFFFFFFFFu >> a.nbits)) : a.low;
}
// the owner's lanes zero their queue counters before the partner is told to start (split_go)
HD void split_reset(SplitQ &q, int lane) { if (lane < NMODE) { q.prod[lane] = 0; q.cons[lane] = 0; } }
// Range half of tokens p[0..n): wave collective like stream_seg_t.  `blk` counts this lane's token blocks of the run (it runs on over
// the segments of one stream).
template <bool RES>
HD void stream_seg_R(int &range, u8 *cx, SplitQ &q, int lane, int &blk, const u16 *p, int n) {
    const int last_blk = imax((n - 1) >> 3, 0);
    const int ql = lane < NMODE ? lane : 0;
    u32 *const row = q.rec[ql];
    U4 cur = g_ld128(p);
    int cons_seen = lds_ld_i32(&q.cons[ql]);
    NOUNROLL
    for (int k0 = 0; WAVE_ANY(k0 < n); k0 += 8) {
        const u16 *pn = p + 8 * imin((k0 >> 3) + 1, last_blk);
#ifdef IMCVT_HOSTEMU
        const U4 nxt = g_ld128(pn);
#else
        u32x4 nv;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(nv) : "v"(pn) : "memory");
#endif
        if (k0 < n) {
            // room in the ring: the partner has read block blk - QDEPTH (re-read its counter only when the last look is too old)
            while (WAVE_ANY(blk - cons_seen >= QDEPTH)) { if (blk - cons_seen >= QDEPTH) { pipe_pause(); cons_seen = lds_ld_i32(&q.cons[ql]); } }
            u32 rec[8];
            if constexpr (RES) {
                u32 lw[8];
                UNROLL_FULL
                for (int j = 0; j < 8; j++) lw[j] = SM.T.pst[(tok_of(cur, j) >> 1) & 127u].x;
                UNROLL_FULL
                for (int j = 0; j < 8; j++) rec[j] = token_R_res(range, tok_of(cur, j), lw[j]);
            } else {
#ifdef IMCVT_R_PLAIN
                UNROLL_FULL
                for (int j = 0; j < 8; j++) rec[j] = token_R(range, cx, tok_of(cur, j));
#else
                block_R8(range, cx, cur, rec);
#endif
            }
            u32 *d = row + (blk & (QDEPTH - 1)) * 8;
            UNROLL_FULL
            for (int j = 0; j < 8; j++) d[j] = rec[j];
#ifndef IMCVT_HOSTEMU
            asm volatile("" ::: "memory");                  // the records are issued before the counter (LDS serves a wavefront's accesses in order)
#endif
            blk++;
            lds_st_i32(&q.prod[ql], blk);
        }
#ifdef IMCVT_HOSTEMU
        cur = nxt;
#else
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(nv) : : "memory");
        cur.x = nv.x; cur.y = nv.y; cur.z = nv.z; cur.w = nv.w;
#endif
    }
}
// ... the same with the tokens in LDS (the kept PU winners of a wide workgroup's NxN trial, PuX::kept): no load to wait for
HD void stream_seg_R_ldsrc(int &range, u8 *cx, SplitQ &q, int lane, int &blk, const u16 *p, int n) {
    const int last_blk = imax((n - 1) >> 3, 0);
    const int ql = lane < NMODE ? lane : 0;
    u32 *const row = q.rec[ql];
    const u32a *pw = (const u32a *)p;
    U4 cur; cur.x = pw[0]; cur.y = pw[1]; cur.z = pw[2]; cur.w = pw[3];
    int cons_seen = lds_ld_i32(&q.cons[ql]);
    NOUNROLL
    for (int k0 = 0; WAVE_ANY(k0 < n); k0 += 8) {
        const u32a *pn = pw + 4 * imin((k0 >> 3) + 1, last_blk);
        U4 nxt; nxt.x = pn[0]; nxt.y = pn[1]; nxt.z = pn[2]; nxt.w = pn[3];
        if (k0 < n) {
            while (WAVE_ANY(blk - cons_seen >= QDEPTH)) { if (blk - cons_seen >= QDEPTH) { pipe_pause(); cons_seen = lds_ld_i32(&q.cons[ql]); } }
            u32 rec[8];
            block_R8(range, cx, cur, rec);
            u32 *d = row + (blk & (QDEPTH - 1)) * 8;
            UNROLL_FULL
            for (int j = 0; j < 8; j++) d[j] = rec[j];
#ifndef IMCVT_HOSTEMU
            asm volatile("" ::: "memory");
#endif
            blk++;
            lds_st_i32(&q.prod[ql], blk);
        }
        cur = nxt;
    }
}
// ---- the range half cut once more: context stage (stream_seg_C) and range stage (stream_seg_Rq), see CtxQ ----
// context side of a token block, one token ahead like block_R8: out[2 j] = the LPS ranges of the state token j's bin meets, out[2 j + 1] = token | (LPS path) << 16
HD void block_C8(u8 *cx, const U4 &cur, u32 out[16]) {
    u32 tok = tok_of(cur, 0);
    u32 cim = tok >> 8;
    int ci = (int)(cim < (u32)CX_PAD ? cim : (u32)CX_PAD);
    int pz = cx[ci];
    uint2 e = SM.T.pst[pz];
    UNROLL_FULL
    for (int j = 0; j < 8; j++) {
        const int is_lps = (int)(tok ^ (u32)pz) & 1;
        const int nx = (int)((is_lps ? e.y : e.y >> 8) & 255u);
        u32 tokn = 0; int cin = 0, pzn = 0; uint2 en, ef; en.x = en.y = ef.x = ef.y = 0;
        if (j < 7) {
            tokn = tok_of(cur, j + 1);
            const u32 cimn = tokn >> 8;
            cin = (int)(cimn < (u32)CX_PAD ? cimn : (u32)CX_PAD);
            pzn = cx[cin];
            ef = SM.T.pst[nx];
            en = SM.T.pst[pzn];
        }
        cx[ci] = (u8)nx;
        out[2 * j] = e.x; out[2 * j + 1] = tok | (u32)is_lps << 16;
        if (j < 7) {
            const int same = cin == ci;
            pz = same ? nx : pzn;
            e.x = same ? ef.x : en.x; e.y = same ? ef.y : en.y;
            tok = tokn; ci = cin;
        }
    }
}
// The same on a context copy that holds TABLE ENTRIES (8 bytes per context: pst[state], whose bits 16..22 of .y are the state) instead of state bytes: a token's
// entry is ONE LDS read at an address the token alone gives — issued a token ahead, nothing on the chain waits for a second, dependent read (state -> entry);
// the entry of the state the token leaves behind is read off the chain and stored back (and handed on when the next token meets the same context).
HD void block_C8e(uint2 *cx8, const U4 &cur, u32 out[16]) {
    u32 tok = tok_of(cur, 0);
    u32 cim = tok >> 8;
    int ci = (int)(cim < (u32)CX_PAD ? cim : (u32)CX_PAD);
    uint2 e = cx8[ci];
    UNROLL_FULL
    for (int j = 0; j < 8; j++) {
        const int is_lps = (int)(tok ^ (e.y >> 16)) & 1;
        const int nx = (int)((is_lps ? e.y : e.y >> 8) & 255u);
        const uint2 ef = SM.T.pst[nx];
        u32 tokn = 0; int cin = 0; uint2 en; en.x = en.y = 0;
        if (j < 7) {
            tokn = tok_of(cur, j + 1);
            const u32 cimn = tokn >> 8;
            cin = (int)(cimn < (u32)CX_PAD ? cimn : (u32)CX_PAD);
            en = cx8[cin];                                                 // (read before the store below: stale when cin == ci — then ef is handed on instead)
        }
        cx8[ci] = ef;
        out[2 * j] = e.x; out[2 * j + 1] = tok | (u32)is_lps << 16;
        if (j < 7) {
            const int same = cin == ci;
            e.x = same ? ef.x : en.x; e.y = same ? ef.y : en.y;
            tok = tokn; ci = cin;
        }
    }
}
// tokens p[0..n) (LDS when lds_src, else global memory) through the context stage; `cblk` counts this lane's blocks of the run
template <bool LDS_SRC, bool E8 = false>
HD void stream_seg_C(u8 *cx, CtxQ &cq, int lane, int &cblk, const u16 *p, int n) {      // E8: cx is a copy of table entries (uint2 per context, block_C8e)
    const int last_blk = imax((n - 1) >> 3, 0);
    const int ql = lane < NMODE ? lane : 0;
    u32 *const row = cq.rec[ql];
    U4 cur;
    if constexpr (LDS_SRC) { const u32a *pw = (const u32a *)p; cur.x = pw[0]; cur.y = pw[1]; cur.z = pw[2]; cur.w = pw[3]; } else cur = g_ld128(p);
    int cons_seen = lds_ld_i32(&cq.cons[ql]);
    NOUNROLL
    for (int k0 = 0; WAVE_ANY(k0 < n); k0 += 8) {
        const u16 *pn = p + 8 * imin((k0 >> 3) + 1, last_blk);
        U4 nxt;
        if constexpr (LDS_SRC) { const u32a *pw = (const u32a *)pn; nxt.x = pw[0]; nxt.y = pw[1]; nxt.z = pw[2]; nxt.w = pw[3]; } else nxt = g_ld128(pn);
        if (k0 < n) {
            while (WAVE_ANY(cblk - cons_seen >= QDEPTH)) { if (cblk - cons_seen >= QDEPTH) { pipe_pause(); cons_seen = lds_ld_i32(&cq.cons[ql]); } }
            u32 out[16];
            if constexpr (E8) block_C8e((uint2 *)cx, cur, out); else block_C8(cx, cur, out);
            u32 *d = row + (cblk & (QDEPTH - 1)) * 16;
            UNROLL_FULL
            for (int j = 0; j < 16; j++) d[j] = out[j];
#ifndef IMCVT_HOSTEMU
            asm volatile("" ::: "memory");
#endif
            cblk++;
            lds_st_i32(&cq.prod[ql], cblk);
        }
        cur = nxt;
    }
}
// ... and the range stage over its records: the range arithmetic of block_R8 alone; leaves the byte half's records in q as stream_seg_R does
HD void stream_seg_Rq(int &range, CtxQ &cq, SplitQ &q, int lane, int &cblk, int &blk, int n) {
    const int ql = lane < NMODE ? lane : 0;
    const u32 *const crow = cq.rec[ql];
    u32 *const row = q.rec[ql];
    int prod_seen = lds_ld_i32(&cq.prod[ql]), cons_seen = lds_ld_i32(&q.cons[ql]);
    NOUNROLL
    for (int k0 = 0; WAVE_ANY(k0 < n); k0 += 8) {
        if (k0 < n) {
            while (WAVE_ANY(prod_seen <= cblk)) { if (prod_seen <= cblk) { pipe_pause(); prod_seen = lds_ld_i32(&cq.prod[ql]); } }
            while (WAVE_ANY(blk - cons_seen >= QDEPTH)) { if (blk - cons_seen >= QDEPTH) { pipe_pause(); cons_seen = lds_ld_i32(&q.cons[ql]); } }
            const u32 *in = crow + (cblk & (QDEPTH - 1)) * 16;
            u32 rec[8];
            UNROLL_FULL
            for (int j = 0; j < 8; j++) {
                const u32 ex = in[2 * j], meta = in[2 * j + 1];
                const u32 tok = meta & 0xFFFFu;
                const int is_lps = (int)(meta >> 16) & 1, byp = tok >= 0x8000u;
                const int lps = (int)((ex >> ((range >> 3) & 24)) & 0xFF);
                const int rm = range - lps;
                const int r2 = is_lps ? lps : rm;
                const int sh = clz_nz((u32)r2) - 23;
                const int nb_ = byp ? (int)((tok >> 8) & 15u) : sh;
                const int add = (is_lps & !byp) ? rm : 0;
                rec[j] = (u32)add | (u32)range << 9 | (u32)nb_ << 18 | (byp ? (tok & 255u) << 22 : 0u);
                range = byp ? range : (r2 << sh);
            }
            u32 *d = row + (blk & (QDEPTH - 1)) * 8;
            UNROLL_FULL
            for (int j = 0; j < 8; j++) d[j] = rec[j];
#ifndef IMCVT_HOSTEMU
            asm volatile("" ::: "memory");
#endif
            blk++; cblk++;
            lds_st_i32(&q.prod[ql], blk);
            lds_st_i32(&cq.cons[ql], cblk);
        }
    }
}
// The range half on RESOLVED tokens that lie in LDS (a lane's own row, or its row at a partner: hevc_frame.h pu_step_wide), 16-byte-block
// padded with idle tokens like a stream in memory.
HD void stream_seg_R_lds(int &range, SplitQ &q, int lane, int &blk, const u16 *p, int n) {
    const int last_blk = imax((n - 1) >> 3, 0);
    const int ql = lane < NMODE ? lane : 0;
    u32 *const row = q.rec[ql];
    const u32a *pw = (const u32a *)p;
    U4 cur; cur.x = pw[0]; cur.y = pw[1]; cur.z = pw[2]; cur.w = pw[3];
    int cons_seen = lds_ld_i32(&q.cons[ql]);
    NOUNROLL
    for (int k0 = 0; WAVE_ANY(k0 < n); k0 += 8) {
        const u32a *pn = pw + 4 * imin((k0 >> 3) + 1, last_blk);
        U4 nxt; nxt.x = pn[0]; nxt.y = pn[1]; nxt.z = pn[2]; nxt.w = pn[3];
        if (k0 < n) {
            while (WAVE_ANY(blk - cons_seen >= QDEPTH)) { if (blk - cons_seen >= QDEPTH) { pipe_pause(); cons_seen = lds_ld_i32(&q.cons[ql]); } }
            u32 lw[8], rec[8];
            UNROLL_FULL
            for (int j = 0; j < 8; j++) lw[j] = SM.T.pst[(tok_of(cur, j) >> 1) & 127u].x;
            UNROLL_FULL
            for (int j = 0; j < 8; j++) rec[j] = token_R_res2(range, tok_of(cur, j), lw[j]);
            u32 *d = row + (blk & (QDEPTH - 1)) * 8;
            UNROLL_FULL
            for (int j = 0; j < 8; j++) d[j] = rec[j];
#ifndef IMCVT_HOSTEMU
            asm volatile("" ::: "memory");
#endif
            blk++;
            lds_st_i32(&q.prod[ql], blk);
        }
        cur = nxt;
    }
}
// Byte half of the pricing of a PU's candidates (:1504-1518) over the lean records of stream_seg_R_lds: low and the bit position token by token,
// the leads into the lane's lead sink as everywhere (the bytes are never wanted: the sink's count and its guard give the length).
HD void lead_take(Arith &a, u16 *ring, int &qn, int nb_, int v) {
    a.low = (a.low << nb_) + v;
    a.nbits -= nb_;
    const int need = a.nbits < 12;                                                // :858-862
    ring[qn & (LRING - 1)] = (u16)((u32)a.low >> ((24 - a.nbits) & 31));          // always written; only kept when `need`
    qn += need;
    a.nbits += need ? 8 : 0;
    a.low = need ? (i32)((u32)a.low & (0xFFFFFFFFu >> a.nbits)) : a.low;
}
HD void stream_seg_L1(Arith &a, LeadSink &sink, int &qn, SplitQ &q, int lane, int &blk, int n) {
    const int ql = lane < NMODE ? lane : 0;
    const u32 *const row = q.rec[ql];
    int prod_seen = 0;
    NOUNROLL
    for (int k0 = 0; WAVE_ANY(k0 < n); k0 += 8) {
        if (k0 < n) {
            while (WAVE_ANY(prod_seen <= blk)) { if (prod_seen <= blk) { prod_seen = lds_ld_i32(&q.prod[ql]); if (prod_seen <= blk) pipe_pause(); } }
#ifndef IMCVT_HOSTEMU
            asm volatile("" ::: "memory");
#endif
            const u32 *sp = row + (blk & (QDEPTH - 1)) * 8;
            u32 rec[8];
            UNROLL_FULL
            for (int j = 0; j < 8; j++) rec[j] = sp[j];
#ifndef IMCVT_HOSTEMU
            asm volatile("" ::: "memory");
#endif
            blk++;
            lds_st_i32(&q.cons[ql], blk);
            lsink_sync(sink, qn);
            UNROLL_FULL
            for (int j = 0; j < 8; j++) lead_take(a, sink.ring, qn, (int)(rec[j] >> 17), (int)(rec[j] & 0x1FFFFu));      // (token_R_res2's records)
        }
    }
}
// The same over tokens that are ALL bypass chunks (the remaining-level rows a partner makes, pu_part_b): a bypass chunk leaves the range as it
// is (:898-910), so the range half has nothing to do there — the byte half takes the chunks from the row itself, with the range the first
// part ended on.  (An idle token is a chunk of no bins.)
HD void stream_seg_L1_byp(Arith &a, LeadSink &sink, int &qn, const u16 *p, int n, int range) {
    const u32a *pw = (const u32a *)p;
    NOUNROLL
    for (int k0 = 0; WAVE_ANY(k0 < n); k0 += 8) {
        if (k0 < n) {
            const u32a *pb = pw + (k0 >> 1);
            U4 cur; cur.x = pb[0]; cur.y = pb[1]; cur.z = pb[2]; cur.w = pb[3];
            lsink_sync(sink, qn);
            UNROLL_FULL
            for (int j = 0; j < 8; j++) {
                const u32 tok = tok_of(cur, j);
                lead_take(a, sink.ring, qn, (int)((tok >> 8) & 15u), mul24(range, (int)(tok & 255u)));
            }
        }
    }
}
// Byte half: consumes the records of n tokens of this lane.  Same trip count as the owner's stream_seg_R (same n).
HD void stream_seg_L(Arith &a, LeadSink &sink, int &qn, SplitQ &q, int lane, int &blk, int n) {
    const int ql = lane < NMODE ? lane : 0;
    const u32 *const row = q.rec[ql];
    int prod_seen = 0;
    NOUNROLL
    for (int k0 = 0; WAVE_ANY(k0 < n); k0 += 8) {
        if (k0 < n) {
            while (WAVE_ANY(prod_seen <= blk)) { if (prod_seen <= blk) { prod_seen = lds_ld_i32(&q.prod[ql]); if (prod_seen <= blk) pipe_pause(); } }
#ifndef IMCVT_HOSTEMU
            asm volatile("" ::: "memory");
#endif
            const u32 *sp = row + (blk & (QDEPTH - 1)) * 8;
            u32 rec[8];
            UNROLL_FULL
            for (int j = 0; j < 8; j++) rec[j] = sp[j];
#ifndef IMCVT_HOSTEMU
            asm volatile("" ::: "memory");                  // (the reads are issued before the counter store: the owner may overwrite the slot once it sees it)
#endif
            blk++;
            lds_st_i32(&q.cons[ql], blk);
            lsink_sync(sink, qn);
            UNROLL_FULL
            for (int j = 0; j < 8; j++) token_L(a, sink.ring, qn, rec[j]);
        }
    }
}
// What a trial ends with (every lane of the wavefront; `on`: the lane has a stream): the rest of the leads to memory, their number behind
// them, and the byte-level state — counted (bytes = leads: the state the coder started from, cnt moved on by the leads) or, for a lane
// whose leads show the emulation-prevention pattern, by the real logic over its list.
HD void leads_exact(Arith &a, const Arith &a0, const u8 *gbuf, int qn, int mine) {      // (wave collective: `mine` — this lane is one of those it is run for)
    CountSinkT cs; cs.dummy = 0;
    Arith t = a0;
    const int n = mine ? qn : 0;
    NOUNROLL
    for (int i = 0; WAVE_ANY(i < n); i++)
        if (i < n) { const int lead = (int)(u16)g_ld16((const i16 *)(gbuf + 2 * i)); lead_step(t, cs, lead); }
    if (mine) { a.cnt = t.cnt; a.nbytes = t.nbytes; a.bufbyte = t.bufbyte; a.zeros = t.zeros; }
}
HD void trial_finish(Arith &a, const Arith &a0, LeadSink &sink, int qn, int on) {
    if (on) {
        lsink_finish(sink, qn);
        g_st32(sink.gbuf + TRIAL_BYTES - 4, (u32)qn);
        a.cnt = a0.cnt + qn; a.nbytes = a0.nbytes; a.bufbyte = a0.bufbyte; a.zeros = a0.zeros;
    }
    const int ex = on & sink.hit;
    tl_count(66);                                           // (timeline builds: 66 trials finished, 65 of them with lanes on the exact path)
    if (WAVE_ANY(ex)) {
        tl_count(65);
        drain_stores();                                     // (the list is read back from memory)
        leads_exact(a, a0, sink.gbuf, qn, ex);
    }
}
// The winner's leads -> bytes (one wavefront, every lane).  `a`: the coder state the winner's trial started from, with low / range / nbits
// already those it ended with; on return its byte-level part (cnt, nbytes, bufbyte, zeros) is what the byte-level logic of :863-878, :820-831
// leaves after the n leads, and the bytes emitted on the way are at dst[a0.cnt ..).
// A lead is a digit of a long number with a carry bit on top; the bytes buffered on entry are its first digits.  The carry out of digit k is
// c_k | (v_k == 0xFF & carry out of digit k + 1): a carry-lookahead over the ballots of the two predicates — by the scalar adder, 64 digits a
// round, from the last round to the first.  What stays buffered at the end is the last digit that is not a plain 0xFF lead, and the 0xFF
// run behind it.  Emulation prevention is checked on the resolved bytes (a second pass, forward); if it would strike — two zero bytes, then
// one of 3 or less — or nothing anchors the buffer, lane 0 walks the list with the byte-level logic itself (practically never).
#ifdef IMCVT_HOSTEMU
HD u64 brev64(u64 x) { u64 r = 0; for (int i = 0; i < 64; i++) r |= ((x >> i) & 1ull) << (63 - i); return r; }
#else
HD u64 brev64(u64 x) { return __builtin_bitreverse64(x); }
#endif
HD void resolve_leads(Arith &a, const u8 *list, int n, u8 *dst) {
    const Arith a0 = a;
    const int nb0 = a0.nbytes, m = nb0 + n;                 // digits: nb0 buffered bytes (bufbyte, then 0xFF), then the leads
    const int nch = (m + 63) >> 6;
#if defined(IMCVT_RESOLVE_SERIAL) || defined(IMCVT_FORCE_OVF)          // (test builds: lane 0 walks every winner's list)
    int fallback = 1;
#else
    int fallback = 0;
#endif
    LANES(l) {
        // pass 1, last round first: carries; the resolved digits go to dst as if all were emitted (the buffered tail is overwritten later or never read)
        u32 cin = 0;
        int anchor = -1;                                    // last digit that is not a plain 0xFF lead
        NOUNROLL
        for (int ch = nch - 1; ch >= 0; ch--) {
            const int k = ch * 64 + l, live = k < m;
            int L = 0xFF;
            if (live) L = k < nb0 ? (k == 0 ? (a0.bufbyte & 0xFF) | 0x200 : 0xFF) : (int)(u16)g_ld16((const i16 *)(list + 2 * (k - nb0)));      // (0x200: the buffered byte anchors the run behind it)
            if (live && nb0 == 0 && k == 0) L = (L & 0xFF) | 0x200;             // the very first lead of a stream: it becomes the buffered byte, its carry is dropped (:876)
            const int c = (L >> 8) & 1, ff = (L & 0xFF) == 0xFF;
            const u64 G = brev64(wave_ballot(live && c)), P = brev64(wave_ballot(live && ff));
            const u64 A = G | P, S = A + G + (u64)cin;
            const u64 CI = brev64(S ^ A ^ G);               // bit l: carry INTO this lane's digit (= out of the next one)
            cin = (u32)((((A & G) | ((A | G) & ~S)) >> 63) & 1u);
            const int v = ((L & 0xFF) + (int)((CI >> l) & 1u)) & 0xFF;
            if (live) g_st8(dst + a0.cnt + k, v);
            const u64 an = wave_ballot(live && L != 0xFF);
            if (anchor < 0 && an != 0) anchor = ch * 64 + hibit64(an);
        }
        const int jt = anchor;                              // digits 0 .. jt - 1 are emitted, jt .. m - 1 stay buffered
        if (jt < 0) fallback = 1;
        else {
            // pass 2, forward: would emulation prevention have struck?  and the run of zero bytes the emitted part ends with
            wave_sync();
            u32 zprev = a0.zeros >= 2 ? 3u : a0.zeros == 1 ? 1u : 0u;      // bit 0: the byte before is zero, bit 1: the one before that
            int zrun = a0.zeros;
            NOUNROLL
            for (int ch = 0; ch * 64 < jt; ch++) {
                const int k = ch * 64 + l, live = k < jt;
                const int v = live ? (int)g_ld8(dst + a0.cnt + k) : 1;
                const u64 Z = wave_ballot(live && v == 0), T = wave_ballot(live && v <= 3);
                const u64 Z1 = Z << 1 | (zprev & 1u), Z2 = Z << 2 | (u64)(zprev & 1u) << 1 | (zprev >> 1);
                if ((T & Z1 & Z2) != 0) fallback = 1;
                const int cnt = imin(64, jt - ch * 64);
                const u64 livem = cnt >= 64 ? ~0ull : (1ull << cnt) - 1ull;
                const u64 nz = ~Z & livem;                  // non-zero emitted bytes of this round
                zrun = nz ? cnt - 1 - hibit64(nz) : zrun + cnt;
                zprev = (u32)((Z >> (cnt - 1)) & 1u) | (u32)(cnt >= 2 ? (Z >> (cnt - 2)) & 1u : (zprev & 1u)) << 1;
            }
            if (!fallback) {
                const int Lt = jt < nb0 ? a0.bufbyte : (int)(u16)g_ld16((const i16 *)(list + 2 * (jt - nb0)));
                a.cnt = a0.cnt + jt; a.nbytes = m - jt; a.bufbyte = (jt < nb0 || jt == 0) ? Lt : (Lt & 0xFF); a.zeros = zrun;      // (the buffered byte of the entry state stays as it is; a stream's very first lead is kept whole, :876)
            }
        }
        if (fallback) {                                     // lane 0, lead by lead (every lane computes the same state; one lane stores)
            Arith t = a0;
            Sink sk; sk.base = dst; sk.off = 0;
            NOUNROLL
            for (int i = 0; i < n; i++) {
                const int lead = (int)(u16)g_ld16((const i16 *)(list + 2 * i));
                if (l == 0) lead_step(t, sk, lead); else { CountSinkT cs; cs.dummy = 0; lead_step(t, cs, lead); }
            }
            a.cnt = t.cnt; a.nbytes = t.nbytes; a.bufbyte = t.bufbyte; a.zeros = t.zeros;
        }
    }
}
template <bool RES = false>
HD void stream_run(Arith &a, u8 *cx, LaneMem *lm, u8 *gbuf, const u16 *p, int n, int on) {
    const Arith a0 = a;
    LeadSink sink; lsink_begin(sink, a0, lm->ring, gbuf);
    int qn = 0;
    stream_seg_t<RES>(a, cx, sink, qn, p, n);
    const long long tx3 = prof_now();
    trial_finish(a, a0, sink, qn, on);
    prof_add(PF_X3, tx3);
}
// One trial: contexts copied from cx_src, coder state `a` in/out.  Wave collective (`on` = this lane has a stream).
#ifdef IMCVT_TOKSTAT
static long long g_tokstat[4];    // trials, tokens, lanes with a stream
#endif
HD void run_trial(Arith &a, const u8 *cx_src, u8 *cx, LaneMem *lm, u8 *gbuf, const u16 *p, int n, int on) {
#ifdef IMCVT_TOKSTAT
    if (on) { g_tokstat[1] += n; g_tokstat[2]++; if (n > g_tokstat[3]) g_tokstat[3] = n; }
#endif
    const long long tx1 = prof_now();
    if (on) for (int i = 0; i < CTX_STRIDE; i += 4) *(u32a *)(cx + i) = *(const u32a *)(cx_src + i);
    prof_add(PF_X1, tx1);                                   // (IMCVT_PROF builds: x1 = context copy, x2 = wait for the first token block, x3 = after the last block)
    stream_run(a, cx, lm, gbuf, p, on ? n : 0, on);
}
// One trial on resolved tokens (the 4x4 PU candidates: fresh contexts, state hints in the tokens): no context copy at all.
HD void run_trial_r(Arith &a, const u8 *cx_fresh, u8 *cx, LaneMem *lm, u8 *gbuf, const u16 *p, int n, int on) {
    (void)cx_fresh;
#ifdef IMCVT_TOKSTAT
    if (on) { g_tokstat[1] += n; g_tokstat[2]++; if (n > g_tokstat[3]) g_tokstat[3] = n; }
#endif
    stream_run<true>(a, cx, lm, gbuf, p, on ? n : 0, on);
}

// ---------------------------------------------------------------------------------------------------
// One stream coded on WAVE-UNIFORM values (the NxN trial of an 8x8 CU in launches without a pipe wave: one stream, nobody to
// share a wavefront with).  As lane code it is a whole pass of 72 vector instructions per token with one live lane — 9 % of the
// vector instructions of a full device — on the wave every other wave of the workgroup is waiting for.  Here every lane carries
// the same coder state and every loaded value is declared uniform (readfirstlane), so the arithmetic, the branches (plain
// code_token / carry_out: no need for the branch-free form) and the byte-level logic run on the SCALAR unit, which the other
// wavefronts of the SIMD are not queueing for; the vector pipe sees the loads, their readfirstlanes and the stores of lane 0.
// Same bins, same order, same arithmetic as :858-932.
// ---------------------------------------------------------------------------------------------------
#ifdef IMCVT_HOSTEMU
#define UNI(x) (x)
#define UNI_RUN(l) ((l) == 0)        // (the emulation's lanes are fibers that do not run in lock-step: one of them walks the stream)
#define UNI_STORE(l) 1
#else
#define UNI(x) ((int)__builtin_amdgcn_readfirstlane((int)(x)))
#define UNI_RUN(l) 1
#define UNI_STORE(l) ((l) == 0)
#endif
struct UniSink { u8 *base; u32 off; int st; };            // as Sink; st: this lane performs the stores
HD void sink_put(UniSink &s, int i, int v) { if (s.st) g_st8(s.base + (u32)(s.off + (u32)i), v); }
HD void code_token_uni(Arith &a, u8 *cx, UniSink &sink, u32 tok) {
    if (tok & 0x8000u) {                                                            // bypass chunk, :898-910
        const int nb_ = (int)((tok >> 8) & 15u);
        a.low = (a.low << nb_) + a.range * (int)(tok & 255u);
        a.nbits -= nb_;
    } else {                                                                        // context-coded bin, :913-932
        const int ci = (int)(tok >> 8), bin = (int)(tok & 1u);
        const int pz = UNI(cx[ci]);
        const int ex = UNI(SM.T.pst[pz].x), ey = UNI(SM.T.pst[pz].y);
        const int lps = (int)(((u32)ex >> (((a.range >> 6) & 3) * 8)) & 0xFF);
        const int rm = a.range - lps;
        const int is_lps = (bin ^ pz) & 1;
        const int sh = is_lps ? imin(6, clz32((u32)lps) - 23) : (rm < 256);
        if (sink.st) cx[ci] = (u8)(is_lps ? ey : ey >> 8);
        a.low = (a.low + (is_lps ? rm : 0)) << sh;
        a.range = (is_lps ? lps : rm) << sh;
        a.nbits -= sh;
    }
    carry_out(a, sink);
}
// p: 16-byte aligned, padded to a token block with idle tokens.  Wave collective; every lane ends with the same coder state.
HD void stream_run_uni(Arith &a, u8 *cx, u8 *gbuf, const u16 *p, int n, int lane) {
    UniSink sink; sink.base = gbuf; sink.off = (u32)(0 - a.cnt); sink.st = UNI_STORE(lane);
    NOUNROLL
    for (int k0 = 0; k0 < n; k0 += 8) {
        const U4 b = g_ld128(p + k0);
        const u32 w[4] = { (u32)UNI(b.x), (u32)UNI(b.y), (u32)UNI(b.z), (u32)UNI(b.w) };
        for (int j = 0; j < 8; j++) code_token_uni(a, cx, sink, (j & 1) ? (w[j >> 1] >> 16) : (w[j >> 1] & 0xFFFFu));
    }
}

