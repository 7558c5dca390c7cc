// valu_rate_probe.hip — [developer measurement tool, not product code] what one gfx950 SIMD can issue.
//
// Question ("weak" 2): the encoder kernel's lone waves issue one instruction per ~7.4 cycles and the bench line
// prices a plain 32-bit VALU instruction at 4 cycles per wave64 — is that the hardware's rate, and how much of it is dependence?
// Every kernel below is ONE instruction kind in an unrolled stream of 64, with C = 1, 2, 4 or 8 independent dependence chains
// interleaved (C = 1: every instruction reads the previous one's result), run by W waves per SIMD.  Per wave the shader clock
// (s_memtime) brackets the stream; reported: cycles per instruction as ONE wave sees it (latency view) and wave-instructions per
// cycle per SIMD (throughput view = W x 1 / that).
//
// build: hipcc --offload-arch=gfx950 -O2 tools/valu_rate_probe.hip -o tools/valu_rate_probe.bin
// run:   tools/valu_rate_probe.bin [json-out]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum Kind { K_ADD, K_LSHL, K_AND, K_CNDMASK, K_MUL24, K_MULLO, K_SDWA, K_BFE, K_FFBH, K_PERM, K_MED3, K_LSHLADD, K_CMPSEL, K_RFL, K_RFL_SALU,
            K_SALU, K_SALU_MUL, K_LDS_CHASE, K_LDS_U8, K_LDS_B64, K_LDS_WR_RD, K_GLB_CHASE, K_MOVX, K_N };
static const char *kind_name[K_N] = { "v_add_u32", "v_lshlrev_b32", "v_and_b32", "v_cndmask_b32", "v_mul_i32_i24", "v_mul_lo_u32", "v_add_u32_sdwa", "v_bfe_u32", "v_ffbh_u32",
    "v_perm_b32", "v_med3_i32", "v_lshl_add_u32", "v_cmp+v_cndmask (pair)", "v_readfirstlane+v_add (pair)", "v_readfirstlane+s_add+v_add (triple)",
    "s_add_u32", "s_mul_i32", "ds_read_b32 chase (+wait)", "ds_read_u8 chase (+wait)", "ds_read_b64 chase (+wait)", "ds_write_b32+ds_read_b32 (+wait)", "global_load_dword chase (+wait)", "v_mov_b32" };
// instructions a "unit" of the stream consists of (pairs / triples count as that many)
static const int kind_len[K_N] = { 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 3, 1, 1, 1, 1, 1, 2, 1, 1 };

// chain c of C: operand %c.  %8 = a loop-invariant VGPR, %9/%10 = SGPR operands where a kind needs them
#define R8_C1(op) op(0) op(0) op(0) op(0) op(0) op(0) op(0) op(0)
#define R8_C2(op) op(0) op(1) op(0) op(1) op(0) op(1) op(0) op(1)
#define R8_C4(op) op(0) op(1) op(2) op(3) op(0) op(1) op(2) op(3)
#define R8_C8(op) op(0) op(1) op(2) op(3) op(4) op(5) op(6) op(7)
#define R64(r8, op) r8(op) r8(op) r8(op) r8(op) r8(op) r8(op) r8(op) r8(op)
#define STREAM(r8, op) asm volatile(R64(r8, op) : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(x), "s"(sx) : "vcc", "s20", "s21", "memory")
#define BY_CHAINS(op) do { if (C == 1) STREAM(R8_C1, op); else if (C == 2) STREAM(R8_C2, op); else if (C == 4) STREAM(R8_C4, op); else STREAM(R8_C8, op); } while (0)

#define OP_ADD(n)     "v_add_u32 %" #n ", %" #n ", %8\n"
#define OP_LSHL(n)    "v_lshlrev_b32 %" #n ", 1, %" #n "\n"
#define OP_AND(n)     "v_and_b32 %" #n ", %" #n ", %8\n"
#define OP_CNDMASK(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define OP_MUL24(n)   "v_mul_i32_i24 %" #n ", %" #n ", %8\n"
#define OP_MULLO(n)   "v_mul_lo_u32 %" #n ", %" #n ", %8\n"
#define OP_SDWA(n)    "v_add_u32_sdwa %" #n ", %" #n ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
#define OP_BFE(n)     "v_bfe_u32 %" #n ", %" #n ", 1, 9\n"
#define OP_FFBH(n)    "v_ffbh_u32 %" #n ", %" #n "\n"
#define OP_PERM(n)    "v_perm_b32 %" #n ", %" #n ", %8, %8\n"
#define OP_MED3(n)    "v_med3_i32 %" #n ", %" #n ", %8, %8\n"
#define OP_LSHLADD(n) "v_lshl_add_u32 %" #n ", %" #n ", 1, %8\n"
#define OP_CMPSEL(n)  "v_cmp_lt_u32 vcc, %" #n ", %8\nv_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define OP_RFL(n)     "v_readfirstlane_b32 s20, %" #n "\nv_add_u32 %" #n ", s20, %8\n"
#define OP_RFLS(n)    "v_readfirstlane_b32 s20, %" #n "\ns_add_u32 s20, s20, %9\nv_add_u32 %" #n ", s20, %8\n"
#define OP_MOVX(n)    "v_mov_b32 %" #n ", %8\n"
#define OP_LDS(n)     "ds_read_b32 %" #n ", %" #n "\ns_waitcnt lgkmcnt(0)\n"
#define OP_LDS8(n)    "ds_read_u8 %" #n ", %" #n "\ns_waitcnt lgkmcnt(0)\n"
#define OP_LDSWR(n)   "ds_write_b32 %" #n ", %" #n "\nds_read_b32 %" #n ", %" #n "\ns_waitcnt lgkmcnt(0)\n"

template <int K, int C>
__global__ void probe(unsigned long long *out, int iters, unsigned seed, const unsigned *gbuf) {
    extern __shared__ unsigned lds[];
    unsigned r[8], x = seed | 1u, sx = seed & 3u;
    const unsigned lane = threadIdx.x & 63u;
    // LDS words hold their own byte address: a dependent read returns the address it was read from (a chase that stays put, any bank pattern we like)
    for (unsigned i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i * 4u;
    __syncthreads();
    for (int c = 0; c < 8; c++) r[c] = (K == K_LDS_CHASE || K == K_LDS_U8 || K == K_LDS_WR_RD || K == K_LDS_B64) ? (lane * 4u + 256u * c + (threadIdx.x >> 6) * 2048u) & 16383u : (K == K_GLB_CHASE) ? (lane * 4u + 256u * c) : seed + lane + c;
    if (K == K_LDS_U8) for (int c = 0; c < 8; c++) r[c] = 0;       // byte 0 of word 0 is 0: the chase reads address 0 for ever (all lanes one address: a broadcast)
    asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(lane), "v"(32u) : "vcc");
    unsigned long long t0 = 0, t1 = 0;
    for (int pass = 0; pass < 2; pass++) {          // pass 0 warms the instruction cache
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; it++) {
            if constexpr (K == K_ADD) BY_CHAINS(OP_ADD);
            else if constexpr (K == K_LSHL) BY_CHAINS(OP_LSHL);
            else if constexpr (K == K_AND) BY_CHAINS(OP_AND);
            else if constexpr (K == K_CNDMASK) BY_CHAINS(OP_CNDMASK);
            else if constexpr (K == K_MUL24) BY_CHAINS(OP_MUL24);
            else if constexpr (K == K_MULLO) BY_CHAINS(OP_MULLO);
            else if constexpr (K == K_SDWA) BY_CHAINS(OP_SDWA);
            else if constexpr (K == K_BFE) BY_CHAINS(OP_BFE);
            else if constexpr (K == K_FFBH) BY_CHAINS(OP_FFBH);
            else if constexpr (K == K_PERM) BY_CHAINS(OP_PERM);
            else if constexpr (K == K_MED3) BY_CHAINS(OP_MED3);
            else if constexpr (K == K_LSHLADD) BY_CHAINS(OP_LSHLADD);
            else if constexpr (K == K_CMPSEL) BY_CHAINS(OP_CMPSEL);
            else if constexpr (K == K_RFL) BY_CHAINS(OP_RFL);
            else if constexpr (K == K_RFL_SALU) BY_CHAINS(OP_RFLS);
            else if constexpr (K == K_MOVX) BY_CHAINS(OP_MOVX);
            else if constexpr (K == K_LDS_CHASE) BY_CHAINS(OP_LDS);
            else if constexpr (K == K_LDS_U8) BY_CHAINS(OP_LDS8);
            else if constexpr (K == K_LDS_WR_RD) BY_CHAINS(OP_LDSWR);
        }
        t1 = __builtin_readcyclecounter();
    }
    unsigned acc = 0;
    for (int c = 0; c < 8; c++) acc ^= r[c];
    if (acc == 0x12345u) out[4096] = acc;                 // keep the chains alive
    if (lane == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    (void)gbuf;
}
// scalar chains, global chase, 64-bit LDS chase: separate bodies (different operand classes)
template <int K, int C>
__global__ void probe_s(unsigned long long *out, int iters, unsigned seed, const unsigned *gbuf) {
    extern __shared__ unsigned lds[];
    const unsigned lane = threadIdx.x & 63u;
    for (unsigned i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i * 4u;
    __syncthreads();
    unsigned long long t0 = 0, t1 = 0;
    unsigned s[4] = { seed, seed + 1, seed + 2, seed + 3 }, sx = seed | 3u;
    unsigned v[4] = { lane * 4u, lane * 4u + 256u, lane * 4u + 512u, lane * 4u + 768u };
    unsigned long long w[4] = { lane * 8ull, lane * 8ull + 512, lane * 8ull + 1024, lane * 8ull + 1536 };
    for (int pass = 0; pass < 2; pass++) {
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; it++) {
            if constexpr (K == K_SALU) {
#define OP_S(n) "s_add_u32 %" #n ", %" #n ", %4\n"
                if (C == 1) asm volatile(R64(R8_C1, OP_S) : "+s"(s[0]), "+s"(s[1]), "+s"(s[2]), "+s"(s[3]) : "s"(sx) : "scc");
                else asm volatile(R64(R8_C4, OP_S) : "+s"(s[0]), "+s"(s[1]), "+s"(s[2]), "+s"(s[3]) : "s"(sx) : "scc");
            } else if constexpr (K == K_SALU_MUL) {
#define OP_SM(n) "s_mul_i32 %" #n ", %" #n ", %4\n"
                if (C == 1) asm volatile(R64(R8_C1, OP_SM) : "+s"(s[0]), "+s"(s[1]), "+s"(s[2]), "+s"(s[3]) : "s"(sx) : "scc");
                else asm volatile(R64(R8_C4, OP_SM) : "+s"(s[0]), "+s"(s[1]), "+s"(s[2]), "+s"(s[3]) : "s"(sx) : "scc");
            } else if constexpr (K == K_GLB_CHASE) {
#define OP_G(n) "global_load_dword %" #n ", %" #n ", %4\ns_waitcnt vmcnt(0)\n"
                if (C == 1) asm volatile(R8_C1(OP_G) R8_C1(OP_G) : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) : "s"(gbuf) : "memory");
                else asm volatile(R8_C4(OP_G) R8_C4(OP_G) : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) : "s"(gbuf) : "memory");
            }
        }
        t1 = __builtin_readcyclecounter();
    }
    if ((s[0] ^ s[1] ^ s[2] ^ s[3] ^ v[0] ^ v[1] ^ v[2] ^ v[3] ^ (unsigned)w[0] ^ (unsigned)w[1] ^ (unsigned)w[2] ^ (unsigned)w[3]) == 0x12345u) out[4096] = 1;
    if (lane == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

// ---- control-flow and synchronisation costs ----------------------------------------------------------------------------
// a tiny loop: one VALU instruction + counter + taken branch per iteration
__global__ void probe_branch(unsigned long long *out, int iters, unsigned seed) {
    unsigned r = seed + threadIdx.x, n = (unsigned)iters * 64u;
    const unsigned long long t0 = __builtin_readcyclecounter();
    asm volatile("1:\n v_add_u32 %0, %0, %0\n s_sub_u32 %1, %1, 1\n s_cmp_lg_u32 %1, 0\n s_cbranch_scc1 1b\n" : "+v"(r), "+s"(n) : : "scc");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (r == 0x12345u) out[4096] = r;
    if ((threadIdx.x & 63u) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}
// workgroup barriers back to back
__global__ void probe_barrier(unsigned long long *out, int iters) {
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters * 64; i++) __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63u) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}
// LDS flag ping-pong between wave 0 and wave `peer` of a workgroup: wave 0 stores n, the peer sees it and stores n back.  sleepq: s_sleep argument while polling (0: spin)
template <int SLEEP>
__global__ void probe_pingpong(unsigned long long *out, int iters, int peer) {
    __shared__ int fa, fb;
    if (threadIdx.x == 0) { fa = 0; fb = 0; }
    __syncthreads();
    const int w = threadIdx.x >> 6;
    const unsigned long long t0 = __builtin_readcyclecounter();
    const int n = iters * 16;
    if (w == 0) {
        for (int i = 1; i <= n; i++) {
            __hip_atomic_store(&fa, i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            while (__hip_atomic_load(&fb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != i) { if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP); }
        }
    } else if (w == peer) {
        for (int i = 1; i <= n; i++) {
            while (__hip_atomic_load(&fa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != i) { if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP); }
            __hip_atomic_store(&fb, i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63u) == 0) out[blockIdx.x * 16 + w] = t1 - t0;
}

struct Res { std::string name; int chains, threads, blocks, wps; double cyc_per_inst_wave, inst_per_cyc_simd, wall_ghz; };
static std::vector<Res> results;
static unsigned long long *d_out; static unsigned *d_g;

template <class F>
static void run(const char *name, int ninst_per_iter, int chains, int blocks, int threads, size_t lds, int iters, F launch) {
    CHK(hipMemset(d_out, 0, 8 * 4100));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    CHK(hipEventRecord(e0));
    launch(dim3(blocks), dim3(threads), lds);
    CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize());
    float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(blocks * 16);
    CHK(hipMemcpy(h.data(), d_out, 8 * h.size(), hipMemcpyDeviceToHost));
    std::vector<double> cyc;
    const int nw = threads / 64;
    for (int b = 0; b < blocks; b++) for (int w = 0; w < nw; w++) if (h[b * 16 + w]) cyc.push_back((double)h[b * 16 + w]);
    std::sort(cyc.begin(), cyc.end());
    const double med = cyc.empty() ? 0 : cyc[cyc.size() / 2];
    const double per = med / ((double)iters * ninst_per_iter);
    const int wps = (nw + 3) / 4;       // waves per SIMD when the block sits alone on its compute unit
    Res r; r.name = name; r.chains = chains; r.threads = threads; r.blocks = blocks; r.wps = wps; r.cyc_per_inst_wave = per; r.inst_per_cyc_simd = per > 0 ? (nw >= 4 ? (double)nw / 4 : 1.0) / per : 0;
    r.wall_ghz = ms > 0 ? med * 2 / (ms * 1e6) : 0;     // (two passes inside the kernel)
    results.push_back(r);
    printf("%-40s chains %d  %4d thr x %3d blk (%d wave/SIMD): %7.2f cyc/inst per wave   %6.3f inst/cyc/SIMD   [kernel %.3f ms]\n", name, chains, threads, blocks, wps, per, r.inst_per_cyc_simd, ms);
    fflush(stdout);
}

template <int K, int C> static void run_kind(int blocks, int threads, int iters) {
    const size_t lds = 100 * 1024;       // one block per compute unit
    CHK(hipFuncSetAttribute((const void *)probe<K, C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    run(kind_name[K], 64 * kind_len[K], C, blocks, threads, lds, iters, [&](dim3 g, dim3 b, size_t l) { hipLaunchKernelGGL((probe<K, C>), g, b, l, 0, d_out, iters, 12345u, d_g); });
}
template <int K, int C> static void run_kind_s(int blocks, int threads, int iters, int per_iter) {
    const size_t lds = 100 * 1024;
    CHK(hipFuncSetAttribute((const void *)probe_s<K, C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    run(kind_name[K], per_iter, C, blocks, threads, lds, iters, [&](dim3 g, dim3 b, size_t l) { hipLaunchKernelGGL((probe_s<K, C>), g, b, l, 0, d_out, iters, 12345u, d_g); });
}
template <int K> static void sweep(int iters) {
    // latency view: one wave per SIMD on 8 compute units (what a single frame's workgroups are), chains 1 / 2 / 4 / 8
    run_kind<K, 1>(8, 256, iters); run_kind<K, 2>(8, 256, iters); run_kind<K, 4>(8, 256, iters); run_kind<K, 8>(8, 256, iters);
    // throughput view: the whole device, 1 / 2 / 4 waves per SIMD, dependent and 8 independent chains
    run_kind<K, 1>(256, 512, iters); run_kind<K, 1>(256, 1024, iters);
    run_kind<K, 8>(256, 256, iters); run_kind<K, 8>(256, 512, iters); run_kind<K, 8>(256, 1024, iters);
}

int main(int argc, char **argv) {
    CHK(hipMalloc(&d_out, 8 * 4100));
    std::vector<unsigned> g(1 << 16);
    for (size_t i = 0; i < g.size(); i++) g[i] = (unsigned)(i * 4);
    CHK(hipMalloc(&d_g, 4 * g.size())); CHK(hipMemcpy(d_g, g.data(), 4 * g.size(), hipMemcpyHostToDevice));
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    printf("device %s, %d CUs, clockRate %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    const int it = 2000;
    sweep<K_ADD>(it); sweep<K_LSHL>(it); sweep<K_AND>(it); sweep<K_CNDMASK>(it); sweep<K_MUL24>(it); sweep<K_MULLO>(it); sweep<K_SDWA>(it);
    sweep<K_BFE>(it); sweep<K_FFBH>(it); sweep<K_PERM>(it); sweep<K_MED3>(it); sweep<K_LSHLADD>(it); sweep<K_CMPSEL>(it); sweep<K_MOVX>(it);
    run_kind<K_RFL, 1>(8, 256, it); run_kind<K_RFL, 4>(8, 256, it); run_kind<K_RFL, 1>(256, 1024, it);
    run_kind<K_RFL_SALU, 1>(8, 256, it); run_kind<K_RFL_SALU, 4>(8, 256, it);
    run_kind_s<K_SALU, 1>(8, 256, it, 64); run_kind_s<K_SALU, 4>(8, 256, it, 64); run_kind_s<K_SALU, 4>(256, 1024, it, 64);
    run_kind_s<K_SALU_MUL, 1>(8, 256, it, 64); run_kind_s<K_SALU_MUL, 4>(8, 256, it, 64);
    run_kind<K_LDS_CHASE, 1>(8, 256, 200); run_kind<K_LDS_CHASE, 4>(8, 256, 200); run_kind<K_LDS_CHASE, 1>(8, 64, 200); run_kind<K_LDS_CHASE, 1>(256, 1024, 200);
    run_kind<K_LDS_U8, 1>(8, 256, 200);
    run_kind<K_LDS_WR_RD, 1>(8, 256, 200); run_kind<K_LDS_WR_RD, 4>(8, 256, 200);
    run_kind_s<K_GLB_CHASE, 1>(8, 256, 200, 16); run_kind_s<K_GLB_CHASE, 4>(8, 256, 200, 16);
    run("loop: v_add + s_sub + s_cmp + taken branch", 64 * 4, 1, 8, 256, 0, it, [&](dim3 gd, dim3 b, size_t) { hipLaunchKernelGGL(probe_branch, gd, b, 0, 0, d_out, it, 5u); });
    for (int thr : { 256, 512, 768, 1024 })
        run("s_barrier back to back", 64, 1, 8, thr, 0, 200, [&](dim3 gd, dim3 b, size_t) { hipLaunchKernelGGL(probe_barrier, gd, b, 0, 0, d_out, 200); });
    for (int peer : { 1, 2, 3, 4 }) {
        run("LDS flag round trip, spin (peer wave below)", 16, peer, 8, 512, 0, 500, [&](dim3 gd, dim3 b, size_t) { hipLaunchKernelGGL((probe_pingpong<0>), gd, b, 0, 0, d_out, 500, peer); });
        run("LDS flag round trip, s_sleep 2 (peer below)", 16, peer, 8, 512, 0, 500, [&](dim3 gd, dim3 b, size_t) { hipLaunchKernelGGL((probe_pingpong<2>), gd, b, 0, 0, d_out, 500, peer); });
    }
    if (argc > 1) {
        FILE *f = fopen(argv[1], "w");
        if (f) {
            fprintf(f, "{\"device\": \"%s\", \"cus\": %d, \"results\": [\n", p.name, p.multiProcessorCount);
            for (size_t i = 0; i < results.size(); i++) {
                const Res &r = results[i];
                fprintf(f, "  {\"inst\": \"%s\", \"chains\": %d, \"threads\": %d, \"blocks\": %d, \"waves_per_simd\": %d, \"cycles_per_inst_one_wave\": %.3f, \"inst_per_cycle_per_simd\": %.4f}%s\n",
                        r.name.c_str(), r.chains, r.threads, r.blocks, r.wps, r.cyc_per_inst_wave, r.inst_per_cyc_simd, i + 1 < results.size() ? "," : "");
            }
            fprintf(f, "]}\n"); fclose(f);
        }
    }
    return 0;
}
