// valu_sgpr_probe.hip — [developer measurement tool] follow-up to valu_rate_probe: what a VALU instruction costs a lone wavefront when one of
// its operands is a scalar register or a lane mask (vcc / an SGPR pair) that was written LONG ago ("stale") or just before ("fresh").
// build: hipcc --offload-arch=gfx950 -O2 tools/valu_sgpr_probe.hip -o /tmp/vsp ; run: /tmp/vsp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define R8(op) op op op op op op op op
#define R64(op) R8(op) R8(op) R8(op) R8(op) R8(op) R8(op) R8(op) R8(op)
enum { K_ADD_VV, K_ADD_SV, K_ADD_SV_INL, K_ADD_LIT, K_CND_VCC, K_CND_SPAIR, K_CND_E64_VCC, K_CMP_ONLY, K_CMP_SPAIR_ONLY, K_CMP_CND, K_CMP_x_CND, K_CMPS_CND, K_MOV_SV, K_MIN_SV, K_SALU_VALU, K_EXECMASK, K_READLANE, K_N };
static const char *names[K_N] = { "v_add_u32 v,v,v", "v_add_u32 v,s(stale),v", "v_add_u32 v,17(inline),v", "v_add_u32 v,0x12345(literal),v", "v_cndmask_b32 v,v,v,vcc (stale vcc, VOP2)",
    "v_cndmask_b32_e64 v,v,v,s[a:b] (stale pair)", "v_cndmask_b32_e64 v,v,0,vcc (stale vcc, VOP3)", "v_cmp_lt_u32 vcc,v,v only", "v_cmp_lt_u32_e64 s[a:b],v,v only",
    "v_cmp vcc + v_cndmask vcc (pair, fresh)", "v_cmp vcc + 2 v_add + v_cndmask vcc (4 instr)", "v_cmp_e64 s[a:b] + v_cndmask_e64 s[a:b] (pair, fresh)", "v_mov_b32 v,s(stale)", "v_min_u32 v,s(stale),v",
    "s_add_u32 s + v_add_u32 v,s,v (pair: SALU result into VALU)", "s_and_saveexec + v_add + s_or exec (3 instr)", "v_readlane_b32 s,v,3 + v_add v,s,v (pair)" };
static const int lens[K_N] = { 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 4, 2, 1, 1, 2, 3, 2 };
template <int K>
__global__ void probe(unsigned long long *out, int iters, unsigned seed) {
    unsigned r = seed + threadIdx.x, x = seed | 1u, y = seed * 3u;
    unsigned s0 = seed & 7u;
    unsigned long long m = (threadIdx.x & 1) ? 0x5555555555555555ull : 0x3333333333333333ull, sv = 0;
    asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(threadIdx.x & 63u), "v"(32u) : "vcc");
    asm volatile("s_mov_b64 %0, 0x33333333" : "=s"(m));
    unsigned long long t0 = 0, t1 = 0;
    for (int pass = 0; pass < 2; pass++) {
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; it++) {
            if constexpr (K == K_ADD_VV) asm volatile(R64("v_add_u32 %0, %0, %1\n") : "+v"(r) : "v"(x));
            else if constexpr (K == K_ADD_SV) asm volatile(R64("v_add_u32 %0, %1, %0\n") : "+v"(r) : "s"(s0));
            else if constexpr (K == K_ADD_SV_INL) asm volatile(R64("v_add_u32 %0, 17, %0\n") : "+v"(r));
            else if constexpr (K == K_ADD_LIT) asm volatile(R64("v_add_u32 %0, 0x12345, %0\n") : "+v"(r));
            else if constexpr (K == K_CND_VCC) asm volatile(R64("v_cndmask_b32 %0, %0, %1, vcc\n") : "+v"(r) : "v"(x));
            else if constexpr (K == K_CND_SPAIR) asm volatile(R64("v_cndmask_b32_e64 %0, %0, %1, %2\n") : "+v"(r) : "v"(x), "s"(m));
            else if constexpr (K == K_CND_E64_VCC) asm volatile(R64("v_cndmask_b32_e64 %0, %0, 0, vcc\n") : "+v"(r));
            else if constexpr (K == K_CMP_ONLY) asm volatile(R64("v_cmp_lt_u32 vcc, %0, %1\n") : : "v"(r), "v"(x) : "vcc");
            else if constexpr (K == K_CMP_SPAIR_ONLY) asm volatile(R64("v_cmp_lt_u32_e64 %0, %1, %2\n") : "=s"(sv) : "v"(r), "v"(x));
            else if constexpr (K == K_CMP_CND) asm volatile(R64("v_cmp_lt_u32 vcc, %0, %1\nv_cndmask_b32 %0, %0, %1, vcc\n") : "+v"(r) : "v"(x) : "vcc");
            else if constexpr (K == K_CMP_x_CND) asm volatile(R64("v_cmp_lt_u32 vcc, %0, %1\nv_add_u32 %2, %2, %1\nv_add_u32 %2, %2, %1\nv_cndmask_b32 %0, %0, %1, vcc\n") : "+v"(r) : "v"(x), "v"(y) : "vcc");
            else if constexpr (K == K_CMPS_CND) asm volatile(R64("v_cmp_lt_u32_e64 %2, %0, %1\nv_cndmask_b32_e64 %0, %0, %1, %2\n") : "+v"(r) : "v"(x), "s"(sv));
            else if constexpr (K == K_MOV_SV) asm volatile(R64("v_mov_b32 %0, %1\n") : "+v"(r) : "s"(s0));
            else if constexpr (K == K_MIN_SV) asm volatile(R64("v_min_u32 %0, %1, %0\n") : "+v"(r) : "s"(s0));
            else if constexpr (K == K_SALU_VALU) asm volatile(R64("s_add_u32 %1, %1, 1\nv_add_u32 %0, %1, %0\n") : "+v"(r), "+s"(s0) : : "scc");
            else if constexpr (K == K_EXECMASK) asm volatile(R64("s_and_saveexec_b64 %1, vcc\nv_add_u32 %0, %0, %0\ns_or_b64 exec, exec, %1\n") : "+v"(r), "=s"(sv) : : "scc");
            else if constexpr (K == K_READLANE) asm volatile(R64("v_readlane_b32 %1, %0, 3\nv_add_u32 %0, %1, %0\n") : "+v"(r), "+s"(s0));
        }
        t1 = __builtin_readcyclecounter();
    }
    if ((r ^ (unsigned)sv ^ s0) == 0x12345u) out[4096] = r;
    if ((threadIdx.x & 63u) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}
static unsigned long long *d_out;
template <int K> static void run(int blocks, int threads, int iters) {
    CHK(hipMemset(d_out, 0, 8 * 4100));
    hipLaunchKernelGGL(probe<K>, dim3(blocks), dim3(threads), 0, 0, d_out, iters, 12345u);
    CHK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(blocks * 16);
    CHK(hipMemcpy(h.data(), d_out, 8 * h.size(), hipMemcpyDeviceToHost));
    std::vector<double> c;
    for (auto v : h) if (v) c.push_back((double)v);
    std::sort(c.begin(), c.end());
    printf("%-62s %4d thr x %3d blk: %7.2f cyc/inst per wave\n", names[K], threads, blocks, c[c.size() / 2] / ((double)iters * 64 * lens[K]));
    fflush(stdout);
}
template <int K> static void both() { run<K>(8, 256, 2000); run<K>(8, 512, 2000); }
int main() {
    CHK(hipMalloc(&d_out, 8 * 4100));
    both<K_ADD_VV>(); both<K_ADD_SV>(); both<K_ADD_SV_INL>(); both<K_ADD_LIT>(); both<K_CND_VCC>(); both<K_CND_SPAIR>(); both<K_CND_E64_VCC>(); both<K_CMP_ONLY>(); both<K_CMP_SPAIR_ONLY>();
    both<K_CMP_CND>(); both<K_CMP_x_CND>(); both<K_CMPS_CND>(); both<K_MOV_SV>(); both<K_MIN_SV>(); both<K_SALU_VALU>(); both<K_EXECMASK>(); both<K_READLANE>();
    return 0;
}
