// vcc_probe.hip — [developer measurement tool] when is v_cndmask_b32 slow?  (valu_sgpr_probe: 64 back-to-back VOP2 v_cndmask on one old vcc cost 18 cycles each, the VOP3 form 5.4)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
#define R8(op) op op op op op op op op
#define R64(op) R8(op) R8(op) R8(op) R8(op) R8(op) R8(op) R8(op) R8(op)
#define ADD "v_add_u32 %1, %1, %2\n"
#define CMP "v_cmp_lt_u32 vcc, %0, %2\n"
#define CND "v_cndmask_b32 %0, %0, %2, vcc\n"
#define CND2 "v_cndmask_b32 %1, %1, %2, vcc\n"
#define CNDE "v_cndmask_b32_e64 %0, %0, %2, vcc\n"
#define CNDE2 "v_cndmask_b32_e64 %1, %1, %2, vcc\n"
enum { K0, K1, K2, K3, K4, K5, K6, K7, K8, KN };
static const char *names[KN] = { "cmp, cnd (2)", "cmp, add x4, cnd (6)", "cmp, add x8, cnd (10)", "cmp, cnd, cnd2 (3)", "cmp, cnd, cnd2, cnd, cnd2 (5)", "cmp, cnd_e64, cnd2_e64, cnd_e64, cnd2_e64 (5)",
                                 "s_mov vcc, cnd (2)", "cmp, add, cnd, add, cnd2, add, cnd, add, cnd2 (9)", "cmp, s_nop 4, cnd, cnd2 (4)" };
static const int lens[KN] = { 2, 6, 10, 3, 5, 5, 2, 9, 4 };
template <int K> __global__ void k(unsigned long long *out, int iters, unsigned seed) {
    unsigned a = seed + threadIdx.x, b = seed * 3 + threadIdx.x, x = seed | 1u;
    unsigned long long t0 = 0, t1 = 0;
    for (int pass = 0; pass < 2; pass++) { t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; it++) {
            if constexpr (K == K0) asm volatile(R64(CMP CND) : "+v"(a), "+v"(b) : "v"(x) : "vcc");
            else if constexpr (K == K1) asm volatile(R64(CMP ADD ADD ADD ADD CND) : "+v"(a), "+v"(b) : "v"(x) : "vcc");
            else if constexpr (K == K2) asm volatile(R64(CMP ADD ADD ADD ADD ADD ADD ADD ADD CND) : "+v"(a), "+v"(b) : "v"(x) : "vcc");
            else if constexpr (K == K3) asm volatile(R64(CMP CND CND2) : "+v"(a), "+v"(b) : "v"(x) : "vcc");
            else if constexpr (K == K4) asm volatile(R64(CMP CND CND2 CND CND2) : "+v"(a), "+v"(b) : "v"(x) : "vcc");
            else if constexpr (K == K5) asm volatile(R64(CMP CNDE CNDE2 CNDE CNDE2) : "+v"(a), "+v"(b) : "v"(x) : "vcc");
            else if constexpr (K == K6) asm volatile(R64("s_mov_b64 vcc, 0x33\n" CND) : "+v"(a), "+v"(b) : "v"(x) : "vcc");
            else if constexpr (K == K7) asm volatile(R64(CMP ADD CND ADD CND2 ADD CND ADD CND2) : "+v"(a), "+v"(b) : "v"(x) : "vcc");
            else if constexpr (K == K8) asm volatile(R64(CMP "s_nop 4\n" CND CND2) : "+v"(a), "+v"(b) : "v"(x) : "vcc");
        }
        t1 = __builtin_readcyclecounter(); }
    if ((a ^ b) == 0x12345u) out[4096] = a;
    if ((threadIdx.x & 63u) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int K> static void run(unsigned long long *d) {
    hipMemset(d, 0, 8 * 4100);
    hipLaunchKernelGGL(k<K>, dim3(8), dim3(512), 0, 0, d, 500, 12345u); hipDeviceSynchronize();
    std::vector<unsigned long long> h(8 * 16); hipMemcpy(h.data(), d, 8 * h.size(), hipMemcpyDeviceToHost);
    std::vector<double> v; for (auto c : h) if (c) v.push_back((double)c); std::sort(v.begin(), v.end());
    printf("%-62s %7.2f cycles per group, %6.2f per instruction\n", names[K], v[v.size() / 2] / (500.0 * 64), v[v.size() / 2] / (500.0 * 64 * lens[K]));
}
int main() { unsigned long long *d; hipMalloc(&d, 8 * 4100); run<K0>(d); run<K1>(d); run<K2>(d); run<K3>(d); run<K4>(d); run<K5>(d); run<K6>(d); run<K7>(d); run<K8>(d); return 0; }
