// mfma_probe.hip — groundwork for moving the N >= 16 transforms onto the matrix cores: verifies, on the
// device, the operand / result register layouts of the int8 MFMA instructions of gfx950 that an exact integer DCT can use
// (i8 x i8 -> i32, limb-split operands).  Prints PASS/FAIL per instruction and the layout formulas it checked.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_probe.hip -o gpurun_out/mfma_probe && gpurun_out/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// D = A(32 x K) * B(K x 32), A row-major i8 [32][K], B row-major i8 [K][32]; one wave.
// Assumed layouts (checked against a CPU product):
//   32x32x16 (v_mfma_i32_32x32x16_i8): lane l holds A[l%32][8*(l/32) .. +7] and B[8*(l/32) .. +7][l%32], 8 bytes each (one i64)
//   32x32x32 (v_mfma_i32_32x32x32_i8): lane l holds A[l%32][16*(l/32) .. +15] and B[16*(l/32) .. +15][l%32], 16 bytes each (v4i32)
//   result  : lane l, register r (0..15) holds D[8*(r/4) + 4*(l/32) + r%4][l%32]
__global__ void k32x32x16(const int8_t *A, const int8_t *B, int *D) {
    const int l = threadIdx.x, i = l & 31, h = l >> 5;
    long a = 0, b = 0;
    for (int t = 0; t < 8; t++) { a |= (long)(uint8_t)A[i * 16 + 8 * h + t] << (8 * t); b |= (long)(uint8_t)B[(8 * h + t) * 32 + i] << (8 * t); }
    v16i c = {0};
    c = __builtin_amdgcn_mfma_i32_32x32x16_i8(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; r++) D[(8 * (r / 4) + 4 * h + r % 4) * 32 + i] = c[r];
}
__global__ void k32x32x32(const int8_t *A, const int8_t *B, int *D) {
    const int l = threadIdx.x, i = l & 31, h = l >> 5;
    v4i a = {0}, b = {0};
    for (int t = 0; t < 16; t++) {
        a[t / 4] |= (int)(uint8_t)A[i * 32 + 16 * h + t] << (8 * (t % 4));
        b[t / 4] |= (int)(uint8_t)B[(16 * h + t) * 32 + i] << (8 * (t % 4));
    }
    v16i c = {0};
    c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; r++) D[(8 * (r / 4) + 4 * h + r % 4) * 32 + i] = c[r];
}
//   16x16x64 (v_mfma_i32_16x16x64_i8): lane l holds A[l%16][16*(l/16) .. +15], B[16*(l/16) .. +15][l%16]; result lane l, reg r: D[4*(l/16) + r][l%16]
__global__ void k16x16x64(const int8_t *A, const int8_t *B, int *D) {
    const int l = threadIdx.x, i = l & 15, h = l >> 4;
    v4i a = {0}, b = {0};
    for (int t = 0; t < 16; t++) {
        a[t / 4] |= (int)(uint8_t)A[i * 64 + 16 * h + t] << (8 * (t % 4));
        b[t / 4] |= (int)(uint8_t)B[(16 * h + t) * 16 + i] << (8 * (t % 4));
    }
    v4i c = {0};
    c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) D[(4 * h + r) * 16 + i] = c[r];
}
//   16x16x32 (v_mfma_i32_16x16x32_i8): lane l holds A[l%16][8*(l/16) .. +7], B[8*(l/16) .. +7][l%16] (one i64 each); result as 16x16x64
__global__ void k16x16x32(const int8_t *A, const int8_t *B, int *D) {
    const int l = threadIdx.x, i = l & 15, h = l >> 4;
    long a = 0, b = 0;
    for (int t = 0; t < 8; t++) { a |= (long)(uint8_t)A[i * 32 + 8 * h + t] << (8 * t); b |= (long)(uint8_t)B[(8 * h + t) * 16 + i] << (8 * t); }
    v4i c = {0};
    c = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) D[(4 * h + r) * 16 + i] = c[r];
}
static int check(const char *name, int M, int N, int K, void (*launch)(const int8_t *, const int8_t *, int *)) {
    int8_t *hA = (int8_t *)malloc(M * K), *hB = (int8_t *)malloc(K * N); int *hD = (int *)malloc(M * N * 4), *ref = (int *)malloc(M * N * 4);
    unsigned s = 12345; for (int i = 0; i < M * K; i++) { s = s * 1664525u + 1013904223u; hA[i] = (int8_t)(s >> 24); }
    for (int i = 0; i < K * N; i++) { s = s * 1664525u + 1013904223u; hB[i] = (int8_t)(s >> 24); }
    for (int i = 0; i < M; i++) for (int j = 0; j < N; j++) { int a = 0; for (int k = 0; k < K; k++) a += (int)hA[i * K + k] * (int)hB[k * N + j]; ref[i * N + j] = a; }
    int8_t *dA, *dB; int *dD;
    hipMalloc(&dA, M * K); hipMalloc(&dB, K * N); hipMalloc(&dD, M * N * 4);
    hipMemcpy(dA, hA, M * K, hipMemcpyHostToDevice); hipMemcpy(dB, hB, K * N, hipMemcpyHostToDevice); hipMemset(dD, 0xFF, M * N * 4);
    launch(dA, dB, dD); hipDeviceSynchronize();
    hipMemcpy(hD, dD, M * N * 4, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < M * N; i++) bad += hD[i] != ref[i];
    printf("%-28s %s (%d of %d elements differ)\n", name, bad ? "FAIL" : "PASS", bad, M * N);
    return bad;
}
int main() {
    int bad = 0;
    bad += check("v_mfma_i32_32x32x16_i8", 32, 32, 16, [](const int8_t *a, const int8_t *b, int *d) { hipLaunchKernelGGL(k32x32x16, 1, 64, 0, 0, a, b, d); });
    bad += check("v_mfma_i32_32x32x32_i8", 32, 32, 32, [](const int8_t *a, const int8_t *b, int *d) { hipLaunchKernelGGL(k32x32x32, 1, 64, 0, 0, a, b, d); });
    bad += check("v_mfma_i32_16x16x64_i8", 16, 16, 64, [](const int8_t *a, const int8_t *b, int *d) { hipLaunchKernelGGL(k16x16x64, 1, 64, 0, 0, a, b, d); });
    bad += check("v_mfma_i32_16x16x32_i8", 16, 16, 32, [](const int8_t *a, const int8_t *b, int *d) { hipLaunchKernelGGL(k16x16x32, 1, 64, 0, 0, a, b, d); });
    return bad != 0;
}
