// simd_map_probe.hip — [developer measurement tool] which SIMD each wavefront of a 512- / 768-thread workgroup lands on (HW_ID.simd_id),
// for workgroups alone on their compute unit (100 KB of LDS each).  build: hipcc --offload-arch=gfx950 -O2 tools/simd_map_probe.hip -o /tmp/simd_map_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned *out) {
    extern __shared__ unsigned lds[];
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_REG_HW_ID: wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13]
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = hw;
    lds[threadIdx.x] = hw; __syncthreads();
}
int main() {
    unsigned *d; hipMalloc(&d, 4 * 16 * 64); 
    for (int thr : {256, 512, 768}) {
        hipMemset(d, 0xFF, 4 * 16 * 64);
        hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
        hipLaunchKernelGGL(k, dim3(16), dim3(thr), 100 * 1024, 0, d);
        hipDeviceSynchronize();
        unsigned h[16 * 64]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("%d threads: simd id of waves 0.. per block\n", thr);
        for (int b = 0; b < 16; b++) { printf("  blk %2d cu %2u:", b, (h[b * 16] >> 8) & 15); for (int w = 0; w < thr / 64; w++) printf(" %u", (h[b * 16 + w] >> 4) & 3); printf("   (wave ids"); for (int w = 0; w < thr / 64; w++) printf(" %u", h[b * 16 + w] & 15); printf(")\n"); }
    }
    return 0;
}
