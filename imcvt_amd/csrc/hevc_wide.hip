// hevc_wide.hip — the encoder kernel once more, instantiated for WIDE launches only (512-thread workgroups, one per compute unit, hevc_core.h
// "Wide workgroups").  Same device source as hevc_hip.hip's hevc_encode_frames (hevc_frame.h kernel_main), different budget: a wide workgroup
// owns its compute unit at two wavefronts per SIMD, so each wavefront may use 256 registers instead of the 168 that four 192-thread workgroups
// per compute unit leave, and the file is built WITH the backend's loop-invariant code motion (imcvt_amd/build.py) — at 168 registers the
// hoisted values are spilled on the spot, at 256 they fit.  A wide workgroup is work-bound on its compute unit: spill
// stores and reloads are instructions it pays for.  Results are identical; IMCVT_HEVC_WIDE_KERNEL=0 launches hevc_encode_frames instead (A/B).
#include <hip/hip_runtime.h>
#include "hevc_frame.h"

__global__ __launch_bounds__(WG_THREADS_WIDE, 2) void hevc_encode_frames_wide(const Tables *gT, const ColdTables *gK, const FrameJob *jobs, const u8 *hdrs, int njobs,
                                                                              const Scratch *scr, int *counter, i32 *trace, int trace_cap, unsigned long long *prof,
                                                                              TeamMail *mail, PoolQ *pq, int team_size, int nteams, int nhelp, int post16, int post32, int lim16, int lim32, int prio, int quota, unsigned long long *fclk, int role, int block0, int npart) {
    KArgs A;
    A.gT = gT; A.gK = gK; A.jobs = jobs; A.hdrs = hdrs; A.njobs = njobs; A.scr = scr; A.counter = counter; A.trace = trace; A.trace_cap = trace_cap; A.prof = prof;
    A.mail = mail; A.pq = pq; A.team_size = team_size; A.nteams = nteams; A.nhelp = nhelp; A.post16 = post16; A.post32 = post32; A.lim16 = lim16; A.lim32 = lim32; A.prio = prio; A.quota = quota; A.fclk = fclk;
    A.role = role; A.npart = npart;
    kernel_main(A, (int)blockIdx.x + block0);
}

// host side: what hevc_hip.hip needs of this kernel (launch, dynamic LDS limit, occupancy, private segment size)
extern "C" int imcvt_wide_kernel_prepare(int *blocks_per_cu, int *scratch_bytes_per_lane) {
    if (hipFuncSetAttribute((const void *)hevc_encode_frames_wide, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WIDE_LDS_BYTES) != hipSuccess) { (void)hipGetLastError(); return -1; }
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, hevc_encode_frames_wide, WG_THREADS_WIDE, WIDE_LDS_BYTES) != hipSuccess || nb < 1) { (void)hipGetLastError(); return -1; }
    hipFuncAttributes fa;
    if (hipFuncGetAttributes(&fa, (const void *)hevc_encode_frames_wide) != hipSuccess) { (void)hipGetLastError(); return -1; }
    if (blocks_per_cu) *blocks_per_cu = nb;
    if (scratch_bytes_per_lane) *scratch_bytes_per_lane = (int)fa.localSizeBytes;
    return 0;
}
extern "C" void imcvt_wide_kernel_launch(int grid, void *stream, const Tables *gT, const ColdTables *gK, const FrameJob *jobs, const u8 *hdrs, int njobs,
                                         const Scratch *scr, int *counter, i32 *trace, int trace_cap, unsigned long long *prof,
                                         TeamMail *mail, PoolQ *pq, int team_size, int nteams, int nhelp, int post16, int post32, int lim16, int lim32, int prio, int quota, unsigned long long *fclk, int role, int block0, int npart) {
    hipLaunchKernelGGL(hevc_encode_frames_wide, dim3(grid), dim3(WG_THREADS_WIDE), WIDE_LDS_BYTES, (hipStream_t)stream, gT, gK, jobs, hdrs, njobs, scr, counter, trace, trace_cap, prof,
                       mail, pq, team_size, nteams, nhelp, post16, post32, lim16, lim32, prio, quota, fclk, role, block0, npart);
}
