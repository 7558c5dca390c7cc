// hevc_hip.hip — gfx950 kernel entry and the C-ABI host shim of libimcvt_hevc.so (include/imcvt_hevc.h).
//
// One persistent workgroup per frame slot: workgroups pull frame indices from an atomic counter, so any
// batch size runs on a grid sized to the device (2 workgroups per CU by default).  The frame's CTUs are a
// strict serial chain (the RD rate is the live CABAC position), so parallelism is candidates x pixels
// inside a workgroup and frames across workgroups / GPUs.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "hevc_frame.h"
#include "hevc_tables.h"
#include "../../include/imcvt_hevc.h"

#ifndef KERNEL_MIN_WAVES
#define KERNEL_MIN_WAVES 3          // waves per SIMD the register budget is cut for: 3 = 168 registers (four 192-thread or three 256-thread workgroups per compute unit)
#endif
#ifndef KERNEL_MAX_THREADS
#define KERNEL_MAX_THREADS WG_THREADS_WIDE
#endif
__global__ __launch_bounds__(KERNEL_MAX_THREADS, KERNEL_MIN_WAVES) void hevc_encode_frames(const Tables *gT, const ColdTables *gK, const FrameJob *jobs, const u8 *hdrs, int njobs,
                                                                 const Scratch *scr, int *counter, i32 *trace, int trace_cap, unsigned long long *prof,
                                                                 TeamMail *mail, PoolQ *pq, int team_size, int nteams, int nhelp, int post16, int post32, int lim16, int lim32, int prio, int quota, unsigned long long *fclk, int role, int block0, int npart) {
    KArgs A;
    A.gT = gT; A.gK = gK; A.jobs = jobs; A.hdrs = hdrs; A.njobs = njobs; A.scr = scr; A.counter = counter; A.trace = trace; A.trace_cap = trace_cap; A.prof = prof;
    A.mail = mail; A.pq = pq; A.team_size = team_size; A.nteams = nteams; A.nhelp = nhelp; A.post16 = post16; A.post32 = post32; A.lim16 = lim16; A.lim32 = lim32; A.prio = prio; A.quota = quota; A.fclk = fclk;
    A.role = role; A.npart = npart;
    kernel_main(A, (int)blockIdx.x + block0);
}
// the same kernel instantiated for wide launches (hevc_wide.hip: 256 registers per wavefront, built with loop-invariant code motion).  Weak: a library
// linked from this file alone (the A/B and test variants, tools/gpu_variant.sh) runs its wide launches on the instantiation above.
extern "C" __attribute__((weak)) int imcvt_wide_kernel_prepare(int *blocks_per_cu, int *scratch_bytes_per_lane);
extern "C" __attribute__((weak)) void imcvt_wide_kernel_launch(int grid, void *stream, const Tables *gT, const ColdTables *gK, const FrameJob *jobs, const u8 *hdrs, int njobs,
                                         const Scratch *scr, int *counter, i32 *trace, int trace_cap, unsigned long long *prof,
                                         TeamMail *mail, PoolQ *pq, int team_size, int nteams, int nhelp, int post16, int post32, int lim16, int lim32, int prio, int quota, unsigned long long *fclk, int role, int block0, int npart);

// ---------------------------------------------------------------------------------------------------
struct imcvt_hevc_ctx {
    int device = 0, max_wg = 0, cus = 0;
    Tables *d_tables = nullptr; ColdTables *d_cold = nullptr;
    Scratch *d_scratch = nullptr;
    void *d_pool = nullptr;            // backing store of all per-workgroup scratch
    int *d_counter = nullptr;
    TeamMail *d_mail = nullptr; int mail_cap = 0;          // one mailbox set per main workgroup
    PoolQ *d_pq = nullptr;                                 // request queues of the running launch
    FrameJob *d_jobs = nullptr; u8 *d_hdrs = nullptr; int jobs_cap = 0;
    FrameJob *h_jobs = nullptr; u8 *h_hdrs = nullptr;      // pinned staging
    hipEvent_t ev0 = nullptr, ev1 = nullptr; bool timed = false;
    int *d_trace = nullptr; int trace_cap = 0;
    unsigned long long *d_fclk = nullptr;   // optional per-frame clocks (imcvt_hevc_set_frame_clock)
    unsigned long long *d_prof = nullptr;   // [3 roles][NWAVES][PF_N] cycle totals (non-zero only in -DIMCVT_PROF builds)
    int force_team = 0;                     // 0: choose per launch; 1: no helpers; 2 / 3: one / two helper workgroups per main workgroup
    int force_mains = 0, force_help = 0;    // > 0: exactly this launch shape (debug / tuning)
    int post16 = -1, post32 = -1;           // pool tuning: per mille of the 16x16 / 32x32 CUs offered to the helpers (IMCVT_POOL_POST16 / _POST32; < 0: from the launch shape)
    int lim16 = -1, lim32 = -1, prio = -1;  // pool tuning (IMCVT_POOL_LIM16 / _LIM32 / _PRIO; < 0: defaults from the launch shape)
    int last_mains = 0, last_help = 0;
    int pipe = -1, pipe_wg = 0, last_pipe = 0;   // pipe wave (256-thread workgroups): < 0 whenever the launch fits pipe_wg workgroups (3 per CU), 0 never, 1 as -1 (forced on where it fits)
    int occ_wg = 0, occ_pipe = 0;           // workgroups per compute unit the HIP occupancy API reports for 192- / 256-thread workgroups of this kernel
    int census_wg = 0, census_pipe = 0;     // workgroups of a max_wg / pipe_wg launch that were resident at once when the context was created (0: not measured)
    int wide = -1, wide_wg = 0, occ_wide = 0, last_wide = 0;   // wide workgroups (512 threads: pipe wave + four partner wavefronts, one workgroup per compute unit): < 0 whenever a pipe-wave launch fits wide_wg workgroups, 0 never, 1 as -1
    int pending_err = 0;                    // an earlier launch that nobody asked about ended badly (watchdog): reported by the next imcvt_hevc_last_status
    int wide_kernel = 0, wide_scratch = 0;  // wide launches run hevc_encode_frames_wide (hevc_wide.hip; IMCVT_HEVC_WIDE_KERNEL=0: the common instantiation), its private segment per lane
    int partners = -1, last_part = 0;       // partner workgroups (wide pools: the 2Nx2N sets of a main workgroup's 8x8 CUs on a second compute unit, hevc_frame.h): < 0 (default) / 1 wherever they fit, 0 never
    int split = -1, split_hpc = 0, last_split = 0;      // a pool spread over two cooperating launches (launch_split): < 0 where planned, 0 never, 1 as < 0; helper workgroups per compute unit of the helpers' set (0: default)
    hipStream_t st_split[2] = { nullptr, nullptr }; int split_cus[2] = { 0, 0 };      // streams bound to two disjoint sets of compute units, and how many each holds
    hipEvent_t ev_split[3] = { nullptr, nullptr, nullptr };
    u32 *prog = nullptr;                    // progress records of the next launches' frames, two words each (imcvt_hevc_set_progress), or null
};

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "imcvt_hevc: %s failed: %s\n", #x, hipGetErrorString(e_)); return IMCVT_ERR_HIP; } } while (0)

static bool have_device() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        fprintf(stderr, "imcvt_hevc: no HIP device visible — this library has no CPU fallback\n");
        return false;
    }
    return true;
}

extern "C" const char *imcvt_hevc_version(void) { return "imcvt_hevc gfx950 r6 (wg=192; 256 with a pipe wave when the launch leaves room; 512 - pipe wave + four partner wavefronts, trial coders and PU steps split over wavefronts - when every workgroup gets a compute unit; a frame per workgroup, or main workgroups + a pool of helper workgroups when the batch leaves room; N = 16 / 32 transforms on the matrix cores; trial coders leave the leads of their bytes, the winner's become bytes by a carry look-ahead over ballots)"; }
extern "C" int imcvt_hevc_padded(int v) { return ((v < 8192 ? v : 8192) + 31) / 32 * 32; }
extern "C" long long imcvt_hevc_stream_bound(int h, int w) { return 2LL * (w + 32) * (h + 32) + 65536; }

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// Balance of a launch with helpers.  Per CTU a main workgroup spends about 1.35 units on the 8x8 CUs (which it always walks itself)
// where the candidate sets of the four 16x16 CUs and of the 32x32 CU cost about 1 unit each (cycle counters of round 2 and 3,
// DESIGN.md §4; 1.5 until TU 0 of the four-TU shape was shared — ten interleaved launch pairs of the bench shape at the end of
// round 3, profiles/r03y5_split_ab.log: median 5.27 s with 56 % of the 16x16 CUs offered against 5.32 s with 63 %).  With h
// helpers per main workgroup the shares x (16x16) and y (32x32) that are handed over should make both
// sides finish together: 1.35 + (1 - x) + (1 - y) = (x + y) / h.  32x32 requests go first (their answers are needed last).
// Returns the share of kind 0 (16x16) / 1 (32x32) per mille.
static int pool_split(int nmains, int nhelp, int kind) {
    if (nmains < 1 || nhelp < 1) return 0;
    const double h = (double)nhelp / nmains, total = 3.35 * h / (1.0 + h);    // x + y
    double y = total < 1.0 ? total : 1.0, x = total - y;
    if (x > 1.0) x = 1.0;
    const int v = (int)((kind == 0 ? x : y) * 1000.0 + 0.5);
    return v < 0 ? 0 : v > 1000 ? 1000 : v;
}
// An offered CU is still kept while this many requests of its kind wait unclaimed in the main workgroup's queue shard (every
// helper is busy and the queue is long): a safety net under the split above.
static int pool_limit(int nmains, int nhelp, int kind) {
    if (nhelp >= 2 * nmains) return 1 << 20;
    const int per_shard = (nhelp + POOL_SHARDS - 1) / POOL_SHARDS;
    const int v = kind == 0 ? per_shard / 2 : per_shard;
    return v > 1 ? v : 1;
}
// pipe: the workgroups carry a fourth wavefront (hevc_frame.h nxn_pipe) and its LDS slice; three such workgroups fit a CU
// pipe 2: wide workgroups (512 threads: pipe wave + four partner wavefronts and their record queues), one per compute unit
// role / block0: 0 / 0 for a launch that holds the whole pool; a pool spread over two launches (launch_split) passes 1 / 0 for the main workgroups' and 2 / nmains for the helpers'
// (block0: where this launch's workgroups start in the scratch table and among the queue shards)
static void launch(imcvt_hevc_ctx *c, int grid, hipStream_t stream, int njobs, int team_size, int nmains, int nhelp, int pipe = 0, int role = 0, int block0 = 0, int split16 = -1, int split32 = -1, int npart = 0) {
    const int p16 = c->post16 >= 0 ? c->post16 : split16 >= 0 ? split16 : pool_split(nmains, nhelp, 0), p32 = c->post32 >= 0 ? c->post32 : split32 >= 0 ? split32 : pool_split(nmains, nhelp, 1);
    const int l16 = c->lim16 >= 0 ? c->lim16 : pool_limit(nmains, nhelp, 0), l32 = c->lim32 >= 0 ? c->lim32 : pool_limit(nmains, nhelp, 1);
    const int prio = c->prio >= 0 ? c->prio : (nhelp >= 2 * nmains ? 2 : 0);
    static const int by_arrival = getenv("IMCVT_POOL_ROLES_BY_ARRIVAL") ? atoi(getenv("IMCVT_POOL_ROLES_BY_ARRIVAL")) : 0;      // (A/B: main workgroups = the first two arrivals of every compute unit, the rule of rounds 3 - 5)
    const int quota = (by_arrival ? -1 : 1) * ((nmains + c->cus - 1) / (c->cus > 0 ? c->cus : 1));
    if (pipe >= 2 && c->wide_kernel) {
        imcvt_wide_kernel_launch(grid, (void *)stream, c->d_tables, c->d_cold, (const FrameJob *)c->d_jobs, (const u8 *)c->d_hdrs, njobs, (const Scratch *)c->d_scratch, c->d_counter, c->d_trace, c->trace_cap, c->d_prof,
                                 c->d_mail, c->d_pq, team_size, nmains, nhelp, p16, p32, l16, l32, prio, quota, c->d_fclk, role, block0, npart);
        return;
    }
    hipLaunchKernelGGL(hevc_encode_frames, dim3(grid), dim3(pipe >= 2 ? WG_THREADS_WIDE : pipe ? WG_THREADS_PIPE : WG_THREADS), pipe >= 2 ? WIDE_LDS_BYTES : pipe ? PIPE_LDS_BYTES : 0, stream, c->d_tables, c->d_cold, (const FrameJob *)c->d_jobs, (const u8 *)c->d_hdrs, njobs,
                       (const Scratch *)c->d_scratch, c->d_counter, c->d_trace, c->trace_cap, c->d_prof, c->d_mail, c->d_pq, team_size, nmains, nhelp,
                       p16, p32, l16, l32, prio, quota, c->d_fclk, role, block0, npart);
}

// residency census: `grid` workgroups that count themselves, wait ~1 ms and record how many had started by then (kernel_main, team_size < 0)
static int census(imcvt_hevc_ctx *c, int grid, int pipe) {
    if (grid < 1 || hipMemset(c->d_counter, 0, 8 * sizeof(int)) != hipSuccess) return 0;
    launch(c, grid, 0, 0, -1, 0, 0, pipe);
    int v[2] = { 0, 0 };
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(v, c->d_counter, sizeof v, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return v[1];
}
// ... taken twice when the first count is far below the grid (under three quarters: another process or stream held compute units at that
// moment): the plans of the context's whole life should not rest on one bad millisecond; the better of the two counts
static int census_checked(imcvt_hevc_ctx *c, int grid, int pipe) {
    int v = census(c, grid, pipe);
    if (v > 0 && 4 * v < 3 * grid) { const int v2 = census(c, grid, pipe); if (v2 > v) v = v2; }
    return v;
}
extern "C" imcvt_hevc_ctx *imcvt_hevc_create(int max_workgroups) {
    if (!have_device()) return nullptr;
    imcvt_hevc_ctx *c = new imcvt_hevc_ctx();
    hipDeviceProp_t prop;
    if (hipGetDevice(&c->device) != hipSuccess || hipGetDeviceProperties(&prop, c->device) != hipSuccess) { delete c; return nullptr; }
    {   // Full residency (4 workgroups x 3 waves per CU) needs private-segment memory for every resident wave; ROCr
        // throttles wave launch when that exceeds its scratch limit, so raise the limit to what this kernel needs.
        hipFuncAttributes fa; size_t cur = 0, mx = 0;
        if (hipFuncGetAttributes(&fa, (const void *)hevc_encode_frames) == hipSuccess
            && hipDeviceGetLimit(&cur, hipExtLimitScratchCurrent) == hipSuccess && hipDeviceGetLimit(&mx, hipExtLimitScratchMax) == hipSuccess) {
            const size_t need = (size_t)fa.localSizeBytes * 64 * 32 * prop.multiProcessorCount + (64u << 20);   // sized for 32 waves/CU like ROCr does
            if (getenv("IMCVT_HEVC_VERBOSE")) fprintf(stderr, "imcvt_hevc: scratch limit cur=%zu max=%zu need=%zu (%zu B/lane)\n", cur, mx, need, (size_t)fa.localSizeBytes);
            if (cur < need) (void)hipDeviceSetLimit(hipExtLimitScratchCurrent, need < mx ? need : mx);
        } else (void)hipGetLastError();
    }
    c->cus = prop.multiProcessorCount;
    // How many workgroups of a launch can be resident at once is what every pool shape is planned against (main workgroups wait for
    // helpers, helpers poll): it comes from the occupancy API for the actual block sizes and dynamic LDS — 4 per compute unit for
    // 192 threads (40.9 KB of LDS, 168 registers), 3 for 256 threads with the pipe wave's slice — and is checked against a census
    // launch below (the API has been seen one block high for some register counts).
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&c->occ_wg, hevc_encode_frames, WG_THREADS, 0) != hipSuccess || c->occ_wg < 1) { (void)hipGetLastError(); c->occ_wg = 4; }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&c->occ_pipe, hevc_encode_frames, WG_THREADS_PIPE, PIPE_LDS_BYTES) != hipSuccess || c->occ_pipe < 1) { (void)hipGetLastError(); c->occ_pipe = 3; }
    // wide workgroups: static + dynamic LDS exceed the 64 KB a kernel gets without asking
    if (hipFuncSetAttribute((const void *)hevc_encode_frames, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WIDE_LDS_BYTES) != hipSuccess
        || hipOccupancyMaxActiveBlocksPerMultiprocessor(&c->occ_wide, hevc_encode_frames, WG_THREADS_WIDE, WIDE_LDS_BYTES) != hipSuccess || c->occ_wide < 1) { (void)hipGetLastError(); c->occ_wide = 0; }
    {   // wide launches have an instantiation of their own (hevc_wide.hip); IMCVT_HEVC_WIDE_KERNEL=0 keeps them on the common one (A/B)
        const char *e = getenv("IMCVT_HEVC_WIDE_KERNEL");
        int nb = 0, sb = 0;
        if (c->occ_wide > 0 && !(e && atoi(e) == 0) && imcvt_wide_kernel_prepare && imcvt_wide_kernel_launch && imcvt_wide_kernel_prepare(&nb, &sb) == 0) {
            c->wide_kernel = 1; c->wide_scratch = sb;
            if (nb < c->occ_wide) c->occ_wide = nb;
            size_t cur = 0, mx = 0;      // its private segment may be the larger one: the scratch ring is sized for whichever kernel needs more
            if (hipDeviceGetLimit(&cur, hipExtLimitScratchCurrent) == hipSuccess && hipDeviceGetLimit(&mx, hipExtLimitScratchMax) == hipSuccess) {
                const size_t need = (size_t)sb * 64 * 32 * prop.multiProcessorCount + (64u << 20);
                if (cur < need) (void)hipDeviceSetLimit(hipExtLimitScratchCurrent, need < mx ? need : mx);
            } else (void)hipGetLastError();
            if (getenv("IMCVT_HEVC_VERBOSE")) fprintf(stderr, "imcvt_hevc: wide kernel: %d per compute unit, %d B/lane private segment\n", nb, sb);
        }
    }
    if (const char *e = getenv("IMCVT_HEVC_PARTNERS")) c->partners = atoi(e);
    if (const char *e = getenv("IMCVT_HEVC_SPLIT")) c->split = atoi(e);
    if (const char *e = getenv("IMCVT_HEVC_SPLIT_HPC")) c->split_hpc = atoi(e);
    c->wide_wg = max_workgroups > 0 ? 0 : c->occ_wide * prop.multiProcessorCount;      // (a context with an explicit workgroup budget plans without them)
    if (const char *e = getenv("IMCVT_HEVC_WIDE")) c->wide = atoi(e);
    c->max_wg = max_workgroups > 0 ? max_workgroups : c->occ_wg * prop.multiProcessorCount;
    if (const char *e = getenv("IMCVT_HEVC_TEAM")) imcvt_hevc_set_team(c, atoi(e));      // clamped to 0..3 like the API call
    if (const char *e = getenv("IMCVT_POOL_POST16")) c->post16 = atoi(e);
    if (const char *e = getenv("IMCVT_POOL_POST32")) c->post32 = atoi(e);
    if (const char *e = getenv("IMCVT_POOL_LIM16")) c->lim16 = atoi(e);
    if (const char *e = getenv("IMCVT_POOL_LIM32")) c->lim32 = atoi(e);
    if (const char *e = getenv("IMCVT_POOL_PRIO")) c->prio = atoi(e);
    c->pipe_wg = max_workgroups > 0 ? (int)((long long)c->max_wg * c->occ_pipe / c->occ_wg) : c->occ_pipe * prop.multiProcessorCount;     // registers (4 x 168 per workgroup) and LDS (40.9 + 6.9 KB) admit 3 per CU
    if (const char *e = getenv("IMCVT_HEVC_PIPE")) c->pipe = atoi(e);
    c->mail_cap = c->max_wg / 2 + 8;
    Tables *T = new Tables(); ColdTables *K = new ColdTables();
    imcvt::build_tables(*T, *K);
    const size_t per_wg = scratch_bytes_per_wg();
    bool ok = hipMalloc(&c->d_tables, sizeof(Tables)) == hipSuccess
           && hipMemcpy(c->d_tables, T, sizeof(Tables), hipMemcpyHostToDevice) == hipSuccess
           && hipMalloc(&c->d_cold, sizeof(ColdTables)) == hipSuccess
           && hipMemcpy(c->d_cold, K, sizeof(ColdTables), hipMemcpyHostToDevice) == hipSuccess
           && hipMalloc(&c->d_pool, per_wg * c->max_wg) == hipSuccess
           && hipMalloc(&c->d_scratch, sizeof(Scratch) * c->max_wg) == hipSuccess
           && hipMalloc(&c->d_counter, 8 * sizeof(int)) == hipSuccess
           && hipMalloc(&c->d_mail, sizeof(TeamMail) * c->mail_cap) == hipSuccess
           && hipMalloc(&c->d_pq, sizeof(PoolQ)) == hipSuccess
           && hipMalloc(&c->d_prof, sizeof(unsigned long long) * (3 * NWAVES * PF_N + REG_N)) == hipSuccess
           && hipMemset(c->d_prof, 0, sizeof(unsigned long long) * (3 * NWAVES * PF_N + REG_N)) == hipSuccess
           && hipEventCreate(&c->ev0) == hipSuccess && hipEventCreate(&c->ev1) == hipSuccess;
    delete T; delete K;
    if (ok) {
        std::vector<Scratch> hs(c->max_wg);
        const bool rev = getenv("IMCVT_SCRATCH_REV") != nullptr;      // (probe: does a workgroup's speed follow its scratch slot or its place in the dispatch order?  profiles/r06za_scratch_rev.log)
        for (int i = 0; i < c->max_wg; i++) scratch_carve(hs[i], (u8 *)c->d_pool + per_wg * (rev ? c->max_wg - 1 - i : i));
        ok = hipMemcpy(c->d_scratch, hs.data(), sizeof(Scratch) * c->max_wg, hipMemcpyHostToDevice) == hipSuccess;
    }
    if (!ok) { fprintf(stderr, "imcvt_hevc: context allocation failed\n"); imcvt_hevc_destroy(c); return nullptr; }
    // ROCr sizes the private-segment ring from the dispatches it has seen: the first full-grid launches of a process run
    // with fewer resident waves (measured: 1.6x the steady kernel time, twice).  Three empty full-grid launches (no frames:
    // every workgroup leaves at once) bring the ring to size before real work arrives.
    auto prewarm = [&]() {
        bool good = true;
        if (c->wide_kernel && c->wide_wg > 0) {      // the wide instantiation's private segment first: the LAST launches the runtime sees before real work are full 192-thread ones
            good = hipMemset(c->d_counter, 0, 8 * sizeof(int)) == hipSuccess;      // (a 256-thread census launch in that place left a later full launch with ~940 of 1000 workgroups resident, profiles/r04c_census_probe.log)
            if (good) launch(c, c->wide_wg, 0, 0, 1, 0, 0, 2);
            good = good && hipDeviceSynchronize() == hipSuccess;
        }
        for (int i = 0; i < 3 && good; i++) {
            good = hipMemset(c->d_counter, 0, 8 * sizeof(int)) == hipSuccess;
            if (good) launch(c, c->max_wg, 0, 0, 1, 0, 0);
            good = good && hipDeviceSynchronize() == hipSuccess;
        }
        return good;
    };
    if (!getenv("IMCVT_HEVC_NO_PREWARM")) {
        if (!prewarm()) { fprintf(stderr, "imcvt_hevc: pre-warm launch failed\n"); imcvt_hevc_destroy(c); return nullptr; }
        // census: a launch of max_wg (pipe_wg) workgroups that only count themselves — what is resident at once is what the plans may use
        if (max_workgroups <= 0 && !getenv("IMCVT_HEVC_NO_CENSUS")) {
            if (c->wide_wg > 0) { const int cw = census_checked(c, c->wide_wg, 2);      // (whatever IMCVT_HEVC_WIDE says now: imcvt_hevc_set_wide may turn them on later)
                if (cw > 0 && cw < c->wide_wg) { fprintf(stderr, "imcvt_hevc: %d of %d wide workgroups resident - planning with %d\n", cw, c->wide_wg, cw); c->wide_wg = cw; } }
            c->census_pipe = census_checked(c, c->pipe_wg, 1); c->census_wg = census_checked(c, c->max_wg, 0);
            if (c->census_wg > 0 && c->census_wg < c->max_wg) { fprintf(stderr, "imcvt_hevc: %d of %d workgroups resident (occupancy API: %d per compute unit) - planning with %d\n", c->census_wg, c->max_wg, c->occ_wg, c->census_wg); c->max_wg = c->census_wg; }
            if (c->census_pipe > 0 && c->census_pipe < c->pipe_wg) { fprintf(stderr, "imcvt_hevc: %d of %d pipe-wave workgroups resident (occupancy API: %d per compute unit) - planning with %d\n", c->census_pipe, c->pipe_wg, c->occ_pipe, c->census_pipe); c->pipe_wg = c->census_pipe; }
            if (c->pipe_wg > c->max_wg) c->pipe_wg = c->max_wg;
            // (the 256-thread census launch left the SECOND full 192-thread launch after it with ~940 of 1000 workgroups resident,
            // profiles/r04c_census_probe.log: the last launches the runtime sees before real work are full 192-thread ones again)
            if (!prewarm()) { fprintf(stderr, "imcvt_hevc: pre-warm launch failed\n"); imcvt_hevc_destroy(c); return nullptr; }
            // ... and three SHORT real launches of the largest pool (max_wg / 2 frames of 64 x 64 zeros: 16 ms each): the first three full launches of a process do not get every
            // workgroup of a pool that uses every slot resident — 1021 - 1023 of 1024, the others start when the first leave, 4.6 / 5.3 / 5.0 s for the bench batch — and neither the
            // empty launches nor the census above change that; real ones do (profiles/r06zw_first_launches.log: 4.59 - 4.60 s with 1024 resident from the first long launch on).
            if (!getenv("IMCVT_HEVC_NO_WARM_POOL") && c->max_wg >= 64) {
                const int n = c->max_wg / 2, hw = 64;
                const size_t per = align256((size_t)imcvt_hevc_stream_bound(hw, hw)) + 2 * align256((size_t)hw * hw) + 256;
                u8 *buf = nullptr;
                if (hipMalloc(&buf, per * (size_t)n) == hipSuccess) {
                    bool good = hipMemset(buf, 0, per * (size_t)n) == hipSuccess;
                    std::vector<imcvt_hevc_frame> fr((size_t)n);
                    for (int i = 0; i < n; i++) {
                        u8 *b0 = buf + per * (size_t)i;
                        fr[(size_t)i].d_img = b0; fr[(size_t)i].d_rcon = b0 + align256((size_t)hw * hw); fr[(size_t)i].d_out = b0 + 2 * align256((size_t)hw * hw);
                        fr[(size_t)i].d_len = (int *)(b0 + per - 256); fr[(size_t)i].h = hw; fr[(size_t)i].w = hw; fr[(size_t)i].qpd6 = 0;
                    }
                    for (int i = 0; i < 3 && good; i++) good = imcvt_hevc_encode_device(c, n, fr.data(), nullptr) == 0 && hipDeviceSynchronize() == hipSuccess;
                    (void)hipFree(buf);
                    if (!good) { (void)hipGetLastError(); fprintf(stderr, "imcvt_hevc: warm-up pool launch failed\n"); imcvt_hevc_destroy(c); return nullptr; }
                }
            }
        }
    }
    return c;
}

extern "C" void imcvt_hevc_destroy(imcvt_hevc_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->timed && c->ev1) (void)hipEventSynchronize(c->ev1);
    hipFree(c->d_tables); hipFree(c->d_cold); hipFree(c->d_pool); hipFree(c->d_scratch); hipFree(c->d_counter); hipFree(c->d_prof); hipFree(c->d_mail); hipFree(c->d_pq);
    hipFree(c->d_jobs); hipFree(c->d_hdrs);
    if (c->h_jobs) hipHostFree(c->h_jobs);
    if (c->h_hdrs) hipHostFree(c->h_hdrs);
    if (c->ev0) hipEventDestroy(c->ev0);
    if (c->ev1) hipEventDestroy(c->ev1);
    for (int i = 0; i < 2; i++) if (c->st_split[i]) hipStreamDestroy(c->st_split[i]);
    for (int i = 0; i < 3; i++) if (c->ev_split[i]) hipEventDestroy(c->ev_split[i]);
    delete c;
}

extern "C" void imcvt_hevc_set_frame_clock(imcvt_hevc_ctx *c, unsigned long long *d_buf) { if (c) c->d_fclk = d_buf; }
extern "C" void imcvt_hevc_set_progress(imcvt_hevc_ctx *c, unsigned int *words) { if (c) c->prog = (u32 *)words; }
extern "C" void imcvt_hevc_set_trace(imcvt_hevc_ctx *c, int *d_trace, int cap) { if (c) { c->d_trace = d_trace; c->trace_cap = cap; } }
extern "C" void imcvt_hevc_set_pipe(imcvt_hevc_ctx *c, int mode) { if (c) c->pipe = mode; }
extern "C" int imcvt_hevc_last_pipe(imcvt_hevc_ctx *c) { return c ? c->last_pipe : IMCVT_ERR_ARG; }
extern "C" void imcvt_hevc_set_wide(imcvt_hevc_ctx *c, int mode) { if (c) c->wide = mode; }
extern "C" int imcvt_hevc_last_wide(imcvt_hevc_ctx *c) { return c ? c->last_wide : IMCVT_ERR_ARG; }
// pure: a launch of `grid` workgroups that runs with the pipe wave runs wide workgroups when every one of them gets a compute unit of
// its own with a sixteenth of the `wide_wg` resident 512-thread workgroups to spare (a forced shape may fill the last one)
extern "C" int imcvt_hevc_plan_wide(int use_pipe, int grid, int wide_wg, int forced_shape) {
    if (!use_pipe || grid < 1 || wide_wg < 1) return 0;
    return grid <= (forced_shape ? wide_wg : wide_wg - wide_wg / 16) ? 1 : 0;
}
// ... and a pool that does not fit that way still runs wide when its main workgroups take at most half of the compute units: the helpers are
// cut to the rest (*nhelp; one per main workgroup at least) — 128 frames: 128 + 128 wide workgroups (2.92 s) against 128 + 256 of 256 threads
// (3.28 s), profiles/r05s_wide_128f_shapes.log.  Every workgroup then needs a compute unit of its own, none to spare: nothing in a pool depends
// on a workgroup that is not running, so a missing compute unit costs time, not results.  Returns 1 if the launch runs wide.
extern "C" int imcvt_hevc_plan_wide_pool(int use_pipe, int mode, int forced_shape, int wide_wg, const int *nmains, int *nhelp) {
    if (!nmains || !nhelp) return 0;
    if (imcvt_hevc_plan_wide(use_pipe, *nmains + *nhelp, wide_wg, forced_shape)) return 1;
    if (!use_pipe || forced_shape || mode < 2 || *nmains < 1 || 2 * *nmains > wide_wg) return 0;
    if (*nhelp > wide_wg - *nmains) *nhelp = wide_wg - *nmains;      // (cut, never raised: a plan that already fits the compute units keeps its helper count)
    return 1;
}
// A pool spread over TWO cooperating launches (round 6).  One launch has one workgroup size, so a pool whose main workgroups are wide (a compute unit
// each) has wide helpers too — one per compute unit, and a pool of 81 .. 128 main workgroups is left with fewer than two helpers each: 128 frames, one
// GPU's share of BASELINE configs[3] at N = 4, ran 128 + 128 wide workgroups, every main workgroup keeping a third of its 16x16 CUs for itself
// (profiles/r05s_wide_128f_shapes.log).  The pool's protocol lives in global memory and does not care which launch a workgroup belongs to
// (hevc_frame.h: roles, mail, queues), so such a pool runs as two launches on two streams bound to disjoint sets of compute units
// (hipExtStreamCreateWithCUMask): `nmains` wide main workgroups on as many compute units, and 192-thread helper workgroups, `hpc` per compute unit,
// on the others.  Nothing depends on a workgroup that is not running (a main workgroup posts requests only once a helper has reported in), so the
// order in which the two launches become resident costs time at most.
// pure: returns 1 and the helper count if a pool of nmains main workgroups (mode = what imcvt_hevc_plan returned) should run split on a device of
// `cus` compute units that holds wide_wg wide and occ_wg 192-thread workgroups per compute unit; hpc: helper workgroups per compute unit (0: default)
extern "C" int imcvt_hevc_plan_split(int mode, int nmains, int cus, int wide_wg, int occ_wg, int hpc, int *nhelp) {
    if (mode != 2 || nmains < 1 || cus < 2 || wide_wg < cus - cus / 16 || occ_wg < 1) return 0;      // (nearly every compute unit must be able to hold a wide workgroup)
    if (2 * nmains > cus) return 0;                                       // the main workgroups get one half of the compute units, the helpers the other
    // one launch of wide workgroups gives a main workgroup (wide_wg - nmains) / nmains helpers; down to 1.4 it is as fast as the split (96 frames: 2.32 s
    // against 2.35 s; 128 frames, one helper each: 2.76 s against 2.40 s, profiles/r06a_ab.log)
    if (5 * (wide_wg - nmains) >= 7 * nmains) return 0;
    if (hpc < 1) hpc = 3;
    if (hpc > occ_wg) hpc = occ_wg;
    int h = (cus - cus / 2) * hpc;
    if (h > 4 * nmains) h = 4 * nmains;                                   // (a main workgroup has two requests out at most; the rest would only poll)
    if (h < nmains) return 0;
    if (nhelp) *nhelp = h;
    return 1;
}
// streams bound to the two halves of the compute units, created once per context (a stream whose mask was changed after another had been destroyed ran on the old
// mask: 112 main workgroups on the 96 compute units of the launch before took two rounds, profiles/r06a_ab.log)
static int split_streams(imcvt_hevc_ctx *c) {
    if (c->st_split[0]) return 0;
    const int words = (c->cus + 31) / 32, cus_a = c->cus / 2;
    std::vector<uint32_t> ma((size_t)words, 0u), mb((size_t)words, 0u);
    // (the driver deals the mask's bits out over XCDs, then shader engines, then arrays: a run of consecutive bits is spread evenly over the device)
    for (int i = 0; i < c->cus; i++) (i < cus_a ? ma : mb)[(size_t)(i / 32)] |= 1u << (i % 32);
    HIPCHK(hipExtStreamCreateWithCUMask(&c->st_split[0], (uint32_t)words, ma.data()));
    HIPCHK(hipExtStreamCreateWithCUMask(&c->st_split[1], (uint32_t)words, mb.data()));
    for (int i = 0; i < 3; i++) if (!c->ev_split[i]) HIPCHK(hipEventCreateWithFlags(&c->ev_split[i], hipEventDisableTiming));
    c->split_cus[0] = cus_a; c->split_cus[1] = c->cus - cus_a;
    return 0;
}
// Partner workgroups (round 6): in a wide pool every main workgroup gets a second compute unit for the two 2Nx2N candidate sets of its 8x8 CUs (hevc_frame.h
// "8x8 CUs with a partner workgroup") when the launch has room: one partner per main workgroup beside the planned helpers, or — a pool that just misses
// that — with the helpers cut to what is left, never below 1.5 per main workgroup.  pure; returns the number of partner workgroups (0 or nmains).
extern "C" int imcvt_hevc_plan_partners(int nmains, int *nhelp, int wide_wg, int forced_shape) {
    if (nmains < 1 || !nhelp || wide_wg < 1) return 0;
    const int cap = forced_shape ? wide_wg : wide_wg - wide_wg / 16;
    if (2 * nmains + *nhelp <= cap) return nmains;
    if (!forced_shape && 2 * nmains + (3 * nmains + 1) / 2 <= cap) { *nhelp = cap - 2 * nmains; return nmains; }
    return 0;
}
extern "C" void imcvt_hevc_set_partners(imcvt_hevc_ctx *c, int mode) { if (c) c->partners = mode; }
extern "C" int imcvt_hevc_last_partners(imcvt_hevc_ctx *c) { return c ? c->last_part : IMCVT_ERR_ARG; }
extern "C" void imcvt_hevc_set_split(imcvt_hevc_ctx *c, int mode, int helpers_per_cu) { if (c) { c->split = mode; c->split_hpc = helpers_per_cu; } }
extern "C" int imcvt_hevc_last_split(imcvt_hevc_ctx *c) { return c ? c->last_split : IMCVT_ERR_ARG; }
extern "C" void imcvt_hevc_set_team(imcvt_hevc_ctx *c, int team_size) { if (c) c->force_team = team_size < 0 ? 0 : team_size > 3 ? 3 : team_size; }
extern "C" int imcvt_hevc_last_team(imcvt_hevc_ctx *c, int *nteams) {
    if (!c) return IMCVT_ERR_ARG;
    if (nteams) *nteams = c->last_help > 0 ? c->last_mains : 0;
    return c->last_help <= 0 ? 1 : c->last_help >= 2 * c->last_mains ? 3 : 2;
}
extern "C" int imcvt_hevc_last_shape(imcvt_hevc_ctx *c, int *nmains, int *nhelp) {
    if (!c) return IMCVT_ERR_ARG;
    if (nmains) *nmains = c->last_mains;
    if (nhelp) *nhelp = c->last_help;
    return c->last_help > 0 ? 2 : 1;
}

// Launch shape.  A frame's CTUs are a serial chain; a workgroup that encodes its frame alone keeps 3 wavefronts busy for ~8.7 s
// (1080p, qpd6 0).  With helpers it hands the 70 unsplit candidates of every 16x16 / 32x32 CU to a pool of helper workgroups and
// walks only the 8x8 CUs itself (~4.1 s); the helper work of a frame is about as long as the main workgroup's own, so a pool as
// large as the mains keeps up with them, and more than two helpers per main cannot be used (a main workgroup has at most one
// request of each kind outstanding).  Every workgroup of the launch must be resident (helpers poll, mains wait for answers).
//   n <= max_wg / 2       n mains, min(2 n, max_wg - n) helpers: one round
//   n <= 5 max_wg / 8     max_wg / 2 mains and as many helpers; the mains pull the remaining frames as they finish
//   beyond                a frame per workgroup, max_wg of them, no helpers (the device is full either way and the hand-offs cost)
// pure: the launch shape for n frames on a device that holds max_wg workgroups (force_team 0: choose; 1: no helpers; 2 / 3: one / two
// helpers per main workgroup).  Returns 1 (frames per workgroup; *nmains workgroups) or 2 (pool; *nmains + *nhelp workgroups).
extern "C" int imcvt_hevc_plan(int n, int max_wg, int force_team, int *nmains_out, int *nhelp_out) {
    int d0 = 0, d1 = 0; int *nmains = nmains_out ? nmains_out : &d0, *nhelp = nhelp_out ? nhelp_out : &d1;
    *nmains = 0; *nhelp = 0;
    if (n < 1 || max_wg < 1) return 1;
    force_team = force_team < 0 ? 0 : force_team > 3 ? 3 : force_team;
    const int mail_cap = max_wg / 2 + 8;
    *nmains = n < max_wg ? n : max_wg;
    if (force_team == 1 || max_wg < 2) return 1;
    int m, h;
    if (force_team >= 2) {                                   // fixed ratio, as many mains as fit
        const int per = force_team;                          // workgroups per main
        m = n < max_wg / per ? n : max_wg / per;
        if (m < 1) return 1;
        h = (per - 1) * m;
    } else {
        if ((long long)n * 8 > (long long)max_wg * 5) return 1;
        m = n < max_wg / 2 ? n : max_wg / 2;
        if (m < 1) return 1;
        // (Rounds 3 - 5 and most of round 6 kept a sixteenth of the workgroup slots free: beyond 15/16 one launch in ten ran long, whatever rule chose the main workgroups
        // (profiles/r06u_pool_fill.log ... r06zj_fuller_pool_new_rule.log).  With the empty launches of the re-warm in front of every full launch (imcvt_hevc_encode_device) a pool of
        // max_wg workgroups is resident at once and ran 4.57 - 4.62 s in 52 launches of 52 against 4.80 - 4.82 s for 512 + 448: every slot is used.)
        const int room = max_wg - m;
        h = 2 * m < room ? 2 * m : room;
        if (h < 1) return 1;
    }
    if (m > mail_cap) m = mail_cap;
    if (2 * m > POOL_SHARDS * POOL_QCAP) m = POOL_SHARDS * POOL_QCAP / 2;
    *nmains = m; *nhelp = h;
    return 2;
}
// Pipe wave (256-thread workgroups, three per CU instead of four), as a pure function of the shape imcvt_hevc_plan chose: launches
// that leave a quarter of the workgroup slots free are latency-bound (every frame waits for the serial chain of its 8x8 CUs), so
// their workgroups get the fourth wavefront that takes the NxN trial off that chain; fuller launches keep four 192-thread
// workgroups per CU.  The same sixteenth of the slots stays free as in imcvt_hevc_plan; a pool that just misses the limit gives up
// helpers for it as long as 1.5 per main workgroup remain (*nhelp is reduced); a forced shape may fill the last slot.
// mode: what imcvt_hevc_plan returned.  Returns 1 if the launch runs with the pipe wave.
static int plan_pipe_wg(int mode, int max_wg, int pipe_wg, int forced_shape, int *nmains, int *nhelp) {      // pipe_wg: resident 256-thread workgroups (occupancy API x compute units)
    if (!nmains || !nhelp || max_wg < 4 || pipe_wg < 3) return 0;
    const int pipe_cap = pipe_wg - pipe_wg / 16;
    if (mode > 1 && !forced_shape && *nmains + *nhelp > pipe_cap && *nmains + (3 * *nmains + 1) / 2 <= pipe_cap) *nhelp = pipe_cap - *nmains;
    return *nmains + *nhelp <= (forced_shape ? pipe_wg : pipe_cap) ? 1 : 0;
}
extern "C" int imcvt_hevc_plan_pipe(int mode, int max_wg, int forced_shape, int *nmains, int *nhelp) { return plan_pipe_wg(mode, max_wg, max_wg / 4 * 3, forced_shape, nmains, nhelp); }
// *forced: the shape is imcvt_hevc_set_shape's.  A forced shape the context cannot hold (workgroups, mailboxes, queue capacity) is an
// argument error — returns 0 — before anything is written anywhere.
static int pick_shape(const imcvt_hevc_ctx *c, int n, int *nmains, int *nhelp, int *forced) {
    *forced = 0;
    if (c->force_mains > 0 && c->force_help > 0) {
        const int m = c->force_mains < n ? c->force_mains : n;
        if (m + c->force_help > c->max_wg || m > c->mail_cap || 2 * m > POOL_SHARDS * POOL_QCAP) { *nmains = m; *nhelp = c->force_help; return 0; }
        *nmains = m; *nhelp = c->force_help; *forced = 1;
        return 2;
    }
    return imcvt_hevc_plan(n, c->max_wg, c->force_team, nmains, nhelp);
}
extern "C" void imcvt_hevc_set_pool_tuning(imcvt_hevc_ctx *c, int lim16, int lim32, int prio) { if (c) { c->lim16 = lim16; c->lim32 = lim32; c->prio = prio; } }
extern "C" void imcvt_hevc_set_pool_split(imcvt_hevc_ctx *c, int post16, int post32) { if (c) { c->post16 = post16; c->post32 = post32; } }
extern "C" void imcvt_hevc_set_shape(imcvt_hevc_ctx *c, int nmains, int nhelp) {
    if (!c) return;
    const int ok = nmains > 0 && nhelp > 0;               // both or neither: anything else returns to the automatic choice
    c->force_mains = ok ? nmains : 0; c->force_help = ok ? nhelp : 0;
}

extern "C" int imcvt_hevc_encode_device(imcvt_hevc_ctx *c, int n, const imcvt_hevc_frame *frames, void *stream_) {
    if (!c || n < 0 || (n > 0 && !frames)) return IMCVT_ERR_ARG;
    if (n == 0) return 0;
    HIPCHK(hipSetDevice(c->device));
    hipStream_t stream = (hipStream_t)stream_;
    // One launch in flight per context: job table, frame counter, mailboxes and per-workgroup scratch belong to the launch
    // that is running, whatever stream it was put on.
    if (c->timed) {
        HIPCHK(hipEventSynchronize(c->ev1));
        if (c->last_help > 0 && !c->pending_err) {       // the launch this one replaces: keep its verdict for whoever asks next (the queues are about to be zeroed)
            u32 ab = 0;
            if (hipMemcpy(&ab, &c->d_pq->abort, sizeof ab, hipMemcpyDeviceToHost) == hipSuccess && ab != 0u) {
                c->pending_err = IMCVT_ERR_WATCHDOG;
                fprintf(stderr, "imcvt_hevc: the previous launch of this context (%d + %d workgroups) was abandoned by the device watchdog - its results are invalid\n", c->last_mains, c->last_help);
            }
        }
    }
    int nmains = 0, nhelp = 0, forced = 0;
    const int mode = pick_shape(c, n, &nmains, &nhelp, &forced);
    int use_split = 0, split_help = 0;
    if (mode == 2 && !forced && c->force_team == 0 && c->pipe != 0 && c->wide != 0 && c->split != 0)
        use_split = imcvt_hevc_plan_split(mode, nmains, c->cus, c->wide_wg, c->occ_wg, c->split_hpc, &split_help) && nmains + split_help <= c->max_wg;
    int use_pipe = (mode > 0 && c->pipe != 0) ? plan_pipe_wg(mode, c->max_wg, c->pipe_wg, forced, &nmains, &nhelp) : 0;
    int use_wide = 0;
    if (use_split) { nhelp = split_help; use_pipe = 1; }
    if (mode > 0 && c->pipe != 0 && c->wide != 0) {      // (a pool too large for 256-thread workgroups with its planned helpers may still fit wide ones with fewer)
        int h2 = nhelp;
        use_wide = use_split ? 1 : imcvt_hevc_plan_wide_pool(1, mode, forced, c->wide_wg, &nmains, &h2);
        if (use_wide && !use_split) { nhelp = h2; use_pipe = 1; }
    }
    if (use_split && split_streams(c) != 0) { (void)hipGetLastError(); return IMCVT_ERR_HIP; }
    int npart = 0;
    if (use_wide && !use_split && mode == 2 && c->partners != 0) npart = imcvt_hevc_plan_partners(nmains, &nhelp, c->wide_wg, forced);
    const int grid = nmains + nhelp + npart;
    if (mode <= 0 || grid < 1 || grid > c->max_wg || (mode > 1 && (nmains > c->mail_cap || 2 * nmains > POOL_SHARDS * POOL_QCAP))) {
        fprintf(stderr, "imcvt_hevc: launch shape %d + %d exceeds the context (%d workgroups, %d mailboxes)\n", nmains, nhelp, c->max_wg, c->mail_cap);
        return IMCVT_ERR_ARG;
    }
    if (n > c->jobs_cap) {
        hipFree(c->d_jobs); hipFree(c->d_hdrs);
        if (c->h_jobs) hipHostFree(c->h_jobs);
        if (c->h_hdrs) hipHostFree(c->h_hdrs);
        c->d_jobs = nullptr; c->d_hdrs = nullptr; c->h_jobs = nullptr; c->h_hdrs = nullptr; c->jobs_cap = 0;
        HIPCHK(hipMalloc(&c->d_jobs, sizeof(FrameJob) * n));
        HIPCHK(hipMalloc(&c->d_hdrs, (size_t)HDR_MAX * n));
        HIPCHK(hipHostMalloc(&c->h_jobs, sizeof(FrameJob) * n));
        HIPCHK(hipHostMalloc(&c->h_hdrs, (size_t)HDR_MAX * n));
        c->jobs_cap = n;
    }
    for (int i = 0; i < n; i++) {
        const imcvt_hevc_frame &f = frames[i];
        if (f.qpd6 < 0 || f.qpd6 > 4 || f.h < 1 || f.w < 1 || !f.d_img || !f.d_out || !f.d_rcon || !f.d_len) return IMCVT_ERR_ARG;
        FrameJob &j = c->h_jobs[i];
        j.img = f.d_img; j.out = f.d_out; j.rcon = f.d_rcon; j.out_len = f.d_len;
        j.h = f.h; j.w = f.w; j.hp = imcvt_hevc_padded(f.h); j.wp = imcvt_hevc_padded(f.w); j.q = f.qpd6;
        j.prog = c->prog ? c->prog + 2 * (size_t)i : nullptr;
        j.hdr_len = imcvt::build_headers(c->h_hdrs + (size_t)HDR_MAX * i, f.qpd6, j.hp, j.wp);
    }
    {   // A launch of another workgroup size leaves the device in a state in which every full launch of 192-thread workgroups after it — of this context or any other of the
        // process, on any stream — runs 6 - 7 % longer (the bench batch 5.11 - 5.17 s instead of 4.81 s, for as long as the process lives: profiles/r06zm_slow_process2.log; all
        // 960 workgroups resident and started within 40 us either way; what the state is has not been found).  The launches a context is created with — empty ones, every workgroup
        // leaves at once: one of wide workgroups, three full ones of 192 threads — bring the fast state back (profiles/r06zn_slow_process3.log); so they are repeated, on the
        // launch's own stream, whenever a full 192-thread launch follows a launch of another kind on this device.  (IMCVT_HEVC_NO_REWARM=1: A/B.)
        static std::atomic<int> last_kind[64];      // per device: 0 / 1 full-grid 192-thread launches last (a context's creation ends with them), 2 pipe-wave, 3 wide or split
        const int kind = (use_wide || use_split) ? 3 : use_pipe ? 2 : 1, dv = c->device & 63;
        // ... and in front of EVERY such launch they do more: a pool that uses every workgroup slot — 512 + 512 — then lands with all 1024 workgroups resident within 40 us and ran
        // 4.57 - 4.62 s in 52 launches of 52, where without them one launch in ten had workgroups that started only when others left or three main workgroups on a compute unit and took
        // 5.0 - 5.6 s (profiles/r06zu_rewarm_always.log, r06zv_rewarm_always_1024.log).  So: always (< 0.3 ms), and the plan uses every slot (imcvt_hevc_plan).
        if (kind == 1 && 2 * grid > c->max_wg && (last_kind[dv].load() > 1 || !getenv("IMCVT_HEVC_REWARM_ON_CHANGE_ONLY")) && !getenv("IMCVT_HEVC_NO_REWARM")) {
            if (c->wide_kernel && c->wide_wg > 0) { HIPCHK(hipMemsetAsync(c->d_counter, 0, 8 * sizeof(int), stream)); launch(c, c->wide_wg, stream, 0, 1, 0, 0, 2); }
            for (int i = 0; i < 3; i++) { HIPCHK(hipMemsetAsync(c->d_counter, 0, 8 * sizeof(int), stream)); launch(c, c->max_wg, stream, 0, 1, 0, 0); }
            HIPCHK(hipGetLastError());
        }
        last_kind[dv].store(kind);
    }
    HIPCHK(hipMemcpyAsync(c->d_jobs, c->h_jobs, sizeof(FrameJob) * n, hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemcpyAsync(c->d_hdrs, c->h_hdrs, (size_t)HDR_MAX * n, hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemsetAsync(c->d_counter, 0, 4 * sizeof(int), stream));
    HIPCHK(hipMemsetAsync(c->d_counter + 4, 0xFF, 2 * sizeof(int), stream));      // earliest start: a minimum
    HIPCHK(hipMemsetAsync(c->d_counter + 6, 0, 2 * sizeof(int), stream));
    if (mode > 1) {
        HIPCHK(hipMemsetAsync(c->d_mail, 0, sizeof(TeamMail) * nmains, stream));     // sequence numbers restart with every launch (nmains <= mail_cap: checked above)
        HIPCHK(hipMemsetAsync(c->d_pq, 0, sizeof(PoolQ), stream));
    }
    c->last_mains = nmains; c->last_help = nhelp;
    c->last_pipe = use_pipe;
    c->last_wide = use_wide;
    c->last_split = use_split;
    c->last_part = npart;

    HIPCHK(hipEventRecord(c->ev0, stream));
    if (use_split) {          // the main workgroups (wide) on their compute units, the helpers (192 threads) on the others; both end before `stream` goes on
        HIPCHK(hipEventRecord(c->ev_split[0], stream));
        HIPCHK(hipStreamWaitEvent(c->st_split[0], c->ev_split[0], 0)); HIPCHK(hipStreamWaitEvent(c->st_split[1], c->ev_split[0], 0));
        const int all = nhelp >= 2 * nmains ? 1000 : -1;      // two helpers per main workgroup: everything is offered
        launch(c, nmains, c->st_split[0], n, mode, nmains, nhelp, 2, 1, 0, all, all);
        launch(c, nhelp, c->st_split[1], n, mode, nmains, nhelp, 0, 2, nmains, all, all);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(c->ev_split[1], c->st_split[0])); HIPCHK(hipEventRecord(c->ev_split[2], c->st_split[1]));
        HIPCHK(hipStreamWaitEvent(stream, c->ev_split[1], 0)); HIPCHK(hipStreamWaitEvent(stream, c->ev_split[2], 0));
    } else {
    launch(c, grid, stream, n, mode, nmains, nhelp, c->last_wide ? 2 : c->last_pipe, 0, 0, -1, -1, npart);
    HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(c->ev1, stream));
    c->timed = true;
    return 0;
}

extern "C" int imcvt_hevc_debug_occupancy(int *blocks_per_cu, int *cus, int *lds_per_block, int *lds_per_cu) {
    if (!have_device()) return IMCVT_ERR_NO_DEVICE;
    int nb = 0, dev = 0; hipDeviceProp_t prop; hipFuncAttributes fa;
    HIPCHK(hipGetDevice(&dev)); HIPCHK(hipGetDeviceProperties(&prop, dev));
    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, hevc_encode_frames, WG_THREADS, 0));
    HIPCHK(hipFuncGetAttributes(&fa, (const void *)hevc_encode_frames));
    if (blocks_per_cu) *blocks_per_cu = nb;
    if (cus) *cus = prop.multiProcessorCount;
    if (lds_per_block) *lds_per_block = (int)fa.sharedSizeBytes;
    if (lds_per_cu) *lds_per_cu = (int)prop.maxSharedMemoryPerMultiProcessor;
    return 0;
}

extern "C" int imcvt_hevc_debug_census(imcvt_hevc_ctx *c, int grid) {
    if (!c || grid < 1) return IMCVT_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    if (c->timed) HIPCHK(hipEventSynchronize(c->ev1));
    return census(c, grid, 0);
}
extern "C" int imcvt_hevc_residency(imcvt_hevc_ctx *c, int *max_wg, int *pipe_wg, int *occ_per_cu, int *occ_pipe_per_cu, int *census_wg, int *census_pipe) {
    if (!c) return IMCVT_ERR_ARG;
    if (max_wg) *max_wg = c->max_wg;
    if (pipe_wg) *pipe_wg = c->pipe_wg;
    if (occ_per_cu) *occ_per_cu = c->occ_wg;
    if (occ_pipe_per_cu) *occ_pipe_per_cu = c->occ_pipe;
    if (census_wg) *census_wg = c->census_wg;
    if (census_pipe) *census_pipe = c->census_pipe;
    return c->cus;
}
// Test aid: a co-tenant.  `grid` workgroups of 256 threads that hold `lds_bytes` of LDS each and spin for about `ms` milliseconds on
// `stream` — what another kernel on the same device does to this library's launches (tests/test_gpu_parity.py).
__global__ void imcvt_filler(int ticks100) {
    extern __shared__ int filler_lds[];
    if (threadIdx.x == 0) { filler_lds[0] = ticks100; const unsigned long long t0 = wall_clock64(); while (wall_clock64() - t0 < (unsigned long long)filler_lds[0]) __builtin_amdgcn_s_sleep(64); }
    __syncthreads();
}
extern "C" int imcvt_hevc_debug_filler(int grid, int lds_bytes, int ms, void *stream) {
    if (grid < 1 || lds_bytes < 4 || lds_bytes > 160 * 1024 || ms < 0) return IMCVT_ERR_ARG;
    if (lds_bytes > 64 * 1024) HIPCHK(hipFuncSetAttribute((const void *)imcvt_filler, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipLaunchKernelGGL(imcvt_filler, dim3(grid), dim3(256), lds_bytes, (hipStream_t)stream, ms * 100000);
    HIPCHK(hipGetLastError());
    return 0;
}

// Waits for the context's last launch and reports how it ended: 0, or IMCVT_ERR_WATCHDOG when a wait between workgroups gave up
// (hevc_frame.h wd_poll) — the launch's outputs are then invalid.
extern "C" int imcvt_hevc_last_status(imcvt_hevc_ctx *c) {
    if (!c) return IMCVT_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    if (!c->timed) return 0;
    HIPCHK(hipEventSynchronize(c->ev1));
    const int earlier = c->pending_err; c->pending_err = 0;      // an earlier launch nobody asked about (imcvt_hevc_encode_device latched it)
    if (c->last_help <= 0) return earlier;
    u32 v[16] = { 0 };
    HIPCHK(hipMemcpy(v, &c->d_pq->abort, sizeof v, hipMemcpyDeviceToHost));
    if (v[0] == 0) return earlier;
    fprintf(stderr, "imcvt_hevc: device watchdog: a %s gave up after %.0f s (slot/shard %u, seq/ticket %u, main workgroup %u, frame %d; launch %d + %d workgroups) — results invalid\n",
            v[1] == 1 ? "main workgroup waiting for a helper's answer" : v[1] == 3 ? "main workgroup waiting for its partner's answer" : "helper waiting for a ticket's owner", (double)WD_TICKS / 1e8, v[2], v[3], v[4], (int)v[5], c->last_mains, c->last_help);
    if (v[1] == 1) fprintf(stderr, "  when it gave up: its shard's queue of that kind had head %u tail %u; request flag %#x (0x1xxxx: taken by the helper with that home shard)\n", v[6], v[7], v[8]);
    if (v[1] == 1) fprintf(stderr, "  clocks (ms before the watchdog fired): wait began %.2f, request taken %.2f, inputs staged %.2f, candidates evaluated %.2f, answer about to be published %.2f (stamps of an earlier request of this mailbox if larger than the wait)\n",
                           (double)(i32)(v[13] - v[14]) / 1e5, (double)(i32)(v[13] - v[9]) / 1e5, (double)(i32)(v[13] - v[10]) / 1e5, (double)(i32)(v[13] - v[11]) / 1e5, (double)(i32)(v[13] - v[12]) / 1e5);
    if (getenv("IMCVT_HEVC_VERBOSE")) {            // the queues as the watchdog left them
        PoolQ *q = new PoolQ();
        if (hipMemcpy(q, c->d_pq, sizeof(PoolQ), hipMemcpyDeviceToHost) == hipSuccess) {
            fprintf(stderr, "  frames_done %u alive %u mains_taken %u progress %u\n", q->frames_done, q->alive, q->mains_taken, q->progress);
            for (int i = 0; i < POOL_SHARDS; i++) fprintf(stderr, "  shard %2d: 16x16 head %u tail %u | 32x32 head %u tail %u\n", i, q->sh[i].head[0], q->sh[i].tail[0], q->sh[i].head[1], q->sh[i].tail[1]);
            if (v[1] == 1 && (int)v[4] < c->mail_cap && v[2] < POOL_KINDS) {
                MailSlot *ms = new MailSlot();
                if (hipMemcpy(ms, &c->d_mail[v[4]].s[v[2]], sizeof(MailSlot), hipMemcpyDeviceToHost) == hipSuccess)
                    fprintf(stderr, "  its mailbox: request seq %d (op %d frame %d cy %d cx %d N %d y0 %d x0 %d), result flag %d\n", ms->req.seq, ms->req.op, ms->req.frame, ms->req.cy, ms->req.cx, ms->req.N, ms->req.y0, ms->req.x0, ms->res_flag);
                const PoolShard &sh = q->sh[v[4] % POOL_SHARDS];
                for (int k = 0; k < POOL_QCAP; k++) if (sh.ring[v[2]][k]) fprintf(stderr, "  its shard's ring[%u][%d] = %u\n", v[2], k, sh.ring[v[2]][k]);
                delete ms;
            }
        }
        delete q;
    }
    return IMCVT_ERR_WATCHDOG;
}

extern "C" int imcvt_hevc_last_resident(imcvt_hevc_ctx *c) {
    if (!c) return IMCVT_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    if (c->timed) HIPCHK(hipEventSynchronize(c->ev1));
    int v = 0;
    HIPCHK(hipMemcpy(&v, c->d_counter + 3, sizeof v, hipMemcpyDeviceToHost));
    return v;
}
extern "C" long long imcvt_hevc_last_start_spread_us(imcvt_hevc_ctx *c) {
    if (!c) return IMCVT_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    if (c->timed) HIPCHK(hipEventSynchronize(c->ev1));
    unsigned long long v[2] = { 0, 0 };
    HIPCHK(hipMemcpy(v, c->d_counter + 4, sizeof v, hipMemcpyDeviceToHost));
    return v[1] >= v[0] ? (long long)((v[1] - v[0]) / 100) : -1;
}

extern "C" int imcvt_hevc_debug_prof(imcvt_hevc_ctx *c, unsigned long long *out, int n, int reset) {
    if (!c || !out) return IMCVT_ERR_ARG;
    const int have = 3 * NWAVES * PF_N + REG_N;      // (phase cycles of -DIMCVT_PROF builds, then the region counters of -DIMCVT_REGCNT builds)
    HIPCHK(hipSetDevice(c->device));
    if (hipDeviceSynchronize() != hipSuccess) return IMCVT_ERR_HIP;
    if (hipMemcpy(out, c->d_prof, sizeof(unsigned long long) * (n < have ? n : have), hipMemcpyDeviceToHost) != hipSuccess) return IMCVT_ERR_HIP;
    if (reset && hipMemset(c->d_prof, 0, sizeof(unsigned long long) * have) != hipSuccess) return IMCVT_ERR_HIP;
    return have;
}

extern "C" float imcvt_hevc_last_kernel_ms(imcvt_hevc_ctx *c) {
    if (!c || !c->timed) return -1.f;
    float ms = -1.f;
    if (hipEventSynchronize(c->ev1) != hipSuccess || hipEventElapsedTime(&ms, c->ev0, c->ev1) != hipSuccess) return -1.f;
    return ms;
}

// ---------------------------------------------------------------------------------------------------
// Host-pointer entry points (the reference's own interface).  A batch fans out over every visible device: frame i goes
// to device i mod D; each device has its own context, streams and a grow-only slab that is reused from call to call.
//
// Transfers (round 6).  The caller's buffers are pageable; a pageable copy runs at ~4 GB/s on this platform, and in round 5 all of them sat
// before (512 inputs) and after (1024 outputs) the launch: 0.63 s beside 5.0 s of kernel.  Now
//   in   large batches go through a few pinned staging buffers filled by worker threads (memcpy + asynchronous copy per 8 MB chunk, two
//        buffers per thread), so the link, not the page tables, sets the pace;
//   out  every frame carries a progress record in pinned host memory (FrameJob::prog, hevc_frame.h publish_progress: CTU rows whose
//        reconstruction is final, stream bytes that are final), and the calling thread copies finished rows and bytes to the caller's buffers
//        WHILE the launch runs; what is left when the kernel ends is the last rows of the last frames.
// Small batches (the reference's one-picture calls) keep the plain path: copies, launch, copies.
// ---------------------------------------------------------------------------------------------------
#ifndef UP_CHUNK
#define UP_CHUNK ((size_t)8 << 20)      // bytes per staging buffer
#endif
#ifndef UP_LANES
#define UP_LANES 3                      // upload threads per device, two staging buffers each (streams per device stay at four: kernel, copy-out / lane 0, lanes 1 and 2)
#endif
#ifndef STAGE_MIN_BYTES
#define STAGE_MIN_BYTES ((size_t)32 << 20)      // batches with less input than this are copied as before
#endif
#ifndef FOLLOW_MIN_BYTES
#define FOLLOW_MIN_BYTES ((size_t)16 << 20)     // batches with less output than this are collected after the launch, as before
#endif
#ifndef FOLLOW_ROWS
#define FOLLOW_ROWS 2                   // CTU rows of a frame that are worth a copy of their own while the launch runs
#endif
struct UpLane { hipStream_t st = nullptr; bool own_stream = false; u8 *buf[2] = { nullptr, nullptr }; hipEvent_t ev[2] = { nullptr, nullptr }; bool busy[2] = { false, false }; };
struct DevState {
    int dev = 0;
    imcvt_hevc_ctx *ctx = nullptr;
    hipStream_t st = nullptr, st_copy = nullptr;
    u8 *slab = nullptr; size_t slab_cap = 0;
    std::vector<int> idx;                       // frames of the current call
    std::vector<size_t> off_img, off_out, off_rc;
    size_t off_len = 0;
    std::vector<imcvt_hevc_frame> fr;
    std::vector<int> lens;
    UpLane up[UP_LANES];                        // staging (created by the first batch that is large enough)
    u32 *h_prog = nullptr; int prog_cap = 0;    // pinned progress records of the running launch, two words per frame
    std::vector<u32> got_rows, got_pos;         // CTU rows / stream bytes of each frame that have reached the caller's buffers
    int rc = 0; bool launched = false, staged_out = false;
    double t_up = 0, t_follow = 0, t_tail = 0; size_t followed = 0, tail = 0;      // statistics of the last call (imcvt_hevc_batch_transfer_stats)
};
static std::mutex g_lock;
static std::vector<DevState> g_devs;
static int g_last_devices = 0;
static double g_xfer[6];                        // last batch: seconds uploading, following, collecting the tail; bytes copied while the launch ran / after it

static int dev_init(DevState &d) {
    if (d.ctx) return 0;
    HIPCHK(hipSetDevice(d.dev));
    d.ctx = imcvt_hevc_create(0);
    if (!d.ctx) return IMCVT_ERR_NO_DEVICE;
    HIPCHK(hipStreamCreateWithFlags(&d.st, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&d.st_copy, hipStreamNonBlocking));
    return 0;
}
static void dev_release(DevState &d) {
    if (!d.ctx) return;
    (void)hipSetDevice(d.dev);
    (void)hipStreamSynchronize(d.st);
    imcvt_hevc_destroy(d.ctx); d.ctx = nullptr;
    for (UpLane &u : d.up) {
        for (int k = 0; k < 2; k++) { if (u.buf[k]) (void)hipHostFree(u.buf[k]); if (u.ev[k]) (void)hipEventDestroy(u.ev[k]); u.buf[k] = nullptr; u.ev[k] = nullptr; u.busy[k] = false; }
        if (u.own_stream && u.st) (void)hipStreamDestroy(u.st);
        u.st = nullptr; u.own_stream = false;
    }
    if (d.h_prog) (void)hipHostFree(d.h_prog);
    d.h_prog = nullptr; d.prog_cap = 0;
    (void)hipStreamDestroy(d.st); d.st = nullptr;
    (void)hipStreamDestroy(d.st_copy); d.st_copy = nullptr;
    (void)hipFree(d.slab); d.slab = nullptr; d.slab_cap = 0;
}

extern "C" int imcvt_hevc_batch_devices(void) { return g_last_devices; }
extern "C" double imcvt_hevc_batch_kernel_ms(void) { std::lock_guard<std::mutex> guard(g_lock); return g_xfer[5]; }      // the longest kernel time (HIP events) over the devices of the last host-pointer batch
extern "C" void imcvt_hevc_batch_transfer_stats(double *upload_s, double *follow_s, double *tail_s, double *bytes_during, double *bytes_after) {
    std::lock_guard<std::mutex> guard(g_lock);
    if (upload_s) *upload_s = g_xfer[0];
    if (follow_s) *follow_s = g_xfer[1];
    if (tail_s) *tail_s = g_xfer[2];
    if (bytes_during) *bytes_during = g_xfer[3];
    if (bytes_after) *bytes_after = g_xfer[4];
}

extern "C" void imcvt_hevc_shutdown(void) {
    std::lock_guard<std::mutex> guard(g_lock);
    for (DevState &d : g_devs) dev_release(d);
    g_devs.clear();
}

struct BatchArgs { unsigned char *const *pbuffers; const unsigned char *const *imgs; unsigned char *const *rcons; int *ysz, *xsz; const int *qpd6; int *out_len; };
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static int staging_ready(DevState &d);
// inputs of one device's frames through the staging buffers: the chunks of all frames are one list, the lanes take them in turn
static int upload_staged(DevState &d, const BatchArgs &A) {
    struct Item { const u8 *src; u8 *dst; size_t len; };
    std::vector<Item> items;
    for (size_t j = 0; j < d.idx.size(); j++) {
        const int i = d.idx[j];
        const size_t n = (size_t)A.ysz[i] * A.xsz[i];
        for (size_t o = 0; o < n; o += UP_CHUNK) items.push_back({ A.imgs[i] + o, d.slab + d.off_img[j] + o, n - o < UP_CHUNK ? n - o : UP_CHUNK });
    }
    if (int e = staging_ready(d)) return e;
    std::atomic<size_t> next(0);
    std::atomic<int> err(0);
    auto lane = [&](int k) {
        UpLane &u = d.up[k];
        if (hipSetDevice(d.dev) != hipSuccess) { err = IMCVT_ERR_HIP; return; }
        for (int turn = 0; err == 0; turn ^= 1) {
            const size_t it = next++;
            if (it >= items.size()) break;
            if (u.busy[turn] && hipEventSynchronize(u.ev[turn]) != hipSuccess) { err = IMCVT_ERR_HIP; break; }
            memcpy(u.buf[turn], items[it].src, items[it].len);
            if (hipMemcpyAsync(items[it].dst, u.buf[turn], items[it].len, hipMemcpyHostToDevice, u.st) != hipSuccess || hipEventRecord(u.ev[turn], u.st) != hipSuccess) { err = IMCVT_ERR_HIP; break; }
            u.busy[turn] = true;
        }
        if (hipStreamSynchronize(u.st) != hipSuccess) err = IMCVT_ERR_HIP;
        u.busy[0] = u.busy[1] = false;
    };
    std::vector<std::thread> th;
    for (int k = 1; k < UP_LANES; k++) th.emplace_back(lane, k);
    lane(0);
    for (std::thread &t : th) t.join();
    return err;
}

// what the device has finished of frame j and the caller's buffers do not have yet, as copy items.  fin: the launch is over (everything that is left);
// otherwise only what the frame's progress record says is final, and only in pieces worth a copy.
struct OutItem { const u8 *src; u8 *dst; size_t len; };
static void collect_frame(DevState &d, const BatchArgs &A, int j, bool fin, std::vector<OutItem> &items) {
    const int i = d.idx[(size_t)j], hp = imcvt_hevc_padded(A.ysz[i]), wp = imcvt_hevc_padded(A.xsz[i]);
    u32 rows, pos;
    if (fin) { rows = (u32)(hp / 32); pos = (u32)d.lens[(size_t)j]; }
    else {
        const volatile u32 *pr = d.h_prog + 2 * (size_t)j;
        const u32 r = pr[0]; pos = pr[1]; rows = r & ~PROG_DONE;
        if (rows > (u32)(hp / 32) || pos > (u32)imcvt_hevc_stream_bound(A.ysz[i], A.xsz[i])) return;      // (never: a record that makes no sense is ignored, the final pass collects the frame)
        if (!(r & PROG_DONE)) {
            if (rows < d.got_rows[(size_t)j] + FOLLOW_ROWS) rows = d.got_rows[(size_t)j];
            if (pos < d.got_pos[(size_t)j] + 65536u) pos = d.got_pos[(size_t)j];
        }
    }
    if (rows > d.got_rows[(size_t)j]) {
        const size_t o = (size_t)d.got_rows[(size_t)j] * 32 * wp, n = (size_t)(rows - d.got_rows[(size_t)j]) * 32 * wp;
        items.push_back({ d.fr[(size_t)j].d_rcon + o, A.rcons[i] + o, n });
        d.got_rows[(size_t)j] = rows;
    }
    if (pos > d.got_pos[(size_t)j]) {
        const size_t o = d.got_pos[(size_t)j], n = (size_t)pos - o;
        items.push_back({ d.fr[(size_t)j].d_out + o, A.pbuffers[i] + o, n });
        d.got_pos[(size_t)j] = pos;
    }
}
static int staging_ready(DevState &d) {      // streams, pinned buffers and events of the lanes (created by the first batch that is large enough)
    for (int k = 0; k < UP_LANES; k++) {
        UpLane &u = d.up[k];
        if (!u.st) { if (k == 0) u.st = d.st_copy; else { if (hipStreamCreateWithFlags(&u.st, hipStreamNonBlocking) != hipSuccess) return IMCVT_ERR_HIP; u.own_stream = true; } }
        for (int b = 0; b < 2; b++) {
            if (!u.buf[b] && hipHostMalloc((void **)&u.buf[b], UP_CHUNK, hipHostMallocDefault) != hipSuccess) return IMCVT_ERR_HIP;
            if (!u.ev[b] && hipEventCreateWithFlags(&u.ev[b], hipEventDisableTiming) != hipSuccess) return IMCVT_ERR_HIP;
        }
    }
    return 0;
}
// copy items out.  how 0: straight into the caller's (pageable) memory, asynchronous on st_copy; how 1: through lane 0's pinned buffers by the calling thread
// (device -> pinned by the copy engine, pinned -> caller by memcpy); how 2: through all lanes' buffers by worker threads (what is left when the launch has ended)
static int copy_out(DevState &d, const std::vector<OutItem> &items, int how, size_t *bytes) {
    size_t total = 0;
    for (const OutItem &it : items) total += it.len;
    *bytes += total;
    if (items.empty()) return 0;
    if (how == 0) {
        for (const OutItem &it : items) if (hipMemcpyAsync(it.dst, it.src, it.len, hipMemcpyDeviceToHost, d.st_copy) != hipSuccess) return IMCVT_ERR_HIP;
        return 0;
    }
    if (int e = staging_ready(d)) return e;
    std::vector<OutItem> chunks;
    for (const OutItem &it : items) for (size_t o = 0; o < it.len; o += UP_CHUNK) chunks.push_back({ it.src + o, it.dst + o, it.len - o < UP_CHUNK ? it.len - o : UP_CHUNK });
    std::atomic<size_t> next(0);
    std::atomic<int> err(0);
    auto lane = [&](int k) {
        UpLane &u = d.up[k];
        if (hipSetDevice(d.dev) != hipSuccess) { err = IMCVT_ERR_HIP; return; }
        size_t mine[2] = { 0, 0 }; bool have[2] = { false, false };
        for (int turn = 0; err == 0; turn ^= 1) {      // the copy engine fills one buffer while this thread empties the other
            const size_t it = next++;
            const bool more = it < chunks.size();
            if (more) {
                if (hipMemcpyAsync(u.buf[turn], chunks[it].src, chunks[it].len, hipMemcpyDeviceToHost, u.st) != hipSuccess || hipEventRecord(u.ev[turn], u.st) != hipSuccess) { err = IMCVT_ERR_HIP; break; }
                mine[turn] = it; have[turn] = true;
            }
            const int other = turn ^ 1;
            if (have[other]) {
                if (hipEventSynchronize(u.ev[other]) != hipSuccess) { err = IMCVT_ERR_HIP; break; }
                memcpy(chunks[mine[other]].dst, u.buf[other], chunks[mine[other]].len);
                have[other] = false;
            }
            if (!more) {
                if (have[turn]) { if (hipEventSynchronize(u.ev[turn]) != hipSuccess) { err = IMCVT_ERR_HIP; break; } memcpy(chunks[mine[turn]].dst, u.buf[turn], chunks[mine[turn]].len); have[turn] = false; }
                break;
            }
        }
        if (err != 0) (void)hipStreamSynchronize(u.st);
    };
    if (how == 1) { lane(0); return err; }
    std::vector<std::thread> th;
    for (int k = 1; k < UP_LANES; k++) th.emplace_back(lane, k);
    lane(0);
    for (std::thread &t : th) t.join();
    return err;
}

// one device's share of a batch, start to finish (its own thread when the batch spans several devices)
static void run_device(DevState &d, const BatchArgs &A) {
    d.rc = 0; d.launched = false; d.t_up = d.t_follow = d.t_tail = 0; d.followed = d.tail = 0;
    const int m = (int)d.idx.size();
    auto fail = [&](int e) { if (d.rc == 0) d.rc = e; };
    if (hipSetDevice(d.dev) != hipSuccess) { fail(IMCVT_ERR_HIP); return; }
    if (int e = dev_init(d)) { fail(e); return; }
    d.off_img.resize((size_t)m); d.off_out.resize((size_t)m); d.off_rc.resize((size_t)m); d.fr.resize((size_t)m); d.lens.assign((size_t)m, 0);
    d.got_rows.assign((size_t)m, 0u); d.got_pos.assign((size_t)m, 0u);
    size_t total = 0, bytes_in = 0, bytes_out = 0;       // one device slab: [img | out | rcon] per frame, then the lengths
    for (int j = 0; j < m; j++) {
        // the reference indexes img with the ORIGINAL stride but only up to the padded (<=8192) extent (:1621)
        const int i = d.idx[(size_t)j], hp = imcvt_hevc_padded(A.ysz[i]), wp = imcvt_hevc_padded(A.xsz[i]);
        d.off_img[(size_t)j] = total; total += align256((size_t)A.ysz[i] * A.xsz[i]); bytes_in += (size_t)A.ysz[i] * A.xsz[i];
        d.off_out[(size_t)j] = total; total += align256((size_t)imcvt_hevc_stream_bound(A.ysz[i], A.xsz[i]));
        d.off_rc[(size_t)j] = total;  total += align256((size_t)hp * wp); bytes_out += (size_t)hp * wp;
    }
    d.off_len = total; total += align256(sizeof(int) * (size_t)m);
    if (total > d.slab_cap) {
        (void)hipFree(d.slab); d.slab = nullptr; d.slab_cap = 0;
        if (hipMalloc(&d.slab, total) != hipSuccess) { fprintf(stderr, "imcvt_hevc: cannot allocate %zu bytes on device %d\n", total, d.dev); fail(IMCVT_ERR_HIP); return; }
        d.slab_cap = total;
    }
    for (int j = 0; j < m; j++) {
        const int i = d.idx[(size_t)j];
        imcvt_hevc_frame &f = d.fr[(size_t)j];
        f.d_img = d.slab + d.off_img[(size_t)j]; f.d_out = d.slab + d.off_out[(size_t)j]; f.d_rcon = d.slab + d.off_rc[(size_t)j];
        f.d_len = (int *)(d.slab + d.off_len) + j; f.h = A.ysz[i]; f.w = A.xsz[i]; f.qpd6 = A.qpd6[i];
    }
    const bool plain = getenv("IMCVT_HEVC_PLAIN_COPIES") != nullptr;      // (A/B and tests: the round-5 path for every batch)
    double t0 = now_s();
    if (!plain && bytes_in >= STAGE_MIN_BYTES) { if (int e = upload_staged(d, A)) { fail(e); return; } }
    else for (int j = 0; j < m; j++) {
        const int i = d.idx[(size_t)j];
        if (hipMemcpyAsync(d.slab + d.off_img[(size_t)j], A.imgs[i], (size_t)A.ysz[i] * A.xsz[i], hipMemcpyHostToDevice, d.st) != hipSuccess) { fail(IMCVT_ERR_HIP); return; }
    }
    d.t_up = now_s() - t0;
    // IMCVT_HEVC_FOLLOW (A/B): 0 collect after the launch; 1 follow it with copies straight into the caller's memory; 2 records on, collect after; 3 (default) follow it
    // through the pinned staging buffers
    int fmode = 3;
    if (const char *e = getenv("IMCVT_HEVC_FOLLOW")) fmode = atoi(e);
    const bool follow = !plain && fmode != 0 && bytes_out >= FOLLOW_MIN_BYTES;
    d.staged_out = !plain && bytes_out >= FOLLOW_MIN_BYTES;
    if (follow) {
        if (m > d.prog_cap) {
            if (d.h_prog) (void)hipHostFree(d.h_prog);
            d.h_prog = nullptr; d.prog_cap = 0;
            if (hipHostMalloc((void **)&d.h_prog, sizeof(u32) * 2 * (size_t)m, hipHostMallocDefault) != hipSuccess) { fail(IMCVT_ERR_HIP); return; }
            d.prog_cap = m;
        }
        memset(d.h_prog, 0, sizeof(u32) * 2 * (size_t)m);
    }
    imcvt_hevc_set_progress(d.ctx, follow ? d.h_prog : nullptr);
    if (int e = imcvt_hevc_encode_device(d.ctx, m, d.fr.data(), d.st)) { fail(e); return; }       // asynchronous
    d.launched = true;
    t0 = now_s();
    if (follow) {
        for (;;) {
            const hipError_t q = hipEventQuery(d.ctx->ev1);
            if (q != hipErrorNotReady) { if (q != hipSuccess) { (void)hipGetLastError(); fail(IMCVT_ERR_HIP); } break; }
            size_t got = 0;
            if (fmode != 2) {
                std::vector<OutItem> items;
                for (int j = 0; j < m; j++) collect_frame(d, A, j, false, items);
                if (int e = copy_out(d, items, fmode == 1 ? 0 : 1, &got)) { fail(e); break; }
            }
            d.followed += got;
            if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
    }
    d.t_follow = now_s() - t0;
}
static void finish_device(DevState &d, const BatchArgs &A) {      // the launch is drained even after an error; then whatever has not been copied yet
    if (!d.ctx || d.idx.empty()) return;
    const int m = (int)d.idx.size();
    auto fail = [&](int e) { if (d.rc == 0) d.rc = e; };
    const double t0 = now_s();
    if (hipSetDevice(d.dev) != hipSuccess) { fail(IMCVT_ERR_HIP); return; }
    if (hipStreamSynchronize(d.st) != hipSuccess) fail(IMCVT_ERR_HIP);
    if (hipStreamSynchronize(d.st_copy) != hipSuccess) fail(IMCVT_ERR_HIP);
    if (!d.launched) return;
    if (d.rc == 0) d.rc = imcvt_hevc_last_status(d.ctx);
    if (d.rc == 0 && hipMemcpy(d.lens.data(), d.slab + d.off_len, sizeof(int) * (size_t)m, hipMemcpyDeviceToHost) != hipSuccess) fail(IMCVT_ERR_HIP);
    std::vector<OutItem> items;
    for (int j = 0; j < m && d.rc == 0; j++) {
        const int i = d.idx[(size_t)j];
        if (d.lens[(size_t)j] < 1 || (long long)d.lens[(size_t)j] > imcvt_hevc_stream_bound(A.ysz[i], A.xsz[i]) || (u32)d.lens[(size_t)j] < d.got_pos[(size_t)j]) { fail(IMCVT_ERR_HIP); break; }
        collect_frame(d, A, j, true, items);
    }
    if (d.rc == 0) {
        size_t left = 0;
        for (const OutItem &it : items) left += it.len;
        if (int e = copy_out(d, items, (d.staged_out && left >= ((size_t)4 << 20)) ? 2 : 0, &d.tail)) fail(e);
    }
    if (hipStreamSynchronize(d.st_copy) != hipSuccess) fail(IMCVT_ERR_HIP);
    d.t_tail = now_s() - t0;
}

// One merged batch on the devices (called by one thread at a time: the submission queue's leader, below).
static int encode_batch_on_devices(int n, unsigned char *const *pbuffers, const unsigned char *const *imgs,
                                   unsigned char *const *rcons, int *ysz, int *xsz, const int *qpd6, int *out_len) {
    if (!have_device()) return IMCVT_ERR_NO_DEVICE;
    std::lock_guard<std::mutex> guard(g_lock);
    int prev_dev = 0;
    (void)hipGetDevice(&prev_dev);
    if (g_devs.empty()) {
        int nd = 0;
        HIPCHK(hipGetDeviceCount(&nd));
        if (const char *e = getenv("IMCVT_HEVC_DEVICES")) { const int lim = atoi(e); if (lim >= 1 && lim < nd) nd = lim; }
        // Test seam: IMCVT_HEVC_FAKE_DEVICES=k runs the fan-out below with k LOGICAL devices that all sit on physical device 0, each with its
        // own context, stream and slab — the i mod D split, the per-device slabs and the collection order on a box with one GPU.
        int fake = 0;
        if (const char *e = getenv("IMCVT_HEVC_FAKE_DEVICES")) { fake = atoi(e); if (fake < 1 || fake > 64) fake = 0; }
        g_devs.resize(fake ? fake : nd);
        for (int i = 0; i < (int)g_devs.size(); i++) g_devs[i].dev = fake ? 0 : i;
    }
    if (n == 0) return 0;
    // a device joins when it gets at least one frame; with one frame the caller's current device does the work
    const int D = n < (int)g_devs.size() ? n : (int)g_devs.size();
    const int first = (D == 1 && prev_dev < (int)g_devs.size()) ? prev_dev : 0;
    const BatchArgs A = { pbuffers, imgs, rcons, ysz, xsz, qpd6, out_len };
    auto dev_of = [&](int k) -> DevState & { return g_devs[(size_t)((first + k) % (int)g_devs.size())]; };
    for (int k = 0; k < D; k++) { DevState &d = dev_of(k); d.idx.clear(); for (int i = k; i < n; i += D) d.idx.push_back(i); }
    {   // upload, launch and follow: one thread per device (this thread takes the first)
        std::vector<std::thread> th;
        for (int k = 1; k < D; k++) th.emplace_back([&, k]() { run_device(dev_of(k), A); });
        run_device(dev_of(0), A);
        for (std::thread &t : th) t.join();
    }
    int rc = 0;
    for (int k = 0; k < D; k++) if (rc == 0) rc = dev_of(k).rc;
    for (int k = 0; k < D; k++) { DevState &d = dev_of(k); if (rc != 0 && d.rc == 0) d.rc = rc; finish_device(d, A); if (rc == 0) rc = d.rc; }      // (after an error nothing more is copied, but every stream is drained)
    for (int k = 0; k < 6; k++) g_xfer[k] = 0;
    for (int k = 0; k < D; k++) {
        DevState &d = dev_of(k);
        if (rc == 0) for (size_t j = 0; j < d.idx.size(); j++) { const int i = d.idx[j]; out_len[i] = d.lens[j]; ysz[i] = imcvt_hevc_padded(ysz[i]); xsz[i] = imcvt_hevc_padded(xsz[i]); }
        g_xfer[0] = d.t_up > g_xfer[0] ? d.t_up : g_xfer[0]; g_xfer[1] = d.t_follow > g_xfer[1] ? d.t_follow : g_xfer[1]; g_xfer[2] += d.t_tail;
        g_xfer[3] += (double)d.followed; g_xfer[4] += (double)d.tail;
        if (d.launched) { const double km = (double)imcvt_hevc_last_kernel_ms(d.ctx); if (km > g_xfer[5]) g_xfer[5] = km; }
        d.idx.clear();
    }
    g_last_devices = D;
    (void)hipSetDevice(prev_dev);
    return rc;
}

// ---------------------------------------------------------------------------------------------------
// Submission queue.  The reference's entry point is re-entrant (src/HEVCe/HEVCe.c:1569 has no mutable globals): N threads may call
// it at once.  Here one frame alone occupies a handful of workgroups for seconds, so concurrent callers must not queue up behind
// each other: calls that arrive while a batch is being formed (a short window) or while the previous one runs are merged into ONE
// device batch.  The first caller to find no leader becomes the leader: it waits the window, takes everything submitted so far,
// runs it as one batch, hands every caller its results and gives the leadership up; a caller that is still waiting takes it over.
// ---------------------------------------------------------------------------------------------------
typedef int (*batch_backend_t)(int, unsigned char *const *, const unsigned char *const *, unsigned char *const *, int *, int *, const int *, int *);
struct Submission {
    int n; unsigned char *const *pbuffers; const unsigned char *const *imgs; unsigned char *const *rcons; int *ysz, *xsz; const int *qpd6; int *out_len;
    int rc = 0; bool done = false;
};
static std::mutex g_qmu;
static std::condition_variable g_qcv;
static std::vector<Submission *> g_pending;
static bool g_leader = false;
static batch_backend_t g_backend = encode_batch_on_devices;
static long g_q_calls = 0, g_q_batches = 0, g_q_max_batch = 0;
#ifndef MERGE_MAX_FRAMES
#define MERGE_MAX_FRAMES 2048      // frames one merged round takes at most (whole submissions; a single larger call still goes through alone)
#endif
static int coalesce_window_us() {
    static int us = -1;
    if (us < 0) { const char *e = getenv("IMCVT_HEVC_COALESCE_US"); us = e ? atoi(e) : 300; if (us < 0) us = 0; if (us > 1000000) us = 1000000; }
    return us;
}
// Test aids: a stand-in for the device batch (NULL: the real one) so that the queue's logic runs without a GPU; its counters.
extern "C" void imcvt_hevc_debug_set_backend(void *fn) { std::lock_guard<std::mutex> g(g_qmu); g_backend = fn ? (batch_backend_t)fn : encode_batch_on_devices; }
extern "C" void imcvt_hevc_coalesce_stats(long *calls, long *batches, long *max_batch, int reset) {
    std::lock_guard<std::mutex> g(g_qmu);
    if (calls) *calls = g_q_calls;
    if (batches) *batches = g_q_batches;
    if (max_batch) *max_batch = g_q_max_batch;
    if (reset) { g_q_calls = 0; g_q_batches = 0; g_q_max_batch = 0; }
}
static std::chrono::steady_clock::time_point g_last_crowd;           // when callers were last seen arriving together
static void lead_one_round(std::unique_lock<std::mutex> &lk) {       // called with the queue locked and g_leader == true; returns the same way
    // The window is only worth its 300 us when callers do arrive together: a second submission is already waiting, or a round of the
    // last 100 ms carried more than one.  A lone caller (the reference's serial file loop) goes straight through.
    const int win = coalesce_window_us();
    const auto now = std::chrono::steady_clock::now();
    const bool crowd = g_pending.size() > 1 || (g_last_crowd.time_since_epoch().count() != 0 && now - g_last_crowd < std::chrono::milliseconds(100));
    if (win > 0 && crowd) { lk.unlock(); std::this_thread::sleep_for(std::chrono::microseconds(win)); lk.lock(); }
    if (g_pending.size() > 1) g_last_crowd = std::chrono::steady_clock::now();
    // one round takes whole submissions up to MERGE_MAX_FRAMES frames (always the first one): what is left waits for the next leader, so
    // the slab a round needs is bounded by what its callers asked for, not by how many callers happened to arrive together
    std::vector<Submission *> take;
    size_t total = 0;
    {
        size_t k = 0;
        while (k < g_pending.size() && (k == 0 || total + (size_t)g_pending[k]->n <= (size_t)MERGE_MAX_FRAMES)) { total += (size_t)g_pending[k]->n; k++; }
        take.assign(g_pending.begin(), g_pending.begin() + (long)k);
        g_pending.erase(g_pending.begin(), g_pending.begin() + (long)k);
    }
    const batch_backend_t backend = g_backend;
    lk.unlock();
    try {                                               // (whatever happens in here, every caller of this round is released with a return code)
        std::vector<unsigned char *> pb(total), rc_(total); std::vector<const unsigned char *> im(total);
        std::vector<int> ys(total), xs(total), q(total), len(total, 0);
        size_t k = 0;
        for (const Submission *u : take) for (int i = 0; i < u->n; i++, k++) { pb[k] = u->pbuffers[i]; im[k] = u->imgs[i]; rc_[k] = u->rcons[i]; ys[k] = u->ysz[i]; xs[k] = u->xsz[i]; q[k] = u->qpd6[i]; }
        const int rc = total ? backend((int)total, pb.data(), im.data(), rc_.data(), ys.data(), xs.data(), q.data(), len.data()) : 0;
        k = 0;
        for (Submission *u : take) { for (int i = 0; i < u->n; i++, k++) if (rc == 0) { u->ysz[i] = ys[k]; u->xsz[i] = xs[k]; u->out_len[i] = len[k]; } u->rc = rc; }
        // a merged batch failed on its ARGUMENTS: every submission again on its own, so that only the caller whose frames cannot be encoded sees the
        // error.  Device-wide failures (HIP error, watchdog, no device) go to every caller of the round as they are: relaunching on a broken device
        // once per caller would multiply everybody's wait.
        if (rc == IMCVT_ERR_ARG && take.size() > 1) {
            for (Submission *u : take) {
                std::vector<int> y1(u->ysz, u->ysz + u->n), x1(u->xsz, u->xsz + u->n), l1((size_t)u->n, 0);
                u->rc = u->n ? backend(u->n, u->pbuffers, u->imgs, u->rcons, y1.data(), x1.data(), u->qpd6, l1.data()) : 0;
                if (u->rc == 0) for (int i = 0; i < u->n; i++) { u->ysz[i] = y1[(size_t)i]; u->xsz[i] = x1[(size_t)i]; u->out_len[i] = l1[(size_t)i]; }
            }
        }
    } catch (...) {
        fprintf(stderr, "imcvt_hevc: out of host memory while merging %zu frames of %zu callers\n", total, take.size());
        for (Submission *u : take) u->rc = IMCVT_ERR_HIP;
    }
    lk.lock();
    g_q_batches++; if ((long)total > g_q_max_batch) g_q_max_batch = (long)total;
    for (Submission *u : take) u->done = true;
}
extern "C" int HEVCImageEncoderBatch(int n, unsigned char *const *pbuffers, const unsigned char *const *imgs,
                                     unsigned char *const *rcons, int *ysz, int *xsz, const int *qpd6, int *out_len) {
    if (n < 0 || (n > 0 && (!pbuffers || !imgs || !rcons || !ysz || !xsz || !qpd6 || !out_len))) return IMCVT_ERR_ARG;
    for (int i = 0; i < n; i++) if (qpd6[i] < 0 || qpd6[i] > 4 || ysz[i] < 1 || xsz[i] < 1 || !pbuffers[i] || !imgs[i] || !rcons[i]) return IMCVT_ERR_ARG;
    Submission me; me.n = n; me.pbuffers = pbuffers; me.imgs = imgs; me.rcons = rcons; me.ysz = ysz; me.xsz = xsz; me.qpd6 = qpd6; me.out_len = out_len;
    std::unique_lock<std::mutex> lk(g_qmu);
    if (g_backend == encode_batch_on_devices && n > 0) { lk.unlock(); if (!have_device()) return IMCVT_ERR_NO_DEVICE; lk.lock(); }
    g_q_calls++;
    g_pending.push_back(&me);
    while (!me.done) {
        if (!g_leader) {
            g_leader = true;
            lead_one_round(lk);
            g_leader = false;
            g_qcv.notify_all();                         // results are out; whoever still waits takes the next round
        } else g_qcv.wait(lk);
    }
    return me.rc;
}

extern "C" int HEVCImageEncoder(unsigned char *pbuffer, const unsigned char *img, unsigned char *img_rcon,
                                int *ysz, int *xsz, const int qpd6) {
    if (!pbuffer || !img || !img_rcon || !ysz || !xsz) return IMCVT_ERR_ARG;
    int len = 0;
    unsigned char *pb[1] = { pbuffer }; const unsigned char *im[1] = { img }; unsigned char *rc_[1] = { img_rcon };
    int q[1] = { qpd6 };
    const int rc = HEVCImageEncoderBatch(1, pb, im, rc_, ysz, xsz, q, &len);
    return rc < 0 ? rc : len;
}

extern "C" int writeHEVCImageFile(const char *p_filename, const uint8_t *p_buf, int is_rgb, uint32_t height, uint32_t width, int qpd6) {
    const size_t npx = (size_t)height * width;
    const size_t n_out = (size_t)imcvt_hevc_stream_bound((int)height, (int)width), n_img = (size_t)(width + 32) * (height + 32) + 1048576;
    unsigned char *slab = (unsigned char *)malloc(n_out + 2 * n_img);
    if (!slab) return 1;
    unsigned char *orig = slab + n_out, *rcon = orig + n_img;
    if (is_rgb) {
        printf("   warning: this HEVCencoder currently only support gray 8-bit image instead of RGB image. Only compress the green channel of this image.\n");
        for (size_t i = 0; i < npx; i++) orig[i] = p_buf[i * 3 + 1];
    } else memcpy(orig, p_buf, npx);
    int h = (int)height, w = (int)width, failed = 1;
    const int len = HEVCImageEncoder(slab, orig, rcon, &h, &w, qpd6);
    if (len > 0 && h > 0 && w > 0) {
        FILE *fp = fopen(p_filename, "wb");
        if (fp) { failed = ((size_t)len != fwrite(slab, 1, (size_t)len, fp)); fclose(fp); }
    }
    free(slab);
    return failed;
}
