// jls_hip.hip — gfx950 kernel and C-ABI host shim of libimcvt_jls.so (include/imcvt_jls.h).
// One wavefront per plane (jls_core.h says why); persistent workgroups pull planes from an atomic counter.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <vector>

#include "jls_core.h"
#include "jls_par.h"
#include "../../include/imcvt_jls.h"

struct PlaneJob {
    const uint8_t *src;      // first sample of the plane
    uint8_t *out;            // where this job's bytes go
    long long *len;          // bytes written
    int stride;              // bytes between samples (1 gray, 3 interleaved RGB)
    int h, w, near;
    int framing;             // 1: complete gray file (SOI, SOF, SOS, scan, EOI); 0: the scan's bytes only
};

#define JLS_THREADS 64
#define JLS_MAX_WALKERS 8
// One wavefront walks up to L planes at once, one per lane ("walkers"): a walker's chain is a dependent scalar program, so
// a wave with a single walker occupies its SIMD's issue slots for one lane's worth of work.  With L walkers per wave the
// same instruction stream advances L planes (lanes in different coding modes — run / run interruption / regular — take
// turns under the exec mask), and the workgroups that the LDS of a CU admits are spread one wave per SIMD.
__global__ __launch_bounds__(JLS_THREADS) void jls_encode_planes(const PlaneJob *jobs, int njobs, int *counter, int L, int wmax, int rows3) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    __shared__ int next;
    const int tid = (int)threadIdx.x;
    const int rs = (wmax + 1 + 15) & ~15;                               // row buffers: w + 1 samples, 16-byte multiples
    // one walker: contexts + two row buffers (lossless: a row is its own reconstruction, the buffers alternate between "this row"
    // and "the row above"); near-lossless launches carry a third one for the source row
    const int area = 364 * (int)sizeof(jls::PCtx) + (rows3 ? 3 : 2) * rs;
    for (;;) {
        if (tid == 0) next = atomicAdd(counter, L);
        __syncthreads();
        const int j0 = next;
        __syncthreads();
        if (j0 >= njobs) break;
        const int nl = (njobs - j0 < L) ? njobs - j0 : L;
        const int mine = tid < nl;
        const PlaneJob job = jobs[j0 + (mine ? tid : 0)];
        JLS_LDS uint8_t *my = (JLS_LDS uint8_t *)lds + (mine ? tid : 0) * area;
        jls::CtxMem cx = (jls::CtxMem)my;
        jls::RowMem buf0 = (jls::RowMem)my + 364 * sizeof(jls::PCtx), buf1 = buf0 + rs, srcrow = buf1 + rs;
        int hmax = 0;
        for (int i = 0; i < nl; i++) { const int hi = jobs[j0 + i].h; hmax = hi > hmax ? hi : hmax; }
        jls::Plane S;
        int hdr = 0;
        if (mine) {
            if (job.framing) { hdr = jls::frame_header(job.out, 1, job.h, job.w); hdr = jls::scan_header(job.out, hdr, 1, job.near); }
            jls::plane_begin(S, cx, job.w, job.near, job.out + hdr);
        }
        for (int y = 0; y < hmax; y++) {
            for (int i = 0; i < nl; i++) {                              // all 64 lanes stream row y of every walker's plane into its LDS area
                const PlaneJob ji = jobs[j0 + i];
                if (y < ji.h) {
                    jls::RowMem si = (jls::RowMem)lds + i * area + 364 * sizeof(jls::PCtx) + (rows3 ? 2 * rs : (y & 1) * rs);
                    const JLS_GLB uint8_t *g = (const JLS_GLB uint8_t *)ji.src + (size_t)y * ji.w * ji.stride;
                    for (int x = tid; x < ji.w; x += JLS_THREADS) si[x] = g[(size_t)x * ji.stride];     // 64 consecutive samples per instruction
                }
            }
            __syncthreads();
            if (mine && y < job.h) {                                    // last row's reconstruction is the row above
                jls::RowMem rec = (y & 1) ? buf1 : buf0, prev = (y & 1) ? buf0 : buf1;
                jls::plane_row(S, cx, y, rows3 ? srcrow : rec, rec, prev);
            }
            __syncthreads();
        }
        if (mine) {
            long long n = hdr + jls::plane_end(S);
            if (job.framing) n = jls::put_be(job.out, (int)n, 0xFFD9u, 2);
            *job.len = n;
        }
        __syncthreads();
    }
}

// ---- one lossless plane over the whole GPU (jls_par.h): one kernel per step, blockIdx.y = plane -----------------------
#define PAR_T (long)blockIdx.x * blockDim.x + threadIdx.x
__global__ void jls_par_k1(const jls::ParPlane *pl) { const jls::ParPlane P = pl[blockIdx.y]; const long t = PAR_T; if (t < (long)P.h * P.w) jls::k1_classify(P, t); }
#define K2_ROWS 32
__global__ __launch_bounds__(K2_ROWS) void jls_par_k2(const jls::ParPlane *pl) {        // the (row, context) cell counters of the block's rows live in LDS
    __shared__ uint32_t cnt[K2_ROWS][365];                                            // (odd stride: the rows' counters fall into different banks)
    const jls::ParPlane P = pl[blockIdx.y];
    const long row = PAR_T;
    if (row >= P.h) return;
    jls::k2_rows(P, row, (JLS_LDS uint32_t *)cnt[threadIdx.x]);
    for (int i = 0; i < 364; i++) P.rowcnt[(size_t)row * 364 + i] = cnt[threadIdx.x][i];
}
__global__ void jls_par_k3(const jls::ParPlane *pl) { const jls::ParPlane P = pl[blockIdx.y]; const long t = PAR_T; if (t < 365) jls::k3_cells(P, t); }
__global__ void jls_par_k3b(const jls::ParPlane *pl) { if (threadIdx.x == 0) jls::k3_bases(pl[blockIdx.y]); }
__global__ void jls_par_k4(const jls::ParPlane *pl) { const jls::ParPlane P = pl[blockIdx.y]; const long t = PAR_T; if (t < (long)P.h * P.w) jls::k4_scatter(P, t); }
// one chain per wavefront: a chain is a dependent scalar program.  The 364 regular chains are run by ALL lanes on wave-uniform
// values (scalar-unit code, every lane stores the same code word to the same address); the run chain by lane 0.
#ifdef JLS_CHAIN_VECTOR
#define K5_ALL_LANES 0
#else
#define K5_ALL_LANES (blockIdx.x < 364)
#endif
__global__ __launch_bounds__(64) void jls_par_k5(const jls::ParPlane *pl) { if (K5_ALL_LANES || threadIdx.x == 0) jls::k5_chain(pl[blockIdx.y], (long)blockIdx.x); }
__global__ void jls_par_k6(const jls::ParPlane *pl) { const jls::ParPlane P = pl[blockIdx.y]; const long t = PAR_T; if (t < (long)P.h * P.w) jls::k6_len(P, t); }
__global__ void jls_par_k6b(const jls::ParPlane *pl) { const jls::ParPlane P = pl[blockIdx.y]; const long t = PAR_T; if (t < P.h) jls::k6_rowscan(P, t); }
__global__ void jls_par_k6c(const jls::ParPlane *pl) { if (threadIdx.x == 0) jls::k6_rows(pl[blockIdx.y]); }
__global__ void jls_par_k7(const jls::ParPlane *pl) { const jls::ParPlane P = pl[blockIdx.y]; const long t = PAR_T; if (t < (long)P.h * P.w) jls::k7_pack(P, t); }
__global__ void jls_par_k8a(const jls::ParPlane *pl) { const jls::ParPlane P = pl[blockIdx.y]; const long t = PAR_T; if (t < 16 * (long)jls::par_chunks_max((size_t)P.h * P.w)) jls::k8_simulate(P, t); }
__global__ void jls_par_k8b(const jls::ParPlane *pl, const PlaneJob *jobs) {          // the real entry state of every chunk; the stream's length and framing
    if (threadIdx.x != 0) return;
    const jls::ParPlane P = pl[blockIdx.y];
    jls::k8_chain(P, (long)jls::par_chunks_max((size_t)P.h * P.w));
    long long n = P.hdr + (long long)P.total[1];
    if (jobs[blockIdx.y].framing) n = jls::put_be(P.out, (int)n, 0xFFD9u, 2);
    *jobs[blockIdx.y].len = n;
}
__global__ void jls_par_k8c(const jls::ParPlane *pl) { const jls::ParPlane P = pl[blockIdx.y]; const long t = PAR_T; const long cm = (long)jls::par_chunks_max((size_t)P.h * P.w); if (t < cm) jls::k8_write(P, t, cm); }
__global__ void jls_par_hdr(const jls::ParPlane *pl, const PlaneJob *jobs) {          // gray file framing in front of the scan (:206-237)
    if (threadIdx.x != 0 || !jobs[blockIdx.y].framing) return;
    const PlaneJob j = jobs[blockIdx.y];
    const int at = jls::frame_header(j.out, 1, j.h, j.w);
    jls::scan_header(j.out, at, 1, 0);
}

// ---------------------------------------------------------------------------------------------------
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "imcvt_jls: %s failed: %s\n", #x, hipGetErrorString(e_)); return IMCVT_JLS_ERR_HIP; } } while (0)

namespace {
struct State {
    std::mutex mu;
    bool ready = false;
    int cus = 0, device = 0;
    int *d_counter = nullptr;
    PlaneJob *d_jobs = nullptr, *h_jobs = nullptr; int cap = 0;
    hipEvent_t e0 = nullptr, e1 = nullptr; bool timed = false; hipStream_t last_stream = nullptr;
    size_t lds_set = 0;
    uint8_t *d_work = nullptr; size_t work_cap = 0;      // work arrays of the plane-parallel path (grow-only)
    jls::ParPlane *d_par = nullptr, *h_par = nullptr; int par_cap = 0;
    int last_par = 0;
} G;

bool have_device() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { fprintf(stderr, "imcvt_jls: no HIP device visible — this library has no CPU fallback\n"); return false; }
    return true;
}
int init_locked() {
    if (G.ready) return 0;
    if (!have_device()) return IMCVT_JLS_ERR_NO_DEVICE;
    hipDeviceProp_t prop; int dev = 0;
    HIPCHK(hipGetDevice(&dev)); HIPCHK(hipGetDeviceProperties(&prop, dev));
    G.cus = prop.multiProcessorCount; G.device = dev;
    HIPCHK(hipMalloc(&G.d_counter, sizeof(int)));
    HIPCHK(hipEventCreate(&G.e0)); HIPCHK(hipEventCreate(&G.e1));
    G.ready = true;
    return 0;
}
size_t walker_bytes(int w, int rows3) { return 364 * sizeof(jls::PCtx) + (rows3 ? 3 : 2) * (size_t)((w + 1 + 15) & ~15); }

// launch n plane jobs (host array) on `stream`
int launch_par_locked(int n, const PlaneJob *jobs, hipStream_t stream);
bool use_par(int n, const PlaneJob *jobs);
int launch_locked(int n, const PlaneJob *jobs, hipStream_t stream) {
    // one launch in flight: the job table, the plane counter and the work arrays belong to the running launch, whatever stream it is on
    HIPCHK(hipSetDevice(G.device));
    if (G.timed) HIPCHK(hipEventSynchronize(G.e1));
    if (use_par(n, jobs)) return launch_par_locked(n, jobs, stream);
    G.last_par = 0;
    if (n > G.cap) {
        if (G.d_jobs) hipFree(G.d_jobs);
        if (G.h_jobs) hipHostFree(G.h_jobs);
        G.d_jobs = nullptr; G.h_jobs = nullptr; G.cap = 0;
        HIPCHK(hipMalloc(&G.d_jobs, sizeof(PlaneJob) * n));
        HIPCHK(hipHostMalloc(&G.h_jobs, sizeof(PlaneJob) * n));
        G.cap = n;
    }
    int wmax = 1, rows3 = 0;
    for (int i = 0; i < n; i++) { G.h_jobs[i] = jobs[i]; if (jobs[i].w > wmax) wmax = jobs[i].w; if (jobs[i].near > 0) rows3 = 1; }
    // walkers per wave: measured on 1024 .. 8192 planes of 1080p, two per wave is the optimum (4.3 Gpx/s; one: 3.0, three: 2.9 —
    // more walkers per wave serialise on coding-mode divergence and on the per-row barrier, fewer leave the waves VALU-issue bound)
    const size_t area = walker_bytes(wmax, rows3);
    int L = (n >= 2 * G.cus) ? 2 : 1;
    if (const char *e = getenv("IMCVT_JLS_WALKERS")) { const int v = atoi(e); if (v >= 1 && v <= JLS_MAX_WALKERS) L = v; }
    if ((size_t)L * area > 160 * 1024 - 1024) L = (int)((160 * 1024 - 1024) / area) > 0 ? (int)((160 * 1024 - 1024) / area) : 1;
    const size_t lds = (size_t)L * area;
    if (lds > G.lds_set) { HIPCHK(hipFuncSetAttribute((const void *)jls_encode_planes, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); G.lds_set = lds; }
    HIPCHK(hipMemcpyAsync(G.d_jobs, G.h_jobs, sizeof(PlaneJob) * n, hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemsetAsync(G.d_counter, 0, sizeof(int), stream));
    int wg_per_cu = (int)((160 * 1024) / (lds + 64)); if (wg_per_cu > 32) wg_per_cu = 32; if (wg_per_cu < 1) wg_per_cu = 1;
    int grid = G.cus * wg_per_cu; const int groups = (n + L - 1) / L; if (grid > groups) grid = groups;
    HIPCHK(hipEventRecord(G.e0, stream));
    hipLaunchKernelGGL(jls_encode_planes, dim3(grid), dim3(JLS_THREADS), lds, stream, G.d_jobs, n, G.d_counter, L, wmax, rows3);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(G.e1, stream));
    G.timed = true; G.last_stream = stream;
    return 0;
}

// ---- the plane-parallel path: lossless planes, few of them (jls_par.h) ----
#define PAR_GROUP 16          // planes per pass (their work arrays are allocated together)
bool use_par(int n, const PlaneJob *jobs) {
    if (const char *e = getenv("IMCVT_JLS_PAR")) return atoi(e) != 0 && [&] { for (int i = 0; i < n; i++) if (jobs[i].near) return false; return true; }();
    if (n > 64) return false;                      // many planes: one walker per plane fills the device better (4.3 Gpx/s at 4096 planes)
    for (int i = 0; i < n; i++) if (jobs[i].near) return false;
    return true;
}
int launch_par_locked(int n, const PlaneJob *jobs, hipStream_t stream) {
    if (n > G.cap) {
        if (G.d_jobs) hipFree(G.d_jobs);
        if (G.h_jobs) hipHostFree(G.h_jobs);
        G.d_jobs = nullptr; G.h_jobs = nullptr; G.cap = 0;
        HIPCHK(hipMalloc(&G.d_jobs, sizeof(PlaneJob) * n));
        HIPCHK(hipHostMalloc(&G.h_jobs, sizeof(PlaneJob) * n));
        G.cap = n;
    }
    if (PAR_GROUP > G.par_cap) {
        HIPCHK(hipMalloc(&G.d_par, sizeof(jls::ParPlane) * PAR_GROUP));
        HIPCHK(hipHostMalloc(&G.h_par, sizeof(jls::ParPlane) * PAR_GROUP));
        G.par_cap = PAR_GROUP;
    }
    HIPCHK(hipEventRecord(G.e0, stream));
    for (int g0 = 0; g0 < n; g0 += PAR_GROUP) {
        const int m = (n - g0 < PAR_GROUP) ? n - g0 : PAR_GROUP;
        size_t need = 0; long npx_max = 1, cm_max = 1; int h_max = 1;
        for (int i = 0; i < m; i++) need += jls::par_workspace(jobs[g0 + i].h, jobs[g0 + i].w);
        if (need > G.work_cap) {
            HIPCHK(hipStreamSynchronize(stream));
            if (G.d_work) hipFree(G.d_work);
            G.d_work = nullptr; G.work_cap = 0;
            HIPCHK(hipMalloc(&G.d_work, need));
            G.work_cap = need;
        }
        if (g0) HIPCHK(hipStreamSynchronize(stream));    // the pinned plane table is rewritten per group
        size_t off = 0;
        for (int i = 0; i < m; i++) {
            const PlaneJob &j = jobs[g0 + i];
            G.h_jobs[g0 + i] = j;
            jls::ParPlane &P = G.h_par[i];
            P.src = j.src; P.stride = j.stride; P.h = j.h; P.w = j.w; P.out = j.out;
            P.hdr = j.framing ? jls::FRAME_HDR_GRAY + jls::SCAN_HDR : 0;
            jls::par_carve(P, G.d_work + off);
            off += jls::par_workspace(j.h, j.w);
            const long npx = (long)j.h * j.w;
            if (npx > npx_max) npx_max = npx;
            if (j.h > h_max) h_max = j.h;
            const long cm = (long)jls::par_chunks_max((size_t)npx); if (cm > cm_max) cm_max = cm;
            HIPCHK(hipMemsetAsync(P.bits, 0, jls::par_bits_bytes(j.h, j.w), stream));
        }
        HIPCHK(hipMemcpyAsync(G.d_par, G.h_par, sizeof(jls::ParPlane) * m, hipMemcpyHostToDevice, stream));
        HIPCHK(hipMemcpyAsync(G.d_jobs + g0, G.h_jobs + g0, sizeof(PlaneJob) * m, hipMemcpyHostToDevice, stream));
        const dim3 gpx((unsigned)((npx_max + 255) / 256), (unsigned)m), one(1, (unsigned)m);
        const PlaneJob *dj = G.d_jobs + g0;
        hipLaunchKernelGGL(jls_par_hdr, one, dim3(64), 0, stream, G.d_par, dj);
        hipLaunchKernelGGL(jls_par_k1, gpx, dim3(256), 0, stream, G.d_par);
        hipLaunchKernelGGL(jls_par_k2, dim3((unsigned)((h_max + K2_ROWS - 1) / K2_ROWS), (unsigned)m), dim3(K2_ROWS), 0, stream, G.d_par);
        hipLaunchKernelGGL(jls_par_k3, dim3(6, (unsigned)m), dim3(64), 0, stream, G.d_par);
        hipLaunchKernelGGL(jls_par_k3b, one, dim3(64), 0, stream, G.d_par);
        hipLaunchKernelGGL(jls_par_k4, gpx, dim3(256), 0, stream, G.d_par);
        hipLaunchKernelGGL(jls_par_k5, dim3(365, (unsigned)m), dim3(64), 0, stream, G.d_par);
        hipLaunchKernelGGL(jls_par_k6, gpx, dim3(256), 0, stream, G.d_par);
        hipLaunchKernelGGL(jls_par_k6b, dim3((unsigned)((h_max + 63) / 64), (unsigned)m), dim3(64), 0, stream, G.d_par);
        hipLaunchKernelGGL(jls_par_k6c, one, dim3(64), 0, stream, G.d_par);
        hipLaunchKernelGGL(jls_par_k7, gpx, dim3(256), 0, stream, G.d_par);
        hipLaunchKernelGGL(jls_par_k8a, dim3((unsigned)((16 * cm_max + 255) / 256), (unsigned)m), dim3(256), 0, stream, G.d_par);
        hipLaunchKernelGGL(jls_par_k8b, one, dim3(64), 0, stream, G.d_par, dj);
        hipLaunchKernelGGL(jls_par_k8c, dim3((unsigned)((cm_max + 63) / 64), (unsigned)m), dim3(64), 0, stream, G.d_par);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(G.e1, stream));
    G.timed = true; G.last_stream = stream; G.last_par = 1;
    return 0;
}
}  // namespace

extern "C" const char *imcvt_jls_version(void) { return "imcvt_jls gfx950 r2 (lossless planes in small batches: context chains over the whole device; else wave-per-plane walkers)"; }
extern "C" int imcvt_jls_last_path(void) { return G.last_par; }
extern "C" long long imcvt_jls_stream_bound(int h, int w) { return 8LL * w * h + 65536; }

extern "C" int imcvt_jls_encode_device(int n, const imcvt_jls_plane *planes, void *stream_) {
    if (n < 0 || (n > 0 && !planes)) return IMCVT_JLS_ERR_ARG;
    if (n == 0) return 0;
    std::lock_guard<std::mutex> lk(G.mu);
    int rc = init_locked(); if (rc) return rc;
    std::vector<PlaneJob> jobs(n);
    for (int i = 0; i < n; i++) {
        const imcvt_jls_plane &p = planes[i];
        if (!p.d_img || !p.d_out || !p.d_len || p.h < 1 || p.w < 1 || p.h > 32767 || p.w > 32767 || p.near < 0 || p.near > 255) return IMCVT_JLS_ERR_ARG;
        jobs[i] = PlaneJob{p.d_img, p.d_out, p.d_len, 1, p.h, p.w, p.near, 1};
    }
    return launch_locked(n, jobs.data(), (hipStream_t)stream_);
}
extern "C" float imcvt_jls_last_kernel_ms(void) {
    std::lock_guard<std::mutex> lk(G.mu);
    if (!G.timed) return -1.f;
    float ms = -1.f;
    if (hipEventSynchronize(G.e1) != hipSuccess || hipEventElapsedTime(&ms, G.e0, G.e1) != hipSuccess) return -1.f;
    return ms;
}

extern "C" long long imcvt_jls_encode(const uint8_t *img, int is_rgb, int h, int w, int near, uint8_t *out) {
    if (!img || !out || h < 1 || w < 1 || h > 32767 || w > 32767 || near < 0 || near > 255) return IMCVT_JLS_ERR_ARG;
    std::lock_guard<std::mutex> lk(G.mu);
    int rc = init_locked(); if (rc) return rc;
    const int planes = is_rgb ? 3 : 1;
    const size_t npx = (size_t)h * w, in_bytes = npx * planes;
    const size_t per_plane = (size_t)imcvt_jls_stream_bound(h, w);
    uint8_t *d_img = nullptr, *d_out = nullptr; long long *d_len = nullptr;
    long long total = IMCVT_JLS_ERR_HIP;
    std::vector<long long> lens(planes, 0);
    do {
        if (hipMalloc(&d_img, in_bytes) != hipSuccess || hipMalloc(&d_out, per_plane * planes) != hipSuccess || hipMalloc(&d_len, sizeof(long long) * planes) != hipSuccess) break;
        if (hipMemcpy(d_img, img, in_bytes, hipMemcpyHostToDevice) != hipSuccess) break;
        PlaneJob jobs[3];
        for (int c = 0; c < planes; c++) jobs[c] = PlaneJob{d_img + c, d_out + per_plane * c, d_len + c, planes, h, w, near, is_rgb ? 0 : 1};
        if (launch_locked(planes, jobs, nullptr) != 0) break;
        if (hipDeviceSynchronize() != hipSuccess) break;
        if (hipMemcpy(lens.data(), d_len, sizeof(long long) * planes, hipMemcpyDeviceToHost) != hipSuccess) break;
        if (!is_rgb) {
            if (hipMemcpy(out, d_out, (size_t)lens[0], hipMemcpyDeviceToHost) != hipSuccess) break;
            total = lens[0];
        } else {                                         // the three colour scans were coded concurrently; frame them in order (:415-425)
            int at = jls::frame_header(out, 3, h, w);
            bool ok = true;
            for (int c = 0; c < 3 && ok; c++) {
                at = jls::scan_header(out, at, c + 1, near);
                ok = hipMemcpy(out + at, d_out + per_plane * c, (size_t)lens[c], hipMemcpyDeviceToHost) == hipSuccess;
                at += (int)lens[c];
            }
            if (!ok) break;
            at = jls::put_be(out, at, 0xFFD9u, 2);
            total = at;
        }
    } while (0);
    hipFree(d_img); hipFree(d_out); hipFree(d_len);
    if (total < 0) fprintf(stderr, "imcvt_jls: device encode failed: %s\n", hipGetErrorString(hipGetLastError()));
    return total;
}

extern "C" int writeJLSImageFile(const char *p_filename, const uint8_t *p_buf, int is_rgb, uint32_t height, uint32_t width, int near) {
    if (width < 1 || width > 32767 || height < 1 || height > 32767) return 1;             // :437
    uint8_t *buf = (uint8_t *)malloc((size_t)imcvt_jls_stream_bound((int)height, (int)width) * (is_rgb ? 3 : 1));
    if (!buf) return 1;
    const long long n = imcvt_jls_encode(p_buf, is_rgb, (int)height, (int)width, near, buf);
    int failed = 1;
    if (n > 0) {
        FILE *fp = fopen(p_filename, "wb");
        if (fp) { failed = fwrite(buf, 1, (size_t)n, fp) != (size_t)n; fclose(fp); }
    }
    free(buf);
    return failed;
}
