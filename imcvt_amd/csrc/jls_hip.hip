// jls_hip.hip — gfx950 kernel and C-ABI host shim of libimcvt_jls.so (include/imcvt_jls.h).
// One wavefront per plane (jls_core.h says why); persistent workgroups pull planes from an atomic counter.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <vector>

#include "jls_core.h"
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
__global__ __launch_bounds__(JLS_THREADS) void jls_encode_planes(const PlaneJob *jobs, int njobs, int *counter) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    __shared__ int next;
    jls::CtxMem cx = (jls::CtxMem)(JLS_LDS uint8_t *)lds;              // 364 contexts, 16 bytes each
    const int tid = (int)threadIdx.x;
    for (;;) {
        if (tid == 0) next = atomicAdd(counter, 1);
        __syncthreads();
        const int j = next;
        __syncthreads();
        if (j >= njobs) break;
        const PlaneJob job = jobs[j];
        const int w = job.w, rs = (w + 1 + 15) & ~15;                  // row buffers: w + 1 samples, 16-byte multiples
        jls::RowMem src = (jls::RowMem)lds + 364 * sizeof(jls::Ctx), rec = src + rs, prev = rec + rs;
        jls::Plane S;
        int hdr = 0;
        if (tid == 0) {
            if (job.framing) { hdr = jls::frame_header(job.out, 1, job.h, w); hdr = jls::scan_header(job.out, hdr, 1, job.near); }
            jls::plane_begin(S, cx, w, job.near, job.out + hdr);
        }
        for (int y = 0; y < job.h; y++) {
            { jls::RowMem t = prev; prev = rec; rec = t; }              // last row's reconstruction becomes the row above
            const JLS_GLB uint8_t *g = (const JLS_GLB uint8_t *)job.src + (size_t)y * w * job.stride;
            for (int x = tid; x < w; x += JLS_THREADS) src[x] = g[(size_t)x * job.stride];      // 64 consecutive samples per instruction
            __syncthreads();
            if (tid == 0) jls::plane_row(S, cx, y, src, rec, prev);
            __syncthreads();
        }
        if (tid == 0) {
            long long n = hdr + jls::plane_end(S);
            if (job.framing) n = jls::put_be(job.out, (int)n, 0xFFD9u, 2);
            *job.len = n;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "imcvt_jls: %s failed: %s\n", #x, hipGetErrorString(e_)); return IMCVT_JLS_ERR_HIP; } } while (0)

namespace {
struct State {
    std::mutex mu;
    bool ready = false;
    int cus = 0;
    int *d_counter = nullptr;
    PlaneJob *d_jobs = nullptr, *h_jobs = nullptr; int cap = 0;
    hipEvent_t e0 = nullptr, e1 = nullptr; bool timed = false; hipStream_t last_stream = nullptr;
    size_t lds_set = 0;
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
    G.cus = prop.multiProcessorCount;
    HIPCHK(hipMalloc(&G.d_counter, sizeof(int)));
    HIPCHK(hipEventCreate(&G.e0)); HIPCHK(hipEventCreate(&G.e1));
    G.ready = true;
    return 0;
}
size_t lds_bytes(int w) { return 364 * sizeof(jls::Ctx) + 3 * (size_t)((w + 1 + 15) & ~15); }

// launch n plane jobs (host array) on `stream`
int launch_locked(int n, const PlaneJob *jobs, hipStream_t stream) {
    if (n > G.cap) {
        if (G.d_jobs) hipFree(G.d_jobs);
        if (G.h_jobs) hipHostFree(G.h_jobs);
        G.d_jobs = nullptr; G.h_jobs = nullptr; G.cap = 0;
        HIPCHK(hipMalloc(&G.d_jobs, sizeof(PlaneJob) * n));
        HIPCHK(hipHostMalloc(&G.h_jobs, sizeof(PlaneJob) * n));
        G.cap = n;
    } else HIPCHK(hipStreamSynchronize(stream));
    int wmax = 1;
    for (int i = 0; i < n; i++) { G.h_jobs[i] = jobs[i]; if (jobs[i].w > wmax) wmax = jobs[i].w; }
    const size_t lds = lds_bytes(wmax);
    if (lds > G.lds_set) { HIPCHK(hipFuncSetAttribute((const void *)jls_encode_planes, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); G.lds_set = lds; }
    HIPCHK(hipMemcpyAsync(G.d_jobs, G.h_jobs, sizeof(PlaneJob) * n, hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemsetAsync(G.d_counter, 0, sizeof(int), stream));
    // as many one-wave workgroups as the LDS of the chip admits (160 KB per CU), at most 32 per CU
    int per_cu = (int)((160 * 1024) / (lds + 64)); if (per_cu > 32) per_cu = 32; if (per_cu < 1) per_cu = 1;
    int grid = G.cus * per_cu; if (grid > n) grid = n;
    HIPCHK(hipEventRecord(G.e0, stream));
    hipLaunchKernelGGL(jls_encode_planes, dim3(grid), dim3(JLS_THREADS), lds, stream, G.d_jobs, n, G.d_counter);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(G.e1, stream));
    G.timed = true; G.last_stream = stream;
    return 0;
}
}  // namespace

extern "C" const char *imcvt_jls_version(void) { return "imcvt_jls gfx950 r1 (wave-per-plane)"; }
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
