/*
 * imcvt_hevc.h — C ABI of libimcvt_hevc.so, the MI355X (gfx950) implementation of ImCvt's H.265 intra
 * encode hot path.  Plain pointers and sizes only; no C++ or torch types.
 *
 * Section 1 is the reference's own interface for this path, symbol for symbol, so the reference's
 * main.c / imageio_hevc.c link against this library unchanged.
 * Section 2 is the batch / device-resident surface the reference does not have (its file loop,
 * src/main.c:162, is the seam it plugs into).
 *
 * Every entry point needs a visible gfx950 device.  There is no CPU fallback: without a device the
 * encoders return IMCVT_ERR_NO_DEVICE (<0) after printing one line to stderr.
 */
#ifndef IMCVT_HEVC_H
#define IMCVT_HEVC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IMCVT_ERR_NO_DEVICE  (-1)
#define IMCVT_ERR_HIP        (-2)
#define IMCVT_ERR_ARG        (-3)
#define IMCVT_ERR_WATCHDOG   (-4)   /* a wait between cooperating workgroups on the device gave up: the launch was abandoned */

/* ---------------------------------------------------------------------------------------------------
 * 1. Drop-in replacements
 * --------------------------------------------------------------------------------------------------- */

/* Replaces HEVCImageEncoder — reference src/HEVCe/HEVCe.h:5-12, defined src/HEVCe/HEVCe.c:1569-1646.
 * Same arguments, same ownership (caller owns every buffer, all HOST pointers), same results:
 *   pbuffer   out: the .h265 byte stream (caller sizes it as src/imageio_hevc.c:14 does)
 *   img       in : ysz*xsz gray8, tightly packed
 *   img_rcon  out: reconstruction, yszn*xszn bytes with the PADDED stride xszn
 *   ysz,xsz   in : true size; out: size padded to a multiple of 32 (capped at 8192, :1580-1581)
 *   qpd6      0..4 (not validated by the reference; this library returns IMCVT_ERR_ARG outside 0..4)
 * Returns the stream length in bytes (>0), or a negative IMCVT_ERR_* (the reference's caller treats
 * <=0 as failure, src/imageio_hevc.c:38).  Re-entrant like the reference's (which has no mutable globals): calls from several
 * threads at once are merged into one device batch (see HEVCImageEncoderBatch) instead of running one after the other. */
int HEVCImageEncoder(unsigned char *pbuffer, const unsigned char *img, unsigned char *img_rcon,
                     int *ysz, int *xsz, const int qpd6);

/* Replaces writeHEVCImageFile — reference src/imageio.h:21, defined src/imageio_hevc.c:9-53.
 * RGB input: prints the reference's warning and encodes the green channel (:21-27).
 * Returns 0 on success, 1 on failure (allocation, encoder, fopen, short write). */
int writeHEVCImageFile(const char *p_filename, const uint8_t *p_buf, int is_rgb,
                       uint32_t height, uint32_t width, int qpd6);

/* ---------------------------------------------------------------------------------------------------
 * 2. Batch surface (new; per-frame semantics identical to HEVCImageEncoder)
 * --------------------------------------------------------------------------------------------------- */

/* n independent frames with HOST pointers — the seam is the reference's serial file loop, src/main.c:162-211.
 * Frames fan out over every visible device (frame i -> device i mod D, D = min(n, devices); a single frame
 * stays on the caller's current device), are encoded concurrently and copied back; results do not depend on
 * D.  ysz[i]/xsz[i] are updated to the padded sizes, out_len[i] receives each stream length.  Device memory
 * and contexts are created on first use and reused by later calls (imcvt_hevc_shutdown releases them).
 * Environment: IMCVT_HEVC_DEVICES=k limits the fan-out to the first k devices; IMCVT_HEVC_FAKE_DEVICES=k (test seam) runs the
 * fan-out with k logical devices that all sit on physical device 0, each with its own context, stream and slab.
 * Concurrent callers: calls that arrive within a short window (IMCVT_HEVC_COALESCE_US, default 300 us) or while the previous
 * batch is running are merged into ONE device batch — the first caller waits the window (only when callers have been seen
 * arriving together in the last 100 ms: a lone caller is not delayed), takes everything submitted so far, runs it and hands
 * every caller its results; 16 threads with a frame each cost about one or two frames' time, not sixteen.
 * Returns 0, or a negative IMCVT_ERR_*.   */
int HEVCImageEncoderBatch(int n, unsigned char *const *pbuffers, const unsigned char *const *imgs,
                          unsigned char *const *rcons, int *ysz, int *xsz, const int *qpd6, int *out_len);

/* Number of devices the last HEVCImageEncoderBatch / HEVCImageEncoder call used. */
int imcvt_hevc_batch_devices(void);
/* How the last host-pointer batch moved its data (any pointer may be NULL): seconds spent uploading the inputs (large batches go through pinned
 * staging buffers filled by worker threads), seconds the calling thread followed the running launch (copying finished CTU rows of the
 * reconstructions and finished stream bytes to the caller's buffers while the kernel ran), seconds collecting what was left once the launch had
 * ended; bytes copied out while the launch ran / after it.  The reference's writeHEVCImageFile (src/imageio_hevc.c:14-52) has no counterpart:
 * its encoder works in the caller's memory. */
void imcvt_hevc_batch_transfer_stats(double *upload_s, double *follow_s, double *tail_s, double *bytes_during, double *bytes_after);
/* Kernel time of the last host-pointer batch in milliseconds (HIP events on the launch stream; the longest over its devices). */
double imcvt_hevc_batch_kernel_ms(void);
/* Releases the contexts, streams and device memory the host-pointer entry points hold (they are re-created on the next call). */
void imcvt_hevc_shutdown(void);

/* One frame of a device-resident batch.  All pointers are DEVICE pointers (hipMalloc / torch). */
typedef struct imcvt_hevc_frame {
    const unsigned char *d_img;    /* h*w gray8                                              */
    unsigned char       *d_out;    /* stream buffer, >= imcvt_hevc_stream_bound(h,w) bytes   */
    unsigned char       *d_rcon;   /* hp*wp reconstruction (padded stride)                   */
    int                 *d_len;    /* receives the stream length                             */
    int                  h, w;     /* true size                                              */
    int                  qpd6;     /* 0..4                                                   */
} imcvt_hevc_frame;

typedef struct imcvt_hevc_ctx imcvt_hevc_ctx;

/* Creates an encoder context on the current HIP device: uploads the constant tables and allocates the
 * per-workgroup scratch for up to max_workgroups concurrent frames (0 = 4 per compute unit).
 * Returns NULL when no device is present. */
imcvt_hevc_ctx *imcvt_hevc_create(int max_workgroups);
void            imcvt_hevc_destroy(imcvt_hevc_ctx *ctx);

/* Worst-case stream bytes for an h x w frame (the reference's own bound, src/imageio_hevc.c:14). */
long long imcvt_hevc_stream_bound(int h, int w);
/* Padded dimension ((min(v,8192)+31)/32*32, reference :1580-1581). */
int imcvt_hevc_padded(int v);

/* Encodes n device-resident frames on `stream` (a hipStream_t, may be NULL for the default stream).
 * Asynchronous: returns after the launch; results are valid once the stream has been synchronised.
 * ONE launch is in flight per context: a call first waits for the context's previous launch (whatever stream
 * that was on), because the job table, the frame queue and the per-workgroup scratch belong to the running
 * launch; use one context per stream for concurrent launches.  The call makes the context's device current.
 * Returns 0 or a negative IMCVT_ERR_*. */
int imcvt_hevc_encode_device(imcvt_hevc_ctx *ctx, int n, const imcvt_hevc_frame *frames, void *stream);

/* Progress records for the frames of the NEXT launches of this context: `words` points at 2 x n 32-bit words (pinned host memory or device
 * memory, zeroed by the caller), NULL turns the records off (the default).  While a launch runs the device keeps frame i's record current:
 * words[2 i] = CTU rows (32 picture rows each) whose reconstruction in d_rcon is final, with bit 31 set once the frame is finished;
 * words[2 i + 1] = leading bytes of d_out that are final.  Everything a record names has been written back from the device's caches, so a
 * copy engine may read it while the kernel is still running — this is how the host-pointer entry points overlap their device-to-host copies
 * with the launch.  Costs one cache write-back per CTU row of every frame. */
void imcvt_hevc_set_progress(imcvt_hevc_ctx *ctx, unsigned int *words);

/* Helper workgroups: 0 = chosen per launch (a frame per workgroup when the batch fills the device, else main
 * workgroups that walk the 8x8 CUs of their frames plus a pool of helper workgroups that evaluate the 16x16 / 32x32
 * candidate sets for all of them), 1 = none, 2 / 3 = one / two helpers per main workgroup.
 * Results are identical for every setting.  Environment override at context creation: IMCVT_HEVC_TEAM. */
void imcvt_hevc_set_team(imcvt_hevc_ctx *ctx, int team_size);
/* Pipe wave: workgroups of 256 threads whose fourth wavefront prices the NxN candidate of every 8x8 CU ahead of time (all 35
 * possible modes of the last PU), off the serial chain of the wave that walks the PUs — shorter frames when the device is not
 * full (three such workgroups fit a compute unit instead of four).  mode < 0 (default) / 1: used whenever the launch fits three
 * workgroups per compute unit; 0: never.  Results are identical.  Environment override at context creation: IMCVT_HEVC_PIPE. */
void imcvt_hevc_set_pipe(imcvt_hevc_ctx *ctx, int mode);
/* 1 if the last launch ran with the pipe wave, else 0. */
int imcvt_hevc_last_pipe(imcvt_hevc_ctx *ctx);
/* Wide workgroups (512 threads: the pipe wave and four partner wavefronts, each running the byte half of the trial coders whose
 * range half its owner wavefront runs, hevc_core.h stream_seg_R / stream_seg_L; one workgroup per compute unit).  mode < 0
 * (default): whenever a pipe-wave launch leaves every workgroup a compute unit of its own; 0: never; 1: as < 0.  Results are
 * identical.  Environment override at context creation: IMCVT_HEVC_WIDE. */
void imcvt_hevc_set_wide(imcvt_hevc_ctx *ctx, int mode);
/* 1 if the last launch ran wide workgroups, else 0. */
int imcvt_hevc_last_wide(imcvt_hevc_ctx *ctx);
/* Pure: does a launch of `grid` workgroups that runs with the pipe wave (use_pipe) run wide workgroups, given `wide_wg` resident
 * 512-thread workgroups (occupancy x compute units)?  A sixteenth of them stays free unless the shape is forced. */
int imcvt_hevc_plan_wide(int use_pipe, int grid, int wide_wg, int forced_shape);
/* Pure: the same for a launch shape (*nmains, *nhelp as imcvt_hevc_plan / imcvt_hevc_plan_pipe left them; mode = what imcvt_hevc_plan
 * returned): a pool that does not fit as planned still runs wide when its main workgroups take at most half of the `wide_wg`
 * workgroups — *nhelp is cut to the rest.  Returns 1 if the launch runs wide workgroups. */
int imcvt_hevc_plan_wide_pool(int use_pipe, int mode, int forced_shape, int wide_wg, const int *nmains, int *nhelp);
/* Partner workgroups (wide pools): every main workgroup gets a second compute unit that evaluates the two 2Nx2N candidate sets of its 8x8 CUs while it walks their
 * NxN chains alone (four wavefronts, a SIMD each): 64 frames 2.27 -> 2.09 s, one 1080p frame 2.20 -> 2.17 s.  mode < 0 (default) / 1: wherever
 * imcvt_hevc_plan_partners finds room; 0: never.  Results are identical.  Environment at context creation: IMCVT_HEVC_PARTNERS. */
void imcvt_hevc_set_partners(imcvt_hevc_ctx *ctx, int mode);
/* Partner workgroups of the last launch (0 or its main workgroups). */
int imcvt_hevc_last_partners(imcvt_hevc_ctx *ctx);
/* Pure: partner workgroups for a wide pool of nmains main and *nhelp helper workgroups on a device that holds wide_wg wide workgroups: nmains when they fit beside the
 * helpers (a sixteenth of the compute units stays free unless the shape is forced), or with *nhelp cut to the rest as long as 1.5 helpers per main workgroup remain; else 0. */
int imcvt_hevc_plan_partners(int nmains, int *nhelp, int wide_wg, int forced_shape);
/* A pool spread over two cooperating launches: wide main workgroups (512 threads, a compute unit each) on one set of compute units, 192-thread
 * helper workgroups, several per compute unit, on the others (streams with disjoint compute-unit masks) — for pools of 107 .. 128 main workgroups,
 * whose wide shape in ONE launch leaves every main workgroup fewer than 1.4 (wide) helpers.  mode < 0 (default) / 1: wherever
 * imcvt_hevc_plan_split says; 0: never; helpers_per_cu 0: default (3).  Results are identical.  Environment at context creation: IMCVT_HEVC_SPLIT, IMCVT_HEVC_SPLIT_HPC. */
void imcvt_hevc_set_split(imcvt_hevc_ctx *ctx, int mode, int helpers_per_cu);
/* 1 if the last launch was such a pair of launches, else 0. */
int imcvt_hevc_last_split(imcvt_hevc_ctx *ctx);
/* Pure: should a pool of nmains main workgroups (mode = what imcvt_hevc_plan returned) run as two launches on a device of `cus` compute units that
 * holds wide_wg wide workgroups and occ_wg 192-thread workgroups per compute unit?  Returns 1 and the helper count in *nhelp. */
int imcvt_hevc_plan_split(int mode, int nmains, int cus, int wide_wg, int occ_wg, int helpers_per_cu, int *nhelp);
/* That choice as a pure function (no device needed), applied to the shape imcvt_hevc_plan returned (mode = its return value; *nmains,
 * *nhelp = its outputs): returns 1 if the launch runs 256-thread workgroups with the pipe wave — it does when it fits 15/16 of three
 * workgroups per compute unit (max_workgroups * 3 / 4), and a pool that misses that by little gives up helpers for it (*nhelp is
 * reduced, never below 1.5 per main workgroup).  forced_shape: the shape was set by imcvt_hevc_set_shape (it may fill the last slot). */
int imcvt_hevc_plan_pipe(int mode, int max_workgroups, int forced_shape, int *nmains, int *nhelp);
/* The choice itself, as a pure function (no device needed): launch shape for n_frames on a device that holds
 * max_workgroups resident workgroups of the encoder kernel (1024 on MI355X).  Returns 1 (a frame per workgroup,
 * *nmains workgroups, *nhelp = 0) or 2 (*nmains main workgroups + a pool of *nhelp helper workgroups). */
int imcvt_hevc_plan(int n_frames, int max_workgroups, int force_team, int *nmains, int *nhelp);
/* Shape of the last launch: returns 1 / 2 / 3 (no helpers / fewer than two / two or more helpers per main workgroup);
 * *nteams receives the main workgroups of a launch with helpers (0 without). */
int imcvt_hevc_last_team(imcvt_hevc_ctx *ctx, int *nteams);
/* The same in full: returns 1 (no helpers) or 2 (pool), *nmains and *nhelp the workgroups of each kind. */
int imcvt_hevc_last_shape(imcvt_hevc_ctx *ctx, int *nmains, int *nhelp);
/* Debug / tuning aid: the next launches use exactly nmains main and nhelp helper workgroups (fewer mains when there are fewer
 * frames); anything but two positive numbers returns to imcvt_hevc_set_team's choice.  A shape the context cannot hold (more
 * workgroups than are resident at once, more main workgroups than mailboxes) makes imcvt_hevc_encode_device return
 * IMCVT_ERR_ARG before it touches any memory. */
void imcvt_hevc_set_shape(imcvt_hevc_ctx *ctx, int nmains, int nhelp);
/* Debug / tuning aid for launches with helpers: a main workgroup posts a 16x16 / 32x32 request only while fewer than lim16 / lim32
 * requests of that kind wait unclaimed in its queue shard (otherwise it evaluates the CU itself); prio >= 2 raises the wave
 * priority of the main workgroups.  Negative values return to the defaults derived from the launch shape.  Environment
 * overrides at context creation: IMCVT_POOL_LIM16, IMCVT_POOL_LIM32, IMCVT_POOL_PRIO.  Results do not depend on any of them. */
void imcvt_hevc_set_pool_tuning(imcvt_hevc_ctx *ctx, int lim16, int lim32, int prio);
/* Debug / tuning aid: per mille of the 16x16 / 32x32 CUs a main workgroup offers to the helpers (the rest it evaluates itself);
 * negative: derived from the launch shape.  Environment overrides: IMCVT_POOL_POST16, IMCVT_POOL_POST32. */
void imcvt_hevc_set_pool_split(imcvt_hevc_ctx *ctx, int post16, int post32);

/* Kernel-only time of the last imcvt_hevc_encode_device call on this context, in milliseconds, from HIP
 * events recorded on the launch stream (synchronises that stream).  <0 if nothing was launched. */
float imcvt_hevc_last_kernel_ms(imcvt_hevc_ctx *ctx);

/* Debug aid: decision trace of frame 0 of the next launch (8 ints per CU: y, x, size, kind, mode(s), cost, 0, 0)
 * into a device buffer of cap ints; pass NULL to disable. */
void imcvt_hevc_set_trace(imcvt_hevc_ctx *ctx, int *d_trace, int cap);

/* Waits for the context's last launch and returns how it ended: 0, or IMCVT_ERR_WATCHDOG (one line on stderr) when a wait
 * between cooperating workgroups exceeded 20 s and the launch was abandoned — its outputs are invalid.  The host-pointer entry
 * points check this themselves; callers of imcvt_hevc_encode_device call it after synchronising. */
int imcvt_hevc_last_status(imcvt_hevc_ctx *ctx);

/* Debug aid: per-frame clocks of the next launches into a device buffer of 4 x n 64-bit words (start, end in 100 MHz ticks, block
 * index, CUs the main workgroup kept because the helpers were busy); NULL to disable. */
void imcvt_hevc_set_frame_clock(imcvt_hevc_ctx *ctx, unsigned long long *d_buf);

/* Debug aid: the largest number of workgroups of the context's last launch that ran at the same time (waits for the launch). */
int imcvt_hevc_last_resident(imcvt_hevc_ctx *ctx);
/* Debug aid: microseconds between the start of the first and of the last workgroup of the context's last launch. */
long long imcvt_hevc_last_start_spread_us(imcvt_hevc_ctx *ctx);

/* Debug aid: per-wave cycle totals by phase ([3 roles][waves][categories], zeros unless the library was built with
 * -DIMCVT_PROF); copies up to n counters to `out`, optionally resets them; returns the number available. */
int imcvt_hevc_debug_prof(imcvt_hevc_ctx *ctx, unsigned long long *out, int n, int reset);

/* Debug aid: launches `grid` workgroups of the encoder kernel that only count themselves, wait ~1 ms and record how many
 * had started by then: the number of workgroups of such a launch that are resident at the same time. */
int imcvt_hevc_debug_census(imcvt_hevc_ctx *ctx, int grid);

/* Debug aid: what the HIP occupancy API reports for the encoder kernel on the current device. */
int imcvt_hevc_debug_occupancy(int *blocks_per_cu, int *cus, int *lds_per_block, int *lds_per_cu);

/* What the context plans its launches against: returns the device's compute units; *max_wg / *pipe_wg = workgroups of 192 / 256
 * threads (the latter with the pipe wave's dynamic LDS) that a launch may count on being resident at once — the HIP occupancy
 * API's blocks per compute unit (*occ_per_cu, *occ_pipe_per_cu) x compute units, lowered to what a census launch at context
 * creation found resident (*census_wg, *census_pipe; 0: not measured, e.g. with an explicit max_workgroups).  Any pointer may be NULL. */
int imcvt_hevc_residency(imcvt_hevc_ctx *ctx, int *max_wg, int *pipe_wg, int *occ_per_cu, int *occ_pipe_per_cu, int *census_wg, int *census_pipe);

/* Test aid: a co-tenant kernel — `grid` workgroups of 256 threads holding lds_bytes of LDS each that spin for about `ms` ms on
 * `stream` (a hipStream_t): what another kernel on the device does to this library's launches. */
int imcvt_hevc_debug_filler(int grid, int lds_bytes, int ms, void *stream);

/* Test aids for the submission queue of the host-pointer entry points: a stand-in for the device batch (a function with
 * HEVCImageEncoderBatch's signature; NULL restores the real one) so that the queue's logic can be exercised without a GPU, and
 * its counters (calls submitted, device batches run, frames in the largest batch). */
void imcvt_hevc_debug_set_backend(void *fn);
void imcvt_hevc_coalesce_stats(long *calls, long *batches, long *max_batch, int reset);

/* Library / build information, e.g. "imcvt_hevc gfx950 r4 ...". */
const char *imcvt_hevc_version(void);

#ifdef __cplusplus
}
#endif
#endif
