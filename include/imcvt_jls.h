/*
 * imcvt_jls.h — C ABI of libimcvt_jls.so, the MI355X (gfx950) implementation of ImCvt's JPEG-LS encoder
 * (BASELINE config 5; reference src/imageio_jls.c).  Plain pointers and sizes only.
 *
 * Section 1 is the reference's own interface for this path (src/imageio.h:20); section 2 is the in-memory and
 * device-resident batch surface (the reference's encoder functions are `static`, :402-426, so there is nothing
 * in-memory to mirror).  Every entry point needs a gfx950 device; there is no CPU fallback
 * (IMCVT_JLS_ERR_NO_DEVICE after one line on stderr).
 */
#ifndef IMCVT_JLS_H
#define IMCVT_JLS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define IMCVT_JLS_ERR_NO_DEVICE (-1)
#define IMCVT_JLS_ERR_HIP       (-2)
#define IMCVT_JLS_ERR_ARG       (-3)

/* 1. Replaces writeJLSImageFile — reference src/imageio.h:20, defined src/imageio_jls.c:428-477.
 *    p_buf: height*width gray8 or height*width*3 RGB24; near 0..4 (JPEG-LS NEAR; the reference's CLI only produces
 *    0..4, src/main.c:154).  Returns 0 on success, 1 on failure (size outside 1..32767 (:437), allocation, device,
 *    fopen, short write). */
int writeJLSImageFile(const char *p_filename, const uint8_t *p_buf, int is_rgb, uint32_t height, uint32_t width, int near);

/* 2a. The same stream in memory (HOST pointers).  out must hold imcvt_jls_stream_bound(h, w) bytes.
 *     Returns the stream length, or a negative IMCVT_JLS_ERR_*. */
long long imcvt_jls_encode(const uint8_t *img, int is_rgb, int h, int w, int near, uint8_t *out);
/* The reference's own output allocation (src/imageio_jls.c:440): 8*w*h + 65536. */
long long imcvt_jls_stream_bound(int h, int w);

/* 2b. Device-resident batch of gray planes, all planes concurrently (few lossless planes: each spread over the device).  All pointers are
 *     DEVICE pointers.  d_out receives the complete .jls stream of the plane (headers and EOI included). */
typedef struct imcvt_jls_plane {
    const unsigned char *d_img;   /* h*w gray8 */
    unsigned char       *d_out;   /* >= imcvt_jls_stream_bound(h, w) bytes */
    long long           *d_len;   /* receives the stream length */
    int                  h, w, near;
} imcvt_jls_plane;
/* Asynchronous on `stream` (hipStream_t, may be NULL).  Returns 0 or a negative IMCVT_JLS_ERR_*. */
int imcvt_jls_encode_device(int n, const imcvt_jls_plane *planes, void *stream);
/* Kernel-only time of the last imcvt_jls_encode_device call (HIP events on its stream; synchronises it). */
float imcvt_jls_last_kernel_ms(void);
const char *imcvt_jls_version(void);
/* Which path the last launch took: 1 = one plane spread over the device (lossless planes, at most 64 of them: data-parallel
 * classification, one lane per context chain, prefix sums of code lengths, chunked bit stuffing — jls_par.h), 0 = one
 * walker per plane (near-lossless, or many planes).  Environment override: IMCVT_JLS_PAR=0/1 (lossless only). */
int imcvt_jls_last_path(void);

#ifdef __cplusplus
}
#endif
#endif
