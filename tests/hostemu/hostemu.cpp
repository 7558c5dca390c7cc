// tests/hostemu/hostemu.cpp — TEST-ONLY harness: compiles the device encoder source (imcvt_amd/csrc/hevc_core.h,
// hevc_frame.h) for the host with a wavefront emulated as a serial loop over 64 lanes.  It exists so the
// bit-exactness of the kernel LOGIC can be checked against the oracle on a machine without a GPU; it is not
// part of the product, is never loaded by imcvt_amd, and is not a fallback.
#define IMCVT_HOSTEMU 1
#include <stdlib.h>
#include <string.h>
#include "../../imcvt_amd/csrc/hevc_frame.h"
#include "../../imcvt_amd/csrc/hevc_tables.h"

extern "C" int hostemu_HEVCImageEncoder(unsigned char *pbuffer, const unsigned char *img, unsigned char *img_rcon,
                                        int *ysz, int *xsz, int qpd6, int *trace, int trace_cap) {
    static Tables T; static ColdTables K; static int ready = 0;
    if (!ready) { imcvt::build_tables(T, K); ready = 1; }
    const int h = *ysz, w = *xsz;
    const int hp = ((h < 8192 ? h : 8192) + 31) / 32 * 32, wp = ((w < 8192 ? w : 8192) + 31) / 32 * 32;
    Shm *S = (Shm *)calloc(1, sizeof(Shm));
    Scratch sc;
    sc.lv = (i16 *)calloc((size_t)NWAVES * LV_PER_WAVE, sizeof(i16));
    sc.bytes = (u8 *)calloc((size_t)NWAVES * NMODE * TRIAL_BYTES, 1);
    sc.above_sz = (u8 *)calloc((size_t)wp / 4 + 8, 1);
    sc.trace = trace; sc.trace_cap = trace_cap;
    u8 hdr[96];
    FrameJob job;
    int out_len = 0;
    job.img = img; job.out = pbuffer; job.rcon = img_rcon; job.h = h; job.w = w; job.hp = hp; job.wp = wp; job.q = qpd6;
    job.hdr_len = imcvt::build_headers(hdr, qpd6, hp, wp);
    job.out_len = &out_len;
    g_shm_host = S; sc.prof = nullptr;
    encode_frame(&T, &K, job, sc, hdr);
    free(sc.lv); free(sc.bytes); free(sc.above_sz); free(S);
    *ysz = hp; *xsz = wp;
    return out_len;
}
extern "C" int hostemu_shm_bytes(void) { return (int)sizeof(Shm); }
