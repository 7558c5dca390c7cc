/*
 * jls_oracle.c — TEST INFRASTRUCTURE ONLY.  A plain-C restatement of the reference's JPEG-LS encoder
 * (/root/reference/src/imageio_jls.c), in-memory instead of file based, used by tests/ and tools/ as the CPU
 * checker of the HIP path (imcvt_amd/csrc/jls_hip.hip).  The product never links or calls it.
 *
 * Pinned: tests/test_jls.py checks it byte-for-byte against (a) the golden digests of tests/golden/jls_kat.json,
 * generated from the compiled reference (oracle/_ref/libref_jls.so) by tests/golden/make_jls_golden.py, and
 * (b) the reference itself wherever oracle/_ref exists.
 *
 * Each function cites the reference lines it restates.  The structure is our own (one state struct, a byte-wise
 * bit packer) — the arithmetic is the reference's.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int a, b, c, n; } Ctx;                 /* A, B, C, N of one context (:241-244) */
typedef struct {
    uint8_t *out; size_t len;                           /* byte sink */
    unsigned acc; int free_bits;                        /* partial byte: `free_bits` unused low bits (8, or 7 after a 0xFF) */
} Bits;

static void bits_init(Bits *w, uint8_t *out) { w->out = out; w->len = 0; w->acc = 0; w->free_bits = 8; }
static void put_byte(Bits *w, unsigned v) { w->out[w->len++] = (uint8_t)v; }
static void put_be(Bits *w, unsigned v, int nbytes) { while (nbytes-- > 0) put_byte(w, (v >> (8 * nbytes)) & 0xFF); }   /* :153-159 */
/* one bit, MSB first; after a 0xFF byte the next byte starts with a stuffed 0 bit (:162-174) */
static void put_bit(Bits *w, int bit) {
    w->free_bits--;
    if (bit) w->acc |= 1u << w->free_bits;
    if (w->free_bits == 0) {
        put_byte(w, w->acc);
        w->free_bits = (w->acc == 0xFF) ? 7 : 8;
        w->acc = 0;
    }
}
static void put_bits(Bits *w, int v, int n) { while (n-- > 0) put_bit(w, (v >> n) & 1); }                               /* :177-180 */
static void put_zeros_one(Bits *w, int zeros) { while (zeros-- > 0) put_bit(w, 0); put_bit(w, 1); }
/* anything but a fresh 8-bit byte is written out — also the empty 7-bit byte that follows a 0xFF (:183-190: bitmask < 0x80) */
static void bits_flush(Bits *w) { if (w->free_bits != 8) { put_byte(w, w->acc); w->acc = 0; w->free_bits = 8; } }

static const int kJ[32] = {0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,5,5,6,6,7,7,8,9,10,11,12,13,14,15};   /* :14 */

typedef struct { int alpha, t1, t2, t3, quant, qbeta, qbpp, limit, a_init, near; } Par;
static Par parameters(int bpp, int near) {                                                               /* :26-38 */
    Par p; int f;
    p.near = near; p.alpha = 1 << bpp;
    f = ((p.alpha < 4096 ? p.alpha : 4096) + 127) / 256;
    p.t1 = f + 2 + 3 * near; p.t2 = 4 * f + 3 + 5 * near; p.t3 = 17 * f + 4 + 7 * near;
    p.quant = 2 * near + 1; p.qbeta = (p.alpha + 4 * near) / p.quant;
    for (p.qbpp = 1; (1 << p.qbpp) < p.qbeta; p.qbpp++) {}
    p.limit = 4 * bpp - p.qbpp - 1;
    p.a_init = (p.qbeta + 32) / 64; if (p.a_init < 2) p.a_init = 2;
    return p;
}
static int grad(const Par *p, int v) {                                                                    /* :67-76 */
    const int m = v < 0 ? -v : v, s = v < 0 ? -1 : 1;
    return m >= p->t3 ? 4 * s : m >= p->t2 ? 3 * s : m >= p->t1 ? 2 * s : m > p->near ? s : 0;
}
static int med(int a, int b, int c) {                                                                     /* :87-94 */
    const int lo = a < b ? a : b, hi = a < b ? b : a;
    return c >= hi ? lo : c <= lo ? hi : a + b - c;
}
static int quant_err(const Par *p, int e) { return e < 0 ? -((p->near - e) / p->quant) : (p->near + e) / p->quant; }   /* :97-102 */
static int wrap_err(const Par *p, int e) { if (e < 0) e += p->qbeta; if (e >= (p->qbeta + 1) / 2) e -= p->qbeta; return e; }   /* :105-111 */
static int golomb_k(int a, int n, int ri) { int k = 0; if (ri) a += n >> 1; while ((n << k) < a) k++; return k; }   /* :114-121 */
static void golomb(Bits *w, const Par *p, int limit, int v, int k) {                                       /* :193-203 */
    if ((v >> k) < limit) { put_zeros_one(w, v >> k); put_bits(w, v, k); }
    else { put_zeros_one(w, limit); put_bits(w, v - 1, p->qbpp); }
}
static int clampi(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }

/* one scan over a plane of ints that is both source and reconstruction (:240-399; img and imgrcon alias, :406) */
static void scan(Bits *w, int bpp, int near, int h, int wd, int *px) {
    Ctx cx[364], ri[2];
    const Par p = parameters(clampi(bpp, 8, 16), near);
    int run_idx = 0, y, x, i;
    for (i = 0; i < 364; i++) { cx[i].a = p.a_init; cx[i].b = 0; cx[i].c = 0; cx[i].n = 1; }
    for (i = 0; i < 2; i++) { ri[i].a = p.a_init; ri[i].b = 0; ri[i].c = 0; ri[i].n = 1; }
    for (y = 0; y < h; y++) {
        int in_run = 0, run_len = 0;
        for (x = 0; x < wd; x++) {
            const int v = px[y * wd + x];
            int a = 0, b = 0, c = 0, d = 0, q, sgn;                                    /* neighbourhood :46-65 */
            if (y > 0) { b = px[(y - 1) * wd + x]; d = (x + 1 < wd) ? px[(y - 1) * wd + x + 1] : b; }
            if (x == 0) { a = b; if (y > 1) c = px[(y - 2) * wd]; }
            else { a = px[y * wd + x - 1]; if (y > 0) c = px[(y - 1) * wd + x - 1]; }
            q = 81 * grad(&p, d - b) + 9 * grad(&p, b - c) + grad(&p, c - a);          /* :79-84 */
            sgn = q < 0 ? -1 : 1; if (q < 0) q = -q;
            if (q == 0) in_run = 1;
            if (in_run && (v - a <= near && a - v <= near)) {                          /* run continues :291-303 */
                px[y * wd + x] = a;
                if (++run_len >= (1 << kJ[run_idx])) { put_bit(w, 1); run_len -= 1 << kJ[run_idx]; if (run_idx < 31) run_idx++; }
                if (x == wd - 1 && run_len > 0) put_bit(w, 1);
            } else if (in_run) {                                                       /* run interruption :305-344 */
                const int glimit = p.limit - 1 - kJ[run_idx];
                int t, pred, e, k, map, me; Ctx *r;
                in_run = 0;
                put_bits(w, run_len, kJ[run_idx] + 1);
                run_len = 0; if (run_idx > 0) run_idx--;
                t = (a - b <= near && b - a <= near);
                sgn = (a > b + near) ? -1 : 1;
                pred = t ? a : b;
                e = quant_err(&p, sgn * (v - pred));
                px[y * wd + x] = near ? clampi(pred + sgn * p.quant * e, 0, p.alpha - 1) : v;
                e = wrap_err(&p, e);
                r = &ri[t];
                k = golomb_k(r->a, r->n, t);
                map = (e != 0) && ((e > 0) == (k == 0 && 2 * r->b < r->n));
                me = 2 * (e < 0 ? -e : e) - t - map;
                golomb(w, &p, glimit, me, k);
                if (e < 0) r->b++;
                r->a += (me + 1 - t) >> 1;
                if (r->n >= 64) { r->a >>= 1; r->b >>= 1; r->n >>= 1; }
                r->n++;
            } else {                                                                   /* regular mode :346-394 */
                Ctx *r = &cx[q - 1];
                int pred, e, k, map, me;
                run_len = 0;
                pred = clampi(med(a, b, c) + sgn * r->c, 0, p.alpha - 1);
                e = quant_err(&p, sgn * (v - pred));
                px[y * wd + x] = near ? clampi(pred + sgn * p.quant * e, 0, p.alpha - 1) : v;
                e = wrap_err(&p, e);
                k = golomb_k(r->a, r->n, 0);
                map = (k == 0) && (2 * r->b <= -r->n) && (near == 0);
                me = 2 * (e < 0 ? -e : e);
                if (e < 0) me -= map + 1; else me += map;
                golomb(w, &p, p.limit, me, k);
                r->b += e * p.quant; r->a += e < 0 ? -e : e;
                if (r->n >= 64) { r->a >>= 1; r->b >>= 1; r->n >>= 1; }
                r->n++;
                if (r->b <= -r->n) { r->b += r->n; if (r->b < -r->n + 1) r->b = -r->n + 1; r->c--; }
                else if (r->b > 0) { r->b -= r->n; if (r->b > 0) r->b = 0; r->c++; }
                r->c = clampi(r->c, -128, 127);
            }
        }
    }
    bits_flush(w);
}

/* Whole file image in memory (:402-426 and the headers :206-237).  img: h*w gray8 or h*w*3 RGB24.  Returns the length. */
long long jls_oracle_encode(const uint8_t *img, int is_rgb, int h, int w, int near, uint8_t *out) {
    const int planes = is_rgb ? 3 : 1;
    const size_t n = (size_t)h * w;
    int *plane = (int *)malloc(n * sizeof(int));
    Bits bw; int c; size_t i;
    if (!plane) return -1;
    bits_init(&bw, out);
    put_be(&bw, 0xFFD8, 2);
    put_be(&bw, is_rgb ? 0xFFF70011u : 0xFFF7000Bu, 4);
    put_be(&bw, 8, 1); put_be(&bw, (unsigned)h, 2); put_be(&bw, (unsigned)w, 2); put_be(&bw, (unsigned)planes, 1);
    for (c = 1; c <= planes; c++) put_be(&bw, ((unsigned)c << 16) | 0x1100u, 3);
    for (c = 1; c <= planes; c++) {
        put_be(&bw, 0xFFDA, 2); put_be(&bw, 8, 2); put_be(&bw, 1, 1); put_be(&bw, (unsigned)c, 1); put_be(&bw, 0, 1);
        put_be(&bw, (unsigned)near, 1); put_be(&bw, 0, 2);
        for (i = 0; i < n; i++) plane[i] = img[i * planes + (c - 1)];
        scan(&bw, 8, near, h, w, plane);
    }
    put_be(&bw, 0xFFD9, 2);
    free(plane);
    return (long long)bw.len;
}
long long jls_oracle_bound(int h, int w) { return 8LL * w * h + 65536; }            /* the reference's own buffer size (:440) */
