// imcvt_cli.cpp — command-line converter with the reference's interface (src/main.c): same switch parsing
// (:87-132), same per-file progress / error / summary lines (:135-140, :177, :216-221), same exit code (number of
// failed files, :223; -1 with the usage text when no file is given, :156-159), same output-suffix dispatch (:189-204).
//
// What is MI355X-native here: the reference converts one file at a time (:162), and one H.265 frame keeps a single
// compute unit busy, so before the reference's sequential loop is replayed this driver loads every input bound for
// .h265 / .265 / .hevc and encodes them in ONE device batch (HEVCImageEncoderBatch, include/imcvt_hevc.h §2).  The
// loop then runs in the reference's order and only writes the already encoded streams, so stdout, exit code and file
// contents are those of the reference.
//
// Scope: inputs are tried as PNM, PNG, BMP, QOI in the reference's order (:183-186); outputs are PNM,
// PNG, BMP, QOI, H.265 and — when built with the JPEG-LS module — .jls (pnm_io / png_io / bmp_io / qoi_io.cpp).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "../../../include/imcvt_hevc.h"

extern "C" uint8_t *loadPNMImageFile(const char *, int *, uint32_t *, uint32_t *);
extern "C" uint8_t *loadBMPImageFile(const char *, int *, uint32_t *, uint32_t *);
extern "C" uint8_t *loadQOIImageFile(const char *, int *, uint32_t *, uint32_t *);
extern "C" uint8_t *imcvt_load_png(const char *, int *, uint32_t *, uint32_t *, int quiet);
extern "C" int writePNMImageFile(const char *, const uint8_t *, int, uint32_t, uint32_t);
extern "C" int writePNGImageFile(const char *, const uint8_t *, int, uint32_t, uint32_t);
extern "C" int writeBMPImageFile(const char *, const uint8_t *, int, uint32_t, uint32_t);
extern "C" int writeQOIImageFile(const char *, const uint8_t *, int, uint32_t, uint32_t);
#ifdef IMCVT_WITH_JLS
extern "C" int writeJLSImageFile(const char *, const uint8_t *, int, uint32_t, uint32_t, int);
#endif

static const int kMaxFiles = 999;              // src/main.c:84

static void usage() {
    fputs("ImCvt (MI355X build of the H.265 intra path)\n"
          "usage:  imcvt [-switches] <in1> -o <out1> [<in2> -o <out2>] ...\n"
          "  <in>  : PNM (P1..P6), PNG (8-bit gray / RGB / RGBA), BMP, QOI — recognised by content\n"
          "  <out> : .pnm .pgm .ppm .png .bmp .qoi | .h265 .265 .hevc (gray 8-bit, encoded on the GPU)"
#ifdef IMCVT_WITH_JLS
          " | .jls"
#endif
          "\n  switches: -f  overwrite existing outputs;  -0 .. -4  H.265 (qp-4)/6"
#ifdef IMCVT_WITH_JLS
          " / JPEG-LS NEAR"
#endif
          "\n\n", stdout);
}

static char lower(char c) { return (c >= 'A' && c <= 'Z') ? (char)(c + 32) : c; }
static bool ends_with_nocase(const char *s, const char *suffix) {             // src/main.c:33-47
    const size_t n = strlen(s), m = strlen(suffix);
    if (m > n) return false;
    for (size_t i = 0; i < m; i++) if (lower(s[n - m + i]) != lower(suffix[i])) return false;
    return true;
}
// src with its extension replaced by `ext` (an extension is a '.' after the last path separator), :50-73
static std::string with_extension(const char *src, const char *ext) {
    std::string s(src);
    size_t cut = s.size();
    for (size_t i = s.size(); i-- > 0;) {
        if (s[i] == '/' || s[i] == '\\') break;
        if (s[i] == '.') { cut = i; break; }
    }
    return s.substr(0, cut) + "." + ext;
}
static bool exists(const char *name) { FILE *f = fopen(name, "rb"); if (f) fclose(f); return f != NULL; }
static bool is_hevc_name(const char *n) { return ends_with_nocase(n, "h265") || ends_with_nocase(n, "265") || ends_with_nocase(n, "hevc"); }

// the reference's loader chain (:183-186); quiet: no stdout lines (the PNG loader has some)
static uint8_t *load_image(const char *src, int *rgb, uint32_t *h, uint32_t *w, bool quiet) {
    uint8_t *px = loadPNMImageFile(src, rgb, h, w);
    if (!px) px = imcvt_load_png(src, rgb, h, w, quiet);
    if (!px) px = loadBMPImageFile(src, rgb, h, w);
    if (!px) px = loadQOIImageFile(src, rgb, h, w);
    return px;
}

struct Job { const char *src = NULL; std::string dst; bool dst_given = false; };
struct Encoded { bool tried = false, ok = false; std::vector<unsigned char> stream; int len = 0; uint32_t h = 0, w = 0; unsigned long long hash = 0; };
static unsigned long long gray_hash(const uint8_t *px, int rgb, size_t n) {      // FNV-1a over the samples the encoder sees (green of RGB, src/imageio_hevc.c:24-26)
    unsigned long long hsh = 1469598103934665603ull;
    for (size_t k = 0; k < n; k++) { hsh ^= rgb ? px[3 * k + 1] : px[k]; hsh *= 1099511628211ull; }
    return hsh;
}

int main(int argc, char **argv) {
    bool sw[128] = {false};
    std::vector<Job> jobs;
    bool next_is_dst = false;
    for (int i = 1; i < argc; i++) {                                         // src/main.c:104-129
        const char *a = argv[i];
        if (a[0] == '-') {
            for (a++; *a; a++) {
                if ((unsigned char)*a < 128) sw[(unsigned char)*a] = true;
                if (*a == 'o') next_is_dst = true;
            }
        } else if ((int)jobs.size() < kMaxFiles) {
            if (next_is_dst) {
                next_is_dst = false;
                if (!jobs.empty()) { jobs.back().dst = a; jobs.back().dst_given = true; }
            } else {
                Job j; j.src = a; jobs.push_back(j);
            }
        }
    }
    const bool force = sw['F'] || sw['f'];
    const int level = sw['4'] ? 4 : sw['3'] ? 3 : sw['2'] ? 2 : sw['1'] ? 1 : 0;   // H.265 (qp-4)/6 and JPEG-LS NEAR share it (:154)
    if (jobs.empty()) { usage(); return -1; }
    for (Job &j : jobs) if (!j.dst_given) j.dst = with_extension(j.src, "png");    // :171-175

    // ---- H.265-bound inputs are encoded ahead of the reference's loop, in device batches (one frame keeps only a few compute
    // units busy; a batch fans out over the GPUs).  The loop below stays the authority: it reloads every input and uses an
    // early result only if the samples are still the ones that were encoded, so a command line whose earlier job rewrites a
    // later job's input behaves like the reference's strictly sequential loop (such inputs are not pre-encoded at all).
    const int n = (int)jobs.size();
    std::vector<Encoded> enc(n);
    {
        const size_t kBudgetPx = (size_t)256 << 20;      // samples per batch: ~256 MB in, ~0.8 GB of stream / reconstruction buffers
        std::vector<int> idx, hs, ws, qs, lens;
        std::vector<std::vector<unsigned char>> gray, rcon;
        size_t px_in_batch = 0;
        auto flush = [&]() {
            const int m = (int)idx.size();
            if (m > 0) {
                std::vector<unsigned char *> outs(m), rcs(m); std::vector<const unsigned char *> ins(m);
                rcon.assign(m, std::vector<unsigned char>()); lens.assign(m, 0);
                for (int k = 0; k < m; k++) {
                    enc[idx[k]].stream.resize((size_t)imcvt_hevc_stream_bound(hs[k], ws[k]));
                    rcon[k].resize((size_t)imcvt_hevc_padded(hs[k]) * imcvt_hevc_padded(ws[k]));
                    outs[k] = enc[idx[k]].stream.data(); rcs[k] = rcon[k].data(); ins[k] = gray[k].data();
                }
                const int rc = HEVCImageEncoderBatch(m, outs.data(), ins.data(), rcs.data(), hs.data(), ws.data(), qs.data(), lens.data());
                for (int k = 0; k < m; k++) {
                    Encoded &e = enc[idx[k]];
                    e.ok = rc == 0 && lens[k] > 0; e.len = lens[k];
                    if (e.ok) { e.stream.resize((size_t)e.len); e.stream.shrink_to_fit(); }
                    else { e.tried = false; std::vector<unsigned char>().swap(e.stream); }      // the loop encodes this file on its own
                }
            }
            idx.clear(); hs.clear(); ws.clear(); qs.clear(); gray.clear(); rcon.clear(); px_in_batch = 0;
        };
        for (int i = 0; i < n; i++) {
            if (!is_hevc_name(jobs[i].dst.c_str())) continue;
            bool rewritten = false;                      // an earlier job of this command line writes this input
            for (int k = 0; k < i && !rewritten; k++) rewritten = jobs[k].dst == jobs[i].src;
            if (rewritten) continue;
            int rgb = 0; uint32_t h = 0, w = 0;
            uint8_t *px = load_image(jobs[i].src, &rgb, &h, &w, true);
            if (!px) continue;
            const size_t npx = (size_t)h * w;
            if (px_in_batch && px_in_batch + npx > kBudgetPx) flush();
            std::vector<unsigned char> g(npx);
            for (size_t k = 0; k < npx; k++) g[k] = rgb ? px[3 * k + 1] : px[k];
            enc[i].tried = true; enc[i].h = h; enc[i].w = w; enc[i].hash = gray_hash(px, rgb, npx);
            free(px);
            idx.push_back(i); hs.push_back((int)h); ws.push_back((int)w); qs.push_back(level);
            gray.push_back(std::move(g));
            px_in_batch += npx;
        }
        flush();
    }

    // ---- the reference's loop (:162-211)
    int converted = 0;
    for (int i = 0; i < n; i++) {
        const char *src = jobs[i].src, *dst = jobs[i].dst.c_str();
        printf("(%d/%d)  %s -> %s\n", i + 1, n, src, dst);
        if (!exists(src)) { printf("   ***ERROR: %s not exist\n", src); continue; }
        if (!force && exists(dst)) { printf("   ***ERROR: %s already exist\n", dst); continue; }
        int rgb = 0; uint32_t h = 0, w = 0;
        uint8_t *px = load_image(src, &rgb, &h, &w, false);
        if (!px) { printf("   ***ERROR: open %s failed\n", src); continue; }
        int failed;
        if (ends_with_nocase(dst, "pnm") || ends_with_nocase(dst, "ppm") || ends_with_nocase(dst, "pgm")) {
            failed = writePNMImageFile(dst, px, rgb, h, w);
        } else if (ends_with_nocase(dst, "png")) {
            failed = writePNGImageFile(dst, px, rgb, h, w);
        } else if (ends_with_nocase(dst, "bmp")) {
            failed = writeBMPImageFile(dst, px, rgb, h, w);
        } else if (ends_with_nocase(dst, "qoi")) {
            failed = writeQOIImageFile(dst, px, rgb, h, w);
#ifdef IMCVT_WITH_JLS
        } else if (ends_with_nocase(dst, "jls")) {
            failed = writeJLSImageFile(dst, px, rgb, h, w, level);
#endif
        } else if (is_hevc_name(dst)) {
            if (enc[i].tried && enc[i].ok && enc[i].h == h && enc[i].w == w && enc[i].hash == gray_hash(px, rgb, (size_t)h * w)) {
                if (rgb) printf("   warning: this HEVCencoder currently only support gray 8-bit image instead of RGB image. Only compress the green channel of this image.\n");
                failed = 1;
                FILE *fp = fopen(dst, "wb");
                if (fp) { failed = fwrite(enc[i].stream.data(), 1, (size_t)enc[i].len, fp) != (size_t)enc[i].len; fclose(fp); }
            } else {
                failed = writeHEVCImageFile(dst, px, rgb, h, w, level);        // not pre-encoded, its batch failed, or the input changed since: encode what the loop just loaded
            }
        } else {
            free(px);
            printf("   ***ERROR: unsupported output suffix: %s\n", dst);
            continue;
        }
        free(px);
        if (failed) { printf("   ***ERROR: write %s failed\n", dst); continue; }
        converted++;
    }
    const int failures = n - converted;
    if (n > 1) {                                                             // :216-221
        printf("\nsummary:");
        if (converted) printf("  %d file converted", converted);
        if (failures) printf("  %d failed", failures);
        printf("\n");
    }
    return failures;
}
