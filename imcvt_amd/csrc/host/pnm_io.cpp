// pnm_io.cpp — PNM reader / writer with the reference's observable behaviour (src/imageio_pnm.c:10-29 writer,
// :33-58 number scanner, :73-148 loader; interface src/imageio.h:8,14).  Host-only code of the drop-in converter
// (SURVEY.md §8f rank 1): it feeds gray8 / RGB24 pixel buffers to the H.265 path.
//
// Behaviours kept on purpose (each checked against the compiled reference in tests/test_host_cli.py):
//   * header numbers: anything that is not a digit or a '#' comment is a separator; '#' runs to CR or LF (:35-57)
//   * after the last header number the rest of that line is skipped, up to and including the next LF (:96-98) —
//     if the number was terminated by LF itself nothing is skipped
//   * maxval is read for P2/P3/P5/P6 only, must be 1..255 and is NOT used to rescale samples (:86-93, :136)
//   * raw PBM: rows are byte-padded, a set bit is black (0), a clear bit white (255); a short file marks the load
//     as failed but only after the whole picture was walked (:114-131)
//   * plain formats: every sample is one scanned number, stored modulo 256; plain PBM maps non-zero to 0 (:133-138)
//   * the pixel buffer is over-allocated by 8 bytes (:106) because the PBM unpacker always writes 8 pixels
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

namespace {

struct Scanner {
    FILE *fp;
    int last;                     // the byte that ended the previous number (or EOF)
    explicit Scanner(FILE *f) : fp(f), last(0) {}
    // next decimal number of the stream, -1 at end of file
    int number() {
        for (;;) {
            int c = fgetc(fp);
            if (c == EOF) { last = EOF; return -1; }
            if (c == '#') {
                do { c = fgetc(fp); } while (c != EOF && c != '\r' && c != '\n');
                if (c == EOF) { last = EOF; return -1; }
                continue;
            }
            if (c < '0' || c > '9') continue;
            int v = 0;
            do { v = v * 10 + (c - '0'); c = fgetc(fp); } while (c >= '0' && c <= '9');
            last = c;
            return v;
        }
    }
    void finish_line() { while (last != '\n' && last != EOF) last = fgetc(fp); }
};

}  // namespace

extern "C" int writePNMImageFile(const char *p_filename, const uint8_t *p_buf, int is_rgb, uint32_t height, uint32_t width) {
    if (width < 1 || height < 1) return 1;
    FILE *fp = fopen(p_filename, "wb");
    if (!fp) return 1;
    fprintf(fp, "P%c\n%d %d\n255\n", is_rgb ? '6' : '5', (int)width, (int)height);
    const size_t bytes = (size_t)(is_rgb ? 3 : 1) * width * height;
    const int short_write = fwrite(p_buf, 1, bytes, fp) != bytes;
    fclose(fp);
    return short_write;
}

extern "C" uint8_t *loadPNMImageFile(const char *p_filename, int *p_is_rgb, uint32_t *p_height, uint32_t *p_width) {
    FILE *fp = fopen(p_filename, "rb");
    if (!fp) return NULL;
    const int magic = fgetc(fp), kind = fgetc(fp) - '0';
    Scanner in(fp);
    const int w = in.number(), h = in.number();
    int maxval = 1;
    const bool has_maxval = kind == 2 || kind == 3 || kind == 5 || kind == 6;
    if (has_maxval) maxval = in.number();
    if (magic != 'P' || kind < 1 || kind > 6 || w < 1 || h < 1 || maxval < 1 || maxval > 255) { fclose(fp); return NULL; }
    in.finish_line();

    const int rgb = kind == 3 || kind == 6;
    *p_width = (uint32_t)w; *p_height = (uint32_t)h; *p_is_rgb = rgb;
    const size_t samples = (size_t)(rgb ? 3 : 1) * w * h;
    uint8_t *pix = (uint8_t *)malloc(samples + 8);
    if (pix) {
        bool bad = false;
        if (kind >= 5) {
            bad = fread(pix, 1, samples, fp) != samples;
        } else if (kind == 4) {
            for (size_t y = 0; y < (size_t)h; y++) {
                uint8_t *row = pix + y * (size_t)w;
                for (size_t x = 0; x < (size_t)w; x += 8) {
                    const int packed = fgetc(fp);                // EOF (-1) unpacks to eight black pixels, like the reference
                    bad = bad || packed == EOF;
                    for (int b = 0; b < 8; b++) row[x + b] = ((packed >> (7 - b)) & 1) ? 0 : 255;
                }
            }
        } else {
            for (size_t i = 0; i < samples; i++) {
                const int v = in.number();
                bad = bad || v < 0;
                pix[i] = (uint8_t)(kind != 1 ? v : (v ? 0 : 255));
            }
        }
        if (bad) { free(pix); pix = NULL; }
    }
    fclose(fp);
    return pix;
}
