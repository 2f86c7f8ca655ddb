/* TEST INFRASTRUCTURE ONLY -- never linked into the product library.
 *
 * Seam around the reference's own WAV layer: /root/reference/receiver/wav.c is compiled where it
 * lies (oracle/Makefile, target ref_wav) and driven exactly as the capture thread drives it
 * (receiver/capt_sched.c:87-101 PrepWav/EndWav, :516 wav_write).  Pins the product's nvx_wav_*
 * (navtex_amd/csrc/nvx_wav.c) to the reference in both directions:
 *
 *   ref_wav write <frames.bin> <out.wav> [rate]   int16 I,Q frames -> wav_open(WRITE) + the four wav_set_* calls of
 *                                                 capt_sched.c:92-95 + wav_write + wav_close
 *   ref_wav read  <in.wav> <frames.bin>           wav_open(READ) + wav_get_* + wav_read (wav.c:494-528) -> raw frames;
 *                                                 stdout: "format channels rate sample_size length"
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "wav.h"

static void *slurp(const char *path, size_t *n)
{
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    fseek(f, 0, SEEK_END);
    long len = ftell(f);
    fseek(f, 0, SEEK_SET);
    void *p = malloc(len > 0 ? (size_t)len : 1);
    if (len > 0 && fread(p, 1, (size_t)len, f) != (size_t)len) { perror("fread"); exit(2); }
    fclose(f);
    *n = (size_t)len;
    return p;
}

int main(int argc, char **argv)
{
    if (argc >= 4 && !strcmp(argv[1], "write")) {
        size_t bytes = 0;
        void *frames = slurp(argv[2], &bytes);
        const int rate = argc > 4 ? atoi(argv[4]) : 252000;
        WavFile *fp = wav_open(argv[3], WAV_OPEN_WRITE);
        if (!fp) return 3;
        wav_set_format(fp, WAV_FORMAT_PCM);
        wav_set_num_channels(fp, 2);
        wav_set_sample_rate(fp, (WavU32)rate);
        wav_set_sample_size(fp, sizeof(short));
        /* the consumer loop hands over spans as they come: two calls, so appending is exercised too */
        const size_t n = bytes / 4, first = n / 3;
        size_t w = wav_write(fp, frames, first);
        w += wav_write(fp, (char *)frames + 4 * first, n - first);
        wav_close(fp);
        free(frames);
        return w == n ? 0 : 4;
    }
    if (argc >= 4 && !strcmp(argv[1], "read")) {
        WavFile *fp = wav_open(argv[2], WAV_OPEN_READ);
        if (!fp) return 3;
        const size_t n = wav_get_length(fp);
        printf("%u %u %u %zu %zu\n", (unsigned)wav_get_format(fp), (unsigned)wav_get_num_channels(fp),
               (unsigned)wav_get_sample_rate(fp), wav_get_sample_size(fp), n);
        const size_t frame = wav_get_sample_size(fp) * wav_get_num_channels(fp);
        void *buf = malloc(n * frame + 1);
        const size_t got = wav_read(fp, buf, n);
        wav_close(fp);
        FILE *o = fopen(argv[3], "wb");
        if (!o) { perror(argv[3]); return 2; }
        fwrite(buf, frame, got, o);
        fclose(o);
        free(buf);
        return got == n ? 0 : 4;
    }
    fprintf(stderr, "usage: ref_wav write <frames.bin> <out.wav> [rate] | ref_wav read <in.wav> <frames.bin>\n");
    return 1;
}
