/* nvx_wav.c -- minimal PCM WAV reader / writer for the file path.
 *
 * The reference links a general WAV library (receiver/wav.c, wav.h) but only
 * ever writes 2-channel / 16-bit / 252 kHz captures with it
 * (receiver/capt_sched.c:87-101, :516).  This is the subset that path needs,
 * with the same call shape (open / set_* / read / write / close, counts in
 * frames, errors through a thread-local message instead of return codes,
 * wav.c:32,137-140) and the same canonical 44-byte header:
 *   "RIFF" size "WAVE" "fmt " 16 fmt=1 ch rate byterate align bits "data" size
 * Reading skips unknown chunks, so files from other writers load too.        */
#include "navtex_amd.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct nvx_wav {
    FILE *fp;
    uint32_t mode;
    uint16_t format, channels, bits;
    uint32_t rate;
    long data_start;
    uint32_t data_bytes;        /* header value (read) / running total (write) */
    size_t pos_frames;
};

static __thread char wav_errbuf[160];
const char *nvx_wav_err(void) { return wav_errbuf; }
static void seterr(const char *m) { snprintf(wav_errbuf, sizeof wav_errbuf, "%s", m); }

static void put16(unsigned char *p, uint16_t v) { p[0] = (unsigned char)v; p[1] = (unsigned char)(v >> 8); }
static void put32(unsigned char *p, uint32_t v) { for (int i = 0; i < 4; i++) p[i] = (unsigned char)(v >> (8 * i)); }
static uint16_t get16(const unsigned char *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
static uint32_t get32(const unsigned char *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

static size_t frame_bytes(const nvx_wav *w) { return (size_t)w->channels * (w->bits / 8); }

static int write_header(nvx_wav *w)
{
    unsigned char h[44];
    uint32_t align = (uint32_t)frame_bytes(w);
    memcpy(h, "RIFF", 4);      put32(h + 4, 36 + w->data_bytes);
    memcpy(h + 8, "WAVE", 4);  memcpy(h + 12, "fmt ", 4); put32(h + 16, 16);
    put16(h + 20, w->format);  put16(h + 22, w->channels);
    put32(h + 24, w->rate);    put32(h + 28, w->rate * align);
    put16(h + 32, (uint16_t)align); put16(h + 34, w->bits);
    memcpy(h + 36, "data", 4); put32(h + 40, w->data_bytes);
    if (fseek(w->fp, 0, SEEK_SET) != 0 || fwrite(h, 1, 44, w->fp) != 44) { seterr("wav: header write failed"); return -1; }
    return 0;
}

static int read_header(nvx_wav *w)
{
    unsigned char b[12];
    if (fread(b, 1, 12, w->fp) != 12 || memcmp(b, "RIFF", 4) || memcmp(b + 8, "WAVE", 4)) { seterr("wav: not a RIFF/WAVE file"); return -1; }
    int have_fmt = 0;
    for (;;) {
        unsigned char c[8];
        if (fread(c, 1, 8, w->fp) != 8) { seterr("wav: no data chunk"); return -1; }
        uint32_t len = get32(c + 4);
        if (!memcmp(c, "fmt ", 4)) {
            unsigned char f[16];
            if (len < 16 || fread(f, 1, 16, w->fp) != 16) { seterr("wav: short fmt chunk"); return -1; }
            w->format = get16(f); w->channels = get16(f + 2); w->rate = get32(f + 4); w->bits = get16(f + 14);
            if (len > 16) fseek(w->fp, (long)(len - 16 + (len & 1)), SEEK_CUR);
            have_fmt = 1;
        } else if (!memcmp(c, "data", 4)) {
            if (!have_fmt) { seterr("wav: data before fmt"); return -1; }
            w->data_bytes = len; w->data_start = ftell(w->fp);
            return 0;
        } else {
            fseek(w->fp, (long)(len + (len & 1)), SEEK_CUR);
        }
    }
}

nvx_wav *nvx_wav_open(const char *filename, uint32_t mode)
{
    wav_errbuf[0] = 0;
    if (mode != NVX_WAV_OPEN_READ && mode != NVX_WAV_OPEN_WRITE) { seterr("wav: bad mode"); return NULL; }
    nvx_wav *w = (nvx_wav *)calloc(1, sizeof *w);
    if (!w) { seterr("wav: out of memory"); return NULL; }
    w->mode = mode;
    w->fp = fopen(filename, mode == NVX_WAV_OPEN_READ ? "rb" : "wb");
    if (!w->fp) { seterr("wav: cannot open file"); free(w); return NULL; }
    if (mode == NVX_WAV_OPEN_READ) {
        if (read_header(w) != 0) { fclose(w->fp); free(w); return NULL; }
    } else {
        w->format = 1; w->channels = 2; w->bits = 16; w->rate = 44100;   /* wav.c defaults */
        w->data_start = 44;
        if (write_header(w) != 0) { fclose(w->fp); free(w); return NULL; }
    }
    return w;
}

int nvx_wav_close(nvx_wav *w)
{
    if (!w) return 0;
    int rc = 0;
    if (w->mode == NVX_WAV_OPEN_WRITE) {
        if (w->data_bytes & 1) { fputc(0, w->fp); }
        rc = write_header(w);
    }
    if (fclose(w->fp) != 0) rc = -1;
    free(w);
    return rc;
}

size_t nvx_wav_read(nvx_wav *w, void *buffer, size_t frames)
{
    if (!w || w->mode != NVX_WAV_OPEN_READ) { seterr("wav: not open for reading"); return 0; }
    size_t fb = frame_bytes(w);
    if (!fb) return 0;
    size_t total = w->data_bytes / fb;
    if (w->pos_frames + frames > total) frames = total - w->pos_frames;
    size_t got = fread(buffer, fb, frames, w->fp);
    w->pos_frames += got;
    return got;
}

size_t nvx_wav_write(nvx_wav *w, const void *buffer, size_t frames)
{
    if (!w || w->mode != NVX_WAV_OPEN_WRITE) { seterr("wav: not open for writing"); return 0; }
    size_t fb = frame_bytes(w);
    if (w->pos_frames == 0) { if (write_header(w) != 0) return 0; fseek(w->fp, 44, SEEK_SET); }
    size_t put = fwrite(buffer, fb, frames, w->fp);
    w->pos_frames += put;
    w->data_bytes += (uint32_t)(put * fb);
    return put;
}

void nvx_wav_set_format(nvx_wav *w, uint16_t f)          { if (w) w->format = f; }
void nvx_wav_set_num_channels(nvx_wav *w, uint16_t n)    { if (w) w->channels = n; }
void nvx_wav_set_sample_rate(nvx_wav *w, uint32_t r)     { if (w) w->rate = r; }
void nvx_wav_set_sample_size(nvx_wav *w, size_t bytes)   { if (w) w->bits = (uint16_t)(bytes * 8); }
uint16_t nvx_wav_get_format(const nvx_wav *w)            { return w ? w->format : 0; }
uint16_t nvx_wav_get_num_channels(const nvx_wav *w)      { return w ? w->channels : 0; }
uint32_t nvx_wav_get_sample_rate(const nvx_wav *w)       { return w ? w->rate : 0; }
size_t   nvx_wav_get_sample_size(const nvx_wav *w)       { return w ? (size_t)(w->bits / 8) : 0; }
size_t   nvx_wav_get_length(const nvx_wav *w)            { size_t fb = w ? frame_bytes(w) : 0; return fb ? w->data_bytes / fb : 0; }
