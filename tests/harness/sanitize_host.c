/* Sanitizer driver for the host-side C of the product (nvx_sitor.c, nvx_wav.c,
 * nvx_synth_host.c, nvx_store.c, nvx_fsm.h), built with -fsanitize=address,undefined by
 * tests/test_sanitizers.py.  Feeds bit strings from a file, random bits, WAV round
 * trips and generator calls; any memory error or UB aborts the process.          */
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "navtex_amd.h"
#include "nvx_fsm.h"

void nvx_set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }

static unsigned long n_msgs, n_trace;
static void on_msg(void *u, const char *bbbb, const char *msg, int freq) { (void)u; (void)freq; n_msgs += strlen(bbbb) + strlen(msg) > 0; }
static void on_trace(void *u, const char *t) { (void)u; n_trace += strlen(t); }

int main(int argc, char **argv)
{
    /* 1. character layer on every line of the case file (one 'B'/'Y' string per line) */
    if (argc > 1) {
        FILE *f = fopen(argv[1], "r");
        if (!f) return 2;
        size_t cap = 1 << 22; char *line = malloc(cap);
        while (fgets(line, (int)cap, f)) {
            nvx_sitor *s = nvx_sitor_new(518, on_msg, NULL);
            nvx_sitor_set_trace(s, on_trace, NULL);
            nvx_sitor_receive_bits(s, line, strlen(line));
            nvx_sitor_reset(s);
            nvx_sitor_receive_bits(s, line, strlen(line) / 2);
            nvx_sitor_free(s);
        }
        free(line); fclose(f);
    }
    /* 2. long random bit streams: no line feeds for long stretches -> the 5000-byte buffers */
    unsigned x = 12345;
    nvx_sitor *s = nvx_sitor_new(490, on_msg, NULL);
    for (long i = 0; i < 3000000; i++) { x = x * 1103515245u + 12345u; nvx_sitor_receive_bit(s, (x >> 16) & 1 ? 'B' : 'Y'); }
    /* a legal stream whose lines never end: phasing, then 6000 x 'E' without line feed */
    char big[8000]; memset(big, 'E', 6000); big[6000] = 0;
    size_t nb = nvx_sitor_encode(big, 10, NULL, 0);
    char *bits = malloc(nb + 1);
    nvx_sitor_encode(big, 10, bits, nb);
    nvx_sitor_receive_bits(s, bits, nb);
    free(bits);
    nvx_sitor_free(s);
    /* 3. encoder with every byte value */
    char all[256]; for (int i = 1; i < 256; i++) all[i - 1] = (char)i; all[255] = 0;
    nb = nvx_sitor_encode(all, 3, NULL, 0);
    bits = malloc(nb ? nb : 1); nvx_sitor_encode(all, 3, bits, nb); free(bits);
    /* 4. WAV round trip + truncated / garbage files */
    const char *path = argc > 2 ? argv[2] : "/tmp/nvx_san.wav";
    short iq[2 * 1000]; for (int i = 0; i < 2000; i++) iq[i] = (short)(i * 37);
    nvx_wav *w = nvx_wav_open(path, NVX_WAV_OPEN_WRITE);
    nvx_wav_set_format(w, 1); nvx_wav_set_num_channels(w, 2); nvx_wav_set_sample_rate(w, 252000); nvx_wav_set_sample_size(w, 2);
    nvx_wav_write(w, iq, 1000); nvx_wav_close(w);
    w = nvx_wav_open(path, NVX_WAV_OPEN_READ);
    short back[2 * 1200]; size_t got = nvx_wav_read(w, back, 1200);
    if (got != 1000 || memcmp(back, iq, sizeof iq)) return 3;
    nvx_wav_close(w);
    FILE *g = fopen(path, "wb"); fwrite("RIFF\4\0\0\0WAVEfmt ", 1, 16, g); fclose(g);
    if (nvx_wav_open(path, NVX_WAV_OPEN_READ)) return 4;
    /* 5. generator: all carrier counts, both rates, odd offsets */
    for (int nc = 0; nc <= NVX_SYNTH_MAX_CARRIERS; nc += 5) {
        nvx_synth_stream st; memset(&st, 0, sizeof st);
        st.seed = 7; st.noise_amp = 100; st.n_carriers = nc;
        for (int c = 0; c < nc; c++) { st.carrier[c].freq_hz = 14000 - 1000 * c; st.carrier[c].shift_hz = 85; st.carrier[c].amplitude = 1000;
                                       st.carrier[c].bit_offset = 17 * c; st.carrier[c].n_bits = 5; st.carrier[c].bits = "BYYBY"; }
        short out[2 * 7001];
        if (nvx_synth_host(&st, NVX_RATE_IN, 123456789ull, 7001, out) != NVX_OK) return 5;
        if (nvx_synth_host(&st, NVX_RATE_RAW, 0, 7001, out) != NVX_OK) return 5;
    }
    /* 6. SQLite sink: schema, replace, purge, errors (argv[3] = database path) */
    if (argc > 3) {
        nvx_store *st = NULL;
        if (nvx_store_open(argv[3], 1, &st) != NVX_OK) return 6;
        nvx_store_set_time(st, 1741350896);
        for (int i = 0; i < 40; i++) {
            char id[8]; snprintf(id, sizeof id, "P%c%02d", 'A' + i % 5, i % 7);
            if (nvx_store_add_message(st, id, "ZCZC\nTEXT\nNNNN\n", i & 1 ? 518 : 490) != 0) return 6;
        }
        if (nvx_store_add_message(st, "", "", 518) != 0 || nvx_store_add_message(NULL, "x", "y", 1) != -1) return 6;
        nvx_store_set_time(st, 1741350896 + 80 * 3600);
        if (nvx_store_purge(st, 0) <= 0) return 6;
        unsigned long long a = 0, f = 0; nvx_store_stats(st, (uint64_t *)&a, (uint64_t *)&f);
        if (a != 41 || f != 0) return 6;
        nvx_store_close(st);
        nvx_store *bad = NULL;
        if (nvx_store_open("/nonexistent-dir/x.db", 1, &bad) == NVX_OK) return 6;
    }
    /* 7. demodulator FSM: every table entry, then random words through the per-sample rule and the tables */
    {
        static uint32_t tab[NVX_FSM_TABLE_ALLOC];
        for (int p1 = 0; p1 < 9; p1++) for (int so = 0; so < 10; so++) for (int a = 0; a < 9; a++) for (int b = 0; b < 9; b++)
            tab[NVX_FSM_KEY(p1, so, a, b)] = nvx_fsm_table_entry(p1, so, a, b);
        for (int p = 0; p < 10; p++) for (int r = 0; r < 10; r++) tab[NVX_FSM_TIMING_BASE + p * 10 + r] = nvx_fsm_timing_entry(p, r);
        nvx_fsm_regs r = { 0, NVX_FSM_UNSYNCED, 0, -1 };
        unsigned y = 99; unsigned long bits = 0;
        for (int m = 0; m < 200000; m++) {
            y = y * 1664525u + 1013904223u;
            unsigned w = ((y >> 8) & 0x1ffu) | ((m < 60 ? 15u : (y >> 20) % 9u) << 12);
            int n; bits += nvx_fsm_period(tab, w, &r, &n) & ((1u << n) - 1u); bits += (unsigned long)n;
        }
        if (!bits) return 7;
    }
    printf("sanitize ok: %lu messages, %lu trace bytes\n", n_msgs, n_trace);
    return 0;
}
