/* Drop-in link test.  A stand-in for the reference's receiver main program that
 * keeps exactly the shape receiver/capt_sched.c has around the hot path:
 *   - the three extern declarations (capt_sched.c:17-19),
 *   - init_dsp() -> init_fir_filter1(), then init_fir2_wrapper() (:552-555, :612),
 *   - an interleaved int16 ring filled by a producer shaped like StreamACallback
 *     (:105-148, planar xi/xq -> I,Q,I,Q... after in_idx, index = last written),
 *   - the consumer loop (:484-528): spans [from..to] with wrap, and for each pair
 *     sample_in_1((double)buf[i], (double)buf[i+1]),
 *   - its own add_message() (as message_store.o provides, message_store.c:59-97).
 * It links against libnavtex_amd.so INSTEAD of fir1cpp/fir2cpp/fir3cpp/decoder/
 * nav_b_sm/nav_sched objects.  Input: interleaved int16 IQ file at 252 kS/s.
 * Output: one line per message on stdout: freq|bbbb|message with \n escaped.   */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

void init_fir_filter1();
void sample_in_1(double sample_I, double sample_Q);
void init_fir2_wrapper();
int nvx_shim_flush(void);          /* only so the test can terminate a finite file */

static short *sample_buffer;
static unsigned int s_buffer_size;
static int in_idx, out_idx;

int add_message(char *bbbb, char *message, int freq)
{
    printf("%d|%s|", freq, bbbb);
    for (char *p = message; *p; p++) { if (*p == '\n') fputs("\\n", stdout); else fputc(*p, stdout); }
    fputc('\n', stdout);
    return 0;
}

static void StreamACallback(short *xi, short *xq, void *params, unsigned int numSamples, unsigned int reset, void *cbContext)
{
    (void)params; (void)reset; (void)cbContext;
    unsigned int next_idx = (unsigned int)in_idx + 1;
    next_idx %= s_buffer_size;
    for (unsigned int i = 0; i < numSamples; i++) {
        sample_buffer[next_idx] = xi[i]; next_idx++; next_idx %= s_buffer_size;
        sample_buffer[next_idx] = xq[i]; next_idx++; next_idx %= s_buffer_size;
    }
    in_idx = (next_idx == 0) ? (int)s_buffer_size - 1 : (int)next_idx - 1;
}

static void consume(void)
{
    int c_in_idx = in_idx;
    while (out_idx != c_in_idx) {
        int from = out_idx + 1; from %= (int)s_buffer_size;
        int to = c_in_idx;
        if (from > to) to = (int)s_buffer_size - 1;
        int num_samples = to - from + 1, index = from;
        for (int i = 0; i < num_samples; i += 2) {
            sample_in_1((double)sample_buffer[index], (double)sample_buffer[index + 1]);
            index += 2;
        }
        out_idx = to;
    }
}

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    init_fir_filter1();                /* init_dsp() */
    init_fir2_wrapper();
    s_buffer_size = 252000 * 2 * 2;    /* 2 s ring (the reference keeps 8 s) */
    sample_buffer = malloc(s_buffer_size * sizeof(short));
    in_idx = 1; out_idx = 1;
    enum { CHUNK = 1008 };             /* the vendor library delivers ~1 k-sample callbacks */
    short iq[2 * CHUNK], xi[CHUNK], xq[CHUNK];
    size_t n;
    unsigned jitter = 12345;
    for (;;) {
        jitter = jitter * 1103515245u + 12345u;
        size_t want = 1 + (jitter >> 16) % CHUNK;           /* jittered numSamples */
        n = fread(iq, 4, want, f);
        if (!n) break;
        for (size_t i = 0; i < n; i++) { xi[i] = iq[2 * i]; xq[i] = iq[2 * i + 1]; }
        StreamACallback(xi, xq, NULL, (unsigned int)n, 0, NULL);
        consume();
    }
    fclose(f);
    nvx_shim_flush();
    return 0;
}
