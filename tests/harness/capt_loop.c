/* Drop-in link test: a stand-in for the receiver's main program.
 *
 * What it keeps from the reference's capture layer (receiver/capt_sched.c) is the
 * CONTRACT around the hot path, not the code:
 *   - the DSP is reached only through init_fir_filter1(), init_fir2_wrapper() and
 *     sample_in_1(I, Q) with each int16 widened to double (capt_sched.c:17-19, :511, :554, :612);
 *   - samples arrive as planar xi[] / xq[] callbacks of varying length and are kept
 *     interleaved in a ring whose cursor names the LAST WRITTEN slot (capt_sched.c:105-148, :443-446);
 *   - a consumer drains the ring in contiguous spans, splitting at the wrap (capt_sched.c:484-528);
 *   - messages leave through the program's own add_message() (message_store.c:59-97).
 * It links against libnavtex_amd.so in place of the reference's DSP objects.
 * Input: interleaved int16 IQ at 252 kS/s.  Output: "freq|bbbb|text" per message, '\n' escaped. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

void init_fir_filter1();
void sample_in_1(double sample_I, double sample_Q);
void init_fir2_wrapper();
int nvx_shim_finish(void);         /* only so that a finite file can be finished: its last, partial frame at its true length */

int add_message(char *bbbb, char *message, int freq)
{
    printf("%d|%s|", freq, bbbb);
    for (const char *c = message; *c; ++c)
        if (*c == '\n') fputs("\\n", stdout); else fputc(*c, stdout);
    fputc('\n', stdout);
    return 0;
}

typedef struct {
    short *slot;                   /* I,Q,I,Q,... */
    unsigned n_slots;              /* even */
    unsigned written;              /* index of the newest valid slot */
    unsigned consumed;             /* index of the newest slot already handed to the DSP */
} iq_ring;

static void ring_put(iq_ring *r, const short *xi, const short *xq, unsigned count)
{
    unsigned w = r->written;
    for (unsigned k = 0; k < count; ++k) {
        w = (w + 1) % r->n_slots; r->slot[w] = xi[k];
        w = (w + 1) % r->n_slots; r->slot[w] = xq[k];
    }
    r->written = w;
}

static void ring_drain(iq_ring *r)
{
    const unsigned target = r->written;
    while (r->consumed != target) {
        unsigned first = (r->consumed + 1) % r->n_slots;
        unsigned last = (first <= target) ? target : r->n_slots - 1;     /* stop at the wrap, come round again */
        for (unsigned s = first; s < last; s += 2)
            sample_in_1((double)r->slot[s], (double)r->slot[s + 1]);
        r->consumed = last;
    }
}

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    FILE *in = fopen(argv[1], "rb");
    if (!in) return 2;

    init_fir_filter1();
    init_fir2_wrapper();

    iq_ring ring;
    ring.n_slots = 252000u * 2u * 2u;                 /* two seconds (the reference keeps eight) */
    ring.slot = malloc(ring.n_slots * sizeof(short));
    ring.written = ring.consumed = 1;                 /* odd start: I lands on even slots, as in the reference */

    enum { MAX_CB = 1008 };                           /* callbacks of about a thousand samples, length jittered */
    short pairs[2 * MAX_CB], xi[MAX_CB], xq[MAX_CB];
    unsigned lcg = 12345;
    for (;;) {
        lcg = lcg * 1103515245u + 12345u;
        size_t want = 1 + (lcg >> 16) % MAX_CB;
        size_t got = fread(pairs, 2 * sizeof(short), want, in);
        if (!got) break;
        for (size_t k = 0; k < got; ++k) { xi[k] = pairs[2 * k]; xq[k] = pairs[2 * k + 1]; }
        ring_put(&ring, xi, xq, (unsigned)got);
        ring_drain(&ring);
    }
    fclose(in);
    free(ring.slot);
    /* "noflush": what an unmodified capt_sched.c does -- it never flushes.  The messages of every launched frame must
     * still reach add_message (the library's housekeeping takes finished work in: 2 ms after a launch went out, every 50 ms otherwise); give it a moment. */
    if (argc > 2 && !strcmp(argv[2], "noflush")) { usleep(1500000); return 0; }
    return nvx_shim_finish() == 0 ? 0 : 1;
}
