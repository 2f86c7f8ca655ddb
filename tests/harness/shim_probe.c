/* Probes of the reference-shaped surface (include/navtex_amd.h section A) where it deliberately does NOT behave like the
 * reference -- DESIGN.md section 4.4, tests/test_gpu_deviations.py.  Linked against libnavtex_amd.so alone; needs a GPU.
 *
 *   shim_probe reinit   <iq.bin> <n>   samples [0, n) | init_fir_filter1(); init_fir2_wrapper(); | samples [n, end) | finish
 *   shim_probe domain   <iq.bin> <k>   every k-th sample handed over as value + 0.25 / value - 0.25 (I / Q), then finish
 *   shim_probe refinish <iq.bin>       the whole file, finish, the whole file again WITHOUT an init, finish
 *
 * Input: interleaved int16 IQ at 252 kS/s.  Output, one line each:
 *   bits518 <B/Y...>   bits490 <B/Y...>   (the singleton's bits at the end; `refinish` prints them after each finish)
 *   stats <sample_in_1 calls> <calls outside the input domain>
 *   msg <freq>|<bbbb>                      per add_message call, in order
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

void init_fir_filter1();
void sample_in_1(double sample_I, double sample_Q);
void init_fir2_wrapper();
int nvx_shim_finish(void);
size_t nvx_shim_bits(int chain, char *out, size_t cap);
int nvx_shim_stats(uint64_t *samples, uint64_t *off_domain);
const char *nvx_last_error(void);

int add_message(char *bbbb, char *message, int freq) { (void)message; printf("msg %d|%s\n", freq, bbbb); return 0; }

static void print_bits(void)
{
    static char buf[1 << 20];
    for (int c = 0; c < 2; c++) {
        size_t n = nvx_shim_bits(c, buf, sizeof buf - 1);
        buf[n] = 0;
        printf("bits%d %s\n", c == 0 ? 518 : 490, buf);
    }
}

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    FILE *f = fopen(argv[2], "rb");
    if (!f) return 2;
    fseek(f, 0, SEEK_END);
    const size_t n = (size_t)ftell(f) / 4;
    fseek(f, 0, SEEK_SET);
    short *iq = malloc(n * 4 + 4);
    if (fread(iq, 4, n, f) != n) return 2;
    fclose(f);
    const size_t arg = argc > 3 ? (size_t)strtoull(argv[3], NULL, 10) : 0;

    init_fir_filter1();
    init_fir2_wrapper();
    if (!strcmp(argv[1], "reinit")) {
        for (size_t k = 0; k < n; k++) {
            if (k == arg) { init_fir_filter1(); init_fir2_wrapper(); }
            sample_in_1((double)iq[2 * k], (double)iq[2 * k + 1]);
        }
        if (nvx_shim_finish()) { fprintf(stderr, "finish: %s\n", nvx_last_error()); return 1; }
        print_bits();
    } else if (!strcmp(argv[1], "domain")) {
        for (size_t k = 0; k < n; k++) {
            const double d = (arg && k % arg == 0) ? 0.25 : 0.0;
            sample_in_1((double)iq[2 * k] + d, (double)iq[2 * k + 1] - d);
        }
        if (nvx_shim_finish()) { fprintf(stderr, "finish: %s\n", nvx_last_error()); return 1; }
        print_bits();
    } else if (!strcmp(argv[1], "refinish")) {
        for (int round = 0; round < 2; round++) {
            for (size_t k = 0; k < n; k++) sample_in_1((double)iq[2 * k], (double)iq[2 * k + 1]);
            if (nvx_shim_finish()) { fprintf(stderr, "finish: %s\n", nvx_last_error()); return 1; }
            print_bits();
        }
    } else return 2;
    uint64_t calls = 0, off = 0;
    nvx_shim_stats(&calls, &off);
    printf("stats %llu %llu\n", (unsigned long long)calls, (unsigned long long)off);
    return 0;
}
