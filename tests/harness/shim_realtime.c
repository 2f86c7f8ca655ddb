/* The reference-shaped surface at the REAL rate (TEST INFRASTRUCTURE): a receiver's main loop as receiver/capt_sched.c has
 * it -- the consumer wakes every 50 ms (capt_sched.c:486) and hands what has arrived since to the DSP sample by sample
 * through sample_in_1 (capt_sched.c:509-513) -- fed from a file of int16 IQ at 252 kS/s that "arrives" at 252 kS/s of wall
 * clock.  Links against libnavtex_amd.so in place of the reference's DSP objects; its own add_message notes when every
 * message arrived.  Output: "FED frame ms" (when the sample_in_1 call of a frame's last sample was made), "MSG ms freq|bbbb",
 * "LAT frames p50 p99 max" (nvx_shim_latency), times in ms since the start. */
#define _DEFAULT_SOURCE
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

void init_fir_filter1();
void sample_in_1(double sample_I, double sample_Q);
void init_fir2_wrapper();
int nvx_shim_latency(uint64_t *frames, double *p50_ms, double *p99_ms, double *max_ms, double *last_ms, int reset);

static double t0;
static double now_ms(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }

int add_message(char *bbbb, char *message, int freq)
{
    (void)message;
    printf("MSG %.3f %d|%s\n", now_ms() - t0, freq, bbbb);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    FILE *in = fopen(argv[1], "rb");
    if (!in) return 2;
    fseek(in, 0, SEEK_END); const long bytes = ftell(in); fseek(in, 0, SEEK_SET);
    const size_t n = (size_t)bytes / 4;
    short *iq = malloc((size_t)bytes);
    if (fread(iq, 4, n, in) != n) return 2;
    fclose(in);
    const double rate = 252000.0, tick_ms = 50.0;
    const size_t frame = 80640;
    init_fir_filter1();
    init_fir2_wrapper();
    t0 = now_ms();
    size_t pos = 0;
    double burst_max = 0.0, burst_sum = 0.0; int bursts = 0;
    for (int tick = 1; pos < n; tick++) {
        const double due = t0 + tick * tick_ms;
        double d = due - now_ms();
        if (d > 0) { struct timespec ts = { (time_t)(d / 1e3), (long)((d - 1e3 * (long)(d / 1e3)) * 1e6) }; nanosleep(&ts, NULL); }
        size_t avail = (size_t)((now_ms() - t0) * rate / 1e3);
        if (avail > n) avail = n;
        const double b0 = now_ms();
        for (; pos < avail; pos++) {
            if ((pos + 1) % frame == 0) printf("FED %zu %.3f\n", (pos + 1) / frame - 1, now_ms() - t0);
            sample_in_1((double)iq[2 * pos], (double)iq[2 * pos + 1]);
        }
        const double b = now_ms() - b0;
        burst_sum += b; bursts++; if (b > burst_max) burst_max = b;
    }
    printf("BURST %d %.3f %.3f\n", bursts, burst_sum / bursts, burst_max);      /* how long handing over 50 ms of samples took: mean, max */
    usleep(300000);                               /* an unmodified capt_sched.c never flushes: the library's housekeeping delivers */
    uint64_t frames = 0; double p50 = -1, p99 = -1, mx = -1, last = -1;
    if (nvx_shim_latency(&frames, &p50, &p99, &mx, &last, 0) != 0) return 3;
    printf("LAT %llu %.3f %.3f %.3f\n", (unsigned long long)frames, p50, p99, mx);
    free(iq);
    return 0;
}
