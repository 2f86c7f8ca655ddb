// icache_probe.hip -- does instruction fetch limit a CU when its waves run DIFFERENT parts of a long unrolled fp64 body?
// (MI355X, gfx950.)  Every workgroup is one wave; W waves per CU; a wave runs `iters` trips over a straight-line body of
// N independent fp64 multiplies / adds (8 accumulator chains, 8 bytes per instruction), entered at a wave-dependent
// offset so that the waves of a SIMD are spread over the body.  Prints the achieved share of the fp64 issue rate
// (4 cycles per wave-instruction per SIMD) for bodies of 0.5 / 4 / 16 / 32 / 64 / 128 KiB.
//   hipcc --offload-arch=gfx950 -O3 tools/icache_probe.hip -o tools/_bin/icache_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define OP8 \
    asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a0) : "v"(c)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(a1) : "v"(c)); \
    asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a2) : "v"(c)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(a3) : "v"(c)); \
    asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a4) : "v"(c)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(a5) : "v"(c)); \
    asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a6) : "v"(c)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(a7) : "v"(c));
#define OP64 OP8 OP8 OP8 OP8 OP8 OP8 OP8 OP8
#define OP512 OP64 OP64 OP64 OP64 OP64 OP64 OP64 OP64
#define OP2K OP512 OP512 OP512 OP512
#define OP8K OP2K OP2K OP2K OP2K

// BODY: number of 64-instruction blocks (512 B each)
template <int BLOCKS>
__global__ __launch_bounds__(64) void k(double *out, int iters, int spread)
{
    double a0 = 1.0, a1 = 2.0, a2 = 3.0, a3 = 4.0, a4 = 5.0, a5 = 6.0, a6 = 7.0, a7 = 8.0;
    const double c = 1.0000000001 + out[0];
    // desynchronise: wave b first runs (b * spread) % BLOCKS blocks' worth of a short loop
    int pre = spread ? (int)((blockIdx.x * (unsigned)spread) % (unsigned)(BLOCKS * 8)) : 0;
    for (int i = 0; i < pre; i++) { OP8 }
    for (int it = 0; it < iters; it++) {
        if (BLOCKS == 1) { OP64 }
        else if (BLOCKS == 8) { OP512 }
        else if (BLOCKS == 32) { OP2K }
        else if (BLOCKS == 64) { OP2K OP2K }
        else if (BLOCKS == 128) { OP8K }
        else if (BLOCKS == 256) { OP8K OP8K }
    }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678) out[blockIdx.x] = a0;
}

template <int BLOCKS> static void run(int waves, int spread, double *d)
{
    const long total = 1L << 22;                       // wave-instructions per wave, whatever the body
    const int iters = (int)(total / (BLOCKS * 64));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<BLOCKS>), dim3(256 * waves), dim3(64), 0, 0, d, 4, spread);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k<BLOCKS>), dim3(256 * waves), dim3(64), 0, 0, d, iters, spread);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // per SIMD: waves/4 waves x total instructions x 4 cycles
    const double cyc_ideal = (double)waves / 4.0 * (double)iters * BLOCKS * 64 * 4.0;
    const double ghz_if_ideal = cyc_ideal / (ms * 1e6);
    printf("body %6.1f KiB  %2d waves/CU  spread %3d: %8.3f ms  -> %5.2f G wave-cycles/s per SIMD (= clock if the fp64 pipe never idles)\n",
           BLOCKS * 0.5, waves, spread, ms, ghz_if_ideal);
    fflush(stdout);
}

int main()
{
    double *d; hipMalloc(&d, 1 << 22); hipMemset(d, 0, 1 << 22);
    for (int waves : { 4, 12 }) {
        for (int spread : { 0, 37 }) {
            run<1>(waves, spread, d);
            run<8>(waves, spread, d);
            run<32>(waves, spread, d);
            run<64>(waves, spread, d);
            run<128>(waves, spread, d);
            run<256>(waves, spread, d);
        }
    }
    return 0;
}
