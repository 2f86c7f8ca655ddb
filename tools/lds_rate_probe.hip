// lds_rate_probe.hip -- how many LDS wave-instructions per cycle does a CU accept?  (MI355X, gfx950)
// Every workgroup is one wave; W waves per CU (grid = 256 * W, LDS footprint sized so that exactly W fit); a wave issues
// back-to-back conflict-free LDS reads of one kind (lanes read consecutive words: 512 B per ds_read_b64, 1 KiB per
// ds_read_b128), 12 in flight, optionally with fp64 multiplies in between (4 per read: the FIR1 ratio).
//   lds_rate_probe            prints ns per wave-instruction per CU and the implied cycles at the clock the run held
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((address_space(3))) volatile double lds_vdouble;
typedef double d2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) volatile d2 lds_vd2;

template <int KIND, int FP64_PER_READ, int LDS_BYTES>
__global__ __launch_bounds__(64) void k(double *out, int iters)
{
    __shared__ double lds[LDS_BYTES / 8];
    const int lane = threadIdx.x;
    for (int i = lane; i < LDS_BYTES / 8; i += 64) lds[i] = i * 0.5;
    double acc0 = 1.0, acc1 = 2.0, s = 0.0;
    const lds_vdouble *p64 = (const lds_vdouble *)&lds[lane];
    const lds_vd2 *p128 = (const lds_vd2 *)&lds[2 * lane];
    typedef __attribute__((address_space(3))) volatile short lds_vshort;
    const lds_vshort *p16 = (const lds_vshort *)((const short *)lds + lane);     // KIND 2: a lane pair shares a dword (packed int16 I | Q)
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 12; j++) {
            double v;
            if (KIND == 0) v = p64[64 * (j % 8)];
            else if (KIND == 2) v = (double)(int)p16[128 * (j % 8)];          // ds_read_i16 + v_cvt_f64_i32
            else { d2 t = p128[64 * (j % 4)]; v = t.x + 0 * t.y; }
            s += v;                                   // one dependent add per read keeps the read alive
#pragma unroll
            for (int f = 0; f < FP64_PER_READ; f++) { if (f & 1) acc1 = acc1 * 0.9999999; else acc0 = acc0 * 1.0000001; }
        }
    }
    if (s + acc0 + acc1 == 12345.678) out[blockIdx.x] = s;
}

template <int KIND, int FP, int LDS_BYTES> static void run(const char *name, int waves, double *d)
{
    const int iters = 20000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<KIND, FP, LDS_BYTES>), dim3(256 * waves), dim3(64), 0, 0, d, 200);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k<KIND, FP, LDS_BYTES>), dim3(256 * waves), dim3(64), 0, 0, d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double reads_per_cu = (double)waves * iters * 12;
    const double cyc = ms * 1e6 / reads_per_cu * 2.1, valu = (FP + 1) * 4.0 / 4.0;      // VALU-bound cycles per read per CU: 4 cycles per fp64, 4 SIMDs
    printf("%-52s %2d waves/CU: %8.3f ms  %5.2f cycles per read per CU at 2.1 GHz; fp64 issue alone %5.2f -> %3.0f %%\n", name, waves, ms, cyc, valu, 100.0 * valu / cyc);
    fflush(stdout);
}

int main()
{
    double *d; hipMalloc(&d, 1 << 20);
    // 13.6 KB per wave -> 11 waves per CU (the cascade's residency); 40 KB -> 4 (one per SIMD)
    // fp64 operations per read = FP64_PER_READ multiplies + the one add that consumes the value
    run<0, 0, 13888>("ds_read_b64 + 1 fp64", 11, d);
    run<1, 0, 13888>("ds_read_b128 + 1 fp64", 11, d);
    run<0, 2, 13888>("ds_read_b64 + 3 fp64", 11, d);
    run<0, 3, 13888>("ds_read_b64 + 4 fp64 (FIR1 now: 3.75)", 11, d);
    run<0, 4, 13888>("ds_read_b64 + 5 fp64 (three outputs per lane: 4.9)", 11, d);
    run<0, 6, 13888>("ds_read_b64 + 7 fp64 (four outputs per lane: 7.3)", 11, d);
    run<0, 9, 13888>("ds_read_b64 + 10 fp64", 11, d);
    run<1, 3, 13888>("ds_read_b128 + 4 fp64 (round 1: 4)", 11, d);
    run<1, 9, 13888>("ds_read_b128 + 10 fp64", 11, d);
    run<2, 0, 13888>("ds_read_i16 + cvt + 1 fp64", 11, d);
    run<2, 3, 13888>("ds_read_i16 + cvt + 4 fp64 (int16 window)", 11, d);
    run<2, 3, 10000>("ds_read_i16 + cvt + 4 fp64 (int16 window)", 15, d);
    run<0, 3, 10000>("ds_read_b64 + 4 fp64", 16, d);
    run<0, 3, 20000>("ds_read_b64 + 4 fp64", 8, d);
    run<0, 3, 40000>("ds_read_b64 + 4 fp64", 4, d);
    return 0;
}
