// valu_probe.hip -- issue cost of the instructions the cascade kernel uses:
// long unrolled independent chains, 1 / 2 / 4 waves per SIMD on every CU, wall
// time -> ns per wave-instruction per SIMD (multiply by the clock for cycles).
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/valu_probe.hip -o tools/_bin/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef short s2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ s2 as_s2(int v) { return __builtin_bit_cast(s2, v); }

template <int OP>
__global__ void probe(double *out, int iters)
{
    double a0 = threadIdx.x * 1e-3 + 1.0, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double c = out[0];            // runtime constant
    int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3;
    float f0 = threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;
    const s2 sel = { 1, 0 };
    for (int k = 0; k < iters; k++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (OP == 0) { a0 *= c; a1 *= c; a2 *= c; a3 *= c; a4 *= c; a5 *= c; a6 *= c; a7 *= c; }
            if (OP == 1) { a0 += c; a1 += c; a2 += c; a3 += c; a4 += c; a5 += c; a6 += c; a7 += c; }
            if (OP == 2) { a0 = __builtin_fma(a0, c, c); a1 = __builtin_fma(a1, c, c); a2 = __builtin_fma(a2, c, c); a3 = __builtin_fma(a3, c, c);
                           a4 = __builtin_fma(a4, c, c); a5 = __builtin_fma(a5, c, c); a6 = __builtin_fma(a6, c, c); a7 = __builtin_fma(a7, c, c); }
            if (OP == 3) { i0 = __builtin_amdgcn_sdot2(as_s2(i1), sel, i0, false); i1 = __builtin_amdgcn_sdot2(as_s2(i2), sel, i1, false);
                           i2 = __builtin_amdgcn_sdot2(as_s2(i3), sel, i2, false); i3 = __builtin_amdgcn_sdot2(as_s2(i0), sel, i3, false);
                           i0 = __builtin_amdgcn_sdot2(as_s2(i1), sel, i0, false); i1 = __builtin_amdgcn_sdot2(as_s2(i2), sel, i1, false);
                           i2 = __builtin_amdgcn_sdot2(as_s2(i3), sel, i2, false); i3 = __builtin_amdgcn_sdot2(as_s2(i0), sel, i3, false); }
            if (OP == 4) { f0 += f1; f1 += f2; f2 += f3; f3 += f0; f0 += f1; f1 += f2; f2 += f3; f3 += f0; }
            if (OP == 5) { a0 = (double)i0; a1 = (double)i1; a2 = (double)i2; a3 = (double)i3; i0 += (int)(__double_as_longlong(a0) & 1); i1 ^= i0; i2 ^= i1; i3 ^= i2;
                           a4 = (double)i0; a5 = (double)i1; a6 = (double)i2; a7 = (double)i3; }
            if (OP == 6) { i0 += i1; i1 ^= i2; i2 += i3; i3 ^= i0; i0 += i1; i1 ^= i2; i2 += i3; i3 ^= i0; }
        }
    }
    out[1 + blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + i0 + i1 + i2 + i3 + f0 + f1 + f2 + f3;
}

template <int OP> void run(const char *name, int opsPerIter, double *out, int threads)
{
    const int iters = 4000, blocks = 256;
    hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(threads), 0, 0, out, iters);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(threads), 0, 0, out, iters);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    double n = (double)iters * 16 * opsPerIter;
    double wavesPerSimd = threads / 256.0;
    printf("%-24s %d wave(s)/SIMD: %7.3f ms -> %.3f ns per wave-instr per SIMD\n", name, (int)wavesPerSimd, ms, ms * 1e6 / (n * wavesPerSimd));
}

int main()
{
    double *out;
    CK(hipMalloc(&out, (1 + 256 * 1024) * sizeof(double)));
    double one = 1.0000001; CK(hipMemcpy(out, &one, 8, hipMemcpyHostToDevice));
    for (int threads : {256, 512, 1024}) {
        run<4>("v_add_f32", 8, out, threads);
        run<6>("v_add/xor_u32", 8, out, threads);
        run<0>("v_mul_f64", 8, out, threads);
        run<1>("v_add_f64", 8, out, threads);
        run<2>("v_fma_f64", 8, out, threads);
        run<3>("v_dot2c_i32_i16", 8, out, threads);
        run<5>("8 cvt_f64_i32 + 4 int", 12, out, threads);
    }
    return 0;
}
