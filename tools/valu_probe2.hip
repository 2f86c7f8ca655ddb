// valu_probe2.hip -- candidate instructions for stage 0 (sign-extend + accumulate):
// v_mad_i32_i16 (op_sel picks the half), v_bfe_i32 + v_add3_u32, v_dot2c_i32_i16.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int OP>
__global__ void probe(int *out, int iters)
{
    int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    int w = out[0] + threadIdx.x * 65537;
    for (int k = 0; k < iters; k++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (OP == 0) {
                asm volatile("v_mad_i32_i16 %0, %1, 1, %0\n\tv_mad_i32_i16 %2, %1, 1, %2 op_sel:[1,0,0,0]\n\t"
                             "v_mad_i32_i16 %3, %1, 1, %3\n\tv_mad_i32_i16 %4, %1, 1, %4 op_sel:[1,0,0,0]\n\t"
                             "v_mad_i32_i16 %5, %1, 1, %5\n\tv_mad_i32_i16 %6, %1, 1, %6 op_sel:[1,0,0,0]\n\t"
                             "v_mad_i32_i16 %7, %1, 1, %7\n\tv_mad_i32_i16 %8, %1, 1, %8 op_sel:[1,0,0,0]"
                             : "+v"(a0), "+v"(w), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            }
            if (OP == 1) {
                asm volatile("v_bfe_i32 %0, %8, 0, 16\n\tv_ashrrev_i32 %1, 16, %8\n\tv_bfe_i32 %2, %8, 0, 16\n\tv_ashrrev_i32 %3, 16, %8\n\t"
                             "v_add3_u32 %4, %0, %2, %4\n\tv_add3_u32 %5, %1, %3, %5\n\tv_add3_u32 %6, %0, %2, %6\n\tv_add3_u32 %7, %1, %3, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w));
            }
            if (OP == 2) {
                asm volatile("v_dot2c_i32_i16 %0, 1, %8\n\tv_dot2c_i32_i16 %1, 0x10000, %8\n\tv_dot2c_i32_i16 %2, 1, %8\n\tv_dot2c_i32_i16 %3, 0x10000, %8\n\t"
                             "v_dot2c_i32_i16 %4, 1, %8\n\tv_dot2c_i32_i16 %5, 0x10000, %8\n\tv_dot2c_i32_i16 %6, 1, %8\n\tv_dot2c_i32_i16 %7, 0x10000, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w));
            }
            if (OP == 3) {
                asm volatile("v_pk_add_i16 %0, %0, %8\n\tv_pk_add_i16 %1, %1, %8\n\tv_pk_add_i16 %2, %2, %8\n\tv_pk_add_i16 %3, %3, %8\n\t"
                             "v_pk_add_i16 %4, %4, %8\n\tv_pk_add_i16 %5, %5, %8\n\tv_pk_add_i16 %6, %6, %8\n\tv_pk_add_i16 %7, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w));
            }
        }
    }
    out[1 + blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int OP> void run(const char *name, int *out, int threads)
{
    const int iters = 4000, blocks = 256;
    hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(threads), 0, 0, out, iters);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(threads), 0, 0, out, iters);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    double n = (double)iters * 16 * 8;
    printf("%-34s %d wave(s)/SIMD: %7.3f ms -> %.3f ns per wave-instr per SIMD\n", name, threads / 256, ms, ms * 1e6 / (n * (threads / 256.0)));
}

int main()
{
    int *out; CK(hipMalloc(&out, (1 + 256 * 1024) * sizeof(int))); CK(hipMemset(out, 0, 4));
    for (int threads : {256, 1024}) {
        run<0>("v_mad_i32_i16 (x1, op_sel)", out, threads);
        run<1>("bfe/ashr + add3 (8 ops)", out, threads);
        run<2>("v_dot2c_i32_i16", out, threads);
        run<3>("v_pk_add_i16", out, threads);
    }
    return 0;
}
