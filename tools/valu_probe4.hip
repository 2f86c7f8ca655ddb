// valu_probe4.hip -- issue cost of the individual instructions in the cascade kernel's pass loop,
// exact encodings pinned with inline asm; 4 waves per SIMD on every CU, 8 independent chains.
// build: hipcc --offload-arch=gfx950 -O3 tools/valu_probe4.hip -o tools/_bin/valu_probe4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)

template <int OP>
__global__ void probe(double *out, int iters)
{
    int i[8]; double d[8];
    for (int k = 0; k < 8; k++) { i[k] = threadIdx.x + k; d[k] = threadIdx.x * 1e-3 + k + 1.0; }
    const unsigned long long mask = 0x5555555555555555ull;     // odd/even lane mask in an SGPR pair
    const double c = out[0];
    int s = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
#define S_CND64(k)  asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(i[k]) : "v"(i[(k + 1) & 7]), "s"(mask));
#define S_CND32(k)  asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(i[k]) : "v"(i[(k + 1) & 7]) : );
#define S_CVT(k)    asm volatile("v_cvt_f64_i32_e32 %0, %1" : "=v"(d[k]) : "v"(i[k]));
#define S_ADDDPP(k) asm volatile("s_nop 1\n v_add_u32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(i[k]) : "v"(i[(k + 4) & 7]));
#define S_ADD3(k)   asm volatile("v_add3_u32 %0, %0, %1, 4" : "+v"(i[k]) : "v"(i[(k + 1) & 7]));
#define S_ASHR(k)   asm volatile("v_ashrrev_i32_e32 %0, 3, %0" : "+v"(i[k]));
#define S_ADD(k)    asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(i[k]) : "v"(i[(k + 1) & 7]));
#define S_RDLANE(k) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s) : "v"(i[k]));
#define S_WRLANE(k) asm volatile("v_writelane_b32 %0, %1, 3" : "+v"(i[k]) : "s"(s));
#define S_MULS(k)   asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[k]) : "s"(c));
#define S_MULV(k)   asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[k]) : "v"(c));
#define S_ADDF(k)   asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[k]) : "v"(d[(k + 1) & 7]));
#define S_XORL(k)   asm volatile("v_xor_b32_e32 %0, 0x80008000, %0" : "+v"(i[k]));
#define S_PERM(k)   asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(i[k]) : "v"(i[(k + 1) & 7]), "v"(i[(k + 2) & 7]));
#define S_LDSW(k)   asm volatile("ds_write_b64 %0, %1" :: "v"((threadIdx.x & 63) * 8), "v"(d[k]) : "memory");
#define S_LDSR(k)   asm volatile("ds_read_b128 %0, %1" : "=v"(q[k & 1]) : "v"((threadIdx.x & 63) * 16) : "memory");
            if (OP == 0) { REP8(S_CND64) }
            if (OP == 1) { REP8(S_CND32) }
            if (OP == 2) { REP8(S_CVT) }
            if (OP == 3) { REP8(S_ADDDPP) }
            if (OP == 4) { REP8(S_ADD3) }
            if (OP == 5) { REP8(S_ASHR) }
            if (OP == 6) { REP8(S_ADD) }
            if (OP == 7) { REP8(S_RDLANE) }
            if (OP == 8) { REP8(S_WRLANE) }
            if (OP == 9) { REP8(S_MULS) }
            if (OP == 10) { REP8(S_MULV) }
            if (OP == 11) { REP8(S_ADDF) }
            if (OP == 12) { REP8(S_XORL) }
            if (OP == 13) { REP8(S_PERM) }
            if (OP == 14) { REP8(S_LDSW) }
        }
    }
    double r = s;
    for (int k = 0; k < 8; k++) r += d[k] + i[k];
    out[1 + blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OP> void run(const char *name, double *out, int threads)
{
    const int iters = 2000, blocks = 256;
    hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(threads), 4096, 0, out, iters);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(threads), 4096, 0, out, iters);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double n = (double)iters * 16 * 8, wavesPerSimd = threads / 256.0;
    printf("%-44s %d wave(s)/SIMD: %7.3f ms -> %.3f ns per instr per SIMD\n", name, (int)wavesPerSimd, ms, ms * 1e6 / (n * wavesPerSimd));
}

int main()
{
    double *out;
    CK(hipMalloc(&out, (1 + 256 * 1024) * sizeof(double)));
    double one = 1.0000001; CK(hipMemcpy(out, &one, 8, hipMemcpyHostToDevice));
    const int threads = 1024;
    run<6>("v_add_u32_e32 (reference: VOP2)", out, threads);
    run<11>("v_add_f64 vgpr,vgpr (reference: fp64)", out, threads);
    run<0>("v_cndmask_b32_e64, mask in SGPR pair", out, threads);
    run<1>("v_cndmask_b32_e32, mask in vcc", out, threads);
    run<2>("v_cvt_f64_i32_e32", out, threads);
    run<3>("s_nop 1 + v_add_u32_dpp quad_perm", out, threads);
    run<4>("v_add3_u32", out, threads);
    run<5>("v_ashrrev_i32_e32", out, threads);
    run<7>("v_readlane_b32", out, threads);
    run<8>("v_writelane_b32", out, threads);
    run<9>("v_mul_f64 vgpr, sgpr pair", out, threads);
    run<10>("v_mul_f64 vgpr, vgpr", out, threads);
    run<12>("v_xor_b32_e32 with 32-bit literal", out, threads);
    run<13>("v_perm_b32", out, threads);
    run<14>("ds_write_b64 (conflict-free)", out, threads);
    return 0;
}
