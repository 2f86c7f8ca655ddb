// valu_probe3.hip -- is SDWA half-word addition cheaper than v_dot2c_i32_i16 for stage 0?
// Times (a) bare chains of v_add_u32_sdwa, (b) the whole stage-0 block of the cascade kernel
// in its v_dot2c form and in an SDWA pair-add form, 1 and 4 waves per SIMD on every CU.
// build: hipcc --offload-arch=gfx950 -O3 tools/valu_probe3.hip -o tools/_bin/valu_probe3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef short s2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ s2 as_s2(unsigned v) { return __builtin_bit_cast(s2, v); }

__device__ __forceinline__ int sdwa_acc_lo(int acc, unsigned w)
{
    asm("v_add_u32_sdwa %0, sext(%1), %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD" : "+v"(acc) : "v"(w));
    return acc;
}
__device__ __forceinline__ int sdwa_lo_lo(unsigned a, unsigned b)
{
    int r;
    asm("v_add_u32_sdwa %0, sext(%1), sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_0" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ int sdwa_hi_hi(unsigned a, unsigned b)
{
    int r;
    asm("v_add_u32_sdwa %0, sext(%1), sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ double block_dot2c(u32x4 v, s2 sel_mine, s2 sel_other)
{
    int mine = 2, other = 2;
    mine  = __builtin_amdgcn_sdot2(as_s2(v.x), sel_mine,  mine,  false);
    other = __builtin_amdgcn_sdot2(as_s2(v.x), sel_other, other, false);
    mine  = __builtin_amdgcn_sdot2(as_s2(v.y), sel_mine,  mine,  false);
    other = __builtin_amdgcn_sdot2(as_s2(v.y), sel_other, other, false);
    mine  = __builtin_amdgcn_sdot2(as_s2(v.z), sel_mine,  mine,  false);
    other = __builtin_amdgcn_sdot2(as_s2(v.z), sel_other, other, false);
    mine  = __builtin_amdgcn_sdot2(as_s2(v.w), sel_mine,  mine,  false);
    other = __builtin_amdgcn_sdot2(as_s2(v.w), sel_other, other, false);
    const int tot = mine + __builtin_amdgcn_mov_dpp(other, 0xB1, 0xF, 0xF, true);
    return (double)(tot >> 3);
}

__device__ __forceinline__ double block_sdwa(u32x4 v, bool odd)
{
    const int sI = sdwa_lo_lo(v.x, v.y) + sdwa_lo_lo(v.z, v.w);
    const int sQ = sdwa_hi_hi(v.x, v.y) + sdwa_hi_hi(v.z, v.w);
    const int x = odd ? sQ : sI, y = odd ? sI : sQ;
    const int tot = x + __builtin_amdgcn_mov_dpp(y, 0xB1, 0xF, 0xF, true) + 4;
    return (double)(tot >> 3);
}

template <int OP>
__global__ void probe(double *out, const u32x4 *in, int iters)
{
    u32x4 v = in[threadIdx.x & 63];
    const bool odd = threadIdx.x & 1;
    const s2 sel_mine = odd ? s2{0, 1} : s2{1, 0}, sel_other = odd ? s2{1, 0} : s2{0, 1};
    int a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    double acc = 0.0;
    for (int k = 0; k < iters; k++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (OP == 0) {
                a0 = sdwa_acc_lo(a0, v.x); a1 = sdwa_acc_lo(a1, v.y); a2 = sdwa_acc_lo(a2, v.z); a3 = sdwa_acc_lo(a3, v.w);
                a4 = sdwa_acc_lo(a4, v.x); a5 = sdwa_acc_lo(a5, v.y); a6 = sdwa_acc_lo(a6, v.z); a7 = sdwa_acc_lo(a7, v.w);
            }
            if (OP == 1) {
                a0 = __builtin_amdgcn_sdot2(as_s2(v.x), sel_mine, a0, false); a1 = __builtin_amdgcn_sdot2(as_s2(v.y), sel_mine, a1, false);
                a2 = __builtin_amdgcn_sdot2(as_s2(v.z), sel_mine, a2, false); a3 = __builtin_amdgcn_sdot2(as_s2(v.w), sel_mine, a3, false);
                a4 = __builtin_amdgcn_sdot2(as_s2(v.x), sel_other, a4, false); a5 = __builtin_amdgcn_sdot2(as_s2(v.y), sel_other, a5, false);
                a6 = __builtin_amdgcn_sdot2(as_s2(v.z), sel_other, a6, false); a7 = __builtin_amdgcn_sdot2(as_s2(v.w), sel_other, a7, false);
            }
            if (OP == 2) { const double d = block_dot2c(v, sel_mine, sel_other); v.x += (unsigned)(long long)d; v.y ^= v.x; acc += d; }
            if (OP == 3) { const double d = block_sdwa(v, odd);                 v.x += (unsigned)(long long)d; v.y ^= v.x; acc += d; }
        }
    }
    out[1 + blockIdx.x * blockDim.x + threadIdx.x] = acc + a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + v.x;
}

__global__ void check(double *out, const u32x4 *in)
{
    const u32x4 v = in[threadIdx.x];
    const bool odd = threadIdx.x & 1;
    const s2 sel_mine = odd ? s2{0, 1} : s2{1, 0}, sel_other = odd ? s2{1, 0} : s2{0, 1};
    out[threadIdx.x] = block_dot2c(v, sel_mine, sel_other);
    out[64 + threadIdx.x] = block_sdwa(v, odd);
}

template <int OP> void run(const char *name, int unitsPerIter, double *out, const u32x4 *in, int threads)
{
    const int iters = 4000, blocks = 256;
    hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(threads), 0, 0, out, in, iters);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(threads), 0, 0, out, in, iters);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double n = (double)iters * 16 * unitsPerIter, wavesPerSimd = threads / 256.0;
    printf("%-34s %d wave(s)/SIMD: %7.3f ms -> %.3f ns per unit per SIMD\n", name, (int)wavesPerSimd, ms, ms * 1e6 / (n * wavesPerSimd));
}

int main()
{
    double *out; u32x4 *in;
    CK(hipMalloc(&out, (1 + 256 * 1024) * sizeof(double)));
    CK(hipMalloc(&in, 64 * sizeof(u32x4)));
    unsigned h[256]; for (int i = 0; i < 256; i++) h[i] = 0x12345u * (i + 1) ^ 0xdeadbeefu;
    CK(hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice));
    // parity of the two block forms first: random words plus the extremes
    {
        unsigned hv[256]; unsigned x = 12345u; int bad = 0;
        for (int rep = 0; rep < 200; rep++) {
            for (int i = 0; i < 256; i++) { x = x * 1664525u + 1013904223u; hv[i] = rep == 0 ? 0x80008000u : rep == 1 ? 0x7fff7fffu : rep == 2 ? 0x80007fffu : x; }
            CK(hipMemcpy(in, hv, sizeof hv, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, 0, out, in);
            double r[128]; CK(hipMemcpy(r, out, sizeof r, hipMemcpyDeviceToHost));
            for (int l = 0; l < 64; l++) {
                long long want = 4;                                  // host restatement: (sum of 8 + 4) >> 3, floor
                for (int q = 0; q < 8; q++) { unsigned w = hv[4 * (l & ~1) + q]; want += (l & 1) ? (short)(w >> 16) : (short)(w & 0xffff); }
                want = want >> 3;
                if (r[l] != (double)want || r[64 + l] != (double)want) bad++;
            }
        }
        printf("stage-0 block forms vs host restatement: %d mismatches\n", bad);
        CK(hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice));
    }
    for (int threads : {256, 1024}) {
        run<0>("v_add_u32_sdwa (instr)", 8, out, in, threads);
        run<1>("v_dot2c_i32_i16 (instr)", 8, out, in, threads);
        run<2>("stage-0 block, dot2c form (block)", 1, out, in, threads);
        run<3>("stage-0 block, sdwa form (block)", 1, out, in, threads);
    }
    return 0;
}
