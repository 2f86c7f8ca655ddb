// hbm_probe.hip -- read-bandwidth ceilings on this GPU for the access patterns
// the cascade kernel could use (guide rule: measure a known-good reference on
// the same hardware before calling anything a ceiling).
//   A  grid-stride, fully coalesced dwordx4 nt read of the whole buffer
//   B  the cascade's pattern: one 64-lane workgroup per stream, 8 x 1 KiB per
//      pass, streams [n_streams][pitch]; LDS padding sets the residency
// build: hipcc --offload-arch=gfx950 -O3 tools/hbm_probe.hip -o /tmp/hbm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void read_gridstride(const u32x4 *p, size_t n16, unsigned *sink)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    unsigned acc = 0;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        u32x4 a = __builtin_nontemporal_load(p + i), b = __builtin_nontemporal_load(p + i + stride);
        u32x4 c = __builtin_nontemporal_load(p + i + 2 * stride), d = __builtin_nontemporal_load(p + i + 3 * stride);
        acc += a.x ^ b.y ^ c.z ^ d.w;
    }
    for (; i < n16; i += stride) acc += __builtin_nontemporal_load(p + i).x;
    if (acc == 0x12345678u) *sink = acc;
}

template <int LDSB, int DEPTH>
__global__ __launch_bounds__(64) void read_streams(const u32x4 *p, size_t pitch16, int passes, unsigned *sink)
{
    __shared__ char pad[LDSB];
    const u32x4 *src = p + (size_t)blockIdx.x * pitch16 + threadIdx.x;
    unsigned acc = 0;
    u32x4 buf[DEPTH][8];
#pragma unroll
    for (int d = 0; d < DEPTH; d++)
#pragma unroll
        for (int j = 0; j < 8; j++) buf[d][j] = __builtin_nontemporal_load(src + (size_t)d * 512 + 64 * j);
    for (int q = 0; q < passes; q += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
#pragma unroll
            for (int j = 0; j < 8; j++) acc += buf[d][j].x ^ buf[d][j].w;
            if (q + d + DEPTH < passes) {
#pragma unroll
                for (int j = 0; j < 8; j++) buf[d][j] = __builtin_nontemporal_load(src + (size_t)(q + d + DEPTH) * 512 + 64 * j);
            }
        }
    }
    if (acc == 0x12345678u) { *sink = acc; pad[threadIdx.x] = 1; *sink += pad[(threadIdx.x + 1) & 63]; }
}

// burst variant: NB x 8 KiB contiguous requested back to back, then consumed, then the next burst
template <int LDSB, int NB>
__global__ __launch_bounds__(64) void read_streams_burst(const u32x4 *p, size_t pitch16, int passes, unsigned *sink)
{
    __shared__ char pad[LDSB];
    const u32x4 *src = p + (size_t)blockIdx.x * pitch16 + threadIdx.x;
    unsigned acc = 0;
    u32x4 buf[NB * 8];
    for (int q = 0; q + NB <= passes; q += NB) {
#pragma unroll
        for (int j = 0; j < NB * 8; j++) buf[j] = __builtin_nontemporal_load(src + (size_t)q * 512 + 64 * j);
#pragma unroll
        for (int j = 0; j < NB * 8; j++) acc += buf[j].x ^ buf[j].w;
    }
    if (acc == 0x12345678u) { *sink = acc; pad[threadIdx.x] = 1; *sink += pad[(threadIdx.x + 1) & 63]; }
}

template <typename F> static float time_ms(F f, int reps)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main(int argc, char **argv)
{
    int streams = argc > 1 ? atoi(argv[1]) : 4096;
    int frames = argc > 2 ? atoi(argv[2]) : 12;
    size_t per = (size_t)frames * 645120 * 4;           // bytes per stream
    size_t bytes = per * streams;
    void *d; unsigned *sink;
    CK(hipMalloc(&d, bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(d, 1, bytes));
    int passes = frames * 315;
    printf("streams %d frames %d bytes %.1f GB\n", streams, frames, bytes / 1e9);
    for (int blocks : {1024, 2048, 4096, 8192}) {
        float ms = time_ms([&] { hipLaunchKernelGGL(read_gridstride, dim3(blocks), dim3(256), 0, 0, (const u32x4 *)d, bytes / 16, sink); }, 3);
        printf("A grid-stride %5d blocks: %.2f ms  %.0f GB/s\n", blocks, ms, bytes / ms / 1e6);
    }
#define B(LDSB, DEPTH) { float ms = time_ms([&] { hipLaunchKernelGGL((read_streams<LDSB, DEPTH>), dim3(streams), dim3(64), 0, 0, (const u32x4 *)d, per / 16, passes, sink); }, 3); \
        printf("B per-stream lds %6d depth %d: %.2f ms  %.0f GB/s\n", LDSB, DEPTH, ms, bytes / ms / 1e6); }
#define BB(LDSB, NB) { float ms = time_ms([&] { hipLaunchKernelGGL((read_streams_burst<LDSB, NB>), dim3(streams), dim3(64), 0, 0, (const u32x4 *)d, per / 16, passes, sink); }, 3); \
        printf("C burst %d x 8 KiB lds %6d: %.2f ms  %.0f GB/s\n", NB, LDSB, ms, bytes / ms / 1e6); }
    BB(14480, 1) BB(14480, 2) BB(14480, 4) BB(20000, 2) BB(20000, 4) BB(40000, 2) BB(40000, 4) BB(80000, 4)
    B(64, 1) B(64, 2) B(4096, 1) B(4096, 2) B(9216, 1) B(9216, 2) B(14480, 1) B(14480, 2) B(20000, 1) B(20000, 2) B(40000, 1) B(40000, 2)
    return 0;
}
