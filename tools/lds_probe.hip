// lds_probe.hip -- does a 512-thread workgroup with N bytes of static LDS launch and finish on this device?
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int BYTES>
__global__ __launch_bounds__(512) void k(int *out)
{
    __shared__ int lds[BYTES / 4];
    lds[threadIdx.x] = threadIdx.x;
    lds[BYTES / 4 - 1 - threadIdx.x] = 7;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = lds[5] + lds[BYTES / 4 - 3];
}
template <int BYTES> void run(int *d)
{
    int per_cu = -1;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k<BYTES>, 512, 0);
    hipLaunchKernelGGL(k<BYTES>, dim3(4), dim3(512), 0, 0, d);
    hipError_t l = hipGetLastError();
    hipError_t s = hipDeviceSynchronize();
    int h[4] = { 0 };
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("LDS %6d B: occupancy query %s per_cu %d, launch %s, sync %s, out %d\n", BYTES, hipGetErrorString(e), per_cu, hipGetErrorString(l), hipGetErrorString(s), h[0]);
    fflush(stdout);
}
int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("sharedMemPerBlock %zu maxSharedMemoryPerMultiProcessor %zu\n", p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor);
    int *d; hipMalloc(&d, 64);
    run<32768>(d); run<65536>(d); run<98304>(d); run<151216>(d); run<163840>(d);
    return 0;
}
