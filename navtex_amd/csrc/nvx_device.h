// nvx_device.h -- small device-side helpers shared by the single-wave kernels (cascade, channeliser).
#ifndef NVX_DEVICE_H
#define NVX_DEVICE_H

#include <hip/hip_runtime.h>

// A single wave owns all LDS it touches; LDS instructions of one wave execute
// in program order, so cross-lane hand-offs need no s_barrier and no waitcnt --
// only the compiler must be kept from reordering the accesses.
#define NVX_WAVE_LDS_FENCE() asm volatile("" ::: "memory")

typedef short nvx_short2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));   // native vector: nontemporal builtin needs it

// NB: __builtin_bit_cast applied directly to a vector-element expression (v.x)
// reads element 0 for every component with this compiler; go through a by-value
// scalar instead.
__device__ __forceinline__ nvx_short2 as_short2(unsigned w) { return __builtin_bit_cast(nvx_short2, w); }

#endif
