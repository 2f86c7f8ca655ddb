// nvx_device.h -- small device-side helpers shared by the single-wave kernels (cascade, channeliser).
#ifndef NVX_DEVICE_H
#define NVX_DEVICE_H

#include <hip/hip_runtime.h>
#include <type_traits>

// A single wave owns all LDS it touches; LDS instructions of one wave execute
// in program order, so cross-lane hand-offs need no s_barrier and no waitcnt --
// only the compiler must be kept from reordering the accesses.
#define NVX_WAVE_LDS_FENCE() asm volatile("" ::: "memory")

// Volatile LDS double: every access stays one ds_read_b64 / ds_write_b64.  The compiler otherwise pairs neighbouring
// 8-byte LDS accesses into ds_read2_b64, which the LDS services as two half-rate accesses (8 cycles per wave
// instruction against 2 for a ds_read_b64 of 512 contiguous bytes: MI355X_MICROARCH.md, LDS table).
typedef __attribute__((address_space(3))) volatile double lds_vdouble;

typedef double nvx_d2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) volatile nvx_d2 lds_vd2;          // one ds_read_b128 / ds_write_b128 per access

// Software pipelining by hand: a volatile access after NVX_PIN_AFTER(v) cannot be issued before v has been computed
// (the empty asm is volatile, so it keeps its order against volatile accesses, and it "modifies" v).  Without it the
// scheduler hoists every LDS read of an unrolled FIR to the top of the block and the kernel needs 140 more VGPRs.
#define NVX_PIN_AFTER(v) asm volatile("" : "+v"(v))

// compile-time loop: f(std::integral_constant<int, I>) for I = 0 .. N-1, in order
template <int I, int N, typename F>
__device__ __forceinline__ void nvx_static_for(F &&f)
{
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); nvx_static_for<I + 1, N>(f); }
}

// A wave-uniform read of launch-constant data (written by the host before the kernel started, never by the kernel):
// through the constant address space, so that it is a scalar load (s_load) and not a vector load in every lane.
__device__ __forceinline__ unsigned long long nvx_load_const_u64(const void *p)
{
    return *(const __attribute__((address_space(4))) unsigned long long *)(unsigned long long)p;
}

typedef short nvx_short2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));   // native vector: nontemporal builtin needs it

// NB: __builtin_bit_cast applied directly to a vector-element expression (v.x)
// reads element 0 for every component with this compiler; go through a by-value
// scalar instead.
__device__ __forceinline__ nvx_short2 as_short2(unsigned w) { return __builtin_bit_cast(nvx_short2, w); }

#endif
