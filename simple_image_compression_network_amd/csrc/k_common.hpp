// Device-side helpers shared by the MFMA kernels (k_mfma.hip, k_rgb.hip).
#pragma once
#include "sicn_internal.h"

namespace sicn {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void *)(p))

constexpr int SUB_BYTES = PATCH_PIX * KSTEP;  // 10880 bytes per sub-patch
constexpr int RING = 4;                       // weight-tile ring slots
constexpr int PF = 3;                         // weight tiles in flight ahead of the consumer
constexpr uint32_t OOB = 0x80000000u;         // beyond any image: the buffer range check returns 0

template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void block_barrier()
{
    // LDS-DMA results are ordered for readers by the issuing wave's vmcnt + this barrier; a
    // __syncthreads() here would add a full vmcnt(0) and drain the weight prefetch.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int NSUB>
struct PatchGeom {
    static constexpr int BYTES = NSUB * SUB_BYTES;
    static constexpr int PIECES = (BYTES + 1023) / 1024;  // 1 KiB = one wave-wide 16-byte LDS-DMA
    static constexpr int ALLOC = PIECES * 1024;
    static constexpr int ROUNDS = (PIECES + 3) / 4;
};

// relu7((v) mod 256) for four accumulators, packed little-endian into one dword.
__device__ __forceinline__ uint32_t pack4_relu7(int a, int b, int c, int d)
{
    int sa = max((int)(int8_t)a, 0), sb = max((int)(int8_t)b, 0);
    int sc = max((int)(int8_t)c, 0), sd = max((int)(int8_t)d, 0);
    return (uint32_t)sa | ((uint32_t)sb << 8) | ((uint32_t)sc << 16) | ((uint32_t)sd << 24);
}


}  // namespace sicn
