// Microbenchmark (round 3): what does the ONE-wave-per-SIMD, 128 x 128-per-wave form (accumulators pinned in AGPRs) keep of
// its bare-loop rate once the pass carries what a real kernel's pass carries — LDS-DMA requests, a counted vmcnt wait, a
// workgroup barrier — and where in the pass should the requests sit?  One workgroup of 4 waves per CU (512 registers per
// wave), every LDS read an asm statement with a hand-placed s_waitcnt lgkmcnt, so the order written is the order issued.
//   NDMA : LDS-DMA requests (1 KiB each, L2-resident source) per wave and pass of 64 MFMAs
//   SPREAD: 0 = all requests behind the pass's first 4 MFMAs, 1 = one request every 64 / NDMA MFMAs
//   BAR  : 0 none, 1 = s_barrier per pass, 2 = per two passes
// Prints POP/s, shader cycles per pass (s_memtime) and the clock the chip held (s_memtime / s_memrealtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));

#define MFMA(acc, A, B) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+a"(acc) : "v"(A), "v"(B))
#define LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void *)(p))

__device__ __forceinline__ v4i lds_read(uint32_t addr)
{
    v4i v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}

template <int NDMA, int SPREAD, int BAR>
__device__ __forceinline__ void pass(v4i (&acc)[8][8], const v4i (&pc)[8], const v4i (&wc)[8], v4i (&pn)[8], v4i (&wn)[8],
                                     uint32_t pbase, uint32_t wbase, const unsigned char *gsrc, unsigned char *ring, int it, int lane)
{
    // the 16 reads of the next pass, one per 3 MFMAs from MFMA 2 on: the last one is issued at MFMA 47
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    int issued = 0, dma_done = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int m = j * 8 + i;
            MFMA(acc[i][j], wc[j], pc[i]);
            if (m % 3 == 2 && issued < 16) {
                if (issued < 8) pn[issued] = lds_read(pbase + issued * 1024);
                else wn[issued - 8] = lds_read(wbase + (issued - 8) * 1024);
                issued++;
            }
            if (NDMA > 0) {
                const bool now = SPREAD ? (m % (64 / NDMA) == 3 && dma_done < NDMA) : (m == 3);
                if (now) {
                    const int n = SPREAD ? 1 : NDMA;
#pragma unroll
                    for (int d = 0; d < n; d++) {
                        const int slot = (it * NDMA + dma_done) & 15;
                        __builtin_amdgcn_global_load_lds(GLB_PTR(gsrc + ((it * NDMA + dma_done) & 255) * 1024 + lane * 16),
                                                         LDS_PTR(ring + slot * 1024), 16, 0, 0);
                        dma_done++;
                    }
                }
            }
        }
    }
    if (NDMA > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");   // the previous pass's requests have landed
    if (BAR == 1 || (BAR == 2 && (it & 1))) __builtin_amdgcn_s_barrier();
}

template <int NDMA, int SPREAD, int BAR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_wide(const int *__restrict__ src, const unsigned char *__restrict__ gsrc,
                                                                                         int *__restrict__ out, unsigned long long *__restrict__ stamps, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < 16384; i += 256) ((int *)smem)[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned char *ring = smem + 65536 + w * 16384;   // this wave's 16 DMA slots
    v4i acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) acc[i][j] = v4i{0, 0, 0, 0};
    v4i pa[8], wa[8], pb[8], wb[8];
    const uint32_t lane_off = (uint32_t)(uintptr_t)LDS_PTR(smem) + lane * 16;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        pa[i] = lds_read(lane_off + w * 1024 + i * 1024);
        wa[i] = lds_read(lane_off + 32768 + i * 1024);
    }
    asm volatile("s_nop 7" ::: "memory");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it += 2) {
        pass<NDMA, SPREAD, BAR>(acc, pa, wa, pb, wb, lane_off + (((it + 1) * 8192 + w * 1024) & 0x7FFF), lane_off + 32768 + (((it + 1) * 8192) & 0x3FFF), gsrc, ring, it, lane);
        pass<NDMA, SPREAD, BAR>(acc, pb, wb, pa, wa, lane_off + (((it + 2) * 8192 + w * 1024) & 0x7FFF), lane_off + 32768 + (((it + 2) * 8192) & 0x3FFF), gsrc, ring, it + 1, lane);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_nop 15\n s_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    int s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 8; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) s += acc[i][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        stamps[blockIdx.x * 2] = t1 - t0;
        stamps[blockIdx.x * 2 + 1] = r1 - r0;
    }
}

// the two-waves-per-SIMD form the product kernels use (64 x 128 per wave, 12 reads per 32 MFMAs, compiler-scheduled reads), with
// the same kind of requests: 2 workgroups of 4 waves per CU
template <int NDMA>
__global__ __launch_bounds__(256, 2) void k_small(const int *__restrict__ src, const unsigned char *__restrict__ gsrc, int *__restrict__ out,
                                                 unsigned long long *__restrict__ stamps, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < 16384; i += 256) ((int *)smem)[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned char *ring = smem + 65536 + w * 4096;
    v4i acc[4][8];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 8; j++) acc[i][j] = v4i{0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
        const unsigned char *base = smem + ((it * 8192 + w * 1024) & 0x7FFF);
        v4i pf[4], wf[8];
        for (int i = 0; i < 4; i++) pf[i] = *(const v4i *)(base + ((i * 1024 + lane * 16) & 0x3FFF));
        for (int j = 0; j < 8; j++) wf[j] = *(const v4i *)(smem + 32768 + ((it * 8192 + j * 1024 + lane * 16) & 0x7FFF));
#pragma unroll
        for (int d = 0; d < NDMA; d++)
            __builtin_amdgcn_global_load_lds(GLB_PTR(gsrc + ((it * NDMA + d) & 255) * 1024 + lane * 16), LDS_PTR(ring + ((it * NDMA + d) & 3) * 1024), 16, 0, 0);
        for (int j = 0; j < 8; j++)
            for (int i = 0; i < 4; i++) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[j], pf[i], acc[i][j], 0, 0, 0);
        if (NDMA > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    int s = 0;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 8; j++)
            for (int r = 0; r < 4; r++) s += acc[i][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        stamps[blockIdx.x * 2] = t1 - t0;
        stamps[blockIdx.x * 2 + 1] = r1 - r0;
    }
}

// the NARROW one-wave-per-SIMD form: 64 positions x 128 channels per wave (128 accumulators: a second set of 128 could then
// drain in the background), 12 reads per 32 MFMAs; NVALU filler VALU instructions (the background drain of the other
// accumulator set: v_accvgpr_read + pack) and NST 1-KiB stores per pass, spread between the MFMAs
template <int NDMA, int NVALU, int NST>
__device__ __forceinline__ void npass(v4i (&acc)[4][8], v4i (&other)[4][8], const v4i (&pc)[4], const v4i (&wc)[8], v4i (&pn)[4], v4i (&wn)[8],
                                      uint32_t pbase, uint32_t wbase, const unsigned char *gsrc, unsigned char *ring, int it, int lane,
                                      int *sink, int &junk)
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    int issued = 0, dma_done = 0, valu_done = 0, st_done = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int m = j * 4 + i;
            MFMA(acc[i][j], wc[j], pc[i]);
            if (m % 2 == 1 && issued < 12) {
                if (issued < 4) pn[issued] = lds_read(pbase + issued * 1024);
                else wn[issued - 4] = lds_read(wbase + (issued - 4) * 1024);
                issued++;
            }
            if (NDMA > 0 && m % (32 / NDMA) == 2 && dma_done < NDMA) {
                __builtin_amdgcn_global_load_lds(GLB_PTR(gsrc + ((it * NDMA + dma_done) & 255) * 1024 + lane * 16),
                                                 LDS_PTR(ring + ((it * NDMA + dma_done) & 15) * 1024), 16, 0, 0);
                dma_done++;
            }
            // background drain: per MFMA NVALU / 32 instructions: an accvgpr read + perm / max mix on the OTHER accumulator set
            const int want = (m + 1) * NVALU / 32;
            while (valu_done < want) {
                const int k = valu_done % 64;
                int t = other[(k >> 3) & 3][k & 7][valu_done & 3];       // v_accvgpr_read
                asm volatile("v_perm_b32 %0, %1, %0, %2" : "+v"(junk) : "v"(t), "s"(0x040c000c));
                valu_done += 2;
            }
            if (NST > 0 && m % (32 / NST) == 5 && st_done < NST) {
                v4i v = {junk, junk, junk, junk};
                __builtin_nontemporal_store(v, (v4i *)(sink + ((it * NST + st_done) & 1023) * 256 + lane * 4));
                st_done++;
            }
        }
    }
    if (NDMA > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (NDMA + NST)) : "memory");
    __builtin_amdgcn_s_barrier();
}

template <int NDMA, int NVALU, int NST>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_narrow(const int *__restrict__ src, const unsigned char *__restrict__ gsrc,
                                                                                           int *__restrict__ out, unsigned long long *__restrict__ stamps, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < 16384; i += 256) ((int *)smem)[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned char *ring = smem + 65536 + w * 16384;
    v4i acc[4][8], other[4][8];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            acc[i][j] = v4i{0, 0, 0, 0};
            other[i][j] = v4i{i, j, 1, 2};
            asm volatile("" : "+a"(other[i][j]));
        }
    v4i pa[4], wa[8], pb[4], wb[8];
    const uint32_t lane_off = (uint32_t)(uintptr_t)LDS_PTR(smem) + lane * 16;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        if (i < 4) pa[i] = lds_read(lane_off + w * 1024 + i * 1024);
        wa[i] = lds_read(lane_off + 32768 + i * 1024);
    }
    asm volatile("s_nop 7" ::: "memory");
    int junk = lane;
    int *sink = out + 4096 * 64 + blockIdx.x * 0;   // stores go to a scratch region behind the results
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it += 2) {
        npass<NDMA, NVALU, NST>(acc, other, pa, wa, pb, wb, lane_off + (((it + 1) * 8192 + w * 1024) & 0x7FFF), lane_off + 32768 + (((it + 1) * 8192) & 0x3FFF), gsrc, ring, it, lane, sink, junk);
        npass<NDMA, NVALU, NST>(acc, other, pb, wb, pa, wa, lane_off + (((it + 2) * 8192 + w * 1024) & 0x7FFF), lane_off + 32768 + (((it + 2) * 8192) & 0x3FFF), gsrc, ring, it + 1, lane, sink, junk);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_nop 15\n s_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    int s = junk;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 8; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) s += acc[i][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        stamps[blockIdx.x * 2] = t1 - t0;
        stamps[blockIdx.x * 2 + 1] = r1 - r0;
    }
}

struct Variant {
    const char *name;
    const void *fn;
    int wide;   // 1: 4 waves x 64 MFMAs per pass; 0: 2 x 4 waves x 32; 2: narrow, 4 waves x 32 per pass (twice the passes)
    size_t lds;
};

int main()
{
    const int iters = 3000;
    std::vector<int> h(16384);
    srand(1);
    for (auto &v : h) v = (rand() ^ (rand() << 16)) & 0x7f7f7f7f;   // relu-like bytes
    std::vector<unsigned char> g(256 * 1024);
    for (auto &v : g) v = (unsigned char)(((rand() & 15) - 8) & 0xff);   // sign-extended nibbles
    int *src, *out;
    unsigned char *gsrc;
    unsigned long long *stamps;
    hipMalloc(&src, 65536);
    hipMalloc(&gsrc, g.size());
    hipMalloc(&out, 4096 * 256 * 4 + 1024 * 1024 * 4);
    hipMalloc(&stamps, 4096 * 16);
    hipMemcpy(src, h.data(), 65536, hipMemcpyHostToDevice);
    hipMemcpy(gsrc, g.data(), g.size(), hipMemcpyHostToDevice);
#define W(N, S, B) Variant{"wide NDMA=" #N " spread=" #S " bar=" #B, (const void *)k_wide<N, S, B>, 1, 65536 + 65536}
#define S(N) Variant{"small (2 waves/SIMD) NDMA=" #N, (const void *)k_small<N>, 0, 65536 + 16384}
#define N(D, V, T) Variant{"narrow (1 wave/SIMD, 64x128) NDMA=" #D " drainVALU=" #V " stores=" #T, (const void *)k_narrow<D, V, T>, 2, 65536 + 65536}
    Variant vs[] = {S(3), W(4, 1, 1), N(0, 0, 0), N(2, 0, 0), N(3, 0, 0), N(2, 32, 1), N(3, 32, 1), N(2, 48, 1), N(3, 64, 2)};
    for (auto &v : vs) hipFuncSetAttribute(v.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)v.lds);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    std::vector<unsigned long long> hs(4096 * 2);
    for (int round = 0; round < 3; round++)
        for (auto &v : vs) {
            const int blocks = v.wide ? 1024 : 2048;
            const int iters_v = v.wide == 2 ? 2 * iters : iters;   // same total work: wide = 4 waves x 64 MFMAs, small = 4 waves x 32 MFMAs per pass
            void *args[] = {&src, &gsrc, &out, &stamps, (void *)&iters_v};
            hipEventRecord(a);
            hipLaunchKernel(v.fn, dim3(blocks), dim3(256), args, v.lds, 0);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            hipMemcpy(hs.data(), stamps, blocks * 16, hipMemcpyDeviceToHost);
            double cyc = 0, rt = 0;
            for (int i = 0; i < blocks; i++) {
                cyc += (double)hs[2 * i];
                rt += (double)hs[2 * i + 1];
            }
            const double ops = 2.0 * blocks * 4 * (double)iters_v * (v.wide == 1 ? 64 : 32) * 16 * 16 * 64;
            printf("round %d %-40s %.3f ms  %.3f POP/s  %.0f cycles/pass  %.2f GHz\n", round, v.name, ms, ops / ms / 1e12, cyc / blocks / iters_v,
                   cyc / rt * 0.1);
        }
    return 0;
}
