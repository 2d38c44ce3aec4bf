// Microbenchmark (round 2): the one-wave-per-SIMD, 128 x 128-per-wave tile again — but with the accumulators PINNED in
// AGPRs (inline-asm MFMA with a tied "+a" operand: hipcc's own allocation shuffled them through 908 v_accvgpr moves per
// 128 MFMAs in mfma_bigtile.hip) and the next pass's fragment reads interleaved by hand, one ds_read_b128 per 4 MFMAs.
//   A: 2 workgroups x 4 waves per CU, 64 x 128 per wave, 12 reads / 32 MFMAs            (what k_mfma16 does today)
//   C: 2 workgroups x 2 waves per CU (1 wave per SIMD), 128 x 128 per wave, 16 reads / 64 MFMAs, double-buffered fragments
//   D: as C, with a workgroup barrier per pass (the kernels need one to publish the next weight tiles)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));

#define MFMA(acc, A, B) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+a"(acc) : "v"(A), "v"(B))

__global__ __launch_bounds__(256, 2) void k_small(const int *__restrict__ src, int *__restrict__ out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < 16384; i += 256) ((int *)smem)[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    v4i acc[4][8];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 8; j++) acc[i][j] = v4i{0, 0, 0, 0};
    for (int it = 0; it < iters; it++) {
        const unsigned char *base = smem + ((it * 8192 + w * 1024) & 0xFFFF);
        v4i pf[4], wf[8];
        for (int i = 0; i < 4; i++) pf[i] = *(const v4i *)(base + ((i * 1024 + lane * 16) & 0x3FFF));
        for (int j = 0; j < 8; j++) wf[j] = *(const v4i *)(smem + 32768 + ((it * 8192 + j * 1024 + lane * 16) & 0x7FFF));
        for (int j = 0; j < 8; j++) for (int i = 0; i < 4; i++)
            acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[j], pf[i], acc[i][j], 0, 0, 0);
    }
    int s = 0;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 8; j++) for (int r = 0; r < 4; r++) s += acc[i][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// one pass: 64 MFMAs on (pc, wc) while the 16 fragments of the next pass are read into (pn, wn), one read per 4 MFMAs
template <bool BAR>
__device__ __forceinline__ void pass(v4i (&acc)[8][8], const v4i (&pc)[8], const v4i (&wc)[8], v4i (&pn)[8], v4i (&wn)[8],
                                     const unsigned char *smem, int it, int w, int lane)
{
    const unsigned char *pb = smem + ((it * 8192 + w * 1024) & 0xFFFF) + lane * 16;
    const unsigned char *wb = smem + 32768 + ((it * 8192) & 0x3FFF) + lane * 16;
#pragma unroll
    for (int j = 0; j < 8; j++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            MFMA(acc[i][j], wc[j], pc[i]);
            if ((i & 3) == 3) {
                const int r = j * 2 + (i >> 2);            // 0..15: pixel fragments first, then weight fragments
                if (r < 8) pn[r] = *(const v4i *)(pb + r * 1024);
                else wn[r - 8] = *(const v4i *)(wb + (r - 8) * 1024);
            }
        }
    }
    if (BAR) __builtin_amdgcn_s_barrier();
}

template <bool BAR>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_agpr(const int *__restrict__ src, int *__restrict__ out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < 16384; i += 128) ((int *)smem)[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    v4i acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) acc[i][j] = v4i{0, 0, 0, 0};
    v4i pa[8], wa[8], pb[8], wb[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        pa[i] = *(const v4i *)(smem + w * 1024 + lane * 16 + i * 1024);
        wa[i] = *(const v4i *)(smem + 32768 + lane * 16 + i * 1024);
    }
    for (int it = 0; it < iters; it += 2) {
        pass<BAR>(acc, pa, wa, pb, wb, smem, it + 1, w, lane);
        pass<BAR>(acc, pb, wb, pa, wa, smem, it + 2, w, lane);
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    int s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 8; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) s += acc[i][j][r];
    out[blockIdx.x * 128 + threadIdx.x] = s;
}

int main()
{
    const int iters = 4000;
    std::vector<int> h(16384);
    srand(1);
    for (auto &v : h) v = rand() ^ (rand() << 16);
    int *src, *out;
    hipMalloc(&src, 65536);
    hipMalloc(&out, 4096 * 256 * 4);
    hipMemcpy(src, h.data(), 65536, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void *)k_small, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void *)k_agpr<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void *)k_agpr<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const char *names[3] = {"A 2 waves/SIMD  64x128, 12 reads/32 MFMA", "C 1 wave/SIMD  128x128 AGPR, 16 reads/64 MFMA", "D = C + barrier per pass"};
    for (int round = 0; round < 4; round++)
        for (int v = 0; v < 3; v++) {
            const int blocks = 2048;   // A: 2048 x 4 waves x 32 tiles; C/D: 2048 x 2 waves x 64 tiles: same total work
            hipEventRecord(a);
            if (v == 0) hipLaunchKernelGGL(k_small, dim3(blocks), dim3(256), 65536, 0, src, out, iters);
            else if (v == 1) hipLaunchKernelGGL(k_agpr<false>, dim3(blocks), dim3(128), 65536, 0, src, out, iters);
            else hipLaunchKernelGGL(k_agpr<true>, dim3(blocks), dim3(128), 65536, 0, src, out, iters);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            const double ops = 2.0 * blocks * (v ? 2 : 4) * (double)iters * (v ? 64 : 32) * 16 * 16 * 64;
            printf("round %d %s: %.3f ms  %.1f TOP/s\n", round, names[v], ms, ops / ms / 1e9);
        }
    return 0;
}
