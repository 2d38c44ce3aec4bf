// Microbenchmark (round 2): the pipelined conv / deconv kernels turned out to be ISSUE-bound (tools/pass_stamps.py): does the
// bigger MFMA — v_mfma_i32_32x32x32_i8, twice the MACs per instruction, 32 pipe cycles of which it holds the SIMD's issue
// port for 8 — win once a pass carries what the real passes carry?  Both shapes on the same wave tile (64 positions x 128
// channels), 2 workgroups of 4 waves per CU, per K = 64 "pass":
//     12 ds_read_b128 of fragments, NDMA LDS-DMA requests of 1 KiB (global_load_lds), a counted vmcnt wait that leaves two
//     passes' requests in flight, one workgroup barrier,
//     32 x 16x16x64  or  16 x 32x32x32 MFMAs.
// Operands as the net has them (ReLU pixels, sign-extended nibble weights).  Last three lines: the same passes by ONE 8-wave
// workgroup per CU (its LDS image is 72 KiB, so two would fit — launch_bounds keeps it at one).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_issue_pressure mfma_issue_pressure.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void *)(p))

template <int SHAPE, int NDMA, int NTHR>
__global__ __launch_bounds__(NTHR, 512 / NTHR) void k(const int *__restrict__ src, const unsigned char *__restrict__ stream, int *__restrict__ out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 64 KiB operands + 8 KiB DMA landing zone
    for (int i = threadIdx.x; i < 16384; i += NTHR) ((int *)smem)[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned char *land = smem + 65536 + (w & 3) * 2048;
    const unsigned char *gsrc = stream + (size_t)(blockIdx.x & 63) * 65536 + lane * 16;
    v16i acc32[2][4];
    v4i acc16[4][8];
    if (SHAPE == 0) { for (int i = 0; i < 2; i++) for (int j = 0; j < 4; j++) for (int r = 0; r < 16; r++) acc32[i][j][r] = 0; }
    else { for (int i = 0; i < 4; i++) for (int j = 0; j < 8; j++) acc16[i][j] = v4i{0, 0, 0, 0}; }
    for (int it = 0; it < iters; it++) {
        // fragments of this pass (12 reads)
        v4i pf[4], wf[8];
        const unsigned char *base = smem + ((it * 8192 + (w & 3) * 1024) & 0x7FFF);
#pragma unroll
        for (int i = 0; i < 4; i++) pf[i] = *(const v4i *)(base + ((i * 1024 + lane * 16) & 0x3FFF));
#pragma unroll
        for (int j = 0; j < 8; j++) wf[j] = *(const v4i *)(smem + 32768 + ((it * 8192 + j * 1024 + lane * 16) & 0x7FFF));
        // this pass's requests
#pragma unroll
        for (int d = 0; d < NDMA; d++)
            __builtin_amdgcn_global_load_lds(GLB_PTR(gsrc + ((it * NDMA + d) & 31) * 1024), LDS_PTR(land + (d & 1) * 1024), 16, 0, 0);
        if (SHAPE == 0) {
#pragma unroll
            for (int kk = 0; kk < 2; kk++)   // two K = 32 steps: weights wf[4 kk .. 4 kk + 3], pixels pf[2 kk .. 2 kk + 1]
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int i = 0; i < 2; i++)
                        acc32[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[4 * kk + j], pf[2 * kk + i], acc32[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++)
#pragma unroll
                for (int i = 0; i < 4; i++) acc16[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[j], pf[i], acc16[i][j], 0, 0, 0);
        }
        if (NDMA > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NDMA) : "memory");
        __builtin_amdgcn_s_barrier();
    }
    int s = 0;
    if (SHAPE == 0) { for (int i = 0; i < 2; i++) for (int j = 0; j < 4; j++) for (int r = 0; r < 16; r++) s += acc32[i][j][r]; }
    else { for (int i = 0; i < 4; i++) for (int j = 0; j < 8; j++) for (int r = 0; r < 4; r++) s += acc16[i][j][r]; }
    out[blockIdx.x * NTHR + threadIdx.x] = s + land[lane];
}

template <int SHAPE, int NDMA, int NTHR = 256>
static void run(const char *name, const int *src, const unsigned char *stream, int *out, int round)
{
    const int iters = 2000, blocks = 2048 * 256 / NTHR;
    const size_t lds = 65536 + 8192;
    hipFuncSetAttribute((const void *)k<SHAPE, NDMA, NTHR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL((k<SHAPE, NDMA, NTHR>), dim3(blocks), dim3(NTHR), lds, 0, src, stream, out, iters);
    hipEventRecord(a);
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL((k<SHAPE, NDMA, NTHR>), dim3(blocks), dim3(NTHR), lds, 0, src, stream, out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    ms /= 3;
    const double ops = 2.0 * blocks * (NTHR / 64) * (double)iters * 64 * 128 * 64;
    printf("round %d %-9s %d requests per pass: %.3f ms  %.2f POP/s\n", round, name, NDMA, ms, ops / ms / 1e12);
    fflush(stdout);
}

int main()
{
    std::vector<unsigned char> h(65536);
    srand(7);
    for (int i = 0; i < 32768; i++) { const int r = rand(); h[i] = (r & 1) ? 1 + ((r >> 8) % 127) : 0; }
    for (int i = 32768; i < 65536; i++) h[i] = (unsigned char)(signed char)(((rand() >> 5) & 15) - 8);
    int *src, *out;
    unsigned char *stream;
    hipMalloc(&src, 65536);
    hipMalloc(&out, 2048 * 256 * 4);
    hipMalloc(&stream, 64 * 65536);
    hipMemcpy(src, h.data(), 65536, hipMemcpyHostToDevice);
    hipMemset(stream, 1, 64 * 65536);
    for (int round = 0; round < 3; round++) {
        run<1, 0>("16x16x64", src, stream, out, round);
        run<0, 0>("32x32x32", src, stream, out, round);
        run<1, 2>("16x16x64", src, stream, out, round);
        run<0, 2>("32x32x32", src, stream, out, round);
        run<1, 3>("16x16x64", src, stream, out, round);
        run<0, 3>("32x32x32", src, stream, out, round);
        // ONE workgroup of 8 waves per CU instead of two of 4 (both waves of a SIMD behind the same barrier; a shared weight ring
        // and a taller tile would halve the requests per wave and pass)
        run<1, 1, 512>("8w 16x16", src, stream, out, round);
        run<1, 2, 512>("8w 16x16", src, stream, out, round);
        run<1, 3, 512>("8w 16x16", src, stream, out, round);
    }
    return 0;
}
