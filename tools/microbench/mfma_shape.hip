// Microbenchmark: does the int8 MFMA shape matter on this (power-limited) chip?
// Same output tile per wave (64 positions x 128 channels, 128 accumulator registers), every operand
// re-read from LDS with ds_read_b128, random data, 2 workgroups of 4 waves per CU.
//   A: v_mfma_i32_32x32x32_i8, 6 fragment reads + 8 MFMAs per K=32 step
//   B: v_mfma_i32_16x16x64_i8, 12 fragment reads + 32 MFMAs per K=64 step
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(256, 2) void k(const int *__restrict__ src, int *__restrict__ out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < 16384; i += 256) ((int *)smem)[i] = src[i];   // 64 KiB of random bytes
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (SHAPE == 0) {
        v16i acc[2][4];
        for (int i = 0; i < 2; i++) for (int j = 0; j < 4; j++) for (int r = 0; r < 16; r++) acc[i][j][r] = 0;
        for (int it = 0; it < iters; it++) {
            const unsigned char *base = smem + ((it * 4096 + w * 1024) & 0xFFFF);
            v4i pf[2], wf[4];
            for (int i = 0; i < 2; i++) pf[i] = *(const v4i *)(base + ((i * 2048 + lane * 16) & 0x3FFF));
            for (int j = 0; j < 4; j++) wf[j] = *(const v4i *)(smem + 32768 + ((it * 4096 + j * 1024 + lane * 16) & 0x7FFF));
            for (int j = 0; j < 4; j++) for (int i = 0; i < 2; i++)
                acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[j], pf[i], acc[i][j], 0, 0, 0);
        }
        int s = 0;
        for (int i = 0; i < 2; i++) for (int j = 0; j < 4; j++) for (int r = 0; r < 16; r++) s += acc[i][j][r];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    } else {
        v4i acc[4][8];
        for (int i = 0; i < 4; i++) for (int j = 0; j < 8; j++) acc[i][j] = v4i{0, 0, 0, 0};
        for (int it = 0; it < iters; it += 2) {
            const unsigned char *base = smem + ((it * 4096 + w * 1024) & 0xFFFF);
            v4i pf[4], wf[8];
            for (int i = 0; i < 4; i++) pf[i] = *(const v4i *)(base + ((i * 1024 + lane * 16) & 0x3FFF));
            for (int j = 0; j < 8; j++) wf[j] = *(const v4i *)(smem + 32768 + ((it * 4096 + j * 1024 + lane * 16) & 0x7FFF));
            for (int j = 0; j < 8; j++) for (int i = 0; i < 4; i++)
                acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[j], pf[i], acc[i][j], 0, 0, 0);
        }
        int s = 0;
        for (int i = 0; i < 4; i++) for (int j = 0; j < 8; j++) for (int r = 0; r < 4; r++) s += acc[i][j][r];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    }
}

int main()
{
    const int iters = 4000, blocks = 512 * 4;
    std::vector<int> h(16384);
    srand(1);
    for (auto &v : h) v = rand() ^ (rand() << 16);
    int *src, *out;
    hipMalloc(&src, 65536);
    hipMalloc(&out, blocks * 256 * 4);
    hipMemcpy(src, h.data(), 65536, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void *)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void *)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int round = 0; round < 4; round++)
        for (int shape = 0; shape < 2; shape++) {
            hipEventRecord(a);
            if (shape == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 65536, 0, src, out, iters);
            else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 65536, 0, src, out, iters);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            const double ops = 2.0 * blocks * 4 * (double)iters * 8 * 32 * 32 * 32;
            printf("round %d shape %s: %.3f ms  %.1f TOP/s\n", round, shape ? "16x16x64" : "32x32x32", ms, ops / ms / 1e9);
        }
    return 0;
}
