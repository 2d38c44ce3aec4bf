// Microbenchmark: would a one-wave-per-SIMD design with a 128 x 128 output tile per wave (256
// accumulator registers, 16 fragment reads per 64 MFMAs instead of 12 per 32) beat the 2-waves-per-SIMD
// 64 x 128 form the kernels use?  Same random LDS operands, no barriers, no DMA.
//   A: 2 workgroups x 4 waves per CU, 64 x 128 per wave   (= mfma_shape.hip, shape 16x16x64)
//   B: 1 workgroup  x 4 waves per CU, 128 x 128 per wave, fragments of the next K step read under the MFMAs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));

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

__device__ __forceinline__ void read_frags(v4i (&pf)[8], v4i (&wf)[8], const unsigned char *smem, int it, int w, int lane)
{
    const unsigned char *base = smem + ((it * 8192 + w * 1024) & 0xFFFF);
    for (int i = 0; i < 8; i++) pf[i] = *(const v4i *)(base + ((i * 1024 + lane * 16) & 0x7FFF));
    for (int j = 0; j < 8; j++) wf[j] = *(const v4i *)(smem + 32768 + ((it * 8192 + j * 1024 + lane * 16) & 0x7FFF));
}

__global__ __launch_bounds__(256, 1) void k_big(const int *__restrict__ src, int *__restrict__ out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < 16384; i += 256) ((int *)smem)[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    v4i acc[8][8];
    for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) acc[i][j] = v4i{0, 0, 0, 0};
    v4i pa[8], wa[8], pb[8], wb[8];
    read_frags(pa, wa, smem, 0, w, lane);
    for (int it = 0; it < iters; it += 2) {
        read_frags(pb, wb, smem, it + 1, w, lane);
        for (int j = 0; j < 8; j++) for (int i = 0; i < 8; i++)
            acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa[j], pa[i], acc[i][j], 0, 0, 0);
        read_frags(pa, wa, smem, it + 2, w, lane);
        for (int j = 0; j < 8; j++) for (int i = 0; i < 8; i++)
            acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wb[j], pb[i], acc[i][j], 0, 0, 0);
    }
    int s = 0;
    for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) for (int r = 0; r < 4; r++) s += acc[i][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
    const int iters = 4000;
    std::vector<int> h(16384);
    srand(1);
    for (auto &v : h) v = rand() ^ (rand() << 16);
    int *src, *out;
    hipMalloc(&src, 65536);
    hipMalloc(&out, 2048 * 256 * 4);
    hipMemcpy(src, h.data(), 65536, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void *)k_small, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void *)k_big, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int round = 0; round < 4; round++)
        for (int v = 0; v < 2; v++) {
            const int blocks = v ? 1024 : 2048;   // same total work
            hipEventRecord(a);
            if (v == 0) hipLaunchKernelGGL(k_small, dim3(blocks), dim3(256), 65536, 0, src, out, iters);
            else hipLaunchKernelGGL(k_big, dim3(blocks), dim3(256), 65536, 0, src, out, iters);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            const double ops = 2.0 * blocks * 4 * (double)iters * (v ? 64 : 32) * 16 * 16 * 64;
            printf("round %d %s: %.3f ms  %.1f TOP/s\n", round, v ? "1 wave/SIMD 128x128" : "2 waves/SIMD 64x128", ms, ops / ms / 1e9);
        }
    return 0;
}
