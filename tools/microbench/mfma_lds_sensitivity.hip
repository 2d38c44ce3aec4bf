// How much of the (power-limited) MFMA rate do the LDS operand reads cost?  Same wave tile as the kernels
// (64 x 128 outputs, v_mfma_i32_16x16x64_i8, 2 workgroups x 4 waves per CU), READS fragment reads per 32 MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));

template <int READS>   // 12 = every operand every K step (the kernels), 6, 0 = operands stay in registers
__global__ __launch_bounds__(256, 2) void k(const int *__restrict__ src, int *__restrict__ out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < 16384; i += 256) ((int *)smem)[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    v4i acc[4][8];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 8; j++) acc[i][j] = v4i{0, 0, 0, 0};
    v4i pf[4], wf[8];
    for (int i = 0; i < 4; i++) pf[i] = *(const v4i *)(smem + ((w * 1024 + i * 1024 + lane * 16) & 0x3FFF));
    for (int j = 0; j < 8; j++) wf[j] = *(const v4i *)(smem + 32768 + ((j * 1024 + lane * 16) & 0x7FFF));
    for (int it = 0; it < iters; it++) {
        const unsigned char *base = smem + ((it * 8192 + w * 1024) & 0xFFFF);
        if (READS >= 6) for (int i = 0; i < 4; i++) pf[i] = *(const v4i *)(base + ((i * 1024 + lane * 16) & 0x3FFF));
        if (READS == 6) for (int j = 0; j < 2; j++) wf[(it * 2 + j) & 7] = *(const v4i *)(smem + 32768 + ((it * 8192 + j * 1024 + lane * 16) & 0x7FFF));
        if (READS == 12) for (int j = 0; j < 8; j++) wf[j] = *(const v4i *)(smem + 32768 + ((it * 8192 + j * 1024 + lane * 16) & 0x7FFF));
        for (int j = 0; j < 8; j++) for (int i = 0; i < 4; i++)
            acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[j], pf[i], acc[i][j], 0, 0, 0);
    }
    int s = 0;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 8; j++) for (int r = 0; r < 4; r++) s += acc[i][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int READS>
static void run(const int *src, int *out, int iters, int round)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipFuncSetAttribute((const void *)k<READS>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<READS>, dim3(2048), dim3(256), 65536, 0, src, out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("round %d  %2d fragment reads / 32 MFMAs: %.3f ms  %.1f TOP/s\n", round, READS, ms, 2.0 * 2048 * 4 * (double)iters * 32 * 16 * 16 * 64 / ms / 1e9);
}

int main()
{
    const int iters = 4000;
    std::vector<int> h(16384);
    srand(1);
    for (auto &v : h) v = rand() ^ (rand() << 16);
    int *src, *out;
    hipMalloc(&src, 65536); hipMalloc(&out, 2048 * 256 * 4);
    hipMemcpy(src, h.data(), 65536, hipMemcpyHostToDevice);
    for (int round = 0; round < 3; round++) { run<12>(src, out, iters, round); run<6>(src, out, iters, round); run<0>(src, out, iters, round); }
    return 0;
}
