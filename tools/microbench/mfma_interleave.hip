// Microbenchmark: can the next K step's fragment reads be spread through the current MFMA cluster WITHOUT a second
// fragment register set?  j-major MFMAs: after the 4 MFMAs of weight tile j its 4 registers are dead, so the next
// step's wf[j] can be read into them right away; the 4 pixel fragments follow the last MFMA.
//   A: reads, then 32 MFMAs (what k_mfma16 does)        B: (4 MFMAs, 1 read) x 8, then 4 reads
// Same tile as the kernels (64 x 128 per wave, 2 workgroups x 4 waves per CU), random LDS operands, no barriers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(const int *__restrict__ src, int *__restrict__ out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < 16384; i += 256) ((int *)smem)[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    v4i acc[4][8];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 8; j++) acc[i][j] = v4i{0, 0, 0, 0};
    v4i pf[4], wf[8];
    auto paddr = [&](int it, int i) { return smem + ((it * 8192 + w * 1024 + i * 1024 + lane * 16) & 0x3FFF); };
    auto waddr = [&](int it, int j) { return smem + 32768 + ((it * 8192 + j * 1024 + lane * 16) & 0x7FFF); };
    for (int i = 0; i < 4; i++) pf[i] = *(const v4i *)paddr(0, i);
    for (int j = 0; j < 8; j++) wf[j] = *(const v4i *)waddr(0, j);
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
            for (int j = 0; j < 8; j++) for (int i = 0; i < 4; i++)
                acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[j], pf[i], acc[i][j], 0, 0, 0);
            for (int i = 0; i < 4; i++) pf[i] = *(const v4i *)paddr(it + 1, i);
            for (int j = 0; j < 8; j++) wf[j] = *(const v4i *)waddr(it + 1, j);
            __builtin_amdgcn_sched_group_barrier(0x008, 32, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) {
#pragma unroll
                for (int i = 0; i < 4; i++) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[j], pf[i], acc[i][j], 0, 0, 0);
                wf[j] = *(const v4i *)waddr(it + 1, j);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            for (int i = 0; i < 4; i++) pf[i] = *(const v4i *)paddr(it + 1, i);
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        }
    }
    int s = 0;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 8; j++) for (int r = 0; r < 4; r++) s += acc[i][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s + pf[0][0] + wf[0][0];
}

template <int MODE>
static void run(const int *src, int *out, int iters, int round)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(2048), dim3(256), 65536, 0, src, out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("round %d  %s: %.3f ms  %.1f TOP/s\n", round, MODE ? "reads spread through the MFMA cluster" : "reads after the MFMA cluster         ", ms,
           2.0 * 2048 * 4 * (double)iters * 32 * 16 * 16 * 64 / ms / 1e9);
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
    for (int round = 0; round < 3; round++) { run<0>(src, out, iters, round); run<1>(src, out, iters, round); }
    return 0;
}
