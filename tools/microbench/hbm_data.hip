// Does the store (and load) bandwidth of this chip depend on the DATA?  The L0 store pattern (hbm_stream.hip: k_write_l0) with
// constant bytes against pseudo-random bytes (a different word per lane and iteration), and a read sweep of a buffer filled with
// one or the other.   hipcc --offload-arch=gfx950 -O3 hbm_data.hip -o hbm_data
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// L0-like: workgroup = strip of 32 pixels, walks rows; per row-tile writes 4 planes x 8 rows x 1 KiB
template <int RANDOM>
__global__ __launch_bounds__(256) void k_write_l0(uint8_t *out, int W, int H, uint32_t v)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int X0 = blockIdx.x * 32;
    const int rows_per = H / gridDim.y;
    uint32_t s = mix(v + threadIdx.x * 977u + blockIdx.x * 131071u + blockIdx.z * 8191u);
    for (int y0 = blockIdx.y * rows_per; y0 < (blockIdx.y + 1) * rows_per; y0 += 8)
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 4; j++) {
                uint4 val = {v, v + 1, v + 2, v + 3};
                if (RANDOM == 1) { s = s * 1664525u + 1013904223u; val = {s, s ^ 0x9e3779b9u, s * 3u, s * 5u + 1u}; }
                if (RANDOM == 2) { s = s * 1664525u + 1013904223u; val = {s & 0x7f7f7f7fu, (s >> 1) & 0x7f7f7f7fu, 0u, (s * 5u) & 0x7f7f7f7fu}; }   // ReLU-like bytes
                const size_t off = ((size_t)j * W * H + (size_t)(y0 + 2 * w + i) * W + X0 + (lane & 31)) * 32 + 16 * (lane >> 5);
                *(uint4 *)(out + (size_t)blockIdx.z * W * H * 128 + off) = val;
            }
}

__global__ __launch_bounds__(256) void k_read_reg(const uint4 *in, size_t n16, int *sink)
{
    uint4 acc = {0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        uint4 a = in[i], b = in[i + stride], c = in[i + 2 * stride], d = in[i + 3 * stride];
        acc.x ^= a.x ^ b.x ^ c.x ^ d.x; acc.y ^= a.y ^ b.y ^ c.y ^ d.y; acc.z ^= a.z ^ b.z ^ c.z ^ d.z; acc.w ^= a.w ^ b.w ^ c.w ^ d.w;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) *sink = 1;
}

int main()
{
    const int W = 1920, H = 1080, N = 8;
    const size_t bytes = (size_t)W * H * 128 * N;   // 2.12 GB
    uint8_t *buf; int *sink;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 1, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, auto launch) {
        float best = 1e9f, sum = 0;
        for (int r = 0; r < 8; r++) {
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; if (r >= 3) sum += ms;
        }
        printf("%-52s best %7.3f ms  %6.2f TB/s   mean of last 5 %7.3f ms\n", name, best, bytes / best / 1e9, sum / 5);
    };
    for (int yc : {1, 15}) {
        char nm[96];
        snprintf(nm, 96, "write L0 pattern, constant data, y_chunks %2d", yc);
        timeit(nm, [&] { hipLaunchKernelGGL(k_write_l0<0>, dim3(W / 32, yc, N), dim3(256), 0, 0, buf, W, H, 9u); });
        snprintf(nm, 96, "write L0 pattern, random data,   y_chunks %2d", yc);
        timeit(nm, [&] { hipLaunchKernelGGL(k_write_l0<1>, dim3(W / 32, yc, N), dim3(256), 0, 0, buf, W, H, 9u); });
        snprintf(nm, 96, "write L0 pattern, ReLU-like data, y_chunks %2d", yc);
        timeit(nm, [&] { hipLaunchKernelGGL(k_write_l0<2>, dim3(W / 32, yc, N), dim3(256), 0, 0, buf, W, H, 9u); });
    }
    // reads of what the last write left (ReLU-like), of constant bytes, of random bytes
    timeit("read registers, ReLU-like data", [&] { hipLaunchKernelGGL(k_read_reg, dim3(8192), dim3(256), 0, 0, (const uint4 *)buf, bytes / 16, sink); });
    hipLaunchKernelGGL(k_write_l0<1>, dim3(W / 32, 1, N), dim3(256), 0, 0, buf, W, H, 9u);
    timeit("read registers, random data", [&] { hipLaunchKernelGGL(k_read_reg, dim3(8192), dim3(256), 0, 0, (const uint4 *)buf, bytes / 16, sink); });
    CK(hipMemset(buf, 1, bytes));
    timeit("read registers, constant data", [&] { hipLaunchKernelGGL(k_read_reg, dim3(8192), dim3(256), 0, 0, (const uint4 *)buf, bytes / 16, sink); });
    return 0;
}
