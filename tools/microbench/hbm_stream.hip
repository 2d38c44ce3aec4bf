// Ceilings for the two HBM-bound layers: how fast can this chip stream 2.1 GB in (LDS-DMA) or out
// (16 B/lane stores) with the access shapes k_l7 / k_l0 use?   hipcc --offload-arch=gfx950 -O3 hbm_stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// each workgroup sweeps chunks of `chunk` bytes: chunk c of workgroup b = offset (c * nwg + b) * chunk
// ("interleaved": neighbouring workgroups read neighbouring chunks, like neighbouring strips)
template <int RING>
__global__ __launch_bounds__(256) void k_read_dma(const uint8_t *in, size_t bytes, int chunk, int *sink)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t nchunks = bytes / chunk;
    int slot = 0;
    for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const uint8_t *src = in + c * chunk;
        for (int p = w; p < chunk / 1024; p += 4) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + p * 1024 + lane * 16),
                                             LDS_PTR(smem + (slot * (chunk / 1024) + p) * 1024), 16, 0, 0);
        }
        slot = (slot + 1) % RING;
        if (RING == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RING > 1 ? 4 * (RING - 1) : 0) : "memory");   // chunk = 16 KiB: 4 loads per wave
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (sink && smem[threadIdx.x] == 0x77 && bytes == 1) *sink = 1;
}

// L7-like: a workgroup walks a strip; per step it reads `nruns` runs of `run` bytes, run r of step t at
//   plane(r) * plane_stride + (t * rows_per_step + row(r)) * row_pitch + strip * run
// (planes x rows per step = nruns; neighbouring workgroups read neighbouring runs of the same rows)
__global__ __launch_bounds__(256) void k_read_runs(const uint8_t *in, size_t plane_stride, int row_pitch, int planes, int rows_per_step,
                                                   int run, int steps, int strips, int *sink)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int strip = blockIdx.x % strips, chunk = blockIdx.x / strips;
    const int nruns = planes * rows_per_step, lanes_per_run = run / 16, total = nruns * lanes_per_run;
    for (int t = 0; t < steps; t++) {
        const int row0 = (chunk * steps + t) * rows_per_step;
        for (int i = w * 64 + lane, k = w; i < total + 63; i += 256, k += 4) {
            const int ii = i < total ? i : total - 1;
            const int r = ii / lanes_per_run, o = ii - r * lanes_per_run;
            const int pl = r % planes, row = r / planes;
            const uint8_t *src = in + (size_t)pl * plane_stride + (size_t)(row0 + row) * row_pitch + (size_t)strip * run + o * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             LDS_PTR(smem + ((t & 1) * 20 + k) * 1024), 16, 0, 0);
            if (i - lane + 64 >= total + 63) break;
        }
        asm volatile("s_waitcnt vmcnt(5)" ::: "memory");   // one step (<= 5 loads per wave) stays in flight
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (sink && smem[threadIdx.x] == 0x77 && steps == -1) *sink = 1;
}

// plain 16-byte loads into registers, xor-reduced
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

// 16 B/lane stores; a wave writes `run` contiguous bytes per instruction group, runs of a workgroup are `gap` apart
__global__ __launch_bounds__(256) void k_write(uint4 *out, size_t n16, uint32_t v)
{
    const size_t stride = (size_t)gridDim.x * 256;
    const uint4 val = {v, v + 1, v + 2, v + 3};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) out[i] = val;
}

// L0-like: workgroup = strip of 32 pixels, walks rows; per row-tile writes 4 planes x 8 rows x 1 KiB
__global__ __launch_bounds__(256) void k_write_l0(uint8_t *out, int W, int H, uint32_t v)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int X0 = blockIdx.x * 32;
    const uint4 val = {v, v + 1, v + 2, v + 3};
    const int rows_per = H / gridDim.y;
    for (int y0 = blockIdx.y * rows_per; y0 < (blockIdx.y + 1) * rows_per; y0 += 8)
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 4; j++) {
                const size_t off = ((size_t)j * W * H + (size_t)(y0 + 2 * w + i) * W + X0 + (lane & 31)) * 32 + 16 * (lane >> 5);
                *(uint4 *)(out + (size_t)blockIdx.z * W * H * 128 + off) = val;
            }
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
        float best = 1e9f;
        for (int r = 0; r < 5; r++) {
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("%-44s %7.3f ms  %6.2f TB/s\n", name, best, bytes / best / 1e9);
    };
    for (int wgs : {256, 512, 768, 1024, 2048}) {
        char nm[96];
        snprintf(nm, 96, "read  LDS-DMA 16KiB chunks ring1  %4d WGs", wgs);
        timeit(nm, [&] { hipLaunchKernelGGL(k_read_dma<1>, dim3(wgs), dim3(256), 16384, 0, buf, bytes, 16384, sink); });
        snprintf(nm, 96, "read  LDS-DMA 16KiB chunks ring2  %4d WGs", wgs);
        timeit(nm, [&] { hipLaunchKernelGGL(k_read_dma<2>, dim3(wgs), dim3(256), 32768, 0, buf, bytes, 16384, sink); });
        snprintf(nm, 96, "read  LDS-DMA 16KiB chunks ring3  %4d WGs", wgs);
        timeit(nm, [&] { hipLaunchKernelGGL(k_read_dma<3>, dim3(wgs), dim3(256), 49152, 0, buf, bytes, 16384, sink); });
    }
    for (int wgs : {1024, 2048, 4096, 8192}) {
        char nm[96];
        snprintf(nm, 96, "read  registers 16B/lane x4       %4d WGs", wgs);
        timeit(nm, [&] { hipLaunchKernelGGL(k_read_reg, dim3(wgs), dim3(256), 0, 0, (const uint4 *)buf, bytes / 16, sink); });
        snprintf(nm, 96, "write linear 16B/lane             %4d WGs", wgs);
        timeit(nm, [&] { hipLaunchKernelGGL(k_write, dim3(wgs), dim3(256), 0, 0, (uint4 *)buf, bytes / 16, 7u); });
    }
    {
        // 2.12 GB as `planes` planes of 1080*1920*128/planes bytes; a step = 4 rows of a 32-position strip
        struct Cfg { const char *name; int planes, run, rows_per_step; } cfgs[] = {
            {"read  runs NHWC    1 plane  x 4352 B x 4 rows", 1, 4352, 4}, {"read  runs GROUP   4 planes x 1088 B x 4 rows", 4, 1088, 4},
            {"read  runs PHASE  16 planes x  544 B x 2 rows", 16, 544, 2}, {"read  runs PHASE  16 planes x 1088 B x 2 rows (64-wide strips, 2 steps)", 16, 1088, 2},
            {"read  runs         2 planes x 2176 B x 4 rows", 2, 2176, 4}};
        for (auto &c : cfgs) {
            const int strips = 60, steps = 20;
            const int row_pitch = strips * c.run;                    // runs of neighbouring strips are adjacent
            const int rows_total = (int)(bytes / ((size_t)c.planes * row_pitch)) / (c.rows_per_step * steps) * (c.rows_per_step * steps);
            const size_t plane_stride = (size_t)row_pitch * rows_total;   // planes * plane_stride <= bytes
            const int chunks = rows_total / (c.rows_per_step * steps);
            float best = 1e9f;
            for (int r = 0; r < 5; r++) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(k_read_runs, dim3(strips * chunks), dim3(256), 40960, 0, buf, plane_stride, row_pitch, c.planes,
                                   c.rows_per_step, c.run, steps, strips, sink);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            const double moved = (double)strips * chunks * steps * c.planes * c.rows_per_step * c.run;
            printf("%-72s %7.3f ms  %6.2f TB/s  (%.2f GB)\n", c.name, best, moved / best / 1e9, moved / 1e9);
        }
    }
    for (int yc : {1, 5, 9, 27}) {
        char nm[96];
        snprintf(nm, 96, "write L0 pattern (GROUP) y_chunks %3d", yc);
        timeit(nm, [&] { hipLaunchKernelGGL(k_write_l0, dim3(W / 32, yc, N), dim3(256), 0, 0, buf, W, H, 9u); });
    }
    return 0;
}
