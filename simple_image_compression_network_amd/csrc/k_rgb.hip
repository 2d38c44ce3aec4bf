// The two RGB-facing layers of the net — HBM-bound streaming kernels that still use int8 MFMA so
// the arithmetic stays far below the memory time:
//
//   k_l0 : conv2d<> with IFM_CH = 3 (layer 0, conv_nonsquare_top.cpp:282-286 / 299-301):
//          reads 3 B/pixel, writes OFM_CH B per output pixel (the bound: SURVEY.md §8d).
//   k_l7 : deconv522<> with OFM_CH = 3 (layer 7, conv_nonsquare_top.cpp:351-353):
//          reads IFM_CH B/pixel (the bound), writes 4 x 3 B per input position.
//
// Both follow the closed forms of SURVEY.md §8(a) a8/a9 (bit-exact, wrap mod 256, relu7).
#include <cstdlib>

#include <algorithm>

#include "k_l0_common.hpp"

namespace sicn {

// One workgroup = a vertical run of up to L0_CHUNK tiles of one 32-pixel strip.  The output stores
// of this kernel complete slowly when the chip streams 4-6 TB/s of them, and vmcnt retires in
// order: a wait for ANY load issued after a tile's stores is a wait for those stores (with one wait
// per tile the kernel ran at 4.5 TB/s of stores even without its MFMAs).  So the loop contains no
// load at all: the raw pixels of the whole run (3 B/pixel: 30 KB for 9 tiles), the weights and the
// bias are brought into LDS by one burst of LDS-DMA in the prologue, and per tile the kernel only
// expands the next tile's pixels LDS -> LDS, runs its MFMAs and issues its 8 stores per wave
// (lanes outside the image: out-of-range offset of a buffer descriptor), separated by a raw barrier.
// The two workgroups of a CU cover each other's prologue.
template <int NTJ>
__global__ __launch_bounds__(256, 2) void k_l0(const uint8_t *__restrict__ in, uint8_t *__restrict__ out,
                                               const int8_t *__restrict__ w_l0,
                                               const int8_t *__restrict__ bias, int IW, int IH, int OW,
                                               int OH, int tiles_y, int ty_per, int out_layout, uint32_t act_floor)
{
    constexpr int COUT = NTJ * 32;
    constexpr int TB = COUT * KSTEP;
    constexpr int WBYTES = 5 * TB;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *wl = smem;
    uint8_t *patch0 = smem + WBYTES;                  // two patches of L0_PATCH bytes (16-byte multiples)
    uint8_t *bias_lds = patch0 + 2 * L0_PATCH;
    uint8_t *raw = bias_lds + 128;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, kh = lane >> 5;
    const int img = blockIdx.z;
    const int X0 = blockIdx.x * TILE_X;
    const int ty_begin = blockIdx.y * ty_per, ty_end = min(tiles_y, ty_begin + ty_per);   // ty_per <= L0_CHUNK
    if (ty_begin >= ty_end) return;  // before any LDS-DMA is issued

    const int im_bytes = IH * IW * 3;
    const int img_byte0 = img * im_bytes;                       // launch_l0 guarantees the tensor is < 2 GiB
    const int tensor_bytes4 = ((int)gridDim.z * im_bytes + 3) & ~3;
    uint8_t *out_img = out + (size_t)img * OH * OW * COUT;
    const TensorMap om = tensor_map(out_layout, COUT, OW, OH);

    // ---- prologue: everything this workgroup will ever read ------------------------------------
    {
        // raw rows 2*Y - 2 .. of the run, 52 dwords each from the dword holding pixel 2*X0-2: lane l of
        // instruction k fetches dword idx = 64k + l (row idx / 52, dword idx % 52) through a descriptor over
        // the WHOLE input tensor (its base is allocator-aligned, an image base need not be; rows outside the
        // image and offsets outside the tensor read 0)
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, tensor_bytes4, 0x00020000);
        const int rows = 2 * L0_TY * (ty_end - ty_begin) + 3;
        const int n_instr = (rows * L0_RAW_DW + 63) / 64;
        for (int k = w; k < n_instr; k += 4) {
            const int idx = 64 * k + lane;
            const int r = idx / L0_RAW_DW, c = idx - r * L0_RAW_DW;
            const int iy = 2 * L0_TY * ty_begin - 2 + r;
            const int o = ((img_byte0 + (iy * IW + 2 * X0 - 2) * 3) & ~3) + 4 * c;   // negative only left of the very first pixel
            const bool ok = r < rows && iy >= 0 && iy < IH && o >= 0;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(raw + k * 256), 4, ok ? (uint32_t)o : OOB, 0, 0, 0);
        }
        // weights: 5 tiles of [COUT][32 B] (row permutation + half swizzle as in k_mfma), linear copy
        for (int piece = w; piece < WBYTES / 1024; piece += 4)
            __builtin_amdgcn_global_load_lds(GLB_PTR(w_l0 + piece * 1024 + lane * 16), LDS_PTR(wl + piece * 1024), 16, 0, 0);
        if (tid < COUT / 4) ((uint32_t *)bias_lds)[tid] = ((const uint32_t *)bias)[tid];
    }
    uint32_t wrow[NTJ];
#pragma unroll
    for (int j = 0; j < NTJ; j++) wrow[j] = (uint32_t)((j * 32 + m) * 32 + ((kh ^ ((m >> 3) & 1)) << 4));
    // the builtin (not inline asm): hipcc then KNOWS no LDS-DMA is pending and puts no vmcnt(0) of
    // its own in front of the LDS accesses of the loop
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    block_barrier();
    l0_expand(raw, patch0, img_byte0, tid, ty_begin * L0_TY, X0, IW, IH);
    l0_expand(raw, patch0, img_byte0, tid + 256, ty_begin * L0_TY, X0, IW, IH);
    block_barrier();

    int buf = 0;
    for (int tile_y = ty_begin; tile_y < ty_end; tile_y++, buf ^= 1) {
        const int Y0 = tile_y * L0_TY;
        const uint8_t *patch = patch0 + buf * L0_PATCH;
        if (tile_y + 1 < ty_end) {   // pixels of the next tile -> the patch nobody reads in this iteration
            const uint8_t *rsrc = raw + (tile_y + 1 - ty_begin) * (2 * L0_TY * L0_RAW_DW * 4);
            uint8_t *nxt = patch0 + (buf ^ 1) * L0_PATCH;
            l0_expand(rsrc, nxt, img_byte0, tid, Y0 + L0_TY, X0, IW, IH);
            l0_expand(rsrc, nxt, img_byte0, tid + 256, Y0 + L0_TY, X0, IW, IH);
        }

        v16i acc[2][NTJ];
#pragma unroll
        for (int j = 0; j < NTJ; j++) {
            const v4i b4 = *(const v4i *)(bias_lds + j * 32 + 16 * kh);
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int bv = (int)(int8_t)((uint32_t)b4[r >> 2] >> (8 * (r & 3)));
                acc[0][j][r] = bv;
                acc[1][j][r] = bv;
            }
        }
#ifndef SICN_EXP_L0_NO_MFMA   // experiment builds only (tools/ab_libs.sh)
#pragma unroll
        for (int ky = 0; ky < 5; ky++) {
            v4i wf[NTJ], pf[2];
#pragma unroll
            for (int j = 0; j < NTJ; j++) wf[j] = *(const v4i *)(wl + ky * TB + wrow[j]);
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const uint8_t *src = patch + (2 * (2 * w + i) + ky) * L0_PITCH + 8 * m + 16 * kh;
                const uint2 lo = *(const uint2 *)src, hi = *(const uint2 *)(src + 8);
                pf[i] = v4i{(int)lo.x, (int)lo.y, (int)hi.x, (int)hi.y};
            }
#pragma unroll
            for (int j = 0; j < NTJ; j++)
#pragma unroll
                for (int i = 0; i < 2; i++)
                    acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[j], pf[i], acc[i][j], 0, 0, 0);
        }
#endif
        __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)out_img, 0, OH * OW * COUT, 0x00020000);
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int gy = Y0 + 2 * w + i, gx = X0 + m;
            const bool ok = gy < OH && gx < OW;
#pragma unroll
            for (int j = 0; j < NTJ; j++) {
                const v16i a = acc[i][j];
                v4i v;
                v[0] = (int)pack4_relu7(a[0], a[1], a[2], a[3], act_floor & ACT_FLOOR_MASK);
                v[1] = (int)pack4_relu7(a[4], a[5], a[6], a[7], act_floor & ACT_FLOOR_MASK);
                v[2] = (int)pack4_relu7(a[8], a[9], a[10], a[11], act_floor & ACT_FLOOR_MASK);
                v[3] = (int)pack4_relu7(a[12], a[13], a[14], a[15], act_floor & ACT_FLOOR_MASK);
                const uint32_t off = ok ? tensor_offset(om, gy, gx, (uint32_t)j) + 16u * kh : OOB;
#ifdef SICN_L0_NT
                if (act_floor & ACT_NT_STORE)
                    __builtin_amdgcn_raw_buffer_store_b128(v, ro, off, 0, 2);
                else
#endif
                    __builtin_amdgcn_raw_buffer_store_b128(v, ro, off, 0, 0);
            }
        }
        block_barrier();  // next patch complete, this patch free (raw barrier: the stores stay in flight)
    }
}

static_assert(L0_RAW_BYTES >= (L0_RAW_ROWS * L0_RAW_DW + 63) / 64 * 256, "k_l0: whole request instructions fit the raw buffer");
#ifdef SICN_ALT_KERNELS   // measured 11 % slower than k_l0: ALT build only (libsicn_alt.so), the product library rejects l0_form = 2
// ---- the PERSISTENT form of layer 0 (round 4) -------------------------------------------------------------------------------------
// k_l0's cost model on 8 x 4K is 0.40 ms + 0.365 ms / (tiles per run): every workgroup pays a prologue — the burst that brings its
// run's raw pixels, the weights and the bias into LDS, one memory latency, two barriers — that only the CU's other workgroup
// covers.  Here two workgroups per CU each walk MANY runs of at most L0P_RUN tiles: the weights and the bias are loaded once,
// and the raw pixels of the NEXT run arrive by LDS-DMA in a second buffer while the current run is computed.  The tile loop still
// contains no wait that could catch a store: the one wait per run that the next run's pixels need is a COUNTED vmcnt (the
// requests are older than the 8 stores per tile issued since), followed by one extra barrier per run.
constexpr int L0P_RUN = 7;                                  // tiles per run: two raw buffers of 23.9 KB fit beside weights and patches
constexpr int L0P_RAW_ROWS = 2 * L0_TY * L0P_RUN + 3;       // 115
// a buffer holds WHOLE request instructions (64 lanes x 16 B): the lanes past the last row still write (zeros) — 512 bytes past a
// buffer sized by rows alone, i.e. into the other buffer or, from the second one, past the workgroup's LDS and into the weights of
// the CU's other workgroup (LDS-DMA is not held to the allocation: seen as wrong outputs at 2 x 4K, never with one workgroup per CU)
constexpr int L0P_RAW_BYTES = (L0P_RAW_ROWS * (L0_RAW_DW / 4) + 63) / 64 * 1024;
static_assert(L0P_RAW_BYTES >= L0P_RAW_ROWS * L0_RAW_DW * 4 + 12, "the last quad over-reads 3 dwords");
// one kernel row's MFMA operands of k_l0p: NTJ weight fragments and the pixel fragments of the wave's two output rows
template <int NTJ>
struct L0pFrag {
    v4i wf[NTJ], pf[2];
    template <int KY>
    __device__ __forceinline__ void request(uint32_t wl_addr, const uint32_t (&wrow)[NTJ], uint32_t patch_addr)
    {
#pragma unroll
        for (int j = 0; j < NTJ; j++)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[j]) : "v"(wl_addr + wrow[j]), "n"(KY * NTJ * 32 * KSTEP) : "memory");
        // 8-byte aligned: two qwords; row 2 (2 w + i) + ky of the patch
        asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(pf[0]) : "v"(patch_addr), "n"(KY * L0_PITCH / 8), "n"(KY * L0_PITCH / 8 + 1) : "memory");
        asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(pf[1]) : "v"(patch_addr), "n"((KY + 2) * L0_PITCH / 8), "n"((KY + 2) * L0_PITCH / 8 + 1) : "memory");
    }
    template <int YOUNGER>
    __device__ __forceinline__ void wait()
    {
        static_assert(NTJ == 4, "the operand list below names four weight fragments");
        asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(wf[0]), "+v"(wf[1]), "+v"(wf[2]), "+v"(wf[3]), "+v"(pf[0]), "+v"(pf[1]) : "n"(YOUNGER) : "memory");
    }
    __device__ __forceinline__ void mfma(v16i (&acc)[2][NTJ]) const
    {
#pragma unroll
        for (int j = 0; j < NTJ; j++)
#pragma unroll
            for (int i = 0; i < 2; i++) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[j], pf[i], acc[i][j], 0, 0, 0);
    }
};
static_assert((L0_PITCH % 8) == 0 && (6 * L0_PITCH / 8 + 1) < 256, "ds_read2_b64 offsets are 8-bit counts of qwords");

template <int NTJ>
__global__ __launch_bounds__(256, 2) void k_l0p(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, const int8_t *__restrict__ w_l0,
                                                const int8_t *__restrict__ bias, int IW, int IH, int OW, int OH, int tiles_x, int tiles_y,
                                                int runs_y, int n_images, int out_layout, uint32_t act_floor)
{
    constexpr int COUT = NTJ * 32;
    constexpr int TB = COUT * KSTEP;
    constexpr int WBYTES = 5 * TB;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *wl = smem;
    uint8_t *patch0 = smem + WBYTES;                  // two patches of L0_PATCH bytes
    uint8_t *bias_lds = patch0 + 2 * L0_PATCH;
    uint8_t *raw0 = bias_lds + 128;                   // two raw buffers of L0P_RAW_BYTES

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, kh = lane >> 5;
    const int total_runs = tiles_x * runs_y * n_images;
    int run = (int)blockIdx.x;
    if (run >= total_runs) return;   // before any LDS-DMA is issued

    const int im_bytes = IH * IW * 3;
    const int tensor_bytes4 = (n_images * im_bytes + 3) & ~3;
    const TensorMap om = tensor_map(out_layout, COUT, OW, OH);
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, tensor_bytes4, 0x00020000);

    struct Run { int img, X0, ty_begin, ty_end, img_byte0; };
    auto locate = [&](int r) {   // strips fastest: neighbouring workgroups read neighbouring columns of the same input rows
        const int img = r / (tiles_x * runs_y), rem = r - img * (tiles_x * runs_y), ry = rem / tiles_x, bx = rem - ry * tiles_x;
        const int tb = ry * L0P_RUN;
        return Run{img, bx * TILE_X, tb, min(tiles_y, tb + L0P_RUN), img * im_bytes};
    };
    // the raw rows 2 Y - 2 .. of a run, 52 dwords each from the dword holding pixel 2 X0 - 2 (the image k_l0 keeps), moved SIXTEEN
    // bytes per lane: 13 units per row, lane l of instruction k fetches unit 64 k + l.  (Not only a quarter of the requests: hipcc
    // tracks the 4-byte `buffer_load_dword .. lds` as a pending LDS write and put an s_waitcnt vmcnt(0) — i.e. a wait for the tile's
    // STORES — in front of every LDS access of the tile loop; the 16-byte form it leaves alone, as in the other kernels.)
    // Rows outside the image read 0.  A unit is never dropped for a negative offset: the only row that starts before the tensor
    // is row 0 of image 0 in strip 0 (offset -8: two dwords of padding, then pixels 0 .. 2) — its first unit is fetched from
    // offset 0 into the slots 8 bytes on... which LDS-DMA cannot do (a lane's destination is fixed), so that one unit is zero-filled
    // here and patched by two ordinary dword loads in the prologue of the workgroup that owns run 0 (always its first run).
    constexpr int UNITS = L0_RAW_DW / 4;   // 13
    static_assert(L0_RAW_DW % 4 == 0, "a raw row is a whole number of 16-byte units");
    auto request_raw = [&](const Run &r, uint8_t *raw) {
        const int rows = 2 * L0_TY * (r.ty_end - r.ty_begin) + 3;
        const int n_instr = (rows * UNITS + 63) / 64;
        for (int k = w; k < n_instr; k += 4) {
            const int idx = 64 * k + lane;
            const int rr = idx / UNITS, c = idx - rr * UNITS;
            const int iy = 2 * L0_TY * r.ty_begin - 2 + rr;
            const int o = ((r.img_byte0 + (iy * IW + 2 * r.X0 - 2) * 3) & ~3) + 16 * c;
            const bool ok = rr < rows && iy >= 0 && iy < IH && o >= 0;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(raw + k * 1024), 16, ok ? (uint32_t)o : OOB, 0, 0, 0);
        }
    };

    Run cur = locate(run);
    request_raw(cur, raw0);
    for (int piece = w; piece < WBYTES / 1024; piece += 4)
        __builtin_amdgcn_global_load_lds(GLB_PTR(w_l0 + piece * 1024 + lane * 16), LDS_PTR(wl + piece * 1024), 16, 0, 0);
    if (tid < COUT / 4) ((uint32_t *)bias_lds)[tid] = ((const uint32_t *)bias)[tid];
    uint32_t wrow[NTJ];
#pragma unroll
    for (int j = 0; j < NTJ; j++) wrow[j] = (uint32_t)((j * 32 + m) * 32 + ((kh ^ ((m >> 3) & 1)) << 4));
    // the one unit that starts before the tensor (see request_raw): raw row 2 (input row 0) of run 0, bytes 8 .. 15 = tensor bytes 0 .. 7
    uint32_t corner = 0;
    const bool patch_corner = run == 0 && tid < 2;
    if (patch_corner) corner = ((const uint32_t *)in)[tid];   // the tensor holds at least one pixel row of >= 1 pixel... and 8 bytes? checked by the launcher
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), the builtin: hipcc then knows nothing is pending (see k_l0)
    block_barrier();
    if (patch_corner) ((uint32_t *)(raw0 + 2 * L0_RAW_DW * 4 + 8))[tid] = corner;
    block_barrier();
    l0_expand(raw0, patch0, cur.img_byte0, tid, cur.ty_begin * L0_TY, cur.X0, IW, IH);
    l0_expand(raw0, patch0, cur.img_byte0, tid + 256, cur.ty_begin * L0_TY, cur.X0, IW, IH);
    block_barrier();

    int buf = 0, rbuf = 0;
    for (;;) {
        const int next_run = run + (int)gridDim.x;
        const bool has_next = next_run < total_runs;
        Run nxt = cur;
        uint8_t *raw = raw0 + rbuf * L0P_RAW_BYTES, *raw_next = raw0 + (rbuf ^ 1) * L0P_RAW_BYTES;
        if (has_next) {   // the next run's pixels: in flight under this whole run (the buffer was free since the previous run ended)
            nxt = locate(next_run);
            request_raw(nxt, raw_next);
        }
        uint8_t *out_img = out + (size_t)cur.img * OH * OW * COUT;
        __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)out_img, 0, OH * OW * COUT, 0x00020000);
        const int n_tiles = cur.ty_end - cur.ty_begin;
        for (int t = 0; t < n_tiles; t++, buf ^= 1) {
            const int Y0 = (cur.ty_begin + t) * L0_TY;
            const uint8_t *patch = patch0 + buf * L0_PATCH;
            uint8_t *pnext = patch0 + (buf ^ 1) * L0_PATCH;
            if (t + 1 < n_tiles) {   // pixels of the run's next tile -> the patch nobody reads in this iteration
                const uint8_t *rsrc = raw + (t + 1) * (2 * L0_TY * L0_RAW_DW * 4);
                l0_expand<true>(rsrc, pnext, cur.img_byte0, tid, Y0 + L0_TY, cur.X0, IW, IH);
                l0_expand<true>(rsrc, pnext, cur.img_byte0, tid + 256, Y0 + L0_TY, cur.X0, IW, IH);
            } else if (has_next) {
                // the next run's first tile.  Its raw rows were requested before this run's first store: everything but the
                // 8 (t) stores issued since has to have landed — counted, so no store is waited for — in every wave
                switch (t) {
                case 0: wait_vmcnt<0>(); break;
                case 1: wait_vmcnt<8>(); break;
                case 2: wait_vmcnt<16>(); break;
                case 3: wait_vmcnt<24>(); break;
                case 4: wait_vmcnt<32>(); break;
                case 5: wait_vmcnt<40>(); break;
                default: wait_vmcnt<48>(); break;
                }
                block_barrier();
                l0_expand<true>(raw_next, pnext, nxt.img_byte0, tid, nxt.ty_begin * L0_TY, nxt.X0, IW, IH);
                l0_expand<true>(raw_next, pnext, nxt.img_byte0, tid + 256, nxt.ty_begin * L0_TY, nxt.X0, IW, IH);
            }

            v16i acc[2][NTJ];
#pragma unroll
            for (int j = 0; j < NTJ; j++) {
                const v4i b4 = *(const v4i *)(bias_lds + j * 32 + 16 * kh);
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int bv = (int)(int8_t)((uint32_t)b4[r >> 2] >> (8 * (r & 3)));
                    acc[0][j][r] = bv;
                    acc[1][j][r] = bv;
                }
            }
            // The fragment reads are asm (a read hipcc can see is ordered behind the pending LDS-DMA with a vmcnt(0), i.e. behind the
            // tile's stores), so their pipeline is written out: kernel row ky + 1 is requested before the MFMAs of row ky, and each
            // wait NAMES the registers it releases — an MFMA depends on those, not on a bare s_waitcnt statement (the first build had
            // one, hipcc moved the MFMAs in front of it: a quarter of all outputs wrong).  LDS operations hipcc issues itself in
            // between can only make the counted waits stricter.
            const uint32_t wa = (uint32_t)(uintptr_t)LDS_PTR(wl), pa = (uint32_t)(uintptr_t)LDS_PTR(patch + 4 * w * L0_PITCH + 8 * m + 16 * kh);
            L0pFrag<NTJ> fa, fb;
            fa.template request<0>(wa, wrow, pa);
            fb.template request<1>(wa, wrow, pa);
            fa.template wait<NTJ + 2>();
            fa.mfma(acc);
            fa.template request<2>(wa, wrow, pa);
            fb.template wait<NTJ + 2>();
            fb.mfma(acc);
            fb.template request<3>(wa, wrow, pa);
            fa.template wait<NTJ + 2>();
            fa.mfma(acc);
            fa.template request<4>(wa, wrow, pa);
            fb.template wait<NTJ + 2>();
            fb.mfma(acc);
            fa.template wait<0>();
            fa.mfma(acc);
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int gy = Y0 + 2 * w + i, gx = cur.X0 + m;
                const bool ok = gy < OH && gx < OW;
#pragma unroll
                for (int j = 0; j < NTJ; j++) {
                    const v16i a = acc[i][j];
                    v4i v;
                    v[0] = (int)pack4_relu7(a[0], a[1], a[2], a[3], act_floor & ACT_FLOOR_MASK);
                    v[1] = (int)pack4_relu7(a[4], a[5], a[6], a[7], act_floor & ACT_FLOOR_MASK);
                    v[2] = (int)pack4_relu7(a[8], a[9], a[10], a[11], act_floor & ACT_FLOOR_MASK);
                    v[3] = (int)pack4_relu7(a[12], a[13], a[14], a[15], act_floor & ACT_FLOOR_MASK);
                    const uint32_t off = ok ? tensor_offset(om, gy, gx, (uint32_t)j) + 16u * kh : OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(v, ro, off, 0, 0);   // ALWAYS issued (out-of-image lanes: out of range): the counted wait relies on 8 per tile
                }
            }
            block_barrier();  // next patch complete, this patch free (raw barrier: the stores stay in flight)
        }
        if (!has_next) break;
        run = next_run;
        cur = nxt;
        rbuf ^= 1;
    }
}
#endif   // SICN_ALT_KERNELS: k_l0p

// Test hook: sicn_options.strip_chunks = n forces the number of vertical chunks a strip is cut into (the
// default heuristic gives small images one step per workgroup, which never exercises the rolling window).

size_t l0_bytes(int cout) { return (size_t)5 * cout * KSTEP; }

void pack_l0(const int8_t *w_okc, int cout, int8_t *dst)
{
    // w_okc: [cout][75], k = (ky*5+kx)*3 + c
    for (int ky = 0; ky < 5; ky++)
        for (int row = 0; row < cout; row++) {
            const int j = row >> 5, rho = row & 31;
            const int ch = j * 32 + 16 * ((rho >> 2) & 1) + 4 * (rho >> 3) + (rho & 3);
            int8_t logical[32];
            for (int s = 0; s < 8; s++)
                for (int c = 0; c < 4; c++)
                    logical[s * 4 + c] = (s < 5 && c < 3) ? w_okc[(size_t)ch * 75 + (ky * 5 + s) * 3 + c] : 0;
            const int g = (row >> 3) & 1;
            int8_t *t = dst + ((size_t)ky * cout + row) * 32;
            for (int h = 0; h < 2; h++)
                for (int b = 0; b < 16; b++) t[((h ^ g) << 4) + b] = logical[h * 16 + b];
        }
}

hipError_t launch_l0(const LayerGeom &g, const sicn_weights &w, const uint8_t *in, uint8_t *out,
                     int n_images, hipStream_t stream, int out_layout, const sicn_options &o, const ChipGeom &chip, bool relu)
{
    const int tiles_x = (g.OW + TILE_X - 1) / TILE_X, tiles_y = (g.OH + L0_TY - 1) / L0_TY;
    // runs of at most L0_CHUNK tiles (the LDS holds a run's pixels), evened out; small images: shorter runs, enough workgroups
    // (about four per CU, sicn_plan.h); the test hook (strip_chunks) can only shorten them
    const L0Cut cut = l0_chunks(tiles_x, tiles_y, n_images, L0_CHUNK, o.strip_chunks, chip);
    const int y_chunks = cut.y_chunks, ty_per = cut.ty_per;
    dim3 grid((unsigned)tiles_x, (unsigned)y_chunks, (unsigned)n_images);
    if ((size_t)g.IH * g.IW * 3 * (size_t)n_images + 4 >= (size_t)OOB) return hipErrorInvalidValue;
    if ((size_t)g.OH * g.OW * g.COUT >= (size_t)OOB) return hipErrorInvalidValue;   // buffer-descriptor stores
#ifndef SICN_ALT_KERNELS
    if (o.l0_form == 2) return hipErrorInvalidValue;   // k_l0p: ALT build only
#else
    // the persistent form (k_l0p): sicn_options.l0_form = 2 only.  Measured on 8 x 4K (r04, A/B in one process): 0.481 ms against
    // k_l0's 0.434 — the layer is bound by issue slots (40 MFMAs of 32 cycles + the packing per tile and wave), not by the prologue
    // the persistent form removes; DESIGN.md §3.2.  Kept as a tested alternative.
    const int runs_y = (tiles_y + L0P_RUN - 1) / L0P_RUN;
    const long total_runs = (long)tiles_x * runs_y * n_images;
    // (the corner patch of k_l0p reads the tensor's first 8 bytes: at least three pixels)
    const bool persistent = o.l0_form == 2 && g.COUT == 128 && (size_t)g.IH * g.IW * n_images >= 3;
    if (persistent) {
        if (total_runs > 0x7fffffffL) return hipErrorInvalidValue;
        const size_t lds = 5 * 128 * KSTEP + 2 * L0_PATCH + 128 + 2 * L0P_RAW_BYTES;
        hipError_t e = hipFuncSetAttribute((const void *)k_l0p<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        long cap = o.persistent_grid > 0 ? o.persistent_grid : 2L * chip.n_cu;
        const unsigned wgs = (unsigned)std::min<long>(cap < 1 ? 1 : cap, total_runs);
        hipLaunchKernelGGL(k_l0p<4>, dim3(wgs), dim3(256), lds, stream, in, out, w.d_w_l0, w.d_bias, g.IW, g.IH, g.OW, g.OH, tiles_x, tiles_y,
                           runs_y, n_images, out_layout, relu ? ACT_FLOOR_RELU : ACT_FLOOR_RAW);
        return hipGetLastError();
    }
#endif
    if (g.COUT == 128) {
        const size_t lds = 5 * 128 * KSTEP + 2 * L0_PATCH + 128 + L0_RAW_BYTES;
        hipError_t e = hipFuncSetAttribute((const void *)k_l0<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_l0<4>, grid, dim3(256), lds, stream, in, out, w.d_w_l0, w.d_bias, g.IW, g.IH,
                           g.OW, g.OH, tiles_y, ty_per, out_layout,
                           (relu ? ACT_FLOOR_RELU : ACT_FLOOR_RAW) | (nt_store_wanted((size_t)g.OH * g.OW * g.COUT * n_images) ? ACT_NT_STORE : 0u));
    } else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

// =============================================================================================
// Layer 7.  Per input position (y,x) the 4 output phases x 3 channels are 12 dot products over the
// 3x3 input neighbourhood: row (4*phase + c) of a 16-row virtual weight matrix W' holds
// W[c][2dy-py][2dx-px][:] at neighbourhood tap (dy,dx) (zero where that tap does not belong to the
// phase).  v_mfma_i32_16x16x64_i8: A = W' (16 rows x 64 K bytes), B = 16 positions, 18 K steps
// (9 taps x two 64-channel halves).  Output tile: lane (position l&15, phase l>>4) holds the 3
// channel bytes of out[2y+py][2x+px].
//   * W' (18 KB) is held in REGISTERS (72 VGPRs) for the whole vertical strip a workgroup walks.
//   * The strip is walked in steps of 4 input rows (wave w = row w, 32 positions).  The input
//     lives in a ROLLING LDS window: position P = r*36 + tx (r = row counted from the top of the
//     strip, 34 of the 36 columns used) sits at ring slot P mod 368.  One step needs rows
//     4t+2..4t+7 and, while it computes, the 4 rows of the next step (exactly 9 pieces of 16
//     positions per region) are fetched by LDS-DMA into the slots the previous step freed: every
//     input row is fetched once per strip (fetch = 34/32 of the input) and the load latency is
//     hidden behind a whole step.  One barrier per step.
//   * Two LDS regions (K-group parity kg&1), each [slot][4 chunks x 16 B] with chunk
//     c2 = 2*half + (kg>>1) stored at c2 ^ ((P>>2)&3): a 16-lane ds_read_b128 group covers 16
//     distinct positions at ONE c2, i.e. 16 distinct 16-byte slots of a 256-byte LDS row.  Ring
//     wrap and step advance are multiples of 16 positions, so the XOR term is a lane constant.
//     Which channels a K byte stands for is free (W' is packed to match): region R holds the
//     32-channel groups 2R, 2R+1 whole, so every LDS-DMA lane pair fetches one contiguous 32-byte
//     group entry (region-interleaved halves made each instruction use half of every line it touched).
//   * RGB output is transposed through a 384-byte per-wave LDS staging tile and stored as dwords
//     through a buffer descriptor (masked lanes = out of range), so a step always issues exactly
//     two stores and the wait for the prefetched rows is a counted vmcnt that leaves them in flight.
// =============================================================================================
#ifndef SICN_L7_AUX
#define SICN_L7_AUX 0
#endif
#ifndef SICN_EXP_L7_STORE
#define SICN_EXP_L7_STORE 0
#endif
constexpr int L7_AUX = SICN_L7_AUX;                       // cache policy of the input stream (2 = nt)
constexpr int L7_PITCH = 36;                             // positions per window row (34 used)
constexpr int L7_ROWS = 4;                               // input rows per step
constexpr int L7_STEP_PIECES = L7_ROWS * L7_PITCH / 16;  // 9 LDS-DMA pieces per region per step
#ifndef SICN_L7_AHEAD
#define SICN_L7_AHEAD 1
#endif
constexpr int L7_AHEAD = SICN_L7_AHEAD;                  // steps of rows in flight ahead of the step being computed
constexpr int L7_RING_PIECES = ((4 * L7_AHEAD + 6) * L7_PITCH + 15) / 16;   // 1 ahead: 23 pieces = 368 positions >= 10 rows
constexpr int L7_WGS = L7_AHEAD == 1 ? 3 : 2;            // workgroups per CU the LDS ring allows
constexpr int L7_RING_POS = L7_RING_PIECES * 16;
constexpr int L7_REGION = L7_RING_PIECES * 1024;
constexpr int L7_STAGE = 2 * 192;                        // 2 output rows x 64 pixels x 3 B per wave
static_assert(L7_ROWS * L7_PITCH % 16 == 0 && L7_RING_POS >= (4 * L7_AHEAD + 6) * L7_PITCH, "ring geometry");

// LDS-DMA pieces `first + 4i` (i < n) of a run of window rows: piece j < npieces covers region
// j / per_region, positions (j % per_region)*16 .. +15 of the run; row 0 of the run is input row iy0
// and lands at ring piece slot pslot0.  Only rows iy_min <= iy < iy_max (inside the image) are
// fetched, the rest is zero fill.  Every wave issues exactly N loads (the vmcnt bookkeeping of the
// caller depends on it): a piece index past the run becomes an out-of-range load into `scratch`.
template <int N>
__device__ __forceinline__ void l7_load_rows(uint8_t *patch, uint8_t *scratch, const uint8_t *in_img, int in_img_bytes, int w,
                                             int lane, int per_region, int iy0, int iy_min, int iy_max, int pslot0, int X0,
                                             int IW, const TensorMap &tm)
{
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)in_img, 0, in_img_bytes, 0x00020000);
    const int c2 = (lane & 3) ^ ((lane >> 4) & 3);   // stored chunk (lane&3) of position P holds c2 ^ ((P>>2)&3)
#pragma unroll
    for (int i = 0; i < N; i++) {
        const int j = w + 4 * i;
        const bool real = j < 2 * per_region;   // wave-uniform
        const int region = j >= per_region ? 1 : 0, kk = j - region * per_region;
        const int q = kk * 16 + (lane >> 2);
        const int row = q / L7_PITCH, tx = q - row * L7_PITCH;
        const int iy = iy0 + row, ix = X0 - 1 + tx;
        const bool ok = real && tx < TILE_X + 2 && iy >= iy_min && iy < iy_max && ix >= 0 && ix < IW;
        // region R keeps channel groups 2R, 2R+1 whole: chunk c2 = 16 bytes (c2&1) of group 2R + (c2>>1), so
        // a lane pair fetches one contiguous 32-byte group entry and a piece reads whole 512-byte runs
        const uint32_t off = ok ? tensor_offset(tm, iy, ix, (uint32_t)(2 * region + (c2 >> 1))) + 16u * (c2 & 1) : OOB;
        int slot = pslot0 + kk;
        slot = slot >= L7_RING_PIECES ? slot - L7_RING_PIECES : slot;
        uint8_t *dst = real ? patch + region * L7_REGION + slot * 1024 : scratch;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(dst), 16, off, 0, 0, L7_AUX);
    }
}

#ifdef SICN_EXP_L7_STAMP   // diagnostic build: cycles per phase of a step, summed over all waves and steps (s_memtime)
__device__ unsigned long long g_l7_stamp[8];
#define L7_T(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_[i] += t_ - tp_; tp_ = t_; }
#else
#define L7_T(i)
#endif
// SPLIT (round 4, VERDICT r3 item 4: "loader wave(s) issuing only LDS-DMA, consumer waves doing reads + MFMA + stores"): the
// workgroup has a FIFTH wave that issues every row request of a step (what the four waves shared, five apiece, and paid 941 of a
// step's 3954 cycles for: the requests wait for queue space) and waits for them to land; the four consumer waves never touch
// vmcnt for a load.  One barrier per step as before.  15 waves per CU need <= 128 VGPRs: the per-tap addressing constants are
// recomputed instead of kept (18 registers).
template <bool SPLIT>
__device__ __forceinline__ void l7_body(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, const int8_t *__restrict__ w_l7,
                                        const int8_t *__restrict__ bias, int IW, int IH, int OW, int OH, int steps_y, int y_chunks,
                                        int tiles_x, int n_images, int in_layout, int n_xcd)
{
    constexpr int CIN = 128;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *patch = smem;
    uint8_t *stage = smem + 2 * L7_REGION;
    uint8_t *scratch = stage + 4 * L7_STAGE;   // 1 KiB sink of the padding loads

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // logical work list: strip (fastest), chunk, image — an XCD gets whole rows of neighbouring strips
    const int item = xcd_logical_index(tiles_x * y_chunks * n_images, n_xcd);
    if (item < 0) return;
    const int bx = item % tiles_x, by = (item / tiles_x) % y_chunks, img = item / (tiles_x * y_chunks);
    const int X0 = bx * TILE_X;
    const int per = (steps_y + y_chunks - 1) / y_chunks;
    const int s_begin = by * per, s_end = min(steps_y, s_begin + per);
    if (s_begin >= s_end) return;  // before any LDS-DMA is issued

    // lane roles in v_mfma_i32_16x16x64_i8: A row / B column = lane & 15, K bytes 16*(lane>>4)..+15
    const int m = lane & 15, kg = lane >> 4;
    const bool loader = SPLIT && w == 4;        // wave-uniform
    const int wr = SPLIT ? (w & 3) : w;         // the consumer's row inside a step
    v4i wf[18];
    if (!loader) {
#pragma unroll
        for (int s = 0; s < 18; s++) wf[s] = *(const v4i *)(w_l7 + (s * 16 + m) * 64 + kg * 16);
    }
    const int b0 = bias[0], b1 = bias[1], b2 = bias[2];

    const int in_img_bytes = IH * IW * CIN;
    const uint8_t *in_img = in + (size_t)img * in_img_bytes;
    uint8_t *out_img = out + (size_t)img * OH * OW * 3;
    const int py = kg >> 1, px = kg & 1;
    uint8_t *my_stage = stage + wr * L7_STAGE;
    const bool fast_rows = ((OW * 3) & 3) == 0 && X0 + TILE_X <= IW;

    // window row r <-> input row 4*s_begin - 3 + r; rows 0,1 are never read (they only make the
    // prologue a whole number of pieces)
    const int iy_top = 4 * s_begin - 3;
    const int iy_max = min(IH, 4 * s_end + 1);   // last row this chunk reads
    const TensorMap tm = tensor_map(in_layout, CIN, IW, IH);
    if constexpr (SPLIT) {
        if (loader) {
#pragma unroll
            for (int vw = 0; vw < 4; vw++)
                l7_load_rows<9>(patch, scratch, in_img, in_img_bytes, vw, lane, 2 * L7_STEP_PIECES, iy_top, max(iy_top + 2, 0), iy_max, 0,
                                X0, IW, tm);
        }
    } else
        l7_load_rows<9>(patch, scratch, in_img, in_img_bytes, w, lane, 2 * L7_STEP_PIECES, iy_top, max(iy_top + 2, 0), iy_max, 0, X0,
                        IW, tm);
    // block k = window rows 4k+4 .. 4k+7 = the rows step k adds; blocks 1 .. AHEAD-1 start now
    int pnext = 2 * L7_STEP_PIECES;        // ring piece slot of the next block to be requested
    int ynext = iy_top + 8;                // its first input row
    static_assert(!SPLIT || L7_AHEAD == 1, "the loader wave keeps one step of rows in flight");
#pragma unroll
    for (int k = 1; k < L7_AHEAD; k++) {
        l7_load_rows<5>(patch, scratch, in_img, in_img_bytes, w, lane, L7_STEP_PIECES, ynext, 0, iy_max, pnext, X0, IW, tm);
        ynext += L7_ROWS;
        pnext += L7_STEP_PIECES;
        pnext = pnext >= L7_RING_PIECES ? pnext - L7_RING_PIECES : pnext;
    }

    // per-lane fragment addressing: P = (4t + 2 + w + dy)*36 + 16c + m + dx
    auto frag_a = [&](int tap) { return (uint32_t)((2 + wr + tap / 3) * L7_PITCH + m + tap % 3); };
    auto frag_s = [&](uint32_t a) { return (uint32_t)((kg & 1) * L7_REGION) + ((((uint32_t)(kg >> 1)) ^ ((a >> 2) & 3u)) << 4); };
    uint32_t fa[SPLIT ? 1 : 9], fs[SPLIT ? 1 : 9];
    if constexpr (!SPLIT) {
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            fa[tap] = frag_a(tap);
            fs[tap] = frag_s(fa[tap]);
        }
    }
    if constexpr (SPLIT) {
        if (loader) {
            // ---- the loader wave: a step's row requests, their landing, the step's barrier; nothing else ----------------------
            wait_vmcnt<0>();
            block_barrier();
            for (int s = s_begin; s < s_end; s++) {
#pragma unroll
                for (int vw = 0; vw < 4; vw++)
                    l7_load_rows<5>(patch, scratch, in_img, in_img_bytes, vw, lane, L7_STEP_PIECES, ynext, 0, iy_max, pnext, X0, IW, tm);
                ynext += L7_ROWS;
                wait_vmcnt<0>();     // the next step's rows have landed ...
                block_barrier();     // ... and the consumers are done with this step's
                pnext += L7_STEP_PIECES;
                pnext = pnext >= L7_RING_PIECES ? pnext - L7_RING_PIECES : pnext;
            }
            return;
        }
        block_barrier();             // consumers: the loader's prologue has landed
    } else {
        wait_vmcnt<5 * (L7_AHEAD - 1)>();
        block_barrier();
    }

    int base = 0;                          // (144 t) mod ring: ring slot of window row 4t, column 0
#ifdef SICN_EXP_L7_STAMP
    unsigned long long st_[6] = {0, 0, 0, 0, 0, 0}, tp_ = __builtin_amdgcn_s_memtime();
#endif
    for (int s = s_begin; s < s_end; s++) {
        const int Y = 4 * s;               // first input row of this step
        // always issued (rows past the chunk are zero fill): the counted waits below rely on it
        if constexpr (!SPLIT) {
            l7_load_rows<5>(patch, scratch, in_img, in_img_bytes, w, lane, L7_STEP_PIECES, ynext, 0, iy_max, pnext, X0, IW, tm);
            ynext += L7_ROWS;
        }
        L7_T(0)   // the row requests (address arithmetic + issue)

#ifdef SICN_EXP_L7_DMA_ONLY   // timing experiment (wrong bytes): the row requests, their counted wait and the barrier, nothing else
        wait_vmcnt<5 * (L7_AHEAD - 1)>();
        block_barrier();
        base += L7_ROWS * L7_PITCH;
        base = base >= L7_RING_POS ? base - L7_RING_POS : base;
        pnext += L7_STEP_PIECES;
        pnext = pnext >= L7_RING_PIECES ? pnext - L7_RING_PIECES : pnext;
        continue;
#endif
        v4i acc[2];
        acc[0] = acc[1] = v4i{b0, b1, b2, 0};  // C row 4*kg + r = phase kg, channel r
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const uint32_t fa_t = SPLIT ? frag_a(tap) : fa[SPLIT ? 0 : tap], fs_t = SPLIT ? frag_s(fa_t) : fs[SPLIT ? 0 : tap];
                uint32_t slot = (uint32_t)base + fa_t + 16u * c;
                slot = min(slot, slot - (uint32_t)L7_RING_POS);   // one wrap at most
                const uint32_t addr = slot * 64u + fs_t;
#if defined(SICN_EXP_L7_READS_ONLY)   // timing experiments (wrong bytes): the fragment reads without the MFMAs ...
                const v4i p0 = *(const v4i *)(patch + addr);
                const v4i p1 = *(const v4i *)(patch + (addr ^ 32u));
                acc[c][0] ^= p0[0] ^ p1[3];
                acc[c][1] ^= p0[1] ^ p1[2];
#elif defined(SICN_EXP_L7_MFMA_ONLY)    // ... and the MFMAs without the fragment reads
                acc[c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[tap * 2 + 0], wf[(tap * 2 + 5) % 18], acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[tap * 2 + 1], wf[(tap * 2 + 7) % 18], acc[c], 0, 0, 0);
                (void)addr;
#elif !defined(SICN_EXP_L7_NO_MFMA)
                const v4i p0 = *(const v4i *)(patch + addr);
                const v4i p1 = *(const v4i *)(patch + (addr ^ 32u));   // channels 64..127: c2 ^ 2
                acc[c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[tap * 2 + 0], p0, acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[tap * 2 + 1], p1, acc[c], 0, 0, 0);
#endif
            }
        }
        asm volatile("" :: "v"(acc[0]), "v"(acc[1]));
        L7_T(1)   // fragment reads + MFMAs issued
        // ---- epilogue ----------------------------------------------------------------------
        const int gy = Y + wr;
        if (fast_rows) {
            // stage [2 rows = py][64 pixels = 2*(16c+m)+px][3] and write the rows as dwords
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const uint32_t v = pack4_relu7(acc[c][0], acc[c][1], acc[c][2], 0);
                uint8_t *d = my_stage + py * 192 + (2 * (16 * c + m) + px) * 3;
                d[0] = (uint8_t)v;
                d[1] = (uint8_t)(v >> 8);
                d[2] = (uint8_t)(v >> 16);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private staging: no barrier needed
            __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)out_img, 0, OH * OW * 3, 0x00020000);
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int idx = k * 64 + lane;                   // dword index in the 2 x 48 tile
                const int row = idx >= 48 ? 1 : 0, col = idx - 48 * row;
                const bool ok = idx < 96 && gy < IH;
                const uint32_t v = *(const uint32_t *)(my_stage + (idx < 96 ? idx : 0) * 4);
                uint32_t off = ok ? (uint32_t)(((2 * gy + row) * OW + 2 * X0) * 3 + col * 4) : OOB;
#if SICN_EXP_L7_STORE == 2   // timing experiment (wrong bytes): the same stores, all into the first 256 KiB (no HBM write traffic)
                off = ok ? (off & 0x3FFFCu) : OOB;
#endif
#if SICN_EXP_L7_STORE != 1   // 1 = timing experiment without the stores
                __builtin_amdgcn_raw_buffer_store_b32(v, ro, off, 0, 0);
#endif
            }
            L7_T(2)   // pack, staging, stores issued
            // leave in flight: this step's 2 stores and the AHEAD-1 younger row blocks (5 loads + 2 stores each)
#if SICN_EXP_L7_STORE == 1
            wait_vmcnt<5 * (L7_AHEAD - 1)>();
#else
            if constexpr (SPLIT)
                wait_vmcnt<6>();     // a consumer's only outstanding memory operations are its stores: keep at most three steps' worth
            else
                wait_vmcnt<2 + 7 * (L7_AHEAD - 1)>();
#endif
        } else {
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const int gx = X0 + 16 * c + m;
                if (gy < IH && gx < IW) {
                    const uint32_t v = pack4_relu7(acc[c][0], acc[c][1], acc[c][2], 0);
                    uint8_t *dst = out_img + ((size_t)(2 * gy + py) * OW + 2 * gx + px) * 3;
                    dst[0] = (uint8_t)v;
                    dst[1] = (uint8_t)(v >> 8);
                    dst[2] = (uint8_t)(v >> 16);
                }
            }
            wait_vmcnt<0>();
        }
        L7_T(3)   // counted wait for the next rows
#ifndef SICN_EXP_L7_NO_BARRIER   // timing experiment only (races)
        block_barrier();   // next rows landed for every wave, this step's rows are free
#endif
        L7_T(4)   // barrier
        base += L7_ROWS * L7_PITCH;
        base = base >= L7_RING_POS ? base - L7_RING_POS : base;
        pnext += L7_STEP_PIECES;
        pnext = pnext >= L7_RING_PIECES ? pnext - L7_RING_PIECES : pnext;
    }
    wait_vmcnt<0>();   // the zero-fill tail of the prefetch must land before the LDS is released
#ifdef SICN_EXP_L7_STAMP
    if (lane == 0) {
        for (int i = 0; i < 5; i++) atomicAdd(&g_l7_stamp[i], st_[i]);
        atomicAdd(&g_l7_stamp[5], (unsigned long long)(s_end - s_begin));
    }
#endif
}
__global__ __launch_bounds__(256, L7_WGS) void k_l7(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, const int8_t *__restrict__ w_l7,
                                                    const int8_t *__restrict__ bias, int IW, int IH, int OW, int OH, int steps_y, int y_chunks,
                                                    int tiles_x, int n_images, int in_layout, int n_xcd)
{
    l7_body<false>(in, out, w_l7, bias, IW, IH, OW, OH, steps_y, y_chunks, tiles_x, n_images, in_layout, n_xcd);
}
#ifdef SICN_ALT_KERNELS   // measured 8 % slower than k_l7: ALT build only, the product library rejects l7_loader = 2
// five waves: four consumers + the loader; three workgroups per CU = 15 waves, i.e. four on three of the SIMDs: 128 VGPRs at most
__global__ __launch_bounds__(320) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_l7s(
    const uint8_t *__restrict__ in, uint8_t *__restrict__ out, const int8_t *__restrict__ w_l7, const int8_t *__restrict__ bias, int IW, int IH,
    int OW, int OH, int steps_y, int y_chunks, int tiles_x, int n_images, int in_layout, int n_xcd)
{
    l7_body<true>(in, out, w_l7, bias, IW, IH, OW, OH, steps_y, y_chunks, tiles_x, n_images, in_layout, n_xcd);
}
#endif

#ifdef SICN_EXP_L7_STAMP
extern "C" int sicn_debug_l7_stamps(unsigned long long *out8)   // reads and clears the sums
{
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_l7_stamp), sizeof z) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_l7_stamp), z, sizeof z) == hipSuccess ? 0 : -1;
}
#endif

size_t l7_bytes(int cin) { return (size_t)18 * 16 * 64 * (cin / 128); }

void pack_l7(const int8_t *w_okc, int cin, int8_t *dst)
{
    // w_okc: [3][25*cin]; dst: [tap 9][half 2][row 16][64 B], row = 4*phase + c
    for (int tap = 0; tap < 9; tap++)
        for (int half = 0; half < 2; half++)
            for (int row = 0; row < 16; row++) {
                const int ph = row >> 2, c = row & 3, py = ph >> 1, px = ph & 1;
                const int dy = tap / 3, dx = tap % 3, ky = 2 * dy - py, kx = 2 * dx - px;
                int8_t *t = dst + (((size_t)tap * 2 + half) * 16 + row) * 64;
                // K byte b of step `half` is read by lane group kg = b>>4 from LDS region kg&1, chunk
                // 2*half + (kg>>1): channel 64*(kg&1) + 32*half + 16*(kg>>1) + (b&15)   (see l7_load_rows)
                for (int b = 0; b < 64; b++) {
                    const int kg = b >> 4, ch = 64 * (kg & 1) + 32 * half + 16 * (kg >> 1) + (b & 15);
                    t[b] = (c < 3 && ky >= 0 && ky < 5 && kx >= 0 && kx < 5)
                               ? w_okc[(size_t)c * 25 * cin + (ky * 5 + kx) * cin + ch]
                               : 0;
                }
            }
}

hipError_t launch_l7(const LayerGeom &g, const sicn_weights &w, const uint8_t *in, uint8_t *out,
                     int n_images, hipStream_t stream, int in_layout, const sicn_options &o, const ChipGeom &chip)
{
    if (g.CIN != 128 || g.COUT != 3) return hipErrorInvalidValue;
    if ((size_t)g.IH * g.IW * g.CIN >= (size_t)OOB) return hipErrorInvalidValue;
    if ((size_t)g.OH * g.OW * 3 >= (size_t)OOB) return hipErrorInvalidValue;   // buffer-descriptor stores
    const int tiles_x = (g.IW + TILE_X - 1) / TILE_X, steps_y = (g.IH + L7_ROWS - 1) / L7_ROWS;
    // about two workgroups per CU in all (three would fit) — measured
    // r02 (tools/strip_sweep.py): 1080p x 1 best at 16 - 24 chunks of 30 strips, x 4 at 4 - 6 of 120, 4K x 8 at 1 of 480;
    // every extra chunk re-fetches six halo rows and re-loads the 18 KB of weights
    // (r03, re-measured on one 1080p image, 30 strips: 13 / 15 / 17 / 19 / 21 / 25 chunks -> 23 / 22 / 19 / 24 / 23 / 21 us: the best
    // cut is the one that stays just under TWO workgroups per CU, 510 of 512; the 640 of round 2 was the middle of a flat region)
    const int y_chunks = l7_chunks(tiles_x, n_images, steps_y, o.strip_chunks, chip);
    const size_t lds = 2 * L7_REGION + 4 * L7_STAGE + 1024;
    const dim3 grid(xcd_grid_size((long)tiles_x * y_chunks * n_images, chip.n_xcd));
#ifdef SICN_ALT_KERNELS
    if (o.l7_loader == 2) {   // the loader-wave form (k_l7s): see l7_body; 0 / 1 = the four-wave kernel
        hipError_t es = hipFuncSetAttribute((const void *)k_l7s, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (es != hipSuccess) return es;
        hipLaunchKernelGGL(k_l7s, grid, dim3(320), lds, stream, in, out, w.d_w_l7, w.d_bias, g.IW, g.IH, g.OW, g.OH, steps_y, y_chunks,
                           tiles_x, n_images, in_layout, chip.n_xcd);
        return hipGetLastError();
    }
#else
    if (o.l7_loader == 2) return hipErrorInvalidValue;   // k_l7s: ALT build only
#endif
    hipError_t e = hipFuncSetAttribute((const void *)k_l7, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_l7, grid, dim3(256), lds, stream, in, out, w.d_w_l7, w.d_bias, g.IW, g.IH, g.OW, g.OH, steps_y, y_chunks,
                       tiles_x, n_images, in_layout, chip.n_xcd);
    return hipGetLastError();
}

}  // namespace sicn
