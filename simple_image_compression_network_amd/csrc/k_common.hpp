// Device-side helpers shared by the MFMA kernels (k_mfma.hip, k_rgb.hip).
#pragma once
#include "sicn_internal.h"
#include "sicn_plan.h"

// Timing-experiment switches that make a kernel give WRONG results on purpose (what does the barrier / the wait / the pack cost?)
// live in the kernel sources because they must cut into the middle of a pass; none of them can reach a product build by
// accident: they refuse to compile without -DSICN_ALLOW_WRONG_RESULTS.
#if defined(SICN_EXP_NOWAIT) || defined(SICN_EXP_NO_PASS_BARRIER) || defined(SICN_X_NOBAR) || defined(SICN_X_NOWAIT) ||         \
    defined(SICN_XW_NOREAD) || defined(SICN_XW_NOPACK) || defined(SICN_XW_NOSTORE) || defined(SICN_XW_NOWAIT) ||                \
    defined(SICN_EXP_L7_DMA_ONLY) || defined(SICN_EXP_L7_READS_ONLY) || defined(SICN_EXP_L7_MFMA_ONLY) ||                      \
    defined(SICN_EXP_L7_NO_MFMA) || defined(SICN_EXP_L7_NO_BARRIER) || (defined(SICN_EXP_L7_STORE) && SICN_EXP_L7_STORE != 0)
#ifndef SICN_ALLOW_WRONG_RESULTS
#error "this switch builds kernels that give wrong results on purpose (timing experiments): add -DSICN_ALLOW_WRONG_RESULTS"
#endif
#endif

namespace sicn {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void *)(p))

// One sub-patch = (TILE_Y+2) x (TILE_X+2) positions x 32 channel bytes = 10880 B, filled by 1-KiB
// LDS-DMA pieces (one wave-wide 16-byte load each) and padded to 12 pieces so that every wave of a
// 4-wave workgroup issues the same number of pieces (3) per sub-patch.
constexpr int SUB_BYTES = PATCH_PIX * KSTEP;  // 10880
constexpr int SUB_PIECES = 12;
constexpr int SUB_ALLOC = SUB_PIECES * 1024;  // 12288
#ifndef SICN_RING
#define SICN_RING 4
#endif
#ifndef SICN_PF
#define SICN_PF 3
#endif
constexpr int RING = SICN_RING;               // weight-tile ring slots
constexpr int PF = SICN_PF;                   // weight tiles in flight ahead of the consumer
constexpr uint32_t OOB = 0x80000000u;         // beyond any image: the buffer range check returns 0

template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void block_barrier()
{
    // LDS-DMA results are ordered for readers by the issuing wave's vmcnt + this barrier; a
    // __syncthreads() here would add a full vmcnt(0) and drain the weight prefetch.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// relu7((v) mod 256) for four accumulators, packed little-endian into one dword — 5 VALU ops:
// the low byte of each accumulator is moved to the HIGH byte of a 16-bit lane (v_perm_b32), so the
// byte's sign is the lane's sign and v_pk_max_i16(x, 0) is the ReLU; a last v_perm_b32 gathers the
// four high bytes.
typedef short v2s __attribute__((ext_vector_type(2)));
// `floor2`: the two 16-bit lanes the bytes are max'ed against: 0 = the reference's ReLU; ACT_FLOOR_RAW (-32768 in
// both lanes) makes the max an identity, i.e. the lane BEFORE the ReLU (input of the GDN extension) — same instructions.
constexpr uint32_t ACT_FLOOR_RELU = 0u, ACT_FLOOR_RAW = 0x80008000u;
// the same kernel argument carries one more flag in a bit that is 0 in both floors: store the output non-temporal
constexpr uint32_t ACT_NT_STORE = 1u, ACT_FLOOR_MASK = 0x80008000u;
#ifndef SICN_NT_MIN_MB
#define SICN_NT_MIN_MB 128
#endif
// outputs that cannot stay in L2 / Infinity Cache anyway (>= 128 MiB per launch) bypass them
__host__ inline bool nt_store_wanted(size_t out_bytes) { return out_bytes >= ((size_t)SICN_NT_MIN_MB << 20); }
__device__ __forceinline__ uint32_t pack4_relu7(int a, int b, int c, int d, uint32_t floor2 = ACT_FLOOR_RELU)
{
    const uint32_t ab = __builtin_amdgcn_perm((uint32_t)b, (uint32_t)a, 0x040c000cu);  // [0, a.b0, 0, b.b0]
    const uint32_t cd = __builtin_amdgcn_perm((uint32_t)d, (uint32_t)c, 0x040c000cu);
    const v2s z = __builtin_bit_cast(v2s, floor2);
    const v2s mab = __builtin_elementwise_max(__builtin_bit_cast(v2s, ab), z);
    const v2s mcd = __builtin_elementwise_max(__builtin_bit_cast(v2s, cd), z);
    return __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, mcd), __builtin_bit_cast(uint32_t, mab), 0x07050301u);
}

// Tensor layouts.
//   LAYOUT_NHWC  : [H][W][C]                 the ABI layout = the reference's stream bytes
//   LAYOUT_GROUP : [C/32][H][W][32]          internal, between two layers of a chain: every 32-channel
//                  group is its own plane, so a kernel that walks K group by group touches each
//                  128-byte line once and a lane-per-pixel epilogue stores 1 KiB per wave instruction
//   LAYOUT_PHASE : [y&1][x&1][C/32][H/2][W/2][32]   GROUP split by pixel parity: what a deconv
//                  phase (py,px) produces is then CONTIGUOUS (its outputs are the pixels of one
//                  parity), instead of 32-byte pieces at a 64-byte stride
constexpr int LAYOUT_NHWC = 0, LAYOUT_GROUP = 1, LAYOUT_PHASE = 2;

// byte offset of channel group g of pixel (y, x) in an image of W x H pixels, C channels
__device__ __forceinline__ uint32_t tensor_offset(int layout, int y, int x, uint32_t g, int C, int W, int H)
{
    if (layout == LAYOUT_NHWC) return (uint32_t)(y * W + x) * (uint32_t)C + g * 32u;
    if (layout == LAYOUT_GROUP) return (g * (uint32_t)(W * H) + (uint32_t)(y * W + x)) * 32u;
    const uint32_t hw = (uint32_t)((W >> 1) * (H >> 1));
    const uint32_t plane = (uint32_t)((y & 1) * 2 + (x & 1)) * (uint32_t)(C >> 5) + g;
    return (plane * hw + (uint32_t)((y >> 1) * (W >> 1) + (x >> 1))) * 32u;
}

// The same three layouts as strides (wave-uniform), for kernels that compute many offsets per step:
//   offset = ((y>>sh)*row + (x>>sh))*pix + g*grp + ((y&sh)*2 + (x&sh))*ph          (no branches)
struct TensorMap {
    uint32_t sh, row, pix, grp, ph;
};
__device__ __forceinline__ TensorMap tensor_map(int layout, int C, int W, int H)
{
    if (layout == LAYOUT_NHWC) return TensorMap{0u, (uint32_t)W, (uint32_t)C, 32u, 0u};
    if (layout == LAYOUT_GROUP) return TensorMap{0u, (uint32_t)W, 32u, (uint32_t)(W * H) * 32u, 0u};
    const uint32_t hw = (uint32_t)((W >> 1) * (H >> 1));
    return TensorMap{1u, (uint32_t)(W >> 1), 32u, hw * 32u, (uint32_t)(C >> 5) * hw * 32u};
}
__device__ __forceinline__ uint32_t tensor_offset(const TensorMap &t, int y, int x, uint32_t g)
{
    return (((uint32_t)y >> t.sh) * t.row + ((uint32_t)x >> t.sh)) * t.pix + g * t.grp +
           (((uint32_t)y & t.sh) * 2u + ((uint32_t)x & t.sh)) * t.ph;
}

// Source offset (bytes from the image base) of this lane's 16 bytes of LDS-DMA piece `k` of a
// sub-patch: position p = k*32 + lane/2 of the (TILE_Y+2) x (TILE_X+2) window whose origin is
// (Yb, Xb) in "patch coordinates"; patch coordinate (ty,tx) maps to input pixel
// (s*(Yb+ty)+ay, s*(Xb+tx)+ax).  The LDS image keeps logical half h of position p at physical
// half h ^ ((p>>3)&1) (bank-conflict-free ds_read_b128).
__device__ __forceinline__ uint32_t piece_src_offset(int k, int lane, int Yb, int Xb, int s, int ay, int ax,
                                                     int IW, int IH, int layout, uint32_t g, int C, bool swizzle = true)
{
    const int p = k * 32 + (lane >> 1);
    const int ty = p / PATCH_X, tx = p - ty * PATCH_X;
    // 32x32x32 fragments want half h of position p at h ^ ((p>>3)&1); 16x16x64 fragments want it plain
    const int hlog = swizzle ? (lane & 1) ^ ((p >> 3) & 1) : (lane & 1);
    const int iy = s * (Yb + ty) + ay, ix = s * (Xb + tx) + ax;
    const bool ok = p < PATCH_PIX && iy >= 0 && iy < IH && ix >= 0 && ix < IW;
    return ok ? tensor_offset(layout, iy, ix, g, C, IW, IH) + hlog * 16 : OOB;
}

// Tile geometry of the 16x16x64 kernels: TILE_Y x TX positions (TX = 32, or 16 for the layers whose
// accumulators / patch allow only one 8 x 32 workgroup per CU), sub-patch of (TILE_Y+2) x (TX+2)
// positions x 32 B, padded to SLOTS pieces per wave so that every wave issues the same number.
template <int TX>
struct Geo {
    static constexpr int PX = TX + 2;                              // patch row pitch in positions
    static constexpr int PIX = PATCH_Y * PX;                       // 340 / 180 positions
    static constexpr int SLOTS = (PIX * KSTEP + 4095) / 4096;      // LDS-DMA pieces per wave per sub-patch: 3 / 2
    static constexpr int ALLOC = SLOTS * 4 * 1024;                 // 12288 / 8192
    static constexpr int XT = TX / 16;                             // 16-position column tiles per row
    static constexpr int NC = 2 * XT;                              // column tiles per wave (2 rows)
};

// The same with the layout as strides: patch position and validity of this lane in piece `k`
// (unswizzled image, 16x16x64 kernels), channel group 0; group g is `+ g * tm.grp`.
struct PieceSrc {
    uint32_t off;   // byte offset of channel group 0 (meaningless if !ok)
    bool ok;
};
template <int TX>
__device__ __forceinline__ PieceSrc piece_src(const TensorMap &tm, int k, int lane, int Yb, int Xb, int s, int ay, int ax,
                                              int IW, int IH)
{
    const int p = k * 32 + (lane >> 1);
    const int ty = p / Geo<TX>::PX, tx = p - ty * Geo<TX>::PX;
    const int iy = s * (Yb + ty) + ay, ix = s * (Xb + tx) + ax;
    PieceSrc r;
    r.ok = p < Geo<TX>::PIX && iy >= 0 && iy < IH && ix >= 0 && ix < IW;
    r.off = tensor_offset(tm, iy, ix, 0u) + (uint32_t)(lane & 1) * 16u;
    return r;
}

// one LDS-DMA piece (1 KiB) of sub-patch `sub` (sub-patches are ALLOC bytes apart)
template <int ALLOC = SUB_ALLOC>
__device__ __forceinline__ void load_piece(uint8_t *patch, const uint8_t *in_img, int in_img_bytes, int sub,
                                           int k, uint32_t off)
{
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)in_img, 0, in_img_bytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(patch + sub * ALLOC + k * 1024), 16, off, 0, 0, 0);
}

// XCD-aware work mapping.  Workgroups of a 1-D grid are dealt to the 8 XCDs round-robin (linear id % 8)
// and each XCD has its own L2, so neighbouring tiles / strips (which share halo pixels and the 128-byte
// lines at their common edge) should run on the SAME XCD at about the same time: XCD x works through
// the contiguous range [x * per, (x+1) * per) of the logical work list.  Returns the logical index of
// this workgroup, or -1 if it has none (the grid is padded to a multiple of 8).
// n_xcd is a kernel argument: the launcher takes it from the device (sicn_plan.h, chip_geom()); host and device only have to agree.
__device__ __forceinline__ int xcd_logical_index(int n_items, int n_xcd)
{
    const int per = (n_items + n_xcd - 1) / n_xcd;
    const int l = (int)blockIdx.x, idx = (l % n_xcd) * per + l / n_xcd;
    return (l / n_xcd < per && idx < n_items) ? idx : -1;
}

// ---- conv tap order: by parity plane (a,b) = (ky&1, kx&1), so that a plane's LDS buffer is
// ---- free for the next channel group as soon as its taps are done (k_mfma.hip) -------------
struct Tap { int ky, kx; };
__host__ __device__ constexpr Tap conv_tap(int t)
{
    int n = 0;
    for (int a = 0; a < 2; a++)
        for (int b = 0; b < 2; b++)
            for (int ky = a; ky < 5; ky += 2)
                for (int kx = b; kx < 5; kx += 2) {
                    if (n == t) return Tap{ky, kx};
                    n++;
                }
    return Tap{-1, -1};
}

}  // namespace sicn
