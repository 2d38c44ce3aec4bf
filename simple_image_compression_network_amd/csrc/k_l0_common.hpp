// Layer 0 (RGB -> channels): the RGBX patch geometry and the raw-rows -> patch expansion shared by k_rgb.hip (k_l0, k_l0p) and
// k_l0g.hip (layer 0 with the GDN applied before its one store).
#pragma once
#include "k_common.hpp"

namespace sicn {

// =============================================================================================
// Layer 0.  K = 75 is re-laid as 5 rows (ky) x 32 bytes: the 3-byte pixels are expanded to RGBX
// dwords in LDS, so the 5 taps of one kernel row are 20 contiguous bytes starting at an 8-byte
// aligned address (8*x - 8); the K step is padded to 8 pixel slots = 32 bytes with ZERO WEIGHTS for
// slots 5..7 and for the X byte (whatever data sits there is multiplied by 0).
// Same work split as k_mfma: 8 x 32 output tile, wave w = rows 2w, 2w+1, all output channels.
// =============================================================================================
constexpr int L0_QUADS = 18;                  // 4-pixel groups per patch row: 72 slots >= 2*31+4+4
constexpr int L0_PITCH = L0_QUADS * 16;       // 288 bytes, 16-byte aligned rows
constexpr int L0_TY = 8;                      // output rows per tile (wave w = rows 2w, 2w+1)
constexpr int L0_ROWS = 2 * L0_TY + 3;        // 19 input rows per tile
constexpr int L0_NQUAD = L0_ROWS * L0_QUADS;  // 342 quads per tile, <= 2 per thread
constexpr int L0_PATCH = L0_ROWS * L0_PITCH;  // 5472 bytes
#ifndef SICN_L0_CHUNK
#define SICN_L0_CHUNK 9
#endif
constexpr int L0_CHUNK = SICN_L0_CHUNK;       // tiles a workgroup walks at most (its raw pixels are LDS-resident)
constexpr int L0_RAW_DW = 52;                 // dwords kept per input row: 67 pixels = 201 B + 3 B of alignment slack
constexpr int L0_RAW_ROWS = 2 * L0_TY * L0_CHUNK + 3;
constexpr int L0_RAW_BYTES = (L0_RAW_ROWS * L0_RAW_DW * 4 + 12 + 255) / 256 * 256;   // + the 3 dwords the last quad over-reads

// raw rows -> RGBX patch: quad = 4 consecutive pixels of one patch row = 12 bytes starting `sh`
// bytes into dwords 3g .. 3g+3 of the raw row (sh = byte phase of the row's first dword),
// re-aligned in registers; pixels outside the image become 0 (the raw row holds the neighbouring
// row's bytes there).  `raw` points at the raw row of patch row 0.
// ASM_STORE (k_l0p): the patch store is an asm statement.  With LDS-DMA in flight (the next run's raw rows) hipcc orders every LDS
// store it knows about behind the pending DMA with an s_waitcnt vmcnt(0) — which is also a wait for the tile's output stores; an
// asm store it does not track, and the barrier that publishes the patch waits for lgkmcnt(0) anyway.
template <bool ASM_STORE = false>
__device__ __forceinline__ void l0_expand(const uint8_t *raw, uint8_t *patch, int img_byte0, int quad, int Y0, int X0,
                                          int IW, int IH)
{
    if (quad >= L0_NQUAD) return;
    const int r = quad / L0_QUADS, g = quad - r * L0_QUADS;
    const int iy = 2 * Y0 - 2 + r, ix0 = 2 * X0 - 2 + 4 * g;
    const uint32_t sh = (uint32_t)(img_byte0 + (iy * IW + 2 * X0 - 2) * 3) & 3u;
    const uint32_t *src = (const uint32_t *)(raw + (r * L0_RAW_DW + 3 * g) * 4);
    const uint32_t d0 = src[0], d1 = src[1], d2 = src[2], d3 = src[3];   // g = 17 runs 3 dwords into the next row: zero-weight slots
    const uint32_t w0 = __builtin_amdgcn_alignbyte(d1, d0, sh);
    const uint32_t w1 = __builtin_amdgcn_alignbyte(d2, d1, sh);
    const uint32_t w2 = __builtin_amdgcn_alignbyte(d3, d2, sh);
    const bool row_ok = iy >= 0 && iy < IH;
    uint4 v;
    v.x = (row_ok && ix0 + 0 >= 0 && ix0 + 0 < IW) ? (w0 & 0xFFFFFFu) : 0u;
    v.y = (row_ok && ix0 + 1 >= 0 && ix0 + 1 < IW) ? (__builtin_amdgcn_alignbyte(w1, w0, 3) & 0xFFFFFFu) : 0u;
    v.z = (row_ok && ix0 + 2 >= 0 && ix0 + 2 < IW) ? (__builtin_amdgcn_alignbyte(w2, w1, 2) & 0xFFFFFFu) : 0u;
    v.w = (row_ok && ix0 + 3 >= 0 && ix0 + 3 < IW) ? (w2 >> 8) : 0u;
    if constexpr (ASM_STORE) {
        const uint32_t a = (uint32_t)(uintptr_t)LDS_PTR(patch + quad * 16);
        const v4i d = {(int)v.x, (int)v.y, (int)v.z, (int)v.w};
        asm volatile("ds_write_b128 %0, %1" ::"v"(a), "v"(d) : "memory");
    } else
        *(uint4 *)(patch + quad * 16) = v;
}

}  // namespace sicn
