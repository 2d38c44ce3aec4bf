// Launch planning: every number a launcher derives from the SIZE OF THE CHIP lives here, as pure host functions of a
// ChipGeom (compute units, XCDs) — nothing in csrc/ hard-wires 256 CUs / 8 XCDs any more.  chip_geom() reads the current
// device's hipDeviceProp_t once per device (sicn_abi.hip); the functions below are what tests/test_abi_load.py walks for
// 256, 128 and 32 CUs through sicn_debug_plan (no GPU needed).
//
// The rule (include/sicn.h, "Device"): the device must be a gfx950 part (gcnArchName starts with "gfx950"), else every entry
// point that would touch it returns SICN_ENODEV.  n_cu = multiProcessorCount.  n_xcd = the largest power of two <= 8 with at
// least 20 CUs per XCD (MI355X SPX: 256 CUs -> 8 XCDs of 32; DPX 128 -> 4; QPX 64 -> 2; CPX 32 -> 1; a part with fused-off CUs,
// e.g. 240, keeps its 8).  hipDeviceProp_t has no XCD count; the mapping only steers which tiles share an L2, never results.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace sicn {

struct ChipGeom {
    int n_cu;    // compute units of the device (partition) the stream runs on
    int n_xcd;   // XCDs = L2 domains; workgroups of a 1-D grid are dealt to them round-robin
};

inline int xcd_count_for(int n_cu)
{
    int n = 8;
    while (n > 1 && n_cu / n < 20) n >>= 1;
    return n;
}
inline ChipGeom chip_from_cus(int n_cu) { return ChipGeom{n_cu < 1 ? 1 : n_cu, xcd_count_for(n_cu)}; }

// XCD-aware work list (k_common.hpp xcd_logical_index is the device side): grids are padded to a multiple of n_xcd
inline unsigned xcd_grid_size(long n_items, int n_xcd) { return (unsigned)((n_items + n_xcd - 1) / n_xcd * n_xcd); }
// host mirror of the device mapping: logical item of workgroup `block`, or -1
inline long xcd_item_of(long block, long n_items, int n_xcd)
{
    const long per = (n_items + n_xcd - 1) / n_xcd;
    const long idx = (block % n_xcd) * per + block / n_xcd;
    return (block / n_xcd < per && idx < n_items) ? idx : -1;
}

// ---- the wide persistent kernels (k_mfma16x.hip): one resident workgroup per CU ---------------------------------------------
constexpr int WIDE_MIN_TILES_PER_CU = 4;
// workgroups of a wide launch: one per CU, at most one per tile, a multiple of n_xcd; grid_cap > 0 (tests) lowers it
inline unsigned wide_grid(long total_tiles, int grid_cap, const ChipGeom &c)
{
    long cap = c.n_cu / c.n_xcd * c.n_xcd;
    if (cap < c.n_xcd) cap = c.n_xcd;
    if (grid_cap > 0) {
        cap = (long)grid_cap / c.n_xcd * c.n_xcd;
        if (cap < c.n_xcd) cap = c.n_xcd;
    }
    const long need = (total_tiles + c.n_xcd - 1) / c.n_xcd * c.n_xcd;
    return (unsigned)(need < cap ? need : cap);
}
// the dynamic part of the wide kernels' tile deal (k_mfma16x.hip DealX: one ticket counter per XCD + one mailbox per workgroup in the call's
// workspace).  It pays where a workgroup walks many tiles (measured r05 on 256 CUs: 32 tiles each in layers 1 / 6 of 8 x 4K, - 13 and - 9 us);
// with 8 (layers 2 / 5) one tile is 12 % of a workgroup's work, nothing can be balanced and the look at the other XCDs' counters at the end
// only costs (+ 3 us): static there
constexpr int WIDE_DEAL_MAX_XCDS = 16, WIDE_DEAL_MAX_WORKGROUPS = 512, WIDE_DEAL_MIN_TILES_PER_WORKGROUP = 16;
inline bool wide_deal_pays(long total_tiles, unsigned grid, const ChipGeom &c)
{
    return grid <= (unsigned)WIDE_DEAL_MAX_WORKGROUPS && c.n_xcd <= WIDE_DEAL_MAX_XCDS && total_tiles >= (long)WIDE_DEAL_MIN_TILES_PER_WORKGROUP * grid;
}
// automatic choice of the wide form: from WIDE_MIN_TILES_PER_CU tiles (16 x 32) per CU on; the conv already from 3.5 rounds
// when the last round is at least 90 % full (measured r03 on 256 CUs: 1020 tiles 131 - 140 us against 144 - 149)
inline bool wide_automatic(long tiles_w, bool deconv, const ChipGeom &c)
{
    const long n = c.n_cu;
    const long rounds = (tiles_w + n - 1) / n;
    const bool full_rounds = tiles_w * 10 >= rounds * n * 9;
    return tiles_w >= n * WIDE_MIN_TILES_PER_CU || (!deconv && full_rounds && 2 * tiles_w >= 7 * n);
}

// ---- the 8 x 16 / 8 x 32 kernels (k_mfma16.hip, k_mfma16p.hip): two resident workgroups per CU ----------------------------
// 8 x 32 tiles from about 0.8 of one residency on (measured r03 on 256 CUs: the bar sat at 416 - 434 of 512 tiles)
inline bool narrow_tile_wanted(long tiles32, const ChipGeom &c) { return tiles32 * 16 < 13L * 2 * c.n_cu; }
// output-channel split of the 8 x 16 kernels: grids that leave half of the CUs without a workgroup (measured r02: <= 128 of 256)
inline bool split_n_automatic(long tiles16, const ChipGeom &c) { return tiles16 * 2 <= c.n_cu; }
// K split (input-channel group pairs over workgroups of blockIdx.z; every slice stores its partial output bytes, the last one to
// arrive adds them mod 256 and applies the activation — k_mfma16p.hip).  Built in round 4 for grids that, even after the
// output-channel split, leave most CUs idle — and MEASURED A LOSS on this chip, so it is never automatic (sicn_options.split_k > 1
// forces it; the parity suite runs it on every MFMA shape): a 256 x 256 image, layers 1 - 5, 25-pass chains instead of 50 / 75:
// 11.6 -> 23.2, 11.5 -> 13.6, 11.6 -> 15.1, 16.9 -> 20.4, 14.5 -> 19.2 us (profiles/r04_ksplit_small_configs.txt).  Halving a
// 50-pass chain saves 2.3 us; making one workgroup's stores visible to a workgroup on another XCD costs an L2 write-back, a
// device-scope compare-and-swap and an L2-bypassing read — three dependent trips to memory, more than the 4.5 us of a kernel
// boundary, and the write-backs of the 16 workgroups an XCD holds queue up behind one another.
// Returns the number of K slices (1 = no split); the kernels split into nq / 2 slices of one channel-group pair each.
inline int split_k_automatic(long /*workgroups_after_split_n*/, int /*nq*/, bool /*deconv*/, size_t /*out_bytes*/, const ChipGeom &)
{
    return 1;
}

// ---- layer 7 (k_l7): vertical strips of 32 input columns, cut into y_chunks runs ----------------------------------------------
// just under TWO workgroups per CU in all (measured r03: 510 of 512 best); every cut re-fetches six halo rows and the weights
inline int l7_chunks(int tiles_x, int n_images, int steps_y, int forced, const ChipGeom &c)
{
    long y = forced > 0 ? forced : (2L * c.n_cu) / ((long)tiles_x * n_images);
    if (y < 1) y = 1;
    if (y > steps_y) y = steps_y;
    return (int)y;
}

// ---- the RGB layer with the previous layer's activation (k_l7g): strips of 62 columns, steps of 4 rows, wgs_per_cu (1 or 2) workgroups per CU at a
// time.  Every cut costs about two steps (the prologue activates six rows, two rounds of items): the cut with the fewest
// step-times on the busiest CU, the smallest such
constexpr int L7G_PLAN_COLS = 62, L7G_PLAN_ROWS = 4;   // the geometry k_l7g.hip instantiates (static_assert there)
inline int l7g_chunks(int tiles_x, int n_images, int steps_y, int forced, int wgs_per_cu, const ChipGeom &c)
{
    if (forced > 0) return forced < steps_y ? forced : steps_y;
    const long strips = (long)tiles_x * n_images;
    int best = 1;
    long best_cost = -1;
    for (int y = 1; y <= steps_y && y <= 64; y++) {
        const long per = (steps_y + y - 1) / y, rounds = (strips * y + (long)wgs_per_cu * c.n_cu - 1) / ((long)wgs_per_cu * c.n_cu),
                   cost = rounds * (per + 2);
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            best = y;
        }
    }
    return best;
}

// ---- layer 0 (k_l0): vertical runs of at most `max_run` tiles per workgroup, about four workgroups per CU on small images -----
struct L0Cut { int y_chunks, ty_per; };
inline L0Cut l0_chunks(int tiles_x, int tiles_y, int n_images, int max_run, int forced, const ChipGeom &c)
{
    int y_chunks = (tiles_y + max_run - 1) / max_run;
    long want = (4L * c.n_cu + (long)tiles_x * n_images - 1) / ((long)tiles_x * n_images);
    if (forced > 0) want = forced;
    if (want > y_chunks) y_chunks = want > tiles_y ? tiles_y : (int)want;
    const int ty_per = (tiles_y + y_chunks - 1) / y_chunks;
    return L0Cut{(tiles_y + ty_per - 1) / ty_per, ty_per};
}

}  // namespace sicn
