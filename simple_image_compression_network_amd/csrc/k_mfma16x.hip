// conv2d<> / deconv522<> 128 -> 128 (layers 1, 2 / 5, 6: 70 % of the step) — the WIDE PERSISTENT form (round 3).
//
// Where k_mfma16p.hip stands: 2 workgroups x 4 waves per CU, 64 positions x 128 channels per wave, 12 fragment
// reads per 32 MFMAs, every workgroup streams its own copy of the weights; the passes are issue-bound and a bare loop of that
// structure tops out at 3.44 POP/s (tools/microbench/wide_dma.hip, profiles/r03_microbench_wide_dma.txt).  The same
// microbenchmark says what the alternative keeps once it carries a real pass's load: ONE wave per SIMD with 128 positions x 128
// channels per wave (256 accumulators pinned in AGPRs, 16 fragment reads per 64 MFMAs), 4 LDS-DMA requests, a counted wait and a
// workgroup barrier per pass: 3.75 POP/s, +9 %.  Round 2's k_conv128w had that tile but lost it all in its per-tile prologue and
// epilogue, which nothing covers when a SIMD holds a single wave.  Here
//   * one workgroup of 4 waves per CU owns 16 x 32 positions (wave w = rows 4w .. 4w+3) and is PERSISTENT: it walks through its
//     share of the tile list, the input of the next tile arrives under the current one, the weight stream wraps: no prologue;
//   * the ACCUMULATOR HAND-OVER (read, bias, ReLU, pack, store of a finished tile / deconv phase) is woven into the first pass
//     of the next one: that pass's MFMAs take C = 0 and overwrite an accumulator right after it has been read out (in-kernel
//     stamps: done as a block between two tiles the hand-over was 6350 cycles = 10 % of a conv tile, with a single wave per SIMD
//     every VALU instruction costs 4 issue cycles).  The bias is added in the pack (v_pk_add_u16 on the byte that sits in the
//     high half of a 16-bit lane), so the accumulators are never initialised at all;
//   * the tile list is dealt STATICALLY (XCD x works through the contiguous range [x * per, (x + 1) * per) of the list, its 32
//     workgroups interleaved): with one workgroup per CU all tiles take the same time, so there is no scheduler state at all;
//   * one weight ring per CU (not two): half the weight requests per MFMA; a 16 x 32 tile has 10 % less halo than two 8 x 32;
//   * every LDS read is an asm statement with an immediate offset (no address arithmetic in the loop) and hand-placed waits, the
//     MFMAs are asm statements with the accumulator tied to an AGPR: the order written is the order issued;
//   * the ring has 10 slots = 5 passes: a 50-pass tile is 0 mod 5, so every slot and every offset of the unrolled tile is a
//     compile-time constant, and 3 passes' requests may be in flight at a barrier.
// Conv:   4 parity planes of one channel group (18 x 34 positions x 32 B, 20 KiB each) with a rolling refresh + ring = 120 KB.
// Deconv: the 18 x 34 patch with all 4 channel groups, as 2 group PAIRS in 3 rotating 40-KiB buffers (the next tile's first pair
//         lands in the spare buffer, its second pair in the buffer the current tile's last phase frees first) + ring = 160 KB.
// Hazards hipcc cannot see inside asm are covered by hand and CHECKED: tools/isa_hazards.py runs over this file's ISA in build().
#include <algorithm>

#include "k_common.hpp"

namespace sicn {
namespace xw {

constexpr int TY = 16, TX = 32, PY = TY + 2, PX = TX + 2, PIX = PY * PX;   // 18 x 34 = 612 patch positions per plane / group
constexpr int PIECES = (PIX * KSTEP + 1023) / 1024;                       // 20 LDS-DMA pieces of 1 KiB
constexpr int SLOTS = PIECES / 4;                                         // 5 per wave
static_assert(PIECES % 4 == 0, "every wave issues the same number of pieces");
constexpr int PLANE = PIECES * 1024;                                      // 20480
constexpr int RING = 10, TB = 128 * KSTEP;                                // 10 slots x 4 KiB: one K step of 128 output channels
constexpr int FLIGHT = 3;                                                 // passes whose requests may be in flight at a barrier
constexpr int NSTORE = 16;                                                // output stores per wave and accumulator hand-over
constexpr int NPASS = 50;                                                 // passes (of two K steps) per tile

typedef __attribute__((address_space(3))) uint8_t lds_u8;   // LDS pointers stay 32-bit (a generic pointer costs a null check per cast)

#define SICN_MFMA_A(ACC, A, B) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+a"(ACC) : "v"(A), "v"(B))
// C = bias: the accumulator starts here.  The D operand is tied all the same ("+a"): the new value must live in the AGPRs of the old
// one, and whatever still wants the old value (the hand-over's reads) is thereby ordered in front of this statement.  (With an
// output-only operand hipcc gave the new values other registers and moved them through VGPRs with v_accvgpr_read right behind the
// MFMA — inside the MFMA's latency, where the hazard recogniser does not look for an asm statement: wrong bytes.)
#define SICN_MFMA_AC(ACC, A, B, C) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %3" : "+a"(ACC) : "v"(A), "v"(B), "a"(C))
// accumulator tiles j = 7 (8 of the 64) live in VGPRs: that leaves 28 AGPRs for the bias of tiles j = 0 .. 6, which the first pass
// of a tile / phase takes as its C operand (C and D of an MFMA must sit in the same register file), and those 8 tiles need no
// v_accvgpr_read at the hand-over
#define SICN_MFMA_V(ACC, A, B) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B))
#define SICN_MFMA_VC(ACC, A, B, C) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %3" : "+v"(ACC) : "v"(A), "v"(B), "v"(C))
constexpr int JV = 7;   // accumulator tiles j >= JV live in VGPRs

template <int OFF>
__device__ __forceinline__ v4i lds_read(uint32_t addr)
{
    static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field");
    v4i v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}

#ifdef SICN_STAMP   // diagnostic build only (tools/x_stamps.py): per workgroup {cycles, 100 MHz ticks, tiles, hand-over cycles, start, HW_ID, XCC_ID}
__device__ unsigned long long *g_sicn_stamp_x = nullptr;
extern "C" int sicn_debug_stamp_buffer_x(void *p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_sicn_stamp_x), &p, sizeof p); }
#endif

// ---- what an accumulator hand-over needs ------------------------------------------------------------------------------------
struct HandX {
    __amdgpu_buffer_rsrc_t ro;   // the output image
    uint32_t off[8];             // per pixel fragment: byte offset of this lane's 16 bytes of channel group g >> 1 (OOB outside)
    uint32_t soff;               // scalar part of the offset (deconv: the phase's parity plane / pixel)
    uint32_t grp2;               // 2 channel groups = 64 channels on: the second store of a pixel fragment
    uint32_t floor2;             // the pack's floor as a signed value: 0 (ReLU) / -128 (identity)
};

// relu7(v mod 256) of four accumulators packed into one dword by four SDWA instructions: max(sign-extended byte 0 of the
// accumulator, floor) written to byte k of the result (floor 0 = the reference's ReLU, -128 = identity: the lane before the ReLU
// for the GDN extension).  One instruction per accumulator register; pack4_relu7 (perm, perm, max, max, perm) takes 5 per 4.
__device__ __forceinline__ uint32_t pack4_sdwa(const v4i &a, int floor32)
{
    uint32_t out;
    asm("v_max_i32_sdwa %0, sext(%1), %2 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(out) : "v"(a[0]), "s"(floor32));
    asm("v_max_i32_sdwa %0, sext(%1), %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:DWORD" : "+v"(out) : "v"(a[1]), "s"(floor32));
    asm("v_max_i32_sdwa %0, sext(%1), %2 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:DWORD" : "+v"(out) : "v"(a[2]), "s"(floor32));
    asm("v_max_i32_sdwa %0, sext(%1), %2 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:DWORD" : "+v"(out) : "v"(a[3]), "s"(floor32));
    return out;
}

// this lane's bias as the first pass's C operand: register r of accumulator tile j is channel 64 (j >> 2) + 16 g + 4 (j & 3) + r.
// (Until the accumulators moved: added in the pack, v_pk_add_u16 on the byte in the high half of a 16-bit lane — 128 VALU
// instructions per hand-over at 3.2 issue cycles each.)
struct BiasX {
    v4i c[8];
};
__device__ __forceinline__ void load_bias_x(BiasX &b, const int8_t *bias, int g)
{
#pragma unroll
    for (int J = 0; J < 2; J++) {
        const v4i b4 = *(const v4i *)(bias + 64 * J + 16 * g);
#pragma unroll
        for (int d = 0; d < 4; d++) {
#pragma unroll
            for (int r = 0; r < 4; r++) b.c[4 * J + d][r] = (int)(int8_t)((uint32_t)b4[d] >> (8 * r));
            if (4 * J + d < JV)
                asm volatile("" : "+a"(b.c[4 * J + d]));
            else
                asm volatile("" : "+v"(b.c[4 * J + d]));
        }
    }
}

// ---- one pass: 64 MFMAs on the current fragments (pc, wc); rd(r) issues read r (0 .. 15) of the next pass's fragments, dma(k)
// ---- request k (0 .. 3) of this pass ----------------------------------------------------------------------------------------
//   KIND 0: accumulate.   KIND 2: the accumulators start here (C = 0), and every accumulator is read out, biased, packed and
//   stored (hand-over of the tile / phase that has just ended) right before the MFMA that overwrites it.
//   VM: requests that may stay in flight at the barrier; EXTRA: plus the hand-over's stores while `stores` (they are younger than
//   what the wait is after during the first FLIGHT passes behind a hand-over: counted, not waited for)
//   (Tried and removed: keeping the packed results in VGPRs and storing them two per pass behind the hand-over.  Once the
//   bias registers were gone it fitted — 204 - 220 VGPRs, no scratch — and the hand-over pass did get 400 cycles shorter, but the
//   deconv tile got 1300 cycles LONGER: a store costs its ~80 issue cycles wherever it sits.  The cost model is additive.)
template <int KIND, int VM, int EXTRA, bool NT, class Rd, class Dma>
__device__ __forceinline__ void pass_x(v4i (&acc)[8][8], const v4i (&pc)[8], const v4i (&wc)[8], Rd rd, Dma dma, bool stores, const HandX &h,
                                       const BiasX &bias)
{
    if constexpr (KIND == 0) {
        int issued = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int m = j * 8 + i;
                if (j < JV)
                    SICN_MFMA_A(acc[i][j], wc[j], pc[i]);
                else
                    SICN_MFMA_V(acc[i][j], wc[j], pc[i]);
                // the 16 reads of the next pass: one behind every third MFMA, the last one behind MFMA 47
                if (m % 3 == 2 && issued < 16) {
                    switch (issued) {
#define SICN_R(R) case R: rd(std::integral_constant<int, R>{}); break;
                        SICN_R(0) SICN_R(1) SICN_R(2) SICN_R(3) SICN_R(4) SICN_R(5) SICN_R(6) SICN_R(7)
                        SICN_R(8) SICN_R(9) SICN_R(10) SICN_R(11) SICN_R(12) SICN_R(13) SICN_R(14) SICN_R(15)
#undef SICN_R
                    }
                    issued++;
                }
                if (m == 4) dma(std::integral_constant<int, 0>{});
                if (m == 20) dma(std::integral_constant<int, 1>{});
                if (m == 36) dma(std::integral_constant<int, 2>{});
                if (m == 52) dma(std::integral_constant<int, 3>{});
            }
        }
    } else {
        // 16 regions of 4 MFMAs: pixel fragment i, accumulator tiles 4 J .. 4 J + 3 = 16 consecutive channels of 16 positions.
        // (Reading region k + 1's accumulators out under region k's packs — a software pipeline over the regions — measured
        // nothing: the pass is bound by the number of VALU instructions, 3.2 cycles each, not by their dependences.)
        __builtin_amdgcn_sched_barrier(0);   // nothing of the hand-over moves up into the previous pass
#pragma unroll
        for (int i = 0; i < 8; i++) {
#pragma unroll
            for (int J = 0; J < 2; J++) {
                const int k = 2 * i + J;
                v4i old[4];
#ifndef SICN_XW_NOREAD
#pragma unroll
                for (int d = 0; d < 4; d++) old[d] = acc[i][4 * J + d];   // read out (v_accvgpr_read) in front of the tied MFMA below
#endif
                v4i v;
                // a tile that lives in VGPRs is packed straight from its registers, BEFORE the tied MFMA overwrites them (packed
                // behind it, hipcc would copy the four registers first)
#pragma unroll
                for (int d = 0; d < 4; d++)
                    if (4 * J + d >= JV) {
                        v[d] = (int)pack4_sdwa(old[d], (int)h.floor2);
                        asm volatile("" : "+v"(v[d]));
                    }
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    if (4 * J + d < JV)
                        SICN_MFMA_AC(acc[i][4 * J + d], wc[4 * J + d], pc[i], bias.c[4 * J + d]);
                    else
                        SICN_MFMA_VC(acc[i][4 * J + d], wc[4 * J + d], pc[i], bias.c[4 * J + d]);
                }
#if defined(SICN_XW_NOPACK)    // timing experiments (wrong results): what of the hand-over costs what
#pragma unroll
                for (int d = 0; d < 4; d++) v[d] = old[d][0] ^ old[d][1] ^ old[d][2] ^ old[d][3];
#elif defined(SICN_XW_NOREAD)
                v = v4i{1, 2, 3, 4};
#else
#pragma unroll
                for (int d = 0; d < 4; d++)
                    if (4 * J + d < JV) v[d] = (int)pack4_sdwa(old[d], (int)h.floor2);
#endif
#ifdef SICN_XW_NOSTORE
                asm volatile("" ::"v"(v));
#else
                __builtin_amdgcn_raw_buffer_store_b128(v, h.ro, h.off[i], h.soff + (uint32_t)J * h.grp2, NT ? 2 : 0);
#endif
                switch (k) {
#define SICN_R(R) case R: rd(std::integral_constant<int, R>{}); break;
                    SICN_R(0) SICN_R(1) SICN_R(2) SICN_R(3) SICN_R(4) SICN_R(5) SICN_R(6) SICN_R(7)
                    SICN_R(8) SICN_R(9) SICN_R(10) SICN_R(11) SICN_R(12) SICN_R(13) SICN_R(14) SICN_R(15)
#undef SICN_R
                }
                if (k == 1) dma(std::integral_constant<int, 0>{});
                if (k == 5) dma(std::integral_constant<int, 1>{});
                if (k == 9) dma(std::integral_constant<int, 2>{});
                if (k == 13) dma(std::integral_constant<int, 3>{});
                __builtin_amdgcn_sched_barrier(0);   // a region's reads / packs / store stay with its MFMAs
            }
        }
    }
#if defined(SICN_XW_NOWAIT)   // timing experiment (wrong results): never wait for a request
    wait_vmcnt<48>();
#else
#ifdef SICN_XW_NOSTORE
    if (false)
#else
    if (EXTRA > 0 && stores)
#endif
        wait_vmcnt<VM + EXTRA>();
    else
        wait_vmcnt<VM>();
#endif
    block_barrier();   // lgkmcnt(0) (this wave's reads of the next pass are complete) + s_barrier
}

struct TileX {
    int img, Y0, X0;
};

// Workgroups of a persistent launch start together and take the same time per tile: without help all 256 CUs reach their
// accumulator hand-overs at the same moment and 16 MB of output stores hit the memory system in one burst (in-kernel stamps:
// a hand-over pass took 5600 - 8300 cycles against ~2900 of instruction issue).  Each workgroup therefore starts a different
// fraction of SICN_XW_STAGGER x 2048 cycles late (slot s of an XCD's 32: s / 32 of it), which spreads the bursts for good.
#ifndef SICN_XW_STAGGER
#define SICN_XW_STAGGER 8
#endif
__device__ __forceinline__ void stagger_x(int n_xcd)
{
    const int slot = ((int)blockIdx.x / n_xcd) & 31;
    for (int i = 0; i < slot * SICN_XW_STAGGER; i++) __builtin_amdgcn_s_sleep(1);   // 64 cycles each
}

// per-fragment output offsets of a tile: pixel fragment c = row 4 w + (c >> 1), column tile c & 1 of the 16 x 32 positions;
// (MW, MH) bounds the M grid (conv: output pixels, scale 1; deconv: input pixels, whose outputs are (2 y + py, 2 x + px))
__device__ __forceinline__ void set_out_off_x(uint32_t (&off)[8], const TensorMap &om, const TileX &t, int w, int lane, int MW, int MH, int scale)
{
    const int pos = lane & 15, g = lane >> 4;
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const int gy = t.Y0 + 4 * w + (c >> 1), gx = t.X0 + (c & 1) * 16 + pos;
        off[c] = (gy < MH && gx < MW) ? tensor_offset(om, scale * gy, scale * gx, (uint32_t)(g >> 1)) + 16u * (uint32_t)(g & 1) : OOB;
    }
}

// ---- the tile deal across the XCDs (round 5) --------------------------------------------------------------------------------
// The eight XCDs do not hold the same clock under this load (in-kernel stamps: 1.92 - 2.08 GHz, the same XCDs slow in every layer of a
// pass), so with the same number of tiles per XCD the kernel ends 3.3 - 4.5 % after its average workgroup.  With a `deal` area (words of
// the net's workspace, zeroed once per forward pass) a workgroup walks TWO tiles of the static deal and takes every further one from a
// ticket counter: its own XCD's (the contiguous range of §3.1 stays contiguous) and, once that is used up, the other XCDs' in turn.
//   words [0, 16): ticket counters, one per XCD (low 32 bits);  word 16 + blockIdx.x: the workgroup's mailbox = (sequence << 32) | (item + 1)
// Wave 0 takes the ticket for the workgroup's tile j + 2 while tile j runs (one lane, an asm atomic nobody waits for: its result is read
// six passes later, and checked), resolves it and publishes it in the mailbox; ALL four waves read the mailbox with a scalar load early in
// tile j + 1 — complete at that pass's barrier (lgkmcnt(0)), no LDS word needed (k_deconv_x has none to give) — and spin only if the
// sequence number is not there yet.  deal == nullptr (a single layer outside a net): the static deal, as before.
struct DealX {
    unsigned long long ws;   // the deal area's address, or 0: the static deal.  Kept in a VGPR pair (the kernels have no SGPR to spare)
    int xcd, n_xcd, per, stride, total;
};
constexpr int DEAL_MAILBOX0 = 16;
constexpr uint32_t DEAL_PENDING = 0xFFFFFFFFu;

// item of ticket t of XCD y, or -1: tickets count the tiles BEHIND the two static rounds of y's range
__device__ __forceinline__ int deal_item(const DealX &d, int y, uint32_t t)
{
    const long idx = (long)y * d.per + 2L * d.stride + (long)t;
    const long end = min((long)d.total, (long)(y + 1) * d.per);
    return (t < 0x7fffffffu && idx < end) ? (int)idx : -1;
}
// wave 0: one lane takes a ticket from XCD y's counter; the value arrives in tk some time later (DEAL_PENDING until then)
__device__ __forceinline__ void deal_ticket_issue(const DealX &d, int y, uint32_t &tk, int lane)
{
    tk = DEAL_PENDING;
    asm volatile("" : "+v"(tk));
    if (lane == 0) {
        const unsigned long long p = d.ws + (unsigned long long)y * 8u;
        const uint32_t one = 1u;
        asm volatile("global_atomic_add %0, %1, %2, off sc0" : "+v"(tk) : "v"(p), "v"(one) : "memory");
    }
}
__device__ __forceinline__ uint32_t deal_ticket_value(uint32_t &tk)   // wave 0; blocks only if the atomic has not come back
{
    asm volatile("" : "+v"(tk));
    uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)tk);
    if (t == DEAL_PENDING) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" : "+v"(tk));
        t = (uint32_t)__builtin_amdgcn_readfirstlane((int)tk);
    }
    return t;
}
// wave 0: the ticket (of the own XCD) -> an item.  Own range used up: ONE look at all counters (lane y reads XCD y's), then a ticket from
// the XCD with the most tiles left — two round trips (~ 3 us, once in a workgroup's life) instead of up to seven blind ones (measured:
// + 10 us on every launch); a lost race looks again.
__device__ __forceinline__ int deal_resolve(const DealX &d, uint32_t &tk, int lane)
{
    uint32_t t = deal_ticket_value(tk);
    int it = deal_item(d, d.xcd, t);
    if (it >= 0) return it;
#pragma unroll 1
    for (int attempt = 0; attempt < 4; attempt++) {
        uint32_t c = 0x7fffffffu;
        asm volatile("" : "+v"(c));
        if (lane < d.n_xcd) {
            const unsigned long long p = d.ws + (unsigned long long)lane * 8u;
            asm volatile("global_load_dword %0, %1, off sc0 sc1" : "+v"(c) : "v"(p) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" : "+v"(c));
        int best = -1;
        long best_left = 0;
#pragma unroll 1
        for (int y = 0; y < d.n_xcd; y++) {
            const long taken = (long)(uint32_t)__builtin_amdgcn_readlane((int)c, y);
            const long left = min((long)d.total, (long)(y + 1) * d.per) - (long)y * d.per - 2L * d.stride - taken;
            if (left > best_left) {
                best_left = left;
                best = y;
            }
        }
        if (best < 0) return -1;
        deal_ticket_issue(d, best, tk, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        it = deal_item(d, best, deal_ticket_value(tk));
        if (it >= 0) return it;
    }
    return -1;
}
__device__ __forceinline__ void deal_publish(const DealX &d, uint32_t seq, int item, int lane)   // wave 0
{
    if (lane == 0) {
        const unsigned long long p = d.ws + (unsigned long long)(DEAL_MAILBOX0 + (int)blockIdx.x) * 8u;
        const unsigned long long v = ((unsigned long long)seq << 32) | (unsigned long long)(uint32_t)(item + 1);
        asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    }
}
// all waves: request the mailbox (every lane the same 8 bytes; nobody waits: the passes' counted waits retire it) ...
__device__ __forceinline__ void deal_request(const DealX &d, unsigned long long &m)
{
    const unsigned long long p = d.ws + (unsigned long long)(DEAL_MAILBOX0 + (int)blockIdx.x) * 8u;
    m = 0;
    asm volatile("" : "+v"(m));
    asm volatile("global_load_dwordx2 %0, %1, off sc1" : "+v"(m) : "v"(p) : "memory");
}
// ... and take it: the item published under sequence number `seq`, -1 = none.  Re-reads (blocking, bounded: a bug must not hang the
// chip) only if the load has not landed or wave 0 has not published yet.
__device__ __forceinline__ int deal_take(const DealX &d, unsigned long long &m, uint32_t seq)
{
    asm volatile("" : "+v"(m));
    uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)m), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(m >> 32));
    if (hi != seq) {
#pragma unroll 1
        for (int spin = 0; spin < (1 << 16) && hi != seq; spin++) {
            deal_request(d, m);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("" : "+v"(m));
            lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)m);
            hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(m >> 32));
            if (hi != seq) __builtin_amdgcn_s_sleep(8);
        }
        if (hi != seq) return -1;
    }
    return (int)lo - 1;
}

// =====================================================================================================================
// conv2d<>
// =====================================================================================================================
// The DMA schedule: one 25-pass window = two channel groups (50 K steps); a tile is two windows.  Plane pl of the group the
// window starts with is last READ (fragments are fetched one pass ahead of their MFMAs) in pass 3 / 6 / 9 / 11, of the window's
// second group in pass 15 / 18 / 21 / 23; it is re-filled from the pass after that.  What a pass reads was requested at least
// FLIGHT + 1 passes earlier.  KIND: which group a piece belongs to —
//   CUR0: the window's first group (the tail of plane 2 and plane 3, requested in the window's first passes)
//   CUR1: the window's second group
//   NEXT0: the first group of the NEXT window (in a tile's second window: of the workgroup's next tile)
enum { K_CUR0 = 0, K_CUR1 = 1, K_NEXT0 = 2 };
struct Refresh {
    int plane, slot, kind;
};
__host__ __device__ constexpr Refresh refresh_x(int P, int idx)   // the idx-th (0 / 1) piece of window pass P, plane < 0: none
{
    constexpr Refresh none{-1, 0, 0};
    switch (P) {
    case 0: return idx == 0 ? Refresh{2, 3, K_CUR0} : Refresh{3, 0, K_CUR0};
    case 1: return idx == 0 ? Refresh{2, 4, K_CUR0} : Refresh{3, 1, K_CUR0};
    case 2: return idx == 0 ? Refresh{3, 2, K_CUR0} : none;
    case 3: return idx == 0 ? Refresh{3, 3, K_CUR0} : none;
    case 4: return idx == 0 ? Refresh{3, 4, K_CUR0} : Refresh{0, 0, K_CUR1};
    case 5: return idx == 0 ? Refresh{0, 1, K_CUR1} : none;
    case 6: return idx == 0 ? Refresh{0, 2, K_CUR1} : Refresh{0, 3, K_CUR1};
    case 7: return idx == 0 ? Refresh{0, 4, K_CUR1} : Refresh{1, 0, K_CUR1};
    case 8: return idx == 0 ? Refresh{1, 1, K_CUR1} : none;
    case 9: return idx == 0 ? Refresh{1, 2, K_CUR1} : none;
    case 10: return idx == 0 ? Refresh{1, 3, K_CUR1} : Refresh{2, 0, K_CUR1};
    case 11: return idx == 0 ? Refresh{1, 4, K_CUR1} : Refresh{2, 1, K_CUR1};
    case 12: return idx == 0 ? Refresh{2, 2, K_CUR1} : Refresh{3, 0, K_CUR1};
    case 13: return idx == 0 ? Refresh{2, 3, K_CUR1} : Refresh{3, 1, K_CUR1};
    case 14: return idx == 0 ? Refresh{2, 4, K_CUR1} : Refresh{3, 2, K_CUR1};
    case 15: return idx == 0 ? Refresh{3, 3, K_CUR1} : Refresh{3, 4, K_CUR1};
    case 16: return idx == 0 ? Refresh{0, 0, K_NEXT0} : none;
    case 17: return idx == 0 ? Refresh{0, 1, K_NEXT0} : none;
    case 18: return idx == 0 ? Refresh{0, 2, K_NEXT0} : none;
    case 19: return idx == 0 ? Refresh{0, 3, K_NEXT0} : Refresh{1, 0, K_NEXT0};
    case 20: return idx == 0 ? Refresh{0, 4, K_NEXT0} : Refresh{1, 1, K_NEXT0};
    case 21: return idx == 0 ? Refresh{1, 2, K_NEXT0} : none;
    case 22: return idx == 0 ? Refresh{1, 3, K_NEXT0} : Refresh{2, 0, K_NEXT0};
    case 23: return idx == 0 ? Refresh{1, 4, K_NEXT0} : Refresh{2, 1, K_NEXT0};
    case 24: return idx == 0 ? Refresh{2, 2, K_NEXT0} : none;
    }
    return none;
}
__host__ __device__ constexpr int requests_x(int P)   // LDS-DMA instructions a wave issues in window pass P (any integer P)
{
    const int p = ((P % 25) + 25) % 25;
    return 2 + (refresh_x(p, 0).plane >= 0) + (refresh_x(p, 1).plane >= 0);
}
__host__ __device__ constexpr int in_flight_x(int P)   // requests of the last FLIGHT passes
{
    int s = 0;
    for (int i = 0; i < FLIGHT; i++) s += requests_x(P - i);
    return s;
}
// the schedule against the walk, checked at compile time: a piece of plane pl, kind k, requested in pass P must be requested
// after the plane's last read and FLIGHT + 1 passes before its first read
__host__ __device__ constexpr bool schedule_ok()
{
    int count[3][4] = {};
    for (int P = 0; P < 25; P++)
        for (int i = 0; i < 2; i++) {
            const Refresh r = refresh_x(P, i);
            if (r.plane < 0) continue;
            count[r.kind][r.plane] |= 1 << r.slot;
            // steps (window-local) at which the OLD content of the plane is last used and the NEW content first used
            const int first_tap[4] = {0, 9, 15, 21}, last_tap[4] = {8, 14, 20, 24};
            int last_use_step = 0, first_use_step = 0;   // relative to this window's step 0
            if (r.kind == K_CUR0) { last_use_step = last_tap[r.plane] - 25; first_use_step = first_tap[r.plane]; }
            else if (r.kind == K_CUR1) { last_use_step = last_tap[r.plane]; first_use_step = 25 + first_tap[r.plane]; }
            else { last_use_step = 25 + last_tap[r.plane]; first_use_step = 50 + first_tap[r.plane]; }
            // MFMAs of step s run in pass floor(s / 2) and their fragments are read one pass earlier
            auto fl = [](int s) { return s >= 0 ? s / 2 : -((-s + 1) / 2); };
            const int last_read = fl(last_use_step) - 1, first_read = fl(first_use_step) - 1;
            if (P <= last_read) return false;               // would overwrite fragments still to be read
            if (P > first_read - 1 - FLIGHT) return false;   // would not be covered by the counted wait before its first read
        }
    // every plane's 5 slots exactly once per group: CUR0 carries plane 2 slots 3, 4 and plane 3; NEXT0 the rest
    for (int pl = 0; pl < 4; pl++) {
        if (count[K_CUR1][pl] != 31) return false;
        if ((count[K_CUR0][pl] | count[K_NEXT0][pl]) != 31 || (count[K_CUR0][pl] & count[K_NEXT0][pl]) != 0) return false;
    }
    return true;
}
static_assert(schedule_ok(), "conv refresh schedule violates the read / landing windows");

__host__ __device__ constexpr uint32_t tap_off_x(int t)   // byte offset of tap t (plane-ordered walk) inside the planes
{
    const Tap a = conv_tap(t);
    return (uint32_t)(((a.ky & 1) * 2 + (a.kx & 1)) * PLANE + ((a.ky >> 1) * PX + (a.kx >> 1)) * 32);
}

struct ConvXCtx {
    uint32_t lane_pix;      // LDS address of this lane's pixel fragment 0 at tap offset 0: ((4w) * PX + pos) * 32 + half * 16
    uint32_t lane_wt;       // LDS address of this lane's weight fragment 0 in ring slot 0 (+ one slot for the upper K half)
    lds_u8 *planes_w, *ring_w;       // planes / ring + w * 1024: where this wave's piece of a slot goes (LDS-DMA destination)
    uint32_t lane_w16;               // lane * 16 + w * 1024: this lane's bytes of this wave's piece of a weight tile
    const int8_t *wstream;
    const uint8_t *dma_img;          // the image the plane refresh reads: moves to the next tile's ahead of the tile
    int img_bytes;
    uint32_t grp;                    // byte stride between channel groups of the input
    bool hi;
};

// source offsets (channel group 0) of this wave's 5 pieces of each plane for the tile at (Y0, X0): piece k = slot * 4 + w
__device__ __forceinline__ void set_poff_x(uint32_t (&poff)[4][SLOTS], const TensorMap &im, const TileX &t, int w, int lane, int IW, int IH,
                                           bool valid)
{
#pragma unroll
    for (int slot = 0; slot < SLOTS; slot++) {
        const int p = (slot * 4 + w) * 32 + (lane >> 1);
        const int ty = p / PX, tx = p - ty * PX;
#pragma unroll
        for (int pl = 0; pl < 4; pl++) {
            const int iy = 2 * (t.Y0 - 1 + ty) + (pl >> 1), ix = 2 * (t.X0 - 1 + tx) + (pl & 1);
            const bool ok = valid && p < PIX && iy >= 0 && iy < IH && ix >= 0 && ix < IW;
            poff[pl][slot] = ok ? tensor_offset(im, iy, ix, 0u) + (uint32_t)(lane & 1) * 16u : OOB;
        }
    }
}

// pass T (0 .. 49) of a conv tile: the window pass is T % 25, the window T / 25
template <int T, int KIND, bool NT>
__device__ __forceinline__ void conv_pass_x(v4i (&acc)[8][8], const v4i (&pc)[8], const v4i (&wc)[8], v4i (&pn)[8], v4i (&wn)[8],
                                            const ConvXCtx &c, const uint32_t (&poff)[4][SLOTS], bool stores, const HandX &h,
                                            const BiasX &bias)
{
    constexpr int P = T % 25, WIN = T / 25;
    // ---- fragments of the next pass (T + 1, wrapping into the next tile: same offsets, the ring does not care) --------------
    constexpr int TN = (T + 1) % NPASS;
    constexpr uint32_t offNA = tap_off_x((2 * TN) % 25), offNB = tap_off_x((2 * TN + 1) % 25);
    constexpr int slotN = (2 * TN) % RING;                    // the upper K half reads slot slotN + 1 (lane_wt carries that)
    const uint32_t pixn = c.lane_pix + (c.hi ? offNB : offNA);
    const uint32_t wtn = c.lane_wt;
    auto rd = [&](auto r_tag) {
        constexpr int R = decltype(r_tag)::value;
        if constexpr (R < 8)
            pn[R] = lds_read<((R >> 1) * PX + (R & 1) * 16) * 32>(pixn);
        else
            wn[R - 8] = lds_read<slotN * TB + (R - 8) * 16 * 32>(wtn);
    };
    // ---- this pass's requests: weight tiles 5 passes ahead, the plane pieces of the schedule -------------------------------
    auto dma = [&](auto idx_tag) {
        constexpr int idx = decltype(idx_tag)::value;   // 0, 1: weight tiles; 2, 3: plane pieces
        if constexpr (idx < 2) {
            constexpr int step = (2 * T + RING + idx) % (2 * NPASS), slot = (2 * T + idx) % RING;   // slot of step s is s % RING
            __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)c.wstream, 0, 2 * NPASS * TB, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, c.ring_w + slot * TB, 16, c.lane_w16, (uint32_t)(step * TB), 0, 0);
        } else {
            constexpr Refresh r = refresh_x(P, idx - 2);
#ifdef SICN_XW_NOPLANE   // timing experiment (wrong results): the passes without their plane requests
            constexpr bool plane_dma = false;
#else
            constexpr bool plane_dma = true;
#endif
            if constexpr (r.plane >= 0 && plane_dma) {
                // group of the piece: window WIN holds groups 2 WIN, 2 WIN + 1; NEXT0 of the second window = group 0 (of the next tile)
                constexpr int q = r.kind == K_CUR0 ? 2 * WIN : r.kind == K_CUR1 ? 2 * WIN + 1 : (2 * WIN + 2) % 4;
                __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)c.dma_img, 0, c.img_bytes, 0x00020000);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, c.planes_w + r.plane * PLANE + r.slot * 4096, 16,
                                                         poff[r.plane][r.slot] + (uint32_t)q * c.grp, 0, 0, 0);
            }
        }
    };
#ifdef SICN_XW_NOPLANE
    constexpr int VM = 2 * FLIGHT;
#else
    constexpr int VM = in_flight_x(P);
#endif
    pass_x<KIND, VM, (T < FLIGHT ? NSTORE : 0), NT>(acc, pc, wc, rd, dma, stores, h, bias);
}

template <int T, int END, bool NT>
__device__ __forceinline__ void conv_passes_x(v4i (&acc)[8][8], v4i (&pa)[8], v4i (&wa)[8], v4i (&pb)[8], v4i (&wb)[8], const ConvXCtx &c,
                                              const uint32_t (&poff)[4][SLOTS], bool stores, const HandX &h, const BiasX &bias)
{
    if constexpr ((T & 1) == 0)
        conv_pass_x<T, 0, NT>(acc, pa, wa, pb, wb, c, poff, stores, h, bias);
    else
        conv_pass_x<T, 0, NT>(acc, pb, wb, pa, wa, c, poff, stores, h, bias);
    if constexpr (T + 1 < END) conv_passes_x<T + 1, END, NT>(acc, pa, wa, pb, wb, c, poff, stores, h, bias);
}

constexpr size_t CONV_LDS = 4 * PLANE + RING * TB;
constexpr int SWITCH_T = 25 + 16;                 // from this pass of a tile on, every plane request belongs to the next tile

template <bool NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_conv_x(
    const uint8_t *__restrict__ in, uint8_t *__restrict__ out, const int8_t *__restrict__ wstream, const int8_t *__restrict__ bias_g, int IW,
    int IH, int OW, int OH, int tiles_x, int n_tiles, int n_images, int in_layout, int out_layout, uint32_t act_floor, int n_xcd,
    unsigned long long *deal)
{
    constexpr int CIN = 128, COUT = 128;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pos = lane & 15, g = lane >> 4, half = g & 1;
    const bool hi = (g >> 1) != 0;
    const int in_img_bytes = IH * IW * CIN, out_img_bytes = OH * OW * COUT;
    const TensorMap im = tensor_map(in_layout, CIN, IW, IH), om = tensor_map(out_layout, COUT, OW, OH);
    // this workgroup's tiles: XCD x = l % n_xcd owns [x * per, (x + 1) * per) of the (tile x, tile y, image) list; its gridDim.x / 8
    // workgroups take every (gridDim.x / 8)-th one — neighbours in the list run on one XCD at about the same time
    const int total = n_tiles * n_images, per = (total + n_xcd - 1) / n_xcd;
    const int xcd = (int)blockIdx.x % n_xcd, stride = (int)gridDim.x / n_xcd;
    int item = xcd * per + (int)blockIdx.x / n_xcd;
    const int item_end = min(total, (xcd + 1) * per);
    if (item >= item_end) return;   // before any barrier or request
    const DealX dl{(unsigned long long)(uintptr_t)deal, xcd, n_xcd, per, stride, total};
    stagger_x(n_xcd);
    auto coord = [&](int it) {
        const int img = it / n_tiles, tile = it - img * n_tiles, ty = tile / tiles_x;
        return TileX{img, ty * TY, (tile - ty * tiles_x) * TX};
    };
    TileX tc = coord(item);
    uint32_t poff[4][SLOTS];
    set_poff_x(poff, im, tc, w, lane, IW, IH, true);
    const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(smem);
    ConvXCtx ctx{lds0 + (uint32_t)(((4 * w) * PX + pos) * 32 + half * 16),
                 lds0 + 4 * PLANE + (uint32_t)(pos * 32 + half * 16) + (hi ? (uint32_t)TB : 0u),
                 (lds_u8 *)LDS_PTR(smem) + w * 1024,
                 (lds_u8 *)LDS_PTR(smem) + 4 * PLANE + w * 1024,
                 (uint32_t)(lane * 16 + w * 1024),
                 wstream,
                 in + (size_t)tc.img * in_img_bytes,
                 in_img_bytes,
                 im.grp,
                 hi};
    BiasX bias;
    load_bias_x(bias, bias_g, g);
    // ---- prologue of the FIRST tile only: what the previous window's NEXT0 requests would have brought (planes 0, 1 and
    // ---- slots 0 .. 2 of plane 2 of group 0) + the weight tiles of passes 0 .. 4 -------------------------------------------
    {
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)ctx.dma_img, 0, in_img_bytes, 0x00020000);
#pragma unroll
        for (int pl = 0; pl < 3; pl++)
#pragma unroll
            for (int slot = 0; slot < (pl < 2 ? SLOTS : 3); slot++)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, ctx.planes_w + pl * PLANE + slot * 4096, 16, poff[pl][slot], 0, 0, 0);
        __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)wstream, 0, 2 * NPASS * TB, 0x00020000);
#pragma unroll
        for (int s = 0; s < RING; s++) __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, ctx.ring_w + s * TB, 16, ctx.lane_w16, (uint32_t)(s * TB), 0, 0);
    }
    wait_vmcnt<0>();
    block_barrier();
    v4i acc[8][8];
    v4i pa[8], wa[8], pb[8], wb[8];
    {   // fragments of pass 0: taps 0 / 1 of group 0, ring slots 0 / 1
        const uint32_t p0 = ctx.lane_pix + (hi ? tap_off_x(1) : tap_off_x(0));
#define SICN_R(R) pa[R] = lds_read<((R >> 1) * PX + (R & 1) * 16) * 32>(p0); wa[R] = lds_read<R * 16 * 32>(ctx.lane_wt);
        SICN_R(0) SICN_R(1) SICN_R(2) SICN_R(3) SICN_R(4) SICN_R(5) SICN_R(6) SICN_R(7)
#undef SICN_R
        // pass 0's requests overwrite the ring slots just read (prefetch distance = ring size): all four waves must have read them
        // first — the passes' own barriers give that from pass 1 on (round 4: found in k_mfma16p.hip's K-split kernels)
        block_barrier();
    }
    HandX h;
    h.soff = 0u;
    h.grp2 = 2u * om.grp;
    h.floor2 = (act_floor & ACT_FLOOR_MASK) == ACT_FLOOR_RELU ? 0u : (uint32_t)-128;
    h.ro = __builtin_amdgcn_make_buffer_rsrc((void *)out, 0, 0, 0x00020000);
#pragma unroll
    for (int c = 0; c < 8; c++) h.off[c] = OOB;
    // the first pass of the first tile: accumulators start at the bias, nothing to hand over
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            acc[i][j] = bias.c[j];
            if (j < JV)
                asm volatile("" : "+a"(acc[i][j]));
            else
                asm volatile("" : "+v"(acc[i][j]));
        }
    asm volatile("s_nop 3" ::: "memory");   // v_accvgpr_write / v_mov -> asm MFMA reading it as SrcC
    conv_pass_x<0, 0, NT>(acc, pa, wa, pb, wb, ctx, poff, false, h, bias);
#ifdef SICN_STAMP
    const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long st_hand = 0;
    int st_tiles = 1;
#endif
    bool stores = false;   // a hand-over's stores may be in flight (not in the first tile)
    int tile_no = 0;         // tiles this workgroup has finished (DealX: the mailbox's sequence numbers)
    uint32_t deal_tk = DEAL_PENDING;
    unsigned long long deal_mail = 0;
    bool deal_done = false;  // wave 0: "none" has been published
#pragma unroll 1
    for (;;) {
        {   // keep hipcc from hoisting the tile's ~180 LDS-DMA destinations (M0 values) out of the tile loop: it did, and spilled
            // 115 SGPRs into VGPR lanes — a v_readlane + wait states in front of every request
            lds_u8 *pw = ctx.planes_w, *rw_ = ctx.ring_w;
            asm volatile("" : "+s"(pw), "+s"(rw_));
            ctx.planes_w = pw;
            ctx.ring_w = rw_;
        }
        // the next tile: tile 1 is the static deal's; from then on it was published in the mailbox while the previous tile ran (DealX)
        int next = item + stride;
        const bool dyn = dl.ws != 0, later = dyn && tile_no > 0;
        conv_passes_x<1, 3, NT>(acc, pa, wa, pb, wb, ctx, poff, stores, h, bias);
        if (later) deal_request(dl, deal_mail);
        conv_passes_x<3, 7, NT>(acc, pa, wa, pb, wb, ctx, poff, stores, h, bias);
        if (later) next = deal_take(dl, deal_mail, (uint32_t)tile_no + 1u);
        // a ticket is only taken for a tile this workgroup will get to: the one behind `next`
        const bool has_next = dyn ? (next >= 0 && (tile_no > 0 || next < item_end)) : next < item_end;
        const bool goes_on = dyn && has_next && w == 0 && !deal_done;
        if (goes_on) deal_ticket_issue(dl, dl.xcd, deal_tk, lane);
        conv_passes_x<7, 13, NT>(acc, pa, wa, pb, wb, ctx, poff, stores, h, bias);
        if (goes_on) {
            const int it2 = deal_resolve(dl, deal_tk, lane);
            deal_publish(dl, (uint32_t)tile_no + 2u, it2, lane);
            deal_done = it2 < 0;
        }
        conv_passes_x<13, SWITCH_T, NT>(acc, pa, wa, pb, wb, ctx, poff, stores, h, bias);
        // every plane request for THIS tile has been issued: the rest of the tile fetches the next tile's first group
        const TileX tn = coord(has_next ? next : item);
        int lane_l;   // recomputed, not kept
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_l));
        set_poff_x(poff, im, tn, w, lane_l, IW, IH, has_next);
        ctx.dma_img = in + (size_t)tn.img * in_img_bytes;
        // where the finished tile goes
        h.ro = __builtin_amdgcn_make_buffer_rsrc((void *)(out + (size_t)tc.img * out_img_bytes), 0, out_img_bytes, 0x00020000);
        set_out_off_x(h.off, om, tc, w, lane_l, OW, OH, 1);
        conv_passes_x<SWITCH_T, NPASS, NT>(acc, pa, wa, pb, wb, ctx, poff, false, h, bias);
        // pass 0 of the next tile with the hand-over of this one woven in (after the last tile: the same pass on zero-filled
        // planes, whose results nobody reads — one pass in 32 tiles, and the tile loop stays one body)
        asm volatile("s_nop 7" ::: "memory");   // the last MFMAs' results -> v_accvgpr_read
#ifdef SICN_STAMP
        const unsigned long long st_p0 = __builtin_amdgcn_s_memtime();
#endif
        conv_pass_x<0, 2, NT>(acc, pa, wa, pb, wb, ctx, poff, true, h, bias);
        stores = true;
#ifdef SICN_STAMP
        st_hand += __builtin_amdgcn_s_memtime() - st_p0;
#endif
        if (!has_next) break;
        item = next;
        tc = tn;
        tile_no++;
#ifdef SICN_STAMP
        st_tiles++;
#endif
    }
#ifdef SICN_STAMP
    if (g_sicn_stamp_x && tid == 0) {
        unsigned long long *o = g_sicn_stamp_x + (size_t)blockIdx.x * 8 + 2048 * (n_tiles > 500 ? 0 : 1);   // slot 0 / 1: the larger / smaller conv of the 8 x 4K net
        o[0] = __builtin_amdgcn_s_memtime() - st_t0;
        o[1] = __builtin_amdgcn_s_memrealtime() - st_r0;
        o[2] = (unsigned long long)st_tiles;
        o[3] = st_hand;
        o[4] = st_t0;
        o[5] = __builtin_amdgcn_s_getreg(4 | (31 << 11));    // HW_REG_HW_ID
        o[6] = __builtin_amdgcn_s_getreg(20 | (31 << 11));   // HW_REG_XCC_ID
    }
#endif
    wait_vmcnt<0>();   // the wrapped tail of the prefetch (weights, out-of-range plane pieces) must land before the LDS is released
}

// =====================================================================================================================
// deconv522<>
// =====================================================================================================================
// The walk of a tile: 4 output phases (py, px), each over its taps (ky = py mod 2, kx = px mod 2: 9 / 6 / 6 / 4) and the two
// channel-group pairs; one pass = one tap, one pair (lanes 0-31 the pair's first group, 32-63 its second).  Phases 0 and 3 go
// pair by pair (all taps of pair 0, then all taps of pair 1), so that pair 0's buffer is free 4 passes before the tile ends
// and pair 1 is not needed before pass 9: that is when the next tile's pair 1 is brought in.
struct DPass {
    int phase, pair, iy, ix;   // tap (ky, kx) = (2 iy + py, 2 ix + px)
};
__host__ __device__ constexpr DPass dpass(int T)
{
    if (T < 18) return DPass{0, T / 9, (T % 9) / 3, (T % 9) % 3};
    if (T < 30) return DPass{1, (T - 18) % 2, ((T - 18) / 2) / 2, ((T - 18) / 2) % 2};
    if (T < 42) return DPass{2, (T - 30) % 2, ((T - 30) / 2) / 3, ((T - 30) / 2) % 3};
    return DPass{3, (T - 42) / 4, ((T - 42) % 4) / 2, ((T - 42) % 4) % 2};
}
__host__ __device__ constexpr bool dphase_start(int T) { return T == 0 || T == 18 || T == 30 || T == 42; }
__host__ __device__ constexpr uint32_t dtap_off(int T)   // byte offset of pass T's tap inside a group's 18 x 34 patch
{
    const DPass p = dpass(T);
    return (uint32_t)(((p.iy + (p.phase >> 1)) * PX + p.ix + (p.phase & 1)) * 32);
}
// patch requests (one piece per wave each): the current tile's pair-1 pieces 5 .. 9 in passes 0 .. 4 (its pieces 0 .. 4 came in
// the previous tile's passes 45 .. 49), the next tile's pair 0 (10 pieces) every 4th pass from 6 on, the next tile's pair-1
// pieces 0 .. 4 in passes 45 .. 49.  Piece k of a pair: group k / 5 of the pair, slot k % 5.
struct DReq {
    int kind, k;   // kind < 0: none; 0: current tile's pair 1; 1: next tile's pair 0; 2: next tile's pair 1
};
__host__ __device__ constexpr DReq dreq(int T)
{
    if (T < 5) return DReq{0, 5 + T};
    if (T >= 6 && T <= 42 && (T - 6) % 4 == 0) return DReq{1, (T - 6) / 4};
    if (T >= 45) return DReq{2, T - 45};
    return DReq{-1, 0};
}
__host__ __device__ constexpr int drequests(int T) { return 2 + (dreq(((T % NPASS) + NPASS) % NPASS).kind >= 0); }
__host__ __device__ constexpr int d_in_flight(int T)
{
    int s = 0;
    for (int i = 0; i < FLIGHT; i++) s += drequests(T - i);
    return s;
}
__host__ __device__ constexpr bool dschedule_ok()
{
    // pair 0 of a tile is read (one pass ahead) from pass -1 (for pass 0) to pass 44 (for pass 45); pair 1 from pass 8 (for pass 9)
    // to pass 48.  Buffers: the next tile's pair 0 goes to the spare buffer (= the previous tile's pair-1 buffer: free from pass 0),
    // its pair 1 to this tile's pair-0 buffer (free from pass 45).
    int seen0 = 0, seen1 = 0;
    for (int T = 0; T < NPASS; T++) {
        const DReq r = dreq(T);
        if (r.kind == 1) {
            if (T > NPASS - 1 - 1 - FLIGHT) return false;   // read from pass 49 on
            seen0 |= 1 << r.k;
        }
        if (r.kind == 2) {
            if (T < 45) return false;
            seen1 |= 1 << r.k;
        }
        if (r.kind == 0) {
            if (T > 8 - 1 - FLIGHT) return false;           // read from pass 8 on
            seen1 |= 1 << r.k;
        }
    }
    for (int T = 0; T < NPASS; T++) {   // the walk itself: pair 0 never after pass 45, pair 1 never before pass 9
        if (dpass(T).pair == 0 && T > 45) return false;
        if (dpass(T).pair == 1 && T < 9) return false;
    }
    return seen0 == 1023 && seen1 == 1023;
}
static_assert(dschedule_ok(), "deconv patch schedule violates the read / landing windows");

constexpr int PAIRBUF = 2 * PLANE;   // one group pair: 40 KiB
constexpr size_t DECONV_LDS = 3 * PAIRBUF + RING * TB;   // 160 KiB

struct DeconvXCtx {
    uint32_t lane_pix;      // LDS address of this lane's pixel fragment 0 at tap offset 0 in buffer 0 (+ one group for the upper K half)
    uint32_t lane_wt;
    lds_u8 *patch_w, *ring_w;
    uint32_t lane_w16;
    const int8_t *wstream;
    const uint8_t *cur_img, *next_img;
    int img_bytes;
    uint32_t grp;
    uint32_t buf0, buf1, buf2;   // byte offsets of the buffers holding this tile's pair 0, pair 1, and the spare one
};

__device__ __forceinline__ void set_dpoff_x(uint32_t (&poff)[SLOTS], const TensorMap &im, const TileX &t, int w, int lane, int IW, int IH, bool valid)
{
#pragma unroll
    for (int slot = 0; slot < SLOTS; slot++) {
        const int p = (slot * 4 + w) * 32 + (lane >> 1);
        const int ty = p / PX, tx = p - ty * PX;
        const int iy = t.Y0 - 1 + ty, ix = t.X0 - 1 + tx;
        const bool ok = valid && p < PIX && iy >= 0 && iy < IH && ix >= 0 && ix < IW;
        poff[slot] = ok ? tensor_offset(im, iy, ix, 0u) + (uint32_t)(lane & 1) * 16u : OOB;
    }
}

template <int T, int KIND, bool NT>
__device__ __forceinline__ void deconv_pass_x(v4i (&acc)[8][8], const v4i (&pc)[8], const v4i (&wc)[8], v4i (&pn)[8], v4i (&wn)[8],
                                              const DeconvXCtx &c, const uint32_t (&poff_cur)[SLOTS], const uint32_t (&poff_next)[SLOTS],
                                              bool stores, const HandX &h, const BiasX &bias)
{
    constexpr int TN = (T + 1) % NPASS;
    constexpr DPass N = dpass(TN);
    constexpr uint32_t toffN = dtap_off(TN);
    constexpr int slotN = (2 * TN) % RING;
    // pass 49 fetches the first fragments of the NEXT tile, whose pair 0 sits in this tile's spare buffer
    const uint32_t pixn = c.lane_pix + (T == NPASS - 1 ? c.buf2 : N.pair ? c.buf1 : c.buf0);
    const uint32_t wtn = c.lane_wt;
    auto rd = [&](auto r_tag) {
        constexpr int R = decltype(r_tag)::value;
        if constexpr (R < 8)
            pn[R] = lds_read<toffN + ((R >> 1) * PX + (R & 1) * 16) * 32>(pixn);
        else
            wn[R - 8] = lds_read<slotN * TB + (R - 8) * 16 * 32>(wtn);
    };
    auto dma = [&](auto idx_tag) {
        constexpr int idx = decltype(idx_tag)::value;   // 0, 1: weight tiles; 2: the patch piece; 3: nothing
        if constexpr (idx < 2) {
            constexpr int step = (2 * T + RING + idx) % (2 * NPASS), slot = (2 * T + idx) % RING;
            __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)c.wstream, 0, 2 * NPASS * TB, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, c.ring_w + slot * TB, 16, c.lane_w16, (uint32_t)(step * TB), 0, 0);
        } else if constexpr (idx == 2) {
            constexpr DReq r = dreq(T);
            if constexpr (r.kind >= 0) {
                constexpr int gq = r.k / SLOTS, slot = r.k % SLOTS;   // group inside the pair, slot
                // kind 0: this tile's pair 1 -> its buffer; 1: next tile's pair 0 -> the spare buffer; 2: next tile's pair 1 -> this tile's pair-0 buffer
                const uint32_t dst = (r.kind == 0 ? c.buf1 : r.kind == 1 ? c.buf2 : c.buf0) + (uint32_t)(gq * PLANE + slot * 4096);
                constexpr int q = (r.kind == 1 ? 0 : 2) + gq;
                __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(r.kind == 0 ? c.cur_img : c.next_img), 0, c.img_bytes, 0x00020000);
                const uint32_t off = (r.kind == 0 ? poff_cur[slot] : poff_next[slot]) + (uint32_t)q * c.grp;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, c.patch_w + dst, 16, off, 0, 0, 0);
            }
        }
    };
    // stores of a hand-over are counted, not waited for, during the FLIGHT passes that start with it
    constexpr bool behind = dphase_start(T) || dphase_start((T + NPASS - 1) % NPASS) || dphase_start((T + NPASS - 2) % NPASS);
    static_assert(FLIGHT == 3, "`behind` spells out FLIGHT passes");
    pass_x<KIND, d_in_flight(T), (behind ? NSTORE : 0), NT>(acc, pc, wc, rd, dma, stores, h, bias);
}

template <bool NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_deconv_x(
    const uint8_t *__restrict__ in, uint8_t *__restrict__ out, const int8_t *__restrict__ wstream, const int8_t *__restrict__ bias_g, int IW,
    int IH, int OW, int OH, int tiles_x, int n_tiles, int n_images, int in_layout, int out_layout, uint32_t act_floor, int n_xcd,
    unsigned long long *deal)
{
    constexpr int CIN = 128, COUT = 128;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pos = lane & 15, g = lane >> 4, half = g & 1;
    const bool hi = (g >> 1) != 0;
    const int in_img_bytes = IH * IW * CIN, out_img_bytes = OH * OW * COUT;
    const TensorMap im = tensor_map(in_layout, CIN, IW, IH), om = tensor_map(out_layout, COUT, OW, OH);
    const int total = n_tiles * n_images, per = (total + n_xcd - 1) / n_xcd;
    const int xcd = (int)blockIdx.x % n_xcd, stride = (int)gridDim.x / n_xcd;
    int item = xcd * per + (int)blockIdx.x / n_xcd;
    const int item_end = min(total, (xcd + 1) * per);
    if (item >= item_end) return;
    const DealX dl{(unsigned long long)(uintptr_t)deal, xcd, n_xcd, per, stride, total};
    stagger_x(n_xcd);
    auto coord = [&](int it) {
        const int img = it / n_tiles, tile = it - img * n_tiles, ty = tile / tiles_x;
        return TileX{img, ty * TY, (tile - ty * tiles_x) * TX};
    };
    TileX tc = coord(item);
    uint32_t poff_cur[SLOTS], poff_next[SLOTS];
    set_dpoff_x(poff_cur, im, tc, w, lane, IW, IH, true);
#pragma unroll
    for (int s = 0; s < SLOTS; s++) poff_next[s] = OOB;
    const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(smem);
    DeconvXCtx ctx{lds0 + (uint32_t)(((4 * w) * PX + pos) * 32 + half * 16) + (hi ? (uint32_t)PLANE : 0u),
                   lds0 + 3 * PAIRBUF + (uint32_t)(pos * 32 + half * 16) + (hi ? (uint32_t)TB : 0u),
                   (lds_u8 *)LDS_PTR(smem) + w * 1024,
                   (lds_u8 *)LDS_PTR(smem) + 3 * PAIRBUF + w * 1024,
                   (uint32_t)(lane * 16 + w * 1024),
                   wstream,
                   in + (size_t)tc.img * in_img_bytes,
                   in + (size_t)tc.img * in_img_bytes,
                   in_img_bytes,
                   im.grp,
                   0u,
                   (uint32_t)PAIRBUF,
                   (uint32_t)(2 * PAIRBUF)};
    BiasX bias;
    load_bias_x(bias, bias_g, g);
    // ---- prologue of the FIRST tile only: its pair 0, pieces 0 .. 4 of its pair 1, the weight tiles of passes 0 .. 4 --------
    {
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)ctx.cur_img, 0, in_img_bytes, 0x00020000);
#pragma unroll
        for (int k = 0; k < 3 * SLOTS; k++) {
            const int q = k / SLOTS, slot = k % SLOTS;   // groups 0, 1 (pair 0, buffer 0) and group 2 (pair 1, buffer 1)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, ctx.patch_w + (q < 2 ? q * PLANE : PAIRBUF) + slot * 4096, 16,
                                                     poff_cur[slot] + (uint32_t)q * im.grp, 0, 0, 0);
        }
        __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)wstream, 0, 2 * NPASS * TB, 0x00020000);
#pragma unroll
        for (int s = 0; s < RING; s++) __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, ctx.ring_w + s * TB, 16, ctx.lane_w16, (uint32_t)(s * TB), 0, 0);
    }
    wait_vmcnt<0>();
    block_barrier();
    v4i acc[8][8];
    v4i pa[8], wa[8], pb[8], wb[8];
    {   // fragments of pass 0: phase 0, tap 0, pair 0 (buffer 0), ring slots 0 / 1
        const uint32_t p0 = ctx.lane_pix;
#define SICN_R(R) pa[R] = lds_read<((R >> 1) * PX + (R & 1) * 16) * 32>(p0); wa[R] = lds_read<R * 16 * 32>(ctx.lane_wt);
        SICN_R(0) SICN_R(1) SICN_R(2) SICN_R(3) SICN_R(4) SICN_R(5) SICN_R(6) SICN_R(7)
#undef SICN_R
        // pass 0's requests overwrite the ring slots just read (prefetch distance = ring size): all four waves must have read them
        // first — the passes' own barriers give that from pass 1 on (round 4: found in k_mfma16p.hip's K-split kernels)
        block_barrier();
    }
    HandX h;
    h.soff = 0u;
    h.grp2 = 2u * om.grp;
    h.floor2 = (act_floor & ACT_FLOOR_MASK) == ACT_FLOOR_RELU ? 0u : (uint32_t)-128;
    h.ro = __builtin_amdgcn_make_buffer_rsrc((void *)out, 0, 0, 0x00020000);
#pragma unroll
    for (int c = 0; c < 8; c++) h.off[c] = OOB;
    // scalar offset of output phase (py, px): output pixel (2 y + py, 2 x + px) against (2 y, 2 x)
    auto phase_soff = [&](int ph) { return tensor_offset(om, ph >> 1, ph & 1, 0u); };
    int next = item + stride;
    bool has_next = next < item_end;
    TileX tn = coord(has_next ? next : item);
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            acc[i][j] = bias.c[j];
            if (j < JV)
                asm volatile("" : "+a"(acc[i][j]));
            else
                asm volatile("" : "+v"(acc[i][j]));
        }
    asm volatile("s_nop 3" ::: "memory");   // v_accvgpr_write / v_mov -> asm MFMA reading it as SrcC
    deconv_pass_x<0, 0, NT>(acc, pa, wa, pb, wb, ctx, poff_cur, poff_next, false, h, bias);
#ifdef SICN_STAMP
    const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long st_hand = 0;
    int st_tiles = 1;
#endif
    bool stores = false;   // a hand-over's stores may be in flight (in the first tile: from its phase 1 on)
    int tile_no = 0;         // tiles this workgroup has finished (DealX: the mailbox's sequence numbers)
    uint32_t deal_tk = DEAL_PENDING;
    unsigned long long deal_mail = 0;
    bool deal_done = false;  // wave 0: "none" has been published
#pragma unroll 1
    for (;;) {
        {
            lds_u8 *pw = ctx.patch_w, *rw_ = ctx.ring_w;
            asm volatile("" : "+s"(pw), "+s"(rw_));
            ctx.patch_w = pw;
            ctx.ring_w = rw_;
        }
        // passes 1 .. 5; from pass 6 on the patch requests belong to the next tile
#define SICN_DP(T, ST) \
    if constexpr (((T) & 1) == 0) \
        deconv_pass_x<T, dphase_start(T) ? 2 : 0, NT>(acc, pa, wa, pb, wb, ctx, poff_cur, poff_next, ST, h, bias); \
    else \
        deconv_pass_x<T, dphase_start(T) ? 2 : 0, NT>(acc, pb, wb, pa, wa, ctx, poff_cur, poff_next, ST, h, bias);
        // `stores`: passes 1 .. 7 store the previous tile's last phase (none in the first tile), and a wait counts the stores of
        // the passes up to FLIGHT - 1 back
        // the next tile (its patch is requested from pass 6 on): tile 1 is the static deal's, from then on it comes from the mailbox,
        // where wave 0 published it while the previous tile ran (DealX)
        const bool dyn = dl.ws != 0, later = dyn && tile_no > 0;
        SICN_DP(1, stores)
        if (later) deal_request(dl, deal_mail);
        SICN_DP(2, stores) SICN_DP(3, stores) SICN_DP(4, stores) SICN_DP(5, stores)
        if (later) {
            next = deal_take(dl, deal_mail, (uint32_t)tile_no + 1u);
            has_next = next >= 0;
            tn = coord(has_next ? next : item);
        }
        const bool goes_on = dyn && has_next && w == 0 && !deal_done;   // a ticket is only taken for a tile this workgroup will get to
        {
            int lane_l;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_l));
            set_dpoff_x(poff_next, im, tn, w, lane_l, IW, IH, has_next);
            ctx.next_img = in + (size_t)tn.img * in_img_bytes;
        }
        SICN_DP(6, stores) SICN_DP(7, stores)
        if (goes_on) deal_ticket_issue(dl, dl.xcd, deal_tk, lane);
        SICN_DP(8, stores) SICN_DP(9, stores)
        {   // the previous tile's last stores are out: from here on the hand-overs are this tile's phases
            int lane_l;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_l));
            h.ro = __builtin_amdgcn_make_buffer_rsrc((void *)(out + (size_t)tc.img * out_img_bytes), 0, out_img_bytes, 0x00020000);
            set_out_off_x(h.off, om, tc, w, lane_l, IW, IH, 2);
        }
        SICN_DP(10, true) SICN_DP(11, true) SICN_DP(12, true) SICN_DP(13, true)
        SICN_DP(14, true)
        if (goes_on) {
            const int it2 = deal_resolve(dl, deal_tk, lane);
            deal_publish(dl, (uint32_t)tile_no + 2u, it2, lane);
            deal_done = it2 < 0;
        }
        SICN_DP(15, true) SICN_DP(16, true) SICN_DP(17, true)
        h.soff = phase_soff(0);
        asm volatile("s_nop 7" ::: "memory");   // the last MFMAs' results -> v_accvgpr_read
        SICN_DP(18, true) SICN_DP(19, true) SICN_DP(20, true) SICN_DP(21, true) SICN_DP(22, true) SICN_DP(23, true) SICN_DP(24, true) SICN_DP(25, true)
        SICN_DP(26, true) SICN_DP(27, true) SICN_DP(28, true) SICN_DP(29, true)
        h.soff = phase_soff(1);
        asm volatile("s_nop 7" ::: "memory");
        SICN_DP(30, true) SICN_DP(31, true) SICN_DP(32, true) SICN_DP(33, true) SICN_DP(34, true) SICN_DP(35, true) SICN_DP(36, true) SICN_DP(37, true)
        SICN_DP(38, true) SICN_DP(39, true) SICN_DP(40, true) SICN_DP(41, true)
        h.soff = phase_soff(2);
        asm volatile("s_nop 7" ::: "memory");
        SICN_DP(42, true) SICN_DP(43, true) SICN_DP(44, true) SICN_DP(45, true) SICN_DP(46, true) SICN_DP(47, true) SICN_DP(48, true) SICN_DP(49, true)
        h.soff = phase_soff(3);
        asm volatile("s_nop 7" ::: "memory");
        // the next tile: its pair 0 sits in the spare buffer, its pair 1 (pieces 0 .. 4 so far) in this tile's pair-0 buffer
        {
            const uint32_t b0 = ctx.buf0, b1 = ctx.buf1, b2 = ctx.buf2;
            ctx.buf0 = b2;
            ctx.buf1 = b0;
            ctx.buf2 = b1;
        }
        ctx.cur_img = ctx.next_img;
#pragma unroll
        for (int s = 0; s < SLOTS; s++) poff_cur[s] = poff_next[s];
#ifdef SICN_STAMP
        const unsigned long long st_p0 = __builtin_amdgcn_s_memtime();
#endif
        // pass 0 of the next tile with the hand-over of this tile's phase 3 woven in (after the last tile: on a zero-filled patch)
        deconv_pass_x<0, 2, NT>(acc, pa, wa, pb, wb, ctx, poff_cur, poff_next, true, h, bias);
#undef SICN_DP
        stores = true;
#ifdef SICN_STAMP
        st_hand += __builtin_amdgcn_s_memtime() - st_p0;
#endif
        if (!has_next) break;
        item = next;
        tc = tn;
        tile_no++;
        if (!dl.ws) {   // the static deal; with a deal area the next tile is read from the mailbox at the top of the loop
            next = item + stride;
            has_next = next < item_end;
            tn = coord(has_next ? next : item);
        }
#ifdef SICN_STAMP
        st_tiles++;
#endif
    }
#ifdef SICN_STAMP
    if (g_sicn_stamp_x && tid == 0) {
        unsigned long long *o = g_sicn_stamp_x + (size_t)blockIdx.x * 8 + 2048 * (n_tiles > 500 ? 2 : 3);   // slot 2 / 3: the larger / smaller deconv
        o[0] = __builtin_amdgcn_s_memtime() - st_t0;
        o[1] = __builtin_amdgcn_s_memrealtime() - st_r0;
        o[2] = (unsigned long long)st_tiles;
        o[3] = st_hand;
        o[4] = st_t0;
        o[5] = __builtin_amdgcn_s_getreg(4 | (31 << 11));
        o[6] = __builtin_amdgcn_s_getreg(20 | (31 << 11));
    }
#endif
    wait_vmcnt<0>();
}

}  // namespace xw

// ---- host side --------------------------------------------------------------------------------------------------------------
// the deconv's weight stream in the order k_deconv_x walks it (dpass): tile 2 T + h = (tap of pass T, channel group 2 pair + h),
// rows as in pack_mfma16_stream
size_t mfma16x_deconv_stream_bytes(int cin, int cout) { return (cin == 128 && cout == 128) ? (size_t)2 * xw::NPASS * cout * KSTEP : 0; }
void pack_mfma16x_deconv_stream(const int8_t *w_okc, int cin, int cout, int8_t *dst)
{
    const int kk = 25 * cin;
    const size_t tb = (size_t)cout * KSTEP;
    for (int T = 0; T < xw::NPASS; T++) {
        const xw::DPass p = xw::dpass(T);
        const int ky = 2 * p.iy + (p.phase >> 1), kx = 2 * p.ix + (p.phase & 1);
        for (int hh = 0; hh < 2; hh++) {
            int8_t *tile = dst + (size_t)(2 * T + hh) * tb;
            const int q = 2 * p.pair + hh;
            for (int row = 0; row < cout; row++) {
                const int j = row >> 4, rho = row & 15;
                const int ch = 64 * (j >> 2) + 16 * (rho >> 2) + 4 * (j & 3) + (rho & 3);
                const int8_t *src = w_okc + (size_t)ch * kk + (ky * 5 + kx) * cin + q * 32;
                for (int b = 0; b < 32; b++) tile[row * 32 + b] = src[b];
            }
        }
    }
}

// conv / deconv 128 -> 128 on 16 x 32 tiles by one persistent workgroup per CU
bool wide_supported(const LayerGeom &g) { return g.CIN == 128 && g.COUT == 128; }

hipError_t launch_wide(const LayerGeom &g, const sicn_weights &w, const uint8_t *in, uint8_t *out, int n_images, hipStream_t stream,
                       int in_layout, int out_layout, bool relu, int grid_cap, const ChipGeom &chip, unsigned long long *deal)
{
    using namespace xw;
    if (!wide_supported(g)) return hipErrorInvalidValue;
    if (g.transposed && !w.d_w_mfma16x) return hipErrorInvalidValue;
    if ((size_t)g.IH * g.IW * g.CIN >= (size_t)OOB || (size_t)g.OH * g.OW * g.COUT >= (size_t)OOB) return hipErrorInvalidValue;
    const int MW = g.transposed ? g.IW : g.OW, MH = g.transposed ? g.IH : g.OH;
    const int tiles_x = (MW + TX - 1) / TX, tiles_y = (MH + TY - 1) / TY;
    const long total = (long)tiles_x * tiles_y * n_images;
    if (total <= 0 || total > 0x7fffffffL) return hipErrorInvalidValue;
    const uint32_t flags = relu ? ACT_FLOOR_RELU : ACT_FLOOR_RAW;
    const bool nt = nt_store_wanted((size_t)g.OH * g.OW * g.COUT * n_images);
    const unsigned grid = wide_grid(total, grid_cap, chip);   // one resident per CU, a multiple of the XCD count (sicn_plan.h)
    const void *fn = g.transposed ? (nt ? (const void *)&k_deconv_x<true> : (const void *)&k_deconv_x<false>)
                                  : (nt ? (const void *)&k_conv_x<true> : (const void *)&k_conv_x<false>);
    const size_t lds = g.transposed ? DECONV_LDS : CONV_LDS;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const int8_t *ws = g.transposed ? w.d_w_mfma16x : w.d_w_mfma16;
    int IW = g.IW, IH = g.IH, OW = g.OW, OH = g.OH, n_tiles = tiles_x * tiles_y, txs = tiles_x;
    uint32_t fl = flags;
    int n_xcd = chip.n_xcd;
    static_assert(DEAL_MAILBOX0 == WIDE_DEAL_MAX_XCDS, "the mailboxes start behind the XCDs' counters");
    if (!wide_deal_pays(total, grid, chip)) deal = nullptr;   // few tiles per workgroup: all of them dealt statically (sicn_plan.h)
    void *args[] = {(void *)&in, (void *)&out, (void *)&ws, (void *)&w.d_bias, &IW, &IH, &OW, &OH, &txs, &n_tiles, &n_images, &in_layout, &out_layout, &fl, &n_xcd,
                    (void *)&deal};
    e = hipLaunchKernel(fn, dim3(grid), dim3(256), args, lds, stream);
    if (e != hipSuccess) return e;
    return hipGetLastError();
}

}  // namespace sicn
