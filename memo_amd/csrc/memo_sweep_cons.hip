// memo_sweep_cons.hip -- conservation sweep: doubling scatter + top-down fold, the
// k <= 1 fill, the side pass for rows with end < start, tile-shape choice and the ABI entry points.
// Replaces /root/reference/src/memo_query.py:42-63 + the argmax of :70.
#include "memo_sweep.h"
#include "memo_sweep_fold.h"

using namespace memo;

namespace {

// ------------------------------------------------------------------------------------------
// conservation: doubling scatter + top-down fold
// ------------------------------------------------------------------------------------------
template <typename Rows, int W, int U, int T, typename OutT>
__global__ __launch_bounds__(T) void sweep_conservation_kernel(const SweepArgs A) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int tid = threadIdx.x;
    constexpr int LS = W + kLevelSkew;  // words between level arrays
    Tile t;
#ifdef MEMO_STAMPS
    unsigned long long stamp_t0 = __builtin_amdgcn_s_memtime();
#endif
    if (!locate_tile<W>(A, t)) return;
    MEMO_STAMP(0);  // tile location (kernarg + two bucket-table loads)

    // the clipped interval [c, h) is the union of two blocks of 2^j, j = floor(log2(h - c))
    Rows::template for_each<T, U>(
        A, t,
        [&]() {  // every level starts at the sentinel column N (memo_query.py:53-54)
            const uint32_t sent = (uint32_t)(A.ncols - 1);
            const uint4 sv = make_uint4(sent, sent, sent, sent);
            uint4 *p = reinterpret_cast<uint4 *>(lds);
            for (int i = tid; i < A.nlev * (LS / 4); i += T) p[i] = sv;
            lds_barrier();
            MEMO_STAMP(1);  // issue of the first loads + LDS clear + barrier
        },
        [&](int c, int h, int col) {
            if (h > c) {
                const int j = 31 - __builtin_clz((unsigned)(h - c));  // h - c >= 1
                uint32_t *lv = lds + j * LS;
                atomicMin(lv + c, (uint32_t)col);               // block [c, c + 2^j)
                atomicMin(lv + (h - (1 << j)), (uint32_t)col);  // block [h - 2^j, h)
            }
        });
    MEMO_STAMP(2);  // waiting for rows + scatter
    __syncthreads();
    MEMO_STAMP(3);  // barrier after the scatter

    // fold: a block of 2^j at x covers the blocks of 2^(j-1) at x and x + 2^(j-1)
    for (int j = A.nlev - 1; j >= 2; --j) {  // (level 1 -> 0 happens in store_conservation)
        const int half = 1 << (j - 1);
        const uint32_t *hi = lds + j * LS;
        uint32_t *lo = lds + (j - 1) * LS;
        for (int x = 4 * tid; x < W; x += 4 * T) {
            const uint4 v = *reinterpret_cast<const uint4 *>(hi + x);
            uint4 u;
            if (half >= 4) {
                u = x >= half ? *reinterpret_cast<const uint4 *>(hi + x - half)
                              : make_uint4(~0u, ~0u, ~0u, ~0u);
            } else if (half == 2) {
                const uint2 q = x >= 2 ? *reinterpret_cast<const uint2 *>(hi + x - 2)
                                       : make_uint2(~0u, ~0u);
                u = make_uint4(q.x, q.y, v.x, v.y);
            } else {
                const uint32_t q = x >= 1 ? hi[x - 1] : ~0u;
                u = make_uint4(q, v.x, v.y, v.z);
            }
            uint4 w = *reinterpret_cast<const uint4 *>(lo + x);
            w.x = min(w.x, min(v.x, u.x));
            w.y = min(w.y, min(v.y, u.y));
            w.z = min(w.z, min(v.z, u.z));
            w.w = min(w.w, min(v.w, u.w));
            *reinterpret_cast<uint4 *>(lo + x) = w;
        }
        __syncthreads();
    }

    MEMO_STAMP(4);  // fold
    store_conservation<OutT, T>(A, t, lds, A.nlev > 1 ? lds + LS : nullptr, 0);  // as uint16 or (num_docs <= 255) uint8
    MEMO_STAMP(5);  // store
#ifdef MEMO_STAMPS
    if (threadIdx.x == 0 && A.stamps) A.stamps[8ull * blockIdx.x + 7] = 1;
#endif
}


// ------------------------------------------------------------------------------------------
// conservation, packed rows whose annot is known to be inside the matrix: the same doubling scatter
// WITHOUT clipping.
//
// Every level array carries a halo -- HL cells left of the tile, HR right of it -- wide enough for
// any interval a row of the slice can have (start - a in [0, W + k - 1 + bucket - 2], length n = k - 1 -
// overlap in [1, k - 1]); what falls into the right halo is never read back, the left halo is
// folded like the rest (a block that starts left of the tile can cover tile positions).  With
// nothing to clip, the level and both cells follow from n and start - a alone: 10 VALU
// instructions per row instead of 17, and 60 VGPRs instead of 70 (8 waves per SIMD).  Levels are
// stored by leading-zero count (slot = clz(n) - clz(k - 1)): slot 0 = longest blocks.  The tile
// width W = A.w is what leaves the level array (HL + W + HR) at a round size; it is a multiple of
// the bucket width, not a power of two.
// (Measured and dropped, profiles/r01_unclipped_scatter.txt: a workgroup walking 2..32 tiles with
// the next tile's rows in flight under the fold -- 6-17 % slower than one tile per workgroup.)
// ------------------------------------------------------------------------------------------
#ifndef MEMO_HALO_WAVES
#define MEMO_HALO_WAVES 8
#endif
// diagnostic builds only (tools/build_variant.sh): bit 0 / 1 = drop the first / second ds_min of a row,
// 2 = no fold passes, 3 = no clear, 4 = no scatter arithmetic at all (rows are loaded and dropped).
// Results are wrong with any of them; they size what each phase costs.
#ifndef MEMO_ABLATE
#define MEMO_ABLATE 0
#endif

// every level starts at the sentinel column N (memo_query.py:53-54); under the first loads
template <int T>
__device__ __forceinline__ void halo_clear(const SweepArgs &A, uint32_t *lds, uint32_t sent) {
    const uint4 sv = make_uint4(sent, sent, sent, sent);
    uint4 *p = reinterpret_cast<uint4 *>(lds);
    if (!(MEMO_ABLATE & 8))
        for (int i = threadIdx.x; i < A.nlev * (A.ls / 4); i += T) p[i] = sv;
    lds_barrier();
}

// fold: a block of 2^j at x covers the blocks of 2^(j-1) at x and x + 2^(j-1); the left halo too.
// Then the last fold + store (store_conservation).
template <typename OutT, int T, int TOP>
__device__ __forceinline__ void halo_fold_store(const SweepArgs &A, const Tile &t, uint32_t *lds) {
    const int LS = A.ls, HL = A.hl, W = A.w;
    const int cells = HL + W;
    for (int slot = 0; slot + 2 < A.nlev && !(MEMO_ABLATE & 4); ++slot) {  // (the last fold happens in store_conservation)
        const int half = 1 << (A.nlev - 2 - slot);
        const uint32_t *hi = lds + slot * LS;
        uint32_t *lo = lds + (slot + 1) * LS;
        for (int x = 4 * threadIdx.x; x < cells; x += 4 * T) {
            const uint4 v = *reinterpret_cast<const uint4 *>(hi + x);
            uint4 u;
            if (half >= 4) {
                u = x >= half ? *reinterpret_cast<const uint4 *>(hi + x - half)
                              : make_uint4(~0u, ~0u, ~0u, ~0u);
            } else if (half == 2) {
                const uint2 q = x >= 2 ? *reinterpret_cast<const uint2 *>(hi + x - 2)
                                       : make_uint2(~0u, ~0u);
                u = make_uint4(q.x, q.y, v.x, v.y);
            } else {
                const uint32_t q = x >= 1 ? hi[x - 1] : ~0u;
                u = make_uint4(q, v.x, v.y, v.z);
            }
            uint4 r = *reinterpret_cast<const uint4 *>(lo + x);
            r.x = min(r.x, min(v.x, u.x));
            r.y = min(r.y, min(v.y, u.y));
            r.z = min(r.z, min(v.z, u.z));
            r.w = min(r.w, min(v.w, u.w));
            *reinterpret_cast<uint4 *>(lo + x) = r;
        }
        lds_barrier();
    }
    store_conservation<OutT, T, TOP>(A, t, lds + (A.nlev - 1) * LS + HL,
                                      A.nlev > 1 ? lds + (A.nlev - 2) * LS + HL : nullptr, -HL);
}

// The same folds, in registers.  Every level is read ONCE (one conflict-free ds_read_b128 per lane and level) and
// the cascade  M_top = L_top,  M_(j-1)[x] = min(L_(j-1)[x], M_j[x], M_j[x - 2^(j-1)])  runs on a lane's four cells
// with the shifted operand taken from the lanes to its left: through the DPP operand of v_min_u32 itself for one
// lane, through ds_bpermute (the LDS crossbar, no bank access) for more; the result goes straight to the output.
// No intermediate level is written back and no barrier separates the steps.  A wave covers 256 cells of which the
// leftmost 2^(nlev-3) lanes only supply context (their own results would need cells of the previous wave), so
// consecutive waves overlap by that much; wave 0's context lanes hold the first cells of the left halo, whose own
// results nobody stores and left of which no block can start -- every lane reads real cells, no ds_read is
// conditional.  Needs the tile grid aligned with the output (t.a - qs a multiple of 4: the lanes' cells are aligned
// in the level arrays); the caller falls back to halo_fold_store otherwise.
// History (profiles/r02_fold_in_registers.txt): the compiler's rendering of this fold (lane_shr1() + min3: a copy,
// a v_mov_dpp and a share of a min3 per shifted operand, all-ones stand-ins for lanes left of the array) took
// ~100 VALU instructions per wave at k = 31 -- 30 % of a sweep whose VALU is 70 % busy -- and lost to the LDS
// passes from six levels up; written as below it takes ~40, is 2-5 % of the whole sweep faster at k = 21 / 31,
// and wins up to seven levels (k = 64: 0.53 -> 0.48 ms; k = 101 / 128 on doubling arrays: -2 / -3 %).
// (fold_step_dpp<J>: memo_sweep_fold.h)

template <typename OutT, int T, int TOP>
__device__ __forceinline__ void halo_fold_store_dpp(const SweepArgs &A, const Tile &t, const uint32_t *lds) {
    const int LS = A.ls, HL = A.hl, W = A.w, nlev = A.nlev;
    const int cells = HL + W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NW = T / 64;
    const int ctx = nlev <= 1 ? 0 : (nlev <= 3 ? 1 : 1 << (nlev - 3));  // context lanes at the left of a wave
    const int valid = 64 - ctx;
    OutT *out = static_cast<OutT *>(A.out);
    const int64_t ob = t.a - A.qs - HL;  // output index of cell 0
    const int64_t o_lo = t.a - A.qs + t.x_lo, o_hi = t.a - A.qs + t.x_hi;
    for (int base = wave * 4 * valid; base + 4 * ctx < cells; base += NW * 4 * valid) {
        const int x0 = base + 4 * lane;             // this lane's cells x0 .. x0 + 3
        const uint32_t *p = lds + min(x0, LS - 4);  // (past the array: lanes whose results are dropped below)
        uint4 M = *reinterpret_cast<const uint4 *>(p);
        // blocks of 2^J fold in when that level exists (wave-uniform branches around straight-line steps)
        if (nlev >= 7) fold_step_dpp<5>(M, *reinterpret_cast<const uint4 *>(p + (nlev - 6) * LS), lane);
        if (nlev >= 6) fold_step_dpp<4>(M, *reinterpret_cast<const uint4 *>(p + (nlev - 5) * LS), lane);
        if (nlev >= 5) fold_step_dpp<3>(M, *reinterpret_cast<const uint4 *>(p + (nlev - 4) * LS), lane);
        if (nlev >= 4) fold_step_dpp<2>(M, *reinterpret_cast<const uint4 *>(p + (nlev - 3) * LS), lane);
        if (nlev >= 3) fold_step_dpp<1>(M, *reinterpret_cast<const uint4 *>(p + (nlev - 2) * LS), lane);
        if (nlev >= 2) fold_step_dpp<0>(M, *reinterpret_cast<const uint4 *>(p + (nlev - 1) * LS), lane);
        if (lane < ctx || x0 >= cells) continue;
        const int64_t g = ob + x0;
        if (g >= o_lo && g + 4 <= o_hi) {
            if constexpr (sizeof(OutT) == 1 && TOP == 24) {  // the four top bytes, two v_perm_b32 and an or
                store_four(out + g, __builtin_amdgcn_perm(M.y, M.x, 0x0c0c0703u) | __builtin_amdgcn_perm(M.w, M.z, 0x07030c0cu));
            } else {
                if (TOP) M = make_uint4(M.x >> TOP, M.y >> TOP, M.z >> TOP, M.w >> TOP);
                if constexpr (sizeof(OutT) == 1)
                    store_four(out + g, M.x | (M.y << 8) | (M.z << 16) | (M.w << 24));
                else
                    store_four(out + g, M.x | (M.y << 16), M.z | (M.w << 16));
            }
        } else {
            const uint32_t v[4] = {M.x >> TOP, M.y >> TOP, M.z >> TOP, M.w >> TOP};
            for (int i = 0; i < 4; ++i)
                if (g + i >= o_lo && g + i < o_hi) out[g + i] = (OutT)v[i];
        }
    }
}

#ifndef MEMO_FOLD_REG
#define MEMO_FOLD_REG 1
#endif
#ifndef MEMO_FOLD_DPP_LEVELS
#define MEMO_FOLD_DPP_LEVELS 7  // most levels folded in registers (<= 7: 2^(levels - 3) context lanes per wave)
#endif
template <typename OutT, int T, int TOP>
__device__ __forceinline__ void halo_finish(const SweepArgs &A, const Tile &t, uint32_t *lds) {
    if (MEMO_FOLD_REG && A.nlev <= MEMO_FOLD_DPP_LEVELS)  // (any window: store_four takes the address as it comes)
        halo_fold_store_dpp<OutT, T, TOP>(A, t, lds);
    else
        halo_fold_store<OutT, T, TOP>(A, t, lds);
}

// The row loops below run every row as ONE branch-free block: v_cmpx puts "this row writes" (n > 0) into EXEC itself,
// the address arithmetic and both ds_min run on those lanes only, and an s_mov restores EXEC (every lane of a wave
// is active in a row loop: rows outside the slice arrive as rows that cannot write).  The compiler's form of the
// same test -- v_cmp, s_and_saveexec, s_cbranch_execz, ..., s_or -- costs three scalar instructions and a branch
// per row, and the CU's one scalar unit was as busy as its vector pipes (1.3e8 SALU against 1.5e8 VALU
// wave-instructions per launch): dense rows, sustained, k = 21 / 31 / 64: 0.3218 -> 0.3133, 0.3249 -> 0.318,
// 0.383 -> 0.374 ms (profiles/r02_dense_rows_ab.txt).  MEMO_ROW_CMPX=0 builds the branchy form for A/B.
#ifndef MEMO_ROW_CMPX
#define MEMO_ROW_CMPX 1
#endif

// 4- and 6-byte rows.  The 4-byte rows carry their order in the top byte of the word, and the cells
// take the WORD (ds_min_u32 of the row as it was loaded: the min of the words has the min order on
// top, the junk below it only breaks ties) -- one VALU instruction per row less than extracting it;
// the store keeps the top bits (TOP = 24 / 20; needs num_docs to fit the field for the sentinel).  Otherwise the cells
// hold the order itself: the word's top byte shifted down, or the 16-bit order column.
template <typename Rows, int U, int T, typename OutT, int TOP>
__global__ __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(MEMO_HALO_WAVES, 8)))
void sweep_conservation_halo_kernel(const SweepArgs A) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    static_assert(TOP == 0 || (!Rows::kAnnot16 && TOP == Rows::kTopShift), "the order rides in the word only in the 4-byte formats");
    const int LS = A.ls, HL = A.hl, W = A.w;
    Tile t;
#ifdef MEMO_STAMPS
    unsigned long long stamp_t0 = __builtin_amdgcn_s_memtime();
#endif
    if (!locate_tile_w(A, t, W)) return;
    MEMO_STAMP(0);  // tile location (kernarg + two bucket-table loads)
    uint4 V[U];
    uint2 N[U];
    Rows::template issue<T, U>(A, t, 0, V, N);
    const uint32_t sent = (uint32_t)(A.ncols - 1);
    halo_clear<T>(A, lds, TOP ? (sent << TOP) | ((1u << TOP) - 1u) : sent);
    MEMO_STAMP(1);  // issue of the loads + LDS clear + barrier

    const int km1 = A.km1;
    // LDS byte address of tile slot x on the level with clz(n) = f:  base + 4 * ((f - fmin) * LS + HL + x)
    const uint32_t ls4 = 4u * (uint32_t)LS;
    const uint32_t lds_base = (uint32_t)(size_t)(__attribute__((address_space(3))) uint32_t *)lds;
    const uint32_t bias4 = pin_vgpr((int)(lds_base + 4u * (uint32_t)HL - (uint32_t)(32 - A.nlev) * ls4));
    const uint32_t top_bit = pin_vgpr((int)0x80000000u);
    const uint32_t key = pin_vgpr((int)Rows::tile_key(t.a));
    auto scatter = [&](uint32_t w, uint32_t col) {
        if (MEMO_ABLATE & 16) {  // keep the loads alive, nothing else
            asm volatile("" ::"v"(w), "v"(col));
            return;
        }
        if (MEMO_ROW_CMPX && !(MEMO_ABLATE & 3)) {  // n, the test, start - a, the two cells, both ds_min: one block
            uint32_t r0, r1, r2, nn;
            MEMO_EXEC_ALL_ONES(A.status);
            if constexpr (!Rows::kW12)
                asm volatile(
                    "v_sub_u32_sdwa %3, %5, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t"
                    "v_cmpx_lt_i32 vcc, 0, %3\n\t"
                    "v_sub_u16 %1, %4, %6\n\t"
                    "v_ffbh_u32 %0, %3\n\t"
                    "v_mad_u32_u24 %2, %0, %7, %8\n\t"
                    "v_lshl_add_u32 %2, %1, 2, %2\n\t"
                    "v_mad_i32_i24 %1, %3, -4, %2\n\t"
                    "v_ashrrev_i32 %0, %0, %9\n\t"
                    "v_lshl_add_u32 %2, %0, 2, %2\n\t"
                    "ds_min_u32 %1, %10\n\t"
                    "ds_min_u32 %2, %10\n\t"
                    "s_mov_b64 exec, -1"
                    : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(nn)
                    : "v"(w), "s"(km1), "v"(key), "s"(ls4), "v"(bias4), "v"(top_bit), "v"(TOP ? w : col)
                    : "memory", "vcc");
            else
                asm volatile(
                    "v_sub_u32_sdwa %3, %5, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"
                    "v_cmpx_lt_i32 vcc, 0, %3\n\t"
                    "v_sub_u32 %1, %4, %6\n\t"
                    "v_bfe_u32 %1, %1, 8, 12\n\t"
                    "v_ffbh_u32 %0, %3\n\t"
                    "v_mad_u32_u24 %2, %0, %7, %8\n\t"
                    "v_lshl_add_u32 %2, %1, 2, %2\n\t"
                    "v_mad_i32_i24 %1, %3, -4, %2\n\t"
                    "v_ashrrev_i32 %0, %0, %9\n\t"
                    "v_lshl_add_u32 %2, %0, 2, %2\n\t"
                    "ds_min_u32 %1, %10\n\t"
                    "ds_min_u32 %2, %10\n\t"
                    "s_mov_b64 exec, -1"
                    : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(nn)
                    : "v"(w), "s"(km1), "v"(key), "s"(ls4), "v"(bias4), "v"(top_bit), "v"(TOP ? w : col)
                    : "memory", "vcc");
            return;
        }
        const int n = km1 - Rows::len(w);  // length of [end - (k-1), start)
        if (n > 0) {
            // f = clz(n); 2^j = 2^31 >> f; x4 = address of cell `start` on level f;
            // ds_min into the cells of blocks [start - n, .. + 2^j) and [start - 2^j, start).
            // The compiler's own rendering of this needs 13 VALU instructions, one of them a
            // quarter-rate multiply.
            const uint32_t d = Rows::rel_start(w, key);  // start - a: v_sub_u16, or subtract + bit-field extract (12-bit form)
            uint32_t r0, r1, r2;
            asm volatile(
                "v_ffbh_u32 %0, %3\n\t"
                "v_mad_u32_u24 %2, %0, %5, %6\n\t"
                "v_lshl_add_u32 %2, %4, 2, %2\n\t"
                "v_mad_i32_i24 %1, %3, -4, %2\n\t"
                "v_ashrrev_i32 %0, %0, %7\n\t"
                "v_lshl_add_u32 %2, %0, 2, %2\n\t"
#if !(MEMO_ABLATE & 1)
                "ds_min_u32 %1, %8\n\t"
#endif
#if !(MEMO_ABLATE & 2)
                "ds_min_u32 %2, %8"
#endif
                : "=&v"(r0), "=&v"(r1), "=&v"(r2)
                : "v"(n), "v"(d), "s"(ls4), "v"(bias4), "v"(top_bit), "v"(TOP ? w : col)
                : "memory");
        }
    };
    Rows::template consume<T, U>(A, t, 0, V, N, scatter);
    for (uint32_t b = 1, nb = Rows::template batches<T, U>(t); b < nb; ++b) {  // a dense tile: the rest
        Rows::template issue<T, U>(A, t, b, V, N);
        Rows::template consume<T, U>(A, t, b, V, N, scatter);
    }
    MEMO_STAMP(2);  // waiting for rows + scatter
    lds_barrier();  // waits for lgkmcnt(0): the ds_min above are invisible to the compiler
    MEMO_STAMP(3);  // barrier after the scatter
    halo_finish<OutT, T, TOP>(A, t, lds);
    MEMO_STAMP(5);  // folds + store
#ifdef MEMO_STAMPS
    if (threadIdx.x == 0 && A.stamps) A.stamps[8ull * blockIdx.x + 7] = 1;
#endif
}

// The same sweep on the dense rows (PackedRows3: five rows per 16 bytes; k - 1 <= 63, level arrays of at most
// 1024 cells).  Per row: 16-bit subtract (start - a, length untouched below it), and, subtract, compare | ffbh,
// bfe, mad, lshl_add, mad, ashr, lshl_add, ds_min x 2 (four rows of five as they were loaded -- the dword is the
// ds_min operand; the fifth after a v_perm_b32 and a shift).
// Where it stands (profiles/r02_dense_rows_ab.txt): back to back -- the device settled at its power cap -- this kernel
// takes 0.324 ms on config 3 where the 4-byte kernel takes 0.374 (another device: 0.331 against 0.343): 19 % fewer bytes,
// 13 % less time; what keeps it from the full 19 % is the power cap (1370 W of 1400 at 2.2 GHz; the 4-byte kernel
// runs at 2.36 GHz), i.e. what it executes per row.  Moving the arithmetic out of the row loop into a per-tile table
// indexed by the length (64 entries of two LDS offsets: 4 VALU per row fewer, one ds_read_b64 more) made it slower
// (0.46 ms: every row then waits for an LDS round trip).
template <int U, int T, typename OutT, bool A9 = false>
__global__ __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(MEMO_HALO_WAVES, 8)))
void sweep_conservation_halo3_kernel(const SweepArgs A) {
    static_assert(!A9 || sizeof(OutT) == 2, "more than 255 genomes: uint16 results");
    constexpr int TOP = A9 ? 23 : 24;  // a cell = order << TOP | tie-breaking bits
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    using Rows = PackedRows3;
    const int LS = A.ls, HL = A.hl, W = A.w;
    Tile t;
    if (!locate_tile_w(A, t, W)) return;
    uint4 V[U];
    Rows::template issue<T, U>(A, t, 0, V);
    // diagnostic builds only (tools/build_variant.sh): what is the sweep short of?  N more scalar / vector instructions
    // per wave that do nothing -- if the time follows the scalar ones, the CU's one scalar unit is the bound
#if defined(MEMO_EXTRA_SALU) || defined(MEMO_EXTRA_VALU)
    {
        uint32_t sx = (uint32_t)A.km1, vx = threadIdx.x;
#ifdef MEMO_EXTRA_SALU
#pragma unroll
        for (int i = 0; i < MEMO_EXTRA_SALU; ++i) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sx)::"scc");
#endif
#ifdef MEMO_EXTRA_VALU
#pragma unroll
        for (int i = 0; i < MEMO_EXTRA_VALU; ++i) asm volatile("v_add_u32 %0, %0, 1" : "+v"(vx));
#endif
        if (sx + vx == 0xFFFFFFF0u) atomicOr(A.status, 64);  // (keeps them alive)
    }
#endif
    halo_clear<T>(A, lds, ((uint32_t)(A.ncols - 1) << TOP) | ((1u << TOP) - 1u));

    const int km1 = A.km1;
    const uint32_t ls4 = 4u * (uint32_t)LS;
    const uint32_t lds_base = (uint32_t)(size_t)(__attribute__((address_space(3))) uint32_t *)lds;
    const uint32_t bias4 = pin_vgpr((int)(lds_base + 4u * (uint32_t)HL - (uint32_t)(32 - A.nlev) * ls4));
    const uint32_t top_bit = pin_vgpr((int)0x80000000u);
    const uint32_t a10s = pin_vgpr((int)(((uint32_t)t.a & 1023u) << 6));
    // r = (start - a) mod 2^10 << 6 | length;  data = a word with the row's order in its top byte
    auto scatter = [&](uint32_t r, uint32_t data) {
        if (MEMO_ABLATE & 16) {  // keep the loads alive, nothing else
            asm volatile("" ::"v"(r), "v"(data));
            return;
        }
        const int n = km1 - (int)(r & 63u);
        if (n > 0) {
            uint32_t r0, r1, r2;
            asm volatile(
                "v_ffbh_u32 %0, %3\n\t"
                "v_bfe_u32 %1, %4, 6, 10\n\t"
                "v_mad_u32_u24 %2, %0, %5, %6\n\t"
                "v_lshl_add_u32 %2, %1, 2, %2\n\t"
                "v_mad_i32_i24 %1, %3, -4, %2\n\t"
                "v_ashrrev_i32 %0, %0, %7\n\t"
                "v_lshl_add_u32 %2, %0, 2, %2\n\t"
                "ds_min_u32 %1, %8\n\t"
                "ds_min_u32 %2, %8"
                : "=&v"(r0), "=&v"(r1), "=&v"(r2)
                : "v"(n), "v"(r), "s"(ls4), "v"(bias4), "v"(top_bit), "v"(data)
                : "memory");
        }
    };
    auto g = [&](uint32_t b, uint32_t data) {  // 16-bit subtract on the low half; the result's high half is zero
        if (MEMO_ROW_CMPX) {
            // the whole row in one block, no branch: v_cmpx puts "this row writes" into EXEC itself, everything after it
            // runs on those lanes only, s_mov restores EXEC (every lane of the wave is active in the row loop)
            uint32_t r0, r1, r2, r3;
            MEMO_EXEC_ALL_ONES(A.status);
            asm volatile(
                "v_sub_u16 %3, %4, %5\n\t"
                "v_and_b32 %0, 63, %3\n\t"
                "v_sub_u32 %0, %6, %0\n\t"
                "v_cmpx_lt_i32 vcc, 0, %0\n\t"
                "v_ffbh_u32 %1, %0\n\t"
                "v_bfe_u32 %3, %3, 6, 10\n\t"
                "v_mad_u32_u24 %2, %1, %7, %8\n\t"
                "v_lshl_add_u32 %2, %3, 2, %2\n\t"
                "v_mad_i32_i24 %3, %0, -4, %2\n\t"
                "v_ashrrev_i32 %1, %1, %9\n\t"
                "v_lshl_add_u32 %2, %1, 2, %2\n\t"
                "ds_min_u32 %3, %10\n\t"
                "ds_min_u32 %2, %10\n\t"
                "s_mov_b64 exec, -1"
                : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
                : "v"(b), "v"(a10s), "s"(km1), "s"(ls4), "v"(bias4), "v"(top_bit), "v"(data)
                : "memory", "vcc");
            return;
        }
        uint32_t r;
        asm("v_sub_u16 %0, %1, %2" : "=v"(r) : "v"(b), "v"(a10s));
        scatter(r, data);
    };
    Rows::template consume<T, U, decltype(g), A9>(A, t, 0, V, g);
    for (uint32_t b = 1, nb = Rows::template batches<T, U>(t); b < nb; ++b) {  // a dense tile: the rest
        Rows::template issue<T, U>(A, t, b, V);
        Rows::template consume<T, U, decltype(g), A9>(A, t, b, V, g);
    }
    lds_barrier();  // waits for lgkmcnt(0): the ds_min above are invisible to the compiler
    halo_finish<OutT, T, TOP>(A, t, lds);
}

// ------------------------------------------------------------------------------------------
// conservation, unclipped, RADIX-4 levels: blocks of 1, 4, 16, 64 positions.
//
// Doubling levels cost 4 bytes of LDS per position and level -- 28 B at k = 101, 32 B at k = 256 -- and one
// fold step per level; at large k those folds and the k-1 halo of short tiles are what the sweep spends its
// time on (profiles/r02_ablation.txt: fold passes 0.11 of 0.63 ms at k = 101).  With levels a factor FOUR
// apart there are four arrays for any k <= 256: an interval of n positions, 4^i <= n < 4^(i+1), is covered
// by blocks of S = 4^i at start - n and start - S, plus one at start - n + S when n > 2S and one at
// start - n + 2S when n > 3S (2 - 4 ds_min per row instead of 2), and a block of 4S folds into the four
// blocks of S under it.  Tiles are twice as long in the same LDS, the halo's share of the array shrinks with
// them, and three fold steps replace six or seven: 64 -> 16 as one LDS pass, 16 -> 4 and 4 -> 1 in registers
// (DPP shifts only).  Used from k = 65 up; below that the doubling arrays are faster (see the launcher).
// ------------------------------------------------------------------------------------------
template <typename OutT, int T, int TOP>
__device__ __forceinline__ void r4_fold_store(const SweepArgs &A, const Tile &t, uint32_t *lds) {
    const int LS = A.ls, HL = A.hl, W = A.w, m = A.nlev;
    const int cells = HL + W;
    // levels above 16 fold down through LDS (shifts of 16 cells and more are not a DPP's reach).  (The mixed arrays have a fold of
    // their own since round 4: plan_fold_store.)
    // Radix-4 levels: slot s holds blocks of 4^(m-1-s); after this loop slot m-3 (blocks of 16) has everything
    // above it folded in
    for (int slot = 0; slot + 3 < m; ++slot) {
        const int S = 1 << (2 * (m - 2 - slot));  // size of the blocks being folded INTO (>= 16)
        const uint32_t *hi = lds + slot * LS;
        uint32_t *lo = lds + (slot + 1) * LS;
        for (int x = 4 * threadIdx.x; x < cells; x += 4 * T) {
            uint4 r = *reinterpret_cast<const uint4 *>(lo + x);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (x >= q * S) {
                    const uint4 v = *reinterpret_cast<const uint4 *>(hi + x - q * S);
                    r.x = min(r.x, v.x);
                    r.y = min(r.y, v.y);
                    r.z = min(r.z, v.z);
                    r.w = min(r.w, v.w);
                }
            }
            *reinterpret_cast<uint4 *>(lo + x) = r;
        }
        lds_barrier();
    }
    // 16 -> 4 and 4 -> 1 in registers, as in halo_fold_store_dpp: four cells per lane, lane 0 of a wave on cell
    // `base` (wave 0's context lanes hold the first cells of the left halo), operands from the lanes to the left
    // through DPP, one in-place asm block per step (s_nop 1: the two wait states a DPP source written by the
    // instruction before needs).
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NW = T / 64;
    const int ctx = m >= 3 ? 4 : (m == 2 ? 1 : 0);  // context lanes: 15 / 3 / 0 cells to the left
    const int valid = 64 - ctx;
    OutT *out = static_cast<OutT *>(A.out);
    const int64_t ob = t.a - A.qs - HL;  // output index of cell 0
    const int64_t o_lo = t.a - A.qs + t.x_lo, o_hi = t.a - A.qs + t.x_hi;
#define MEMO_DPP_MIN(dst, src) "v_min_u32_dpp " dst ", " src ", " dst " wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define MEMO_DPP_MOV(dst, src) "v_mov_b32_dpp " dst ", " src " wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
    for (int base = wave * 4 * valid; base + 4 * ctx < cells; base += NW * 4 * valid) {
        const int x0 = base + 4 * lane;             // this lane's cells x0 .. x0 + 3
        const uint32_t *p = lds + min(x0, LS - 4);  // (past the array: lanes whose results are dropped below)
        uint4 R = *reinterpret_cast<const uint4 *>(p + (m - 1) * LS);  // blocks of 1
        if (m >= 2) {
            uint4 M = *reinterpret_cast<const uint4 *>(p + (m - 2) * LS);  // blocks of 4
            if (m >= 3) {
                // blocks of 16 -> blocks of 4: cells x - 4, x - 8, x - 12 are the same component 1, 2, 3 lanes left.
                // B = min over two lanes in place, P = B one lane left; M = min(M, B, P one more lane left)
                uint4 B = *reinterpret_cast<const uint4 *>(p + (m - 3) * LS), P;
                asm("s_nop 1\n\t" MEMO_DPP_MIN("%4", "%4") MEMO_DPP_MIN("%5", "%5") MEMO_DPP_MIN("%6", "%6") MEMO_DPP_MIN("%7", "%7")
                    MEMO_DPP_MOV("%8", "%4") MEMO_DPP_MOV("%9", "%5") MEMO_DPP_MOV("%10", "%6") MEMO_DPP_MOV("%11", "%7")
                    "v_min_u32 %0, %4, %0\n\tv_min_u32 %1, %5, %1\n\tv_min_u32 %2, %6, %2\n\tv_min_u32 %3, %7, %3\n\t"
                    MEMO_DPP_MIN("%0", "%8") MEMO_DPP_MIN("%1", "%9") MEMO_DPP_MIN("%2", "%10") MEMO_DPP_MIN("%3", "%11")
                    : "+v"(M.x), "+v"(M.y), "+v"(M.z), "+v"(M.w), "+v"(B.x), "+v"(B.y), "+v"(B.z), "+v"(B.w),
                      "=&v"(P.x), "=&v"(P.y), "=&v"(P.z), "=&v"(P.w));  // (P of lane 0: whatever was there; a context lane)
            }
            // blocks of 4 -> positions: cell x takes the blocks at x, x - 1, x - 2, x - 3 (the last ones of the lane to the left)
            asm("s_nop 1\n\t"
                "v_min3_u32 %3, %3, %7, %6\n\tv_min3_u32 %3, %3, %5, %4\n\t"
                "v_min3_u32 %2, %2, %6, %5\n\tv_min_u32 %2, %2, %4\n\t"
                "v_min3_u32 %1, %1, %5, %4\n\tv_min_u32 %0, %0, %4\n\t"
                MEMO_DPP_MIN("%2", "%7") MEMO_DPP_MIN("%1", "%7") MEMO_DPP_MIN("%0", "%7")
                MEMO_DPP_MIN("%1", "%6") MEMO_DPP_MIN("%0", "%6") MEMO_DPP_MIN("%0", "%5")
                : "+v"(R.x), "+v"(R.y), "+v"(R.z), "+v"(R.w) : "v"(M.x), "v"(M.y), "v"(M.z), "v"(M.w));
        }
        if (lane < ctx || x0 >= cells) continue;
        const int64_t g = ob + x0;
        if (g >= o_lo && g + 4 <= o_hi) {
            if constexpr (sizeof(OutT) == 1 && TOP == 24) {  // the four top bytes, two v_perm_b32 and an or
                store_four(out + g, __builtin_amdgcn_perm(R.y, R.x, 0x0c0c0703u) | __builtin_amdgcn_perm(R.w, R.z, 0x07030c0cu));
            } else {
                if (TOP) R = make_uint4(R.x >> TOP, R.y >> TOP, R.z >> TOP, R.w >> TOP);
                if constexpr (sizeof(OutT) == 1)
                    store_four(out + g, R.x | (R.y << 8) | (R.z << 16) | (R.w << 24));
                else
                    store_four(out + g, R.x | (R.y << 16), R.z | (R.w << 16));
            }
        } else {  // window edges
            const uint32_t v[4] = {R.x >> TOP, R.y >> TOP, R.z >> TOP, R.w >> TOP};
            for (int i = 0; i < 4; ++i)
                if (g + i >= o_lo && g + i < o_hi) out[g + i] = (OutT)v[i];
        }
    }
#undef MEMO_DPP_MIN
#undef MEMO_DPP_MOV
}

template <typename Rows, int U, int T, typename OutT, int TOP>
__global__ __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(MEMO_HALO_WAVES, 8)))
void sweep_conservation_r4_kernel(const SweepArgs A) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    static_assert(TOP == 0 || (!Rows::kAnnot16 && TOP == Rows::kTopShift), "the order rides in the word only in the 4-byte formats");
    const int LS = A.ls, HL = A.hl, W = A.w;
    Tile t;
    if (!locate_tile_w(A, t, W)) return;
    uint4 V[U];
    uint2 N[U];
    Rows::template issue<T, U>(A, t, 0, V, N);
    const uint32_t sent = (uint32_t)(A.ncols - 1);
    halo_clear<T>(A, lds, TOP ? (sent << TOP) | ((1u << TOP) - 1u) : sent);

    const int km1 = A.km1;
    // LDS byte address of tile slot x on level i (blocks of 4^i):  level0 - i * 4 LS + 4 x
    const uint32_t ls4 = 4u * (uint32_t)LS;
    const uint32_t lds_base = (uint32_t)(size_t)(__attribute__((address_space(3))) uint32_t *)lds;
    // level i (blocks of 4^i) sits at level0 - i * 4 LS; with i' = clz(n) >> 1 = 15 - i:  levelK + i' * 4 LS
    const uint32_t levelK = pin_vgpr((int)(lds_base + (uint32_t)(A.nlev - 1) * ls4 + 4u * (uint32_t)HL - 15u * ls4));
    const uint32_t top_bit = pin_vgpr((int)0x80000000u);
    const uint32_t key = pin_vgpr((int)Rows::tile_key(t.a));
    auto scatter = [&](uint32_t w, uint32_t col) {
        if (MEMO_ROW_CMPX) {
            // the first two blocks of a row as one branch-free block (EXEC narrowed to "n > 0" by v_cmpx, restored at the end);
            // the third and fourth stay behind the compiler's branches: where no row of a wave needs them (k = 128 on config 3)
            // skipping them beats issuing them with no lane -- 0.395 against 0.415 ms (profiles/r02_mixed_levels.txt)
            uint32_t tmp, a1, a2, s4, q, dd;
            MEMO_EXEC_ALL_ONES(A.status);
            // diagnostic builds (tools/build_variant.sh): what do the lanes of one instruction on ONE cell cost -- rows of a start
            // and a level share their second block's cell?  32: the second ds_min goes to the first block's cell (as many instructions,
            // no two lanes on one address but for equal rows); 64: no second ds_min.  Wrong results either way.
#if MEMO_ABLATE & 32
#define MEMO_R4_SECOND "ds_min_u32 %1, %11\n\t"
#elif MEMO_ABLATE & 64
#define MEMO_R4_SECOND
#else
#define MEMO_R4_SECOND "ds_min_u32 %2, %11\n\t"
#endif
#define MEMO_R4_BLOCK(LEN_SEL, REL_START)                                                                         \
            asm volatile(                                                                                          \
                "v_mov_b32 %4, 0\n\t"                 /* q = 0 where the row does not write */                     \
                "v_sub_u32_sdwa %0, %7, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:" LEN_SEL "\n\t" \
                "v_cmpx_lt_i32 vcc, 0, %0\n\t"                                                                    \
                REL_START                                                                                          \
                "v_ffbh_u32 %4, %0\n\t"                                                                           \
                "v_lshrrev_b32 %4, 1, %4\n\t"        /* i' = 15 - i */                                            \
                "v_mad_u32_u24 %1, %4, %9, %10\n\t"  /* level i */                                                \
                "v_lshl_add_u32 %2, %5, 2, %1\n\t"   /* cell `start` on level i (%5 = start - a) */               \
                "v_mad_i32_i24 %1, %0, -4, %2\n\t"   /* a1: cell start - n */                                     \
                "v_lshl_add_u32 %4, %4, 1, -1\n\t"   /* 2 i' - 1 = 29 - 2 i */                                    \
                "v_lshrrev_b32 %3, %4, %12\n\t"      /* 4 S (bytes) */                                            \
                "v_sub_u32 %2, %2, %3\n\t"           /* a2: cell start - S */                                     \
                "ds_min_u32 %1, %11\n\t"                                                                          \
                MEMO_R4_SECOND                                                                                     \
                "v_sub_u32 %4, 29, %4\n\t"           /* 2 i */                                                    \
                "v_lshrrev_b32 %4, %4, %0\n\t"       /* q */                                                      \
                "s_mov_b64 exec, -1"                                                                               \
                : "=&v"(tmp), "=&v"(a1), "=&v"(a2), "=&v"(s4), "=&v"(q), "=&v"(dd)                                 \
                : "v"(w), "s"(km1), "v"(key), "s"(ls4), "v"(levelK), "v"(TOP ? w : col), "v"(top_bit)              \
                : "memory", "vcc")
            if constexpr (!Rows::kW12)
                MEMO_R4_BLOCK("BYTE_2", "v_sub_u16 %5, %6, %8\n\t");
            else
                MEMO_R4_BLOCK("BYTE_0", "v_sub_u32 %5, %6, %8\n\tv_bfe_u32 %5, %5, 8, 12\n\t");
#undef MEMO_R4_BLOCK
#undef MEMO_R4_SECOND
            if (q >= 2) {
                const uint32_t data = TOP ? w : col;
                uint32_t a3, a4;
                asm volatile(
                    "v_add_u32 %0, %1, %2\n\t"        // a3 = a1 + 4 S
                    "ds_min_u32 %0, %3"
                    : "=&v"(a3)
                    : "v"(a1), "v"(s4), "v"(data)
                    : "memory");
                if (q == 3) {
                    asm volatile(
                        "v_add_u32 %0, %1, %2\n\t"    // a4 = a3 + 4 S
                        "ds_min_u32 %0, %3"
                        : "=&v"(a4)
                        : "v"(a3), "v"(s4), "v"(data)
                        : "memory");
                }
            }
            return;
        }
        const int n = km1 - Rows::len(w);  // length of [end - (k-1), start)
        if (n > 0) {
            // i = floor(log4 n), S = 4^i, q = n >> 2i (the leading base-4 digit).  Blocks [start - n, +S) and
            // [start - S, start); q >= 2: one more at start - n + S; q = 3: and one at start - n + 2S.  By hand: 14 - 16
            // VALU instructions per row after the length / validity / start triple (the compiler's rendering took 24,
            // one of them a quarter-rate 32-bit multiply; the first hand-written one 16 - 19).
            const uint32_t data = TOP ? w : col;
            const uint32_t d = Rows::rel_start(w, key);  // start - a
            uint32_t tmp, a1, a2, s4, q;
            asm volatile(
                "v_ffbh_u32 %0, %5\n\t"
                "v_lshrrev_b32 %0, 1, %0\n\t"        // i' = 15 - i
                "v_mad_u32_u24 %1, %0, %7, %8\n\t"   // level i
                "v_lshl_add_u32 %2, %6, 2, %1\n\t"   // cell `start` on level i
                "v_mad_i32_i24 %1, %5, -4, %2\n\t"   // a1: cell start - n
                "v_lshl_add_u32 %0, %0, 1, -1\n\t"   // 2 i' - 1 = 29 - 2 i
                "v_lshrrev_b32 %3, %0, %10\n\t"      // 4 S (bytes) = 2^31 >> (29 - 2 i)
                "v_sub_u32 %2, %2, %3\n\t"           // a2: cell start - S
                "ds_min_u32 %1, %9\n\t"
                "ds_min_u32 %2, %9\n\t"
                "v_sub_u32 %0, 29, %0\n\t"           // 2 i
                "v_lshrrev_b32 %4, %0, %5"             // q
                : "=&v"(tmp), "=&v"(a1), "=&v"(a2), "=&v"(s4), "=&v"(q)
                : "v"(n), "v"(d), "s"(ls4), "v"(levelK), "v"(data), "v"(top_bit)
                : "memory");
            if (q >= 2) {
                uint32_t a3, a4;
                asm volatile(
                    "v_add_u32 %0, %1, %2\n\t"        // a3 = a1 + 4 S
                    "ds_min_u32 %0, %3"
                    : "=&v"(a3)
                    : "v"(a1), "v"(s4), "v"(data)
                    : "memory");
                if (q == 3) {
                    asm volatile(
                        "v_add_u32 %0, %1, %2\n\t"    // a4 = a3 + 4 S
                        "ds_min_u32 %0, %3"
                        : "=&v"(a4)
                        : "v"(a3), "v"(s4), "v"(data)
                        : "memory");
                }
            }
        }
    };
    Rows::template consume<T, U>(A, t, 0, V, N, scatter);
    for (uint32_t b = 1, nb = Rows::template batches<T, U>(t); b < nb; ++b) {  // a dense tile: the rest
        Rows::template issue<T, U>(A, t, b, V, N);
        Rows::template consume<T, U>(A, t, b, V, N, scatter);
    }
    lds_barrier();  // waits for lgkmcnt(0): the ds_min above are invisible to the compiler
    __syncthreads();
    r4_fold_store<OutT, T, TOP>(A, t, lds);
}

// ------------------------------------------------------------------------------------------
// Level plan of the mixed arrays (round 4): only the arrays some row of the index can write to are allocated, cleared and
// folded.  Which lengths n = k - 1 - overlap occur at a k follows from WHICH overlaps occur in the index -- known exactly
// since the rows were packed (memo_index::len_seen) -- and on BASELINE's generator (overlaps 0 .. 59) a k = 256 query
// has every interval in 196 .. 255: blocks of 128 only.  Six arrays (1, 4, 16, 32, 64, 128) become two (128, and the
// blocks of 16 the fold goes through), the tile grows from 1120 to 4576 positions in the same 40 KiB, and what a tile
// spends on clearing, folding and its k - 1 halo shrinks with them.  An index whose overlaps reach every level (rows
// built from sequences) gets the arrays it always got.
//   slots 0 .. D-1   the doubling RANGE: blocks of 2^(31 - ftop - d), from the largest populated size down to the
//                    smallest populated size >= 16 (a slot inside the range may be unpopulated: never cleared, never read)
//   then             the blocks of 16 when the range ends above them (the fold's target: written before it is read)
//   then             blocks of 4, blocks of 1 -- where some row can have fewer than 16 positions
// A.lvmask: bits 0..7 = populated slots of the range, kPlanSep16 / kPlanHas4 / kPlanHas1; A.nlev = D; A.ftop.
// ------------------------------------------------------------------------------------------
constexpr uint32_t kPlanSep16 = 1u << 16, kPlanHas4 = 1u << 17, kPlanHas1 = 1u << 18;

struct LevelPlan {
    int D, sep, has4, has1;
    __host__ __device__ int s16() const { return sep ? D : D - 1; }
    __host__ __device__ int s4() const { return D + sep; }
    __host__ __device__ int s1() const { return D + sep + has4; }
    __host__ __device__ int arrays() const { return D + sep + has4 + has1; }
};

__host__ __device__ inline LevelPlan plan_of(int D, uint32_t mask) {
    LevelPlan p;
    p.D = D;
    p.sep = (mask & kPlanSep16) ? 1 : 0;
    p.has4 = (mask & kPlanHas4) ? 1 : 0;
    p.has1 = (mask & kPlanHas1) ? 1 : 0;
    return p;
}

// r = min(r, the N blocks of a level at x, x - 16, ..., x - 16 (N - 1)): every read issued before the first is used (the
// loop form kept one read in flight: k = 256, fifteen reads per cell quartet, 0.55 -> 0.64 ms).  Cells left of the array do
// not exist: such a read repeats the leftmost one that does (min is idempotent), so no read is conditional.
template <int N>
__device__ __forceinline__ void fold_blocks_back(const uint32_t *hi, int x, uint4 &r) {
    const int room = x & ~15;  // the furthest multiple of 16 this lane may step back
    uint4 v[N];
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = *reinterpret_cast<const uint4 *>(hi + x - min(16 * i, room));
#pragma unroll
    for (int i = 0; i < N; ++i) {
        r.x = min(r.x, v[i].x);
        r.y = min(r.y, v[i].y);
        r.z = min(r.z, v[i].z);
        r.w = min(r.w, v[i].w);
    }
}

template <int T>
__device__ __forceinline__ void plan_clear(const SweepArgs &A, uint32_t *lds, uint32_t sent) {
    const LevelPlan P = plan_of(A.nlev, A.lvmask);
    const uint4 sv = make_uint4(sent, sent, sent, sent);
    const int per = A.ls / 4;
    // which arrays start at the sentinel: the populated slots of the range (one no row writes to is never read) and the blocks
    // of 4 / of 1; the separate target of the fold is written before it is read.  (Unrolled over the eight arrays a plan can
    // have, each behind a wave-uniform test: straight-line code the EXEC scan of tests/test_host_cpu.py can follow.)
    uint32_t wanted = A.lvmask & ((1u << P.D) - 1u);
    if (P.has4) wanted |= 1u << P.s4();
    if (P.has1) wanted |= 1u << P.s1();
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        if ((wanted >> a) & 1u) {
            uint4 *p = reinterpret_cast<uint4 *>(lds + a * A.ls);
            for (int i = threadIdx.x; i < per; i += T) p[i] = sv;
        }
    }
    lds_barrier();
}

template <typename OutT, int T, int TOP>
__device__ __forceinline__ void plan_fold_store(const SweepArgs &A, const Tile &t, uint32_t *lds) {
    const int LS = A.ls, HL = A.hl, W = A.w;
    const LevelPlan P = plan_of(A.nlev, A.lvmask);
    const int cells = HL + W;
    // every populated level above 16 into the blocks of 16, in ONE LDS pass: a block of 16 * 2^e at x covers the blocks
    // of 16 at x, x + 16, ..., x + 16 (2^e - 1)
    const int top = P.sep ? P.D - 1 : P.D - 2;  // last range slot above the blocks of 16
    if (top >= 0) {
        uint32_t *lo = lds + P.s16() * LS;
        for (int x = 4 * threadIdx.x; x < cells; x += 4 * T) {
            // (one 16-byte read whatever the plan: written as `sep ? ones : load` the compiler split it into four guarded
            // 4-byte reads -- k = 256 0.55 -> 0.64 ms; a separate target holds nothing yet, what is read there is dropped)
            uint4 r = *reinterpret_cast<const uint4 *>(lo + x);
            if (P.sep) r = make_uint4(~0u, ~0u, ~0u, ~0u);
            for (int slot = top; slot >= 0; --slot) {
                if (!((A.lvmask >> slot) & 1u)) continue;
                const uint32_t *hi = lds + slot * LS;
                switch (31 - A.ftop - slot) {  // log2 of this level's blocks (wave-uniform)
                    case 5: fold_blocks_back<2>(hi, x, r); break;
                    case 6: fold_blocks_back<4>(hi, x, r); break;
                    default: fold_blocks_back<8>(hi, x, r); break;  // blocks of 128 (k - 1 <= 255)
                }
            }
            *reinterpret_cast<uint4 *>(lo + x) = r;
        }
        lds_barrier();
    }
    // 16 -> 4 and 4 -> 1 in registers (r4_fold_store's last part; arrays that do not exist read as "no row")
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NW = T / 64;
    constexpr int ctx = 4, valid = 64 - ctx;  // context lanes: 15 cells to the left
    OutT *out = static_cast<OutT *>(A.out);
    const int64_t ob = t.a - A.qs - HL;  // output index of cell 0
    const int64_t o_lo = t.a - A.qs + t.x_lo, o_hi = t.a - A.qs + t.x_hi;
    // (arrays the plan does not have are read where the blocks of 16 are -- one unconditional 16-byte read each -- and
    // replaced by "no row" afterwards)
    const uint32_t *L16 = lds + P.s16() * LS, *L4 = lds + (P.has4 ? P.s4() : P.s16()) * LS,
                   *L1 = lds + (P.has1 ? P.s1() : P.s16()) * LS;
#define MEMO_DPP_MIN(dst, src) "v_min_u32_dpp " dst ", " src ", " dst " wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define MEMO_DPP_MOV(dst, src) "v_mov_b32_dpp " dst ", " src " wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
    for (int base = wave * 4 * valid; base + 4 * ctx < cells; base += NW * 4 * valid) {
        const int x0 = base + 4 * lane;             // this lane's cells x0 .. x0 + 3
        const int xr = min(x0, LS - 4);             // (past the array: lanes whose results are dropped below)
        const uint4 ones = make_uint4(~0u, ~0u, ~0u, ~0u);
        uint4 R = *reinterpret_cast<const uint4 *>(L1 + xr);  // blocks of 1
        uint4 M = *reinterpret_cast<const uint4 *>(L4 + xr);  // blocks of 4
        if (!P.has1) R = ones;
        if (!P.has4) M = ones;
        {
            // blocks of 16 -> blocks of 4: cells x - 4, x - 8, x - 12 are the same component 1, 2, 3 lanes left.
            // B = min over two lanes in place, P = B one lane left; M = min(M, B, P one more lane left)
            uint4 B = *reinterpret_cast<const uint4 *>(L16 + xr), Q;
            asm("s_nop 1\n\t" MEMO_DPP_MIN("%4", "%4") MEMO_DPP_MIN("%5", "%5") MEMO_DPP_MIN("%6", "%6") MEMO_DPP_MIN("%7", "%7")
                MEMO_DPP_MOV("%8", "%4") MEMO_DPP_MOV("%9", "%5") MEMO_DPP_MOV("%10", "%6") MEMO_DPP_MOV("%11", "%7")
                "v_min_u32 %0, %4, %0\n\tv_min_u32 %1, %5, %1\n\tv_min_u32 %2, %6, %2\n\tv_min_u32 %3, %7, %3\n\t"
                MEMO_DPP_MIN("%0", "%8") MEMO_DPP_MIN("%1", "%9") MEMO_DPP_MIN("%2", "%10") MEMO_DPP_MIN("%3", "%11")
                : "+v"(M.x), "+v"(M.y), "+v"(M.z), "+v"(M.w), "+v"(B.x), "+v"(B.y), "+v"(B.z), "+v"(B.w),
                  "=&v"(Q.x), "=&v"(Q.y), "=&v"(Q.z), "=&v"(Q.w));  // (Q of lane 0: whatever was there; a context lane)
        }
        // blocks of 4 -> positions: cell x takes the blocks at x, x - 1, x - 2, x - 3 (the last ones of the lane to the left)
        asm("s_nop 1\n\t"
            "v_min3_u32 %3, %3, %7, %6\n\tv_min3_u32 %3, %3, %5, %4\n\t"
            "v_min3_u32 %2, %2, %6, %5\n\tv_min_u32 %2, %2, %4\n\t"
            "v_min3_u32 %1, %1, %5, %4\n\tv_min_u32 %0, %0, %4\n\t"
            MEMO_DPP_MIN("%2", "%7") MEMO_DPP_MIN("%1", "%7") MEMO_DPP_MIN("%0", "%7")
            MEMO_DPP_MIN("%1", "%6") MEMO_DPP_MIN("%0", "%6") MEMO_DPP_MIN("%0", "%5")
            : "+v"(R.x), "+v"(R.y), "+v"(R.z), "+v"(R.w) : "v"(M.x), "v"(M.y), "v"(M.z), "v"(M.w));
        if (lane < ctx || x0 >= cells) continue;
        const int64_t g = ob + x0;
        if (g >= o_lo && g + 4 <= o_hi) {
            if constexpr (sizeof(OutT) == 1 && TOP == 24) {  // the four top bytes, two v_perm_b32 and an or
                store_four(out + g, __builtin_amdgcn_perm(R.y, R.x, 0x0c0c0703u) | __builtin_amdgcn_perm(R.w, R.z, 0x07030c0cu));
            } else {
                if (TOP) R = make_uint4(R.x >> TOP, R.y >> TOP, R.z >> TOP, R.w >> TOP);
                if constexpr (sizeof(OutT) == 1)
                    store_four(out + g, R.x | (R.y << 8) | (R.z << 16) | (R.w << 24));
                else
                    store_four(out + g, R.x | (R.y << 16), R.z | (R.w << 16));
            }
        } else {  // window edges
            const uint32_t v[4] = {R.x >> TOP, R.y >> TOP, R.z >> TOP, R.w >> TOP};
            for (int i = 0; i < 4; ++i)
                if (g + i >= o_lo && g + i < o_hi) out[g + i] = (OutT)v[i];
        }
    }
#undef MEMO_DPP_MIN
#undef MEMO_DPP_MOV
}

// ------------------------------------------------------------------------------------------
// conservation, unclipped, MIXED levels: blocks of 1, 4, 16 and then doubling -- 32, 64, 128.
//
// What the radix-4 arrays cost is LDS atomics: an interval of n positions with 2S < n < 4S takes three or four
// blocks of S, and the sweep's time follows the blocks per row (config 3, profiles/r02_mixed_levels.txt: k = 128,
// where every interval has 64 <= n < 128 and takes two blocks, 0.39 ms; k = 101, 2.6 blocks per row, 0.515;
// k = 200, 3.5 blocks, 0.545).  At k >= 65 most intervals are long, and from 16 positions up these arrays go in
// steps of two: n >= 16 takes the two blocks of 2^floor(log2 n) of the doubling scatter, at its price (ffbh +
// five instructions after the length / validity / start triple); only n < 16 takes the radix-4 route (blocks of 4
// or of 1, two to four of them) under a branch that a wave full of long intervals never enters.  Five arrays at
// k <= 128 and six at k <= 256, where doubling alone needs seven and eight; the fold is r4_fold_store's: the
// doubling levels into the blocks of 16 in one LDS pass, 16 -> 4 -> 1 in registers.
// ------------------------------------------------------------------------------------------
template <typename Rows, int U, int T, typename OutT, int TOP>
__global__ __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(MEMO_HALO_WAVES, 8)))
void sweep_conservation_mixed_kernel(const SweepArgs A) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    static_assert(TOP == 0 || (!Rows::kAnnot16 && TOP == Rows::kTopShift), "the order rides in the word only in the 4-byte formats");
    const int LS = A.ls, HL = A.hl, W = A.w;
    const LevelPlan P = plan_of(A.nlev, A.lvmask);
    Tile t;
    if (!locate_tile_w(A, t, W)) return;
    uint4 V[U];
    uint2 N[U];
    Rows::template issue<T, U>(A, t, 0, V, N);
    const uint32_t sent = (uint32_t)(A.ncols - 1);
    plan_clear<T>(A, lds, TOP ? (sent << TOP) | ((1u << TOP) - 1u) : sent);

    const int km1 = A.km1;
    // slot of the blocks of 2^(31-f), f = clz(n) <= 27:  f - ftop (the range: see the level plan above);
    // LDS byte address of tile slot x there:  bias4 + f * 4 LS + 4 x
    const uint32_t ls4 = 4u * (uint32_t)LS;
    const uint32_t lds_base = (uint32_t)(size_t)(__attribute__((address_space(3))) uint32_t *)lds;
    const uint32_t bias4 = pin_vgpr((int)(lds_base + 4u * (uint32_t)HL - (uint32_t)A.ftop * ls4));
    const uint32_t top_bit = pin_vgpr((int)0x80000000u);
    const uint32_t key = pin_vgpr((int)Rows::tile_key(t.a));
    // (rows of fewer than 16 positions exist only where the plan has arrays for them)
    uint32_t *const cells4 = lds + P.s4() * LS + HL, *const cells1 = lds + P.s1() * LS + HL;
    auto scatter = [&](uint32_t w, uint32_t col) {
        if (MEMO_ROW_CMPX) {
            // The long intervals (n >= 16) as one branch-free block: v_cmpx narrows EXEC to "n > 0", then to "clz(n) < 28";
            // the doubling arithmetic and both ds_min run on those lanes; EXEC restored.  What is left for the
            // compiler's branch is the rare short interval (0 < n < 16).
            int n;
            uint32_t r0, r1, r2, d;
            MEMO_EXEC_ALL_ONES(A.status);
            if constexpr (!Rows::kW12)
                asm volatile(
                    "v_sub_u32_sdwa %3, %6, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t"
                    "v_sub_u16 %4, %5, %7\n\t"
                    "v_ffbh_u32 %0, %3\n\t"
                    "v_cmpx_lt_i32 vcc, 0, %3\n\t"
                    "v_cmpx_gt_u32 vcc, 28, %0\n\t"
                    "v_mad_u32_u24 %2, %0, %8, %9\n\t"
                    "v_lshl_add_u32 %2, %4, 2, %2\n\t"
                    "v_mad_i32_i24 %1, %3, -4, %2\n\t"
                    "v_ashrrev_i32 %0, %0, %10\n\t"
                    "v_lshl_add_u32 %2, %0, 2, %2\n\t"
                    "ds_min_u32 %1, %11\n\t"
                    "ds_min_u32 %2, %11\n\t"
                    "s_mov_b64 exec, -1"
                    : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(n), "=&v"(d)
                    : "v"(w), "s"(km1), "v"(key), "s"(ls4), "v"(bias4), "v"(top_bit), "v"(TOP ? w : col)
                    : "memory", "vcc");
            else
                asm volatile(
                    "v_sub_u32_sdwa %3, %6, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"
                    "v_sub_u32 %4, %5, %7\n\t"
                    "v_bfe_u32 %4, %4, 8, 12\n\t"
                    "v_ffbh_u32 %0, %3\n\t"
                    "v_cmpx_lt_i32 vcc, 0, %3\n\t"
                    "v_cmpx_gt_u32 vcc, 28, %0\n\t"
                    "v_mad_u32_u24 %2, %0, %8, %9\n\t"
                    "v_lshl_add_u32 %2, %4, 2, %2\n\t"
                    "v_mad_i32_i24 %1, %3, -4, %2\n\t"
                    "v_ashrrev_i32 %0, %0, %10\n\t"
                    "v_lshl_add_u32 %2, %0, 2, %2\n\t"
                    "ds_min_u32 %1, %11\n\t"
                    "ds_min_u32 %2, %11\n\t"
                    "s_mov_b64 exec, -1"
                    : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(n), "=&v"(d)
                    : "v"(w), "s"(km1), "v"(key), "s"(ls4), "v"(bias4), "v"(top_bit), "v"(TOP ? w : col)
                    : "memory", "vcc");
            if ((uint32_t)(n - 1) < 15u) {  // 0 < n < 16: blocks of S = 4 (n >= 4) or 1 at start - n and start - S; two more while n >= 2S, 3S
                const uint32_t data = TOP ? w : col;
                const bool four = n >= 4;
                uint32_t *lv = (four ? cells4 : cells1) + d;
                const int S = four ? 4 : 1, q = four ? n >> 2 : n;
                atomicMin(lv - n, data);
                atomicMin(lv - S, data);
                if (q >= 2) atomicMin(lv - n + S, data);
                if (q == 3) atomicMin(lv - n + 2 * S, data);
            }
            return;
        }
        const int n = km1 - Rows::len(w);  // length of [end - (k-1), start)
        if (n > 0) {
            const uint32_t data = TOP ? w : col;
            const uint32_t d = Rows::rel_start(w, key);  // start - a
            const int f = __builtin_clz((unsigned)n);
            if (f <= 27) {  // n >= 16: blocks [start - n, .. + 2^j) and [start - 2^j, start), j = floor(log2 n)
                uint32_t r0, r1, r2;
                asm volatile(
                    "v_mad_u32_u24 %2, %5, %6, %7\n\t"
                    "v_lshl_add_u32 %2, %4, 2, %2\n\t"
                    "v_mad_i32_i24 %1, %3, -4, %2\n\t"
                    "v_ashrrev_i32 %0, %5, %8\n\t"
                    "v_lshl_add_u32 %2, %0, 2, %2\n\t"
                    "ds_min_u32 %1, %9\n\t"
                    "ds_min_u32 %2, %9"
                    : "=&v"(r0), "=&v"(r1), "=&v"(r2)
                    : "v"(n), "v"(d), "v"(f), "s"(ls4), "v"(bias4), "v"(top_bit), "v"(data)
                    : "memory");
            } else {  // n < 16: blocks of S = 4 (n >= 4) or 1 at start - n and start - S; two more while n >= 2S, 3S
                const bool four = n >= 4;
                uint32_t *lv = (four ? cells4 : cells1) + d;
                const int S = four ? 4 : 1, q = four ? n >> 2 : n;
                atomicMin(lv - n, data);
                atomicMin(lv - S, data);
                if (q >= 2) atomicMin(lv - n + S, data);
                if (q == 3) atomicMin(lv - n + 2 * S, data);
            }
        }
    };
    Rows::template consume<T, U>(A, t, 0, V, N, scatter);
    for (uint32_t b = 1, nb = Rows::template batches<T, U>(t); b < nb; ++b) {  // a dense tile: the rest
        Rows::template issue<T, U>(A, t, b, V, N);
        Rows::template consume<T, U>(A, t, b, V, N, scatter);
    }
    lds_barrier();  // waits for lgkmcnt(0): the ds_min above are invisible to the compiler
    __syncthreads();
    plan_fold_store<OutT, T, TOP>(A, t, lds);
}

// k <= 1: no row can write (casted_end >= start always), every position keeps its initial value
template <typename OutT>
__global__ void fill_conservation_kernel(OutT *out, int64_t n, OutT v) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        out[i] = v;
}

// One workgroup per row with end < start: its interval [clip(e-qs-(k-1)), clip(s-qs)) can be any
// length, so it is applied straight to the result in HBM, after the sweep, with atomics (rows may
// overlap each other).  filter_pq keeps such a row iff qs < start < qe + k (memo_query.py:25-27).
// The result has no 8- or 16-bit atomics, so a value is min-ed by a CAS on its 32-bit word; the word
// that holds the last L % (4 / sizeof(OutT)) values would reach past the caller's buffer, so those
// (at most 3) positions are left to long_rows_tail_kernel.
template <typename OutT>
__device__ __forceinline__ bool long_row_interval(int64_t s, int64_t e, int64_t o, int64_t qs, int64_t qe, int km1,
                                                  int ncols, int *status, int64_t &c, int64_t &hi, int64_t &cc,
                                                  int64_t fqs, int64_t fqe) {
    // [fqs, fqe) = the window the reference's filter sees (= [qs, qe) unless this sweep is a sub-window of it)
    if (!(s > fqs && s < fqe + km1 + 1)) return false;
    const int64_t L = qe - qs;
    hi = s - qs > L ? L : s - qs;
    c = e - qs - km1;
    c = c < 0 ? 0 : c;
    if (c >= hi) return false;
    cc = o < 0 ? o + ncols : o;
    if ((uint64_t)cc >= (uint64_t)ncols) {
        if (threadIdx.x == 0) atomicOr(status, kStatusBadAnnot);
        return false;
    }
    return true;
}

template <typename OutT>
__global__ void long_rows_conservation_kernel(const int64_t *ls, const int64_t *le, const int64_t *lo,
                                              int64_t qs, int64_t qe, int km1, int ncols, OutT *out,
                                              int *status, int64_t fqs, int64_t fqe) {
    int64_t c, hi, cc;
    if (!long_row_interval<OutT>(ls[blockIdx.x], le[blockIdx.x], lo[blockIdx.x], qs, qe, km1, ncols, status, c, hi, cc,
                                 fqs, fqe))
        return;
    constexpr int PER = 4 / (int)sizeof(OutT);  // results per 32-bit word
    const int64_t whole = (qe - qs) / PER * PER;  // positions whose word lies inside the buffer
    if (hi > whole) hi = whole;
    uint32_t *words = reinterpret_cast<uint32_t *>(out);
    for (int64_t p = c + threadIdx.x; p < hi; p += blockDim.x) {
        uint32_t *wp = words + p / PER;
        const int sh = (int)(p % PER) * 8 * (int)sizeof(OutT);
        const uint32_t field = (sizeof(OutT) == 2 ? 0xFFFFu : 0xFFu) << sh;
        uint32_t old = *wp;
        while (((old & field) >> sh) > (uint32_t)cc) {  // out[p] = min(out[p], col), on the field only
            const uint32_t seen = atomicCAS(wp, old, (old & ~field) | ((uint32_t)cc << sh));
            if (seen == old) break;
            old = seen;
        }
    }
}

// the last L % PER positions: one workgroup reduces, per position, the smallest column over ALL rows
// with end < start and stores it with a plain element store (nothing else writes there any more)
template <typename OutT>
__global__ __launch_bounds__(256) void long_rows_tail_kernel(const int64_t *ls, const int64_t *le, const int64_t *lo,
                                                             uint64_t n_long, int64_t qs, int64_t qe, int km1,
                                                             int ncols, OutT *out, int *status, int64_t fqs,
                                                             int64_t fqe) {
    constexpr int PER = 4 / (int)sizeof(OutT);
    const int64_t L = qe - qs, whole = L / PER * PER;
    __shared__ uint32_t best[4];
    if (threadIdx.x < 4) best[threadIdx.x] = 0xFFFFFFFFu;
    __syncthreads();
    for (uint64_t r = threadIdx.x; r < n_long; r += 256) {
        int64_t c, hi, cc;
        // (the bad-annot flag is raised by thread 0 of the main kernel's workgroup for this row)
        const int64_t s = ls[r], e = le[r], o = lo[r];
        if (!(s > fqs && s < fqe + km1 + 1)) continue;
        hi = s - qs > L ? L : s - qs;
        c = e - qs - km1;
        c = c < 0 ? 0 : c;
        cc = o < 0 ? o + ncols : o;
        if (c >= hi || (uint64_t)cc >= (uint64_t)ncols) continue;
        for (int64_t p = whole > c ? whole : c; p < hi; ++p) atomicMin(&best[p - whole], (uint32_t)cc);
    }
    __syncthreads();
    if (threadIdx.x < L - whole && best[threadIdx.x] < (uint32_t)out[whole + threadIdx.x])
        out[whole + threadIdx.x] = (OutT)best[threadIdx.x];
}

template <typename Rows, typename OutT>
SweepKernel cons_kernel(int w, int waves) {
#define MEMO_CASE(WW)                                                                         \
    case WW:                                                                                  \
        return waves == 4 ? (SweepKernel)sweep_conservation_kernel<Rows, WW, Rows::kLoads, 256, OutT>   \
                          : (SweepKernel)sweep_conservation_kernel<Rows, WW, Rows::kLoads, 64, OutT>;
    switch (w) {
        MEMO_CASE(256)
        MEMO_CASE(512)
        MEMO_CASE(1024)
        MEMO_CASE(2048)
        MEMO_CASE(4096)
    }
#undef MEMO_CASE
    return nullptr;
}

}  // namespace

#ifndef MEMO_HALO_LOADS
#define MEMO_HALO_LOADS 6
#endif
constexpr int kHaloLoads = MEMO_HALO_LOADS;  // 16-byte loads in flight per lane

template <typename Rows, typename OutT, int TOP>
static SweepKernel halo_kernel(int waves) {
    return waves == 8   ? (SweepKernel)sweep_conservation_halo_kernel<Rows, kHaloLoads, 512, OutT, TOP>
           : waves == 4 ? (SweepKernel)sweep_conservation_halo_kernel<Rows, kHaloLoads, 256, OutT, TOP>
                        : (SweepKernel)sweep_conservation_halo_kernel<Rows, kHaloLoads, 64, OutT, TOP>;
}

template <typename Rows, typename OutT, int TOP>
static SweepKernel r4_kernel(int waves) {
    return waves == 8   ? (SweepKernel)sweep_conservation_r4_kernel<Rows, kHaloLoads, 512, OutT, TOP>
           : waves == 4 ? (SweepKernel)sweep_conservation_r4_kernel<Rows, kHaloLoads, 256, OutT, TOP>
                        : (SweepKernel)sweep_conservation_r4_kernel<Rows, kHaloLoads, 64, OutT, TOP>;
}

template <typename Rows, typename OutT, int TOP>
static SweepKernel mixed_kernel(int waves) {
    return waves == 8   ? (SweepKernel)sweep_conservation_mixed_kernel<Rows, kHaloLoads, 512, OutT, TOP>
           : waves == 4 ? (SweepKernel)sweep_conservation_mixed_kernel<Rows, kHaloLoads, 256, OutT, TOP>
                        : (SweepKernel)sweep_conservation_mixed_kernel<Rows, kHaloLoads, 64, OutT, TOP>;
}

template <typename OutT>
static SweepKernel halo3_kernel(int waves, bool annot9) {
    if constexpr (sizeof(OutT) == 2) {
        if (annot9)  // (256 .. 511 genomes)
            return waves == 8   ? (SweepKernel)sweep_conservation_halo3_kernel<PackedRows3::kLoads, 512, OutT, true>
                   : waves == 4 ? (SweepKernel)sweep_conservation_halo3_kernel<PackedRows3::kLoads, 256, OutT, true>
                                : (SweepKernel)sweep_conservation_halo3_kernel<PackedRows3::kLoads, 64, OutT, true>;
    }
    return waves == 8   ? (SweepKernel)sweep_conservation_halo3_kernel<PackedRows3::kLoads, 512, OutT>
           : waves == 4 ? (SweepKernel)sweep_conservation_halo3_kernel<PackedRows3::kLoads, 256, OutT>
                        : (SweepKernel)sweep_conservation_halo3_kernel<PackedRows3::kLoads, 64, OutT>;
}

template <typename OutT>
static int long_rows_conservation(const memo_index *ix, int64_t qs, int64_t qe, int32_t k, int ncols,
                                  OutT *d_out, hipStream_t st) {
    if (!ix->n_long || g_prepare_only) return MEMO_OK;
    if (int prc = refuse_plan_pointer(d_out)) return prc;
    const int64_t fqs = ix->whole_set ? ix->whole_qs : qs, fqe = ix->whole_set ? ix->whole_qe : qe;
    hipLaunchKernelGGL((long_rows_conservation_kernel<OutT>), dim3((unsigned)ix->n_long), dim3(256), 0, st,
                       ix->ls, ix->le, ix->lo, qs, qe, k - 1, ncols, d_out, ix->d_status, fqs, fqe);
    if ((qe - qs) % (4 / (int)sizeof(OutT)))
        hipLaunchKernelGGL((long_rows_tail_kernel<OutT>), dim3(1), dim3(256), 0, st, ix->ls, ix->le, ix->lo,
                           (uint64_t)ix->n_long, qs, qe, k - 1, ncols, d_out, ix->d_status, fqs, fqe);
    HIP_TRY(hipGetLastError());
    return MEMO_OK;
}

// Which level arrays for the unclipped sweep at k - 1 >= 64: 2 = doubling, 3 = radix-4, 4 = mixed.
// What separates them is LDS atomics per row against level arrays per tile (profiles/r02_mixed_levels.txt, config 3):
// every row costs two blocks on doubling and mixed arrays (mixed: when its interval has 16 positions or more), two
// to four on radix-4 arrays -- 0.39 ms at two blocks per row, +0.14 ms per extra block -- and every array beyond
// four costs clear, fold and tile length: five arrays +0.03 ms, six +0.1, doubling's seven or eight +0.17 / +0.25.
// The interval lengths n = k - 1 - overlap are known in distribution from the overlaps sampled when the packed rows
// were made (memo_index.len_hist), so: many intervals under 16 positions (mixed would keep entering its slow path)
// -> doubling up to seven levels (k <= 128; its fold runs in registers) and on dense indexes, else radix-4; radix-4
// within 0.15 blocks per row of two (0.9 where mixed needs six arrays) -> radix-4; else mixed.
// The level plan of the mixed arrays for k - 1 = km1 (see plan_of above): returns D, the slots of the doubling range; *ftop =
// clz of the range's largest block size; *mask = populated slots | kPlanSep16 | kPlanHas4 | kPlanHas1.  `all`: every array a
// k - 1 of this size can need (rounds 2-3; also what an index without an exact census gets).
static int level_plan(const memo_index *ix, int km1, bool all, int *ftop, uint32_t *mask) {
    all = all || !ix->len_seen_exact;
    bool lv[32] = {false};
    bool has4 = false, has1 = false;
    for (int len = 0; len < km1 && len < 256; ++len) {
        if (!all && !((ix->len_seen[len >> 5] >> (len & 31)) & 1u)) continue;  // no row of the index has this overlap
        const int n = km1 - len;
        if (n >= 16) lv[__builtin_clz((unsigned)n)] = true;
        else if (n >= 4) has4 = true;
        else has1 = true;
    }
    int top = 32, bot = -1;
    for (int f = 0; f <= 27; ++f)
        if (lv[f]) {
            top = f < top ? f : top;
            bot = f;
        }
    if (bot < 0) top = bot = 27, lv[27] = true;  // (no row of 16 positions or more: the blocks of 16 alone, which nobody writes)
    uint32_t m = 0;
    for (int f = top; f <= bot; ++f)
        if (lv[f]) m |= 1u << (f - top);
    if (bot < 27) m |= kPlanSep16;
    if (has4) m |= kPlanHas4;
    if (has1) m |= kPlanHas1;
    *ftop = top;
    *mask = m;
    return bot - top + 1;
}

static int pick_levels(const memo_index *ix, int k, bool moderate) {
    const int km1 = k - 1, fallback = (km1 < 128 || !moderate) ? 2 : 3;
    if (!ix->len_hist_rows) return fallback;
    double rows = 0, blocks4 = 0, small = 0;
    for (int len = 0; len < 256 && len < km1; ++len) {
        const double w = ix->len_hist[len];
        const int n = km1 - len;
        const int i = floor_log2((uint32_t)n) >> 1, q = n >> (2 * i);  // radix-4: blocks of 4^i, leading digit q
        rows += w;
        blocks4 += w * (q == 1 ? 2 : q + 1);
        if (n < 16) small += w;
    }
    if (rows <= 0) return fallback;
    // What decides (profiles/r04_large_k.txt): level arrays per tile against blocks per row -- and on an index of moderate density
    // the arrays weigh more (rows built from sequences, 2 x 20 Mbp x 50 genomes, one more array costs what 0.3 blocks per row
    // cost).  The mixed arrays take two blocks per row and exist only where some row can write (level_plan): with no more of them
    // than radix-4's four they are the choice (BASELINE's generator at every k >= 65: two or three arrays).  Where the overlaps
    // reach every level (rows built from sequences: five or six mixed arrays, seven or eight doubling ones) radix-4 arrays win
    // while a row takes few blocks -- k = 101: 2.3 blocks per row, 0.185 ms against 0.204 doubling and 0.208 mixed; k = 128: 0.198
    // against 0.228; k = 160: 0.248 against 0.272 mixed -- and lose beyond ~2.9 (k = 65: 0.181 against 0.168 doubling; k = 200:
    // 0.284 against 0.279 mixed).  Many intervals under 16 positions keep the mixed arrays out (their slow path).
    const bool few_small = small <= 0.03 * rows;
    int ftop = 0;
    uint32_t mask = 0;
    const int planned = ix->len_seen_exact ? plan_of(level_plan(ix, km1, false, &ftop, &mask), mask).arrays() : 99;
    if (few_small && planned <= 4) return 4;
    if (moderate && blocks4 <= 2.9 * rows) return 3;
    if (few_small) return ix->len_seen_exact || blocks4 > (km1 >= 128 && moderate ? 2.9 : 2.15) * rows ? 4 : 3;
    return fallback;
}

#ifndef MEMO_TABLE_DEFAULT
#define MEMO_TABLE_DEFAULT 1  // the dense rows are swept by the table-driven kernel wherever the query fits it (0: A/B builds)
#endif

template <typename OutT>
static int query_conservation(memo_index_t *ix, int64_t qs, int64_t qe, int32_t k, int32_t num_docs,
                              OutT *d_out, void *stream) {
    int rc = check_query_args(ix, qs, qe, k, num_docs, d_out);
    if (rc) return rc;
    if (sizeof(OutT) == 1 && num_docs > 255)
        return fail(MEMO_EINVAL, "uint8 results need num_docs <= 255, got %d", num_docs);
    if (qe <= qs) return MEMO_OK;
    DeviceGuard guard(ix->device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (k <= 1 || ix->rows == 0) {
        if (g_prepare_only) return MEMO_OK;
        if ((rc = refuse_plan_pointer(d_out))) return rc;
        hipLaunchKernelGGL((fill_conservation_kernel<OutT>), dim3(2048), dim3(256), 0, st, d_out,
                           qe - qs, (OutT)num_docs);
        HIP_TRY(hipGetLastError());
        return long_rows_conservation<OutT>(ix, qs, qe, k, num_docs + 1, d_out, st);
    }
    SweepArgs A;
    fill_args(ix, A, qs, qe, k, d_out);
    A.ncols = num_docs + 1;
    A.nlev = floor_log2((uint32_t)(k - 1)) + 1;
    A.nwords = 0;
    int fmt;
    if ((rc = pick_rows(ix, k, fmt))) return rc;
    hipStream_t st_early = static_cast<hipStream_t>(stream);
    // the 4-byte words a sweep reads: the k-class view of them where one exists or is due (packed_rows_for, memo_view.hip)
    auto use_words = [&]() -> int {
        ix->last_rows_read = ix->rows;  // (6-byte rows and the int64 columns have no views)
        if (fmt != 4 && fmt != 12) return MEMO_OK;
        uint32_t *vpk = nullptr;
        int64_t *vboff = nullptr;
        uint64_t vrows = 0;
        const int vrc = packed_rows_for(ix, k - 1, qe - qs, false, st_early, &vpk, &vboff, &vrows);
        if (vrc) return vrc;
        A.pk = vpk;
        A.boff = vboff;
        ix->last_rows_read = vrows;
        return MEMO_OK;
    };
    // Tile shape, from interleaved A/B on one device (profiles/r01_ab_*.txt).
    //  int64 rows (HBM-bound): four waves share a 4096-position tile -- fewest k-1 row halos per
    //    position; 1-3 % over one wave per 1024 positions at k <= 32, 10 % at k = 101.
    //  packed rows (4-6x fewer bytes; LDS-atomic / issue-bound): waves per CU matter, but so does
    //    the k-1 halo: 1024 positions x 4 waves wins at k = 31 (20 KiB, 8 workgroups per CU) and at
    //    k = 101 (28 KiB) over 512 or 2048 positions.
    // Short windows want many small tiles either way.
    const memo_tuning &tune = ix->tune;
    int w = tune.tile_w, waves = tune.waves == 1 || tune.waves == 4 ? tune.waves : 0;
    while (w & (w - 1)) w &= w - 1;  // (the clipped kernels and the doubling arrays come in powers of two)
    if (w > 4096) w = 4096;
    // int64 rows on a sparse index (< 2 rows per position: profiles/r01_sparse_index_tiles.txt) are
    // no longer HBM-bound per tile; they want the packed rows' shape (more workgroups per CU)
    const double span = (double)(ix->max_s - ix->min_s) + 1.0;
    const bool sparse = (double)ix->rows < 2.0 * span;
    if (!w) {
        const size_t budget = (fmt || sparse) ? 32 * 1024 : 80 * 1024;
        w = 4096;
        while ((size_t)A.nlev * w * 4 > budget && w > 256) w >>= 1;
        while (w > 1024 && (qe - qs) / w < 8192) w >>= 1;                 // big 4-wave tiles: a few thousand suffice
        while (w > 256 && w <= 1024 && (qe - qs) / w < 32768) w >>= 1;   // short windows: many small tiles
    }
    if (!waves) waves = w >= 1024 ? 4 : 1;  // short windows end up with small tiles: one wave each
    const bool checked = ix->max_annot >= (uint64_t)A.ncols;  // some row could be outside the matrix
    // unclipped scatter (PackedRows, every annot inside the matrix): w is the size of a level array,
    // HL + tile + HR cells (start - a <= tile + k + 30 inside a slice); the tile is what is left
    // after the halo, rounded down to whole buckets
    // (below one row per position the halo's extra clear and fold cost more than the scatter saves)
    bool halo = fmt && !checked && (tune.scatter >= 2 || (tune.scatter == 0 && (double)ix->rows >= span));
    // Radix-4 levels (sweep_conservation_r4_kernel) from k = 65 up on indexes of moderate density: four arrays
    // where doubling needs seven or eight.  Interleaved A/B (profiles/r02_radix4_levels.txt; doubling -> radix-4,
    // ms): config 3 (5 rows per position) k = 101 0.564 -> 0.514, k = 256 0.951 -> 0.668, with arrays of 2560
    // cells shared by eight waves (40 KiB, four workgroups per CU).  At k = 21 / 31 / 64 the doubling arrays stay
    // ahead (0.375 vs 0.40, 0.46 vs 0.50), and so they do at any k on a dense index -- config 5, 25 rows per
    // position: k = 101 0.89 vs 1.08, k = 256 1.14 vs 1.55 -- where a row's third and fourth block cost more LDS
    // atomics than the fold steps save.
    const bool moderate = (double)ix->rows < 12.0 * span;
    // (debug switch scatter = 5: the mixed arrays with the library's level plan, whatever pick_levels would say)
    const int levels = !halo || fmt == 3 ? 0 : tune.scatter == 5 ? 4 : tune.scatter >= 2 ? tune.scatter : (k - 1 >= 64 ? pick_levels(ix, k, moderate) : 2);
    if (levels == 3) {
        const int bw = 1 << ix->bshift;
        const int hl = (k - 1 + 3) & ~3, hr = (k - 1 + bw - 1 + 3) & ~3;
        const int m = (floor_log2((uint32_t)(k - 1)) >> 1) + 1;
        int ls = tune.tile_w ? tune.tile_w : 2560;
        if (ls > 8192) ls = 8192;
        if (fmt == 12 && ls > 4096) ls = 4096;  // (12-bit start field: start - a < array size <= 2^12)
        // short windows: enough tiles to fill the chip (a few thousand of them)
        while (!tune.tile_w && ls > 640 && (qe - qs) / (ls - hl - hr > bw ? ls - hl - hr : bw) < 4096) ls = (ls / 2) & ~3;
        const int tw = (ls - hl - hr) / bw * bw;
        if (tw >= bw && 2 * tw >= hl + hr && (size_t)m * (hl + tw + hr) * 4 <= 160 * 1024) {
            A.nlev = m;
            A.hl = hl;
            A.w = tw;
            A.ls = hl + tw + hr;
            waves = tune.waves == 1 || tune.waves == 4 || tune.waves == 8 ? tune.waves : (ls >= 2048 ? 8 : 4);
            // the order rides in the row word when the sentinel num_docs fits its field (8 / 12 bits)
            SweepKernel kern = fmt == 4    ? (num_docs <= 255 ? r4_kernel<PackedRows<false, false>, OutT, 24>(waves)
                                                              : r4_kernel<PackedRows<false, false>, OutT, 0>(waves))
                               : fmt == 12 ? (num_docs <= 4095 ? r4_kernel<PackedRows<false, false, true>, OutT, 20>(waves)
                                                               : r4_kernel<PackedRows<false, false, true>, OutT, 0>(waves))
                                           : r4_kernel<PackedRows<true, false>, OutT, 0>(waves);
            if ((rc = use_words())) return rc;
            if ((rc = launch_tiles(kern, A, tw, 64 * waves, (size_t)m * A.ls * 4, st))) return rc;
            ix->last_sweep = 3;
            return long_rows_conservation<OutT>(ix, qs, qe, k, A.ncols, d_out, st);
        }
    }
    // Mixed levels (sweep_conservation_mixed_kernel): 1, 4, 16, then doubling -- of which only the arrays some row of the index
    // can write to exist (level_plan; the debug switch scatter = 4 asks for all of them, the layout of rounds 2-3).  k - 1 >= 16.
    if (levels == 4 && k - 1 >= 16) {
        const int bw = 1 << ix->bshift;
        const int hl = (k - 1 + 3) & ~3, hr = (k - 1 + bw - 1 + 3) & ~3;
        int ftop = 0;
        uint32_t mask = 0;
        const int D = level_plan(ix, k - 1, tune.scatter == 4, &ftop, &mask);
        const int m = plan_of(D, mask).arrays();
        int ls = tune.tile_w ? tune.tile_w : (40 * 1024 / (4 * m)) & ~63;
        if (ls > 8192) ls = 8192;
        if (fmt == 12 && ls > 4096) ls = 4096;  // (12-bit start field: start - a < array size <= 2^12)
        while (!tune.tile_w && ls > 640 && (qe - qs) / (ls - hl - hr > bw ? ls - hl - hr : bw) < 4096) ls = (ls / 2) & ~3;
        const int tw = (ls - hl - hr) / bw * bw;
        if (tw >= bw && 2 * tw >= hl + hr && (size_t)m * (hl + tw + hr) * 4 <= 160 * 1024) {
            A.nlev = D;
            A.lvmask = mask;
            A.ftop = ftop;
            A.hl = hl;
            A.w = tw;
            A.ls = hl + tw + hr;
            waves = tune.waves == 1 || tune.waves == 4 || tune.waves == 8 ? tune.waves : (ls >= 1536 ? 8 : 4);
            SweepKernel kern = fmt == 4    ? (num_docs <= 255 ? mixed_kernel<PackedRows<false, false>, OutT, 24>(waves)
                                                              : mixed_kernel<PackedRows<false, false>, OutT, 0>(waves))
                               : fmt == 12 ? (num_docs <= 4095 ? mixed_kernel<PackedRows<false, false, true>, OutT, 20>(waves)
                                                               : mixed_kernel<PackedRows<false, false, true>, OutT, 0>(waves))
                                           : mixed_kernel<PackedRows<true, false>, OutT, 0>(waves);
            if ((rc = use_words())) return rc;
            if ((rc = launch_tiles(kern, A, tw, 64 * waves, (size_t)m * A.ls * 4, st))) return rc;
            ix->last_sweep = 4;
            ix->last_arrays = m;
            return long_rows_conservation<OutT>(ix, qs, qe, k, A.ncols, d_out, st);
        }
    }
    if (halo) {
        // A/B (profiles/r01_unclipped_scatter.txt): arrays of 1024 cells x 4 waves win at every window
        // length from 10^6 positions up and at k = 21 .. 101
        const int bw = 1 << ix->bshift;  // a slice ends at a bucket boundary: start - a <= tile + k - 1 + bw - 2
        const int hl = (k - 1 + 3) & ~3, hr = (k - 1 + bw - 1 + 3) & ~3;
        if (!tune.tile_w) w = k - 1 >= 128 ? 2048 : 1024;  // (k = 256 on config 5: 1.40 ms with 1024 cells x 4 waves, 1.14 with 2048 x 8)
        int tw = 0;
        for (;; w <<= 1) {
            tw = (w - hl - hr) / bw * bw;
            if ((tw >= bw && 2 * tw >= hl + hr) || w >= 4096) break;  // keep the halo under two thirds of the array
        }
        while ((size_t)A.nlev * (hl + tw + hr) * 4 > 160 * 1024 && tw > bw) tw = (tw / 2 + bw - 1) / bw * bw;
        if (tw < bw) halo = false;
        else {
            A.hl = hl;
            A.w = tw;
            A.ls = hl + tw + hr;
            if (tune.waves == 0) waves = w >= 2048 ? 8 : (w >= 1024 ? 4 : 1);
            if (tune.waves == 8) waves = 8;
            // The dense rows where they are resident and can answer (k - 1 <= 63, level arrays within 2^10 cells,
            // num_docs <= 255): a fifth fewer bytes for three more instructions per row.  Back to back -- thousands
            // of launches, the device settled at its power cap (1345-1375 W of 1400) -- they are 13 % faster at
            // k = 21 / 31 and 4-7 % at k = 48 / 64 (config 3: 0.324 against 0.374 ms on one device;
            // profiles/r02_dense_rows_ab.txt).  Short interleaved timings had shown them level: a switch of kernels
            // sets off a swing of the clocks that lasts some thirty launches, and the dense kernel, which draws more
            // power per unit of time, sits on the cap at a lower clock (2.2 against 2.36 GHz).
            const bool top8 = num_docs <= 255;
            // (the dense rows may leave out the rows that can never write at k <= 64 -- memo_common.h: boff3 -- and then have
            // their own row numbers and bucket table; they answer only while they still hold a row per position)
            const uint64_t drows = ix->boff3 ? ix->rows3 : ix->rows;
            // 256 .. 511 genomes: the table-driven kernel's nine-bit form (memo_sweep_cons3t.hip: A9), uint16 results, or not the dense rows
            const bool top9 = !top8 && num_docs <= 511 && ix->max_annot <= 511 && sizeof(OutT) == 2;
            const bool three = ix->p3 && (!ix->pk || !tune.force_packed) && k - 1 <= 63 && A.ls <= 1024 && (top8 || top9) &&
                         ((double)drows >= span || !ix->pk);
            int view_cap = 0;  // (a view whose cap is k - 1 holds exactly the rows that write at this k: the table-driven kernel's row blocks drop their test)
            int rpg = 5;       // rows per 16-byte group of the source handed out: 5, or 6 (a view whose groups carry their bucket: memo_view.hip)
            ix->last_variant = 0;
            // the dense rows of this k's class (a view that leaves out the rows that cannot write at this k), or all of them.  Views of
            // six rows per group are for the table-driven kernel alone: a query it cannot take (a negative window start, no room for
            // the tile table) asks again for five-row groups.
            const bool table = three && (tune.persistent == 5 || (tune.persistent == 0 && MEMO_TABLE_DEFAULT));
            for (int attempt = 0; three && attempt < 2; ++attempt) {
                const bool can_six = attempt == 0 && table && top8 && A.nlev <= 5 && qs >= 0;
                uint32_t *vp3 = nullptr;
                int64_t *vboff = nullptr;
                uint64_t vrows = 0;
                if ((rc = dense_rows_for(ix, k - 1, qe - qs, st, &vp3, &vboff, &vrows, &view_cap, can_six, &rpg, attempt == 0))) return rc;
                A.p3 = vp3;
                A.boff = vboff;
                ix->last_rows_read = vrows;
                if (table) {
                    // the tile's row slice from a table built once per (index, k): memo_sweep_cons3t.hip; 1 = does not fit
                    const int trc = launch_halo3t(ix, A, tw, (int)sizeof(OutT), st, top9, view_cap == k - 1 && !tune.no_all_write, rpg == 6);
                    if (trc < 0) return trc;
                    if (trc == MEMO_OK) {
                        ix->last_sweep = 5;
                        ix->last_variant = rpg == 6 ? 3 : 2;
                        return long_rows_conservation<OutT>(ix, qs, qe, k, A.ncols, d_out, st);
                    }
                }
                if (rpg != 6) break;  // (five-row groups: the kernel without a table takes them)
            }
            if (three && rpg == 6) return fail(MEMO_EHIP, "a six-row view reached a sweep that cannot read it");
            if (!three && fmt == 3) {
                halo = false;  // (below: the int64 columns, or an error when they are gone too)
            } else {
            SweepKernel kern = three      ? halo3_kernel<OutT>(waves, top9)  // (no tile table: no room, a negative window start)
                               : fmt == 4 ? (top8 ? halo_kernel<PackedRows<false, false>, OutT, 24>(waves)
                                                  : halo_kernel<PackedRows<false, false>, OutT, 0>(waves))
                               : fmt == 12 ? (num_docs <= 4095 ? halo_kernel<PackedRows<false, false, true>, OutT, 20>(waves)
                                                               : halo_kernel<PackedRows<false, false, true>, OutT, 0>(waves))
                                          : halo_kernel<PackedRows<true, false>, OutT, 0>(waves);
            if (!three && (rc = use_words())) return rc;
            if ((rc = launch_tiles(kern, A, tw, 64 * waves, (size_t)A.nlev * A.ls * 4, st))) return rc;
            ix->last_sweep = three ? 5 : 2;
            }
        }
    }
    if (!halo && fmt == 3) {  // the dense rows cannot answer this one (tile shape, num_docs > 255, sparse index)
        if (!ix->has_wide)
            return fail(MEMO_EINVAL, "this query needs the 4-byte rows or the int64 columns, which this index dropped");
        fmt = 0;
        w = tune.tile_w;  // tile shape for int64 rows
        while (w & (w - 1)) w &= w - 1;
        if (w > 4096) w = 4096;
        if (!w) {
            w = 4096;
            while ((size_t)A.nlev * w * 4 > (sparse ? 32u : 80u) * 1024 && w > 256) w >>= 1;
            while (w > 1024 && (qe - qs) / w < 8192) w >>= 1;
            while (w > 256 && w <= 1024 && (qe - qs) / w < 32768) w >>= 1;
        }
        waves = tune.waves == 1 || tune.waves == 4 ? tune.waves : (w >= 1024 ? 4 : 1);
    }
    if (!halo) {
        if (fmt == 12 && w > 2048) w = 2048;  // a 12-bit start field: the slice of a tile spans less than 2^12 positions
        while ((size_t)A.nlev * (w + kLevelSkew) * 4 > 160 * 1024 && w > 256) w >>= 1;
        A.hl = 0;
        A.w = w;
        A.ls = w + kLevelSkew;
        SweepKernel kern = fmt == 4   ? (checked ? cons_kernel<PackedRows<false, true>, OutT>(w, waves)
                                                 : cons_kernel<PackedRows<false, false>, OutT>(w, waves))
                           : fmt == 12 ? (checked ? cons_kernel<PackedRows<false, true, true>, OutT>(w, waves)
                                                  : cons_kernel<PackedRows<false, false, true>, OutT>(w, waves))
                           : fmt == 6 ? (checked ? cons_kernel<PackedRows<true, true>, OutT>(w, waves)
                                                 : cons_kernel<PackedRows<true, false>, OutT>(w, waves))
                                      : cons_kernel<WideRows, OutT>(w, waves);
        if (!kern) return fail(MEMO_EINVAL, "unsupported tile width %d", w);
        ix->last_rows_read = ix->rows;
        if ((rc = use_words())) return rc;
        if ((rc = launch_tiles(kern, A, w, 64 * waves, (size_t)A.nlev * A.ls * 4, st))) return rc;
        ix->last_sweep = 1;
    }
    return long_rows_conservation<OutT>(ix, qs, qe, k, A.ncols, d_out, st);
}

extern "C" {

int memo_query_conservation_dev(memo_index_t *ix, int64_t qs, int64_t qe, int32_t k,
                                int32_t num_docs, uint16_t *d_out, void *stream) {
    return query_conservation<uint16_t>(ix, qs, qe, k, num_docs, d_out, stream);
}

int memo_query_conservation_u8_dev(memo_index_t *ix, int64_t qs, int64_t qe, int32_t k,
                                   int32_t num_docs, uint8_t *d_out, void *stream) {
    return query_conservation<uint8_t>(ix, qs, qe, k, num_docs, d_out, stream);
}

}  // extern "C"
