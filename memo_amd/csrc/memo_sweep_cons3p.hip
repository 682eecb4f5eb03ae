// memo_sweep_cons3p.hip -- the conservation sweep on the dense rows as PERSISTENT workgroups whose rows arrive by
// LDS-DMA (round 3; the counterpart of /root/reference/src/memo_query.py:42-63 + :70 like every sweep here).
//
// What sweep_conservation_halo3_kernel (memo_sweep_cons.hip: one workgroup per tile) leaves on the table, by the
// counters of profiles/r02_sq_counters.txt: every wave spends ~110 scalar instructions finding its tile (64-bit tile
// arithmetic, two divisions by five, two dependent bucket-table loads) before its first row load can leave, the CU's
// one scalar unit is the busiest issue resource (71 %), and a tile has loads in flight for about two thirds of its
// life.  Here a workgroup walks a contiguous run of tiles:
//   * tile geometry is incremental and 32-bit (bucket index += W / 32, rows relative to the run's first group), the
//     two bucket-table entries of tile t+2 are fetched (scalar loads) while tile t is swept;
//   * the rows of tile t+1 are streamed by LDS-DMA (global_load_lds_dwordx4: no VGPRs held, 1 KiB per
//     wave-instruction) into the second of two 16 KiB stages while tile t scatters and folds; a counted vmcnt
//     leaves them in flight across the tile's barriers (raw s_barrier + lgkmcnt only);
//   * a lane takes its groups of five rows out of the stage with ds_read_b128 (conflict-free: consecutive lanes,
//     consecutive 16 bytes) and runs the same branch-free row block (v_cmpx ... ds_min x 2 ... s_mov exec);
//   * rows outside the tile's slice are masked by row number in the first and last piece only (an extra v_cmpx
//     pair); interior pieces carry no test;
//   * clear, fold and store are unrolled for the number of level arrays (template parameter).
// Every LDS access inside the tile loop is inline asm: a compiler-visible LDS read after an LDS-DMA makes hipcc wait
// vmcnt(0) (cdna_hip_programming.md, "Pipelining across barriers"), which would drain the prefetch.
// LDS per workgroup: NLEV x 4 KiB of level arrays + 2 x 16 KiB of stages (52 KiB at k = 31: three workgroups per CU).
#include "memo_sweep_dense.h"

using namespace memo;
using namespace memo::dense;

namespace {

// the J-th group of a lane (group tid + 256 J of the stage): read it, run its five rows; false = past the tile's groups
template <int J>
__device__ __forceinline__ bool stage_piece(uint32_t stage, int tid, int wave, uint32_t gbase, uint32_t gleft, const Geo &g,
                                            const RowConst &C, uint32_t span) {
    const uint32_t pg = (uint32_t)(J * 256 + wave * 64);  // first group of this wave's piece, in the stage
    if (pg >= gleft) return false;
    u32x4 Vn;
    asm volatile("ds_read_b128 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(Vn)
                 : "v"(stage + 16u * (uint32_t)tid), "n"(J * 256 * 16)
                 : "memory");
    const uint4 V = make_uint4(Vn.x, Vn.y, Vn.z, Vn.w);
    const uint32_t row0 = 5u * (gbase + pg);  // first row of the piece, counted from row 5 * g0
    if (row0 >= g.first && row0 + 320u <= g.end) {
        group_rows<false>(V, C, 0, 0);
    } else {
        const uint32_t tmp = 5u * (gbase + (uint32_t)(J * 256 + tid)) - g.first;
        group_rows<true>(V, C, tmp, span);
    }
    return true;
}

// MODE 0: rows by LDS-DMA into two stages (NLEV x 4 KiB + 32 KiB of LDS: three workgroups per CU);
// MODE 1: rows into registers at the head of a tile (NLEV x 4 KiB: eight workgroups per CU, like the tile-per-workgroup
//         kernel, minus its per-tile scalar work); MODE 2: the next tile's rows into the same registers as soon as the
//         scatter has used them up -- in flight under the barrier, the fold, the store and the clear.
template <int NLEV, typename OutT, int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MODE == 0 ? 3 : 8, 8)))
void sweep_conservation_halo3p_kernel(const SweepArgs A) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    constexpr int T = 256, NW = 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int HL = A.hl, W = A.w;
    const uint32_t lds_base = (uint32_t)(size_t)(__attribute__((address_space(3))) uint32_t *)lds;
    const uint32_t stage_base = lds_base + (uint32_t)NLEV * 4096u;

    // this workgroup's run of tiles: each XCD group (blockIdx % 8) gets a contiguous range of runs
    const int64_t blk = blockIdx.x;
    const int64_t run = (blk & 7) * A.tiles_per_xcd + (blk >> 3);  // (tiles_per_xcd: runs per XCD group here)
    int64_t tile = run * A.tiles_per_wg;
    if (tile >= A.ntiles) return;
    const int ntile = (int)(A.ntiles - tile < A.tiles_per_wg ? A.ntiles - tile : A.tiles_per_wg);

    // 32-bit bucket arithmetic (the launcher guarantees it fits): bucket of the tile's first position, relative
    // to the table; += W / bucket width per tile
    const int bw_shift = A.bshift, last = (int)(A.nb - 1);
    const int wb = W >> bw_shift;
    int bi0 = (int)(((A.tile0 + tile * W) >> bw_shift) - A.bbase);
    const int reach_full = (W + A.km1 + (1 << bw_shift) - 1) >> bw_shift;
    const int reach_last = (A.x_hi_last + A.km1 + (1 << bw_shift) - 1) >> bw_shift;
    const int64_t last_tile = A.ntiles - 1;
    auto clampi = [&](int v) { return v < 0 ? 0 : (v > last ? last : v); };
    // The two bucket-table entries of a tile, by SCALAR loads issued from inline asm: left to the compiler they become
    // vector loads inside the tile loop (stores to the result may alias the table for all it knows) with an s_waitcnt
    // vmcnt(0) behind them -- which drains the rows in flight.  The values are valid after the next lgkmcnt(0):
    // load_slice_async() is called right in front of a barrier_lds() and its results are not touched before it.
    auto load_slice_async = [&](int b0, bool is_last, Slice &s) {
        const int64_t *p0 = A.boff + clampi(b0), *p1 = A.boff + clampi(b0 + (is_last ? reach_last : reach_full));
        asm volatile("s_load_dwordx2 %0, %2, 0x0\n\ts_load_dwordx2 %1, %3, 0x0" : "=&s"(s.r0), "=&s"(s.r1) : "s"(p0), "s"(p1) : "memory");
    };
    Slice s_cur;
    load_slice_async(bi0, tile == last_tile, s_cur);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(s_cur.r0), "+s"(s_cur.r1)::"memory");
    // the run's base group: everything after this is relative to it, in 32 bits
    const uint64_t G0 = (s_cur.r0 / 5) & ~(uint64_t)7;
    const uint64_t R5 = 5 * G0;
    const uint4 *src_base = reinterpret_cast<const uint4 *>(A.p3) + G0 + lane;
    bool huge = false;
    auto geometry = [&](const Slice &s) {
        Geo g;
        const uint64_t q0w = s.r0 - R5, q1w = s.r1 - R5;
        huge |= (q1w >> 32) != 0;
        const uint32_t q0 = (uint32_t)q0w, q1 = (uint32_t)q1w;
        g.g0 = (uint32_t)(((uint64_t)q0 * 0xCCCCCCCDull) >> 34) & ~7u;
        const uint32_t g1 = (uint32_t)(((uint64_t)(q1 + 4) * 0xCCCCCCCDull) >> 34);
        g.ng = g1 - g.g0;
        g.first = q0 - 5 * g.g0;
        g.end = q1 - 5 * g.g0;
        return g;
    };
    // LDS-DMA of (up to) one stage of a tile's groups, pieces of 64 groups dealt to the waves; returns the number
    // of DMA instructions THIS wave issued
    auto issue_dma = [&](const Geo &g, uint32_t chunk, uint32_t stage) {
        const uint32_t pieces_all = (g.ng + 63) >> 6;
        const uint32_t p_lo = chunk * (kStageGroups / 64);
        uint32_t p_hi = p_lo + kStageGroups / 64;
        p_hi = p_hi < pieces_all ? p_hi : pieces_all;
        int n = 0;
        for (uint32_t p = p_lo + (uint32_t)wave; p < p_hi; p += NW) {
            const uint4 *src = src_base + g.g0 + p * 64;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(uintptr_t)src,
                                             (__attribute__((address_space(3))) void *)(size_t)(stage + (p - p_lo) * 1024u), 16, 0, 0);
            ++n;
        }
        return n;
    };
    auto wait_dma = [&](int leave) {  // this wave's DMA older than its last `leave` instructions have landed
        switch (leave) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        }
    };

    const uint32_t sent = ((uint32_t)(A.ncols - 1) << 24) | 0x00FFFFFFu;
    RowConst C;
    C.km1 = A.km1;
    C.status = A.status;
    C.ls4 = 4u * kLS;
    C.bias4 = (uint32_t)pin_vgpr((int)(lds_base + 4u * (uint32_t)HL - (uint32_t)(32 - NLEV) * C.ls4));
    C.top_bit = (uint32_t)pin_vgpr((int)0x80000000u);
    OutT *out = static_cast<OutT *>(A.out);
    const int cells = HL + W;
    constexpr int ctx = NLEV <= 1 ? 0 : (NLEV <= 3 ? 1 : 1 << (NLEV - 3));
    constexpr int valid = 64 - ctx;

    // MODE 1 / 2: a lane's four groups of a batch of 1024 (group tid + 256 j), loaded whole waves at a time
    // (Four loads in a row, none under a branch: with conditional loads in a loop hipcc puts an s_waitcnt vmcnt(0) in
    // front of every one of them -- the registers may still be the target of the load before -- and the four loads
    // run one after the other.  A piece past the tile's groups loads ONE group for the whole wave instead: one request.)
    uint4 V[4];
    const uint4 *src0 = reinterpret_cast<const uint4 *>(A.p3) + G0;
    auto issue_regs = [&](const Geo &g, uint32_t batch) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t pg = batch * kStageGroups + (uint32_t)(j * T + wave * 64);
            V[j] = src0[g.g0 + (pg < g.ng ? pg + (uint32_t)lane : 0u)];
        }
    };

    Geo g_cur = geometry(s_cur);
    if constexpr (MODE == 0) issue_dma(g_cur, 0, stage_base);
    else issue_regs(g_cur, 0);
    Slice s_nxt = s_cur;
    if (ntile > 1) {
        load_slice_async(bi0 + wb, tile + 1 == last_tile, s_nxt);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(s_nxt.r0), "+s"(s_nxt.r1)::"memory");
    }
    clear_levels<NLEV>(lds_base, sent);
    int64_t a = A.tile0 + tile * W;
    int64_t ob = a - A.qs - HL;  // output index of cell 0 of the level arrays

    for (int t = 0; t < ntile; ++t, ++tile, a += W, ob += W, bi0 += wb) {
        const uint32_t stage = stage_base + (uint32_t)(t & 1) * kStageBytes;
        // tile t+1: its rows leave now (MODE 0: into the other stage, free since tile t-1's scatter), tile t+2's table entries too
        Geo g_nxt = g_cur;
        int n_nxt = 0;
        if (t + 1 < ntile) {
            g_nxt = geometry(s_nxt);
            if constexpr (MODE == 0) n_nxt = issue_dma(g_nxt, 0, stage_base + (uint32_t)((t + 1) & 1) * kStageBytes);
        }
        if constexpr (MODE == 0) wait_dma(n_nxt);
        if (t + 2 < ntile) load_slice_async(bi0 + 2 * wb, tile + 2 == last_tile, s_nxt);  // (valid behind the barrier)
        barrier_lds();  // every wave's pieces of tile t have landed (MODE 0); the level arrays are clear

        C.a10s = (uint32_t)pin_vgpr((int)(((uint32_t)a & 1023u) << 6));
        const uint32_t span = g_cur.end - g_cur.first;
        const uint32_t nchunks = (g_cur.ng + kStageGroups - 1) / kStageGroups;
        for (uint32_t chunk = 0; chunk == 0 || chunk < nchunks; ++chunk) {
            if (chunk) {  // a tile with more rows than a stage / a batch holds (rare): the rest, synchronously
                if constexpr (MODE == 0) {
                    barrier_lds();
                    issue_dma(g_cur, chunk, stage);
                    wait_dma(0);
                    barrier_lds();
                } else {
                    issue_regs(g_cur, chunk);
                }
            }
            const uint32_t gbase = chunk * kStageGroups;
            const uint32_t gleft = g_cur.ng > gbase ? g_cur.ng - gbase : 0;
            // a lane takes groups tid, tid + 256, ... of the stage / batch: wave w's j-th group is in piece 4 j + w
            static_assert(kStageGroups / T == 4, "four groups per lane and stage");
            if constexpr (MODE == 0) {
                (void)(stage_piece<0>(stage, tid, wave, gbase, gleft, g_cur, C, span) &&
                       stage_piece<1>(stage, tid, wave, gbase, gleft, g_cur, C, span) &&
                       stage_piece<2>(stage, tid, wave, gbase, gleft, g_cur, C, span) &&
                       stage_piece<3>(stage, tid, wave, gbase, gleft, g_cur, C, span));
            } else {
                (void)(reg_piece<0>(V[0], tid, wave, gbase, gleft, g_cur, C, span) &&
                       reg_piece<1>(V[1], tid, wave, gbase, gleft, g_cur, C, span) &&
                       reg_piece<2>(V[2], tid, wave, gbase, gleft, g_cur, C, span) &&
                       reg_piece<3>(V[3], tid, wave, gbase, gleft, g_cur, C, span));
            }
        }
        if constexpr (MODE == 2)
            if (t + 1 < ntile) issue_regs(g_nxt, 0);  // the registers are free again: tile t+1's rows fly under the rest of tile t
        barrier_lds();  // (lgkmcnt(0): the ds_min above are invisible to the compiler)

        // fold in registers + store (halo_fold_store_dpp of memo_sweep_cons.hip, unrolled for NLEV), then clear
        const int64_t o_lo = a - A.qs + (tile == 0 ? A.x_lo_first : 0);
        const int64_t o_hi = a - A.qs + (tile == last_tile ? A.x_hi_last : W);
        for (int base = wave * 4 * valid; base + 4 * ctx < cells; base += NW * 4 * valid) {
            const int x0 = base + 4 * lane;
            const int xr = x0 < kLS - 4 ? x0 : kLS - 4;  // (past the array: lanes whose results are dropped below)
            u32x4 L[6];
            read_levels<NLEV>(lds_base + 4u * (uint32_t)xr, L);
            auto lv = [&](int i) { return make_uint4(L[i].x, L[i].y, L[i].z, L[i].w); };
            uint4 M = lv(0);
            if constexpr (NLEV >= 6) fold_step_dpp<4>(M, lv(NLEV - 5), lane);
            if constexpr (NLEV >= 5) fold_step_dpp<3>(M, lv(NLEV - 4), lane);
            if constexpr (NLEV >= 4) fold_step_dpp<2>(M, lv(NLEV - 3), lane);
            if constexpr (NLEV >= 3) fold_step_dpp<1>(M, lv(NLEV - 2), lane);
            if constexpr (NLEV >= 2) fold_step_dpp<0>(M, lv(NLEV - 1), lane);
            if (lane < ctx || x0 >= cells) continue;
            const int64_t g = ob + x0;
            if (g >= o_lo && g + 4 <= o_hi) {
                if constexpr (sizeof(OutT) == 1) {
                    *reinterpret_cast<uint32_t *>(out + g) = __builtin_amdgcn_perm(M.y, M.x, 0x0c0c0703u) |
                                                             __builtin_amdgcn_perm(M.w, M.z, 0x07030c0cu);
                } else {
                    *reinterpret_cast<uint2 *>(out + g) = make_uint2(__builtin_amdgcn_perm(M.y, M.x, 0x0c070c03u),
                                                                     __builtin_amdgcn_perm(M.w, M.z, 0x0c070c03u));
                }
            } else {
                const uint32_t v[4] = {M.x >> 24, M.y >> 24, M.z >> 24, M.w >> 24};
                for (int i = 0; i < 4; ++i)
                    if (g + i >= o_lo && g + i < o_hi) out[g + i] = (OutT)v[i];
            }
        }
        if (t + 1 < ntile) {
            barrier_lds();  // every wave has read the level arrays
            if constexpr (MODE == 1) issue_regs(g_nxt, 0);
            clear_levels<NLEV>(lds_base, sent);
        }
        g_cur = g_nxt;
    }
    if (huge && tid == 0) atomicOr(A.status, kStatusHugeSlice);
}

template <typename OutT, int MODE>
SweepKernel kernel_for(int nlev) {
    switch (nlev) {
        case 1: return (SweepKernel)sweep_conservation_halo3p_kernel<1, OutT, MODE>;
        case 2: return (SweepKernel)sweep_conservation_halo3p_kernel<2, OutT, MODE>;
        case 3: return (SweepKernel)sweep_conservation_halo3p_kernel<3, OutT, MODE>;
        case 4: return (SweepKernel)sweep_conservation_halo3p_kernel<4, OutT, MODE>;
        case 5: return (SweepKernel)sweep_conservation_halo3p_kernel<5, OutT, MODE>;
        case 6: return (SweepKernel)sweep_conservation_halo3p_kernel<6, OutT, MODE>;
    }
    return nullptr;
}

template <typename OutT>
SweepKernel kernel_for(int nlev, int mode) {
    return mode == 0 ? kernel_for<OutT, 0>(nlev) : mode == 1 ? kernel_for<OutT, 1>(nlev) : kernel_for<OutT, 2>(nlev);
}

}  // namespace

namespace memo {

// Launch the persistent dense-row sweep if this query fits it (else return 1: the caller takes the one-workgroup-per-
// tile kernel).  A: filled for the unclipped sweep (hl, w, ls, nlev, ncols); tw = tile width.
static int launch_halo3p(SweepArgs &A, int tw, int elem_bytes, int device, int mode, hipStream_t st) {
    if (!A.p3 || A.ls > kLS || A.nlev < 1 || A.nlev > 6 || A.km1 > 63) return 1;
    int64_t q = A.qs / tw;
    if (A.qs % tw < 0) --q;
    const int64_t tile0 = q * tw;
    if (A.qs < 0 || ((tile0 - A.qs) & 3)) return 1;  // the register fold needs the tile grid on the output's 4-position raster
    const int64_t ntiles = ((A.qe - tile0) + tw - 1) / tw;
    const int bw = 1 << A.bshift;
    if (A.nb >= ((int64_t)1 << 31) || ((A.qe + A.km1 + 2 * bw) >> A.bshift) - A.bbase >= ((int64_t)1 << 31) ||
        (tile0 >> A.bshift) - A.bbase <= -((int64_t)1 << 31))
        return 1;  // (32-bit bucket arithmetic in the kernel)
    static thread_local int cus[64] = {0};
    if (device < 0 || device >= 64) return 1;
    if (!cus[device]) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) != hipSuccess) return 1;
        cus[device] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const size_t lds = (size_t)A.nlev * 4096 + (mode == 0 ? 2 * kStageBytes : 0);
    int per_cu = (int)((160 * 1024) / lds);
    per_cu = per_cu > 8 ? 8 : per_cu;  // (8 workgroups of 4 waves fill a CU)
    int64_t wgs = (int64_t)cus[device] * per_cu;
    wgs = (wgs + 7) / 8 * 8;
    if (ntiles < 4 * wgs) return 1;  // short windows: one workgroup per tile fills the chip better
    A.tile0 = tile0;
    A.ntiles = ntiles;
    A.tiles_per_wg = (ntiles + wgs - 1) / wgs;
    const int64_t runs = (ntiles + A.tiles_per_wg - 1) / A.tiles_per_wg;
    A.tiles_per_xcd = (runs + 7) / 8;  // runs per XCD group
    A.x_lo_first = (int)(A.qs - tile0);
    A.x_hi_last = (int)(A.qe - (tile0 + (ntiles - 1) * tw));
    SweepKernel kern = elem_bytes == 1 ? kernel_for<uint8_t>(A.nlev, mode) : kernel_for<uint16_t>(A.nlev, mode);
    if (!kern) return 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)(A.tiles_per_xcd * 8)), dim3(256), lds, st, A);
    if (hipGetLastError() != hipSuccess) return fail(MEMO_EHIP, "launch of the persistent dense-row sweep failed");
    return MEMO_OK;
}

namespace {
struct Register {
    Register() { g_persistent_launch = launch_halo3p; }
} register_persistent_launch;
}  // namespace

}  // namespace memo
