// memo_sweep_cons3t.hip -- the conservation sweep on the dense rows, one workgroup per tile, with the tile's row slice
// read from a TILE TABLE instead of being worked out by every wave (round 3; the counterpart of
// /root/reference/src/memo_query.py:42-63 + :70 like every sweep here).
//
// sweep_conservation_halo3_kernel (memo_sweep_cons.hip) spends ~300 scalar instructions per wave, ~130 of them finding
// its tile: 64-bit tile arithmetic, two divisions by five, two dependent bucket-table loads, bounds.  The CU has one
// scalar unit, and the sweep's time follows its scalar instruction count (+100 do-nothing SALU per wave: +6.6 %; +100
// VALU: +5.3 %; profiles/r03_issue_diagnostic.txt).  A tile's slice -- first group, groups, first and last valid row --
// depends on the index, the tile width and k only (tiles are aligned in PIVOT coordinates), so it is computed once per
// (index, k) by tile_table_kernel, 32 bytes per tile (3.4 MB for BASELINE config 3), kept with the index, and every
// later query's waves fetch their tile with ONE s_load_dwordx8.  The rest of the tile body is the lean one of
// memo_sweep_dense.h: clear, level reads and fold unrolled for the number of level arrays (template parameter), four
// row loads in a row, rows masked by number only in the pieces that straddle an end of the slice.
#include "memo_sweep_dense.h"

using namespace memo;
using namespace memo::dense;

namespace {

struct TileDesc {  // 32 bytes
    uint32_t g0_lo, g0_hi;  // first group of the slice (five-row groups: a multiple of 8)
    uint32_t ng;            // groups; 0xFFFFFFFF: more than 2^32 rows reach this tile (unsupported)
    uint32_t first, end;    // rows [first, end) of the slice, counted from row 5 * g0
    uint32_t pad[3];
};

// tile T covers pivot positions [T * w, (T + 1) * w); its rows: T * w <= start < roundup((T + 1) * w + k - 1, bucket)
// rpg: rows per group -- 5, or 6 (A/B: groups that carry their bucket)
__global__ void tile_table_kernel(const int64_t *boff, int64_t nb, int64_t bbase, int bshift, int w, int km1, int64_t ntab,
                                  TileDesc *out, int rpg) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= ntab) return;
    const int64_t a = t * w, last = nb - 1, lim = a + w + km1;
    int64_t b0 = (a >> bshift) - bbase, b1 = ((lim + ((int64_t)1 << bshift) - 1) >> bshift) - bbase;
    b0 = b0 < 0 ? 0 : (b0 > last ? last : b0);
    b1 = b1 < 0 ? 0 : (b1 > last ? last : b1);
    const uint64_t r0 = a <= 0 ? 0 : (uint64_t)boff[b0], r1 = (uint64_t)boff[b1];
    // (five-row groups: from the 128-byte line that holds the slice's first group -- a wave's 1 KiB load is eight whole lines;
    // six-row groups end at bucket boundaries: the slice is whole groups from its first one)
    const uint64_t g0 = rpg == 6 ? r0 / 6 : (r0 / 5) & ~(uint64_t)7, g1 = (r1 + (uint64_t)rpg - 1) / (uint64_t)rpg;
    TileDesc d;
    d.g0_lo = (uint32_t)g0;
    d.g0_hi = (uint32_t)(g0 >> 32);
    d.ng = r1 - (uint64_t)rpg * g0 >= 0xFFFF0000ull ? 0xFFFFFFFFu : (uint32_t)(g1 - g0);
    d.first = (uint32_t)(r0 - (uint64_t)rpg * g0);
    d.end = (uint32_t)(r1 - (uint64_t)rpg * g0);
    d.pad[0] = d.pad[1] = d.pad[2] = 0;
    out[t] = d;
}

// Diagnostic builds only (tools/build_variant.sh NAME -DMEMO_T_ABLATE=bits; results are wrong, they size a phase:
// profiles/r03_phase_ablation.txt): 1 = rows loaded and dropped, 2 = no clear of the level arrays, 4 = no fold (the finest
// level is stored as it is), 8 = no store, 16 = no row loads
#ifndef MEMO_T_ABLATE
#define MEMO_T_ABLATE 0
#endif

// T = 256: four waves per tile (eight tiles = 32 waves per CU), the only form instantiated (T = 128 lost: see the launcher)
// A9: an index of 256 .. 511 genomes -- the order in the top NINE bits of a level cell (memo_sweep_dense.h: MEMO_ROW9_AT), uint16 results
// AW: the row source is a k-class view whose cap is this k - 1 -- every row of it writes, the row blocks carry no test (memo_sweep_dense.h)
// SIX: the row source is a k-class view in groups of six rows that carry their bucket (memo_view.hip: view_build_kernel<6>; the
// library's choice of view where it applies: -2.3 % at k = 31 against five-row views, profiles/r05_view_pass.txt)
// SP: a row source of few rows per tile (under ~3/4 of the 1024 groups a batch of loads covers: the k-class views of config 3, every
// sequence-built index) -- a piece past the tile's slice issues NO load (a wave-uniform branch).  Rounds 3-5 issued all four loads of a
// lane unconditionally, a dead piece fetching one group for its 64 lanes: one request to memory, but a full trip through the
// vector memory pipeline and 1 KiB of registers written, nine to fourteen times per tile at k = 31 ... 9 -- and the sweep is
// sensitive there (four MORE loads per lane, a later tile's rows prefetched into L2, cost it 8-15 %).  Skipping them: k = 31 -3 %,
// k = 21 -3.4 %, k = 17 -4.5 %, k = 9 -6 % on config 3's views.  Not for tiles of several full batches (config 5: +2 ... +6 %:
// behind the branches the compiler waits for ALL of a batch's loads before its first piece), hence a template flag the
// launcher sets from the rows per tile (profiles/r06_headline.txt; masking the dead loads off with EXEC = 0 instead gained nothing).
template <int NLEV, typename OutT, int T, bool A9 = false, bool AW = false, bool SIX = false, bool SP = false>
__global__ __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(T == 256 ? 8 : 4, 8)))
void sweep_conservation_halo3t_kernel(const SweepArgs A) {
    static_assert(!A9 || sizeof(OutT) == 2, "more than 255 genomes: uint16 results");
    static_assert(!(A9 && SIX), "six-row groups hold eight-bit annots");
    constexpr int SH = A9 ? 23 : 24;  // a cell = order << SH | tie-breaking bits
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    constexpr int NW = T / 64, NL = kStageGroups / T;  // waves; 16-byte groups per lane and batch
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t blk = blockIdx.x;
    const uint32_t tile = (blk & 7u) * (uint32_t)A.tiles_per_xcd + (blk >> 3);  // each XCD group: a contiguous run of tiles
    if (tile >= (uint32_t)A.ntiles) return;
    const uint32_t tabs = (uint32_t)A.tile_abs0 + tile;  // the tile's number in pivot coordinates
    const TileDesc *dp = static_cast<const TileDesc *>(A.ttab) + (tabs < (uint32_t)A.ntab ? tabs : (uint32_t)A.ntab - 1u);
    const uint4 d0 = *reinterpret_cast<const uint4 *>(dp);
    const uint32_t d_end = dp->end;
    Geo g;
    g.g0 = 0;
    g.ng = d0.z;
    g.first = d0.w;
    g.end = d_end;
    if (g.ng == 0xFFFFFFFFu) {
        if (tid == 0) atomicOr(A.status, kStatusHugeSlice);
        return;
    }
    const uint4 *src0 = reinterpret_cast<const uint4 *>(A.p3) + (((uint64_t)d0.y << 32) | d0.x);
    // a lane's four groups of a batch of 1024 (group tid + 256 j): four loads in a row, none under a branch, a piece past the
    // tile's groups loading ONE group for the whole wave (one request) -- or, SP, no load at all for such a piece
    uint4 V[NL];
    auto issue = [&](uint32_t batch) {
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const uint32_t pg = batch * kStageGroups + (uint32_t)(j * T + wave * 64);
            if (MEMO_T_ABLATE & 16) {
                V[j] = make_uint4(0x0000003Fu, 0x0000003Fu, 0x0000003Fu, 0x0000003Fu);  // (rows that cannot write)
                continue;
            }
            // (a piece that straddles the slice's end reads up to 63 groups BEHIND the slice -- half a KiB per tile on average: TCP_TCC_READ_REQ
            // is 24 % above a k = 17 view's bytes, profiles/r05_six_rows.txt.  Those are the next tile's first lines, which that tile --
            // same XCD, running beside this one -- fetches anyway: clamping the lanes to the slice's last group measured 0.5-1.5 % SLOWER,
            // same box, alternating builds; -DMEMO_SLICE_CLAMP keeps the experiment.)
            const uint32_t at = pg + (uint32_t)lane;
#ifdef MEMO_SLICE_CLAMP
            V[j] = src0[pg < g.ng ? (at < g.ng ? at : g.ng - 1u) : 0u];
#else
            if constexpr (SP) {
                if (pg < g.ng) V[j] = src0[at];
            } else {
                V[j] = src0[pg < g.ng ? at : 0u];
            }
#endif
        }
    };
    issue(0);
    const uint32_t lds_base = (uint32_t)(size_t)(__attribute__((address_space(3))) uint32_t *)lds;
    const int HL = A.hl, W = A.w;
    if (!(MEMO_T_ABLATE & 2)) clear_levels<NLEV, T>(lds_base, ((uint32_t)(A.ncols - 1) << SH) | ((1u << SH) - 1u));
    RowConst C;
    C.km1 = A.km1;
    C.status = A.status;
    C.ls4 = 4u * kLS;
    C.bias4 = (uint32_t)pin_vgpr((int)(lds_base + 4u * (uint32_t)HL - (uint32_t)(32 - NLEV) * C.ls4));
    C.top_bit = (uint32_t)pin_vgpr((int)0x80000000u);
    C.a10s = (uint32_t)pin_vgpr((int)(((tabs * (uint32_t)W) & 1023u) << 6));
    SixConst C6;
    C6.km1 = A.km1;
    C6.status = A.status;
    C6.ls4 = C.ls4;
    C6.bias4 = C.bias4;
    C6.top_bit = C.top_bit;
    C6.nega = (uint32_t)pin_vgpr((int)((0u - tabs * (uint32_t)W) & 1023u));
    const uint32_t span = g.end - g.first;
    barrier_lds();  // the level arrays are clear

    const uint32_t nbatch = (g.ng + kStageGroups - 1) / kStageGroups;
    for (uint32_t batch = 0; batch == 0 || batch < nbatch; ++batch) {
        if (batch) issue(batch);  // (a tile with more than 5120 rows: the rest)
        if (MEMO_T_ABLATE & 1) {
#pragma unroll
            for (int j = 0; j < NL; ++j) asm volatile("" ::"v"(V[j].x), "v"(V[j].y), "v"(V[j].z), "v"(V[j].w));
            continue;
        }
        const uint32_t gbase = batch * kStageGroups;
        const uint32_t gleft = g.ng > gbase ? g.ng - gbase : 0;
        if constexpr (SIX) {
            six_pieces<T, NL, AW>(V, lane, wave, gleft, C6);
            continue;
        }
        // a wave whose four pieces (64 groups each, 256 apart) all lie inside the slice: twenty rows, not one test
        const uint32_t w_row0 = 5u * (gbase + (uint32_t)wave * 64u);
        if (w_row0 >= g.first && w_row0 + 5u * ((uint32_t)(NL - 1) * T + 64u) <= g.end) {
#pragma unroll
            for (int j = 0; j < NL; ++j) group_rows<false, A9, AW>(V[j], C, 0, 0);
            continue;
        }
        reg_pieces<T, NL, 0, A9, AW>(V, tid, wave, gbase, gleft, g, C, span);
    }
    barrier_lds();  // (lgkmcnt(0): the ds_min above are invisible to the compiler)

    // fold in registers + store (halo_fold_store_dpp of memo_sweep_cons.hip, unrolled for NLEV)
    OutT *out = static_cast<OutT *>(A.out);
    const int cells = HL + W;
    constexpr int ctx = NLEV <= 1 ? 0 : (NLEV <= 3 ? 1 : 1 << (NLEV - 3));
    constexpr int valid = 64 - ctx;
    const int64_t a_rel = (int64_t)tabs * W - A.qs;  // the tile's first position, as an output index
    const int64_t ob = a_rel - HL;
    const int64_t o_lo = a_rel + (tile == 0 ? A.x_lo_first : 0);
    const int64_t o_hi = a_rel + (tile == (uint32_t)A.ntiles - 1u ? A.x_hi_last : W);
    for (int base = wave * 4 * valid; base + 4 * ctx < cells; base += NW * 4 * valid) {
        const int x0 = base + 4 * lane;
        const int xr = x0 < kLS - 4 ? x0 : kLS - 4;  // (past the array: lanes whose results are dropped below)
        u32x4 L[6];
        read_levels<NLEV>(lds_base + 4u * (uint32_t)xr, L);
        auto lv = [&](int i) { return make_uint4(L[i].x, L[i].y, L[i].z, L[i].w); };
        uint4 M = lv(0);
        if constexpr (MEMO_T_ABLATE & 4) {
            M = lv(NLEV - 1);
        } else {
        if constexpr (NLEV >= 6) fold_step_dpp<4>(M, lv(NLEV - 5), lane);
        if constexpr (NLEV >= 5) fold_step_dpp<3>(M, lv(NLEV - 4), lane);
        if constexpr (NLEV >= 4) fold_step_dpp<2>(M, lv(NLEV - 3), lane);
        if constexpr (NLEV >= 3) fold_step_dpp<1>(M, lv(NLEV - 2), lane);
        if constexpr (NLEV >= 2) fold_step_dpp<0>(M, lv(NLEV - 1), lane);
        }
        if (MEMO_T_ABLATE & 8) {
            asm volatile("" ::"v"(M.x), "v"(M.y), "v"(M.z), "v"(M.w));
            continue;
        }
        if (lane < ctx || x0 >= cells) continue;
        const int64_t o = ob + x0;
        if (o >= o_lo && o + 4 <= o_hi) {
            if constexpr (A9) {  // (store_four, memo_sweep.h: the address is whatever the window's start makes of it)
                store_four(out + o, (M.x >> 23) | ((M.y >> 23) << 16), (M.z >> 23) | ((M.w >> 23) << 16));
            } else if constexpr (sizeof(OutT) == 1) {
                store_four(out + o, __builtin_amdgcn_perm(M.y, M.x, 0x0c0c0703u) | __builtin_amdgcn_perm(M.w, M.z, 0x07030c0cu));
            } else {
                store_four(out + o, __builtin_amdgcn_perm(M.y, M.x, 0x0c070c03u), __builtin_amdgcn_perm(M.w, M.z, 0x0c070c03u));
            }
        } else {
            const uint32_t v[4] = {M.x >> SH, M.y >> SH, M.z >> SH, M.w >> SH};
            for (int i = 0; i < 4; ++i)
                if (o + i >= o_lo && o + i < o_hi) out[o + i] = (OutT)v[i];
        }
    }
}

template <typename OutT, int T, bool A9, bool AW, bool SP>
SweepKernel kernel_for(int nlev) {
    switch (nlev) {
        case 1: return (SweepKernel)sweep_conservation_halo3t_kernel<1, OutT, T, A9, AW, false, SP>;
        case 2: return (SweepKernel)sweep_conservation_halo3t_kernel<2, OutT, T, A9, AW, false, SP>;
        case 3: return (SweepKernel)sweep_conservation_halo3t_kernel<3, OutT, T, A9, AW, false, SP>;
        case 4: return (SweepKernel)sweep_conservation_halo3t_kernel<4, OutT, T, A9, AW, false, SP>;
        case 5: return (SweepKernel)sweep_conservation_halo3t_kernel<5, OutT, T, A9, AW, false, SP>;
        case 6: return (SweepKernel)sweep_conservation_halo3t_kernel<6, OutT, T, A9, AW, false, SP>;
    }
    return nullptr;
}

template <bool AW, bool SP>
SweepKernel kernel_of(int nlev, int elem_bytes, bool annot9) {  // (256 .. 511 genomes: nine-bit orders, uint16 results)
    return annot9 ? kernel_for<uint16_t, 256, true, AW, SP>(nlev)
                  : (elem_bytes == 1 ? kernel_for<uint8_t, 256, false, AW, SP>(nlev) : kernel_for<uint16_t, 256, false, AW, SP>(nlev));
}

template <typename OutT, bool AW, bool SP>
SweepKernel kernel_six(int nlev) {
    switch (nlev) {
        case 1: return (SweepKernel)sweep_conservation_halo3t_kernel<1, OutT, 256, false, AW, true, SP>;
        case 2: return (SweepKernel)sweep_conservation_halo3t_kernel<2, OutT, 256, false, AW, true, SP>;
        case 3: return (SweepKernel)sweep_conservation_halo3t_kernel<3, OutT, 256, false, AW, true, SP>;
        case 4: return (SweepKernel)sweep_conservation_halo3t_kernel<4, OutT, 256, false, AW, true, SP>;
        case 5: return (SweepKernel)sweep_conservation_halo3t_kernel<5, OutT, 256, false, AW, true, SP>;
    }
    return nullptr;
}

}  // namespace

namespace memo {

void drop_tile_tables(memo_index *ix) {  // (callers have the device drained: pack / destroy paths)
    for (memo_index::TileTable &t : ix->ttabs)
        if (t.d) (void)hipFree(t.d);
    ix->ttabs.clear();
}

// the table of (row source, tile width, k): built by the first query that needs it (or by memo_index_prepare), kept with the
// index -- one per combination in use, so that a service that cycles through k classes finds every one of them again
// (round 3 kept four and rebuilt, behind a hipDeviceSynchronize, on the fifth: ADVICE r03); past kMaxTileTables the least
// recently used one is retired (memo_common.h: no wait on the query path)
static int tile_table(memo_index *ix, const void *rows_of, const int64_t *boff, int w, int km1, hipStream_t st, const void **tab,
                      int64_t *ntab, int rpg) {
    memo_index::TileTable *slot = nullptr;
    for (memo_index::TileTable &t : ix->ttabs)
        if (t.d && t.w == w && t.km1 == km1 && t.rows_of == rows_of) slot = &t;
    if (!slot) {
        const int64_t top = ix->max_s < 0 ? 0 : ix->max_s;
        const int64_t n = (top + km1 + ((int64_t)1 << ix->bshift)) / w + 3;  // past the last row: empty slices
        if (n >= ((int64_t)1 << 31)) return 1;
        if (ix->ttabs.size() >= kMaxTileTables) {
            size_t lru = 0;
            for (size_t i = 1; i < ix->ttabs.size(); ++i)
                if (ix->ttabs[i].stamp < ix->ttabs[lru].stamp) lru = i;
            retire(ix, ix->ttabs[lru].d, (uint64_t)ix->ttabs[lru].n * sizeof(TileDesc));
            ix->ttabs.erase(ix->ttabs.begin() + (long)lru);
        }
        void *d = nullptr;
        const hipError_t aerr = side_alloc(&d, (size_t)n * sizeof(TileDesc));
        if (aerr == hipErrorOutOfMemory) return kNoRoom;  // (the caller takes the kernel that needs no table)
        HIP_TRY(aerr);
        hipLaunchKernelGGL(tile_table_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, boff, (int64_t)ix->nb, ix->bbase,
                           ix->bshift, w, km1, n, static_cast<TileDesc *>(d), rpg);
        hipError_t err = hipGetLastError();
        // complete before this call returns: the next query may come on another stream, and nothing would order its sweep
        // behind this kernel (once per index, tile width and k: some tens of microseconds)
        if (err == hipSuccess) err = hipStreamSynchronize(st);
        if (err != hipSuccess) {
            (void)hipFree(d);
            return fail(MEMO_EHIP, "tile table: %s", hipGetErrorString(err));
        }
        memo_index::TileTable t;
        t.rows_of = rows_of;
        t.w = w;
        t.km1 = km1;
        t.d = d;
        t.n = n;
        ix->ttabs.push_back(t);
        slot = &ix->ttabs.back();
    }
    slot->stamp = ++ix->ttab_clock;
    *tab = slot->d;
    *ntab = slot->n;
    return MEMO_OK;
}

// Launch the table-driven dense-row sweep if this query fits it (else return 1: the caller takes
// sweep_conservation_halo3_kernel).  A: filled for the unclipped sweep (hl, w, ls, nlev, ncols); tw = tile width.
int launch_halo3t(memo_index *ix, SweepArgs &A, int tw, int elem_bytes, hipStream_t st, bool annot9, bool all_write, bool six) {
    if (annot9 && elem_bytes != 2) return 1;
    if (six && (annot9 || A.nlev > 5 || ix->bshift != 5)) return 1;
    if (!A.p3 || A.ls > kLS || A.nlev < 1 || A.nlev > 6 || A.km1 > 63 || A.qs < 0) return 1;
    const int64_t q = A.qs / tw, tile0 = q * tw;
#ifdef MEMO_TABLE_RASTER_ONLY  // (A/B builds: rounds 3-4 took only windows whose start is a multiple of four -- aligned result stores)
    if ((tile0 - A.qs) & 3) return 1;
#endif
    const int64_t ntiles = ((A.qe - tile0) + tw - 1) / tw;
    if (ntiles + 8 >= ((int64_t)1 << 31) || q + ntiles >= ((int64_t)1 << 31)) return 1;
    const void *tab = nullptr;
    int64_t ntab = 0;
    const int rc = tile_table(ix, A.p3, A.boff, tw, A.km1, st, &tab, &ntab, six ? 6 : 5);
    if (rc) return rc;
    A.tile0 = tile0;
    A.ntiles = ntiles;
    A.tiles_per_xcd = (ntiles + 7) / 8;
    A.tile_abs0 = q;
    A.ttab = tab;
    A.ntab = ntab;
    A.x_lo_first = (int)(A.qs - tile0);
    A.x_hi_last = (int)(A.qe - (tile0 + (ntiles - 1) * tw));
    // (Two waves per tile, each with twice the rows -- T = 128 -- were tried for row sources with few rows per position, where
    // what a wave does around its rows outweighs the rows: 7-9 % SLOWER on the k-class views of config 3 (0.217 against
    // 0.203 ms at k = 31, 0.168 against 0.154 at k = 17) and 1-4 % slower on all the rows; profiles/r03_views.txt.  Sixteen
    // waves per CU hide the scatter's LDS latency worse than thirty-two, whatever they save in instructions.)
    // few rows per tile?  (the rows the sweep reads, spread over the index's span: groups per tile against the 1024 of a batch)
    const double span = (double)(ix->max_s - ix->min_s) + 1.0;
    const double groups_per_tile = (double)ix->last_rows_read / (six ? 6.0 : 5.0) * (double)tw / (span > 1.0 ? span : 1.0);
#ifdef MEMO_SPARSE_NEVER   // (A/B builds: rounds 3-5's loads everywhere / the skipped loads everywhere)
    const bool sp = false;
#elif defined(MEMO_SPARSE_ALWAYS)
    const bool sp = true;
#else
    const bool sp = groups_per_tile < 768.0;
#endif
    SweepKernel kern = sp ? (all_write ? kernel_of<true, true>(A.nlev, elem_bytes, annot9) : kernel_of<false, true>(A.nlev, elem_bytes, annot9))
                          : (all_write ? kernel_of<true, false>(A.nlev, elem_bytes, annot9) : kernel_of<false, false>(A.nlev, elem_bytes, annot9));
    if (six) {
        if (sp)
            kern = elem_bytes == 1 ? (all_write ? kernel_six<uint8_t, true, true>(A.nlev) : kernel_six<uint8_t, false, true>(A.nlev))
                                   : (all_write ? kernel_six<uint16_t, true, true>(A.nlev) : kernel_six<uint16_t, false, true>(A.nlev));
        else
            kern = elem_bytes == 1 ? (all_write ? kernel_six<uint8_t, true, false>(A.nlev) : kernel_six<uint8_t, false, false>(A.nlev))
                                   : (all_write ? kernel_six<uint16_t, true, false>(A.nlev) : kernel_six<uint16_t, false, false>(A.nlev));
    }
    if (!kern) return 1;
    if (g_prepare_only) return MEMO_OK;  // memo_index_prepare: the table is built, nothing is launched
    if (int prc = refuse_plan_pointer(A.out)) return prc;
    hipLaunchKernelGGL(kern, dim3((unsigned)(A.tiles_per_xcd * 8)), dim3(256), (size_t)A.nlev * 4096, st, A);
    HIP_TRY(hipGetLastError());
    return MEMO_OK;
}

}  // namespace memo
