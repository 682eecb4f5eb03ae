// memo_view.hip -- k-class VIEWS of an index's rows, and the rule that decides when one is built.
//
// /root/reference/src/memo_query.py:45-49 recentres the rows, casts the shadow and DROPS the rows that cannot write at the
// query's k -- per query.  A resident index does the dropping once per k class: a view holds the rows whose overlap is below
// the class's cap (all a query with k - 1 <= cap can be touched by), with a bucket table of its own.  This file builds the
// views (dense rows: count -> scan -> one fused pass that compacts, places the rows inside their 16-byte groups against LDS
// bank conflicts and packs them; 4-byte words: keep / scan / scatter + the order inside the buckets), keeps them within their
// memory budget, and decides WHEN a view (or the query order of the 4-byte rows) is worth its pass: the ski-rental rule of
// view_due() below.
#include "memo_common.h"

using namespace memo;

namespace {

// two-level exclusive scan of count[] (n entries): local[i] = prefix inside i's block of 1024, blocksum[b] = the block's total
__global__ __launch_bounds__(256) void scan_local_kernel(const uint32_t *count, uint64_t n, uint32_t *local, uint64_t *blocksum) {
    __shared__ uint32_t part[256];
    const uint64_t base = blockIdx.x * (uint64_t)1024 + 4 * threadIdx.x;
    uint32_t v[4], sum = 0;
    for (int i = 0; i < 4; ++i) {
        v[i] = base + i < n ? count[base + i] : 0;
        sum += v[i];
    }
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const uint32_t add = threadIdx.x >= (unsigned)d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum;
    for (int i = 0; i < 4; ++i) {
        if (base + i < n) local[base + i] = run;
        run += v[i];
    }
    if (threadIdx.x == 255) blocksum[blockIdx.x] = part[255];
}

// exclusive scan of blocksum[] in place (one workgroup; nb entries), total -> blocksum[nb]
__global__ __launch_bounds__(1024) void scan_blocks_kernel(uint64_t *blocksum, uint64_t nb) {
    __shared__ uint64_t part[1024];
    __shared__ uint64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint64_t at = 0; at < nb; at += 1024) {
        const uint64_t i = at + threadIdx.x;
        const uint64_t v = i < nb ? blocksum[i] : 0;
        part[threadIdx.x] = v;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            const uint64_t add = threadIdx.x >= (unsigned)d ? part[threadIdx.x - d] : 0;
            __syncthreads();
            part[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < nb) blocksum[i] = carry + part[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += part[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) blocksum[nb] = carry;
}

__device__ __forceinline__ uint64_t kept_before(uint64_t r, const uint32_t *keep, const uint32_t *local, const uint64_t *blockpre) {
    const uint64_t w = r >> 5;
    return blockpre[w >> 10] + local[w] + (uint32_t)__popc(keep[w] & ((1u << (r & 31)) - 1u));
}

// boff3[b] = rows that stay among the first boff[b] rows; the last entry is pinned to the total
__global__ void dense_table_kernel(const int64_t *boff, uint64_t nb, uint64_t rows, uint64_t total, const uint32_t *keep,
                                   const uint32_t *local, const uint64_t *blockpre, int64_t *boff3) {
    const uint64_t b = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (b >= nb) return;
    const uint64_t r = (uint64_t)boff[b];
    boff3[b] = (b == nb - 1 || r >= rows) ? (int64_t)total : (int64_t)kept_before(r, keep, local, blockpre);
}

// the same for the 4-byte words (formats 4 and 12): keep the rows whose overlap byte is below cap
__global__ __launch_bounds__(256) void packed_keep_kernel(const uint32_t *pk, uint64_t rows, int len_shift, uint32_t cap, uint32_t *keep,
                                                          uint32_t *count) {
    const uint64_t top = (rows + 255) & ~(uint64_t)255;
    for (uint64_t r = blockIdx.x * (uint64_t)256 + threadIdx.x; r < top; r += (uint64_t)gridDim.x * 256) {
        const uint32_t len = r < rows ? (pk[r] >> len_shift) & 0xFFu : 255u;
        const unsigned long long m = __ballot(r < rows && len < cap);
        const int lane = threadIdx.x & 63;
        if ((lane & 31) == 0 && (r >> 5) < ((rows + 31) >> 5)) {
            const uint32_t half = (uint32_t)(m >> (lane & 32));
            keep[r >> 5] = half;
            count[r >> 5] = (uint32_t)__popc(half);
        }
    }
}

__global__ __launch_bounds__(256) void packed_scatter_kernel(const uint32_t *pk, uint64_t rows, const uint32_t *keep, const uint32_t *local,
                                                             const uint64_t *blockpre, uint32_t *out) {
    for (uint64_t r = blockIdx.x * (uint64_t)256 + threadIdx.x; r < rows; r += (uint64_t)gridDim.x * 256)
        if ((keep[r >> 5] >> (r & 31)) & 1u) out[kept_before(r, keep, local, blockpre)] = pk[r];
}

// ======================================================================================================================
// Dense k-class views, round 5: count -> scan -> ONE fused pass.
//
// Round 4 built a dense view with five kernels -- keep bits (a lane per ROW: five lanes fetched the same 16 bytes), a scatter
// to 4-byte words, the places of the rows inside their groups chosen by one LANE per bucket straight from HBM with one wave per
// workgroup and 42 KiB of LDS (4.0 ms for config 3: three waves per CU, every lane on its own cache lines), the packing, a
// table -- 8.2 ms for BASELINE config 3 at k = 31, the price of 65 whole-chromosome sweeps.  Now:
//   1. view_count_kernel: a lane per 16-byte GROUP: the five keep bits of its rows as a byte, kept rows per 64 groups;
//   2. the two-level scan of those counts; view_table_kernel: the view's bucket table (kept rows before every bucket);
//   3. view_build_kernel<P>: one WAVE per run of up to 64 buckets: streams the run's source groups with coalesced 16-byte
//      loads (the source is read twice in all, nothing else), compacts the kept rows into LDS in source order (ballot +
//      mbcnt: deterministic), then lane j takes bucket j of the run: the
//      choice of one of P places per row (view_place_bucket: round 4's greedy rule, taken with colour MASKS per bank instead of
//      bank masks per colour and without its sort) -- when places were asked for -- and the wave packs the
//      groups and writes them with coalesced 16-byte stores.  P = 5: PackedRows3 groups, the view's rows back to back (a
//      group may straddle two buckets: the two edge groups of a run are written with atomicOr into zeroed groups); P = 6:
//      groups of six rows that carry their bucket and end at bucket boundaries (2.67 B per row; memo_sweep_dense.h:
//      group_rows6), places a bucket leaves empty hold a copy of its last row (min is idempotent).
// A bucket whose kept rows do not fit the LDS stage streams through it in pieces, in source order.
// ======================================================================================================================
// kept rows a piece stages (+ 8 carried): with places, the rows of 64 buckets of BASELINE's shape (a lane each: 39 KiB of LDS, four
// waves per CU -- the greedy loop is what the pass costs, and it wants every lane busy); without, a third of that (13 KiB, twelve
// waves per CU: the pass is a stream, and waves in flight are what hide HBM's latency)
constexpr int kViewCapPlaced = 5120, kViewCapPlain = 2048;
constexpr int kViewSlack = 6 * 64 + 16;     // slots beyond the rows: six-row views pad every bucket to whole groups
constexpr int kViewRun = 64;                // buckets per run: a lane each
constexpr int kColourMax5 = 128, kColourMax6 = 96;  // buckets of more rows keep the order they come in (as in round 4)
inline size_t view_lds_bytes(int cap, bool place) { return (size_t)(cap + 8) * 4 + (size_t)(cap + kViewSlack) * 2 + (place ? 4 * 32 * 64 : 0); }

// row i of group V as W = B | annot << 16   (B = (start mod 2^10) << 6 | min(overlap, 63): 16 bits; annot: 9 bits)
template <int I>
__device__ __forceinline__ uint32_t group_row(const uint4 &V) {
    const uint32_t hi = V.w >> 16;  // (bit i: the ninth annot bit of row i)
    if constexpr (I == 0) return (V.x & 0xFFFFu) | ((V.x >> 24) << 16) | ((hi & 1u) << 24);
    if constexpr (I == 1) return (V.y & 0xFFFFu) | ((V.y >> 24) << 16) | (((hi >> 1) & 1u) << 24);
    if constexpr (I == 2) return (V.z & 0xFFFFu) | ((V.z >> 24) << 16) | (((hi >> 2) & 1u) << 24);
    if constexpr (I == 3) return (V.w & 0xFFFFu) | ((V.w >> 24) << 16) | (((hi >> 3) & 1u) << 24);
    return ((V.x >> 16) & 0xFFu) | (((V.y >> 16) & 0xFFu) << 8) | (((V.z >> 16) & 0xFFu) << 16) | (((hi >> 4) & 1u) << 24);
}

// the keep bits of a group's five rows: overlap < cap, row number < rows
__device__ __forceinline__ uint32_t group_keep(const uint4 &V, uint64_t g, uint64_t rows, uint32_t cap) {
    uint32_t m = ((V.x & 63u) < cap ? 1u : 0u) | ((V.y & 63u) < cap ? 2u : 0u) | ((V.z & 63u) < cap ? 4u : 0u) |
                 ((V.w & 63u) < cap ? 8u : 0u) | (((V.x >> 16) & 63u) < cap ? 16u : 0u);
    const uint64_t r = 5 * g;
    if (r + 5 > rows) m &= r >= rows ? 0u : (1u << (uint32_t)(rows - r)) - 1u;
    return m;
}

__global__ __launch_bounds__(256) void view_count_kernel(const uint4 *__restrict__ p3, uint64_t rows, uint32_t cap,
                                                         uint8_t *__restrict__ keep8, uint32_t *__restrict__ count) {
    const uint64_t groups = (rows + 4) / 5, chunks = (groups + 63) >> 6;
    const int lane = threadIdx.x & 63;
    for (uint64_t chunk = blockIdx.x * 4ull + (threadIdx.x >> 6); chunk < chunks; chunk += gridDim.x * 4ull) {
        const uint64_t g = (chunk << 6) + (uint64_t)lane;
        uint32_t mask = 0;
        if (g < groups) mask = group_keep(p3[g], g, rows, cap);
        keep8[g] = (uint8_t)mask;  // (keep8 holds chunks * 64 bytes)
        uint32_t c = (uint32_t)__popc(mask);
        for (int off = 32; off; off >>= 1) c += (uint32_t)__shfl_xor((int)c, off, 64);
        if (lane == 0) count[chunk] = c;
    }
}

// kept rows among the first r source rows (keep8: a byte per group; local / blockpre: the scan of the kept rows per 64 groups)
__device__ __forceinline__ uint64_t view_kept_before(uint64_t r, uint64_t rows, uint64_t total, const uint8_t *keep8,
                                                     const uint32_t *local, const uint64_t *blockpre) {
    if (r >= rows) return total;
    const uint64_t g = r / 5, chunk = g >> 6;
    const uint32_t gl = (uint32_t)(g & 63), i = (uint32_t)(r - 5 * g);
    uint64_t n = blockpre[chunk >> 10] + local[chunk];
    const uint64_t *p = reinterpret_cast<const uint64_t *>(keep8 + (chunk << 6));
    for (uint32_t j = 0; j < (gl >> 3); ++j) n += (uint64_t)__popcll(p[j]);
    const uint32_t rem = gl & 7;
    const uint64_t last = p[gl >> 3];
    n += (uint64_t)__popcll(last & ((1ull << (8 * rem)) - 1ull));
    n += (uint64_t)__popc((uint32_t)(last >> (8 * rem)) & ((1u << i) - 1u));
    return n;
}

// boffv[b] = kept rows among the first boff[b] source rows; the last entry is pinned to the total
__global__ void view_table_kernel(const int64_t *__restrict__ boff, uint64_t nb, uint64_t rows, const uint8_t *__restrict__ keep8,
                                  const uint32_t *__restrict__ local, const uint64_t *__restrict__ blockpre, uint64_t nblk,
                                  int64_t *__restrict__ boffv) {
    const uint64_t b = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (b >= nb) return;
    const uint64_t total = blockpre[nblk];
    boffv[b] = b == nb - 1 ? (int64_t)total : (int64_t)view_kept_before((uint64_t)boff[b], rows, total, keep8, local, blockpre);
}

// groups of P rows every bucket of the view needs: count[b] = ceil(rows of bucket b / P)
__global__ void view_group_counts_kernel(const int64_t *__restrict__ boffv, int64_t nbuckets, uint32_t *__restrict__ count, int rpg) {
    const int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (b < nbuckets) count[b] = (uint32_t)((boffv[b + 1] - boffv[b] + rpg - 1) / rpg);
}

// P = 5: the groups two runs share (a run's first kept row sits inside a group): zeroed before view_build_kernel or-s into them
__global__ void view_zero_edges_kernel(const int64_t *__restrict__ boffv, int64_t nbuckets, int run_buckets, uint4 *__restrict__ out) {
    const int64_t run = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t nruns = (nbuckets + run_buckets - 1) / run_buckets;
    if (run > nruns) return;
    const int64_t b = run * run_buckets < nbuckets ? run * run_buckets : nbuckets;
    const int64_t v = boffv[b];
    if (v % 5) out[v / 5] = make_uint4(0, 0, 0, 0);
}

struct ViewArgs {
    const uint4 *src;        // the dense rows the view is a view of
    const int64_t *boff;     // ... and their bucket table (nbuckets + 1 entries)
    const int64_t *boffv;    // kept rows before every bucket (nbuckets + 1 entries; the view's own table when P = 5)
    const uint32_t *glocal;  // P = 6: groups before every bucket = gblock[b >> 10] + glocal[b]
    const uint64_t *gblock;
    int64_t nbuckets;
    uint64_t rows;           // source rows
    uint4 *out;              // the view's groups
    int64_t *boff6;          // P = 6: the view's bucket table in row numbers: 6 x groups before the bucket (nbuckets + 1 entries)
    uint32_t cap;            // a row stays when its overlap is below cap
    int km1;                 // the k - 1 whose level arrays the places are chosen for (the class's cap); 0: rows keep their order
    int run_buckets;         // buckets per run (<= kViewRun)
    int stage_rows;          // kept rows the LDS stage holds (kViewCapPlaced / kViewCapPlain)
};

__device__ __forceinline__ uint32_t lanes_below(unsigned long long ballot) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(ballot >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ballot, 0u));
}

// one wave's LDS (dynamic): the kept rows of a piece in source order, slot -> row, colour masks per bank
struct ViewLds {
    uint32_t cap;     // rows the stage holds (ViewArgs::stage_rows)
    uint32_t *stage;  // [cap + 8]: W of the kept rows; the last 8: rows carried from the piece before (and the slot nobody reads)
    uint16_t *inv;    // [cap + kViewSlack]: which staged row a slot of the output holds
    uint8_t *am1;     // [32][64]: colours that hold a row whose FIRST block falls on bank a (a column per lane: its bucket's)
    uint8_t *am2;     // ... two rows or more
    uint8_t *bm1;     // the same for the second block
    uint8_t *bm2;
};

// source rows [r_lo, r_hi) -> their kept rows, in source order, at stage[at ...]; returns how many.
// U 16-byte loads per lane in flight, and the next U requested before these are used (the wave has the registers: its
// occupancy is set by its LDS, so nothing else hides HBM's latency).
__device__ __forceinline__ uint32_t view_load_compact(const ViewArgs &a, const ViewLds &L, uint64_t r_lo, uint64_t r_hi, uint32_t at,
                                                      int lane) {
    const uint64_t g_lo = r_lo / 5, g_hi = (r_hi + 4) / 5;
    constexpr int U = 8;
    uint32_t n = at;
    uint4 V[U], W[U];
    auto request = [&](uint4 (&R)[U], uint64_t g0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t g = g0 + (uint64_t)(64 * u + lane);
            R[u] = g < g_hi ? a.src[g] : make_uint4(63u, 63u, 63u, 63u);
        }
    };
    auto compact = [&](const uint4 (&R)[U], uint64_t g0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t g = g0 + (uint64_t)(64 * u + lane);
            if (g0 + (uint64_t)(64 * u) >= g_hi) break;  // (wave-uniform)
            uint32_t m = g < g_hi ? group_keep(R[u], g, a.rows, a.cap) : 0u;
            const uint64_t r = 5 * g;  // rows of the group outside [r_lo, r_hi): the first and the last group of the range
            if (r < r_lo) m &= ~((1u << (uint32_t)(r_lo - r)) - 1u);
            if (r + 5 > r_hi) m &= r >= r_hi ? 0u : (1u << (uint32_t)(r_hi - r)) - 1u;
            const unsigned long long b0 = __ballot(m & 1u), b1 = __ballot(m & 2u), b2 = __ballot(m & 4u), b3 = __ballot(m & 8u),
                                     b4 = __ballot(m & 16u);
            uint32_t idx = n + lanes_below(b0) + lanes_below(b1) + lanes_below(b2) + lanes_below(b3) + lanes_below(b4);
            // (no branches: a row that goes is stored to a slot nobody reads -- plain stores of many lanes to one address cost one)
            const uint32_t kNowhere = L.cap + 7u;
            L.stage[(m & 1u) ? idx : kNowhere] = group_row<0>(R[u]);
            idx += m & 1u;
            L.stage[(m & 2u) ? idx : kNowhere] = group_row<1>(R[u]);
            idx += (m >> 1) & 1u;
            L.stage[(m & 4u) ? idx : kNowhere] = group_row<2>(R[u]);
            idx += (m >> 2) & 1u;
            L.stage[(m & 8u) ? idx : kNowhere] = group_row<3>(R[u]);
            idx += (m >> 3) & 1u;
            L.stage[(m & 16u) ? idx : kNowhere] = group_row<4>(R[u]);
            n += (uint32_t)(__popcll(b0) + __popcll(b1) + __popcll(b2) + __popcll(b3) + __popcll(b4));
        }
    };
    request(V, g_lo);
    for (uint64_t g0 = g_lo; g0 < g_hi; g0 += 2 * 64 * U) {
        if (g0 + 64 * U < g_hi) request(W, g0 + 64 * U);
        compact(V, g0);
        if (g0 + 64 * U >= g_hi) break;
        if (g0 + 2 * 64 * U < g_hi) request(V, g0 + 2 * 64 * U);
        compact(W, g0 + 64 * U);
    }
    return n - at;
}

// W (group_row) -> the fields of a five-row group / of a six-row group
__device__ __forceinline__ uint4 pack_five(const uint32_t (&w)[5]) {
    uint32_t hi = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) hi |= ((w[i] >> 24) & 1u) << i;
    const uint32_t b4 = w[4] & 0xFFFFu;
    return make_uint4((w[0] & 0xFFFFu) | ((b4 & 0xFFu) << 16) | (((w[0] >> 16) & 0xFFu) << 24),
                      (w[1] & 0xFFFFu) | ((b4 >> 8) << 16) | (((w[1] >> 16) & 0xFFu) << 24),
                      (w[2] & 0xFFFFu) | (((w[4] >> 16) & 0xFFu) << 16) | (((w[2] >> 16) & 0xFFu) << 24),
                      (w[3] & 0xFFFFu) | (hi << 16) | (((w[3] >> 16) & 0xFFu) << 24));
}

// (layout: memo_interleave.hip history / memo_sweep_dense.h: group_rows6)   lo = start mod 32 | overlap << 5
__device__ __forceinline__ uint4 pack_six(const uint32_t (&w)[6]) {
    uint32_t lo[6], an[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const uint32_t ov = w[i] & 63u;
        lo[i] = ((w[i] >> 6) & 31u) | ((ov > 31u ? 31u : ov) << 5);
        an[i] = (w[i] >> 16) & 0xFFu;
    }
    const uint32_t bucket5 = (w[0] >> 11) & 31u;  // (start mod 2^10) >> 5: every row of a group lies in the group's bucket
    return make_uint4(lo[0] | (lo[4] << 10) | (an[0] << 24), lo[1] | (an[4] << 10) | (an[1] << 24),
                      lo[2] | (an[5] << 10) | (bucket5 << 18) | (an[2] << 24), lo[3] | (lo[5] << 10) | (an[3] << 24));
}

// The slots inv[0 .. ns) as groups of P from group `gout` on: every whole group with one 16-byte store; P = 5 only: the rows
// before slot `from` of the first group belong to the run before this one, and with `tail` the last, partial group goes out
// too -- both with atomicOr into groups that were zeroed (view_zero_edges_kernel).  Returns the whole groups.
template <int P>
__device__ __forceinline__ uint32_t view_emit(const ViewArgs &a, const ViewLds &L, uint64_t gout, uint32_t ns, uint32_t from, bool tail,
                                              int lane) {
    const uint32_t whole = ns / P, all = whole + ((tail && ns % P) ? 1u : 0u);
    for (uint32_t t = (uint32_t)lane; t < all; t += 64) {
        const uint32_t lo = t == 0 ? from : 0u, hi = ns - P * t < (uint32_t)P ? ns - P * t : (uint32_t)P;
        if constexpr (P == 6) {
            uint32_t w[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) w[i] = L.stage[L.inv[6 * t + i]];
            a.out[gout + t] = pack_six(w);
        } else {
            uint32_t w[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) w[i] = ((uint32_t)i >= lo && (uint32_t)i < hi) ? L.stage[L.inv[5 * t + i]] : 0u;
            const uint4 g = pack_five(w);
            if (lo == 0 && hi == 5) {
                a.out[gout + t] = g;
            } else {  // (fields of rows that are not this run's are zero: or-ing leaves the neighbour's alone)
                uint32_t *p = reinterpret_cast<uint32_t *>(a.out + gout + t);
                if (g.x) atomicOr(p + 0, g.x);
                if (g.y) atomicOr(p + 1, g.y);
                if (g.z) atomicOr(p + 2, g.z);
                if (g.w) atomicOr(p + 3, g.w);
            }
        }
    }
    return whole;
}

// lane's bucket: rows stage[off .. off + n) -> inv[bslot + place]: the place of every row among the bucket's slots.
// P = 5: slot q of the bucket is the view's row vb + q, its place in its group (vb + q) mod 5; P = 6: 6 * ng slots, place q mod 6.
// The lanes of a wave run this together, a bucket each.
//
// The places (round 4's one-lane-per-bucket kernel, NOTEBOOK.md; the cost model: profiles/r04_lds_atomics.txt, tools/view_order_model.py).  The sweep gives
// a lane one group, and a wave's i-th row instruction visits place i of 64 consecutive groups: a half-wave's 32 atomics of one
// instruction are place i of 32 groups.  A row's two ds_min go to cell A = start - (k - 1) + overlap (first block) and B = start -
// 2^level (second block); the 32 atomics cost max(2, lanes on the fullest bank) cycles.  So the rows of a bucket are split into P
// COLOURS (= places) with, as nearly as a greedy pass gets it, no three rows of one colour on one bank of A nor of B: each row, in
// the order the rows come, takes the colour of least penalty -- 4 for a second row on its A bank, 16 more for a third, 5 / 16 for B --
// then the emptier colour, then the lower one.  Round 4 kept a bank mask per colour and priced every colour for every row, and
// sorted the rows by their A bank first (175 instructions per row and three passes; one lane per bucket straight from HBM: 4 ms
// for config 3).  Here: colour MASKS per bank, in LDS bytes (a column per lane), from which the eight penalty classes are a few
// ANDs; no sort -- by the model the sort is worth 0.23 of 7.65 -> 5.5 cycles per row instruction and half-wave (5.76 without it),
// 0.5 % of a sweep, and it was a third of this pass's instructions and a quarter of its LDS.
template <int P>
__device__ __forceinline__ void view_place_bucket(const ViewArgs &a, const ViewLds &L, uint32_t off, uint32_t n, uint32_t bslot,
                                                  uint32_t vb_mod5, int lane) {
    const uint32_t ng = (n + P - 1) / P;
    const bool colour = a.km1 > 0 && n >= 6 && n <= (uint32_t)(P == 6 ? kColourMax6 : kColourMax5);
    if (!colour) {  // as they come (P = 6: the places the bucket leaves empty hold a copy of its last row)
        if (n)
            for (uint32_t q = 0; q < (P == 6 ? 6 * ng : n); ++q) L.inv[bslot + q] = (uint16_t)(off + (q < n ? q : n - 1));
    }
    if (!__ballot(colour)) return;
    const uint32_t km1 = (uint32_t)a.km1;
    // rooms: how many slots of every place the bucket has (packed: 5 bits per place); skips: its first slot of every place
    uint32_t rooms = 0, skips = 0, notfull = 0;
#pragma unroll
    for (int c = 0; c < P; ++c) {
        uint32_t skip = (uint32_t)c, room = ng;
        if constexpr (P == 5) {
            skip = ((uint32_t)c + 5u - vb_mod5) % 5u;
            room = n > skip ? (n - skip + 4u) / 5u : 0u;
        }
        rooms |= room << (5 * c);
        skips |= skip << (3 * c);
        notfull |= room ? 1u << c : 0u;
    }
    const uint32_t nc = colour ? n : 0u;  // rows this lane places
    uint8_t *a1p = L.am1 + lane, *a2p = L.am2 + lane, *b1p = L.bm1 + lane, *b2p = L.bm2 + lane;
    uint32_t loads = 0;
    uint32_t w_next = nc ? L.stage[off] : 0u;
    for (uint32_t j = 0; __ballot(j < nc); ++j) {
        const uint32_t w = w_next;
        const bool on = j < nc;
        w_next = j + 1 < nc ? L.stage[off + j + 1] : 0u;  // (a row ahead: the read is in flight under this row's arithmetic)
        if (!on) continue;
        const uint32_t ov = w & 63u, s = (w >> 6) & 1023u;
        const uint32_t nn = km1 - ov;  // (>= 1: the view holds the rows whose overlap is below the cap)
        const uint32_t ra = (s - nn) & 31u;                                           // first block: cell start - (k - 1) + overlap
        const uint32_t rb = (s - (1u << (31 - __clz((int)(nn ? nn : 1u))))) & 31u;  // second block: cell start - 2^level
        const uint32_t a1 = a1p[64u * ra], a2 = a2p[64u * ra], b1 = b1p[64u * rb], b2 = b2p[64u * rb];
        // the colours by rising penalty: 0 | 4 | 5 | 9 | 20 | 21 | 25 | 41
        const uint32_t x0 = notfull & ~a1, x1 = notfull & a1 & ~a2, x2 = notfull & a2;
        const uint32_t y0 = ~b1, y1 = b1 & ~b2, y2 = b2;
        uint32_t m = x0 & y0;
        if (!m) m = x1 & y0;
        if (!m) m = x0 & y1;
        if (!m) m = x1 & y1;
        if (!m) m = x2 & y0;
        if (!m) m = x0 & y2;
        if (!m) m = (x2 & y1) | (x1 & y2);
        if (!m) m = notfull;
        uint32_t best = 0xFFFFFFFFu;
#pragma unroll
        for (int c = 0; c < P; ++c) {
            const uint32_t key = (((loads >> (5 * c)) & 31u) << 3) | (uint32_t)c | (((m >> c) & 1u) ? 0u : 0x100u);
            best = key < best ? key : best;
        }
        const uint32_t c = best & 7u, ld = (best >> 3) & 31u;
        const uint32_t slot = P == 6 ? 6u * ld + c : ((skips >> (3 * c)) & 7u) + 5u * ld;
        L.inv[bslot + slot] = (uint16_t)(off + j);
        const uint32_t bit = 1u << c;
        a2p[64u * ra] = (uint8_t)(a2 | (a1 & bit));
        a1p[64u * ra] = (uint8_t)(a1 | bit);
        b2p[64u * rb] = (uint8_t)(b2 | (b1 & bit));
        b1p[64u * rb] = (uint8_t)(b1 | bit);
        loads += 1u << (5 * c);
        if (ld + 1u == ((rooms >> (5 * c)) & 31u)) notfull &= ~bit;
    }
    if constexpr (P == 6) {  // the places no row took (6 ng - n of them, five at most): a copy of the bucket's last row
        if (nc)
            for (int c = 0; c < 6; ++c)
                for (uint32_t g = (loads >> (5 * c)) & 31u; g < ng; ++g) L.inv[bslot + 6u * g + (uint32_t)c] = (uint16_t)(off + n - 1);
    }
}

template <int P>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2))) void view_build_kernel(const ViewArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t view_lds[];
    ViewLds L;
    L.stage = view_lds;
    L.cap = (uint32_t)a.stage_rows;
    L.inv = reinterpret_cast<uint16_t *>(L.stage + L.cap + 8);
    L.am1 = reinterpret_cast<uint8_t *>(L.inv + L.cap + kViewSlack);  // (the four mask arrays exist only when places are chosen: a.km1 > 0)
    L.am2 = L.am1 + 32 * 64;
    L.bm1 = L.am2 + 32 * 64;
    L.bm2 = L.bm1 + 32 * 64;
    const int lane = threadIdx.x;
    const int RB = a.run_buckets;
    const int64_t nruns = (a.nbuckets + RB - 1) / RB;
    for (int64_t run = blockIdx.x; run < nruns; run += gridDim.x) {
        const int64_t b0 = run * RB;
        const int nbk = (int)(a.nbuckets - b0 < RB ? a.nbuckets - b0 : RB);
        const int64_t bj = b0 + (lane < nbk ? lane : nbk);  // (lanes past the run hold its end)
        const uint64_t sb = (uint64_t)a.boff[bj], vb = (uint64_t)a.boffv[bj];
        const int64_t bn = bj < a.nbuckets ? bj + 1 : a.nbuckets;
        const uint64_t se = (uint64_t)a.boff[bn], ve = (uint64_t)a.boffv[bn];
        uint64_t gq = 0;  // P = 6: groups of the view before this lane's bucket
        if constexpr (P == 6) {
            gq = bj < a.nbuckets ? a.gblock[bj >> 10] + a.glocal[bj] : 0;
            if (lane < nbk) a.boff6[bj] = (int64_t)(6 * gq);
            if (b0 + nbk == a.nbuckets && lane == nbk - 1) a.boff6[a.nbuckets] = (int64_t)(6 * (gq + (ve - vb + 5) / 6));
        }
        const uint64_t v_run0 = (uint64_t)__shfl((long long)vb, 0, 64);
        // P = 5: slot 0 of the output staging is the view's row pbase (a multiple of 5); rows before pfrom are not this run's
        uint64_t pbase = v_run0 / 5 * 5, pfrom = v_run0;
        // P = 5: slots of the last, partial group so far: rows carried from the piece before (staged at stage[cap ...]) -- and, at
        // the head of a run, the rows of that group that belong to the run before (never read: view_emit masks them by `from`)
        uint32_t carried = P == 5 ? (uint32_t)(pfrom - pbase) : 0u;
        if (lane < 8) L.inv[lane] = (uint16_t)(L.cap + lane);  // (slots of the first group that are not this run's: never used, but read)
        __syncthreads();
        int s = 0;
        while (s < nbk) {
            const uint64_t v0 = (uint64_t)__shfl((long long)vb, s, 64);
            const unsigned long long fits = __ballot(lane >= s && lane < nbk && ve - v0 <= (uint64_t)L.cap);
            const int e = s + (int)__popcll(fits);  // (ve rises with the lane: the buckets that fit are s .. e - 1)
            if (e > s) {
                // ---- buckets s .. e - 1 whole: load, place, emit ----
                const uint64_t r_lo = (uint64_t)__shfl((long long)sb, s, 64), r_hi = (uint64_t)__shfl((long long)se, e - 1, 64);
                const uint64_t v1 = (uint64_t)__shfl((long long)ve, e - 1, 64);
                const uint32_t got = view_load_compact(a, L, r_lo, r_hi, 0, lane);
                (void)got;
                __syncthreads();
                const bool mine = lane >= s && lane < e;
                const uint32_t n = mine ? (uint32_t)(ve - vb) : 0u, off = (uint32_t)(vb - v0);
                uint64_t gbase = 0;
                uint32_t bslot;
                if constexpr (P == 6) {
                    gbase = (uint64_t)__shfl((long long)gq, s, 64);
                    bslot = (uint32_t)(6 * (gq - gbase));
                } else {
                    bslot = (uint32_t)(vb - pbase);
                }
                if (a.km1 > 0)
                    for (int i = lane; i < 4 * 32 * 64 / 4; i += 64) reinterpret_cast<uint32_t *>(L.am1)[i] = 0;  // (no colour holds a row yet)
                __syncthreads();
                view_place_bucket<P>(a, L, off, n, bslot, (uint32_t)(vb % 5), lane);
                __syncthreads();
                if constexpr (P == 6) {
                    const uint64_t gend = (uint64_t)__shfl((long long)(gq + (ve - vb + 5) / 6), e - 1, 64);
                    view_emit<6>(a, L, gbase, (uint32_t)(6 * (gend - gbase)), 0, false, lane);
                    __syncthreads();
                } else {
                    const uint32_t ns = (uint32_t)(v1 - pbase);
                    const bool last = e == nbk;
                    const uint32_t whole = view_emit<5>(a, L, pbase / 5, ns, (uint32_t)(pfrom - pbase), last, lane);
                    __syncthreads();
                    // the rows of the last, partial group stay for the next piece of this run
                    const uint32_t left = ns - 5 * whole;
                    uint32_t w = 0;
                    if (!last && (uint32_t)lane < left) w = L.stage[L.inv[5 * whole + lane]];
                    __syncthreads();
                    if (!last && (uint32_t)lane < left) {
                        L.stage[L.cap + lane] = w;
                        L.inv[lane] = (uint16_t)(L.cap + lane);
                    }
                    pbase += 5ull * whole;
                    if (pfrom < pbase) pfrom = pbase;
                    carried = left;
                    __syncthreads();
                }
                s = e;
                continue;
            }
            // ---- bucket s alone holds more kept rows than the stage: through it in pieces, in source order ----
            const uint64_t r_lo = (uint64_t)__shfl((long long)sb, s, 64), r_hi = (uint64_t)__shfl((long long)se, s, 64);
            const uint64_t v1 = (uint64_t)__shfl((long long)ve, s, 64);
            uint64_t gcur = 0;
            if constexpr (P == 6) {
                gcur = (uint64_t)__shfl((long long)gq, s, 64);
                carried = 0;
            }
            uint64_t done = v0;  // kept rows of the bucket emitted or carried so far
            const uint64_t piece = (uint64_t)L.cap / 5 * 5;  // source rows per piece
            for (uint64_t r = r_lo; r < r_hi; r += piece) {
                const uint64_t r_end = r + piece < r_hi ? r + piece : r_hi;
                const uint32_t got = view_load_compact(a, L, r, r_end, 0, lane);
                __syncthreads();
                for (uint32_t q = (uint32_t)lane; q < got; q += 64) L.inv[carried + q] = (uint16_t)q;
                __syncthreads();
                done += got;
                const bool last_piece = r_end >= r_hi;
                uint32_t ns = carried + got, whole;
                if constexpr (P == 6) {
                    if (last_piece && ns % 6) {  // (the places the bucket leaves empty: a copy of its last row)
                        const uint32_t pad = 6 - ns % 6;
                        const uint16_t lastrow = L.inv[ns - 1];
                        __syncthreads();
                        if ((uint32_t)lane < pad) L.inv[ns + lane] = lastrow;
                        ns += pad;
                        __syncthreads();
                    }
                    whole = view_emit<6>(a, L, gcur, ns, 0, false, lane);
                    gcur += whole;
                } else {
                    const bool last = last_piece && s + 1 == nbk;
                    whole = view_emit<5>(a, L, pbase / 5, ns, (uint32_t)(pfrom - pbase), last, lane);
                    pbase += 5ull * whole;
                    if (pfrom < pbase) pfrom = pbase;
                }
                __syncthreads();
                const uint32_t left = ns - P * whole;
                uint32_t w = 0;
                if ((uint32_t)lane < left) w = L.stage[L.inv[P * whole + lane]];
                __syncthreads();
                if ((uint32_t)lane < left) {
                    L.stage[L.cap + lane] = w;
                    L.inv[lane] = (uint16_t)(L.cap + lane);
                }
                carried = left;
                __syncthreads();
            }
            (void)done;
            (void)v1;
            if constexpr (P == 6) carried = 0;
            s += 1;
        }
    }
}


}  // namespace

namespace memo {
// The rows of `src` (dense groups, bucket table, row count) whose length field is below `cap`, as dense rows of their own
// with their own bucket table -- or nothing (out->p3 stays NULL) when fewer than min_tenths tenths of the rows would go.
// A row with length >= cap cannot write at any k with k - 1 <= cap.  Synchronous on stream `st`.
// len_shift >= 0: src_p3 / out_p3 are 4-byte WORDS (formats 4 / 12: the overlap byte sits at bit len_shift) instead of dense groups.
// Device memory for what a query builds on the side (views, tile tables).  These are optimisations: when the device has no
// room for them the query runs on the rows it has (callers see kNoRoom, not an error).  memo_debug_fail_side_allocations (AB
// library) makes every such allocation fail: the test of that path.
hipError_t side_alloc(void **p, size_t bytes) {
    if (g_side_alloc_fails) return hipErrorOutOfMemory;  // (memo_debug_fail_side_allocations of the AB library)
    const hipError_t err = hipMalloc(p, bytes);
    if (err == hipErrorOutOfMemory) (void)hipGetLastError();  // (not sticky: later calls are clean)
    return err;
}

// The dense rows of `src` (groups, bucket table of nb entries, row count) whose overlap is below `cap`, as a view of their own
// -- groups of rpg = 5 rows back to back with the kept-rows table, or of rpg = 6 rows that carry their bucket, with a table in
// units of (padded) rows -- or nothing (*out_p3 stays NULL) when fewer than min_tenths tenths of the rows would go.
// colour_km1 > 0: the place of a row inside its group is chosen for the level arrays of k - 1 = colour_km1.  Count, scan and
// the fused pass (view_build_kernel) are queued on `st`; the call waits for them twice (the kept rows decide the allocation).
static int dense_view_build(int device, const uint32_t *src_p3, const int64_t *src_boff, uint64_t rows, uint64_t nb, int cap, int min_tenths,
                            hipStream_t st, int rpg, int colour_km1, uint32_t **out_p3, int64_t **out_boff, uint64_t *out_rows,
                            uint64_t *out_padded) {
    *out_p3 = nullptr;
    *out_boff = nullptr;
    if (!rows || rows >= ((uint64_t)1 << 38) || nb < 2) return MEMO_OK;
    DeviceGuard guard(device);
    const uint64_t groups = (rows + 4) / 5, chunks = (groups + 63) >> 6, nblk = (chunks + 1023) >> 10;
    const uint64_t nbk = nb - 1, nblk6 = (nbk + 1023) >> 10;
    // one allocation for everything that goes again: keep bytes, the two scans' arrays
    auto up = [](uint64_t x) { return (x + 255) & ~(uint64_t)255; };
    const uint64_t o_keep = 0, o_count = o_keep + up(chunks * 64), o_bpre = o_count + up(chunks * 4), o_gcount = o_bpre + up((nblk + 1) * 8),
                   o_gblock = o_gcount + up(rpg == 6 ? nbk * 4 + 4 : 0), tmp_bytes = o_gblock + up(rpg == 6 ? (nblk6 + 1) * 8 : 0);
    char *tmp = nullptr;
    uint4 *outg = nullptr;
    int64_t *boffv = nullptr, *boff6 = nullptr;
    int rc = MEMO_OK;
    do {
        hipError_t err = side_alloc((void **)&tmp, tmp_bytes);
        if (err == hipSuccess) err = side_alloc((void **)&boffv, nb * 8);
        if (err == hipErrorOutOfMemory) { rc = kNoRoom; break; }
        if (err != hipSuccess) { rc = fail(MEMO_EHIP, "dense view: %s", hipGetErrorString(err)); break; }
        uint8_t *keep8 = reinterpret_cast<uint8_t *>(tmp + o_keep);
        uint32_t *count = reinterpret_cast<uint32_t *>(tmp + o_count);
        uint64_t *blockpre = reinterpret_cast<uint64_t *>(tmp + o_bpre);
        uint32_t *gcount = reinterpret_cast<uint32_t *>(tmp + o_gcount);
        uint64_t *gblock = reinterpret_cast<uint64_t *>(tmp + o_gblock);
        const uint4 *p3 = reinterpret_cast<const uint4 *>(src_p3);
        const unsigned cgrid = (unsigned)((chunks + 3) / 4 < 256 * 32 ? (chunks + 3) / 4 : 256 * 32);
        hipLaunchKernelGGL(view_count_kernel, dim3(cgrid), dim3(256), 0, st, p3, rows, (uint32_t)cap, keep8, count);
        hipLaunchKernelGGL(scan_local_kernel, dim3((unsigned)nblk), dim3(256), 0, st, count, chunks, count, blockpre);
        hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(1024), 0, st, blockpre, nblk);
        hipLaunchKernelGGL(view_table_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, st, src_boff, nb, rows, keep8, count, blockpre,
                           nblk, boffv);
        uint64_t total = 0, total6 = 0;
        err = hipGetLastError();
        if (rpg == 6 && err == hipSuccess) {  // groups per bucket -> groups before every bucket
            hipLaunchKernelGGL(view_group_counts_kernel, dim3((unsigned)((nbk + 255) / 256)), dim3(256), 0, st, boffv, (int64_t)nbk, gcount, 6);
            hipLaunchKernelGGL(scan_local_kernel, dim3((unsigned)nblk6), dim3(256), 0, st, gcount, nbk, gcount, gblock);
            hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(1024), 0, st, gblock, nblk6);
            err = hipGetLastError();
            if (err == hipSuccess) err = hipMemcpyAsync(&total6, gblock + nblk6, 8, hipMemcpyDeviceToHost, st);
        }
        if (err == hipSuccess) err = hipMemcpyAsync(&total, blockpre + nblk, 8, hipMemcpyDeviceToHost, st);
        if (err == hipSuccess) err = hipStreamSynchronize(st);
        if (err != hipSuccess) { rc = fail(MEMO_EHIP, "dense view: %s", hipGetErrorString(err)); break; }
        if (total + rows / 10 * (uint64_t)min_tenths > rows) break;  // too few would go
        const uint64_t padded = rpg == 6 ? 6 * total6 : ((total + 15) & ~(uint64_t)15) + kPadRows;
        const uint64_t ngroups = rpg == 6 ? total6 + 64 : dense_groups_for(padded), used = rpg == 6 ? total6 : total / 5;
        err = side_alloc((void **)&outg, ngroups * 16);
        if (err == hipSuccess && rpg == 6) err = side_alloc((void **)&boff6, nb * 8);
        if (err == hipErrorOutOfMemory) { rc = kNoRoom; break; }
        if (err == hipSuccess) err = hipMemsetAsync(outg + used, 0, (ngroups - used) * 16, st);  // (behind the rows; P = 5: the last, partial group too)
        if (err != hipSuccess) { rc = fail(MEMO_EHIP, "dense view: %s", hipGetErrorString(err)); break; }
        ViewArgs a;
        a.src = p3;
        a.boff = src_boff;
        a.boffv = boffv;
        a.glocal = gcount;
        a.gblock = gblock;
        a.nbuckets = (int64_t)nbk;
        a.rows = rows;
        a.out = outg;
        a.boff6 = boff6;
        a.cap = (uint32_t)cap;
        a.km1 = colour_km1;
        // buckets per run: as many as (nearly always) fit the stage whole, so that a run is one piece and every lane has a bucket
        const double per_bucket = (double)total / (double)nbk;
        a.stage_rows = colour_km1 > 0 ? kViewCapPlaced : kViewCapPlain;
        int rb = per_bucket > 1.0 ? (int)(0.85 * a.stage_rows / per_bucket) : kViewRun;
        a.run_buckets = rb > kViewRun ? kViewRun : (rb < 4 ? 4 : rb);
        const size_t lds_bytes = view_lds_bytes(a.stage_rows, colour_km1 > 0);
        const int64_t nruns = ((int64_t)nbk + a.run_buckets - 1) / a.run_buckets;
        if (rpg == 5)
            hipLaunchKernelGGL(view_zero_edges_kernel, dim3((unsigned)((nruns + 1 + 255) / 256)), dim3(256), 0, st, boffv, (int64_t)nbk,
                               a.run_buckets, outg);
        const unsigned grid = (unsigned)(nruns < 256 * 16 * 4 ? nruns : 256 * 16 * 4);
        if (rpg == 6)
            hipLaunchKernelGGL(view_build_kernel<6>, dim3(grid), dim3(64), lds_bytes, st, a);
        else
            hipLaunchKernelGGL(view_build_kernel<5>, dim3(grid), dim3(64), lds_bytes, st, a);
        err = hipGetLastError();
        if (err == hipSuccess) err = hipStreamSynchronize(st);
        if (err != hipSuccess) { rc = fail(MEMO_EHIP, "dense view: %s", hipGetErrorString(err)); break; }
        *out_p3 = reinterpret_cast<uint32_t *>(outg);
        *out_boff = rpg == 6 ? boff6 : boffv;
        *out_rows = total;
        *out_padded = padded;
        outg = nullptr;
        if (rpg == 6) boff6 = nullptr; else boffv = nullptr;
    } while (0);
    (void)hipFree(tmp);
    (void)hipFree(outg);
    (void)hipFree(boffv);
    (void)hipFree(boff6);
    return rc;
}

// The same for the 4-byte WORDS (formats 4 / 12: the overlap byte sits at bit len_shift): the words whose overlap is below cap, in
// the order they come, with the kept-rows table -- keep bits, scan, scatter (the order inside the view's buckets is the caller's).
static int packed_filter(int device, const uint32_t *src, const int64_t *src_boff, uint64_t rows, uint64_t nb, int cap, int min_tenths,
                         hipStream_t st, int len_shift, uint32_t **out_pk, int64_t **out_boff, uint64_t *out_rows, uint64_t *out_padded) {
    *out_pk = nullptr;
    *out_boff = nullptr;
    if (!rows || rows >= ((uint64_t)1 << 38)) return MEMO_OK;
    DeviceGuard guard(device);
    const uint64_t n32 = (rows + 31) >> 5, nblk = (n32 + 1023) >> 10;
    const unsigned row_grid = (unsigned)((rows + 255) / 256 < ((uint64_t)1 << 20) ? (rows + 255) / 256 : (uint64_t)1 << 20);
    uint32_t *keep = nullptr, *local = nullptr, *words = nullptr;
    uint64_t *blockpre = nullptr;
    int64_t *boffv = nullptr;
    int rc = MEMO_OK;
    do {
        hipError_t err = side_alloc((void **)&keep, n32 * 4 + 4);
        if (err == hipSuccess) err = side_alloc((void **)&local, n32 * 4);
        if (err == hipSuccess) err = side_alloc((void **)&blockpre, (nblk + 1) * 8);
        if (err == hipErrorOutOfMemory) { rc = kNoRoom; break; }
        if (err != hipSuccess) { rc = fail(MEMO_EHIP, "packed view: %s", hipGetErrorString(err)); break; }
        hipLaunchKernelGGL(packed_keep_kernel, dim3(row_grid), dim3(256), 0, st, src, rows, len_shift, (uint32_t)cap, keep, local);
        hipLaunchKernelGGL(scan_local_kernel, dim3((unsigned)nblk), dim3(256), 0, st, local, n32, local, blockpre);
        hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(1024), 0, st, blockpre, nblk);
        uint64_t total = 0;
        err = hipGetLastError();
        if (err == hipSuccess) err = hipMemcpyAsync(&total, blockpre + nblk, 8, hipMemcpyDeviceToHost, st);
        if (err == hipSuccess) err = hipStreamSynchronize(st);
        if (err != hipSuccess) { rc = fail(MEMO_EHIP, "packed view: %s", hipGetErrorString(err)); break; }
        if (total + rows / 10 * (uint64_t)min_tenths > rows) break;  // too few would go
        const uint64_t padded = ((total + 15) & ~(uint64_t)15) + kPadRows;
        err = side_alloc((void **)&words, padded * 4);
        if (err == hipSuccess) err = side_alloc((void **)&boffv, nb * 8);
        if (err == hipErrorOutOfMemory) { rc = kNoRoom; break; }
        if (err == hipSuccess) err = hipMemsetAsync(words + total, 0, (padded - total) * 4, st);
        if (err != hipSuccess) { rc = fail(MEMO_EHIP, "packed view: %s", hipGetErrorString(err)); break; }
        hipLaunchKernelGGL(packed_scatter_kernel, dim3(row_grid), dim3(256), 0, st, src, rows, keep, local, blockpre, words);
        hipLaunchKernelGGL(dense_table_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, st, src_boff, nb, rows, total, keep, local,
                           blockpre, boffv);
        err = hipGetLastError();
        if (err == hipSuccess) err = hipStreamSynchronize(st);
        if (err != hipSuccess) { rc = fail(MEMO_EHIP, "packed view: %s", hipGetErrorString(err)); break; }
        *out_pk = words;
        *out_boff = boffv;
        *out_rows = total;
        *out_padded = padded;
        words = nullptr;
        boffv = nullptr;
    } while (0);
    (void)hipFree(keep);
    (void)hipFree(local);
    (void)hipFree(blockpre);
    (void)hipFree(words);
    (void)hipFree(boffv);
    return rc;
}

// ix->p3 holds every row of the index (rows3 == rows, no boff3).  When more than a tenth of them can never write at
// k <= 64 (6-bit length field saturated), rebuild the dense rows without them, with a bucket table of their own.
int dense_compact(memo_index *ix) {
    if (!ix->p3 || ix->boff3 || !ix->rows || g_dense_keep_all) return MEMO_OK;
    uint32_t *p3n = nullptr;
    int64_t *boff3 = nullptr;
    uint64_t total = 0, padded3 = 0;
    int rc = dense_view_build(ix->device, ix->p3, ix->boff, ix->rows, ix->nb, 63, 1, nullptr, 5, 0, &p3n, &boff3, &total, &padded3);
    if (rc == kNoRoom) rc = MEMO_OK;  // (no room for a second copy: every row stays)
    if (rc || !p3n) return rc;
    DeviceGuard guard(ix->device);
    drop_tile_tables(ix);
    (void)hipFree(ix->p3);
    ix->p3 = p3n;
    ix->boff3 = boff3;
    ix->rows3 = total;
    ix->padded3 = padded3;
    return MEMO_OK;
}

thread_local bool g_dense_keep_all = false;   // (AB library: memo_debug_dense_keep_all)
thread_local int g_one_shot_way = 0;          // (AB library: memo_debug_one_shot_way: 1 = int64 columns, 2 = 4-byte words)

void retire(memo_index *ix, void *p, uint64_t bytes) {
    if (!p) return;
    memo_index::Retired r;
    r.p = p;
    r.bytes = bytes;
    ix->retired.push_back(r);
    ix->retired_bytes += bytes;
}

void flush_retired(memo_index *ix) {
    for (memo_index::Retired &r : ix->retired) (void)hipFree(r.p);
    ix->retired.clear();
    ix->retired_bytes = 0;
}

static void retire_view(memo_index *ix, memo_index::DenseView &v, bool dense) {
    for (size_t i = 0; i < ix->ttabs.size();) {  // the tile tables made for it go with it (a later allocation may land on its address)
        if (ix->ttabs[i].rows_of == v.p3) {
            retire(ix, ix->ttabs[i].d, (uint64_t)ix->ttabs[i].n * 32);
            ix->ttabs.erase(ix->ttabs.begin() + (long)i);
        } else {
            ++i;
        }
    }
    retire(ix, v.p3, v.bytes);
    retire(ix, v.boff, ix->nb * 8);
    const int again = v.backoff < (1 << 16) ? v.backoff * 4 : v.backoff;  // back-off: see DenseView
    const int ask = v.ask_after ? (v.ask_after < (1 << 16) ? v.ask_after * 4 : v.ask_after) : 16;
    v = memo_index::DenseView();
    v.backoff = again;
    v.ask_after = ask;
}

void drop_dense_views(memo_index *ix) {
    for (memo_index::DenseView &v : ix->views) {
        (void)hipFree(v.p3);
        (void)hipFree(v.boff);
        v = memo_index::DenseView();
    }
    for (memo_index::DenseView &v : ix->views6) {
        (void)hipFree(v.p3);
        (void)hipFree(v.boff);
        v = memo_index::DenseView();
    }
}

void drop_packed_views(memo_index *ix) {
    for (memo_index::DenseView &v : ix->pviews) {
        (void)hipFree(v.p3);
        (void)hipFree(v.boff);
        v = memo_index::DenseView();
    }
}

// All the views of one row source together may take view_budget_pct percent (200 by default: memo_index_set_option) of the bytes
// of the rows they are views of (sixteen classes of the dense rows would come to 4.5 times on BASELINE's generator); past that
// the least recently used view is RETIRED -- with the tile tables made for it -- and its class starts from nothing again,
// towards a threshold four times the last one (a service that cycles through more classes than the budget holds settles on
// the classes that fit and reads all the rows for the others, instead of rebuilding a view every few queries: ADVICE r03).
// Nothing is waited for here: a sweep queued on any of the caller's streams may still read the view, so its buffers go to
// the index's retire list (memo_common.h) -- unless that list has itself grown past the budget: then the device is drained.
static void keep_views_in_budget(memo_index *ix, const memo_index::DenseView *fresh, uint64_t base_bytes, bool dense) {
    auto bytes_of = [&](const memo_index::DenseView &v) -> uint64_t {
        return v.p3 ? v.bytes + ix->nb * 8 : 0;
    };
    const uint64_t budget = base_bytes / 100 * (uint64_t)ix->view_budget_pct;
    constexpr int kDense = (int)(sizeof(ix->views) / sizeof(ix->views[0])), kPacked = (int)(sizeof(ix->pviews) / sizeof(ix->pviews[0]));
    for (;;) {  // (the five- and six-row views of the dense rows share one budget)
        uint64_t total = 0;
        memo_index::DenseView *lru = nullptr;
        auto look = [&](memo_index::DenseView &v) {
            total += bytes_of(v);
            if (v.p3 && &v != fresh && (!lru || v.stamp < lru->stamp)) lru = &v;
        };
        if (dense) {
            for (int i = 0; i < kDense; ++i) look(ix->views[i]);
            for (int i = 0; i < kDense; ++i) look(ix->views6[i]);
        } else {
            for (int i = 0; i < kPacked; ++i) look(ix->pviews[i]);
        }
        if (total <= budget || !lru) break;
        retire_view(ix, *lru, dense);
    }
    if (ix->retired_bytes > budget + base_bytes) {  // (rare: many evictions and no memo_query_check in between)
        (void)hipDeviceSynchronize();
        flush_retired(ix);
    }
}

// ---- when is a pass over the rows worth it? ---------------------------------------------------------------------------------
// memo_query.py:45-49 drops the rows that cannot write at the query's k in every query; a view drops them once -- at the price
// of a pass over the rows.  Rounds 3-4 paid that price inside the fifth query of a class, whatever the queries were: on BASELINE
// config 3 a view paid for itself after 65 whole-chromosome queries, so queries 5 .. 69 of a class were a net loss, and a host
// that sweeps 1-Mbp windows paid 8 ms for a view that saves it a microsecond per query.  Now every query of a
// class that runs without its view adds what the view would have saved it -- the rows of its window the view leaves out x
// what a sweep pays per row -- and the view is built by the query that finds the sum has reached the view's estimated cost:
// the ski-rental rule (never more than twice the cost of having known the future).  memo_index_prepare builds at once;
// MEMO_OPT_BUILD_COST_PCT scales the threshold (0: the class's first query builds).  The same rule decides when rows that
// came in start order are brought into the query order (order_due).
// Calibration (MI355X, profiles/r05_view_pass.txt): a sweep's time per row it reads; a pass's time per row of its source --
// replaced by what the index's own last pass measured -- plus what allocations and the two waits cost whatever the size.
constexpr double kSweepNsPerRow = 0.00048;      // config 3, k = 31: (0.298 - 0.179 ms) / 2.5e8 rows a view spares
constexpr double kDenseViewNsPerRow = 0.0036;   // count + scan + the fused pass, rows in the order they come (2.0 ms for config 3's 5e8 rows)
constexpr double kPlacedViewNsPerRow = 0.0078;  // ... with the places of the rows chosen (4.1 ms)
constexpr double kPlacedGainPerKm1 = 0.0013;    // what places buy a sweep on the view: 4.4 % at k = 31, 2 % at k = 21 (profiles/r04_view_levels.txt)
constexpr double kPackedViewNsPerRow = 0.03;    // keep, scan, scatter, and the order inside the view's buckets
constexpr double kOrderNsPerRow = 0.014;        // copy + the order inside the buckets (more on indexes of many rows per start)
constexpr double kPassFixedNs = 200e3;

static double rows_in_window(const memo_index *ix, double src_rows, int64_t window, int km1) {
    const double span = (double)ix->max_s - (double)ix->min_s + 1.0;
    const double w = (double)window + (double)km1 + (double)((int64_t)1 << ix->bshift);
    return span <= w ? src_rows : src_rows * w / span;
}

// the share of the index's rows whose overlap is below cap, from the census taken when the rows came into being (a sample;
// 0.5 where there is none)
static double share_below(const memo_index *ix, int cap) {
    if (!ix->len_hist_rows) return 0.5;
    double below = 0;
    for (int v = 0; v < cap && v < 256; ++v) below += ix->len_hist[v];
    return below / (double)ix->len_hist_rows;
}

// kind: 0 a view of the dense rows, 1 of the 4-byte words.  spared: the share of src_rows the view leaves out.
static bool view_due(memo_index *ix, memo_index::DenseView &v, int kind, double src_rows, double spared, int64_t window, int km1) {
    if (g_prepare_only) return true;
    if (spared > 0) v.lost_ns += rows_in_window(ix, src_rows, window, km1) * spared * kSweepNsPerRow;
    if (++v.seen <= v.ask_after) return false;
    constexpr double kConst[3] = {kDenseViewNsPerRow, kPackedViewNsPerRow, kPlacedViewNsPerRow};
    const double per_row = ix->view_ns_per_row[kind] > 0 ? ix->view_ns_per_row[kind] : kConst[kind];
    const double cost = (src_rows * per_row + kPassFixedNs) * (double)v.backoff * (double)ix->build_cost_pct / 100.0;
    return v.lost_ns >= cost;
}

static void view_built(memo_index *ix, float build_ms, int kind, double src_rows) {  // what the pass cost, for the next estimate
    constexpr double kConst[3] = {kDenseViewNsPerRow, kPackedViewNsPerRow, kPlacedViewNsPerRow};
    const double ns = (double)build_ms * 1e6 - kPassFixedNs;
    // within a factor of four of the calibrated constant: the pass is timed with its allocations, and a hipMalloc that stalls (350 ms
    // for 0.85 GB seen on one box: gpurun r5valid) must not make every later view of the index look two hundred times as dear
    const double floor = 0.25 * kConst[kind], ceil = 4.0 * kConst[kind];
    if (src_rows > 0) ix->view_ns_per_row[kind] = ns / src_rows > floor ? (ns / src_rows < ceil ? ns / src_rows : ceil) : floor;
}

// A dense view that exists with its rows in the order they came: is it time to build it again with their places chosen
// (memo_view.hip: view_place_bucket)?  The places buy a sweep a few percent; the pass that chooses them costs twice the pass
// without -- so a class gets its view when the view has paid for itself (view_due), and the places when THEY have.
static bool places_due(memo_index *ix, memo_index::DenseView &v, double src_rows, int64_t window, int km1, bool six) {
    if (g_prepare_only) return true;
    const double read = rows_in_window(ix, (double)(six ? v.padded : v.rows), window, km1);
    v.unplaced_ns += read * 1.5 * kSweepNsPerRow * kPlacedGainPerKm1 * (double)km1;  // (a sweep on a view: 0.72 ps per row it reads)
    const double per_row = ix->view_ns_per_row[2] > 0 ? ix->view_ns_per_row[2] : kPlacedViewNsPerRow;
    return v.unplaced_ns >= (src_rows * per_row + kPassFixedNs) * (double)v.backoff * (double)ix->build_cost_pct / 100.0;
}

// The 4-byte rows brought into a query order (memo_interleave.hip: 2 = the conservation order, 3 = the membership order, 0 =
// start order).  In place: no sweep may be reading them, so the device is drained first (memo_index_pack, the A/B switch).
int order_words_now(memo_index *ix, int mode) {
    ix->order_pending = 0;
    if (mode == 3 && ix->bshift != 5) mode = 2;
    if (!ix->pk || !ix->rows || (ix->packed_fmt != 4 && ix->packed_fmt != 12) || mode == ix->row_order) return MEMO_OK;
    DeviceGuard guard(ix->device);
    HIP_TRY(hipDeviceSynchronize());
    drop_packed_views(ix);  // (views are subsets in the old order)
    if (int rc = interleave_words(ix->pk, ix->boff, ix->nb, ix->bshift, ix->packed_fmt, mode, nullptr, ix->d_scratch)) return rc;
    HIP_TRY(hipDeviceSynchronize());
    ix->row_order = mode;
    return MEMO_OK;
}

// The same from inside a query: OUT of place, on the caller's stream -- sweeps queued on the caller's other streams may still
// read the rows as they are, so nothing is overwritten and nothing waits for the device: a second copy is ordered,
// the call waits for ITS stream (like the query that builds a view), the index switches to the copy, the old rows and their
// views go to the retire list.  No room for the copy: the rows stay as they are (kNoRoom).
static int order_words_on(memo_index *ix, int mode, hipStream_t st) {
    if (mode == 3 && ix->bshift != 5) mode = 2;
    if (!ix->pk || !ix->rows || (ix->packed_fmt != 4 && ix->packed_fmt != 12) || mode == ix->row_order) {
        ix->order_pending = 0;
        return MEMO_OK;
    }
    DeviceGuard guard(ix->device);
    uint32_t *copy = nullptr;
    const size_t bytes = ix->packed_rows * sizeof(uint32_t);
    const hipError_t aerr = side_alloc((void **)&copy, bytes);
    if (aerr == hipErrorOutOfMemory) return kNoRoom;
    HIP_TRY(aerr);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t err = hipEventCreate(&e0);
    if (err == hipSuccess) err = hipEventCreate(&e1);
    if (err == hipSuccess) err = hipEventRecord(e0, st);
    if (err == hipSuccess) err = hipMemcpyAsync(copy, ix->pk, bytes, hipMemcpyDeviceToDevice, st);
    int rc = MEMO_OK;
    if (err == hipSuccess) rc = interleave_words(copy, ix->boff, ix->nb, ix->bshift, ix->packed_fmt, mode, st, ix->d_scratch);
    if (err == hipSuccess && !rc) err = hipEventRecord(e1, st);
    if (err == hipSuccess && !rc) err = hipEventSynchronize(e1);
    float ms = 0.f;
    if (err == hipSuccess && !rc) (void)hipEventElapsedTime(&ms, e0, e1);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (err != hipSuccess || rc) {
        (void)hipStreamSynchronize(st);
        (void)hipFree(copy);
        return rc ? rc : fail(MEMO_EHIP, "ordering the rows: %s", hipGetErrorString(err));
    }
    for (memo_index::DenseView &v : ix->pviews)  // (views are subsets in the old order)
        if (v.p3) {
            const int keep = v.backoff, ask = v.ask_after;  // (not an eviction: the class is as due as it was)
            retire_view(ix, v, false);
            v.backoff = keep;
            v.ask_after = ask;
        }
    retire(ix, ix->pk, bytes);
    ix->pk = copy;
    ix->row_order = mode;
    ix->order_pending = 0;
    const double ns = (double)ms * 1e6 - kPassFixedNs;
    const double per = ns / (double)ix->rows;  // (within a factor of four of the constant, like view_built)
    ix->order_ns_per_row = per > 0.25 * kOrderNsPerRow ? (per < 4.0 * kOrderNsPerRow ? per : 4.0 * kOrderNsPerRow) : 0.25 * kOrderNsPerRow;
    return MEMO_OK;
}

// Is it time to bring the 4-byte rows into the query order?  Rows that came in through the builder or an import are in start
// order; what that costs a sweep depends on how many rows share a start (a half-wave's atomics on one cell: profiles/
// r04_row_order.txt -- 1.5 % of a sweep at 5 rows per start, 7 % at 25, 20 % at 25 and k = 101), and the ordering pass costs
// a dozen sweeps or more: the queries so far must have lost that much (order_due), or memo_index_prepare asks for it.
// WHICH order follows from the kind of query that has paid for the pass: conservation deals a bucket's rows over its starts
// (interleave mode 2), membership over annot mod 32 (mode 3: the lanes of a half-wave on different genomes' plane rows).  Round 4
// had measured the two level on config 4 -- under a planes kernel whose every row waited for the LDS (memo_sweep_memb.hip:
// planes_put); since round 5 it does not, and on the sequence-built index the membership order is 6-7 % ahead at k = 31 and 101
// (config 4: level; profiles/r05_large_k.txt).  Rows that are in the OTHER kind's order (memo_index_pack orders for conservation:
// it cannot know) are re-ordered by the same rule with that gain, each time four times later: kinds that alternate settle.
constexpr double kMembershipOrderGain = 0.06;
static int keep_row_order(memo_index *ix, int64_t window, int km1, bool membership, hipStream_t st) {
    if (!ix->pk || (ix->packed_fmt != 4 && ix->packed_fmt != 12)) return MEMO_OK;
    int want = ix->tune.row_order ? ix->tune.row_order - 1 : (membership ? 3 : kRowOrderDefault);
    if (want == 3 && ix->bshift != 5) want = 2;  // (as order_words_on maps it)
    if (want == ix->row_order) return MEMO_OK;
    const bool pending = ix->order_pending != 0;  // the rows are in start order
    if (!pending && (ix->tune.row_order || ix->row_order == 0)) return MEMO_OK;  // (an order asked for by name, or none wanted: stays)
    if (!g_prepare_only) {
        const double span = (double)ix->max_s - (double)ix->min_s + 1.0, per_start = (double)ix->rows / (span > 1 ? span : 1);
        double gain = 0.003 * per_start * (km1 >= 64 ? 3.0 : 1.0);
        gain = gain > 0.2 ? 0.2 : gain;
        if (!pending && membership) gain = kMembershipOrderGain;
        ix->order_lost_ns += rows_in_window(ix, (double)ix->rows, window, km1) * 1.3 * kSweepNsPerRow * gain;
        const double per_row = ix->order_ns_per_row > 0 ? ix->order_ns_per_row : kOrderNsPerRow;
        const double cost = ((double)ix->rows * per_row + kPassFixedNs) * (double)ix->order_backoff * (double)ix->build_cost_pct / 100.0;
        if (ix->order_lost_ns < cost) return MEMO_OK;
    }
    const int rc = order_words_on(ix, want, st);
    if (rc == kNoRoom) {  // (the pressure may pass: look again, later)
        ix->order_lost_ns = 0;
        if (ix->order_backoff < (1 << 16)) ix->order_backoff *= 4;
        return MEMO_OK;
    }
    if (!rc) {
        ix->order_lost_ns = 0;
        if (!pending && ix->order_backoff < (1 << 16)) ix->order_backoff *= 4;  // (a change of kinds: the next one has to be worth more)
    }
    return rc;
}

// the class of k - 1 = km1 for the 4-byte words: caps in steps of 2 up to 32 (an odd k -- 21, 31 -- gets exactly the rows that
// can write), of 8 up to 64, of 16 up to 128 (twenty-four classes)
static int view_slot(int km1, int *cap) {
    if (km1 <= 32) {
        *cap = 2 * ((km1 + 1) / 2);
        return *cap / 2 - 1;
    }
    if (km1 <= 64) {
        *cap = 8 * ((km1 + 7) / 8);
        return 16 + (*cap - 40) / 8;
    }
    if (km1 <= 128) {
        *cap = 16 * ((km1 + 15) / 16);
        return 20 + (*cap - 80) / 16;
    }
    return -1;
}

// one view: built on `st` between two events, the caller's stream waited for (later queries may come on other streams)
template <typename Build>
static int build_timed(memo_index *ix, memo_index::DenseView &v, hipStream_t st, Build build) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIP_TRY(hipEventCreate(&e0));
    if (hipEventCreate(&e1) != hipSuccess) {
        (void)hipEventDestroy(e0);
        return fail(MEMO_EHIP, "hipEventCreate failed");
    }
    (void)hipEventRecord(e0, st);
    const int rc = build();
    (void)hipEventRecord(e1, st);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&v.build_ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

// The same k-class views for the 4-byte words (formats 4 and 12; what membership queries, k > 64 and indexes of more than 511
// genomes read): the rows whose overlap is below the class's cap (2, 4 ... 32, 40 ... 64, 80 ... 128), with their own bucket table, built
// when it has become worth it (view_due) and spares a fifth of the rows.  BASELINE config 5 at k = 101 sweeps ... its 8.4 * 10^8 rows that way.
int packed_rows_for(memo_index *ix, int km1, int64_t window, bool membership, hipStream_t st, uint32_t **pk, int64_t **boff, uint64_t *rows) {
    if (int rc = keep_row_order(ix, window, km1, membership, st)) return rc;
    *pk = ix->pk;
    *boff = ix->boff;
    *rows = ix->rows;
    ix->last_view_ms = 0.f;
    if (!ix->views_on || ix->tune.no_views || km1 < 1 || !ix->pk || (ix->packed_fmt != 4 && ix->packed_fmt != 12)) return MEMO_OK;
    int cap = 0;
    const int slot = view_slot(km1, &cap);
    if (slot < 0) return MEMO_OK;
    memo_index::DenseView &v = ix->pviews[slot];
    if (v.state == 0 && !view_due(ix, v, 1, (double)ix->rows, 1.0 - share_below(ix, cap), window, km1)) return MEMO_OK;
    if (v.state == 0) {
        DeviceGuard guard(ix->device);
        const int rc = build_timed(ix, v, st, [&]() {
            int r = packed_filter(ix->device, ix->pk, ix->boff, ix->rows, ix->nb, cap, 2, st, ix->packed_fmt == 12 ? 0 : 16, &v.p3, &v.boff,
                                  &v.rows, &v.padded);
            // (what the filter leaves of an interleaved bucket is no longer dealt evenly: the view's buckets are ordered again)
            if (!r && v.p3 && ix->row_order) r = interleave_words(v.p3, v.boff, ix->nb, ix->bshift, ix->packed_fmt, ix->row_order, st, ix->d_scratch);
            return r;
        });
        if (rc && rc != kNoRoom) return rc;  // (no room on the device for a view: the sweep reads all the rows)
        v.cap = cap;
        v.bytes = v.padded * 4;
        v.state = v.p3 ? 1 : (rc == kNoRoom ? 0 : 2);
        if (rc == kNoRoom) {  // (the pressure may pass: look again, but not with every query)
            v.lost_ns = 0;
            v.seen = 0;
            if (v.backoff < (1 << 16)) v.backoff *= 4;
            v.ask_after = v.ask_after ? (v.ask_after < (1 << 16) ? v.ask_after * 4 : v.ask_after) : 16;
        }
        if (v.state == 1) {
            ++ix->view_builds;
            ix->last_view_ms = v.build_ms;
            view_built(ix, v.build_ms, 1, (double)ix->rows);
            keep_views_in_budget(ix, &v, ix->rows * 4, false);
        }
    }
    if (v.state == 1) {
        v.stamp = ++ix->view_clock;
        *pk = v.p3;
        *boff = v.boff;
        *rows = v.rows;
    }
    return MEMO_OK;
}

// The dense rows a conservation / membership sweep with k - 1 = km1 should read: the k-class VIEW that leaves out the rows
// whose overlap is cap or more (cap = 2, 4, 6 ... 32, the smallest that is >= km1: such a row cannot write at this k --
// memo_query.py:49 drops it per query; here it is dropped once per index and class) when that spares a fifth of the rows or
// more, else the dense rows themselves.  A view is built when its class's queries have lost more to its absence than it costs
// (view_due), or by memo_index_prepare, and kept with the index.
// allow_six: the caller can read groups of SIX rows that carry their bucket (the table-driven conservation sweep: 2.67 B per row
// instead of 3.2; profiles/r04_view_levels.txt: -3 % at k = 31, -7 % at k = 17 against five-row views).  Such a view is what the
// library builds where it applies -- buckets of 32 positions, annots of eight bits, overlaps below 32 -- and where the padding of
// every bucket to whole groups (2.5 rows on average) stays small against the bucket; *rpg says which kind was handed out.
int dense_rows_for(memo_index *ix, int km1, int64_t window, hipStream_t st, uint32_t **p3, int64_t **boff, uint64_t *rows, int *view_cap,
                   bool allow_six, int *rpg, bool account) {
    if (view_cap) *view_cap = 0;  // (the cap of the view handed out: its rows are exactly those with overlap < cap)
    if (rpg) *rpg = 5;
    ix->last_view_placed = 0;
    ix->last_view_rpg = 5;
    *p3 = ix->p3;
    *boff = ix->boff3 ? ix->boff3 : ix->boff;
    *rows = ix->boff3 ? ix->rows3 : ix->rows;
    ix->last_view_ms = 0.f;
    if (!ix->views_on || ix->tune.no_views || km1 > 32 || km1 < 1) return MEMO_OK;
    const int slot = (km1 + 1) / 2 - 1, cap = 2 * (slot + 1);  // classes of two: k - 1 <= 2, 4, 6 ... 32 (an odd k: exactly its rows)
    const double src_rows = (double)*rows, kept = (double)ix->rows * share_below(ix, cap);
    const double spared = src_rows > kept ? 1.0 - kept / src_rows : 0.0;
    // six rows per group?  (5-bit starts and overlaps, 8-bit annots; the sweep's form for them has at most five level arrays)
    bool six = allow_six && rpg && ix->bshift == 5 && ix->max_annot <= 255 && km1 <= 31 && ix->view_rows != 5 && g_six_views != 0;
    if (six && ix->view_rows != 6 && g_six_views != 1) six = kept >= 40.0 * (double)(ix->nb > 1 ? ix->nb - 1 : 1);
    memo_index::DenseView *vp = six ? &ix->views6[slot] : &ix->views[slot];
    if (ix->view_rows == 0 && g_six_views < 0) {  // (the library's choice: whichever kind is there already)
        if (vp->state != 1 && allow_six && rpg && ix->bshift == 5 && km1 <= 31 && ix->views6[slot].state == 1) vp = &ix->views6[slot], six = true;
        if (vp->state != 1 && ix->views[slot].state == 1) vp = &ix->views[slot], six = false;
    }
    memo_index::DenseView &v = *vp;
    if (!account) {  // (the same query asking again: what is there, no ledger, no build)
        if (v.state != 1) return MEMO_OK;
    } else if (v.state == 0 && !view_due(ix, v, 0, src_rows, spared, window, km1)) {
        return MEMO_OK;
    }
    const bool can_place = ix->view_places && g_view_colouring != 0;
    const uint32_t *src_p3 = *p3;
    const int64_t *src_boff = *boff;
    const uint64_t nsrc = *rows;
    const uint64_t base_bytes = dense_groups_for(ix->boff3 ? ix->padded3 : ix->padded) * 16;
    const int rpg_arg = six ? 6 : 5;
    if (v.state == 0) {
        DeviceGuard guard(ix->device);
        const bool place = can_place && g_prepare_only;  // (asked for: everything at once; a query: first the view)
        const int rc = build_timed(ix, v, st, [&]() {
            return dense_view_build(ix->device, src_p3, src_boff, nsrc, ix->nb, cap, 2, st, rpg_arg, place ? cap : 0, &v.p3, &v.boff, &v.rows,
                                    &v.padded);
        });
        if (rc && rc != kNoRoom) return rc;  // (no room on the device for a view: the sweep reads all the rows)
        v.cap = cap;
        v.bytes = dense_view_bytes(v.padded, rpg_arg);
        v.state = v.p3 ? 1 : (rc == kNoRoom ? 0 : 2);
        if (rc == kNoRoom) {
            v.lost_ns = 0;
            v.seen = 0;
            if (v.backoff < (1 << 16)) v.backoff *= 4;
            v.ask_after = v.ask_after ? (v.ask_after < (1 << 16) ? v.ask_after * 4 : v.ask_after) : 16;
        }
        if (v.state == 1) {
            ++ix->view_builds;
            v.placed = place ? 1 : 0;
            if (place) ++ix->view_placings;
            ix->last_view_ms = v.build_ms;
            view_built(ix, v.build_ms, place ? 2 : 0, src_rows);
            keep_views_in_budget(ix, &v, base_bytes, true);
        }
    } else if (account && v.state == 1 && !v.placed && can_place && places_due(ix, v, src_rows, window, km1, six)) {
        // the same view again, its rows placed: built beside the one in use (sweeps queued on the caller's other streams may still
        // read that one), then the class switches over and the old copy waits on the retire list with its tile tables
        DeviceGuard guard(ix->device);
        memo_index::DenseView nv;
        const int rc = build_timed(ix, nv, st, [&]() {
            return dense_view_build(ix->device, src_p3, src_boff, nsrc, ix->nb, cap, 0, st, rpg_arg, cap, &nv.p3, &nv.boff, &nv.rows, &nv.padded);
        });
        if (rc && rc != kNoRoom) return rc;
        if (rc == kNoRoom || !nv.p3) {  // (no room for the second copy: the view stays as it is; look again much later)
            v.unplaced_ns = 0;
            if (v.backoff < (1 << 16)) v.backoff *= 4;
        } else {
            const int backoff = v.backoff, ask = v.ask_after;
            retire_view(ix, v, true);
            v = nv;
            v.cap = cap;
            v.bytes = dense_view_bytes(v.padded, rpg_arg);
            v.state = 1;
            v.placed = 1;
            v.backoff = backoff;
            v.ask_after = ask;
            ++ix->view_placings;
            ix->last_view_ms = v.build_ms;
            view_built(ix, v.build_ms, 2, src_rows);
            keep_views_in_budget(ix, &v, base_bytes, true);
        }
    }
    if (v.state == 1) {
        v.stamp = ++ix->view_clock;
        *p3 = v.p3;
        *boff = v.boff;
        *rows = six ? v.padded : v.rows;  // (six rows per group: the places a bucket leaves empty are read and swept like rows)
        if (view_cap) *view_cap = v.cap;
        if (rpg) *rpg = six ? 6 : 5;
        ix->last_view_placed = v.placed;
        ix->last_view_rpg = six ? 6 : 5;
    }
    return MEMO_OK;
}
}  // namespace memo

extern "C" {

int memo_index_export_view(memo_index_t *ix, int32_t k, int32_t rows_per_group, void *groups, int64_t *boff, uint64_t *rows,
                           uint64_t *group_count, int32_t *cap) {
    if (!ix) return fail(MEMO_EINVAL, "index is NULL");
    if (rows_per_group != 5 && rows_per_group != 6) return fail(MEMO_EINVAL, "rows_per_group is 5 or 6");
    if (rows) *rows = 0;
    if (group_count) *group_count = 0;
    if (cap) *cap = 0;
    const int km1 = k - 1;
    if (km1 < 1 || km1 > 32) return MEMO_OK;  // (no class: nothing to export)
    const memo_index::DenseView &v = (rows_per_group == 6 ? ix->views6 : ix->views)[(km1 + 1) / 2 - 1];
    if (v.state != 1 || !v.p3) return MEMO_OK;  // (not built: memo_index_prepare does that)
    const uint64_t ng = rows_per_group == 6 ? v.padded / 6 : (v.rows + 4) / 5;
    if (rows) *rows = v.rows;
    if (group_count) *group_count = ng;
    if (cap) *cap = v.cap;
    if (!groups && !boff) return MEMO_OK;
    if (!groups || !boff) return fail(MEMO_EINVAL, "groups and boff: both or neither");
    int rc;
    if ((rc = download_pipelined(ix->device, groups, v.p3, (size_t)ng * 16, nullptr))) return rc;
    return download_pipelined(ix->device, boff, v.boff, ix->nb * 8, nullptr);
}

}  // extern "C"
