// memo_sweep.hip -- the hot path: MI355X (gfx950 / CDNA4) sweep kernels and their launchers.
//
// Replaces /root/reference/src/memo_query.py:42-63 and :70 (memo_init + memo_query + the argmax
// reduction of print_res).  DESIGN.md section 3 has the algorithm; in short:
//
//   * a workgroup of one or four 64-lane waves owns one TILE of W consecutive pivot positions
//     and never talks to another workgroup;
//   * the rows that can touch the tile are a contiguous slice of the start-sorted index, found
//     with two loads from a bucket table built once per index; they are streamed from HBM with
//     16-byte-per-lane loads, either as the three int64 columns (WideRows) or as the packed
//     4/6-byte rows of memo_index_pack (PackedRows);
//   * conservation: each row's interval, clipped to the tile, is covered by two power-of-two
//     blocks (one ds_min_u32 each into the level-log2 array); the levels are then folded
//     top-down so that level 0 holds min(order) per position;
//   * membership: per-genome bit planes (a row is one run of bits) + an in-register 32 x 32 bit
//     transpose, or the doubling scheme on bit cells;
//   * min / or are idempotent and commutative: overlapping blocks and the arrival order of the
//     atomics cannot change a bit of the result.
//
// Integer work only: no MFMA.  Bound: HBM bandwidth (int64 rows), HBM + LDS atomics (packed rows).
#include <map>
#include <utility>

#include "memo_common.h"

using namespace memo;

namespace {

// ------------------------------------------------------------------------------------------
// kernel arguments
// ------------------------------------------------------------------------------------------
struct SweepArgs {
    const int64_t *s, *e, *o;
    const uint32_t *pk;
    const uint16_t *pa;
    const int64_t *boff;
    int64_t nb;
    int64_t qs, qe;
    int64_t tile0;          // pivot position of tile 0 (multiple of the tile width, <= qs)
    int64_t ntiles;
    int64_t tiles_per_xcd;  // ceil(ntiles / 8)
    int64_t blocks_per_xcd; // workgroups per XCD group; < tiles_per_xcd when workgroups are persistent
    void *out;
    int *status;
    int bshift;
    int km1;    // k - 1 (>= 1 here; k <= 1 never reaches a sweep kernel)
    int ncols;  // result columns: num_docs + 1 (conservation) / num_docs (membership)
    int nlev;   // doubling levels: floor(log2(k-1)) + 1
    int nwords; // membership: 32-bit words per position handled by this launch
    int word_base;  // membership runs: first genome word of this launch (num_docs too large for one
    int out_words;  //   LDS tile is swept in slices of genome words); out_words = words per position
    unsigned long long *stamps;  // diagnostic builds (-DMEMO_STAMPS): per-phase cycle sums
};

// Diagnostic builds only (never in the product library): wave 0 of every workgroup stores the
// shader cycles it spent in each phase of the conservation sweep to stamps[8 * block + phase]
// (a buffer of its own, set with memo_debug_set_stamp_buffer; plain stores, no contention).
#ifdef MEMO_STAMPS
#define MEMO_STAMP(i)                                                                         \
    do {                                                                                      \
        const unsigned long long now__ = __builtin_amdgcn_s_memtime();                        \
        if (threadIdx.x == 0 && A.stamps) A.stamps[8ull * blockIdx.x + (i)] = now__ - stamp_t0; \
        stamp_t0 = now__;                                                                     \
    } while (0)
#else
#define MEMO_STAMP(i) do { } while (0)
#endif

// blockIdx -> tile.  Blocks are dealt round-robin over the 8 XCDs (b % 8 labels the XCD
// group), so give each group one contiguous run of tiles: neighbouring tiles share their
// k-1 halo rows and the cache lines that straddle the tile boundary, and those then hit in
// that XCD's L2 instead of being fetched twice.  Speed only -- results do not depend on it.
// A persistent workgroup (blocks_per_xcd < tiles_per_xcd) walks its XCD group's run with stride
// blocks_per_xcd: `it` is its iteration.  Returns -1 past the end of the run.
__device__ __forceinline__ int64_t tile_of_block(const SweepArgs &A, int it) {
    const int64_t b = blockIdx.x;
    const int64_t j = (b >> 3) + (int64_t)it * A.blocks_per_xcd;
    return j < A.tiles_per_xcd ? (b & 7) * A.tiles_per_xcd + j : -1;
}

// Row slice [r0, r1) that can touch positions [lo_abs, hi_abs) of a tile starting at a:
// rows with  a <= start < roundup(hi_abs + k - 1, bucket).
__device__ __forceinline__ void row_slice(const SweepArgs &A, int64_t a, int64_t hi_abs,
                                          uint64_t &r0, uint64_t &r1) {
    const int64_t last = A.nb - 1;
    int64_t b0 = a <= 0 ? 0 : (a >> A.bshift);
    const int64_t lim = hi_abs + A.km1;  // rows with start >= lim cannot reach the tile
    int64_t b1 = lim <= 0 ? 0 : ((lim + ((int64_t)1 << A.bshift) - 1) >> A.bshift);
    b0 = b0 > last ? last : b0;
    b1 = b1 > last ? last : b1;
    r0 = a <= 0 ? 0 : (uint64_t)A.boff[b0];
    r1 = (uint64_t)A.boff[b1];
}

__device__ __forceinline__ int clamp_to_tile(int64_t v, int lo, int hi) {
    const int64_t l = lo, h = hi;
    return (int)(v < l ? l : (v > h ? h : v));
}

// ------------------------------------------------------------------------------------------
// shared pieces of the sweep kernels.  T = threads per workgroup (64 = one wave owns the tile;
// 256 = four waves share it and meet at workgroup barriers between the phases).
// ------------------------------------------------------------------------------------------
#ifndef MEMO_KU
#define MEMO_KU 4
#endif

// Level arrays of the conservation sweep are W + kLevelSkew words apart: rows that hit the same
// position on different levels then fall into different LDS banks.
#ifndef MEMO_SKEW
#define MEMO_SKEW 0
#endif
constexpr int kLevelSkew = MEMO_SKEW;

struct Tile {
    int64_t a;     // pivot position of tile slot 0
    int x_lo, x_hi;  // slots of the tile that lie inside the window
    uint64_t r0, r1;  // row slice
};

template <int W>
__device__ __forceinline__ bool locate_tile(const SweepArgs &A, Tile &t, int it) {
    const int64_t tile = tile_of_block(A, it);
    if (tile < 0 || tile >= A.ntiles) return false;
    t.a = A.tile0 + tile * W;
    t.x_lo = (int)(A.qs > t.a ? A.qs - t.a : 0);
    t.x_hi = (int)(A.qe - t.a < W ? A.qe - t.a : W);
    row_slice(A, t.a, t.a + t.x_hi, t.r0, t.r1);
    if (t.r1 - t.r0 >= 0xFFFF0000ull) {  // the row loops count a tile's rows in 32 bits
        if (threadIdx.x == 0) atomicOr(A.status, kStatusHugeSlice);
        return false;
    }
    return true;
}

// Workgroup barrier for LDS hand-offs that leaves global loads in flight: __syncthreads() would
// wait for vmcnt(0) first (cdna_hip_programming.md, "Pipelining across barriers").
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Row sources.  Each streams the tile's row slice once and hands f(c, h, col) the rows that
// write: [c, h) = the row's interval clipped to the tile (memo_query.py:46-49: recentre,
// shadow-cast by k-1, clip, keep casted_end < start), col = its column after the index check
// of :62 (NumPy/Numba wrap a negative index once; anything else outside the matrix is the
// reference's IndexError / UB and sets the sticky status flag).
__device__ __forceinline__ bool check_col(const SweepArgs &A, int64_t o, int &col) {
    const int64_t cc = o < 0 ? o + A.ncols : o;
    if ((uint64_t)cc >= (uint64_t)A.ncols) {
        atomicOr(A.status, kStatusBadAnnot);
        return false;
    }
    col = (int)cc;
    return true;
}

// the Parquet columns as they are: 3 x int64 per row.  2 rows per lane per column per load
// (16 B / lane, 1 KiB / wave), U loads of each column in flight per lane.
struct WideRows {
    static constexpr int kLoads = MEMO_KU;  // loads of each column in flight per lane
    // `between` runs once, in every thread, before any row is handed to f: the kernels clear their
    // LDS tile there.  PackedRows issues its first batch of loads before it; here (ten batches per
    // tile, HBM-bound) that ordering measured 5 % slower, so the tile is cleared first.
    template <int T, int U, typename B, typename F>
    static __device__ __forceinline__ void for_each(const SweepArgs &A, const Tile &t, B between, F f) {
        between();
        const int tid = threadIdx.x;
        auto one = [&](int64_t s, int64_t e, int64_t o) {
            const int h = clamp_to_tile(s - t.a, t.x_lo, t.x_hi);
            const int c = clamp_to_tile(e - t.a - A.km1, t.x_lo, t.x_hi);
            int col;
            // end < start: the row may reach further left than k-1 positions; long_rows_kernel owns it
            if (h > c && e >= s && check_col(A, o, col)) f(c, h, col);
        };
        // 32-bit row numbers relative to the 128-byte-aligned start of the slice
        const uint64_t base0 = t.r0 & ~(uint64_t)15;
        const uint32_t end = (uint32_t)(t.r1 - base0);
        const int64_t *ps = A.s + base0, *pe = A.e + base0, *po = A.o + base0;
        for (uint32_t rel = 2 * tid; rel < end; rel += 2 * T * U) {
            longlong2 S[U], E[U], O[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t r = rel + (uint32_t)u * 2 * T;
                if (r < end) {
                    S[u] = *reinterpret_cast<const longlong2 *>(ps + r);
                    E[u] = *reinterpret_cast<const longlong2 *>(pe + r);
                    O[u] = *reinterpret_cast<const longlong2 *>(po + r);
                } else {
                    S[u] = make_longlong2(kSentinel, kSentinel);
                    E[u] = S[u];
                    O[u] = make_longlong2(0, 0);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                // wave-uniform: nothing of this wave's load is inside the slice
                if ((uint32_t)__builtin_amdgcn_readfirstlane(rel + (uint32_t)u * 2 * T) >= end) break;
                one(S[u].x, E[u].x, O[u].x);
                one(S[u].y, E[u].y, O[u].y);
            }
        }
    }
};

// clamp(v, lo, hi) for lo <= hi in one instruction; hi is wave-uniform (one SGPR operand is all a
// gfx9 VALU instruction may read), lo is a VGPR pinned by pin_vgpr() so that it is not
// re-materialised from its SGPR before every use
__device__ __forceinline__ int med3(int v, int lo, int hi) {
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(lo), "s"(hi));
    return r;
}

__device__ __forceinline__ int pin_vgpr(int uniform) {
    int r;
    asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "s"(uniform));
    return r;
}

// packed rows (memo_index_pack): one 32-bit word per row -- start mod 2^16, min(end - start,
// 255), annot (8 bits; ANNOT16: in a second 16-bit column).  Inside a row slice every start
// lies in [a, a + W + k + 32), far less than 2^16 from the tile start, so the low 16 bits
// give the tile-relative start exactly; rows outside [r0, r1) are masked by index.  Exact
// for k - 1 <= 255: a saturated length clips to "does not write" just as the true one does.
// 4 rows per lane per load (16 B / lane).
// CHECKED = false is chosen by the host when the largest annot of the index (known since
// memo_index_pack) is inside the result matrix, so that no row can raise the reference's
// IndexError; the column test then leaves the loop.
template <bool ANNOT16, bool CHECKED>
struct PackedRows {
    static constexpr int kLoads = 2 * MEMO_KU;  // A/B: 8 x 16 B in flight per lane, 5 % over 4
    template <int T, int U, typename B, typename F>
    static __device__ __forceinline__ void for_each(const SweepArgs &A, const Tile &t, B between, F f) {
        const int tid = threadIdx.x;
        const uint32_t a16 = (uint32_t)t.a & 0xFFFFu;
        // 32-bit row numbers relative to the 128-byte-aligned start of the slice
        const uint64_t base0 = t.r0 & ~(uint64_t)31;
        const uint32_t first = (uint32_t)(t.r0 - base0), end = (uint32_t)(t.r1 - base0);
        const uint32_t *pk = A.pk + base0;
        const uint16_t *pa = ANNOT16 ? A.pa + base0 : nullptr;
        const int x_lo = pin_vgpr(t.x_lo), x_hi = t.x_hi, km1 = A.km1;
        const uint32_t ncols = (uint32_t)A.ncols;
        // a row that cannot write: start == a, overlap 255 >= k - 1  ->  c >= h
        const uint32_t dead = a16 | 0x00FF0000u;
        uint32_t bad = 0;
        auto one = [&](uint32_t w, uint32_t annot) {
            const int d = (int)((w - a16) & 0xFFFFu);  // start - a
            int h = med3(d, x_lo, x_hi);
            const int c = med3(d + (int)__builtin_amdgcn_ubfe(w, 16, 8) - km1, x_lo, x_hi);
            if (CHECKED && annot >= ncols) {
                bad |= (uint32_t)(h > c);
                h = c;
            }
            f(c, h, (int)annot);  // f writes iff h > c
        };
        bool first_batch = true;
        for (uint32_t rel = 4 * tid; first_batch || rel < end; rel += 4 * T * U) {
            uint4 V[U];
            uint2 N[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t r = rel + (uint32_t)u * 4 * T;
                if (r < end) {
                    V[u] = *reinterpret_cast<const uint4 *>(pk + r);
                    if (ANNOT16) N[u] = *reinterpret_cast<const uint2 *>(pa + r);
                } else {
                    V[u] = make_uint4(dead, dead, dead, dead);
                    N[u] = make_uint2(0u, 0u);
                }
            }
            if (first_batch) {
                between();
                first_batch = false;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t r = rel + (uint32_t)u * 4 * T;
                // wave-uniform: only a load that straddles an end of the slice masks rows by number
                const uint32_t wave_lo = __builtin_amdgcn_readfirstlane(r) & ~(uint32_t)255;
                if (wave_lo >= end) break;  // nothing of this wave's load is inside the slice
                if (!(wave_lo >= first && wave_lo + 256 <= end)) {
                    V[u].x = (r + 0 >= first && r + 0 < end) ? V[u].x : dead;
                    V[u].y = (r + 1 >= first && r + 1 < end) ? V[u].y : dead;
                    V[u].z = (r + 2 >= first && r + 2 < end) ? V[u].z : dead;
                    V[u].w = (r + 3 >= first && r + 3 < end) ? V[u].w : dead;
                }
                one(V[u].x, ANNOT16 ? (N[u].x & 0xFFFFu) : (V[u].x >> 24));
                one(V[u].y, ANNOT16 ? (N[u].x >> 16) : (V[u].y >> 24));
                one(V[u].z, ANNOT16 ? (N[u].y & 0xFFFFu) : (V[u].z >> 24));
                one(V[u].w, ANNOT16 ? (N[u].y >> 16) : (V[u].w >> 24));
            }
        }
        if (CHECKED && bad) atomicOr(A.status, kStatusBadAnnot);
    }
};

__device__ __forceinline__ uint32_t full_word(int ncols, int w) {  // genomes 32w .. 32w+31 that exist
    const int left = ncols - 32 * w;
    return left >= 32 ? 0xFFFFFFFFu : (left <= 0 ? 0u : ((1u << left) - 1u));
}

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
    bool have = locate_tile<W>(A, t, 0);
    for (int it = 0; have; ++it) {
    // a persistent workgroup looks its next tile up now: the two bucket-table loads (2-4k cycles
    // when HBM is busy) then return under this tile's work instead of in front of the next one's
    Tile t_next;
    const bool have_next = locate_tile<W>(A, t_next, it + 1);
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
    for (int j = A.nlev - 1; j >= 1; --j) {
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
    // write level 0 as OutT (uint16, or uint8 when num_docs <= 255), in 16-byte pieces aligned
    // in the OUTPUT (the tile grid is aligned in pivot coordinates, the output starts at qs)
    constexpr int PER = 16 / (int)sizeof(OutT);  // positions per 16-byte store
    OutT *out = static_cast<OutT *>(A.out);
    const int64_t ob = t.a - A.qs;  // output index of tile slot 0
    const int64_t o_lo = ob + t.x_lo, o_hi = ob + t.x_hi;
    for (int64_t g = (o_lo & ~(int64_t)(PER - 1)) + PER * tid; g < o_hi; g += PER * T) {
        const int x = (int)(g - ob);
        if (g >= o_lo && g + PER <= o_hi) {
            uint32_t pk[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (sizeof(OutT) == 2)
                    pk[q] = lds[x + 2 * q] | (lds[x + 2 * q + 1] << 16);
                else
                    pk[q] = lds[x + 4 * q] | (lds[x + 4 * q + 1] << 8) | (lds[x + 4 * q + 2] << 16) |
                            (lds[x + 4 * q + 3] << 24);
            }
            *reinterpret_cast<uint4 *>(out + g) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        } else {
            for (int i = 0; i < PER; ++i)
                if (g + i >= o_lo && g + i < o_hi) out[g + i] = (OutT)lds[x + i];
        }
    }
    MEMO_STAMP(5);  // store
#ifdef MEMO_STAMPS
    if (threadIdx.x == 0 && A.stamps && it == 0) A.stamps[8ull * blockIdx.x + 7] = 1;
#endif
    if (have_next) __syncthreads();  // the LDS tile is reused
    t = t_next;
    have = have_next;
    }
}


// ------------------------------------------------------------------------------------------
// membership.  Result word w of position x:  full_word(w) & ~absent[x][w].
//   DOUBLING = false: one ds_or per covered (position, genome) bit into absent[x][w].
//   DOUBLING = true : the same two-blocks-per-row scatter and top-down fold as conservation,
//                     on cells of nw words (or instead of min); nlev * W * nw words of LDS.
// ------------------------------------------------------------------------------------------
template <int T>
__device__ __forceinline__ void store_membership(const SweepArgs &A, const Tile &t,
                                                 const uint32_t *absent, int nw) {
    // slots [x_lo, x_hi) are one contiguous run of words in LDS and in the output
    uint32_t *out = static_cast<uint32_t *>(A.out);
    const int tid = threadIdx.x;
    const int64_t ob = (t.a - A.qs) * nw;  // output word of LDS word 0
    const int64_t o_lo = ob + (int64_t)t.x_lo * nw, o_hi = ob + (int64_t)t.x_hi * nw;
    for (int64_t g = (o_lo & ~(int64_t)3) + 4 * tid; g < o_hi; g += 4 * T) {
        const int x = (int)(g - ob);
        uint32_t v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool in = g + i >= o_lo && g + i < o_hi;
            v[i] = in ? (full_word(A.ncols, (x + i) % nw) & ~absent[x + i]) : 0u;
        }
        if (g >= o_lo && g + 4 <= o_hi) {
            *reinterpret_cast<uint4 *>(out + g) = make_uint4(v[0], v[1], v[2], v[3]);
        } else {
            for (int i = 0; i < 4; ++i)
                if (g + i >= o_lo && g + i < o_hi) out[g + i] = v[i];
        }
    }
}

template <typename Rows, int W, int U, int T, bool DOUBLING>
__global__ __launch_bounds__(T) void sweep_membership_kernel(const SweepArgs A) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int tid = threadIdx.x;
    Tile t;
    const int nw = A.nwords;
    const int nlev = DOUBLING ? A.nlev : 1;
    const int plane = W * nw;  // words per level
    bool have = locate_tile<W>(A, t, 0);
    for (int it = 0; have; ++it) {
    Tile t_next;  // persistent workgroups: next tile's bucket-table loads fly under this tile's work
    const bool have_next = locate_tile<W>(A, t_next, it + 1);

    auto clear_tile = [&]() {
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
        uint4 *p = reinterpret_cast<uint4 *>(lds);
        for (int i = tid; i < nlev * plane / 4; i += T) p[i] = z;
        lds_barrier();
    };
    Rows::template for_each<T, U>(A, t, clear_tile, [&](int c, int h, int col) {
        if (h <= c) return;
        const uint32_t bit = 1u << (col & 31);
        const int word = col >> 5;
        if (DOUBLING) {
            const int j = 31 - __builtin_clz((unsigned)(h - c));  // h - c >= 1
            uint32_t *lv = lds + j * plane + word;
            atomicOr(lv + c * nw, bit);
            atomicOr(lv + (h - (1 << j)) * nw, bit);
        } else {
            uint32_t *cell = lds + c * nw + word;
            for (int x = c; x < h; ++x, cell += nw) atomicOr(cell, bit);  // rec[c:h, a] = False
        }
    });
    __syncthreads();

    if (DOUBLING) {
        for (int j = nlev - 1; j >= 1; --j) {
            const int shift = (1 << (j - 1)) * nw;  // half a block, in words
            const uint32_t *hi = lds + j * plane;
            uint32_t *lo = lds + (j - 1) * plane;
            if ((shift & 3) == 0) {
                for (int i = 4 * tid; i < plane; i += 4 * T) {
                    const uint4 v = *reinterpret_cast<const uint4 *>(hi + i);
                    const uint4 u = i >= shift ? *reinterpret_cast<const uint4 *>(hi + i - shift)
                                               : make_uint4(0u, 0u, 0u, 0u);
                    uint4 w = *reinterpret_cast<const uint4 *>(lo + i);
                    w.x |= v.x | u.x;
                    w.y |= v.y | u.y;
                    w.z |= v.z | u.z;
                    w.w |= v.w | u.w;
                    *reinterpret_cast<uint4 *>(lo + i) = w;
                }
            } else {
                for (int i = tid; i < plane; i += T)
                    lo[i] |= hi[i] | (i >= shift ? hi[i - shift] : 0u);
            }
            __syncthreads();
        }
    }
    store_membership<T>(A, t, lds, nw);
    if (have_next) __syncthreads();  // the LDS tile is reused
    t = t_next;
    have = have_next;
    }
}

// ------------------------------------------------------------------------------------------
// membership, "runs" form: bit planes per GENOME instead of per position.
//   absent[g][P] (P = position / 32) holds 32 positions of genome g per word, so a row's interval
//   [c, h) is one run of bits: one ds_or_b32 when it stays inside a word, two when it straddles
//   (more only for k - 1 > 32), and rows of different genomes never share a word.  No levels, no
//   fold; 4 * W * nw bytes of LDS.  Each lane then transposes 32 genomes x 32 positions in
//   registers (5 butterfly stages) into the position-major result words and stores them.
// ------------------------------------------------------------------------------------------
template <int J>
__device__ __forceinline__ void transpose32_stage(uint32_t (&m)[32]) {
    constexpr uint32_t mask = J == 16 ? 0x0000FFFFu : J == 8 ? 0x00FF00FFu : J == 4 ? 0x0F0F0F0Fu
                              : J == 2 ? 0x33333333u : 0x55555555u;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        if ((k & J) == 0) {  // swap the high J-bit halves of m[k] with the low halves of m[k + J]
            const uint32_t tt = ((m[k] >> J) ^ m[k + J]) & mask;
            m[k] ^= tt << J;
            m[k + J] ^= tt;
        }
    }
}

__device__ __forceinline__ void transpose32(uint32_t (&m)[32]) {  // m[j] bit i  <-  m[i] bit j
    transpose32_stage<16>(m);
    transpose32_stage<8>(m);
    transpose32_stage<4>(m);
    transpose32_stage<2>(m);
    transpose32_stage<1>(m);
}

template <typename Rows, int W, int U, int T>
__global__ __launch_bounds__(T) void sweep_membership_runs_kernel(const SweepArgs A) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int tid = threadIdx.x;
    constexpr int PW = W / 32;  // words per genome row
    Tile t;
    const int nw = A.nwords;
    bool have = locate_tile<W>(A, t, 0);
    for (int it = 0; have; ++it) {
    Tile t_next;  // persistent workgroups: next tile's bucket-table loads fly under this tile's work
    const bool have_next = locate_tile<W>(A, t_next, it + 1);
    // genome g lives at g * PITCH + (g >> 5) * skew.  PITCH is odd, so the scatter's bank is
    // (genome + word) mod 32 -- with a pitch of PW (a multiple of 32) every genome would land on
    // the banks of its position word alone.  In the transpose phase 32 lanes read word P of genome
    // groups G = 0..nw-1 for 32 / nw consecutive P: skew = 32 / nw puts them on 32 different banks.
    constexpr int PITCH = PW + 1;
    const int skew = A.nlev;  // membership runs: the launcher passes the skew in nlev
    const int total = 32 * nw * PITCH + nw * skew;
    auto clear_tile = [&]() {
        for (int i = tid; i < total; i += T) lds[i] = 0;
        lds_barrier();
    };
    const int g_lo = 32 * A.word_base, g_n = 32 * nw;  // genomes of this launch's slice
    Rows::template for_each<T, U>(A, t, clear_tile, [&](int c, int h, int col) {
        col -= g_lo;
        if (h <= c || (unsigned)col >= (unsigned)g_n) return;
        uint32_t *row = lds + col * PITCH + (col >> 5) * skew;
        const int w0 = c >> 5, w1 = (h - 1) >> 5;
        const uint32_t first = 0xFFFFFFFFu << (c & 31), last = 0xFFFFFFFFu >> (31 - ((h - 1) & 31));
        atomicOr(row + w0, w0 == w1 ? first & last : first);  // one instruction for both shapes
        if (w1 > w0) {
            for (int w = w0 + 1; w < w1; ++w) atomicOr(row + w, 0xFFFFFFFFu);
            atomicOr(row + w1, last);
        }
    });
    __syncthreads();

    uint32_t *out = static_cast<uint32_t *>(A.out);
    const int64_t ob = t.a - A.qs;  // output position of tile slot 0
    for (int b = tid; b < nw * PW; b += T) {
        const int G = b % nw, P = b / nw;  // genome group, position word
        if (32 * P + 32 <= t.x_lo || 32 * P >= t.x_hi) continue;
        uint32_t m[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) m[i] = lds[(32 * G + i) * PITCH + G * skew + P];
        transpose32(m);
        const uint32_t full = full_word(A.ncols, A.word_base + G);
        const int64_t ow = A.out_words;
        uint32_t *dst = out + (ob + 32 * P) * ow + A.word_base + G;
        if (32 * P >= t.x_lo && 32 * P + 32 <= t.x_hi) {
#pragma unroll
            for (int j = 0; j < 32; ++j) dst[(int64_t)j * ow] = full & ~m[j];
        } else {
#pragma unroll
            for (int j = 0; j < 32; ++j)
                if (32 * P + j >= t.x_lo && 32 * P + j < t.x_hi) dst[(int64_t)j * ow] = full & ~m[j];
        }
    }
    if (have_next) __syncthreads();  // the LDS tile is reused
    t = t_next;
    have = have_next;
    }
}


// k <= 1: no row can write (casted_end >= start always), every position keeps its initial value
template <typename OutT>
__global__ void fill_conservation_kernel(OutT *out, int64_t n, OutT v) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        out[i] = v;
}

__global__ void fill_membership_kernel(uint32_t *out, int64_t n, int nw, int ncols) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int left = ncols - 32 * (int)(i % nw);
        out[i] = left >= 32 ? 0xFFFFFFFFu : ((1u << left) - 1u);
    }
}

// One workgroup per row with end < start: its interval [clip(e-qs-(k-1)), clip(s-qs)) can be any
// length, so it is applied straight to the result in HBM, after the sweep, with atomics (rows may
// overlap each other).  filter_pq keeps such a row iff qs < start < qe + k (memo_query.py:25-27).
template <typename OutT>
__global__ void long_rows_conservation_kernel(const int64_t *ls, const int64_t *le, const int64_t *lo,
                                              int64_t qs, int64_t qe, int km1, int ncols, OutT *out,
                                              int *status) {
    const int64_t s = ls[blockIdx.x], e = le[blockIdx.x], o = lo[blockIdx.x];
    if (!(s > qs && s < qe + km1 + 1)) return;
    const int64_t L = qe - qs;
    const int64_t hi = s - qs > L ? L : s - qs;
    int64_t c = e - qs - km1;
    c = c < 0 ? 0 : c;
    if (c >= hi) return;
    const int64_t cc = o < 0 ? o + ncols : o;
    if ((uint64_t)cc >= (uint64_t)ncols) {
        if (threadIdx.x == 0) atomicOr(status, kStatusBadAnnot);
        return;
    }
    constexpr int PER = 4 / (int)sizeof(OutT);  // results per 32-bit word
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

__global__ void long_rows_membership_kernel(const int64_t *ls, const int64_t *le, const int64_t *lo,
                                            int64_t qs, int64_t qe, int km1, int ncols, int nw,
                                            uint32_t *out, int *status) {
    const int64_t s = ls[blockIdx.x], e = le[blockIdx.x], o = lo[blockIdx.x];
    if (!(s > qs && s < qe + km1 + 1)) return;
    const int64_t L = qe - qs;
    const int64_t hi = s - qs > L ? L : s - qs;
    int64_t c = e - qs - km1;
    c = c < 0 ? 0 : c;
    if (c >= hi) return;
    const int64_t cc = o < 0 ? o + ncols : o;
    if ((uint64_t)cc >= (uint64_t)ncols) {
        if (threadIdx.x == 0) atomicOr(status, kStatusBadAnnot);
        return;
    }
    const uint32_t keep = ~(1u << (cc & 31));
    for (int64_t p = c + threadIdx.x; p < hi; p += blockDim.x) atomicAnd(out + p * nw + (cc >> 5), keep);
}


// ------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------
int g_tile_w = 0;     // 0 = choose per query
int g_waves = 0;      // waves per tile: 0 = choose, 1 or 4
int g_memb_algo = 0;  // membership: 0 = choose, 1 = direct scatter, 2 = doubling
int g_force_wide = 0; // 1 = read the int64 columns even when packed rows exist
unsigned long long *g_stamp_buffer = nullptr;  // -DMEMO_STAMPS builds: 8 words per workgroup
int g_persist = 0;    // 0 = choose, 1 = one workgroup per tile, 2 = persistent workgroups

// Persistent workgroups measured 7-20 % SLOWER on every workload (profiles/r01_persistent_ab.txt):
// resident workgroups that start together stay in step -- every CU loads, then every CU folds --
// whereas one workgroup per tile staggers them as earlier ones retire, which is what overlaps the
// memory phase of one tile with the LDS phase of another.  Kept as an A/B switch only.
bool use_persistent(int /*fmt*/) { return g_persist == 2; }
bool g_env_read = false;

void read_env_once() {
    if (g_env_read) return;
    g_env_read = true;
    if (const char *v = getenv("MEMO_TILE_W")) g_tile_w = atoi(v);
    if (const char *v = getenv("MEMO_WAVES")) g_waves = atoi(v);
    if (const char *v = getenv("MEMO_MEMB_ALGO")) g_memb_algo = atoi(v);
    if (const char *v = getenv("MEMO_ROWS")) g_force_wide = strcmp(v, "wide") == 0;
    if (const char *v = getenv("MEMO_PERSIST")) g_persist = atoi(v);
}

int floor_log2(uint32_t v) { return 31 - __builtin_clz(v); }

using SweepKernel = void (*)(const SweepArgs);

// tiles are aligned in pivot coordinates: tile 0 starts at floor(qs / w) * w.
// persistent: launch only as many workgroups as the device keeps resident; each walks its XCD
// group's run of tiles and looks the next tile up while it works on the current one.
int launch_tiles(SweepKernel kernel, SweepArgs &A, int w, int threads, size_t lds, hipStream_t st,
                 bool persistent) {
    const int sh = floor_log2((uint32_t)w);
    A.tile0 = (A.qs >> sh) << sh;  // >> on a negative int64 is arithmetic: floor
    A.ntiles = ((A.qe - A.tile0) + w - 1) >> sh;
    A.tiles_per_xcd = (A.ntiles + 7) / 8;
    A.blocks_per_xcd = A.tiles_per_xcd;
    if (lds > 160 * 1024) return fail(MEMO_EINVAL, "tile needs %zu bytes of LDS (> 160 KiB)", lds);
    if (lds > 64 * 1024) {  // opt in to large dynamic LDS once per (thread, device, kernel, size)
        thread_local std::map<std::pair<const void *, int>, size_t> granted;
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        size_t &have = granted[{reinterpret_cast<const void *>(kernel), dev}];
        if (have < lds) {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            have = lds;
        }
    }
    if (persistent) {
        int per_cu = 0, cus = 0, dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(kernel),
                                                             threads, lds));
        const int64_t resident = ((int64_t)cus * (per_cu > 0 ? per_cu : 1) + 7) / 8;  // per XCD group
        if (resident < A.blocks_per_xcd) A.blocks_per_xcd = resident;
    }
    if (A.blocks_per_xcd * 8 * threads >= ((int64_t)1 << 32))
        return fail(MEMO_EINVAL, "window too long for one launch at tile width %d", w);
    hipLaunchKernelGGL(kernel, dim3((unsigned)(A.blocks_per_xcd * 8)), dim3(threads), lds, st, A);
    HIP_TRY(hipGetLastError());
    return MEMO_OK;
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

template <typename Rows>
SweepKernel memb_kernel(int w, int waves, bool doubling) {
#define MEMO_CASE(WW)                                                                              \
    case WW:                                                                                       \
        if (doubling)                                                                              \
            return waves == 4 ? (SweepKernel)sweep_membership_kernel<Rows, WW, Rows::kLoads, 256, true>      \
                              : (SweepKernel)sweep_membership_kernel<Rows, WW, Rows::kLoads, 64, true>;      \
        return waves == 4 ? (SweepKernel)sweep_membership_kernel<Rows, WW, Rows::kLoads, 256, false>         \
                          : (SweepKernel)sweep_membership_kernel<Rows, WW, Rows::kLoads, 64, false>;
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

template <typename Rows>
SweepKernel memb_runs_kernel(int w, int waves) {
#define MEMO_CASE(WW)                                                                                \
    case WW:                                                                                         \
        return waves == 4 ? (SweepKernel)sweep_membership_runs_kernel<Rows, WW, Rows::kLoads, 256>   \
                          : (SweepKernel)sweep_membership_runs_kernel<Rows, WW, Rows::kLoads, 64>;
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

// which row source a query reads: packed when the index has it and k - 1 <= 255 (MEMO_ROWS=wide
// forces the int64 columns), else the int64 columns
int pick_rows(const memo_index *ix, int32_t k, int &fmt) {
    fmt = 0;
    if (ix->packed_fmt && k - 1 <= 255 && !(g_force_wide && ix->has_wide)) fmt = ix->packed_fmt;
    if (!fmt && !ix->has_wide)
        return fail(MEMO_EINVAL, "k = %d needs the int64 columns, which this index dropped when it was packed", k);
    return MEMO_OK;
}

int check_query_args(const memo_index *ix, int64_t qs, int64_t qe, int32_t k, int32_t num_docs,
                     const void *d_out) {
    if (!ix) return fail(MEMO_EINVAL, "index is NULL");
    if (!ix->finalized) return fail(MEMO_ENOTREADY, "index not finalized");
    if (num_docs < 1 || num_docs > 65534)
        return fail(MEMO_EINVAL, "num_docs must be in [1, 65534], got %d", num_docs);
    if (qs <= -kCoordLimit || qe >= kCoordLimit || qs >= kCoordLimit || qe <= -kCoordLimit)
        return fail(MEMO_EINVAL, "window coordinates out of range");
    if (k >= (1 << 30) || k <= -(1 << 30)) return fail(MEMO_EINVAL, "k out of range");
    if (qe < qs)  // np.zeros([true_len, ...]) with true_len < 0 (memo_query.py:51,53)
        return fail(MEMO_EINVAL, "ValueError: negative dimensions are not allowed (window end < start)");
    if (qe > qs && !d_out) return fail(MEMO_EINVAL, "output pointer is NULL");
    if (qe > qs && ((uintptr_t)d_out & 15)) return fail(MEMO_EINVAL, "output must be 16-byte aligned");
    if (qe - qs > ((int64_t)1 << 40)) return fail(MEMO_EINVAL, "window longer than 2^40");
    return MEMO_OK;
}

void fill_args(const memo_index *ix, SweepArgs &A, int64_t qs, int64_t qe, int32_t k, void *d_out) {
    A.s = ix->s;
    A.e = ix->e;
    A.o = ix->o;
    A.pk = ix->pk;
    A.pa = ix->pa;
    A.boff = ix->boff;
    A.nb = (int64_t)ix->nb;
    A.qs = qs;
    A.qe = qe;
    A.out = d_out;
    A.status = ix->d_status;
    A.stamps = g_stamp_buffer;
    A.bshift = ix->bshift;
    A.km1 = k - 1;
}

}  // namespace

template <typename OutT>
static int long_rows_conservation(const memo_index *ix, int64_t qs, int64_t qe, int32_t k, int ncols,
                                  OutT *d_out, hipStream_t st) {
    if (!ix->n_long) return MEMO_OK;
    hipLaunchKernelGGL((long_rows_conservation_kernel<OutT>), dim3((unsigned)ix->n_long), dim3(256), 0, st,
                       ix->ls, ix->le, ix->lo, qs, qe, k - 1, ncols, d_out, ix->d_status);
    HIP_TRY(hipGetLastError());
    return MEMO_OK;
}

static int long_rows_membership(const memo_index *ix, int64_t qs, int64_t qe, int32_t k, int ncols, int nw,
                                uint32_t *d_out, hipStream_t st) {
    if (!ix->n_long) return MEMO_OK;
    hipLaunchKernelGGL(long_rows_membership_kernel, dim3((unsigned)ix->n_long), dim3(256), 0, st, ix->ls,
                       ix->le, ix->lo, qs, qe, k - 1, ncols, nw, d_out, ix->d_status);
    HIP_TRY(hipGetLastError());
    return MEMO_OK;
}

template <typename OutT>
static int query_conservation(memo_index_t *ix, int64_t qs, int64_t qe, int32_t k, int32_t num_docs,
                              OutT *d_out, void *stream) {
    read_env_once();
    int rc = check_query_args(ix, qs, qe, k, num_docs, d_out);
    if (rc) return rc;
    if (sizeof(OutT) == 1 && num_docs > 255)
        return fail(MEMO_EINVAL, "uint8 results need num_docs <= 255, got %d", num_docs);
    if (qe <= qs) return MEMO_OK;
    DeviceGuard guard(ix->device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (k <= 1 || ix->rows == 0) {
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
    // Tile shape, from interleaved A/B on one device (profiles/r01_ab_*.txt).
    //  int64 rows (HBM-bound): four waves share a 4096-position tile -- fewest k-1 row halos per
    //    position; 1-3 % over one wave per 1024 positions at k <= 32, 10 % at k = 101.
    //  packed rows (4-6x fewer bytes; LDS-atomic / issue-bound): waves per CU matter, but so does
    //    the k-1 halo: 1024 positions x 4 waves wins at k = 31 (20 KiB, 8 workgroups per CU) and at
    //    k = 101 (28 KiB) over 512 or 2048 positions.
    // Short windows want many small tiles either way.
    int w = g_tile_w, waves = g_waves == 1 || g_waves == 4 ? g_waves : 0;
    if (!w) {
        // int64 rows on a sparse index (< 2 rows per position: profiles/r01_sparse_index_tiles.txt) are
        // no longer HBM-bound per tile; they want the packed rows' shape (more workgroups per CU)
        const double span = (double)(ix->max_s - ix->min_s) + 1.0;
        const bool sparse = (double)ix->rows < 2.0 * span;
        const size_t budget = (fmt || sparse) ? 32 * 1024 : 80 * 1024;
        w = 4096;
        while ((size_t)A.nlev * w * 4 > budget && w > 256) w >>= 1;
        while (w > 256 && (qe - qs) / w < 32768) w >>= 1;
    }
    if (!waves) waves = w >= 1024 ? 4 : 1;  // short windows end up with small tiles: one wave each
    while ((size_t)A.nlev * w * 4 > 160 * 1024 && w > 256) w >>= 1;
    const bool checked = ix->max_annot >= (uint64_t)A.ncols;  // some row could be outside the matrix
    SweepKernel kern = fmt == 4   ? (checked ? cons_kernel<PackedRows<false, true>, OutT>(w, waves)
                                             : cons_kernel<PackedRows<false, false>, OutT>(w, waves))
                       : fmt == 6 ? (checked ? cons_kernel<PackedRows<true, true>, OutT>(w, waves)
                                             : cons_kernel<PackedRows<true, false>, OutT>(w, waves))
                                  : cons_kernel<WideRows, OutT>(w, waves);
    if (!kern) return fail(MEMO_EINVAL, "unsupported tile width %d", w);
    if ((rc = launch_tiles(kern, A, w, 64 * waves, (size_t)A.nlev * (w + kLevelSkew) * 4, st, use_persistent(fmt)))) return rc;
    return long_rows_conservation<OutT>(ix, qs, qe, k, A.ncols, d_out, st);
}

extern "C" {

int memo_debug_set_stamp_buffer(uint64_t *d_buffer) {
    g_stamp_buffer = reinterpret_cast<unsigned long long *>(d_buffer);
    return MEMO_OK;
}

int memo_set_persistent(int32_t mode) {
    read_env_once();
    if (mode < 0 || mode > 2) return fail(MEMO_EINVAL, "mode must be 0 (choose), 1 (off) or 2 (on)");
    g_persist = mode;
    return MEMO_OK;
}

int memo_set_row_source(int32_t source) {
    read_env_once();
    if (source != 0 && source != 1) return fail(MEMO_EINVAL, "source must be 0 (packed when present) or 1 (int64 columns)");
    g_force_wide = source;
    return MEMO_OK;
}

int memo_set_tuning(int32_t tile_w, int32_t waves, int32_t membership_algo) {
    read_env_once();
    if (tile_w != 0 && tile_w != 256 && tile_w != 512 && tile_w != 1024 && tile_w != 2048 &&
        tile_w != 4096)
        return fail(MEMO_EINVAL, "tile_w must be 0, 256, 512, 1024, 2048 or 4096");
    if (waves != 0 && waves != 1 && waves != 4) return fail(MEMO_EINVAL, "waves must be 0, 1 or 4");
    if (membership_algo < 0 || membership_algo > 3) return fail(MEMO_EINVAL, "membership_algo must be 0..3");
    g_tile_w = tile_w;
    g_waves = waves;
    g_memb_algo = membership_algo;
    return MEMO_OK;
}

int memo_query_conservation_dev(memo_index_t *ix, int64_t qs, int64_t qe, int32_t k,
                                int32_t num_docs, uint16_t *d_out, void *stream) {
    return query_conservation<uint16_t>(ix, qs, qe, k, num_docs, d_out, stream);
}

int memo_query_conservation_u8_dev(memo_index_t *ix, int64_t qs, int64_t qe, int32_t k,
                                   int32_t num_docs, uint8_t *d_out, void *stream) {
    return query_conservation<uint8_t>(ix, qs, qe, k, num_docs, d_out, stream);
}

int memo_query_membership_dev(memo_index_t *ix, int64_t qs, int64_t qe, int32_t k,
                              int32_t num_docs, uint32_t *d_out, void *stream) {
    read_env_once();
    int rc = check_query_args(ix, qs, qe, k, num_docs, d_out);
    if (rc) return rc;
    if (qe <= qs) return MEMO_OK;
    DeviceGuard guard(ix->device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nw = (num_docs + 31) / 32;
    if (k <= 1 || ix->rows == 0) {
        hipLaunchKernelGGL(fill_membership_kernel, dim3(2048), dim3(256), 0, st, d_out,
                           (qe - qs) * nw, nw, num_docs);
        HIP_TRY(hipGetLastError());
        return long_rows_membership(ix, qs, qe, k, num_docs, nw, d_out, st);
    }
    SweepArgs A;
    fill_args(ix, A, qs, qe, k, d_out);
    A.ncols = num_docs;
    A.nlev = floor_log2((uint32_t)(k - 1)) + 1;
    A.nwords = nw;
    int fmt;
    if ((rc = pick_rows(ix, k, fmt))) return rc;
    // algorithm: 3 = runs (bit planes per genome + register transpose), 2 = doubling, 1 = direct
    const size_t per_pos_doubling = (size_t)A.nlev * nw * 4;
    int algo = g_memb_algo;
    // A/B on config 4 (profiles/r01_membership_algorithms.txt): packed rows 0.87 ms runs vs 1.13 ms
    // doubling; int64 rows (HBM-bound either way) 2.52 ms doubling vs 2.64 ms runs
    if (!algo) algo = (fmt || per_pos_doubling * 256 > 40 * 1024) ? 3 : 2;
    // whatever was asked for: a tile of 256 positions has to fit in LDS, else runs (which can slice)
    if ((algo == 2 ? per_pos_doubling : (size_t)nw * 4) * 256 > 128 * 1024 || nw > 64) algo = 3;
    const bool checked = ix->max_annot >= (uint64_t)A.ncols;
    int w = g_tile_w, waves = g_waves == 1 || g_waves == 4 ? g_waves : 0;
    A.word_base = 0;
    A.out_words = nw;
    if (algo == 3) {
        if (!waves) waves = 4;
        // 4 * nw bytes of LDS per position: beyond 2048 genomes even a 256-position tile is too big,
        // so the genome words are swept in slices of 64 (the rows are read once per slice; every
        // slice writes its own words of the result)
        const int slice = nw <= 64 ? nw : 64;
        if (!w) {  // a lane transposes one 32 x 32 block: keep nw * W / 32 >= threads
            w = 4096;
            while ((size_t)slice * 4 * w > 32 * 1024 && w > 256) w >>= 1;
            while (w > 256 && (qe - qs) / w < 16384) w >>= 1;
        }
        int skew = 1;
        while (skew * 2 * slice <= 32) skew *= 2;  // largest power of two <= 32 / nw (1 when nw > 16)
        A.nlev = skew;
        auto lds_bytes = [&](int ww) { return ((size_t)32 * slice * (ww / 32 + 1) + (size_t)slice * skew) * 4; };
        while (lds_bytes(w) > 160 * 1024 && w > 256) w >>= 1;
        const size_t lds = lds_bytes(w);
        SweepKernel kern = fmt == 4   ? (checked ? memb_runs_kernel<PackedRows<false, true>>(w, waves)
                                                 : memb_runs_kernel<PackedRows<false, false>>(w, waves))
                           : fmt == 6 ? (checked ? memb_runs_kernel<PackedRows<true, true>>(w, waves)
                                                 : memb_runs_kernel<PackedRows<true, false>>(w, waves))
                                      : memb_runs_kernel<WideRows>(w, waves);
        if (!kern) return fail(MEMO_EINVAL, "unsupported tile width %d", w);
        for (int base = 0; base < nw; base += slice) {
            A.word_base = base;
            A.nwords = nw - base < slice ? nw - base : slice;
            if ((rc = launch_tiles(kern, A, w, 64 * waves, lds, st, use_persistent(fmt)))) return rc;
        }
        return long_rows_membership(ix, qs, qe, k, A.ncols, nw, d_out, st);
    }
    const bool doubling = algo == 2;
    const size_t per_pos = doubling ? per_pos_doubling : (size_t)nw * 4;
    if (!waves) waves = doubling ? 4 : 1;
    if (!w) {  // config 4 A/B: int64 rows 512 positions x 4 waves (40 KiB); packed rows 256 x 4 (20 KiB)
        const size_t budget = (waves == 4 ? (fmt ? 20u : 40u) : 20u) * 1024;
        w = 4096;
        while (per_pos * w > budget && w > 256) w >>= 1;
        while (w > 256 && (qe - qs) / w < 16384) w >>= 1;
    }
    while (per_pos * w > 160 * 1024 && w > 256) w >>= 1;
    SweepKernel kern = fmt == 4   ? (checked ? memb_kernel<PackedRows<false, true>>(w, waves, doubling)
                                             : memb_kernel<PackedRows<false, false>>(w, waves, doubling))
                       : fmt == 6 ? (checked ? memb_kernel<PackedRows<true, true>>(w, waves, doubling)
                                             : memb_kernel<PackedRows<true, false>>(w, waves, doubling))
                                  : memb_kernel<WideRows>(w, waves, doubling);
    if (!kern) return fail(MEMO_EINVAL, "unsupported tile width %d", w);
    if ((rc = launch_tiles(kern, A, w, 64 * waves, per_pos * w, st, use_persistent(fmt)))) return rc;
    return long_rows_membership(ix, qs, qe, k, A.ncols, nw, d_out, st);
}

int memo_query_check(memo_index_t *ix, void *stream) {
    if (!ix) return fail(MEMO_EINVAL, "index is NULL");
    DeviceGuard guard(ix->device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    int flags = 0;
    HIP_TRY(hipMemcpyAsync(&flags, ix->d_status, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (flags) {
        HIP_TRY(hipMemsetAsync(ix->d_status, 0, sizeof(int), st));
        HIP_TRY(hipStreamSynchronize(st));
        if (flags & kStatusHugeSlice)
            return fail(MEMO_EINVAL, "more than 2^32 index rows can reach one tile of the window: unsupported");
        if (flags & kStatusBadAnnot)
            return fail(MEMO_EINVAL,
                        "a row that covers the window has an order/genome column outside the "
                        "result matrix (num_docs too small?) -- the reference raises IndexError here");
        return fail(MEMO_EINVAL, "device status 0x%x", flags);
    }
    return MEMO_OK;
}

}  // extern "C"
