// memo_sweep.h -- shared by the sweep translation units (memo_sweep.hip: launch plumbing and tuning;
// memo_sweep_cons.hip: conservation; memo_sweep_memb.hip: membership).  Device code here is inline.
#ifndef MEMO_SWEEP_H
#define MEMO_SWEEP_H

#include "memo_common.h"

namespace memo {

// ------------------------------------------------------------------------------------------
// kernel arguments
// ------------------------------------------------------------------------------------------
struct SweepArgs {
    const int64_t *s, *e, *o;
    const uint32_t *pk;
    const uint16_t *pa;
    const uint32_t *p3;     // dense rows (PackedRows3): 16 bytes per 5 rows
    const int64_t *boff;
    int64_t nb;
    int64_t bbase;          // bucket of boff[0]
    int64_t qs, qe;
    int64_t tile0;          // pivot position of tile 0 (multiple of the tile width, <= qs)
    int64_t ntiles;
    int64_t tiles_per_xcd;  // ceil(ntiles / 8)
    int x_lo_first, x_hi_last;  //   ... and the window's edges inside its first / last tile
    int64_t tile_abs0;      // table-driven dense sweep (memo_sweep_cons3t.hip): tile 0's number in pivot coordinates (tile0 / w),
    const void *ttab;       //   the tile table (32 bytes per tile) and its length
    int64_t ntab;
    void *out;
    int *status;
    int bshift;
    int km1;    // k - 1 (>= 1 here; k <= 1 never reaches a sweep kernel)
    int ncols;  // result columns: num_docs + 1 (conservation) / num_docs (membership)
    int nlev;   // doubling levels: floor(log2(k-1)) + 1.  Membership bit planes reuse it: runs kernel = words
                //   of skew per 32-genome group; planes kernel = words of halo left of the tile
    int nwords; // membership: 32-bit words per position handled by this launch
    int ls, hl;  // conservation, unclipped scatter: words per level array, words of left halo
                 //   (membership planes kernel: words per genome's plane row, words of skew per 32-genome group)
    int w;       //   ... and the tile width of both (a multiple of the bucket width, not a template parameter there)
    uint32_t magic;  // membership planes: ceil(2^32 / (32 * nwords)), for q / (32 * nwords) by v_mul_hi
    uint32_t lvmask; // conservation, mixed level arrays (memo_sweep_cons.hip: level_plan): which arrays exist / are populated
    int ftop;        //   ... and clz of the largest populated block size (the range's slot 0)
    int word_base;  // membership runs: first genome word of this launch (num_docs too large for one
    int out_words;  //   LDS tile is swept in slices of genome words); out_words = words per position
    unsigned long long *stamps;  // diagnostic builds (-DMEMO_STAMPS): per-phase cycle sums
};

// Four results in one store, at whatever address the window's start makes of it: a window that begins off the 4-position
// raster stores 4 / 8 bytes at an address that is a multiple of the element size only.  Global memory takes that on gfx950
// (unaligned access mode, what HSA asks of the global segment) at the same speed -- config 3 from position 1 or 2: 0.185 ms,
// as from 0; rounds 1-4 sent such windows to scalar stores or a slower kernel (0.295; profiles/r04_unaligned_windows.txt).
// The types tell the compiler; the stores still come out as one global_store_dword / global_store_dwordx2 each.
typedef uint32_t __attribute__((aligned(1))) u32_any;
typedef uint64_t __attribute__((aligned(2))) u64_any;
__device__ __forceinline__ void store_four(uint8_t *p, uint32_t v) { *reinterpret_cast<u32_any *>(p) = v; }
__device__ __forceinline__ void store_four(uint16_t *p, uint32_t lo, uint32_t hi) {
    *reinterpret_cast<u64_any *>(p) = (uint64_t)lo | ((uint64_t)hi << 32);
}

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
// (One workgroup per tile.  Persistent workgroups walking a run of tiles measured 7-20 % slower on
// every workload, profiles/r01_persistent_ab.txt, and are gone.)
__device__ __forceinline__ int64_t tile_of_block(const SweepArgs &A) {
    const int64_t b = blockIdx.x;
    return (b & 7) * A.tiles_per_xcd + (b >> 3);
}

// Row slice [r0, r1) that can touch positions [lo_abs, hi_abs) of a tile starting at a:
// rows with  a <= start < roundup(hi_abs + k - 1, bucket).
__device__ __forceinline__ void row_slice(const SweepArgs &A, int64_t a, int64_t hi_abs,
                                          uint64_t &r0, uint64_t &r1) {
    const int64_t last = A.nb - 1;
    int64_t b0 = a <= 0 ? 0 : (a >> A.bshift) - A.bbase;  // (the table of a region slice starts at bucket bbase:
    const int64_t lim = hi_abs + A.km1;  // rows with start >= lim cannot reach the tile        nothing lies before it)
    int64_t b1 = lim <= 0 ? 0 : ((lim + ((int64_t)1 << A.bshift) - 1) >> A.bshift) - A.bbase;
    b0 = b0 < 0 ? 0 : (b0 > last ? last : b0);
    b1 = b1 < 0 ? 0 : (b1 > last ? last : b1);
#ifdef MEMO_SYNTH_LOCATE
    // diagnostic build only: the bucket table of the synthetic config-3 index in closed form (5 rows per
    // position, start_i = 1 + i / 5) -- what would the sweep gain if a tile's row slice cost no memory access?
    (void)last;
    auto first_row = [&](int64_t pos) { return pos <= 1 ? (int64_t)0 : 5 * (pos - 1); };
    const int64_t rows_total = A.boff ? (int64_t)MEMO_SYNTH_LOCATE : 0;
    int64_t q0 = first_row((b0 + A.bbase) << A.bshift), q1 = first_row((b1 + A.bbase) << A.bshift);
    q0 = q0 > rows_total ? rows_total : q0;
    q1 = q1 > rows_total ? rows_total : q1;
    r0 = a <= 0 ? 0 : (uint64_t)q0;
    r1 = (uint64_t)q1;
#else
    r0 = a <= 0 ? 0 : (uint64_t)A.boff[b0];  // (rows with a negative start lie before bucket 0)
    r1 = (uint64_t)A.boff[b1];
#endif
}

// The branch-free row blocks (v_cmpx ... s_mov_b64 exec, -1) are right only when every lane is enabled on entry, which
// the compiler is never told.  Two guards: a scan of the shipped code (tests/test_host_cpu.py::
// test_row_blocks_run_with_every_lane_enabled) and, in -DMEMO_EXEC_CHECK builds of the AB library (tools/build_variant.sh
// execcheck -DMEMO_EXEC_CHECK; the GPU tier and the fuzzer run on it once per round), this test in front of every block.
#ifdef MEMO_EXEC_CHECK
#define MEMO_EXEC_ALL_ONES(status)                                                              \
    do {                                                                                        \
        if (__builtin_amdgcn_read_exec() != ~0ull) atomicOr((status), memo::kStatusExecNarrow); \
    } while (0)
#else
#define MEMO_EXEC_ALL_ONES(status) do { } while (0)
#endif

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

__device__ __forceinline__ bool locate_tile_w(const SweepArgs &A, Tile &t, int W) {
    const int64_t tile = tile_of_block(A);
    if (tile >= A.ntiles) return false;
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

template <int W>
__device__ __forceinline__ bool locate_tile(const SweepArgs &A, Tile &t) {
    return locate_tile_w(A, t, W);
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

// packed rows (memo_index_pack): one 32-bit word per row, in one of three layouts
//     annot <= 255   start mod 2^16 | min(end - start, 255) << 16 | annot << 24                  (format 4)
//     annot <= 4095  min(end - start, 255) | (start mod 2^12) << 8 | annot << 20      W12       (format 12)
//     else           the first word with annot 0, the annot in a second, 16-bit column  ANNOT16  (format 6)
// Inside a row slice every start lies in [a, a + W + k + 32), far less than 2^12 from the tile start
// (tiles of at most 2048 positions with W12), so the start field gives the tile-relative start exactly;
// rows outside [r0, r1) are masked by index.  Exact for k - 1 <= 255: a saturated length clips to "does
// not write" just as the true one does.  4 rows per lane per load (16 B / lane).  The byte-aligned fields
// of format 4 cost one instruction each (16-bit subtract, SDWA byte operand); format 12 keeps the length
// in byte 0 and pays one more for the start (subtract + bit-field extract) -- against format 6's second
// load per four rows it is 33 % fewer bytes (BASELINE config 5: 500 genomes).
// CHECKED = false is chosen by the host when the largest annot of the index (known since
// memo_index_pack) is inside the result matrix, so that no row can raise the reference's
// IndexError; the column test then leaves the loop.
template <bool ANNOT16, bool CHECKED, bool W12 = false>
struct PackedRows {
    static_assert(!(ANNOT16 && W12), "a 12-bit annot rides in the word");
    static constexpr int kLoads = 2 * MEMO_KU;  // A/B: 8 x 16 B in flight per lane, 5 % over 4
    static constexpr bool kAnnot16 = ANNOT16;
    static constexpr bool kW12 = W12;
    static constexpr int kTopShift = W12 ? 20 : 24;  // where the annot sits when it rides in the word
    // what a tile subtracts from a row's start field
    static __device__ __forceinline__ uint32_t tile_key(int64_t a) {
        return W12 ? ((uint32_t)a & 0xFFFu) << 8 : (uint32_t)a & 0xFFFFu;
    }
    // start - a, for a row of the tile's slice (key = tile_key(a), in a VGPR)
    static __device__ __forceinline__ uint32_t rel_start(uint32_t w, uint32_t key) {
        if (W12) return __builtin_amdgcn_ubfe(w - key, 8, 12);  // (a borrow into the annot field does not reach the 12 bits)
        uint32_t d;  // gfx9 16-bit VALU results have a zero high half
        asm("v_sub_u16 %0, %1, %2" : "=v"(d) : "v"(w), "v"(key));
        return d;
    }
    static __device__ __forceinline__ int len(uint32_t w) {  // min(end - start, 255)
        return W12 ? (int)(w & 0xFFu) : (int)__builtin_amdgcn_ubfe(w, 16, 8);
    }
    static __device__ __forceinline__ uint32_t word_annot(uint32_t w) { return W12 ? w >> 20 : w >> 24; }
    // f(c, h, col): the clipped interval, as WideRows hands it out
    template <int T, int U, typename B, typename F>
    static __device__ __forceinline__ void for_each(const SweepArgs &A, const Tile &t, B between, F f) {
        const uint32_t key = (uint32_t)pin_vgpr((int)tile_key(t.a));
        const int x_lo = pin_vgpr(t.x_lo), x_hi = t.x_hi, km1 = A.km1;
        const uint32_t ncols = (uint32_t)A.ncols;
        uint32_t bad = 0;
        for_each_raw<T, U>(A, t, between, [&](uint32_t w, uint32_t annot) {
            const int d = (int)rel_start(w, key);  // start - a
            int h = med3(d, x_lo, x_hi);
            const int c = med3(d + len(w) - km1, x_lo, x_hi);
            if (CHECKED && annot >= ncols) {
                bad |= (uint32_t)(h > c);
                h = c;
            }
            f(c, h, (int)annot);  // f writes iff h > c
        });
        if (CHECKED && bad) atomicOr(A.status, kStatusBadAnnot);
    }

    // g(word, annot): every row of the slice as it is stored; the rows a 16-byte load holds outside
    // the slice arrive as a word that cannot write (start == a, overlap 255 >= k - 1)
    template <int T, int U, typename B, typename G>
    static __device__ __forceinline__ void for_each_raw(const SweepArgs &A, const Tile &t, B between, G g) {
        const uint32_t nb = batches<T, U>(t);
        for (uint32_t b = 0; b == 0 || b < nb; ++b) {
            uint4 V[U];
            uint2 N[U];
            issue<T, U>(A, t, b, V, N);
            if (b == 0) between();
            consume<T, U>(A, t, b, V, N, g);
        }
    }

    // The two halves of a batch (U loads of 16 B per lane), for kernels that put other work between
    // issuing a tile's loads and using them.  Row numbers are 32-bit, relative to the 128-byte-aligned
    // start of the slice.
    template <int T, int U>
    static __device__ __forceinline__ uint32_t batches(const Tile &t) {
        const uint32_t end = (uint32_t)(t.r1 - (t.r0 & ~(uint64_t)31));
        return (end + 4 * T * U - 1) / (4 * T * U);
    }

    template <int T, int U>
    static __device__ __forceinline__ void issue(const SweepArgs &A, const Tile &t, uint32_t batch, uint4 (&V)[U],
                                                 uint2 (&N)[U]) {
        const uint64_t base0 = t.r0 & ~(uint64_t)31;
        const uint32_t end = (uint32_t)(t.r1 - base0);
        const uint32_t *pk = A.pk + base0;
        const uint16_t *pa = ANNOT16 ? A.pa + base0 : nullptr;
        const uint32_t rel = batch * (4 * T * U) + 4 * threadIdx.x;
        const uint32_t wave_rel = __builtin_amdgcn_readfirstlane(rel) & ~(uint32_t)255;  // once: the rest is scalar arithmetic
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t r = rel + (uint32_t)u * 4 * T;
            // wave-uniform (consume() tests the same): a wave loads its 256 rows or nothing.  The rows
            // it reads past the slice lie inside the index or its kPadRows sentinel rows.
            if (wave_rel + (uint32_t)u * 4 * T < end) {
                V[u] = *reinterpret_cast<const uint4 *>(pk + r);
                if (ANNOT16) N[u] = *reinterpret_cast<const uint2 *>(pa + r);
            }
        }
    }

    template <int T, int U, typename G>
    static __device__ __forceinline__ void consume(const SweepArgs &A, const Tile &t, uint32_t batch, uint4 (&V)[U],
                                                   uint2 (&N)[U], G g) {
        const uint64_t base0 = t.r0 & ~(uint64_t)31;
        const uint32_t first = (uint32_t)(t.r0 - base0), end = (uint32_t)(t.r1 - base0);
        const uint32_t dead = tile_key(t.a) | (W12 ? 0x000000FFu : 0x00FF0000u);  // start = a, length 255: never writes
        const uint32_t rel = batch * (4 * T * U) + 4 * threadIdx.x;
        const uint32_t wave_rel = __builtin_amdgcn_readfirstlane(rel) & ~(uint32_t)255;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t r = rel + (uint32_t)u * 4 * T;
            // wave-uniform: only a load that straddles an end of the slice masks rows by number
            const uint32_t wave_lo = wave_rel + (uint32_t)u * 4 * T;
            if (wave_lo >= end) break;  // nothing of this wave's load is inside the slice
            if (!(wave_lo >= first && wave_lo + 256 <= end)) {
                V[u].x = (r + 0 >= first && r + 0 < end) ? V[u].x : dead;
                V[u].y = (r + 1 >= first && r + 1 < end) ? V[u].y : dead;
                V[u].z = (r + 2 >= first && r + 2 < end) ? V[u].z : dead;
                V[u].w = (r + 3 >= first && r + 3 < end) ? V[u].w : dead;
            }
            g(V[u].x, ANNOT16 ? (N[u].x & 0xFFFFu) : word_annot(V[u].x));
            g(V[u].y, ANNOT16 ? (N[u].x >> 16) : word_annot(V[u].y));
            g(V[u].z, ANNOT16 ? (N[u].y & 0xFFFFu) : word_annot(V[u].z));
            g(V[u].w, ANNOT16 ? (N[u].y >> 16) : word_annot(V[u].w));
        }
    }
};

// Dense rows (memo_index_pack_dense): start mod 2^10, min(end - start, 63), annot (8 bits) -- 24 bits per
// row, FIVE rows per 16-byte group (3.2 B per row), one aligned global_load_dwordx4 per lane and group:
//     dword j = B_j | X_j << 16 | A_j << 24   (j = 0 .. 3)      B = (start & 1023) << 6 | min(end - start, 63)
//     X_0 = B_4 & 255, X_1 = B_4 >> 8, X_2 = A_4, X_3 = 0         A = annot
// Rows 0 .. 3 are used as they are loaded: the (start, length) field is the low half of the dword, where the 16-bit
// VALU forms reach it, and the annot its top byte -- the dword itself is what ds_min_u32 takes (order in place, as
// in the 4-byte format: what lies below the top byte only breaks ties).  Row 4 is put together from the spare
// bytes: one v_perm_b32 for its field, one shift for its annot.  (First layout of this format: fields in both
// halves of dwords 0 .. 2, the annots as bytes of dwords 2 and 3 -- four shifts per five rows and SDWA forms for
// the high halves; this one takes 0.4 VALU instructions per row less.)
// The start lives in the TOP ten bits of B: (B - (a & 1023) << 6) mod 2^16 leaves the length alone and gives
// (start - a) mod 2^10 with no borrow to repair.
// Exact for k - 1 <= 63 (a saturated length clips to "does not write" just as the true one does) in kernels
// whose row slice spans fewer than 2^10 positions (the unclipped conservation sweep, level arrays <= 1024 cells).
// History (profiles/r02_dense_rows_ab.txt): 12 bytes per 4 rows fetched with global_load_dwordx3 ran 21 %
// slower than the 4-byte rows; two planes (16 + 8 bytes per 8 rows, two loads) and five rows per 16 bytes are
// 13 % faster than the 4-byte rows back to back.
struct PackedRows3 {
    static constexpr int kLoads = 6;          // 16-byte loads in flight per lane: 30 rows
    static constexpr bool kAnnot16 = false;
    static constexpr uint32_t kWaveRows = 320;

    // the slice in units of groups: [g0, g1), g0 a multiple of 8 (128 bytes)
    static __device__ __forceinline__ void span(const Tile &t, uint64_t &g0, uint32_t &ngroups, uint32_t &first,
                                                uint32_t &end) {
        g0 = (t.r0 / 5) & ~(uint64_t)7;
        const uint64_t g1 = (t.r1 + 4) / 5;
        ngroups = (uint32_t)(g1 - g0);
        first = (uint32_t)(t.r0 - 5 * g0);
        end = (uint32_t)(t.r1 - 5 * g0);
    }

    template <int T, int U>
    static __device__ __forceinline__ uint32_t batches(const Tile &t) {
        uint64_t g0;
        uint32_t ng, first, end;
        span(t, g0, ng, first, end);
        return (ng + T * U - 1) / (T * U);
    }

    template <int T, int U>
    static __device__ __forceinline__ void issue(const SweepArgs &A, const Tile &t, uint32_t batch, uint4 (&V)[U]) {
        uint64_t g0;
        uint32_t ng, first, end;
        span(t, g0, ng, first, end);
        const uint4 *p = reinterpret_cast<const uint4 *>(A.p3) + g0;
        const uint32_t q0 = batch * (T * U) + threadIdx.x;
        const uint32_t wave_q0 = __builtin_amdgcn_readfirstlane(q0) & ~(uint32_t)63;  // once: the rest is scalar arithmetic
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // wave-uniform (consume() tests the same): a wave loads its 64 groups or nothing
            if (wave_q0 + (uint32_t)u * T < ng) V[u] = p[q0 + (uint32_t)u * T];
        }
    }

    // g(b, data): the row's B field is the low half of b (the high half is other rows' business), data a word with
    // its annot in the top byte.  Rows outside [r0, r1) get the dead field (start = a, length 63: never writes when
    // k - 1 <= 63).
    // A9 (indexes of 256 .. 511 genomes): data carries the order in its top NINE bits -- (ninth annot bit, from bit 16 + i of the
    // group's last dword : the dword) >> 1
    template <int T, int U, typename G, bool A9 = false>
    static __device__ __forceinline__ void consume(const SweepArgs &A, const Tile &t, uint32_t batch, uint4 (&V)[U], G g) {
        uint64_t g0;
        uint32_t ng, first, end;
        span(t, g0, ng, first, end);
        const uint32_t dead = ((((uint32_t)t.a & 1023u) << 6) | 63u);
        const uint32_t q0 = batch * (T * U) + threadIdx.x;
        const uint32_t wave_q0 = __builtin_amdgcn_readfirstlane(q0) & ~(uint32_t)63;
        // a wave's load lies wholly inside the slice iff its first group is in [in_lo, in_hi] (scalar, once per batch)
        const uint32_t in_lo = (first + 4) / 5;
        const int32_t in_hi = end >= kWaveRows ? (int32_t)((end - kWaveRows) / 5) : -1;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t q = q0 + (uint32_t)u * T;
            const uint32_t wave_q = wave_q0 + (uint32_t)u * T;
            if (wave_q >= ng) break;  // nothing of this wave's load is inside the slice
            // (the fields are overwritten in place -- V[u] is dead after this -- so that the common path needs no copies)
            if (wave_q < in_lo || (int32_t)wave_q > in_hi) {  // a load that straddles an end of the slice
                const uint32_t r = 5 * q;
                auto in = [&](uint32_t i) { return r + i >= first && r + i < end; };
                if (!in(4)) {  // row 4's field: byte 2 of dwords 0 and 1
                    V[u].x = (V[u].x & 0xFF00FFFFu) | ((dead & 0xFFu) << 16);
                    V[u].y = (V[u].y & 0xFF00FFFFu) | ((dead >> 8) << 16);
                }
                if (!in(0)) V[u].x = (V[u].x & 0xFFFF0000u) | dead;
                if (!in(1)) V[u].y = (V[u].y & 0xFFFF0000u) | dead;
                if (!in(2)) V[u].z = (V[u].z & 0xFFFF0000u) | dead;
                if (!in(3)) V[u].w = (V[u].w & 0xFFFF0000u) | dead;
            }
            if constexpr (A9) {
                const uint32_t hi = V[u].w >> 16;
                g(V[u].x, __builtin_amdgcn_alignbit(hi, V[u].x, 1));
                g(V[u].y, __builtin_amdgcn_alignbit(hi >> 1, V[u].y, 1));
                g(V[u].z, __builtin_amdgcn_alignbit(hi >> 2, V[u].z, 1));
                g(V[u].w, __builtin_amdgcn_alignbit(hi >> 3, V[u].w, 1));
                g(__builtin_amdgcn_perm(V[u].y, V[u].x, 0x0c0c0602u), __builtin_amdgcn_alignbit(hi >> 4, V[u].z << 8, 1));
                continue;
            }
            g(V[u].x, V[u].x);
            g(V[u].y, V[u].y);
            g(V[u].z, V[u].z);
            g(V[u].w, V[u].w);
            g(__builtin_amdgcn_perm(V[u].y, V[u].x, 0x0c0c0602u), V[u].z << 8);  // row 4: B_4 from the spare bytes, A_4 to the top
        }
    }
};

// Conservation results of one tile, LDS -> HBM, with the last fold on the way.  lv0[x] = level 0 of
// tile slot x (blocks of one position), lv1 = level 1 (blocks of two: lv1[x] covers x and x + 1) or
// NULL when k - 1 = 1; the result is min(lv0[x], lv1[x], lv1[x - 1]), lv1 being readable from index
// x_min on.  A lane takes 4 neighbouring slots (ds_read_b128 x 2 + one word; the 64 lanes of a wave
// read 1 KiB in a row, no bank conflicts -- 16 slots per lane would put the lanes 16 words apart, 4
// banks for the whole wave) and stores them as 4 or 8 bytes: 256-512 contiguous bytes per
// wave-instruction.  Pieces are aligned in the OUTPUT (the tile grid is aligned in pivot
// coordinates, the output starts at qs); when that leaves the LDS side unaligned the cells are read
// one by one.
// TOP != 0: the cells hold whole row words whose top bits (from bit TOP: 24 or 20) are the order (the unclipped
// kernels min the words as they are -- the junk below the order only breaks ties); the result is those bits.
template <typename OutT, int T, int TOP = 0>
__device__ __forceinline__ void store_conservation(const SweepArgs &A, const Tile &t, const uint32_t *lv0,
                                                   const uint32_t *lv1, int x_min) {
    OutT *out = static_cast<OutT *>(A.out);
    const int64_t ob = t.a - A.qs;  // output index of tile slot 0
    const int64_t o_lo = ob + t.x_lo, o_hi = ob + t.x_hi;
    const bool aligned = (ob & 3) == 0;
    auto one = [&](int x) {
        uint32_t r = lv0[x];
        if (lv1) {
            r = min(r, lv1[x]);
            if (x > x_min) r = min(r, lv1[x - 1]);
        }
        return TOP ? r >> TOP : r;
    };
    for (int64_t g = (o_lo & ~(int64_t)3) + 4 * threadIdx.x; g < o_hi; g += 4 * T) {
        const int x = (int)(g - ob);
        if (g >= o_lo && g + 4 <= o_hi) {
            uint4 v;
            if (aligned) {
                v = *reinterpret_cast<const uint4 *>(lv0 + x);
                if (lv1) {
                    const uint4 u = *reinterpret_cast<const uint4 *>(lv1 + x);
                    const uint32_t left = x > x_min ? lv1[x - 1] : ~0u;
                    v.x = min(v.x, min(u.x, left));
                    v.y = min(v.y, min(u.y, u.x));
                    v.z = min(v.z, min(u.z, u.y));
                    v.w = min(v.w, min(u.w, u.z));
                }
                if (TOP && TOP != 24) v = make_uint4(v.x >> TOP, v.y >> TOP, v.z >> TOP, v.w >> TOP);
                if (TOP == 24) {  // byte 3 of each word -> the packed result, one v_perm_b32 per two words
                    if (sizeof(OutT) == 1) {
                        const uint32_t lo = __builtin_amdgcn_perm(v.y, v.x, 0x0C0C0703u);  // [x3, y3, 0, 0]
                        const uint32_t hi = __builtin_amdgcn_perm(v.w, v.z, 0x07030C0Cu);  // [0, 0, z3, w3]
                        *reinterpret_cast<uint32_t *>(out + g) = lo | hi;
                    } else {
                        *reinterpret_cast<uint2 *>(out + g) = make_uint2(__builtin_amdgcn_perm(v.y, v.x, 0x0C070C03u),
                                                                         __builtin_amdgcn_perm(v.w, v.z, 0x0C070C03u));
                    }
                    continue;
                }
            } else {
                v = make_uint4(one(x), one(x + 1), one(x + 2), one(x + 3));
            }
            if (sizeof(OutT) == 1)
                *reinterpret_cast<uint32_t *>(out + g) = v.x | (v.y << 8) | (v.z << 16) | (v.w << 24);
            else
                *reinterpret_cast<uint2 *>(out + g) = make_uint2(v.x | (v.y << 16), v.z | (v.w << 16));
        } else {
            for (int i = 0; i < 4; ++i)
                if (g + i >= o_lo && g + i < o_hi) out[g + i] = (OutT)one(x + i);
        }
    }
}

__device__ __forceinline__ uint32_t full_word(int ncols, int w) {  // genomes 32w .. 32w+31 that exist
    const int left = ncols - 32 * w;
    return left >= 32 ? 0xFFFFFFFFu : (left <= 0 ? 0u : ((1u << left) - 1u));
}


// ------------------------------------------------------------------------------------------
// launch plumbing (defined in memo_sweep.hip).  Kernel-shape choices come from ix->tune (all zero in
// the product: the library chooses per query).
// ------------------------------------------------------------------------------------------
#ifdef MEMO_STAMPS
extern unsigned long long *g_stamp_buffer;  // diagnostic builds: 8 words per workgroup
#endif

using SweepKernel = void (*)(const SweepArgs);

inline int floor_log2(uint32_t v) { return 31 - __builtin_clz(v); }
int launch_tiles(SweepKernel kernel, SweepArgs &A, int w, int threads, size_t lds, hipStream_t st);
// memo_sweep_cons3t.hip: the table-driven dense-row sweep; 1 = this query does not fit it
int launch_halo3t(memo_index *ix, SweepArgs &A, int tw, int elem_bytes, hipStream_t st, bool annot9 = false, bool all_write = false,
                  bool six = false);
int pick_rows(const memo_index *ix, int32_t k, int &fmt);
int check_query_args(const memo_index *ix, int64_t qs, int64_t qe, int32_t k, int32_t num_docs,
                     const void *d_out);
void fill_args(const memo_index *ix, SweepArgs &A, int64_t qs, int64_t qe, int32_t k, void *d_out);

}  // namespace memo

#endif  // MEMO_SWEEP_H
