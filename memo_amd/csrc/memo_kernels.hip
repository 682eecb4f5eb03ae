// memo_kernels.hip -- MI355X (gfx950 / CDNA4) kernels + C ABI of the MEMO windowed query path.
//
// Replaces /root/reference/src/memo_query.py:42-63 and :70 (memo_init + memo_query +
// the argmax reduction of print_res).  See DESIGN.md for the algorithm; in short:
//
//   * one 64-lane wave owns one TILE of W consecutive pivot positions and is fully
//     independent of every other wave (no inter-workgroup traffic, no barriers that
//     span waves);
//   * the rows that can touch the tile are a contiguous slice of the start-sorted
//     columns, found with two loads from a bucket table built once per index;
//   * rows are streamed from HBM with 16-byte-per-lane coalesced loads, each row is
//     clipped to the tile and scattered into LDS;
//       conservation: the clipped interval [c, h) is covered by two power-of-two
//         blocks, one ds_min_u32 each into the level-log2 array; afterwards the levels
//         are folded top-down (block of 2^j -> two blocks of 2^(j-1)) so that level 0
//         holds min(order) per position.  min is idempotent and commutative, so the
//         overlap of the two blocks and the arrival order of atomics cannot change a
//         bit of the result.
//       membership: one ds_and_b32 per covered (position, genome) bit.
//   * the tile is written out with 16-byte stores.
//
// Integer work only (int64 compares, min, and): no MFMA.  Bound: HBM bandwidth.
#include <hip/hip_runtime.h>

#include <climits>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "memo_amd.h"

extern "C" int memo_sort_rows_by_start(int64_t *s, int64_t *e, int64_t *o, uint64_t rows,
                                       uint64_t padded_rows, hipStream_t stream, char *err,
                                       size_t errcap);

namespace {

// ------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------
thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                     \
    do {                                                                                  \
        hipError_t err__ = (expr);                                                        \
        if (err__ != hipSuccess)                                                          \
            return fail(MEMO_EHIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(err__),     \
                        __FILE__, __LINE__);                                              \
    } while (0)

struct DeviceGuard {  // the caller (e.g. torch) keeps its own notion of the current device
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = (prev == dev) || (hipSetDevice(dev) == hipSuccess);
    }
    ~DeviceGuard() {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
};

constexpr int kWave = 64;
constexpr uint64_t kPadRows = 4096;           // sentinel rows behind the last real row
constexpr int64_t kSentinel = INT64_MAX / 4;  // start/end of a padding row: clips to "empty"
constexpr int kDefaultBucketShift = 5;        // 32 pivot positions per bucket
constexpr int64_t kCoordLimit = (int64_t)1 << 61;

constexpr int kStatusBadAnnot = 1;

}  // namespace

struct memo_index {
    int device = 0;
    uint64_t rows = 0;
    uint64_t padded = 0;
    int64_t *s = nullptr, *e = nullptr, *o = nullptr;
    int64_t *boff = nullptr;  // boff[b] = first row with start >= (b << bshift); boff[nb-1] == rows
    uint64_t nb = 0;
    int bshift = 0;
    int64_t min_s = 0, max_s = -1;
    int finalized = 0;
    int was_sorted = 0;
    int *d_status = nullptr;   // sticky flags set by the sweep kernels
    uint64_t *d_scratch = nullptr;  // finalize(): [0] unsorted pairs, [1] rows with end < start
};

namespace {

// ------------------------------------------------------------------------------------------
// kernel arguments
// ------------------------------------------------------------------------------------------
struct SweepArgs {
    const int64_t *s, *e, *o;
    const int64_t *boff;
    int64_t nb;
    uint64_t rows;
    int64_t qs, qe;
    int64_t tile0;          // pivot position of tile 0 (multiple of the tile width, <= qs)
    int64_t ntiles;
    int64_t tiles_per_xcd;  // ceil(ntiles / 8)
    void *out;
    int *status;
    int bshift;
    int km1;    // k - 1 (>= 1 here; k <= 1 never reaches a sweep kernel)
    int ncols;  // result columns: num_docs + 1 (conservation) / num_docs (membership)
    int nlev;   // conservation: floor(log2(k-1)) + 1;  membership: words per position
};

// blockIdx -> tile.  Blocks are dealt round-robin over the 8 XCDs (b % 8 labels the XCD
// group), so give each group one contiguous run of tiles: neighbouring tiles share their
// k-1 halo rows and the cache lines that straddle the tile boundary, and those then hit in
// that XCD's L2 instead of being fetched twice.  Speed only -- results do not depend on it.
__device__ __forceinline__ int64_t tile_of_block(const SweepArgs &A) {
    const int64_t b = blockIdx.x;
    return (b & 7) * A.tiles_per_xcd + (b >> 3);
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
// conservation: doubling scatter + top-down fold
// ------------------------------------------------------------------------------------------
template <int W>
__device__ __forceinline__ void cons_row(uint32_t *lds, const SweepArgs &A, int64_t a, int x_lo,
                                         int x_hi, int64_t s, int64_t e, int64_t o) {
    // memo_query.py:46-48 restricted to the tile: recentre, shadow-cast by k-1, clip
    const int h = clamp_to_tile(s - a, x_lo, x_hi);
    const int c = clamp_to_tile(e - a - A.km1, x_lo, x_hi);
    const int len = h - c;  // :49  keep rows with casted_end < start
    if (len > 0) {
        int64_t col = o < 0 ? o + A.ncols : o;  // NumPy/Numba negative-index wrap
        if ((uint64_t)col >= (uint64_t)A.ncols) {
            atomicOr(A.status, kStatusBadAnnot);  // reference: IndexError / UB
        } else {
            const int j = 31 - __clz(len);
            uint32_t *lv = lds + j * W;
            atomicMin(lv + c, (uint32_t)col);                // block [c, c + 2^j)
            atomicMin(lv + (h - (1 << j)), (uint32_t)col);   // block [h - 2^j, h)
        }
    }
}

template <int W, int U, typename OutT>
__global__ __launch_bounds__(kWave) void sweep_conservation_kernel(const SweepArgs A) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int lane = threadIdx.x;
    const int64_t tile = tile_of_block(A);
    if (tile >= A.ntiles) return;
    const int64_t a = A.tile0 + tile * W;
    const int x_lo = (int)(A.qs > a ? A.qs - a : 0);
    const int x_hi = (int)(A.qe - a < W ? A.qe - a : W);

    // every level starts at the sentinel column N (memo_query.py:53-54)
    {
        const uint32_t sent = (uint32_t)(A.ncols - 1);
        const uint4 sv = make_uint4(sent, sent, sent, sent);
        uint4 *p = reinterpret_cast<uint4 *>(lds);
        for (int i = lane; i < A.nlev * (W / 4); i += kWave) p[i] = sv;
    }
    uint64_t r0, r1;
    row_slice(A, a, a + x_hi, r0, r1);
    __syncthreads();

    // stream the row slice: 2 rows per lane per column per load (16 B / lane, 1 KiB / wave)
    for (uint64_t base = (r0 & ~(uint64_t)15) + 2 * lane; base < r1; base += 2 * kWave * U) {
        longlong2 S[U], E[U], O[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t idx = base + (uint64_t)u * 2 * kWave;
            if (idx < r1) {
                S[u] = *reinterpret_cast<const longlong2 *>(A.s + idx);
                E[u] = *reinterpret_cast<const longlong2 *>(A.e + idx);
                O[u] = *reinterpret_cast<const longlong2 *>(A.o + idx);
            } else {
                S[u] = make_longlong2(kSentinel, kSentinel);
                E[u] = S[u];
                O[u] = make_longlong2(0, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            cons_row<W>(lds, A, a, x_lo, x_hi, S[u].x, E[u].x, O[u].x);
            cons_row<W>(lds, A, a, x_lo, x_hi, S[u].y, E[u].y, O[u].y);
        }
    }
    __syncthreads();

    // fold: a block of 2^j at x covers the blocks of 2^(j-1) at x and x + 2^(j-1)
    for (int j = A.nlev - 1; j >= 1; --j) {
        const int half = 1 << (j - 1);
        const uint32_t *hi = lds + j * W;
        uint32_t *lo = lds + (j - 1) * W;
        for (int x = 4 * lane; x < W; x += 4 * kWave) {
            const uint4 v = *reinterpret_cast<const uint4 *>(hi + x);
            uint4 u;
            if (half >= 4) {
                u = x >= half ? *reinterpret_cast<const uint4 *>(hi + x - half)
                              : make_uint4(~0u, ~0u, ~0u, ~0u);
            } else if (half == 2) {
                const uint2 t = x >= 2 ? *reinterpret_cast<const uint2 *>(hi + x - 2)
                                       : make_uint2(~0u, ~0u);
                u = make_uint4(t.x, t.y, v.x, v.y);
            } else {
                const uint32_t t = x >= 1 ? hi[x - 1] : ~0u;
                u = make_uint4(t, v.x, v.y, v.z);
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

    // write level 0 as OutT (uint16, or uint8 when num_docs <= 255), in 16-byte pieces aligned
    // in the OUTPUT (the tile grid is aligned in pivot coordinates, the output starts at qs)
    constexpr int PER = 16 / (int)sizeof(OutT);  // positions per 16-byte store
    OutT *out = static_cast<OutT *>(A.out);
    const int64_t ob = a - A.qs;  // output index of tile position 0
    const int64_t o_lo = ob + x_lo, o_hi = ob + x_hi;
    for (int64_t g = (o_lo & ~(int64_t)(PER - 1)) + PER * lane; g < o_hi; g += PER * kWave) {
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
}

// ------------------------------------------------------------------------------------------
// membership: one LDS and-atomic per covered (position, genome) bit
// ------------------------------------------------------------------------------------------
template <int W>
__device__ __forceinline__ void memb_row(uint32_t *lds, const SweepArgs &A, int64_t a, int x_lo,
                                         int x_hi, int64_t s, int64_t e, int64_t o) {
    const int h = clamp_to_tile(s - a, x_lo, x_hi);
    const int c = clamp_to_tile(e - a - A.km1, x_lo, x_hi);
    if (h > c) {
        int64_t col = o < 0 ? o + A.ncols : o;
        if ((uint64_t)col >= (uint64_t)A.ncols) {
            atomicOr(A.status, kStatusBadAnnot);
        } else {
            const int nw = A.nlev;
            uint32_t *cell = lds + c * nw + ((int)col >> 5);
            const uint32_t keep = ~(1u << ((int)col & 31));
            for (int x = c; x < h; ++x, cell += nw) atomicAnd(cell, keep);  // rec[c:h, a] = False
        }
    }
}

template <int W, int U>
__global__ __launch_bounds__(kWave) void sweep_membership_kernel(const SweepArgs A) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int lane = threadIdx.x;
    const int64_t tile = tile_of_block(A);
    if (tile >= A.ntiles) return;
    const int64_t a = A.tile0 + tile * W;
    const int x_lo = (int)(A.qs > a ? A.qs - a : 0);
    const int x_hi = (int)(A.qe - a < W ? A.qe - a : W);
    const int nw = A.nlev;

    // rec = ones([L, N])  (memo_query.py:51); bits >= N stay 0
    for (int i = lane; i < W * nw; i += kWave) {
        const int left = A.ncols - 32 * (i % nw);
        lds[i] = left >= 32 ? 0xFFFFFFFFu : ((1u << left) - 1u);
    }
    uint64_t r0, r1;
    row_slice(A, a, a + x_hi, r0, r1);
    __syncthreads();

    for (uint64_t base = (r0 & ~(uint64_t)15) + 2 * lane; base < r1; base += 2 * kWave * U) {
        longlong2 S[U], E[U], O[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t idx = base + (uint64_t)u * 2 * kWave;
            if (idx < r1) {
                S[u] = *reinterpret_cast<const longlong2 *>(A.s + idx);
                E[u] = *reinterpret_cast<const longlong2 *>(A.e + idx);
                O[u] = *reinterpret_cast<const longlong2 *>(A.o + idx);
            } else {
                S[u] = make_longlong2(kSentinel, kSentinel);
                E[u] = S[u];
                O[u] = make_longlong2(0, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            memb_row<W>(lds, A, a, x_lo, x_hi, S[u].x, E[u].x, O[u].x);
            memb_row<W>(lds, A, a, x_lo, x_hi, S[u].y, E[u].y, O[u].y);
        }
    }
    __syncthreads();

    // positions [x_lo, x_hi) are one contiguous run of words in LDS and in the output
    uint32_t *out = static_cast<uint32_t *>(A.out);
    const int64_t ob = (a - A.qs) * nw;  // output word of LDS word 0
    const int64_t o_lo = ob + (int64_t)x_lo * nw, o_hi = ob + (int64_t)x_hi * nw;
    for (int64_t g = (o_lo & ~(int64_t)3) + 4 * lane; g < o_hi; g += 4 * kWave) {
        const int x = (int)(g - ob);
        if (g >= o_lo && g + 4 <= o_hi) {
            *reinterpret_cast<uint4 *>(out + g) = make_uint4(lds[x], lds[x + 1], lds[x + 2], lds[x + 3]);
        } else {
            for (int i = 0; i < 4; ++i)
                if (g + i >= o_lo && g + i < o_hi) out[g + i] = lds[x + i];
        }
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

// ------------------------------------------------------------------------------------------
// index build: validation, padding, bucket table, synthetic rows
// ------------------------------------------------------------------------------------------
__global__ void check_rows_kernel(const int64_t *s, const int64_t *e, uint64_t rows,
                                  uint64_t *scratch) {
    uint64_t unsorted = 0, longrow = 0, wild = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < rows;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const int64_t si = s[i], ei = e[i];
        if (i > 0 && s[i - 1] > si) ++unsorted;
        if (ei < si) ++longrow;
        if (si <= -kCoordLimit || si >= kCoordLimit || ei <= -kCoordLimit || ei >= kCoordLimit) ++wild;
    }
    if (unsorted) atomicAdd((unsigned long long *)&scratch[0], (unsigned long long)unsorted);
    if (longrow) atomicAdd((unsigned long long *)&scratch[1], (unsigned long long)longrow);
    if (wild) atomicAdd((unsigned long long *)&scratch[2], (unsigned long long)wild);
}

__global__ void pad_rows_kernel(int64_t *s, int64_t *e, int64_t *o, uint64_t rows, uint64_t padded) {
    const uint64_t i = rows + blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i < padded) {
        s[i] = kSentinel;
        e[i] = kSentinel;
        o[i] = 0;
    }
}

// boff[b] = lower_bound(start, b << shift); the last bucket is pinned to `rows`
__global__ void bucket_table_kernel(const int64_t *s, uint64_t rows, int64_t *boff, uint64_t nb,
                                    int shift) {
    const uint64_t b = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (b >= nb) return;
    if (b == nb - 1) {
        boff[b] = (int64_t)rows;
        return;
    }
    const int64_t key = (int64_t)(b << shift);
    uint64_t lo = 0, hi = rows;
    while (lo < hi) {
        const uint64_t mid = lo + ((hi - lo) >> 1);
        if (s[mid] < key) lo = mid + 1; else hi = mid;
    }
    boff[b] = (int64_t)lo;
}

__device__ __forceinline__ uint64_t mix64(uint64_t seed, uint64_t x) {
    uint64_t z = seed + (x + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void synth_rows_kernel(int64_t *s, int64_t *e, int64_t *o, uint64_t rows,
                                  uint64_t row_begin, uint64_t num, uint64_t den, uint64_t nm1,
                                  uint64_t seed) {
    for (uint64_t j = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; j < rows;
         j += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i = row_begin + j;
        const int64_t st = 1 + (int64_t)((i * den) / num);
        s[j] = st;
        e[j] = st + (int64_t)(mix64(seed, 2 * i) % 60);
        o[j] = 1 + (int64_t)(mix64(seed, 2 * i + 1) % nm1);
    }
}

// ------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------
int g_tile_w = 0;    // 0 = choose per query
int g_variant = 0;   // reserved for kernel A/B
bool g_env_read = false;

void read_env_once() {
    if (g_env_read) return;
    g_env_read = true;
    if (const char *v = getenv("MEMO_TILE_W")) g_tile_w = atoi(v);
    if (const char *v = getenv("MEMO_VARIANT")) g_variant = atoi(v);
}

int floor_log2(uint32_t v) { return 31 - __builtin_clz(v); }

struct Window {
    int64_t tile0, ntiles, tiles_per_xcd;
};

Window make_window(int64_t qs, int64_t qe, int w) {
    Window win;
    // floor to a multiple of w (w is a power of two; >> on a negative int64 is arithmetic)
    const int sh = floor_log2((uint32_t)w);
    win.tile0 = (qs >> sh) << sh;
    win.ntiles = ((qe - win.tile0) + w - 1) >> sh;
    win.tiles_per_xcd = (win.ntiles + 7) / 8;
    return win;
}

template <int W, int U, typename OutT>
int launch_cons(SweepArgs &A, hipStream_t st) {
    const Window win = make_window(A.qs, A.qe, W);
    if (win.tiles_per_xcd * 8 * kWave >= ((int64_t)1 << 32))
        return fail(MEMO_EINVAL, "window too long for one launch at tile width %d", W);
    A.tile0 = win.tile0;
    A.ntiles = win.ntiles;
    A.tiles_per_xcd = win.tiles_per_xcd;
    const size_t lds = (size_t)A.nlev * W * sizeof(uint32_t);
    if (lds > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(sweep_conservation_kernel<W, U, OutT>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((sweep_conservation_kernel<W, U, OutT>), dim3((unsigned)(win.tiles_per_xcd * 8)),
                       dim3(kWave), lds, st, A);
    HIP_TRY(hipGetLastError());
    return MEMO_OK;
}

template <int W, int U>
int launch_memb(SweepArgs &A, hipStream_t st) {
    const Window win = make_window(A.qs, A.qe, W);
    if (win.tiles_per_xcd * 8 * kWave >= ((int64_t)1 << 32))
        return fail(MEMO_EINVAL, "window too long for one launch at tile width %d", W);
    A.tile0 = win.tile0;
    A.ntiles = win.ntiles;
    A.tiles_per_xcd = win.tiles_per_xcd;
    const size_t lds = (size_t)A.nlev * W * sizeof(uint32_t);
    if (lds > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(sweep_membership_kernel<W, U>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((sweep_membership_kernel<W, U>), dim3((unsigned)(win.tiles_per_xcd * 8)),
                       dim3(kWave), lds, st, A);
    HIP_TRY(hipGetLastError());
    return MEMO_OK;
}

int check_query_args(const memo_index *ix, int64_t qs, int64_t qe, int32_t k, int32_t num_docs,
                     const void *d_out, bool membership) {
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
    (void)membership;
    return MEMO_OK;
}

void fill_args(const memo_index *ix, SweepArgs &A, int64_t qs, int64_t qe, int32_t k, void *d_out) {
    A.s = ix->s;
    A.e = ix->e;
    A.o = ix->o;
    A.boff = ix->boff;
    A.nb = (int64_t)ix->nb;
    A.rows = ix->rows;
    A.qs = qs;
    A.qe = qe;
    A.out = d_out;
    A.status = ix->d_status;
    A.bshift = ix->bshift;
    A.km1 = k - 1;
}

}  // namespace

template <typename OutT>
static int query_conservation(memo_index_t *ix, int64_t qs, int64_t qe, int32_t k, int32_t num_docs,
                              OutT *d_out, void *stream) {
    read_env_once();
    int rc = check_query_args(ix, qs, qe, k, num_docs, d_out, false);
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
        return MEMO_OK;
    }
    SweepArgs A;
    fill_args(ix, A, qs, qe, k, d_out);
    A.ncols = num_docs + 1;
    A.nlev = floor_log2((uint32_t)(k - 1)) + 1;
    int w = g_tile_w;
    if (w == 0) w = A.nlev <= 8 ? 1024 : (A.nlev <= 16 ? 512 : 256);
    while ((size_t)A.nlev * w * 4 > 128 * 1024 && w > 256) w >>= 1;
    if ((size_t)A.nlev * w * 4 > 160 * 1024) return fail(MEMO_EINVAL, "k too large for the LDS tile");
    switch (w) {
        case 256: return launch_cons<256, 4, OutT>(A, st);
        case 512: return launch_cons<512, 4, OutT>(A, st);
        case 1024: return launch_cons<1024, 4, OutT>(A, st);
        case 2048: return launch_cons<2048, 4, OutT>(A, st);
        case 4096: return launch_cons<4096, 4, OutT>(A, st);
    }
    return fail(MEMO_EINVAL, "unsupported tile width %d", w);
}

// ==========================================================================================
// C ABI
// ==========================================================================================
extern "C" {

const char *memo_last_error(void) { return g_err; }

const char *memo_version(void) { return "memo_amd 0.1 (gfx950)"; }

int memo_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int memo_set_tuning(int32_t tile_w, int32_t variant) {
    read_env_once();
    if (tile_w != 0 && tile_w != 256 && tile_w != 512 && tile_w != 1024 && tile_w != 2048 &&
        tile_w != 4096)
        return fail(MEMO_EINVAL, "tile_w must be 0, 256, 512, 1024, 2048 or 4096");
    g_tile_w = tile_w;
    g_variant = variant;
    return MEMO_OK;
}

int memo_index_create(uint64_t rows, int32_t device, memo_index_t **out) {
    if (!out) return fail(MEMO_EINVAL, "out is NULL");
    *out = nullptr;
    if (rows > ((uint64_t)1 << 40)) return fail(MEMO_EINVAL, "too many rows");
    int ndev = memo_device_count();
    if (device < 0 || device >= ndev)
        return fail(MEMO_EHIP, "HIP device %d not available (%d visible)", device, ndev);
    DeviceGuard guard(device);
    if (!guard.ok) return fail(MEMO_EHIP, "cannot select HIP device %d", device);
    memo_index *ix = new (std::nothrow) memo_index();
    if (!ix) return fail(MEMO_EHIP, "out of host memory");
    ix->device = device;
    ix->rows = rows;
    ix->padded = ((rows + 15) & ~(uint64_t)15) + kPadRows;
    const size_t bytes = ix->padded * sizeof(int64_t);
    hipError_t err = hipMalloc(&ix->s, bytes);
    if (err == hipSuccess) err = hipMalloc(&ix->e, bytes);
    if (err == hipSuccess) err = hipMalloc(&ix->o, bytes);
    if (err == hipSuccess) err = hipMalloc(&ix->d_status, 64);
    if (err == hipSuccess) err = hipMalloc(&ix->d_scratch, 64);
    if (err == hipSuccess) err = hipMemset(ix->d_status, 0, 64);
    if (err != hipSuccess) {
        memo_index_destroy(ix);
        return fail(MEMO_EHIP, "hipMalloc of %zu bytes x3 failed: %s", bytes, hipGetErrorString(err));
    }
    *out = ix;
    return MEMO_OK;
}

void memo_index_destroy(memo_index_t *ix) {
    if (!ix) return;
    DeviceGuard guard(ix->device);
    (void)hipFree(ix->s);
    (void)hipFree(ix->e);
    (void)hipFree(ix->o);
    (void)hipFree(ix->boff);
    (void)hipFree(ix->d_status);
    (void)hipFree(ix->d_scratch);
    delete ix;
}

int memo_index_upload(memo_index_t *ix, const int64_t *start, const int64_t *end,
                      const int64_t *annot, uint64_t rows) {
    if (!ix) return fail(MEMO_EINVAL, "index is NULL");
    if (rows != ix->rows) return fail(MEMO_EINVAL, "upload of %llu rows into an index of %llu",
                                      (unsigned long long)rows, (unsigned long long)ix->rows);
    if (rows && (!start || !end || !annot)) return fail(MEMO_EINVAL, "column pointer is NULL");
    DeviceGuard guard(ix->device);
    if (rows) {
        HIP_TRY(hipMemcpy(ix->s, start, rows * sizeof(int64_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(ix->e, end, rows * sizeof(int64_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(ix->o, annot, rows * sizeof(int64_t), hipMemcpyHostToDevice));
    }
    ix->finalized = 0;
    return MEMO_OK;
}

int memo_index_columns(memo_index_t *ix, int64_t **d_start, int64_t **d_end, int64_t **d_annot) {
    if (!ix) return fail(MEMO_EINVAL, "index is NULL");
    if (d_start) *d_start = ix->s;
    if (d_end) *d_end = ix->e;
    if (d_annot) *d_annot = ix->o;
    ix->finalized = 0;  // the caller may be about to rewrite the rows
    return MEMO_OK;
}

int memo_index_finalize(memo_index_t *ix, int32_t bucket_shift, int32_t allow_sort) {
    if (!ix) return fail(MEMO_EINVAL, "index is NULL");
    if (bucket_shift <= 0) bucket_shift = kDefaultBucketShift;
    if (bucket_shift > 8) return fail(MEMO_EINVAL, "bucket_shift must be <= 8 (tile width 256)");
    DeviceGuard guard(ix->device);
    hipStream_t st = nullptr;
    const uint64_t rows = ix->rows;
    {
        const uint64_t npad = ix->padded - rows;
        hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)((npad + 255) / 256)), dim3(256), 0, st,
                           ix->s, ix->e, ix->o, rows, ix->padded);
        HIP_TRY(hipGetLastError());
    }
    uint64_t h[8] = {0};
    ix->was_sorted = 1;
    if (rows) {
        HIP_TRY(hipMemsetAsync(ix->d_scratch, 0, 64, st));
        const unsigned grid = (unsigned)(rows / 256 + 1 < 4096 ? rows / 256 + 1 : 4096);
        hipLaunchKernelGGL(check_rows_kernel, dim3(grid), dim3(256), 0, st, ix->s, ix->e, rows,
                           ix->d_scratch);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpy(h, ix->d_scratch, 24, hipMemcpyDeviceToHost));
        if (h[2]) return fail(MEMO_EINVAL, "%llu rows have coordinates beyond +-2^61", (unsigned long long)h[2]);
        if (h[1])
            return fail(MEMO_ELONGROW, "%llu rows have end < start: not a MEMO overlap index",
                        (unsigned long long)h[1]);
        if (h[0]) {
            ix->was_sorted = 0;
            if (!allow_sort)
                return fail(MEMO_EUNSORTED, "rows are not sorted by start (%llu descents)",
                            (unsigned long long)h[0]);
            char msg[256] = "";
            if (memo_sort_rows_by_start(ix->s, ix->e, ix->o, rows, ix->padded, st, msg, sizeof msg) != 0)
                return fail(MEMO_EHIP, "device sort failed: %s", msg);
        }
        HIP_TRY(hipMemcpy(&ix->min_s, ix->s, sizeof(int64_t), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(&ix->max_s, ix->s + (rows - 1), sizeof(int64_t), hipMemcpyDeviceToHost));
    } else {
        ix->min_s = 0;
        ix->max_s = -1;
    }
    // buckets 0 .. ceil((max_s + 1) / width), plus one pinned to `rows`
    const int64_t top = ix->max_s < 0 ? 0 : ix->max_s;
    const uint64_t nb = (uint64_t)((top >> bucket_shift) + 3);
    if (ix->boff) {
        (void)hipFree(ix->boff);
        ix->boff = nullptr;
    }
    HIP_TRY(hipMalloc(&ix->boff, nb * sizeof(int64_t)));
    hipLaunchKernelGGL(bucket_table_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, st,
                       ix->s, rows, ix->boff, nb, bucket_shift);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    ix->nb = nb;
    ix->bshift = bucket_shift;
    ix->finalized = 1;
    return MEMO_OK;
}

int memo_index_get_info(const memo_index_t *ix, memo_index_info_t *info) {
    if (!ix || !info) return fail(MEMO_EINVAL, "NULL argument");
    info->rows = ix->rows;
    info->min_start = ix->min_s;
    info->max_start = ix->max_s;
    info->device = ix->device;
    info->bucket_shift = ix->bshift;
    info->buckets = ix->nb;
    info->was_sorted = ix->was_sorted;
    info->finalized = ix->finalized;
    info->device_bytes = ix->padded * 3 * sizeof(int64_t) + ix->nb * sizeof(int64_t) + 128;
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
    int rc = check_query_args(ix, qs, qe, k, num_docs, d_out, true);
    if (rc) return rc;
    if (qe <= qs) return MEMO_OK;
    DeviceGuard guard(ix->device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nw = (num_docs + 31) / 32;
    if (k <= 1 || ix->rows == 0) {
        hipLaunchKernelGGL(fill_membership_kernel, dim3(2048), dim3(256), 0, st, d_out,
                           (qe - qs) * nw, nw, num_docs);
        HIP_TRY(hipGetLastError());
        return MEMO_OK;
    }
    SweepArgs A;
    fill_args(ix, A, qs, qe, k, d_out);
    A.ncols = num_docs;
    A.nlev = nw;
    int w = g_tile_w;
    if (w == 0) {
        w = 1024;
        while ((size_t)nw * w * 4 > 20 * 1024 && w > 256) w >>= 1;
    }
    while ((size_t)nw * w * 4 > 128 * 1024 && w > 256) w >>= 1;
    if ((size_t)nw * w * 4 > 160 * 1024) return fail(MEMO_EINVAL, "num_docs too large for the LDS tile");
    switch (w) {
        case 256: return launch_memb<256, 4>(A, st);
        case 512: return launch_memb<512, 4>(A, st);
        case 1024: return launch_memb<1024, 4>(A, st);
        case 2048: return launch_memb<2048, 4>(A, st);
        case 4096: return launch_memb<4096, 4>(A, st);
    }
    return fail(MEMO_EINVAL, "unsupported tile width %d", w);
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
        if (flags & kStatusBadAnnot)
            return fail(MEMO_EINVAL,
                        "a row that covers the window has an order/genome column outside the "
                        "result matrix (num_docs too small?) -- the reference raises IndexError here");
        return fail(MEMO_EINVAL, "device status 0x%x", flags);
    }
    return MEMO_OK;
}

static int one_shot(const int64_t *start, const int64_t *end, const int64_t *annot, uint64_t rows,
                    int64_t qs, int64_t qe, int32_t k, int32_t num_docs, void *out, int32_t device,
                    bool membership) {
    memo_index_t *ix = nullptr;
    int rc = memo_index_create(rows, device, &ix);
    if (rc) return rc;
    void *d_out = nullptr;
    do {
        if ((rc = memo_index_upload(ix, start, end, annot, rows))) break;
        if ((rc = memo_index_finalize(ix, 0, 1))) break;
        if (qe < qs) { rc = fail(MEMO_EINVAL, "ValueError: negative dimensions are not allowed (window end < start)"); break; }
        const int64_t L = qe - qs;
        if (L > 0 && !out) { rc = fail(MEMO_EINVAL, "output pointer is NULL"); break; }
        const size_t bytes = membership ? (size_t)L * ((num_docs + 31) / 32) * 4 : (size_t)L * 2;
        DeviceGuard guard(device);
        if (bytes) {
            hipError_t err = hipMalloc(&d_out, bytes);
            if (err != hipSuccess) { rc = fail(MEMO_EHIP, "hipMalloc(%zu): %s", bytes, hipGetErrorString(err)); break; }
        }
        rc = membership ? memo_query_membership_dev(ix, qs, qe, k, num_docs, (uint32_t *)d_out, nullptr)
                        : memo_query_conservation_dev(ix, qs, qe, k, num_docs, (uint16_t *)d_out, nullptr);
        if (rc) break;
        if ((rc = memo_query_check(ix, nullptr))) break;
        if (bytes) {
            hipError_t err = hipMemcpy(out, d_out, bytes, hipMemcpyDeviceToHost);
            if (err != hipSuccess) { rc = fail(MEMO_EHIP, "hipMemcpy D2H: %s", hipGetErrorString(err)); break; }
        }
    } while (0);
    if (d_out) {
        DeviceGuard guard(device);
        (void)hipFree(d_out);
    }
    memo_index_destroy(ix);
    return rc;
}

int memo_conservation(const int64_t *start, const int64_t *end, const int64_t *annot, uint64_t rows,
                      int64_t qs, int64_t qe, int32_t k, int32_t num_docs, uint16_t *out,
                      int32_t device) {
    return one_shot(start, end, annot, rows, qs, qe, k, num_docs, out, device, false);
}

int memo_membership(const int64_t *start, const int64_t *end, const int64_t *annot, uint64_t rows,
                    int64_t qs, int64_t qe, int32_t k, int32_t num_docs, uint32_t *out_bits,
                    int32_t device) {
    return one_shot(start, end, annot, rows, qs, qe, k, num_docs, out_bits, device, true);
}

int memo_synth_fill(memo_index_t *ix, uint64_t row_begin, uint64_t num, uint64_t den,
                    int32_t num_docs, uint64_t seed) {
    if (!ix) return fail(MEMO_EINVAL, "index is NULL");
    if (num == 0 || den == 0 || num_docs < 2) return fail(MEMO_EINVAL, "bad generator parameters");
    DeviceGuard guard(ix->device);
    if (ix->rows) {
        hipLaunchKernelGGL(synth_rows_kernel, dim3(4096), dim3(256), 0, nullptr, ix->s, ix->e, ix->o,
                           ix->rows, row_begin, num, den, (uint64_t)(num_docs - 1), seed);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
    }
    ix->finalized = 0;
    return MEMO_OK;
}

// ---- raw device buffers for hosts that do not bring their own allocator ---------------------
int memo_dev_malloc(int32_t device, size_t bytes, void **out) {
    if (!out) return fail(MEMO_EINVAL, "out is NULL");
    *out = nullptr;
    DeviceGuard guard(device);
    if (!guard.ok) return fail(MEMO_EHIP, "cannot select HIP device %d", device);
    HIP_TRY(hipMalloc(out, bytes ? bytes : 16));
    return MEMO_OK;
}

int memo_dev_free(int32_t device, void *p) {
    DeviceGuard guard(device);
    HIP_TRY(hipFree(p));
    return MEMO_OK;
}

int memo_dev_download(int32_t device, void *host, const void *dev, size_t bytes, void *stream) {
    DeviceGuard guard(device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (bytes) HIP_TRY(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return MEMO_OK;
}

// ---- print_res (memo_query.py:65-71) -----------------------------------------------------
size_t memo_emit_conservation(const uint16_t *vec, int64_t L, char *buf, size_t cap) {
    if (L <= 0) {  // print(*[], sep='\n') still writes the newline
        if (cap >= 1 && buf) buf[0] = '\n';
        return 1;
    }
    size_t need = 0;
    for (int64_t i = 0; i < L; ++i) {
        const unsigned v = vec[i];
        need += v < 10 ? 2 : v < 100 ? 3 : v < 1000 ? 4 : v < 10000 ? 5 : 6;
    }
    if (need > cap || !buf) return need;
    char *p = buf;
    for (int64_t i = 0; i < L; ++i) {
        unsigned v = vec[i];
        char tmp[6];
        int n = 0;
        do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
        while (n) *p++ = tmp[--n];
        *p++ = '\n';
    }
    return need;
}

size_t memo_emit_membership(const uint32_t *bits, int64_t L, int32_t num_docs, char *buf, size_t cap) {
    if (L <= 0) return 0;
    const size_t per_line = num_docs > 0 ? (size_t)2 * num_docs : 1;
    const size_t need = per_line * (size_t)L;
    if (need > cap || !buf) return need;
    const int nw = (num_docs + 31) / 32;
    char *p = buf;
    for (int64_t i = 0; i < L; ++i) {
        const uint32_t *row = bits + i * nw;
        for (int g = 0; g < num_docs; ++g) {
            *p++ = (char)('0' + ((row[g >> 5] >> (g & 31)) & 1u));
            *p++ = ' ';
        }
        if (num_docs > 0) p[-1] = '\n'; else *p++ = '\n';
    }
    return need;
}

}  // extern "C"
