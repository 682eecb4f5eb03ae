// memo_index.hip -- the resident index: upload, validation, bucket table, packed rows; the
// one-shot host entry points; `memo view` binning; synthetic rows; small device-buffer helpers.
//
// Counterpart of the arrays /root/reference/src/memo_query.py hands from filter_pq to memo_init
// (:28-36, :45): three int64 columns, kept in HBM so that many windows reuse one upload.
#include <chrono>
#include <cstdarg>
#include <exception>
#include <cstddef>
#include <new>

#include "memo_common.h"
#include "memo_hostcore.h"

using namespace memo;

namespace memo {
void drop_dense(memo_index *ix) {
    drop_tile_tables(ix);
    drop_dense_views(ix);
    (void)hipFree(ix->p3);
    (void)hipFree(ix->boff3);
    ix->p3 = nullptr;
    ix->boff3 = nullptr;
    ix->rows3 = ix->padded3 = 0;
}
}  // namespace memo

namespace memo {
thread_local int g_last_one_shot_sweep = 0;  // memo_index_info_t.last_sweep of this thread's last one-shot call (memo_debug.hip)
}

extern "C" __attribute__((visibility("hidden"))) int memo_sort_rows_by_start(int64_t *s, int64_t *e, int64_t *o, uint64_t rows,
                                       uint64_t padded_rows, hipStream_t stream, char *err,
                                       size_t errcap);

namespace memo {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace memo

namespace {

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

// finalize: copy the rows with end < start aside (order is irrelevant: min / and commute)
__global__ void collect_long_rows_kernel(const int64_t *s, const int64_t *e, const int64_t *o, uint64_t rows,
                                         int64_t *ls, int64_t *le, int64_t *lo, unsigned long long *count) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < rows;
         i += (uint64_t)gridDim.x * blockDim.x)
        if (e[i] < s[i]) {
            const unsigned long long k = atomicAdd(count, 1ull);
            ls[k] = s[i];
            le[k] = e[i];
            lo[k] = o[i];
        }
}

// memo_index_pack: annot range census, then one word (+ optional 16-bit annot) per row
// (step > 1: a SAMPLE, every step-th row -- memo_index_pack guesses the word layout from it and packs at once; the packing kernel
// takes the exact census on the way: reading the annot column twice was 0.85 ms of config 3's packing pass)
__global__ void annot_census_kernel(const int64_t *o, uint64_t rows, uint64_t *scratch, uint64_t step) {
    uint64_t outside = 0, over8 = 0, top = 0;
    for (uint64_t i = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) * step; i < rows;
         i += (uint64_t)gridDim.x * blockDim.x * step) {
        const int64_t v = o[i];
        if (v < 0 || v > 65535) ++outside;
        else if ((uint64_t)v > top) top = (uint64_t)v;
        if (v > 255) ++over8;
    }
    if (outside) atomicAdd((unsigned long long *)&scratch[3], (unsigned long long)outside);
    if (over8) atomicAdd((unsigned long long *)&scratch[4], (unsigned long long)over8);
    if (top) atomicMax((unsigned long long *)&scratch[5], (unsigned long long)top);
}

// fmt: 4 = start16 | len8 << 16 | annot8 << 24;  12 = len8 | start12 << 8 | annot12 << 20;  6 = the first word with
// annot 0 + a 16-bit annot column  (PackedRows, memo_sweep.h)
// ... and, with `census`, what annot_census_kernel counts, of every row: scratch[3] annots outside [0, 65535], [5] the largest
__global__ void pack_rows_kernel(const int64_t *s, const int64_t *e, const int64_t *o, uint64_t rows,
                                 uint64_t padded, uint32_t *pk, uint16_t *pa, int fmt, uint64_t *census) {
    uint64_t outside = 0, top = 0;
    auto word = [&](uint64_t i, int64_t si, int64_t ei, int64_t v, uint32_t &a) -> uint32_t {
        uint32_t w = 0;
        a = 0;
        if (i < rows) {
            const int64_t len = ei - si;
            // end < start (handled by long_rows_kernel) packs as "never writes", like len >= 255
            const uint32_t l8 = (uint32_t)(len > 255 || len < 0 ? 255 : len);
            if (v < 0 || v > 65535) ++outside;
            else if ((uint64_t)v > top) top = (uint64_t)v;
            a = (uint32_t)v;
            w = fmt == 12 ? l8 | (((uint32_t)si & 0xFFFu) << 8) | (a << 20) : ((uint32_t)si & 0xFFFFu) | (l8 << 16);
        }
        if (fmt == 4) w |= a << 24;
        return w;
    };
    // two rows per lane and load (16 bytes of each column; `padded` is a multiple of 16 rows and the columns are that long), two
    // such pairs in flight: 24 B in, 4 B out per row at 5.1 TB/s with one row per lane and load (2.75 ms per 5 * 10^8 rows)
    const uint64_t stride = 2 * (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = 2 * (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x); i < padded; i += 2 * stride) {
        const uint64_t j = i + stride;
        const bool two = j < padded;
        longlong2 S0 = make_longlong2(0, 0), E0 = S0, O0 = S0, S1 = S0, E1 = S0, O1 = S0;
        // (read once, never again: non-temporal, so that 12 GB of columns do not sweep the L2 and the Infinity Cache)
        auto load2 = [](const int64_t *p) {
            longlong2 v;
            v.x = __builtin_nontemporal_load(p);
            v.y = __builtin_nontemporal_load(p + 1);
            return v;
        };
        if (i < rows) {  // (behind the last row only the words' zeros are written: an empty index has no columns at all)
            S0 = load2(s + i);
            E0 = load2(e + i);
            O0 = load2(o + i);
        }
        if (j < rows) {
            S1 = load2(s + j);
            E1 = load2(e + j);
            O1 = load2(o + j);
        }
        uint32_t a0, a1;
        const uint32_t w0 = word(i, S0.x, E0.x, O0.x, a0), w1 = word(i + 1, S0.y, E0.y, O0.y, a1);
        *reinterpret_cast<uint2 *>(pk + i) = make_uint2(w0, w1);
        if (fmt == 6) *reinterpret_cast<uint32_t *>(pa + i) = a0 | (a1 << 16);
        if (two) {
            const uint32_t w2 = word(j, S1.x, E1.x, O1.x, a0), w3 = word(j + 1, S1.y, E1.y, O1.y, a1);
            *reinterpret_cast<uint2 *>(pk + j) = make_uint2(w2, w3);
            if (fmt == 6) *reinterpret_cast<uint32_t *>(pa + j) = a0 | (a1 << 16);
        }
    }
    if (census) {  // one pair of atomics per wave
        for (int off = 32; off; off >>= 1) {
            const uint64_t t = (uint64_t)__shfl_xor((long long)top, off, 64);
            top = t > top ? t : top;
            outside += (uint64_t)__shfl_xor((long long)outside, off, 64);
        }
        if ((threadIdx.x & 63) == 0) {
            if (outside) atomicAdd((unsigned long long *)&census[3], (unsigned long long)outside);
            if (top) atomicMax((unsigned long long *)&census[5], (unsigned long long)top);
        }
    }
}

// memo_index_pack_dense: 4-byte words -> dense rows, five per 16-byte group (layout: PackedRows3, memo_sweep.h).
// f12: the words are format 12 (overlap | start << 8 | annot << 20) with annots of up to NINE bits: the ninth bit of row i's
// annot goes to bit 16 + i of the group's last dword (the byte no row used while annots had eight)
__global__ void pack3_rows_kernel(const uint32_t *pk, uint64_t padded, uint64_t groups, uint4 *p3, int f12) {
    for (uint64_t g = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; g < groups;
         g += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t B[5], A[5], hi = 0;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const uint32_t x = 5 * g + i < padded ? pk[5 * g + i] : 0u;
            const uint32_t len = f12 ? x & 0xFFu : (x >> 16) & 0xFFu, start = f12 ? x >> 8 : x, annot = f12 ? (x >> 20) & 0x1FFu : x >> 24;
            B[i] = ((start & 1023u) << 6) | (len > 63u ? 63u : len);  // (start & 1023) << 6 | min(length, 63)
            A[i] = annot & 0xFFu;
            hi |= (annot >> 8) << i;
        }
        p3[g] = make_uint4(B[0] | ((B[4] & 0xFFu) << 16) | (A[0] << 24), B[1] | ((B[4] >> 8) << 16) | (A[1] << 24),
                           B[2] | (A[4] << 16) | (A[2] << 24), B[3] | (hi << 16) | (A[3] << 24));
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

// `memo view` binning (plot_conservation.py:52-56): counts[b][v] = #{p in [edge[b], edge[b+1]) : vec[p] == v}
// for v in 0..num_docs.  One workgroup per (bin, slice of the bin); LDS histogram when it fits.
template <bool LDS_HIST>
__global__ __launch_bounds__(256) void bin_conservation_kernel(const uint16_t *vec, const int64_t *edges,
                                                               int ncols, int slices,
                                                               unsigned long long *counts) {
    extern __shared__ uint32_t hist[];
    const int b = blockIdx.x / slices, sl = blockIdx.x % slices;
    const int64_t lo = edges[b], hi = edges[b + 1];
    const int64_t per = (hi - lo + slices - 1) / slices;
    const int64_t p0 = lo + sl * per, p1 = p0 + per < hi ? p0 + per : hi;
    if (LDS_HIST) {
        for (int i = threadIdx.x; i < ncols; i += 256) hist[i] = 0;
        __syncthreads();
    }
    for (int64_t p = p0 + threadIdx.x; p < p1; p += 256) {
        const int v = vec[p];
        if (v < ncols) {
            if (LDS_HIST) atomicAdd(&hist[v], 1u);
            else atomicAdd(&counts[(int64_t)b * ncols + v], 1ull);
        }
    }
    if (LDS_HIST) {
        __syncthreads();
        for (int i = threadIdx.x; i < ncols; i += 256)
            if (hist[i]) atomicAdd(&counts[(int64_t)b * ncols + i], (unsigned long long)hist[i]);
    }
}

// Transport coding of uint8 conservation results for the multi-GPU gather: one nibble per position
// (values >= 15 become 15 and go to an exception list as position << 8 | value).  Lossless; halves
// what a slice puts on its xGMI link when few values reach 15.
__global__ __launch_bounds__(256) void nibble_pack_kernel(const uint8_t *in, int64_t n, uint32_t *nib,
                                                          unsigned long long *exc, unsigned int *count,
                                                          unsigned int cap) {
    // A workgroup codes 32768 consecutive positions (16 rounds of 256 threads x 8 positions).  Exceptions
    // are collected in LDS -- no barrier between the rounds, LDS atomics order themselves -- and appended
    // with ONE global atomic per workgroup: a quarter of a million same-address atomics would cost
    // milliseconds, and round 1's version, which met at three barriers per round, ran at a fifth of its
    // memory bound.  More than 4096 exceptions in 32768 positions (an eighth of them >= 15) is not data this
    // coding is for: the count is saturated so that the receiver sees an incomplete slice.
    constexpr int kRounds = 16, kHeld = 4096;
    __shared__ unsigned long long held[kHeld];
    __shared__ unsigned int n_held, base;
    if (threadIdx.x == 0) n_held = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) count[1] = cap;  // header word 1
    __syncthreads();
    const int64_t groups = (n + 7) / 8;
#pragma unroll 4
    for (int r = 0; r < kRounds; ++r) {
        const int64_t g = ((int64_t)blockIdx.x * kRounds + r) * 256 + threadIdx.x;
        if (g >= groups) break;
        unsigned long long eight = 0;  // 8 results in one load (the tail group byte by byte)
        if (g * 8 + 8 <= n) {
            eight = *reinterpret_cast<const unsigned long long *>(in + g * 8);
        } else {
            for (int i = 0; g * 8 + i < n; ++i) eight |= (unsigned long long)in[g * 8 + i] << (8 * i);
        }
        uint32_t word = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t v = (uint32_t)(eight >> (8 * i)) & 0xFFu;
            if (v >= 15u) {
                const unsigned int slot = atomicAdd(&n_held, 1u);
                if (slot < (unsigned)kHeld) held[slot] = ((unsigned long long)(g * 8 + i) << 8) | v;
            }
            word |= (v < 15u ? v : 15u) << (4 * i);
        }
        nib[g] = word;
    }
    __syncthreads();
    const unsigned int mine = n_held;
    if (!mine) return;
    if (mine > (unsigned)kHeld) {
        if (threadIdx.x == 0) count[2] = 1;  // header word 2: overflow -- the slice is incomplete, whatever the capacity
        return;
    }
    if (threadIdx.x == 0) base = atomicAdd(count, mine);
    __syncthreads();
    for (unsigned int i = threadIdx.x; i < mine; i += 256)
        if (base + i < cap) exc[base + i] = held[i];
}

__global__ void nibble_unpack_kernel(const uint32_t *nib, int64_t n, uint8_t *out) {
    for (int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; g * 8 < n;
         g += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t word = nib[g];
        unsigned long long eight = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) eight |= (unsigned long long)((word >> (4 * i)) & 15u) << (8 * i);
        if (g * 8 + 8 <= n) {
            *reinterpret_cast<unsigned long long *>(out + g * 8) = eight;
        } else {
            for (int i = 0; g * 8 + i < n; ++i) out[g * 8 + i] = (uint8_t)(eight >> (8 * i));
        }
    }
}

__global__ void nibble_exceptions_kernel(const unsigned long long *exc, const unsigned int *head, int64_t n,
                                         uint8_t *out) {
    const unsigned int count = head[0] < head[1] ? head[0] : head[1];  // found, capacity
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
        const unsigned long long e = exc[i];
        const int64_t p = (int64_t)(e >> 8);
        if (p < n) out[p] = (uint8_t)(e & 0xFF);
    }
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

}  // namespace

extern "C" {

const char *memo_last_error(void) { return g_err; }

const char *memo_version(void) { return "memo_amd 0.1 (gfx950)"; }

int memo_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static void drop_packed(memo_index *ix) {  // the rows are about to change
    drop_dense(ix);
    drop_packed_views(ix);
    (void)hipFree(ix->pk);
    (void)hipFree(ix->pa);
    ix->pk = nullptr;
    ix->pa = nullptr;
    ix->packed_fmt = 0;
    ix->packed_rows = 0;
}

// the rows are about to change but the index keeps its size: the packed copy is stale, its buffers
// can serve the next memo_index_pack
static void stale_packed(memo_index *ix) {
    drop_dense(ix);
    drop_packed_views(ix);
    ix->packed_fmt = 0;
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
    drop_dense(ix);
    drop_packed_views(ix);
    (void)hipFree(ix->boff);
    (void)hipFree(ix->pk);
    (void)hipFree(ix->pa);
    (void)hipFree(ix->ls);
    (void)hipFree(ix->le);
    (void)hipFree(ix->lo);
    (void)hipFree(ix->d_status);
    (void)hipFree(ix->d_scratch);
    flush_retired(ix);  // (hipFree waits for the device)
    delete ix;
}

int memo_index_upload(memo_index_t *ix, const int64_t *start, const int64_t *end,
                      const int64_t *annot, uint64_t rows) {
    if (!ix) return fail(MEMO_EINVAL, "index is NULL");
    if (rows != ix->rows) return fail(MEMO_EINVAL, "upload of %llu rows into an index of %llu",
                                      (unsigned long long)rows, (unsigned long long)ix->rows);
    if (rows && (!start || !end || !annot)) return fail(MEMO_EINVAL, "column pointer is NULL");
    if (!ix->has_wide) return fail(MEMO_EINVAL, "the int64 columns of this index were dropped by memo_index_pack");
    DeviceGuard guard(ix->device);
    drop_packed(ix);
    if (rows) {
        HIP_TRY(hipMemcpy(ix->s, start, rows * sizeof(int64_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(ix->e, end, rows * sizeof(int64_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(ix->o, annot, rows * sizeof(int64_t), hipMemcpyHostToDevice));
    }
    ix->finalized = 0;
    return MEMO_OK;
}

int memo_index_upload_rows(memo_index_t *ix, uint64_t row_offset, const int64_t *start,
                           const int64_t *end, const int64_t *annot, uint64_t rows) {
    if (!ix) return fail(MEMO_EINVAL, "index is NULL");
    if (row_offset > ix->rows || rows > ix->rows - row_offset)
        return fail(MEMO_EINVAL, "rows [%llu, +%llu) do not fit an index of %llu rows",
                    (unsigned long long)row_offset, (unsigned long long)rows, (unsigned long long)ix->rows);
    if (rows && (!start || !end || !annot)) return fail(MEMO_EINVAL, "column pointer is NULL");
    if (!ix->has_wide) return fail(MEMO_EINVAL, "the int64 columns of this index were dropped by memo_index_pack");
    DeviceGuard guard(ix->device);
    drop_packed(ix);
    if (rows) {
        HIP_TRY(hipMemcpy(ix->s + row_offset, start, rows * sizeof(int64_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(ix->e + row_offset, end, rows * sizeof(int64_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(ix->o + row_offset, annot, rows * sizeof(int64_t), hipMemcpyHostToDevice));
    }
    ix->finalized = 0;
    return MEMO_OK;
}

int memo_index_truncate(memo_index_t *ix, uint64_t rows) {
    if (!ix) return fail(MEMO_EINVAL, "index is NULL");
    if (rows > ix->rows) return fail(MEMO_EINVAL, "cannot grow an index (%llu > %llu rows)",
                                     (unsigned long long)rows, (unsigned long long)ix->rows);
    if (!ix->has_wide) return fail(MEMO_EINVAL, "the int64 columns of this index were dropped by memo_index_pack");
    {
        DeviceGuard guard(ix->device);
        drop_packed(ix);
    }
    ix->rows = rows;  // `padded` keeps the allocated size; finalize() rewrites the sentinel rows behind `rows`
    ix->finalized = 0;
    return MEMO_OK;
}

int memo_index_columns(memo_index_t *ix, int64_t **d_start, int64_t **d_end, int64_t **d_annot) {
    if (!ix) return fail(MEMO_EINVAL, "index is NULL");
    if (!ix->has_wide) return fail(MEMO_EINVAL, "the int64 columns of this index were dropped by memo_index_pack");
    {
        DeviceGuard guard(ix->device);
        drop_packed(ix);
    }
    if (d_start) *d_start = ix->s;
    if (d_end) *d_end = ix->e;
    if (d_annot) *d_annot = ix->o;
    ix->finalized = 0;  // the caller may be about to rewrite the rows
    return MEMO_OK;
}

int memo_index_finalize(memo_index_t *ix, int32_t bucket_shift, int32_t allow_sort) {
    if (!ix) return fail(MEMO_EINVAL, "index is NULL");
    if (!ix->has_wide) return fail(MEMO_EINVAL, "the int64 columns of this index were dropped by memo_index_pack");
    if (bucket_shift <= 0) bucket_shift = kDefaultBucketShift;
    if (bucket_shift > 8) return fail(MEMO_EINVAL, "bucket_shift must be <= 8 (tile width 256)");
    DeviceGuard guard(ix->device);
    hipStream_t st = nullptr;
    const uint64_t rows = ix->rows;
    stale_packed(ix);  // a device sort below would leave packed rows stale; pack again after finalize
    ix->finalized = 0;
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
        (void)hipFree(ix->ls);
        (void)hipFree(ix->le);
        (void)hipFree(ix->lo);
        ix->ls = ix->le = ix->lo = nullptr;
        ix->n_long = 0;
        if (h[1]) {  // rows with end < start: set aside for long_rows_kernel
            if (h[1] > ((uint64_t)1 << 22))
                return fail(MEMO_ELONGROW, "%llu rows have end < start: not a MEMO overlap index",
                            (unsigned long long)h[1]);
            HIP_TRY(hipMalloc(&ix->ls, h[1] * sizeof(int64_t)));
            HIP_TRY(hipMalloc(&ix->le, h[1] * sizeof(int64_t)));
            HIP_TRY(hipMalloc(&ix->lo, h[1] * sizeof(int64_t)));
            HIP_TRY(hipMemsetAsync(ix->d_scratch + 6, 0, 8, st));
            hipLaunchKernelGGL(collect_long_rows_kernel, dim3(grid), dim3(256), 0, st, ix->s, ix->e, ix->o, rows,
                               ix->ls, ix->le, ix->lo, (unsigned long long *)(ix->d_scratch + 6));
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(st));
            ix->n_long = h[1];
        }
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

}  // extern "C"

namespace {
// one workgroup per sampled block of 1024 rows: LDS histogram of the overlap byte, then its non-empty bins to HBM
__global__ __launch_bounds__(256) void len_census_kernel(const uint32_t *__restrict__ pk, uint64_t rows, int shift,
                                                         uint64_t step, unsigned int *__restrict__ hist) {
    __shared__ unsigned int h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t r = ((uint64_t)blockIdx.x * step * 256 + threadIdx.x) * 4;
    if (r < rows) {  // (pk is padded: the 16 bytes are there)
        const uint4 v = *reinterpret_cast<const uint4 *>(pk + r);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        for (int i = 0; i < 4; ++i)
            if (r + i < rows) atomicAdd(&h[(w[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}
}  // namespace


namespace {
// the same for dense rows: one workgroup per sampled block of 256 groups (1280 rows)
__global__ __launch_bounds__(256) void dense_census_kernel(const uint4 *__restrict__ p3, uint64_t rows, uint64_t step,
                                                           unsigned int *__restrict__ hist) {
    __shared__ unsigned int h[64];
    if (threadIdx.x < 64) h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t g = (uint64_t)blockIdx.x * step * 256 + threadIdx.x;
    if (5 * g < rows) {
        const uint4 v = p3[g];
        const uint32_t len[5] = {v.x & 63u, v.y & 63u, v.z & 63u, v.w & 63u, (v.x >> 16) & 63u};
        for (int i = 0; i < 5; ++i)
            if (5 * g + i < rows) atomicAdd(&h[len[i]], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 64 && h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}

// which overlap values (min(end - start, 255)) occur among ALL the rows: one pass, a flag per value in LDS (plain stores: lanes
// that hit the same flag merge), the workgroup's flags or-ed into eight words in HBM
__global__ __launch_bounds__(256) void len_seen_kernel(const uint32_t *__restrict__ pk, uint64_t rows, int shift,
                                                       unsigned int *__restrict__ seen) {
    __shared__ uint32_t flag[256];
    flag[threadIdx.x] = 0;
    __syncthreads();
    for (uint64_t r = 4 * (blockIdx.x * (uint64_t)256 + threadIdx.x); r < rows; r += 4 * (uint64_t)gridDim.x * 256) {
        const uint4 v = *reinterpret_cast<const uint4 *>(pk + r);  // (pk is padded: the 16 bytes are there)
        flag[(v.x >> shift) & 255u] = 1u;
        if (r + 1 < rows) flag[(v.y >> shift) & 255u] = 1u;
        if (r + 2 < rows) flag[(v.z >> shift) & 255u] = 1u;
        if (r + 3 < rows) flag[(v.w >> shift) & 255u] = 1u;
    }
    __syncthreads();
    // one atomic per word and workgroup (a flag per thread was 60 atomics per workgroup on the same two words of HBM: 5 * 10^5 of
    // them, one after the other in the L2 -- 5.0 of this pass's 5.0 ms on 5 * 10^8 rows)
    const unsigned long long b = __ballot(flag[threadIdx.x] != 0);
    if ((threadIdx.x & 63) == 0) {
        const unsigned int lo = (unsigned int)b, hi = (unsigned int)(b >> 32), w = threadIdx.x >> 5;
        if (lo) atomicOr(&seen[w], lo);
        if (hi) atomicOr(&seen[w + 1], hi);
    }
}
}  // namespace

int memo_len_census(memo_index *ix) {
    ix->len_hist_rows = 0;
    ix->len_seen_exact = 0;
    for (unsigned int &c : ix->len_seen) c = 0;
    for (unsigned int &c : ix->len_hist) c = 0;
    if (!ix->rows) return MEMO_OK;
    DeviceGuard guard(ix->device);
    unsigned int *d_hist = nullptr;
    if (!ix->pk && ix->p3) {
        // an index that holds the dense rows only (the builder's dense way in, memo_index_import_dense): the same sampled
        // histogram from their 6-bit overlap fields (63 = "63 or more") -- what the rule that decides when a k-class view is
        // worth its pass (memo_view.hip: view_due) estimates the view's size from
        const uint64_t drows = ix->boff3 ? ix->rows3 : ix->rows, groups = (drows + 4) / 5;
        if (!groups) return MEMO_OK;
        HIP_TRY(hipMalloc(&d_hist, sizeof(ix->len_hist)));
        const uint64_t blocks = (groups + 255) / 256, step = blocks / 4096 + 1, grid = (blocks + step - 1) / step;
        hipError_t err = hipMemsetAsync(d_hist, 0, sizeof(ix->len_hist), nullptr);
        if (err == hipSuccess) {
            hipLaunchKernelGGL(dense_census_kernel, dim3((unsigned)grid), dim3(256), 0, nullptr, reinterpret_cast<const uint4 *>(ix->p3),
                               drows, step, d_hist);
            err = hipGetLastError();
        }
        if (err == hipSuccess) err = hipMemcpy(ix->len_hist, d_hist, sizeof(ix->len_hist), hipMemcpyDeviceToHost);
        (void)hipFree(d_hist);
        if (err != hipSuccess) return fail(MEMO_EHIP, "overlap census: %s", hipGetErrorString(err));
        for (unsigned int c : ix->len_hist) ix->len_hist_rows += c;
        // (the histogram is of the dense rows: when rows that never write were left out of them, scale it to the index's rows so
        // that shares are shares of ix->rows, as they are for an index with 4-byte rows)
        if (ix->boff3 && ix->len_hist_rows) {
            const double gone = (double)(ix->rows - ix->rows3) / (double)ix->rows3;
            ix->len_hist[255] += (unsigned int)(gone * (double)ix->len_hist_rows);
            ix->len_hist_rows += (unsigned int)(gone * (double)ix->len_hist_rows);
        }
        return MEMO_OK;
    }
    if (!ix->pk || (ix->packed_fmt != 4 && ix->packed_fmt != 6 && ix->packed_fmt != 12)) return MEMO_OK;
    HIP_TRY(hipMalloc(&d_hist, sizeof(ix->len_hist)));
    const uint64_t blocks = (ix->rows + 1023) / 1024, step = blocks / 4096 + 1, grid = (blocks + step - 1) / step;
    hipError_t err = hipMemsetAsync(d_hist, 0, sizeof(ix->len_hist), nullptr);
    if (err == hipSuccess) {
        hipLaunchKernelGGL(len_census_kernel, dim3((unsigned)grid), dim3(256), 0, nullptr, ix->pk, ix->rows,
                           ix->packed_fmt == 12 ? 0 : 16, step, d_hist);
        err = hipGetLastError();
    }
    if (err == hipSuccess) err = hipMemcpy(ix->len_hist, d_hist, sizeof(ix->len_hist), hipMemcpyDeviceToHost);
    // ... and, exactly, WHICH overlaps occur (every row, not a sample): the sweeps for k - 1 >= 64 allocate, clear and fold
    // only the level arrays some row of the index can write to (memo_sweep_cons.hip: level_plan)
    if (err == hipSuccess) err = hipMemsetAsync(d_hist, 0, 32, nullptr);
    if (err == hipSuccess) {
        const uint64_t wg = (ix->rows + 1023) / 1024;
        hipLaunchKernelGGL(len_seen_kernel, dim3((unsigned)(wg < 4096 ? wg : 4096)), dim3(256), 0, nullptr, ix->pk, ix->rows,
                           ix->packed_fmt == 12 ? 0 : 16, d_hist);
        err = hipGetLastError();
    }
    if (err == hipSuccess) err = hipMemcpy(ix->len_seen, d_hist, 32, hipMemcpyDeviceToHost);
    (void)hipFree(d_hist);
    if (err != hipSuccess) return fail(MEMO_EHIP, "overlap census: %s", hipGetErrorString(err));
    for (unsigned int c : ix->len_hist) ix->len_hist_rows += c;
    ix->len_seen_exact = 1;
    return MEMO_OK;
}

extern "C" {

int memo_index_pack(memo_index_t *ix, int32_t keep_wide) {
    if (!ix) return fail(MEMO_EINVAL, "index is NULL");
    if (!ix->finalized) return fail(MEMO_ENOTREADY, "index not finalized");
    if (!ix->has_wide) {  // packed already (a builder's or an imported index): bring its rows into the query order now
        if (!ix->packed_fmt) return fail(MEMO_EINVAL, "nothing to pack");
        return ix->order_pending ? order_words_now(ix, row_order_mode(ix)) : MEMO_OK;
    }
    if (ix->rows && ix->min_s < 0) return fail(MEMO_EINVAL, "rows with a negative start cannot be packed");
    DeviceGuard guard(ix->device);
    hipStream_t st = nullptr;
    // the pass is timed on the device (info.pack_ms): SURVEY.md 8(d) wants the narrowing pass reported
    // apart from the query.  Event pair around the census and the packing kernel; allocation is outside.
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    HIP_TRY(hipEventCreate(&ev0));
    if (hipEventCreate(&ev1) != hipSuccess) {
        (void)hipEventDestroy(ev0);
        return fail(MEMO_EHIP, "hipEventCreate failed");
    }
    struct Events {
        hipEvent_t a, b;
        ~Events() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); }
    } events{ev0, ev1};
    // a packed copy of the same size is reused (packing again after a re-finalize, or to time the pass)
    drop_dense(ix);  // derived from the words that are about to be rewritten
    drop_packed_views(ix);
    const bool had = ix->pk && ix->packed_rows == ix->padded;
    if (!had) drop_packed(ix);
    ix->packed_fmt = 0;
    if (!ix->pk) HIP_TRY(hipMalloc(&ix->pk, ix->padded * sizeof(uint32_t)));
    ix->packed_rows = ix->padded;
    uint64_t h[8] = {0};
    HIP_TRY(hipEventRecord(ev0, st));
    // The layout follows from the largest annot: guessed from a sample of the annot column (every 256th row), packed at once with
    // the exact census taken on the way, packed again only when a row the sample missed needs a wider layout.
    auto layout_of = [](uint64_t top) { return top <= 255 ? 4 : (top <= 4095 ? 12 : 6); };
    int fmt = 4;
    if (ix->rows) {
        HIP_TRY(hipMemsetAsync(ix->d_scratch, 0, 64, st));
        // (~2.6 * 10^5 samples: each is a cache line of its own from a step of 16 on, and every 256th row took 0.21 ms on 5 * 10^8 rows)
        const uint64_t step = ix->rows > (1u << 20) ? (ix->rows >> 18 > 256 ? ix->rows >> 18 : 256) : 1, samples = ix->rows / step + 1;
        const unsigned grid = (unsigned)(samples / 256 + 1 < 4096 ? samples / 256 + 1 : 4096);
        hipLaunchKernelGGL(annot_census_kernel, dim3(grid), dim3(256), 0, st, ix->o, ix->rows, ix->d_scratch, step);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpy(h, ix->d_scratch, 64, hipMemcpyDeviceToHost));
        if (h[3])
            return fail(MEMO_EINVAL, "rows have an annot outside [0, 65535]: cannot be packed");
        fmt = layout_of(h[5]);
        for (int pass = 0; pass < 2; ++pass) {
            if (fmt == 6 && !ix->pa) HIP_TRY(hipMalloc(&ix->pa, ix->padded * sizeof(uint16_t)));
            HIP_TRY(hipMemsetAsync(ix->d_scratch, 0, 64, st));
            hipLaunchKernelGGL(pack_rows_kernel, dim3(4096), dim3(256), 0, st, ix->s, ix->e, ix->o, ix->rows, ix->padded, ix->pk,
                               fmt == 6 ? ix->pa : nullptr, fmt, pass == 0 ? ix->d_scratch : nullptr);
            HIP_TRY(hipGetLastError());
            if (pass) break;
            HIP_TRY(hipMemcpyAsync(h, ix->d_scratch, 64, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            if (h[3])
                return fail(MEMO_EINVAL, "%llu rows have an annot outside [0, 65535]: cannot be packed", (unsigned long long)h[3]);
            if (layout_of(h[5]) == fmt) break;
            fmt = layout_of(h[5]);  // (a row the sample missed: once more, in the layout it needs)
        }
    } else {
        hipLaunchKernelGGL(pack_rows_kernel, dim3(64), dim3(256), 0, st, ix->s, ix->e, ix->o, ix->rows, ix->padded, ix->pk, nullptr, fmt,
                           nullptr);
        HIP_TRY(hipGetLastError());
    }
    ix->max_annot = h[5];
    if (fmt != 6 && ix->pa) {
        (void)hipFree(ix->pa);
        ix->pa = nullptr;
    }
    ix->row_order = 0;
    if (const int mode = row_order_mode(ix); mode && (fmt == 4 || fmt == 12) && ix->rows) {  // the order inside a bucket (memo_interleave.hip)
        if (int rc = interleave_words(ix->pk, ix->boff, ix->nb, ix->bshift, fmt, mode, st, ix->d_scratch)) return rc;
        ix->row_order = mode;
    }
    HIP_TRY(hipEventRecord(ev1, st));
    HIP_TRY(hipStreamSynchronize(st));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, ev0, ev1));
    ix->pack_ms = ms;
    ix->packed_fmt = fmt;
    if (int rc = memo_len_census(ix)) return rc;
    if (!keep_wide) {
        (void)hipFree(ix->s);
        (void)hipFree(ix->e);
        (void)hipFree(ix->o);
        ix->s = ix->e = ix->o = nullptr;
        ix->has_wide = 0;
    }
    return MEMO_OK;
}

int memo_index_pack_dense(memo_index_t *ix, int32_t keep_packed) {
    if (!ix) return fail(MEMO_EINVAL, "index is NULL");
    if (!ix->finalized) return fail(MEMO_ENOTREADY, "index not finalized");
    if (ix->p3) {
        if (!keep_packed && ix->pk) {
            DeviceGuard guard(ix->device);
            drop_packed_views(ix);
            (void)hipFree(ix->pk);
            ix->pk = nullptr;
            ix->packed_rows = 0;
        }
        return MEMO_OK;
    }
    if (!ix->pk || (ix->packed_fmt != 4 && !(ix->packed_fmt == 12 && ix->max_annot <= 511)))
        return fail(MEMO_EINVAL, "dense rows are built from the 4-byte rows: memo_index_pack first, and every annot <= 511");
    DeviceGuard guard(ix->device);
    hipStream_t st = nullptr;
    const uint64_t groups = dense_groups_for(ix->padded);
    HIP_TRY(hipMalloc(&ix->p3, groups * 16));
    hipLaunchKernelGGL(pack3_rows_kernel, dim3(4096), dim3(256), 0, st, ix->pk, ix->padded, groups,
                       reinterpret_cast<uint4 *>(ix->p3), ix->packed_fmt == 12 ? 1 : 0);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    ix->rows3 = ix->rows;
    ix->padded3 = ix->padded;
    if (int rc = dense_compact(ix)) return rc;
    if (!keep_packed) {
        drop_packed_views(ix);
        (void)hipFree(ix->pk);
        ix->pk = nullptr;
        ix->packed_rows = 0;
    }
    return MEMO_OK;
}

static void fill_info(const memo_index *ix, memo_index_info_t *info) {
    memset(info, 0, sizeof *info);
    info->version = MEMO_INDEX_INFO_VERSION;
    info->rows = ix->rows;
    info->min_start = ix->min_s;
    info->max_start = ix->max_s;
    info->device = ix->device;
    info->bucket_shift = ix->bshift;
    info->buckets = ix->nb;
    info->was_sorted = ix->was_sorted;
    info->finalized = ix->finalized;
    info->packed_format = ix->packed_fmt;
    info->has_wide = ix->has_wide;
    info->pack_ms = ix->pack_ms;
    info->last_sweep = ix->last_sweep;
    info->last_variant = ix->last_variant;
    info->dense_rows = ix->p3 ? 1 : 0;
    info->long_rows = ix->n_long;
    info->max_annot = ix->max_annot;
    info->bucket_base = ix->bbase;
    info->device_bytes = (ix->has_wide ? ix->padded * 3 * sizeof(int64_t) : 0) + ix->nb * sizeof(int64_t) + 128 +
                         (ix->pk ? ix->padded * 4 : 0) + (ix->pa ? ix->padded * 2 : 0) +
                         (ix->p3 ? dense_groups_for(ix->boff3 ? ix->padded3 : ix->padded) * 16 : 0) + (ix->boff3 ? ix->nb * 8 : 0);
    info->dense_row_count = ix->p3 ? (ix->boff3 ? ix->rows3 : ix->rows) : 0;
    info->last_rows_read = ix->last_rows_read;
    info->last_view_ms = ix->last_view_ms;
    info->row_order = ix->row_order;
    uint64_t side = ix->retired_bytes;
    for (const memo_index::DenseView &v : ix->views) {
        side += v.p3 ? v.bytes + ix->nb * 8 : 0;
        info->views_resident += v.p3 ? 1 : 0;
    }
    for (const memo_index::DenseView &v : ix->views6) {
        side += v.p3 ? v.bytes + ix->nb * 8 : 0;
        info->views_resident += v.p3 ? 1 : 0;
    }
    for (const memo_index::DenseView &v : ix->pviews) {
        side += v.p3 ? v.bytes + ix->nb * 8 : 0;
        info->views_resident += v.p3 ? 1 : 0;
    }
    for (const memo_index::TileTable &t : ix->ttabs) side += (uint64_t)t.n * 32;
    info->tile_tables_resident = (int32_t)ix->ttabs.size();
    info->side_bytes = side;
    info->device_bytes += side;
    info->view_builds = ix->view_builds;
    info->last_level_arrays = ix->last_arrays;
    info->view_placings = ix->view_placings;
    info->last_view_placed = ix->last_view_placed;
    info->last_view_rows_per_group = ix->last_view_rpg;
}

int memo_index_get_info_v5(const memo_index_t *ix, memo_index_info_t *info) {
    if (!ix || !info) return fail(MEMO_EINVAL, "NULL argument");
    const uint32_t have = info->struct_bytes;
    if (have < 16)
        return fail(MEMO_EINVAL, "memo_index_info_t.struct_bytes = %u: set it to sizeof(memo_index_info_t) before the call", have);
    memo_index_info_t full;
    fill_info(ix, &full);
    // whole leading fields only: a size that ends inside a field is rounded down to where that field begins
#define MEMO_INFO_FIELD(f) (uint32_t) offsetof(memo_index_info_t, f)
    static const uint32_t starts[] = {
        MEMO_INFO_FIELD(struct_bytes), MEMO_INFO_FIELD(version), MEMO_INFO_FIELD(rows), MEMO_INFO_FIELD(min_start), MEMO_INFO_FIELD(max_start),
        MEMO_INFO_FIELD(device), MEMO_INFO_FIELD(bucket_shift), MEMO_INFO_FIELD(buckets), MEMO_INFO_FIELD(was_sorted), MEMO_INFO_FIELD(finalized),
        MEMO_INFO_FIELD(device_bytes), MEMO_INFO_FIELD(packed_format), MEMO_INFO_FIELD(has_wide), MEMO_INFO_FIELD(pack_ms), MEMO_INFO_FIELD(dense_rows),
        MEMO_INFO_FIELD(long_rows), MEMO_INFO_FIELD(max_annot), MEMO_INFO_FIELD(bucket_base), MEMO_INFO_FIELD(last_sweep), MEMO_INFO_FIELD(last_variant),
        MEMO_INFO_FIELD(dense_row_count), MEMO_INFO_FIELD(last_rows_read), MEMO_INFO_FIELD(last_view_ms), MEMO_INFO_FIELD(row_order),
        MEMO_INFO_FIELD(side_bytes), MEMO_INFO_FIELD(views_resident), MEMO_INFO_FIELD(tile_tables_resident), MEMO_INFO_FIELD(view_builds),
        MEMO_INFO_FIELD(last_level_arrays), MEMO_INFO_FIELD(last_view_placed), MEMO_INFO_FIELD(view_placings),
        MEMO_INFO_FIELD(last_view_rows_per_group), MEMO_INFO_FIELD(reserved), (uint32_t)sizeof(memo_index_info_t)};
#undef MEMO_INFO_FIELD
    uint32_t n = 0;
    for (uint32_t s : starts)
        if (s <= have && s > n) n = s;
    full.struct_bytes = n;
    memcpy(info, &full, n);
    return MEMO_OK;
}

int memo_index_set_option(memo_index_t *ix, int32_t option, int64_t value) {
    if (!ix) return fail(MEMO_EINVAL, "index is NULL");
    if (option == MEMO_OPT_VIEWS) {
        if (value != 0 && value != 1) return fail(MEMO_EINVAL, "MEMO_OPT_VIEWS takes 0 or 1");
        const int before = ix->views_on;
        ix->views_on = (int)value;
        if (!value) {
            DeviceGuard guard(ix->device);
            HIP_TRY(hipDeviceSynchronize());
            for (int kind = 0; kind < 2; ++kind)
            for (memo_index::DenseView &v : (kind ? ix->views6 : ix->views))
                if (v.p3) {  // (their tile tables go with them)
                    for (size_t i = 0; i < ix->ttabs.size();)
                        if (ix->ttabs[i].rows_of == v.p3) {
                            (void)hipFree(ix->ttabs[i].d);
                            ix->ttabs.erase(ix->ttabs.begin() + (long)i);
                        } else {
                            ++i;
                        }
                }
            drop_dense_views(ix);
            drop_packed_views(ix);
            flush_retired(ix);
        }
        return before;
    }
    if (option == MEMO_OPT_VIEW_BUDGET_PCT) {
        if (value < 0 || value > 1600) return fail(MEMO_EINVAL, "MEMO_OPT_VIEW_BUDGET_PCT takes 0 .. 1600");
        const int before = ix->view_budget_pct;
        ix->view_budget_pct = (int)value;
        return before;
    }
    if (option == MEMO_OPT_BUILD_COST_PCT) {
        if (value < 0 || value > 100000) return fail(MEMO_EINVAL, "MEMO_OPT_BUILD_COST_PCT takes 0 .. 100000");
        const int before = ix->build_cost_pct;
        ix->build_cost_pct = (int)value;
        return before;
    }
    if (option == MEMO_OPT_VIEW_PLACES) {
        if (value != 0 && value != 1) return fail(MEMO_EINVAL, "MEMO_OPT_VIEW_PLACES takes 0 or 1");
        const int before = ix->view_places;
        ix->view_places = (int)value;
        return before;
    }
    if (option == MEMO_OPT_VIEW_ROWS) {
        if (value != 0 && value != 5 && value != 6) return fail(MEMO_EINVAL, "MEMO_OPT_VIEW_ROWS takes 0 (the library's choice), 5 or 6");
        const int before = ix->view_rows;
        ix->view_rows = (int)value;
        return before;
    }
    return fail(MEMO_EINVAL, "unknown index option %d", option);
}

int memo_index_prepare(memo_index_t *ix, int32_t k, int32_t num_docs, int32_t membership, int64_t window_hint, void *stream,
                       uint64_t *bytes_taken) {
    if (bytes_taken) *bytes_taken = 0;
    if (!ix) return fail(MEMO_EINVAL, "index is NULL");
    if (!ix->finalized) return fail(MEMO_ENOTREADY, "index not finalized");
    if (window_hint < 0) return fail(MEMO_EINVAL, "window_hint must be >= 0");
    if (k <= 1 || !ix->rows) return MEMO_OK;  // (no row can write: nothing to build)
    DeviceGuard guard(ix->device);
    memo_index_info_t before, after;
    fill_info(ix, &before);
    // the window the queries to come are like: it starts on every kernel's tile grid, and nothing is ever written to d_out
    const int64_t top = ix->max_s < 0 ? 0 : ix->max_s;
    const int64_t len = window_hint > 0 ? window_hint : (top + 1 > 4096 ? top + 1 : 4096);
    void *const never_written = kNeverWritten;
    int rc;
    {
        struct PlanOnly {  // (the query path plans and builds, and launches nothing: reset on every way out of this scope)
            PlanOnly() { g_prepare_only = true; }
            ~PlanOnly() { g_prepare_only = false; }
        } plan_only;
        try {
            rc = membership ? memo_query_membership_dev(ix, 0, len, k, num_docs, static_cast<uint32_t *>(never_written), stream)
                 : num_docs <= 255 ? memo_query_conservation_u8_dev(ix, 0, len, k, num_docs, static_cast<uint8_t *>(never_written), stream)
                                   : memo_query_conservation_dev(ix, 0, len, k, num_docs, static_cast<uint16_t *>(never_written), stream);
        } catch (const std::exception &ex) {  // (the query path's std::vector / std::map may throw: no exception crosses the C ABI)
            rc = fail(MEMO_EHIP, "memo_index_prepare: %s", ex.what());
        }
    }
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());  // (views and tables are complete; and what the builds took out of service can go)
    flush_retired(ix);
    fill_info(ix, &after);
    if (bytes_taken) *bytes_taken = after.device_bytes > before.device_bytes ? after.device_bytes - before.device_bytes : 0;
    return MEMO_OK;
}

// The drop-in for memo_query.py:103-104 + :70: host columns in, host result out.  Rows that can be packed
// (start-sorted, start >= 0, annot in [0, 65535]: every index dap_to_bed.py writes) and k <= 256 take the
// fast way in -- narrowed on the host into pinned memory, 4-6 B/row over PCIe, PackedRows kernels
// (memo_hostpack.hip); anything else is uploaded as int64 columns and finalized on the device.
// stride 1: three columns; 3: ROWS -- filter_pq's own [M, 3] array, row-major (start = the array, end = start + 1, annot = start + 2)
static int one_shot(const int64_t *start, const int64_t *end, const int64_t *annot, uint64_t rows,
                    int64_t qs, int64_t qe, int32_t k, int32_t num_docs, void *out, int32_t device,
                    bool membership, int stride = 1) {
    if (rows && (!start || !end || !annot)) return fail(MEMO_EINVAL, "column pointer is NULL");
    const uint64_t st = (uint64_t)stride;
    memo_index_t *ix = nullptr;
    int rc = MEMO_OK;
    // MEMO_TIMING=1: phase times of the call on stderr (host clock; every phase ends synchronised)
    const bool timing = getenv("MEMO_TIMING") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
    };
    const auto t0 = now();
    const double pinned0 = pinned_alloc_ms_total();
    double in_ms[3] = {0, 0, 0};  // the last builder's device allocation / packing + copies / finish (table, census, destroy)
    if (rows && k > 1 && k - 1 <= 255 && g_one_shot_way != 1) {
        // the dense rows first (3.2 B per row over PCIe and in HBM, sweep_conservation_halo3_kernel) when they can
        // answer THIS query -- judged from the first and last start before the rows are touched, and again from the
        // largest annot once they have been packed; else (or when a row does not fit them) the 4-byte words
        const bool try_dense = g_one_shot_way != 2 &&
                               memo_dense_rows_can_answer(rows, start[0], start[(rows - 1) * st], 0, k, num_docs, membership);
        for (int dense = try_dense ? 1 : 0; dense >= 0 && !ix; --dense) {
            memo_builder_t *b = nullptr;
            const auto ta = now();
            if ((rc = memo_builder_create_rows(rows, device, 0, dense ? MEMO_ROWS_DENSE : MEMO_ROWS_PACKED, &b))) return rc;
            const auto tb = now();
            rc = stride == 1 ? memo_builder_push(b, start, end, annot, rows) : memo_builder_push_rows(b, start, rows);
            const auto tc = now();
            if (!rc) rc = memo_builder_finish(b, &ix);
            const int why = builder_why(b);
            memo_builder_destroy(b);
            in_ms[0] = ms(ta, tb), in_ms[1] = ms(tb, tc), in_ms[2] = ms(tc, now());
            if (rc == MEMO_EUNPACKABLE) {
                rc = MEMO_OK;
                ix = nullptr;
                if (dense && (why & ~16)) break;  // (unsorted, negative start, wild annot: the 4-byte words would refuse them too)
            } else if (rc) {
                return rc;
            } else if (dense && !memo_dense_rows_can_answer(ix->rows, ix->min_s, ix->max_s, ix->max_annot, k, num_docs, membership)) {
                // (the rule query_conservation applies: ALL the index's rows against its span -- rows that can never write
                // may have left the dense rows, memo_common.h: boff3 -- and the largest annot against the result matrix)
                memo_index_destroy(ix);  // (an annot outside the result matrix: the 4-byte kernels flag the reference's IndexError)
                ix = nullptr;
            }
        }
    }
    if (!ix) {
        // (rows that could not be packed on the host -- unsorted, wild annots, k > 256: the int64 columns go up and the device validates
        // and sorts.  From ROWS the three columns are made here first: the rare way, one more pass over the host's memory)
        std::vector<int64_t> cols;
        if (stride != 1 && rows) {
            try {
                cols.resize(3 * rows);
            } catch (const std::exception &) {
                return fail(MEMO_EHIP, "out of host memory for the columns of %llu rows", (unsigned long long)rows);
            }
            int64_t *cs = cols.data(), *ce = cs + rows, *ca = ce + rows;
            HostPool::get().run((int)((rows + 65535) / 65536), [&](int t) {
                const uint64_t i0 = (uint64_t)t * 65536, i1 = i0 + 65536 < rows ? i0 + 65536 : rows;
                for (uint64_t i = i0; i < i1; ++i) cs[i] = start[3 * i], ce[i] = start[3 * i + 1], ca[i] = start[3 * i + 2];
            });
            start = cs, end = ce, annot = ca;
        }
        if ((rc = memo_index_create(rows, device, &ix))) return rc;
        rc = memo_index_upload(ix, start, end, annot, rows);
        if (!rc) rc = memo_index_finalize(ix, 0, 1);
        if (rc) {
            memo_index_destroy(ix);
            return rc;
        }
    }
    const auto t1 = now();
    auto t2 = t1;
    void *d_out = nullptr;
    size_t bytes = 0;
    do {
        if (qe < qs) { rc = fail(MEMO_EINVAL, "ValueError: negative dimensions are not allowed (window end < start)"); break; }
        const int64_t L = qe - qs;
        if (L > 0 && !out) { rc = fail(MEMO_EINVAL, "output pointer is NULL"); break; }
        bytes = membership ? (size_t)L * ((num_docs + 31) / 32) * 4 : (size_t)L * 2;
        DeviceGuard guard(device);
        if (bytes) {
            hipError_t err = hipMalloc(&d_out, bytes);
            if (err != hipSuccess) { rc = fail(MEMO_EHIP, "hipMalloc(%zu): %s", bytes, hipGetErrorString(err)); break; }
        }
        rc = membership ? memo_query_membership_dev(ix, qs, qe, k, num_docs, (uint32_t *)d_out, nullptr)
                        : memo_query_conservation_dev(ix, qs, qe, k, num_docs, (uint16_t *)d_out, nullptr);
        if (rc) break;
        if ((rc = memo_query_check(ix, nullptr))) break;
        g_last_one_shot_sweep = ix->last_sweep;
        t2 = now();
        if (bytes) rc = download_pipelined(device, out, d_out, bytes, nullptr);
    } while (0);
    if (timing && !rc) {
        const auto t3 = now();
        fprintf(stderr,
                "memo one-shot: %llu rows %s: rows in %.1f ms (%.1f GB/s of int64 columns; allocation %.1f, packing + copies %.1f "
                "with %d host threads, finish %.1f; pinned slots allocated in this call %.1f), result alloc + sweep + check %.1f ms, result out %.1f ms (%.1f GB/s), total %.1f ms\n",
                (unsigned long long)rows, ix->has_wide ? "as int64 columns" : (ix->packed_fmt == 6 ? "packed to 6 B" : ix->pk ? "packed to 4 B" : "packed to 3.2 B (dense rows)"),
                ms(t0, t1), rows * 24.0 / 1e6 / (ms(t0, t1) + 1e-9), in_ms[0], in_ms[1], memo_host_threads(nullptr, nullptr), in_ms[2], pinned_alloc_ms_total() - pinned0, ms(t1, t2), ms(t2, t3),
                bytes / 1e6 / (ms(t2, t3) + 1e-9), ms(t0, t3));
    }
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

int memo_conservation_rows(const int64_t *rows3, uint64_t rows, int64_t qs, int64_t qe, int32_t k, int32_t num_docs, uint16_t *out,
                           int32_t device) {
    if (rows && !rows3) return fail(MEMO_EINVAL, "rows pointer is NULL");
    return one_shot(rows3, rows3 + 1, rows3 + 2, rows, qs, qe, k, num_docs, out, device, false, 3);
}

int memo_membership_rows(const int64_t *rows3, uint64_t rows, int64_t qs, int64_t qe, int32_t k, int32_t num_docs, uint32_t *out_bits,
                         int32_t device) {
    if (rows && !rows3) return fail(MEMO_EINVAL, "rows pointer is NULL");
    return one_shot(rows3, rows3 + 1, rows3 + 2, rows, qs, qe, k, num_docs, out_bits, device, true, 3);
}

int memo_synth_fill(memo_index_t *ix, uint64_t row_begin, uint64_t num, uint64_t den,
                    int32_t num_docs, uint64_t seed) {
    if (!ix) return fail(MEMO_EINVAL, "index is NULL");
    if (num == 0 || den == 0 || num_docs < 2) return fail(MEMO_EINVAL, "bad generator parameters");
    if (!ix->has_wide) return fail(MEMO_EINVAL, "the int64 columns were dropped");
    DeviceGuard guard(ix->device);
    drop_packed(ix);
    if (ix->rows) {
        hipLaunchKernelGGL(synth_rows_kernel, dim3(4096), dim3(256), 0, nullptr, ix->s, ix->e, ix->o,
                           ix->rows, row_begin, num, den, (uint64_t)(num_docs - 1), seed);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
    }
    ix->finalized = 0;
    return MEMO_OK;
}

int memo_bin_conservation_dev(const uint16_t *d_vec, int64_t L, const int64_t *edges, int32_t nbins,
                              int32_t num_docs, uint64_t *counts, int32_t device, void *stream) {
    if (nbins < 1 || num_docs < 1 || num_docs > 65534 || !edges || !counts || (L > 0 && !d_vec))
        return fail(MEMO_EINVAL, "bad binning arguments");
    for (int i = 0; i < nbins; ++i)
        if (edges[i] < 0 || edges[i] > edges[i + 1] || edges[i + 1] > L)
            return fail(MEMO_EINVAL, "bin edges must be non-decreasing inside [0, L]");
    DeviceGuard guard(device);
    if (!guard.ok) return fail(MEMO_EHIP, "cannot select HIP device %d", device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int ncols = num_docs + 1;
    const size_t cbytes = (size_t)nbins * ncols * sizeof(uint64_t);
    const size_t ebytes = ((size_t)(nbins + 1) * sizeof(int64_t) + 15) & ~(size_t)15;
    // edges + counts live in one scratch buffer per (thread, device) that only ever grows: `memo view`
    // bins the same vector at several resolutions, and a hipMalloc / hipFree pair per call costs more
    // than the histogram
    struct Scratch {
        char *p = nullptr;
        size_t cap = 0;
        int dev = -1;  // (never freed at thread exit: the HIP runtime may be gone by then)
    };
    thread_local Scratch scratch;
    if (scratch.dev != device || scratch.cap < ebytes + cbytes) {
        if (scratch.p) (void)hipFree(scratch.p);
        scratch.p = nullptr;
        scratch.cap = 0;
        const size_t want = ebytes + cbytes < (1u << 20) ? (1u << 20) : ebytes + cbytes;
        hipError_t err = hipMalloc(&scratch.p, want);
        if (err != hipSuccess) return fail(MEMO_EHIP, "hipMalloc(%zu): %s", want, hipGetErrorString(err));
        scratch.cap = want;
        scratch.dev = device;
    }
    int64_t *d_edges = reinterpret_cast<int64_t *>(scratch.p);
    unsigned long long *d_counts = reinterpret_cast<unsigned long long *>(scratch.p + ebytes);
    hipError_t err = hipMemcpyAsync(d_edges, edges, (size_t)(nbins + 1) * sizeof(int64_t), hipMemcpyHostToDevice, st);
    if (err == hipSuccess) err = hipMemsetAsync(d_counts, 0, cbytes, st);
    if (err == hipSuccess) {
        // enough workgroups to fill the chip, at least one per bin
        int slices = (int)((2048 + nbins - 1) / nbins);
        const int64_t longest = (L + nbins - 1) / nbins;
        while (slices > 1 && longest / slices < 4096) --slices;
        if ((size_t)ncols * 4 <= 48 * 1024)
            hipLaunchKernelGGL(bin_conservation_kernel<true>, dim3((unsigned)(nbins * slices)), dim3(256),
                               (size_t)ncols * 4, st, d_vec, d_edges, ncols, slices, d_counts);
        else
            hipLaunchKernelGGL(bin_conservation_kernel<false>, dim3((unsigned)(nbins * slices)), dim3(256), 0, st,
                               d_vec, d_edges, ncols, slices, d_counts);
        err = hipGetLastError();
    }
    if (err == hipSuccess) err = hipMemcpyAsync(counts, d_counts, cbytes, hipMemcpyDeviceToHost, st);
    if (err == hipSuccess) err = hipStreamSynchronize(st);
    if (err != hipSuccess) return fail(MEMO_EHIP, "binning failed: %s", hipGetErrorString(err));
    return MEMO_OK;
}

// wire layout: [count u32, cap u32, overflow u32, 4 B pad][nibbles: 4 * ceil(n / 8) B][exceptions: cap * 8 B]
size_t memo_transport_bytes(int64_t n, uint32_t cap) {
    return 16 + (size_t)((n + 7) / 8) * 4 + (size_t)cap * 8;
}

int memo_transport_pack_dev(const uint8_t *d_vec, int64_t n, uint32_t cap, void *d_wire, int32_t device,
                            void *stream) {
    if (n < 0 || (n > 0 && (!d_vec || !d_wire))) return fail(MEMO_EINVAL, "bad transport arguments");
    if (((uintptr_t)d_vec & 7) || ((uintptr_t)d_wire & 7)) return fail(MEMO_EINVAL, "transport buffers must be 8-byte aligned");
    DeviceGuard guard(device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    char *w = static_cast<char *>(d_wire);
    HIP_TRY(hipMemsetAsync(w, 0, 16, st));  // count = 0; the kernel fills in the capacity (no host staging)
    const int64_t groups = (n + 7) / 8;
    {
        const int64_t blocks = (groups + 256 * 16 - 1) / (256 * 16);  // 32768 positions per workgroup
        if (blocks >= ((int64_t)1 << 31)) return fail(MEMO_EINVAL, "slice too long for one launch");
        const unsigned grid = (unsigned)(blocks ? blocks : 1);  // (an empty slice still gets its header)
        hipLaunchKernelGGL(nibble_pack_kernel, dim3(grid), dim3(256), 0, st, d_vec, n,
                           reinterpret_cast<uint32_t *>(w + 16),
                           reinterpret_cast<unsigned long long *>(w + 16 + groups * 4),
                           reinterpret_cast<unsigned int *>(w), cap);
        HIP_TRY(hipGetLastError());
    }
    return MEMO_OK;
}

int memo_transport_unpack_dev(const void *d_wire, int64_t n, uint8_t *d_vec, int32_t device, void *stream) {
    if (n < 0 || (n > 0 && (!d_vec || !d_wire))) return fail(MEMO_EINVAL, "bad transport arguments");
    if (((uintptr_t)d_vec & 7) || ((uintptr_t)d_wire & 7)) return fail(MEMO_EINVAL, "transport buffers must be 8-byte aligned");
    DeviceGuard guard(device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const char *w = static_cast<const char *>(d_wire);
    const int64_t groups = (n + 7) / 8;
    if (groups) {
        const unsigned grid = (unsigned)(groups / 256 + 1 < 8192 ? groups / 256 + 1 : 8192);
        hipLaunchKernelGGL(nibble_unpack_kernel, dim3(grid), dim3(256), 0, st,
                           reinterpret_cast<const uint32_t *>(w + 16), n, d_vec);
        hipLaunchKernelGGL(nibble_exceptions_kernel, dim3(256), dim3(256), 0, st,
                           reinterpret_cast<const unsigned long long *>(w + 16 + groups * 4),
                           reinterpret_cast<const unsigned int *>(w), n, d_vec);
        HIP_TRY(hipGetLastError());
    }
    return MEMO_OK;
}

// exceptions the sender found (host value; synchronises `stream`).  More than the wire's capacity
// means the slice cannot travel in this coding.
int memo_transport_exceptions(const void *d_wire, int32_t device, void *stream, uint32_t *found, uint32_t *cap) {
    if (!d_wire || !found || !cap) return fail(MEMO_EINVAL, "NULL argument");
    DeviceGuard guard(device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    uint32_t head[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(head, d_wire, 16, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *found = head[2] ? 0xFFFFFFFFu : head[0];  // word 2: a workgroup overflowed its staging (more than an eighth exceptions)
    *cap = head[1];
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

int memo_dev_upload(int32_t device, void *dev, const void *host, size_t bytes, void *stream) {
    DeviceGuard guard(device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (bytes) HIP_TRY(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    return MEMO_OK;
}

int memo_dev_download(int32_t device, void *host, const void *dev, size_t bytes, void *stream) {
    DeviceGuard guard(device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (bytes) HIP_TRY(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return MEMO_OK;
}

}  // extern "C"
