// memo_sort.hip -- sort an index's rows by start on the device.
//
// Only reached when memo_index_finalize() finds rows that are NOT start-sorted, which no
// index written by the reference's dap_to_bed.py is (rows are emitted in pivot order,
// dap_to_bed.py:119-124).  The reference itself never needs an order (memo_query.py:61-62
// visits rows in any order and min / and do not care), so any permutation that sorts the
// start column gives the same query results; stability is irrelevant.
//
// The key sort is rocPRIM's radix sort (a plain library sort, off the hot path, kept in its
// own translation unit because the headers are slow to compile); the gather is ours.
#include <cstring>  // rocprim's texture iterator calls memset without including it

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <cstdint>
#include <cstdio>

namespace {

__global__ void iota_kernel(uint64_t *p, uint64_t n) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n;
         i += (uint64_t)gridDim.x * blockDim.x)
        p[i] = i;
}

__global__ void gather_kernel(const int64_t *src, const uint64_t *perm, int64_t *dst, uint64_t n) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n;
         i += (uint64_t)gridDim.x * blockDim.x)
        dst[i] = src[perm[i]];
}

#define SORT_TRY(expr)                                                                   \
    do {                                                                                 \
        hipError_t err__ = (expr);                                                       \
        if (err__ != hipSuccess) {                                                       \
            snprintf(err, errcap, "%s: %s", #expr, hipGetErrorString(err__));            \
            rc = -1;                                                                     \
            goto done;                                                                   \
        }                                                                                \
    } while (0)

}  // namespace

extern "C" __attribute__((visibility("hidden"))) int memo_sort_rows_by_start(int64_t *s, int64_t *e, int64_t *o, uint64_t rows,
                                       uint64_t padded_rows, hipStream_t stream, char *err,
                                       size_t errcap) {
    (void)padded_rows;
    int rc = 0;
    int64_t *keys = nullptr, *tmpcol = nullptr;
    uint64_t *iota = nullptr, *perm = nullptr;
    void *tmp = nullptr;
    size_t tmp_bytes = 0;
    const size_t col = rows * sizeof(int64_t);
    const unsigned grid = (unsigned)(rows / 256 + 1 < 8192 ? rows / 256 + 1 : 8192);
    SORT_TRY(hipMalloc(&keys, col));
    SORT_TRY(hipMalloc(&tmpcol, col));
    SORT_TRY(hipMalloc(&iota, col));
    SORT_TRY(hipMalloc(&perm, col));
    hipLaunchKernelGGL(iota_kernel, dim3(grid), dim3(256), 0, stream, iota, rows);
    SORT_TRY(hipGetLastError());
    SORT_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, s, keys, iota, perm, rows, 0, 64, stream));
    SORT_TRY(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16));
    SORT_TRY(rocprim::radix_sort_pairs(tmp, tmp_bytes, s, keys, iota, perm, rows, 0, 64, stream));
    SORT_TRY(hipMemcpyAsync(s, keys, col, hipMemcpyDeviceToDevice, stream));
    hipLaunchKernelGGL(gather_kernel, dim3(grid), dim3(256), 0, stream, e, perm, tmpcol, rows);
    SORT_TRY(hipGetLastError());
    SORT_TRY(hipMemcpyAsync(e, tmpcol, col, hipMemcpyDeviceToDevice, stream));
    hipLaunchKernelGGL(gather_kernel, dim3(grid), dim3(256), 0, stream, o, perm, tmpcol, rows);
    SORT_TRY(hipGetLastError());
    SORT_TRY(hipMemcpyAsync(o, tmpcol, col, hipMemcpyDeviceToDevice, stream));
    SORT_TRY(hipStreamSynchronize(stream));
done:
    (void)hipFree(keys);
    (void)hipFree(tmpcol);
    (void)hipFree(iota);
    (void)hipFree(perm);
    (void)hipFree(tmp);
    return rc;
}
