// memo_debug.hip -- A/B switches and profiling aids (include/memo_amd_debug.h).  Linked only into
// libmemo_amd_ab.so, which is the product library plus this file: the tests that walk every tile
// shape, tests/fuzz_gpu.py, tools/ab.py and the PMC calibration pass load that one.  The product
// library exports none of this and its kernel-shape choices cannot be changed from outside.
#include "memo_amd_debug.h"
#include "memo_sweep.h"

using namespace memo;

namespace {

// PMC calibration: reads every row of the three columns exactly once with the sweep's own
// access shape (16 B per lane, 1 KiB per wave-instruction) and nothing else, so that
// FETCH_SIZE can be checked against a known byte count (24 B x padded rows) in the same run.
__global__ void stream_rows_kernel(const int64_t *s, const int64_t *e, const int64_t *o,
                                   uint64_t rows, unsigned long long *sink) {
    long long acc = 0;
    for (uint64_t i = 2 * (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x); i < rows;
         i += 2 * (uint64_t)gridDim.x * blockDim.x) {
        const longlong2 a = *reinterpret_cast<const longlong2 *>(s + i);
        const longlong2 b = *reinterpret_cast<const longlong2 *>(e + i);
        const longlong2 c = *reinterpret_cast<const longlong2 *>(o + i);
        acc += a.x ^ a.y ^ b.x ^ b.y ^ c.x ^ c.y;
    }
    if (acc == 0x7fffffffffffffffll) atomicAdd(sink, 1ull);  // keeps the loads alive
}

}  // namespace

extern "C" {

int memo_debug_set_tuning(memo_index_t *ix, int32_t tile_w, int32_t waves, int32_t membership_algo,
                          int32_t row_source, int32_t scatter) {
    if (!ix) return fail(MEMO_EINVAL, "index is NULL");
    if (tile_w != 0 && (tile_w < 256 || tile_w > 8192 || tile_w % 64))
        return fail(MEMO_EINVAL, "tile_w must be 0 or a multiple of 64 in [256, 8192] (powers of two for the clipped kernels)");
    if (waves != 0 && waves != 1 && waves != 4 && waves != 8) return fail(MEMO_EINVAL, "waves must be 0, 1, 4 or 8");
    if (membership_algo != 0 && (membership_algo < 2 || membership_algo > 4))
        return fail(MEMO_EINVAL, "membership_algo must be 0 (choose), 2 (doubling), 3 (runs) or 4 (planes)");
    if (row_source < 0 || row_source > 13 || row_source == 4 || row_source == 6 || row_source == 7 || row_source == 11 || row_source == 12)
        return fail(MEMO_EINVAL, "row_source must be 0 (library's choice: dense rows where they are resident and can answer, else "
                                 "the 4- / 6-byte rows, else the int64 columns), 1 (int64 columns), 2 (same as 0), 3 (4- / 6-byte "
                                 "rows even where the dense rows could answer), 5 (dense rows, every wave works its tile out), 8, 9, 10 or 13 "
                                 "(include/memo_amd_debug.h; 4, 6, 7, 11, 12 were round 3's persistent sweeps: profiles/r03_persistent_sweep.txt)");
    if (scatter < 0 || scatter > 5)
        return fail(MEMO_EINVAL, "scatter must be 0 (choose), 1 (clipped), 2 (unclipped, doubling levels), 3 (unclipped, radix-4 levels) "
                                 "4 (unclipped, mixed levels, every array) or 5 (mixed levels, the arrays of the library's level plan)");
    ix->tune.tile_w = tile_w;
    ix->tune.waves = waves;
    ix->tune.memb_algo = membership_algo;
    ix->tune.force_wide = row_source == 1;
    ix->tune.force_packed = row_source == 3;
    ix->tune.persistent = row_source == 5 || row_source == 10 ? 1 : row_source == 8 ? 5 : 0;
    ix->tune.no_views = row_source == 9 || row_source == 5;
    ix->tune.no_all_write = row_source == 13;
    ix->tune.scatter = scatter;
    return MEMO_OK;
}

int memo_debug_row_order(memo_index_t *ix, int32_t order) {
    if (!ix) return fail(MEMO_EINVAL, "index is NULL");
    if (order < 0 || order > 4)
        return fail(MEMO_EINVAL, "row order must be 0 (the library's), 1 (start order), 2 (chunks dealt over the starts), 3 (+ by overlap mod 32) "
                                 "or 4 (dealt over annot mod 32: the membership order)");
    ix->tune.row_order = order;
    return order_words_now(ix, row_order_mode(ix));  // (resident 4-byte rows: now; every later memo_index_pack: as asked)
}

int memo_debug_no_views(memo_index_t *ix, int32_t on) {
    if (!ix) return fail(MEMO_EINVAL, "index is NULL");
    ix->tune.no_views = on ? 1 : 0;  // (views already built stay resident: bench.py times the same index with and without)
    return MEMO_OK;
}

int memo_debug_view_colouring(int32_t on) {  // (views already built keep the order they have)
    g_view_colouring = on ? 1 : 0;
    return MEMO_OK;
}

int memo_debug_six_views(int32_t on) {
    g_six_views = on < 0 ? -1 : (on ? 1 : 0);
    return MEMO_OK;
}

int memo_debug_fail_side_allocations(int32_t on) {
    g_side_alloc_fails = on != 0;
    return MEMO_OK;
}

int memo_debug_dense_keep_all(int32_t on) {
    g_dense_keep_all = on != 0;
    return MEMO_OK;
}

int memo_debug_one_shot_way(int32_t way) {
    if (way < 0 || way > 2) return fail(MEMO_EINVAL, "one-shot way: 0 the library's, 1 int64 columns, 2 4-byte words");
    g_one_shot_way = way;
    return MEMO_OK;
}

int memo_debug_stream_rows(memo_index_t *ix, void *stream) {
    if (!ix) return fail(MEMO_EINVAL, "index is NULL");
    if (!ix->has_wide) return fail(MEMO_EINVAL, "the int64 columns were dropped");
    DeviceGuard guard(ix->device);
    hipLaunchKernelGGL(stream_rows_kernel, dim3(256 * 8), dim3(256), 0, static_cast<hipStream_t>(stream),
                       ix->s, ix->e, ix->o, ix->rows & ~(uint64_t)1,
                       reinterpret_cast<unsigned long long *>(ix->d_scratch));
    HIP_TRY(hipGetLastError());
    return MEMO_OK;
}

int memo_debug_last_one_shot_sweep(void) { return g_last_one_shot_sweep; }

int memo_debug_set_stamp_buffer(uint64_t *d_buffer) {
#ifdef MEMO_STAMPS
    g_stamp_buffer = reinterpret_cast<unsigned long long *>(d_buffer);
    return MEMO_OK;
#else
    (void)d_buffer;
    return fail(MEMO_EINVAL, "this build carries no phase stamps (compile with EXTRA=-DMEMO_STAMPS)");
#endif
}

}  // extern "C"
