// memo_interleave.hip -- the order of the 4-byte rows INSIDE a start bucket (round 4).
//
// The sweeps never rely on the order of the rows inside a bucket: a tile's row slice begins and ends at bucket
// boundaries (memo_sweep.h: row_slice), rows outside it are masked by row number, and min / and are commutative
// (/root/reference/src/memo_query.py:60-62 visits the rows in file order and sets one bool per covered cell: any order
// gives the same matrix).  What the order does decide is what a wave's 64 LDS atomics of one row instruction collide
// on.  Measured on gfx950 (tools/lds_atomic_bench.hip, profiles/r04_lds_atomics.txt): a ds_min_u32 / ds_or_b32
// wave-instruction costs 4.3 cycles of its CU's LDS pipe when its lanes hit different banks; lanes on one bank with
// DIFFERENT addresses cost one cycle each per half-wave (two of them ride in the instruction's own two cycles: a
// 2-way bank conflict is free); lanes on the SAME address cost two cycles each (k lanes: 2k - 1 cycles per half-wave;
// all 64 on one cell: 128).  In start order the rows of one pivot position sit in neighbouring lanes, and the second
// block of a conservation row -- the cell `start - 2^j` of level j -- is the same cell for every row of that position
// on that level: BASELINE config 5 (25 rows per position) pays 26 cycles per wave-instruction there, six times the
// conflict-free price, and its k = 101 sweep was bound by exactly that (0.77 ms; the rows stream in 0.57).
//
// The order built here, per bucket: the rows of every start are cut into chunks of four (one lane's 16-byte load) and
// the chunks are dealt round-robin over the bucket's starts -- pass 0 holds the first chunk of every start, in start
// order, pass 1 the second, ... -- so that the 32 lanes of a half-wave hold 32 different, consecutive starts: their
// second blocks fall on 32 different cells in 32 different banks.  Inside a start the rows are ordered by their overlap
// mod 32 (then overlap, then annot: the order is a function of the bucket's rows alone, whatever order they arrive
// in), so that the q-th chunk of every start holds that start's q-th quartet of overlaps: the first block of a row lies
// at cell `start - (k - 1) + overlap`, and lanes whose starts step by one and whose overlaps agree mod 32 up to a few
// positions spread over the banks better than the pseudo-random overlaps of start order do.
// Config-5 shard, k = 101, sustained (profiles/r04_row_order.txt): 0.771 -> 0.60-0.62 ms; k = 31: 0.334 -> 0.310.
//
// In place: a bucket is staged in LDS before a word of it is written back, and buckets are disjoint.  Buckets of more
// than kMaxBucketRows rows stay as they are (correct, only slower).  Formats 4 and 12 (one word per row).
//
// Sort key of a row inside its bucket (25 + bshift - 5 bits): start in bucket | x | annot, x = the overlap byte -- in
// mode 2 rotated so that overlap mod 32 leads (x = (ov & 31) << 3 | ov >> 5).  The key and the bucket's common start
// bits give the word back, so only keys are staged.
#include "memo_common.h"

namespace memo {

namespace {

constexpr int kMaxBucketRows = 8192;  // rows of a bucket a workgroup can stage (32 KiB of keys)
constexpr int kWaveRows = 256;        // ... and one wave alone (default bucket width: four buckets per workgroup at a time)
constexpr int kWaveQ = 16;            // wave path: chunk table for starts of up to 4 * kWaveQ rows (else: the counting loop)

struct KeyCodec {
    int fmt12, mode, bshift;
    int cshift;  // the class (what the chunks are dealt over) sits at key >> cshift: the start in its bucket (modes 0-2), annot mod 32 (mode 3)
    __device__ __forceinline__ uint32_t key(uint32_t w) const {
        const uint32_t start = fmt12 ? (w >> 8) & 0xFFFu : w & 0xFFFFu;
        const uint32_t ov = fmt12 ? w & 0xFFu : (w >> 16) & 0xFFu;
        const uint32_t annot = fmt12 ? w >> 20 : w >> 24;
        const uint32_t s = start & ((1u << bshift) - 1u);
        if (mode == 3)  // class = annot mod 32 | end in the bucket (9 bits) | annot | start in the bucket (bshift == 5)
            return ((annot & 31u) << 26) | ((s + ov) << 17) | (annot << 5) | s;
        const uint32_t x = mode == 2 ? ((ov & 31u) << 3) | (ov >> 5) : ov;
        return (s << 20) | (x << 12) | annot;
    }
    // high = the start field's bits above the bucket (the same for every row of a bucket)
    __device__ __forceinline__ uint32_t word(uint32_t k, uint32_t high) const {
        uint32_t start, ov, annot;
        if (mode == 3) {
            const uint32_t s = k & 31u;
            start = high | s;
            annot = (k >> 5) & 0xFFFu;
            ov = ((k >> 17) & 0x1FFu) - s;
        } else {
            const uint32_t x = (k >> 12) & 0xFFu;
            start = high | (k >> 20);
            annot = k & 0xFFFu;
            ov = mode == 2 ? ((x & 7u) << 5) | (x >> 3) : x;
        }
        return fmt12 ? ov | (start << 8) | (annot << 20) : start | (ov << 16) | (annot << 24);
    }
    __device__ __forceinline__ uint32_t high_of(uint32_t w) const {
        const uint32_t start = fmt12 ? (w >> 8) & 0xFFFu : w & 0xFFFFu;
        return start & ~((1u << bshift) - 1u);
    }
};

// LDS written by some lanes of ONE wave, read by others of the same wave: LDS operations of a wave execute in order;
// this keeps the compiler from moving or caching them across the step (s_waitcnt lgkmcnt(0), no s_barrier)
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

constexpr uint32_t kCountedClass = 96;  // rows of one class a row's place is counted through (longer classes: the sort)

// the place of row e (key k) among the c rows of its class, which lie at key[f .. f + c) in any order: the rows with a smaller
// key, or an equal one further left
__device__ __forceinline__ uint32_t counted_rank(const uint32_t *key, uint32_t f, uint32_t c, uint32_t e, uint32_t k) {
    uint32_t rank = 0;
    for (uint32_t j = f; j < f + c; ++j) {
        const uint32_t z = key[j];
        rank += (z < k || (z == k && j < e)) ? 1u : 0u;
    }
    return rank;
}

// ---- one wave, one bucket of at most 256 rows, 32 starts (the default bucket width) -------------------------------
__device__ __forceinline__ void wave_bucket(uint32_t *__restrict__ words, int64_t r0, int R, const KeyCodec &C, uint32_t *key /*[256]*/,
                                            uint32_t *first /*[32]*/, uint32_t *cnt /*[32]*/, uint32_t *table /*[kWaveQ * 32]*/) {
    const int lane = threadIdx.x & 63;
    int P = 64;
    while (P < R) P <<= 1;
    uint32_t high = 0;
    for (int e = lane; e < P; e += 64) {
        uint32_t k = 0xFFFFFFFFu;
        if (e < R) {
            const uint32_t w = words[r0 + e];
            k = C.key(w);
            high = C.high_of(w);
        }
        key[e] = k;
    }
    high = (uint32_t)__builtin_amdgcn_readfirstlane((int)high);  // (lane 0 always holds a row: R >= 2)
    if (lane < 32) cnt[lane] = 0;
    wave_sync();
    // Rows as memo_index_pack leaves them come grouped by class already (start order): then a row's place in its class is
    // COUNTED (the rows of its class with a smaller key: a handful of LDS reads) instead of sorting the bucket -- config 3's
    // 5 * 10^8 rows: 14.2 ms of ordering pass with the sort (profiles/r04_pack_pass.txt).  Rows that were dealt before (a change of order) take the sort.
    bool in_order = true;
    for (int e = lane; e < R; e += 64)
        if (e && (key[e - 1] >> C.cshift) > (key[e] >> C.cshift)) in_order = false;
    bool grouped = __ballot(!in_order) == 0ull;
    auto sort_keys = [&]() {
        for (int size = 2; size <= P; size <<= 1) {
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                for (int c = lane; c < (P >> 1); c += 64) {
                    const int lo = 2 * c - (c & (stride - 1)), hi = lo + stride;
                    const bool up = (lo & size) == 0;
                    const uint32_t a = key[lo], z = key[hi];
                    if ((a > z) == up) {
                        key[lo] = z;
                        key[hi] = a;
                    }
                }
                wave_sync();
            }
        }
    };
    if (!grouped) sort_keys();
    for (int e = lane; e < R; e += 64) {
        const uint32_t s = key[e] >> C.cshift;
        if (e == 0 || (key[e - 1] >> C.cshift) != s) first[s] = (uint32_t)e;
        if (e == R - 1 || (key[e + 1] >> C.cshift) != s) cnt[s] = (uint32_t)e + 1u;  // (its end, for now)
    }
    wave_sync();
    uint32_t c_mine = 0;
    if (lane < 32 && cnt[lane]) {
        c_mine = cnt[lane] - first[lane];
        cnt[lane] = c_mine;
    }
    wave_sync();
    uint32_t maxc = c_mine;
    for (int off = 16; off > 0; off >>= 1) maxc = max(maxc, (uint32_t)__shfl_xor((int)maxc, off, 64));
    maxc = (uint32_t)__builtin_amdgcn_readfirstlane((int)maxc);
    if (grouped && maxc > kCountedClass) {  // (a class too long to count through: the sort; the classes stay where they are)
        sort_keys();
        grouped = false;
    }
    const bool tabled = C.mode != 0 && maxc <= 4u * kWaveQ;
    if (tabled) {
        // where every chunk (pass q, start s) begins: an exclusive scan of the chunk sizes in (q, s) order, two passes at a time
        const uint32_t c_s = (uint32_t)__shfl((int)c_mine, lane & 31, 64);
        uint32_t running = 0;
        for (uint32_t q2 = 0; 8u * q2 < maxc; ++q2) {
            const uint32_t q4 = 4u * (2u * q2 + (uint32_t)(lane >> 5));
            const uint32_t v = c_s > q4 ? (c_s - q4 < 4u ? c_s - q4 : 4u) : 0u;
            uint32_t inc = v;
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t t = (uint32_t)__shfl_up((int)inc, off, 64);
                if (lane >= off) inc += t;
            }
            table[64u * q2 + (uint32_t)lane] = running + inc - v;
            running += (uint32_t)__shfl((int)inc, 63, 64);
        }
        wave_sync();
    }
    for (int e = lane; e < R; e += 64) {
        const uint32_t k = key[e];
        uint32_t pos = (uint32_t)e;  // mode 0: the sorted order itself
        if (C.mode || grouped) {
            const uint32_t s = k >> C.cshift;
            uint32_t rank = (uint32_t)e - first[s];
            if (grouped) rank = counted_rank(key, first[s], cnt[s], (uint32_t)e, k);
            if (!C.mode) {
                pos = first[s] + rank;
            } else if (tabled) {
                pos = table[32u * (rank >> 2) + s] + (rank & 3u);
            } else {
                const uint32_t q4 = rank & ~3u;
                pos = rank & 3u;
                for (uint32_t t = 0; t < 32u; ++t) {
                    const uint32_t c = cnt[t];
                    pos += c < q4 ? c : q4;
                    if (t < s) {
                        const uint32_t left = c > q4 ? c - q4 : 0u;
                        pos += left < 4u ? left : 4u;
                    }
                }
            }
        }
        words[r0 + pos] = C.word(k, high);
    }
    wave_sync();  // (key / first / cnt / table are reused by this wave's next bucket)
}

// ---- two buckets per wave, a LANE per start (rows that arrive in start order, at most eight per start) -------------
// What the pass costs is instructions (wave_bucket: ~500 per bucket of 160 rows, most of them in loops whose lanes are a
// third idle).  Where the rows of a bucket come grouped by start and no start has more than eight -- BASELINE config 3: five
// each -- lane s owns start s: its (up to eight) keys in registers, ordered by a sorting network, its two chunks of four
// placed by two 32-lane prefix sums, written as 16-byte pieces that follow one another along the lanes (coalesced).
// ~130 instructions per bucket.  Anything else -- a start with more rows, rows out of start order, the membership order
// (its classes are not starts) -- returns false and takes wave_bucket.
__device__ __forceinline__ void sort8(uint32_t (&k)[8]) {  // (Batcher's odd-even merge sort: 19 compare-exchanges)
    auto cx = [&](int a, int b) {
        const uint32_t lo = k[a] < k[b] ? k[a] : k[b], hi = k[a] < k[b] ? k[b] : k[a];
        k[a] = lo;
        k[b] = hi;
    };
    cx(0, 1); cx(2, 3); cx(4, 5); cx(6, 7);
    cx(0, 2); cx(1, 3); cx(4, 6); cx(5, 7);
    cx(1, 2); cx(5, 6);
    cx(0, 4); cx(1, 5); cx(2, 6); cx(3, 7);
    cx(2, 4); cx(3, 5);
    cx(1, 2); cx(3, 4); cx(5, 6);
}

// buckets b and b + 1: rows [r0, r1) and [r1, r2); key[] holds 512 words, cnt / first 64 each.  Returns false (nothing
// written) when the pair does not qualify.
__device__ __forceinline__ bool lane_pair(uint32_t *__restrict__ words, int64_t r0, int64_t r1, int64_t r2, const KeyCodec &C, uint32_t *key,
                                          uint32_t *first, uint32_t *cnt, uint32_t *high2) {
    const int lane = threadIdx.x & 63;
    const int R = (int)(r2 - r0), RA = (int)(r1 - r0);
    cnt[lane] = 0;
    first[lane] = 0xFFFFFFFFu;
    wave_sync();
    bool ok = true;
    for (int e = lane; e < ((R + 63) & ~63); e += 64) {
        uint32_t k = 0xFFFFFFFFu, prev = 0;
        if (e < R) {
            const uint32_t w = words[r0 + e];
            k = C.key(w);
            if (e == 0) high2[0] = C.high_of(w);
            if (e == RA) high2[1] = C.high_of(w);
            key[e] = k;
            const uint32_t cls = (k >> 20) + (e >= RA ? 32u : 0u);
            atomicAdd(&cnt[cls], 1u);
            atomicMin(&first[cls], (uint32_t)e);
        }
        // start order inside a bucket: a row's start is not below the one before it (same bucket)
        prev = (uint32_t)__shfl_up((int)k, 1, 64);
        if (e < R && (e & 63) && e != RA && (prev >> 20) > (k >> 20)) ok = false;
    }
    wave_sync();
    // (the rows at the seams of the 64-row rounds: checked through LDS)
    for (int e = 64 + lane * 64; e < R; e += 64 * 64)
        if (e != RA && (key[e - 1] >> 20) > (key[e] >> 20)) ok = false;
    const uint32_t c = cnt[lane], f = first[lane];
    // grouped by start: the rows of a class are consecutive -- first + count covers them iff no other class lies between, which
    // start order guarantees; eight at most
    if (c > 8) ok = false;
    if (__ballot(!ok)) return false;
    uint32_t k[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) k[i] = (uint32_t)i < c ? key[f + i] : 0xFFFFFFFFu;
    sort8(k);
    // chunks: pass 0 holds the first four rows of every start, pass 1 the rest; inside a pass, start order
    const uint32_t v0 = c < 4 ? c : 4, v1 = c > 4 ? c - 4 : 0;
    uint32_t s0 = v0, s1 = v1;
    for (int off = 1; off < 32; off <<= 1) {
        const uint32_t t0 = (uint32_t)__shfl_up((int)s0, off, 32), t1 = (uint32_t)__shfl_up((int)s1, off, 32);
        if ((lane & 31) >= off) {
            s0 += t0;
            s1 += t1;
        }
    }
    const uint32_t all0 = (uint32_t)__shfl((int)s0, 31, 32);  // rows of the bucket in pass 0
    const int64_t base = lane < 32 ? r0 : r1;
    const uint32_t high = high2[lane >> 5];
    uint32_t *d0 = words + base + (s0 - v0), *d1 = words + base + all0 + (s1 - v1);
    wave_sync();  // (every key is in registers: the rows may be overwritten)
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if ((uint32_t)i < v0) d0[i] = C.word(k[i], high);
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if ((uint32_t)i < v1) d1[i] = C.word(k[4 + i], high);
    return true;
}

// ---- a whole workgroup, one bucket of up to kMaxBucketRows rows, any bucket width ---------------------------------
__device__ __forceinline__ void block_bucket(uint32_t *__restrict__ words, int64_t r0, int R, const KeyCodec &C, uint32_t *key,
                                             uint32_t *first /*[256]*/, uint32_t *cnt /*[256]*/, uint32_t *shared_high) {
    const int tid = threadIdx.x;
    const int S = 1 << C.bshift;
    int P = 2;
    while (P < R) P <<= 1;
    for (int i = tid; i < P; i += 256) {
        uint32_t k = 0xFFFFFFFFu;
        if (i < R) {
            const uint32_t w = words[r0 + i];
            k = C.key(w);
            if (i == 0) *shared_high = C.high_of(w);
        }
        key[i] = k;
    }
    for (int i = tid; i < S; i += 256) cnt[i] = 0;
    __syncthreads();
    const uint32_t high = *shared_high;
    bool in_order = true;  // (as in wave_bucket: rows that come grouped by class are counted into place, not sorted)
    for (int i = tid; i < R; i += 256)
        if (i && (key[i - 1] >> C.cshift) > (key[i] >> C.cshift)) in_order = false;
    bool grouped = __syncthreads_and(in_order ? 1 : 0) != 0;
    auto sort_keys = [&]() {
        for (int size = 2; size <= P; size <<= 1) {
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                for (int c = tid; c < (P >> 1); c += 256) {
                    const int lo = 2 * c - (c & (stride - 1)), hi = lo + stride;
                    const bool up = (lo & size) == 0;
                    const uint32_t a = key[lo], z = key[hi];
                    if ((a > z) == up) {
                        key[lo] = z;
                        key[hi] = a;
                    }
                }
                __syncthreads();
            }
        }
    };
    if (!grouped) sort_keys();
    for (int i = tid; i < R; i += 256) {
        const uint32_t s = key[i] >> C.cshift;
        if (i == 0 || (key[i - 1] >> C.cshift) != s) first[s] = (uint32_t)i;
        if (i == R - 1 || (key[i + 1] >> C.cshift) != s) cnt[s] = (uint32_t)i + 1u;
    }
    __syncthreads();
    uint32_t longest = 0;
    for (int i = tid; i < S; i += 256)
        if (cnt[i]) {
            cnt[i] -= first[i];
            longest = longest > cnt[i] ? longest : cnt[i];
        }
    if (grouped && !__syncthreads_and(longest <= kCountedClass ? 1 : 0)) {
        sort_keys();
        grouped = false;
    }
    __syncthreads();
    for (int i = tid; i < R; i += 256) {
        const uint32_t k = key[i];
        uint32_t pos = (uint32_t)i;
        if (C.mode || grouped) {
            const uint32_t s = k >> C.cshift;
            const uint32_t rank = grouped ? counted_rank(key, first[s], cnt[s], (uint32_t)i, k) : (uint32_t)i - first[s], q4 = rank & ~3u;
            pos = rank & 3u;
            if (!C.mode) {
                words[r0 + first[s] + rank] = C.word(k, high);
                continue;
            }
            for (int t = 0; t < S; ++t) {
                const uint32_t c = cnt[t];
                pos += c < q4 ? c : q4;  // rows of start t in earlier passes
                if ((uint32_t)t < s) {
                    const uint32_t left = c > q4 ? c - q4 : 0u;
                    pos += left < 4u ? left : 4u;  // this pass, starts before s
                }
            }
        }
        words[r0 + pos] = C.word(k, high);
    }
    __syncthreads();  // (the staging arrays are reused by the next bucket)
}

// mode 0: back to start order (start, overlap, annot); 1: chunks of four dealt over the starts, rows of a start by
// (overlap, annot); 2: the same with the rows of a start by (overlap mod 32, overlap, annot); 3: the MEMBERSHIP order --
// chunks of four dealt over the 32 classes of annot mod 32, the rows of a class by their end (then annot, start): a membership
// sweep's ds_or goes to plane row `annot` (odd pitch: 32 consecutive residues are 32 banks) at the word of the run's first
// bit, end - (k - 1), so the 32 lanes of a half-wave -- 32 residues, ends of one quantile of the bucket -- hit 32 banks
//
// Two kernels share a launch's buckets, eight at a time ("a turn"): the turns whose eight buckets are small (<= 256 rows each,
// buckets of 32 positions: every index of up to ~8 rows per position) go to interleave_small_kernel -- two buckets per WAVE (a lane per start where no start has more than eight
// rows: lane_pair; else a bucket at a time: wave_bucket), 4.6 KiB of LDS per wave, so that a CU holds 32 waves of them -- and all the others to interleave_large_kernel, a bucket per workgroup
// with 32 KiB of keys.  (Round 4 had one kernel with the large one's LDS: 12 waves per CU; the pass waits on LDS round trips
// and dependent steps, and BASELINE config 3's 3.1 * 10^6 buckets took 6.1 ms.)  Each kernel skips the other's turns.
constexpr int kTurn = 8;  // buckets a workgroup takes at a time: two per wave

__device__ __forceinline__ bool small_turn(const int64_t (&r)[kTurn + 1], int bshift) {
    int64_t longest = 0;
    for (int j = 0; j < kTurn; ++j) longest = r[j + 1] - r[j] > longest ? r[j + 1] - r[j] : longest;
    return bshift == 5 && longest <= kWaveRows;
}

__global__ __launch_bounds__(256) void interleave_small_kernel(uint32_t *__restrict__ words, const int64_t *__restrict__ boff,
                                                               int64_t nbuckets, int bshift, int fmt12, int mode, unsigned int *left_over) {
    __shared__ uint32_t key[4][2 * kWaveRows];
    __shared__ uint32_t first[4][64], cnt[4][64];
    __shared__ uint32_t table[4][kWaveQ * 32];
    __shared__ uint32_t high2[4][2];
    KeyCodec C;
    C.fmt12 = fmt12;
    C.mode = mode;
    C.bshift = bshift;
    C.cshift = mode == 3 ? 26 : 20;
    const int wave = threadIdx.x >> 6;
    for (int64_t b0 = kTurn * (int64_t)blockIdx.x; b0 < nbuckets; b0 += kTurn * (int64_t)gridDim.x) {
        int64_t r[kTurn + 1];
        for (int j = 0; j <= kTurn; ++j) r[j] = boff[b0 + j < nbuckets ? b0 + j : nbuckets];
        if (!small_turn(r, bshift)) {  // (workgroup-uniform; the waves never meet at a barrier here)
            if (threadIdx.x == 0 && *left_over == 0) atomicOr(left_over, 1u);  // (interleave_large_kernel has something to do)
            continue;
        }
        const int64_t ra = r[2 * wave], rb = r[2 * wave + 1], rc = r[2 * wave + 2];
        if (rc - ra < 2) continue;
        if ((mode == 1 || mode == 2) && lane_pair(words, ra, rb, rc, C, key[wave], first[wave], cnt[wave], high2[wave])) continue;
        if (rb - ra >= 2) wave_bucket(words, ra, (int)(rb - ra), C, key[wave], first[wave], cnt[wave], table[wave]);
        if (rc - rb >= 2) wave_bucket(words, rb, (int)(rc - rb), C, key[wave], first[wave], cnt[wave], table[wave]);
    }
}

__global__ __launch_bounds__(256) void interleave_large_kernel(uint32_t *__restrict__ words, const int64_t *__restrict__ boff,
                                                               int64_t nbuckets, int bshift, int fmt12, int mode, const unsigned int *left_over) {
    if (left_over && *left_over == 0) return;  // (every turn was a small one: 0.47 ms of skipping for config 3's 3.9 * 10^5 turns otherwise)
    __shared__ uint32_t key[kMaxBucketRows];
    __shared__ uint32_t first[256], cnt[256];
    __shared__ uint32_t shared_high;
    KeyCodec C;
    C.fmt12 = fmt12;
    C.mode = mode;
    C.bshift = bshift;
    C.cshift = mode == 3 ? 26 : 20;
    for (int64_t b0 = kTurn * (int64_t)blockIdx.x; b0 < nbuckets; b0 += kTurn * (int64_t)gridDim.x) {
        int64_t r[kTurn + 1];
        for (int j = 0; j <= kTurn; ++j) r[j] = boff[b0 + j < nbuckets ? b0 + j : nbuckets];
        if (small_turn(r, bshift)) continue;
        for (int j = 0; j < kTurn; ++j) {
            const int64_t R = r[j + 1] - r[j];
            if (R >= 2 && R <= kMaxBucketRows) block_bucket(words, r[j], (int)R, C, key, first, cnt, &shared_high);
        }
    }
}

}  // namespace

// words: rows of formats 4 / 12, boff: their bucket table (nb entries, the last pinned to the row count).  Queued on st.
int interleave_words(uint32_t *words, const int64_t *boff, uint64_t nb, int bshift, int fmt, int mode, hipStream_t st, uint64_t *scratch) {
    if (!words || !boff || nb < 2 || (fmt != 4 && fmt != 12) || bshift < 1 || bshift > 8) return MEMO_OK;
    if (mode == 3 && bshift != 5) mode = 2;  // (the membership order's key is laid out for 32 starts per bucket)
    const int64_t nbuckets = (int64_t)nb - 1;
    const int64_t turns = (nbuckets + kTurn - 1) / kTurn;
    const unsigned grid = (unsigned)(turns < 256 * 64 ? turns : 256 * 64);
    unsigned int *left_over = scratch && bshift == 5 ? reinterpret_cast<unsigned int *>(scratch + 7) : nullptr;  // (the index's scratch words)
    if (bshift == 5) {
        if (left_over) HIP_TRY(hipMemsetAsync(left_over, 0, 4, st));
        if (!left_over) return fail(MEMO_EINVAL, "interleave_words needs the index's scratch words");
        hipLaunchKernelGGL(interleave_small_kernel, dim3(grid), dim3(256), 0, st, words, boff, nbuckets, bshift, fmt == 12 ? 1 : 0, mode, left_over);
    }
    hipLaunchKernelGGL(interleave_large_kernel, dim3(grid < 256 * 16 ? grid : 256 * 16), dim3(256), 0, st, words, boff, nbuckets, bshift,
                       fmt == 12 ? 1 : 0, mode, (const unsigned int *)left_over);
    HIP_TRY(hipGetLastError());
    return MEMO_OK;
}

}  // namespace memo
