// memo_transport.hip -- second transport coding of uint8 conservation results for the multi-GPU
// gather.  The sweep produces a slice faster than a peer's xGMI link to the
// root carries it as bytes, so what a slice costs on the wire decides N > 1.
//
// Conservation values are minima over the rows that cover a position: small values dominate
// (config 3: 32 %, 22 %, 15 % for 1, 2, 3; entropy 2.8 bits).  "Dense" coding:
//   stream A   2 bits per position: 1, 2, 3 = the value; 0 = escape              1024 B per 4096 positions
//   stream B   one nibble per escape, in position order: 0 = value 0, 1..14 = value 4..17,
//              15 = see the exception list.  A workgroup codes a block of 32768 positions, collects
//              its nibbles in LDS and takes exactly the bytes it needs from the B region with one
//              atomic add; a table holds every block's offset and nibble count (8 B per block), so
//              the order of the blocks inside B does not matter.
//   exceptions position << 8 | value for nibble 15 (8 B each), collected per workgroup in LDS
// One pass in both directions, no global scan.  Neighbouring positions share their covering rows,
// so the escapes of a chunk vary far more than a binomial would (config 3: mean 1254, up to 1664
// per 4096 positions): fixed-size slots sized for the worst chunk would give the saving away,
// which is why B is allocated exactly.  The caller sizes the B region and the exception list for
// its data from one trial pack (memo_transport_dense_stats) and checks that both were enough.
//
// Not on the reference's path (the reference has one process and no wire): it carries the input
// of print_res (memo_query.py:65-71) from the ranks that computed it to the one that prints it.
#include "memo_common.h"

using namespace memo;

namespace {

constexpr int kChunk = 4096;     // positions per round of a workgroup
constexpr int kPerThread = 16;   // positions per thread and round (one 16-byte load / store)
constexpr int kBlock = 8 * kChunk;  // positions per workgroup: one allocation in the B region
constexpr int kHeld = 1024;  // exceptions a workgroup collects in LDS (more than that: one global atomic each)

constexpr int kRounds = kBlock / kChunk;

// inclusive prefix of v over the 64 lanes of the wave
__device__ __forceinline__ int wave_inclusive_scan(int v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// Both kernels have the same shape: a workgroup of 256 threads owns a block of 32768 positions as
// 8 rounds of 4096 (thread t: 16 consecutive positions per round, so loads and stores are whole
// 16-byte pieces, 1 KiB per wave-instruction).  Phase 1 touches all 8 rounds at once -- 8 loads in
// flight per lane -- and leaves every (round, wave)'s escape count in LDS; one barrier; phase 2
// ranks each thread's escapes inside the block (position order = round, wave, lane) from those
// 32 counts and its own wave prefix.
//
// head: [0] exceptions found, [1] their capacity, [2] B bytes taken, [3] B capacity
__global__ __launch_bounds__(256) void dense_pack_kernel(const uint8_t *in, int64_t n, uint32_t *A, uint8_t *B,
                                                         uint2 *table, unsigned long long *exc,
                                                         unsigned int *head, unsigned int cap, unsigned int b_cap) {
    __shared__ uint32_t nib[kBlock / 8 + 2];      // every position an escape: 32768 nibbles = 16 KiB (+ spill words)
    __shared__ unsigned long long held[kHeld];    // exceptions, appended with one global atomic at the end
    __shared__ int wave_sum[kRounds][4];
    __shared__ unsigned int n_held, base, b_off;
    const int tid = threadIdx.x, wv = tid >> 6;
    const int64_t block = blockIdx.x;
    if (tid == 0) n_held = 0;
    if (block == 0 && tid == 0) {
        head[1] = cap;
        head[3] = b_cap;
    }
    for (int i = tid; i < kBlock / 8; i += 256) nib[i] = 0;

    uint32_t w[kRounds][4];
    int prefix[kRounds];
    uint32_t esc[kRounds];
#pragma unroll
    for (int c = 0; c < kRounds; ++c) {
        const int64_t p0 = (block * kRounds + c) * kChunk + (int64_t)tid * kPerThread;
        w[c][0] = w[c][1] = w[c][2] = w[c][3] = 0x01010101u;  // past the end: value 1, no escape
        if (p0 + kPerThread <= n) {
            const uint4 q = *reinterpret_cast<const uint4 *>(in + p0);
            w[c][0] = q.x, w[c][1] = q.y, w[c][2] = q.z, w[c][3] = q.w;
        } else {
            for (int i = 0; i < kPerThread && p0 + i < n; ++i)
                w[c][i >> 2] = (w[c][i >> 2] & ~(0xFFu << (8 * (i & 3)))) | ((uint32_t)in[p0 + i] << (8 * (i & 3)));
        }
    }
#pragma unroll
    for (int c = 0; c < kRounds; ++c) {
        uint32_t a = 0, e = 0;
#pragma unroll
        for (int i = 0; i < kPerThread; ++i) {
            const uint32_t v = (w[c][i >> 2] >> (8 * (i & 3))) & 0xFFu;
            const bool direct = v - 1u < 3u;
            a |= (direct ? v : 0u) << (2 * i);
            e |= (direct ? 0u : 1u) << i;
        }
        esc[c] = e;
        const int64_t chunk = block * kRounds + c;
        if (chunk * kChunk < n) A[chunk * 256 + tid] = a;
        const int pc = __popc(e), inc = wave_inclusive_scan(pc);
        prefix[c] = inc - pc;
        if ((tid & 63) == 63) wave_sum[c][wv] = inc;
    }
    __syncthreads();

    int run = 0;  // nibbles of the rounds before this one
#pragma unroll
    for (int c = 0; c < kRounds; ++c) {
        const int s0 = wave_sum[c][0], s1 = wave_sum[c][1], s2 = wave_sum[c][2], s3 = wave_sum[c][3];
        int r = run + (wv > 0 ? s0 : 0) + (wv > 1 ? s1 : 0) + (wv > 2 ? s2 : 0) + prefix[c];
        run += s0 + s1 + s2 + s3;
        const int64_t p0 = (block * kRounds + c) * kChunk + (int64_t)tid * kPerThread;
        // this thread's nibbles of the round as one string (at most 16 nibbles), built without
        // branches, then or-ed into the block's nibble array at nibble r: three LDS atomics
        // whatever the count (a loop over the escapes costs one atomic each and diverges)
        uint64_t str = 0;
        uint32_t big = 0;
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < kPerThread; ++i) {
            const uint32_t v = (w[c][i >> 2] >> (8 * (i & 3))) & 0xFFu;
            const uint32_t is_esc = (esc[c] >> i) & 1u;
            const uint32_t e = v == 0 ? 0u : (v <= 17u ? v - 3u : 15u);
            str |= (uint64_t)(is_esc ? e : 0u) << (4 * cnt);
            big |= (v > 17u ? 1u : 0u) << i;
            cnt += (int)is_esc;
        }
        if (cnt) {
            const int sh = 4 * (r & 7);
            const uint64_t low = str << sh;
            const uint32_t top = sh ? (uint32_t)(str >> (64 - sh)) : 0u;
            uint32_t *cell = &nib[r >> 3];
            atomicOr(cell, (uint32_t)low);
            if ((uint32_t)(low >> 32)) atomicOr(cell + 1, (uint32_t)(low >> 32));
            if (top) atomicOr(cell + 2, top);
        }
        while (big) {  // values > 17: nibble 15 + an exception (one position in a thousand on config 3)
            const int i = __ffs(big) - 1;
            big &= big - 1;
            const uint32_t v = (w[c][i >> 2] >> (8 * (i & 3))) & 0xFFu;
            const unsigned long long entry = ((unsigned long long)(p0 + i) << 8) | v;
            const unsigned int slot = atomicAdd(&n_held, 1u);
            if (slot < (unsigned int)kHeld) {
                held[slot] = entry;
            } else {  // more than kHeld exceptions in one block: straight to the list, one atomic each
                const unsigned int at = atomicAdd(head, 1u);
                if (at < cap) exc[at] = entry;
            }
        }
    }
    // take (nibbles + 1) / 2 bytes, rounded to 4, from the B region
    const unsigned int bytes = (unsigned int)(((run + 1) / 2 + 3) & ~3);
    __syncthreads();
    const unsigned int mine = n_held < (unsigned int)kHeld ? n_held : (unsigned int)kHeld;
    if (tid == 0) {
        b_off = bytes ? atomicAdd(head + 2, bytes) : 0u;
        table[block] = make_uint2(b_off, (unsigned int)run);
        if (mine) base = atomicAdd(head, mine);
    }
    __syncthreads();
    if (bytes && (uint64_t)b_off + bytes <= b_cap) {  // else: head[2] > head[3] tells the caller
        uint32_t *dst = reinterpret_cast<uint32_t *>(B + b_off);
        for (unsigned int i = tid; i < bytes / 4; i += 256) dst[i] = nib[i];
    }
    for (unsigned int i = tid; i < mine; i += 256)
        if (base + i < cap) exc[base + i] = held[i];
}

__global__ __launch_bounds__(256) void dense_unpack_kernel(const uint32_t *A, const uint8_t *B, const uint2 *table,
                                                           unsigned int b_cap, int64_t n, uint8_t *out) {
    __shared__ uint32_t nib[kBlock / 8];
    __shared__ int wave_sum[kRounds][4];
    const int tid = threadIdx.x, wv = tid >> 6;
    const int64_t block = blockIdx.x;
    const uint2 where = table[block];
    const unsigned int bytes = (unsigned int)((((int)where.y + 1) / 2 + 3) & ~3);
    const bool have = (uint64_t)where.x + bytes <= b_cap && where.y <= (unsigned int)kBlock;
    uint32_t a[kRounds];
#pragma unroll
    for (int c = 0; c < kRounds; ++c) {
        const int64_t chunk = block * kRounds + c;
        a[c] = chunk * kChunk < n ? A[chunk * 256 + tid] : 0x55555555u;  // past the end: no escapes
    }
    const uint32_t *src = reinterpret_cast<const uint32_t *>(B + where.x);
    for (unsigned int i = tid; i < bytes / 4 && have; i += 256) nib[i] = src[i];
    int prefix[kRounds];
#pragma unroll
    for (int c = 0; c < kRounds; ++c) {
        // a position is an escape iff both bits of its code are 0
        const uint32_t any = a[c] | (a[c] >> 1);
        const int pc = kPerThread - __popc(any & 0x55555555u), inc = wave_inclusive_scan(pc);
        prefix[c] = inc - pc;
        if ((tid & 63) == 63) wave_sum[c][wv] = inc;
    }
    __syncthreads();
    int run = 0;
#pragma unroll
    for (int c = 0; c < kRounds; ++c) {
        const int s0 = wave_sum[c][0], s1 = wave_sum[c][1], s2 = wave_sum[c][2], s3 = wave_sum[c][3];
        int r = run + (wv > 0 ? s0 : 0) + (wv > 1 ? s1 : 0) + (wv > 2 ? s2 : 0) + prefix[c];
        run += s0 + s1 + s2 + s3;
        const int64_t p0 = (block * kRounds + c) * kChunk + (int64_t)tid * kPerThread;
        if (p0 >= n) continue;
        uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < kPerThread; ++i) {
            uint32_t v = (a[c] >> (2 * i)) & 3u;
            if (v == 0u) {  // nibble 15 is a placeholder: the exception pass writes the value
                const uint32_t e = have && r < kBlock ? (nib[r >> 3] >> (4 * (r & 7))) & 15u : 0u;
                v = e == 0u ? 0u : e + 3u;
                ++r;
            }
            w[i >> 2] |= v << (8 * (i & 3));
        }
        if (p0 + kPerThread <= n) {
            *reinterpret_cast<uint4 *>(out + p0) = make_uint4(w[0], w[1], w[2], w[3]);
        } else {
            for (int i = 0; i < kPerThread && p0 + i < n; ++i) out[p0 + i] = (uint8_t)(w[i >> 2] >> (8 * (i & 3)));
        }
    }
}

__global__ void dense_exceptions_kernel(const unsigned long long *exc, const unsigned int *head, int64_t n,
                                        uint8_t *out) {
    const unsigned int count = head[0] < head[1] ? head[0] : head[1];  // found, capacity
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
        const unsigned long long e = exc[i];
        const int64_t p = (int64_t)(e >> 8);
        if (p < n) out[p] = (uint8_t)(e & 0xFF);
    }
}

// ------------------------------------------------------------------------------------------
// Third coding, "runs": a conservation value is the minimum over the rows that cover a position, and neighbouring
// positions share almost all of their rows -- the value changes at one position in ten on config 3 (k = 31), at one in
// 350 at k = 101.
//   stream A   1 bit per position: 1 = the value differs from the one before (the first position of every
//              block of 32768 is marked always, so a block decodes on its own)            4096 B per block
//   stream B   one byte per marked position, in position order: the new value; a workgroup collects its bytes
//              in LDS and takes exactly what it needs (rounded to 4) from the B region with one atomic add;
//              the table holds every block's offset and byte count
// 1 + 8 * 0.099 = 1.8 bits per position on config 3 (dense coding: 3.3; nibbles: 4.25), no escapes, no exception
// list.  Same shape as the dense kernels: 256 threads own 32768 positions as 8 rounds of 4096, 16 positions per
// thread and round; ranks inside the block from the wave prefix and the (round, wave) sums.
// head: [0] B bytes taken, [1] B capacity
// ------------------------------------------------------------------------------------------
constexpr int kRunsStaged = 24576;  // bytes of a block's values staged in LDS (config 3: ~3300 per block; config 5 at k = 21, two bytes per
                                    // change: ~11 000 -- past round 4's 8192 every value was a dependent load from HBM, and that slice's
                                    // decode took 33 us against 22 for k = 31: profiles/r05_transport.txt); more: straight to / from HBM

// V = uint8_t (results of at most 255 genomes) or uint16_t (BASELINE config 5: 500 genomes -- round 4: a config-5 slice
// travelled as 67 MB of plain bytes, 0.9 ms of link against sweeps of 0.2-0.6 ms).  Same streams; B holds sizeof(V)
// bytes per marked position, a thread loads its sixteen positions as one or two 16-byte pieces.
template <typename V>
__global__ __launch_bounds__(256) void runs_pack_kernel(const V *in, int64_t n, uint16_t *A, uint8_t *B,
                                                        uint2 *table, unsigned int *head, unsigned int b_cap) {
    constexpr int VB = (int)sizeof(V), NW = 4 * VB;  // bytes per value; dwords per thread and round
    constexpr uint32_t VMASK = VB == 1 ? 0xFFu : 0xFFFFu;
    __shared__ __attribute__((aligned(16))) uint8_t vals[kRunsStaged];
    __shared__ int wave_sum[kRounds][4];
    __shared__ uint32_t wave_last[kRounds][4];  // the last value of every (round, wave): the next wave's "value before"
    __shared__ unsigned int b_off;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const int64_t block = blockIdx.x;
    if (block == 0 && tid == 0) head[1] = b_cap;

    uint32_t w[kRounds][NW];
    uint32_t before[kRounds];  // the value at the position before this thread's sixteen
    int prefix[kRounds];
    uint32_t marks[kRounds];
    auto value = [&](int c, int i) -> uint32_t {  // (static c and i everywhere below: registers are never indexed dynamically)
        return VB == 1 ? (w[c][i >> 2] >> (8 * (i & 3))) & 0xFFu : (w[c][i >> 1] >> (16 * (i & 1))) & 0xFFFFu;
    };
#pragma unroll
    for (int c = 0; c < kRounds; ++c) {
        const int64_t p0 = (block * kRounds + c) * kChunk + (int64_t)tid * kPerThread;
#pragma unroll
        for (int j = 0; j < NW; ++j) w[c][j] = 0;
        if (p0 + kPerThread <= n) {
#pragma unroll
            for (int h = 0; h < VB; ++h) {
                const uint4 q = *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint8_t *>(in + p0) + 16 * h);
                w[c][4 * h + 0] = q.x, w[c][4 * h + 1] = q.y, w[c][4 * h + 2] = q.z, w[c][4 * h + 3] = q.w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < kPerThread; ++i)
                if (p0 + i < n) {
                    if (VB == 1) w[c][i >> 2] |= (uint32_t)in[p0 + i] << (8 * (i & 3));
                    else w[c][i >> 1] |= (uint32_t)in[p0 + i] << (16 * (i & 1));
                }
        }
    }
    // the value before a thread's sixteen positions: its left neighbour's last one; lane 0 takes it from the wave
    // before (through LDS), the block's first thread needs none (a block starts with a marked position)
#pragma unroll
    for (int c = 0; c < kRounds; ++c) {
        const uint32_t last = value(c, kPerThread - 1);
        before[c] = (uint32_t)__shfl_up((int)last, 1, 64);
        if (lane == 63) wave_last[c][wv] = last;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < kRounds; ++c) {
        const int64_t p0 = (block * kRounds + c) * kChunk + (int64_t)tid * kPerThread;
        if (lane == 0) before[c] = wv > 0 ? wave_last[c][wv - 1] : (c > 0 ? wave_last[c - 1][3] : 0u);
        uint32_t m = 0, prev = before[c];
#pragma unroll
        for (int i = 0; i < kPerThread; ++i) {
            const uint32_t v = value(c, i);
            m |= (v != prev ? 1u : 0u) << i;
            prev = v;
        }
        if (c == 0 && tid == 0) m |= 1u;                                    // a block starts with a value
        const int64_t left = n - p0;                                         // positions of mine that exist
        m = left >= kPerThread ? m : (left <= 0 ? 0u : m & ((1u << left) - 1u));
        marks[c] = m;
        if (p0 < n) A[p0 / kPerThread] = (uint16_t)m;
        const int pc = __popc(m), inc = wave_inclusive_scan(pc);
        prefix[c] = inc - pc;
        if (lane == 63) wave_sum[c][wv] = inc;
    }
    __syncthreads();

    int total = 0;
#pragma unroll
    for (int c = 0; c < kRounds; ++c) total += wave_sum[c][0] + wave_sum[c][1] + wave_sum[c][2] + wave_sum[c][3];
    const unsigned int bytes = (unsigned int)((total * VB + 3) & ~3);
    const bool staged = total * VB <= kRunsStaged;  // (block-uniform)
    if (tid == 0) {
        b_off = bytes ? atomicAdd(head, bytes) : 0u;
        table[block] = make_uint2(b_off, (unsigned int)total);
    }
    __syncthreads();
    const bool fits = (uint64_t)b_off + bytes <= b_cap;  // else: head[0] > head[1] tells the caller
    auto scatter_values = [&](V *dst) {  // (LDS or HBM)
        int run = 0;  // values of the rounds before this one
#pragma unroll
        for (int c = 0; c < kRounds; ++c) {
            const int s0 = wave_sum[c][0], s1 = wave_sum[c][1], s2 = wave_sum[c][2], s3 = wave_sum[c][3];
            int r = run + (wv > 0 ? s0 : 0) + (wv > 1 ? s1 : 0) + (wv > 2 ? s2 : 0) + prefix[c];
            run += s0 + s1 + s2 + s3;
            const uint32_t m = marks[c];
            if (m) {  // (static value numbers: a loop over the set bits would index the registers dynamically --
#pragma unroll        //  a chain of selects per value, 8x the instructions of the whole kernel)
                for (int j = 0; j < 4; ++j) {
                    if ((m >> (4 * j)) & 0xFu) {  // (one position in ten is marked: most quartets have none)
#pragma unroll
                        for (int i = 4 * j; i < 4 * j + 4; ++i)
                            if ((m >> i) & 1u) dst[r++] = (V)(value(c, i) & VMASK);
                    }
                }
            }
        }
    };
    if (staged)
        scatter_values(reinterpret_cast<V *>(vals));
    else if (fits)
        scatter_values(reinterpret_cast<V *>(B + b_off));
    if (staged) {
        __syncthreads();
        if (bytes && fits) {
            uint32_t *out = reinterpret_cast<uint32_t *>(B + b_off);
            const uint32_t *src = reinterpret_cast<const uint32_t *>(vals);
            for (unsigned int i = tid; i < bytes / 4; i += 256) out[i] = src[i];
        }
    }
}

// one block of 32768 positions of one slice (shared by the kernel of one slice and the kernel of a whole gather step's slices)
template <typename V>
__device__ __forceinline__ void runs_unpack_block(const uint16_t *A, const uint8_t *B, const uint2 *table, unsigned int b_cap, int64_t n,
                                                  V *out, int64_t block) {
    constexpr int VB = (int)sizeof(V), NW = 4 * VB;
    __shared__ __attribute__((aligned(16))) uint8_t vals[kRunsStaged];
    __shared__ int wave_sum[kRounds][4];
    const int tid = threadIdx.x, wv = tid >> 6;
    const uint2 ent = table[block];
    const unsigned int bytes = (ent.y * (unsigned int)VB + 3u) & ~3u;
    if ((uint64_t)ent.x + bytes > b_cap) return;  // this block did not fit on the sender's side (stats say so)
    const bool staged = ent.y * (unsigned int)VB <= (unsigned int)kRunsStaged;
    if (staged) {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(B + ent.x);
        uint32_t *dst = reinterpret_cast<uint32_t *>(vals);
        for (unsigned int i = tid; i < bytes / 4; i += 256) dst[i] = src[i];
    }
    const V *val = staged ? reinterpret_cast<const V *>(vals) : reinterpret_cast<const V *>(B + ent.x);
    uint32_t marks[kRounds];
    int prefix[kRounds];
#pragma unroll
    for (int c = 0; c < kRounds; ++c) {
        const int64_t p0 = (block * kRounds + c) * kChunk + (int64_t)tid * kPerThread;
        marks[c] = p0 < n ? A[p0 / kPerThread] : 0u;
        const int pc = __popc(marks[c]), inc = wave_inclusive_scan(pc);
        prefix[c] = inc - pc;
        if ((tid & 63) == 63) wave_sum[c][wv] = inc;
    }
    __syncthreads();
    int run = 0;
#pragma unroll
    for (int c = 0; c < kRounds; ++c) {
        const int s0 = wave_sum[c][0], s1 = wave_sum[c][1], s2 = wave_sum[c][2], s3 = wave_sum[c][3];
        int r = run + (wv > 0 ? s0 : 0) + (wv > 1 ? s1 : 0) + (wv > 2 ? s2 : 0) + prefix[c];  // marks before mine
        run += s0 + s1 + s2 + s3;
        const int64_t p0 = (block * kRounds + c) * kChunk + (int64_t)tid * kPerThread;
        if (p0 >= n) continue;
        uint32_t cur = r > 0 ? val[r - 1] : 0u;  // (r == 0 only for the block's first thread, whose first bit is set)
        uint32_t q[NW];
#pragma unroll
        for (int j = 0; j < NW; ++j) q[j] = 0;
        const uint32_t m = marks[c];
#pragma unroll
        for (int i = 0; i < kPerThread; ++i) {
            if ((m >> i) & 1u) cur = val[r++];
            if (VB == 1) q[i >> 2] |= cur << (8 * (i & 3));
            else q[i >> 1] |= cur << (16 * (i & 1));
        }
        if (p0 + kPerThread <= n) {
#pragma unroll
            for (int h = 0; h < VB; ++h)
                *reinterpret_cast<uint4 *>(reinterpret_cast<uint8_t *>(out + p0) + 16 * h) =
                    make_uint4(q[4 * h], q[4 * h + 1], q[4 * h + 2], q[4 * h + 3]);
        } else {
#pragma unroll
            for (int i = 0; i < kPerThread; ++i)
                if (p0 + i < n) out[p0 + i] = VB == 1 ? (V)(q[i >> 2] >> (8 * (i & 3))) : (V)(q[i >> 1] >> (16 * (i & 1)));
        }
    }
}

template <typename V>
__global__ __launch_bounds__(256) void runs_unpack_kernel(const uint16_t *A, const uint8_t *B, const uint2 *table,
                                                          unsigned int b_cap, int64_t n, V *out) {
    runs_unpack_block<V>(A, B, table, b_cap, n, out, (int64_t)blockIdx.x);
}

// the slices rank 0 received in one gather step, decoded by ONE launch: a slice of 2^25 positions is 1024 workgroups -- four
// per CU, one short wave of them -- so seven launches one after the other cost seven launch-and-drain times (7 x 21-35 us on
// BASELINE config 5, more than the 0.18 ms sweep of k = 21 beside them: profiles/r04_scaling_model_c5.txt); together they
// fill the device once
constexpr int kManySlices = 16;
struct RunsMany {
    const char *wire[kManySlices];
    void *out[kManySlices];
};

template <typename V>
__global__ __launch_bounds__(256) void runs_unpack_many_kernel(const RunsMany m, int64_t blocks, size_t t_off, size_t a_off, size_t b_off,
                                                               unsigned int b_cap, int64_t n) {
    const int64_t slice = (int64_t)blockIdx.x / blocks, block = (int64_t)blockIdx.x % blocks;
    const char *w = m.wire[slice];
    runs_unpack_block<V>(reinterpret_cast<const uint16_t *>(w + a_off), reinterpret_cast<const uint8_t *>(w + b_off),
                         reinterpret_cast<const uint2 *>(w + t_off), b_cap, n, static_cast<V *>(m.out[slice]), block);
}

struct RunsLayout {
    int64_t blocks;
    size_t t_off, a_off, b_off, bytes;
};

RunsLayout runs_layout(int64_t n, uint32_t b_cap) {
    RunsLayout l;
    l.blocks = (n + kBlock - 1) / kBlock;
    l.t_off = 16;
    l.a_off = l.t_off + (size_t)l.blocks * 8;
    l.b_off = (l.a_off + (size_t)l.blocks * (kBlock / 8) + 15) & ~(size_t)15;
    l.bytes = (l.b_off + (size_t)b_cap + 15) & ~(size_t)15;
    return l;
}

int runs_check_args(const void *d_vec, const void *d_wire, int64_t n, uint32_t b_cap, int value_bytes = 1) {
    if (n < 0 || (n > 0 && (!d_vec || !d_wire))) return fail(MEMO_EINVAL, "bad transport arguments");
    // 32-bit offsets into the B region: the worst case -- every position marked, n values + 4 bytes per block of rounding
    if (n * value_bytes + 4 * ((n + kBlock - 1) / kBlock) >= ((int64_t)1 << 32) - 64)
        return fail(MEMO_EINVAL, "slice too long for the runs coding (at most ~2^32 bytes of values per slice)");
    if (b_cap % 4) return fail(MEMO_EINVAL, "the B region's capacity must be a multiple of 4");
    if (((uintptr_t)d_vec & 15) || ((uintptr_t)d_wire & 15))
        return fail(MEMO_EINVAL, "transport buffers must be 16-byte aligned");
    return MEMO_OK;
}

struct Layout {
    int64_t chunks, blocks;
    size_t t_off, a_off, b_off, x_off, bytes;
};

Layout layout(int64_t n, uint32_t b_cap, uint32_t cap) {
    Layout l;
    l.chunks = (n + kChunk - 1) / kChunk;
    l.blocks = (n + kBlock - 1) / kBlock;
    l.t_off = 16;
    l.a_off = l.t_off + (size_t)l.blocks * 8;
    l.b_off = l.a_off + (size_t)l.chunks * 1024;
    l.x_off = (l.b_off + (size_t)b_cap + 7) & ~(size_t)7;
    l.bytes = (l.x_off + (size_t)cap * 8 + 15) & ~(size_t)15;
    return l;
}

int check_args(const void *d_vec, const void *d_wire, int64_t n, uint32_t b_cap) {
    if (n < 0 || (n > 0 && (!d_vec || !d_wire))) return fail(MEMO_EINVAL, "bad transport arguments");
    // the B region is addressed with 32-bit offsets (table entries, the head[2] counter): the worst case --
    // every position an escape, n / 2 bytes + 4 per block of rounding -- has to fit
    if (n / 2 + 4 * ((n + kBlock - 1) / kBlock) >= ((int64_t)1 << 32) - 64)
        return fail(MEMO_EINVAL, "slice too long for the dense coding (at most ~2^33 positions per slice)");
    if (b_cap % 4) return fail(MEMO_EINVAL, "the B region's capacity must be a multiple of 4");
    if (((uintptr_t)d_vec & 15) || ((uintptr_t)d_wire & 15))
        return fail(MEMO_EINVAL, "transport buffers must be 16-byte aligned");
    return MEMO_OK;
}

}  // namespace

extern "C" {

// wire layout: [exceptions found u32, their capacity u32, B bytes taken u32, B capacity u32]
//              [table: (offset in B, nibbles) per 32768 positions][A: 1024 B per 4096 positions]
//              [B: b_capacity bytes][exceptions: cap * 8 B]
size_t memo_transport_dense_bytes(int64_t n, uint32_t b_capacity, uint32_t cap) {
    return n < 0 ? 0 : layout(n, b_capacity, cap).bytes;
}

int memo_transport_dense_pack_dev(const uint8_t *d_vec, int64_t n, uint32_t b_capacity, uint32_t cap,
                                  void *d_wire, int32_t device, void *stream) {
    int rc = check_args(d_vec, d_wire, n, b_capacity);
    if (rc) return rc;
    DeviceGuard guard(device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    char *w = static_cast<char *>(d_wire);
    if (!w) return MEMO_OK;
    const Layout l = layout(n, b_capacity, cap);
    HIP_TRY(hipMemsetAsync(w, 0, 16, st));
    if (l.blocks) {
        if (l.blocks >= ((int64_t)1 << 31)) return fail(MEMO_EINVAL, "slice too long for one launch");
        hipLaunchKernelGGL(dense_pack_kernel, dim3((unsigned)l.blocks), dim3(256), 0, st, d_vec, n,
                           reinterpret_cast<uint32_t *>(w + l.a_off), reinterpret_cast<uint8_t *>(w + l.b_off),
                           reinterpret_cast<uint2 *>(w + l.t_off),
                           reinterpret_cast<unsigned long long *>(w + l.x_off),
                           reinterpret_cast<unsigned int *>(w), cap, b_capacity);
        HIP_TRY(hipGetLastError());
    }
    return MEMO_OK;
}

int memo_transport_dense_unpack_dev(const void *d_wire, int64_t n, uint32_t b_capacity, uint32_t cap,
                                    uint8_t *d_vec, int32_t device, void *stream) {
    int rc = check_args(d_vec, d_wire, n, b_capacity);
    if (rc) return rc;
    DeviceGuard guard(device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const char *w = static_cast<const char *>(d_wire);
    const Layout l = layout(n, b_capacity, cap);
    if (l.blocks) {
        hipLaunchKernelGGL(dense_unpack_kernel, dim3((unsigned)l.blocks), dim3(256), 0, st,
                           reinterpret_cast<const uint32_t *>(w + l.a_off),
                           reinterpret_cast<const uint8_t *>(w + l.b_off),
                           reinterpret_cast<const uint2 *>(w + l.t_off), b_capacity, n, d_vec);
        hipLaunchKernelGGL(dense_exceptions_kernel, dim3(256), dim3(256), 0, st,
                           reinterpret_cast<const unsigned long long *>(w + l.x_off),
                           reinterpret_cast<const unsigned int *>(w), n, d_vec);
        HIP_TRY(hipGetLastError());
    }
    return MEMO_OK;
}

// what the sender found (host values; synchronises `stream`): exceptions against their capacity,
// B bytes taken against the B region's capacity.  The slice is complete iff neither exceeds.
int memo_transport_dense_stats(const void *d_wire, int32_t device, void *stream, uint32_t *found, uint32_t *cap,
                               uint32_t *b_taken, uint32_t *b_capacity) {
    if (!d_wire || !found || !cap || !b_taken || !b_capacity) return fail(MEMO_EINVAL, "NULL argument");
    DeviceGuard guard(device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    uint32_t head[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(head, d_wire, 16, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *found = head[0];
    *cap = head[1];
    *b_taken = head[2];
    *b_capacity = head[3];
    return MEMO_OK;
}

// ---- "runs" coding: wire layout [B bytes taken u32, B capacity u32, 0, 0][table: (offset in B, bytes) per 32768
//      positions][A: one bit per position, 4096 B per block][B: b_capacity bytes]
size_t memo_transport_runs_bytes(int64_t n, uint32_t b_capacity) {
    return n < 0 ? 0 : runs_layout(n, b_capacity).bytes;
}

}  // extern "C"

namespace {
template <typename V>
int runs_pack(const V *d_vec, int64_t n, uint32_t b_capacity, void *d_wire, int32_t device, void *stream) {
    int rc = runs_check_args(d_vec, d_wire, n, b_capacity, (int)sizeof(V));
    if (rc) return rc;
    DeviceGuard guard(device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    char *w = static_cast<char *>(d_wire);
    if (!w) return MEMO_OK;
    const RunsLayout l = runs_layout(n, b_capacity);
    HIP_TRY(hipMemsetAsync(w, 0, 16, st));
    if (l.blocks) {
        if (l.blocks >= ((int64_t)1 << 31)) return fail(MEMO_EINVAL, "slice too long for one launch");
        hipLaunchKernelGGL(runs_pack_kernel<V>, dim3((unsigned)l.blocks), dim3(256), 0, st, d_vec, n,
                           reinterpret_cast<uint16_t *>(w + l.a_off), reinterpret_cast<uint8_t *>(w + l.b_off),
                           reinterpret_cast<uint2 *>(w + l.t_off), reinterpret_cast<unsigned int *>(w), b_capacity);
        HIP_TRY(hipGetLastError());
    } else {
        HIP_TRY(hipMemcpyAsync(w + 4, &b_capacity, 4, hipMemcpyHostToDevice, st));
    }
    return MEMO_OK;
}

template <typename V>
int runs_unpack(const void *d_wire, int64_t n, uint32_t b_capacity, V *d_vec, int32_t device, void *stream) {
    int rc = runs_check_args(d_vec, d_wire, n, b_capacity, (int)sizeof(V));
    if (rc) return rc;
    DeviceGuard guard(device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const char *w = static_cast<const char *>(d_wire);
    const RunsLayout l = runs_layout(n, b_capacity);
    if (l.blocks) {
        hipLaunchKernelGGL(runs_unpack_kernel<V>, dim3((unsigned)l.blocks), dim3(256), 0, st,
                           reinterpret_cast<const uint16_t *>(w + l.a_off), reinterpret_cast<const uint8_t *>(w + l.b_off),
                           reinterpret_cast<const uint2 *>(w + l.t_off), b_capacity, n, d_vec);
        HIP_TRY(hipGetLastError());
    }
    return MEMO_OK;
}
}  // namespace

extern "C" {

int memo_transport_runs_pack_dev(const uint8_t *d_vec, int64_t n, uint32_t b_capacity, void *d_wire, int32_t device,
                                 void *stream) {
    return runs_pack<uint8_t>(d_vec, n, b_capacity, d_wire, device, stream);
}

int memo_transport_runs_unpack_dev(const void *d_wire, int64_t n, uint32_t b_capacity, uint8_t *d_vec, int32_t device,
                                   void *stream) {
    return runs_unpack<uint8_t>(d_wire, n, b_capacity, d_vec, device, stream);
}

// the same coding for uint16 results (more than 255 genomes): two bytes per marked position in the B region
int memo_transport_runs16_pack_dev(const uint16_t *d_vec, int64_t n, uint32_t b_capacity, void *d_wire, int32_t device,
                                   void *stream) {
    return runs_pack<uint16_t>(d_vec, n, b_capacity, d_wire, device, stream);
}

int memo_transport_runs16_unpack_dev(const void *d_wire, int64_t n, uint32_t b_capacity, uint16_t *d_vec, int32_t device,
                                     void *stream) {
    return runs_unpack<uint16_t>(d_wire, n, b_capacity, d_vec, device, stream);
}

// count slices of the same length and capacity (what one gather step brings to rank 0) -> count result vectors, one launch
// (value_bytes 1: uint8 results, 2: uint16).  d_wires / d_vecs: HOST arrays of device pointers.
int memo_transport_runs_unpack_many_dev(const void *const *d_wires, void *const *d_vecs, int32_t count, int64_t n, uint32_t b_capacity,
                                        int32_t value_bytes, int32_t device, void *stream) {
    if (count < 0 || (count > 0 && (!d_wires || !d_vecs)) || (value_bytes != 1 && value_bytes != 2))
        return fail(MEMO_EINVAL, "bad transport arguments");
    for (int32_t i = 0; i < count; ++i)
        if (int rc = runs_check_args(d_vecs[i], d_wires[i], n, b_capacity, value_bytes)) return rc;
    DeviceGuard guard(device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const RunsLayout l = runs_layout(n, b_capacity);
    if (!l.blocks) return MEMO_OK;
    for (int32_t at = 0; at < count; at += kManySlices) {
        const int32_t now = count - at < kManySlices ? count - at : kManySlices;
        if (l.blocks * now >= ((int64_t)1 << 31)) return fail(MEMO_EINVAL, "slices too long for one launch");
        RunsMany m;
        for (int32_t i = 0; i < kManySlices; ++i) {
            m.wire[i] = static_cast<const char *>(d_wires[at + (i < now ? i : 0)]);
            m.out[i] = d_vecs[at + (i < now ? i : 0)];
        }
        if (value_bytes == 1)
            hipLaunchKernelGGL(runs_unpack_many_kernel<uint8_t>, dim3((unsigned)(l.blocks * now)), dim3(256), 0, st, m, l.blocks, l.t_off,
                               l.a_off, l.b_off, b_capacity, n);
        else
            hipLaunchKernelGGL(runs_unpack_many_kernel<uint16_t>, dim3((unsigned)(l.blocks * now)), dim3(256), 0, st, m, l.blocks, l.t_off,
                               l.a_off, l.b_off, b_capacity, n);
        HIP_TRY(hipGetLastError());
    }
    return MEMO_OK;
}

// what the sender needed (host values; synchronises `stream`): B bytes taken against the B region's capacity.
// The slice is complete iff taken <= capacity.
int memo_transport_runs_stats(const void *d_wire, int32_t device, void *stream, uint32_t *b_taken, uint32_t *b_capacity) {
    if (!d_wire || !b_taken || !b_capacity) return fail(MEMO_EINVAL, "NULL argument");
    DeviceGuard guard(device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    uint32_t head[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(head, d_wire, 16, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *b_taken = head[0];
    *b_capacity = head[1];
    return MEMO_OK;
}

}  // extern "C"
