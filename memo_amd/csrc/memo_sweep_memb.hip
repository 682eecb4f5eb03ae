// memo_sweep_memb.hip -- membership sweeps: per-genome bit planes + register transpose ("runs",
// "planes") and the doubling scheme on bit cells; the k <= 1 fill, the side
// pass for rows with end < start, algorithm / tile-shape choice and the ABI entry point.
// Replaces /root/reference/src/memo_query.py:42-63 with rec = ones([L, N]) (:51) as bit rows.
#include "memo_sweep.h"

using namespace memo;

namespace {

// ------------------------------------------------------------------------------------------
// membership, doubling form.  Result word w of position x:  full_word(w) & ~absent[x][w].
//   The same two-blocks-per-row scatter and top-down fold as conservation, on cells of nw words
//   (or instead of min); nlev * W * nw words of LDS.  (The first version, one ds_or per covered
//   (position, genome) bit, took 4.35 ms on config 4 against 2.5 and is gone.)
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

template <typename Rows, int W, int U, int T>
__global__ __launch_bounds__(T) void sweep_membership_kernel(const SweepArgs A) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int tid = threadIdx.x;
    Tile t;
    const int nw = A.nwords;
    const int nlev = A.nlev;
    const int plane = W * nw;  // words per level
    if (!locate_tile<W>(A, t)) return;

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
        const int j = 31 - __builtin_clz((unsigned)(h - c));  // h - c >= 1
        uint32_t *lv = lds + j * plane + word;
        atomicOr(lv + c * nw, bit);  // rec[c:h, a] = False as two blocks of 2^j
        atomicOr(lv + (h - (1 << j)) * nw, bit);
    });
    __syncthreads();

    {
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
    // swap the high J-bit halves of m[k] with the low halves of m[k + J].  Whole bytes move with one v_perm_b32 per word (two
    // instructions a pair); below a byte a shift and a bit-field insert per word (four a pair) -- 256 vector instructions per
    // 32 x 32 block where the xor-swap (shift, xor-and, shift, xor, xor) takes 400: the planes kernels are bound by vector issue
    // on sparse indexes (profiles/r05_sq_counters_realistic.txt)
    constexpr uint32_t mask = J == 4 ? 0x0F0F0F0Fu : J == 2 ? 0x33333333u : 0x55555555u;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        if ((k & J) == 0) {
            const uint32_t a = m[k], b = m[k + J];
            if (J == 16) {
                m[k] = __builtin_amdgcn_perm(b, a, 0x05040100u);      // a.lo16 | b.lo16 << 16
                m[k + J] = __builtin_amdgcn_perm(b, a, 0x07060302u);  // a.hi16 | b.hi16 << 16
            } else if (J == 8) {
                m[k] = __builtin_amdgcn_perm(b, a, 0x06020400u);      // bytes a0 b0 a2 b2
                m[k + J] = __builtin_amdgcn_perm(b, a, 0x07030501u);  // bytes a1 b1 a3 b3
            } else {
                // (x & mask) | (y & ~mask) as ONE v_bfi_b32 (the compiler's rendering of the C expression took an extra and)
                const uint32_t bs = b << J, as = a >> J;
                asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(m[k]) : "s"(mask), "v"(a), "v"(bs));
                asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(m[k + J]) : "s"(mask), "v"(as), "v"(b));
            }
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
    if (!locate_tile<W>(A, t)) return;
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
            // (whole words: plain stores -- atomicOr(p, ~0u) compiles to ds_wrxchg_rtn_b32 and a wait for its result: planes_put)
            for (int w = w0 + 1; w < w1; ++w) __hip_atomic_store(row + w, 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
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
}


// One row of a planes kernel: the run of n = k - 1 - overlap bits that ends at bit d of plane row `col`; hand-written blocks,
// EXEC narrowed by v_cmpx and put back at the end as the block found it (every lane of the wave is active in the row loops today).
//   MW = 0 (k - 1 <= 31): the run fits two words.  12 vector instructions with the row's three fields and no scalar ones (the
//   compiler's branchy form: 17 and 8): v_bfm_b32 for both masks, a saturating subtract for the bits that spill into the second
//   word; a row that cannot write issues nothing (a zero or-ed into LDS costs what any atomic costs: the first version, without
//   the test, was 8 - 13 % slower through the dead rows of the loads that straddle a slice's end alone), a run inside one word
//   no second ds_or.
//   MW = 2, 3, 4, 8: longer runs, see the block.
//   base: the planes' LDS byte address in a VGPR; p4, s4: 4 * PITCH, 4 * SKEW; SK: SKEW != 0 (4, 8 or 16 result words).
// (-DMEMO_EXEC_MINUS_ONE: rounds 2-5's ending, `s_mov_b64 exec, -1`, for A/B -- profiles/r06_membership.txt)
#ifdef MEMO_EXEC_MINUS_ONE
#define MEMO_PLANES_KEEP ""
#define MEMO_PLANES_BACK "s_mov_b64 exec, -1"
#else
#define MEMO_PLANES_KEEP "s_mov_b64 s[38:39], exec\n\t"
#define MEMO_PLANES_BACK "s_mov_b64 exec, s[38:39]"
#endif
template <int MW, bool SK>
__device__ __forceinline__ void planes_put(int *status, uint32_t base, int km1, uint32_t p4, uint32_t s4, uint32_t len, uint32_t d,
                                           uint32_t col) {
    uint32_t t;
    MEMO_EXEC_ALL_ONES(status);  // (the row loops are wave-uniform: every lane of the wave is here)
    if constexpr (SK) {
        const uint32_t colhi = col >> 5;
        asm volatile("v_mad_u32_u24 %0, %1, %2, %3\n\t"
                     "v_mad_u32_u24 %0, %4, %5, %0"
                     : "=&v"(t) : "v"(col), "s"(p4), "v"(base), "v"(colhi), "s"(s4));
    } else {
        asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(t) : "v"(col), "s"(p4), "v"(base));
    }
    if constexpr (MW == 0) {
        uint32_t n, first, wq, addr, lo;
        // (EXEC is put back as the block found it -- s[38:39], declared clobbered -- not forced to -1: that would switch on lanes a future
        // caller had off)
        asm volatile(
            MEMO_PLANES_KEEP
            "v_sub_u32 %0, %6, %7\n\t"             // n = k - 1 - overlap
            "v_cmpx_lt_i32 vcc, 0, %0\n\t"         // the rest on the lanes whose row writes (the dead rows of a load that straddles
                                                   // an end of the slice too: two atomics that add nothing cost what any two cost)
            "v_sub_u32 %1, %8, %0\n\t"             // first bit of the run
            "v_lshrrev_b32 %2, 5, %1\n\t"          // its word
            "v_lshl_add_u32 %3, %2, 2, %5\n\t"     // its address
            "v_bfm_b32 %4, %0, %1\n\t"             // ((1 << n) - 1) << (first & 31)
            "v_lshl_add_u32 %2, %2, 5, 32\n\t"     // first bit of the next word
            "v_sub_u32_e64 %2, %8, %2 clamp\n\t"   // bits of the run in it
            "v_bfm_b32 %2, %2, 0\n\t"
            "ds_or_b32 %3, %4\n\t"
            "v_cmpx_ne_u32 vcc, 0, %2\n\t"         // (a run inside one word -- about half of them at k = 31 -- has no second ds_or)
            "ds_or_b32 %3, %2 offset:4\n\t"
            MEMO_PLANES_BACK
            : "=&v"(n), "=&v"(first), "=&v"(wq), "=&v"(addr), "=&v"(lo)
            : "v"(t), "s"(km1), "v"(len), "v"(d)
            : "memory", "vcc", "s38", "s39");
    } else {
        // Runs of up to MW + 1 words (MW = the most words a run can reach past its first: (k - 1 + 30) / 32, rounded up to 2, 3, 4
        // or 8): the first word (with the last one's mask where the run ends in it), the last word on the lanes whose run has
        // one, then whole words -- plain stores: all-ones or-ed into a word is all-ones stored, whatever another lane's ds_or does
        // before or after, and a store returns nothing (the compiler's rendering of atomicOr(p, ~0u) was ds_wrxchg_rtn_b32 with an
        // s_waitcnt lgkmcnt(0) per word and, the returned register being reused, one more wait in EVERY row, the short path's too:
        // unnoticed from round 2 to round 5) -- behind v_cmpx tests that only ever narrow EXEC: no loop, no branch.  One block.
        uint32_t n, first, last, w0, w1, more, head, tail, ones, a0;
#define MEMO_PLANES_WORD(I) "v_cmpx_lt_u32 vcc, " #I ", %5\n\tds_write_b32 %9, %8 offset:" #I "*4\n\t"
#define MEMO_PLANES_LONG(WORDS)                                                                                                     \
        asm volatile(                                                                                                               \
            MEMO_PLANES_KEEP                       /* EXEC as the block found it */                                                \
            "v_sub_u32 %0, %11, %12\n\t"           /* n = k - 1 - overlap */                                                       \
            "v_cmpx_lt_i32 vcc, 0, %0\n\t"         /* the lanes whose row writes */                                                \
            "v_sub_u32 %1, %13, %0\n\t"            /* first bit of the run */                                                      \
            "v_add_u32 %2, -1, %13\n\t"            /* last bit */                                                                  \
            "v_lshrrev_b32 %3, 5, %1\n\t"          /* their words */                                                               \
            "v_lshrrev_b32 %4, 5, %2\n\t"                                                                                          \
            "v_sub_u32 %5, %4, %3\n\t"             /* words past the first */                                                      \
            "v_lshlrev_b32_e64 %6, %1, -1\n\t"     /* -1 << (first & 31) */                                                        \
            "v_not_b32 %2, %2\n\t"                                                                                                 \
            "v_lshrrev_b32_e64 %7, %2, -1\n\t"     /* -1 >> (31 - (last & 31)) */                                                  \
            "v_cmp_eq_u32 vcc, 0, %5\n\t"                                                                                          \
            "v_cndmask_b32 %8, -1, %7, vcc\n\t"    /* the run ends in its first word: the last word's mask there; else all ones */ \
            "v_and_b32 %6, %6, %8\n\t"                                                                                             \
            "v_lshl_add_u32 %9, %3, 2, %10\n\t"    /* address of the first word */                                                 \
            "ds_or_b32 %9, %6\n\t"                                                                                                 \
            "v_cmpx_lt_u32 vcc, 0, %5\n\t"         /* runs of more than one word: the last one (%8 is all ones on these lanes) */  \
            "v_lshl_add_u32 %3, %4, 2, %10\n\t"                                                                                    \
            "ds_or_b32 %3, %7\n\t"                                                                                                 \
            WORDS                                  /* whole words: tests that only ever narrow EXEC */                             \
            MEMO_PLANES_BACK                                                                                                       \
            : "=&v"(n), "=&v"(first), "=&v"(last), "=&v"(w0), "=&v"(w1), "=&v"(more), "=&v"(head), "=&v"(tail), "=&v"(ones), "=&v"(a0) \
            : "v"(t), "s"(km1), "v"(len), "v"(d)                                                                                   \
            : "memory", "vcc", "s38", "s39")
        if constexpr (MW <= 2) MEMO_PLANES_LONG(MEMO_PLANES_WORD(1));
        else if constexpr (MW == 3) MEMO_PLANES_LONG(MEMO_PLANES_WORD(1) MEMO_PLANES_WORD(2));
        else if constexpr (MW == 4) MEMO_PLANES_LONG(MEMO_PLANES_WORD(1) MEMO_PLANES_WORD(2) MEMO_PLANES_WORD(3));
        else
            MEMO_PLANES_LONG(MEMO_PLANES_WORD(1) MEMO_PLANES_WORD(2) MEMO_PLANES_WORD(3) MEMO_PLANES_WORD(4) MEMO_PLANES_WORD(5)
                             MEMO_PLANES_WORD(6) MEMO_PLANES_WORD(7));
#undef MEMO_PLANES_LONG
#undef MEMO_PLANES_WORD
    }
}

// The second half of both planes kernels: the planes are complete (all scatters issued); transpose every 32 x 32 block
// in registers, stage the tile position-major in LDS over the planes, copy it out in whole 16-byte pieces.
template <int T>
__device__ __forceinline__ void planes_transpose_store(const SweepArgs &A, const Tile &t, uint32_t *lds) {
    const int tid = threadIdx.x;
    const int W = A.w, nw = A.nwords, PITCH = A.ls, SKEW = A.hl, HLW = A.nlev;
    __syncthreads();

    // transpose: lane (G, p), G fastest
    const int PW = W / 32, blocks = nw * PW;
    const int G = tid % nw, p = tid / nw;
    uint32_t m[32];
    if (tid < blocks) {
        const uint32_t *src = lds + (32 * G) * PITCH + G * SKEW + p + HLW;
#pragma unroll
        for (int i = 0; i < 32; ++i) m[i] = src[i * PITCH];
    }
    __syncthreads();  // the planes are dead: the staged result goes over them
    const int stride = 32 * nw + nw;  // staged words per position word (nw of padding)
    if (tid < blocks) {
        transpose32(m);
        const uint32_t full = full_word(A.ncols, G);
        uint32_t *dst = lds + p * stride + G;
#pragma unroll
        for (int j = 0; j < 32; ++j) dst[j * nw] = full & ~m[j];
    }
    __syncthreads();

    // copy: slots [x_lo, x_hi) are one contiguous run of words in the output
    uint32_t *out = static_cast<uint32_t *>(A.out);
    const int64_t ob = (t.a - A.qs) * nw;  // output word of tile slot 0, word 0
    const int64_t o_lo = ob + (int64_t)t.x_lo * nw, o_hi = ob + (int64_t)t.x_hi * nw;
    // staged word of output word q of the tile: q + (q / (32 nw)) * nw; the quotient by v_mul_hi (q < 2^15)
    auto staged = [&](int q) { return lds[q + (int)__umulhi((uint32_t)q, A.magic) * nw]; };
    for (int64_t g = (o_lo & ~(int64_t)3) + 4 * tid; g < o_hi; g += 4 * T) {
        const int q = (int)(g - ob);  // may be negative by up to 3 at the window's first piece
        if (g >= o_lo && g + 4 <= o_hi) {
            uint4 v;
            if ((nw & 3) == 0) {  // 4 | nw: a piece never straddles a position word, and is 16-byte aligned in LDS
                v = *reinterpret_cast<const uint4 *>(lds + q + (int)__umulhi((uint32_t)q, A.magic) * nw);
            } else {
                v.x = staged(q);
                v.y = staged(q + 1);
                v.z = staged(q + 2);
                v.w = staged(q + 3);
            }
            *reinterpret_cast<uint4 *>(out + g) = v;
        } else {
            for (int i = 0; i < 4; ++i)
                if (g + i >= o_lo && g + i < o_hi) out[g + i] = staged(q + i);
        }
    }
}

// ------------------------------------------------------------------------------------------
// membership, bit planes per genome, unclipped, with the result staged through LDS
// (packed rows whose annot is known to be inside the matrix, at most 16 result words).
//
//   * a genome's plane row covers ceil((k-1)/32) words left of the tile and a few right of it, so that
//     the run of any row of the slice (start - a <= W + k - 1 + bucket - 2, n = k - 1 - overlap bits
//     ending at start) fits without clipping.  The row itself is planes_put: MW = 0 (k <= 32) the run's two words, else
//     first word, last word, whole words -- hand-written blocks, one instantiation per reach of a run;
//   * lane (G, p) then reads the 32 plane rows of genome group G at position word p, transposes
//     the 32 x 32 bits in registers and writes the 32 result words position-major into LDS, over
//     the planes (all reads are behind a barrier by then);
//   * the workgroup copies the staged tile to the result in whole 16-byte pieces, 1 KiB per
//     wave-instruction (the transposing lanes themselves could only store 4-byte words 16*nw apart).
//   Plane rows are PITCH words apart (odd) with SKEW words per 32-genome group, the staged words
//   nw words of padding per position word: both chosen so that the lanes of a wave (nw groups x
//   64/nw position words) fall on 64 different banks.
// ------------------------------------------------------------------------------------------
template <typename Rows, int U, int T, int MW, bool SK>
__global__ __launch_bounds__(T) void sweep_membership_planes_kernel(const SweepArgs A) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int tid = threadIdx.x;
    const int W = A.w, nw = A.nwords, PITCH = A.ls, SKEW = A.hl;
    Tile t;
    if (!locate_tile_w(A, t, W)) return;
    uint4 V[U];
    uint2 N[U];
    Rows::template issue<T, U>(A, t, 0, V, N);
    {
        const int plane_pieces = (32 * nw * PITCH + nw * SKEW + 1 + 3) / 4;
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
        uint4 *pz = reinterpret_cast<uint4 *>(lds);
        for (int i = tid; i < plane_pieces; i += T) pz[i] = z;
        lds_barrier();
    }

    const int km1 = A.km1;
    const int HLW = A.nlev;  // words of halo left of the tile: ceil((k - 1) / 32)
    const uint32_t keym = pin_vgpr((int)Rows::tile_key(t.a - 32 * HLW));  // bit 32 * HLW of a plane row = tile slot 0
    const uint32_t base = pin_vgpr((int)(uint32_t)(size_t)(__attribute__((address_space(3))) uint32_t *)lds);
    const uint32_t p4 = 4u * (uint32_t)PITCH, s4 = 4u * (uint32_t)SKEW;
    auto scatter = [&](uint32_t w, uint32_t col) { planes_put<MW, SK>(A.status, base, km1, p4, s4, (uint32_t)Rows::len(w), Rows::rel_start(w, keym), col); };
    Rows::template consume<T, U>(A, t, 0, V, N, scatter);
    for (uint32_t b = 1, nb = Rows::template batches<T, U>(t); b < nb; ++b) {  // a dense tile: the rest
        Rows::template issue<T, U>(A, t, b, V, N);
        Rows::template consume<T, U>(A, t, b, V, N, scatter);
    }
    planes_transpose_store<T>(A, t, lds);
}

// The same on the dense rows (PackedRows3: five rows per 16 bytes, a fifth fewer bytes to read; k - 1 <= 63, at most 255
// genomes).  A row's field after the 16-bit subtract of the tile's key is (start - a + 32 HLW) << 6 | overlap, so the
// plane row's bit numbers have to stay below 2^10: the launcher sizes the tile for that (W + k - 1 + bucket + 32 HLW <=
// 1024).
template <int U, int T, int MW, bool SK>
__global__ __launch_bounds__(T) void sweep_membership_planes3_kernel(const SweepArgs A) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    using Rows = PackedRows3;
    const int tid = threadIdx.x;
    const int W = A.w, nw = A.nwords, PITCH = A.ls, SKEW = A.hl;
    Tile t;
    if (!locate_tile_w(A, t, W)) return;
    uint4 V[U];
    Rows::template issue<T, U>(A, t, 0, V);
    {
        const int plane_pieces = (32 * nw * PITCH + nw * SKEW + 1 + 3) / 4;
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
        uint4 *pz = reinterpret_cast<uint4 *>(lds);
        for (int i = tid; i < plane_pieces; i += T) pz[i] = z;
        lds_barrier();
    }
    const int km1 = A.km1;
    const int HLW = A.nlev;  // words of halo left of the tile: ceil((k - 1) / 32)
    const uint32_t key6 = pin_vgpr((int)((((uint32_t)(t.a - 32 * HLW)) & 1023u) << 6));  // bit 32 * HLW of a plane row = tile slot 0
    // r = (start - a + 32 HLW) << 6 | overlap
    const uint32_t base = pin_vgpr((int)(uint32_t)(size_t)(__attribute__((address_space(3))) uint32_t *)lds);
    const uint32_t p4 = 4u * (uint32_t)PITCH, s4 = 4u * (uint32_t)SKEW;
    auto put = [&](uint32_t r, uint32_t col) { planes_put<MW, SK>(A.status, base, km1, p4, s4, r & 63u, r >> 6, col); };
    auto g = [&](uint32_t b, uint32_t data) {  // 16-bit subtract on the low half; the result's high half is zero
        uint32_t r;
        asm("v_sub_u16 %0, %1, %2" : "=v"(r) : "v"(b), "v"(key6));
        put(r, data >> 24);
    };
    Rows::template consume<T, U>(A, t, 0, V, g);
    for (uint32_t b = 1, nb = Rows::template batches<T, U>(t); b < nb; ++b) {  // a dense tile: the rest
        Rows::template issue<T, U>(A, t, b, V);
        Rows::template consume<T, U>(A, t, b, V, g);
    }
    planes_transpose_store<T>(A, t, lds);
}

template <typename Rows, int T, int MW>
SweepKernel planes_kernel_m(bool skewed) {
    return skewed ? (SweepKernel)sweep_membership_planes_kernel<Rows, 6, T, MW, true> : (SweepKernel)sweep_membership_planes_kernel<Rows, 6, T, MW, false>;
}

template <typename Rows, int T>
SweepKernel planes_kernel_t(int mw, bool skewed) {
    return mw == 0   ? planes_kernel_m<Rows, T, 0>(skewed)
           : mw == 2 ? planes_kernel_m<Rows, T, 2>(skewed)
           : mw == 3 ? planes_kernel_m<Rows, T, 3>(skewed)
           : mw == 4 ? planes_kernel_m<Rows, T, 4>(skewed)
                     : planes_kernel_m<Rows, T, 8>(skewed);
}

template <typename Rows>
SweepKernel planes_kernel(int T, int mw, bool skewed) {
    return T == 64 ? planes_kernel_t<Rows, 64>(mw, skewed) : planes_kernel_t<Rows, 256>(mw, skewed);
}

__global__ void fill_membership_kernel(uint32_t *out, int64_t n, int nw, int ncols) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int left = ncols - 32 * (int)(i % nw);
        out[i] = left >= 32 ? 0xFFFFFFFFu : ((1u << left) - 1u);
    }
}

__global__ void long_rows_membership_kernel(const int64_t *ls, const int64_t *le, const int64_t *lo,
                                            int64_t qs, int64_t qe, int km1, int ncols, int nw,
                                            uint32_t *out, int *status, int64_t fqs, int64_t fqe) {
    const int64_t s = ls[blockIdx.x], e = le[blockIdx.x], o = lo[blockIdx.x];
    if (!(s > fqs && s < fqe + km1 + 1)) return;  // the reference's filter, on the whole window (memo_common.h: whole_set)
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


template <typename Rows>
SweepKernel memb_kernel(int w, int waves) {
#define MEMO_CASE(WW)                                                                              \
    case WW:                                                                                       \
        return waves == 4 ? (SweepKernel)sweep_membership_kernel<Rows, WW, Rows::kLoads, 256>      \
                          : (SweepKernel)sweep_membership_kernel<Rows, WW, Rows::kLoads, 64>;
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
}  // namespace

static int long_rows_membership(const memo_index *ix, int64_t qs, int64_t qe, int32_t k, int ncols, int nw,
                                uint32_t *d_out, hipStream_t st) {
    if (!ix->n_long || g_prepare_only) return MEMO_OK;
    if (int prc = refuse_plan_pointer(d_out)) return prc;
    hipLaunchKernelGGL(long_rows_membership_kernel, dim3((unsigned)ix->n_long), dim3(256), 0, st, ix->ls,
                       ix->le, ix->lo, qs, qe, k - 1, ncols, nw, d_out, ix->d_status,
                       ix->whole_set ? ix->whole_qs : qs, ix->whole_set ? ix->whole_qe : qe);
    HIP_TRY(hipGetLastError());
    return MEMO_OK;
}

extern "C" {

int memo_query_membership_dev(memo_index_t *ix, int64_t qs, int64_t qe, int32_t k,
                              int32_t num_docs, uint32_t *d_out, void *stream) {
    int rc = check_query_args(ix, qs, qe, k, num_docs, d_out);
    if (rc) return rc;
    if (qe <= qs) return MEMO_OK;
    DeviceGuard guard(ix->device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nw = (num_docs + 31) / 32;
    if (k <= 1 || ix->rows == 0) {
        if (g_prepare_only) return MEMO_OK;
        if ((rc = refuse_plan_pointer(d_out))) return rc;
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
    // the 4-byte words a sweep reads: the k-class view of them where one exists or is due (packed_rows_for, memo_view.hip)
    auto use_words = [&]() -> int {
        ix->last_rows_read = ix->rows;
        if (fmt != 4 && fmt != 12) return MEMO_OK;
        uint32_t *vpk = nullptr;
        int64_t *vboff = nullptr;
        uint64_t vrows = 0;
        const int vrc = packed_rows_for(ix, k - 1, qe - qs, true, st, &vpk, &vboff, &vrows);
        if (vrc) return vrc;
        A.pk = vpk;
        A.boff = vboff;
        ix->last_rows_read = vrows;
        return MEMO_OK;
    };
    ix->last_sweep = 7;  // a membership kernel on the 4- / 6-byte rows or the int64 columns (6: on the dense rows)
    const memo_tuning &tune0 = ix->tune;
    // The dense rows where an index holds no 4- / 6-byte rows and they can answer: the planes kernel on them (k - 1 <= 63,
    // at most 255 genomes, every annot inside the matrix, an index of >= 1 row per position like the other unclipped
    // kernels; a tile whose plane rows stay within the 10-bit start field).  What this buys is a 3.2-byte-per-row index
    // that answers both queries; it is not faster than the 4-byte rows here (config 4, back to back: 0.658 against 0.634 ms
    // at k = 31, 0.81 against 0.76 at k = 48 -- a membership sweep writes 1.6 GB, the 0.4 GB of rows it saves are paid
    // for by the longer decode and the shorter tiles; profiles/r02_dense_rows_ab.txt), so resident 4-byte rows are preferred.
    {
        const int bw = 1 << ix->bshift, hlw = (k - 1 + 31) / 32;
        const double span = (double)(ix->max_s - ix->min_s) + 1.0;
        int tw = (1024 - (k - 1) - bw - 32 * hlw) / bw * bw;
        tw = tw / 32 * 32;
        if (tune0.tile_w && tune0.tile_w < tw) tw = tune0.tile_w / bw * bw / 32 * 32;
        if (nw * (tw / 32) > 256) tw = 32 * (256 / nw);
        const bool dense_ok = ix->p3 && !ix->pk && !(tune0.force_wide && ix->has_wide) && k - 1 <= 63 &&
                              num_docs <= 255 && ix->max_annot < (uint64_t)num_docs && (double)ix->rows >= span &&
                              (tune0.memb_algo == 0 || tune0.memb_algo == 4) && tw >= bw && tw >= 32 && tw % bw == 0;
        if (dense_ok) {
            {   // the dense rows of this k's class (a view that leaves out the rows that cannot write at this k), or all of them
                uint32_t *vp3 = nullptr;
                int64_t *vboff = nullptr;
                uint64_t vrows = 0;
                if ((rc = dense_rows_for(ix, k - 1, qe - qs, st, &vp3, &vboff, &vrows))) return rc;
                A.p3 = vp3;
                A.boff = vboff;
                ix->last_rows_read = vrows;
            }
            const int pw = tw / 32;
            int skew = 0;
            for (int pow2 = 4; pow2 <= 64; pow2 <<= 1)
                if (nw == pow2) skew = (64 / nw + 32) & 63;
            A.w = tw;
            A.nlev = hlw;
            A.ls = (hlw + pw + ((k - 1 + bw - 3) >> 5) + 1) | 1;
            A.hl = skew;
            A.magic = (uint32_t)((((uint64_t)1 << 32) + 32 * nw - 1) / (32 * nw));
            A.word_base = 0;
            A.out_words = nw;
            const size_t planes = ((size_t)32 * nw * A.ls + (size_t)nw * skew + 8) * 4;
            const size_t staged = ((size_t)pw * (32 * nw + nw) + 4) * 4;
            // (k - 1 <= 63 here: a run reaches at most two words past its first)
            SweepKernel kern = k - 1 <= 31 ? (skew ? (SweepKernel)sweep_membership_planes3_kernel<PackedRows3::kLoads, 256, 0, true>
                                                   : (SweepKernel)sweep_membership_planes3_kernel<PackedRows3::kLoads, 256, 0, false>)
                                           : (skew ? (SweepKernel)sweep_membership_planes3_kernel<PackedRows3::kLoads, 256, 2, true>
                                                   : (SweepKernel)sweep_membership_planes3_kernel<PackedRows3::kLoads, 256, 2, false>);
            if ((rc = launch_tiles(kern, A, tw, 256, planes > staged ? planes : staged, st))) return rc;
            ix->last_sweep = 6;
            return long_rows_membership(ix, qs, qe, k, A.ncols, nw, d_out, st);
        }
    }
    if (fmt == 3) {  // only the dense rows are left, and they cannot answer this one
        if (!ix->has_wide)
            return fail(MEMO_EINVAL, "this membership query needs the 4-byte rows or the int64 columns, which this index dropped");
        fmt = 0;
    }
    // algorithm: 4 = planes (unclipped bit planes per genome, result staged), 3 = runs (clipped bit planes
    // + register transpose), 2 = doubling
    const memo_tuning &tune = ix->tune;
    const size_t per_pos_doubling = (size_t)A.nlev * nw * 4;
    int algo = tune.memb_algo;
    // A/B on config 4 (profiles/r01_membership_algorithms.txt): packed rows 0.87 ms runs vs 1.13 ms
    // doubling; int64 rows (HBM-bound either way) 2.52 ms doubling vs 2.64 ms runs
    if (!algo) algo = (fmt || per_pos_doubling * 256 > 40 * 1024) ? 3 : 2;
    // whatever was asked for: a tile of 256 positions has to fit in LDS, else runs (which can slice)
    if (algo == 2 && (per_pos_doubling * 256 > 128 * 1024 || nw > 64)) algo = 3;
    const bool checked = ix->max_annot >= (uint64_t)A.ncols;
    int w = tune.tile_w, waves = tune.waves == 1 || tune.waves == 4 ? tune.waves : 0;
    while (w & (w - 1)) w &= w - 1;  // (tile widths are powers of two here; the debug switch also takes other array sizes)
    if (w > 4096) w = 4096;
    A.word_base = 0;
    A.out_words = nw;
    // 4 = unclipped bit planes + staged result: packed rows with every annot inside the matrix, at most
    // 16 result words; otherwise whatever else was chosen
    if ((algo == 4 || (tune.memb_algo == 0 && algo == 3)) && fmt && !checked && nw <= 16) {
        const int bw = 1 << ix->bshift, T = waves == 1 ? 64 : 256;
        // ~16 KiB of planes per tile whatever the number of result words (128 of the 256 lanes transpose a block each): 1024
        // positions at four words (config 4), 2048 at two, 4096 at one, 512 at eight -- on the sequence-built index (50 genomes: two
        // words) 2048 against 1024 positions: 0.267 against 0.297 ms at k = 31; 4096 is slower again, as 2048 and 512 are at four
        // words; 250 genomes: 512 against 1024: 1.39 against 1.59 ms (profiles/r05_large_k.txt)
        int tw = w ? w : (4096 / nw < 256 ? 256 : 4096 / nw / 32 * 32);  // (whole 32-position words)
        if (fmt == 12 && tw > 2048) tw = 2048;       // (12-bit start field)
        if (tw > 32 * (T / nw)) tw = 32 * (T / nw);  // one 32 x 32 block per lane
        tw = tw / bw * bw;
        if (tw >= bw && tw >= 32) {
            const int pw = tw / 32;
            int skew = 0;
            for (int pow2 = 4; pow2 <= 64; pow2 <<= 1)
                if (nw == pow2) skew = (64 / nw + 32) & 63;  // G * (32 * PITCH + skew) = G * 64 / nw (mod 64), PITCH odd
            A.w = tw;
            const int hlw = (k - 1 + 31) / 32;
            A.nlev = hlw;  // words of halo left of the tile
            // last bit of a run: 32 hlw + tile + k - 1 + bw - 3; one more word for the short path's second ds_or
            A.ls = (hlw + pw + ((k - 1 + bw - 3) >> 5) + 1) | 1;
            A.hl = skew;
            A.magic = (uint32_t)((((uint64_t)1 << 32) + 32 * nw - 1) / (32 * nw));
            const size_t planes = ((size_t)32 * nw * A.ls + (size_t)nw * skew + 8) * 4;
            const size_t staged = ((size_t)pw * (32 * nw + nw) + 4) * 4;
            // the row block by the most words a run can reach past its first: 0 = the two-word block (k - 1 <= 31), else 2, 3, 4 or 8
            const int reach = (k - 1 + 30) / 32, mw = k - 1 <= 31 ? 0 : (reach <= 2 ? 2 : reach <= 3 ? 3 : reach <= 4 ? 4 : 8);
            SweepKernel kern = fmt == 4    ? planes_kernel<PackedRows<false, false>>(T, mw, skew != 0)
                               : fmt == 12 ? planes_kernel<PackedRows<false, false, true>>(T, mw, skew != 0)
                                           : planes_kernel<PackedRows<true, false>>(T, mw, skew != 0);
            if ((rc = use_words())) return rc;
            if ((rc = launch_tiles(kern, A, tw, T, planes > staged ? planes : staged, st))) return rc;
            return long_rows_membership(ix, qs, qe, k, A.ncols, nw, d_out, st);
        }
    }
    if (algo == 4) algo = 3;
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
        if (fmt == 12 && w > 2048) w = 2048;  // (12-bit start field)
        while (lds_bytes(w) > 160 * 1024 && w > 256) w >>= 1;
        const size_t lds = lds_bytes(w);
        SweepKernel kern = fmt == 4   ? (checked ? memb_runs_kernel<PackedRows<false, true>>(w, waves)
                                                 : memb_runs_kernel<PackedRows<false, false>>(w, waves))
                           : fmt == 12 ? (checked ? memb_runs_kernel<PackedRows<false, true, true>>(w, waves)
                                                  : memb_runs_kernel<PackedRows<false, false, true>>(w, waves))
                           : fmt == 6 ? (checked ? memb_runs_kernel<PackedRows<true, true>>(w, waves)
                                                 : memb_runs_kernel<PackedRows<true, false>>(w, waves))
                                      : memb_runs_kernel<WideRows>(w, waves);
        if (!kern) return fail(MEMO_EINVAL, "unsupported tile width %d", w);
        if ((rc = use_words())) return rc;  // (once per query: every slice of genome words reads the same rows)
        for (int base = 0; base < nw; base += slice) {
            A.word_base = base;
            A.nwords = nw - base < slice ? nw - base : slice;
            if ((rc = launch_tiles(kern, A, w, 64 * waves, lds, st))) return rc;
        }
        return long_rows_membership(ix, qs, qe, k, A.ncols, nw, d_out, st);
    }
    const size_t per_pos = per_pos_doubling;
    if (!waves) waves = 4;
    if (!w) {  // config 4 A/B: int64 rows 512 positions x 4 waves (40 KiB); packed rows 256 x 4 (20 KiB)
        const size_t budget = (waves == 4 ? (fmt ? 20u : 40u) : 20u) * 1024;
        w = 4096;
        while (per_pos * w > budget && w > 256) w >>= 1;
        while (w > 256 && (qe - qs) / w < 16384) w >>= 1;
    }
    while (per_pos * w > 160 * 1024 && w > 256) w >>= 1;
    if (fmt == 12 && w > 2048) w = 2048;  // (12-bit start field)
    SweepKernel kern = fmt == 4   ? (checked ? memb_kernel<PackedRows<false, true>>(w, waves)
                                             : memb_kernel<PackedRows<false, false>>(w, waves))
                       : fmt == 12 ? (checked ? memb_kernel<PackedRows<false, true, true>>(w, waves)
                                              : memb_kernel<PackedRows<false, false, true>>(w, waves))
                       : fmt == 6 ? (checked ? memb_kernel<PackedRows<true, true>>(w, waves)
                                             : memb_kernel<PackedRows<true, false>>(w, waves))
                                  : memb_kernel<WideRows>(w, waves);
    if (!kern) return fail(MEMO_EINVAL, "unsupported tile width %d", w);
    if ((rc = use_words())) return rc;
    if ((rc = launch_tiles(kern, A, w, 64 * waves, per_pos * w, st))) return rc;
    return long_rows_membership(ix, qs, qe, k, A.ncols, nw, d_out, st);
}

}  // extern "C"
