// memo_sweep_dense.h -- device pieces of the sweep that reads the dense rows with a lean, unrolled tile body
// (memo_sweep_cons3t.hip: the table-driven kernel): level clear / read in inline asm, the branch-free row block (plain and
// masked by row number), a group's five rows -- or six, for views whose groups carry their bucket --, the register fold + store.
#ifndef MEMO_SWEEP_DENSE_H
#define MEMO_SWEEP_DENSE_H

#include "memo_sweep.h"
#include "memo_sweep_fold.h"

namespace memo {
namespace dense {

constexpr int kLS = 1024;              // cells per level array (the dense rows' 10-bit start field)
constexpr int kStageGroups = 1024;     // groups (16 B) per stage
constexpr uint32_t kStageBytes = kStageGroups * 16;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));  // (a native vector: what a 128-bit asm operand has to be)

__device__ __forceinline__ void barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// what a tile needs, relative to the run's first group (all 32-bit)
struct Geo {
    uint32_t g0;     // first group of the slice (a multiple of 8), relative to the run's base group
    uint32_t ng;     // groups
    uint32_t first;  // rows [first, end) of the slice, counted from row 5 * g0
    uint32_t end;
};

// bucket-table entries of a tile (row numbers, absolute)
struct Slice {
    uint64_t r0, r1;
};

template <int N, int STEP, int I = 0>
__device__ __forceinline__ void clear_n(uint32_t at, const u32x4 &sv) {  // N stores of 16 B per lane, STEP bytes apart
    if constexpr (I < N) {
        asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(at), "v"(sv), "n"(I * STEP) : "memory");
        clear_n<N, STEP, I + 1>(at, sv);
    }
}

template <int NLEV, int T = 256>
__device__ __forceinline__ void clear_levels(uint32_t lds_base, uint32_t sent) {
    // every level starts at the sentinel column N (memo_query.py:53-54): NLEV x 4 KiB, 16 B per lane and store
    const u32x4 sv = {sent, sent, sent, sent};
    const uint32_t at = lds_base + 16u * threadIdx.x;
    static_assert(NLEV >= 1 && NLEV <= 6, "1 .. 6 level arrays (k - 1 <= 63)");
    if constexpr (T != 256) {  // (T threads cover 16 T bytes per store)
        clear_n<NLEV * 256 / T, 16 * T>(at, sv);
        return;
    }
#ifndef MEMO_CLEAR_B128  // (-DMEMO_CLEAR_B128: round 3's clear, one ds_write_b128 per lane and level, for A/B)
    // ds_write_addtid_b32 -- address = M0 + offset + 4 * lane, no address VGPR -- stores 256 B per wave-instruction in 2 cycles of
    // the LDS pipe where ds_write_b128 takes ~13 for 1 KiB (MI355X_MICROARCH.md, LDS): 4 NLEV instructions per wave instead of
    // NLEV, 8 cycles per KiB instead of 13.  Config 3 on the k-class view, sustained: k = 31 0.2040 -> 0.1990 ms, k = 21 0.1680 ->
    // 0.1635 (profiles/r04_view_levels.txt).  M0 is saved and restored inside the statement.
    {
        const uint32_t m0v = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds_base + (threadIdx.x >> 6) * 256u));
        uint32_t keep;
#define MEMO_AT(off) "ds_write_addtid_b32 %2 offset:" #off "\n\t"
#define MEMO_AT4(k) MEMO_AT(k) MEMO_AT(k + 1024) MEMO_AT(k + 2048) MEMO_AT(k + 3072)
#define MEMO_AT_OPEN "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 1\n\t"  /* (a DS instruction that reads M0 needs wait states behind the s_mov: without them the first store of three waves in four went astray -- tools/addtid_probe.hip) */
#define MEMO_AT_CLOSE "s_mov_b32 m0, %0" : "=&s"(keep) : "s"(m0v), "v"(sent) : "memory"
        if constexpr (NLEV == 1) asm volatile(MEMO_AT_OPEN MEMO_AT4(0) MEMO_AT_CLOSE);
        if constexpr (NLEV == 2) asm volatile(MEMO_AT_OPEN MEMO_AT4(0) MEMO_AT4(4096) MEMO_AT_CLOSE);
        if constexpr (NLEV == 3) asm volatile(MEMO_AT_OPEN MEMO_AT4(0) MEMO_AT4(4096) MEMO_AT4(8192) MEMO_AT_CLOSE);
        if constexpr (NLEV == 4) asm volatile(MEMO_AT_OPEN MEMO_AT4(0) MEMO_AT4(4096) MEMO_AT4(8192) MEMO_AT4(12288) MEMO_AT_CLOSE);
        if constexpr (NLEV == 5)
            asm volatile(MEMO_AT_OPEN MEMO_AT4(0) MEMO_AT4(4096) MEMO_AT4(8192) MEMO_AT4(12288) MEMO_AT4(16384) MEMO_AT_CLOSE);
        if constexpr (NLEV == 6)
            asm volatile(MEMO_AT_OPEN MEMO_AT4(0) MEMO_AT4(4096) MEMO_AT4(8192) MEMO_AT4(12288) MEMO_AT4(16384) MEMO_AT4(20480) MEMO_AT_CLOSE);
#undef MEMO_AT_CLOSE
#undef MEMO_AT_OPEN
#undef MEMO_AT4
#undef MEMO_AT
        return;
    }
#endif
#define MEMO_CLR(off) "ds_write_b128 %0, %1 offset:" #off "\n\t"
    if constexpr (NLEV == 1) asm volatile(MEMO_CLR(0) :: "v"(at), "v"(sv) : "memory");
    if constexpr (NLEV == 2) asm volatile(MEMO_CLR(0) MEMO_CLR(4096) :: "v"(at), "v"(sv) : "memory");
    if constexpr (NLEV == 3) asm volatile(MEMO_CLR(0) MEMO_CLR(4096) MEMO_CLR(8192) :: "v"(at), "v"(sv) : "memory");
    if constexpr (NLEV == 4) asm volatile(MEMO_CLR(0) MEMO_CLR(4096) MEMO_CLR(8192) MEMO_CLR(12288) :: "v"(at), "v"(sv) : "memory");
    if constexpr (NLEV == 5)
        asm volatile(MEMO_CLR(0) MEMO_CLR(4096) MEMO_CLR(8192) MEMO_CLR(12288) MEMO_CLR(16384) :: "v"(at), "v"(sv) : "memory");
    if constexpr (NLEV == 6)
        asm volatile(MEMO_CLR(0) MEMO_CLR(4096) MEMO_CLR(8192) MEMO_CLR(12288) MEMO_CLR(16384) MEMO_CLR(20480) :: "v"(at), "v"(sv) : "memory");
#undef MEMO_CLR
}

// one ds_read_b128 per level at the same cell (levels are 4 KiB apart), waited for in the same statement
template <int NLEV>
__device__ __forceinline__ void read_levels(uint32_t addr, u32x4 (&L)[6]) {
#define MEMO_RD(i, off) "ds_read_b128 %" #i ", %" MEMO_ADDR " offset:" #off "\n\t"
#define MEMO_ADDR "1"
    if constexpr (NLEV == 1) asm volatile(MEMO_RD(0, 0) "s_waitcnt lgkmcnt(0)" : "=&v"(L[0]) : "v"(addr) : "memory");
#undef MEMO_ADDR
#define MEMO_ADDR "2"
    if constexpr (NLEV == 2)
        asm volatile(MEMO_RD(0, 0) MEMO_RD(1, 4096) "s_waitcnt lgkmcnt(0)" : "=&v"(L[0]), "=&v"(L[1]) : "v"(addr) : "memory");
#undef MEMO_ADDR
#define MEMO_ADDR "3"
    if constexpr (NLEV == 3)
        asm volatile(MEMO_RD(0, 0) MEMO_RD(1, 4096) MEMO_RD(2, 8192) "s_waitcnt lgkmcnt(0)"
                     : "=&v"(L[0]), "=&v"(L[1]), "=&v"(L[2]) : "v"(addr) : "memory");
#undef MEMO_ADDR
#define MEMO_ADDR "4"
    if constexpr (NLEV == 4)
        asm volatile(MEMO_RD(0, 0) MEMO_RD(1, 4096) MEMO_RD(2, 8192) MEMO_RD(3, 12288) "s_waitcnt lgkmcnt(0)"
                     : "=&v"(L[0]), "=&v"(L[1]), "=&v"(L[2]), "=&v"(L[3]) : "v"(addr) : "memory");
#undef MEMO_ADDR
#define MEMO_ADDR "5"
    if constexpr (NLEV == 5)
        asm volatile(MEMO_RD(0, 0) MEMO_RD(1, 4096) MEMO_RD(2, 8192) MEMO_RD(3, 12288) MEMO_RD(4, 16384) "s_waitcnt lgkmcnt(0)"
                     : "=&v"(L[0]), "=&v"(L[1]), "=&v"(L[2]), "=&v"(L[3]), "=&v"(L[4]) : "v"(addr) : "memory");
#undef MEMO_ADDR
#define MEMO_ADDR "6"
    if constexpr (NLEV == 6)
        asm volatile(MEMO_RD(0, 0) MEMO_RD(1, 4096) MEMO_RD(2, 8192) MEMO_RD(3, 12288) MEMO_RD(4, 16384) MEMO_RD(5, 20480)
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(L[0]), "=&v"(L[1]), "=&v"(L[2]), "=&v"(L[3]), "=&v"(L[4]), "=&v"(L[5]) : "v"(addr) : "memory");
#undef MEMO_ADDR
#undef MEMO_RD
}

// The row block of sweep_conservation_halo3_kernel: 16-bit subtract (start - a, length untouched below it), length,
// "this row writes" into EXEC, level and both cells, ds_min x 2, EXEC restored.  MASKED: the row's number (tmp + I,
// counted from the slice's first row) is tested against the slice's row count first.
struct RowConst {
    uint32_t a10s, bias4, top_bit;  // VGPRs
    int km1;                        // SGPRs
    uint32_t ls4;
    int *status;                    // (the index's sticky flags: -DMEMO_EXEC_CHECK builds report through them)
};

// One group's five rows as ONE statement (PackedRows3: rows 0 .. 3 as loaded, row 4 from the spare bytes).  Five
// separate statements cost an s_nop each: the compiler cannot see into an asm block and pads for a hazard the next
// block might have (none here: an SALU write of EXEC needs no wait states before a VALU that only uses it as its mask).
// operands: %0-%3 temporaries; %4-%7 the group's dwords; %8 / %9 row 4's field and order word; %10 a10s, %11 km1 (SGPR),
// %12 ls4 (SGPR), %13 bias4, %14 top_bit; masked form: %15 tmp, %16 span (SGPR)
// TEST / DONE: "v_cmpx_lt_i32 vcc, 0, %0" + the s_mov that restores EXEC -- or nothing where every row is known to write (AW: a
// k-class view whose cap is this query's k - 1 holds exactly the rows with overlap < k - 1: one vector and one scalar instruction
// per row less)
#define MEMO_ROW3_TEST "v_cmpx_lt_i32 vcc, 0, %0\n\t"
#define MEMO_ROW3_DONE "s_mov_b64 exec, -1\n\t"
#define MEMO_ROW3_AT_(B, D, TEST, DONE)                  \
    "v_sub_u16 %3, " B ", %10\n\t"                       \
    "v_and_b32 %0, 63, %3\n\t"                           \
    "v_sub_u32 %0, %11, %0\n\t"                          \
    TEST                                                 \
    "v_ffbh_u32 %1, %0\n\t"                              \
    "v_bfe_u32 %3, %3, 6, 10\n\t"                        \
    "v_mad_u32_u24 %2, %1, %12, %13\n\t"                 \
    "v_lshl_add_u32 %2, %3, 2, %2\n\t"                   \
    "v_mad_i32_i24 %3, %0, -4, %2\n\t"                   \
    "v_ashrrev_i32 %1, %1, %14\n\t"                      \
    "v_lshl_add_u32 %2, %1, 2, %2\n\t"                   \
    "ds_min_u32 %3, " D "\n\t"                           \
    "ds_min_u32 %2, " D "\n\t"                           \
    DONE
#define MEMO_ROW3_AT(B, D) MEMO_ROW3_AT_(B, D, MEMO_ROW3_TEST, MEMO_ROW3_DONE)
#define MEMO_ROW3_AW(B, D) MEMO_ROW3_AT_(B, D, "", "")              /* every row writes, every lane holds a row */
#define MEMO_ROW3_AWM(B, D) MEMO_ROW3_AT_(B, D, "", MEMO_ROW3_DONE)  /* ... behind a mask by row number */
#define MEMO_ROW3_MASK(I) "v_add_u32 %0, %15, " #I "\n\tv_cmpx_gt_u32 vcc, %16, %0\n\t"

// Indexes of 256 .. 511 genomes (A9): the ninth bit of row i's annot sits at bit 16 + i of the group's last dword
// (pack3_rows_kernel), and a level cell holds the order in its top NINE bits: the data of a row's two ds_min is
// (ninth bit : dword) >> 1 -- two more vector instructions per row.  operands as above, shifted by one: %4 the data.
#define MEMO_ROW9_AT_(B, D, SH, TEST, DONE)              \
    "v_lshrrev_b32 %4, " SH ", %8\n\t"                   \
    "v_alignbit_b32 %4, %4, " D ", 1\n\t"                \
    "v_sub_u16 %3, " B ", %11\n\t"                       \
    "v_and_b32 %0, 63, %3\n\t"                           \
    "v_sub_u32 %0, %12, %0\n\t"                          \
    TEST                                                 \
    "v_ffbh_u32 %1, %0\n\t"                              \
    "v_bfe_u32 %3, %3, 6, 10\n\t"                        \
    "v_mad_u32_u24 %2, %1, %13, %14\n\t"                 \
    "v_lshl_add_u32 %2, %3, 2, %2\n\t"                   \
    "v_mad_i32_i24 %3, %0, -4, %2\n\t"                   \
    "v_ashrrev_i32 %1, %1, %15\n\t"                      \
    "v_lshl_add_u32 %2, %1, 2, %2\n\t"                   \
    "ds_min_u32 %3, %4\n\t"                              \
    "ds_min_u32 %2, %4\n\t"                              \
    DONE
#define MEMO_ROW9_AT(B, D, SH) MEMO_ROW9_AT_(B, D, SH, MEMO_ROW3_TEST, MEMO_ROW3_DONE)
#define MEMO_ROW9_AW(B, D, SH) MEMO_ROW9_AT_(B, D, SH, "", "")
#define MEMO_ROW9_AWM(B, D, SH) MEMO_ROW9_AT_(B, D, SH, "", MEMO_ROW3_DONE)
#define MEMO_ROW9_MASK(I) "v_add_u32 %0, %16, " #I "\n\tv_cmpx_gt_u32 vcc, %17, %0\n\t"

#define MEMO_G3_OUT : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
#define MEMO_G3_IN "v"(V.x), "v"(V.y), "v"(V.z), "v"(V.w), "v"(b4), "v"(d4), "v"(C.a10s), "s"(C.km1), "s"(C.ls4), "v"(C.bias4), "v"(C.top_bit)
#define MEMO_G9_OUT : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4)
// AW: every row of this source writes at this k (see MEMO_ROW3_TEST)
template <bool MASKED, bool A9 = false, bool AW = false>
__device__ __forceinline__ void group_rows(const uint4 &V, const RowConst &C, uint32_t tmp, uint32_t span) {
    const uint32_t b4 = __builtin_amdgcn_perm(V.y, V.x, 0x0c0c0602u), d4 = V.z << 8;
    uint32_t r0, r1, r2, r3;
    MEMO_EXEC_ALL_ONES(C.status);
    if constexpr (A9) {
        uint32_t r4;
        if constexpr (MASKED && AW) {
            asm volatile(MEMO_ROW9_MASK(0) MEMO_ROW9_AWM("%5", "%5", "16") MEMO_ROW9_MASK(1) MEMO_ROW9_AWM("%6", "%6", "17")
                         MEMO_ROW9_MASK(2) MEMO_ROW9_AWM("%7", "%7", "18") MEMO_ROW9_MASK(3) MEMO_ROW9_AWM("%8", "%8", "19")
                         MEMO_ROW9_MASK(4) MEMO_ROW9_AWM("%9", "%10", "20")
                         MEMO_G9_OUT : MEMO_G3_IN, "v"(tmp), "s"(span) : "memory", "vcc");
        } else if constexpr (MASKED) {
            asm volatile(MEMO_ROW9_MASK(0) MEMO_ROW9_AT("%5", "%5", "16") MEMO_ROW9_MASK(1) MEMO_ROW9_AT("%6", "%6", "17")
                         MEMO_ROW9_MASK(2) MEMO_ROW9_AT("%7", "%7", "18") MEMO_ROW9_MASK(3) MEMO_ROW9_AT("%8", "%8", "19")
                         MEMO_ROW9_MASK(4) MEMO_ROW9_AT("%9", "%10", "20")
                         MEMO_G9_OUT : MEMO_G3_IN, "v"(tmp), "s"(span) : "memory", "vcc");
        } else if constexpr (AW) {
            asm volatile(MEMO_ROW9_AW("%5", "%5", "16") MEMO_ROW9_AW("%6", "%6", "17") MEMO_ROW9_AW("%7", "%7", "18")
                         MEMO_ROW9_AW("%8", "%8", "19") MEMO_ROW9_AW("%9", "%10", "20")
                         MEMO_G9_OUT : MEMO_G3_IN : "memory");
        } else {
            asm volatile(MEMO_ROW9_AT("%5", "%5", "16") MEMO_ROW9_AT("%6", "%6", "17") MEMO_ROW9_AT("%7", "%7", "18")
                         MEMO_ROW9_AT("%8", "%8", "19") MEMO_ROW9_AT("%9", "%10", "20")
                         MEMO_G9_OUT : MEMO_G3_IN : "memory", "vcc");
        }
    } else if constexpr (MASKED && AW) {
        asm volatile(MEMO_ROW3_MASK(0) MEMO_ROW3_AWM("%4", "%4") MEMO_ROW3_MASK(1) MEMO_ROW3_AWM("%5", "%5")
                     MEMO_ROW3_MASK(2) MEMO_ROW3_AWM("%6", "%6") MEMO_ROW3_MASK(3) MEMO_ROW3_AWM("%7", "%7")
                     MEMO_ROW3_MASK(4) MEMO_ROW3_AWM("%8", "%9")
                     MEMO_G3_OUT : MEMO_G3_IN, "v"(tmp), "s"(span) : "memory", "vcc");
    } else if constexpr (MASKED) {
        asm volatile(MEMO_ROW3_MASK(0) MEMO_ROW3_AT("%4", "%4") MEMO_ROW3_MASK(1) MEMO_ROW3_AT("%5", "%5")
                     MEMO_ROW3_MASK(2) MEMO_ROW3_AT("%6", "%6") MEMO_ROW3_MASK(3) MEMO_ROW3_AT("%7", "%7")
                     MEMO_ROW3_MASK(4) MEMO_ROW3_AT("%8", "%9")
                     MEMO_G3_OUT : MEMO_G3_IN, "v"(tmp), "s"(span) : "memory", "vcc");
    } else if constexpr (AW) {
        asm volatile(MEMO_ROW3_AW("%4", "%4") MEMO_ROW3_AW("%5", "%5") MEMO_ROW3_AW("%6", "%6") MEMO_ROW3_AW("%7", "%7")
                     MEMO_ROW3_AW("%8", "%9")
                     MEMO_G3_OUT : MEMO_G3_IN : "memory");
    } else {
        asm volatile(MEMO_ROW3_AT("%4", "%4") MEMO_ROW3_AT("%5", "%5") MEMO_ROW3_AT("%6", "%6") MEMO_ROW3_AT("%7", "%7")
                     MEMO_ROW3_AT("%8", "%9")
                     MEMO_G3_OUT : MEMO_G3_IN : "memory", "vcc");
    }
}

// ---- groups of SIX rows that carry their bucket (memo_view.hip: view_build_kernel<6>, pack_six) ---------------------------------
// A row's ten bits (start mod 32 | overlap << 5) are the low bits of LO, its ds_min operand D has its annot on top; BIAS = the level
// arrays' bias + 4 * the cell of the group's bucket.  operands: %0-%3 temporaries; %4-%9 LO of rows 0 .. 5; %10-%15 their D;
// %16 BIAS; %17 km1 (SGPR), %18 ls4 (SGPR), %19 top_bit; masked form: %20 the lane's group number, %21 groups left (SGPR)
#define MEMO_ROW6_AT_(LO, D, TEST, DONE)                 \
    "v_bfe_u32 %0, " LO ", 5, 5\n\t"                     \
    "v_sub_u32 %0, %17, %0\n\t"                          \
    TEST                                                 \
    "v_ffbh_u32 %1, %0\n\t"                              \
    "v_and_b32 %3, 31, " LO "\n\t"                       \
    "v_mad_u32_u24 %2, %1, %18, %16\n\t"                 \
    "v_lshl_add_u32 %2, %3, 2, %2\n\t"                   \
    "v_mad_i32_i24 %3, %0, -4, %2\n\t"                   \
    "v_ashrrev_i32 %1, %1, %19\n\t"                      \
    "v_lshl_add_u32 %2, %1, 2, %2\n\t"                   \
    "ds_min_u32 %3, " D "\n\t"                           \
    "ds_min_u32 %2, " D "\n\t"                           \
    DONE
#define MEMO_ROW6_MASK "v_cmpx_gt_u32 vcc, %21, %20\n\t"
#define MEMO_SIX_ROWS(PRE, TEST, DONE)                                                                                          \
    PRE MEMO_ROW6_AT_("%4", "%10", TEST, DONE) PRE MEMO_ROW6_AT_("%5", "%11", TEST, DONE) PRE MEMO_ROW6_AT_("%6", "%12", TEST, DONE) \
    PRE MEMO_ROW6_AT_("%7", "%13", TEST, DONE) PRE MEMO_ROW6_AT_("%8", "%14", TEST, DONE) PRE MEMO_ROW6_AT_("%9", "%15", TEST, DONE)

struct SixConst {
    uint32_t nega, bias4, top_bit;  // VGPRs: -(tile start mod 1024), the level arrays' bias, 0x80000000
    int km1;                        // SGPRs
    uint32_t ls4;
    int *status;
};

// lg: the lane's group number inside the piece's count, left: groups of the slice left at this piece (MASKED: lanes past it do nothing)
template <bool MASKED, bool AW>
__device__ __forceinline__ void group_rows6(const uint4 &V, const SixConst &C, uint32_t lg, uint32_t left) {
    const uint32_t lo4 = V.x >> 10, d4 = V.y << 14, lo5 = V.w >> 10, d5 = V.z << 14;
    // the cell of the bucket's first position in the tile's arrays: (bucket mod 32) * 32 - tile start, mod 1024
    const uint32_t cell = (__builtin_amdgcn_ubfe(V.z, 18, 5) * 32u + C.nega) & 1023u;
    const uint32_t bias = C.bias4 + 4u * cell;
    uint32_t r0, r1, r2, r3;
    MEMO_EXEC_ALL_ONES(C.status);
#define MEMO_SIX_IN "v"(V.x), "v"(V.y), "v"(V.z), "v"(V.w), "v"(lo4), "v"(lo5), "v"(V.x), "v"(V.y), "v"(V.z), "v"(V.w), "v"(d4), "v"(d5), \
                    "v"(bias), "s"(C.km1), "s"(C.ls4), "v"(C.top_bit)
    if constexpr (MASKED && AW) {
        asm volatile(MEMO_SIX_ROWS(MEMO_ROW6_MASK, "", MEMO_ROW3_DONE) MEMO_G3_OUT : MEMO_SIX_IN, "v"(lg), "s"(left) : "memory", "vcc");
    } else if constexpr (MASKED) {
        asm volatile(MEMO_SIX_ROWS(MEMO_ROW6_MASK, MEMO_ROW3_TEST, MEMO_ROW3_DONE) MEMO_G3_OUT : MEMO_SIX_IN, "v"(lg), "s"(left) : "memory", "vcc");
    } else if constexpr (AW) {
        asm volatile(MEMO_SIX_ROWS("", "", "") MEMO_G3_OUT : MEMO_SIX_IN : "memory");
    } else {
        asm volatile(MEMO_SIX_ROWS("", MEMO_ROW3_TEST, MEMO_ROW3_DONE) MEMO_G3_OUT : MEMO_SIX_IN : "memory", "vcc");
    }
#undef MEMO_SIX_IN
}

// the NL groups of a lane (group J * T + tid of the batch), until the slice's groups end: a piece wholly inside runs unmasked
template <int T, int NL, bool AW, int J = 0>
__device__ __forceinline__ void six_pieces(const uint4 (&V)[NL], int lane, int wave, uint32_t gleft, const SixConst &C) {
    if constexpr (J < NL) {
        const uint32_t pg = (uint32_t)(J * T + wave * 64);
        if (pg >= gleft) return;
        if (pg + 64u <= gleft) group_rows6<false, AW>(V[J], C, 0, 0);
        else group_rows6<true, AW>(V[J], C, pg + (uint32_t)lane, gleft);
        six_pieces<T, NL, AW, J + 1>(V, lane, wave, gleft, C);
    }
}

// the J-th group of a lane, already in registers (MODE 1 / 2)
template <int J, int T = 256, bool A9 = false, bool AW = false>
__device__ __forceinline__ bool reg_piece(const uint4 &V, int tid, int wave, uint32_t gbase, uint32_t gleft, const Geo &g,
                                          const RowConst &C, uint32_t span) {
    const uint32_t pg = (uint32_t)(J * T + wave * 64);
    if (pg >= gleft) return false;
    const uint32_t row0 = 5u * (gbase + pg);
    if (row0 >= g.first && row0 + 320u <= g.end) {
        group_rows<false, A9, AW>(V, C, 0, 0);
    } else {
        const uint32_t tmp = 5u * (gbase + (uint32_t)(J * T + tid)) - g.first;
        group_rows<true, A9, AW>(V, C, tmp, span);
    }
    return true;
}

// the NL groups of a lane, one after the other, until the tile's groups end
template <int T, int NL, int J = 0, bool A9 = false, bool AW = false>
__device__ __forceinline__ void reg_pieces(const uint4 (&V)[NL], int tid, int wave, uint32_t gbase, uint32_t gleft, const Geo &g,
                                           const RowConst &C, uint32_t span) {
    if constexpr (J < NL) {
        if (reg_piece<J, T, A9, AW>(V[J], tid, wave, gbase, gleft, g, C, span))
            reg_pieces<T, NL, J + 1, A9, AW>(V, tid, wave, gbase, gleft, g, C, span);
    }
}


}  // namespace dense
}  // namespace memo

#endif  // MEMO_SWEEP_DENSE_H
