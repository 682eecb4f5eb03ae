// memo_sweep_fold.h -- the register fold step shared by the conservation sweeps (memo_sweep_cons.hip, memo_sweep_cons3t.hip)
#ifndef MEMO_SWEEP_FOLD_H
#define MEMO_SWEEP_FOLD_H

#include "memo_sweep.h"

namespace memo {

// M_(j-1)[x] = min(L[x], M_j[x], M_j[x - half]),  half = 2^J cells, on a lane's four cells, IN PLACE (one asm block
// per step: the compiler, left to itself, computes into fresh registers and copies them back at the join of the
// wave-uniform branch around the step).  The DPP operations come first -- they read the left lane's M before any
// lane overwrites it -- and fold their operand into L; s_nop 1: a DPP source written by the instruction before
// needs two wait states, and the compiler does not see into the string.
#define MEMO_DPP_MIN(dst, src) "v_min_u32_dpp " dst ", " src ", " dst " wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
template <int J>
__device__ __forceinline__ void fold_step_dpp(uint4 &M, uint4 L, int lane) {
    if constexpr (J >= 3) {
        const int src = (lane - (1 << (J - 2))) << 2;  // (negative: context lanes, whose result is dropped)
        const uint32_t sx = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)M.x), sy = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)M.y);
        const uint32_t sz = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)M.z), sw = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)M.w);
        M = make_uint4(min(L.x, min(M.x, sx)), min(L.y, min(M.y, sy)), min(L.z, min(M.z, sz)), min(L.w, min(M.w, sw)));
    } else if constexpr (J == 2) {
        asm("s_nop 1\n\t" MEMO_DPP_MIN("%4", "%0") MEMO_DPP_MIN("%5", "%1") MEMO_DPP_MIN("%6", "%2") MEMO_DPP_MIN("%7", "%3")
            "v_min_u32 %0, %4, %0\n\tv_min_u32 %1, %5, %1\n\tv_min_u32 %2, %6, %2\n\tv_min_u32 %3, %7, %3"
            : "+v"(M.x), "+v"(M.y), "+v"(M.z), "+v"(M.w), "+v"(L.x), "+v"(L.y), "+v"(L.z), "+v"(L.w));
    } else if constexpr (J == 1) {
        asm("s_nop 1\n\t" MEMO_DPP_MIN("%4", "%2") MEMO_DPP_MIN("%5", "%3")
            "v_min3_u32 %2, %6, %2, %0\n\tv_min3_u32 %3, %7, %3, %1\n\tv_min_u32 %0, %4, %0\n\tv_min_u32 %1, %5, %1"
            : "+v"(M.x), "+v"(M.y), "+v"(M.z), "+v"(M.w), "+v"(L.x), "+v"(L.y) : "v"(L.z), "v"(L.w));
    } else {
        asm("s_nop 1\n\t" MEMO_DPP_MIN("%4", "%3")
            "v_min3_u32 %3, %7, %3, %2\n\tv_min3_u32 %2, %6, %2, %1\n\tv_min3_u32 %1, %5, %1, %0\n\tv_min_u32 %0, %4, %0"
            : "+v"(M.x), "+v"(M.y), "+v"(M.z), "+v"(M.w), "+v"(L.x) : "v"(L.y), "v"(L.z), "v"(L.w));
    }
}
#undef MEMO_DPP_MIN

}  // namespace memo

#endif  // MEMO_SWEEP_FOLD_H
