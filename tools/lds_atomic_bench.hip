// tools/lds_atomic_bench.hip -- what does an LDS atomic cost on gfx950, by access pattern?  (round 4: the k >= 65 conservation
// sweeps and the membership planes are bound by LDS atomics; SQ_LDS_BANK_CONFLICT is half of SQ_LDS_IDX_ACTIVE there.)
// One number per (operation, pattern, occupancy): shader cycles per wave-instruction and per CU, from the wall time of a
// launch in which every wave issues N of them back to back (the LDS pipe is shared by the CU's waves, so cycles per CU =
// time x clock / (N x waves per CU)); the clock is taken from s_memtime (shader cycles) over s_memrealtime (100 MHz).
//   hipcc --offload-arch=gfx950 -O3 tools/lds_atomic_bench.hip -o tools/lds_atomic_bench && tools/lds_atomic_bench
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e__ = (x);                                                         \
        if (e__ != hipSuccess) {                                                      \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e__));                  \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

enum Op { MIN32, OR32, MIN64, WRITE32, MINRTN32 };

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

// dword index of lane `lane` in iteration-independent pattern `pat` (cells: the array has 2048 dwords per wave)
__device__ __forceinline__ uint32_t cell_of(int pat, uint32_t lane, uint32_t wave_seed) {
    switch (pat) {
        case 0: return lane;                     // consecutive: conflict-free
        case 1: return 0;                        // every lane the same address
        case 2: return lane >> 1;                // pairs share an address
        case 3: return lane >> 2;                // fours share an address
        case 4: return lane >> 3;                // eights share an address
        case 5: return 2 * lane;                 // stride 2: lanes l and l + 16 share a bank (2-way, different addresses)
        case 6: return 4 * lane;                 // stride 4: 4-way
        case 7: return 32 * (lane & 31) + (lane >> 5);  // one bank per half-wave, 32 different addresses (32-way)
        case 8: return mix(lane * 2654435761u + wave_seed) & 1023u;   // random cells of 1024 (the first block of a row)
        case 9: return (lane * 4 / 5) + ((mix(lane + wave_seed) & 1u) ? 1024u : 0u);  // second block, config 3: start = 0.8 lane, two levels
        case 10: return (lane * 4 / 25) + ((mix(lane + wave_seed) & 1u) ? 1024u : 0u);  // second block, config 5: 25 rows per position
        case 11: return (mix(lane * 2654435761u + wave_seed) & 31u) * 33u;  // random cells, padded rows (bank = cell mod 32 still random)
        case 12: return lane ^ 1u;               // consecutive, neighbours swapped (still conflict-free)
    }
    return lane;
}

template <int OP>
__global__ __launch_bounds__(256) void bench(int pat, int iters, unsigned long long *clk) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t i = threadIdx.x; i < 4u * 3072u; i += 256) lds[i] = 0xFFFFFFFFu;
    __syncthreads();
    const uint32_t seed = blockIdx.x * 4u + wave;
    const uint32_t base = (uint32_t)(size_t)(__attribute__((address_space(3))) uint32_t *)lds + wave * 3072u * 4u;
    uint32_t addr = base + (OP == MIN64 ? 8u : 4u) * cell_of(pat, lane, seed);
    uint32_t data = mix(lane + seed) | 0x01000000u, data_hi = lane;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#define REP8(INS)                                                                                                     \
        asm volatile(INS " offset:0\n\t" INS " offset:256\n\t" INS " offset:512\n\t" INS " offset:768\n\t"          \
                     INS " offset:1024\n\t" INS " offset:1280\n\t" INS " offset:1536\n\t" INS " offset:1792\n\t"     \
                     :: "v"(addr), "v"(data) : "memory")
        if constexpr (OP == MIN32) REP8("ds_min_u32 %0, %1");
        if constexpr (OP == OR32) REP8("ds_or_b32 %0, %1");
        if constexpr (OP == WRITE32) REP8("ds_write_b32 %0, %1");
        if constexpr (OP == MIN64) {
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            const u32x2 d2 = {data, data_hi};
            asm volatile("ds_min_u64 %0, %1 offset:0\n\tds_min_u64 %0, %1 offset:512\n\tds_min_u64 %0, %1 offset:1024\n\t"
                         "ds_min_u64 %0, %1 offset:1536\n\tds_min_u64 %0, %1 offset:2048\n\tds_min_u64 %0, %1 offset:2560\n\t"
                         "ds_min_u64 %0, %1 offset:3072\n\tds_min_u64 %0, %1 offset:3584\n\t"
                         :: "v"(addr), "v"(d2) : "memory");
        }
        if constexpr (OP == MINRTN32) {
            uint32_t r[8];
            asm volatile("ds_min_rtn_u32 %0, %8, %9 offset:0\n\tds_min_rtn_u32 %1, %8, %9 offset:256\n\t"
                         "ds_min_rtn_u32 %2, %8, %9 offset:512\n\tds_min_rtn_u32 %3, %8, %9 offset:768\n\t"
                         "ds_min_rtn_u32 %4, %8, %9 offset:1024\n\tds_min_rtn_u32 %5, %8, %9 offset:1280\n\t"
                         "ds_min_rtn_u32 %6, %8, %9 offset:1536\n\tds_min_rtn_u32 %7, %8, %9 offset:1792\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
                         : "v"(addr), "v"(data) : "memory");
            data ^= r[0] & r[7] & 1u;
        }
        data += 0x00010000u;  // (a changing operand: the minimum keeps moving)
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        clk[0] = t1 - t0;
        clk[1] = r1 - r0;
    }
    __syncthreads();
    if (lds[threadIdx.x] == 0x12345u) clk[2] = 1;  // (keeps the LDS traffic alive)
}

template <int OP>
static void run(const char *name, int pat, int wg_per_cu, int iters, unsigned long long *d_clk) {
    const int grid = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    // 12 KiB per wave: three workgroups of four waves fit one CU
    const size_t lds = 4 * 3072 * 4;
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(bench<OP>, dim3(grid), dim3(256), lds, 0, pat, iters, d_clk);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
    }
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[3];
    CHECK(hipMemcpy(h, d_clk, sizeof h, hipMemcpyDeviceToHost));
    const double ghz = h[1] ? (double)h[0] / (double)h[1] * 0.1 : 0.0;
    const double instr_per_wave = 8.0 * iters;
    const double wave_cyc = (double)h[0] / instr_per_wave;                       // one wave's view (block 0, wave 0)
    const double cu_cyc = ms * 1e-3 * ghz * 1e9 / (instr_per_wave * 4.0 * wg_per_cu);  // LDS pipe: cycles per wave-instruction per CU
    printf("%-8s pat %2d  wg/cu %d  kernel %.4f ms  clock %.2f GHz  wave sees %.1f cyc/instr  CU spends %.2f cyc/instr\n", name, pat,
           wg_per_cu, ms, ghz, wave_cyc, cu_cyc);
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
}

int main() {
    unsigned long long *d_clk = nullptr;
    CHECK(hipMalloc(&d_clk, 64));
    CHECK(hipMemset(d_clk, 0, 64));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(bench<MIN32>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(bench<OR32>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(bench<MIN64>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(bench<WRITE32>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(bench<MINRTN32>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    const int iters = 2000;
    for (int wg : {1, 3}) {
        for (int pat = 0; pat <= 12; ++pat) run<MIN32>("min_u32", pat, wg, iters, d_clk);
        for (int pat : {0, 1, 2, 5, 7, 8}) run<OR32>("or_b32", pat, wg, iters, d_clk);
        for (int pat : {0, 1, 5, 8}) run<MIN64>("min_u64", pat, wg, iters, d_clk);
        for (int pat : {0, 1, 5, 8}) run<WRITE32>("write32", pat, wg, iters, d_clk);
        for (int pat : {0, 8}) run<MINRTN32>("minrtn32", pat, wg, iters, d_clk);
    }
    return 0;
}
