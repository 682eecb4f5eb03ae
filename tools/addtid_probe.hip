// tools/addtid_probe.hip -- where does ds_write_addtid_b32 store?  (round 4: the clear of the level arrays)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(unsigned *out) {
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = 0;
    __syncthreads();
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned *)lds;
    const unsigned m0v = __builtin_amdgcn_readfirstlane(base + (threadIdx.x >> 6) * 256u);
    unsigned keep;
    const unsigned v = 0x1000u + threadIdx.x;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 1\n\tds_write_addtid_b32 %2 offset:0\n\tds_write_addtid_b32 %2 offset:4096\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(m0v), "v"(v) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 256) out[i] = lds[i];
}
int main() {
    unsigned *d, h[2048];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 8192, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int shown = 0;
    for (int i = 0; i < 2048 && shown < 24; ++i)
        if (h[i] && (i % 64 == 0 || i % 64 == 63)) printf("lds[%d] = 0x%x (thread %u)\n", i, h[i], h[i] - 0x1000u), ++shown;
    int n = 0; for (int i = 0; i < 2048; ++i) n += h[i] != 0;
    printf("%d words written\n", n);
    return 0;
}
