// tools/dispatch_probe.hip -- what launching the headline kernel's GRID costs with nothing in it: 107 760 workgroups of 256
// threads and 20 KiB of dynamic LDS (config 3 at k = 31: one workgroup per 928-position tile), each doing one 4-byte load,
// a barrier and one store -- against the same with 8 x fewer, 8 x fatter workgroups.  The floor under the per-tile pipeline.
// hipcc --offload-arch=gfx950 -O3 tools/dispatch_probe.hip -o tools/dispatch_probe && tools/dispatch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ __launch_bounds__(256) void touch(const unsigned *in, unsigned *out, int work) {
    extern __shared__ unsigned lds[];
    unsigned v = in[blockIdx.x];
    for (int i = 0; i < work; ++i) lds[threadIdx.x + 256 * i] = v + i;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = lds[(v & 255u)] + v;
}
int main() {
    const int n = 107760;
    unsigned *in, *out;
    hipMalloc(&in, n * 4); hipMalloc(&out, n * 4); hipMemset(in, 0, n * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int lds_kib : {0, 20, 40}) for (int grid : {n, n / 2, n / 8}) for (int work : {0, 20}) {
        std::vector<float> ms;
        for (int rep = 0; rep < 60; ++rep) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(touch, dim3(grid), dim3(256), lds_kib * 1024 + 1024 * work + 1024, 0, in, out, work);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float t; hipEventElapsedTime(&t, e0, e1); ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        printf("grid %6d x 256 threads, %2d KiB LDS asked, %2d LDS stores per thread: median %.4f ms (min %.4f) = %.2f ns per workgroup\n", grid, lds_kib + work + 1, work,
               ms[ms.size() / 2], ms[0], ms[ms.size() / 2] * 1e6 / grid);
    }
    return 0;
}
