// Development probe: read-only streaming ceiling of the box (int4 loads, buffers larger than the 256 MiB MALL).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
template <int UNROLL>
__global__ __launch_bounds__(256) void rd(const int4* __restrict__ p, size_t n4, int* sink) {
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x);
    const size_t stride = (size_t)gridDim.x * 256;
    int acc = 0;
    for (; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
        int4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = p[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678) *sink = acc;
}
int main() {
    const size_t bytes = (size_t)2 << 30;
    int4* p; int* sink;
    hipMalloc(&p, bytes); hipMalloc(&sink, 4);
    hipMemset(p, 1, bytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int grid : {2048, 4096, 8192, 16384}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a);
            hipLaunchKernelGGL(rd<4>, dim3(grid), dim3(256), 0, 0, p, bytes / 16, sink);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep) printf("read 2 GiB grid %5d unroll 4: %.1f us  %.0f GB/s\n", grid, ms * 1000, bytes / ms / 1e6);
        }
    }
    // 220 MB-sized pass over a rotating window (what one stream_kernel launch moves)
    const size_t win = (size_t)220 << 20;
    for (int rep = 0; rep < 6; ++rep) {
        const size_t off = (rep % 8) * win / 16;
        hipEventRecord(a);
        hipLaunchKernelGGL(rd<4>, dim3(4096), dim3(256), 0, 0, p + off, win / 16, sink);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("read 220 MiB window %d: %.1f us  %.0f GB/s\n", rep, ms * 1000, win / ms / 1e6);
    }
    return 0;
}
