// Development probe: read (or write) bandwidth of P-byte pieces laid out at stride S, with an optional per-piece skew -
// the access pattern of the record loop's block segments (14 KB used of every 128 KB).
//   hipcc --offload-arch=gfx950 -O3 tools/probe_stride.hip -o /tmp/probe_stride && /tmp/probe_stride
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

__global__ __launch_bounds__(256) void read_kernel(const uint4* __restrict__ src, size_t stride16, uint32_t piece16, uint32_t skew_mul,
                                                   uint32_t wrap16, unsigned long long* __restrict__ sink) {
    const size_t base = (size_t)blockIdx.x * stride16;
    const uint32_t skew = skew_mul ? ((blockIdx.x * skew_mul) % wrap16) & ~15u : 0u;
    uint32_t acc = 0;
    for (uint32_t j = threadIdx.x; j < piece16; j += 256) {
        const uint32_t jj = wrap16 ? (j + skew) % wrap16 : j;
        const uint4 v = src[base + jj];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

__global__ __launch_bounds__(256) void write_kernel(uint4* __restrict__ dst, size_t stride16, uint32_t piece16, uint32_t skew_mul,
                                                    uint32_t wrap16) {
    const size_t base = (size_t)blockIdx.x * stride16;
    const uint32_t skew = skew_mul ? ((blockIdx.x * skew_mul) % wrap16) & ~15u : 0u;
    for (uint32_t j = threadIdx.x; j < piece16; j += 256) {
        const uint32_t jj = wrap16 ? (j + skew) % wrap16 : j;
        dst[base + jj] = make_uint4(j, jj, blockIdx.x, 7u);
    }
}

int main() {
    const size_t pieces = 49152;                 // 2 x 24.4 k segments
    const size_t S = 128 << 10;
    uint4* buf;
    unsigned long long* sink;
    hipMalloc(&buf, pieces * S);
    hipMalloc(&sink, 8);
    hipMemset(buf, 1, pieces * S);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    struct Case { const char* name; size_t stride; uint32_t piece; uint32_t skew; uint32_t wrap; } cases[] = {
        {"dense 14 KB pieces", 14336, 14336, 0, 0},
        {"14 KB of every 128 KB", S, 14336, 0, 0},
        {"14 KB of every 128 KB, skewed start (wrap)", S, 14336, 2731 * 16, (uint32_t)(S / 16)},
        {"14 KB of every 136 KB (odd stride)", S + 8192, 14336, 0, 0},
        {"28 KB of every 256 KB", 2 * S, 28672, 0, 0},
        {"14 KB of every 32 KB", 32768, 14336, 0, 0},
        {"14 KB of every 16 KB", 16384, 14336, 0, 0},
    };
    for (auto& c : cases) {
        const size_t n = c.stride * pieces > pieces * S ? pieces * S / c.stride : pieces;
        for (int mode = 0; mode < 2; ++mode) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(a);
                if (mode == 0) hipLaunchKernelGGL(read_kernel, dim3(n), dim3(256), 0, 0, buf, c.stride / 16, c.piece / 16, c.skew / 16, c.wrap, sink);
                else hipLaunchKernelGGL(write_kernel, dim3(n), dim3(256), 0, 0, buf, c.stride / 16, c.piece / 16, c.skew / 16, c.wrap);
                hipEventRecord(b);
                hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
            }
            printf("%-46s %s %8.3f ms  %7.1f GB/s\n", c.name, mode ? "write" : "read ", best, n * (double)c.piece / best / 1e6);
        }
    }
    return 0;
}
