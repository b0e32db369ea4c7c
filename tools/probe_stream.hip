// Development probe (not part of the product): bandwidth ladder for the streaming pass.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int SUB, int LEVEL>
__global__ __launch_bounds__(256) void probe(const int32_t* __restrict__ tid, const int32_t* __restrict__ mtid,
                                             const uint8_t* __restrict__ mapq, const uint16_t* __restrict__ qlen,
                                             int64_t n, unsigned long long* __restrict__ aligned,
                                             unsigned long long* __restrict__ bitmask, int* __restrict__ sink) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t base = (int64_t)blockIdx.x * (1024 * SUB);
    if (base + 1024 * SUB > n) return;
    int4 a[SUB], b[SUB];
    uchar4 m[SUB];
    ushort4 q[SUB];
#pragma unroll
    for (int s = 0; s < SUB; ++s) {
        const int64_t i0 = base + s * 1024 + t * 4;
        a[s] = *reinterpret_cast<const int4*>(tid + i0);
        if (LEVEL >= 1) {
            b[s] = *reinterpret_cast<const int4*>(mtid + i0);
            m[s] = *reinterpret_cast<const uchar4*>(mapq + i0);
            q[s] = *reinterpret_cast<const ushort4*>(qlen + i0);
        }
    }
    int acc = 0;
    int32_t acc_ref = -1;
    int acc_sum = 0;
#pragma unroll
    for (int s = 0; s < SUB; ++s) {
        if (LEVEL == 0) { acc ^= a[s].x ^ a[s].y ^ a[s].z ^ a[s].w; continue; }
        const int32_t x[4] = {a[s].x, a[s].y, a[s].z, a[s].w};
        const int32_t y[4] = {b[s].x, b[s].y, b[s].z, b[s].w};
        const uint32_t mq[4] = {m[s].x, m[s].y, m[s].z, m[s].w};
        const uint32_t ql[4] = {q[s].x, q[s].y, q[s].z, q[s].w};
        bool cand[4];
        int mine = 0;
        bool uni = true;
        const int32_t ref = __builtin_amdgcn_readfirstlane(x[0]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            cand[k] = x[k] != y[k];
            uni = uni && x[k] == ref;
            if (!cand[k] && (mq[k] >= 11 || mq[k] == 0)) mine += ql[k];
        }
        if (LEVEL == 1) { acc += mine + (cand[0] ? 1 : 0) + (cand[3] ? 2 : 0); continue; }
        const unsigned long long b0 = __ballot(cand[0]), b1 = __ballot(cand[1]), b2 = __ballot(cand[2]), b3 = __ballot(cand[3]);
        if (lane < 4) {
            const int64_t g = ((int64_t)blockIdx.x * SUB + s) * 4 + wave;
            bitmask[g * 4 + lane] = lane == 0 ? b0 : lane == 1 ? b1 : lane == 2 ? b2 : b3;
        }
        if (LEVEL == 2) { acc += mine; continue; }
        if (__all(uni)) {
            int tot = mine;
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) tot += __shfl_xor(tot, d, 64);
            if (ref != acc_ref) {
                if (lane == 0 && acc_sum && acc_ref >= 0) atomicAdd(&aligned[acc_ref], (unsigned long long)acc_sum);
                acc_ref = ref; acc_sum = 0;
            }
            acc_sum += tot;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (!cand[k] && x[k] >= 0 && (mq[k] >= 11 || mq[k] == 0)) atomicAdd(&aligned[x[k]], (unsigned long long)ql[k]);
        }
    }
    if (LEVEL == 3 && lane == 0 && acc_sum && acc_ref >= 0) atomicAdd(&aligned[acc_ref], (unsigned long long)acc_sum);
    if (LEVEL < 3 && acc == 0x7fffffff) sink[0] = acc;
}

template <int SUB, int LEVEL>
static float run(const int32_t* tid, const int32_t* mtid, const uint8_t* mapq, const uint16_t* qlen, int64_t n,
                 unsigned long long* aligned, unsigned long long* bitmask, int* sink, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = (int)(n / (1024 * SUB));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((probe<SUB, LEVEL>), dim3(blocks), dim3(256), 0, 0, tid, mtid, mapq, qlen, n, aligned, bitmask, sink);
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((probe<SUB, LEVEL>), dim3(blocks), dim3(256), 0, 0, tid, mtid, mapq, qlen, n, aligned, bitmask, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps * 1000.f;
}

extern "C" void probe_all(const int32_t* tid, const int32_t* mtid, const uint8_t* mapq, const uint16_t* qlen, int64_t n,
                          unsigned long long* aligned, unsigned long long* bitmask, int* sink) {
    const int reps = 20;
#define R(S, L) printf("SUB=%d LEVEL=%d : %8.1f us\n", S, L, run<S, L>(tid, mtid, mapq, qlen, n, aligned, bitmask, sink, reps));
    R(1, 0) R(2, 0) R(4, 0) R(8, 0)
    R(1, 1) R(2, 1) R(4, 1) R(8, 1)
    R(1, 2) R(2, 2) R(4, 2)
    R(1, 3) R(2, 3) R(4, 3) R(8, 3)
    fflush(stdout);
}
