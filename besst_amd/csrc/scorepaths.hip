// ScorePaths on the link graph: the connectivity weights of a batch of candidate paths,
// ExtendLargeScaffolds.py:29-130 - the data-parallel piece of path extension (the path search stays on the host).
//
// A path is a list of contig ends; ends at even and odd positions alternate.  For every end of the path and every
// link edge at that end the reference asks where the other end sits in the path (:45-59 / :84-93):
//   calculate_connectivity                 even position: partner at an odd position -> good unless already visited,
//                                          otherwise bad; odd position: partner not at an even position -> bad,
//                                          at an even one that is still to come -> bad as well (:46-58)
//   calculate_connectivity_contamination   partner at a position of the other parity -> good, otherwise bad (:84-93);
//                                          the good weight is halved afterwards (:94)
// Sixteen lanes per path (four paths per wavefront): the path sits in LDS, the lanes take its ends in turn, walk
// that end's link edges and look every partner up by scanning the path (the lanes of a group read the same LDS word
// per step: a broadcast).  With a whole wavefront per path and the lanes spread over one end's edges, three of 64
// lanes had work (200 k paths: 5.3 ms).  Work per path is (sum of degrees) x (path length) compares - paths are tens of ends long, the search
// caps them at 100 (:559) - and the sums are exact integers, so the score (a float division of the two sums, :63-67)
// is formed on the host exactly as Python does.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "common.h"
#include "../../include/besst_amd.h"

namespace besst {

namespace {

constexpr int kGroup = 16;                    // lanes per path: candidate paths are tens of ends long
constexpr int kPathThreads = 256;
constexpr int kPathsPerBlock = kPathThreads / kGroup;
constexpr int kPathLds = 256;                 // ends of a path kept in LDS; longer paths are scanned from global memory

__global__ __launch_bounds__(kPathThreads) void score_paths_kernel(
    const int64_t* __restrict__ row_ptr, const int32_t* __restrict__ col, const int32_t* __restrict__ weight,
    const int64_t* __restrict__ path_ptr, const int32_t* __restrict__ path_nodes, int64_t n_paths, int contamination,
    long long* __restrict__ good_out, long long* __restrict__ bad_out) {
    __shared__ int32_t s_path[kPathsPerBlock][kPathLds];
    const int sub = threadIdx.x % kGroup, grp = threadIdx.x / kGroup;
    const int64_t p = (int64_t)blockIdx.x * kPathsPerBlock + grp;
    const bool live = p < n_paths;
    const int64_t begin = live ? path_ptr[p] : 0;
    const int len = live ? (int)(path_ptr[p + 1] - begin) : 0;
    const int32_t* path = path_nodes + begin;
    const bool in_lds = len <= kPathLds;
    if (in_lds)
        for (int j = sub; j < len; j += kGroup) s_path[grp][j] = path[j];
    __syncthreads();
    long long good = 0, bad = 0;
    // the lanes of a group take the ends of the path in turn; each walks that end's link edges and looks the partner
    // up by scanning the path (the lanes of a group read the same LDS word per step: a broadcast)
    for (int i = sub; i < len; i += kGroup) {
        const int32_t node = in_lds ? s_path[grp][i] : path[i];
        const int64_t e1 = row_ptr[node + 1];
        for (int64_t e = row_ptr[node]; e < e1; ++e) {
            const int32_t nbr = col[e];
            if ((nbr >> 1) == (node >> 1)) continue;           // the scaffold's own other end (:45, node[0] != nbr[0])
            const long long w = weight[e];
            bool at_odd = false, at_even = false, visited = false;
            for (int j = 0; j < len; ++j) {
                const int32_t x = in_lds ? s_path[grp][j] : path[j];
                if (x == nbr) {
                    if (j & 1) at_odd = true; else at_even = true;
                    if (j < i) visited = true;
                }
            }
            if (contamination) {
                if ((i & 1) == 0 ? at_odd : at_even) good += w; else bad += w;
            } else if ((i & 1) == 0) {
                if (at_odd) { if (!visited) good += w; }
                else bad += w;
            } else {
                if (!at_even) bad += w;
                else if (!visited) bad += w;
            }
        }
    }
#pragma unroll
    for (int d = kGroup / 2; d >= 1; d >>= 1) {
        good += __shfl_xor(good, d, kGroup);
        bad += __shfl_xor(bad, d, kGroup);
    }
    if (live && sub == 0) { good_out[p] = good; bad_out[p] = bad; }
}

}  // namespace

}  // namespace besst

using namespace besst;

extern "C" {

int besst_dev_score_paths(void* stream_, int64_t n_nodes, const int64_t* row_ptr, const int32_t* col,
                          const int32_t* weight, int64_t n_paths, const int64_t* path_ptr, const int32_t* path_nodes,
                          int32_t contamination, int64_t* good, int64_t* bad) {
    hipStream_t s = static_cast<hipStream_t>(stream_);
    BESST_REQUIRE(n_nodes >= 0 && n_nodes < ((int64_t)1 << 31), "score_paths: node count out of range");
    BESST_REQUIRE(n_paths >= 0 && n_paths < ((int64_t)1 << 31) * kPathsPerBlock, "score_paths: path count out of range");
    if (n_paths == 0) return BESST_OK;
    BESST_REQUIRE(row_ptr && path_ptr && path_nodes && good && bad, "score_paths: null pointer");
    const uint32_t blocks = (uint32_t)((n_paths + kPathsPerBlock - 1) / kPathsPerBlock);
    hipLaunchKernelGGL(score_paths_kernel, dim3(blocks), dim3(kPathThreads), 0, s, row_ptr, col, weight, path_ptr,
                       path_nodes, n_paths, contamination ? 1 : 0, reinterpret_cast<long long*>(good),
                       reinterpret_cast<long long*>(bad));
    BESST_HIP_TRY(hipGetLastError());
    return BESST_OK;
}

int besst_score_paths(int device, int64_t n_nodes, const int64_t* row_ptr, const int32_t* col, const int32_t* weight,
                      int64_t n_paths, const int64_t* path_ptr, const int32_t* path_nodes, int32_t contamination,
                      int64_t* good, int64_t* bad) {
    BESST_REQUIRE(n_nodes >= 0 && n_nodes < ((int64_t)1 << 31), "score_paths: node count out of range");
    BESST_REQUIRE(n_paths >= 0, "score_paths: negative path count");
    if (n_paths == 0) return BESST_OK;
    BESST_REQUIRE(row_ptr && path_ptr && path_nodes && good && bad, "score_paths: null pointer");
    const int64_t n_links = row_ptr[n_nodes], n_ends = path_ptr[n_paths];
    BESST_REQUIRE(n_links >= 0 && n_ends >= 0 && (n_links == 0 || (col && weight)), "score_paths: bad CSR arrays");
    for (int64_t i = 0; i < n_ends; ++i)
        BESST_REQUIRE(path_nodes[i] >= 0 && path_nodes[i] < n_nodes, "score_paths: path node out of range");
    BESST_HIP_TRY(hipSetDevice(device));
    size_t off = 0;
    auto carve = [&off](size_t bytes) { const size_t at = off; off += align_up(bytes + 8, 256); return at; };
    const size_t o_row = carve((size_t)(n_nodes + 1) * 8), o_col = carve((size_t)n_links * 4),
                 o_w = carve((size_t)n_links * 4), o_pp = carve((size_t)(n_paths + 1) * 8),
                 o_pn = carve((size_t)n_ends * 4), o_good = carve((size_t)n_paths * 8), o_bad = carve((size_t)n_paths * 8);
    char* d = nullptr;
    BESST_HIP_TRY(hipMalloc(&d, off));
    int rc = BESST_OK;
    hipError_t e = hipSuccess;
    auto up = [&](size_t at, const void* src, size_t bytes) {
        if (rc == BESST_OK && bytes && (e = hipMemcpy(d + at, src, bytes, hipMemcpyHostToDevice)) != hipSuccess) {
            set_error("score_paths: copy to the device failed: %s", hipGetErrorString(e));
            rc = BESST_ERR_HIP;
        }
    };
    up(o_row, row_ptr, (size_t)(n_nodes + 1) * 8);
    up(o_col, col, (size_t)n_links * 4);
    up(o_w, weight, (size_t)n_links * 4);
    up(o_pp, path_ptr, (size_t)(n_paths + 1) * 8);
    up(o_pn, path_nodes, (size_t)n_ends * 4);
    if (rc == BESST_OK)
        rc = besst_dev_score_paths(nullptr, n_nodes, (const int64_t*)(d + o_row), (const int32_t*)(d + o_col),
                                   (const int32_t*)(d + o_w), n_paths, (const int64_t*)(d + o_pp),
                                   (const int32_t*)(d + o_pn), contamination, (int64_t*)(d + o_good),
                                   (int64_t*)(d + o_bad));
    if (rc == BESST_OK) {
        if ((e = hipMemcpy(good, d + o_good, (size_t)n_paths * 8, hipMemcpyDeviceToHost)) != hipSuccess ||
            (e = hipMemcpy(bad, d + o_bad, (size_t)n_paths * 8, hipMemcpyDeviceToHost)) != hipSuccess) {
            set_error("score_paths: copy from the device failed: %s", hipGetErrorString(e));
            rc = BESST_ERR_HIP;
        }
    }
    (void)hipFree(d);
    return rc;
}

}  // extern "C"
