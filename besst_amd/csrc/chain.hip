// Chain extraction of the linearised scaffold graph: the data-parallel part of MakeScaffolds.NewContigsScaffolds /
// UpdateInfo (BESST/MakeScaffolds.py:270-341, 344-482).
//
// After steps 1-4 (linearize.hip) every node (scaffold end) has at most one link edge, so the graph is a set of paths.
// The reference walks each path from one end, scaffold by scaffold: a scaffold entered through its 'L' end keeps its
// orientation, one entered through 'R' is flipped (:363-410); its contigs move by the running position, which then
// advances by the scaffold's length and by the gap of the link edge that is crossed next (at least 1, :468-471).
//
// Here every scaffold end h is a list element: "leave the scaffold through h" leads across h's link edge into the
// neighbour scaffold and on through ITS far end, with weight gap(h) + length(neighbour); an end without a link is a
// terminal.  Pointer doubling (ceil(log2(longest path)) + 1 passes over 2 x scaffolds elements) gives every end
//   term[h]  the terminal end reached by walking out through h (the end of the path on that side)
//   dist[h]  everything that lies beyond h on that side: lengths of the scaffolds there + the gaps in between
//   low[h]   the smallest node order (position in G.nodes()) among the nodes beyond h
// from which the host picks, per path, the start end (the terminal that comes first in node order, like the
// reference's scan of the component), each scaffold's position (dist of the end facing the start) and orientation,
// and numbers the new scaffolds in nx.connected_components order (component of the smallest node order first).
#include "common.h"

namespace besst {

namespace {

struct ChainState {
    int32_t* nxt;
    long long* dist;
    int32_t* low;
};

__global__ __launch_bounds__(256) void chain_init_kernel(int32_t n_nodes, const int32_t* __restrict__ link,
                                                         const int32_t* __restrict__ gap,
                                                         const int32_t* __restrict__ slen,
                                                         const int32_t* __restrict__ order, ChainState st) {
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= n_nodes) return;
    const int32_t v = link[h];
    if (v < 0 || v >= n_nodes) {                             // no link on this end: terminal
        st.nxt[h] = h;
        st.dist[h] = 0;
        st.low[h] = 0x7fffffff;
        return;
    }
    const int32_t far = v ^ 1;                               // on through the neighbour scaffold's other end
    st.nxt[h] = far;
    st.dist[h] = (long long)gap[h] + (long long)slen[v >> 1];
    const int32_t a = order[v], b = order[far];
    st.low[h] = a < b ? a : b;
}

// one doubling pass, in -> out; *changed is set when any element moved
__global__ __launch_bounds__(256) void chain_jump_kernel(int32_t n_nodes, ChainState in, ChainState out,
                                                         uint32_t* __restrict__ changed) {
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= n_nodes) return;
    const int32_t n = in.nxt[h];
    if (n == h) {                                            // terminal
        out.nxt[h] = h; out.dist[h] = in.dist[h]; out.low[h] = in.low[h];
        return;
    }
    const int32_t nn = in.nxt[n];
    if (nn == n) {                                           // already points at its terminal
        out.nxt[h] = n; out.dist[h] = in.dist[h]; out.low[h] = in.low[h];
        return;
    }
    out.nxt[h] = nn;
    out.dist[h] = in.dist[h] + in.dist[n];
    const int32_t a = in.low[h], b = in.low[n];
    out.low[h] = a < b ? a : b;
    *changed = 1u;
}

inline size_t chain_state_bytes(int64_t n_nodes) {
    return align_up((size_t)n_nodes * 4, 256) * 2 + align_up((size_t)n_nodes * 8, 256);
}

ChainState carve_state(char* p, int64_t n_nodes) {
    ChainState s;
    s.nxt = reinterpret_cast<int32_t*>(p); p += align_up((size_t)n_nodes * 4, 256);
    s.low = reinterpret_cast<int32_t*>(p); p += align_up((size_t)n_nodes * 4, 256);
    s.dist = reinterpret_cast<long long*>(p);
    return s;
}

}  // namespace

}  // namespace besst

using namespace besst;

extern "C" {

size_t besst_dev_chain_workspace_bytes(int64_t n_scaffolds) {
    const int64_t n = n_scaffolds > 0 ? 2 * n_scaffolds : 2;
    return 2 * chain_state_bytes(n) + 256;
}

int besst_dev_chain_scaffolds(void* stream, int64_t n_scaffolds, const int32_t* link, const int32_t* gap,
                              const int32_t* scaffold_length, const int32_t* node_order, void* workspace,
                              size_t workspace_bytes, int32_t* terminal, int64_t* beyond, int32_t* lowest_order,
                              int32_t* h_passes) {
    BESST_REQUIRE(n_scaffolds >= 0 && n_scaffolds < ((int64_t)1 << 30), "chain_scaffolds: scaffold count out of range");
    if (h_passes) *h_passes = 0;
    if (n_scaffolds == 0) return BESST_OK;
    BESST_REQUIRE(link && gap && scaffold_length && node_order && terminal && beyond && lowest_order,
                  "chain_scaffolds: null pointer");
    BESST_REQUIRE(workspace && workspace_bytes >= besst_dev_chain_workspace_bytes(n_scaffolds),
                  "chain_scaffolds: workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int32_t n = (int32_t)(2 * n_scaffolds);
    char* p = static_cast<char*>(workspace);
    ChainState st[2] = {carve_state(p, n), carve_state(p + chain_state_bytes(n), n)};
    uint32_t* changed = reinterpret_cast<uint32_t*>(p + 2 * chain_state_bytes(n));
    const dim3 grid((uint32_t)((n + 255) / 256)), block(256);
    hipLaunchKernelGGL(chain_init_kernel, grid, block, 0, s, n, link, gap, scaffold_length, node_order, st[0]);
    int cur = 0, passes = 0;
    for (; passes < 32; ++passes) {                          // 2^31 elements at most; a pass that moves nothing ends it
        BESST_HIP_TRY(hipMemsetAsync(changed, 0, 4, s));
        hipLaunchKernelGGL(chain_jump_kernel, grid, block, 0, s, n, st[cur], st[cur ^ 1], changed);
        cur ^= 1;
        uint32_t flag = 0;
        BESST_HIP_TRY(hipMemcpyAsync(&flag, changed, 4, hipMemcpyDeviceToHost, s));
        BESST_HIP_TRY(hipStreamSynchronize(s));
        if (!flag) { ++passes; break; }
    }
    BESST_HIP_TRY(hipMemcpyAsync(terminal, st[cur].nxt, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
    BESST_HIP_TRY(hipMemcpyAsync(beyond, st[cur].dist, (size_t)n * 8, hipMemcpyDeviceToDevice, s));
    BESST_HIP_TRY(hipMemcpyAsync(lowest_order, st[cur].low, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
    BESST_HIP_TRY(hipGetLastError());
    if (h_passes) *h_passes = passes;
    return BESST_OK;
}

int besst_chain_scaffolds(int device, int64_t n_scaffolds, const int32_t* link, const int32_t* gap,
                          const int32_t* scaffold_length, const int32_t* node_order, int32_t* terminal, int64_t* beyond,
                          int32_t* lowest_order, int32_t* passes) {
    BESST_REQUIRE(n_scaffolds >= 0 && n_scaffolds < ((int64_t)1 << 30), "chain_scaffolds: scaffold count out of range");
    if (passes) *passes = 0;
    if (n_scaffolds == 0) return BESST_OK;
    BESST_HIP_TRY(hipSetDevice(device));
    const size_t n = (size_t)n_scaffolds * 2;
    const size_t ws = besst_dev_chain_workspace_bytes(n_scaffolds);
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t o = off; off += align_up(bytes, 256); return o; };
    const size_t o_ws = carve(ws), o_link = carve(n * 4), o_gap = carve(n * 4), o_len = carve(n * 2), o_ord = carve(n * 4),
                 o_term = carve(n * 4), o_dist = carve(n * 8), o_low = carve(n * 4);
    char* d = nullptr;
    BESST_HIP_TRY(hipMalloc(&d, off));
    hipStream_t s = nullptr;
    int rc = BESST_OK;
    auto fail = [&](hipError_t e, const char* what) {
        set_error("chain_scaffolds: %s failed: %s", what, hipGetErrorString(e));
        rc = BESST_ERR_HIP;
    };
    hipError_t e = hipStreamCreate(&s);
    if (e != hipSuccess) fail(e, "hipStreamCreate");
    if (rc == BESST_OK) {
        if ((e = hipMemcpyAsync(d + o_link, link, n * 4, hipMemcpyHostToDevice, s)) != hipSuccess) fail(e, "copy in");
        else if ((e = hipMemcpyAsync(d + o_gap, gap, n * 4, hipMemcpyHostToDevice, s)) != hipSuccess) fail(e, "copy in");
        else if ((e = hipMemcpyAsync(d + o_len, scaffold_length, n * 2, hipMemcpyHostToDevice, s)) != hipSuccess) fail(e, "copy in");
        else if ((e = hipMemcpyAsync(d + o_ord, node_order, n * 4, hipMemcpyHostToDevice, s)) != hipSuccess) fail(e, "copy in");
    }
    if (rc == BESST_OK)
        rc = besst_dev_chain_scaffolds(s, n_scaffolds, (const int32_t*)(d + o_link), (const int32_t*)(d + o_gap),
                                       (const int32_t*)(d + o_len), (const int32_t*)(d + o_ord), d + o_ws, ws,
                                       (int32_t*)(d + o_term), (int64_t*)(d + o_dist), (int32_t*)(d + o_low), passes);
    if (rc == BESST_OK) {
        if ((e = hipMemcpyAsync(terminal, d + o_term, n * 4, hipMemcpyDeviceToHost, s)) != hipSuccess) fail(e, "copy out");
        else if ((e = hipMemcpyAsync(beyond, d + o_dist, n * 8, hipMemcpyDeviceToHost, s)) != hipSuccess) fail(e, "copy out");
        else if ((e = hipMemcpyAsync(lowest_order, d + o_low, n * 4, hipMemcpyDeviceToHost, s)) != hipSuccess) fail(e, "copy out");
        else if ((e = hipStreamSynchronize(s)) != hipSuccess) fail(e, "synchronize");
    }
    if (s) (void)hipStreamDestroy(s);
    (void)hipFree(d);
    return rc;
}

}  // extern "C"
