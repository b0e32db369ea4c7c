// Scaffold-graph linearisation on the emitted edge table: steps 1-4 of the reference's
// MakeScaffolds.Algorithm (MakeScaffolds.py:75-82) as data-parallel kernels.
//
//   step 1/3  RemoveIsolatedContigs (:134-144)              a scaffold without link edges on either side goes
//   step 2    RemoveAmbiguousRegionsUsingScore (:206-241)   + remove_edges (:156-204)
//   step 4    RemoveLoops (:248-274)                        scaffolds on cycles go
//
// Step 2 is sequential in the reference: edges are visited by descending score and remove_edges runs on edge[0],
// then edge[1]; what a node decides depends on what its earlier-visited neighbours already removed.  Only a node's
// FIRST visit can change anything (afterwards it holds at most one scoring edge and never gains one), and the
// time of that visit is fixed up front: it is the node's best edge in (score desc, G.edges() index asc) order,
// edge[0] before edge[1].  So the sequential sweep is a priority-ordered process on the nodes, and it is replayed
// exactly by rounds: in every round the nodes all of whose not-yet-decided neighbours come later decide at once
// (two neighbours are never both ready).  The number of rounds is the longest chain of link-adjacent nodes with
// increasing visit times - a handful on real graphs; the loop runs until nothing is pending.
//
// Step 4: after step 2 a node has at most one link edge, so components are paths or single cycles; a node is on a
// cycle iff the walk "cross the scaffold, follow the link" never ends, which pointer doubling decides in
// ceil(log2(n)) + 1 passes.  The union of the cycle basis' nodes does not depend on the traversal order.
//
// Nodes are compact ids: scaffold k has the nodes 2k ('L') and 2k+1 ('R').
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "common.h"
#include "../../include/besst_amd.h"

namespace besst {

namespace {

constexpr int kLinThreads = 256;
constexpr int32_t kNever = 0x7fffffff;

struct LinArgs {
    const int32_t* a;
    const int32_t* b;
    const double* score;
    uint32_t m, n_nodes, n_scaf;
    int steps;                        // bit 0: step 1, bit 1: step 2, bit 2: step 3, bit 3: step 4
    // per node
    unsigned long long* best_bits;    // bits of the best (largest positive) score over the node's edges
    uint32_t* best_i;                 // smallest edge index among those with that score
    uint32_t* deg;                    // link edges (step 1) / surviving link edges (step 3)
    int32_t* done_at;                 // round in which the node decided (kNever before)
    int32_t* blocked_at;              // last round in which an earlier undecided neighbour was seen
    unsigned long long* top1;         // best and runner-up score bits among the live scoring edges at decision time
    unsigned long long* top2;
    uint32_t* cnt;
    uint32_t* ntop;
    uint8_t* amb;                     // 1: the node found its two best scores within 0.8 of each other
    int32_t* mate;
    int32_t* jump[2];
    int32_t* lab[2];
    // per edge
    uint8_t* alive;
    uint32_t* list[2];                // pending edges (an undecided endpoint), rebuilt every round; round 0 reads all edges
    // per scaffold
    uint8_t* removed_by;              // 0: still there; 1 / 3 / 4: the step that removed the scaffold
    // [0] isolated step 1, [1] isolated step 3, [2] directed cycles, [3] ambivalent nodes, [4..5] lengths of the two
    // pending-edge lists, [7] changes of the last counted doubling pass, [6] link edges at nodes with more than one (step 4 without step 2)
    unsigned long long* counters;
};

// Sum of v over the workgroup, added to *counter by ONE atomic (every thread of the workgroup must call it).
// Per-wave atomics on one address serialise device-wide: with a wave per 64 nodes the counting kernels spent
// 0.3 - 0.7 ms in them on a 4 M-node graph.
__device__ __forceinline__ void block_add(uint32_t v, unsigned long long* counter) {
    __shared__ uint32_t s_part[kLinThreads / 64];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
#pragma unroll
        for (int w = 0; w < kLinThreads / 64; ++w) tot += s_part[w];
        if (tot) atomicAdd(counter, (unsigned long long)tot);
    }
}

__device__ __forceinline__ unsigned long long score_bits(double s) { return (unsigned long long)__double_as_longlong(s); }

// visit time of x before that of y?  (best score desc, edge index asc, edge[0] before edge[1])
__device__ __forceinline__ bool earlier(const LinArgs& g, uint32_t x, uint32_t y) {
    const unsigned long long bx = g.best_bits[x], by = g.best_bits[y];
    if (bx != by) return bx > by;
    const uint32_t ix = g.best_i[x], iy = g.best_i[y];
    if (ix != iy) return ix < iy;
    return (uint32_t)g.a[ix] == x;            // same edge: its edge[0] end is visited first
}

__global__ __launch_bounds__(kLinThreads) void lin_best_kernel(LinArgs g) {
    const uint32_t i = blockIdx.x * kLinThreads + threadIdx.x;
    if (i >= g.m) return;
    const uint32_t a = g.a[i], b = g.b[i];
    atomicAdd(&g.deg[a], 1u);
    atomicAdd(&g.deg[b], 1u);
    const double s = g.score[i];
    const bool pos = 0.0 < s;                  // 'zero-scoring' in the reference is "not 0 < score" (:163)
    const bool step2 = (g.steps & 2) != 0;
    g.alive[i] = (pos || !step2) ? 1 : 0;
    if (pos && step2) {
        atomicMax(&g.best_bits[a], score_bits(s));
        atomicMax(&g.best_bits[b], score_bits(s));
    }
}

__global__ __launch_bounds__(kLinThreads) void lin_best_index_kernel(LinArgs g) {
    const uint32_t i = blockIdx.x * kLinThreads + threadIdx.x;
    if (i >= g.m || !g.alive[i]) return;
    const unsigned long long sb = score_bits(g.score[i]);
    const uint32_t a = g.a[i], b = g.b[i];
    if (g.best_bits[a] == sb) atomicMin(&g.best_i[a], i);
    if (g.best_bits[b] == sb) atomicMin(&g.best_i[b], i);
}

// steps 1 and 3: which = 0 / 1
__global__ __launch_bounds__(kLinThreads) void lin_isolated_kernel(LinArgs g, int which) {
    uint32_t gone = 0;
    for (uint32_t k = blockIdx.x * kLinThreads + threadIdx.x; k < g.n_scaf; k += gridDim.x * kLinThreads)
        if (!g.removed_by[k] && g.deg[2 * k] == 0 && g.deg[2 * k + 1] == 0) {
            g.removed_by[k] = which ? 3 : 1;
            ++gone;
        }
    block_add(gone, &g.counters[which]);
}

__device__ __forceinline__ bool is_done(const LinArgs& g, uint32_t x, int round) { return g.done_at[x] < round; }
__device__ __forceinline__ bool is_ready(const LinArgs& g, uint32_t x, int round) {
    return !is_done(g, x, round) && g.blocked_at[x] != round;
}

// The edges a round has to look at: those with an undecided endpoint when the round began.  lin_block_kernel of
// round r reads the list of round r - 1 (all edges in round 0) and writes the list the other three phases of
// round r - and the block phase of round r + 1 - iterate; counters[4 + parity] hold the list lengths.  Most nodes
// decide in the first two rounds, so later rounds touch a small fraction of the edges.
__device__ __forceinline__ uint32_t list_len(const LinArgs& g, int which) { return (uint32_t)g.counters[4 + which]; }

// round, phase 1: the later endpoint of an edge between two undecided nodes has to wait.  A workgroup looks at
// kBlockItems * 256 list entries and reserves its share of the next list with one atomic.
constexpr int kBlockItems = 8;
__global__ __launch_bounds__(kLinThreads) void lin_block_kernel(LinArgs g, int round) {
    __shared__ uint32_t s_wave[kLinThreads / 64];
    __shared__ uint32_t s_base;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int cur = round & 1, next = cur ^ 1;
    const uint32_t n_cur = round == 0 ? g.m : list_len(g, cur);
    const uint32_t base = blockIdx.x * (kLinThreads * kBlockItems);
    uint32_t idx[kBlockItems];
    uint32_t mine = 0, flags = 0;
#pragma unroll
    for (int r = 0; r < kBlockItems; ++r) {
        const uint32_t j = base + r * kLinThreads + t;
        idx[r] = 0;
        if (j >= n_cur) continue;
        const uint32_t i = round == 0 ? j : g.list[cur][j];
        idx[r] = i;
        if (!g.alive[i]) continue;
        const uint32_t a = g.a[i], b = g.b[i];
        const bool da = is_done(g, a, round), db = is_done(g, b, round);
        if (!da && !db) g.blocked_at[earlier(g, a, b) ? b : a] = round;
        if (!da || !db) { flags |= 1u << r; ++mine; }
    }
    // exclusive scan of the per-thread counts over the workgroup
    uint32_t x = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(x, d, 64);
        if (lane >= d) x += o;
    }
    if (lane == 63) s_wave[wave] = x;
    __syncthreads();
    uint32_t before = x - mine, total = 0;
#pragma unroll
    for (int w = 0; w < kLinThreads / 64; ++w) {
        if (w < wave) before += s_wave[w];
        total += s_wave[w];
    }
    if (t == 0) s_base = total ? (uint32_t)atomicAdd(&g.counters[4 + next], (unsigned long long)total) : 0u;
    __syncthreads();
    uint32_t out = s_base + before;
#pragma unroll
    for (int r = 0; r < kBlockItems; ++r)
        if (flags & (1u << r)) g.list[next][out++] = idx[r];
}

// phase 2: best score and number of live scoring edges of every ready node
__global__ __launch_bounds__(kLinThreads) void lin_top_kernel(LinArgs g, int round) {
    const uint32_t j = blockIdx.x * kLinThreads + threadIdx.x;
    if (j >= list_len(g, (round & 1) ^ 1)) return;
    const uint32_t i = g.list[(round & 1) ^ 1][j];
    const unsigned long long sb = score_bits(g.score[i]);
    const uint32_t ends[2] = {(uint32_t)g.a[i], (uint32_t)g.b[i]};
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const uint32_t x = ends[e];
        if (!is_ready(g, x, round)) continue;
        atomicMax(&g.top1[x], sb);
        atomicAdd(&g.cnt[x], 1u);
    }
}

// phase 3: how many edges share the best score, and the best score below it
__global__ __launch_bounds__(kLinThreads) void lin_second_kernel(LinArgs g, int round) {
    const uint32_t j = blockIdx.x * kLinThreads + threadIdx.x;
    if (j >= list_len(g, (round & 1) ^ 1)) return;
    const uint32_t i = g.list[(round & 1) ^ 1][j];
    const unsigned long long sb = score_bits(g.score[i]);
    const uint32_t ends[2] = {(uint32_t)g.a[i], (uint32_t)g.b[i]};
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const uint32_t x = ends[e];
        if (!is_ready(g, x, round) || g.cnt[x] < 2) continue;
        if (sb == g.top1[x]) atomicAdd(&g.ntop[x], 1u);
        else atomicMax(&g.top2[x], sb);
    }
}

// phase 4: the ready nodes decide (MakeScaffolds.py:181-188) and are done from the next round on
__global__ __launch_bounds__(kLinThreads) void lin_decide_kernel(LinArgs g, int round) {
    const uint32_t j = blockIdx.x * kLinThreads + threadIdx.x;
    if (j >= list_len(g, (round & 1) ^ 1)) return;
    const uint32_t i = g.list[(round & 1) ^ 1][j];
    const unsigned long long sb = score_bits(g.score[i]);
    const uint32_t ends[2] = {(uint32_t)g.a[i], (uint32_t)g.b[i]};
    bool drop = false;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const uint32_t x = ends[e];
        if (!is_ready(g, x, round)) continue;
        g.done_at[x] = round;                       // same value from every edge of x; readers compare with < round
        if (g.cnt[x] < 2) continue;
        const unsigned long long t1 = g.top1[x];
        const unsigned long long t2 = g.ntop[x] >= 2 ? t1 : g.top2[x];
        const double s1 = __longlong_as_double((long long)t1), s2 = __longlong_as_double((long long)t2);
        if (s2 / s1 > 0.8) {
            drop = true;
            if (sb == t1) {                         // the holders of the top score record the event (same values)
                g.amb[x] = 1;
                if (g.ntop[x] >= 2) g.top2[x] = t1;
            }
        } else if (sb != t1) {
            drop = true;
        }
    }
    if (drop) g.alive[i] = 0;
}

__global__ __launch_bounds__(kLinThreads) void lin_count_amb_kernel(LinArgs g) {
    uint32_t amb = 0;
    for (uint32_t x = blockIdx.x * kLinThreads + threadIdx.x; x < g.n_nodes; x += gridDim.x * kLinThreads)
        amb += g.amb[x] ? 1u : 0u;
    block_add(amb, &g.counters[3]);
}

__global__ __launch_bounds__(kLinThreads) void lin_degree_kernel(LinArgs g) {
    const uint32_t i = blockIdx.x * kLinThreads + threadIdx.x;
    if (i >= g.m || !g.alive[i]) return;
    atomicAdd(&g.deg[g.a[i]], 1u);
    atomicAdd(&g.deg[g.b[i]], 1u);
}

__global__ __launch_bounds__(kLinThreads) void lin_mate_kernel(LinArgs g) {
    const uint32_t i = blockIdx.x * kLinThreads + threadIdx.x;
    if (i >= g.m || !g.alive[i]) return;
    const uint32_t a = g.a[i], b = g.b[i];
    if (g.removed_by[a >> 1] || g.removed_by[b >> 1]) return;
    if (g.deg[a] > 1 || g.deg[b] > 1) {              // not the graph step 2 leaves: the caller gets an error
        atomicAdd(&g.counters[6], 1ull);
        return;
    }
    g.mate[a] = (int32_t)b;                          // at most one live link edge per node after step 2
    g.mate[b] = (int32_t)a;
}

// walk: enter a scaffold at x, leave through its other end x ^ 1, follow that end's link
__global__ __launch_bounds__(kLinThreads) void lin_walk_init_kernel(LinArgs g) {
    const uint32_t x = blockIdx.x * kLinThreads + threadIdx.x;
    if (x >= g.n_nodes) return;
    g.jump[0][x] = g.mate[x ^ 1];
    g.lab[0][x] = (int32_t)x;
}

__global__ __launch_bounds__(kLinThreads) void lin_walk_double_kernel(LinArgs g, int from) {
    const uint32_t x = blockIdx.x * kLinThreads + threadIdx.x;
    if (x >= g.n_nodes) return;
    const int32_t j = g.jump[from][x];
    int32_t l = g.lab[from][x];
    int32_t jj = -1;
    if (j >= 0) {
        jj = g.jump[from][j];
        const int32_t lj = g.lab[from][j];
        l = lj < l ? lj : l;
    }
    g.jump[from ^ 1][x] = jj;
    g.lab[from ^ 1][x] = l;
}

// Same pass, counting what it changed (a walk that ended, a label that fell).  A pass that changes nothing proves
// that every finite walk has ended (a longer one would have a suffix ending in this pass) and that every cycle's
// minimum has gone all the way round, so the host stops doubling there instead of after ceil(log2 n) + 1 passes.
__global__ __launch_bounds__(kLinThreads) void lin_walk_double_count_kernel(LinArgs g, int from) {
    uint32_t changed = 0;
    for (uint32_t x = blockIdx.x * kLinThreads + threadIdx.x; x < g.n_nodes; x += gridDim.x * kLinThreads) {
        const int32_t j = g.jump[from][x];
        const int32_t l0 = g.lab[from][x];
        int32_t l = l0, jj = -1;
        if (j >= 0) {
            jj = g.jump[from][j];
            const int32_t lj = g.lab[from][j];
            l = lj < l ? lj : l;
        }
        g.jump[from ^ 1][x] = jj;
        g.lab[from ^ 1][x] = l;
        changed += ((j >= 0 && jj < 0) || l != l0) ? 1u : 0u;
    }
    block_add(changed, &g.counters[7]);
}

__global__ __launch_bounds__(kLinThreads) void lin_cycle_kernel(LinArgs g, int from) {
    uint32_t heads = 0;
    for (uint32_t x = blockIdx.x * kLinThreads + threadIdx.x; x < g.n_nodes; x += gridDim.x * kLinThreads)
        if (g.jump[from][x] >= 0) {                   // the walk from x never ends: x is on a cycle
            g.removed_by[x >> 1] = 4;
            heads += g.lab[from][x] == (int32_t)x;    // one head per directed cycle, two directions per cycle
        }
    block_add(heads, &g.counters[2]);
}

size_t carve(size_t& off, size_t bytes) {
    const size_t at = off;
    off += align_up(bytes, 256);
    return at;
}

struct LinLayout {
    size_t best_bits, best_i, deg, done_at, blocked_at, top1, top2, cnt, ntop, amb, mate, jump0, jump1, lab0, lab1,
        alive, present, counters, list0, list1, total;
};

LinLayout lin_layout(int64_t n_scaf, int64_t m) {
    const size_t n = (size_t)(n_scaf > 0 ? n_scaf : 1) * 2, e = (size_t)(m > 0 ? m : 1);
    LinLayout L;
    size_t off = 0;
    // zero-initialised block first ...
    L.best_bits = carve(off, n * 8);
    L.deg = carve(off, n * 4);
    L.top1 = carve(off, n * 8);
    L.top2 = carve(off, n * 8);
    L.cnt = carve(off, n * 4);
    L.ntop = carve(off, n * 4);
    L.amb = carve(off, n);
    L.counters = carve(off, 8 * 8);
    // ... then the 0xff-initialised one (best_i = none, mate = -1, blocked_at = -1) ...
    L.best_i = carve(off, n * 4);
    L.mate = carve(off, n * 4);
    L.blocked_at = carve(off, n * 4);
    // ... and the rest
    L.done_at = carve(off, n * 4);
    L.jump0 = carve(off, n * 4);
    L.jump1 = carve(off, n * 4);
    L.lab0 = carve(off, n * 4);
    L.lab1 = carve(off, n * 4);
    L.alive = carve(off, e);
    L.present = carve(off, n / 2);
    L.list0 = carve(off, e * 4);
    L.list1 = carve(off, e * 4);
    L.total = off;
    return L;
}

__global__ __launch_bounds__(kLinThreads) void lin_fill_kernel(int32_t* p, uint32_t n, int32_t v) {
    const uint32_t i = blockIdx.x * kLinThreads + threadIdx.x;
    if (i < n) p[i] = v;
}

}  // namespace

}  // namespace besst

using namespace besst;

extern "C" {

size_t besst_dev_linearize_workspace_bytes(int64_t n_scaffolds, int64_t n_edges) {
    return lin_layout(n_scaffolds, n_edges).total;
}

int besst_dev_linearize(void* stream_, int32_t steps, int64_t n_scaffolds, int64_t n_edges, const int32_t* a, const int32_t* b,
                        const double* score, void* workspace, size_t workspace_bytes, uint8_t* edge_alive,
                        uint8_t* scaffold_removed_by, uint8_t* node_ambivalent, double* node_top, double* node_second,
                        uint32_t* node_best_edge, int64_t* counters) {
    hipStream_t s = static_cast<hipStream_t>(stream_);
    BESST_REQUIRE(steps > 0 && steps < 16, "linearize: steps is a mask of bits 0..3");
    BESST_REQUIRE(n_scaffolds >= 0 && n_scaffolds < ((int64_t)1 << 30), "linearize: scaffold count out of range");
    BESST_REQUIRE(n_edges >= 0 && n_edges < ((int64_t)1 << 31), "linearize: edge count out of range");
    BESST_REQUIRE(counters && (n_scaffolds == 0 || scaffold_removed_by), "linearize: null output pointer");
    BESST_REQUIRE(n_edges == 0 || (a && b && score && edge_alive), "linearize: null edge column");
    BESST_REQUIRE(n_scaffolds == 0 || (node_ambivalent && node_top && node_second && node_best_edge),
                  "linearize: null node output");
    const LinLayout L = lin_layout(n_scaffolds, n_edges);
    BESST_REQUIRE(workspace && workspace_bytes >= L.total, "linearize: workspace too small");
    char* w = static_cast<char*>(workspace);
    LinArgs g;
    g.a = a; g.b = b; g.score = score;
    g.steps = steps;
    g.m = (uint32_t)n_edges; g.n_scaf = (uint32_t)n_scaffolds; g.n_nodes = 2 * g.n_scaf;
    g.best_bits = (unsigned long long*)(w + L.best_bits);
    g.best_i = (uint32_t*)(w + L.best_i);
    g.deg = (uint32_t*)(w + L.deg);
    g.done_at = (int32_t*)(w + L.done_at);
    g.blocked_at = (int32_t*)(w + L.blocked_at);
    g.top1 = (unsigned long long*)(w + L.top1);
    g.top2 = (unsigned long long*)(w + L.top2);
    g.cnt = (uint32_t*)(w + L.cnt);
    g.ntop = (uint32_t*)(w + L.ntop);
    g.amb = (uint8_t*)(w + L.amb);
    g.mate = (int32_t*)(w + L.mate);
    g.jump[0] = (int32_t*)(w + L.jump0); g.jump[1] = (int32_t*)(w + L.jump1);
    g.lab[0] = (int32_t*)(w + L.lab0); g.lab[1] = (int32_t*)(w + L.lab1);
    g.alive = (uint8_t*)(w + L.alive);
    g.list[0] = (uint32_t*)(w + L.list0); g.list[1] = (uint32_t*)(w + L.list1);
    g.removed_by = (uint8_t*)(w + L.present);
    g.counters = (unsigned long long*)(w + L.counters);

    BESST_HIP_TRY(hipMemsetAsync(w + L.best_bits, 0, L.best_i - L.best_bits, s));
    BESST_HIP_TRY(hipMemsetAsync(w + L.best_i, 0xff, L.done_at - L.best_i, s));
    BESST_HIP_TRY(hipMemsetAsync(w + L.present, 0, (size_t)(n_scaffolds > 0 ? n_scaffolds : 1), s));
    const uint32_t nb_e = (g.m + kLinThreads - 1) / kLinThreads, nb_n = (g.n_nodes + kLinThreads - 1) / kLinThreads;
    const uint32_t nb_s = (g.n_scaf + kLinThreads - 1) / kLinThreads;
    // the counting kernels stride over their range with at most this many workgroups: one atomic each
    constexpr uint32_t kCountBlocks = 1024;
    const uint32_t nb_sc = nb_s < kCountBlocks ? nb_s : kCountBlocks, nb_nc = nb_n < kCountBlocks ? nb_n : kCountBlocks;
    unsigned long long host_counters[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int rounds = 0;
    if (g.n_scaf) {
        hipLaunchKernelGGL(lin_fill_kernel, dim3(nb_n), dim3(kLinThreads), 0, s, g.done_at, g.n_nodes, kNever);
        if (g.m) {
            hipLaunchKernelGGL(lin_best_kernel, dim3(nb_e), dim3(kLinThreads), 0, s, g);
            hipLaunchKernelGGL(lin_best_index_kernel, dim3(nb_e), dim3(kLinThreads), 0, s, g);
        }
        if (steps & 1) hipLaunchKernelGGL(lin_isolated_kernel, dim3(nb_sc), dim3(kLinThreads), 0, s, g, 0);
        // step 2: rounds until no live edge has an undecided endpoint (checked every kBatch rounds)
        constexpr int kBatch = 4;
        uint32_t bound = g.m;                          // upper bound of the pending list (lists only shrink)
        while (g.m && (steps & 2)) {
            const uint32_t nb_l = (bound + kLinThreads - 1) / kLinThreads;
            for (int r = 0; r < kBatch; ++r, ++rounds) {
                BESST_HIP_TRY(hipMemsetAsync(&g.counters[4 + ((rounds & 1) ^ 1)], 0, 8, s));
                hipLaunchKernelGGL(lin_block_kernel, dim3((nb_l + kBlockItems - 1) / kBlockItems), dim3(kLinThreads), 0, s, g,
                                   rounds);
                hipLaunchKernelGGL(lin_top_kernel, dim3(nb_l), dim3(kLinThreads), 0, s, g, rounds);
                hipLaunchKernelGGL(lin_second_kernel, dim3(nb_l), dim3(kLinThreads), 0, s, g, rounds);
                hipLaunchKernelGGL(lin_decide_kernel, dim3(nb_l), dim3(kLinThreads), 0, s, g, rounds);
            }
            // list of the batch's LAST round: the edges that still had an undecided endpoint when it began
            BESST_HIP_TRY(hipMemcpyAsync(host_counters, g.counters, sizeof(host_counters), hipMemcpyDeviceToHost, s));
            BESST_HIP_TRY(hipStreamSynchronize(s));
            bound = (uint32_t)host_counters[4 + (rounds & 1)];
            if (bound == 0) break;
            BESST_REQUIRE(rounds < (1 << 24), "linearize: step 2 did not converge");
        }
        if (steps & 2) hipLaunchKernelGGL(lin_count_amb_kernel, dim3(nb_nc), dim3(kLinThreads), 0, s, g);
        if (steps & 12) {                              // degrees over the surviving link edges
            BESST_HIP_TRY(hipMemsetAsync(g.deg, 0, (size_t)g.n_nodes * 4, s));
            if (g.m) hipLaunchKernelGGL(lin_degree_kernel, dim3(nb_e), dim3(kLinThreads), 0, s, g);
        }
        if (steps & 4) hipLaunchKernelGGL(lin_isolated_kernel, dim3(nb_sc), dim3(kLinThreads), 0, s, g, 1);
        if (steps & 8) {
            if (g.m) hipLaunchKernelGGL(lin_mate_kernel, dim3(nb_e), dim3(kLinThreads), 0, s, g);
            hipLaunchKernelGGL(lin_walk_init_kernel, dim3(nb_n), dim3(kLinThreads), 0, s, g);
            int from = 0, passes = 0;
            // walks longer than n_scaf steps are cycles: ceil(log2 n) + 1 passes at most, fewer when a pass (checked
            // every other one) changes nothing
            for (uint64_t reach = 1; reach < 2ull * g.n_scaf; reach <<= 1, ++passes) {
                if (passes & 1) {
                    BESST_HIP_TRY(hipMemsetAsync(&g.counters[7], 0, 8, s));
                    hipLaunchKernelGGL(lin_walk_double_count_kernel, dim3(nb_nc), dim3(kLinThreads), 0, s, g, from);
                    from ^= 1;
                    BESST_HIP_TRY(hipMemcpyAsync(host_counters, g.counters, sizeof(host_counters), hipMemcpyDeviceToHost, s));
                    BESST_HIP_TRY(hipStreamSynchronize(s));
                    if (host_counters[7] == 0) break;
                } else {
                    hipLaunchKernelGGL(lin_walk_double_kernel, dim3(nb_n), dim3(kLinThreads), 0, s, g, from);
                    from ^= 1;
                }
            }
            hipLaunchKernelGGL(lin_cycle_kernel, dim3(nb_nc), dim3(kLinThreads), 0, s, g, from);
        }
        BESST_HIP_TRY(hipGetLastError());
    }
    // results
    if (g.m) BESST_HIP_TRY(hipMemcpyAsync(edge_alive, g.alive, g.m, hipMemcpyDeviceToDevice, s));
    if (g.n_scaf) {
        BESST_HIP_TRY(hipMemcpyAsync(scaffold_removed_by, g.removed_by, g.n_scaf, hipMemcpyDeviceToDevice, s));
        BESST_HIP_TRY(hipMemcpyAsync(node_ambivalent, g.amb, g.n_nodes, hipMemcpyDeviceToDevice, s));
        BESST_HIP_TRY(hipMemcpyAsync(node_top, g.top1, (size_t)g.n_nodes * 8, hipMemcpyDeviceToDevice, s));
        BESST_HIP_TRY(hipMemcpyAsync(node_second, g.top2, (size_t)g.n_nodes * 8, hipMemcpyDeviceToDevice, s));
        BESST_HIP_TRY(hipMemcpyAsync(node_best_edge, g.best_i, (size_t)g.n_nodes * 4, hipMemcpyDeviceToDevice, s));
        BESST_HIP_TRY(hipMemcpyAsync(host_counters, g.counters, sizeof(host_counters), hipMemcpyDeviceToHost, s));
    }
    BESST_HIP_TRY(hipStreamSynchronize(s));
    counters[0] = (int64_t)host_counters[0];
    counters[1] = (int64_t)host_counters[1];
    counters[2] = (int64_t)(host_counters[2] / 2);
    counters[3] = (int64_t)host_counters[3];
    counters[4] = rounds;
    if (host_counters[6]) {
        set_error("linearize: step 4 without step 2 on a graph where %llu link edges meet a node that has several "
                  "(RemoveLoops expects the graph RemoveAmbiguousRegionsUsingScore leaves)", host_counters[6]);
        return BESST_ERR_STATE;
    }
    return BESST_OK;
}

int besst_linearize(int device, int32_t steps, int64_t n_scaffolds, int64_t n_edges, const int32_t* a, const int32_t* b,
                    const double* score, uint8_t* edge_alive, uint8_t* scaffold_removed_by, uint8_t* node_ambivalent,
                    double* node_top, double* node_second, uint32_t* node_best_edge, int64_t* counters) {
    BESST_REQUIRE(n_scaffolds >= 0 && n_scaffolds < ((int64_t)1 << 30), "linearize: scaffold count out of range");
    BESST_REQUIRE(n_edges >= 0 && n_edges < ((int64_t)1 << 31), "linearize: edge count out of range");
    BESST_HIP_TRY(hipSetDevice(device));
    const size_t m = (size_t)n_edges, n = (size_t)n_scaffolds * 2;
    const size_t ws = besst_dev_linearize_workspace_bytes(n_scaffolds, n_edges);
    // one allocation: workspace | a | b | score | outputs
    size_t off = 0;
    const size_t o_ws = carve(off, ws), o_a = carve(off, m * 4 + 4), o_b = carve(off, m * 4 + 4),
                 o_sc = carve(off, m * 8 + 8), o_alive = carve(off, m + 1), o_pres = carve(off, n / 2 + 1),
                 o_amb = carve(off, n + 1), o_top = carve(off, n * 8 + 8), o_sec = carve(off, n * 8 + 8),
                 o_best = carve(off, n * 4 + 4);
    char* d = nullptr;
    BESST_HIP_TRY(hipMalloc(&d, off));
    hipStream_t s = nullptr;
    int rc = BESST_OK;
    auto fail = [&](hipError_t e, const char* what) {
        set_error("linearize: %s failed: %s", what, hipGetErrorString(e));
        rc = BESST_ERR_HIP;
    };
    hipError_t e = hipStreamCreate(&s);
    if (e != hipSuccess) fail(e, "hipStreamCreate");
    if (rc == BESST_OK && m) {
        if ((e = hipMemcpyAsync(d + o_a, a, m * 4, hipMemcpyHostToDevice, s)) != hipSuccess) fail(e, "copy of a");
        else if ((e = hipMemcpyAsync(d + o_b, b, m * 4, hipMemcpyHostToDevice, s)) != hipSuccess) fail(e, "copy of b");
        else if ((e = hipMemcpyAsync(d + o_sc, score, m * 8, hipMemcpyHostToDevice, s)) != hipSuccess) fail(e, "copy of score");
    }
    if (rc == BESST_OK)
        rc = besst_dev_linearize(s, steps, n_scaffolds, n_edges, (const int32_t*)(d + o_a), (const int32_t*)(d + o_b),
                                 (const double*)(d + o_sc), d + o_ws, ws, (uint8_t*)(d + o_alive),
                                 (uint8_t*)(d + o_pres), (uint8_t*)(d + o_amb), (double*)(d + o_top),
                                 (double*)(d + o_sec), (uint32_t*)(d + o_best), counters);
    if (rc == BESST_OK) {
        if (m && (e = hipMemcpy(edge_alive, d + o_alive, m, hipMemcpyDeviceToHost)) != hipSuccess) fail(e, "copy of edge_alive");
        if (rc == BESST_OK && n) {
            if ((e = hipMemcpy(scaffold_removed_by, d + o_pres, n / 2, hipMemcpyDeviceToHost)) != hipSuccess) fail(e, "copy out");
            else if ((e = hipMemcpy(node_ambivalent, d + o_amb, n, hipMemcpyDeviceToHost)) != hipSuccess) fail(e, "copy out");
            else if ((e = hipMemcpy(node_top, d + o_top, n * 8, hipMemcpyDeviceToHost)) != hipSuccess) fail(e, "copy out");
            else if ((e = hipMemcpy(node_second, d + o_sec, n * 8, hipMemcpyDeviceToHost)) != hipSuccess) fail(e, "copy out");
            else if ((e = hipMemcpy(node_best_edge, d + o_best, n * 4, hipMemcpyDeviceToHost)) != hipSuccess) fail(e, "copy out");
        }
    }
    if (s) (void)hipStreamDestroy(s);
    (void)hipFree(d);
    return rc;
}

}  // extern "C"
