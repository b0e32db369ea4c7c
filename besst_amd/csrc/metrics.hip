// Library-statistics sampling on gfx950: the three `for read in bam_file` scans of libmetrics
// (libmetrics.py:63-84 contamination, :293-303 insert-size sample) as one ordered pass.
//
// Per record (15 B: tid, mtid, tlen 4 B each, flag 2, mapq 1):
//   A  insert-size observation    fr: is_proper_aligned_unique_innie, rf: ..._outie, and tid among the
//                                 1000 longest references                     [bam_parser.py:22-29, :294-301]
//   B  tid among the 1000 longest references (sample_counter)                 [libmetrics.py:65-66]
//   C  B and not unmapped (counter_total)                                     [:67-68]
//   D  B and opposite-orientation pair with read_len < fragment size          [:71-81]
// The reference stops collecting A after exactly 1,000,000 observations and stops the contamination
// scan at the record where sample_counter reaches 1,000,000 - both are "first N in stream order".
// Ordered semantics come from exclusive prefixes of A, B and D in stream order (per tile, once every tile has been counted:
// metrics_stage / tiles / place below): they decide inclusion and the output slot, so isize_out /
// contam_out hold |tlen| in BAM order exactly like the Python lists.  The float finishing (means, trimming, GetDistr) replays the
// reference's operation order on the host from those lists.
#include "common.h"

namespace besst {

namespace {

constexpr int kMetThreads = 256;
constexpr int kMetVec = 4;
constexpr int kMetTile = kMetThreads * kMetVec;   // 1024 records per block
constexpr long long kSampleCap = 1000000;

// state layout (int64): 0 nA seen, 1 nB seen, 2 nD seen, 3 counter_total (C within cut-off),
//                        4 n_contam (D within cut-off), 5 records scanned
struct Flags4 {
    uint32_t a, b, c, d;   // bit k = record k of the thread
    int32_t val[kMetVec];  // |tlen|
};

// the predicates of one record (bam_parser.py:22-29, libmetrics.py:63-84, 293-303); branch free: `top` only gates the bits
__device__ __forceinline__ void eval_record(const MetricsArgs& m, int k, int32_t tid, int32_t mtid, int32_t tlen, uint32_t flag,
                                            uint32_t mapq, bool top, Flags4& f) {
    const int64_t t64 = tlen;
    const int64_t at = t64 < 0 ? -t64 : t64;
    f.val[k] = (int32_t)at;
    const uint32_t bit = top ? 1u << k : 0u;
    f.b |= bit;
    if (!(flag & kFlagUnmapped)) f.c |= bit;
    const bool rev = flag & kFlagReverse, mrev = flag & kFlagMateReverse;
    const bool base = (flag & kFlagRead2) && tid == mtid && !(flag & kFlagMateUnmapped) && (int32_t)mapq > m.min_mapq &&
                      !(flag & kFlagSecondary);
    const bool innie = base && ((rev && !mrev && tlen < 0) || (!rev && mrev && tlen > 0));
    const bool outie = base && ((rev && !mrev && tlen > 0) || (!rev && mrev && tlen < 0));
    if (m.rf ? outie : innie) f.a |= bit;
    const bool d = m.rf ? (innie && m.read_len < (double)at) : (outie && m.read_len < (double)at + 2.0 * m.read_len);
    if (d) f.d |= bit;
}

// four records of a thread, any alignment and any end (the count kernel, and the one-pass kernel's ragged tiles).
// Every predicate asks for a record on one of the 1000 longest contigs (bam_parser.py:22-29 is only ever called under
// libmetrics.py:63,293's `sample in largest` test), so the reference id is read first and a WAVE none of whose records
// lies on such a contig - nearly every wave of a library on many contigs - reads nothing else.
__device__ __forceinline__ Flags4 eval4(const MetricsArgs& m, int64_t i0, int64_t end) {
    Flags4 f;
    f.a = f.b = f.c = f.d = 0;
    int32_t tid[kMetVec], mtid[kMetVec], tlen[kMetVec];
    uint32_t flag[kMetVec], mapq[kMetVec];
    const bool whole = i0 + kMetVec <= end && (i0 & 3) == 0;
    typedef int v4i __attribute__((ext_vector_type(4)));
    typedef unsigned int v2u __attribute__((ext_vector_type(2)));
    if (whole) {
        // (every record is read once: non-temporal, as in the record loop)
        const v4i v0 = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(m.tid + i0));
        tid[0] = v0.x; tid[1] = v0.y; tid[2] = v0.z; tid[3] = v0.w;
    } else {
#pragma unroll
        for (int k = 0; k < kMetVec; ++k) tid[k] = i0 + k < end ? m.tid[i0 + k] : -1;
    }
    bool top[kMetVec], any = false;
#pragma unroll
    for (int k = 0; k < kMetVec; ++k) {
        top[k] = (uint32_t)tid[k] < (uint32_t)m.n_contigs && m.top_mask[tid[k]] != 0;
        any = any || top[k];
    }
#pragma unroll
    for (int k = 0; k < kMetVec; ++k) f.val[k] = 0;
    if (__ballot(any) == 0ull) return f;                     // uniform
    if (whole) {
        const v4i v1 = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(m.mtid + i0));
        const v4i v2 = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(m.tlen + i0));
        const v2u fl = __builtin_nontemporal_load(reinterpret_cast<const v2u*>(m.flag + i0));
        const uint32_t mq = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(m.mapq + i0));
        mtid[0] = v1.x; mtid[1] = v1.y; mtid[2] = v1.z; mtid[3] = v1.w;
        tlen[0] = v2.x; tlen[1] = v2.y; tlen[2] = v2.z; tlen[3] = v2.w;
        flag[0] = fl.x & 0xffffu; flag[1] = fl.x >> 16; flag[2] = fl.y & 0xffffu; flag[3] = fl.y >> 16;
        mapq[0] = mq & 255u; mapq[1] = (mq >> 8) & 255u; mapq[2] = (mq >> 16) & 255u; mapq[3] = mq >> 24;
    } else {
#pragma unroll
        for (int k = 0; k < kMetVec; ++k) {
            const int64_t i = i0 + k;
            const bool in = i < end;
            mtid[k] = in ? m.mtid[i] : -2;
            tlen[k] = in ? m.tlen[i] : 0;
            flag[k] = in ? m.flag[i] : 0;
            mapq[k] = in ? m.mapq[i] : 0;
        }
    }
#pragma unroll
    for (int k = 0; k < kMetVec; ++k) eval_record(m, k, tid[k], mtid[k], tlen[k], flag[k], mapq[k], top[k], f);
    return f;
}

// A whole, aligned tile of the one-pass kernel: kSubs x four records per thread.  First the reference ids (every load of the
// tile issued before the first is waited for) and their top-1000 look-ups (a byte gather through the id, clamped instead of
// branched around); a wave without a record on such a contig is done - false, 4 of the 15 bytes of a record read, a seventh
// of the instructions -, the others read the four other columns (again all loads in flight together) and evaluate.
// (Written sub-tile by sub-tile the compiler kept each sub-tile's loads, and each of its four look-ups, behind the branches
// of the one before: 20 round trips in a row, 15 of a tile's 27 us.)
template <int kSubs>
__device__ __forceinline__ bool eval_tile(const MetricsArgs& m, int64_t i0, int64_t stride, Flags4 (&f)[kSubs]) {
    typedef int v4i __attribute__((ext_vector_type(4)));
    typedef unsigned int v2u __attribute__((ext_vector_type(2)));
    v4i tid[kSubs];
#pragma unroll
    for (int u = 0; u < kSubs; ++u) tid[u] = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(m.tid + i0 + (int64_t)u * stride));
    uint8_t top[kSubs][kMetVec];
#pragma unroll
    for (int u = 0; u < kSubs; ++u)
#pragma unroll
        for (int k = 0; k < kMetVec; ++k) {
            const uint32_t c = (uint32_t)tid[u][k];
            top[u][k] = m.top_mask[c < (uint32_t)m.n_contigs ? c : 0u];
        }
    bool any = false;
#pragma unroll
    for (int u = 0; u < kSubs; ++u) {
        f[u].a = f[u].b = f[u].c = f[u].d = 0;
#pragma unroll
        for (int k = 0; k < kMetVec; ++k) {
            f[u].val[k] = 0;
            any = any || ((uint32_t)tid[u][k] < (uint32_t)m.n_contigs && top[u][k] != 0);
        }
    }
    if (__ballot(any) == 0ull) return false;                 // uniform
    v4i mtid[kSubs], tlen[kSubs];
    v2u fl[kSubs];
    uint32_t mq[kSubs];
#pragma unroll
    for (int u = 0; u < kSubs; ++u) {
        const int64_t i = i0 + (int64_t)u * stride;
        mtid[u] = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(m.mtid + i));
        tlen[u] = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(m.tlen + i));
        fl[u] = __builtin_nontemporal_load(reinterpret_cast<const v2u*>(m.flag + i));
        mq[u] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(m.mapq + i));
    }
#pragma unroll
    for (int u = 0; u < kSubs; ++u) {
#pragma unroll
        for (int k = 0; k < kMetVec; ++k) {
            const uint32_t flag = (k & 1) ? fl[u][k >> 1] >> 16 : fl[u][k >> 1] & 0xffffu;
            const bool is_top = (uint32_t)tid[u][k] < (uint32_t)m.n_contigs && top[u][k] != 0;
            eval_record(m, k, tid[u][k], mtid[u][k], tlen[u][k], flag, (mq[u] >> (8 * k)) & 255u, is_top, f[u]);
        }
    }
    return true;
}

__device__ __forceinline__ int wsum(int v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

__global__ __launch_bounds__(kMetThreads) void metrics_count_kernel(MetricsArgs m, int64_t start, int64_t end,
                                                                    uint32_t* __restrict__ blk) {
    __shared__ int s[4][3];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t i0 = start + (int64_t)blockIdx.x * kMetTile + (int64_t)t * kMetVec;
    const Flags4 f = eval4(m, i0, end);
    const int a = wsum(__popc(f.a)), b = wsum(__popc(f.b)), d = wsum(__popc(f.d));
    if (lane == 0) { s[wave][0] = a; s[wave][1] = b; s[wave][2] = d; }
    __syncthreads();
    if (t < 3) blk[blockIdx.x * 3 + t] = (uint32_t)(s[0][t] + s[1][t] + s[2][t] + s[3][t]);
}

// exclusive scan of the three per-block counts; bases continue from the running state
__global__ __launch_bounds__(1024) void metrics_scan_kernel(uint32_t* __restrict__ blk, uint32_t nblocks,
                                                            const long long* __restrict__ state,
                                                            long long* __restrict__ base /* [nblocks*3] */,
                                                            long long* __restrict__ totals /* [3] */) {
    __shared__ long long s_w[16][3];
    __shared__ long long s_carry[3];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t < 3) s_carry[t] = state[t];
    __syncthreads();
    for (uint32_t c0 = 0; c0 < nblocks; c0 += 1024) {
        const uint32_t b = c0 + t;
        long long v[3], x[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) { v[j] = b < nblocks ? (long long)blk[b * 3 + j] : 0; x[j] = v[j]; }
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const long long o = __shfl_up(x[j], d, 64);
                if (lane >= d) x[j] += o;
            }
        }
        if (lane == 63) { s_w[wave][0] = x[0]; s_w[wave][1] = x[1]; s_w[wave][2] = x[2]; }
        __syncthreads();
        long long pre[3] = {s_carry[0], s_carry[1], s_carry[2]};
        for (int w = 0; w < wave; ++w) { pre[0] += s_w[w][0]; pre[1] += s_w[w][1]; pre[2] += s_w[w][2]; }
        if (b < nblocks) {
#pragma unroll
            for (int j = 0; j < 3; ++j) base[(size_t)b * 3 + j] = pre[j] + x[j] - v[j];
        }
        __syncthreads();
        if (t == 1023) { s_carry[0] = pre[0] + x[0]; s_carry[1] = pre[1] + x[1]; s_carry[2] = pre[2] + x[2]; }
        __syncthreads();
    }
    if (t < 3) totals[t] = s_carry[t];
}

// ---- ONE read of the records, no waiting between tiles (round 4).  Three forms came before: count + scan + emit read every
// record twice (0.11 of the pass's 22 B/pair roofline); a single kernel with a decoupled look-back over packed tile
// descriptors read them once but ran at 0.21 - 0.33: a tile cannot place its samples before every tile in front of it has
// its loads back, and with 1280 tiles in flight the slowest of those is always late (7 of a tile's 20 us were that wait, 1.5
// the arrival ticket; polling through the scalar data path changed nothing: the wait is for data, not for the poll).  So the
// samples are STAGED where no order is needed and placed when the order is known:
//   metrics_stage_kernel   a workgroup per tile of kMetSubs x 1024 records, any order: the flags of its sixteen records per
//                          thread (all loads in flight together), the tile's four counts, and the tile's A and D samples
//                          compacted in stream order at the TILE's place in the staging arrays;
//   metrics_tiles_kernel   one workgroup: exclusive scan of the tiles' counts behind the stream's running counts, the counts
//                          of the tiles that lie wholly inside the 1,000,000 cut-off of B, the one tile that straddles it;
//   metrics_place_kernel   a workgroup per tile: its staged samples copied to their places in the lists (nothing for a tile
//                          behind the cut-offs); the straddling tile evaluates its records again to find the cut.
// A call is cut into parts of kMetPart records (the staging arrays' size) that follow each other on the stream without
// the host; a tile of a part that begins with both lists full returns at once.
#ifndef BESST_MET_SUBS
#define BESST_MET_SUBS 4
#endif
constexpr int kMetSubs = BESST_MET_SUBS;
constexpr int kMetBig = kMetSubs * kMetTile;             // records per tile
constexpr int64_t kMetPart = (int64_t)64 << 20;          // records per part: 16 384 tiles, 2 x 256 MB of staging
static_assert(kMetPart % kMetBig == 0, "parts are whole tiles");

__device__ __forceinline__ long long wsum64(long long v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// The flags of a tile's records (whole aligned tiles: eval_tile, else the generic path) and, per sub-tile, the thread's counts
// a | b << 16 | d << 32 | c << 48 (`mine`), their inclusive scan over the wave (`incl`) and the waves' totals in s_w - what a
// thread needs to know its records' ranks inside the tile.  -> false (uniform over the wave): none of the wave's records
// counts anywhere (no scans, zeros in s_w).
__device__ __forceinline__ bool tile_flags(const MetricsArgs& m, int64_t tile0, int64_t end, int t, Flags4 (&f)[kMetSubs],
                                           unsigned long long (&mine)[kMetSubs], unsigned long long (&incl)[kMetSubs],
                                           unsigned long long (*s_w)[4], bool scan) {
    const int lane = t & 63, wave = t >> 6;
    bool any = true;
    if ((tile0 & 3) == 0 && tile0 + kMetBig <= end && m.n_contigs > 0) {           // uniform
        any = eval_tile<kMetSubs>(m, tile0 + (int64_t)t * kMetVec, kMetTile, f);
    } else {
        bool some = false;
#pragma unroll
        for (int u = 0; u < kMetSubs; ++u) {
            f[u] = eval4(m, tile0 + (int64_t)u * kMetTile + (int64_t)t * kMetVec, end);
            some = some || (f[u].b != 0u);                   // (every other bit is set together with b's)
        }
        any = __ballot(some) != 0ull;
    }
    if (!any) {                                              // uniform
#pragma unroll
        for (int u = 0; u < kMetSubs; ++u) {
            mine[u] = incl[u] = 0ull;
            if (lane == 63) s_w[u][wave] = 0ull;
        }
        return false;
    }
#pragma unroll
    for (int u = 0; u < kMetSubs; ++u) {
        mine[u] = (unsigned long long)__popc(f[u].a) | ((unsigned long long)__popc(f[u].b) << 16) |
                  ((unsigned long long)__popc(f[u].d) << 32) | ((unsigned long long)__popc(f[u].c) << 48);
        unsigned long long x = mine[u];
        if (scan) {                                          // uniform
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const unsigned long long o = __shfl_up(x, d, 64);
                if (lane >= d) x += o;
            }
        } else {
            x = (unsigned long long)wsum64((long long)x);    // (only the wave's total is needed)
        }
        incl[u] = x;
        if (lane == 63) s_w[u][wave] = x;
    }
    return true;
}

__global__ __launch_bounds__(kMetThreads) void metrics_stage_kernel(MetricsArgs m, int64_t start, int64_t end, int want_isize,
                                                                    const unsigned long long* __restrict__ state,
                                                                    uint32_t* __restrict__ tilecnt,
                                                                    int32_t* __restrict__ stage_a, int32_t* __restrict__ stage_d) {
    __shared__ unsigned long long s_w[kMetSubs][4];          // per sub-tile and wave: the four counts, 16 bits apart
    const int t = threadIdx.x, wave = t >> 6;
    const uint32_t tile = blockIdx.x;
    // what the stream had counted when this part began: a list that is full takes no more
    const bool live_a = want_isize && (long long)state[0] < kSampleCap;
    const bool live_b = (long long)state[1] < kSampleCap;
    if (!live_a && !live_b) {                                // uniform: nothing this tile could add
        if (t < 4) tilecnt[tile * 4u + (uint32_t)t] = 0u;
        return;
    }
    const int64_t tile0 = start + (int64_t)tile * kMetBig;
    Flags4 f[kMetSubs];
    unsigned long long mine[kMetSubs], incl[kMetSubs];
    const bool any = tile_flags(m, tile0, end, t, f, mine, incl, s_w, true);
    __syncthreads();
    if (!any && wave != 0) return;                           // uniform per wave: nothing to place (wave 0 still adds the tile up)
    unsigned long long run = 0;                              // the counts of the sub-tiles before the one at hand
    const size_t at = (size_t)tile * kMetBig;
#pragma unroll
    for (int u = 0; u < kMetSubs; ++u) {
        unsigned long long sub = 0, mywaves = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const unsigned long long v = s_w[u][w];
            sub += v;                                        // (fields of at most 1024 per sub-tile, 4096 per tile: no carries)
            if (w < wave) mywaves += v;
        }
        const unsigned long long ex = run + mywaves + incl[u] - mine[u];
        uint32_t ra = (uint32_t)(ex & 0xffffu), rd = (uint32_t)((ex >> 32) & 0xffffu);
        if (any) {                                           // uniform
#pragma unroll
            for (int k = 0; k < kMetVec; ++k) {
                const uint32_t bit = 1u << k;
                if (f[u].a & bit) {
                    if (live_a) stage_a[at + ra] = f[u].val[k];
                    ra++;
                }
                if (f[u].d & bit) {
                    if (live_b) stage_d[at + rd] = f[u].val[k];
                    rd++;
                }
            }
        }
        run += sub;
    }
    if (t < 4) tilecnt[tile * 4u + (uint32_t)t] = (uint32_t)((run >> (t == 0 ? 0 : t == 1 ? 16 : t == 2 ? 32 : 48)) & 0xffffu);   // a b d c
}

// info: [0] the tile in which B passes its cut-off (-1: none in this part).  Every WAVE takes a run of consecutive tiles, 64 at
// a time (coalesced): first the run's totals, one barrier, then the run again with the totals of the waves before it as
// carry.  (A loop of 1024 tiles per turn over the whole workgroup, two barriers and three 64-bit scans each: 52 us for a
// part's 16 384 tiles, a quarter of the pass; a run of tiles per THREAD, strided loads and stores: 45 us.)
__global__ __launch_bounds__(1024) void metrics_tiles_kernel(const uint32_t* __restrict__ tilecnt, uint32_t nt, long long records,
                                                             unsigned long long* state, long long* __restrict__ base /* [nt * 3] */,
                                                             int32_t* __restrict__ info) {
    __shared__ uint32_t s_w[16][3];
    __shared__ unsigned long long s_in_c, s_in_d;
    __shared__ int s_cut;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) { s_in_c = 0ull; s_in_d = 0ull; s_cut = -1; }
    const uint32_t per = (((nt + 15u) / 16u) + 63u) & ~63u;  // tiles per wave, whole rounds of 64
    const uint32_t lo = (uint32_t)wave * per < nt ? (uint32_t)wave * per : nt, hi = lo + per < nt ? lo + per : nt;
    const uint4* cnt4 = reinterpret_cast<const uint4*>(tilecnt);                  // a b d c
    uint32_t v[3] = {0u, 0u, 0u};                            // (a part holds at most 2^26 records: 32 bits)
    for (uint32_t b = lo + (uint32_t)lane; b < hi; b += 64u) {
        const uint4 q = cnt4[b];
        v[0] += q.x; v[1] += q.y; v[2] += q.z;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) v[j] = (uint32_t)wsum((int)v[j]);
    if (lane == 0) { s_w[wave][0] = v[0]; s_w[wave][1] = v[1]; s_w[wave][2] = v[2]; }
    __syncthreads();
    long long carry[3], total[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        uint32_t pre = 0, all = 0;
        for (int w = 0; w < 16; ++w) {
            if (w < wave) pre += s_w[w][j];
            all += s_w[w][j];
        }
        const long long before = (long long)state[j];
        carry[j] = before + (long long)pre;
        total[j] = before + (long long)all;
    }
    long long in_c = 0, in_d = 0;
    for (uint32_t r0 = lo; r0 < hi; r0 += 64u) {             // uniform per wave
        const uint32_t b = r0 + (uint32_t)lane;
        uint4 q = make_uint4(0u, 0u, 0u, 0u);
        if (b < hi) q = cnt4[b];
        uint32_t x[3] = {q.x, q.y, q.z};
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const uint32_t o = (uint32_t)__shfl_up((int)x[j], d, 64);
                if (lane >= d) x[j] += o;
            }
        }
        const long long ex1 = carry[1] + (long long)(x[1] - q.y);
        if (b < hi) {
            base[(size_t)b * 3] = carry[0] + (long long)(x[0] - q.x);
            base[(size_t)b * 3 + 1] = ex1;
            base[(size_t)b * 3 + 2] = carry[2] + (long long)(x[2] - q.z);
            if (ex1 + (long long)q.y <= kSampleCap) { in_c += q.w; in_d += q.z; }     // wholly inside the cut-off of B
            else if (ex1 < kSampleCap) s_cut = (int)b;                                // (one tile at most)
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) carry[j] += (long long)(uint32_t)__builtin_amdgcn_readlane((int)x[j], 63);
    }
    in_c = wsum64(in_c);
    in_d = wsum64(in_d);
    if (lane == 0) { atomicAdd(&s_in_c, (unsigned long long)in_c); atomicAdd(&s_in_d, (unsigned long long)in_d); }
    __syncthreads();                                         // (every thread has read state[0..2] by now)
    if (t < 3) state[t] = (unsigned long long)total[t];
    if (t == 3) state[3] += s_in_c;
    if (t == 4) state[4] += s_in_d;
    if (t == 5) state[5] += (unsigned long long)records;
    if (t == 6) info[0] = s_cut;
}

__global__ __launch_bounds__(kMetThreads) void metrics_place_kernel(MetricsArgs m, int64_t start, int64_t end, int want_isize,
                                                                    const uint32_t* __restrict__ tilecnt,
                                                                    const long long* __restrict__ base,
                                                                    const int32_t* __restrict__ info,
                                                                    const int32_t* __restrict__ stage_a,
                                                                    const int32_t* __restrict__ stage_d,
                                                                    int32_t* __restrict__ isize_out, int32_t* __restrict__ contam_out,
                                                                    unsigned long long* state) {
    __shared__ unsigned long long s_w[kMetSubs][4];
    __shared__ int s_tot[4][2];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t tile = blockIdx.x;
    const uint32_t na = tilecnt[tile * 4u], nb = tilecnt[tile * 4u + 1u], nd = tilecnt[tile * 4u + 2u];
    const long long pa = base[(size_t)tile * 3], pb = base[(size_t)tile * 3 + 1], pd = base[(size_t)tile * 3 + 2];
    const size_t at = (size_t)tile * kMetBig;
    if (want_isize && pa < kSampleCap) {
        const long long room = kSampleCap - pa;
        const uint32_t n = (long long)na < room ? na : (uint32_t)room;
        for (uint32_t i = (uint32_t)t; i < n; i += kMetThreads) isize_out[pa + i] = stage_a[at + i];
    }
    if (pb + (long long)nb <= kSampleCap) {                  // wholly inside the cut-off of B: every D sample counts
        for (uint32_t i = (uint32_t)t; i < nd; i += kMetThreads) contam_out[pd + i] = stage_d[at + i];
        return;
    }
    if ((int)tile != info[0]) return;                        // behind the cut-off
    // the tile in which B reaches 1,000,000: which of its records come before that is found by looking at them again
    const int64_t tile0 = start + (int64_t)tile * kMetBig;
    Flags4 f[kMetSubs];
    unsigned long long mine[kMetSubs], incl[kMetSubs];
    tile_flags(m, tile0, end, t, f, mine, incl, s_w, true);
    __syncthreads();
    unsigned long long run = 0;
    int n_c = 0, n_d = 0;
#pragma unroll
    for (int u = 0; u < kMetSubs; ++u) {
        unsigned long long sub = 0, mywaves = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const unsigned long long v = s_w[u][w];
            sub += v;
            if (w < wave) mywaves += v;
        }
        const unsigned long long ex = run + mywaves + incl[u] - mine[u];
        long long rb = pb + (long long)((ex >> 16) & 0xffffu), rd = pd + (long long)((ex >> 32) & 0xffffu);
#pragma unroll
        for (int k = 0; k < kMetVec; ++k) {
            const uint32_t bit = 1u << k;
            if (f[u].b & bit) {
                const bool in = rb < kSampleCap;             // this record is among the first 1,000,000 on the top contigs
                rb++;
                if (in) {
                    if (f[u].c & bit) n_c++;
                    if (f[u].d & bit) { contam_out[rd] = f[u].val[k]; n_d++; }
                }
            }
            if (f[u].d & bit) rd++;
        }
        run += sub;
    }
    n_c = wsum(n_c);
    n_d = wsum(n_d);
    if (lane == 0) { s_tot[wave][0] = n_c; s_tot[wave][1] = n_d; }
    __syncthreads();
    if (t < 2) {
        const int v = s_tot[0][t] + s_tot[1][t] + s_tot[2][t] + s_tot[3][t];
        if (v) atomicAdd(&state[3 + t], (unsigned long long)v);
    }
}

// the count-only form's last step: the running counts take the scan's totals
__global__ void metrics_commit_kernel(long long* __restrict__ state, const long long* __restrict__ totals,
                                      long long scanned) {
    if (threadIdx.x < 3) state[threadIdx.x] = totals[threadIdx.x];
    if (threadIdx.x == 3) state[5] += scanned;
}

__global__ __launch_bounds__(256) void value_hist_kernel(const int32_t* __restrict__ values, int64_t n,
                                                         int64_t n_bins, unsigned long long* __restrict__ hist,
                                                         unsigned long long* __restrict__ overflow) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned long long over = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int32_t v = values[i];
        if (v >= 0 && v < n_bins) atomicAdd(&hist[v], 1ull);
        else over++;
    }
    if (over) atomicAdd(overflow, over);
}

}  // namespace

namespace {
struct MetWs { size_t tilecnt, base, info, stage_a, stage_d, total; };
MetWs metrics_staged_layout(int64_t n) {
    const int64_t part = n < kMetPart ? n : kMetPart;
    const size_t nt = (size_t)((part + kMetBig - 1) / kMetBig);
    MetWs w;
    size_t off = 0;
    w.tilecnt = off; off += align_up(nt * 16, 256);
    w.base = off; off += align_up(nt * 24, 256);
    w.info = off; off += 256;
    w.stage_a = off; off += align_up(nt * (size_t)kMetBig * 4, 256);
    w.stage_d = off; off += align_up(nt * (size_t)kMetBig * 4, 256);
    w.total = off;
    return w;
}
}  // namespace

size_t metrics_workspace_bytes(int64_t n) {
    const size_t nb = (size_t)((n + kMetTile - 1) / kMetTile) + 1;
    const size_t counting = align_up(nb * 3 * 4, 256) + align_up(nb * 3 * 8, 256) + 256;        // the count-only form
    const size_t staged = metrics_staged_layout(n < 1 ? 1 : n).total;
    return counting > staged ? counting : staged;
}

// Scan records [start, start+count).  state: 6 x int64 on the device (see layout above); state[0..2] are exact up to the
// cut-offs and at least the cut-off beyond them (a part that begins with both lists full is not looked at).
// count_only: only the three running counts (state[0..2]) advance - the first phase of a sharded scan, whose
// slices need the counts of the slices before them to place their samples.
int launch_metrics(hipStream_t s, const MetricsArgs& a, int64_t start, int64_t count, int32_t* isize_out,
                   int32_t* contam_out, int64_t* state, void* ws, size_t ws_bytes, bool count_only) {
    if (count <= 0) return BESST_OK;
    BESST_REQUIRE((start & 3) == 0, "metrics: chunk start must be a multiple of 4");
    BESST_REQUIRE(ws && ws_bytes >= metrics_workspace_bytes(count), "metrics: workspace too small");
    char* p = static_cast<char*>(ws);
    const int64_t end = start + count;
    ProfScope ps(s, kProfMetrics);
    if (!count_only) {
        const MetWs w = metrics_staged_layout(count);
        uint32_t* tilecnt = reinterpret_cast<uint32_t*>(p + w.tilecnt);
        long long* base = reinterpret_cast<long long*>(p + w.base);
        int32_t* info = reinterpret_cast<int32_t*>(p + w.info);
        int32_t* stage_a = reinterpret_cast<int32_t*>(p + w.stage_a);
        int32_t* stage_d = reinterpret_cast<int32_t*>(p + w.stage_d);
        unsigned long long* st = reinterpret_cast<unsigned long long*>(state);
        const int want_isize = isize_out != nullptr ? 1 : 0;
        for (int64_t at = start; at < end; at += kMetPart) {
            const int64_t part_end = at + kMetPart < end ? at + kMetPart : end;
            const uint32_t nt = (uint32_t)((part_end - at + kMetBig - 1) / kMetBig);
            hipLaunchKernelGGL(metrics_stage_kernel, dim3(nt), dim3(kMetThreads), 0, s, a, at, part_end, want_isize, st, tilecnt,
                               stage_a, stage_d);
            hipLaunchKernelGGL(metrics_tiles_kernel, dim3(1), dim3(1024), 0, s, tilecnt, nt, (long long)(part_end - at), st, base, info);
            hipLaunchKernelGGL(metrics_place_kernel, dim3(nt), dim3(kMetThreads), 0, s, a, at, part_end, want_isize, tilecnt, base,
                               info, stage_a, stage_d, isize_out, contam_out, st);
        }
        BESST_HIP_TRY(hipGetLastError());
        return BESST_OK;
    }
    // count only (the first phase of a sharded scan): one read, the three totals
    const uint32_t nb = (uint32_t)((count + kMetTile - 1) / kMetTile);
    uint32_t* blk = reinterpret_cast<uint32_t*>(p);
    long long* base = reinterpret_cast<long long*>(p + align_up((size_t)(nb + 1) * 3 * 4, 256));
    long long* totals = reinterpret_cast<long long*>(p + align_up((size_t)(nb + 1) * 3 * 4, 256) +
                                                     align_up((size_t)(nb + 1) * 3 * 8, 256));
    hipLaunchKernelGGL(metrics_count_kernel, dim3(nb), dim3(kMetThreads), 0, s, a, start, end, blk);
    hipLaunchKernelGGL(metrics_scan_kernel, dim3(1), dim3(1024), 0, s, blk, nb,
                       reinterpret_cast<const long long*>(state), base, totals);
    hipLaunchKernelGGL(metrics_commit_kernel, dim3(1), dim3(64), 0, s, reinterpret_cast<long long*>(state), totals, 0ll);
    BESST_HIP_TRY(hipGetLastError());
    return BESST_OK;
}

// ---- is the stream sorted by coordinate? ---------------------------------------------------------------------------------
// The reference refuses a BAM without an index (libmetrics.py:237-241: bam_file.fetch() raises), and an index exists only
// for a coordinate-sorted file.  The record loop here is exact on any order but several times slower on an unsorted one
// (its run naming relies on neighbouring records sharing contigs), so get_metrics says so: this pass compares every record's
// (reference id, position) with its predecessor's - reference -1 (unplaced reads) sorts last, as samtools sort leaves it -
// and reports the first record that lies in front of its predecessor.  8 bytes per record, one launch.
__global__ __launch_bounds__(256) void stream_order_kernel(const int32_t* __restrict__ tid, const int32_t* __restrict__ pos, long long n,
                                                           unsigned long long* __restrict__ first_unsorted) {
    const long long stride = (long long)gridDim.x * 256 * 4;
    unsigned long long worst = ~0ull;
    for (long long base = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; base < n; base += stride) {
        // four records per thread: aligned 16-byte loads, plus the record in front of them
        int32_t t[5], p[5];
        if (base + 4 <= n) {
            const int4 tv = *reinterpret_cast<const int4*>(tid + base), pv = *reinterpret_cast<const int4*>(pos + base);
            t[1] = tv.x; t[2] = tv.y; t[3] = tv.z; t[4] = tv.w;
            p[1] = pv.x; p[2] = pv.y; p[3] = pv.z; p[4] = pv.w;
        } else {
            for (int k = 0; k < 4; ++k) {
                const bool in = base + k < n;
                t[k + 1] = in ? tid[base + k] : -1;
                p[k + 1] = in ? pos[base + k] : 0x7fffffff;
            }
        }
        t[0] = base > 0 ? tid[base - 1] : (int32_t)0x80000000;   // (nothing in front of the first record: the smallest key)
        p[0] = base > 0 ? pos[base - 1] : (int32_t)0x80000000;
#pragma unroll
        for (int k = 1; k < 5; ++k) {
            // key: reference id as unsigned (so that -1 is the largest), then the position + 1 (-1 for a read without one)
            const unsigned long long a = ((unsigned long long)(uint32_t)t[k - 1] << 32) | ((uint32_t)p[k - 1] + 1u);
            const unsigned long long b = ((unsigned long long)(uint32_t)t[k] << 32) | ((uint32_t)p[k] + 1u);
            if (k - 1 == 0 && base == 0) continue;
            if (base + k - 1 < n && b < a) {
                const unsigned long long at = (unsigned long long)(base + k - 1);
                worst = at < worst ? at : worst;
            }
        }
    }
    if (worst != ~0ull) atomicMin(first_unsorted, worst);
}

int launch_stream_order(hipStream_t s, const int32_t* tid, const int32_t* pos, int64_t n, unsigned long long* first_unsorted) {
    if (n <= 1) return BESST_OK;
    const int64_t want = (n + 1023) / 1024;
    const int blocks = (int)(want < 8192 ? want : 8192);
    hipLaunchKernelGGL(stream_order_kernel, dim3(blocks), dim3(256), 0, s, tid, pos, (long long)n, first_unsorted);
    BESST_HIP_TRY(hipGetLastError());
    return BESST_OK;
}

int launch_value_histogram(hipStream_t s, const int32_t* values, int64_t n, int64_t n_bins,
                           unsigned long long* hist, unsigned long long* overflow) {
    if (n <= 0) return BESST_OK;
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(value_hist_kernel, dim3(blocks), dim3(256), 0, s, values, n, n_bins, hist, overflow);
    BESST_HIP_TRY(hipGetLastError());
    return BESST_OK;
}

}  // namespace besst
