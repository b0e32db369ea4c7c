// Stage 1 of the scaffold-graph build: the per-record body of CreateGraph.PE on gfx950.
//
// One pass over the SoA record columns (19 B/record: tid mtid pos mpos 4 B each, flag 2, mapq 1,
// qlen 2) does, per record (reference lines in brackets):
//   * contig membership and class lookup          [CreateGraph.py:118-130]   one 16-B gather
//   * coverage numerator += qlen                   [:138-139]                 wave-reduced atomics
//   * fishy (BWA flag quirk) tuple                 [:141-163, CheckDir :678-688]
//   * non_unique tally                             [:166-167]
//   * link dispatch case A / case B                [:169-206]
//   * PosDirCalculatorPE / MP                      [:1024-1076]               fp64, no contraction
//   * CreateEdge duplicate chain + acceptance      [:812-871]
// and emits accepted link tuples and fishy tuples IN STREAM ORDER.
//
// Order-dependent semantics: a record is a duplicate iff its (obs1, obs2) equals that of the previous
// record that reached CreateEdge, anywhere earlier in the stream.  Inside a workgroup the chain is
// resolved with a wave ballot + LDS hand-off; across workgroups each block publishes a 32-byte
// summary (first/last reaching observation, head record) and a single-workgroup "stitch" kernel
// resolves every block's head against its predecessor, fixes the counters, and scans the per-block
// tuple counts so that a third kernel can compact the block-local segments into one ordered stream.
// No inter-workgroup communication happens inside a launch.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include "common.h"

namespace besst {

namespace {

constexpr uint32_t kNoSlot = 0xffffffffu;

// evaluation bits of one record
constexpr uint32_t EV_COV = 1u, EV_FISHY = 2u, EV_NONUNIQ = 4u, EV_REACH = 8u,
                   EV_DOUBLE = 32u, EV_MAPQ0 = 64u, EV_CASEA = 128u, EV_FIRSTMIN = 256u;

struct Eval {
    uint32_t bits;
    int32_t o1, o2;
    uint32_t n_min, n_max;   // node codes (scaffold * 2 + side) of the edge this record may support
};

struct CEDelta {
    int count, too_long, dup, nus;
    bool keep;
};

// CreateEdge call sequence for one record (CreateGraph.py:176-183,812-871): first call against the
// running prev_obs, optional second call (G_prime) with prev_obs reset to (-1,-1).
__device__ __forceinline__ CEDelta create_edge(int o1, int o2, int p1, int p2, bool accept, bool dbl,
                                               bool mapq0, bool detect) {
    CEDelta d{0, 0, 0, 0, false};
    d.nus += mapq0 ? 1 : 0;
    if (o1 == p1 && o2 == p2) {
        d.dup++;
        if (detect) return d;
    }
    if (accept) {
        d.count++;
        d.keep = true;
    } else {
        d.too_long++;
    }
    if (dbl) {
        d.nus += mapq0 ? 1 : 0;
        if (o1 == -1 && o2 == -1) {
            d.dup++;
            if (detect) return d;
        }
        if (accept) d.count++; else d.too_long++;
    }
    return d;
}

// One end of PosDirCalculatorPE / PosDirCalculatorMP (CreateGraph.py:1024-1076).  The MP calculator is
// the PE one with the read strand inverted.  read_len may be fractional: Python evaluates
// `cpos + rpos + read_len` and `slen - cpos - (clen - rpos - read_len)` left to right in float and
// truncates with int(); the same two roundings are done here in fp64 (built with -ffp-contract=off).
__device__ __forceinline__ void posdir(bool rf, bool cdir, bool rdir, int32_t cpos, int32_t rpos,
                                       int32_t slen, int32_t clen, double read_len, int32_t& obs,
                                       uint32_t& side) {
    if (rf) rdir = !rdir;
    if (rdir) {
        if (cdir) {
            obs = (int32_t)((int64_t)slen - cpos - rpos);
            side = 1;
        } else {
            obs = (int32_t)((int64_t)cpos + ((int64_t)clen - rpos));
            side = 0;
        }
    } else {
        if (cdir) {
            double v = (double)((int64_t)cpos + rpos) + read_len;
            obs = (int32_t)v;
            side = 0;
        } else {
            double inner = (double)((int64_t)clen - rpos) - read_len;
            double v = (double)((int64_t)slen - cpos) - inner;
            obs = (int32_t)v;
            side = 1;
        }
    }
}

// Evaluate one record given its two contig rows (gathered by the caller so that several records' gathers
// can be in flight together); in_range = both tids are valid header indexes.
__device__ __forceinline__ Eval eval_record(const ClassifyArgs& a, bool in_range, const ContigRow& c1,
                                            const ContigRow& c2, int32_t tid, int32_t mtid, int32_t pos,
                                            int32_t mpos, uint32_t flag, uint32_t mapq) {
    Eval e;
    e.bits = 0;
    e.o1 = e.o2 = 0;
    e.n_min = e.n_max = 0;
    if (!in_range) return e;
    const uint32_t cls1 = c1.w0 >> 29, cls2 = c2.w0 >> 29;
    if (cls1 == BESST_CLS_ABSENT || cls2 == BESST_CLS_ABSENT) return e;
    const uint32_t scaf1 = c1.w0 & kScafIdMask, scaf2 = c2.w0 & kScafIdMask;
    const bool dir1 = (c1.w0 >> 28) & 1u, dir2 = (c2.w0 >> 28) & 1u;
    const bool rdir = !(flag & kFlagReverse), mdir = !(flag & kFlagMateReverse);
    const bool rf = a.rf != 0;

    if ((int32_t)mapq >= a.min_mapq || mapq == 0) e.bits |= EV_COV;
    if (mapq == 0) e.bits |= EV_MAPQ0;
    const bool other = tid != mtid;
    if (other && mapq == 0) e.bits |= EV_NONUNIQ;

    if ((flag & kFlagUnmapped) && (flag & kFlagRead1) && scaf1 != scaf2) {
        // CheckDir: PosDir with zeroed coordinates, only the sides matter
        const uint32_t s1 = (dir1 == (rf ? !rdir : rdir)) ? 1u : 0u;
        const uint32_t s2 = (dir2 == (rf ? !mdir : mdir)) ? 1u : 0u;
        const uint32_t n1 = scaf1 * 2 + s1, n2 = scaf2 * 2 + s2;
        e.n_min = n1 < n2 ? n1 : n2;
        e.n_max = n1 < n2 ? n2 : n1;
        e.bits |= EV_FISHY;
        return e;   // an unmapped record cannot also be a link candidate (:169)
    }

    if (!(other && (flag & kFlagRead2) && !(flag & kFlagUnmapped) && (int32_t)mapq >= a.min_mapq))
        return e;
    bool case_a = false;
    if (cls1 == BESST_CLS_LARGE && cls2 == BESST_CLS_LARGE && scaf1 != scaf2) {
        case_a = true;                                           // case A (:170-183)
    } else if (a.extend_paths) {                                 // case B (:184-206)
        const bool sm1 = cls1 == BESST_CLS_SMALL, sm2 = cls2 == BESST_CLS_SMALL;
        if (!((sm1 && sm2 && scaf1 != scaf2) || (sm1 != sm2))) return e;
    } else {
        return e;
    }
    uint32_t s1, s2;
    posdir(rf, dir1, rdir, c1.ctg_pos, pos, c1.scaf_len, c1.ctg_len, a.read_len, e.o1, s1);
    posdir(rf, dir2, mdir, c2.ctg_pos, mpos, c2.scaf_len, c2.ctg_len, a.read_len, e.o2, s2);
    const bool dbl = case_a && a.extend_paths && !a.no_score;    // second CreateEdge call for G_prime
    e.bits |= EV_REACH | (case_a ? EV_CASEA : 0u) | (dbl ? EV_DOUBLE : 0u);
    const uint32_t n1 = scaf1 * 2 + s1, n2 = scaf2 * 2 + s2;
    e.n_min = n1 < n2 ? n1 : n2;
    e.n_max = n1 < n2 ? n2 : n1;
    if (n1 < n2) e.bits |= EV_FIRSTMIN;
    return e;
}

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// ---------------------------------------------------------------------------------------------------------
// stream_kernel: the bandwidth-bound pass.  Per record it needs only tid, mtid, mapq, qlen (11 B):
//   * tid == mtid (98 % of a real stream): the record can only contribute coverage            [:138-139]
//   * tid != mtid: "candidate" - everything else in the loop body requires contig1 != contig2 or
//     different scaffolds (:141-169); its bit is published and candidate_kernel does the rest,
//     including the candidate's own coverage contribution.
// The stream is (tid,pos)-sorted, so the 256 records of a wave normally share one tid: coverage is a wave
// reduction accumulated across the workgroup's sub-tiles and flushed with one 64-bit atomic per run.
// ---------------------------------------------------------------------------------------------------------
// Add `val` to dst[key] for every active lane with ONE atomic per distinct key of the wave.  The stream is
// (tid,pos)-sorted, so a wave sees one or two distinct contigs; per-lane atomics on the same two addresses
// were measured at ~0.1 ns each chip-wide and dominated the pass.  After 8 distinct keys the remaining lanes
// fall back to their own atomics (unsorted input stays correct, only slower).
__device__ __forceinline__ void wave_add_by_key(unsigned long long* dst, int32_t key, int val, bool active,
                                                int lane) {
    unsigned long long mask = __ballot(active);
    int iter = 0;
    while (mask) {
        if (iter++ == 8) {
            if (active) atomicAdd(&dst[key], (unsigned long long)val);
            break;
        }
        const int leader = __ffsll((long long)mask) - 1;
        const int32_t k = __shfl(key, leader, 64);
        const bool match = active && key == k;
        const int tot = wave_sum(match ? val : 0);
        if (lane == leader && tot) atomicAdd(&dst[k], (unsigned long long)tot);
        active = active && !match;
        mask = __ballot(active);
    }
}

__device__ __forceinline__ void flush_cov(const ClassifyArgs& a, unsigned long long* aligned, int lane,
                                          int32_t ref, int sum) {
    // No class lookup here (it would be a dependent memory round trip at the end of every wave): any in-range
    // tid is credited and compact_kernel clears the entries of contigs that are not in the table.
    if (lane == 0 && sum && (uint32_t)ref < (uint32_t)a.n_contigs)
        atomicAdd(&aligned[ref], (unsigned long long)sum);
}

__device__ __forceinline__ void eval_group(const ClassifyArgs& a, int64_t g, unsigned long long b0,
                                           unsigned long long b1, unsigned long long b2, unsigned long long b3,
                                           unsigned long long* __restrict__ aligned, uint4* __restrict__ staging,
                                           int lane);

template <bool kFuseEval>
__global__ __launch_bounds__(kStreamThreads) void stream_kernel(ClassifyArgs a,
                                                                unsigned long long* __restrict__ aligned,
                                                                unsigned long long* __restrict__ bitmask,
                                                                uint4* __restrict__ staging) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t block_base = (int64_t)blockIdx.x * kStreamTile;
    int4 v_tid[kStreamSubTiles], v_mtid[kStreamSubTiles];
    uchar4 v_mapq[kStreamSubTiles];
    ushort4 v_qlen[kStreamSubTiles];
    if (block_base + kStreamTile <= a.n) {
        // full tile: issue every load of the workgroup's 4096 records before touching any of them
#pragma unroll
        for (int st = 0; st < kStreamSubTiles; ++st) {
            const int64_t i0 = block_base + (int64_t)st * kStreamSubTile + (int64_t)t * kStreamVec;
#ifdef BESST_NT
            v_tid[st] = __builtin_nontemporal_load(reinterpret_cast<const int4*>(a.tid + i0));
            v_mtid[st] = __builtin_nontemporal_load(reinterpret_cast<const int4*>(a.mtid + i0));
            v_mapq[st] = __builtin_nontemporal_load(reinterpret_cast<const uchar4*>(a.mapq + i0));
            v_qlen[st] = __builtin_nontemporal_load(reinterpret_cast<const ushort4*>(a.qlen + i0));
#else
            v_tid[st] = *reinterpret_cast<const int4*>(a.tid + i0);
            v_mtid[st] = *reinterpret_cast<const int4*>(a.mtid + i0);
            v_mapq[st] = *reinterpret_cast<const uchar4*>(a.mapq + i0);
            v_qlen[st] = *reinterpret_cast<const ushort4*>(a.qlen + i0);
#endif
        }
    } else {
#pragma unroll
        for (int st = 0; st < kStreamSubTiles; ++st) {
            const int64_t i0 = block_base + (int64_t)st * kStreamSubTile + (int64_t)t * kStreamVec;
            int32_t x[4], y[4];
            uint32_t m[4], q[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int64_t i = i0 + k;
                const bool in = i < a.n;
                x[k] = in ? a.tid[i] : -1;
                y[k] = in ? a.mtid[i] : -1;
                m[k] = in ? a.mapq[i] : 0;
                q[k] = in ? a.qlen[i] : 0;
            }
            v_tid[st] = make_int4(x[0], x[1], x[2], x[3]);
            v_mtid[st] = make_int4(y[0], y[1], y[2], y[3]);
            v_mapq[st] = make_uchar4((unsigned char)m[0], (unsigned char)m[1], (unsigned char)m[2], (unsigned char)m[3]);
            v_qlen[st] = make_ushort4((unsigned short)q[0], (unsigned short)q[1], (unsigned short)q[2], (unsigned short)q[3]);
        }
    }
    int32_t acc_ref = -1;
    int acc_sum = 0;
#pragma unroll
    for (int st = 0; st < kStreamSubTiles; ++st) {
        const int32_t r_tid[4] = {v_tid[st].x, v_tid[st].y, v_tid[st].z, v_tid[st].w};
        const int32_t r_mtid[4] = {v_mtid[st].x, v_mtid[st].y, v_mtid[st].z, v_mtid[st].w};
        const uint32_t r_mapq[4] = {v_mapq[st].x, v_mapq[st].y, v_mapq[st].z, v_mapq[st].w};
        const uint32_t r_qlen[4] = {v_qlen[st].x, v_qlen[st].y, v_qlen[st].z, v_qlen[st].w};
        const int32_t ref = __builtin_amdgcn_readfirstlane(r_tid[0]);
        bool uni = true;
        int mine = 0;
        bool cand[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            cand[k] = r_tid[k] != r_mtid[k];
            uni = uni && (r_tid[k] == ref);
            const bool cov = ((int32_t)r_mapq[k] >= a.min_mapq || r_mapq[k] == 0);
            if (!cand[k] && cov) mine += (int)r_qlen[k];
        }
        if (__all(uni)) {
            const int tot = wave_sum(mine);
            if (ref != acc_ref) {
                flush_cov(a, aligned, lane, acc_ref, acc_sum);
                acc_ref = ref;
                acc_sum = 0;
            }
            acc_sum += tot;
        } else {
            // a contig boundary (or unsorted input) inside the wave: one atomic per distinct contig and slot
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool cov = ((int32_t)r_mapq[k] >= a.min_mapq || r_mapq[k] == 0);
                const bool act = !cand[k] && cov && (uint32_t)r_tid[k] < (uint32_t)a.n_contigs;
                wave_add_by_key(aligned, r_tid[k], (int)r_qlen[k], act, lane);
            }
        }
        // candidate bits of this wave's group: word k, bit l  <->  record group_base + 4*l + k
        const unsigned long long b0 = __ballot(cand[0]), b1 = __ballot(cand[1]);
        const unsigned long long b2 = __ballot(cand[2]), b3 = __ballot(cand[3]);
        const int64_t g = ((int64_t)blockIdx.x * kStreamSubTiles + st) * 4 + wave;
        if (lane < 4) bitmask[g * 4 + lane] = lane == 0 ? b0 : lane == 1 ? b1 : lane == 2 ? b2 : b3;
        if (kFuseEval && (b0 | b1 | b2 | b3) != 0ull)      // dense libraries: evaluate while the lines are hot
            eval_group(a, g, b0, b1, b2, b3, aligned, staging, lane);
    }
    flush_cov(a, aligned, lane, acc_ref, acc_sum);
}

// ---------------------------------------------------------------------------------------------------------
// eval_kernel: evaluate every candidate, balanced and order-free.  One wave per 256-record group, in the same
// record <-> lane layout as stream_kernel (lane l owns records 4l..4l+3), so the record fields are fetched
// with the same coalesced vector loads (only by lanes that own a candidate) and a lane evaluates at most four
// records.  Waves whose group has no candidate exit after one broadcast load of their 32 bytes of bits.  The
// evaluated candidates are written as 16-byte entries, compacted in record order inside the group's staging
// slice (group g -> staging[g*256 ...]); candidates' coverage is added with one atomic per distinct contig.
//   entry = { obs1, obs2, node_min | REACH<<29 | FISHY<<30 | NONUNIQ<<31,
//                         node_max | MAPQ0<<29 | CASEA<<30 | FIRSTMIN<<31 }
// ---------------------------------------------------------------------------------------------------------
constexpr int kEvalGroupsPerWave = 1;

__device__ __forceinline__ void eval_group(const ClassifyArgs& a, int64_t g, unsigned long long b0,
                                           unsigned long long b1, unsigned long long b2, unsigned long long b3,
                                           unsigned long long* __restrict__ aligned, uint4* __restrict__ staging,
                                           int lane);

// One wave per group; waves whose group has no candidate exit after one broadcast load of their 32 bytes of bits.
// (Letting a wave walk several groups was measured slower: the two dependent memory round trips of every active
// group then serialise inside the wave, while the launch of ~N/256 mostly empty waves costs ~20 us on C2.)
__global__ __launch_bounds__(256) void eval_kernel(ClassifyArgs a, const unsigned long long* __restrict__ bitmask,
                                                   int64_t n_groups, unsigned long long* __restrict__ aligned,
                                                   uint4* __restrict__ staging) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t g = (int64_t)blockIdx.x * 4 + wave;
    if (g >= n_groups) return;
    const ulonglong2 w0 = *reinterpret_cast<const ulonglong2*>(bitmask + g * 4);
    const ulonglong2 w1 = *reinterpret_cast<const ulonglong2*>(bitmask + g * 4 + 2);
    if ((w0.x | w0.y | w1.x | w1.y) == 0ull) return;
    eval_group(a, g, w0.x, w0.y, w1.x, w1.y, aligned, staging, lane);
}

__device__ __forceinline__ void eval_group(const ClassifyArgs& a, int64_t g, unsigned long long b0,
                                           unsigned long long b1, unsigned long long b2, unsigned long long b3,
                                           unsigned long long* __restrict__ aligned, uint4* __restrict__ staging,
                                           int lane) {
    const ulonglong2 w0 = make_ulonglong2(b0, b1), w1 = make_ulonglong2(b2, b3);
    const bool c[4] = {(bool)((w0.x >> lane) & 1ull), (bool)((w0.y >> lane) & 1ull), (bool)((w1.x >> lane) & 1ull),
                       (bool)((w1.y >> lane) & 1ull)};
    const int cnt = (int)c[0] + (int)c[1] + (int)c[2] + (int)c[3];
    const int excl = wave_incl_scan(cnt, lane) - cnt;
    const int64_t i0 = g * kGroup + (int64_t)lane * 4;
    int32_t r_tid[4], r_mtid[4], r_pos[4], r_mpos[4];
    uint32_t r_flag[4], r_mapq[4], r_qlen[4];
    if (cnt) {
        if (i0 + 4 <= a.n) {
            const int4 v0 = *reinterpret_cast<const int4*>(a.tid + i0);
            const int4 v1 = *reinterpret_cast<const int4*>(a.mtid + i0);
            const int4 v2 = *reinterpret_cast<const int4*>(a.pos + i0);
            const int4 v3 = *reinterpret_cast<const int4*>(a.mpos + i0);
            const ushort4 f = *reinterpret_cast<const ushort4*>(a.flag + i0);
            const uchar4 m = *reinterpret_cast<const uchar4*>(a.mapq + i0);
            const ushort4 q = *reinterpret_cast<const ushort4*>(a.qlen + i0);
            r_tid[0] = v0.x; r_tid[1] = v0.y; r_tid[2] = v0.z; r_tid[3] = v0.w;
            r_mtid[0] = v1.x; r_mtid[1] = v1.y; r_mtid[2] = v1.z; r_mtid[3] = v1.w;
            r_pos[0] = v2.x; r_pos[1] = v2.y; r_pos[2] = v2.z; r_pos[3] = v2.w;
            r_mpos[0] = v3.x; r_mpos[1] = v3.y; r_mpos[2] = v3.z; r_mpos[3] = v3.w;
            r_flag[0] = f.x; r_flag[1] = f.y; r_flag[2] = f.z; r_flag[3] = f.w;
            r_mapq[0] = m.x; r_mapq[1] = m.y; r_mapq[2] = m.z; r_mapq[3] = m.w;
            r_qlen[0] = q.x; r_qlen[1] = q.y; r_qlen[2] = q.z; r_qlen[3] = q.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int64_t i = i0 + k;
                const bool in = i < a.n;
                r_tid[k] = in ? a.tid[i] : -1;
                r_mtid[k] = in ? a.mtid[i] : -1;
                r_pos[k] = in ? a.pos[i] : 0;
                r_mpos[k] = in ? a.mpos[i] : 0;
                r_flag[k] = in ? a.flag[i] : 0;
                r_mapq[k] = in ? a.mapq[i] : 0;
                r_qlen[k] = in ? a.qlen[i] : 0;
            }
        }
    }
    // contig rows of all this lane's candidates first (their gathers overlap), then the arithmetic
    ContigRow c1[4], c2[4];
    bool in_range[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        in_range[k] = c[k] && (uint32_t)r_tid[k] < (uint32_t)a.n_contigs && (uint32_t)r_mtid[k] < (uint32_t)a.n_contigs;
        if (in_range[k]) {
            c1[k] = a.table[r_tid[k]];
            c2[k] = a.table[r_mtid[k]];
        }
    }
    int32_t cov_tid = -1;
    int cov_sum = 0;
    int slot = excl;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (!c[k]) continue;
        const Eval e = eval_record(a, in_range[k], c1[k], c2[k], r_tid[k], r_mtid[k], r_pos[k], r_mpos[k], r_flag[k],
                                   r_mapq[k]);
        if (e.bits & EV_COV) {
            if (r_tid[k] != cov_tid && cov_sum) {          // a second contig inside one lane: rare, flush directly
                atomicAdd(&aligned[cov_tid], (unsigned long long)cov_sum);
                cov_sum = 0;
            }
            cov_tid = r_tid[k];
            cov_sum += (int)r_qlen[k];
        }
        uint4 ent;
        ent.x = (uint32_t)e.o1;
        ent.y = (uint32_t)e.o2;
        ent.z = e.n_min | ((e.bits & EV_REACH) ? 1u << 29 : 0u) | ((e.bits & EV_FISHY) ? 1u << 30 : 0u) |
                ((e.bits & EV_NONUNIQ) ? 1u << 31 : 0u);
        ent.w = e.n_max | ((e.bits & EV_MAPQ0) ? 1u << 29 : 0u) | ((e.bits & EV_CASEA) ? 1u << 30 : 0u) |
                ((e.bits & EV_FIRSTMIN) ? 1u << 31 : 0u);
        staging[g * kGroup + slot] = ent;
        slot++;
    }
    wave_add_by_key(aligned, cov_tid, cov_sum, cov_sum != 0, lane);
}

// ---------------------------------------------------------------------------------------------------------
// ordered_kernel: the order-dependent part, over the evaluated candidates only.  Single-wave workgroups; the
// workgroup covers kCandGroups consecutive groups and walks their staged entries 64 at a time in stream order:
// previous reaching observation via ballot, CreateEdge semantics (acceptance rule :840 re-evaluated from the
// stored observations), ordered slots for the emitted tuples.  The chain is carried in registers, nothing but a
// 64-entry prefix table lives in LDS.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kCandThreads) void ordered_kernel(
    ClassifyArgs a, const unsigned long long* __restrict__ bitmask, int64_t n_groups,
    const uint4* __restrict__ staging, uint64_t* __restrict__ seg_keys, uint64_t* __restrict__ seg_payload,
    BlockSummary* __restrict__ summ) {
    __shared__ int s_pre[kCandThreads + 1];
    const int lane = threadIdx.x;
    const int64_t block_base = (int64_t)blockIdx.x * kClsTile;
    const int64_t g0 = (int64_t)blockIdx.x * kCandGroups;
    const int64_t g = g0 + lane;
    int cnt = 0;
    if (lane < kCandGroups && g < n_groups) {
        const ulonglong2 w0 = *reinterpret_cast<const ulonglong2*>(bitmask + g * 4);
        const ulonglong2 w1 = *reinterpret_cast<const ulonglong2*>(bitmask + g * 4 + 2);
        cnt = __popcll(w0.x) + __popcll(w0.y) + __popcll(w1.x) + __popcll(w1.y);
    }
    const int incl = wave_incl_scan(cnt, lane);
    const int total = __shfl(incl, 63, 64);
    s_pre[lane] = incl - cnt;
    if (lane == 0) s_pre[kCandThreads] = total;
    __syncthreads();

    // chain state, identical in every lane
    bool prev_known = false;
    int32_t prev1 = 0, prev2 = 0;
    bool blk_has = false;
    int emit_base = 0;
    bool head_present = false;
    int32_t head1 = 0, head2 = 0;
    uint32_t head_info = 0, head_slot = kNoSlot;
    int c_count = 0, c_nonuniq = 0, c_nus = 0, c_dup = 0, c_long = 0, c_fishy = 0, c_reach = 0;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const uint32_t mask_a = a.no_score ? BESST_MASK_GPRIME : (BESST_MASK_G | (a.extend_paths ? BESST_MASK_GPRIME : 0u));

    constexpr int kAhead = 4;    // chunks whose entries are fetched before the chain consumes them
    for (int c0 = 0; c0 < total; c0 += kCandThreads * kAhead) {
        uint4 ents[kAhead];
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            const int j = c0 + u * kCandThreads + lane;
            ents[u] = make_uint4(0u, 0u, 0u, 0u);
            if (j < total) {
                int lo = 0, hi = kCandThreads;          // last group whose exclusive prefix is <= j
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (s_pre[mid] <= j) lo = mid; else hi = mid;
                }
                ents[u] = staging[(g0 + lo) * kGroup + (j - s_pre[lo])];
            }
        }
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            if (c0 + u * kCandThreads >= total) break;       // uniform
            const uint4 ent = ents[u];
            const int32_t o1 = (int32_t)ent.x, o2 = (int32_t)ent.y;
            const bool reach = (ent.z >> 29) & 1u, fishy = (ent.z >> 30) & 1u, nonuniq = (ent.z >> 31) & 1u;
            const bool mapq0 = (ent.w >> 29) & 1u, case_a = (ent.w >> 30) & 1u, first_min = (ent.w >> 31) & 1u;
            const uint32_t n_min = ent.z & 0x1fffffffu, n_max = ent.w & 0x1fffffffu;
            const bool dbl = case_a && a.extend_paths && !a.no_score;
            c_nonuniq += nonuniq ? 1 : 0;
            c_fishy += fishy ? 1 : 0;
            c_reach += reach ? 1 : 0;
            const unsigned long long has_mask = __ballot(reach);
            bool pk = prev_known;
            int32_t p1 = prev1, p2 = prev2;
            {
                const unsigned long long below = has_mask & lt_mask;
                const int src = below ? 63 - __clzll((long long)below) : 0;
                const int32_t q1 = __shfl(o1, src, 64), q2 = __shfl(o2, src, 64);
                if (below) { pk = true; p1 = q1; p2 = q2; }
            }
            const bool accept = reach && ((double)((int64_t)o1 + o2) < a.ins_size_threshold) && o1 > 25 && o2 > 25;
            bool emit = fishy;
            bool is_head = false;
            if (reach) {
                if (!pk) {
                    is_head = true;                      // first reaching record of the workgroup
                    emit = accept;
                } else {
                    const CEDelta d = create_edge(o1, o2, p1, p2, accept, dbl, mapq0, a.detect_dup != 0);
                    c_count += d.count; c_nus += d.nus; c_dup += d.dup; c_long += d.too_long;
                    emit = d.keep;
                }
            }
            const unsigned long long emit_mask = __ballot(emit);
            const int slot = emit_base + __popcll(emit_mask & lt_mask);
            if (emit) {
                const uint32_t mask = case_a ? mask_a : BESST_MASK_GPRIME;
                const uint32_t lo = fishy ? 0u : (uint32_t)(first_min ? o1 : o2);
                const uint32_t hi = fishy ? 0u : ((uint32_t)(first_min ? o2 : o1) | (mask << 30));
                seg_keys[block_base + slot] = ((((uint64_t)n_min << a.node_bits) | n_max) << 1) | (fishy ? 1u : 0u);
                seg_payload[block_base + slot] = (uint64_t)lo | ((uint64_t)hi << 32);
            }
            // the head is unique per workgroup; broadcast it to every lane
            const unsigned long long head_mask = __ballot(is_head);
            if (head_mask) {
                const int hl = __ffsll((long long)head_mask) - 1;
                head_present = true;
                head1 = __shfl(o1, hl, 64);
                head2 = __shfl(o2, hl, 64);
                const bool h_acc = __shfl((int)accept, hl, 64), h_dbl = __shfl((int)dbl, hl, 64);
                const bool h_mq0 = __shfl((int)mapq0, hl, 64);
                head_info = (h_acc ? 9u : 0u) | (h_dbl ? 2u : 0u) | (h_mq0 ? 4u : 0u);
                const int hs = __shfl(slot, hl, 64);
                head_slot = h_acc ? (uint32_t)hs : kNoSlot;
            }
            if (has_mask) {
                const int src = 63 - __clzll((long long)has_mask);
                blk_has = true;
                prev_known = true;
                prev1 = __shfl(o1, src, 64);
                prev2 = __shfl(o2, src, 64);
            }
            emit_base += __popcll(emit_mask);
        }
    }

    int tot[7];
    {
        const int vals[7] = {c_count, c_nonuniq, c_nus, c_dup, c_long, c_fishy, c_reach};
#pragma unroll
        for (int f = 0; f < 7; ++f) tot[f] = wave_sum(vals[f]);
    }
    if (lane == 0) {
        BlockSummary s;
#pragma unroll
        for (int f = 0; f < 7; ++f) s.ctr[f] = (uint32_t)tot[f];
        s.ctr[7] = 0;
        s.n_emit = (uint32_t)emit_base;
        s.has_reach = blk_has ? 1u : 0u;
        s.first_o1 = head_present ? head1 : 0;
        s.first_o2 = head_present ? head2 : 0;
        s.last_o1 = prev1;
        s.last_o2 = prev2;
        s.head_info = head_present ? head_info : 0u;
        s.head_slot = head_slot;
        summ[blockIdx.x] = s;
    }
}

// ---- stitch: resolve block heads, fix counters, scan tuple counts ------------------------------------
template <bool kMax>
__device__ __forceinline__ int block_incl_scan_1024(int v, int* s_w, int t) {
    const int lane = t & 63, wave = t >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(v, d, 64);
        if (lane >= d) v = kMax ? (o > v ? o : v) : v + o;
    }
    __syncthreads();
    if (lane == 63) s_w[wave] = v;
    __syncthreads();
    int acc = kMax ? -1 : 0;
    for (int w = 0; w < wave; ++w) acc = kMax ? (s_w[w] > acc ? s_w[w] : acc) : acc + s_w[w];
    return kMax ? (acc > v ? acc : v) : v + acc;
}

__global__ __launch_bounds__(1024) void stitch_kernel(const BlockSummary* __restrict__ summ,
                                                      uint32_t nblocks, int32_t* carry, int detect,
                                                      uint32_t* __restrict__ offsets,
                                                      uint32_t* __restrict__ skip_slot,
                                                      uint32_t* n_out,
                                                      unsigned long long* counters) {
    __shared__ int s_w[16];
    __shared__ int32_t s_l1[1024], s_l2[1024];
    __shared__ int32_t s_carry[2];
    __shared__ int s_base;
    __shared__ int s_redc[16][7];
    const int t = threadIdx.x;
    if (t == 0) { s_carry[0] = carry[0]; s_carry[1] = carry[1]; s_base = 0; }
    __syncthreads();
    int c_count = 0, c_long = 0, c_dup = 0, c_nus = 0, c_nonuniq = 0, c_fishy = 0, c_reach = 0;
    for (uint32_t c0 = 0; c0 < nblocks; c0 += 1024) {
        const uint32_t b = c0 + t;
        BlockSummary s;
        s.n_emit = 0; s.has_reach = 0; s.first_o1 = s.first_o2 = s.last_o1 = s.last_o2 = 0;
        s.head_info = 0; s.head_slot = kNoSlot;
        if (b < nblocks) {
            s = summ[b];
            c_count += (int)s.ctr[0]; c_nonuniq += (int)s.ctr[1]; c_nus += (int)s.ctr[2]; c_dup += (int)s.ctr[3];
            c_long += (int)s.ctr[4]; c_fishy += (int)s.ctr[5]; c_reach += (int)s.ctr[6];
        }
        s_l1[t] = s.last_o1;
        s_l2[t] = s.last_o2;
        const int incl = block_incl_scan_1024<true>(s.has_reach ? t : -1, s_w, t);
        __syncthreads();
        const int excl = __shfl_up(incl, 1, 64);
        int prev_idx;
        if ((t & 63) == 0) {
            prev_idx = -1;
            for (int w = 0; w < (t >> 6); ++w) prev_idx = s_w[w] > prev_idx ? s_w[w] : prev_idx;
        } else {
            prev_idx = excl;
        }
        int32_t p1 = prev_idx >= 0 ? s_l1[prev_idx] : s_carry[0];
        int32_t p2 = prev_idx >= 0 ? s_l2[prev_idx] : s_carry[1];
        uint32_t n_final = s.n_emit, skip = kNoSlot;
        if (s.has_reach) {
            const CEDelta d = create_edge(s.first_o1, s.first_o2, p1, p2, s.head_info & 1u,
                                          s.head_info & 2u, s.head_info & 4u, detect != 0);
            c_count += d.count; c_long += d.too_long; c_dup += d.dup; c_nus += d.nus;
            if ((s.head_info & 8u) && !d.keep) { n_final -= 1; skip = s.head_slot; }
        }
        const int sum_incl = block_incl_scan_1024<false>((int)n_final, s_w, t);
        const int base = s_base;
        if (b < nblocks) {
            offsets[b] = (uint32_t)(base + sum_incl - (int)n_final);
            skip_slot[b] = skip;
        }
        __syncthreads();
        if (t == 1023) {
            s_base = base + sum_incl;
            if (incl >= 0) { s_carry[0] = s_l1[incl]; s_carry[1] = s_l2[incl]; }
        }
        __syncthreads();
    }
    // counters fixed up by the heads
    // besst_counters fields: count 0, non_unique 1, non_unique_for_scaf 2, nr_of_duplicates 3,
    // reads_with_too_long_insert 4, fishy_reads 5, n_reach 7 (n_tuples, 6, below)
    int vals[7] = {c_count, c_nonuniq, c_nus, c_dup, c_long, c_fishy, c_reach};
    const int lane = t & 63, wave = t >> 6;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int v = wave_sum(vals[j]);
        if (lane == 0) s_redc[wave][j] = v;
    }
    __syncthreads();
    if (t < 7) {
        long long v = 0;
        for (int w = 0; w < 16; ++w) v += s_redc[w][t];
        if (v) atomicAdd(&counters[t < 6 ? t : 7], (unsigned long long)v);
    }
    if (t == 0) {
        carry[0] = s_carry[0];
        carry[1] = s_carry[1];
        *n_out = (uint32_t)s_base;
        atomicAdd(&counters[6], (unsigned long long)s_base);
    }
}

__global__ __launch_bounds__(256) void compact_kernel(const BlockSummary* __restrict__ summ,
                                                      const uint32_t* __restrict__ offsets,
                                                      const uint32_t* __restrict__ skip_slot,
                                                      const uint64_t* __restrict__ seg_keys,
                                                      const uint64_t* __restrict__ seg_payload,
                                                      uint64_t* __restrict__ keys,
                                                      uint64_t* __restrict__ payload,
                                                      const uint8_t* __restrict__ cls8, int32_t n_contigs,
                                                      unsigned long long* __restrict__ aligned) {
    // coverage of contigs that are not in the table is not part of cont_aligned_len (CreateGraph.py:89-95)
    for (int32_t c = (int32_t)(blockIdx.x * blockDim.x + threadIdx.x); c < n_contigs; c += (int32_t)(gridDim.x * blockDim.x))
        if (!cls8[c]) aligned[c] = 0;
    const uint32_t b = blockIdx.x;
    const uint32_t n = summ[b].n_emit;
    if (n == 0) return;
    const uint32_t skip = skip_slot[b], off = offsets[b];
    const int64_t base = (int64_t)b * kClsTile;
    for (uint32_t j = threadIdx.x; j < n; j += blockDim.x) {
        if (j == skip) continue;
        const uint32_t dst = off + j - (j > skip ? 1u : 0u);
        keys[dst] = seg_keys[base + j];
        payload[dst] = seg_payload[base + j];
    }
}

struct ClsWorkspace {
    uint64_t* seg_keys;
    uint64_t* seg_payload;
    BlockSummary* summ;
    uint32_t* offsets;
    uint32_t* skip;
    unsigned long long* bitmask;
    uint4* staging;
    int64_t n_groups;
    size_t total;
};

ClsWorkspace carve(void* ws, int64_t n) {
    const int64_t nblocks = (n + kClsTile - 1) / kClsTile;
    const size_t seg = (size_t)nblocks * kClsTile * sizeof(uint64_t);
    ClsWorkspace w;
    char* p = static_cast<char*>(ws);
    size_t off = 0;
    w.seg_keys = reinterpret_cast<uint64_t*>(p + off); off += align_up(seg, 256);
    w.seg_payload = reinterpret_cast<uint64_t*>(p + off); off += align_up(seg, 256);
    w.summ = reinterpret_cast<BlockSummary*>(p + off); off += align_up((size_t)nblocks * sizeof(BlockSummary), 256);
    w.offsets = reinterpret_cast<uint32_t*>(p + off); off += align_up((size_t)nblocks * 4, 256);
    w.skip = reinterpret_cast<uint32_t*>(p + off); off += align_up((size_t)nblocks * 4, 256);
    // candidate bits: stream_kernel writes whole workgroups, so round the group count up to its tile
    const int64_t stream_blocks = (n + kStreamTile - 1) / kStreamTile;
    w.n_groups = stream_blocks * (kStreamTile / kGroup);
    w.bitmask = reinterpret_cast<unsigned long long*>(p + off); off += align_up((size_t)w.n_groups * 32, 256);
    // evaluated candidates, 16 B each, group g at [g*256, ...): every record may be a candidate
    w.staging = reinterpret_cast<uint4*>(p + off); off += align_up((size_t)w.n_groups * kGroup * sizeof(uint4), 256);
    w.total = off;
    return w;
}

}  // namespace

size_t classify_workspace_bytes(int64_t n) {
    if (n < 1) n = 1;
    return carve(nullptr, n).total;
}

namespace {

// last record of the slice that reached CreateEdge: {has, obs1, obs2, 0}
__global__ __launch_bounds__(256) void tail_kernel(const BlockSummary* __restrict__ summ, uint32_t nblocks,
                                                   int32_t* __restrict__ tail) {
    __shared__ int s_best[4];
    int best = -1;
    for (uint32_t b = threadIdx.x; b < nblocks; b += blockDim.x)
        if (summ[b].has_reach) best = (int)b > best ? (int)b : best;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const int o = __shfl_xor(best, d, 64);
        best = o > best ? o : best;
    }
    if ((threadIdx.x & 63) == 0) s_best[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        best = max(max(s_best[0], s_best[1]), max(s_best[2], s_best[3]));
        tail[0] = best >= 0 ? 1 : 0;
        tail[1] = best >= 0 ? summ[best].last_o1 : 0;
        tail[2] = best >= 0 ? summ[best].last_o2 : 0;
        tail[3] = 0;
    }
}

__global__ void zero_tail_kernel(int32_t* tail) {
    if (threadIdx.x < 4) tail[threadIdx.x] = 0;
}

// prev_obs entering rank `rank`: the tail of the nearest earlier rank that has one, else what is in carry
__global__ void resolve_carry_kernel(const int32_t* __restrict__ tails, int rank, int32_t* __restrict__ carry) {
    if (threadIdx.x != 0) return;
    for (int j = 0; j < rank; ++j)
        if (tails[j * 4]) { carry[0] = tails[j * 4 + 1]; carry[1] = tails[j * 4 + 2]; }
}

}  // namespace

int launch_classify_scan(hipStream_t s, const ClassifyArgs& a, int64_t* aligned, besst_counters* counters,
                         void* ws, size_t ws_bytes) {
    if (a.n <= 0) return BESST_OK;
    BESST_REQUIRE(a.n < (int64_t)1 << 32, "classify: more than 2^32-1 records in one call");
    const ClsWorkspace w = carve(ws, a.n);
    BESST_REQUIRE(ws != nullptr && ws_bytes >= w.total, "classify: workspace too small");
    const uint32_t nblocks = (uint32_t)((a.n + kClsTile - 1) / kClsTile);
    const uint32_t stream_blocks = (uint32_t)((a.n + kStreamTile - 1) / kStreamTile);
    // Mate-pair ('rf') libraries have inserts comparable to the contig lengths, so a large share of the records
    // are candidates and evaluating them inside the streaming pass (lines still hot, no second sweep) wins a few
    // percent; for paired-end libraries the separate, mostly-empty eval launch is as fast.  BESST_FUSE_EVAL=0/1
    // overrides the choice (development knob).
    static const int fuse_env = getenv("BESST_FUSE_EVAL") ? atoi(getenv("BESST_FUSE_EVAL")) : -1;
    const bool fuse = fuse_env >= 0 ? fuse_env != 0 : a.rf != 0;
    {
        ProfScope ps(s, kProfClassify);
        if (fuse)
            hipLaunchKernelGGL(stream_kernel<true>, dim3(stream_blocks), dim3(kStreamThreads), 0, s, a,
                               reinterpret_cast<unsigned long long*>(aligned), w.bitmask, w.staging);
        else
            hipLaunchKernelGGL(stream_kernel<false>, dim3(stream_blocks), dim3(kStreamThreads), 0, s, a,
                               reinterpret_cast<unsigned long long*>(aligned), w.bitmask, w.staging);
    }
    if (!fuse) {
        ProfScope ps(s, kProfCandidate);
        hipLaunchKernelGGL(eval_kernel, dim3((uint32_t)((w.n_groups + 4 * kEvalGroupsPerWave - 1) / (4 * kEvalGroupsPerWave))), dim3(256), 0, s, a, w.bitmask, w.n_groups,
                           reinterpret_cast<unsigned long long*>(aligned), w.staging);
    }
    {
        ProfScope ps(s, kProfOrdered);
        hipLaunchKernelGGL(ordered_kernel, dim3(nblocks), dim3(kCandThreads), 0, s, a, w.bitmask, w.n_groups, w.staging,
                           w.seg_keys, w.seg_payload, w.summ);
    }
    (void)counters;
    BESST_HIP_TRY(hipGetLastError());
    return BESST_OK;
}

int launch_classify_tail(hipStream_t s, int64_t n, int32_t* tail, void* ws, size_t ws_bytes) {
    if (n <= 0) {
        hipLaunchKernelGGL(zero_tail_kernel, dim3(1), dim3(64), 0, s, tail);
    } else {
        const ClsWorkspace w = carve(ws, n);
        BESST_REQUIRE(ws != nullptr && ws_bytes >= w.total, "classify: workspace too small");
        hipLaunchKernelGGL(tail_kernel, dim3(1), dim3(256), 0, s, w.summ, (uint32_t)((n + kClsTile - 1) / kClsTile), tail);
    }
    BESST_HIP_TRY(hipGetLastError());
    return BESST_OK;
}

int launch_resolve_carry(hipStream_t s, const int32_t* tails, int rank, int32_t* carry) {
    hipLaunchKernelGGL(resolve_carry_kernel, dim3(1), dim3(64), 0, s, tails, rank, carry);
    BESST_HIP_TRY(hipGetLastError());
    return BESST_OK;
}

int launch_classify_emit(hipStream_t s, int64_t n, int detect_dup, int32_t* carry, uint64_t* keys,
                         uint64_t* payload, uint32_t* n_out, besst_counters* counters, void* ws,
                         size_t ws_bytes, const uint8_t* cls8, int32_t n_contigs, int64_t* aligned) {
    if (n <= 0) {
        BESST_HIP_TRY(hipMemsetAsync(n_out, 0, sizeof(uint32_t), s));
        return BESST_OK;
    }
    const ClsWorkspace w = carve(ws, n);
    BESST_REQUIRE(ws != nullptr && ws_bytes >= w.total, "classify: workspace too small");
    const uint32_t nblocks = (uint32_t)((n + kClsTile - 1) / kClsTile);
    auto* ctr = reinterpret_cast<unsigned long long*>(counters);
    {
        ProfScope ps(s, kProfStitch);
        hipLaunchKernelGGL(stitch_kernel, dim3(1), dim3(1024), 0, s, w.summ, nblocks, carry, detect_dup, w.offsets,
                           w.skip, n_out, ctr);
    }
    {
        ProfScope ps(s, kProfCompact);
        hipLaunchKernelGGL(compact_kernel, dim3(nblocks), dim3(256), 0, s, w.summ, w.offsets, w.skip, w.seg_keys,
                           w.seg_payload, keys, payload, cls8, n_contigs, reinterpret_cast<unsigned long long*>(aligned));
    }
    BESST_HIP_TRY(hipGetLastError());
    return BESST_OK;
}

int launch_classify(hipStream_t s, const ClassifyArgs& a, int32_t* carry, int64_t* aligned, uint64_t* keys,
                    uint64_t* payload, uint32_t* n_out, besst_counters* counters, void* ws, size_t ws_bytes) {
    int rc = launch_classify_scan(s, a, aligned, counters, ws, ws_bytes);
    if (rc) return rc;
    return launch_classify_emit(s, a.n, a.detect_dup, carry, keys, payload, n_out, counters, ws, ws_bytes, a.cls8,
                                a.n_contigs, aligned);
}

}  // namespace besst
