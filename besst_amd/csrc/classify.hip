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

#include "common.h"

namespace besst {

namespace {

constexpr uint32_t kNoSlot = 0xffffffffu;

// evaluation bits of one record
constexpr uint32_t EV_COV = 1u, EV_FISHY = 2u, EV_NONUNIQ = 4u, EV_REACH = 8u, EV_ACCEPT = 16u,
                   EV_DOUBLE = 32u, EV_MAPQ0 = 64u;

struct Eval {
    uint32_t bits;
    int32_t o1, o2;
    uint64_t key;     // sort key of the tuple this record may emit
    uint32_t lo, hi;  // payload words
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

__device__ __forceinline__ Eval eval_record(const ClassifyArgs& a, int32_t tid, int32_t mtid,
                                            int32_t pos, int32_t mpos, uint32_t flag, uint32_t mapq) {
    Eval e;
    e.bits = 0;
    e.o1 = e.o2 = 0;
    e.key = 0;
    e.lo = e.hi = 0;
    if ((uint32_t)tid >= (uint32_t)a.n_contigs || (uint32_t)mtid >= (uint32_t)a.n_contigs) return e;
    const ContigRow c1 = a.table[tid];
    const ContigRow c2 = (mtid == tid) ? c1 : a.table[mtid];
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
        const uint64_t n1 = (uint64_t)scaf1 * 2 + s1, n2 = (uint64_t)scaf2 * 2 + s2;
        const uint64_t lo = n1 < n2 ? n1 : n2, hi = n1 < n2 ? n2 : n1;
        e.key = (((lo << a.node_bits) | hi) << 1) | 1u;
        e.bits |= EV_FISHY;
        return e;   // an unmapped record cannot also be a link candidate (:169)
    }

    if (!(other && (flag & kFlagRead2) && !(flag & kFlagUnmapped) && (int32_t)mapq >= a.min_mapq))
        return e;
    uint32_t mask;
    bool dbl = false;
    if (cls1 == BESST_CLS_LARGE && cls2 == BESST_CLS_LARGE && scaf1 != scaf2) {
        if (a.no_score) {
            mask = BESST_MASK_GPRIME;
        } else {
            mask = BESST_MASK_G | (a.extend_paths ? BESST_MASK_GPRIME : 0u);
            dbl = a.extend_paths != 0;
        }
    } else if (a.extend_paths) {
        const bool sm1 = cls1 == BESST_CLS_SMALL, sm2 = cls2 == BESST_CLS_SMALL;
        if ((sm1 && sm2 && scaf1 != scaf2) || (sm1 != sm2)) mask = BESST_MASK_GPRIME;
        else return e;
    } else {
        return e;
    }
    uint32_t s1, s2;
    posdir(rf, dir1, rdir, c1.ctg_pos, pos, c1.scaf_len, c1.ctg_len, a.read_len, e.o1, s1);
    posdir(rf, dir2, mdir, c2.ctg_pos, mpos, c2.scaf_len, c2.ctg_len, a.read_len, e.o2, s2);
    e.bits |= EV_REACH | (dbl ? EV_DOUBLE : 0u);
    const bool accept = ((double)((int64_t)e.o1 + e.o2) < a.ins_size_threshold) && e.o1 > 25 && e.o2 > 25;
    if (accept) {
        e.bits |= EV_ACCEPT;
        const uint64_t n1 = (uint64_t)scaf1 * 2 + s1, n2 = (uint64_t)scaf2 * 2 + s2;
        const bool first_min = n1 < n2;
        const uint64_t lo = first_min ? n1 : n2, hi = first_min ? n2 : n1;
        e.key = ((lo << a.node_bits) | hi) << 1;
        e.lo = (uint32_t)(first_min ? e.o1 : e.o2);
        e.hi = (uint32_t)(first_min ? e.o2 : e.o1) | (mask << 30);
    }
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

__global__ __launch_bounds__(kClsThreads) void classify_kernel(
    ClassifyArgs a, unsigned long long* __restrict__ aligned, uint64_t* __restrict__ seg_keys,
    uint64_t* __restrict__ seg_payload, BlockSummary* __restrict__ summ,
    unsigned long long* __restrict__ counters) {
    __shared__ int32_t s_has[4], s_o1[4], s_o2[4];
    __shared__ int32_t s_cnt[4];
    __shared__ int32_t s_head[5];       // o1, o2, info, slot, present
    __shared__ int32_t s_red[4][8];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t block_base = (int64_t)blockIdx.x * kClsTile;
    if (t == 0) { s_head[4] = 0; s_head[3] = (int32_t)kNoSlot; }

    // block-carried state, identical in every thread
    bool prev_known = false;
    int32_t prev1 = 0, prev2 = 0;
    bool blk_has = false;
    int32_t last1 = 0, last2 = 0;
    int emit_base = 0;
    int c_count = 0, c_nonuniq = 0, c_nus = 0, c_dup = 0, c_long = 0, c_fishy = 0, c_reach = 0;

    for (int st = 0; st < kClsSubTiles; ++st) {
        const int64_t i0 = block_base + (int64_t)st * kClsSubTile + (int64_t)t * kClsVec;
        int32_t r_tid[kClsVec], r_mtid[kClsVec], r_pos[kClsVec], r_mpos[kClsVec];
        uint32_t r_flag[kClsVec], r_mapq[kClsVec], r_qlen[kClsVec];
        if (i0 + kClsVec <= a.n) {
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
            for (int k = 0; k < kClsVec; ++k) {
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

        // ---- phase 1: evaluate, coverage, per-thread chain summary --------------------------------
        Eval e[kClsVec];
        bool th_has = false;
        int32_t tl1 = 0, tl2 = 0;
#pragma unroll
        for (int k = 0; k < kClsVec; ++k) {
            e[k] = eval_record(a, r_tid[k], r_mtid[k], r_pos[k], r_mpos[k], r_flag[k], r_mapq[k]);
            if (e[k].bits & EV_REACH) { th_has = true; tl1 = e[k].o1; tl2 = e[k].o2; }
        }
        {   // coverage: the stream is (tid,pos)-sorted, so a wave's records usually share one tid
            const int32_t ref_tid = __shfl(r_tid[0], 0, 64);
            bool uni = true;
            int mine = 0;
#pragma unroll
            for (int k = 0; k < kClsVec; ++k) {
                uni = uni && (r_tid[k] == ref_tid);
                if (e[k].bits & EV_COV) mine += (int)r_qlen[k];
            }
            if (__all(uni)) {
                const int tot = wave_sum(mine);
                if (lane == 0 && tot) atomicAdd(&aligned[ref_tid], (unsigned long long)tot);
            } else {
                int run_tid = -1, run_sum = 0;
#pragma unroll
                for (int k = 0; k < kClsVec; ++k) {
                    if (!(e[k].bits & EV_COV)) continue;
                    if (r_tid[k] != run_tid) {
                        if (run_sum) atomicAdd(&aligned[run_tid], (unsigned long long)run_sum);
                        run_tid = r_tid[k];
                        run_sum = 0;
                    }
                    run_sum += (int)r_qlen[k];
                }
                if (run_sum) atomicAdd(&aligned[run_tid], (unsigned long long)run_sum);
            }
        }
        const unsigned long long has_mask = __ballot(th_has);
        if (lane == 0) s_has[wave] = has_mask != 0ull;
        if (has_mask != 0ull && lane == 63 - __clzll((long long)has_mask)) { s_o1[wave] = tl1; s_o2[wave] = tl2; }
        __syncthreads();

        // ---- phase 2: incoming prev_obs for this thread ---------------------------------------------
        bool pk = prev_known;
        int32_t p1 = prev1, p2 = prev2;
        for (int w = 0; w < wave; ++w)
            if (s_has[w]) { pk = true; p1 = s_o1[w]; p2 = s_o2[w]; }
        {
            const unsigned long long below = has_mask & ((1ull << lane) - 1ull);
            const int src = below ? 63 - __clzll((long long)below) : 0;
            const int32_t q1 = __shfl(tl1, src, 64), q2 = __shfl(tl2, src, 64);
            if (below) { pk = true; p1 = q1; p2 = q2; }
        }
        // sequential CreateEdge semantics over this thread's records
        int n_emit = 0;
        uint32_t emit_bits = 0;
        int head_k = -1;
        int32_t h1 = 0, h2 = 0;
        uint32_t hbits = 0;
#pragma unroll
        for (int k = 0; k < kClsVec; ++k) {
            const uint32_t b = e[k].bits;
            if (b & EV_NONUNIQ) c_nonuniq++;
            if (b & EV_FISHY) { c_fishy++; emit_bits |= 1u << k; n_emit++; }
            if (!(b & EV_REACH)) continue;
            c_reach++;
            const bool accept = b & EV_ACCEPT;
            if (!pk) {
                // head of the block: predecessor unknown, resolved by the stitch kernel
                head_k = k;
                h1 = e[k].o1; h2 = e[k].o2; hbits = b;
                if (accept) { emit_bits |= 1u << k; n_emit++; }
            } else {
                const CEDelta d = create_edge(e[k].o1, e[k].o2, p1, p2, accept, b & EV_DOUBLE,
                                              b & EV_MAPQ0, a.detect_dup != 0);
                c_count += d.count; c_long += d.too_long; c_dup += d.dup; c_nus += d.nus;
                if (d.keep) { emit_bits |= 1u << k; n_emit++; }
            }
            pk = true; p1 = e[k].o1; p2 = e[k].o2;
        }
        // ordered slots inside the block-local segment
        const int incl = wave_incl_scan(n_emit, lane);
        if (lane == 63) s_cnt[wave] = incl;
        __syncthreads();
        int slot = emit_base + incl - n_emit;
        int sub_total = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w < wave) slot += s_cnt[w];
            sub_total += s_cnt[w];
        }
#pragma unroll
        for (int k = 0; k < kClsVec; ++k) {
            if (!(emit_bits & (1u << k))) continue;
            if (k == head_k) s_head[3] = slot;
            seg_keys[block_base + slot] = e[k].key;
            seg_payload[block_base + slot] = (uint64_t)e[k].lo | ((uint64_t)e[k].hi << 32);
            slot++;
        }
        if (head_k >= 0) {
            s_head[0] = h1; s_head[1] = h2;
            s_head[2] = (int32_t)(((hbits & EV_ACCEPT) ? 9u : 0u) | ((hbits & EV_DOUBLE) ? 2u : 0u) |
                                  ((hbits & EV_MAPQ0) ? 4u : 0u));
            s_head[4] = 1;
        }
        // advance the block carry (every thread computes the same values)
        for (int w = 0; w < 4; ++w)
            if (s_has[w]) {
                if (!blk_has) blk_has = true;
                prev_known = true; prev1 = s_o1[w]; prev2 = s_o2[w];
                last1 = prev1; last2 = prev2;
            }
        emit_base += sub_total;
        __syncthreads();   // s_has / s_o* / s_cnt are rewritten by the next sub-tile
    }

    // ---- block epilogue: counters and summary ---------------------------------------------------------
    int vals[7] = {c_count, c_nonuniq, c_nus, c_dup, c_long, c_fishy, c_reach};
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int v = wave_sum(vals[j]);
        if (lane == 0) s_red[wave][j] = v;
    }
    __syncthreads();
    if (t < 7) {
        const long long v = (long long)s_red[0][t] + s_red[1][t] + s_red[2][t] + s_red[3][t];
        // besst_counters field order: count, non_unique, non_unique_for_scaf, nr_of_duplicates,
        // reads_with_too_long_insert, fishy_reads, n_tuples, n_reach
        const int field = t < 6 ? t : 7;
        if (v) atomicAdd(&counters[field], (unsigned long long)v);
    }
    if (t == 0) {
        BlockSummary s;
        s.n_emit = (uint32_t)emit_base;
        s.has_reach = blk_has ? 1u : 0u;
        s.first_o1 = s_head[4] ? s_head[0] : 0;
        s.first_o2 = s_head[4] ? s_head[1] : 0;
        s.last_o1 = last1;
        s.last_o2 = last2;
        s.head_info = s_head[4] ? (uint32_t)s_head[2] : 0u;
        s.head_slot = (uint32_t)s_head[3];
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
    __shared__ int s_redc[16][4];
    const int t = threadIdx.x;
    if (t == 0) { s_carry[0] = carry[0]; s_carry[1] = carry[1]; s_base = 0; }
    __syncthreads();
    int c_count = 0, c_long = 0, c_dup = 0, c_nus = 0;
    for (uint32_t c0 = 0; c0 < nblocks; c0 += 1024) {
        const uint32_t b = c0 + t;
        BlockSummary s;
        s.n_emit = 0; s.has_reach = 0; s.first_o1 = s.first_o2 = s.last_o1 = s.last_o2 = 0;
        s.head_info = 0; s.head_slot = kNoSlot;
        if (b < nblocks) s = summ[b];
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
    int vals[4] = {c_count, c_nus, c_dup, c_long};
    const int lane = t & 63, wave = t >> 6;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int v = wave_sum(vals[j]);
        if (lane == 0) s_redc[wave][j] = v;
    }
    __syncthreads();
    if (t < 4) {
        long long v = 0;
        for (int w = 0; w < 16; ++w) v += s_redc[w][t];
        // fields: count(0), non_unique_for_scaf(2), nr_of_duplicates(3), reads_with_too_long_insert(4)
        const int field = t == 0 ? 0 : t == 1 ? 2 : t == 2 ? 3 : 4;
        if (v) atomicAdd(&counters[field], (unsigned long long)v);
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
                                                      uint64_t* __restrict__ payload) {
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
    ProfScope ps(s, kProfClassify);
    hipLaunchKernelGGL(classify_kernel, dim3(nblocks), dim3(kClsThreads), 0, s, a,
                       reinterpret_cast<unsigned long long*>(aligned), w.seg_keys, w.seg_payload, w.summ,
                       reinterpret_cast<unsigned long long*>(counters));
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
                         size_t ws_bytes) {
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
                           w.seg_payload, keys, payload);
    }
    BESST_HIP_TRY(hipGetLastError());
    return BESST_OK;
}

int launch_classify(hipStream_t s, const ClassifyArgs& a, int32_t* carry, int64_t* aligned, uint64_t* keys,
                    uint64_t* payload, uint32_t* n_out, besst_counters* counters, void* ws, size_t ws_bytes) {
    int rc = launch_classify_scan(s, a, aligned, counters, ws, ws_bytes);
    if (rc) return rc;
    return launch_classify_emit(s, a.n, a.detect_dup, carry, keys, payload, n_out, counters, ws, ws_bytes);
}

}  // namespace besst
