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

#include <type_traits>

#include "common.h"

namespace besst {

namespace {

constexpr uint32_t kNoSlot = 0xffffffffu;

// evaluation bits of one record
constexpr uint32_t EV_COV = 1u, EV_FISHY = 2u, EV_NONUNIQ = 4u, EV_REACH = 8u,
                   EV_DOUBLE = 32u, EV_MAPQ0 = 64u, EV_CASEA = 128u, EV_FIRSTMIN = 256u;

// obs1 + obs2 < ins_size_threshold for two observations above 25 (so their sum is a whole number in (50, 2^32)) as an
// unsigned compare against ceil(threshold); the threshold is below 2^30 (api.hip), one that is not positive accepts nothing
__device__ __forceinline__ uint32_t ins_thr_u32(const ClassifyArgs& a) {
    return a.ins_thr_int <= 0 ? 0u : (uint32_t)a.ins_thr_int;
}

struct Eval {
    uint32_t bits;
    int32_t o1, o2;
    uint32_t n_min, n_max;   // node codes (scaffold * 2 + side) of the edge this record may support
};

// One end of PosDirCalculatorPE / PosDirCalculatorMP (CreateGraph.py:1024-1076).  The MP calculator is
// the PE one with the read strand inverted.  read_len may be fractional: Python evaluates
// `cpos + rpos + read_len` and `slen - cpos - (clen - rpos - read_len)` left to right in float and
// truncates with int(); the same two roundings are done here in fp64 (built with -ffp-contract=off).
// rl_int >= 0: read_len is that whole number - every intermediate is an integer below 2^53, the float evaluation is
// exact and the same sums are taken in int64 (an fp64 instruction costs the SIMD two to four integer ones, and the
// record loop of a mate-pair library is bound by instruction issue).
__device__ __forceinline__ void posdir(bool rf, bool cdir, bool rdir, int32_t cpos, int32_t rpos,
                                       int32_t slen, int32_t clen, double read_len, int64_t rl_int, int32_t& obs,
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
        if (rl_int >= 0) {                                   // uniform
            // (wrapping 32-bit sums: the low 32 bits of the int64 result, which is what the cast keeps)
            const uint32_t rl = (uint32_t)rl_int;
            const uint32_t v = cdir ? (uint32_t)cpos + (uint32_t)rpos + rl
                                    : ((uint32_t)slen - (uint32_t)cpos) - (((uint32_t)clen - (uint32_t)rpos) - rl);
            obs = (int32_t)v;
            side = cdir ? 0 : 1;
        } else if (cdir) {
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
    posdir(rf, dir1, rdir, c1.ctg_pos, pos, c1.scaf_len, c1.ctg_len, a.read_len, a.read_len_int, e.o1, s1);
    posdir(rf, dir2, mdir, c2.ctg_pos, mpos, c2.scaf_len, c2.ctg_len, a.read_len, a.read_len_int, e.o2, s2);
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
//     different scaffolds (:141-169); its bit is published and ordered_kernel does the rest,
//     including the candidate's own coverage contribution.
// The stream is (tid,pos)-sorted, so the 256 records of a wave normally share one tid: coverage is a wave
// reduction accumulated across the workgroup's sub-tiles and flushed with one 64-bit atomic per run.
// ---------------------------------------------------------------------------------------------------------
// Add val to dst[key] with ONE atomic per run of equal keys in the wave (segmented scan over the run heads; exact
// for any key order, one atomic per contig for a sorted stream).  Lanes with val == 0 only carry their key.
__device__ __forceinline__ void wave_add_runs(unsigned long long* dst, int32_t key, int val, int lane) {
    const int32_t up = __shfl_up(key, 1, 64);
    const bool head = lane == 0 || key != up;
    const unsigned long long heads = __ballot(head);
    const int start = 63 - __clzll((long long)(heads & (~0ull >> (63 - lane))));
    int v = val;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(v, d, 64);
        if (lane - d >= start) v += o;
    }
    const bool tail = lane == 63 || ((heads >> (lane + 1)) & 1ull);
    if (tail && v) atomicAdd(&dst[key], (unsigned long long)v);
}

// kBits: the records' mate-elsewhere bits are there (ClassifyArgs::mate_bits) - a full tile then reads 7 1/8 bytes per
// record instead of 11: `mtid` is not read at all here (ordered_kernel gathers the candidates' own).
template <bool kBits>
__global__ __launch_bounds__(kStreamThreads) void stream_kernel(ClassifyArgs a,
                                                                unsigned long long* __restrict__ aligned,
                                                                unsigned long long* __restrict__ bitmask) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t block_base = (int64_t)blockIdx.x * kStreamTile;
    int4 v_tid[kStreamSubTiles], v_mtid[kStreamSubTiles];
    uchar4 v_mapq[kStreamSubTiles];
    ushort4 v_qlen[kStreamSubTiles];
    uint32_t v_bits[kStreamSubTiles];                        // kBits: the lane's four bits (full tiles)
    const bool full_tile = block_base + kStreamTile <= a.n;
    if (full_tile) {
        // full tile: issue every load of the workgroup's 4096 records before touching any of them
#pragma unroll
        for (int st = 0; st < kStreamSubTiles; ++st) {
            const int64_t i0 = block_base + (int64_t)st * kStreamSubTile + (int64_t)t * kStreamVec;
#ifndef BESST_STREAM_PLAIN_LOADS
            {   // (non-temporal: every record is read once; C2's step 111 -> 108 us)
                typedef int v4i __attribute__((ext_vector_type(4)));
                typedef unsigned short v4h __attribute__((ext_vector_type(4)));
                const v4i t4 = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(a.tid + i0));
                v4i m4 = t4;
                if constexpr (kBits) {
                    // (i0 is a multiple of 4: the lane's four bits are one nibble of byte i0 / 8)
                    v_bits[st] = ((uint32_t)a.mate_bits[i0 >> 3] >> (uint32_t)(i0 & 4)) & 15u;
                } else {
                    m4 = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(a.mtid + i0));
                }
                const uint32_t q1 = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(a.mapq + i0));
                const v4h q4 = __builtin_nontemporal_load(reinterpret_cast<const v4h*>(a.qlen + i0));
                v_tid[st] = make_int4(t4.x, t4.y, t4.z, t4.w);
                v_mtid[st] = make_int4(m4.x, m4.y, m4.z, m4.w);
                v_mapq[st] = make_uchar4(q1 & 255u, (q1 >> 8) & 255u, (q1 >> 16) & 255u, q1 >> 24);
                v_qlen[st] = make_ushort4(q4.x, q4.y, q4.z, q4.w);
            }
#else
            v_tid[st] = *reinterpret_cast<const int4*>(a.tid + i0);
            v_mtid[st] = *reinterpret_cast<const int4*>(a.mtid + i0);
            v_mapq[st] = *reinterpret_cast<const uchar4*>(a.mapq + i0);
            v_qlen[st] = *reinterpret_cast<const ushort4*>(a.qlen + i0);
            if constexpr (kBits) v_bits[st] = ((uint32_t)a.mate_bits[i0 >> 3] >> (uint32_t)(i0 & 4)) & 15u;
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
            if constexpr (kBits) cand[k] = full_tile ? ((v_bits[st] >> k) & 1u) != 0u : r_tid[k] != r_mtid[k];
            else cand[k] = r_tid[k] != r_mtid[k];
            uni = uni && (r_tid[k] == ref);
            const bool cov = ((int32_t)r_mapq[k] >= a.min_mapq || r_mapq[k] == 0);
            if (!cand[k] && cov) mine += (int)r_qlen[k];
        }
        // The wave's coverage sum is NOT added here: it is left as (contig, sum) next to the candidate bits, and
        // ordered_kernel - one lane per group - adds the runs of equal contigs of its 64 groups.  An atomic per wave
        // was harmless on C2 (8 per contig) but serialised at ~0.3 us each when a small genome puts a thousand
        // waves on every contig (60 contigs / 16 M records: 0.32 ms for this pass instead of 0.04).
        int2 part = make_int2(-1, 0);
        if (__all(uni)) {
            part = make_int2(ref, wave_sum(mine));
        } else {
            // a contig boundary (or unsorted input) inside the wave.  Lanes own 4 consecutive records: a lane whose
            // records share one contig contributes (contig, sum) to a run-segmented wave reduction - one atomic per
            // run of lanes, i.e. per contig of a sorted stream; a lane that itself straddles a boundary adds its
            // records directly and breaks the run.  (Fragmented assemblies put a dozen contigs in every wave: the
            // previous per-slot keyed reduction with its per-lane fallback ran this pass at 0.6 TB/s there.)
            const bool lane_uni = r_tid[0] == r_tid[1] && r_tid[0] == r_tid[2] && r_tid[0] == r_tid[3];
            int32_t key = (int32_t)(0x80000000u | (uint32_t)lane);       // unique: never equal to a neighbour's key
            int val = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool cov = ((int32_t)r_mapq[k] >= a.min_mapq || r_mapq[k] == 0);
                const bool act = !cand[k] && cov && (uint32_t)r_tid[k] < (uint32_t)a.n_contigs;
                if (!act) continue;
                if (lane_uni) val += (int)r_qlen[k];
                else atomicAdd(&aligned[r_tid[k]], (unsigned long long)r_qlen[k]);
            }
            if (lane_uni) key = r_tid[0];
            wave_add_runs(aligned, key, val, lane);
        }
        // candidate bits of this wave's group: word k, bit l  <->  record group_base + 4*l + k
        const unsigned long long b0 = __ballot(cand[0]), b1 = __ballot(cand[1]);
        const unsigned long long b2 = __ballot(cand[2]), b3 = __ballot(cand[3]);
        const int64_t g = ((int64_t)blockIdx.x * kStreamSubTiles + st) * 4 + wave;
        // group record: 6 x u64 = 4 candidate words, (contig, coverage sum), padding - one store instruction
        if (lane < 5) {
            const unsigned long long cw = (unsigned long long)(uint32_t)part.x | ((unsigned long long)(uint32_t)part.y << 32);
            bitmask[g * kGroupWords + lane] = lane == 0 ? b0 : lane == 1 ? b1 : lane == 2 ? b2 : lane == 3 ? b3 : cw;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// ordered_kernel: everything a candidate needs, evaluation included.  One 4-wave workgroup per kCandGroups
// consecutive groups.  Evaluation is order free and spread over the waves, 64 candidates per wave and chunk,
// kAhead chunks in flight: r-th set bit of the group's mask -> record gather (7 columns) -> two contig rows ->
// observations (PosDirCalculator) -> 16-byte entry in LDS.  Wave 0 then walks the entries in stream order for
// the order-dependent part: previous reaching observation via ballot, CreateEdge semantics (acceptance rule
// :840), ordered slots for the emitted tuples; the chain state lives in its registers.
// History on C2 (20 M records, 306 k candidates): a separate evaluation launch with one wave per group cost
// ~N/256 mostly idle waves and a staging round trip (31 + 18 us); single-wave workgroups doing both were set
// by their busiest block's serial rounds (31 us at 8192 records per block); this version takes 21 us.
// Handing tid/mtid/mapq/qlen over from stream_kernel instead of re-gathering them did not help here and cost
// the streaming pass 2-3 us.
// ---------------------------------------------------------------------------------------------------------
// r-th candidate (record order) of a group: word k, bit l <-> record 4*l + k.  Returns the record's offset in
// the group.
__device__ __forceinline__ int select_candidate(unsigned long long w0, unsigned long long w1, unsigned long long w2,
                                                unsigned long long w3, int r) {
    int lo = 0;                                         // largest l with (#candidates in lanes < l) <= r
#pragma unroll
    for (int step = 32; step > 0; step >>= 1) {
        const int mid = lo + step;
        const unsigned long long m = (1ull << mid) - 1ull;
        const int below = __popcll(w0 & m) + __popcll(w1 & m) + __popcll(w2 & m) + __popcll(w3 & m);
        if (below <= r) lo = mid;
    }
    const unsigned long long m = (1ull << lo) - 1ull;
    int rest = r - (__popcll(w0 & m) + __popcll(w1 & m) + __popcll(w2 & m) + __popcll(w3 & m));
    const int b[4] = {(int)((w0 >> lo) & 1ull), (int)((w1 >> lo) & 1ull), (int)((w2 >> lo) & 1ull),
                      (int)((w3 >> lo) & 1ull)};
    int k = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (b[q]) {
            if (rest == 0) { k = q; rest = -1; }
            else if (rest > 0) --rest;
        }
    }
    return lo * 4 + k;
}

// The order-dependent part of the record loop, run by ONE wave over evaluated entries in stream order, 64 at a time:
// previous reaching observation via ballot, CreateEdge semantics (acceptance rule :840), ordered slots for the
// emitted tuples in the block's segment.  The state is identical in every lane.  Entry layout:
//   { obs1, obs2, node_min | REACH<<29 | FISHY<<30 | NONUNIQ<<31, node_max | MAPQ0<<29 | CASEA<<30 | FIRSTMIN<<31 }
struct Chain {
    bool prev_known = false;
    int32_t prev1 = 0, prev2 = 0;
    bool blk_has = false;
    int emit_base = 0;
    bool head_present = false;
    int32_t head1 = 0, head2 = 0;
    uint32_t head_info = 0, head_slot = kNoSlot;
    int c_count = 0, c_nonuniq = 0, c_nus = 0, c_dup = 0, c_long = 0, c_fishy = 0, c_reach = 0;

    __device__ __forceinline__ void step(const ClassifyArgs& a, const uint4 ent, const bool live, const int lane,
                                         uint64_t* __restrict__ seg_keys, uint64_t* __restrict__ seg_payload,
                                         const int64_t block_base) {
        const unsigned long long lt_mask = (1ull << lane) - 1ull;
        const uint32_t thr_u32 = ins_thr_u32(a);
        const uint32_t mask_a = a.no_score ? BESST_MASK_GPRIME : (BESST_MASK_G | (a.extend_paths ? BESST_MASK_GPRIME : 0u));
        const int32_t o1 = (int32_t)ent.x, o2 = (int32_t)ent.y;
        const bool reach = live && ((ent.z >> 29) & 1u), fishy = live && ((ent.z >> 30) & 1u);
        const bool nonuniq = live && ((ent.z >> 31) & 1u);
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
        const bool accept = reach && o1 > 25 && o2 > 25 && ((uint32_t)o1 + (uint32_t)o2 < thr_u32);
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

    // lane f publishes plane f of the block's summary
    __device__ __forceinline__ void publish(SummView summ, const uint32_t block, const int lane) {
        int tot[7];
        const int vals[7] = {c_count, c_nonuniq, c_nus, c_dup, c_long, c_fishy, c_reach};
#pragma unroll
        for (int f = 0; f < 7; ++f) tot[f] = wave_sum(vals[f]);
        uint32_t v = 0;
        if (lane == kSumEmit) v = (uint32_t)emit_base;
        if (lane == kSumHas) v = blk_has ? 1u : 0u;
        if (lane == kSumFirst1) v = head_present ? (uint32_t)head1 : 0u;
        if (lane == kSumFirst2) v = head_present ? (uint32_t)head2 : 0u;
        if (lane == kSumLast1) v = (uint32_t)prev1;
        if (lane == kSumLast2) v = (uint32_t)prev2;
        if (lane == kSumHeadInfo) v = head_present ? head_info : 0u;
        if (lane == kSumHeadSlot) v = head_slot;
#pragma unroll
        for (int f = 0; f < 7; ++f)
            if (lane == kSumCtr0 + f) v = (uint32_t)tot[f];
        if (lane < kSumPlanes) summ.at(lane, block) = v;
    }
};

__device__ __forceinline__ uint4 pack_entry(const Eval& e) {
    uint4 ent;
    ent.x = (uint32_t)e.o1;
    ent.y = (uint32_t)e.o2;
    ent.z = e.n_min | ((e.bits & EV_REACH) ? 1u << 29 : 0u) | ((e.bits & EV_FISHY) ? 1u << 30 : 0u) |
            ((e.bits & EV_NONUNIQ) ? 1u << 31 : 0u);
    ent.w = e.n_max | ((e.bits & EV_MAPQ0) ? 1u << 29 : 0u) | ((e.bits & EV_CASEA) ? 1u << 30 : 0u) |
            ((e.bits & EV_FIRSTMIN) ? 1u << 31 : 0u);
    return ent;
}

// Occupancy over prefetch depth: two chunks in flight and a register cap for five workgroups per CU (the dense
// mate-pair case is bound by gathers in flight: 0.43 -> 0.37 ms on a C3 slice; C2 is indifferent).
#ifndef BESST_ORD_MIN_BLOCKS
#define BESST_ORD_MIN_BLOCKS 5
#endif
constexpr int kOrdWaves = 4;                              // waves of an ordered_kernel workgroup
constexpr int kOrdThreads = kOrdWaves * 64;
#ifndef BESST_ORD_AHEAD
#define BESST_ORD_AHEAD 2
#endif
constexpr int kAhead = BESST_ORD_AHEAD;                                 // chunks a wave evaluates per round, all gathers in flight
constexpr int kOrdRound = kOrdWaves * kAhead * 64;        // candidates per round (1024): 16 KB of entries in LDS

__global__ __launch_bounds__(kOrdThreads, BESST_ORD_MIN_BLOCKS) void ordered_kernel(
    ClassifyArgs a, const unsigned long long* __restrict__ bitmask, int64_t n_groups,
    unsigned long long* __restrict__ aligned, uint64_t* __restrict__ seg_keys,
    uint64_t* __restrict__ seg_payload, SummView summ) {
    __shared__ int s_pre[kCandThreads + 1];
    __shared__ unsigned long long s_bits[kCandThreads * 4];
    // { obs1, obs2, node_min | REACH<<29 | FISHY<<30 | NONUNIQ<<31, node_max | MAPQ0<<29 | CASEA<<30 | FIRSTMIN<<31 }
    __shared__ uint4 s_ent[kOrdRound];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t block_base = (int64_t)blockIdx.x * kClsTile;
    const int64_t g0 = (int64_t)blockIdx.x * kCandGroups;
    if (wave == 0) {
        const int64_t g = g0 + lane;
        int cnt = 0;
        ulonglong2 w0 = make_ulonglong2(0ull, 0ull), w1 = w0;
        int2 part = make_int2(-1, 0);
        if (lane < kCandGroups && g < n_groups) {
            w0 = *reinterpret_cast<const ulonglong2*>(bitmask + g * kGroupWords);
            w1 = *reinterpret_cast<const ulonglong2*>(bitmask + g * kGroupWords + 2);
            const unsigned long long cw = bitmask[g * kGroupWords + 4];
            part = make_int2((int32_t)(uint32_t)cw, (int32_t)(uint32_t)(cw >> 32));
            cnt = __popcll(w0.x) + __popcll(w0.y) + __popcll(w1.x) + __popcll(w1.y);
        }
        // coverage of the tid == mtid records, left by stream_kernel per group: one atomic per run of equal contigs
        // (any in-range tid is credited; compact_kernel clears the entries of contigs that are not in the table)
        wave_add_runs(aligned, part.x, (uint32_t)part.x < (uint32_t)a.n_contigs ? part.y : 0, lane);
        s_bits[lane * 4 + 0] = w0.x; s_bits[lane * 4 + 1] = w0.y;
        s_bits[lane * 4 + 2] = w1.x; s_bits[lane * 4 + 3] = w1.y;
        const int incl = wave_incl_scan(cnt, lane);
        s_pre[lane] = incl - cnt;
        if (lane == 63) s_pre[kCandThreads] = incl;
    }
    __syncthreads();
    const int total = s_pre[kCandThreads];

    Chain chain;                                         // wave 0 only

    for (int c0 = 0; c0 < total; c0 += kOrdRound) {
        // ---- evaluation, order free: chunk q = u * kOrdWaves + wave of the round goes to this wave's slot u, so a
        // round with few candidates still spreads over the waves
        {
            int32_t r_tid[kAhead], r_mtid[kAhead], r_pos[kAhead], r_mpos[kAhead];
            uint32_t r_flag[kAhead], r_mapq[kAhead], r_qlen[kAhead];
            bool valid[kAhead];
#pragma unroll
            for (int u = 0; u < kAhead; ++u) {              // record gathers of every chunk first ...
                const int j = c0 + (u * kOrdWaves + wave) * 64 + lane;
                valid[u] = j < total;
                r_tid[u] = r_mtid[u] = -1;
                r_pos[u] = r_mpos[u] = 0;
                r_flag[u] = r_mapq[u] = r_qlen[u] = 0;
                if (valid[u]) {
                    int lo = 0, hi = kCandThreads;          // last group whose exclusive prefix is <= j
                    while (hi - lo > 1) {
                        const int mid = (lo + hi) >> 1;
                        if (s_pre[mid] <= j) lo = mid; else hi = mid;
                    }
                    const int off = select_candidate(s_bits[lo * 4], s_bits[lo * 4 + 1], s_bits[lo * 4 + 2],
                                                     s_bits[lo * 4 + 3], j - s_pre[lo]);
                    const int64_t i = (g0 + lo) * kGroup + off;
                    r_tid[u] = a.tid[i];
                    r_mtid[u] = a.mtid[i];
                    r_pos[u] = a.pos[i];
                    r_mpos[u] = a.mpos[i];
                    r_flag[u] = a.flag[i];
                    r_mapq[u] = a.mapq[i];
                    r_qlen[u] = a.qlen[i];
                }
            }
            ContigRow c1[kAhead], c2[kAhead];
            bool in_range[kAhead];
#pragma unroll
            for (int u = 0; u < kAhead; ++u) {              // ... then their contig rows ...
                in_range[u] = valid[u] && (uint32_t)r_tid[u] < (uint32_t)a.n_contigs &&
                              (uint32_t)r_mtid[u] < (uint32_t)a.n_contigs;
                if (in_range[u]) {
                    c1[u] = a.table[r_tid[u]];
                    c2[u] = a.table[r_mtid[u]];
                }
            }
#pragma unroll
            for (int u = 0; u < kAhead; ++u) {              // ... then the arithmetic
                const int q = u * kOrdWaves + wave;
                if (c0 + q * 64 >= total) break;             // uniform in the wave
                const Eval e = eval_record(a, in_range[u], c1[u], c2[u], r_tid[u], r_mtid[u], r_pos[u], r_mpos[u],
                                           r_flag[u], r_mapq[u]);
                const uint4 ent = pack_entry(e);
                s_ent[q * 64 + lane] = ent;
                // the candidates' own coverage (the streaming pass credits only tid == mtid records)
                wave_add_runs(aligned, r_tid[u], (e.bits & EV_COV) ? (int)r_qlen[u] : 0, lane);
            }
        }
        __syncthreads();
        // ---- the order-dependent part, wave 0 over the round's entries in stream order
        if (wave == 0) {
            const int round_n = total - c0 < kOrdRound ? total - c0 : kOrdRound;
            for (int q0 = 0; q0 < round_n; q0 += 64)         // entries past `total` were not written: masked by `live`
                chain.step(a, s_ent[q0 + lane], q0 + lane < round_n, lane, seg_keys, seg_payload, block_base);
        }
        if (c0 + kOrdRound < total) __syncthreads();         // the next round overwrites the entries (uniform)
    }
    if (wave != 0) return;
    chain.publish(summ, blockIdx.x, lane);
}

// ---------------------------------------------------------------------------------------------------------
// fused_wave_kernel: the record loop of a candidate-DENSE library (mate pairs: a fifth of the records have their mate
// on another contig) in ONE pass over the seven columns.  stream_kernel + ordered_kernel read such a library twice - the
// candidates' scattered column reads re-read the whole record set - and ordered_kernel spends most of its instructions
// locating the r-th set bit of a group mask.  Here a block of 16 384 records belongs to ONE WAVE (a single-wave workgroup,
// no barrier anywhere) that walks it in 64 sub-tiles of 256 records:
//   coalesced loads of the seven columns; tid == mtid records only add coverage (a running (contig, sum), flushed with
//   one atomic when the contig changes); candidates are compacted, in stream order, into the wave's own LDS ring and
//   evaluated 64 at a time: lane j evaluates candidate j - two contig rows (L2), PosDirCalculator, link dispatch -, then
//   the order-dependent part (duplicate chain, acceptance, counters, ordered emission into the block's segment) as
//   ballots over wave-uniform (scalar) state: prev_obs, emission base, head.
// It publishes the same block summary as ordered_kernel, so stitch_kernel / compact_kernel and the sharded build's head /
// tail logic are shared by both paths.  besst_lib_params.record_path selects the path (the host samples the candidate
// density).  (Round 2's form of this pass - four waves per block that met at ~65 workgroup barriers per block and passed
// the chain's state through LDS: whenever one wave waited for memory the other three waited with it, 1.74 ms on full C3
// against 1.45 here - was removed in round 4.)
// ---------------------------------------------------------------------------------------------------------
constexpr int kFwSub = 256;                              // records per sub-tile: four per lane
constexpr int kFwRing = kFwSub + 64;                     // a sub-tile's worth of candidates + an unfinished round
// (the loop below keeps two sub-tiles of column loads and one round of contig rows in flight: 123 VGPRs.  Forced
// into the 96 of five waves per SIMD it spills and is 3 % slower than at four.)
#ifndef BESST_FW_MIN_WAVES
#define BESST_FW_MIN_WAVES 4
#endif

__device__ __forceinline__ int wave_sum_dpp(int v) {     // the sum in every lane?  no: wave-uniform, read from lane 63
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return __builtin_amdgcn_readlane(v, 63);
}

// kRuns (round 4): the wave also GROUPS what it emits (RunLayout, common.h) - stage 2 then reads neither keys nor tuples
// back to find the runs of equal keys (rg_group_kernel: 0.68 GB read + 0.34 GB written per C3 step), it gets a run byte per
// tuple and one table entry per run.  Per evaluation round the distinct keys among the <= 64 emitted tuples (two or three in
// a coordinate-sorted stream) are peeled off - first emitting lane's key, one compare + ballot for the round's tuples, one
// for the open runs the lanes hold - so the cost is per distinct key and round, not per tuple; no LDS, three registers.
// kBits: ClassifyArgs::mate_bits is there - `mtid` moves from the columns every record is read from to the columns that only
// the lanes with a candidate read (pos / mpos / flag): 4 of a record's 11 always-read bytes become 1/8.
template <bool kRuns, bool kBits>
__global__ __launch_bounds__(64, BESST_FW_MIN_WAVES) void fused_wave_kernel(
    ClassifyArgs a, unsigned long long* __restrict__ aligned, uint64_t* __restrict__ seg_keys,
    uint64_t* __restrict__ seg_payload, SummView summ, uint32_t* __restrict__ run_status) {
    __shared__ uint32_t s_buf[5][kFwRing];                // tid, mtid, pos, mpos, flag | mapq << 16 of the queued candidates
    __shared__ unsigned short s_qlen[kFwRing];
    __shared__ uint32_t s_ph[kRuns ? 1 : 256];            // the sort's two digit histograms, 16 bits per counter (<= 16 384 tuples)
    const int lane = threadIdx.x;
    const int64_t block_base = (int64_t)blockIdx.x * kClsTile;
    // (lanes below this one in a ballot: v_mbcnt_lo/hi on the mask - a (1 << lane) - 1 kept in registers for the whole
    // block cost two of the 128 a wave has at four per SIMD, and this loop spills)
    auto below_cnt = [](unsigned long long m) {
        return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    };
    const uint32_t thr_u32 = ins_thr_u32(a);
    const bool all_present = a.cls8[a.n_contigs] != 0;    // (uniform: the byte behind the class bytes, besst_dev_pack_contigs)
    if (!kRuns && a.ps_table) {                           // uniform
#pragma unroll
        for (int q = 0; q < 4; ++q) s_ph[q * 64 + lane] = 0;
    }
    // ---- kRuns: the open chunk (uniform: its first slot, its tuples, its runs; s_rk: the keys of its runs, a hash table
    // whose slot number IS the run's name inside the chunk) and the block's totals
    __shared__ unsigned long long s_rk[kRuns ? kRlSlots : 1];
    __shared__ uint32_t s_rn[kRuns ? kRlSlots : 1], s_rf[kRuns ? kRlSlots : 1];   // tuples and first slot of each run
    char* const rl_region = reinterpret_cast<char*>(seg_keys + block_base);
    int rl_start = 0, rl_n = 0, rl_k = 0, rl_chunks = 0, rl_runs = 0, rl_over = 0;
    if constexpr (kRuns) {
#pragma unroll
        for (int q = 0; q < kRlSlots / 64; ++q) {
            s_rk[q * 64 + lane] = kRlEmpty;
            s_rn[q * 64 + lane] = 0u;
            s_rf[q * 64 + lane] = 0xffffu;
        }
    }
    auto close_chunk = [&]() {
        if (rl_n == 0) return;                               // uniform
        // the slots in use, in slot order, as a dense list of rl_k entries: key, tuples | first slot << 16, slot
        int before = 0;
#pragma unroll
        for (int q = 0; q < kRlSlots / 64; ++q) {
            const unsigned long long k = s_rk[q * 64 + lane];
            const uint32_t meta = s_rn[q * 64 + lane] | (s_rf[q * 64 + lane] << 16);
            const unsigned long long used = __ballot(k != kRlEmpty);
            if (k != kRlEmpty && rl_chunks < kRlMaxChunks) {
                const int e = rl_chunks * kRlSlots + before + below_cnt(used);
                reinterpret_cast<unsigned long long*>(rl_region + kRlKeys)[e] = k;
                reinterpret_cast<uint32_t*>(rl_region + kRlMeta)[e] = meta;
                reinterpret_cast<uint8_t*>(rl_region + kRlOrd)[e] = (uint8_t)(q * 64 + lane);
            }
            before += __popcll(used);
            s_rk[q * 64 + lane] = kRlEmpty;
            s_rn[q * 64 + lane] = 0u;
            s_rf[q * 64 + lane] = 0xffffu;
        }
        if (rl_chunks < kRlMaxChunks) {
            if (lane == 0)
                reinterpret_cast<uint64_t*>(rl_region + kRlHdr)[rl_chunks] =
                    (uint64_t)(uint32_t)rl_start | ((uint64_t)(uint32_t)rl_n << 16) | ((uint64_t)(uint32_t)rl_k << 32);
        } else {
            rl_over = 1;
        }
        ++rl_chunks;
        rl_runs += rl_k;
        rl_start += rl_n;
        rl_n = 0;
        rl_k = 0;
    };
    // the block's share of the counters, two per register (a lane sees at most 257 rounds and 64 sub-tiles: < 2^16 each);
    // the records that reached CreateEdge are counted from the ballot the chain needs anyway
    uint32_t c_count_nus = 0, c_dup_long = 0, c_nonuniq_fishy = 0;
    int c_reach_u = 0;
    int32_t run_tid = -1;                                 // the wave's running coverage (uniform)
    int run_sum = 0;
    int q_head = 0, q_count = 0;                          // the queue of candidates not yet evaluated (uniform)
    // the chain's state (uniform): the last record of the block so far that reached CreateEdge, where the next tuple goes,
    // and the block's first reaching record (the one stitch_kernel resolves against the blocks before)
    int st_known = 0, st_p1 = 0, st_p2 = 0, st_emit = 0;
    int hd_present = 0, hd_o1 = 0, hd_o2 = 0, hd_info = 0, hd_slot = (int)kNoSlot;
    // a rare coverage correction: the index is taken inside the branch (an empty asm hides it from the scheduler) - hoisted
    // above it, its sign-extended 64-bit form was kept alive for every record of a sub-tile and spilled to scratch, whose
    // stores then sat in the same in-order queue as the loop's prefetches
    auto cov_add = [&](int32_t t, unsigned long long v) {
        asm volatile("" : "+v"(t));
        atomicAdd(&aligned[t], v);
    };
    // An evaluation round in two halves: round_fetch takes up to 64 candidates off the queue into registers and issues
    // the gathers of their two contig rows; round_finish does the rest once the rows are there.  In the pipelined loop
    // below the gathers of a round travel with the column loads of the sub-tiles ahead and are waited for together
    // (vmcnt counts in order: a wait for a gather issued behind a prefetch would wait for the prefetch as well).
    struct Round {
        int cnt;
        int32_t tid, mtid, pos, mpos;
        uint32_t fm, qlen;
        ContigRow c1, c2;
    };
    auto round_fetch = [&](const int cnt) {
        Round r;
        r.cnt = cnt;
        r.tid = -1; r.mtid = -1; r.pos = 0; r.mpos = 0; r.fm = 0; r.qlen = 0;
        if (lane < cnt) {
            int j = q_head + lane;
            j = j >= kFwRing ? j - kFwRing : j;
            r.tid = (int32_t)s_buf[0][j]; r.mtid = (int32_t)s_buf[1][j];
            r.pos = (int32_t)s_buf[2][j]; r.mpos = (int32_t)s_buf[3][j];
            r.fm = s_buf[4][j];
            r.qlen = s_qlen[j];
        }
        // (every lane gathers - row 0 where it has nothing to look up -, so the number of loads in flight does not
        // depend on the data)
        const bool in_range = lane < cnt && (uint32_t)r.tid < (uint32_t)a.n_contigs && (uint32_t)r.mtid < (uint32_t)a.n_contigs;
        r.c1 = a.table[in_range ? r.tid : 0];
        r.c2 = a.table[in_range ? r.mtid : 0];
        q_head += cnt;
        q_head = q_head >= kFwRing ? q_head - kFwRing : q_head;
        q_count -= cnt;
        return r;
    };
    auto round_finish = [&](const Round& r) {
        const int cnt = r.cnt;
        const bool live = lane < cnt;
        const int32_t tid = r.tid, mtid = r.mtid, pos = r.pos, mpos = r.mpos;
        const uint32_t fm = r.fm, qlen = r.qlen;
        const bool in_range = live && (uint32_t)tid < (uint32_t)a.n_contigs && (uint32_t)mtid < (uint32_t)a.n_contigs;
        const Eval e = eval_record(a, in_range, r.c1, r.c2, tid, mtid, pos, mpos, fm & 0xffffu, fm >> 16);
        // (the coverage credited with the streamed records is taken back when a contig is not in the table)
        if (in_range && !(e.bits & EV_COV) && ((int32_t)(fm >> 16) >= a.min_mapq || (fm >> 16) == 0u))
            cov_add(tid, 0ull - (unsigned long long)qlen);
        const bool reach = live && (e.bits & EV_REACH), fishy = live && (e.bits & EV_FISHY);
        const bool mapq0 = (e.bits & EV_MAPQ0) != 0, case_a = (e.bits & EV_CASEA) != 0;
        const bool dbl = (e.bits & EV_DOUBLE) != 0;
        c_nonuniq_fishy += ((live && (e.bits & EV_NONUNIQ)) ? 1u : 0u) | (fishy ? 0x10000u : 0u);
        const int32_t o1 = e.o1, o2 = e.o2;
        const unsigned long long has_mask = __ballot(reach);
        c_reach_u += __popcll(has_mask);
        unsigned long long below;
        {
            int l = lane;
            asm volatile("" : "+v"(l));                      // (not hoisted: the mask is rebuilt where it is needed)
            const uint32_t lo_m = l < 32 ? (1u << l) - 1u : 0xffffffffu;
            const uint32_t hi_m = l < 32 ? 0u : (1u << (l - 32)) - 1u;
            below = has_mask & (((unsigned long long)hi_m << 32) | lo_m);
        }
        // "previous record that reached CreateEdge" [:835-838, 869-870]: the nearest reaching lane below, else the state
        bool pk = below != 0ull;
        int32_t p1, p2;
        {
            const int src = below ? 63 - __clzll((long long)below) : 0;
            p1 = __shfl(o1, src, 64);
            p2 = __shfl(o2, src, 64);
        }
        if (reach && !pk) { pk = st_known != 0; p1 = st_p1; p2 = st_p2; }
        const bool accept = reach && o1 > 25 && o2 > 25 && ((uint32_t)o1 + (uint32_t)o2 < thr_u32);
        bool emit = fishy, is_head = false;
        if (reach) {
            if (!pk) {
                is_head = true;                              // first reaching record of the block: stitch_kernel's
                emit = accept;
            } else {
                const CEDelta d = create_edge(o1, o2, p1, p2, accept, dbl, mapq0, a.detect_dup != 0);
                c_count_nus += (uint32_t)d.count | ((uint32_t)d.nus << 16);
                c_dup_long += (uint32_t)d.dup | ((uint32_t)d.too_long << 16);
                emit = d.keep;
            }
        }
        const unsigned long long emit_mask = __ballot(emit);
        const int slot = st_emit + below_cnt(emit_mask);
        const uint32_t mask_a = a.no_score ? BESST_MASK_GPRIME : (BESST_MASK_G | (a.extend_paths ? BESST_MASK_GPRIME : 0u));
        const uint32_t mask = case_a ? mask_a : BESST_MASK_GPRIME;
        const bool first_min = (e.bits & EV_FIRSTMIN) != 0;
        const uint32_t lo = fishy ? 0u : (uint32_t)(first_min ? o1 : o2);
        const uint32_t hi = fishy ? 0u : ((uint32_t)(first_min ? o2 : o1) | (mask << 30));
        const uint64_t key = ((((uint64_t)e.n_min << a.node_bits) | e.n_max) << 1) | (fishy ? 1u : 0u);
        if constexpr (kRuns) {
            if (emit_mask) {                                 // uniform
                const int n_emit = __popcll(emit_mask);
                if (rl_k >= kRlCloseRuns || rl_n + n_emit > kRlChunkTuples) close_chunk();
                // every emitting lane looks its key up in the chunk's table - linear probing from a slot the two node
                // numbers pick; partners of one scaffold end are neighbours, so a chunk's handful of keys rarely collide -
                // and the first lane to bring a key claims a slot for it.  (Peeling the round's distinct keys off one by
                // one - first lane's key, compare + ballot against the round and against the open runs - cost 35
                // instructions per distinct key and round, all on the wave's critical path: +0.27 ms on full C3.)
                uint32_t my_run = kRlNoRun;                  // (the block's head: in no run, the stitch may still drop it)
                bool pending = emit && !is_head, fresh = false, tried = false;
                uint32_t h = (e.n_max + 5u * e.n_min + (fishy ? 64u : 0u)) & (uint32_t)(kRlSlots - 1);
                while (__ballot(pending) != 0ull) {          // one turn unless keys collide
                    if (pending) {
                        const unsigned long long cur = s_rk[h];
                        bool hit = cur == key;
                        if (cur == kRlEmpty) {
                            const unsigned long long old = atomicCAS(&s_rk[h], kRlEmpty, (unsigned long long)key);
                            fresh = old == kRlEmpty;
                            hit = fresh || old == key;
                            tried = true;
                        }
                        if (hit) { my_run = h; pending = false; }
                        else h = (h + 1u) & (uint32_t)(kRlSlots - 1);
                    }
                }
                rl_k += __popcll(__ballot(fresh));
                // the run's size and first slot for the listing pass (LDS atomics whose result nobody waits for: +0.04 ms on
                // full C3 - what the listing pass saves by not reading the run bytes back -; lanes that share a key walk
                // the same slots in step, so those that found their key's slot empty are exactly the tuples of a run that
                // begins in this round)
                if (my_run != kRlNoRun) {
                    atomicAdd(&s_rn[my_run], 1u);
                    if (tried) atomicMin(&s_rf[my_run], (uint32_t)(slot - rl_start));
                }
                if (emit) {
                    seg_payload[block_base + slot] = (uint64_t)lo | ((uint64_t)hi << 32);
                    reinterpret_cast<uint8_t*>(rl_region + kRlRid)[slot] = (uint8_t)my_run;
                    if (is_head) *reinterpret_cast<uint64_t*>(rl_region + kRlHeadKey) = key;
                }
                rl_n += n_emit;
            }
        } else if (emit) {
            seg_keys[block_base + slot] = key;
            seg_payload[block_base + slot] = (uint64_t)lo | ((uint64_t)hi << 32);
            if (a.ps_table) {                                // uniform
                const uint64_t k = key - a.ps_base;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const uint32_t d = (uint32_t)(k >> (a.ps_shift + 8 * q)) & 255u;
                    const uint32_t f = (uint32_t)__builtin_amdgcn_readfirstlane((int)d);
                    const unsigned long long m = __ballot(d == f);
                    uint32_t add = 0, at = d;
                    if (d != f) add = 1u;
                    else if (below_cnt(m) == 0) { add = (uint32_t)__popcll(m); at = f; }
                    if (add) atomicAdd(&s_ph[128 * q + (at >> 1)], add << (16 * (at & 1u)));
                }
            }
        }
        const unsigned long long head_mask = __ballot(is_head);
        if (head_mask) {                                     // uniform; at most once per block
            const int hl = __ffsll((long long)head_mask) - 1;
            hd_present = 1;
            hd_o1 = __builtin_amdgcn_readlane(o1, hl);
            hd_o2 = __builtin_amdgcn_readlane(o2, hl);
            const int info = (int)((accept ? 9u : 0u) | (dbl ? 2u : 0u) | (mapq0 ? 4u : 0u));
            hd_info = __builtin_amdgcn_readlane(info, hl);
            hd_slot = __builtin_amdgcn_readlane(accept ? slot : (int)kNoSlot, hl);
        }
        if (has_mask) {                                      // uniform
            const int last = 63 - __clzll((long long)has_mask);
            st_known = 1;
            st_p1 = __builtin_amdgcn_readlane(o1, last);
            st_p2 = __builtin_amdgcn_readlane(o2, last);
        }
        st_emit += __popcll(emit_mask);
    };
    auto run_round = [&](const int cnt) {
        const Round r = round_fetch(cnt);
        round_finish(r);
    };
    // ---- one sub-tile of 256 records, its seven columns in registers
    auto process = [&](const int32_t (&r_tid)[4], const int32_t (&r_mtid)[4], const int32_t (&r_pos)[4],
                       const int32_t (&r_mpos)[4], const uint32_t (&r_flag)[4], const uint32_t (&r_mapq)[4],
                       const uint32_t (&r_qlen)[4]) {
        // ---- coverage [:138-139], candidates included on the cheap part of their condition
        const int32_t ref = __builtin_amdgcn_readfirstlane(r_tid[0]);
        bool uni = true;
        int mine = 0;
        bool cand[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            cand[k] = r_tid[k] != r_mtid[k];
            uni = uni && (r_tid[k] == ref);
            const bool cov = ((int32_t)r_mapq[k] >= a.min_mapq || r_mapq[k] == 0);
            if (cov && (!cand[k] || (uint32_t)r_mtid[k] < (uint32_t)a.n_contigs)) mine += (int)r_qlen[k];
        }
        if (__all(uni)) {
            const int s = wave_sum_dpp(mine);
            if (ref != run_tid) {
                if (lane == 0 && run_sum && (uint32_t)run_tid < (uint32_t)a.n_contigs)
                    atomicAdd(&aligned[run_tid], (unsigned long long)run_sum);
                run_tid = ref;
                run_sum = 0;
            }
            run_sum += s;
        } else {
            const bool lane_uni = r_tid[0] == r_tid[1] && r_tid[0] == r_tid[2] && r_tid[0] == r_tid[3];
            int32_t key = (int32_t)(0x80000000u | (uint32_t)lane);
            int val = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool cov = ((int32_t)r_mapq[k] >= a.min_mapq || r_mapq[k] == 0);
                const bool act = cov && (uint32_t)r_tid[k] < (uint32_t)a.n_contigs &&
                                 (!cand[k] || (uint32_t)r_mtid[k] < (uint32_t)a.n_contigs);
                if (!act) continue;
                if (lane_uni) val += (int)r_qlen[k];
                else cov_add(r_tid[k], (unsigned long long)r_qlen[k]);
            }
            if (lane_uni) key = r_tid[0];
            wave_add_runs(aligned, key, val, lane);
        }
        // ---- Only a candidate that can become a link [:169: read 2, mapped, mapq >= min_mapq] or a BWA-quirk read
        // [:141: unmapped read 1] has to be evaluated.  The others - every read 1 with its mate elsewhere: half of the
        // candidates - can only count as non-unique [:166-167] and, when one of the two contigs is not in the table
        // [:127-130], lose the coverage they were credited with above; both need the contigs' classes only, and when
        // every contig of the header is in the table (all_present) not even those.  They never enter the ring: the
        // evaluation rounds halve.  (With four waves per block and a barrier behind the queue this filter put the flag
        // loads on every wave's critical path and lost; a wave on its own waits for them here either way.)
        bool full[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t fl = r_flag[k];
            const bool link = (fl & kFlagRead2) && !(fl & kFlagUnmapped) && (int32_t)r_mapq[k] >= a.min_mapq;
            const bool quirk = (fl & kFlagUnmapped) && (fl & kFlagRead1);
            full[k] = cand[k] && (link || quirk);
            if (cand[k] && !full[k] && (uint32_t)r_tid[k] < (uint32_t)a.n_contigs && (uint32_t)r_mtid[k] < (uint32_t)a.n_contigs) {
                bool present = true;
                if (!all_present) present = a.cls8[r_tid[k]] != BESST_CLS_ABSENT && a.cls8[r_mtid[k]] != BESST_CLS_ABSENT;
                if (present) c_nonuniq_fishy += r_mapq[k] == 0 ? 1u : 0u;
                else if ((int32_t)r_mapq[k] >= a.min_mapq || r_mapq[k] == 0)
                    cov_add(r_tid[k], 0ull - (unsigned long long)r_qlen[k]);
            }
        }
        // ---- candidates -> the wave's ring, in record order (record = 4 * lane + k)
        const unsigned long long b0 = __ballot(full[0]), b1 = __ballot(full[1]);
        const unsigned long long b2 = __ballot(full[2]), b3 = __ballot(full[3]);
        const int total = __popcll(b0) + __popcll(b1) + __popcll(b2) + __popcll(b3);
        if (total == 0) return;                              // uniform
        int slot = below_cnt(b0) + below_cnt(b1) + below_cnt(b2) + below_cnt(b3);
        slot += q_head + q_count;                            // behind the queued ones (at most 63 + 256 entries in all)
        slot = slot >= kFwRing ? slot - kFwRing : slot;
        slot = slot >= kFwRing ? slot - kFwRing : slot;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (full[k]) {
                s_buf[0][slot] = (uint32_t)r_tid[k];
                s_buf[1][slot] = (uint32_t)r_mtid[k];
                s_buf[2][slot] = (uint32_t)r_pos[k];
                s_buf[3][slot] = (uint32_t)r_mpos[k];
                s_buf[4][slot] = (r_flag[k] & 0xffffu) | (r_mapq[k] << 16);
                s_qlen[slot] = (unsigned short)r_qlen[k];
                ++slot;
                slot = slot == kFwRing ? 0 : slot;
            }
        }
        __builtin_amdgcn_wave_barrier();                     // (one wave: its LDS operations complete in order)
        q_count += total;
    };
    if (block_base + kClsTile <= a.n) {
        // A block that lies inside the stream (all but the last): ONE wait per sub-tile.  By the counters a wave of the
        // unpipelined loop sat in s_waitcnt for 72 % of its life - tid / mtid first, then pos / mpos / flag where they
        // are needed, then the contig rows of every evaluation round, each a memory latency with nothing of the wave's
        // own to do meanwhile - while the SIMDs issued vector instructions half of the time.  Here everything the next
        // iteration needs is issued together at the top of this one, in front of the sub-tile's arithmetic: pos / mpos /
        // flag of sub-tile st + 1 (tid / mtid of st + 1 arrived an iteration ago), tid / mtid / mapq / qlen of st + 2,
        // and the contig rows of the next 64 queued candidates, whose round is finished at the top of the next
        // iteration.  (Issued at the END of an iteration the same loads are needed at once: 1.53 -> 1.71 ms.)  The candidate columns are
        // loaded by every lane - a lane without a candidate reads the sub-tile's first sector again - so that the
        // number of outstanding loads is the same on every path and the waits the compiler places are exact.
        // (the byte and halfword columns stay packed as they were loaded - one and two registers - until process() takes
        // them apart: three sub-tiles of columns are live at once)
        struct ColsA { int4 tid, mtid; uint32_t mapq; uint2 qlen; uint32_t bits; };     // (kBits: mtid unused, else bits)
        struct ColsC { int4 pos, mpos, mtid; uint2 flag; };                              // (mtid: kBits only)
        // Addresses: the block's first record of every column is a uniform pointer (a scalar register pair), a lane adds
        // a 32-bit offset to it (the saddr + voffset form of the load) - as 64-bit pointers per lane and column the seven
        // columns held fourteen vector registers through the loop, and the loop spilled.
        const int32_t* const b_tid = a.tid + block_base;
        const int32_t* const b_mtid = a.mtid + block_base;
        const int32_t* const b_pos = a.pos + block_base;
        const int32_t* const b_mpos = a.mpos + block_base;
        const uint16_t* const b_flag = a.flag + block_base;
        const uint8_t* const b_mapq = a.mapq + block_base;
        const uint16_t* const b_qlen = a.qlen + block_base;
        const uint8_t* const b_bits = kBits ? a.mate_bits + (block_base >> 3) : nullptr;
        typedef int v4i __attribute__((ext_vector_type(4)));
        auto load_a = [&](int st) {
            const uint32_t li = (uint32_t)st * (uint32_t)(kFwSub / 4) + (uint32_t)lane;   // the lane's group of four records
            ColsA v;
#ifndef BESST_FW_PLAIN_LOADS
            // (non-temporal: every record is read once - marking these four streams so took 2.3 % off the kernel; the same
            // mark on the candidate columns, where idle lanes re-read one sector, or on the segment stores made it slower)
            const v4i t4 = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(b_tid) + li);
            v4i m4 = t4;
            v.bits = 0u;
            if constexpr (kBits) {
                // (the lane's four records begin at block_base + 4 li: their bits are a nibble of byte li / 2 of the block's)
                v.bits = ((uint32_t)b_bits[li >> 1] >> ((li & 1u) * 4u)) & 15u;
            } else {
                m4 = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(b_mtid) + li);
            }
            const uint32_t q1 = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(b_mapq) + li);
            typedef unsigned int v2u __attribute__((ext_vector_type(2)));
            const v2u q4 = __builtin_nontemporal_load(reinterpret_cast<const v2u*>(b_qlen) + li);
            v.tid = make_int4(t4.x, t4.y, t4.z, t4.w);
            v.mtid = make_int4(m4.x, m4.y, m4.z, m4.w);
            v.mapq = q1;
            v.qlen = make_uint2(q4.x, q4.y);
#else
            v.tid = reinterpret_cast<const int4*>(b_tid)[li];
            v.bits = 0u;
            if constexpr (kBits) {
                v.mtid = v.tid;
                v.bits = ((uint32_t)b_bits[li >> 1] >> ((li & 1u) * 4u)) & 15u;
            } else {
                v.mtid = reinterpret_cast<const int4*>(b_mtid)[li];
            }
            v.mapq = reinterpret_cast<const uint32_t*>(b_mapq)[li];
            v.qlen = reinterpret_cast<const uint2*>(b_qlen)[li];
#endif
            return v;
        };
        auto load_c = [&](int st, const ColsA& v) {
            const bool need = kBits ? v.bits != 0u
                                    : (v.tid.x != v.mtid.x || v.tid.y != v.mtid.y || v.tid.z != v.mtid.z || v.tid.w != v.mtid.w);
            // (a lane without a candidate reads the first 16 bytes of the BLOCK's columns - one sector per column and block,
            // in cache after its first use - and not its sub-tile's first sector, which nobody else may want: 0.3 GB less
            // from HBM, -3 %)
            const uint32_t li = need ? (uint32_t)st * (uint32_t)(kFwSub / 4) + (uint32_t)lane : 0u;
            ColsC c;
            c.pos = reinterpret_cast<const int4*>(b_pos)[li];
            c.mpos = reinterpret_cast<const int4*>(b_mpos)[li];
            c.flag = reinterpret_cast<const uint2*>(b_flag)[li];
            if constexpr (kBits) c.mtid = reinterpret_cast<const int4*>(b_mtid)[li];
            else c.mtid = v.mtid;
            return c;
        };
        constexpr int kSubs = kClsTile / kFwSub;
        ColsA a0 = load_a(0), a1 = load_a(1);
        ColsC c0 = load_c(0, a0);
        Round pend;                                          // the round whose rows are in flight (cnt == 0: none)
        pend.cnt = 0; pend.tid = pend.mtid = -1; pend.pos = pend.mpos = 0; pend.fm = pend.qlen = 0;
        pend.c1 = a.table[0]; pend.c2 = pend.c1;
        for (int st = 0; st < kSubs; ++st) {
            // everything issued at the top of the iteration before - it had that iteration's arithmetic to arrive - is
            // waited for here, once
            if (pend.cnt) round_finish(pend);                // uniform
            while (q_count >= 128) run_round(64);            // a burst of candidates: rounds with a wait of their own
            // ---- the loads of the iterations to come, issued together
            const int st1 = st + 1 < kSubs ? st + 1 : kSubs - 1, st2 = st + 2 < kSubs ? st + 2 : kSubs - 1;
            const ColsC c1 = load_c(st1, a1);
            const ColsA a2 = load_a(st2);
            pend = round_fetch(q_count >= 64 ? 64 : 0);      // (fewer than 64 are left queued: the ring holds the 256 to come)
            {
                const int32_t r_tid[4] = {a0.tid.x, a0.tid.y, a0.tid.z, a0.tid.w};
                // (kBits: a lane that holds a candidate has read the `mtid` of its four records with the candidate columns; for
                // a lane that holds none they equal `tid`)
                const bool own = kBits && a0.bits != 0u;
                const int32_t r_mtid[4] = {own ? c0.mtid.x : a0.mtid.x, own ? c0.mtid.y : a0.mtid.y, own ? c0.mtid.z : a0.mtid.z,
                                           own ? c0.mtid.w : a0.mtid.w};
                const int32_t r_pos[4] = {c0.pos.x, c0.pos.y, c0.pos.z, c0.pos.w};
                const int32_t r_mpos[4] = {c0.mpos.x, c0.mpos.y, c0.mpos.z, c0.mpos.w};
                const uint32_t r_flag[4] = {c0.flag.x & 0xffffu, c0.flag.x >> 16, c0.flag.y & 0xffffu, c0.flag.y >> 16};
                const uint32_t r_mapq[4] = {a0.mapq & 255u, (a0.mapq >> 8) & 255u, (a0.mapq >> 16) & 255u, a0.mapq >> 24};
                const uint32_t r_qlen[4] = {a0.qlen.x & 0xffffu, a0.qlen.x >> 16, a0.qlen.y & 0xffffu, a0.qlen.y >> 16};
                process(r_tid, r_mtid, r_pos, r_mpos, r_flag, r_mapq, r_qlen);
            }
            a0 = a1; a1 = a2; c0 = c1;
        }
        if (pend.cnt) round_finish(pend);
        while (q_count >= 64) run_round(64);                 // what the last sub-tiles queued
    } else {
        // the stream's last block: element-wise loads behind the end check
        for (int st = 0; st < kClsTile / kFwSub; ++st) {
            const int64_t sub_base = block_base + (int64_t)st * kFwSub;
            if (sub_base >= a.n) break;                     // uniform
            const int64_t i0 = sub_base + (int64_t)lane * 4;
            int32_t r_tid[4], r_mtid[4], r_pos[4], r_mpos[4];
            uint32_t r_flag[4], r_mapq[4], r_qlen[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int64_t i = i0 + k;
                const bool in = i < a.n;
                r_tid[k] = in ? a.tid[i] : -1;
                r_mtid[k] = in ? a.mtid[i] : -1;           // tid == mtid == -1: no candidate, out of range: no coverage
                r_pos[k] = in ? a.pos[i] : 0;
                r_mpos[k] = in ? a.mpos[i] : 0;
                r_flag[k] = in ? a.flag[i] : 0;
                r_mapq[k] = in ? a.mapq[i] : 0;
                r_qlen[k] = in ? a.qlen[i] : 0;
            }
            process(r_tid, r_mtid, r_pos, r_mpos, r_flag, r_mapq, r_qlen);
            while (q_count >= 64) run_round(64);
        }
    }
    if (q_count > 0) run_round(q_count);                    // the unfinished round (uniform)
    if (lane == 0 && run_sum && (uint32_t)run_tid < (uint32_t)a.n_contigs)
        atomicAdd(&aligned[run_tid], (unsigned long long)run_sum);
    // ---- block summary (same planes as ordered_kernel)
    const int vals[7] = {(int)(c_count_nus & 0xffffu), (int)(c_nonuniq_fishy & 0xffffu), (int)(c_count_nus >> 16),
                         (int)(c_dup_long & 0xffffu), (int)(c_dup_long >> 16), (int)(c_nonuniq_fishy >> 16), 0};
    int tot[7];
#pragma unroll
    for (int f = 0; f < 6; ++f) tot[f] = wave_sum_dpp(vals[f]);
    tot[6] = c_reach_u;
    if constexpr (kRuns) {
        close_chunk();
        if (rl_over && lane == 0) atomicOr(run_status, 1u);
    }
    if (!kRuns && a.ps_table) {
        __builtin_amdgcn_wave_barrier();
        uint32_t* row = a.ps_table + (size_t)(blockIdx.x & (uint32_t)(a.ps_rows - 1)) * 512u;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int d = q * 64 + lane;
            const uint32_t c = (s_ph[d >> 1] >> (16 * (d & 1))) & 0xffffu;
            if (c) atomicAdd(&row[d], c);
        }
    }
    uint32_t v = 0;
    if (lane == kSumEmit) v = (uint32_t)st_emit;
    if (lane == kSumHas) v = st_known ? 1u : 0u;
    if (lane == kSumFirst1) v = hd_present ? (uint32_t)hd_o1 : 0u;
    if (lane == kSumFirst2) v = hd_present ? (uint32_t)hd_o2 : 0u;
    if (lane == kSumLast1) v = (uint32_t)st_p1;
    if (lane == kSumLast2) v = (uint32_t)st_p2;
    if (lane == kSumHeadInfo) v = hd_present ? (uint32_t)hd_info : 0u;
    if (lane == kSumHeadSlot) v = hd_present ? (uint32_t)hd_slot : kNoSlot;
#pragma unroll
    for (int f = 0; f < 7; ++f)
        if (lane == kSumCtr0 + f) v = (uint32_t)tot[f];
    if constexpr (kRuns) {
        if (lane == kSumChunks) v = (uint32_t)rl_chunks;
        if (lane == kSumRuns) v = (uint32_t)rl_runs + ((hd_present && hd_slot != (int)kNoSlot) ? 1u : 0u);
    }
    if (lane < kSumPlanes) summ.at(lane, blockIdx.x) = v;
}

// ---- stitch: resolve block heads, fix counters, scan tuple counts ------------------------------------
// One workgroup per SPAN of 4096 blocks: one lane per block and round, kStitchRounds rounds.  The summaries of all
// rounds are fetched up front (one memory round trip); both scans - "nearest earlier block that reached CreateEdge"
// (a max-scan of block indexes) and the tuple offsets (a sum-scan) - run as wave scans of every round at once plus ONE
// wave scan over the 64 (round, wave) totals: six barriers per span.  (A barrier-heavy scan per 1024 blocks cost
// 4.3 us per round.)
// A stream of more than one span (C3: 24 k blocks = 6 spans; one workgroup walking them took 70 us - every load and
// instruction of the stage through ONE compute unit) is stitched in two launches without any waiting between
// workgroups: stitch_spans_kernel leaves per span what the later ones need to know - its last reaching record, its
// first reaching record (the one that needs a predecessor from an earlier span) and its tuple count with that head's
// tuple still in - and every workgroup of stitch_kernel replays the aggregates of the spans before its own (a handful
// of CreateEdge evaluations) to get the prev_obs it starts from and the offset of its first tuple.
constexpr int kStitchRounds = 4;
constexpr uint32_t kStitchSpan = 1024u * kStitchRounds;
static_assert(kStitchRounds * 16 == 64, "the (round, wave) totals are scanned by one wave");

struct StitchAgg {
    int32_t has_any, l1, l2;            // the span's last record that reached CreateEdge
    int32_t head_present, f1, f2;       // its first one: resolved against the spans before
    uint32_t head_info;
    uint32_t total;                     // tuples of the span, that head's provisional tuple included
    uint32_t runs;                      // runs of the span (kSumRuns), that head's run of one included
};

__global__ __launch_bounds__(1024) void stitch_spans_kernel(SummView summ, uint32_t nblocks, uint32_t span, int detect,
                                                            const int32_t* __restrict__ carry,
                                                            StitchAgg* __restrict__ agg, int32_t* __restrict__ carry_in) {
    __shared__ int32_t s_l1[1024 * kStitchRounds], s_l2[1024 * kStitchRounds];
    __shared__ int s_tab[64];
    __shared__ int s_total;
    __shared__ int32_t s_first[4];
    __shared__ int s_sum[16], s_sum_runs[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t c0 = blockIdx.x * span;
    if (blockIdx.x == 0 && t == 0) { carry_in[0] = carry[0]; carry_in[1] = carry[1]; }   // stitch_kernel overwrites carry
    if (t == 0) { s_first[0] = -1; }
    uint32_t n_emit[kStitchRounds], head_info[kStitchRounds], has[kStitchRounds], n_runs[kStitchRounds];
    int32_t f1[kStitchRounds], f2[kStitchRounds], l1[kStitchRounds], l2[kStitchRounds];
#pragma unroll
    for (int r = 0; r < kStitchRounds; ++r) {
        const uint32_t b = c0 + (uint32_t)r * 1024u + (uint32_t)t;
        n_emit[r] = 0; has[r] = 0; f1[r] = f2[r] = l1[r] = l2[r] = 0; head_info[r] = 0; n_runs[r] = 0;
        if (b < nblocks && (uint32_t)r * 1024u + (uint32_t)t < span) {
            n_emit[r] = summ.at(kSumEmit, b); has[r] = summ.at(kSumHas, b);
            f1[r] = (int32_t)summ.at(kSumFirst1, b); f2[r] = (int32_t)summ.at(kSumFirst2, b);
            l1[r] = (int32_t)summ.at(kSumLast1, b); l2[r] = (int32_t)summ.at(kSumLast2, b);
            head_info[r] = summ.at(kSumHeadInfo, b);
            n_runs[r] = summ.at(kSumRuns, b);
        }
    }
    int incl[kStitchRounds];
#pragma unroll
    for (int r = 0; r < kStitchRounds; ++r) {
        int v = has[r] ? r * 1024 + t : -1;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(v, d, 64);
            if (lane >= d) v = o > v ? o : v;
        }
        incl[r] = v;
        s_l1[r * 1024 + t] = l1[r];
        s_l2[r * 1024 + t] = l2[r];
        if (lane == 63) s_tab[r * 16 + wave] = v;
    }
    __syncthreads();
    if (wave == 0) {
        int v = s_tab[lane];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(v, d, 64);
            if (lane >= d) v = o > v ? o : v;
        }
        const int ex = __shfl_up(v, 1, 64);
        s_tab[lane] = lane ? ex : -1;
        if (lane == 63) s_total = v;
    }
    __syncthreads();
    const int last_idx = s_total;
    int mine = 0, mine_runs = 0;
#pragma unroll
    for (int r = 0; r < kStitchRounds; ++r) {
        int ex = __shfl_up(incl[r], 1, 64);
        if (lane == 0) ex = -1;
        const int wpre = s_tab[r * 16 + wave];
        const int pi = ex > wpre ? ex : wpre;
        int n_final = (int)n_emit[r], r_final = (int)n_runs[r];
        if (has[r] && pi < 0) {                              // the span's first reaching block (one thread at most)
            s_first[0] = 1; s_first[1] = f1[r]; s_first[2] = f2[r]; s_first[3] = (int32_t)head_info[r];
        } else if (has[r]) {
            const CEDelta d = create_edge(f1[r], f2[r], s_l1[pi], s_l2[pi], head_info[r] & 1u, head_info[r] & 2u,
                                          head_info[r] & 4u, detect != 0);
            if ((head_info[r] & 8u) && !d.keep) { n_final -= 1; r_final -= r_final > 0 ? 1 : 0; }
        }
        mine += n_final;
        mine_runs += r_final;
    }
    mine = wave_sum(mine);
    mine_runs = wave_sum(mine_runs);
    if (lane == 0) { s_sum[wave] = mine; s_sum_runs[wave] = mine_runs; }
    __syncthreads();
    if (t == 0) {
        StitchAgg a;
        a.has_any = last_idx >= 0 ? 1 : 0;
        a.l1 = last_idx >= 0 ? s_l1[last_idx] : 0;
        a.l2 = last_idx >= 0 ? s_l2[last_idx] : 0;
        a.head_present = s_first[0] > 0 ? 1 : 0;
        a.f1 = s_first[1]; a.f2 = s_first[2]; a.head_info = (uint32_t)s_first[3];
        int tot = 0, tot_runs = 0;
        for (int w = 0; w < 16; ++w) { tot += s_sum[w]; tot_runs += s_sum_runs[w]; }
        a.total = (uint32_t)tot;
        a.runs = (uint32_t)tot_runs;
        agg[blockIdx.x] = a;
    }
}

// kRuns: the record loop grouped its runs and the blocks' run offsets ride in the upper half of the scanned word; without
// them the scans are 32 bits wide (C2's single stitch workgroup is a chain of such scans: 64-bit shuffles cost it 1.1 us of 10)
template <bool kRuns>
__global__ __launch_bounds__(1024) void stitch_kernel(SummView summ, uint32_t nblocks, int32_t* carry, int detect,
                                                      uint32_t* __restrict__ offsets,
                                                      uint32_t* __restrict__ skip_slot,
                                                      uint32_t* n_out,
                                                      unsigned long long* counters,
                                                      const int32_t* __restrict__ tails, int rank,
                                                      int32_t* __restrict__ slice_info,
                                                      const StitchAgg* __restrict__ agg,
                                                      const int32_t* __restrict__ carry_in, uint32_t span,
                                                      uint32_t* __restrict__ run_offsets) {
    // run_offsets (a record loop that grouped its runs, kSumRuns): where each block's runs begin in the stream-ordered
    // run list - the same scan as the tuple offsets, in the upper half of a 64-bit word; a head the stitch drops takes its
    // run of one with it
    // slice_info != nullptr (sharded build without a tail exchange): the slice's FIRST reaching record is left
    // unresolved - its provisional tuple stays, its CreateEdge call is not counted - and described in slice_info
    // { any reaching, last obs1, last obs2, head present, head obs1, head obs2, head info, head tuple position },
    // which travels in the exchange headers; the owners resolve it against the slices before (unpack_kernel).
    __shared__ int32_t s_l1[1024 * kStitchRounds], s_l2[1024 * kStitchRounds];
    __shared__ int s_any;            // a block before this span reached CreateEdge
    __shared__ int s_head_block;     // block of the slice head (speculative mode), -1 unless it lies in this span
    __shared__ int32_t s_head[4];    // its obs1, obs2, info, slot
    __shared__ int s_tab[64];
    __shared__ int s_total;
    using acc_t = typename std::conditional<kRuns, long long, int>::type;
    auto pack_counts = [](uint32_t tuples, uint32_t runs) -> acc_t {
        if (kRuns) return (acc_t)((long long)tuples | ((long long)runs << 32));
        return (acc_t)tuples;
    };
    __shared__ acc_t s_tab64[64];
    __shared__ acc_t s_total64;
    __shared__ int32_t s_carry[2];
    __shared__ acc_t s_base;             // tuples | runs << 32 in front of this span
    __shared__ int s_redc[16][7];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const bool last_span = blockIdx.x == gridDim.x - 1;
    if (t == 0) {
        // prev_obs entering the stream (agg == nullptr: a single span, carry is read before anything overwrites it)
        int32_t p1 = agg ? carry_in[0] : carry[0], p2 = agg ? carry_in[1] : carry[1];
        // multi-GPU: prev_obs entering this rank = tail {has, obs1, obs2, 0} of the nearest earlier rank that has one
        if (tails)
            for (int j = 0; j < rank; ++j)
                if (tails[j * 4]) { p1 = tails[j * 4 + 1]; p2 = tails[j * 4 + 2]; }
        // the spans before this one: what their first reaching records resolve to, where their tuples end
        int any = 0;
        acc_t base = 0;
        for (uint32_t j = 0; j < blockIdx.x; ++j) {
            const StitchAgg a = agg[j];
            acc_t tot = (acc_t)a.total;
            if (kRuns) tot = (acc_t)((long long)a.total | ((long long)a.runs << 32));
            if (a.head_present && !(slice_info && !any)) {   // (the slice head keeps its provisional tuple)
                const CEDelta d = create_edge(a.f1, a.f2, p1, p2, a.head_info & 1u, a.head_info & 2u, a.head_info & 4u,
                                              detect != 0);
                if ((a.head_info & 8u) && !d.keep) tot -= 1ll + (a.runs ? 1ll << 32 : 0ll);
            }
            base += tot;
            if (a.has_any) { p1 = a.l1; p2 = a.l2; any = 1; }
        }
        s_carry[0] = p1; s_carry[1] = p2; s_base = base;
        s_any = any; s_head_block = -1;
    }
    __syncthreads();
    int c_count = 0, c_long = 0, c_dup = 0, c_nus = 0, c_nonuniq = 0, c_fishy = 0, c_reach = 0;
    {
        const uint32_t c0 = blockIdx.x * span;
        uint32_t n_emit[kStitchRounds], head_info[kStitchRounds], head_slot[kStitchRounds], has[kStitchRounds];
        uint32_t n_runs[kStitchRounds];
        int32_t f1[kStitchRounds], f2[kStitchRounds], l1[kStitchRounds], l2[kStitchRounds];
#pragma unroll
        for (int r = 0; r < kStitchRounds; ++r) {
            const uint32_t b = c0 + (uint32_t)r * 1024u + (uint32_t)t;
            n_emit[r] = 0; has[r] = 0; f1[r] = f2[r] = l1[r] = l2[r] = 0; head_info[r] = 0; head_slot[r] = kNoSlot;
            n_runs[r] = 0;
            if (b < nblocks && (uint32_t)r * 1024u + (uint32_t)t < span) {
                n_runs[r] = kRuns ? summ.at(kSumRuns, b) : 0u;
                n_emit[r] = summ.at(kSumEmit, b); has[r] = summ.at(kSumHas, b);
                f1[r] = (int32_t)summ.at(kSumFirst1, b); f2[r] = (int32_t)summ.at(kSumFirst2, b);
                l1[r] = (int32_t)summ.at(kSumLast1, b); l2[r] = (int32_t)summ.at(kSumLast2, b);
                head_info[r] = summ.at(kSumHeadInfo, b); head_slot[r] = summ.at(kSumHeadSlot, b);
                c_count += (int)summ.at(kSumCtr0 + 0, b); c_nonuniq += (int)summ.at(kSumCtr0 + 1, b);
                c_nus += (int)summ.at(kSumCtr0 + 2, b); c_dup += (int)summ.at(kSumCtr0 + 3, b);
                c_long += (int)summ.at(kSumCtr0 + 4, b); c_fishy += (int)summ.at(kSumCtr0 + 5, b);
                c_reach += (int)summ.at(kSumCtr0 + 6, b);
            }
        }
        // ---- nearest earlier block (of this span) with a reaching record
        int incl[kStitchRounds];
#pragma unroll
        for (int r = 0; r < kStitchRounds; ++r) {
            int v = has[r] ? r * 1024 + t : -1;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int o = __shfl_up(v, d, 64);
                if (lane >= d) v = o > v ? o : v;
            }
            incl[r] = v;
            s_l1[r * 1024 + t] = l1[r];
            s_l2[r * 1024 + t] = l2[r];
            if (lane == 63) s_tab[r * 16 + wave] = v;
        }
        __syncthreads();
        if (wave == 0) {
            int v = s_tab[lane];
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int o = __shfl_up(v, d, 64);
                if (lane >= d) v = o > v ? o : v;
            }
            const int ex = __shfl_up(v, 1, 64);
            s_tab[lane] = lane ? ex : -1;
            if (lane == 63) s_total = v;
        }
        __syncthreads();
        const int last_idx = s_total;
        uint32_t n_final[kStitchRounds], skip[kStitchRounds];
#pragma unroll
        for (int r = 0; r < kStitchRounds; ++r) {
            int ex = __shfl_up(incl[r], 1, 64);
            if (lane == 0) ex = -1;
            const int wpre = s_tab[r * 16 + wave];
            const int pi = ex > wpre ? ex : wpre;
            const int32_t p1 = pi >= 0 ? s_l1[pi] : s_carry[0];
            const int32_t p2 = pi >= 0 ? s_l2[pi] : s_carry[1];
            n_final[r] = n_emit[r];
            skip[r] = kNoSlot;
            if (has[r] && slice_info && pi < 0 && !s_any) {        // the slice head: resolved by the owners
                s_head_block = (int)(c0 + (uint32_t)r * 1024u + (uint32_t)t);
                s_head[0] = f1[r]; s_head[1] = f2[r]; s_head[2] = (int32_t)head_info[r]; s_head[3] = (int32_t)head_slot[r];
            } else if (has[r]) {
                const CEDelta d = create_edge(f1[r], f2[r], p1, p2, head_info[r] & 1u, head_info[r] & 2u,
                                              head_info[r] & 4u, detect != 0);
                c_count += d.count; c_long += d.too_long; c_dup += d.dup; c_nus += d.nus;
                if ((head_info[r] & 8u) && !d.keep) {
                    n_final[r] -= 1; skip[r] = head_slot[r];
                    n_runs[r] -= n_runs[r] ? 1u : 0u;
                }
            }
        }
        __syncthreads();
        // ---- tuple offsets (and run offsets: upper half of the word)
        acc_t sincl[kStitchRounds];
#pragma unroll
        for (int r = 0; r < kStitchRounds; ++r) {
            acc_t v = pack_counts(n_final[r], n_runs[r]);
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const acc_t o = __shfl_up(v, d, 64);
                if (lane >= d) v += o;
            }
            sincl[r] = v;
            if (lane == 63) s_tab64[r * 16 + wave] = v;
        }
        __syncthreads();
        if (wave == 0) {
            const acc_t own = s_tab64[lane];
            acc_t v = own;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const acc_t o = __shfl_up(v, d, 64);
                if (lane >= d) v += o;
            }
            s_tab64[lane] = v - own;
            if (lane == 63) s_total64 = v;
        }
        __syncthreads();
        const acc_t base = s_base;
#pragma unroll
        for (int r = 0; r < kStitchRounds; ++r) {
            const uint32_t b = c0 + (uint32_t)r * 1024u + (uint32_t)t;
            if (b < nblocks && (uint32_t)r * 1024u + (uint32_t)t < span) {
                const acc_t ex = base + s_tab64[r * 16 + wave] + sincl[r] - pack_counts(n_final[r], n_runs[r]);
                offsets[b] = (uint32_t)ex;
                skip_slot[b] = skip[r];
                if (kRuns) run_offsets[b] = (uint32_t)((long long)ex >> 32);
            }
        }
        const acc_t span_total = s_total64;
        __syncthreads();
        if (t == 0) {
            s_base = base + span_total;
            if (last_idx >= 0) { s_carry[0] = s_l1[last_idx]; s_carry[1] = s_l2[last_idx]; s_any = 1; }
        }
        __syncthreads();
    }
    // counters fixed up by the heads
    // besst_counters fields: count 0, non_unique 1, non_unique_for_scaf 2, nr_of_duplicates 3,
    // reads_with_too_long_insert 4, fishy_reads 5, n_reach 7 (n_tuples, 6, below)
    int vals[7] = {c_count, c_nonuniq, c_nus, c_dup, c_long, c_fishy, c_reach};
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
        if (slice_info && s_head_block >= 0) {               // the span that holds the slice head describes it
            slice_info[3] = 1;
            slice_info[4] = s_head[0]; slice_info[5] = s_head[1]; slice_info[6] = s_head[2];
            // position of the head's provisional tuple in the slice's compacted stream (-1: none was emitted)
            slice_info[7] = (s_head[2] & 8) ? (int32_t)(offsets[s_head_block] + (uint32_t)s_head[3]) : -1;
        }
        if (last_span) {
            carry[0] = s_carry[0];
            carry[1] = s_carry[1];
            *n_out = (uint32_t)s_base;
            atomicAdd(&counters[6], (unsigned long long)(uint32_t)s_base);
            if (slice_info) {
                slice_info[0] = s_any; slice_info[1] = s_carry[0]; slice_info[2] = s_carry[1];
                if (!s_any) {                                // no record of the slice reached CreateEdge: no head either
                    slice_info[3] = 0; slice_info[4] = 0; slice_info[5] = 0; slice_info[6] = 0; slice_info[7] = -1;
                }
            }
        }
    }
}

// The record loop counted the sort digits of every tuple it wrote; the head tuples the stitch has dropped since
// (skip_slot) are taken out of the counts again (row 0 of the table: only the column sums matter).
__global__ __launch_bounds__(256) void presort_fixup_kernel(const uint64_t* __restrict__ seg_keys,
                                                            const uint32_t* __restrict__ skip_slot, uint32_t nblocks,
                                                            PresortSpec ps, const uint8_t* __restrict__ cls8,
                                                            int32_t n_contigs, unsigned long long* __restrict__ aligned,
                                                            const uint32_t* __restrict__ offsets,
                                                            const uint32_t* __restrict__ n_out,
                                                            uint32_t* __restrict__ chunk_first) {
    // when compact_kernel does not run, its side job is done here: coverage of contigs that are not in the table is
    // not part of cont_aligned_len (CreateGraph.py:89-95)
    if (cls8)
        for (int32_t c = (int32_t)(blockIdx.x * blockDim.x + threadIdx.x); c < n_contigs; c += (int32_t)(gridDim.x * blockDim.x))
            if (!cls8[c]) aligned[c] = 0;
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks) return;
    if (chunk_first) {
        // a stage 2 that reads the segments chunk by chunk (runs.hip) starts every chunk at a known block: the chunks
        // whose first dense position lies in this block's range
        const uint32_t lo = offsets[b], hi = b + 1u < nblocks ? offsets[b + 1u] : *n_out;
        for (uint32_t c = (lo + (uint32_t)kRunChunk - 1u) / (uint32_t)kRunChunk; (uint64_t)c * kRunChunk < hi; ++c) chunk_first[c] = b;
    }
    const uint32_t skip = skip_slot[b];
    if (skip == kNoSlot || !ps.count) return;
    const uint64_t k = seg_keys[(size_t)b * kClsTile + skip] - ps.key_base;
    atomicSub(&ps.table[(uint32_t)(k >> ps.shift) & 255u], 1u);
    atomicSub(&ps.table[256u + ((uint32_t)(k >> (ps.shift + 8)) & 255u)], 1u);
}

__global__ __launch_bounds__(256) void compact_kernel(SummView summ,
                                                      const uint32_t* __restrict__ offsets,
                                                      const uint32_t* __restrict__ skip_slot,
                                                      const uint64_t* __restrict__ seg_keys,
                                                      const uint64_t* __restrict__ seg_payload,
                                                      uint64_t* __restrict__ keys,
                                                      uint64_t* __restrict__ payload,
                                                      const uint8_t* __restrict__ cls8, int32_t n_contigs,
                                                      unsigned long long* __restrict__ aligned, PresortSpec ps) {
    __shared__ uint32_t s_h[512];                        // the block's share of the sort's two digit histograms
    if (ps.table) {                                      // uniform
        s_h[threadIdx.x] = 0;
        s_h[threadIdx.x + 256] = 0;
    }
    // coverage of contigs that are not in the table is not part of cont_aligned_len (CreateGraph.py:89-95)
    for (int32_t c = (int32_t)(blockIdx.x * blockDim.x + threadIdx.x); c < n_contigs; c += (int32_t)(gridDim.x * blockDim.x))
        if (!cls8[c]) aligned[c] = 0;
    const uint32_t b = blockIdx.x;
    const int64_t base = (int64_t)b * kClsTile;
    // the first 256 segment entries are fetched together with the block's bookkeeping (the segment is always
    // kClsTile entries long, so the speculative read is in bounds): one memory round trip for nearly every block
    const uint64_t k0 = seg_keys[base + threadIdx.x], p0 = seg_payload[base + threadIdx.x];
    const uint32_t n = summ.at(kSumEmit, b);
    const uint32_t skip = skip_slot[b], off = offsets[b];
    if (n == 0) return;
    if (ps.table) __syncthreads();                       // (n is the same in every thread)
    for (uint32_t j = threadIdx.x; j < n; j += blockDim.x) {
        if (j == skip) continue;
        const uint32_t dst = off + j - (j > skip ? 1u : 0u);
        const uint64_t key = j < blockDim.x ? k0 : seg_keys[base + j];
        keys[dst] = key;
        payload[dst] = j < blockDim.x ? p0 : seg_payload[base + j];
        if (ps.table && dst < ps.cap) {
            // tuples of one contig follow each other and half of them have that contig's own end as their smaller node:
            // the lanes that share the first active lane's digit add once, together (else they pile up on one counter)
            const uint64_t k = key - ps.key_base;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const uint32_t d = (uint32_t)(k >> (ps.shift + 8 * q)) & 255u;
                const uint32_t f = (uint32_t)__builtin_amdgcn_readfirstlane((int)d);
                const unsigned long long m = __ballot(d == f);
                if (d != f) atomicAdd(&s_h[256 * q + d], 1u);
                else if ((m & ((1ull << (threadIdx.x & 63)) - 1ull)) == 0ull) atomicAdd(&s_h[256 * q + f], (uint32_t)__popcll(m));
            }
        }
    }
    if (!ps.table) return;
    __syncthreads();
    uint32_t* row = ps.table + (size_t)(b & (uint32_t)(ps.rows - 1)) * 512u;
    for (int d = threadIdx.x; d < 512; d += blockDim.x) {
        const uint32_t v = s_h[d];
        if (v) atomicAdd(&row[d], v);
    }
}

struct ClsWorkspace {
    uint64_t* seg_keys;
    uint64_t* seg_payload;
    SummView summ;
    uint32_t* offsets;
    uint32_t* skip;
    uint32_t* chunk_first;          // per chunk of kRunChunk tuples of the dense stream: its first block
    uint32_t* run_offsets;          // per block: its first run in the stream-ordered run list (record loop that groups runs)
    uint32_t* run_status;           // one word: a block's run tables overflowed
    struct StitchAgg* agg;          // one per span of blocks, + the prev_obs entering the stream (2 x int32) behind them
    uint32_t agg_spans;
    unsigned long long* bitmask;
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
    w.summ.p = reinterpret_cast<uint32_t*>(p + off);
    w.summ.stride = (uint32_t)align_up((size_t)nblocks, 64);
    off += align_up((size_t)w.summ.stride * kSumPlanes * sizeof(uint32_t), 256);
    w.offsets = reinterpret_cast<uint32_t*>(p + off); off += align_up((size_t)nblocks * 4, 256);
    w.skip = reinterpret_cast<uint32_t*>(p + off); off += align_up((size_t)nblocks * 4, 256);
    w.chunk_first = reinterpret_cast<uint32_t*>(p + off); off += align_up(((size_t)n / kRunChunk + 2) * 4, 256);
    w.run_offsets = reinterpret_cast<uint32_t*>(p + off); off += align_up((size_t)nblocks * 4, 256);
    w.run_status = reinterpret_cast<uint32_t*>(p + off); off += 256;
    // (room for 4096 spans whatever the stream: the test knob BESST_STITCH_SPAN cuts small streams into many)
    const size_t spans = (size_t)((nblocks + kStitchSpan - 1) / kStitchSpan) + 4096;
    w.agg_spans = (uint32_t)spans;
    w.agg = reinterpret_cast<StitchAgg*>(p + off); off += align_up(spans * sizeof(StitchAgg) + 16, 256);
    // candidate bits: stream_kernel writes whole workgroups, so round the group count up to its tile
    const int64_t stream_blocks = (n + kStreamTile - 1) / kStreamTile;
    w.n_groups = stream_blocks * (kStreamTile / kGroup);
    w.bitmask = reinterpret_cast<unsigned long long*>(p + off); off += align_up((size_t)w.n_groups * kGroupWords * 8, 256);
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
__global__ __launch_bounds__(256) void tail_kernel(SummView summ, uint32_t nblocks,
                                                   int32_t* __restrict__ tail) {
    __shared__ int s_best[4];
    int best = -1;
    for (uint32_t b = threadIdx.x; b < nblocks; b += blockDim.x)
        if (summ.at(kSumHas, b)) best = (int)b > best ? (int)b : best;
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
        tail[1] = best >= 0 ? (int32_t)summ.at(kSumLast1, (uint32_t)best) : 0;
        tail[2] = best >= 0 ? (int32_t)summ.at(kSumLast2, (uint32_t)best) : 0;
        tail[3] = 0;
    }
}

__global__ void zero_tail_kernel(int32_t* tail) {
    if (threadIdx.x < 4) tail[threadIdx.x] = 0;
}

// counts[0] += records with tid != mtid, counts[1] += records looked at, over every step-th 1024-record tile.  A few
// hundred workgroups walk the sampled tiles and add ONE pair of atomics each (an atomic pair per wave and tile was
// 33 k contended device-scope atomics: 0.4 ms for a 4 M-record sample).
__global__ __launch_bounds__(256) void density_kernel(const int32_t* __restrict__ tid, const int32_t* __restrict__ mtid,
                                                      int64_t n, int64_t n_tiles, int64_t step,
                                                      unsigned long long* __restrict__ counts) {
    __shared__ int s_c[4], s_m[4];
    int c = 0, m = 0;
    for (int64_t tile = (int64_t)blockIdx.x * step; tile < n_tiles; tile += (int64_t)gridDim.x * step) {
        const int64_t i0 = tile * 1024 + (int64_t)threadIdx.x * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i0 + k < n) { ++m; c += tid[i0 + k] != mtid[i0 + k] ? 1 : 0; }
    }
    c = wave_sum(c);
    m = wave_sum(m);
    if ((threadIdx.x & 63) == 0) { s_c[threadIdx.x >> 6] = c; s_m[threadIdx.x >> 6] = m; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&counts[0], (unsigned long long)(s_c[0] + s_c[1] + s_c[2] + s_c[3]));
        atomicAdd(&counts[1], (unsigned long long)(s_m[0] + s_m[1] + s_m[2] + s_m[3]));
    }
}

// prev_obs entering rank `rank`: the tail of the nearest earlier rank that has one, else what is in carry
__global__ void resolve_carry_kernel(const int32_t* __restrict__ tails, int rank, int32_t* __restrict__ carry) {
    if (threadIdx.x != 0) return;
    for (int j = 0; j < rank; ++j)
        if (tails[j * 4]) { carry[0] = tails[j * 4 + 1]; carry[1] = tails[j * 4 + 2]; }
}

}  // namespace

bool classify_can_group_runs(const ClassifyArgs& a) {
    // BESST_LOOP_RUNS=0 (A/B runs, tests): the record loop writes keys and rg_group_kernel finds the runs
    static const int knob = [] { const char* e = getenv("BESST_LOOP_RUNS"); return e ? atoi(e) : 1; }();
    return knob != 0 && a.record_path == 1 && a.n > 0;
}

int launch_classify_scan(hipStream_t s, const ClassifyArgs& a, int64_t* aligned, besst_counters* counters,
                         void* ws, size_t ws_bytes, bool group_runs) {
    if (a.n <= 0) return BESST_OK;
    BESST_REQUIRE(a.n < (int64_t)1 << 32, "classify: more than 2^32-1 records in one call");
    const ClsWorkspace w = carve(ws, a.n);
    BESST_REQUIRE(ws != nullptr && ws_bytes >= w.total, "classify: workspace too small");
    const uint32_t nblocks = (uint32_t)((a.n + kClsTile - 1) / kClsTile);
    const uint32_t stream_blocks = (uint32_t)((a.n + kStreamTile - 1) / kStreamTile);
    // (A single fused pass for candidate-dense mate-pair libraries - coalesced loads of all seven columns, candidates
    // compacted and evaluated in LDS, chain by wave 0 - was built and measured on a C3 slice: correct, but 0.70 ms
    // against 0.16 + 0.42 ms for the two passes; at 135 VGPRs and a barrier-separated chain per 1024 records it is
    // latency bound at 3 waves per SIMD.  The split design below serves every library.)
    BESST_REQUIRE(!group_runs || classify_can_group_runs(a), "classify: this record loop cannot group runs");
    if (a.record_path == 1) {
        if (group_runs) BESST_HIP_TRY(hipMemsetAsync(w.run_status, 0, 4, s));
        ProfScope ps(s, kProfFusedWave);
        auto* al = reinterpret_cast<unsigned long long*>(aligned);
        if (group_runs && a.mate_bits)
            hipLaunchKernelGGL((fused_wave_kernel<true, true>), dim3(nblocks), dim3(64), 0, s, a, al, w.seg_keys, w.seg_payload, w.summ, w.run_status);
        else if (group_runs)
            hipLaunchKernelGGL((fused_wave_kernel<true, false>), dim3(nblocks), dim3(64), 0, s, a, al, w.seg_keys, w.seg_payload, w.summ, w.run_status);
        else if (a.mate_bits)
            hipLaunchKernelGGL((fused_wave_kernel<false, true>), dim3(nblocks), dim3(64), 0, s, a, al, w.seg_keys, w.seg_payload, w.summ, w.run_status);
        else
            hipLaunchKernelGGL((fused_wave_kernel<false, false>), dim3(nblocks), dim3(64), 0, s, a, al, w.seg_keys, w.seg_payload, w.summ, w.run_status);
        BESST_HIP_TRY(hipGetLastError());
        return BESST_OK;
    }
    {
        ProfScope ps(s, kProfStream);
        if (a.mate_bits)
            hipLaunchKernelGGL(stream_kernel<true>, dim3(stream_blocks), dim3(kStreamThreads), 0, s, a,
                               reinterpret_cast<unsigned long long*>(aligned), w.bitmask);
        else
            hipLaunchKernelGGL(stream_kernel<false>, dim3(stream_blocks), dim3(kStreamThreads), 0, s, a,
                               reinterpret_cast<unsigned long long*>(aligned), w.bitmask);
    }
    {
        ProfScope ps(s, kProfOrdered);
        hipLaunchKernelGGL(ordered_kernel, dim3(nblocks), dim3(kOrdThreads), 0, s, a, w.bitmask, w.n_groups,
                           reinterpret_cast<unsigned long long*>(aligned), w.seg_keys, w.seg_payload, w.summ);
    }
    (void)counters;
    BESST_HIP_TRY(hipGetLastError());
    return BESST_OK;
}

namespace {
// one thread per byte of the bit column: eight records' tid != mtid (records behind n: 0)
__global__ __launch_bounds__(256) void mate_bits_kernel(const int32_t* __restrict__ tid, const int32_t* __restrict__ mtid,
                                                        int64_t first_byte, int64_t n_bytes, int64_t n, uint8_t* __restrict__ bits) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n_bytes) return;
    const int64_t i0 = (first_byte + k) * 8;
    uint32_t b = 0;
    if (i0 + 8 <= n) {
        const int4 t0 = *reinterpret_cast<const int4*>(tid + i0), t1 = *reinterpret_cast<const int4*>(tid + i0 + 4);
        const int4 m0 = *reinterpret_cast<const int4*>(mtid + i0), m1 = *reinterpret_cast<const int4*>(mtid + i0 + 4);
        b = (t0.x != m0.x ? 1u : 0u) | (t0.y != m0.y ? 2u : 0u) | (t0.z != m0.z ? 4u : 0u) | (t0.w != m0.w ? 8u : 0u) |
            (t1.x != m1.x ? 16u : 0u) | (t1.y != m1.y ? 32u : 0u) | (t1.z != m1.z ? 64u : 0u) | (t1.w != m1.w ? 128u : 0u);
    } else {
        for (int j = 0; j < 8; ++j)
            if (i0 + j < n && tid[i0 + j] != mtid[i0 + j]) b |= 1u << j;
    }
    bits[first_byte + k] = (uint8_t)b;
}
}  // namespace

int launch_mate_bits(hipStream_t s, const int32_t* tid, const int32_t* mtid, int64_t lo, int64_t hi, int64_t n, uint8_t* bits) {
    if (hi > n) hi = n;
    if (lo < 0) lo = 0;
    if (hi <= lo) return BESST_OK;
    const int64_t first_byte = lo >> 3, n_bytes = ((hi + 7) >> 3) - first_byte;
    hipLaunchKernelGGL(mate_bits_kernel, dim3((uint32_t)((n_bytes + 255) / 256)), dim3(256), 0, s, tid, mtid, first_byte, n_bytes, n, bits);
    BESST_HIP_TRY(hipGetLastError());
    return BESST_OK;
}

int launch_candidate_density(hipStream_t s, int64_t n, const int32_t* tid, const int32_t* mtid, int64_t sample_records,
                             unsigned long long* counts) {
    BESST_HIP_TRY(hipMemsetAsync(counts, 0, 16, s));
    if (n <= 0) return BESST_OK;
    const int64_t n_tiles = (n + 1023) / 1024;
    int64_t want = sample_records > 0 ? (sample_records + 1023) / 1024 : n_tiles;
    if (want < 1) want = 1;
    if (want > n_tiles) want = n_tiles;
    const int64_t step = n_tiles / want;                    // every step-th tile
    int64_t blocks = (n_tiles + step - 1) / step;
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(density_kernel, dim3((uint32_t)blocks), dim3(256), 0, s, tid, mtid, n, n_tiles, step, counts);
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
                         size_t ws_bytes, const uint8_t* cls8, int32_t n_contigs, int64_t* aligned,
                         const int32_t* tails, int rank, int32_t* slice_info, PresortSpec* presort) {
    PresortSpec pre{};
    if (presort && presort->table) {
        pre = *presort;
        BESST_REQUIRE(pre.rows > 0 && (pre.rows & (pre.rows - 1)) == 0 && pre.shift >= 0 && pre.shift <= 47, "classify: bad presort description");
        // (in_record_loop: launch_classify cleared the table before the record loop ran)
        if (!pre.in_record_loop) BESST_HIP_TRY(hipMemsetAsync(pre.table, 0, (size_t)pre.rows * 512 * sizeof(uint32_t), s));
    }
    if (n <= 0) {
        BESST_HIP_TRY(hipMemsetAsync(n_out, 0, sizeof(uint32_t), s));
        if (slice_info) BESST_HIP_TRY(hipMemsetAsync(slice_info, 0, 8 * sizeof(int32_t), s));
        if (presort) presort->segmented = 0;
        return BESST_OK;
    }
    const ClsWorkspace w = carve(ws, n);
    BESST_REQUIRE(ws != nullptr && ws_bytes >= w.total, "classify: workspace too small");
    const uint32_t nblocks = (uint32_t)((n + kClsTile - 1) / kClsTile);
    auto* ctr = reinterpret_cast<unsigned long long*>(counters);
    const bool grouped = pre.table && pre.in_record_loop == 3;   // the record loop left runs, not keys
    BESST_REQUIRE(!grouped || (pre.segmented && !slice_info && !tails), "classify: grouped runs need the segmented hand-over");
    uint32_t* run_offsets = grouped ? w.run_offsets : nullptr;
    // (Folding the stitch into compact_kernel - every workgroup working its own offset out from the summaries of the
    // blocks before it - was built three ways (scans, scan-free with a walk-back, batched plane loads) and always
    // landed at 15-17 us against 10 + 5 us for the two launches: 1221 workgroups re-reading the same 350 cache lines
    // is an L2 hot spot.  The single-workgroup stitch stays.)
    {
        ProfScope ps(s, kProfStitch);
        // BESST_STITCH_SPAN (tests): fewer blocks per span, so that small inputs exercise the hand-over between spans
        uint32_t span = kStitchSpan;
        if (const char* e = getenv("BESST_STITCH_SPAN")) {
            const long v = atol(e);
            if (v >= 1 && v < (long)kStitchSpan) span = (uint32_t)v;
        }
        const uint32_t spans = (nblocks + span - 1) / span;
        BESST_REQUIRE(spans <= w.agg_spans, "classify: too many stitch spans for the workspace");
        if (spans > 1) {
            int32_t* carry_in = reinterpret_cast<int32_t*>(w.agg + w.agg_spans);
            hipLaunchKernelGGL(stitch_spans_kernel, dim3(spans), dim3(1024), 0, s, w.summ, nblocks, span, detect_dup, carry,
                               w.agg, carry_in);
            if (run_offsets)
                hipLaunchKernelGGL(stitch_kernel<true>, dim3(spans), dim3(1024), 0, s, w.summ, nblocks, carry, detect_dup, w.offsets,
                                   w.skip, n_out, ctr, tails, rank, slice_info, w.agg, carry_in, span, run_offsets);
            else
                hipLaunchKernelGGL(stitch_kernel<false>, dim3(spans), dim3(1024), 0, s, w.summ, nblocks, carry, detect_dup, w.offsets,
                                   w.skip, n_out, ctr, tails, rank, slice_info, w.agg, carry_in, span, run_offsets);
        } else {
            if (run_offsets)
                hipLaunchKernelGGL(stitch_kernel<true>, dim3(1), dim3(1024), 0, s, w.summ, nblocks, carry, detect_dup, w.offsets,
                                   w.skip, n_out, ctr, tails, rank, slice_info, (const StitchAgg*)nullptr,
                                   (const int32_t*)nullptr, span, run_offsets);
            else
                hipLaunchKernelGGL(stitch_kernel<false>, dim3(1), dim3(1024), 0, s, w.summ, nblocks, carry, detect_dup, w.offsets,
                                   w.skip, n_out, ctr, tails, rank, slice_info, (const StitchAgg*)nullptr,
                                   (const int32_t*)nullptr, span, run_offsets);
        }
    }
    {
        PresortSpec in_compact = pre;
        const bool segmented = pre.table && pre.in_record_loop && pre.segmented && !slice_info && !tails;
        if (pre.table && pre.in_record_loop) {
            ProfScope ps(s, kProfFixup);
            hipLaunchKernelGGL(presort_fixup_kernel, dim3((nblocks + 255) / 256), dim3(256), 0, s, w.seg_keys, w.skip, nblocks, pre,
                               segmented ? cls8 : nullptr, n_contigs, reinterpret_cast<unsigned long long*>(aligned),
                               w.offsets, n_out, (segmented && !grouped) ? w.chunk_first : nullptr);
            in_compact.table = nullptr;
        }
        if (segmented) {                                     // the sort's first pass reads the segments: no dense copy
            presort->segmented = 1;
            presort->seg = SegSource{w.seg_keys, w.seg_payload, w.offsets, w.skip, nblocks, (uint32_t)kClsTile, payload, w.chunk_first,
                                     grouped ? w.run_offsets : nullptr, w.summ.p, w.summ.stride, w.run_status};
        } else {
            if (presort) presort->segmented = 0;
            ProfScope ps(s, kProfCompact);
            hipLaunchKernelGGL(compact_kernel, dim3(nblocks), dim3(256), 0, s, w.summ, w.offsets, w.skip, w.seg_keys,
                               w.seg_payload, keys, payload, cls8, n_contigs, reinterpret_cast<unsigned long long*>(aligned),
                               in_compact);
        }
    }
    BESST_HIP_TRY(hipGetLastError());
    return BESST_OK;
}

int launch_classify(hipStream_t s, const ClassifyArgs& a, int32_t* carry, int64_t* aligned, uint64_t* keys,
                    uint64_t* payload, uint32_t* n_out, besst_counters* counters, void* ws, size_t ws_bytes,
                    PresortSpec* presort) {
    PresortSpec pre{};
    ClassifyArgs b = a;
    if (presort && presort->table) {
        pre = *presort;
        pre.in_record_loop = a.record_path == 1 && a.n > 0;
        if (pre.in_record_loop && pre.count) {
            BESST_REQUIRE(pre.rows > 0 && (pre.rows & (pre.rows - 1)) == 0, "classify: bad presort description");
            BESST_HIP_TRY(hipMemsetAsync(pre.table, 0, (size_t)pre.rows * 512 * sizeof(uint32_t), s));
            b.ps_table = pre.table; b.ps_rows = pre.rows; b.ps_shift = pre.shift; b.ps_base = pre.key_base;
        }
        // (a stage 2 that groups runs never reads the histograms: the loop hands its segments over and counts nothing -
        // and where it can, it finds the runs itself)
    }
    const bool group_runs = pre.table && pre.in_record_loop && !pre.count && pre.segmented && classify_can_group_runs(a);
    int rc = launch_classify_scan(s, b, aligned, counters, ws, ws_bytes, group_runs);
    if (rc) return rc;
    if (pre.in_record_loop) pre.in_record_loop = pre.count ? 1 : (group_runs ? 3 : 2);
    rc = launch_classify_emit(s, a.n, a.detect_dup, carry, keys, payload, n_out, counters, ws, ws_bytes, a.cls8,
                              a.n_contigs, aligned, nullptr, 0, nullptr, pre.table ? &pre : nullptr);
    if (presort) {
        presort->in_record_loop = pre.in_record_loop;
        presort->segmented = pre.table ? pre.segmented : 0;
        presort->seg = pre.seg;
    }
    return rc;
}

}  // namespace besst
