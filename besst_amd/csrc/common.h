// Shared declarations for the gfx950 kernels of libbesst_amd.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/besst_amd.h"

namespace besst {

// ---- error plumbing -----------------------------------------------------------------------------
void set_error(const char* fmt, ...);

#define BESST_HIP_TRY(expr)                                                                  \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            besst::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                             __LINE__);                                                      \
            return BESST_ERR_HIP;                                                            \
        }                                                                                    \
    } while (0)

#define BESST_REQUIRE(cond, msg)                 \
    do {                                         \
        if (!(cond)) {                           \
            besst::set_error("%s", msg);         \
            return BESST_ERR_ARG;                \
        }                                        \
    } while (0)

// ---- the BAM reader's position in its file (bam_reader.hip), for the streamed ingest of api.hip
}  // namespace besst
struct besst_bam;
namespace besst {
int64_t bam_file_bytes(besst_bam* b);
int64_t bam_file_position(besst_bam* b);
// for the device ingest (besst_ctx_push_bam_device): the mapped file, where the next unread record lies (file offset of
// its BGZF block + offset inside the inflated block), a parallel copy on the reader's pool, and "read to the end"
const uint8_t* bam_file_map(besst_bam* b);
bool bam_record_position(besst_bam* b, int64_t* block_file_off, uint32_t* in_block_off);
bool bam_parallel_read(besst_bam* b, void* dst, int64_t file_off, size_t bytes);
void bam_mark_consumed(besst_bam* b, int64_t saturated_qlen);

// ---- BAM ingest on the GPU (bgzf_gpu.hip) ---------------------------------------------------------------
struct BgzfBlock {             // one BGZF block of a chunk: its DEFLATE payload in the chunk's compressed bytes, its place in
    uint32_t src_off, src_len; // the chunk's inflated scratch (256-byte aligned) and ISIZE
    uint32_t dst_off_lo, dst_off_hi;
    uint32_t dst_len, crc;     // and the CRC-32 of the inflated bytes from the gzip trailer
};
constexpr int kBamBlockRecs = 2048;   // a 64 KiB block holds < 65536 / 36 records
struct BamColumns {
    int32_t *tid, *mtid, *pos, *mpos, *tlen;
    uint16_t *flag, *qlen;
    uint8_t* mapq;
    int32_t *head_rlen, *head_alen;   // libmetrics' read-length step: the first records' query_length / reference_length
    uint16_t* head_qlen;
};
int launch_bgzf_inflate(hipStream_t s, const uint8_t* src, const BgzfBlock* blocks, uint32_t n_blocks, uint8_t* dst,
                        uint32_t* status, uint32_t* symbols = nullptr);
size_t bgzf_inflate_symbol_places(size_t inflated_bytes, size_t n_blocks);
// modes of a chunk's walk: the first start of the chunk is a guess nobody vouches for (the first chunk of a part of the
// file whose entry is not known yet) / only the record in the tail slot counts (the blocks behind a part's end, inflated
// for the bytes of the part's last record); summary: 12 words (bgzf_gpu.hip)
constexpr uint32_t kWalkFirstGuessed = 1u, kWalkOverhang = 2u;
int launch_bam_walk_scan(hipStream_t s, const uint8_t* inflated, const BgzfBlock* blocks, uint32_t n_blocks, uint64_t chunk_end,
                         int32_t n_ref, uint32_t forced_block, uint32_t forced_entry, uint32_t mode, const uint32_t* status, uint16_t* offs,
                         uint32_t* count, uint32_t* exits, uint32_t* rec_base, uint32_t* guess, uint32_t* tail_at, uint32_t* summary);
int launch_bam_decode(hipStream_t s, const uint8_t* inflated, const BgzfBlock* blocks, uint32_t n_blocks, const uint16_t* offs,
                      const uint32_t* count, const uint32_t* rec_base, const BamColumns& col, int64_t out_base,
                      int64_t rel_base, int64_t head_records, uint32_t* flags);

// ---- per-kernel timing (HIP events on the launch stream; off unless besst_prof_enable(1)) ----------
enum ProfSlot {
    // one slot per kernel (or per group of kernels that always run back to back): the names besst_prof_slot_name returns
    // are the kernels' own names, the ones a rocprofv3 trace of the same run shows
    kProfStream = 0, kProfFusedWave, kProfOrdered, kProfStitch, kProfFixup, kProfCompact,
    kProfRadixHist, kProfRadixScan, kProfRadixScatter, kProfBucketSort, kProfBucketReduce, kProfRowHeads, kProfRowScan, kProfRowReduce,
    kProfOsHist, kProfOsOffsets, kProfOsScatter, kProfOsBucket, kProfOsBucketRows, kProfOsReduce, kProfOsFixup,
    kProfMetrics, kProfScore, kProfRunGroup, kProfRunCompact, kProfRunScan, kProfRunCopy, kProfMsdPartition,
    kProfRunList, kProfRunPlace, kProfRunRows, kProfSlots
};
static_assert(kProfSlots <= 32, "besst_prof_enable takes a 32-bit slot mask");
struct ProfScope {
    hipStream_t s;
    int idx;
    ProfScope(hipStream_t stream, int slot);
    ~ProfScope();
};

// ---- record flag bits (SAM) -----------------------------------------------------------------------
constexpr uint32_t kFlagUnmapped = 0x4, kFlagMateUnmapped = 0x8, kFlagReverse = 0x10,
                   kFlagMateReverse = 0x20, kFlagRead1 = 0x40, kFlagRead2 = 0x80,
                   kFlagSecondary = 0x100;

// ---- contig table row: one 16-byte gather per lookup ---------------------------------------------
// w0 = scaffold id (bits 0..27) | direction << 28 | class << 29
struct __attribute__((aligned(16))) ContigRow {
    uint32_t w0;
    int32_t scaf_len;
    int32_t ctg_pos;
    int32_t ctg_len;
};
constexpr uint32_t kScafIdMask = (1u << 28) - 1;

// ---- classify stage geometry ------------------------------------------------------------------------
// stream_kernel: 256 threads x 4 records per sub-tile, kStreamSubTiles sub-tiles per workgroup.  A wave covers
// a GROUP of 256 consecutive records per sub-tile and publishes their candidate bits as 4 x u64 (32 bytes).
constexpr int kStreamThreads = 256;
constexpr int kStreamVec = 4;
constexpr int kStreamSubTile = kStreamThreads * kStreamVec;      // 1024 records
#ifndef BESST_STREAM_SUBTILES
#define BESST_STREAM_SUBTILES 1
#endif
constexpr int kStreamSubTiles = BESST_STREAM_SUBTILES;
constexpr int kStreamTile = kStreamSubTile * kStreamSubTiles;    // 4096 records per workgroup
constexpr int kGroup = 64 * kStreamVec;                          // 256 records per candidate-bit group
constexpr int kGroupWords = 6;                                   // u64 per group record: 4 candidate words, coverage, pad
// ordered_kernel: one workgroup per kCandGroups * 256 records (one lane of its first wave per group) -> one
// block summary each.  With the evaluation spread over the workgroup's four waves the busiest block no longer
// sets the kernel time, so the larger block wins (fewer summaries for the single-workgroup stitch): on C2
// ordered + stitch took 33.4 us with 64 groups, 37.7 us with 32.
constexpr int kCandThreads = 64;
#ifndef BESST_CAND_GROUPS
#define BESST_CAND_GROUPS 64
#endif
constexpr int kCandGroups = BESST_CAND_GROUPS;                   // groups (lanes that own one) per workgroup
constexpr int kClsTile = kCandGroups * kGroup;                   // records per summary block

// Per-block summaries of ordered_kernel, resolved by the single-workgroup stitch kernel.  Stored as planes of
// `stride` u32 (plane f of block b at p[f * stride + b]) so that stitch, one lane per block, reads them coalesced:
// as 64-byte structs the single CU running stitch spent most of its time on one cache line per lane and load.
enum SummPlane {
    kSumEmit = 0,     // tuples written to the block's local segment (head tuple included)
    kSumHas,          // block holds >= 1 record that reached CreateEdge
    kSumFirst1, kSumFirst2,   // head = first reaching record of the block
    kSumLast1, kSumLast2,     // last reaching record of the block
    kSumHeadInfo,     // bit0 accept, bit1 double call, bit2 mapq == 0, bit3 has slot
    kSumHeadSlot,     // local slot of the head's tuple
    kSumCtr0,         // the block's share of besst_counters fields 0..5 and 7 (summed by stitch: thousands of
                      // workgroups hitting the same seven device-scope atomics were the slowest part of the kernel)
    kSumChunks = kSumCtr0 + 7,   // record loop that groups runs while it emits (RunLayout): chunks it closed ...
    kSumRuns,         // ... and runs in them, + 1 for the head's provisional tuple (a run of its own); else 0
    kSumPlanes
};
struct SummView {
    uint32_t* p;
    uint32_t stride;
    __device__ __forceinline__ uint32_t& at(int plane, uint32_t b) const { return p[(size_t)plane * stride + b]; }
};

// Launch-time constants of the record loop.
struct ClassifyArgs {
    const int32_t* tid;
    const int32_t* mtid;
    const int32_t* pos;
    const int32_t* mpos;
    const uint16_t* flag;
    const uint8_t* mapq;
    const uint16_t* qlen;
    const ContigRow* table;
    const uint8_t* cls8;    // class per tid (follows the rows in the packed table)
    // one bit per record, bit i % 8 of byte i / 8: the record's mate lies on another reference (tid != mtid).  Made when the
    // records become resident (besst_dev_mate_bits: the ingest's decode, push_records); with it the record loop reads
    // `mtid` only for the lanes that hold such a record.  nullptr: the loop compares the two columns itself.
    const uint8_t* mate_bits;
    int64_t n;
    int32_t n_contigs;
    int32_t node_bits;
    double read_len;
    double ins_size_threshold;
    int64_t read_len_int;  // read_len when it is a whole number in [0, 2^31), else -1 (integer form of PosDirCalculator)
    int64_t ins_thr_int;   // ceil(ins_size_threshold): for whole numbers x, x < threshold <=> x < ins_thr_int
    int32_t min_mapq;
    int32_t rf;            // orientation == 'rf'
    int32_t detect_dup;
    int32_t extend_paths;
    int32_t no_score;
    int32_t record_path;   // 0: stream_kernel + ordered_kernel (sparse candidates), 1: fused_wave_kernel (dense)
    // fused_wave_kernel counts the sort's two stream-pass digits of every tuple it emits (PresortSpec; nullptr: off)
    uint32_t* ps_table;
    int32_t ps_rows, ps_shift;
    uint64_t ps_base;
};

// ---- sort / reduce geometry -------------------------------------------------------------------------
constexpr int kSortThreads = 256;
// Keys per thread of a sort tile (build knob).  Swept on C2 (~140 k tuples): 8 and 4 make the MSD scatter faster
// (15 -> 10 us) but push the stream past the scan-free table limit, and the row-scan launch they then need costs
// what they saved (step 112-113 vs 113-115 us); the owner partition of the sharded build, whose table is 8x smaller,
// does use 1024-tuple tiles (launch_partition).
#ifndef BESST_SORT_ITEMS
#define BESST_SORT_ITEMS 16
#endif
constexpr int kSortItems = BESST_SORT_ITEMS;
constexpr int kSortTile = kSortThreads * kSortItems;   // 4096 keys per block
constexpr int kRadixBits = 8;
constexpr int kRadix = 1 << kRadixBits;

constexpr int kRedThreads = 256;
constexpr int kRedItems = 8;
constexpr int kRedTile = kRedThreads * kRedItems;      // 2048 tuples per block

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct CEDelta {
    int count, too_long, dup, nus;
    bool keep;
};

// CreateEdge call sequence for one record (shared by the record loop, the stitch and - sharded build - the unpack
// stage, which resolves slice heads).  One record (CreateGraph.py:176-183,812-871): first call against the
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

// digit selector of the radix kernels: mode 0 = 8-bit digit at `shift`, mode 1 = owner rank of the key
struct DigitSel {
    int mode;
    int shift;
    int node_bits;
    uint32_t world;
    int bits;      // digit width in mode 0
    uint64_t base; // mode 0: the digit is taken from key - base (scaffold ids of a later library start far above 1)
};

// Owner rank of a scaffold's edges (multi-GPU key partition): multiplicative hash, then mod world.
__host__ __device__ inline uint32_t owner_of_scaffold(uint32_t scaffold_id, uint32_t world) {
    return ((scaffold_id * 2654435761u) >> 15) % world;
}

// Digit histograms of the sort's first two stream-wide passes, taken by compact_kernel while it has the keys in
// registers (saves the sort its own read of the key stream): table[rows][2][256], rows a power of two, a block adds
// its counts to row (block & (rows - 1)); digits of key - key_base at `shift` and `shift + 8`; tuples at or beyond
// `cap` (the sort's capacity) are not counted.  Filled in by sort_presort_spec for the streams whose sort uses it.
// The ordered tuple stream as the record loop and the stitch leave it - one segment of kClsTile slots per block, the
// dense position of each block's first tuple, the slot of the head tuple the stitch dropped - for a sort whose first
// stream pass reads the segments itself (and writes the dense payload on its way), so that compact_kernel need not run.
// tuples per chunk of the run-grouped form (runs.hip): one wave, BESST_RG_ROUNDS words per lane (build knob).  Measured
// on full C3 (42.7 M tuples): 16 words (1024-tuple chunks, 161 VGPRs, three waves per SIMD) 308 us for the grouping
// kernel, 8 words (five waves per SIMD, a third more runs) 231 us, 4 words 235 us - with the run sort growing.
#ifndef BESST_RG_ROUNDS
#define BESST_RG_ROUNDS 8
#endif
constexpr int kRunChunk = 64 * BESST_RG_ROUNDS;
struct SegSource {
    const uint64_t* seg_keys;
    const uint64_t* seg_payload;
    const uint32_t* offsets;
    const uint32_t* skip;
    uint32_t nblocks, tile;
    uint64_t* payload_out;
    const uint32_t* chunk_first;     // per chunk of kRunChunk dense positions: the block its first position lies in (or null)
    // the record loop grouped the runs itself (RunLayout below; null: it wrote keys): where each block's runs begin in the
    // stream-ordered run list (stitch), the block summaries, and a word that is non-zero when a block's tables overflowed
    const uint32_t* run_offsets;
    const uint32_t* summ;
    uint32_t summ_stride;
    const uint32_t* run_status;
};

// Runs of equal keys found by the record loop itself (fused_wave_kernel<true>): the wave that emits a block's tuples
// keeps the distinct keys of the last few hundred of them in a small hash table in LDS - the open CHUNK's runs, a run's
// name is its slot -, writes that name in one byte next to every tuple's payload, and no keys: nothing for a grouping
// pass to read back.  A chunk is closed (its table written out and cleared) in front of an evaluation round that would
// take it beyond kRlChunkTuples tuples or that finds kRlCloseRuns runs open - so a table never fills: at most
// kRlCloseRuns - 1 + 64 < kRlSlots keys -; the block's first reaching record (the one the stitch may still drop) is in
// no run (byte kRlNoRun) and becomes a run of its own if it stays.  What the wave leaves lies in the block's KEY segment
// (kClsTile x 8 bytes, unused in this form):
//   [kRlRid, + kClsTile)                   run byte per tuple slot
//   [kRlHeadKey, + 8)                      key of the head's tuple
//   [kRlHdr, + 8 kRlMaxChunks)             per chunk: first slot | tuples << 16 | runs << 32
//   [kRlKeys, + 8 kRlSlots kRlMaxChunks)   per chunk: the keys of its runs, dense (the slots in use, in slot order) ...
//   [kRlMeta, + 4 kRlSlots kRlMaxChunks)   ... their tuples | first slot (relative to the chunk's) << 16 ...
//   [kRlOrd, + kRlSlots kRlMaxChunks)      ... and their slots, i.e. the names the run bytes use
// More than kRlMaxChunks chunks in a block (keys that do not cluster: every round closes a chunk) set *run_status: stage 2
// then reports BESST_ROWS_RUN_OVERFLOW as it does for its own run buffers, and the pass is repeated tuple by tuple.
constexpr int kRlChunkTuples = 512;
constexpr int kRlSlots = 128;
constexpr int kRlCloseRuns = 64;
constexpr int kRlMaxChunks = 64;
constexpr uint32_t kRlNoRun = 255u;
constexpr unsigned long long kRlEmpty = ~0ull;           // (a key is below 2^59)
constexpr size_t kRlRid = 0;
constexpr size_t kRlHeadKey = (size_t)kClsTile;
constexpr size_t kRlHdr = kRlHeadKey + 64;
constexpr size_t kRlKeys = kRlHdr + 8 * (size_t)kRlMaxChunks;
constexpr size_t kRlMeta = kRlKeys + 8 * (size_t)kRlSlots * kRlMaxChunks;
constexpr size_t kRlOrd = kRlMeta + 4 * (size_t)kRlSlots * kRlMaxChunks;
static_assert(kRlOrd + (size_t)kRlSlots * kRlMaxChunks <= (size_t)kClsTile * 8, "the run tables live in the block's key segment");
static_assert(kClsTile <= 65536 && kRlChunkTuples < 65536 && kRlSlots <= (int)kRlNoRun, "slots, counts and run names are packed");
static_assert(kRlCloseRuns - 1 + 64 < kRlSlots, "a chunk's table must never fill");

struct PresortSpec {
    uint32_t* table;
    int rows, shift;
    uint64_t key_base;
    uint32_t cap;
    int in_record_loop;      // the fused record loop counts while it emits (the head tuples the stitch drops are taken
                             // out again); else compact_kernel counts.  Out: 2 = the loop handed its segments over
                             // WITHOUT counting (`count` was 0), 3 = it grouped the runs itself (seg.run_offsets)
    int count;               // in: stage 2 will want the digit histograms (0: it groups runs and never reads them)
    int segmented;           // in: the sort can read block segments; out: it has to (compact_kernel did not run, `seg`)
    SegSource seg;
};

// ---- stage launchers (defined in the .hip files) -----------------------------------------------------
size_t classify_workspace_bytes(int64_t n);
int launch_classify(hipStream_t s, const ClassifyArgs& a, int32_t* carry, int64_t* aligned,
                    uint64_t* keys, uint64_t* payload, uint32_t* n_out, besst_counters* counters,
                    void* ws, size_t ws_bytes, PresortSpec* presort = nullptr);
// the same split in three phases for the multi-GPU path (the duplicate chain crosses rank boundaries)
int launch_classify_scan(hipStream_t s, const ClassifyArgs& a, int64_t* aligned, besst_counters* counters,
                         void* ws, size_t ws_bytes, bool group_runs = false);
// true: launch_classify_scan(..., group_runs = true) on these arguments leaves runs (RunLayout) instead of keys
bool classify_can_group_runs(const ClassifyArgs& a);
int launch_candidate_density(hipStream_t s, int64_t n, const int32_t* tid, const int32_t* mtid, int64_t sample_records,
                             unsigned long long* counts);
// bits of records [lo, hi) (whole bytes: from lo rounded down to hi rounded up to a multiple of 8, clipped to n: the records in
// front of lo that share lo's byte are read again)
int launch_mate_bits(hipStream_t s, const int32_t* tid, const int32_t* mtid, int64_t lo, int64_t hi, int64_t n, uint8_t* bits);
int launch_classify_tail(hipStream_t s, int64_t n, int32_t* tail, void* ws, size_t ws_bytes);
int launch_resolve_carry(hipStream_t s, const int32_t* tails, int rank, int32_t* carry);
int launch_classify_emit(hipStream_t s, int64_t n, int detect_dup, int32_t* carry, uint64_t* keys,
                         uint64_t* payload, uint32_t* n_out, besst_counters* counters, void* ws,
                         size_t ws_bytes, const uint8_t* cls8, int32_t n_contigs, int64_t* aligned,
                         const int32_t* tails, int rank, int32_t* slice_info, PresortSpec* presort = nullptr);

size_t reduce_workspace_bytes(int64_t cap);
// true (and *out filled in): launch_sort_reduce with these arguments takes its histograms from out->table when called
// with hist_ready
bool sort_presort_spec(int64_t cap, int key_bits, uint64_t key_base, void* ws, size_t ws_bytes, PresortSpec* out);
bool onesweep_presort_spec(int64_t cap, int key_bits, uint64_t key_base, void* ws, PresortSpec* out);
int launch_sort_reduce(hipStream_t s, int64_t cap, const uint32_t* n_tuples, int key_bits,
                       const uint64_t* keys, const uint64_t* payload, uint64_t* row_key,
                       uint32_t* row_mask, uint32_t* row_n, int64_t* row_sum, int64_t* row_sum_sq,
                       uint32_t* row_first, uint32_t* row_offset, int32_t* obs_lo, int32_t* obs_hi,
                       uint32_t* n_rows, void* ws, size_t ws_bytes, const uint32_t* first_map = nullptr,
                       uint64_t key_base = 0, bool hist_ready = false, const SegSource* seg = nullptr,
                       uint32_t flags = 0 /* BESST_REDUCE_* */);
// run-grouped form for large streams (runs.hip): the chunks' runs of equal keys are sorted, not the tuples.  grouped:
// cap words for the payload grouped by run; staged_rows: cap 40-byte records (the chained-scan workspace's row staging)
bool runs_enabled(int64_t cap);
size_t runs_workspace_bytes(int64_t cap);
int launch_runs_reduce(hipStream_t s, int64_t cap, const uint32_t* n_tuples, int key_bits, const uint64_t* keys,
                       const uint64_t* payload, const SegSource* seg, uint64_t* grouped, void* staged_rows,
                       uint64_t* row_key, uint32_t* row_mask, uint32_t* row_n, int64_t* row_sum, int64_t* row_sum_sq,
                       uint32_t* row_first, uint32_t* row_offset, int32_t* obs_lo, int32_t* obs_hi, uint32_t* n_rows,
                       void* ws, size_t ws_bytes, const uint32_t* first_map, uint64_t key_base);
void* onesweep_staged_rows(void* ws, int64_t cap);
// chained-scan sort + atomic-free reduction for large streams (onesweep.hip); buf_* = the ping-pong buffers of the
// sort/reduce workspace
size_t onesweep_workspace_bytes(int64_t cap);
int launch_onesweep_sort_reduce(hipStream_t s, int64_t cap, const uint32_t* n_tuples, int key_bits,
                                const uint64_t* keys, const uint64_t* payload, uint64_t* buf_keys[2],
                                uint32_t* buf_idx[2], uint64_t* row_key, uint32_t* row_mask, uint32_t* row_n,
                                int64_t* row_sum, int64_t* row_sum_sq, uint32_t* row_first, uint32_t* row_offset,
                                int32_t* obs_lo, int32_t* obs_hi, uint32_t* n_rows, void* ws, size_t ws_bytes,
                                const uint32_t* first_map, uint64_t key_base, bool hist_ready = false,
                                const SegSource* seg = nullptr);
size_t exchange_region_bytes(int64_t pair_cap);
size_t exchange_stride_bytes(int64_t pair_cap, int64_t rider_bytes);
int launch_partition(hipStream_t s, int64_t cap, const uint32_t* n_tuples, int node_bits, int world,
                     const uint64_t* keys, const uint64_t* payload, int64_t pair_cap, void* send, void* ws,
                     size_t ws_bytes, const void* rider, int64_t rider_bytes, const int32_t* slice_info);
int launch_unpack(hipStream_t s, int world, int64_t pair_cap, const void* recv, uint64_t* keys, uint64_t* payload,
                  uint32_t* gidx, uint32_t* n_out, uint32_t* overflow, void* rider_sum, int64_t rider_bytes,
                  int speculative, int rank, int detect_dup, int32_t* all_info, besst_counters* counters);

struct MetricsArgs {
    const int32_t* tid;
    const int32_t* mtid;
    const int32_t* tlen;
    const uint16_t* flag;
    const uint8_t* mapq;
    const uint8_t* top_mask;   // device, n_contigs bytes
    int64_t n;
    int32_t n_contigs;
    int32_t rf;
    int32_t min_mapq;
    double read_len;
};
size_t metrics_workspace_bytes(int64_t n);
int launch_metrics(hipStream_t s, const MetricsArgs& a, int64_t start, int64_t count,
                   int32_t* isize_out, int32_t* contam_out, int64_t* state, void* ws, size_t ws_bytes,
                   bool count_only = false);
int launch_value_histogram(hipStream_t s, const int32_t* values, int64_t n, int64_t n_bins,
                           unsigned long long* hist, unsigned long long* overflow);
int launch_stream_order(hipStream_t s, const int32_t* tid, const int32_t* pos, int64_t n, unsigned long long* first_unsorted);

struct ScoreArgs {
    const uint32_t* row;       // edge-table row per scored edge
    const uint8_t* swap;
    const int32_t* len1;
    const int32_t* len2;
    const uint32_t* row_n;
    const int64_t* row_sum;
    const uint32_t* row_offset;
    const int32_t* obs_lo;
    const int32_t* obs_hi;
    double mean, sigma, read_len;
    int64_t n_edges;
};
size_t score_workspace_bytes(int64_t n_edges, int64_t n_tuples);
int launch_gap_table(hipStream_t s, double mean, double sigma, double r, double c_len, int32_t d_lower, int32_t n, double* out);
int launch_score(hipStream_t s, const ScoreArgs& a, double* gap, double* sd0, int32_t* ks_h,
                 uint8_t* flags, void* ws, size_t ws_bytes);
// log-normal branch of GiveScoreOnEdges (CreateGraph.py:485-494,522-531): the pmf's prefix tables and the gap scan
struct LogNormalArgs {
    double mu, sigma;          // param.lognormal_mean / lognormal_sigma
    int64_t x_max;             // support of the pmf: 1 .. x_max
    const double* F0;          // x_max + 1 entries each (launch_lognormal_tables)
    const double* F1;
    int32_t max_gap;           // len(conditional_stddevs) - 1
};
size_t lognormal_tables_workspace_bytes(int64_t x_max);
int launch_lognormal_tables(hipStream_t s, double mu, double sigma, int64_t x_max, double* F0, double* F1, void* ws,
                            size_t ws_bytes);
int launch_score_lognormal(hipStream_t s, const ScoreArgs& a, const LogNormalArgs& l, double* gap, double* sd0,
                           int32_t* ks_h, uint8_t* flags, void* ws, size_t ws_bytes);
int launch_conditional_stddevs(hipStream_t s, const double* f, int64_t max_isize, const int32_t* steps, int32_t n_steps,
                               double* out);

}  // namespace besst
