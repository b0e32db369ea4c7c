/*
 * besst_amd.h - C ABI of the MI355X scaffold-graph construction library (libbesst_amd.so)
 *
 * This is the drop-in boundary for BESST's hot path
 *     bam_parser -> libmetrics/find_bimodality -> CreateGraph/e_nr_links
 * The reference has no plugin registry: the seam is two Python calls per library,
 *     libmetrics.get_metrics(bam_file, param, Information)          runBESST:168
 *     CreateGraph.PE(Contigs, Scaffolds, Information, C_dict, param,
 *                    small_contigs, small_scaffolds, bam_file)      runBESST:182
 * and its only native precedent is a ctypes call with a caller-allocated result struct and
 * an int return (BESST/diploid/wrapper_sw.py:12-24, swmodule.cpp:20-28,44).  The entry points
 * below follow that convention: extern "C", plain pointers and sizes, int status (0 = ok),
 * never throw, never exit.  besst_amd/_lib.py binds them with ctypes; INTEGRATION.md shows
 * the stub a BESST maintainer would add.
 *
 * Two layers:
 *   besst_ctx_*  host-buffer API.  The context owns all device memory; inputs are host
 *                (numpy) buffers, results come back in caller-allocated host buffers sized by
 *                a preceding *_count call.  This is what the Python drop-in uses.
 *   besst_dev_*  device-pointer API.  Every pointer is an HBM pointer owned by the caller
 *                (e.g. a torch tensor's data_ptr()), work is enqueued on the given hipStream_t
 *                and nothing is allocated or synchronised.  Used by bench.py and by the
 *                multi-GPU path, where the RCCL exchange sits between two besst_dev_ stages.
 *
 * Threading: one caller thread per context (like the reference's single-threaded loop).
 */
#ifndef BESST_AMD_H
#define BESST_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 1: rounds 1-4.  2: + the sharded-scan and BAM-slice entry points of round 5 and the log-normal scoring entry points of
 * round 6 (additions only: a caller built against 1 keeps working).  3: besst_lib_params grew by `mate_bits` (a caller
 * built against 2 passes a shorter struct: recompile), + besst_dev_mate_bits. */
#define BESST_ABI_VERSION 3

/* status codes */
#define BESST_OK 0
#define BESST_ERR_ARG 1     /* bad argument (null pointer, misaligned column, size out of range) */
#define BESST_ERR_HIP 2     /* a HIP runtime call failed; see besst_last_error() */
#define BESST_ERR_STATE 3   /* call order violated (e.g. build before set_library) */
#define BESST_ERR_NOMEM 4
#define BESST_ERR_UNSUPPORTED 5 /* valid input in a form this entry point does not handle; nothing was changed - the
                                * comment of the function names the general one to call instead */

/* Stage 2 (besst_dev_reduce*) reports a table it could not build in the size word the caller reads anyway: *n_rows is
 * one of these instead of a row count, and no other output of the call is valid.  (The besst_ctx_* layer turns the first
 * into BESST_ERR_HIP and handles the second itself.) */
#define BESST_ROWS_SORT_FAILED 0xFFFFFFFFu  /* a wait between workgroups inside a sort launch gave up (chained-scan look-back
                                             * for the predecessor tile; small streams: a partition tile for the others' counts) */
#define BESST_ROWS_RUN_OVERFLOW 0xFFFFFFFEu /* run-grouped form: more runs of equal keys than its buffers hold (a stream
                                             * whose keys do not cluster); repeat the call with BESST_REDUCE_NO_RUNS   */
/* flags of besst_dev_reduce_flags / besst_presort.flags */
#define BESST_REDUCE_NO_RUNS 1u             /* large streams: sort the tuples (chained-scan passes), not runs of them   */

/* contig classes (membership in Contigs / small_contigs, CreateGraph.py:127-130) */
#define BESST_CLS_ABSENT 0
#define BESST_CLS_LARGE 1
#define BESST_CLS_SMALL 2

/* edge-table membership bits: which graph(s) CreateEdge was called for (CreateGraph.py:170-206) */
#define BESST_MASK_G 1u
#define BESST_MASK_GPRIME 2u

/* Per-library constants of the record loop.  Mirrors the fields CreateGraph.PE / CreateEdge read
 * from Parameter.parameter (CreateGraph.py:138,169,175-184,830-840). */
typedef struct besst_lib_params {
    double read_len;           /* param.read_len: may be fractional when inferred (libmetrics.py:265) */
    double ins_size_threshold; /* param.ins_size_threshold (-T), float when inferred                 */
    int32_t min_mapq;          /* param.min_mapq (--min_mapq, default 11)                            */
    int32_t orientation;       /* 0 = 'fr' (PosDirCalculatorPE), 1 = 'rf' (PosDirCalculatorMP)       */
    int32_t detect_duplicate;  /* param.detect_duplicate (-d, default on)                            */
    int32_t extend_paths;      /* param.extend_paths (-y, default on)                                */
    int32_t no_score;          /* param.no_score (--no_score)                                        */
    int32_t record_path;       /* which form of the record loop runs (results are identical): 0 = two passes, the
                                * second over the tid != mtid records only (paired-end libraries: ~1 % of the
                                * records); 1 = one fused pass (mate-pair libraries: ~20 %); pick it from
                                * besst_dev_candidate_density() - the besst_ctx_* layer does that itself          */
    const void* mate_bits;     /* besst_dev_* layer (the besst_ctx_* layer keeps its own): one bit per record of the stream
                                * the call is given - bit i % 8 of byte i / 8 set iff tid[i] != mtid[i] -, made by
                                * besst_dev_mate_bits when the records became resident; the record loop then reads `mtid`
                                * only for the lanes that hold such a record (CreateGraph.py:141-169: everything but the
                                * coverage sum asks for contig1 != contig2).  NULL: the loop compares the columns itself.
                                * (ABI version 3: the struct grew by this member.)                                 */
} besst_lib_params;

/* Tallies of the record loop: Parameter.counters (Parameter.py:113-124) plus the fishy-read count
 * `ctr` of CreateGraph.py:100,163 and the sizes of the emitted tuple stream. */
typedef struct besst_counters {
    int64_t count;                      /* counter.count ("USEFUL READS", per CreateEdge call)      */
    int64_t non_unique;                 /* counter.non_unique                                        */
    int64_t non_unique_for_scaf;        /* counter.non_unique_for_scaf                               */
    int64_t nr_of_duplicates;           /* counter.nr_of_duplicates                                  */
    int64_t reads_with_too_long_insert; /* counter.reads_with_too_long_insert                        */
    int64_t fishy_reads;                /* ctr                                                       */
    int64_t n_tuples;                   /* link + fishy tuples written to the sort buffer            */
    int64_t n_reach;                    /* records that reached CreateEdge                           */
    int32_t prev_obs1;                  /* counter.prev_obs1 after the last record                   */
    int32_t prev_obs2;                  /* counter.prev_obs2                                         */
} besst_counters;

/* Library-statistics sampling pass (libmetrics.py:49-131,283-304).  All counts are over the scanned
 * prefix of the stream, honouring the reference's 1,000,000-sample cut-offs in stream order. */
typedef struct besst_metrics_counts {
    int64_t n_isize;        /* observations collected for the insert-size sample (<= 1,000,000)     */
    int64_t n_contam;       /* opposite-orientation fragments collected (contamination_reads)       */
    int64_t counter_total;  /* mapped records on the 1000 longest contigs within the cut-off       */
    int64_t sample_counter; /* records on the 1000 longest contigs within the cut-off (<= 1,000,000) */
    int64_t records_scanned;
} besst_metrics_counts;

typedef struct besst_ctx besst_ctx;

/* ------------------------------------------------------------------------------------------------
 * library
 * ---------------------------------------------------------------------------------------------- */
int besst_abi_version(void);
/* Last error text of the calling thread's most recent failing call (never NULL). */
const char* besst_last_error(void);

/* The pinned staging buffers of besst_ctx_push_bam / besst_ctx_push_bam_device are kept in a process-wide pool between
 * calls (pinning and unpinning host memory costs ~90 ms per GB: a third of the ingest of a 40 M-record file); at most 1 GiB
 * stays cached.  This call frees what is idle in the pool (a long-lived host process that is done reading files). */
void besst_release_cached_memory(void);

/* Number of visible HIP devices, or a negative status. */
int besst_device_count(void);

/* Per-kernel timing with HIP events recorded on the launch stream (bench.py's roofline numbers).
 * Off by default; slot_mask selects the kernels to time (bit i = slot i, 0 = off).
 * besst_prof_collect synchronises the recorded events, sums elapsed milliseconds and
 * launch counts per slot (besst_prof_slots() entries, names from besst_prof_slot_name) and resets.
 * besst_prof_sample_every(n) times only every n-th launch of an enabled slot: an event pair costs the stream
 * ~3 us, which a 120 us step notices when every launch is timed. */
void besst_prof_enable(uint32_t slot_mask);
void besst_prof_sample_every(uint32_t n);
int besst_prof_slots(void);
const char* besst_prof_slot_name(int slot);
int besst_prof_collect(int n_slots, double* ms, int64_t* launches);

/* ------------------------------------------------------------------------------------------------
 * host-buffer API (context owns HBM)
 * ---------------------------------------------------------------------------------------------- */
besst_ctx* besst_ctx_create(int device);
void besst_ctx_destroy(besst_ctx* ctx);

/* Contig table, one row per BAM header entry (tid order).  Replaces the Contigs / small_contigs /
 * Scaffolds / small_scaffolds lookups of the record loop (CreateGraph.py:127-130,144-174,819-829).
 * scaf_id must be in [1, 2^28). */
int besst_ctx_set_contigs(besst_ctx* ctx, int64_t n_contigs, const int32_t* scaf_id,
                          const int32_t* scaf_len, const int32_t* ctg_pos, const int32_t* ctg_len,
                          const uint8_t* direction, const uint8_t* cls);

int besst_ctx_set_library(besst_ctx* ctx, const besst_lib_params* params);

/* Drop all records held by the context (start of a new library). */
int besst_ctx_clear_records(besst_ctx* ctx);

/* Append a batch of alignment records (SoA columns = the pysam attributes the hot path reads:
 * rname mrnm pos mpos tlen flag mapq qlen).  Records stay resident in HBM. */
int besst_ctx_push_records(besst_ctx* ctx, int64_t n, const int32_t* tid, const int32_t* mtid,
                           const int32_t* pos, const int32_t* mpos, const int32_t* tlen,
                           const uint16_t* flag, const uint8_t* mapq, const uint16_t* qlen);

/* The resident records back on the host (any column pointer may be NULL): what a caller that ingested a file straight
 * into HBM (below) needs to look at the records themselves; also how the tests compare the two ingest forms. */
int besst_ctx_record_count(besst_ctx* ctx, int64_t* n_records);
/* The resident columns where they lie: eight HBM pointers (tid mtid pos mpos tlen flag mapq qlen) for the besst_dev_*
 * layer - a rank of the sharded build works on the slice its besst_ctx_push_bam_device_part left here, no copy.  Valid
 * until the context's records change or the context is destroyed. */
int besst_ctx_record_pointers(besst_ctx* ctx, int64_t* n_records, uint64_t* ptrs);
int besst_ctx_fetch_records(besst_ctx* ctx, int64_t first, int64_t n, int32_t* tid, int32_t* mtid, int32_t* pos,
                            int32_t* mpos, int32_t* tlen, uint16_t* flag, uint8_t* mapq, uint16_t* qlen);

/* The same from a BAM file, streamed (replaces `pysam.Samfile(param.bamfile, 'rb')` + the iteration of runBESST:162,
 * CreateGraph.py:111, libmetrics.py:63,257,293): the reader's host threads inflate and decode chunk k + 1 into pinned
 * staging columns while chunk k's asynchronous copies travel to HBM.  `bam` is an open besst_bam (below) positioned at
 * its first unread record; reads to the end of the file.  The first head_records records' rlen / alen / qlen - all
 * libmetrics' read-length step looks at (libmetrics.py:246-273: 1000 records) - are returned in the head_* arrays.
 * chunk_records <= 0 picks 4 Mi records (two staging sets of 19 B per record). */
typedef struct besst_bam besst_bam;
typedef struct {
    int64_t records;             /* records appended to the context */
    int64_t chunks;
    int64_t bytes_h2d;
    double seconds;              /* wall time of the call */
    double decode_seconds;       /* of which the calling thread spent in the reader (inflate + record decode; device form:
                                  * copying the compressed bytes off the mapping into pinned staging) */
    double copy_wait_seconds;    /* and blocked on copies (device form: and kernels) that had not finished */
    int64_t inflated_bytes;      /* device form: bytes the BGZF blocks inflated to */
    int64_t blocks;              /* device form: BGZF blocks */
    int32_t on_device;           /* 1: inflate + record decode ran on the GPU (bytes_h2d = the compressed bytes) */
    int32_t starts_repaired;     /* device form: blocks whose guessed first record start the verification replaced (0 in htslib's layout) */
} besst_ingest_stats;
int besst_ctx_push_bam(besst_ctx* ctx, besst_bam* bam, int64_t chunk_records, int64_t head_records, int32_t* head_rlen,
                       int32_t* head_alen, uint16_t* head_qlen, besst_ingest_stats* stats);

/* The same with BGZF inflate, record walk and record decode ON THE GPU (csrc/bgzf_gpu.hip): the file's compressed bytes
 * are what crosses PCIe - staged through pinned memory chunk by chunk (chunk_blocks BGZF blocks, <= 0: 8192), chunk
 * j + 1 uploaded while chunk j inflates (one wave per block).  ANY block layout: htslib's (samtools, bwa | samtools, this
 * library's writer), where every BGZF block begins with a record, and htsjdk's / Picard's, where blocks are cut wherever
 * 64 KiB of data end - a chunk's blocks are inflated back to back, the first record start of every block is guessed from
 * the bytes and VERIFIED from block to block (a wrong guess is replaced by what the block before it says and the block
 * is walked again), and the record that a chunk does not finish travels in front of the next chunk's first block (at most
 * 4 MiB).  Returns BESST_ERR_UNSUPPORTED - context and reader unchanged - for a block the device does not inflate or
 * whose CRC-32 does not match its gzip trailer and for record starts that cannot be established: call besst_ctx_push_bam
 * then (which checks the CRC-32 too and reports the file as corrupt, as htslib would). */
int besst_ctx_push_bam_device(besst_ctx* ctx, besst_bam* bam, int64_t chunk_blocks, int64_t head_records, int32_t* head_rlen,
                              int32_t* head_alen, uint16_t* head_qlen, besst_ingest_stats* stats);

/* The same for one PART of the file's records - multi-GPU ingest: rank r of W calls it with (r, W) and holds the r-th
 * contiguous slice of the stream, the slice phase 1 of the sharded graph build works on.  The file is
 * cut at the BGZF block boundaries nearest to part / parts of its bytes; every caller finds the same boundaries on its own
 * (gzip magic + BC subfield + two chained blocks behind it).  The head_* arrays describe the part's first records.
 * parts > 1 needs htslib's layout, where every block boundary is a record boundary: BESST_ERR_UNSUPPORTED (context and
 * reader unchanged) for a file whose records straddle blocks - such files take besst_ctx_push_bam_device_slice. */
int besst_ctx_push_bam_device_part(besst_ctx* ctx, besst_bam* bam, int32_t part, int32_t parts, int64_t chunk_blocks,
                                   int64_t head_records, int32_t* head_rlen, int32_t* head_alen, uint16_t* head_qlen,
                                   besst_ingest_stats* stats);

/* Slice `part` of `parts` of a file in ANY block layout (multi-GPU ingest of files whose records straddle BGZF blocks:
 * htsjdk / Picard).  The slices are cut at block boundaries as above, and a record belongs to the slice it BEGINS in.  Where
 * a slice's first record begins is the one thing a rank cannot know alone:
 *   first_skip < 0   the rank guesses (the heuristics of the block-to-block verification; everything behind the guess is
 *                    verified as usual);
 *   first_skip >= 0  the first record begins that many inflated bytes behind the slice's first block's first byte.
 * boundary[0] reports the offset used, boundary[1] how many bytes of the slice's last record lie in the next slice (they
 * are read from the blocks that follow, at most 4 MiB).  The callers exchange the two numbers: slice r is right iff
 * boundary[0] of slice r equals boundary[1] of slice r - 1 (slice 0 begins behind the header and is always right); a slice
 * whose guess was wrong is read again - into a fresh context - with first_skip = boundary[1] of the slice before
 * (besst_amd.distributed.ingest_slice runs this protocol).  In htslib's layout both numbers are 0. */
int besst_ctx_push_bam_device_slice(besst_ctx* ctx, besst_bam* bam, int32_t part, int32_t parts, int64_t chunk_blocks,
                                    int64_t first_skip, int64_t* boundary, int64_t head_records, int32_t* head_rlen,
                                    int32_t* head_alen, uint16_t* head_qlen, besst_ingest_stats* stats);

/* Test hook of the device inflate: the BGZF blocks of `bgzf` (n_bytes, host) inflated on `device`, their output
 * concatenated in out (capacity out_cap, length in *out_len).  BESST_ERR_UNSUPPORTED when a block does not inflate
 * (its index and status in besst_last_error()). */
int besst_bgzf_inflate_device(int device, const void* bgzf, size_t n_bytes, void* out, size_t out_cap, size_t* out_len);

/* libmetrics sampling (replaces the three `for read in bam_file` scans, libmetrics.py:63,257,293).
 * top_mask[tid] != 0 marks the 1000 longest references.  orientation/min_mapq/read_len as in
 * besst_lib_params.  isize_out / contam_out receive |tlen| of the qualifying records in stream order
 * (capacity 1,000,000 each); the host adds 2*read_len where the reference does. */
int besst_ctx_metrics_sample(besst_ctx* ctx, const uint8_t* top_mask, int32_t orientation,
                             int32_t min_mapq, double read_len, int32_t want_isize,
                             int32_t* isize_out, int32_t* contam_out, besst_metrics_counts* counts);

/* Is the resident stream sorted by coordinate?  Replaces the reference's index check - `bam_file.fetch(cont_names[0])`
 * raising for a BAM without an index, BESST/libmetrics.py:237-241; an index exists only for a coordinate-sorted file.
 * *first_unsorted: -1, or the index of the first record whose (reference id, position) lies in front of its predecessor's
 * (reference -1 sorts last, as `samtools sort` leaves it).  The graph build is exact on any order but several times slower
 * on an unsorted stream; libmetrics.get_metrics turns a hit into the reference's message as a WARNING.  first_key /
 * last_key (each 2 x int32, may be NULL): (tid, pos) of the first and last resident record, for the slices of a sharded
 * stream to compare across ranks.  besst_dev_stream_order: the same pass on caller-owned columns, *first_unsorted on the
 * device. */
int besst_ctx_stream_order(besst_ctx* ctx, int64_t* first_unsorted, int32_t* first_key, int32_t* last_key);
int besst_dev_stream_order(void* stream, int64_t n, const int32_t* tid, const int32_t* pos, int64_t* first_unsorted);

/* Count-per-value histogram of a sample on the device (input to find_bimodality.split_distribution,
 * find_bimodality.py:109-132).  hist_out has n_bins entries; values >= n_bins are counted in
 * *overflow. */
int besst_ctx_value_histogram(besst_ctx* ctx, const int32_t* values, int64_t n, int64_t n_bins,
                              int64_t* hist_out, int64_t* overflow);

/* Record loop + CreateEdge + edge-table reduction (CreateGraph.py:111-211,812-871) over every record
 * pushed so far. */
int besst_ctx_build_graph(besst_ctx* ctx);

/* Sizes of the result: edge rows (distinct (node pair, fishy) keys) and link/fishy tuples. */
int besst_ctx_edge_count(besst_ctx* ctx, int64_t* n_rows, int64_t* n_tuples);

/* Edge rows sorted by key.  key = ((min_node << node_bits) | max_node) << 1 | is_fishy with
 * node = scaffold_id * 2 + (side == 'R').  For link rows: n = nr_links, sum_obs = obs,
 * sum_obs_sq = obs_sq, mask = BESST_MASK_* bits; for fishy rows n = fishy count.  first_idx is the
 * position of the row's first tuple in the emitted tuple stream (monotone in BAM order), offset the
 * start of the row's slice in the observation arrays. */
int besst_ctx_fetch_edges(besst_ctx* ctx, uint64_t* key, uint32_t* mask, uint32_t* n,
                          int64_t* sum_obs, int64_t* sum_obs_sq, uint32_t* first_idx,
                          uint32_t* offset, int32_t* node_bits);

/* Per-tuple observations grouped by edge row, BAM order inside a row: obs_lo belongs to the row's
 * min node, obs_hi to its max node; observations[i] = obs_lo[i] + obs_hi[i] (CreateGraph.py:842-862). */
int besst_ctx_fetch_observations(besst_ctx* ctx, int32_t* obs_lo, int32_t* obs_hi);
/* The same as ONE column: out[i] = obs_lo[i] + obs_hi[i], the entries of an edge's `observations` list
 * (BESST/CreateGraph.py:849,862) - summed on the device, half the bytes across PCIe.  It works on a stream and a buffer of
 * its own and is the one call of the ctx layer that may run on a second host thread while the caller's thread goes on with
 * besst_ctx_score_edges / the other fetches (the Python drop-in starts it in the background when the table has been built
 * and joins it when an edge's observations are first read, or the session closes). */
int besst_ctx_fetch_observation_sums(besst_ctx* ctx, int32_t* out);

/* cont_aligned_len numerators (CreateGraph.py:138-139), one int64 per tid. */
int besst_ctx_fetch_coverage(besst_ctx* ctx, int64_t* aligned);

int besst_ctx_fetch_counters(besst_ctx* ctx, besst_counters* out);

/* GiveScoreOnEdges, normal-distribution branch (CreateGraph.py:498-614), for n_edges rows of the
 * table built by the last besst_ctx_build_graph.  row[i] indexes the edge table; swap[i] != 0 means
 * the graph iterates the edge as (max_node, min_node), i.e. l1 comes from obs_hi; len1/len2 are the
 * scaffold lengths of the first/second endpoint in that orientation.  Outputs per edge:
 *   gap      int(gap)                      (CreateGraph.py:511-541)
 *   sd0      expected std-dev tr_sk_std_dev or 2^32 (:548-558)
 *   ks_h     integer h with KS statistic = h / n (:582-606)
 *   flags    bit0: both scaffolds > 2 sigma (ML gap used), bit1: -gap > len (score forced to 0) */
int besst_ctx_score_edges(besst_ctx* ctx, int64_t n_edges, const uint32_t* row, const uint8_t* swap,
                          const int32_t* len1, const int32_t* len2, double mean, double sigma,
                          double read_len, double* gap, double* sd0, int32_t* ks_h, uint8_t* flags);

/* GiveScoreOnEdges, log-normal branch (param.lognormal: a skewed library; CreateGraph.py:485-494,522-531,549-553).  As
 * besst_ctx_score_edges, but an edge whose scaffolds are both longer than 2 sigma gets the gap of
 * mathstats.log_normal_param_est.GapEstimator(ln_mu, ln_sigma, read_len, observations, len1, c2_len=len2) (CreateGraph.py:526;
 * un-vendored, restated in besst_amd/mathstats_compat.py): the integer d maximising
 *     L(d) = sum_i log f(o_i + d) - n log g(d),   f = log-normal pmf on 1 .. x_max,   g(d) = sum_x w(x; d) f(x)
 * over the gaps that keep every o_i + d inside the support - coarse scan with stride 64, then the 129 gaps around the
 * coarse optimum - clamped to max_gap = len(conditional_stddevs) - 1 (:527-528).  The pmf's prefix tables are built on
 * the device at the first call and kept while (ln_mu, ln_sigma, x_max) stay the same.  x_max is passed in (the caller's
 * int(min(exp(ln_mu + 6 ln_sigma), 4e6))) so that host and device agree on the support.  There is no sd0 output: the
 * caller indexes the conditional sigma table (besst_ctx_conditional_stddevs) with the gap (:549-553). */
int besst_ctx_score_edges_lognormal(besst_ctx* ctx, int64_t n_edges, const uint32_t* row, const uint8_t* swap,
                                    const int32_t* len1, const int32_t* len2, double mean, double sigma,
                                    double read_len, double ln_mu, double ln_sigma, int64_t x_max, int32_t max_gap,
                                    double* gap, int32_t* ks_h, uint8_t* flags);

/* get_conditional_stddevs (CreateGraph.py:436-469): out[k] = sigma of the density f(x) * max(0, x - steps[k] + 1) over
 * x = 0 .. max_isize, where density[x] = f(x) is param.empirical_distribution laid out densely (0 where it has no entry).
 * The caller repeats out[k] for the gaps up to the next step as the reference does.  Host pointers. */
int besst_ctx_conditional_stddevs(besst_ctx* ctx, const double* density, int64_t max_isize, const int32_t* steps,
                                  int32_t n_steps, double* out);

/* ------------------------------------------------------------------------------------------------
 * BAM front-end (host): BGZF inflate + record decode into the SoA columns, replacing the
 * `pysam.Samfile(param.bamfile, 'rb')` iteration of runBESST:162 / CreateGraph.py:111 /
 * libmetrics.py:63,257,293.  qlen = query_alignment_length, rlen = query_length, alen = reference_length
 * (pysam 0.8 attribute names).  n_threads inflate workers (libdeflate if present, else zlib).
 * ---------------------------------------------------------------------------------------------- */
besst_bam* besst_bam_open(const char* path, int n_threads);
void besst_bam_close(besst_bam* bam);
int64_t besst_bam_n_references(const besst_bam* bam);
const char* besst_bam_reference_name(const besst_bam* bam, int64_t index);
/* All names at once, NUL-terminated and back to back (a header of 2 M contigs is 2 M calls otherwise): returns the bytes
 * needed; buf is filled when cap is at least that. */
int64_t besst_bam_reference_names(const besst_bam* bam, char* buf, int64_t cap);
int besst_bam_reference_lengths(const besst_bam* bam, int32_t* out);
/* Returns the number of records decoded (0 at end of file) or a negative status.  qlen is a 16-bit column: an aligned
 * query longer than 65535 bases (no paired short read is) is stored as 65535 and counted. */
int64_t besst_bam_clamped_records(const besst_bam* bam);
int64_t besst_bam_read_records(besst_bam* bam, int64_t max_records, int32_t* tid, int32_t* mtid, int32_t* pos,
                               int32_t* mpos, int32_t* tlen, uint16_t* flag, uint8_t* mapq, uint16_t* qlen,
                               int32_t* rlen, int32_t* alen);
/* Test / bench scaffolding, not part of the graph path: the columns as a BAM file in htslib's block layout (name
 * "r<index>", CIGAR [clip S] qlen M, rlen bases) - what tests and bench.py read back through the reader above.
 * level: zlib's 0..9; + 16: pseudo-random bases and slowly changing qualities (a file that compresses like a sequencer's,
 * ~3 x) instead of constant bytes (~13 x). */
int besst_bam_write_records(const char* path, int64_t n_ref, const char* const* ref_names, const int32_t* ref_lengths,
                            int64_t n, const int32_t* tid, const int32_t* mtid, const int32_t* pos, const int32_t* mpos,
                            const int32_t* tlen, const uint16_t* flag, const uint8_t* mapq, const uint16_t* qlen,
                            const int32_t* rlen, int n_threads, int level);

/* ------------------------------------------------------------------------------------------------
 * Host float finishing of libmetrics (no GPU): the statistics on the <= 1,000,000 sampled insert sizes,
 * replayed in the reference's exact operation order (libmetrics.py:22-28,88-110,141-223,316-343) so the
 * doubles are bit-identical to CPython's.  values = abs(tlen) in BAM order; is_float/offset select the
 * float form abs(tlen) + offset (offset = 2*read_len).
 * ---------------------------------------------------------------------------------------------- */
int besst_host_isize_stats(const int32_t* values, int64_t n, int32_t is_float, double offset,
                           int64_t* kept_out, int64_t* n_kept, double* stats_out /* [5] */);
int besst_host_contam_stats(const int32_t* values, int64_t n, int32_t is_float, double offset,
                            int64_t* n_final, double* stats_out /* [4] */);
int besst_host_getdistr(const int32_t* values, const int64_t* kept, int64_t n_kept, int32_t is_float,
                        double offset, const int32_t* contig_lengths, int64_t n_contigs,
                        double* adjusted_out, int64_t adjusted_cap, int64_t* n_adjusted,
                        double* out /* [26] */);

/* ------------------------------------------------------------------------------------------------
 * device-pointer API (caller owns HBM; all pointers are device pointers unless noted)
 * ---------------------------------------------------------------------------------------------- */

/* Scratch bytes needed by besst_dev_classify / besst_dev_reduce for up to n records / tuples. */
size_t besst_dev_classify_workspace_bytes(int64_t n_records);
size_t besst_dev_reduce_workspace_bytes(int64_t n_tuples);

/* Pack the contig table into the 16-byte rows the kernels gather from, followed by one class byte per
 * contig (host pointers in, device pointer out; table must hold besst_dev_contig_table_bytes(n) bytes). */
size_t besst_dev_contig_table_bytes(int64_t n_contigs);
/* dst[0, bytes) <- src[0, bytes), both on the device and 16-byte aligned, in one small launch: how a pass's state block
 * (coverage numerators | counters | carry, zero / (-1, -1) at the head of every pass: BESST/CreateGraph.py:89-99) is
 * restored from a template. */
int besst_dev_restore_state(void* stream, void* dst, const void* src, int64_t bytes);
int besst_dev_pack_contigs(void* stream, int64_t n_contigs, const int32_t* h_scaf_id,
                           const int32_t* h_scaf_len, const int32_t* h_ctg_pos,
                           const int32_t* h_ctg_len, const uint8_t* h_direction,
                           const uint8_t* h_cls, void* d_table);

/* Stage 1: per-record classification, coverage accumulation, CreateEdge semantics (duplicate
 * chain, acceptance), ordered emission of link/fishy tuples.
 *   carry   int32[2] device: counter.prev_obs1/2 entering the batch, updated on exit
 *   aligned int64[n_contigs] device, accumulated into (zero it before the first batch)
 *   keys / payload  uint64[capacity >= n] device: emitted tuples in BAM order
 *   n_out   uint32 device: number of tuples emitted by this call
 *   counters  besst_counters device struct, accumulated into
 * Columns must be 16-byte aligned. */
int besst_dev_classify(void* stream, int64_t n, const int32_t* tid, const int32_t* mtid,
                       const int32_t* pos, const int32_t* mpos, const uint16_t* flag,
                       const uint8_t* mapq, const uint16_t* qlen, int64_t n_contigs,
                       const void* contig_table, const besst_lib_params* h_params, int32_t node_bits,
                       int32_t* carry, int64_t* aligned, uint64_t* keys, uint64_t* payload,
                       uint32_t* n_out, besst_counters* counters, void* workspace,
                       size_t workspace_bytes);

/* Share of the records whose mate lies on another contig (tid != mtid) - the only records the record loop does more
 * than add coverage for (CreateGraph.py:141-206) - counted over evenly spaced 1024-record tiles covering about
 * `sample_records` records; SYNCHRONISES the stream.  *record_path = the besst_lib_params.record_path to use
 * (1 from BESST_DENSE_CANDIDATE_SHARE on).  counts_scratch: 16 bytes of device memory. */
#define BESST_DENSE_CANDIDATE_SHARE 0.05
int besst_dev_candidate_density(void* stream, int64_t n, const int32_t* tid, const int32_t* mtid,
                                int64_t sample_records, void* counts_scratch, double* h_share,
                                int32_t* h_record_path);

/* Stage 2: sort of the tuples by (key, position in the stream) - observably a stable sort by key - and segmented
 * reduction into edge rows (up to 4 M tuples: one MSD partition + per-bucket sort and reduction; beyond, up to 2^30:
 * run-grouped - every 1024 consecutive tuples are grouped into runs of equal keys and the runs are sorted - or, with
 * BESST_REDUCE_NO_RUNS, chained-scan radix passes over the tuples).  *n_rows may come back as BESST_ROWS_*.
 *   n_tuples  uint32 device: number of valid tuples in keys/payload (<= capacity)
 *   key_base  a lower bound of every key (0 is always valid).  Scaffold ids keep growing across passes
 *             (param.scaffold_indexer, MakeScaffolds.py:276), so from the second library on all keys share a long
 *             common prefix; with key_base = ((2 * min scaffold id) << node_bits) << 1 the sort works on key - key_base
 *   key_bits  number of significant bits of key - key_base (2 * node_bits + 1 with key_base 0)
 * Outputs (capacity entries each): row_* arrays, obs_lo/obs_hi grouped by row, n_rows (uint32). */
int besst_dev_reduce(void* stream, int64_t capacity, const uint32_t* n_tuples, int32_t key_bits,
                     const uint64_t* keys, const uint64_t* payload, uint64_t* row_key,
                     uint32_t* row_mask, uint32_t* row_n, int64_t* row_sum, int64_t* row_sum_sq,
                     uint32_t* row_first, uint32_t* row_offset, int32_t* obs_lo, int32_t* obs_hi,
                     uint32_t* n_rows, void* workspace, size_t workspace_bytes, const uint32_t* first_map,
                     uint64_t key_base);
/* the same with BESST_REDUCE_* flags (besst_dev_reduce passes 0) */
int besst_dev_reduce_flags(void* stream, int64_t capacity, const uint32_t* n_tuples, int32_t key_bits,
                           const uint64_t* keys, const uint64_t* payload, uint64_t* row_key,
                           uint32_t* row_mask, uint32_t* row_n, int64_t* row_sum, int64_t* row_sum_sq,
                           uint32_t* row_first, uint32_t* row_offset, int32_t* obs_lo, int32_t* obs_hi,
                           uint32_t* n_rows, void* workspace, size_t workspace_bytes, const uint32_t* first_map,
                           uint64_t key_base, uint32_t flags);

/* Stage 1 + 2 on buffers that stay allocated (a resident builder): the large-stream form of stage 2 starts with a
 * histogram read of the whole key stream, which stage 1 can take on its way out (it has every key in registers when it
 * writes the ordered stream).  besst_dev_reduce_presort describes that hand-over for a given stage-2 call: it returns
 * 1 and fills *out when besst_dev_reduce with this capacity / key_bits / key_base / workspace would take its
 * histograms from `out->table` (a region of that workspace), 0 when it would not (small streams, keys that do not
 * pack).  besst_dev_classify_presort is besst_dev_classify that also fills the table (presort may be NULL);
 * besst_dev_reduce_presorted is besst_dev_reduce that trusts it (h_presort: the structure that classify call filled in;
 * keys may be NULL when it says `segmented`) - same workspace, same capacity, same key range, and the tuples of
 * exactly that classify call.  (Replaces nothing in the reference: it is the seam between
 * CreateGraph.py:141-206, where links are found, and :842-862, where they are summed per edge.) */
typedef struct besst_presort {
    uint32_t* table;      /* device: [rows][2][256] counters inside the stage-2 workspace */
    int32_t rows;         /* power of two */
    int32_t shift;        /* digits of key - key_base at shift and shift + 8 */
    uint64_t key_base;
    uint32_t capacity;    /* tuples at or beyond it are not counted (stage 2 ignores them too) */
    uint32_t flags;       /* in, read by besst_dev_reduce_presorted: BESST_REDUCE_* */
    /* The tuple stream itself can be handed over as the record loop leaves it - one segment per 16 384-record block,
     * ordered by the block offsets of the stitch - when stage 2's first stream pass can read it that way:
     * besst_dev_reduce_presort sets `segmented` to say so, besst_dev_classify_presort leaves it 1 (and fills the seg_*
     * fields) when it did NOT write the dense keys / payload, and besst_dev_reduce_presorted then reads the segments and -
     * in_record_loop 1 only - writes the dense payload (the `payload` argument of both calls) itself.  With in_record_loop 2
     * or 3 NEITHER dense column exists after the pass: stage 2 works on the segments and writes the row columns and the
     * observations only; `keys` / `payload` keep whatever an earlier pass left there (a caller that wants the dense tuple
     * stream classifies with BESST_REDUCE_NO_RUNS in `flags`, or without a presort).  in_record_loop (out): 1 = the record loop
     * counted the digits while it emitted; 2 = it handed its segments over without counting, because `flags` did not carry
     * BESST_REDUCE_NO_RUNS and stage 2 then groups runs and reads no histogram; 3 = as 2, and the record loop grouped the
     * runs of equal keys itself while it emitted: the key segments then hold run tables and a run byte per tuple instead
     * of keys (seg_run_* below), and stage 2 only sorts the list of runs and places the observations.  After 2 or 3 a
     * repeat of besst_dev_reduce_presorted WITH that flag (after BESST_ROWS_RUN_OVERFLOW) needs a repeat of the classify
     * call with it first. */
    int32_t segmented;
    int32_t in_record_loop;
    const uint64_t* seg_keys;
    const uint64_t* seg_payload;
    const uint32_t* seg_offsets;
    const uint32_t* seg_skip;
    uint32_t seg_blocks;
    uint32_t seg_tile;
    uint64_t* payload_out;
    const uint32_t* seg_chunk_first; /* per 512 positions of the ordered stream: the block the first of them lies in */
    /* in_record_loop == 3 only (else NULL / 0): per block the place of its first run in the stream-ordered run list, the
     * record loop's block summaries (planes of seg_summ_stride words) and the word that says a block's run tables overflowed */
    const uint32_t* seg_run_offsets;
    const uint32_t* seg_summ;
    uint32_t seg_summ_stride;
    const uint32_t* seg_run_status;
} besst_presort;
int besst_dev_reduce_presort(int64_t capacity, int32_t key_bits, uint64_t key_base, void* workspace,
                             size_t workspace_bytes, besst_presort* h_out);
int besst_dev_classify_presort(void* stream, int64_t n, const int32_t* tid, const int32_t* mtid,
                               const int32_t* pos, const int32_t* mpos, const uint16_t* flag,
                               const uint8_t* mapq, const uint16_t* qlen, int64_t n_contigs,
                               const void* contig_table, const besst_lib_params* h_params, int32_t node_bits,
                               int32_t* carry, int64_t* aligned, uint64_t* keys, uint64_t* payload,
                               uint32_t* n_out, besst_counters* counters, void* workspace,
                               size_t workspace_bytes, besst_presort* h_presort);
int besst_dev_reduce_presorted(void* stream, int64_t capacity, const uint32_t* n_tuples, int32_t key_bits,
                               const uint64_t* keys, const uint64_t* payload, uint64_t* row_key,
                               uint32_t* row_mask, uint32_t* row_n, int64_t* row_sum, int64_t* row_sum_sq,
                               uint32_t* row_first, uint32_t* row_offset, int32_t* obs_lo, int32_t* obs_hi,
                               uint32_t* n_rows, void* workspace, size_t workspace_bytes,
                               const uint32_t* first_map, uint64_t key_base, const besst_presort* h_presort);

/* ---- multi-GPU path ----------------------------------------------------
 * Ranks own contiguous slices of the (tid,pos)-sorted stream.  The duplicate chain of CreateEdge
 * (CreateGraph.py:835-838,869-870) crosses slice boundaries, so stage 1 is split in three phases:
 *   scan  - the per-record kernel on the local slice (independent of the incoming prev_obs)
 *   tail  - {has, obs1, obs2, 0} of the slice's last record that reached CreateEdge  -> all_gather
 *   emit  - resolve block/slice heads against the true incoming prev_obs and write the ordered tuples
 * besst_dev_classify_emit picks the incoming prev_obs of `rank` from the gathered tails (world x 4 int32; pass
 * NULL for a single slice) - the tail of the nearest earlier rank that has one, else what `carry` holds, i.e. the
 * prev_obs entering rank 0; besst_dev_resolve_carry does only that step.
 * Then tuples are stably partitioned by owner rank (besst_owner_of_scaffold of the key's min scaffold)
 * into `world` fixed-capacity regions for ONE equal-split all-to-all; besst_dev_unpack rebuilds an
 * ordered stream on the receiver with a global emit index per tuple (first_map of besst_dev_reduce)
 * and raises *overflow if any region was truncated (retry with a larger pair_capacity). */
int besst_dev_classify_scan(void* stream, int64_t n, const int32_t* tid, const int32_t* mtid,
                            const int32_t* pos, const int32_t* mpos, const uint16_t* flag,
                            const uint8_t* mapq, const uint16_t* qlen, int64_t n_contigs,
                            const void* contig_table, const besst_lib_params* h_params, int32_t node_bits,
                            int64_t* aligned, besst_counters* counters, void* workspace,
                            size_t workspace_bytes);
int besst_dev_classify_tail(void* stream, int64_t n, int32_t* tail, void* workspace, size_t workspace_bytes);
int besst_dev_resolve_carry(void* stream, const int32_t* tails, int32_t rank, int32_t* carry);
int besst_dev_classify_emit(void* stream, int64_t n, int32_t detect_duplicate, int32_t* carry, uint64_t* keys,
                            uint64_t* payload, uint32_t* n_out, besst_counters* counters, void* workspace,
                            size_t workspace_bytes, int64_t n_contigs, const void* contig_table,
                            int64_t* aligned, const int32_t* tails, int32_t rank, int32_t* slice_info);
/* out[i] = d + sigma^2 g'(d) / g(d) at d = d_lower + i, i < n, for two contigs of length contig_len: the left-hand side of
 * the GapEst ML condition over a run of gaps.  mathstats' PreCalcMLvaluesOfdLongContigs (MakeScaffolds.py:68) rounds
 * these values and inverts the map into its {observation -> gap} table; besst_amd/mathstats_compat.py does that on the
 * host from this array.  _dev: device pointer out, nothing synchronised; _ctx: host pointer out. */
int besst_dev_gap_condition_table(void* stream, double mean, double sigma, double read_len, double contig_len,
                                  int32_t d_lower, int32_t n, double* out);
int besst_ctx_gap_condition_table(besst_ctx* ctx, double mean, double sigma, double read_len, double contig_len,
                                  int32_t d_lower, int32_t n, double* h_out);

/* Per-edge scoring on caller-owned device buffers (the device-pointer form of besst_ctx_score_edges;
 * CreateGraph.py:498-614): ML gap by bisection, expected sigma, KS numerator h = max|#{l1 <= x} - #{l2 <= x}| on the
 * centred per-end observations.  In the sharded build every rank scores the rows it owns.
 *   row / swap / len1 / len2   one entry per scored edge (device): row index into the edge table, whether the
 *                              edge's first scaffold is the key's max node, the two scaffold lengths
 *   row_n .. obs_hi            the edge table of besst_dev_reduce
 *   workspace                  [ uint64 big_off[n_edges] | int32 scratch ]: big_off[e] = start (in ints) of edge e's
 *                              2 * next_pow2(n) scratch ints when next_pow2(n) > 8192, else ignored; the scratch
 *                              part follows at the next 256-byte boundary after the offsets */
int besst_dev_score_edges(void* stream, int64_t n_edges, const uint32_t* row, const uint8_t* swap,
                          const int32_t* len1, const int32_t* len2, const uint32_t* row_n,
                          const int64_t* row_sum, const uint32_t* row_offset, const int32_t* obs_lo,
                          const int32_t* obs_hi, double mean, double sigma, double read_len, double* gap,
                          double* sd0, int32_t* ks_h, uint8_t* flags, void* workspace, size_t workspace_bytes);

/* The mate-elsewhere bit column of a resident record stream (part of the record layout, made once when the records arrive:
 * by the ingest, by whoever uploads columns): bits[i / 8] bit i % 8 = (tid[i] != mtid[i]), besst_dev_mate_bits_bytes(n)
 * bytes.  Passed to the record loop in besst_lib_params.mate_bits. */
size_t besst_dev_mate_bits_bytes(int64_t n_records);
int besst_dev_mate_bits(void* stream, int64_t n_records, const int32_t* tid, const int32_t* mtid, void* bits);

/* The log-normal branch on caller-owned device buffers (device-pointer forms of besst_ctx_score_edges_lognormal and
 * besst_ctx_conditional_stddevs).
 *   besst_dev_lognormal_tables   F0[k] = sum_{x <= k} f(x), F1[k] = sum_{x <= k} x f(x) for k = 0 .. x_max (x_max + 1
 *                                doubles each), f the pmf of LogNormal(mu, sigma) on the integers; workspace:
 *                                besst_dev_lognormal_tables_workspace_bytes(x_max)
 *   besst_dev_score_edges_lognormal   workspace as for besst_dev_score_edges plus one more [8 bytes x n_edges, rounded up
 *                                to 256] at its end */
size_t besst_dev_lognormal_tables_workspace_bytes(int64_t x_max);
int besst_dev_lognormal_tables(void* stream, double mu, double sigma, int64_t x_max, double* F0, double* F1,
                               void* workspace, size_t workspace_bytes);
int besst_dev_score_edges_lognormal(void* stream, int64_t n_edges, const uint32_t* row, const uint8_t* swap,
                                    const int32_t* len1, const int32_t* len2, const uint32_t* row_n,
                                    const int64_t* row_sum, const uint32_t* row_offset, const int32_t* obs_lo,
                                    const int32_t* obs_hi, double mean, double sigma, double read_len, double ln_mu,
                                    double ln_sigma, int64_t x_max, const double* F0, const double* F1, int32_t max_gap,
                                    double* gap, int32_t* ks_h, uint8_t* flags, void* workspace, size_t workspace_bytes);
int besst_dev_conditional_stddevs(void* stream, const double* density, int64_t max_isize, const int32_t* steps,
                                  int32_t n_steps, double* out);

/* libmetrics sampling on caller-owned device columns (the device-pointer form of besst_ctx_metrics_sample;
 * libmetrics.py:63-84,293-303).  `state` is 6 x int64 on the device, in/out:
 *   [0] records so far that qualify for the insert-size sample   (is_proper_aligned_unique_innie/outie on a top contig)
 *   [1] records so far on a top contig                            (the contamination scan's sample_counter)
 *   [2] contamination observations so far
 *   [3] counter_total, [4] n_contam - both only over records inside the first-1,000,000 cut-off, [5] records scanned
 * A record's sample slot is its running count, so a slice of the stream whose state[0..2] was preset to the
 * counts of the slices before it writes exactly its share of the two 1,000,000-entry sample buffers (everything
 * else stays untouched): with zeroed buffers, an all-reduce(sum) over the slices' buffers is the ordered sample
 * of the whole stream.  count_only != 0 advances state[0..2] only (first phase of a sharded scan) and
 * leaves them exact; the sampling form leaves [0..2] exact up to the cut-offs and at least 1,000,000 beyond them (the
 * call works in parts of 64 Mi records and does not look at a part that begins with both samples full).
 * n is limited to 2^36 records per call (a C5 library is 2.67 x 10^9); workspace: besst_dev_metrics_workspace_bytes(n) - the tile counts and, for
 * min(n, 64 Mi) records, 8 bytes of sample staging per record. */
size_t besst_dev_metrics_workspace_bytes(int64_t n_records);
int besst_dev_metrics_sample(void* stream, int64_t n, const int32_t* tid, const int32_t* mtid,
                             const int32_t* tlen, const uint16_t* flag, const uint8_t* mapq,
                             int64_t n_contigs, const uint8_t* top_mask, int32_t orientation,
                             int32_t min_mapq, double read_len, int32_t count_only, int32_t* isize_out,
                             int32_t* contam_out, int64_t* state, void* workspace, size_t workspace_bytes);
size_t besst_dev_exchange_region_bytes(int64_t pair_capacity);
/* Speculative slice heads (the sharded build's default, no tail exchange): besst_dev_classify_emit with
 * `slice_info` (8 x int32, device) instead of `tails` leaves the slice's first record that reaches CreateEdge
 * unresolved and fills slice_info = { any reaching record, tail obs1, tail obs2, head present, head obs1, head obs2,
 * head flags (1 accept, 2 double call, 4 mapq 0, 8 tuple emitted), position of the head's tuple or -1 };
 * besst_dev_partition(..., slice_info) copies it into words 4..11 of every region header and records where the head's
 * tuple went (words 12, 13); besst_dev_unpack(..., speculative_heads = 1, rank, detect_duplicate, all_slice_info,
 * counters) replays the chain over the sources (CreateGraph.py:835-870), corrects the summed counters, drops a
 * duplicate head's tuple at its owner and shifts the global emit indexes; all_slice_info (world x 8, device, may be
 * null) receives every source's description.
 * A region may be followed by a RIDER: `rider_bytes` (multiple of 8) of 64-bit words that every source copies behind
 * the tuples of each of its regions (besst_dev_partition) and every receiver sums over its sources
 * (besst_dev_unpack, written to `rider_sum`).  The sharded build sends the coverage numerators and counters of
 * small assemblies this way instead of all-reducing them: one collective less per step.  The exchange
 * then moves besst_dev_exchange_stride_bytes(pair_capacity, rider_bytes) bytes per (source, destination) pair;
 * rider_bytes = 0: no rider, stride = region. */
size_t besst_dev_exchange_stride_bytes(int64_t pair_capacity, int64_t rider_bytes);
uint32_t besst_owner_of_scaffold(uint32_t scaffold_id, uint32_t world);
/* workspace: besst_dev_reduce_workspace_bytes(capacity) */
int besst_dev_partition(void* stream, int64_t capacity, const uint32_t* n_tuples, int32_t node_bits,
                        int32_t world, const uint64_t* keys, const uint64_t* payload, int64_t pair_capacity,
                        void* send_buffer, void* workspace, size_t workspace_bytes, const void* rider,
                        int64_t rider_bytes, const int32_t* slice_info);
int besst_dev_unpack(void* stream, int32_t world, int64_t pair_capacity, const void* recv_buffer, uint64_t* keys,
                     uint64_t* payload, uint32_t* gidx, uint32_t* n_out, uint32_t* overflow, void* rider_sum,
                     int64_t rider_bytes, int32_t speculative_heads, int32_t rank, int32_t detect_duplicate,
                     int32_t* all_slice_info, besst_counters* counters);

/* ---- Scaffold-graph linearisation on the scored edge table -----------------------------------
 * Steps 1-4 of MakeScaffolds.Algorithm (MakeScaffolds.py:75-82) in one call:
 *   step 1/3  RemoveIsolatedContigs            (MakeScaffolds.py:134-144)
 *   step 2    RemoveAmbiguousRegionsUsingScore (MakeScaffolds.py:206-241) with remove_edges (:156-204)
 *   step 4    RemoveLoops                      (MakeScaffolds.py:248-274)
 * Nodes are compact ids: scaffold k has the nodes 2k ('L') and 2k+1 ('R').
 *   steps            mask of the steps to run: 1 = step 1, 2 = step 2, 4 = step 3, 8 = step 4 (15 = all, in the order of
 *                    MakeScaffolds.Algorithm).  Without step 2 every edge counts as a link edge whatever its score;
 *                    step 4 then requires at most one link edge per node (BESST_ERR_STATE otherwise)
 *   a, b, score      the link edges of G that carry a score, in G.edges() order (a = edge[0], b = edge[1]); the
 *                    order decides ties between equal scores exactly as Python's stable sort does (:216-217)
 *   edge_alive       out, n_edges: 1 = the edge survives step 2
 *   scaffold_removed_by  out, n_scaffolds: 0 = the scaffold survives; 1 / 3 / 4 = the step that removed it
 *   node_ambivalent  out, 2 * n_scaffolds: 1 = the node printed 'SCORES AMBVIVALENT' (:183); node_top / node_second
 *                    hold the two scores it printed, node_best_edge the edge of the node's first visit (events are
 *                    replayed in visiting order by sorting on (score desc, node_best_edge asc, is-edge[1]))
 *   counters         out, HOST array of 8: [0] scaffolds removed by step 1, [1] by step 3, [2] cycles found by
 *                    step 4, [3] ambivalent nodes, [4] rounds step 2 took
 * besst_linearize takes host pointers (copies in, runs, copies out, like the besst_ctx_* calls);
 * besst_dev_linearize takes device pointers for everything but `counters` and synchronises the stream. */
int besst_linearize(int device, int32_t steps, int64_t n_scaffolds, int64_t n_edges, const int32_t* a, const int32_t* b,
                    const double* score, uint8_t* edge_alive, uint8_t* scaffold_removed_by,
                    uint8_t* node_ambivalent, double* node_top, double* node_second,
                    uint32_t* node_best_edge, int64_t* counters);
size_t besst_dev_linearize_workspace_bytes(int64_t n_scaffolds, int64_t n_edges);
int besst_dev_linearize(void* stream, int32_t steps, int64_t n_scaffolds, int64_t n_edges, const int32_t* a, const int32_t* b,
                        const double* score, void* workspace, size_t workspace_bytes, uint8_t* edge_alive,
                        uint8_t* scaffold_removed_by, uint8_t* node_ambivalent, double* node_top,
                        double* node_second, uint32_t* node_best_edge, int64_t* counters);

/* ---- chain extraction of the linearised scaffold graph (the rest of SURVEY 8(f) rank 3) ----------------------------
 * The data-parallel part of MakeScaffolds.NewContigsScaffolds / UpdateInfo (MakeScaffolds.py:270-341, 344-482): after
 * steps 1-4 every scaffold end has at most one link edge, i.e. the graph is a set of paths, and the reference walks each
 * path from one end.  Nodes: 2 * scaffold + (side == 'R'), scaffolds numbered in node order.
 *   link[2n]             the node at the other end of a node's link edge, -1 without one
 *   gap[2n]              the gap the walk adds when it crosses that edge (max(1, int(avg_gap)), MakeScaffolds.py:468-471)
 *   scaffold_length[n]   s_length of every scaffold;  node_order[2n]  position of every node in G.nodes()
 * Out, per node h (walking OUT of the scaffold through end h):
 *   terminal[2n]         the end of the path on that side (h itself when h has no link)
 *   beyond[2n]           lengths of the scaffolds beyond h on that side + the gaps in between (int64)
 *   lowest_order[2n]     smallest node_order among the nodes beyond h (INT32_MAX when there are none)
 * The host mirror (besst_amd/MakeScaffolds.py) turns these into the reference's Scaffolds / Contigs state.
 * besst_chain_scaffolds takes host pointers; besst_dev_chain_scaffolds device pointers (it synchronises the stream
 * between its doubling passes to find out when nothing moves any more); *passes = doubling passes run. */
size_t besst_dev_chain_workspace_bytes(int64_t n_scaffolds);
int besst_dev_chain_scaffolds(void* stream, int64_t n_scaffolds, const int32_t* link, const int32_t* gap,
                              const int32_t* scaffold_length, const int32_t* node_order, void* workspace,
                              size_t workspace_bytes, int32_t* terminal, int64_t* beyond, int32_t* lowest_order,
                              int32_t* h_passes);
int besst_chain_scaffolds(int device, int64_t n_scaffolds, const int32_t* link, const int32_t* gap,
                          const int32_t* scaffold_length, const int32_t* node_order, int32_t* terminal, int64_t* beyond,
                          int32_t* lowest_order, int32_t* passes);

/* ---- ScorePaths on the link graph -------------------------------------------------------------
 * Connectivity weights of a batch of candidate paths: calculate_connectivity / calculate_connectivity_contamination
 * of ScorePaths (ExtendLargeScaffolds.py:29-130); the path search itself stays on the host.
 *   n_nodes, row_ptr, col, weight   CSR of the LINK edges of the graph in both directions (node = 2 * scaffold +
 *                                   (side == 'R'); weight = nr_links); row_ptr has n_nodes + 1 entries
 *   n_paths, path_ptr, path_nodes   CSR of the paths (lists of nodes), path_ptr has n_paths + 1 entries
 *   contamination                   0: calculate_connectivity (:33-69), 1: the contamination variant (:72-105)
 *   good, bad                       out, n_paths: good_link_weight (BEFORE the division by two of :94) and
 *                                   bad_link_weight; score = good / float(bad) is formed by the caller (:63-67)
 * besst_score_paths takes host pointers; besst_dev_score_paths device pointers and only enqueues on `stream`. */
int besst_score_paths(int device, int64_t n_nodes, const int64_t* row_ptr, const int32_t* col,
                      const int32_t* weight, int64_t n_paths, const int64_t* path_ptr, const int32_t* path_nodes,
                      int32_t contamination, int64_t* good, int64_t* bad);
int besst_dev_score_paths(void* stream, int64_t n_nodes, const int64_t* row_ptr, const int32_t* col,
                          const int32_t* weight, int64_t n_paths, const int64_t* path_ptr,
                          const int32_t* path_nodes, int32_t contamination, int64_t* good, int64_t* bad);

#ifdef __cplusplus
}
#endif
#endif /* BESST_AMD_H */
