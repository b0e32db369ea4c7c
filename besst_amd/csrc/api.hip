// C ABI of libbesst_amd.so: argument checking, the HBM-owning context, and the host-side float
// finishing that must replay the reference's operation order.  See include/besst_amd.h.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <chrono>
#include <mutex>
#include <type_traits>
#include <vector>

#include "common.h"

namespace besst {

static thread_local char g_error[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

struct ProfRec {
    int slot;
    hipEvent_t a, b;
};
static uint32_t g_prof_mask = 0;
static uint32_t g_prof_every = 1;                  // record every n-th launch of an enabled slot
static uint32_t g_prof_seen[kProfSlots] = {};
static std::vector<ProfRec> g_prof;

ProfScope::ProfScope(hipStream_t stream, int slot) : s(stream), idx(-1) {
    if (!((g_prof_mask >> slot) & 1u)) return;
    if ((g_prof_seen[slot]++ % g_prof_every) != 0) return;
    ProfRec r;
    r.slot = slot;
    if (hipEventCreate(&r.a) != hipSuccess) return;
    if (hipEventCreate(&r.b) != hipSuccess) { (void)hipEventDestroy(r.a); return; }
    (void)hipEventRecord(r.a, s);
    g_prof.push_back(r);
    idx = (int)g_prof.size() - 1;
}
ProfScope::~ProfScope() {
    if (idx >= 0) (void)hipEventRecord(g_prof[(size_t)idx].b, s);
}

namespace {

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;   // elements
    int ensure(size_t n) {
        if (n <= cap) return BESST_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = n + n / 8 + 64;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), want * sizeof(T));
        if (e != hipSuccess) {
            set_error("hipMalloc(%zu bytes) failed: %s", want * sizeof(T), hipGetErrorString(e));
            return BESST_ERR_NOMEM;
        }
        cap = want;
        return BESST_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

int bits_for(uint64_t v) {
    int b = 1;
    while ((v >> b) != 0) ++b;
    return b;
}

}  // namespace

}  // namespace besst

using namespace besst;

// ---- pinned staging buffers are kept -------------------------------------------------------------------------------
// Pinning and unpinning host memory costs ~90 ms per GB on the bench host - 47 of the 220 ms an ingest of a 40 M-record
// file took, most of it the release at the end of the call.  The staging buffers of the two ingest forms therefore come
// from a process-wide pool and go back to it: a later call (the next library's file, the other form, the next context)
// finds them pinned.  The pool holds at most kPinnedKeep bytes (what comes back beyond that is freed), is never freed at
// exit (the runtime may be gone by then), and besst_release_cached_memory() empties it.
namespace {
// Work nobody waits for (freeing an ingest's device scratch): threads that are joined when the next one starts, when a
// context is destroyed and by besst_release_cached_memory() - never left running behind the library's last call.
struct Background {
    std::mutex mu;
    std::vector<std::thread> threads;
    ~Background() {                                          // (process exit with a context never destroyed: let them go)
        for (std::thread& t : threads)
            if (t.joinable()) t.detach();
    }
    void join_all() {
        std::vector<std::thread> mine;
        {
            std::lock_guard<std::mutex> g(mu);
            mine.swap(threads);
        }
        for (std::thread& t : mine)
            if (t.joinable()) t.join();
    }
    template <class F>
    void run(F f) {
        join_all();
        std::lock_guard<std::mutex> g(mu);
        threads.emplace_back(std::move(f));
    }
};
Background g_background;

// What every device ingest needs whatever the file: four streams (4 ms each to create: 17 of a call's 18 ms of set-up), twelve
// events and three small buffers.  One set per device stays with the process; a call takes it (a second call on the same
// device at the same time makes its own and destroys it), besst_release_cached_memory() does not touch it (a few KB).
struct IngestKit {
    std::mutex mu;
    bool busy = false;
    hipStream_t work[3] = {nullptr, nullptr, nullptr}, copy = nullptr;
    hipEvent_t ev[3][4] = {};
    char* heads = nullptr;
    size_t heads_bytes = 0;
    uint32_t* d_flags = nullptr;
    uint32_t* summ_host = nullptr;
};
IngestKit g_ingest_kit[16];
constexpr size_t kPinnedKeep = (size_t)1 << 30;
struct PinnedPool {
    struct Entry { void* p; size_t bytes; bool busy; };
    std::mutex mu;
    std::vector<Entry> all;
    void* acquire(size_t bytes) {
        {
            std::lock_guard<std::mutex> g(mu);
            Entry* best = nullptr;
            for (Entry& e : all)
                if (!e.busy && e.bytes >= bytes && e.bytes <= bytes + bytes / 2 + ((size_t)1 << 20) && (!best || e.bytes < best->bytes)) best = &e;
            if (best) { best->busy = true; return best->p; }
        }
        void* p = nullptr;
        if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
            trim(0);                                          // (what is cached may be what is missing)
            if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
        }
        std::lock_guard<std::mutex> g(mu);
        all.push_back(Entry{p, bytes, true});
        return p;
    }
    void give_back(void* p) {
        if (!p) return;
        {
            std::lock_guard<std::mutex> g(mu);
            for (Entry& e : all)
                if (e.p == p) e.busy = false;
        }
        trim(kPinnedKeep);
    }
    // free idle buffers, largest first, until at most `keep` idle bytes are left
    void trim(size_t keep) {
        for (;;) {
            void* victim = nullptr;
            {
                std::lock_guard<std::mutex> g(mu);
                size_t idle = 0;
                size_t at = all.size();
                for (size_t i = 0; i < all.size(); ++i)
                    if (!all[i].busy) {
                        idle += all[i].bytes;
                        if (at == all.size() || all[i].bytes > all[at].bytes) at = i;
                    }
                if (idle <= keep || at == all.size()) return;
                victim = all[at].p;
                all.erase(all.begin() + (long)at);
            }
            (void)hipHostFree(victim);
        }
    }
};
PinnedPool g_pinned;
}  // namespace

struct besst_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // contig table
    int64_t n_contigs = 0;
    int32_t node_bits = 1;
    uint64_t key_base = 0;
    int32_t key_bits = 3;
    DevBuf<ContigRow> table;
    DevBuf<int64_t> aligned;
    // library
    bool have_lib = false;
    besst_lib_params lib{};
    // resident records
    int64_t n_records = 0;
    DevBuf<int32_t> tid, mtid, pos, mpos, tlen;
    DevBuf<uint16_t> flag, qlen;
    DevBuf<uint8_t> mapq;
    DevBuf<uint8_t> mate_bits;       // one bit per record: tid != mtid (ClassifyArgs::mate_bits), valid for the first bits_upto records
    int64_t bits_upto = 0;
    // tuple stream + edge table
    DevBuf<uint64_t> keys, payload, row_key;
    DevBuf<uint32_t> row_mask, row_n, row_first, row_offset;
    DevBuf<int64_t> row_sum, row_sum_sq;
    DevBuf<int32_t> obs_lo, obs_hi;
    DevBuf<int32_t> obs_sum;         // besst_ctx_fetch_observation_sums: obs_lo + obs_hi, made and copied on side_stream
    hipStream_t side_stream = nullptr;
    DevBuf<char> ws;
    DevBuf<char> small;      // counters + carry + n_out + n_rows
    bool built = false;
    int64_t n_rows = 0, n_tuples = 0;
    // misc scratch for metrics / scoring
    DevBuf<uint8_t> top_mask;
    DevBuf<int32_t> sample_a, sample_b;
    DevBuf<char> aux;
    // prefix tables of the log-normal pmf (besst_ctx_score_edges_lognormal), kept while (mu, sigma, x_max) stay the same
    DevBuf<double> ln_tables;
    double ln_mu = 0.0, ln_sigma = 0.0;
    int64_t ln_x_max = 0;
};

namespace {

struct SmallBlock {
    besst_counters counters;
    int32_t carry[2];
    uint32_t n_out;
    uint32_t n_rows;
};

int use_device(besst_ctx* c) {
    BESST_HIP_TRY(hipSetDevice(c->device));
    return BESST_OK;
}

template <typename T>
int grow_copy(besst_ctx* c, DevBuf<T>& buf, int64_t have, const T* src, int64_t n) {
    if ((size_t)(have + n) > buf.cap) {
        DevBuf<T> bigger;
        int rc = bigger.ensure((size_t)(have + n) * 3 / 2 + 1024);
        if (rc) return rc;
        if (have) BESST_HIP_TRY(hipMemcpyAsync(bigger.p, buf.p, (size_t)have * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
        BESST_HIP_TRY(hipStreamSynchronize(c->stream));
        buf.release();
        buf = bigger;
    }
    BESST_HIP_TRY(hipMemcpyAsync(buf.p + have, src, (size_t)n * sizeof(T), hipMemcpyHostToDevice, c->stream));
    return BESST_OK;
}

}  // namespace

extern "C" {

int besst_abi_version(void) { return BESST_ABI_VERSION; }

const char* besst_last_error(void) { return g_error; }

void besst_release_cached_memory(void) {
    g_background.join_all();
    g_pinned.trim(0);
}

void besst_prof_enable(uint32_t slot_mask) {
    g_prof_mask = slot_mask;
    for (int i = 0; i < kProfSlots; ++i) g_prof_seen[i] = 0;
}

void besst_prof_sample_every(uint32_t n) { g_prof_every = n ? n : 1; }

int besst_prof_slots(void) { return kProfSlots; }

const char* besst_prof_slot_name(int slot) {
    static const char* names[kProfSlots] = {
        "stream_kernel", "fused_wave_kernel", "ordered_kernel", "stitch_spans_kernel+stitch_kernel",
        "presort_fixup_kernel", "compact_kernel",
        "radix_hist_kernel", "radix_rowscan_kernel", "radix_scatter_kernel", "bucket_sort_kernel", "bucket_reduce_kernel",
        "row_heads_kernel", "row_scan_kernel", "row_reduce_kernel",
        "os_hist_kernel", "os_offsets_kernel", "os_seg_tiles_kernel+os_scatter_kernel",
        "os_bucket_start_kernel+os_bucket_wave_kernel+os_bucket_wave_lds_kernel+os_bucket_sort_kernel", "os_bucket_rows_kernel",
        "os_reduce_kernel", "os_fixup_kernel",
        "metrics_kernels", "score_kernels", "rg_group_kernel", "rg_compact_kernel",
        "rg_tile_sums_kernel+rg_dst_kernel(+rg_rows_kernel)", "rg_copy_kernel", "msd_partition_kernel",
        "rl_list_kernel", "rl_place_kernel", "rl_rows_kernel"};
    return (slot >= 0 && slot < kProfSlots) ? names[slot] : "";
}

int besst_prof_collect(int n_slots, double* ms, int64_t* launches) {
    BESST_REQUIRE(ms && launches && n_slots >= kProfSlots, "prof_collect: need kProfSlots entries");
    for (int i = 0; i < n_slots; ++i) { ms[i] = 0.0; launches[i] = 0; }
    int rc = BESST_OK;
    for (ProfRec& r : g_prof) {
        float t = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) {
            ms[r.slot] += (double)t;
            launches[r.slot] += 1;
        } else {
            rc = BESST_ERR_HIP;
        }
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    g_prof.clear();
    if (rc) set_error("prof_collect: an event could not be read");
    return rc;
}

int besst_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
        return -BESST_ERR_HIP;
    }
    return n;
}

besst_ctx* besst_ctx_create(int device) {
    int n = besst_device_count();
    if (n <= 0) {
        if (n == 0) set_error("no HIP device visible");
        return nullptr;
    }
    if (device < 0 || device >= n) {
        set_error("device %d out of range (have %d)", device, n);
        return nullptr;
    }
    besst_ctx* c = new besst_ctx();
    c->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&c->stream) != hipSuccess) {
        set_error("could not create a stream on device %d", device);
        delete c;
        return nullptr;
    }
    if (c->small.ensure(sizeof(SmallBlock)) != BESST_OK) {
        delete c;
        return nullptr;
    }
    return c;
}

void besst_ctx_destroy(besst_ctx* c) {
    if (!c) return;
    g_background.join_all();
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    c->table.release(); c->aligned.release();
    c->tid.release(); c->mtid.release(); c->pos.release(); c->mpos.release(); c->tlen.release();
    c->flag.release(); c->qlen.release(); c->mapq.release(); c->mate_bits.release();
    c->keys.release(); c->payload.release(); c->row_key.release();
    c->row_mask.release(); c->row_n.release(); c->row_first.release(); c->row_offset.release();
    c->row_sum.release(); c->row_sum_sq.release(); c->obs_lo.release(); c->obs_hi.release();
    c->ws.release(); c->small.release(); c->top_mask.release(); c->sample_a.release();
    c->sample_b.release(); c->aux.release(); c->ln_tables.release();
    if (c->side_stream) {
        (void)hipStreamSynchronize(c->side_stream);
        (void)hipStreamDestroy(c->side_stream);
    }
    c->obs_sum.release();
    (void)hipStreamDestroy(c->stream);
    delete c;
}

int besst_dev_pack_contigs(void* stream, int64_t n, const int32_t* scaf_id, const int32_t* scaf_len,
                           const int32_t* ctg_pos, const int32_t* ctg_len, const uint8_t* direction,
                           const uint8_t* cls, void* d_table) {
    BESST_REQUIRE(n >= 0 && n < ((int64_t)1 << 31), "pack_contigs: n_contigs out of range");
    BESST_REQUIRE(n == 0 || (scaf_id && scaf_len && ctg_pos && ctg_len && direction && cls && d_table),
                  "pack_contigs: null pointer");
    std::vector<ContigRow> rows((size_t)n);
    // the class bytes follow the rows, and behind them one byte that says "every contig of the header is in the table"
    // (true for a first library: the record loop then never has to look a class up to know that a record counts)
    std::vector<uint8_t> cls_x((size_t)n + 1);
    uint8_t all_present = 1;
    for (int64_t i = 0; i < n; ++i) {
        BESST_REQUIRE(cls[i] <= BESST_CLS_SMALL, "pack_contigs: class must be 0, 1 or 2");
        cls_x[(size_t)i] = cls[i];
        if (cls[i] == BESST_CLS_ABSENT) all_present = 0;
        if (cls[i] != BESST_CLS_ABSENT)
            BESST_REQUIRE(scaf_id[i] >= 1 && (uint32_t)scaf_id[i] <= kScafIdMask,
                          "pack_contigs: scaffold id must be in [1, 2^28)");
        rows[(size_t)i].w0 = ((uint32_t)scaf_id[i] & kScafIdMask) | ((direction[i] ? 1u : 0u) << 28) |
                             ((uint32_t)cls[i] << 29);
        rows[(size_t)i].scaf_len = scaf_len[i];
        rows[(size_t)i].ctg_pos = ctg_pos[i];
        rows[(size_t)i].ctg_len = ctg_len[i];
    }
    if (n) {
        BESST_HIP_TRY(hipMemcpyAsync(d_table, rows.data(), (size_t)n * sizeof(ContigRow), hipMemcpyHostToDevice,
                                     static_cast<hipStream_t>(stream)));
        // class bytes follow the rows (the streaming kernel needs only the class of a contig)
        cls_x[(size_t)n] = all_present;
        BESST_HIP_TRY(hipMemcpyAsync(static_cast<char*>(d_table) + (size_t)n * sizeof(ContigRow), cls_x.data(), (size_t)n + 1,
                                     hipMemcpyHostToDevice, static_cast<hipStream_t>(stream)));
        BESST_HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));   // rows is a local
    }
    return BESST_OK;
}

int besst_ctx_set_contigs(besst_ctx* c, int64_t n, const int32_t* scaf_id, const int32_t* scaf_len,
                          const int32_t* ctg_pos, const int32_t* ctg_len, const uint8_t* direction,
                          const uint8_t* cls) {
    BESST_REQUIRE(c, "null context");
    int rc = use_device(c);
    if (rc) return rc;
    if ((rc = c->table.ensure((size_t)n + (size_t)n / 16 + 2))) return rc;
    if ((rc = c->aligned.ensure((size_t)n + 1))) return rc;
    if ((rc = besst_dev_pack_contigs(c->stream, n, scaf_id, scaf_len, ctg_pos, ctg_len, direction, cls, c->table.p)))
        return rc;
    uint32_t max_id = 1, min_id = 0xffffffffu;
    for (int64_t i = 0; i < n; ++i)
        if (cls[i] != BESST_CLS_ABSENT) {
            if ((uint32_t)scaf_id[i] > max_id) max_id = (uint32_t)scaf_id[i];
            if ((uint32_t)scaf_id[i] < min_id) min_id = (uint32_t)scaf_id[i];
        }
    if (min_id > max_id) min_id = max_id;
    c->node_bits = bits_for((uint64_t)max_id * 2 + 1);
    // every key is >= the one of (lowest node, lowest node): later libraries' scaffold ids start far above 1
    c->key_base = (((uint64_t)min_id * 2) << c->node_bits) << 1;
    c->key_bits = bits_for((((((uint64_t)max_id * 2 + 1) << c->node_bits) | ((uint64_t)max_id * 2 + 1)) << 1 | 1ull) - c->key_base);
    c->n_contigs = n;
    c->built = false;
    return BESST_OK;
}

int besst_ctx_set_library(besst_ctx* c, const besst_lib_params* p) {
    BESST_REQUIRE(c && p, "null pointer");
    BESST_REQUIRE(p->orientation == 0 || p->orientation == 1, "orientation must be 0 (fr) or 1 (rf)");
    BESST_REQUIRE(p->ins_size_threshold < 1073741824.0, "ins_size_threshold must be below 2^30");
    BESST_REQUIRE(p->read_len == p->read_len && p->ins_size_threshold == p->ins_size_threshold, "NaN parameter");
    c->lib = *p;
    c->have_lib = true;
    c->built = false;
    return BESST_OK;
}

int besst_ctx_clear_records(besst_ctx* c) {
    BESST_REQUIRE(c, "null context");
    c->n_records = 0;
    c->bits_upto = 0;
    c->built = false;
    return BESST_OK;
}

int besst_ctx_push_records(besst_ctx* c, int64_t n, const int32_t* tid, const int32_t* mtid, const int32_t* pos,
                           const int32_t* mpos, const int32_t* tlen, const uint16_t* flag, const uint8_t* mapq,
                           const uint16_t* qlen) {
    BESST_REQUIRE(c, "null context");
    BESST_REQUIRE(n >= 0, "negative record count");
    if (n == 0) return BESST_OK;
    BESST_REQUIRE(tid && mtid && pos && mpos && tlen && flag && mapq && qlen, "null column");
    BESST_REQUIRE(c->n_records + n < ((int64_t)1 << 32), "more than 2^32-1 records in one context");
    int rc = use_device(c);
    if (rc) return rc;
    const int64_t have = c->n_records;
    if ((rc = grow_copy(c, c->tid, have, tid, n))) return rc;
    if ((rc = grow_copy(c, c->mtid, have, mtid, n))) return rc;
    if ((rc = grow_copy(c, c->pos, have, pos, n))) return rc;
    if ((rc = grow_copy(c, c->mpos, have, mpos, n))) return rc;
    if ((rc = grow_copy(c, c->tlen, have, tlen, n))) return rc;
    if ((rc = grow_copy(c, c->flag, have, flag, n))) return rc;
    if ((rc = grow_copy(c, c->mapq, have, mapq, n))) return rc;
    if ((rc = grow_copy(c, c->qlen, have, qlen, n))) return rc;
    BESST_HIP_TRY(hipStreamSynchronize(c->stream));   // caller may reuse its host buffers
    c->n_records += n;
    c->built = false;
    return BESST_OK;
}

int besst_ctx_record_count(besst_ctx* c, int64_t* n) {
    BESST_REQUIRE(c && n, "null pointer");
    *n = c->n_records;
    return BESST_OK;
}

int besst_ctx_record_pointers(besst_ctx* c, int64_t* n, uint64_t* ptrs) {
    BESST_REQUIRE(c && n && ptrs, "null pointer");
    *n = c->n_records;
    const void* p[8] = {c->tid.p, c->mtid.p, c->pos.p, c->mpos.p, c->tlen.p, c->flag.p, c->mapq.p, c->qlen.p};
    for (int k = 0; k < 8; ++k) ptrs[k] = (uint64_t)(uintptr_t)p[k];
    return BESST_OK;
}

int besst_ctx_fetch_records(besst_ctx* c, int64_t first, int64_t n, int32_t* tid, int32_t* mtid, int32_t* pos, int32_t* mpos,
                            int32_t* tlen, uint16_t* flag, uint8_t* mapq, uint16_t* qlen) {
    BESST_REQUIRE(c, "null context");
    BESST_REQUIRE(first >= 0 && n >= 0 && first + n <= c->n_records, "fetch_records: range outside the resident records");
    if (n == 0) return BESST_OK;
    int rc = use_device(c);
    if (rc) return rc;
    auto down = [&](void* dst, const void* src, size_t bytes) -> hipError_t {
        return dst ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream) : hipSuccess;
    };
    const size_t m = (size_t)n;
    BESST_HIP_TRY(down(tid, c->tid.p + first, m * 4));
    BESST_HIP_TRY(down(mtid, c->mtid.p + first, m * 4));
    BESST_HIP_TRY(down(pos, c->pos.p + first, m * 4));
    BESST_HIP_TRY(down(mpos, c->mpos.p + first, m * 4));
    BESST_HIP_TRY(down(tlen, c->tlen.p + first, m * 4));
    BESST_HIP_TRY(down(flag, c->flag.p + first, m * 2));
    BESST_HIP_TRY(down(mapq, c->mapq.p + first, m));
    BESST_HIP_TRY(down(qlen, c->qlen.p + first, m * 2));
    BESST_HIP_TRY(hipStreamSynchronize(c->stream));
    return BESST_OK;
}

// Reserve room for `total` records in every column (one reallocation + device copy instead of a chain of them).
static int reserve_records(besst_ctx* c, int64_t total) {
    const int64_t have = c->n_records;
    auto grow = [&](auto& buf) -> int {
        using T = typename std::remove_reference<decltype(*buf.p)>::type;
        if ((size_t)total <= buf.cap) return BESST_OK;
        DevBuf<T> bigger;
        int rc = bigger.ensure((size_t)total);
        if (rc) return rc;
        if (have) BESST_HIP_TRY(hipMemcpyAsync(bigger.p, buf.p, (size_t)have * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
        BESST_HIP_TRY(hipStreamSynchronize(c->stream));
        buf.release();
        buf = bigger;
        return BESST_OK;
    };
    int rc;
    if ((rc = grow(c->tid)) || (rc = grow(c->mtid)) || (rc = grow(c->pos)) || (rc = grow(c->mpos)) || (rc = grow(c->tlen)) ||
        (rc = grow(c->flag)) || (rc = grow(c->mapq)) || (rc = grow(c->qlen)))
        return rc;
    return BESST_OK;
}

// BAM file -> resident records, streamed: chunks of the file are inflated and decoded by the reader's host threads into
// one of two sets of PINNED staging columns while the previous chunk's eight asynchronous copies are still on their way
// to HBM - decode and upload overlap, no pageable copy, no host-side concatenation of the whole stream.
int besst_ctx_push_bam(besst_ctx* c, besst_bam* bam, int64_t chunk_records, int64_t head_records, int32_t* head_rlen,
                       int32_t* head_alen, uint16_t* head_qlen, besst_ingest_stats* stats) {
    BESST_REQUIRE(c && bam, "push_bam: null context or reader");
    BESST_REQUIRE(head_records >= 0 && (head_records == 0 || (head_rlen && head_alen && head_qlen)), "push_bam: head buffers missing");
    if (chunk_records <= 0) chunk_records = (int64_t)4 << 20;
    if (chunk_records < 1024) chunk_records = 1024;
    int rc = use_device(c);
    if (rc) return rc;
    const auto t_start = std::chrono::steady_clock::now();
    struct Slot {
        int32_t *tid = nullptr, *mtid = nullptr, *pos = nullptr, *mpos = nullptr, *tlen = nullptr;
        uint16_t *flag = nullptr, *qlen = nullptr;
        uint8_t* mapq = nullptr;
        hipEvent_t done = nullptr;
        bool busy = false;
    } slot[2];
    std::vector<int32_t> rlen((size_t)chunk_records), alen((size_t)chunk_records);
    auto release = [&]() {
        for (Slot& sl : slot) {
            void* ptrs[8] = {sl.tid, sl.mtid, sl.pos, sl.mpos, sl.tlen, sl.flag, sl.qlen, sl.mapq};
            for (void* q : ptrs) g_pinned.give_back(q);
            if (sl.done) (void)hipEventDestroy(sl.done);
            sl = Slot();
        }
    };
    auto pinned = [&](void** out, size_t bytes) { return (*out = g_pinned.acquire(bytes)) != nullptr; };
    bool ok = true;
    for (Slot& sl : slot) {
        const size_t m = (size_t)chunk_records;
        ok = ok && pinned((void**)&sl.tid, m * 4) && pinned((void**)&sl.mtid, m * 4) && pinned((void**)&sl.pos, m * 4) &&
             pinned((void**)&sl.mpos, m * 4) && pinned((void**)&sl.tlen, m * 4) && pinned((void**)&sl.flag, m * 2) &&
             pinned((void**)&sl.qlen, m * 2) && pinned((void**)&sl.mapq, m) && hipEventCreate(&sl.done) == hipSuccess;
    }
    if (!ok) {
        release();
        set_error("push_bam: cannot allocate pinned staging buffers (2 x %lld records)", (long long)chunk_records);
        return BESST_ERR_NOMEM;
    }
    double decode_s = 0.0, wait_s = 0.0;
    int64_t pushed = 0, chunks = 0, bytes = 0;
    const int64_t file_bytes = bam_file_bytes(bam);
    rc = BESST_OK;
    for (int k = 0;; k ^= 1) {
        Slot& sl = slot[k];
        if (sl.busy) {                                       // the copies that last used this slot
            const auto t0 = std::chrono::steady_clock::now();
            if (hipEventSynchronize(sl.done) != hipSuccess) { set_error("push_bam: a host-to-device copy failed"); rc = BESST_ERR_HIP; break; }
            wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            sl.busy = false;
        }
        const auto t0 = std::chrono::steady_clock::now();
        const int64_t got = besst_bam_read_records(bam, chunk_records, sl.tid, sl.mtid, sl.pos, sl.mpos, sl.tlen, sl.flag, sl.mapq,
                                                   sl.qlen, rlen.data(), alen.data());
        decode_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (got < 0) { rc = (int)-got; break; }              // (the reader has set the error text)
        if (got == 0) break;
        for (int64_t i = 0; i < got && pushed + i < head_records; ++i) {
            head_rlen[pushed + i] = rlen[(size_t)i];
            head_alen[pushed + i] = alen[(size_t)i];
            head_qlen[pushed + i] = sl.qlen[i];
        }
        const int64_t have = c->n_records + pushed;
        if (have + got >= ((int64_t)1 << 32)) { set_error("more than 2^32-1 records in one context"); rc = BESST_ERR_ARG; break; }
        if ((size_t)(have + got) > c->tid.cap) {
            // room for the whole file at the rate of the bytes read so far (+ 6 %), at least for this chunk
            const int64_t at = bam_file_position(bam);
            int64_t want = have + got;
            if (file_bytes > 0 && at > 0 && at < file_bytes)
                want = c->n_records + (int64_t)((double)(pushed + got) * ((double)file_bytes / (double)at) * 1.06) + 4096;
            if (want < have + got) want = have + got;
            if (want >= ((int64_t)1 << 32)) want = ((int64_t)1 << 32) - 1;
            const int64_t keep = c->n_records;
            c->n_records = have;                             // what reserve_records has to carry over
            rc = reserve_records(c, want);
            c->n_records = keep;
            if (rc) break;
        }
        const size_t m = (size_t)got;
        hipError_t e = hipSuccess;
        auto up = [&](void* dst, const void* src, size_t nbytes) {
            if (e == hipSuccess) e = hipMemcpyAsync(dst, src, nbytes, hipMemcpyHostToDevice, c->stream);
            bytes += (int64_t)nbytes;
        };
        up(c->tid.p + have, sl.tid, m * 4); up(c->mtid.p + have, sl.mtid, m * 4); up(c->pos.p + have, sl.pos, m * 4);
        up(c->mpos.p + have, sl.mpos, m * 4); up(c->tlen.p + have, sl.tlen, m * 4); up(c->flag.p + have, sl.flag, m * 2);
        up(c->mapq.p + have, sl.mapq, m); up(c->qlen.p + have, sl.qlen, m * 2);
        if (e == hipSuccess) e = hipEventRecord(sl.done, c->stream);
        if (e != hipSuccess) { set_error("push_bam: %s", hipGetErrorString(e)); rc = BESST_ERR_HIP; break; }
        sl.busy = true;
        pushed += got;
        ++chunks;
    }
    const auto tw = std::chrono::steady_clock::now();
    const hipError_t es = hipStreamSynchronize(c->stream);
    wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - tw).count();
    release();
    if (rc == BESST_OK && es != hipSuccess) { set_error("push_bam: %s", hipGetErrorString(es)); rc = BESST_ERR_HIP; }
    if (rc) return rc;
    c->n_records += pushed;
    c->built = false;
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        stats->records = pushed;
        stats->chunks = chunks;
        stats->bytes_h2d = bytes;
        stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
        stats->decode_seconds = decode_s;
        stats->copy_wait_seconds = wait_s;
    }
    return BESST_OK;
}

// ---- BAM ingest on the GPU (bgzf_gpu.hip) ----------------------------------------------------------------------------
namespace {

// The BGZF blocks of [*fpos, ...) that fit one chunk: descriptors with offsets relative to the chunk's first byte,
// inflated places 256-byte aligned.  Stops at max_blocks, at comp_cap compressed bytes, or at the end of the file.
// false: not a BGZF block where one should be.
// dst0 / back_to_back: where the first block's bytes go and whether the blocks follow each other without padding (the
// ingest: a record may run on into the next block) or at 256-byte boundaries (the inflate test hook)
bool scan_bgzf_chunk(const uint8_t* map, size_t map_len, size_t* fpos, size_t max_blocks, size_t comp_cap, BgzfBlock* out,
                     uint32_t* n_out, size_t* comp_bytes, size_t* inflated_bytes, bool more_follows = false, size_t dst0 = 0,
                     bool back_to_back = false) {
    auto le16 = [](const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); };
    auto le32 = [](const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); };
    const size_t begin = *fpos;
    size_t at = begin, dst = dst0;
    uint32_t n = 0;
    while (n < max_blocks && at < map_len) {
        const uint8_t* hdr = map + at;
        if (map_len - at < 18) {                             // `map` is a window of the file: the block continues behind it
            if (more_follows) break;
            return false;
        }
        if (hdr[0] != 31 || hdr[1] != 139 || hdr[2] != 8 || !(hdr[3] & 4)) return false;
        const uint32_t xlen = le16(hdr + 10);
        if (xlen < 6 || hdr[12] != 'B' || hdr[13] != 'C' || le16(hdr + 14) != 2) return false;
        const size_t bsize = (size_t)le16(hdr + 16) + 1;
        if (bsize < 18) return false;
        if (map_len - at < bsize) {
            if (more_follows) break;
            return false;
        }
        const size_t rest = bsize - 18, extra_left = xlen - 6;
        if (rest < extra_left + 8) return false;
        if (at + bsize - begin > comp_cap) {
            if (n == 0) return false;                        // (a block is at most 64 KiB: the cap is far larger)
            break;
        }
        const uint32_t isize = le32(hdr + bsize - 4);
        if (isize > 65536u) return false;
        BgzfBlock& b = out[n++];
        b.src_off = (uint32_t)(at + 18 + extra_left - begin);
        b.src_len = (uint32_t)(rest - extra_left - 8);
        b.dst_off_lo = (uint32_t)dst;
        b.dst_off_hi = (uint32_t)((uint64_t)dst >> 32);
        b.dst_len = isize;
        b.crc = le32(hdr + bsize - 8);
        dst += back_to_back ? (size_t)isize : align_up((size_t)isize, 256);
        at += bsize;
    }
    *fpos = at;
    *n_out = n;
    *comp_bytes = at - begin;
    *inflated_bytes = dst - dst0;
    return true;
}

// First BGZF block boundary at or behind `from`: the gzip magic with the BC subfield, a plausible BSIZE, and two further
// blocks (or the end of the file) chained behind it - payload bytes that happen to spell a header do not survive that.
size_t find_bgzf_boundary(const uint8_t* map, size_t map_len, size_t from) {
    auto le16 = [](const uint8_t* p) { return (size_t)p[0] | ((size_t)p[1] << 8); };
    auto block_at = [&](size_t at, size_t* bsize) {
        if (map_len - at < 28) return false;
        const uint8_t* h = map + at;
        if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4) || le16(h + 10) < 6 || h[12] != 'B' || h[13] != 'C' || le16(h + 14) != 2)
            return false;
        *bsize = le16(h + 16) + 1;
        return *bsize >= 28 && *bsize <= map_len - at;
    };
    for (size_t at = from; at + 28 <= map_len; ++at) {
        size_t b0 = 0, b1 = 0, b2 = 0;
        if (!block_at(at, &b0)) continue;
        const size_t n1 = at + b0;
        if (n1 == map_len) return at;
        if (!block_at(n1, &b1)) continue;
        const size_t n2 = n1 + b1;
        if (n2 == map_len || block_at(n2, &b2)) return at;
    }
    return map_len;
}

}  // namespace

// BAM file -> resident records with the inflate and the record decode on the GPU: the file's COMPRESSED bytes are read into
// pinned memory by the reader's threads and uploaded chunk by chunk; per chunk one wave per BGZF block inflates, one lane per
// block walks its records, a scan places them and a thread per record fills the columns.  Three slots, each with its own
// staging, scratch and stream: chunk j + 1 is INFLATING and chunk j + 2 read, uploaded and queued behind it while the host
// waits for chunk j's record count (the columns may have to grow before its decode) - the tail of one chunk's waves and the
// head of the next share the chip.  Any block layout: a chunk's blocks are inflated back to back behind a slot that receives the record the chunk before
// left unfinished, and the record starts are guessed per block and verified from block to block (bgzf_gpu.hip).  A block
// the device cannot inflate returns BESST_ERR_UNSUPPORTED with context and reader untouched, and the caller takes
// besst_ctx_push_bam.
int besst_ctx_push_bam_device(besst_ctx* c, besst_bam* bam, int64_t chunk_blocks, int64_t head_records, int32_t* head_rlen,
                              int32_t* head_alen, uint16_t* head_qlen, besst_ingest_stats* stats) {
    return besst_ctx_push_bam_device_part(c, bam, 0, 1, chunk_blocks, head_records, head_rlen, head_alen, head_qlen, stats);
}

// The same for ONE PART of the file's records (multi-GPU ingest: rank r of W takes part r of W and holds the r-th slice of
// the stream, which is what phase 1 of the sharded build works on): the file is cut at the BGZF block boundaries nearest
// to part / parts of its bytes - in htslib's layout every block begins with a record, so every boundary is a valid place
// to start, and every rank finds the same boundaries on its own.
namespace {
// first_skip / boundary: the slice form (besst_ctx_push_bam_device_slice; boundary == nullptr: the part form, which takes
// htslib's layout only and begins every part with its first block's first byte).
int push_bam_device_impl(besst_ctx* c, besst_bam* bam, int32_t part, int32_t parts, int64_t chunk_blocks, int64_t head_records,
                         int32_t* head_rlen, int32_t* head_alen, uint16_t* head_qlen, besst_ingest_stats* stats,
                         int64_t first_skip, int64_t* boundary) {
    BESST_REQUIRE(c && bam, "push_bam_device: null context or reader");
    BESST_REQUIRE(parts >= 1 && part >= 0 && part < parts, "push_bam_device: part must be in [0, parts)");
    BESST_REQUIRE(head_records >= 0 && (head_records == 0 || (head_rlen && head_alen && head_qlen)),
                  "push_bam_device: head buffers missing");
    if (chunk_blocks <= 0) chunk_blocks = 5120;              // (one full chip of the inflate kernel's waves: 1024 SIMDs x 5; twice that reads 10 % slower)
    if (chunk_blocks < 64) chunk_blocks = 64;
    if (chunk_blocks > 65536) chunk_blocks = 65536;
    int rc = use_device(c);
    if (rc) return rc;
    const auto t_start = std::chrono::steady_clock::now();
    int64_t f0 = 0;
    uint32_t u0 = 0;
    if (!bam_record_position(bam, &f0, &u0)) {
        set_error("push_bam_device: the reader is inside a record that straddles two batches");
        return BESST_ERR_UNSUPPORTED;
    }
    size_t map_len = (size_t)bam_file_bytes(bam);        // (from here on: the end of this call's part of the file)
    const size_t whole_file = map_len;
    const bool slice = boundary != nullptr;
    if (parts > 1) {
        const uint8_t* map = bam_file_map(bam);
        const size_t file_len = map_len;
        auto cut = [&](int32_t k) -> size_t {
            if (k <= 0) return (size_t)f0;
            if (k >= parts) return file_len;
            const size_t at = find_bgzf_boundary(map, file_len, (size_t)((double)file_len * (double)k / (double)parts));
            return at < (size_t)f0 ? (size_t)f0 : at;
        };
        const size_t begin = cut(part);
        map_len = cut(part + 1);
        if (begin != (size_t)f0) u0 = 0;
        f0 = (int64_t)begin;
        if (map_len < begin) map_len = begin;
    }
    size_t nb = (size_t)chunk_blocks;
    // a chunk: nb blocks or comp_cap compressed bytes, whichever comes first.  The staging slots are pinned (~70 us per
    // MB to allocate and release), so they are sized from the file's first blocks - ~5 KB each in a file of constant
    // qualities, ~18 KB in a sequencer's - with a third in hand; denser blocks further on just make a chunk hold fewer.
    // A file (or part) of fewer blocks than a chunk gets slots for what it holds: every slot carries 64 KiB of inflated
    // scratch per block, 0.5 GB at the default chunk, whatever the file's size.
    size_t comp_cap = (size_t)160 << 20;
    double first_per_block = 0.0;                            // compressed bytes per block over the file's first blocks
    {
        const uint8_t* map = bam_file_map(bam);
        size_t at = (size_t)f0, seen = 0;
        while (seen < 256 && at + 18 <= map_len && map[at] == 31 && map[at + 1] == 139) {
            at += ((size_t)map[at + 16] | ((size_t)map[at + 17] << 8)) + 1;
            ++seen;
        }
        if (seen >= 1 && at <= map_len + 65536) {
            const double per_block = (double)(at - (size_t)f0) / (double)seen;
            if (seen >= 16) first_per_block = per_block;
            const size_t blocks = at >= map_len ? seen : (size_t)((double)(map_len - (size_t)f0) / per_block * 1.25) + 64;
            if (blocks < nb) nb = blocks < 64 ? 64 : blocks;
        }
        if (seen >= 16 && at <= map_len) {
            const size_t guess = align_up((size_t)((double)(at - (size_t)f0) / (double)seen * (double)nb * 1.35) + ((size_t)4 << 20), 4096);
            if (guess < comp_cap) comp_cap = guess;
        }
    }
    if (comp_cap > map_len - (size_t)f0 + 65536) comp_cap = align_up(map_len - (size_t)f0 + 65536, 4096);
    if (comp_cap < ((size_t)1 << 20)) comp_cap = (size_t)1 << 20;
    // descriptor 0 of every chunk is the slot for the tail of the chunk before (a record that its bytes did not finish)
    const size_t nbw = nb + 1;
    constexpr size_t kTailRoom = (size_t)4 << 20;            // bytes a chunk may carry into the next one (one record)
    const size_t desc_bytes = align_up(nbw * sizeof(BgzfBlock), 4096);
    const size_t slot_bytes = desc_bytes + comp_cap + 4096;      // (the bit reader's windows run up to 512 bytes past a payload)
    struct Chunk { uint32_t n_blocks = 0, first_off = 0; size_t comp = 0, inflated = 0, file_end = 0; };
    constexpr int kSlots = 3;        // chunk j's starts being verified, j + 1 inflating, j + 2 on its way to the device
    struct Slot {
        char* pin = nullptr;         // pinned: descriptors, then the chunk's bytes as they lie in the file
        char* dev = nullptr;         // the same on the device
        uint8_t* inflated = nullptr;
        uint32_t* symbols = nullptr; // the inflate kernel's symbol buffer: four bytes per byte of `inflated` (touched: per symbol)
        uint16_t* offs = nullptr;
        uint32_t* words = nullptr;   // status | count | exits | rec_base | guess | tail_at (nbw each), then 8 summary words
        hipStream_t work = nullptr;
        hipEvent_t h2d_done = nullptr, slot_free = nullptr, summ_done = nullptr, tail_taken = nullptr;
        Chunk ck;
    } sl[kSlots];
    char* heads = nullptr;           // head_rlen | head_alen | head_qlen on the device
    uint32_t* d_flags = nullptr;     // corrupt-record bit, saturated-qlen count
    uint32_t* summ_host = nullptr;   // pinned: kSlots x 12 summary words | [40] [41] flag words | [48..] kSlots tail descriptors
    hipStream_t copy_stream = nullptr;
    IngestKit* kit = nullptr;        // the device's cached streams / events / small buffers, if no other call holds them
    const size_t head_n = (size_t)(head_records > 0 ? head_records : 1);
    const size_t inflated_cap = kTailRoom + nb * 65536 + 4096;
    double unpin_s = 0.0;
    // What the call allocated goes back when it ends: the pinned staging to its pool at once; the device scratch, events and
    // streams - 12-15 ms of hipFree / destroy calls, a tenth of a 40 M-record ingest - on a thread of their own
    // (`background`: the successful end; every stream has been synchronised by then), nobody waits for it.
    auto release = [&](bool background = false) {
        std::vector<void*> dev_mem, host_mem;
        std::vector<hipEvent_t> events;
        std::vector<hipStream_t> streams;
        if (kit && !background) {
            // a call that did not end cleanly: a stream or event of it may be in an error state (a failed copy or kernel), and
            // a handle kept for the process would hand that state to every later ingest on this device.  Nothing is cached:
            // the kit's handles are destroyed with the call's own, the next call makes fresh ones.
            kit->heads = nullptr; kit->heads_bytes = 0;      // (== heads when it was reused: freed below)
            for (int k = 0; k < kSlots; ++k) {
                kit->work[k] = nullptr;
                for (int j = 0; j < 4; ++j) kit->ev[k][j] = nullptr;
            }
            kit->copy = nullptr; kit->d_flags = nullptr; kit->summ_host = nullptr;
            { std::lock_guard<std::mutex> g(kit->mu); kit->busy = false; }
            kit = nullptr;
        }
        for (Slot& q : sl) {
            const auto t0 = std::chrono::steady_clock::now();
            g_pinned.give_back(q.pin);
            unpin_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            for (void* m : {(void*)q.dev, (void*)q.inflated, (void*)q.symbols, (void*)q.offs, (void*)q.words})
                if (m) dev_mem.push_back(m);
            if (!kit) {
                for (hipEvent_t e : {q.h2d_done, q.slot_free, q.summ_done, q.tail_taken})
                    if (e) events.push_back(e);
                if (q.work) streams.push_back(q.work);
            }
        }
        if (kit) {                                           // (a call that synchronised cleanly: its handles serve the next one)
            for (int k = 0; k < kSlots; ++k) {
                kit->work[k] = sl[k].work;
                kit->ev[k][0] = sl[k].h2d_done; kit->ev[k][1] = sl[k].slot_free; kit->ev[k][2] = sl[k].summ_done; kit->ev[k][3] = sl[k].tail_taken;
            }
            kit->copy = copy_stream;
            kit->d_flags = d_flags;
            kit->summ_host = summ_host;
            if (heads && heads != kit->heads) { kit->heads = heads; kit->heads_bytes = head_n * 10; }
            std::lock_guard<std::mutex> g(kit->mu);
            kit->busy = false;
        } else {
            if (heads) dev_mem.push_back(heads);
            if (d_flags) dev_mem.push_back(d_flags);
            if (summ_host) host_mem.push_back(summ_host);
            if (copy_stream) streams.push_back(copy_stream);
        }
        for (Slot& q : sl) q = Slot();
        heads = nullptr; d_flags = nullptr; summ_host = nullptr; copy_stream = nullptr;
        const int device = c->device;
        auto drop = [device, dev_mem, host_mem, events, streams]() {
            (void)hipSetDevice(device);
            for (hipEvent_t e : events) (void)hipEventDestroy(e);
            for (hipStream_t st : streams) (void)hipStreamDestroy(st);
            for (void* m : dev_mem) (void)hipFree(m);
            for (void* m : host_mem) (void)hipHostFree(m);
        };
        if (background) g_background.run(drop);
        else drop();
    };
    // All three slots, and their streams, before anything is queued.
    // The three slots' streams and the copy stream must run beside each other.  The runtime spreads a process's streams
    // over a handful of hardware queues PER PRIORITY LEVEL, in creation order, together with every other stream of the
    // process (the context's, the caller's: torch's): two of ours on one queue and chunk j's walk / scan / decode wait
    // behind chunk j + 1's whole inflate - 1.83 instead of 1.40 s for full-size C3 in a process that had made other
    // streams before, 1.40 in one that had not.  So the slots' streams are created at the LOWEST priority, a level nobody
    // else in the process uses (its queues are theirs alone; nothing else runs during an ingest for them to yield to), and
    // the copy stream at the highest.
    int prio_low = 0, prio_high = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
    double alloc_s = 0.0;
    if (c->device >= 0 && c->device < 16) {
        IngestKit& k = g_ingest_kit[c->device];
        std::lock_guard<std::mutex> g(k.mu);
        if (!k.busy) { k.busy = true; kit = &k; }
    }

    auto alloc_slot = [&](int k) -> bool {
        Slot& q = sl[k];
        if (q.pin) return true;
        const auto t0 = std::chrono::steady_clock::now();
        const bool got = (q.pin = static_cast<char*>(g_pinned.acquire(slot_bytes))) != nullptr &&
             hipMalloc((void**)&q.dev, slot_bytes) == hipSuccess && hipMalloc((void**)&q.inflated, inflated_cap) == hipSuccess &&
             hipMalloc((void**)&q.symbols, 4 * bgzf_inflate_symbol_places(inflated_cap, nbw)) == hipSuccess &&
             hipMalloc((void**)&q.offs, nbw * (size_t)kBamBlockRecs * sizeof(uint16_t)) == hipSuccess &&
             hipMalloc((void**)&q.words, (nbw * 6 + 12) * sizeof(uint32_t)) == hipSuccess &&
             (q.work || hipStreamCreateWithPriority(&q.work, hipStreamNonBlocking, prio_low) == hipSuccess) &&
             (q.h2d_done || hipEventCreateWithFlags(&q.h2d_done, hipEventDisableTiming) == hipSuccess) &&
             (q.slot_free || hipEventCreateWithFlags(&q.slot_free, hipEventDisableTiming) == hipSuccess) &&
             (q.summ_done || hipEventCreateWithFlags(&q.summ_done, hipEventDisableTiming) == hipSuccess) &&
             (q.tail_taken || hipEventCreateWithFlags(&q.tail_taken, hipEventDisableTiming) == hipSuccess);
        alloc_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return got;
    };
    // Slot 0 now; slots 1 and 2 - 2 x (~190 MB pinned + ~0.7 GB of HBM): 90 of the 130 ms a first ingest spent allocating -
    // on a helper thread while the first chunk is read, uploaded and queued (joined before the second chunk is staged, and
    // before anything is released).  Their streams are created here, in order: the queue placement above depends on it.
    if (kit) {                                               // what an earlier call on this device left
        for (int k = 0; k < kSlots; ++k) {
            sl[k].work = kit->work[k];
            sl[k].h2d_done = kit->ev[k][0]; sl[k].slot_free = kit->ev[k][1]; sl[k].summ_done = kit->ev[k][2]; sl[k].tail_taken = kit->ev[k][3];
        }
        copy_stream = kit->copy;
        d_flags = kit->d_flags;
        summ_host = kit->summ_host;
        if (kit->heads_bytes >= head_n * 10) heads = kit->heads;
        else if (kit->heads) { (void)hipFree(kit->heads); kit->heads = nullptr; kit->heads_bytes = 0; }
    }
    bool ok = alloc_slot(0);
    // (every slot's stream AND its four events are made here, on the calling thread: the helper below only allocates memory,
    // so no handle the queueing code reads - sl[2].tail_taken while chunk 0 is enqueued - is ever written beside it)
    for (int k = 1; k < kSlots && ok; ++k)
        ok = (sl[k].work || hipStreamCreateWithPriority(&sl[k].work, hipStreamNonBlocking, prio_low) == hipSuccess) &&
             (sl[k].h2d_done || hipEventCreateWithFlags(&sl[k].h2d_done, hipEventDisableTiming) == hipSuccess) &&
             (sl[k].slot_free || hipEventCreateWithFlags(&sl[k].slot_free, hipEventDisableTiming) == hipSuccess) &&
             (sl[k].summ_done || hipEventCreateWithFlags(&sl[k].summ_done, hipEventDisableTiming) == hipSuccess) &&
             (sl[k].tail_taken || hipEventCreateWithFlags(&sl[k].tail_taken, hipEventDisableTiming) == hipSuccess);
    ok = ok && (heads || hipMalloc((void**)&heads, head_n * 10) == hipSuccess) &&
         (d_flags || hipMalloc((void**)&d_flags, 2 * sizeof(uint32_t)) == hipSuccess) &&
         (summ_host || hipHostMalloc((void**)&summ_host, 128 * sizeof(uint32_t), hipHostMallocDefault) == hipSuccess) &&
         (copy_stream || hipStreamCreateWithPriority(&copy_stream, hipStreamNonBlocking, prio_high) == hipSuccess);
    std::thread alloc_helper;
    std::atomic<bool> helper_ok(true);
    double alloc_wait_s = 0.0;                               // (what the calling thread spent waiting for the helper)
    auto alloc_join = [&]() -> bool {
        if (alloc_helper.joinable()) {
            const auto t0 = std::chrono::steady_clock::now();
            alloc_helper.join();
            alloc_wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }
        return helper_ok.load();
    };
    if (ok && map_len - (size_t)f0 > comp_cap / 2) {             // (a file of less than a chunk or so never uses them)
        const int device = c->device;
        alloc_helper = std::thread([&, device] {
            if (hipSetDevice(device) != hipSuccess) { helper_ok = false; return; }
            for (int k = 1; k < kSlots; ++k)
                if (!alloc_slot(k)) { helper_ok = false; return; }
        });
    }
    if (!ok) {
        alloc_join();
        release();
        set_error("push_bam_device: cannot allocate the staging / scratch buffers (%zu MB pinned, %zu MB of HBM)",
                  (kSlots * slot_bytes) >> 20, (kSlots * (slot_bytes + inflated_cap + 4 * bgzf_inflate_symbol_places(inflated_cap, nbw))) >> 20);
        return BESST_ERR_NOMEM;
    }
    double stage_s = 0.0, wait_s = 0.0;
    int64_t pushed = 0, chunks = 0, comp_total = 0, inflated_total = 0, blocks_total = 0, repaired = 0;
    size_t fpos = (size_t)f0;
    double bytes_per_block = first_per_block;
    rc = BESST_OK;
    auto hip_fail = [&](hipError_t e) { set_error("push_bam_device: %s", hipGetErrorString(e)); rc = BESST_ERR_HIP; };
    size_t max_blocks = nb;                                  // (blocks per chunk: fewer for the blocks behind a part's end)
    bool overhang = false;                                   // the chunk at hand holds the blocks behind the part's end
    // read chunk j (the next blocks of the file) into slot j % kSlots and start its upload
    auto stage = [&](int64_t j) -> bool {
        Slot& q = sl[j % kSlots];
        if (fpos >= map_len) { q.ck = Chunk(); return true; }   // (nothing left: the slot is not touched)
        if ((j > 0 && !alloc_join()) || !alloc_slot((int)(j % kSlots))) {
            set_error("push_bam_device: cannot allocate the staging / scratch buffers (%zu MB pinned, %zu MB of HBM)",
                      (kSlots * slot_bytes) >> 20, (kSlots * (slot_bytes + inflated_cap + 4 * bgzf_inflate_symbol_places(inflated_cap, nbw))) >> 20);
            rc = BESST_ERR_NOMEM;
            return false;
        }
        if (j >= kSlots) {                                        // the upload that last read this pinned slot
            const auto t0 = std::chrono::steady_clock::now();
            const hipError_t e = hipEventSynchronize(q.h2d_done);
            wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (e != hipSuccess) { hip_fail(e); return false; }
        }
        const auto t0 = std::chrono::steady_clock::now();
        const size_t begin = fpos;
        q.ck = Chunk();
        q.ck.first_off = j == 0 ? u0 : 0u;
        if (begin >= map_len) return true;
        // A window of the file is READ into the pinned slot by the reader's threads (pread: no page faults, unlike a copy
        // off the mapping, where the header walk alone touched every page) and the block headers are walked there; the
        // window is sized from the blocks seen so far, the block it cuts is read again with the next chunk.
        size_t want = map_len - begin < comp_cap ? map_len - begin : comp_cap;
        // the first two chunks are short ones - a fifth and a half of a chunk -, so that the chip has something to inflate a
        // millisecond or two into the call instead of after a whole chunk's read and upload (40 M records, eight calls
        // each way on one box: 0.119 against 0.125 s); the window of the very first read is sized from the file's first
        // blocks (bytes_per_block starts at their average)
        size_t cap_blocks = max_blocks;
        if (!overhang && j < 2) {
            cap_blocks = j == 0 ? nb / 5 : nb / 2;
            if (cap_blocks < 64) cap_blocks = nb < 64 ? nb : 64;
        }
        if (bytes_per_block > 0.0) {
            const size_t guess = (size_t)((double)cap_blocks * bytes_per_block * 1.08) + 65536;
            if (guess < want) want = guess;
        }
        if (!bam_parallel_read(bam, q.pin + desc_bytes, (int64_t)begin, want)) {
            set_error("push_bam_device: reading the file failed at offset %zu", begin);
            rc = BESST_ERR_ARG;
            return false;
        }
        size_t used = 0;
        BgzfBlock* desc = reinterpret_cast<BgzfBlock*>(q.pin);
        desc[0] = BgzfBlock{0u, 0u, (uint32_t)kTailRoom, 0u, 0u, 0u};   // the tail slot: empty until the chunk before says otherwise
        if (!scan_bgzf_chunk(reinterpret_cast<const uint8_t*>(q.pin + desc_bytes), want, &used, cap_blocks, comp_cap, desc + 1,
                             &q.ck.n_blocks, &q.ck.comp, &q.ck.inflated, begin + want < map_len, kTailRoom, true) ||
            (q.ck.n_blocks == 0 && want > 0)) {
            set_error("push_bam_device: not a BGZF block at file offset %zu", begin + used);
            rc = BESST_ERR_UNSUPPORTED;
            return false;
        }
        fpos = begin + q.ck.comp;
        q.ck.file_end = fpos;
        bytes_per_block = (double)q.ck.comp / (double)q.ck.n_blocks;
        stage_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        hipError_t e = hipSuccess;
        if (j >= kSlots) e = hipStreamWaitEvent(copy_stream, q.slot_free, 0);    // the kernels that last read this device slot
        if (e == hipSuccess) e = hipMemcpyAsync(q.dev, q.pin, ((size_t)q.ck.n_blocks + 1) * sizeof(BgzfBlock), hipMemcpyHostToDevice, copy_stream);
        if (e == hipSuccess) e = hipMemcpyAsync(q.dev + desc_bytes, q.pin + desc_bytes, q.ck.comp + 1024, hipMemcpyHostToDevice, copy_stream);
        if (e == hipSuccess) e = hipEventRecord(q.h2d_done, copy_stream);
        if (e != hipSuccess) { hip_fail(e); return false; }
        comp_total += (int64_t)q.ck.comp;
        inflated_total += (int64_t)q.ck.inflated;
        blocks_total += q.ck.n_blocks;
        return true;
    };
    // inflate + CRC of the chunk in slot k on the slot's stream (descriptor 0, the tail slot, is empty here: skipped)
    auto enqueue_inflate = [&](int k) -> bool {
        Slot& q = sl[k];
        hipError_t e = hipStreamWaitEvent(q.work, q.h2d_done, 0);
        // (the chunk two before inflated into this buffer; its tail may still be on its way to the chunk in between)
        if (e == hipSuccess) e = hipStreamWaitEvent(q.work, q.tail_taken, 0);
        if (e != hipSuccess) { hip_fail(e); return false; }
        if (launch_bgzf_inflate(q.work, reinterpret_cast<const uint8_t*>(q.dev + desc_bytes), reinterpret_cast<const BgzfBlock*>(q.dev),
                                q.ck.n_blocks + 1, q.inflated, q.words, q.symbols)) {
            rc = BESST_ERR_HIP;
            return false;
        }
        return true;
    };
    // where the records of the chunk in slot k begin (entry guesses, walks, verification, scan), its summary on the way to the
    // host.  tail_len bytes at `tail_at` of the buffer of the slot BEFORE are the record the chunk before did not finish: they
    // are copied in front of this chunk's first block and become its block 0.  first_entry: where the first record begins
    // in block 1 when there is no tail (the end of the header in the file's first chunk, else 0).
    const int32_t n_ref = besst_bam_n_references(bam);
    auto enqueue_walk = [&](int k, uint64_t tail_at, uint32_t tail_len, uint32_t forced_block, uint32_t forced_entry, uint32_t mode) -> bool {
        Slot& q = sl[k];
        Slot& other = sl[(k + kSlots - 1) % kSlots];
        uint32_t* w = q.words;
        hipError_t e = hipSuccess;
        if (tail_len) {
            BgzfBlock* patch = reinterpret_cast<BgzfBlock*>(summ_host + 48) + k;
            *patch = BgzfBlock{0u, 0u, (uint32_t)(kTailRoom - tail_len), 0u, tail_len, 0u};
            e = hipMemcpyAsync(q.dev, patch, sizeof(BgzfBlock), hipMemcpyHostToDevice, q.work);
            if (e == hipSuccess)
                e = hipMemcpyAsync(q.inflated + kTailRoom - tail_len, other.inflated + tail_at, tail_len, hipMemcpyDeviceToDevice, q.work);
        }
        if (e == hipSuccess && other.tail_taken) e = hipEventRecord(other.tail_taken, q.work);   // (a slot no chunk has used yet has no buffer to protect)
        if (e != hipSuccess) { hip_fail(e); return false; }
        if (launch_bam_walk_scan(q.work, q.inflated, reinterpret_cast<const BgzfBlock*>(q.dev), q.ck.n_blocks + 1,
                                 (uint64_t)kTailRoom + q.ck.inflated, n_ref, tail_len ? 0xffffffffu : forced_block, forced_entry, mode, w, q.offs,
                                 w + nbw, w + 2 * nbw, w + 3 * nbw, w + 4 * nbw, w + 5 * nbw, w + 6 * nbw)) {
            rc = BESST_ERR_HIP;
            return false;
        }
        e = hipMemcpyAsync(summ_host + 12 * k, w + 6 * nbw, 12 * sizeof(uint32_t), hipMemcpyDeviceToHost, q.work);
        if (e == hipSuccess) e = hipEventRecord(q.summ_done, q.work);
        if (e != hipSuccess) { hip_fail(e); return false; }
        return true;
    };
    BamColumns col{};
    col.head_rlen = reinterpret_cast<int32_t*>(heads);
    col.head_alen = reinterpret_cast<int32_t*>(heads + head_n * 4);
    col.head_qlen = reinterpret_cast<uint16_t*>(heads + head_n * 8);
    {
        hipError_t e = hipMemsetAsync(d_flags, 0, 2 * sizeof(uint32_t), c->stream);
        if (e == hipSuccess) e = hipMemsetAsync(heads, 0, head_n * 10, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);   // the slots' streams start behind these
        if (e != hipSuccess) hip_fail(e);
    }
    // where the first chunk's first record begins: the end of the header / the first byte of a part of a file in htslib's
    // layout; a slice behind the first one: where the caller says (the bytes in front belong to the last record of the slice
    // before), or a guess that the caller will check against what the slice before reports
    const double setup_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    uint32_t fb0 = 1u, fe0 = u0, mode0 = 0u;
    int64_t no_start_left = -1;                              // >= 0: a slice no record begins in; so many bytes of the record before lie behind it
    if (rc == BESST_OK && stage(0) && sl[0].ck.n_blocks) {
        if (slice && part > 0 && first_skip < 0) {
            fb0 = 0xffffffffu; fe0 = 0u; mode0 = kWalkFirstGuessed;
        } else if (slice && part > 0) {
            const BgzfBlock* desc = reinterpret_cast<const BgzfBlock*>(sl[0].pin);
            uint64_t skip = (uint64_t)first_skip;
            fb0 = 0u;
            for (uint32_t i = 1; i <= sl[0].ck.n_blocks; ++i) {
                if (skip < desc[i].dst_len) { fb0 = i; fe0 = (uint32_t)skip; break; }
                skip -= desc[i].dst_len;
            }
            if (fb0 == 0u && fpos >= map_len) {
                // the WHOLE slice lies in this chunk and no record begins in it (its blocks are the tail of the record before -
                // or hold nothing: the EOF marker of a file with fewer blocks than ranks): an empty slice, what comes in goes on
                no_start_left = (int64_t)skip;
                sl[0].ck = Chunk();
            } else if (fb0 == 0u) {
                set_error("push_bam_device: the slice's first record begins behind its first chunk (%lld bytes in)", (long long)first_skip);
                rc = BESST_ERR_UNSUPPORTED;
            }
        }
        if (rc == BESST_OK && sl[0].ck.n_blocks && enqueue_inflate(0) && enqueue_walk(0, 0, 0u, fb0, fe0, mode0) && stage(1) && sl[1].ck.n_blocks)
            enqueue_inflate(1);
    }
    int64_t first_at = -1, carry_out = 0, over_bytes = 0;    // (the slice form's answers)
    for (int64_t j = 0; rc == BESST_OK && sl[j % kSlots].ck.n_blocks; ++j) {
        Slot& q = sl[j % kSlots];
        Slot& nx = sl[(j + 1) % kSlots];
        // chunk j + 2: read, uploaded and queued behind chunk j + 1's inflate while chunk j's count is on its way (with two
        // slots the inflate of chunk j + 2 could not be queued before chunk j's verdict had been seen AND the file read:
        // the chip idled between two inflates whenever the two took longer than one inflate)
        if (nx.ck.n_blocks) {
            if (!stage(j + 2)) break;
            if (sl[(j + 2) % kSlots].ck.n_blocks && !enqueue_inflate((int)((j + 2) % kSlots))) break;
        }
        {
            const auto t0 = std::chrono::steady_clock::now();
            const hipError_t e = hipEventSynchronize(q.summ_done);
            wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (e != hipSuccess) { hip_fail(e); break; }
        }
        const uint32_t* sm = summ_host + 12 * (j % kSlots);
        if (!sm[1]) {
            if (sm[3]) set_error("push_bam_device: block %u of chunk %lld did not inflate on the device (status %u)", sm[2] ? sm[2] - 1u : 0u, (long long)j, sm[3]);
            else set_error("push_bam_device: the records of chunk %lld could not be located on the device (block %u: a record start "
                           "that its neighbours do not confirm, or a corrupt length)", (long long)j, sm[2] ? sm[2] - 1u : 0u);
            rc = BESST_ERR_UNSUPPORTED;
            break;
        }
        if (parts > 1 && sm[7] && !slice) {
            set_error("push_bam_device: a record straddles BGZF blocks (chunk %lld): a part of such a file cannot be cut at a block", (long long)j);
            rc = BESST_ERR_UNSUPPORTED;
            break;
        }
        const uint32_t tail_len = sm[4];
        if (tail_len > (uint32_t)kTailRoom) {
            set_error("push_bam_device: a record of more than %zu MB", kTailRoom >> 20);
            rc = BESST_ERR_UNSUPPORTED;
            break;
        }
        if (j == 0) {
            const uint64_t at = (uint64_t)sm[8] | ((uint64_t)sm[9] << 32);
            first_at = at == ~0ull ? -1 : (int64_t)(at - (uint64_t)kTailRoom);
            if (slice && first_at < 0) {
                set_error("push_bam_device: no record begins in the first chunk of the slice");
                rc = BESST_ERR_UNSUPPORTED;
                break;
            }
        }
        if (overhang) {                                      // bytes of the slice's last record that lie in the next slice
            if (tail_len) over_bytes += (int64_t)q.ck.inflated;              // (all of this chunk, and the record goes on)
            else carry_out = over_bytes + (int64_t)sm[8];
        }
        if (nx.ck.n_blocks) {
            // chunk j + 1's records can be located now: it starts with chunk j's unfinished record, if there is one
            if (!enqueue_walk((int)((j + 1) % kSlots), (uint64_t)sm[5] | ((uint64_t)sm[6] << 32), tail_len, 1u, 0u, 0u)) break;
        } else if (tail_len && slice && (overhang ? fpos < whole_file : map_len < whole_file)) {
            // the slice's last record runs on behind the slice's end: the blocks that follow are inflated for its bytes (and for
            // nothing else: the records that begin in them are the next slice's) - chunk after chunk until the record ends
            // (a few blocks at first - a record seldom runs over more than one or two -, four times as many while it goes on: the
            // blocks belong to the next slice, and a damaged one among them is that slice's to report)
            max_blocks = overhang ? (max_blocks * 4 < nb ? max_blocks * 4 : nb) : (nb < 64 ? nb : 64);
            overhang = true;
            map_len = whole_file;
            sl[(j + 2) % kSlots].ck = Chunk();
            if (!stage(j + 1)) break;
            if (!nx.ck.n_blocks) { set_error("push_bam_device: the file ends inside a record"); rc = BESST_ERR_ARG; break; }
            if (!enqueue_inflate((int)((j + 1) % kSlots))) break;
            if (!enqueue_walk((int)((j + 1) % kSlots), (uint64_t)sm[5] | ((uint64_t)sm[6] << 32), tail_len, 0xffffffffu, 0u, kWalkOverhang)) break;
        } else if (tail_len) {
            set_error("push_bam_device: the %s ends inside a record", parts > 1 ? "part of the file" : "file");
            rc = parts > 1 ? BESST_ERR_UNSUPPORTED : BESST_ERR_ARG;
            break;
        }
        const int64_t got = (int64_t)sm[0];
        repaired += (int64_t)sm[2];
        const int64_t have = c->n_records + pushed;
        if (have + got >= ((int64_t)1 << 32)) { set_error("more than 2^32-1 records in one context"); rc = BESST_ERR_ARG; break; }
        if ((size_t)(have + got) > c->tid.cap) {
            // room for the whole file at the rate of the bytes read so far (+ 6 %), at least for this chunk; the decode of
            // the chunk before may still be writing the columns that are about to move
            int64_t want = have + got;
            const size_t at = q.ck.file_end;                 // end of chunk j in the file
            if (at > (size_t)f0 && at < map_len)
                want = c->n_records + (int64_t)((double)(pushed + got) * ((double)(map_len - (size_t)f0) / (double)(at - (size_t)f0)) * 1.06) + 4096;
            if (want < have + got) want = have + got;
            if (want >= ((int64_t)1 << 32)) want = ((int64_t)1 << 32) - 1;
            const auto t0 = std::chrono::steady_clock::now();
            hipError_t e = hipSuccess;
            for (Slot& o : sl)                               // (slot_free: behind a slot's last decode)
                if (e == hipSuccess && o.slot_free) e = hipEventSynchronize(o.slot_free);
            wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (e != hipSuccess) { hip_fail(e); break; }
            const int64_t keep = c->n_records;
            c->n_records = have;
            rc = reserve_records(c, want);
            c->n_records = keep;
            if (rc) break;
        }
        col.tid = c->tid.p; col.mtid = c->mtid.p; col.pos = c->pos.p; col.mpos = c->mpos.p; col.tlen = c->tlen.p;
        col.flag = c->flag.p; col.qlen = c->qlen.p; col.mapq = c->mapq.p;
        if (launch_bam_decode(q.work, q.inflated, reinterpret_cast<const BgzfBlock*>(q.dev), q.ck.n_blocks + 1, q.offs, q.words + nbw,
                              q.words + 3 * nbw, col, have, pushed, head_records, d_flags)) { rc = BESST_ERR_HIP; break; }
        const hipError_t e = hipEventRecord(q.slot_free, q.work);
        if (e != hipSuccess) { hip_fail(e); break; }
        pushed += got;
        ++chunks;
    }
    const auto tw = std::chrono::steady_clock::now();
    hipError_t e0 = hipSuccess, e1 = hipSuccess;
    for (Slot& q : sl) {
        if (!q.work) continue;
        const hipError_t e = hipStreamSynchronize(q.work);
        if (e != hipSuccess) e0 = e;
    }
    const hipError_t ec = hipStreamSynchronize(copy_stream);
    if (rc == BESST_OK && (e0 != hipSuccess || e1 != hipSuccess || ec != hipSuccess)) hip_fail(e0 != hipSuccess ? e0 : e1 != hipSuccess ? e1 : ec);
    if (rc == BESST_OK) {
        hipError_t e = hipMemcpyAsync(summ_host + 40, d_flags, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
        const int64_t hn = pushed < head_records ? pushed : head_records;
        if (e == hipSuccess && hn > 0) {
            e = hipMemcpyAsync(head_rlen, col.head_rlen, (size_t)hn * 4, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(head_alen, col.head_alen, (size_t)hn * 4, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(head_qlen, col.head_qlen, (size_t)hn * 2, hipMemcpyDeviceToHost, c->stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) hip_fail(e);
    }
    wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - tw).count();
    if (rc == BESST_OK && (summ_host[40] & 1u)) {
        set_error("push_bam_device: corrupt record (its name and CIGAR do not fit its length)");
        rc = BESST_ERR_ARG;
    }
    const uint32_t saturated = rc == BESST_OK ? summ_host[41] : 0u;
    const auto t_rel = std::chrono::steady_clock::now();
    alloc_join();
    release(rc == BESST_OK);
    if (const char* e = getenv("BESST_INGEST_PROFILE"); e && atoi(e))
        fprintf(stderr, "[push_bam_device] setup %.3f s  alloc %.3f s (%.3f of it waited for)  staging %.3f s  waiting %.3f s  release %.3f s (unpinning %.3f)  total %.3f s\n", setup_s, alloc_s, alloc_wait_s, stage_s,
                wait_s, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_rel).count(), unpin_s,
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count());
    if (rc) return rc;
    c->n_records += pushed;
    c->built = false;
    bam_mark_consumed(bam, (int64_t)saturated);
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        stats->records = pushed;
        stats->chunks = chunks;
        stats->bytes_h2d = comp_total;
        stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
        stats->decode_seconds = stage_s;
        stats->copy_wait_seconds = wait_s;
        stats->inflated_bytes = inflated_total;
        stats->blocks = blocks_total;
        stats->on_device = 1;
        stats->starts_repaired = (int32_t)(repaired > 0x7fffffff ? 0x7fffffff : repaired);
    }
    if (boundary) {
        if (chunks == 0 && first_at < 0) {                   // a slice without a block (more ranks than blocks): what comes in goes out
            first_at = first_skip > 0 ? first_skip : 0;
            carry_out = no_start_left >= 0 ? no_start_left : first_at;
        }
        boundary[0] = first_at;
        boundary[1] = carry_out;
    }
    return BESST_OK;
}
}  // namespace

int besst_ctx_push_bam_device_part(besst_ctx* c, besst_bam* bam, int32_t part, int32_t parts, int64_t chunk_blocks, int64_t head_records,
                                   int32_t* head_rlen, int32_t* head_alen, uint16_t* head_qlen, besst_ingest_stats* stats) {
    return push_bam_device_impl(c, bam, part, parts, chunk_blocks, head_records, head_rlen, head_alen, head_qlen, stats, -1, nullptr);
}

// Slice `part` of `parts` of a file in ANY block layout (multi-GPU ingest of files whose records straddle BGZF blocks): the
// slices are cut at block boundaries as above, and a record belongs to the slice it BEGINS in.  Where a slice's first record
// begins is the one thing a rank cannot know alone: first_skip < 0 lets it guess (the heuristics of the block-to-block
// verification; everything behind the guess is verified as usual), and boundary[0] reports the offset used - in inflated
// bytes from the slice's first block -, boundary[1] how many bytes of the slice's last record lie in the next slice.  The
// callers exchange these two numbers: slice r is right iff boundary[0] of slice r equals boundary[1] of slice r - 1 (slice 0
// begins behind the header and is always right); a slice whose guess was wrong is read again with first_skip = that
// number (besst_amd.distributed.ingest_slice does this).  The bytes of a slice's last record that lie behind its end are
// read from the blocks that follow (at most 4 MiB).
int besst_ctx_push_bam_device_slice(besst_ctx* c, besst_bam* bam, int32_t part, int32_t parts, int64_t chunk_blocks, int64_t first_skip,
                                    int64_t* boundary, int64_t head_records, int32_t* head_rlen, int32_t* head_alen,
                                    uint16_t* head_qlen, besst_ingest_stats* stats) {
    BESST_REQUIRE(boundary, "push_bam_device_slice: boundary is null");
    return push_bam_device_impl(c, bam, part, parts, chunk_blocks, head_records, head_rlen, head_alen, head_qlen, stats, first_skip,
                                boundary);
}

int besst_bgzf_inflate_device(int device, const void* bgzf, size_t n_bytes, void* out, size_t out_cap, size_t* out_len) {
    BESST_REQUIRE(bgzf && out_len && (out || out_cap == 0), "bgzf_inflate_device: null pointer");
    BESST_HIP_TRY(hipSetDevice(device));
    const uint8_t* map = static_cast<const uint8_t*>(bgzf);
    size_t nb = 4096;                                        // (BESST_INFLATE_HOOK_BLOCKS: blocks per launch, for timing runs)
    if (const char* e = getenv("BESST_INFLATE_HOOK_BLOCKS"); e && atoi(e) > 0) nb = (size_t)atoi(e);
    const size_t comp_cap = nb * 65536;
    std::vector<BgzfBlock> desc(nb);
    std::vector<uint32_t> status(nb);
    std::vector<uint8_t> host;
    char* d_comp = nullptr;
    uint8_t* d_inf = nullptr;
    BgzfBlock* d_desc = nullptr;
    uint32_t* d_status = nullptr;
    uint32_t* d_sym = nullptr;
    auto release = [&]() {
        if (d_comp) (void)hipFree(d_comp);
        if (d_inf) (void)hipFree(d_inf);
        if (d_sym) (void)hipFree(d_sym);
        if (d_desc) (void)hipFree(d_desc);
        if (d_status) (void)hipFree(d_status);
    };
    size_t fpos = 0, written = 0, block0 = 0;
    int rc = BESST_OK;
    while (fpos < n_bytes && rc == BESST_OK) {
        const size_t begin = fpos;
        uint32_t n = 0;
        size_t comp = 0, inflated = 0;
        if (!scan_bgzf_chunk(map, n_bytes, &fpos, nb, comp_cap, desc.data(), &n, &comp, &inflated)) {
            set_error("bgzf_inflate_device: not a BGZF block at offset %zu", fpos);
            rc = BESST_ERR_ARG;
            break;
        }
        if (n == 0) break;
        release();
        d_comp = nullptr; d_inf = nullptr; d_desc = nullptr; d_status = nullptr; d_sym = nullptr;
        hipError_t e = hipMalloc((void**)&d_comp, comp + 4096);
        if (e == hipSuccess) e = hipMalloc((void**)&d_inf, inflated + 4096);
        if (e == hipSuccess) e = hipMalloc((void**)&d_sym, 4 * bgzf_inflate_symbol_places(inflated + 4096, n));
        if (e == hipSuccess) e = hipMalloc((void**)&d_desc, (size_t)n * sizeof(BgzfBlock));
        if (e == hipSuccess) e = hipMalloc((void**)&d_status, (size_t)n * 4);
        if (e == hipSuccess) e = hipMemset(d_comp + comp, 0, 4096);
        if (e == hipSuccess) e = hipMemcpy(d_comp, map + begin, comp, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(d_desc, desc.data(), (size_t)n * sizeof(BgzfBlock), hipMemcpyHostToDevice);
        if (e != hipSuccess) { set_error("bgzf_inflate_device: %s", hipGetErrorString(e)); rc = BESST_ERR_HIP; break; }
        if ((rc = launch_bgzf_inflate(nullptr, reinterpret_cast<const uint8_t*>(d_comp), d_desc, n, d_inf, d_status, d_sym))) break;
        host.resize(inflated);
        e = hipMemcpy(status.data(), d_status, (size_t)n * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess && inflated) e = hipMemcpy(host.data(), d_inf, inflated, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { set_error("bgzf_inflate_device: %s", hipGetErrorString(e)); rc = BESST_ERR_HIP; break; }
        for (uint32_t b = 0; b < n; ++b) {
            if (status[b]) {
                set_error("bgzf_inflate_device: block %zu did not inflate (status %u)", block0 + b, status[b]);
                rc = BESST_ERR_UNSUPPORTED;
                break;
            }
            if (written + desc[b].dst_len > out_cap) { set_error("bgzf_inflate_device: output buffer too small"); rc = BESST_ERR_ARG; break; }
            memcpy(static_cast<uint8_t*>(out) + written, host.data() + (((size_t)desc[b].dst_off_hi << 32) | desc[b].dst_off_lo), desc[b].dst_len);
            written += desc[b].dst_len;
        }
        block0 += n;
    }
    release();
    if (rc) return rc;
    *out_len = written;
    return BESST_OK;
}

size_t besst_dev_classify_workspace_bytes(int64_t n_records) { return classify_workspace_bytes(n_records); }
namespace {
__global__ __launch_bounds__(256) void copy_words_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, uint32_t n16,
                                                         const uint8_t* __restrict__ src_tail, uint8_t* __restrict__ dst_tail, uint32_t n_tail) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n16) dst[i] = src[i];
    if (i < n_tail) dst_tail[i] = src_tail[i];
}
}  // namespace

// The state block of a pass (coverage numerators, counters, carry: ~8 bytes per contig) restored from its template at the
// head of every step: one launch of one kernel, a thread per 16 bytes (the runtime's buffer copy took 5.6 us for C2's 80 KB -
// 6 % of that config's whole step).
int besst_dev_restore_state(void* stream, void* dst, const void* src, int64_t bytes) {
    BESST_REQUIRE(bytes >= 0 && (bytes == 0 || (dst && src)), "dev_restore_state: null pointer or negative size");
    BESST_REQUIRE(((uintptr_t)dst & 15u) == 0 && ((uintptr_t)src & 15u) == 0, "dev_restore_state: buffers must be 16-byte aligned");
    if (bytes == 0) return BESST_OK;
    const uint32_t n16 = (uint32_t)(bytes / 16), n_tail = (uint32_t)(bytes % 16);
    const uint32_t blocks = (n16 + 255u) / 256u;
    hipLaunchKernelGGL(copy_words_kernel, dim3(blocks ? blocks : 1u), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const uint4*>(src), static_cast<uint4*>(dst), n16,
                       static_cast<const uint8_t*>(src) + (size_t)n16 * 16, static_cast<uint8_t*>(dst) + (size_t)n16 * 16, n_tail);
    BESST_HIP_TRY(hipGetLastError());
    return BESST_OK;
}

size_t besst_dev_contig_table_bytes(int64_t n_contigs) { return (size_t)(n_contigs > 0 ? n_contigs : 0) * 17 + 16; }
size_t besst_dev_reduce_workspace_bytes(int64_t n_tuples) { return reduce_workspace_bytes(n_tuples); }

int besst_dev_classify(void* stream, int64_t n, const int32_t* tid, const int32_t* mtid, const int32_t* pos,
                       const int32_t* mpos, const uint16_t* flag, const uint8_t* mapq, const uint16_t* qlen,
                       int64_t n_contigs, const void* contig_table, const besst_lib_params* p, int32_t node_bits,
                       int32_t* carry, int64_t* aligned, uint64_t* keys, uint64_t* payload, uint32_t* n_out,
                       besst_counters* counters, void* workspace, size_t workspace_bytes);

int besst_dev_reduce_flags(void* stream, int64_t capacity, const uint32_t* n_tuples, int32_t key_bits, const uint64_t* keys,
                           const uint64_t* payload, uint64_t* row_key, uint32_t* row_mask, uint32_t* row_n, int64_t* row_sum,
                           int64_t* row_sum_sq, uint32_t* row_first, uint32_t* row_offset, int32_t* obs_lo, int32_t* obs_hi,
                           uint32_t* n_rows, void* workspace, size_t workspace_bytes, const uint32_t* first_map,
                           uint64_t key_base, uint32_t flags) {
    BESST_REQUIRE(n_tuples && n_rows, "reduce: null size pointer");
    BESST_REQUIRE(capacity == 0 || (keys && payload && row_key && row_mask && row_n && row_sum && row_sum_sq &&
                                    row_first && row_offset && obs_lo && obs_hi),
                  "reduce: null buffer");
    return launch_sort_reduce(static_cast<hipStream_t>(stream), capacity, n_tuples, key_bits, keys, payload, row_key,
                              row_mask, row_n, row_sum, row_sum_sq, row_first, row_offset, obs_lo, obs_hi, n_rows,
                              workspace, workspace_bytes, first_map, key_base, false, nullptr, flags);
}

int besst_dev_reduce(void* stream, int64_t capacity, const uint32_t* n_tuples, int32_t key_bits, const uint64_t* keys,
                     const uint64_t* payload, uint64_t* row_key, uint32_t* row_mask, uint32_t* row_n, int64_t* row_sum,
                     int64_t* row_sum_sq, uint32_t* row_first, uint32_t* row_offset, int32_t* obs_lo, int32_t* obs_hi,
                     uint32_t* n_rows, void* workspace, size_t workspace_bytes, const uint32_t* first_map,
                     uint64_t key_base) {
    return besst_dev_reduce_flags(stream, capacity, n_tuples, key_bits, keys, payload, row_key, row_mask, row_n, row_sum,
                                  row_sum_sq, row_first, row_offset, obs_lo, obs_hi, n_rows, workspace, workspace_bytes,
                                  first_map, key_base, 0u);
}

static int fill_classify_args(ClassifyArgs& a, int64_t n, const int32_t* tid, const int32_t* mtid, const int32_t* pos,
                              const int32_t* mpos, const uint16_t* flag, const uint8_t* mapq, const uint16_t* qlen,
                              int64_t n_contigs, const void* contig_table, const besst_lib_params* p,
                              int32_t node_bits) {
    BESST_REQUIRE(p, "classify: null params");
    BESST_REQUIRE(n >= 0, "classify: negative record count");
    BESST_REQUIRE(n == 0 || (tid && mtid && pos && mpos && flag && mapq && qlen), "classify: null column");
    BESST_REQUIRE(aligned16(tid) && aligned16(mtid) && aligned16(pos) && aligned16(mpos) && aligned16(flag) &&
                      aligned16(mapq) && aligned16(qlen),
                  "classify: record columns must be 16-byte aligned");
    BESST_REQUIRE(contig_table && aligned16(contig_table), "classify: contig table null or misaligned");
    BESST_REQUIRE(n_contigs > 0 && n_contigs < ((int64_t)1 << 31), "classify: n_contigs out of range");
    BESST_REQUIRE(node_bits >= 1 && node_bits <= 29, "classify: node_bits must be in [1, 29]");
    BESST_REQUIRE(p->orientation == 0 || p->orientation == 1, "classify: orientation must be 0 or 1");
    BESST_REQUIRE(p->ins_size_threshold < 1073741824.0, "classify: ins_size_threshold must be below 2^30");
    a.tid = tid; a.mtid = mtid; a.pos = pos; a.mpos = mpos; a.flag = flag; a.mapq = mapq; a.qlen = qlen;
    a.table = static_cast<const ContigRow*>(contig_table);
    a.cls8 = static_cast<const uint8_t*>(contig_table) + (size_t)n_contigs * sizeof(ContigRow);
    a.n = n;
    a.n_contigs = (int32_t)n_contigs;
    a.node_bits = node_bits;
    a.read_len = p->read_len;
    a.read_len_int = (p->read_len >= 0.0 && p->read_len < 2147483648.0 && p->read_len == floor(p->read_len)) ? (int64_t)p->read_len : -1;
    a.ins_size_threshold = p->ins_size_threshold;
    a.ins_thr_int = p->ins_size_threshold <= -4611686018427387904.0 ? INT64_MIN : (int64_t)ceil(p->ins_size_threshold);   // NaN and >= 2^30 were refused above
    a.min_mapq = p->min_mapq;
    a.rf = p->orientation;
    a.detect_dup = p->detect_duplicate;
    a.extend_paths = p->extend_paths;
    a.no_score = p->no_score;
    a.record_path = p->record_path;
    a.mate_bits = static_cast<const uint8_t*>(p->mate_bits);
    a.ps_table = nullptr; a.ps_rows = 0; a.ps_shift = 0; a.ps_base = 0;
    BESST_REQUIRE(p->record_path == 0 || p->record_path == 1, "classify: record_path must be 0 or 1");
    return BESST_OK;
}

int besst_dev_classify(void* stream, int64_t n, const int32_t* tid, const int32_t* mtid, const int32_t* pos,
                       const int32_t* mpos, const uint16_t* flag, const uint8_t* mapq, const uint16_t* qlen,
                       int64_t n_contigs, const void* contig_table, const besst_lib_params* p, int32_t node_bits,
                       int32_t* carry, int64_t* aligned, uint64_t* keys, uint64_t* payload, uint32_t* n_out,
                       besst_counters* counters, void* workspace, size_t workspace_bytes) {
    ClassifyArgs a;
    int rc = fill_classify_args(a, n, tid, mtid, pos, mpos, flag, mapq, qlen, n_contigs, contig_table, p, node_bits);
    if (rc) return rc;
    BESST_REQUIRE(carry && aligned && keys && payload && n_out && counters, "classify: null output");
    return launch_classify(static_cast<hipStream_t>(stream), a, carry, aligned, keys, payload, n_out, counters,
                           workspace, workspace_bytes);
}

static void presort_to_spec(const besst_presort* h, PresortSpec& ps) {
    ps = PresortSpec{};
    if (!h || !h->table) return;
    ps.table = h->table; ps.rows = h->rows; ps.shift = h->shift; ps.key_base = h->key_base; ps.cap = h->capacity;
    ps.segmented = h->segmented; ps.in_record_loop = h->in_record_loop;
    // the run-grouped stage 2 (what a stream with this description takes unless the flag says otherwise) has no use for
    // the digit histograms: the record loop then does not count them
    ps.count = (runs_enabled((int64_t)h->capacity) && !(h->flags & BESST_REDUCE_NO_RUNS)) ? 0 : 1;
    ps.seg = SegSource{h->seg_keys, h->seg_payload, h->seg_offsets, h->seg_skip, h->seg_blocks, h->seg_tile, h->payload_out,
                       h->seg_chunk_first, h->seg_run_offsets, h->seg_summ, h->seg_summ_stride, h->seg_run_status};
}

int besst_dev_reduce_presort(int64_t capacity, int32_t key_bits, uint64_t key_base, void* workspace,
                             size_t workspace_bytes, besst_presort* h_out) {
    BESST_REQUIRE(h_out, "reduce_presort: null output");
    memset(h_out, 0, sizeof(*h_out));
    PresortSpec ps{};
    if (!sort_presort_spec(capacity, key_bits, key_base, workspace, workspace_bytes, &ps)) return 0;
    h_out->table = ps.table; h_out->rows = ps.rows; h_out->shift = ps.shift; h_out->key_base = ps.key_base;
    h_out->capacity = ps.cap;
    h_out->segmented = 1;            // every stream that takes the table can also be read from its block segments
    return 1;
}

int besst_dev_classify_presort(void* stream, int64_t n, const int32_t* tid, const int32_t* mtid, const int32_t* pos,
                               const int32_t* mpos, const uint16_t* flag, const uint8_t* mapq, const uint16_t* qlen,
                               int64_t n_contigs, const void* contig_table, const besst_lib_params* p, int32_t node_bits,
                               int32_t* carry, int64_t* aligned, uint64_t* keys, uint64_t* payload, uint32_t* n_out,
                               besst_counters* counters, void* workspace, size_t workspace_bytes,
                               besst_presort* h_presort) {
    ClassifyArgs a;
    int rc = fill_classify_args(a, n, tid, mtid, pos, mpos, flag, mapq, qlen, n_contigs, contig_table, p, node_bits);
    if (rc) return rc;
    BESST_REQUIRE(carry && aligned && keys && payload && n_out && counters, "classify: null output");
    PresortSpec ps;
    presort_to_spec(h_presort, ps);
    ps.in_record_loop = 0;
    rc = launch_classify(static_cast<hipStream_t>(stream), a, carry, aligned, keys, payload, n_out, counters,
                         workspace, workspace_bytes, ps.table ? &ps : nullptr);
    if (h_presort && h_presort->table) {
        h_presort->segmented = ps.segmented; h_presort->in_record_loop = ps.in_record_loop;
        h_presort->seg_keys = ps.seg.seg_keys; h_presort->seg_payload = ps.seg.seg_payload;
        h_presort->seg_offsets = ps.seg.offsets; h_presort->seg_skip = ps.seg.skip;
        h_presort->seg_blocks = ps.seg.nblocks; h_presort->seg_tile = ps.seg.tile; h_presort->payload_out = ps.seg.payload_out;
        h_presort->seg_chunk_first = ps.seg.chunk_first;
        h_presort->seg_run_offsets = ps.seg.run_offsets; h_presort->seg_summ = ps.seg.summ;
        h_presort->seg_summ_stride = ps.seg.summ_stride; h_presort->seg_run_status = ps.seg.run_status;
    }
    return rc;
}

int besst_dev_reduce_presorted(void* stream, int64_t capacity, const uint32_t* n_tuples, int32_t key_bits,
                               const uint64_t* keys, const uint64_t* payload, uint64_t* row_key, uint32_t* row_mask,
                               uint32_t* row_n, int64_t* row_sum, int64_t* row_sum_sq, uint32_t* row_first,
                               uint32_t* row_offset, int32_t* obs_lo, int32_t* obs_hi, uint32_t* n_rows,
                               void* workspace, size_t workspace_bytes, const uint32_t* first_map, uint64_t key_base,
                               const besst_presort* h_presort) {
    BESST_REQUIRE(n_tuples && n_rows, "reduce: null size pointer");
    BESST_REQUIRE(h_presort && h_presort->table, "reduce_presorted: no hand-over description");
    PresortSpec ps;
    presort_to_spec(h_presort, ps);
    const bool seg = ps.segmented != 0;
    BESST_REQUIRE(capacity == 0 || ((keys || seg) && payload && row_key && row_mask && row_n && row_sum && row_sum_sq &&
                                    row_first && row_offset && obs_lo && obs_hi),
                  "reduce: null buffer");
    BESST_REQUIRE(!seg || (ps.seg.seg_keys && ps.seg.seg_payload && ps.seg.offsets && ps.seg.skip &&
                           ps.seg.payload_out == payload && ps.seg.tile > 0),
                  "reduce_presorted: incomplete segment description");
    BESST_REQUIRE(h_presort->in_record_loop != 3 || (seg && ps.seg.run_offsets && ps.seg.summ && ps.seg.run_status),
                  "reduce_presorted: incomplete description of the record loop's runs");
    if (h_presort->in_record_loop >= 2 && (h_presort->flags & BESST_REDUCE_NO_RUNS)) {
        set_error("reduce_presorted: the classify call behind this description did not count the sort's digits (its flags "
                  "asked for the run-grouped form): repeat it with BESST_REDUCE_NO_RUNS in `flags`");
        return BESST_ERR_STATE;
    }
    return launch_sort_reduce(static_cast<hipStream_t>(stream), capacity, n_tuples, key_bits, keys, payload, row_key,
                              row_mask, row_n, row_sum, row_sum_sq, row_first, row_offset, obs_lo, obs_hi, n_rows,
                              workspace, workspace_bytes, first_map, key_base, true, seg ? &ps.seg : nullptr,
                              h_presort->flags);
}

int besst_dev_candidate_density(void* stream, int64_t n, const int32_t* tid, const int32_t* mtid,
                                int64_t sample_records, void* counts_scratch, double* h_share,
                                int32_t* h_record_path) {
    BESST_REQUIRE(n >= 0 && (n == 0 || (tid && mtid)) && counts_scratch && h_share && h_record_path,
                  "candidate_density: bad argument");
    unsigned long long host[2] = {0ull, 0ull};
    if (n > 0) {
        int rc = launch_candidate_density(static_cast<hipStream_t>(stream), n, tid, mtid, sample_records,
                                          static_cast<unsigned long long*>(counts_scratch));
        if (rc) return rc;
        BESST_HIP_TRY(hipMemcpyAsync(host, counts_scratch, sizeof(host), hipMemcpyDeviceToHost,
                                     static_cast<hipStream_t>(stream)));
        BESST_HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    }
    *h_share = host[1] ? (double)host[0] / (double)host[1] : 0.0;
    *h_record_path = *h_share >= BESST_DENSE_CANDIDATE_SHARE ? 1 : 0;
    return BESST_OK;
}

int besst_dev_classify_scan(void* stream, int64_t n, const int32_t* tid, const int32_t* mtid, const int32_t* pos,
                            const int32_t* mpos, const uint16_t* flag, const uint8_t* mapq, const uint16_t* qlen,
                            int64_t n_contigs, const void* contig_table, const besst_lib_params* p, int32_t node_bits,
                            int64_t* aligned, besst_counters* counters, void* workspace, size_t workspace_bytes) {
    ClassifyArgs a;
    int rc = fill_classify_args(a, n, tid, mtid, pos, mpos, flag, mapq, qlen, n_contigs, contig_table, p, node_bits);
    if (rc) return rc;
    BESST_REQUIRE(aligned && counters, "classify_scan: null output");
    return launch_classify_scan(static_cast<hipStream_t>(stream), a, aligned, counters, workspace, workspace_bytes);
}

int besst_dev_classify_tail(void* stream, int64_t n, int32_t* tail, void* workspace, size_t workspace_bytes) {
    BESST_REQUIRE(tail, "classify_tail: null output");
    return launch_classify_tail(static_cast<hipStream_t>(stream), n, tail, workspace, workspace_bytes);
}

int besst_dev_resolve_carry(void* stream, const int32_t* tails, int32_t rank, int32_t* carry) {
    BESST_REQUIRE(tails && carry && rank >= 0, "resolve_carry: bad argument");
    return launch_resolve_carry(static_cast<hipStream_t>(stream), tails, rank, carry);
}

int besst_dev_classify_emit(void* stream, int64_t n, int32_t detect_duplicate, int32_t* carry, uint64_t* keys,
                            uint64_t* payload, uint32_t* n_out, besst_counters* counters, void* workspace,
                            size_t workspace_bytes, int64_t n_contigs, const void* contig_table, int64_t* aligned,
                            const int32_t* tails, int32_t rank, int32_t* slice_info) {
    BESST_REQUIRE(!(tails && slice_info), "classify_emit: tails and slice_info exclude each other");
    BESST_REQUIRE(carry && keys && payload && n_out && counters && contig_table && aligned,
                  "classify_emit: null pointer");
    BESST_REQUIRE(n_contigs > 0 && n_contigs < ((int64_t)1 << 31), "classify_emit: n_contigs out of range");
    BESST_REQUIRE(rank >= 0 && rank <= 65536, "classify_emit: rank out of range");
    const uint8_t* cls8 = static_cast<const uint8_t*>(contig_table) + (size_t)n_contigs * sizeof(ContigRow);
    return launch_classify_emit(static_cast<hipStream_t>(stream), n, detect_duplicate, carry, keys, payload, n_out,
                                counters, workspace, workspace_bytes, cls8, (int32_t)n_contigs, aligned, tails, rank,
                                slice_info);
}

size_t besst_dev_exchange_region_bytes(int64_t pair_capacity) { return exchange_region_bytes(pair_capacity); }
size_t besst_dev_exchange_stride_bytes(int64_t pair_capacity, int64_t rider_bytes) {
    return exchange_stride_bytes(pair_capacity, rider_bytes < 0 ? 0 : rider_bytes);
}

uint32_t besst_owner_of_scaffold(uint32_t scaffold_id, uint32_t world) { return owner_of_scaffold(scaffold_id, world ? world : 1); }

int besst_dev_partition(void* stream, int64_t capacity, const uint32_t* n_tuples, int32_t node_bits, int32_t world,
                        const uint64_t* keys, const uint64_t* payload, int64_t pair_capacity, void* send_buffer,
                        void* workspace, size_t workspace_bytes, const void* rider, int64_t rider_bytes,
                        const int32_t* slice_info) {
    BESST_REQUIRE(n_tuples && keys && payload && send_buffer, "partition: null pointer");
    return launch_partition(static_cast<hipStream_t>(stream), capacity, n_tuples, node_bits, world, keys, payload,
                            pair_capacity, send_buffer, workspace, workspace_bytes, rider, rider_bytes, slice_info);
}

int besst_dev_unpack(void* stream, int32_t world, int64_t pair_capacity, const void* recv_buffer, uint64_t* keys,
                     uint64_t* payload, uint32_t* gidx, uint32_t* n_out, uint32_t* overflow, void* rider_sum,
                     int64_t rider_bytes, int32_t speculative_heads, int32_t rank, int32_t detect_duplicate,
                     int32_t* all_slice_info, besst_counters* counters) {
    BESST_REQUIRE(recv_buffer && keys && payload && gidx && n_out && overflow, "unpack: null pointer");
    return launch_unpack(static_cast<hipStream_t>(stream), world, pair_capacity, recv_buffer, keys, payload, gidx,
                         n_out, overflow, rider_sum, rider_bytes, speculative_heads, rank, detect_duplicate,
                         all_slice_info, counters);
}

int besst_dev_score_edges(void* stream, int64_t n_edges, const uint32_t* row, const uint8_t* swap, const int32_t* len1,
                          const int32_t* len2, const uint32_t* row_n, const int64_t* row_sum, const uint32_t* row_offset,
                          const int32_t* obs_lo, const int32_t* obs_hi, double mean, double sigma, double read_len,
                          double* gap, double* sd0, int32_t* ks_h, uint8_t* flags, void* workspace,
                          size_t workspace_bytes) {
    BESST_REQUIRE(n_edges >= 0 && n_edges < ((int64_t)1 << 31), "dev_score_edges: edge count out of range");
    if (n_edges == 0) return BESST_OK;
    BESST_REQUIRE(row && swap && len1 && len2 && row_n && row_sum && row_offset && obs_lo && obs_hi && gap && sd0 &&
                      ks_h && flags && workspace,
                  "dev_score_edges: null pointer");
    BESST_REQUIRE(sigma > 0.0, "dev_score_edges: sigma must be positive");
    BESST_REQUIRE(workspace_bytes >= align_up((size_t)n_edges * 8, 256), "dev_score_edges: workspace too small");
    ScoreArgs a;
    a.row = row; a.swap = swap; a.len1 = len1; a.len2 = len2;
    a.row_n = row_n; a.row_sum = row_sum; a.row_offset = row_offset;
    a.obs_lo = obs_lo; a.obs_hi = obs_hi;
    a.mean = mean; a.sigma = sigma; a.read_len = read_len;
    a.n_edges = n_edges;
    return launch_score(static_cast<hipStream_t>(stream), a, gap, sd0, ks_h, flags, workspace, workspace_bytes);
}

int besst_dev_score_edges_lognormal(void* stream, int64_t n_edges, const uint32_t* row, const uint8_t* swap,
                                    const int32_t* len1, const int32_t* len2, const uint32_t* row_n, const int64_t* row_sum,
                                    const uint32_t* row_offset, const int32_t* obs_lo, const int32_t* obs_hi, double mean,
                                    double sigma, double read_len, double ln_mu, double ln_sigma, int64_t x_max,
                                    const double* F0, const double* F1, int32_t max_gap, double* gap, int32_t* ks_h,
                                    uint8_t* flags, void* workspace, size_t workspace_bytes) {
    BESST_REQUIRE(n_edges >= 0 && n_edges < ((int64_t)1 << 31), "dev_score_edges_lognormal: edge count out of range");
    if (n_edges == 0) return BESST_OK;
    BESST_REQUIRE(row && swap && len1 && len2 && row_n && row_sum && row_offset && obs_lo && obs_hi && gap && ks_h &&
                      flags && workspace && F0 && F1,
                  "dev_score_edges_lognormal: null pointer");
    BESST_REQUIRE(sigma > 0.0 && ln_sigma > 0.0 && x_max >= 1 && max_gap >= 0, "dev_score_edges_lognormal: parameters out of range");
    // [ big_off | sd0 (unused by the caller: the conditional sigma is looked up with the gap) | sort scratch ]
    const size_t head = align_up((size_t)n_edges * 8, 256);
    BESST_REQUIRE(workspace_bytes >= 2 * head, "dev_score_edges_lognormal: workspace too small");
    ScoreArgs a;
    a.row = row; a.swap = swap; a.len1 = len1; a.len2 = len2;
    a.row_n = row_n; a.row_sum = row_sum; a.row_offset = row_offset;
    a.obs_lo = obs_lo; a.obs_hi = obs_hi;
    a.mean = mean; a.sigma = sigma; a.read_len = read_len;
    a.n_edges = n_edges;
    LogNormalArgs l{ln_mu, ln_sigma, x_max, F0, F1, max_gap};
    // the kernel wants the sort scratch right behind the offsets: the sd0 column sits at the END of the workspace
    auto* sd0 = reinterpret_cast<double*>(static_cast<char*>(workspace) + workspace_bytes - head);
    return launch_score_lognormal(static_cast<hipStream_t>(stream), a, l, gap, sd0, ks_h, flags, workspace, workspace_bytes - head);
}

size_t besst_dev_mate_bits_bytes(int64_t n_records) { return (size_t)(((n_records > 0 ? n_records : 0) + 7) / 8) + 16; }

int besst_dev_mate_bits(void* stream, int64_t n, const int32_t* tid, const int32_t* mtid, void* bits) {
    BESST_REQUIRE(n >= 0, "dev_mate_bits: negative record count");
    if (n == 0) return BESST_OK;
    BESST_REQUIRE(tid && mtid && bits, "dev_mate_bits: null pointer");
    BESST_REQUIRE(aligned16(tid) && aligned16(mtid), "dev_mate_bits: columns must be 16-byte aligned");
    return launch_mate_bits(static_cast<hipStream_t>(stream), tid, mtid, 0, n, n, static_cast<uint8_t*>(bits));
}

size_t besst_dev_lognormal_tables_workspace_bytes(int64_t x_max) { return lognormal_tables_workspace_bytes(x_max); }

int besst_dev_lognormal_tables(void* stream, double mu, double sigma, int64_t x_max, double* F0, double* F1, void* workspace,
                               size_t workspace_bytes) {
    return launch_lognormal_tables(static_cast<hipStream_t>(stream), mu, sigma, x_max, F0, F1, workspace, workspace_bytes);
}

int besst_dev_conditional_stddevs(void* stream, const double* density, int64_t max_isize, const int32_t* steps,
                                  int32_t n_steps, double* out) {
    BESST_REQUIRE(n_steps >= 0, "dev_conditional_stddevs: negative step count");
    return launch_conditional_stddevs(static_cast<hipStream_t>(stream), density, max_isize, steps, n_steps, out);
}

size_t besst_dev_metrics_workspace_bytes(int64_t n_records) { return metrics_workspace_bytes(n_records < 1 ? 1 : n_records); }

int besst_dev_metrics_sample(void* stream, int64_t n, const int32_t* tid, const int32_t* mtid, const int32_t* tlen,
                             const uint16_t* flag, const uint8_t* mapq, int64_t n_contigs, const uint8_t* top_mask,
                             int32_t orientation, int32_t min_mapq, double read_len, int32_t count_only,
                             int32_t* isize_out, int32_t* contam_out, int64_t* state, void* workspace,
                             size_t workspace_bytes) {
    // (every record index in metrics.hip is 64 bits wide and the sampling form works in parts of 64 Mi records; the bound is
    // what the count-only form's one launch of n / 4096 workgroups and their uint32 counts were checked for)
    BESST_REQUIRE(n >= 0 && n < ((int64_t)1 << 36), "dev_metrics_sample: n out of range");
    BESST_REQUIRE(n_contigs > 0 && n_contigs < ((int64_t)1 << 31), "dev_metrics_sample: n_contigs out of range");
    BESST_REQUIRE(orientation == 0 || orientation == 1, "dev_metrics_sample: orientation must be 0 or 1");
    BESST_REQUIRE(state && top_mask, "dev_metrics_sample: null pointer");
    BESST_REQUIRE(count_only || contam_out, "dev_metrics_sample: contam_out is null");
    BESST_REQUIRE(n == 0 || (tid && mtid && tlen && flag && mapq), "dev_metrics_sample: null column");
    MetricsArgs a;
    a.tid = tid; a.mtid = mtid; a.tlen = tlen; a.flag = flag; a.mapq = mapq;
    a.top_mask = top_mask;
    a.n = n;
    a.n_contigs = (int32_t)n_contigs;
    a.rf = orientation;
    a.min_mapq = min_mapq;
    a.read_len = read_len;
    return launch_metrics(static_cast<hipStream_t>(stream), a, 0, n, isize_out, contam_out, state, workspace,
                          workspace_bytes, count_only != 0);
}

int besst_ctx_build_graph(besst_ctx* c) {
    BESST_REQUIRE(c, "null context");
    if (!c->have_lib || c->n_contigs <= 0) {
        set_error("build_graph: set_contigs and set_library must be called first");
        return BESST_ERR_STATE;
    }
    int rc = use_device(c);
    if (rc) return rc;
    const int64_t n = c->n_records;
    auto* sb = reinterpret_cast<SmallBlock*>(c->small.p);
    SmallBlock init;
    memset(&init, 0, sizeof(init));
    init.carry[0] = -1;   // counters(0, 0, 0, 0, -1, -1, 0): CreateGraph.py:98
    init.carry[1] = -1;
    BESST_HIP_TRY(hipMemcpyAsync(sb, &init, sizeof(init), hipMemcpyHostToDevice, c->stream));
    BESST_HIP_TRY(hipMemsetAsync(c->aligned.p, 0, (size_t)c->n_contigs * sizeof(int64_t), c->stream));
    const size_t cap1 = (size_t)(n > 0 ? n : 1);
    if ((rc = c->keys.ensure(cap1))) return rc;
    if ((rc = c->payload.ensure(cap1))) return rc;
    if ((rc = c->ws.ensure(classify_workspace_bytes(n)))) return rc;
    besst_lib_params lp = c->lib;
    {   // which form of the record loop: one fused pass for candidate-dense (mate-pair) libraries
        double share = 0.0;
        if ((rc = c->aux.ensure(64))) return rc;
        rc = besst_dev_candidate_density(c->stream, n, c->tid.p, c->mtid.p, 4 << 20, c->aux.p, &share, &lp.record_path);
        if (rc) return rc;
        const char* forced = getenv("BESST_RECORD_PATH");   // tests and experiments: "0" / "1"
        if (forced && (forced[0] == '0' || forced[0] == '1') && forced[1] == 0) lp.record_path = forced[0] - '0';
    }
    {   // the mate-elsewhere bits of the records that have none yet (everything pushed since the last build): 8 bytes read
        // per record once, after which every pass over these records leaves `mtid` alone where tid == mtid.
        // BESST_MATE_BITS=0 (tests): the loop compares the columns itself.
        const char* off = getenv("BESST_MATE_BITS");
        lp.mate_bits = nullptr;
        if (!(off && off[0] == '0' && off[1] == 0) && n > 0) {
            const size_t need = (size_t)((n + 7) / 8) + 16;
            if (need > c->mate_bits.cap) {                   // (growing drops what was there)
                if ((rc = c->mate_bits.ensure(need + need / 2))) return rc;
                c->bits_upto = 0;
            }
            if (c->bits_upto < n) {
                if ((rc = launch_mate_bits(c->stream, c->tid.p, c->mtid.p, c->bits_upto, n, n, c->mate_bits.p))) return rc;
                c->bits_upto = n;
            }
            lp.mate_bits = c->mate_bits.p;
        }
    }
    rc = besst_dev_classify(c->stream, n, c->tid.p, c->mtid.p, c->pos.p, c->mpos.p, c->flag.p, c->mapq.p, c->qlen.p,
                            c->n_contigs, c->table.p, &lp, c->node_bits, sb->carry, c->aligned.p, c->keys.p,
                            c->payload.p, &sb->n_out, &sb->counters, c->ws.p, c->ws.cap);
    if (rc) return rc;
    SmallBlock host;
    BESST_HIP_TRY(hipMemcpyAsync(&host, sb, sizeof(host), hipMemcpyDeviceToHost, c->stream));
    BESST_HIP_TRY(hipStreamSynchronize(c->stream));
    const int64_t L = host.n_out;
    c->n_tuples = L;
    const size_t cap2 = (size_t)(L > 0 ? L : 1);
    if ((rc = c->row_key.ensure(cap2))) return rc;
    if ((rc = c->row_mask.ensure(cap2))) return rc;
    if ((rc = c->row_n.ensure(cap2))) return rc;
    if ((rc = c->row_first.ensure(cap2))) return rc;
    if ((rc = c->row_offset.ensure(cap2))) return rc;
    if ((rc = c->row_sum.ensure(cap2))) return rc;
    if ((rc = c->row_sum_sq.ensure(cap2))) return rc;
    if ((rc = c->obs_lo.ensure(cap2))) return rc;
    if ((rc = c->obs_hi.ensure(cap2))) return rc;
    if ((rc = c->ws.ensure(reduce_workspace_bytes(L)))) return rc;
    // large streams take the run-grouped form first; one whose keys do not cluster says so in n_rows and is sorted
    // tuple by tuple instead (include/besst_amd.h, BESST_ROWS_*)
    for (uint32_t flags = 0;; flags = BESST_REDUCE_NO_RUNS) {
        rc = besst_dev_reduce_flags(c->stream, L, &sb->n_out, c->key_bits, c->keys.p, c->payload.p, c->row_key.p,
                                    c->row_mask.p, c->row_n.p, c->row_sum.p, c->row_sum_sq.p, c->row_first.p,
                                    c->row_offset.p, c->obs_lo.p, c->obs_hi.p, &sb->n_rows, c->ws.p, c->ws.cap, nullptr,
                                    c->key_base, flags);
        if (rc) return rc;
        BESST_HIP_TRY(hipMemcpyAsync(&host, sb, sizeof(host), hipMemcpyDeviceToHost, c->stream));
        BESST_HIP_TRY(hipStreamSynchronize(c->stream));
        if (host.n_rows == BESST_ROWS_RUN_OVERFLOW && flags == 0) continue;
        break;
    }
    if (host.n_rows == BESST_ROWS_SORT_FAILED || host.n_rows == BESST_ROWS_RUN_OVERFLOW) {
        set_error("build_graph: the sort could not finish (status word 0x%08x): a chained-scan look-back gave up", host.n_rows);
        return BESST_ERR_HIP;
    }
    c->n_rows = host.n_rows;
    c->built = true;
    return BESST_OK;
}

#define BESST_NEED_BUILT(c)                                                   \
    do {                                                                      \
        BESST_REQUIRE(c, "null context");                                     \
        if (!(c)->built) {                                                    \
            set_error("no edge table: call besst_ctx_build_graph first");     \
            return BESST_ERR_STATE;                                           \
        }                                                                     \
    } while (0)

int besst_ctx_edge_count(besst_ctx* c, int64_t* n_rows, int64_t* n_tuples) {
    BESST_NEED_BUILT(c);
    if (n_rows) *n_rows = c->n_rows;
    if (n_tuples) *n_tuples = c->n_tuples;
    return BESST_OK;
}

int besst_ctx_fetch_edges(besst_ctx* c, uint64_t* key, uint32_t* mask, uint32_t* n, int64_t* sum_obs,
                          int64_t* sum_obs_sq, uint32_t* first_idx, uint32_t* offset, int32_t* node_bits) {
    BESST_NEED_BUILT(c);
    int rc = use_device(c);
    if (rc) return rc;
    const size_t r = (size_t)c->n_rows;
    if (node_bits) *node_bits = c->node_bits;
    if (r) {
        BESST_REQUIRE(key && mask && n && sum_obs && sum_obs_sq && first_idx && offset, "fetch_edges: null buffer");
        BESST_HIP_TRY(hipMemcpyAsync(key, c->row_key.p, r * 8, hipMemcpyDeviceToHost, c->stream));
        BESST_HIP_TRY(hipMemcpyAsync(mask, c->row_mask.p, r * 4, hipMemcpyDeviceToHost, c->stream));
        BESST_HIP_TRY(hipMemcpyAsync(n, c->row_n.p, r * 4, hipMemcpyDeviceToHost, c->stream));
        BESST_HIP_TRY(hipMemcpyAsync(sum_obs, c->row_sum.p, r * 8, hipMemcpyDeviceToHost, c->stream));
        BESST_HIP_TRY(hipMemcpyAsync(sum_obs_sq, c->row_sum_sq.p, r * 8, hipMemcpyDeviceToHost, c->stream));
        BESST_HIP_TRY(hipMemcpyAsync(first_idx, c->row_first.p, r * 4, hipMemcpyDeviceToHost, c->stream));
        BESST_HIP_TRY(hipMemcpyAsync(offset, c->row_offset.p, r * 4, hipMemcpyDeviceToHost, c->stream));
        BESST_HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return BESST_OK;
}

int besst_ctx_fetch_observations(besst_ctx* c, int32_t* obs_lo, int32_t* obs_hi) {
    BESST_NEED_BUILT(c);
    int rc = use_device(c);
    if (rc) return rc;
    const size_t L = (size_t)c->n_tuples;
    if (L) {
        BESST_REQUIRE(obs_lo && obs_hi, "fetch_observations: null buffer");
        BESST_HIP_TRY(hipMemcpyAsync(obs_lo, c->obs_lo.p, L * 4, hipMemcpyDeviceToHost, c->stream));
        BESST_HIP_TRY(hipMemcpyAsync(obs_hi, c->obs_hi.p, L * 4, hipMemcpyDeviceToHost, c->stream));
        BESST_HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return BESST_OK;
}

namespace {
__global__ __launch_bounds__(256) void obs_sum_kernel(const int32_t* __restrict__ lo, const int32_t* __restrict__ hi, int32_t* __restrict__ out,
                                                      long long n) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) out[i] = lo[i] + hi[i];
}
}  // namespace

// One observation per link (obs1 + obs2: what the reference keeps in an edge's 'observations', CreateGraph.py:849,862),
// summed on the device and copied on a stream of the context's own: the ONE call of the ctx layer that may run on a second
// host thread beside the calling thread's (score_edges, the fetches): it touches nothing those use but reads obs_lo / obs_hi.
int besst_ctx_fetch_observation_sums(besst_ctx* c, int32_t* out) {
    BESST_NEED_BUILT(c);
    if (hipSetDevice(c->device) != hipSuccess) { set_error("fetch_observation_sums: cannot select device %d", c->device); return BESST_ERR_HIP; }
    const int64_t L = c->n_tuples;
    if (L <= 0) return BESST_OK;
    BESST_REQUIRE(out, "fetch_observation_sums: null buffer");
    if (!c->side_stream) BESST_HIP_TRY(hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking));
    int rc = c->obs_sum.ensure((size_t)L);
    if (rc) return rc;
    const int64_t want = (L + 1023) / 1024;
    hipLaunchKernelGGL(obs_sum_kernel, dim3((unsigned)(want < 4096 ? want : 4096)), dim3(256), 0, c->side_stream, c->obs_lo.p, c->obs_hi.p,
                       c->obs_sum.p, (long long)L);
    BESST_HIP_TRY(hipGetLastError());
    BESST_HIP_TRY(hipMemcpyAsync(out, c->obs_sum.p, (size_t)L * 4, hipMemcpyDeviceToHost, c->side_stream));
    BESST_HIP_TRY(hipStreamSynchronize(c->side_stream));
    return BESST_OK;
}

int besst_ctx_fetch_coverage(besst_ctx* c, int64_t* aligned) {
    BESST_NEED_BUILT(c);
    BESST_REQUIRE(aligned, "fetch_coverage: null buffer");
    int rc = use_device(c);
    if (rc) return rc;
    BESST_HIP_TRY(hipMemcpyAsync(aligned, c->aligned.p, (size_t)c->n_contigs * 8, hipMemcpyDeviceToHost, c->stream));
    BESST_HIP_TRY(hipStreamSynchronize(c->stream));
    return BESST_OK;
}

int besst_ctx_fetch_counters(besst_ctx* c, besst_counters* out) {
    BESST_NEED_BUILT(c);
    BESST_REQUIRE(out, "fetch_counters: null buffer");
    int rc = use_device(c);
    if (rc) return rc;
    SmallBlock host;
    BESST_HIP_TRY(hipMemcpyAsync(&host, c->small.p, sizeof(host), hipMemcpyDeviceToHost, c->stream));
    BESST_HIP_TRY(hipStreamSynchronize(c->stream));
    *out = host.counters;
    out->prev_obs1 = host.carry[0];
    out->prev_obs2 = host.carry[1];
    return BESST_OK;
}

}  // extern "C"

extern "C" {

int besst_dev_stream_order(void* stream, int64_t n, const int32_t* tid, const int32_t* pos, int64_t* first_unsorted) {
    BESST_REQUIRE(n >= 0 && first_unsorted && (n == 0 || (tid && pos)), "dev_stream_order: null pointer or negative count");
    hipStream_t s = static_cast<hipStream_t>(stream);
    BESST_HIP_TRY(hipMemsetAsync(first_unsorted, 0xff, sizeof(int64_t), s));     // -1: sorted
    return launch_stream_order(s, tid, pos, n, reinterpret_cast<unsigned long long*>(first_unsorted));
}

int besst_ctx_stream_order(besst_ctx* c, int64_t* first_unsorted, int32_t* first_key, int32_t* last_key) {
    BESST_REQUIRE(c && first_unsorted, "stream_order: null pointer");
    int rc = use_device(c);
    if (rc) return rc;
    if ((rc = c->aux.ensure(64))) return rc;
    int64_t* word = reinterpret_cast<int64_t*>(c->aux.p);
    if ((rc = besst_dev_stream_order(c->stream, c->n_records, c->tid.p, c->pos.p, word))) return rc;
    BESST_HIP_TRY(hipMemcpyAsync(first_unsorted, word, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    // (tid, pos) of the first and the last resident record: what a slice of a sharded stream compares with its neighbours
    const int64_t n = c->n_records;
    if (first_key) {
        first_key[0] = first_key[1] = 0;
        if (n > 0) {
            BESST_HIP_TRY(hipMemcpyAsync(first_key, c->tid.p, 4, hipMemcpyDeviceToHost, c->stream));
            BESST_HIP_TRY(hipMemcpyAsync(first_key + 1, c->pos.p, 4, hipMemcpyDeviceToHost, c->stream));
        }
    }
    if (last_key) {
        last_key[0] = last_key[1] = 0;
        if (n > 0) {
            BESST_HIP_TRY(hipMemcpyAsync(last_key, c->tid.p + (n - 1), 4, hipMemcpyDeviceToHost, c->stream));
            BESST_HIP_TRY(hipMemcpyAsync(last_key + 1, c->pos.p + (n - 1), 4, hipMemcpyDeviceToHost, c->stream));
        }
    }
    BESST_HIP_TRY(hipStreamSynchronize(c->stream));
    return BESST_OK;
}

int besst_ctx_metrics_sample(besst_ctx* c, const uint8_t* top_mask, int32_t orientation, int32_t min_mapq,
                             double read_len, int32_t want_isize, int32_t* isize_out, int32_t* contam_out,
                             besst_metrics_counts* counts) {
    BESST_REQUIRE(c && top_mask && contam_out && counts, "metrics_sample: null pointer");
    BESST_REQUIRE(!want_isize || isize_out, "metrics_sample: isize_out is null");
    BESST_REQUIRE(orientation == 0 || orientation == 1, "metrics_sample: orientation must be 0 or 1");
    if (c->n_contigs <= 0) {
        set_error("metrics_sample: set_contigs must be called first (defines the reference count)");
        return BESST_ERR_STATE;
    }
    int rc = use_device(c);
    if (rc) return rc;
    constexpr int64_t kCap = 1000000;
    constexpr int64_t kChunk = 16 << 20;      // (most libraries fill their samples within it: one launch)
    if ((rc = c->top_mask.ensure((size_t)c->n_contigs))) return rc;
    if ((rc = c->sample_a.ensure((size_t)kCap))) return rc;
    if ((rc = c->sample_b.ensure((size_t)kCap))) return rc;
    // Chunks: the reference stops each scan at its 1,000,000th qualifying record (libmetrics.py:83,302), the host learns the
    // counts between chunks.  The first chunk is kChunk records; each later one is sized from the rate seen so far so that
    // it should finish the scan (x 1.25, at most kChunkMax): a launch + a round trip to the host per 4 M records was most of
    // the pass's time when the samples fill late (60 M records: 15 chunks, 1.1 ms for 0.18 ms worth of bytes).
    constexpr int64_t kChunkMax = (int64_t)256 << 20;
    const int64_t ws_records = c->n_records < kChunkMax ? (c->n_records > kChunk ? c->n_records : kChunk) : kChunkMax;
    if ((rc = c->aux.ensure(metrics_workspace_bytes(ws_records) + 64))) return rc;
    BESST_HIP_TRY(hipMemcpyAsync(c->top_mask.p, top_mask, (size_t)c->n_contigs, hipMemcpyHostToDevice, c->stream));
    int64_t* state = reinterpret_cast<int64_t*>(c->aux.p);
    char* ws = c->aux.p + 64;
    BESST_HIP_TRY(hipMemsetAsync(state, 0, 64, c->stream));
    MetricsArgs a;
    a.tid = c->tid.p; a.mtid = c->mtid.p; a.tlen = c->tlen.p; a.flag = c->flag.p; a.mapq = c->mapq.p;
    a.top_mask = c->top_mask.p;
    a.n = c->n_records;
    a.n_contigs = (int32_t)c->n_contigs;
    a.rf = orientation;
    a.min_mapq = min_mapq;
    a.read_len = read_len;
    int64_t host[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t next = kChunk;
    for (int64_t start = 0; start < c->n_records;) {
        const int64_t cnt = c->n_records - start < next ? c->n_records - start : next;
        rc = launch_metrics(c->stream, a, start, cnt, want_isize ? c->sample_a.p : nullptr, c->sample_b.p, state, ws,
                            c->aux.cap - 64);
        if (rc) return rc;
        BESST_HIP_TRY(hipMemcpyAsync(host, state, 48, hipMemcpyDeviceToHost, c->stream));
        BESST_HIP_TRY(hipStreamSynchronize(c->stream));
        start += cnt;
        // the reference stops each scan once its 1,000,000-sample cut-off is reached (libmetrics.py:83,302)
        if ((!want_isize || host[0] >= kCap) && host[1] >= kCap) break;
        // records still to scan at the rate of the slower of the two counts (none seen yet: as many as allowed)
        const int64_t slow = (want_isize && host[0] < host[1]) ? host[0] : host[1];
        double need = slow > 0 ? (double)(kCap - slow) * (double)start / (double)slow * 1.25 : (double)kChunkMax;
        if (need > (double)kChunkMax) need = (double)kChunkMax;
        next = ((int64_t)need + kChunk - 1) / kChunk * kChunk;   // (a multiple of 4: launch_metrics wants aligned starts)
        if (next < kChunk) next = kChunk;
    }
    counts->n_isize = want_isize ? (host[0] < kCap ? host[0] : kCap) : 0;
    counts->sample_counter = host[1] < kCap ? host[1] : kCap;
    counts->counter_total = host[3];
    counts->n_contam = host[4];
    counts->records_scanned = host[5];
    if (counts->n_isize)
        BESST_HIP_TRY(hipMemcpyAsync(isize_out, c->sample_a.p, (size_t)counts->n_isize * 4, hipMemcpyDeviceToHost, c->stream));
    if (counts->n_contam)
        BESST_HIP_TRY(hipMemcpyAsync(contam_out, c->sample_b.p, (size_t)counts->n_contam * 4, hipMemcpyDeviceToHost, c->stream));
    BESST_HIP_TRY(hipStreamSynchronize(c->stream));
    return BESST_OK;
}

int besst_ctx_value_histogram(besst_ctx* c, const int32_t* values, int64_t n, int64_t n_bins, int64_t* hist_out,
                              int64_t* overflow) {
    BESST_REQUIRE(c && hist_out && overflow, "value_histogram: null pointer");
    BESST_REQUIRE(n >= 0 && n_bins > 0 && n_bins < ((int64_t)1 << 31), "value_histogram: size out of range");
    BESST_REQUIRE(n == 0 || values, "value_histogram: null values");
    int rc = use_device(c);
    if (rc) return rc;
    if ((rc = c->sample_a.ensure((size_t)(n > 0 ? n : 1)))) return rc;
    if ((rc = c->aux.ensure((size_t)(n_bins + 1) * 8))) return rc;
    auto* hist = reinterpret_cast<unsigned long long*>(c->aux.p);
    BESST_HIP_TRY(hipMemsetAsync(hist, 0, (size_t)(n_bins + 1) * 8, c->stream));
    if (n) BESST_HIP_TRY(hipMemcpyAsync(c->sample_a.p, values, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    if ((rc = launch_value_histogram(c->stream, c->sample_a.p, n, n_bins, hist, hist + n_bins))) return rc;
    BESST_HIP_TRY(hipMemcpyAsync(hist_out, hist, (size_t)n_bins * 8, hipMemcpyDeviceToHost, c->stream));
    BESST_HIP_TRY(hipMemcpyAsync(overflow, hist + n_bins, 8, hipMemcpyDeviceToHost, c->stream));
    BESST_HIP_TRY(hipStreamSynchronize(c->stream));
    return BESST_OK;
}

int besst_dev_gap_condition_table(void* stream, double mean, double sigma, double read_len, double contig_len,
                                  int32_t d_lower, int32_t n, double* out) {
    BESST_REQUIRE(n >= 0 && (n == 0 || out) && sigma > 0.0, "gap_condition_table: bad argument");
    return launch_gap_table(static_cast<hipStream_t>(stream), mean, sigma, read_len, contig_len, d_lower, n, out);
}

int besst_ctx_gap_condition_table(besst_ctx* c, double mean, double sigma, double read_len, double contig_len,
                                  int32_t d_lower, int32_t n, double* h_out) {
    BESST_REQUIRE(c && n >= 0 && (n == 0 || h_out) && sigma > 0.0, "gap_condition_table: bad argument");
    if (n == 0) return BESST_OK;
    int rc = use_device(c);
    if (rc) return rc;
    if ((rc = c->aux.ensure((size_t)n * sizeof(double)))) return rc;
    auto* d_out = reinterpret_cast<double*>(c->aux.p);
    if ((rc = launch_gap_table(c->stream, mean, sigma, read_len, contig_len, d_lower, n, d_out))) return rc;
    BESST_HIP_TRY(hipMemcpyAsync(h_out, d_out, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    BESST_HIP_TRY(hipStreamSynchronize(c->stream));
    return BESST_OK;
}

namespace {
// besst_ctx_score_edges / besst_ctx_score_edges_lognormal: upload the edge list, lay the scratch out, run, download.
int ctx_score_impl(besst_ctx* c, int64_t n_edges, const uint32_t* row, const uint8_t* swap, const int32_t* len1,
                   const int32_t* len2, double mean, double sigma, double read_len, const LogNormalArgs* ln, double* gap,
                   double* sd0, int32_t* ks_h, uint8_t* flags) {
    int rc;
    // scratch offsets for edges too large for the LDS sort
    std::vector<uint32_t> h_n((size_t)c->n_rows);
    BESST_HIP_TRY(hipMemcpyAsync(h_n.data(), c->row_n.p, (size_t)c->n_rows * 4, hipMemcpyDeviceToHost, c->stream));
    BESST_HIP_TRY(hipStreamSynchronize(c->stream));
    std::vector<unsigned long long> big_off((size_t)n_edges, 0ull);
    unsigned long long big_total = 0;
    for (int64_t e = 0; e < n_edges; ++e) {
        BESST_REQUIRE((int64_t)row[e] < c->n_rows, "score_edges: row index out of range");
        const uint32_t n = h_n[row[e]];
        BESST_REQUIRE(n >= 1, "score_edges: empty row");
        unsigned long long np = 1;
        while (np < n) np <<= 1;
        if (np > 8192) { big_off[(size_t)e] = big_total; big_total += 2 * np; }
    }
    const size_t m = (size_t)n_edges;
    const size_t in_bytes = align_up(m * 4, 256) + align_up(m, 256) + 2 * align_up(m * 4, 256);
    const size_t out_bytes = 2 * align_up(m * 8, 256) + align_up(m * 4, 256) + align_up(m, 256);
    const size_t ws_bytes = align_up(m * 8, 256) + align_up((size_t)big_total * 4 + 4, 256);
    if ((rc = c->aux.ensure(in_bytes + out_bytes + ws_bytes))) return rc;
    char* p = c->aux.p;
    auto take = [&p](size_t bytes) { char* q = p; p += align_up(bytes, 256); return q; };
    auto* d_row = reinterpret_cast<uint32_t*>(take(m * 4));
    auto* d_swap = reinterpret_cast<uint8_t*>(take(m));
    auto* d_len1 = reinterpret_cast<int32_t*>(take(m * 4));
    auto* d_len2 = reinterpret_cast<int32_t*>(take(m * 4));
    auto* d_gap = reinterpret_cast<double*>(take(m * 8));
    auto* d_sd0 = reinterpret_cast<double*>(take(m * 8));
    auto* d_ks = reinterpret_cast<int32_t*>(take(m * 4));
    auto* d_flags = reinterpret_cast<uint8_t*>(take(m));
    char* d_ws = p;
    BESST_HIP_TRY(hipMemcpyAsync(d_row, row, m * 4, hipMemcpyHostToDevice, c->stream));
    BESST_HIP_TRY(hipMemcpyAsync(d_swap, swap, m, hipMemcpyHostToDevice, c->stream));
    BESST_HIP_TRY(hipMemcpyAsync(d_len1, len1, m * 4, hipMemcpyHostToDevice, c->stream));
    BESST_HIP_TRY(hipMemcpyAsync(d_len2, len2, m * 4, hipMemcpyHostToDevice, c->stream));
    BESST_HIP_TRY(hipMemcpyAsync(d_ws, big_off.data(), m * 8, hipMemcpyHostToDevice, c->stream));
    ScoreArgs a;
    a.row = d_row; a.swap = d_swap; a.len1 = d_len1; a.len2 = d_len2;
    a.row_n = c->row_n.p; a.row_sum = c->row_sum.p; a.row_offset = c->row_offset.p;
    a.obs_lo = c->obs_lo.p; a.obs_hi = c->obs_hi.p;
    a.mean = mean; a.sigma = sigma; a.read_len = read_len;
    a.n_edges = n_edges;
    if (ln) rc = launch_score_lognormal(c->stream, a, *ln, d_gap, d_sd0, d_ks, d_flags, d_ws, ws_bytes);
    else rc = launch_score(c->stream, a, d_gap, d_sd0, d_ks, d_flags, d_ws, ws_bytes);
    if (rc) return rc;
    BESST_HIP_TRY(hipMemcpyAsync(gap, d_gap, m * 8, hipMemcpyDeviceToHost, c->stream));
    if (sd0) BESST_HIP_TRY(hipMemcpyAsync(sd0, d_sd0, m * 8, hipMemcpyDeviceToHost, c->stream));
    BESST_HIP_TRY(hipMemcpyAsync(ks_h, d_ks, m * 4, hipMemcpyDeviceToHost, c->stream));
    BESST_HIP_TRY(hipMemcpyAsync(flags, d_flags, m, hipMemcpyDeviceToHost, c->stream));
    BESST_HIP_TRY(hipStreamSynchronize(c->stream));
    return BESST_OK;
}
}  // namespace

int besst_ctx_score_edges(besst_ctx* c, int64_t n_edges, const uint32_t* row, const uint8_t* swap, const int32_t* len1,
                          const int32_t* len2, double mean, double sigma, double read_len, double* gap, double* sd0,
                          int32_t* ks_h, uint8_t* flags) {
    BESST_NEED_BUILT(c);
    BESST_REQUIRE(n_edges >= 0, "score_edges: negative edge count");
    if (n_edges == 0) return BESST_OK;
    BESST_REQUIRE(row && swap && len1 && len2 && gap && sd0 && ks_h && flags, "score_edges: null pointer");
    BESST_REQUIRE(sigma > 0.0, "score_edges: sigma must be positive");
    int rc = use_device(c);
    if (rc) return rc;
    return ctx_score_impl(c, n_edges, row, swap, len1, len2, mean, sigma, read_len, nullptr, gap, sd0, ks_h, flags);
}

int besst_ctx_score_edges_lognormal(besst_ctx* c, int64_t n_edges, const uint32_t* row, const uint8_t* swap,
                                    const int32_t* len1, const int32_t* len2, double mean, double sigma, double read_len,
                                    double ln_mu, double ln_sigma, int64_t x_max, int32_t max_gap, double* gap,
                                    int32_t* ks_h, uint8_t* flags) {
    BESST_NEED_BUILT(c);
    BESST_REQUIRE(n_edges >= 0, "score_edges_lognormal: negative edge count");
    if (n_edges == 0) return BESST_OK;
    BESST_REQUIRE(row && swap && len1 && len2 && gap && ks_h && flags, "score_edges_lognormal: null pointer");
    BESST_REQUIRE(sigma > 0.0 && ln_sigma > 0.0, "score_edges_lognormal: sigma must be positive");
    BESST_REQUIRE(x_max >= 1 && x_max < ((int64_t)1 << 31) && max_gap >= 0, "score_edges_lognormal: parameters out of range");
    int rc = use_device(c);
    if (rc) return rc;
    if (!(c->ln_tables.p && c->ln_mu == ln_mu && c->ln_sigma == ln_sigma && c->ln_x_max == x_max)) {
        c->ln_x_max = 0;
        if ((rc = c->ln_tables.ensure(2 * (size_t)(x_max + 1)))) return rc;
        const size_t wsb = lognormal_tables_workspace_bytes(x_max);
        if ((rc = c->aux.ensure(wsb))) return rc;
        if ((rc = launch_lognormal_tables(c->stream, ln_mu, ln_sigma, x_max, c->ln_tables.p, c->ln_tables.p + (x_max + 1),
                                          c->aux.p, wsb)))
            return rc;
        BESST_HIP_TRY(hipStreamSynchronize(c->stream));          // aux is laid out anew below
        c->ln_mu = ln_mu; c->ln_sigma = ln_sigma; c->ln_x_max = x_max;
    }
    LogNormalArgs ln{ln_mu, ln_sigma, x_max, c->ln_tables.p, c->ln_tables.p + (x_max + 1), max_gap};
    return ctx_score_impl(c, n_edges, row, swap, len1, len2, mean, sigma, read_len, &ln, gap, nullptr, ks_h, flags);
}

int besst_ctx_conditional_stddevs(besst_ctx* c, const double* density, int64_t max_isize, const int32_t* steps,
                                  int32_t n_steps, double* h_out) {
    BESST_REQUIRE(c, "conditional_stddevs: null context");
    BESST_REQUIRE(n_steps >= 0 && max_isize >= 0, "conditional_stddevs: negative size");
    if (n_steps == 0) return BESST_OK;
    BESST_REQUIRE(density && steps && h_out, "conditional_stddevs: null pointer");
    int rc = use_device(c);
    if (rc) return rc;
    const size_t fb = align_up((size_t)(max_isize + 1) * 8, 256), sb = align_up((size_t)n_steps * 4, 256);
    if ((rc = c->aux.ensure(fb + sb + align_up((size_t)n_steps * 8, 256)))) return rc;
    auto* d_f = reinterpret_cast<double*>(c->aux.p);
    auto* d_steps = reinterpret_cast<int32_t*>(c->aux.p + fb);
    auto* d_out = reinterpret_cast<double*>(c->aux.p + fb + sb);
    BESST_HIP_TRY(hipMemcpyAsync(d_f, density, (size_t)(max_isize + 1) * 8, hipMemcpyHostToDevice, c->stream));
    BESST_HIP_TRY(hipMemcpyAsync(d_steps, steps, (size_t)n_steps * 4, hipMemcpyHostToDevice, c->stream));
    if ((rc = launch_conditional_stddevs(c->stream, d_f, max_isize, d_steps, n_steps, d_out))) return rc;
    BESST_HIP_TRY(hipMemcpyAsync(h_out, d_out, (size_t)n_steps * 8, hipMemcpyDeviceToHost, c->stream));
    BESST_HIP_TRY(hipStreamSynchronize(c->stream));
    return BESST_OK;
}

}  // extern "C"
