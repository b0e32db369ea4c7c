// Host-side BAM front-end: BGZF inflate + record decode straight into the SoA columns the kernels consume.
// (SURVEY.md section 8(f) rank 1 - the replacement for `pysam.Samfile` iteration, runBESST:162,
// CreateGraph.py:111, libmetrics.py:63,257,293.  No HIP code here; it lives in libbesst_amd.so so that one
// ctypes binding serves the whole path.)
//
// Columns follow the pysam-0.8 attributes the reference reads:
//   tid = refID, mtid = next_refID, pos, mpos = next_pos, tlen, flag, mapq,
//   qlen = query_alignment_length  (CIGAR M/I/=/X; soft clips excluded)
//   rlen = query_length            (l_seq; 0 when the sequence is absent)
//   alen = reference_length        (CIGAR M/D/N/=/X; 0 when there is no CIGAR)
// Supplementary / secondary records are passed through unfiltered, like the reference does.
//
// BGZF blocks are independent deflate streams of <= 64 KiB: a batch of 1024 blocks is read sequentially and
// inflated by a persistent pool of threads (libdeflate when the shared object is present, zlib otherwise).  The
// records - which may straddle block boundaries - are then located by one cheap sequential walk over their
// length prefixes and decoded into the columns by the same pool, each thread a contiguous range of records.
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "common.h"

namespace {

// ---- optional libdeflate (header-less binding of its stable C API) ------------------------------------------------
typedef void* (*ld_alloc_t)(void);
typedef int (*ld_decomp_t)(void*, const void*, size_t, void*, size_t, size_t*);
typedef void (*ld_free_t)(void*);
struct LibDeflate {
    ld_alloc_t alloc = nullptr;
    ld_decomp_t decompress = nullptr;
    ld_free_t free_ = nullptr;
    LibDeflate() {
        void* h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        alloc = (ld_alloc_t)dlsym(h, "libdeflate_alloc_decompressor");
        decompress = (ld_decomp_t)dlsym(h, "libdeflate_deflate_decompress");
        free_ = (ld_free_t)dlsym(h, "libdeflate_free_decompressor");
        if (!alloc || !decompress || !free_) alloc = nullptr;
    }
    bool ok() const { return alloc != nullptr; }
};
const LibDeflate& libdeflate() {
    static LibDeflate ld;
    return ld;
}

bool inflate_raw(const uint8_t* src, size_t n_src, uint8_t* dst, size_t n_dst, void* ld_ctx) {
    if (ld_ctx) {
        size_t got = 0;
        return libdeflate().decompress(ld_ctx, src, n_src, dst, n_dst, &got) == 0 && got == n_dst;
    }
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, -15) != Z_OK) return false;
    zs.next_in = const_cast<Bytef*>(src);
    zs.avail_in = (uInt)n_src;
    zs.next_out = dst;
    zs.avail_out = (uInt)n_dst;
    const int rc = inflate(&zs, Z_FINISH);
    const bool ok = rc == Z_STREAM_END && zs.total_out == n_dst;
    inflateEnd(&zs);
    return ok;
}

// Persistent workers: parallel_for(n, fn) runs fn(i, worker) for i in [0, n), indexes handed out dynamically; the
// calling thread takes part as worker 0.
class Pool {
  public:
    explicit Pool(int n_threads) : n_(n_threads < 1 ? 1 : n_threads) {
        for (int t = 1; t < n_; ++t) threads_.emplace_back([this, t] { loop(t); });
    }
    ~Pool() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
            ++epoch_;
        }
        cv_.notify_all();
        for (auto& th : threads_) th.join();
    }
    int size() const { return n_; }
    void parallel_for(size_t n, const std::function<void(size_t, int)>& fn) {
        if (n == 0) return;
        if (n_ == 1 || n == 1) {
            for (size_t i = 0; i < n; ++i) fn(i, 0);
            return;
        }
        {
            std::lock_guard<std::mutex> lk(m_);
            fn_ = &fn;
            total_ = n;
            next_.store(0);
            pending_ = n_ - 1;
            ++epoch_;
        }
        cv_.notify_all();
        drain(0);
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [this] { return pending_ == 0; });
        fn_ = nullptr;
    }

  private:
    void drain(int worker) {
        for (;;) {
            const size_t i = next_.fetch_add(1);
            if (i >= total_) break;
            (*fn_)(i, worker);
        }
    }
    void loop(int worker) {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return epoch_ != seen; });
                seen = epoch_;
                if (stop_) return;
            }
            drain(worker);
            {
                std::lock_guard<std::mutex> lk(m_);
                if (--pending_ == 0) done_.notify_one();
            }
        }
    }
    int n_;
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(size_t, int)>* fn_ = nullptr;
    std::atomic<size_t> next_{0};
    size_t total_ = 0;
    int pending_ = 0;
    uint64_t epoch_ = 0;
    bool stop_ = false;
};

constexpr size_t kBatchBlocks = 1024;      // <= 64 MiB of inflated bytes per batch

// Byte buffer whose resize() does not zero-fill (std::vector's value-initialisation of every 64 MiB batch was a
// third of the single-thread read time).
class Bytes {
  public:
    ~Bytes() { free(p_); }
    uint8_t* data() { return p_; }
    const uint8_t* data() const { return p_; }
    size_t size() const { return n_; }
    void clear() { n_ = 0; }
    void resize(size_t n) {
        if (n > cap_) {
            size_t c = cap_ ? cap_ : 4096;
            while (c < n) c *= 2;
            p_ = static_cast<uint8_t*>(realloc(p_, c));
            cap_ = c;
        }
        n_ = n;
    }
    void erase_front(size_t k) {
        if (k >= n_) { n_ = 0; return; }
        memmove(p_, p_ + k, n_ - k);
        n_ -= k;
    }

  private:
    uint8_t* p_ = nullptr;
    size_t n_ = 0, cap_ = 0;
};

constexpr size_t kMaxBlockRecs = 2048;     // a 64 KiB block holds < 65536 / 36 records

struct BlockRecs {
    size_t start, end;         // the block's bytes inside `inflated`
    size_t stop;               // where the block-local walk stopped (== end when no record straddles out of it)
    uint32_t count;            // records found; their offsets relative to `start` live in blk_offs[index * kMaxBlockRecs ...]
};

struct Block {
    size_t src_off, src_len;   // deflate payload inside the batch buffer
    size_t dst_off, dst_len;   // position inside the inflated buffer
};

uint32_t le32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
uint16_t le16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

}  // namespace

struct besst_bam {
    FILE* fp = nullptr;
    int n_threads = 1;
    std::vector<std::string> ref_names;
    std::vector<int32_t> ref_lengths;
    Bytes inflated;                  // undecoded tail + freshly inflated bytes
    size_t cursor = 0;               // next undecoded byte in `inflated`
    bool eof = false;
    Bytes raw;                       // compressed batch
    std::string error;
    Pool* pool = nullptr;
    double t_read = 0, t_inflate = 0, t_walk = 0, t_decode = 0;   // seconds per phase (BESST_BAM_PROFILE=1 prints them)
    std::vector<void*> ld_ctx;       // one libdeflate decompressor per worker
    std::vector<size_t> rec_off;     // offsets (into `inflated`) of the records located by the last walk
    std::vector<BlockRecs> brecs;    // speculative per-block walks of the current batch, in stream order
    std::vector<uint32_t> blk_offs;
    size_t next_brec = 0;            // first block whose start is >= cursor
    int64_t n_clamped = 0;           // records whose aligned query length was saturated at 65535

    // Inflate the next batch of BGZF blocks and append to `inflated` (after dropping consumed bytes).
    bool fill(size_t want_blocks) {
        if (cursor > 0) {
            inflated.erase_front(cursor);
            cursor = 0;
        }
        brecs.clear();                // coordinates of the previous batch are gone
        next_brec = 0;
        if (eof) return true;
        const auto tp0 = std::chrono::steady_clock::now();
        raw.clear();
        std::vector<Block> blocks;
        size_t dst_total = inflated.size();
        for (size_t b = 0; b < want_blocks; ++b) {
            uint8_t hdr[18];
            const size_t got = fread(hdr, 1, 18, fp);
            if (got == 0) { eof = true; break; }
            if (got != 18 || hdr[0] != 31 || hdr[1] != 139 || hdr[2] != 8 || !(hdr[3] & 4)) {
                error = "not a BGZF block (bad gzip header)";
                return false;
            }
            const uint32_t xlen = le16(hdr + 10);
            // the BC subfield holding BSIZE is the first extra field in every htslib-written file
            if (xlen < 6 || hdr[12] != 'B' || hdr[13] != 'C' || le16(hdr + 14) != 2) {
                error = "BGZF block without a leading BC extra field";
                return false;
            }
            const size_t bsize = (size_t)le16(hdr + 16) + 1;
            const size_t rest = bsize - 18;
            const size_t at = raw.size();
            raw.resize(at + rest);
            if (fread(raw.data() + at, 1, rest, fp) != rest) { error = "truncated BGZF block"; return false; }
            const size_t extra_left = xlen - 6;
            if (rest < extra_left + 8) { error = "corrupt BGZF block"; return false; }
            const size_t payload = rest - extra_left - 8;
            const uint32_t isize = le32(raw.data() + at + rest - 4);
            blocks.push_back(Block{at + extra_left, payload, dst_total, isize});
            dst_total += isize;
        }
        inflated.resize(dst_total);
        brecs.resize(blocks.size());
        if (blk_offs.size() < blocks.size() * kMaxBlockRecs) blk_offs.resize(blocks.size() * kMaxBlockRecs);
        const auto tp1 = std::chrono::steady_clock::now();
        std::atomic<bool> ok(true);
        pool->parallel_for(blocks.size(), [&](size_t bi, int worker) {
            const Block& k = blocks[bi];
            BlockRecs& br = brecs[bi];
            br.start = k.dst_off;
            br.end = k.dst_off + k.dst_len;
            br.stop = br.start;
            br.count = 0;
            if (k.dst_len == 0) return;           // the empty EOF marker block
            if (!inflate_raw(raw.data() + k.src_off, k.src_len, inflated.data() + k.dst_off, k.dst_len,
                             ld_ctx.empty() ? nullptr : ld_ctx[(size_t)worker])) {
                ok = false;
                return;
            }
            // speculative walk: valid iff a record starts at the block's first byte
            const uint8_t* base = inflated.data();
            uint32_t* offs = blk_offs.data() + bi * kMaxBlockRecs;
            size_t cur = br.start;
            while (br.end - cur >= 4 && br.count < kMaxBlockRecs) {
                const uint32_t block_size = le32(base + cur);
                if (block_size < 32 || br.end - cur < 4 + (size_t)block_size) break;
                offs[br.count++] = (uint32_t)(cur - br.start);
                cur += 4 + (size_t)block_size;
            }
            br.stop = cur;
        });
        const auto tp2 = std::chrono::steady_clock::now();
        t_read += std::chrono::duration<double>(tp1 - tp0).count();
        t_inflate += std::chrono::duration<double>(tp2 - tp1).count();
        if (!ok.load()) { error = "inflate failed (corrupt BGZF payload)"; return false; }
        return true;
    }

    // make at least n undecoded bytes available; false at clean EOF or on error
    bool need(size_t n) {
        while (inflated.size() - cursor < n) {
            if (eof) return false;
            if (!fill(kBatchBlocks)) return false;
        }
        return true;
    }
};

extern "C" {

besst_bam* besst_bam_open(const char* path, int n_threads) {
    if (!path) { besst::set_error("bam_open: null path"); return nullptr; }
    FILE* fp = fopen(path, "rb");
    if (!fp) { besst::set_error("bam_open: cannot open %s", path); return nullptr; }
    setvbuf(fp, nullptr, _IOFBF, 8 << 20);      // the block headers are read with 18-byte freads
    besst_bam* b = new besst_bam();
    b->fp = fp;
    b->n_threads = n_threads > 0 ? n_threads : 1;
    b->pool = new Pool(b->n_threads);
    if (libdeflate().ok())
        for (int t = 0; t < b->n_threads; ++t) b->ld_ctx.push_back(libdeflate().alloc());
    auto fail = [&](const char* msg) {
        besst::set_error("bam_open(%s): %s", path, b->error.empty() ? msg : b->error.c_str());
        besst_bam_close(b);
        return (besst_bam*)nullptr;
    };
    if (!b->need(12) || memcmp(b->inflated.data(), "BAM\1", 4) != 0) return fail("not a BAM file");
    const uint32_t l_text = le32(b->inflated.data() + 4);
    if (!b->need(12 + (size_t)l_text)) return fail("truncated header");
    b->cursor = 8 + l_text;
    if (!b->need(4)) return fail("truncated header");
    const uint32_t n_ref = le32(b->inflated.data() + b->cursor);
    b->cursor += 4;
    for (uint32_t r = 0; r < n_ref; ++r) {
        if (!b->need(4)) return fail("truncated reference table");
        const uint32_t l_name = le32(b->inflated.data() + b->cursor);
        if (!b->need(4 + (size_t)l_name + 4)) return fail("truncated reference table");
        const char* nm = reinterpret_cast<const char*>(b->inflated.data() + b->cursor + 4);
        b->ref_names.emplace_back(nm, l_name ? l_name - 1 : 0);
        b->ref_lengths.push_back((int32_t)le32(b->inflated.data() + b->cursor + 4 + l_name));
        b->cursor += 8 + l_name;
    }
    return b;
}

void besst_bam_close(besst_bam* b) {
    if (!b) return;
    if (b->fp) fclose(b->fp);
    delete b->pool;
    for (void* c : b->ld_ctx) libdeflate().free_(c);
    delete b;
}

int64_t besst_bam_clamped_records(const besst_bam* b) { return b ? b->n_clamped : -1; }

int64_t besst_bam_n_references(const besst_bam* b) { return b ? (int64_t)b->ref_names.size() : -1; }

const char* besst_bam_reference_name(const besst_bam* b, int64_t i) {
    return (b && i >= 0 && (size_t)i < b->ref_names.size()) ? b->ref_names[(size_t)i].c_str() : "";
}

int besst_bam_reference_lengths(const besst_bam* b, int32_t* out) {
    BESST_REQUIRE(b && out, "bam_reference_lengths: null pointer");
    for (size_t i = 0; i < b->ref_lengths.size(); ++i) out[i] = b->ref_lengths[i];
    return BESST_OK;
}

// Decode up to max_records alignment records into the columns; returns the number decoded (0 = end of file) or
// a negative status.  qlen saturates at 65535 (the device column is 16 bit; paired short reads never get near) and
// besst_bam_clamped_records counts the records it happened to.
int64_t besst_bam_read_records(besst_bam* b, int64_t max_records, int32_t* tid, int32_t* mtid, int32_t* pos,
                               int32_t* mpos, int32_t* tlen, uint16_t* flag, uint8_t* mapq, uint16_t* qlen,
                               int32_t* rlen, int32_t* alen) {
    if (!b || !tid || !mtid || !pos || !mpos || !tlen || !flag || !mapq || !qlen || !rlen || !alen) {
        besst::set_error("bam_read_records: null pointer");
        return -BESST_ERR_ARG;
    }
    int64_t n = 0;
    while (n < max_records) {
        // ---- sequential walk: locate the complete records available in the inflated bytes
        if (!b->need(4)) break;
        const auto tw0 = std::chrono::steady_clock::now();
        b->rec_off.clear();
        size_t cur = b->cursor;
        bool bad = false;
        for (;;) {
            const size_t quota = (size_t)(max_records - n) - b->rec_off.size();
            if (quota == 0) break;
            while (b->next_brec < b->brecs.size() && b->brecs[b->next_brec].start < cur) ++b->next_brec;
            if (b->next_brec < b->brecs.size() && b->brecs[b->next_brec].start == cur && b->brecs[b->next_brec].count) {
                // the walk arrived exactly at a block's first byte: its speculative offsets are the true ones
                const BlockRecs& br = b->brecs[b->next_brec];
                const uint32_t* offs = b->blk_offs.data() + b->next_brec * kMaxBlockRecs;
                const size_t take = br.count < quota ? br.count : quota;
                const size_t at = b->rec_off.size();
                b->rec_off.resize(at + take);
                for (size_t i = 0; i < take; ++i) b->rec_off[at + i] = br.start + offs[i];
                cur = take == br.count ? br.stop : br.start + offs[take];
                continue;
            }
            if (b->inflated.size() - cur < 4) break;
            const uint32_t block_size = le32(b->inflated.data() + cur);
            if (block_size < 32) { bad = true; break; }
            if (b->inflated.size() - cur < 4 + (size_t)block_size) break;
            b->rec_off.push_back(cur);
            cur += 4 + (size_t)block_size;
        }
        if (bad) { besst::set_error("bam_read_records: corrupt record"); return -BESST_ERR_ARG; }
        if (b->rec_off.empty()) {
            // the next record straddles the batch: pull more blocks (need() fails at a truncated file)
            const uint32_t block_size = le32(b->inflated.data() + b->cursor);
            if (!b->need(4 + (size_t)block_size)) {
                besst::set_error("bam_read_records: truncated record%s%s", b->error.empty() ? "" : ": ", b->error.c_str());
                return -BESST_ERR_ARG;
            }
            continue;
        }
        // ---- parallel decode, a contiguous range of records per task
        const auto tw1 = std::chrono::steady_clock::now();
        const size_t m = b->rec_off.size();
        const size_t n_tasks = m < 4096 ? 1 : (size_t)b->pool->size() * 4;
        std::atomic<bool> corrupt(false);
        std::atomic<int64_t> clamped(0);
        const uint8_t* base = b->inflated.data();
        const size_t* offs = b->rec_off.data();
        b->pool->parallel_for(n_tasks, [&](size_t task, int) {
            const size_t i0 = m * task / n_tasks, i1 = m * (task + 1) / n_tasks;
            for (size_t i = i0; i < i1; ++i) {
                const uint8_t* r = base + offs[i] + 4;
                const uint32_t block_size = le32(base + offs[i]);
                const size_t o = (size_t)n + i;
                tid[o] = (int32_t)le32(r);
                pos[o] = (int32_t)le32(r + 4);
                const uint32_t l_read_name = r[8];
                mapq[o] = r[9];
                const uint32_t n_cigar = le16(r + 12);
                flag[o] = le16(r + 14);
                const uint32_t l_seq = le32(r + 16);
                mtid[o] = (int32_t)le32(r + 20);
                mpos[o] = (int32_t)le32(r + 24);
                tlen[o] = (int32_t)le32(r + 28);
                if (32 + l_read_name + 4ull * n_cigar > block_size) { corrupt = true; return; }
                const uint8_t* cg = r + 32 + l_read_name;
                // pysam 0.8.4's AlignedRead properties, which is what the reference reads (CreateGraph.py:138 qlen;
                // libmetrics.py:258-262 rlen / alen):
                //   qlen = query_alignment_length = qend - qstart, qstart = the leading soft clips (hard clips in front
                //          of them skipped), qend = l_seq - or, for a record without sequence, the M/I/S/=/X total of
                //          the CIGAR - minus the trailing soft clips.  A record WITHOUT a CIGAR (BWA's unmapped read
                //          placed at its mate) therefore has qlen = l_seq, and the reference does add it to the
                //          coverage of the contig it is placed on (mapq 0 passes the test of CreateGraph.py:138-139).
                //   alen = reference_length = the M/D/N/=/X total (None -> 0 without a CIGAR)
                // The CG:B,I long-CIGAR convention postdates that pysam: the placeholder <l_seq>S<n>N is read as it stands.
                int64_t q_total = 0, ref_len = 0, lead = 0, trail = 0;
                bool in_lead = true;
                for (uint32_t c = 0; c < n_cigar; ++c) {
                    const uint32_t v = le32(cg + 4 * c);
                    const uint32_t op = v & 15u, len = v >> 4;
                    // M=0 I=1 D=2 N=3 S=4 H=5 P=6 '='=7 X=8
                    if (op == 0 || op == 1 || op == 4 || op == 7 || op == 8) q_total += len;
                    if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) ref_len += len;
                    if (in_lead) {
                        if (op == 4) lead += len;
                        else if (op != 5) in_lead = false;
                    }
                }
                for (uint32_t c = n_cigar; c-- > 1;) {      // (pysam's getQueryEnd never looks at the first operation)
                    const uint32_t v = le32(cg + 4 * c);
                    const uint32_t op = v & 15u, len = v >> 4;
                    if (op == 4) trail += len;
                    else if (op != 5) break;
                }
                int64_t q_aln = (l_seq ? (int64_t)l_seq : q_total) - lead - trail;
                if (q_aln < 0) q_aln = 0;                    // a CIGAR of clips only: pysam gives a negative length
                if (q_aln > 65535) {                         // the qlen column is 16 bits wide, like RecordBatch's: saturate
                    q_aln = 65535;                           // and count (besst_bam_clamped_records) - paired short reads
                    clamped.fetch_add(1, std::memory_order_relaxed);   // never get near, one long alignment must not
                }                                            // make the file unreadable
                qlen[o] = (uint16_t)q_aln;
                rlen[o] = (int32_t)l_seq;
                alen[o] = (int32_t)ref_len;
            }
        });
        if (corrupt.load()) { besst::set_error("bam_read_records: corrupt record"); return -BESST_ERR_ARG; }
        b->n_clamped += clamped.load();
        b->cursor = cur;
        n += (int64_t)m;
        const auto tw2 = std::chrono::steady_clock::now();
        b->t_walk += std::chrono::duration<double>(tw1 - tw0).count();
        b->t_decode += std::chrono::duration<double>(tw2 - tw1).count();
    }
    if (getenv("BESST_BAM_PROFILE"))
        fprintf(stderr, "[bam] read %.3f s  inflate %.3f s  walk %.3f s  decode %.3f s (cumulative, %d threads)\n",
                b->t_read, b->t_inflate, b->t_walk, b->t_decode, b->n_threads);
    if (n == 0 && !b->error.empty()) {
        besst::set_error("bam_read_records: %s", b->error.c_str());
        return -BESST_ERR_ARG;
    }
    return n;
}

}  // extern "C"
