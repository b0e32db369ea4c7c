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
// BGZF blocks are independent deflate streams of <= 64 KiB.  The file is mapped; a batch of 4096 blocks is located by
// walking the block headers in the mapping (two cache lines per block, no copy) and inflated straight out of it by a
// persistent pool of threads (libdeflate when the shared object is present, zlib otherwise), each worker also walking
// the block it has just inflated as if it began with a record (htslib never lets a record straddle a block).  The
// sequential part is then one step per BLOCK - the speculative offsets of a block are the true ones where the walk
// arrives exactly at its first byte, records of other layouts are located one by one - and the columns are filled by
// the same pool, a block (or a run of loose records) per task.
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
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
typedef uint32_t (*ld_crc32_t)(uint32_t, const void*, size_t);
struct LibDeflate {
    ld_alloc_t alloc = nullptr;
    ld_decomp_t decompress = nullptr;
    ld_free_t free_ = nullptr;
    ld_crc32_t crc32_ = nullptr;
    LibDeflate() {
        void* h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        alloc = (ld_alloc_t)dlsym(h, "libdeflate_alloc_decompressor");
        decompress = (ld_decomp_t)dlsym(h, "libdeflate_deflate_decompress");
        free_ = (ld_free_t)dlsym(h, "libdeflate_free_decompressor");
        crc32_ = (ld_crc32_t)dlsym(h, "libdeflate_crc32");
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

constexpr size_t kBatchBlocks = 4096;      // <= 256 MiB of inflated bytes per batch: two pool dispatches per ~1.3 M records

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
    uint32_t crc;              // CRC-32 of the inflated bytes (gzip trailer)
};

// the block's CRC-32 as htslib checks it (bgzf.c): a payload that inflates to the right size with the wrong bytes is an error
uint32_t crc32_of(const uint8_t* p, size_t n) {
    if (libdeflate().crc32_) return libdeflate().crc32_(0, p, n);
    return (uint32_t)crc32(crc32(0L, Z_NULL, 0), p, (uInt)n);
}

uint32_t le32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
uint16_t le16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

}  // namespace

struct besst_bam {
    int fd = -1;
    const uint8_t* map = nullptr;    // the whole file
    const uint8_t* copy_map = nullptr;   // a second, plain mapping of it for bam_parallel_read (made on first use)
    size_t map_len = 0;
    size_t file_off = 0;             // first byte of the next BGZF block
    int n_threads = 1;
    std::vector<std::string> ref_names;
    std::vector<int32_t> ref_lengths;
    Bytes inflated;                  // undecoded tail + freshly inflated bytes
    size_t cursor = 0;               // next undecoded byte in `inflated`
    bool eof = false;
    std::string error;
    Pool* pool = nullptr;
    double t_read = 0, t_inflate = 0, t_walk = 0, t_decode = 0;   // seconds per phase (BESST_BAM_PROFILE=1 prints them)
    std::vector<void*> ld_ctx;       // one libdeflate decompressor per worker
    bool check_crc = true;           // every block's CRC-32 against its gzip trailer, as htslib does
    std::vector<BlockRecs> brecs;    // speculative per-block walks of the current batch, in stream order
    std::vector<size_t> brec_file_off;   // where each of those blocks begins in the file
    std::vector<uint32_t> blk_offs;
    size_t next_brec = 0;            // first block whose start is >= cursor
    int64_t n_clamped = 0;           // records whose aligned query length was saturated at 65535
    // the plan of one decode dispatch: whole blocks (their speculative offsets) and runs of loose records
    struct Seg { size_t block; size_t first, count; int64_t out; };   // block == SIZE_MAX: loose[first .. first + count)
    std::vector<Seg> plan;
    std::vector<size_t> loose;

    // Inflate the next batch of BGZF blocks and append to `inflated` (after dropping consumed bytes).
    bool fill(size_t want_blocks) {
        if (cursor > 0) {
            inflated.erase_front(cursor);
            cursor = 0;
        }
        brecs.clear();                // coordinates of the previous batch are gone
        brec_file_off.clear();
        next_brec = 0;
        if (eof) return true;
        const auto tp0 = std::chrono::steady_clock::now();
        std::vector<Block> blocks;
        blocks.reserve(want_blocks);
        size_t dst_total = inflated.size();
        for (size_t b = 0; b < want_blocks; ++b) {
            if (file_off == map_len) { eof = true; break; }
            const uint8_t* hdr = map + file_off;
            if (map_len - file_off < 18 || hdr[0] != 31 || hdr[1] != 139 || hdr[2] != 8 || !(hdr[3] & 4)) {
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
            if (bsize < 18 || map_len - file_off < bsize) { error = "truncated BGZF block"; return false; }
            const size_t rest = bsize - 18;
            const size_t extra_left = xlen - 6;
            if (rest < extra_left + 8) { error = "corrupt BGZF block"; return false; }
            const size_t payload = rest - extra_left - 8;
            const uint32_t isize = le32(hdr + bsize - 4);
            blocks.push_back(Block{file_off + 18 + extra_left, payload, dst_total, isize, le32(hdr + bsize - 8)});
            brec_file_off.push_back(file_off);
            dst_total += isize;
            file_off += bsize;
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
            if (!inflate_raw(map + k.src_off, k.src_len, inflated.data() + k.dst_off, k.dst_len,
                             ld_ctx.empty() ? nullptr : ld_ctx[(size_t)worker]) ||
                (check_crc && crc32_of(inflated.data() + k.dst_off, k.dst_len) != k.crc)) {
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
        if (!ok.load()) { error = "inflate failed (corrupt BGZF payload or CRC-32 mismatch)"; return false; }
        return true;
    }

    // make at least n undecoded bytes available; false at clean EOF or on error
    size_t batch_blocks = 16;        // blocks per fill(): a few while the header is parsed (a device ingest inflates nothing
                                     // else on the host), kBatchBlocks from the first record on
    bool need(size_t n) {
        while (inflated.size() - cursor < n) {
            if (eof) return false;
            if (!fill(batch_blocks)) return false;
        }
        return true;
    }
};

extern "C" {

besst_bam* besst_bam_open(const char* path, int n_threads) {
    if (!path) { besst::set_error("bam_open: null path"); return nullptr; }
    const int fd = open(path, O_RDONLY);
    if (fd < 0) { besst::set_error("bam_open: cannot open %s", path); return nullptr; }
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) { close(fd); besst::set_error("bam_open: %s is not a regular file", path); return nullptr; }
    besst_bam* b = new besst_bam();
    b->fd = fd;
    b->map_len = (size_t)st.st_size;
    if (b->map_len) {
        void* m = mmap(nullptr, b->map_len, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) { close(fd); delete b; besst::set_error("bam_open: cannot map %s", path); return nullptr; }
        (void)madvise(m, b->map_len, MADV_SEQUENTIAL);
        b->map = static_cast<const uint8_t*>(m);
    }
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
    b->batch_blocks = kBatchBlocks;
    return b;
}

void besst_bam_close(besst_bam* b) {
    if (!b) return;
    if (b->map) munmap(const_cast<uint8_t*>(b->map), b->map_len);
    if (b->copy_map) {
        // Tearing down the staging mapping's page-table entries marks every page accessed on the way (0.12 s of one thread for
        // a 5.6 GB file; from the copying threads - MADV_DONTNEED after each piece - it cost the first pass 0.28 s of staging).
        // A thread of its own does it, nobody waits for it - and piece by piece with MADV_DONTNEED, which holds the address
        // space's lock shared: one munmap of the whole mapping held it exclusively for those 0.12 s, and the caller's next
        // mmap / munmap (the reader's own mapping two lines further down, the allocator) waited behind it.
        char* m = reinterpret_cast<char*>(const_cast<uint8_t*>(b->copy_map));
        const size_t len = b->map_len;
        std::thread([m, len] {
            constexpr size_t kPiece = (size_t)4 << 20;          // (short holds of the address space's lock: the caller's allocations go on beside it)
            for (size_t at = 0; at < len; at += kPiece) (void)madvise(m + at, len - at < kPiece ? len - at : kPiece, MADV_DONTNEED);
            (void)munmap(m, len);
        }).detach();
    }
    if (b->fd >= 0) close(b->fd);
    delete b->pool;
    for (void* c : b->ld_ctx) libdeflate().free_(c);
    delete b;
}

int64_t besst_bam_clamped_records(const besst_bam* b) { return b ? b->n_clamped : -1; }

int64_t besst_bam_n_references(const besst_bam* b) { return b ? (int64_t)b->ref_names.size() : -1; }

const char* besst_bam_reference_name(const besst_bam* b, int64_t i) {
    return (b && i >= 0 && (size_t)i < b->ref_names.size()) ? b->ref_names[(size_t)i].c_str() : "";
}

int64_t besst_bam_reference_names(const besst_bam* b, char* buf, int64_t cap) {
    if (!b) return -1;
    int64_t need = 0;
    for (const std::string& n : b->ref_names) need += (int64_t)n.size() + 1;
    if (buf && cap >= need) {
        char* p = buf;
        for (const std::string& n : b->ref_names) {
            memcpy(p, n.c_str(), n.size() + 1);
            p += n.size() + 1;
        }
    }
    return need;
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
        // ---- sequential part: one step per block (or per loose record) up to the call's quota
        if (!b->need(4)) break;
        const auto tw0 = std::chrono::steady_clock::now();
        b->plan.clear();
        b->loose.clear();
        size_t cur = b->cursor;
        int64_t planned = 0;
        bool bad = false;
        for (;;) {
            const size_t quota = (size_t)(max_records - n - planned);
            if (quota == 0) break;
            while (b->next_brec < b->brecs.size() && b->brecs[b->next_brec].start < cur) ++b->next_brec;
            if (b->next_brec < b->brecs.size() && b->brecs[b->next_brec].start == cur && b->brecs[b->next_brec].count) {
                // the walk arrived exactly at a block's first byte: its speculative offsets are the true ones
                const BlockRecs& br = b->brecs[b->next_brec];
                const uint32_t* offs = b->blk_offs.data() + b->next_brec * kMaxBlockRecs;
                const size_t take = br.count < quota ? br.count : quota;
                b->plan.push_back(besst_bam::Seg{b->next_brec, 0, take, n + planned});
                planned += (int64_t)take;
                cur = take == br.count ? br.stop : br.start + offs[take];
                continue;
            }
            if (b->inflated.size() - cur < 4) break;
            const uint32_t block_size = le32(b->inflated.data() + cur);
            if (block_size < 32) { bad = true; break; }
            if (b->inflated.size() - cur < 4 + (size_t)block_size) break;
            if (!b->plan.empty() && b->plan.back().block == SIZE_MAX && b->plan.back().count < 1024)
                ++b->plan.back().count;                      // (runs of up to 1024 loose records per task)
            else
                b->plan.push_back(besst_bam::Seg{SIZE_MAX, b->loose.size(), 1, n + planned});
            b->loose.push_back(cur);
            ++planned;
            cur += 4 + (size_t)block_size;
        }
        if (bad) { besst::set_error("bam_read_records: corrupt record"); return -BESST_ERR_ARG; }
        if (planned == 0) {
            // the next record straddles the batch: pull more blocks (need() fails at a truncated file)
            const uint32_t block_size = le32(b->inflated.data() + b->cursor);
            if (!b->need(4 + (size_t)block_size)) {
                besst::set_error("bam_read_records: truncated record%s%s", b->error.empty() ? "" : ": ", b->error.c_str());
                return -BESST_ERR_ARG;
            }
            continue;
        }
        // ---- parallel decode, a block (or a run of loose records) per task
        const auto tw1 = std::chrono::steady_clock::now();
        std::atomic<bool> corrupt(false);
        std::atomic<int64_t> clamped(0);
        const uint8_t* base = b->inflated.data();
        auto decode = [&](const uint8_t* rec, size_t o) {
            const uint8_t* r = rec + 4;
            const uint32_t block_size = le32(rec);
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
            for (uint32_t c = n_cigar; c-- > 1;) {          // (pysam's getQueryEnd never looks at the first operation)
                const uint32_t v = le32(cg + 4 * c);
                const uint32_t op = v & 15u, len = v >> 4;
                if (op == 4) trail += len;
                else if (op != 5) break;
            }
            int64_t q_aln = (l_seq ? (int64_t)l_seq : q_total) - lead - trail;
            if (q_aln < 0) q_aln = 0;                        // a CIGAR of clips only: pysam gives a negative length
            if (q_aln > 65535) {                             // the qlen column is 16 bits wide, like RecordBatch's: saturate
                q_aln = 65535;                               // and count (besst_bam_clamped_records) - paired short reads
                clamped.fetch_add(1, std::memory_order_relaxed);   // never get near, one long alignment must not
            }                                                // make the file unreadable
            qlen[o] = (uint16_t)q_aln;
            rlen[o] = (int32_t)l_seq;
            alen[o] = (int32_t)ref_len;
        };
        b->pool->parallel_for(b->plan.size(), [&](size_t task, int) {
            const besst_bam::Seg& sg = b->plan[task];
            if (sg.block == SIZE_MAX) {
                for (size_t i = 0; i < sg.count; ++i) decode(base + b->loose[sg.first + i], (size_t)sg.out + i);
            } else {
                const BlockRecs& br = b->brecs[sg.block];
                const uint32_t* offs = b->blk_offs.data() + sg.block * kMaxBlockRecs;
                for (size_t i = 0; i < sg.count; ++i) decode(base + br.start + offs[sg.first + i], (size_t)sg.out + i);
            }
        });
        if (corrupt.load()) { besst::set_error("bam_read_records: corrupt record"); return -BESST_ERR_ARG; }
        b->n_clamped += clamped.load();
        b->cursor = cur;
        n += planned;
        const auto tw2 = std::chrono::steady_clock::now();
        b->t_walk += std::chrono::duration<double>(tw1 - tw0).count();
        b->t_decode += std::chrono::duration<double>(tw2 - tw1).count();
    }
    if (const char* e = getenv("BESST_BAM_PROFILE"); e && atoi(e))
        fprintf(stderr, "[bam] read %.3f s  inflate %.3f s  walk %.3f s  decode %.3f s (cumulative, %d threads)\n",
                b->t_read, b->t_inflate, b->t_walk, b->t_decode, b->n_threads);
    if (n == 0 && !b->error.empty()) {
        besst::set_error("bam_read_records: %s", b->error.c_str());
        return -BESST_ERR_ARG;
    }
    return n;
}

/* ---------------------------------------------------------------------------------------------------------------
 * Test / bench scaffolding: write record columns as a BAM file with htslib's block layout (a block is flushed before a
 * record that would not fit, so every block starts with a record).  Per record: name "r<index>", CIGAR [clip S] qlen M
 * with clip = rlen - qlen, rlen bases and qualities - qlen / rlen / alen round-trip through besst_bam_read_records.
 * Blocks are built and deflated by n_threads workers.  level: zlib's 0..9, + 16 for sequencer-like bases and qualities
 * (see below) instead of constant bytes.  Nothing in the graph path calls this.
 * --------------------------------------------------------------------------------------------------------------- */
int besst_bam_write_records(const char* path, int64_t n_ref, const char* const* ref_names, const int32_t* ref_lengths,
                            int64_t n, const int32_t* tid, const int32_t* mtid, const int32_t* pos, const int32_t* mpos,
                            const int32_t* tlen, const uint16_t* flag, const uint8_t* mapq, const uint16_t* qlen,
                            const int32_t* rlen, int n_threads, int level) {
    BESST_REQUIRE(path && n_ref >= 0 && n >= 0 && (n_ref == 0 || (ref_names && ref_lengths)), "bam_write_records: bad argument");
    BESST_REQUIRE(n == 0 || (tid && mtid && pos && mpos && tlen && flag && mapq && qlen && rlen), "bam_write_records: null column");
    FILE* fp = fopen(path, "wb");
    if (!fp) { besst::set_error("bam_write_records: cannot create %s", path); return BESST_ERR_ARG; }
    auto put32 = [](std::vector<uint8_t>& v, uint32_t x) { for (int k = 0; k < 4; ++k) v.push_back((uint8_t)(x >> (8 * k))); };
    auto bgzf = [&](const uint8_t* data, size_t len, std::vector<uint8_t>& out) {
        z_stream zs;
        memset(&zs, 0, sizeof(zs));
        if (deflateInit2(&zs, level & 15, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
        std::vector<uint8_t> payload(deflateBound(&zs, (uLong)len) + 16);
        zs.next_in = const_cast<Bytef*>(data);
        zs.avail_in = (uInt)len;
        zs.next_out = payload.data();
        zs.avail_out = (uInt)payload.size();
        const int rc = deflate(&zs, Z_FINISH);
        const size_t plen = zs.total_out;
        deflateEnd(&zs);
        if (rc != Z_STREAM_END || plen + 26 > 65536) return false;
        const uint32_t bsize = (uint32_t)plen + 25;
        const uint8_t head[18] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0, (uint8_t)bsize, (uint8_t)(bsize >> 8)};
        out.assign(head, head + 18);
        out.insert(out.end(), payload.begin(), payload.begin() + (long)plen);
        const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), data, (uInt)len);
        for (int k = 0; k < 4; ++k) out.push_back((uint8_t)(crc >> (8 * k)));
        for (int k = 0; k < 4; ++k) out.push_back((uint8_t)((uint32_t)len >> (8 * k)));
        return true;
    };
    // header: blocks of its own
    std::vector<uint8_t> hdr;
    const char text[] = "@HD\tVN:1.0\tSO:coordinate\n";
    hdr.insert(hdr.end(), {'B', 'A', 'M', 1});
    put32(hdr, (uint32_t)(sizeof(text) - 1));
    hdr.insert(hdr.end(), text, text + sizeof(text) - 1);
    put32(hdr, (uint32_t)n_ref);
    for (int64_t r = 0; r < n_ref; ++r) {
        const size_t l = strlen(ref_names[r]) + 1;
        put32(hdr, (uint32_t)l);
        hdr.insert(hdr.end(), ref_names[r], ref_names[r] + l);
        put32(hdr, (uint32_t)ref_lengths[r]);
    }
    bool ok = true;
    std::vector<uint8_t> blk;
    for (size_t off = 0; off < hdr.size() && ok; off += 60000) {
        const size_t len = hdr.size() - off < 60000 ? hdr.size() - off : 60000;
        ok = bgzf(hdr.data() + off, len, blk) && fwrite(blk.data(), 1, blk.size(), fp) == blk.size();
    }
    // records: sizes -> block boundaries (sequential, arithmetic only) -> blocks built and deflated in parallel, in
    // groups that are written out in order
    auto name_len = [](int64_t i) { size_t l = 3; for (int64_t v = i; v >= 10; v /= 10) ++l; return l; };   // 'r' digits NUL
    auto rec_bytes = [&](int64_t i) {
        const size_t q = qlen[i], sl = (size_t)(rlen[i] > 0 ? rlen[i] : 0);
        const size_t clip = sl > q ? sl - q : 0;
        return 4 + 32 + name_len(i) + 4 * ((clip ? 1 : 0) + (q ? 1 : 0)) + (sl + 1) / 2 + sl;
    };
    std::vector<int64_t> first;                              // first record of every block
    {
        size_t fill = 0;
        for (int64_t i = 0; i < n; ++i) {
            const size_t b = rec_bytes(i);
            if (first.empty() || fill + b > 60000) { first.push_back(i); fill = 0; }
            fill += b;
        }
        first.push_back(n);
    }
    const size_t n_blocks = first.size() - 1;
    Pool pool(n_threads > 0 ? n_threads : 1);
    const size_t group = 4096;
    std::vector<std::vector<uint8_t>> outs(group);
    std::atomic<bool> good(true);
    for (size_t g0 = 0; g0 < n_blocks && ok; g0 += group) {
        const size_t g1 = g0 + group < n_blocks ? g0 + group : n_blocks;
        pool.parallel_for(g1 - g0, [&](size_t k, int) {
            std::vector<uint8_t> raw;
            raw.reserve(61000);
            for (int64_t i = first[g0 + k]; i < first[g0 + k + 1]; ++i) {
                const uint32_t q = qlen[i], sl = (uint32_t)(rlen[i] > 0 ? rlen[i] : 0);
                const uint32_t clip = sl > q ? sl - q : 0;
                const uint32_t n_cig = (clip ? 1u : 0u) + (q ? 1u : 0u);
                char nm[24];
                const int nl = snprintf(nm, sizeof(nm), "r%lld", (long long)i) + 1;
                put32(raw, (uint32_t)(rec_bytes(i) - 4));
                put32(raw, (uint32_t)tid[i]);
                put32(raw, (uint32_t)pos[i]);
                raw.push_back((uint8_t)nl);
                raw.push_back(mapq[i]);
                raw.push_back(0x48); raw.push_back(0x12);    // bin 4680
                raw.push_back((uint8_t)n_cig); raw.push_back(0);
                raw.push_back((uint8_t)flag[i]); raw.push_back((uint8_t)(flag[i] >> 8));
                put32(raw, sl);
                put32(raw, (uint32_t)mtid[i]);
                put32(raw, (uint32_t)mpos[i]);
                put32(raw, (uint32_t)tlen[i]);
                raw.insert(raw.end(), nm, nm + nl);
                if (clip) put32(raw, (clip << 4) | 4u);
                if (q) put32(raw, (q << 4) | 0u);
                if (!(level & 16)) {
                    raw.insert(raw.end(), (sl + 1) / 2, (uint8_t)0x11);
                    raw.insert(raw.end(), sl, (uint8_t)0xff);
                } else {
                    // level | 16: bases and qualities that compress like a sequencer's (pseudo-random bases: 4 bits each;
                    // qualities from a few values that change slowly along the read) instead of constant bytes
                    uint64_t x = 0x9e3779b97f4a7c15ull * (uint64_t)(i + 1);
                    auto next = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
                    static const uint8_t kBase[4] = {1, 2, 4, 8};
                    for (uint32_t k = 0; k < (sl + 1) / 2; ++k) {
                        const uint64_t r = next();
                        raw.push_back((uint8_t)((kBase[r & 3] << 4) | kBase[(r >> 2) & 3]));
                    }
                    static const uint8_t kQual[8] = {37, 37, 37, 32, 37, 25, 37, 11};
                    uint8_t q8 = 37;
                    for (uint32_t k = 0; k < sl; ++k) {
                        const uint64_t r = next();
                        if ((r & 7) == 0) q8 = kQual[(r >> 3) & 7];
                        raw.push_back(q8);
                    }
                }
            }
            if (!bgzf(raw.data(), raw.size(), outs[k])) good = false;
        });
        if (!good.load()) { ok = false; break; }
        for (size_t k = 0; k < g1 - g0 && ok; ++k) ok = fwrite(outs[k].data(), 1, outs[k].size(), fp) == outs[k].size();
    }
    if (ok) ok = bgzf(nullptr, 0, blk) && fwrite(blk.data(), 1, blk.size(), fp) == blk.size();    // the EOF marker block
    if (fclose(fp) != 0) ok = false;
    if (!ok) { besst::set_error("bam_write_records: writing %s failed", path); return BESST_ERR_ARG; }
    return BESST_OK;
}

}  // extern "C"

namespace besst {
// for besst_ctx_push_bam (api.hip): how far the reader is through its file
int64_t bam_file_bytes(besst_bam* b) { return b ? (int64_t)b->map_len : 0; }
int64_t bam_file_position(besst_bam* b) { return b ? (int64_t)b->file_off : 0; }

// ---- for the device ingest (besst_ctx_push_bam_device)
const uint8_t* bam_file_map(besst_bam* b) { return b ? b->map : nullptr; }

// Where the next unread record lies: the file offset of its BGZF block and its offset in that block's inflated bytes.
// false when the reader holds bytes of a batch that is gone (a record that straddled two batches): not a position a
// block-wise reader can start from.
bool bam_record_position(besst_bam* b, int64_t* block_file_off, uint32_t* in_block_off) {
    if (!b) return false;
    for (size_t i = 0; i < b->brecs.size(); ++i) {
        if (b->brecs[i].end <= b->cursor) continue;
        if (b->brecs[i].start > b->cursor) return false;
        *block_file_off = (int64_t)b->brec_file_off[i];
        *in_block_off = (uint32_t)(b->cursor - b->brecs[i].start);
        return true;
    }
    if (b->cursor != b->inflated.size()) return false;
    *block_file_off = (int64_t)b->file_off;
    *in_block_off = 0;
    return true;
}

// `bytes` of the file from file_off into dst (pinned staging), by the pool's threads in 1 MiB pieces.  The bytes are COPIED
// OFF A MAPPING of the file, not pread: on the first pass over a file that has just been written (page cache or tmpfs) 32
// threads pread 19 - 35 GB/s and copy 80 - 155 GB/s off a mapping on the GPU box (tools/probe_pread.cpp: every page's second
// touch moves it to the active list under the LRU lock in the read path, a mapped page is not marked at all), and later
// passes are no slower either.  The mapping is a plain shared one of its own - the reader's carries MADV_SEQUENTIAL for the
// host form's walk - and its faults are spread over the threads (the single-threaded header walk of round 3, which touched
// every page from one thread, is what made a mapping look slow then).  BESST_STAGE_PREAD=1, or no mapping: pread.
bool bam_parallel_read(besst_bam* b, void* dst, int64_t file_off, size_t bytes) {
    if (!b || b->fd < 0) return false;
    static const bool use_pread = [] { const char* e = getenv("BESST_STAGE_PREAD"); return e && atoi(e) != 0; }();
    if (!use_pread && !b->copy_map && b->map_len) {
        void* m = mmap(nullptr, b->map_len, PROT_READ, MAP_SHARED, b->fd, 0);
        if (m != MAP_FAILED) b->copy_map = static_cast<const uint8_t*>(m);
    }
    const uint8_t* from = use_pread ? nullptr : b->copy_map;
    if (from && ((size_t)file_off > b->map_len || bytes > b->map_len - (size_t)file_off)) return false;
    constexpr size_t kPiece = (size_t)1 << 20;
    const size_t pieces = (bytes + kPiece - 1) / kPiece;
    std::atomic<bool> ok(true);
    auto piece = [&](size_t i, int) {
        size_t o = i * kPiece;
        const size_t end = o + kPiece < bytes ? o + kPiece : bytes;
        if (from) {
            const uint8_t* src = from + (size_t)file_off + o;
            memcpy(static_cast<char*>(dst) + o, src, end - o);
            return;
        }
        while (o < end) {
            const ssize_t got = pread(b->fd, static_cast<char*>(dst) + o, end - o, (off_t)(file_off + (int64_t)o));
            if (got <= 0) { ok = false; return; }
            o += (size_t)got;
        }
    };
    if (!b->pool || pieces <= 1) {
        for (size_t i = 0; i < pieces; ++i) piece(i, 0);
    } else {
        b->pool->parallel_for(pieces, piece);
    }
    return ok.load();
}

void bam_mark_consumed(besst_bam* b, int64_t saturated_qlen) {
    if (!b) return;
    b->n_clamped += saturated_qlen;
    b->file_off = b->map_len;
    b->eof = true;
    b->inflated.clear();
    b->cursor = 0;
    b->brecs.clear();
    b->brec_file_off.clear();
    b->next_brec = 0;
}
}  // namespace besst
