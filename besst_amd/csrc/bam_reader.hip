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
// BGZF blocks are independent deflate streams of <= 64 KiB: a batch of blocks is read sequentially, inflated by
// a pool of threads (libdeflate when the shared object is present, zlib otherwise), and the records - which may
// straddle block boundaries - are decoded from the inflated stream in order.
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>
#include <zlib.h>

#include <atomic>
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
    std::vector<uint8_t> inflated;   // undecoded tail + freshly inflated bytes
    size_t cursor = 0;               // next undecoded byte in `inflated`
    bool eof = false;
    std::vector<uint8_t> raw;        // compressed batch
    std::string error;

    // Inflate the next batch of BGZF blocks and append to `inflated` (after dropping consumed bytes).
    bool fill(size_t want_blocks) {
        if (cursor > 0) {
            inflated.erase(inflated.begin(), inflated.begin() + (long)cursor);
            cursor = 0;
        }
        if (eof) return true;
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
        std::atomic<bool> ok(true);
        const int nt = n_threads < 1 ? 1 : n_threads;
        auto work = [&](int tid) {
            void* ctx = libdeflate().ok() ? libdeflate().alloc() : nullptr;
            for (size_t b = (size_t)tid; b < blocks.size(); b += (size_t)nt) {
                const Block& k = blocks[b];
                if (k.dst_len == 0) continue;     // the empty EOF marker block
                if (!inflate_raw(raw.data() + k.src_off, k.src_len, inflated.data() + k.dst_off, k.dst_len, ctx)) ok = false;
            }
            if (ctx) libdeflate().free_(ctx);
        };
        if (nt == 1 || blocks.size() < 4) {
            for (int t = 0; t < nt; ++t) work(t);
        } else {
            std::vector<std::thread> pool;
            for (int t = 0; t < nt; ++t) pool.emplace_back(work, t);
            for (auto& th : pool) th.join();
        }
        if (!ok.load()) { error = "inflate failed (corrupt BGZF payload)"; return false; }
        return true;
    }

    // make at least n undecoded bytes available; false at clean EOF or on error
    bool need(size_t n) {
        while (inflated.size() - cursor < n) {
            if (eof) return false;
            if (!fill(256)) return false;
        }
        return true;
    }
};

extern "C" {

besst_bam* besst_bam_open(const char* path, int n_threads) {
    if (!path) { besst::set_error("bam_open: null path"); return nullptr; }
    FILE* fp = fopen(path, "rb");
    if (!fp) { besst::set_error("bam_open: cannot open %s", path); return nullptr; }
    besst_bam* b = new besst_bam();
    b->fp = fp;
    b->n_threads = n_threads > 0 ? n_threads : 1;
    auto fail = [&](const char* msg) {
        besst::set_error("bam_open(%s): %s", path, b->error.empty() ? msg : b->error.c_str());
        fclose(fp);
        delete b;
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
    delete b;
}

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
// a negative status.  qlen is clamped to 65535 (the device column is 16 bit; paired short reads never get near).
int64_t besst_bam_read_records(besst_bam* b, int64_t max_records, int32_t* tid, int32_t* mtid, int32_t* pos,
                               int32_t* mpos, int32_t* tlen, uint16_t* flag, uint8_t* mapq, uint16_t* qlen,
                               int32_t* rlen, int32_t* alen) {
    if (!b || !tid || !mtid || !pos || !mpos || !tlen || !flag || !mapq || !qlen || !rlen || !alen) {
        besst::set_error("bam_read_records: null pointer");
        return -BESST_ERR_ARG;
    }
    int64_t n = 0;
    while (n < max_records) {
        if (!b->need(4)) break;
        const uint32_t block_size = le32(b->inflated.data() + b->cursor);
        if (block_size < 32) { besst::set_error("bam_read_records: corrupt record"); return -BESST_ERR_ARG; }
        if (!b->need(4 + (size_t)block_size)) {
            besst::set_error("bam_read_records: truncated record%s%s", b->error.empty() ? "" : ": ", b->error.c_str());
            return -BESST_ERR_ARG;
        }
        const uint8_t* r = b->inflated.data() + b->cursor + 4;
        tid[n] = (int32_t)le32(r);
        pos[n] = (int32_t)le32(r + 4);
        const uint32_t l_read_name = r[8];
        mapq[n] = r[9];
        const uint32_t n_cigar = le16(r + 12);
        flag[n] = le16(r + 14);
        const uint32_t l_seq = le32(r + 16);
        mtid[n] = (int32_t)le32(r + 20);
        mpos[n] = (int32_t)le32(r + 24);
        tlen[n] = (int32_t)le32(r + 28);
        if (32 + l_read_name + 4ull * n_cigar > block_size) { besst::set_error("bam_read_records: corrupt record"); return -BESST_ERR_ARG; }
        const uint8_t* cg = r + 32 + l_read_name;
        int64_t q_aln = 0, ref_len = 0;
        for (uint32_t c = 0; c < n_cigar; ++c) {
            const uint32_t v = le32(cg + 4 * c);
            const uint32_t op = v & 15u, len = v >> 4;
            // M=0 I=1 D=2 N=3 S=4 H=5 P=6 '='=7 X=8
            if (op == 0 || op == 1 || op == 7 || op == 8) q_aln += len;
            if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) ref_len += len;
        }
        qlen[n] = (uint16_t)(q_aln > 65535 ? 65535 : q_aln);
        rlen[n] = (int32_t)l_seq;
        alen[n] = (int32_t)ref_len;
        b->cursor += 4 + (size_t)block_size;
        ++n;
    }
    if (n == 0 && !b->error.empty()) {
        besst::set_error("bam_read_records: %s", b->error.c_str());
        return -BESST_ERR_ARG;
    }
    return n;
}

}  // extern "C"
