// svx_bam.cpp -- native BGZF/BAM ingestion: file -> packed structure of arrays (host side of libsvx.so).
//
// Replaces what the reference obtains record by record from pysam/htslib (AlignmentFile iteration at
// src/collection/run_collection.py:23-26, src/collection/collect_signatures.py:128-155).  The file is
// streamed: a chunk of whole BGZF blocks is read, inflated block-parallel (BGZF blocks are independent
// gzip members; zlib raw inflate on a thread pool), its records are chained once and their fixed fields,
// CIGAR words, names and (on request) 4-bit sequences are appended to growing arrays; the decompressed
// bytes of a chunk (mostly SEQ/QUAL, which the hot path never reads) are dropped before the next chunk is
// read, so the resident size is that of the packed arrays that go to the GPU unchanged (svx_cigar_scan
// input), not that of the file.  SAMv1 section 4 layouts, including CIGARs with more than 65535 operations
// (CG:B,I tag).  With two virtual offsets from the .bai index only that byte range is read.
#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/svx.h"

namespace {

inline uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
inline uint16_t rd16(const uint8_t* p) { uint16_t v; memcpy(&v, p, 2); return v; }

// append-only storage for the distinct read names (stable addresses for the hash map's keys)
struct Arena {
    static constexpr size_t SLAB = 1 << 20;
    std::vector<std::unique_ptr<char[]>> slabs;
    size_t used = SLAB;
    std::string_view keep(std::string_view s)
    {
        if (s.size() > SLAB) { slabs.emplace_back(new char[s.size()]); memcpy(slabs.back().get(), s.data(), s.size()); used = SLAB; return {slabs.back().get(), s.size()}; }
        if (used + s.size() > SLAB) { slabs.emplace_back(new char[SLAB]); used = 0; }
        char* at = slabs.back().get() + used;
        memcpy(at, s.data(), s.size());
        used += s.size();
        return {at, s.size()};
    }
};

struct Bam {
    std::string header_text;
    std::vector<std::string> ref_names;
    std::vector<int32_t> ref_lens;
    // one entry per record
    std::vector<int32_t> tid, pos, l_seq, name_id;
    std::vector<uint16_t> flag;
    std::vector<uint8_t> mapq;
    std::vector<uint64_t> cig_off{0};      // CSR over CIGAR words, n_rec + 1
    std::vector<uint32_t> cigar;
    std::vector<uint64_t> seq_off;         // byte offset of each record's 4-bit SEQ in `seq` (SVX_BAM_KEEP_SEQ)
    std::vector<uint8_t> seq;
    // distinct QNAMEs in order of first occurrence
    Arena arena;
    std::vector<std::string_view> names;
    std::unordered_map<std::string_view, int32_t> seen;
    uint64_t names_bytes = 0;
};

// CIGARs with more than 65535 operations live in the CG:B,I tag; the record then carries "<l_seq>S<ref_len>N"
// (SAMv1 4.2.2).  Returns the tag's uint32 array and its length, or NULL.
const uint8_t* find_long_cigar(const uint8_t* aux, const uint8_t* end, uint32_t* count)
{
    while (aux + 3 <= end) {
        const uint8_t t0 = aux[0], t1 = aux[1], ty = aux[2];
        const uint8_t* p = aux + 3;
        switch (ty) {
        case 'A': case 'c': case 'C': p += 1; break;
        case 's': case 'S': p += 2; break;
        case 'i': case 'I': case 'f': p += 4; break;
        case 'Z': case 'H': while (p < end && *p) ++p; ++p; break;
        case 'B': {
            if (p + 5 > end) return nullptr;
            const uint8_t sub = p[0];
            const uint32_t n = rd32(p + 1);
            const uint32_t width = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
            if (t0 == 'C' && t1 == 'G' && sub == 'I') {
                if (p + 5 + (uint64_t)n * 4 > end) return nullptr;
                *count = n;
                return p + 5;
            }
            p += 5 + (uint64_t)n * width;
            break;
        }
        default: return nullptr;
        }
        aux = p;
    }
    return nullptr;
}

// The inflated stream of the chunk in hand: grown without zero filling (a vector's resize would memset gigabytes that
// inflate overwrites right away) and reused from chunk to chunk.
struct RawBuf {
    std::unique_ptr<uint8_t[]> p;
    size_t n = 0, cap = 0;
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    const uint8_t& operator[](size_t i) const { return p[i]; }
    uint8_t* data() { return p.get(); }
    const uint8_t* data() const { return p.get(); }
    void resize(size_t want)
    {
        if (want > cap) {
            const size_t grown = std::max(want, cap + cap / 2);
            std::unique_ptr<uint8_t[]> q(new uint8_t[grown]);
            if (n) memcpy(q.get(), p.get(), n);
            p = std::move(q);
            cap = grown;
        }
        n = want;
    }
    void drop_front(size_t k) { if (k) { memmove(p.get(), p.get() + k, n - k); n -= k; } }
    void clear() { n = 0; }
};

// SVX_TIMING=1: seconds spent reading, inflating, chaining records (serial) and scattering fields (parallel), on stderr
struct BamClock {
    double read = 0, inflate = 0, chain = 0, scatter = 0;
    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
};
BamClock g_clock;

// Raw-deflate decoder of one BGZF block.  libdeflate (whole-buffer decoder, 2-3x zlib's rate on BGZF blocks) when
// libdeflate.so.0 is on the machine -- bound with dlopen: the image ships the library without its header -- else zlib.
struct Deflate {
    using alloc_fn = void* (*)();
    using free_fn = void (*)(void*);
    using run_fn = int (*)(void*, const void*, size_t, void*, size_t, size_t*);
    using crc_fn = uint32_t (*)(uint32_t, const void*, size_t);
    alloc_fn alloc = nullptr;
    free_fn release = nullptr;
    run_fn run = nullptr;
    crc_fn crc = nullptr;                                     // libdeflate_crc32 (carry-less multiply: ~10 GB/s per core; zlib's table walk: ~1)
    Deflate()
    {
        if (getenv("SVX_BAM_ZLIB")) return;                  // A/B switch: force zlib
        void* h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("libdeflate.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        alloc = (alloc_fn)dlsym(h, "libdeflate_alloc_decompressor");
        release = (free_fn)dlsym(h, "libdeflate_free_decompressor");
        run = (run_fn)dlsym(h, "libdeflate_deflate_decompress");
        crc = (crc_fn)dlsym(h, "libdeflate_crc32");
        if (!alloc || !release || !run) alloc = nullptr;
    }
    bool fast() const { return alloc != nullptr; }
};
const Deflate& deflate_lib() { static const Deflate d; return d; }

// The CRC32 of a BGZF block's inflated bytes against the one its footer carries (RFC 1952 2.3.1; htslib verifies it on
// every block behind aln_file.fetch, run_collection.py:23-26).  SVX_BGZF_CRC=0 switches the check off.
bool bgzf_crc_wanted() { static const bool on = [] { const char* e = getenv("SVX_BGZF_CRC"); return !(e && e[0] == '0'); }(); return on; }
uint32_t crc32_of(const uint8_t* p, size_t n)
{
    if (deflate_lib().crc) return deflate_lib().crc(0, p, n);
    return (uint32_t)crc32(crc32(0L, Z_NULL, 0), p, (uInt)n);
}

// one per worker thread and call
struct BlockInflater {
    void* ld = nullptr;
    BlockInflater() { if (deflate_lib().fast()) ld = deflate_lib().alloc(); }
    ~BlockInflater() { if (ld) deflate_lib().release(ld); }
    bool operator()(const uint8_t* in, size_t n_in, uint8_t* out, size_t n_out)
    {
        if (n_out == 0) return true;
        if (ld) {
            size_t got = 0;
            return deflate_lib().run(ld, in, n_in, out, n_out, &got) == 0 && got == n_out;
        }
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, -15) != Z_OK) return false;
        zs.next_in = const_cast<Bytef*>(in);
        zs.avail_in = (uInt)n_in;
        zs.next_out = out;
        zs.avail_out = (uInt)n_out;
        const int rc = inflate(&zs, Z_FINISH);
        inflateEnd(&zs);
        return rc == Z_STREAM_END && zs.avail_out == 0;
    }
};

// Persistent worker threads: run(n, grain, fn) hands out [lo, hi) slices of `grain` items from an atomic counter to
// the workers and the caller and returns when all are done.  (Round 2 spawned up to 64 std::threads per chunk.)
class Pool {
public:
    explicit Pool(int threads)
    {
        for (int t = 1; t < threads; ++t) workers_.emplace_back([this] { loop(); });
    }
    ~Pool()
    {
        { std::lock_guard<std::mutex> g(m_); stop_ = true; ++gen_; }
        cv_.notify_all();
        for (auto& th : workers_) th.join();
    }
    int size() const { return (int)workers_.size() + 1; }
    void run(size_t n, size_t grain, const std::function<void(size_t, size_t)>& fn)
    {
        if (n == 0) return;
        grain = std::max<size_t>(1, grain);
        if (workers_.empty() || n <= grain) { fn(0, n); return; }
        {
            std::lock_guard<std::mutex> g(m_);
            fn_ = &fn; n_ = n; grain_ = grain; next_ = 0; active_ = 0; ++gen_;
        }
        cv_.notify_all();
        drain();
        std::unique_lock<std::mutex> g(m_);
        done_.wait(g, [this] { return active_ == 0 && next_ >= n_; });
        fn_ = nullptr;
    }

private:
    void drain()
    {
        for (;;) {
            const size_t lo = next_.fetch_add(grain_);
            if (lo >= n_) return;
            (*fn_)(lo, std::min(n_, lo + grain_));
        }
    }
    void loop()
    {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                if (!fn_) continue;
                ++active_;
            }
            drain();
            {
                std::lock_guard<std::mutex> g(m_);
                --active_;
            }
            done_.notify_one();
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(size_t, size_t)>* fn_ = nullptr;
    size_t n_ = 0, grain_ = 1;
    std::atomic<size_t> next_{0};
    int active_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
};

// A run of whole BGZF blocks, still compressed (what the reader thread hands to the inflating one).
struct CChunk {
    struct Blk { uint64_t coff; uint32_t data, csize, isize, bsize; };   // file offset; payload offset / size in `bytes`; inflated size; block size
    RawBuf bytes;                                            // (no zero fill: fread overwrites it)
    std::vector<Blk> blk;
    uint64_t inflated = 0;
};

// Sequential reader of file[pos, end): read() fills a CChunk with the next whole blocks (about `want` compressed bytes,
// at most MAX_INFLATED inflated ones); a block cut by a read stays in the carry.  One fread per chunk: the page cache
// (or the disk) is copied once, by one thread, while other threads inflate the chunk before.
class BgzfReader {
public:
    static constexpr uint64_t MAX_INFLATED = 96ull << 20;
    // end == ~0ull: to the end of the file, which must then end with a whole block; otherwise a byte range that may be
    // cut inside its last block (callers ask for one block more than they need)
    BgzfReader(FILE* f, uint64_t pos, uint64_t end) : f_(f), pos_(pos), end_(end), strict_(end == ~0ull) { fseeko(f_, (off_t)pos, SEEK_SET); }
    bool done() const { return eof_ && carry_.empty(); }
    const std::string& error() const { return err_; }
    double seconds = 0;                                     // spent in fread

    bool read(CChunk& c, size_t want)                       // false on error; c.blk empty: nothing (more) to read
    {
        c.blk.clear();
        c.inflated = 0;
        if (done()) return true;
        const size_t have = carry_.size();
        const uint64_t ask = capped_ ? 0 : std::min<uint64_t>(want, end_ - pos_);   // no new read while the carry holds whole blocks
        c.bytes.resize(have + ask);
        if (have) memcpy(c.bytes.data(), carry_.data(), have);
        const double t0 = BamClock::now();
        const size_t got = ask ? fread(c.bytes.data() + have, 1, ask, f_) : 0;
        seconds += BamClock::now() - t0;
        if (got < ask) eof_ = true;
        c.bytes.resize(have + got);
        const uint64_t base = pos_ - have;                  // file offset of bytes[0]
        pos_ += got;
        if (pos_ >= end_) eof_ = true;
        const RawBuf& b = c.bytes;
        uint64_t p = 0;
        while (p + 18 <= b.size()) {
            if (!(b[p] == 0x1f && b[p + 1] == 0x8b && b[p + 2] == 8 && (b[p + 3] & 4))) { err_ = "not a BGZF block"; return false; }
            const uint32_t xlen = rd16(&b[p + 10]);
            if (p + 12 + xlen > b.size()) break;
            uint32_t bsize = 0;
            bool found = false;
            for (uint64_t q = p + 12; q + 4 <= p + 12 + xlen;) {
                const uint32_t slen = rd16(&b[q + 2]);
                if (b[q] == 'B' && b[q + 1] == 'C') { bsize = rd16(&b[q + 4]); found = true; }
                q += 4 + slen;
            }
            if (!found) { err_ = "corrupt BGZF block"; return false; }
            if (p + bsize + 1 > b.size()) break;            // partial block: wait for the next read
            const uint64_t data = p + 12 + xlen, end = p + bsize + 1;
            if (end < data + 8) { err_ = "corrupt BGZF block"; return false; }
            const uint32_t isize = rd32(&b[end - 4]);
            c.blk.push_back({base + p, (uint32_t)data, (uint32_t)(end - 8 - data), isize, (uint32_t)(end - p)});
            c.inflated += isize;
            p = end;
            if (c.inflated >= MAX_INFLATED) break;           // highly compressible input: the rest waits in the carry
        }
        capped_ = c.inflated >= MAX_INFLATED && p != b.size();
        if (eof_ && p != b.size() && !capped_ && strict_) { err_ = "truncated BGZF file"; return false; }
        // (a byte range cut in the middle of its last block: callers ask for one block more than they need)
        carry_.assign(b.data() + p, b.data() + b.size());
        if (eof_ && !capped_) carry_.clear();
        return true;
    }

private:
    FILE* f_;
    uint64_t pos_, end_;
    bool strict_, eof_ = false, capped_ = false;
    std::vector<uint8_t> carry_;
    std::string err_;
};

// position of a block's inflated bytes in an output buffer
struct OutBlock { uint64_t coff, dst; uint32_t isize, csize; };

// Inflates the blocks of `c` behind out.size(), block-parallel on `pool`; `blocks` receives where each one went.
bool inflate_chunk(const CChunk& c, RawBuf& out, Pool* pool, std::vector<OutBlock>* blocks)
{
    blocks->clear();
    uint64_t total = out.size();
    for (const auto& k : c.blk) { blocks->push_back({k.coff, total, k.isize, k.bsize}); total += k.isize; }
    out.resize(total);
    std::atomic<bool> ok{true};
    const std::vector<OutBlock>& ob = *blocks;
    // slices from a shared counter, about eight per thread: the blocks of a chunk inflate at different speeds
    pool->run(c.blk.size(), std::max<size_t>(1, c.blk.size() / (8 * (size_t)pool->size())), [&](size_t lo, size_t hi) {
        thread_local BlockInflater inflate_block;
        for (size_t i = lo; i < hi && ok; ++i) {
            const uint8_t* payload = &c.bytes[c.blk[i].data];
            if (!inflate_block(payload, c.blk[i].csize, out.data() + ob[i].dst, ob[i].isize)) { ok = false; return; }
            // the block's footer follows its payload: CRC32 of the inflated bytes, ISIZE
            if (bgzf_crc_wanted() && crc32_of(out.data() + ob[i].dst, ob[i].isize) != rd32(payload + c.blk[i].csize)) { ok = false; return; }
        }
    });
    return ok;
}

// The two steps in one call, for the callers that decode synchronously (header, svx_bam_open*).
class BgzfStream {
public:
    size_t chunk;                                        // compressed bytes read per step
    using Block = OutBlock;
    BgzfStream(FILE* f, uint64_t pos, uint64_t end, Pool* pool, size_t first_chunk) : chunk(first_chunk), reader_(f, pos, end), pool_(pool) {}
    const std::vector<Block>& blocks() const { return blocks_; }
    bool done() const { return reader_.done(); }
    const std::string& error() const { return err_; }
    bool next(RawBuf& out)                               // false on error; appends nothing when done()
    {
        blocks_.clear();
        if (done()) return true;
        if (!reader_.read(cc_, chunk)) { err_ = reader_.error(); return false; }
        g_clock.read = reader_.seconds;
        const double t0 = BamClock::now();
        const bool ok = inflate_chunk(cc_, out, pool_, &blocks_);
        g_clock.inflate += BamClock::now() - t0;
        if (!ok) { err_ = "BGZF inflate failed (corrupt block or CRC32 mismatch)"; return false; }
        return true;
    }

private:
    BgzfReader reader_;
    Pool* pool_;
    CChunk cc_;
    std::vector<Block> blocks_;
    std::string err_;
};

// Header (magic, text, reference dictionary) at the start of `buf`: 0 = needs more bytes, -1 = not BAM, else its size.
long long parse_header(const RawBuf& buf, Bam* b)
{
    if (buf.size() < 12) return 0;
    if (memcmp(buf.data(), "BAM\1", 4) != 0) return -1;
    const uint64_t l_text = rd32(&buf[4]);
    uint64_t p = 8 + l_text;
    if (p + 4 > buf.size()) return 0;
    const uint32_t n_ref = rd32(&buf[p]); p += 4;
    uint64_t q = p;
    for (uint32_t i = 0; i < n_ref; ++i) {
        if (q + 4 > buf.size()) return 0;
        const uint64_t l_name = rd32(&buf[q]);
        if (q + 8 + l_name > buf.size()) return 0;
        q += 8 + l_name;
    }
    b->header_text.assign(reinterpret_cast<const char*>(&buf[8]), strnlen(reinterpret_cast<const char*>(&buf[8]), l_text));
    for (uint32_t i = 0; i < n_ref; ++i) {
        const uint32_t l_name = rd32(&buf[p]);
        b->ref_names.emplace_back(reinterpret_cast<const char*>(&buf[p + 4]), l_name ? l_name - 1 : 0);
        b->ref_lens.push_back((int32_t)rd32(&buf[p + 4 + l_name]));
        p += 8 + l_name;
    }
    return (long long)p;
}

// Appends the whole records of buf[from, limit) to the arrays; returns the offset of the first byte not consumed
// (a partial record at the end stays for the next chunk), or -1 on a malformed record.
//
// split: a part holds the records of ONE reference (svx_bam_stream_*): the walk stops in front of the first record
// whose reference differs from the part's, and *tid_changed says so.
long long parse_records(const RawBuf& buf, uint64_t from, uint64_t limit, Pool* pool, bool keep_seq, Bam* b, bool split = false,
                        bool* tid_changed = nullptr)
{
    struct Rec { uint64_t at; const uint8_t* cig; uint32_t n_cig; };
    std::vector<Rec> recs;
    const double t_chain = BamClock::now();
    uint64_t p = from, words = 0, seq_bytes = 0;
    while (p + 4 <= limit) {
        const uint64_t bs = rd32(&buf[p]);
        if (p + 4 + bs > limit) break;
        if (bs < 32) return -1;
        if (p + 4 + bs + 128 <= limit) {                       // the next record's header: ~20 KB ahead in freshly inflated memory
            __builtin_prefetch(&buf[p + 4 + bs]);
            __builtin_prefetch(&buf[p + 4 + bs + 64]);
        }
        const uint8_t* rec = &buf[p + 4];
        if (split) {
            const int32_t rtid = (int32_t)rd32(rec);
            if (!b->tid.empty() || !recs.empty()) {
                const int32_t cur = b->tid.empty() ? (int32_t)rd32(&buf[recs[0].at + 4]) : b->tid[0];
                if (rtid != cur) { *tid_changed = true; break; }
            }
        }
        const uint32_t l_name = rec[8], l_seq = rd32(rec + 16);
        uint32_t n_cig = rd16(rec + 12);
        if (32ull + l_name + 4ull * n_cig + (l_seq + 1ull) / 2 + l_seq > bs) return -1;
        const uint8_t* cig = rec + 32 + l_name;
        if (n_cig == 2) {                                     // possible CG-tag placeholder
            const uint32_t w0 = rd32(cig), w1 = rd32(cig + 4);
            if ((w0 & 15) == 4 && (w0 >> 4) == l_seq && (w1 & 15) == 3) {
                uint32_t cnt = 0;
                const uint8_t* at = find_long_cigar(cig + 8 + (l_seq + 1) / 2 + l_seq, rec + bs, &cnt);
                if (at) { cig = at; n_cig = cnt; }
            }
        }
        recs.push_back({p, cig, n_cig});
        words += n_cig;
        seq_bytes += (l_seq + 1ull) / 2;
        // QNAME ids by first occurrence (sequential: the order defines the ids)
        std::string_view nm(reinterpret_cast<const char*>(rec + 32), l_name ? l_name - 1 : 0);
        auto it = b->seen.find(nm);
        if (it == b->seen.end()) {
            const std::string_view kept = b->arena.keep(nm);
            it = b->seen.emplace(kept, (int32_t)b->names.size()).first;
            b->names.push_back(kept);
            b->names_bytes += kept.size() + 1;
        }
        b->name_id.push_back(it->second);
        p += 4 + bs;
    }
    const size_t n0 = b->tid.size(), n = recs.size();
    b->tid.resize(n0 + n); b->pos.resize(n0 + n); b->l_seq.resize(n0 + n); b->flag.resize(n0 + n); b->mapq.resize(n0 + n);
    b->cig_off.resize(n0 + n + 1);
    const uint64_t w0 = b->cigar.size();
    b->cigar.resize(w0 + words);
    const uint64_t s0 = b->seq.size();
    if (keep_seq) { b->seq.resize(s0 + seq_bytes); b->seq_off.resize(n0 + n); }
    {   // CSR offsets (sequential prefix), then the scatter in parallel
        uint64_t w = w0, s = s0;
        for (size_t i = 0; i < n; ++i) {
            w += recs[i].n_cig;
            b->cig_off[n0 + i + 1] = w;
            if (keep_seq) { b->seq_off[n0 + i] = s; s += (rd32(&buf[recs[i].at + 4 + 16]) + 1ull) / 2; }
        }
    }
    const double t_scatter = BamClock::now();
    g_clock.chain += t_scatter - t_chain;
    pool->run(n, 2048, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            const uint8_t* rec = &buf[recs[i].at + 4];
            const size_t k = n0 + i;
            b->tid[k] = (int32_t)rd32(rec); b->pos[k] = (int32_t)rd32(rec + 4);
            b->mapq[k] = rec[9];
            b->flag[k] = rd16(rec + 14);
            b->l_seq[k] = (int32_t)rd32(rec + 16);
            memcpy(&b->cigar[b->cig_off[k]], recs[i].cig, 4ull * recs[i].n_cig);
            if (keep_seq) memcpy(&b->seq[b->seq_off[k]], rec + 32 + rec[8] + 4ull * rd16(rec + 12), ((uint32_t)b->l_seq[k] + 1ull) / 2);
        }
    });
    g_clock.scatter += BamClock::now() - t_scatter;
    return (long long)p;
}

thread_local std::string g_bam_error;
int default_threads();

void* bam_open_impl(const char* path, int threads, int flags, bool ranged, uint64_t voff_beg, uint64_t voff_end)
{
    g_bam_error.clear();
    if (threads <= 0) threads = default_threads();        // the CPUs this process may run on (not the machine's), at most 64
    const bool keep_seq = flags & SVX_BAM_KEEP_SEQ;
    FILE* f = fopen(path, "rb");
    if (!f) { g_bam_error = std::string("cannot open ") + path; return nullptr; }
    struct Closer { FILE* f; ~Closer() { fclose(f); } } closer{f};
    Pool workers(threads);
    Pool* pool = &workers;
    std::unique_ptr<Bam> b(new Bam());
    RawBuf buf;
    auto fail = [&](const std::string& why) -> void* { g_bam_error = why; return nullptr; };

    // header: from the start of the file, chunk by chunk until the reference dictionary is complete
    // SVX_BAM_CHUNK (bytes) shrinks the read size so that tests cross chunk boundaries on small files
    const char* env = getenv("SVX_BAM_CHUNK");
    const size_t steady = env && atol(env) > 0 ? (size_t)atol(env) : (16u << 20);   // inflated chunk stays cache-warm for the parse
    BgzfStream head(f, 0, ~0ull, pool, std::min<size_t>(steady, 256u << 10));
    long long hdr = 0;
    while (hdr == 0) {
        if (head.done()) return fail(buf.empty() ? "empty file" : "truncated BAM header");
        if (!head.next(buf)) return fail(head.error());
        hdr = parse_header(buf, b.get());
        head.chunk = std::min<size_t>(head.chunk * 4, steady);
    }
    if (hdr < 0) return fail("not a BAM file");
    head.chunk = steady;

    if (!ranged) {
        uint64_t cur = (uint64_t)hdr;
        for (;;) {
            const long long used = parse_records(buf, cur, buf.size(), pool, keep_seq, b.get());
            if (used < 0) return fail("malformed BAM record");
            buf.drop_front((size_t)used);                       // keeps a partial record for the next chunk
            cur = 0;
            if (head.done()) break;
            if (!head.next(buf)) return fail(head.error());
        }
        if (!buf.empty()) return fail("truncated BAM record");
    } else if (voff_end > voff_beg) {
        // the compressed range [coffset(voff_beg), end of the block at coffset(voff_end)): one spare block is read
        const uint64_t c0 = voff_beg >> 16, c1 = voff_end >> 16;
        BgzfStream body(f, c0, c1 + 65536 + 26, pool, steady);
        buf.clear();
        uint64_t cur = voff_beg & 0xffff, limit = ~0ull;      // limit: position in buf of (c1, voff_end & 0xffff)
        bool reached = false;
        while (!reached && !body.done()) {
            if (!body.next(buf)) return fail(body.error());
            for (const auto& blk : body.blocks()) {
                if (reached) break;
                if (blk.coff == c1) { limit = blk.dst + (voff_end & 0xffff); reached = true; }
                else if (blk.coff > c1) { limit = blk.dst; reached = true; }
            }
            const uint64_t stop = std::min<uint64_t>(limit, buf.size());
            if (cur > stop) {                                  // the first block has not arrived in full yet
                if (reached) return fail("BAM index does not match the file");
                continue;
            }
            const long long used = parse_records(buf, cur, stop, pool, keep_seq, b.get());
            if (used < 0) return fail("malformed BAM record");
            if (reached) { if ((uint64_t)used != stop) return fail("BAM index does not match the file"); break; }
            buf.drop_front((size_t)used);
            cur = 0;
        }
        if (!reached) return fail("BAM index does not match the file");
    }
    if (getenv("SVX_TIMING"))
        fprintf(stderr, "svx_bam_open (%d threads, %s): read %.3f s, inflate %.3f s, chain records %.3f s, scatter fields %.3f s\n", threads,
                deflate_lib().fast() ? "libdeflate" : "zlib", g_clock.read, g_clock.inflate, g_clock.chain, g_clock.scatter);
    g_clock = BamClock();
    return b.release();
}


// ------------------------------------------------------------------------------------------------------------------
// Streaming ingestion, one reference at a time (svx_bam_stream_*): the consumer gets the records of chromosome k as
// soon as they are decoded, while chromosome k+1 is being read and inflated.  Two threads behind the handle:
//
//   feeder   reads the next chunk of whole BGZF blocks and inflates it on the worker pool into one of a few chunk
//            buffers (data starts HEAD bytes into the buffer);
//   parser   walks the records of the chunk in hand (the partial record a chunk ends with is copied in front of the
//            next chunk's data, into the HEAD room), splits at every change of reference and queues finished parts.
//
// The reference fetches window by window through the index (run_collection.py:23-26); here a rank's byte ranges (its
// chromosomes, from the .bai) or the whole file are streamed once, in file order.
struct Stream {
    static constexpr size_t HEAD = 8u << 20;                 // room in front of a chunk for the previous chunk's partial record
    struct Chunk {
        RawBuf buf;
        size_t begin = HEAD;                                  // first byte to parse
        size_t limit = 0;                                     // one past the last byte to parse (range mode: the end virtual offset)
        bool range_end = false;                               // nothing of this range follows
        bool fresh = true;                                    // first chunk of a range: no carry from the chunk before
    };
    std::string path;
    FILE* f = nullptr;
    int flags = 0;
    std::unique_ptr<Pool> inflate_pool, scatter_pool;
    std::vector<std::pair<uint64_t, uint64_t>> ranges;       // virtual offsets; empty: everything behind the header
    Bam proto;                                               // header text + reference dictionary, copied into every part
    uint64_t body_voff = 0;                                  // virtual offset of the first record (whole-file mode)

    std::mutex m;
    std::condition_variable cv;
    std::deque<std::unique_ptr<Chunk>> filled, spare;        // feeder -> parser, parser -> feeder
    std::deque<std::unique_ptr<Bam>> ready;                  // parser -> consumer
    size_t max_ready = 2;
    bool feeder_done = false, parser_done = false, cancel = false;
    std::string error;
    std::thread feeder, parser;
    double t_read = 0, t_inflate = 0, t_chain = 0, t_wait_chunk = 0, t_wait_ready = 0;
    uint64_t bytes_in = 0, bytes_out = 0;

    ~Stream()
    {
        { std::lock_guard<std::mutex> g(m); cancel = true; }
        cv.notify_all();
        if (reader.joinable()) reader.join();
        if (feeder.joinable()) feeder.join();
        if (parser.joinable()) parser.join();
        if (f) fclose(f);
    }

    void fail(const std::string& why)
    {
        std::lock_guard<std::mutex> g(m);
        if (error.empty()) error = why;
        cancel = true;
        cv.notify_all();
    }

    // ---- reader: compressed chunks, one fread each, a few ahead of the inflate ------------------------------------
    struct CItem {
        CChunk c;
        int range = 0;                                        // index into the range list
        bool first = false, last = false;                     // first / last chunk of its range
    };
    std::deque<std::unique_ptr<CItem>> cfilled, cspare;       // reader -> feeder, feeder -> reader
    bool reader_done = false;
    std::thread reader;
    std::vector<std::pair<uint64_t, uint64_t>> todo;

    void read_ahead()
    {
        const char* env = getenv("SVX_BAM_CHUNK");
        const size_t steady = env && atol(env) > 0 ? (size_t)atol(env) : (40u << 20);      // ~96 MB inflated per step
        for (size_t ri = 0; ri < todo.size(); ++ri) {
            const auto& r = todo[ri];
            const bool open_end = r.second == ~0ull;
            if (!open_end && r.second <= r.first) continue;
            const uint64_t c0 = r.first >> 16, c1 = open_end ? ~0ull : r.second >> 16;
            BgzfReader body(f, c0, open_end ? ~0ull : c1 + 65536 + 26);
            bool first = true;
            for (;;) {
                std::unique_ptr<CItem> it;
                {
                    std::unique_lock<std::mutex> g(m);
                    cv.wait(g, [&] { return cancel || !cspare.empty(); });
                    if (cancel) return;
                    it = std::move(cspare.front());
                    cspare.pop_front();
                }
                do {                                            // (a read shorter than one block hands over nothing: read on)
                    if (!body.read(it->c, steady)) { fail(body.error()); return; }
                } while (it->c.blk.empty() && !body.done());
                it->range = (int)ri;
                it->first = first;
                it->last = body.done();
                first = false;
                const bool last = it->last;
                {
                    std::lock_guard<std::mutex> g(m);
                    cfilled.push_back(std::move(it));
                }
                cv.notify_all();
                if (last) break;
            }
            t_read += body.seconds;
        }
        {
            std::lock_guard<std::mutex> g(m);
            reader_done = true;
        }
        cv.notify_all();
    }

    // ---- feeder: inflates the chunks the reader has ready -------------------------------------------------------------
    void feed()
    {
        std::vector<OutBlock> blocks;
        bool reached = false;                                   // the current range's end offset has been seen: skip its tail
        for (;;) {
            std::unique_ptr<CItem> it;
            {
                std::unique_lock<std::mutex> g(m);
                cv.wait(g, [&] { return cancel || !cfilled.empty() || reader_done; });
                if (cancel) return;
                if (cfilled.empty()) break;
                it = std::move(cfilled.front());
                cfilled.pop_front();
            }
            const auto& r = todo[it->range];
            const bool open_end = r.second == ~0ull;
            const uint64_t c1 = open_end ? ~0ull : r.second >> 16;
            if (it->first) reached = false;
            if (!reached) {
                std::unique_ptr<Chunk> ch;
                {
                    std::unique_lock<std::mutex> g(m);
                    cv.wait(g, [&] { return cancel || !spare.empty(); });
                    if (cancel) return;
                    ch = std::move(spare.front());
                    spare.pop_front();
                }
                ch->buf.resize(HEAD);                           // the inflated bytes go behind the HEAD room
                const double t0 = BamClock::now();
                if (!inflate_chunk(it->c, ch->buf, inflate_pool.get(), &blocks)) { fail("BGZF inflate failed (corrupt block or CRC32 mismatch)"); return; }
                t_inflate += BamClock::now() - t0;
                ch->fresh = it->first;
                ch->begin = HEAD + (it->first ? (size_t)(r.first & 0xffff) : 0);
                ch->limit = ch->buf.size();
                for (const auto& blk : blocks) {
                    bytes_out += blk.isize;
                    bytes_in += blk.csize;
                    if (open_end || reached) continue;
                    if (blk.coff == c1) { ch->limit = blk.dst + (size_t)(r.second & 0xffff); reached = true; }
                    else if (blk.coff > c1) { ch->limit = blk.dst; reached = true; }
                }
                if (it->last) {
                    if (!open_end && !reached) { fail("BAM index does not match the file"); return; }
                    reached = true;
                }
                if (it->first && ch->begin > ch->limit) { fail("BAM index does not match the file"); return; }
                ch->range_end = reached;
                {
                    std::lock_guard<std::mutex> g(m);
                    filled.push_back(std::move(ch));
                }
            }
            {
                std::lock_guard<std::mutex> g(m);
                cspare.push_back(std::move(it));
            }
            cv.notify_all();
        }
        {
            std::lock_guard<std::mutex> g(m);
            feeder_done = true;
        }
        cv.notify_all();
    }

    // ---- parser ------------------------------------------------------------------------------------------------
    std::unique_ptr<Bam> new_part() const
    {
        std::unique_ptr<Bam> b(new Bam());
        b->header_text = proto.header_text;
        b->ref_names = proto.ref_names;
        b->ref_lens = proto.ref_lens;
        b->seen.reserve(1u << 18);                            // a chromosome holds 10^5..10^6 reads: no rehash while the part grows
        return b;
    }

    bool publish(std::unique_ptr<Bam>& part)                  // false: cancelled
    {
        if (part->tid.empty()) return true;
        std::unique_lock<std::mutex> g(m);
        const double t0 = BamClock::now();
        cv.wait(g, [&] { return cancel || ready.size() < max_ready; });
        t_wait_ready += BamClock::now() - t0;
        if (cancel) return false;
        ready.push_back(std::move(part));
        g.unlock();
        cv.notify_all();
        part = new_part();
        return true;
    }

    void parse()
    {
        const bool keep_seq = flags & SVX_BAM_KEEP_SEQ;
        std::unique_ptr<Bam> part = new_part();
        std::vector<uint8_t> carry;                           // partial record left by the previous chunk of the range
        for (;;) {
            std::unique_ptr<Chunk> ch;
            {
                std::unique_lock<std::mutex> g(m);
                const double t0 = BamClock::now();
                cv.wait(g, [&] { return cancel || !filled.empty() || feeder_done; });
                t_wait_chunk += BamClock::now() - t0;
                if (cancel) return;
                if (filled.empty()) break;                    // feeder_done and nothing left
                ch = std::move(filled.front());
                filled.pop_front();
            }
            if (ch->fresh) {
                if (!carry.empty()) { fail("truncated BAM record"); return; }
            } else if (!carry.empty()) {
                if (carry.size() > ch->begin) {               // a record longer than the HEAD room: make room (rare: > 8 MB)
                    RawBuf bigger;
                    bigger.resize(carry.size() + (ch->buf.size() - ch->begin));
                    memcpy(bigger.data() + carry.size(), ch->buf.data() + ch->begin, ch->buf.size() - ch->begin);
                    const size_t shift = carry.size() - ch->begin;
                    ch->limit += shift;
                    ch->begin = carry.size();
                    ch->buf = std::move(bigger);
                }
                memcpy(ch->buf.data() + ch->begin - carry.size(), carry.data(), carry.size());
                ch->begin -= carry.size();
                carry.clear();
            }
            const double t0 = BamClock::now();
            uint64_t cur = ch->begin;
            const uint64_t stop = std::min<uint64_t>(ch->limit, ch->buf.size());
            for (;;) {
                bool changed = false;
                const long long used = parse_records(ch->buf, cur, stop, scatter_pool.get(), keep_seq, part.get(), true, &changed);
                if (used < 0) { fail("malformed BAM record"); return; }
                cur = (uint64_t)used;
                if (!changed) break;
                t_chain += BamClock::now() - t0;
                if (!publish(part)) return;
            }
            t_chain += BamClock::now() - t0;
            if (cur < stop) {
                if (ch->range_end) { fail(ch->limit < ch->buf.size() ? "BAM index does not match the file" : "truncated BAM record"); return; }
                carry.assign(ch->buf.data() + cur, ch->buf.data() + stop);
            }
            const bool range_end = ch->range_end;
            {
                std::lock_guard<std::mutex> g(m);
                spare.push_back(std::move(ch));
            }
            cv.notify_all();
            if (range_end && !publish(part)) return;          // ranges are whole references: never merged across a gap
        }
        if (!carry.empty()) { fail("truncated BAM record"); return; }
        if (!publish(part)) return;
        {
            std::lock_guard<std::mutex> g(m);
            parser_done = true;
        }
        cv.notify_all();
        if (getenv("SVX_TIMING"))
            fprintf(stderr, "svx_bam_stream (%d inflate threads, %s): %.1f MB in, %.1f MB inflated; feeder read %.3f s, inflate %.3f s; "
                            "parser walk+scatter %.3f s, waited for chunks %.3f s, for the consumer %.3f s\n",
                    inflate_pool->size(), deflate_lib().fast() ? "libdeflate" : "zlib", bytes_in / 1e6, bytes_out / 1e6, t_read, t_inflate,
                    t_chain, t_wait_chunk, t_wait_ready);
    }
};

int default_threads()
{
    cpu_set_t set;
    const int avail = sched_getaffinity(0, sizeof set, &set) == 0 ? CPU_COUNT(&set) : (int)std::thread::hardware_concurrency();
    return std::max(1, std::min(128, avail));
}

}  // namespace

extern "C" {

void* svx_bam_open(const char* path, int threads, int flags) { return bam_open_impl(path, threads, flags, false, 0, 0); }

// Only the records between two BGZF virtual offsets (from the .bai index: one chromosome, one rank's shard);
// the header / reference dictionary is always decoded.  voff_end <= voff_beg: header only.
void* svx_bam_open_range(const char* path, int threads, int flags, uint64_t voff_beg, uint64_t voff_end)
{
    return bam_open_impl(path, threads, flags, true, voff_beg, voff_end);
}

const char* svx_bam_error(void) { return g_bam_error.c_str(); }

// sizes: [n_records, n_cigar_words, n_refs, n_names, names_bytes, header_bytes, ref_names_bytes, seq_bytes]
void svx_bam_sizes(void* h, uint64_t* sizes)
{
    const Bam* b = static_cast<const Bam*>(h);
    uint64_t rn = 0;
    for (auto& s : b->ref_names) rn += s.size() + 1;
    sizes[0] = b->tid.size(); sizes[1] = b->cigar.size(); sizes[2] = b->ref_names.size();
    sizes[3] = b->names.size(); sizes[4] = b->names_bytes; sizes[5] = b->header_text.size();
    sizes[6] = rn; sizes[7] = b->seq.size();
}

// Fill caller-owned arrays (all sized from svx_bam_sizes).  seq_off may be NULL; otherwise it receives the byte
// offset of each record's 4-bit SEQ inside the pool svx_bam_seq() exposes (handles opened with SVX_BAM_KEEP_SEQ).
void svx_bam_export(void* h, int threads, int32_t* tid, int32_t* pos, uint16_t* flag, uint8_t* mapq, int32_t* l_seq,
                    int32_t* name_id, int64_t* cig_off, uint32_t* cigar, char* names, char* header, char* ref_names,
                    int32_t* ref_lens, int64_t* seq_off)
{
    (void)threads;
    const Bam* b = static_cast<const Bam*>(h);
    const size_t n = b->tid.size();
    if (n) {
        memcpy(tid, b->tid.data(), 4 * n); memcpy(pos, b->pos.data(), 4 * n); memcpy(flag, b->flag.data(), 2 * n);
        memcpy(mapq, b->mapq.data(), n); memcpy(l_seq, b->l_seq.data(), 4 * n); memcpy(name_id, b->name_id.data(), 4 * n);
    }
    for (size_t i = 0; i <= n; ++i) cig_off[i] = (int64_t)b->cig_off[i];
    if (!b->cigar.empty()) memcpy(cigar, b->cigar.data(), 4 * b->cigar.size());
    if (seq_off)
        for (size_t i = 0; i < n; ++i) seq_off[i] = i < b->seq_off.size() ? (int64_t)b->seq_off[i] : 0;
    char* w = names;
    for (const auto& nm : b->names) { memcpy(w, nm.data(), nm.size()); w[nm.size()] = '\n'; w += nm.size() + 1; }
    memcpy(header, b->header_text.data(), b->header_text.size());
    w = ref_names;
    for (size_t i = 0; i < b->ref_names.size(); ++i) {
        memcpy(w, b->ref_names[i].data(), b->ref_names[i].size()); w[b->ref_names[i].size()] = '\n';
        w += b->ref_names[i].size() + 1;
        ref_lens[i] = b->ref_lens[i];
    }
}

const uint8_t* svx_bam_seq(void* h) { return static_cast<const Bam*>(h)->seq.data(); }

// ---- streaming: one part per reference, decoded ahead of the consumer ------------------------------------------
void* svx_bam_stream_open(const char* path, int threads, int flags, const uint64_t* voffs, int n_ranges)
{
    g_bam_error.clear();
    if (threads <= 0) threads = default_threads();
    std::unique_ptr<Stream> s(new Stream());
    s->path = path;
    s->flags = flags;
    s->f = fopen(path, "rb");
    if (!s->f) { g_bam_error = std::string("cannot open ") + path; return nullptr; }
    for (int i = 0; i < n_ranges; ++i) s->ranges.push_back({voffs[2 * i], voffs[2 * i + 1]});
    // the parser's field scatter is a few memcpy per record: a handful of threads; everything else inflates
    const int scatter = std::max(1, std::min(8, threads / 8));
    s->inflate_pool.reset(new Pool(std::max(1, threads - scatter)));
    s->scatter_pool.reset(new Pool(scatter));
    {   // header: synchronously, from the start of the file; remembers where the records begin
        RawBuf buf;
        BgzfStream head(s->f, 0, ~0ull, s->inflate_pool.get(), 256u << 10);
        long long hdr = 0;
        std::vector<BgzfStream::Block> seen;
        while (hdr == 0) {
            if (head.done()) { g_bam_error = buf.empty() ? "empty file" : "truncated BAM header"; return nullptr; }
            if (!head.next(buf)) { g_bam_error = head.error(); return nullptr; }
            for (const auto& b : head.blocks()) seen.push_back(b);
            hdr = parse_header(buf, &s->proto);
            head.chunk = std::min<size_t>(head.chunk * 4, 16u << 20);
        }
        if (hdr < 0) { g_bam_error = "not a BAM file"; return nullptr; }
        // virtual offset of byte `hdr` of the inflated stream: inside a block that was read, or the start of the block
        // behind the last one (a header that fills its blocks exactly, as htslib writes it)
        uint64_t voff = ~0ull;
        for (const auto& b : seen)
            if ((uint64_t)hdr < b.dst + b.isize) { voff = (b.coff << 16) | ((uint64_t)hdr - b.dst); break; }
        if (voff == ~0ull) voff = (seen.back().coff + seen.back().csize) << 16;
        s->body_voff = voff;
    }
    for (int i = 0; i < 3; ++i) s->spare.emplace_back(new Stream::Chunk());
    for (int i = 0; i < 3; ++i) s->cspare.emplace_back(new Stream::CItem());
    s->todo = s->ranges;
    if (s->todo.empty()) s->todo.push_back({s->body_voff, ~0ull});
    Stream* raw = s.get();
    s->reader = std::thread([raw] { raw->read_ahead(); });
    s->feeder = std::thread([raw] { raw->feed(); });
    s->parser = std::thread([raw] { raw->parse(); });
    return s.release();
}

// Next reference's records: a handle for svx_bam_sizes / svx_bam_export / svx_bam_seq / svx_bam_close, or NULL with
// *status = 0 at the end of the stream, -1 on an error (svx_bam_error()).  Blocks while the part is being decoded.
void* svx_bam_stream_next(void* stream, int* status)
{
    Stream* s = static_cast<Stream*>(stream);
    std::unique_lock<std::mutex> g(s->m);
    s->cv.wait(g, [&] { return !s->ready.empty() || s->parser_done || !s->error.empty(); });
    if (!s->ready.empty()) {
        std::unique_ptr<Bam> part = std::move(s->ready.front());
        s->ready.pop_front();
        g.unlock();
        s->cv.notify_all();
        *status = 1;
        return part.release();
    }
    if (!s->error.empty()) { g_bam_error = s->error; *status = -1; return nullptr; }
    *status = 0;
    return nullptr;
}

void svx_bam_stream_close(void* stream) { delete static_cast<Stream*>(stream); }

// ---- host helpers of the device-side ingestion (svx_inflate.hip, svx_bamdev.hip) ------------------------------------
// file[off, off + n) -> dst (pinned memory of the caller), `threads` positional reads at once: one thread copies the page
// cache at ~10 GB/s, the upload runs at ~55
int svx_read_range(const char* path, uint64_t off, uint64_t n, uint8_t* dst, int threads)
{
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) { g_bam_error = std::string("cannot open ") + path; return SVX_EINVAL; }
    threads = std::max(1, std::min(threads, 32));
    std::atomic<bool> ok{true};
    const uint64_t piece = ((n + threads - 1) / threads + 4095) & ~4095ull;
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t) {
        const uint64_t lo = std::min<uint64_t>(n, t * piece), hi = std::min<uint64_t>(n, lo + piece);
        if (lo >= hi) break;
        pool.emplace_back([=, &ok] {
            uint64_t done = lo;
            while (done < hi) {
                const ssize_t got = pread(fd, dst + done, hi - done, (off_t)(off + done));
                if (got <= 0) { ok = false; return; }
                done += (uint64_t)got;
            }
        });
    }
    for (auto& th : pool) th.join();
    ::close(fd);
    if (!ok) { g_bam_error = "short read"; return SVX_EINVAL; }
    return SVX_OK;
}

// Whole BGZF blocks at the start of bytes[0, n): for each its payload offset / size in `bytes`, its inflated size and
// its own offset (+ base_coff = its file offset).  -> number of blocks (at most cap), or -1 (not BGZF).  *used = bytes
// covered by them (a block cut by the end of the buffer is not counted).
int64_t svx_bgzf_index(const uint8_t* b, uint64_t n, uint64_t base_coff, uint64_t cap, uint64_t* src_off, uint32_t* src_len,
                       uint32_t* isize, uint64_t* coff, uint64_t* used)
{
    uint64_t p = 0, k = 0;
    while (p + 18 <= n && k < cap) {
        if (!(b[p] == 0x1f && b[p + 1] == 0x8b && b[p + 2] == 8 && (b[p + 3] & 4))) return -1;
        const uint32_t xlen = rd16(&b[p + 10]);
        if (p + 12 + xlen > n) break;
        uint32_t bsize = 0;
        bool found = false;
        for (uint64_t q = p + 12; q + 4 <= p + 12 + xlen;) {
            const uint32_t slen = rd16(&b[q + 2]);
            if (b[q] == 'B' && b[q + 1] == 'C') { bsize = rd16(&b[q + 4]); found = true; }
            q += 4 + slen;
        }
        if (!found) return -1;
        if (p + bsize + 1 > n) break;
        const uint64_t data = p + 12 + xlen, end = p + bsize + 1;
        if (end < data + 8) return -1;
        src_off[k] = data;
        src_len[k] = (uint32_t)(end - 8 - data);
        isize[k] = rd32(&b[end - 4]);
        coff[k] = base_coff + p;
        ++k;
        p = end;
    }
    *used = p;
    return (int64_t)k;
}

// QNAME ids by first occurrence (what the host decoder computes while it chains the records): names = the records'
// names, '\n'-separated, name_off [n + 1] their offsets.  name_id [n]; uniq receives the distinct names in id order,
// '\n'-separated (*uniq_bytes of them).  -> number of distinct names.
int64_t svx_name_ids(const uint8_t* names, const int64_t* name_off, uint64_t n, int32_t* name_id, uint8_t* uniq, uint64_t* uniq_bytes)
{
    std::unordered_map<std::string_view, int32_t> seen;
    seen.reserve(n);
    uint64_t w = 0;
    int32_t next = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const std::string_view nm(reinterpret_cast<const char*>(names + name_off[i]), (size_t)(name_off[i + 1] - name_off[i] - 1));
        auto it = seen.find(nm);
        if (it == seen.end()) {
            it = seen.emplace(nm, next++).first;
            memcpy(uniq + w, nm.data(), nm.size());
            uniq[w + nm.size()] = '\n';
            w += nm.size() + 1;
        }
        name_id[i] = it->second;
    }
    *uniq_bytes = w;
    return next;
}

void svx_bam_close(void* h) { delete static_cast<Bam*>(h); }

}  // extern "C"
