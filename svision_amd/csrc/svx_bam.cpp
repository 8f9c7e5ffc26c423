// svx_bam.cpp -- native BGZF/BAM ingestion: file -> packed structure of arrays (host side of libsvx.so).
//
// Replaces what the reference obtains record by record from pysam/htslib (AlignmentFile iteration at
// src/collection/run_collection.py:23-26, src/collection/collect_signatures.py:128-155): the whole file is
// inflated block-parallel (BGZF blocks are independent gzip members; zlib raw inflate on a thread pool),
// record offsets are chained once, and the fixed fields / CIGAR words / names / 4-bit sequences are
// scattered in parallel into caller-owned arrays that go to the GPU unchanged (svx_cigar_scan input).
// SAMv1 section 4 layouts, including CIGARs with more than 65535 operations (CG:B,I tag).
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/svx.h"

namespace {

struct Bam {
    std::vector<uint8_t> raw;              // decompressed file
    std::string header_text;
    std::vector<std::string> ref_names;
    std::vector<int32_t> ref_lens;
    std::vector<uint64_t> rec_off;         // offset of each record's block_size field
    std::vector<uint64_t> cig_off;         // CSR over CIGAR words, n_rec + 1
    std::vector<uint64_t> cig_src;         // byte offset of each record's CIGAR words (record body or CG:B,I tag)
    std::vector<int32_t> name_id;
    std::vector<uint32_t> name_first;      // record index of the first occurrence of every distinct QNAME
    uint64_t names_bytes = 0;
    std::string error;
};

inline uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
inline uint16_t rd16(const uint8_t* p) { uint16_t v; memcpy(&v, p, 2); return v; }

// CIGARs with more than 65535 operations live in the CG:B,I tag; the record then carries "<l_seq>S<ref_len>N"
// (SAMv1 4.2.2).  Returns the byte offset of the tag's uint32 array and its length, or 0.
uint64_t find_long_cigar(const std::vector<uint8_t>& r, uint64_t aux, uint64_t end, uint32_t* count)
{
    while (aux + 3 <= end) {
        const uint8_t t0 = r[aux], t1 = r[aux + 1], ty = r[aux + 2];
        uint64_t p = aux + 3;
        switch (ty) {
        case 'A': case 'c': case 'C': p += 1; break;
        case 's': case 'S': p += 2; break;
        case 'i': case 'I': case 'f': p += 4; break;
        case 'Z': case 'H': while (p < end && r[p]) ++p; ++p; break;
        case 'B': {
            if (p + 5 > end) return 0;
            const uint8_t sub = r[p];
            const uint32_t n = rd32(&r[p + 1]);
            const uint32_t width = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
            if (t0 == 'C' && t1 == 'G' && sub == 'I') { *count = n; return p + 5; }
            p += 5 + (uint64_t)n * width;
            break;
        }
        default: return 0;
        }
        aux = p;
    }
    return 0;
}

template <class F>
void parallel_for(size_t n, int threads, F fn)
{
    if (threads <= 1 || n < 2) { fn(0, n); return; }
    std::vector<std::thread> pool;
    const size_t chunk = (n + threads - 1) / threads;
    for (int t = 0; t < threads; ++t) {
        const size_t lo = std::min(n, t * chunk), hi = std::min(n, lo + chunk);
        if (lo < hi) pool.emplace_back([=] { fn(lo, hi); });
    }
    for (auto& th : pool) th.join();
}

// Inflate the BGZF blocks of file[from, to) (block-aligned) into `out`; max_blocks limits the number of blocks (0 = all).
bool inflate_file(const std::vector<uint8_t>& file, int threads, std::vector<uint8_t>& out, std::string& err,
                  uint64_t from = 0, uint64_t to = ~0ull, size_t max_blocks = 0)
{
    struct Blk { uint64_t src, csize, dst; uint32_t isize; };
    std::vector<Blk> blocks;
    uint64_t p = from, total = 0;
    const uint64_t n = std::min<uint64_t>(file.size(), to);
    while (p + 18 <= n && (max_blocks == 0 || blocks.size() < max_blocks)) {
        if (!(file[p] == 0x1f && file[p + 1] == 0x8b && file[p + 2] == 8 && (file[p + 3] & 4))) { err = "not a BGZF block"; return false; }
        const uint32_t xlen = rd16(&file[p + 10]);
        uint32_t bsize = 0;
        bool found = false;
        for (uint64_t q = p + 12; q + 4 <= p + 12 + xlen;) {
            const uint32_t slen = rd16(&file[q + 2]);
            if (file[q] == 'B' && file[q + 1] == 'C') { bsize = rd16(&file[q + 4]); found = true; }
            q += 4 + slen;
        }
        if (!found || p + bsize + 1 > n) { err = "corrupt BGZF block"; return false; }
        const uint64_t data = p + 12 + xlen, end = p + bsize + 1;
        const uint32_t isize = rd32(&file[end - 4]);
        blocks.push_back({data, end - 8 - data, total, isize});
        total += isize;
        p = end;
    }
    out.resize(total);
    std::atomic<bool> ok{true};
    parallel_for(blocks.size(), threads, [&](size_t lo, size_t hi) {
        z_stream zs;
        for (size_t i = lo; i < hi && ok; ++i) {
            const Blk& b = blocks[i];
            if (b.isize == 0) continue;
            memset(&zs, 0, sizeof zs);
            if (inflateInit2(&zs, -15) != Z_OK) { ok = false; return; }
            zs.next_in = const_cast<Bytef*>(&file[b.src]);
            zs.avail_in = (uInt)b.csize;
            zs.next_out = &out[b.dst];
            zs.avail_out = b.isize;
            const int rc = inflate(&zs, Z_FINISH);
            inflateEnd(&zs);
            if (rc != Z_STREAM_END || zs.avail_out != 0) { ok = false; return; }
        }
    });
    if (!ok) { err = "BGZF inflate failed"; return false; }
    return true;
}

}  // namespace

extern "C" {

// Decode a BAM file.  Returns an opaque handle (NULL on failure; message via svx_bam_error(NULL)).
static thread_local std::string g_bam_error;

static bool read_bytes(const char* path, uint64_t from, uint64_t to, std::vector<uint8_t>& buf)
{
    FILE* f = fopen(path, "rb");
    if (!f) { g_bam_error = std::string("cannot open ") + path; return false; }
    fseek(f, 0, SEEK_END);
    const uint64_t sz = (uint64_t)ftell(f);
    to = std::min(to, sz);
    from = std::min(from, to);
    buf.resize(to - from);
    fseek(f, (long)from, SEEK_SET);
    const bool ok = buf.empty() || fread(buf.data(), 1, buf.size(), f) == buf.size();
    fclose(f);
    if (!ok) g_bam_error = "short read";
    return ok;
}

static void* bam_open_impl(const char* path, int threads, bool ranged, uint64_t voff_beg, uint64_t voff_end)
{
    g_bam_error.clear();
    if (threads <= 0) threads = (int)std::max(1u, std::min(64u, std::thread::hardware_concurrency()));
    Bam* b = new Bam();
    std::vector<uint8_t> file;
    uint64_t rec_begin = 0, rec_end = ~0ull;      // record byte range inside b->raw (ranged mode)
    if (!ranged) {
        if (!read_bytes(path, 0, ~0ull, file)) { delete b; return nullptr; }
        if (!inflate_file(file, threads, b->raw, b->error)) { g_bam_error = b->error; delete b; return nullptr; }
    } else {
        // header: inflate leading blocks until the reference dictionary is complete (grow the window if needed)
        std::vector<uint8_t> head;
        for (uint64_t want = 1 << 20;; want *= 4) {
            if (!read_bytes(path, 0, want, file)) { delete b; return nullptr; }
            // drop a trailing partial block: keep only whole blocks
            head.clear();
            std::string err;
            uint64_t p = 0;
            while (p + 18 <= file.size()) {
                if (!(file[p] == 0x1f && file[p + 1] == 0x8b)) break;
                const uint32_t xlen = rd16(&file[p + 10]);
                if (p + 12 + xlen > file.size()) break;
                uint32_t bsize = 0;
                for (uint64_t q = p + 12; q + 4 <= p + 12 + xlen;) {
                    const uint32_t slen = rd16(&file[q + 2]);
                    if (file[q] == 'B' && file[q + 1] == 'C') bsize = rd16(&file[q + 4]);
                    q += 4 + slen;
                }
                if (!bsize || p + bsize + 1 > file.size()) break;
                p += bsize + 1;
            }
            if (!inflate_file(file, 1, head, err, 0, p)) { g_bam_error = err; delete b; return nullptr; }
            bool complete = false;
            if (head.size() >= 12 && memcmp(head.data(), "BAM\1", 4) == 0) {
                uint64_t q = 8 + rd32(&head[4]);
                if (q + 4 <= head.size()) {
                    const uint32_t n_ref = rd32(&head[q]); q += 4;
                    uint32_t i = 0;
                    for (; i < n_ref && q + 4 <= head.size(); ++i) {
                        const uint32_t l_name = rd32(&head[q]);
                        if (q + 8 + l_name > head.size()) break;
                        q += 8 + l_name;
                    }
                    complete = i == n_ref;
                }
            } else if (head.size() >= 4) { g_bam_error = "not a BAM file"; delete b; return nullptr; }
            if (complete || want > file.size() * 2 + (1 << 20)) break;     // whole file read already
        }
        // records: the compressed range [coffset(voff_beg), block after coffset(voff_end)]
        std::vector<uint8_t> body;
        if (voff_end > voff_beg) {
            const uint64_t c0 = voff_beg >> 16, c1 = voff_end >> 16;
            if (!read_bytes(path, c0, c1 + 65536 + 26, file)) { delete b; return nullptr; }
            std::string err;
            // inflate whole blocks from c0 up to and including the block that starts at c1
            uint64_t p = 0, stop = 0;
            std::vector<uint64_t> starts;
            while (p + 18 <= file.size()) {
                if (!(file[p] == 0x1f && file[p + 1] == 0x8b)) break;
                const uint32_t xlen = rd16(&file[p + 10]);
                uint32_t bsize = 0;
                for (uint64_t q = p + 12; q + 4 <= p + 12 + xlen && q + 6 <= file.size();) {
                    const uint32_t slen = rd16(&file[q + 2]);
                    if (file[q] == 'B' && file[q + 1] == 'C') bsize = rd16(&file[q + 4]);
                    q += 4 + slen;
                }
                if (!bsize || p + bsize + 1 > file.size()) break;
                starts.push_back(p);
                p += bsize + 1;
                stop = p;
                if (starts.back() + c0 >= c1) break;
            }
            if (!inflate_file(file, threads, body, err, 0, stop)) { g_bam_error = err; delete b; return nullptr; }
            // uncompressed offset of the last block = sum of the isizes before it
            uint64_t last_u = 0;
            for (size_t i = 0; i + 1 < starts.size(); ++i) {
                const uint64_t end = (i + 1 < starts.size()) ? starts[i + 1] : stop;
                last_u += rd32(&file[end - 4]);
            }
            rec_begin = voff_beg & 0xffff;
            rec_end = (starts.empty() || starts.back() + c0 < c1) ? body.size() : last_u + (voff_end & 0xffff);
        }
        // stitch: header bytes followed by the record range so that the common parser below applies
        uint64_t hdr_end = 8 + rd32(&head[4]);
        const uint32_t n_ref = rd32(&head[hdr_end]); hdr_end += 4;
        for (uint32_t i = 0; i < n_ref; ++i) hdr_end += 8 + rd32(&head[hdr_end]);
        b->raw.assign(head.begin(), head.begin() + hdr_end);
        if (!body.empty() && rec_end > rec_begin) b->raw.insert(b->raw.end(), body.begin() + rec_begin, body.begin() + std::min<uint64_t>(rec_end, body.size()));
    }
    const std::vector<uint8_t>& r = b->raw;
    if (r.size() < 12 || memcmp(r.data(), "BAM\1", 4) != 0) { g_bam_error = "not a BAM file"; delete b; return nullptr; }
    uint64_t p = 4;
    const uint32_t l_text = rd32(&r[p]); p += 4;
    b->header_text.assign(reinterpret_cast<const char*>(&r[p]), strnlen(reinterpret_cast<const char*>(&r[p]), l_text));
    p += l_text;
    const uint32_t n_ref = rd32(&r[p]); p += 4;
    for (uint32_t i = 0; i < n_ref; ++i) {
        const uint32_t l_name = rd32(&r[p]);
        b->ref_names.emplace_back(reinterpret_cast<const char*>(&r[p + 4]), l_name ? l_name - 1 : 0);
        b->ref_lens.push_back((int32_t)rd32(&r[p + 4 + l_name]));
        p += 8 + l_name;
    }
    // chain the record offsets, build the CIGAR CSR and the first-occurrence QNAME ids
    std::unordered_map<std::string_view, int32_t> seen;
    b->cig_off.push_back(0);
    while (p + 4 <= r.size()) {
        const uint32_t bs = rd32(&r[p]);
        if (p + 4 + bs > r.size() || bs < 32) { g_bam_error = "truncated BAM record"; delete b; return nullptr; }
        b->rec_off.push_back(p);
        const uint8_t* rec = &r[p + 4];
        const uint32_t l_name = rec[8];
        uint32_t n_cig = rd16(rec + 12);
        uint64_t src = p + 4 + 32 + l_name;
        if (n_cig == 2) {                                     // possible CG-tag placeholder
            const uint32_t w0 = rd32(&r[src]), w1 = rd32(&r[src + 4]);
            const uint32_t l_seq = rd32(rec + 16);
            if ((w0 & 15) == 4 && (w0 >> 4) == l_seq && (w1 & 15) == 3) {
                uint32_t cnt = 0;
                const uint64_t aux = src + 8 + (l_seq + 1) / 2 + l_seq;
                const uint64_t at = find_long_cigar(r, aux, p + 4 + bs, &cnt);
                if (at) { src = at; n_cig = cnt; }
            }
        }
        b->cig_src.push_back(src);
        b->cig_off.push_back(b->cig_off.back() + n_cig);
        std::string_view nm(reinterpret_cast<const char*>(rec + 32), l_name ? l_name - 1 : 0);
        auto it = seen.find(nm);
        if (it == seen.end()) {
            it = seen.emplace(nm, (int32_t)b->name_first.size()).first;
            b->name_first.push_back((uint32_t)(b->rec_off.size() - 1));
            b->names_bytes += nm.size() + 1;
        }
        b->name_id.push_back(it->second);
        p += 4 + bs;
    }
    return b;
}

void* svx_bam_open(const char* path, int threads) { return bam_open_impl(path, threads, false, 0, 0); }

// Only the records between two BGZF virtual offsets (from the .bai index: one chromosome, one rank's shard);
// the header / reference dictionary is always decoded.  voff_end <= voff_beg: header only.
void* svx_bam_open_range(const char* path, int threads, uint64_t voff_beg, uint64_t voff_end)
{
    return bam_open_impl(path, threads, true, voff_beg, voff_end);
}

const char* svx_bam_error(void) { return g_bam_error.c_str(); }

// sizes: [n_records, n_cigar_words, n_refs, n_names, names_bytes, header_bytes, ref_names_bytes, raw_bytes]
void svx_bam_sizes(void* h, uint64_t* sizes)
{
    const Bam* b = static_cast<const Bam*>(h);
    uint64_t rn = 0;
    for (auto& s : b->ref_names) rn += s.size() + 1;
    sizes[0] = b->rec_off.size(); sizes[1] = b->cig_off.back(); sizes[2] = b->ref_names.size();
    sizes[3] = b->name_first.size(); sizes[4] = b->names_bytes; sizes[5] = b->header_text.size();
    sizes[6] = rn; sizes[7] = b->raw.size();
}

// Fill caller-owned arrays (all sized from svx_bam_sizes).  seq_off may be NULL; otherwise it receives the byte
// offset of each record's 4-bit SEQ inside the decompressed file, which svx_bam_raw() exposes.
void svx_bam_export(void* h, int threads, int32_t* tid, int32_t* pos, uint16_t* flag, uint8_t* mapq, int32_t* l_seq,
                    int32_t* name_id, int64_t* cig_off, uint32_t* cigar, char* names, char* header, char* ref_names,
                    int32_t* ref_lens, int64_t* seq_off)
{
    const Bam* b = static_cast<const Bam*>(h);
    const std::vector<uint8_t>& r = b->raw;
    const size_t n = b->rec_off.size();
    if (threads <= 0) threads = (int)std::max(1u, std::min(64u, std::thread::hardware_concurrency()));
    parallel_for(n, threads, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            const uint8_t* rec = &r[b->rec_off[i] + 4];
            tid[i] = (int32_t)rd32(rec); pos[i] = (int32_t)rd32(rec + 4);
            const uint32_t l_name = rec[8];
            mapq[i] = rec[9];
            const uint32_t n_cig_field = rd16(rec + 12);
            const uint64_t n_cig = b->cig_off[i + 1] - b->cig_off[i];
            flag[i] = rd16(rec + 14);
            l_seq[i] = (int32_t)rd32(rec + 16);
            name_id[i] = b->name_id[i];
            cig_off[i] = (int64_t)b->cig_off[i];
            memcpy(cigar + b->cig_off[i], &r[b->cig_src[i]], 4ull * n_cig);
            if (seq_off) seq_off[i] = (int64_t)(b->rec_off[i] + 4 + 32 + l_name + 4ull * n_cig_field);
        }
    });
    cig_off[n] = (int64_t)b->cig_off[n];
    char* w = names;
    for (uint32_t first : b->name_first) {
        const uint8_t* rec = &r[b->rec_off[first] + 4];
        const uint32_t l = rec[8] ? rec[8] - 1 : 0;
        memcpy(w, rec + 32, l); w[l] = '\n'; w += l + 1;
    }
    memcpy(header, b->header_text.data(), b->header_text.size());
    w = ref_names;
    for (size_t i = 0; i < b->ref_names.size(); ++i) {
        memcpy(w, b->ref_names[i].data(), b->ref_names[i].size()); w[b->ref_names[i].size()] = '\n';
        w += b->ref_names[i].size() + 1;
        ref_lens[i] = b->ref_lens[i];
    }
}

const uint8_t* svx_bam_raw(void* h) { return static_cast<const Bam*>(h)->raw.data(); }

void svx_bam_close(void* h) { delete static_cast<Bam*>(h); }

}  // extern "C"
