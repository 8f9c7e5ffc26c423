// svx_bamdev.hip -- BAM records -> packed structure of arrays, on the device (gfx950).
//
// Second half of the device-side ingestion (first half: svx_inflate.hip).  The inflated record stream of a chromosome
// sits in HBM; what the hot path needs from it is ~3 % of its bytes: the fixed fields, the QNAME and the CIGAR words of
// every record (SAMv1 4.2) -- the packed arrays svx_cigar_scan consumes, which therefore never cross the PCIe bus.
//
// Records are chained by their block_size fields, a serial walk.  The .bai linear index breaks the chain: it stores the
// virtual offset of the first record overlapping every 16 kb of the reference (SAMv1 5.1.3), i.e. thousands of known
// record starts per chromosome.  One lane (pass 1) / one wave (pass 2) walks from one such start to the next (a few dozen records):
//
//   pass 1  svx_bam_walk_count   per start: records, CIGAR words, QNAME bytes; the walk must END exactly on the next start
//   (host: exclusive prefix sums -> where every lane writes)
//   pass 2  svx_bam_walk_extract a lane per start notes where every record begins and where its words / name go, then a WAVE PER
//           RECORD copies: tid / pos / flag / mapq / l_seq, CIGAR words, QNAMEs ('\n'-separated)
//
// Integer exact: the arrays equal the host decoder's (svx_bam.cpp) element for element (tests/test_gpu_inflate.py).
// Records with a CG:B,I long CIGAR (> 65535 operations) are followed into their optional fields (find_cg_tag).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/svx.h"

namespace {

constexpr int BLOCK = 64;

__device__ __forceinline__ uint32_t ld32(const uint8_t* p)    // unaligned little-endian load
{
    return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24;
}

// A CIGAR of more than 65,535 operations sits in the optional field CG:B,I and the record's own CIGAR is the placeholder
// "<l_seq>S<span>N" (SAMv1 4.2.2).  -> the tag's words (and their number), or nullptr when the record has no such field.
__device__ inline const uint8_t* find_cg_tag(const uint8_t* rec, uint32_t bs, uint32_t l_name, uint32_t n_cig, uint32_t l_seq, uint32_t* count)
{
    const uint8_t* q = rec + 32 + l_name + 4ull * n_cig + (l_seq + 1ull) / 2 + l_seq;
    const uint8_t* end = rec + bs;
    while (q + 3 <= end) {
        const uint8_t t0 = q[0], t1 = q[1], type = q[2];
        q += 3;
        if (type == 'B') {
            if (q + 5 > end) return nullptr;
            const uint8_t sub = q[0];
            const uint32_t n = ld32(q + 1);
            const uint32_t width = sub == 'c' || sub == 'C' ? 1u : sub == 's' || sub == 'S' ? 2u : sub == 'i' || sub == 'I' || sub == 'f' ? 4u : 0u;
            if (width == 0u || (uint64_t)(end - (q + 5)) < (uint64_t)n * width) return nullptr;
            if (t0 == 'C' && t1 == 'G' && sub == 'I') { *count = n; return q + 5; }
            q += 5 + (uint64_t)n * width;
        } else if (type == 'Z' || type == 'H') {
            while (q < end && *q) ++q;
            ++q;
        } else {
            const uint32_t width = type == 'A' || type == 'c' || type == 'C' ? 1u : type == 's' || type == 'S' ? 2u : type == 'i' || type == 'I' || type == 'f' ? 4u : 0u;
            if (width == 0u) return nullptr;
            q += width;
        }
    }
    return nullptr;
}

__device__ inline bool is_cg_placeholder(const uint8_t* rec, uint32_t l_name, uint32_t n_cig, uint32_t l_seq)
{
    if (n_cig != 2) return false;
    const uint32_t w0 = ld32(rec + 32 + l_name), w1 = ld32(rec + 36 + l_name);
    return (w0 & 15) == 4 && (w0 >> 4) == l_seq && (w1 & 15) == 3;
}

// counts per start: [0] records, [1] CIGAR words, [2] QNAME bytes (incl. one separator per record), [3] status
// status: 0 ok, 1 walk does not end on the next start (index does not match the data), 2 malformed record
__global__ __launch_bounds__(BLOCK)
void bam_walk_count_kernel(const uint8_t* __restrict__ raw, const uint64_t* __restrict__ starts, uint32_t n_starts,
                           uint64_t* __restrict__ counts)
{
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n_starts) return;
    uint64_t p = starts[i];
    const uint64_t end = starts[i + 1];
    uint64_t n = 0, words = 0, name_bytes = 0, status = 0;
    while (p < end) {
        if (p + 36 > end) { status = 2; break; }
        const uint32_t bs = ld32(raw + p);
        const uint8_t* rec = raw + p + 4;
        const uint32_t l_name = rec[8], n_cig = (uint32_t)rec[12] | (uint32_t)rec[13] << 8, l_seq = ld32(rec + 16);
        if (bs < 32 || p + 4 + bs > end || 32ull + l_name + 4ull * n_cig + (l_seq + 1ull) / 2 + l_seq > bs) { status = 2; break; }
        uint32_t n_words = n_cig;
        if (is_cg_placeholder(rec, l_name, n_cig, l_seq)) {     // "<l_seq>S<span>N": the real CIGAR is in the CG tag
            if (find_cg_tag(rec, bs, l_name, n_cig, l_seq, &n_words) == nullptr) n_words = n_cig;     // (a look-alike without the tag is what it says: two operations, as in the host reader)
        }
        ++n;
        words += n_words;
        name_bytes += l_name ? l_name : 1;                      // l_name counts the NUL: the bytes + one separator
        p += 4 + bs;
    }
    if (status == 0 && p != end) status = 1;
    counts[4ull * i + 0] = n;
    counts[4ull * i + 1] = words;
    counts[4ull * i + 2] = name_bytes;
    counts[4ull * i + 3] = status;
}

// Pass 2, in two launches (round 6).  Until then ONE wave walked a start's whole record chain and copied as it went: fine for
// HiFi (thirty records of 150 words between two index entries, 0.1 ms per chromosome), but on ONT-shaped data -- reads of 50 kb
// with 5,000 words each, spanning so many 16 kb bins that most index entries repeat -- a wave had dozens of 20 KB copies to do
// one after the other: 14.7 ms per slice of the ONT stand-in, on the critical path of its hand-over
// (profiles/r06_ont_kernel_stats.csv).  Now the chain is walked once more by a LANE per start, which only notes where every
// record begins and where its words and its name go (index kernel), and the copies are made by a WAVE PER RECORD (copy kernel).
//
// base per start (exclusive prefix sums of the counts): [0] first record, [1] first CIGAR word, [2] first QNAME byte.
// The record's byte offset travels from the first kernel to the second in the record's own tid / pos entries (two 32-bit halves),
// which the second kernel then overwrites with the fields themselves: no scratch array, the caller's buffers as they were.
__global__ __launch_bounds__(BLOCK)
void bam_walk_index_kernel(const uint8_t* __restrict__ raw, const uint64_t* __restrict__ starts, uint32_t n_starts,
                           const uint64_t* __restrict__ base, int32_t* __restrict__ tid, int32_t* __restrict__ pos,
                           int64_t* __restrict__ cig_off, int64_t* __restrict__ name_off)
{
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n_starts) return;
    uint64_t p = starts[i];
    const uint64_t end = starts[i + 1];
    uint64_t k = base[3ull * i + 0], w = base[3ull * i + 1], nb = base[3ull * i + 2];
    while (p < end) {                                           // (the count pass has checked the chain: it ends on `end`)
        const uint32_t bs = ld32(raw + p);
        const uint8_t* rec = raw + p + 4;
        const uint32_t l_name = rec[8], n_cig = (uint32_t)rec[12] | (uint32_t)rec[13] << 8, l_seq = ld32(rec + 16);
        uint32_t n_words = n_cig;
        if (is_cg_placeholder(rec, l_name, n_cig, l_seq)) {
            if (find_cg_tag(rec, bs, l_name, n_cig, l_seq, &n_words) == nullptr) n_words = n_cig;
        }
        tid[k] = (int32_t)(uint32_t)p;
        pos[k] = (int32_t)(uint32_t)(p >> 32);
        cig_off[k] = (int64_t)w;
        name_off[k] = (int64_t)nb;
        w += n_words;
        nb += l_name ? l_name : 1;
        ++k;
        p += 4ull + bs;
    }
    // the closing entries of the two CSR arrays (the totals), by the last interval's lane: the caller needed two fill launches
    // per chromosome for them, each a few hundred microseconds of waiting for room next to the inflate and the CNN
    if (i == n_starts - 1) { cig_off[k] = (int64_t)w; name_off[k] = (int64_t)nb; }
}

// One WAVE per record: lane 0 writes the fixed fields, the lanes share the copies -- QNAME bytes 64 at a time, CIGAR words 64 at a
// time, each from the two aligned dwords around it (a record starts at any byte).
__global__ __launch_bounds__(BLOCK)
void bam_walk_copy_kernel(const uint8_t* __restrict__ raw, uint32_t n_records, int32_t* __restrict__ tid, int32_t* __restrict__ pos,
                          uint16_t* __restrict__ flag, uint8_t* __restrict__ mapq, int32_t* __restrict__ l_seq_out,
                          const int64_t* __restrict__ cig_off, uint32_t* __restrict__ cigar, const int64_t* __restrict__ name_off,
                          uint8_t* __restrict__ names)
{
    const uint32_t k = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    if (k >= n_records) return;
    // (every lane reads the two halves before lane 0 overwrites them below: one wave, program order)
    const uint32_t p_lo = (uint32_t)__builtin_amdgcn_readfirstlane(tid[k]), p_hi = (uint32_t)__builtin_amdgcn_readfirstlane(pos[k]);
    const uint64_t p = (uint64_t)p_hi << 32 | p_lo;
    const uint64_t w = (uint64_t)cig_off[k], nb = (uint64_t)name_off[k];
    const uint8_t* rec = raw + p + 4;
    const uint32_t bs = (uint32_t)__builtin_amdgcn_readfirstlane((int)ld32(raw + p));
    const uint32_t l_name = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec[8]);
    uint32_t n_cig = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)rec[12] | (uint32_t)rec[13] << 8));
    const uint32_t l_seq = (uint32_t)__builtin_amdgcn_readfirstlane((int)ld32(rec + 16));
    const int32_t f_tid = (int32_t)ld32(rec), f_pos = (int32_t)ld32(rec + 4);
    if (lane == 0) {
        tid[k] = f_tid;
        pos[k] = f_pos;
        mapq[k] = rec[9];
        flag[k] = (uint16_t)((uint32_t)rec[14] | (uint32_t)rec[15] << 8);
        l_seq_out[k] = (int32_t)l_seq;
    }
    const uint8_t* nm = rec + 32;
    const uint32_t nn = l_name ? l_name - 1 : 0;
    for (uint32_t j = lane; j < nn; j += BLOCK) names[nb + j] = nm[j];
    if (lane == 0) names[nb + nn] = '\n';
    const uint8_t* cg = rec + 32 + l_name;
    if (is_cg_placeholder(rec, l_name, n_cig, l_seq)) {         // (uniform: every lane walks the same optional fields)
        uint32_t n_real = 0;
        const uint8_t* real = find_cg_tag(rec, bs, l_name, n_cig, l_seq, &n_real);
        if (real != nullptr) { cg = real; n_cig = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_real); }
    }
    // word j = bytes [4 j, 4 j + 4) behind cg: the aligned dwords a[j], a[j + 1] around it, shifted (raw is 16-byte aligned --
    // a torch allocation --, padded behind its end by the inflate's output slack: the dword behind the last word is readable)
    const uintptr_t addr = reinterpret_cast<uintptr_t>(cg);
    const uint32_t* __restrict__ al = reinterpret_cast<const uint32_t*>(addr & ~(uintptr_t)3);
    const uint32_t sh = (uint32_t)(addr & 3u) * 8u;
    for (uint32_t j = lane; j < n_cig; j += BLOCK) {
        const uint32_t lo = al[j];
        uint32_t v = lo;
        if (sh) v = (lo >> sh) | (al[j + 1] << (32u - sh));
        cigar[w + j] = v;
    }
}

}  // namespace

// pass 1: d_starts [n_starts + 1] byte offsets into d_raw (the last entry = end of the part); d_counts [n_starts][4]
extern "C" int svx_bam_walk_count(const uint8_t* d_raw, const uint64_t* d_starts, uint32_t n_starts, uint64_t* d_counts, void* stream)
{
    if (n_starts == 0) return SVX_OK;
    if (!d_raw || !d_starts || !d_counts) return SVX_EINVAL;
    hipLaunchKernelGGL(bam_walk_count_kernel, dim3((n_starts + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, static_cast<hipStream_t>(stream),
                       d_raw, d_starts, n_starts, d_counts);
    return hipGetLastError() == hipSuccess ? SVX_OK : SVX_ELAUNCH;
}

// pass 2: d_base [n_starts][3] = exclusive prefix sums of the first three counts; the output arrays are sized by their
// totals (d_cig_off / d_name_off: one entry per record + the closing entry = the totals, written by the last interval's wave)
extern "C" int svx_bam_walk_extract(const uint8_t* d_raw, const uint64_t* d_starts, uint32_t n_starts, const uint64_t* d_base,
                                    int32_t* d_tid, int32_t* d_pos, uint16_t* d_flag, uint8_t* d_mapq, int32_t* d_l_seq,
                                    int64_t* d_cig_off, uint32_t* d_cigar, int64_t* d_name_off, uint8_t* d_names, uint32_t n_records,
                                    void* stream)
{
    if (n_starts == 0) return SVX_OK;
    if (!d_raw || !d_starts || !d_base || !d_tid || !d_pos || !d_flag || !d_mapq || !d_l_seq || !d_cig_off || !d_cigar || !d_name_off || !d_names)
        return SVX_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(bam_walk_index_kernel, dim3((n_starts + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st,
                       d_raw, d_starts, n_starts, d_base, d_tid, d_pos, d_cig_off, d_name_off);
    if (n_records)
        hipLaunchKernelGGL(bam_walk_copy_kernel, dim3(n_records), dim3(BLOCK), 0, st,
                           d_raw, n_records, d_tid, d_pos, d_flag, d_mapq, d_l_seq, d_cig_off, d_cigar, d_name_off, d_names);
    return hipGetLastError() == hipSuccess ? SVX_OK : SVX_ELAUNCH;
}
