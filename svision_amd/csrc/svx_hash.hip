// svx_hash.hip -- k-mer seed-and-extend of the --hash re-aligner on gfx950 (MI355X), integer-exact.
//
// Reference: HashAligner.makePairwiseAlignment / extendKmersForward / extendKmersReverse
// (src/segmentplot/hash_aligner.py:145-239, :37-99) as driven by hashplot_unmapped
// (src/segmentplot/run_hash_lineplot.py:52-85): the reference window `y` is first aligned with itself
// (k-mers of y that occur twice or more among y's own k-mers on either strand become "avoid" k-mers), then the
// unmapped read piece `x` is placed on y: every k-mer of y that is not avoided is looked up among x's k-mers
// (both strands), every hit whose previous base does not match too is extended without gaps until the first
// mismatch (which is counted into the length), an 'N', or the last-but-one base, and hits of at least
// `window` bases are kept.  What comes after -- the filter against the window's self repeats, the greedy
// merge of collinear hits, the choice of the longest -- is a few list operations on a handful of hits and stays on
// the host (svision_amd/segmentplot/hash_aligner.py).
//
// One workgroup per job (jobs are independent: one per unmapped piece / long insertion of a read), bases packed
// as 4-bit symbols (0..4 = A C G T N, 5..9 = a c g t n of soft-masked references, 10..14 = R Y K M S; upstream's
// k-mers are the raw strings, so symbols only match themselves and only upper-case ACGT have a complement other
// than N), k-mers as 4 bits per base in 64 bits (k <= 13):
//   pass A  y's k-mers of both strands are counted in an open-addressing table in global scratch (atomicCAS claim,
//           atomicAdd count); a y position whose k-mer has count >= 2 is avoided, one with count 1 is seeded against
//           its single occurrence (the self pass of the reference: its hits only matter if they are off the diagonal);
//   pass B  x's k-mers (both strands, < 2 x 2048 of them) are sorted in LDS by (k-mer, insertion order) -- the order
//           of the reference's per-k-mer position lists -- and every non-avoided y position binary-searches them.
// Hits are written in the reference's loop order (y position ascending, then list order): count -> workgroup prefix ->
// emit, no atomics on the output, so the host can replay them through the order-dependent merge.
// Quirks kept: the k-mer loops stop at len - (k + 1); the extension never reads the last base; the seed rule looks at
// the base before the k-mer on both sequences.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/svx.h"

namespace {

constexpr int BLOCK = 256;
constexpr int MAX_X = 2048;                       // longest piece handled on the device (--max_hash_len default: 1000); 32 KB of LDS
constexpr unsigned long long EMPTY = ~0ull;

struct Job {                                      // mirrors SvxHashJob (include/svx.h)
    uint64_t x_off, y_off;                        // byte offsets into the packed base array
    uint32_t x_len, y_len;
    uint64_t table_off;                           // first slot of this job's scratch table (slots of 16 bytes)
    uint32_t table_slots;                         // power of two >= 4 * (2 * y_len)
    uint32_t hit_cap;                             // capacity of each of the job's two hit lists
    uint64_t hit_off;                             // first hit record of the job (A list, then B list)
};

__device__ inline int comp(int c) { return c < 4 ? 3 - c : 4; }          // A<->T, C<->G, N stays N (classes.py:21-39)

// base i of the reverse complement of s[0..len)
__device__ inline int rc_at(const uint8_t* s, int len, int i) { return comp(s[len - 1 - i]); }

template <bool REV>
__device__ inline int base_at(const uint8_t* s, int len, int i) { return REV ? rc_at(s, len, i) : s[i]; }

template <bool REV>
__device__ inline unsigned long long kmer_code(const uint8_t* s, int len, int i, int k)
{
    unsigned long long c = 0;
    for (int j = 0; j < k; ++j) c = (c << 4) | (unsigned long long)base_at<REV>(s, len, i + j);
    return c;
}

// ungapped extension (:37-62 / :81-99), mismatchNum = 0: returns the match length (the first mismatch is counted)
template <bool REV>
__device__ inline int extend(const uint8_t* x, int xl, const uint8_t* y, int yl, int xp, int yp, int k)
{
    int n = k;
    while (true) {
        if (xp + n >= xl - 1 || yp + n >= yl - 1) break;
        const int a = base_at<REV>(x, xl, xp + n), b = y[yp + n];
        if (a == 4 || b == 4) break;
        ++n;
        if (a != b) break;
    }
    return n;
}

// one table entry against y position i: -> match length, or 0 when the seed is skipped (:196,204)
__device__ inline int seed(const uint8_t* x, int xl, const uint8_t* y, int yl, int pos, int i, int k)
{
    if (pos >= 0) {
        if (pos > 0 && i > 0 && x[pos - 1] == y[i - 1]) return 0;
        return extend<false>(x, xl, y, yl, pos, i, k);
    }
    const int rp = -1 - pos;
    if (rp > 0 && i > 0 && rc_at(x, xl, rp - 1) == y[i - 1]) return 0;
    return extend<true>(x, xl, y, yl, rp, i, k);
}

__device__ inline uint32_t hash_slot(unsigned long long code, uint32_t mask) { return (uint32_t)((code * 0x9E3779B97F4A7C15ull) >> 32) & mask; }

// exclusive prefix over the workgroup's per-thread counts; returns the total
__device__ inline uint32_t block_scan(uint32_t v, uint32_t* warp_sums, uint32_t& before)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t incl = v;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
    if (lane == 63) warp_sums[wv] = incl;
    __syncthreads();
    uint32_t base = 0, total = 0;
    for (int w = 0; w < BLOCK / 64; ++w) { if (w < wv) base += warp_sums[w]; total += warp_sums[w]; }
    __syncthreads();
    before = base + incl - v;
    return total;
}

__global__ __launch_bounds__(BLOCK)
void hash_seeds_kernel(const uint8_t* __restrict__ bases, const Job* __restrict__ jobs, unsigned long long* __restrict__ table,
                       int32_t* __restrict__ hits, uint32_t* __restrict__ counts, int k, int window)
{
    __shared__ unsigned long long xs[2 * MAX_X];          // pass B: x's k-mers, (code << 12 | insertion order)
    __shared__ uint32_t warp_sums[BLOCK / 64];
    const Job job = jobs[blockIdx.x];
    const uint8_t* x = bases + job.x_off;
    const uint8_t* y = bases + job.y_off;
    const int xl = (int)job.x_len, yl = (int)job.y_len;
    const int tid = threadIdx.x;
    const int ny = yl - (k + 1) > 0 ? yl - (k + 1) : 0;   // k-mer positions of y on each strand (:150,162,176)
    const int nx = xl - (k + 1) > 0 ? xl - (k + 1) : 0;
    unsigned long long* tab = table + job.table_off * 2;      // slot = {k-mer, count | position << 32}
    const uint32_t mask = job.table_slots - 1;
    int32_t* hits_a = hits + job.hit_off * 4;
    int32_t* hits_b = hits_a + (size_t)job.hit_cap * 4;

    // ---- pass A: count y's own k-mers (both strands) ----
    for (uint32_t s = tid; s < job.table_slots; s += BLOCK) { tab[2 * s] = EMPTY; tab[2 * s + 1] = 0; }
    __syncthreads();
    for (int e = tid; e < 2 * ny; e += BLOCK) {
        const bool rev = e >= ny;
        const int i = rev ? e - ny : e;
        const unsigned long long code = rev ? kmer_code<true>(y, yl, i, k) : kmer_code<false>(y, yl, i, k);
        uint32_t s = hash_slot(code, mask);
        while (true) {
            const unsigned long long old = atomicCAS(&tab[2 * s], EMPTY, code);
            if (old == EMPTY || old == code) break;
            s = (s + 1) & mask;
        }
        // count in the low word; the position in the high word is meaningful only while the count stays 1
        atomicAdd(&tab[2 * s + 1], 1ull + ((unsigned long long)(uint32_t)(rev ? -1 - i : i) << 32));
    }
    __syncthreads();

    // ---- pass B table: x's k-mers sorted by (code, insertion order) in LDS ----
    const int nt = 2 * nx;
    int np2 = 1;
    while (np2 < nt) np2 <<= 1;
    for (int e = tid; e < np2; e += BLOCK) {
        unsigned long long v = ~0ull;
        if (e < nt) {
            const bool rev = e >= nx;
            const int i = rev ? e - nx : e;
            const unsigned long long code = rev ? kmer_code<true>(x, xl, i, k) : kmer_code<false>(x, xl, i, k);
            v = (code << 12) | (unsigned)e;
        }
        xs[e] = v;
    }
    __syncthreads();
    for (int size = 2; size <= np2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int e = tid; e < np2; e += BLOCK) {
                const int p = e ^ stride;
                if (p > e) {
                    const unsigned long long a = xs[e], b = xs[p];
                    const bool up = (e & size) == 0;
                    if ((a > b) == up) { xs[e] = b; xs[p] = a; }
                }
            }
            __syncthreads();
        }

    // ---- seeds, in the reference's order: y position ascending, then table order; count -> prefix -> emit ----
    uint32_t total_a = 0, total_b = 0;
    for (int chunk = 0; chunk < ny; chunk += BLOCK) {
        const int i = chunk + tid;
        uint32_t ca = 0, cb = 0;
        int pos_a = 0, len_a = 0, lo = 0, hi = 0;
        if (i < ny) {
            const unsigned long long code = kmer_code<false>(y, yl, i, k);
            uint32_t s = hash_slot(code, mask);
            while (tab[2 * s] != code) s = (s + 1) & mask;                       // y's own k-mer: always present
            const uint32_t cnt = (uint32_t)tab[2 * s + 1];
            if (cnt == 1) {                                                      // self pass: the k-mer's only occurrence
                pos_a = (int)(uint32_t)(tab[2 * s + 1] >> 32);
                len_a = seed(y, yl, y, yl, pos_a, i, k);
                ca = len_a >= window ? 1u : 0u;
            }
            if (cnt < 2) {                                                       // not an "avoid" k-mer: look it up in x
                const unsigned long long key = code << 12;
                int a = 0, b = nt;
                while (a < b) { const int m = (a + b) >> 1; if (xs[m] < key) a = m + 1; else b = m; }
                lo = a;
                hi = a;
                while (hi < nt && (xs[hi] >> 12) == code) ++hi;
                for (int e = lo; e < hi; ++e) {
                    const int ord = (int)(xs[e] & 0xfffu);
                    const int pos = ord >= nx ? -1 - (ord - nx) : ord;
                    cb += seed(x, xl, y, yl, pos, i, k) >= window ? 1u : 0u;
                }
            }
        }
        uint32_t before_a, before_b;
        const uint32_t sum_a = block_scan(ca, warp_sums, before_a);
        const uint32_t sum_b = block_scan(cb, warp_sums, before_b);
        if (ca && total_a + before_a < job.hit_cap) {
            int32_t* h = hits_a + (size_t)(total_a + before_a) * 4;
            h[0] = i; h[1] = pos_a >= 0 ? pos_a : -1 - pos_a; h[2] = len_a; h[3] = pos_a >= 0 ? 1 : 0;
        }
        if (cb) {
            uint32_t slot = total_b + before_b;
            for (int e = lo; e < hi; ++e) {
                const int ord = (int)(xs[e] & 0xfffu);
                const int pos = ord >= nx ? -1 - (ord - nx) : ord;
                const int n = seed(x, xl, y, yl, pos, i, k);
                if (n >= window) {
                    if (slot < job.hit_cap) {
                        int32_t* h = hits_b + (size_t)slot * 4;
                        h[0] = i; h[1] = pos >= 0 ? pos : -1 - pos; h[2] = n; h[3] = pos >= 0 ? 1 : 0;
                    }
                    ++slot;
                }
            }
        }
        total_a += sum_a;
        total_b += sum_b;
    }
    if (tid == 0) { counts[2 * blockIdx.x] = total_a; counts[2 * blockIdx.x + 1] = total_b; }
}

}  // namespace

extern "C" int svx_hash_seeds(const uint8_t* d_bases, const SvxHashJob* d_jobs, uint32_t n_jobs, uint64_t* d_table,
                              int32_t* d_hits, uint32_t* d_counts, uint32_t k, uint32_t window, uint32_t max_x_len, void* stream)
{
    if (n_jobs == 0) return SVX_OK;
    if (!d_bases || !d_jobs || !d_table || !d_hits || !d_counts) return SVX_EINVAL;
    if (k < 2 || k > 13 || max_x_len > (uint32_t)MAX_X) return SVX_EINVAL;          // 4 bits x k + 12 order bits in 64; LDS table
    static_assert(sizeof(SvxHashJob) == sizeof(Job), "SvxHashJob layout");
    hipLaunchKernelGGL(hash_seeds_kernel, dim3(n_jobs), dim3(BLOCK), 0, static_cast<hipStream_t>(stream), d_bases,
                       reinterpret_cast<const Job*>(d_jobs), reinterpret_cast<unsigned long long*>(d_table), d_hits, d_counts, (int)k, (int)window);
    return hipGetLastError() == hipSuccess ? SVX_OK : SVX_ELAUNCH;
}
