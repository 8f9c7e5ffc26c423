// svx_cigar.hip -- per-alignment CIGAR / segment scan for gfx950 (MI355X).
//
// Replaces the Python CIGAR walk of analyze_inside_align
// (reference src/collection/analyze_reads.py:828-853) and the pysam-derived
// reference_end / query_alignment_start / query_alignment_end for a whole
// batch of alignments whose packed BAM CIGAR words already sit in HBM.
//
// HBM-read bound: 4 B per CIGAR op, 16 B per alignment of CSR/start data,
// 24 B written per long gap (rare).  Two data passes so that the output is
// deterministic and sorted by (alignment, op) without a sort:
//   1. count_kernel : sixteen lanes per alignment (four alignments in flight per
//                     wave) stream the CIGAR once and reduce the per-alignment
//                     spans, clip runs and the number of long gaps;
//   2. scan_*       : exclusive scan of the counts -> CSR offsets d_gap_off;
//   3. emit_kernel  : a block checks 256 gap counts at once; only alignments that own
//                     a gap are re-read (a few % of the reads) by a whole wave;
//                     wave-level prefix sums of read/ref advance
//                     give readPos/refPos at every op, ballot-ranked stores
//                     keep op order.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/svx.h"

namespace {

constexpr int WAVE = 64;
constexpr int BLOCK = 256;
constexpr int WAVES_PER_BLOCK = BLOCK / WAVE;
constexpr int SCAN_ITEMS = 8;                       // per thread in the offset scan
constexpr int SCAN_TILE = BLOCK * SCAN_ITEMS;       // 2048 counts per block

// op codes: M0 I1 D2 N3 S4 H5 P6 =7 X8.  Read-advancing per the reference walk:
// M,I,N(!),S,H(as S),=,X ; ref-advancing: M,D,=,X (N does not move refPos).
__device__ inline bool adv_read(uint32_t op) { return (0x1BBu >> op) & 1u; }   // 0,1,3,4,5,7,8
__device__ inline bool adv_ref(uint32_t op)  { return (0x185u >> op) & 1u; }   // 0,2,7,8
__device__ inline bool span_ref(uint32_t op) { return (0x18Du >> op) & 1u; }   // 0,2,3,7,8 (reference_end)
__device__ inline bool in_query(uint32_t op) { return (0x1B3u >> op) & 1u; }   // 0,1,4,5,7,8
__device__ inline bool is_clip(uint32_t op)  { return op == 4u || op == 5u; }

__device__ inline long long wave_sum(long long v)
{
#pragma unroll
    for (int o = WAVE / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}
__device__ inline unsigned wave_sum_u(unsigned v)
{
#pragma unroll
    for (int o = WAVE / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}

constexpr int GROUP = 16;                           // lanes per alignment in the count pass
constexpr int ALN_PER_BLOCK = BLOCK / GROUP;

__device__ inline unsigned group_sum(unsigned v)
{
#pragma unroll
    for (int o = GROUP / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}
__device__ inline int group_min(int v)
{
#pragma unroll
    for (int o = GROUP / 2; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, WAVE));
    return v;
}
__device__ inline int group_max(int v)
{
#pragma unroll
    for (int o = GROUP / 2; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, WAVE));
    return v;
}

// Count pass.  HiFi CIGARs are 30-300 ops: a whole wave per alignment leaves most lanes idle and the
// kernel latency bound (offset load -> CIGAR load -> store, one alignment in flight per wave).  Sixteen
// lanes per alignment keep four alignments in flight per wave; sums are 32-bit modular (identical to the
// truncated 64-bit sums of the restatement).  Clip runs: every lane tracks the first / last non-clip op
// index it saw; after the group reduction the (0-2) clip ops outside [first, last] are re-read in parallel.
__global__ __launch_bounds__(BLOCK)
void count_kernel(const uint32_t* __restrict__ cigar, const uint64_t* __restrict__ cig_off,
                  uint32_t n_aln, int32_t min_sv, uint32_t* __restrict__ cnt, int32_t* __restrict__ stats)
{
    const int sub = threadIdx.x & (GROUP - 1);
    const uint32_t a = blockIdx.x * ALN_PER_BLOCK + (threadIdx.x / GROUP);
    const bool live = a < n_aln;
    const uint64_t b = live ? cig_off[a] : 0, e = live ? cig_off[a + 1] : 0;
    const long long n = (long long)(e - b);
    unsigned ref_span = 0, qlen = 0, ngap = 0;
    int first = 0x7fffffff, last = -1;                 // first / last non-clip op index (clamped to int)
    for (long long i = sub; i < n; i += 4 * GROUP) {
        uint32_t w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long ii = i + (long long)u * GROUP;
            w[u] = ii < n ? cigar[b + ii] : 0u;            // "0M": inert in every sum
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t op = w[u] & 15u, len = w[u] >> 4;
            if (span_ref(op)) ref_span += len;
            if (in_query(op)) qlen += len;
            ngap += ((op == 1u) | (op == 2u)) & ((long long)len >= (long long)min_sv);
            const long long ii = i + (long long)u * GROUP;
            if (ii < n && !is_clip(op)) {
                const int idx = ii > 0x7ffffffe ? 0x7ffffffe : (int)ii;
                first = min(first, idx);
                last = max(last, idx);
            }
        }
    }
    ngap = group_sum(ngap);
    if (live && sub == 0) cnt[a] = ngap;
    if (stats) {
        ref_span = group_sum(ref_span);
        qlen = group_sum(qlen);
        first = group_min(first);
        last = group_max(last);
        unsigned lead = 0, trail = 0;
        if (last < 0) {
            lead = qlen;                               // empty or all-clip CIGAR: everything is leading clip
        } else {
            for (long long i = sub; i < first; i += GROUP) lead += cigar[b + i] >> 4;
            for (long long i = (long long)last + 1 + sub; i < n; i += GROUP) trail += cigar[b + i] >> 4;
            lead = group_sum(lead);
            trail = group_sum(trail);
        }
        if (live && sub == 0) {
            int4 s;
            s.x = (int)ref_span; s.y = (int)lead; s.z = (int)trail; s.w = (int)qlen;
            reinterpret_cast<int4*>(stats)[a] = s;
        }
    }
}

// ---- exclusive scan of cnt[n] into off[n+1] (3 small kernels) ---------------------------
__global__ __launch_bounds__(BLOCK)
void scan_tiles_kernel(const uint32_t* __restrict__ cnt, uint32_t n, uint32_t* __restrict__ off,
                       uint32_t* __restrict__ tile_sum)
{
    __shared__ uint32_t wsum[WAVES_PER_BLOCK];
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS], t = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) { v[i] = base + i < n ? cnt[base + i] : 0u; t += v[i]; }
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6;
    uint32_t inc = t;
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) { uint32_t u = __shfl_up(inc, o, WAVE); if (lane >= o) inc += u; }
    if (lane == WAVE - 1) wsum[wv] = inc;
    __syncthreads();
    uint32_t pre = inc - t;
    for (int w = 0; w < wv; ++w) pre += wsum[w];
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) { if (base + i < n) off[base + i] = pre; pre += v[i]; }
    if (threadIdx.x == BLOCK - 1) tile_sum[blockIdx.x] = pre;
}

__global__ __launch_bounds__(WAVE)
void scan_tile_sums_kernel(uint32_t* __restrict__ tile_sum, uint32_t n_tiles, uint32_t* __restrict__ off, uint32_t n)
{
    // one wave walks the tile sums (n_aln / 2048 of them) 64 at a time
    const int lane = threadIdx.x;
    uint32_t carry = 0;
    for (uint32_t i = 0; i < n_tiles; i += WAVE) {
        const uint32_t t = i + lane < n_tiles ? tile_sum[i + lane] : 0u;
        uint32_t inc = t;
#pragma unroll
        for (int o = 1; o < WAVE; o <<= 1) { uint32_t u = __shfl_up(inc, o, WAVE); if (lane >= o) inc += u; }
        if (i + lane < n_tiles) tile_sum[i + lane] = carry + inc - t;
        carry += __shfl(inc, WAVE - 1, WAVE);
    }
    if (lane == 0) off[n] = carry;
}

__global__ __launch_bounds__(BLOCK)
void scan_add_kernel(uint32_t* __restrict__ off, uint32_t n, const uint32_t* __restrict__ tile_pre)
{
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) off[i] += tile_pre[i / SCAN_TILE];
}

// Emit pass: a block looks at 256 alignments' gap counts with one coalesced load; each wave then
// walks only the (rare) alignments of its 64 that own a long gap.
__global__ __launch_bounds__(BLOCK)
void emit_kernel(const uint32_t* __restrict__ cigar, const uint64_t* __restrict__ cig_off,
                 const int32_t* __restrict__ ref_start, uint32_t n_aln, int32_t min_sv,
                 const uint32_t* __restrict__ gap_off, SvxGap* __restrict__ gaps, uint64_t gaps_cap)
{
    const int lane = threadIdx.x & (WAVE - 1);
    const uint32_t mine = blockIdx.x * BLOCK + threadIdx.x;
    const bool has = mine < n_aln && gap_off[mine + 1] != gap_off[mine];
    unsigned long long todo = __ballot(has);
    const uint32_t wave_base = blockIdx.x * BLOCK + (threadIdx.x & ~(WAVE - 1));
    while (todo) {
        const int bit = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const uint32_t a = wave_base + bit;
        uint64_t dst = gap_off[a];
        const uint64_t b = cig_off[a], e = cig_off[a + 1];
        long long read_pos = 0, ref_pos = ref_start[a];    // carried across 64-op chunks
        for (uint64_t j0 = b; j0 < e; j0 += WAVE) {
            const uint64_t j = j0 + lane;
            const uint32_t w = j < e ? cigar[j] : 6u;
            const uint32_t op = w & 15u;
            const long long len = w >> 4;
            const long long dr = adv_read(op) ? len : 0, df = adv_ref(op) ? len : 0;
            long long ir = dr, irf = df;                   // inclusive wave prefix sums
#pragma unroll
            for (int o = 1; o < WAVE; o <<= 1) {
                const long long ur = __shfl_up(ir, o, WAVE), uf = __shfl_up(irf, o, WAVE);
                if (lane >= o) { ir += ur; irf += uf; }
            }
            const bool hit = ((op == 1u) | (op == 2u)) & (len >= (long long)min_sv);
            const unsigned long long m = __ballot(hit);
            if (hit) {
                const uint64_t slot = dst + __popcll(m & ((1ull << lane) - 1ull));
                if (slot < gaps_cap) {
                    SvxGap g;
                    g.aln = a; g.op = (uint32_t)(j - b);
                    g.read_pos = (int32_t)(read_pos + ir - dr);
                    g.ref_pos = (int32_t)(ref_pos + irf - df);
                    g.len = (int32_t)len; g.kind = op;     // I=1, D=2 match SVX_GAP_*
                    gaps[slot] = g;
                }
            }
            dst += __popcll(m);
            read_pos += __shfl(ir, WAVE - 1, WAVE);
            ref_pos += __shfl(irf, WAVE - 1, WAVE);
        }
    }
}

}  // namespace

extern "C" size_t svx_cigar_scan_ws_bytes(uint32_t n_aln)
{
    const size_t tiles = ((size_t)n_aln + SCAN_TILE - 1) / SCAN_TILE + 1;
    // [cnt: n_aln u32][tile_sum: tiles u32], 256 B aligned pieces
    return (((size_t)n_aln * 4 + 255) & ~(size_t)255) + ((tiles * 4 + 255) & ~(size_t)255);
}

extern "C" int svx_cigar_scan(const uint32_t* d_cigar, const uint64_t* d_cig_off,
                              const int32_t* d_ref_start, uint32_t n_aln, int32_t min_sv,
                              SvxGap* d_gaps, uint64_t gaps_cap, uint32_t* d_gap_off,
                              int32_t* d_stats, void* d_ws, void* stream)
{
    if (!d_gap_off) return SVX_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (n_aln == 0) {
        return hipMemsetAsync(d_gap_off, 0, sizeof(uint32_t), st) == hipSuccess ? SVX_OK : SVX_ELAUNCH;
    }
    if (!d_cigar || !d_cig_off || !d_ref_start || !d_ws || (!d_gaps && gaps_cap)) return SVX_EINVAL;
    if (d_stats && (reinterpret_cast<uintptr_t>(d_stats) & 15u)) return SVX_EINVAL;
    uint32_t* cnt = static_cast<uint32_t*>(d_ws);
    uint32_t* tile_sum = reinterpret_cast<uint32_t*>(static_cast<char*>(d_ws) + (((size_t)n_aln * 4 + 255) & ~(size_t)255));
    const uint32_t count_blocks = (n_aln + ALN_PER_BLOCK - 1) / ALN_PER_BLOCK;
    const uint32_t emit_blocks = (n_aln + BLOCK - 1) / BLOCK;
    const uint32_t tiles = (n_aln + SCAN_TILE - 1) / SCAN_TILE;
    hipLaunchKernelGGL(count_kernel, dim3(count_blocks), dim3(BLOCK), 0, st, d_cigar, d_cig_off, n_aln, min_sv, cnt, d_stats);
    hipLaunchKernelGGL(scan_tiles_kernel, dim3(tiles), dim3(BLOCK), 0, st, cnt, n_aln, d_gap_off, tile_sum);
    hipLaunchKernelGGL(scan_tile_sums_kernel, dim3(1), dim3(WAVE), 0, st, tile_sum, tiles, d_gap_off, n_aln);
    hipLaunchKernelGGL(scan_add_kernel, dim3((n_aln + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, d_gap_off, n_aln, tile_sum);
    hipLaunchKernelGGL(emit_kernel, dim3(emit_blocks), dim3(BLOCK), 0, st, d_cigar, d_cig_off, d_ref_start, n_aln, min_sv,
                       d_gap_off, d_gaps, gaps_cap);
    return hipGetLastError() == hipSuccess ? SVX_OK : SVX_ELAUNCH;
}
