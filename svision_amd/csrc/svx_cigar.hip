// svx_cigar.hip -- per-alignment CIGAR / segment scan for gfx950 (MI355X).
//
// Replaces the Python CIGAR walk of analyze_inside_align
// (reference src/collection/analyze_reads.py:828-853) and the pysam-derived
// reference_end / query_alignment_start / query_alignment_end for a whole
// batch of alignments whose packed BAM CIGAR words already sit in HBM.
//
// HBM-read bound: 4 B per CIGAR op, 16 B per alignment of CSR/start data,
// 24 B written per long gap (rare).  No atomics on results, output deterministic and
// sorted by (alignment, op) without a sort:
//   1. count_kernel  : eight lanes per alignment (eight alignments in flight per wave)
//                      stream the CIGAR once in 16-byte quads and reduce the per-alignment
//                      spans, clip runs and the number of long gaps; every workgroup also
//                      stores its gap and owner totals.  An alignment of more than 512 words
//                      (ONT) is only started here and put on a list;
//   1b. count_long_kernel + retotal_kernel: one wave per listed alignment streams the rest of it
//                      with eight 16-byte loads in flight per lane and adds its (integer) sums;
//                      the totals of the count workgroups concerned are recomputed from the counts;
//   2. scan_kernel   : exclusive prefix of the totals per tile of 256 alignments (one workgroup);
//   3. offsets_kernel: a workgroup per tile scans its counts into the CSR offsets d_gap_off
//                      and writes the alignments that own a gap (a few % of HiFi reads, most
//                      ONT reads) into a work list, in alignment order;
//   4. emit_kernel   : resident waves walk the work list, one wave per alignment, 256
//                      CIGAR words per step requested two steps ahead: wave prefix sums of
//                      read/ref advance give readPos/refPos at every op, ballot-ranked stores
//                      keep op order.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/svx.h"

namespace {

constexpr int WAVE = 64;
constexpr int BLOCK = 256;

// op codes: M0 I1 D2 N3 S4 H5 P6 =7 X8.  Read-advancing per the reference walk:
// M,I,N(!),S,H(as S),=,X ; ref-advancing: M,D,=,X (N does not move refPos).
__device__ inline bool adv_read(uint32_t op) { return (0x1BBu >> op) & 1u; }   // 0,1,3,4,5,7,8
__device__ inline bool adv_ref(uint32_t op)  { return (0x185u >> op) & 1u; }   // 0,2,7,8
__device__ inline bool span_ref(uint32_t op) { return (0x18Du >> op) & 1u; }   // 0,2,3,7,8 (reference_end)
__device__ inline bool in_query(uint32_t op) { return (0x1B3u >> op) & 1u; }   // 0,1,4,5,7,8
__device__ inline bool is_clip(uint32_t op)  { return op == 4u || op == 5u; }

__device__ inline unsigned wave_sum_u(unsigned v)
{
#pragma unroll
    for (int o = WAVE / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}

constexpr int TILE_SHIFT = 8;
constexpr int TILE = 1 << TILE_SHIFT;               // alignments per workgroup of the count / offsets passes
#ifndef SVX_CGROUP
#define SVX_CGROUP 8
#define SVX_CQUADS 2
#endif
constexpr int CGROUP = SVX_CGROUP;                  // lanes per alignment in the count pass
constexpr int CQUADS = SVX_CQUADS;                  // 16-byte loads in flight per lane
constexpr int ALN_PER_CBLOCK = BLOCK / CGROUP;      // alignments per workgroup of the count pass
constexpr int CBLOCKS_PER_TILE = TILE / ALN_PER_CBLOCK;
constexpr int LONG_Q = 128;                         // quads (512 words) of an alignment the eight-lane count pass handles itself

__device__ inline unsigned cgroup_sum(unsigned v)
{
#pragma unroll
    for (int o = CGROUP / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}

// one CIGAR word into the three per-alignment sums (32-bit modular; "0M" is inert)
__device__ inline void tally(uint32_t w, int32_t min_sv, unsigned& ref_span, unsigned& qlen, unsigned& ngap)
{
    const uint32_t op = w & 15u, len = w >> 4;
    ref_span += len & (0u - ((0x18Du >> op) & 1u));                 // M D N = X (reference_end)
    qlen += len & (0u - ((0x1B3u >> op) & 1u));                     // M I S H = X
    ngap += (uint32_t)((op - 1u) < 2u) & (uint32_t)((int32_t)len >= min_sv);
}

// Count pass.  HiFi CIGARs are 30-300 ops: a whole wave per alignment leaves most lanes idle and the
// kernel latency bound (offset load -> CIGAR load -> store, one alignment in flight per wave).  Eight lanes
// per alignment keep eight alignments in flight per wave, and every lane streams 16-byte quads of CIGAR
// words (two in flight): whole aligned quads are tallied unmasked and the (at most 3 + 3) words of the
// neighbouring alignments that the first and last quad drag in are subtracted once per alignment -- the sums
// are 32-bit modular (identical to the truncated 64-bit sums of the restatement), so that is exact.  The clip
// runs (query_alignment_start / _end) are the maximal runs of S/H words at either end, found with a ballot
// over the first / last eight words (longer runs loop).  All loads of an alignment are requested up front.
// The gap count of an alignment goes to gap_off[a] (turned into an offset by the offsets pass); the tile's
// workgroup's totals (gaps, alignments owning one) are plain stores: no atomics anywhere.
__global__ __launch_bounds__(BLOCK)
void count_kernel(const uint32_t* __restrict__ cigar, const uint64_t* __restrict__ cig_off,
                  uint32_t n_aln, int32_t min_sv, uint32_t* __restrict__ gap_off, int32_t* __restrict__ stats,
                  uint2* __restrict__ block_tot, uint32_t* __restrict__ long_list, uint32_t* __restrict__ long_count)
{
    __shared__ uint32_t s_tot[3];                        // gaps, owners, waves done
    if (threadIdx.x < 3) s_tot[threadIdx.x] = 0u;
    __syncthreads();                                     // the only barrier: before the waves drift apart
    const int sub = threadIdx.x & (CGROUP - 1);
    const int gshift = (threadIdx.x & (WAVE - 1)) & ~(CGROUP - 1);
    const uint32_t a = blockIdx.x * ALN_PER_CBLOCK + (threadIdx.x / CGROUP);
    const bool live = a < n_aln;
    const uint64_t full = cig_off[n_aln] >> 2;           // quads that lie entirely inside the array
    const uint64_t b = live ? cig_off[a] : 0, e = live ? cig_off[a + 1] : 0;
    const long long n = (long long)(e - b);
    const uint64_t q0 = b >> 2, q_end = min((e + 3) >> 2, full);
    // an alignment of more than LONG_Q quads (ONT: 10^3-10^5 ops) leaves everything behind its first LONG_Q quads to
    // count_long_kernel (a wave of its own, deep load pipeline): for its eight lanes here it would be a chain of
    // hundreds of memory round trips that the whole launch waits for.  All sums are modular and additive, so the
    // corrections below (computed from q_end) and the partial sums of the two kernels simply add up.
    const bool is_long = q_end - q0 > (uint64_t)LONG_Q;
    const uint64_t q1 = is_long ? q0 + LONG_Q : q_end;
    {                                                    // one atomic per wave: the (up to eight) group leaders share a reservation
        const unsigned long long m = __ballot(is_long && sub == 0);
        if (m) {
            const int wl = threadIdx.x & (WAVE - 1), first = __ffsll((long long)m) - 1;
            uint32_t base = 0;
            if (wl == first) base = atomicAdd(long_count, (uint32_t)__popcll(m));
            base = __shfl(base, first, WAVE);
            if (is_long && sub == 0) long_list[base + (uint32_t)__popcll(m & ((1ull << wl) - 1ull))] = a;
        }
    }
    // the words at either end for the clip runs, the neighbours' words inside the first / last quad (lanes 0-2
    // look before b and from e on), the alignment's words beyond the last whole quad of the array (at most 3,
    // last alignments only), then the quads
    const long long it = n - 1 - sub;
    const uint32_t w_head = sub < n ? cigar[b + sub] : 0u;
    const uint32_t w_tail = it >= 0 ? cigar[b + it] : 0u;
    const uint64_t fl = (b & ~3ull) + sub, ft = e + sub, tg = max(b, 4 * full) + sub;
    const uint32_t w_before = (q0 < q_end && sub < 3 && fl < b) ? cigar[fl] : 0u;
    const uint32_t w_after = (q0 < q_end && sub < 3 && ft < 4 * q_end) ? cigar[ft] : 0u;
    const uint32_t w_loose = (sub < 3 && tg < e) ? cigar[tg] : 0u;
    unsigned ref_span = 0, qlen = 0, ngap = 0;
    const uint4* __restrict__ quads = reinterpret_cast<const uint4*>(cigar);
    for (uint64_t q = q0 + sub; q < q1; q += CQUADS * CGROUP) {
        uint4 w[CQUADS];
#pragma unroll
        for (int u = 0; u < CQUADS; ++u) w[u] = quads[min(q + u * CGROUP, q1 - 1)];    // clamped, dropped below when out of range
#pragma unroll
        for (int u = 0; u < CQUADS; ++u) {
            const bool in = q + u * CGROUP < q1;
            tally(in ? w[u].x : 0u, min_sv, ref_span, qlen, ngap); tally(in ? w[u].y : 0u, min_sv, ref_span, qlen, ngap);
            tally(in ? w[u].z : 0u, min_sv, ref_span, qlen, ngap); tally(in ? w[u].w : 0u, min_sv, ref_span, qlen, ngap);
        }
    }
    tally(w_loose, min_sv, ref_span, qlen, ngap);
    {
        unsigned fr = 0, fq = 0, fn = 0;
        tally(w_before, min_sv, fr, fq, fn);
        tally(w_after, min_sv, fr, fq, fn);
        ref_span -= fr; qlen -= fq; ngap -= fn;
    }
    ngap = cgroup_sum(ngap);
    if (live && sub == 0) gap_off[a] = ngap;
    // workgroup totals without a closing barrier and off the waves' critical path: the (few) owners add to LDS,
    // fire and forget; the last wave to arrive stores the totals (LDS operations of a wave stay in order)
    if (sub == 0 && ngap != 0u) { atomicAdd(&s_tot[0], ngap); atomicAdd(&s_tot[1], 1u); }
    if ((threadIdx.x & (WAVE - 1)) == 0 && atomicAdd(&s_tot[2], 1u) == BLOCK / WAVE - 1)
        block_tot[blockIdx.x] = make_uint2(atomicAdd(&s_tot[0], 0u), atomicAdd(&s_tot[1], 0u));
    if (stats) {
        ref_span = cgroup_sum(ref_span);
        qlen = cgroup_sum(qlen);
        unsigned lead = 0, trail = 0;
        long long n_lead = 0;                                  // words in the leading clip run
        {
            const unsigned m = (unsigned)(__ballot(sub < n && is_clip(w_head & 15u)) >> gshift) & ((1u << CGROUP) - 1u);
            int run = __ffs((int)~m) - 1;                      // 0..CGROUP
            if (sub < run) lead += w_head >> 4;
            n_lead = run;
            for (long long base = CGROUP; run == CGROUP && base < n; base += CGROUP) {     // longer runs: rare
                const long long i = base + sub;
                const uint32_t w = i < n ? cigar[b + i] : 0u;
                const unsigned mm = (unsigned)(__ballot(i < n && is_clip(w & 15u)) >> gshift) & ((1u << CGROUP) - 1u);
                run = __ffs((int)~mm) - 1;
                if (sub < run) lead += w >> 4;
                n_lead += run;
            }
        }
        if (n_lead < n) {                                      // an all-clip CIGAR is all leading clip
            const unsigned m = (unsigned)(__ballot(it >= n_lead && is_clip(w_tail & 15u)) >> gshift) & ((1u << CGROUP) - 1u);
            int run = __ffs((int)~m) - 1;
            if (sub < run) trail += w_tail >> 4;
            for (long long base = CGROUP; run == CGROUP && base < n - n_lead; base += CGROUP) {
                const long long i = n - 1 - base - sub;
                const uint32_t w = i >= n_lead ? cigar[b + i] : 0u;
                const unsigned mm = (unsigned)(__ballot(i >= n_lead && is_clip(w & 15u)) >> gshift) & ((1u << CGROUP) - 1u);
                run = __ffs((int)~mm) - 1;
                if (sub < run) trail += w >> 4;
            }
        }
        lead = cgroup_sum(lead);
        trail = cgroup_sum(trail);
        if (live && sub == 0) {
            int4 s4;
            s4.x = (int)ref_span; s4.y = (int)lead; s4.z = (int)trail; s4.w = (int)qlen;
            reinterpret_cast<int4*>(stats)[a] = s4;
        }
    }
}

// Long alignments (listed by count_kernel, any order): one wave each, everything behind the first LONG_Q quads,
// LQUADS 16-byte loads in flight per lane (2048 words per step).  Adds its sums to what count_kernel stored
// (retotal_kernel then refreshes the totals of the count workgroups concerned).
constexpr int LQUADS = 8;

__global__ __launch_bounds__(BLOCK)
void count_long_kernel(const uint32_t* __restrict__ cigar, const uint64_t* __restrict__ cig_off, uint32_t n_aln, int32_t min_sv,
                       uint32_t* __restrict__ gap_off, int32_t* __restrict__ stats,
                       const uint32_t* __restrict__ long_list, const uint32_t* __restrict__ long_count)
{
    const int lane = threadIdx.x & (WAVE - 1);
    const uint32_t n_long = *long_count, n_waves = gridDim.x * (BLOCK / WAVE);
    const uint64_t full = cig_off[n_aln] >> 2;
    const uint4* __restrict__ quads = reinterpret_cast<const uint4*>(cigar);
    for (uint32_t k = blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6); k < n_long; k += n_waves) {
        const uint32_t a = long_list[k];
        const uint64_t b = cig_off[a], e = cig_off[a + 1];
        const uint64_t q0 = (b >> 2) + LONG_Q, q1 = min((e + 3) >> 2, full);
        unsigned ref_span = 0, qlen = 0, ngap = 0;
        for (uint64_t q = q0 + lane; q < q1; q += (uint64_t)LQUADS * WAVE) {
            uint4 w[LQUADS];
#pragma unroll
            for (int u = 0; u < LQUADS; ++u) w[u] = quads[min(q + (uint64_t)u * WAVE, q1 - 1)];
#pragma unroll
            for (int u = 0; u < LQUADS; ++u) {
                const bool in = q + (uint64_t)u * WAVE < q1;
                tally(in ? w[u].x : 0u, min_sv, ref_span, qlen, ngap); tally(in ? w[u].y : 0u, min_sv, ref_span, qlen, ngap);
                tally(in ? w[u].z : 0u, min_sv, ref_span, qlen, ngap); tally(in ? w[u].w : 0u, min_sv, ref_span, qlen, ngap);
            }
        }
        ref_span = wave_sum_u(ref_span); qlen = wave_sum_u(qlen); ngap = wave_sum_u(ngap);
        if (lane == 0) {
            if (stats) { stats[4 * (size_t)a] += (int)ref_span; stats[4 * (size_t)a + 3] += (int)qlen; }
            if (ngap) gap_off[a] += ngap;
        }
    }
}

// Totals of the count workgroups that hold a long alignment, recomputed from the final per-alignment counts (the
// count pass stored them before count_long_kernel added its share): one wave per listed alignment rewrites the totals
// of that alignment's group of ALN_PER_CBLOCK alignments -- idempotent, so several long alignments of one group are harmless.
__global__ __launch_bounds__(BLOCK)
void retotal_kernel(const uint32_t* __restrict__ gap_off, uint32_t n_aln, uint2* __restrict__ block_tot,
                    const uint32_t* __restrict__ long_list, const uint32_t* __restrict__ long_count)
{
    static_assert(ALN_PER_CBLOCK <= WAVE, "one lane per alignment of a count workgroup");
    const int lane = threadIdx.x & (WAVE - 1);
    const uint32_t n_long = *long_count, n_waves = gridDim.x * (BLOCK / WAVE);
    for (uint32_t k = blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6); k < n_long; k += n_waves) {
        const uint32_t blk = long_list[k] / ALN_PER_CBLOCK, a = blk * ALN_PER_CBLOCK + lane;
        const uint32_t c = (lane < ALN_PER_CBLOCK && a < n_aln) ? gap_off[a] : 0u;
        const unsigned gaps = wave_sum_u(c), owners = (unsigned)__popcll(__ballot(c != 0u));
        if (lane == 0) block_tot[blk] = make_uint2(gaps, owners);
    }
}

// Exclusive prefix over the tiles of 256 alignments (a tile = CBLOCKS_PER_TILE consecutive count workgroups);
// tile_pre[n_tiles] receives the grand totals.  One workgroup of 1024 threads; per step the totals of 1024 tiles
// are read with coalesced loads (all requested at once), folded per tile with lane shuffles and scanned.
constexpr int SBLOCK = 1024;

__global__ __launch_bounds__(SBLOCK)
void scan_kernel(const uint2* __restrict__ block_tot, uint32_t n_blocks, uint2* __restrict__ tile_pre, uint32_t n_tiles)
{
    __shared__ uint2 s_tile[SBLOCK];
    __shared__ uint2 s_wave[SBLOCK / WAVE];
    const int t = threadIdx.x, lane = t & (WAVE - 1), wv = t >> 6;
    uint2 carry = make_uint2(0u, 0u);                      // kept by every thread
    for (uint32_t base = 0; base < n_tiles; base += SBLOCK) {
        uint2 x[CBLOCKS_PER_TILE];
#pragma unroll
        for (int r = 0; r < CBLOCKS_PER_TILE; ++r) {
            const uint32_t blk = base * CBLOCKS_PER_TILE + r * SBLOCK + t;
            x[r] = block_tot[min(blk, n_blocks - 1)];
            if (blk >= n_blocks) x[r] = make_uint2(0u, 0u);
        }
#pragma unroll
        for (int r = 0; r < CBLOCKS_PER_TILE; ++r) {
#pragma unroll
            for (int o = 1; o < CBLOCKS_PER_TILE; o <<= 1) { x[r].x += __shfl_xor(x[r].x, o, WAVE); x[r].y += __shfl_xor(x[r].y, o, WAVE); }
            if ((t & (CBLOCKS_PER_TILE - 1)) == 0) s_tile[(r * SBLOCK + t) / CBLOCKS_PER_TILE] = x[r];
        }
        __syncthreads();
        const uint2 v = s_tile[t];
        uint2 inc = v;
#pragma unroll
        for (int o = 1; o < WAVE; o <<= 1) {
            const uint32_t ux = __shfl_up(inc.x, o, WAVE), uy = __shfl_up(inc.y, o, WAVE);
            if (lane >= o) { inc.x += ux; inc.y += uy; }
        }
        if (lane == WAVE - 1) s_wave[wv] = inc;
        __syncthreads();
        uint2 pre = carry, all = carry;
#pragma unroll
        for (int w = 0; w < SBLOCK / WAVE; ++w) {
            const uint2 y = s_wave[w];
            if (w < wv) { pre.x += y.x; pre.y += y.y; }
            all.x += y.x; all.y += y.y;
        }
        if (base + t < n_tiles) tile_pre[base + t] = make_uint2(pre.x + inc.x - v.x, pre.y + inc.y - v.y);
        carry = all;
        __syncthreads();
    }
    if (t == 0) tile_pre[n_tiles] = carry;
}

// Offsets pass, one workgroup per tile of 256 alignments: a workgroup scan on top of the tile's prefix turns
// the 256 counts into CSR offsets; the (few) alignments that own a long gap go to the work list (alignment,
// first slot) at the position given by the prefix of the owner counts, i.e. in alignment order.
__global__ __launch_bounds__(BLOCK)
void offsets_kernel(uint32_t n_aln, uint32_t* __restrict__ gap_off, const uint2* __restrict__ tile_pre, uint2* __restrict__ work)
{
    __shared__ uint32_t s_wave[2 * BLOCK / WAVE];
    constexpr int NW = BLOCK / WAVE;
    const int t = threadIdx.x, lane = t & (WAVE - 1), wv = t >> 6;
    const uint32_t tile = blockIdx.x;
    const uint32_t a = (tile << TILE_SHIFT) + t;
    const uint32_t c = a < n_aln ? gap_off[a] : 0u;
    const uint2 before = tile_pre[tile];
    uint32_t inc = c;
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) { const uint32_t u = __shfl_up(inc, o, WAVE); if (lane >= o) inc += u; }
    const unsigned long long owners = __ballot(c != 0u);
    if (lane == WAVE - 1) { s_wave[wv] = inc; s_wave[NW + wv] = (uint32_t)__popcll(owners); }
    __syncthreads();
    uint32_t off = before.x + inc - c, rank = before.y + (uint32_t)__popcll(owners & ((1ull << lane) - 1ull));
#pragma unroll
    for (int w = 0; w < NW; ++w)
        if (w < wv) { off += s_wave[w]; rank += s_wave[NW + w]; }
    if (a < n_aln) {
        gap_off[a] = off;
        if (a == n_aln - 1) gap_off[n_aln] = off + c;
        if (c) work[rank] = make_uint2(a, off);
    }
}

// Emit pass: resident waves take the work list with a grid stride, one wave per alignment, 4 consecutive
// CIGAR words per lane (256 words per step).  Lane-local sums + a wave prefix sum of the read / reference
// advance (32-bit modular = the truncated 64-bit positions) give readPos / refPos at every word; ballot
// ranks keep the stores in op order.
__global__ __launch_bounds__(BLOCK)
void emit_kernel(const uint32_t* __restrict__ cigar, const uint64_t* __restrict__ cig_off,
                 const int32_t* __restrict__ ref_start, int32_t min_sv, SvxGap* __restrict__ gaps, uint64_t gaps_cap,
                 const uint2* __restrict__ totals, const uint2* __restrict__ work)
{
    const int lane = threadIdx.x & (WAVE - 1);
    const uint32_t n_work = totals->y;                     // alignments owning a long gap
    const uint32_t n_waves = gridDim.x * (BLOCK / WAVE);
    const unsigned long long below = (1ull << lane) - 1ull;
    for (uint32_t k = blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6); k < n_work; k += n_waves) {
        const uint2 item = work[k];
        const uint32_t a = item.x;
        uint32_t dst = item.y;
        const uint64_t b = cig_off[a];
        const long long n = (long long)(cig_off[a + 1] - b);
        uint32_t read_pos = 0, ref_pos = (uint32_t)ref_start[a];
        // the words of a step are requested two steps ahead (three register sets, statically rotated): the positions
        // are a serial chain over the steps, so an ultra-long read (ONT: 10^4-10^5 ops, 40-400 steps) would otherwise
        // pay one memory round trip per step
        auto load = [&](long long j0, uint32_t (&w)[4]) {
            const long long j = j0 + 4 * lane;
#pragma unroll
            for (int u = 0; u < 4; ++u) w[u] = j + u < n ? cigar[b + j + u] : 6u;      // "0P": advances nothing
        };
        auto step = [&](long long j0, const uint32_t (&w)[4]) {
            const long long j = j0 + 4 * lane;
            uint32_t dr[4], df[4], tr = 0, tf = 0;
            bool hit[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t op = w[u] & 15u, len = w[u] >> 4;
                dr[u] = tr; df[u] = tf;                        // advance of this lane's earlier words
                tr += adv_read(op) ? len : 0u;
                tf += adv_ref(op) ? len : 0u;
                hit[u] = ((op - 1u) < 2u) & ((int32_t)len >= min_sv);
            }
            uint32_t ir = tr, irf = tf;                        // inclusive wave prefix sums of the lane totals
#pragma unroll
            for (int o = 1; o < WAVE; o <<= 1) {
                const uint32_t ur = __shfl_up(ir, o, WAVE), uf = __shfl_up(irf, o, WAVE);
                if (lane >= o) { ir += ur; irf += uf; }
            }
            // slot of a hit = hits in the lanes below (all four of their words) + this lane's earlier hits
            uint32_t mine = 0, all = 0, lower = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned long long m = __ballot(hit[u]);
                lower += (uint32_t)__popcll(m & below);
                all += (uint32_t)__popcll(m);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (hit[u]) {
                    const uint64_t slot = (uint64_t)dst + lower + mine;
                    if (slot < gaps_cap) {
                        SvxGap g;
                        g.aln = a; g.op = (uint32_t)(j + u);
                        g.read_pos = (int32_t)(read_pos + ir - tr + dr[u]);
                        g.ref_pos = (int32_t)(ref_pos + irf - tf + df[u]);
                        g.len = (int32_t)(w[u] >> 4); g.kind = w[u] & 15u;     // I=1, D=2 match SVX_GAP_*
                        gaps[slot] = g;
                    }
                    ++mine;
                }
            }
            dst += all;
            read_pos += __shfl(ir, WAVE - 1, WAVE);
            ref_pos += __shfl(irf, WAVE - 1, WAVE);
        };
        constexpr long long STEP = 4 * WAVE;
        uint32_t w0[4], w1[4], w2[4];
        load(0, w0);
        load(STEP, w1);
        for (long long j0 = 0; j0 < n; j0 += 3 * STEP) {
            load(j0 + 2 * STEP, w2);
            step(j0, w0);
            if (j0 + STEP >= n) break;
            load(j0 + 3 * STEP, w0);
            step(j0 + STEP, w1);
            if (j0 + 2 * STEP >= n) break;
            load(j0 + 4 * STEP, w1);
            step(j0 + 2 * STEP, w2);
        }
    }
}

// workspace: [block_tot: uint2 {gaps, owners} per count workgroup][tile_pre: uint2 per 256 alignments, + 1 for the
// totals] | [work: uint2 per alignment]
inline size_t ws_tile_offset(uint32_t n_aln)
{
    const size_t blocks = ((size_t)n_aln + ALN_PER_CBLOCK - 1) / ALN_PER_CBLOCK;
    return (blocks * sizeof(uint2) + 255) & ~(size_t)255;
}
inline size_t ws_work_offset(uint32_t n_aln)
{
    const size_t tiles = (((size_t)n_aln + TILE - 1) >> TILE_SHIFT) + 1;
    return ws_tile_offset(n_aln) + ((tiles * sizeof(uint2) + 255) & ~(size_t)255);
}

}  // namespace

extern "C" size_t svx_cigar_scan_ws_bytes(uint32_t n_aln)
{
    return ws_work_offset(n_aln) + (size_t)n_aln * sizeof(uint2);
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
    if ((reinterpret_cast<uintptr_t>(d_ws) & 7u) || (reinterpret_cast<uintptr_t>(d_cigar) & 15u)) return SVX_EINVAL;
    const uint32_t tiles = (n_aln + TILE - 1) >> TILE_SHIFT, count_blocks = (n_aln + ALN_PER_CBLOCK - 1) / ALN_PER_CBLOCK;
    uint2* block_tot = static_cast<uint2*>(d_ws);
    uint2* tile_pre = reinterpret_cast<uint2*>(static_cast<char*>(d_ws) + ws_tile_offset(n_aln));
    uint2* work = reinterpret_cast<uint2*>(static_cast<char*>(d_ws) + ws_work_offset(n_aln));
    // list of the long alignments: built by the count pass in the (not yet used) work area, its counter in the area's last word
    uint32_t* long_list = reinterpret_cast<uint32_t*>(work);
    uint32_t* long_count = reinterpret_cast<uint32_t*>(work + n_aln) - 1;
    if (hipMemsetAsync(long_count, 0, sizeof(uint32_t), st) != hipSuccess) return SVX_ELAUNCH;
    hipLaunchKernelGGL(count_kernel, dim3(count_blocks), dim3(BLOCK), 0, st, d_cigar, d_cig_off, n_aln, min_sv, d_gap_off, d_stats, block_tot,
                       long_list, long_count);
    const uint32_t long_blocks = min(2048u, (n_aln + 3u) / 4u);
    hipLaunchKernelGGL(count_long_kernel, dim3(long_blocks), dim3(BLOCK), 0, st, d_cigar, d_cig_off, n_aln, min_sv,
                       d_gap_off, d_stats, long_list, long_count);
    hipLaunchKernelGGL(retotal_kernel, dim3(long_blocks), dim3(BLOCK), 0, st, d_gap_off, n_aln, block_tot, long_list, long_count);
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(SBLOCK), 0, st, block_tot, count_blocks, tile_pre, tiles);
    hipLaunchKernelGGL(offsets_kernel, dim3(tiles), dim3(BLOCK), 0, st, n_aln, d_gap_off, tile_pre, work);
    // resident waves (8 workgroups per CU at most); small inputs get one wave per 4 alignments
    const uint32_t emit_blocks = min(2048u, (n_aln + 15u) / 16u);
    hipLaunchKernelGGL(emit_kernel, dim3(emit_blocks), dim3(BLOCK), 0, st, d_cigar, d_cig_off, d_ref_start, min_sv, d_gaps, gaps_cap,
                       tile_pre + tiles, work);
    return hipGetLastError() == hipSuccess ? SVX_OK : SVX_ELAUNCH;
}
