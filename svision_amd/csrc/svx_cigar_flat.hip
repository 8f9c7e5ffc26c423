// svx_cigar_flat.hip -- svx_cigar_scan for LONG alignments (ONT ultra-long, assembly contigs: 10^3-10^6 CIGAR operations each) in
// ONE pass over the words, on gfx950.  Same contract and the same outputs, bit for bit, as the three kernels of svx_cigar.hip
// (reference src/collection/analyze_reads.py:828-853 + pysam's reference_end / query_alignment_*).
//
// Why a second form.  svx_cigar.hip gives eight lanes to an alignment: right for HiFi (30-300 operations), but an alignment of
// more than 512 words is finished by ONE wave at the end of the count pass and walked again by ONE wave in the emit pass, 256
// words per step with a serial position chain -- the longest read of an ONT sample (10^5 words) is a chain of ~400 steps that
// the whole launch waits for, and every alignment that owns a long gap (nearly all of them) is read twice: 0.17 of the HBM
// peak on the ONT-shaped launch (profiles/r04_bench_cigar.json).
//
// Here the FLAT array of words is cut into chunks of 2,048 words, one wave per chunk, whatever alignment the words belong to:
//   * the wave loads its 8 KB at once (eight 16-byte loads per lane, coalesced), finds the alignment its first word belongs to
//     (a 64-ary search over d_cig_off: 3-4 probes) and the alignment ends inside the chunk (the next 64 offsets: one load);
//   * every lane tallies its 32 words (reference span, query length, N length, long gaps) per 16-byte quad; the sums at an
//     alignment boundary are the quads below it + part of the one quad that straddles it;
//   * what the chunk cannot know -- how many long gaps lie in front of it, and, if its first alignment started in an earlier
//     chunk, that alignment's sums up to here -- comes from the chunks in front: the eight chunks of a workgroup exchange
//     their gap counts and the sums of the alignment open at their ends through LDS (one barrier), the workgroups through a
//     decoupled LOOK-BACK over 64-bit descriptors (64 per step): a workgroup publishes its totals as soon as it has tallied,
//     first as "this workgroup alone", then, its own look-back done, as "everything up to here" (one barrier more hands the
//     result to its chunks).  With a descriptor per chunk the ~5,000 chunks in flight looked back over each other: 1.3 G polls;
//   * the alignment's statistics are written by the chunk that holds its END, its CSR offset by the chunk that holds its START,
//     its gaps -- positions from the prefix sums inside the chunk + the carried sums -- by the chunks that hold them, at
//     (gaps in front of the chunk) + (rank inside the chunk): sorted by (alignment, operation) without a sort.
// Every word is read once; nothing is written but the outputs and 40 bytes of descriptors per workgroup (16,384 words).  Algorithmic bytes as for
// svx_cigar_scan: 4 B x words + 16 B x alignments read, 16 B x alignments + 24 B x long gaps written.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/svx.h"

namespace {

#ifndef SVX_FLAT_DBG
#define SVX_FLAT_DBG 0                               // measurements only (wrong output): 1 = no look-back between workgroups, 2 = nothing behind the tallies, 4 = no tallies
#endif
#ifndef SVX_FLAT_QUADS
#define SVX_FLAT_QUADS 8
#endif
#ifndef SVX_FLAT_BLOCK
#define SVX_FLAT_BLOCK 512
#endif
constexpr int WAVE = 64, FBLOCK = SVX_FLAT_BLOCK, FQ = SVX_FLAT_QUADS, NW = FBLOCK / WAVE;     // FQ: 16-byte quads per lane (a lane takes 4 FQ CONSECUTIVE words)
constexpr uint32_t LW = 4u * FQ, FCH = LW * WAVE;    // words per lane (32) and per chunk (2,048)
constexpr unsigned long long F_VALID = 1ull << 63, F_CLOSED = 1ull << 62, G_AGG = 1ull << 62, G_INC = 2ull << 62, G_FLAG = 3ull << 62;

// Inclusive prefix sum over the wave with data-parallel primitives (row shifts inside the rows of 16 lanes, then the two row
// broadcasts of gfx9): seven moves in the vector ALU.  (__shfl_* is ds_bpermute on this target -- a round trip through the LDS
// crossbar per step; six dependent ones per sum and a dozen sums per chunk were most of a chunk's time behind its tallies.)
__device__ inline unsigned wscan(unsigned v)
{
    unsigned r = v;
    r += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);      // row_shr:1
    r += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);      // row_shr:2
    r += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x113, 0xf, 0xf, true);      // row_shr:3
    r += (unsigned)__builtin_amdgcn_update_dpp(0, (int)r, 0x114, 0xf, 0xe, true);      // row_shr:4, banks 1-3
    r += (unsigned)__builtin_amdgcn_update_dpp(0, (int)r, 0x118, 0xf, 0xc, true);      // row_shr:8, banks 2-3
    r += (unsigned)__builtin_amdgcn_update_dpp(0, (int)r, 0x142, 0xa, 0xf, false);     // row_bcast:15 into rows 1 and 3
    r += (unsigned)__builtin_amdgcn_update_dpp(0, (int)r, 0x143, 0xc, 0xf, false);     // row_bcast:31 into rows 2 and 3
    return r;
}
__device__ inline unsigned wsum(unsigned v) { return (unsigned)__builtin_amdgcn_readlane((int)wscan(v), WAVE - 1); }

struct Sums { unsigned span, qlen, nlen, ngap; };    // reference span (M D N = X), query length (M I S H = X), N length, long gaps
__device__ inline void add(Sums& a, const Sums& b) { a.span += b.span; a.qlen += b.qlen; a.nlen += b.nlen; a.ngap += b.ngap; }
__device__ inline Sums sub(const Sums& a, const Sums& b) { return Sums{a.span - b.span, a.qlen - b.qlen, a.nlen - b.nlen, a.ngap - b.ngap}; }
__device__ inline Sums wsum(const Sums& s) { return Sums{wsum(s.span), wsum(s.qlen), wsum(s.nlen), wsum(s.ngap)}; }
__device__ inline bool is_gap(uint32_t w, int32_t min_sv) { return ((w & 15u) - 1u) < 2u && (int32_t)(w >> 4) >= min_sv; }
__device__ inline void tally(uint32_t w, int32_t min_sv, Sums& s)
{
    const uint32_t op = w & 15u, len = w >> 4;
    s.span += len & (uint32_t)__builtin_amdgcn_sbfe(0x18D, op, 1u);
    s.qlen += len & (uint32_t)__builtin_amdgcn_sbfe(0x1B3, op, 1u);
    s.nlen += len & (uint32_t)__builtin_amdgcn_sbfe(0x008, op, 1u);
    s.ngap -= (uint32_t)((int32_t)len >= min_sv ? __builtin_amdgcn_sbfe(0x6, op, 1u) : 0);
}
__device__ inline uint32_t word_of(const uint4& q, int k) { return k == 0 ? q.x : k == 1 ? q.y : k == 2 ? q.z : q.w; }
__device__ inline bool is_clip(uint32_t op) { return op == 4u || op == 5u; }

struct Desc {                                         // per WORKGROUP (NW chunks): one array of 64-bit words each, zeroed before the launch
    unsigned long long* gaps;                         // [flag:2 | long gaps: this chunk alone (G_AGG) / everything up to its end (G_INC)]
    unsigned long long* p1; unsigned long long* p2;   // the alignment open at the chunk's end, this chunk's words of it: [span | qlen], [F_VALID | nlen]
    unsigned long long* i1; unsigned long long* i2;   // the same from the alignment's start on -- or F_CLOSED: the chunk ends on a boundary
};

__device__ inline void publish(unsigned long long* a, unsigned long long va, unsigned long long* b, unsigned long long vb, uint32_t c)
{
    // the data word first, acknowledged by the memory system, then the word that carries the flag
    __hip_atomic_store(&a[c], va, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_waitcnt(0x0f70);              // vmcnt(0)
    __hip_atomic_store(&b[c], vb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one workgroup's tile (NW chunks); every return lies behind the tile's last barrier -- or is taken by the whole workgroup
__device__ __forceinline__ void flat_tile(const uint32_t* __restrict__ cigar, const uint64_t* __restrict__ cig_off, const int32_t* __restrict__ ref_start,
                       uint32_t n_aln, int32_t min_sv, SvxGap* __restrict__ gaps, uint64_t gaps_cap, uint32_t* __restrict__ gap_off,
                       int32_t* __restrict__ stats, Desc d, uint32_t wg, const uint32_t* __restrict__ first_aln)
{
    const uint32_t lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6;
    const uint32_t c = wg * NW + wv;
    const uint64_t begin = cig_off[0], end = cig_off[n_aln];
    if (begin == end) {                               // no words at all: every alignment is empty; the first chunk says so
        if (c == 0) {
            for (uint32_t a = lane; a < n_aln; a += WAVE) {
                gap_off[a] = 0u;
                if (stats) reinterpret_cast<int4*>(stats)[a] = make_int4(0, 0, 0, 0);
            }
            if (lane == 0) gap_off[n_aln] = 0u;
        }
        return;
    }
    // A workgroup = NW consecutive chunks.  The chunks of a workgroup exchange their sums through LDS, the workgroups through
    // the descriptors: with a descriptor per CHUNK, the ~5,000 chunks in flight each looked back over the thousands in front of
    // them that had not finished their own look-back either (1.3 G polling loads per launch: 3 x slower than the two-pass form).
    const uint64_t wg0 = (begin & ~3ull) + (uint64_t)wg * NW * FCH;
    if (wg0 >= end) return;                           // behind the words (the launch was sized by an upper bound): the whole workgroup
    const uint32_t n_act = (uint32_t)min((uint64_t)NW, (end - wg0 + FCH - 1) / FCH);     // its chunks that hold words: a prefix
    const bool active = wv < n_act;
    const uint64_t w0 = wg0 + (uint64_t)wv * FCH;
    const uint64_t w1 = min(w0 + FCH, end), wf = max(w0, begin);
    __shared__ uint32_t s_loc[NW][5];                 // per chunk: long gaps, state of the alignment open at its end (0 none, 1 whole, 2 a part), its span / qlen / nlen
    __shared__ uint32_t s_wg[4];                      // the workgroup's look-back: gaps in front, carried span / qlen / nlen
    __shared__ uint64_t s_end[NW][WAVE];              // per chunk: the first 64 alignment ends inside it ...
    __shared__ uint4 s_pre[NW][WAVE];                 // ... and the chunk's sums below each (what a gap's positions are relative to)
    uint32_t a0 = 0, n_ends = 0, a_tail = 0;
    uint64_t last_end = wf;
    bool tail_open = false, head_carried = false;
    Sums chunk{0u, 0u, 0u, 0u}, tail_own{0u, 0u, 0u, 0u}, tot{0u, 0u, 0u, 0u};
    const uint64_t l0 = w0 + (uint64_t)LW * lane;     // this lane's first word
    // the sums of the chunk's words below word x (wf <= x <= w1), wave-uniform: the lanes below it + part of the one it cuts
    auto below = [&](uint64_t x) {
        const uint32_t rel = (uint32_t)(x - w0), L = rel / LW, r = rel % LW;
        Sums s{0u, 0u, 0u, 0u};
        if (lane < L) s = tot;
        if (r) {                                      // the cut lane's first r words, a lane each (LW <= 64)
            const uint64_t i = w0 + (uint64_t)LW * L + lane;
            if (lane < r && i >= begin) tally(cigar[i], min_sv, s);
        }
        return wsum(s);
    };
    if (active) {
    // ---- the chunk's words: a lane takes LW CONSECUTIVE words (FQ 16-byte loads; the lanes of a load are 128 bytes apart, the
    // loads of a lane fill its lines); words outside [begin, end) count as "0M".  Nothing but the lane's four sums is kept: a
    // prefix at a boundary = the lanes below it + part of the one it cuts (first version: quads interleaved over the lanes and
    // every quad's sums in LDS -- 8 KB per wave, which held the chip to 16 waves per CU).
    {
        uint4 q[FQ];
        if (w0 >= begin && w0 + FCH <= end) {         // the whole chunk lies inside the words: eight plain loads, all in flight at once
            const uint4* p = reinterpret_cast<const uint4*>(cigar + l0);      // (the masked form below puts every load into a branch of
#pragma unroll                                                                 // its own, and the wave waited for them one by one:
            for (int u = 0; u < FQ; ++u) q[u] = p[u];                          // 1.3 TB/s with nothing but the loads in the kernel)
        } else {
#pragma unroll
            for (int u = 0; u < FQ; ++u) {
                const uint64_t i0 = l0 + 4ull * u;
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (i0 + 4 <= end) v = *reinterpret_cast<const uint4*>(cigar + i0);
                else if (i0 < end) { v.x = cigar[i0]; if (i0 + 1 < end) v.y = cigar[i0 + 1]; if (i0 + 2 < end) v.z = cigar[i0 + 2]; }
                if (i0 < begin) { v.x = 0u; if (i0 + 1 < begin) v.y = 0u; if (i0 + 2 < begin) v.z = 0u; if (i0 + 3 < begin) v.w = 0u; }
                q[u] = v;
            }
        }
#pragma unroll
        for (int u = 0; u < FQ; ++u) {
            if (SVX_FLAT_DBG & 4) { tot.ngap += q[u].x ^ q[u].y ^ q[u].z ^ q[u].w; continue; }
            tally(q[u].x, min_sv, tot); tally(q[u].y, min_sv, tot); tally(q[u].z, min_sv, tot); tally(q[u].w, min_sv, tot);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    chunk = wsum(tot);
    if (SVX_FLAT_DBG & 2) { if (chunk.ngap == 0xffffffffu) gap_off[0] = 1; return; }
    // ---- a0: the alignment that holds word wf (the smallest a with cig_off[a + 1] > wf), from the map cigar_map_kernel has written:
    // a 64-ary search over d_cig_off here was four dependent memory round trips on every chunk's critical path
    a0 = first_aln[c];
    // ---- the alignment open at the chunk's end (if any) is a0 + (alignment ends in the chunk); its words in this chunk = the
    // chunk's sums - the sums below its start.  A first pass over the ends only to find the last one (long alignments: none or one).
    for (uint32_t a = a0;; a += WAVE) {
        const uint64_t e = (uint64_t)a + lane < n_aln ? cig_off[a + lane + 1] : ~0ull;
        const unsigned long long m = __ballot(e <= w1);
        const uint32_t k = (uint32_t)__popcll(m);
        if (k) last_end = __shfl(e, (int)k - 1, WAVE);
        n_ends += k;
        if (k < WAVE) break;
    }
    a_tail = a0 + n_ends;
    tail_open = a_tail < n_aln && last_end < w1;                  // (an alignment that starts exactly at w1 belongs to the next chunk)
    head_carried = c > 0 && cig_off[a0] < w0;                     // a0 started in an earlier chunk
    if (tail_open) tail_own = n_ends == 0 ? chunk : sub(chunk, below(last_end));     // (a0 holds word wf: nothing of the chunk lies in front of it)
    }   // active
    if (lane == 0) {
        s_loc[wv][0] = chunk.ngap;
        s_loc[wv][1] = !tail_open ? 0u : (n_ends == 0 && head_carried) ? 2u : 1u;
        s_loc[wv][2] = tail_own.span; s_loc[wv][3] = tail_own.qlen; s_loc[wv][4] = tail_own.nlen;
    }
    __syncthreads();
    // ---- inside the workgroup: the gaps of the chunks in front, and what a0 carries from them (lane j looks at chunk wv - 1 - j)
    uint32_t gfront = 0;
    Sums carry{0u, 0u, 0u, 0u};
    bool need_x = false;                              // a0 started in front of the workgroup
    if (active) {
        const bool in = lane < wv;
        const uint32_t v = in ? wv - 1 - lane : 0;
        gfront = wsum(in ? s_loc[v][0] : 0u);
        if (head_carried) {
            const uint32_t st = in ? s_loc[v][1] : 2u;
            const unsigned long long stop_m = __ballot(in && st != 2u);
            const int stop = stop_m ? __ffsll((long long)stop_m) - 1 : WAVE;
            const bool take = in && (int)lane <= stop && st != 0u;
            carry.span = wsum(take ? s_loc[v][2] : 0u); carry.qlen = wsum(take ? s_loc[v][3] : 0u); carry.nlen = wsum(take ? s_loc[v][4] : 0u);
            need_x = stop_m == 0;
        }
    }
    // ---- between the workgroups: its first wave publishes the workgroup's gaps and the alignment open at its end -- "this
    // workgroup alone" until its own look-back is through, "everything up to here" afterwards -- and looks back
    if (wv == 0) {
        const uint32_t la = n_act - 1;
        const bool in = lane <= la;
        const uint32_t v = in ? la - lane : 0;
        const uint32_t wg_gaps = wsum(in ? s_loc[v][0] : 0u);
        const uint32_t st = in ? s_loc[v][1] : 1u;
        const uint32_t st_last = s_loc[la][1];
        const unsigned long long stop_m = __ballot(in && st != 2u);
        const int stop = stop_m ? __ffsll((long long)stop_m) - 1 : WAVE;
        const bool take = in && (int)lane <= stop && st != 0u;
        const Sums wt{wsum(take ? s_loc[v][2] : 0u), wsum(take ? s_loc[v][3] : 0u), wsum(take ? s_loc[v][4] : 0u), 0u};
        const bool wg_partial = st_last != 0u && stop_m == 0;      // every chunk of the workgroup lies inside ONE alignment that started in front of it
        if (lane == 0) {
            __hip_atomic_store(&d.gaps[wg], G_AGG | wg_gaps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (st_last == 0u) __hip_atomic_store(&d.i2[wg], F_VALID | F_CLOSED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (wg_partial) publish(d.p1, (unsigned long long)wt.span << 32 | wt.qlen, d.p2, F_VALID | wt.nlen, wg);
            else publish(d.i1, (unsigned long long)wt.span << 32 | wt.qlen, d.i2, F_VALID | wt.nlen, wg);
        }
        uint32_t gx = 0;
        Sums cx{0u, 0u, 0u, 0u};
        if (wg > 0 && !(SVX_FLAT_DBG & 1)) {
            bool need_g = true, need_c = head_carried;    // (this wave's a0 = the workgroup's)
            long long j = (long long)wg - 1;
            while (need_g || need_c) {
                const long long i = j - lane;
                unsigned long long g = G_INC, t2 = F_VALID | F_CLOSED, t1 = 0ull;
                bool incl = true;                         // (in front of the first workgroup: nothing, inclusively)
                if (i >= 0) {
                    int spins = 0;
                    if (need_g) { do { g = __hip_atomic_load(&d.gaps[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((g & G_FLAG) == 0 && ++spins < (1 << 22)); }
                    if (need_c) {
                        spins = 0;
                        for (;;) {
                            t2 = __hip_atomic_load(&d.i2[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (t2 & F_VALID) { incl = true; t1 = (t2 & F_CLOSED) ? 0ull : __hip_atomic_load(&d.i1[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                            t2 = __hip_atomic_load(&d.p2[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (t2 & F_VALID) { incl = false; t1 = __hip_atomic_load(&d.p1[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                            if (++spins >= (1 << 22)) { t2 = 0ull; break; }
                        }
                    }
                    if ((need_g && (g & G_FLAG) == 0) || (need_c && (t2 & F_VALID) == 0)) atomicMax(&gap_off[n_aln], SVX_SCAN_FAILED);   // never seen; loud if it ever is
                }
                if (need_g) {
                    const unsigned long long im = __ballot((g & G_FLAG) == G_INC);
                    const int stp = im ? __ffsll((long long)im) - 1 : WAVE - 1;
                    gx += wsum((int)lane <= stp ? (uint32_t)g : 0u);
                    if (im) need_g = false;
                }
                if (need_c) {
                    const unsigned long long im = __ballot(incl);
                    const int stp = im ? __ffsll((long long)im) - 1 : WAVE - 1;
                    const bool tk = (int)lane <= stp;
                    cx.span += wsum(tk ? (uint32_t)(t1 >> 32) : 0u);
                    cx.qlen += wsum(tk ? (uint32_t)t1 : 0u);
                    cx.nlen += wsum(tk ? (uint32_t)t2 : 0u);
                    if (im) need_c = false;
                }
                j -= WAVE;
            }
        }
        if (lane == 0) {
            __hip_atomic_store(&d.gaps[wg], G_INC | (unsigned long long)(uint32_t)(gx + wg_gaps), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (wg_partial)
                publish(d.i1, (unsigned long long)(cx.span + wt.span) << 32 | (uint32_t)(cx.qlen + wt.qlen), d.i2, F_VALID | (uint32_t)(cx.nlen + wt.nlen), wg);
            s_wg[0] = gx; s_wg[1] = cx.span; s_wg[2] = cx.qlen; s_wg[3] = cx.nlen;
        }
        need_x = head_carried;
    }
    __syncthreads();
    if (!active) return;
    gfront += s_wg[0];
    if (need_x) { carry.span += s_wg[1]; carry.qlen += s_wg[2]; carry.nlen += s_wg[3]; }
    // ---- the first chunk also owns the empty alignments in front of the first word
    if (c == 0) {
        for (uint32_t a = lane; a < a0; a += WAVE) {
            gap_off[a] = 0u;
            if (stats) reinterpret_cast<int4*>(stats)[a] = make_int4(0, 0, 0, 0);
        }
    }
    if (lane == 0 && !head_carried) gap_off[a0] = gfront;                        // a0 starts with this chunk's first word
    // ---- the alignments that end in the chunk (lane j: alignment a + j): their statistics and -- they start here as well, a0 apart --
    // their CSR offsets; the alignment open at the chunk's end starts behind the last of them
    uint32_t k_first = 0;                             // (the first 64 ends and the sums below them go to LDS: what the gaps' positions are relative to)
    {
        Sums prev{0u, 0u, 0u, 0u};                    // the sums below the previous end
        bool first_batch = true;
        for (uint32_t a = a0;; a += WAVE) {
            const bool have = (uint64_t)a + lane < n_aln;
            const uint64_t e = have ? cig_off[a + lane + 1] : ~0ull;
            const unsigned long long m = __ballot(e <= w1);
            const uint32_t k = (uint32_t)__popcll(m);
            Sums mine{0u, 0u, 0u, 0u};                // the sums below this lane's end
            for (uint32_t jj = 0; jj < k; ++jj) {
                const uint64_t x = __shfl(e, (int)jj, WAVE);
                const Sums s = below(x);
                if (lane == jj) mine = s;
            }
            Sums before;
            before.span = __shfl_up(mine.span, 1, WAVE); before.qlen = __shfl_up(mine.qlen, 1, WAVE);
            before.nlen = __shfl_up(mine.nlen, 1, WAVE); before.ngap = __shfl_up(mine.ngap, 1, WAVE);
            if (lane == 0) before = prev;
            if (first_batch) { s_end[wv][lane] = lane < k ? e : ~0ull; s_pre[wv][lane] = make_uint4(mine.span, mine.qlen, mine.nlen, mine.ngap); k_first = k; }
            if (lane < k) {
                Sums own = sub(mine, before);
                if (first_batch && lane == 0 && head_carried) { own.span += carry.span; own.qlen += carry.qlen; own.nlen += carry.nlen; }
                const uint32_t al = a + lane;
                if (stats) {
                    // clip runs: the maximal runs of S / H words at either end (an all-clip CIGAR is all leading clip)
                    const uint64_t b = cig_off[al];
                    const long long n = (long long)(e - b);
                    unsigned lead = 0, trail = 0;
                    long long nl = 0;
                    while (nl < n) { const uint32_t w = cigar[b + nl]; if (!is_clip(w & 15u)) break; lead += w >> 4; ++nl; }
                    for (long long t = n - 1; t >= nl; --t) { const uint32_t w = cigar[b + t]; if (!is_clip(w & 15u)) break; trail += w >> 4; }
                    reinterpret_cast<int4*>(stats)[al] = make_int4((int)own.span, (int)lead, (int)trail, (int)own.qlen);
                }
                if (al != a0) gap_off[al] = gfront + before.ngap;                  // it starts where the one before it ends: in this chunk
                if (lane == k - 1 && k < WAVE && tail_open) gap_off[al + 1] = gfront + mine.ngap;     // ... and so does the open one behind the last end
            }
            if (k) { prev.span = __shfl(mine.span, (int)k - 1, WAVE); prev.qlen = __shfl(mine.qlen, (int)k - 1, WAVE);
                     prev.nlen = __shfl(mine.nlen, (int)k - 1, WAVE); prev.ngap = __shfl(mine.ngap, (int)k - 1, WAVE); }
            first_batch = false;
            if (k < WAVE) break;
        }
    }
    if (w1 == end && lane == 0) atomicMax(&gap_off[n_aln], gfront + chunk.ngap);          // the total (atomicMax: SVX_SCAN_FAILED stays)
    // ---- the chunk's long gaps: slot = gaps in front of the chunk + gaps of the chunk below the word; positions = the sums from
    // the alignment's start to the word = (chunk below the word) - (chunk below the alignment's start) [+ what a0 carries]
    if (chunk.ngap == 0) return;
    Sums at{wscan(tot.span), wscan(tot.qlen), wscan(tot.nlen), wscan(tot.ngap)};     // (inclusive) prefix of the lanes' sums
    at = sub(at, tot);
    if (tot.ngap == 0) return;
    // (rare: the lane walks its words once more, out of the cache)
#pragma unroll 1
    for (uint64_t idx = max(l0, begin); idx < min(l0 + LW, end); ++idx) {
        const uint32_t w = cigar[idx];
        if (is_gap(w, min_sv)) {
            // the word's alignment = a0 + (ends at or below it); the sums below that alignment's start
            uint32_t cnt = 0;
            while (cnt < k_first && s_end[wv][cnt] <= idx) ++cnt;
            uint32_t al = a0 + cnt;
            Sums rel = at;
            if (cnt) { const uint4 pv = s_pre[wv][cnt - 1]; rel = sub(at, Sums{pv.x, pv.y, pv.z, pv.w}); }
            else if (head_carried) { rel.span += carry.span; rel.qlen += carry.qlen; rel.nlen += carry.nlen; }
            uint64_t b = cig_off[al];
            if (cnt == WAVE) {                        // more than 64 alignments end in this chunk and the word lies behind the 64th: by foot
                while (al + 1 < n_aln && cig_off[al + 1] <= idx) ++al;
                b = cig_off[al];
                rel = Sums{0u, 0u, 0u, 0u};
                for (uint64_t x = b; x < idx; ++x) tally(cigar[x], min_sv, rel);
            }
            const uint64_t slot = (uint64_t)gfront + at.ngap;
            if (slot < gaps_cap) {
                SvxGap g;
                g.aln = al; g.op = (uint32_t)(idx - b);
                g.read_pos = (int32_t)(rel.qlen + rel.nlen);
                g.ref_pos = (int32_t)((uint32_t)ref_start[al] + rel.span - rel.nlen);
                g.len = (int32_t)(w >> 4); g.kind = w & 15u;
                gaps[slot] = g;
            }
        }
        tally(w, min_sv, at);
    }
}

// first_aln[c] = the alignment that holds chunk c's first word: every alignment writes the entries of the chunks that begin inside
// it (a thread per alignment; a long one loops over its chunks).
__global__ __launch_bounds__(256)
// n_first: the entries first_aln has (sized from the caller's n_words_max).  An array that holds more words than the caller said
// would be written behind its end: the entries stay inside, and the launch is marked failed (SVX_SCAN_FAILED in gap_off[n_aln], as
// svx_cigar_scan answers the same mistake -- ADVICE r5).
__global__ __launch_bounds__(256)
void cigar_map_kernel(const uint64_t* __restrict__ cig_off, uint32_t n_aln, uint32_t* __restrict__ first_aln, uint64_t n_first,
                      uint64_t n_words_max, uint32_t* __restrict__ gap_off)
{
    const uint32_t a = blockIdx.x * 256 + threadIdx.x;
    if (a >= n_aln) return;
    const uint64_t begin = cig_off[0], base = begin & ~3ull, b = cig_off[a], e = cig_off[a + 1];
    if (a == n_aln - 1 && e - begin > n_words_max) atomicMax(&gap_off[n_aln], SVX_SCAN_FAILED);
    if (e == b) return;
    if (b <= begin) first_aln[0] = a;                 // (chunk 0 begins at `begin`, in front of which only empty alignments lie)
    for (uint64_t c = (b - base + FCH - 1) / FCH; base + c * FCH < e && c < n_first; ++c) if (c) first_aln[c] = a;
}

// Resident workgroups walk the tiles with a grid stride, in order: a launch of one short-lived workgroup per tile (10^4 of them
// for an ONT chromosome) was bound by the rate at which workgroups are dispatched, not by memory (1.3 TB/s with the loads alone in
// the kernel).  Tile t's predecessors are taken earlier by the same or by a resident workgroup: the look-back ends.
__global__ __launch_bounds__(FBLOCK)
void cigar_flat_kernel(const uint32_t* __restrict__ cigar, const uint64_t* __restrict__ cig_off, const int32_t* __restrict__ ref_start,
                       uint32_t n_aln, int32_t min_sv, SvxGap* __restrict__ gaps, uint64_t gaps_cap, uint32_t* __restrict__ gap_off,
                       int32_t* __restrict__ stats, Desc d, uint32_t n_tiles, unsigned int* __restrict__ next_tile,
                       const uint32_t* __restrict__ first_aln)
{
    // (tiles are TAKEN in order from a counter, not assigned by index: a workgroup that is not resident yet holds no tile that a
    // resident one could wait for)
    __shared__ uint32_t s_tile;
    for (;;) {
        if (threadIdx.x == 0) s_tile = atomicAdd(next_tile, 1u);
        __syncthreads();
        const uint32_t t = s_tile;
        if (t >= n_tiles) return;
        flat_tile(cigar, cig_off, ref_start, n_aln, min_sv, gaps, gaps_cap, gap_off, stats, d, t, first_aln);
        __syncthreads();                              // (the tile's LDS -- s_tile included -- is free again)
    }
}

}  // namespace

extern "C" size_t svx_cigar_scan_flat_ws_bytes(uint64_t n_words_max)
{
    const size_t groups = ((size_t)((n_words_max + 3 + FCH - 1) / FCH) + 1 + NW - 1) / NW;
    return (5 * groups + 1) * sizeof(unsigned long long) + groups * NW * sizeof(uint32_t);       // (+ the tile counter, + the chunks' first alignments)
}

// svx_cigar_scan for long alignments: the same inputs and outputs (include/svx.h), one pass.  n_words_max: an upper bound of
// d_cig_off[n_aln] - d_cig_off[0] (the launch is sized by it; the offsets themselves are device memory); d_ws:
// svx_cigar_scan_flat_ws_bytes(n_words_max) bytes, 8-byte aligned.
extern "C" int svx_cigar_scan_flat(const uint32_t* d_cigar, const uint64_t* d_cig_off, const int32_t* d_ref_start, uint32_t n_aln,
                                   uint64_t n_words_max, int32_t min_sv, SvxGap* d_gaps, uint64_t gaps_cap, uint32_t* d_gap_off,
                                   int32_t* d_stats, void* d_ws, uint64_t ws_bytes, void* stream)
{
    if (!d_gap_off) return SVX_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (n_aln == 0) return hipMemsetAsync(d_gap_off, 0, sizeof(uint32_t), st) == hipSuccess ? SVX_OK : SVX_ELAUNCH;
    if (!d_cigar || !d_cig_off || !d_ref_start || !d_ws || (!d_gaps && gaps_cap)) return SVX_EINVAL;
    if (d_stats && (reinterpret_cast<uintptr_t>(d_stats) & 15u)) return SVX_EINVAL;
    if ((reinterpret_cast<uintptr_t>(d_ws) & 7u) || (reinterpret_cast<uintptr_t>(d_cigar) & 15u)) return SVX_EINVAL;
    if (ws_bytes < svx_cigar_scan_flat_ws_bytes(n_words_max)) return SVX_EINVAL;
    const size_t chunks = ((size_t)((n_words_max + 3 + FCH - 1) / FCH) + 1 + NW - 1) / NW;        // workgroups of NW chunks
    if (chunks >= (1ull << 28)) return SVX_EINVAL;
    if (hipMemsetAsync(d_ws, 0, (5 * chunks + 1) * sizeof(unsigned long long), st) != hipSuccess) return SVX_ELAUNCH;
    if (hipMemsetAsync(d_gap_off + n_aln, 0, sizeof(uint32_t), st) != hipSuccess) return SVX_ELAUNCH;
    unsigned long long* w = static_cast<unsigned long long*>(d_ws);
    Desc d{w, w + chunks, w + 2 * chunks, w + 3 * chunks, w + 4 * chunks};
    const char* gs = getenv("SVX_FLAT_GRID");
    const unsigned grid = (unsigned)min((size_t)(gs ? atoi(gs) : 768), chunks);       // 256 CUs x 3 resident workgroups of 512 threads
    uint32_t* first_aln = reinterpret_cast<uint32_t*>(w + 5 * chunks + 1);
    hipLaunchKernelGGL(cigar_map_kernel, dim3((n_aln + 255) / 256), dim3(256), 0, st, d_cig_off, n_aln, first_aln, (uint64_t)chunks * NW, n_words_max, d_gap_off);
    hipLaunchKernelGGL(cigar_flat_kernel, dim3(grid), dim3(FBLOCK), 0, st,
                       d_cigar, d_cig_off, d_ref_start, n_aln, min_sv, d_gaps, gaps_cap, d_gap_off, d_stats, d, (uint32_t)chunks,
                       reinterpret_cast<unsigned int*>(w + 5 * chunks), first_aln);
    return hipGetLastError() == hipSuccess ? SVX_OK : SVX_ELAUNCH;
}
