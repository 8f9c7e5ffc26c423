// svx_cigar.hip -- per-alignment CIGAR / segment scan for gfx950 (MI355X).
//
// Replaces the Python CIGAR walk of analyze_inside_align
// (reference src/collection/analyze_reads.py:828-853) and the pysam-derived
// reference_end / query_alignment_start / query_alignment_end for a whole
// batch of alignments whose packed BAM CIGAR words already sit in HBM.
//
// HBM-read bound: 4 B per CIGAR op, 16 B per alignment of CSR/start data,
// 24 B written per long gap (rare).  No atomics on results, output deterministic and
// sorted by (alignment, op) without a sort.  Round 4: THREE kernels (six until then -- count, count_long, retotal, scan,
// offsets, emit: at HiFi sizes a third of a scan's time was the launches' own latency):
//   1. count_kernel  : eight lanes per alignment (eight alignments in flight per wave) stream the CIGAR once in
//                      16-byte quads and reduce the per-alignment spans, clip runs and the number of long gaps.  An
//                      alignment of more than 512 words (ONT) is only started by its eight lanes; its wave finishes it at
//                      the end of the kernel with eight 16-byte loads in flight per lane;
//   2. offsets_kernel: a workgroup per tile of 1024 alignments turns the counts into the CSR offsets d_gap_off -- the
//                      prefix over the tiles by a decoupled look-back in the same launch (one packed 64-bit descriptor per
//                      tile, 64 of them per step) -- and writes the alignments that own a gap (a few % of HiFi reads, most
//                      ONT reads) into a work list, in alignment order;
//   3. emit_kernel   : resident waves walk the work list, one wave per alignment, 256
//                      CIGAR words per step requested two steps ahead: wave prefix sums of
//                      read/ref advance give readPos/refPos at every op, ballot-ranked stores
//                      keep op order.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
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

// Inclusive prefix sum over the wave in the vector ALU (row shifts inside the rows of 16 lanes + the two row broadcasts of gfx9):
// seven data-parallel moves instead of six ds_bpermute round trips through the LDS crossbar (what __shfl_up compiles to here).
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
    uint32_t r = v;
    r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);      // row_shr:1
    r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);      // row_shr:2
    r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x113, 0xf, 0xf, true);      // row_shr:3
    r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0x114, 0xf, 0xe, true);      // row_shr:4, banks 1-3
    r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0x118, 0xf, 0xc, true);      // row_shr:8, banks 2-3
    r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0x142, 0xa, 0xf, false);     // row_bcast:15 into rows 1 and 3
    r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0x143, 0xc, 0xf, false);     // row_bcast:31 into rows 2 and 3
    return r;
}

__device__ inline unsigned wave_sum_u(unsigned v)
{
#pragma unroll
    for (int o = WAVE / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}

// The count pass gives an alignment G lanes that keep Q 16-byte loads in flight each (template parameters: the entry point picks
// <4, 4> for launches whose alignments average at most SHORT_MEAN words -- HiFi: sixteen alignments per wave, whose per-alignment
// instructions are the larger half of the pass -- and <8, 4> otherwise).
constexpr uint32_t LONG_Q = 128, MID_LONG_Q = 256;   // quads of an alignment the count pass's groups handle themselves (its head; longer: frames behind it) -- 512 words,
                                                    // or 1,024 in launches of 257-1,024 words per alignment (there most alignments then have no frames at all: a step
                                                    // of a wave for the 300 words behind a 512-word head cost more than the head)
#ifndef SVX_WIDE_Q
#define SVX_WIDE_Q 4                                 // 16-byte loads in flight per lane of the eight-lane count pass
#endif
constexpr uint64_t SHORT_MEAN = 256;
constexpr uint64_t SHARE_MEAN = 1024;                // mean words per alignment from which the frames get a launch of their own (frames_kernel)
constexpr uint32_t MIN_RANGE_SHIFT = 11;             // a range of frames_kernel: at least one frame's words

// Sum over the G lanes of an alignment, complete in the group's FIRST lane only (the others end with partial sums: nothing
// reads them): row shifts in the vector ALU (the groups lie inside the rows of 16 lanes; a lane beyond the row reads 0) -- three
// fused shift-adds where __shfl_xor was three ds_bpermute round trips, five sums per alignment.
template <int G>
__device__ __forceinline__ unsigned cgroup_sum(unsigned v)
{
    static_assert(G == 4 || G == 8 || G == 16, "groups lie inside a row of 16 lanes");
    if (G >= 16) v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x108, 0xf, 0xf, true);     // row_shl:8
    if (G >= 8)  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x104, 0xf, 0xf, true);     // row_shl:4
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x102, 0xf, 0xf, true);                       // row_shl:2
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xf, 0xf, true);                       // row_shl:1
    return v;
}

// one CIGAR word into the three per-alignment sums (32-bit modular; "0M" is inert).  The count pass is bound by the vector
// ALU, not by memory (PMC, profiles/r04_pmc_cigar.txt: 605 VALU instructions per wave of eight alignments, the SIMD's vector
// port busy for the whole launch), so the instruction count per word is its speed: a signed one-bit field extract turns "is this
// op in the set" into an AND mask in one instruction (v_bfe_i32: 0 or -1), where shift / and / compare / select took three.
// Round 5: the extract's offset is the WORD itself -- the instruction reads the offset's low five bits, op and the length's
// lowest bit, so the sets are written twice (bits 0-8 and 16-24) and `w & 15` is gone; "at least min_sv bases" is one unsigned
// compare of the word with min_sv << 4 (scan_threshold()).
constexpr uint32_t K_REF = 0x018D018Du, K_QRY = 0x01B301B3u, K_GAP = 0x00060006u;      // M D N = X (reference_end) | M I S H = X | I D
__device__ inline uint32_t scan_threshold(int32_t min_sv)
{
    return min_sv <= 0 ? 0u : min_sv >= (1 << 28) ? 0xFFFFFFFFu : (uint32_t)min_sv << 4;      // (0xFFFFFFFF is op 15: in no set)
}
__device__ inline void tally(uint32_t w, uint32_t m16, unsigned& ref_span, unsigned& qlen, unsigned& ngap)
{
    const uint32_t len = w >> 4;
    ref_span += len & (uint32_t)__builtin_amdgcn_sbfe(K_REF, w, 1u);
    qlen += len & (uint32_t)__builtin_amdgcn_sbfe(K_QRY, w, 1u);
    ngap -= (uint32_t)(w >= m16 ? __builtin_amdgcn_sbfe(K_GAP, w, 1u) : 0);                 // I, D of at least min_sv bases
}

// ---- Frames.  An alignment of more than LONG_Q quads (ONT, assembly contigs: 10^3-10^6 ops) is cut into FRAMES of 128 quads
// (512 words) behind its first LONG_Q quads: frame f = quads [q0 + LONG_Q + 128 f, ... + 128) up to the alignment's last whole
// quad.  A frame belongs to ONE alignment, and the count pass leaves its sums -- read advance, reference advance, long gaps --
// in a record of its own, frames[first quad >> 7] (first quads of different frames lie at least 128 quads apart: the records need
// no allocation and no initialisation).  A wave takes four frames in one STEP (2,048 words: eight 16-byte loads per lane, a
// row of sixteen lanes on each frame -- 256 consecutive bytes per load -- so that a frame's sums are row shifts).  The emit pass turns
// the records of an alignment into the positions in front of every frame with one wave prefix sum per 64 frames and walks only
// the frames that hold a long gap -- one in fourteen on ONT data -- where until round 5 it walked every word of every gap owner
// again, one step after the other (the serial position chain).
constexpr int LQUADS = 8;                           // 16-byte loads in flight per lane while a wave takes a step
constexpr uint32_t FRAME_Q = 128, STEP_Q = 512;
__host__ __device__ inline uint64_t frame_slots(uint64_t n_words) { return (n_words >> 9) + 2; }      // records: one per 128 quads of the array
constexpr uint32_t K_SKIP = 0x00080008u;             // N: advances the read position, not the reference position (analyze_reads.py:831-832)

// inclusive sums inside the rows of sixteen lanes (a row's last lane: the row's sum) and, from them, over the wave (lane 63)
__device__ __forceinline__ uint32_t row_incl_scan(uint32_t v)
{
    uint32_t r = v;
    r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);      // row_shr:1
    r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);      // row_shr:2
    r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x113, 0xf, 0xf, true);      // row_shr:3
    r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0x114, 0xf, 0xe, true);      // row_shr:4, banks 1-3
    r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0x118, 0xf, 0xc, true);      // row_shr:8, banks 2-3
    return r;
}
__device__ __forceinline__ uint32_t rows_to_wave(uint32_t r)
{
    r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0x142, 0xa, 0xf, false);     // row_bcast:15 into rows 1 and 3
    r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0x143, 0xc, 0xf, false);     // row_bcast:31 into rows 2 and 3
    return r;
}

// one step of alignment `al` in the count pass: quads [S, qe), at most STEP_Q of them, S a frame's first quad
__device__ __forceinline__ void count_step(const uint4* __restrict__ quads, uint64_t S, uint64_t qe, uint32_t al, uint32_t m16,
                                           uint4* __restrict__ frames, uint64_t n_slots, uint32_t* __restrict__ gap_off, int32_t* __restrict__ stats, int wl)
{
    unsigned r = 0, l = 0, g = 0, x = 0;
    const uint32_t nq = (uint32_t)(qe - S);              // 1 .. STEP_Q
    // every load is issued before the first tally (the fences keep the scheduler from holding some back behind the tallies of
    // others: a round trip each), and none sits in a branch of its own (a load whose value is only used under a condition is
    // moved under it: eight round trips): the last step of an alignment clamps the quad index and masks with AND
    uint4 w[LQUADS];
    uint32_t keep[LQUADS];
    const uint4* __restrict__ p = quads + S;
#pragma unroll
    for (int u = 0; u < LQUADS; ++u) {
        const uint32_t qi = ((uint32_t)wl >> 4) * FRAME_Q + (uint32_t)u * 16u + ((uint32_t)wl & 15u);      // a row of lanes = a frame, 256 consecutive bytes per load
        w[u] = p[min(qi, nq - 1u)];
        keep[u] = qi < nq ? ~0u : 0u;                    // ("0M" is inert)
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < LQUADS; ++u) {
        const uint32_t w4[4] = {w[u].x & keep[u], w[u].y & keep[u], w[u].z & keep[u], w[u].w & keep[u]};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            tally(w4[k], m16, r, l, g);
            x += (w4[k] >> 4) & (uint32_t)__builtin_amdgcn_sbfe(K_SKIP, w4[k], 1u);
        }
        __builtin_amdgcn_sched_barrier(0);               // quad after quad: left alone the scheduler spreads the 32 tallies over 160 registers
    }
    r = row_incl_scan(r); l = row_incl_scan(l); g = row_incl_scan(g); x = row_incl_scan(x);
    const uint64_t F = S + (uint64_t)(wl >> 4) * FRAME_Q;                   // the frame of this lane's row
    if ((wl & 15) == 15 && F < qe && (F >> 7) < n_slots)                    // (n_slots: the caller's word count, checked by the count pass)
        frames[F >> 7] = make_uint4(l + x, r - x, g, 0u);                   // M I N S H = X | M D = X | long gaps
    r = rows_to_wave(r); l = rows_to_wave(l); g = rows_to_wave(g);
    if (wl == WAVE - 1) {
        // (atomics: performed in the L2, behind the leaders' stores to the same words, which were acknowledged before the first step)
        if (stats) { atomicAdd(reinterpret_cast<unsigned*>(stats) + 4 * (size_t)al, r); atomicAdd(reinterpret_cast<unsigned*>(stats) + 4 * (size_t)al + 3, l); }
        if (g) atomicAdd(&gap_off[al], g);
    }
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
// descriptor of a count workgroup for the look-back: [flag:2 | owners:30 | gaps:32]
constexpr unsigned long long D_AGG = 1ull << 62, D_INC = 2ull << 62, D_FLAG = 3ull << 62;
__device__ inline unsigned long long d_pack(uint32_t gaps, uint32_t owners) { return (unsigned long long)(owners & 0x3fffffffu) << 32 | gaps; }

template <int G, int Q, bool SHARE>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(5, 8)))
void count_kernel(const uint32_t* __restrict__ cigar, const uint64_t* __restrict__ cig_off,
                  uint32_t n_aln, int32_t min_sv, uint32_t* __restrict__ gap_off, int32_t* __restrict__ stats,
                  unsigned long long* __restrict__ desc, uint32_t n_tiles, uint4* __restrict__ frames, uint64_t n_words,
                  uint4* __restrict__ range_aln, uint32_t range_shift, uint32_t long_q)
{
    // (the look-back descriptors of the offsets pass behind this kernel start out empty: zeroed here instead of by a memset
    // launch of its own -- 5 us of a 70 us scan; there are more count workgroups than tiles)
    if (threadIdx.x == 0 && blockIdx.x < n_tiles) desc[blockIdx.x] = 0ull;
    // (the total: written with atomicMax by the offsets pass, see there.  A caller whose array holds more words than it said
    // gets SVX_SCAN_FAILED instead: the frame records are sized by that number)
    if (threadIdx.x == 0 && blockIdx.x == 0) gap_off[n_aln] = cig_off[n_aln] > n_words ? SVX_SCAN_FAILED : 0u;
    const uint64_t n_slots = frame_slots(n_words);
    const int sub = threadIdx.x & (G - 1);
    const int wl = threadIdx.x & (WAVE - 1);
    const int gshift = wl & ~(G - 1);
    const uint32_t a = blockIdx.x * (BLOCK / G) + (threadIdx.x / G);
    const bool live = a < n_aln;
    const uint32_t m16 = scan_threshold(min_sv);
    const uint64_t total = cig_off[n_aln];
    const uint64_t full = total >> 2;                    // quads that lie entirely inside the array
    const uint32_t ac = min(a, n_aln - 1);               // (n_aln > 0: the entry point launches nothing otherwise)
    const uint64_t b = cig_off[ac], e_next = cig_off[ac + 1], e = live ? e_next : b;       // (both loads in flight at once)
    const long long n = (long long)(e - b);
    const uint64_t q0 = b >> 2, q_end = min((e + 3) >> 2, full);
    // an alignment of more than LONG_Q quads (ONT: 10^3-10^5 ops) is only STARTED by its eight lanes -- for them it would be
    // a chain of hundreds of memory round trips that the whole launch waits for -- and finished by the whole wave at the END
    // of this kernel, when everything else is stored and few registers are live (in the middle of it the deep load
    // pipeline's registers cost every HiFi launch a third of its occupancy: 163 instead of 74; until round 4 a list + a
    // kernel of its own).  All sums are modular and additive: the corrections (computed from q_end) and the two partial
    // sums simply add up.
    const bool some = q0 < q_end;
    const bool is_long = some && q_end - q0 > (uint64_t)long_q;
    const uint32_t nq = some ? (is_long ? long_q : (uint32_t)(q_end - q0)) : 0u;      // quads taken here
    // the words at either end for the clip runs, the neighbours' words inside the first / last quad (lanes 0-2
    // look before b and from e on), the alignment's words beyond the last whole quad of the array (at most 3,
    // last alignments only), then the quads.  Everything is addressed from the alignment's own first word / quad / end with
    // 32-bit offsets (round 5: the 64-bit compares and selects around every load were a sixth of the kernel's vector instructions).
    const uint32_t* __restrict__ cb = cigar + b;
    const uint32_t* __restrict__ ce = cigar + e;
    const long long it = n - 1 - sub;
    const uint32_t w_head = sub < n ? cb[sub] : 0u;
    const uint32_t w_tail = it >= 0 ? ce[-1 - sub] : 0u;
    const int lead_in = (int)(b & 3ull);                                 // words of the neighbour in front inside the first quad
    const int trail_in = (int)(uint32_t)(4 * q_end - e);                 // words behind e inside the last quad taken: -3 .. 3, as the low word says
    const uint64_t lb = max(b, 4 * full);
    const int loose_n = e > lb ? (int)(uint32_t)(e - lb) : 0;            // <= 3: total < 4 * full + 4
    const uint32_t w_before = (some && sub < lead_in) ? cb[sub - lead_in] : 0u;
    const uint32_t w_after = (some && sub < trail_in) ? ce[sub] : 0u;
    const uint32_t w_loose = sub < loose_n ? cigar[lb + sub] : 0u;
    unsigned ref_span = 0, qlen = 0, ngap = 0;
    const uint4* __restrict__ quads = reinterpret_cast<const uint4*>(cigar);
    const uint4* __restrict__ qb = quads + q0;
    for (uint32_t q = sub; q < nq; q += Q * G) {
        uint4 w[Q];
#pragma unroll
        for (int u = 0; u < Q; ++u) w[u] = qb[min(q + u * G, nq - 1)];        // clamped, dropped below when out of range
#pragma unroll
        for (int u = 0; u < Q; ++u) {
            const bool in = q + u * G < nq;
            tally(in ? w[u].x : 0u, m16, ref_span, qlen, ngap); tally(in ? w[u].y : 0u, m16, ref_span, qlen, ngap);
            tally(in ? w[u].z : 0u, m16, ref_span, qlen, ngap); tally(in ? w[u].w : 0u, m16, ref_span, qlen, ngap);
        }
    }
    tally(w_loose, m16, ref_span, qlen, ngap);
    {
        unsigned fr = 0, fq = 0, fn = 0;
        tally(w_before, m16, fr, fq, fn);
        tally(w_after, m16, fr, fq, fn);
        ref_span -= fr; qlen -= fq; ngap -= fn;
    }
    ngap = cgroup_sum<G>(ngap);
    ref_span = cgroup_sum<G>(ref_span);
    qlen = cgroup_sum<G>(qlen);
    if (live && sub == 0) gap_off[a] = ngap;             // (turned into an offset by the offsets pass)
    if (stats) {
        unsigned lead = 0, trail = 0;
        long long n_lead = 0;                                  // words in the leading clip run
        {
            const unsigned m = (unsigned)(__ballot(sub < n && is_clip(w_head & 15u)) >> gshift) & ((1u << G) - 1u);
            int run = __ffs((int)~m) - 1;                      // 0..G
            if (sub < run) lead += w_head >> 4;
            n_lead = run;
            for (long long base = G; run == G && base < n; base += G) {     // longer runs: rare
                const long long i = base + sub;
                const uint32_t w = i < n ? cigar[b + i] : 0u;
                const unsigned mm = (unsigned)(__ballot(i < n && is_clip(w & 15u)) >> gshift) & ((1u << G) - 1u);
                run = __ffs((int)~mm) - 1;
                if (sub < run) lead += w >> 4;
                n_lead += run;
            }
        }
        if (n_lead < n) {                                      // an all-clip CIGAR is all leading clip
            const unsigned m = (unsigned)(__ballot(it >= n_lead && is_clip(w_tail & 15u)) >> gshift) & ((1u << G) - 1u);
            int run = __ffs((int)~m) - 1;
            if (sub < run) trail += w_tail >> 4;
            for (long long base = G; run == G && base < n - n_lead; base += G) {
                const long long i = n - 1 - base - sub;
                const uint32_t w = i >= n_lead ? cigar[b + i] : 0u;
                const unsigned mm = (unsigned)(__ballot(i >= n_lead && is_clip(w & 15u)) >> gshift) & ((1u << G) - 1u);
                run = __ffs((int)~mm) - 1;
                if (sub < run) trail += w >> 4;
            }
        }
        lead = cgroup_sum<G>(lead);
        trail = cgroup_sum<G>(trail);
        if (live && sub == 0) {
            int4 s4;
            s4.x = (int)ref_span; s4.y = (int)lead; s4.z = (int)trail; s4.w = (int)qlen;
            reinterpret_cast<int4*>(stats)[a] = s4;
        }
    }
    // ---- the frames of the long alignments.  SHARE = false (launches of short alignments, where a long one is an exception): the
    // wave takes the frames of its own long alignments, one after the other.  SHARE = true (ONT, contigs: every alignment is long
    // and their lengths spread over two orders of magnitude -- a launch that leaves the frames to the waves that own the alignments
    // ends with its slowest wave, at twice the mean; shared by the workgroup: 1.5 x): the frames are left to frames_kernel, which
    // cuts the ARRAY into ranges of equal size, and all this pass adds is the map the ranges start from -- the alignment that
    // holds a range's first word, written by that alignment's leader (a range is 2^range_shift words).
    if constexpr (!SHARE) {
        unsigned long long lm = __ballot(is_long && sub == 0);
        if (lm) __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0): the leaders' stores above are acknowledged by the L2 before the atomics go there
        while (lm) {
            const int src = __ffsll((long long)lm) - 1;
            lm &= lm - 1;
            const uint64_t qa = __shfl(q0, src, WAVE) + long_q, qb = __shfl(q_end, src, WAVE);
            const uint32_t al = __shfl(a, src, WAVE);
#pragma clang loop unroll(disable)
            for (uint64_t S = qa; S < qb; S += STEP_Q) count_step(quads, S, min(S + STEP_Q, qb), al, m16, frames, n_slots, gap_off, stats, wl);
        }
    } else {
        if (live && sub == 0)
            for (uint64_t r = a ? (b + (1ull << range_shift) - 1) >> range_shift : b >> range_shift; (r << range_shift) < e && r <= (n_words >> range_shift); ++r) range_aln[r] = make_uint4(a, (uint32_t)b, (uint32_t)(b >> 32), (uint32_t)min((uint64_t)n, 0xFFFFFFFFull));      // (the first alignment: the range its first word lies in too)
    }
}

// The frames of a launch of long alignments (count_kernel<.., true> in front of it).  A wave takes one range of the array
// -- 2^range_shift words: every wave the same number of frames, give or take one -- and in it every frame that STARTS there:
// from the alignment that holds the range's first word (the count pass's map) it walks the offsets, 64 per load, with scalar
// arithmetic; four frames are one count_step.  The leaders' stores of the count pass are complete (a launch of their own lies
// between), the sums of an alignment's frames reach them as atomics.
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(5, 8)))
void frames_kernel(const uint32_t* __restrict__ cigar, const uint64_t* __restrict__ cig_off, uint32_t n_aln, int32_t min_sv,
                   uint32_t* __restrict__ gap_off, int32_t* __restrict__ stats, uint4* __restrict__ frames, uint64_t n_words,
                   const uint4* __restrict__ range_aln, uint32_t range_shift, uint32_t long_q)
{
    const int wl = threadIdx.x & (WAVE - 1);
    const uint64_t range = (uint64_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6)));
    const uint64_t total = cig_off[n_aln], full = total >> 2;
    const uint64_t lo = range << (range_shift - 2), hi = lo + (1ull << (range_shift - 2));      // in quads
    if ((range << range_shift) >= total || ((range + 1) << range_shift) <= cig_off[0]) return;      // behind the last alignment / in front of the first (a window of a larger array)
    const uint32_t m16 = scan_threshold(min_sv);
    const uint64_t n_slots = frame_slots(n_words);
    const uint4* __restrict__ quads = reinterpret_cast<const uint4*>(cigar);
    // one alignment: the frames of it that start in [lo, hi); false: its first frame, and so every later alignment's, starts behind the range
    auto take = [&](uint32_t a, uint64_t b, uint64_t e) {
        const uint64_t q0 = b >> 2, q_end = min((e + 3) >> 2, full), qa = q0 + long_q;
        if (qa >= hi) return false;
        if (!(q0 < q_end && q_end - q0 > (uint64_t)long_q)) return true;
        const uint64_t k = qa >= lo ? 0ull : (lo - qa + STEP_Q - 1) / STEP_Q;
        const uint64_t stop = min(hi, q_end);
#pragma clang loop unroll(disable)
        for (uint64_t S = qa + k * STEP_Q; S < stop; S += STEP_Q) count_step(quads, S, min(S + STEP_Q, q_end), a, m16, frames, n_slots, gap_off, stats, wl);
        return true;
    };
    // the alignment that holds the range's first word comes with its offsets (the map's entry: one load in front of the frame's),
    // the ones behind it -- if the range reaches that far -- from the offsets array, 64 per load
    uint4 ent = range_aln[range];
    ent.x = (uint32_t)__builtin_amdgcn_readfirstlane((int)ent.x); ent.y = (uint32_t)__builtin_amdgcn_readfirstlane((int)ent.y);
    ent.z = (uint32_t)__builtin_amdgcn_readfirstlane((int)ent.z); ent.w = (uint32_t)__builtin_amdgcn_readfirstlane((int)ent.w);
    uint32_t a = ent.x;
    {
        const uint64_t b = (uint64_t)ent.y | (uint64_t)ent.z << 32;
        uint64_t e = b + ent.w;
        if (ent.w == 0xFFFFFFFFu) e = cig_off[a + 1];      // (an alignment of 2^32 words or more)
        if (!take(a, b, e)) return;
        if ((e >> 2) + long_q >= hi) return;             // (the next alignment starts at or behind e)
        ++a;
    }
    for (;;) {
        const uint32_t idx = min(a + (uint32_t)wl, n_aln);
        const uint64_t o0 = cig_off[idx], o1 = cig_off[min(idx + 1u, n_aln)];
        for (int i = 0; i < WAVE; ++i, ++a) {
            if (a >= n_aln) return;
            const uint64_t b = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)o0, i) | (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(o0 >> 32), i) << 32;
            const uint64_t e = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)o1, i) | (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(o1 >> 32), i) << 32;
            if (!take(a, b, e)) return;
        }
    }
}

// Offsets pass with a decoupled look-back, one workgroup per tile of 1024 alignments (four per thread): the tile's counts ->
// its totals (gaps, owners), published as one packed 64-bit descriptor; the workgroup's first wave looks back over the
// tiles in front of it (64 descriptors per step) for the exclusive prefix; a workgroup scan on top of it turns the counts
// into the CSR offsets d_gap_off, and the (few) alignments that own a long gap go to the work list (alignment, first slot),
// in alignment order.  (Until round 4 a one-workgroup scan kernel over per-workgroup totals sat between the count pass and
// this one -- plus two launches for the long alignments: six dependent launches per scan.  The look-back inside the COUNT
// pass -- one launch less still -- was measured and is slower: 62 k count workgroups each end in a wave that waits.)
constexpr int OTILE = 4 * BLOCK;

__global__ __launch_bounds__(BLOCK)
void offsets_kernel(uint32_t n_aln, uint32_t* __restrict__ gap_off, unsigned long long* __restrict__ desc, uint2* __restrict__ totals,
                    uint2* __restrict__ work)
{
    constexpr int NW = BLOCK / WAVE;
    __shared__ uint32_t s_wave[2 * NW];
    __shared__ uint32_t s_ex[2];
    const int t = threadIdx.x, wl = t & (WAVE - 1), wv = t >> 6;
    const uint32_t v = blockIdx.x;
    const uint32_t a0 = v * OTILE + 4u * (uint32_t)t;
    uint32_t c[4];
    if (a0 + 3 < n_aln) { const uint4 q = *reinterpret_cast<const uint4*>(gap_off + a0); c[0] = q.x; c[1] = q.y; c[2] = q.z; c[3] = q.w; }
    else {
#pragma unroll
        for (int u = 0; u < 4; ++u) c[u] = a0 + u < n_aln ? gap_off[a0 + u] : 0u;
    }
    const uint32_t mine_g = c[0] + c[1] + c[2] + c[3];
    const uint32_t mine_o = (uint32_t)(c[0] != 0u) + (uint32_t)(c[1] != 0u) + (uint32_t)(c[2] != 0u) + (uint32_t)(c[3] != 0u);
    const uint32_t inc_g = wave_incl_scan(mine_g), inc_o = wave_incl_scan(mine_o);
    if (wl == WAVE - 1) { s_wave[wv] = inc_g; s_wave[NW + wv] = inc_o; }
    __syncthreads();
    uint32_t pre_g = 0, pre_o = 0, tot_g = 0, tot_o = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        if (w < wv) { pre_g += s_wave[w]; pre_o += s_wave[NW + w]; }
        tot_g += s_wave[w]; tot_o += s_wave[NW + w];
    }
    if (wv == 0) {
        uint32_t ex_g = 0, ex_o = 0;                     // totals of the tiles in front
        if (v > 0) {
            if (wl == 0) __hip_atomic_store(&desc[v], D_AGG | d_pack(tot_g, tot_o), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // Workgroups are dispatched in the order of their index (every XCD takes its share in order), so the ones in
            // front have at least started and none of them waits for this one: the spin ends.  (Bounded all the same: a
            // descriptor that never arrives would otherwise hang the device.)
            long long j = (long long)v - 1;
            for (;;) {
                const long long i = j - wl;
                unsigned long long d = D_INC;            // in front of the first tile: an inclusive prefix of nothing
                if (i >= 0) {
                    int spins = 0;
                    do { d = __hip_atomic_load(&desc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((d & D_FLAG) == 0 && ++spins < (1 << 20));
                    if ((d & D_FLAG) == 0) atomicMax(&gap_off[n_aln], SVX_SCAN_FAILED);      // never seen on this hardware; loud if it ever is
                }
                const unsigned long long incl = __ballot((d & D_FLAG) == D_INC);
                const int stop = incl ? __ffsll((long long)incl) - 1 : WAVE - 1;      // the nearest tile whose inclusive prefix is known
                const uint32_t g = wl <= stop ? (uint32_t)d : 0u, o = wl <= stop ? (uint32_t)(d >> 32) & 0x3fffffffu : 0u;
                ex_g += wave_sum_u(g); ex_o += wave_sum_u(o);
                if (incl) break;
                j -= WAVE;
            }
        }
        if (wl == 0) {
            __hip_atomic_store(&desc[v], D_INC | d_pack(ex_g + tot_g, ex_o + tot_o), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_ex[0] = ex_g; s_ex[1] = ex_o;
        }
    }
    __syncthreads();
    uint32_t off = s_ex[0] + pre_g + inc_g - mine_g, rank = s_ex[1] + pre_o + inc_o - mine_o;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const uint32_t a = a0 + u;
        if (a < n_aln) {
            gap_off[a] = off;
            if (c[u]) work[rank++] = make_uint2(a, off);
            off += c[u];
            if (a == n_aln - 1) { atomicMax(&gap_off[n_aln], off); *totals = make_uint2(off, rank); }
        }
    }
}

// Emit pass: resident waves take the work list with a grid stride, one wave per alignment, 4 consecutive
// CIGAR words per lane (256 words per step).  Lane-local sums + a wave prefix sum of the read / reference
// advance (32-bit modular = the truncated 64-bit positions) give readPos / refPos at every word; ballot
// ranks keep the stores in op order.
__global__ __launch_bounds__(BLOCK)
void emit_kernel(const uint32_t* __restrict__ cigar, const uint64_t* __restrict__ cig_off,
                 const int32_t* __restrict__ ref_start, int32_t min_sv, SvxGap* __restrict__ gaps, uint64_t gaps_cap,
                 const uint2* __restrict__ totals, const uint2* __restrict__ work, uint32_t n_aln, const uint4* __restrict__ frames, uint64_t n_slots, uint32_t long_q)
{
    const int lane = threadIdx.x & (WAVE - 1);
    const uint64_t full = cig_off[n_aln] >> 2;             // quads that lie entirely inside the array
    const uint32_t n_work = totals->y;                     // alignments owning a long gap
    const uint32_t n_waves = gridDim.x * (BLOCK / WAVE);
    const unsigned long long below = (1ull << lane) - 1ull;
    // An owner's item, offsets and reference start are the head of a chain of dependent loads (item -> offsets -> records and
    // words -> ...) that the wave would walk afresh for each of its owners: they are requested ahead -- the item two owners, the
    // offsets one owner in front of their use -- at wave-uniform addresses (scalar loads into scalar registers: no vector
    // register is held for them).
    const uint32_t k0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6)));
    auto item_of = [&](uint32_t k) { return k < n_work ? work[k] : make_uint2(0u, 0u); };
    uint2 item = item_of(k0), item_next = item_of(k0 + n_waves);
    uint64_t off_b = cig_off[item.x], off_e = cig_off[item.x + 1];
    int32_t start_ref = ref_start[item.x];
    for (uint32_t k = k0; k < n_work; k += n_waves) {
        const uint32_t a = item.x;
        uint32_t dst = item.y;
        const uint64_t b = off_b;
        const long long n = (long long)(off_e - b);
        uint32_t read_pos = 0, ref_pos = (uint32_t)start_ref;
        // (the next owner's offsets and the item behind it: in flight while this owner is walked; n_aln > 0 here, and item_of()
        // answers alignment 0 behind the list's end: valid addresses)
        item = item_next;
        item_next = item_of(k + 2 * n_waves);
        off_b = cig_off[item.x]; off_e = cig_off[item.x + 1];
        start_ref = ref_start[item.x];
        long long lim = n;                                     // (the narrow steps mask the words at and behind it)
        // the words of a step are requested two steps ahead (three register sets, statically rotated): the positions
        // are a serial chain over the steps, so an ultra-long read (ONT: 10^4-10^5 ops, 40-400 steps) would otherwise
        // pay one memory round trip per step
        auto load = [&](long long j0, uint32_t (&w)[4]) {
            const long long j = j0 + 4 * lane;
#pragma unroll
            for (int u = 0; u < 4; ++u) w[u] = j + u < lim ? cigar[b + j + u] : 6u;      // "0P": advances nothing
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
            const uint32_t ir = wave_incl_scan(tr), irf = wave_incl_scan(tf);     // inclusive wave prefix sums of the lane totals
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
            read_pos += (uint32_t)__builtin_amdgcn_readlane((int)ir, WAVE - 1);
            ref_pos += (uint32_t)__builtin_amdgcn_readlane((int)irf, WAVE - 1);
        };
        constexpr long long STEP = 4 * WAVE;
        auto narrow = [&](long long jlo, long long jhi) {      // the words [jlo, jhi) of the alignment, 256 per step
            lim = jhi;
            uint32_t w0[4], w1[4], w2[4];
            load(jlo, w0);
            load(jlo + STEP, w1);
            for (long long j0 = jlo; j0 < jhi; j0 += 3 * STEP) {
                load(j0 + 2 * STEP, w2);
                step(j0, w0);
                if (j0 + STEP >= jhi) break;
                load(j0 + 3 * STEP, w0);
                step(j0 + STEP, w1);
                if (j0 + 2 * STEP >= jhi) break;
                load(j0 + 4 * STEP, w1);
                step(j0 + 2 * STEP, w2);
            }
        };
        // the count pass's own test (there: is_long), from the same offsets
        const uint64_t e = b + (uint64_t)n, q0 = b >> 2, q_end = min((e + 3) >> 2, full);
        const bool is_long = q0 < q_end && q_end - q0 > (uint64_t)long_q;
        // A long alignment (ONT, assembly contigs).  Behind its first LONG_Q quads lie its frames, whose sums the count pass has
        // left in `frames`: 64 records per step become the positions and the output slot in front of every frame (one set of wave
        // prefix sums), and only a frame that holds a long gap is walked -- two narrow steps from the frame's own positions.
        // (Round 4 walked every word of a gap owner in 256-word steps, round 5's first form in 2,048-word steps: a serial chain of
        // one memory round trip per step, 40-400 steps for an ONT read, all of whose words were read a second time.)
        // The pieces of an alignment -- its head, its frames with a gap, what lies behind the array's last whole quad (at most
        // three words) -- go through ONE call of narrow() in a loop: four inlined copies cost 124 registers against 76.
        auto excl = [&](uint32_t v, uint32_t& total) {
            const uint32_t inc = wave_incl_scan(v);
            total = (uint32_t)__builtin_amdgcn_readlane((int)inc, WAVE - 1);
            return inc - v;
        };
        const uint64_t qa = q0 + long_q;
        const uint32_t n_frames = is_long ? ((uint32_t)(q_end - qa) + FRAME_Q - 1) / FRAME_Q : 0u;
        const long long j_tail = 4 * (long long)q_end - (long long)b;
        auto record = [&](uint32_t k) {
            const uint64_t slot = (qa >> 7) + (uint64_t)k;
            return (k < n_frames && slot < n_slots) ? frames[slot] : make_uint4(0u, 0u, 0u, 0u);
        };
        uint4 rec = record((uint32_t)lane);                    // (requested before the head is walked)
        uint32_t base_r = 0, base_f = 0, base_d = 0;           // the positions and the slot in front of the current batch of 64 frames
        uint32_t er = 0, ef = 0, eg = 0, Tr = 0, Tf = 0, Tg = 0, k0 = 0;
        unsigned long long hot = 0;
        auto batch = [&]() {
            er = excl(rec.x, Tr); ef = excl(rec.y, Tf); eg = excl(rec.z, Tg);
            hot = __ballot(rec.z != 0u);
        };
        int stage = 0;                                         // 0: the head (a short alignment: all of it), 1: frames, 2: the tail
        long long jlo = 0, jhi = is_long ? 4 * (long long)qa - (long long)b : n;
        for (;;) {
            narrow(jlo, jhi);
            if (!is_long || stage == 2) break;
            if (stage == 0) { base_r = read_pos; base_f = ref_pos; base_d = dst; stage = 1; batch(); }
            while (!hot) {
                base_r += Tr; base_f += Tf; base_d += Tg; k0 += WAVE;
                if (k0 >= n_frames) break;
                rec = record(k0 + (uint32_t)lane);
                batch();
            }
            if (hot) {
                const int i = __ffsll((long long)hot) - 1;
                hot &= hot - 1;
                jlo = 4 * (long long)(qa + (uint64_t)(k0 + (uint32_t)i) * FRAME_Q) - (long long)b;      // the frame's first word, counted from the alignment's
                jhi = min(jlo + 4 * (long long)FRAME_Q, min(n, j_tail));
                read_pos = base_r + (uint32_t)__builtin_amdgcn_readlane((int)er, i); ref_pos = base_f + (uint32_t)__builtin_amdgcn_readlane((int)ef, i);
                dst = base_d + (uint32_t)__builtin_amdgcn_readlane((int)eg, i);
                continue;
            }
            if (j_tail >= n) break;
            stage = 2; read_pos = base_r; ref_pos = base_f; dst = base_d; jlo = j_tail; jhi = n;
        }
    }
}

// workspace: [desc: one 64-bit descriptor per tile of the offsets pass (zeroed by the count pass)][totals: uint2] | [work: uint2 per alignment] | [frames: uint4 per 128 quads]
inline size_t ws_totals_offset(uint32_t n_aln)
{
    const size_t tiles = ((size_t)n_aln + OTILE - 1) / OTILE;
    return tiles * sizeof(unsigned long long);
}
inline size_t ws_work_offset(uint32_t n_aln)
{
    return (ws_totals_offset(n_aln) + sizeof(uint2) + 255) & ~(size_t)255;
}
inline size_t ws_frames_offset(uint32_t n_aln)
{
    return (ws_work_offset(n_aln) + (size_t)n_aln * sizeof(uint2) + 255) & ~(size_t)255;
}
inline size_t ws_ranges_offset(uint32_t n_aln, uint64_t n_words)
{
    return ws_frames_offset(n_aln) + (size_t)frame_slots(n_words) * sizeof(uint4);
}

}  // namespace

extern "C" size_t svx_cigar_scan_ws_bytes(uint32_t n_aln, uint64_t n_words)
{
    return ws_ranges_offset(n_aln, n_words) + ((size_t)(n_words >> MIN_RANGE_SHIFT) + 2) * sizeof(uint4);
}

extern "C" int svx_cigar_scan(const uint32_t* d_cigar, const uint64_t* d_cig_off,
                              const int32_t* d_ref_start, uint32_t n_aln, uint64_t n_words, int32_t min_sv,
                              SvxGap* d_gaps, uint64_t gaps_cap, uint32_t* d_gap_off,
                              int32_t* d_stats, void* d_ws, uint64_t ws_bytes, uint32_t flags, void* stream)
{
    if (!d_gap_off) return SVX_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (n_aln == 0) {
        return hipMemsetAsync(d_gap_off, 0, sizeof(uint32_t), st) == hipSuccess ? SVX_OK : SVX_ELAUNCH;
    }
    if (!d_cigar || !d_cig_off || !d_ref_start || !d_ws || (!d_gaps && gaps_cap)) return SVX_EINVAL;
    if (d_stats && (reinterpret_cast<uintptr_t>(d_stats) & 15u)) return SVX_EINVAL;
    if ((reinterpret_cast<uintptr_t>(d_ws) & 15u) || (reinterpret_cast<uintptr_t>(d_cigar) & 15u) || (reinterpret_cast<uintptr_t>(d_gap_off) & 15u)) return SVX_EINVAL;
    if (n_aln >= (1u << 30)) return SVX_EINVAL;            // (the look-back descriptors keep the owner count in 30 bits)
    if (ws_bytes < svx_cigar_scan_ws_bytes(n_aln, n_words)) return SVX_EINVAL;
    if (flags & ~(uint32_t)(SVX_SCAN_LANES4 | SVX_SCAN_LANES8 | SVX_SCAN_SHARED | SVX_SCAN_UNSHARED)) return SVX_EINVAL;
    unsigned long long* desc = static_cast<unsigned long long*>(d_ws);
    uint2* totals = reinterpret_cast<uint2*>(static_cast<char*>(d_ws) + ws_totals_offset(n_aln));
    uint2* work = reinterpret_cast<uint2*>(static_cast<char*>(d_ws) + ws_work_offset(n_aln));
    uint4* frames = reinterpret_cast<uint4*>(static_cast<char*>(d_ws) + ws_frames_offset(n_aln));
    const uint64_t n_slots = frame_slots(n_words);
    const uint32_t n_tiles = (n_aln + OTILE - 1) / OTILE;
    // the count pass's shape, by the launch's mean words per alignment (the flags fix it: A/B runs and tests): short alignments
    // (HiFi) four lanes each and a long one finished by its own wave; long ones eight lanes, their frames shared by the workgroup
    const bool narrow = (flags & SVX_SCAN_LANES4) ? true : (flags & SVX_SCAN_LANES8) ? false : n_words <= SHORT_MEAN * (uint64_t)n_aln;
    const bool share = (flags & SVX_SCAN_SHARED) ? true : (flags & SVX_SCAN_UNSHARED) ? false : n_words > SHARE_MEAN * (uint64_t)n_aln;
    const uint32_t long_q = (!narrow && !share) ? MID_LONG_Q : LONG_Q;
    uint4* range_aln = reinterpret_cast<uint4*>(static_cast<char*>(d_ws) + ws_ranges_offset(n_aln, n_words));
    uint32_t range_shift = MIN_RANGE_SHIFT;              // ranges of frames_kernel: one frame's words -- measured (ONT-shaped launch): 275 us, 286 / 296 / 332 us with ranges of 2 / 4 / 16 frames
    static const int forced_shift = getenv("SVX_RANGE_SHIFT") ? atoi(getenv("SVX_RANGE_SHIFT")) : 0;      // (experiment)
    if (forced_shift >= (int)MIN_RANGE_SHIFT) range_shift = (uint32_t)forced_shift;
    static const unsigned count_lds = getenv("SVX_COUNT_LDS") ? (unsigned)atoi(getenv("SVX_COUNT_LDS")) : 0u;      // (experiment: caps the workgroups per CU)
#define SVX_COUNT(G, Q, S) hipLaunchKernelGGL((count_kernel<G, Q, S>), dim3((n_aln + BLOCK / G - 1) / (BLOCK / G)), dim3(BLOCK), count_lds, st, \
                                              d_cigar, d_cig_off, n_aln, min_sv, d_gap_off, d_stats, desc, n_tiles, frames, n_words, range_aln, range_shift, long_q)
    if (narrow) { if (share) SVX_COUNT(4, 4, true); else SVX_COUNT(4, 4, false); }
    else        { if (share) SVX_COUNT(8, SVX_WIDE_Q, true); else SVX_COUNT(8, SVX_WIDE_Q, false); }
#undef SVX_COUNT
    if (share) {
        const uint64_t n_ranges = (n_words >> range_shift) + 1;
        hipLaunchKernelGGL(frames_kernel, dim3((uint32_t)((n_ranges + BLOCK / WAVE - 1) / (BLOCK / WAVE))), dim3(BLOCK), 0, st,
                           d_cigar, d_cig_off, n_aln, min_sv, d_gap_off, d_stats, frames, n_words, range_aln, range_shift, long_q);
    }
    hipLaunchKernelGGL(offsets_kernel, dim3(n_tiles), dim3(BLOCK), 0, st, n_aln, d_gap_off, desc, totals, work);
    // resident waves (8 workgroups per CU at most); small inputs get one wave per 4 alignments
    const uint32_t emit_blocks = min(2048u, (n_aln + 15u) / 16u);
    hipLaunchKernelGGL(emit_kernel, dim3(emit_blocks), dim3(BLOCK), 0, st, d_cigar, d_cig_off, d_ref_start, min_sv, d_gaps, gaps_cap,
                       totals, work, n_aln, frames, n_slots, long_q);
    return hipGetLastError() == hipSuccess ? SVX_OK : SVX_ELAUNCH;
}
