// svx_inflate2.hip -- BGZF inflate on gfx950 in two kernels: parallel Huffman decoding per block, then the LZ77 copies.
//
// The lane-per-block kernel of svx_inflate.hip pays ~300 instructions and two or three dependent memory round trips per
// decoded symbol (420 B of LDS per lane hold its Huffman tables: 1.5 waves per SIMD, nobody to hide them): 64 ms per block
// whatever the launch holds.  What is sequential in DEFLATE is the LZ77 window, not the Huffman code -- a decoder started
// at an arbitrary bit re-synchronises with the true token boundaries within a few dozen bits (measured on BAM data:
// median 50 bits, 99 % within 500; tools/exp/spec_inflate_sim.cpp is the CPU model of everything below).  So:
//
//   A. bgzf_tokens_kernel -- one WAVE per BGZF block.  Per DEFLATE block the wave parses the header (uniform, scalar bit
//      buffer) and builds ONE set of look-up tables in LDS (10-bit literal/length root + sub-tables, 8-bit distance root:
//      entries carry base value and extra-bit count); then the compressed bits are cut into chunks of 64 segments of
//      SEG_BITS bits, staged in LDS, and lane i decodes segment i -- pass S1 from the segment's first bit (speculative: its
//      last token boundary is almost always a true one), pass S2 from its predecessor's end, repeated for the lanes whose
//      start moved until nothing moves (lane 0 is true, so lane k is true after <= k rounds; in practice 1.15); S2 also sizes
//      every segment's output.  Pass S3 decodes once more and TRANSCODES: the tokens leave as byte-aligned LZ sequences,
//      u32 header [literals:8 | match length:9 | distance - 1:15] + the literal bytes, every lane writing its own part of
//      the block's stream (offsets by wave prefix sums).  No LZ77 copy happens here: nothing in this kernel waits for a
//      store.
//   B. bgzf_lz_kernel -- one LANE per block walks its sequence stream: literal runs and matches are copied in steps of up
//      to eight bytes through a ring of the lane's last 256 output bytes in LDS (near matches are read from it, output leaves
//      in whole aligned 16-byte chunks; svx_lz_core.hpp).  No tables: every block of a launch is resident at once and a turn
//      costs ~100 instructions instead of ~300; ~10 k turns per 64 KB block instead of 36 k.
//
// Byte-identical to zlib by construction (tests/test_gpu_inflate.py runs this pair through every case the other kernels
// pass).  Blocks whose sequence stream would not fit its slot (1.5 x ISIZE + 1 KB: only pathological streams -- hundreds of
// tiny DEFLATE blocks -- get there) are left to the wave-per-block kernel of svx_inflate.hip by the entry point.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/svx.h"
#include "svx_lz_core.hpp"

#ifndef SVX_TOK_DBG
#define SVX_TOK_DBG 0                // measurements only (wrong output): 1 = the Huffman tables are built twice, 2 = no transcode pass, 4 = no S1 pass
#endif

namespace {

constexpr int LANES = 64;
#ifndef SVX_TOK_LB
#define SVX_TOK_LB 9
#define SVX_TOK_DB 7
#endif
#ifndef SVX_TOK_WAVES
#define SVX_TOK_WAVES 4
#endif
// Root bits of the literal / length and the distance table, and the entries they can need with their sub-tables: zlib's ENOUGH
// (inftrees.h: 852 entries for 286 symbols at 9 root bits, 592 for 30 symbols at 6, codes of up to 15 bits; a sub-table spans
// the longest code below its root entry, there as here) -- more root bits need no more sub-table entries.  Round 5: 9 / 7 bits
// instead of 10 / 8 and the builder's scratch inside the chunk buffer: 9.9 KB of LDS per wave instead of 14.1, SIXTEEN waves
// per CU instead of eleven -- the kernel's time follows its waves almost linearly (34 / 42 / 56 / 83 ms at 11 / 8 / 6 / 4 per CU).
constexpr int LB = SVX_TOK_LB, DB = SVX_TOK_DB;
constexpr int LIT_CAP = (1 << LB) + (LB >= 9 ? 340 : 512), DIST_CAP = (1 << DB) + 528;
#ifndef SVX_TOK_SEG
#define SVX_TOK_SEG 512
#endif
constexpr int SEG_BITS = SVX_TOK_SEG;                // compressed bits per lane and chunk
constexpr int CHUNK_WORDS = LANES * SEG_BITS / 32 + 8;                    // staged dwords: the chunk + what the last tokens may read behind it
enum { K_BAD = 0, K_LIT = 1, K_LEN = 2, K_EOB = 3, K_SUB = 4 };
enum { INF_OK = 0, INF_BAD_TYPE = 1, INF_BAD_STORED = 2, INF_BAD_TABLE = 3, INF_BAD_CODE = 4, INF_OUT_OVERRUN = 5, INF_IN_OVERRUN = 6, INF_SHORT = 7,
       INF_BAD_DIST = 8, INF_TOKENS_OVERFLOW = 10 };

__constant__ uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__constant__ uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__constant__ uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__constant__ uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__constant__ uint8_t CLEN_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// table entry: bits 0..3 code bits consumed at this level, 4..7 kind, 8.. payload
//   literal: 8..15 byte | length: 8..16 base, 17..19 extra bits | distance: 8..22 base, 23..26 extra bits | sub-table: 8..19 start, 20..23 bits
struct LitPayload {
    __device__ __forceinline__ uint32_t operator()(int s) const
    {
        if (s < 256) return (uint32_t)K_LIT << 4 | (uint32_t)s << 8;
        if (s == 256) return (uint32_t)K_EOB << 4;
        if (s - 257 >= 29) return (uint32_t)K_BAD << 4;
        return (uint32_t)K_LEN << 4 | (uint32_t)LEN_BASE[s - 257] << 8 | (uint32_t)LEN_EXTRA[s - 257] << 17;
    }
};
struct DistPayload {
    __device__ __forceinline__ uint32_t operator()(int s) const
    {
        return s >= 30 ? (uint32_t)K_BAD << 4 : ((uint32_t)K_LEN << 4 | (uint32_t)DIST_BASE[s] << 8 | (uint32_t)DIST_EXTRA[s] << 23);
    }
};
struct ClenPayload { __device__ __forceinline__ uint32_t operator()(int s) const { return (uint32_t)K_LIT << 4 | (uint32_t)s << 8; } };

struct Lds {
    uint32_t lit[LIT_CAP];                           // 3.3 KB
    uint32_t dist[DIST_CAP];                         // 2.6 KB (also: the code-length code's table, scratch of the literal table's build)
    uint32_t chunk[CHUNK_WORDS];                     // 4 KB: the compressed bytes of the chunk in hand
    // what the table builder reads and its scratch live in the chunk buffer (staged only when the tables stand): its first KB is
    // build_tables' tmp, then
    __device__ __forceinline__ uint8_t* lens() { return reinterpret_cast<uint8_t*>(chunk) + 1024; }       // [320] code lengths: literal / length symbols at 0, distance symbols at 288
    __device__ __forceinline__ uint8_t* cl() { return reinterpret_cast<uint8_t*>(chunk) + 1344; }         // [32] lengths of the code-length code
    __device__ __forceinline__ uint16_t* code() { return reinterpret_cast<uint16_t*>(chunk) + 704; }      // [320] canonical code of every symbol
};
static_assert(sizeof(Lds) <= 10240 && LANES * SEG_BITS / 8 >= 2048, "sixteen waves per CU; the builder's scratch fits the chunk buffer");

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ void lds_fence() { __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier(); }   // lgkmcnt(0); one wave: in order

// uniform bit reader for the block headers: the stream's next 256 bytes in one VGPR (lane i = dword i), the bit buffer scalar
struct HeadReader {
    const uint32_t* words;
    uint32_t cur, nxt;
    uint64_t chunk_next, word_end;
    int k;
    uint64_t buf;
    int cnt;
    uint32_t fed;                                    // bits handed out since seek()
    __device__ __forceinline__ uint32_t fetch(uint64_t first) { const uint64_t i = first + threadIdx.x; return i < word_end ? words[i] : 0u; }
    __device__ __forceinline__ void refill()
    {
        if (cnt <= 32) {
            const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)cur, k);
            if (++k == 64) { cur = nxt; nxt = fetch(chunk_next); chunk_next += 64; k = 0; }
            buf |= (uint64_t)v << cnt;
            cnt += 32;
        }
    }
    // position the reader at bit `bit` of the stream that starts at byte address `base` and holds `bytes` bytes
    __device__ __forceinline__ void seek(const uint8_t* base, uint32_t bytes, uint32_t bit)
    {
        const uintptr_t a = reinterpret_cast<uintptr_t>(base) + (bit >> 3);
        words = reinterpret_cast<const uint32_t*>(a & ~(uintptr_t)3);
        word_end = ((reinterpret_cast<uintptr_t>(base) + bytes + 3) >> 2) - (reinterpret_cast<uintptr_t>(words) >> 2) + 1;   // (+ a dword of the footer: readable)
        cur = fetch(0); nxt = fetch(64); chunk_next = 128;
        k = 0; buf = 0; cnt = 0;
        refill();
        const int skip = (int)(a & 3) * 8 + (int)(bit & 7);
        buf >>= skip; cnt -= skip;
        fed = 0;
    }
    __device__ __forceinline__ uint32_t bits(int n)            // n <= 16
    {
        refill();
        const uint32_t v = (uint32_t)buf & ((1u << n) - 1u);
        buf >>= n; cnt -= n; fed += (uint32_t)n;
        return v;
    }
};

// Canonical Huffman tables of one alphabet, built by the wave: code lengths lens[0..n) (LDS) -> root table of 2^root entries +
// sub-tables for longer codes, behind it.  `tmp`: 2^root bytes of scratch in LDS.  -> entries used, 0 = over-subscribed / too large.
template <class Payload>
__device__ int build_tables(const uint8_t* lens, int n, int root, uint32_t* tab, int cap, uint16_t* code_of, uint8_t* tmp, Payload payload)
{
    const int lane = threadIdx.x;
    const unsigned long long lt = (1ull << lane) - 1ull;
    int count[16];
#pragma unroll
    for (int l = 0; l < 16; ++l) count[l] = 0;
    for (int s0 = 0; s0 < n; s0 += LANES) {
        const int l = s0 + lane < n ? (int)lens[s0 + lane] : 0;
#pragma unroll
        for (int q = 1; q < 16; ++q) count[q] += (int)__popcll(__ballot(l == q));
    }
    int next[16], left = 1, code = 0;
    next[0] = 0;
#pragma unroll
    for (int l = 1; l <= 15; ++l) {
        left = (left << 1) - count[l];
        code = (code + (l > 1 ? count[l - 1] : 0)) << 1;
        next[l] = code;
    }
    if (left < 0) return 0;
    // codes in symbol order within each length
    for (int s0 = 0; s0 < n; s0 += LANES) {
        const int l = s0 + lane < n ? (int)lens[s0 + lane] : 0;
        int mine = 0;
#pragma unroll
        for (int q = 1; q < 16; ++q) {
            const unsigned long long m = __ballot(l == q);
            if (l == q) mine = next[q] + (int)__popcll(m & lt);
            next[q] += (int)__popcll(m);
        }
        if (s0 + lane < n) code_of[s0 + lane] = (uint16_t)mine;
    }
    const int nroot = 1 << root;
    for (int i = lane; i < nroot; i += LANES) { tab[i] = 0; tmp[i] = 0; }
    lds_fence();
    // sub-table bits per root prefix: the longest code below it.  (The symbols' lengths and codes are read 64 at a time into
    // a register and handed round with v_readlane: one LDS round trip per 64 symbols instead of two per symbol.)
    for (int s0 = 0; s0 < n; s0 += LANES) {
        const int lv = s0 + lane < n ? (int)lens[s0 + lane] : 0, cv = s0 + lane < n ? (int)code_of[s0 + lane] : 0;
        unsigned long long longs = __ballot(lv > root);
        while (longs) {
            const int j = __ffsll((long long)longs) - 1;
            longs &= longs - 1;
            const int l = __builtin_amdgcn_readlane(lv, j);
            const uint32_t r = __brev((uint32_t)__builtin_amdgcn_readlane(cv, j)) >> (32 - l);
            const int pre = (int)(r & (uint32_t)(nroot - 1));
            if (lane == 0 && (int)tmp[pre] < l - root) tmp[pre] = (uint8_t)(l - root);
            lds_fence();
        }
    }
    // allocate: exclusive prefix of 2^bits over the root entries (each lane owns nroot / 64 consecutive entries)
    int top = nroot;
    {
        const int per = nroot / LANES, i0 = lane * per;
        int mine = 0;
        for (int i = 0; i < per; ++i) { const int b = tmp[i0 + i]; mine += b ? 1 << b : 0; }
        int inc = mine;
#pragma unroll
        for (int o = 1; o < LANES; o <<= 1) { const int u = __shfl_up(inc, o, LANES); if (lane >= o) inc += u; }
        const int total = __shfl(inc, LANES - 1, LANES);
        if (nroot + total > cap) return 0;
        int at = nroot + inc - mine;
        for (int i = 0; i < per; ++i) {
            const int b = tmp[i0 + i];
            if (b) { tab[i0 + i] = (uint32_t)K_SUB << 4 | (uint32_t)at << 8 | (uint32_t)b << 20; at += 1 << b; }
        }
        for (int i = nroot + lane; i < nroot + total; i += LANES) tab[i] = 0;
        top = nroot + total;
    }
    lds_fence();
    // entries: symbol after symbol (the used ones), the wave fills a symbol's replicas
    for (int s0 = 0; s0 < n; s0 += LANES) {
        const int lv = s0 + lane < n ? (int)lens[s0 + lane] : 0, cv = s0 + lane < n ? (int)code_of[s0 + lane] : 0;
        unsigned long long used = __ballot(lv != 0);
        while (used) {
            const int j = __ffsll((long long)used) - 1;
            used &= used - 1;
            const int s = s0 + j, l = __builtin_amdgcn_readlane(lv, j);
            const uint32_t r = __brev((uint32_t)__builtin_amdgcn_readlane(cv, j)) >> (32 - l);
            if (l <= root) {
                const uint32_t e = payload(s) | (uint32_t)l;
                for (uint32_t i = r + ((uint32_t)lane << l); i < (uint32_t)nroot; i += (uint32_t)LANES << l) tab[i] = e;
            } else {
                const uint32_t p = tab[r & (uint32_t)(nroot - 1)];
                const uint32_t start = (p >> 8) & 0xfffu, sb = (p >> 20) & 15u;
                const uint32_t e = payload(s) | (uint32_t)(l - root);
                for (uint32_t i = (r >> root) + ((uint32_t)lane << (l - root)); i < (1u << sb); i += (uint32_t)LANES << (l - root)) tab[start + i] = e;
            }
        }
    }
    lds_fence();
    return top;
}

// 64 bits of the chunk from bit `rel` of the staged words (rel < 32 * (CHUNK_WORDS - 2)) as two dwords: two funnel shifts
// (v_alignbit_b32, full rate).  A token is at most 15 + 5 code and extra bits of the literal / length alphabet -- all inside the
// first dword -- and 15 + 13 of the distance alphabet, which start at bit <= 20: one more funnel shift brings them into one
// dword as well.  (Round 4, first version: one 64-bit value and 64-bit shifts throughout -- fourteen of them per token, at
// half the rate of the 32-bit ones: the tokens kernel is bound by the vector ALU, profiles/r04_pmc_sq_inflate.txt.)
struct Bits { uint32_t lo, hi; };
__device__ __forceinline__ Bits peek(const uint32_t* chunk, uint32_t rel)
{
    const uint32_t w = rel >> 5, sh = rel & 31u;
    const uint32_t d0 = chunk[w], d1 = chunk[w + 1], d2 = chunk[w + 2];
    return Bits{__builtin_amdgcn_alignbit(d1, d0, sh), __builtin_amdgcn_alignbit(d2, d1, sh)};
}

struct Tok { uint32_t kind, used, val, len, dist; };

__device__ __forceinline__ Tok token(const Lds& t, uint32_t rel)
{
    const Bits b = peek(t.chunk, rel);
    uint32_t e = t.lit[b.lo & ((1u << LB) - 1u)];
    uint32_t used;
    if (((e >> 4) & 15u) == K_SUB) {
        e = t.lit[((e >> 8) & 0xfffu) + ((b.lo >> LB) & ((1u << ((e >> 20) & 15u)) - 1u))];
        used = LB + (e & 15u);
    } else used = e & 15u;
    Tok k{(e >> 4) & 15u, used, (e >> 8) & 255u, 0u, 0u};
    if (k.kind != K_LEN) return k;
    const uint32_t xb = (e >> 17) & 7u;
    k.len = ((e >> 8) & 511u) + ((b.lo >> used) & ((1u << xb) - 1u));          // used + xb <= 20
    used += xb;
    const uint32_t db32 = __builtin_amdgcn_alignbit(b.hi, b.lo, used);          // the 32 bits from bit `used` (<= 20) on
    uint32_t d = t.dist[db32 & ((1u << DB) - 1u)];
    uint32_t dused;
    if (((d >> 4) & 15u) == K_SUB) {
        d = t.dist[((d >> 8) & 0xfffu) + ((db32 >> DB) & ((1u << ((d >> 20) & 15u)) - 1u))];
        dused = DB + (d & 15u);
    } else dused = d & 15u;
    if (((d >> 4) & 15u) != K_LEN) { k.kind = K_BAD; k.used = used + dused; return k; }
    const uint32_t nb = (d >> 23) & 15u;
    k.dist = ((d >> 8) & 0x7fffu) + ((db32 >> dused) & ((1u << nb) - 1u));      // dused + nb <= 28
    k.used = used + dused + nb;
    return k;
}

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

__device__ __forceinline__ uint32_t wave_excl_sum(uint32_t v, uint32_t* total)
{
    const uint32_t inc = wave_incl_scan(v);
    *total = (uint32_t)__builtin_amdgcn_readlane((int)inc, LANES - 1);
    return inc - v;
}

// where block b's sequence stream lies in the workspace: 1.5 x its inflated bytes + 1 KB
__host__ __device__ inline uint64_t stream_base(const uint64_t* dst_off, uint32_t b) { const uint64_t d = dst_off[b] - dst_off[0]; return d + (d >> 1) + 1024ull * b; }

__device__ __forceinline__ void put32(uint8_t* p, uint32_t v) { __builtin_memcpy(p, &v, 4); }

// SPLIT: the stream format of bgzf_lz_wave_kernel -- the sequences' u32 headers as an array that grows upwards from the slot's
// first 4-byte boundary, their literal bytes downwards from the slot's end (literal j of the block at end[-1 - j]): 64 headers are
// one coalesced load, the positions of 64 sequences' outputs and literals two wave prefix sums.  stream_cnt[b] = (sequences,
// literal bytes).  Not SPLIT: headers and literals interleaved (bgzf_lz_kernel walks them), stream_cnt[b].x = the stream's bytes.
template <bool SPLIT>
__global__ __launch_bounds__(LANES) __attribute__((amdgpu_waves_per_eu(SVX_TOK_WAVES, 8)))
void bgzf_tokens_kernel(const uint8_t* __restrict__ comp, const uint64_t* __restrict__ src_off, const uint32_t* __restrict__ src_len,
                        const uint64_t* __restrict__ dst_off, uint32_t n_blocks, uint8_t* __restrict__ streams, uint2* __restrict__ stream_cnt,
                        uint32_t* __restrict__ status)
{
    __shared__ Lds t;
    const uint32_t b = blockIdx.x;
    const int lane = threadIdx.x;
    const uint8_t* src = comp + src_off[b];
    const uint32_t nbytes = src_len[b], nbits = nbytes * 8u;
    const uint32_t isize = (uint32_t)(dst_off[b + 1] - dst_off[b]);
    uint8_t* stream = streams + stream_base(dst_off, b);
    const uint32_t cap = (uint32_t)(stream_base(dst_off, b + 1) - stream_base(dst_off, b));
    const uint32_t pad = SPLIT ? (uint32_t)(-(intptr_t)reinterpret_cast<uintptr_t>(stream)) & 3u : 0u;
    uint8_t* const hdr0 = stream + pad;              // (SPLIT) the header array; the literals end at stream + cap
    uint8_t* const lit_end = stream + cap;
    uint32_t P = 0, W = 0, Q = pad;                  // (uniform) bit position, bytes decoded, stream bytes written
    uint32_t H = 0, Lc = 0;                          // (SPLIT, uniform) sequences and literal bytes written
    int err = INF_OK;
    bool last = isize == 0;                          // an empty block (the EOF marker): nothing to decode
    HeadReader hr;
    while (!last && err == INF_OK) {
        hr.seek(src, nbytes, P);
        last = hr.bits(1) != 0;
        const uint32_t type = hr.bits(2);
        if (type == 0) {                             // stored: to the byte boundary, LEN, ~LEN, bytes -> literal-only sequences
            hr.bits((int)((P + hr.fed) & 7u ? 8u - ((P + hr.fed) & 7u) : 0u));
            const uint32_t len = hr.bits(16), nlen = hr.bits(16);
            P += hr.fed;
            if ((len ^ nlen) != 0xffffu) { err = INF_BAD_STORED; break; }
            if (W + len > isize) { err = INF_OUT_OVERRUN; break; }
            if (P + 8u * len > nbits) { err = INF_IN_OVERRUN; break; }
            const uint32_t nseq = (len + 254u) / 255u;
            if (Q + len + 4u * nseq > cap) { err = INF_TOKENS_OVERFLOW; break; }
            const uint8_t* from = src + (P >> 3);
            for (uint32_t s = lane; s < nseq; s += LANES) {
                const uint32_t n = min(255u, len - 255u * s);
                if (SPLIT) {
                    put32(hdr0 + 4u * (H + s), n);
                    uint8_t* q = lit_end - 1 - (Lc + 255u * s);
                    for (uint32_t i = 0; i < n; ++i) q[-(int)i] = from[255u * s + i];
                } else {
                    uint8_t* q = stream + Q + 259u * s;
                    put32(q, n);
                    for (uint32_t i = 0; i < n; ++i) q[4 + i] = from[255u * s + i];
                }
            }
            Q += len + 4u * nseq; W += len; P += 8u * len; H += nseq; Lc += len;
            continue;
        }
        if (type == 3) { err = INF_BAD_TYPE; break; }
        int nlen = 288, ndist = 30;
        if (type == 1) {                             // fixed code (RFC 1951 3.2.6)
            for (int s = lane; s < 288; s += LANES) t.lens()[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
            if (lane < 32) t.lens()[288 + lane] = lane < 30 ? 5 : 0;
            lds_fence();
        } else {                                     // dynamic code (3.2.7)
            nlen = (int)hr.bits(5) + 257; ndist = (int)hr.bits(5) + 1;
            const int ncode = (int)hr.bits(4) + 4;
            if (nlen > 286 || ndist > 30) { err = INF_BAD_TABLE; break; }
            if (lane < 32) t.cl()[lane] = 0;
            lds_fence();
            for (int i = 0; i < ncode; ++i) { const uint32_t v = hr.bits(3); if (lane == 0) t.cl()[CLEN_ORDER[i]] = (uint8_t)v; }
            lds_fence();
            // the code-length code: a 7-bit table in the distance table's place (scratch: the chunk buffer)
            if (!build_tables(t.cl(), 19, 7, t.dist, 128, t.code(), reinterpret_cast<uint8_t*>(t.chunk), ClenPayload{})) { err = INF_BAD_TABLE; break; }
            for (int s = lane; s < 320; s += LANES) t.lens()[s] = 0;
            lds_fence();
            int i = 0;
            while (i < nlen + ndist) {
                hr.refill();
                const uint32_t e = (uint32_t)uni((int)t.dist[(uint32_t)hr.buf & 127u]);
                if (((e >> 4) & 15u) != K_LIT) { err = INF_BAD_TABLE; break; }
                hr.buf >>= (e & 15u); hr.cnt -= (int)(e & 15u); hr.fed += e & 15u;
                const int sym = (int)((e >> 8) & 255u);
                int value = sym, rep = 1;
                if (sym == 16) {
                    if (i == 0) { err = INF_BAD_TABLE; break; }
                    const int j = i - 1;
                    value = uni((int)t.lens()[j < nlen ? j : 288 + (j - nlen)]);
                    rep = 3 + (int)hr.bits(2);
                } else if (sym == 17) { value = 0; rep = 3 + (int)hr.bits(3); }
                else if (sym == 18) { value = 0; rep = 11 + (int)hr.bits(7); }
                if (i + rep > nlen + ndist) { err = INF_BAD_TABLE; break; }
                for (int r = lane; r < rep; r += LANES) { const int x = i + r; t.lens()[x < nlen ? x : 288 + (x - nlen)] = (uint8_t)value; }
                lds_fence();
                i += rep;
            }
            if (err != INF_OK) break;
            if (uni((int)t.lens()[256]) == 0) { err = INF_BAD_TABLE; break; }
        }
        P += hr.fed;
#if SVX_TOK_DBG & 1
        build_tables(t.lens(), nlen, LB, t.lit, LIT_CAP, t.code(), reinterpret_cast<uint8_t*>(t.dist), LitPayload{});
        build_tables(t.lens() + 288, ndist, DB, t.dist, DIST_CAP, t.code(), reinterpret_cast<uint8_t*>(t.chunk), DistPayload{});
#endif
        if (!build_tables(t.lens(), nlen, LB, t.lit, LIT_CAP, t.code(), reinterpret_cast<uint8_t*>(t.dist), LitPayload{}) ||
            !build_tables(t.lens() + 288, ndist, DB, t.dist, DIST_CAP, t.code(), reinterpret_cast<uint8_t*>(t.chunk), DistPayload{})) { err = INF_BAD_TABLE; break; }
        // ---- the block's tokens, chunk by chunk
        bool eob = false;
        while (!eob && err == INF_OK) {
            if (P >= nbits) { err = INF_IN_OVERRUN; break; }
            // stage the chunk: dword w of the buffer = bytes [4w, 4w + 4) from the byte that holds bit P
            const uint32_t byte0 = P >> 3, bit0 = byte0 * 8u;
            for (int w = lane; w < CHUNK_WORDS; w += LANES) {
                uint32_t v = 0;
                const uint32_t at = byte0 + 4u * (uint32_t)w;
                if (at + 4u <= nbytes + 8u) __builtin_memcpy(&v, src + at, 4);      // (the 8-byte footer behind the payload is readable)
                t.chunk[w] = v;
            }
            lds_fence();
            const uint32_t q0 = P + (uint32_t)lane * SEG_BITS, q1 = q0 + SEG_BITS;   // this lane's segment
            // S1: from the segment's first bit
            uint32_t p = q0;
#if SVX_TOK_DBG & 4
            p = q1;
#endif
            while (__any(p < q1 && p < nbits)) {
                if (p < q1 && p < nbits) { const Tok k = token(t, p - bit0); p += k.kind == K_BAD ? 1u : k.used; }
            }
            // S2: from the predecessor's end; again for the lanes whose start moved
            uint32_t start = (uint32_t)__shfl_up((int)p, 1, LANES);
            if (lane == 0) start = P;
            uint32_t f = 0, olen = 0, enc = 0, lit = 0, flag = 0, nsq = 0;
            bool dirty = true;
            int first = LANES;
            for (;;) {
                bool run = dirty;
                if (dirty) { p = start; olen = 0; enc = 0; lit = 0; flag = 0; nsq = 0; }
                while (__any(run)) {
                    // (one branch around the decode, the bookkeeping behind it with selects: written as the nest of ifs it is,
                    // every path back into the loop carried half a dozen register moves -- a third of the loop's vector
                    // instructions, in a kernel the vector ALU bounds)
                    const bool go = run && p < q1 && p < nbits;
                    Tok k{K_BAD, 0u, 0u, 0u, 0u};
                    if (go) k = token(t, p - bit0);
                    const bool bad = run && p < q1 && (p >= nbits || k.kind == K_BAD);
                    const bool adv = go && k.kind != K_BAD;
                    const bool eob = adv && k.kind == K_EOB, isl = adv && k.kind == K_LIT, ism = adv && k.kind == K_LEN;
                    p += adv ? k.used : 0u;
                    const bool full = isl && lit == 255u;
                    enc += ism ? 4u + lit : full ? 259u : 0u;
                    if (SPLIT) nsq += (ism || full) ? 1u : 0u;
                    lit = isl ? (full ? 1u : lit + 1u) : ism ? 0u : lit;
                    olen += ism ? k.len : isl ? 1u : 0u;
                    flag = bad ? 2u : eob ? 1u : flag;
                    const bool stop = run && (!go || bad || eob);        // (!go: the segment's end, or the stream's)
                    f = stop ? p : f;
                    run = run && !stop;
                }
                const unsigned long long fm = __ballot(flag != 0);
                first = fm ? __ffsll((long long)fm) - 1 : LANES;
                const uint32_t prev = (uint32_t)__shfl_up((int)f, 1, LANES);
                dirty = lane > 0 && lane <= first && prev != start;
                if (dirty) start = prev;
                if (!__any(dirty)) break;
            }
            if (first < LANES && uni((int)__shfl((int)flag, first, LANES)) == 2) { err = INF_BAD_CODE; break; }
            const bool valid = lane <= first;
            if (lit) { enc += 4u + lit; nsq += 1u; }
            uint32_t tot_o, tot_e, tot_h = 0;
            const uint32_t o = W + wave_excl_sum(valid ? olen : 0u, &tot_o);
            const uint32_t qoff = Q + wave_excl_sum(valid ? enc : 0u, &tot_e);
            const uint32_t hoff = SPLIT ? H + wave_excl_sum(valid ? nsq : 0u, &tot_h) : 0u;      // this lane's first sequence / first literal
            const uint32_t loff = SPLIT ? Lc + (qoff - Q) - 4u * (hoff - H) : 0u;
            if (W + tot_o > isize) { err = INF_OUT_OVERRUN; break; }
            if (Q + tot_e > cap) { err = INF_TOKENS_OVERFLOW; break; }
            // S3: transcode
#if SVX_TOK_DBG & 2
            if (false)
#endif
            {
                uint8_t* hdr = SPLIT ? hdr0 + 4u * hoff : stream + qoff;
                uint8_t* lp = lit_end - loff;                    // (SPLIT) one behind the next literal's place
                uint32_t w = o, nl = 0;
                bool bad_dist = false;
                p = start;
                bool run = valid;
                while (__any(run)) {                           // (flattened like S2: the stores are the only branches)
                    const bool go = run && p < f;
                    Tok k{K_EOB, 0u, 0u, 0u, 0u};
                    if (go) k = token(t, p - bit0);
                    p += k.used;
                    const bool isl = go && k.kind == K_LIT, ism = go && k.kind == K_LEN;
                    const bool full = isl && nl == 255u;
                    if (full) put32(hdr, 255u);
                    hdr += full ? (SPLIT ? 4 : 259) : 0;
                    nl = full ? 0u : nl;
                    if (SPLIT) { lp -= isl ? 1 : 0; if (isl) *lp = (uint8_t)k.val; }
                    else if (isl) hdr[4 + nl] = (uint8_t)k.val;
                    if (ism) put32(hdr, nl | k.len << 8 | (k.dist - 1u) << 17);
                    const bool far = ism && k.dist > w;          // a distance beyond the start of the output
                    bad_dist = bad_dist || far;
                    hdr += ism ? (SPLIT ? 4 : 4 + nl) : 0;
                    nl = ism ? 0u : nl + (isl ? 1u : 0u);
                    w += ism ? k.len : isl ? 1u : 0u;
                    run = (isl || ism) && !far;                  // (not: the segment's end, the end-of-block code, a bad distance)
                }
                if (valid && nl) put32(hdr, nl);
                if (__any(bad_dist)) { err = INF_BAD_DIST; break; }
            }
            W += tot_o; Q += tot_e;
            if (SPLIT) { Lc += tot_e - 4u * tot_h; H += tot_h; }
            if (first < LANES) { eob = true; P = (uint32_t)__shfl((int)f, first, LANES); }
            else P = (uint32_t)__shfl((int)f, LANES - 1, LANES);
        }
    }
    if (err == INF_OK && W != isize) err = INF_SHORT;
    if (lane == 0) { status[b] = (uint32_t)err; stream_cnt[b] = SPLIT ? make_uint2(H, Lc) : make_uint2(Q, 0u); }
}

// ------------------------------------------------------------------------------------------------------------------
// B: one lane per block copies its sequences (svx_lz_core.hpp: the loop itself, shared with the CPU model).  Every lane keeps the
// last 256 bytes of its output in LDS -- near matches never touch memory, output leaves in whole aligned 16-byte chunks.
constexpr int LZ_LANES = 64, RING_STRIDE = svx_lz::RING + 16;      // (+ 16: lanes at the same ring offset fall into different banks)

__global__ __launch_bounds__(LZ_LANES)
void bgzf_lz_kernel(const uint8_t* __restrict__ streams, const uint2* __restrict__ stream_cnt, const uint64_t* __restrict__ dst_off,
                    uint32_t n_blocks, uint8_t* out, uint32_t* __restrict__ status)
{
    // (dynamic although constant: with static LDS the compiler raises the kernel's VGPR allocation to what its LDS-limited
    // occupancy leaves room for -- 129 registers instead of the 66 in use -- and the convolutions this kernel runs next to
    // cannot have them: svx_cnn.hip, encode_conv1_kernel)
    extern __shared__ __attribute__((aligned(16))) uint8_t rings[];
    const uint32_t b = blockIdx.x * LZ_LANES + threadIdx.x;
    if (b >= n_blocks) return;
    if (status[b] != 0) return;
    const int err = svx_lz::decode_block(streams + stream_base(dst_off, b), stream_cnt[b].x, out, dst_off[b], dst_off[b + 1],
                                         rings + threadIdx.x * RING_STRIDE);
    if (err != svx_lz::LZ_OK) status[b] = (uint32_t)err;
}


// ------------------------------------------------------------------------------------------------------------------
// B', round 5: one WAVE per block.  The lane-per-block kernel above needs ~25 ms for a block whatever the launch holds (one lane
// walks the block's ~9,500 sequences, each a chain of dependent steps), so every group of chromosomes -- the first one of a job
// above all: the pipeline behind waits for it -- pays that latency, and its dribbling 16-byte stores and far-match loads move
// five times the output through the fabric.  Here the wave keeps the block's last 32 KB of output -- DEFLATE's whole window --
// in LDS, indexed by the output's own address, so that a match never reads memory; 64 headers are one coalesced load and two
// wave prefix sums give every sequence its output and literal positions; the sequences are then executed in order, all lanes
// on one -- literal bytes from a staged window of the literal stream (filled 1 KB at a time, a load ahead), match bytes from the
// ring (an overlapping match reads its period: source = start + i mod distance, all of it written before) --, and the output
// leaves in whole aligned 16-byte chunks, 4 KB at a time, read back from the ring: every byte of the stream is fetched once and
// every byte of the output written once.  ~0.5-1 ms per block: a launch of n blocks takes n / 1024 rounds of that (36 KB of LDS
// per wave: four waves per CU), where the lane kernel takes its 25 ms + 0.1 ms per 1,000 blocks -- the entry point picks.
constexpr int WAVE_LZ_BELOW = 12000;
constexpr uint32_t WRING = 32768, WMASK = WRING - 1, WSTAGE = 4096, WFLUSH = 4096;

__global__ __launch_bounds__(LANES)
void bgzf_lz_wave_kernel(const uint8_t* __restrict__ streams, const uint2* __restrict__ stream_cnt, const uint64_t* __restrict__ dst_off,
                         uint32_t n_blocks, uint8_t* out, uint32_t* __restrict__ status)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t wlds[];   // [ring: WRING][literal window: WSTAGE][the batch's sequences: 1 KB]
    uint8_t* const ring = wlds;
    uint8_t* const stage = wlds + WRING;
    const uint32_t b = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    if (status[b] != 0) return;
    const uint64_t lo = dst_off[b], hi = dst_off[b + 1];
    if (hi == lo) return;
    const uint32_t nseq = stream_cnt[b].x, nlit = stream_cnt[b].y;
    const uint8_t* slot = streams + stream_base(dst_off, b);
    const uint32_t cap = (uint32_t)(stream_base(dst_off, b + 1) - stream_base(dst_off, b));
    const uint32_t* hdr = reinterpret_cast<const uint32_t*>(slot + ((uint32_t)(-(intptr_t)reinterpret_cast<uintptr_t>(slot)) & 3u));
    // literal j lies at lit_end[-1 - j]; in 16-byte chunks counted downwards from the chunk that holds lit_end - 1 ... : with
    // t = j + 16 - r (r = lit_end mod 16; r = 0: the chunk above is never needed), chunk t >> 4 is the 16 bytes at a16 - 16 (t >> 4)
    // and the byte is its 15 - (t & 15)-th -- staged at (chunk * 16) mod WSTAGE, literal j is the window's byte (t ^ 15) mod WSTAGE
    const uint8_t* lit_end = slot + cap;
    const uint32_t r = (uint32_t)reinterpret_cast<uintptr_t>(lit_end) & 15u;
    const uint8_t* a16 = lit_end - r;
    auto chunk_ptr = [&](uint32_t c) {
        const uint8_t* q = a16 - 16ull * c;
        return reinterpret_cast<const uint4*>(q < streams ? streams : q);          // (the first block's window may start below its slot: never used)
    };
    uint32_t staged = 0;                                 // chunks in the window
    uint4 pend = *chunk_ptr(lane);                       // chunks [staged, staged + 64) on their way
    auto need_chunk = [&](uint32_t cmax) {               // the window holds chunk cmax afterwards
        while (cmax >= staged) {
            *reinterpret_cast<uint4*>(stage + ((16u * (staged + lane)) & (WSTAGE - 1))) = pend;
            staged += LANES;
            pend = *chunk_ptr(staged + lane);
        }
    };
    uint64_t W = lo, flushed = lo;                       // (uniform) next output byte, everything below is in memory
    uint32_t L = 0;                                      // literals consumed
    int err = svx_lz::LZ_OK;
    auto flush_to = [&](uint64_t upto) {                 // [flushed, upto) leaves the ring; upto: a multiple of 16, or hi
        if (flushed < upto && (flushed & 15ull)) {       // the block's first bytes, up to the first boundary: another block's lie in front
            const uint64_t e = min(upto, (flushed + 15ull) & ~15ull);
            if (flushed + lane < e) out[flushed + lane] = ring[(uint32_t)(flushed + lane) & WMASK];
            flushed = e;
        }
        for (uint64_t a = flushed + 16ull * lane; a + 16 <= upto; a += 16ull * LANES)
            *reinterpret_cast<uint4*>(out + a) = *reinterpret_cast<const uint4*>(ring + ((uint32_t)a & WMASK));
        const uint64_t whole = flushed + ((upto - flushed) & ~15ull);
        if (whole + lane < upto) out[whole + lane] = ring[(uint32_t)(whole + lane) & WMASK];     // (upto == hi: the last, partial chunk)
        flushed = upto;
    };
    // one sequence, all lanes on it (long literal runs, long or overlapping matches, whatever a step below cannot take)
    auto careful = [&](uint32_t hk, uint64_t w, uint32_t l0) {
        const uint32_t knl = hk & 255u, kml = (hk >> 8) & 511u, kd = (hk >> 17) + 1u;
        if (knl) {
            const uint32_t t0 = l0 + 16u - r;
            need_chunk((t0 + knl - 1u) >> 4);
            for (uint32_t i = lane; i < knl; i += LANES)
                ring[(uint32_t)(w + i) & WMASK] = stage[((t0 + i) ^ 15u) & (WSTAGE - 1)];
        }
        if (kml) {
            const uint64_t wm = w + knl;
            if (kd > wm - lo) { err = svx_lz::LZ_BAD_DIST; return; }
            const uint32_t from = (uint32_t)(wm - kd);
            if (kd >= kml) {
                for (uint32_t i = lane; i < kml; i += LANES) ring[(uint32_t)(wm + i) & WMASK] = ring[(from + i) & WMASK];
            } else {                                     // the match overlaps its own output: its period, as often as it takes
                const float inv = 1.0f / (float)kd;
                for (uint32_t i = lane; i < kml; i += LANES) {
                    const uint32_t q = (uint32_t)((float)i * inv);
                    int32_t rem = (int32_t)(i - q * kd);
                    if (rem < 0) rem += (int32_t)kd;
                    if (rem >= (int32_t)kd) rem -= (int32_t)kd;
                    ring[(uint32_t)(wm + i) & WMASK] = ring[(from + (uint32_t)rem) & WMASK];
                }
            }
        }
    };
    uint4* const meta = reinterpret_cast<uint4*>(wlds + WRING + WSTAGE);      // [64] (header, output offset, literal offset) of the batch's sequences
    constexpr uint32_t SEQ_LANES = 8, STEP_SEQS = LANES / SEQ_LANES;     // lanes per sequence, sequences per step
    const uint32_t q8 = lane / SEQ_LANES, i8 = lane & (SEQ_LANES - 1u);
    uint32_t hnext = lane < nseq ? hdr[lane] : 0u;
    for (uint32_t s0 = 0; s0 < nseq && err == svx_lz::LZ_OK; s0 += LANES) {
        const uint32_t h = hnext;
        hnext = s0 + LANES + lane < nseq ? hdr[s0 + LANES + lane] : 0u;
        uint32_t T, Lt;
        {
            const uint32_t nl = h & 255u, ml = (h >> 8) & 511u;
            const uint32_t wofs = wave_excl_sum(nl + ml, &T), lofs = wave_excl_sum(nl, &Lt);
            meta[lane] = make_uint4(h, wofs, lofs, 0u);
        }
        if (W + T > hi || L + Lt > nlit) { err = svx_lz::LZ_OUT_OVERRUN; break; }
        const uint32_t nb = min((uint32_t)LANES, nseq - s0);
        // the literals of a batch are normally ~100 bytes: staged here, once; a batch of long runs stages sequence by sequence
        const bool bulk = Lt <= 1024u;
        if (bulk && Lt) need_chunk((L + Lt + 15u - r) >> 4);
        const uint32_t base32 = (uint32_t)W, rel0 = (uint32_t)(W - lo);
        // A STEP takes the next (up to) eight sequences, eight lanes each (four of sixteen, which take 97 % of the sequences instead
        // of 89 %, were measured: 10 % slower -- the steps are cut short by dependent matches, not by long ones): lane (q, i) copies literal byte i and match byte i of
        // sequence k0 + q.  Literals first, then the matches' reads, then their writes -- in order in the LDS --, so a step may
        // hold every sequence whose match reads nothing a match of the SAME step writes (conservatively: nothing at or behind
        // the step's first output byte; the step's first sequence is exempt: what it reads is older, or its own literals) and
        // that has at most eight literal and eight match bytes (89 % on HiFi-like data).  The first sequence that does not fit
        // ends the step; if it is the step's first, all lanes take it alone.
        uint32_t k0 = 0;
        uint4 mnext = meta[min(q8, nb - 1u)];
        while (k0 < nb) {
            const uint32_t k = k0 + q8;
            const bool valid = k < nb;
            const uint4 m = mnext;
            const uint32_t nl = m.x & 255u, ml = (m.x >> 8) & 511u, d = (m.x >> 17) + 1u;
            const int32_t step0 = __builtin_amdgcn_readfirstlane((int32_t)m.y);
            if ((uint32_t)(base32 + (uint32_t)step0 - (uint32_t)flushed) >= WFLUSH + 16u) flush_to((W + (uint32_t)step0) & ~15ull);
            uint32_t imod = i8;
            if (d < ml) imod = i8 - d * (uint32_t)((float)i8 * __builtin_amdgcn_rcpf((float)d) + 1e-3f);
            const int32_t srel = (int32_t)(m.y + nl - d + imod);
            const bool rd = valid && i8 < ml;
            // (a distance of nearly the whole window: the ring slot the match reads is the slot of a byte up to 64 positions AHEAD --
            // a literal of a later sequence of the step, written before the matches read, would be in it: such a match goes alone)
            const bool bad = valid && (nl > 8u || ml > 8u || !bulk || d > WRING - 512u || (rd && q8 > 0 && srel >= step0));
            const unsigned long long bm = __ballot(bad);
            const uint32_t n_ok = bm ? (uint32_t)(__ffsll((long long)bm) - 1) / SEQ_LANES : min(STEP_SEQS, nb - k0);
            // the next step's sequences are known now: their descriptors are on their way while this step's bytes move
            const uint32_t k1 = k0 + (n_ok ? n_ok : 1u);
            mnext = meta[min(k1 + q8, nb - 1u)];
            if (n_ok) {
                const bool act = valid && q8 < n_ok;
                if (__any(act && rd && d > rel0 + m.y + nl)) { err = svx_lz::LZ_BAD_DIST; break; }
                const uint32_t w32 = base32 + m.y;
                if (__any(act && i8 < nl)) {
                    if (act && i8 < nl) ring[(w32 + i8) & WMASK] = stage[((L + m.z + 16u - r + i8) ^ 15u) & (WSTAGE - 1)];
                }
                if (act && rd) ring[(w32 + nl + i8) & WMASK] = ring[(base32 + (uint32_t)srel) & WMASK];
                k0 += n_ok;
            } else {
                careful((uint32_t)__builtin_amdgcn_readfirstlane((int32_t)m.x), W + (uint32_t)step0, L + (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)m.z));
                if (err != svx_lz::LZ_OK) break;
                k0 += 1;
            }
        }
        W += T; L += Lt;
        if ((uint32_t)((uint32_t)W - (uint32_t)flushed) >= WFLUSH + 16u) flush_to(W & ~15ull);
    }
    if (err == svx_lz::LZ_OK && W != hi) err = svx_lz::LZ_SHORT;
    if (err == svx_lz::LZ_OK) flush_to(hi);
    else if (lane == 0) status[b] = (uint32_t)err;
}

}  // namespace

extern "C" size_t svx_bgzf_inflate_fast_ws_bytes(uint64_t inflated_bytes, uint32_t n_blocks)
{
    // [per block (sequences, literal bytes) -- or the stream's bytes --: 8 n, rounded up to 256][the blocks' slots: 1.5 x ISIZE + 1 KB each][64 bytes: svx_lz::decode_block reads a
    // header and the 12 bytes behind it with one 16-byte load, also at the end of the last slot]
    return (size_t)((((size_t)8 * n_blocks + 255) & ~(size_t)255) + inflated_bytes + (inflated_bytes >> 1) + 1024ull * n_blocks + 64);
}

// The contract of svx_bgzf_inflate with a workspace (svx_bgzf_inflate_fast_ws_bytes(d_dst_off[n] - d_dst_off[0], n) bytes, 16-byte
// aligned): the blocks' LZ sequence streams and their lengths live there between the two kernels.
extern "C" __attribute__((visibility("hidden"))) int svx_bgzf_inflate_wave_only(const uint8_t* d_comp, const uint64_t* d_src_off, const uint32_t* d_src_len, const uint64_t* d_dst_off,
                                          uint32_t n_blocks, uint8_t* d_out, uint32_t* d_status, uint32_t only, void* stream);

// (include/svx_experimental.h: the LZ kernel by name -- 0 = by the size of the launch, 1 = lane per block, 2 = wave per block)
extern "C" int svx_bgzf_inflate_fast_lz(const uint8_t* d_comp, const uint64_t* d_src_off, const uint32_t* d_src_len, const uint64_t* d_dst_off,
                                        uint32_t n_blocks, uint64_t inflated_bytes, uint8_t* d_out, uint32_t* d_status, void* d_ws, uint64_t ws_bytes,
                                        int lz_kernel, void* stream_tokens, void* stream_lz)
{
    if (n_blocks == 0) return SVX_OK;
    if (!d_comp || !d_src_off || !d_src_len || !d_dst_off || !d_out || !d_status || !d_ws) return SVX_EINVAL;
    if (reinterpret_cast<uintptr_t>(d_ws) & 15u) return SVX_EINVAL;
    // inflated_bytes = d_dst_off[n] - d_dst_off[0], which the caller summed on the host (the array itself is device memory):
    // the slots of the sequence streams are laid out by it, so a smaller workspace would be overrun by the tokens kernel
    if (ws_bytes < svx_bgzf_inflate_fast_ws_bytes(inflated_bytes, n_blocks)) return SVX_EINVAL;
    hipStream_t sa = static_cast<hipStream_t>(stream_tokens), sb = static_cast<hipStream_t>(stream_lz);
    uint2* stream_cnt = static_cast<uint2*>(d_ws);
    uint8_t* streams = static_cast<uint8_t*>(d_ws) + (((size_t)8 * n_blocks + 255) & ~(size_t)255);
    // which LZ kernel: one wave per block (B': ~1 ms per block, 1,024 blocks at once) below WAVE_LZ_BELOW blocks, one lane per
    // block (B: ~25 ms whatever the launch holds + 0.1 ms per 1,000 blocks) above.  lz_kernel 1 | 2: by name (measurements; an
    // argument, not the environment: launches of several threads are in flight at once -- ADVICE r5).
    static const uint32_t below = getenv("SVX_WAVE_LZ_BELOW") ? (uint32_t)atoi(getenv("SVX_WAVE_LZ_BELOW")) : (uint32_t)WAVE_LZ_BELOW;
    if (lz_kernel < 0 || lz_kernel > 2) return SVX_EINVAL;
    const bool wave_lz = lz_kernel ? lz_kernel == 2 : n_blocks < below;
    const char* only = getenv("SVX_INFLATE2_ONLY");                          // measurements: "A" = kernel A alone (the output stays unwritten),
    if (!only && getenv("SVX_INFLATE2_ONLY_A")) only = "A";                  // "B" = kernel B alone on the streams an earlier call left in the same workspace
    if (!only || only[0] != 'B') {
        if (sa != sb) {                                                      // A reads what the caller's stream (stream_lz) has prepared: the tables, d_status
            hipEvent_t ready;
            if (hipEventCreateWithFlags(&ready, hipEventDisableTiming) != hipSuccess) return SVX_ELAUNCH;
            (void)hipEventRecord(ready, sb);
            (void)hipStreamWaitEvent(sa, ready, 0);
            (void)hipEventDestroy(ready);
        }
        static const unsigned tok_lds = getenv("SVX_TOK_LDS") ? (unsigned)atoi(getenv("SVX_TOK_LDS")) : 0u;      // (experiment: extra LDS per wave = fewer waves per CU)
        if (wave_lz) hipLaunchKernelGGL(bgzf_tokens_kernel<true>, dim3(n_blocks), dim3(LANES), tok_lds, sa, d_comp, d_src_off, d_src_len, d_dst_off, n_blocks, streams, stream_cnt, d_status);
        else hipLaunchKernelGGL(bgzf_tokens_kernel<false>, dim3(n_blocks), dim3(LANES), tok_lds, sa, d_comp, d_src_off, d_src_len, d_dst_off, n_blocks, streams, stream_cnt, d_status);
        if (sa != sb) {
            hipEvent_t done;
            if (hipEventCreateWithFlags(&done, hipEventDisableTiming) != hipSuccess) return SVX_ELAUNCH;
            (void)hipEventRecord(done, sa);
            (void)hipStreamWaitEvent(sb, done, 0);
            (void)hipEventDestroy(done);
        }
    }
    if (only && only[0] == 'A') return SVX_OK;
    if (wave_lz) hipLaunchKernelGGL(bgzf_lz_wave_kernel, dim3(n_blocks), dim3(LANES), WRING + WSTAGE + 1024, sb, streams, stream_cnt, d_dst_off, n_blocks, d_out, d_status);
    else hipLaunchKernelGGL(bgzf_lz_kernel, dim3((n_blocks + LZ_LANES - 1) / LZ_LANES), dim3(LZ_LANES), LZ_LANES * RING_STRIDE, sb, streams, stream_cnt, d_dst_off, n_blocks, d_out, d_status);
    if (hipGetLastError() != hipSuccess) return SVX_ELAUNCH;
    // the (pathological) blocks whose sequence stream did not fit its slot: the wave-per-block kernel, those blocks only
    return svx_bgzf_inflate_wave_only(d_comp, d_src_off, d_src_len, d_dst_off, n_blocks, d_out, d_status, INF_TOKENS_OVERFLOW, stream_lz);
}

extern "C" int svx_bgzf_inflate_fast_on(const uint8_t* d_comp, const uint64_t* d_src_off, const uint32_t* d_src_len, const uint64_t* d_dst_off,
                                        uint32_t n_blocks, uint64_t inflated_bytes, uint8_t* d_out, uint32_t* d_status, void* d_ws, uint64_t ws_bytes,
                                        void* stream_tokens, void* stream_lz)
{
    return svx_bgzf_inflate_fast_lz(d_comp, d_src_off, d_src_len, d_dst_off, n_blocks, inflated_bytes, d_out, d_status, d_ws, ws_bytes, 0, stream_tokens, stream_lz);
}

extern "C" int svx_bgzf_inflate_fast(const uint8_t* d_comp, const uint64_t* d_src_off, const uint32_t* d_src_len, const uint64_t* d_dst_off,
                                     uint32_t n_blocks, uint64_t inflated_bytes, uint8_t* d_out, uint32_t* d_status, void* d_ws, uint64_t ws_bytes,
                                     void* stream)
{
    return svx_bgzf_inflate_fast_on(d_comp, d_src_off, d_src_len, d_dst_off, n_blocks, inflated_bytes, d_out, d_status, d_ws, ws_bytes, stream, stream);
}
