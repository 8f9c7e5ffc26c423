// svx_inflate.hip -- BGZF inflate on gfx950 (MI355X): one LANE per BGZF block.
//
// Ingestion is bound by DEFLATE decoding: a HiFi BAM inflates to ~22 KB per read (bases + qualities
// the hot path never looks at, interleaved with the CIGARs it needs), a host core decodes ~0.6 GB/s of it
// (libdeflate), and the GPU boxes this was built on give a container the CPU time of 16 cores: ~10 GB/s, four times
// less than the device pipeline consumes.  BGZF makes the problem embarrassingly parallel -- every block (<= 64 KB
// inflated) is an independent raw-DEFLATE stream (SAMv1 4.1) and a chromosome has 10^4..10^5 of them -- so the blocks
// are decoded on the device, one lane each, from the compressed bytes uploaded as they sit in the file:
//
//   * a lane owns its block from the first bit to the last byte: no cross-lane dependency, no barrier, no atomics;
//   * Huffman decoding is canonical (RFC 1951 3.2.2) and branch-free: the left-aligned code limits of the fifteen code
//     lengths of the block in hand live in REGISTERS, a symbol's length is one plus the number of limits its next 15
//     bits reach (fourteen compare / add-with-carry pairs: no divergence between the lanes of a wave, whose blocks are
//     at unrelated places of their streams), and the symbol comes from the lane's private slice of LDS (420 B per lane:
//     one wave per workgroup, six workgroups per CU);
//   * input: a 64-bit bit buffer per lane fed from 16-byte loads issued a whole vector ahead of their use (every lane
//     streams its own block: a load is a round trip to L2 or HBM, not an L1 hit); output: literals are gathered into
//     an aligned dword before they are stored, matches are copied byte by byte from the lane's own earlier output
//     (its stores are visible to its later loads).
//
// Integer exact by construction: the result is the byte stream zlib / libdeflate produce (tests/test_gpu_inflate.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "../../include/svx.h"

namespace {

constexpr int LANES = 64;                 // one wave per workgroup: the LDS slice of a lane is indexed by its lane id
constexpr int LIT_SYMS = 288, DIST_SYMS = 32, LANE_DWORDS = 9 + (LIT_SYMS + DIST_SYMS) / 4 + 16;     // 105 dwords = 420 B of LDS per lane
constexpr int LANE_DWORDS_PRIV = DIST_SYMS / 4 + 16;        // 24 dwords = 96 B per lane when the literal / length symbols sit in private memory

// (the tables of the RFC, kept for the compile-time check below)
constexpr uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
constexpr uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
constexpr uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
constexpr uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
// base value and extra bits of a length code (257..285 -> li = 0..28) and of a distance code (0..29), RFC 1951 3.2.5, as
// arithmetic: a table in constant memory costs a match four dependent vector loads through L2 (its index differs per lane, or
// sits in a VGPR), and a level-1 encoder emits a match every few symbols.
__device__ __forceinline__ uint32_t len_extra(int li) { return li < 8 || li == 28 ? 0u : (uint32_t)(li >> 2) - 1u; }
__device__ __forceinline__ uint32_t len_base(int li) { return li < 8 ? 3u + (uint32_t)li : li == 28 ? 258u : 3u + ((4u + ((uint32_t)li & 3u)) << ((uint32_t)(li >> 2) - 1u)); }
__device__ __forceinline__ uint32_t dist_extra(int ds) { return ds < 4 ? 0u : (uint32_t)(ds >> 1) - 1u; }
__device__ __forceinline__ uint32_t dist_base(int ds) { return ds < 4 ? 1u + (uint32_t)ds : 1u + ((2u + ((uint32_t)ds & 1u)) << ((uint32_t)(ds >> 1) - 1u)); }

__constant__ uint8_t CLEN_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// per-length code counts of one alphabet, lengths 1..15, 10 bits each (a count is at most 288): three per dword.
// Five named dwords, never an array: one dynamically indexed access anywhere in the kernel would move the whole thing
// to scratch memory, and the symbol loop reads all fifteen counts for every symbol (first version: 6 scratch loads and
// 6 scratch stores per decoded symbol, SQ_INSTS_VMEM_RD / _WR).
struct Counts {
    uint32_t w0, w1, w2, w3, w4;
    __device__ __forceinline__ void clear() { w0 = w1 = w2 = w3 = w4 = 0; }
    __device__ __forceinline__ uint32_t word(int q) const { return q == 0 ? w0 : q == 1 ? w1 : q == 2 ? w2 : q == 3 ? w3 : w4; }
    __device__ __forceinline__ uint32_t get(int len) const { const int k = len - 1, q = k / 3; return (word(q) >> (10 * (k - 3 * q))) & 1023u; }
    __device__ __forceinline__ void add(int len, uint32_t by = 1)
    {
        const int k = len - 1, q = k / 3;
        const uint32_t inc = by << (10 * (k - 3 * q));
        w0 += q == 0 ? inc : 0u; w1 += q == 1 ? inc : 0u; w2 += q == 2 ? inc : 0u; w3 += q == 3 ? inc : 0u; w4 += q == 4 ? inc : 0u;
    }
};

struct BitReader {
    const uint4* vecs;                    // the compressed buffer as aligned 16-byte vectors
    uint64_t vnext, vend;                 // next vector to fetch, one past the last one that may be fetched
    uint4 cur, nxt;                       // vector being consumed; the one behind it, fetched a whole vector ahead
    int curw;                             // next dword of `cur`
    uint64_t buf;                         // unread bits, LSB first
    int cnt;                              // valid bits in buf
    int64_t fed, limit;                   // bits handed out so far / bits the stream holds
    __device__ __forceinline__ uint4 fetch()
    {
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (vnext < vend) v = vecs[vnext];                     // (past the end zeros are shifted in; exhausted() notices)
        ++vnext;
        return v;
    }
    __device__ __forceinline__ void init(const uint8_t* base, uint64_t byte_off, uint64_t byte_len)
    {
        vecs = reinterpret_cast<const uint4*>(base);
        vnext = byte_off >> 4;
        vend = (byte_off + byte_len + 15) >> 4;
        cur = fetch();
        nxt = fetch();
        curw = (int)(byte_off & 15) >> 2;
        buf = 0; cnt = 0;
        refill();
        const int skip = (int)(byte_off & 3) * 8;             // bytes in front of the stream inside its first dword
        buf >>= skip;
        cnt -= skip;
        fed = 0;
        limit = (int64_t)byte_len * 8;
    }
    __device__ __forceinline__ void refill()
    {
        if (cnt <= 32) {
            const uint32_t v = curw == 0 ? cur.x : curw == 1 ? cur.y : curw == 2 ? cur.z : cur.w;
            if (++curw == 4) { cur = nxt; nxt = fetch(); curw = 0; }      // the new vector is needed four refills from now
            buf |= (uint64_t)v << cnt;
            cnt += 32;
        }
    }
    __device__ __forceinline__ void drop(int n) { buf >>= n; cnt -= n; fed += n; }
    __device__ __forceinline__ uint32_t bits(int n)           // n <= 16
    {
        refill();
        const uint32_t v = (uint32_t)buf & ((1u << n) - 1u);
        drop(n);
        return v;
    }
    __device__ __forceinline__ bool exhausted() const { return fed > limit; }   // more bits consumed than the stream holds
};

// A lane's symbol tables in LDS, sorted by (code length, symbol): the literal / length alphabet as the low bytes of its
// symbols plus one bit per entry for the ninth (symbols 256..287), the distance alphabet as bytes -- 356 bytes per lane
// instead of 640 with 16-bit entries: seven one-wave workgroups per CU instead of four.
struct LitSyms {
    uint8_t* lo;                          // [288]
    uint32_t* hi;                         // [9] bit i = entry i is >= 256
    __device__ __forceinline__ int get(int i) const { return (int)lo[i] | (int)((hi[i >> 5] >> (i & 31)) & 1u) << 8; }
    __device__ __forceinline__ void clear() { for (int k = 0; k < 9; ++k) hi[k] = 0; }
    __device__ __forceinline__ void put(int i, int sym) { lo[i] = (uint8_t)sym; if (sym & 256) hi[i >> 5] |= 1u << (i & 31); }
};
// The same alphabet in the lane's PRIVATE memory (16-bit entries): the LDS slice shrinks to the distance symbols and the
// bases -- 96 bytes per lane, twelve one-wave workgroups per CU instead of six.  A wave is slower (a private load is a trip
// to L1 / L2 where the LDS read took ~100 cycles: 68 ms per block instead of 64), but 196,608 blocks are in flight at
// once instead of 98,304: the variant for launches that would need a second round of the LDS version.
struct PrivSyms {
    uint16_t* v;                          // [288]
    __device__ __forceinline__ int get(int i) const { return (int)v[i]; }
    __device__ __forceinline__ void clear() {}
    __device__ __forceinline__ void put(int i, int sym) { v[i] = (uint16_t)sym; }
};
struct ByteSyms {
    uint8_t* lo;
    __device__ __forceinline__ int get(int i) const { return (int)lo[i]; }
    __device__ __forceinline__ void clear() {}
    __device__ __forceinline__ void put(int i, int sym) { lo[i] = (uint8_t)sym; }
};

// Decoder state of one alphabet: for every code length L the left-aligned (15-bit) code one past the last code of that
// length -- the canonical codes of length L are exactly the 15-bit prefixes in [lim[L-2], lim[L-1]) -- in fifteen
// registers (static indices only), and in the lane's LDS slice base[L] = (table slot of the first symbol of length L) -
// (first code of length L), so that a symbol is  syms[(peek >> (15 - L)) + base[L]].
struct Dec {
    uint32_t lim[15];
    int16_t* base;                        // LDS, [16]
};

// One symbol, branch-free: the next 15 bits MSB-first (`peek`), its code length = 1 + the number of limits it reaches
// (fourteen compare + add-with-carry pairs instead of the divergent walk over the lengths: the wave paid every symbol
// for its slowest lane's code), then two LDS reads.  -1: not a code of this alphabet.
template <class Syms>
__device__ __forceinline__ int decode_symbol(BitReader& br, const Dec& d, const Syms& syms)
{
    br.refill();
    const uint32_t peek = __brev((uint32_t)br.buf) >> 17;
    int len = 1;
#pragma unroll
    for (int k = 0; k < 14; ++k) len += peek >= d.lim[k] ? 1 : 0;
    if (peek >= d.lim[14]) return -1;
    br.drop(len);
    return syms.get((int)(peek >> (15 - len)) + (int)d.base[len]);
}

// code lengths -> the limits (registers), base[] and the symbols sorted by (length, symbol) in the lane's LDS slice
// (RFC 1951 3.2.2)
template <class Syms>
__device__ __forceinline__ bool build(const uint8_t* lens, int n, Dec& d, Syms& syms)
{
    Counts c;
    c.clear();
    syms.clear();
    // The code lengths sit in the lane's scratch memory (a round trip to L2 per access, and only this lane of the wave is
    // here: the others wait).  Sixteen lengths per load, the loads of a pass issued ahead of their use: a byte-wise walk
    // (two dependent loads per symbol, 2 x 316 per block header) was half of the kernel's time (SQ_INSTS_VMEM_RD: 7 per
    // decoded symbol).
    const uint4* lens16 = reinterpret_cast<const uint4*>(lens);
    const int chunks = (n + 15) >> 4;
#pragma unroll 1
    for (int q = 0; q < chunks; ++q) {
        const uint4 v = lens16[q];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int len = (int)((w[k >> 2] >> (8 * (k & 3))) & 255u);
            if (len && 16 * q + k < n) c.add(len);
        }
    }
    int left = 1;                                               // over-subscription check
    Counts offs;                                                // first table slot of every length (same packing)
    offs.clear();
    uint32_t run = 0, code = 0;
#pragma unroll
    for (int len = 1; len <= 15; ++len) {
        const uint32_t count = c.get(len);
        left <<= 1;
        left -= (int)count;
        offs.add(len, run);
        d.base[len] = (int16_t)((int)run - (int)code);
        run += count;
        code += count;
        d.lim[len - 1] = code << (15 - len);                    // (an over-subscribed code overflows 15 bits: rejected below)
        code <<= 1;
    }
    if (left < 0) return false;
#pragma unroll 1
    for (int q = 0; q < chunks; ++q) {
        const uint4 v = lens16[q];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int len = (int)((w[k >> 2] >> (8 * (k & 3))) & 255u);
            if (len && 16 * q + k < n) { syms.put((int)offs.get(len), 16 * q + k); offs.add(len); }
        }
    }
    return true;                                                // (incomplete codes are legal for a single distance code)
}

struct Writer {
    uint8_t* base;                        // the whole inflated buffer
    uint64_t lo, pos, hi;                 // the block's byte range [lo, hi) and the next byte to write
    uint32_t acc;                         // pending bytes of the aligned dword that ends at pos (nacc of them)
    int nacc;
    __device__ __forceinline__ void flush()
    {
        for (int i = 0; i < nacc; ++i) base[pos - nacc + i] = (uint8_t)(acc >> (8 * i));
        nacc = 0; acc = 0;
    }
    __device__ __forceinline__ void literal(uint32_t b)
    {
        if (nacc == 0 && (pos & 3)) { base[pos++] = (uint8_t)b; return; }     // not on a dword boundary yet
        acc |= b << (8 * nacc);
        ++nacc; ++pos;
        if (nacc == 4) { *reinterpret_cast<uint32_t*>(base + pos - 4) = acc; nacc = 0; acc = 0; }
    }
};

// A match is copied in steps of up to eight bytes, ONE step per turn of the symbol loop: the step's source is fetched with
// one unaligned 8-byte load at the top of the turn, the turn's Huffman decoding runs while it is on its way, and the bytes
// leave with one unaligned 8-byte store.  (First versions copied a whole match byte by byte -- load, wait, store -- inside
// the turn that decoded it: the 63 other lanes of the wave, whose blocks are at unrelated places of their streams, waited
// for every byte's round trip; ~3,800 of the 4,800 cycles of an average turn.)
//   * distance < 8: the eight bytes are the period of `dist` bytes repeated; afterwards the distance is replaced by its
//     smallest multiple >= 8 (the output is periodic from pos - dist on, so the source may be any whole period back);
//   * the eight-byte store may carry up to seven bytes of garbage behind the step's last byte: they land in this lane's own
//     not yet written output (never behind the block's end: byte-wise there) and later stores of the same lane replace them.
struct MatchCopy {
    uint32_t len, dist;                   // bytes still to copy (0: none pending), their distance
    uint64_t v;                           // the step's source bytes (fetch -> commit)
    bool wide;                            // this step runs on 8-byte accesses
    __device__ __forceinline__ void fetch(const Writer& w)
    {
        wide = len != 0 && w.pos + 8 <= w.hi;              // (then the source's eight bytes end before the block does as well)
        if (wide) __builtin_memcpy(&v, w.base + w.pos - dist, 8);
    }
    __device__ __forceinline__ void commit(Writer& w)
    {
        if (len == 0) return;
        const uint32_t n = len < 8u ? len : 8u;
        if (wide) {
            if (dist < 8u) {
                const uint32_t sh = 8u * dist;
                uint64_t p = v & ((1ull << sh) - 1ull);
                p |= p << sh;
                if (2u * sh < 64u) p |= p << (2u * sh);
                if (4u * sh < 64u) p |= p << (4u * sh);
                v = p;
                dist *= (7u + dist) / dist;                    // smallest multiple of the period that is >= 8
            }
            __builtin_memcpy(w.base + w.pos, &v, 8);
        } else {
            for (uint32_t i = 0; i < n; ++i) w.base[w.pos + i] = w.base[w.pos + i - dist];
        }
        w.pos += n;
        len -= n;
    }
};

template <bool PRIVATE_SYMS> struct LitStore;
template <> struct LitStore<false> { static __device__ __forceinline__ LitSyms make(uint32_t* mine, uint16_t*) { return LitSyms{reinterpret_cast<uint8_t*>(mine + 9), mine}; } };
template <> struct LitStore<true> { static __device__ __forceinline__ PrivSyms make(uint32_t*, uint16_t* priv) { return PrivSyms{priv}; } };
template <bool PRIVATE_SYMS>
__device__ __forceinline__ auto make_lit_syms(uint32_t* mine, uint16_t* priv) { return LitStore<PRIVATE_SYMS>::make(mine, priv); }

// status per block: 0 ok, else the reason
enum { INF_OK = 0, INF_BAD_TYPE = 1, INF_BAD_STORED = 2, INF_BAD_TABLE = 3, INF_BAD_CODE = 4, INF_OUT_OVERRUN = 5, INF_IN_OVERRUN = 6, INF_SHORT = 7, INF_BAD_DIST = 8 };
enum { ST_HEADER = 0, ST_SYMBOLS = 1, ST_DONE = 2 };

template <bool PRIVATE_SYMS>
__global__ __launch_bounds__(LANES)
void bgzf_inflate_kernel(const uint8_t* __restrict__ comp, const uint64_t* __restrict__ src_off, const uint32_t* __restrict__ src_len,
                         const uint64_t* __restrict__ dst_off, uint32_t n_blocks, uint8_t* out, uint32_t* __restrict__ status)
{
    constexpr int DW = PRIVATE_SYMS ? LANE_DWORDS_PRIV : LANE_DWORDS;
    __shared__ uint32_t lds[LANES * DW];
    const uint32_t b = blockIdx.x * LANES + threadIdx.x;
    if (b >= n_blocks) return;
    uint32_t* mine = lds + threadIdx.x * DW;                  // [9 mask dwords][288 literal bytes][32 distance bytes][2 x 16 bases], or the last two only
    uint16_t lit_private[PRIVATE_SYMS ? LIT_SYMS : 1];
    typename std::conditional<PRIVATE_SYMS, PrivSyms, LitSyms>::type lit_syms = make_lit_syms<PRIVATE_SYMS>(mine, lit_private);
    ByteSyms dist_syms{PRIVATE_SYMS ? reinterpret_cast<uint8_t*>(mine) : reinterpret_cast<uint8_t*>(mine + 9) + LIT_SYMS};
    Dec lc, dc, cc;
    lc.base = reinterpret_cast<int16_t*>(PRIVATE_SYMS ? mine + DIST_SYMS / 4 : mine + 9 + (LIT_SYMS + DIST_SYMS) / 4);
    dc.base = lc.base + 16;
    cc.base = dc.base;                                          // (the code-length code borrows the distance slices)
    BitReader br;
    br.init(comp, src_off[b], src_len[b]);
    Writer w{out, dst_off[b], dst_off[b], dst_off[b + 1], 0u, 0};
    MatchCopy mc{0u, 0u, 0ull, false};
    int err = INF_OK;
    __attribute__((aligned(16))) uint8_t lens[LIT_SYMS + DIST_SYMS];
    bool last = false;
    int state = w.hi == w.lo ? ST_DONE : ST_HEADER;             // an empty block (the EOF marker): nothing to decode
    // One loop for the whole block, every lane in its own state: a lane that reaches the end of a DEFLATE block parses the
    // next header while the others go on decoding (an inner symbol loop per DEFLATE block made every lane wait, at every
    // block boundary, for the slowest of the wave).
    while (state != ST_DONE) {
        if (state == ST_HEADER) {
            last = br.bits(1) != 0;
            const uint32_t type = br.bits(2);
            state = ST_SYMBOLS;
            if (type == 0) {                                    // stored: to the byte boundary, LEN, ~LEN, bytes
                br.bits(br.cnt & 7);
                const uint32_t len = br.bits(16), nlen = br.bits(16);
                if ((len ^ nlen) != 0xffffu) err = INF_BAD_STORED;
                else if (w.pos + len > w.hi) err = INF_OUT_OVERRUN;
                else for (uint32_t i = 0; i < len; ++i) w.literal(br.bits(8));
                if (br.exhausted()) err = INF_IN_OVERRUN;
                state = last ? ST_DONE : ST_HEADER;
            } else if (type == 1) {                             // fixed code (RFC 1951 3.2.6)
                for (int s = 0; s < 144; ++s) lens[s] = 8;
                for (int s = 144; s < 256; ++s) lens[s] = 9;
                for (int s = 256; s < 280; ++s) lens[s] = 7;
                for (int s = 280; s < 288; ++s) lens[s] = 8;
                for (int s = 0; s < 30; ++s) lens[LIT_SYMS + s] = 5;
                build(lens, 288, lc, lit_syms);
                build(lens + LIT_SYMS, 30, dc, dist_syms);
            } else if (type == 2) {                             // dynamic code (3.2.7)
                const int nlen = (int)br.bits(5) + 257, ndist = (int)br.bits(5) + 1, ncode = (int)br.bits(4) + 4;
                if (nlen > 286 || ndist > 30) err = INF_BAD_TABLE;
                else {
                    __attribute__((aligned(16))) uint8_t cl[32];
                    for (int i = 0; i < 32; ++i) cl[i] = 0;
                    for (int i = 0; i < ncode; ++i) cl[CLEN_ORDER[i]] = (uint8_t)br.bits(3);
                    if (!build(cl, 19, cc, dist_syms)) err = INF_BAD_TABLE;     // (the code-length code borrows the distance slice)
                    int i = 0;
                    while (err == INF_OK && i < nlen + ndist) {
                        const int sym = decode_symbol(br, cc, dist_syms);
                        if (sym < 0) { err = INF_BAD_TABLE; break; }
                        if (sym < 16) { lens[i < nlen ? i : LIT_SYMS + (i - nlen)] = (uint8_t)sym; ++i; continue; }
                        int prev = 0, rep;
                        if (sym == 16) {
                            if (i == 0) { err = INF_BAD_TABLE; break; }
                            const int j = i - 1;
                            prev = lens[j < nlen ? j : LIT_SYMS + (j - nlen)];
                            rep = 3 + (int)br.bits(2);
                        } else if (sym == 17) rep = 3 + (int)br.bits(3);
                        else rep = 11 + (int)br.bits(7);
                        if (i + rep > nlen + ndist) { err = INF_BAD_TABLE; break; }
                        for (; rep > 0; --rep, ++i) lens[i < nlen ? i : LIT_SYMS + (i - nlen)] = (uint8_t)prev;
                    }
                    if (err == INF_OK && lens[256] == 0) err = INF_BAD_TABLE;
                    if (err == INF_OK && (!build(lens, nlen, lc, lit_syms) || !build(lens + LIT_SYMS, ndist, dc, dist_syms))) err = INF_BAD_TABLE;
                }
            } else {
                err = INF_BAD_TYPE;
            }
            if (err != INF_OK) state = ST_DONE;
            continue;
        }
        // one turn of the symbol loop: a step of the pending match (if any) and -- unless more of it stays -- one symbol
        mc.fetch(w);
        int sym = -2;
        if (mc.len <= 8u) sym = decode_symbol(br, lc, lit_syms);
        mc.commit(w);
        if (sym == -2) continue;
        if (sym < 256) {
            if (sym < 0) { err = INF_BAD_CODE; state = ST_DONE; }
            else if (w.pos >= w.hi) { err = INF_OUT_OVERRUN; state = ST_DONE; }
            else w.literal((uint32_t)sym);
        } else if (sym == 256) {
            if (br.exhausted()) { err = INF_IN_OVERRUN; state = ST_DONE; }
            else state = last ? ST_DONE : ST_HEADER;
        } else {
            const int li = sym - 257;
            if (li >= 29) { err = INF_BAD_CODE; state = ST_DONE; continue; }
            const uint32_t len = len_base(li) + br.bits((int)len_extra(li));
            const int ds = decode_symbol(br, dc, dist_syms);
            if (ds < 0 || ds >= 30) { err = INF_BAD_CODE; state = ST_DONE; continue; }
            const uint32_t dist = dist_base(ds) + br.bits((int)dist_extra(ds));
            if (dist > w.pos - w.lo) { err = INF_BAD_DIST; state = ST_DONE; continue; }
            if (w.pos + len > w.hi) { err = INF_OUT_OVERRUN; state = ST_DONE; continue; }
            w.flush();                                          // the literals in hand are part of what the match may copy
            mc.len = len;
            mc.dist = dist;
        }
    }
    w.flush();
    if (err == INF_OK && w.pos != w.hi) err = INF_SHORT;
    status[b] = (uint32_t)err;
}


// ------------------------------------------------------------------------------------------------------------------
// One WAVE per BGZF block, uniform control flow.
//
// The lane-per-block kernel above pays, in almost every iteration, for whichever of its 64 unrelated streams happens to
// refill its bit buffer, store or copy a match (PMC: 7 vector-memory loads per decoded symbol, SQ_WAIT_ANY 70 %).  Here a
// wave owns ONE block: the bit buffer, the table walk and every branch are wave-uniform (the compiler keeps them on the
// scalar unit and in scalar branches), and the 64 lanes are the wide parts of the job:
//   * input: the stream's next 256 bytes sit in one VGPR (lane i holds dword i), fetched with one coalesced load, the
//     following 256 bytes already on their way; the bit buffer takes its dwords with v_readlane;
//   * Huffman decoding: 10-bit (literal / length) and 8-bit (distance) first-level tables in LDS, entry = symbol << 4 |
//     code length; longer codes (rare) walk the canonical limits;
//   * output: a 64-byte line is gathered in one VGPR (lane = byte) and leaves as one coalesced store; a match copies up to
//     min(length, distance, bytes left in the line) bytes per step, every lane fetching its source byte from the line
//     in hand (ds_bpermute) or from the bytes this wave stored earlier.
constexpr int W_LIT_BITS = 10, W_DIST_BITS = 8;
struct WaveLds {
    uint16_t lit_tab[1 << W_LIT_BITS];      // (symbol << 4) | length; 0 = a longer code
    uint16_t dist_tab[1 << W_DIST_BITS];
    uint16_t lit_syms[LIT_SYMS];            // symbols sorted by (length, symbol): the canonical fallback
    uint16_t dist_syms[DIST_SYMS];
    uint32_t lit_lim[16], dist_lim[16];     // [L - 1] = left-aligned 15-bit limit of length L
    int32_t lit_base[16], dist_base[16];    // [L] = first slot of length L - first code of length L
    uint8_t lens[LIT_SYMS + DIST_SYMS];
    uint8_t cl[32];
};

struct WaveReader {
    const uint32_t* words;
    uint32_t cur, nxt;                      // VGPRs: lane i = dword i of the chunk in hand / of the next one
    uint64_t chunk_next, word_end;          // (uniform) index of the first dword of the chunk behind `nxt`; one past the last dword
    int k;                                  // (uniform) next dword of `cur`
    uint64_t buf;                           // (uniform) unread bits, LSB first
    int cnt;
    int64_t fed, limit;
    __device__ __forceinline__ uint32_t fetch(uint64_t first)
    {
        const uint64_t i = first + threadIdx.x;
        return i < word_end ? words[i] : 0u;
    }
    __device__ __forceinline__ void init(const uint8_t* base, uint64_t byte_off, uint64_t byte_len)
    {
        words = reinterpret_cast<const uint32_t*>(base);
        const uint64_t w0 = byte_off >> 2;
        word_end = (byte_off + byte_len + 3) >> 2;
        cur = fetch(w0);
        nxt = fetch(w0 + 64);
        chunk_next = w0 + 128;
        k = 0; buf = 0; cnt = 0;
        refill();
        const int skip = (int)(byte_off & 3) * 8;
        buf >>= skip;
        cnt -= skip;
        fed = 0;
        limit = (int64_t)byte_len * 8;
    }
    __device__ __forceinline__ void refill()
    {
        if (cnt <= 32) {
            const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)cur, k);
            if (++k == 64) { cur = nxt; nxt = fetch(chunk_next); chunk_next += 64; k = 0; }
            buf |= (uint64_t)v << cnt;
            cnt += 32;
        }
    }
    __device__ __forceinline__ void drop(int n) { buf >>= n; cnt -= n; fed += n; }
    __device__ __forceinline__ uint32_t bits(int n)
    {
        refill();
        const uint32_t v = (uint32_t)buf & ((1u << n) - 1u);
        drop(n);
        return v;
    }
    __device__ __forceinline__ bool exhausted() const { return fed > limit; }
};

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// canonical tables of one alphabet from its code lengths (all uniform; a wave builds ~4 of these per block)
__device__ __forceinline__ bool wave_build(const uint8_t* lens, int n, uint16_t* tab, int tab_bits, uint16_t* syms, uint32_t* lim, int32_t* base)
{
    int count[16];
#pragma unroll
    for (int L = 0; L < 16; ++L) count[L] = 0;
    for (int s = 0; s < n; ++s) {
        const int L = uni((int)lens[s]);
#pragma unroll
        for (int q = 1; q < 16; ++q) count[q] += L == q ? 1 : 0;
    }
    int left = 1, run = 0, code = 0;
    int first[16], slot[16];
#pragma unroll
    for (int L = 1; L <= 15; ++L) {
        left <<= 1;
        left -= count[L];
        first[L] = code;
        slot[L] = run;
        base[L] = run - code;
        run += count[L];
        code += count[L];
        lim[L - 1] = (uint32_t)code << (15 - L);
        code <<= 1;
    }
    if (left < 0) return false;
    const int tab_n = 1 << tab_bits;
    for (int i = threadIdx.x; i < tab_n; i += LANES) tab[i] = 0;
    __builtin_amdgcn_s_waitcnt(0xc07f);                         // lgkmcnt(0): the clears land before the entries (same wave, in order)
    for (int s = 0; s < n; ++s) {
        const int L = uni((int)lens[s]);
        if (L == 0) continue;
        int c = 0, at = 0;
#pragma unroll
        for (int q = 1; q < 16; ++q) if (L == q) { c = first[q]++; at = slot[q]++; }
        syms[at] = (uint16_t)s;
        if (L <= tab_bits) {                                   // every table index whose low L bits are the code, LSB first
            const uint32_t r = __brev((uint32_t)c) >> (32 - L);
            const uint16_t e = (uint16_t)((s << 4) | L);
            for (uint32_t i = r + ((uint32_t)threadIdx.x << L); i < (uint32_t)tab_n; i += (uint32_t)LANES << L) tab[i] = e;
        }
    }
    return true;
}

// the canonical walk for a code longer than the first-level table (rare): out of line, so that its LDS reads are not hoisted
// into every iteration of the symbol loop (they were: five LDS reads per symbol instead of one)
__device__ __attribute__((noinline)) int wave_long_symbol(uint32_t peek, int tab_bits, const uint16_t* syms, const uint32_t* lim, const int32_t* base)
{
    for (int L = tab_bits + 1; L <= 15; ++L)
        if (peek < lim[L - 1]) return (L << 16) | (int)syms[(int)(peek >> (15 - L)) + base[L]];
    return -1;
}

// one symbol: first-level table, else the canonical walk (uniform)
__device__ __forceinline__ int wave_symbol(WaveReader& br, const uint16_t* tab, int tab_bits, const uint16_t* syms, const uint32_t* lim,
                                           const int32_t* base)
{
    br.refill();
    const int e = uni((int)tab[(uint32_t)br.buf & ((1u << tab_bits) - 1u)]);
    if (__builtin_expect(e != 0, 1)) { br.drop(e & 15); return e >> 4; }
    const int r = uni(wave_long_symbol(__brev((uint32_t)br.buf) >> 17, tab_bits, syms, lim, base));
    if (r < 0) return -1;
    br.drop(r >> 16);
    return r & 0xffff;
}

struct WaveWriter {
    uint8_t* base;
    uint64_t lo, pos, hi;                   // (uniform) the block's byte range and the next byte
    uint32_t line;                          // VGPR: lane j = byte (pos & ~63) + j of the line in hand
    __device__ __forceinline__ void store_line(uint64_t upto)   // bytes [max(lo, line start), upto) of the line leave
    {
        const uint64_t l0 = (upto - 1) & ~63ull;
        const uint64_t a = l0 + threadIdx.x;
        if (a >= lo && a < upto) base[a] = (uint8_t)line;
    }
    __device__ __forceinline__ void literal(uint32_t b)
    {
        if ((pos & 63) == threadIdx.x) line = b;
        ++pos;
        if ((pos & 63) == 0) store_line(pos);
    }
    __device__ __forceinline__ void copy(uint32_t dist, uint32_t len)
    {
        while (len) {
            const uint32_t j0 = (uint32_t)(pos & 63);
            uint32_t n = 64 - j0;
            if (n > len) n = len;
            if (n > dist) n = dist;                             // the sources of a step lie in front of it
            const uint64_t l0 = pos & ~63ull;
            const int j = (int)threadIdx.x;
            const bool mine = (uint32_t)j >= j0 && (uint32_t)j < j0 + n;
            const int64_t a = (int64_t)l0 + j - (int64_t)dist;  // source byte of lane j
            const int from_line = __builtin_amdgcn_ds_bpermute(((j - (int)dist) & 63) << 2, (int)line);
            uint32_t v = (uint32_t)from_line & 255u;
            if (mine && a < (int64_t)l0) v = base[a];           // stored by this wave earlier (program order, same L1)
            if (mine) line = v;
            pos += n;
            len -= n;
            if ((pos & 63) == 0) store_line(pos);
        }
    }
    __device__ __forceinline__ void finish() { if (pos & 63) store_line(pos); }
};

__global__ __launch_bounds__(LANES)
void bgzf_inflate_wave_kernel(const uint8_t* __restrict__ comp, const uint64_t* __restrict__ src_off, const uint32_t* __restrict__ src_len,
                              const uint64_t* __restrict__ dst_off, uint32_t n_blocks, uint8_t* out, uint32_t* __restrict__ status, uint32_t only)
{
    __shared__ WaveLds t;
    const uint32_t b = blockIdx.x;
    if (only != 0u && uni((int)status[b]) != (int)only) return;     // (svx_bgzf_inflate_fast: the blocks it left over, those only)
    WaveReader br;
    br.init(comp, src_off[b], src_len[b]);
    WaveWriter w{out, dst_off[b], dst_off[b], dst_off[b + 1], 0u};
    int err = INF_OK;
    bool last = w.hi == w.lo;
    while (!last && err == INF_OK) {
        last = br.bits(1) != 0;
        const uint32_t type = br.bits(2);
        if (type == 0) {
            br.bits(br.cnt & 7);
            const uint32_t len = br.bits(16), nlen = br.bits(16);
            if ((len ^ nlen) != 0xffffu) { err = INF_BAD_STORED; break; }
            if (w.pos + len > w.hi) { err = INF_OUT_OVERRUN; break; }
            for (uint32_t i = 0; i < len; ++i) w.literal(br.bits(8));
            if (br.exhausted()) { err = INF_IN_OVERRUN; break; }   // a truncated payload is padded with zero bits: not a stored block
        } else if (type == 1 || type == 2) {
            if (type == 1) {
                for (int s = threadIdx.x; s < 288; s += LANES) t.lens[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
                if (threadIdx.x < 30) t.lens[LIT_SYMS + threadIdx.x] = 5;
                __builtin_amdgcn_s_waitcnt(0xc07f);
                wave_build(t.lens, 288, t.lit_tab, W_LIT_BITS, t.lit_syms, t.lit_lim, t.lit_base);
                wave_build(t.lens + LIT_SYMS, 30, t.dist_tab, W_DIST_BITS, t.dist_syms, t.dist_lim, t.dist_base);
            } else {
                const int nlen = (int)br.bits(5) + 257, ndist = (int)br.bits(5) + 1, ncode = (int)br.bits(4) + 4;
                if (nlen > 286 || ndist > 30) { err = INF_BAD_TABLE; break; }
                if (threadIdx.x < 32) t.cl[threadIdx.x] = 0;
                __builtin_amdgcn_s_waitcnt(0xc07f);
                for (int i = 0; i < ncode; ++i) { const uint32_t v = br.bits(3); if (threadIdx.x == 0) t.cl[CLEN_ORDER[i]] = (uint8_t)v; }
                __builtin_amdgcn_s_waitcnt(0xc07f);
                // the code-length code borrows the distance tables (7-bit codes: an 8-bit table holds them all)
                if (!wave_build(t.cl, 19, t.dist_tab, W_DIST_BITS, t.dist_syms, t.dist_lim, t.dist_base)) { err = INF_BAD_TABLE; break; }
                __builtin_amdgcn_s_waitcnt(0xc07f);
                int i = 0;
                while (i < nlen + ndist) {
                    const int sym = wave_symbol(br, t.dist_tab, W_DIST_BITS, t.dist_syms, t.dist_lim, t.dist_base);
                    if (sym < 0) { err = INF_BAD_TABLE; break; }
                    int value = sym, rep = 1;
                    if (sym == 16) {
                        if (i == 0) { err = INF_BAD_TABLE; break; }
                        const int jx = i - 1;
                        value = uni((int)t.lens[jx < nlen ? jx : LIT_SYMS + (jx - nlen)]);
                        rep = 3 + (int)br.bits(2);
                    } else if (sym == 17) { value = 0; rep = 3 + (int)br.bits(3); }
                    else if (sym == 18) { value = 0; rep = 11 + (int)br.bits(7); }
                    if (i + rep > nlen + ndist) { err = INF_BAD_TABLE; break; }
                    if ((int)threadIdx.x < rep) {               // up to 138 repeats: lanes 0..63 twice, then once more
                        for (int r = threadIdx.x; r < rep; r += LANES) { const int x = i + r; t.lens[x < nlen ? x : LIT_SYMS + (x - nlen)] = (uint8_t)value; }
                    }
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                    i += rep;
                }
                if (err != INF_OK) break;
                if (uni((int)t.lens[256]) == 0) { err = INF_BAD_TABLE; break; }
                if (!wave_build(t.lens, nlen, t.lit_tab, W_LIT_BITS, t.lit_syms, t.lit_lim, t.lit_base) ||
                    !wave_build(t.lens + LIT_SYMS, ndist, t.dist_tab, W_DIST_BITS, t.dist_syms, t.dist_lim, t.dist_base)) { err = INF_BAD_TABLE; break; }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            for (;;) {
                const int sym = wave_symbol(br, t.lit_tab, W_LIT_BITS, t.lit_syms, t.lit_lim, t.lit_base);
                if (sym < 256) {
                    if (sym < 0) { err = INF_BAD_CODE; break; }
                    if (w.pos >= w.hi) { err = INF_OUT_OVERRUN; break; }
                    w.literal((uint32_t)sym);
                    continue;
                }
                if (sym == 256) break;
                const int li = sym - 257;
                if (li >= 29) { err = INF_BAD_CODE; break; }
                const uint32_t len = len_base(li) + br.bits((int)len_extra(li));
                const int ds = wave_symbol(br, t.dist_tab, W_DIST_BITS, t.dist_syms, t.dist_lim, t.dist_base);
                if (ds < 0 || ds >= 30) { err = INF_BAD_CODE; break; }
                const uint32_t dist = dist_base(ds) + br.bits((int)dist_extra(ds));
                if (dist > w.pos - w.lo) { err = INF_BAD_DIST; break; }
                if (w.pos + len > w.hi) { err = INF_OUT_OVERRUN; break; }
                w.copy(dist, len);
            }
            if (br.exhausted()) err = INF_IN_OVERRUN;
        } else {
            err = INF_BAD_TYPE;
        }
    }
    w.finish();
    if (err == INF_OK && w.pos != w.hi) err = INF_SHORT;
    if (threadIdx.x == 0) status[b] = (uint32_t)err;
}

}  // namespace

// BGZF blocks -> their inflated bytes, all blocks of a launch in parallel (one lane per block).
//   d_comp      compressed bytes as they sit in the file (any run of whole blocks), 16-byte aligned, readable up to the
//               next multiple of 16 behind the last payload byte
//   d_src_off   [n] byte offset in d_comp of every block's DEFLATE payload (behind the 18-byte header)
//   d_src_len   [n] payload bytes (BSIZE - xlen - 19)
//   d_dst_off   [n + 1] byte offset in d_out of every block's inflated bytes: the running sum of the ISIZE fields
//   d_status    [n] 0 = block decoded to exactly its ISIZE bytes; anything else: corrupt block (the host falls back)
namespace {
constexpr uint32_t ONE_ROUND_BLOCKS = 6u * 256u * LANES;       // the blocks the LDS version holds at once: six one-wave workgroups on each of 256 CUs
template <bool PRIVATE_SYMS>
int launch_lanes(const uint8_t* d_comp, const uint64_t* d_src_off, const uint32_t* d_src_len, const uint64_t* d_dst_off,
                 uint32_t n_blocks, uint8_t* d_out, uint32_t* d_status, void* stream)
{
    if (n_blocks == 0) return SVX_OK;
    if (!d_comp || !d_src_off || !d_src_len || !d_dst_off || !d_out || !d_status) return SVX_EINVAL;
    if (reinterpret_cast<uintptr_t>(d_comp) & 15u) return SVX_EINVAL;
    hipLaunchKernelGGL(bgzf_inflate_kernel<PRIVATE_SYMS>, dim3((n_blocks + LANES - 1) / LANES), dim3(LANES), 0, static_cast<hipStream_t>(stream),
                       d_comp, d_src_off, d_src_len, d_dst_off, n_blocks, d_out, d_status);
    return hipGetLastError() == hipSuccess ? SVX_OK : SVX_ELAUNCH;
}
}  // namespace

// A launch of more blocks than the LDS version holds at once would take it two rounds (98 k blocks: 74 ms, 113 k: 135 ms);
// the private-memory version holds twice as many (113 k: 94 ms, 170 k: 125 ms against 146) and takes launches of that size.
extern "C" int svx_bgzf_inflate(const uint8_t* d_comp, const uint64_t* d_src_off, const uint32_t* d_src_len, const uint64_t* d_dst_off,
                                uint32_t n_blocks, uint8_t* d_out, uint32_t* d_status, void* stream)
{
    return n_blocks > ONE_ROUND_BLOCKS ? launch_lanes<true>(d_comp, d_src_off, d_src_len, d_dst_off, n_blocks, d_out, d_status, stream)
                                       : launch_lanes<false>(d_comp, d_src_off, d_src_len, d_dst_off, n_blocks, d_out, d_status, stream);
}

// (the two versions by name: tests and measurements)
extern "C" int svx_bgzf_inflate_lds(const uint8_t* d_comp, const uint64_t* d_src_off, const uint32_t* d_src_len, const uint64_t* d_dst_off,
                                    uint32_t n_blocks, uint8_t* d_out, uint32_t* d_status, void* stream)
{
    return launch_lanes<false>(d_comp, d_src_off, d_src_len, d_dst_off, n_blocks, d_out, d_status, stream);
}
extern "C" int svx_bgzf_inflate_private(const uint8_t* d_comp, const uint64_t* d_src_off, const uint32_t* d_src_len, const uint64_t* d_dst_off,
                                        uint32_t n_blocks, uint8_t* d_out, uint32_t* d_status, void* stream)
{
    return launch_lanes<true>(d_comp, d_src_off, d_src_len, d_dst_off, n_blocks, d_out, d_status, stream);
}

// The same contract with the wave-per-block kernel (uniform control flow: bit buffer and table walk on the scalar unit, the
// lanes as input / output / match-copy engine).  Measured 19.8 GB/s against the lane-per-block kernel's 31.8 at 113 k blocks
// (it is bound by the CU's one scalar issue per cycle: ~40 scalar instructions per symbol), level with it at <= 28 k blocks:
// kept as the second implementation of the interface (tests run both), not the default.
extern "C" int svx_bgzf_inflate_wave(const uint8_t* d_comp, const uint64_t* d_src_off, const uint32_t* d_src_len, const uint64_t* d_dst_off,
                                     uint32_t n_blocks, uint8_t* d_out, uint32_t* d_status, void* stream)
{
    if (n_blocks == 0) return SVX_OK;
    if (!d_comp || !d_src_off || !d_src_len || !d_dst_off || !d_out || !d_status) return SVX_EINVAL;
    if (reinterpret_cast<uintptr_t>(d_comp) & 15u) return SVX_EINVAL;
    hipLaunchKernelGGL(bgzf_inflate_wave_kernel, dim3(n_blocks), dim3(LANES), 0, static_cast<hipStream_t>(stream),
                       d_comp, d_src_off, d_src_len, d_dst_off, n_blocks, d_out, d_status, 0u);
    return hipGetLastError() == hipSuccess ? SVX_OK : SVX_ELAUNCH;
}

// (library-internal) the wave-per-block kernel on the blocks whose status is `only`, which it replaces by its own verdict
extern "C" __attribute__((visibility("hidden")))
int svx_bgzf_inflate_wave_only(const uint8_t* d_comp, const uint64_t* d_src_off, const uint32_t* d_src_len, const uint64_t* d_dst_off,
                               uint32_t n_blocks, uint8_t* d_out, uint32_t* d_status, uint32_t only, void* stream)
{
    hipLaunchKernelGGL(bgzf_inflate_wave_kernel, dim3(n_blocks), dim3(LANES), 0, static_cast<hipStream_t>(stream),
                       d_comp, d_src_off, d_src_len, d_dst_off, n_blocks, d_out, d_status, only);
    return hipGetLastError() == hipSuccess ? SVX_OK : SVX_ELAUNCH;
}

// compile-time check of the arithmetic length / distance tables against the RFC's
namespace {
constexpr bool tables_agree()
{
    for (int li = 0; li < 29; ++li) {
        const unsigned e = li < 8 || li == 28 ? 0u : (unsigned)(li >> 2) - 1u;
        const unsigned b = li < 8 ? 3u + (unsigned)li : li == 28 ? 258u : 3u + ((4u + ((unsigned)li & 3u)) << ((unsigned)(li >> 2) - 1u));
        if (e != LEN_EXTRA[li] || b != LEN_BASE[li]) return false;
    }
    for (int ds = 0; ds < 30; ++ds) {
        const unsigned e = ds < 4 ? 0u : (unsigned)(ds >> 1) - 1u;
        const unsigned b = ds < 4 ? 1u + (unsigned)ds : 1u + ((2u + ((unsigned)ds & 1u)) << ((unsigned)(ds >> 1) - 1u));
        if (e != DIST_EXTRA[ds] || b != DIST_BASE[ds]) return false;
    }
    return true;
}
static_assert(tables_agree(), "length / distance code arithmetic differs from RFC 1951 3.2.5");
}  // namespace
