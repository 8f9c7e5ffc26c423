// svx_lz_core.hpp -- the per-block LZ copy loop of bgzf_lz_kernel (svx_inflate2.hip), host- and device-compilable: the lockstep
// CPU model (tools/exp/spec_inflate_sim.cpp, LZCORE=1) runs this very code against zlib.
//
// Input: a block's LZ sequence stream (svx_inflate2.hip: u32 header [literals:8 | match length:9 | distance - 1:15] + the
// literal bytes).  Output: the block's bytes out[lo, hi).
//
// Round-4 first version: every step was an unaligned 8-byte global store followed by a global load of the next match's
// source -- a dependent memory round trip per step, and (PMC) 8-9 x the output's bytes through the fabric, because 85 k lanes
// each dribble 8 bytes every few microseconds and their partly written lines do not survive in L2.  Here a lane keeps its
// most recent RING bytes of output in its own slice of LDS:
//   * literals and matches are written to the ring (byte stores: no alignment cases);
//   * a match whose source lies within the ring (distance <= NEAR) is read from it -- a few LDS round trips instead of a
//     store acknowledgement + a load from HBM; overlapping matches (distance < length) replicate the period in a register;
//   * a farther match reads global memory: everything below the last 64-byte boundary has been stored;
//   * output leaves as whole 64-byte lines (four aligned 16-byte stores), read back from the ring when a line completes (the
//     block's first and last partial line byte-wise: their neighbours belong to other lanes);
//   * the next sequence's header and its first 12 literal bytes are one 16-byte load issued when the current header is parsed
//     (pulling the stream towards the L2 hundreds of bytes ahead of its use was measured: no gain -- what a turn waits for are
//     its ~200 dependent instructions, the LDS round trips and the far matches' loads).
#pragma once
#ifndef SVX_LZ_DBG
#define SVX_LZ_DBG 0                 // measurements only: bit 0 = far sources are not loaded, bit 1 = lines are not stored (wrong output)
#endif
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define SVX_HD __host__ __device__ __forceinline__
#else
#define SVX_HD inline
#endif

namespace svx_lz {

constexpr uint32_t RING = 256;                       // bytes of recent output per lane (a power of two, a multiple of 16)
constexpr uint32_t NEAR = RING - 16;                 // matches up to this distance read the ring (farther ones read memory: everything
                                                     // below the last 64-byte boundary has been stored, i.e. everything more than 63 bytes back)
enum { LZ_OK = 0, LZ_OUT_OVERRUN = 5, LZ_SHORT = 7, LZ_BAD_DIST = 8 };

struct Seq16 { uint32_t w[4]; };                     // a header and the 12 bytes behind it

SVX_HD void load16(Seq16& d, const uint8_t* p) { memcpy(&d, p, 16); }

// n <= 8 bytes of v to the ring at output position w
SVX_HD void ring_put(uint8_t* ring, uint64_t w, uint64_t v, uint32_t n)
{
#pragma unroll
    for (uint32_t i = 0; i < 8; ++i)
        if (i < n) ring[(uint32_t)(w + i) & (RING - 1)] = (uint8_t)(v >> (8 * i));
}

// eight bytes of the ring from output position s (bytes behind the newest one are whatever the ring holds): three aligned
// dwords and a funnel shift (eight byte reads + seven shift-ors before)
SVX_HD uint64_t ring_get8(const uint8_t* ring, uint64_t s)
{
    const uint32_t* r32 = reinterpret_cast<const uint32_t*>(__builtin_assume_aligned(ring, 16));
    const uint32_t o = (uint32_t)s & (RING - 1), d = o >> 2, sh = (o & 3u) * 8u;
    const uint32_t a = r32[d], b = r32[(d + 1) & (RING / 4 - 1)], c = r32[(d + 2) & (RING / 4 - 1)];
    const uint64_t lo = ((uint64_t)b << 32 | a) >> sh;
    return sh ? lo | (uint64_t)c << (64 - sh) : lo;
}

// the 64-byte line that ends at the boundary `upto` (a multiple of LINE in the address space of `out`) leaves the ring: four
// aligned 16-byte stores back to back -- a whole line of the memory system (16-byte chunks, one at a time, still cost 2.5 x the
// output's bytes in write traffic: PMC WRITE_SIZE)
constexpr uint32_t LINE = 64;
SVX_HD void flush_line(const uint8_t* ring, uint8_t* out, uint64_t upto, uint64_t lo)
{
    const uint64_t from = upto - LINE;
    const uint32_t r = (uint32_t)from & (RING - 1);  // LINE-aligned in the ring as well (ring offsets = positions mod RING,
    if (from >= lo) {                                // the ring itself is 16-byte aligned)
        uint32_t c[LINE / 4];
        memcpy(c, static_cast<const uint8_t*>(__builtin_assume_aligned(ring, 16)) + r, LINE);
        if (!(SVX_LZ_DBG & 2)) memcpy(out + from, c, LINE);
    } else {                                         // the block's first line: the bytes in front of lo are another block's
        for (uint64_t a = lo; a < upto; ++a) out[a] = ring[(uint32_t)a & (RING - 1)];
    }
}

// -> LZ_OK or an error; `ring`: RING bytes of scratch owned by the caller (LDS on the device).
// One loop, every lane of a wave in its own state (the lanes' blocks are at unrelated places): a turn parses a header if one is
// due, then takes one step of up to eight literal bytes and / or one step of up to eight match bytes.
SVX_HD int decode_block(const uint8_t* stream, uint32_t stream_len, uint8_t* out, uint64_t lo, uint64_t hi, uint8_t* ring)
{
    const uint8_t* p = stream;
    const uint8_t* const p_end = stream + stream_len;
    uint64_t w = lo;
    Seq16 nx;
    load16(nx, p);                                   // (behind the stream's end: the next slot or the workspace's slack)
    uint32_t lit = 0, mlen = 0, dist = 0, done = 0, lit_all = 0, lv_hi = 0;
    uint64_t lv = 0, fv = 0;
    const uint8_t* lp = p;
    bool near = false, pre = false;
    for (;;) {
        if (lit == 0 && mlen == 0) {                 // next sequence
            if (p >= p_end) break;
            const uint32_t h = nx.w[0];
            lit = h & 255u; mlen = (h >> 8) & 511u; dist = (h >> 17) + 1u;
            p += 4;
            if (w + lit + mlen > hi || p + lit > p_end) return LZ_OUT_OVERRUN;
            if (mlen && dist > w + lit - lo) return LZ_BAD_DIST;
            lv = (uint64_t)nx.w[1] | (uint64_t)nx.w[2] << 32;          // the first 8 literal bytes (12 are in hand)
            lv_hi = nx.w[3];
            lp = p; lit_all = lit; done = 0;
            p += lit;
            near = dist <= NEAR;
            // a far match's first eight source bytes are requested now if they are in memory already: everything below the last
            // 64-byte boundary in front of w is (the literals of this sequence are not)
            pre = mlen != 0 && !near && dist >= lit + 8u + 64u;
            if (pre && !(SVX_LZ_DBG & 1)) memcpy(&fv, out + (w + lit - dist), 8);
            load16(nx, p);                           // the next header + its first literals
        }
        if (lit) {                                   // a step of the literal run
            const uint32_t n = lit < 8u ? lit : 8u;
            uint64_t v;
            if (done == 0) v = lv;
            else if (done == 8 && lit_all <= 12u) v = lv_hi;
            else { v = 0; memcpy(&v, lp + done, 8); }
            ring_put(ring, w, v, n);
            const uint64_t w2 = w + n;
            if ((w2 & ~63ull) != (w & ~63ull)) flush_line(ring, out, w2 & ~63ull, lo);
            w = w2; done += n; lit -= n;
        }
        if (lit == 0 && mlen) {                      // a step of the match
            const uint32_t n = mlen < 8u ? mlen : 8u;
            uint64_t v;
            if (near) {
                v = ring_get8(ring, w - dist);
                if (dist < 8u) {                     // the period of `dist` bytes, repeated through the eight
                    const uint32_t sh = 8u * dist;
                    uint64_t q = v & ((1ull << sh) - 1ull);
                    q |= q << sh;
                    if (2u * sh < 64u) q |= q << (2u * sh);
                    if (4u * sh < 64u) q |= q << (4u * sh);
                    v = q;
                }
            } else if (pre) { v = fv; pre = false; }
            else if (!(SVX_LZ_DBG & 1)) memcpy(&v, out + (w - dist), 8);
            else v = 0;
            ring_put(ring, w, v, n);
            const uint64_t w2 = w + n;
            if ((w2 & ~63ull) != (w & ~63ull)) flush_line(ring, out, w2 & ~63ull, lo);
            w = w2; mlen -= n;
        }
    }
    if (w != hi) return LZ_SHORT;
    for (uint64_t a = (w & ~63ull) > lo ? (w & ~63ull) : lo; a < hi; ++a) out[a] = ring[(uint32_t)a & (RING - 1)];     // the last, partial line
    return LZ_OK;
}

}  // namespace svx_lz
