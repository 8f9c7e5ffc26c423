// spec_inflate_sim.cpp -- lockstep CPU model of the speculative wave-per-block BGZF inflate (svx_inflate_spec.hip).
//
// Design study, not product and not oracle: 64 "lanes" decode ONE raw-DEFLATE stream together.  The compressed bits of a
// DEFLATE block are cut into chunks of 64 segments of S bits; lane i starts decoding at segment i's first bit although
// that is usually not a token boundary (pass S1): Huffman streams re-synchronise within a few dozen bits, so the lane's
// LAST token boundary -- the first one at or behind the next segment's start -- is almost always a true one.  Pass S2
// re-decodes every segment from its predecessor's end (lane 0 from the chunk's true start) and repeats for the lanes whose
// start moved until nothing moves (lane 0 is true, so lane k is true after at most k rounds; in practice two); it also
// sums the output length of every segment.  A prefix sum gives every lane its output offset, and pass S3 decodes a third
// time and writes: literals at once, matches in steps of <= 8 bytes as soon as their source bytes exist -- every lane
// publishes how far it has written (w[lane]) after every turn, a match remembers the lane j its source currently lies in
// and a step never crosses that lane's segment end.
//
// The model keeps the GPU's visibility rules: loads of a turn see the stores of earlier turns only; a wide (8-byte) store may
// put garbage behind the step's last byte inside the lane's own segment.  Usage: spec_inflate_sim file.bam [S] [max_blocks]
#include <zlib.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../svision_amd/csrc/svx_lz_core.hpp"

static const int LANES = 64, LB = 10, DB = 8;
enum { K_BAD = 0, K_LIT = 1, K_LEN = 2, K_EOB = 3, K_SUB = 4, K_DIST = 2 };
static const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
static const uint8_t CLEN_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

struct Stream {
    const uint8_t* d; int64_t nbits;
    uint64_t peek(int64_t p) const {                      // >= 57 bits from bit position p (zeros behind the end)
        uint64_t v = 0; int64_t b = p >> 3;
        for (int i = 0; i < 8; ++i) { int64_t a = b + i; uint64_t x = (a * 8 < nbits + 7 && a >= 0 && a < (nbits + 7) / 8) ? d[a] : 0; v |= x << (8 * i); }
        return v >> (p & 7);
    }
};
static uint32_t rev(uint32_t c, int n) { uint32_t r = 0; for (int i = 0; i < n; ++i) r |= ((c >> i) & 1u) << (n - 1 - i); return r; }

// entry: bits 0..3 code bits at this level, 4..7 kind, 8.. payload
struct Tables { uint32_t lit[(1 << LB) + 1024], dist[(1 << DB) + 1024]; int lit_n, dist_n; };
static uint32_t lit_payload(int s) {
    if (s < 256) return K_LIT << 4 | (uint32_t)s << 8;
    if (s == 256) return K_EOB << 4;
    if (s - 257 >= 29) return K_BAD << 4;
    return K_LEN << 4 | (uint32_t)LEN_BASE[s - 257] << 8 | (uint32_t)LEN_EXTRA[s - 257] << 17;
}
static uint32_t dist_payload(int s) { return s >= 30 ? (uint32_t)K_BAD << 4 : (K_DIST << 4 | (uint32_t)DIST_BASE[s] << 8 | (uint32_t)DIST_EXTRA[s] << 23); }
static bool build(const uint8_t* lens, int n, int root, uint32_t* tab, int cap, int* used, uint32_t (*payload)(int)) {
    int count[16] = {0};
    for (int s = 0; s < n; ++s) count[lens[s]]++;
    count[0] = 0;
    int left = 1; uint32_t next[16], code = 0;
    for (int l = 1; l <= 15; ++l) { left <<= 1; left -= count[l]; if (left < 0) return false; code = (code + count[l - 1]) << 1; next[l] = code; }
    for (int i = 0; i < (1 << root); ++i) tab[i] = 0;
    int top = 1 << root;
    // long codes first pass: how many extra bits each root prefix needs
    std::vector<int> sub_bits(1 << root, 0);
    for (int s = 0; s < n; ++s) { int l = lens[s]; if (l > root) { uint32_t c = next[l] + 0; (void)c; } }
    uint32_t nx[16]; memcpy(nx, next, sizeof nx);
    std::vector<uint32_t> codes(n, 0);
    for (int s = 0; s < n; ++s) if (lens[s]) codes[s] = nx[lens[s]]++;
    for (int s = 0; s < n; ++s) { int l = lens[s]; if (l > root) { uint32_t r = rev(codes[s], l); int pre = r & ((1 << root) - 1); if (l - root > sub_bits[pre]) sub_bits[pre] = l - root; } }
    for (int pre = 0; pre < (1 << root); ++pre) if (sub_bits[pre]) {
        if (top + (1 << sub_bits[pre]) > cap) return false;
        tab[pre] = K_SUB << 4 | (uint32_t)top << 8 | (uint32_t)sub_bits[pre] << 20;
        for (int i = 0; i < (1 << sub_bits[pre]); ++i) tab[top + i] = 0;
        top += 1 << sub_bits[pre];
    }
    for (int s = 0; s < n; ++s) {
        int l = lens[s]; if (!l) continue;
        uint32_t r = rev(codes[s], l);
        if (l <= root) { for (uint32_t i = r; i < (1u << root); i += 1u << l) tab[i] = payload(s) | (uint32_t)l; }
        else {
            int pre = r & ((1 << root) - 1); uint32_t e = tab[pre]; int sb = (e >> 20) & 15; uint32_t start = (e >> 8) & 0xfff;
            for (uint32_t i = r >> root; i < (1u << sb); i += 1u << (l - root)) tab[start + i] = payload(s) | (uint32_t)(l - root);
        }
    }
    *used = top;
    return true;
}

struct Tok { int kind; int used; uint32_t val, len, dist; };   // kind: K_LIT / K_LEN (match) / K_EOB / K_BAD
static Tok token(const Stream& in, const Tables& t, int64_t p) {
    uint64_t bits = in.peek(p);
    uint32_t e = t.lit[bits & ((1 << LB) - 1)];
    int used;
    if (((e >> 4) & 15) == K_SUB) { uint32_t e2 = t.lit[((e >> 8) & 0xfff) + ((bits >> LB) & ((1u << ((e >> 20) & 15)) - 1))]; used = LB + (e2 & 15); e = e2; }
    else used = e & 15;
    int kind = (e >> 4) & 15;
    Tok k{kind, used, 0, 0, 0};
    if (kind == K_LIT) { k.val = (e >> 8) & 255; return k; }
    if (kind != K_LEN) return k;                          // EOB or BAD
    int xb = (e >> 17) & 7;
    k.len = ((e >> 8) & 511) + (uint32_t)((bits >> used) & ((1u << xb) - 1));
    used += xb;
    uint32_t d = t.dist[(bits >> used) & ((1 << DB) - 1)];
    if (((d >> 4) & 15) == K_SUB) { uint32_t d2 = t.dist[((d >> 8) & 0xfff) + ((bits >> (used + DB)) & ((1u << ((d >> 20) & 15)) - 1))]; used += DB + (d2 & 15); d = d2; }
    else used += d & 15;
    if (((d >> 4) & 15) != K_DIST) { k.kind = K_BAD; k.used = used; return k; }
    int db = (d >> 23) & 15;
    k.dist = ((d >> 8) & 0x7fff) + (uint32_t)((bits >> used) & ((1u << db) - 1));
    used += db;
    k.used = used;
    return k;
}

static bool g_seq = false;
static std::vector<uint8_t> g_stream;           // the block's LZ sequence stream (kernel A's output, kernel B's input)

// kernel B: one lane, the block's sequences in order
static int lz_replay(const std::vector<uint8_t>& s, uint8_t* out, size_t cap, size_t* produced, long* steps) {
    size_t p = 0, w = 0;
    while (p < s.size()) {
        uint32_t h; memcpy(&h, &s[p], 4); p += 4;
        uint32_t lit = h & 255, mlen = (h >> 8) & 511, dist = (h >> 17) + 1;
        if (w + lit + mlen > cap || p + lit > s.size()) return 95;
        for (uint32_t k = 0; k < lit; k += 8) ++*steps;
        memcpy(out + w, &s[p], lit); p += lit; w += lit;
        if (mlen) { if (dist > w) return 8; for (uint32_t k = 0; k < mlen; ++k) { out[w] = out[w - dist]; ++w; } for (uint32_t k = 0; k < mlen; k += 8) ++*steps; }
        ++*steps;
    }
    *produced = w; return 0;
}

struct Stats { long chunks = 0, s1_turns = 0, s2_turns = 0, s2_rounds = 0, s3_turns = 0, tokens = 0, stall = 0, headers = 0; };

// -> 0 ok, else error code; out must have room for `cap` bytes (+8 slack)
static int spec_inflate(const uint8_t* src, size_t n, uint8_t* out, size_t cap, size_t* produced, int S, Stats& st) {
    Stream in{src, (int64_t)n * 8};
    int64_t P = 0; size_t W = 0;
    static Tables t;
    for (;;) {
        uint64_t h = in.peek(P);
        int last = h & 1, type = (h >> 1) & 3; P += 3;
        st.headers++;
        if (type == 0) {
            P = (P + 7) & ~7ll;
            uint64_t v = in.peek(P); uint32_t len = v & 0xffff, nlen = (v >> 16) & 0xffff; P += 32;
            if ((len ^ nlen) != 0xffff) return 2;
            if (W + len > cap) return 5;
            if (P + 8ll * len > in.nbits) return 6;
            if (g_seq) { for (uint32_t k = 0; k < len; k += 255) { uint32_t n2 = len - k < 255 ? len - k : 255; uint32_t h2 = n2; size_t q2 = g_stream.size(); g_stream.resize(q2 + 4 + n2); memcpy(&g_stream[q2], &h2, 4); memcpy(&g_stream[q2 + 4], src + (P >> 3) + k, n2); } }
            else memcpy(out + W, src + (P >> 3), len);
            W += len; P += 8ll * len;
            if (last) break; else continue;
        }
        uint8_t lens[320] = {0};
        int nlen, ndist;
        if (type == 1) { for (int s = 0; s < 144; ++s) lens[s] = 8; for (int s = 144; s < 256; ++s) lens[s] = 9; for (int s = 256; s < 280; ++s) lens[s] = 7; for (int s = 280; s < 288; ++s) lens[s] = 8; for (int s = 0; s < 30; ++s) lens[288 + s] = 5; nlen = 288; ndist = 30; }
        else if (type == 2) {
            uint64_t v = in.peek(P); nlen = (v & 31) + 257; ndist = ((v >> 5) & 31) + 1; int ncode = ((v >> 10) & 15) + 4; P += 14;
            if (nlen > 286 || ndist > 30) return 3;
            uint8_t cl[19] = {0};
            for (int i = 0; i < ncode; ++i) { cl[CLEN_ORDER[i]] = in.peek(P) & 7; P += 3; }
            static uint32_t ct[(1 << 7) + 64]; int cu;
            if (!build(cl, 19, 7, ct, 128 + 64, &cu, [](int s) { return (uint32_t)(K_LIT << 4 | s << 8); })) return 3;
            uint8_t all[320] = {0}; int i = 0;
            while (i < nlen + ndist) {
                uint32_t e = ct[in.peek(P) & 127]; if (((e >> 4) & 15) != K_LIT) return 3; P += e & 15; int sym = (e >> 8) & 255;
                if (sym < 16) { all[i++] = sym; continue; }
                int prev = 0, rep;
                if (sym == 16) { if (!i) return 3; prev = all[i - 1]; rep = 3 + (in.peek(P) & 3); P += 2; }
                else if (sym == 17) { rep = 3 + (in.peek(P) & 7); P += 3; } else { rep = 11 + (in.peek(P) & 127); P += 7; }
                if (i + rep > nlen + ndist) return 3;
                while (rep--) all[i++] = prev;
            }
            memcpy(lens, all, nlen); memcpy(lens + 288, all + nlen, ndist);
            if (lens[256] == 0) return 3;
        } else return 1;
        if (!build(lens, type == 1 ? 288 : nlen, LB, t.lit, (1 << LB) + 1024, &t.lit_n, lit_payload)) return 3;
        if (!build(lens + 288, type == 1 ? 30 : ndist, DB, t.dist, (1 << DB) + 1024, &t.dist_n, dist_payload)) return 3;
        if (getenv("SERIAL")) {
            for (;;) { Tok k = token(in, t, P); P += k.used; if (k.kind == K_LIT) out[W++] = k.val; else if (k.kind == K_LEN) { if (k.dist > W) { fprintf(stderr, "serial: dist %u > W %zu len %u at bit %lld\n", k.dist, W, k.len, (long long)P); return 8; } for (uint32_t x = 0; x < k.len; ++x) { out[W] = out[W - k.dist]; ++W; } } else if (k.kind == K_EOB) break; else return 4; }
            if (last) break; else continue;
        }
        // ---- chunks of 64 segments of S bits
        bool eob = false;
        while (!eob) {
            st.chunks++;
            int64_t q[LANES + 1]; for (int i = 0; i <= LANES; ++i) q[i] = P + (int64_t)i * S;
            int64_t e[LANES], start[LANES], f[LANES]; uint32_t olen[LANES]; int flag[LANES];      // flag: 0 none, 1 eob, 2 error
            // S1
            { int64_t p[LANES]; bool act[LANES]; for (int i = 0; i < LANES; ++i) { p[i] = q[i]; act[i] = true; }
              for (;;) { bool any = false;
                for (int i = 0; i < LANES; ++i) { if (!act[i]) continue; if (p[i] >= q[i + 1] || p[i] >= in.nbits) { act[i] = false; e[i] = p[i]; continue; } any = true;
                    Tok k = token(in, t, p[i]); if (k.kind == K_BAD) p[i] += 1; else p[i] += k.used; }
                if (!any) break; st.s1_turns++; } }
            // S2
            bool dirty[LANES]; for (int i = 0; i < LANES; ++i) { start[i] = i ? e[i - 1] : P; dirty[i] = true; }
            for (;;) {
                st.s2_rounds++;
                int64_t p[LANES]; bool act[LANES];
                for (int i = 0; i < LANES; ++i) { act[i] = dirty[i]; if (act[i]) { p[i] = start[i]; olen[i] = 0; flag[i] = 0; } }
                for (;;) { bool any = false;
                    for (int i = 0; i < LANES; ++i) { if (!act[i]) continue; if (p[i] >= q[i + 1]) { act[i] = false; f[i] = p[i]; continue; } any = true;
                        if (p[i] >= in.nbits) { flag[i] = 2; act[i] = false; f[i] = p[i]; continue; }
                        Tok k = token(in, t, p[i]);
                        if (k.kind == K_BAD) { flag[i] = 2; act[i] = false; f[i] = p[i]; continue; }
                        p[i] += k.used;
                        if (k.kind == K_EOB) { flag[i] = 1; act[i] = false; f[i] = p[i]; continue; }
                        olen[i] += k.kind == K_LIT ? 1 : k.len; }
                    if (!any) break; st.s2_turns++; }
                int first = LANES; for (int i = 0; i < LANES; ++i) if (flag[i]) { first = i; break; }
                bool again = false;
                for (int i = 0; i < LANES; ++i) { dirty[i] = false; if (i && i <= first && f[i - 1] != start[i]) { start[i] = f[i - 1]; dirty[i] = true; again = true; } }
                if (!again) break;
            }
            int first = LANES; for (int i = 0; i < LANES; ++i) if (flag[i]) { first = i; break; }
            if (first < LANES && flag[first] == 2) return 4;
            if (g_seq) {
                // S3': every lane re-decodes its segment and appends LZ sequences -- u32 header [litlen:8 | matchlen:9 | dist-1:15]
                // followed by litlen literal bytes -- to ITS part of the block's stream; sizes first (what S2 would also count)
                size_t enc[LANES + 1]; enc[0] = g_stream.size();
                size_t o2[LANES + 1]; o2[0] = W;
                for (int i = 0; i < LANES; ++i) {
                    size_t bytes = 0; uint32_t lit = 0; int64_t p = start[i];
                    if (i <= first) while (p < f[i]) { Tok k = token(in, t, p); p += k.used;
                        if (k.kind == K_LIT) { if (lit == 255) { bytes += 4 + 255; lit = 0; } ++lit; }
                        else if (k.kind == K_LEN) { bytes += 4 + lit; lit = 0; }
                        else break; }
                    if (lit) bytes += 4 + lit;
                    enc[i + 1] = enc[i] + bytes; o2[i + 1] = o2[i] + (i <= first ? olen[i] : 0);
                }
                if (o2[LANES] > cap) return 5;
                g_stream.resize(enc[LANES]);
                for (int i = 0; i <= first && i < LANES; ++i) {
                    size_t q2 = enc[i]; uint8_t lits[256]; uint32_t lit = 0; int64_t p = start[i]; size_t w = o2[i];
                    auto flush = [&](uint32_t mlen, uint32_t dist) { uint32_t h = lit | mlen << 8 | (dist ? dist - 1 : 0) << 17; memcpy(&g_stream[q2], &h, 4); memcpy(&g_stream[q2 + 4], lits, lit); q2 += 4 + lit; lit = 0; };
                    while (p < f[i]) { Tok k = token(in, t, p); p += k.used; st.tokens++;
                        if (k.kind == K_LIT) { if (lit == 255) flush(0, 0); lits[lit++] = (uint8_t)k.val; w += 1; }
                        else if (k.kind == K_LEN) { if (k.dist > w) return 8; flush(k.len, k.dist); w += k.len; }
                        else break; }
                    if (lit) flush(0, 0);
                    if (q2 != enc[i + 1] || w != o2[i + 1]) return 94;
                }
                W = o2[LANES];
                if (first < LANES) { eob = true; P = f[first]; } else P = f[LANES - 1];
                continue;
            }
            // S3: offsets
            size_t o[LANES + 1]; o[0] = W; for (int i = 0; i < LANES; ++i) o[i + 1] = o[i] + (i <= first ? olen[i] : 0);
            if (o[LANES] > cap) return 5;
            int64_t p[LANES]; size_t w[LANES], wpub[LANES]; bool act[LANES];
            uint32_t mlen[LANES], mdist[LANES]; int mj[LANES];
            for (int i = 0; i < LANES; ++i) { p[i] = start[i]; w[i] = wpub[i] = o[i]; act[i] = i <= first; mlen[i] = 0; }
            for (;;) { bool any = false;
                struct St { int lane; size_t at; int n; uint8_t b[8]; bool wide; }; std::vector<St> stores;
                for (int i = 0; i < LANES; ++i) { if (!act[i]) continue; any = true;
                    if (mlen[i]) {                                   // a step of the pending match
                        size_t a = w[i] - mdist[i];
                        int j = mj[i];
                        while (j < i && a >= o[j + 1]) ++j;          // (the GPU does one hop per turn; same result)
                        mj[i] = j;
                        uint32_t nstep = mlen[i] < 8 ? mlen[i] : 8;
                        if (j < i) { size_t room = o[j + 1] - a; if (room < nstep) nstep = (uint32_t)room; }
                        bool ready = j == i || j < 0 || wpub[j] >= a + nstep;
                        if (!ready) { st.stall++; continue; }
                        St s{i, w[i], (int)nstep, {0}, false};
                        for (uint32_t k = 0; k < nstep; ++k) s.b[k] = out[a + (k % mdist[i])];     // dist < nstep only when the source is the lane's own output
                        s.wide = w[i] + 8 <= o[i + 1];
                        for (int k = nstep; k < 8; ++k) s.b[k] = 0xEE;
                        stores.push_back(s);
                        w[i] += nstep; mlen[i] -= nstep;
                        continue;
                    }
                    if (p[i] == f[i]) { act[i] = false; if (w[i] != o[i + 1]) return 90; continue; }
                    if (p[i] > f[i]) return 91;
                    Tok k = token(in, t, p[i]); p[i] += k.used; st.tokens++;
                    if (k.kind == K_LIT) { St s{i, w[i], 1, {(uint8_t)k.val}, false}; stores.push_back(s); w[i] += 1; }
                    else if (k.kind == K_LEN) {
                        if (k.dist > w[i]) return 8;
                        mlen[i] = k.len; mdist[i] = k.dist;
                        size_t a = w[i] - k.dist; int j = -1;          // largest j with o[j] <= a, -1: in front of the chunk
                        if (a >= o[0]) { int lo = 0, hi = i; while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (o[mid] <= a) lo = mid; else hi = mid - 1; } j = lo; }
                        mj[i] = j;
                    } else if (k.kind == K_EOB) { if (p[i] != f[i]) return 92; }
                    else return 93;
                }
                if (!any) break; st.s3_turns++;
                for (auto& s : stores) { int nn = s.wide ? 8 : s.n; for (int k = 0; k < nn; ++k) out[s.at + k] = s.b[k]; }
                for (int i = 0; i < LANES; ++i) wpub[i] = w[i];
            }
            W = o[LANES];
            if (first < LANES) { eob = true; P = f[first]; } else P = f[LANES - 1];
        }
        if (last) break;
    }
    *produced = W;
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s file.bam [S] [max_blocks]\n", argv[0]); return 2; }
    int S = argc > 2 ? atoi(argv[2]) : 512; long maxb = argc > 3 ? atol(argv[3]) : 1 << 30;
    FILE* fp = fopen(argv[1], "rb"); if (!fp) { perror("open"); return 1; }
    std::vector<uint8_t> raw; { uint8_t buf[1 << 16]; size_t k; while ((k = fread(buf, 1, sizeof buf, fp)) > 0) raw.insert(raw.end(), buf, buf + k); } fclose(fp);
    size_t p = 0; long nb = 0, bad = 0; Stats st; double bytes = 0; long lz_steps = 0; double stream_bytes = 0;
    g_seq = getenv("SEQ") != nullptr;
    std::vector<uint8_t> out(65536 + 64), want(65536 + 64);
    while (p + 18 <= raw.size() && nb < maxb) {
        int xlen = raw[p + 10] | raw[p + 11] << 8; int bsize = -1;
        for (size_t qq = p + 12; qq + 4 <= p + 12 + xlen;) { int sl = raw[qq + 2] | raw[qq + 3] << 8; if (raw[qq] == 66 && raw[qq + 1] == 67) bsize = raw[qq + 4] | raw[qq + 5] << 8; qq += 4 + sl; }
        if (bsize < 0) break;
        size_t end = p + bsize + 1, s0 = p + 12 + xlen, sl = end - 8 - s0;
        uint32_t isize = raw[end - 4] | raw[end - 3] << 8 | raw[end - 2] << 16 | (uint32_t)raw[end - 1] << 24;
        z_stream z; memset(&z, 0, sizeof z); inflateInit2(&z, -15); z.next_in = &raw[s0]; z.avail_in = sl; z.next_out = want.data(); z.avail_out = 65536; int zr = inflate(&z, Z_FINISH); inflateEnd(&z);
        size_t got = 0; memset(out.data(), 0xAA, out.size());
        g_stream.clear();
        int rc = spec_inflate(&raw[s0], sl, out.data(), isize, &got, S, st);
        if (g_seq && !rc && getenv("LZCORE")) {          // the device's own LZ loop (svx_lz_core.hpp), the block placed at an odd offset of a larger buffer
            static std::vector<uint8_t> big(1 << 17); static uint8_t ring[svx_lz::RING];
            const uint64_t lo = 16 * 100 + (nb % 16), hi = lo + isize;
            memset(big.data(), 0xCC, big.size());
            std::vector<uint8_t> st2(g_stream); st2.resize(st2.size() + 32, 0x5A);
            rc = svx_lz::decode_block(st2.data(), (uint32_t)g_stream.size(), big.data(), lo, hi, ring);
            if (!rc) { memcpy(out.data(), big.data() + lo, isize); if (big[lo - 1] != 0xCC || big[hi] != 0xCC) rc = 97; }   // nothing written outside [lo, hi)
            stream_bytes += g_stream.size();
        } else
        if (g_seq && !rc) { size_t got2 = 0; rc = lz_replay(g_stream, out.data(), isize, &got2, &lz_steps); if (!rc && got2 != got) rc = 96; stream_bytes += g_stream.size(); }
        if (rc || got != isize || zr != Z_STREAM_END || memcmp(out.data(), want.data(), isize)) { if (bad < 5) fprintf(stderr, "block %ld: rc %d got %zu isize %u zlib %d\n", nb, rc, got, isize, zr); ++bad; }
        bytes += isize; ++nb; p = end;
    }
    printf("S=%d: %ld blocks, %ld differ; per block: %.1f headers, %.1f chunks, turns S1 %.0f S2 %.0f (%.2f rounds/chunk) S3 %.0f (stalled lane-turns %.0f), tokens %.0f; bytes %.0f\n", S, nb, bad,
           (double)st.headers / nb, (double)st.chunks / nb, (double)st.s1_turns / nb, (double)st.s2_turns / nb, (double)st.s2_rounds / st.chunks, (double)st.s3_turns / nb, (double)st.stall / nb, (double)st.tokens / nb, bytes / nb);
    if (g_seq) printf("  sequence stream: %.0f bytes per block (%.2f x the output), kernel-B steps per block %.0f\n", stream_bytes / nb, stream_bytes / bytes, (double)lz_steps / nb);
    return bad != 0;
}
