// svx_crc.hip -- CRC32 of the inflated BGZF blocks on gfx950, checked against the blocks' footers.
//
// htslib verifies the CRC32 of every BGZF block it inflates (behind aln_file.fetch, run_collection.py:23-26; RFC 1952
// 2.3.1: polynomial 0xEDB88320 reflected, initial value and final xor 0xFFFFFFFF).  A block that inflates to exactly its
// ISIZE bytes and keeps the record chain intact would otherwise pass a flipped bit in a CIGAR length silently.
//
// One wave per block, coalesced.  The block's bytes are read as dwords, lane l taking dwords l, l + 64, l + 128, ... of the
// block RIGHT-ALIGNED in a virtual buffer of 16,384 dwords (a CRC register that starts at 0 is not changed by leading
// zeros, so every block, whatever its size, ends at the same virtual position and the constants are the same for all
// blocks).  A CRC is linear over GF(2): the lane keeps the register R of its own sparse subsequence -- every step is
// R <- R * x^2048 + dword (the 64 dwords to the lane's next one): four table look-ups in LDS, slicing-by-4 with tables built
// for that stride -- then R * x^(32 (64 - l)) accounts for the dwords behind the lane's last one and the x^32 every CRC
// owes (one 32-step carry-less multiply by a per-lane constant), and the 64 registers are xor-ed together.  The initial
// 0xFFFFFFFF is the complement of the block's first four bytes, the final xor a complement of the result.
// (tools / tests: the same arithmetic in NumPy against zlib.crc32, tests/test_crc_cpu.py.)
//
// Algorithmic bytes: ISIZE read once per block (HBM read bound; 9.3 GB for the bench's 20-window file); look-ups: 1 per
// byte, in LDS.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>
#include "../../include/svx.h"

namespace {

constexpr uint32_t POLY = 0xEDB88320u;
constexpr int LANES = 64, VDWORDS = 16384, WAVES = 4;        // a BGZF block inflates to at most 65,536 bytes

// reflected arithmetic: bit 31 of a register is the coefficient of x^0, bit 0 that of x^31
__host__ __device__ inline uint32_t times_x(uint32_t r, int bits)
{
    for (int k = 0; k < bits; ++k) r = (r >> 1) ^ ((r & 1u) ? POLY : 0u);
    return r;
}

struct Tables {
    uint32_t stride[4][256];                                 // stride[k][b] = (b << 8k) * x^2048 mod P
    uint32_t tail[LANES];                                    // x^(32 (64 - l)) mod P
};

// built once per process on the host (2 M shift steps), one copy in the memory of every device that asks for it
const Tables* device_tables()
{
    static Tables* host = nullptr;
    static std::once_flag once;
    static std::mutex lock;
    static const Tables* d_tab[64] = {};
    std::call_once(once, [] {
        Tables* h = new Tables;
        for (int k = 0; k < 4; ++k)
            for (int b = 0; b < 256; ++b) {
                // b << 8k advanced by 2048 bits: 8 bits at a time through the first table (stride[..] of 8 bits = the classic table)
                h->stride[k][b] = times_x((uint32_t)b << (8 * k), 2048);
            }
        for (int l = 0; l < LANES; ++l) h->tail[l] = times_x(0x80000000u, 32 * (LANES - l));
        host = h;
    });
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> g(lock);
    if (!d_tab[dev]) {
        void* d = nullptr;
        if (hipMalloc(&d, sizeof(Tables)) == hipSuccess && hipMemcpy(d, host, sizeof(Tables), hipMemcpyHostToDevice) == hipSuccess)
            d_tab[dev] = static_cast<const Tables*>(d);
    }
    return d_tab[dev];
}

// a * b mod P (32 shift-and-add steps)
__device__ __forceinline__ uint32_t gf_mul(uint32_t a, uint32_t b)
{
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        r ^= (a & 0x80000000u) ? b : 0u;
        a <<= 1;
        b = (b >> 1) ^ ((b & 1u) ? POLY : 0u);
    }
    return r;
}

__global__ __launch_bounds__(LANES * WAVES)
void bgzf_crc32_kernel(const uint8_t* __restrict__ out, const uint64_t* __restrict__ dst_off, const uint8_t* __restrict__ comp,
                       const uint64_t* __restrict__ src_off, const uint32_t* __restrict__ src_len, uint32_t n_blocks,
                       uint32_t* __restrict__ status, const Tables* __restrict__ tab)
{
    __shared__ uint32_t t[4][256];
    for (int i = threadIdx.x; i < 1024; i += LANES * WAVES) t[i >> 8][i & 255] = tab->stride[i >> 8][i & 255];
    __syncthreads();
    const uint32_t b = blockIdx.x * WAVES + (threadIdx.x >> 6);
    if (b >= n_blocks) return;
    const int lane = threadIdx.x & 63;
    const uint64_t lo = dst_off[b];
    const uint64_t len64 = dst_off[b + 1] - lo;
    if (len64 > 65536u) {                                     // not a BGZF block
        if (lane == 0 && status[b] == 0) status[b] = SVX_INFLATE_BAD_CRC;
        return;
    }
    const uint32_t len = (uint32_t)len64;
    uint32_t got;
    if (len < 4) {                                            // the EOF marker (0 bytes) and other tiny blocks: byte by byte
        uint32_t r = 0xffffffffu;
        for (uint32_t k = 0; k < len; ++k) r = times_x(r ^ out[lo + k], 8);
        got = ~r;
    } else {
        const uint32_t pad = 65536u - len;                    // virtual byte v is data byte v - pad
        const uint8_t* base = out + lo - pad;                 // (never dereferenced in front of out + lo)
        uint32_t R = 0;
        for (uint32_t g = (pad >> 2) / LANES; g < VDWORDS / LANES; ++g) {
            const uint32_t v = 4u * (g * LANES + (uint32_t)lane);
            uint32_t d = 0;
            if (v >= pad) {
                __builtin_memcpy(&d, base + v, 4);            // (unaligned when ISIZE is not a multiple of 4)
            } else if (v + 4 > pad) {                         // the dword that straddles the start of the data
                for (uint32_t k = pad - v; k < 4; ++k) d |= (uint32_t)base[v + k] << (8 * k);
            }
            // initial value 0xFFFFFFFF = the first four data bytes complemented: bytes [pad, pad + 4) of the virtual buffer
            const int first = (int)pad - (int)v;              // index in this dword of the first data byte
            if (first > -4 && first < 4) {
                const uint32_t m = first >= 0 ? 0xffffffffu << (8 * first) : 0xffffffffu >> (8 * -first);
                d ^= m;
            }
            R = t[0][R & 255u] ^ t[1][(R >> 8) & 255u] ^ t[2][(R >> 16) & 255u] ^ t[3][R >> 24] ^ d;
        }
        uint32_t acc = gf_mul(R, tab->tail[lane]);
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) acc ^= (uint32_t)__shfl_xor((int)acc, s, 64);
        got = ~acc;
    }
    if (lane == 0) {
        const uint8_t* f = comp + src_off[b] + src_len[b];    // the footer follows the payload: CRC32, ISIZE
        const uint32_t want = (uint32_t)f[0] | (uint32_t)f[1] << 8 | (uint32_t)f[2] << 16 | (uint32_t)f[3] << 24;
        if (got != want && status[b] == 0) status[b] = SVX_INFLATE_BAD_CRC;
    }
}

}  // namespace

extern "C" int svx_bgzf_crc32(const uint8_t* d_out, const uint64_t* d_dst_off, const uint8_t* d_comp, const uint64_t* d_src_off,
                              const uint32_t* d_src_len, uint32_t n_blocks, uint32_t* d_status, void* stream)
{
    if (n_blocks == 0) return SVX_OK;
    if (!d_out || !d_dst_off || !d_comp || !d_src_off || !d_src_len || !d_status) return SVX_EINVAL;
    const Tables* tab = device_tables();
    if (!tab) return SVX_ELAUNCH;
    hipLaunchKernelGGL(bgzf_crc32_kernel, dim3((n_blocks + WAVES - 1) / WAVES), dim3(LANES * WAVES), 0, static_cast<hipStream_t>(stream),
                       d_out, d_dst_off, d_comp, d_src_off, d_src_len, n_blocks, d_status, tab);
    return hipGetLastError() == hipSuccess ? SVX_OK : SVX_ELAUNCH;
}
