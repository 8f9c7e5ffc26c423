// svx_conv.hip -- fp32 implicit-GEMM convolution on the gfx950 matrix cores (MI355X).
//
// The dense contractions of the reference CNN after the first layer: tf.nn.conv2d, stride 1, SAME
// padding, optional 2-way channel groups (src/network/alexnet.py:34,39,42,45 via :109-129), with the
// bias + ReLU of :132-135 fused in the epilogue.
//
// GEMM view per group:   D[n][m] = sum_k  Wt[n][k] * X[k][m]
//   n = output channel in the group, m = (image, y, x) output pixel, k = (ky, kx, c).
//
// Layouts ("C8", include/svx.h): activations [image][C/8][H][W][8] -- the 8 channels of an octet are the 32-byte
// sector of a pixel -- and weights packed once per model as [ky][kx][Cin_g/8][Cout][8] (svx.h: from the checkpoint's
// HWIO by splitting the input-channel axis).  With them every operand fetch of a lane is ONE 16-byte load that feeds
// four MFMAs (lane l of the wave holds channels 4*(l>>5) .. +3 of an octet: k-pair j of the octet multiplies channel
// 4*(l>>5)+j), a wave's load covers 32 consecutive pixels x 32 B = 1 KB of contiguous memory in dense mode and whole
// sectors of gathered pixels in list mode, and the epilogue stores whole sectors too (no partial-line writes from
// scattered pixels).  Measured motivation: with 4-byte fragment loads (NCHW / HWIO) the kernel was bound by the
// vector-memory address path -- one wave-wide dword load per MFMA costs the texture addresser about as many cycles as
// four times the data in one dwordx4 -- not by the matrix pipe: shapes with more activation loads per MFMA were slower
// at equal MFMA count, and launches overlapped on several streams gained nothing.
//
// Design: one WAVE owns one output tile of 32*NA channels x 32*NB pixels (NA*NB accumulators of
// v_mfma_f32_32x32x2_f32) and feeds itself: no LDS staging, no barrier, no inter-wave dependency.  The fp32 MFMA
// runs at the vector rate (64 cycles per instruction per SIMD), so the operand traffic per FLOP out of L1/L2 is what
// an LDS-staged 64 x 128 workgroup tile needs (round 1), without its costs: the slice barrier every 16 MFMAs, the
// LDS round trip, the select per staged value and the tile-count quantisation of 4 lock-stepped waves.
//   * Loads are buffer loads (wave-uniform descriptor + per-lane byte offset + uniform SGPR offset): the k loop
//     advances the two SGPR offsets only -- no per-load VALU -- and a tap that falls outside the image, or a
//     column past the end of the pixel list, gets a per-lane offset beyond the descriptor's range, which the
//     hardware answers with 0.0: SAME padding costs nothing in the loop and nothing is selected afterwards.
//   * Operands are fetched two octets (8 k-pairs, >= 1000 cycles of MFMA work) ahead into a ring of three register
//     sets, statically indexed by unrolling three octets; the loads of an octet are spread between the MFMAs of
//     another (a wave issues in order: a cluster of loads longer than one MFMA's 64 cycles would idle the pipe).
//   * k order per output element is fixed -- (octet, ky, kx, j, half) in the 3x3 layers, (ky, kx, octet, j, half) in the
//     5x5 one -- in every shape and mode: the active-set path
//     stays bit-identical to the dense path.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/svx.h"

#include "svx_shadow.hpp"

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int THREADS = 256, WAVES = THREADS / 64;
constexpr int FILL_PIX = 128, FILL_OCT = 8;       // background copy unit of one workgroup: 128 pixels x 8 octets (64 channels)
constexpr unsigned OOB = 0x80000000u;             // per-lane byte offset no descriptor covers (tensors are < 2 GB)

// wave tile shapes (NA x 32 channels, NB x 32 pixels), in order of preference at equal cost.  Measured on MI355X
// (tools/ab_conv.py, tools/stage_bench.py): two small waves per SIMD cover each other's stalls better than one big wave
// covers its own, weight rows are cheaper to fetch than activation columns, and 64 x 32 (two accumulators) is the best
// or within 2 % of the best shape for every layer once several launches overlap (graph replays on 3-4 streams); the
// larger shapes only win a launch running alone whose tile count they happen to quantise better.
constexpr int N_SHAPES = 5;
constexpr int SHAPE_NA[N_SHAPES] = {2, 1, 2, 2, 1};
constexpr int SHAPE_NB[N_SHAPES] = {1, 3, 3, 2, 2};
#ifndef SVX_CONV_TAP_INNER
#define SVX_CONV_TAP_INNER 1
#endif
#ifndef SVX_CONV_XCD_2D
#define SVX_CONV_XCD_2D 1
#endif
#ifndef SVX_LIST_SHAPES
#define SVX_LIST_SHAPES 2
#endif
constexpr int LIST_SHAPES = SVX_LIST_SHAPES;   // list mode (pixel count known on the device only) picks among the first two

struct ConvArgs {
    const float* in; const float* w; const float* bias; float* out;
    int nimg, Cin, Cout, H, W, groups, relu;
    const int32_t* pixels; const uint32_t* pixel_count; const float* background;
    int n_simd;                                   // SIMDs of the device (4 per CU)
};

// all counts fit 32 bits: the input tensor is < 2 GB, so there are < 2^25 pixels and < 2^24 tiles of any kind
__host__ __device__ inline int conv_wave_tiles(int M, int cout_g, int groups, int shape)
{
    return ((M + 32 * SHAPE_NB[shape] - 1) / (32 * SHAPE_NB[shape])) * groups * (cout_g / (32 * SHAPE_NA[shape]));
}

// compute workgroups of a launch (a multiple of 8: the same number on every XCD)
inline int conv_compute_wgs(int M, int cout_g, int groups, int shape)
{
#if SVX_CONV_XCD_2D
    const int m_tiles = (M + 32 * SHAPE_NB[shape] - 1) / (32 * SHAPE_NB[shape]);
    const int ny = groups * (cout_g / (32 * SHAPE_NA[shape])), halves = (ny & 1) ? 1 : 2;
    return 8 * ((((m_tiles * halves + 7) / 8) * (ny / halves) + WAVES - 1) / WAVES);
#else
    return 8 * (((conv_wave_tiles(M, cout_g, groups, shape) + WAVES - 1) / WAVES + 7) / 8);
#endif
}

// The busiest SIMD runs ceil(tiles / SIMDs) waves of NA * NB accumulators each: pick the shape that minimises that.
__host__ __device__ inline int conv_pick_shape(int M, int cout_g, int groups, int n_simd, int n_shapes)
{
    int best = -1, best_units = 0;
    for (int s = 0; s < n_shapes; ++s) {
        if (cout_g % (32 * SHAPE_NA[s])) continue;
        const int tiles = conv_wave_tiles(M, cout_g, groups, s);
        const int units = ((tiles + n_simd - 1) / n_simd) * SHAPE_NA[s] * SHAPE_NB[s];
        if (best < 0 || units < best_units) { best = s; best_units = units; }
    }
    return best;
}

__device__ __forceinline__ v4f buf_load4(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff)
{
    return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)soff, 0));
}

template <int KS, int NA, int NB>
__device__ __forceinline__
void conv_wave_tile(const ConvArgs& a, int Mtot, int Mall)
{
    constexpr int P = KS / 2;
    const int HW = a.H * a.W;
    const int CinG = a.Cin / a.groups, CoutG = a.Cout / a.groups;
    const int n_tiles = CoutG / (32 * NA), ny = a.groups * n_tiles;
    const int m_tiles = (Mtot + 32 * NB - 1) / (32 * NB);
#if SVX_CONV_XCD_2D
    // XCD-aware order (workgroup b runs on XCD b % 8, each XCD has its own 4 MB L2): the 8 XCDs are 4 pixel ranges x 2
    // channel halves.  An XCD then needs HALF the layer's weights (0.6-1.75 MB: they stay in its L2 however many rounds of
    // workgroups the launch takes) and, in the grouped layers -- a channel half is a group there -- only that group's half
    // of the input channels: every activation byte is fetched by one XCD, in conv3 (one group) by two.  With 8 pixel ranges
    // x all channel tiles every XCD streamed all the weights and later rounds of workgroups found them evicted
    // (PMC FETCH_SIZE of conv3..5 at 256 images: 5x their operands).
    const int halves = (ny & 1) ? 1 : 2;                                // (an odd number of channel tiles: 8 pixel ranges x all of them)
    const int n_half = ny / halves;                                     // channel tiles of this XCD
    const int mp = (m_tiles * halves + 7) / 8;                          // pixel tiles per pixel range
    const int per_c = (mp * n_half + WAVES - 1) / WAVES;                // compute workgroups per XCD
#else
    const int total_c = m_tiles * ny;                                   // wave tiles, channel tile fastest
    const int wg_c = (total_c + WAVES - 1) / WAVES;
#endif
    const int oct_units = (a.Cout / 8 + FILL_OCT - 1) / FILL_OCT;
    const int fill_units = (a.pixels && a.background) ? ((Mall - Mtot + FILL_PIX - 1) / FILL_PIX) * oct_units : 0;
    // (Background units spread evenly BETWEEN the compute workgroups instead of behind them were measured: slower.  Memory-
    // bound workgroups do not ride along with matrix-bound ones even inside one launch -- on conv2 at 256 images the copy of
    // the 61 % inactive pixels costs 80 us on top of 335, 2.5x its stand-alone time -- which is why the layer with the most
    // inactive pixels, conv2, no longer copies at all: its consumer reads the background itself, svx_bias_relu_pool_lrn.)
    // XCD-aware order (workgroup b runs on XCD b % 8, each XCD has its own 4 MB L2): every XCD gets an equal contiguous
    // run of the compute workgroups -- the activation slice of a pixel tile is fetched into one L2 and re-used by all its
    // channel tiles, whose waves sit in the same workgroup (one L1) -- and, behind it, of the background units
#if SVX_CONV_XCD_2D
    const int per_f = (fill_units + 7) / 8;
#else
    const int per_c = (wg_c + 7) / 8, per_f = (fill_units + 7) / 8;
#endif
    const int local = blockIdx.x >> 3, xcd = blockIdx.x & 7;
    const int tid = threadIdx.x;
    if (local >= per_c) {
        if (local >= per_c + per_f) return;
        const int unit = xcd * per_f + (local - per_c);
        if (unit >= fill_units) return;
        // background unit: FILL_PIX inactive pixels (two lanes per pixel: the halves of its 32-B sector, so that a run of
        // consecutive pixels is one contiguous store) x FILL_OCT octets
        const int mt = unit / oct_units;
        const int o0 = (unit - mt * oct_units) * FILL_OCT;
        const int q = Mtot + mt * FILL_PIX + (tid >> 1);
        if (q < Mall) {
            const int id = a.pixels[q];
            const int bb = id / HW, pp = id - bb * HW;
            const int o1 = min(o0 + FILL_OCT, a.Cout / 8);
            for (int o = o0; o < o1; ++o)
                reinterpret_cast<float4*>(a.out)[(((size_t)bb * (a.Cout / 8) + o) * HW + pp) * 2 + (tid & 1)] =
                    reinterpret_cast<const float4*>(a.background)[((size_t)o * HW + pp) * 2 + (tid & 1)];
        }
        return;
    }
#if SVX_CONV_XCD_2D
    const int wl = local * WAVES + (tid >> 6);                          // wave tile of this XCD, channel tile fastest
    if (wl >= mp * n_half) return;
    const int mt = (xcd / halves) * mp + wl / n_half;
    if (mt >= m_tiles) return;
    const int yy_ = (xcd % halves) * n_half + wl % n_half;
#else
    const int wg = xcd * per_c + local;
    const int wt = wg * WAVES + (tid >> 6);
    if (wg >= wg_c || wt >= total_c) return;
    const int mt = wt / ny;
    const int yy_ = wt - mt * ny;
#endif
    const int g = yy_ / n_tiles;
    const int n0 = (yy_ - g * n_tiles) * 32 * NA;
    const int m0 = mt * 32 * NB;

    const int lane = tid & 63, hi = lane >> 5, lo = lane & 31;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0,
                                            (int)((long long)a.nimg * a.Cin * HW * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0,
                                            (int)((long long)KS * KS * CinG * a.Cout * 4), 0x00020000);

    // B side: the lane's pixel in each of the NB 32-pixel columns of the tile
    int py[NB], px[NB];
    unsigned pbase[NB];                 // byte offset of (image, first octet of the group, pixel 0) + the lane's half
    bool pok[NB];
#pragma unroll
    for (int t = 0; t < NB; ++t) {
        const int mm = m0 + 32 * t + lo;
        pok[t] = mm < Mtot;
        const int pid = !pok[t] ? 0 : (a.pixels ? a.pixels[mm] : mm);
        const int b = pid / HW;
        const int pix = pid - b * HW;
        py[t] = pix / a.W;
        px[t] = pix - py[t] * a.W;
        pbase[t] = (unsigned)((b * (a.Cin / 8) + g * (CinG / 8)) * HW * 32 + 16 * hi);
    }
    // A side: 32 consecutive output channels per accumulator row block, the lane's half of the octet
    unsigned voff_a[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) voff_a[i] = (unsigned)((g * CoutG + n0 + 32 * i + lo) * 32 + 16 * hi);

    v16f acc[NA][NB];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int t = 0; t < NB; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.0f;

    const int octs = CinG / 8;                            // octets (4 k-pairs each) per filter tap
    const int Q = KS * KS * octs;
    const unsigned step_a = (unsigned)(a.Cout * 32), step_b = (unsigned)(HW * 32);

    // state of the LOAD iterator (runs two octets ahead of the MFMAs); all wave-uniform except voff_b
    int lky = 0, lkx = 0, lq = 0;
    unsigned soff_a = 0, soff_b = 0;
    unsigned voff_b[NB];
    auto set_tap = [&]() {
#pragma unroll
        for (int t = 0; t < NB; ++t) {
            const int yy = py[t] + lky - P, xx = px[t] + lkx - P;
            const bool ok = pok[t] & ((unsigned)yy < (unsigned)a.H) & ((unsigned)xx < (unsigned)a.W);      // branch-free
            voff_b[t] = ok ? pbase[t] + (unsigned)((yy * a.W + xx) * 32) : OOB;
        }
    };
    constexpr int R = 3, L = NA + NB, MF = 4 * NA * NB;
    v4f ra[R][NA], rb[R][NB];
    auto load_one = [&](int slot, int q) {
        if (q < NA) ra[slot][q] = buf_load4(rs_w, voff_a[q], soff_a);
        else        rb[slot][q - NA] = buf_load4(rs_x, voff_b[q - NA], soff_b);
    };
    // 3x3 layers -- k order (octet, ky, kx): all taps of an octet before the next octet.  The 9 (25) taps of a pixel tile read the same
    // few KB of its neighbourhood in that octet -- L1 hits -- where the order (ky, kx, octet) streamed the whole 32-octet
    // slice once per tap and had it re-fetched through L2 and, at the launch sizes of the pipeline, through the fabric
    // (PMC FETCH_SIZE 3-5x the operands).  The weights are walked with a stride (tap-major packing kept: the pack is
    // part of the ABI); the per-lane tap offsets are recomputed every stage (a dozen VALU instructions next to 8+ MFMAs:
    // free where several waves share a SIMD -- the pipeline's launches -- and 6-19 % of a 64-image dense launch whose waves
    // sit alone on theirs; per-tile validity bits + a uniform tap offset instead were measured: 5 % slower in list mode).
    // 5x5 layer (conv2, 6 octets per tap) -- k order (ky, kx, octet): its slice per tap is small enough to stay cached, and
    // with 25 taps per octet the per-stage tap arithmetic made the dense launch 11 % slower for no change in traffic.
    constexpr bool TAP_INNER = SVX_CONV_TAP_INNER && KS == 3;
    const unsigned step_tap = (unsigned)octs * step_a;
    auto advance = [&]() {                                // past the end the prefetches read valid memory or zeros: never used
        if (TAP_INNER) {
            soff_a += step_tap;
            if (++lkx == KS) { lkx = 0; ++lky; }
            if (lky == KS) { lky = 0; ++lq; soff_a = (unsigned)lq * step_a; soff_b += step_b; }
            set_tap();
        } else {
            soff_a += step_a;
            soff_b += step_b;
            if (++lq == octs) {
                lq = 0; soff_b = 0;
                if (++lkx == KS) { lkx = 0; ++lky; }
                set_tap();
            }
        }
    };
    // One stage: the 4 * NA * NB MFMAs of the octet in ring slot `cs` with the L loads of a later octet (into slot `ls`)
    // spread between them.  sched_barrier pins the order: left alone, the compiler sinks the loads behind the MFMAs,
    // which collapses the prefetch distance.
    auto stage = [&](int ls, int cs) {
#pragma unroll
        for (int m = 0; m < MF; ++m) {
#pragma unroll
            for (int q = 0; q < L; ++q)
                if (q * MF / L == m) load_one(ls, q);
            const int j = m / (NA * NB), r = m - j * (NA * NB);
            const int i = r / NB, t = r - i * NB;
            acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[cs][i][j], rb[cs][t][j], acc[i][t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        advance();
        __builtin_amdgcn_sched_barrier(0);
    };

    set_tap();
#pragma unroll
    for (int s = 0; s < R - 1; ++s) {
#pragma unroll
        for (int q = 0; q < L; ++q) load_one(s, q);
        advance();
    }
    __builtin_amdgcn_sched_barrier(0);
    // no exits inside the unrolled body (the accumulators would be copied between the exits' register assignments):
    // whole triples first, then the one or two octets a Q that is not a multiple of 3 leaves (their "prefetches" run past
    // the end of the tensors and read zeros)
    const int Q3 = Q - Q % R;
    for (int q = 0; q < Q3; q += R) {
        stage(2, 0);
        stage(0, 1);
        stage(1, 2);
    }
    if (Q - Q3 >= 1) stage(2, 0);
    if (Q - Q3 >= 2) stage(0, 1);

    // epilogue: D[row = channel][col = pixel]; lane holds col = lane & 31, rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5),
    // i.e. for u = r >> 2 the four channels 4 * hi .. + 3 of octet u of the 32-channel block: one 16-B store each
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int c0 = g * CoutG + n0 + 32 * i;               // first channel of the block (a multiple of 32)
        v4f bv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) bv[u] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
        if (a.bias) {
#pragma unroll
            for (int u = 0; u < 4; ++u) bv[u] = *reinterpret_cast<const v4f*>(a.bias + c0 + 8 * u + 4 * hi);
        }
#pragma unroll
        for (int t = 0; t < NB; ++t) {
            const int mm = m0 + 32 * t + lo;
            if (mm >= Mtot) continue;
            const int pid2 = a.pixels ? a.pixels[mm] : mm;
            const int bb = pid2 / HW;
            const int pp = pid2 - bb * HW;
            v4f* o = reinterpret_cast<v4f*>(a.out) + (((size_t)bb * (a.Cout / 8) + c0 / 8) * HW + pp) * 2 + hi;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v4f v = v4f{acc[i][t][4 * u], acc[i][t][4 * u + 1], acc[i][t][4 * u + 2], acc[i][t][4 * u + 3]} + bv[u];
                if (a.relu) { v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f); v.z = fmaxf(v.z, 0.0f); v.w = fmaxf(v.w, 0.0f); }
                o[(size_t)u * HW * 2] = v;
            }
        }
    }
}

// dense mode: the host knows the pixel count and launches the instantiation of the shape it picked
template <int KS, int SHAPE>
__global__ __launch_bounds__(THREADS, 2)
void conv_wave_kernel(const ConvArgs a)
{
    const int Mall = a.nimg * a.H * a.W;
    conv_wave_tile<KS, SHAPE_NA[SHAPE], SHAPE_NB[SHAPE]>(a, Mall, Mall);
}

// list mode: the pixel count lives on the device, so the workgroup picks the shape itself (the grid is sized for the
// shape with the most tiles and for every pixel; surplus workgroups leave at once).  shape >= 0 forces one (experiments).
template <int KS>
__global__ __launch_bounds__(THREADS, 2)
void conv_wave_list_kernel(const ConvArgs a, int shape)
{
    SVX_SHADOW_ROOM();
    const int Mall = a.nimg * a.H * a.W;
    int Mtot = Mall;
    { const long long c = (long long)*a.pixel_count; if (c * 100 < (long long)Mall * SVX_CONV_DENSE_PCT) Mtot = (int)c; }
    if (shape < 0) shape = conv_pick_shape(Mtot, a.Cout / a.groups, a.groups, a.n_simd, LIST_SHAPES);
    if (shape == 0)      conv_wave_tile<KS, SHAPE_NA[0], SHAPE_NB[0]>(a, Mtot, Mall);
#if SVX_LIST_SHAPES > 2                           // experiments only: every shape in one kernel costs its registers (178) in all of them
    else if (shape == 2) conv_wave_tile<KS, SHAPE_NA[2], SHAPE_NB[2]>(a, Mtot, Mall);
    else if (shape == 3) conv_wave_tile<KS, SHAPE_NA[3], SHAPE_NB[3]>(a, Mtot, Mall);
    else if (shape == 4) conv_wave_tile<KS, SHAPE_NA[4], SHAPE_NB[4]>(a, Mtot, Mall);
#endif
    else                 conv_wave_tile<KS, SHAPE_NA[1], SHAPE_NB[1]>(a, Mtot, Mall);
}

template <int KS>
void launch_conv(int shape, int wgs, hipStream_t st, const ConvArgs& a)
{
    switch (shape) {
#define SVX_CASE(S_) case S_: hipLaunchKernelGGL((conv_wave_kernel<KS, S_>), dim3((unsigned)wgs), dim3(THREADS), 0, st, a); break
    SVX_CASE(0); SVX_CASE(1); SVX_CASE(2); SVX_CASE(3);
    default: hipLaunchKernelGGL((conv_wave_kernel<KS, 4>), dim3((unsigned)wgs), dim3(THREADS), 0, st, a); break;
#undef SVX_CASE
    }
}

int device_simds()
{
    static int n = 0;
    if (n == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
        n = 4 * cus;
    }
    return n;
}

}  // namespace

extern "C" int svx_conv2d_same(const float* d_in, const float* d_w_packed, const float* d_bias, float* d_out, uint32_t n,
                               uint32_t cin, uint32_t cout, uint32_t height, uint32_t width, uint32_t ksize,
                               uint32_t groups, int relu, const int32_t* d_pixels, const uint32_t* d_pixel_count,
                               const float* d_background, void* stream)
{
    if (n == 0) return SVX_OK;
    if (!d_in || !d_w_packed || !d_out || groups == 0 || cin % groups || cout % groups) return SVX_EINVAL;
    if ((d_pixels == nullptr) != (d_pixel_count == nullptr) || (d_background && !d_pixels)) return SVX_EINVAL;
    const uint32_t cin_g = cin / groups, cout_g = cout / groups;
    if (cin_g % 16 || cout_g % 64 || (ksize != 3 && ksize != 5)) return SVX_EINVAL;
    for (const void* p : {(const void*)d_in, (const void*)d_w_packed, (const void*)d_out, (const void*)d_bias, (const void*)d_background})
        if (reinterpret_cast<uintptr_t>(p) & 15u) return SVX_EINVAL;                   // 16-byte operand loads and stores
    // 31-bit byte offsets into the input and the weights (buffer descriptors; bit 31 marks "outside"), 31-bit pixel ids
    if ((uint64_t)n * cin * height * width * 4 > 0x7fffffffull || (uint64_t)ksize * ksize * cin_g * cout * 4 > 0x7fffffffull) return SVX_EINVAL;
    const int mall = (int)(n * height * width);
    hipStream_t st = static_cast<hipStream_t>(stream);
    ConvArgs a{d_in, d_w_packed, d_bias, d_out, (int)n, (int)cin, (int)cout, (int)height, (int)width, (int)groups, relu,
               d_pixels, d_pixel_count, d_background, device_simds()};
    int shape = d_pixels ? -1 : conv_pick_shape(mall, (int)cout_g, (int)groups, a.n_simd, N_SHAPES);
#ifdef SVX_CONV_EXPERIMENT
    if (const char* e = getenv("SVX_CONV_SHAPE")) if (atoi(e) >= 0 && atoi(e) < (d_pixels ? LIST_SHAPES : N_SHAPES)) shape = atoi(e);
#endif
    if (d_pixels) {
        int wgs = 0;
        for (int s = 0; s < LIST_SHAPES; ++s) { const int t = conv_compute_wgs(mall, (int)cout_g, (int)groups, s); if (t > wgs) wgs = t; }
        if (d_background) wgs += 8 * ((((mall + FILL_PIX - 1) / FILL_PIX) * (int)((cout / 8 + FILL_OCT - 1) / FILL_OCT) + 7) / 8 + 1);
        if (ksize == 3) hipLaunchKernelGGL(conv_wave_list_kernel<3>, dim3((unsigned)wgs), dim3(THREADS), 0, st, a, shape);
        else            hipLaunchKernelGGL(conv_wave_list_kernel<5>, dim3((unsigned)wgs), dim3(THREADS), 0, st, a, shape);
    } else {
        const int wgs = conv_compute_wgs(mall, (int)cout_g, (int)groups, shape);
        if (ksize == 3) launch_conv<3>(shape, wgs, st, a);
        else            launch_conv<5>(shape, wgs, st, a);
    }
    return hipGetLastError() == hipSuccess ? SVX_OK : SVX_ELAUNCH;
}
