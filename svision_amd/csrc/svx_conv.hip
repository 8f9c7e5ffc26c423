// svx_conv.hip -- fp32 implicit-GEMM convolution on the gfx950 matrix cores (MI355X).
//
// The dense contractions of the reference CNN after the first layer: tf.nn.conv2d, stride 1, SAME
// padding, optional 2-way channel groups (src/network/alexnet.py:34,39,42,45 via :109-129), with the
// bias + ReLU of :132-135 fused in the epilogue.  NCHW activations, weights in the checkpoint's own
// HWIO layout [kh][kw][Cin/groups][Cout] (the group split is a slice of the last axis, exactly as
// tf.split(axis=3) does it).
//
// GEMM view per group:   D[n][m] = sum_k  Wt[n][k] * X[k][m]
//   n = output channel in the group, m = (image, y, x) output pixel, k = (ky, kx, c).
// Rows of D are channels and columns are pixels so that each MFMA accumulator register holds 32
// consecutive pixels of one channel plane: coalesced 128-B stores into NCHW.
//
// Design (round 2): one WAVE owns one output tile of 32*NA channels x 32*NB pixels (NA*NB accumulators of
// v_mfma_f32_32x32x2_f32) and feeds itself: there is no LDS staging, no barrier and no inter-wave dependency.
// The fp32 MFMA runs at the vector rate (64 cycles per instruction per SIMD), so a 64 x 64 wave tile needs one
// 4-byte operand load per lane per MFMA -- 16 B/clk per CU out of the L1/L2 -- which is the same L2 traffic per
// FLOP the LDS-staged 64 x 128 workgroup tile of round 1 had, without its costs: the slice barrier every 16 MFMAs,
// the LDS round trip, the select per staged value and, above all, the tile-count quantisation (510 workgroups of
// 4 lock-stepped waves on 512 slots; 64 x 64 fall-back tiles with one accumulator per wave for conv5 and for every
// active-set launch).  Here a launch is a flat list of wave tiles, 2028 of 64 x 32 for a dense 13 x 13 layer (two waves
// per SIMD, 99 % of the slots filled), and the shape is chosen per launch by a cost model fitted on the chip (below).
//   * MFMA A fragment = weights: lane l reads W[k0 + (l >> 5)][n0 + (l & 31)] -- two 128-B rows of the HWIO tensor;
//     B fragment = activations: lane l reads X[c0 + (l >> 5)][pixel (l & 31) of the tile, shifted by the tap].
//   * Loads are buffer loads (wave-uniform descriptor + per-lane byte offset + uniform SGPR offset): the k loop
//     advances the two SGPR offsets only -- no per-load VALU -- and a tap that falls outside the image, or a
//     column past the end of the pixel list, gets a per-lane offset beyond the descriptor's range, which the
//     hardware answers with 0.0: SAME padding costs nothing in the loop and nothing is selected afterwards.
//   * Fragments are fetched R - 1 k-pairs (>= 900 cycles of MFMA work) ahead into a ring of R register sets,
//     statically indexed by unrolling R k-pairs; R divides the k-pairs of a filter tap, so the per-lane offsets
//     change (once per tap) only between two unrolled blocks.
//   * k order per output element is (ky, kx, c) ascending in every shape and mode: the active-set path stays
//     bit-identical to the dense path.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/svx.h"

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int THREADS = 256, WAVES = THREADS / 64;
constexpr int FILL_PIX = 64, FILL_CH = 64;        // background copy unit of one workgroup
constexpr unsigned OOB = 0x80000000u;             // per-lane byte offset no descriptor covers (tensors are < 2 GB)
#ifndef SVX_CONV_DENSE_PCT
#define SVX_CONV_DENSE_PCT 97             // a pixel list this full (percent) is not worth following: every pixel is computed
#endif

// wave tile shapes (NA x 32 channels, NB x 32 pixels), in order of preference at equal cost.  Measured on MI355X
// (tools/ab_conv.py, tools/stage_bench.py): weight rows are cheaper to fetch than activation columns (aligned 128-B rows
// vs shifted / gathered pixels), two small waves per SIMD cover each other's stalls better than one big wave covers its
// own, and 64 x 32 (two accumulators, 72 VGPRs) is the best or within 2 % of the best shape for every layer once several
// launches overlap (graph replays on 3-4 streams); 32 x 96 and 64 x 96 only win a launch running alone whose tile count
// they happen to quantise better (conv5 dense: 904 tiles on 1024 SIMDs; conv2 dense).
constexpr int N_SHAPES = 5;
constexpr int SHAPE_NA[N_SHAPES] = {2, 1, 2, 2, 1};
constexpr int SHAPE_NB[N_SHAPES] = {1, 3, 3, 2, 2};
constexpr int LIST_SHAPE = 0;             // list mode: the pixel count is on the device; 64 x 32 whatever it is

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

// The busiest SIMD runs ceil(tiles / SIMDs) waves of NA * NB accumulators each: pick the shape that minimises that.
__host__ __device__ inline int conv_pick_shape(int M, int cout_g, int groups, int n_simd)
{
    int best = -1, best_units = 0;
    for (int s = 0; s < N_SHAPES; ++s) {
        if (cout_g % (32 * SHAPE_NA[s])) continue;
        const int tiles = conv_wave_tiles(M, cout_g, groups, s);
        const int units = ((tiles + n_simd - 1) / n_simd) * SHAPE_NA[s] * SHAPE_NB[s];
        if (best < 0 || units < best_units) { best = s; best_units = units; }
    }
    return best;
}

__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)voff, (int)soff, 0));
}

template <int KS, int NA, int NB>
__device__ __forceinline__
void conv_wave_tile(const ConvArgs& a, int Mtot, int Mall)
{
    constexpr int P = KS / 2;
    constexpr int R = (NA * NB <= 3) ? 8 : 4;           // ring of fragment register sets = k-pairs per unrolled block
    const int HW = a.H * a.W;
    const int CinG = a.Cin / a.groups, CoutG = a.Cout / a.groups;
    const int n_tiles = CoutG / (32 * NA), ny = a.groups * n_tiles;
    const int m_tiles = (Mtot + 32 * NB - 1) / (32 * NB);
    const int total_c = m_tiles * ny;                                   // wave tiles, channel tile fastest
    const int wg_c = (total_c + WAVES - 1) / WAVES;
    const int fill_units = (a.pixels && a.background) ? ((Mall - Mtot + FILL_PIX - 1) / FILL_PIX) * (a.Cout / FILL_CH) : 0;
    // XCD-aware order (workgroup b runs on XCD b % 8, each XCD has its own 4 MB L2): every XCD gets an equal contiguous
    // run of the compute workgroups -- the activation slice of a pixel tile is fetched into one L2 and re-used by all its
    // channel tiles, whose waves sit in the same workgroup (one L1) -- and, behind it, of the background units
    const int per_c = (wg_c + 7) / 8, per_f = (fill_units + 7) / 8;
    const int local = blockIdx.x >> 3, xcd = blockIdx.x & 7;
    const int tid = threadIdx.x;
    if (local >= per_c) {
        if (local >= per_c + per_f) return;
        const int unit = xcd * per_f + (local - per_c);
        if (unit >= fill_units) return;
        // background unit: FILL_PIX inactive pixels x FILL_CH channels, lanes along the (ascending) pixel list
        const int cu = a.Cout / FILL_CH;
        const int mt = unit / cu;
        const int c0 = (unit - mt * cu) * FILL_CH;
        const int q = Mtot + mt * FILL_PIX + (tid & (FILL_PIX - 1));
        if (q < Mall) {
            const int id = a.pixels[q];
            const int bb = id / HW, pp = id - bb * HW;
            for (int c = tid / FILL_PIX; c < FILL_CH; c += THREADS / FILL_PIX)
                a.out[((size_t)bb * a.Cout + c0 + c) * HW + pp] = a.background[(size_t)(c0 + c) * HW + pp];
        }
        return;
    }
    const int wg = xcd * per_c + local;
    const int wt = wg * WAVES + (tid >> 6);
    if (wg >= wg_c || wt >= total_c) return;
    const int mt = wt / ny;
    const int yy_ = wt - mt * ny;
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
    unsigned pbase[NB];                 // byte offset of (image, first channel of the group + hi, pixel 0)
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
        pbase[t] = (unsigned)(((b * a.Cin + g * CinG + hi) * HW) * 4);
    }
    // A side: row hi of the k-pair, 32 consecutive output channels per accumulator row block
    unsigned voff_a[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) voff_a[i] = (unsigned)((hi * a.Cout + g * CoutG + n0 + 32 * i + lo) * 4);

    v16f acc[NA][NB];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int t = 0; t < NB; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.0f;

    const int cpairs = CinG / 2;                          // k-pairs per filter tap (a multiple of 8)
    const int nblk = KS * KS * (cpairs / R);
    const unsigned step_a = (unsigned)(2 * a.Cout * 4), step_b = (unsigned)(2 * HW * 4);

    // state of the LOAD iterator (runs R - 1 k-pairs ahead of the MFMAs); all wave-uniform except voff_b
    int lky = 0, lkx = 0, lcp = 0;
    unsigned soff_a = 0, soff_b = 0;
    unsigned voff_b[NB];
    auto set_tap = [&]() {
#pragma unroll
        for (int t = 0; t < NB; ++t) {
            const int yy = py[t] + lky - P, xx = px[t] + lkx - P;
            const bool ok = pok[t] & ((unsigned)yy < (unsigned)a.H) & ((unsigned)xx < (unsigned)a.W);      // branch-free
            voff_b[t] = ok ? pbase[t] + (unsigned)((yy * a.W + xx) * 4) : OOB;
        }
    };
    float ra[R][NA], rb[R][NB];
    auto load_one = [&](int slot, int q) {
#if defined(SVX_ABL) && SVX_ABL == 2        /* ablation: no loads in the loop (MFMAs on whatever the registers hold) */
        if (soff_a > 64u * step_a) return;
#endif
        if (q < NA) ra[slot][q] = buf_load(rs_w, voff_a[q], soff_a);
#if defined(SVX_ABL) && SVX_ABL == 1        /* ablation: activation loads replaced by aligned 2 x 128-B row loads (wrong results) */
        else        rb[slot][q - NA] = buf_load(rs_w, voff_a[0] + 128u * (q - NA + 1), soff_a);
#else
        else        rb[slot][q - NA] = buf_load(rs_x, voff_b[q - NA], soff_b);
#endif
    };
    auto load_stage = [&](int slot) {
#pragma unroll
        for (int q = 0; q < NA + NB; ++q) load_one(slot, q);
        soff_a += step_a;
        soff_b += step_b;
    };
    auto next_block = [&]() {                             // the load iterator enters the next block of R k-pairs
        lcp += R;
        if (lcp == cpairs) {
            lcp = 0; soff_b = 0;
            if (++lkx == KS) { lkx = 0; ++lky; }
            set_tap();
        }
    };
    // One stage: the MFMAs of k-pair `cs` with the loads of a later k-pair (into ring slot `ls`) spread between them --
    // an MFMA occupies the pipe for 64 cycles and a wave issues in order, so a cluster of loads between two MFMAs that
    // takes longer than that to issue leaves the pipe idle; one or two loads per gap do not.  sched_barrier pins the
    // order (left alone, the compiler sinks every load of a block behind the block's MFMAs, which collapses the
    // prefetch distance to one stage).
    auto stage = [&](int ls, int cs, bool with_loads) {
        constexpr int L = NA + NB, MF = NA * NB;
#pragma unroll
        for (int j = 0; j < MF; ++j) {
            if (with_loads) {
#pragma unroll
                for (int q = j * L / MF; q < (j + 1) * L / MF; ++q) load_one(ls, q);
            }
            const int i = j / NB, t = j - i * NB;
            acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[cs][i], rb[cs][t], acc[i][t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (with_loads) { soff_a += step_a; soff_b += step_b; }
    };

    set_tap();
#pragma unroll
    for (int s = 0; s < R - 1; ++s) load_stage(s);
    __builtin_amdgcn_sched_barrier(0);
    for (int blk = 0; blk + 1 < nblk; ++blk) {
        stage(R - 1, 0, true);
        next_block();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 1; s < R; ++s) stage(s - 1, s, true);
    }
    stage(R - 1, 0, true);
#pragma unroll
    for (int s = 1; s < R; ++s) stage(0, s, false);

    // epilogue: D[row = channel][col = pixel]; lane holds col = lane & 31, rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        float bv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) bv[r] = 0.0f;
        if (a.bias) {                                          // one batch of loads, not one round trip per row
            const float* bp = a.bias + g * CoutG + n0 + 32 * i + 4 * hi;
#pragma unroll
            for (int r = 0; r < 16; ++r) bv[r] = bp[(r & 3) + 8 * (r >> 2)];
        }
#pragma unroll
        for (int t = 0; t < NB; ++t) {
            const int mm = m0 + 32 * t + lo;
            if (mm >= Mtot) continue;
            const int pid2 = a.pixels ? a.pixels[mm] : mm;
            const int bb = pid2 / HW;
            const int pp = pid2 - bb * HW;
            float* o = a.out + ((size_t)bb * a.Cout + (size_t)g * CoutG + n0 + 32 * i + 4 * hi) * HW + pp;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nl = (r & 3) + 8 * (r >> 2);
                float v = acc[i][t][r] + bv[r];
                if (a.relu) v = fmaxf(v, 0.0f);
                o[(size_t)nl * HW] = v;
            }
        }
    }
}

// dense mode: the host knows the pixel count and picks the shape; list mode: the count lives on the device
// (the grid is sized for every pixel; surplus workgroups leave at once)
template <int KS, int SHAPE>
__global__ __launch_bounds__(THREADS, 2)
void conv_wave_kernel(const ConvArgs a)
{
    const int Mall = a.nimg * a.H * a.W;
    int Mtot = Mall;
    if (a.pixels) { const long long c = (long long)*a.pixel_count; if (c * 100 < (long long)Mall * SVX_CONV_DENSE_PCT) Mtot = (int)c; }
    conv_wave_tile<KS, SHAPE_NA[SHAPE], SHAPE_NB[SHAPE]>(a, Mtot, Mall);
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

extern "C" int svx_conv2d_same(const float* d_in, const float* d_w_hwio, const float* d_bias, float* d_out, uint32_t n,
                               uint32_t cin, uint32_t cout, uint32_t height, uint32_t width, uint32_t ksize,
                               uint32_t groups, int relu, const int32_t* d_pixels, const uint32_t* d_pixel_count,
                               const float* d_background, void* stream)
{
    if (n == 0) return SVX_OK;
    if (!d_in || !d_w_hwio || !d_out || groups == 0 || cin % groups || cout % groups) return SVX_EINVAL;
    if ((d_pixels == nullptr) != (d_pixel_count == nullptr) || (d_background && !d_pixels)) return SVX_EINVAL;
    const uint32_t cin_g = cin / groups, cout_g = cout / groups;
    if (cin_g % 16 || cout_g % 64 || (ksize != 3 && ksize != 5)) return SVX_EINVAL;
    // 31-bit byte offsets into the input and the weights (buffer descriptors; bit 31 marks "outside"), 31-bit pixel ids
    if ((uint64_t)n * cin * height * width * 4 > 0x7fffffffull || (uint64_t)ksize * ksize * cin_g * cout * 4 > 0x7fffffffull) return SVX_EINVAL;
    const int mall = (int)(n * height * width);
    hipStream_t st = static_cast<hipStream_t>(stream);
    ConvArgs a{d_in, d_w_hwio, d_bias, d_out, (int)n, (int)cin, (int)cout, (int)height, (int)width, (int)groups, relu,
               d_pixels, d_pixel_count, d_background, device_simds()};
    int shape = d_pixels ? LIST_SHAPE : conv_pick_shape(mall, (int)cout_g, (int)groups, a.n_simd);
#ifdef SVX_CONV_EXPERIMENT
    if (const char* e = getenv("SVX_CONV_SHAPE")) if (atoi(e) >= 0 && atoi(e) < N_SHAPES) shape = atoi(e);
#endif
    int wgs = 8 * (((conv_wave_tiles(mall, (int)cout_g, (int)groups, shape) + WAVES - 1) / WAVES + 7) / 8);
    if (d_pixels && d_background) wgs += 8 * ((((mall + FILL_PIX - 1) / FILL_PIX) * (int)(cout / FILL_CH) + 7) / 8 + 1);
    if (ksize == 3) launch_conv<3>(shape, wgs, st, a);
    else            launch_conv<5>(shape, wgs, st, a);
    return hipGetLastError() == hipSuccess ? SVX_OK : SVX_ELAUNCH;
}
