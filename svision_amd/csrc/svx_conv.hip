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
// Tiling: 256 threads = 4 waves compute a 64 (n) x 128 (m) tile; each wave owns 32 x 64 (two
// v_mfma_f32_32x32x2_f32 accumulators, 32 VGPRs).  K is walked in slices of 16 that never straddle a
// filter tap (Cin/groups is a multiple of 16), so the padding test is one predicate per slice and the
// eight activation loads of a lane are a constant stride apart.  Both operands are staged in LDS as
// [k][n] / [k][m] (unit stride across lanes: no bank conflicts on write or on the ds_read_b32 fragment
// reads), double buffered; within a slice the wave's own LDS reads / stores and the global loads of slice
// k+1 are interleaved with its 16 MFMAs (software pipelined, one barrier per slice).  fp32 MFMA issues
// every 64 cycles, so 24 KB of LDS traffic per slice is far below the LDS rate; the kernel is
// matrix-pipe bound.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/svx.h"

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int BN = 64, THREADS = 256;
#ifndef SVX_CONV_DENSE_PCT
#define SVX_CONV_DENSE_PCT 97             // a pixel list this full (percent) is not worth following: every pixel is computed
#endif
#ifndef SVX_CONV_BM64_BELOW
#define SVX_CONV_BM64_BELOW 448          // 64 x 128 tiles below this many of them: use 64 x 64
#endif

constexpr int lds_floats(int bk, int bm) { return 2 * bk * (BN + bm); }

template <int KS, int BK, int BM>
__device__ __forceinline__
void conv_igemm_tile(const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias,
                     float* __restrict__ out, int nimg, int Cin, int Cout, int H, int W, int groups, int relu,
                     const int32_t* __restrict__ pixels, const uint32_t* __restrict__ pixel_count,
                     const float* __restrict__ background, float* lds)
{
    constexpr int P = KS / 2;
    float (*Ws)[BK][BN] = reinterpret_cast<float (*)[BK][BN]>(lds);                    // [2][BK][BN]
    float (*Xs)[BK][BM] = reinterpret_cast<float (*)[BK][BM]>(lds + 2 * BK * BN);      // [2][BK][BM]

    const int CinG = Cin / groups, CoutG = Cout / groups;
    const int n_tiles = CoutG / BN;
    const int HW = H * W;
    // columns of the GEMM: every output pixel, or only the first *pixel_count entries of the pixel permutation (the
    // active set; the grid is sized for all pixels).  The pixels behind them receive the background (image independent
    // response to an empty image) from the workgroups the active tiles leave over.  When nearly all pixels are active
    // (97 %) everything is computed; below that the list still pays (measured with 81 % and 91 % active).
    const long long Mall = (long long)nimg * HW;
    long long Mtot = Mall;
    if (pixels) { const long long a = (long long)*pixel_count; if (a * 100 < Mall * SVX_CONV_DENSE_PCT) Mtot = a; }
    // XCD-aware tile order (workgroup b runs on XCD b % 8, each XCD has its own 4 MB L2): the (pixel tile,
    // channel tile) pairs, channel tile fastest, are cut into 8 equal contiguous runs, one per XCD, so the
    // activation slice of a pixel tile is fetched into one L2 once and re-used by all its (group, n-tile)
    // pairs -- measured 6-8x less L2-miss traffic -- without adding a dispatch round (equal run lengths)
    const int ny = groups * n_tiles;
    const long long m_tiles = (Mtot + BM - 1) / BM;
    const long long fill_tiles = (pixels && background) ? (Mall - Mtot + BM - 1) / BM : 0;
    // every XCD gets an equal contiguous run of the compute pairs and, behind it, of the background pairs
    const long long total_c = m_tiles * ny, total_f = fill_tiles * ny;
    const long long per_c = (total_c + 7) / 8, per_f = (total_f + 7) / 8;
    const long long local = blockIdx.x >> 3, xcd = blockIdx.x & 7;
    long long pair;
    bool fill = false;
    if (local < per_c) { pair = xcd * per_c + local; if (pair >= total_c) return; }
    else if (local < per_c + per_f) { pair = xcd * per_f + (local - per_c); if (pair >= total_f) return; fill = true; }
    else return;
    const int mt = (int)(pair / ny), yy_ = (int)(pair - (long long)mt * ny);
    const int g = yy_ / n_tiles;
    const int n0 = (yy_ - g * n_tiles) * BN;
    if (fill) {
        // background tile: BM inactive pixels x BN channels, lanes along the (ascending) pixel list
        const long long q = Mtot + (long long)mt * BM + (threadIdx.x & (BM - 1));
        if (q < Mall) {
            const int id = pixels[q];
            const int bb = id / HW, pp = id - bb * HW;
            const int c0 = g * CoutG + n0;
            for (int c = threadIdx.x / BM; c < BN; c += THREADS / BM)
                out[((size_t)bb * Cout + c0 + c) * HW + pp] = background[(size_t)(c0 + c) * HW + pp];
        }
        return;
    }
    const int m0 = mt * BM;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave & 1, wm = wave >> 1;
    constexpr int MT = BM / 64;                    // 32-pixel MFMA tiles per wave (2 waves along m)
    constexpr int WM = BM / 2;                     // pixels per wave

    // activation loader: one output pixel (column m) and k rows xk0, xk0+2, ..., xk0+14 per lane.  Addresses are
    // (uniform row base in SGPRs) + (one 32-bit lane offset): no per-load address arithmetic.
    constexpr int XSTEP = THREADS / BM;            // k rows between two loads of a lane (2 or 4)
    const int xm = tid & (BM - 1), xk0 = tid / BM;
    const long long m = (long long)m0 + xm;
    const bool m_ok = m < Mtot;
    const long long pid = !m_ok ? 0 : (pixels ? (long long)pixels[m] : m);
    const int b = (int)(pid / HW);
    const int pix = (int)(pid - (long long)b * HW);
    const int y = pix / W, x = pix - y * W;
    const unsigned tbase = (unsigned)((b * Cin + g * CinG + xk0) * HW);
    constexpr int XR = BK / XSTEP, WR = BK / 16;   // activation rows / weight float4s per lane and slice
    const char* rowp[XR];
#pragma unroll
    for (int i = 0; i < XR; ++i) rowp[i] = reinterpret_cast<const char*>(in + (size_t)(XSTEP * i) * HW);
    // weight loader: k row wk, four consecutive output channels
    const int wk = tid >> 4, wn4 = (tid & 15) * 4;
    const unsigned wconst = (unsigned)(wk * Cout + g * CoutG + n0 + wn4);

    const int cblocks = CinG / BK;
    const int nk = KS * KS * cblocks;

    v16f acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;

    float xr[XR];
    float4 wr[WR];
    bool xr_ok = false;
    int ky = 0, kx = 0, cb = 0;           // decomposition of the slice being LOADED (all wave-uniform)

    // branch-free: out-of-image taps load from a clamped (valid) address and are zeroed by a select at LDS-store
    // time, so the loads can be issued ahead of the matrix instructions and nothing waits on them before those
    auto load_slice = [&]() {
        const int yy = y + ky - P, xx = x + kx - P;
        xr_ok = m_ok && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
        const int yc = min(max(yy, 0), H - 1), xc = min(max(xx, 0), W - 1);
        const unsigned ob = (tbase + (unsigned)(yc * W + xc) + (unsigned)(cb * BK * HW)) * 4u;   // byte offset < 4 GB
#pragma unroll
        for (int i = 0; i < XR; ++i) xr[i] = *reinterpret_cast<const float*>(rowp[i] + ob);
        const char* wrow = reinterpret_cast<const char*>(w + (size_t)((ky * KS + kx) * CinG + cb * BK) * Cout);   // uniform
#pragma unroll
        for (int j = 0; j < WR; ++j) wr[j] = *reinterpret_cast<const float4*>(wrow + (wconst + (unsigned)(16 * j * Cout)) * 4u);
        if (++cb == cblocks) { cb = 0; if (++kx == KS) { kx = 0; ++ky; } }
    };
    auto store_slice = [&](int buf) {
#pragma unroll
        for (int i = 0; i < XR; ++i) Xs[buf][xk0 + XSTEP * i][xm] = xr_ok ? xr[i] : 0.0f;
#pragma unroll
        for (int j = 0; j < WR; ++j) *reinterpret_cast<float4*>(&Ws[buf][wk + 16 * j][wn4]) = wr[j];
    };

    load_slice();
    store_slice(0);
    __syncthreads();

    const int fk = lane >> 5, fj = lane & 31;
    // One slice: the non-matrix instructions sit between the MFMAs of the wave itself -- fragment reads one
    // k-pair ahead in a ring of register sets, the next slice's global loads (no per-load address arithmetic) up front,
    // its LDS stores under the last four k-pairs -- so a wave keeps the matrix pipe fed without relying on the other
    // resident wave being in its matrix phase (sched_barrier pins the order); one barrier per slice.
    auto read_frag = [&](int buf, int kk, float& a, float (&b)[MT]) {
        a = Ws[buf][2 * kk + fk][wn * 32 + fj];
#pragma unroll
        for (int t = 0; t < MT; ++t) b[t] = Xs[buf][2 * kk + fk][wm * WM + t * 32 + fj];
    };
#ifndef SVX_CONV_AHEAD
#define SVX_CONV_AHEAD 1                     // k-pairs the fragment reads run ahead of the MFMAs (1, 2, 3 measured: 735 / 755 / 760 us per batch)
#define SVX_CONV_S0 4                        // first k-pair that carries LDS stores of the next slice
#endif
    auto compute_slice = [&](int buf, bool prefetch) {
        constexpr int NKK = BK / 2, AHEAD = SVX_CONV_AHEAD, RING = AHEAD + 1, S0 = SVX_CONV_S0, SPAN = NKK - S0;
        float ra[RING], rb[RING][MT];
#pragma unroll
        for (int kk = 0; kk < AHEAD; ++kk) read_frag(buf, kk, ra[kk], rb[kk]);
        if (prefetch) load_slice();
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            if (kk + AHEAD < NKK) read_frag(buf, kk + AHEAD, ra[(kk + AHEAD) % RING], rb[(kk + AHEAD) % RING]);
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[kk % RING], rb[kk % RING][t], acc[t], 0, 0, 0);
            if (prefetch && kk >= S0) {
                // a share of the next slice's LDS stores per k-pair
                const int q = kk - S0;
#pragma unroll
                for (int i = q * XR / SPAN; i < (q + 1) * XR / SPAN; ++i) Xs[buf ^ 1][xk0 + XSTEP * i][xm] = xr_ok ? xr[i] : 0.0f;
                if (q < WR) *reinterpret_cast<float4*>(&Ws[buf ^ 1][wk + 16 * q][wn4]) = wr[q];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int kt = 0; kt + 1 < nk; ++kt) {
        compute_slice(kt & 1, true);
        __syncthreads();
    }
    compute_slice((nk - 1) & 1, false);

    // epilogue: D[row = channel][col = pixel]; lane holds col = lane & 31, rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    float bv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bv[r] = 0.0f;
    if (bias) {                                            // one batch of loads, not one round trip per row
        const float* bp = bias + g * CoutG + n0 + wn * 32 + 4 * fk;
#pragma unroll
        for (int r = 0; r < 16; ++r) bv[r] = bp[(r & 3) + 8 * (r >> 2)];
    }
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const long long mm = (long long)m0 + wm * WM + t * 32 + fj;
        if (mm >= Mtot) continue;
        const long long pid2 = pixels ? (long long)pixels[mm] : mm;
        const int bb = (int)(pid2 / HW);
        const int pp = (int)(pid2 - (long long)bb * HW);
        float* o = out + ((size_t)bb * Cout + (size_t)g * CoutG + n0 + wn * 32 + 4 * fk) * HW + pp;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int nl = (r & 3) + 8 * (r >> 2);
            float v = acc[t][r] + bv[r];
            if (relu) v = fmaxf(v, 0.0f);
            o[(size_t)nl * HW] = v;
        }
    }
}

// dense mode: the host picks the tile shape
template <int KS, int BK, int BM>
__global__ __launch_bounds__(THREADS, 2)
void conv_igemm_kernel(const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias,
                       float* __restrict__ out, int nimg, int Cin, int Cout, int H, int W, int groups, int relu)
{
    __shared__ float lds[lds_floats(BK, BM)];
    conv_igemm_tile<KS, BK, BM>(in, w, bias, out, nimg, Cin, Cout, H, W, groups, relu, nullptr, nullptr, nullptr, lds);
}

// list mode: the host does not know the list's length, so the workgroup picks the tile shape itself -- 64 x 128 unless
// that would leave the 256 CUs under two resident workgroups each (the grid is sized for the smaller tiles)
template <int KS, int BK>
__global__ __launch_bounds__(THREADS, 2)
void conv_igemm_list_kernel(const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias,
                            float* __restrict__ out, int nimg, int Cin, int Cout, int H, int W, int groups, int relu,
                            const int32_t* __restrict__ pixels, const uint32_t* __restrict__ pixel_count,
                            const float* __restrict__ background)
{
    __shared__ float lds[lds_floats(BK, 128)];
    const long long Mall = (long long)nimg * H * W;
    long long Mtot = (long long)*pixel_count;
    if (Mtot * 100 >= Mall * SVX_CONV_DENSE_PCT) Mtot = Mall;
    const long long tiles128 = ((Mtot + 127) / 128) * groups * ((Cout / groups) / BN);
    if (tiles128 < SVX_CONV_BM64_BELOW)
        conv_igemm_tile<KS, BK, 64>(in, w, bias, out, nimg, Cin, Cout, H, W, groups, relu, pixels, pixel_count, background, lds);
    else
        conv_igemm_tile<KS, BK, 128>(in, w, bias, out, nimg, Cin, Cout, H, W, groups, relu, pixels, pixel_count, background, lds);
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
    if (cin_g % 16 || cout_g % BN || (ksize != 3 && ksize != 5)) return SVX_EINVAL;
    if ((reinterpret_cast<uintptr_t>(d_w_hwio) & 15u) || (cout % 4)) return SVX_EINVAL;
    // 32-bit byte offsets into the input, 32-bit pixel ids
    if ((uint64_t)n * cin * height * width * 4 > 0xffffffffull || (uint64_t)n * height * width > 0x7fffffffull) return SVX_EINVAL;
    const long long mtot = (long long)n * height * width;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // 64 x 128 tiles unless they leave the 256 CUs under two resident workgroups each (then 64 x 64: the partial last
    // dispatch round of the big tiles costs more than the extra fragment reads of the small ones)
    const long long tiles128 = ((mtot + 127) / 128) * groups * (cout_g / BN);
    const int bm = tiles128 < SVX_CONV_BM64_BELOW ? 64 : 128;
    if (d_pixels) {
        const long long tt = ((mtot + 63) / 64 + 1) * groups * (cout_g / BN) + 16;     /* active and background tiles both round up, per XCD too */
        const dim3 grid((unsigned)(8 * ((tt + 7) / 8)));
        if (ksize == 3)
            hipLaunchKernelGGL((conv_igemm_list_kernel<3, 16>), grid, dim3(THREADS), 0, st, d_in, d_w_hwio, d_bias, d_out, (int)n, (int)cin,
                               (int)cout, (int)height, (int)width, (int)groups, relu, d_pixels, d_pixel_count, d_background);
        else
            hipLaunchKernelGGL((conv_igemm_list_kernel<5, 16>), grid, dim3(THREADS), 0, st, d_in, d_w_hwio, d_bias, d_out, (int)n, (int)cin,
                               (int)cout, (int)height, (int)width, (int)groups, relu, d_pixels, d_pixel_count, d_background);
        return hipGetLastError() == hipSuccess ? SVX_OK : SVX_ELAUNCH;
    }
#define SVX_LAUNCH_CONV(KS_, BM_) do { \
        const long long tt = ((mtot + BM_ - 1) / BM_) * groups * (cout_g / BN); \
        hipLaunchKernelGGL((conv_igemm_kernel<KS_, 16, BM_>), dim3((unsigned)(8 * ((tt + 7) / 8))), dim3(THREADS), 0, st, d_in, d_w_hwio, \
            d_bias, d_out, (int)n, (int)cin, (int)cout, (int)height, (int)width, (int)groups, relu); } while (0)
    if (ksize == 3) { if (bm == 64) SVX_LAUNCH_CONV(3, 64); else SVX_LAUNCH_CONV(3, 128); }
    else            { if (bm == 64) SVX_LAUNCH_CONV(5, 64); else SVX_LAUNCH_CONV(5, 128); }
#undef SVX_LAUNCH_CONV
    return hipGetLastError() == hipSuccess ? SVX_OK : SVX_ELAUNCH;
}
