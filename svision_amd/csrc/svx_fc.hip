// svx_fc.hip -- fully connected layers fc6 / fc7 on the gfx950 fp32 matrix cores (MI355X).
//
// Reference: tf.nn.xw_plus_b + relu (src/network/alexnet.py:49-55 via :141-155): out[m][n] = relu(b[n] + sum_k x[m][k] W[k][n]).
// At the CNN batch (M = 64 images) the layers are balanced between HBM and the matrix pipe: 151 + 67 MB of weights
// streamed per batch (36 us at 6 TB/s) against 4.8 + 2.1 GFLOP of fp32 MFMA (44 us at the 157 TFLOP/s peak), so the
// kernel is a weight STREAM feeding MFMAs: same wave-tile scheme as svx_conv.hip (one wave = 32*NA neurons x 32*NB
// images, NA*NB accumulators of v_mfma_f32_32x32x2_f32, operands by 16-byte buffer loads three octets of k ahead, no
// LDS, no barrier), with the weights packed once per model as [N/32][K/8][32][8] so that a wave's loads walk one
// contiguous 1 KB-per-octet stream, and K split over SPLITS waves so that ~1000 waves keep enough bytes in flight
// to saturate the HBM.  The split's partial sums go to a scratch slab (L2 / Infinity-Cache resident: 8-17 MB rewritten
// every batch) and are added in FIXED order by a second small kernel that also applies bias + ReLU: results are
// bit-reproducible from run to run, unlike an atomic-add reduction.
// Why not the vendor GEMM: hipBLASLt picks stream-K kernels for these shapes; two or three of them running at once on
// different HIP streams (the pipeline replays one graph per stream) can wait for each other's non-resident workgroups
// forever -- observed as a hang at batch 128 with 3 streams.  A wave here never waits for another wave.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/svx.h"
#include "svx_shadow.hpp"

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int THREADS = 256, WAVES = THREADS / 64;

__device__ __forceinline__ v4f buf_load4(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff)
{
    return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)soff, 0));
}

#ifndef SVX_FC_RING_A
#define SVX_FC_RING_A 6
#endif

template <int U> struct StageIndex { static constexpr int value = U; };

// stage(0) .. stage(N - 1) with compile-time indices
template <int N, int U = 0, typename F>
__device__ __forceinline__ void svx_unroll_stages(F& f)
{
    if constexpr (U < N) { f(StageIndex<U>{}); svx_unroll_stages<N, U + 1>(f); }
}

// the first `left` (< N) stages: no exits inside the unrolled body above (see svx_conv.hip)
template <int N, int U = 0, typename F>
__device__ __forceinline__ void svx_tail_stages(F& f, int left)
{
    if constexpr (U < N - 1) {
        if (left > U) { f(StageIndex<U>{}); svx_tail_stages<N, U + 1>(f, left); }
    }
}

struct FcArgs { const float* x; const float* w; float* part; int M, N, K, splits; };

template <int NA, int NB>
__global__ __launch_bounds__(THREADS, 2)
void fc_splitk_kernel(const FcArgs a)
{
    SVX_SHADOW_ROOM();
    const int n_tiles = a.N / (32 * NA), m_tiles = (a.M + 32 * NB - 1) / (32 * NB);
    const int total = m_tiles * a.splits * n_tiles;                 // neuron tile fastest: neighbours share the x slice
    const int wt = blockIdx.x * WAVES + (threadIdx.x >> 6);
    if (wt >= total) return;
    const int nt = wt % n_tiles, rest = wt / n_tiles;
    const int s = rest % a.splits, mt = rest / a.splits;
    const int lane = threadIdx.x & 63, hi = lane >> 5, lo = lane & 31;
    const int KQ = a.K / 8;
    const int q0 = (int)((long long)KQ * s / a.splits), q1 = (int)((long long)KQ * (s + 1) / a.splits);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, (int)((long long)a.N * a.K * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)((long long)a.M * a.K * 4), 0x00020000);
    unsigned voff_a[NA], voff_b[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) voff_a[i] = (unsigned)((((nt * NA + i) * KQ + q0) * 32 + lo) * 32 + 16 * hi);
#pragma unroll
    for (int t = 0; t < NB; ++t) voff_b[t] = (unsigned)(((mt * NB + t) * 32 + lo) * a.K * 4 + (q0 * 8 + 4 * hi) * 4);   // rows >= M: outside -> 0
    v16f acc[NA][NB];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int t = 0; t < NB; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.0f;
    // Two rings: the weights are a once-only HBM stream (latency ~2 us: a wave keeps RA - 1 octets = 5 KB of them in
    // flight; at the two octets of the activations' ring ~1000 waves held 2 MB in flight and the stream ran at 3 TB/s),
    // the activations come out of L2 (RB - 1 = 2 octets ahead, as in svx_conv.hip).  RA is a multiple of RB: the body is
    // unrolled RA stages so that every ring slot is a compile-time register.
    constexpr int RA = SVX_FC_RING_A, RB = 3, MF = 4 * NA * NB, L = NA + NB;
    static_assert(RA % RB == 0, "weight ring must be a multiple of the activation ring");
    v4f ra[RA][NA], rb[RB][NB];
    unsigned soff_a = 0, soff_b = 0;                      // of the NEXT octet each ring loads
    auto load_a = [&](int slot) {
#pragma unroll
        for (int i = 0; i < NA; ++i) ra[slot][i] = buf_load4(rs_w, voff_a[i], soff_a);
        soff_a += 1024u;
    };
    auto load_b1 = [&](int slot, int t) { rb[slot][t] = buf_load4(rs_x, voff_b[t], soff_b); };
    // stage u (0 <= u < RA, compile time): MFMAs of the octet in slots (u % RA, u % RB); the weight load for the octet
    // RA - 1 later and the activation loads for the octet RB - 1 later are spread between them
    auto stage = [&](auto u_) {
        constexpr int u = decltype(u_)::value;
        constexpr int sa = u % RA, sb = u % RB, la = (u + RA - 1) % RA, lb = (u + RB - 1) % RB;
#pragma unroll
        for (int m = 0; m < MF; ++m) {
            if (m == 0) load_a(la);
#pragma unroll
            for (int t = 0; t < NB; ++t)
                if ((t + 1) * MF / L == m) load_b1(lb, t);
            const int j = m / (NA * NB), r = m - j * (NA * NB);
            const int i = r / NB, t = r - i * NB;
            acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[sa][i][j], rb[sb][t][j], acc[i][t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        soff_b += 32u;
    };
    const int Q = q1 - q0;
    // prologue (a split shorter than the rings reads a neighbour's weights / x columns: valid memory or zeros, never used)
#pragma unroll
    for (int p = 0; p < RA - 1; ++p) load_a(p);
#pragma unroll
    for (int p = 0; p < RB - 1; ++p) {
#pragma unroll
        for (int t = 0; t < NB; ++t) load_b1(p, t);
        soff_b += 32u;
    }
    __builtin_amdgcn_sched_barrier(0);
    const int QU = Q - Q % RA;
    for (int q = 0; q < QU; q += RA) svx_unroll_stages<RA>(stage);
    svx_tail_stages<RA>(stage, Q - QU);
    // partial sums: part[s][m][n]; lane holds image m = lo of the tile column, neurons 8u + 4hi + (0..3) of the 32-block
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int t = 0; t < NB; ++t) {
            const int m = (mt * NB + t) * 32 + lo;
            if (m >= a.M) continue;
            float* o = a.part + ((size_t)s * a.M + m) * a.N + (nt * NA + i) * 32 + 4 * hi;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                *reinterpret_cast<v4f*>(o + 8 * u) = v4f{acc[i][t][4 * u], acc[i][t][4 * u + 1], acc[i][t][4 * u + 2], acc[i][t][4 * u + 3]};
        }
}

// out[m][n] = act(bias[n] + part[0][m][n] + part[1][m][n] + ...): fixed order, one float4 per lane
__global__ __launch_bounds__(THREADS)
void fc_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias, float* __restrict__ out, int MN4, int N4, int splits, int relu)
{
    const int e = blockIdx.x * THREADS + threadIdx.x;
    if (e >= MN4) return;
    const float4* p = reinterpret_cast<const float4*>(part);
    float4 v = reinterpret_cast<const float4*>(bias)[e % N4];
    for (int s = 0; s < splits; ++s) { const float4 q = p[(size_t)s * MN4 + e]; v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
    if (relu) { v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f); v.z = fmaxf(v.z, 0.0f); v.w = fmaxf(v.w, 0.0f); }
    reinterpret_cast<float4*>(out)[e] = v;
}

// wave tile: 32 neurons x 64 images up to one batch of 64; 64 x 64 (four independent accumulator chains, a quarter fewer
// operand loads per MFMA) from two batches on, where it is 10 % faster on fc6 (tools/ab_fc.py)
int fc_na(uint32_t m, uint32_t n)
{
    int na = (m > 64 && n % 64 == 0) ? 2 : 1;
#ifdef SVX_CONV_EXPERIMENT
    if (const char* e = getenv("SVX_FC_NA")) if (atoi(e) == 1 || (atoi(e) == 2 && m > 32 && n % 64 == 0)) na = atoi(e);
#endif
    return na;
}

int fc_splits(uint32_t m, uint32_t n, uint32_t k)
{
    // about one wave per SIMD (1024 on MI355X), each of >= 24 octets: measured best at M = 64 and M = 128 (tools/ab_fc.py;
    // twice as many cost 10-20 % more in partial-sum traffic and prologues, fewer leave SIMDs without a wave)
    const int tiles = (int)(n / (32 * fc_na(m, n))) * (int)((m + 63) / 64);
    int s = (1024 + tiles - 1) / tiles;
    const int max_s = (int)(k / 8) / 24;
    if (s > max_s) s = max_s;
#ifdef SVX_CONV_EXPERIMENT
    if (const char* e = getenv("SVX_FC_SPLITS")) if (atoi(e) >= 1 && atoi(e) <= max_s) s = atoi(e);
#endif
    return s < 1 ? 1 : s;
}

}  // namespace

extern "C" size_t svx_fc_ws_bytes(uint32_t m, uint32_t n, uint32_t k)
{
    return (size_t)fc_splits(m, n, k) * m * n * sizeof(float);
}

extern "C" int svx_fc_bias_act(const float* d_x, const float* d_w_packed, const float* d_bias, float* d_out, float* d_ws,
                               uint32_t m, uint32_t n, uint32_t k, int relu, void* stream)
{
    if (m == 0) return SVX_OK;
    if (!d_x || !d_w_packed || !d_bias || !d_out || !d_ws || n % 32 || k % 8 || k < 24 * 8) return SVX_EINVAL;
    for (const void* p : {(const void*)d_x, (const void*)d_w_packed, (const void*)d_bias, (const void*)d_out, (const void*)d_ws})
        if (reinterpret_cast<uintptr_t>(p) & 15u) return SVX_EINVAL;
    if ((uint64_t)n * k * 4 > 0x7fffffffull || (uint64_t)m * k * 4 > 0x7fffffffull) return SVX_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    FcArgs a{d_x, d_w_packed, d_ws, (int)m, (int)n, (int)k, fc_splits(m, n, k)};
    const int m_tiles = (int)((m + 63) / 64);
    const int na = fc_na(m, n);
    const int waves = m_tiles * a.splits * (int)(n / (32 * na));
    const dim3 grid((unsigned)((waves + WAVES - 1) / WAVES));
    if (na == 2)      hipLaunchKernelGGL((fc_splitk_kernel<2, 2>), grid, dim3(THREADS), 0, st, a);
    else if (m <= 32) hipLaunchKernelGGL((fc_splitk_kernel<1, 1>), grid, dim3(THREADS), 0, st, a);
    else              hipLaunchKernelGGL((fc_splitk_kernel<1, 2>), grid, dim3(THREADS), 0, st, a);
    const int mn4 = (int)((uint64_t)m * n / 4);
    hipLaunchKernelGGL(fc_reduce_kernel, dim3((mn4 + THREADS - 1) / THREADS), dim3(THREADS), 0, st, d_ws, d_bias, d_out, mn4, (int)(n / 4), a.splits, relu);
    return hipGetLastError() == hipSuccess ? SVX_OK : SVX_ELAUNCH;
}
