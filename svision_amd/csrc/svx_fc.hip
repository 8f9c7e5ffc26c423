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

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int THREADS = 256, WAVES = THREADS / 64;

__device__ __forceinline__ v4f buf_load4(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff)
{
    return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)soff, 0));
}

struct FcArgs { const float* x; const float* w; float* part; int M, N, K, splits; };

template <int NA, int NB>
__global__ __launch_bounds__(THREADS, 2)
void fc_splitk_kernel(const FcArgs a)
{
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
    constexpr int R = 3, L = NA + NB, MF = 4 * NA * NB;
    v4f ra[R][NA], rb[R][NB];
    unsigned soff_a = 0, soff_b = 0;
    auto load_one = [&](int slot, int q) {
        if (q < NA) ra[slot][q] = buf_load4(rs_w, voff_a[q], soff_a);
        else        rb[slot][q - NA] = buf_load4(rs_x, voff_b[q - NA], soff_b);
    };
    auto stage = [&](int ls, int cs, bool loads) {
#pragma unroll
        for (int m = 0; m < MF; ++m) {
            if (loads) {
#pragma unroll
                for (int q = 0; q < L; ++q)
                    if (q * MF / L == m) load_one(ls, q);
            }
            const int j = m / (NA * NB), r = m - j * (NA * NB);
            const int i = r / NB, t = r - i * NB;
            acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[cs][i][j], rb[cs][t][j], acc[i][t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (loads) { soff_a += 1024u; soff_b += 32u; }
    };
    const int Q = q1 - q0;
#pragma unroll
    for (int p = 0; p < R - 1; ++p) {                      // prologue: two octets ahead (a split shorter than that reads
#pragma unroll                                             // a neighbour's weights / x columns: valid memory, never used)
        for (int q = 0; q < L; ++q) load_one(p, q);
        soff_a += 1024u; soff_b += 32u;
    }
    __builtin_amdgcn_sched_barrier(0);
    const int Q3 = Q - Q % R;
    for (int q = 0; q < Q3; q += R) { stage(2, 0, true); stage(0, 1, true); stage(1, 2, true); }
    if (Q - Q3 >= 1) stage(2, 0, true);
    if (Q - Q3 >= 2) stage(0, 1, true);
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

int fc_splits(uint32_t m, uint32_t n, uint32_t k)
{
    // about one wave per SIMD (1024 on MI355X), each of >= 24 octets: measured best at M = 64 and M = 128 (tools/ab_fc.py:
    // 8 / 4 splits; 16 / 8 cost 10-20 % more in partial-sum traffic and prologues)
    const int tiles = (int)(n / 32) * (int)((m + 63) / 64);
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
    const int waves = m_tiles * a.splits * (int)(n / 32);
    int na = 1;
#ifdef SVX_CONV_EXPERIMENT
    if (const char* e = getenv("SVX_FC_NA")) na = atoi(e);
#endif
    if (na == 2 && m > 32 && n % 64 == 0) {
        hipLaunchKernelGGL((fc_splitk_kernel<2, 2>), dim3((waves / 2 + WAVES - 1) / WAVES), dim3(THREADS), 0, st, a);
    } else if (m <= 32) hipLaunchKernelGGL((fc_splitk_kernel<1, 1>), dim3((waves + WAVES - 1) / WAVES), dim3(THREADS), 0, st, a);
    else         hipLaunchKernelGGL((fc_splitk_kernel<1, 2>), dim3((waves + WAVES - 1) / WAVES), dim3(THREADS), 0, st, a);
    const int mn4 = (int)((uint64_t)m * n / 4);
    hipLaunchKernelGGL(fc_reduce_kernel, dim3((mn4 + THREADS - 1) / THREADS), dim3(THREADS), 0, st, d_ws, d_bias, d_out, mn4, (int)(n / 4), a.splits, relu);
    return hipGetLastError() == hipSuccess ? SVX_OK : SVX_ELAUNCH;
}
