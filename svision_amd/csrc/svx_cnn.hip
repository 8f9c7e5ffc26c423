// svx_cnn.hip -- memory-bound glue of the AlexNet forward for gfx950 (MI355X), fp32.
//
// The dense contractions (conv / fc) stay on the matrix cores; everything between them
// in the reference graph
//     conv -> bias_add -> relu -> max_pool 3x3/2 VALID -> LRN(radius 2, alpha 2e-5, beta .75, bias 1)
//     (src/network/alexnet.py:29-31,34-36,45-46; helpers :132-135, :158-161, :164-166)
// is one pass here instead of five elementwise launches: each workgroup produces one pooled
// row of one image for ALL channels, keeps the pooled values in LDS, and applies the
// cross-channel normalisation from there.  HBM traffic = conv output read once (+ halo
// rows through L2) and the 4x smaller pooled tensor written once.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/svx.h"

namespace {

constexpr int BLOCK = 256;

// relu(max(window) + bias) == max(relu(x + bias)) : + and relu are monotonic.
__global__ __launch_bounds__(BLOCK)
void bias_relu_pool_lrn_kernel(const float* __restrict__ x, const float* __restrict__ bias, float* __restrict__ y,
                               int C, int H, int W, int OH, int OW, int lrn, int radius, float alpha, float beta, float k)
{
    extern __shared__ __attribute__((aligned(16))) float pooled[];    // [C][OW]
    const int b = blockIdx.x / OH;
    const int oy = blockIdx.x - b * OH;
    const float* xb = x + (size_t)b * C * H * W + (size_t)(2 * oy) * W;
    const int n = C * OW;
    for (int idx = threadIdx.x; idx < n; idx += BLOCK) {
        const int c = idx / OW, ox = idx - c * OW;
        const float* p = xb + (size_t)c * H * W + 2 * ox;
        float m = p[0];
        m = fmaxf(m, p[1]); m = fmaxf(m, p[2]);
        m = fmaxf(m, p[W]); m = fmaxf(m, p[W + 1]); m = fmaxf(m, p[W + 2]);
        m = fmaxf(m, p[2 * W]); m = fmaxf(m, p[2 * W + 1]); m = fmaxf(m, p[2 * W + 2]);
        pooled[idx] = fmaxf(m + bias[c], 0.0f);
    }
    __syncthreads();
    float* yb = y + (size_t)b * C * OH * OW + (size_t)oy * OW;
    for (int idx = threadIdx.x; idx < n; idx += BLOCK) {
        const int c = idx / OW, ox = idx - c * OW;
        float v = pooled[idx];
        if (lrn) {
            float s = 0.0f;
            const int lo = max(0, c - radius), hi = min(C - 1, c + radius);
            for (int j = lo; j <= hi; ++j) { const float q = pooled[j * OW + ox]; s += q * q; }
            v = v / powf(k + alpha * s, beta);
        }
        yb[(size_t)c * OH * OW + ox] = v;
    }
}

}  // namespace

extern "C" int svx_bias_relu_pool_lrn(const float* d_x, const float* d_bias, float* d_y, uint32_t n, uint32_t channels,
                                      uint32_t height, uint32_t width, int lrn, uint32_t radius, float alpha, float beta,
                                      float k, void* stream)
{
    if (n == 0) return SVX_OK;
    if (!d_x || !d_bias || !d_y || channels == 0 || height < 3 || width < 3) return SVX_EINVAL;
    const int OH = (int)(height - 3) / 2 + 1, OW = (int)(width - 3) / 2 + 1;
    const size_t lds = (size_t)channels * OW * sizeof(float);
    if (lds > 64 * 1024) return SVX_EINVAL;
    hipLaunchKernelGGL(bias_relu_pool_lrn_kernel, dim3(n * OH), dim3(BLOCK), lds, static_cast<hipStream_t>(stream),
                       d_x, d_bias, d_y, (int)channels, (int)height, (int)width, OH, OW, lrn, (int)radius, alpha, beta, k);
    return hipGetLastError() == hipSuccess ? SVX_OK : SVX_ELAUNCH;
}
