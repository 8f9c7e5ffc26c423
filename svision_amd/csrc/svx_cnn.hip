// svx_cnn.hip -- memory-bound glue of the AlexNet forward for gfx950 (MI355X), fp32.
//
// The dense contractions (conv / fc) stay on the matrix cores; everything between them
// in the reference graph
//     conv -> bias_add -> relu -> max_pool 3x3/2 VALID -> LRN(radius 2, alpha 2e-5, beta .75, bias 1)
//     (src/network/alexnet.py:29-31,34-36,45-46; helpers :132-135, :158-161, :164-166)
// is one pass here instead of five elementwise launches: each workgroup produces one pooled
// row of one image for ALL channels, keeps the pooled values in LDS, and applies the
// cross-channel normalisation from there.  HBM traffic = conv output read once (+ halo
// rows through L2) and the 4x smaller pooled tensor written once.  Activations are in the
// C8 layout of include/svx.h ([image][C/8][H][W][8]).
#include "svx_raster_common.hpp"

namespace {

constexpr int BLOCK = 256, WAVE = 64;

// v / (k + alpha*s)^beta.  The reference network uses beta = 0.75: x^-0.75 = rsqrt(x) * rsqrt(sqrt(x)),
// three hardware ops (a few ulp) instead of a ~100-instruction powf.
__device__ inline float lrn_scale(float v, float x, float beta)
{
    if (beta == 0.75f) return v * __frsqrt_rn(x) * __frsqrt_rn(__fsqrt_rn(x));
    return v / powf(x, beta);
}

__device__ inline float4 max4(float4 a, float4 b) { return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w)); }

// relu(max(window) + bias) == max(relu(x + bias)) : + and relu are monotonic.
// C8 layout in and out ([image][C/8][H][W][8], include/svx.h): a lane owns four channels (half an octet) of one pooled
// pixel -- nine 16-byte loads, one 16-byte store -- and consecutive lanes walk the two halves of the octet, then the
// pooled pixels of the row: every load and store instruction touches whole 32-byte sectors (with one channel per lane
// the kernel issued four times the loads and ran at a third of the HBM rate).
__global__ __launch_bounds__(BLOCK)
void bias_relu_pool_lrn_kernel(const float* __restrict__ x, const float* __restrict__ bias, float* __restrict__ y,
                               int C, int H, int W, int OH, int OW, int lrn, int radius, float alpha, float beta, float k,
                               const uint32_t* __restrict__ active_rows, const float* __restrict__ background)
{
    extern __shared__ __attribute__((aligned(16))) float pooled[];    // [OW][C + 1]
    const int b = blockIdx.x / OH;
    const int oy = blockIdx.x - b * OH;
    const int CP = C + 1, HW = H * W;
    const float* xb = x + ((size_t)b * (C / 8) * HW + (size_t)(2 * oy) * W) * 8;
    // active_rows (with background): bit x of word [image][row] = that input pixel was computed by the active-set convolution;
    // the others were NOT written and are read from the background tensor (their exact value) instead -- the producer saves
    // the copy, this kernel most of its HBM reads (the background of a layer is L2 resident)
    const float* gb = background ? background + (size_t)(2 * oy) * W * 8 : xb;
    uint32_t rows[3] = {~0u, ~0u, ~0u};
    if (active_rows) {
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) rows[dy] = active_rows[(size_t)b * H + 2 * oy + dy];
    }
    const int n = (C / 4) * OW;
    for (int idx = threadIdx.x; idx < n; idx += BLOCK) {
        const int h = idx & 1, rest = idx >> 1;
        const int oct = rest / OW, ox = rest - oct * OW;
        const size_t at = ((size_t)oct * HW + 2 * ox) * 8 + 4 * h;
        const float4* p = reinterpret_cast<const float4*>(xb + at);   // pixel = 2 float4
        const float4* g = reinterpret_cast<const float4*>(gb + at);
        float4 m;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int off = 2 * (dy * W + dx);
                const float4 v = ((rows[dy] >> (2 * ox + dx)) & 1u) ? p[off] : g[off];
                m = (dy | dx) ? max4(m, v) : v;
            }
        const int c = oct * 8 + 4 * h;
        const float4 bv = *reinterpret_cast<const float4*>(bias + c);
        float* q = pooled + ox * CP + c;
        q[0] = fmaxf(m.x + bv.x, 0.0f); q[1] = fmaxf(m.y + bv.y, 0.0f); q[2] = fmaxf(m.z + bv.z, 0.0f); q[3] = fmaxf(m.w + bv.w, 0.0f);
    }
    __syncthreads();
    float* yb = y + ((size_t)b * (C / 8) * OH * OW + (size_t)oy * OW) * 8;
    for (int idx = threadIdx.x; idx < n; idx += BLOCK) {
        const int h = idx & 1, rest = idx >> 1;
        const int oct = rest / OW, ox = rest - oct * OW;
        const int c = oct * 8 + 4 * h;
        const float* q = pooled + ox * CP;
        float v[4] = {q[c], q[c + 1], q[c + 2], q[c + 3]};
        if (lrn) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float s = 0.0f;                               // ascending channel order, as tf.nn.lrn's window sum
                const int lo = max(0, c + i - radius), hi = min(C - 1, c + i + radius);
                for (int j = lo; j <= hi; ++j) s += q[j] * q[j];
                v[i] = lrn_scale(v[i], k + alpha * s, beta);
            }
        }
        *reinterpret_cast<float4*>(yb + ((size_t)oct * OH * OW + ox) * 8 + 4 * h) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// ---------------------------------------------------------------------------------------------
// Encode + first layer in one kernel.  The similarity image is 0/255 on <= 3 thin lines and 0
// elsewhere, so conv1 (11x11 stride 4 VALID, 3 -> 96) of the mean-subtracted image is
//     base[k] + 255 * sum_{set pixels (ch,r,c) in the window} w[r-4oy][c-4ox][ch][k],
//     base[k] = bias[k] - sum_{ky,kx,ch} mean[ch] * w[ky][kx][ch][k]      (VALID: position independent)
// i.e. a few dozen 96-wide weight rows per output position instead of 363.  The 227x227x3 fp32
// image (618 KB) is never materialised: the bit planes are drawn in LDS (same code as
// svx_rasterize), each lane owns (one pooled pixel, 8 output channels), walks the 3x3 conv
// positions under it, extracts the 11-bit window masks from the bit planes and accumulates the
// weight rows of the set taps (checkpoint layout HWIO = [tap][96 channels], 32 B per lane,
// L2 resident).  ReLU, max-pool and the cross-channel LRN follow from LDS.
// threads per encode workgroup: 256 measured best (93 us per 256 images; 128: 137, 192: 106, 320: 108, 384: 110, 512: 110) --
// six workgroups per CU instead of five, the phases of a workgroup are too short to need more lanes
#ifndef SVX_ENC_BLOCK
#define SVX_ENC_BLOCK 256
#endif
constexpr int C1 = 96, C1_GROUPS = 12, P1 = 27, ENC_BLOCK = SVX_ENC_BLOCK, C1P = C1 + 1;   // conv1 output is 55x55, pooled 27x27

__device__ inline unsigned window_mask(const unsigned* row_words, int c0)
{
    const int w = c0 >> 5, sh = c0 & 31;
    unsigned m = row_words[w] >> sh;
    if (sh > 21) m |= row_words[w + 1] << (32 - sh);
    return m & 0x7FFu;
}

#ifdef SVX_ENC_PROFILE
__device__ unsigned long long svx_enc_prof[8];
#define ENC_MARK(i_) do { __syncthreads(); if (threadIdx.x == 0) { const unsigned long long t_ = wall_clock64(); atomicAdd(&svx_enc_prof[i_], t_ - t_prev); t_prev = t_; } } while (0)
#else
#define ENC_MARK(i_) do { } while (0)
#endif

// ENC_ROWS pooled rows per workgroup (the planes are drawn once for all of them, their touched windows share one queue).
// ONE is the measured optimum: the kernel is a chain of short latency-bound phases and lives on the number of
// workgroups a CU holds (four at 34 KB of LDS); three rows per workgroup -- a third of the drawing, fuller passes over the
// lanes, but two resident workgroups -- take 141 us per 128 images instead of 78 (tools/exp/ab_encode.py).
#ifndef SVX_ENC_ROWS
#define SVX_ENC_ROWS 1
#endif
constexpr int ENC_ROWS = SVX_ENC_ROWS;
static_assert(P1 % ENC_ROWS == 0 && ENC_ROWS * WAVE <= ENC_BLOCK, "rows per workgroup: a divisor of 27, one wave each for the masks");

struct EncLds {
    unsigned bits[2 * svx_raster::PLANE_WORDS];       // plane 0 (all segments) and the reverse-segment plane; plane 1 = plane 0 & colmask
    unsigned colcnt[svx_raster::IMG];
    unsigned colmask[svx_raster::ROW_WORDS + 1];      // (+1: window_mask may look one word past a row)
    unsigned pooled_bits[ENC_ROWS][P1 * C1P];         // [row][ox][k] pooled activations as float bit patterns (padded: no bank conflicts)
    unsigned rowany[ENC_ROWS][3][svx_raster::ROW_WORDS];
    int has_empty[ENC_ROWS][P1];                      // per conv row of a strip: OR of its 11 image rows x 3 planes
    unsigned short queue[ENC_ROWS * P1 * 9];
    int n_queue;
};

__global__ __launch_bounds__(ENC_BLOCK)
void encode_conv1_kernel(const int32_t* __restrict__ records, const float* __restrict__ w1, const float* __restrict__ base,
                         float* __restrict__ y, int lrn, int radius, float alpha, float beta, float kk,
                         uint32_t* __restrict__ touched)
{
    using namespace svx_raster;
    // The LDS is taken as DYNAMIC shared memory although its size is a constant: for a kernel whose occupancy its static LDS
    // limits, the compiler raises the VGPR allocation in the kernel descriptor to the most that occupancy leaves room for
    // (AMDGPUAsmPrinter: getMinNumVGPRs(max waves per EU)) -- 73 instead of the 38 this kernel uses, 129 instead of 66 for
    // bgzf_lz_kernel -- and registers a wave does not use are registers the waves of OTHER kernels (the convolutions of the
    // next stream's launch, which fill this one's tail) cannot have.
    extern __shared__ __attribute__((aligned(16))) unsigned char enc_lds_raw[];
    EncLds& L = *reinterpret_cast<EncLds*>(enc_lds_raw);
    unsigned (&bits)[2 * PLANE_WORDS] = L.bits;
    unsigned (&colcnt)[IMG] = L.colcnt;
    unsigned (&colmask)[ROW_WORDS + 1] = L.colmask;
    unsigned (&pooled_bits)[ENC_ROWS][P1 * C1P] = L.pooled_bits;
    unsigned (&rowany)[ENC_ROWS][3][ROW_WORDS] = L.rowany;
    int (&has_empty)[ENC_ROWS][P1] = L.has_empty;
    unsigned short (&queue)[ENC_ROWS * P1 * 9] = L.queue;
    int& n_queue = L.n_queue;

    constexpr int STRIPS = P1 / ENC_ROWS;
    const int img = blockIdx.x / STRIPS;
    const int oyp0 = (blockIdx.x - img * STRIPS) * ENC_ROWS;          // first pooled row of this workgroup
#ifdef SVX_ENC_PROFILE
    unsigned long long t_prev = wall_clock64();
#endif
    if (threadIdx.x == 0) colmask[ROW_WORDS] = 0;
    draw_planes<ENC_BLOCK, false>(records + (size_t)img * 12, bits, colcnt, colmask);
    ENC_MARK(0);

    const int tid = threadIdx.x;
    if (tid < ENC_ROWS * 3 * ROW_WORDS) {
        const int rr = tid / (3 * ROW_WORDS), t2 = tid - rr * 3 * ROW_WORDS;
        const int dy = t2 / ROW_WORDS, w = t2 - dy * ROW_WORDS;
        const int r0 = 4 * (2 * (oyp0 + rr) + dy);
        unsigned any = 0;
        for (int ky = 0; ky < 11; ++ky)
            any |= bits[(r0 + ky) * ROW_WORDS + w] | bits[PLANE_WORDS + (r0 + ky) * ROW_WORDS + w];       // plane 1 is a subset of plane 0
        rowany[rr][dy][w] = any;
    }
    __syncthreads();
    {                                                 // does this pooled pixel see at least one empty window?  (wave rr: row rr)
        const int rr = tid >> 6, lane = tid & (WAVE - 1);
        bool any_empty = false, any_set = false;
        if (rr < ENC_ROWS && lane < P1)
            for (int win = 0; win < 9; ++win) {
                const bool empty = window_mask(rowany[rr][win / 3], 4 * (2 * lane + win % 3)) == 0;
                any_empty |= empty;
                any_set |= !empty;
            }
        if (rr < ENC_ROWS && lane < P1) has_empty[rr][lane] = any_empty ? 1 : 0;
        if (tid == 0) n_queue = 0;
        // pooled pixels with a set tap under them: everything else in this row is the constant background vector
        const unsigned long long t = __ballot(rr < ENC_ROWS && lane < P1 && any_set);
        if (touched && rr < ENC_ROWS && lane == 0) touched[(size_t)img * P1 + oyp0 + rr] = (uint32_t)t;
    }
    __syncthreads();
    for (int i = tid; i < ENC_ROWS * P1 * C1; i += ENC_BLOCK) {  // empty windows all respond relu(base[k]); relu floor otherwise
        const int rr = i / (P1 * C1), i2 = i - rr * (P1 * C1);
        const int ox = i2 / C1, k = i2 - ox * C1;
        pooled_bits[rr][ox * C1P + k] = has_empty[rr][ox] ? __float_as_uint(fmaxf(base[k], 0.0f)) : 0u;
    }
    // ~90 % of the 27 x 9 (pooled pixel, conv window under it) pairs of a row are empty (constant response base[k]);
    // the touched ones are compacted into a queue so that every lane below has work: one lane per (touched window,
    // 8-channel group) walks the window's 33 row masks and adds the weight rows of its set taps.  Max-pool = integer
    // atomic max on the (non-negative) float bit patterns: exact and order independent.
    for (int e = tid; e < ENC_ROWS * P1 * 9; e += ENC_BLOCK) {
        const int rr = e / (P1 * 9), pw = e - rr * (P1 * 9);
        const int oxp = pw / 9, win = pw - oxp * 9;
        if (window_mask(rowany[rr][win / 3], 4 * (2 * oxp + win % 3)) != 0) queue[atomicAdd(&n_queue, 1)] = (unsigned short)e;
    }
    __syncthreads();
    ENC_MARK(1);
    const int n_items = n_queue * C1_GROUPS;
    for (int item = tid; item < n_items; item += ENC_BLOCK) {
        const int g = item % C1_GROUPS, e = queue[item / C1_GROUPS];
        const int rr = e / (P1 * 9), pw = e - rr * (P1 * 9);
        const int oxp = pw / 9, win = pw - oxp * 9;
        const int dy = win / 3, dx = win - dy * 3;
        const int oy = 2 * (oyp0 + rr) + dy, ox = 2 * oxp + dx;
        const float4 b0 = reinterpret_cast<const float4*>(base)[2 * g];
        const float4 b1 = reinterpret_cast<const float4*>(base)[2 * g + 1];
        float acc[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        {
            // (A two-deep software pipeline of the weight-row loads -- request a tap's rows, add the previous tap's -- was
            // measured and is not faster: the phase is bound by the 66 dependent LDS mask reads per item and by the
            // workgroups sharing a CU, not by the L2 latency of the rows; tools/exp/enc_prof.py.)
            const unsigned cm = window_mask(colmask, 4 * ox);          // columns with >= 2 hits (row independent): channel 1 = channel 0 & cm
            for (int ky = 0; ky < 11; ++ky) {
                const int r = 4 * oy + ky;
                const unsigned m0 = window_mask(bits + r * ROW_WORDS, 4 * ox);
                const unsigned m2 = window_mask(bits + PLANE_WORDS + r * ROW_WORDS, 4 * ox);
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    unsigned m = ch == 0 ? m0 : (ch == 1 ? (m0 & cm) : m2);
                    while (m) {
                        const int kx = __ffs(m) - 1;
                        m &= m - 1;
                        const float4* wp = reinterpret_cast<const float4*>(w1 + ((ky * 11 + kx) * 3 + ch) * C1) + 2 * g;
                        const float4 u0 = wp[0], u1 = wp[1];
                        acc[0] = fmaf(255.0f, u0.x, acc[0]); acc[1] = fmaf(255.0f, u0.y, acc[1]);
                        acc[2] = fmaf(255.0f, u0.z, acc[2]); acc[3] = fmaf(255.0f, u0.w, acc[3]);
                        acc[4] = fmaf(255.0f, u1.x, acc[4]); acc[5] = fmaf(255.0f, u1.y, acc[5]);
                        acc[6] = fmaf(255.0f, u1.z, acc[6]); acc[7] = fmaf(255.0f, u1.w, acc[7]);
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (acc[j] > 0.0f) atomicMax(&pooled_bits[rr][oxp * C1P + 8 * g + j], __float_as_uint(acc[j]));
    }
    __syncthreads();
    ENC_MARK(2);
    // C8 output [image][12 octets][27][27][8]: consecutive lanes -> the 8 channels of an octet, then consecutive ox
    // (one contiguous 864-byte run per octet and row)
    for (int i = tid; i < ENC_ROWS * C1 * P1; i += ENC_BLOCK) {
        const int rr = i / (C1 * P1), idx = i - rr * (C1 * P1);
        const float* pooled = reinterpret_cast<const float*>(pooled_bits[rr]);
        float* yb = y + ((size_t)img * (C1 / 8) * P1 * P1 + (size_t)(oyp0 + rr) * P1) * 8;
        const int c8 = idx & 7, rest = idx >> 3;
        const int oct = rest / P1, ox = rest - oct * P1;
        const int k = oct * 8 + c8;
        float v = pooled[ox * C1P + k];
        if (lrn) {
            float s = 0.0f;
            const int lo = max(0, k - radius), hi = min(C1 - 1, k + radius);
            for (int j = lo; j <= hi; ++j) { const float q = pooled[ox * C1P + j]; s += q * q; }
            v = lrn_scale(v, kk + alpha * s, beta);
        }
        yb[((size_t)oct * P1 * P1 + ox) * 8 + c8] = v;
    }
    ENC_MARK(3);
#ifdef SVX_ENC_PROFILE
    if (threadIdx.x == 0) { atomicAdd(&svx_enc_prof[4], 1ull); atomicAdd(&svx_enc_prof[5], (unsigned long long)n_queue); }
#endif
}

}  // namespace

#ifdef SVX_ENC_PROFILE
extern "C" int svx_debug_enc_prof(unsigned long long* out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(svx_enc_prof), sizeof(svx_enc_prof)) != hipSuccess) return SVX_ELAUNCH;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(svx_enc_prof), z, sizeof(z)) != hipSuccess) return SVX_ELAUNCH; }
    return SVX_OK;
}
#endif

extern "C" int svx_encode_conv1(const int32_t* d_records, uint32_t n, const float* d_w1, const float* d_base, float* d_y,
                                int lrn, uint32_t radius, float alpha, float beta, float k, uint32_t* d_touched, void* stream)
{
    if (n == 0) return SVX_OK;
    if (!d_records || !d_w1 || !d_base || !d_y) return SVX_EINVAL;
    if ((reinterpret_cast<uintptr_t>(d_w1) & 15u) || (reinterpret_cast<uintptr_t>(d_base) & 15u)) return SVX_EINVAL;
    hipLaunchKernelGGL(encode_conv1_kernel, dim3(n * (P1 / ENC_ROWS)), dim3(ENC_BLOCK), sizeof(EncLds), static_cast<hipStream_t>(stream),
                       d_records, d_w1, d_base, d_y, lrn, (int)radius, alpha, beta, k, d_touched);
    return hipGetLastError() == hipSuccess ? SVX_OK : SVX_ELAUNCH;
}

namespace {
// Which outputs of conv2..conv5 can differ from the network's response to an empty image?  Everything downstream of
// the first layer is local: a pooled conv1 pixel without a set tap under it holds the constant background vector,
// a conv output whose window sees only background inputs holds the (position dependent, image independent) background
// response, and so on through the pools.  A workgroup per image turns the 27 touched-row words of svx_encode_conv1
// into the active masks of the four convolutions (5x5 dilation -> 3x3/2 pool -> three 3x3 dilations); a first pass
// counts, a second one derives every image's offsets from the counts and writes, per layer, a permutation of all
// pixel ids (image * H*W + y * W + x): the active ones first (ascending), then the inactive ones (ascending) --
// svx_conv2d_same computes the first part and copies the background into the second.
constexpr int A1 = 27, A2 = 13;

__device__ inline uint32_t dilate(uint32_t m, int r, uint32_t full) { uint32_t o = m; for (int i = 1; i <= r; ++i) o |= (m << i) | (m >> i); return o & full; }

// masks of one image in LDS: rows [0,27) conv2, then 13 each for conv3, conv4, conv5
constexpr int MASK_ROWS = A1 + 3 * A2;

__device__ inline void image_masks(const uint32_t* __restrict__ touched_rows, uint32_t* masks, uint32_t* tmp, int t)
{
    // all threads of the workgroup call this; t = thread index (>= 32 threads do the work)
    const uint32_t full1 = (1u << A1) - 1u, full2 = (1u << A2) - 1u;
    if (t < A1) tmp[t] = dilate(touched_rows[t], 2, full1);
    __syncthreads();
    if (t < A1) {                                            // conv2: 5x5 window
        uint32_t v = 0;
        for (int d = -2; d <= 2; ++d) if (t + d >= 0 && t + d < A1) v |= tmp[t + d];
        masks[t] = v;
    }
    __syncthreads();
    if (t < A2) {                                            // pool2: 3x3 stride 2 (VALID), then the conv3 row dilation
        const uint32_t rows = masks[2 * t] | masks[2 * t + 1] | masks[2 * t + 2];
        uint32_t v = 0;
        for (int x = 0; x < A2; ++x) if ((rows >> (2 * x)) & 7u) v |= 1u << x;
        tmp[t] = dilate(v, 1, full2);
    }
    __syncthreads();
    for (int l = 0; l < 3; ++l) {                            // conv3, conv4, conv5: 3x3 windows
        uint32_t v = 0;
        if (t < A2) {
            v = tmp[t];
            if (t > 0) v |= tmp[t - 1];
            if (t + 1 < A2) v |= tmp[t + 1];
            masks[A1 + l * A2 + t] = v;
        }
        __syncthreads();
        if (t < A2) tmp[t] = dilate(v, 1, full2);
        __syncthreads();
    }
}

// pass 1: a workgroup per image: the number of active pixels of the four layers
__global__ __launch_bounds__(64)
void active_counts_kernel(const uint32_t* __restrict__ touched, uint32_t* __restrict__ per_image)
{
    __shared__ uint32_t masks[MASK_ROWS], tmp[A1];
    const int t = threadIdx.x;
    image_masks(touched + (size_t)blockIdx.x * A1, masks, tmp, t);
    if (t < 4) {
        const int lo = t == 0 ? 0 : A1 + (t - 1) * A2, hi = t == 0 ? A1 : lo + A2;
        uint32_t c = 0;
        for (int r = lo; r < hi; ++r) c += __popc(masks[r]);
        per_image[blockIdx.x * 4 + t] = c;
    }
}

// pass 2: a workgroup per image: its offsets = sum of the earlier images' counts, then the ascending pixel ids
__global__ __launch_bounds__(BLOCK)
void active_lists_kernel(const uint32_t* __restrict__ touched, const uint32_t* __restrict__ per_image, uint32_t n,
                         int32_t* __restrict__ list2, int32_t* __restrict__ list3, int32_t* __restrict__ list4,
                         int32_t* __restrict__ list5, uint32_t* __restrict__ counts, unsigned long long* __restrict__ totals,
                         uint32_t* __restrict__ active2)
{
    __shared__ uint32_t masks[MASK_ROWS], tmp[A1], rowoff[MASK_ROWS];
    __shared__ uint32_t s_part[8][BLOCK / WAVE];
    __shared__ uint32_t inoff[MASK_ROWS];                    // first slot of every mask row's inactive pixels
    const int t = threadIdx.x, lane = t & (WAVE - 1), wv = t >> 6;
    const uint32_t img = blockIdx.x;
    uint32_t part[8] = {0, 0, 0, 0, 0, 0, 0, 0};             // [0..4): images before this one, [4..8): all images
    for (uint32_t i = t; i < n; i += BLOCK) {
        const uint4 c = reinterpret_cast<const uint4*>(per_image)[i];
        if (i < img) { part[0] += c.x; part[1] += c.y; part[2] += c.z; part[3] += c.w; }
        part[4] += c.x; part[5] += c.y; part[6] += c.z; part[7] += c.w;
    }
    for (int l = 0; l < 8; ++l) {
        uint32_t v = part[l];
        for (int o = WAVE / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
        if (lane == 0) s_part[l][wv] = v;
    }
    image_masks(touched + (size_t)img * A1, masks, tmp, t);          // (contains barriers)
    if (t < 4) {
        uint32_t before = 0, total = 0;
        for (int w = 0; w < BLOCK / WAVE; ++w) { before += s_part[t][w]; total += s_part[4 + t][w]; }
        const int lo = t == 0 ? 0 : A1 + (t - 1) * A2, hi = t == 0 ? A1 : lo + A2;
        const uint32_t width = t == 0 ? A1 : A2, hw = width * width;
        // active pixels fill [0, total) in image order; the inactive ones follow in [total, n * hw), also in image order
        uint32_t run = before, irun = total + (img * hw - before);
        for (int r = lo; r < hi; ++r) {
            const uint32_t c = __popc(masks[r]);
            rowoff[r] = run; inoff[r] = irun;
            run += c; irun += width - c;
        }
        if (img == n - 1) {
            counts[t] = total;
            if (totals) {                                    // running sums over launches (measurement: executed work)
                const unsigned long long all = (unsigned long long)n * hw;
                // what svx_conv2d_same will compute: every pixel once the list is SVX_CONV_DENSE_PCT full
                atomicAdd(&totals[t], (unsigned long long)total * 100ull >= all * SVX_CONV_DENSE_PCT ? all : (unsigned long long)total);
                if (t == 0) atomicAdd(&totals[4], (unsigned long long)n);
            }
        }
    }
    __syncthreads();
    if (active2 && t < A1) active2[(size_t)img * A1 + t] = masks[t];          // conv2's active pixels as row masks (for its consumer)
    for (int p = t; p < A1 * A1; p += BLOCK) {
        const int y = p / A1, x = p - y * A1;
        const uint32_t m = masks[y], below = (1u << x) - 1u;
        const uint32_t slot = ((m >> x) & 1u) ? rowoff[y] + __popc(m & below) : inoff[y] + __popc(~m & below);
        list2[slot] = (int32_t)(img * (A1 * A1) + p);
    }
    int32_t* ls[3] = {list3, list4, list5};
    for (int p = t; p < 3 * A2 * A2; p += BLOCK) {
        const int l = p / (A2 * A2), q = p - l * (A2 * A2);
        const int y = q / A2, x = q - y * A2, r = A1 + l * A2 + y;
        const uint32_t m = masks[r], below = (1u << x) - 1u;
        const uint32_t slot = ((m >> x) & 1u) ? rowoff[r] + __popc(m & below) : inoff[r] + __popc(~m & below);
        ls[l][slot] = (int32_t)(img * (A2 * A2) + q);
    }
}
}  // namespace

extern "C" int svx_alexnet_active_sets(const uint32_t* d_touched, uint32_t n, int32_t* d_list2, int32_t* d_list3,
                                       int32_t* d_list4, int32_t* d_list5, uint32_t* d_counts, uint32_t* d_ws,
                                       uint64_t* d_totals, uint32_t* d_active2, void* stream)
{
    if (!d_counts) return SVX_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (n == 0) return hipMemsetAsync(d_counts, 0, 4 * sizeof(uint32_t), st) == hipSuccess ? SVX_OK : SVX_ELAUNCH;
    if (!d_touched || !d_list2 || !d_list3 || !d_list4 || !d_list5 || !d_ws || (reinterpret_cast<uintptr_t>(d_ws) & 15u)) return SVX_EINVAL;
    hipLaunchKernelGGL(active_counts_kernel, dim3(n), dim3(64), 0, st, d_touched, d_ws);
    hipLaunchKernelGGL(active_lists_kernel, dim3(n), dim3(BLOCK), 0, st, d_touched, d_ws, n, d_list2, d_list3, d_list4, d_list5, d_counts,
                       reinterpret_cast<unsigned long long*>(d_totals), d_active2);
    return hipGetLastError() == hipSuccess ? SVX_OK : SVX_ELAUNCH;
}

namespace {
// Last layer + outputs in one launch: logits = x @ W8^T + b8 (xw_plus_b, src/network/alexnet.py:58,148),
// tf.nn.softmax and tf.argmax (first maximal index) as fetched at src/network/predict.py:209.  One
// workgroup per image, one wave per class; packed row = [softmax x5, class, logits x5, pad].
constexpr int FC8_IN = 4096, FC8_OUT = 5;
__global__ __launch_bounds__(64 * FC8_OUT)
void fc8_softmax_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                        float* __restrict__ out)
{
    __shared__ float logit[FC8_OUT];
    const int img = blockIdx.x, cls = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float4* xv = reinterpret_cast<const float4*>(x + (size_t)img * FC8_IN);
    const float4* wv = reinterpret_cast<const float4*>(w + (size_t)cls * FC8_IN);
    float acc = 0.0f;
#pragma unroll 4
    for (int i = lane; i < FC8_IN / 4; i += 64) {
        const float4 a = xv[i], b = wv[i];
        acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc); acc = fmaf(a.z, b.z, acc); acc = fmaf(a.w, b.w, acc);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) logit[cls] = acc + bias[cls];
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = logit[0];
        int best = 0;
        for (int j = 1; j < FC8_OUT; ++j) if (logit[j] > m) { m = logit[j]; best = j; }
        float e[FC8_OUT], s = 0.0f;
        for (int j = 0; j < FC8_OUT; ++j) { e[j] = expf(logit[j] - m); s += e[j]; }
        float* o = out + (size_t)img * 12;
        for (int j = 0; j < FC8_OUT; ++j) { o[j] = e[j] / s; o[6 + j] = logit[j]; }
        o[5] = (float)best;
        o[11] = 0.0f;
    }
}
}  // namespace

extern "C" int svx_fc8_softmax(const float* d_x, const float* d_w, const float* d_bias, float* d_out, uint32_t n, void* stream)
{
    if (n == 0) return SVX_OK;
    if (!d_x || !d_w || !d_bias || !d_out) return SVX_EINVAL;
    if ((reinterpret_cast<uintptr_t>(d_x) & 15u) || (reinterpret_cast<uintptr_t>(d_w) & 15u)) return SVX_EINVAL;
    hipLaunchKernelGGL(fc8_softmax_kernel, dim3(n), dim3(64 * FC8_OUT), 0, static_cast<hipStream_t>(stream), d_x, d_w, d_bias, d_out);
    return hipGetLastError() == hipSuccess ? SVX_OK : SVX_ELAUNCH;
}

extern "C" int svx_bias_relu_pool_lrn(const float* d_x, const float* d_bias, float* d_y, uint32_t n, uint32_t channels,
                                      uint32_t height, uint32_t width, int lrn, uint32_t radius, float alpha, float beta,
                                      float k, const uint32_t* d_active_rows, const float* d_background, void* stream)
{
    if (n == 0) return SVX_OK;
    if (!d_x || !d_bias || !d_y || channels == 0 || channels % 8 || height < 3 || width < 3) return SVX_EINVAL;
    if ((reinterpret_cast<uintptr_t>(d_x) | reinterpret_cast<uintptr_t>(d_bias) | reinterpret_cast<uintptr_t>(d_y)) & 15u) return SVX_EINVAL;
    if ((d_active_rows == nullptr) != (d_background == nullptr) || (reinterpret_cast<uintptr_t>(d_background) & 15u) || (d_active_rows && width > 32)) return SVX_EINVAL;
    const int OH = (int)(height - 3) / 2 + 1, OW = (int)(width - 3) / 2 + 1;
    const size_t lds = (size_t)(channels + 1) * OW * sizeof(float);
    if (lds > 64 * 1024) return SVX_EINVAL;
    hipLaunchKernelGGL(bias_relu_pool_lrn_kernel, dim3(n * OH), dim3(BLOCK), lds, static_cast<hipStream_t>(stream),
                       d_x, d_bias, d_y, (int)channels, (int)height, (int)width, OH, OW, lrn, (int)radius, alpha, beta, k,
                       d_active_rows, d_background);
    return hipGetLastError() == hipSuccess ? SVX_OK : SVX_ELAUNCH;
}
