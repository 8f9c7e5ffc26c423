// svx_raster.hip -- similarity-image rasteriser for gfx950 (MI355X).
//
// One launch encodes n segment pairs into the float32 batch tensor the CNN
// consumes (mean-subtracted, NHWC or NCHW).  Behaviour follows the reference
//   PlotSingleImg.plot            src/segmentplot/plot_segment.py:33-68
//   BatchGenerator.next_batch     src/network/create_batch.py:103-152
//   Segment.__init__              src/segmentplot/classes.py:44-54
// with cv2.line(thickness 1, LINE_8) = OpenCV clipLine + LineIterator.
//
// Mapping to the machine: the kernel is HBM-write bound (618,348 B per image
// against 48 B read).  A workgroup owns one contiguous slice of one image.
// It first rebuilds the image's three 227x227 bit planes in LDS (22 KB: a few
// hundred LDS atomics, one lane per Bresenham step via the closed form of the
// LineIterator error recurrence), derives channel 1 from per-column counts,
// then streams its slice to HBM with 16-byte stores aligned on the *global*
// address (image bases are only 4-byte aligned: 618,348 % 16 = 12).
#include "svx_raster_common.hpp"

namespace {

using namespace svx_raster;
constexpr int BLOCK = 256;

__device__ inline float elem_value(const unsigned* bits, int e, int layout, float m0, float m1, float m2)
{
    int ch, pix;
    if (layout == SVX_LAYOUT_NHWC) { pix = e / 3; ch = e - pix * 3; }
    else                           { ch = e / PLANE_ELEMS; pix = e - ch * PLANE_ELEMS; }
    int r = pix / IMG, c = pix - r * IMG;
    unsigned w = bits[ch * PLANE_WORDS + r * ROW_WORDS + (c >> 5)];
    float mean = ch == 0 ? m0 : (ch == 1 ? m1 : m2);
    return ((w >> (c & 31)) & 1u) ? 255.0f - mean : -mean;
}

// Four consecutive output floats starting at element e (e + 3 inside the image).  The planar layout reads
// one bit-plane word for the whole group; groups that straddle an image row, and the interleaved NHWC
// layout (measured slower with a grouped form), use the per-element form.
template <int LAYOUT>
__device__ inline float4 elem_group(const unsigned* bits, int e, float m0, float m1, float m2)
{
    float4 v;
    if (LAYOUT == SVX_LAYOUT_NCHW) {
        const int ch = e / PLANE_ELEMS, pix = e - ch * PLANE_ELEMS;
        const int r = pix / IMG, c = pix - r * IMG;
        if (c + 3 < IMG) {
            const unsigned* row = bits + ch * PLANE_WORDS + r * ROW_WORDS;
            const int w0 = c >> 5, sh = c & 31;
            unsigned b = row[w0] >> sh;
            if (sh > 28) b |= row[w0 + 1] << (32 - sh);
            const float mean = ch == 0 ? m0 : (ch == 1 ? m1 : m2);
            const float lo = -mean, hi = 255.0f - mean;
            v.x = (b & 1u) ? hi : lo; v.y = (b & 2u) ? hi : lo; v.z = (b & 4u) ? hi : lo; v.w = (b & 8u) ? hi : lo;
            return v;
        }
    }
    v.x = elem_value(bits, e + 0, LAYOUT, m0, m1, m2);
    v.y = elem_value(bits, e + 1, LAYOUT, m0, m1, m2);
    v.z = elem_value(bits, e + 2, LAYOUT, m0, m1, m2);
    v.w = elem_value(bits, e + 3, LAYOUT, m0, m1, m2);
    return v;
}

template <int LAYOUT>
__global__ __launch_bounds__(BLOCK)
void raster_kernel(const int32_t* __restrict__ records, uint32_t n, float* __restrict__ out,
                   int strips, float m0, float m1, float m2)
{
    constexpr int layout = LAYOUT;
    // plane 0: all segments, plane 1: columns with >= 2 hits, plane 2: reverse segments
    __shared__ unsigned bits[3 * PLANE_WORDS];
    __shared__ unsigned colcnt[IMG];
    __shared__ unsigned colmask[ROW_WORDS];

    const uint32_t img = blockIdx.x / strips;
    const int strip = blockIdx.x - img * strips;
    const int tid = threadIdx.x;

    draw_planes<BLOCK>(records + (size_t)img * 12, bits, colcnt, colmask);

    // stream this block's slice [e_lo, e_hi) of the image
    const int per = (IMG_ELEMS + strips - 1) / strips;
    const int e_lo = strip * per;
    const int e_hi = min(IMG_ELEMS, e_lo + per);
    if (e_lo >= e_hi) return;
    const long long base = (long long)img * IMG_ELEMS;       // global float index of element 0
    float* gout = out + base;
    // global float indices; 16-byte groups are aligned on the tensor base (assumed 16 B aligned)
    const long long g_lo = base + e_lo, g_hi = base + e_hi;
    const long long q_lo = (g_lo + 3) >> 2, q_hi = g_hi >> 2;   // full float4 groups [q_lo, q_hi)
    if (q_lo >= q_hi) {                                          // tiny slice: scalar only
        for (int e = e_lo + tid; e < e_hi; e += BLOCK) gout[e] = elem_value(bits, e, layout, m0, m1, m2);
        return;
    }
    const int head_end = (int)(q_lo * 4 - base);                 // elements [e_lo, head_end) scalar
    const int tail_beg = (int)(q_hi * 4 - base);                 // elements [tail_beg, e_hi) scalar
    if (tid < head_end - e_lo) gout[e_lo + tid] = elem_value(bits, e_lo + tid, layout, m0, m1, m2);
    if (tid < e_hi - tail_beg) gout[tail_beg + tid] = elem_value(bits, tail_beg + tid, layout, m0, m1, m2);
    float4* out4 = reinterpret_cast<float4*>(out);
    for (long long q = q_lo + tid; q < q_hi; q += BLOCK) {
        out4[q] = elem_group<LAYOUT>(bits, (int)(q * 4 - base), m0, m1, m2);
    }
}

}  // namespace

extern "C" int svx_rasterize(const int32_t* d_records, uint32_t n, float* d_out, int layout,
                             const float* mean, void* stream)
{
    if (n == 0) return SVX_OK;
    if (!d_records || !d_out || !mean) return SVX_EINVAL;
    if (layout != SVX_LAYOUT_NHWC && layout != SVX_LAYOUT_NCHW) return SVX_EINVAL;
    if ((reinterpret_cast<uintptr_t>(d_out) & 15u) != 0) return SVX_EINVAL;
    if ((uint64_t)n * 16 > 0x7fffffffull) return SVX_EINVAL;
    // enough workgroups to cover 256 CUs several times over even for one CNN batch
    int strips = 1;
    while (strips < 16 && (uint64_t)n * strips < 2048) strips *= 2;
    dim3 grid(n * strips), block(BLOCK);
    if (layout == SVX_LAYOUT_NCHW)
        hipLaunchKernelGGL(raster_kernel<SVX_LAYOUT_NCHW>, grid, block, 0, static_cast<hipStream_t>(stream),
                           d_records, n, d_out, strips, mean[0], mean[1], mean[2]);
    else
        hipLaunchKernelGGL(raster_kernel<SVX_LAYOUT_NHWC>, grid, block, 0, static_cast<hipStream_t>(stream),
                           d_records, n, d_out, strips, mean[0], mean[1], mean[2]);
    return hipGetLastError() == hipSuccess ? SVX_OK : SVX_ELAUNCH;
}
