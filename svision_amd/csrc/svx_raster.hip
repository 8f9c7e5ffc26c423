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
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/svx.h"

namespace {

constexpr int IMG = SVX_IMG;
constexpr int ROW_WORDS = 8;                  // 227 bits -> 8 x u32 per row
constexpr int PLANE_WORDS = IMG * ROW_WORDS;  // 1816
constexpr int IMG_ELEMS = IMG * IMG * 3;      // 154,587 floats per image
constexpr int PLANE_ELEMS = IMG * IMG;        // 51,529
constexpr int BLOCK = 256;

struct Line {
    int x0, y0;      // first pixel (left endpoint after the left-to-right swap)
    int dx, dy;      // major / minor extents (after the steep swap), both >= 0
    int sy;          // sign of the row step
    int steep;       // 1: rows are the major axis
    int count;       // pixels to draw, 0 when fully clipped
};

// OpenCV clipLine on a 227x227 image; all arithmetic as upstream: outcodes,
// rows first, intersection in double truncated toward zero, the second point's
// clip sees the first point already moved.
__device__ inline bool clip_line(long long& x1, long long& y1, long long& x2, long long& y2)
{
    const long long right = IMG - 1, bottom = IMG - 1;
    int c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8;
    int c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8;
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
        long long a;
        if (c1 & 12) {
            a = c1 < 8 ? 0 : bottom;
            x1 += (long long)((double)(a - y1) * (double)(x2 - x1) / (double)(y2 - y1));
            y1 = a;
            c1 = (x1 < 0) + (x1 > right) * 2;
        }
        if (c2 & 12) {
            a = c2 < 8 ? 0 : bottom;
            x2 += (long long)((double)(a - y2) * (double)(x2 - x1) / (double)(y2 - y1));
            y2 = a;
            c2 = (x2 < 0) + (x2 > right) * 2;
        }
        if ((c1 & c2) == 0 && (c1 | c2) != 0) {
            if (c1) {
                a = c1 == 1 ? 0 : right;
                y1 += (long long)((double)(a - x1) * (double)(y2 - y1) / (double)(x2 - x1));
                x1 = a;
                c1 = 0;
            }
            if (c2) {
                a = c2 == 1 ? 0 : right;
                y2 += (long long)((double)(a - x2) * (double)(y2 - y1) / (double)(x2 - x1));
                x2 = a;
                c2 = 0;
            }
        }
    }
    return (c1 | c2) == 0;
}

// LineIterator(img, pt1, pt2, 8, leftToRight = true) set-up.
__device__ inline Line setup_line(long long x1, long long y1, long long x2, long long y2)
{
    Line l;
    l.count = 0; l.x0 = l.y0 = l.dx = l.dy = l.steep = 0; l.sy = 1;
    if ((unsigned long long)x1 >= (unsigned long long)IMG || (unsigned long long)x2 >= (unsigned long long)IMG ||
        (unsigned long long)y1 >= (unsigned long long)IMG || (unsigned long long)y2 >= (unsigned long long)IMG) {
        if (!clip_line(x1, y1, x2, y2)) return l;
    }
    int dx = (int)(x2 - x1), dy = (int)(y2 - y1);
    if (dx < 0) { dx = -dx; dy = -dy; x1 = x2; y1 = y2; }
    l.sy = dy < 0 ? -1 : 1;
    if (dy < 0) dy = -dy;
    l.steep = dy > dx;
    if (l.steep) { int t = dx; dx = dy; dy = t; }
    l.x0 = (int)x1; l.y0 = (int)y1; l.dx = dx; l.dy = dy;
    l.count = dx + 1;
    return l;
}

// Pixel k of the walk.  The iterator draws, then does
//   mask = err < 0;  err += -2*dy + (mask ? 2*dx : 0);  major += 1;  minor += mask
// from err0 = dx - 2*dy.  The number of minor steps taken before pixel k is
//   m(k) = ceil((2*dy*k - dx) / (2*dx)) clamped at 0 = (2*dy*k + dx - 1) / (2*dx)
// (integer division; ties, err == 0, do not step).
__device__ inline void line_pixel(const Line& l, int k, int& col, int& row)
{
    int m = l.dx > 0 ? (2 * l.dy * k + l.dx - 1) / (2 * l.dx) : 0;
    if (l.steep) { row = l.y0 + l.sy * k; col = l.x0 + m; }
    else         { col = l.x0 + k;        row = l.y0 + l.sy * m; }
}

// C cast (long long)(double): truncation toward zero == Python int(float).
__device__ inline long long scale_coord(int v, double ratio) { return (long long)((double)v / ratio); }

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

__global__ __launch_bounds__(BLOCK)
void raster_kernel(const int32_t* __restrict__ records, uint32_t n, float* __restrict__ out,
                   int layout, int strips, float m0, float m1, float m2)
{
    // plane 0: all segments, plane 1: columns with >= 2 hits, plane 2: reverse segments
    __shared__ unsigned bits[3 * PLANE_WORDS];
    __shared__ unsigned colcnt[IMG];
    __shared__ unsigned colmask[ROW_WORDS];

    const uint32_t img = blockIdx.x / strips;
    const int strip = blockIdx.x - img * strips;
    const int tid = threadIdx.x;

    for (int i = tid; i < 3 * PLANE_WORDS; i += BLOCK) bits[i] = 0;
    if (tid < IMG) colcnt[tid] = 0;
    if (tid < ROW_WORDS) colmask[tid] = 0;

    // every lane derives the (wave-uniform) line set-ups itself: 12 ints, a few doubles
    const int32_t* r = records + (size_t)img * 12;
    const int read_len = r[10], ref_len = r[11];
    double ratio = (double)(read_len > ref_len ? read_len : ref_len) / 227.0;
    if (ratio < 1) ratio = 1;
    Line lines[2];
    int rev[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int xs = r[s * 5 + 0], ys = r[s * 5 + 2];
        const long long len = (long long)r[s * 5 + 3] - (long long)ys;
        const int fwd = r[s * 5 + 4] != 0;
        const long long xe = fwd ? (long long)xs + (len - 1) : (long long)xs - (len - 1);
        const long long ye = (long long)ys + (len - 1);
        const long long cs = scale_coord(ys, ratio), rs = scale_coord(xs, ratio);
        const long long ce = (long long)((double)ye / ratio), re = (long long)((double)xe / ratio);
        rev[s] = !fwd;
        lines[s] = fwd ? setup_line(cs, rs, ce, re) : setup_line(ce, re, cs, rs);
    }
    __syncthreads();

    // draw: one lane per Bresenham step (<= 227 per line)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        for (int k = tid; k < lines[s].count; k += BLOCK) {
            int col, row;
            line_pixel(lines[s], k, col, row);
            const unsigned bit = 1u << (col & 31);
            const int w = row * ROW_WORDS + (col >> 5);
            const unsigned old = atomicOr(&bits[w], bit);
            if (!(old & bit)) atomicAdd(&colcnt[col], 1u);
            if (rev[s]) atomicOr(&bits[2 * PLANE_WORDS + w], bit);
        }
    }
    __syncthreads();
    if (tid < IMG && colcnt[tid] >= 2) atomicOr(&colmask[tid >> 5], 1u << (tid & 31));
    __syncthreads();
    for (int i = tid; i < PLANE_WORDS; i += BLOCK) bits[PLANE_WORDS + i] = bits[i] & colmask[i & (ROW_WORDS - 1)];
    __syncthreads();

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
        const int e = (int)(q * 4 - base);
        float4 v;
        v.x = elem_value(bits, e + 0, layout, m0, m1, m2);
        v.y = elem_value(bits, e + 1, layout, m0, m1, m2);
        v.z = elem_value(bits, e + 2, layout, m0, m1, m2);
        v.w = elem_value(bits, e + 3, layout, m0, m1, m2);
        out4[q] = v;
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
    hipLaunchKernelGGL(raster_kernel, grid, block, 0, static_cast<hipStream_t>(stream),
                       d_records, n, d_out, layout, strips, mean[0], mean[1], mean[2]);
    return hipGetLastError() == hipSuccess ? SVX_OK : SVX_ELAUNCH;
}
