// svx_raster_common.hpp -- bit-plane rasterisation of one segment pair into LDS, shared by
// svx_rasterize (dense batch tensor) and svx_encode_conv1 (sparse first convolution).
// Behaviour: PlotSingleImg.plot (reference src/segmentplot/plot_segment.py:33-68) with
// cv2.line = OpenCV clipLine + 8-connected LineIterator; Segment geometry as rebuilt by
// BatchGenerator (src/network/create_batch.py:118,132; src/segmentplot/classes.py:50-54).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/svx.h"

namespace svx_raster {


constexpr int IMG = SVX_IMG;
constexpr int ROW_WORDS = 8;                  // 227 bits -> 8 x u32 per row
constexpr int PLANE_WORDS = IMG * ROW_WORDS;  // 1816
constexpr int IMG_ELEMS = IMG * IMG * 3;      // 154,587 floats per image
constexpr int PLANE_ELEMS = IMG * IMG;        // 51,529

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


// Draw one record's three bit planes into LDS.  bits: 3 * PLANE_WORDS words (plane 0: all segments,
// plane 1: columns with >= 2 hits, plane 2: reverse segments), colcnt: IMG words, colmask: ROW_WORDS
// words.  Must be called by every thread of the block; ends with a barrier.
// PLANE1 = false: plane 1 is not materialised (it is plane 0 AND the column mask: a consumer that reads windows of a
// few columns applies the mask itself) and ``bits`` holds two planes only: [plane 0][plane 2].
template <int BLOCK_T, bool PLANE1 = true>
__device__ inline void draw_planes(const int32_t* __restrict__ r, unsigned* bits, unsigned* colcnt, unsigned* colmask)
{
    const int tid = threadIdx.x;
    constexpr int REV = PLANE1 ? 2 * PLANE_WORDS : PLANE_WORDS;       // word offset of the reverse-segment plane
    for (int i = tid; i < (PLANE1 ? 3 : 2) * PLANE_WORDS; i += BLOCK_T) bits[i] = 0;
    for (int i = tid; i < IMG; i += BLOCK_T) colcnt[i] = 0;
    if (tid < ROW_WORDS) colmask[tid] = 0;

    // two lanes derive the two line set-ups (fp64 scaling + clipLine) and publish them through LDS
    __shared__ Line sh_lines[2];
    __shared__ int sh_rev[2];
    if (tid < 2) {
        const int s = tid;
        const int read_len = r[10], ref_len = r[11];
        double ratio = (double)(read_len > ref_len ? read_len : ref_len) / 227.0;
        if (ratio < 1) ratio = 1;
        const int xs = r[s * 5 + 0], ys = r[s * 5 + 2];
        const long long len = (long long)r[s * 5 + 3] - (long long)ys;
        const int fwd = r[s * 5 + 4] != 0;
        const long long xe = fwd ? (long long)xs + (len - 1) : (long long)xs - (len - 1);
        const long long ye = (long long)ys + (len - 1);
        const long long cs = scale_coord(ys, ratio), rs = scale_coord(xs, ratio);
        const long long ce = (long long)((double)ye / ratio), re = (long long)((double)xe / ratio);
        sh_rev[s] = !fwd;
        sh_lines[s] = fwd ? setup_line(cs, rs, ce, re) : setup_line(ce, re, cs, rs);
    }
    __syncthreads();
    Line lines[2] = {sh_lines[0], sh_lines[1]};
    const int rev[2] = {sh_rev[0], sh_rev[1]};

    // draw: one lane per Bresenham step (<= 227 per line)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        for (int k = tid; k < lines[s].count; k += BLOCK_T) {
            int col, row;
            line_pixel(lines[s], k, col, row);
            const unsigned bit = 1u << (col & 31);
            const int w = row * ROW_WORDS + (col >> 5);
            const unsigned old = atomicOr(&bits[w], bit);
            if (!(old & bit)) atomicAdd(&colcnt[col], 1u);
            if (rev[s]) atomicOr(&bits[REV + w], bit);
        }
    }
    __syncthreads();
    for (int c = tid; c < IMG; c += BLOCK_T)
        if (colcnt[c] >= 2) atomicOr(&colmask[c >> 5], 1u << (c & 31));
    __syncthreads();
    if (PLANE1) {
        for (int i = tid; i < PLANE_WORDS; i += BLOCK_T) bits[PLANE_WORDS + i] = bits[i] & colmask[i & (ROW_WORDS - 1)];
        __syncthreads();
    }
}

}  // namespace svx_raster
