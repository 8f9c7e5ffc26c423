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

// NCHW group without a branch.  The three planes are ONE array of 681 rows x 256 bits: element e sits in row R = e / 227 (plane
// R / 227) at column c = e % 227, so the four elements of a group are four consecutive bits of a 64-bit window of that array -- 29
// bits farther on (256 - 227 unused bits per row) for the elements behind the end of a row, which continue in the next row, i.e.
// possibly in the next PLANE: the mean follows the row.  (Round 6.  The form above leaves the groups that straddle a row to the
// per-element path: one lane in 57, so practically every wave ran both paths -- ~125 vector instructions per 16 bytes where this
// takes ~45; with it a quarter of the waves keeps the memory system as busy, and FEWER waves in flight write faster:
// tools/exp/raster_bw.hip, profiles/r06_raster_bw.txt.)
__device__ inline float4 elem_group_rows(const unsigned* bits, int e, float m0, float m1, float m2)
{
    const int R = e / IMG, c = e - R * IMG;
    const int A = R * (32 * ROW_WORDS) + c, wi = A >> 5, sh = A & 31;
    const unsigned long long win = (((unsigned long long)bits[wi + 1] << 32) | bits[wi]) >> sh;
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bool wrap = c + i >= IMG;
        const unsigned b = (unsigned)(win >> (i + (wrap ? 32 * ROW_WORDS - IMG : 0))) & 1u;
        const int Ri = R + (wrap ? 1 : 0);
        const float mean = Ri < IMG ? m0 : (Ri < 2 * IMG ? m1 : m2);
        v[i] = b ? 255.0f - mean : -mean;
    }
    return make_float4(v[0], v[1], v[2], v[3]);
}

// NHWC: element e of the image is (pixel e / 3, channel e % 3).  After the planes are drawn they are INTERLEAVED once per image
// into `tri`: row r = 768 bits (24 words), bit 3 c + ch = plane ch at (r, c) -- 32 pixels of the three planes become three words
// through a 256-entry table that spreads a byte over every third bit -- so that the four elements of a group are again four
// consecutive bits of a 64-bit window; the (at most three) elements behind the end of a row are the first bits of the next row's
// first word.  681 % 3 == 0: the channel of element e + i is (e % 3 + i) % 3 whatever the row.
constexpr int TRI_ROW_WORDS = 3 * ROW_WORDS;                 // 24
constexpr int TRI_ROW_ELEMS = 3 * IMG;                       // 681 elements (bits) of a row in use
constexpr int TRI_WORDS = (IMG + 1) * TRI_ROW_WORDS;         // + one row: the look-ahead word of the last row

__device__ inline void interleave_planes(const unsigned* bits, unsigned* tri, const unsigned* spread, int tid)
{
    for (int u = tid; u < IMG * ROW_WORDS; u += BLOCK) {     // unit u: 32 pixels of row u / 8
        const unsigned p0 = bits[u], p1 = bits[PLANE_WORDS + u], p2 = bits[2 * PLANE_WORDS + u];
        unsigned long long t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            t[j] = (unsigned long long)(spread[(p0 >> (8 * j)) & 255u] | (spread[(p1 >> (8 * j)) & 255u] << 1) | (spread[(p2 >> (8 * j)) & 255u] << 2));
        const unsigned long long lo = t[0] | (t[1] << 24) | (t[2] << 48);          // bits 0..63 of the 96
        const unsigned hi = (unsigned)(t[2] >> 16) | (unsigned)(t[3] << 8);        // bits 64..95
        unsigned* o = tri + 3 * u;                             // (u = r * 8 + wj -> word r * 24 + 3 wj)
        o[0] = (unsigned)lo; o[1] = (unsigned)(lo >> 32); o[2] = hi;
    }
    if (tid < TRI_ROW_WORDS) tri[IMG * TRI_ROW_WORDS + tid] = 0;
}

__device__ inline float4 elem_group_tri(const unsigned* tri, int e, float m0, float m1, float m2)
{
    const int r = e / TRI_ROW_ELEMS, o = e - r * TRI_ROW_ELEMS;
    const int A = r * (32 * TRI_ROW_WORDS) + o, wi = A >> 5, sh = A & 31;
    const unsigned long long win = (((unsigned long long)tri[wi + 1] << 32) | tri[wi]) >> sh;
    const unsigned nx = tri[(r + 1) * TRI_ROW_WORDS];
    const int ch0 = e - 3 * (e / 3);
    const float ma = ch0 == 0 ? m0 : (ch0 == 1 ? m1 : m2), mb = ch0 == 0 ? m1 : (ch0 == 1 ? m2 : m0), mc = ch0 == 0 ? m2 : (ch0 == 1 ? m0 : m1);
    const float mean[4] = {ma, mb, mc, ma};
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int over = o + i - TRI_ROW_ELEMS;                // >= 0: the element is bit `over` of the next row
        const unsigned b = over >= 0 ? (nx >> over) & 1u : (unsigned)(win >> i) & 1u;
        v[i] = b ? 255.0f - mean[i] : -mean[i];
    }
    return make_float4(v[0], v[1], v[2], v[3]);
}

// A RESIDENT grid: workgroup w takes the work items (image, strip) w, w + grid, ...; a work item is one contiguous slice of
// one image.  NCHW launches run one workgroup per CU (svx_rasterize): a write-only stream is fastest with few waves in flight
// (a float4 fill of the same 1.27 GB: 6.5 TB/s from 256 workgroups, 5.3 from 2,048, 4.1 from 4,096), and the branch-free group
// above is cheap enough for four waves per CU to produce what the CU can write.
template <int LAYOUT>
__global__ __launch_bounds__(BLOCK)
void raster_kernel(const int32_t* __restrict__ records, uint32_t n, float* __restrict__ out,
                   int strips, float m0, float m1, float m2)
{
    constexpr int layout = LAYOUT;
    // plane 0: all segments, plane 1: columns with >= 2 hits, plane 2: reverse segments (+ 4 zero words: elem_group_rows reads
    // one word beyond the last row's)
    __shared__ unsigned bits[3 * PLANE_WORDS + 4];
    __shared__ unsigned colcnt[IMG];
    __shared__ unsigned colmask[ROW_WORDS];
    __shared__ unsigned tri[LAYOUT == SVX_LAYOUT_NHWC ? TRI_WORDS : 1];          // NHWC: the planes interleaved (elem_group_tri)
    __shared__ unsigned spread[LAYOUT == SVX_LAYOUT_NHWC ? 256 : 1];             // byte -> its bits at every third position
    const int tid = threadIdx.x;
    if (tid < 4) bits[3 * PLANE_WORDS + tid] = 0;
    if (LAYOUT == SVX_LAYOUT_NHWC) {
        unsigned sp = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) sp |= (((unsigned)tid >> k) & 1u) << (3 * k);
        spread[tid & 255] = sp;                                // (BLOCK == 256: one entry per thread; visible behind draw_planes' barriers)
    }
    const int per = ((IMG_ELEMS + strips - 1) / strips + 3) & ~3;
    const long long items = (long long)n * strips;
    int drawn = -1;
    for (long long item = blockIdx.x; item < items; item += gridDim.x) {
        const int img = (int)(item / strips), strip = (int)(item - (long long)img * strips);
        if (img != drawn) {
            __syncthreads();                                  // (the previous item's readers are through with the planes)
            draw_planes<BLOCK>(records + (size_t)img * 12, bits, colcnt, colmask);
            if (LAYOUT == SVX_LAYOUT_NHWC) { interleave_planes(bits, tri, spread, tid); __syncthreads(); }
            drawn = img;
        }
        // stream this item's slice [e_lo, e_hi) of the image: 16-byte stores aligned on the TENSOR (image bases are only 4-byte
        // aligned: 618,348 % 16 = 12), the elements in front of the first and behind the last full group one by one
        const int e_lo = strip * per;
        const int e_hi = min(IMG_ELEMS, e_lo + per);
        if (e_lo >= e_hi) continue;
        const long long base = (long long)img * IMG_ELEMS;   // global float index of element 0
        float* gout = out + base;
        const int skew = (int)((4 - (base & 3)) & 3);         // the image's first group boundary
        const int g_lo = e_lo <= skew ? skew : skew + ((e_lo - skew + 3) & ~3);
        const int g_hi = skew + ((e_hi - skew) & ~3);
        if (g_lo >= g_hi) {                                   // tiny slice: scalar only
            for (int e = e_lo + tid; e < e_hi; e += BLOCK) gout[e] = elem_value(bits, e, layout, m0, m1, m2);
            continue;
        }
        if (tid < g_lo - e_lo) gout[e_lo + tid] = elem_value(bits, e_lo + tid, layout, m0, m1, m2);
        if (tid < e_hi - g_hi) gout[g_hi + tid] = elem_value(bits, g_hi + tid, layout, m0, m1, m2);
        for (int e = g_lo + 4 * tid; e < g_hi; e += 4 * BLOCK)
            *reinterpret_cast<float4*>(gout + e) = LAYOUT == SVX_LAYOUT_NCHW ? elem_group_rows(bits, e, m0, m1, m2)
                                                                            : elem_group_tri(tri, e, m0, m1, m2);
    }
}

int device_cus()
{
    static int n = 0;
    if (n == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
        n = cus;
    }
    return n;
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
    // one resident workgroup per CU, an image cut into as many slices as it takes to give every CU one (a launch of one CNN
    // batch: 4 slices per image)
    const int resident = device_cus();
    int strips = 1;
    while (strips < 16 && (uint64_t)n * strips < (uint64_t)resident) strips *= 2;
    dim3 grid((unsigned)min((uint64_t)n * strips, (uint64_t)resident)), block(BLOCK);
    if (layout == SVX_LAYOUT_NCHW)
        hipLaunchKernelGGL(raster_kernel<SVX_LAYOUT_NCHW>, grid, block, 0, static_cast<hipStream_t>(stream),
                           d_records, n, d_out, strips, mean[0], mean[1], mean[2]);
    else
        hipLaunchKernelGGL(raster_kernel<SVX_LAYOUT_NHWC>, grid, block, 0, static_cast<hipStream_t>(stream),
                           d_records, n, d_out, strips, mean[0], mean[1], mean[2]);
    return hipGetLastError() == hipSuccess ? SVX_OK : SVX_ELAUNCH;
}
