// svx_cluster.hip -- pairwise signature distances of the clustering step on gfx950 (MI355X), fp64.
//
// cluster_partitions (reference src/collection/cluster_signatures.py:68-130) hands every partition to SciPy's
// linkage(data, method="average", metric=span_position_distance): pdist evaluates the Python callback :132-141 on
// all n (n - 1) / 2 pairs of double rows (tstart, tend, 1000).  Partitions are independent and so are pairs: one
// launch fills the condensed matrices of ALL partitions of a window, one lane per pair, in pdist order
// (row i, then j = i + 1 .. n - 1), so the result can be handed to linkage() unchanged.  HBM-write bound:
// 8 bytes per pair, coalesced; the 16 bytes per signature stay in L2.
//
// Arithmetic = the callback's, in IEEE doubles (the host compiles without fast-math; `/` on doubles is the
// correctly rounded division sequence): floor((s + e) / 2) centres, min of the three position differences over the
// normaliser, |span difference| / max(span) -- NaN when both spans are 0 (0 / 0), as NumPy gives it.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/svx.h"

namespace {

constexpr int BLOCK = 256;

__global__ __launch_bounds__(BLOCK)
void span_position_distance_kernel(const double* __restrict__ start, const double* __restrict__ end,
                                   const uint64_t* __restrict__ part_off, uint32_t n_parts,
                                   const uint64_t* __restrict__ out_off, double normalizer, double* __restrict__ out)
{
    const uint64_t total = out_off[n_parts];
    for (uint64_t g = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; g < total; g += (uint64_t)gridDim.x * BLOCK) {
        uint32_t lo = 0, hi = n_parts;                        // partition p with out_off[p] <= g < out_off[p + 1]
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (out_off[mid] <= g) lo = mid; else hi = mid; }
        const uint64_t base = part_off[lo], n = part_off[lo + 1] - base, k = g - out_off[lo];
        // row of the k-th pair: largest i with i * n - i (i + 1) / 2 <= k (sqrt guess, then exact integer fix-up)
        const double dn = (double)n;
        int64_t i = (int64_t)(dn - 0.5 - sqrt((dn - 0.5) * (dn - 0.5) - 2.0 * (double)k));
        if (i < 0) i = 0;
        if (i > (int64_t)n - 2) i = (int64_t)n - 2;
        while (i > 0 && (uint64_t)i * n - (uint64_t)i * (i + 1) / 2 > k) --i;
        while ((uint64_t)(i + 1) * n - (uint64_t)(i + 1) * (i + 2) / 2 <= k) ++i;
        const uint64_t j = k - ((uint64_t)i * n - (uint64_t)i * (i + 1) / 2) + i + 1;
        const double s1 = start[base + i], e1 = end[base + i], s2 = start[base + j], e2 = end[base + j];
        const double span1 = e1 - s1, span2 = e2 - s2;
        const double c1 = floor((s1 + e1) / 2.0), c2 = floor((s2 + e2) / 2.0);
        const double pos = fmin(fmin(fabs(s1 - s2), fabs(e1 - e2)), fabs(c1 - c2)) / normalizer;
        const double spd = fabs(span1 - span2) / (span1 > span2 ? span1 : span2);
        out[g] = pos + spd;
    }
}

}  // namespace

extern "C" int svx_span_position_distance(const double* d_start, const double* d_end, const uint64_t* d_part_off,
                                          uint32_t n_parts, const uint64_t* d_out_off, uint64_t total_pairs,
                                          double normalizer, double* d_out, void* stream)
{
    if (n_parts == 0 || total_pairs == 0) return SVX_OK;
    if (!d_start || !d_end || !d_part_off || !d_out_off || !d_out) return SVX_EINVAL;
    uint64_t blocks = (total_pairs + BLOCK - 1) / BLOCK;
    if (blocks > 256 * 32) blocks = 256 * 32;                 // grid-stride beyond 32 workgroups per CU
    hipLaunchKernelGGL(span_position_distance_kernel, dim3((unsigned)blocks), dim3(BLOCK), 0, static_cast<hipStream_t>(stream),
                       d_start, d_end, d_part_off, n_parts, d_out_off, normalizer, d_out);
    return hipGetLastError() == hipSuccess ? SVX_OK : SVX_ELAUNCH;
}
