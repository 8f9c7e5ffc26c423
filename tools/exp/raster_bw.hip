// raster_bw.hip -- how close is raster_kernel to what the chip WRITES?  
// Build (cross-compiles here, runs on the GPU box):  hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/exp/raster_bw tools/exp/raster_bw.hip
// Prints microseconds and TB/s for: hipMemsetAsync, a float4 fill kernel with the raster kernel's launch shape (plain and nontemporal
// stores), svx_rasterize as shipped (NCHW / NHWC), and variants of its store loop -- all on the same n images (default 2048 =
// 1.27 GB written), median of REPS launches timed with HIP events.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "../../svision_amd/csrc/svx_raster.hip"

using namespace svx_raster;
typedef float vf4 __attribute__((ext_vector_type(4)));
__device__ inline void nt_store(float4* p, const float4& v) { __builtin_nontemporal_store(vf4{v.x, v.y, v.z, v.w}, reinterpret_cast<vf4*>(p)); }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <bool NT>
__global__ __launch_bounds__(256) void fill_kernel(float4* out, long long per_wg, long long total)
{
    const long long lo = (long long)blockIdx.x * per_wg, hi = min(total, lo + per_wg);
    const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    for (long long q = lo + threadIdx.x; q < hi; q += 256) {
        if (NT) nt_store(&out[q], v); else out[q] = v;
    }
}

// memset-like: every thread of the (small, resident) grid strides over the whole tensor -- the chip writes ONE moving window
template <int U>
__global__ __launch_bounds__(256) void fill_stride_kernel(float4* out, long long total)
{
    const long long nthreads = (long long)gridDim.x * 256;
    const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total; q += nthreads * U) {
#pragma unroll
        for (int u = 0; u < U; ++u) if (q + u * nthreads < total) out[q + u * nthreads] = v;
    }
}
// every lane writes 64 contiguous bytes (4 x 16 B), a wave 4 KB, per iteration
__global__ __launch_bounds__(256) void fill_lane64_kernel(float4* out, long long per_wg, long long total)
{
    const long long lo = (long long)blockIdx.x * per_wg, hi = min(total, lo + per_wg);
    const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    for (long long q = lo + threadIdx.x * 4; q < hi; q += 1024) {
#pragma unroll
        for (int u = 0; u < 4; ++u) if (q + u < hi) out[q + u] = v;
    }
}

// NCHW group without a branch: the three planes are one array of 681 rows x 256 bits, element e sits in row R = e / 227 at column
// c = e % 227; the four elements of a group are four consecutive bits of a 64-bit window -- 29 bits farther on where the group
// runs over the end of a row (256 - 227 unused bits per row; the next row may be the next PLANE: the mean follows R)
__device__ inline float4 group_nchw_bf(const unsigned* bits, int e, float m0, float m1, float m2)
{
    const int R = e / IMG, c = e - R * IMG;
    const int A = R * 256 + c, wi = A >> 5, sh = A & 31;
    const unsigned long long win = (((unsigned long long)bits[wi + 1] << 32) | bits[wi]) >> sh;
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bool wrap = c + i >= IMG;
        const unsigned b = (unsigned)(win >> (i + (wrap ? 29 : 0))) & 1u;
        const int Ri = R + (wrap ? 1 : 0);
        const float mean = Ri < IMG ? m0 : (Ri < 2 * IMG ? m1 : m2);
        v[i] = b ? 255.0f - mean : -mean;
    }
    return make_float4(v[0], v[1], v[2], v[3]);
}

template <int BLK>
__global__ __launch_bounds__(BLK) void raster_bf(const int32_t* __restrict__ records, uint32_t n, float* __restrict__ out,
                                                 int strips, float m0, float m1, float m2)
{
    __shared__ unsigned bits[3 * PLANE_WORDS + 4];
    __shared__ unsigned colcnt[IMG];
    __shared__ unsigned colmask[ROW_WORDS];
    const int tid = threadIdx.x;
    if (tid < 4) bits[3 * PLANE_WORDS + tid] = 0;
    const int per = ((IMG_ELEMS + strips - 1) / strips + 3) & ~3;
    int drawn = -1;
    for (long long item = blockIdx.x; item < (long long)n * strips; item += gridDim.x) {
        const int img = (int)(item / strips), strip = (int)(item - (long long)img * strips);
        if (img != drawn) { __syncthreads(); draw_planes<BLK>(records + (size_t)img * 12, bits, colcnt, colmask); drawn = img; }
        const int e_lo = strip * per, e_hi = min(IMG_ELEMS, e_lo + per);
        if (e_lo >= e_hi) continue;
        const long long base = (long long)img * IMG_ELEMS;
        float* gout = out + base;
        // float4 groups aligned on the tensor: the image's first group starts at element `skew` = (-base) mod 4
        const int skew = (int)((4 - (base & 3)) & 3);
        const int g_lo = e_lo <= skew ? skew : skew + ((e_lo - skew + 3) & ~3);       // first group boundary >= e_lo
        const int g_hi = skew + ((e_hi - skew) & ~3);                                  // last group boundary <= e_hi
        if (g_lo >= g_hi) { for (int e = e_lo + tid; e < e_hi; e += BLK) gout[e] = elem_value(bits, e, SVX_LAYOUT_NCHW, m0, m1, m2); continue; }
        if (tid < g_lo - e_lo) gout[e_lo + tid] = elem_value(bits, e_lo + tid, SVX_LAYOUT_NCHW, m0, m1, m2);
        if (tid < e_hi - g_hi) gout[g_hi + tid] = elem_value(bits, g_hi + tid, SVX_LAYOUT_NCHW, m0, m1, m2);
        for (int e = g_lo + 4 * tid; e < g_hi; e += 4 * BLK)
            *reinterpret_cast<float4*>(gout + e) = group_nchw_bf(bits, e, m0, m1, m2);
    }
}

// a resident grid whose workgroups take CHUNKS (of `chunk4` float4) in turn: chunk = an image -> as many streams as workgroups
__global__ __launch_bounds__(256) void fill_chunks_kernel(float4* out, long long chunk4, long long total)
{
    const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    const long long n_chunks = (total + chunk4 - 1) / chunk4;
    for (long long c = blockIdx.x; c < n_chunks; c += gridDim.x) {
        const long long lo = c * chunk4, hi = min(total, lo + chunk4);
        for (long long q = lo + threadIdx.x; q < hi; q += 256) out[q] = v;
    }
}

// the shipped kernel's body as a RESIDENT grid: workgroup w takes the work items (image, strip) w, w + grid, ... -- consecutive
// workgroups write consecutive strips of one image
template <int LAYOUT>
__global__ __launch_bounds__(256) void raster_persist(const int32_t* __restrict__ records, uint32_t n, float* __restrict__ out,
                                                      int strips, float m0, float m1, float m2)
{
    __shared__ unsigned bits[3 * PLANE_WORDS];
    __shared__ unsigned colcnt[IMG];
    __shared__ unsigned colmask[ROW_WORDS];
    const int tid = threadIdx.x;
    const int per = ((IMG_ELEMS + strips - 1) / strips + 3) & ~3;
    int drawn = -1;
    for (long long item = blockIdx.x; item < (long long)n * strips; item += gridDim.x) {
        const int img = (int)(item / strips), strip = (int)(item - (long long)img * strips);
        if (img != drawn) { __syncthreads(); draw_planes<256>(records + (size_t)img * 12, bits, colcnt, colmask); drawn = img; }
        const int e_lo = strip * per, e_hi = min(IMG_ELEMS, e_lo + per);
        if (e_lo >= e_hi) continue;
        const long long base = (long long)img * IMG_ELEMS;
        float* gout = out + base;
        const long long g_lo = base + e_lo, g_hi = base + e_hi;
        const long long q_lo = (g_lo + 3) >> 2, q_hi = g_hi >> 2;
        if (q_lo >= q_hi) { for (int e = e_lo + tid; e < e_hi; e += 256) gout[e] = elem_value(bits, e, LAYOUT, m0, m1, m2); continue; }
        const int head_end = (int)(q_lo * 4 - base), tail_beg = (int)(q_hi * 4 - base);
        if (tid < head_end - e_lo) gout[e_lo + tid] = elem_value(bits, e_lo + tid, LAYOUT, m0, m1, m2);
        if (tid < e_hi - tail_beg) gout[tail_beg + tid] = elem_value(bits, tail_beg + tid, LAYOUT, m0, m1, m2);
        float4* out4 = reinterpret_cast<float4*>(out);
        for (long long q = q_lo + tid; q < q_hi; q += 256) out4[q] = elem_group<LAYOUT>(bits, (int)(q * 4 - base), m0, m1, m2);
    }
}

// the shipped kernel's body with the store loop as a parameter: NT = nontemporal stores, U = float4 groups per thread and iteration
template <int LAYOUT, bool NT, int U>
__global__ __launch_bounds__(256) void raster_variant(const int32_t* __restrict__ records, uint32_t n, float* __restrict__ out,
                                                      int strips, float m0, float m1, float m2)
{
    __shared__ unsigned bits[3 * PLANE_WORDS];
    __shared__ unsigned colcnt[IMG];
    __shared__ unsigned colmask[ROW_WORDS];
    const uint32_t img = blockIdx.x / strips;
    const int strip = blockIdx.x - img * strips;
    const int tid = threadIdx.x;
    draw_planes<256>(records + (size_t)img * 12, bits, colcnt, colmask);
    const int per = (IMG_ELEMS + strips - 1) / strips;
    const int e_lo = strip * per;
    const int e_hi = min(IMG_ELEMS, e_lo + per);
    if (e_lo >= e_hi) return;
    const long long base = (long long)img * IMG_ELEMS;
    float* gout = out + base;
    const long long g_lo = base + e_lo, g_hi = base + e_hi;
    const long long q_lo = (g_lo + 3) >> 2, q_hi = g_hi >> 2;
    if (q_lo >= q_hi) { for (int e = e_lo + tid; e < e_hi; e += 256) gout[e] = elem_value(bits, e, LAYOUT, m0, m1, m2); return; }
    const int head_end = (int)(q_lo * 4 - base), tail_beg = (int)(q_hi * 4 - base);
    if (tid < head_end - e_lo) gout[e_lo + tid] = elem_value(bits, e_lo + tid, LAYOUT, m0, m1, m2);
    if (tid < e_hi - tail_beg) gout[tail_beg + tid] = elem_value(bits, tail_beg + tid, LAYOUT, m0, m1, m2);
    float4* out4 = reinterpret_cast<float4*>(out);
    for (long long q0 = q_lo; q0 < q_hi; q0 += 256 * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const long long q = q0 + u * 256 + tid; if (q < q_hi) v[u] = elem_group<LAYOUT>(bits, (int)(q * 4 - base), m0, m1, m2); }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long q = q0 + u * 256 + tid;
            if (q < q_hi) { if (NT) nt_store(&out4[q], v[u]); else out4[q] = v[u]; }
        }
    }
}

template <typename F>
double median_us(F&& launch, int reps)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    std::vector<float> t;
    launch(); CK(hipDeviceSynchronize());
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms * 1e3f);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main(int argc, char** argv)
{
    const uint32_t n = argc > 1 ? (uint32_t)atoi(argv[1]) : 2048;
    const int reps = argc > 2 ? atoi(argv[2]) : 21;
    const size_t floats = (size_t)n * IMG_ELEMS, bytes = floats * 4;
    float* out; CK(hipMalloc(&out, bytes + 64));
    std::vector<int32_t> rec((size_t)n * 12);
    srand(1);
    for (uint32_t i = 0; i < n; ++i) {                     // plausible segment pairs (TSV columns 1..12)
        int32_t* r = &rec[(size_t)i * 12];
        const int rl = 8000 + rand() % 8000, fl = 8000 + rand() % 8000;
        r[0] = rand() % (rl / 2); r[1] = r[0] + rl / 3; r[2] = rand() % (fl / 2); r[3] = r[2] + fl / 3; r[4] = rand() & 1;
        r[5] = rl / 2 + rand() % (rl / 3); r[6] = r[5] + rl / 8; r[7] = fl / 2 + rand() % (fl / 3); r[8] = r[7] + fl / 8; r[9] = rand() & 1;
        r[10] = rl; r[11] = fl;
    }
    int32_t* d_rec; CK(hipMalloc(&d_rec, rec.size() * 4)); CK(hipMemcpy(d_rec, rec.data(), rec.size() * 4, hipMemcpyHostToDevice));
    const float mean[3] = {104.f, 117.f, 124.f};
    auto report = [&](const char* what, double us) { printf("%-58s %9.1f us  %6.3f TB/s  %.3f of 8 TB/s\n", what, us, bytes / us / 1e6, bytes / us / 1e6 / 8.0); fflush(stdout); };
    printf("n = %u images, %.3f GB written per launch, median of %d\n", n, bytes / 1e9, reps);
    report("hipMemsetAsync", median_us([&] { CK(hipMemsetAsync(out, 0, bytes, 0)); }, reps));
    const long long total4 = (long long)(floats / 4);
    for (int wgs : {2048}) {
        const long long per = (total4 + wgs - 1) / wgs;
        char nm[96];
        snprintf(nm, sizeof nm, "fill float4, %d workgroups x 256", wgs);
        report(nm, median_us([&] { hipLaunchKernelGGL(fill_kernel<false>, dim3(wgs), dim3(256), 0, 0, (float4*)out, per, total4); }, reps));
        snprintf(nm, sizeof nm, "fill float4 nontemporal, %d workgroups x 256", wgs);
        report(nm, median_us([&] { hipLaunchKernelGGL(fill_kernel<true>, dim3(wgs), dim3(256), 0, 0, (float4*)out, per, total4); }, reps));
    }
    for (int wgs : {256, 512, 1024, 2048, 4096}) {
        char nm[96];
        snprintf(nm, sizeof nm, "fill grid-stride (memset-like), %d workgroups x 256", wgs);
        report(nm, median_us([&] { hipLaunchKernelGGL(fill_stride_kernel<1>, dim3(wgs), dim3(256), 0, 0, (float4*)out, total4); }, reps));
        snprintf(nm, sizeof nm, "fill grid-stride unroll 4, %d workgroups x 256", wgs);
        report(nm, median_us([&] { hipLaunchKernelGGL(fill_stride_kernel<4>, dim3(wgs), dim3(256), 0, 0, (float4*)out, total4); }, reps));
    }
    for (int wgs : {2048, 16384}) {
        const long long per = ((total4 + wgs - 1) / wgs + 1023) / 1024 * 1024;
        char nm[96];
        snprintf(nm, sizeof nm, "fill 64 B per lane, %d workgroups x 256", wgs);
        report(nm, median_us([&] { hipLaunchKernelGGL(fill_lane64_kernel, dim3(wgs), dim3(256), 0, 0, (float4*)out, per, total4); }, reps));
    }
    for (int wgs : {256, 512}) for (long long chunk : {(long long)IMG_ELEMS / 4 + 1, 2432LL, 256LL}) {
        char nm[96];
        snprintf(nm, sizeof nm, "fill chunks of %lld B in turn, %d resident workgroups", chunk * 16, wgs);
        report(nm, median_us([&] { hipLaunchKernelGGL(fill_chunks_kernel, dim3(wgs), dim3(256), 0, 0, (float4*)out, chunk, total4); }, reps));
    }
    for (int wgs : {256}) for (int S : {1}) {
        char nm[96];
        snprintf(nm, sizeof nm, "raster resident NCHW, %d workgroups, %d strips per image", wgs, S);
        report(nm, median_us([&] { hipLaunchKernelGGL((raster_persist<SVX_LAYOUT_NCHW>), dim3(wgs), dim3(256), 0, 0, d_rec, n, out, S, mean[0], mean[1], mean[2]); }, reps));
    }
    {   // the branch-free form == the shipped kernel, bit for bit
        std::vector<float> a(floats), b(floats);
        svx_rasterize(d_rec, n, out, SVX_LAYOUT_NCHW, mean, nullptr); CK(hipMemcpy(a.data(), out, bytes, hipMemcpyDeviceToHost));
        CK(hipMemset(out, 0xff, bytes));
        hipLaunchKernelGGL((raster_bf<256>), dim3(777), dim3(256), 0, 0, d_rec, n, out, 3, mean[0], mean[1], mean[2]); CK(hipMemcpy(b.data(), out, bytes, hipMemcpyDeviceToHost));
        printf("branch-free NCHW == shipped: %s\n", memcmp(a.data(), b.data(), bytes) == 0 ? "yes" : "NO");
    }
    for (int wgs : {256, 512, 2048}) for (int S : {1, 2}) {
        char nm[96];
        snprintf(nm, sizeof nm, "raster branch-free NCHW, %d workgroups, %d strips per image", wgs, S);
        report(nm, median_us([&] { hipLaunchKernelGGL((raster_bf<256>), dim3(wgs), dim3(256), 0, 0, d_rec, n, out, S, mean[0], mean[1], mean[2]); }, reps));
    }
    for (int wgs : {128, 256, 512}) for (int S : {1, 2}) {
        char nm[96];
        snprintf(nm, sizeof nm, "raster branch-free NCHW 512 threads, %d workgroups, %d strips", wgs, S);
        report(nm, median_us([&] { hipLaunchKernelGGL((raster_bf<512>), dim3(wgs), dim3(512), 0, 0, d_rec, n, out, S, mean[0], mean[1], mean[2]); }, reps));
        snprintf(nm, sizeof nm, "raster branch-free NCHW 1024 threads, %d workgroups, %d strips", wgs, S);
        report(nm, median_us([&] { hipLaunchKernelGGL((raster_bf<1024>), dim3(wgs), dim3(1024), 0, 0, d_rec, n, out, S, mean[0], mean[1], mean[2]); }, reps));
    }
    report("svx_rasterize NCHW (shipped)", median_us([&] { svx_rasterize(d_rec, n, out, SVX_LAYOUT_NCHW, mean, nullptr); }, reps));
    report("svx_rasterize NHWC (shipped)", median_us([&] { svx_rasterize(d_rec, n, out, SVX_LAYOUT_NHWC, mean, nullptr); }, reps));
#define VAR(L_, NT_, U_, S_) { char nm[96]; snprintf(nm, sizeof nm, "variant %s nt=%d unroll=%d strips=%d", #L_, NT_, U_, S_); \
    report(nm, median_us([&] { hipLaunchKernelGGL((raster_variant<SVX_LAYOUT_##L_, NT_, U_>), dim3(n * S_), dim3(256), 0, 0, d_rec, n, out, S_, mean[0], mean[1], mean[2]); }, reps)); }
    VAR(NCHW, false, 1, 1) VAR(NCHW, true, 1, 1) VAR(NCHW, false, 2, 1) VAR(NCHW, true, 2, 1) VAR(NCHW, false, 4, 1) VAR(NCHW, true, 4, 1)
    VAR(NCHW, false, 1, 2) VAR(NCHW, true, 1, 2) VAR(NCHW, true, 2, 2) VAR(NCHW, true, 2, 4) VAR(NCHW, true, 4, 4) VAR(NCHW, false, 4, 4)
    VAR(NHWC, false, 1, 1) VAR(NHWC, true, 1, 1) VAR(NHWC, true, 2, 1) VAR(NHWC, true, 4, 2)
    return 0;
}
