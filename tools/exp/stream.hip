// Read-bandwidth experiments for the CIGAR count pass (not part of the product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ inline void tally(uint32_t w, int32_t min_sv, unsigned& ref_span, unsigned& qlen, unsigned& ngap)
{
    const uint32_t op = w & 15u, len = w >> 4;
    ref_span += len & (0u - ((0x18Du >> op) & 1u));
    qlen += len & (0u - ((0x1B3u >> op) & 1u));
    ngap += (uint32_t)((op - 1u) < 2u) & (uint32_t)((int32_t)len >= min_sv);
}

template <int UNROLL, bool TALLY>
__global__ __launch_bounds__(256) void stream_kernel(const uint4* __restrict__ in, size_t nq, uint32_t* out)
{
    unsigned r = 0, q = 0, g = 0;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < nq; i += UNROLL * stride) {
        uint4 w[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) w[u] = in[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (TALLY) { tally(w[u].x, 50, r, q, g); tally(w[u].y, 50, r, q, g); tally(w[u].z, 50, r, q, g); tally(w[u].w, 50, r, q, g); }
            else r ^= w[u].x ^ w[u].y ^ w[u].z ^ w[u].w;
        }
    }
    if ((r ^ q ^ g) == 0x12345678u) out[0] = r;
}

typedef float vf4 __attribute__((ext_vector_type(4)));
template <int NT>
__global__ __launch_bounds__(256) void write_kernel(const uint4* __restrict__ in, size_t nq, uint32_t* out)
{
    vf4* o = reinterpret_cast<vf4*>(const_cast<uint4*>(in));
    const size_t stride = (size_t)gridDim.x * 256;
    const vf4 v = {1.f, 2.f, 3.f, (float)blockIdx.x};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nq; i += stride) {
        if (NT) __builtin_nontemporal_store(v, &o[i]); else o[i] = v;
    }
}
// contiguous slice per workgroup (like the rasteriser: one image per workgroup)
template <int NT>
__global__ __launch_bounds__(256) void write_slice_kernel(const uint4* __restrict__ in, size_t nq, uint32_t* out)
{
    vf4* o = reinterpret_cast<vf4*>(const_cast<uint4*>(in));
    const size_t per = (nq + gridDim.x - 1) / gridDim.x;
    const size_t lo = per * blockIdx.x, hi = lo + per < nq ? lo + per : nq;
    const vf4 v = {1.f, 2.f, 3.f, (float)blockIdx.x};
    for (size_t i = lo + threadIdx.x; i < hi; i += 256) {
        if (NT) __builtin_nontemporal_store(v, &o[i]); else o[i] = v;
    }
}

// one burst per wave: every wave loads 8 KB (eight 16-byte loads per lane) once and leaves -- the shape of svx_cigar_scan_flat's tiles
template <int BLOCK, bool CONTIG, bool WSUM>
__global__ __launch_bounds__(BLOCK) void burst_kernel(const uint4* __restrict__ in, size_t nq, uint32_t* out)
{
    const size_t wave = ((size_t)blockIdx.x * BLOCK + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const size_t q0 = wave * 512;
    if (q0 + 512 > nq) return;
    uint4 w[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) w[u] = CONTIG ? in[q0 + 8 * lane + u] : in[q0 + 64 * u + lane];
    unsigned r = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) r ^= w[u].x ^ w[u].y ^ w[u].z ^ w[u].w;
    if (WSUM) { for (int o = 32; o > 0; o >>= 1) r += __shfl_xor(r, o, 64); }
    if (r == 0x12345678u) out[0] = r;
}

int main()
{
    const size_t bytes = 1200ull << 20;
    uint4* d; uint32_t* o;
    CK(hipMalloc(&d, bytes)); CK(hipMalloc(&o, 4));
    CK(hipMemset(d, 0x21, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t nq = bytes / 16;
    auto run = [&](const char* name, auto kern, int grid) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            for (int k = 0; k < 10; ++k) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, nq, o);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s grid %6d: %7.1f us  %6.2f TB/s\n", name, grid, ms * 100, bytes / (ms * 1e-4) / 1e12);
        return 0;
    };
    {
        auto runb = [&](const char* name, auto kern, int block) {
            const int grid = (int)((nq / 512) * 64 / block);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                for (int k = 0; k < 10; ++k) hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, d, nq, o);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%-34s grid %6d x %4d: %7.1f us  %6.2f TB/s\n", name, grid, block, ms * 100, bytes / (ms * 1e-4) / 1e12);
        };
        runb("burst coalesced", burst_kernel<256, false, false>, 256);
        runb("burst coalesced 512", burst_kernel<512, false, false>, 512);
        runb("burst lane-contiguous", burst_kernel<256, true, false>, 256);
        runb("burst lane-contiguous 512", burst_kernel<512, true, false>, 512);
        runb("burst coalesced + wave sum", burst_kernel<256, false, true>, 256);
        runb("burst coalesced 64", burst_kernel<64, false, false>, 64);
    }
    for (int grid : {2048, 4096, 16384}) {
        run("stream u1", stream_kernel<1, false>, grid);
        run("stream u2", stream_kernel<2, false>, grid);
        run("stream u4", stream_kernel<4, false>, grid);
        run("stream+tally u2", stream_kernel<2, true>, grid);
        run("stream+tally u4", stream_kernel<4, true>, grid);
    }
    {
        auto runb = [&](const char* name, auto kern, int block) {
            const int grid = (int)((nq / 512) * 64 / block);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                for (int k = 0; k < 10; ++k) hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, d, nq, o);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%-34s grid %6d x %4d: %7.1f us  %6.2f TB/s\n", name, grid, block, ms * 100, bytes / (ms * 1e-4) / 1e12);
        };
        runb("burst coalesced", burst_kernel<256, false, false>, 256);
        runb("burst coalesced 512", burst_kernel<512, false, false>, 512);
        runb("burst lane-contiguous", burst_kernel<256, true, false>, 256);
        runb("burst lane-contiguous 512", burst_kernel<512, true, false>, 512);
        runb("burst coalesced + wave sum", burst_kernel<256, false, true>, 256);
        runb("burst coalesced 64", burst_kernel<64, false, false>, 64);
    }
    for (int grid : {2048, 4096, 16384}) {
        run("write grid-stride", write_kernel<0>, grid);
        run("write grid-stride nontemporal", write_kernel<1>, grid);
        run("write slices", write_slice_kernel<0>, grid);
        run("write slices nontemporal", write_slice_kernel<1>, grid);
    }
    return 0;
}
