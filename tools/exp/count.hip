// Times the product's count / emit kernels in isolation (experiment harness; includes the product source).
#include "../../svision_amd/csrc/svx_cigar.hip"
#include <cstdio>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv)
{
    const uint32_t n_aln = argc > 1 ? atoi(argv[1]) : 2000000;
    const int mean = argc > 2 ? atoi(argv[2]) : 150;
    std::mt19937 rng(1);
    std::poisson_distribution<int> pois(mean);
    std::vector<uint64_t> off(n_aln + 1, 0);
    for (uint32_t i = 0; i < n_aln; ++i) off[i + 1] = off[i] + std::max(1, pois(rng));
    const size_t words = off[n_aln];
    std::vector<uint32_t> cig(words);
    for (auto& w : cig) { uint32_t r = rng(); w = ((r >> 8) % 40 + 1) << 4 | ((r & 7) == 0 ? 1 : (r & 7) == 1 ? 2 : (r & 7) == 2 ? 8 : 7); }
    for (uint32_t i = 0; i < n_aln; ++i) {
        if (off[i + 1] - off[i] < 3) continue;
        if (rng() % 5 < 2) cig[off[i]] = (rng() % 20000 + 1) << 4 | 4;
        if (rng() % 5 < 2) cig[off[i + 1] - 1] = (rng() % 20000 + 1) << 4 | 5;
        if (rng() % 2000 == 0) cig[off[i] + 1] = (rng() % 3000 + 50) << 4 | 1;
    }
    std::vector<int32_t> pos(n_aln, 1000);
    uint32_t *d_c, *d_g, *d_ws; uint64_t* d_o; int32_t *d_s, *d_p; SvxGap* d_gaps;
    const size_t ws = svx_cigar_scan_ws_bytes(n_aln);
    CK(hipMalloc(&d_c, words * 4)); CK(hipMalloc(&d_o, (n_aln + 1) * 8)); CK(hipMalloc(&d_g, (n_aln + 1) * 4)); CK(hipMalloc(&d_s, n_aln * 16));
    CK(hipMalloc(&d_p, n_aln * 4)); CK(hipMalloc(&d_ws, ws)); CK(hipMalloc(&d_gaps, sizeof(SvxGap) << 20));
    CK(hipMemcpy(d_c, cig.data(), words * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_o, off.data(), (n_aln + 1) * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_p, pos.data(), n_aln * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = words * 4.0 + 32.0 * n_aln;
    const uint32_t tiles = (n_aln + TILE - 1) >> TILE_SHIFT;
    float ms;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        for (int k = 0; k < 10; ++k) {
            hipLaunchKernelGGL(count_kernel, dim3((n_aln + ALN_PER_CBLOCK - 1) / ALN_PER_CBLOCK), dim3(BLOCK), 0, 0, d_c, d_o, n_aln, 50, d_g, d_s, reinterpret_cast<uint2*>(d_ws));
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    hipEventElapsedTime(&ms, e0, e1);
    printf("count_kernel G%d Q%d %u x %d: %7.1f us  %5.2f TB/s\n", CGROUP, CQUADS, n_aln, mean, ms * 100, bytes / (ms * 1e-4) / 1e12);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        for (int k = 0; k < 10; ++k) svx_cigar_scan(d_c, d_o, d_p, n_aln, 0, 50, d_gaps, 1 << 20, d_g, d_s, d_ws, nullptr);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    hipEventElapsedTime(&ms, e0, e1);
    uint32_t total = 0; CK(hipMemcpy(&total, d_g + n_aln, 4, hipMemcpyDeviceToHost));
    printf("svx_cigar_scan %u x %d: %7.1f us  %5.2f TB/s  (%u gaps)\n", n_aln, mean, ms * 100, (bytes + 24.0 * total) / (ms * 1e-4) / 1e12, total);
    return 0;
}
