/* svx_experimental.h -- exports of libsvx.so that the DEFAULT path never calls.
 *
 * The product contract is include/svx.h.  What is declared here are earlier or alternative implementations of two of its
 * entry points, kept because their parity tests and A/B measurements are cheap to keep (tests/test_gpu_kernels.py,
 * tests/test_gpu_inflate.py, tools/bench_cigar.py) and because the numbers that retired them are quoted in DESIGN.md:
 *
 *   svx_cigar_scan_flat           one-pass chunked scan with a decoupled look-back: bit-identical to svx_cigar_scan, 3x slower
 *                                 on HiFi launches and slower on the ONT launch too (profiles/r05_*): never picked
 *                                 (kernels.FLAT_SCAN_FROM = None); reachable only through mode="flat" / SVX_SCAN_MODE=flat
 *   svx_bgzf_inflate{,_lds,_private}  the lane-per-block inflate of round 3 (60-125 ms per launch whatever it holds)
 *   svx_bgzf_inflate_wave         the wave-per-block single-kernel inflate (17 ms per 5,120 blocks)
 *   svx_bgzf_inflate_fast_lz      svx_bgzf_inflate_fast_on with the LZ kernel named by an argument (measurements; was the
 *                                 SVX_LZ environment variable, a process-global read on every call -- ADVICE r5)
 *
 * The default path inflates with svx_bgzf_inflate_fast_on (svx.h) and scans with svx_cigar_scan.  A maintainer binding the
 * library needs nothing from this file.
 */
#ifndef SVX_EXPERIMENTAL_H
#define SVX_EXPERIMENTAL_H
#include "svx.h"
#ifdef __cplusplus
extern "C" {
#endif

/* The same scan -- same inputs, same outputs bit for bit -- for LONG alignments (ONT ultra-long reads, assembly contigs: 10^3-10^6
 * operations each) in ONE pass: the flat array of words is cut into chunks of 2,048, one wave per chunk whatever alignment the
 * words belong to; the sums an alignment carries into a chunk and the number of long gaps in front of it come from a decoupled
 * look-back over the chunks in front (svx_cigar_flat.hip).  svx_cigar_scan walks an alignment of more than 512 words with one
 * wave, twice (0.17 of the HBM peak on an ONT-shaped launch); this form reads every word once with every wave of the chip.
 * Slower than svx_cigar_scan on short alignments (a chunk of HiFi reads holds a dozen boundaries): callers pick by the mean
 * number of words per alignment (svision_amd/kernels.py: >= 1,024).
 *   n_words_max  an upper bound of d_cig_off[n_aln] - d_cig_off[0] (the launch is sized by it; the offsets are device memory)
 *   d_ws         svx_cigar_scan_flat_ws_bytes(n_words_max) bytes, 8-byte aligned (SVX_EINVAL if ws_bytes is less) */
size_t svx_cigar_scan_flat_ws_bytes(uint64_t n_words_max);
int svx_cigar_scan_flat(const uint32_t* d_cigar, const uint64_t* d_cig_off, const int32_t* d_ref_start, uint32_t n_aln,
                        uint64_t n_words_max, int32_t min_sv, SvxGap* d_gaps, uint64_t gaps_cap, uint32_t* d_gap_off,
                        int32_t* d_stats, void* d_ws, uint64_t ws_bytes, void* stream);

/* The contract of svx_bgzf_inflate_fast (svx.h: d_comp / d_src_off / d_src_len / d_dst_off / d_out / d_status) without a
 * workspace, every block of a launch decoded by one lane, all blocks in parallel (svx_inflate.hip). */
int            svx_bgzf_inflate(const uint8_t* d_comp, const uint64_t* d_src_off, const uint32_t* d_src_len,
                                const uint64_t* d_dst_off, uint32_t n_blocks, uint8_t* d_out, uint32_t* d_status, void* stream);
/* svx_bgzf_inflate picks between two versions of the lane-per-block kernel by the size of the launch; by name:
 * _lds: the lane's symbol tables in LDS (420 B per lane: 98,304 blocks on the chip at once; 64-74 ms per round),
 * _private: the literal / length symbols in private memory (96 B of LDS per lane: 196,608 blocks at once; 68-125 ms),
 * which takes the launches the first would need two rounds for. */
int            svx_bgzf_inflate_lds(const uint8_t* d_comp, const uint64_t* d_src_off, const uint32_t* d_src_len,
                                    const uint64_t* d_dst_off, uint32_t n_blocks, uint8_t* d_out, uint32_t* d_status, void* stream);
int            svx_bgzf_inflate_private(const uint8_t* d_comp, const uint64_t* d_src_off, const uint32_t* d_src_len,
                                        const uint64_t* d_dst_off, uint32_t n_blocks, uint8_t* d_out, uint32_t* d_status, void* stream);
/* the same contract, one WAVE per block (uniform control flow; its time is proportional to the launch -- 17 ms per 5,120
 * blocks -- where the lane kernel needs 60+ ms for one block as for 98 k: the faster one below ~20 k blocks per launch) */
int            svx_bgzf_inflate_wave(const uint8_t* d_comp, const uint64_t* d_src_off, const uint32_t* d_src_len,
                                     const uint64_t* d_dst_off, uint32_t n_blocks, uint8_t* d_out, uint32_t* d_status, void* stream);

/* svx_bgzf_inflate_fast_on with the LZ kernel by name: lz_kernel 0 = by the size of the launch (what svx_bgzf_inflate_fast_on
 * does), 1 = one lane per block, 2 = one wave per block. */
int            svx_bgzf_inflate_fast_lz(const uint8_t* d_comp, const uint64_t* d_src_off, const uint32_t* d_src_len,
                                        const uint64_t* d_dst_off, uint32_t n_blocks, uint64_t inflated_bytes, uint8_t* d_out,
                                        uint32_t* d_status, void* d_ws, uint64_t ws_bytes, int lz_kernel, void* stream_tokens,
                                        void* stream_lz);

#ifdef __cplusplus
}
#endif
#endif
