/* svx.h -- C ABI of libsvx.so, the MI355X (gfx950) hot path of SVision.
 *
 * The reference (xjtu-omics/SVision v1.4) is pure Python and has no FFI layer;
 * each entry point below replaces one Python hot loop and is bound from the
 * Python host with ctypes (svision_amd/_lib.py; INTEGRATION.md shows the stub a
 * reference maintainer would add).  Conventions:
 *   - every pointer named d_* is a DEVICE pointer owned by the caller
 *     (e.g. a torch tensor's data_ptr()); nothing is allocated or freed inside;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all
 *     work is enqueued asynchronously on it, no host synchronisation inside;
 *   - return value: 0 on success, a negative SVX_E* code otherwise; never throws;
 *   - re-entrant; one process per GPU.
 */
#ifndef SVX_H
#define SVX_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVX_VERSION 420            /* 0.3.0: + svx_bgzf_crc32, svx_bgzf_inflate_fast (360); + svx_bgzf_inflate_fast_on: the two kernels on two streams (370);
                                    * svx_bgzf_inflate_fast / _on take the inflated byte count and refuse a workspace that is too small (380);
                                    * + svx_cigar_scan_flat: the scan of long alignments in one pass (390); svx_cigar_scan takes the word count of its launch, its workspace's size and shape flags (400);
                                    * the exports the default path never calls moved to svx_experimental.h, + svx_bgzf_inflate_fast_lz there (410);
                                    * svx_bam_walk_extract takes the number of records (a wave per record copies: ONT-shaped slices 14.7 -> ~1 ms) (420) */

#define SVX_OK            0
#define SVX_EINVAL       (-1)      /* bad argument (null pointer, bad layout...) */
#define SVX_ECAPACITY    (-2)      /* caller-provided output capacity too small */
#define SVX_ELAUNCH      (-3)      /* HIP reported a launch error */

#define SVX_IMG           227      /* similarity image is 227 x 227 x 3 */
#define SVX_LAYOUT_NHWC   0        /* [n][227][227][3]  (reference batch layout) */
#define SVX_LAYOUT_NCHW   1        /* [n][3][227][227]  (planar, for the CNN)    */

#define SVX_GAP_INS       1
#define SVX_GAP_DEL       2

#define SVX_CONV_DENSE_PCT 97      /* a pixel list this full (percent) is not followed: svx_conv2d_same computes every pixel */

/* "C8" activation layout of the CNN entry points: float32 [image][C/8][H][W][8] -- channel c of pixel (y, x) lives at
 * ((image * C/8 + c/8) * H*W + y*W + x) * 8 + c%8, so the 8 channels of an octet are one 32-byte sector per pixel
 * (C % 8 == 0).  It lets every MFMA operand fetch and every epilogue store of svx_conv2d_same be a 16-byte access of
 * whole sectors, for dense pixel ranges and for gathered (active-set) pixels alike. */

/* One long CIGAR gap (I or D with length >= min_sv), 24 bytes.
 * Replaces the entries of `all_long_gaps` built at
 * reference src/collection/analyze_reads.py:828-853. */
typedef struct SvxGap {
    uint32_t aln;        /* alignment index in the batch                       */
    uint32_t op;         /* index of the op inside that alignment's CIGAR      */
    int32_t  read_pos;   /* readPos before the op (leading clips included)     */
    int32_t  ref_pos;    /* refPos before the op                               */
    int32_t  len;        /* op length                                          */
    uint32_t kind;       /* SVX_GAP_INS / SVX_GAP_DEL                          */
} SvxGap;

int         svx_version(void);
const char* svx_strerror(int code);
/* host: crc32c (Castagnoli) of a buffer -- the tensor / block checksums of the -m checkpoint (predict.py:181-184's
 * Saver().restore verifies them inside TensorFlow) */
uint32_t    svx_crc32c(const void* data, size_t n);

#define SVX_SCAN_FAILED   0xFFFFFFFFu

/* Bytes of device scratch svx_cigar_scan needs for n_aln alignments of n_words CIGAR words (the frame records of its long
 * alignments and the map of their launch take n_words * 5 / 128 bytes of it). */
size_t svx_cigar_scan_ws_bytes(uint32_t n_aln, uint64_t n_words);

/* flags of svx_cigar_scan: 0 = the count pass's shape follows the launch's mean words per alignment; the bits fix it (A/B runs,
 * tests) -- never a result */
#define SVX_SCAN_LANES4   1u       /* four lanes per alignment (default up to 256 words per alignment) */
#define SVX_SCAN_LANES8   2u       /* eight */
#define SVX_SCAN_SHARED   4u       /* the frames of long alignments in a launch of their own, equal ranges of the array (default from 1,024 words per alignment) */
#define SVX_SCAN_UNSHARED 8u       /* a long alignment is finished by its own wave */

/* Per-alignment CIGAR / segment scan.
 * Replaces the Python loop of analyze_inside_align
 * (reference src/collection/analyze_reads.py:828-853) and the pysam-derived
 * reference_end / query_alignment_start / query_alignment_end
 * (src/collection/analyze_reads.py:650-667) for a whole batch of alignments.
 *
 *   d_cigar     packed BAM CIGAR words (len << 4 | op) of all alignments, concatenated;
 *               16-byte aligned (read in quads), d_cig_off[n_aln] words long
 *   d_cig_off   [n_aln + 1] word offsets into d_cigar (CSR)
 *   d_ref_start [n_aln] 0-based reference start of each alignment
 *   min_sv      options.min_sv_size
 *   d_gaps      [gaps_cap] out: long gaps sorted by (aln, op)
 *   d_gap_off   [n_aln + 1] out, 16-byte aligned: CSR offsets into d_gaps per alignment
 *               (d_gap_off[n_aln] = total number of long gaps, even when it
 *               exceeds gaps_cap; gaps beyond gaps_cap are not written;
 *               SVX_SCAN_FAILED = the offsets pass gave up waiting for a tile in front of it
 *               -- workgroups are dispatched in index order on this hardware, so this is a
 *               bug or a different dispatcher, never data --: the other outputs are invalid)
 *   d_stats     [n_aln][4] out, may be NULL: ref_span (M,D,N,=,X),
 *               lead_clip (leading S/H), trail_clip (trailing S/H),
 *               query_len (M,I,S,H,=,X)
 *   d_ws        scratch of ws_bytes >= svx_cigar_scan_ws_bytes(n_aln, n_words) bytes, 16-byte aligned, contents ignored
 *   n_aln       < 2^30
 *   n_words     the CIGAR words d_cigar holds, >= d_cig_off[n_aln] (the caller knows it as the length of its array; an array that
 *               holds more than it said: SVX_SCAN_FAILED in d_gap_off[n_aln]).  It sizes the frame records and picks the count pass's shape
 * Three launches (count -> offsets, its prefix over the tiles by a decoupled look-back -> emit), no atomics on results.
 * An alignment of more than 512 words (1,024 in launches of 257-1,024 words per alignment) is cut, behind that head, into frames
 * of 512 words whose sums the count pass keeps: the emit pass walks only the frames that hold a long gap.  A launch of long alignments (more than 1,024 words each on average: ONT, contigs)
 * takes the frames in a launch of its own between count and offsets, the array cut into equal ranges.
 * H is treated as S (the reference rewrites H to S, collect_signatures.py:91);
 * N advances the read position only (analyze_reads.py:831-832). */
int svx_cigar_scan(const uint32_t* d_cigar, const uint64_t* d_cig_off,
                   const int32_t* d_ref_start, uint32_t n_aln, uint64_t n_words, int32_t min_sv,
                   SvxGap* d_gaps, uint64_t gaps_cap, uint32_t* d_gap_off,
                   int32_t* d_stats, void* d_ws, uint64_t ws_bytes, uint32_t flags, void* stream);

/* Similarity-image rasteriser (+ mean subtraction, + layout).
 * Replaces BatchGenerator.next_batch's per-image loop
 * (reference src/network/create_batch.py:103-152) and PlotSingleImg.plot
 * (src/segmentplot/plot_segment.py:33-68, incl. cv2.line) for n images.
 *
 *   d_records  [n][12] int32: seg1 x_start,x_end,y_start,y_end,forward(0/1),
 *              seg2 idem, read_len, ref_len  (TSV columns 1..12)
 *   d_out      float32 [n][227][227][3] (NHWC) or [n][3][227][227] (NCHW):
 *              255 - mean[c] on line pixels, -mean[c] elsewhere
 *   mean       host pointer to 3 floats (reference: 104, 117, 124) */
int svx_rasterize(const int32_t* d_records, uint32_t n, float* d_out, int layout,
                  const float* mean, void* stream);

/* Encode + first CNN layer without materialising the image ("sparse conv1").
 * Replaces, for n segment pairs, BatchGenerator.next_batch + PlotSingleImg.plot (as svx_rasterize)
 * AND conv1 -> relu -> pool1 -> norm1 of the reference graph (src/network/alexnet.py:29-31):
 * the image is 255 on a few hundred line pixels and 0 elsewhere, so the 11x11/4 VALID convolution
 * of the mean-subtracted image is base[k] + 255 * (sum of the weight rows of the set pixels).
 *   d_records [n][12] int32 as for svx_rasterize
 *   d_w1      conv1/weights in checkpoint layout HWIO [11][11][3][96], 16-B aligned
 *   d_base    [96]: biases[k] - sum_{ky,kx,ch} mean[ch] * w[ky][kx][ch][k], 16-B aligned
 *   d_y       float32 [n][12][27][27][8] (C8) = norm1 output
 *   d_touched [n][27] or NULL: bit x of word [i][y] = pooled pixel (y, x) of image i has a set tap under it; every
 *             other pixel holds the same constant vector (the response to an empty image) */
int svx_encode_conv1(const int32_t* d_records, uint32_t n, const float* d_w1, const float* d_base, float* d_y,
                     int lrn, uint32_t radius, float alpha, float beta, float k, uint32_t* d_touched, void* stream);

/* Active sets of the AlexNet body (conv2 5x5 on 27x27 -> pool 3x3/2 -> conv3..conv5 3x3 on 13x13,
 * src/network/alexnet.py:34-46): the outputs that can differ from the network's response to an empty image, given
 * the touched pixels of the first layer.  The similarity image is a few thin lines, so only ~37 % of conv2's and
 * 50-90 % of conv3..5's outputs have a line in their receptive field; all others equal a precomputed, image
 * independent background tensor (exactly: every operation is local).
 *   d_list2 [n*729], d_list3 / d_list4 / d_list5 [n*169]: out, permutations of all pixel ids image * H*W + y * W + x:
 *             the active ones first (ascending), then the inactive ones (ascending)
 *   d_counts [4]: out, number of active entries in the four lists;  d_ws: scratch, 16 * n bytes, 16-B aligned
 *   d_totals: NULL, or uint64 [5] running sums the launch ADDS to (never cleared here): [0..4) the output pixels
 *             svx_conv2d_same computes for conv2..conv5 given these lists (all of them once a list is
 *             SVX_CONV_DENSE_PCT full), [4] images -- the executed work of a run, for measurement
 *   d_active2: NULL, or out [n][27]: bit x of word [i][y] = conv2 output pixel (y, x) of image i is in the active set
 *             (what svx_bias_relu_pool_lrn needs to read the others from the background tensor) */
int svx_alexnet_active_sets(const uint32_t* d_touched, uint32_t n, int32_t* d_list2, int32_t* d_list3,
                            int32_t* d_list4, int32_t* d_list5, uint32_t* d_counts, uint32_t* d_ws,
                            uint64_t* d_totals, uint32_t* d_active2, void* stream);

/* Fused conv epilogue: bias add + ReLU + 3x3/2 VALID max-pool (+ TF local response
 * normalisation across channels when lrn != 0), C8 float32 in and out.
 * Replaces the elementwise chain between two convolutions of the reference graph:
 * tf.nn.bias_add + relu (src/network/alexnet.py:132-135), max_pool (:158-161), lrn (:164-166)
 * as composed at alexnet.py:29-31, :34-36, :45-46.
 *   d_x   C8 [n][channels/8][height][width][8]  raw convolution output (no bias); channels % 8 == 0
 *   d_y   C8 [n][channels/8][(height-3)/2+1][(width-3)/2+1][8]          (d_x, d_bias, d_y 16-byte aligned)
 *   LRN:  y = p / (k + alpha * sum_{|j-c| <= radius} p_j^2)^beta   (alpha NOT divided by the window)
 *   d_active_rows + d_background: both NULL, or [n][height] row masks (bit x = pixel (y, x) of d_x was computed; width <= 32)
 *         and the C8 [channels/8][height][width][8] raw response of the producing convolution to an empty image: pixels
 *         whose bit is clear are NOT read from d_x (the active-set convolution need not have written them) but from the
 *         background -- their exact value.  The producer saves the copy of its inactive pixels (61 % of conv2's output),
 *         this kernel the HBM reads of them. */
int svx_bias_relu_pool_lrn(const float* d_x, const float* d_bias, float* d_y, uint32_t n, uint32_t channels,
                           uint32_t height, uint32_t width, int lrn, uint32_t radius, float alpha, float beta,
                           float k, const uint32_t* d_active_rows, const float* d_background, void* stream);

/* Fully connected layer with bias (+ ReLU) on the fp32 matrix cores: out[m][n] = act(bias[n] + sum_k x[m][k] * W[n][k]).
 * Replaces tf.nn.xw_plus_b + relu of fc6 / fc7 (reference src/network/alexnet.py:49-55 via :141-155).
 *   d_x        float32 [m][k] row major
 *   d_w_packed float32 [n/32][k/8][32][8]: packed[b][q][l][j] = W[32 b + l][8 q + j], W = the checkpoint's
 *              [k][n] tensor transposed -- packed once per model (a wave's weight loads are one contiguous stream)
 *   d_out      float32 [m][n]
 *   d_ws       scratch of svx_fc_ws_bytes(m, n, k) bytes (split-K partial sums, added in fixed order: results are
 *              bit-reproducible), 16-byte aligned
 * Requires n % 32 == 0, k % 8 == 0, k >= 192, 16-byte aligned pointers, tensors below 2 GB. */
size_t svx_fc_ws_bytes(uint32_t m, uint32_t n, uint32_t k);
int svx_fc_bias_act(const float* d_x, const float* d_w_packed, const float* d_bias, float* d_out, float* d_ws,
                    uint32_t m, uint32_t n, uint32_t k, int relu, void* stream);

/* fc8 + softmax + argmax in one launch: logits = x @ W^T + b (tf xw_plus_b, src/network/alexnet.py:58,148),
 * tf.nn.softmax and tf.argmax as fetched at src/network/predict.py:209.
 *   d_x [n][4096] fc7 activations, d_w [5][4096] (fc8/weights transposed), d_bias [5]
 *   d_out [n][12] = softmax[5], class (as float), logits[5], 0 */
int svx_fc8_softmax(const float* d_x, const float* d_w, const float* d_bias, float* d_out, uint32_t n, void* stream);

/* fp32 implicit-GEMM convolution on the matrix cores (v_mfma_f32_32x32x2_f32): stride 1, SAME padding,
 * square kernel 3 or 5, optional channel groups, optional fused bias + ReLU.
 * Replaces tf.nn.conv2d (+ split/concat for groups, + bias_add + relu) of the reference layers conv2..conv5
 * (src/network/alexnet.py:34,39,42,45 via :109-135).
 *   d_in       C8 float32 [n][cin/8][height][width][8]
 *   d_w_packed float32 [ksize][ksize][cin_g/8][cout][8], cin_g = cin/groups: the checkpoint tensor
 *              (HWIO [ksize][ksize][cin_g][cout]) with its input-channel axis split into octets and the octet's 8
 *              channels moved innermost -- packed[ky][kx][q][o][j] = hwio[ky][kx][8q + j][o] -- once per model
 *   d_bias     float32 [cout] or NULL (raw convolution output, e.g. in front of svx_bias_relu_pool_lrn)
 *   d_out      C8 float32 [n][cout/8][height][width][8]
 *   d_pixels, d_pixel_count: NULL, or a permutation of all n*H*W output pixel ids (image * H*W + y * W + x) and, in
 *             device memory, the number of leading entries that are active (svx_alexnet_active_sets): only those
 *             outputs are computed (all of them when they are more than 97 %)
 *   d_background: NULL (the other pixels of d_out are left as they are) or C8 float32 [cout/8][height][width][8], the
 *             layer's response to an empty image, copied to the pixels behind the active ones
 * Requires cin_g % 16 == 0, (cout/groups) % 64 == 0, 16-byte aligned pointers, input and weight tensors below 2 GB. */
int svx_conv2d_same(const float* d_in, const float* d_w_packed, const float* d_bias, float* d_out, uint32_t n,
                    uint32_t cin, uint32_t cout, uint32_t height, uint32_t width, uint32_t ksize,
                    uint32_t groups, int relu, const int32_t* d_pixels, const uint32_t* d_pixel_count,
                    const float* d_background, void* stream);

/* Pairwise signature distances of the clustering step, all partitions of a window in one launch (fp64).
 * Replaces the Python-callback pdist inside linkage(data, method="average", metric=span_position_distance)
 * (reference src/collection/cluster_signatures.py:114 with the metric of :132-141):
 *   d(a, b) = min(|s_a - s_b|, |e_a - e_b|, |floor((s_a+e_a)/2) - floor((s_b+e_b)/2)|) / normalizer
 *             + |span_a - span_b| / max(span_a, span_b)          (NaN when both spans are 0, as NumPy's 0/0)
 *   d_start, d_end  [part_off[n_parts]] signature tstart / tend as doubles, partitions concatenated
 *   d_part_off      [n_parts + 1] first signature of every partition
 *   d_out_off       [n_parts + 1] first output element of every partition; partition p of n signatures owns
 *                   n (n - 1) / 2 doubles in scipy's condensed (pdist) order: (0,1), (0,2), ..., (1,2), ...
 *   total_pairs     = d_out_off[n_parts] (host copy, sizes the grid)
 *   d_out           [total_pairs] */
int svx_span_position_distance(const double* d_start, const double* d_end, const uint64_t* d_part_off,
                               uint32_t n_parts, const uint64_t* d_out_off, uint64_t total_pairs,
                               double normalizer, double* d_out, void* stream);

/* k-mer seed-and-extend of the --hash re-aligner, a batch of independent jobs per launch (integer-exact).
 * Replaces the two HashAligner.run passes of hashplot_unmapped (reference src/segmentplot/run_hash_lineplot.py:70-78:
 * makePairwiseAlignment, extendKmersForward / extendKmersReverse, src/segmentplot/hash_aligner.py:145-239, :37-99) up to
 * the raw hit lists; the self-repeat filter, the collinear merge and select_longest stay on the host.
 * Job = (x: the unmapped read piece, y: its reference window), bases packed one per byte as 4-bit symbols:
 * 0..4 = A C G T N, 5..9 = a c g t n, 10..14 = R Y K M S (upstream's k-mers are raw strings: a symbol matches only
 * itself, only upper-case ACGT complement to something else than N, only 'N' stops an extension); a sequence with
 * any other character must be sent down the host path.  For every job the launch writes, in the reference's loop order
 * (y position ascending, then the k-mer's position list: forward positions ascending, then reverse-strand ones):
 *   list A  hits of y against itself for the k-mers of y that occur once among y's k-mers of both strands
 *   list B  hits of x (both strands) on y for the k-mers of y that are not "avoided" (occur once in y)
 * each hit = int32[4] {y position, x position (forward) or position in the reverse complement, match length, forward}
 * with length >= window; d_counts[2 j], d_counts[2 j + 1] = the numbers of hits of job j (they may exceed hit_cap:
 * the lists are then truncated and the caller must redo the job on the host).
 *   d_table  scratch, 16 bytes per slot, sum of table_slots slots; contents ignored
 * Requires repeat_thresh = 2 and mismatchNum = 0 (what hashplot_unmapped passes), 2 <= k <= 13, x_len <= max_x_len <= 2048. */
typedef struct SvxHashJob {
    uint64_t x_off, y_off;       /* byte offsets of the two sequences in d_bases                      */
    uint32_t x_len, y_len;
    uint64_t table_off;          /* first scratch slot of the job                                        */
    uint32_t table_slots;        /* power of two >= 8 * y_len                                            */
    uint32_t hit_cap;            /* capacity of each of the job's two hit lists                         */
    uint64_t hit_off;            /* first hit record of the job: list A at hit_off, list B at + hit_cap */
} SvxHashJob;
int svx_hash_seeds(const uint8_t* d_bases, const SvxHashJob* d_jobs, uint32_t n_jobs, uint64_t* d_table,
                   int32_t* d_hits, uint32_t* d_counts, uint32_t k, uint32_t window, uint32_t max_x_len, void* stream);

/* ---- host side: native BGZF/BAM ingestion (no device work) -------------------------------------------
 * Replaces the per-record pysam iteration of the reference (aln_file.fetch at
 * src/collection/run_collection.py:23-26, field reads at src/collection/collect_signatures.py:128-155):
 * the file is streamed in chunks of BGZF blocks (block-parallel inflate) and the records' fields are appended to
 * packed arrays; SEQ is kept only on request, QUAL and tags never (resident size = the arrays, not the file).
 *   svx_bam_open    -> opaque handle or NULL (svx_bam_error() tells why); threads <= 0: all cores (max 128);
 *                      flags: SVX_BAM_KEEP_SEQ keeps the 4-bit read bases (the --hash re-aligner needs them)
 *   svx_bam_sizes   -> sizes[8] = n_records, n_cigar_words, n_refs, n_names, names_bytes, header_bytes,
 *                      ref_names_bytes, seq_bytes
 *   svx_bam_export  -> fills tid/pos/flag/mapq/l_seq/name_id [n_records], cig_off [n_records+1], cigar
 *                      [n_cigar_words], names / ref_names ('\n'-separated, first-occurrence order), header text,
 *                      ref_lens [n_refs], and (if not NULL) seq_off = byte offset of each record's 4-bit SEQ
 *                      inside the pool exposed by svx_bam_seq() (seq_bytes long) */
#define SVX_BAM_KEEP_SEQ 1
void*          svx_bam_open(const char* path, int threads, int flags);
/* only the records between two BGZF virtual offsets taken from the .bai index (one chromosome = one rank's shard;
 * replaces AlignmentFile.fetch(chrom, ...) random access, run_collection.py:26); voff_end <= voff_beg: header only */
void*          svx_bam_open_range(const char* path, int threads, int flags, uint64_t voff_beg, uint64_t voff_end);
const char*    svx_bam_error(void);
void           svx_bam_sizes(void* handle, uint64_t* sizes);
void           svx_bam_export(void* handle, int threads, int32_t* tid, int32_t* pos, uint16_t* flag, uint8_t* mapq,
                              int32_t* l_seq, int32_t* name_id, int64_t* cig_off, uint32_t* cigar, char* names,
                              char* header, char* ref_names, int32_t* ref_lens, int64_t* seq_off);
const uint8_t* svx_bam_seq(void* handle);
void           svx_bam_close(void* handle);

/* BGZF inflate on the device (svx_inflate2.hip).  Replaces the host-side DEFLATE decoding of htslib / pysam behind
 * run_collection.py:23-26 where host cores are the scarce resource.  d_comp: the compressed bytes as they sit in the file
 * (16-byte aligned -- SVX_EINVAL otherwise --, readable up to the next multiple of 16 behind the last payload); d_src_off /
 * d_src_len [n]: byte offset in d_comp and size of every block's DEFLATE payload (behind the block header, in front of CRC32 +
 * ISIZE); d_dst_off [n + 1]: running sum of the ISIZE fields = where every block's bytes go in d_out; d_status [n]: 0 = the
 * block inflated to exactly ISIZE bytes, anything else = corrupt (the caller falls back to the host decoder).
 * Two kernels: (A) one WAVE per block decodes the Huffman code in parallel -- 64
 * segments of the compressed bits per step, every lane from its segment's first bit, re-synchronised with its predecessor --
 * and transcodes the tokens into a byte-aligned LZ sequence stream; (B) one LANE per block copies literals and matches
 * from that stream.  inflated_bytes: d_dst_off[n] - d_dst_off[0] (the caller summed the ISIZE fields on the host).  d_ws: at least
 * svx_bgzf_inflate_fast_ws_bytes(inflated_bytes, n) bytes (16-byte aligned; SVX_EINVAL if ws_bytes is less) for the
 * sequence streams.  Same statuses; blocks whose stream would not fit its slot (pathological: hundreds of tiny DEFLATE
 * blocks) are decoded by the wave-per-block kernel inside the call. */
size_t         svx_bgzf_inflate_fast_ws_bytes(uint64_t inflated_bytes, uint32_t n_blocks);
int            svx_bgzf_inflate_fast(const uint8_t* d_comp, const uint64_t* d_src_off, const uint32_t* d_src_len,
                                     const uint64_t* d_dst_off, uint32_t n_blocks, uint64_t inflated_bytes, uint8_t* d_out,
                                     uint32_t* d_status, void* d_ws, uint64_t ws_bytes, void* stream);
/* The same with kernel A on stream_tokens and kernel B (and whatever follows in the caller's order: the CRC, the record walk)
 * on stream_lz, which waits for A through an event.  A caller with several launches in flight puts every A on ONE stream:
 * the tokens kernels own the chip while they run, so two of them launched side by side on two streams finish together --
 * late --, while one after the other the first launch's blocks go on to their LZ copies (latency-bound, next to the second
 * launch's A) a whole A earlier.  stream_tokens == stream_lz: svx_bgzf_inflate_fast. */
int            svx_bgzf_inflate_fast_on(const uint8_t* d_comp, const uint64_t* d_src_off, const uint32_t* d_src_len,
                                        const uint64_t* d_dst_off, uint32_t n_blocks, uint64_t inflated_bytes, uint8_t* d_out,
                                        uint32_t* d_status, void* d_ws, uint64_t ws_bytes, void* stream_tokens, void* stream_lz);
/* CRC32 of every inflated block against the block's footer -- the four bytes behind its DEFLATE payload in d_comp (RFC 1952
 * 2.3.1) -- what htslib checks on every block behind pysam's fetch (/root/reference/src/collection/run_collection.py:23-26).
 * d_out / d_dst_off / d_comp / d_src_off / d_src_len: as svx_bgzf_inflate_fast took and wrote them.  d_status [n]: left alone where
 * the CRC agrees or a status is already set, SVX_INFLATE_BAD_CRC where it differs.  One wave per block, coalesced dword
 * reads, one LDS look-up per byte (svx_crc.hip). */
#define SVX_INFLATE_BAD_CRC 9
int            svx_bgzf_crc32(const uint8_t* d_out, const uint64_t* d_dst_off, const uint8_t* d_comp, const uint64_t* d_src_off,
                              const uint32_t* d_src_len, uint32_t n_blocks, uint32_t* d_status, void* stream);
/* BAM records in an inflated stream on the device -> packed arrays (svx_bamdev.hip).  d_starts [n_starts + 1]: byte
 * offsets in d_raw of known record starts (from the .bai linear index), ascending, the last entry = end of the part.
 *   svx_bam_walk_count    d_counts [n_starts][4] = records, CIGAR words, QNAME bytes (one separator per record) between
 *                         start i and start i + 1, and a status: 0 ok, 1 the walk misses the next start, 2 malformed
 *                         record.  A CIGAR of more than 65,535 operations is read from the record's CG:B,I tag
 *                         (SAMv1 4.2.2), by both passes
 *   svx_bam_walk_extract  d_base [n_starts][3] = exclusive prefix sums of the first three counts; fills tid / pos / flag /
 *                         mapq / l_seq [records], cig_off / name_off [records + 1] (offsets of each record's words / name;
 *                         the closing entry = the totals: ABI 370, the caller appended it before),
 *                         cigar [words], names [bytes] ('\n' behind every name).  n_records = the sum of the records counts
 *                         (ABI 420: a lane per start notes where every record lies, a wave per record copies it -- d_tid / d_pos
 *                         carry the record's byte offset between the two launches; d_raw is read in aligned dwords: readable
 *                         up to the next multiple of 4 behind its last byte) */
int            svx_bam_walk_count(const uint8_t* d_raw, const uint64_t* d_starts, uint32_t n_starts, uint64_t* d_counts, void* stream);
int            svx_bam_walk_extract(const uint8_t* d_raw, const uint64_t* d_starts, uint32_t n_starts, const uint64_t* d_base,
                                    int32_t* d_tid, int32_t* d_pos, uint16_t* d_flag, uint8_t* d_mapq, int32_t* d_l_seq,
                                    int64_t* d_cig_off, uint32_t* d_cigar, int64_t* d_name_off, uint8_t* d_names, uint32_t n_records,
                                    void* stream);
/* host helpers of the device-side ingestion: parallel positional read into caller memory; the whole BGZF blocks of a
 * buffer (payload offset / size, ISIZE, file offset; -> their number or -1, *used = bytes they cover); QNAME ids by
 * first occurrence (-> number of distinct names, written '\n'-separated to uniq) */
int            svx_read_range(const char* path, uint64_t off, uint64_t n, uint8_t* dst, int threads);
int64_t        svx_bgzf_index(const uint8_t* bytes, uint64_t n, uint64_t base_coff, uint64_t cap, uint64_t* src_off,
                              uint32_t* src_len, uint32_t* isize, uint64_t* coff, uint64_t* used);
int64_t        svx_name_ids(const uint8_t* names, const int64_t* name_off, uint64_t n, int32_t* name_id, uint8_t* uniq,
                            uint64_t* uniq_bytes);

/* Streaming ingestion, one reference sequence at a time (replaces the reference's window-by-window
 * AlignmentFile.fetch(chrom, start, end), run_collection.py:23-26, by one pass over the file that hands chromosome k
 * to the caller while chromosome k+1 is being read and inflated on the handle's own threads):
 *   svx_bam_stream_open  -> opaque stream or NULL (svx_bam_error()); voffs = n_ranges pairs of BGZF virtual offsets
 *                           from the .bai (a rank's chromosomes, ascending), n_ranges = 0: every record of the file;
 *                           threads <= 0: the CPUs of the process (max 128); flags as svx_bam_open
 *   svx_bam_stream_next  -> the next reference's records as a handle for svx_bam_sizes / svx_bam_export /
 *                           svx_bam_seq / svx_bam_close (QNAME ids count from 0 in every part), or NULL with
 *                           *status = 0 at the end, -1 on error; blocks while that part is being decoded
 *   svx_bam_stream_close -> stops the threads, frees what was not handed out */
void*          svx_bam_stream_open(const char* path, int threads, int flags, const uint64_t* voffs, int n_ranges);
void*          svx_bam_stream_next(void* stream, int* status);
void           svx_bam_stream_close(void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SVX_H */
