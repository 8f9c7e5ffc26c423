/* ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Plain-C, single-thread restatement of the two integer kernels of the SVision
 * hot path.  Used only by tests/, __graft_entry__.smoke() and the cpu_baseline
 * leg of bench.py, as the checker / timed CPU port.  The product library
 * (libsvx.so) never links or calls this file.
 *
 *  - oracle_rasterize : PlotSingleImg.plot + BatchGenerator.next_batch
 *      /root/reference/src/segmentplot/plot_segment.py:8-73
 *      /root/reference/src/network/create_batch.py:88-155
 *      /root/reference/src/segmentplot/classes.py:42-54
 *    including OpenCV cv2.line(thickness 1, LINE_8, shift 0) = clipLine +
 *    LineIterator, restated from OpenCV's published algorithm (OpenCV is not
 *    vendored by the reference nor installed here: PARITY UNPINNED for the
 *    line arithmetic itself).
 *  - oracle_cigar_scan : analyze_inside_align's CIGAR walk
 *      /root/reference/src/collection/analyze_reads.py:828-853
 *    plus the pysam-derived per-alignment spans (SURVEY 8(a')).
 */
#include <stdint.h>
#include <string.h>
#include <stdlib.h>

#define IMG 227

static int clip_line(int64_t w, int64_t h, int64_t *px1, int64_t *py1, int64_t *px2, int64_t *py2)
{
    int64_t x1 = *px1, y1 = *py1, x2 = *px2, y2 = *py2;
    int64_t right = w - 1, bottom = h - 1, a;
    int c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8;
    int c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8;
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
        if (c1 & 12) {
            a = c1 < 8 ? 0 : bottom;
            x1 += (int64_t)((double)(a - y1) * (double)(x2 - x1) / (double)(y2 - y1));
            y1 = a;
            c1 = (x1 < 0) + (x1 > right) * 2;
        }
        if (c2 & 12) {
            a = c2 < 8 ? 0 : bottom;
            x2 += (int64_t)((double)(a - y2) * (double)(x2 - x1) / (double)(y2 - y1));
            y2 = a;
            c2 = (x2 < 0) + (x2 > right) * 2;
        }
        if ((c1 & c2) == 0 && (c1 | c2) != 0) {
            if (c1) {
                a = c1 == 1 ? 0 : right;
                y1 += (int64_t)((double)(a - x1) * (double)(y2 - y1) / (double)(x2 - x1));
                x1 = a;
                c1 = 0;
            }
            if (c2) {
                a = c2 == 1 ? 0 : right;
                y2 += (int64_t)((double)(a - x2) * (double)(y2 - y1) / (double)(x2 - x1));
                x2 = a;
                c2 = 0;
            }
        }
    }
    *px1 = x1; *py1 = y1; *px2 = x2; *py2 = y2;
    return (c1 | c2) == 0;
}

/* cv2.line on a 227x227 byte plane (value 255). pt = (x=col, y=row). */
static void draw_line(uint8_t *plane, int64_t x1, int64_t y1, int64_t x2, int64_t y2)
{
    if ((uint64_t)x1 >= IMG || (uint64_t)x2 >= IMG || (uint64_t)y1 >= IMG || (uint64_t)y2 >= IMG) {
        if (!clip_line(IMG, IMG, &x1, &y1, &x2, &y2))
            return;
    }
    int64_t dx = x2 - x1, dy = y2 - y1;
    if (dx < 0) { dx = -dx; dy = -dy; x1 = x2; y1 = y2; }
    int64_t sy = dy < 0 ? -1 : 1;
    if (dy < 0) dy = -dy;
    int steep = dy > dx;
    if (steep) { int64_t t = dx; dx = dy; dy = t; }
    int64_t err = dx - 2 * dy, x = x1, y = y1;
    for (int64_t i = 0; i <= dx; ++i) {
        plane[y * IMG + x] = 255;
        int minor = err < 0;
        err += -2 * dy + (minor ? 2 * dx : 0);
        if (steep) { y += sy; x += minor; }
        else       { x += 1;  y += minor ? sy : 0; }
    }
}

/* C cast (int64)(double) truncates toward zero, like Python int(float). */
static int64_t scale(int64_t v, double ratio) { return (int64_t)((double)v / ratio); }

/* records: [n][12] int32 = x0,x1,y0,y1,fwd, x0,x1,y0,y1,fwd, read_len, ref_len
 * out: float32, layout 0 = NHWC [n][227][227][3], 1 = NCHW [n][3][227][227] */
int oracle_rasterize(const int32_t *records, uint32_t n, float *out, int layout, const float *mean)
{
    uint8_t *ch0 = (uint8_t *)malloc(3 * IMG * IMG);
    if (!ch0) return -1;
    uint8_t *ch1 = ch0 + IMG * IMG, *ch2 = ch1 + IMG * IMG;
    for (uint32_t i = 0; i < n; ++i) {
        const int32_t *r = records + (size_t)i * 12;
        memset(ch0, 0, 3 * IMG * IMG);
        int64_t m = r[10] > r[11] ? r[10] : r[11];
        double ratio = (double)m / 227.0;
        if (ratio < 1) ratio = 1;
        for (int k = 0; k < 10; k += 5) {
            int64_t xs = r[k], ys = r[k + 2], len = (int64_t)r[k + 3] - r[k + 2];
            int fwd = r[k + 4] != 0;
            int64_t xe = fwd ? xs + (len - 1) : xs - (len - 1);
            int64_t ye = ys + (len - 1);
            if (fwd) {
                draw_line(ch0, scale(ys, ratio), scale(xs, ratio), scale(ye, ratio), scale(xe, ratio));
            } else {
                draw_line(ch0, scale(ye, ratio), scale(xe, ratio), scale(ys, ratio), scale(xs, ratio));
                draw_line(ch2, scale(ye, ratio), scale(xe, ratio), scale(ys, ratio), scale(xs, ratio));
            }
        }
        for (int c = 0; c < IMG; ++c) {
            int cnt = 0;
            for (int y = 0; y < IMG; ++y) cnt += ch0[y * IMG + c] != 0;
            if (cnt >= 2)
                for (int y = 0; y < IMG; ++y) if (ch0[y * IMG + c]) ch1[y * IMG + c] = 255;
        }
        float *o = out + (size_t)i * 3 * IMG * IMG;
        if (layout == 0) {
            for (int p = 0; p < IMG * IMG; ++p) {
                o[p * 3 + 0] = (float)ch0[p] - mean[0];
                o[p * 3 + 1] = (float)ch1[p] - mean[1];
                o[p * 3 + 2] = (float)ch2[p] - mean[2];
            }
        } else {
            for (int p = 0; p < IMG * IMG; ++p) {
                o[p] = (float)ch0[p] - mean[0];
                o[IMG * IMG + p] = (float)ch1[p] - mean[1];
                o[2 * IMG * IMG + p] = (float)ch2[p] - mean[2];
            }
        }
    }
    free(ch0);
    return 0;
}

typedef struct {
    uint32_t aln;      /* alignment index */
    uint32_t op;       /* index of the CIGAR op inside the alignment */
    int32_t read_pos;  /* readPos before the op */
    int32_t ref_pos;   /* refPos before the op */
    int32_t len;       /* op length */
    uint32_t kind;     /* 1 = I, 2 = D */
} OracleGap;

/* cigar: packed BAM words (len<<4|op); cig_off[n_aln+1]; stats: [n_aln][4] int32 =
 * ref_span(M,D,N,=,X), lead_clip(S/H), trail_clip(S/H), query_len(M,I,S,H,=,X).
 * Gaps are emitted in (alignment, op) order.  Returns 0, or -2 if gaps_cap was
 * too small (gap_count still holds the full count). */
int oracle_cigar_scan(const uint32_t *cigar, const uint64_t *cig_off, const int32_t *ref_start,
                      uint32_t n_aln, int32_t min_sv, OracleGap *gaps, uint64_t gaps_cap,
                      uint64_t *gap_count, int32_t *stats)
{
    uint64_t ng = 0;
    for (uint32_t a = 0; a < n_aln; ++a) {
        int64_t read_pos = 0, ref_pos = ref_start[a];
        int64_t ref_span = 0, qlen = 0, lead = 0, trail = 0;
        int in_lead = 1;
        uint64_t b = cig_off[a], e = cig_off[a + 1];
        for (uint64_t j = b; j < e; ++j) {
            uint32_t op = cigar[j] & 15u;
            int64_t len = cigar[j] >> 4;
            int clip = (op == 4 || op == 5);
            if (clip) { if (in_lead) lead += len; trail += len; }
            else { in_lead = 0; trail = 0; }
            switch (op) {
            case 3: ref_span += len; read_pos += len; break;              /* N: reference quirk */
            case 4: case 5: read_pos += len; qlen += len; break;
            case 1:
                if (len >= min_sv) {
                    if (ng < gaps_cap) { OracleGap g = {a, (uint32_t)(j - b), (int32_t)read_pos, (int32_t)ref_pos, (int32_t)len, 1}; gaps[ng] = g; }
                    ++ng;
                }
                read_pos += len; qlen += len; break;
            case 2:
                if (len >= min_sv) {
                    if (ng < gaps_cap) { OracleGap g = {a, (uint32_t)(j - b), (int32_t)read_pos, (int32_t)ref_pos, (int32_t)len, 2}; gaps[ng] = g; }
                    ++ng;
                }
                ref_pos += len; ref_span += len; break;
            case 0: case 7: case 8:
                ref_pos += len; read_pos += len; ref_span += len; qlen += len; break;
            default: break;
            }
        }
        if (in_lead) trail = 0;       /* all-clip CIGAR */
        if (stats) {
            stats[(size_t)a * 4 + 0] = (int32_t)ref_span;
            stats[(size_t)a * 4 + 1] = (int32_t)lead;
            stats[(size_t)a * 4 + 2] = (int32_t)trail;
            stats[(size_t)a * 4 + 3] = (int32_t)qlen;
        }
    }
    *gap_count = ng;
    return ng > gaps_cap ? -2 : 0;
}
