/* decode_nms_ref.c — CPU ORACLE (test infrastructure, NOT the product path): the per-class mask + greedy NMS of
 * keras_inference.py:113-135 in C, a restatement of oracle/decode_ref.py `decode_image`'s second half (same order of operations,
 * one fp32 rounding per TF op; compiled with -ffp-contract=off).  decode_ref.py stays the reference the tests read; this file exists
 * because bench.py's cpu_baseline would otherwise time a one-thread Python loop instead of the CPU path, and is
 * held bit-equal to decode_ref.py by tests/test_oracle_decode.py.
 *
 * TF 1.14 core/kernels/non_max_suppression_op.cc semantics (un-vendored third party, parity unpinned - see decode_ref.py's header):
 * candidates in descending score order (ties: ascending box index), dropped iff IoU(candidate, any selected) > iou_threshold,
 * IoU on min/max-normalised corners, non-positive area => IoU 0, stop at max_output_size. */
#include <stdint.h>
#include <stdlib.h>

static float iou_tf(const float *bi, const float *bj) {
    const float ymin_i = bi[0] < bi[2] ? bi[0] : bi[2], xmin_i = bi[1] < bi[3] ? bi[1] : bi[3];
    const float ymax_i = bi[0] > bi[2] ? bi[0] : bi[2], xmax_i = bi[1] > bi[3] ? bi[1] : bi[3];
    const float ymin_j = bj[0] < bj[2] ? bj[0] : bj[2], xmin_j = bj[1] < bj[3] ? bj[1] : bj[3];
    const float ymax_j = bj[0] > bj[2] ? bj[0] : bj[2], xmax_j = bj[1] > bj[3] ? bj[1] : bj[3];
    const float hi = ymax_i - ymin_i, wi = xmax_i - xmin_i, hj = ymax_j - ymin_j, wj = xmax_j - xmin_j;
    const float area_i = hi * wi, area_j = hj * wj;
    if (area_i <= 0.f || area_j <= 0.f) return 0.f;
    const float iy0 = ymin_i > ymin_j ? ymin_i : ymin_j, ix0 = xmin_i > xmin_j ? xmin_i : xmin_j;
    const float iy1 = ymax_i < ymax_j ? ymax_i : ymax_j, ix1 = xmax_i < xmax_j ? xmax_i : xmax_j;
    float dh = iy1 - iy0, dw = ix1 - ix0;
    if (dh < 0.f) dh = 0.f;
    if (dw < 0.f) dw = 0.f;
    const float inter = dh * dw;
    const float sum = area_i + area_j;
    const float uni = sum - inter;
    return inter / uni;
}

typedef struct {
    float s;
    int32_t i;
} cand_t;
static int cand_cmp(const void *a, const void *b) {
    const cand_t *x = (const cand_t *)a, *y = (const cand_t *)b;
    if (x->s > y->s) return -1;
    if (x->s < y->s) return 1;
    return (x->i > y->i) - (x->i < y->i);
}

/* One image.  boxes [n][4] (ymin,xmin,ymax,xmax), scores [n][nc].  rows_out [nc*max_out][6] = top,left,bottom,right,score,class in the
 * reference's class-major order; idx_out = box index of every row.  Returns the row count. */
int yk_ref_nms_image(const float *boxes, const float *scores, int n, int nc, float obj_thresh, float iou_thresh, int max_out, float *rows_out,
                     int32_t *idx_out) {
    cand_t *cand = (cand_t *)malloc(sizeof(cand_t) * (size_t)(n > 0 ? n : 1));
    int32_t *sel = (int32_t *)malloc(sizeof(int32_t) * (size_t)(max_out > 0 ? max_out : 1));
    int k = 0;
    for (int c = 0; c < nc; ++c) {
        int m = 0;
        for (int i = 0; i < n; ++i)
            if (scores[(size_t)i * nc + c] >= obj_thresh) {
                cand[m].s = scores[(size_t)i * nc + c];
                cand[m].i = i;
                ++m;
            }
        if (!m) continue;
        qsort(cand, (size_t)m, sizeof(cand_t), cand_cmp);
        int ns = 0;
        for (int q = 0; q < m && ns < max_out; ++q) {
            int keep = 1;
            for (int s = ns - 1; s >= 0; --s)
                if (iou_tf(boxes + (size_t)cand[q].i * 4, boxes + (size_t)sel[s] * 4) > iou_thresh) {
                    keep = 0;
                    break;
                }
            if (!keep) continue;
            sel[ns++] = cand[q].i;
            float *r = rows_out + (size_t)k * 6;
            const float *b = boxes + (size_t)cand[q].i * 4;
            r[0] = b[0]; r[1] = b[1]; r[2] = b[2]; r[3] = b[3];
            r[4] = cand[q].s;
            r[5] = (float)c;
            idx_out[k++] = cand[q].i;
        }
    }
    free(cand);
    free(sel);
    return k;
}

/* A batch, images in parallel (OpenMP): boxes [B][n][4], scores [B][n][nc] -> rows_out [B][nc*max_out][6], idx_out [B][nc*max_out], counts [B] */
void yk_ref_nms_batch(const float *boxes, const float *scores, int B, int n, int nc, float obj_thresh, float iou_thresh, int max_out,
                      float *rows_out, int32_t *idx_out, int32_t *counts, int threads) {
    const size_t cap = (size_t)nc * max_out;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
    for (int b = 0; b < B; ++b)
        counts[b] = yk_ref_nms_image(boxes + (size_t)b * n * 4, scores + (size_t)b * n * nc, n, nc, obj_thresh, iou_thresh, max_out,
                                     rows_out + (size_t)b * cap * 6, idx_out + (size_t)b * cap);
}
