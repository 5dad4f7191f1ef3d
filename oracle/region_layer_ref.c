/* region_layer_ref.c — CPU ORACLE (test infrastructure, NOT the product path).
 *
 * A restatement, in this repo's own structure, of the algorithm of the
 * reference's yolo3_frame_test_public/region_layer.c.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * shipped library (libyolo_hip.so) never does.
 *
 * Pinned: yes — tests/test_oracle_region.py compares every output of this file
 * with (a) the committed golden vectors under tests/golden/region_*.npz, which
 * were produced by running the reference's own region_layer.c here
 * (oracle/build_ref.sh -> oracle/_ref/libregion_ref.so, generator
 * tests/golden/make_region_golden.py), and (b) the live oracle/_ref build when
 * it is present.
 *
 * Float semantics are kept operation-for-operation (fp32 with the reference's
 * two double promotions), so on x86-64 the results are bit-identical to the
 * reference except where libm's expf differs (same glibc here -> identical).
 *
 * Operates on plain arrays instead of region_layer_t:
 *   in/out : CHW fp32 [A*(5+C)][H][W], element (n, e, loc) at (n*(5+C)+e)*H*W + loc
 *   boxes  : [A*H*W][4] = x,y,w,h ; box index = n*H*W + loc     (region_layer.c:190)
 *   probs  : [A*H*W][C+1]                                        (region_layer.c:47,58)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static float logistic(float v) { return 1.f / (1.f + expf(-v)); } /* region_layer.c:75 */

/* C2: forward_region_layer, region_layer.c:121-137 (+ activate_array :77-82, softmax :91-109) */
void rlref_forward(const float *in, float *out, int W, int H, int A, int C) {
    const int hw = W * H, E = 5 + C;
    memcpy(out, in, sizeof(float) * (size_t)A * E * hw);
    for (int n = 0; n < A; ++n) {
        const float *src = in + (size_t)n * E * hw;
        float *dst = out + (size_t)n * E * hw;
        for (int i = 0; i < 2 * hw; ++i) dst[i] = logistic(src[i]);               /* x, y planes */
        for (int i = 0; i < hw; ++i) dst[4 * hw + i] = logistic(src[4 * hw + i]); /* objectness */
        for (int loc = 0; loc < hw; ++loc) {                                       /* class softmax */
            const float *ci = src + 5 * hw + loc;
            float *co = dst + 5 * hw + loc;
            float top = ci[0];
            for (int j = 0; j < C; ++j)
                if (ci[j * hw] > top) top = ci[j * hw];
            float total = 0;
            for (int j = 0; j < C; ++j) {
                float e = expf(ci[j * hw] - top);
                total += e;
                co[j * hw] = e;
            }
            for (int j = 0; j < C; ++j) co[j * hw] /= total;
        }
    }
}

/* C3: get_region_boxes :177-214, get_region_box :166-175, correct_region_boxes :139-164 */
void rlref_boxes(const float *act, const float *anchor, int W, int H, int A, int C, float threshold,
                 uint32_t net_w, uint32_t net_h, uint32_t image_w, uint32_t image_h, float *boxes, float *probs) {
    const int hw = W * H, E = 5 + C;
    for (int loc = 0; loc < hw; ++loc) {
        const int row = loc / W, col = loc % W;
        for (int n = 0; n < A; ++n) {
            const int bi = n * hw + loc;
            const float *p = act + (size_t)n * E * hw + loc;
            float *pr = probs + (size_t)bi * (C + 1);
            const float obj = p[4 * hw];
            float *b = boxes + 4 * (size_t)bi;
            b[0] = (col + p[0 * hw]) / W;
            b[1] = (row + p[1 * hw]) / H;
            b[2] = expf(p[2 * hw]) * anchor[2 * n];
            b[3] = expf(p[3 * hw]) * anchor[2 * n + 1];
            float best = 0;
            for (int j = 0; j < C; ++j) {
                float pj = obj * p[(5 + j) * hw];
                pr[j] = (pj > threshold) ? pj : 0;
                if (pj > best) best = pj;
            }
            pr[C] = best;
        }
    }
    /* letterbox un-map with integer new_w/new_h (unsigned arithmetic as in the reference) */
    int new_w, new_h;
    if (((float)net_w / image_w) < ((float)net_h / image_h)) {
        new_w = net_w;
        new_h = (image_h * net_w) / image_w;
    } else {
        new_h = net_h;
        new_w = (image_w * net_h) / image_h;
    }
    const int nb = A * hw;
    for (int i = 0; i < nb; ++i) {
        float *b = boxes + 4 * (size_t)i;
        b[0] = (b[0] - (net_w - new_w) / 2. / net_w) / ((float)new_w / net_w);   /* double promote, :158 */
        b[1] = (b[1] - (net_h - new_h) / 2. / net_h) / ((float)new_h / net_h);
        b[2] *= (float)net_w / new_w;
        b[3] *= (float)net_h / new_h;
    }
}

/* :228-254 */
static float span(float c1, float s1, float c2, float s2) {
    float lo1 = c1 - s1 / 2, lo2 = c2 - s2 / 2;
    float lo = lo1 > lo2 ? lo1 : lo2;
    float hi1 = c1 + s1 / 2, hi2 = c2 + s2 / 2;
    float hi = hi1 < hi2 ? hi1 : hi2;
    return hi - lo;
}
float rlref_iou(const float *a, const float *b) {
    float w = span(a[0], a[2], b[0], b[2]);
    float h = span(a[1], a[3], b[1], b[3]);
    float inter = (w < 0 || h < 0) ? 0 : w * h;
    float uni = a[2] * a[3] + b[2] * b[3] - inter;
    return inter / uni;
}

typedef struct {
    float p;
    int idx;
} rank_t;
static int by_prob_desc(const void *x, const void *y) {
    const rank_t *a = (const rank_t *)x, *b = (const rank_t *)y;
    if (a->p > b->p) return -1;
    if (a->p < b->p) return 1;
    return (a->idx > b->idx) - (a->idx < b->idx); /* ties: ascending box index (documented choice) */
}

/* C4: do_nms_sort :256-283.  The reference sorts all boxes per class with qsort (order of equal
 * probabilities unspecified); only exact ties between non-zero probabilities could differ. */
void rlref_nms(const float *boxes, float *probs, int nb, int C, float nms_value) {
    rank_t *r = (rank_t *)malloc(sizeof(rank_t) * (size_t)nb);
    for (int k = 0; k < C; ++k) {
        for (int i = 0; i < nb; ++i) {
            r[i].p = probs[(size_t)i * (C + 1) + k];
            r[i].idx = i;
        }
        qsort(r, nb, sizeof(rank_t), by_prob_desc);
        for (int i = 0; i < nb; ++i) {
            float *pi = probs + (size_t)r[i].idx * (C + 1) + k;
            if (*pi == 0) continue;
            const float *a = boxes + 4 * (size_t)r[i].idx;
            for (int j = i + 1; j < nb; ++j) {
                const float *b = boxes + 4 * (size_t)r[j].idx;
                if (rlref_iou(a, b) > nms_value) probs[(size_t)r[j].idx * (C + 1) + k] = 0;
            }
        }
    }
    free(r);
}

static uint32_t to_u32(float v) { return (uint32_t)(int64_t)v; } /* defined form of :397-400's UB cast */

/* C5: region_layer_draw_boxes :385-404 (+ max_index :285-296). dets: rows of 6 uint32
 * (x1,y1,x2,y2,class,prob-bits).  Returns the number of callbacks the reference would make. */
int rlref_draw(const float *boxes, const float *probs, int nb, int C, float threshold, uint32_t image_w,
               uint32_t image_h, uint32_t *dets, int max_dets) {
    int n = 0;
    for (int i = 0; i < nb; ++i) {
        const float *pr = probs + (size_t)i * (C + 1);
        int cls = 0;
        float top = pr[0];
        for (int j = 1; j < C; ++j)
            if (pr[j] > top) {
                top = pr[j];
                cls = j;
            }
        if (top > threshold) {
            const float *b = boxes + 4 * (size_t)i;
            if (n < max_dets) {
                uint32_t *d = dets + 6 * (size_t)n;
                d[0] = to_u32(b[0] * image_w - (b[2] * image_w / 2));
                d[1] = to_u32(b[1] * image_h - (b[3] * image_h / 2));
                d[2] = to_u32(b[0] * image_w + (b[2] * image_w / 2));
                d[3] = to_u32(b[1] * image_h + (b[3] * image_h / 2));
                d[4] = (uint32_t)cls;
                memcpy(&d[5], &top, 4);
            }
            ++n;
        }
    }
    return n;
}

/* region_layer_run :378-383 on plain arrays */
void rlref_run(const float *in, float *out, const float *anchor, int W, int H, int A, int C, float threshold,
               float nms_value, uint32_t net_w, uint32_t net_h, uint32_t image_w, uint32_t image_h, float *boxes,
               float *probs) {
    rlref_forward(in, out, W, H, A, C);
    rlref_boxes(out, anchor, W, H, A, C, threshold, net_w, net_h, image_w, image_h, boxes, probs);
    rlref_nms(boxes, probs, A * W * H, C, nms_value);
}
