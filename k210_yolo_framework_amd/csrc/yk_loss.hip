// yk_loss.hip — YOLO loss forward + dL/dy_pred + ignore mask + precision/recall counters for one output layer.
//
// Replaces, as ONE fused pass per image instead of a TF graph with a Python loop
// over the batch (tools/utils.py:698-705):
//   create_loss_fn / loss_fn        tools/utils.py:708-793
//   calc_ignore_mask, tf_iou        tools/utils.py:662-705, 617-659
//   tf_xywh_to_all / _to_grid       tools/utils.py:524-572
//   Yolo_Precision / Yolo_Recall    tools/custom.py:13-75   (raw logit vs threshold, custom.py:33)
// The gradient is TF autodiff's: ignore_mask comes from a comparison and carries none.
// One workgroup per image: (1) the image's ground-truth boxes (cells with y_true conf > obj_thresh) are
// compacted into LDS, (2) every prediction is decoded, matched against them (best IoU), contributes its five
// loss terms and writes its 5+C gradient entries, (3) a block reduction in fp64 produces per-image partials;
// a second tiny kernel adds the partials in image order (deterministic).
#include "yk_common.h"

#define YK_LOSS_MAXGT 1024

struct loss_args {
    int h, w, A, C, E, batch_size;
    float anchors[YK_MAX_ANCHORS][2];
    float obj_thresh, iou_thresh, obj_w, noobj_w, wh_w;
};

__device__ __forceinline__ float l_sigmoid(float v) { return 1.f / (1.f + expf(-v)); }
__device__ __forceinline__ float l_bce(float z, float x) { return fmaxf(x, 0.f) - x * z + log1pf(expf(-fabsf(x))); }

__global__ void __launch_bounds__(256) yolo_loss_kernel(loss_args a, const float *__restrict__ y_true,
                                                        const float *__restrict__ y_pred, float *__restrict__ grad,
                                                        float *__restrict__ ignore_out, double *__restrict__ partial) {
    __shared__ float4 gt[YK_LOSS_MAXGT];
    __shared__ int ngt;
    __shared__ double red[9][4];
    // grid (batch, chunks of 256 boxes): one workgroup per image left 16 workgroups on the chip for 41 us; every chunk gathers the image's
    // ground truth itself (a strided scan of P objectness values) and handles one box per thread
    const int b = blockIdx.x, tid = threadIdx.x;
    const int P = a.h * a.w * a.A;
    const float *yt = y_true + (size_t)b * P * a.E, *yp = y_pred + (size_t)b * P * a.E;
    if (tid == 0) ngt = 0;
    __syncthreads();
    for (int p = tid; p < P; p += 256) {
        const float *t = yt + (size_t)p * a.E;
        if (t[4] > a.obj_thresh) {
            const int k = atomicAdd(&ngt, 1);
            if (k < YK_LOSS_MAXGT) gt[k] = make_float4(t[0], t[1], t[2], t[3]);
        }
    }
    __syncthreads();
    const int n = ngt;
    const float inv_bs = 1.f / (float)a.batch_size;
    double s_xy = 0, s_wh = 0, s_obj = 0, s_noobj = 0, s_cls = 0;
    int tp = 0, fp = 0, fn = 0;
    for (int p = blockIdx.y * 256 + tid; p < min(P, (int)(blockIdx.y + 1) * 256); p += 256) {
        const float *t = yt + (size_t)p * a.E, *q = yp + (size_t)p * a.E;
        const int an = p % a.A, cell = p / a.A, col = cell % a.w, row = cell / a.w;
        const float px = q[0], py = q[1], pw = q[2], ph = q[3], pc = q[4];
        const float tx = t[0], ty = t[1], tw = t[2], th = t[3], tc = t[4];
        // prediction in image scale (tf_xywh_to_all)
        const float sx = l_sigmoid(px), sy = l_sigmoid(py);
        const float ax = (sx + (float)col) / (float)a.w, ay = (sy + (float)row) / (float)a.h;
        const float aw = expf(pw) * a.anchors[an][0], ah = expf(ph) * a.anchors[an][1];
        // ignore mask: best IoU against this image's ground truth (empty set -> -inf -> 1)
        float best = -INFINITY;
        if (n <= YK_LOSS_MAXGT) {
            for (int k = 0; k < n; ++k) {
                const float4 g = gt[k];
                const float iw = fmaxf(fminf(ax + aw / 2.f, g.x + g.z / 2.f) - fmaxf(ax - aw / 2.f, g.x - g.z / 2.f), 0.f);
                const float ih = fmaxf(fminf(ay + ah / 2.f, g.y + g.w / 2.f) - fmaxf(ay - ah / 2.f, g.y - g.w / 2.f), 0.f);
                const float inter = iw * ih;
                best = fmaxf(best, inter / (aw * ah + g.z * g.w - inter));
            }
        } else {   // more boxes than LDS slots: scan the label tensor directly
            for (int k = 0; k < P; ++k) {
                const float *g = yt + (size_t)k * a.E;
                if (!(g[4] > a.obj_thresh)) continue;
                const float iw = fmaxf(fminf(ax + aw / 2.f, g[0] + g[2] / 2.f) - fmaxf(ax - aw / 2.f, g[0] - g[2] / 2.f), 0.f);
                const float ih = fmaxf(fminf(ay + ah / 2.f, g[1] + g[3] / 2.f) - fmaxf(ay - ah / 2.f, g[1] - g[3] / 2.f), 0.f);
                const float inter = iw * ih;
                best = fmaxf(best, inter / (aw * ah + g[2] * g[3] - inter));
            }
        }
        const float ign = (best < a.iou_thresh) ? 1.f : 0.f;
        if (ignore_out) ignore_out[(size_t)b * P + p] = ign;
        const bool ob = tc > a.obj_thresh;
        const float obj = tc;
        // targets in grid scale
        const float gx = tx * (float)a.w - (float)col, gy = ty * (float)a.h - (float)row;
        const float gw = ob ? logf(tw / a.anchors[an][0]) : 0.f, gh = ob ? logf(th / a.anchors[an][1]) : 0.f;
        const float cw = 2.f - tw * th;
        s_xy += (double)(obj * cw * (l_bce(gx, px) + l_bce(gy, py)));
        s_wh += (double)(obj * cw * a.wh_w * ((gw - pw) * (gw - pw) + (gh - ph) * (gh - ph)));
        const float bc = l_bce(tc, pc);
        s_obj += (double)(obj * bc);
        s_noobj += (double)((1.f - obj) * ign * bc);
        float cls = 0.f;
        float *gr = grad ? grad + ((size_t)b * P + p) * a.E : nullptr;
        for (int c = 0; c < a.C; ++c) {
            const float x = q[5 + c], z = t[5 + c];
            cls += l_bce(z, x);
            if (gr) gr[5 + c] = obj * (l_sigmoid(x) - z) * inv_bs;
        }
        s_cls += (double)(obj * cls);
        if (gr) {
            gr[0] = obj * cw * (sx - gx) * inv_bs;
            gr[1] = obj * cw * (sy - gy) * inv_bs;
            gr[2] = obj * cw * a.wh_w * 2.f * (pw - gw) * inv_bs;
            gr[3] = obj * cw * a.wh_w * 2.f * (ph - gh) * inv_bs;
            gr[4] = (a.obj_w * obj + a.noobj_w * (1.f - obj) * ign) * (l_sigmoid(pc) - tc) * inv_bs;
        }
        const bool pp = pc > a.obj_thresh;     // custom.py:33: raw logit
        tp += (ob && pp);
        fp += (!ob && pp);
        fn += (ob && !pp);
    }
    double v[9] = {s_xy, s_wh, s_obj, s_noobj, s_cls, (double)tp, (double)fp, (double)fn, 0.0};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o, 64);
        if ((tid & 63) == 0) red[k][tid >> 6] = v[k];
    }
    __syncthreads();
    if (tid < 8) partial[((size_t)b * gridDim.y + blockIdx.y) * 8 + tid] = red[tid][0] + red[tid][1] + red[tid][2] + red[tid][3];
}

// out_loss = {total, xy, wh, obj, noobj, cls}; counts += {tp, fp, fn}
__global__ void yolo_loss_finish_kernel(loss_args a, int batch, const double *__restrict__ partial, float *__restrict__ out_loss,
                                        float *__restrict__ counts) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = 0; b < batch; ++b)
        for (int k = 0; k < 8; ++k) s[k] += partial[(size_t)b * 8 + k];
    const double bs = (double)a.batch_size;
    const double xy = s[0] / bs, wh = s[1] / bs, ob = a.obj_w * s[2] / bs, no = a.noobj_w * s[3] / bs, cl = s[4] / bs;
    out_loss[0] = (float)(ob + no + cl + xy + wh);   // utils.py:789
    out_loss[1] = (float)xy;
    out_loss[2] = (float)wh;
    out_loss[3] = (float)ob;
    out_loss[4] = (float)no;
    out_loss[5] = (float)cl;
    if (counts) {
        counts[0] += (float)s[5];
        counts[1] += (float)s[6];
        counts[2] += (float)s[7];
    }
}

extern "C" int yk_yolo_loss(const yk_loss_cfg_t *cfg, const float *d_y_true, const float *d_y_pred, int batch, float *d_loss,
                            float *d_grad, float *d_ignore, float *d_counts, void *stream) {
    if (!cfg || !d_y_true || !d_y_pred || !d_loss || batch <= 0 || cfg->out_h <= 0 || cfg->out_w <= 0 || cfg->anchor_num <= 0 ||
        cfg->anchor_num > YK_MAX_ANCHORS || cfg->class_num <= 0 || cfg->batch_size <= 0) {
        yk_set_error("yk_yolo_loss: bad argument");
        return YK_ERR_ARG;
    }
    int dev = yk_current_device();
    if (dev < 0) {
        yk_set_error("yk_yolo_loss: no HIP device");
        return YK_ERR_NO_DEVICE;
    }
    loss_args a;
    a.h = cfg->out_h;
    a.w = cfg->out_w;
    a.A = cfg->anchor_num;
    a.C = cfg->class_num;
    a.E = 5 + cfg->class_num;
    a.batch_size = cfg->batch_size;
    for (int n = 0; n < a.A; ++n) {
        a.anchors[n][0] = cfg->anchors[n][0];
        a.anchors[n][1] = cfg->anchors[n][1];
    }
    a.obj_thresh = cfg->obj_thresh;
    a.iou_thresh = cfg->iou_thresh;
    a.obj_w = cfg->obj_weight;
    a.noobj_w = cfg->noobj_weight;
    a.wh_w = cfg->wh_weight;
    const int chunks = (a.h * a.w * a.A + 255) / 256;
    double *partial = (double *)yk_scratch(dev, stream, 1, sizeof(double) * 8 * batch * chunks);
    if (!partial) return YK_ERR_NOMEM;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(yolo_loss_kernel, dim3(batch, chunks), dim3(256), 0, st, a, d_y_true, d_y_pred, d_grad, d_ignore, partial);
    hipLaunchKernelGGL(yolo_loss_finish_kernel, dim3(1), dim3(64), 0, st, a, batch * chunks, partial, d_loss, d_counts);   // (image, chunk) in order
    YK_HIP(hipGetLastError());
    return YK_OK;
}
