/* yolo_net_ref.c — CPU ORACLE (test infrastructure, NOT the product path).
 *
 * Plain-C fp32 NHWC interpreter of a layer plan (k210_yolo_framework_amd/netspec.py),
 * i.e. a restatement of what the reference obtains from tf.keras for
 *   Conv2D / DepthwiseConv2D / BatchNormalization(inference) / LeakyReLU / ReLU /
 *   ReLU(6) / MaxPooling2D('same') / UpSampling2D(2) / Concatenate / Add
 * as called from models/yolonet.py:12-260, models/keras_mobilenet.py:291-436 and
 * models/keras_mobilenet_v2.py:426-485.
 *
 * PARITY UNPINNED against the reference itself: the arithmetic of these layers
 * lives in tensorflow_gpu==1.14.0 (requirements.txt:3), which is not vendored,
 * cannot be installed here (Python 3.10, no network) and for which the reference
 * holds no golden vectors (it has no tests).  The substitute arbiter
 * is a second, independent implementation: tests/test_oracle_net.py checks this
 * file layer-by-layer and end-to-end against torch-CPU functional ops built
 * straight from the Keras layer parameters (unfolded BatchNorm).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * emulate_f16 != 0 mirrors the storage rule of the HIP engine (weights and every
 * stored activation rounded to IEEE fp16, fp32 accumulation, fp32 scale/bias,
 * fp32 network outputs) so that kernel bugs can be told apart from fp16 drift.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { OP_CONV = 1, OP_DWCONV = 2, OP_MAXPOOL = 3, OP_UPSAMPLE = 4, OP_CONCAT = 5, OP_ADD = 6 };
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_RELU6 = 2, ACT_LEAKY = 3 };
enum { F_TYPE = 0, F_IN0, F_IN1, F_OUT, F_CIN, F_COUT, F_K, F_STRIDE, F_PAD_T, F_PAD_L, F_ACT, F_ALPHA,
       F_W_OFF, F_SCALE_OFF, F_BIAS_OFF, F_FLAGS, F_IN_H, F_IN_W, F_OUT_H, F_OUT_W, OP_FIELDS = 24 };

/* round-to-nearest-even fp32 -> fp16 -> fp32, software (no F16C dependence) */
static float f16_round(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t sign = x & 0x80000000u;
    uint32_t ax = x & 0x7fffffffu;
    float r;
    if (ax >= 0x7f800000u) return f;                 /* inf / nan */
    if (ax >= 0x477ff000u) {                         /* >= 65520 -> inf */
        uint32_t inf = sign | 0x7f800000u;
        memcpy(&r, &inf, 4);
        return r;
    }
    if (ax < 0x38800000u) {                          /* subnormal half: quantum 2^-24 */
        float a = fabsf(f);
        float q = a * 16777216.0f;                   /* exact scaling */
        float rq = nearbyintf(q);                    /* RNE in default rounding mode */
        r = rq / 16777216.0f;
        return sign ? -r : r;
    }
    uint32_t lsb = (ax >> 13) & 1u;
    ax += 0xfffu + lsb;
    ax &= ~0x1fffu;
    x = sign | ax;
    memcpy(&r, &x, 4);
    return r;
}

static inline float act_fn(float v, int act, float alpha) {
    switch (act) {
    case ACT_RELU: return v > 0 ? v : 0;
    case ACT_RELU6: return v < 0 ? 0 : (v > 6.f ? 6.f : v);
    case ACT_LEAKY: return v >= 0 ? v : v * alpha;   /* keras LeakyReLU: alpha*x for x<0 */
    default: return v;
    }
}

typedef struct {
    int h, w, c;
    float *d;
} tens_t;

/* returns 0 ok, <0 on error.  input: fp32 [batch][H][W][3] already normalised.
 * out_ptrs[i]: caller buffer for tensor out_ids[i], fp32 [batch][h][w][c].
 * dump_id >= 0 additionally copies that tensor into dump (fp32 NHWC). */
/* OpenMP team size of this library's loops.  The default (every logical CPU, spinning at barriers) collapses on a shared host; the
 * python wrapper caps it at 32 unless ORACLE_THREADS says otherwise, bench.py's cpu_baseline asks for all cores explicitly. */
#ifdef _OPENMP
#include <omp.h>
int yk_ref_set_threads(int n) {
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
}
#else
int yk_ref_set_threads(int n) {
    (void)n;
    return 1;
}
#endif

int yk_ref_forward_ex(const int32_t *ops, int n_ops, const int32_t *tensors, int n_t, const float *blob_in,
                      size_t blob_len, int n_in, const int32_t *in_ids, const float *const *in_ptrs, int batch,
                      int emulate_f16, const int32_t *out_ids, int n_out, float **out_ptrs, int dump_id, float *dump) {
    tens_t *T = (tens_t *)calloc((size_t)n_t, sizeof(tens_t));
    if (!T) return -1;
    for (int i = 0; i < n_t; ++i) {
        T[i].h = tensors[4 * i];
        T[i].w = tensors[4 * i + 1];
        T[i].c = tensors[4 * i + 2];
    }
    float *blob = (float *)malloc(sizeof(float) * blob_len);
    memcpy(blob, blob_in, sizeof(float) * blob_len);
    if (emulate_f16) { /* weights only; scale/bias stay fp32 */
        for (int i = 0; i < n_ops; ++i) {
            const int32_t *o = ops + (size_t)i * OP_FIELDS;
            if (o[F_TYPE] == OP_CONV) {
                size_t n = (size_t)o[F_COUT] * o[F_K] * o[F_K] * o[F_CIN];
                for (size_t j = 0; j < n; ++j) blob[o[F_W_OFF] + j] = f16_round(blob[o[F_W_OFF] + j]);
            } else if (o[F_TYPE] == OP_DWCONV) {
                size_t n = (size_t)9 * o[F_CIN];
                for (size_t j = 0; j < n; ++j) blob[o[F_W_OFF] + j] = f16_round(blob[o[F_W_OFF] + j]);
            }
        }
    }
    /* pre-filled tensors: the network input (id 0) or, for layer-at-a-time checks, any set of tensors */
    for (int i = 0; i < n_in; ++i) {
        const int id = in_ids[i];
        size_t in_elems = (size_t)batch * T[id].h * T[id].w * T[id].c;
        T[id].d = (float *)malloc(sizeof(float) * in_elems);
        memcpy(T[id].d, in_ptrs[i], sizeof(float) * in_elems);
        if (emulate_f16) /* the normalised image is stored in fp16 like every other activation (idempotent for the rest) */
            for (size_t j = 0; j < in_elems; ++j) T[id].d[j] = f16_round(T[id].d[j]);
    }

    int rc = 0;
    for (int i = 0; i < n_ops && rc == 0; ++i) {
        const int32_t *o = ops + (size_t)i * OP_FIELDS;
        const tens_t *X = &T[o[F_IN0]];
        tens_t *Y = &T[o[F_OUT]];
        const int Hi = X->h, Wi = X->w, Ci = X->c, Ho = Y->h, Wo = Y->w, Co = Y->c;
        if (!X->d || (o[F_IN1] >= 0 && !T[o[F_IN1]].d)) { rc = -4; break; }   /* input never produced */
        free(Y->d);
        Y->d = (float *)malloc(sizeof(float) * (size_t)batch * Ho * Wo * Co);
        if (!Y->d) { rc = -2; break; }
        const int net_out = o[F_FLAGS] & 1;
        const int round_out = emulate_f16 && !net_out;
        float alpha;
        memcpy(&alpha, &o[F_ALPHA], 4);
        const int k = o[F_K], st = o[F_STRIDE], pt = o[F_PAD_T], pl = o[F_PAD_L], act = o[F_ACT];
        switch (o[F_TYPE]) {
        case OP_CONV: {
            const float *Wt = blob + o[F_W_OFF], *sc = blob + o[F_SCALE_OFF], *bs = blob + o[F_BIAS_OFF];
#pragma omp parallel for collapse(2) schedule(static)
            for (int b = 0; b < batch; ++b)
                for (int oy = 0; oy < Ho; ++oy)
                    for (int ox = 0; ox < Wo; ++ox) {
                        float *y = Y->d + (((size_t)b * Ho + oy) * Wo + ox) * Co;
                        for (int co = 0; co < Co; ++co) {
                            float acc = 0.f;
                            for (int ky = 0; ky < k; ++ky) {
                                int iy = oy * st - pt + ky;
                                if (iy < 0 || iy >= Hi) continue;
                                for (int kx = 0; kx < k; ++kx) {
                                    int ix = ox * st - pl + kx;
                                    if (ix < 0 || ix >= Wi) continue;
                                    const float *x = X->d + (((size_t)b * Hi + iy) * Wi + ix) * Ci;
                                    const float *w = Wt + (((size_t)co * k + ky) * k + kx) * Ci;
                                    float s = 0.f;
#pragma omp simd reduction(+ : s)
                                    for (int ci = 0; ci < Ci; ++ci) s += x[ci] * w[ci];
                                    acc += s;
                                }
                            }
                            float v = act_fn(acc * sc[co] + bs[co], act, alpha);
                            y[co] = round_out ? f16_round(v) : v;
                        }
                    }
        } break;
        case OP_DWCONV: {
            const float *Wt = blob + o[F_W_OFF], *sc = blob + o[F_SCALE_OFF], *bs = blob + o[F_BIAS_OFF];
#pragma omp parallel for collapse(2) schedule(static)
            for (int b = 0; b < batch; ++b)
                for (int oy = 0; oy < Ho; ++oy)
                    for (int ox = 0; ox < Wo; ++ox) {
                        float *y = Y->d + (((size_t)b * Ho + oy) * Wo + ox) * Co;
                        for (int c = 0; c < Co; ++c) y[c] = 0.f;
                        for (int ky = 0; ky < 3; ++ky) {
                            int iy = oy * st - pt + ky;
                            if (iy < 0 || iy >= Hi) continue;
                            for (int kx = 0; kx < 3; ++kx) {
                                int ix = ox * st - pl + kx;
                                if (ix < 0 || ix >= Wi) continue;
                                const float *x = X->d + (((size_t)b * Hi + iy) * Wi + ix) * Ci;
                                const float *w = Wt + (size_t)(ky * 3 + kx) * Ci;
#pragma omp simd
                                for (int c = 0; c < Co; ++c) y[c] += x[c] * w[c];
                            }
                        }
                        for (int c = 0; c < Co; ++c) {
                            float v = act_fn(y[c] * sc[c] + bs[c], act, alpha);
                            y[c] = round_out ? f16_round(v) : v;
                        }
                    }
        } break;
        case OP_MAXPOOL: { /* 2x2, keras 'same': window clipped at the bottom/right edge */
#pragma omp parallel for collapse(2) schedule(static)
            for (int b = 0; b < batch; ++b)
                for (int oy = 0; oy < Ho; ++oy)
                    for (int ox = 0; ox < Wo; ++ox) {
                        float *y = Y->d + (((size_t)b * Ho + oy) * Wo + ox) * Co;
                        for (int c = 0; c < Co; ++c) {
                            float m = -INFINITY;
                            for (int ky = 0; ky < 2; ++ky)
                                for (int kx = 0; kx < 2; ++kx) {
                                    int iy = oy * st + ky, ix = ox * st + kx;
                                    if (iy >= Hi || ix >= Wi) continue;
                                    float v = X->d[(((size_t)b * Hi + iy) * Wi + ix) * Ci + c];
                                    if (v > m) m = v;
                                }
                            y[c] = m;
                        }
                    }
        } break;
        case OP_UPSAMPLE: { /* nearest, x2 */
            for (int b = 0; b < batch; ++b)
                for (int oy = 0; oy < Ho; ++oy)
                    for (int ox = 0; ox < Wo; ++ox)
                        memcpy(Y->d + (((size_t)b * Ho + oy) * Wo + ox) * Co,
                               X->d + (((size_t)b * Hi + oy / 2) * Wi + ox / 2) * Ci, sizeof(float) * Ci);
        } break;
        case OP_CONCAT: {
            const tens_t *Z = &T[o[F_IN1]];
            for (size_t p = 0; p < (size_t)batch * Ho * Wo; ++p) {
                memcpy(Y->d + p * Co, X->d + p * Ci, sizeof(float) * Ci);
                memcpy(Y->d + p * Co + Ci, Z->d + p * Z->c, sizeof(float) * Z->c);
            }
        } break;
        case OP_ADD: {
            const tens_t *Z = &T[o[F_IN1]];
            size_t n = (size_t)batch * Ho * Wo * Co;
            for (size_t p = 0; p < n; ++p) {
                float v = X->d[p] + Z->d[p];
                Y->d[p] = round_out ? f16_round(v) : v;
            }
        } break;
        default: rc = -3;
        }
    }
    if (rc == 0) {
        for (int i = 0; i < n_out; ++i) {
            const tens_t *t = &T[out_ids[i]];
            memcpy(out_ptrs[i], t->d, sizeof(float) * (size_t)batch * t->h * t->w * t->c);
        }
        if (dump_id >= 0 && dump_id < n_t && T[dump_id].d)
            memcpy(dump, T[dump_id].d, sizeof(float) * (size_t)batch * T[dump_id].h * T[dump_id].w * T[dump_id].c);
    }
    for (int i = 0; i < n_t; ++i) free(T[i].d);
    free(T);
    free(blob);
    return rc;
}

int yk_ref_forward(const int32_t *ops, int n_ops, const int32_t *tensors, int n_t, const float *blob_in,
                   size_t blob_len, const float *input, int batch, int emulate_f16, const int32_t *out_ids,
                   int n_out, float **out_ptrs, int dump_id, float *dump) {
    const int32_t id0 = 0;
    const float *p0 = input;
    return yk_ref_forward_ex(ops, n_ops, tensors, n_t, blob_in, blob_len, 1, &id0, &p0, batch, emulate_f16, out_ids,
                             n_out, out_ptrs, dump_id, dump);
}

/* Helper._process_img's normalisation (tools/utils.py:405): img / np.max(img), computed in
 * float64 as numpy does and rounded once to fp32. frames: u8 [batch][H][W][3]. */
void yk_ref_normalise_u8(const uint8_t *frames, int batch, size_t per_image, float *out) {
    for (int b = 0; b < batch; ++b) {
        const uint8_t *f = frames + (size_t)b * per_image;
        uint8_t m = 0;
        for (size_t i = 0; i < per_image; ++i)
            if (f[i] > m) m = f[i];
        double dm = (double)m;
        for (size_t i = 0; i < per_image; ++i) out[(size_t)b * per_image + i] = (float)((double)f[i] / dm);
    }
}

float yk_ref_f16_round(float f) { return f16_round(f); }
