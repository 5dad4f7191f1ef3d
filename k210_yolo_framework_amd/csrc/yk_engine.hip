// yk_engine.hip — plan compiler + executor behind yk_plan_create / yk_run_* / yk_get_output.
//
// A plan (k210_yolo_framework_amd/netspec.py) is compiled ONCE into a flat launch list:
//   * UpSampling2D / Concatenate never materialise: they become addressing modes of the consuming
//     conv's A-operand loader (yolonet.py:31-38 head pattern, Darknet FPN :167-172);
//   * Add (MobileNet-v2 / Darknet residual) is folded into the producing conv's epilogue;
//   * DepthwiseConv2D+BN+ReLU followed by its 1x1 Conv2D+BN+LeakyReLU is fused into one launch
//     where profitable (the depthwise tile lives only in LDS) — see YK_FUSE_DWPW;
//   * every weight tensor is converted to fp16 with its reduction axis padded to the activation
//     channel pitch, BatchNorm stays an fp32 (scale, bias) epilogue.
// Running a plan replays the launch list on the caller's stream; nothing is allocated at run time.
#include <algorithm>
#include <string>
#include <vector>

#include "yk_conv.h"

namespace {

uint16_t f2h_bits(float f) {   // round-to-nearest-even fp32 -> fp16 bits
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | ((ax > 0x7f800000u) ? 0x200u : 0));
    if (ax >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);
    if (ax < 0x38800000u) {   // subnormal half
        if (ax < 0x33000000u) return (uint16_t)sign;   // < 2^-25 -> 0 (ties-to-even at exactly 2^-25 -> 0)
        const int e = (int)(ax >> 23);
        uint32_t man = (ax & 0x7fffffu) | 0x800000u;
        const int shift = 126 - e;                     // 14..24
        const uint32_t lsb = 1u << shift, half = lsb >> 1;
        uint32_t r = man >> shift;
        const uint32_t rem = man & (lsb - 1);
        if (rem > half || (rem == half && (r & 1))) ++r;
        return (uint16_t)(sign | r);
    }
    const uint32_t lsb = (ax >> 13) & 1u;
    ax += 0xfffu + lsb;
    return (uint16_t)(sign | ((ax - 0x38000000u) >> 13));
}
float h2f_bits(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1f, m = h & 0x3ffu, x;
    if (e == 0) {
        if (m == 0) x = sign;
        else {
            int sh = 0;
            while (!(m & 0x400u)) {
                m <<= 1;
                ++sh;
            }
            x = sign | ((uint32_t)(113 - sh) << 23) | ((m & 0x3ffu) << 13);
        }
    } else if (e == 31) x = sign | 0x7f800000u | (m << 13);
    else x = sign | ((e + 112) << 23) | (m << 13);
    float f;
    memcpy(&f, &x, 4);
    return f;
}

enum { K_FIRST = 1, K_DW, K_IGEMM, K_POOL, K_ADD, K_U8MAX, K_REDUCE, K_REDUCE_PW };
enum { T_REAL = 0, T_UP = 1, T_CAT = 2 };

struct tinfo {
    int h = 0, w = 0, c = 0, cp = 0;
    int kind = T_REAL, src0 = -1, src1 = -1;
    bool net_out = false, is_input = false;
    yk_half *d = nullptr;
    float *d32 = nullptr;
    int uses = 0;
};

struct launch {
    int kind = 0, cfg = 0;
    igemm_args g;
    igemm_args g2;              // K_REDUCE_PW: the 1x1 conv finished in the same launch
    int Ho2 = 0, Wo2 = 0;
    first_args f;
    dw_args d;
    pool_args p;
    const yk_half *add_a = nullptr, *add_b = nullptr;
    yk_half *add_o = nullptr;
    size_t add_n8_per_image = 0;
    int Ho = 0, Wo = 0;
    bool out_f32 = false;
    std::string name;
    double flops = 0, bytes = 0;
};

}   // namespace

struct yk_plan {
    int device = 0, max_batch = 0;
    std::vector<tinfo> T;
    std::vector<launch> L;
    std::vector<void *> allocs;
    std::vector<int> outputs;
    unsigned *d_imgmax = nullptr;
    float *d_slab = nullptr;
    long long *d_dbg = nullptr;
    int dbg_launch = -1;
    size_t slab_bytes = 0;
    int in_h = 0, in_w = 0;
    int last_batch = 0;
    yk_xplan *x = nullptr;       // precision 1 ("f16x2"): the plan lives in yk_exact.hip, everything below forwards to it
};

static int dev_alloc(yk_plan *p, void **ptr, size_t bytes, bool zero) {
    YK_HIP(hipMalloc(ptr, bytes));
    p->allocs.push_back(*ptr);
    if (zero) YK_HIP(hipMemset(*ptr, 0, bytes));
    return YK_OK;
}
static int upload(yk_plan *p, void **ptr, const void *src, size_t bytes) {
    int rc = dev_alloc(p, ptr, bytes, false);
    if (rc) return rc;
    YK_HIP(hipMemcpy(*ptr, src, bytes, hipMemcpyHostToDevice));
    return YK_OK;
}
// scale/bias padded with zeros so the epilogue may read past N unguarded
static int upload_sb(yk_plan *p, const float *blob, int off, int n, const float **d) {
    std::vector<float> v((size_t)n + 256, 0.f);
    memcpy(v.data(), blob + off, sizeof(float) * n);
    void *q;
    int rc = upload(p, &q, v.data(), v.size() * sizeof(float));
    *d = (const float *)q;
    return rc;
}

static bool env_flag(const char *name, bool dflt) { return yk_env_flag(name, dflt); }

extern "C" void yk_plan_destroy(yk_plan_t *p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    for (void *q : p->allocs) (void)hipFree(q);
    yk_xplan_destroy(p->x);
    delete p;
}

extern "C" int yk_plan_create(yk_plan_t **out, const int32_t *ops, int n_ops, const int32_t *tensors, int n_tensors,
                              const float *blob, size_t blob_len, const int32_t *outputs, int n_outputs, int max_batch,
                              int device) {
    return yk_plan_create_ex(out, ops, n_ops, tensors, n_tensors, blob, blob_len, outputs, n_outputs, max_batch, device, YK_PRECISION_F16X2);
}

extern "C" int yk_plan_create_ex(yk_plan_t **out, const int32_t *ops, int n_ops, const int32_t *tensors, int n_tensors,
                                 const float *blob, size_t blob_len, const int32_t *outputs, int n_outputs, int max_batch,
                                 int device, int precision) {
    if (!out || !ops || !tensors || !blob || !outputs || n_ops <= 0 || n_tensors <= 0 || max_batch <= 0) {
        yk_set_error("yk_plan_create: bad argument");
        return YK_ERR_ARG;
    }
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        yk_set_error("yk_plan_create: no HIP device visible (this library has no CPU path)");
        return YK_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= ndev) {
        yk_set_error("yk_plan_create: device %d out of range (%d visible)", device, ndev);
        return YK_ERR_NO_DEVICE;
    }
    YK_HIP(hipSetDevice(device));
    yk_plan *p = new yk_plan();
    p->device = device;
    p->max_batch = max_batch;
    int rc = YK_OK;
    auto fail = [&](int code) {
        yk_plan_destroy(p);
        return code;
    };

    p->T.resize(n_tensors);
    for (int i = 0; i < n_tensors; ++i) {
        tinfo &t = p->T[i];
        t.h = tensors[4 * i];
        t.w = tensors[4 * i + 1];
        t.c = tensors[4 * i + 2];
        t.cp = yk_pad8(t.c);
        t.is_input = tensors[4 * i + 3] != 0;
    }
    if (!p->T[0].is_input || p->T[0].c != 3) {
        yk_set_error("yk_plan_create: tensor 0 must be the 3-channel network input");
        return fail(YK_ERR_UNSUPPORTED);
    }
    p->in_h = p->T[0].h;
    p->in_w = p->T[0].w;
    for (int i = 0; i < n_outputs; ++i) {
        if (outputs[i] < 0 || outputs[i] >= n_tensors) {
            yk_set_error("yk_plan_create: bad output id");
            return fail(YK_ERR_ARG);
        }
        p->outputs.push_back(outputs[i]);
    }
    const int schedule = precision & YK_SCHEDULE_MASK;
    precision &= ~YK_SCHEDULE_MASK;
    if (precision == YK_PRECISION_F16X2) {
        rc = yk_xplan_create(&p->x, ops, n_ops, tensors, n_tensors, blob, blob_len, outputs, n_outputs, max_batch, schedule == YK_SCHEDULE_LATENCY);
        if (rc) return fail(rc);
        *out = p;
        return YK_OK;
    }
    if (precision != YK_PRECISION_F16) {
        yk_set_error("yk_plan_create_ex: unknown precision %d", precision);
        return fail(YK_ERR_ARG);
    }
    // pass 1: views, use counts, output flags
    for (int i = 0; i < n_ops; ++i) {
        const int32_t *o = ops + (size_t)i * YK_OP_FIELDS;
        const int ty = o[YK_F_TYPE], in0 = o[YK_F_IN0], in1 = o[YK_F_IN1], ot = o[YK_F_OUT];
        if (in0 < 0 || in0 >= n_tensors || ot <= 0 || ot >= n_tensors || in1 >= n_tensors) {
            yk_set_error("yk_plan_create: op %d has a bad tensor id", i);
            return fail(YK_ERR_ARG);
        }
        p->T[in0].uses++;
        if (in1 >= 0) p->T[in1].uses++;
        if (ty == YK_OP_UPSAMPLE) {
            p->T[ot].kind = T_UP;
            p->T[ot].src0 = in0;
        } else if (ty == YK_OP_CONCAT) {
            p->T[ot].kind = T_CAT;
            p->T[ot].src0 = in0;
            p->T[ot].src1 = in1;
        }
        if ((ty == YK_OP_CONV) && (o[YK_F_FLAGS] & YK_FLAG_NET_OUTPUT)) p->T[ot].net_out = true;
        if ((ty == YK_OP_CONV || ty == YK_OP_DWCONV) &&
            ((size_t)std::max(o[YK_F_W_OFF], std::max(o[YK_F_SCALE_OFF], o[YK_F_BIAS_OFF])) >= blob_len ||
             o[YK_F_W_OFF] < 0)) {
            yk_set_error("yk_plan_create: op %d weight offset outside the blob", i);
            return fail(YK_ERR_ARG);
        }
    }
    for (int t : p->outputs) p->T[t].uses++;

    // fusion decisions need the op list; decide ADD-folding first
    std::vector<int> add_of(n_ops, -1);   // conv op i -> index of the ADD folded into it
    std::vector<char> skip(n_ops, 0);
    for (int i = 0; i + 1 < n_ops; ++i) {
        const int32_t *o = ops + (size_t)i * YK_OP_FIELDS, *q = o + YK_OP_FIELDS;
        if (o[YK_F_TYPE] == YK_OP_CONV && q[YK_F_TYPE] == YK_OP_ADD && !(o[YK_F_FLAGS] & YK_FLAG_NET_OUTPUT)) {
            const int y = o[YK_F_OUT];
            const int other = (q[YK_F_IN0] == y) ? q[YK_F_IN1] : (q[YK_F_IN1] == y ? q[YK_F_IN0] : -1);
            if (other >= 0 && other != y && p->T[y].uses == 1 && p->T[other].kind == T_REAL && !p->T[other].is_input) {
                add_of[i] = i + 1;
                skip[i + 1] = 1;
            }
        }
    }
    // depthwise -> pointwise fusion (decided here, realised below)
    const bool fuse_dwpw = env_flag("YK_FUSE_DWPW", true);
    std::vector<int> dw_of(n_ops, -1);    // 1x1 conv op i -> index of the DWCONV fused in front of it
    if (fuse_dwpw) {
        for (int i = 0; i + 1 < n_ops; ++i) {
            const int32_t *o = ops + (size_t)i * YK_OP_FIELDS, *q = o + YK_OP_FIELDS;
            if (o[YK_F_TYPE] == YK_OP_DWCONV && q[YK_F_TYPE] == YK_OP_CONV && q[YK_F_K] == 1 &&
                q[YK_F_IN0] == o[YK_F_OUT] && p->T[o[YK_F_OUT]].uses == 1 && !(q[YK_F_FLAGS] & YK_FLAG_NET_OUTPUT) &&
                p->T[o[YK_F_IN0]].kind == T_REAL && !p->T[o[YK_F_IN0]].is_input && o[YK_F_ACT] != YK_ACT_LEAKY &&
                yk_igemm_fused_ok(yk_pad8(o[YK_F_CIN]), q[YK_F_COUT])) {
                dw_of[i + 1] = i;
                skip[i] = 1;
            }
        }
    }

    // allocate real tensors
    for (int i = 1; i < n_tensors; ++i) {
        tinfo &t = p->T[i];
        if (t.kind != T_REAL) continue;
        bool produced_fused_away = false;
        for (int k = 0; k < n_ops; ++k) {
            const int32_t *o = ops + (size_t)k * YK_OP_FIELDS;
            if (o[YK_F_OUT] == i && ((skip[k] && o[YK_F_TYPE] == YK_OP_DWCONV) || add_of[k] >= 0)) produced_fused_away = true;
        }
        if (produced_fused_away) continue;   // lives only in LDS / registers
        if (t.net_out) {
            rc = dev_alloc(p, (void **)&t.d32, (size_t)max_batch * t.h * t.w * t.c * sizeof(float), true);
        } else {
            rc = dev_alloc(p, (void **)&t.d, ((size_t)max_batch * t.h * t.w * t.cp + 64) * sizeof(yk_half), true);
        }
        if (rc) return fail(rc);
    }
    rc = dev_alloc(p, (void **)&p->d_imgmax, sizeof(unsigned) * max_batch * 32, true);   // YK_MAXP partials per image
    if (rc) return fail(rc);

    // pass 2: launches
    {
        launch l;
        l.kind = K_U8MAX;   // only issued by yk_run_u8: Helper._process_img's np.max(img)
        l.name = "u8_max";
        l.bytes = (double)p->in_h * p->in_w * 3;
        p->L.push_back(l);
    }
    for (int i = 0; i < n_ops; ++i) {
        if (skip[i]) continue;
        const int32_t *o = ops + (size_t)i * YK_OP_FIELDS;
        const int ty = o[YK_F_TYPE];
        if (ty == YK_OP_UPSAMPLE || ty == YK_OP_CONCAT) continue;
        const tinfo &X = p->T[o[YK_F_IN0]];
        tinfo &Y = p->T[o[YK_F_OUT]];
        float alpha;
        memcpy(&alpha, &o[YK_F_ALPHA], 4);
        launch l;
        l.Ho = Y.h;
        l.Wo = Y.w;
        char nm[96];
        if (ty == YK_OP_CONV && X.is_input) {
            // ---- stem conv
            if (o[YK_F_K] != 3 || add_of[i] >= 0 || Y.net_out) {
                yk_set_error("op %d: stem conv must be 3x3", i);
                return fail(YK_ERR_UNSUPPORTED);
            }
            const int co = o[YK_F_COUT];
            std::vector<float> w((size_t)27 * co);
            for (int c = 0; c < co; ++c)
                for (int t = 0; t < 27; ++t) w[(size_t)t * co + c] = h2f_bits(f2h_bits(blob[o[YK_F_W_OFF] + (size_t)c * 27 + t]));
            void *dw_;
            if ((rc = upload(p, &dw_, w.data(), w.size() * sizeof(float)))) return fail(rc);
            l.kind = K_FIRST;
            first_args &f = l.f;
            memset(&f, 0, sizeof(f));
            f.Hi = X.h; f.Wi = X.w; f.Ho = Y.h; f.Wo = Y.w;
            f.stride = o[YK_F_STRIDE]; f.pad_t = o[YK_F_PAD_T]; f.pad_l = o[YK_F_PAD_L];
            f.Cout = co; f.outp = Y.cp; f.w = (const float *)dw_;
            if (co <= 32) {                                    // weights again, in the MFMA stem's k order
                std::vector<uint16_t> wm(32 * 32, 0);
                for (int n = 0; n < co; ++n)
                    for (int ky = 0; ky < 3; ++ky)
                        for (int j = 0; j < 9; ++j)
                            wm[(size_t)n * 32 + (j < 8 ? ky * 8 + j : 24 + ky)] = f2h_bits(blob[o[YK_F_W_OFF] + (size_t)n * 27 + ky * 9 + j]);
                void *dm;
                if ((rc = upload(p, &dm, wm.data(), wm.size() * 2))) return fail(rc);
                f.wm = (const yk_half *)dm;
            }
            if ((rc = upload_sb(p, blob, o[YK_F_SCALE_OFF], co, &f.scale))) return fail(rc);
            if ((rc = upload_sb(p, blob, o[YK_F_BIAS_OFF], co, &f.bias))) return fail(rc);
            f.act = o[YK_F_ACT]; f.alpha = alpha; f.out = Y.d;
            yk_act_params(f.act, f.alpha, &f.slope, &f.cap);
            if (Y.cp != co) {
                yk_set_error("op %d: stem conv Cout must be a multiple of 8", i);
                return fail(YK_ERR_UNSUPPORTED);
            }
            snprintf(nm, sizeof nm, "stem3x3s%d_%d", f.stride, co);
            l.flops = 2.0 * Y.h * Y.w * 27 * co;
            l.bytes = (double)X.h * X.w * 3 * 2 + (double)Y.h * Y.w * co * 2;
        } else if (ty == YK_OP_CONV) {
            // ---- implicit GEMM conv (+ folded Add, + fused depthwise producer)
            l.kind = K_IGEMM;
            igemm_args &g = l.g;
            memset(&g, 0, sizeof(g));
            g.lda_pad = yk_fused_pad();
            const tinfo *s0 = &X, *s1 = nullptr;
            int up0 = 0;
            if (X.kind == T_CAT) {
                s0 = &p->T[X.src0];
                s1 = &p->T[X.src1];
            }
            if (s0->kind == T_UP) {
                up0 = 1;
                s0 = &p->T[s0->src0];
            }
            const int dwi = dw_of[i];
            const int32_t *dwo = dwi >= 0 ? ops + (size_t)dwi * YK_OP_FIELDS : nullptr;
            const tinfo *dwX = dwo ? &p->T[dwo[YK_F_IN0]] : nullptr;
            if (dwo) s0 = dwX;
            if (s0->kind != T_REAL || (s1 && s1->kind != T_REAL) || s0->is_input || (s1 && s1->is_input) || !s0->d ||
                (s1 && !s1->d)) {
                yk_set_error("op %d: unsupported input view nesting", i);
                return fail(YK_ERR_UNSUPPORTED);
            }
            const int ks = o[YK_F_K], co = o[YK_F_COUT];
            const int c0 = dwo ? X.c : s0->c, c0p = yk_pad8(c0), c1 = s1 ? s1->c : 0, c1p = s1 ? s1->cp : 0;
            if (c0 + c1 != o[YK_F_CIN] || (ks != 1 && ks != 3)) {
                yk_set_error("op %d: conv shape mismatch (cin %d vs %d+%d, k=%d)", i, o[YK_F_CIN], c0, c1, ks);
                return fail(YK_ERR_UNSUPPORTED);
            }
            g.in0 = s0->d; g.in1 = s1 ? s1->d : nullptr;
            g.c0p = c0p; g.c1p = c1p; g.up0 = up0;
            g.Hi = X.h; g.Wi = X.w; g.Ho = Y.h; g.Wo = Y.w;
            g.ks = ks; g.stride = o[YK_F_STRIDE]; g.pad_t = o[YK_F_PAD_T]; g.pad_l = o[YK_F_PAD_L];
            g.N = co; g.K = ks * ks * (c0p + c1p);
            std::vector<uint16_t> w((size_t)co * g.K, 0);
            const int cin = o[YK_F_CIN];
            for (int n = 0; n < co; ++n)
                for (int t = 0; t < ks * ks; ++t)
                    for (int c = 0; c < cin; ++c) {
                        const int pos = c < c0 ? c : c0p + (c - c0);
                        w[(size_t)n * g.K + (size_t)t * (c0p + c1p) + pos] =
                            f2h_bits(blob[o[YK_F_W_OFF] + ((size_t)n * ks * ks + t) * cin + c]);
                    }
            void *dwt;
            if ((rc = upload(p, &dwt, w.data(), w.size() * 2))) return fail(rc);
            g.w = (const yk_half *)dwt;
            g.w_bytes = (uint32_t)(w.size() * 2);
            g.wfrag = nullptr; g.wfrag_bytes = 0; g.nb16 = (co + 15) / 16;
            if (g.K % 64 == 0 && yk_dev_env("YK_PIPE_BR") && yk_dev_env("YK_PIPE_BR")[0] == '1') {     // fragment-order copy for yk_igemm_br.h (developer builds)
                const int nb = g.nb16, nsteps = g.K / 64;
                std::vector<uint16_t> wf((size_t)nsteps * nb * 1024, 0);
                for (int st = 0; st < nsteps; ++st)
                    for (int b = 0; b < nb; ++b)
                        for (int hs = 0; hs < 2; ++hs)
                            for (int ln = 0; ln < 64; ++ln) {
                                const int n = b * 16 + (ln & 15), k0 = st * 64 + (hs * 4 + (ln >> 4)) * 8;
                                if (n >= co) continue;
                                for (int e = 0; e < 8; ++e)
                                    wf[(((size_t)st * nb + b) * 2 + hs) * 512 + (size_t)ln * 8 + e] = w[(size_t)n * g.K + k0 + e];
                            }
                void *dfr;
                if ((rc = upload(p, &dfr, wf.data(), wf.size() * 2))) return fail(rc);
                g.wfrag = (const yk_half *)dfr;
                g.wfrag_bytes = (uint32_t)(wf.size() * 2);
            }
            if ((rc = upload_sb(p, blob, o[YK_F_SCALE_OFF], co, &g.scale))) return fail(rc);
            if ((rc = upload_sb(p, blob, o[YK_F_BIAS_OFF], co, &g.bias))) return fail(rc);
            g.act = o[YK_F_ACT]; g.alpha = alpha;
            yk_act_params(g.act, g.alpha, &g.slope, &g.cap);
            g.fd_hw = yk_make_fastdiv((uint32_t)(Y.h * Y.w));
            g.fd_wo = yk_make_fastdiv((uint32_t)Y.w);
            g.fd_ctp = yk_make_fastdiv((uint32_t)(c0p + c1p));
            g.split_k = 1;
            g.in0_bytes = (uint32_t)std::min<size_t>((size_t)max_batch * s0->h * s0->w * s0->cp * 2, 0xffffffffu);
            g.in1_bytes = s1 ? (uint32_t)std::min<size_t>((size_t)max_batch * s1->h * s1->w * s1->cp * 2, 0xffffffffu) : 0u;
            if (g.in0_bytes >= 0x40000000u || g.in1_bytes >= 0x40000000u) {
                yk_set_error("op %d: activation tensor >= 1 GiB; lower max_batch", i);
                return fail(YK_ERR_UNSUPPORTED);
            }
            tinfo *dst = &Y;
            if (add_of[i] >= 0) {
                const int32_t *q = ops + (size_t)add_of[i] * YK_OP_FIELDS;
                const int other = (q[YK_F_IN0] == o[YK_F_OUT]) ? q[YK_F_IN1] : q[YK_F_IN0];
                g.res = p->T[other].d;
                g.resp = p->T[other].cp;
                dst = &p->T[q[YK_F_OUT]];
            }
            const bool f32 = dst->net_out;
            g.out = f32 ? (void *)dst->d32 : (void *)dst->d;
            g.outp = f32 ? dst->c : dst->cp;
            g.fd_vpr = yk_make_fastdiv((uint32_t)std::max(1, g.outp >> 3));
            if (!g.out) {
                yk_set_error("op %d: output tensor not allocated", i);
                return fail(YK_ERR_UNSUPPORTED);
            }
            double dw_flops = 0, dw_bytes = 0;
            if (dwo) {
                float dalpha;
                memcpy(&dalpha, &dwo[YK_F_ALPHA], 4);
                (void)dalpha;
                std::vector<uint16_t> dww((size_t)9 * c0p, 0);
                for (int t = 0; t < 9; ++t)
                    for (int c = 0; c < c0; ++c) dww[(size_t)t * c0p + c] = f2h_bits(blob[dwo[YK_F_W_OFF] + (size_t)t * c0 + c]);
                void *dd;
                if ((rc = upload(p, &dd, dww.data(), dww.size() * 2))) return fail(rc);
                g.dw_w = (const yk_half *)dd;
                if ((rc = upload_sb(p, blob, dwo[YK_F_SCALE_OFF], c0, &g.dw_scale))) return fail(rc);
                if ((rc = upload_sb(p, blob, dwo[YK_F_BIAS_OFF], c0, &g.dw_bias))) return fail(rc);
                g.dw_act = dwo[YK_F_ACT]; g.dw_stride = dwo[YK_F_STRIDE];
                g.dw_pad_t = dwo[YK_F_PAD_T]; g.dw_pad_l = dwo[YK_F_PAD_L];
                g.dw_Hi = dwX->h; g.dw_Wi = dwX->w;
                yk_act_params(g.dw_act, 0.f, &g.dw_slope, &g.dw_cap);
                g.fd_g = yk_make_fastdiv((uint32_t)(c0p >> 3));
                dw_flops = 2.0 * X.h * X.w * 9 * c0;
                dw_bytes = ((double)dwX->h * dwX->w * c0 + (double)X.h * X.w * c0) * 2;
            }
            g.M = max_batch * Y.h * Y.w;   // for config choice; patched per run
            l.cfg = dwo ? yk_igemm_fused_pick(g) : yk_igemm_pick(g, f32);
            l.out_f32 = f32;
            if (dwo && (l.cfg == FUSED_DMA || l.cfg == LR_T1 || l.cfg == LR_T2)) {
                if (l.cfg == FUSED_DMA) yk_fdma_fill(g);
                // the LDS-DMA staged kernel and the LR kernels read their pointwise panel in MFMA fragment order: [16-channel slice][k-step of 32][lane][8],
                // element = W[slice*16 + (lane & 15)][kstep*32 + (lane >> 4)*8 + e]; one wave-load = 1 KB contiguous
                const int Kp = (c0p + 31) & ~31, nkf = Kp / 32, nsl = (co + 15) / 16;
                std::vector<uint16_t> wf((size_t)nsl * nkf * 512, 0);
                for (int sl = 0; sl < nsl; ++sl)
                    for (int ks2 = 0; ks2 < nkf; ++ks2)
                        for (int ln = 0; ln < 64; ++ln)
                            for (int e = 0; e < 8; ++e) {
                                const int n = sl * 16 + (ln & 15), k = ks2 * 32 + (ln >> 4) * 8 + e;
                                if (n < co && k < g.K) wf[(((size_t)sl * nkf + ks2) * 64 + ln) * 8 + e] = w[(size_t)n * g.K + k];
                            }
                void *dwf;
                if ((rc = upload(p, &dwf, wf.data(), wf.size() * 2))) return fail(rc);
                g.w = (const yk_half *)dwf;
                g.w_bytes = (uint32_t)(wf.size() * 2);
            }
            if (!dwo && env_flag("YK_SPLITK", true)) {
                g.split_k = yk_igemm_split(l.cfg, g);
                if (g.split_k > 1) {
                    g.ldn = (co + 15) & ~15;
                    p->slab_bytes = std::max(p->slab_bytes, (size_t)g.split_k * g.M * g.ldn * sizeof(float));
                }
            }
            snprintf(nm, sizeof nm, "%sconv%dx%ds%d_%dto%d%s%s[%s]", dwo ? "dw3x3+" : "", ks, ks, g.stride, o[YK_F_CIN], co,
                     g.res ? "+add" : "", s1 ? "+upcat" : (up0 ? "+up" : ""),
                     dwo ? yk_igemm_fused_name(l.cfg) : yk_igemm_name(l.cfg));
            if (g.split_k > 1) snprintf(nm + strlen(nm), sizeof nm - strlen(nm), "/splitk%d", g.split_k);
            l.flops = 2.0 * Y.h * Y.w * ks * ks * (double)o[YK_F_CIN] * co + dw_flops;
            l.bytes = ((double)X.h * X.w * o[YK_F_CIN] + (double)Y.h * Y.w * co) * 2 + dw_bytes;
        } else if (ty == YK_OP_DWCONV) {
            if (X.kind != T_REAL || X.is_input) {
                yk_set_error("op %d: depthwise conv on a view/input", i);
                return fail(YK_ERR_UNSUPPORTED);
            }
            l.kind = K_DW;
            dw_args &d = l.d;
            memset(&d, 0, sizeof(d));
            const int c = X.c, cp = X.cp;
            std::vector<uint16_t> w((size_t)9 * cp, 0);
            for (int t = 0; t < 9; ++t)
                for (int k = 0; k < c; ++k) w[(size_t)t * cp + k] = f2h_bits(blob[o[YK_F_W_OFF] + (size_t)t * c + k]);
            void *dd;
            if ((rc = upload(p, &dd, w.data(), w.size() * 2))) return fail(rc);
            d.in = X.d; d.Hi = X.h; d.Wi = X.w; d.Ho = Y.h; d.Wo = Y.w; d.Cp = cp;
            d.stride = o[YK_F_STRIDE]; d.pad_t = o[YK_F_PAD_T]; d.pad_l = o[YK_F_PAD_L];
            d.w = (const yk_half *)dd;
            if ((rc = upload_sb(p, blob, o[YK_F_SCALE_OFF], c, &d.scale))) return fail(rc);
            if ((rc = upload_sb(p, blob, o[YK_F_BIAS_OFF], c, &d.bias))) return fail(rc);
            d.act = o[YK_F_ACT]; d.alpha = alpha; d.out = Y.d;
            yk_act_params(d.act, d.alpha, &d.slope, &d.cap);
            snprintf(nm, sizeof nm, "dw3x3s%d_%d", d.stride, c);
            l.flops = 2.0 * Y.h * Y.w * 9 * c;
            l.bytes = ((double)X.h * X.w * c + (double)Y.h * Y.w * c) * 2;
        } else if (ty == YK_OP_MAXPOOL) {
            if (X.kind != T_REAL || X.is_input) {
                yk_set_error("op %d: max pool on a view/input", i);
                return fail(YK_ERR_UNSUPPORTED);
            }
            l.kind = K_POOL;
            pool_args &q = l.p;
            memset(&q, 0, sizeof(q));
            q.in = X.d; q.Hi = X.h; q.Wi = X.w; q.Ho = Y.h; q.Wo = Y.w; q.Cp = X.cp; q.stride = o[YK_F_STRIDE]; q.out = Y.d;
            snprintf(nm, sizeof nm, "maxpool2x2s%d_%d", q.stride, X.c);
            l.bytes = ((double)X.h * X.w * X.c + (double)Y.h * Y.w * Y.c) * 2;
        } else if (ty == YK_OP_ADD) {
            const tinfo &Z = p->T[o[YK_F_IN1]];
            if (X.kind != T_REAL || Z.kind != T_REAL || !X.d || !Z.d || !Y.d) {
                yk_set_error("op %d: standalone Add on views", i);
                return fail(YK_ERR_UNSUPPORTED);
            }
            l.kind = K_ADD;
            l.add_a = X.d; l.add_b = Z.d; l.add_o = Y.d;
            l.add_n8_per_image = (size_t)Y.h * Y.w * Y.cp / 8;
            snprintf(nm, sizeof nm, "add_%d", Y.c);
            l.bytes = 3.0 * Y.h * Y.w * Y.c * 2;
        } else {
            yk_set_error("op %d: unknown op type %d", i, ty);
            return fail(YK_ERR_UNSUPPORTED);
        }
        l.name = nm;
        p->L.push_back(l);
        if (l.kind == K_IGEMM && l.g.split_k > 1) {   // deterministic finishing pass of split-K, its own launch
            launch r = l;
            r.kind = K_REDUCE;
            r.name = std::string("splitk_reduce") + std::to_string(l.g.split_k) + "_" + std::to_string(l.g.N);
            r.flops = 0;
            r.bytes = ((double)l.g.split_k * 4 + 2) * l.Ho * l.Wo * l.g.ldn;   // slabs read (fp32) + tile written (fp16)
            p->L.push_back(r);
        }
    }
    // split-K finishing pass followed by the 1x1 fp32 head conv that reads it -> one launch (reduce_pw_kernel)
    if (env_flag("YK_REDUCE_PW", true)) {
        for (size_t i = 0; i + 1 < p->L.size(); ++i) {
            launch &r = p->L[i];
            launch &c = p->L[i + 1];
            if (r.kind != K_REDUCE || r.out_f32 || c.kind != K_IGEMM) continue;
            igemm_args cg = c.g;
            cg.M = r.g.M;
            if (!yk_reduce_pw_ok(r.g, cg, c.out_f32)) continue;
            r.kind = K_REDUCE_PW;
            r.g2 = c.g;
            r.Ho2 = c.Ho; r.Wo2 = c.Wo;
            r.name += "+" + c.name.substr(0, c.name.find('['));
            r.flops += c.flops;
            r.bytes += c.bytes;
            p->L.erase(p->L.begin() + i + 1);
        }
    }
    if (p->slab_bytes) {
        if ((rc = dev_alloc(p, (void **)&p->d_slab, p->slab_bytes, true))) return fail(rc);
    }
    for (int t : p->outputs)
        if (!p->T[t].d32) {
            yk_set_error("yk_plan_create: output tensor %d is not produced by a NET_OUTPUT conv", t);
            return fail(YK_ERR_UNSUPPORTED);
        }
    YK_HIP(hipDeviceSynchronize());
    *out = p;
    return YK_OK;
}

static int run_plan(yk_plan *p, const void *d_in, int in_f32, int batch, void *stream, hipEvent_t *ev = nullptr) {
    if (!p || !d_in || batch <= 0 || batch > p->max_batch) {
        yk_set_error("yk_run: bad plan/input/batch (max_batch=%d)", p ? p->max_batch : 0);
        return YK_ERR_ARG;
    }
    YK_HIP(hipSetDevice(p->device));
    hipStream_t st = (hipStream_t)stream;
    if (p->x) {
        p->last_batch = batch;
        return yk_xplan_run(p->x, d_in, in_f32, batch, st, ev);
    }
    int li = 0;
    for (launch &l : p->L) {
        int rc = YK_OK;
        if (ev) YK_HIP(hipEventRecord(ev[2 * li], st));
        switch (l.kind) {
        case K_U8MAX:
            if (!in_f32) {
                rc = yk_launch_u8_max((const uint8_t *)d_in, (size_t)p->in_h * p->in_w * 3, batch, p->d_imgmax, st);
            }
            break;
        case K_FIRST: {
            first_args f = l.f;
            f.in = d_in; f.in_f32 = in_f32; f.img_max = p->d_imgmax; f.B = batch;
            rc = yk_launch_first(f, st);
        } break;
        case K_IGEMM: {
            igemm_args g = l.g;
            g.M = batch * l.Ho * l.Wo;
            g.slab = p->d_slab;
            g.dbg = (li == p->dbg_launch) ? p->d_dbg : nullptr;
            rc = g.dw_w ? yk_launch_igemm_fused(l.cfg, g, st) : yk_launch_igemm(l.cfg, g, st);
        } break;
        case K_REDUCE: {
            igemm_args g = l.g;
            g.M = batch * l.Ho * l.Wo;
            g.slab = p->d_slab;
            rc = yk_launch_splitk_reduce(g, l.out_f32, st);
        } break;
        case K_REDUCE_PW: {
            igemm_args g = l.g, g2 = l.g2;
            g.M = batch * l.Ho * l.Wo;
            g.slab = p->d_slab;
            g2.M = batch * l.Ho2 * l.Wo2;
            rc = yk_launch_reduce_pw(g, g2, st);
        } break;
        case K_DW: {
            dw_args d = l.d;
            d.B = batch;
            rc = yk_launch_dw(d, st);
        } break;
        case K_POOL: {
            pool_args q = l.p;
            q.B = batch;
            rc = yk_launch_pool(q, st);
        } break;
        case K_ADD: rc = yk_launch_add(l.add_a, l.add_b, l.add_o, l.add_n8_per_image * batch, st); break;
        }
        if (rc) return rc;
        if (ev) YK_HIP(hipEventRecord(ev[2 * li + 1], st));
        ++li;
    }
    YK_HIP(hipGetLastError());
    p->last_batch = batch;
    return YK_OK;
}

// Per-launch timing with HIP events recorded on the SAME stream the kernels run on.
// ms_out[i] = median duration of launch i over `iters` replays (i < yk_plan_launch_count).
extern "C" int yk_plan_profile(yk_plan_t *p, const uint8_t *d_frames, int batch, int iters, void *stream, float *ms_out) {
    if (!p || !ms_out || iters <= 0) {
        yk_set_error("yk_plan_profile: bad argument");
        return YK_ERR_ARG;
    }
    const int n = yk_plan_launch_count(p);
    std::vector<hipEvent_t> ev(2 * n);
    for (auto &e : ev) YK_HIP(hipEventCreate(&e));
    std::vector<std::vector<float>> samples(n);
    int rc = YK_OK;
    for (int it = -3; it < iters && rc == YK_OK; ++it) {                  // three untimed replays first (clocks, caches, allocator)
        rc = run_plan(p, d_frames, 0, batch, stream, ev.data());
        if (rc) break;
        YK_HIP(hipStreamSynchronize((hipStream_t)stream));
        if (it < 0) continue;
        for (int i = 0; i < n; ++i) {
            float ms = 0.f;
            YK_HIP(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]));
            samples[i].push_back(ms);
        }
    }
    for (auto &e : ev) (void)hipEventDestroy(e);
    for (int i = 0; i < n; ++i) {                                         // median over the replays: one slow replay (a clock dip, another
        std::vector<float> &v = samples[i];                               // process' burst) must not define a launch's duration
        std::sort(v.begin(), v.end());
        ms_out[i] = v.empty() ? 0.f : (v.size() & 1 ? v[v.size() / 2] : 0.5f * (v[v.size() / 2 - 1] + v[v.size() / 2]));
    }
    return rc;
}

extern "C" int yk_run_u8(yk_plan_t *p, const uint8_t *d_frames, int batch, void *stream) {
    return run_plan(p, d_frames, 0, batch, stream);
}
extern "C" int yk_run_f32(yk_plan_t *p, const float *d_input, int batch, void *stream) {
    return run_plan(p, d_input, 1, batch, stream);
}

extern "C" int yk_get_output(yk_plan_t *p, int idx, float **d_ptr, size_t *bytes, int *h, int *w, int *c) {
    if (!p || idx < 0 || idx >= (int)p->outputs.size()) {
        yk_set_error("yk_get_output: bad index");
        return YK_ERR_ARG;
    }
    if (p->x) return yk_xplan_output(p->x, idx, d_ptr, bytes, h, w, c);
    const tinfo &t = p->T[p->outputs[idx]];
    if (d_ptr) *d_ptr = t.d32;
    if (bytes) *bytes = (size_t)p->max_batch * t.h * t.w * t.c * sizeof(float);
    if (h) *h = t.h;
    if (w) *w = t.w;
    if (c) *c = t.c;
    return YK_OK;
}

extern "C" int yk_debug_read_tensor(yk_plan_t *p, int tid, int batch, float *h_dst, size_t dst_elems) {
    if (!p || tid <= 0 || tid >= (int)p->T.size() || batch <= 0 || batch > p->max_batch || !h_dst) {
        yk_set_error("yk_debug_read_tensor: bad argument");
        return YK_ERR_ARG;
    }
    if (p->x) {
        YK_HIP(hipSetDevice(p->device));
        return yk_xplan_read_tensor(p->x, tid, batch, h_dst, dst_elems);
    }
    const tinfo &t = p->T[tid];
    const size_t n = (size_t)batch * t.h * t.w * t.c;
    if (dst_elems < n) {
        yk_set_error("yk_debug_read_tensor: destination too small");
        return YK_ERR_ARG;
    }
    YK_HIP(hipSetDevice(p->device));
    YK_HIP(hipDeviceSynchronize());
    if (t.d32) {
        YK_HIP(hipMemcpy(h_dst, t.d32, n * sizeof(float), hipMemcpyDeviceToHost));
        return YK_OK;
    }
    if (!t.d) {
        yk_set_error("yk_debug_read_tensor: tensor %d is a view or was fused away", tid);
        return YK_ERR_UNSUPPORTED;
    }
    std::vector<uint16_t> h((size_t)batch * t.h * t.w * t.cp);
    YK_HIP(hipMemcpy(h.data(), t.d, h.size() * 2, hipMemcpyDeviceToHost));
    const size_t pix = (size_t)batch * t.h * t.w;
    for (size_t q = 0; q < pix; ++q)
        for (int c = 0; c < t.c; ++c) h_dst[q * t.c + c] = h2f_bits(h[q * t.cp + c]);
    return YK_OK;
}

// dev instrumentation: arm phase timestamps for launch `li`, run once (u8 path), copy out [n_wg][8] ticks (100 MHz)
extern "C" int yk_debug_phase_stamps(yk_plan_t *p, int li, const uint8_t *d_frames, int batch, void *stream,
                                     long long *h_out, int max_wg) {
    if (p && p->x) {       // f16x2 plan: [n_wg][16] stamps of a fused block launch
        YK_HIP(hipSetDevice(p->device));
        return yk_xplan_phase_stamps(p->x, li, d_frames, batch, (hipStream_t)stream, h_out, max_wg);
    }
    if (!p || li < 0 || li >= (int)p->L.size()) return YK_ERR_ARG;
    YK_HIP(hipSetDevice(p->device));
    if (!p->d_dbg) {
        int rc = dev_alloc(p, (void **)&p->d_dbg, sizeof(long long) * 8 * 65536, true);
        if (rc) return rc;
    }
    YK_HIP(hipMemset(p->d_dbg, 0, sizeof(long long) * 8 * 65536));
    p->dbg_launch = li;
    int rc = run_plan(p, d_frames, 0, batch, stream);
    p->dbg_launch = -1;
    if (rc) return rc;
    YK_HIP(hipStreamSynchronize((hipStream_t)stream));
    if (max_wg > 65536) max_wg = 65536;
    YK_HIP(hipMemcpy(h_out, p->d_dbg, sizeof(long long) * 8 * max_wg, hipMemcpyDeviceToHost));
    return YK_OK;
}

// Waits for the device and reports a sticky device-side failure of an earlier run of this plan (today: the persistent late-backbone
// stage of the f16x2 mode could not assemble a workgroup cluster; see yk_xpersist.h).  YK_OK otherwise.
extern "C" int yk_plan_check(yk_plan_t *p) {
    if (!p) {
        yk_set_error("yk_plan_check: bad argument");
        return YK_ERR_ARG;
    }
    YK_HIP(hipSetDevice(p->device));
    if (p->x) return yk_xplan_check(p->x);
    YK_HIP(hipDeviceSynchronize());
    return YK_OK;
}

// The same word WITHOUT waiting for the device: nonzero = a run that has already finished failed on the device (today: a cluster launch
// of the f16x2 latency schedule could not assemble a cluster).  The caller has synchronised with the runs it asks about (an event, a
// stream); `clear` resets the word after reading it.  Costs a host memory read: the word lives in mapped host memory.
extern "C" int yk_plan_peek_error(yk_plan_t *p, int clear, unsigned *error_out) {
    if (!p || !error_out) {
        yk_set_error("yk_plan_peek_error: bad argument");
        return YK_ERR_ARG;
    }
    *error_out = p->x ? yk_xplan_peek_error(p->x, clear) : 0u;
    return YK_OK;
}

// test hook: stores `value` into the error word the way a timed-out cluster barrier does
extern "C" int yk_plan_debug_set_error(yk_plan_t *p, unsigned value) {
    if (!p || !p->x) {
        yk_set_error("yk_plan_debug_set_error: needs an f16x2 plan");
        return YK_ERR_ARG;
    }
    yk_xplan_debug_set_error(p->x, value);
    return YK_OK;
}

extern "C" int yk_plan_launch_count(const yk_plan_t *p) { return !p ? 0 : (p->x ? yk_xplan_launch_count(p->x) : (int)p->L.size()); }

extern "C" int yk_plan_launch_info(const yk_plan_t *p, int i, char *name, size_t name_len, double *flops_per_image,
                                   double *bytes_per_image) {
    if (p && p->x) {
        const char *nm = "";
        double fl = 0, by = 0;
        int rc = yk_xplan_launch_info(p->x, i, &nm, &fl, &by);
        if (rc) return rc;
        if (name && name_len) snprintf(name, name_len, "%s", nm);
        if (flops_per_image) *flops_per_image = fl;
        if (bytes_per_image) *bytes_per_image = by;
        return YK_OK;
    }
    if (!p || i < 0 || i >= (int)p->L.size()) return YK_ERR_ARG;
    if (name && name_len) snprintf(name, name_len, "%s", p->L[i].name.c_str());
    if (flops_per_image) *flops_per_image = p->L[i].flops;
    if (bytes_per_image) *bytes_per_image = p->L[i].bytes;
    return YK_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// A step as ONE hipGraph.  Everything issued on `stream` between yk_graph_begin and yk_graph_end (yk_run_*, yk_decode_py*,
// yk_letterbox_u8, yk_memcpy_async ...) is recorded instead of executed; yk_graph_launch replays it with one host call
// (the eager step is ~30 launches at 3-4 us of host time each: SURVEY 7 step 8).  The capture is thread-local: other host threads
// (an input pipeline's producer) may keep allocating and copying on their own streams meanwhile.  A replay uses the pointers and
// sizes of the capture: the step must have run once eagerly on the same stream before (the decode scratch is allocated on first use
// and is keyed by stream), the buffers it names must stay alive, and the plan must not be replayed on two streams at once.
// ---------------------------------------------------------------------------------------------------------------------
struct yk_graph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    size_t nodes = 0, kernels = 0;
};

extern "C" int yk_graph_begin(void *stream) {
    if (!stream) {
        yk_set_error("yk_graph_begin: the default stream cannot be captured; pass a created stream");
        return YK_ERR_ARG;
    }
    YK_HIP(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
    return YK_OK;
}

extern "C" int yk_graph_end(void *stream, yk_graph_t **out) {
    if (!out) {
        yk_set_error("yk_graph_end: bad argument");
        return YK_ERR_ARG;
    }
    *out = nullptr;
    hipGraph_t g = nullptr;
    YK_HIP(hipStreamEndCapture((hipStream_t)stream, &g));
    if (!g) {
        yk_set_error("yk_graph_end: the capture was invalidated (a call that cannot be captured ran on the stream)");
        return YK_ERR_HIP;
    }
    yk_graph *r = new yk_graph();
    r->graph = g;
    hipError_t e = hipGraphInstantiate(&r->exec, g, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        yk_set_error("yk_graph_end: hipGraphInstantiate -> %s", hipGetErrorString(e));
        (void)hipGraphDestroy(g);
        delete r;
        return YK_ERR_HIP;
    }
    if (hipGraphGetNodes(g, nullptr, &r->nodes) == hipSuccess && r->nodes) {
        std::vector<hipGraphNode_t> nd(r->nodes);
        size_t n = r->nodes;
        if (hipGraphGetNodes(g, nd.data(), &n) == hipSuccess)
            for (size_t i = 0; i < n; ++i) {
                hipGraphNodeType ty;
                if (hipGraphNodeGetType(nd[i], &ty) == hipSuccess && ty == hipGraphNodeTypeKernel) ++r->kernels;
            }
    }
    *out = r;
    return YK_OK;
}

extern "C" int yk_graph_launch(yk_graph_t *g, void *stream) {
    if (!g || !g->exec) {
        yk_set_error("yk_graph_launch: bad graph");
        return YK_ERR_ARG;
    }
    YK_HIP(hipGraphLaunch(g->exec, (hipStream_t)stream));
    return YK_OK;
}

extern "C" int yk_graph_node_count(const yk_graph_t *g) { return g ? (int)g->nodes : 0; }
extern "C" int yk_graph_kernel_node_count(const yk_graph_t *g) { return g ? (int)g->kernels : 0; }

extern "C" void yk_graph_destroy(yk_graph_t *g) {
    if (!g) return;
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
}

extern "C" int yk_memcpy_async(void *dst, const void *src, size_t bytes, void *stream) {
    if (!dst || !src) {
        yk_set_error("yk_memcpy_async: bad argument");
        return YK_ERR_ARG;
    }
    YK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, (hipStream_t)stream));
    return YK_OK;
}

// device address of pinned host memory (hipHostMalloc / hipHostRegister'ed): what a kernel must be given to write detections
// straight into host memory (yk_decode_py_packed)
extern "C" int yk_host_device_ptr(void *h_ptr, void **d_ptr) {
    if (!h_ptr || !d_ptr) {
        yk_set_error("yk_host_device_ptr: bad argument");
        return YK_ERR_ARG;
    }
    YK_HIP(hipHostGetDevicePointer(d_ptr, h_ptr, 0));
    return YK_OK;
}

// library-owned streams (see include/yolo_hip.h): created back to back they land on consecutive hardware queues
extern "C" int yk_stream_create(void **stream_out, int priority) {
    if (!stream_out) {
        yk_set_error("yk_stream_create: bad argument");
        return YK_ERR_ARG;
    }
    hipStream_t s = nullptr;
    YK_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, priority));
    *stream_out = (void *)s;
    return YK_OK;
}
extern "C" int yk_stream_destroy(void *stream) {
    if (!stream) return YK_OK;
    YK_HIP(hipStreamSynchronize((hipStream_t)stream));
    yk_scratch_release_stream(stream);
    YK_HIP(hipStreamDestroy((hipStream_t)stream));
    return YK_OK;
}
extern "C" int yk_stream_query_priority(void *stream, int *priority_out) {
    if (!priority_out) {
        yk_set_error("yk_stream_query_priority: bad argument");
        return YK_ERR_ARG;
    }
    YK_HIP(hipStreamGetPriority((hipStream_t)stream, priority_out));
    return YK_OK;
}
