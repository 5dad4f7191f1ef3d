// yk_conv.h — launch wrappers of the gfx950 conv-stack kernels (yk_conv.hip).
//
// Storage rule (mirrored by oracle/yolo_net_ref.c emulate_f16): activations are fp16 NHWC with the
// channel pitch padded to a multiple of 8 (pad lanes hold zeros); weights fp16, reduction axis
// contiguous and zero-padded the same way; accumulation, BatchNorm scale/bias and activation in
// fp32; network outputs fp32 with exact pitch.
#pragma once
#include "yk_common.h"

typedef _Float16 yk_half;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

static inline int yk_pad8(int c) { return (c + 7) & ~7; }

// unsigned division by a launch-invariant divisor without the ~40-instruction software divide:
// q = (umulhi(n, mul) + n) >> shift, exact for n < 2^31 (Granlund-Montgomery round-up method)
struct yk_fastdiv {
    uint32_t mul, shift;
};
static inline yk_fastdiv yk_make_fastdiv(uint32_t d) {
    yk_fastdiv f;
    uint32_t s = 0;
    while ((1ull << s) < d) ++s;
    f.shift = s;
    f.mul = (uint32_t)((((1ull << s) - d) << 32) / d + 1);
    return f;
}
// activation as y = min(max(v, v*slope), cap): relu slope 0, leaky slope alpha, none slope 1; relu6 cap 6
static inline void yk_act_params(int act, float alpha, float *slope, float *cap) {
    *slope = (act == YK_ACT_NONE) ? 1.f : (act == YK_ACT_LEAKY ? alpha : 0.f);
    *cap = (act == YK_ACT_RELU6) ? 6.f : __builtin_huge_valf();
}

// ---- implicit-GEMM conv (1x1 / 3x3, stride 1/2, optional [up2(src0), src1] concat input) -------
struct igemm_args {
    const yk_half *in0, *in1;   // in1 may be null
    int c0p, c1p;               // channel pitch of the sources (c1p = 0 without concat)
    int up0;                    // src0 is read through a nearest 2x upsample
    int Hi, Wi;                 // logical input size (after upsample / concat)
    int Ho, Wo;
    int ks, stride, pad_t, pad_l;
    int M, N, K;                // M = B*Ho*Wo, N = Cout, K = ks*ks*(c0p+c1p)
    const yk_half *w;           // [N][K]
    // the same weights in MFMA-fragment order [K/64 steps][ceil(N/16) blocks][2 half-steps][64 lanes][8] (null unless the plan was built
    // for the register-fragment ring kernel, yk_igemm_br.h): a wave's load of one block and half-step is 1 KB of contiguous memory
    const yk_half *wfrag;
    uint32_t wfrag_bytes;
    int nb16;
    const float *scale, *bias;  // [N]
    int act;
    float alpha;
    float slope, cap;           // yk_act_params(act, alpha)
    yk_fastdiv fd_hw, fd_wo;    // division by Ho*Wo and Wo
    yk_fastdiv fd_g;            // division by c0p/8 (fused depthwise producer)
    yk_fastdiv fd_vpr;          // division by outp/8 (flat tile store)
    yk_fastdiv fd_ctp;          // division by c0p+c1p (k -> tap)
    uint32_t in0_bytes;         // byte size of in0 (buffer-load bounds)
    uint32_t in1_bytes, w_bytes;
    int split_k;                // >1: partial sums go to `slab` [split][M][ldn] fp32, finished by yk_launch_splitk_reduce
    float *slab;
    int ldn;
    const yk_half *res;         // residual added after the activation (pitch resp), or null
    int resp;
    void *out;                  // fp16 pitch outp, or fp32 pitch outp when out_f32
    int outp;
    // optional fused depthwise producer (dw3x3 + BN + act feeding the 1x1 GEMM as its A operand)
    const yk_half *dw_w;        // [9][c0p] fp16 or null
    const float *dw_scale, *dw_bias;
    int dw_act, dw_stride, dw_pad_t, dw_pad_l, dw_Hi, dw_Wi;
    float dw_slope, dw_cap;
    int lda_pad;                // LDS pitch padding (halfs) of the fused kernels' depthwise tile (yk_fused_pad())
    long long *dbg;             // dev instrumentation: per-workgroup phase timestamps (null in production)
    // patch geometry of the LDS-DMA staged fused kernel (yk_fused_dma.h), fixed at plan creation for max_batch so that an image's
    // arithmetic never depends on how many images share the launch
    int fp_npw, fp_ks, fp_mt, fp_tr, fp_tc, fp_ns;
    unsigned fp_lds;
};
void yk_fdma_fill(igemm_args &a);   // computes the fp_* fields (a.M = max_batch * Ho * Wo)
enum { IGEMM_128x64 = 0, IGEMM_128x48, IGEMM_128x96, IGEMM_128x192, IGEMM_64x64, IGEMM_128x128, IGEMM_F32_64x80,
       IGEMM_F32_128x64, IGEMM_128x64K64, IGEMM_64x128, IGEMM_64x192, IGEMM_256x128 /* developer build only */, IGEMM_256x128W4, IGEMM_128x256, IGEMM_128x128R, IGEMM_256x256, IGEMM_LC_256x128, IGEMM_LC_128x128, IGEMM_LC_128x256, IGEMM_NUM };
int yk_launch_igemm(int cfg, const igemm_args &a, hipStream_t st);
int yk_fused_pad();
int yk_igemm_pick(const igemm_args &a, bool out_f32);
// split-K policy for a picked config (1 = no split) and the finishing pass
int yk_igemm_split(int cfg, const igemm_args &a);
int yk_launch_splitk_reduce(const igemm_args &a, bool out_f32, hipStream_t st);
// split-K finishing pass + the 1x1 fp32-output conv that consumes it, one launch
bool yk_reduce_pw_ok(const igemm_args &a, const igemm_args &b, bool b_out_f32);
int yk_launch_reduce_pw(const igemm_args &a, const igemm_args &b, hipStream_t st);
const char *yk_igemm_name(int cfg);

// fused DepthwiseConv2D(3x3)+BN+act -> Conv2D(1x1)+BN+act: the depthwise tile is produced straight
// into LDS (never written to HBM) and consumed as the MFMA pixel operand; weights stream from L2.
enum { FUSED_128x48 = 0, FUSED_128x96, FUSED_64x192, FUSED_32x192,
       WIDE_4x3_T4, WIDE_2x6_T4, WIDE_2x6_T2, WIDE_1x12_T4, WIDE_1x12_T2, LR_T1, LR_T2, WIDE_1x12_T5, WIDE_1x12_T9, FUSED_DMA, FUSED_NUM };
bool yk_igemm_fused_ok(int c0p, int cout);
int yk_igemm_fused_pick(const igemm_args &a);
const char *yk_igemm_fused_name(int cfg);
int yk_launch_igemm_fused(int cfg, const igemm_args &a, hipStream_t st);

// ---- first conv: 3 input channels, u8 (with fused img/max(img)) or fp32 input -------------------
struct first_args {
    const void *in;             // u8 or f32 [B][Hi][Wi][3]
    const unsigned *img_max;    // [B] per-image max (u8 path) or null
    int in_f32;
    int B, Hi, Wi, Ho, Wo, stride, pad_t, pad_l, Cout, outp;
    const float *w;             // [27][Cout] fp32 holding fp16-rounded values, tap-major
    const yk_half *wm;          // [32][32] fp16, MFMA operand order of stem_mfma_kernel (k = ky*8+j | 24+ky), or null
    const float *scale, *bias;
    int act;
    float alpha;
    float slope, cap;
    yk_half *out;
};
int yk_launch_first(const first_args &a, hipStream_t st);
int yk_launch_u8_max(const uint8_t *frames, size_t per_image, int batch, unsigned *img_max, hipStream_t st, uint32_t *zero = nullptr,
                     size_t zero_words = 0);

// ---- depthwise 3x3 ------------------------------------------------------------------------------
struct dw_args {
    const yk_half *in;
    int B, Hi, Wi, Ho, Wo, Cp, stride, pad_t, pad_l;
    const yk_half *w;           // [9][Cp]
    const float *scale, *bias;  // [Cp] (pad lanes: scale 0, bias 0)
    int act;
    float alpha;
    float slope, cap;
    yk_half *out;
};
int yk_launch_dw(const dw_args &a, hipStream_t st);

// ---- 2x2 max pool ('same') ----------------------------------------------------------------------
struct pool_args {
    const yk_half *in;
    int B, Hi, Wi, Ho, Wo, Cp, stride;
    yk_half *out;
};
int yk_launch_pool(const pool_args &a, hipStream_t st);

// ---- residual add fallback (only when it cannot be fused into a conv epilogue) ------------------
int yk_launch_add(const yk_half *a, const yk_half *b, yk_half *out, size_t n8, hipStream_t st);

// ---- "f16x2" precision mode (yk_exact.hip): fp32 activations, compensated fp16 MFMA operands ----------------------
struct yk_xplan;
int yk_xplan_create(yk_xplan **out, const int32_t *ops, int n_ops, const int32_t *tensors, int n_tensors, const float *blob,
                    size_t blob_len, const int32_t *outputs, int n_outputs, int max_batch, int latency_schedule);
void yk_xplan_destroy(yk_xplan *p);
int yk_xplan_run(yk_xplan *p, const void *d_in, int in_f32, int batch, hipStream_t st, hipEvent_t *ev);
int yk_xplan_output(yk_xplan *p, int idx, float **d_ptr, size_t *bytes, int *h, int *w, int *c);
int yk_xplan_read_tensor(yk_xplan *p, int tid, int batch, float *h_dst, size_t dst_elems);
int yk_xplan_launch_count(const yk_xplan *p);
int yk_xplan_check(yk_xplan *p);
unsigned yk_xplan_peek_error(yk_xplan *p, int clear);
void yk_xplan_debug_set_error(yk_xplan *p, unsigned value);
int yk_xplan_phase_stamps(yk_xplan *p, int li, const void *d_in, int batch, hipStream_t st, long long *h_out, int max_wg);
int yk_xplan_launch_info(const yk_xplan *p, int i, const char **name, double *flops, double *bytes);
