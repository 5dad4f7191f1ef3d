// yk_region.hip — C-mode region layer on gfx950 + the region_layer.h drop-in ABI.
//
// Replaces yolo3_frame_test_public/region_layer.c:
//   forward_region_layer :121-137 + get_region_boxes :177-214 + correct_region_boxes :139-164
//        -> region_decode_kernel  (one thread per box; same fp32 operation order per box,
//           the two double promotions of :158-159 kept)
//   do_nms_sort :256-283  -> region_nms_kernel (one wavefront per (image, class), yk_nms.h)
//   region_layer_draw_boxes :385-404 -> host loop over the mirrored buffers
//
// Compiled with -ffp-contract=off: every a*b+c stays two roundings, as in the reference's
// x86-64 build, so thresholds and IoU comparisons see the same fp32 values.
#include <iterator>
#include <map>
#include <mutex>
#include <utility>
#include <stdarg.h>
#include <stdlib.h>

#include "yk_common.h"
#include "yk_nms.h"

// ------------------------------------------------------------------ error / scratch
static thread_local char g_err[512] = "";
void yk_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char *yk_last_error(void) { return g_err; }
extern "C" int yk_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

struct scratch_key {
    int dev;
    void *stream;
    int slot;
    bool operator<(const scratch_key &o) const {
        if (dev != o.dev) return dev < o.dev;
        if (stream != o.stream) return stream < o.stream;
        return slot < o.slot;
    }
};
struct scratch_buf {
    void *p;
    size_t bytes;
};
static std::mutex g_scratch_mu;
static std::map<scratch_key, scratch_buf> g_scratch;
// how often a scratch buffer of (device, stream) has MOVED (grown past its slack: freed and allocated again).  A captured graph holds
// the pointers of its capture; a holder of such graphs compares generations before a replay and drops the graphs when they differ.
static std::map<std::pair<int, void *>, unsigned long long> g_scratch_gen;
void *yk_scratch(int device, void *stream, int slot, size_t bytes) {
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    scratch_buf &b = g_scratch[scratch_key{device, stream, slot}];
    if (b.bytes < bytes) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing((hipStream_t)stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) {
            // an allocation cannot be captured, and a graph replays the pointers it was captured with
            yk_set_error("yk_scratch: stream is capturing and its scratch must grow (%zu > %zu bytes): run the step once eagerly on this stream, at its "
                         "largest batch, before yk_graph_begin", bytes, b.bytes);
            return nullptr;
        }
        if (b.p) {
            (void)hipStreamSynchronize((hipStream_t)stream);
            (void)hipFree(b.p);
            ++g_scratch_gen[std::make_pair(device, stream)];
        }
        size_t want = bytes + bytes / 2;
        if (hipMalloc(&b.p, want) != hipSuccess) {
            b.p = nullptr;
            b.bytes = 0;
            yk_set_error("yk_scratch: hipMalloc(%zu) failed", want);
            return nullptr;
        }
        b.bytes = want;
    }
    return b.p;
}

extern "C" unsigned long long yk_scratch_generation(void *stream) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    auto it = g_scratch_gen.find(std::make_pair(dev, stream));
    return it == g_scratch_gen.end() ? 0ull : it->second;
}
// frees the scratch buffers keyed by a stream that is about to be destroyed (its handle value may be handed out again)
void yk_scratch_release_stream(void *stream) {
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    for (auto it = g_scratch.begin(); it != g_scratch.end();) {
        if (it->first.stream == stream) {
            if (it->second.p) (void)hipFree(it->second.p);
            it = g_scratch.erase(it);
        } else {
            ++it;
        }
    }
    for (auto it = g_scratch_gen.begin(); it != g_scratch_gen.end();) it = it->first.second == stream ? g_scratch_gen.erase(it) : std::next(it);
}

// ------------------------------------------------------------------ kernels
struct region_args {
    int W, H, A, C;
    int net_w, net_h, image_w, image_h;
    float threshold, nms_value;
    float anchor[2 * YK_MAX_ANCHORS];
    long long sb, sn, se, sy, sx;
    // letterbox constants precomputed on the host with the reference's exact expression types
    double off_x, off_y;      // (net_w - new_w) / 2. / net_w           (double)
    float ratio_x, ratio_y;   // (float)new_w / net_w                   (float)
    float scale_w, scale_h;   // (float)net_w / new_w                   (float)
};

// every exp of the C path goes through the glibc-identical evaluation (yk_common.h): the reference calls libm on the host
__device__ __forceinline__ float region_logistic(float v) { return 1.f / (1.f + yk_expf_glibc(-v)); }

// one thread per box; box index bi = n*H*W + loc  (region_layer.c:190)
__global__ void __launch_bounds__(256) region_decode_kernel(region_args a, const float *__restrict__ in, int batch,
                                                            float *__restrict__ out, float *__restrict__ boxes,
                                                            float *__restrict__ probs) {
    const int hw = a.W * a.H, nb = a.A * hw, E = 5 + a.C;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= batch * nb) return;
    const int b = t / nb, bi = t - b * nb;
    const int n = bi / hw, loc = bi - n * hw;
    const int row = loc / a.W, col = loc - row * a.W;
    const float *p = in + b * a.sb + n * a.sn + row * a.sy + col * a.sx;
    const float tx = p[0], ty = p[a.se], tw = p[2 * a.se], th = p[3 * a.se], to = p[4 * a.se];
    const float sx = region_logistic(tx), sy = region_logistic(ty), obj = region_logistic(to);
    float *o = out ? out + ((size_t)b * a.A + n) * E * hw + loc : nullptr;   // CHW like rl->output
    if (o) {
        o[0] = sx;
        o[hw] = sy;
        o[2 * hw] = tw;
        o[3 * hw] = th;
        o[4 * hw] = obj;
    }
    // softmax over classes, same sequential order as region_layer.c:96-108
    const float *cl = p + 5 * a.se;
    float top = cl[0];
    for (int j = 0; j < a.C; ++j) {
        float v = cl[j * a.se];
        if (v > top) top = v;
    }
    float total = 0.f;
    for (int j = 0; j < a.C; ++j) total += yk_expf_glibc(cl[j * a.se] - top);
    float *pr = probs + ((size_t)b * nb + bi) * (a.C + 1);
    float best = 0.f;
    for (int j = 0; j < a.C; ++j) {
        float e = yk_expf_glibc(cl[j * a.se] - top) / total;
        if (o) o[(5 + j) * hw] = e;
        float pj = obj * e;
        pr[j] = (pj > a.threshold) ? pj : 0.f;
        if (pj > best) best = pj;
    }
    pr[a.C] = best;
    float bx = (col + sx) / a.W;
    float by = (row + sy) / a.H;
    float bw = yk_expf_glibc(tw) * a.anchor[2 * n];
    float bh = yk_expf_glibc(th) * a.anchor[2 * n + 1];
    bx = (float)(((double)bx - a.off_x) / (double)a.ratio_x);
    by = (float)(((double)by - a.off_y) / (double)a.ratio_y);
    bw *= a.scale_w;
    bh *= a.scale_h;
    float4 *bo = reinterpret_cast<float4 *>(boxes) + (size_t)b * nb + bi;
    *bo = make_float4(bx, by, bw, bh);
}

// centre-format IoU, region_layer.c:228-254 (operation order kept)
__device__ __forceinline__ float region_span(float c1, float s1, float c2, float s2) {
    float lo1 = c1 - s1 / 2, lo2 = c2 - s2 / 2;
    float lo = lo1 > lo2 ? lo1 : lo2;
    float hi1 = c1 + s1 / 2, hi2 = c2 + s2 / 2;
    float hi = hi1 < hi2 ? hi1 : hi2;
    return hi - lo;
}
__device__ __forceinline__ float region_iou(const float4 &a, const float4 &b) {
    float w = region_span(a.x, a.z, b.x, b.z);
    float h = region_span(a.y, a.w, b.y, b.w);
    float inter = (w < 0 || h < 0) ? 0.f : w * h;
    float uni = a.z * a.w + b.z * b.w - inter;
    return inter / uni;
}

// grid = (classes, batch), block = one wavefront
__global__ void __launch_bounds__(64) region_nms_kernel(int nb, int C, float nms_value, const float *__restrict__ boxes,
                                                        float *__restrict__ probs) {
    __shared__ yk_cand_lds L;
    const int k = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    const float4 *bx = reinterpret_cast<const float4 *>(boxes) + (size_t)b * nb;
    float *pr = probs + (size_t)b * nb * (C + 1) + k;
    const int stride = C + 1;
    // gather non-zero probabilities of class k, ascending box index
    int n = 0;
    for (int base = 0; base < nb; base += 64) {
        int i = base + lane;
        float p = (i < nb) ? pr[(size_t)i * stride] : 0.f;
        bool f = (i < nb) && (p != 0.f);
        unsigned long long m = __ballot(f);
        int pos = n + __popcll(m & ((1ull << lane) - 1ull));
        if (f && pos < YK_NMS_MAXC) {
            L.s[pos] = p;
            L.idx[pos] = i;
            L.box[pos] = bx[i];
        }
        n += __popcll(m);
    }
    __syncthreads();
    if (n <= YK_NMS_MAXC) {
        if (n < 2) return;
        yk_wave_greedy_nms(
            n, L.s, L.idx, L.box, nms_value, 0x7fffffff, [](const float4 &a, const float4 &c) { return region_iou(a, c); },
            [](int, int) {}, [&](int pos) { pr[(size_t)L.idx[pos] * stride] = 0.f; });
        return;
    }
    // overflow path (more candidates than LDS slots): same algorithm in place on global memory;
    // a kept box is parked as -p until the end.
    for (;;) {
        float best = -INFINITY;
        int bidx = 0x7fffffff, bpos = -1;
        for (int i = lane; i < nb; i += 64) {
            float v = pr[(size_t)i * stride];
            if (v > 0.f && (bpos < 0 || v > best)) {   // ascending i per lane => first max has the lowest index
                best = v;
                bidx = i;
                bpos = i;
            }
        }
        yk_wave_argmax(best, bidx, bpos);
        if (bpos < 0) break;
        const float4 wb = bx[bpos];
        __syncthreads();
        if (lane == 0) pr[(size_t)bpos * stride] = -best;
        __threadfence_block();
        __syncthreads();
        for (int i = lane; i < nb; i += 64) {
            float v = pr[(size_t)i * stride];
            if (v > 0.f && region_iou(wb, bx[i]) > nms_value) pr[(size_t)i * stride] = 0.f;
        }
        __threadfence_block();
        __syncthreads();
    }
    for (int i = lane; i < nb; i += 64) {
        float v = pr[(size_t)i * stride];
        if (v < 0.f) pr[(size_t)i * stride] = -v;
    }
}

// ------------------------------------------------------------------ host side
static int fill_args(region_args &a, const yk_region_cfg_t *c) {
    if (!c || c->layer_w <= 0 || c->layer_h <= 0 || c->anchor_num <= 0 || c->anchor_num > YK_MAX_ANCHORS ||
        c->classes <= 0 || c->net_w <= 0 || c->net_h <= 0 || c->image_w <= 0 || c->image_h <= 0) {
        yk_set_error("yk_region: bad configuration");
        return YK_ERR_ARG;
    }
    a.W = c->layer_w;
    a.H = c->layer_h;
    a.A = c->anchor_num;
    a.C = c->classes;
    a.net_w = c->net_w;
    a.net_h = c->net_h;
    a.image_w = c->image_w;
    a.image_h = c->image_h;
    a.threshold = c->threshold;
    a.nms_value = c->nms_value;
    memcpy(a.anchor, c->anchor, sizeof(float) * 2 * c->anchor_num);
    a.sb = c->stride_b;
    a.sn = c->stride_n;
    a.se = c->stride_e;
    a.sy = c->stride_y;
    a.sx = c->stride_x;
    // correct_region_boxes, region_layer.c:139-161, with the reference's integer/unsigned types
    uint32_t net_width = (uint32_t)c->net_w, net_height = (uint32_t)c->net_h;
    uint32_t image_width = (uint32_t)c->image_w, image_height = (uint32_t)c->image_h;
    int new_w, new_h;
    if (((float)net_width / image_width) < ((float)net_height / image_height)) {
        new_w = net_width;
        new_h = (image_height * net_width) / image_width;
    } else {
        new_h = net_height;
        new_w = (image_width * net_height) / image_height;
    }
    a.off_x = (net_width - new_w) / 2. / net_width;
    a.off_y = (net_height - new_h) / 2. / net_height;
    a.ratio_x = (float)new_w / net_width;
    a.ratio_y = (float)new_h / net_height;
    a.scale_w = (float)net_width / new_w;
    a.scale_h = (float)net_height / new_h;
    return YK_OK;
}

extern "C" int yk_region_batched(const yk_region_cfg_t *cfg, const float *d_input, int batch, float *d_output,
                                 float *d_boxes, float *d_probs, void *stream) {
    region_args a;
    int rc = fill_args(a, cfg);
    if (rc) return rc;
    if (!d_input || !d_boxes || !d_probs || batch <= 0) {
        yk_set_error("yk_region_batched: null pointer / batch");
        return YK_ERR_ARG;
    }
    const int nb = a.A * a.W * a.H;
    const int total = batch * nb;
    hipLaunchKernelGGL(region_decode_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, d_input,
                       batch, d_output, d_boxes, d_probs);
    hipLaunchKernelGGL(region_nms_kernel, dim3(a.C, batch), dim3(64), 0, (hipStream_t)stream, nb, a.C, a.nms_value,
                       d_boxes, d_probs);
    YK_HIP(hipGetLastError());
    return YK_OK;
}

// ---- drop-in region_layer.h ABI -------------------------------------------------
struct rl_dev {
    float *d_in = nullptr, *d_out = nullptr, *d_boxes = nullptr, *d_probs = nullptr;
    hipStream_t stream = nullptr;
};
static std::mutex g_rl_mu;
static std::map<region_layer_t *, rl_dev> g_rl;

static void rl_dev_free(rl_dev &d) {
    if (d.d_in) (void)hipFree(d.d_in);
    if (d.d_out) (void)hipFree(d.d_out);
    if (d.d_boxes) (void)hipFree(d.d_boxes);
    if (d.d_probs) (void)hipFree(d.d_probs);
    if (d.stream) (void)hipStreamDestroy(d.stream);
    d = rl_dev();
}

extern "C" int region_layer_init(region_layer_t *rl, int width, int height, int channels, int origin_width,
                                 int origin_height) {
    int flag = 0;
    rl->coords = 4;
    rl->image_width = 320;   // hard-coded in the reference regardless of arguments, region_layer.c:24-25
    rl->image_height = 224;
    rl->classes = channels / rl->anchor_number - 5;
    rl->net_width = origin_width;
    rl->net_height = origin_height;
    rl->layer_width = width;
    rl->layer_height = height;
    rl->boxes_number = rl->layer_width * rl->layer_height * rl->anchor_number;
    rl->output_number = rl->boxes_number * (rl->classes + rl->coords + 1);
    rl->output = nullptr;
    rl->boxes = nullptr;
    rl->probs_buf = nullptr;
    rl->probs = nullptr;
    rl_dev d;
    rl->output = (float *)malloc(rl->output_number * sizeof(float));
    if (!rl->output) { flag = -1; goto fail; }
    rl->boxes = malloc(rl->boxes_number * 4 * sizeof(float));
    if (!rl->boxes) { flag = -2; goto fail; }
    rl->probs_buf = (float *)malloc((size_t)rl->boxes_number * (rl->classes + 1) * sizeof(float));
    if (!rl->probs_buf) { flag = -3; goto fail; }
    rl->probs = (float **)malloc(rl->boxes_number * sizeof(float *));
    if (!rl->probs) { flag = -4; goto fail; }
    for (uint32_t i = 0; i < rl->boxes_number; i++) rl->probs[i] = &(rl->probs_buf[i * (rl->classes + 1)]);
    // device mirror: no GPU => loud failure, never a CPU fallback
    if (rl->anchor_number > YK_MAX_ANCHORS || (int)rl->classes <= 0 ||
        hipMalloc(&d.d_in, rl->output_number * sizeof(float)) != hipSuccess ||
        hipMalloc(&d.d_out, rl->output_number * sizeof(float)) != hipSuccess ||
        hipMalloc(&d.d_boxes, rl->boxes_number * 4 * sizeof(float)) != hipSuccess ||
        hipMalloc(&d.d_probs, (size_t)rl->boxes_number * (rl->classes + 1) * sizeof(float)) != hipSuccess ||
        hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking) != hipSuccess) {
        yk_set_error("region_layer_init: no usable HIP device / device allocation failed");
        rl_dev_free(d);
        flag = -5;
        goto fail;
    }
    {
        std::lock_guard<std::mutex> lk(g_rl_mu);
        g_rl[rl] = d;
    }
    return 0;
fail:
    free(rl->output);
    free(rl->boxes);
    free(rl->probs_buf);
    free(rl->probs);
    rl->output = nullptr;
    rl->boxes = nullptr;
    rl->probs_buf = nullptr;
    rl->probs = nullptr;
    return flag;
}

extern "C" void region_layer_deinit(region_layer_t *rl) {
    {
        std::lock_guard<std::mutex> lk(g_rl_mu);
        auto it = g_rl.find(rl);
        if (it != g_rl.end()) {
            rl_dev_free(it->second);
            g_rl.erase(it);
        }
    }
    free(rl->output);
    free(rl->boxes);
    free(rl->probs_buf);
    free(rl->probs);
}

extern "C" void region_layer_run(region_layer_t *rl, obj_info_t *obj_info) {
    (void)obj_info;   // region_layer_output is commented out in the reference (region_layer.c:382)
    rl_dev d;
    {
        std::lock_guard<std::mutex> lk(g_rl_mu);
        auto it = g_rl.find(rl);
        if (it == g_rl.end()) {
            yk_set_error("region_layer_run: region_layer_init was not called / failed for this layer");
            fprintf(stderr, "libyolo_hip: %s\n", yk_last_error());
            abort();   // the reference would crash on the NULL buffers too; never compute on the CPU instead
        }
        d = it->second;
    }
    yk_region_cfg_t c;
    memset(&c, 0, sizeof(c));
    c.layer_w = rl->layer_width;
    c.layer_h = rl->layer_height;
    c.anchor_num = rl->anchor_number;
    c.classes = rl->classes;
    c.net_w = rl->net_width;
    c.net_h = rl->net_height;
    c.image_w = rl->image_width;
    c.image_h = rl->image_height;
    c.threshold = rl->threshold;
    c.nms_value = rl->nms_value;
    memcpy(c.anchor, rl->anchor, sizeof(float) * 2 * rl->anchor_number);
    const long long hw = (long long)rl->layer_width * rl->layer_height, E = 5 + rl->classes;
    c.stride_b = rl->anchor_number * E * hw;   // CHW [A*E][H][W], entry_index region_layer.c:84-89
    c.stride_n = E * hw;
    c.stride_e = hw;
    c.stride_y = rl->layer_width;
    c.stride_x = 1;
    const size_t out_bytes = rl->output_number * sizeof(float);
    const size_t box_bytes = (size_t)rl->boxes_number * 4 * sizeof(float);
    const size_t prob_bytes = (size_t)rl->boxes_number * (rl->classes + 1) * sizeof(float);
    bool ok = hipMemcpyAsync(d.d_in, rl->input, out_bytes, hipMemcpyHostToDevice, d.stream) == hipSuccess &&
              yk_region_batched(&c, d.d_in, 1, d.d_out, d.d_boxes, d.d_probs, d.stream) == YK_OK &&
              hipMemcpyAsync(rl->output, d.d_out, out_bytes, hipMemcpyDeviceToHost, d.stream) == hipSuccess &&
              hipMemcpyAsync(rl->boxes, d.d_boxes, box_bytes, hipMemcpyDeviceToHost, d.stream) == hipSuccess &&
              hipMemcpyAsync(rl->probs_buf, d.d_probs, prob_bytes, hipMemcpyDeviceToHost, d.stream) == hipSuccess &&
              hipStreamSynchronize(d.stream) == hipSuccess;
    if (!ok) {
        fprintf(stderr, "libyolo_hip: region_layer_run failed on the device: %s\n", yk_last_error());
        abort();
    }
}

// (uint32_t) of a float as the reference's x86-64 build computes it (cvttss2si to 64 bits, then truncation): NaN and values outside the
// int64 range give the "integer indefinite" 0x8000000000000000, i.e. 0 after truncation.  Spelled out: the plain cast is undefined
// behaviour in C for those inputs (found by the UBSan build on a box of garbage coordinates, tools/run_asan.sh).
static inline uint32_t to_u32(float v) {
    if (!(v >= -9223372036854775808.0f && v < 9223372036854775808.0f)) return 0u;
    return (uint32_t)(int64_t)v;
}

extern "C" void region_layer_draw_boxes(region_layer_t *rl, callback_draw_box callback) {
    const uint32_t image_width = rl->image_width, image_height = rl->image_height;
    const float threshold = rl->threshold;
    const float *boxes = (const float *)rl->boxes;
    for (uint32_t i = 0; i < rl->boxes_number; ++i) {
        const float *pr = rl->probs[i];
        int cls = 0;
        float top = pr[0];
        for (uint32_t j = 1; j < rl->classes; ++j)
            if (pr[j] > top) {
                top = pr[j];
                cls = (int)j;
            }
        if (top > threshold) {
            const float *b = boxes + 4 * (size_t)i;
            uint32_t x1 = to_u32(b[0] * image_width - (b[2] * image_width / 2));
            uint32_t y1 = to_u32(b[1] * image_height - (b[3] * image_height / 2));
            uint32_t x2 = to_u32(b[0] * image_width + (b[2] * image_width / 2));
            uint32_t y2 = to_u32(b[1] * image_height + (b[3] * image_height / 2));
            callback(x1, y1, x2, y2, (uint32_t)cls, top);
        }
    }
}
