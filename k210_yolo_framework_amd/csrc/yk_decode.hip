// yk_decode.hip — Python-mode decode + per-class NMS on gfx950 (batched).
//
// Replaces, for a whole batch and without host round trips,
//   keras_inference.py:94-111   sigmoid(cls)*sigmoid(conf), reshape(-1,..) in (layer,h,w,anchor) order
//   tools/utils.py:545-546      tf_xywh_to_all
//   keras_inference.py:51-72    correct_box
//   keras_inference.py:113-135  scores >= obj_thresh ; per class tf.image.non_max_suppression(…, 30, iou)
// which the reference runs as ~100 eager TF op dispatches per image.
//
// Three launches: decode (thread per box, coalesced class-major score planes) -> NMS (one
// wavefront per (image, class), yk_nms.h) -> compaction to the reference's class-major row order.
// Compiled with -ffp-contract=off: one rounding per TF op, like the eager fp32 graph.
#include "yk_common.h"
#include "yk_nms.h"

struct decode_args {
    int L, A, C, E;
    int in_h, in_w;
    int out_h[YK_MAX_LAYERS], out_w[YK_MAX_LAYERS];
    int off[YK_MAX_LAYERS + 1];   // first global box index of each layer
    float anchors[YK_MAX_LAYERS][YK_MAX_ANCHORS][2];
    const float *pred[YK_MAX_LAYERS];
};

__device__ __forceinline__ float tf_sigmoid(float v) { return 1.f / (1.f + expf(-v)); }

__global__ void __launch_bounds__(256) decode_py_kernel(decode_args a, int batch, const float *__restrict__ image_hw,
                                                        float4 *__restrict__ boxes, float *__restrict__ scores_t) {
    const int ntot = a.off[a.L];
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= batch * ntot) return;
    const int b = t / ntot, g = t - b * ntot;
    int l = 0;
    while (l + 1 < a.L && g >= a.off[l + 1]) ++l;
    const int r = g - a.off[l];
    const int cell = r / a.A, an = r - cell * a.A;
    const int h = a.out_h[l], w = a.out_w[l];
    const int row = cell / w, col = cell - row * w;
    const float *p = a.pred[l] + ((size_t)(b * h + row) * w + col) * a.A * a.E + (size_t)an * a.E;

    const float conf = tf_sigmoid(p[4]);
    for (int c = 0; c < a.C; ++c) scores_t[((size_t)b * a.C + c) * ntot + g] = tf_sigmoid(p[5 + c]) * conf;

    // tf_xywh_to_all (tools/utils.py:545-546)
    const float x = (tf_sigmoid(p[0]) + (float)col) / (float)w;
    const float y = (tf_sigmoid(p[1]) + (float)row) / (float)h;
    const float bw = expf(p[2]) * a.anchors[l][an][0];
    const float bh = expf(p[3]) * a.anchors[l][an][1];

    // correct_box (keras_inference.py:51-72), (y,x) order
    const float in_h = (float)a.in_h, in_w = (float)a.in_w;
    const float im_h = image_hw ? image_hw[2 * b] : in_h, im_w = image_hw ? image_hw[2 * b + 1] : in_w;
    const float rh = in_h / im_h, rw = in_w / im_w;
    const float m = rh < rw ? rh : rw;
    const float new_h = rintf(im_h * m), new_w = rintf(im_w * m);       // tf.round = half-to-even
    const float off_y = (in_h - new_h) / 2.f / in_h, off_x = (in_w - new_w) / 2.f / in_w;
    const float sc_y = in_h / new_h, sc_x = in_w / new_w;
    const float cy = (y - off_y) * sc_y, cx = (x - off_x) * sc_x;
    const float hh = bh * sc_y, ww = bw * sc_x;
    const float hy = hh / 2.f, hx = ww / 2.f;
    boxes[(size_t)b * ntot + g] = make_float4((cy - hy) * im_h, (cx - hx) * im_w, (cy + hy) * im_h, (cx + hx) * im_w);
}

// TF 1.14 non_max_suppression_op.cc IOU on (y1,x1,y2,x2)
__device__ __forceinline__ float tf_iou(const float4 &i, const float4 &j) {
    const float ymin_i = fminf(i.x, i.z), xmin_i = fminf(i.y, i.w), ymax_i = fmaxf(i.x, i.z), xmax_i = fmaxf(i.y, i.w);
    const float ymin_j = fminf(j.x, j.z), xmin_j = fminf(j.y, j.w), ymax_j = fmaxf(j.x, j.z), xmax_j = fmaxf(j.y, j.w);
    const float area_i = (ymax_i - ymin_i) * (xmax_i - xmin_i);
    const float area_j = (ymax_j - ymin_j) * (xmax_j - xmin_j);
    if (area_i <= 0.f || area_j <= 0.f) return 0.f;
    const float iy0 = fmaxf(ymin_i, ymin_j), ix0 = fmaxf(xmin_i, xmin_j);
    const float iy1 = fminf(ymax_i, ymax_j), ix1 = fminf(xmax_i, xmax_j);
    const float inter = fmaxf(iy1 - iy0, 0.f) * fmaxf(ix1 - ix0, 0.f);
    return inter / (area_i + area_j - inter);
}

// grid (C, batch), one wavefront.  sel_g/sel_s: [batch][C][max_out]; cnt: [batch][C]
// MAXC: LDS capacity in candidates.  1 088 (26 KB) covers the 224x320 heads (1 050 boxes) and leaves the CU's LDS to the conv kernels
// of other batches in flight; 2 048 (48 KB) otherwise.
template <int MAXC>
struct nms_lds {
    float s[MAXC];
    int idx[MAXC];
    float4 box[MAXC];
};
template <int MAXC>
__global__ void __launch_bounds__(64) nms_py_kernel(int ntot, int C, float obj_thresh, float iou_thresh, int max_out,
                                                    const float4 *__restrict__ boxes, float *__restrict__ scores_t,
                                                    int *__restrict__ sel_g, float *__restrict__ sel_s,
                                                    int *__restrict__ cnt) {
    __shared__ nms_lds<MAXC> L;
    const int c = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    float *sc = scores_t + ((size_t)b * C + c) * ntot;
    const float4 *bx = boxes + (size_t)b * ntot;
    int *og = sel_g + ((size_t)b * C + c) * max_out;
    float *os = sel_s + ((size_t)b * C + c) * max_out;
    int n = 0;
    float cmin = INFINITY, cmax = -INFINITY;                          // candidate score range: all equal = the order is the box index = the scan order
    for (int base0 = 0; base0 < ntot; base0 += 64 * 8) {
        float sv[8];                                                  // 8 independent loads in flight
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base0 + u * 64 + lane;
            sv[u] = (i < ntot) ? sc[i] : -INFINITY;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base0 + u * 64 + lane;
            const bool f = (i < ntot) && (sv[u] >= obj_thresh);      // keras_inference.py:116 (>=)
            const unsigned long long m = __ballot(f);
            const int pos = n + __popcll(m & ((1ull << lane) - 1ull));
            if (f && pos < MAXC) {
                L.s[pos] = sv[u];
                L.idx[pos] = i;
            }
            if (f) {
                cmin = fminf(cmin, sv[u]);
                cmax = fmaxf(cmax, sv[u]);
            }
            n += __popcll(m);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        cmin = fminf(cmin, __shfl_xor(cmin, o, 64));
        cmax = fmaxf(cmax, __shfl_xor(cmax, o, 64));
    }
    const bool all_tied = n > 0 && cmin == cmax;
    __syncthreads();
    // boxes of the candidates, fetched in bulk AFTER the scan: a load inside the ballot loop above would put one global
    // round trip on the critical path per unrolled step (measured: 24 serialised latencies, ~40 us of a 53 us kernel)
    {
        const int nc = min(n, MAXC);
        for (int c0 = 0; c0 < nc; c0 += 256) {
            float4 t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = c0 + u * 64 + lane;
                t[u] = bx[c < nc ? L.idx[c] : 0];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = c0 + u * 64 + lane;
                if (c < nc) L.box[c] = t[u];
            }
        }
    }
    __syncthreads();
    int kept = 0;
    // ---- sorted sweep (round 6): the first nc candidates of the LDS arrays in descending (score, -index) order - a bitonic sort of one
    // 64-bit key per candidate, (score bits, ~index : 20, LDS position : 12), written over L.s | L.idx - then TF's own formulation as in the
    // fast path below: every candidate is tested against the boxes selected so far, which live one per lane in registers (max_out <= 64),
    // ACROSS calls (the chunks of the overflow path).  One IoU + one ballot per candidate instead of two LDS passes over all of them per
    // SELECTED box: 1 000 - 2 000 tied candidates per class (saturated logits) took ~0.5 ms per wave that way.
    float q_y0 = 0.f, q_x0 = 0.f, q_y1 = 0.f, q_x1 = 0.f, q_a = 0.f;
    const bool q_guard = iou_thresh > 0.f;
    const float q_hi = q_guard ? iou_thresh * (1.f + 2e-6f) : INFINITY, q_lo = q_guard ? iou_thresh * (1.f - 2e-6f) : -INFINITY;
    const bool sweep_ok = max_out <= 64 && ntot < (1 << 20);
    auto sorted_sweep = [&](int nc, bool presorted) {
        // presorted: the LDS arrays already ARE in key order (every score ties: order = box index = scan order): no sort, keys built in place
        unsigned long long *key = reinterpret_cast<unsigned long long *>(L.s);      // L.s and L.idx are adjacent: 8 bytes per candidate
        int P = 64;
        while (P < nc && !presorted) P <<= 1;                                       // (callers make sure P <= MAXC)
        if (presorted) P = nc;
        constexpr int PL = (MAXC + 63) / 64;
        uint32_t sv_[PL], iv_[PL];
#pragma unroll
        for (int k = 0; k < PL; ++k) {
            const int c = lane + 64 * k;
            sv_[k] = c < nc ? __float_as_uint(L.s[c]) : 0u;
            iv_[k] = c < nc ? (uint32_t)L.idx[c] : 0u;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PL; ++k) {
            const int c = lane + 64 * k;
            if (c < P) key[c] = c < nc ? (((unsigned long long)sv_[k] << 32) | ((unsigned long long)((~iv_[k]) & 0xfffffu) << 12) | (unsigned)c) : 0ull;
            if (c < nc) {                                                           // corners normalised once (min / max are idempotent)
                const float4 q = L.box[c];
                L.box[c] = make_float4(fminf(q.x, q.z), fminf(q.y, q.w), fmaxf(q.x, q.z), fmaxf(q.y, q.w));
            }
        }
        __syncthreads();
        for (int k = 2; k <= P && !presorted; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = lane; t < (P >> 1); t += 64) {
                    const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
                    const unsigned long long a = key[i], b = key[l];
                    const bool desc = (i & k) == 0;
                    if ((a < b) == desc) {
                        key[i] = b;
                        key[l] = a;
                    }
                }
                __syncthreads();
            }
        for (int base = 0; base < nc && kept < max_out; base += 64) {
            const int me = base + lane;
            const unsigned long long kk = me < nc ? key[me] : 0ull;
            const int pos = (int)(kk & 0xfffu);
            const float4 bb = L.box[pos];
            const float ar = (bb.z - bb.x) * (bb.w - bb.y);
            const int sbits = (int)(uint32_t)(kk >> 32);
            const int iv = (int)((~(uint32_t)(kk >> 12)) & 0xfffffu);
            const int cnt_ = min(64, nc - base);
            for (int t = 0; t < cnt_ && kept < max_out; ++t) {
                const float cy0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bb.x), t));
                const float cx0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bb.y), t));
                const float cy1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bb.z), t));
                const float cx1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bb.w), t));
                const float ca = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ar), t));
                const float inter = fmaxf(fminf(cy1, q_y1) - fmaxf(cy0, q_y0), 0.f) * fmaxf(fminf(cx1, q_x1) - fmaxf(cx0, q_x0), 0.f);
                const float uni = ca + q_a - inter;
                const bool valid = (lane < kept) && (ca > 0.f) && (q_a > 0.f);        // zero-area boxes never overlap (TF)
                unsigned long long hits = __ballot(valid && inter > q_hi * uni);
                if (hits == 0ull) {
                    const bool band = valid && inter >= q_lo * uni;
                    if (__ballot(band) != 0ull) hits = __ballot(band && (inter / uni > iou_thresh));
                }
                if (hits == 0ull) {
                    if (lane == kept) {
                        q_y0 = cy0;
                        q_x0 = cx0;
                        q_y1 = cy1;
                        q_x1 = cx1;
                        q_a = ca;
                    }
                    const int gi = __builtin_amdgcn_readlane(iv, t);
                    const int gs = __builtin_amdgcn_readlane(sbits, t);
                    if (lane == 0) {
                        og[kept] = gi;
                        os[kept] = __int_as_float(gs);
                    }
                    ++kept;
                }
            }
        }
        __syncthreads();
    };
    if (n <= 512 && max_out <= 64) {
        // Fast path (the usual case: a few dozen candidates per class).  TF's own formulation: visit the
        // candidates in descending score order and test each against the boxes selected so far.  The selected
        // boxes live one per lane in registers, so a candidate costs one IoU + one ballot: no LDS writes, no barriers.
        int *ord = reinterpret_cast<int *>(L.box + MAXC) - 512;   // tail of the box array is free (n <= 512)
        unsigned long long *key = reinterpret_cast<unsigned long long *>(ord) - 512;
        // rank sort on one 64-bit key per candidate: (score bits, ~box index) — scores are >= 0, so the bit pattern orders
        // like the float; ties fall to the lower box index
        for (int c = lane; c < n; c += 64)
            key[c] = ((unsigned long long)__float_as_uint(L.s[c]) << 32) | (unsigned)(~L.idx[c]);
        __syncthreads();
        for (int c = lane; c < n; c += 64) {
            const unsigned long long kc = key[c];
            int r = 0;
            for (int j = 0; j < n; ++j) r += key[j] > kc ? 1 : 0;
            ord[r] = c;
            // corners normalised once (tf_iou does it per call; min/max are idempotent, so the other paths are unaffected)
            const float4 q = L.box[c];
            L.box[c] = make_float4(fminf(q.x, q.z), fminf(q.y, q.w), fmaxf(q.x, q.z), fmaxf(q.y, q.w));
        }
        __syncthreads();
        // iou > thr is decided without the division whenever inter is clearly above / below thr*union (2e-6 guard band, 30x the
        // rounding of the quotient); only the band itself takes the exact fp32 division the reference performs
        const bool guard = iou_thresh > 0.f;
        const float thr_hi = guard ? iou_thresh * (1.f + 2e-6f) : INFINITY, thr_lo = guard ? iou_thresh * (1.f - 2e-6f) : -INFINITY;
        float my0 = 0.f, mx0 = 0.f, my1 = 0.f, mx1 = 0.f, ma = 0.f;
        // candidates are pulled into registers 64 at a time (lane t holds the t-th best of the block) and handed
        // out with v_readlane: the serial loop has no dependent LDS read in it
        for (int base = 0; base < n && kept < max_out; base += 64) {
            const int me = base + lane;
            const int pos = (me < n) ? ord[me] : 0;
            const float4 bb = L.box[pos];
            const float ar = (bb.z - bb.x) * (bb.w - bb.y);
            const float sv = L.s[pos];
            const int iv = L.idx[pos];
            const int cnt = min(64, n - base);
            for (int t = 0; t < cnt && kept < max_out; ++t) {
                const float cy0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bb.x), t));
                const float cx0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bb.y), t));
                const float cy1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bb.z), t));
                const float cx1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bb.w), t));
                const float ca = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ar), t));
                const float inter = fmaxf(fminf(cy1, my1) - fmaxf(cy0, my0), 0.f) * fmaxf(fminf(cx1, mx1) - fmaxf(cx0, mx0), 0.f);
                const float uni = ca + ma - inter;
                const bool valid = (lane < kept) && (ca > 0.f) && (ma > 0.f);          // zero-area boxes never overlap (TF)
                unsigned long long hits = __ballot(valid && inter > thr_hi * uni);
                if (hits == 0ull) {
                    const bool band = valid && inter >= thr_lo * uni;
                    if (__ballot(band) != 0ull) hits = __ballot(band && (inter / uni > iou_thresh));
                }
                if (hits == 0ull) {
                    if (lane == kept) {
                        my0 = cy0;
                        mx0 = cx0;
                        my1 = cy1;
                        mx1 = cx1;
                        ma = ca;
                    }
                    const int gi = __builtin_amdgcn_readlane(iv, t);
                    const float gs = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sv), t));
                    if (lane == 0) {
                        og[kept] = gi;
                        os[kept] = gs;
                    }
                    ++kept;
                }
            }
        }
    } else if (all_tied && sweep_ok) {
        // every candidate has the SAME score (saturated logits: what the random-init Darknet-53 of the benchmark produces, 1 000 - 10 647
        // candidates per class at exactly 1.0): descending (score, -index) order is ascending box index, which is the order the scan
        // compacted them in.  The LDS already holds the first chunk, sorted; later chunks are the next MAXC candidates of the same scan.
        int done = 0;
        for (;;) {
            const int nc = min(n - done, MAXC);
            sorted_sweep(nc, true);
            done += nc;
            if (kept >= max_out || done >= n) break;
            __syncthreads();
            int seen = 0;                                             // candidates ranked [done, done + MAXC) in scan order
            for (int base0 = 0; base0 < ntot; base0 += 64 * 8) {
                float sv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = base0 + u * 64 + lane;
                    sv[u] = (i < ntot) ? sc[i] : -INFINITY;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = base0 + u * 64 + lane;
                    const bool f = (i < ntot) && (sv[u] >= obj_thresh);
                    const unsigned long long m = __ballot(f);
                    const int pos = seen + __popcll(m & ((1ull << lane) - 1ull)) - done;
                    if (f && pos >= 0 && pos < MAXC) {
                        L.s[pos] = sv[u];
                        L.idx[pos] = i;
                    }
                    seen += __popcll(m);
                }
            }
            __syncthreads();
            const int nn = min(n - done, MAXC);
            for (int c0 = 0; c0 < nn; c0 += 256) {
                float4 t[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int q = c0 + u * 64 + lane;
                    t[u] = bx[q < nn ? L.idx[q] : 0];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int q = c0 + u * 64 + lane;
                    if (q < nn) L.box[q] = t[u];
                }
            }
            __syncthreads();
        }
    } else if (n <= MAXC && sweep_ok && (n <= 1024 ? 1024 : 2048) <= MAXC) {
        sorted_sweep(n, false);
    } else if (n <= MAXC) {
        kept = yk_wave_greedy_nms(
            n, L.s, L.idx, L.box, iou_thresh, max_out, [](const float4 &a, const float4 &d) { return tf_iou(d, a); },
            [&](int rank, int pos) {
                if (lane == 0) {
                    og[rank] = L.idx[pos];
                    os[rank] = L.s[pos];
                }
            },
            [](int) {});
    } else {
        // More candidates than the LDS holds (degenerate inputs: saturated logits - what 75 undamped random-init Darknet layers produce).  Greedy
        // NMS only ever needs the candidates in descending (score, -index) order, and a candidate's fate depends only on the boxes selected
        // before it - so the list is processed in CHUNKS of at most MAXC consecutive keys: find a key `lo` with count([lo, hi)) <= MAXC, gather
        // that interval into LDS, drop what the boxes selected so far suppress, run the LDS greedy on the rest, continue below lo.
        // `lo` comes from an MSD RADIX SELECT on the 64-bit key (score bits, ~index): one histogram pass (2^NBL bins in LDS) per level, the
        // bin that crosses MAXC is split by the next level - 1-2 passes for distinct scores, <= 5 when thousands of scores tie (then the
        // index bits decide).  Round 5 bisected on the key with one counting pass per step: ~45 passes per chunk, 1.44 ms of a 5.7 ms
        // Darknet-53 step at 32 images (profiles/r06_darknet_pipeline_kernel_stats.csv); the old in-place loop before that: 10 ms.
        constexpr int NBL = MAXC >= 2048 ? 11 : 10, NB = 1 << NBL;
        const unsigned long long key_min = (unsigned long long)__float_as_uint(fmaxf(obj_thresh, 0.f)) << 32;
        auto key_of = [&](float v, int i) { return ((unsigned long long)__float_as_uint(v) << 32) | (unsigned)(~i); };
        int *hist = L.idx;                                        // free between chunks (the selected indices live in selg / og)
        // histogram of the keys in [lo_c, hi_c) over bins of 2^sh keys
        auto hist_pass = [&](unsigned long long lo_c, unsigned long long hi_c, int sh) {
            for (int q = lane; q < NB; q += 64) hist[q] = 0;
            __syncthreads();
            for (int i0 = 0; i0 < ntot; i0 += 64 * 16) {          // sixteen independent loads in flight per lane: the pass is latency-bound
                float v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int i = i0 + u * 64 + lane;
                    v[u] = (i < ntot) ? sc[i] : -INFINITY;
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int i = i0 + u * 64 + lane;
                    if (i < ntot && v[u] >= obj_thresh) {
                        const unsigned long long k = key_of(v[u], i);
                        if (k >= lo_c && k < hi_c) atomicAdd(&hist[(int)((k - lo_c) >> sh)], 1);
                    }
                }
            }
            __syncthreads();
        };
        __shared__ int selg[256];                                 // global indices of the boxes selected so far (this wave only)
        const int *selp = (max_out <= 256) ? selg : og;
        unsigned long long hi = ~0ull;
        while (kept < max_out) {
            // ---- lo: the lowest key such that [lo, hi) holds <= MAXC candidates (a half-full chunk is good enough)
            unsigned long long lo_c = key_min, hi_c = hi, lo = key_min;
            if (hi_c <= lo_c) break;
            int taken = 0;                                        // candidates in [hi_c, hi): already inside the chunk
            for (;;) {
                const unsigned long long range = hi_c - lo_c;                     // >= 1
                const int bits = range > 1ull ? 64 - __builtin_clzll(range - 1ull) : 0;
                const int sh = bits > NBL ? bits - NBL : 0;
                const int nb = (int)(((range - 1ull) >> sh) + 1ull);              // <= NB
                hist_pass(lo_c, hi_c, sh);
                // walk the bins from the top: lane l owns bins nb-1-32l ... nb-32-32l (NB = 64 * 32 or 64 * 16)
                constexpr int PER = NB / 64;
                int mine[PER], sum = 0;
#pragma unroll
                for (int j = 0; j < PER; ++j) {
                    const int q = nb - 1 - (lane * PER + j);
                    mine[j] = q >= 0 ? hist[q] : 0;
                    sum += mine[j];
                }
                int pre = sum;                                                    // inclusive prefix over the lanes
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int t = __shfl_up(pre, o, 64);
                    if (lane >= o) pre += t;
                }
                const int before = pre - sum;                                     // candidates in the bins above this lane's
                const unsigned long long cross = __ballot(taken + pre > MAXC);
                if (cross == 0ull) {                                              // everything down to lo_c fits
                    lo = lo_c;
                    break;
                }
                const int cl = __ffsll((long long)cross) - 1;                     // the lane whose bins hold the crossing one
                int tq = 0, tk = 0;                                               // crossing bin (counted from the top), candidates above it
                if (lane == cl) {
                    int run = taken + before;
#pragma unroll
                    for (int j = 0; j < PER; ++j) {
                        if (run + mine[j] > MAXC) {
                            tq = lane * PER + j;
                            tk = run;
                            break;
                        }
                        run += mine[j];
                    }
                }
                tq = __shfl(tq, cl, 64);
                tk = __shfl(tk, cl, 64);
                const int qb = nb - 1 - tq;                                       // the crossing bin's index
                const unsigned long long b_lo = lo_c + ((unsigned long long)qb << sh), b_hi = qb + 1 >= nb ? hi_c : b_lo + (1ull << sh);
                taken = tk;
                if (taken >= MAXC / 2 || sh == 0) {                               // good enough / bins are single keys: the chunk ends above the crossing bin
                    lo = b_hi;
                    break;
                }
                lo_c = b_lo;                                                      // split the crossing bin
                hi_c = b_hi;
                // inside ONE score value the keys differ in ~index only, and ~i >= 2^32 - ntot: skip the empty part of the interval
                // (thousands of tied scores: the next level then bins the indices directly)
                if ((lo_c >> 32) == ((hi_c - 1ull) >> 32)) {
                    const unsigned long long first = (lo_c & 0xffffffff00000000ull) | (0x100000000ull - (unsigned long long)ntot);
                    if (first > lo_c && first < hi_c) lo_c = first;
                }
                __syncthreads();
            }
            if (lo >= hi) break;                                                  // (cannot happen: a bin of one key holds at most one candidate)
            // gather the chunk [lo, hi): scores and indices first, boxes in bulk afterwards (no load inside the ballot chain)
            __syncthreads();
            int nc = 0;
            for (int i0 = 0; i0 < ntot; i0 += 64 * 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = i0 + u * 64 + lane;
                    v[u] = (i < ntot) ? sc[i] : -INFINITY;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = i0 + u * 64 + lane;
                    bool f = (i < ntot) && (v[u] >= obj_thresh);
                    if (f) {
                        const unsigned long long k = key_of(v[u], i);
                        f = k >= lo && k < hi;
                    }
                    const unsigned long long mk = __ballot(f);
                    const int pos = nc + __popcll(mk & ((1ull << lane) - 1ull));
                    if (f && pos < MAXC) {
                        L.s[pos] = v[u];
                        L.idx[pos] = i;
                    }
                    nc += __popcll(mk);
                }
            }
            __syncthreads();
            for (int q0 = 0; q0 < nc; q0 += 256) {
                float4 t[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int q = q0 + u * 64 + lane;
                    t[u] = bx[q < nc ? L.idx[q] : 0];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int q = q0 + u * 64 + lane;
                    if (q < nc) L.box[q] = t[u];
                }
            }
            __syncthreads();
            if (sweep_ok && MAXC >= 2048) {                           // sorted sweep: the selected boxes carry over in registers
                sorted_sweep(min(nc, MAXC), false);
                if (lo == key_min) break;
                hi = lo;
                continue;
            }
            // candidates overlapping a box selected in an earlier chunk are dead already
            for (int q = lane; q < nc; q += 64) {
                const float4 cb = L.box[q];
                bool dead = false;
                for (int k = 0; k < kept && !dead; ++k) dead = tf_iou(cb, bx[selp[k]]) > iou_thresh;
                if (dead) L.s[q] = -INFINITY;
            }
            __syncthreads();
            const int base = kept;
            kept += yk_wave_greedy_nms(
                nc, L.s, L.idx, L.box, iou_thresh, max_out - base, [](const float4 &a, const float4 &d) { return tf_iou(d, a); },
                [&](int rank, int pos) {
                    if (lane == 0) {
                        og[base + rank] = L.idx[pos];
                        os[base + rank] = L.s[pos];
                        if (base + rank < 256) selg[base + rank] = L.idx[pos];
                    }
                },
                [](int) {});
            __threadfence_block();
            __syncthreads();
            if (lo == key_min) break;
            hi = lo;
        }
    }
    if (lane == 0) cnt[b * C + c] = kept;
}

// grid (batch), 256 threads: class-major concatenation (keras_inference.py:133-135); one thread per (class, rank) slot
__global__ void __launch_bounds__(256) compact_py_kernel(int ntot, int C, int max_out, const float4 *__restrict__ boxes,
                                                         const int *__restrict__ sel_g, const float *__restrict__ sel_s,
                                                         const int *__restrict__ cnt, float *__restrict__ dets,
                                                         int *__restrict__ counts, int *__restrict__ box_index) {
    extern __shared__ int base[];                   // [C + 1] exclusive prefix of the per-class counts
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) {
        int acc = 0;
        for (int c = 0; c < C; ++c) {
            base[c] = acc;
            acc += cnt[b * C + c];
        }
        base[C] = acc;
        counts[b] = acc;
    }
    __syncthreads();
    for (int slot = tid; slot < C * max_out; slot += 256) {
        const int c = slot / max_out, j = slot - c * max_out;
        if (j < base[c + 1] - base[c]) {
            const int g = sel_g[((size_t)b * C + c) * max_out + j];
            const float4 bb = boxes[(size_t)b * ntot + g];
            float *d = dets + ((size_t)b * C * max_out + base[c] + j) * 6;
            d[0] = bb.x;
            d[1] = bb.y;
            d[2] = bb.z;
            d[3] = bb.w;
            d[4] = sel_s[((size_t)b * C + c) * max_out + j];
            d[5] = (float)c;
            if (box_index) box_index[(size_t)b * C * max_out + base[c] + j] = g;
        }
    }
}

// The same concatenation ACROSS the batch: rows of image b start at offsets[b] (offsets[batch] = total), nothing between the images.
// `rows` / `offsets` / `box_index` may be device memory or pinned, device-mapped HOST memory: in the second case the detections land on
// the host at their live size (a few KB) with no device-to-host copy to size or to wait for.  grid (batch), 256 threads.
__global__ void __launch_bounds__(256) compact_packed_kernel(int ntot, int C, int max_out, int batch, const float4 *__restrict__ boxes,
                                                            const int *__restrict__ sel_g, const float *__restrict__ sel_s,
                                                            const int *__restrict__ cnt, float *rows, int *offsets, int *box_index,
                                                            float *__restrict__ dets, int *__restrict__ counts) {
    extern __shared__ int base[];                   // [C + 1] exclusive prefix of this image's per-class counts, [C + 1] = rows of the images before it
    __shared__ int part[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    int before = 0;
    for (int i = tid; i < b * C; i += 256) before += cnt[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o, 64);
    if ((tid & 63) == 0) part[tid >> 6] = before;
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int c = 0; c < C; ++c) {
            base[c] = acc;
            acc += cnt[b * C + c];
        }
        base[C] = acc;
        const int first = part[0] + part[1] + part[2] + part[3];
        base[C + 1] = first;
        offsets[b] = first;
        if (b == batch - 1) offsets[batch] = first + acc;
        if (counts) counts[b] = acc;
    }
    __syncthreads();
    const int first = base[C + 1];
    for (int slot = tid; slot < C * max_out; slot += 256) {
        const int c = slot / max_out, j = slot - c * max_out;
        if (j < base[c + 1] - base[c]) {
            const int g = sel_g[((size_t)b * C + c) * max_out + j];
            const float4 bb = boxes[(size_t)b * ntot + g];
            const float sv = sel_s[((size_t)b * C + c) * max_out + j];
            float *d = rows + ((size_t)first + base[c] + j) * 6;
            reinterpret_cast<float2 *>(d)[0] = make_float2(bb.x, bb.y);        // rows are 24 bytes: 8-byte aligned
            reinterpret_cast<float2 *>(d)[1] = make_float2(bb.z, bb.w);
            reinterpret_cast<float2 *>(d)[2] = make_float2(sv, (float)c);
            if (box_index) box_index[(size_t)first + base[c] + j] = g;
            if (dets) {
                float *e = dets + ((size_t)b * C * max_out + base[c] + j) * 6;
                e[0] = bb.x; e[1] = bb.y; e[2] = bb.z; e[3] = bb.w; e[4] = sv; e[5] = (float)c;
            }
        }
    }
}

static int decode_impl(const yk_decode_cfg_t *cfg, const float *const *d_pred, int batch, const float *d_image_hw, float obj_thresh,
                       float iou_thresh, int max_out, float *d_dets, int32_t *d_counts, int32_t *d_box_index, float *rows, int32_t *offsets,
                       int32_t *rows_index, void *stream);

extern "C" int yk_decode_py(const yk_decode_cfg_t *cfg, const float *const *d_pred, int batch, const float *d_image_hw,
                            float obj_thresh, float iou_thresh, int max_out, float *d_dets, int32_t *d_counts,
                            void *stream) {
    return yk_decode_py_ex(cfg, d_pred, batch, d_image_hw, obj_thresh, iou_thresh, max_out, d_dets, d_counts, nullptr, stream);
}

extern "C" int yk_decode_py_ex(const yk_decode_cfg_t *cfg, const float *const *d_pred, int batch, const float *d_image_hw,
                               float obj_thresh, float iou_thresh, int max_out, float *d_dets, int32_t *d_counts,
                               int32_t *d_box_index, void *stream) {
    if (!d_dets || !d_counts) {
        yk_set_error("yk_decode_py: bad argument");
        return YK_ERR_ARG;
    }
    return decode_impl(cfg, d_pred, batch, d_image_hw, obj_thresh, iou_thresh, max_out, d_dets, d_counts, d_box_index, nullptr, nullptr, nullptr, stream);
}

extern "C" int yk_decode_py_packed(const yk_decode_cfg_t *cfg, const float *const *d_pred, int batch, const float *d_image_hw,
                                   float obj_thresh, float iou_thresh, int max_out, float *rows, int32_t *offsets, int32_t *rows_index,
                                   float *d_dets, int32_t *d_counts, void *stream) {
    if (!rows || !offsets) {
        yk_set_error("yk_decode_py_packed: bad argument");
        return YK_ERR_ARG;
    }
    return decode_impl(cfg, d_pred, batch, d_image_hw, obj_thresh, iou_thresh, max_out, d_dets, d_counts, nullptr, rows, offsets, rows_index, stream);
}

static int decode_impl(const yk_decode_cfg_t *cfg, const float *const *d_pred, int batch, const float *d_image_hw, float obj_thresh,
                       float iou_thresh, int max_out, float *d_dets, int32_t *d_counts, int32_t *d_box_index, float *rows, int32_t *offsets,
                       int32_t *rows_index, void *stream) {
    if (!cfg || !d_pred || batch <= 0 || max_out <= 0 || cfg->n_layers <= 0 ||
        cfg->n_layers > YK_MAX_LAYERS || cfg->anchor_num <= 0 || cfg->anchor_num > YK_MAX_ANCHORS ||
        cfg->class_num <= 0) {
        yk_set_error("yk_decode_py: bad argument");
        return YK_ERR_ARG;
    }
    decode_args a;
    memset(&a, 0, sizeof(a));
    a.L = cfg->n_layers;
    a.A = cfg->anchor_num;
    a.C = cfg->class_num;
    a.E = 5 + cfg->class_num;
    a.in_h = cfg->in_h;
    a.in_w = cfg->in_w;
    int off = 0;
    for (int l = 0; l < a.L; ++l) {
        if (!d_pred[l] || cfg->out_h[l] <= 0 || cfg->out_w[l] <= 0) {
            yk_set_error("yk_decode_py: layer %d null / empty", l);
            return YK_ERR_ARG;
        }
        a.out_h[l] = cfg->out_h[l];
        a.out_w[l] = cfg->out_w[l];
        a.off[l] = off;
        off += cfg->out_h[l] * cfg->out_w[l] * a.A;
        a.pred[l] = d_pred[l];
        for (int n = 0; n < a.A; ++n) {
            a.anchors[l][n][0] = cfg->anchors[l][n][0];
            a.anchors[l][n][1] = cfg->anchors[l][n][1];
        }
    }
    a.off[a.L] = off;
    const int ntot = off;
    int dev = yk_current_device();
    if (dev < 0) {
        yk_set_error("yk_decode_py: no HIP device");
        return YK_ERR_NO_DEVICE;
    }
    const size_t box_b = (size_t)batch * ntot * sizeof(float4);
    const size_t sc_b = (size_t)batch * a.C * ntot * sizeof(float);
    const size_t sel_b = (size_t)batch * a.C * max_out * sizeof(int);
    const size_t cnt_b = (size_t)batch * a.C * sizeof(int);
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    char *ws = (char *)yk_scratch(dev, stream, 0, up(box_b) + up(sc_b) + 2 * up(sel_b) + up(cnt_b));
    if (!ws) return YK_ERR_NOMEM;
    float4 *boxes = (float4 *)ws;
    float *scores_t = (float *)(ws + up(box_b));
    int *sel_g = (int *)(ws + up(box_b) + up(sc_b));
    float *sel_s = (float *)(ws + up(box_b) + up(sc_b) + up(sel_b));
    int *cnt = (int *)(ws + up(box_b) + up(sc_b) + 2 * up(sel_b));
    hipStream_t st = (hipStream_t)stream;
    const int total = batch * ntot;
    hipLaunchKernelGGL(decode_py_kernel, dim3((total + 255) / 256), dim3(256), 0, st, a, batch, d_image_hw, boxes,
                       scores_t);
    if (ntot <= 1088)
        hipLaunchKernelGGL(nms_py_kernel<1088>, dim3(a.C, batch), dim3(64), 0, st, ntot, a.C, obj_thresh, iou_thresh, max_out, boxes, scores_t,
                           sel_g, sel_s, cnt);
    else
        hipLaunchKernelGGL(nms_py_kernel<2048>, dim3(a.C, batch), dim3(64), 0, st, ntot, a.C, obj_thresh, iou_thresh, max_out, boxes, scores_t,
                           sel_g, sel_s, cnt);
    if (rows)
        hipLaunchKernelGGL(compact_packed_kernel, dim3(batch), dim3(256), (a.C + 2) * sizeof(int), st, ntot, a.C, max_out, batch, boxes, sel_g,
                           sel_s, cnt, rows, offsets, rows_index, d_dets, d_counts);
    else
        hipLaunchKernelGGL(compact_py_kernel, dim3(batch), dim3(256), (a.C + 1) * sizeof(int), st, ntot, a.C, max_out, boxes, sel_g, sel_s, cnt,
                           d_dets, d_counts, d_box_index);
    YK_HIP(hipGetLastError());
    return YK_OK;
}
