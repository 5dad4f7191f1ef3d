// yk_nms.h — one-wavefront greedy NMS primitive shared by the C-mode region layer
// (region_layer.c:256-283) and the Python-mode per-class NMS (keras_inference.py:122-131).
//
// Greedy sort-NMS is restated as "iterated arg-max": the best remaining candidate is by
// construction not suppressed by any better one, so it is kept; everything it overlaps by more
// than the threshold dies; repeat.  That gives the same survivor set and the same (score-
// descending) order as sort-then-sweep without a sort, and one iteration costs one LDS pass of
// n/64 elements per lane plus a 6-step cross-lane reduction.  Ties: ascending box index.
//
// Must be called by a workgroup that is exactly ONE wavefront (64 threads).
#pragma once
#include "yk_common.h"

#define YK_NMS_MAXC 2048 /* candidates held in LDS per (image, class); beyond that -> global-memory path */

struct yk_cand_lds {
    float s[YK_NMS_MAXC];
    int idx[YK_NMS_MAXC];
    float4 box[YK_NMS_MAXC];
};

// arg-max over (score desc, idx asc) across the wave; score == -INF means "not a candidate".
__device__ __forceinline__ void yk_wave_argmax(float &best, int &bidx, int &bpos) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ob = __shfl_xor(best, o, 64);
        int oi = __shfl_xor(bidx, o, 64);
        int op = __shfl_xor(bpos, o, 64);
        bool take = (op >= 0) && (bpos < 0 || ob > best || (ob == best && oi < bidx));
        if (take) {
            best = ob;
            bidx = oi;
            bpos = op;
        }
    }
}

// s/idx/box: n candidates in LDS.  keep(rank, pos) is called by every lane (uniform) for each
// survivor in score order; kill(pos) by the lane that owns the suppressed candidate.
template <class IoU, class Keep, class Kill>
__device__ __forceinline__ int yk_wave_greedy_nms(int n, float *s, const int *idx, const float4 *box, float thr,
                                                  int cap, IoU iou, Keep keep, Kill kill) {
    const int lane = threadIdx.x & 63;
    int kept = 0;
    while (kept < cap) {
        float best = -INFINITY;
        int bidx = 0x7fffffff, bpos = -1;
        for (int i = lane; i < n; i += 64) {
            float v = s[i];
            if (v > -INFINITY && (bpos < 0 || v > best || (v == best && idx[i] < bidx))) {
                best = v;
                bidx = idx[i];
                bpos = i;
            }
        }
        yk_wave_argmax(best, bidx, bpos);
        if (bpos < 0) break;
        keep(kept, bpos);
        const float4 wb = box[bpos];
        __syncthreads();
        if (lane == 0) s[bpos] = -INFINITY;
        __syncthreads();
        for (int i = lane; i < n; i += 64) {
            if (s[i] > -INFINITY && iou(wb, box[i]) > thr) {
                s[i] = -INFINITY;
                kill(i);
            }
        }
        __syncthreads();
        ++kept;
    }
    return kept;
}
