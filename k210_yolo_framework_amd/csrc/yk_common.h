// yk_common.h — shared helpers of libyolo_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "yolo_hip.h"

#define YK_WAVE 64

void yk_set_error(const char *fmt, ...);

#define YK_HIP(call)                                                                           \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            yk_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_));  \
            return (e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice) ? YK_ERR_NO_DEVICE  \
                                                                           : YK_ERR_HIP;      \
        }                                                                                      \
    } while (0)

// grow-only device scratch per (device, stream); safe because work on one stream is ordered.
void *yk_scratch(int device, void *stream, int slot, size_t bytes);

static inline int yk_current_device() {
    int d = -1;
    if (hipGetDevice(&d) != hipSuccess) return -1;
    return d;
}

__device__ __forceinline__ float yk_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
