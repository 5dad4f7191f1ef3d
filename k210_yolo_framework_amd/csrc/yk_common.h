// yk_common.h — shared helpers of libyolo_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "yolo_hip.h"

#define YK_WAVE 64

// Tuning switches (YK_IGEMM_FORCE, YK_SPLIT_FORCE, YK_PIPE, ...) exist only in development builds (`make DEV=1`, -DYK_DEV): the
// shipped library never reads the environment on its launch path.
#include <stdlib.h>
#ifdef YK_DEV
static inline const char *yk_dev_env(const char *name) { return getenv(name); }
#else
static inline const char *yk_dev_env(const char *) { return nullptr; }
#endif

// the few switches the shipped library does read, at plan creation only (README "Environment switches")
static inline bool yk_env_flag(const char *name, bool dflt) {
    const char *e = getenv(name);
    if (!e || !*e) return dflt;
    return e[0] != '0';
}

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
void yk_scratch_release_stream(void *stream);

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

// expf as glibc >= 2.27 computes it (sysdeps/ieee754/flt-32/e_expf.c, the ARM optimized-routines algorithm): x*32/ln2 = k + r,
// 2^(k/32) from a 32-entry table, a cubic in r, everything in double, ONE rounding to float at the end.  The reference's C path
// (region_layer.c:75,100,171-172) calls libm's expf on the host; evaluating the SAME approximation on the device makes the C-mode
// outputs bit-identical to it (the double-precision evaluation order only matters at the 2^-53 level, i.e. never for the final
// float; checked here against glibc 2.35 on 1.5e8 inputs: 0 mismatches).  Device libm's expf differs from glibc in ~1 ulp cases.
__device__ __forceinline__ float yk_expf_glibc(float x) {
    static constexpr unsigned long long T[32] = {
        0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull,
        0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull,
        0x3feedea64c123422ull, 0x3feece086061892dull, 0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull,
        0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
        0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull, 0x3feee89f995ad3adull,
        0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
        0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
    constexpr double N = 32.0, InvLn2N = 0x1.71547652b82fep+0 * N;
    constexpr double C0 = 0x1.c6af84b912394p-5 / N / N / N, C1 = 0x1.ebfce50fac4f3p-3 / N / N, C2 = 0x1.62e42ff0c52d6p-1 / N;
    if (x != x) return x + x;
    if (x > 0x1.62e42ep6f) return __builtin_huge_valf();
    if (x < -0x1.9fe368p6f) return 0.f;
    const double z = InvLn2N * (double)x;
    const double kd = rint(z);                       // round-half-even, as the 0x1.8p52 shift trick does
    const long long ki = (long long)kd;
    const double r = z - kd;
    const unsigned long long t = T[ki & 31] + ((unsigned long long)ki << 47);
    const double s = __longlong_as_double((long long)t);
    double y = C2 * r + 1.0;
    y = (C0 * r + C1) * (r * r) + y;
    return (float)(y * s);
}
