// yk_pre.hip — GPU pre-processing immediately before the path:
// Helper._process_img's letterbox (tools/utils.py:378-399): scale = min(in_wh / img_wh),
// translation = ((in_wh - img_wh*scale)/2).astype(int), then
// skimage.transform.warp(img, AffineTransform(scale, translation).inverse, output_shape=in_hw, order=1,
//                        mode='constant', cval=0, preserve_range=True).astype('uint8')
// restated as: out(x,y) = bilinear(in, ((x-tx)/s, (y-ty)/s)) with zero outside, float64 math, truncating cast.
// skimage 0.15 is third-party and absent: parity unpinned except for the identity case (dog.jpg).
// The per-image max normalisation that follows (utils.py:405) is fused into the stem conv (yk_run_u8).
#include "yk_common.h"

__global__ void __launch_bounds__(256) letterbox_u8_kernel(const uint8_t *__restrict__ src, int batch, int sh, int sw,
                                                           uint8_t *__restrict__ dst, int dh, int dw, double scale, int tx, int ty) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)batch * dh * dw;
    if (idx >= total) return;
    const int x = (int)(idx % dw), y = (int)((idx / dw) % dh), b = (int)(idx / ((size_t)dw * dh));
    const double fx = ((double)x - (double)tx) / scale, fy = ((double)y - (double)ty) / scale;
    uint8_t *o = dst + idx * 3;
    if (!(fx > -1.0 && fx < (double)sw && fy > -1.0 && fy < (double)sh)) {
        o[0] = o[1] = o[2] = 0;
        return;
    }
    const double x0f = floor(fx), y0f = floor(fy);
    const int x0 = (int)x0f, y0 = (int)y0f;
    const double ax = fx - x0f, ay = fy - y0f;
    const uint8_t *im = src + (size_t)b * sh * sw * 3;
    auto px = [&](int yy, int xx, int c) -> double {
        return (yy >= 0 && yy < sh && xx >= 0 && xx < sw) ? (double)im[((size_t)yy * sw + xx) * 3 + c] : 0.0;
    };
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double v = px(y0, x0, c) * (1.0 - ay) * (1.0 - ax) + px(y0, x0 + 1, c) * (1.0 - ay) * ax +
                         px(y0 + 1, x0, c) * ay * (1.0 - ax) + px(y0 + 1, x0 + 1, c) * ay * ax;
        o[c] = (uint8_t)v;    // astype('uint8'): truncation
    }
}

extern "C" int yk_letterbox_u8(const uint8_t *d_src, int batch, int src_h, int src_w, uint8_t *d_dst, int dst_h, int dst_w,
                               void *stream) {
    if (!d_src || !d_dst || batch <= 0 || src_h <= 0 || src_w <= 0 || dst_h <= 0 || dst_w <= 0) {
        yk_set_error("yk_letterbox_u8: bad argument");
        return YK_ERR_ARG;
    }
    if (yk_current_device() < 0) {
        yk_set_error("yk_letterbox_u8: no HIP device");
        return YK_ERR_NO_DEVICE;
    }
    const double sx = (double)dst_w / (double)src_w, sy = (double)dst_h / (double)src_h;
    const double scale = sx < sy ? sx : sy;                                    // utils.py:381-382
    const int tx = (int)(((double)dst_w - (double)src_w * scale) / 2.0);       // .astype(int): truncation, utils.py:385
    const int ty = (int)(((double)dst_h - (double)src_h * scale) / 2.0);
    const size_t total = (size_t)batch * dst_h * dst_w;
    hipLaunchKernelGGL(letterbox_u8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_src, batch,
                       src_h, src_w, d_dst, dst_h, dst_w, scale, tx, ty);
    YK_HIP(hipGetLastError());
    return YK_OK;
}
