// kf_render.hip - renderDepthKernel and renderTrackKernel (cpp:814-878, commons.h:86-147); renderVolume marches rays and lives in kf_raycast.hip.
#include "kf_dev.h"

namespace kf {

// renderDepthKernel (cpp:814-842) with the double-precision ramp of commons.h:86-147
__device__ __forceinline__ uchar4 gs2rgb(double h) {
    double r = 0, g = 0, b = 0;
    const double v = 0.75, m = 0.25, sv = 0.6667;
    h *= 6.0;
    const int sextant = (int)h;
    const double fract = h - sextant;
    const double vsf = v * sv * fract, mid1 = m + vsf, mid2 = v - vsf;
    switch (sextant) {
    case 0: r = v; g = mid1; b = m; break;
    case 1: r = mid2; g = v; b = m; break;
    case 2: r = m; g = v; b = mid1; break;
    case 3: r = m; g = mid2; b = v; break;
    case 4: r = mid1; g = m; b = v; break;
    case 5: r = v; g = m; b = mid2; break;
    default: break;
    }
    return make_uchar4((unsigned char)(r * 255), (unsigned char)(g * 255), (unsigned char)(b * 255), 0);
}

__device__ __forceinline__ void render_depth_pixel(uchar4 *__restrict__ out, const float *__restrict__ depth, unsigned W, unsigned H,
                                                   float nearPlane, float farPlane) {
    const unsigned x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= W || y >= H) return;
    const float rangeScale = 1 / (farPlane - nearPlane);
    const unsigned p = y * W + x;
    const float d = depth[p];
    uchar4 o;
    if (d < nearPlane) o = make_uchar4(255, 255, 255, 0);
    else if (d > farPlane) o = make_uchar4(0, 0, 0, 0);
    else o = gs2rgb((d - nearPlane) * rangeScale);
    out[p] = o;
}

// renderTrackKernel (cpp:844-878) on the int32 result map the fused ICP pass keeps
__device__ __forceinline__ void render_track_pixel(uchar4 *__restrict__ out, const int *__restrict__ result, unsigned W, unsigned H) {
    const unsigned x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= W || y >= H) return;
    uchar4 o;
    switch (result[x + y * W]) {
    case 1: o = make_uchar4(128, 128, 128, 0); break;
    case -1: o = make_uchar4(0, 0, 0, 0); break;
    case -2: o = make_uchar4(255, 0, 0, 0); break;
    case -3: o = make_uchar4(0, 255, 0, 0); break;
    case -4: o = make_uchar4(0, 0, 255, 0); break;
    case -5: o = make_uchar4(255, 255, 0, 0); break;
    default: o = make_uchar4(255, 128, 128, 0); break;
    }
    out[x + y * W] = o;
}

__global__ __launch_bounds__(256) void k_render_depth(uchar4 *__restrict__ out, const float *__restrict__ depth,
                                                      unsigned W, unsigned H, float nearPlane, float farPlane) {
    render_depth_pixel(out, depth, W, H, nearPlane, farPlane);
}

__global__ __launch_bounds__(256) void k_render_track(uchar4 *__restrict__ out, const int *__restrict__ result,
                                                      unsigned W, unsigned H) {
    render_track_pixel(out, result, W, H);
}

// both images of a queued frame in one launch (kf_pipeline.hip run_deferred_renders): blockIdx.z = 0 depth, 1 track
__global__ __launch_bounds__(256) void k_render_depth_track(uchar4 *__restrict__ out_depth, const float *__restrict__ depth,
                                                            uchar4 *__restrict__ out_track, const int *__restrict__ result,
                                                            unsigned W, unsigned H, float nearPlane, float farPlane) {
    if (blockIdx.z == 0) render_depth_pixel(out_depth, depth, W, H, nearPlane, farPlane);
    else render_track_pixel(out_track, result, W, H);
}

// Images from HBM into pinned host memory by 16-byte stores over PCIe (the read-back of a queued frame's renders as a kernel
// instead of copy-engine commands: KFH_RENDER_COPY_KERNEL, kf_pipeline.hip queue_readback / issue_readback).  Up to three images
// per launch, walked as one range of 16-byte pieces.
__global__ __launch_bounds__(256) void k_copy_out(launch::CopyOutSegments g) {
    const size_t total = g.n16[0] + g.n16[1] + g.n16[2];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t j = i;
        int k = 0;
        if (j >= g.n16[0]) { j -= g.n16[0]; k = 1; if (j >= g.n16[1]) { j -= g.n16[1]; k = 2; } }
        g.dst[k][j] = g.src[k][j];
    }
}

namespace launch {

void copy_out(const CopyOutSegments &g, unsigned blocks, hipStream_t s) {
    k_copy_out<<<blocks, 256, 0, s>>>(g); launched("k_copy_out");
}

void render_depth(uchar4 *out, const float *depth, unsigned W, unsigned H, float nearPlane, float farPlane, hipStream_t s) {
    k_render_depth<<<grid_rows(W, H), block_rows(), 0, s>>>(out, depth, W, H, nearPlane, farPlane); launched("k_render_depth");
}

void render_depth_track(uchar4 *out_depth, const float *depth, uchar4 *out_track, const int *result_map, unsigned W, unsigned H,
                        float nearPlane, float farPlane, hipStream_t s) {
    const dim3 g = grid_rows(W, H);
    k_render_depth_track<<<dim3(g.x, g.y, 2), block_rows(), 0, s>>>(out_depth, depth, out_track, result_map, W, H, nearPlane, farPlane); launched("k_render_depth_track");
}

void render_track(uchar4 *out, const int *result_map, unsigned W, unsigned H, hipStream_t s) {
    k_render_track<<<grid_rows(W, H), block_rows(), 0, s>>>(out, result_map, W, H); launched("k_render_track");
}

}  // namespace launch
}  // namespace kf
