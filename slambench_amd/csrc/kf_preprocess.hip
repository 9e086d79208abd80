// kf_preprocess.hip - initVolume, mm2meters + bilateral filter, half-sample, depth2vertex, vertex2normal and the fused input pyramid
// (cpp:147-249, 562-626): the stand-alone kernels over the tile routines of kf_dev.h, and their launchers.
#include "kf_dev.h"

namespace kf {

// =====================================================================================
// initVolumeKernel  (cpp:147-157, commons.h:176-180): every voxel <- short2(32766, 0)
// =====================================================================================
__global__ __launch_bounds__(256) void k_init_volume(uint32_t *__restrict__ vol, size_t n) {
    const uint32_t v = 0x00007FFEu;  // little-endian {x = 32766, y = 0}
    const size_t n4 = n / 4;
    uint4 *v4 = reinterpret_cast<uint4 *>(vol);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) v4[i] = make_uint4(v, v, v, v);
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) vol[i] = v;
}

template <bool FROM_MM, bool FD>
__global__ __launch_bounds__(PBX *PBY) void k_bilateral(float *__restrict__ floatDepth, float *__restrict__ out,
                                                        const void *__restrict__ inv, unsigned W, unsigned H,
                                                        unsigned inW, int ratio, Gauss5 g, float e_d_squared_2, float rcp) {
    __shared__ float tile[PBY + 2 * PR][PBX + 2 * PR + 1];
    __shared__ double exp_tab[32];
    const int tid = threadIdx.y * PBX + threadIdx.x;
    if (tid < 32) exp_tab[tid] = __builtin_bit_cast(double, kExp2fTable[tid]);
    bilateral_block<FROM_MM, FD>(tile, exp_tab, blockIdx.x, blockIdx.y, tid, true, floatDepth, out, inv, W, H, inW, ratio, g, e_d_squared_2, rcp);
}

__global__ __launch_bounds__(256) void k_mm2meters(float *__restrict__ out, unsigned W, unsigned H,
                                                   const uint16_t *__restrict__ in, unsigned inW, int ratio) {
    const unsigned x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= W || y >= H) return;
    out[x + W * y] = in[x * ratio + inW * y * ratio] / 1000.0f;
}

// =====================================================================================
// halfSampleRobustImageKernel (cpp:591-626), r = 1: robust mean of the 2x2 block under (2x, 2y)
// =====================================================================================
__global__ __launch_bounds__(256) void k_half_sample(float *__restrict__ out, const float *__restrict__ in,
                                                     unsigned inW, unsigned outW, unsigned outH, float e_d) {
    const unsigned x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= outW || y >= outH) return;
    const float2 r0 = *reinterpret_cast<const float2 *>(in + 2 * x + (size_t)(2 * y) * inW);
    const float2 r1 = *reinterpret_cast<const float2 *>(in + 2 * x + (size_t)(2 * y + 1) * inW);
    const float center = r0.x;
    float sum = 0.0f, t = 0.0f;
    // (i, j) = (0,0), (0,1), (1,0), (1,1): i is the row offset, j the column offset (cpp:607-608)
    if (fabsf(r0.x - center) < e_d) { sum += 1.0f; t += r0.x; }
    if (fabsf(r0.y - center) < e_d) { sum += 1.0f; t += r0.y; }
    if (fabsf(r1.x - center) < e_d) { sum += 1.0f; t += r1.x; }
    if (fabsf(r1.y - center) < e_d) { sum += 1.0f; t += r1.y; }
    out[x + y * outW] = t / sum;
}

__global__ __launch_bounds__(256) void k_depth2vertex(float4 *__restrict__ vertex, const float *__restrict__ depth,
                                                      unsigned W, unsigned H, M4 invK) {
    const unsigned x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= W || y >= H) return;
    const f3 v = back_project(depth[x + y * W], x, y, invK);
    vertex[x + y * W] = make_float4(v.x, v.y, v.z, 0.f);
}

__global__ __launch_bounds__(256) void k_vertex2normal(float4 *__restrict__ out, const float4 *__restrict__ in,
                                                       unsigned W, unsigned H) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= (int)W || y >= (int)H) return;
    const int xl = imax(x - 1, 0), xr = imin(x + 1, (int)W - 1), yu = imax(y - 1, 0), yd = imin(y + 1, (int)H - 1);
    const float4 l = in[xl + W * y], r = in[xr + W * y], u = in[x + W * yu], d = in[x + W * yd];
    store_normal(out + x + y * W, mk3(l.x, l.y, l.z), mk3(r.x, r.y, r.z), mk3(u.x, u.y, u.z), mk3(d.x, d.y, d.z));
}

// Fused form used by the pipeline: the 34x10 depth tile (1-pixel halo, clamp-to-edge) is staged in
// LDS once; the centre vertex and its four neighbours are back-projected from it with the same
// expression depth2vertexKernel uses, so both outputs carry the bits of the two-pass version.
__global__ __launch_bounds__(PBX *PBY) void k_vertex_normal(float4 *__restrict__ vertex, float4 *__restrict__ normal,
                                                            const float *__restrict__ depth, unsigned W, unsigned H,
                                                            M4 invK) {
    __shared__ float tile[PBY + 2][PBX + 2 + 1];
    const int x0 = blockIdx.x * PBX, y0 = blockIdx.y * PBY;
    const int tid = threadIdx.y * PBX + threadIdx.x;
    constexpr int TW = PBX + 2, TH = PBY + 2;
    for (int idx = tid; idx < TW * TH; idx += PBX * PBY) {
        const int ty = idx / TW, tx = idx - ty * TW;
        const int gx = imax(0, imin(x0 + tx - 1, (int)W - 1));
        const int gy = imax(0, imin(y0 + ty - 1, (int)H - 1));
        tile[ty][tx] = depth[gx + gy * W];
    }
    __syncthreads();
    const int x = x0 + threadIdx.x, y = y0 + threadIdx.y;
    if (x >= (int)W || y >= (int)H) return;
    const int tx = threadIdx.x + 1, ty = threadIdx.y + 1;
    const int xl = imax(x - 1, 0), xr = imin(x + 1, (int)W - 1), yu = imax(y - 1, 0), yd = imin(y + 1, (int)H - 1);
    const f3 c = back_project(tile[ty][tx], x, y, invK);
    const f3 l = back_project(tile[ty][tx - 1], xl, y, invK);
    const f3 r = back_project(tile[ty][tx + 1], xr, y, invK);
    const f3 u = back_project(tile[ty - 1][tx], x, yu, invK);
    const f3 d = back_project(tile[ty + 1][tx], x, yd, invK);
    vertex[x + y * W] = make_float4(c.x, c.y, c.z, 0.f);
    store_normal(normal + x + y * W, l, r, u, d);
}

template <int NLEV>
__global__ __launch_bounds__(PYR_THREADS) void k_pyramid(PyramidArgs a) {
    __shared__ float lds[pyr_lds_floats<NLEV>()];
    pyramid_block<NLEV, PYR_THREADS>(a, lds, blockIdx.x, blockIdx.y, threadIdx.x);
}

namespace launch {

void init_volume(int16_t *vol, unsigned sx, unsigned sy, unsigned sz, hipStream_t s) {
    const size_t n = (size_t)sx * sy * sz;
    k_init_volume<<<2048, 256, 0, s>>>(reinterpret_cast<uint32_t *>(vol), n); launched("k_init_volume");
}

void preprocess(float *floatDepth, float *filtered, unsigned W, unsigned H, const uint16_t *in, unsigned inW, int ratio,
                const Gauss5 &g, float e_d, hipStream_t s) {
    const dim3 grid((W + PBX - 1) / PBX, (H + PBY - 1) / PBY), block(PBX, PBY);
    const float d = e_d * e_d * 2;
    if (divisor_verified(d)) k_bilateral<true, true><<<grid, block, 0, s>>>(floatDepth, filtered, in, W, H, inW, ratio, g, d, 1.0f / d);
    else k_bilateral<true, false><<<grid, block, 0, s>>>(floatDepth, filtered, in, W, H, inW, ratio, g, d, 0.f);
    launched("k_bilateral");
}

void mm2meters(float *out, unsigned W, unsigned H, const uint16_t *in, unsigned inW, int ratio, hipStream_t s) {
    k_mm2meters<<<grid_rows(W, H), block_rows(), 0, s>>>(out, W, H, in, inW, ratio); launched("k_mm2meters");
}

void bilateral(float *out, const float *in, unsigned W, unsigned H, const Gauss5 &g, float e_d, hipStream_t s) {
    const dim3 grid((W + PBX - 1) / PBX, (H + PBY - 1) / PBY), block(PBX, PBY);
    const float d = e_d * e_d * 2;
    if (divisor_verified(d)) k_bilateral<false, true><<<grid, block, 0, s>>>(nullptr, out, in, W, H, W, 1, g, d, 1.0f / d);
    else k_bilateral<false, false><<<grid, block, 0, s>>>(nullptr, out, in, W, H, W, 1, g, d, 0.f);
    launched("k_bilateral");
}

void half_sample(float *out, const float *in, unsigned inW, unsigned inH, float e_d, hipStream_t s) {
    const unsigned outW = inW / 2, outH = inH / 2;
    k_half_sample<<<grid_rows(outW, outH), block_rows(), 0, s>>>(out, in, inW, outW, outH, e_d); launched("k_half_sample");
}

void depth2vertex(float4 *vertex, const float *depth, unsigned W, unsigned H, const M4 &invK, hipStream_t s) {
    k_depth2vertex<<<grid_rows(W, H), block_rows(), 0, s>>>(vertex, depth, W, H, invK); launched("k_depth2vertex");
}

void vertex2normal(float4 *normal, const float4 *vertex, unsigned W, unsigned H, hipStream_t s) {
    k_vertex2normal<<<grid_rows(W, H), block_rows(), 0, s>>>(normal, vertex, W, H); launched("k_vertex2normal");
}

void vertex_normal(float4 *vertex, float4 *normal, const float *depth, unsigned W, unsigned H, const M4 &invK, hipStream_t s) {
    const dim3 grid((W + PBX - 1) / PBX, (H + PBY - 1) / PBY), block(PBX, PBY);
    k_vertex_normal<<<grid, block, 0, s>>>(vertex, normal, depth, W, H, invK); launched("k_vertex_normal");
}

bool pyramid(int n_levels, float *const *scaled, float4 *const *vertex, float4 *const *normal, const float4 *const *stale_normal,
             const M4 *invK, unsigned W, unsigned H, float e_d, hipStream_t s) {
    if (!pyramid_fusable(n_levels, W, H)) return false;   // keep odd sizes on the simple path
    PyramidArgs a;
    for (int l = 0; l < PYR_MAX_LEVELS; ++l) {
        a.scaled[l] = l < n_levels ? scaled[l] : nullptr;
        a.vertex[l] = l < n_levels ? vertex[l] : nullptr;
        a.normal[l] = l < n_levels ? normal[l] : nullptr;
        a.stale_normal[l] = (l < n_levels && stale_normal) ? stale_normal[l] : nullptr;
        a.invK[l] = l < n_levels ? invK[l] : identity4();
    }
    a.W = W; a.H = H; a.e_d = e_d; a.through = false;
    const dim3 grid((W + PYR_T - 1) / PYR_T, (H + PYR_T - 1) / PYR_T);
    switch (n_levels) {
    case 1: k_pyramid<1><<<grid, 256, 0, s>>>(a); launched("k_pyramid"); break;
    case 2: k_pyramid<2><<<grid, 256, 0, s>>>(a); launched("k_pyramid"); break;
    case 3: k_pyramid<3><<<grid, 256, 0, s>>>(a); launched("k_pyramid"); break;
    default: k_pyramid<4><<<grid, 256, 0, s>>>(a); launched("k_pyramid"); break;
    }
    return true;
}

unsigned bilateral_tiles(unsigned W, unsigned H) { return ((W + PBX - 1) / PBX) * ((H + PBY - 1) / PBY); }
unsigned pyramid_tiles(unsigned W, unsigned H) { return ((W + PYR_T - 1) / PYR_T) * ((H + PYR_T - 1) / PYR_T); }
bool pyramid_fusable(int n_levels, unsigned W, unsigned H) {
    return n_levels >= 1 && n_levels <= PYR_MAX_LEVELS && W % (1u << (n_levels - 1)) == 0 && H % (1u << (n_levels - 1)) == 0;
}

}  // namespace launch
}  // namespace kf
