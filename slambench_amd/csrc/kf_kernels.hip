// kf_kernels.hip — hand-written HIP kernels (gfx950 / CDNA4, wave64) for the KinectFusion hot path.
//
// One kernel per row of SURVEY.md §8(a); the arithmetic of each follows the reference CPP backend
// (kfusion/src/cpp/kernels.cpp, cited as cpp:LINE, and kfusion/include/commons.h) operation by
// operation — the file is compiled with -ffp-contract=off and correctly rounded fp32 divide/sqrt so
// that integer outputs (TSDF, colours, result codes) match the CPU bit for bit and float outputs differ
// only where libm does (expf in the bilateral filter) or where a sum is re-associated (ICP reduction).
//
// Nothing here is a dense contraction: no MFMA.  Every kernel is bound by HBM/L2 traffic or by launch
// latency; the layout choices are about coalescing (64 consecutive x per wave), LDS-staged stencils and
// keeping the pose on the device so the 19-iteration ICP loop never round-trips to the host.
#include "kf_kernels.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

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

// =====================================================================================
// mm2metersKernel (cpp:562-589) and bilateralFilterKernel (cpp:159-198), fused.
// A 32x8 block stages a 36x12 tile of metre depths in LDS (border taps clamp to the edge)
// and each thread walks the 5x5 window from LDS.  Loop order (x offset outer, y offset inner) and
// the running sums follow cpp:179-193; expf is glibc's algorithm restated (expf_glibc), so the result is bit-identical.
// =====================================================================================
// expf as glibc computes it (sysdeps/ieee754/flt-32/e_expf.c, glibc 2.35; the algorithm of ARM's optimized-routines):
// x*32/ln2 = k + r, exp(x) = 2^(k/32) * p(r) evaluated in double and rounded once to float.  The CPU reference calls
// libm's expf 25 times per pixel in the bilateral filter; restating the same algorithm here (table of 2^(i/32), the
// three polynomial coefficients, the shift trick for the rounding to integer) makes the filter bit-identical to the
// CPU instead of within a few ulp.  Checked on the host against glibc for every float in [-87, 0]: 1 118 699 521
// values, one differs (x = -0x1.f8cbb2p+5, result 4e-28, by one ulp).  Valid for x <= 0 (the filter's argument).
__constant__ unsigned long long kExp2fTable[32] = {
0x3ff0000000000000ull,
0x3fefd9b0d3158574ull,
0x3fefb5586cf9890full,
0x3fef9301d0125b51ull,
0x3fef72b83c7d517bull,
0x3fef54873168b9aaull,
0x3fef387a6e756238ull,
0x3fef1e9df51fdee1ull,
0x3fef06fe0a31b715ull,
0x3feef1a7373aa9cbull,
0x3feedea64c123422ull,
0x3feece086061892dull,
0x3feebfdad5362a27ull,
0x3feeb42b569d4f82ull,
0x3feeab07dd485429ull,
0x3feea47eb03a5585ull,
0x3feea09e667f3bcdull,
0x3fee9f75e8ec5f74ull,
0x3feea11473eb0187ull,
0x3feea589994cce13ull,
0x3feeace5422aa0dbull,
0x3feeb737b0cdc5e5ull,
0x3feec49182a3f090ull,
0x3feed503b23e255dull,
0x3feee89f995ad3adull,
0x3feeff76f2fb5e47ull,
0x3fef199bdd85529cull,
0x3fef3720dcef9069ull,
0x3fef5818dcfba487ull,
0x3fef7c97337b9b5full,
0x3fefa4afa2a490daull,
0x3fefd0765b6e4540ull
};

__device__ __forceinline__ float expf_glibc(float x, const double *tab /* LDS copy of kExp2fTable as doubles' bits */) {
    if (x < -0x1.9fe368p6f) return 0.0f;                 // below log(2^-150): underflows to zero
    const double InvLn2N = 0x1.71547652b82fep+0 * 32, Shift = 0x1.8p52;
    const double C0 = 0x1.c6af84b912394p-5 / 32 / 32 / 32, C1 = 0x1.ebfce50fac4f3p-3 / 32 / 32, C2 = 0x1.62e42ff0c52d6p-1 / 32;
    const double xd = (double)x;
    const double z = InvLn2N * xd;
    double kd = z + Shift;                                // rounds z to the nearest integer in the low mantissa bits
    const unsigned long long ki = __builtin_bit_cast(unsigned long long, kd);
    kd -= Shift;
    const double r = z - kd;
    unsigned long long t = __builtin_bit_cast(unsigned long long, tab[ki & 31]);
    t += ki << (52 - 5);
    const double s = __builtin_bit_cast(double, t);
    const double zz = C0 * r + C1;
    const double r2 = r * r;
    double y = C2 * r + 1.0;
    y = zz * r2 + y;
    y = y * s;
    return (float)y;
}

// a / d for a launch-wide constant d with r = RN(1 / d): the product plus one residual correction.  Exact (= the IEEE quotient)
// for the divisors and numerator ranges k_verify_div has checked on this device - see there.
__device__ __forceinline__ float div_by_const(float a, float d, float r) {
    const float q = a * r;
    return __builtin_fmaf(__builtin_fmaf(-d, q, a), r, q);
}

constexpr int PBX = 32, PBY = 8, PR = 2;

// One 32x8 output tile.  `tid` is the thread's index inside its 256-thread tile team, `active` is false for a team
// that only keeps the barrier company (side-job rounds of the fused ICP launch, see k_icp_iter_bilateral).
// FD: the range term's division by 2 e_d^2 (cpp:184) in the reciprocal form, for a divisor k_verify_div has passed (numerators
// below 2^24; a wave that meets a larger one - or a NaN - takes the IEEE division, all lanes together)
template <bool FROM_MM, bool FD = false>
__device__ __forceinline__ void bilateral_block(float (*tile)[PBX + 2 * PR + 1], const double *exp_tab, int bx, int by, int tid,
                                                bool active, float *__restrict__ floatDepth, float *__restrict__ out,
                                                const void *__restrict__ inv, unsigned W, unsigned H, unsigned inW, int ratio,
                                                const Gauss5 &g, float e_d_squared_2, float rcp_e_d_squared_2 = 0.f) {
    const int x0 = bx * PBX, y0 = by * PBY;
    constexpr int TW = PBX + 2 * PR, TH = PBY + 2 * PR;
    if (active) {
        // every load of the tile is issued before the first is waited for (a cold miss takes ~1 us, and a loop of
        // load - store - load would pay it once per round)
        constexpr int ROUNDS = (TW * TH + PBX * PBY - 1) / (PBX * PBY);
        float v[ROUNDS];
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int idx = tid + r * PBX * PBY;
            if (idx < TW * TH) {
                const int ty = idx / TW, tx = idx - ty * TW;
                const int gx = imax(0, imin(x0 + tx - PR, (int)W - 1));
                const int gy = imax(0, imin(y0 + ty - PR, (int)H - 1));
                if (FROM_MM) v[r] = static_cast<const uint16_t *>(inv)[gx * ratio + inW * gy * ratio] / 1000.0f;
                else v[r] = static_cast<const float *>(inv)[gx + gy * W];
            }
        }
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int idx = tid + r * PBX * PBY;
            if (idx < TW * TH) { const int ty = idx / TW; tile[ty][idx - ty * TW] = v[r]; }
        }
    }
    __syncthreads();
    const int lx = tid & (PBX - 1), ly = tid / PBX;
    const int x = x0 + lx, y = y0 + ly;
    if (!active || x >= (int)W || y >= (int)H) return;
    const unsigned pos = x + y * W;
    const float center = tile[ly + PR][lx + PR];
    if (FROM_MM && floatDepth) floatDepth[pos] = center;
    if (center == 0) { out[pos] = 0; return; }
    float sum = 0.0f, t = 0.0f;
#pragma unroll
    for (int i = -PR; i <= PR; ++i) {
#pragma unroll
        for (int j = -PR; j <= PR; ++j) {
            const float cur = tile[ly + PR + j][lx + PR + i];
            if (cur > 0) {
                const float mod = sq(cur - center);
                float e;
                if (i == 0 && j == 0 && __builtin_amdgcn_ballot_w64(!(mod == 0.f)) == 0ull) {
                    e = 1.0f;      // the centre tap: mod = +0 (unless the depth is infinite), the argument -0, and expf(-0) is exactly 1
                } else {
                    float arg;
                    if (FD && __builtin_amdgcn_ballot_w64(!(mod < 0x1p24f)) == 0ull) arg = -div_by_const(mod, e_d_squared_2, rcp_e_d_squared_2);
                    else arg = -mod / e_d_squared_2;
                    e = expf_glibc(arg, exp_tab);
                }
                const float factor = g.g[i + PR] * g.g[j + PR] * e;
                t += factor * cur;
                sum += factor;
            }
        }
    }
    out[pos] = t / sum;
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

// =====================================================================================
// depth2vertexKernel (cpp:200-218) and vertex2normalKernel (cpp:220-249)
// =====================================================================================
__device__ __forceinline__ f3 back_project(float d, int x, int y, const M4 &invK) {
    return d > 0 ? scale3(rotate(invK, mk3((float)x, (float)y, 1.f)), d) : mk3(0.f, 0.f, 0.f);
}

__global__ __launch_bounds__(256) void k_depth2vertex(float4 *__restrict__ vertex, const float *__restrict__ depth,
                                                      unsigned W, unsigned H, M4 invK) {
    const unsigned x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= W || y >= H) return;
    const f3 v = back_project(depth[x + y * W], x, y, invK);
    vertex[x + y * W] = make_float4(v.x, v.y, v.z, 0.f);
}

// Invalid pixels write only .x: .y/.z keep whatever the previous frame left there.
// `stale`: where the previous frame's normal of this pixel lives when the maps are double-buffered (nullptr = in `out`).
__device__ __forceinline__ void store_normal(float4 *out, f3 left, f3 right, f3 up, f3 down, const float4 *stale = nullptr) {
    if (left.z == 0 || right.z == 0 || up.z == 0 || down.z == 0) {
        if (stale) {
            const float4 o = *stale;
            *out = make_float4(kInvalid, o.y, o.z, o.w);
        } else {
            reinterpret_cast<float *>(out)[0] = kInvalid;
        }
        return;
    }
    const f3 dxv = sub3(right, left), dyv = sub3(down, up);
    const f3 n = normalize3(cross3(dyv, dxv));
    *out = make_float4(n.x, n.y, n.z, 0.f);
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

// =====================================================================================
// Whole input pyramid in one launch: halfSampleRobustImage for levels 1..L-1 (cpp:591-626) and depth2vertex +
// vertex2normal for every level (cpp:200-249), i.e. everything Kfusion::tracking does before the ICP loop
// (cpp:931-945).  A block owns a 32x32 tile of level 0 and the matching 16x16 / 8x8 / 4x4 tiles of the coarser levels.
// The filtered depth tile is staged in LDS with a halo wide enough (2^(L-1) pixels) that every coarser depth the
// block needs - its own tile plus the 1-pixel ring its normals read - can be derived locally, level by level, from
// LDS; each level's depth tile is written out once and its vertices / normals are computed from the LDS copy with
// the expressions of the stand-alone kernels, so all maps carry the same bits.  Five dependent launches become one.
// =====================================================================================
constexpr int PYR_T = 32;   // level-0 tile edge
constexpr int PYR_MAX_LEVELS = 4;

struct PyramidArgs {
    float *scaled[PYR_MAX_LEVELS];      // scaled[0] is the input (bilateral-filtered depth), the others are outputs
    float4 *vertex[PYR_MAX_LEVELS];
    float4 *normal[PYR_MAX_LEVELS];
    const float4 *stale_normal[PYR_MAX_LEVELS];   // previous frame's normal maps when double-buffered, else nullptr
    M4 invK[PYR_MAX_LEVELS];
    unsigned W, H;
    float e_d;
    bool through;       // the maps' stores go through L2 as they are issued (store_through)
};
constexpr int PYR_THREADS = 256;

// halo of level l (in level-l pixels): 1 on the coarsest level, doubling towards level 0
constexpr int pyr_halo(int nlev, int l) { int h = 1; for (int i = nlev - 1; i > l; --i) h *= 2; return h; }
constexpr int pyr_stride(int nlev, int l) { return (PYR_T >> l) + 2 * pyr_halo(nlev, l); }
constexpr int pyr_count(int nlev, int l) { return pyr_stride(nlev, l) * pyr_stride(nlev, l); }
constexpr int pyr_offset(int nlev, int l) { int o = 0; for (int i = 0; i < l; ++i) o += pyr_count(nlev, i); return o; }

template <int NLEV, int L>
struct PyrGeom {
    static constexpr int halo = pyr_halo(NLEV, L);
    static constexpr int tile = PYR_T >> L;
    static constexpr int stride = pyr_stride(NLEV, L);
    static constexpr int count = pyr_count(NLEV, L);
    static constexpr int offset = pyr_offset(NLEV, L);
};

// A 16-byte store that, if `through`, goes through the XCD's L2 to memory as it is issued (`sc0 sc1`) instead of staying
// there dirty until the launch's closing write-back.  Worth it where a side block stores one
// tile's maps and ends (640x480: the launch that carries the upper levels 8.65 -> 8.15 us); where a block works through nine
// rounds of tiles (1280x960) the stores of one round drain while the next computes, and going through costs +2.3 us a launch
// (profiles/r04_ab_pyr_wt.txt).
typedef float v4f_t __attribute__((ext_vector_type(4)));
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_through(bool through, float4 *base, size_t idx, float x, float y, float z, float w) {
    // (a raw buffer store with the sc0 sc1 bits - aux 0x11 on gfx950 - not a volatile access, which the compiler would wait for
    // before issuing the next, and not inline assembly, which its hazard recogniser and its count of outstanding memory
    // operations cannot see: the map's base is wave-uniform, the pixel's byte offset the per-lane part; maps are < 4 GiB)
    if (through) {
        const v4f_t v{x, y, z, w};
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)0xFFFFFFFFu, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u_t, v), rsrc, (int)((unsigned)idx * 16u), 0, 0x11);
    } else base[idx] = make_float4(x, y, z, w);
}

template <int NLEV, int L, int NTHR>
__device__ __forceinline__ void pyramid_level(const PyramidArgs &a, float *lds, int bx, int by, int tid) {
    using G = PyrGeom<NLEV, L>;
    const unsigned Wl = a.W >> L, Hl = a.H >> L;
    const int X0 = (bx * PYR_T) >> L, Y0 = (by * PYR_T) >> L;       // this block's tile origin at level L
    float *d = lds + G::offset;
    if constexpr (L > 0) {
        // depth of level L over the tile plus its halo, from level L-1 in LDS (robust 2x2 mean, cpp:598-623)
        using P = PyrGeom<NLEV, L - 1>;
        const float *dp = lds + P::offset;
        const int PX0 = ((bx * PYR_T) >> (L - 1)) - P::halo, PY0 = ((by * PYR_T) >> (L - 1)) - P::halo;
        for (int idx = tid; idx < G::count; idx += NTHR) {
            const int ly = idx / G::stride, lx = idx - ly * G::stride;
            const int gx = X0 - G::halo + lx, gy = Y0 - G::halo + ly;
            if (gx < 0 || gy < 0 || gx >= (int)Wl || gy >= (int)Hl) continue;
            const float *r0 = dp + (2 * gy - PY0) * P::stride + (2 * gx - PX0);
            const float *r1 = r0 + P::stride;
            const float v00 = r0[0], v01 = r0[1], v10 = r1[0], v11 = r1[1];
            const float center = v00;
            float sum = 0.0f, t = 0.0f;
            if (fabsf(v00 - center) < a.e_d) { sum += 1.0f; t += v00; }
            if (fabsf(v01 - center) < a.e_d) { sum += 1.0f; t += v01; }
            if (fabsf(v10 - center) < a.e_d) { sum += 1.0f; t += v10; }
            if (fabsf(v11 - center) < a.e_d) { sum += 1.0f; t += v11; }
            const float r = t / sum;
            d[idx] = r;
            if (lx >= G::halo && lx < G::halo + G::tile && ly >= G::halo && ly < G::halo + G::tile)
                a.scaled[L][gx + (size_t)gy * Wl] = r;      // the tile itself is this block's to publish
        }
        __syncthreads();
    }
    // vertex and normal maps of the tile (cpp:206-215, 225-247).  Three passes over the thread's pixels (four on level 0, one or
    // none above it) so that the loads of the stale normals - a cold miss each, wherever a wave holds an invalid pixel - are all in
    // flight together instead of one per round; the arithmetic is store_normal's.
    const M4 invK = a.invK[L];
    constexpr int N = G::tile * G::tile, ROUNDS = (N + NTHR - 1) / NTHR;
    f3 vc[ROUNDS], vn[ROUNDS];
    size_t where[ROUNDS];
    bool on[ROUNDS], bad[ROUNDS];
    const float4 *stale = a.stale_normal[L];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int idx = tid + r * NTHR;
        const int ty = idx / G::tile, tx = idx - ty * G::tile;
        const int x = X0 + tx, y = Y0 + ty;
        on[r] = idx < N && x < (int)Wl && y < (int)Hl;
        bad[r] = false;
        if (!on[r]) continue;
        where[r] = x + (size_t)y * Wl;
        const int xl = imax(x - 1, 0), xr = imin(x + 1, (int)Wl - 1), yu = imax(y - 1, 0), yd = imin(y + 1, (int)Hl - 1);
        const int ox = X0 - G::halo, oy = Y0 - G::halo;
        vc[r] = back_project(d[(y - oy) * G::stride + (x - ox)], x, y, invK);
        const f3 l = back_project(d[(y - oy) * G::stride + (xl - ox)], xl, y, invK);
        const f3 rt = back_project(d[(y - oy) * G::stride + (xr - ox)], xr, y, invK);
        const f3 u = back_project(d[(yu - oy) * G::stride + (x - ox)], x, yu, invK);
        const f3 dn = back_project(d[(yd - oy) * G::stride + (x - ox)], x, yd, invK);
        bad[r] = l.z == 0 || rt.z == 0 || u.z == 0 || dn.z == 0;
        if (!bad[r]) vn[r] = normalize3(cross3(sub3(dn, u), sub3(rt, l)));
    }
    float4 old[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r)
        if (on[r] && bad[r] && stale) old[r] = stale[where[r]];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        if (!on[r]) continue;
        store_through(a.through, a.vertex[L], where[r], vc[r].x, vc[r].y, vc[r].z, 0.f);
        if (!bad[r]) store_through(a.through, a.normal[L], where[r], vn[r].x, vn[r].y, vn[r].z, 0.f);
        else if (stale) store_through(a.through, a.normal[L], where[r], kInvalid, old[r].y, old[r].z, old[r].w);      // SURVEY N3, as store_normal
        else reinterpret_cast<float *>(a.normal[L] + where[r])[0] = kInvalid;
    }
}

// One 32x32 level-0 tile by the NTHR threads of a block.  UPPER_ONLY: the maps of level 0 are somebody else's (a
// pyramid_block<1> over the same tile, in another launch); the level-0 depth is still staged, the levels above come from it.
template <int NLEV, int NTHR, bool UPPER_ONLY = false>
__device__ __forceinline__ void pyramid_block(const PyramidArgs &a, float *lds, int bx, int by, int tid) {
    using G0 = PyrGeom<NLEV, 0>;
    // level-0 tile with halo; taps outside the image take the nearest pixel (only the 1-pixel ring is ever read there,
    // by the clamped neighbour accesses of the normals)
    const int X0 = bx * PYR_T - G0::halo, Y0 = by * PYR_T - G0::halo;
    constexpr int ROUNDS = (G0::count + NTHR - 1) / NTHR;      // all loads in flight before the first LDS store (see bilateral_block)
    float v[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int idx = tid + r * NTHR;
        if (idx < G0::count) {
            const int ly = idx / G0::stride, lx = idx - ly * G0::stride;
            const int gx = imax(0, imin(X0 + lx, (int)a.W - 1)), gy = imax(0, imin(Y0 + ly, (int)a.H - 1));
            v[r] = a.scaled[0][gx + (size_t)gy * a.W];
        }
    }
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r)
        if (tid + r * NTHR < G0::count) lds[tid + r * NTHR] = v[r];
    __syncthreads();
    if constexpr (!UPPER_ONLY) pyramid_level<NLEV, 0, NTHR>(a, lds, bx, by, tid);
    if constexpr (NLEV > 1) pyramid_level<NLEV, 1, NTHR>(a, lds, bx, by, tid);
    if constexpr (NLEV > 2) pyramid_level<NLEV, 2, NTHR>(a, lds, bx, by, tid);
    if constexpr (NLEV > 3) pyramid_level<NLEV, 3, NTHR>(a, lds, bx, by, tid);
}

template <int NLEV>
constexpr int pyr_lds_floats() { return PyrGeom<NLEV, NLEV - 1>::offset + PyrGeom<NLEV, NLEV - 1>::count; }

template <int NLEV>
__global__ __launch_bounds__(PYR_THREADS) void k_pyramid(PyramidArgs a) {
    __shared__ float lds[pyr_lds_floats<NLEV>()];
    pyramid_block<NLEV, PYR_THREADS>(a, lds, blockIdx.x, blockIdx.y, threadIdx.x);
}

// =====================================================================================
// trackKernel (cpp:497-560) + reduceKernel (cpp:251-495), fused.
// Each thread associates its pixels, accumulates the 32 sums of the reference layout
//   [0] sum e^2  [1..6] sum e*J  [7..27] upper triangle of J^T J (row-major)
//   [28] inliers [29] #(-4) [30] #(-5) [31] #(-1..-3)
// in registers, then a wave64 shuffle tree and an LDS step produce one 32-float partial per block.
// Partials are summed by k_solve in block order: no atomics, so the result is bitwise reproducible
// run to run (the reference's OpenMP build is not).  The sum is re-associated with
// respect to the serial CPU order, which is the one float difference this path has by design.
// =====================================================================================
constexpr int TR_THREADS = 256;
constexpr unsigned TR_MAX_BLOCKS = 2048;

struct PixelTrack { int result; float error; float J[6]; };

__device__ __forceinline__ PixelTrack track_fetched(float4 n4, float4 v4, const float4 *__restrict__ refV,
                                                    const float4 *__restrict__ refN, unsigned refW, unsigned refH, const M4 &T,
                                                    const M4 &view, int have_reference, float dist_threshold,
                                                    float normal_threshold) {
    PixelTrack r;
    r.error = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) r.J[i] = 0.f;
    if (n4.x == kInvalid) { r.result = -1; return r; }
    const f3 pv = xform(T, mk3(v4.x, v4.y, v4.z));
    unsigned rx = 0, ry = 0;
    if (have_reference) {
        const f3 pp = xform(view, pv);
        const float u = pp.x / pp.z + 0.5f, w = pp.y / pp.z + 0.5f;
        if (u < 0 || u > refW - 1 || w < 0 || w > refH - 1) { r.result = -2; return r; }
        rx = (unsigned)u; ry = (unsigned)w;
    }
    const float4 rn4 = refN[rx + ry * refW];
    const float4 rv4 = refV[rx + ry * refW];   // same pixel as the normal: gather both at once
    if (rn4.x == kInvalid) { r.result = -3; return r; }
    const f3 n_ref = mk3(rn4.x, rn4.y, rn4.z);
    const f3 diff = sub3(mk3(rv4.x, rv4.y, rv4.z), pv);
    const f3 pn = rotate(T, mk3(n4.x, n4.y, n4.z));
    if (length3(diff) > dist_threshold) { r.result = -4; return r; }
    if (dot3(pn, n_ref) < normal_threshold) { r.result = -5; return r; }
    r.result = 1;
    r.error = dot3(n_ref, diff);
    r.J[0] = n_ref.x; r.J[1] = n_ref.y; r.J[2] = n_ref.z;
    const f3 c = cross3(pv, n_ref);
    r.J[3] = c.x; r.J[4] = c.y; r.J[5] = c.z;
    return r;
}

__device__ __forceinline__ PixelTrack track_pixel(unsigned px, unsigned py, const float4 *__restrict__ inV,
                                                  const float4 *__restrict__ inN, unsigned inW,
                                                  const float4 *__restrict__ refV, const float4 *__restrict__ refN,
                                                  unsigned refW, unsigned refH, const M4 &T, const M4 &view,
                                                  int have_reference, float dist_threshold, float normal_threshold) {
    // both input maps are fetched before the first test so the two loads are in flight together
    const float4 n4 = inN[px + py * inW];
    const float4 v4 = inV[px + py * inW];
    return track_fetched(n4, v4, refV, refN, refW, refH, T, view, have_reference, dist_threshold, normal_threshold);
}

__device__ __forceinline__ void accumulate(float (&acc)[32], int result, float e, const float (&J)[6]) {
    if (result < 1) {
        acc[29] += result == -4 ? 1.f : 0.f;
        acc[30] += result == -5 ? 1.f : 0.f;
        acc[31] += result > -4 ? 1.f : 0.f;
        return;
    }
    acc[0] += e * e;
#pragma unroll
    for (int i = 0; i < 6; ++i) acc[1 + i] += e * J[i];
    int k = 7;
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = r; c < 6; ++c) acc[k++] += J[r] * J[c];
    acc[28] += 1.f;
}

// 256 threads -> one 32-float partial.  Fixed tree, so the result is bitwise reproducible.
// Wave step: a halving butterfly.  At each of five steps a lane keeps one half of its remaining values
// and hands the other half to its partner (lane ^ 32, 16, 8, 4, 2), so 32 values x 64 lanes collapse to
// one value per lane with 16+8+4+2+1 shuffles instead of 32x6; a last exchange with lane ^ 1 completes
// the sum.  Lane l ends up owning sum index ((l>>5)&1)*16 + ((l>>4)&1)*8 + ((l>>3)&1)*4 + ((l>>2)&1)*2 + ((l>>1)&1).
template <int N>
__device__ __forceinline__ void halve(const float (&in)[2 * N], float (&out)[N], bool hi, int partner) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const float keep = hi ? in[i + N] : in[i];
        const float give = hi ? in[i] : in[i + N];
        out[i] = keep + __shfl_xor(give, partner, 64);
    }
}
// The same step for the partners lane ^ 32 and lane ^ 16 without the two selects and the LDS crossbar: gfx950's
// v_permlane32_swap exchanges the upper half of one register with the lower half of another (v_permlane16_swap: the odd rows
// of 16 lanes with the even rows), after which "what I keep" and "what my partner gave" sit in the two registers of every
// lane.  Same pairs, same sums, same bits as halve().  (Inline assembly: this compiler's __builtin_amdgcn_permlane32_swap
// hands back its first result twice; the s_nops stand for the wait states the compiler puts around the instruction when it
// schedules it itself - a VALU result feeding a swap, a swap feeding a swap - which it cannot see inside an asm block.)
#ifndef KFH_REDUCE_SWAP
#define KFH_REDUCE_SWAP 1
#endif
#if KFH_REDUCE_SWAP && defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "halve_swap: v_permlane32_swap / v_permlane16_swap and the wait states written around them are gfx950's (build with -DKFH_REDUCE_SWAP=0 elsewhere)"
#endif
template <int N, int PARTNER>
__device__ __forceinline__ void halve_swap(const float (&in)[2 * N], float (&out)[N]) {
    static_assert(PARTNER == 32 || PARTNER == 16, "row swaps");
#pragma unroll
    for (int i = 0; i < N; ++i) {
        float a = in[i], b = in[i + N];
        if (PARTNER == 32) asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
        else asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
        out[i] = a + b;
    }
}
// ... and for lane ^ 8, ^ 2, ^ 1, which stay inside a row of 16 lanes: a DPP operand of the addition (row rotate by 8, quad permutes)
template <int N, int CTRL>
__device__ __forceinline__ void halve_dpp(const float (&in)[2 * N], float (&out)[N], bool hi) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const float keep = hi ? in[i + N] : in[i];
        const float give = hi ? in[i] : in[i + N];
        out[i] = keep + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, give), CTRL, 0xF, 0xF, true));
    }
}

template <int NT>
__device__ __forceinline__ void block_reduce_store(float (&acc)[32], float *__restrict__ partial) {
    __shared__ float sm[NT / 64][32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float a16[16], a8[8], a4[4], a2[2], a1[1];
#if KFH_REDUCE_SWAP
    halve_swap<16, 32>(acc, a16);
    halve_swap<8, 16>(a16, a8);
    halve_dpp<4, 0x128>(a8, a4, lane & 8);          // row_ror:8
    halve<2>(a4, a2, lane & 4, 4);
    halve_dpp<1, 0x4E>(a2, a1, lane & 2);           // quad_perm [2, 3, 0, 1]
    const float v = a1[0] + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a1[0]), 0xB1, 0xF, 0xF, true));   // quad_perm [1, 0, 3, 2]
#else
    halve<16>(acc, a16, lane & 32, 32);
    halve<8>(a16, a8, lane & 16, 16);
    halve<4>(a8, a4, lane & 8, 8);
    halve<2>(a4, a2, lane & 4, 4);
    halve<1>(a2, a1, lane & 2, 2);
    const float v = a1[0] + __shfl_xor(a1[0], 1, 64);
#endif
    const int index = ((lane >> 5) & 1) * 16 + ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
    if ((lane & 1) == 0) sm[wave][index] = v;
    __syncthreads();
    if (threadIdx.x < 32) {
        float t = sm[0][threadIdx.x];
#pragma unroll
        for (int w = 1; w < NT / 64; ++w) t += sm[w][threadIdx.x];
        partial[threadIdx.x] = t;
    }
}

// Self-check of block_reduce_store (kfh_create runs it once per device): the row swaps and DPP adds above sit in inline assembly
// whose wait states the compiler's hazard recogniser cannot see, and a wrong hazard would give wrong sums, not a fault.  Every lane
// of a block gets 32 distinct pseudo-random values; the block's 32 sums must equal, BIT FOR BIT, the same fixed tree
// evaluated with plain shuffles (halve<N> five times, lane ^ 1, then the waves in order).  Run for both instantiations the
// backend launches: ICP_THREADS (the fused ICP kernels) and TR_THREADS (k_track_reduce / k_reduce_trackdata).
template <int NT>
__global__ __launch_bounds__(NT) void k_verify_reduce(unsigned *mismatches, unsigned seed) {
    __shared__ float ref_sm[NT / 64][32];
    __shared__ float got[32];
    float acc[32], a16[16], a8[8], a4[4], a2[2], a1[1];
    unsigned x = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        x = x * 1664525u + 1013904223u;
        acc[i] = (float)(int)(x >> 8) * (1.0f / 8388608.0f) - 1.0f;          // [-1, 1), 24 significant bits: the additions round
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    halve<16>(acc, a16, lane & 32, 32);
    halve<8>(a16, a8, lane & 16, 16);
    halve<4>(a8, a4, lane & 8, 8);
    halve<2>(a4, a2, lane & 4, 4);
    halve<1>(a2, a1, lane & 2, 2);
    const float v = a1[0] + __shfl_xor(a1[0], 1, 64);
    const int index = ((lane >> 5) & 1) * 16 + ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
    if ((lane & 1) == 0) ref_sm[wave][index] = v;
    block_reduce_store<NT>(acc, got);            // (ends with the 32 sums in `got`; its own barrier orders ref_sm as well)
    __syncthreads();
    if (threadIdx.x < 32) {
        float t = ref_sm[0][threadIdx.x];
#pragma unroll
        for (int w = 1; w < NT / 64; ++w) t += ref_sm[w][threadIdx.x];
        if (__builtin_bit_cast(unsigned, t) != __builtin_bit_cast(unsigned, got[threadIdx.x])) atomicAdd(mismatches, 1u);
    }
}

__global__ __launch_bounds__(TR_THREADS) void k_track_reduce(
    float *__restrict__ partials, int *__restrict__ result_map, kfh_track_data *__restrict__ track_out,
    const float4 *__restrict__ inV, const float4 *__restrict__ inN, unsigned inW, unsigned inH,
    const float4 *__restrict__ refV, const float4 *__restrict__ refN, unsigned refW, unsigned refH,
    const M4 *__restrict__ posep, const TrackParams *__restrict__ tp, const int *__restrict__ skip,
    float dist_threshold, float normal_threshold) {
    if (skip && *skip) return;
    const M4 T = *posep;
    const M4 view = tp->view;
    const int have_reference = tp->have_reference;
    float acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = 0.f;
    const unsigned n = inW * inH;
    for (unsigned p = blockIdx.x * TR_THREADS + threadIdx.x; p < n; p += gridDim.x * TR_THREADS) {
        const unsigned py = p / inW, px = p - py * inW;
        const PixelTrack r = track_pixel(px, py, inV, inN, inW, refV, refN, refW, refH, T, view, have_reference,
                                         dist_threshold, normal_threshold);
        // output rows keep the full-resolution stride on every level (cpp:512)
        if (result_map) result_map[px + py * refW] = r.result;
        if (track_out) {
            kfh_track_data *o = track_out + px + py * refW;
            o->result = r.result;
            if (r.result == 1) {   // error / J are only defined for inliers
                o->error = r.error;
#pragma unroll
                for (int i = 0; i < 6; ++i) o->J[i] = r.J[i];
            }
        }
        accumulate(acc, r.result, r.error, r.J);
    }
    if (partials) block_reduce_store<TR_THREADS>(acc, partials + (size_t)blockIdx.x * 32);
}

// reduceKernel on an existing TrackData map (kernel-level API; the pipeline uses the fused kernel)
__global__ __launch_bounds__(TR_THREADS) void k_reduce_trackdata(float *__restrict__ partials,
                                                                 const kfh_track_data *__restrict__ J, unsigned JW,
                                                                 unsigned W, unsigned H) {
    float acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = 0.f;
    const unsigned n = W * H;
    for (unsigned p = blockIdx.x * TR_THREADS + threadIdx.x; p < n; p += gridDim.x * TR_THREADS) {
        const unsigned py = p / W, px = p - py * W;
        const kfh_track_data row = J[px + py * JW];
        accumulate(acc, row.result, row.error, row.J);
    }
    block_reduce_store<TR_THREADS>(acc, partials + (size_t)blockIdx.x * 32);
}

// =====================================================================================
// Partial sums -> reduction row 0, then updatePoseKernel (cpp:759-775) on the device:
//   x = pinv(J^T J) J^T e (double), pose <- exp(x) * pose (float), level done when |x| < icp_threshold.
// One block; the pose never leaves HBM between ICP iterations.
// =====================================================================================
// `ws`: 72 doubles of LDS for the Jacobi fallback (see solve6_ws)
__device__ __forceinline__ bool apply_update(M4 &pose, const float *red, float icp_threshold, bool &fast, double *ws,
                                             unsigned long long *tr = nullptr) {
    double x[6];
    KF_SOLVE_STAMP(tr, 8, red[1]);
    fast = solve6_ldlt(red + 1, x, tr);
    if (!fast) solve6_ws(red + 1, x, reinterpret_cast<double (*)[6]>(ws), reinterpret_cast<double (*)[6]>(ws + 36));
    KF_SOLVE_STAMP(tr, 11, (float)(x[0] + x[5]));
    const M4 delta = se3_exp_to_m4(x);       // (last row 0, 0, 0, 1 by construction)
    KF_SOLVE_STAMP(tr, 12, delta.m[0][0] + delta.m[2][3] + delta.m[1][2]);
#if KFH_SOLVE_LEAN
    pose = last_row_is_rigid(pose) ? matmul_rigid(delta, pose) : matmul(delta, pose);
#else
    pose = matmul(delta, pose);
#endif
    KF_SOLVE_STAMP(tr, 13, pose.m[2][3] + pose.m[0][0] + pose.m[1][1]);
    double n2 = 0.0;
    for (int i = 0; i < 6; ++i) n2 += x[i] * x[i];
    KF_SOLVE_STAMP(tr, 14, (float)n2);
    // norm(x) < icp_threshold (cpp:772), compared as squares: no square root on this serial path
    return icp_threshold > 0.f && n2 < (double)icp_threshold * (double)icp_threshold;
}

__global__ __launch_bounds__(1024) void k_solve(DevState *__restrict__ st, float *__restrict__ out32,
                                                const float *__restrict__ partials, unsigned nblocks, int level,
                                                float icp_threshold, int do_solve) {
    if (st && do_solve && st->level_done[level]) return;
    __shared__ float sm[32][33];
    __shared__ float red[32];
    __shared__ double jacobi_ws[72];
    const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
    float acc = 0.f;
    for (unsigned b = g; b < nblocks; b += 32) acc += partials[(size_t)b * 32 + c];
    sm[g][c] = acc;
    __syncthreads();
    if (threadIdx.x < 32) {
        float v = sm[0][threadIdx.x];
#pragma unroll
        for (int k = 1; k < 32; ++k) v += sm[k][threadIdx.x];
        red[threadIdx.x] = v;
        if (out32) out32[threadIdx.x] = v;
        if (st) st->reduction[threadIdx.x] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0 && st && do_solve) {
        M4 pose = st->pose;
        bool fast = true;
        const bool converged = apply_update(pose, red, icp_threshold, fast, jacobi_ws);
        st->pose = pose;
        st->icp_iterations += 1;
        if (!fast) st->solve_fallbacks += 1;
        if (converged) st->level_done[level] = 1;
    }
}

__global__ void k_update_pose(M4 *pose, const float *reduction32, float icp_threshold, int *converged) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    __shared__ double jacobi_ws[72];
    M4 p = *pose;
    float red[32];
    for (int i = 0; i < 32; ++i) red[i] = reduction32[i];
    bool fast;
    const bool c = apply_update(p, red, icp_threshold, fast, jacobi_ws);
    *pose = p;
    *converged = c ? 1 : 0;
}

// checkPoseKernel (cpp:777-792): the verdict; the callers replace a rejected pose by the one tracking started from
__device__ __forceinline__ bool pose_is_good(const float *red, unsigned W, unsigned H, float track_threshold) {
    return !(((double)sqrtf(red[0] / red[28]) > 2e-2) || (red[28] / (W * H) < track_threshold));
}

__global__ void k_check_pose(M4 *pose, const M4 *oldPose, const float *reduction32, unsigned W, unsigned H, float thr,
                             int *ok) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const bool good = pose_is_good(reduction32, W, H, thr);
    *ok = good ? 1 : 0;
    if (!good) *pose = *oldPose;
}

// =====================================================================================
// Control kernel: the scalar glue of Kfusion::tracking / integration / raycasting (cpp:924-1004) run by one
// thread on the device, so frame gating and pose algebra need no host round trip.
// =====================================================================================
__device__ __forceinline__ void track_begin(DevState *__restrict__ st, bool track_frame, const float k[4]) {
    for (int l = 0; l < KFH_MAX_LEVELS; ++l) st->level_done[l] = track_frame ? 0 : 1;
    // arrival counters of k_icp_tail's grid barrier: cleared here, by the one thread of a launch that precedes the frame's ICP loop
    for (int l = 0; l < KFH_MAX_LEVELS; ++l) st->tail_barrier[l] = 0;
    st->timeouts_at_track_begin = __hip_atomic_load(&st->barrier_timeouts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (track_frame) {
        st->oldPose = st->pose;                                  // cpp:947
        st->track.have_reference = st->have_reference;
        if (st->have_reference)                                  // cpp:948
            st->track.view = matmul(camera_matrix(k[0], k[1], k[2], k[3]), inverse(st->raycastPose));
    }
}

__device__ void control_ops(DevState *__restrict__ st, FrameLogEntry *__restrict__ log, int ops, const ControlArgs &a) {
    const bool track_frame = (a.frame % a.tracking_rate) == 0;   // cpp:927
    if (ops & OP_TRACK_BEGIN) track_begin(st, track_frame, a.k);
    if (ops & OP_TRACK_END) {
        if (track_frame) {                                           // cpp:968-969
            // (the pose is copied in memory only when it is rejected: a local M4 assigned on two paths is kept in an alloca,
            // which the compiler then promotes to 64 KB of LDS - 16 floats for each of a block's 1024 possible threads)
            bool ok = pose_is_good(st->reduction, a.width, a.height, a.track_threshold);
            // a workgroup of k_icp_tail gave up at the grid barrier during this frame's loop: the sums it went on with were
            // incomplete, so the frame counts as not tracked - old pose back, and the reduction row is marked (no inliers) so
            // that the integration gate (cpp:991) fails the same way - and the host is told (DevState::failed_frames)
            const unsigned now = __hip_atomic_load(&st->barrier_timeouts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (now != st->timeouts_at_track_begin) {
                st->timeouts_at_track_begin = now;
                st->failed_frames += 1;
                st->reduction[28] = 0.f;
                ok = false;
            }
            if (!ok) st->pose = st->oldPose;
            st->tracked = ok ? 1 : 0;
        } else {
            st->tracked = 0;
        }
    }
    if (ops & OP_INTEG_BEGIN) {                                      // cpp:991-1000
        const bool ok = pose_is_good(st->reduction, a.width, a.height, a.track_threshold);
        if (!ok) st->pose = st->oldPose;
        const M4 p = st->pose;
        const bool go = (ok && (a.frame % a.integration_rate) == 0) || (a.frame <= 3);
        st->integ.enabled = go ? 1 : 0;
        st->integrated = go ? 1 : 0;
        if (go) {
            st->integ.invTrack = inverse(p);
            st->integ.K = camera_matrix(a.k[0], a.k[1], a.k[2], a.k[3]);
            integrate_bounds(p, a.k, mk3(a.vol_dim[0], a.vol_dim[1], a.vol_dim[2]), &st->integ.cam_bound, &st->integ.far_bound);
        }
    }
    if (ops & OP_RAY_BEGIN) {                                        // cpp:977-982
        if (a.frame > 2) {
            st->raycastPose = st->pose;
            st->have_reference = 1;
            st->ray.view = matmul(st->raycastPose, inverse_camera_matrix(a.k[0], a.k[1], a.k[2], a.k[3]));
            st->ray.enabled = 1;
        } else {
            st->ray.enabled = 0;
        }
    }
    if (ops & OP_FRAME_END) {
        if (log && a.log_capacity) {
            FrameLogEntry &e = log[a.frame % a.log_capacity];
            e.pose = st->pose; e.tracked = st->tracked; e.integrated = st->integrated; e.frame = (int)a.frame; e.pad = 0;
        }
    }
    if (ops & OP_RENDER_BEGIN) {                                     // cpp:1035-1036
        const M4 vp = a.use_view_pose ? a.view_pose : st->pose;
        st->render.view = matmul(vp, inverse_camera_matrix(a.k[0], a.k[1], a.k[2], a.k[3]));
        st->render.enabled = 1;
    }
}

__global__ void k_control(DevState *__restrict__ st, FrameLogEntry *__restrict__ log, int ops, ControlArgs a) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    control_ops(st, log, ops, a);
}

// =====================================================================================
// Fused ICP iteration.  The reference loop (cpp:950-967) is track -> reduce -> host solve -> next track.  Here one
// launch per iteration does:   [prologue] add up the partial sums the PREVIOUS iteration left, solve the 6x6 system,
// update the pose and the level-converged flag;   [body] associate this level's pixels with that pose and leave new
// partial sums.  Every block runs the prologue redundantly on the same inputs (same instruction stream, same bits),
// so no block waits for another one and the launch boundary is the only synchronisation; block 0 records the state
// the next launch starts from.  Compared with separate track / solve launches this removes one dependent launch per
// iteration and the second early-exit launch of every skipped iteration.
// =====================================================================================
constexpr int ICP_THREADS = 1024;   // 16 waves: a wide block adds up the previous partials quickly and keeps their count low
#ifndef KFH_ICP_WAVES_PER_SIMD
#define KFH_ICP_WAVES_PER_SIMD 4
#endif
constexpr int ICP_WAVES_PER_SIMD = KFH_ICP_WAVES_PER_SIMD;   // 8 = two blocks per CU (<= 64 VGPRs, the fp64 solve spills)

struct IcpShared {
    M4 pose;
    int done[KFH_MAX_LEVELS];
    int prev_executed;
    int executed;
    float red[32];
    float sm[ICP_THREADS / 32][33];
    double jacobi_ws[72];
};

// Brings `sh` to the state entering iteration `it`: pose and flags after the update of iteration it-1.
#define ICP_STAMP(i) do { if (trace && blockIdx.x == 0 && threadIdx.x == 0) trace[i] = __builtin_readcyclecounter(); } while (0)
// COHERENT: the state record and the partial sums were written by other workgroups of THIS launch (k_icp_tail, after its
// grid barrier), so they are read with agent-scope atomic loads, which no cache of this CU or XCD may serve stale.
// AHEAD: request the partial rows before the state record (right for launches whose previous iteration usually ran; a
// launch that usually finds it skipped - the tail, the closing launch - would only put a useless round trip in front of
// its own later loads, which return in order).
template <bool COHERENT = false, bool AHEAD = true>
__device__ __forceinline__ void icp_advance(IcpShared &sh, DevState *__restrict__ st, const IcpState *iters,
                                            const float *partials_prev, int it, int prev_level,
                                            unsigned prev_nblocks, float icp_threshold, bool writer,
                                            unsigned long long *trace = nullptr) {
    ICP_STAMP(0);
    // the partial rows of the previous launch are requested before the state record that says whether they are needed
    // (one round trip instead of two on the path of every executed iteration; a skipped one discards them)
    const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
    constexpr int G = ICP_THREADS / 32;
    constexpr int PRE = 8;                        // rows per thread fetched ahead: covers 256 blocks
    float pre[PRE];
    // all PRE loads of a thread are issued back to back (one round trip, not PRE)
#define KFH_FETCH_ROWS()                                                                                                         \
    _Pragma("unroll") for (int k = 0; k < PRE; ++k) {                                                                            \
        const unsigned b = (unsigned)g + (unsigned)k * G;                                                                        \
        pre[k] = 0.f;                                                                                                            \
        if (it > 0 && b < prev_nblocks)                                                                                          \
            pre[k] = COHERENT ? __hip_atomic_load(partials_prev + (size_t)b * 32 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) \
                              : partials_prev[(size_t)b * 32 + c];                                                               \
    }
    if (AHEAD) { KFH_FETCH_ROWS() }
    if (threadIdx.x == 0) {
        if (it == 0) {
            sh.pose = st->pose;
            for (int l = 0; l < KFH_MAX_LEVELS; ++l) sh.done[l] = st->level_done[l];
            sh.prev_executed = 0;       // (the grid-barrier counters of k_icp_tail were cleared by track_begin)
        } else {
            IcpState prev;
            if (COHERENT) {
                static_assert(sizeof(IcpState) % 4 == 0, "word-wise copy");
                const unsigned *src = reinterpret_cast<const unsigned *>(iters + (it - 1));
                unsigned *dst = reinterpret_cast<unsigned *>(&prev);
                for (unsigned w = 0; w < sizeof(IcpState) / 4; ++w) dst[w] = __hip_atomic_load(src + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                prev = iters[it - 1];
            }
            sh.pose = prev.pose;
            for (int l = 0; l < KFH_MAX_LEVELS; ++l) sh.done[l] = prev.done[l];
            sh.prev_executed = prev.executed;
        }
    }
    __syncthreads();
    ICP_STAMP(1);
    if (!sh.prev_executed) return;   // uniform: the previous iteration was skipped, nothing to solve
    if (!AHEAD) { KFH_FETCH_ROWS() }
#undef KFH_FETCH_ROWS
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < PRE; ++k)
        if ((unsigned)g + (unsigned)k * G < prev_nblocks) acc += pre[k];
    for (unsigned b = (unsigned)g + PRE * G; b < prev_nblocks; b += G)     // (more than 256 blocks: KFH_ICP_BLOCKS)
        acc += COHERENT ? __hip_atomic_load(partials_prev + (size_t)b * 32 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : partials_prev[(size_t)b * 32 + c];
    sh.sm[g][c] = acc;
    __syncthreads();
    if (threadIdx.x < 32) {
        float v = sh.sm[0][threadIdx.x];
#pragma unroll
        for (int k = 1; k < G; ++k) v += sh.sm[k][threadIdx.x];
        sh.red[threadIdx.x] = v;
        if (writer) st->reduction[threadIdx.x] = v;
    }
    __syncthreads();
    ICP_STAMP(2);
    if (threadIdx.x == 0) {
        M4 pose = sh.pose;
        bool fast = true;
        const bool converged = (icp_threshold < 0.f) ? false : apply_update(pose, sh.red, icp_threshold, fast, sh.jacobi_ws, blockIdx.x == 0 ? trace : nullptr);   // < 0: timing probe
        sh.pose = pose;
        if (converged) sh.done[prev_level] = 1;
        KF_SOLVE_STAMP(blockIdx.x == 0 ? trace : nullptr, 15, sh.pose.m[0][0]);
        if (writer) {   // counters: atomics without a return value, so that this thread does not wait for the round trips
            atomicAdd(&st->icp_iterations, 1ull);
            atomicAdd(&st->icp_level_iterations[prev_level], 1ull);
            if (!fast) atomicAdd(&st->solve_fallbacks, 1ull);
        }
    }
    ICP_STAMP(3);
    __syncthreads();
}

// =====================================================================================
// Dispatch order of the raycast tiles for the next frame (see k_raycast): counting sort, most expensive first, of each
// tile's cost (which k_raycast has already dilated over the eight neighbours).  Ties go in any order - the
// order is a scheduling hint, never an input of the computation.  One 1024-thread block; `cost` holds n raw costs
// (cleared here) followed by n sort keys.  Runs as a side block of the first ICP launch of the next frame, or alone.
// =====================================================================================
constexpr int RAY_COST_BUCKETS = 64;               // cost classes (raycast: 4 march steps each; the longest marches are ~140 steps)
#ifndef KFH_RAY_SUPER_SHIFT
#define KFH_RAY_SUPER_SHIFT 3
#endif
constexpr int RAY_REGIONS = 8;                     // = XCDs: launch slot s runs on the XCD that all slots = s (mod 8) share
constexpr unsigned RAY_ORDER_KEPT_TILES = 8192;
struct RayOrderShared {
    unsigned hist[RAY_REGIONS][RAY_COST_BUCKETS], scan[RAY_REGIONS][RAY_COST_BUCKETS], offs[RAY_REGIONS][RAY_COST_BUCKETS];
    unsigned count[RAY_REGIONS];
};

// mode 0: raycast tiles (4 cost units per bucket, XCD classes); mode 1: integrate blocks (one class).
// XCD affinity (mode 0): workgroups are dealt round-robin over the 8 XCDs, each with its own L2, so the screen is cut into
// super-tiles of 8x8 tiles (64x64 pixels) that are dealt to 8 classes like a chess board; class k is sorted by cost on
// its own and gets the launch slots k, k + 8, k + 16, ... (as long as every class has tiles left): the rays of a
// super-tile, which sample the same voxels, share an L2 (measured: k_raycast's fetches from beyond L2 drop from 13.2 to 4.7 MB per launch), every
// XCD still starts with its most expensive tiles, and the classes carry equal shares of the screen's cost, unlike bands.
// (Which XCD a residue class lands on is the dispatcher's business; only the sharing matters.)
__device__ __forceinline__ void ray_order_block(RayOrderShared &sh, unsigned *__restrict__ order, unsigned *__restrict__ cost,
                                                unsigned tiles_x, unsigned tiles_y, unsigned mode) {
    const unsigned n = tiles_x * tiles_y, tid = threadIdx.x, nthr = blockDim.x;
    unsigned *key = cost + n;
    const bool staged = n <= RAY_ORDER_KEPT_TILES;
    const unsigned R = mode == 0 ? RAY_REGIONS : 1u;
    constexpr int SS = KFH_RAY_SUPER_SHIFT;                 // super-tile edge = 1 << SS tiles
    const unsigned super_x = (tiles_x + (1u << SS) - 1) >> SS;
    for (unsigned i = tid; i < RAY_REGIONS * RAY_COST_BUCKETS; i += nthr) (&sh.hist[0][0])[i] = 0;      // (nthr = 1024)
    __syncthreads();
    const float inv_tx = 1.0f / (float)tiles_x;
    constexpr int KEEP = RAY_ORDER_KEPT_TILES / 1024;     // keys of up to this many tiles stay in registers between the passes
    // (the kept keys are indexed by unrolled constants only: a run-time index would put the array into scratch memory, and a kernel
    // that declares scratch costs every launch 0.2-0.5 us - these blocks ride on the ICP launches, profiles/r04_launch_cost.txt)
    unsigned keys[KEEP];
    auto classify = [&](unsigned i) {
        const int ty = (int)(((float)i + 0.5f) * inv_tx), tx = (int)i - ty * (int)tiles_x;   // exact: i < 2^23
        unsigned m = cost[i];
        if (mode == 0) m >>= 2;
        const unsigned b = m < RAY_COST_BUCKETS - 1 ? m : RAY_COST_BUCKETS - 1;
        const unsigned r = R == 1 ? 0u : ((unsigned)(tx >> SS) + super_x * (unsigned)(ty >> SS) + (unsigned)(ty >> SS)) % RAY_REGIONS;
        atomicAdd(&sh.hist[r][b], 1u);
        return b | (r << 8);        // (b < 256)
    };
#pragma unroll
    for (int slot = 0; slot < KEEP; ++slot) {
        const unsigned i = tid + (unsigned)slot * nthr;
        keys[slot] = 0;
        if (i < n) {
            const unsigned k = classify(i);
            if (staged) keys[slot] = k; else key[i] = k;
        }
    }
    for (unsigned i = tid + (unsigned)KEEP * nthr; i < n; i += nthr) key[i] = classify(i);
    __syncthreads();
    // per class: offs[b] = number of its tiles in buckets above b.  One wave per class, one lane per bucket: an inclusive
    // scan of the reversed histogram by six shuffle steps (no barrier), minus the own count.
    static_assert(RAY_COST_BUCKETS == 64, "one lane per cost class");
    if (tid < RAY_REGIONS * 64) {
        const unsigned r = tid >> 6, lane = tid & 63u, b = RAY_COST_BUCKETS - 1 - lane;
        const unsigned own = sh.hist[r][b];
        unsigned v = own;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned up = (unsigned)__shfl_up((int)v, d, 64);
            v += lane >= (unsigned)d ? up : 0u;
        }
        sh.offs[r][b] = v - own;
        if (lane == 63) sh.count[r] = v;                       // tiles of the class
    }
    __syncthreads();
    unsigned cmin = sh.count[0];
    for (unsigned q = 1; q < R; ++q) cmin = cmin < sh.count[q] ? cmin : sh.count[q];
    auto place = [&](unsigned i, unsigned k) {
        const unsigned r = k >> 8;
        const unsigned pos = atomicAdd(&sh.offs[r][k & 255u], 1u);       // rank of the tile inside its class
        // slot of (class r, rank pos) when the classes take turns and a class that has run out is passed over
        unsigned s_ = R * pos + r;                                       // while every class still has tiles
        if (pos >= cmin) {
            s_ = 0;
            for (unsigned q = 0; q < R; ++q) {
                const unsigned cq = sh.count[q];
                s_ += cq < pos ? cq : pos;
                s_ += (q < r && cq > pos) ? 1u : 0u;
            }
        }
        order[R == 1 ? pos : s_] = i;
        cost[i] = 0;
    };
#pragma unroll
    for (int slot = 0; slot < KEEP; ++slot) {
        const unsigned i = tid + (unsigned)slot * nthr;
        if (i < n) place(i, staged ? keys[slot] : key[i]);
    }
    for (unsigned i = tid + (unsigned)KEEP * nthr; i < n; i += nthr) place(i, key[i]);
}

__global__ __launch_bounds__(1024) void k_ray_order(unsigned *__restrict__ order, unsigned *__restrict__ cost, unsigned tiles_x, unsigned tiles_y,
                                                    unsigned mode) {
    __shared__ RayOrderShared sh;
    ray_order_block(sh, order, cost, tiles_x, tiles_y, mode);
}

struct IcpIterArgs {
    DevState *st;
    IcpState *iters;
    const float *partials_prev;
    float *partials_out;
    int *result_map;
    int it, level, prev_level;
    unsigned prev_nblocks, nblocks;       // nblocks: ICP blocks of this launch (the grid may hold side-job blocks after them)
    const float4 *inV, *inN;
    unsigned inW, inH;
    const float4 *refV, *refN;
    unsigned refW, refH;
    float icp_threshold, dist_threshold, normal_threshold;
    float dist2_max;                      // >= 0: sqrtf(d) > dist_threshold <=> d > dist2_max for every float d (sqrt_gt_bound, verified on this device); < 0: take the square root
    float inv_inW;                        // RN(1 / inW), used for p / inW while inW * inH <= 2^22 (0 = integer division)
    unsigned long long *trace;            // debugging (KFH_ICP_TRACE): stamps of block 0 / thread 0, else nullptr
    unsigned *ray_order, *ray_cost;       // side block `nblocks` of a plain launch: raycast dispatch order (nullptr = none)
    unsigned ray_tiles_x, ray_tiles_y, order_mode;
};

// state entering iteration `it` (solve of iteration it-1 included), recorded by the writer block; true if `it` is to run
template <bool COHERENT = false, bool AHEAD = true>
__device__ __forceinline__ bool icp_enter(IcpShared &sh, const IcpIterArgs &a, int it, const float *partials_prev, bool writer) {
    DevState *__restrict__ st = a.st;
    unsigned long long *trace = a.trace;
    icp_advance<COHERENT, AHEAD>(sh, st, a.iters, partials_prev, it, a.prev_level, a.prev_nblocks, a.icp_threshold, writer, trace);
    if (threadIdx.x == 0) {
        sh.executed = sh.done[a.level] ? 0 : 1;
        if (writer) {
            IcpState &o = a.iters[it];
            o.pose = sh.pose;
            for (int l = 0; l < KFH_MAX_LEVELS; ++l) o.done[l] = sh.done[l];
            o.executed = sh.executed;
        }
    }
    __syncthreads();
    ICP_STAMP(4);
    return sh.executed != 0;
}

// trackKernel for one pixel (cpp:507-557) with the reduction's accumulation (cpp:297-356) inside the inlier branch: the operations, their
// order and hence the bits are those of track_fetched() + accumulate(), but nothing is carried across the early exits except the
// result code - as a PixelTrack handed from one routine to the other, every exit zeroed the seven error / Jacobian registers again
// (~45 of the ~230 vector instructions per pixel of a kernel that is bound by instruction issue).  The two gathers from the reference
// maps are issued together (the compiler otherwise sinks the vertex and the normal's y / z behind the tests that need them: three
// dependent round trips); length(diff) > dist_threshold is decided on the squared length where sqrt_gt_bound() applies.
#ifndef KFH_ICP_LEAN
#define KFH_ICP_LEAN 1
#endif
#ifndef KFH_ICP_BALANCE
#define KFH_ICP_BALANCE 1      // remainder pixels of a level larger than one pass dealt to the blocks' waves (see icp_track_block)
#endif
struct AssocConsts { M4 T, view; int have_reference; unsigned refW; float wm1, hm1, dist_threshold, dist2_max, normal_threshold; };

__device__ __forceinline__ int associate_accumulate(float (&acc)[32], float4 n4, float4 v4, const float4 *__restrict__ refV,
                                                    const float4 *__restrict__ refN, const AssocConsts &C) {
    if (n4.x == kInvalid) return -1;
    const f3 pv = xform(C.T, mk3(v4.x, v4.y, v4.z));
    unsigned ref = 0;
    if (C.have_reference) {
        const f3 pp = xform(C.view, pv);
        const float u = pp.x / pp.z + 0.5f, w = pp.y / pp.z + 0.5f;
        if (u < 0 || u > C.wm1 || w < 0 || w > C.hm1) return -2;
        ref = (unsigned)u + (unsigned)w * C.refW;
    }
    float4 rn4 = refN[ref];
    float4 rv4 = refV[ref];
    asm volatile("" : "+v"(rn4.x), "+v"(rn4.y), "+v"(rn4.z), "+v"(rv4.x), "+v"(rv4.y), "+v"(rv4.z));   // both gathers requested here, whole
    if (rn4.x == kInvalid) return -3;
    const f3 n_ref = mk3(rn4.x, rn4.y, rn4.z);
    const f3 diff = sub3(mk3(rv4.x, rv4.y, rv4.z), pv);
    const float d2 = dot3(diff, diff);
    if (C.dist2_max >= 0.f ? d2 > C.dist2_max : sqrtf(d2) > C.dist_threshold) return -4;     // (uniform choice)
    const f3 pn = rotate(C.T, mk3(n4.x, n4.y, n4.z));
    if (dot3(pn, n_ref) < C.normal_threshold) return -5;
    const float e = dot3(n_ref, diff);
    const f3 c = cross3(pv, n_ref);
    const float J[6] = {n_ref.x, n_ref.y, n_ref.z, c.x, c.y, c.z};
    acc[0] += e * e;
#pragma unroll
    for (int i = 0; i < 6; ++i) acc[1 + i] += e * J[i];
    int k = 7;
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int q = r; q < 6; ++q) acc[k++] += J[r] * J[q];
    return 1;
}

// association + partial sums of ICP block `vb` (of a.nblocks) with the pose in sh
__device__ __forceinline__ void icp_track_block(IcpShared &sh, const IcpIterArgs &a, unsigned vb, float *partials_out) {
    DevState *__restrict__ st = a.st;
    unsigned long long *trace = a.trace;
    const M4 T = sh.pose;
    const M4 view = st->track.view;
    const int have_reference = st->track.have_reference;
    float acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = 0.f;
    const unsigned n = a.inW * a.inH;
    const float4 *__restrict__ inV = a.inV, *__restrict__ inN = a.inN, *__restrict__ refV = a.refV, *__restrict__ refN = a.refN;
    int *__restrict__ result_map = a.result_map;
#if KFH_ICP_LEAN
    AssocConsts C;
    C.T = T; C.view = view; C.have_reference = have_reference; C.refW = a.refW;
    C.wm1 = (float)(a.refW - 1); C.hm1 = (float)(a.refH - 1);                // (the reference compares with the unsigned size - 1, cpp:527)
    C.dist_threshold = a.dist_threshold; C.dist2_max = a.dist2_max; C.normal_threshold = a.normal_threshold;
    // The four counters of the row ([28] inliers, [29] #(-4), [30] #(-5), [31] #(-1..-3)) are small integers, exact in float
    // whatever the order of the additions: counted per WAVE on the scalar unit (ballot + bit count) and handed to lane 0 below,
    // instead of three or four vector instructions per pixel and lane.
    unsigned cnt_in = 0, cnt_dist = 0, cnt_norm = 0, cnt_other = 0;
    const unsigned row_gap = a.refW - a.inW;     // the result map keeps the full-resolution row stride on every level (cpp:512)
    auto one_pixel = [&](unsigned p) {
        // the input maps are dense (row stride inW): pixel (px, py) is element p
        float4 n4 = inN[p];
        float4 v4 = inV[p];
        asm volatile("" : "+v"(n4.x), "+v"(n4.y), "+v"(n4.z), "+v"(v4.x), "+v"(v4.y), "+v"(v4.z));   // one round trip for both maps
        const int result = associate_accumulate(acc, n4, v4, refV, refN, C);
        unsigned o = p;
        if (row_gap) {      // (uniform) coarse level: p + py * (refW - inW); py = p / inW by a multiplication while that is exact
            const unsigned py = a.inv_inW > 0.f ? (unsigned)(((float)p + 0.5f) * a.inv_inW) : p / a.inW;
            o = p + __umul24(py, row_gap);
        }
        result_map[o] = result;
        cnt_in += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(result == 1));
        cnt_dist += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(result == -4));
        cnt_norm += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(result == -5));
        cnt_other += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(result < 0 && result > -4));
    };
    // Which pixels a block owns.  A level of at most nblocks x 1024 pixels: block vb its 1024 consecutive ones.  A larger level
    // (640x480: 300 groups of 1024 on 256 blocks) in full passes over all blocks - and the remainder, which is less than one pass, in
    // 64-pixel chunks dealt round-robin to the blocks' WAVES (chunk c -> block c % nblocks, wave c / nblocks) instead of by whole
    // groups to the first blocks: those then had two pixels on every thread - eight pixel-rows per SIMD where the others have four -
    // and the launch waited for them; now a block has at most three waves with a second pixel, five rows on three of its SIMDs.
    // The partition is part of the sum tree: it is the same for every schedule (launch per iteration, k_icp_tail's emulated blocks).
    const unsigned stride = a.nblocks * ICP_THREADS;
    const unsigned full = KFH_ICP_BALANCE ? n / stride : 0u;
    if (full == 0) {
        for (unsigned p = vb * ICP_THREADS + threadIdx.x; p < n; p += stride) one_pixel(p);
    } else {
        unsigned p = vb * ICP_THREADS + threadIdx.x;
        for (unsigned k = 0; k < full; ++k, p += stride) one_pixel(p);
        const unsigned chunk = vb + (threadIdx.x >> 6) * a.nblocks;
        const unsigned pr = full * stride + chunk * 64u + (threadIdx.x & 63u);
        if (pr < n) one_pixel(pr);
    }
    if ((threadIdx.x & 63) == 0) { acc[28] = (float)cnt_in; acc[29] = (float)cnt_dist; acc[30] = (float)cnt_norm; acc[31] = (float)cnt_other; }
#else
    for (unsigned p = vb * ICP_THREADS + threadIdx.x; p < n; p += a.nblocks * ICP_THREADS) {
        const unsigned py = p / a.inW, px = p - py * a.inW;
        const PixelTrack r = track_pixel(px, py, inV, inN, a.inW, refV, refN, a.refW, a.refH, T, view, have_reference,
                                         a.dist_threshold, a.normal_threshold);
        result_map[px + py * a.refW] = r.result;   // full-resolution row stride on every level (cpp:512)
        accumulate(acc, r.result, r.error, r.J);
    }
#endif
    ICP_STAMP(5);
    block_reduce_store<ICP_THREADS>(acc, partials_out + (size_t)vb * 32);
    ICP_STAMP(6);
}

__device__ __forceinline__ void icp_iter_block(IcpShared &sh, const IcpIterArgs &a) {
    if (!icp_enter(sh, a, a.it, a.partials_prev, blockIdx.x == 0)) return;
    icp_track_block(sh, a, blockIdx.x, a.partials_out);
}

__global__ __launch_bounds__(ICP_THREADS, ICP_WAVES_PER_SIMD) void k_icp_iter(IcpIterArgs a) {
    __shared__ IcpShared sh;
    icp_iter_block(sh, a);
}

// plain iteration plus one side block that sorts the raycast tiles (first launch of a queued frame)
__global__ __launch_bounds__(ICP_THREADS, ICP_WAVES_PER_SIMD) void k_icp_iter_order(IcpIterArgs a) {
    __shared__ IcpShared sh;
    __shared__ RayOrderShared ro;
    if (blockIdx.x < a.nblocks) { icp_iter_block(sh, a); return; }
    ray_order_block(ro, a.ray_order, a.ray_cost, a.ray_tiles_x, a.ray_tiles_y, a.order_mode);
}

// End of a frame's ICP loop (one thread of the closing launch): final pose and flags, then the control operations that
// follow tracking.  (Staging the head of DevState in LDS for this scalar glue was tried: no gain, the chain is arithmetic.)
__device__ __forceinline__ void close_icp_loop(IcpShared &sh, DevState *st, FrameLogEntry *log, int ops, const ControlArgs &ca) {
    if (threadIdx.x == 0) {
        st->pose = sh.pose;
        for (int l = 0; l < KFH_MAX_LEVELS; ++l) st->level_done[l] = sh.done[l];
        control_ops(st, log, ops, ca);
    }
}

// ---- tail launch --------------------------------------------------------------------------------------------------
// The late iterations of a level are queued for the worst case but almost never run (640x480 living room: level 0 stops
// after 3 of its 10 iterations in 91 % of the frames, after 4 in 7 %), and an empty launch still costs ~2 us of queue
// time.  k_icp_tail runs the last `n_its` iterations of a level in ONE launch: a few workgroups (ICP_TAIL_BLOCKS times the
// units the context has reserved: resident together whatever else the device runs) loop over the iterations, each emulating the
// a.nblocks ICP blocks of an ordinary launch in turn - same pixels per block, same partial rows, same sums, hence the
// same bits - with a grid barrier between iterations (release/acquire on a counter in DevState; the partial sums and the
// state record are re-read with agent-scope loads).  All workgroups take the same decisions from the same data, so they
// leave together as soon as the level has converged, and block 0 fills in the records of the iterations that were not
// needed.  An iteration that does run here is several times slower than in a launch of its own; the schedule keeps the
// first ICP_TAIL_HEAD iterations of a level in ordinary launches so that this is rare.
// A workgroup that waits longer than ~2 s at the barrier (which needs another process to hog the device) counts a
// time-out in DevState::barrier_timeouts and carries on, so the grid always drains; kfh reports the frame as failed.
constexpr unsigned ICP_TAIL_BLOCKS = 32;        // one reservation unit (kf_pipeline.hip); a launch may be given several

struct IcpTailArgs {
    float *part[2]; int n_its; unsigned *barrier; unsigned *timeouts;
    unsigned long long *trace;          // debugging (KFH_ICP_TAIL_TRACE), else nullptr
    unsigned long long timeout_ticks;   // of the 100 MHz wall clock (2 s; KFH_ICP_TAIL_TIMEOUT_TICKS for the error-path test)
    int final_ops;            // >= 0: the launch closes the frame's ICP loop with these control operations (k_icp_final's job)
    FrameLogEntry *log; ControlArgs ca;
};

__device__ __forceinline__ void grid_barrier(unsigned *counter, unsigned goal, unsigned *timeouts, unsigned long long timeout_ticks) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();                                    // release: this workgroup's partial rows / state record
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t0 = wall_clock64();       // 100 MHz
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < goal) {
            __builtin_amdgcn_s_sleep(2);
            if (wall_clock64() - t0 > timeout_ticks) { atomicAdd(timeouts, 1u); break; }
        }
        __threadfence();                                    // acquire
    }
    __syncthreads();
}

// The iterations of the launch that do run (rare), AFTER the verdict the usual case leaves with: written as one loop with that
// verdict inside it, the loop's set-up - its invariants, spilled addresses, the association's constants - was hoisted in front of
// the first barrier and the kernel spent 7.4 k - 12.9 k cycles between entry and the verdict, depending on how the register
// allocation fell; 1.5 k now.  (Inlined: as a called routine its two argument structures were copied to scratch at the kernel's top.)
// Enters with the state of iteration a.it in `sh` and that iteration due.  Returns true if the caller (the writer workgroup of a
// closing launch) is to close the frame's ICP loop.
__device__ __forceinline__ bool icp_tail_iterations(IcpShared &sh, const IcpIterArgs &a, const IcpTailArgs &t) {
    const bool writer = blockIdx.x == 0;
    const bool closes = t.final_ops >= 0;
    for (int j = 0;; ++j) {
        const int it = a.it + j;
        for (unsigned vb = blockIdx.x; vb < a.nblocks; vb += gridDim.x) {
            icp_track_block(sh, a, vb, t.part[it & 1]);
            __syncthreads();                                // block_reduce_store's LDS rows are reused
        }
        if (j + 1 < t.n_its || closes) grid_barrier(t.barrier, (unsigned)(j + 1) * gridDim.x, t.timeouts, t.timeout_ticks);
        if (j + 1 == t.n_its) {
            // every queued iteration ran: the update of the last one, as k_icp_final would apply it
            if (!(closes && writer)) return false;
            icp_advance<true, true>(sh, a.st, a.iters, t.part[it & 1], it + 1, a.level, a.nblocks, a.icp_threshold, true);
            return true;
        }
        if (icp_enter<true, true>(sh, a, it + 1, t.part[it & 1], writer)) continue;
        // the level is done: the remaining iterations would only hand the state on (as their launches did)
        if (!writer) return false;
        if ((int)threadIdx.x < a.it + t.n_its - (it + 2)) {     // one thread per record
            IcpState &o = a.iters[it + 2 + (int)threadIdx.x];
            o.pose = sh.pose;
            for (int l = 0; l < KFH_MAX_LEVELS; ++l) o.done[l] = sh.done[l];
            o.executed = 0;
        }
        return closes;
    }
}

// AHEAD: the partial rows of the previous launch are requested before the state record (right when the iteration before this
// launch usually ran, i.e. when the launch usually has a system to solve: tails that begin early in a level)
template <bool AHEAD>
__global__ __launch_bounds__(ICP_THREADS, ICP_WAVES_PER_SIMD) void k_icp_tail(IcpIterArgs a, IcpTailArgs t) {
    __shared__ IcpShared sh;
    const bool writer = blockIdx.x == 0;
    const bool closes = t.final_ops >= 0;       // this launch also does k_icp_final's work (block 0, after the loop)
    unsigned long long *trace = t.trace;        // debugging (KFH_ICP_TAIL_TRACE): stamps of block 0 / thread 0
    ICP_STAMP(0);
    // the state entering the first queued iteration (the update from the previous launch's sums, if it ran) - and, almost always,
    // the verdict that the level has converged: the workgroups leave, the writer hands the state on and closes the loop
    const bool run = icp_enter<false, AHEAD>(sh, a, a.it, t.part[(a.it + 1) & 1], writer);
    ICP_STAMP(1);
    bool close = closes && writer;
    if (!run) {
        if (!writer) return;
        if ((int)threadIdx.x < t.n_its - 1) {       // the records the skipped iterations would have written, one thread each
            IcpState &o = a.iters[a.it + 1 + (int)threadIdx.x];
            o.pose = sh.pose;
            for (int l = 0; l < KFH_MAX_LEVELS; ++l) o.done[l] = sh.done[l];
            o.executed = 0;
        }
        ICP_STAMP(2);
    } else {
        close = icp_tail_iterations(sh, a, t);
    }
    // block 0 of the launch that ends the frame's ICP loop: what k_icp_final does after its own icp_advance
    if (close) close_icp_loop(sh, a.st, t.log, t.final_ops, t.ca);
    ICP_STAMP(3);
}

// ---- side jobs ----------------------------------------------------------------------------------------------------
// The ICP launches of the coarse pyramid levels occupy 19 (160x120) or 75 (320x240) of the 256 CUs and are bound by
// dependent latencies.  The queued-frame path (kfh_compute_frame) uses the idle CUs for the input-only work of the
// NEXT frame - mm2meters + bilateral filter, then the input pyramid - which depends on nothing the current frame
// computes: blocks past `a.nblocks` run tiles of that work, in teams of 256 threads (one block = 4 teams), each team
// executing the very routine the stand-alone kernel runs.  The maps are double-buffered by the caller, so the ICP
// blocks of this launch and the side blocks never touch the same buffer.
struct SideBilateral {
    float *floatDepth, *out;
    const void *in;
    unsigned W, H, inW;
    int ratio;
    Gauss5 g;
    float e_d_squared_2, rcp;
    unsigned first, count;    // range of 32x8 tiles (row-major over the tile grid) this launch covers
};

struct SidePyramid {
    PyramidArgs a;
    unsigned first, count;    // range of 32x32 tiles
};

constexpr int SIDE_TEAMS = ICP_THREADS / 256;

template <bool FD>
__global__ __launch_bounds__(ICP_THREADS, ICP_WAVES_PER_SIMD) void k_icp_iter_bilateral(IcpIterArgs a, SideBilateral sb) {
    __shared__ IcpShared sh;
    __shared__ float tiles[SIDE_TEAMS][PBY + 2 * PR][PBX + 2 * PR + 1];
    __shared__ double exp_tab[32];
    __shared__ RayOrderShared ro;
    if (blockIdx.x < a.nblocks) { icp_iter_block(sh, a); return; }
    const unsigned order_blocks = a.ray_order ? 1u : 0u;     // the raycast dispatch sort, if asked for, is the last block
    if (order_blocks && blockIdx.x == gridDim.x - 1) { ray_order_block(ro, a.ray_order, a.ray_cost, a.ray_tiles_x, a.ray_tiles_y, a.order_mode); return; }
    const unsigned side = blockIdx.x - a.nblocks, nside = gridDim.x - a.nblocks - order_blocks;
    const int team = threadIdx.x >> 8, tid = threadIdx.x & 255;
    if (threadIdx.x < 32) exp_tab[threadIdx.x] = __builtin_bit_cast(double, kExp2fTable[threadIdx.x]);
    const unsigned tiles_x = (sb.W + PBX - 1) / PBX;
    // tile i of the range goes to block i % nside, team (i / nside) % SIDE_TEAMS: the range is spread over all side blocks
    // (one per idle CU) before any block gets a second tile, and every block runs the same number of rounds
    const unsigned rounds = (sb.count + nside * SIDE_TEAMS - 1) / (nside * SIDE_TEAMS);
    for (unsigned r = 0; r < rounds; ++r) {
        const unsigned i = (r * SIDE_TEAMS + team) * nside + side;
        const bool active = i < sb.count;
        const unsigned t = sb.first + i;
        bilateral_block<true, FD>(tiles[team], exp_tab, (int)(t % tiles_x), (int)(t / tiles_x), tid, active, sb.floatDepth, sb.out, sb.in,
                                  sb.W, sb.H, sb.inW, sb.ratio, sb.g, sb.e_d_squared_2, sb.rcp);
        __syncthreads();   // the tiles are reused by the next round
    }
}

#ifndef KFH_PYR_SIDE_TEAMS
#define KFH_PYR_SIDE_TEAMS 4
#endif
constexpr int PYR_SIDE_TEAMS = KFH_PYR_SIDE_TEAMS;      // tiles a side block of the pyramid job works on at a time (1, 2, 4)
// PART 0: the whole pyramid of a tile; 1: level 0 only (the geometry of a one-level pyramid: a 1-pixel halo); 2: the levels above
// it.  A tile's chain - cold tile load, level 0's maps, then a barrier and a pass per coarser level - is what the launch waits
// for (11.6 us against 7.3 us for the iteration alone at 640x480, 28 against 10 at 1280x960); cut in two it hides behind two
// iterations.  Parts 1 and 2 read the same filtered depth and write different maps, in either order.
template <int NLEV, int PART, int TEAMS = PYR_SIDE_TEAMS>
__global__ __launch_bounds__(ICP_THREADS, ICP_WAVES_PER_SIMD) void k_icp_iter_pyramid(IcpIterArgs a, SidePyramid sp) {
    constexpr int GEOM = PART == 1 ? 1 : NLEV;
    __shared__ IcpShared sh;
    __shared__ float lds[TEAMS][pyr_lds_floats<GEOM>()];
    __shared__ RayOrderShared ro;
    if (blockIdx.x < a.nblocks) { icp_iter_block(sh, a); return; }
    const unsigned order_blocks = a.ray_order ? 1u : 0u;     // a dispatch sort, if asked for, is the last block
    if (order_blocks && blockIdx.x == gridDim.x - 1) { ray_order_block(ro, a.ray_order, a.ray_cost, a.ray_tiles_x, a.ray_tiles_y, a.order_mode); return; }
    const unsigned side = blockIdx.x - a.nblocks, nside = gridDim.x - a.nblocks - order_blocks;
    const unsigned tiles_x = (sp.a.W + PYR_T - 1) / PYR_T;
    // A tile is a chain of barriers (load, then level by level) that lasts ~7 us however many threads work on it, so the block
    // runs TEAMS tiles at a time, each by a team of the size the stand-alone kernel's blocks have (same routine, same
    // bits), instead of one tile by all 1024 threads (the launch that carries the pyramid at 1280x960: 39.5 -> 29 us; eight teams: 27 us there but
    // +7 us at 640x480, profiles/r02_sweeps.md).  Tile i of the range goes to block
    // i % nside, team (i / nside) % TEAMS; a team without a tile runs the routine on a tile position right of the image,
    // which passes every barrier and writes nothing.
    constexpr int TEAM_THREADS = ICP_THREADS / TEAMS;
    const int team = (int)threadIdx.x / TEAM_THREADS, tid = (int)threadIdx.x % TEAM_THREADS;
    const unsigned rounds = (sp.count + nside * TEAMS - 1) / (nside * TEAMS);
    for (unsigned r = 0; r < rounds; ++r) {
        const unsigned i = (r * TEAMS + (unsigned)team) * nside + side;
        const unsigned t = sp.first + i;
        const bool active = i < sp.count;
        pyramid_block<GEOM, TEAM_THREADS, PART == 2>(sp.a, lds[team], active ? (int)(t % tiles_x) : (int)tiles_x, active ? (int)(t / tiles_x) : 0, tid);
        __syncthreads();
    }
}

// Closes the loop: the update of the last iteration, then the scalar glue that follows tracking (checkPose, integrate
// gate, raycast pose, frame log) — one launch instead of a solve and a control launch.
__global__ __launch_bounds__(ICP_THREADS) void k_icp_final(DevState *__restrict__ st, FrameLogEntry *__restrict__ log,
                                                          const IcpState *__restrict__ iters,
                                                          const float *__restrict__ partials_prev, int n_iters,
                                                          int prev_level, unsigned prev_nblocks, float icp_threshold,
                                                          int ops, ControlArgs a) {
    __shared__ IcpShared sh;
    icp_advance<false, false>(sh, st, iters, partials_prev, n_iters, prev_level, prev_nblocks, icp_threshold, true);
    close_icp_loop(sh, st, log, ops, a);
}

// =====================================================================================
// integrateKernel (cpp:628-673).  A wave owns a 16 x 4 patch of (x, y) columns of one z-chunk (KFH_INT_TX; a block 32 x 8 columns):
// every voxel access is four 64-byte segments of the x-fastest short2 grid, and the compact footprint keeps the wave's lanes
// together in what they find (round 1 used 64 x 1: full 256-byte lines, but rows 1.2 m long that cut through everything).  The camera-space position is carried along z by
// repeated addition exactly like the CPU loop (cpp:645-646); a thread that starts at z0 > 0 first replays
// the z0 additions in registers so that its float state carries the bits of the serial march.
// The six running sums are kept as three float pairs and advanced with packed adds (v_pk_add_f32).
//
// Most voxels do not project into the image, and the kernel is instruction bound, not byte bound, so the work per
// rejected voxel is what matters.  The z march is cut into batches of IB voxels:
//   * batch reject: x/z and y/z are monotonic along the column while z stays positive, so if BOTH end voxels of a batch
//     are outside the image on the same side with a 0.1-pixel margin (tested with multiplications), or both are behind
//     the camera, every voxel of the batch fails cpp:647-653 and only the IB packed additions are spent on it;
//   * otherwise each voxel runs the reference tests, with the two IEEE divisions of the projection skipped for voxels
//     that the same multiplication test already puts outside;
//   * the depth gathers of a batch are issued together, then the voxel reads of the survivors;
//   * t = depth - cam.z decides cheaply what the reference computes through diff = t * sqrt(1 + (x/z)^2 + (y/z)^2):
//     since that factor is >= 1, t <= -mu means diff <= -mu (skip, no voxel access) and t >= mu means
//     diff / mu >= 1, i.e. sdf == 1.0f exactly; only the band |t| < mu needs the two divisions and the square root.
// Every decision and every stored short is the one the CPU loop produces.
// =====================================================================================
constexpr short kSkipShortFwd = 26600;
__device__ __forceinline__ unsigned umax_(unsigned a, unsigned b) { return a > b ? a : b; }
__device__ __forceinline__ void mark_bricks(unsigned char *flags, int nbx, int nby, int nbz, int x, int y, int z);

typedef float v2f __attribute__((ext_vector_type(2)));

struct ColumnState { v2f a, b; float cy, cz; };   // (pos.x, pos.y), (pos.z, cam.x), cam.y, cam.z (cz unused when KZ)

// conservative "projects outside the image on this side" bits; valid only where cam.z > 0.
// lo = -(0.5 + margin), wlim/hlim = size - 0.5 + margin:  x/z < lo => x/z + 0.5 < 0;  x/z > wlim => x/z + 0.5 > W - 1.
__device__ __forceinline__ unsigned outside_bits(float camx, float camy, float camz, float lo, float wlim, float hlim) {
    unsigned m = 0;
    m |= (camx < lo * camz) ? 1u : 0u;
    m |= (camx > wlim * camz) ? 2u : 0u;
    m |= (camy < lo * camz) ? 4u : 0u;
    m |= (camy > hlim * camz) ? 8u : 0u;
    return m;
}

#ifndef KFH_INT_BX
#define KFH_INT_BX 32
#endif
#ifndef KFH_INT_BY
#define KFH_INT_BY 8
#endif
#ifndef KFH_INT_TX
#define KFH_INT_TX 16
#endif
constexpr unsigned INT_BX = KFH_INT_BX, INT_BY = KFH_INT_BY, INT_TX = KFH_INT_TX;
static_assert(INT_BX * INT_BY == 256 && INT_BX % INT_TX == 0 && 64 % INT_TX == 0 && (INT_BX / INT_TX) * (INT_BY / (64 / INT_TX)) == 4,
              "a block of k_integrate is 256 columns = four waves of INT_TX x (64 / INT_TX)");

// KZ: the third row of K is (0, 0, 1, 0) (every pinhole matrix camera_matrix() builds; the launcher checks): then
// cam.z = 0*x + 0*y + 1*z + 0 and its step 0*dx + 0*dy + 1*dz are pos.z and its step bit for bit, the two running sums are the
// same sequence of floats, and the march carries five sums instead of six (a packed add becomes a plain one).
template <bool KZ> __device__ __forceinline__ void march_step(ColumnState &S, v2f da, v2f db, v2f dc) {
    S.a += da; S.b += db;
    S.cy += dc.x;
    if (!KZ) S.cz += dc.y;
}
template <bool KZ> __device__ __forceinline__ float cam_z(const ColumnState &S) { return KZ ? S.b.x : S.cz; }

// Stage probes for the instruction census (profiles/r05_integrate_census.txt): ONLY in a build with -DKFH_INT_PROBES=1 (through
// KFH_EXTRA_HIPCC_FLAGS) does the kernel look at `dbg` - results are wrong with a probe on; the shipped library has no such path
// (tests/test_capi_symbols.py greps the default build for the switch's name).
#ifndef KFH_INT_PROBES
#define KFH_INT_PROBES 0
#endif
#if KFH_INT_PROBES
#define KFH_INT_PROBE(bit) ((dbg & (bit)) != 0)
#else
#define KFH_INT_PROBE(bit) false
#endif
#ifndef KFH_INTEGRATE_WAVES_PER_SIMD
#define KFH_INTEGRATE_WAVES_PER_SIMD 5      // occupancy floor handed to the register allocator (5 = at most 96 VGPRs)
#endif
template <int IB, bool KZ>
__global__ __launch_bounds__(256, IB <= 8 ? KFH_INTEGRATE_WAVES_PER_SIMD : 1) void k_integrate(short2 *__restrict__ vol, unsigned sx, unsigned sy, unsigned sz,
                                                   f3 dim, const float *__restrict__ depth, unsigned W, unsigned H,
                                                   const IntegrateParams *__restrict__ ip, float mu, float maxweight,
                                                   unsigned zchunk, unsigned long long *__restrict__ updated_total,
                                                   unsigned char *__restrict__ skip_flags, int dbg,
                                                   unsigned long long *__restrict__ trace,
                                                   const unsigned *__restrict__ order, unsigned *__restrict__ cost) {
    // everything a wave needs before it can decide anything - the launch parameters the control operation left in DevState
    // (gate, matrices, bounds) and its entry of the dispatch order - is requested in one go: written in the order of use
    // these are five dependent scalar round trips, ~1.5 us at the head of every wave, two thirds of which leave right after
    const unsigned slot = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const IntegrateParams P = *ip;
    const unsigned vb = order ? order[slot] : slot;
    // (the empty statement "uses" one word of every piece the structure is fetched in, so that the compiler cannot put the
    // matrices' loads back behind the gate)
    asm volatile("" : : "s"(P.invTrack.m[0][0]), "s"(P.invTrack.m[2][0]), "s"(P.K.m[0][0]), "s"(P.K.m[2][0]), "s"(vb));
    if (!P.enabled) return;
    const unsigned long long t_start = trace ? wall_clock64() : 0ull;
    // Launch slot -> block of the (x / 64, y / 4, z-chunk) grid.  As in k_raycast: the launch ends with its slowest
    // wave (a column chunk lying in the truncation band costs ~10x one outside the frustum), so the blocks that were
    // expensive in the previous launch are dispatched first instead of wherever x-y-z order puts them (at 512^3 the
    // far chunks, i.e. the band, used to start 55 us into a 170 us launch).
    const unsigned vbx = vb % gridDim.x, vby = (vb / gridDim.x) % gridDim.y, vbz = vb / (gridDim.x * gridDim.y);
    // Columns of a block: INT_BX x INT_BY; of a wave: INT_TX x (64 / INT_TX), the waves side by side in x first.  A compact
    // wave footprint keeps the 64 lanes together in what they find - inside or outside the frustum, in front of or behind
    // the surface - and a row of INT_TX voxels is still a 64-byte segment at 16 (profiles/r02_sweeps.md).
    const unsigned wave_ = threadIdx.y, lane_ = threadIdx.x;
    constexpr unsigned WAVES_X = INT_BX / INT_TX, TILE_Y = 64 / INT_TX;
    const unsigned x = vbx * INT_BX + (wave_ % WAVES_X) * INT_TX + lane_ % INT_TX;
    const unsigned y = vby * INT_BY + (wave_ / WAVES_X) * TILE_Y + lane_ / INT_TX;
    unsigned updated = 0, work = 0;
    if (x < sx && y < sy) {
        const M4 invTrack = P.invTrack, K = P.K;
        const f3 delta = rotate(invTrack, mk3(0.f, 0.f, dim.z / sz));
        const f3 cameraDelta = rotate(K, delta);
        const f3 pos0 = xform(invTrack, mk3((x + 0.5f) * dim.x / sx, (y + 0.5f) * dim.y / sy, (0 + 0.5f) * dim.z / sz));
        const f3 cam0 = xform(K, pos0);
        ColumnState S;
        S.a = (v2f){pos0.x, pos0.y}; S.b = (v2f){pos0.z, cam0.x}; S.cy = cam0.y; S.cz = cam0.z;
        const v2f da = (v2f){delta.x, delta.y}, db = (v2f){delta.z, cameraDelta.x}, dc = (v2f){cameraDelta.y, cameraDelta.z};
        const unsigned z0 = vbz * zchunk;
        const unsigned z1 = z0 + zchunk < sz ? z0 + zchunk : sz;
        // Whole-chunk reject BEFORE the replay.  The marched state of voxel z is the exact line pos0 + z * delta plus at
        // most z rounded additions, each off by <= 2^-24 of the value: |error| <= E = (sz + 4) 2^-24 cam_bound for cam.x / cam.y
        // and <= Ez = (sz + 4) 2^-24 far_bound for cam.z = pos.z.  The reference skips a voxel when pos.z < 1e-4 or the pixel
        // cam.x / cam.z + 0.5 falls outside [0, W-1] x [0, H-1] (cpp:647-653); "outside on the left" is cam.x + 0.5 cam.z < 0
        // for cam.z > 0, a LINEAR condition, so if the un-marched end points of the chunk satisfy it with E + 0.5 Ez to
        // spare (plus a quarter pixel for the rounding of the reference's own division), every marched voxel in between
        // does: the chunk updates nothing and the thread leaves without replaying its z0 additions.  Four of five chunks
        // lie outside the frustum; what remains pays the replay, which is what lets the chunks be short (see integrate_grid).
        bool chunk_out = false;
        if (!KFH_INT_PROBE(8)) {
            const float zf = (float)z0, zl = (float)(z1 - 1);
            const float n_err = (float)(sz + 4) * 5.9604645e-8f;
            const float E = n_err * P.cam_bound, Ez = n_err * P.far_bound;
            const float pzf = __builtin_fmaf(zf, delta.z, pos0.z), pzl = __builtin_fmaf(zl, delta.z, pos0.z);
            const float cxf = __builtin_fmaf(zf, cameraDelta.x, cam0.x), cxl = __builtin_fmaf(zl, cameraDelta.x, cam0.x);
            const float cyf = __builtin_fmaf(zf, cameraDelta.y, cam0.y), cyl = __builtin_fmaf(zl, cameraDelta.y, cam0.y);
            const float czf = __builtin_fmaf(zf, cameraDelta.z, cam0.z), czl = __builtin_fmaf(zl, cameraDelta.z, cam0.z);
            const bool behind = pzf < -Ez - 1e-3f && pzl < -Ez - 1e-3f;           // every marched pos.z < 1e-4
            if (czf > Ez + 1e-3f && czl > Ez + 1e-3f) {                            // every marched cam.z > 0
                const float wq = (float)W - 0.25f, hq = (float)H - 0.25f;         // right / bottom: pixel > size - 1, i.e. c > (size - 0.5) z
                const float el = E + 0.75f * Ez, er = E + wq * Ez, eb = E + hq * Ez;
                const bool left = cxf + 0.75f * czf < -el && cxl + 0.75f * czl < -el;
                const bool right = cxf - wq * czf > er && cxl - wq * czl > er;
                const bool top = cyf + 0.75f * czf < -el && cyl + 0.75f * czl < -el;
                const bool bottom = cyf - hq * czf > eb && cyl - hq * czl > eb;
                chunk_out = left || right || top || bottom;
            }
            chunk_out = chunk_out || behind;
        }
        if (!chunk_out)
        for (unsigned z = 0; z < z0; ++z) { march_step<KZ>(S, da, db, dc); }
        const size_t plane = (size_t)sx * sy;
        short2 *cell = vol + x + (size_t)y * sx + (size_t)z0 * plane;
        const float wm1 = (float)(W - 1), hm1 = (float)(H - 1);
        // per-voxel test: the very floats the reference divides, 0.1 px of margin covers the rounding of the product.
        // batch test: interior voxels are the endpoints' interpolation plus at most IB roundings of the running sums, each
        // <= 2^-24 |cam.x|; with cam.z > 0.05 that moves x/z by IB * 2^-24 * max|cam.x| / 0.05 pixels, which must stay below
        // 0.2 of the 0.25 px margin.  max|cam.x| over the volume is bounded per launch (IntegrateParams::cam_bound, from the
        // intrinsics and the camera position): where the bound fails - huge intrinsics or volumes - no batch is rejected
        // on its end voxels and every voxel takes the reference tests.
        const bool batch_ok = (float)IB * 5.9604645e-8f * P.cam_bound < 0.2f * 0.05f;
        const float wlim = (float)W - 0.4f, hlim = (float)H - 0.4f;
        const float wlim_b = (float)W - 0.25f, hlim_b = (float)H - 0.25f;
        const float ptol = 2e-6f * (float)(W > H ? W : H);
        for (unsigned zb = chunk_out ? z1 : z0; zb < z1; zb += IB, cell += (size_t)IB * plane) {
            const unsigned nb = (z1 - zb) < (unsigned)IB ? (z1 - zb) : (unsigned)IB;
            const ColumnState first = S;
            ColumnState last;
            // advance over the batch; `last` is the state of its final voxel, S the state of the next batch's first
            if (nb == (unsigned)IB) {   // whole batch (all but possibly the last of a chunk): straight-line packed adds
#pragma unroll
                for (int i = 0; i < IB - 1; ++i) { march_step<KZ>(S, da, db, dc); }
                last = S;
                march_step<KZ>(S, da, db, dc);
            } else {
                last = S;
                for (unsigned i = 0; i < nb; ++i) {
                    last = S;
                    march_step<KZ>(S, da, db, dc);
                }
            }
            {
                const float z_f = first.b.x, z_l = last.b.x;          // pos.z
                const float cz_f = cam_z<KZ>(first), cz_l = cam_z<KZ>(last);
                const bool behind = z_f < 0.00005f && z_l < 0.00005f;
                bool outside = false;
                if (batch_ok && cz_f > 0.05f && cz_l > 0.05f)
                    outside = (outside_bits(first.b.y, first.cy, cz_f, -0.75f, wlim_b, hlim_b) &
                               outside_bits(last.b.y, last.cy, cz_l, -0.75f, wlim_b, hlim_b)) != 0u;
                if (behind || outside || KFH_INT_PROBE(1)) continue;
            }
            // detailed pass over the batch, replaying the additions from its first voxel
            ColumnState T = first;
            unsigned live = 0;            // bit i: voxel zb+i projects into the image
            unsigned pix[IB];
#pragma unroll
            for (int i = 0; i < IB; ++i) {
                if ((unsigned)i < nb) {
                    const float posz = T.b.x, cx = T.b.y, cy = T.cy, cz = cam_z<KZ>(T);
                    if (!(posz < 0.0001f)) {                                       // cpp:647
                        const bool far_out = cz > 0.f && outside_bits(cx, cy, cz, -0.6f, wlim, hlim) != 0u;
                        if (!far_out) {
                            // cpp:649-654 needs only the comparisons with 0 / W-1 / H-1 and the integer part of
                            // fl(fl(c/z) + 0.5).  c * rcp(z) is within 3 ulp of fl(c/z), i.e. within 3.6e-7 * max(W,H)
                            // pixels wherever those outcomes are undecided; unless it lands within `ptol` (5x that) of
                            // an integer it gives the same decisions and the same integer.  The IEEE divisions run
                            // only for those rare pixels.
                            const float rz = __builtin_amdgcn_rcpf(cz);
                            float px = cx * rz + 0.5f, py = cy * rz + 0.5f;
                            const float nx = rintf(px), ny = rintf(py);
                            if (fabsf(px - nx) < ptol || fabsf(py - ny) < ptol || !(fabsf(px) < 65536.f) || !(fabsf(py) < 65536.f)) {
                                px = cx / cz + 0.5f;
                                py = cy / cz + 0.5f;
                            }
                            if (!(px < 0 || px > wm1 || py < 0 || py > hm1)) {     // cpp:651-653
                                live |= 1u << i;
                                pix[i] = (unsigned)px + (unsigned)py * W;
                            }
                        }
                    }
                    march_step<KZ>(T, da, db, dc);
                }
            }
            ++work;
            if (!live || KFH_INT_PROBE(2)) continue;
            float d[IB];
#pragma unroll
            for (int i = 0; i < IB; ++i)
                if (live & (1u << i)) d[i] = depth[pix[i]];
            // t = depth - cam.z: <= -mu -> rejected by cpp:662; >= mu -> sdf == 1; in between -> full formula
            unsigned need = 0;
            float t_[IB];
            float camz_i = cam_z<KZ>(first);  // cam.z of voxel i, replayed like the positions below
            const float dz_step = KZ ? db.x : dc.y;
#pragma unroll
            for (int i = 0; i < IB; ++i, camz_i += dz_step) {
                if (!(live & (1u << i))) continue;
                if (d[i] == 0) continue;                                           // cpp:655
                const float t = d[i] - camz_i;
                if (t <= -mu) continue;
                t_[i] = t;
                need |= 1u << i;
            }
            if (!need || KFH_INT_PROBE(4)) continue;
            work += 4;
            short2 c[IB];
#pragma unroll
            for (int i = 0; i < IB; ++i)
                if (need & (1u << i)) c[i] = cell[(size_t)i * plane];
            // the world position of a band voxel is replayed from the batch's first state (3 packed adds per voxel) instead
            // of being kept in 24 registers across the two memory round trips above
            ColumnState R = first;
#pragma unroll
            for (int i = 0; i < IB; ++i, R.a += da, R.b += db) {
                if (!(need & (1u << i))) continue;
                float sdf;
                if (t_[i] >= mu) {
                    sdf = 1.f;
                } else {
                    const float rx = R.a.x, ry = R.a.y, rz = R.b.x;
                    const float diff = t_[i] * sqrtf(1 + sq(rx / rz) + sq(ry / rz));
                    if (!(diff > -mu)) continue;
                    sdf = fmin_h(1.f, diff / mu);
                }
                float tsdf = c[i].x * kShort2Float, w = c[i].y;
                tsdf = clamp_h((w * tsdf + sdf) / (w + 1), -1.f, 1.f);
                w = fmin_h(w + 1, maxweight);
                const short ts = (short)(tsdf * 32766.0f);
                cell[(size_t)i * plane] = make_short2(ts, (short)w);
                if (skip_flags && ts < kSkipShortFwd && c[i].x >= kSkipShortFwd)   // flags are sticky: first crossing only
                    mark_bricks(skip_flags, (sx + 7) >> 3, (sy + 7) >> 3, (sz + 7) >> 3, x, y, zb + i);
                ++updated;
            }
        }
    }
    // two of three waves examined nothing (their chunks lie outside the frustum): nothing to add to the cost table or the
    // counter, so they leave here instead of running two six-step wave reductions over zeros
    if (!trace && __builtin_amdgcn_ballot_w64((work | updated) != 0u) == 0ull) return;
    if (cost || trace) {   // cost of the block = the busiest lane of its waves: batches examined + 4 x batches updated
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) work = umax_(work, (unsigned)__shfl_xor((int)work, off, 64));
        if (cost && threadIdx.x == 0 && work) atomicMax(cost + vb, work);
    }
    if (trace && threadIdx.x == 0) {   // debugging (KFH_INTEGRATE_TRACE): per wave {start, end, cost, block}
        const unsigned wave = slot * 4 + threadIdx.y;
        trace[4 * wave] = t_start; trace[4 * wave + 1] = wall_clock64(); trace[4 * wave + 2] = work; trace[4 * wave + 3] = vb;
    }
    // instrumentation (U = voxels that passed all four tests): one atomic per wave
    if (updated_total) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) updated += __shfl_down(updated, off, 64);
        if ((threadIdx.x & 63) == 0 && updated) atomicAdd(updated_total, (unsigned long long)updated);
    }
}

// =====================================================================================
// Volume sampling (commons.h:191-301) and the ray march (cpp:674-725)
// =====================================================================================
// Division by a launch-wide constant (the volume dimension in commons.h:193-195) without the 11-instruction IEEE
// sequence: with r = RN(1/d), q = a*r followed by one residual correction (e = fma(-d, q, a); q = fma(e, r, q)) is the
// correctly rounded quotient for every numerator the march can produce.  That is not taken on trust: the first time a
// divisor is seen, k_verify_div compares the short form with the IEEE quotient for EVERY float in [2^-100, 2^24) - 1.04e9
// values; negative numerators follow by symmetry, smaller ones give +-tiny either way, which the following "- 0.5f"
// absorbs, and pos*size stays below 2^24 for every volume the launchers accept - and one mismatch keeps the IEEE division
// for that volume (template flag FD of the view).
__global__ __launch_bounds__(256) void k_verify_div(float d, float r, unsigned long long *mismatches) {
    const unsigned long long n = 124ull << 23;   // biased exponents 27..150
    unsigned bad = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        const float a = __builtin_bit_cast(float, (27u << 23) + (unsigned)i);
        bad += (__builtin_bit_cast(unsigned, a / d) != __builtin_bit_cast(unsigned, div_by_const(a, d, r))) ? 1u : 0u;
    }
    if (bad) atomicAdd(mismatches, (unsigned long long)bad);
}

// sqrtf(d) > t against d > bound for every non-negative float bit pattern (see sqrt_gt_bound in kf_math.h)
__global__ __launch_bounds__(256) void k_verify_sqrt_bound(float t, float bound, unsigned long long *mismatches) {
    const unsigned long long n = 1ull << 31;
    unsigned bad = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        const float d = __builtin_bit_cast(float, (unsigned)i);
        bad += ((sqrtf(d) > t) != (d > bound)) ? 1u : 0u;
    }
    if (bad) atomicAdd(mismatches, (unsigned long long)bad);
}

template <int FD>
struct VolViewT {
    static constexpr int kFastDiv = FD;
    const short2 *data;
    int sx, sy, sz;
    f3 dim;
    f3 rdim;       // RN(1 / dim), read only when FD = 1
    size_t plane;
};

// Voxel addressing in 32 bits: the launchers refuse volumes above 2^30 voxels, so the BYTE offset of a voxel fits an
// unsigned 32-bit register and the load can use the scalar-base + 32-bit vector-offset form (two integer mads per tap
// instead of two 64-bit multiply-adds).
template <class V>
__device__ __forceinline__ unsigned voxel_byte_offset(const V &v, int x, int y, int z) {
    // 24-bit multiplies (full rate; v_mul_lo_u32 is a quarter-rate instruction): coordinates, sx and the plane size
    // sx*sy are below 2^24 for every volume the context accepts (<= 1024^3), the products below 2^30
    return ((unsigned)x + __umul24((unsigned)y, (unsigned)v.sx) + __umul24((unsigned)z, (unsigned)v.plane)) * 4u;
}

template <class V>
__device__ __forceinline__ float tap(const V &v, int x, int y, int z) {
    const short2 *p = reinterpret_cast<const short2 *>(reinterpret_cast<const char *>(v.data) + voxel_byte_offset(v, x, y, z));
    return (float)p->x;   // commons.h:172-174
}

struct Taps { int lo[3], up[3]; f3 fr; int base[3]; };

template <class V>
__device__ __forceinline__ Taps locate(const V &v, f3 pos) {
    Taps t;
    float sp0, sp1, sp2;
    if constexpr (V::kFastDiv) {
        sp0 = div_by_const(pos.x * v.sx, v.dim.x, v.rdim.x) - 0.5f;
        sp1 = div_by_const(pos.y * v.sy, v.dim.y, v.rdim.y) - 0.5f;
        sp2 = div_by_const(pos.z * v.sz, v.dim.z, v.rdim.z) - 0.5f;
    } else {
        sp0 = (pos.x * v.sx / v.dim.x) - 0.5f; sp1 = (pos.y * v.sy / v.dim.y) - 0.5f; sp2 = (pos.z * v.sz / v.dim.z) - 0.5f;
    }
    const float f0 = floorf(sp0), f1 = floorf(sp1), f2 = floorf(sp2);
    t.base[0] = (int)f0; t.base[1] = (int)f1; t.base[2] = (int)f2;
    t.fr = mk3(sp0 - f0, sp1 - f1, sp2 - f2);
    const int s[3] = {v.sx, v.sy, v.sz};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        // lower = max(base, 0), upper = min(base + 1, size - 1) (commons.h:198-200); the outer clamps only
        // act where the reference would index out of bounds
        t.lo[a] = imin(imax(t.base[a], 0), s[a] - 1);
        t.up[a] = imax(imin(t.base[a] + 1, s[a] - 1), 0);
    }
    return t;
}

// Two x-adjacent voxels in one 8-byte load.  The trilinear taps come in x-pairs (lo.x, up.x) with
// up.x in {lo.x, lo.x + 1}; p = min(lo.x, sx - 2) makes [p, p + 1] cover both without leaving the row, so the
// eight 2-byte gathers of an interpolation become four 8-byte ones (the ray march is bound by the texture
// address path, not by bytes).  The values are the same shorts, so the result is bit-identical.
struct __attribute__((packed, aligned(4))) VoxelPair { int a, b; };

template <class V>
__device__ __forceinline__ void tap_pair(const V &v, int p, int y, int z, bool lo_first, bool up_first, float &vlo,
                                         float &vup) {
    const VoxelPair q = *reinterpret_cast<const VoxelPair *>(reinterpret_cast<const char *>(v.data) + voxel_byte_offset(v, p, y, z));
    const float first = (float)(short)q.a, second = (float)(short)q.b;   // .x is the low half of each short2
    vlo = lo_first ? first : second;
    vup = up_first ? first : second;
}

template <class V>
__device__ __forceinline__ float interp_taps(const V &v, const Taps &t) {
    const float fx = t.fr.x, fy = t.fr.y, fz = t.fr.z;
    const int p = imax(imin(t.lo[0], v.sx - 2), 0);
    const bool lf = t.lo[0] == p, uf = t.up[0] == p;
    float a0, a1, b0, b1, c0, c1, d0, d1;
    tap_pair(v, p, t.lo[1], t.lo[2], lf, uf, a0, a1);
    tap_pair(v, p, t.up[1], t.lo[2], lf, uf, b0, b1);
    tap_pair(v, p, t.lo[1], t.up[2], lf, uf, c0, c1);
    tap_pair(v, p, t.up[1], t.up[2], lf, uf, d0, d1);
    return (((a0 * (1 - fx) + a1 * fx) * (1 - fy) + (b0 * (1 - fx) + b1 * fx) * fy) * (1 - fz)
            + ((c0 * (1 - fx) + c1 * fx) * (1 - fy) + (d0 * (1 - fx) + d1 * fx) * fy) * fz)
           * kShort2Float;
}

template <class V>
__device__ __forceinline__ float interp(const V &v, f3 pos) { return interp_taps(v, locate(v, pos)); }

// Central differences blended trilinearly (commons.h:215-301).  For axis A the four planes are
// ll = max(base-1,0), lo = max(base,0), up = min(base+1,size-1), uu = min(base+2,size-1); the low
// difference is v(up)-v(ll) and the high one v(uu)-v(lo).
template <class V>
__device__ __forceinline__ f3 grad(const V &v, f3 pos) {
    const Taps t = locate(v, pos);
    const int s[3] = {v.sx, v.sy, v.sz};
    int ll[3], uu[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        ll[a] = imin(imax(t.base[a] - 1, 0), s[a] - 1);
        uu[a] = imax(imin(t.base[a] + 2, s[a] - 1), 0);
    }
    const int *lo = t.lo, *up = t.up;
    const float fx = t.fr.x, fy = t.fr.y, fz = t.fr.z;
    f3 g;
    g.x = (((tap(v, up[0], lo[1], lo[2]) - tap(v, ll[0], lo[1], lo[2])) * (1 - fx)
            + (tap(v, uu[0], lo[1], lo[2]) - tap(v, lo[0], lo[1], lo[2])) * fx) * (1 - fy)
           + ((tap(v, up[0], up[1], lo[2]) - tap(v, ll[0], up[1], lo[2])) * (1 - fx)
              + (tap(v, uu[0], up[1], lo[2]) - tap(v, lo[0], up[1], lo[2])) * fx) * fy) * (1 - fz)
          + (((tap(v, up[0], lo[1], up[2]) - tap(v, ll[0], lo[1], up[2])) * (1 - fx)
              + (tap(v, uu[0], lo[1], up[2]) - tap(v, lo[0], lo[1], up[2])) * fx) * (1 - fy)
             + ((tap(v, up[0], up[1], up[2]) - tap(v, ll[0], up[1], up[2])) * (1 - fx)
                + (tap(v, uu[0], up[1], up[2]) - tap(v, lo[0], up[1], up[2])) * fx) * fy) * fz;
    g.y = (((tap(v, lo[0], up[1], lo[2]) - tap(v, lo[0], ll[1], lo[2])) * (1 - fx)
            + (tap(v, up[0], up[1], lo[2]) - tap(v, up[0], ll[1], lo[2])) * fx) * (1 - fy)
           + ((tap(v, lo[0], uu[1], lo[2]) - tap(v, lo[0], lo[1], lo[2])) * (1 - fx)
              + (tap(v, up[0], uu[1], lo[2]) - tap(v, up[0], lo[1], lo[2])) * fx) * fy) * (1 - fz)
          + (((tap(v, lo[0], up[1], up[2]) - tap(v, lo[0], ll[1], up[2])) * (1 - fx)
              + (tap(v, up[0], up[1], up[2]) - tap(v, up[0], ll[1], up[2])) * fx) * (1 - fy)
             + ((tap(v, lo[0], uu[1], up[2]) - tap(v, lo[0], lo[1], up[2])) * (1 - fx)
                + (tap(v, up[0], uu[1], up[2]) - tap(v, up[0], lo[1], up[2])) * fx) * fy) * fz;
    g.z = (((tap(v, lo[0], lo[1], up[2]) - tap(v, lo[0], lo[1], ll[2])) * (1 - fx)
            + (tap(v, up[0], lo[1], up[2]) - tap(v, up[0], lo[1], ll[2])) * fx) * (1 - fy)
           + ((tap(v, lo[0], up[1], up[2]) - tap(v, lo[0], up[1], ll[2])) * (1 - fx)
              + (tap(v, up[0], up[1], up[2]) - tap(v, up[0], up[1], ll[2])) * fx) * fy) * (1 - fz)
          + (((tap(v, lo[0], lo[1], uu[2]) - tap(v, lo[0], lo[1], lo[2])) * (1 - fx)
              + (tap(v, up[0], lo[1], uu[2]) - tap(v, up[0], lo[1], lo[2])) * fx) * (1 - fy)
             + ((tap(v, lo[0], up[1], uu[2]) - tap(v, lo[0], up[1], lo[2])) * (1 - fx)
                + (tap(v, up[0], up[1], uu[2]) - tap(v, up[0], up[1], lo[2])) * fx) * fy) * fz;
    const f3 cell = mk3(v.dim.x / v.sx, v.dim.y / v.sy, v.dim.z / v.sz);
    return scale3(mul3(g, cell), 0.5f * kShort2Float);
}

// The same gradient computed by the four adjacent lanes that marched the ray together (k_raycast<4>): each component is
// a trilinear blend of four x-blended difference pairs, one per (y, z) corner - lane q = jy + 2*jz evaluates its corner's
// pair (4 taps per component instead of 16), and the blend over y then z is two exchange-and-add steps that perform
// exactly the additions of the nested expression above (a*(1-f) + b*f with the operands swapped on the odd lane, which
// IEEE addition does not notice).  All four lanes return the full vector.  12 gathers per lane instead of 32 on one lane
// with the other three idle.
template <class V>
__device__ __forceinline__ f3 grad_quad(const V &v, f3 pos, int q) {
    const Taps t = locate(v, pos);
    const int jy = q & 1, jz = (q >> 1) & 1;
    const int *lo = t.lo, *up = t.up;
    const int ll0 = imin(imax(t.base[0] - 1, 0), v.sx - 1), uu0 = imax(imin(t.base[0] + 2, v.sx - 1), 0);
    const int ll1 = imin(imax(t.base[1] - 1, 0), v.sy - 1), uu1 = imax(imin(t.base[1] + 2, v.sy - 1), 0);
    const int ll2 = imin(imax(t.base[2] - 1, 0), v.sz - 1), uu2 = imax(imin(t.base[2] + 2, v.sz - 1), 0);
    const float fx = t.fr.x, fy = t.fr.y, fz = t.fr.z;
    const float wy = jy ? fy : 1 - fy, wz = jz ? fz : 1 - fz;
    const int Y = jy ? up[1] : lo[1], Z = jz ? up[2] : lo[2];
    // x: differences along x at corner (Y, Z)
    float gx = (tap(v, up[0], Y, Z) - tap(v, ll0, Y, Z)) * (1 - fx) + (tap(v, uu0, Y, Z) - tap(v, lo[0], Y, Z)) * fx;
    // y: low pair (up1 - ll1) on jy = 0, high pair (uu1 - lo1) on jy = 1, in plane Z
    const int Yp = jy ? uu1 : up[1], Ym = jy ? lo[1] : ll1;
    float gy = (tap(v, lo[0], Yp, Z) - tap(v, lo[0], Ym, Z)) * (1 - fx) + (tap(v, up[0], Yp, Z) - tap(v, up[0], Ym, Z)) * fx;
    // z: low pair (up2 - ll2) on jz = 0, high pair (uu2 - lo2) on jz = 1, in row Y
    const int Zp = jz ? uu2 : up[2], Zm = jz ? lo[2] : ll2;
    float gz = (tap(v, lo[0], Y, Zp) - tap(v, lo[0], Y, Zm)) * (1 - fx) + (tap(v, up[0], Y, Zp) - tap(v, up[0], Y, Zm)) * fx;
    gx *= wy; gy *= wy; gz *= wy;
    gx += __shfl_xor(gx, 1, 64); gy += __shfl_xor(gy, 1, 64); gz += __shfl_xor(gz, 1, 64);
    gx *= wz; gy *= wz; gz *= wz;
    gx += __shfl_xor(gx, 2, 64); gy += __shfl_xor(gy, 2, 64); gz += __shfl_xor(gz, 2, 64);
    const f3 cell = mk3(v.dim.x / v.sx, v.dim.y / v.sy, v.dim.z / v.sz);
    return scale3(mul3(mk3(gx, gy, gz), cell), 0.5f * kShort2Float);
}

template <int Q, class V>
__device__ __forceinline__ f3 grad_group(const V &v, f3 pos) {
    if constexpr (Q == 4) return grad_quad(v, pos, threadIdx.x & 3);
    else return grad(v, pos);
}

// ---- exact empty-space skipping -------------------------------------------------------------------------
// One byte per 8x8x8 brick, set by integrate when it stores a TSDF below kSkipShort (0.8118) into any voxel of the
// brick dilated by (-1, +9) voxels per axis.  A march sample whose (approximately located) base voxel lies in an
// unmarked brick has all eight trilinear taps >= kSkipShort, hence a value >= 0.81 > 0.8: the reference loop
// (cpp:709-716) would neither break nor shorten its step there, and the value itself is only needed if the very next
// sample is negative - in which case it is evaluated after the fact.  The march therefore takes exactly the steps the
// CPU takes and returns the same bits, but most free-space samples cost a flag lookup instead of 150 instructions and
// four gathers.  The +-1 dilation absorbs the rounding difference between pos*size/dim (exact path) and pos*(size/dim).
constexpr short kSkipShort = kSkipShortFwd;

struct SkipGrid {
    const unsigned char *flags;   // nullptr = no skipping
    int nbx, nby, nbz;
    float kx, ky, kz;             // size / dim per axis, divided by 8 (voxels -> bricks)
    float bx_max, by_max, bz_max; // last brick index per axis, as floats
    int two_phase;                // loop shape (only with flags)
    unsigned long long *trace;    // debugging (KFH_RAY_TRACE): per wave {start, end} of the 100 MHz wall clock, else nullptr
    const unsigned *order;        // launch slot -> screen tile (most expensive tiles of the last frame first), or nullptr
    unsigned tiles_x, tiles_y;    // the tile grid of the dispatch order: tiles of 2x2 wave tiles, whatever the workgroup size
    unsigned *cost;               // per screen tile: march rounds of its longest ray, for the next frame's order
    // the NEXT queued frame's OP_TRACK_BEGIN (scalar glue, ~3 us as a launch of its own), done by one thread of the last -
    // cheapest - slot while the rest of the launch marches; nothing in this launch reads what it writes
    DevState *next_state;         // nullptr = none
    float next_k[4];
    unsigned next_frame, next_tracking_rate;
};

template <class V>
__device__ __forceinline__ unsigned char brick_flag(const SkipGrid &g, const V &v, f3 pos) {
    // brick of the base voxel: floor((pos * size/dim - 0.5) / 8), clamped to the grid - as three instructions per axis
    // (fused multiply-add, median-of-three clamp in float, truncation: the clamped value is >= 0).  The base voxel is
    // only located to within one voxel here, which is what the dilation of the flags allows for.
    const unsigned bx = (unsigned)(int)__builtin_amdgcn_fmed3f(__builtin_fmaf(pos.x, g.kx, -0.0625f), 0.f, g.bx_max);
    const unsigned by = (unsigned)(int)__builtin_amdgcn_fmed3f(__builtin_fmaf(pos.y, g.ky, -0.0625f), 0.f, g.by_max);
    const unsigned bz = (unsigned)(int)__builtin_amdgcn_fmed3f(__builtin_fmaf(pos.z, g.kz, -0.0625f), 0.f, g.bz_max);
    (void)v;
    return g.flags[__umul24(__umul24(bz, (unsigned)g.nby) + by, (unsigned)g.nbx) + bx];
}
template <class V>
__device__ __forceinline__ bool brick_is_empty(const SkipGrid &g, const V &v, f3 pos) { return brick_flag(g, v, pos) == 0; }

__device__ __forceinline__ void mark_bricks(unsigned char *flags, int nbx, int nby, int nbz, int x, int y, int z) {
    const int x0 = x >= 2 ? (x - 2) >> 3 : 0, x1 = imin((x + 1) >> 3, nbx - 1);
    const int y0 = y >= 2 ? (y - 2) >> 3 : 0, y1 = imin((y + 1) >> 3, nby - 1);
    const int z0 = z >= 2 ? (z - 2) >> 3 : 0, z1 = imin((z + 1) >> 3, nbz - 1);
    for (int bz = z0; bz <= z1; ++bz)
        for (int by = y0; by <= y1; ++by)
            for (int bx = x0; bx <= x1; ++bx) flags[bx + nbx * (by + nby * bz)] = 1;
}

// full rebuild (after reset or when a volume is injected from the host)
__global__ __launch_bounds__(256) void k_build_skip_grid(unsigned char *__restrict__ flags, const short2 *__restrict__ vol,
                                                         unsigned sx, unsigned sy, unsigned sz, int nbx, int nby, int nbz) {
    const size_t n = (size_t)sx * sy * sz;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (vol[i].x < kSkipShort) {
            const unsigned x = i % sx, y = (i / sx) % sy, z = i / ((size_t)sx * sy);
            mark_bricks(flags, nbx, nby, nbz, x, y, z);
        }
    }
}

template <class V>
__device__ __forceinline__ float4 cast_ray(const V &v, const SkipGrid &sg, unsigned px, unsigned py, const M4 &view,
                                           float nearPlane, float farPlane, float step, float largestep, int &work) {
    const f3 origin = mk3(view.m[0][3], view.m[1][3], view.m[2][3]);
    const f3 dir = rotate(view, mk3((float)px, (float)py, 1.f));
    const f3 invR = mk3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
    const f3 tbot = mul3(scale3(invR, -1.0f), origin);
    const f3 ttop = mul3(invR, sub3(v.dim, origin));
    const f3 tmin = mk3(fmin_h(ttop.x, tbot.x), fmin_h(ttop.y, tbot.y), fmin_h(ttop.z, tbot.z));
    const f3 tmax = mk3(fmax_h(ttop.x, tbot.x), fmax_h(ttop.y, tbot.y), fmax_h(ttop.z, tbot.z));
    const float largest_tmin = fmax_h(fmax_h(tmin.x, tmin.y), fmax_h(tmin.x, tmin.z));
    const float smallest_tmax = fmin_h(fmin_h(tmax.x, tmax.y), fmin_h(tmax.x, tmax.z));
    const float tnear = fmax_h(largest_tmin, nearPlane);
    const float tfar = fmin_h(smallest_tmax, farPlane);
    if (tnear < tfar) {
        float t = tnear, stepsize = largestep;
        const f3 p0 = add3(origin, scale3(dir, t));
        // `lazy`: the last accepted sample (f_t) lies in an empty brick; it is known to be >= 0.81 but not evaluated
        bool lazy = sg.flags && brick_is_empty(sg, v, p0);
        float t_ft = t;
        float f_t = lazy ? 1.0f : interp(v, p0);
        float f_tt = 0;
        if (f_t > 0) {
            if (sg.two_phase) {
                // Same visiting order as `for (; t < tfar; t += stepsize)`, written as two phases so that the lanes of
                // a wave first run through their free-space samples in a short loop and then take the expensive
                // sample together.
                while (true) {
                    while (t < tfar && brick_is_empty(sg, v, add3(origin, scale3(dir, t)))) {
                        lazy = true;
                        t_ft = t;
                        t += stepsize;
                    }
                    if (!(t < tfar)) break;
                    f_tt = interp(v, add3(origin, scale3(dir, t)));
                    if (f_tt < 0) break;
                    if (f_tt < 0.8f) stepsize = step;
                    f_t = f_tt;
                    lazy = false;
                    t += stepsize;
                }
            } else {
                for (; t < tfar; t += stepsize) {
                    ++work;
                    const f3 p = add3(origin, scale3(dir, t));
                    if (sg.flags && brick_is_empty(sg, v, p)) { lazy = true; t_ft = t; continue; }
                    f_tt = interp(v, p);
                    if (f_tt < 0) break;
                    if (f_tt < 0.8f) stepsize = step;
                    f_t = f_tt;
                    lazy = false;
                }
            }
            if (f_tt < 0) {
                if (lazy) f_t = interp(v, add3(origin, scale3(dir, t_ft)));
                t = t + stepsize * f_tt / (f_t - f_tt);
                const f3 p = add3(origin, scale3(dir, t));
                return make_float4(p.x, p.y, p.z, t);
            }
        }
    }
    return make_float4(0.f, 0.f, 0.f, 0.f);
}

__device__ __forceinline__ unsigned umin_(unsigned a, unsigned b) { return a < b ? a : b; }

// ---- group-cooperative ray march --------------------------------------------------------------------------
// The kernel time of the one-ray-per-lane march is the longest ray's serial chain (measured: mean 36 samples per
// ray, longest tile 136; 0.8 us per dependent sample => ~110 us, with most SIMDs idle after ~35 us).  Here Q
// adjacent lanes march ONE ray together: the sample times of a round are t, t+s, t+2s, ... (each lane adds the step q
// times, so the float sequence is the serial one), the Q samples are evaluated in parallel, and the first lane at which
// the reference loop would do something other than "continue" (leave the volume, cross zero, or switch from the large to
// the small step, cpp:709-716) decides the ray's next state.  Samples past that lane are discarded.  The chain per ray
// shrinks Q-fold and the frame exposes Q times more waves; decisions and bits are those of the serial loop.
// value of the group's first / last lane: for four lanes per ray a quad permute of the data path (DPP), no LDS round trip
template <int Q>
__device__ __forceinline__ int group_first(int x, int base) {
    if constexpr (Q == 4) return __builtin_amdgcn_mov_dpp(x, 0x00, 0xF, 0xF, true);
    else return __shfl(x, base, 64);
}
template <int Q>
__device__ __forceinline__ int group_last(int x, int base) {
    if constexpr (Q == 4) return __builtin_amdgcn_mov_dpp(x, 0xFF, 0xF, 0xF, true);
    else return __shfl(x, base + Q - 1, 64);
}
template <int Q> __device__ __forceinline__ float group_first(float x, int base) { return __builtin_bit_cast(float, group_first<Q>(__builtin_bit_cast(int, x), base)); }
template <int Q> __device__ __forceinline__ float group_last(float x, int base) { return __builtin_bit_cast(float, group_last<Q>(__builtin_bit_cast(int, x), base)); }

template <int Q, class V>
__device__ __forceinline__ float4 cast_ray_group(const V &v, const SkipGrid &sg, unsigned px, unsigned py,
                                                 const M4 &view, float nearPlane, float farPlane, float step,
                                                 float largestep, int &work) {
    const int lane = threadIdx.x & 63;
    const int q = lane & (Q - 1), base = lane & ~(Q - 1);
    const f3 origin = mk3(view.m[0][3], view.m[1][3], view.m[2][3]);
    const f3 dir = rotate(view, mk3((float)px, (float)py, 1.f));
    const f3 invR = mk3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
    const f3 tbot = mul3(scale3(invR, -1.0f), origin);
    const f3 ttop = mul3(invR, sub3(v.dim, origin));
    const f3 tmin = mk3(fmin_h(ttop.x, tbot.x), fmin_h(ttop.y, tbot.y), fmin_h(ttop.z, tbot.z));
    const f3 tmax = mk3(fmax_h(ttop.x, tbot.x), fmax_h(ttop.y, tbot.y), fmax_h(ttop.z, tbot.z));
    const float largest_tmin = fmax_h(fmax_h(tmin.x, tmin.y), fmax_h(tmin.x, tmin.z));
    const float smallest_tmax = fmin_h(fmin_h(tmax.x, tmax.y), fmin_h(tmax.x, tmax.z));
    const float tnear = fmax_h(largest_tmin, nearPlane);
    const float tfar = fmin_h(smallest_tmax, farPlane);
    const float4 miss = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!(tnear < tfar)) return miss;
    float t = tnear, stepsize = largestep;
    bool small = false;          // stepsize has been switched to `step`
    bool lazy = false;           // f_t lies in an empty brick: >= 0.81, value not evaluated
    float f_t = 0.f, t_ft = t;
    bool first = true;
    while (true) {
        work += Q;
        float tq = t;
#pragma unroll
        for (int i = 0; i < Q - 1; ++i) tq = (i < q) ? tq + stepsize : tq;
        const bool in_range = tq < tfar;
        bool s_lazy = false;
        float f = 0.f;
        if (in_range) {
            const f3 p = add3(origin, scale3(dir, tq));
            if (sg.flags && brick_is_empty(sg, v, p)) s_lazy = true;
            else f = interp(v, p);
        }
        // what the serial loop does at this sample: 0 continue, 1 out of range, 2 zero crossing, 3 large->small step
        int kind = 0;
        if (!in_range) kind = 1;
        else if (!s_lazy && f < 0) kind = 2;
        else if (!s_lazy && !small && f < 0.8f) kind = 3;
        const unsigned long long ball = __ballot(kind != 0);
        const unsigned evmask = (unsigned)((ball >> base) & ((1ull << Q) - 1ull));
        if (first) {
            // cpp:706-708: the march only starts if the sample at tnear (lane 0 of this round) is positive
            first = false;
            const float f0 = group_first<Q>(f, base);
            const int lazy0 = group_first<Q>((int)s_lazy, base);
            const int in0 = group_first<Q>((int)in_range, base);
            if (in0 && !lazy0 && !(f0 > 0)) return miss;
        }
        if (evmask == 0) {   // all Q samples are plain "continue": f_t <- last of them, t <- its time + stepsize
            f_t = group_last<Q>(f, base);
            lazy = group_last<Q>((int)s_lazy, base) != 0;
            t_ft = group_last<Q>(tq, base);
            t = t_ft + stepsize;
            continue;
        }
        const int e = __ffs(evmask) - 1;
        if (e > 0) {
            f_t = __shfl(f, base + e - 1, 64);
            lazy = __shfl((int)s_lazy, base + e - 1, 64) != 0;
            t_ft = __shfl(tq, base + e - 1, 64);
        }
        const int ekind = __shfl(kind, base + e, 64);
        const float te = __shfl(tq, base + e, 64), fe = __shfl(f, base + e, 64);
        if (ekind == 1) return miss;                 // left the volume / far plane without a crossing
        if (ekind == 2) {                            // cpp:717-720
            if (lazy) f_t = interp(v, add3(origin, scale3(dir, t_ft)));
            const float th = te + stepsize * fe / (f_t - fe);
            const f3 p = add3(origin, scale3(dir, th));
            return make_float4(p.x, p.y, p.z, th);
        }
        // cpp:713-715: first sample below 0.8 -> switch to the small step from here on
        small = true;
        stepsize = step;
        f_t = fe;
        lazy = false;
        t = te + stepsize;
    }
}

// Pixel mapping shared by raycast and renderVolume.  A wave marches 64/Q rays laid out as a small screen tile (8x8, 8x4,
// 4x4, 4x2 for Q = 1, 2, 4, 8), the Q lanes of a ray being adjacent; a workgroup is KFH_RAY_BLOCK_WAVES of them (2x1 by default), a dispatch tile 2x2.
#ifndef KFH_RAY_BLOCK_WAVES
#define KFH_RAY_BLOCK_WAVES 2      // waves per ray-march workgroup: 4 (2x2 wave tiles), 2 (2x1) or 1; 2 packs the CUs best (profiles/r01_sweeps.md)
#endif
constexpr int RAY_WX = KFH_RAY_BLOCK_WAVES >= 2 ? 2 : 1, RAY_WY = KFH_RAY_BLOCK_WAVES / RAY_WX;
constexpr int RAY_BLOCK = 64 * KFH_RAY_BLOCK_WAVES;
template <int Q>
__device__ __forceinline__ void tile_pixel(unsigned &x, unsigned &y, unsigned bx, unsigned by) {
    constexpr int R = 64 / Q;                         // rays per wave
    constexpr int RW = (Q == 1 || Q == 2) ? 8 : 4;    // tile width
    constexpr int RH = R / RW;
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned r = lane / Q;
    x = (bx * RAY_WX + (wave % RAY_WX)) * RW + (r % RW);
    y = (by * RAY_WY + (wave / RAY_WX)) * RH + (r / RW);
}

template <int Q, class V>
__device__ __forceinline__ float4 march(const V &v, const SkipGrid &sg, unsigned x, unsigned y, const M4 &view,
                                        float nearPlane, float farPlane, float step, float largestep, int &work) {
    if (Q == 1) return cast_ray(v, sg, x, y, view, nearPlane, farPlane, step, largestep, work);
    return cast_ray_group<Q>(v, sg, x, y, view, nearPlane, farPlane, step, largestep, work);
}

// raycastKernel (cpp:726-757)
template <int Q, int FD>
__global__ __launch_bounds__(RAY_BLOCK) void k_raycast(float4 *__restrict__ vertex, float4 *__restrict__ normal, unsigned W,
                                                 unsigned H, const short2 *__restrict__ vol, unsigned sx, unsigned sy,
                                                 unsigned sz, f3 dim, const RayParams *__restrict__ rp, float nearPlane,
                                                 float farPlane, float step, float largestep, SkipGrid sg, f3 rdim) {
    const RayParams R = *rp;      // gate and view matrix in one request, the dispatch-order entry with them (see k_integrate)
    const unsigned long long t_start = sg.trace ? wall_clock64() : 0ull;
    // Launch slot -> screen tile.  Workgroups are dispatched in slot order and the launch ends with its slowest wave, so
    // the tiles whose rays marched longest in the previous frame (grazing views along a surface: up to 4x the mean) go
    // first instead of wherever the raster order puts them - often the last dispatch round (measured: 65 -> 50 us,
    // profiles/r01_sweeps.md).  The order also keeps the tiles of a 64x64-pixel super-tile on one XCD (ray_order_block).
    // It only decides WHEN and WHERE a tile is computed, never what.
    const unsigned slot = blockIdx.x + blockIdx.y * gridDim.x;
    // A dispatch tile is 2x2 wave tiles.  With workgroups smaller than that (KFH_RAY_BLOCK_WAVES < 4, which pack the CUs
    // better) a tile is PARTS workgroups whose slots are 8 apart, i.e. on the same XCD: slot = 8 * PARTS * j + 8 * part + class.
    constexpr unsigned PARTS = 4 / KFH_RAY_BLOCK_WAVES;
    const unsigned tslot = PARTS == 1 ? slot : (slot / (8 * PARTS)) * 8 + (slot % 8);
    const unsigned part = PARTS == 1 ? 0u : (slot / 8) % PARTS;
    const bool padding = tslot >= sg.tiles_x * sg.tiles_y;      // (the launch is padded to a multiple of 8 * PARTS slots)
    const unsigned tile = (sg.order && !padding) ? sg.order[tslot] : tslot;
    asm volatile("" : : "s"(R.view.m[0][0]), "s"(R.view.m[2][0]), "s"(tile));
    // (the riding preamble first: the host books it as done whenever this launch was enqueued, whatever the raycast gate says -
    // k_icp_tail's barrier counters are reset there)
    if (sg.next_state && slot == gridDim.x * gridDim.y - 1 && threadIdx.x == 0)
        track_begin(sg.next_state, (sg.next_frame % sg.next_tracking_rate) == 0, sg.next_k);
    if (!R.enabled) return;
    if (padding) return;
    unsigned x, y;
    {
        const unsigned tx = tile % sg.tiles_x, ty = tile / sg.tiles_x;
        // workgroup inside the tile: RAY_WX x RAY_WY wave tiles each, (2 / RAY_WX) x (2 / RAY_WY) of them
        const unsigned px_ = part % (2 / RAY_WX), py_ = part / (2 / RAY_WX);
        tile_pixel<Q>(x, y, tx * (2 / RAY_WX) + px_, ty * (2 / RAY_WY) + py_);
    }
    if (x >= W || y >= H) return;     // uniform per ray group: all Q lanes share the pixel
    const M4 view = R.view;
    VolViewT<FD> v;
    v.data = vol; v.sx = sx; v.sy = sy; v.sz = sz; v.dim = dim; v.rdim = rdim; v.plane = (size_t)sx * sy;
    int work = 0;
    const float4 hit = march<Q>(v, sg, x, y, view, nearPlane, farPlane, step, largestep, work);
    if (sg.cost || sg.trace) {   // the longest march of the wave, kept per tile for the next frame's dispatch order
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) work = imax(work, __shfl_xor(work, off, 64));
        // ... dilated over the eight neighbours here, by nine lanes of one atomic instruction, so that the sort reads one
        // number per tile: a silhouette ray that just misses an object marches four times longer than its neighbour that
        // hits it, and which tile holds it changes with every camera step
        const int lane = (int)(threadIdx.x & 63u);
        const int nx = (int)(tile % sg.tiles_x) + lane % 3 - 1, ny = (int)(tile / sg.tiles_x) + lane / 3 - 1;
        if (sg.cost && lane < 9 && nx >= 0 && ny >= 0 && nx < (int)sg.tiles_x && ny < (int)sg.tiles_y)
            atomicMax(sg.cost + nx + ny * (int)sg.tiles_x, (unsigned)work);
    }
    if (sg.trace && (threadIdx.x & 63) == 0) {
        const unsigned wave = slot * KFH_RAY_BLOCK_WAVES + (threadIdx.x >> 6);
        sg.trace[4 * wave] = t_start;
        sg.trace[4 * wave + 1] = wall_clock64();
        sg.trace[4 * wave + 2] = (unsigned long long)work;
        sg.trace[4 * wave + 3] = tile;
    }
    const bool writer = (threadIdx.x & (Q - 1)) == 0;   // one lane of the group finishes the pixel
    if (Q != 4 && !writer) return;
    const unsigned p = x + y * W;
    if (hit.w > 0.0) {                                  // uniform over the lanes of a ray
        const f3 n = grad_group<Q>(v, mk3(hit.x, hit.y, hit.z));
        if (!writer) return;
        vertex[p] = make_float4(hit.x, hit.y, hit.z, 0.f);
        if (length3(n) == 0) {
            reinterpret_cast<float *>(normal + p)[0] = kInvalid;
        } else {
            const f3 nn = normalize3(n);
            normal[p] = make_float4(nn.x, nn.y, nn.z, 0.f);
        }
    } else if (writer) {
        vertex[p] = make_float4(0.f, 0.f, 0.f, 0.f);
        normal[p] = make_float4(kInvalid, 0.f, 0.f, 0.f);
    }
}

// renderVolumeKernel (cpp:880-913)
template <int Q, int FD>
__global__ __launch_bounds__(RAY_BLOCK) void k_render_volume(uchar4 *__restrict__ out, unsigned W, unsigned H,
                                                       const short2 *__restrict__ vol, unsigned sx, unsigned sy,
                                                       unsigned sz, f3 dim, const RayParams *__restrict__ rp,
                                                       float nearPlane, float farPlane, float step, float largestep,
                                                       f3 light, f3 ambient, SkipGrid sg, f3 rdim) {
    unsigned x, y;
    tile_pixel<Q>(x, y, blockIdx.x, blockIdx.y);
    if (x >= W || y >= H) return;
    const M4 view = rp->view;
    VolViewT<FD> v;
    v.data = vol; v.sx = sx; v.sy = sy; v.sz = sz; v.dim = dim; v.rdim = rdim; v.plane = (size_t)sx * sy;
    int work = 0;
    const float4 hit = march<Q>(v, sg, x, y, view, nearPlane, farPlane, step, largestep, work);
    const bool writer = (threadIdx.x & (Q - 1)) == 0;
    if (Q != 4 && !writer) return;
    uchar4 o = make_uchar4(0, 0, 0, 0);
    if (hit.w > 0) {
        const f3 test = mk3(hit.x, hit.y, hit.z);
        const f3 n = grad_group<Q>(v, test);
        if (length3(n) > 0) {
            const f3 diff = normalize3(sub3(light, test));
            const float dir = fmax_h(dot3(normalize3(n), diff), 0.f);
            const f3 col = scale3(mk3(clamp_h(dir + ambient.x, 0.f, 1.f), clamp_h(dir + ambient.y, 0.f, 1.f),
                                      clamp_h(dir + ambient.z, 0.f, 1.f)), 255);
            o = make_uchar4((unsigned char)col.x, (unsigned char)col.y, (unsigned char)col.z, 0);
        }
    }
    if (writer) out[x + y * W] = o;
}

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

__global__ __launch_bounds__(256) void k_render_depth(uchar4 *__restrict__ out, const float *__restrict__ depth,
                                                      unsigned W, unsigned H, float nearPlane, float farPlane) {
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
__global__ __launch_bounds__(256) void k_render_track(uchar4 *__restrict__ out, const int *__restrict__ result,
                                                      unsigned W, unsigned H) {
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

// =====================================================================================
// launch wrappers
// =====================================================================================
namespace launch {

// Every launcher below notes a rejected launch (bad grid / block shape, too much LDS, ...) by kernel name; the pipeline asks
// after each phase it queues, so a failed launch becomes KFH_ERR_HIP of that call instead of stale buffers at the next getter.
static thread_local char g_launch_error[320] = "";
static inline void launched(const char *kernel) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess && !g_launch_error[0])
        snprintf(g_launch_error, sizeof g_launch_error, "launch of %s failed: %s", kernel, hipGetErrorString(e));
}
const char *take_launch_error() {
    static thread_local char out[320];
    if (!g_launch_error[0]) return nullptr;
    snprintf(out, sizeof out, "%s", g_launch_error);
    g_launch_error[0] = 0;
    return out;
}

static inline dim3 grid_rows(unsigned W, unsigned H) { return dim3((W + 63) / 64, (H + 3) / 4); }
static inline dim3 block_rows() { return dim3(64, 4); }

static unsigned track_block_cap() {
    static unsigned cap = 0;
    if (!cap) {
        // tuning knob.  Measured at 640x480 (profiles/r01_sweeps.md): 512 blocks (2.3 pixels per thread, 64 KB of
        // partials for k_solve to add up) beats one pixel per thread (1200 blocks) by 4-5 % of the frame time.
        const char *e = getenv("KFH_TRACK_BLOCKS");
        const long v = e ? atol(e) : 0;
        cap = (v >= 1 && v <= (long)TR_MAX_BLOCKS) ? (unsigned)v : 512;
    }
    return cap;
}

unsigned track_blocks(unsigned w, unsigned h) {
    const unsigned n = (w * h + TR_THREADS - 1) / TR_THREADS;
    const unsigned cap = track_block_cap();
    return n < 1 ? 1 : (n > cap ? cap : n);
}

unsigned track_blocks_max() { return TR_MAX_BLOCKS; }
unsigned icp_tail_unit() { return ICP_TAIL_BLOCKS; }

// Debugging aid shared by the traced kernels: <ENV>=<file> makes the 200th launch write four 64-bit words per wave
// ({start, end} on the 100 MHz wall clock and two kernel-specific values) and dumps them (profiles/ray_trace.py).
struct WaveTrace {
    const char *path = nullptr;
    int launches = 0;
    unsigned long long *buf = nullptr;
    size_t words = 0;
    hipStream_t stream = nullptr;
    explicit WaveTrace(const char *env) : path(getenv(env)) {}
    unsigned long long *arm(size_t waves, hipStream_t s) {     // device buffer for this launch, or nullptr
        if (!path || ++launches != 200) return nullptr;
        words = waves * 4; stream = s;
        if (hipMalloc(&buf, words * sizeof(unsigned long long)) != hipSuccess) { buf = nullptr; return nullptr; }
        hipMemsetAsync(buf, 0, words * sizeof(unsigned long long), s);
        return buf;
    }
    void dump() {
        if (!buf) return;
        std::vector<unsigned long long> h(words);
        hipStreamSynchronize(stream);
        hipMemcpy(h.data(), buf, words * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        if (FILE *f = fopen(path, "wb")) { fwrite(h.data(), sizeof(unsigned long long), words, f); fclose(f); }
        hipFree(buf);
        buf = nullptr;
    }
};

static bool divisor_verified(float d);     // (cache look-up, never a launch: see verified_divisor)

unsigned icp_blocks(unsigned w, unsigned h) {
    // default 256 blocks of 1024 threads (one per CU, ~1.2 pixels per thread at 640x480); KFH_ICP_BLOCKS overrides
    static const long cap_env = [] { const char *e = getenv("KFH_ICP_BLOCKS"); const long v = e ? atol(e) : 0; return (v >= 1 && v <= (long)TR_MAX_BLOCKS) ? v : 256L; }();
    const unsigned n = (w * h + ICP_THREADS - 1) / ICP_THREADS;
    return n < 1 ? 1 : (n > (unsigned)cap_env ? (unsigned)cap_env : n);
}

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

void track_reduce(float *partials, int *result_map, kfh_track_data *track_out, const float4 *inVertex,
                  const float4 *inNormal, unsigned inW, unsigned inH, const float4 *refVertex, const float4 *refNormal,
                  unsigned refW, unsigned refH, const M4 *pose, const TrackParams *tp, const int *skip,
                  float dist_threshold, float normal_threshold, hipStream_t s) {
    k_track_reduce<<<track_blocks(inW, inH), TR_THREADS, 0, s>>>(partials, result_map, track_out, inVertex, inNormal, inW,
                                                               inH, refVertex, refNormal, refW, refH, pose, tp, skip,
                                                               dist_threshold, normal_threshold); launched("k_track_reduce");
}

void reduce_trackdata(float *partials, const kfh_track_data *J, unsigned JW, unsigned W, unsigned H, hipStream_t s) {
    k_reduce_trackdata<<<track_blocks(W, H), TR_THREADS, 0, s>>>(partials, J, JW, W, H); launched("k_reduce_trackdata");
}

void solve(DevState *st, float *out32, const float *partials, unsigned nblocks, int level, float icp_threshold,
           int do_solve, hipStream_t s) {
    k_solve<<<1, 1024, 0, s>>>(st, out32, partials, nblocks, level, icp_threshold, do_solve); launched("k_solve");
}

void control(DevState *st, FrameLogEntry *log, int ops, const ControlArgs &a, hipStream_t s) {
    k_control<<<1, 64, 0, s>>>(st, log, ops, a); launched("k_control");
}

static IcpIterArgs make_icp_args(DevState *st, IcpState *iters, const float *partials_prev, float *partials_out, int *result_map,
                                 int it, int level, int prev_level, unsigned prev_nblocks, unsigned nblocks, const float4 *inV,
                                 const float4 *inN, unsigned inW, unsigned inH, const float4 *refV, const float4 *refN,
                                 unsigned refW, unsigned refH, float icp_threshold, float dist_threshold, float dist2_max, float normal_threshold) {
    IcpIterArgs a;
    a.st = st; a.iters = iters; a.partials_prev = partials_prev; a.partials_out = partials_out; a.result_map = result_map;
    a.it = it; a.level = level; a.prev_level = prev_level; a.prev_nblocks = prev_nblocks; a.nblocks = nblocks;
    a.inV = inV; a.inN = inN; a.inW = inW; a.inH = inH; a.refV = refV; a.refN = refN; a.refW = refW; a.refH = refH;
    a.icp_threshold = icp_threshold; a.dist_threshold = dist_threshold; a.normal_threshold = normal_threshold;
    a.dist2_max = dist2_max;      // (the context's verdict of prepare_sqrt_bound at kfh_create: no look-up on the frame path)
    // py = (unsigned)((p + 0.5f) * RN(1 / inW)) is p / inW while p + 0.5 and the two roundings leave more than the 0.5 / inW
    // between (p + 0.5) / inW and the next integer: certainly for p < 2^22
    a.inv_inW = ((unsigned long long)inW * inH <= (1ull << 22)) ? 1.0f / (float)inW : 0.f;
    a.ray_order = nullptr; a.ray_cost = nullptr; a.ray_tiles_x = a.ray_tiles_y = 0; a.order_mode = 0;
    a.trace = nullptr;
    return a;
}

// side blocks of a launch: one per tile team group, at most what the ICP blocks leave free on the chip
static unsigned side_blocks(unsigned tiles, unsigned icp_nb) {
    const unsigned room = icp_nb < 224 ? 256 - icp_nb : 32;     // one block per CU the ICP blocks leave idle
    return tiles < room ? (tiles ? tiles : 1) : room;
}

// rounds of tiles a side block of a launch with `icp_nb` ICP blocks works through for `tiles` pyramid tiles
unsigned pyramid_side_rounds(unsigned tiles, unsigned icp_nb) {
    const unsigned slots = side_blocks(tiles, icp_nb + 1) * PYR_SIDE_TEAMS;
    return (tiles + slots - 1) / slots;
}
unsigned bilateral_tiles(unsigned W, unsigned H) { return ((W + PBX - 1) / PBX) * ((H + PBY - 1) / PBY); }
unsigned pyramid_tiles(unsigned W, unsigned H) { return ((W + PYR_T - 1) / PYR_T) * ((H + PYR_T - 1) / PYR_T); }
bool pyramid_fusable(int n_levels, unsigned W, unsigned H) {
    return n_levels >= 1 && n_levels <= PYR_MAX_LEVELS && W % (1u << (n_levels - 1)) == 0 && H % (1u << (n_levels - 1)) == 0;
}

void icp_iter(DevState *st, IcpState *iters, const float *partials_prev, float *partials_out, int *result_map, int it,
              int level, int prev_level, unsigned prev_nblocks, unsigned nblocks, const float4 *inV, const float4 *inN,
              unsigned inW, unsigned inH, const float4 *refV, const float4 *refN, unsigned refW, unsigned refH,
              float icp_threshold, float dist_threshold, float dist2_max, float normal_threshold, const SideWork *side, hipStream_t s) {
    IcpIterArgs a = make_icp_args(st, iters, partials_prev, partials_out, result_map, it, level, prev_level, prev_nblocks,
                                        nblocks, inV, inN, inW, inH, refV, refN, refW, refH, icp_threshold, dist_threshold,
                                        dist2_max, normal_threshold);
    if (side && side->kind == SideWork::BILATERAL && side->count) {
        SideBilateral sb;
        sb.floatDepth = side->floatDepth; sb.out = side->filtered; sb.in = side->in; sb.W = side->W; sb.H = side->H;
        sb.inW = side->inW; sb.ratio = side->ratio; sb.g = side->gauss; sb.e_d_squared_2 = side->e_d * side->e_d * 2;
        const bool fd = divisor_verified(sb.e_d_squared_2);
        sb.rcp = fd ? 1.0f / sb.e_d_squared_2 : 0.f;
        sb.first = side->first; sb.count = side->count;
        unsigned extra = 0;
        if (side->ray_order) {   // the raycast dispatch sort shares the launch
            a.ray_order = side->ray_order; a.ray_cost = side->ray_cost; a.ray_tiles_x = side->ray_tiles_x; a.ray_tiles_y = side->ray_tiles_y; a.order_mode = side->order_mode;
            extra = 1;
        }
        const unsigned grid = nblocks + side_blocks(side->count, nblocks + extra) + extra;
        if (fd) k_icp_iter_bilateral<true><<<grid, ICP_THREADS, 0, s>>>(a, sb);
        else k_icp_iter_bilateral<false><<<grid, ICP_THREADS, 0, s>>>(a, sb);
        launched("k_icp_iter_bilateral");
        return;
    }
    if (side && side->kind == SideWork::PYRAMID && side->count) {
        SidePyramid sp;
        for (int l = 0; l < PYR_MAX_LEVELS; ++l) {
            const bool on = l < side->n_levels;
            sp.a.scaled[l] = on ? side->scaled[l] : nullptr;
            sp.a.vertex[l] = on ? side->vertex[l] : nullptr;
            sp.a.normal[l] = on ? side->normal[l] : nullptr;
            sp.a.stale_normal[l] = on ? side->stale_normal[l] : nullptr;
            sp.a.invK[l] = on ? side->invK[l] : identity4();
        }
        sp.a.W = side->W; sp.a.H = side->H; sp.a.e_d = side->e_d;
        static const int through_env = [] { const char *e = getenv("KFH_PYR_THROUGH"); return e ? atoi(e) : -1; }();     // 0 / 1: never / always
        sp.a.through = through_env >= 0 ? through_env == 1 : pyramid_side_rounds(side->count, nblocks) == 1;
        sp.first = side->first; sp.count = side->count;
        unsigned extra = 0;
        if (side->ray_order) {   // a dispatch sort shares the launch
            a.ray_order = side->ray_order; a.ray_cost = side->ray_cost; a.ray_tiles_x = side->ray_tiles_x; a.ray_tiles_y = side->ray_tiles_y; a.order_mode = side->order_mode;
            extra = 1;
        }
        const unsigned grid = nblocks + side_blocks(side->count, nblocks + extra) + extra;
        const int part = side->n_levels >= 2 ? side->pyr_part : 0;
        // level 0 alone: four pixels a thread with teams of 256; with teams of 512, if the side blocks still take all tiles in one
        // round, two (640x480: the launch 9.0 -> 8.65 us, profiles/r04_pyr_teams.txt)
        if (part == 1 && PYR_SIDE_TEAMS > 2 && side->count <= 2 * side_blocks(side->count, nblocks + extra)) k_icp_iter_pyramid<1, 1, 2><<<grid, ICP_THREADS, 0, s>>>(a, sp);
        else if (part == 1) k_icp_iter_pyramid<1, 1><<<grid, ICP_THREADS, 0, s>>>(a, sp);
        else if (part == 2) switch (side->n_levels) {
        case 2: k_icp_iter_pyramid<2, 2><<<grid, ICP_THREADS, 0, s>>>(a, sp); break;
        case 3: k_icp_iter_pyramid<3, 2><<<grid, ICP_THREADS, 0, s>>>(a, sp); break;
        default: k_icp_iter_pyramid<4, 2><<<grid, ICP_THREADS, 0, s>>>(a, sp); break;
        }
        else switch (side->n_levels) {
        case 1: k_icp_iter_pyramid<1, 0><<<grid, ICP_THREADS, 0, s>>>(a, sp); break;
        case 2: k_icp_iter_pyramid<2, 0><<<grid, ICP_THREADS, 0, s>>>(a, sp); break;
        case 3: k_icp_iter_pyramid<3, 0><<<grid, ICP_THREADS, 0, s>>>(a, sp); break;
        default: k_icp_iter_pyramid<4, 0><<<grid, ICP_THREADS, 0, s>>>(a, sp); break;
        }
        launched("k_icp_iter_pyramid");
        return;
    }
    static WaveTrace wt("KFH_ICP_TRACE");
    static const int trace_it = [] { const char *e = getenv("KFH_ICP_TRACE_IT"); return e ? atoi(e) : 10; }();
    if (it == trace_it && !side) { a.trace = wt.arm(4, s); if (a.trace) { k_icp_iter<<<nblocks, ICP_THREADS, 0, s>>>(a); launched("k_icp_iter"); wt.dump(); return; } }
    if (side && side->kind == SideWork::RAY_ORDER) {
        a.ray_order = side->ray_order; a.ray_cost = side->ray_cost; a.ray_tiles_x = side->ray_tiles_x; a.ray_tiles_y = side->ray_tiles_y; a.order_mode = side->order_mode;
        k_icp_iter_order<<<nblocks + 1, ICP_THREADS, 0, s>>>(a); launched("k_icp_iter_order");
        return;
    }
    k_icp_iter<<<nblocks, ICP_THREADS, 0, s>>>(a); launched("k_icp_iter");
}

void icp_tail(DevState *st, IcpState *iters, float *part0, float *part1, int *result_map, int it, int n_its, int level,
              unsigned nblocks, const float4 *inV, const float4 *inN, unsigned inW, unsigned inH, const float4 *refV,
              const float4 *refN, unsigned refW, unsigned refH, float icp_threshold, float dist_threshold, float dist2_max,
              float normal_threshold, FrameLogEntry *log, int final_ops, const ControlArgs *ca, bool rows_ahead, unsigned max_wgs, hipStream_t s) {
    // (the launch before `it` is an ordinary one of the same level: prev_level = level, prev_nblocks = nblocks)
    IcpIterArgs a = make_icp_args(st, iters, (it + 1) & 1 ? part1 : part0, it & 1 ? part1 : part0, result_map, it, level, level,
                                  nblocks, nblocks, inV, inN, inW, inH, refV, refN, refW, refH, icp_threshold, dist_threshold,
                                  dist2_max, normal_threshold);
    IcpTailArgs t;
    t.part[0] = part0; t.part[1] = part1; t.n_its = n_its;
    t.barrier = &st->tail_barrier[level]; t.timeouts = &st->barrier_timeouts;
    // (a forced time-out fails frames: the switch is a test hook, honoured only with KFH_ENABLE_TEST_HOOKS=1 beside it)
    static const long long ticks_env = [] {
        const char *hooks = getenv("KFH_ENABLE_TEST_HOOKS"), *e = getenv("KFH_ICP_TAIL_TIMEOUT_TICKS");
        return (hooks && hooks[0] == '1' && e) ? atoll(e) : -1LL;
    }();
    t.timeout_ticks = ticks_env >= 0 ? (unsigned long long)ticks_env : 200000000ull;
    t.final_ops = ca ? final_ops : -1; t.log = log;
    if (ca) t.ca = *ca; else t.ca = ControlArgs{};
    static WaveTrace wt("KFH_ICP_TAIL_TRACE");
    t.trace = (ca && level == 0) ? wt.arm(2, s) : nullptr;
    const unsigned cap = max_wgs ? max_wgs : ICP_TAIL_BLOCKS;       // (the caller has reserved that many CUs for its barrier launches)
    const unsigned wgs = nblocks < cap ? nblocks : cap;
    if (rows_ahead) k_icp_tail<true><<<wgs, ICP_THREADS, 0, s>>>(a, t);
    else k_icp_tail<false><<<wgs, ICP_THREADS, 0, s>>>(a, t);
    launched("k_icp_tail");
    if (t.trace) wt.dump();
}

void icp_final(DevState *st, FrameLogEntry *log, const IcpState *iters, const float *partials_prev, int n_iters,
               int prev_level, unsigned prev_nblocks, float icp_threshold, int ops, const ControlArgs &a, hipStream_t s) {
    k_icp_final<<<1, ICP_THREADS, 0, s>>>(st, log, iters, partials_prev, n_iters, prev_level, prev_nblocks, icp_threshold, ops, a); launched("k_icp_final");
}

void update_pose(M4 *pose, const float *reduction32, float icp_threshold, int *converged, hipStream_t s) {
    k_update_pose<<<1, 64, 0, s>>>(pose, reduction32, icp_threshold, converged); launched("k_update_pose");
}

void check_pose(M4 *pose, const M4 *oldPose, const float *reduction32, unsigned W, unsigned H, float thr, int *ok, hipStream_t s) {
    k_check_pose<<<1, 64, 0, s>>>(pose, oldPose, reduction32, W, H, thr, ok); launched("k_check_pose");
}

static long env_long(const char *name, long lo, long hi, long dflt);

static SkipGrid make_skip_grid(const unsigned char *flags, unsigned sx, unsigned sy, unsigned sz, f3 dim) {
    SkipGrid g;
    g.flags = flags;
    g.nbx = (sx + 7) >> 3; g.nby = (sy + 7) >> 3; g.nbz = (sz + 7) >> 3;
    g.kx = sx / dim.x * 0.125f; g.ky = sy / dim.y * 0.125f; g.kz = sz / dim.z * 0.125f;
    g.bx_max = (float)((sx - 1) >> 3); g.by_max = (float)((sy - 1) >> 3); g.bz_max = (float)((sz - 1) >> 3);
    static const long tp = env_long("KFH_RAY_TWO_PHASE", 0, 1, 0);
    g.two_phase = (flags && tp) ? 1 : 0;
    g.trace = nullptr;
    g.order = nullptr; g.cost = nullptr;
    g.tiles_x = 0; g.tiles_y = 0;
    g.next_state = nullptr; g.next_frame = 0; g.next_tracking_rate = 1;
    for (int i = 0; i < 4; ++i) g.next_k[i] = 0.f;
    return g;
}

size_t skip_grid_bytes(unsigned sx, unsigned sy, unsigned sz) { return (size_t)((sx + 7) >> 3) * ((sy + 7) >> 3) * ((sz + 7) >> 3); }

void build_skip_grid(unsigned char *flags, const int16_t *vol, unsigned sx, unsigned sy, unsigned sz, hipStream_t s) {
    hipMemsetAsync(flags, 0, skip_grid_bytes(sx, sy, sz), s);
    k_build_skip_grid<<<2048, 256, 0, s>>>(flags, reinterpret_cast<const short2 *>(vol), sx, sy, sz, (sx + 7) >> 3, (sy + 7) >> 3, (sz + 7) >> 3); launched("k_build_skip_grid");
}

// Has the reciprocal form of a / d been verified for this divisor on this device (see k_verify_div)?  Cached; the check
// takes a few milliseconds once per context (kfh_create calls prepare_volume_division).  KFH_EXACT_DIV=0 keeps IEEE.
struct DivisorEntry { int dev; float d; bool ok; };
static std::vector<DivisorEntry> g_divisors;
static std::mutex g_divisors_mu;
// what an earlier verified_divisor() of this device found; false for a divisor nobody has submitted yet (frame-path launchers
// ask this one: it never launches or waits)
static bool divisor_verified(float d) {
    int dev = 0;
    hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_divisors_mu);
    for (const DivisorEntry &e : g_divisors) if (e.dev == dev && e.d == d) return e.ok;
    return false;
}
static bool verified_divisor(float d, hipStream_t s) {
    typedef DivisorEntry Entry;
    std::vector<Entry> &cache = g_divisors;
    std::mutex &mu = g_divisors_mu;
    static const bool allow = [] { const char *e = getenv("KFH_EXACT_DIV"); return !(e && e[0] == '0'); }();
    int dev = 0;
    hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    for (const Entry &e : cache) if (e.dev == dev && e.d == d) return e.ok;
    bool ok = false;
    if (allow && d > 0x1p-10f && d < 0x1p10f) {
        unsigned long long *dm = nullptr;
        if (hipMalloc(&dm, sizeof *dm) == hipSuccess) {
            unsigned long long h = 1;
            hipMemsetAsync(dm, 0, sizeof *dm, s);
            k_verify_div<<<8192, 256, 0, s>>>(d, 1.0f / d, dm); launched("k_verify_div");
            if (hipMemcpyAsync(&h, dm, sizeof h, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess) ok = h == 0;
            hipFree(dm);
        }
    }
    cache.push_back(Entry{dev, d, ok});
    return ok;
}

// Is "sqrtf(d) > t  <=>  d > sqrt_gt_bound(t)" verified for this threshold on this device?  prepare_sqrt_bound (kfh_create) runs
// k_verify_sqrt_bound over EVERY non-negative float bit pattern (2^31 of them: zeros, denormals, infinity and the NaNs included) with
// the square root this library is compiled with; the frame path only looks the verdict up.  Not verified = the ICP association takes
// the square root as the reference does (cpp:543).
struct SqrtBoundEntry { int dev; float t; bool ok; };
static std::vector<SqrtBoundEntry> g_sqrt_bounds;
static std::mutex g_sqrt_bounds_mu;
// returns the bound to compare squared lengths with (>= 0) or -1 = take the square root; the context keeps it
float prepare_sqrt_bound(float t, hipStream_t s) {
    static const bool allow = [] { const char *e = getenv("KFH_EXACT_SQRT"); return !(e && e[0] == '0'); }();
    int dev = 0;
    hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_sqrt_bounds_mu);
    const float bound = sqrt_gt_bound(t);
    for (const SqrtBoundEntry &e : g_sqrt_bounds) if (e.dev == dev && e.t == t) return e.ok ? bound : -1.f;
    bool ok = false, ran = true;
    if (allow && bound >= 0.f) {
        unsigned long long *dm = nullptr;
        ran = hipMalloc(&dm, sizeof *dm) == hipSuccess;
        if (ran) {
            unsigned long long h = 1;
            ran = hipMemsetAsync(dm, 0, sizeof *dm, s) == hipSuccess;
            if (ran) { k_verify_sqrt_bound<<<8192, 256, 0, s>>>(t, bound, dm); ran = hipGetLastError() == hipSuccess; }
            ran = ran && hipMemcpyAsync(&h, dm, sizeof h, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
            hipFree(dm);
            ok = ran && h == 0;
        }
    }
    if (ran) g_sqrt_bounds.push_back(SqrtBoundEntry{dev, t, ok});      // (a check that could not run is not remembered)
    return ok ? bound : -1.f;
}

// all three axes verified and every numerator pos * size below 2^24
static bool fast_division(f3 dim, unsigned sx, unsigned sy, unsigned sz, hipStream_t s) {
    if (!((double)dim.x * sx < 8.0e6 && (double)dim.y * sy < 8.0e6 && (double)dim.z * sz < 8.0e6)) return false;
    return verified_divisor(dim.x, s) && verified_divisor(dim.y, s) && verified_divisor(dim.z, s);
}

void prepare_volume_division(f3 dim, unsigned sx, unsigned sy, unsigned sz, hipStream_t s) { (void)fast_division(dim, sx, sy, sz, s); }

// k_verify_reduce, once per device and process: 1 = the ICP block reduction computes its reference tree on this device (both
// instantiations the backend launches), 0 = it does not (a completed comparison found a mismatch), -1 = the check could not be run
// (allocation, launch, copy or wait failed: a runtime error, reported as such and NOT remembered - the next kfh_create tries again).
int reduction_verified(hipStream_t s) {
    static std::mutex mu;
    static std::vector<std::pair<int, bool>> done;
    int dev = 0;
    hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    for (const auto &e : done) if (e.first == dev) return e.second ? 1 : 0;
    unsigned *dm = nullptr;
    if (hipMalloc(&dm, sizeof *dm) != hipSuccess) return -1;
    unsigned h = 1;
    bool ran = hipMemsetAsync(dm, 0, sizeof *dm, s) == hipSuccess;
    if (ran) {
        k_verify_reduce<1024><<<64, 1024, 0, s>>>(dm, 0x9E3779B9u);
        ran = hipGetLastError() == hipSuccess;           // (examined here, not booked to a context: a rejected launch verifies nothing)
    }
    if (ran) {
        static_assert(ICP_THREADS == 1024, "k_verify_reduce<1024> stands for the ICP kernels' instantiation");
        k_verify_reduce<TR_THREADS><<<64, TR_THREADS, 0, s>>>(dm, 0x7F4A7C15u);
        ran = hipGetLastError() == hipSuccess;
    }
    ran = ran && hipMemcpyAsync(&h, dm, sizeof h, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
    hipFree(dm);
    if (!ran) return -1;
    done.push_back(std::make_pair(dev, h == 0));
    return h == 0 ? 1 : 0;
}
// the bilateral filter's range divisor 2 e_d^2 (its numerators, squared depth differences, stay far below 2^24 for depth in metres)
void prepare_bilateral_division(float e_d, hipStream_t s) { (void)verified_divisor(e_d * e_d * 2, s); }

// Lanes that march one ray together (cast_ray_group).  Sharing a ray shortens the serial chain per ray but leaves lanes with
// discarded samples and a redundant preamble; it pays while there are too few rays to fill the chip with waves of their own.
// k_raycast at 256^3, us for 4 / 2 / 1 lanes per ray (profiles/r02_ray_group_sizes.txt): 320x240 31 / 47 / 60, 640x480 52 / 54 / 66,
// 800x600 73 / 64 / 71, 960x720 103 / 81 / 71, 1280x960 165 / 122 / 91 - so the group size follows the ray count unless
// KFH_RAY_GROUP = 1, 2, 4, 8 fixes it.  Every size gives the same bits (tests/test_hip_fullsize.py).
static int ray_group(unsigned W, unsigned H) {
    static const long g = env_long("KFH_RAY_GROUP", 0, 8, 0);
    if (g == 1 || g == 2 || g == 4 || g == 8) return (int)g;
    const unsigned long long rays = (unsigned long long)W * H;
    return rays >= 600000ull ? 1 : (rays >= 400000ull ? 2 : 4);
}
static dim3 ray_grid(int q, unsigned W, unsigned H) {
    const unsigned rw = (q <= 2) ? 8 : 4, rh = (64 / q) / rw;
    return dim3((W + RAY_WX * rw - 1) / (RAY_WX * rw), (H + RAY_WY * rh - 1) / (RAY_WY * rh));
}

// dispatch tiles: 2x2 wave tiles (the workgroup of KFH_RAY_BLOCK_WAVES = 4), whatever the workgroup size
static dim3 ray_tile_dim(int q, unsigned W, unsigned H) {
    const unsigned rw = (q <= 2) ? 8 : 4, rh = (64 / q) / rw;
    return dim3((W + 2 * rw - 1) / (2 * rw), (H + 2 * rh - 1) / (2 * rh));
}
// launch grid of k_raycast: one workgroup per part of a tile, slots padded to whole rounds of 8 * parts
static dim3 raycast_launch_grid(int q, unsigned W, unsigned H) {
    const dim3 t = ray_tile_dim(q, W, H);
    const unsigned parts = 4 / KFH_RAY_BLOCK_WAVES, n = t.x * t.y;
    if (parts == 1) return t;
    return dim3(((n + 7) / 8) * 8 * parts, 1);
}
static long env_long(const char *name, long lo, long hi, long dflt) {
    const char *e = getenv(name);
    if (!e) return dflt;
    const long v = atol(e);
    return (v >= lo && v <= hi) ? v : dflt;
}

// enough z-chunks to put >= ~8192 waves in flight on 256 CUs (each chunk replays its z0 additions, so not more: at
// 512^3 four chunks of 128 are the measured optimum)
static dim3 integrate_grid(unsigned sx, unsigned sy, unsigned sz, unsigned *zchunk_out) {
    static const long chunks_env = env_long("KFH_INTEGRATE_CHUNKS", 1, 64, 0);
    const unsigned waves_xy = ((sx + 63) / 64) * sy;     // (waves of 64 columns, whatever their shape)
    unsigned nchunks = chunks_env ? (unsigned)chunks_env : (8192 + waves_xy - 1) / waves_xy;
    if (!chunks_env && nchunks < 4 && sz >= 256) nchunks = 4;
    if (nchunks < 1) nchunks = 1;
    if (!chunks_env && nchunks > 16) nchunks = 16;
    if (nchunks > sz) nchunks = sz;
    const unsigned zchunk = (sz + nchunks - 1) / nchunks;
    if (zchunk_out) *zchunk_out = zchunk;
    return dim3((sx + INT_BX - 1) / INT_BX, (sy + INT_BY - 1) / INT_BY, (sz + zchunk - 1) / zchunk);
}
unsigned integrate_blocks(unsigned sx, unsigned sy, unsigned sz) { const dim3 g = integrate_grid(sx, sy, sz, nullptr); return g.x * g.y * g.z; }
bool integrate_reorder_enabled() { static const bool on = [] { const char *e = getenv("KFH_INTEGRATE_REORDER"); return !(e && e[0] == '0'); }(); return on; }
void integrate_order(unsigned *block_order, unsigned *block_cost, unsigned n, hipStream_t s) {   // sort as a launch of its own
    k_ray_order<<<1, 1024, 0, s>>>(block_order, block_cost, n, 1u, 1u); launched("k_ray_order");
}

void integrate(int16_t *vol, unsigned sx, unsigned sy, unsigned sz, f3 dim, const float *depth, unsigned W, unsigned H,
               const IntegrateParams *p, float mu, float maxweight, unsigned long long *updated_total, unsigned char *skip_flags,
               unsigned *block_order, unsigned *block_cost, bool k_trivial, hipStream_t s) {
    static const long batch_env = env_long("KFH_INTEGRATE_BATCH", 1, 16, 8);
#if KFH_INT_PROBES
    // census build only (-DKFH_INT_PROBES=1; results are wrong with a probe on): 1 every batch rejected, 2 stop after the projection
    // pass, 4 stop after the depth classification, 8 no whole-chunk reject; KFH_INT_PROBE_FROM = first launch they apply to
    static const int dbg_env = (int)env_long("KFH_INT_PROBE_STAGE", 0, 15, 0);
    static const long dbg_from = env_long("KFH_INT_PROBE_FROM", 0, 1000000, 0);
    static std::atomic<long> n_launch{0};
    const int dbg = (dbg_env && n_launch.fetch_add(1) >= dbg_from) ? dbg_env : 0;
#else
    const int dbg = 0;
#endif
    unsigned zchunk;
    const dim3 grid = integrate_grid(sx, sy, sz, &zchunk);
    short2 *v = reinterpret_cast<short2 *>(vol);
    static WaveTrace wt("KFH_INTEGRATE_TRACE");
    unsigned long long *trace = wt.arm((size_t)grid.x * grid.y * grid.z * 4, s);
    const bool reorder = integrate_reorder_enabled() && block_order && block_cost;
    const unsigned *order = reorder ? block_order : nullptr;
    unsigned *cost = reorder ? block_cost : nullptr;
    // (k_trivial: the third row of K is (0, 0, 1, 0), see KZ)
#define KFH_LAUNCH_INTEGRATE(IB_)                                                                                        \
    do {                                                                                                                 \
        if (k_trivial) k_integrate<IB_, true><<<grid, block_rows(), 0, s>>>(v, sx, sy, sz, dim, depth, W, H, p, mu, maxweight, zchunk, updated_total, skip_flags, dbg, trace, order, cost);  \
        else k_integrate<IB_, false><<<grid, block_rows(), 0, s>>>(v, sx, sy, sz, dim, depth, W, H, p, mu, maxweight, zchunk, updated_total, skip_flags, dbg, trace, order, cost);           \
    } while (0)
    if (batch_env >= 16) KFH_LAUNCH_INTEGRATE(16);
    else if (batch_env >= 8) KFH_LAUNCH_INTEGRATE(8);
    else if (batch_env >= 4) KFH_LAUNCH_INTEGRATE(4);
    else KFH_LAUNCH_INTEGRATE(1);
#undef KFH_LAUNCH_INTEGRATE
    launched("k_integrate");
    wt.dump();
}

void raycast(float4 *vertex, float4 *normal, unsigned W, unsigned H, const int16_t *vol, unsigned sx, unsigned sy,
             unsigned sz, f3 dim, const RayParams *p, float nearPlane, float farPlane, float step, float largestep,
             const unsigned char *skip_flags, unsigned *tile_order, unsigned *tile_cost, const NextTrackBegin *next, hipStream_t s) {
    const short2 *vv = reinterpret_cast<const short2 *>(vol);
    SkipGrid sg = make_skip_grid(skip_flags, sx, sy, sz, dim);
    if (next && next->state) {
        sg.next_state = next->state; sg.next_frame = next->frame; sg.next_tracking_rate = next->tracking_rate ? next->tracking_rate : 1;
        for (int i = 0; i < 4; ++i) sg.next_k[i] = next->k[i];
    }
    if (ray_reorder_enabled() && tile_order && tile_cost) { sg.order = tile_order; sg.cost = tile_cost; }
    static WaveTrace wt("KFH_RAY_TRACE");
    const dim3 tg = raycast_launch_grid(ray_group(W, H), W, H), td = ray_tile_dim(ray_group(W, H), W, H);
    sg.tiles_x = td.x; sg.tiles_y = td.y;
    sg.trace = wt.arm((size_t)tg.x * tg.y * KFH_RAY_BLOCK_WAVES, s);
    const f3 rdim = mk3(1.0f / dim.x, 1.0f / dim.y, 1.0f / dim.z);
    if (fast_division(dim, sx, sy, sz, s)) {
        switch (ray_group(W, H)) {
        case 1: k_raycast<1, 1><<<raycast_launch_grid(1, W, H), RAY_BLOCK, 0, s>>>(vertex, normal, W, H, vv, sx, sy, sz, dim, p, nearPlane, farPlane, step, largestep, sg, rdim); launched("k_raycast"); break;
        case 2: k_raycast<2, 1><<<raycast_launch_grid(2, W, H), RAY_BLOCK, 0, s>>>(vertex, normal, W, H, vv, sx, sy, sz, dim, p, nearPlane, farPlane, step, largestep, sg, rdim); launched("k_raycast"); break;
        case 8: k_raycast<8, 1><<<raycast_launch_grid(8, W, H), RAY_BLOCK, 0, s>>>(vertex, normal, W, H, vv, sx, sy, sz, dim, p, nearPlane, farPlane, step, largestep, sg, rdim); launched("k_raycast"); break;
        default: k_raycast<4, 1><<<raycast_launch_grid(4, W, H), RAY_BLOCK, 0, s>>>(vertex, normal, W, H, vv, sx, sy, sz, dim, p, nearPlane, farPlane, step, largestep, sg, rdim); launched("k_raycast"); break;
        }
    } else {
        switch (ray_group(W, H)) {
        case 1: k_raycast<1, 0><<<raycast_launch_grid(1, W, H), RAY_BLOCK, 0, s>>>(vertex, normal, W, H, vv, sx, sy, sz, dim, p, nearPlane, farPlane, step, largestep, sg, rdim); launched("k_raycast"); break;
        case 2: k_raycast<2, 0><<<raycast_launch_grid(2, W, H), RAY_BLOCK, 0, s>>>(vertex, normal, W, H, vv, sx, sy, sz, dim, p, nearPlane, farPlane, step, largestep, sg, rdim); launched("k_raycast"); break;
        case 8: k_raycast<8, 0><<<raycast_launch_grid(8, W, H), RAY_BLOCK, 0, s>>>(vertex, normal, W, H, vv, sx, sy, sz, dim, p, nearPlane, farPlane, step, largestep, sg, rdim); launched("k_raycast"); break;
        default: k_raycast<4, 0><<<raycast_launch_grid(4, W, H), RAY_BLOCK, 0, s>>>(vertex, normal, W, H, vv, sx, sy, sz, dim, p, nearPlane, farPlane, step, largestep, sg, rdim); launched("k_raycast"); break;
        }
    }
    wt.dump();
}

void ray_order(unsigned *tile_order, unsigned *tile_cost, unsigned W, unsigned H, hipStream_t s) {
    const dim3 tg = ray_tile_dim(ray_group(W, H), W, H);
    k_ray_order<<<1, 1024, 0, s>>>(tile_order, tile_cost, tg.x, tg.y, 0u); launched("k_ray_order");
}
void ray_tile_grid(unsigned W, unsigned H, unsigned *tx, unsigned *ty) { const dim3 g = ray_tile_dim(ray_group(W, H), W, H); *tx = g.x; *ty = g.y; }
bool ray_reorder_enabled() { static const bool on = [] { const char *e = getenv("KFH_RAY_REORDER"); return !(e && e[0] == '0'); }(); return on; }

unsigned ray_tiles(unsigned W, unsigned H) { const dim3 g = ray_tile_dim(ray_group(W, H), W, H); return g.x * g.y; }

__global__ void k_iota(unsigned *p, unsigned n) {
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = i;
}
void init_ray_order(unsigned *tile_order, unsigned *tile_cost, unsigned n, hipStream_t s) {
    k_iota<<<(n + 255) / 256, 256, 0, s>>>(tile_order, n); launched("k_iota");
    hipMemsetAsync(tile_cost, 0, (size_t)2 * n * sizeof(unsigned), s);
}

void render_depth(uchar4 *out, const float *depth, unsigned W, unsigned H, float nearPlane, float farPlane, hipStream_t s) {
    k_render_depth<<<grid_rows(W, H), block_rows(), 0, s>>>(out, depth, W, H, nearPlane, farPlane); launched("k_render_depth");
}

void render_track(uchar4 *out, const int *result_map, unsigned W, unsigned H, hipStream_t s) {
    k_render_track<<<grid_rows(W, H), block_rows(), 0, s>>>(out, result_map, W, H); launched("k_render_track");
}

void render_volume(uchar4 *out, unsigned W, unsigned H, const int16_t *vol, unsigned sx, unsigned sy, unsigned sz, f3 dim,
                   const RayParams *p, float nearPlane, float farPlane, float step, float largestep, f3 light, f3 ambient,
                   const unsigned char *skip_flags, hipStream_t s) {
    const short2 *vv = reinterpret_cast<const short2 *>(vol);
    const SkipGrid sg = make_skip_grid(skip_flags, sx, sy, sz, dim);
    const f3 rdim = mk3(1.0f / dim.x, 1.0f / dim.y, 1.0f / dim.z);
    if (fast_division(dim, sx, sy, sz, s)) {
        switch (ray_group(W, H)) {
        case 1: k_render_volume<1, 1><<<ray_grid(1, W, H), RAY_BLOCK, 0, s>>>(out, W, H, vv, sx, sy, sz, dim, p, nearPlane, farPlane, step, largestep, light, ambient, sg, rdim); launched("k_render_volume"); break;
        case 2: k_render_volume<2, 1><<<ray_grid(2, W, H), RAY_BLOCK, 0, s>>>(out, W, H, vv, sx, sy, sz, dim, p, nearPlane, farPlane, step, largestep, light, ambient, sg, rdim); launched("k_render_volume"); break;
        case 8: k_render_volume<8, 1><<<ray_grid(8, W, H), RAY_BLOCK, 0, s>>>(out, W, H, vv, sx, sy, sz, dim, p, nearPlane, farPlane, step, largestep, light, ambient, sg, rdim); launched("k_render_volume"); break;
        default: k_render_volume<4, 1><<<ray_grid(4, W, H), RAY_BLOCK, 0, s>>>(out, W, H, vv, sx, sy, sz, dim, p, nearPlane, farPlane, step, largestep, light, ambient, sg, rdim); launched("k_render_volume"); break;
        }
    } else {
        switch (ray_group(W, H)) {
        case 1: k_render_volume<1, 0><<<ray_grid(1, W, H), RAY_BLOCK, 0, s>>>(out, W, H, vv, sx, sy, sz, dim, p, nearPlane, farPlane, step, largestep, light, ambient, sg, rdim); launched("k_render_volume"); break;
        case 2: k_render_volume<2, 0><<<ray_grid(2, W, H), RAY_BLOCK, 0, s>>>(out, W, H, vv, sx, sy, sz, dim, p, nearPlane, farPlane, step, largestep, light, ambient, sg, rdim); launched("k_render_volume"); break;
        case 8: k_render_volume<8, 0><<<ray_grid(8, W, H), RAY_BLOCK, 0, s>>>(out, W, H, vv, sx, sy, sz, dim, p, nearPlane, farPlane, step, largestep, light, ambient, sg, rdim); launched("k_render_volume"); break;
        default: k_render_volume<4, 0><<<ray_grid(4, W, H), RAY_BLOCK, 0, s>>>(out, W, H, vv, sx, sy, sz, dim, p, nearPlane, farPlane, step, largestep, light, ambient, sg, rdim); launched("k_render_volume"); break;
        }
    }
}

}  // namespace launch
}  // namespace kf
