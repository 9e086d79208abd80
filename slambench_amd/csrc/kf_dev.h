// kf_dev.h - device code and host helpers SHARED by the kernel translation units (internal to slambench_amd/csrc).
//
// The hot path is split by reference kernel (kf_preprocess.hip, kf_icp.hip, kf_integrate.hip, kf_raycast.hip, kf_render.hip) with the
// launch-side odds and ends in kf_launch.hip.  What more than one of them needs lives here as inline device code: the tile routines of
// the bilateral filter and the input pyramid (stand-alone kernels AND side blocks of the ICP launches), the control operations (k_control,
// k_icp_tail, k_icp_final, and the tracking preamble that rides on k_raycast), the dispatch sort (k_ray_order and the ICP side block),
// the brick flags (set by integrate, read by the ray march).  All of it is compiled with -ffp-contract=off and correctly rounded fp32
// divide / sqrt (slambench_amd/build.py): integer outputs match the CPU reference bit for bit, float outputs differ only where a sum is
// re-associated (the ICP reduction).  Nothing here is a dense contraction: no MFMA.
#pragma once
#include "kf_kernels.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

namespace kf {

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
static __constant__ unsigned long long kExp2fTable[32] = {
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

// =====================================================================================
// depth2vertexKernel (cpp:200-218) and vertex2normalKernel (cpp:220-249)
// =====================================================================================
__device__ __forceinline__ f3 back_project(float d, int x, int y, const M4 &invK) {
    return d > 0 ? scale3(rotate(invK, mk3((float)x, (float)y, 1.f)), d) : mk3(0.f, 0.f, 0.f);
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

// checkPoseKernel (cpp:777-792): the verdict; the callers replace a rejected pose by the one tracking started from
__device__ __forceinline__ bool pose_is_good(const float *red, unsigned W, unsigned H, float track_threshold) {
    return !(((double)sqrtf(red[0] / red[28]) > 2e-2) || (red[28] / (W * H) < track_threshold));
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

static __device__ void control_ops(DevState *__restrict__ st, FrameLogEntry *__restrict__ log, int ops, const ControlArgs &a) {
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

// ---- brick flags of the exact empty-space skipping (set by k_integrate, read by the ray march: kf_raycast.hip) -----------------
#ifndef KFH_BRICK_SHIFT
#define KFH_BRICK_SHIFT 3      // bricks of 8^3 voxels (2 = 4^3: measured in profiles/r05_sweeps.md)
#endif
constexpr int BRICK_SHIFT = KFH_BRICK_SHIFT, BRICK = 1 << BRICK_SHIFT;
__host__ __device__ inline unsigned bricks_along(unsigned voxels) { return (voxels + BRICK - 1) >> BRICK_SHIFT; }
constexpr short kSkipShortFwd = 26600;
constexpr short kSkipShort = kSkipShortFwd;
__device__ __forceinline__ unsigned umax_(unsigned a, unsigned b) { return a > b ? a : b; }
__device__ __forceinline__ unsigned umin_(unsigned a, unsigned b) { return a < b ? a : b; }
__device__ __forceinline__ void mark_bricks(unsigned char *flags, int nbx, int nby, int nbz, int x, int y, int z) {
    const int x0 = x >= 2 ? (x - 2) >> BRICK_SHIFT : 0, x1 = imin((x + 1) >> BRICK_SHIFT, nbx - 1);
    const int y0 = y >= 2 ? (y - 2) >> BRICK_SHIFT : 0, y1 = imin((y + 1) >> BRICK_SHIFT, nby - 1);
    const int z0 = z >= 2 ? (z - 2) >> BRICK_SHIFT : 0, z1 = imin((z + 1) >> BRICK_SHIFT, nbz - 1);
    for (int bz = z0; bz <= z1; ++bz)
        for (int by = y0; by <= y1; ++by)
            for (int bx = x0; bx <= x1; ++bx) flags[bx + nbx * (by + nby * bz)] = 1;
}

namespace launch {

// Every launcher notes a rejected launch (bad grid / block shape, too much LDS, ...) by kernel name (kf_launch.hip); the pipeline asks
// after each phase it queues, so a failed launch becomes KFH_ERR_HIP of that call instead of stale buffers at the next getter.
void launched(const char *kernel);
// a launcher that will not launch (sizes outside what the kernels' 32-bit addressing covers) books the reason the same way
void refused(const char *why);
// 32-bit voxel byte offsets and 24-bit multiplies in the voxel addressing: <= 2^30 voxels, an x-y plane below 2^24; pixel byte
// offsets likewise: fewer than 2^26 pixels (kfh_create checks the same for a context's own buffers)
inline bool sizes_addressable(unsigned sx, unsigned sy, unsigned sz, unsigned W, unsigned H) {
    const unsigned long long n = (unsigned long long)sx * sy * sz;
    return n >= 1ull && n <= (1ull << 30) && (unsigned long long)sx * sy < (1ull << 24) && (unsigned long long)W * H < (1ull << 26);
}

static inline dim3 grid_rows(unsigned W, unsigned H) { return dim3((W + 63) / 64, (H + 3) / 4); }
static inline dim3 block_rows() { return dim3(64, 4); }

inline long env_long(const char *name, long lo, long hi, long dflt) {
    const char *e = getenv(name);
    if (!e) return dflt;
    const long v = atol(e);
    return (v >= lo && v <= hi) ? v : dflt;
}

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

// reciprocal-form divisions (kf_launch.hip): what an earlier verified_divisor() of this device found - never a launch - and the check itself
bool divisor_verified(float d);
bool verified_divisor(float d, hipStream_t s);
// one launch of the dispatch sort (k_ray_order, kf_launch.hip)
void dispatch_order(unsigned *order, unsigned *cost, unsigned tiles_x, unsigned tiles_y, unsigned mode, hipStream_t s);

}  // namespace launch
}  // namespace kf
