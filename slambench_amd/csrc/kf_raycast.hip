// kf_raycast.hip - Volume::interp / grad (commons.h:191-301), raycastKernel (cpp:674-757), renderVolumeKernel (cpp:880-913) and the
// exact empty-space skipping; their launchers.
#include "kf_dev.h"

namespace kf {

// =====================================================================================
// Volume sampling (commons.h:191-301) and the ray march (cpp:674-725)
// =====================================================================================
// Stage probes for the instruction census (profiles/r05_raycast_census.txt): only in a build with -DKFH_RAY_PROBES=1; results are wrong
// with a probe on and the shipped library has no such path.
#ifndef KFH_RAY_PROBES
#define KFH_RAY_PROBES 0
#endif
#ifndef KFH_RAY_PREFETCH
#define KFH_RAY_PREFETCH 1
#endif

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

struct SkipGrid {
    const unsigned char *flags;   // nullptr = no skipping
    int nbx, nby, nbz;
    float kx, ky, kz;             // size / dim per axis, divided by 8 (voxels -> bricks)
    float bx_max, by_max, bz_max; // last brick index per axis, as floats
    int two_phase;                // loop shape (only with flags)
    int probe;                    // census build only (-DKFH_RAY_PROBES=1, results wrong): 1 = every ray ends after its set-up, 2 = no surface gradient, 4 = every sample in an empty brick
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
    const unsigned bx = (unsigned)(int)__builtin_amdgcn_fmed3f(__builtin_fmaf(pos.x, g.kx, -0.5f / BRICK), 0.f, g.bx_max);
    const unsigned by = (unsigned)(int)__builtin_amdgcn_fmed3f(__builtin_fmaf(pos.y, g.ky, -0.5f / BRICK), 0.f, g.by_max);
    const unsigned bz = (unsigned)(int)__builtin_amdgcn_fmed3f(__builtin_fmaf(pos.z, g.kz, -0.5f / BRICK), 0.f, g.bz_max);
    (void)v;
    return g.flags[__umul24(__umul24(bz, (unsigned)g.nby) + by, (unsigned)g.nbx) + bx];
}
template <class V>
__device__ __forceinline__ bool brick_is_empty(const SkipGrid &g, const V &v, f3 pos) { return brick_flag(g, v, pos) == 0; }

// The same look-up for the sample at parameter t of a ray, without forming the sample's position: the brick coordinate
// (origin + dir t) k - 1/16 as ONE multiply-add per axis of two per-ray constants.  It differs from the look-up on the rounded position
// by a few 2^-24 of a brick - far inside the one-voxel slack the flags' dilation allows for - and in a round that finds only empty
// bricks the position is never needed (6 of a round's ~45 vector instructions).
struct BrickRay { f3 o, d; };
__device__ __forceinline__ BrickRay brick_ray(const SkipGrid &g, f3 origin, f3 dir) {
    BrickRay r;
    r.o = mk3(__builtin_fmaf(origin.x, g.kx, -0.5f / BRICK), __builtin_fmaf(origin.y, g.ky, -0.5f / BRICK), __builtin_fmaf(origin.z, g.kz, -0.5f / BRICK));
    r.d = mk3(dir.x * g.kx, dir.y * g.ky, dir.z * g.kz);
    return r;
}
__device__ __forceinline__ unsigned char brick_flag_at(const SkipGrid &g, const BrickRay &r, float t) {
    const unsigned bx = (unsigned)(int)__builtin_amdgcn_fmed3f(__builtin_fmaf(r.d.x, t, r.o.x), 0.f, g.bx_max);
    const unsigned by = (unsigned)(int)__builtin_amdgcn_fmed3f(__builtin_fmaf(r.d.y, t, r.o.y), 0.f, g.by_max);
    const unsigned bz = (unsigned)(int)__builtin_amdgcn_fmed3f(__builtin_fmaf(r.d.z, t, r.o.z), 0.f, g.bz_max);
    return g.flags[__umul24(__umul24(bz, (unsigned)g.nby) + by, (unsigned)g.nbx) + bx];
}
__device__ __forceinline__ bool brick_is_empty(const SkipGrid &g, const BrickRay &r, float t) { return brick_flag_at(g, r, t) == 0; }

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
        const BrickRay br = sg.flags ? brick_ray(sg, origin, dir) : BrickRay{};
        bool lazy = sg.flags && brick_is_empty(sg, br, t);
        float t_ft = t;
        float f_t = lazy ? 1.0f : interp(v, p0);
        float f_tt = 0;
        if (f_t > 0) {
            if (sg.two_phase) {
                // Same visiting order as `for (; t < tfar; t += stepsize)`, written as two phases so that the lanes of
                // a wave first run through their free-space samples in a short loop and then take the expensive
                // sample together.
                while (true) {
                    while (t < tfar && brick_is_empty(sg, br, t)) {
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
                // One ray per lane (the large images): a sample in an empty brick costs the lane a dependent round trip for one flag byte.
                // While the ray is in free space (`lazy`: its last sample lay in an empty brick) the bricks of the next FOUR samples are
                // looked up together - their times are the serial sums t, t + s, (t + s) + s, ... of the reference loop - and the leading
                // run of empty ones is accepted at once: four look-ups in flight instead of four round trips in a row.  Near a surface
                // (`lazy` false) one sample is examined at a time, as the reference does.
                // (One loop body - look-up(s), then the sample if there is one to take - for all lanes: with `continue` paths the compiler
                // builds an inner loop for the free-space lanes that the sampling lanes of the wave wait out, +80 % at 1280x960.)
                unsigned char fl = 1;        // near a surface: the flag of the sample at t, requested while the previous sample was taken
                bool fl_valid = false;
                while (t < tfar) {
                    bool sample = true;
                    if (sg.flags) {
                        if (lazy) {
                            const float t1 = t + stepsize, t2 = t1 + stepsize, t3 = t2 + stepsize;
                            const bool b0 = brick_is_empty(sg, br, t), b1 = brick_is_empty(sg, br, t1), b2 = brick_is_empty(sg, br, t2), b3 = brick_is_empty(sg, br, t3);
                            const bool e1 = b0 & b1 & (t1 < tfar), e2 = e1 & b2 & (t2 < tfar), e3 = e2 & b3 & (t3 < tfar);
                            work += (int)b0 + (int)e1 + (int)e2 + (int)e3;
                            const float last = e3 ? t3 : (e2 ? t2 : (e1 ? t1 : t));
                            t_ft = b0 ? last : t_ft;
                            t = b0 ? last + stepsize : t;
                            sample = !b0;
                        } else {
                            if (!fl_valid) fl = brick_flag_at(sg, br, t);
                            if (fl == 0) {
                                ++work;
                                lazy = true; t_ft = t; t += stepsize;
                                sample = false;
                            }
                        }
                        fl_valid = false;
                    }
                    if (sample) {
                        ++work;
                        const float tn = t + stepsize;      // the next sample's time unless this one switches the step
                        if (sg.flags) fl = brick_flag_at(sg, br, tn);
                        f_tt = interp(v, add3(origin, scale3(dir, t)));
                        if (f_tt < 0) break;
                        const float s0 = stepsize;
                        if (f_tt < 0.8f) stepsize = step;
                        f_t = f_tt;
                        lazy = false;
                        t += stepsize;
                        fl_valid = sg.flags && stepsize == s0;
                    }
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

// Returns the ray parameter of the hit (cpp:717-720), 0 for a miss; march() forms the position from it.  (A float4 result set inside
// the loop is four more registers that the loop's latch copies out and back on every round - the compiler threads the value of the
// `return`s through the loop - where one will do: the position is three multiplications and three additions after the loop.)
template <int Q, class V>
__device__ __forceinline__ float cast_ray_group(const V &v, const SkipGrid &sg, unsigned px, unsigned py,
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
    const float miss = 0.f;
    if (!(tnear < tfar)) return miss;
#if KFH_RAY_PROBES
    if (sg.probe & 1) return miss;
#endif
    const BrickRay br = sg.flags ? brick_ray(sg, origin, dir) : BrickRay{};
    float t = tnear, stepsize = largestep;
    bool small = false;          // stepsize has been switched to `step`
    bool lazy = false;           // f_t lies in an empty brick: >= 0.81, value not evaluated
    float f_t = 0.f, t_ft = t;
    bool first = true;
#if KFH_RAY_PREFETCH
    // The brick flag of a sample is requested one round ahead: if this round's Q samples are all plain "continue", the next
    // round's times follow from this round's alone (last time + step, then the lanes' own additions), so their flags are in
    // flight while this round's samples are taken - one dependent round trip per round instead of two.  A round that ends in
    // an event (step switch) recomputes times and flags.
    float tq = t;
#pragma unroll
    for (int i = 0; i < Q - 1; ++i) tq = (i < q) ? tq + stepsize : tq;
    unsigned char fl = sg.flags ? brick_flag_at(sg, br, tq) : (unsigned char)1;
#endif
    while (true) {
        work += Q;
#if KFH_RAY_PREFETCH
        const float t_last = group_last<Q>(tq, base);
        float tqn = t_last + stepsize;
#pragma unroll
        for (int i = 0; i < Q - 1; ++i) tqn = (i < q) ? tqn + stepsize : tqn;
        const unsigned char fln = sg.flags ? brick_flag_at(sg, br, tqn) : (unsigned char)1;
#else
        float tq = t;
#pragma unroll
        for (int i = 0; i < Q - 1; ++i) tq = (i < q) ? tq + stepsize : tq;
#endif
        const bool in_range = tq < tfar;
        bool s_lazy = false;
        float f = 0.f;
        if (in_range) {
#if KFH_RAY_PREFETCH
            if (fl == 0) s_lazy = true;
#elif KFH_RAY_PROBES
            if (sg.flags && (brick_is_empty(sg, br, tq) || (sg.probe & 4))) s_lazy = true;
#else
            if (sg.flags && brick_is_empty(sg, br, tq)) s_lazy = true;
#endif
            else f = interp(v, add3(origin, scale3(dir, tq)));
        }
        // what the serial loop does at this sample: 0 continue, 1 out of range, 2 zero crossing, 3 large->small step (as selects: the
        // nested ifs compile to exec-mask branches on the scalar unit, in every round)
        const bool sampled = in_range && !s_lazy;
        int kind = (sampled && !small && f < 0.8f) ? 3 : 0;
        kind = (sampled && f < 0) ? 2 : kind;
        kind = in_range ? kind : 1;
        if (first) {
            // cpp:706-708: the march only starts if the sample at tnear (lane 0 of the first round) is positive; otherwise that
            // lane reports "leave" and, being the group's first event, ends the ray as a miss
            first = false;
            if (q == 0 && sampled && !(f > 0)) kind = 1;
        }
        const unsigned long long ball = __ballot(kind != 0);
        const unsigned evmask = (unsigned)((ball >> base) & ((1ull << Q) - 1ull));
        if (evmask == 0) {   // all Q samples are plain "continue": f_t <- last of them, t <- its time + stepsize
            // (Round 5 measured a look-ahead here - in free space every lane also examines the sample one round ahead, 2 Q samples per
            // round: bit-identical, 47.4 -> 51.0 us at cfg2, and 51.8 with the look-ahead switched off at run time: the loop does not
            // forgive extra code.  profiles/r05_ab_ray_wide.txt)
            f_t = group_last<Q>(f, base);
            lazy = group_last<Q>((int)s_lazy, base) != 0;
#if KFH_RAY_PREFETCH
            t_ft = t_last;
            tq = tqn; fl = fln;
#else
            t_ft = group_last<Q>(tq, base);
            t = t_ft + stepsize;
#endif
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
            return te + stepsize * fe / (f_t - fe);
        }
        // cpp:713-715: first sample below 0.8 -> switch to the small step from here on
        small = true;
        stepsize = step;
        f_t = fe;
        lazy = false;
        t = te + stepsize;
#if KFH_RAY_PREFETCH
        tq = t;
#pragma unroll
        for (int i = 0; i < Q - 1; ++i) tq = (i < q) ? tq + stepsize : tq;
        fl = sg.flags ? brick_flag_at(sg, br, tq) : (unsigned char)1;
#endif
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
    const float th = cast_ray_group<Q>(v, sg, x, y, view, nearPlane, farPlane, step, largestep, work);
    if (!(th > 0.f)) return make_float4(0.f, 0.f, 0.f, 0.f);      // (a hit's parameter is >= nearPlane > 0)
    const f3 origin = mk3(view.m[0][3], view.m[1][3], view.m[2][3]);
    const f3 dir = rotate(view, mk3((float)x, (float)y, 1.f));
    const f3 p = add3(origin, scale3(dir, th));
    return make_float4(p.x, p.y, p.z, th);
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
#if KFH_RAY_PROBES
        const f3 n = (sg.probe & 2) ? mk3(hit.x, hit.y, hit.z) : grad_group<Q>(v, mk3(hit.x, hit.y, hit.z));
#else
        const f3 n = grad_group<Q>(v, mk3(hit.x, hit.y, hit.z));
#endif
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

// Lambert shading of renderVolumeKernel (cpp:897-907) from a hit position and the NORMALISED surface gradient
__device__ __forceinline__ uchar4 shade(f3 test, f3 unit_normal, f3 light, f3 ambient) {
    const f3 diff = normalize3(sub3(light, test));
    const float dir = fmax_h(dot3(unit_normal, diff), 0.f);
    const f3 col = scale3(mk3(clamp_h(dir + ambient.x, 0.f, 1.f), clamp_h(dir + ambient.y, 0.f, 1.f),
                              clamp_h(dir + ambient.z, 0.f, 1.f)), 255);
    return make_uchar4((unsigned char)col.x, (unsigned char)col.y, (unsigned char)col.z, 0);
}

// renderVolumeKernel (cpp:880-913).
// `seen` (or nullptr): the frame's own raycast, to be reused where it applies.  renderVolume marches the rays of `view` from the
// same near plane with the same two step sizes as raycastKernel and differs in its far plane only (farPlane * 2, cpp:1040): when
// the view matrix is bit for bit the one the tracking raycast used (the benchmark driver renders from the live pose, kernels.h:179-184)
// on the same volume, a ray that HIT something there - before the nearer far plane - takes the very same steps here and ends in the
// same crossing, and raycastKernel has stored what the shading needs: the hit position (vertex map) and normalize(grad) (normal
// map, x = -2 where the gradient vanishes: black here).  Only the rays that missed are marched again - their march may go on
// beyond the tracking far plane.  The launcher vouches for the volume and the march parameters, the kernel compares the matrices.
struct SeenRays { const float4 *vertex, *normal; const RayParams *rp; float far_plane; };      // far_plane: the tracking raycast's

// where the ray of pixel (px, py) leaves the volume: smallest_tmax of the slab test, with the expressions of cast_ray (cpp:685-699)
__device__ __forceinline__ float ray_volume_exit(const M4 &view, unsigned px, unsigned py, f3 dim) {
    const f3 origin = mk3(view.m[0][3], view.m[1][3], view.m[2][3]);
    const f3 dir = rotate(view, mk3((float)px, (float)py, 1.f));
    const f3 invR = mk3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
    const f3 tbot = mul3(scale3(invR, -1.0f), origin);
    const f3 ttop = mul3(invR, sub3(dim, origin));
    const f3 tmax = mk3(fmax_h(ttop.x, tbot.x), fmax_h(ttop.y, tbot.y), fmax_h(ttop.z, tbot.z));
    return fmin_h(fmin_h(tmax.x, tmax.y), fmin_h(tmax.x, tmax.z));
}
template <int Q, int FD>
__global__ __launch_bounds__(RAY_BLOCK) void k_render_volume(uchar4 *__restrict__ out, unsigned W, unsigned H,
                                                       const short2 *__restrict__ vol, unsigned sx, unsigned sy,
                                                       unsigned sz, f3 dim, const RayParams *__restrict__ rp,
                                                       float nearPlane, float farPlane, float step, float largestep,
                                                       f3 light, f3 ambient, SkipGrid sg, f3 rdim, SeenRays seen) {
    unsigned x, y;
    tile_pixel<Q>(x, y, blockIdx.x, blockIdx.y);
    if (x >= W || y >= H) return;
    const M4 view = rp->view;
    const bool writer = (threadIdx.x & (Q - 1)) == 0;
    if (seen.rp) {
        const RayParams T = *seen.rp;
        bool same = T.enabled != 0;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) same = same && __builtin_bit_cast(unsigned, T.view.m[r][c]) == __builtin_bit_cast(unsigned, view.m[r][c]);
        if (same) {      // (uniform over the launch)
            const float4 v4 = seen.vertex[x + y * W], n4 = seen.normal[x + y * W];
            if (!(v4.x == 0.f && v4.y == 0.f && v4.z == 0.f)) {      // a hit (uniform over the lanes of a ray: they read the same pixel)
                if (writer) out[x + y * W] = n4.x == kInvalid ? make_uchar4(0, 0, 0, 0) : shade(mk3(v4.x, v4.y, v4.z), mk3(n4.x, n4.y, n4.z), light, ambient);
                return;
            }
            // A ray that missed there and leaves the volume before the TRACKING far plane has the same far end here
            // (min(exit, far) = exit under both far planes): the same march, the same miss - black, without marching.  Only a ray
            // that the tracking far plane cut short is marched (a NaN exit compares false and is marched).
            if (ray_volume_exit(view, x, y, dim) <= seen.far_plane) {
                if (writer) out[x + y * W] = make_uchar4(0, 0, 0, 0);
                return;
            }
        }
    }
    VolViewT<FD> v;
    v.data = vol; v.sx = sx; v.sy = sy; v.sz = sz; v.dim = dim; v.rdim = rdim; v.plane = (size_t)sx * sy;
    int work = 0;
    const float4 hit = march<Q>(v, sg, x, y, view, nearPlane, farPlane, step, largestep, work);
    if (Q != 4 && !writer) return;
    uchar4 o = make_uchar4(0, 0, 0, 0);
    if (hit.w > 0) {
        const f3 test = mk3(hit.x, hit.y, hit.z);
        const f3 n = grad_group<Q>(v, test);
        if (length3(n) > 0) o = shade(test, normalize3(n), light, ambient);
    }
    if (writer) out[x + y * W] = o;
}

namespace launch {

static SkipGrid make_skip_grid(const unsigned char *flags, unsigned sx, unsigned sy, unsigned sz, f3 dim) {
    SkipGrid g;
    g.flags = flags;
    g.nbx = bricks_along(sx); g.nby = bricks_along(sy); g.nbz = bricks_along(sz);
    g.kx = sx / dim.x * (1.0f / BRICK); g.ky = sy / dim.y * (1.0f / BRICK); g.kz = sz / dim.z * (1.0f / BRICK);
    g.bx_max = (float)((sx - 1) >> BRICK_SHIFT); g.by_max = (float)((sy - 1) >> BRICK_SHIFT); g.bz_max = (float)((sz - 1) >> BRICK_SHIFT);
    static const long tp = env_long("KFH_RAY_TWO_PHASE", 0, 1, 0);
    g.two_phase = (flags && tp) ? 1 : 0;
#if KFH_RAY_PROBES
    static const long probe_env = env_long("KFH_RAY_PROBE_STAGE", 0, 7, 0);
    g.probe = (int)probe_env;
#else
    g.probe = 0;
#endif
    g.trace = nullptr;
    g.order = nullptr; g.cost = nullptr;
    g.tiles_x = 0; g.tiles_y = 0;
    g.next_state = nullptr; g.next_frame = 0; g.next_tracking_rate = 1;
    for (int i = 0; i < 4; ++i) g.next_k[i] = 0.f;
    return g;
}

size_t skip_grid_bytes(unsigned sx, unsigned sy, unsigned sz) { return (size_t)bricks_along(sx) * bricks_along(sy) * bricks_along(sz); }

void build_skip_grid(unsigned char *flags, const int16_t *vol, unsigned sx, unsigned sy, unsigned sz, hipStream_t s) {
    hipMemsetAsync(flags, 0, skip_grid_bytes(sx, sy, sz), s);
    k_build_skip_grid<<<2048, 256, 0, s>>>(flags, reinterpret_cast<const short2 *>(vol), sx, sy, sz, bricks_along(sx), bricks_along(sy), bricks_along(sz)); launched("k_build_skip_grid");
}

// all three axes verified and every numerator pos * size below 2^24
static bool fast_division(f3 dim, unsigned sx, unsigned sy, unsigned sz, hipStream_t s) {
    if (!((double)dim.x * sx < 8.0e6 && (double)dim.y * sy < 8.0e6 && (double)dim.z * sz < 8.0e6)) return false;
    return verified_divisor(dim.x, s) && verified_divisor(dim.y, s) && verified_divisor(dim.z, s);
}

void prepare_volume_division(f3 dim, unsigned sx, unsigned sy, unsigned sz, hipStream_t s) { (void)fast_division(dim, sx, sy, sz, s); }

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
void raycast(float4 *vertex, float4 *normal, unsigned W, unsigned H, const int16_t *vol, unsigned sx, unsigned sy,
             unsigned sz, f3 dim, const RayParams *p, float nearPlane, float farPlane, float step, float largestep,
             const unsigned char *skip_flags, unsigned *tile_order, unsigned *tile_cost, const NextTrackBegin *next, hipStream_t s) {
    if (!sizes_addressable(sx, sy, sz, W, H)) { refused("raycast: volume above 2^30 voxels (or an x-y plane of 2^24): outside the march's 32-bit voxel addressing"); return; }
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
    dispatch_order(tile_order, tile_cost, tg.x, tg.y, 0u, s);
}
void ray_tile_grid(unsigned W, unsigned H, unsigned *tx, unsigned *ty) { const dim3 g = ray_tile_dim(ray_group(W, H), W, H); *tx = g.x; *ty = g.y; }
bool ray_reorder_enabled() { static const bool on = [] { const char *e = getenv("KFH_RAY_REORDER"); return !(e && e[0] == '0'); }(); return on; }

unsigned ray_tiles(unsigned W, unsigned H) { const dim3 g = ray_tile_dim(ray_group(W, H), W, H); return g.x * g.y; }

void render_volume(uchar4 *out, unsigned W, unsigned H, const int16_t *vol, unsigned sx, unsigned sy, unsigned sz, f3 dim,
                   const RayParams *p, float nearPlane, float farPlane, float step, float largestep, f3 light, f3 ambient,
                   const unsigned char *skip_flags, const float4 *seen_vertex, const float4 *seen_normal, const RayParams *seen_rp, float seen_far_plane,
                   hipStream_t s) {
    if (!sizes_addressable(sx, sy, sz, W, H)) { refused("render_volume: volume above 2^30 voxels (or an x-y plane of 2^24): outside the march's 32-bit voxel addressing"); return; }
    const short2 *vv = reinterpret_cast<const short2 *>(vol);
    const SkipGrid sg = make_skip_grid(skip_flags, sx, sy, sz, dim);
    const f3 rdim = mk3(1.0f / dim.x, 1.0f / dim.y, 1.0f / dim.z);
    SeenRays seen;
    seen.vertex = seen_vertex; seen.normal = seen_normal; seen.rp = (seen_vertex && seen_normal) ? seen_rp : nullptr; seen.far_plane = seen_far_plane;
    if (fast_division(dim, sx, sy, sz, s)) {
        switch (ray_group(W, H)) {
        case 1: k_render_volume<1, 1><<<ray_grid(1, W, H), RAY_BLOCK, 0, s>>>(out, W, H, vv, sx, sy, sz, dim, p, nearPlane, farPlane, step, largestep, light, ambient, sg, rdim, seen); launched("k_render_volume"); break;
        case 2: k_render_volume<2, 1><<<ray_grid(2, W, H), RAY_BLOCK, 0, s>>>(out, W, H, vv, sx, sy, sz, dim, p, nearPlane, farPlane, step, largestep, light, ambient, sg, rdim, seen); launched("k_render_volume"); break;
        case 8: k_render_volume<8, 1><<<ray_grid(8, W, H), RAY_BLOCK, 0, s>>>(out, W, H, vv, sx, sy, sz, dim, p, nearPlane, farPlane, step, largestep, light, ambient, sg, rdim, seen); launched("k_render_volume"); break;
        default: k_render_volume<4, 1><<<ray_grid(4, W, H), RAY_BLOCK, 0, s>>>(out, W, H, vv, sx, sy, sz, dim, p, nearPlane, farPlane, step, largestep, light, ambient, sg, rdim, seen); launched("k_render_volume"); break;
        }
    } else {
        switch (ray_group(W, H)) {
        case 1: k_render_volume<1, 0><<<ray_grid(1, W, H), RAY_BLOCK, 0, s>>>(out, W, H, vv, sx, sy, sz, dim, p, nearPlane, farPlane, step, largestep, light, ambient, sg, rdim, seen); launched("k_render_volume"); break;
        case 2: k_render_volume<2, 0><<<ray_grid(2, W, H), RAY_BLOCK, 0, s>>>(out, W, H, vv, sx, sy, sz, dim, p, nearPlane, farPlane, step, largestep, light, ambient, sg, rdim, seen); launched("k_render_volume"); break;
        case 8: k_render_volume<8, 0><<<ray_grid(8, W, H), RAY_BLOCK, 0, s>>>(out, W, H, vv, sx, sy, sz, dim, p, nearPlane, farPlane, step, largestep, light, ambient, sg, rdim, seen); launched("k_render_volume"); break;
        default: k_render_volume<4, 0><<<ray_grid(4, W, H), RAY_BLOCK, 0, s>>>(out, W, H, vv, sx, sy, sz, dim, p, nearPlane, farPlane, step, largestep, light, ambient, sg, rdim, seen); launched("k_render_volume"); break;
        }
    }
}

}  // namespace launch
}  // namespace kf
