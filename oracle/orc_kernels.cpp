// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header).  PARITY UNPINNED.
//
// CPU restatement of the KinectFusion per-frame kernels of the reference CPP backend,
// /root/reference/kfusion/src/cpp/kernels.cpp (cited per function as cpp:LINE) and
// kfusion/include/commons.h (commons.h:LINE).  Built twice by oracle/Makefile:
// serial (liborc.so — the parity checker) and -fopenmp (liborc_omp.so — the timed CPU
// baseline, bench.py cpu_baseline.kind = "port").
//
// Deliberate deviations from the reference (each only where the reference is UB):
//  * "no reference frame yet" (raycastPose still the zero matrix) is an
//    explicit flag instead of NaN -> uint conversions; the flag reproduces what the
//    x86 build does (every pixel lands on reference pixel (0,0)).
//  * trilinear taps are clamped to [0,size-1] on both sides so that a ray that leaves
//    the volume through rounding cannot read out of bounds.
#include "orc_math.h"
#include "orc_api.h"
#include <cstdlib>
#include <cstdio>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

using namespace orc;

#ifdef _OPENMP
#define ORC_ROWS _Pragma("omp parallel for schedule(static)")
#else
#define ORC_ROWS
#endif

// ---------------------------------------------------------------- volume (commons.h:149-315)
namespace {

struct Vol {
    unsigned sx, sy, sz;
    f3 dim;
    s2 *data;
    orc_counters *cnt;      // optional instrumentation (U and V)
    unsigned char *touched; // optional: one byte per voxel, set when sampled
};

static const float kShort2Float = 0.00003051944088f;  // commons.h:162

inline size_t vidx(const Vol &v, unsigned x, unsigned y, unsigned z) {
    return (size_t)x + (size_t)y * v.sx + (size_t)z * v.sx * v.sy;
}
// commons.h:172-174 (vs2): raw short as float
inline float tap(const Vol &v, int x, int y, int z) {
    const size_t i = vidx(v, (unsigned)x, (unsigned)y, (unsigned)z);
    if (v.touched) v.touched[i] = 1;
    return (float)v.data[i].x;
}
inline int clampi(int a, int lo, int hi) { return imax(lo, imin(a, hi)); }

struct Taps {
    int lo[3], up[3], ll[3], uu[3];
    f3 fr;
};
// commons.h:193-200, 216-228: scaled position, floor/frac and the four index planes.
inline Taps locate(const Vol &v, f3 pos) {
    Taps t;
    const float sp[3] = {(pos.x * v.sx / v.dim.x) - 0.5f, (pos.y * v.sy / v.dim.y) - 0.5f,
                         (pos.z * v.sz / v.dim.z) - 0.5f};
    const int sz[3] = {(int)v.sx, (int)v.sy, (int)v.sz};
    float fr[3];
    for (int a = 0; a < 3; ++a) {
        const float fl = floorf(sp[a]);
        const int base = (int)fl;
        fr[a] = sp[a] - fl;                                // cutil_math.h:1178-1180
        t.ll[a] = clampi(imax(base - 1, 0), 0, sz[a] - 1);
        t.lo[a] = clampi(imax(base, 0), 0, sz[a] - 1);
        t.up[a] = clampi(imin(base + 1, sz[a] - 1), 0, sz[a] - 1);
        t.uu[a] = clampi(imin(base + 2, sz[a] - 1), 0, sz[a] - 1);
    }
    t.fr = mk3(fr[0], fr[1], fr[2]);
    return t;
}

// commons.h:191-213
inline float interp(const Vol &v, f3 pos) {
    const Taps t = locate(v, pos);
    const float fx = t.fr.x, fy = t.fr.y, fz = t.fr.z;
    const int *lo = t.lo, *up = t.up;
    return (((tap(v, lo[0], lo[1], lo[2]) * (1 - fx) + tap(v, up[0], lo[1], lo[2]) * fx) * (1 - fy)
             + (tap(v, lo[0], up[1], lo[2]) * (1 - fx) + tap(v, up[0], up[1], lo[2]) * fx) * fy) * (1 - fz)
            + ((tap(v, lo[0], lo[1], up[2]) * (1 - fx) + tap(v, up[0], lo[1], up[2]) * fx) * (1 - fy)
               + (tap(v, lo[0], up[1], up[2]) * (1 - fx) + tap(v, up[0], up[1], up[2]) * fx) * fy) * fz)
           * kShort2Float;
}

// One axis of commons.h:232-296.  The same bilinear-of-differences shape is used for
// x, y and z; `a` is the differentiated axis, (b,c) the other two in the order the
// reference blends them (x before y before z).
inline float grad_axis(const Vol &v, const Taps &t, int a) {
    const float f[3] = {t.fr.x, t.fr.y, t.fr.z};
    // d(sel_b, sel_c, hi) = v(upper*) - v(lower*) along axis a at the chosen corner.
    // hi=0: (upper_lower - lower_lower) weighted (1-f[a]); hi=1: (upper_upper - lower_upper) weighted f[a].
    auto diff = [&](int pb, int pc, int b, int c, int hi) -> float {
        int p[3], q[3];
        p[b] = q[b] = pb; p[c] = q[c] = pc;
        p[a] = hi ? t.uu[a] : t.up[a];
        q[a] = hi ? t.lo[a] : t.ll[a];
        return tap(v, p[0], p[1], p[2]) - tap(v, q[0], q[1], q[2]);
    };
    float r;
    if (a == 0) {
        // gradient.x: inner blend on x-differences themselves (weights 1-fx / fx), then y, then z
        const int *lo = t.lo, *up = t.up;
        r = ((diff(lo[1], lo[2], 1, 2, 0) * (1 - f[0]) + diff(lo[1], lo[2], 1, 2, 1) * f[0]) * (1 - f[1])
             + (diff(up[1], lo[2], 1, 2, 0) * (1 - f[0]) + diff(up[1], lo[2], 1, 2, 1) * f[0]) * f[1]) * (1 - f[2])
            + ((diff(lo[1], up[2], 1, 2, 0) * (1 - f[0]) + diff(lo[1], up[2], 1, 2, 1) * f[0]) * (1 - f[1])
               + (diff(up[1], up[2], 1, 2, 0) * (1 - f[0]) + diff(up[1], up[2], 1, 2, 1) * f[0]) * f[1]) * f[2];
    } else if (a == 1) {
        // gradient.y (commons.h:254-274): x-blend of y-differences; the y weights pick low/high difference
        const int *lo = t.lo, *up = t.up;
        r = ((diff(lo[0], lo[2], 0, 2, 0) * (1 - f[0]) + diff(up[0], lo[2], 0, 2, 0) * f[0]) * (1 - f[1])
             + (diff(lo[0], lo[2], 0, 2, 1) * (1 - f[0]) + diff(up[0], lo[2], 0, 2, 1) * f[0]) * f[1]) * (1 - f[2])
            + ((diff(lo[0], up[2], 0, 2, 0) * (1 - f[0]) + diff(up[0], up[2], 0, 2, 0) * f[0]) * (1 - f[1])
               + (diff(lo[0], up[2], 0, 2, 1) * (1 - f[0]) + diff(up[0], up[2], 0, 2, 1) * f[0]) * f[1]) * f[2];
    } else {
        // gradient.z (commons.h:276-296)
        const int *lo = t.lo, *up = t.up;
        r = ((diff(lo[0], lo[1], 0, 1, 0) * (1 - f[0]) + diff(up[0], lo[1], 0, 1, 0) * f[0]) * (1 - f[1])
             + (diff(lo[0], up[1], 0, 1, 0) * (1 - f[0]) + diff(up[0], up[1], 0, 1, 0) * f[0]) * f[1]) * (1 - f[2])
            + ((diff(lo[0], lo[1], 0, 1, 1) * (1 - f[0]) + diff(up[0], lo[1], 0, 1, 1) * f[0]) * (1 - f[1])
               + (diff(lo[0], up[1], 0, 1, 1) * (1 - f[0]) + diff(up[0], up[1], 0, 1, 1) * f[0]) * f[1]) * f[2];
    }
    return r;
}

// commons.h:215-301
inline f3 grad(const Vol &v, f3 pos) {
    const Taps t = locate(v, pos);
    const f3 g = mk3(grad_axis(v, t, 0), grad_axis(v, t, 1), grad_axis(v, t, 2));
    const f3 cell = mk3(v.dim.x / v.sx, v.dim.y / v.sy, v.dim.z / v.sz);
    return scale3(mul3(g, cell), 0.5f * kShort2Float);
}

static thread_local int g_ray_samples = 0;   // instrumentation: interp() calls of the last cast_ray

// cpp:674-725
inline f4 cast_ray(const Vol &v, unsigned px, unsigned py, const M4 &view, float nearPlane, float farPlane,
                   float step, float largestep) {
    const f3 origin = mk3(view.m[0][3], view.m[1][3], view.m[2][3]);
    const f3 dir = rotate(view, mk3((float)px, (float)py, 1.f));
    const f3 invR = mk3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
    const f3 tbot = mul3(scale3(invR, -1.0f), origin);   // (-1 * invR) * origin; s*a == a*s bitwise
    const f3 ttop = mul3(invR, sub3(v.dim, origin));
    const f3 tmin = mk3(fmin_h(ttop.x, tbot.x), fmin_h(ttop.y, tbot.y), fmin_h(ttop.z, tbot.z));
    const f3 tmax = mk3(fmax_h(ttop.x, tbot.x), fmax_h(ttop.y, tbot.y), fmax_h(ttop.z, tbot.z));
    const float largest_tmin = fmax_h(fmax_h(tmin.x, tmin.y), fmax_h(tmin.x, tmin.z));
    const float smallest_tmax = fmin_h(fmin_h(tmax.x, tmax.y), fmin_h(tmax.x, tmax.z));
    const float tnear = fmax_h(largest_tmin, nearPlane);
    const float tfar = fmin_h(smallest_tmax, farPlane);
    f4 miss = {0.f, 0.f, 0.f, 0.f};
    if (tnear < tfar) {
        float t = tnear, stepsize = largestep;
        g_ray_samples = 1;
        float f_t = interp(v, add3(origin, scale3(dir, t)));
        float f_tt = 0;
        if (f_t > 0) {
            for (; t < tfar; t += stepsize) {
                ++g_ray_samples;
                f_tt = interp(v, add3(origin, scale3(dir, t)));
                if (f_tt < 0) break;
                if (f_tt < 0.8f) stepsize = step;
                f_t = f_tt;
            }
            if (f_tt < 0) {
                t = t + stepsize * f_tt / (f_t - f_tt);
                const f3 p = add3(origin, scale3(dir, t));
                f4 hit = {p.x, p.y, p.z, t};
                return hit;
            }
        }
    }
    return miss;
}

inline Vol make_vol(short *data, const unsigned size[3], const float dim[3], orc_counters *cnt) {
    Vol v;
    v.sx = size[0]; v.sy = size[1]; v.sz = size[2];
    v.dim = mk3(dim[0], dim[1], dim[2]);
    v.data = reinterpret_cast<s2 *>(data);
    v.cnt = cnt;
    v.touched = nullptr;
    return v;
}

inline M4 load_m4(const float *p) { M4 m; std::memcpy(&m, p, sizeof m); return m; }

// commons.h:86-147 — double-precision false-colour ramp.
inline u4 gs2rgb(double h) {
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
    u4 o;
    o.x = (unsigned char)(r * 255); o.y = (unsigned char)(g * 255); o.z = (unsigned char)(b * 255); o.w = 0;
    return o;
}

}  // namespace

extern "C" {

// cpp:100-108
void orc_gaussian(float *g, int radius, float delta) {
    for (int i = 0; i < 2 * radius + 1; ++i) {
        const int x = i - 2;   // the reference hard-codes 2 here, not `radius`
        g[i] = expf(-(x * x) / (2 * delta * delta));
    }
}

// cpp:147-157 + commons.h:176-180
void orc_init_volume(short *vol, unsigned sx, unsigned sy, unsigned sz) {
    s2 *d = reinterpret_cast<s2 *>(vol);
    const s2 init = {(short)(1.0f * 32766.0f), (short)0.0f};
    const size_t n = (size_t)sx * sy * sz;
    for (size_t i = 0; i < n; ++i) d[i] = init;
}

// cpp:562-589.  Returns 1 where the reference prints "Invalid ratio." and exit(1)s.
int orc_mm2meters(float *out, unsigned outW, unsigned outH, const unsigned short *in, unsigned inW, unsigned inH) {
    if (inW < outW || inH < outH) return 1;
    if (inW % outW != 0 || inH % outH != 0) return 1;
    if (inW / outW != inH / outH) return 1;
    const int ratio = inW / outW;
    ORC_ROWS
    for (unsigned y = 0; y < outH; ++y)
        for (unsigned x = 0; x < outW; ++x)
            out[x + outW * y] = in[x * ratio + inW * y * ratio] / 1000.0f;
    return 0;
}

// cpp:159-198.  Border taps clamp to the edge on both sides.
void orc_bilateral_filter(float *out, const float *in, unsigned W, unsigned H, const float *gaussian, float e_d, int r) {
    const float e_d_squared_2 = e_d * e_d * 2;
    ORC_ROWS
    for (unsigned y = 0; y < H; ++y)
        for (unsigned x = 0; x < W; ++x) {
            const unsigned pos = x + y * W;
            if (in[pos] == 0) { out[pos] = 0; continue; }
            float sum = 0.0f, t = 0.0f;
            const float center = in[pos];
            for (int i = -r; i <= r; ++i)
                for (int j = -r; j <= r; ++j) {
                    const int cx = imax(0, imin((int)x + i, (int)W - 1));
                    const int cy = imax(0, imin((int)y + j, (int)H - 1));
                    const float cur = in[cx + cy * W];
                    if (cur > 0) {
                        const float mod = sq(cur - center);
                        const float factor = gaussian[i + r] * gaussian[j + r] * expf(-mod / e_d_squared_2);
                        t += factor * cur;
                        sum += factor;
                    }
                }
            out[pos] = t / sum;
        }
}

// cpp:591-626
void orc_half_sample(float *out, const float *in, unsigned inW, unsigned inH, float e_d, int r) {
    const unsigned outW = inW / 2, outH = inH / 2;
    ORC_ROWS
    for (unsigned y = 0; y < outH; ++y)
        for (unsigned x = 0; x < outW; ++x) {
            const int cx = 2 * x, cy = 2 * y;
            float sum = 0.0f, t = 0.0f;
            const float center = in[cx + cy * inW];
            for (int i = -r + 1; i <= r; ++i)
                for (int j = -r + 1; j <= r; ++j) {
                    const int px = imax(0, imin(cx + j, 2 * (int)outW - 1));
                    const int py = imax(0, imin(cy + i, 2 * (int)outH - 1));
                    const float cur = in[px + py * inW];
                    if (fabsf(cur - center) < e_d) { sum += 1.0f; t += cur; }
                }
            out[x + y * outW] = t / sum;
        }
}

// cpp:200-218
void orc_depth2vertex(float *vertex, const float *depth, unsigned W, unsigned H, const float *invK16) {
    const M4 invK = load_m4(invK16);
    f3 *v = reinterpret_cast<f3 *>(vertex);
    ORC_ROWS
    for (unsigned y = 0; y < H; ++y)
        for (unsigned x = 0; x < W; ++x) {
            const float d = depth[x + y * W];
            v[x + y * W] = d > 0 ? scale3(rotate(invK, mk3((float)x, (float)y, 1.f)), d) : mk3(0, 0, 0);
        }
}

// cpp:220-249.  Invalid pixels write only .x — out is read-modify-write.
void orc_vertex2normal(float *out, const float *in, unsigned W, unsigned H) {
    f3 *o = reinterpret_cast<f3 *>(out);
    const f3 *v = reinterpret_cast<const f3 *>(in);
    ORC_ROWS
    for (unsigned y = 0; y < H; ++y)
        for (unsigned x = 0; x < W; ++x) {
            const unsigned xl = imax((int)x - 1, 0), xr = imin((int)x + 1, (int)W - 1);
            const unsigned yu = imax((int)y - 1, 0), yd = imin((int)y + 1, (int)H - 1);
            const f3 left = v[xl + W * y], right = v[xr + W * y], up = v[x + W * yu], down = v[x + W * yd];
            if (left.z == 0 || right.z == 0 || up.z == 0 || down.z == 0) { o[x + y * W].x = kInvalid; continue; }
            const f3 dxv = sub3(right, left), dyv = sub3(down, up);
            o[x + y * W] = normalize3(cross3(dyv, dxv));
        }
}

// cpp:497-560.  Output row stride is refW even on coarse levels (cpp:512).
void orc_track(void *output, const float *inVertex, const float *inNormal, unsigned inW, unsigned inH,
               const float *refVertex, const float *refNormal, unsigned refW, unsigned refH, const float *Ttrack16,
               const float *view16, float dist_threshold, float normal_threshold, int have_reference) {
    TrackRec *out = static_cast<TrackRec *>(output);
    const f3 *iv = reinterpret_cast<const f3 *>(inVertex), *in = reinterpret_cast<const f3 *>(inNormal);
    const f3 *rv = reinterpret_cast<const f3 *>(refVertex), *rn = reinterpret_cast<const f3 *>(refNormal);
    const M4 T = load_m4(Ttrack16), view = load_m4(view16);
    ORC_ROWS
    for (unsigned py = 0; py < inH; ++py)
        for (unsigned px = 0; px < inW; ++px) {
            TrackRec &row = out[px + py * refW];
            const f3 n_in = in[px + py * inW];
            if (n_in.x == kInvalid) { row.result = -1; continue; }
            const f3 pv = xform(T, iv[px + py * inW]);
            unsigned rx = 0, ry = 0;
            if (have_reference) {
                const f3 pp = xform(view, pv);
                const float u = pp.x / pp.z + 0.5f, w = pp.y / pp.z + 0.5f;
                if (u < 0 || u > refW - 1 || w < 0 || w > refH - 1) { row.result = -2; continue; }
                rx = (unsigned)u; ry = (unsigned)w;
            }
            const f3 n_ref = rn[rx + ry * refW];
            if (n_ref.x == kInvalid) { row.result = -3; continue; }
            const f3 diff = sub3(rv[rx + ry * refW], pv);
            const f3 pn = rotate(T, n_in);
            if (length3(diff) > dist_threshold) { row.result = -4; continue; }
            if (dot3(pn, n_ref) < normal_threshold) { row.result = -5; continue; }
            row.result = 1;
            row.error = dot3(n_ref, diff);
            row.J[0] = n_ref.x; row.J[1] = n_ref.y; row.J[2] = n_ref.z;
            const f3 c = cross3(pv, n_ref);
            row.J[3] = c.x; row.J[4] = c.y; row.J[5] = c.z;
        }
}

// cpp:251-390 (new_reduce) + cpp:391-495.  Block b owns rows b, b+8, ...; x ascending;
// float32 running sums; rows 1..7 then added into row 0.  The OpenMP twin parallelises
// over the 8 blocks only, which keeps this order (the reference's own OpenMP reduction
// clause does not: SURVEY N8).
// How the -fopenmp build parallelises the reduction.  0 (default): over the 8 row classes - every class keeps its serial
// chain, so the result is bit-identical to the serial build (what the parity tests need).  1: as the reference's OpenMP
// build does it (cpp:295-296, `#pragma omp parallel for reduction(+:sums0..31)` over the rows of a class): each thread
// adds up its share of the rows and the shares are combined - a re-associated fp32 sum that depends on the thread count
// (SURVEY N8).  Mode 1 is what tests/test_hip_fullsize.py measures the reference's own CPP-vs-OpenMP noise floor with.
static int g_reduce_mode = 0;
void orc_set_omp_reduce_mode(int reference_style) { g_reduce_mode = reference_style ? 1 : 0; }
// threads of the -fopenmp build from now on (0 = leave as is); the re-associated reduction depends on the count
void orc_set_omp_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

// Mode 2 (either build): the SAME re-association with a FIXED schedule - what `#pragma omp parallel for reduction(+:sums0..31)`
// (cpp:295-296) computes with T threads under gcc's static schedule (iteration space of a row class cut into T contiguous
// shares, the first n % T of them one longer; every share summed from 0.0f in row order) when the shares are combined in thread
// order 0..T-1.  libgomp combines them in ARRIVAL order, which is the one thing that varies run to run; fixing it makes the
// reference's own CPP-vs-OpenMP noise floor a reproducible number (tests/test_hip_fullsize.py, long-sequence parity).
static int g_reduce_emulated_threads = 0;
void orc_set_reduce_emulation(int threads) { g_reduce_emulated_threads = threads > 0 ? threads : 0; }

static void reduce_rows_emulated_omp(float *out, const TrackRec *J, unsigned JW, unsigned W, unsigned H, int T) {
    for (int b = 0; b < 8; ++b) {
        const unsigned n = H > (unsigned)b ? (H - (unsigned)b + 7) / 8 : 0;      // iterations y = b, b + 8, ... < H
        float s[32];
        for (int i = 0; i < 32; ++i) s[i] = 0.0f;
        const unsigned q = n / (unsigned)T, r = n % (unsigned)T;
        unsigned first = 0;
        for (int t = 0; t < T; ++t) {
            const unsigned count = q + ((unsigned)t < r ? 1u : 0u);
            float p[32];
            for (int i = 0; i < 32; ++i) p[i] = 0.0f;
            for (unsigned it = first; it < first + count; ++it) {
                const unsigned y = (unsigned)b + 8 * it;
                for (unsigned x = 0; x < W; ++x) {
                    const TrackRec &row = J[x + y * JW];
                    if (row.result < 1) {
                        p[29] += row.result == -4 ? 1 : 0;
                        p[30] += row.result == -5 ? 1 : 0;
                        p[31] += row.result > -4 ? 1 : 0;
                        continue;
                    }
                    p[0] += row.error * row.error;
                    for (int i = 0; i < 6; ++i) p[1 + i] += row.error * row.J[i];
                    int k = 7;
                    for (int rr = 0; rr < 6; ++rr)
                        for (int c = rr; c < 6; ++c) p[k++] += row.J[rr] * row.J[c];
                    p[28] += 1;
                }
            }
            first += count;
            for (int i = 0; i < 32; ++i) s[i] += p[i];       // (a thread without iterations contributes its zeros, as in OpenMP)
        }
        for (int i = 0; i < 32; ++i) out[b * 32 + i] = s[i];
    }
    for (int j = 1; j < 8; ++j)
        for (int i = 0; i < 32; ++i) out[i] += out[j * 32 + i];
}

#ifdef _OPENMP
static void reduce_rows_reference_omp(float *out, const TrackRec *J, unsigned JW, unsigned W, unsigned H) {
    for (int b = 0; b < 8; ++b) {
        float s[32];
        for (int i = 0; i < 32; ++i) s[i] = 0.0f;
#pragma omp parallel for schedule(static) reduction(+ : s[:32])
        for (unsigned y = b; y < H; y += 8)
            for (unsigned x = 0; x < W; ++x) {
                const TrackRec &row = J[x + y * JW];
                if (row.result < 1) {
                    s[29] += row.result == -4 ? 1 : 0;
                    s[30] += row.result == -5 ? 1 : 0;
                    s[31] += row.result > -4 ? 1 : 0;
                    continue;
                }
                s[0] += row.error * row.error;
                for (int i = 0; i < 6; ++i) s[1 + i] += row.error * row.J[i];
                int k = 7;
                for (int r = 0; r < 6; ++r)
                    for (int c = r; c < 6; ++c) s[k++] += row.J[r] * row.J[c];
                s[28] += 1;
            }
        for (int i = 0; i < 32; ++i) out[b * 32 + i] = s[i];
    }
    for (int j = 1; j < 8; ++j)
        for (int i = 0; i < 32; ++i) out[i] += out[j * 32 + i];
}
#endif

void orc_reduce(float *out, const void *Jv, unsigned JW, unsigned JH, unsigned W, unsigned H) {
    (void)JH;
    const TrackRec *J = static_cast<const TrackRec *>(Jv);
    if (g_reduce_emulated_threads > 0) { reduce_rows_emulated_omp(out, J, JW, W, H, g_reduce_emulated_threads); return; }
#ifdef _OPENMP
    if (g_reduce_mode) { reduce_rows_reference_omp(out, J, JW, W, H); return; }
#pragma omp parallel for schedule(static, 1)
#endif
    for (int b = 0; b < 8; ++b) {
        float s[32];
        for (int i = 0; i < 32; ++i) s[i] = 0.0f;
        for (unsigned y = b; y < H; y += 8)
            for (unsigned x = 0; x < W; ++x) {
                const TrackRec &row = J[x + y * JW];
                if (row.result < 1) {
                    s[29] += row.result == -4 ? 1 : 0;
                    s[30] += row.result == -5 ? 1 : 0;
                    s[31] += row.result > -4 ? 1 : 0;
                    continue;
                }
                s[0] += row.error * row.error;
                for (int i = 0; i < 6; ++i) s[1 + i] += row.error * row.J[i];
                int k = 7;
                for (int r = 0; r < 6; ++r)
                    for (int c = r; c < 6; ++c) s[k++] += row.J[r] * row.J[c];
                s[28] += 1;
            }
        for (int i = 0; i < 32; ++i) out[b * 32 + i] = s[i];
    }
    for (int j = 1; j < 8; ++j)
        for (int i = 0; i < 32; ++i) out[i] += out[j * 32 + i];
}

// cpp:759-775
int orc_update_pose(float *pose16, const float *output, float icp_threshold) {
    double x[6];
    solve6(output + 1, x);
    const M4 delta = se3_exp_to_m4(x);
    const M4 p = matmul(delta, load_m4(pose16));
    std::memcpy(pose16, &p, sizeof p);
    double n2 = 0;
    for (int i = 0; i < 6; ++i) n2 += x[i] * x[i];
    return std::sqrt(n2) < icp_threshold ? 1 : 0;
}

// cpp:777-792
int orc_check_pose(float *pose16, const float *oldPose16, const float *output, unsigned W, unsigned H, float track_threshold) {
    if ((std::sqrt(output[0] / output[28]) > 2e-2) || (output[28] / (W * H) < track_threshold)) {
        std::memcpy(pose16, oldPose16, sizeof(M4));
        return 0;
    }
    return 1;
}

void orc_solve6(const float *vals27, double *x6) { solve6(vals27, x6); }

// Scalar / vector helpers exported one by one so that tests/test_oracle_golden.py can hold them against the
// vectors produced by the reference's own cutil_math.h (oracle/_ref/cutil_probe).
float orc_h_fminf(float a, float b) { return fmin_h(a, b); }
float orc_h_fmaxf(float a, float b) { return fmax_h(a, b); }
float orc_h_clampf(float f, float a, float b) { return clamp_h(f, a, b); }
int orc_h_border_clamp(int x, int i, int size) { return imax(0, imin(x + i, size - 1)); }   // the index rule of orc_bilateral_filter
int orc_h_clamp_int(int v, int lo, int hi) { return imax(lo, imin(v, hi)); }
float orc_h_dot3(const float *a, const float *b) { return dot3(mk3(a[0], a[1], a[2]), mk3(b[0], b[1], b[2])); }
void orc_h_cross3(const float *a, const float *b, float *o) { const f3 c = cross3(mk3(a[0], a[1], a[2]), mk3(b[0], b[1], b[2])); o[0] = c.x; o[1] = c.y; o[2] = c.z; }
float orc_h_length3(const float *a) { return length3(mk3(a[0], a[1], a[2])); }
void orc_h_normalize3(const float *a, float *o) { const f3 c = normalize3(mk3(a[0], a[1], a[2])); o[0] = c.x; o[1] = c.y; o[2] = c.z; }
void orc_h_floor_frac(float v, float *fl, float *fr, int *as_int) { const float f = floorf(v); *fl = f; *fr = v - f; *as_int = (int)v; }
void orc_h_constants(float *out12) {
    // e_delta, dist_threshold, normal_threshold, track_threshold, maxweight, nearPlane, farPlane, delta are
    // duplicated in orc_pipeline.cpp; the values used by the kernels' callers come from there
    extern const float orc_constants_table[12];
    for (int i = 0; i < 12; ++i) out12[i] = orc_constants_table[i];
}
void orc_se3_exp(const double *mu6, float *out16) { const M4 m = se3_exp_to_m4(mu6); std::memcpy(out16, &m, sizeof m); }
void orc_inverse(const float *in16, float *out16) { const M4 m = inverse(load_m4(in16)); std::memcpy(out16, &m, sizeof m); }
void orc_matmul(const float *a16, const float *b16, float *out16) { const M4 m = matmul(load_m4(a16), load_m4(b16)); std::memcpy(out16, &m, sizeof m); }

// cpp:628-673
void orc_integrate(short *vol, const unsigned size[3], const float dim[3], const float *depth, unsigned W, unsigned H,
                   const float *invTrack16, const float *K16, float mu, float maxweight, orc_counters *cnt) {
    Vol v = make_vol(vol, size, dim, cnt);
    const M4 invTrack = load_m4(invTrack16), K = load_m4(K16);
    const f3 delta = rotate(invTrack, mk3(0, 0, v.dim.z / v.sz));
    const f3 cameraDelta = rotate(K, delta);
    unsigned long long updated = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) reduction(+ : updated)
#endif
    for (unsigned y = 0; y < v.sy; ++y)
        for (unsigned x = 0; x < v.sx; ++x) {
            // commons.h:186-189 (Volume::pos)
            f3 pos = xform(invTrack, mk3((x + 0.5f) * v.dim.x / v.sx, (y + 0.5f) * v.dim.y / v.sy, (0 + 0.5f) * v.dim.z / v.sz));
            f3 cam = xform(K, pos);
            for (unsigned z = 0; z < v.sz; ++z, pos = add3(pos, delta), cam = add3(cam, cameraDelta)) {
                if (pos.z < 0.0001f) continue;
                const float px = cam.x / cam.z + 0.5f, py = cam.y / cam.z + 0.5f;
                if (px < 0 || px > W - 1 || py < 0 || py > H - 1) continue;
                const unsigned ux = (unsigned)px, uy = (unsigned)py;
                const float d = depth[ux + uy * W];
                if (d == 0) continue;
                const float diff = (d - cam.z) * sqrtf(1 + sq(pos.x / pos.z) + sq(pos.y / pos.z));
                if (diff > -mu) {
                    const float sdf = fmin_h(1.f, diff / mu);
                    s2 &cell = v.data[vidx(v, x, y, z)];
                    float tsdf = cell.x * kShort2Float, w = cell.y;
                    tsdf = clamp_h((w * tsdf + sdf) / (w + 1), -1.f, 1.f);
                    w = fmin_h(w + 1, maxweight);
                    cell.x = (short)(tsdf * 32766.0f);
                    cell.y = (short)w;
                    ++updated;
                }
            }
        }
    if (cnt) cnt->integrate_updated += updated;
}

// cpp:726-757
void orc_raycast(float *vertex, float *normal, unsigned W, unsigned H, const short *vol, const unsigned size[3],
                 const float dim[3], const float *view16, float nearPlane, float farPlane, float step, float largestep,
                 orc_counters *cnt) {
    Vol v = make_vol(const_cast<short *>(vol), size, dim, cnt);
    std::vector<unsigned char> touched;
    if (cnt) { touched.assign((size_t)v.sx * v.sy * v.sz, 0); v.touched = touched.data(); }
    const M4 view = load_m4(view16);
    f3 *vo = reinterpret_cast<f3 *>(vertex), *no = reinterpret_cast<f3 *>(normal);
    ORC_ROWS
    for (unsigned y = 0; y < H; ++y)
        for (unsigned x = 0; x < W; ++x) {
            const f4 hit = cast_ray(v, x, y, view, nearPlane, farPlane, step, largestep);
            const unsigned p = x + y * W;
            if (hit.w > 0.0) {
                const f3 h = mk3(hit.x, hit.y, hit.z);
                vo[p] = h;
                const f3 n = grad(v, h);
                if (length3(n) == 0) no[p].x = kInvalid;
                else no[p] = normalize3(n);
            } else {
                vo[p] = mk3(0, 0, 0);
                no[p] = mk3(kInvalid, 0, 0);
            }
        }
    if (cnt) {
        unsigned long long n = 0;
        for (size_t i = 0; i < touched.size(); ++i) n += touched[i];
        cnt->raycast_distinct_voxels = n;
    }
}

// instrumentation only: number of interp() samples each ray of a raycast takes (same march as orc_raycast)
void orc_raycast_samples(int *samples, unsigned W, unsigned H, const short *vol, const unsigned size[3], const float dim[3],
                         const float *view16, float nearPlane, float farPlane, float step, float largestep) {
    Vol v = make_vol(const_cast<short *>(vol), size, dim, nullptr);
    const M4 view = load_m4(view16);
    for (unsigned y = 0; y < H; ++y)
        for (unsigned x = 0; x < W; ++x) {
            g_ray_samples = 0;
            cast_ray(v, x, y, view, nearPlane, farPlane, step, largestep);
            samples[x + y * W] = g_ray_samples;
        }
}

// cpp:814-842
void orc_render_depth(unsigned char *out4, const float *depth, unsigned W, unsigned H, float nearPlane, float farPlane) {
    u4 *out = reinterpret_cast<u4 *>(out4);
    const float rangeScale = 1 / (farPlane - nearPlane);
    ORC_ROWS
    for (unsigned y = 0; y < H; ++y)
        for (unsigned x = 0; x < W; ++x) {
            const unsigned p = y * W + x;
            const u4 white = {255, 255, 255, 0}, black = {0, 0, 0, 0};
            if (depth[p] < nearPlane) out[p] = white;
            else if (depth[p] > farPlane) out[p] = black;
            else out[p] = gs2rgb((depth[p] - nearPlane) * rangeScale);
        }
}

// cpp:844-878
void orc_render_track(unsigned char *out4, const void *data, unsigned W, unsigned H) {
    u4 *out = reinterpret_cast<u4 *>(out4);
    const TrackRec *d = static_cast<const TrackRec *>(data);
    static const u4 table[7] = {{255, 128, 128, 0}, {128, 128, 128, 0}, {0, 0, 0, 0}, {255, 0, 0, 0},
                                {0, 255, 0, 0},     {0, 0, 255, 0},     {255, 255, 0, 0}};
    ORC_ROWS
    for (unsigned y = 0; y < H; ++y)
        for (unsigned x = 0; x < W; ++x) {
            const int r = d[x + y * W].result;
            int k = 0;
            if (r == 1) k = 1;
            else if (r <= -1 && r >= -5) k = 1 - r;
            out[x + y * W] = table[k];
        }
}

// cpp:880-913
void orc_render_volume(unsigned char *out4, unsigned W, unsigned H, const short *vol, const unsigned size[3],
                       const float dim[3], const float *view16, float nearPlane, float farPlane, float step,
                       float largestep, const float light3[3], const float ambient3[3]) {
    Vol v = make_vol(const_cast<short *>(vol), size, dim, nullptr);
    const M4 view = load_m4(view16);
    const f3 light = mk3(light3[0], light3[1], light3[2]), ambient = mk3(ambient3[0], ambient3[1], ambient3[2]);
    u4 *out = reinterpret_cast<u4 *>(out4);
    ORC_ROWS
    for (unsigned y = 0; y < H; ++y)
        for (unsigned x = 0; x < W; ++x) {
            const unsigned p = x + y * W;
            const u4 black = {0, 0, 0, 0};
            const f4 hit = cast_ray(v, x, y, view, nearPlane, farPlane, step, largestep);
            if (hit.w > 0) {
                const f3 test = mk3(hit.x, hit.y, hit.z);
                const f3 n = grad(v, test);
                if (length3(n) > 0) {
                    const f3 diff = normalize3(sub3(light, test));
                    const float dir = fmax_h(dot3(normalize3(n), diff), 0.f);
                    const f3 col = scale3(mk3(clamp_h(dir + ambient.x, 0.f, 1.f), clamp_h(dir + ambient.y, 0.f, 1.f),
                                              clamp_h(dir + ambient.z, 0.f, 1.f)), 255);
                    const u4 c = {(unsigned char)col.x, (unsigned char)col.y, (unsigned char)col.z, 0};
                    out[p] = c;
                } else out[p] = black;
            } else out[p] = black;
        }
}

}  // extern "C"
