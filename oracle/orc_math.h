// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped HIP path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
//
// PARITY UNPINNED (kernels): the reference CPP backend (kfusion/src/cpp/kernels.cpp)
// cannot be compiled in this image because commons.h:43-45 includes the un-vendored
// third-party TooN (pinned by the reference Makefile:20-24 at git
// 92241416d2a4874fd2334e08a5d417dfea6a1a3f) and the reference ships no tests, fixtures
// or golden vectors.  Everything here is a restatement made by reading
// the reference; only the scalar/vector helper semantics below are pinned, against
// the reference's own thirdparty/cutil_math.h compiled as-is (oracle/_ref, see
// oracle/Makefile + tests/golden/cutil_math_vectors.json).
//
// Scalar / small-vector helpers and pose algebra used by the CPU restatement.
// Host semantics follow kfusion/thirdparty/cutil_math.h (the host branch,
// lines 36-62) and kfusion/include/commons.h:82-84,317-363.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace orc {

struct f2 { float x, y; };
struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };
struct s2 { short x, y; };
struct u4 { unsigned char x, y, z, w; };
// commons.h:317-319 — four float4 rows, row-major.
struct M4 { float m[4][4]; };
// commons.h:325-329
struct TrackRec { int result; float error; float J[6]; };

static const float kInvalid = -2.0f;  // KFUSION_INVALID

// cutil_math.h:43-49 — host fminf/fmaxf are plain ternaries (NaN falls to b).
static inline float fmin_h(float a, float b) { return a < b ? a : b; }
static inline float fmax_h(float a, float b) { return a > b ? a : b; }
// cutil_math.h:972-974
static inline float clamp_h(float f, float a, float b) { return fmax_h(a, fmin_h(f, b)); }
// cutil_math.h:51-57 (only int min/max exist on the host)
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
// commons.h:82-84
static inline float sq(float r) { return r * r; }

static inline f3 mk3(float x, float y, float z) { f3 r = {x, y, z}; return r; }
static inline f3 add3(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 mul3(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline f3 scale3(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
// cutil_math.h:1063-1071 (left-to-right sum)
static inline float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// cutil_math.h:1244-1247
static inline f3 cross3(f3 a, f3 b) {
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// cutil_math.h:1117-1125
static inline float length3(f3 v) { return sqrtf(dot3(v, v)); }
// cutil_math.h:59-61,1147-1152 — rsqrtf == 1/sqrtf on the host
static inline f3 normalize3(f3 v) {
    float inv = 1.0f / sqrtf(dot3(v, v));
    return scale3(v, inv);
}

// commons.h:331-341
static inline f3 xform(const M4 &M, f3 v) {
    return mk3(dot3(mk3(M.m[0][0], M.m[0][1], M.m[0][2]), v) + M.m[0][3],
               dot3(mk3(M.m[1][0], M.m[1][1], M.m[1][2]), v) + M.m[1][3],
               dot3(mk3(M.m[2][0], M.m[2][1], M.m[2][2]), v) + M.m[2][3]);
}
static inline f3 rotate(const M4 &M, f3 v) {
    return mk3(dot3(mk3(M.m[0][0], M.m[0][1], M.m[0][2]), v),
               dot3(mk3(M.m[1][0], M.m[1][1], M.m[1][2]), v),
               dot3(mk3(M.m[2][0], M.m[2][1], M.m[2][2]), v));
}

// commons.h:343-359
static inline M4 camera_matrix(f4 k) {
    M4 K; std::memset(&K, 0, sizeof K);
    K.m[0][0] = k.x; K.m[0][2] = k.z;
    K.m[1][1] = k.y; K.m[1][2] = k.w;
    K.m[2][2] = 1.0f; K.m[3][3] = 1.0f;
    return K;
}
static inline M4 inverse_camera_matrix(f4 k) {
    M4 K; std::memset(&K, 0, sizeof K);
    K.m[0][0] = 1.0f / k.x; K.m[0][2] = -k.z / k.x;
    K.m[1][1] = 1.0f / k.y; K.m[1][2] = -k.w / k.y;
    K.m[2][2] = 1.0f; K.m[3][3] = 1.0f;
    return K;
}

// ---- pose algebra: stands where the reference calls TooN (commons.h:365-412) ----
// TooN is absent from /root/reference; the algorithms are restated from its published
// sources (gaussian_elimination.h, se3.h, so3.h @ 92241416).  UNPINNED.

// commons.h:373-378 — 4x4 float product, inner index ascending, accumulator from 0.
static inline M4 matmul(const M4 &A, const M4 &B) {
    M4 R;
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            float acc = 0.0f;
            for (int k = 0; k < 4; ++k) acc += A.m[r][k] * B.m[k][c];
            R.m[r][c] = acc;
        }
    return R;
}

// commons.h:365-371 — TooN::gaussian_elimination(A, I): forward elimination with
// partial pivoting (row scaled by 1/pivot, the multiplier promoted to double as
// TooN does), then back substitution.  No singularity check: a zero matrix yields
// inf/NaN exactly as in the reference.
static inline M4 inverse(const M4 &Ain) {
    float A[4][4], B[4][4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) { A[i][j] = Ain.m[i][j]; B[i][j] = (i == j) ? 1.0f : 0.0f; }
    for (int i = 0; i < 4; ++i) {
        int piv = i;
        float best = fabsf(A[i][i]);
        for (int r = i + 1; r < 4; ++r) {
            double v = fabsf(A[r][i]);
            if (v > best) { best = (float)v; piv = r; }
        }
        float pivot = A[piv][i];
        float inv_pivot = 1.0f / pivot;
        if (piv != i) {
            for (int j = i; j < 4; ++j) { float t = A[i][j]; A[i][j] = A[piv][j]; A[piv][j] = t; }
            for (int j = 0; j < 4; ++j) { float t = B[i][j]; B[i][j] = B[piv][j]; B[piv][j] = t; }
        }
        for (int j = i + 1; j < 4; ++j) A[i][j] *= inv_pivot;
        for (int j = 0; j < 4; ++j) B[i][j] *= inv_pivot;
        for (int u = i + 1; u < 4; ++u) {
            double factor = A[u][i];
            for (int j = i + 1; j < 4; ++j) A[u][j] = (float)(A[u][j] - factor * A[i][j]);
            for (int j = 0; j < 4; ++j) B[u][j] = (float)(B[u][j] - factor * B[i][j]);
        }
    }
    M4 X;
    for (int i = 3; i >= 0; --i)
        for (int c = 0; c < 4; ++c) {
            float v = B[i][c];
            for (int j = i + 1; j < 4; ++j) v -= A[i][j] * X.m[j][c];
            X.m[i][c] = v;
        }
    return X;
}

// TooN::SE3<double>::exp (se3.h) + rodrigues_so3_exp (so3.h); commons.h:406-412
// materialises it as a float 4x4 ("p * Identity").  mu = (translation3, rotation3).
static inline M4 se3_exp_to_m4(const double mu[6]) {
    const double one_6th = 1.0 / 6.0, one_20th = 1.0 / 20.0;
    const double v[3] = {mu[0], mu[1], mu[2]}, w[3] = {mu[3], mu[4], mu[5]};
    const double theta_sq = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    const double theta = std::sqrt(theta_sq);
    const double cr[3] = {w[1] * v[2] - w[2] * v[1], w[2] * v[0] - w[0] * v[2], w[0] * v[1] - w[1] * v[0]};
    double A, B, t[3];
    if (theta_sq < 1e-8) {
        A = 1.0 - one_6th * theta_sq;
        B = 0.5;
        for (int i = 0; i < 3; ++i) t[i] = v[i] + 0.5 * cr[i];
    } else {
        double C;
        if (theta_sq < 1e-6) {
            C = one_6th * (1.0 - one_20th * theta_sq);
            A = 1.0 - theta_sq * C;
            B = 0.5 - 0.25 * one_6th * theta_sq;
        } else {
            const double inv_theta = 1.0 / theta;
            A = std::sin(theta) * inv_theta;
            B = (1 - std::cos(theta)) * (inv_theta * inv_theta);
            C = (1 - A) * (inv_theta * inv_theta);
        }
        const double wc[3] = {w[1] * cr[2] - w[2] * cr[1], w[2] * cr[0] - w[0] * cr[2], w[0] * cr[1] - w[1] * cr[0]};
        for (int i = 0; i < 3; ++i) t[i] = v[i] + B * cr[i] + C * wc[i];
    }
    double R[3][3];
    {
        const double wx2 = w[0] * w[0], wy2 = w[1] * w[1], wz2 = w[2] * w[2];
        R[0][0] = 1.0 - B * (wy2 + wz2);
        R[1][1] = 1.0 - B * (wx2 + wz2);
        R[2][2] = 1.0 - B * (wx2 + wy2);
    }
    { const double a = A * w[2], b = B * (w[0] * w[1]); R[0][1] = b - a; R[1][0] = b + a; }
    { const double a = A * w[1], b = B * (w[0] * w[2]); R[0][2] = b + a; R[2][0] = b - a; }
    { const double a = A * w[0], b = B * (w[1] * w[2]); R[1][2] = b - a; R[2][1] = b + a; }
    M4 out;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) out.m[r][c] = (float)R[r][c];
        out.m[r][3] = (float)t[r];
    }
    out.m[3][0] = out.m[3][1] = out.m[3][2] = 0.0f; out.m[3][3] = 1.0f;
    return out;
}

// commons.h:380-404 — makeJTJ + GR_SVD<6,6>(C).backsub(b, 1e6).  C is symmetric, so
// its SVD is its eigendecomposition up to signs; x = sum_i [ |l_i|*1e6 > max|l| ]
// (q_i.b / l_i) q_i.  Cyclic Jacobi in double.  vals = 27 floats: b[6], then the
// 21 upper-triangle entries of C, row-major.
static inline void solve6(const float vals[27], double x[6]) {
    double C[6][6], Q[6][6], b[6];
    for (int i = 0; i < 6; ++i) b[i] = vals[i];
    int p = 6;
    for (int r = 0; r < 6; ++r)
        for (int c = r; c < 6; ++c) { C[r][c] = vals[p]; C[c][r] = vals[p]; ++p; }
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) Q[i][j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < 6; ++i) {
            diag += C[i][i] * C[i][i];
            for (int j = i + 1; j < 6; ++j) off += C[i][j] * C[i][j];
        }
        if (off <= 1e-32 * diag || off == 0.0) break;
        for (int i = 0; i < 5; ++i)
            for (int j = i + 1; j < 6; ++j) {
                const double apq = C[i][j];
                if (apq == 0.0) continue;
                const double theta = (C[j][j] - C[i][i]) / (2.0 * apq);
                const double tt = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double cs = 1.0 / std::sqrt(tt * tt + 1.0), sn = tt * cs;
                for (int k = 0; k < 6; ++k) {
                    const double cki = C[k][i], ckj = C[k][j];
                    C[k][i] = cs * cki - sn * ckj;
                    C[k][j] = sn * cki + cs * ckj;
                }
                for (int k = 0; k < 6; ++k) {
                    const double cik = C[i][k], cjk = C[j][k];
                    C[i][k] = cs * cik - sn * cjk;
                    C[j][k] = sn * cik + cs * cjk;
                }
                for (int k = 0; k < 6; ++k) {
                    const double qki = Q[k][i], qkj = Q[k][j];
                    Q[k][i] = cs * qki - sn * qkj;
                    Q[k][j] = sn * qki + cs * qkj;
                }
            }
    }
    double smax = 0.0;
    for (int i = 0; i < 6; ++i) smax = std::fabs(C[i][i]) > smax ? std::fabs(C[i][i]) : smax;
    for (int i = 0; i < 6; ++i) x[i] = 0.0;
    for (int i = 0; i < 6; ++i) {
        const double l = C[i][i];
        if (!(std::fabs(l) * 1e6 > smax)) continue;
        double qb = 0.0;
        for (int k = 0; k < 6; ++k) qb += Q[k][i] * b[k];
        const double coef = qb / l;
        for (int k = 0; k < 6; ++k) x[k] += coef * Q[k][i];
    }
}

}  // namespace orc
