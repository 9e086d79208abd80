// kf_math.h — small-vector and pose algebra for the HIP backend (host + device).
//
// Arithmetic follows the reference CPP backend operation by operation so that, built with
// -ffp-contract=off and correctly rounded fp32 divide/sqrt, the device produces the same bits as
// the host code it replaces: kfusion/thirdparty/cutil_math.h (host branch: fminf/fmaxf are plain
// ternaries, rsqrtf = 1/sqrtf, dot sums left to right) and kfusion/include/commons.h:317-412.
// The pose algebra stands where the reference calls TooN (gaussian_elimination, SE3::exp,
// GR_SVD::backsub) — restated from TooN's published sources, since TooN is not vendored.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#define KF_HD __host__ __device__ __forceinline__
// KFH_SOLVE_STAMPS=1 (diagnostic builds only): cycle stamps inside the solve / pose update of thread 0, written behind the
// stamps of KFH_ICP_TRACE (profiles/icp_trace.py prints them)
#ifndef KFH_SOLVE_STAMPS
#define KFH_SOLVE_STAMPS 0
#endif
#if KFH_SOLVE_STAMPS && defined(__HIP_DEVICE_COMPILE__)
#define KF_SOLVE_STAMP(tr, i, dep) do { if (tr) { asm volatile("" : : "v"(dep)); (tr)[i] = __builtin_readcyclecounter(); } } while (0)
#else
#define KF_SOLVE_STAMP(tr, i, dep) do { } while (0)
#endif
#ifndef KFH_SOLVE_LEAN
#define KFH_SOLVE_LEAN 1      // leaner fp64 solve / exponential / pose product on the single-thread path between two ICP associations
#endif

namespace kf {

struct f3 { float x, y, z; };
struct M4 { float m[4][4]; };

constexpr float kInvalid = -2.0f;                     // KFUSION_INVALID
constexpr float kShort2Float = 0.00003051944088f;     // commons.h:162

KF_HD float fmin_h(float a, float b) { return a < b ? a : b; }   // cutil_math.h:43-45
KF_HD float fmax_h(float a, float b) { return a > b ? a : b; }   // cutil_math.h:47-49
KF_HD float clamp_h(float f, float a, float b) { return fmax_h(a, fmin_h(f, b)); }
KF_HD int imin(int a, int b) { return a < b ? a : b; }
KF_HD int imax(int a, int b) { return a > b ? a : b; }
KF_HD float sq(float r) { return r * r; }

KF_HD f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
KF_HD f3 add3(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
KF_HD f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
KF_HD f3 mul3(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
KF_HD f3 scale3(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
KF_HD float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
KF_HD f3 cross3(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
KF_HD float length3(f3 v) { return sqrtf(dot3(v, v)); }
KF_HD f3 normalize3(f3 v) { const float inv = 1.0f / sqrtf(dot3(v, v)); return scale3(v, inv); }

// sqrtf(d) > t decided without the square root: the largest float X with sqrtf(X) <= t, so that for EVERY float d (NaN included,
// both comparisons are then false)  sqrtf(d) > t  <=>  d > X  - the correctly rounded square root is monotonic.  Found from t * t by
// stepping through neighbouring floats with the host's sqrtf (IEEE); the device's own square root is held to it for all 2^31
// non-negative floats before the bound is used (k_verify_sqrt_bound).  Returns -1 when no such X >= 0 exists (t < 0 or NaN).
inline float sqrt_gt_bound(float t) {
    if (!(t >= 0.0f) || !(t < 1.8e19f)) return -1.0f;       // (t * t must stay finite)
    float x = t * t;
    for (int i = 0; i < 8 && sqrtf(x) > t; ++i) x = nextafterf(x, -1.0f);
    for (int i = 0; i < 8 && sqrtf(nextafterf(x, INFINITY)) <= t; ++i) x = nextafterf(x, INFINITY);
    return (sqrtf(x) <= t && sqrtf(nextafterf(x, INFINITY)) > t) ? x : -1.0f;
}

// commons.h:331-341
KF_HD f3 xform(const M4 &M, f3 v) {
    return mk3(dot3(mk3(M.m[0][0], M.m[0][1], M.m[0][2]), v) + M.m[0][3],
               dot3(mk3(M.m[1][0], M.m[1][1], M.m[1][2]), v) + M.m[1][3],
               dot3(mk3(M.m[2][0], M.m[2][1], M.m[2][2]), v) + M.m[2][3]);
}
KF_HD f3 rotate(const M4 &M, f3 v) {
    return mk3(dot3(mk3(M.m[0][0], M.m[0][1], M.m[0][2]), v), dot3(mk3(M.m[1][0], M.m[1][1], M.m[1][2]), v),
               dot3(mk3(M.m[2][0], M.m[2][1], M.m[2][2]), v));
}

KF_HD M4 zero4() { M4 r; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[i][j] = 0.0f; return r; }
KF_HD M4 identity4() { M4 r = zero4(); for (int i = 0; i < 4; ++i) r.m[i][i] = 1.0f; return r; }
// commons.h:343-359
KF_HD M4 camera_matrix(float fx, float fy, float cx, float cy) {
    M4 K = zero4();
    K.m[0][0] = fx; K.m[0][2] = cx; K.m[1][1] = fy; K.m[1][2] = cy; K.m[2][2] = 1.0f; K.m[3][3] = 1.0f;
    return K;
}
KF_HD M4 inverse_camera_matrix(float fx, float fy, float cx, float cy) {
    M4 K = zero4();
    K.m[0][0] = 1.0f / fx; K.m[0][2] = -cx / fx; K.m[1][1] = 1.0f / fy; K.m[1][2] = -cy / fy;
    K.m[2][2] = 1.0f; K.m[3][3] = 1.0f;
    return K;
}

// commons.h:373-378
KF_HD M4 matmul(const M4 &A, const M4 &B) {
    M4 R;
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            float acc = 0.0f;
            for (int k = 0; k < 4; ++k) acc += A.m[r][k] * B.m[k][c];
            R.m[r][c] = acc;
        }
    return R;
}

// matmul(A, B) for two rigid motions - last rows (0, 0, 0, 1), which the caller has checked - without the 65 operations whose
// operands are those zeros and ones: for finite entries the sums below are the values of the general loop (0 + a b = a b,
// x + a 0 = x, 0 b + ... + 1 b3 = b3; only the sign of a zero result can differ).  Used for pose <- exp(x) * pose (commons.h:404).
KF_HD M4 matmul_rigid(const M4 &A, const M4 &B) {
    M4 R;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 4; ++c) {
            float acc = A.m[r][0] * B.m[0][c];
            acc += A.m[r][1] * B.m[1][c];
            acc += A.m[r][2] * B.m[2][c];
            if (c == 3) acc += A.m[r][3];
            R.m[r][c] = acc;
        }
    }
    R.m[3][0] = 0.0f; R.m[3][1] = 0.0f; R.m[3][2] = 0.0f; R.m[3][3] = 1.0f;
    return R;
}
KF_HD bool last_row_is_rigid(const M4 &A) { return A.m[3][0] == 0.0f && A.m[3][1] == 0.0f && A.m[3][2] == 0.0f && A.m[3][3] == 1.0f; }

// commons.h:365-371: Gauss elimination with partial pivoting against the identity (float storage,
// row multiplier held in double as TooN does), then back substitution.
KF_HD M4 inverse(const M4 &Ain) {
    float A[4][4], B[4][4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) { A[i][j] = Ain.m[i][j]; B[i][j] = (i == j) ? 1.0f : 0.0f; }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int piv = i;
        float best = fabsf(A[i][i]);
#pragma unroll
        for (int r = i + 1; r < 4; ++r) {
            const double v = fabsf(A[r][i]);
            if (v > best) { best = (float)v; piv = r; }
        }
        // row swap written branch-free over a fixed index range so the arrays stay in registers
#pragma unroll
        for (int r = i + 1; r < 4; ++r) {
            if (r == piv) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (j >= i) { const float t = A[i][j]; A[i][j] = A[r][j]; A[r][j] = t; }
                    const float u = B[i][j]; B[i][j] = B[r][j]; B[r][j] = u;
                }
            }
        }
        const float inv_pivot = 1.0f / A[i][i];
#pragma unroll
        for (int j = i + 1; j < 4; ++j) A[i][j] *= inv_pivot;
#pragma unroll
        for (int j = 0; j < 4; ++j) B[i][j] *= inv_pivot;
#pragma unroll
        for (int u = i + 1; u < 4; ++u) {
            const double factor = A[u][i];
#pragma unroll
            for (int j = i + 1; j < 4; ++j) A[u][j] = (float)(A[u][j] - factor * A[i][j]);
#pragma unroll
            for (int j = 0; j < 4; ++j) B[u][j] = (float)(B[u][j] - factor * B[i][j]);
        }
    }
    M4 X;
#pragma unroll
    for (int i = 3; i >= 0; --i)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v = B[i][c];
#pragma unroll
            for (int j = i + 1; j < 4; ++j) v -= A[i][j] * X.m[j][c];
            X.m[i][c] = v;
        }
    return X;
}

// TooN SE3<double>::exp + rodrigues_so3_exp, materialised as a float 4x4 (commons.h:406-412).
KF_HD M4 se3_exp_to_m4(const double mu[6]) {
    const double one_6th = 1.0 / 6.0, one_20th = 1.0 / 20.0;
    const double v0 = mu[0], v1 = mu[1], v2 = mu[2], w0 = mu[3], w1 = mu[4], w2 = mu[5];
    const double theta_sq = w0 * w0 + w1 * w1 + w2 * w2;
    const double theta = sqrt(theta_sq);
    const double c0 = w1 * v2 - w2 * v1, c1 = w2 * v0 - w0 * v2, c2 = w0 * v1 - w1 * v0;
    double A, B, t0, t1, t2;
    if (theta_sq < 1e-8) {
        A = 1.0 - one_6th * theta_sq;
        B = 0.5;
        t0 = v0 + 0.5 * c0; t1 = v1 + 0.5 * c1; t2 = v2 + 0.5 * c2;
    } else {
        double C;
        if (theta_sq < 1e-6) {
            C = one_6th * (1.0 - one_20th * theta_sq);
            A = 1.0 - theta_sq * C;
            B = 0.5 - 0.25 * one_6th * theta_sq;
        } else {
#if KFH_SOLVE_LEAN
            if (theta_sq < 1e-3) {
                // the same three series cut after s^3: the first neglected terms (s^4 / 9!, 2 s^4 / 10!, 6 s^4 / 11!) are below 3e-18 for
                // s < 1e-3, under half an ulp of the double result.  An ICP update turns the camera by milliradians (s ~ 1e-6 ... 1e-4),
                // so this is the branch a frame takes: nine fused multiply-adds on the one thread that every block waits for, instead
                // of ~75 operations of the nine-term chains below.
                const double s = theta_sq;
                A = __builtin_fma(s, __builtin_fma(s, __builtin_fma(s, -1.0 / 5040.0, 1.0 / 120.0), -1.0 / 6.0), 1.0);
                B = __builtin_fma(s, __builtin_fma(s, __builtin_fma(s, -1.0 / 40320.0, 1.0 / 720.0), -1.0 / 24.0), 0.5);
                C = __builtin_fma(s, __builtin_fma(s, __builtin_fma(s, -1.0 / 362880.0, 1.0 / 5040.0), -1.0 / 120.0), 1.0 / 6.0);
            } else
#endif
            if (theta_sq < 0.25) {
                // sin(t)/t, (1 - cos t)/t^2 and (1 - sin(t)/t)/t^2 from their alternating series in s = t^2 (nine terms:
                // the first neglected one is below 1e-22 for s < 0.25), i.e. to a couple of ulps of double - the values
                // are rounded to float below.  An ICP update is a rotation of a few milliradians; the library sin/cos
                // pair costs this single thread ~1500 dependent cycles per iteration, the three Horner chains ~300.
                const double s = theta_sq;
                A = 1.0 - s * (1.0 / 6.0) * (1.0 - s * (1.0 / 20.0) * (1.0 - s * (1.0 / 42.0) * (1.0 - s * (1.0 / 72.0) * (1.0 - s * (1.0 / 110.0) *
                    (1.0 - s * (1.0 / 156.0) * (1.0 - s * (1.0 / 210.0) * (1.0 - s * (1.0 / 272.0))))))));
                B = 0.5 * (1.0 - s * (1.0 / 12.0) * (1.0 - s * (1.0 / 30.0) * (1.0 - s * (1.0 / 56.0) * (1.0 - s * (1.0 / 90.0) * (1.0 - s * (1.0 / 132.0) *
                    (1.0 - s * (1.0 / 182.0) * (1.0 - s * (1.0 / 240.0) * (1.0 - s * (1.0 / 306.0)))))))));
                C = one_6th * (1.0 - s * (1.0 / 20.0) * (1.0 - s * (1.0 / 42.0) * (1.0 - s * (1.0 / 72.0) * (1.0 - s * (1.0 / 110.0) * (1.0 - s * (1.0 / 156.0) *
                    (1.0 - s * (1.0 / 210.0) * (1.0 - s * (1.0 / 272.0) * (1.0 - s * (1.0 / 342.0)))))))));
            } else {
                const double inv_theta = 1.0 / theta;
                A = sin(theta) * inv_theta;
                B = (1 - cos(theta)) * (inv_theta * inv_theta);
                C = (1 - A) * (inv_theta * inv_theta);
            }
        }
        const double d0 = w1 * c2 - w2 * c1, d1 = w2 * c0 - w0 * c2, d2 = w0 * c1 - w1 * c0;
        t0 = v0 + B * c0 + C * d0; t1 = v1 + B * c1 + C * d1; t2 = v2 + B * c2 + C * d2;
    }
    M4 out;
    {
        const double wx2 = w0 * w0, wy2 = w1 * w1, wz2 = w2 * w2;
        out.m[0][0] = (float)(1.0 - B * (wy2 + wz2));
        out.m[1][1] = (float)(1.0 - B * (wx2 + wz2));
        out.m[2][2] = (float)(1.0 - B * (wx2 + wy2));
    }
    { const double a = A * w2, b = B * (w0 * w1); out.m[0][1] = (float)(b - a); out.m[1][0] = (float)(b + a); }
    { const double a = A * w1, b = B * (w0 * w2); out.m[0][2] = (float)(b + a); out.m[2][0] = (float)(b - a); }
    { const double a = A * w0, b = B * (w1 * w2); out.m[1][2] = (float)(b - a); out.m[2][1] = (float)(b + a); }
    out.m[0][3] = (float)t0; out.m[1][3] = (float)t1; out.m[2][3] = (float)t2;
    out.m[3][0] = 0.0f; out.m[3][1] = 0.0f; out.m[3][2] = 0.0f; out.m[3][3] = 1.0f;
    return out;
}

// commons.h:380-404: x = pinv(C) b with singular values below max/1e6 dropped; C symmetric 6x6
// (21 upper-triangle floats, row-major) so the SVD is the eigendecomposition.  Cyclic Jacobi, double.
// vals = {b[6], C upper triangle[21]}.
// C and Q are caller-provided 6x6 work arrays: on the device they live in LDS, so that this rarely taken path
// neither inflates the register allocation of the ICP kernel nor makes it use scratch memory.
__host__ __device__ inline void solve6_ws(const float *vals, double x[6], double (*C)[6], double (*Q)[6]) {
    double b[6];
    for (int i = 0; i < 6; ++i) b[i] = vals[i];
    {
        int p = 6;
        for (int r = 0; r < 6; ++r)
            for (int c = r; c < 6; ++c) { C[r][c] = vals[p]; C[c][r] = vals[p]; ++p; }
    }
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
                const double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double cs = 1.0 / sqrt(tt * tt + 1.0), sn = tt * cs;
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
    for (int i = 0; i < 6; ++i) smax = fabs(C[i][i]) > smax ? fabs(C[i][i]) : smax;
    for (int i = 0; i < 6; ++i) x[i] = 0.0;
    for (int i = 0; i < 6; ++i) {
        const double l = C[i][i];
        if (!(fabs(l) * 1e6 > smax)) continue;
        double qb = 0.0;
        for (int k = 0; k < 6; ++k) qb += Q[k][i] * b[k];
        const double coef = qb / l;
        for (int k = 0; k < 6; ++k) x[k] += coef * Q[k][i];
    }
}

__host__ __device__ inline void solve6(const float *vals, double x[6]) {
    double C[6][6], Q[6][6];
    solve6_ws(vals, x, C, Q);
}

// Fast path of the same solve.  When C is comfortably positive definite the pseudo-inverse of
// commons.h:403 keeps every singular value, so x = C^-1 b, and an LDL^T factorisation (6 divisions, no
// square roots, no iteration) followed by two triangular substitutions gives it to the same ~1e-12.  The
// factorisation certifies its own use (see the bound below); otherwise (the all-zero system of frames 0-3,
// degenerate or badly scaled geometry) the caller falls back to solve6.
// 1/d for the pivots of the factorisation.  On the device: the hardware estimate refined by two Newton steps (5
// instructions, within a couple of ulps) instead of the ~35-instruction correctly rounded division - the six pivots are a
// serial chain run by one thread between two ICP launches, and the solution only has to be good to ~1e-12.
__host__ __device__ inline double pivot_reciprocal(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(d);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    return r;
#else
    return 1.0 / d;
#endif
}

__host__ __device__ inline bool solve6_ldlt(const float *vals, double x[6], unsigned long long *stamps = nullptr) {
    double L[6][6], d[6];
    double c_f2 = 0.0, tr = 0.0;
    {
        int p = 6;
#if KFH_SOLVE_LEAN
        double off2 = 0.0;      // |C|_F^2 = sum of the squared diagonal + twice the squared off-diagonal entries: 22 fused multiply-adds
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = r; c < 6; ++c) {
                const double v = vals[p]; ++p;
                L[c][r] = v;   // lower triangle holds C
                if (c == r) c_f2 = __builtin_fma(v, v, c_f2); else off2 = __builtin_fma(v, v, off2);
            }
        c_f2 = __builtin_fma(2.0, off2, c_f2);
#else
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = r; c < 6; ++c) {
                const double v = vals[p]; ++p;
                L[c][r] = v;   // lower triangle holds C
                c_f2 += (c == r) ? v * v : 2.0 * v * v;
            }
#endif
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) tr += L[i][i];
    if (!(tr > 0.0)) return false;
    const double tiny = tr * 1e-9;
    // in-place LDL^T: L[i][j] (i > j) becomes the unit-lower factor, d the diagonal
    double invd[6], dmin = 1e300;
#if KFH_SOLVE_LEAN
    // row j scaled by the pivots once (u[k] = L[j][k] d[k]) and every inner product as a chain of fused multiply-adds: 95 operations
    // on the serial path instead of 150 (the solution only has to be good to ~1e-12; its bits are not the reference's anyway)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double u[6];
        double dj = L[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) { u[k] = L[j][k] * d[k]; dj = __builtin_fma(-L[j][k], u[k], dj); }
        if (!(dj > tiny)) return false;
        d[j] = dj;
        dmin = dj < dmin ? dj : dmin;
        const double inv = pivot_reciprocal(dj);
        invd[j] = inv;
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
            double v = L[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) v = __builtin_fma(-L[i][k], u[k], v);
            L[i][j] = v * inv;
        }
    }
#else
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double dj = L[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) dj -= L[j][k] * L[j][k] * d[k];
        if (!(dj > tiny)) return false;
        d[j] = dj;
        dmin = dj < dmin ? dj : dmin;
        const double inv = pivot_reciprocal(dj);
        invd[j] = inv;
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
            double v = L[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) v -= L[i][k] * L[j][k] * d[k];
            L[i][j] = v * inv;
        }
    }
#endif
    KF_SOLVE_STAMP(stamps, 9, (float)(L[5][4] + d[5]));
#if KFH_SOLVE_LEAN
    // |L^-1|_F^2 for the certificate, column by column with nothing kept (column c of M = L^-1 solves L m = e_c and needs only its
    // own earlier entries), and x by substitution with L itself - L y = b, z = D^-1 y, L^T x = z - instead of x = M^T D^-1 M b with
    // all of M held: the same operation count, 15 doubles (30 registers) fewer at the peak.  The 1024-thread ICP blocks have 128
    // registers per thread; what this routine does not need stays free for whatever its caller keeps live around it (inside the
    // iteration loop of k_icp_level the old form spilled ~40 values into its dependent chain).
    double m_f2 = 6.0;
#pragma unroll
    for (int c = 0; c < 5; ++c) {
        double m[6];
#pragma unroll
        for (int r = c + 1; r < 6; ++r) {
            double v = -L[r][c];
#pragma unroll
            for (int k = c + 1; k < r; ++k) v = __builtin_fma(-L[r][k], m[k], v);
            m[r] = v;
            m_f2 = __builtin_fma(v, v, m_f2);
        }
    }
    // Certificate: C^-1 = M^T D^-1 M, so |C^-1|_2 <= |M|_F^2 / d_min, and |C|_2 <= |C|_F: when their product is below 5e5
    // no singular value can fall under the reference's max/1e6 cut-off and the plain inverse is what the reference
    // computes too.  The bound over-estimates the condition number ~4x on real ICP systems (cond ~1e3, bound ~4e3:
    // profiles/r01_sweeps.md), so everything up to cond ~1e5 stays on this path.
    KF_SOLVE_STAMP(stamps, 10, (float)m_f2);
    if (!(c_f2 * (m_f2 * m_f2) < (5e5 * 5e5) * (dmin * dmin))) return false;
    // (b is read where it is used: held from the top, its six doubles pushed the factorisation above the 128 registers and two
    // of its values into scratch memory - and a kernel that declares scratch costs every launch 0.2-0.5 us, profiles/r04_launch_cost.txt)
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double v = vals[i];
#pragma unroll
        for (int k = 0; k < i; ++k) v = __builtin_fma(-L[i][k], y[k], v);
        y[i] = v;
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double v = y[i] * invd[i];
#pragma unroll
        for (int k = i + 1; k < 6; ++k) v = __builtin_fma(-L[k][i], x[k], v);
        x[i] = v;
    }
#else
    // M = L^-1 (unit lower triangular), column by column
    double M[6][6], m_f2 = 6.0;
#pragma unroll
    for (int c = 0; c < 6; ++c)
#pragma unroll
        for (int r = c + 1; r < 6; ++r) {
            double v = -L[r][c];
#pragma unroll
            for (int k = c + 1; k < r; ++k) v -= L[r][k] * M[k][c];
            M[r][c] = v;
            m_f2 += v * v;
        }
    // Certificate: C^-1 = M^T D^-1 M, so |C^-1|_2 <= |M|_F^2 / d_min, and |C|_2 <= |C|_F: when their product is below 5e5
    // no singular value can fall under the reference's max/1e6 cut-off and the plain inverse is what the reference
    // computes too.  The bound over-estimates the condition number ~4x on real ICP systems (cond ~1e3, bound ~4e3:
    // profiles/r01_sweeps.md), so everything up to cond ~1e5 stays on this path.
    KF_SOLVE_STAMP(stamps, 10, (float)m_f2);
    if (!(c_f2 * (m_f2 * m_f2) < (5e5 * 5e5) * (dmin * dmin))) return false;
    // x = M^T D^-1 M b  (b is read where it is used: held from the top, its six doubles are what pushed the factorisation
    // above the 128 registers of a 1024-thread block and two of its values into scratch memory - and a kernel that
    // declares scratch costs every launch 0.2-0.5 us, profiles/r04_launch_cost.txt)
    double y[6], b[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) b[i] = vals[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double v = b[i];
#pragma unroll
        for (int k = 0; k < i; ++k) v += M[i][k] * b[k];
        y[i] = v * invd[i];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double v = y[i];
#pragma unroll
        for (int k = i + 1; k < 6; ++k) v += M[k][i] * y[k];
        x[i] = v;
    }
#endif
    return true;
}

}  // namespace kf
