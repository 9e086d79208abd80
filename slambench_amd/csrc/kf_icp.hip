// kf_icp.hip - trackKernel + reduceKernel + updatePose / checkPose (cpp:251-560, 759-792) fused into one launch per ICP iteration, the
// shared launch of a level's late iterations, the side blocks that carry the next frame's preprocessing, and the control kernel.
#include "kf_dev.h"

namespace kf {

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

__global__ void k_check_pose(M4 *pose, const M4 *oldPose, const float *reduction32, unsigned W, unsigned H, float thr,
                             int *ok) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const bool good = pose_is_good(reduction32, W, H, thr);
    *ok = good ? 1 : 0;
    if (!good) *pose = *oldPose;
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

namespace launch {

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

unsigned icp_blocks(unsigned w, unsigned h) {
    // default 256 blocks of 1024 threads (one per CU, ~1.2 pixels per thread at 640x480); KFH_ICP_BLOCKS overrides
    static const long cap_env = [] { const char *e = getenv("KFH_ICP_BLOCKS"); const long v = e ? atol(e) : 0; return (v >= 1 && v <= (long)TR_MAX_BLOCKS) ? v : 256L; }();
    const unsigned n = (w * h + ICP_THREADS - 1) / ICP_THREADS;
    return n < 1 ? 1 : (n > (unsigned)cap_env ? (unsigned)cap_env : n);
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
}  // namespace launch
}  // namespace kf
