// kf_kernels.h — device-resident state and launch wrappers of the HIP KinectFusion kernels.
#pragma once
#include "kf_math.h"
#include "../../include/kfusion_hip.h"

namespace kf {

struct Gauss5 { float g[5]; };

// Parameter blocks that live in HBM and are rewritten by the control kernel, so that a whole frame
// (19 ICP iterations included) is queued without the host ever reading the pose back.
struct TrackParams { M4 view; int have_reference; int pad[3]; };       // view = K * inverse(raycastPose)
struct IntegrateParams { M4 invTrack; M4 K; int enabled; float cam_bound, far_bound; int pad; };   // bounds: integrate_bounds()
struct RayParams { M4 view; int enabled; int pad[3]; };

struct DevState {
    M4 pose;          // camera-to-world, updated in place by k_solve
    M4 oldPose;       // pose at tracking start (cpp/kernels.cpp:947)
    M4 raycastPose;   // pose of the last raycast (cpp/kernels.cpp:978)
    TrackParams track;
    IntegrateParams integ;
    RayParams ray;
    RayParams render;
    float reduction[32];   // summed row 0 of the reference's 8x32 reduction output
    int have_reference;    // 0 until the first raycast (frames 0-3)
    int tracked;
    int integrated;
    int level_done[KFH_MAX_LEVELS];
    unsigned long long icp_iterations;
    unsigned long long integrate_updated;   // U of SURVEY §8(d), summed over all integrate launches
    unsigned long long solve_fallbacks;     // ICP solves that needed the Jacobi path (ill-conditioned J^T J)
    unsigned long long icp_level_iterations[KFH_MAX_LEVELS];   // icp_iterations by pyramid level
    unsigned tail_barrier[KFH_MAX_LEVELS];   // arrival counters of k_icp_tail's grid barrier (zero between frames)
    unsigned barrier_timeouts;               // workgroups that gave up waiting there (never, unless the device is hogged)
    unsigned timeouts_at_track_begin;        // barrier_timeouts when the current frame's tracking began
    unsigned failed_frames;                  // frames whose ICP loop saw a barrier time-out: pose rolled back, reported untracked
};

// Upper bounds over the whole volume of |pos.*| (pos = inverse(pose) * voxel centre: every voxel is within `far` of the
// camera, far = distance from the camera position to the farthest corner of [0, dim]^3, slightly inflated) and of
// |cam.x|, |cam.y| (cam = K * pos: |cam.x| <= fx |pos.x| + cx |pos.z| <= (fx + cx) far).  k_integrate derives from them how far
// a marched state (n rounded additions) can be from the exact line, i.e. which of its shortcuts are safe for this launch.
__host__ __device__ inline void integrate_bounds(const M4 &pose, const float k[4], f3 dim, float *cam_bound, float *far_bound) {
    const float t[3] = {pose.m[0][3], pose.m[1][3], pose.m[2][3]}, d[3] = {dim.x, dim.y, dim.z};
    float far2 = 0.f;
    for (int i = 0; i < 3; ++i) {
        const float a = fabsf(t[i]), b = fabsf(d[i] - t[i]), m = a > b ? a : b;
        far2 += m * m;
    }
    const float far = sqrtf(far2) * 1.001f + 1e-3f;
    const float fx = fabsf(k[0]) + fabsf(k[2]), fy = fabsf(k[1]) + fabsf(k[3]);
    *far_bound = far;
    *cam_bound = (fx > fy ? fx : fy) * far;
}

// The same for an arbitrary 4x4 K (the kernel-level entry point kfh_k_integrate takes a Matrix4 like integrateKernel, cpp:628):
// |cam.i| <= (|Ki0| + |Ki1| + |Ki2|) far + |Ki3|; cam_bound covers rows 0 and 1, far_bound both |pos.z| and |cam.z|.
__host__ __device__ inline void integrate_bounds_matrix(const M4 &pose, const M4 &K, f3 dim, float *cam_bound, float *far_bound) {
    const float unit[4] = {1.f, 1.f, 0.f, 0.f};
    float cb, far;
    integrate_bounds(pose, unit, dim, &cb, &far);
    float row[3];
    for (int i = 0; i < 3; ++i) row[i] = (fabsf(K.m[i][0]) + fabsf(K.m[i][1]) + fabsf(K.m[i][2])) * far + fabsf(K.m[i][3]);
    *cam_bound = row[0] > row[1] ? row[0] : row[1];
    *far_bound = row[2] > far ? row[2] : far;
}

// state entering ICP iteration `it` (pose to associate with, per-level converged flags) and whether it ran
struct IcpState { M4 pose; int done[KFH_MAX_LEVELS]; int executed; int pad[3]; };

struct FrameLogEntry { M4 pose; int tracked; int integrated; int frame; int pad; };

// bits of k_control's op mask, executed in this order by one thread
enum ControlOps {
    OP_TRACK_BEGIN = 1,   // oldPose = pose; view = K * inverse(raycastPose); clear level flags
    OP_TRACK_END = 2,     // checkPoseKernel -> tracked
    OP_INTEG_BEGIN = 4,   // checkPoseKernel again, gate, invTrack = inverse(pose)
    OP_RAY_BEGIN = 8,     // frame > 2: raycastPose = pose; view = raycastPose * invK
    OP_FRAME_END = 16,    // append {pose, tracked, integrated} to the device frame log
    OP_RENDER_BEGIN = 32  // render.view = viewPose * invK
};

struct ControlArgs {
    float k[4];
    unsigned frame, tracking_rate, integration_rate;
    unsigned width, height;
    float track_threshold;
    int use_view_pose;   // OP_RENDER_BEGIN: 1 -> view_pose below, 0 -> current pose
    M4 view_pose;
    unsigned log_capacity;
    float vol_dim[3];    // OP_INTEG_BEGIN: extent of the volume (integrate_bounds)
};

// Input-only work of the NEXT frame that rides on an ICP launch of the current one (kf_icp.hip, "side jobs"):
// a range of 32x8 bilateral tiles or of 32x32 pyramid tiles.  All buffers belong to the other half of the double buffer.
struct SideWork {
    enum Kind { NONE = 0, BILATERAL = 1, PYRAMID = 2, RAY_ORDER = 3 };
    int kind;
    unsigned first, count;            // tile range
    unsigned W, H;
    // BILATERAL (mm2meters + bilateral filter, cpp:562-589 + 159-198)
    float *floatDepth, *filtered;
    const uint16_t *in;
    unsigned inW;
    int ratio;
    Gauss5 gauss;
    float e_d;                        // e_delta for BILATERAL, e_delta * 3 for PYRAMID
    // PYRAMID (cpp:931-945)
    int n_levels;
    int pyr_part;                     // 0: every level of the tiles; 1: level 0 only; 2: the levels above it
    float *scaled[4];
    float4 *vertex[4], *normal[4];
    const float4 *stale_normal[4];
    M4 invK[4];
    // RAY_ORDER (dispatch order of the next raycast, one block)
    unsigned *ray_order, *ray_cost;
    unsigned ray_tiles_x, ray_tiles_y;
    unsigned order_mode;              // 0: raycast tiles, 1: integrate blocks (ray_tiles_x = count, ray_tiles_y = 1)
};

// OP_TRACK_BEGIN of the next queued frame, carried by a raycast launch (kf_raycast.hip, SkipGrid::next_state)
struct NextTrackBegin { DevState *state; float k[4]; unsigned frame, tracking_rate; };

namespace launch {
// first kernel launch the runtime rejected since the last call ("launch of k_x failed: ..."), or nullptr
const char *take_launch_error();
unsigned track_blocks(unsigned w, unsigned h);
unsigned bilateral_tiles(unsigned W, unsigned H);
unsigned pyramid_tiles(unsigned W, unsigned H);
unsigned pyramid_side_rounds(unsigned tiles, unsigned icp_nb);
bool pyramid_fusable(int n_levels, unsigned W, unsigned H);
unsigned track_blocks_max();
unsigned icp_tail_unit();      // workgroups of one co-residency unit of k_icp_tail (32)
unsigned icp_blocks(unsigned w, unsigned h);
void init_volume(int16_t *vol, unsigned sx, unsigned sy, unsigned sz, hipStream_t s);
void preprocess(float *floatDepth, float *filtered, unsigned W, unsigned H, const uint16_t *in, unsigned inW, int ratio,
                const Gauss5 &g, float e_d, hipStream_t s);
void mm2meters(float *out, unsigned W, unsigned H, const uint16_t *in, unsigned inW, int ratio, hipStream_t s);
void bilateral(float *out, const float *in, unsigned W, unsigned H, const Gauss5 &g, float e_d, hipStream_t s);
void half_sample(float *out, const float *in, unsigned inW, unsigned inH, float e_d, hipStream_t s);
void depth2vertex(float4 *vertex, const float *depth, unsigned W, unsigned H, const M4 &invK, hipStream_t s);
void vertex2normal(float4 *normal, const float4 *vertex, unsigned W, unsigned H, hipStream_t s);
void vertex_normal(float4 *vertex, float4 *normal, const float *depth, unsigned W, unsigned H, const M4 &invK, hipStream_t s);
// whole input pyramid (half-sampling + vertex/normal maps of every level) in one launch; false if the shape is not
// covered (more than 4 levels, sizes not divisible by 2^(L-1)) and the per-level kernels must be used
bool pyramid(int n_levels, float *const *scaled, float4 *const *vertex, float4 *const *normal, const float4 *const *stale_normal,
             const M4 *invK, unsigned W, unsigned H, float e_d, hipStream_t s);
// fused ICP pass: per-pixel association + block partial sums.  pose/params/skip are device pointers.
void track_reduce(float *partials, int *result_map, kfh_track_data *track_out, const float4 *inVertex,
                  const float4 *inNormal, unsigned inW, unsigned inH, const float4 *refVertex, const float4 *refNormal,
                  unsigned refW, unsigned refH, const M4 *pose, const TrackParams *tp, const int *skip,
                  float dist_threshold, float normal_threshold, hipStream_t s);
void reduce_trackdata(float *partials, const kfh_track_data *J, unsigned JW, unsigned W, unsigned H, hipStream_t s);
// sums the block partials in a fixed order into st->reduction (or out32) and, if do_solve, updates st->pose
void solve(DevState *st, float *out32, const float *partials, unsigned nblocks, int level, float icp_threshold,
           int do_solve, hipStream_t s);
void control(DevState *st, FrameLogEntry *log, int ops, const ControlArgs &a, hipStream_t s);
// fused ICP iteration: solve of the previous iteration in the prologue, association + partial sums in the body
void icp_iter(DevState *st, IcpState *iters, const float *partials_prev, float *partials_out, int *result_map, int it,
              int level, int prev_level, unsigned prev_nblocks, unsigned nblocks, const float4 *inV, const float4 *inN,
              unsigned inW, unsigned inH, const float4 *refV, const float4 *refN, unsigned refW, unsigned refH,
              float icp_threshold, float dist_threshold, float dist2_max, float normal_threshold, const SideWork *side, hipStream_t s);
// the last n_its iterations of a level (it .. it + n_its - 1, all after an ordinary launch of that level) in one launch;
// if it is the last launch of the frame's loop it can apply the final update and the control operations too
void icp_tail(DevState *st, IcpState *iters, float *part0, float *part1, int *result_map, int it, int n_its, int level,
              unsigned nblocks, const float4 *inV, const float4 *inN, unsigned inW, unsigned inH, const float4 *refV,
              const float4 *refN, unsigned refW, unsigned refH, float icp_threshold, float dist_threshold, float dist2_max, float normal_threshold,
              FrameLogEntry *log, int final_ops, const ControlArgs *ca, bool rows_ahead, unsigned max_wgs, hipStream_t s);   // ca != nullptr: also does icp_final's work; max_wgs: workgroups the barrier may span
void icp_final(DevState *st, FrameLogEntry *log, const IcpState *iters, const float *partials_prev, int n_iters,
               int prev_level, unsigned prev_nblocks, float icp_threshold, int ops, const ControlArgs &a, hipStream_t s);
void update_pose(M4 *pose, const float *reduction32, float icp_threshold, int *converged, hipStream_t s);
void check_pose(M4 *pose, const M4 *oldPose, const float *reduction32, unsigned W, unsigned H, float thr, int *ok, hipStream_t s);
void integrate(int16_t *vol, unsigned sx, unsigned sy, unsigned sz, f3 dim, const float *depth, unsigned W, unsigned H,
               const IntegrateParams *p, float mu, float maxweight, unsigned long long *updated_total, unsigned char *skip_flags,
               unsigned *block_order, unsigned *block_cost, bool k_trivial, hipStream_t s);   // k_trivial: row 2 of p->K is (0, 0, 1, 0)
// integrate dispatch order (most expensive blocks of the previous launch first): integrate_blocks() entries (order) and
// twice that (cost + sort keys); init with init_ray_order, sort with integrate_order or as an ICP side block (order_mode 1)
unsigned integrate_blocks(unsigned sx, unsigned sy, unsigned sz);
bool integrate_reorder_enabled();
void integrate_order(unsigned *block_order, unsigned *block_cost, unsigned n, hipStream_t s);
size_t skip_grid_bytes(unsigned sx, unsigned sy, unsigned sz);
// verifies once per device and divisor (a few ms, synchronises the stream) the reciprocal form of pos*size/dim of the ray march
void prepare_volume_division(f3 dim, unsigned sx, unsigned sy, unsigned sz, hipStream_t s);
void prepare_bilateral_division(float e_d, hipStream_t s);
// verifies once per device and threshold (~1 ms, synchronises the stream) that length > t may be decided on the squared length (sqrt_gt_bound)
float prepare_sqrt_bound(float t, hipStream_t s);
// one-off self-check of the ICP block reduction's inline-assembly lane exchanges against the same tree written with plain shuffles:
// 1 verified, 0 mismatch (remembered per device), -1 the check itself could not run (a HIP error; not remembered)
int reduction_verified(hipStream_t s);
void build_skip_grid(unsigned char *flags, const int16_t *vol, unsigned sx, unsigned sy, unsigned sz, hipStream_t s);
void raycast(float4 *vertex, float4 *normal, unsigned W, unsigned H, const int16_t *vol, unsigned sx, unsigned sy,
             unsigned sz, f3 dim, const RayParams *p, float nearPlane, float farPlane, float step, float largestep,
             const unsigned char *skip_flags, unsigned *tile_order, unsigned *tile_cost, const NextTrackBegin *next, hipStream_t s);
// raycast dispatch order (most expensive screen tiles of the previous launch first): ray_tiles() entries each
unsigned ray_tiles(unsigned W, unsigned H);
void init_ray_order(unsigned *tile_order, unsigned *tile_cost, unsigned n, hipStream_t s);
void ray_order(unsigned *tile_order, unsigned *tile_cost, unsigned W, unsigned H, hipStream_t s);   // sort as a launch of its own
void ray_tile_grid(unsigned W, unsigned H, unsigned *tiles_x, unsigned *tiles_y);
bool ray_reorder_enabled();
struct CopyOutSegments { uint4 *dst[3]; const uint4 *src[3]; size_t n16[3]; };      // dst: pinned host memory as the device sees it; n16: 16-byte pieces (0 = unused)
void copy_out(const CopyOutSegments &g, unsigned blocks, hipStream_t s);
void render_depth(uchar4 *out, const float *depth, unsigned W, unsigned H, float nearPlane, float farPlane, hipStream_t s);
void render_track(uchar4 *out, const int *result_map, unsigned W, unsigned H, hipStream_t s);
void render_depth_track(uchar4 *out_depth, const float *depth, uchar4 *out_track, const int *result_map, unsigned W, unsigned H,
                        float nearPlane, float farPlane, hipStream_t s);      // both in one launch
void render_volume(uchar4 *out, unsigned W, unsigned H, const int16_t *vol, unsigned sx, unsigned sy, unsigned sz, f3 dim,
                   const RayParams *p, float nearPlane, float farPlane, float step, float largestep, f3 light, f3 ambient,
                   const unsigned char *skip_flags, const float4 *seen_vertex, const float4 *seen_normal, const RayParams *seen_rp, float seen_far_plane,
                   hipStream_t s);
// (seen_*: the vertex / normal maps and the parameters of a raycast of the SAME volume with the same near plane and step sizes, or
// nullptr: rays that hit there are shaded from the maps where the view matrices agree bit for bit, kf_raycast.hip k_render_volume)
}  // namespace launch
}  // namespace kf
