/*
 * kfusion_hip.h — C ABI of the MI355X-native KinectFusion kernel backend (libkfusion_hip.so).
 *
 * This is the drop-in boundary for the per-frame hot path of SLAMBench's KinectFusion.  The
 * reference selects a backend at link time behind a C++ class (kfusion/include/kernels.h:83-197);
 * every entry point below replaces one member function or free function of that header and keeps
 * its argument meaning and return value.  include/kfusion/kernels.h wraps this ABI back into the
 * reference's `class Kfusion`, so benchmark.cpp-style callers compile unchanged.
 *
 * Conventions
 *   - plain C types only; matrices are 16 floats, row-major (reference Matrix4, commons.h:317-319);
 *     camera k = {fx, fy, cx, cy} (reference float4 k); uchar4 images are 4 bytes per pixel.
 *   - "host" pointers are ordinary process memory owned by the caller (as in benchmark.cpp:103-110);
 *     "device" pointers are HIP device memory of the context's GPU.
 *   - all work is queued on the context's HIP stream; results handed to the host are complete after
 *     kfh_synchronise() (reference synchroniseDevices(), kernels.h:197).  Functions that return a
 *     tracking / integration verdict synchronise internally, like the reference's bool returns.
 *   - errors: functions returning int use <0 for failure (KFH_ERR_*); fatal configuration errors that
 *     make the reference print and exit(1) (cpp/kernels.cpp:566-576) return KFH_ERR_INVALID_RATIO.
 */
#ifndef KFUSION_HIP_H_
#define KFUSION_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KFH_OK 0
#define KFH_ERR_INVALID_RATIO (-1)
#define KFH_ERR_HIP (-2)
#define KFH_ERR_ARGUMENT (-3)
#define KFH_ERR_IO (-4)
#define KFH_ERR_BARRIER (-5)   /* the ICP grid barrier timed out (device oversubscribed): the frame's pose was rolled back */
#define KFH_MAX_LEVELS 8

typedef struct kfh_context kfh_context;

/* 32-byte per-pixel ICP record, identical to the reference TrackData (commons.h:325-329). */
typedef struct kfh_track_data {
    int32_t result;
    float error;
    float J[6];
} kfh_track_data;

/* ------------------------------------------------------------------ library
 * init()/clean() of kernels.h:77-79 (stubs in the CPP backend, cpp/kernels.cpp:138-145). */
int kfh_init(void);
void kfh_clean(void);
int kfh_device_count(void);
const char *kfh_last_error(void);
const char *kfh_version(void);

/* ------------------------------------------------------------------ lifecycle
 * Kfusion::Kfusion(uint2 inputSize, uint3 volumeResolution, float3 volumeDimensions,
 *                  float3 initPose, std::vector<int>& pyramid)           kernels.h:99-119
 * + Kfusion::languageSpecificConstructor()                               cpp/kernels.cpp:67-112
 * pyramid[l] = ICP iterations at level l, level 0 first (default_parameters.h:391-401).
 * device < 0 keeps the current HIP device.  stream == NULL creates a private stream. */
kfh_context *kfh_create(uint32_t width, uint32_t height, const uint32_t volume_resolution[3],
                        const float volume_dimensions[3], const float init_pose[3], const int32_t *pyramid,
                        int32_t n_levels, int32_t device, void *stream);
/* Second constructor (kernels.h:121-139): full 4x4 initial pose. */
kfh_context *kfh_create_with_pose(uint32_t width, uint32_t height, const uint32_t volume_resolution[3],
                                  const float volume_dimensions[3], const float init_pose16[16],
                                  const int32_t *pyramid, int32_t n_levels, int32_t device, void *stream);
/* Kfusion::~Kfusion()  cpp/kernels.cpp:114-134 */
void kfh_destroy(kfh_context *ctx);
/* Kfusion::reset()  cpp/kernels.cpp:135-137 (initVolumeKernel) */
void kfh_reset(kfh_context *ctx);
/* synchroniseDevices()  kernels.h:197 */
void kfh_synchronise(kfh_context *ctx);
/* the same for every live context of the process: the reference's synchroniseDevices() takes no argument */
void kfh_synchronise_all(void);

/* ------------------------------------------------------------------ per-frame pipeline
 * bool Kfusion::preprocessing(const ushort* inputDepth, const uint2 inputSize)   cpp/kernels.cpp:915-922
 * Returns 1, or KFH_ERR_INVALID_RATIO where the reference exits. */
int kfh_preprocessing(kfh_context *ctx, const uint16_t *host_depth_mm, uint32_t in_width, uint32_t in_height);
/* Same, input already resident in HBM (used by bench.py; the timed region excludes PCIe). */
int kfh_preprocessing_device(kfh_context *ctx, const uint16_t *device_depth_mm, uint32_t in_width, uint32_t in_height);
/* bool Kfusion::tracking(float4 k, float icp_threshold, uint tracking_rate, uint frame)  cpp/kernels.cpp:924-971
 * Returns 1 tracked / 0 not tracked, or a sticky KFH_ERR_* (< 0) once the context has failed: a kernel launch the runtime
 * rejected (KFH_ERR_HIP, the kernel is named in kfh_last_error()) or a time-out of the ICP grid barrier (KFH_ERR_BARRIER:
 * the affected frame's pose was rolled back on the device and the frame counts as not tracked).  The same holds for
 * kfh_integration, kfh_raycasting, kfh_compute_frame and the two verdict getters; include/kfusion/kernels.h turns it into
 * the reference's behaviour for fatal errors (message on stderr, exit(1)). */
int kfh_tracking(kfh_context *ctx, const float k[4], float icp_threshold, uint32_t tracking_rate, uint32_t frame);
/* bool Kfusion::integration(float4 k, uint integration_rate, float mu, uint frame)   cpp/kernels.cpp:988-1004 */
int kfh_integration(kfh_context *ctx, const float k[4], uint32_t integration_rate, float mu, uint32_t frame);
/* bool Kfusion::raycasting(float4 k, float mu, uint frame)   cpp/kernels.cpp:973-986 (always returns 0) */
int kfh_raycasting(kfh_context *ctx, const float k[4], float mu, uint32_t frame);
/* void Kfusion::computeFrame(...)   cpp/kernels.cpp:1048-1055.  Queues the four phases without any
 * host round trip; tracked / integrated are available from kfh_get_tracked()/kfh_get_integrated()
 * after kfh_synchronise().  `depth_is_device` selects which kind of pointer `depth_mm` is. */
int kfh_compute_frame(kfh_context *ctx, const uint16_t *depth_mm, int32_t depth_is_device, uint32_t in_width,
                      uint32_t in_height, const float k[4], uint32_t integration_rate, uint32_t tracking_rate,
                      float icp_threshold, float mu, uint32_t frame);

/* void Kfusion::renderDepth/renderTrack/renderVolume(uchar4* out, uint2 outputSize, ...)  cpp/kernels.cpp:1032-1046
 * `out` is host memory (width*height*4 bytes) that stays valid until then; it is complete after kfh_synchronise().
 * After kfh_compute_frame() the renders show that frame: they are carried out when its deferred second part is (the next
 * kfh_compute_frame(), or any call that needs the frame's result), right behind its raycast. */
void kfh_render_depth(kfh_context *ctx, uint8_t *out_rgba, uint32_t width, uint32_t height);
void kfh_render_track(kfh_context *ctx, uint8_t *out_rgba, uint32_t width, uint32_t height);
void kfh_render_volume(kfh_context *ctx, uint8_t *out_rgba, uint32_t width, uint32_t height, int32_t frame,
                       int32_t raycast_rendering_rate, const float k[4], float largestep);
/* void Kfusion::dumpVolume(const char*)   cpp/kernels.cpp:1006-1030: tsdf shorts only, x fastest. */
int kfh_dump_volume(kfh_context *ctx, const char *filename);

/* Inline accessors of kernels.h:141-194. */
void kfh_get_pose(kfh_context *ctx, float out16[16]);
void kfh_set_pose(kfh_context *ctx, const float in16[16]);
void kfh_get_position(kfh_context *ctx, float out3[3]);
int kfh_get_tracked(kfh_context *ctx);
int kfh_get_integrated(kfh_context *ctx);
/* KFH_OK, or the sticky error of the context (see kfh_tracking); synchronises, so it also covers every queued frame. */
int kfh_get_error(kfh_context *ctx);
void kfh_set_view_pose(kfh_context *ctx, const float *view_pose16_or_null);
void kfh_get_model_dimensions(kfh_context *ctx, float out3[3]);
void kfh_get_model_resolution(kfh_context *ctx, uint32_t out3[3]);
void kfh_get_computation_resolution(kfh_context *ctx, uint32_t out2[2]);

/* Trajectory kept on the device by kfh_compute_frame (one 4x4 pose + flags per frame, no host sync):
 * copies poses of frames [0, n) (pose *after* each frame) and their tracked/integrated flags. */
int kfh_read_frame_log(kfh_context *ctx, uint32_t n, float *poses16_out, int32_t *tracked_out, int32_t *integrated_out);
/* ICP iterations actually executed since the context was created (device counter). */
uint64_t kfh_icp_iterations(kfh_context *ctx);
/* launches a tracked frame's ICP loop takes in this context (the late iterations of a level share one; how many do depends on the
 * co-residency the context reserved at creation - bench.py's roofline entry divides by it) */
uint32_t kfh_icp_launches_per_frame(kfh_context *ctx);
/* The same by pyramid level (index 0 = full resolution); out has KFH_MAX_LEVELS entries. */
void kfh_icp_level_iterations(kfh_context *ctx, uint64_t out[KFH_MAX_LEVELS]);
/* Voxels updated by all integrate launches so far (U of SURVEY.md §8d; device counter). */
uint64_t kfh_integrate_updated(kfh_context *ctx);
/* ICP solves whose normal equations were too ill-conditioned for the LDL^T fast path (device counter). */
uint64_t kfh_solve_fallbacks(kfh_context *ctx);
/* Workgroups that gave up waiting at the grid barrier of the shared late-iteration ICP launch (device counter; stays 0
 * unless other work hogs the device for seconds - a frame it happened in is not to be trusted). */
uint64_t kfh_barrier_timeouts(kfh_context *ctx);

/* Per-phase device time from HIP events recorded on the context's stream around each launch group
 * (the role of the reference's KERNEL_TIMINGS TICK/TOCK, cpp/kernels.cpp:31-35).  phase_mask has bit
 * (1 << KFH_PROF_*) set for each phase to time; 0 turns timing off.  Setting the mask clears the totals. */
#define KFH_PROF_PREPROCESS 0 /* mm2meters + bilateral (one kernel) */
#define KFH_PROF_PYRAMID 1    /* halfSample x(L-1) + vertex/normal xL */
#define KFH_PROF_ICP 2        /* all track+reduce / solve launches of the frame */
#define KFH_PROF_INTEGRATE 3  /* integrateKernel (one kernel) */
#define KFH_PROF_RAYCAST 4    /* raycastKernel (one kernel) */
#define KFH_PROF_RENDER_DEPTH 5
#define KFH_PROF_RENDER_TRACK 6
#define KFH_PROF_RENDER_VOLUME 7
#define KFH_PROF_COUNT 8
/* With KERNEL_TIMINGS in the environment (the reference's switch, cpp/kernels.cpp:69-70) every phase call prints
 * "<name> <nanoseconds> <size>" lines on stderr like the reference's TOCK (cpp/kernels.cpp:33-35); fused launches are
 * reported under bilateralFilterKernel (mm2meters + bilateral), pyramidKernel (half-sampling + vertex/normal maps) and
 * trackKernel (the whole ICP loop: track + reduce + pose update launches of the frame). */
void kfh_set_profiling(kfh_context *ctx, uint32_t phase_mask);
/* Queued frames (kfh_compute_frame) only: put the event pairs around every n-th frame instead of every frame.  An event
 * record drains the queue for ~5 us on this GPU (measured: 6 records per frame cost 23 us of a 235 us frame), so a
 * benchmark that wants per-kernel times AND an undisturbed frame rate samples them.  Default 1. */
void kfh_set_profiling_stride(kfh_context *ctx, uint32_t every_n_frames);
/* Queued frames only: a sampled frame times ONE of icp / integrate / raycast - the phase sampled least so far among
 * those the frame runs (integrate is skipped on frames with frame % integration_rate != 0) - so every phase collects
 * samples whatever the stride and the rates are, and a sampled frame pays for one event pair instead of three. */
void kfh_set_profiling_rotation(kfh_context *ctx, int32_t one_phase_per_sampled_frame);
int kfh_read_profile(kfh_context *ctx, double ms_out[KFH_PROF_COUNT], uint64_t launches_out[KFH_PROF_COUNT]);

/* Copy an internal device buffer to the host for parity checks.  name is one of
 * "floatDepth", "scaledDepth<l>", "inputVertex<l>", "inputNormal<l>", "vertex", "normal"
 * (vertex/normal maps are float4-strided, 16 B per pixel), "trackResult" (int32 per pixel),
 * "reduction" (32 floats), "volume" (short2 per voxel), "pose", "oldPose", "raycastPose".
 * Returns bytes written, or <0. */
int64_t kfh_read_buffer(kfh_context *ctx, const char *name, void *host_out, uint64_t capacity);
int64_t kfh_write_buffer(kfh_context *ctx, const char *name, const void *host_in, uint64_t nbytes);
void *kfh_device_pointer(kfh_context *ctx, const char *name);

/* ------------------------------------------------------------------ kernel level
 * One entry point per compute kernel prototype of kernels.h:18-69.  All pointers are DEVICE memory,
 * `stream` is a hipStream_t (NULL = default stream).  Vertex / normal maps are float4-strided (16 B per
 * pixel, .w unused) instead of the reference's packed float3: they never leave the backend
 * (SURVEY.md §7 "Hard parts").  These launch exactly the kernels the pipeline uses. */
void kfh_k_init_volume(int16_t *vol, const uint32_t size[3], void *stream);                     /* kernels.h:18 */
int kfh_k_mm2meters(float *out, uint32_t out_w, uint32_t out_h, const uint16_t *in, uint32_t in_w, uint32_t in_h,
                    void *stream);                                                                /* kernels.h:38-39 */
void kfh_k_bilateral_filter(float *out, const float *in, uint32_t w, uint32_t h, const float *host_gaussian5,
                            float e_d, int32_t r, void *stream);                                  /* kernels.h:20-21 */
/* fused form used by the pipeline: mm2meters + bilateral in one pass (writes both maps) */
int kfh_k_preprocess(float *float_depth, float *filtered, uint32_t w, uint32_t h, const uint16_t *in, uint32_t in_w,
                     uint32_t in_h, const float *host_gaussian5, float e_d, void *stream);
void kfh_k_half_sample(float *out, const float *in, uint32_t in_w, uint32_t in_h, float e_d, int32_t r,
                       void *stream);                                                             /* kernels.h:41-42 */
void kfh_k_depth2vertex(float *vertex4, const float *depth, uint32_t w, uint32_t h, const float invK16[16],
                        void *stream);                                                            /* kernels.h:23-24 */
void kfh_k_vertex2normal(float *normal4, const float *vertex4, uint32_t w, uint32_t h, void *stream); /* kernels.h:36 */
/* fused form used by the pipeline: depth -> vertex + normal */
void kfh_k_vertex_normal(float *vertex4, float *normal4, const float *depth, uint32_t w, uint32_t h,
                         const float invK16[16], void *stream);
void kfh_k_track(kfh_track_data *output, const float *in_vertex4, const float *in_normal4, uint32_t in_w,
                 uint32_t in_h, const float *ref_vertex4, const float *ref_normal4, uint32_t ref_w, uint32_t ref_h,
                 const float Ttrack16[16], const float view16[16], float dist_threshold, float normal_threshold,
                 int32_t have_reference, void *stream);                                           /* kernels.h:29-34 */
void kfh_k_reduce(float *out8x32, const kfh_track_data *J, uint32_t j_w, uint32_t j_h, uint32_t w, uint32_t h,
                  void *stream);                                                                  /* kernels.h:26-27 */
/* fused form used by the pipeline: track + reduce in one pass; out32 = summed row 0 */
void kfh_k_track_reduce(float *out32, int32_t *result_map, const float *in_vertex4, const float *in_normal4,
                        uint32_t in_w, uint32_t in_h, const float *ref_vertex4, const float *ref_normal4,
                        uint32_t ref_w, uint32_t ref_h, const float Ttrack16[16], const float view16[16],
                        float dist_threshold, float normal_threshold, int32_t have_reference, void *stream);
/* updatePoseKernel / checkPoseKernel (kernels.h:44-48) run on the device; pose16 / reduction are device memory. */
int kfh_k_update_pose(float *pose16, const float *reduction32, float icp_threshold, void *stream);
int kfh_k_check_pose(float *pose16, const float *old_pose16, const float *reduction32, uint32_t w, uint32_t h,
                     float track_threshold, void *stream);
void kfh_k_integrate(int16_t *vol, const uint32_t size[3], const float dim[3], const float *depth, uint32_t w,
                     uint32_t h, const float inv_track16[16], const float K16[16], float mu, float maxweight,
                     void *stream);                                                               /* kernels.h:50-52 */
void kfh_k_raycast(float *vertex4, float *normal4, uint32_t w, uint32_t h, const int16_t *vol,
                   const uint32_t size[3], const float dim[3], const float view16[16], float near_plane,
                   float far_plane, float step, float largestep, void *stream);                   /* kernels.h:54-56 */
void kfh_k_render_depth(uint8_t *out_rgba, const float *depth, uint32_t w, uint32_t h, float near_plane,
                        float far_plane, void *stream);                                           /* kernels.h:60-61 */
void kfh_k_render_track(uint8_t *out_rgba, const int32_t *result_map, uint32_t w, uint32_t h, void *stream); /* :65 */
void kfh_k_render_volume(uint8_t *out_rgba, uint32_t w, uint32_t h, const int16_t *vol, const uint32_t size[3],
                         const float dim[3], const float view16[16], float near_plane, float far_plane, float step,
                         float largestep, const float light[3], const float ambient[3], void *stream); /* :67-69 */

#ifdef __cplusplus
}
#endif
#endif /* KFUSION_HIP_H_ */
