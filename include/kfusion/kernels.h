// kfusion/kernels.h — `class Kfusion` for SLAMBench drivers, backed by the MI355X HIP backend.
//
// Same public interface as the reference's kfusion/include/kernels.h:83-197 (two constructors, reset,
// preprocessing / tracking / integration / raycasting / computeFrame, render*, dumpVolume, the inline
// getters, init / clean / synchroniseDevices), so a driver written against the reference header —
// kfusion/src/benchmark.cpp:47-190 is the model — compiles against this one unchanged and links
// libkfusion_hip.so instead of one of kfusion-{cpp,openmp,opencl,cuda}.  Every method forwards to the C ABI
// of kfusion_hip.h; nothing is computed on the host.
//
// Differences a maintainer should know about (INTEGRATION.md has the full list):
//   * the private layout is one opaque handle, not the reference's member list (kernels.h:85-94): the
//     reference compiles its inline getters into the caller, so this header replaces that one wholesale;
//   * Kfusion is not copyable (it owns device memory); one instance per GPU stream, any number per process.
#ifndef KFUSION_HIP_KERNELS_H_
#define KFUSION_HIP_KERNELS_H_

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../kfusion_hip.h"
#include "commons.h"

inline void init() { kfh_init(); }
inline void clean() { kfh_clean(); }

class Kfusion {
    kfh_context *ctx_;
    Matrix4 pose_;         // host copy of the camera pose, refreshed by every member that can change it (the reference's `pose`, kernels.h:87)
    Matrix4 *view_pose_;   // NULL = follow the camera (kernels.h:179-184)
    uint2 computation_size_;
    uint3 volume_resolution_;
    float3 volume_dimensions_;
    bool tracked_, integrated_;

    Kfusion(const Kfusion &);
    Kfusion &operator=(const Kfusion &);

    void fail_if_null(const char *what) const {
        if (!ctx_) {
            std::fprintf(stderr, "%s failed: %s\n", what, kfh_last_error());
            std::exit(1);   // the reference backends exit(1) on fatal set-up errors
        }
    }
    // a negative verdict is a device-side failure (rejected launch, ICP barrier time-out): fatal, like the reference's set-up errors
    static int checked(int r, const char *what) {
        if (r < 0 && r != KFH_ERR_INVALID_RATIO) {
            std::fprintf(stderr, "%s failed (%d): %s\n", what, r, kfh_last_error());
            std::exit(1);
        }
        return r;
    }
    static void k4(const float4 &k, float out[4]) { out[0] = k.x; out[1] = k.y; out[2] = k.z; out[3] = k.w; }
    // the pose lives on the device; tracking() / computeFrame() wait for their verdict anyway, so the copy costs no extra sync
    void refresh_pose_() { kfh_get_pose(ctx_, &pose_.data[0].x); }

public:
    // kernels.h:99-119
    Kfusion(uint2 inputSize, uint3 volumeResolution, float3 volumeDimensions, float3 initPose, std::vector<int> &pyramid)
        : ctx_(0), view_pose_(0), computation_size_(inputSize), volume_resolution_(volumeResolution),
          volume_dimensions_(volumeDimensions), tracked_(false), integrated_(false) {
        const uint32_t res[3] = {volumeResolution.x, volumeResolution.y, volumeResolution.z};
        const float dim[3] = {volumeDimensions.x, volumeDimensions.y, volumeDimensions.z};
        const float pose[3] = {initPose.x, initPose.y, initPose.z};
        std::vector<int32_t> pyr(pyramid.begin(), pyramid.end());
        ctx_ = kfh_create(inputSize.x, inputSize.y, res, dim, pose, pyr.data(), (int32_t)pyr.size(), -1, 0);
        fail_if_null("Kfusion::Kfusion");
        refresh_pose_();
    }
    // kernels.h:121-139
    Kfusion(uint2 inputSize, uint3 volumeResolution, float3 volumeDimensions, Matrix4 initPose, std::vector<int> &pyramid)
        : ctx_(0), view_pose_(0), computation_size_(inputSize), volume_resolution_(volumeResolution),
          volume_dimensions_(volumeDimensions), tracked_(false), integrated_(false) {
        const uint32_t res[3] = {volumeResolution.x, volumeResolution.y, volumeResolution.z};
        const float dim[3] = {volumeDimensions.x, volumeDimensions.y, volumeDimensions.z};
        std::vector<int32_t> pyr(pyramid.begin(), pyramid.end());
        ctx_ = kfh_create_with_pose(inputSize.x, inputSize.y, res, dim, &initPose.data[0].x, pyr.data(), (int32_t)pyr.size(), -1, 0);
        fail_if_null("Kfusion::Kfusion");
        refresh_pose_();
    }
    ~Kfusion() { kfh_destroy(ctx_); }

    void reset() { kfh_reset(ctx_); }
    bool getTracked() { return tracked_; }
    bool getIntegrated() { return integrated_; }
    float3 getPosition() {
        float p[3];
        kfh_get_position(ctx_, p);
        return make_float3(p[0], p[1], p[2]);
    }

    void computeFrame(const ushort *inputDepth, const uint2 inputSize, float4 k, uint integration_rate, uint tracking_rate,
                      float icp_threshold, float mu, const uint frame) {
        float kk[4];
        k4(k, kk);
        const int r = kfh_compute_frame(ctx_, inputDepth, 0, inputSize.x, inputSize.y, kk, integration_rate, tracking_rate,
                                        icp_threshold, mu, frame);
        if (r == KFH_ERR_INVALID_RATIO) std::exit(1);   // cpp/kernels.cpp:566-576
        checked(r, "Kfusion::computeFrame");
        tracked_ = checked(kfh_get_tracked(ctx_), "Kfusion::computeFrame") != 0;
        integrated_ = checked(kfh_get_integrated(ctx_), "Kfusion::computeFrame") != 0;
        refresh_pose_();
    }

    bool preprocessing(const ushort *inputDepth, const uint2 inputSize) {
        const int r = kfh_preprocessing(ctx_, inputDepth, inputSize.x, inputSize.y);
        if (r == KFH_ERR_INVALID_RATIO) std::exit(1);
        return checked(r, "Kfusion::preprocessing") > 0;
    }
    bool tracking(float4 k, float icp_threshold, uint tracking_rate, uint frame) {
        float kk[4];
        k4(k, kk);
        tracked_ = checked(kfh_tracking(ctx_, kk, icp_threshold, tracking_rate, frame), "Kfusion::tracking") > 0;
        refresh_pose_();   // (a rejected pose is rolled back by this call already - cpp:991 does it in integration() - so pose_ is final here)
        return tracked_;
    }
    bool raycasting(float4 k, float mu, uint frame) {
        float kk[4];
        k4(k, kk);
        return checked(kfh_raycasting(ctx_, kk, mu, frame), "Kfusion::raycasting") > 0;
    }
    bool integration(float4 k, uint integration_rate, float mu, uint frame) {
        float kk[4];
        k4(k, kk);
        integrated_ = checked(kfh_integration(ctx_, kk, integration_rate, mu, frame), "Kfusion::integration") > 0;
        return integrated_;
    }

    void dumpVolume(const char *filename) {
        if (kfh_dump_volume(ctx_, filename) == KFH_ERR_IO) std::exit(1);   // cpp/kernels.cpp:1017-1020
    }
    // The three renders are QUEUED: `out` holds the image after the next synchroniseDevices() (every tock() of benchmark.cpp
    // calls it; the reference's OpenCL backend reads its render targets back the same way, opencl/kernels.cpp:420).  The
    // buffer must stay valid until then; a Kfusion that is destroyed first drops the copy instead of writing to it.
    void renderVolume(uchar4 *out, const uint2 outputSize, int frame, int rate, float4 k, float mu) {
        float kk[4];
        k4(k, kk);
        kfh_set_view_pose(ctx_, view_pose_ ? &view_pose_->data[0].x : 0);
        kfh_render_volume(ctx_, &out->x, outputSize.x, outputSize.y, frame, rate, kk, mu);
    }
    void renderTrack(uchar4 *out, const uint2 outputSize) { kfh_render_track(ctx_, &out->x, outputSize.x, outputSize.y); }
    void renderDepth(uchar4 *out, uint2 outputSize) { kfh_render_depth(ctx_, &out->x, outputSize.x, outputSize.y); }

    Matrix4 getPose() {
        refresh_pose_();
        return pose_;
    }
    void setViewPose(Matrix4 *value = NULL) { view_pose_ = value; }
    // kernels.h:176-184: a pointer to the caller's matrix, or to the LIVE camera pose - a caller that keeps the pointer (the GUI
    // mains do) sees the pose move: pose_ is rewritten in place by tracking() / computeFrame() / getPose()
    Matrix4 *getViewPose() { return view_pose_ ? view_pose_ : &pose_; }
    float3 getModelDimensions() { return volume_dimensions_; }
    uint3 getModelResolution() { return volume_resolution_; }
    uint2 getComputationResolution() { return computation_size_; }

    // not in the reference: the C-ABI handle behind this instance
    kfh_context *handle() { return ctx_; }
};

// kernels.h:197 — the reference's free function has no argument because its backends keep global state: here it waits
// for every live Kfusion of the process (all queued work done, render targets filled)
inline void synchroniseDevices() { kfh_synchronise_all(); }

#endif
