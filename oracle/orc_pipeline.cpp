// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header).  PARITY UNPINNED.
//
// Restates class Kfusion (kfusion/include/kernels.h:83-195) and its CPP pipeline methods
// (kfusion/src/cpp/kernels.cpp:67-137 constructor/reset, 915-1055 preprocessing / tracking /
// integration / raycasting / render* / dumpVolume), with per-instance state instead of the
// reference's file-scope globals (cpp:39-57).
#include "orc_math.h"
#include "orc_api.h"
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

using namespace orc;

namespace {
// kfusion/include/constant_parameters.h:15-26
const float e_delta = 0.1f;
const int radius = 2;
const float dist_threshold = 0.1f;
const float normal_threshold = 0.8f;
const float track_threshold = 0.15f;
const float maxweight = 100.0f;
const float nearPlane = 0.4f;
const float farPlane = 4.0f;
const float delta = 4.0f;
const float light[3] = {1, 1, -1.0};
const float ambient[3] = {(float)0.1, (float)0.1, (float)0.1};
}  // namespace

// for tests/test_oracle_golden.py: e_delta, dist_threshold, normal_threshold, track_threshold, maxweight, nearPlane,
// farPlane, delta, light xyz, ambient (x)
extern "C" const float orc_constants_table[12] = {e_delta, dist_threshold, normal_threshold, track_threshold, maxweight, nearPlane,
                                                   farPlane, delta, light[0], light[1], light[2], ambient[0]};

struct orc_kfusion {
    unsigned W, H;
    unsigned volRes[3];
    float volDim[3];
    float initPose[3];
    std::vector<int> iterations;
    float step;
    M4 pose, oldPose, raycastPose;
    bool have_reference;
    // buffers (cpp:39-55); calloc-zeroed like the reference so stale-field semantics carry over
    std::vector<float> gaussian, floatDepth, vertex, normal, reduction;
    std::vector<std::vector<float> > scaledDepth, inputVertex, inputNormal;
    std::vector<TrackRec> trackingResult;
    std::vector<short> volume;
    orc_counters counters;
    bool count_raycast;   // V instrumentation costs a pass over the volume: off unless asked for
};

static inline unsigned level_div(unsigned v, int l) { return v / (unsigned)(int)pow(2, l); }

extern "C" {

orc_kfusion *orc_kfusion_create(unsigned W, unsigned H, const unsigned volRes[3], const float volDim[3],
                                const float initPose3[3], const int *pyramid, int nLevels) {
    orc_kfusion *k = new orc_kfusion();
    k->W = W; k->H = H;
    for (int i = 0; i < 3; ++i) { k->volRes[i] = volRes[i]; k->volDim[i] = volDim[i]; k->initPose[i] = initPose3[i]; }
    k->iterations.assign(pyramid, pyramid + nLevels);
    // kernels.h:106-109 — SE3<float>(x,y,z,0,0,0) is the identity rotation with that translation.
    std::memset(&k->pose, 0, sizeof(M4));
    for (int i = 0; i < 4; ++i) k->pose.m[i][i] = 1.0f;
    for (int i = 0; i < 3; ++i) k->pose.m[i][3] = initPose3[i];
    std::memset(&k->oldPose, 0, sizeof(M4));
    std::memset(&k->raycastPose, 0, sizeof(M4));  // zero-initialised global in the reference (cpp:53)
    k->have_reference = false;
    // kernels.h:117: step = min(volumeDimensions) / max(volumeResolution)
    const float mind = fmin_h(volDim[0], fmin_h(volDim[1], volDim[2]));
    unsigned maxr = volRes[0] > volRes[1] ? volRes[0] : volRes[1];
    maxr = maxr > volRes[2] ? maxr : volRes[2];
    k->step = mind / maxr;
    // cpp:73-98
    k->reduction.assign(8 * 32, 0.f);
    k->scaledDepth.resize(nLevels); k->inputVertex.resize(nLevels); k->inputNormal.resize(nLevels);
    for (int i = 0; i < nLevels; ++i) {
        const size_t n = (size_t)(W * H) / (size_t)(int)pow(2, i);
        k->scaledDepth[i].assign(n, 0.f);
        k->inputVertex[i].assign(3 * n, 0.f);
        k->inputNormal[i].assign(3 * n, 0.f);
    }
    k->floatDepth.assign((size_t)W * H, 0.f);
    k->vertex.assign((size_t)3 * W * H, 0.f);
    k->normal.assign((size_t)3 * W * H, 0.f);
    TrackRec zero; std::memset(&zero, 0, sizeof zero);
    k->trackingResult.assign((size_t)W * H, zero);
    k->gaussian.assign(2 * radius + 1, 0.f);
    orc_gaussian(k->gaussian.data(), radius, delta);
    k->volume.assign((size_t)2 * volRes[0] * volRes[1] * volRes[2], 0);
    std::memset(&k->counters, 0, sizeof k->counters);
    k->count_raycast = false;
    orc_kfusion_reset(k);
    return k;
}

void orc_kfusion_destroy(orc_kfusion *k) { delete k; }

// cpp:135-137
void orc_kfusion_reset(orc_kfusion *k) { orc_init_volume(k->volume.data(), k->volRes[0], k->volRes[1], k->volRes[2]); }

// cpp:915-922
int orc_kfusion_preprocessing(orc_kfusion *k, const unsigned short *inputDepth, unsigned inW, unsigned inH) {
    if (orc_mm2meters(k->floatDepth.data(), k->W, k->H, inputDepth, inW, inH)) {
        fprintf(stderr, "Invalid ratio.\n");
        return -1;
    }
    orc_bilateral_filter(k->scaledDepth[0].data(), k->floatDepth.data(), k->W, k->H, k->gaussian.data(), e_delta, radius);
    return 1;
}

// cpp:924-971
int orc_kfusion_tracking(orc_kfusion *k, const float k4[4], float icp_threshold, unsigned tracking_rate, unsigned frame) {
    if (frame % tracking_rate != 0) return 0;
    const int L = (int)k->iterations.size();
    for (int i = 1; i < L; ++i)
        orc_half_sample(k->scaledDepth[i].data(), k->scaledDepth[i - 1].data(), level_div(k->W, i - 1),
                        level_div(k->H, i - 1), e_delta * 3, 1);
    unsigned lw = k->W, lh = k->H;
    for (int i = 0; i < L; ++i) {
        const float s = float(1 << i);
        const f4 kl = {k4[0] / s, k4[1] / s, k4[2] / s, k4[3] / s};
        const M4 invK = inverse_camera_matrix(kl);
        orc_depth2vertex(k->inputVertex[i].data(), k->scaledDepth[i].data(), lw, lh, &invK.m[0][0]);
        orc_vertex2normal(k->inputNormal[i].data(), k->inputVertex[i].data(), lw, lh);
        lw /= 2; lh /= 2;
    }
    k->oldPose = k->pose;
    const f4 kk = {k4[0], k4[1], k4[2], k4[3]};
    M4 projectReference;
    if (k->have_reference) projectReference = matmul(camera_matrix(kk), inverse(k->raycastPose));
    else std::memset(&projectReference, 0, sizeof projectReference);
    for (int level = L - 1; level >= 0; --level) {
        const unsigned w = level_div(k->W, level), h = level_div(k->H, level);
        for (int i = 0; i < k->iterations[level]; ++i) {
            orc_track(k->trackingResult.data(), k->inputVertex[level].data(), k->inputNormal[level].data(), w, h,
                      k->vertex.data(), k->normal.data(), k->W, k->H, &k->pose.m[0][0], &projectReference.m[0][0],
                      dist_threshold, normal_threshold, k->have_reference ? 1 : 0);
            orc_reduce(k->reduction.data(), k->trackingResult.data(), k->W, k->H, w, h);
            k->counters.icp_iterations++;
            if (orc_update_pose(&k->pose.m[0][0], k->reduction.data(), icp_threshold)) break;
        }
    }
    return orc_check_pose(&k->pose.m[0][0], &k->oldPose.m[0][0], k->reduction.data(), k->W, k->H, track_threshold);
}

// cpp:988-1004
int orc_kfusion_integration(orc_kfusion *k, const float k4[4], unsigned integration_rate, float mu, unsigned frame) {
    int doIntegrate = orc_check_pose(&k->pose.m[0][0], &k->oldPose.m[0][0], k->reduction.data(), k->W, k->H, track_threshold);
    if ((doIntegrate && ((frame % integration_rate) == 0)) || (frame <= 3)) {
        const f4 kk = {k4[0], k4[1], k4[2], k4[3]};
        const M4 invTrack = inverse(k->pose), K = camera_matrix(kk);
        orc_integrate(k->volume.data(), k->volRes, k->volDim, k->floatDepth.data(), k->W, k->H, &invTrack.m[0][0],
                      &K.m[0][0], mu, maxweight, &k->counters);
        doIntegrate = 1;
    } else {
        doIntegrate = 0;
    }
    return doIntegrate;
}

// cpp:973-986 (always returns false in the reference)
int orc_kfusion_raycasting(orc_kfusion *k, const float k4[4], float mu, unsigned frame) {
    if (frame > 2) {
        k->raycastPose = k->pose;
        k->have_reference = true;
        const f4 kk = {k4[0], k4[1], k4[2], k4[3]};
        const M4 view = matmul(k->raycastPose, inverse_camera_matrix(kk));
        orc_raycast(k->vertex.data(), k->normal.data(), k->W, k->H, k->volume.data(), k->volRes, k->volDim,
                    &view.m[0][0], nearPlane, farPlane, k->step, 0.75f * mu, k->count_raycast ? &k->counters : nullptr);
    }
    return 0;
}

// cpp:1044-1046
void orc_kfusion_render_depth(orc_kfusion *k, unsigned char *out4, unsigned W, unsigned H) {
    orc_render_depth(out4, k->floatDepth.data(), W, H, nearPlane, farPlane);
}
// cpp:1040-1042
void orc_kfusion_render_track(orc_kfusion *k, unsigned char *out4, unsigned W, unsigned H) {
    orc_render_track(out4, k->trackingResult.data(), W, H);
}
// cpp:1032-1038
void orc_kfusion_render_volume(orc_kfusion *k, unsigned char *out4, unsigned W, unsigned H, int frame, int rate,
                               const float k4[4], float largestep) {
    if (frame % rate != 0) return;
    const f4 kk = {k4[0], k4[1], k4[2], k4[3]};
    const M4 view = matmul(k->pose, inverse_camera_matrix(kk));
    orc_render_volume(out4, W, H, k->volume.data(), k->volRes, k->volDim, &view.m[0][0], nearPlane, farPlane * 2.0f,
                      k->step, largestep, light, ambient);
}

void orc_kfusion_get_pose(orc_kfusion *k, float out16[16]) { std::memcpy(out16, &k->pose, sizeof(M4)); }
void orc_kfusion_set_pose(orc_kfusion *k, const float in16[16]) { std::memcpy(&k->pose, in16, sizeof(M4)); }
// kernels.h:147-154
void orc_kfusion_get_position(orc_kfusion *k, float out3[3]) {
    for (int i = 0; i < 3; ++i) out3[i] = k->pose.m[i][3] - k->initPose[i];
}

// cpp:1006-1030: tsdf short only, x fastest
int orc_kfusion_dump_volume(orc_kfusion *k, const char *filename) {
    if (!filename) return 0;
    FILE *f = fopen(filename, "wb");
    if (!f) return -1;
    const size_t n = (size_t)k->volRes[0] * k->volRes[1] * k->volRes[2];
    std::vector<short> tsdf(n);
    for (size_t i = 0; i < n; ++i) tsdf[i] = k->volume[2 * i];
    fwrite(tsdf.data(), sizeof(short), n, f);
    fclose(f);
    return 0;
}

orc_counters *orc_kfusion_counters(orc_kfusion *k) { return &k->counters; }
void orc_kfusion_set_counting(orc_kfusion *k, int raycast_distinct) { k->count_raycast = raycast_distinct != 0; }

void *orc_kfusion_buffer(orc_kfusion *k, const char *name, unsigned long long *nbytes) {
    const std::string n(name);
    auto ret = [&](void *p, size_t b) -> void * { if (nbytes) *nbytes = b; return p; };
    if (n == "floatDepth") return ret(k->floatDepth.data(), k->floatDepth.size() * 4);
    if (n == "vertex") return ret(k->vertex.data(), k->vertex.size() * 4);
    if (n == "normal") return ret(k->normal.data(), k->normal.size() * 4);
    if (n == "trackingResult") return ret(k->trackingResult.data(), k->trackingResult.size() * sizeof(TrackRec));
    if (n == "reduction") return ret(k->reduction.data(), k->reduction.size() * 4);
    if (n == "volume") return ret(k->volume.data(), k->volume.size() * 2);
    if (n == "raycastPose") return ret(&k->raycastPose, sizeof(M4));
    if (n == "oldPose") return ret(&k->oldPose, sizeof(M4));
    for (size_t l = 0; l < k->iterations.size(); ++l) {
        const std::string s = std::to_string(l);
        if (n == "scaledDepth" + s) return ret(k->scaledDepth[l].data(), k->scaledDepth[l].size() * 4);
        if (n == "inputVertex" + s) return ret(k->inputVertex[l].data(), k->inputVertex[l].size() * 4);
        if (n == "inputNormal" + s) return ret(k->inputNormal[l].data(), k->inputNormal[l].size() * 4);
    }
    if (nbytes) *nbytes = 0;
    return nullptr;
}

}  // extern "C"
