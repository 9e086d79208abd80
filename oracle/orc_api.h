// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header).  PARITY UNPINNED.
// C entry points of the CPU restatement, loaded with ctypes by tests/, smoke() and
// bench.py's cpu_baseline leg.  Matrices are 16 floats row-major (commons.h:317-319);
// vertex/normal maps are packed float3 (12 B) as in the reference; TrackData is the
// 32-byte {int result; float error; float J[6]} of commons.h:325-329; the TSDF is
// short2 per voxel, x fastest (commons.h:160-185).
#pragma once
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_counters {
    unsigned long long integrate_updated;       // U of SURVEY §8(d): voxels that passed all tests
    unsigned long long raycast_distinct_voxels; // V of SURVEY §8(d): distinct voxels sampled by the last raycast
    unsigned long long icp_iterations;          // track+reduce passes actually executed
} orc_counters;

void orc_gaussian(float *g, int radius, float delta);
void orc_init_volume(short *vol, unsigned sx, unsigned sy, unsigned sz);
int orc_mm2meters(float *out, unsigned outW, unsigned outH, const unsigned short *in, unsigned inW, unsigned inH);
void orc_bilateral_filter(float *out, const float *in, unsigned W, unsigned H, const float *gaussian, float e_d, int r);
void orc_half_sample(float *out, const float *in, unsigned inW, unsigned inH, float e_d, int r);
void orc_depth2vertex(float *vertex, const float *depth, unsigned W, unsigned H, const float *invK16);
void orc_vertex2normal(float *out, const float *in, unsigned W, unsigned H);
void orc_track(void *output, const float *inVertex, const float *inNormal, unsigned inW, unsigned inH,
               const float *refVertex, const float *refNormal, unsigned refW, unsigned refH, const float *Ttrack16,
               const float *view16, float dist_threshold, float normal_threshold, int have_reference);
void orc_reduce(float *out, const void *J, unsigned JW, unsigned JH, unsigned W, unsigned H);
void orc_set_omp_threads(int n);                      /* -fopenmp build only */
void orc_set_reduce_emulation(int threads);           /* either build: > 0 = the reference's OpenMP reduction (cpp:295-296) with that many threads, fixed combine order; 0 = off */
void orc_set_omp_reduce_mode(int reference_style);   /* -fopenmp build only: 1 = re-associated like the reference's OpenMP build */
int orc_update_pose(float *pose16, const float *output, float icp_threshold);
int orc_check_pose(float *pose16, const float *oldPose16, const float *output, unsigned W, unsigned H, float track_threshold);
void orc_solve6(const float *vals27, double *x6);
void orc_se3_exp(const double *mu6, float *out16);
float orc_h_fminf(float a, float b);
float orc_h_fmaxf(float a, float b);
float orc_h_clampf(float f, float a, float b);
int orc_h_border_clamp(int x, int i, int size);
int orc_h_clamp_int(int v, int lo, int hi);
float orc_h_dot3(const float *a, const float *b);
void orc_h_cross3(const float *a, const float *b, float *o);
float orc_h_length3(const float *a);
void orc_h_normalize3(const float *a, float *o);
void orc_h_floor_frac(float v, float *fl, float *fr, int *as_int);
void orc_h_constants(float *out12);
void orc_inverse(const float *in16, float *out16);
void orc_matmul(const float *a16, const float *b16, float *out16);
void orc_integrate(short *vol, const unsigned size[3], const float dim[3], const float *depth, unsigned W, unsigned H,
                   const float *invTrack16, const float *K16, float mu, float maxweight, orc_counters *cnt);
void orc_raycast(float *vertex, float *normal, unsigned W, unsigned H, const short *vol, const unsigned size[3],
                 const float dim[3], const float *view16, float nearPlane, float farPlane, float step, float largestep,
                 orc_counters *cnt);
void orc_render_depth(unsigned char *out4, const float *depth, unsigned W, unsigned H, float nearPlane, float farPlane);
void orc_render_track(unsigned char *out4, const void *data, unsigned W, unsigned H);
void orc_render_volume(unsigned char *out4, unsigned W, unsigned H, const short *vol, const unsigned size[3],
                       const float dim[3], const float *view16, float nearPlane, float farPlane, float step,
                       float largestep, const float light3[3], const float ambient3[3]);

// ---- pipeline object: restates class Kfusion (kernels.h:83-195) + cpp:67-137,915-1055 ----
typedef struct orc_kfusion orc_kfusion;
orc_kfusion *orc_kfusion_create(unsigned W, unsigned H, const unsigned volRes[3], const float volDim[3],
                                const float initPose3[3], const int *pyramid, int nLevels);
void orc_kfusion_destroy(orc_kfusion *);
void orc_kfusion_reset(orc_kfusion *);
int orc_kfusion_preprocessing(orc_kfusion *, const unsigned short *inputDepth, unsigned inW, unsigned inH);
int orc_kfusion_tracking(orc_kfusion *, const float k4[4], float icp_threshold, unsigned tracking_rate, unsigned frame);
int orc_kfusion_integration(orc_kfusion *, const float k4[4], unsigned integration_rate, float mu, unsigned frame);
int orc_kfusion_raycasting(orc_kfusion *, const float k4[4], float mu, unsigned frame);
void orc_kfusion_render_depth(orc_kfusion *, unsigned char *out4, unsigned W, unsigned H);
void orc_kfusion_render_track(orc_kfusion *, unsigned char *out4, unsigned W, unsigned H);
void orc_kfusion_render_volume(orc_kfusion *, unsigned char *out4, unsigned W, unsigned H, int frame, int rate,
                               const float k4[4], float largestep);
void orc_kfusion_get_pose(orc_kfusion *, float out16[16]);
void orc_kfusion_set_pose(orc_kfusion *, const float in16[16]);
void orc_kfusion_get_position(orc_kfusion *, float out3[3]);
int orc_kfusion_dump_volume(orc_kfusion *, const char *filename);
orc_counters *orc_kfusion_counters(orc_kfusion *);
void orc_kfusion_set_counting(orc_kfusion *, int raycast_distinct);
// Internal buffers for parity checks.  name: "floatDepth", "scaledDepth<l>", "inputVertex<l>",
// "inputNormal<l>", "vertex", "normal", "trackingResult", "reduction", "volume", "raycastPose", "oldPose".
// Returns a borrowed pointer and writes the byte size (0 if unknown).
void *orc_kfusion_buffer(orc_kfusion *, const char *name, unsigned long long *nbytes);

#ifdef __cplusplus
}
#endif
