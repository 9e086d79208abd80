// What a launch of MANY short waves costs (k_raycast at 640x480: 9600 workgroups x 128 threads = 19 200 waves; a launch in which every
// ray ends after its set-up still takes ~15 us): the time of a launch by workgroup shape for the same 1.2 M threads, for a kernel
// that does nothing, one that reads a parameter block and an order entry first (two dependent round trips, as k_raycast's head), and
// one that also does ~170 vector instructions and two 16-byte stores per lane.
//   hipcc --offload-arch=gfx950 -O3 profiles/tools/wave_rate.hip -o /tmp/wave_rate && /tmp/wave_rate
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Params { float m[16]; int enabled; int pad[15]; };

template <int MODE, int VGPRS>
__global__ void k_many(const Params *__restrict__ p, const unsigned *__restrict__ order, float4 *__restrict__ out, int stores) {
    if (VGPRS >= 64) asm volatile("" : : : "v63");
    if (MODE == 0) return;
    const Params P = *p;
    const unsigned slot = order[blockIdx.x];
    if (!P.enabled) return;
    if (MODE == 1) { if (slot == 0xFFFFFFFFu) out[0] = make_float4(P.m[0], 0, 0, 0); return; }
    float x = (float)(slot * blockDim.x + threadIdx.x), y = P.m[1], z = P.m[2];
#pragma unroll
    for (int i = 0; i < 56; ++i) { x = x * P.m[i & 15] + y; y = y * z + x; z = z + x * 0.5f; }      // 3 dependent-ish ops per trip
    if (stores == 1 || (stores == 2 && (threadIdx.x & 3) == 0)) {      // 2: one lane in four, as the four lanes of a ray
        const size_t i = (size_t)slot * blockDim.x + threadIdx.x;
        out[i] = make_float4(x, y, z, 0.f);
        out[i + (size_t)gridDim.x * blockDim.x] = make_float4(z, y, x, 0.f);
    }
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <class K>
static int run(const char *name, K kernel, const Params *p, const unsigned *order, float4 *out, hipStream_t s, int stores = 1) {
    const int threads_total = 640 * 480 * 4;
    const int blocks[5] = {64, 128, 256, 512, 1024};
    for (int bi = 0; bi < 5; ++bi) {
        const int block = blocks[bi], grid = threads_total / block;
        for (int r = 0; r < 20; ++r) kernel<<<grid, block, 0, s>>>(p, order, out, stores);
        CK(hipStreamSynchronize(s));
        const int REPS = 400;
        const double t0 = now();
        for (int r = 0; r < REPS; ++r) kernel<<<grid, block, 0, s>>>(p, order, out, stores);
        CK(hipStreamSynchronize(s));
        printf("%-58s %5d workgroups x %4d threads  %6.2f us per launch\n", name, grid, block, (now() - t0) / REPS * 1e6);
    }
    return 0;
}

int main() {
    Params hp = {};
    for (int i = 0; i < 16; ++i) hp.m[i] = 1.f / (i + 2);
    hp.enabled = 1;
    Params *p = nullptr; unsigned *order = nullptr; float4 *out = nullptr;
    const size_t n = 640 * 480 * 4;
    CK(hipMalloc(&p, sizeof hp)); CK(hipMemcpy(p, &hp, sizeof hp, hipMemcpyHostToDevice));
    CK(hipMalloc(&order, 32768 * 4));
    unsigned *ho = new unsigned[32768];
    for (unsigned i = 0; i < 32768; ++i) ho[i] = i;
    CK(hipMemcpy(order, ho, 32768 * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&out, 2 * n * sizeof(float4)));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    if (run("empty kernel", k_many<0, 0>, p, order, out, s)) return 1;
    if (run("empty kernel, 64 VGPRs", k_many<0, 64>, p, order, out, s)) return 1;
    if (run("parameter block + order entry, then exit", k_many<1, 64>, p, order, out, s)) return 1;
    if (run("... + ~170 vector instructions, no store", k_many<2, 64>, p, order, out, s, 0)) return 1;
    if (run("... + ~170 vector instructions + two 16-byte stores by one lane in four (9.8 MB)", k_many<2, 64>, p, order, out, s, 2)) return 1;
    if (run("... + ~170 vector instructions + two 16-byte stores by every lane (39 MB)", k_many<2, 64>, p, order, out, s, 1)) return 1;
    return 0;
}
