// What a launch in a chain of dependent launches costs on its own, by what the kernel declares: registers, LDS, scratch,
// kernel-argument bytes, code size.  The chain is shaped like a frame's ICP loop (16 launches of 256 / 75 / 19 workgroups x 1024
// threads, each reading one word the previous launch wrote) and every kernel does the minimum: load, barrier, store.
//   hipcc --offload-arch=gfx950 -O3 profiles/tools/launch_cost.hip -o /tmp/launch_cost && /tmp/launch_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct BigArgs { float pad[60]; };      // 240 bytes, like IcpIterArgs

template <int VGPRS, int LDS_BYTES, int SCRATCH, int BIGARG>
__global__ __launch_bounds__(1024) void k_link(const float *__restrict__ in, float *__restrict__ out, BigArgs big, int n) {
    __shared__ float lds[LDS_BYTES / 4 + 1];
    float v = in[(blockIdx.x * 1024 + threadIdx.x) & 0xFFFF];
    if (VGPRS >= 128) asm volatile("" : : : "v127");
    else if (VGPRS >= 64) asm volatile("" : : : "v63");
    if (LDS_BYTES > 4) { lds[threadIdx.x] = v; __syncthreads(); v = lds[(threadIdx.x + 1) & 1023]; }
    if constexpr (SCRATCH > 0) {
        volatile float priv[SCRATCH > 0 ? SCRATCH : 1];
        for (int i = 0; i < SCRATCH; ++i) priv[i] = v + i;
        v = priv[n & (SCRATCH - 1)];       // dynamic index: stays in scratch memory
    }
    if (BIGARG) v += big.pad[n & 31];
    out[(blockIdx.x * 1024 + threadIdx.x) & 0xFFFF] = v;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <class K>
static int run(const char *name, K kernel, float *a, float *b, hipStream_t s) {
    const int CHAIN = 16, REPS = 300;
    BigArgs big = {};
    const int grids[3] = {256, 75, 19};
    for (int gi = 0; gi < 3; ++gi) {
        const int grid = grids[gi];
        auto chain = [&]() { for (int i = 0; i < CHAIN; ++i) kernel<<<grid, 1024, 0, s>>>(i & 1 ? b : a, i & 1 ? a : b, big, 0); };
        for (int r = 0; r < 20; ++r) chain();
        CK(hipStreamSynchronize(s));
        const double t0 = now();
        for (int r = 0; r < REPS; ++r) chain();
        CK(hipStreamSynchronize(s));
        printf("%-44s blocks %3d   %6.2f us per launch\n", name, grid, (now() - t0) / (REPS * CHAIN) * 1e6);
    }
    return 0;
}

int main() {
    float *a = nullptr, *b = nullptr;
    CK(hipMalloc(&a, 65536 * 4)); CK(hipMalloc(&b, 65536 * 4));
    CK(hipMemset(a, 0, 65536 * 4)); CK(hipMemset(b, 0, 65536 * 4));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    if (run("plain (few VGPRs, no LDS, no scratch)", k_link<0, 4, 0, 0>, a, b, s)) return 1;
    if (run("64 VGPRs", k_link<64, 4, 0, 0>, a, b, s)) return 1;
    if (run("128 VGPRs", k_link<128, 4, 0, 0>, a, b, s)) return 1;
    if (run("128 VGPRs + 8 KB LDS", k_link<128, 8192, 0, 0>, a, b, s)) return 1;
    if (run("128 VGPRs + 40 KB LDS", k_link<128, 40960, 0, 0>, a, b, s)) return 1;
    if (run("128 VGPRs + 32 B scratch", k_link<128, 4, 8, 0>, a, b, s)) return 1;
    if (run("128 VGPRs + 256 B scratch", k_link<128, 4, 64, 0>, a, b, s)) return 1;
    if (run("128 VGPRs + 240 B of arguments read", k_link<128, 4, 0, 1>, a, b, s)) return 1;
    if (run("128 VGPRs + 8 KB LDS + 32 B scratch + args", k_link<128, 8192, 8, 1>, a, b, s)) return 1;
    if (run("plain + 32 B scratch", k_link<0, 4, 8, 0>, a, b, s)) return 1;
    return 0;
}
