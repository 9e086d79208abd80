// Launch-boundary cost of a chain of dependent kernels on one stream: plain launches vs one hipGraph replay
// (VERDICT r1 item 6 / SURVEY §7 step 6).  The kernels are shaped like a queued frame's ICP loop: 16 launches, each
// 256 workgroups x 1024 threads (or 19 / 75 workgroups), each reading what the previous one wrote and doing ~W us of work.
//   hipcc --offload-arch=gfx950 -O3 profiles/tools/graph_gap.hip -o gpurun_out/graph_gap && gpurun_out/graph_gap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(1024) void k_link(const float *__restrict__ in, float *__restrict__ out, int spin) {
    // dependent on the previous launch through `in`; `spin` dependent FMAs of work per thread
    float v = in[(blockIdx.x * 1024 + threadIdx.x) & 0xFFFF];
    for (int i = 0; i < spin; ++i) v = __builtin_fmaf(v, 1.0000001f, 1e-9f);
    out[(blockIdx.x * 1024 + threadIdx.x) & 0xFFFF] = v;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const int CHAIN = 16, REPS = 300;
    float *a = nullptr, *b = nullptr;
    CK(hipMalloc(&a, 65536 * 4)); CK(hipMalloc(&b, 65536 * 4));
    CK(hipMemset(a, 0, 65536 * 4)); CK(hipMemset(b, 0, 65536 * 4));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int grids[3] = {256, 75, 19};
    const int spins[3] = {0, 2000, 8000};           // ~0, ~4, ~15 us of work per launch
    printf("%-8s %-8s %14s %14s %14s\n", "blocks", "spin", "stream us/krn", "graph us/krn", "1 krn alone us");
    for (int gi = 0; gi < 3; ++gi) for (int si = 0; si < 3; ++si) {
        const int grid = grids[gi], spin = spins[si];
        auto chain = [&](hipStream_t st) {
            for (int i = 0; i < CHAIN; ++i) k_link<<<grid, 1024, 0, st>>>(i & 1 ? b : a, i & 1 ? a : b, spin);
        };
        // warm up
        for (int r = 0; r < 20; ++r) chain(s);
        CK(hipStreamSynchronize(s));
        double t0 = now();
        for (int r = 0; r < REPS; ++r) chain(s);
        CK(hipStreamSynchronize(s));
        const double t_stream = (now() - t0) / (REPS * CHAIN) * 1e6;
        // the same chain as a graph (stream capture), replayed
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        chain(s);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int r = 0; r < 20; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        t0 = now();
        for (int r = 0; r < REPS; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        const double t_graph = (now() - t0) / (REPS * CHAIN) * 1e6;
        // one kernel's own duration (events around a single launch in a drained queue)
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float ms = 0, best = 1e9f;
        for (int r = 0; r < 20; ++r) {
            CK(hipEventRecord(e0, s));
            k_link<<<grid, 1024, 0, s>>>(a, b, spin);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        printf("%-8d %-8d %14.2f %14.2f %14.2f\n", grid, spin, t_stream, t_graph, best * 1e3);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
