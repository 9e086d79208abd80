// Host-side cost of the calls an upload of one 640x480 depth frame is made of (microseconds per call, mean of 300):
//   hipcc --offload-arch=gfx950 -O2 profiles/tools/upload_cost.hip -o /tmp/upload_cost && /tmp/upload_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
__global__ void k_copy(uint4 *dst, const uint4 *src, unsigned n) {
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ void k_nop() {}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t bytes = 640 * 480 * 2, frames = 64;
    std::vector<unsigned char> src(bytes * frames, 1);
    void *pin[2], *dev[2];
    hipStream_t s, cs;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking); hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
    hipEvent_t e[2];
    for (int i = 0; i < 2; ++i) { hipHostMalloc(&pin[i], bytes, hipHostMallocDefault); hipMalloc(&dev[i], bytes); hipEventCreateWithFlags(&e[i], hipEventDisableTiming); }
    const int N = 300;
    double t_memcpy = 0, t_async = 0, t_rec = 0, t_wait = 0, t_kernel = 0, t_copyk = 0, t_evsync = 0;
    for (int i = 0; i < N + 10; ++i) {
        const int slot = i & 1;
        double t0 = now();
        hipEventSynchronize(e[slot]);
        double t1 = now();
        memcpy(pin[slot], src.data() + (size_t)(i % frames) * bytes, bytes);
        double t2 = now();
        hipMemcpyAsync(dev[slot], pin[slot], bytes, hipMemcpyHostToDevice, cs);
        double t3 = now();
        hipEventRecord(e[slot], cs);
        double t4 = now();
        hipStreamWaitEvent(s, e[slot], 0);
        double t5 = now();
        k_nop<<<1, 64, 0, s>>>();
        double t6 = now();
        k_copy<<<64, 256, 0, cs>>>((uint4 *)dev[slot], (const uint4 *)pin[slot], (unsigned)(bytes / 16));
        double t7 = now();
        if (i >= 10) { t_evsync += t1 - t0; t_memcpy += t2 - t1; t_async += t3 - t2; t_rec += t4 - t3; t_wait += t5 - t4; t_kernel += t6 - t5; t_copyk += t7 - t6; }
        if ((i & 7) == 7) { hipStreamSynchronize(s); hipStreamSynchronize(cs); }
    }
    hipDeviceSynchronize();
    printf("per call, us: hipEventSynchronize %.1f | memcpy 614 KB into pinned %.1f | hipMemcpyAsync H2D %.1f | hipEventRecord %.1f | hipStreamWaitEvent %.1f | empty kernel launch %.1f | copy-kernel launch %.1f\n",
           t_evsync / N, t_memcpy / N, t_async / N, t_rec / N, t_wait / N, t_kernel / N, t_copyk / N);
    // device time of the copy kernel reading pinned memory over PCIe
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int blocks : {16, 64, 256}) {
        hipEventRecord(a, cs);
        for (int i = 0; i < 20; ++i) k_copy<<<blocks, 256, 0, cs>>>((uint4 *)dev[0], (const uint4 *)pin[0], (unsigned)(bytes / 16));
        hipEventRecord(b, cs); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("copy kernel, %3d blocks: %.1f us per 614 KB frame (%.1f GB/s)\n", blocks, ms * 1e3 / 20, bytes / (ms * 1e-3 / 20) / 1e9);
    }
    hipEventRecord(a, cs);
    for (int i = 0; i < 20; ++i) hipMemcpyAsync(dev[0], pin[0], bytes, hipMemcpyHostToDevice, cs);
    hipEventRecord(b, cs); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("hipMemcpyAsync: %.1f us per frame on the device side\n", ms * 1e3 / 20);
    return 0;
}
