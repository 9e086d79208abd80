// What a grid barrier between two "iterations" of one launch costs against the boundary between two dependent launches (4.4 us,
// launch_cost.hip), for a launch shaped like an ICP iteration: G workgroups x 1024 threads, each leaves a 32-float row, every
// workgroup then needs all rows.  Variants:
//   fence   : __threadfence() (L2 write-back / invalidate), returning atomic add, spin with s_sleep (k_icp_tail's barrier)
//   lean    : rows written and read with agent-scope (sc1) accesses that bypass the non-coherent L2 path, no fence:
//             store rows -> s_waitcnt -> no-return atomic add -> spin on an agent-scope load
//   hipcc --offload-arch=gfx950 -O3 profiles/tools/grid_barrier.hip -o /tmp/grid_barrier && /tmp/grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int LEAN>
__global__ __launch_bounds__(1024) void k_loop(float *rows, unsigned *counter, int iters, unsigned long long *cycles, float *out, unsigned *stale) {
    __shared__ float sm[32][33];
    const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
    float acc = (float)threadIdx.x;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        float *my = rows + ((size_t)(it & 1) * gridDim.x + blockIdx.x) * 32;
        // "block reduction": 32 values per workgroup
        if (threadIdx.x < 32) {
            const float v = (float)(threadIdx.x + 32 * (it & 1023)) + 0.25f * (float)(blockIdx.x & 3);    // (exact in float)
            if (LEAN) __hip_atomic_store(my + threadIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else my[threadIdx.x] = v;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned goal = (unsigned)(it + 1) * gridDim.x;
            if (LEAN) {
                __builtin_amdgcn_s_waitcnt(0);      // the workgroup's row stores are acknowledged (write-through)
                __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (int spin = 0; spin < 2000000 && __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < goal; ++spin) { }   // (bounded: the grid always drains)
            } else {
                __threadfence();
                __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (int spin = 0; spin < 2000000 && __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < goal; ++spin) __builtin_amdgcn_s_sleep(2);
                __threadfence();
            }
        }
        __syncthreads();
        // every workgroup adds up all rows (the ICP prologue) - and checks that each is the one written in THIS iteration
        const float *all = rows + (size_t)(it & 1) * gridDim.x * 32;
        float s = 0.f;
        for (unsigned b = g; b < gridDim.x; b += 32) {
            const float r = LEAN ? __hip_atomic_load(all + (size_t)b * 32 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : all[(size_t)b * 32 + c];
            if (r != (float)(c + 32 * (it & 1023)) + 0.25f * (float)(b & 3)) atomicAdd(stale, 1u);
            s += r;
        }
        sm[g][c] = s;
        __syncthreads();
        if (threadIdx.x < 32) { float v = 0.f; for (int k = 0; k < 32; ++k) v += sm[k][threadIdx.x]; acc = v * 1e-9f; }
        __syncthreads();
        acc = sm[0][0] * 1e-9f + (float)threadIdx.x;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { cycles[blockIdx.x] = t1 - t0; out[blockIdx.x] = acc; }
}

int main() {
    const int ITERS = 2000;
    float *rows, *out; unsigned *counter, *stale; unsigned long long *cycles;
    CK(hipMalloc(&rows, 2 * 256 * 32 * 4)); CK(hipMalloc(&out, 256 * 4)); CK(hipMalloc(&counter, 4)); CK(hipMalloc(&stale, 4)); CK(hipMemset(stale, 0, 4)); CK(hipMalloc(&cycles, 256 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grids[4] = {256, 128, 75, 19};
    for (int lean = 0; lean < 2; ++lean) for (int gi = 0; gi < 4; ++gi) {
        const int G = grids[gi];
        float best = 1e9f;
        std::vector<unsigned long long> h(G);
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(counter, 0, 4)); CK(hipMemset(rows, 0, 2 * 256 * 32 * 4));
            CK(hipEventRecord(e0, 0));
            if (lean) k_loop<1><<<G, 1024>>>(rows, counter, ITERS, cycles, out, stale); else k_loop<0><<<G, 1024>>>(rows, counter, ITERS, cycles, out, stale);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        CK(hipMemcpy(h.data(), cycles, G * 8, hipMemcpyDeviceToHost));
        unsigned hs = 0; CK(hipMemcpy(&hs, stale, 4, hipMemcpyDeviceToHost)); CK(hipMemset(stale, 0, 4));
        printf("%-6s workgroups %3d: %6.2f us per iteration (row store + barrier + all-rows sum), %llu cycles/iteration on workgroup 0, stale rows read: %u of %d\n",
               lean ? "lean" : "fence", G, best * 1e3f / ITERS, h[0] / ITERS, hs, 3 * ITERS * G * G * 32);
    }
    return 0;
}
