// VALU issue rates that decide how k_integrate should march (gfx950): v_pk_add_f32 against two v_add_f32, IEEE division,
// v_rcp_f32, at 1 / 2 / 4 / 8 waves per SIMD.  Each thread runs N iterations of 6 independent float chains.
//   hipcc --offload-arch=gfx950 -O3 profiles/tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int n, float d) {
    v2f a = {1.f + threadIdx.x, 2.f}, b = {3.f, 4.f}, c = {5.f, 6.f};
    const v2f dd = {d, d * 1.5f};
    float s0 = a.x, s1 = a.y, s2 = b.x, s3 = b.y, s4 = c.x, s5 = c.y;
    for (int i = 0; i < n; ++i) {
        if (MODE == 0) {          // 3 packed adds
            asm volatile("v_pk_add_f32 %0, %0, %3\n v_pk_add_f32 %1, %1, %3\n v_pk_add_f32 %2, %2, %3" : "+v"(a), "+v"(b), "+v"(c) : "v"(dd));
        } else if (MODE == 1) {   // 6 scalar adds
            asm volatile("v_add_f32 %0, %0, %6\n v_add_f32 %1, %1, %6\n v_add_f32 %2, %2, %6\n v_add_f32 %3, %3, %6\n v_add_f32 %4, %4, %6\n v_add_f32 %5, %5, %6"
                         : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5) : "v"(d));
        } else if (MODE == 2) {   // 6 rcp
            asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5"
                         : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5));
        } else if (MODE == 3) {   // 3 packed fma
            asm volatile("v_pk_fma_f32 %0, %0, %3, %3\n v_pk_fma_f32 %1, %1, %3, %3\n v_pk_fma_f32 %2, %2, %3, %3" : "+v"(a), "+v"(b), "+v"(c) : "v"(dd));
        } else {                  // 6 scalar fma
            asm volatile("v_fma_f32 %0, %0, %6, %6\n v_fma_f32 %1, %1, %6, %6\n v_fma_f32 %2, %2, %6, %6\n v_fma_f32 %3, %3, %6, %6\n v_fma_f32 %4, %4, %6, %6\n v_fma_f32 %5, %5, %6, %6"
                         : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5) : "v"(d));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a.x + a.y + b.x + b.y + c.x + c.y + s0 + s1 + s2 + s3 + s4 + s5;
}

int main() {
    float *out; hipMalloc(&out, 256 * 4096 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int n = 20000;
    const char *names[5] = {"3 x v_pk_add_f32", "6 x v_add_f32", "6 x v_rcp_f32", "3 x v_pk_fma_f32", "6 x v_fma_f32"};
    printf("%-18s %8s %14s %22s\n", "body", "waves/SIMD", "us", "cycles per iteration per SIMD (2.4 GHz)");
    for (int mode = 0; mode < 5; ++mode)
        for (int wps = 1; wps <= 8; wps *= 2) {
            const int blocks = 256 * wps;   // 256 CUs x 4 SIMDs x wps waves = blocks x 4 waves
            float ms = 0, best = 1e9f;
            for (int r = 0; r < 5; ++r) {
                hipEventRecord(e0);
                switch (mode) {
                case 0: k<0><<<blocks, 256>>>(out, n, 1e-3f); break;
                case 1: k<1><<<blocks, 256>>>(out, n, 1e-3f); break;
                case 2: k<2><<<blocks, 256>>>(out, n, 1e-3f); break;
                case 3: k<3><<<blocks, 256>>>(out, n, 1e-3f); break;
                default: k<4><<<blocks, 256>>>(out, n, 1e-3f); break;
                }
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best;
            }
            printf("%-18s %8d %14.1f %22.2f\n", names[mode], wps, best * 1e3, best * 1e-3 * 2.4e9 / n / wps);
        }
    return 0;
}
