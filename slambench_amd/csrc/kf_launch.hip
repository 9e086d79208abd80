// kf_launch.hip - launch-side state shared by the kernel translation units: the per-thread launch error, the dispatch sort as a
// launch of its own, the one-off device self-checks of the reciprocal-form divisions and of the squared-length test.
#include "kf_dev.h"

namespace kf {

__global__ __launch_bounds__(1024) void k_ray_order(unsigned *__restrict__ order, unsigned *__restrict__ cost, unsigned tiles_x, unsigned tiles_y,
                                                    unsigned mode) {
    __shared__ RayOrderShared sh;
    ray_order_block(sh, order, cost, tiles_x, tiles_y, mode);
}

// Division by a launch-wide constant (the volume dimension in commons.h:193-195) without the 11-instruction IEEE
// sequence: with r = RN(1/d), q = a*r followed by one residual correction (e = fma(-d, q, a); q = fma(e, r, q)) is the
// correctly rounded quotient for every numerator the march can produce.  That is not taken on trust: the first time a
// divisor is seen, k_verify_div compares the short form with the IEEE quotient for EVERY float in [2^-100, 2^24) - 1.04e9
// values; negative numerators follow by symmetry, smaller ones give +-tiny either way, which the following "- 0.5f"
// absorbs, and pos*size stays below 2^24 for every volume the launchers accept - and one mismatch keeps the IEEE division
// for that volume (template flag FD of the view).
__global__ __launch_bounds__(256) void k_verify_div(float d, float r, unsigned long long *mismatches) {
    const unsigned long long n = 124ull << 23;   // biased exponents 27..150
    unsigned bad = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        const float a = __builtin_bit_cast(float, (27u << 23) + (unsigned)i);
        bad += (__builtin_bit_cast(unsigned, a / d) != __builtin_bit_cast(unsigned, div_by_const(a, d, r))) ? 1u : 0u;
    }
    if (bad) atomicAdd(mismatches, (unsigned long long)bad);
}

// sqrtf(d) > t against d > bound for every non-negative float bit pattern (see sqrt_gt_bound in kf_math.h)
__global__ __launch_bounds__(256) void k_verify_sqrt_bound(float t, float bound, unsigned long long *mismatches) {
    const unsigned long long n = 1ull << 31;
    unsigned bad = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        const float d = __builtin_bit_cast(float, (unsigned)i);
        bad += ((sqrtf(d) > t) != (d > bound)) ? 1u : 0u;
    }
    if (bad) atomicAdd(mismatches, (unsigned long long)bad);
}

namespace launch {

static thread_local char g_launch_error[320] = "";
void launched(const char *kernel) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess && !g_launch_error[0])
        snprintf(g_launch_error, sizeof g_launch_error, "launch of %s failed: %s", kernel, hipGetErrorString(e));
}
void refused(const char *why) {
    if (!g_launch_error[0]) snprintf(g_launch_error, sizeof g_launch_error, "%s", why);
}
const char *take_launch_error() {
    static thread_local char out[320];
    if (!g_launch_error[0]) return nullptr;
    snprintf(out, sizeof out, "%s", g_launch_error);
    g_launch_error[0] = 0;
    return out;
}

void dispatch_order(unsigned *order, unsigned *cost, unsigned tiles_x, unsigned tiles_y, unsigned mode, hipStream_t s) {
    k_ray_order<<<1, 1024, 0, s>>>(order, cost, tiles_x, tiles_y, mode); launched("k_ray_order");
}

__global__ void k_iota(unsigned *p, unsigned n) {
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = i;
}
void init_ray_order(unsigned *tile_order, unsigned *tile_cost, unsigned n, hipStream_t s) {
    k_iota<<<(n + 255) / 256, 256, 0, s>>>(tile_order, n); launched("k_iota");
    hipMemsetAsync(tile_cost, 0, (size_t)2 * n * sizeof(unsigned), s);
}

// Has the reciprocal form of a / d been verified for this divisor on this device (see k_verify_div)?  Cached; the check
// takes a few milliseconds once per context (kfh_create calls prepare_volume_division).  KFH_EXACT_DIV=0 keeps IEEE.
struct DivisorEntry { int dev; float d; bool ok; };
static std::vector<DivisorEntry> g_divisors;
static std::mutex g_divisors_mu;
// what an earlier verified_divisor() of this device found; false for a divisor nobody has submitted yet (frame-path launchers
// ask this one: it never launches or waits)
bool divisor_verified(float d) {
    int dev = 0;
    hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_divisors_mu);
    for (const DivisorEntry &e : g_divisors) if (e.dev == dev && e.d == d) return e.ok;
    return false;
}
bool verified_divisor(float d, hipStream_t s) {
    typedef DivisorEntry Entry;
    std::vector<Entry> &cache = g_divisors;
    std::mutex &mu = g_divisors_mu;
    static const bool allow = [] { const char *e = getenv("KFH_EXACT_DIV"); return !(e && e[0] == '0'); }();
    int dev = 0;
    hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    for (const Entry &e : cache) if (e.dev == dev && e.d == d) return e.ok;
    bool ok = false;
    if (allow && d > 0x1p-10f && d < 0x1p10f) {
        unsigned long long *dm = nullptr;
        if (hipMalloc(&dm, sizeof *dm) == hipSuccess) {
            unsigned long long h = 1;
            hipMemsetAsync(dm, 0, sizeof *dm, s);
            k_verify_div<<<8192, 256, 0, s>>>(d, 1.0f / d, dm); launched("k_verify_div");
            if (hipMemcpyAsync(&h, dm, sizeof h, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess) ok = h == 0;
            hipFree(dm);
        }
    }
    cache.push_back(Entry{dev, d, ok});
    return ok;
}

// Is "sqrtf(d) > t  <=>  d > sqrt_gt_bound(t)" verified for this threshold on this device?  prepare_sqrt_bound (kfh_create) runs
// k_verify_sqrt_bound over EVERY non-negative float bit pattern (2^31 of them: zeros, denormals, infinity and the NaNs included) with
// the square root this library is compiled with; the frame path only looks the verdict up.  Not verified = the ICP association takes
// the square root as the reference does (cpp:543).
struct SqrtBoundEntry { int dev; float t; bool ok; };
static std::vector<SqrtBoundEntry> g_sqrt_bounds;
static std::mutex g_sqrt_bounds_mu;
// returns the bound to compare squared lengths with (>= 0) or -1 = take the square root; the context keeps it
float prepare_sqrt_bound(float t, hipStream_t s) {
    static const bool allow = [] { const char *e = getenv("KFH_EXACT_SQRT"); return !(e && e[0] == '0'); }();
    int dev = 0;
    hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_sqrt_bounds_mu);
    const float bound = sqrt_gt_bound(t);
    for (const SqrtBoundEntry &e : g_sqrt_bounds) if (e.dev == dev && e.t == t) return e.ok ? bound : -1.f;
    bool ok = false, ran = true;
    if (allow && bound >= 0.f) {
        unsigned long long *dm = nullptr;
        ran = hipMalloc(&dm, sizeof *dm) == hipSuccess;
        if (ran) {
            unsigned long long h = 1;
            ran = hipMemsetAsync(dm, 0, sizeof *dm, s) == hipSuccess;
            if (ran) { k_verify_sqrt_bound<<<8192, 256, 0, s>>>(t, bound, dm); ran = hipGetLastError() == hipSuccess; }
            ran = ran && hipMemcpyAsync(&h, dm, sizeof h, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
            hipFree(dm);
            ok = ran && h == 0;
        }
    }
    if (ran) g_sqrt_bounds.push_back(SqrtBoundEntry{dev, t, ok});      // (a check that could not run is not remembered)
    return ok ? bound : -1.f;
}

// the bilateral filter's range divisor 2 e_d^2 (its numerators, squared depth differences, stay far below 2^24 for depth in metres)
void prepare_bilateral_division(float e_d, hipStream_t s) { (void)verified_divisor(e_d * e_d * 2, s); }

}  // namespace launch
}  // namespace kf
