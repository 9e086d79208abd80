// ORACLE — TEST INFRASTRUCTURE ONLY.
// Driver for the only part of the reference hot path that compiles in this image without
// TooN: kfusion/thirdparty/{vector_types.h,cutil_math.h} and
// kfusion/include/constant_parameters.h, included as they lie under /root/reference
// (oracle/Makefile target `ref`, output oracle/_ref/cutil_probe — never committed).
// It evaluates the reference's own helper functions on fixed inputs and prints JSON that
// tests/golden/make_cutil_vectors.py stores as tests/golden/cutil_math_vectors.json.
// The restatement (orc_math.h) and the HIP device helpers are checked against that file.
#include <cstdio>
#include <cstring>
#include <cmath>
#include <limits>
#include <vector_types.h>
#include <cutil_math.h>
#include <constant_parameters.h>

static unsigned bits(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }

int main() {
    const float nan = std::numeric_limits<float>::quiet_NaN();
    const float inf = std::numeric_limits<float>::infinity();
    const float fv[] = {0.0f, -0.0f, 1.0f, -1.0f, 0.5f, 2.5f, -3.75f, 1e-30f, 3.0e38f, nan, inf, -inf,
                        0.1f, 0.30000001f, 32766.0f, 0.00003051944088f};
    const int nf = sizeof(fv) / sizeof(fv[0]);
    printf("{\n");
    // fminf / fmaxf / clamp(float) on all pairs (cutil_math.h:43-49, 972-974)
    printf("\"fminf\": [");
    for (int i = 0; i < nf; ++i) for (int j = 0; j < nf; ++j)
        printf("%s[%u,%u,%u]", (i + j) ? "," : "", bits(fv[i]), bits(fv[j]), bits(fminf(fv[i], fv[j])));
    printf("],\n\"fmaxf\": [");
    for (int i = 0; i < nf; ++i) for (int j = 0; j < nf; ++j)
        printf("%s[%u,%u,%u]", (i + j) ? "," : "", bits(fv[i]), bits(fv[j]), bits(fmaxf(fv[i], fv[j])));
    printf("],\n\"clampf\": [");
    for (int i = 0; i < nf; ++i)
        printf("%s[%u,%u]", i ? "," : "", bits(fv[i]), bits(clamp(fv[i], -1.f, 1.f)));
    // clamp(uint) as the bilateral filter calls it: clamp(x + i, 0u, size - 1) (cpp:181-182)
    printf("],\n\"clamp_uint_border\": [");
    bool first = true;
    const unsigned sizes[] = {1u, 5u, 640u};
    for (int s = 0; s < 3; ++s)
        for (unsigned x = 0; x < sizes[s]; x += (sizes[s] > 8 ? 213 : 1))
            for (int i = -2; i <= 2; ++i) {
                unsigned r = clamp(x + i, 0u, sizes[s] - 1);
                printf("%s[%u,%u,%d,%u]", first ? "" : ",", sizes[s], x, i, r);
                first = false;
            }
    // int clamp as halfSample uses it (cpp:609-614)
    printf("],\n\"clamp_int2\": [");
    first = true;
    for (int a = -3; a <= 12; a += 3) {
        int2 r = clamp(make_int2(a, a + 1), make_int2(0), make_int2(7, 9));
        printf("%s[%d,%d,%d]", first ? "" : ",", a, r.x, r.y);
        first = false;
    }
    // vector helpers on fixed float3 inputs
    const float3 A = make_float3(0.3f, -1.7f, 2.9f), B = make_float3(-4.1f, 0.25f, 1.0e-3f);
    printf("],\n\"dot\": [%u],\n", bits(dot(A, B)));
    float3 c = cross(A, B);
    printf("\"cross\": [%u,%u,%u],\n", bits(c.x), bits(c.y), bits(c.z));
    printf("\"length\": [%u],\n", bits(length(A)));
    float3 n = normalize(A);
    printf("\"normalize\": [%u,%u,%u],\n", bits(n.x), bits(n.y), bits(n.z));
    float3 fr = fracf(make_float3(-0.25f, 3.75f, -7.0f));
    float3 fl = floorf(make_float3(-0.25f, 3.75f, -7.0f));
    printf("\"fracf\": [%u,%u,%u],\n\"floorf\": [%u,%u,%u],\n", bits(fr.x), bits(fr.y), bits(fr.z), bits(fl.x), bits(fl.y), bits(fl.z));
    int3 ii = make_int3(make_float3(-0.25f, 3.75f, -7.9f));
    printf("\"make_int3_float3\": [%d,%d,%d],\n", ii.x, ii.y, ii.z);
    printf("\"min_float3\": [%u],\n\"max_uint3\": [%u],\n", bits(min(make_float3(4.8f, 2.0f, 3.0f))), max(make_uint3(256, 512, 128)));
    float3 sc = -1 * A;   // the `-1 * invR` form of cpp:685
    printf("\"neg1_times_float3\": [%u,%u,%u],\n", bits(sc.x), bits(sc.y), bits(sc.z));
    float3 q = make_float3(1.0f) / B;
    printf("\"one_over_float3\": [%u,%u,%u],\n", bits(q.x), bits(q.y), bits(q.z));
    // constants (constant_parameters.h:15-26)
    printf("\"constants\": {\"e_delta\": %u, \"radius\": %d, \"dist_threshold\": %u, \"normal_threshold\": %u, "
           "\"track_threshold\": %u, \"maxweight\": %u, \"nearPlane\": %u, \"farPlane\": %u, \"delta\": %u, "
           "\"light\": [%u,%u,%u], \"ambient\": [%u,%u,%u]},\n",
           bits(e_delta), radius, bits(dist_threshold), bits(normal_threshold), bits(track_threshold), bits(maxweight),
           bits(nearPlane), bits(farPlane), bits(delta), bits(light.x), bits(light.y), bits(light.z), bits(ambient.x),
           bits(ambient.y), bits(ambient.z));
    // gaussian taps exactly as cpp:100-108 computes them
    printf("\"gaussian\": [");
    for (unsigned i = 0; i < (unsigned)(radius * 2 + 1); i++) {
        int x = i - 2;
        printf("%s%u", i ? "," : "", bits(expf(-(x * x) / (2 * delta * delta))));
    }
    printf("],\n\"sizes\": {\"float3\": %zu, \"float4\": %zu, \"short2\": %zu, \"uchar4\": %zu}\n}\n", sizeof(float3),
           sizeof(float4), sizeof(short2), sizeof(uchar4));
    return 0;
}
