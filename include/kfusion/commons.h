// kfusion/commons.h — the plain types a SLAMBench driver sees through class Kfusion.
//
// Stand-in for the part of the reference's kfusion/include/commons.h (+ thirdparty/vector_types.h) that
// crosses the Kfusion boundary: uint2 / uint3 / float3 / float4 / uchar3 / uchar4, Matrix4 (commons.h:317-319)
// and a few make_* helpers.  No TooN, no CUDA headers: the pose algebra lives inside libkfusion_hip.so.
#ifndef KFUSION_HIP_COMMONS_H_
#define KFUSION_HIP_COMMONS_H_

// the standard headers the reference's commons.h:21-31 hands to whoever includes <kernels.h> (benchmark.cpp relies on
// <fstream>, <iostream>, <iomanip> and uint16_t arriving this way)
#include <limits.h>
#include <stdint.h>
#include <sys/stat.h>
#include <cassert>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <iterator>
#include <string>
#include <vector>

typedef unsigned int uint;
typedef unsigned short ushort;

struct uint2 { unsigned int x, y; };
struct uint3 { unsigned int x, y, z; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
struct uchar3 { unsigned char x, y, z; };
struct uchar4 { unsigned char x, y, z, w; };
struct Matrix4 { float4 data[4]; };   // row-major rows, 64 bytes, same layout as the reference

static inline uint2 make_uint2(unsigned int x, unsigned int y) { uint2 v = {x, y}; return v; }
static inline uint3 make_uint3(unsigned int x, unsigned int y, unsigned int z) { uint3 v = {x, y, z}; return v; }
static inline float3 make_float3(float x, float y, float z) { float3 v = {x, y, z}; return v; }
static inline float4 make_float4(float x, float y, float z, float w) { float4 v = {x, y, z, w}; return v; }
static inline float3 operator*(const float3 &a, const float3 &b) { return make_float3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline float4 operator/(const float4 &a, float b) { return make_float4(a.x / b, a.y / b, a.z / b, a.w / b); }   // cutil_math.h: float4 / float (an int ratio converts)

// commons.h:52-57
static inline bool is_file(const std::string &path) {
    struct stat buf;
    if (stat(path.c_str(), &buf) != 0) return false;
    return S_ISREG(buf.st_mode);
}

// commons.h:34 — the reference's commons.h pulls in the run-time parameters (struct Configuration); drivers get them through <kernels.h>
#include "default_parameters.h"

#endif
