// TEST INFRASTRUCTURE.  Minimal host stand-in for <cuda_runtime.h> so that the REFERENCE's own kernel headers
// (/root/reference/kernels/permuto_sdf/*.cuh, compiled in place, never copied) build with g++ and run one
// "thread" at a time.  Only what those headers use is provided.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <math.h>

#define __host__
#define __device__
#define __global__
#define __forceinline__ inline
#define __restrict__

struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int3 { int x, y, z; };
struct int4 { int x, y, z, w; };
struct uint2 { unsigned int x, y; };
struct uint3 { unsigned int x, y, z; };
struct uint4 { unsigned int x, y, z, w; };
struct dim3 { unsigned int x = 1, y = 1, z = 1; };

static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float3 make_float3(float x, float y, float z) { return float3{x, y, z}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline int3 make_int3(int x, int y, int z) { return int3{x, y, z}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline uint3 make_uint3(unsigned x, unsigned y, unsigned z) { return uint3{x, y, z}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }

// the "current thread": the driver sets blockIdx.x = item index, blockDim.x = 1, threadIdx.x = 0
extern thread_local uint3 threadIdx;
extern thread_local uint3 blockIdx;
extern thread_local dim3 blockDim;

// float overloads that device code gets from the CUDA headers (declared before helper_math.h's int versions)
static inline float max(float a, float b) { return fmaxf(a, b); }
static inline float min(float a, float b) { return fminf(a, b); }
static inline double max(double a, double b) { return fmax(a, b); }
static inline double min(double a, double b) { return fmin(a, b); }
static inline float max(float a, double b) { return fmaxf(a, (float)b); }
static inline float min(float a, double b) { return fminf(a, (float)b); }
static inline float max(double a, float b) { return fmaxf((float)a, b); }
static inline float min(double a, float b) { return fminf((float)a, b); }
#define __expf(x) expf(x)
static inline unsigned int max(unsigned int a, unsigned int b) { return a > b ? a : b; }
static inline unsigned int min(unsigned int a, unsigned int b) { return a < b ? a : b; }

static inline int atomicAdd(int* addr, int v) {
  int old = *addr;
  *addr = old + v;
  return old;
}
