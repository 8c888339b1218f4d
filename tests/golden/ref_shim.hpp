// tests/golden/ref_shim.hpp -- fixture GENERATOR support (build container only).
// Stand-ins for the CUDA language/builtin surface the reference kernel bodies use, so that
// lines 22-671 of soft_rasterize_cuda_kernel.cu compile unchanged with g++ and run sequentially on
// the host.  This is generator tooling written for this repo; it contains no reference code.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdio>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __restrict__

struct shim_dim3 { int x = 0, y = 0, z = 0; };
static thread_local shim_dim3 blockIdx, blockDim, threadIdx;

// CUDA resolves min/max on mixed float/double arguments to the double overload.
inline float  min(float a, float b)   { return a < b ? a : b; }
inline float  max(float a, float b)   { return a > b ? a : b; }
inline double min(double a, double b) { return a < b ? a : b; }
inline double max(double a, double b) { return a > b ? a : b; }
inline double min(float a, double b)  { return min((double)a, b); }
inline double max(float a, double b)  { return max((double)a, b); }
inline double min(double a, float b)  { return min(a, (double)b); }
inline double max(double a, float b)  { return max(a, (double)b); }

// one host thread replays the grid in launch order, so a plain read-modify-write is the
// deterministic equivalent of atomicAdd
template <class T> inline T atomicAdd(T* p, T v) { T old = *p; *p += v; return old; }

// CUDA exposes float overloads of the libm names in the global namespace
inline float sqrt(float x) { return std::sqrt(x); }
inline float exp(float x)  { return std::exp(x); }
inline float pow(float x, int y)   { return std::pow(x, (float)y); }
inline float pow(float x, float y) { return std::pow(x, y); }
