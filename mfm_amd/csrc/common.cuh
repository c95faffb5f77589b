// Shared device/host helpers for the MFM HIP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MFM_WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- wave-level reductions (64 lanes) -----------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}
// sum over the 16 lanes that share (lane >> 4)
__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__host__ __device__ __forceinline__ int ceil16(int x) { return (x + 15) & ~15; }
