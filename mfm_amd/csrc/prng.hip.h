// Counter-based PRNG on the device: Threefry-2x32 (20 rounds) with the key / counter conventions
// of jax.random (non-partitionable threefry, 64-bit draws), i.e. the conventions the reference's
// call sites rely on (exe_flow_matching.py:153-155,166,212,232,265,268,275,303; mala.py:93;
// bblackjax/util.py:80-82; proposal.py:179).  The CPU oracle (oracle/prng.py) restates the same
// conventions; a draw is a pure function of (key, sample index, draw size), so it does not depend on
// how chains are sharded over workgroups or GPUs.
#pragma once
#include "common.hip.h"

struct Key2 { uint32_t k0, k1; };

__host__ __device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

__host__ __device__ __forceinline__ void threefry2x32(Key2 key, uint32_t c0, uint32_t c1, uint32_t& o0, uint32_t& o1) {
  const uint32_t ks0 = key.k0, ks1 = key.k1, ks2 = key.k0 ^ key.k1 ^ 0x1BD11BDAu;
  uint32_t x0 = c0 + ks0, x1 = c1 + ks1;
#define TF_R(r) x0 += x1; x1 = rotl32(x1, r); x1 ^= x0;
  TF_R(13) TF_R(15) TF_R(26) TF_R(6)   x0 += ks1; x1 += ks2 + 1u;
  TF_R(17) TF_R(29) TF_R(16) TF_R(24)  x0 += ks2; x1 += ks0 + 2u;
  TF_R(13) TF_R(15) TF_R(26) TF_R(6)   x0 += ks0; x1 += ks1 + 3u;
  TF_R(17) TF_R(29) TF_R(16) TF_R(24)  x0 += ks1; x1 += ks2 + 4u;
  TF_R(13) TF_R(15) TF_R(26) TF_R(6)   x0 += ks2; x1 += ks0 + 5u;
#undef TF_R
  o0 = x0; o1 = x1;
}

// jax.random.split(key, num)[idx]: keys[j] = (out[2j], out[2j+1]), out = concat(y0, y1) of
// threefry(key, (i, i + num)), i < num.
__host__ __device__ __forceinline__ Key2 split_at(Key2 key, uint32_t num, uint32_t idx) {
  Key2 r;
  uint32_t w[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    uint32_t m = 2u * idx + q;
    bool lo = m < num;
    uint32_t c0 = lo ? m : m - num;
    uint32_t y0, y1;
    threefry2x32(key, c0, c0 + num, y0, y1);
    w[q] = lo ? y0 : y1;
  }
  r.k0 = w[0]; r.k1 = w[1];
  return r;
}

// 64 random bits of sample `idx` out of a draw of `size` samples.
__host__ __device__ __forceinline__ uint64_t random_bits64(Key2 key, uint32_t idx, uint32_t size) {
  uint32_t y0, y1;
  threefry2x32(key, idx, idx + size, y0, y1);
  return ((uint64_t)y0 << 32) | (uint64_t)y1;
}

__device__ __forceinline__ double bits_to_unit(uint64_t bits) {
  return __longlong_as_double((long long)((bits >> 12) | 0x3FF0000000000000ULL)) - 1.0;
}

// jax.random.uniform(key, shape) in [0, 1), float64
__device__ __forceinline__ double uniform01(Key2 key, uint32_t idx, uint32_t size) {
  return fmax(0.0, bits_to_unit(random_bits64(key, idx, size)));
}

// jax.random.normal: sqrt(2) * erfinv(uniform(nextafter(-1, 0), 1)); two separate roundings for
// u * (hi - lo) + lo exactly as numpy/XLA evaluate it (no FMA contraction).
__device__ __forceinline__ double normal64(Key2 key, uint32_t idx, uint32_t size) {
  const double lo = -0.99999999999999988897769753748;  // nextafter(-1, 0)
  const double span = 1.0 - lo;
  double u = bits_to_unit(random_bits64(key, idx, size));
  u = __dadd_rn(__dmul_rn(u, span), lo);
  u = fmax(lo, u);
#ifdef MFM_EXP_FAKE_NORMAL            // development probe: how much of a kernel is the float64 erfinv?
  return u;
#else
  return 1.4142135623730951 * erfinv(u);
#endif
}
