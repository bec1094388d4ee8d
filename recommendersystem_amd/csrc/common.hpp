// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels.
// wave = 64 lanes everywhere; bf16 is the native __bf16 type (v_cvt_pk_bf16_f32).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "switches.hpp"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define LDS_AS __attribute__((address_space(3)))

namespace rsys {

// ---- error plumbing (C ABI returns int32 status; message is thread-local)
void set_error(const std::string& msg);
#define RSYS_OK 0
#define RSYS_ERR_ARG -1
#define RSYS_ERR_HIP -2
#define RSYS_ERR_STATE -3
#define RSYS_ERR_COMM -4

#define HIP_CHECK(expr)                                                                   \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      rsys::set_error(std::string(#expr) + ": " + hipGetErrorString(_e) + " at " +        \
                      __FILE__ + ":" + std::to_string(__LINE__));                         \
      return RSYS_ERR_HIP;                                                                \
    }                                                                                     \
  } while (0)

#define ARG_CHECK(cond, msg)                                                              \
  do {                                                                                    \
    if (!(cond)) {                                                                        \
      rsys::set_error(std::string("argument check failed: ") + #cond + " : " + (msg));    \
      return RSYS_ERR_ARG;                                                                \
    }                                                                                     \
  } while (0)

// ---- type helpers
template <typename T> struct is_bf16 { static constexpr bool value = false; };
template <> struct is_bf16<bf16> { static constexpr bool value = true; };

__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16 x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float x) { return (bf16)x; }

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// block-wide sum for blockDim.x <= 1024 (multiple of 64); every thread gets the result
__device__ __forceinline__ float block_sum(float v, float* smem /* >= 16 floats */) {
  v = wave_sum(v);
  int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane_id() == 0) smem[w] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += smem[i];
  return r;
}
__device__ __forceinline__ float block_max(float v, float* smem) {
  v = wave_max(v);
  int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane_id() == 0) smem[w] = v;
  __syncthreads();
  float r = smem[0];
  for (int i = 1; i < nw; ++i) r = fmaxf(r, smem[i]);
  return r;
}

// fp8 trunk (f8.hip): running amax of a tensor while its producer writes it.  An amax slot is 64 shards one cache line apart
// (F8_AMAX_SHARD floats; the reader takes the maximum over the shards); a workgroup adds to the shard of its block index and skips
// the atomic when the shard already holds a value at least as large (the value only grows, so a stale read costs an atomic, never
// a result).  m >= 0: the bit patterns of non-negative floats order like ints.
constexpr int F8_AMAX_SHARDS = 64, F8_AMAX_SHARD = 32;
__device__ __forceinline__ void f8_amax_note(float* slot, float m) {
  float* p = slot + (blockIdx.x & (F8_AMAX_SHARDS - 1)) * F8_AMAX_SHARD;
  if (m > __builtin_nontemporal_load(p)) atomicMax((int*)p, __float_as_int(m));
}
// the same without looking first: one atomic nothing waits for (a load-and-compare in front of it would make the wave wait for
// every vector-memory operation it still has in flight -- a GEMM epilogue's stores, a row kernel's last rows)
__device__ __forceinline__ void f8_amax_add(float* slot, float m) {
  atomicMax((int*)(slot + (blockIdx.x & (F8_AMAX_SHARDS - 1)) * F8_AMAX_SHARD), __float_as_int(m));
}
__device__ __forceinline__ float bf16_rounded(float v) { return (float)(bf16)v; }

// Philox4x32-10 counter RNG (Salmon et al. 2011), used for masks and random init.
struct Philox {
  uint32_t k0, k1;
  __host__ __device__ Philox(uint64_t seed) : k0((uint32_t)seed), k1((uint32_t)(seed >> 32)) {}
  __host__ __device__ static inline void mulhilo(uint32_t a, uint32_t b, uint32_t& hi, uint32_t& lo) {
    uint64_t p = (uint64_t)a * b;
    hi = (uint32_t)(p >> 32);
    lo = (uint32_t)p;
  }
  __host__ __device__ inline void gen(uint64_t ctr, uint32_t stream, uint32_t out[4]) const {
    uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = stream, c3 = 0;
    uint32_t a = k0, b = k1;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      uint32_t h0, l0, h1, l1;
      mulhilo(0xD2511F53u, c0, h0, l0);
      mulhilo(0xCD9E8D57u, c2, h1, l1);
      uint32_t n0 = h1 ^ c1 ^ a, n1 = l1, n2 = h0 ^ c3 ^ b, n3 = l0;
      c0 = n0; c1 = n1; c2 = n2; c3 = n3;
      a += 0x9E3779B9u; b += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
  }
};
__host__ __device__ inline float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

}  // namespace rsys
