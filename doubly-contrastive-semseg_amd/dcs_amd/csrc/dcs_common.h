// Shared device/host helpers for libdcs_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stdint.h>
#include "dcs_hip.h"

#define DCS_CHECK_ARG(cond) do { if (!(cond)) return DCS_E_ARG; } while (0)
#define DCS_LAUNCH_RET() do { return hipGetLastError() == hipSuccess ? DCS_OK : DCS_E_LAUNCH; } while (0)

static inline bool dcs_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline hipStream_t dcs_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Blocks b and b+8 share an XCD (and its L2).  Map the hardware block id to a logical tile id
// so that consecutive logical tiles (which share operand panels / halo rows) run on ONE XCD.
// Bijective for every grid size (cdna_hip_programming.md section 5, "XCD swizzle must be bijective").
__device__ __forceinline__ int dcs_xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// Streaming (non-temporal) 16-byte accesses for tensors far larger than L2 + MALL: a single-pass kernel gains 5-29 % of
// HBM bandwidth when its lines do not displace each other in the caches (tools/bn_nt_probe.py: BatchNorm-backward apply
// 4.5-4.8 -> 5.6-5.8 TB/s).  NT as a template flag or as a wave-uniform runtime flag.
typedef float dcs_f32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ float4 ldx4(const float* p) {
  if constexpr (NT) {
    const dcs_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const dcs_f32x4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
  } else {
    return *reinterpret_cast<const float4*>(p);
  }
}
template <bool NT>
__device__ __forceinline__ void stx4(float* p, const float4 v) {
  if constexpr (NT) {
    const dcs_f32x4 w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<dcs_f32x4*>(p));
  } else {
    *reinterpret_cast<float4*>(p) = v;
  }
}
__device__ __forceinline__ float4 ld4s(const float* p, const bool nt) { return nt ? ldx4<true>(p) : ldx4<false>(p); }
__device__ __forceinline__ void st4s(float* p, const float4 v, const bool nt) {
  if (nt) stx4<true>(p, v); else stx4<false>(p, v);
}
#include "dcs_config.h"
// does a tensor of this many bytes stream?
static inline bool dcs_streams(long long bytes) {
  const DcsConfig& c = dcs_config();
  return c.bn_nt != 0 && bytes >= ((long long)c.nt_min_mb << 20);
}

__device__ __forceinline__ float dcs_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double dcs_wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float dcs_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Bilinear source taps of output index o for F.interpolate(mode="bilinear", align_corners=False) with an explicit output
// size (torch's area_pixel_compute_source_index): scale = in / out.
struct Lin { int i0, i1; float w0, w1; };
__device__ __forceinline__ Lin lin_src(int o, float scale, int in) {
  float s = scale * ((float)o + 0.5f) - 0.5f;
  if (s < 0.f) s = 0.f;
  Lin l;
  l.i0 = (int)s;
  if (l.i0 > in - 1) l.i0 = in - 1;
  l.i1 = l.i0 + (l.i0 < in - 1 ? 1 : 0);
  l.w1 = s - (float)l.i0;
  l.w0 = 1.f - l.w1;
  return l;
}
