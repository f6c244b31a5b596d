// Shared device/host helpers for libdcs_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
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
