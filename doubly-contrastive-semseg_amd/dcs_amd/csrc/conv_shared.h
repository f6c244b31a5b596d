// Pieces shared by the convolution kernels (conv_igemm.hip: exact fp32 MFMA; conv_split.hip: fp32 through three bf16
// pieces on the bf16 MFMA): guarded buffer loads, the BatchNorm + ReLU prologue, the epilogue, geometry checks.
#pragma once
#include "dcs_common.h"
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {


__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
// Guarded operand loads are buffer loads: a lane whose element is padding / out of range gets the byte offset
// OOB (>= num_records) and the hardware range check returns 0.  No branch, no select, 32-bit addressing, and
// hipcc keeps the loads in flight across the MFMA block (a conditional load makes it branch around every load and
// drain vmcnt per element: cdna_hip_programming.md, trap (c)).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* base, long long bytes) {
  const unsigned n = bytes > 0x7FFFFFFFll ? 0x7FFFFFFFu : (bytes < 0 ? 0u : (unsigned)bytes);
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, n, 0x00020000);
}
__device__ __forceinline__ float4 bld4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// BatchNorm + ReLU prologue on a staged float4: relu(v * sc + sh), or 0 where the element is padding (lim = 0; +inf on real
// elements): one fma and one v_med3_f32 (clamp to [0, lim]) per float instead of fma + max + select.  Scalar fmas on
// purpose: packed f32 VALU ops cost ~3x their issue slot beside MFMAs (MI355X_MICROARCH.md, filler prices).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 pro_apply(const float4 v, const float4 sc, const float4 sh, const float lim) {
  return make_float4(__builtin_amdgcn_fmed3f(fmaf(v.x, sc.x, sh.x), 0.f, lim), __builtin_amdgcn_fmed3f(fmaf(v.y, sc.y, sh.y), 0.f, lim),
                     __builtin_amdgcn_fmed3f(fmaf(v.z, sc.z, sh.z), 0.f, lim), __builtin_amdgcn_fmed3f(fmaf(v.w, sc.w, sh.w), 0.f, lim));
}

// BatchNorm-backward sums in a data-gradient epilogue (see conv_epilogue): y = the input of the BatchNorm whose
// output gradient this launch produces (same [M][dst_cstride] layout as dst), mask = the tensor whose sign is the ReLU
// mask (nullable), bn = [scale, shift, mean, invstd] x C record, relu = derive the mask from y*scale+shift.
// relu: 0 = ReLU mask from `mask > 0` (or none), 1 = from the recomputed BatchNorm output, 2 = `mask` points to BYTES, the four
// mask bits of every float4 of the tensor (dcs_bn_act's mask8: 1 byte read instead of 16); + 4: the destination receives the
// MASKED value gm instead of the raw gradient (everything behind a residual block's output passes its ReLU, so nobody needs
// the unmasked one -- and the BatchNorm backward that follows need not write gm again)
struct BnBwdEpi { const float* y; const float* mask; const float* bn; int relu; };

// Epilogue shared by the convolution kernels.  C/D layout of the 32x32 MFMA: col = lane&31,
// row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).  Each wave transposes its accumulators 32 rows at a time through a private
// LDS tile and writes 16 B per lane (a quarter wave covers one contiguous run of the pixel's channels) instead of
// 4-byte column-strided stores.  With `stats`, the per-channel sum and sum of squares of this block's outputs (the
// BatchNorm batch statistics of the layer that follows) are reduced in fixed order and written to
// stats[128-pixel row][2][Cout], so the activation is not read again by a separate reduction pass.
// With `bnb.y` (data-gradient launches), stats instead receives, per 128-pixel row, sum(gm) and sum(gm * xhat) with
// gm = (final dst value) * ReLU mask, xhat = (y - mean) * invstd: the two reductions of the BatchNorm backward that
// consumes this gradient, taken while the values are in registers (network/backbone/resnet_pyramid.py:28-36,:71-89
// backward; the separate reduction pass -- 2 of the 5 tensor passes of a BatchNorm backward -- disappears).
// smem: the block's LDS (free after the main loop's final barrier); rowoff[BM]: element offset of every tile row in
// dst, or -1; wave (wm, wn) owns rows wm*TM*32.. and columns wn*TN*32.. of the tile.
template <int BM, int BN, int TM, int TN, int WM, int SMEM_FLOATS>
__device__ __forceinline__ void conv_epilogue(f32x16 (&acc)[TM][TN], float* smem, const long long* rowoff,
                                              const float* __restrict__ bias, float* __restrict__ dst,
                                              const int dst_cstride, const int Cout, const int co0, const int accumulate,
                                              float* __restrict__ stats, const long long mtile,
                                              const unsigned long long M, const int wm, const int wn,
                                              const BnBwdEpi bnb = BnBwdEpi{nullptr, nullptr, nullptr, 0},
                                              const bool all_rows_valid = false) {
  constexpr int EPC = TN * 32;            // columns of a wave's sub-tile
  constexpr int EPL = EPC + 4;            // staging row stride (16-B aligned rows, conflict-free column writes)
  constexpr int EPV = EPC / 4;            // float4 per staged row
  constexpr int EPR = 64 / EPV;           // rows per read-back pass
  static_assert(4 * 32 * EPL + WM * BN * 2 <= SMEM_FLOATS, "staging does not fit");
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, h = lane >> 5;
  float* stg = smem + wid * (32 * EPL);
  float* st = smem + 4 * 32 * EPL;        // [WM][BN][2] statistics scratch
  const bool vec_ok = (dst_cstride & 3) == 0 && (reinterpret_cast<unsigned long long>(dst) & 15ull) == 0;
  // accumulate: bit 0 = add to the destination, bit 1 = the destination (and the BatchNorm-backward operands beside it)
  // is far larger than the caches: streaming accesses (dcs_common.h, ld4s / st4s)
  const bool acc_dst = (accumulate & 1) != 0, nt = (accumulate & 2) != 0;
  float ssum[TN], ssq[TN];
#pragma unroll
  for (int b = 0; b < TN; ++b) { ssum[b] = 0.f; ssq[b] = 0.f; }
  // BatchNorm-backward sums: this lane's four channels colv .. colv+3 over its rows
  float4 b_s0 = zero4(), b_s1 = zero4(), b_sc = zero4(), b_sh = zero4(), b_mu = zero4(), b_is = zero4();
  const bool do_bnb = bnb.y != nullptr;
  {
    const int colv0 = co0 + wn * EPC + (lane % EPV) * 4;
    if (do_bnb && colv0 + 3 < Cout) {
      b_sc = ld4(bnb.bn + colv0); b_sh = ld4(bnb.bn + Cout + colv0);
      b_mu = ld4(bnb.bn + 2 * Cout + colv0); b_is = ld4(bnb.bn + 3 * Cout + colv0);
    }
  }
#pragma unroll
  for (int a = 0; a < TM; ++a) {
    const int row0 = wm * TM * 32 + a * 32;
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int col = co0 + wn * EPC + b * 32 + l31;
      const float bvv = (bias && col < Cout) ? bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rl = (r & 3) + 8 * (r >> 2) + 4 * h;
        const float v = acc[a][b][r] + bvv;
        stg[rl * EPL + b * 32 + l31] = v;
        // all_rows_valid (block-uniform: the tile lies inside the tensor) skips 16 LDS look-ups per 32 x 32 sub-tile
        if (stats && !do_bnb && (all_rows_valid || rowoff[row0 + rl] >= 0)) { ssum[b] += v; ssq[b] = fmaf(v, v, ssq[b]); }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int c4 = lane % EPV, rr = lane / EPV;
    const int colv = co0 + wn * EPC + c4 * 4;
    // Everything the read-back passes LOAD from global memory (old destination values, the BatchNorm-backward operands)
    // is requested up front: the stores of pass p and the loads of pass p + 1 go through pointers the compiler cannot
    // tell apart, so inside the loop every load would wait behind the previous store -- eight (sixteen) serial HBM
    // latencies per 32-row sub-tile.
    constexpr int NP = 32 / EPR;
#ifdef DCS_EPI_NO_PREFETCH
    constexpr bool PRE = false;               // A/B switch (compile time)
#else
    constexpr bool PRE = NP <= 8;
#endif
    float4 pre_o[PRE ? NP : 1], pre_y[PRE ? NP : 1], pre_m[PRE ? NP : 1];
#if defined(DCS_SLP_EXP) && DCS_SLP_EXP == 2
#pragma unroll
    for (int p = 0; p < (PRE ? NP : 1); ++p) { pre_o[p] = zero4(); pre_y[p] = zero4(); pre_m[p] = zero4(); }
#endif
    const bool vrow = vec_ok && colv + 3 < Cout;
    if (PRE && vrow && (acc_dst || do_bnb)) {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const long long ro = rowoff[row0 + p * EPR + rr];
        if (ro < 0) continue;
        if (acc_dst) pre_o[p] = ld4s(dst + ro + colv, nt);
        if (do_bnb) {
          pre_y[p] = ld4s(bnb.y + ro + colv, nt);
          if (bnb.mask) {
            if ((bnb.relu & 3) == 2) pre_m[p].x = __uint_as_float((unsigned)reinterpret_cast<const unsigned char*>(bnb.mask)[(ro + colv) >> 2]);
            else pre_m[p] = ld4s(bnb.mask + ro + colv, nt);
          }
        }
      }
    }
#pragma unroll
    for (int p = 0; p < 32 / EPR; ++p) {
      const int rl = p * EPR + rr;
      const long long ro = rowoff[row0 + rl];
      if (ro < 0 || colv >= Cout) continue;
      float4 v = ld4(&stg[rl * EPL + c4 * 4]);
      float* q = dst + ro + colv;
      if (vec_ok && colv + 3 < Cout) {
        if (acc_dst) { const float4 o = PRE ? pre_o[PRE ? p : 0] : ld4s(q, nt); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
        const bool store_masked = do_bnb && (bnb.relu & 4) != 0;
        if (!store_masked) st4s(q, v, nt);
        if (do_bnb) {
          const float4 yy = PRE ? pre_y[PRE ? p : 0] : ld4s(bnb.y + ro + colv, nt);
          float4 gm = v;
          if (bnb.mask && (bnb.relu & 3) == 2) {
            const unsigned m = PRE ? __float_as_uint(pre_m[PRE ? p : 0].x)
                                   : (unsigned)reinterpret_cast<const unsigned char*>(bnb.mask)[(ro + colv) >> 2];
            gm.x = (m & 1u) ? gm.x : 0.f; gm.y = (m & 2u) ? gm.y : 0.f; gm.z = (m & 4u) ? gm.z : 0.f; gm.w = (m & 8u) ? gm.w : 0.f;
          } else if (bnb.mask) {
            const float4 ms = PRE ? pre_m[PRE ? p : 0] : ld4s(bnb.mask + ro + colv, nt);
            gm.x = ms.x > 0.f ? gm.x : 0.f; gm.y = ms.y > 0.f ? gm.y : 0.f; gm.z = ms.z > 0.f ? gm.z : 0.f; gm.w = ms.w > 0.f ? gm.w : 0.f;
          } else if (bnb.relu & 3) {
            gm.x = fmaf(yy.x, b_sc.x, b_sh.x) > 0.f ? gm.x : 0.f; gm.y = fmaf(yy.y, b_sc.y, b_sh.y) > 0.f ? gm.y : 0.f;
            gm.z = fmaf(yy.z, b_sc.z, b_sh.z) > 0.f ? gm.z : 0.f; gm.w = fmaf(yy.w, b_sc.w, b_sh.w) > 0.f ? gm.w : 0.f;
          }
          if (store_masked) st4s(q, gm, nt);
          b_s0.x += gm.x; b_s0.y += gm.y; b_s0.z += gm.z; b_s0.w += gm.w;
          b_s1.x = fmaf(gm.x, (yy.x - b_mu.x) * b_is.x, b_s1.x); b_s1.y = fmaf(gm.y, (yy.y - b_mu.y) * b_is.y, b_s1.y);
          b_s1.z = fmaf(gm.z, (yy.z - b_mu.z) * b_is.z, b_s1.z); b_s1.w = fmaf(gm.w, (yy.w - b_mu.w) * b_is.w, b_s1.w);
#if defined(DCS_SLP_EXP) && DCS_SLP_EXP == 1
          asm volatile("" : "+v"(b_s0.x), "+v"(b_s0.y), "+v"(b_s0.z), "+v"(b_s0.w));
          asm volatile("" : "+v"(b_s1.x), "+v"(b_s1.y), "+v"(b_s1.z), "+v"(b_s1.w));
#endif
        }
      } else {
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (colv + e < Cout) q[e] = acc_dst ? q[e] + vv[e] : vv[e];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  if (stats) {
    if (do_bnb) {
      // lanes with the same channel group (lane % EPV) hold different rows: fixed-order butterfly over lane / EPV
      float vals[8] = {b_s0.x, b_s0.y, b_s0.z, b_s0.w, b_s1.x, b_s1.y, b_s1.z, b_s1.w};
#pragma unroll
      for (int o = EPV; o < 64; o <<= 1)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          vals[e] += __shfl_xor(vals[e], o, 64);
#if defined(DCS_SLP_EXP) && DCS_SLP_EXP == 3
          asm volatile("" : "+v"(vals[e]));
#endif
        }
      if (lane < EPV) {
        const int cl = wn * EPC + lane * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) { st[(wm * BN + cl + e) * 2] = vals[e]; st[(wm * BN + cl + e) * 2 + 1] = vals[4 + e]; }
      }
    } else {
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        const int cl = wn * EPC + b * 32 + l31;
        float s0 = ssum[b], s1 = ssq[b];
        s0 += __shfl_xor(s0, 32, 64);
        s1 += __shfl_xor(s1, 32, 64);
        if (h == 0) { st[(wm * BN + cl) * 2] = s0; st[(wm * BN + cl) * 2 + 1] = s1; }
      }
    }
    __syncthreads();
    // one statistics row per 128 output pixels, whatever BM is (the host sizes the buffer for 128-pixel rows)
    constexpr int HALVES = BM / 128, WPH = WM / HALVES;
    static_assert(WPH >= 1, "a wave must not straddle two 128-pixel statistics rows");
    if (tid < BN && co0 + tid < Cout) {
#pragma unroll
      for (int hh = 0; hh < HALVES; ++hh) {
        const long long srow = mtile * HALVES + hh;
        if ((unsigned long long)srow * 128ull >= M) break;
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int w = hh * WPH; w < (hh + 1) * WPH; ++w) { s0 += st[(w * BN + tid) * 2]; s1 += st[(w * BN + tid) * 2 + 1]; }
        stats[(srow * 2) * Cout + co0 + tid] = s0;
        stats[(srow * 2 + 1) * Cout + co0 + tid] = s1;
      }
    }
  }
}

inline int check_geom(const DcsConvGeom* g) {
  DCS_CHECK_ARG(g != nullptr);
  DCS_CHECK_ARG(g->N > 0 && g->SH > 0 && g->SW > 0 && g->DH > 0 && g->DW > 0 && g->TY > 0 && g->TX > 0);
  DCS_CHECK_ARG(g->ntaps > 0 && g->ntaps <= DCS_MAX_TAPS && g->Cout > 0 && g->K > 0);
  DCS_CHECK_ARG((g->wstride & 3) == 0 && (g->src_cstride & 3) == 0);
  if (g->stem) {
    DCS_CHECK_ARG(g->K == 4 && g->src_cstride == 4);
  } else {
    DCS_CHECK_ARG((g->K & 3) == 0);
  }
  for (int t = 0; t < g->ntaps; ++t) DCS_CHECK_ARG((g->wofs[t] & 3) == 0 && g->wofs[t] >= 0);
  // destination sub-grid must stay inside the destination tensor
  DCS_CHECK_ARG((g->TY - 1) * g->dsy + g->dy0 < g->DH && (g->TX - 1) * g->dsx + g->dx0 < g->DW);
  DCS_CHECK_ARG(g->dy0 >= 0 && g->dx0 >= 0 && g->dsy > 0 && g->dsx > 0 && g->sy > 0 && g->sx > 0);
  return DCS_OK;
}

}  // namespace
