// Fused embedding-similarity / InfoNCE loss: S = X X^T is NEVER written to memory.
//
// Replaces the chain behind utils/loss.py:339-389 (PixelContrastLoss._contrastive, mode 0) and utils/loss.py:175-204
// (SupConLoss, mode 1) of the reference: matmul -> max -> subtract -> F.normalize -> masked exp / log reductions ->
// mean, and its autograd backward.  For anchors X [A,C] with float labels y [A] (y < 0 marks a padding row of the
// fixed-shape data-parallel all-gather: it takes part in nothing) the kernels produce
//     loss = (1/A_v) sum_i loss_i          and          dX = (G + G^T) X,   G_ij = d loss / d S_ij,
// where, for a valid row i over the valid columns j (it = 1/T):
//     m_i = max_j S_ij it,  u_ij = S_ij it - m_i,  L_ij = u_ij / max(||u_i||_2, 1e-12),  E_ij = exp(L_ij)
//     mode 0:  den_i = sum_{y_j != y_i} E_ij                loss_i = -1/cnt_i sum_{p in pos(i)} (L_ip - log(E_ip + den_i))
//     mode 1:  den_i = sum_{j != i} E_ij                    loss_i = -1/cnt_i sum_j w_ij (L_ij - log den_i)
//     pos(i) = {j != i : y_j == y_i},  w_ij = [j in pos(i)]  (or an explicit [b,b] mask tiled over the views, mode 1).
//
// Two kernel families, picked by A:
//   * A <= DCS_CONTRAST_SMALL_MAX (the per-rank anchor set, <= 608 rows at C3): TWO launches.
//       stats: one block per 16-row strip keeps its S strip [16][A] in LDS (MFMA 16x16x4, every wave a different
//              column tile), so the four dependent row sweeps (max; norm; denominators; positives) read LDS.
//       final: one block per 16-row strip recomputes its S tiles, forms G_ij + G_ji in registers from the row
//              records of BOTH rows (S is symmetric, so G_ji needs no second GEMM), and accumulates
//              dX_I += Gsym(I,J) X_J on the matrix cores; block 0 also reduces the loss.
//   * larger A (the all-gathered global set of the data-parallel step, 4864 rows at C4): see the tile kernels below --
//       three SYMMETRIC statistics sweeps over the upper-triangular 64x64 tiles (each tile feeds the statistics of its
//       rows AND, transposed, of its columns: half the similarity FLOPs per sweep), then the final sweep.
// Mode 1 with A <= small max accumulates S in float64 on the f64 matrix cores (v_mfma_f64_16x16x4_f64): pooled image
// embeddings of one batch are nearly parallel, S_ij it - m_i then cancels 4-5 digits, and a k-ordered fp32 FMA chain
// (what the f32 MFMA is) loses them where the reference's blocked CPU GEMM does not (measured: 7x the reference's own
// fp32-vs-fp64 error on the projection-head gradients; with f64 accumulation 0.1x).  2B <= 512 rows: free.
#include "dcs_common.h"
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int REC = 8;   // floats per row record: m (float, or double in slots 0-1), [2] rn, [3] den, [4] 1/cnt, [5] qv, [6] dot, [7] clamp

template <typename ACC> struct Mfma16;
template <> struct Mfma16<float> {
  typedef f32x4 acc_t;
  static __device__ __forceinline__ acc_t zero() { return acc_t{0.f, 0.f, 0.f, 0.f}; }
  static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int row(int lane, int r) { return 4 * (lane >> 4) + r; }      // C/D: col = lane&15
};
template <> struct Mfma16<double> {
  typedef f64x4 acc_t;
  static __device__ __forceinline__ acc_t zero() { return acc_t{0.0, 0.0, 0.0, 0.0}; }
  static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64((double)a, (double)b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) + 4 * r; }      // f64 C/D map differs
};

__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }
// exp of a normalised logit L in [-1, 0]: v_exp_f32 (1 ulp) of L * log2(e); the argument rounding adds <= 9e-8 relative.
// The epilogues evaluate two of these per similarity element, the library expf (range reduction, denormal handling) is
// ~4x the instructions.
__device__ __forceinline__ float exp_unit(float L) { return __expf(L); }
__device__ __forceinline__ float rcp_fast(float x) { return __frcp_rn(x); }
__device__ __forceinline__ float comp(const float4& v, int c) { return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w)); }

// One 16x16 tile of S = X_I X_J^T.  Lane (r = lane&15, q = lane>>4) holds, per 16-channel step s, the float4
// X[row r][16 s + 4 q .. +3] of BOTH operands; component c of step s is k = 16 s + 4 q + c on both sides, so every MFMA
// (4 k values: q = 0..3) multiplies matching channels whatever the order (any k permutation is legal when A and B agree).
// xa: the strip's rows (kept in registers when C <= 128), rowJ: this lane's row of the column tile (or -1).
template <typename ACC, int NS>
__device__ __forceinline__ typename Mfma16<ACC>::acc_t s_tile(const float4 (&xa)[NS], const float* __restrict__ X, const int ldx,
                                                             const int C, const int rowI, const int rowJ, const int q) {
  typename Mfma16<ACC>::acc_t acc = Mfma16<ACC>::zero();
  const int nchunk = (C + 16 * NS - 1) / (16 * NS);
  for (int ch = 0; ch < nchunk; ++ch) {
    float4 xb[NS], xs[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int k = (ch * NS + s) * 16 + 4 * q;
      xb[s] = (rowJ >= 0 && k < C) ? ldg4(X + (long long)rowJ * ldx + k) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (nchunk > 1) xs[s] = (rowI >= 0 && k < C) ? ldg4(X + (long long)rowI * ldx + k) : make_float4(0.f, 0.f, 0.f, 0.f);
      else xs[s] = xa[s];
    }
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc = Mfma16<ACC>::mma(comp(xs[s], c), comp(xb[s], c), acc);
  }
  return acc;
}

// C <= 128: both operands already in registers
template <typename ACC>
__device__ __forceinline__ typename Mfma16<ACC>::acc_t s_tile_regs(const float4 (&xa)[8], const float4 (&xb)[8]) {
  typename Mfma16<ACC>::acc_t acc = Mfma16<ACC>::zero();
#pragma unroll
  for (int s = 0; s < 8; ++s)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc = Mfma16<ACC>::mma(comp(xa[s], c), comp(xb[s], c), acc);
  return acc;
}
__device__ __forceinline__ void load_rows8(float4 (&x)[8], const float* __restrict__ X, const int ldx, const int C, const int row,
                                           const int q) {
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const int k = 16 * s + 4 * q;
    x[s] = (row >= 0 && k < C) ? ldg4(X + (long long)row * ldx + k) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

template <int W>
__device__ __forceinline__ float grp_sum(float v) {
#pragma unroll
  for (int o = W / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <int W, typename T>
__device__ __forceinline__ T grp_max(T v) {
#pragma unroll
  for (int o = W / 2; o > 0; o >>= 1) { const T w = __shfl_xor(v, o, 64); v = w > v ? w : v; }
  return v;
}

template <typename ACC> __device__ __forceinline__ void rec_put_m(float* r, ACC m);
template <> __device__ __forceinline__ void rec_put_m<float>(float* r, float m) { r[0] = m; r[1] = 0.f; }
template <> __device__ __forceinline__ void rec_put_m<double>(float* r, double m) { *reinterpret_cast<double*>(r) = m; }
template <typename ACC> __device__ __forceinline__ ACC rec_get_m(const float* r);
template <> __device__ __forceinline__ float rec_get_m<float>(const float* r) { return r[0]; }
template <> __device__ __forceinline__ double rec_get_m<double>(const float* r) { return *reinterpret_cast<const double*>(r); }

// weight of column j as a positive of row i (self excluded by the caller)
__device__ __forceinline__ float pos_weight(const float* __restrict__ mask, const int mb, const int i, const int j,
                                            const float yi, const float yj) {
  return mask ? mask[(i % mb) * mb + (j % mb)] : (yi == yj ? 1.f : 0.f);
}

// ------------------------------------------------------------------------------------------------------------------
// small-A statistics: block = one 16-row strip, S strip resident in LDS.
template <typename ACC, int NW>
__global__ __launch_bounds__(NW * 64)
void contrast_small_stats_kernel(const float* __restrict__ X, const int ldx, const float* __restrict__ y, const int ldy,
                                 const float* __restrict__ mask, const int mb, const int A, const int C, const int mode,
                                 const float it, float* __restrict__ rec, float* __restrict__ loss_row, const int AP) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  ACC* Ss = reinterpret_cast<ACC*>(smem_raw);                       // [16][AP]
  float* ys = reinterpret_cast<float*>(Ss + 16 * AP);               // [AP]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, li = lane & 15, q = lane >> 4;
  const int i0 = blockIdx.x * 16;
  for (int j = tid; j < AP; j += NW * 64) ys[j] = j < A ? y[(long long)j * ldy] : -1.f;
  const int rowI = i0 + li < A ? i0 + li : -1;
  float4 xa[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const int k = 16 * s + 4 * q;
    xa[s] = (rowI >= 0 && k < C) ? ldg4(X + (long long)rowI * ldx + k) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int ntile = (A + 15) >> 4;
  if (C <= 128) {
    // software prefetch: the next column tile's rows are in flight (L2) while this tile's 32 MFMAs issue
    float4 xn[8];
    if (wid < ntile) load_rows8(xn, X, ldx, C, 16 * wid + li < A ? 16 * wid + li : -1, q);
    for (int jt = wid; jt < ntile; jt += NW) {
      float4 xb[8];
#pragma unroll
      for (int s = 0; s < 8; ++s) xb[s] = xn[s];
      if (jt + NW < ntile) load_rows8(xn, X, ldx, C, 16 * (jt + NW) + li < A ? 16 * (jt + NW) + li : -1, q);
      const typename Mfma16<ACC>::acc_t acc = s_tile_regs<ACC>(xa, xb);
#pragma unroll
      for (int r = 0; r < 4; ++r) Ss[Mfma16<ACC>::row(lane, r) * AP + 16 * jt + li] = acc[r];
    }
  } else {
    for (int jt = wid; jt < ntile; jt += NW) {
      const int rowJ = 16 * jt + li < A ? 16 * jt + li : -1;
      const typename Mfma16<ACC>::acc_t acc = s_tile<ACC, 8>(xa, X, ldx, C, rowI, rowJ, q);
#pragma unroll
      for (int r = 0; r < 4; ++r) Ss[Mfma16<ACC>::row(lane, r) * AP + 16 * jt + li] = acc[r];
    }
  }
  __syncthreads();

  constexpr int TPR = NW * 4;                                       // threads per row (a lane group inside one wave)
  const int ri = tid / TPR, rl = tid % TPR;
  const int i = i0 + ri;
  const bool live = i < A && ys[i < AP ? i : 0] >= 0.f;
  const float yi = live ? ys[i] : -2.f;
  const ACC* srow = Ss + ri * AP;
  const ACC itA = (ACC)it;
  // sweep 1: row max of S/T over the valid columns (utils/loss.py:363, :179)
  ACC m = (ACC)(-3.0e38f);
  for (int j = rl; j < A; j += TPR)
    if (ys[j] >= 0.f) { const ACC v = srow[j] * itA; m = v > m ? v : m; }
  m = grp_max<TPR, ACC>(m);
  // sweep 2: ||u_i||_2, F.normalize eps 1e-12 (:366, :194)
  float n2 = 0.f;
  for (int j = rl; j < A; j += TPR)
    if (ys[j] >= 0.f) { const float u = (float)(srow[j] * itA - m); n2 = fmaf(u, u, n2); }
  n2 = grp_sum<TPR>(n2);
  const float nraw = sqrtf(n2);
  const float rn = 1.f / fmaxf(nraw, 1e-12f);
  // sweep 3: denominators, positive counts, and the E.L sums the gradient's <dL, L> needs
  float den = 0.f, cnt = 0.f, sEL = 0.f, swL = 0.f;
  for (int j = rl; j < A; j += TPR) {
    const float yj = ys[j];
    if (yj < 0.f) continue;
    const float L = (float)(srow[j] * itA - m) * rn;
    const float E = exp_unit(L);
    if (mode == 0) {
      if (yj != yi) { den += E; sEL = fmaf(E, L, sEL); }               // neg_logits (:376-377)
      else if (j != i) cnt += 1.f;
    } else {
      if (j != i) {                                                    // exp_logits * logits_mask (:196)
        den += E; sEL = fmaf(E, L, sEL);
        const float w = pos_weight(mask, mb, i, j, yi, yj);
        cnt += w; swL = fmaf(w, L, swL);
      }
    }
  }
  den = grp_sum<TPR>(den); cnt = grp_sum<TPR>(cnt); sEL = grp_sum<TPR>(sEL); swL = grp_sum<TPR>(swL);
  float lp, qv = 0.f, dot;
  const float icnt = 1.f / cnt;                                        // cnt == 0 -> inf -> NaN loss like the reference
  if (mode == 0) {
    // sweep 4 (positives only): log-probabilities and q = sum_pos 1/(E + den)
    float slp = 0.f, sq = 0.f, sdl = 0.f;
    for (int j = rl; j < A; j += TPR) {
      if (j == i || ys[j] != yi) continue;
      const float L = (float)(srow[j] * itA - m) * rn;
      const float d = exp_unit(L) + den;
      const float id = rcp_fast(d);
      slp += L - logf(d); sq += id; sdl = fmaf(den * id, L, sdl);
    }
    lp = grp_sum<TPR>(slp); qv = grp_sum<TPR>(sq); sdl = grp_sum<TPR>(sdl);
    dot = (qv * sEL - sdl) * icnt;
  } else {
    lp = swL - cnt * logf(den);
    dot = sEL / den - swL * icnt;
  }
  if (rl == 0 && i < A) {
    float* r = rec + (long long)i * REC;
    if (live) {
      rec_put_m<ACC>(r, m);
      r[2] = rn; r[3] = den; r[4] = icnt; r[5] = qv; r[6] = dot; r[7] = nraw <= 1e-12f ? 1.f : 0.f;
      loss_row[i] = -lp * icnt;                                        // temperature / base_temperature = 1
    } else {
#pragma unroll
      for (int e = 0; e < REC; ++e) r[e] = 0.f;
      loss_row[i] = 0.f;
    }
  }
}

// d loss / d S_ab (before the 1/A_v mean) from the record of row a; same = (y_a == y_b), self = (a == b), w = positive
// weight of b for a.  ACC-precision shift like the statistics sweeps.
template <typename ACC>
__device__ __forceinline__ float g_entry(const ACC s, const float* __restrict__ ra, const ACC itA, const float it,
                                         const int mode, const bool same, const bool self, const float w) {
  const float rn = ra[2], den = ra[3], icnt = ra[4], qv = ra[5], dot = ra[6];
  const float L = (float)(s * itA - rec_get_m<ACC>(ra)) * rn;
  const float E = exp_unit(L);
  float dL;
  if (mode == 0) dL = same ? (self ? 0.f : -(den * rcp_fast(E + den)) * icnt) : E * qv * icnt;
  else dL = self ? 0.f : (E * rcp_fast(den) - w * icnt);
  const float du = ra[7] != 0.f ? dL : dL - L * dot;                  // through F.normalize (clamped norm: no projection)
  return du * rn * it;
}

// small-A final sweep: dX_I = sum_J (G + G^T)(I,J) X_J, one block per 16-row strip, waves split the column tiles.
// gsym_out != null (C > 128: the feature width of DeepLab's pixel contrast): G + G^T is written out [A][ldg] instead
// and the caller finishes dX with one GEMM.
template <typename ACC, int NW>
__global__ __launch_bounds__(NW * 64)
void contrast_small_final_kernel(const float* __restrict__ X, const int ldx, const float* __restrict__ y, const int ldy,
                                 const float* __restrict__ mask, const int mb, const int A, const int C, const int mode,
                                 const float it, const float* __restrict__ rec, const float* __restrict__ loss_row,
                                 float* __restrict__ loss, float* __restrict__ dX, const int lddx,
                                 float* __restrict__ gsym_out, const int ldg) {
  constexpr int XLD = 144;                       // 128 + 16: rows 16 banks apart -> conflict-free k-strided B reads
  constexpr int GLD = 17;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* sm = reinterpret_cast<float*>(smem_raw);
  float* recI = sm;                              // [16][REC]
  float* yI = recI + 16 * REC;                   // [16]
  float* red = yI + 16;                          // [NW] block reduction scratch
  float* Gt = red + NW + (4 - (NW & 3)) % 4;     // per wave [16][GLD]
  float* Xt = Gt + NW * 16 * GLD + (4 - ((NW * 16 * GLD) & 3)) % 4;   // per wave [16][XLD]; later the cross-wave reduce buffer
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, li = lane & 15, q = lane >> 4;
  const int i0 = blockIdx.x * 16;
  // number of valid rows (the mean's denominator): every block counts the labels itself (A floats from L2)
  float nv = 0.f;
  for (int j = tid; j < A; j += NW * 64) nv += y[(long long)j * ldy] >= 0.f ? 1.f : 0.f;
  nv = dcs_wave_sum(nv);
  if (lane == 0) red[wid] = nv;
  if (tid < 16 * REC) recI[tid] = (i0 + tid / REC < A) ? rec[(long long)i0 * REC + tid] : 0.f;
  if (tid < 16) yI[tid] = i0 + tid < A ? y[(long long)(i0 + tid) * ldy] : -1.f;
  __syncthreads();
  float av = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) av += red[w];
  const float inv_av = 1.f / av;
  const int rowI = i0 + li < A ? i0 + li : -1;
  float4 xa[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const int k = 16 * s + 4 * q;
    xa[s] = (rowI >= 0 && k < C) ? ldg4(X + (long long)rowI * ldx + k) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  f32x4 dacc[8];
#pragma unroll
  for (int ct = 0; ct < 8; ++ct) dacc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
  float* gt = Gt + wid * 16 * GLD;
  float* xt = Xt + wid * 16 * XLD;
  const ACC itA = (ACC)it;
  const int ntile = (A + 15) >> 4;
  const bool regs = C <= 128;
  // software prefetch of the next column tile: its rows, record and label are in flight while this tile computes
  float4 xn[8], rna, rnb;
  float yn = -1.f;
  auto prefetch = [&](int jt) {
    const int j = 16 * jt + li;
    const bool ok = jt < ntile && j < A;
    if (regs) load_rows8(xn, X, ldx, C, ok ? j : -1, q);
    rna = ok ? ldg4(rec + (long long)j * REC) : make_float4(0.f, 0.f, 0.f, 0.f);
    rnb = ok ? ldg4(rec + (long long)j * REC + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    yn = ok ? y[(long long)j * ldy] : -1.f;
  };
  prefetch(wid);
  for (int jt = wid; jt < ntile; jt += NW) {
    const int j = 16 * jt + li;
    const int rowJ = j < A ? j : -1;
    float4 xb[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) xb[s] = xn[s];
    const float rj[REC] = {rna.x, rna.y, rna.z, rna.w, rnb.x, rnb.y, rnb.z, rnb.w};
    const float yj = yn;
    prefetch(jt + NW);
    const typename Mfma16<ACC>::acc_t acc = regs ? s_tile_regs<ACC>(xa, xb) : s_tile<ACC, 8>(xa, X, ldx, C, rowI, rowJ, q);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int il = Mfma16<ACC>::row(lane, r);
      const int i = i0 + il;
      const float yi = yI[il];
      float gs = 0.f;
      if (yi >= 0.f && yj >= 0.f) {
        const bool same = yi == yj, self = i == j;
        const float wij = (mode == 1 && !self) ? pos_weight(mask, mb, i, j, yi, yj) : 0.f;
        const float wji = (mode == 1 && !self) ? pos_weight(mask, mb, j, i, yj, yi) : 0.f;
        gs = (g_entry<ACC>(acc[r], recI + il * REC, itA, it, mode, same, self, wij) +
              g_entry<ACC>(acc[r], rj, itA, it, mode, same, self, wji)) * inv_av;
      }
      if (gsym_out) { if (i < A && j < A) gsym_out[(long long)i * ldg + j] = gs; }
      else gt[il * GLD + li] = gs;
    }
    if (gsym_out) continue;
    // X_J tile -> per-wave LDS image [16 j][XLD] (the k-strided B operand of the second product): the very registers
    // that fed the S tile
#pragma unroll
    for (int s = 0; s < 8; ++s) *reinterpret_cast<float4*>(&xt[li * XLD + 16 * s + 4 * q]) = xb[s];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // dX_I[16][C] += Gsym[16 i][16 j] X_J[16 j][C]: A operand Gt[i = li][j = 4 kk + q], B operand xt[j = 4 kk + q][c = 16 ct + li]
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const float a = gt[li * GLD + 4 * kk + q];
#pragma unroll
      for (int ct = 0; ct < 8; ++ct) {
        const float b = xt[(4 * kk + q) * XLD + 16 * ct + li];
        dacc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, dacc[ct], 0, 0, 0);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  if (!gsym_out) {
    // cross-wave reduction in fixed order: red2[w][16][XLD] aliases the X tiles (all waves are past their last read)
    __syncthreads();
    float* red2 = Xt;
#pragma unroll
    for (int ct = 0; ct < 8; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) red2[(wid * 16 + 4 * q + r) * XLD + 16 * ct + li] = dacc[ct][r];
    __syncthreads();
    for (int e = tid; e < 16 * 32; e += NW * 64) {                     // 16 rows x 32 float4
      const int il = e >> 5, c4 = (e & 31) * 4;
      if (i0 + il >= A || c4 >= C) continue;
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        const float4 v = *reinterpret_cast<const float4*>(&red2[(w * 16 + il) * XLD + c4]);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      *reinterpret_cast<float4*>(&dX[(long long)(i0 + il) * lddx + c4]) = s;
    }
  }
  if (blockIdx.x == 0) {
    // loss = (1/A_v) sum_i loss_i, fixed-order tree (deterministic)
    __syncthreads();
    float s = 0.f;
    for (int j = tid; j < A; j += NW * 64) s += loss_row[j];
    s = dcs_wave_sum(s);
    if (lane == 0) red[wid] = s;
    __syncthreads();
    if (tid == 0) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) t += red[w];
      loss[0] = t * inv_av;
    }
  }
}

// ==================================================================================================================
// Large-A family (the all-gathered global anchor set of the data-parallel step: 8 x 608 = 4864 rows at C4).
//
// One kernel skeleton, contrast_strip_kernel<PHASE>: a block owns a 64-row strip I (its A operand lives in registers)
// and walks a chunk of 64-column tiles J; per tile the 4 waves compute S(I,J) = X_I X_J^T with v_mfma_f32_32x32x2
// (X_J staged once per tile in LDS, register-staged prefetch of the next tile) and hand the accumulators to the phase:
//   PHASE 1-3 = row statistics, SYMMETRIC: only tiles J >= I are visited; element (i,j) feeds the statistics of row i
//     (row direction: per-lane accumulators over the whole chunk, ONE cross-lane register-transpose reduction per
//     chunk) and, because S_ji = S_ij, of row j (column direction: 16 in-lane adds + one shuffle per tile).  Half the
//     similarity FLOPs of a row-only sweep.  Partials go to P[slot][row][4]; a combine kernel sums a row's slots in
//     fixed order (deterministic, no float atomics) and derives the row record for the next phase:
//       1: max, sum (s - r), sum (s - r)^2 with the reference shift r_i = S_ii/T   -> m_i, ||u_i||
//       2: den, cnt, sum E L, sum w L                                              -> den_i, 1/cnt_i (mode 1: done)
//       3: (mode 0, positives only) sum log-prob, q = sum 1/(E+den), sum den L/(E+den)   -> loss_i, q_i, <dL, L>_i
//   PHASE 4 = final sweep over ALL tiles J (chunked for parallelism): Gsym(I,J) = G_ij + G_ji in registers from the
//     records of both rows, staged through LDS as the A operand of dX_I += Gsym(I,J) X_J (second MFMA product, X_J
//     already in LDS); chunk partials of dX go to slabs that dcs_reduce_slab-style code sums in fixed order.
// FLOPs: 3 x 0.5 + 2 units of A^2 C 2 (one unit = 6.06 GFLOP at A = 4864, C = 128) against 3 algorithmic units.
constexpr int TB = 64;                 // tile edge
constexpr int XLDL = 132;              // LDS row stride of the X_J tile: 128 + 4 (odd number of 16-B slots: conflict-free b128 reads)

struct StripParams {
  const float* X; int ldx; const float* y; int ldy; const float* mask; int mb;
  int A, C, mode, ntile, CH;           // CH = tiles per chunk
  float it;
  const float* rnorm;                  // [A] reference shift r_i = it * ||x_i||^2
  const float* rec;                    // [A][REC] row records (phases >= 2)
  float* P;                            // [ntile][A][4] partials (phases 1-3)
  float* slab;                         // [nchunk][A][C] dX partials (phase 4)
  float* gsym; int ldg;                // phase 4 with C > 128: Gsym written out instead
  const float* av;                     // [1] number of valid rows
};

// C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
__device__ __forceinline__ int row32(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// Sum (or max) of v[0..16) over the 32 lanes that share lane>>5.  Register-transpose reduction: 16 shuffles instead of
// 80.  Returns in every lane the total of register index (lane & 31) >> 1.
template <bool MAX>
__device__ __forceinline__ float xlane32(const float (&v)[16], const int l31) {
  auto op = [](float a, float b) { return MAX ? fmaxf(a, b) : a + b; };
  float w8[8], w4[4], w2[2];
  const bool b4 = l31 & 16, b3 = l31 & 8, b2 = l31 & 4, b1 = l31 & 2;
#pragma unroll
  for (int k = 0; k < 8; ++k) { const float send = b4 ? v[k] : v[k + 8], keep = b4 ? v[k + 8] : v[k]; w8[k] = op(keep, __shfl_xor(send, 16, 64)); }
#pragma unroll
  for (int k = 0; k < 4; ++k) { const float send = b3 ? w8[k] : w8[k + 4], keep = b3 ? w8[k + 4] : w8[k]; w4[k] = op(keep, __shfl_xor(send, 8, 64)); }
#pragma unroll
  for (int k = 0; k < 2; ++k) { const float send = b2 ? w4[k] : w4[k + 2], keep = b2 ? w4[k + 2] : w4[k]; w2[k] = op(keep, __shfl_xor(send, 4, 64)); }
  const float send = b1 ? w2[0] : w2[1], keep = b1 ? w2[1] : w2[0];
  const float x = op(keep, __shfl_xor(send, 2, 64));
  return op(x, __shfl_xor(x, 1, 64));
}

// Per-row constants of the strip kernels' epilogues ("fast record", 8 floats in LDS), derived once per staged row from
// the row record {m, -, rn, den, 1/cnt, q, <dL,L>, clamp} so that one similarity element costs ~15 VALU instructions per
// direction (the straightforward form measured 146 per element pair: branches around both exponentials, divisions,
// five LDS reads per call):
//   [0] a2 = it rn log2(e)   [1] b2 = -m rn log2(e)      L log2(e) = fma(S, a2, b2),  E = v_exp_f32 of that
//   [2] k  = rn it / A_v (0 for a padding row: its gradient vanishes)          [3] den
//   [4] cp = den / cnt (mode 0) | 1 / cnt (mode 1)     [5] cn = q / cnt (mode 0) | 1 / den (mode 1)
//   [6] dot = <dL, L> (0 when the norm was clamped: no projection)             [7] y (label; < 0 = padding)
// phase 1 only needs [7] and the reference shift, kept in [0].
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }

template <int PHASE, int MODE>
__device__ __forceinline__ void make_fast(float* __restrict__ dst, const float* __restrict__ rec, const float y, const float rnorm,
                                          const float it, const float inv_av, const bool live) {
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
  if (PHASE == 1) a.x = rnorm;
  if (PHASE >= 2 && live) {
    const float4 r0 = ldg4(rec), r1 = ldg4(rec + 4);           // {m, -, rn, den}, {1/cnt, q, dot, clamp}
    const float rn = r0.z;
    a.x = it * rn * LOG2E; a.y = -r0.x * rn * LOG2E; a.z = rn * it * inv_av; a.w = r0.w;
    if (MODE == 0) { b.x = r0.w * r1.x; b.y = r1.y * r1.x; } else { b.x = r1.x; b.y = 1.f / r0.w; }
    b.z = r1.w != 0.f ? 0.f : r1.z;
  }
  b.w = live ? y : -1.f;
  *reinterpret_cast<float4*>(dst) = a;
  *reinterpret_cast<float4*>(dst + 4) = b;
}

// d loss / d S_ab * A_v-normalised, from the fast record R of row a.  pos = b is a positive of a (same label, not self),
// w = its weight (mode 1), self = (a == b).  Branch-free.
template <int MODE>
__device__ __forceinline__ float g_fast(const float s, const float (&R)[8], const bool same, const bool self, const float w) {
  const float L2 = fmaf(s, R[0], R[1]);
  const float E = exp2_fast(L2);
  const float L = L2 * LN2;
  float dL;
  if (MODE == 0) {
    const float dp = -R[4] * rcp_fast(E + R[3]);
    dL = same ? dp : E * R[5];
  } else dL = fmaf(E, R[5], -w * R[4]);
  dL = self ? 0.f : dL;
  return fmaf(-L, R[6], dL) * R[2];
}

template <int PHASE, int MODE>
__global__ __launch_bounds__(256, 2)
void contrast_strip_kernel(const StripParams p) {
  constexpr bool STATS = PHASE <= 3;
  // the strip's own rows live in LDS next to the streamed X_J tile (the per-lane row accumulators / the dX accumulators
  // need the registers).  PHASE 4 swaps the operand roles of the similarity product: A = the streamed tile (rows t),
  // B = the resident strip (rows q), so that the Gsym values, written in place over the accumulators (rows t in the
  // registers, q on the lanes), are DIRECTLY the A operand of the second product dX_Q[q][c] += sum_t Gsym[t][q] X_T[t][c]
  // (v_mfma_f32_32x32x2: A lane (i = lane&31, k = lane>>5); register r of the two lane halves = the k pair
  // row32(r,0), row32(r,1)): no transposition through LDS, no extra barrier.  Gsym is symmetric, so this is row block Q
  // of (G + G^T) X.
  constexpr bool ALDS = true;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* XJ = reinterpret_cast<float*>(smem_raw);                 // [64][XLDL]
  float* XI = XJ + TB * XLDL;                                     // [64][XLDL] (ALDS only)
  float* frI = XI + (ALDS ? TB * XLDL : 0);                       // [64][8] fast records of the strip rows
  float* frJ = frI + TB * 8;                                      // [64][8] fast records of the streamed tile's rows
  float* red = frJ + TB * 8;                                      // [2][64][4] cross-wave reduction scratch
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int wm = wid >> 1, wn = wid & 1;
  const int A = p.A, C = p.C;
  const float it = p.it;

  // block -> (strip I, first tile jbeg, end tile jend)
  int I, jbeg, jend;
  {
    int id = blockIdx.x;
    if (STATS) {
      I = 0;
      for (;;) { const int n = (p.ntile - I + p.CH - 1) / p.CH; if (id < n) break; id -= n; ++I; }
      jbeg = I + id * p.CH;
    } else {
      const int n = (p.ntile + p.CH - 1) / p.CH;
      I = id / n;
      jbeg = (id - I * n) * p.CH;
    }
    jend = jbeg + p.CH < p.ntile ? jbeg + p.CH : p.ntile;
  }
  const int i0 = I * TB;
  const float inv_av = PHASE == 4 ? 1.f / p.av[0] : 0.f;

  // strip-resident row data
  if (tid < TB) {
    const int r = i0 + tid;
    const bool in = r < A;
    const float yv = in ? p.y[(long long)r * p.ldy] : -1.f;
    make_fast<PHASE, MODE>(frI + tid * 8, p.rec + (long long)(in ? r : 0) * REC, yv, in ? p.rnorm[r] : 0.f, it, inv_av,
                           in && yv >= 0.f);
  }
  // A operand: lane (r = l31, h) holds X_I[32 wm + l31][8 g + 4 h .. +3] for the 16 k8-groups of a 128-channel chunk
  const int rowA = i0 + 32 * wm + l31 < A ? i0 + 32 * wm + l31 : -1;
  const int nkc = (C + 127) >> 7;
  float4 af[16];
  auto load_a = [&](int kc) {
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int k = kc * 128 + 8 * g + 4 * h;
      af[g] = (rowA >= 0 && k < C) ? ldg4(p.X + (long long)rowA * p.ldx + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  if (!ALDS && nkc == 1) load_a(0);
  // tile staging: thread -> 8 float4 of a [64][128] chunk image (rows of tile T, channels of chunk kc)
  float4 st[8];
  auto load_j = [&](int T, int kc, bool valid) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int e = tid + 256 * q, r = e >> 5, k = kc * 128 + (e & 31) * 4;
      const int row = T * TB + r;
      st[q] = (valid && row < A && k < C) ? ldg4(p.X + (long long)row * p.ldx + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_j = [&](float* dstT) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int e = tid + 256 * q, r = e >> 5, k4 = (e & 31) * 4;
      *reinterpret_cast<float4*>(&dstT[r * XLDL + k4]) = st[q];
    }
  };
  if (ALDS && nkc == 1) { load_j(I, 0, true); store_j(XI); }

  // phase accumulators
  constexpr bool HAS3 = PHASE == 2 && MODE == 1;
  float ra0[16], ra1[16], ra2[16], ra3[HAS3 ? 16 : 1];           // row direction (stats phases), per-lane partials
  f32x16 dacc[PHASE == 4 ? 4 : 1];                               // phase 4: dX rows 32 wn + .., channels 32 b + lane, partial over wm
#pragma unroll
  for (int r = 0; r < 16; ++r) { ra0[r] = PHASE == 1 ? -3.0e38f : 0.f; ra1[r] = 0.f; ra2[r] = 0.f; if (HAS3) ra3[r] = 0.f; }
#pragma unroll
  for (int b = 0; b < (PHASE == 4 ? 4 : 1); ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) dacc[b][r] = 0.f;

  // software pipeline: the registers `st` hold the NEXT tile (issued right after the previous one was written to LDS),
  // so its L2 / HBM latency is covered by this tile's MFMAs and epilogue
  load_j(jbeg, 0, true);
  // the tile's row records ride the same pipeline: threads 0..63 hold the next tile's fast record of one row
  float4 pf0 = make_float4(0.f, 0.f, 0.f, 0.f), pf1 = make_float4(0.f, 0.f, 0.f, -1.f);
  auto load_aux = [&](int T, bool valid) {
    if (tid < TB) {
      const int r = T * TB + tid;
      const bool in = valid && r < A;
      const float yv = in ? p.y[(long long)r * p.ldy] : -1.f;
      float tmp[8];
      make_fast<PHASE, MODE>(tmp, p.rec + (long long)(in ? r : 0) * REC, yv, in ? p.rnorm[r] : 0.f, it, inv_av, in && yv >= 0.f);
      pf0 = make_float4(tmp[0], tmp[1], tmp[2], tmp[3]); pf1 = make_float4(tmp[4], tmp[5], tmp[6], tmp[7]);
    }
  };
  load_aux(jbeg, true);
  for (int J = jbeg; J < jend; ++J) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int kc = 0; kc < nkc; ++kc) {
      __syncthreads();                                             // readers of the previous XJ image are done
      store_j(XJ);
      if (kc == 0 && tid < TB) {
        *reinterpret_cast<float4*>(frJ + tid * 8) = pf0;
        *reinterpret_cast<float4*>(frJ + tid * 8 + 4) = pf1;
      }
      if (nkc > 1) {
        if (ALDS) { load_j(I, kc, true); store_j(XI); } else load_a(kc);
      }
      if (kc + 1 < nkc) load_j(J, kc + 1, true);
      else { load_j(J + 1, 0, J + 1 < jend); load_aux(J + 1, J + 1 < jend); }                // prefetch
      __syncthreads();
      // phases 1-3: acc[r] = S[i = strip row 32 wm + row32(r,h)][j = tile row 32 wn + lane]
      // phase 4:    acc[r] = S[t = tile row 32 wm + row32(r,h)][q = strip row 32 wn + lane]
      const float* bj = &(PHASE == 4 ? XI : XJ)[(32 * wn + l31) * XLDL + 4 * h];
      const float* ai = &(PHASE == 4 ? XJ : XI)[(32 * wm + l31) * XLDL + 4 * h];
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const float4 b = *reinterpret_cast<const float4*>(bj + 8 * g);
        float4 a;
        if (ALDS) a = *reinterpret_cast<const float4*>(ai + 8 * g); else a = af[g];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
      }
    }
    // ---- epilogue: lane holds S[i = 32 wm + row32(r, h)][j = 32 wn + l31], r = 0..15 (phase 4: [t][q]) ----
    const int jl = 32 * wn + l31, jg = J * TB + jl;
    const bool diag = J == I;
    float Rl[8];                                                   // fast record of the lane's own row (tile row j / strip row q)
    {
      const float* src = (PHASE == 4 ? frI : frJ) + jl * 8;
      const float4 u = *reinterpret_cast<const float4*>(src), v = *reinterpret_cast<const float4*>(src + 4);
      Rl[0] = u.x; Rl[1] = u.y; Rl[2] = u.z; Rl[3] = u.w; Rl[4] = v.x; Rl[5] = v.y; Rl[6] = v.z; Rl[7] = v.w;
    }
    const float* frR = PHASE == 4 ? frJ : frI;                     // records of the rows held in the registers
    const float yl = Rl[7];
    auto col_out = [&](float c0, float c1, float c2, float c3, bool maxfirst) {
      // column-direction results of this tile -> P[I][tile rows j]: the two lane halves, then the two wm waves
      c0 = maxfirst ? fmaxf(c0, __shfl_xor(c0, 32, 64)) : c0 + __shfl_xor(c0, 32, 64);
      c1 += __shfl_xor(c1, 32, 64); c2 += __shfl_xor(c2, 32, 64); c3 += __shfl_xor(c3, 32, 64);
      if (h == 0) { float* q = &red[(wm * TB + jl) * 4]; q[0] = c0; q[1] = c1; q[2] = c2; q[3] = c3; }
      __syncthreads();
      if (tid < TB && J * TB + tid < A) {
        const float* q0 = &red[tid * 4]; const float* q1 = &red[(TB + tid) * 4];
        float* o = p.P + ((long long)I * A + J * TB + tid) * 4;
        o[0] = maxfirst ? fmaxf(q0[0], q1[0]) : q0[0] + q1[0];
        o[1] = q0[1] + q1[1]; o[2] = q0[2] + q1[2]; o[3] = q0[3] + q1[3];
      }
    };
    if (PHASE == 1) {
      float c0 = -3.0e38f, c1 = 0.f, c2 = 0.f;                     // column direction: statistics of row j over the rows i
      const float rj = Rl[0];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float* Rr = frR + (32 * wm + row32(r, h)) * 8;
        const float yi = Rr[7], ri = Rr[0];
        const float v = acc[r] * it;
        if (yi >= 0.f && yl >= 0.f) {
          const float d = v - ri;
          ra0[r] = fmaxf(ra0[r], v); ra1[r] += d; ra2[r] = fmaf(d, d, ra2[r]);
          if (!diag) { const float e = v - rj; c0 = fmaxf(c0, v); c1 += e; c2 = fmaf(e, e, c2); }
        }
      }
      if (!diag) col_out(c0, c1, c2, 0.f, true);                   // uniform per block
    } else if (PHASE == 2) {
      float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int il = 32 * wm + row32(r, h), ig = i0 + il;
        const float* Rr = frR + il * 8;
        const float4 u = *reinterpret_cast<const float4*>(Rr);
        const float yi = Rr[7];
        const float s_ = acc[r];
        const bool valid = yi >= 0.f && yl >= 0.f, same = yi == yl, self = ig == jg;
        // row i, column j: E = exp(L_ij) with row i's shift / norm; branch-free 0/1 weights
        const float L2i = fmaf(s_, u.x, u.y), Ei = exp2_fast(L2i), Li = L2i * LN2;
        const float L2j = fmaf(s_, Rl[0], Rl[1]), Ej = exp2_fast(L2j), Lj = L2j * LN2;
        if (MODE == 0) {
          const float neg = (valid && !same) ? 1.f : 0.f, pos = (valid && same && !self) ? 1.f : 0.f;
          ra0[r] = fmaf(neg, Ei, ra0[r]); ra2[r] = fmaf(neg * Ei, Li, ra2[r]); ra1[r] += pos;
          if (!diag) { c0 = fmaf(neg, Ej, c0); c2 = fmaf(neg * Ej, Lj, c2); c1 += pos; }
        } else {
          const float off = (valid && !self) ? 1.f : 0.f;
          const float wi = off * pos_weight(p.mask, p.mb, ig, jg, yi, yl), wj = off * pos_weight(p.mask, p.mb, jg, ig, yl, yi);
          ra0[r] = fmaf(off, Ei, ra0[r]); ra2[r] = fmaf(off * Ei, Li, ra2[r]); ra1[r] += wi; if (HAS3) ra3[r] = fmaf(wi, Li, ra3[r]);
          if (!diag) { c0 = fmaf(off, Ej, c0); c2 = fmaf(off * Ej, Lj, c2); c1 += wj; c3 = fmaf(wj, Lj, c3); }
        }
      }
      if (!diag) col_out(c0, c1, c2, c3, false);
    } else if (PHASE == 3) {
      float c0 = 0.f, c1 = 0.f, c2 = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int il = 32 * wm + row32(r, h), ig = i0 + il;
        const float* Rr = frR + il * 8;
        const float4 u = *reinterpret_cast<const float4*>(Rr);
        const float yi = Rr[7];
        if (yi < 0.f || yi != yl || ig == jg) continue;            // positives only (~1/19 of the pairs: branch, no select)
        const float s_ = acc[r];
        {
          const float L2 = fmaf(s_, u.x, u.y), L = L2 * LN2, den = u.w;
          const float d = exp2_fast(L2) + den, id = rcp_fast(d);
          ra0[r] += L - __logf(d); ra1[r] += id; ra2[r] = fmaf(den * id, L, ra2[r]);
        }
        if (!diag) {
          const float L2 = fmaf(s_, Rl[0], Rl[1]), L = L2 * LN2, den = Rl[3];
          const float d = exp2_fast(L2) + den, id = rcp_fast(d);
          c0 += L - __logf(d); c1 += id; c2 = fmaf(den * id, L, c2);
        }
      }
      if (!diag) col_out(c0, c1, c2, 0.f, false);
    } else {
      // PHASE 4 (roles swapped, see above): lane = strip row q, registers = tile rows t
      const int qg = i0 + jl;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int tl = 32 * wm + row32(r, h), tg = J * TB + tl;
        const float* Rr = frR + tl * 8;
        const float4 u = *reinterpret_cast<const float4*>(Rr), v = *reinterpret_cast<const float4*>(Rr + 4);
        const float Rt[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
        const float yt = v.w;
        const bool same = yl == yt, self = qg == tg;
        const float wqt = MODE == 1 ? pos_weight(p.mask, p.mb, qg, tg, yl, yt) : 0.f;
        const float wtq = MODE == 1 ? pos_weight(p.mask, p.mb, tg, qg, yt, yl) : 0.f;
        // a padding row has k = 0 in its own record; the partner's term must vanish too
        const float vmask = (yl >= 0.f && yt >= 0.f) ? 1.f : 0.f;
        const float gs = (g_fast<MODE>(acc[r], Rl, same, self, wqt) + g_fast<MODE>(acc[r], Rt, same, self, wtq)) * vmask;
        if (p.gsym) { if (qg < A && tg < A) p.gsym[(long long)qg * p.ldg + tg] = gs; }
        acc[r] = gs;
      }
      if (!p.gsym) {
        // dX_Q[q = lane][c = 32 b + n] += sum over this wave's 32 tile rows t of Gsym[t][q] X_T[t][c]
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float* xb = &XJ[(32 * wm + row32(r, h)) * XLDL + l31];
#pragma unroll
          for (int b = 0; b < 4; ++b) dacc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(acc[r], xb[32 * b], dacc[b], 0, 0, 0);
        }
      }
    }
  }

  if (STATS) {
    // row direction: one cross-lane reduction per chunk, then the two column halves (wn) through LDS
    float t0, t1, t2, t3 = 0.f;
    t0 = PHASE == 1 ? xlane32<true>(ra0, l31) : xlane32<false>(ra0, l31);
    t1 = xlane32<false>(ra1, l31);
    t2 = xlane32<false>(ra2, l31);
    if (HAS3) { float r3[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) r3[r] = ra3[HAS3 ? r : 0];
      t3 = xlane32<false>(r3, l31); }
    __syncthreads();
    if ((l31 & 1) == 0) {
      const int il = 32 * wm + row32(l31 >> 1, h);
      float* q = &red[(wn * TB + il) * 4];
      q[0] = t0; q[1] = t1; q[2] = t2; q[3] = t3;
    }
    __syncthreads();
    if (tid < TB && i0 + tid < A) {
      const float* q0 = &red[tid * 4]; const float* q1 = &red[(TB + tid) * 4];
      float* o = p.P + ((long long)jbeg * A + i0 + tid) * 4;        // slot = first tile of the chunk (>= I)
      o[0] = PHASE == 1 ? fmaxf(q0[0], q1[0]) : q0[0] + q1[0];
      o[1] = q0[1] + q1[1]; o[2] = q0[2] + q1[2]; o[3] = q0[3] + q1[3];
    }
  } else if (!p.gsym) {
    // the two row halves (wm) of every tile hold partial sums for the same strip rows: add them through LDS (the tile
    // images are free now), then one 16-byte store per lane into this chunk's slab
    __syncthreads();
    float* red2 = XJ;                                               // [2][64][XLDL] spans XJ and XI
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) red2[(wm * TB + 32 * wn + row32(r, h)) * XLDL + 32 * b + l31] = dacc[b][r];
    __syncthreads();
    const int chunk = jbeg / p.CH;
    float* o = p.slab + (long long)chunk * A * C;
    for (int e = tid; e < TB * 32; e += 256) {
      const int ql = e >> 5, c4 = (e & 31) * 4;
      if (i0 + ql >= A || c4 >= C) continue;
      const float4 u = *reinterpret_cast<const float4*>(&red2[ql * XLDL + c4]);
      const float4 v = *reinterpret_cast<const float4*>(&red2[(TB + ql) * XLDL + c4]);
      *reinterpret_cast<float4*>(&o[(long long)(i0 + ql) * C + c4]) = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
    }
  }
}

// r_i = it * ||x_i||^2 (the reference shift of phase 1) and the number of valid rows.
__global__ __launch_bounds__(256)
void contrast_prep_kernel(const float* __restrict__ X, int ldx, const float* __restrict__ y, int ldy, int A, int C, float it,
                          float* __restrict__ rnorm, float* __restrict__ av) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int i = blockIdx.x * 4 + w; i < A; i += gridDim.x * 4) {
    float s = 0.f;
    for (int k = lane * 4; k < C; k += 256) { const float4 v = ldg4(X + (long long)i * ldx + k); s = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, s)))); }
    s = dcs_wave_sum(s);
    if (lane == 0) rnorm[i] = s * it;
  }
  if (blockIdx.x == 0) {
    __shared__ float sm[4];
    float n = 0.f;
    for (int j = threadIdx.x; j < A; j += 256) n += y[(long long)j * ldy] >= 0.f ? 1.f : 0.f;
    n = dcs_wave_sum(n);
    if (lane == 0) sm[w] = n;
    __syncthreads();
    if (threadIdx.x == 0) av[0] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
  }
}

// Sum a row's partial slots in fixed order and derive its record.  Slots of row r (strip R = r / 64): column-direction
// results of tiles (t, R), t < R, at slot t; row-direction results of the chunks of strip R at slots R + c * CH.
template <int PHASE>
__global__ __launch_bounds__(256)
void contrast_combine_kernel(const float* __restrict__ P, const float* __restrict__ y, int ldy, const float* __restrict__ rnorm,
                             const float* __restrict__ av, int A, int ntile, int CH, int mode, float* __restrict__ rec,
                             float* __restrict__ loss_row) {
  // 16 lanes per row: lane k sums slots k, k+16, ... (independent loads in flight), then a fixed-order butterfly
  const int r = blockIdx.x * 16 + (threadIdx.x >> 4), k = threadIdx.x & 15;
  const bool inrange = r < A;
  const int rr = inrange ? r : A - 1;
  float* rc = rec + (long long)rr * REC;
  const bool live = inrange && y[(long long)rr * ldy] >= 0.f;
  const int R = rr / TB;
  double a1 = 0.0, a2 = 0.0, a3 = 0.0, a0 = PHASE == 1 ? -3.0e38 : 0.0;
  if (live) {
    // valid slots of row r: t < R (column-direction results of tiles (t, R)) and t = R + c * CH (row-direction chunks)
    for (int t = k; t < ntile; t += 16) {
      if (t >= R && ((t - R) % CH) != 0) continue;
      const float4 v = ldg4(P + ((long long)t * A + rr) * 4);
      if (PHASE == 1) a0 = fmax(a0, (double)v.x); else a0 += (double)v.x;
      a1 += (double)v.y; a2 += (double)v.z; a3 += (double)v.w;
    }
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) {
    const double b0 = __shfl_xor(a0, o, 64), b1 = __shfl_xor(a1, o, 64), b2 = __shfl_xor(a2, o, 64), b3 = __shfl_xor(a3, o, 64);
    if (PHASE == 1) a0 = fmax(a0, b0); else a0 += b0;
    a1 += b1; a2 += b2; a3 += b3;
  }
  if (k != 0 || !inrange) return;
  if (!live) {
    if (PHASE == 1) {
#pragma unroll
      for (int e = 0; e < REC; ++e) rc[e] = 0.f;
      loss_row[r] = 0.f;
    }
    return;
  }
  if (PHASE == 1) {
    // n2 = sum (s - m)^2 = sum (s - r)^2 - 2 (m - r) sum (s - r) + A_v (m - r)^2, in double; 0 <= m - r <= range
    const double dm = a0 - (double)rnorm[r];
    double n2 = a2 - 2.0 * dm * a1 + (double)av[0] * dm * dm;
    if (n2 < 0.0) n2 = 0.0;
    const float nraw = (float)sqrt(n2);
    rc[0] = (float)a0; rc[1] = 0.f; rc[2] = 1.f / fmaxf(nraw, 1e-12f); rc[7] = nraw <= 1e-12f ? 1.f : 0.f;
  } else if (PHASE == 2) {
    const float den = (float)a0, cnt = (float)a1, sEL = (float)a2, swL = (float)a3;
    const float icnt = 1.f / cnt;
    rc[3] = den; rc[4] = icnt;
    if (mode == 1) {
      rc[5] = 0.f; rc[6] = sEL / den - swL * icnt;
      loss_row[r] = -(swL - cnt * logf(den)) * icnt;
    } else {
      rc[6] = sEL;                                                // parked until phase 3 delivers q
    }
  } else {
    const float icnt = rc[4], sEL = rc[6];
    const float lp = (float)a0, qv = (float)a1, sdl = (float)a2;
    rc[5] = qv; rc[6] = (qv * sEL - sdl) * icnt;
    loss_row[r] = -lp * icnt;
  }
}

// dX[i][c] = sum over the chunk slabs (fixed order); block 0 also reduces the loss.
__global__ __launch_bounds__(256)
void contrast_finish_kernel(const float* __restrict__ slab, int nchunk, long long n, float* __restrict__ dX, int C, int lddx,
                            const float* __restrict__ loss_row, const float* __restrict__ av, int A, float* __restrict__ loss) {
  if (slab) {
    for (long long e = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; e < n; e += (long long)gridDim.x * 1024) {
      float4 s = ldg4(slab + e);
      for (int k = 1; k < nchunk; ++k) { const float4 v = ldg4(slab + (long long)k * n + e); s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
      const long long i = e / C; const int c = (int)(e - i * C);
      *reinterpret_cast<float4*>(&dX[i * lddx + c]) = s;
    }
  }
  if (blockIdx.x == 0) {
    __shared__ double sm[4];
    double s = 0.0;
    for (int j = threadIdx.x; j < A; j += 256) s += (double)loss_row[j];
    s = dcs_wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = (float)(((sm[0] + sm[1]) + (sm[2] + sm[3])) / (double)av[0]);
  }
}

size_t strip_smem(int phase) {
  (void)phase;
  return (size_t)(2 * TB * XLDL + 2 * TB * 8 + 2 * TB * 4) * sizeof(float);
}

int large_ws_floats(int A, int C, int64_t* out) {
  const int64_t ntile = (A + TB - 1) / TB;
  // rec, loss_row, rnorm, av(+pad), P [ntile][A][4], slabs [nchunk <= 16][A][C]
  *out = (int64_t)A * (REC + 2) + 64 + ntile * A * 4 + 16ll * A * C + 64;
  return 0;
}

int launch_large(const float* X, int ldx, const float* y, int ldy, const float* mask, int mb, int A, int C, int mode, float it,
                 float* loss, float* dX, int lddx, float* gsym, int ldg, float* ws, hipStream_t s) {
  const int ntile = (A + TB - 1) / TB;
  float* rec = ws;
  float* loss_row = rec + (size_t)A * REC;
  float* rnorm = loss_row + A;
  float* av = rnorm + A;                       // 64-float slot (keeps the partials 16-B aligned)
  float* P = av + 64 - (((size_t)A * (REC + 2)) & 3);
  P += (4 - ((P - ws) & 3)) & 3;
  float* slab = P + (size_t)ntile * A * 4;
  slab += (4 - ((slab - ws) & 3)) & 3;
  StripParams p;
  p.X = X; p.ldx = ldx; p.y = y; p.ldy = ldy; p.mask = mask; p.mb = mb; p.A = A; p.C = C; p.mode = mode; p.ntile = ntile;
  p.it = it; p.rnorm = rnorm; p.rec = rec; p.P = P; p.slab = slab; p.gsym = gsym; p.ldg = ldg; p.av = av;
  // statistics sweeps: ~3 blocks per CU.  chunks per strip I = ceil((ntile - I) / CH)
  int CH = 1;
  for (; CH < 16; ++CH) {
    long long nb = 0;
    for (int I = 0; I < ntile; ++I) nb += (ntile - I + CH - 1) / CH;
    if (nb <= 768) break;
  }
  long long nb_stats = 0;
  for (int I = 0; I < ntile; ++I) nb_stats += (ntile - I + CH - 1) / CH;
  p.CH = CH;
#define LAUNCH_STRIP(PH, grid, block, sh, st, prm)                                                     \
  do {                                                                                                \
    if (mode == 0) hipLaunchKernelGGL((contrast_strip_kernel<PH, 0>), grid, block, sh, st, prm);       \
    else hipLaunchKernelGGL((contrast_strip_kernel<PH, 1>), grid, block, sh, st, prm);                 \
  } while (0)
  hipLaunchKernelGGL(contrast_prep_kernel, dim3(256), dim3(256), 0, s, X, ldx, y, ldy, A, C, it, rnorm, av);
  const dim3 cg((A + 15) / 16);
  auto set_attr = [](const void* f, size_t sh) {
    return sh <= 64 * 1024 || hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh) == hipSuccess;
  };
  if (!set_attr(reinterpret_cast<const void*>(contrast_strip_kernel<1, 0>), strip_smem(1)) ||
      !set_attr(reinterpret_cast<const void*>(contrast_strip_kernel<1, 1>), strip_smem(1)) ||
      !set_attr(reinterpret_cast<const void*>(contrast_strip_kernel<2, 0>), strip_smem(2)) ||
      !set_attr(reinterpret_cast<const void*>(contrast_strip_kernel<2, 1>), strip_smem(2)) ||
      !set_attr(reinterpret_cast<const void*>(contrast_strip_kernel<3, 0>), strip_smem(3)) ||
      !set_attr(reinterpret_cast<const void*>(contrast_strip_kernel<3, 1>), strip_smem(3))) return DCS_E_LAUNCH;
  LAUNCH_STRIP(1, dim3((unsigned)nb_stats), dim3(256), strip_smem(1), s, p);
  hipLaunchKernelGGL(contrast_combine_kernel<1>, cg, dim3(256), 0, s, P, y, ldy, rnorm, av, A, ntile, CH, mode, rec, loss_row);
  LAUNCH_STRIP(2, dim3((unsigned)nb_stats), dim3(256), strip_smem(2), s, p);
  hipLaunchKernelGGL(contrast_combine_kernel<2>, cg, dim3(256), 0, s, P, y, ldy, rnorm, av, A, ntile, CH, mode, rec, loss_row);
  if (mode == 0) {
    LAUNCH_STRIP(3, dim3((unsigned)nb_stats), dim3(256), strip_smem(3), s, p);
    hipLaunchKernelGGL(contrast_combine_kernel<3>, cg, dim3(256), 0, s, P, y, ldy, rnorm, av, A, ntile, CH, mode, rec, loss_row);
  }
  // final sweep: all tiles of every strip, chunked so that ~3 blocks per CU run; <= 16 dX slabs
  int CH4 = (ntile * ntile + 767) / 768;
  if (CH4 < (ntile + 15) / 16) CH4 = (ntile + 15) / 16;
  if (CH4 < 1) CH4 = 1;
  const int nchunk = (ntile + CH4 - 1) / CH4;
  p.CH = CH4;
  if (!set_attr(reinterpret_cast<const void*>(contrast_strip_kernel<4, 0>), strip_smem(4)) ||
      !set_attr(reinterpret_cast<const void*>(contrast_strip_kernel<4, 1>), strip_smem(4))) return DCS_E_LAUNCH;
  LAUNCH_STRIP(4, dim3((unsigned)(ntile * nchunk)), dim3(256), strip_smem(4), s, p);
  const long long n = (long long)A * C;
  hipLaunchKernelGGL(contrast_finish_kernel, dim3(gsym ? 1u : 512u), dim3(256), 0, s, gsym ? nullptr : slab, nchunk, n, dX, C,
                     lddx, loss_row, av, A, loss);
#undef LAUNCH_STRIP
  DCS_LAUNCH_RET();
}

constexpr int SMALL_MAX = 1024;
constexpr int NW_S = 8;

template <typename ACC>
int launch_small(const float* X, int ldx, const float* y, int ldy, const float* mask, int mb, int A, int C, int mode,
                 float it, float* loss, float* dX, int lddx, float* gsym, int ldg, float* ws, hipStream_t s) {
  float* rec = ws;
  float* loss_row = ws + (size_t)A * REC;
  int AP = (A + 15) / 16 * 16;
  AP += (20 - (AP & 31) + 32) & 31;                                   // row stride = 20 mod 32 floats
  const size_t sh1 = (size_t)16 * AP * sizeof(ACC) + (size_t)AP * 4;
  const int nb = (A + 15) / 16;
  auto k1 = contrast_small_stats_kernel<ACC, NW_S>;
  auto k2 = contrast_small_final_kernel<ACC, NW_S>;
  if (sh1 > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh1) != hipSuccess)
    return DCS_E_LAUNCH;
  hipLaunchKernelGGL(k1, dim3(nb), dim3(NW_S * 64), sh1, s, X, ldx, y, ldy, mask, mb, A, C, mode, it, rec, loss_row, AP);
  const size_t sh2 = (size_t)(16 * REC + 16 + NW_S + 4 + NW_S * 16 * 17 + 4 + NW_S * 16 * 144) * 4;
  if (sh2 > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh2) != hipSuccess)
    return DCS_E_LAUNCH;
  hipLaunchKernelGGL(k2, dim3(nb), dim3(NW_S * 64), sh2, s, X, ldx, y, ldy, mask, mb, A, C, mode, it, rec, loss_row, loss, dX,
                     lddx, gsym, ldg);
  DCS_LAUNCH_RET();
}

}  // namespace

extern "C" int dcs_contrast_fused_ws(int A, int C, int64_t* floats) {
  DCS_CHECK_ARG(A > 0 && C > 0 && floats);
  if (A > SMALL_MAX) large_ws_floats(A, C, floats);
  else *floats = (int64_t)A * (REC + 1) + 64;
  return DCS_OK;
}

extern "C" int dcs_contrast_fused(const float* X, int ldx, const float* y, int ldy, const float* mask, int mask_b, int A,
                                  int C, int mode, float inv_temp, float* loss, float* dX, int lddx, float* gsym, int ldg,
                                  float* ws, int64_t ws_floats, void* stream) {
  DCS_CHECK_ARG(X && y && loss && ws && A > 0 && C > 0 && (C & 3) == 0 && (ldx & 3) == 0 && ldx >= C && ldy >= 1);
  DCS_CHECK_ARG(mode == 0 || mode == 1);
  DCS_CHECK_ARG(dcs_aligned16(X) && dcs_aligned16(ws) && (!mask || (mode == 1 && mask_b > 0 && A % mask_b == 0)));
  DCS_CHECK_ARG((dX != nullptr) != (gsym != nullptr));              // exactly one output form
  DCS_CHECK_ARG(!dX || ((lddx & 3) == 0 && lddx >= C && C <= 128 && dcs_aligned16(dX)));
  DCS_CHECK_ARG(!gsym || ldg >= A);
  int64_t need = 0;
  dcs_contrast_fused_ws(A, C, &need);
  DCS_CHECK_ARG(ws_floats >= need);
  hipStream_t s = dcs_stream(stream);
  if (A > SMALL_MAX) {
    DCS_CHECK_ARG((long long)A * (C > A ? C : A) * 16 < 0x7FFFFFFFll * 4ll);
    return launch_large(X, ldx, y, ldy, mask, mask_b, A, C, mode, inv_temp, loss, dX, lddx, gsym, ldg, ws, s);
  }
  if (mode == 1) return launch_small<double>(X, ldx, y, ldy, mask, mask_b, A, C, mode, inv_temp, loss, dX, lddx, gsym, ldg, ws, s);
  return launch_small<float>(X, ldx, y, ldy, mask, mask_b, A, C, mode, inv_temp, loss, dX, lddx, gsym, ldg, ws, s);
}
