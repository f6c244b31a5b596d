// Fused embedding-similarity / InfoNCE loss.  A <= 1024: S = X X^T never leaves the chip; larger A: S lives once in the
// workspace (memory-side cache resident), see contrast_large.h.
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
//   * larger A (the all-gathered global set of the data-parallel step, 4864 rows at C4): contrast_large.h -- S is computed
//       once (upper-triangular tiles on the matrix cores) and kept in the workspace, one wave per row derives the row
//       record with the row in registers, the gradient product forms G + G^T on the fly.
// Mode 1 with A <= small max accumulates S in float64 on the f64 matrix cores (v_mfma_f64_16x16x4_f64): pooled image
// embeddings of one batch are nearly parallel, S_ij it - m_i then cancels 4-5 digits, and a k-ordered fp32 FMA chain
// (what the f32 MFMA is) loses them where the reference's blocked CPU GEMM does not (measured: 7x the reference's own
// fp32-vs-fp64 error on the projection-head gradients; with f64 accumulation 0.1x).  2B <= 512 rows: free.
#include "dcs_common.h"

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
__device__ __forceinline__
void contrast_small_stats_body(unsigned char* smem_raw, const float* __restrict__ X, const int ldx, const float* __restrict__ y,
                               const int ldy, const float* __restrict__ mask, const int mb, const int A, const int C,
                               const int mode, const float it, float* __restrict__ rec, float* __restrict__ loss_row,
                               const int AP) {
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
// FUSED (one cooperative launch, see contrast_small_fused_kernel): the S strip of the statistics phase is still in LDS
// (Sstrip [16][AP]), so the tiles are read back instead of being recomputed.
template <typename ACC, int NW, bool FUSED>
__device__ __forceinline__
void contrast_small_final_body(unsigned char* smem_raw, const ACC* Sstrip, const int AP, const float* __restrict__ X, const int ldx,
                               const float* __restrict__ y, const int ldy, const float* __restrict__ mask, const int mb,
                               const int A, const int C, const int mode, const float it, const float* __restrict__ rec,
                               const float* __restrict__ loss_row, float* __restrict__ loss, float* __restrict__ dX,
                               const int lddx, float* __restrict__ gsym_out, const int ldg) {
  constexpr int XLD = 144;                       // 128 + 16: rows 16 banks apart -> conflict-free k-strided B reads
  constexpr int GLD = 17;
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
    typename Mfma16<ACC>::acc_t acc;
    if constexpr (FUSED) {
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = Sstrip[Mfma16<ACC>::row(lane, r) * AP + 16 * jt + li];
    } else {
      acc = regs ? s_tile_regs<ACC>(xa, xb) : s_tile<ACC, 8>(xa, X, ldx, C, rowI, rowJ, q);
    }
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

template <typename ACC, int NW>
__global__ __launch_bounds__(NW * 64)
void contrast_small_stats_kernel(const float* __restrict__ X, const int ldx, const float* __restrict__ y, const int ldy,
                                 const float* __restrict__ mask, const int mb, const int A, const int C, const int mode,
                                 const float it, float* __restrict__ rec, float* __restrict__ loss_row, const int AP) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  contrast_small_stats_body<ACC, NW>(smem_raw, X, ldx, y, ldy, mask, mb, A, C, mode, it, rec, loss_row, AP);
}

template <typename ACC, int NW>
__global__ __launch_bounds__(NW * 64)
void contrast_small_final_kernel(const float* __restrict__ X, const int ldx, const float* __restrict__ y, const int ldy,
                                 const float* __restrict__ mask, const int mb, const int A, const int C, const int mode,
                                 const float it, const float* __restrict__ rec, const float* __restrict__ loss_row,
                                 float* __restrict__ loss, float* __restrict__ dX, const int lddx,
                                 float* __restrict__ gsym_out, const int ldg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  contrast_small_final_body<ACC, NW, false>(smem_raw, nullptr, 0, X, ldx, y, ldy, mask, mb, A, C, mode, it, rec, loss_row, loss, dX,
                                            lddx, gsym_out, ldg);
}

// Both phases in ONE cooperative launch (A <= 1024: <= 64 blocks, all resident): statistics with the S strip in LDS, a
// grid-wide barrier (every block needs the records of ALL rows for G_ji), then the gradient sweep that reads the strip
// back from LDS instead of recomputing S.  One launch latency instead of two dependent ones, half the similarity FLOPs.
struct SmallFusedArgs {
  const float* X; int ldx; const float* y; int ldy; const float* mask; int mb, A, C, mode; float it;
  float* rec; float* loss_row; float* loss; float* dX; int lddx; float* gsym; int ldg, AP, final_off;
  unsigned* sync;                      // [2] in the workspace: arrivals at the barrier, departures (the last one re-zeroes both)
};
template <typename ACC, int NW>
__global__ __launch_bounds__(NW * 64)
void contrast_small_fused_kernel(const SmallFusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  contrast_small_stats_body<ACC, NW>(smem_raw, a.X, a.ldx, a.y, a.ldy, a.mask, a.mb, a.A, a.C, a.mode, a.it, a.rec, a.loss_row, a.AP);
  // grid-wide barrier on a counter in the workspace (zero on entry: dcs_contrast_fused's contract; <= 64 blocks of one per
  // CU are all resident).  Release: the records / loss rows of this strip; acquire: those of every other strip.  The wait
  // is bounded (a hung peer must not hang the device): ~1 s of polling, then the block goes on.
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    atomicAdd(a.sync, 1u);
    const unsigned target = gridDim.x;
    for (unsigned spin = 0; spin < (1u << 22); ++spin) {
      if (__hip_atomic_load(a.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) break;
      __builtin_amdgcn_s_sleep(2);
    }
    __threadfence();
  }
  __syncthreads();
  contrast_small_final_body<ACC, NW, true>(smem_raw + a.final_off, reinterpret_cast<const ACC*>(smem_raw), a.AP, a.X, a.ldx, a.y,
                                           a.ldy, a.mask, a.mb, a.A, a.C, a.mode, a.it, a.rec, a.loss_row, a.loss, a.dX, a.lddx,
                                           a.gsym, a.ldg);
  // the last block to leave hands the counters back as zeros (the next call's contract)
  __syncthreads();
  if (threadIdx.x == 0 && atomicAdd(a.sync + 1, 1u) == gridDim.x - 1) { a.sync[0] = 0u; a.sync[1] = 0u; __threadfence(); }
}

#include "contrast_large.h"

constexpr int SMALL_MAX = 1024;
constexpr int NW_S = 8;
#define g_small_fused (dcs_config().contrast_fused != 0)

template <typename ACC>
int launch_small(const float* X, int ldx, const float* y, int ldy, const float* mask, int mb, int A, int C, int mode,
                 float it, float* loss, float* dX, int lddx, float* gsym, int ldg, float* ws, float* sync, hipStream_t s) {
  float* rec = ws;
  float* loss_row = ws + (size_t)A * REC;
  int AP = (A + 15) / 16 * 16;
  AP += (20 - (AP & 31) + 32) & 31;                                   // row stride = 20 mod 32 floats
  const size_t sh1 = (size_t)16 * AP * sizeof(ACC) + (size_t)AP * 4;
  const int nb = (A + 15) / 16;
  auto k1 = contrast_small_stats_kernel<ACC, NW_S>;
  auto k2 = contrast_small_final_kernel<ACC, NW_S>;
  const size_t sh2 = (size_t)(16 * REC + 16 + NW_S + 4 + NW_S * 16 * 17 + 4 + NW_S * 16 * 144) * 4;
  // one cooperative launch when both phases' LDS fits next to each other (always for the pixel contrast, A <= 1024)
  const size_t off = (sh1 + 15) & ~(size_t)15;
  if (off + sh2 <= 160 * 1024 && sync != nullptr && g_small_fused) {
    auto kf = contrast_small_fused_kernel<ACC, NW_S>;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kf), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(off + sh2)) != hipSuccess)
      return DCS_E_LAUNCH;
    SmallFusedArgs a{X, ldx, y, ldy, mask, mb, A, C, mode, it, rec, loss_row, loss, dX, lddx, gsym, ldg, AP, (int)off,
                     reinterpret_cast<unsigned*>(sync)};
    hipLaunchKernelGGL(kf, dim3(nb), dim3(NW_S * 64), off + sh2, s, a);
    DCS_LAUNCH_RET();
  }
  if (sh1 > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh1) != hipSuccess)
    return DCS_E_LAUNCH;
  hipLaunchKernelGGL(k1, dim3(nb), dim3(NW_S * 64), sh1, s, X, ldx, y, ldy, mask, mb, A, C, mode, it, rec, loss_row, AP);
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
                                  float* ws, int64_t ws_floats, uint32_t* sync, void* stream) {
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
    if (A > LARGE_MAX) return DCS_E_UNSUPPORTED;                   // the row kernel keeps a row in <= 128 registers per lane
    return launch_large(X, ldx, y, ldy, mask, mask_b, A, C, mode, inv_temp, loss, dX, lddx, gsym, ldg, ws, s);
  }
  float* sy = reinterpret_cast<float*>(sync);
  if (mode == 1) return launch_small<double>(X, ldx, y, ldy, mask, mask_b, A, C, mode, inv_temp, loss, dX, lddx, gsym, ldg, ws, sy, s);
  return launch_small<float>(X, ldx, y, ldy, mask, mask_b, A, C, mode, inv_temp, loss, dX, lddx, gsym, ldg, ws, sy, s);
}
