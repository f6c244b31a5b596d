// Large-A family of the fused similarity / InfoNCE loss (included by contrast_fused.hip, inside its anonymous namespace):
// the all-gathered global anchor set of the data-parallel step, 8 x 608 = 4864 rows at BASELINE config 4.  Round 3.
//
// The reference's loss (utils/loss.py:339-389, :175-204) needs FOUR dependent passes over a row of S = X X^T / T (max; norm
// of the shifted row; denominators; the positives' log-probabilities, which need the denominators) before the gradient can
// be formed.  Round 2 recomputed the S tiles on the matrix cores in every pass (3 half sweeps + 1 full sweep, 440 us at
// A = 4864, every pass latency / epilogue bound at 21-41 % MFMA busy).  Now S is computed ONCE and kept: [AP][AP] fp32 in the
// workspace (95 MB at A = 4864: it lives in the 256 MB memory-side cache between its producer and its two consumers), and
//   contrast_s_kernel    S = X X^T on the matrix cores; only the upper-triangular 64x64 tiles are computed, each is written
//                        twice (as is, and transposed through LDS), so the full matrix exists with half the FLOPs;
//   contrast_row_kernel  ONE block per row i with the row in REGISTERS: max -> norm -> denominators -> positives (compacted
//                        through LDS, they are ~1/19 of a row) -> the row record {m, rn, den, 1/cnt, q,
//                        <dL,L>, clamp} and loss_i.  No partial sums, no combine launches, no cross-block traffic;
//   contrast_dx_kernel   dX_I = sum_J Gsym(I,J) X_J: a 128-row strip per block, Gsym(I,J) = G_ij + G_ji is formed from the
//                        S(I,J) tile and the records of both rows on its way from the staging registers into LDS (the A
//                        operand of the product; X_J is the B operand), K chunks -> slabs summed in fixed order;
//   contrast_gsym_kernel (C > 128, DeepLab's 2048-wide rows) writes Gsym out instead; the caller finishes dX with one GEMM;
//   contrast_finish_kernel sums the slabs and the loss rows (deterministic: no float atomics anywhere).
#pragma once

constexpr int TB = 64;                 // tile edge
constexpr int XLDL = 132;              // LDS row stride of a [64][128] X tile: 128 + 4 (odd number of 16-B slots: conflict-free b128 reads)

// C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
__device__ __forceinline__ int row32(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// Per-row constants of the gradient ("fast record", 8 floats), derived once per staged row from the row record
// {m, -, rn, den, 1/cnt, q, <dL,L>, clamp} so that one similarity element costs ~15 VALU instructions per direction:
//   [0] a2 = it rn log2(e)   [1] b2 = -m rn log2(e)      L log2(e) = fma(S, a2, b2),  E = v_exp_f32 of that
//   [2] k  = rn it / A_v (0 for a padding row: its gradient vanishes)          [3] den
//   [4] cp = den / cnt (mode 0) | 1 / cnt (mode 1)     [5] cn = q / cnt (mode 0) | 1 / den (mode 1)
//   [6] dot = <dL, L> (0 when the norm was clamped: no projection)             [7] y (label; < 0 = padding)
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float rcp_hw(float x) { return __builtin_amdgcn_rcpf(x); }     // v_rcp_f32 (1 ulp), not an IEEE division

template <int MODE>
__device__ __forceinline__ void make_fast(float (&R)[8], const float* __restrict__ rec, const float y, const float it,
                                          const float inv_av, const bool live) {
#pragma unroll
  for (int e = 0; e < 8; ++e) R[e] = 0.f;
  if (live) {
    const float4 r0 = ldg4(rec), r1 = ldg4(rec + 4);           // {m, -, rn, den}, {1/cnt, q, dot, clamp}
    const float rn = r0.z;
    R[0] = it * rn * LOG2E; R[1] = -r0.x * rn * LOG2E; R[2] = rn * it * inv_av; R[3] = r0.w;
    if (MODE == 0) { R[4] = r0.w * r1.x; R[5] = r1.y * r1.x; } else { R[4] = r1.x; R[5] = 1.f / r0.w; }
    R[6] = r1.w != 0.f ? 0.f : r1.z;
  }
  R[7] = live ? y : -1.f;
}

// d loss / d S_ab * A_v-normalised, from the fast record R of row a.  same = equal labels, w = positive weight of b for a
// (mode 1), self = (a == b).  Branch-free.
template <int MODE>
__device__ __forceinline__ float g_fast(const float s, const float (&R)[8], const bool same, const bool self, const float w) {
  const float L2 = fmaf(s, R[0], R[1]);
  const float E = exp2_fast(L2);
  const float L = L2 * LN2;
  float dL;
  if (MODE == 0) {
    const float dp = -R[4] * rcp_hw(E + R[3]);
    dL = same ? dp : E * R[5];
  } else dL = fmaf(E, R[5], -w * R[4]);
  dL = self ? 0.f : dL;
  return fmaf(-L, R[6], dL) * R[2];
}

// Gsym_ij = G_ij + G_ji from S_ij and the fast records of both rows (0 if either row is padding)
template <int MODE>
__device__ __forceinline__ float gsym_entry(const float s, const float (&Ri)[8], const float (&Rj)[8], const int ig, const int jg,
                                            const float* __restrict__ mask, const int mb) {
  const float yi = Ri[7], yj = Rj[7];
  const bool same = yi == yj, self = ig == jg;
  const float wij = MODE == 1 ? pos_weight(mask, mb, ig, jg, yi, yj) : 0.f;
  const float wji = MODE == 1 ? pos_weight(mask, mb, jg, ig, yj, yi) : 0.f;
  const float vmask = (yi >= 0.f && yj >= 0.f) ? 1.f : 0.f;
  return (g_fast<MODE>(s, Ri, same, self, wij) + g_fast<MODE>(s, Rj, same, self, wji)) * vmask;
}

// ------------------------------------------------------------------------------------------------------------------
// S = X X^T, upper-triangular 64x64 tiles, every tile written as is and transposed: the whole [AP][AP] matrix (AP = A
// rounded up to 64; rows / columns >= A come out as exact zeros because their X rows are read as zeros).
// Block = strip I and a chunk of CH tiles J >= I; X_I and the streamed X_J live in LDS ([64][128] chunk images), the next
// tile is prefetched into registers while this one multiplies (v_mfma_f32_32x32x2_f32: fp32 products, as the reference).
__global__ __launch_bounds__(256, 2)
void contrast_s_kernel(const float* __restrict__ X, const int ldx, const int A, const int C, float* __restrict__ S, const int lds,
                       const int ntile, const int CH) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* XJ = reinterpret_cast<float*>(smem_raw);                 // [64][XLDL]; after the product: the transposition tile [64][65]
  float* XI = XJ + TB * XLDL;                                     // [64][XLDL]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int wm = wid >> 1, wn = wid & 1;
  int I = 0, id = blockIdx.x;
  for (;;) { const int n = (ntile - I + CH - 1) / CH; if (id < n) break; id -= n; ++I; }
  const int jbeg = I + id * CH;
  const int jend = jbeg + CH < ntile ? jbeg + CH : ntile;
  const int i0 = I * TB;
  const int nkc = (C + 127) >> 7;
  float4 st[8];
  auto load_t = [&](int T, int kc, bool valid) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int e = tid + 256 * q, r = e >> 5, k = kc * 128 + (e & 31) * 4;
      const int row = T * TB + r;
      st[q] = (valid && row < A && k < C) ? ldg4(X + (long long)row * ldx + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_t = [&](float* dstT) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int e = tid + 256 * q, r = e >> 5, k4 = (e & 31) * 4;
      *reinterpret_cast<float4*>(&dstT[r * XLDL + k4]) = st[q];
    }
  };
  if (nkc == 1) { load_t(I, 0, true); store_t(XI); }
  load_t(jbeg, 0, true);
  for (int J = jbeg; J < jend; ++J) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int kc = 0; kc < nkc; ++kc) {
      __syncthreads();                                             // readers of the previous XJ image / transposition tile are done
      store_t(XJ);
      if (nkc > 1) { load_t(I, kc, true); store_t(XI); }
      if (kc + 1 < nkc) load_t(J, kc + 1, true);
      else load_t(J + 1, 0, J + 1 < jend);                         // prefetch
      __syncthreads();
      const float* bj = &XJ[(32 * wn + l31) * XLDL + 4 * h];
      const float* ai = &XI[(32 * wm + l31) * XLDL + 4 * h];
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const float4 b = *reinterpret_cast<const float4*>(bj + 8 * g);
        const float4 a = *reinterpret_cast<const float4*>(ai + 8 * g);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
      }
    }
    // acc[r] = S[i = 32 wm + row32(r, h)][j = 32 wn + l31]: tile (I, J) straight from the registers (128-byte runs)
    const int j0 = J * TB;
    float* o = S + (long long)(i0 + 32 * wm) * lds + j0 + 32 * wn + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[(long long)row32(r, h) * lds] = acc[r];
    if (J != I) {
      // tile (J, I) = the transpose, through LDS (the X_J image is dead)
      __syncthreads();
      float* T = XJ;                                               // [64 i][65]
#pragma unroll
      for (int r = 0; r < 16; ++r) T[(32 * wm + row32(r, h)) * 65 + 32 * wn + l31] = acc[r];
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int e = tid + 256 * q, jj = e >> 6, ii = e & 63;
        S[(long long)(j0 + jj) * lds + i0 + ii] = T[ii * 65 + jj];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Row statistics: one BLOCK (4 waves) per row, the row (scaled by 1/T) and its labels in registers: thread t holds the
// columns 256 k + t, k < NK (NK = ceil(AP / 256): a compile-time bucket; <= 32 registers, so 6-8 waves per SIMD hide the
// exp / LDS latencies, and the unrolled sweeps stay small).  Padding labels become NaN: the ORDERED comparisons
// `lessgreater` (negatives) and `==` (positives) are both false for them.  The self column is a positive by label and is
// taken out by its (compile-time) register index.  Four block-wide reductions (max; norm; denominators; the positives' sums)
// go through LDS in fixed order.
constexpr int ROW_CAPW = 512;          // positives a wave can park in LDS (more: the masked fallback sweep)

template <int NRED>
__device__ __forceinline__ void block_sum4(float (&v)[NRED], float* red /* [4][NRED] */, const int lane, const int wid) {
#pragma unroll
  for (int e = 0; e < NRED; ++e) v[e] = grp_sum<64>(v[e]);
  if (lane == 0) {
#pragma unroll
    for (int e = 0; e < NRED; ++e) red[wid * NRED + e] = v[e];
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < NRED; ++e) v[e] = (red[e] + red[NRED + e]) + (red[2 * NRED + e] + red[3 * NRED + e]);
}

template <int MODE, int NK>
__global__ __launch_bounds__(256, (NK <= 12 ? 8 : (NK <= 20 ? 6 : 4)))
void contrast_row_kernel(const float* __restrict__ S, const int lds, const float* __restrict__ y, const int ldy,
                         const float* __restrict__ mask, const int mb, const int A, const float it,
                         float* __restrict__ rec, float* __restrict__ loss_row) {
  __shared__ float red_m[4], red_n[4], red_d[4 * 4], red_p[4 * 3];
  __shared__ float plist[4 * ROW_CAPW];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int i = blockIdx.x;
  const float yi = y[(long long)i * ldy];
  float* r = rec + (long long)i * REC;
  if (!(yi >= 0.f)) {                                              // padding row of the fixed-shape gather (block-uniform)
    if (tid < REC) r[tid] = 0.f;
    if (tid == 0) loss_row[i] = 0.f;
    return;
  }
  const float* srow = S + (long long)i * lds;
  const int selfk = tid == (i & 255) ? (i >> 8) : -1;              // register index of the self column in this thread
  const float qnan = __builtin_nanf("");
  float u[NK], yl[NK];
  // sweep 1: u = S/T (-3e38 on padding columns), row max (utils/loss.py:363, :179)
  float m = -3.0e38f;
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const int j = 256 * k + tid;                                   // unconditional loads (clamped index): no exec-masked regions
    const float yv = y[(long long)(j < A ? j : A - 1) * ldy];
    const float v = srow[j < lds ? j : lds - 1] * it;
    const bool ok = j < A && yv >= 0.f;
    yl[k] = ok ? yv : qnan;
    u[k] = ok ? v : -3.0e38f;
    m = fmaxf(m, u[k]);
  }
  m = grp_max<64, float>(m);
  if (lane == 0) red_m[wid] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red_m[0], red_m[1]), fmaxf(red_m[2], red_m[3]));
  // sweep 2: u <- u - m (0 on padding columns), ||u||_2, F.normalize eps 1e-12 (:366, :194)
  float n2[1] = {0.f};
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const float d = u[k] > -1.0e37f ? u[k] - m : 0.f;
    u[k] = d;
    n2[0] = fmaf(d, d, n2[0]);
  }
  block_sum4<1>(n2, red_n, lane, wid);
  const float nraw = sqrtf(n2[0]);
  const float rn = 1.f / fmaxf(nraw, 1e-12f);
  const float rn2 = rn * LOG2E;
  // sweep 3: denominators and the E.L sums the gradient's <dL, L> needs (u <- L = u rn from here on); mode 0 also parks
  // the positives' logits in the wave's LDS list (ballot + prefix count: column order, deterministic)
  float acc4[4] = {0.f, 0.f, 0.f, 0.f};                            // den, sum E L, (mode 1: cnt, sum w L | mode 0: npos of the wave, -)
  float* pl = plist + wid * ROW_CAPW;
  int npos = 0;                                                    // wave-uniform
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const float yj = yl[k];
    const float E = exp2_fast(u[k] * rn2);
    const float L = u[k] * rn;
    u[k] = L;
    if (MODE == 0) {
      const float wn_ = __builtin_islessgreater(yj, yi) ? E : 0.f;   // negatives: a different, valid label
      acc4[0] += wn_; acc4[1] = fmaf(wn_, L, acc4[1]);
      const bool pos = yj == yi && k != selfk;
      const unsigned long long bal = __ballot(pos);
      const int rank = npos + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
      if (pos && rank < ROW_CAPW) pl[rank] = L;
      npos += __popcll(bal);
    } else {
      const bool off = yj == yj && k != selfk;
      const float wo = off ? E : 0.f;
      acc4[0] += wo; acc4[1] = fmaf(wo, L, acc4[1]);
      const float w = off ? pos_weight(mask, mb, i, 256 * k + tid, yi, yj) : 0.f;
      acc4[2] += w; acc4[3] = fmaf(w, L, acc4[3]);
    }
  }
  if (MODE == 0) acc4[2] = lane == 0 ? (float)npos : 0.f;          // exact in fp32 (<= 8192)
  block_sum4<4>(acc4, red_d, lane, wid);
  const float den = acc4[0], sEL = acc4[1];
  float lp, qv = 0.f, dot, icnt;
  if (MODE == 0) {
    // sweep 4 (positives only, ~1/19 of the row): log-probabilities, q = sum_pos 1 / (E + den), sum den L / (E + den)
    float p3[3] = {0.f, 0.f, 0.f};
    auto term = [&](const float L) {
      const float d = exp2_fast(L * LOG2E) + den;
      const float id = rcp_hw(d);
      p3[0] += L - __logf(d); p3[1] += id; p3[2] = fmaf(den * id, L, p3[2]);
    };
    if (npos <= ROW_CAPW) {                                        // (the list was written by this wave; the barrier in block_sum4 ordered it)
      for (int e = lane; e < npos; e += 64) term(pl[e]);
    } else {
      // more positives than the list holds: masked sweep over the registers
#pragma unroll
      for (int k = 0; k < NK; ++k)
        if (yl[k] == yi && k != selfk) term(u[k]);
    }
    block_sum4<3>(p3, red_p, lane, wid);
    const float cnt = acc4[2];
    icnt = 1.f / cnt;                                              // cnt == 0 -> inf -> NaN loss like the reference
    lp = p3[0]; qv = p3[1];
    dot = (qv * sEL - p3[2]) * icnt;
  } else {
    const float cnt = acc4[2], swL = acc4[3];
    icnt = 1.f / cnt;
    lp = swL - cnt * __logf(den);
    dot = sEL / den - swL * icnt;
  }
  if (tid == 0) {
    r[0] = m; r[1] = 0.f; r[2] = rn; r[3] = den; r[4] = icnt; r[5] = qv; r[6] = dot; r[7] = nraw <= 1e-12f ? 1.f : 0.f;
    loss_row[i] = -lp * icnt;                                      // temperature / base_temperature = 1
  }
}

// number of valid rows (the mean's denominator): every block counts the labels itself (A floats from L2), fixed order
__device__ __forceinline__ float count_valid_rows(const float* __restrict__ y, const int ldy, const int A, float* red /* [8] */) {
  float nv = 0.f;
  for (int j = threadIdx.x; j < A; j += blockDim.x) nv += y[(long long)j * ldy] >= 0.f ? 1.f : 0.f;
  nv = dcs_wave_sum(nv);
  if (threadIdx.x < 8) red[threadIdx.x] = 0.f;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = nv;
  __syncthreads();
  return ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
}

// ------------------------------------------------------------------------------------------------------------------
// dX_I += Gsym(I, J) X_J for a 128-row strip I and a chunk of 64-column tiles J; C <= 128.  512 threads: wave w owns the
// strip rows 32 (w >> 1) .. +31 and the channels 64 (w & 1) .. +63 of the product.
//
// Two phases of work per tile -- forming Gsym(I,J) = G_ij + G_ji from the S tile and the two rows' records (~40 VALU
// instructions per element, two exponentials) and the matrix-core product -- cost about the same, and a block's waves run
// them in lock step between barriers: measured one after the other they ADD (122 us at A = 4864: 46 us of it the Gsym
// arithmetic, 45 us the MFMAs, timing experiments with either switched off).  So the kernel is software-pipelined: the
// tile images are double-buffered in LDS and, inside the product loop of tile J, every wave forms ONE EIGHTH of the Gsym
// tile of J + 1 between two groups of 8 MFMAs (the matrix pipe runs a 32x32x2 MFMA for 64 cycles; the VALU is free
// meanwhile).  Staging thread t owns rows (t >> 4) + 32 q (q < 4) and the columns 4 (t & 15) .. +3 of every S tile; slice g
// covers row q = g >> 1, column pair g & 1.  The S / X values of tile J + 2 are requested as soon as those of J + 1 have been
// consumed; the records of every tile row of the chunk are put into LDS once, in the prologue.  One barrier per tile.
constexpr int DXM = 128;               // strip rows per block
constexpr int DXT = 512;               // threads per block
constexpr int GLD = TB + 4;            // LDS row stride of the Gsym tile [128][64]
constexpr int DX_CHMAX = 8;            // tiles per chunk (their row records stay in LDS)
template <int MODE>
__global__ __launch_bounds__(DXT, 2)
void contrast_dx_kernel(const float* __restrict__ X, const int ldx, const float* __restrict__ y, const int ldy,
                        const float* __restrict__ mask, const int mb, const int A, const int C, const float it,
                        const float* __restrict__ S, const int lds, const float* __restrict__ rec, float* __restrict__ slab,
                        const int ntile, const int CH) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* GsB = reinterpret_cast<float*>(smem_raw);                // [2][128][GLD]
  float* XJB = GsB + 2 * DXM * GLD;                               // [2][64][XLDL]
  float* frJB = XJB + 2 * TB * XLDL;                              // [CH <= 8][64][8] fast records of every tile row of the chunk
  float* frI = frJB + DX_CHMAX * TB * 8;                          // [128][8] fast records of the strip rows
  float* red = frI + DXM * 8;                                     // [8]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int wm = wid >> 1, wn = wid & 1;
  const int nchunk = (ntile + CH - 1) / CH;
  const int I2 = blockIdx.x / nchunk, chunk = blockIdx.x - I2 * nchunk;
  const int jbeg = chunk * CH, jend = jbeg + CH < ntile ? jbeg + CH : ntile;
  const int i0 = I2 * DXM;
  const float inv_av = 1.f / count_valid_rows(y, ldy, A, red);
  const int sr = tid >> 4, sc4 = (tid & 15) * 4;
  auto put_record = [&](float* dst, const int row, const bool valid) {    // fast record of one row -> LDS
    const bool in = valid && row < A;
    const float yv = in ? y[(long long)row * ldy] : -1.f;
    float R[8];
    make_fast<MODE>(R, rec + (long long)(in ? row : 0) * REC, yv, it, inv_av, in && yv >= 0.f);
    *reinterpret_cast<float4*>(dst) = make_float4(R[0], R[1], R[2], R[3]);
    *reinterpret_cast<float4*>(dst + 4) = make_float4(R[4], R[5], R[6], R[7]);
  };
  if (tid < DXM) put_record(frI + tid * 8, i0 + tid, true);
  for (int e = tid; e < (jend - jbeg) * TB; e += DXT) put_record(frJB + e * 8, jbeg * TB + e, true);
  float4 sv[4];                                                   // S values (raw) of the tile being staged: 4 rows x 4 columns
  float4 xv[4];                                                   // its X_J rows
  // S is [AP][AP] and AP is a multiple of 64, but a strip of 128 rows may reach past AP: clamp the row (its record is padding)
  long long soff[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { const int r = i0 + sr + 32 * q; soff[q] = (long long)(r < lds ? r : lds - 1) * lds + sc4; }
  auto load_s = [&](const int q, const int J) {
    sv[q] = J < jend ? ldg4(S + soff[q] + J * TB) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  auto load_x = [&](const int q, const int J) {
    const int e = tid + DXT * q, r = e >> 5, k = (e & 31) * 4;
    const int row = J * TB + r;
    xv[q] = (J < jend && row < A && k < C) ? ldg4(X + (long long)row * ldx + k) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  auto store_x = [&](const int q, float* XJ) {
    const int e = tid + DXT * q, r = e >> 5, k4 = (e & 31) * 4;
    *reinterpret_cast<float4*>(&XJ[r * XLDL + k4]) = xv[q];
  };
  // the records the staging needs live in REGISTERS while it runs between the MFMAs (an LDS read there would put a
  // s_waitcnt into the MFMA stream): the thread's 4 strip rows for the whole block, its 4 columns per tile
  float Ri[4][8], Rc[4][8];
  auto load_rc = [&](const float* frJ) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float4 a = *reinterpret_cast<const float4*>(frJ + (sc4 + c) * 8), b = *reinterpret_cast<const float4*>(frJ + (sc4 + c) * 8 + 4);
      Rc[c][0] = a.x; Rc[c][1] = a.y; Rc[c][2] = a.z; Rc[c][3] = a.w; Rc[c][4] = b.x; Rc[c][5] = b.y; Rc[c][6] = b.z; Rc[c][7] = b.w;
    }
  };
  // one eighth of the Gsym tile of tile J: row q = g >> 1, columns sc4 + 2 (g & 1) + {0, 1}
  auto stage_slice = [&](const int g, const int J, float* Gs) {
    const int q = g >> 1, c0 = 2 * (g & 1);
    const int ig = i0 + sr + 32 * q, jg = J * TB + sc4 + c0;
    const float o0 = gsym_entry<MODE>(comp(sv[q], c0), Ri[q], Rc[c0], ig, jg, mask, mb);
    const float o1 = gsym_entry<MODE>(comp(sv[q], c0 + 1), Ri[q], Rc[c0 + 1], ig, jg + 1, mask, mb);
    *reinterpret_cast<float2*>(&Gs[(sr + 32 * q) * GLD + sc4 + c0]) = make_float2(o0, o1);
  };
  f32x16 acc[2];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
  // prologue: tile jbeg is staged on its own, then the S / X values of jbeg + 1 are requested
#pragma unroll
  for (int q = 0; q < 4; ++q) { load_s(q, jbeg); load_x(q, jbeg); }
  __syncthreads();                                                 // the records are in LDS
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float* rr = frI + (sr + 32 * q) * 8;
    const float4 a = *reinterpret_cast<const float4*>(rr), b = *reinterpret_cast<const float4*>(rr + 4);
    Ri[q][0] = a.x; Ri[q][1] = a.y; Ri[q][2] = a.z; Ri[q][3] = a.w; Ri[q][4] = b.x; Ri[q][5] = b.y; Ri[q][6] = b.z; Ri[q][7] = b.w;
  }
  load_rc(frJB);
#pragma unroll
  for (int g = 0; g < 8; ++g) stage_slice(g, jbeg, GsB);
#pragma unroll
  for (int q = 0; q < 4; ++q) { store_x(q, XJB); load_s(q, jbeg + 1); load_x(q, jbeg + 1); }
  __syncthreads();
  for (int J = jbeg; J < jend; ++J) {
    const int b = (J - jbeg) & 1;
    const float* Gs = GsB + b * (DXM * GLD);
    const float* XJ = XJB + b * (TB * XLDL);
    float* GsN = GsB + (b ^ 1) * (DXM * GLD);
    float* XJN = XJB + (b ^ 1) * (TB * XLDL);
    const bool more = J + 1 < jend;                                // block-uniform
    // dX[i = 32 wm + ..][c = 64 wn + 32 b + ..] += sum_k Gsym[i][k] X_J[k][c]
    // A fragment: lane (i = l31, k = 8 g + 4 h + e); B fragment: lane (c = l31, k = 8 g + 4 h + e)
    // operands of one group of 8 MFMAs (k = 8 g .. 8 g + 7): read one group AHEAD of their use, so that no MFMA waits for LDS
    float4 af[2];
    float xb[2][8];
    auto load_ops = [&](const int g, const int slot) {
      af[slot] = *reinterpret_cast<const float4*>(&Gs[(32 * wm + l31) * GLD + 8 * g + 4 * h]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float* p = &XJ[(8 * g + 4 * h + e) * XLDL + 64 * wn + l31];
        xb[slot][2 * e] = p[0]; xb[slot][2 * e + 1] = p[32];
      }
    };
    auto mfma_group = [&](const int g) {
      const int slot = g & 1;
      if (g + 1 < 8) load_ops(g + 1, slot ^ 1);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb)
          acc[bb] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(af[slot], e), xb[slot][2 * e + bb], acc[bb], 0, 0, 0);
    };
    load_ops(0, 0);
    if (more) {
      // groups 0..3: matrix cores only (the S / X values of tile J + 1 were requested at the end of the last step: they
      // are ~4 MFMA groups old when the staging first touches them, so its s_waitcnt finds them delivered)
#pragma unroll
      for (int g = 0; g < 4; ++g) mfma_group(g);
      load_rc(frJB + (J + 1 - jbeg) * (TB * 8));
#pragma unroll
      for (int g = 4; g < 8; ++g) {
        mfma_group(g);
        // behind those 8 MFMAs: one quarter of the next tile's Gsym and of its X rows
        stage_slice(2 * (g - 4), J + 1, GsN);
        stage_slice(2 * (g - 4) + 1, J + 1, GsN);
        store_x(g - 4, XJN);
        // issue order: one MFMA (64 cycles of the matrix pipe), then the independent VALU instructions
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 20, 0);
        }
      }
      // everything staged: request the values of tile J + 2 (no load is in flight while older values are being read)
#pragma unroll
      for (int q = 0; q < 4; ++q) { load_s(q, J + 2); load_x(q, J + 2); }
    } else {
#pragma unroll
      for (int g = 0; g < 8; ++g) mfma_group(g);
    }
    __syncthreads();
  }
  // every wave owns a disjoint 32 x 64 block of the strip's partial dX: straight into this chunk's slab
  float* o = slab + (long long)chunk * A * C;
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int c = 64 * wn + 32 * b + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = i0 + 32 * wm + row32(r, h);
      if (i < A && c < C) o[(long long)i * C + c] = acc[b][r];
    }
  }
}

// Gsym written out (C > 128): one block per 64 x 64 tile of the full matrix, elementwise from S and the records.
template <int MODE>
__global__ __launch_bounds__(256)
void contrast_gsym_kernel(const float* __restrict__ y, const int ldy, const float* __restrict__ mask, const int mb, const int A,
                          const float it, const float* __restrict__ S, const int lds, const float* __restrict__ rec,
                          float* __restrict__ gsym, const int ldg, const int ntile) {
  __shared__ __attribute__((aligned(16))) float frI[TB * 8], frJ[TB * 8];
  __shared__ float red[8];
  const int tid = threadIdx.x;
  const int I = blockIdx.x / ntile, J = blockIdx.x - I * ntile;
  const float inv_av = 1.f / count_valid_rows(y, ldy, A, red);
  if (tid < 2 * TB) {
    const int r = (tid < TB ? I : J) * TB + (tid & (TB - 1));
    const bool in = r < A;
    const float yv = in ? y[(long long)r * ldy] : -1.f;
    float R[8];
    make_fast<MODE>(R, rec + (long long)(in ? r : 0) * REC, yv, it, inv_av, in && yv >= 0.f);
    float* d = (tid < TB ? frI : frJ) + (tid & (TB - 1)) * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) d[e] = R[e];
  }
  __syncthreads();
  for (int e = tid; e < TB * TB; e += 256) {
    const int il = e >> 6, jl = e & 63, ig = I * TB + il, jg = J * TB + jl;
    if (ig >= A || jg >= A) continue;
    float Ri[8], Rj[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { Ri[k] = frI[il * 8 + k]; Rj[k] = frJ[jl * 8 + k]; }
    gsym[(long long)ig * ldg + jg] = gsym_entry<MODE>(S[(long long)ig * lds + jg], Ri, Rj, ig, jg, mask, mb);
  }
}

// dX[i][c] = sum over the chunk slabs (fixed order); block 0 also reduces the loss.
__global__ __launch_bounds__(256)
void contrast_finish_kernel(const float* __restrict__ slab, int nchunk, long long n, float* __restrict__ dX, int C, int lddx,
                            const float* __restrict__ loss_row, const float* __restrict__ y, int ldy, int A,
                            float* __restrict__ loss) {
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
    __shared__ float red[8];
    const float av = count_valid_rows(y, ldy, A, red);
    double s = 0.0;
    for (int j = threadIdx.x; j < A; j += 256) s += (double)loss_row[j];
    s = dcs_wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = (float)(((sm[0] + sm[1]) + (sm[2] + sm[3])) / (double)av);
  }
}

constexpr int LARGE_MAX = 32 * 256;    // rows the row kernel's largest register bucket holds

inline int large_ws_floats(int A, int C, int64_t* out) {
  const int64_t AP = (A + TB - 1) / TB * TB;
  // rec, loss_row (+pad), S [AP][AP], slabs [nchunk <= 16][A][C]
  *out = (int64_t)A * (REC + 1) + 64 + AP * AP + (C <= 128 ? 16ll * A * C : 0) + 64;
  return 0;
}

template <int MODE>
int launch_rows(const float* S, int lds, const float* y, int ldy, const float* mask, int mb, int A, float it, float* rec,
                float* loss_row, hipStream_t s) {
  const int nk = (A + 255) / 256;
  const dim3 grid(A), block(256);
#define DCS_ROWS(NK_) hipLaunchKernelGGL((contrast_row_kernel<MODE, NK_>), grid, block, 0, s, S, lds, y, ldy, mask, mb, A, it, rec, loss_row)
  if (nk <= 6) DCS_ROWS(6);
  else if (nk <= 12) DCS_ROWS(12);
  else if (nk <= 20) DCS_ROWS(20);
  else DCS_ROWS(32);
#undef DCS_ROWS
  return DCS_OK;
}

inline int launch_large(const float* X, int ldx, const float* y, int ldy, const float* mask, int mb, int A, int C, int mode,
                        float it, float* loss, float* dX, int lddx, float* gsym, int ldg, float* ws, hipStream_t s) {
  const int ntile = (A + TB - 1) / TB;
  const int AP = ntile * TB;
  float* rec = ws;
  float* loss_row = rec + (size_t)A * REC;
  float* S = loss_row + A;
  S += (4 - ((S - ws) & 3)) & 3;                                   // 16-byte aligned rows (AP % 64 == 0)
  float* slab = S + (size_t)AP * AP;
  // S: upper-triangular tiles, ~3 blocks per CU.  chunks per strip I = ceil((ntile - I) / CH)
  int CH = 1;
  for (; CH < 16; ++CH) {
    long long nb = 0;
    for (int I = 0; I < ntile; ++I) nb += (ntile - I + CH - 1) / CH;
    if (nb <= 768) break;
  }
  long long nb_s = 0;
  for (int I = 0; I < ntile; ++I) nb_s += (ntile - I + CH - 1) / CH;
  const size_t sh_s = (size_t)2 * TB * XLDL * sizeof(float);
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(contrast_s_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh_s) != hipSuccess)
    return DCS_E_LAUNCH;
  hipLaunchKernelGGL(contrast_s_kernel, dim3((unsigned)nb_s), dim3(256), sh_s, s, X, ldx, A, C, S, AP, ntile, CH);
  const int rc = mode == 0 ? launch_rows<0>(S, AP, y, ldy, mask, mb, A, it, rec, loss_row, s)
                           : launch_rows<1>(S, AP, y, ldy, mask, mb, A, it, rec, loss_row, s);
  if (rc != DCS_OK) return rc;
  if (gsym) {
    const dim3 grid((unsigned)(ntile * ntile));
    if (mode == 0) hipLaunchKernelGGL(contrast_gsym_kernel<0>, grid, dim3(256), 0, s, y, ldy, mask, mb, A, it, S, AP, rec, gsym, ldg, ntile);
    else hipLaunchKernelGGL(contrast_gsym_kernel<1>, grid, dim3(256), 0, s, y, ldy, mask, mb, A, it, S, AP, rec, gsym, ldg, ntile);
    hipLaunchKernelGGL(contrast_finish_kernel, dim3(1), dim3(256), 0, s, (const float*)nullptr, 0, 0ll, dX, C, lddx, loss_row, y, ldy, A, loss);
    DCS_LAUNCH_RET();
  }
  // dX: 128-row strips x chunks of tiles, <= 16 chunks (slabs), ~2 blocks of 8 waves per CU
  const int nstrip = (A + DXM - 1) / DXM;
  int CH4 = (ntile * nstrip + 511) / 512;                          // one 8-wave block per CU: two rounds of blocks
  if (CH4 < (ntile + 15) / 16) CH4 = (ntile + 15) / 16;           // <= 16 slabs
  if (CH4 < 1) CH4 = 1;
  if (CH4 > DX_CHMAX) return DCS_E_UNSUPPORTED;                    // cannot happen for A <= LARGE_MAX (128 tiles / 16)
  const int nchunk = (ntile + CH4 - 1) / CH4;
  const size_t sh_d = (size_t)(2 * DXM * GLD + 2 * TB * XLDL + DX_CHMAX * TB * 8 + DXM * 8 + 8) * sizeof(float);
  auto kd0 = contrast_dx_kernel<0>;
  auto kd1 = contrast_dx_kernel<1>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(mode == 0 ? kd0 : kd1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh_d) != hipSuccess)
    return DCS_E_LAUNCH;
  const dim3 gd((unsigned)(nstrip * nchunk));
  if (mode == 0) hipLaunchKernelGGL(kd0, gd, dim3(DXT), sh_d, s, X, ldx, y, ldy, mask, mb, A, C, it, S, AP, rec, slab, ntile, CH4);
  else hipLaunchKernelGGL(kd1, gd, dim3(DXT), sh_d, s, X, ldx, y, ldy, mask, mb, A, C, it, S, AP, rec, slab, ntile, CH4);
  const long long n = (long long)A * C;
  hipLaunchKernelGGL(contrast_finish_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, slab, nchunk, n, dX, C, lddx,
                     loss_row, y, ldy, A, loss);
  DCS_LAUNCH_RET();
}
