// Convolution as implicit GEMM on the fp32 matrix cores of gfx950
// (v_mfma_f32_32x32x2_f32: exact fp32, 64 FLOP/clk/SIMD).
//
// One gather kernel serves nn.Conv2d forward AND its data gradient (see DcsConvGeom in
// include/dcs_hip.h); a second kernel computes the weight gradient split over pixel ranges.
// They replace the ATen/cuDNN calls behind network/backbone/resnet_pyramid.py:23-25,:110-112,:139
// and network/utils.py:46-47 in the reference.
//
// Tiling (gather kernel): block = 256 threads = 4 waves, BM = 128 output pixels x BN output
// channels, K consumed in chunks of 32 input channels of one filter tap.  Operands are staged
// through registers into LDS rows of 36 floats (32 + 4 pad): lane (row r, half h) reads its four
// k values kk+4h .. kk+4h+3 with one conflict-free ds_read_b128 and feeds them to four
// MFMA 32x32x2 issues (any k permutation is legal as long as A and B agree).
// Global loads of chunk i+1 are issued before the 64 MFMAs of chunk i and written to the other
// LDS buffer afterwards: one barrier per chunk.
#include "dcs_common.h"
#include <cstdlib>

#include "conv_shared.h"

namespace {

template <int BN, bool STEM, int BKT, int BM = 128>
__global__ __launch_bounds__(256, (BKT == 16 && BM * BN <= 128 * 128) ? 3 : 2)
void conv_gather_kernel(const float* __restrict__ src, const float* __restrict__ wgt,
                        const float* __restrict__ bias, float* __restrict__ dst,
                        const DcsConvGeom g, const int accumulate, const int ntiles, float* __restrict__ stats,
                        const int cps, const long long slab_stride, const BnBwdEpi bnb, const float* __restrict__ pro) {
  constexpr int WN = (BN >= 128 || (BN == 64 && BM == 128)) ? 2 : 1;
  constexpr int WM = 4 / WN;
  constexpr int TM = BM / (WM * 32);
  constexpr int TN = BN / (WN * 32);
  constexpr int LDKT = BKT + 4;          // LDS row stride: (BKT+4)*4 B is an odd number of 16-B slots (9 or 5)
  constexpr int C4 = BKT / 4;            // float4 per staged row
  constexpr int RPP = 256 / C4;          // rows staged per slot
  constexpr int NA = BM / RPP;           // A slots per thread
  constexpr int BROWS = BN / RPP;        // B slots per thread
  constexpr int NG = BKT / 8;            // MFMA k8 groups per chunk
  static_assert(BROWS >= 1 && (NG == 4 || NG == 2) && (BM == 128 || BM == 256), "unsupported tile");

  // one array: As[2][BM*LDKT] followed by Bs[2][BN*LDKT]; the epilogue reuses it as per-wave staging tiles
  __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * LDKT];
  float (*As)[BM * LDKT] = reinterpret_cast<float (*)[BM * LDKT]>(smem);
  float (*Bs)[BN * LDKT] = reinterpret_cast<float (*)[BN * LDKT]>(smem + 2 * BM * LDKT);
  __shared__ long long rowoff[BM];
  __shared__ int s_oy[DCS_MAX_TAPS], s_ox[DCS_MAX_TAPS], s_wo[DCS_MAX_TAPS], s_to[DCS_MAX_TAPS];
  // BatchNorm + ReLU prologue (network/utils.py:35-49 _BNReluConv, resnet_pyramid.py:71-89 conv2 of a BasicBlock): the
  // source tensor is the BatchNorm INPUT; scale / shift / max(.,0) are applied to the A operand on its way from the
  // staging registers to LDS, so the activated tensor is never written.  pro = [scale(K), shift(K)] (the head of the
  // BatchNorm record).  Zero padding is padding of the ACTIVATED tensor: out-of-range taps stay exactly 0.
  __shared__ __attribute__((aligned(16))) float s_pro[STEM ? 4 : 2 * DCS_PRO_MAXK];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int wm = wid / WN, wn = wid % WN;
  const int lrow = tid / C4, lcol4 = tid % C4;
  const bool has_pro = !STEM && pro != nullptr;
  if (has_pro)
    for (int e = tid; e < 2 * g.K; e += 256) s_pro[(e < g.K ? 0 : DCS_PRO_MAXK - g.K) + e] = pro[e];

  const int bid = dcs_xcd_remap(blockIdx.x, gridDim.x);
  const int ntile = bid % ntiles, mtile = bid / ntiles;
  const int co0 = ntile * BN;
  const unsigned m0 = (unsigned)mtile * BM;
  const unsigned TYX = (unsigned)g.TY * (unsigned)g.TX;
  const unsigned M = (unsigned)g.N * TYX;

  if (tid < BM) {
    const unsigned m = m0 + tid;
    long long off = -1;
    if (m < M) {
      const int n = (int)(m / TYX);
      const int rem = (int)(m - (unsigned)n * TYX);
      const int ty = rem / g.TX, tx = rem - ty * g.TX;
      off = (((long long)n * g.DH + (ty * g.dsy + g.dy0)) * g.DW + (tx * g.dsx + g.dx0)) * g.dst_cstride;
    }
    rowoff[tid] = off;
  }

  // Source addressing is 32-bit relative to the first image this block touches (a block spans 128 pixels).
  const int n0 = (int)(m0 / TYX);
  const long long img_elems = (long long)g.SH * g.SW * g.src_cstride;
  const __amdgpu_buffer_rsrc_t rsA = make_rsrc(src + (long long)n0 * img_elems, ((long long)g.N - n0) * img_elems * 4);
  const __amdgpu_buffer_rsrc_t rsB = make_rsrc(wgt, (long long)g.Cout * g.wstride * 4);

  if (tid < g.ntaps) {
    s_oy[tid] = g.offy[tid]; s_ox[tid] = g.offx[tid]; s_wo[tid] = g.wofs[tid];
    s_to[tid] = (g.offy[tid] * g.SW + g.offx[tid]) * g.src_cstride;
  }

  int r_base[NA], r_y[NA], r_x[NA];  // element offset of the row's (tap 0,0) source pixel; invalid rows: r_y = -2^20
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const unsigned m = m0 + lrow + RPP * i;
    const bool ok = m < M;
    const unsigned mm = ok ? m : m0;
    const int n = (int)(mm / TYX);
    const int rem = (int)(mm - (unsigned)n * TYX);
    const int ty = rem / g.TX, tx = rem - ty * g.TX;
    r_y[i] = ok ? ty * g.sy : -(1 << 20);
    r_x[i] = tx * g.sx;
    r_base[i] = (((n - n0) * g.SH + ty * g.sy) * g.SW + tx * g.sx) * g.src_cstride;
  }
  int b_base[BROWS];                 // element offset of the weight row, or -1 beyond Cout
#pragma unroll
  for (int i = 0; i < BROWS; ++i) {
    const int co = co0 + lrow + RPP * i;
    b_base[i] = co < g.Cout ? co * g.wstride : -1;
  }
  __syncthreads();

  const int kch = STEM ? 1 : (g.K + BKT - 1) / BKT;
  const int nch = g.ntaps * kch;
  // split-K launches (grid.y > 1): this block reduces chunks [cbeg, cend) only and writes its partial tile to slab
  // blockIdx.y (dst + blockIdx.y * slab_stride); a fixed-order reduce kernel sums the slabs.  Few-tile launches (deep
  // layers of small inputs) otherwise leave most CUs idle behind a 100+-chunk serial loop.
  const int cbeg = (int)blockIdx.y * cps < nch ? (int)blockIdx.y * cps : nch;
  const int cend = cbeg + cps < nch ? cbeg + cps : nch;
  dst += (long long)blockIdx.y * slab_stride;

  // Register staging slots: 0..NA-1 = the A rows of this thread, NA.. = its B rows.
  constexpr int NSLOT = NA + BROWS;
  float4 rs[NSLOT];

  // chunk -> (tap, channel offset); per-chunk scalars are refreshed once per chunk by set_chunk()
  int c_oy = 0, c_ox = 0, c_wo = 0, c_to = 0, c_kc = 0;
  bool c_kvalid = true;
  auto set_chunk = [&](int ch) {
    const int t = ch / kch;
    const int c0 = (ch - t * kch) * BKT;
    c_oy = s_oy[t]; c_ox = s_ox[t]; c_wo = s_wo[t]; c_to = s_to[t];
    c_kc = c0 + lcol4 * 4;
    c_kvalid = STEM ? true : c_kc < g.K;
  };
  float lim[NA];                     // prologue only: +inf if the A slot's registers hold a real (in-range) element, else 0
  float4 p_sc = zero4(), p_sh = zero4();   // prologue scale / shift of the channels of the chunk held in the registers
  auto load_pro = [&](int kc) {
    const int kq = kc < g.K ? kc : 0;
    p_sc = ld4(&s_pro[kq]); p_sh = ld4(&s_pro[DCS_PRO_MAXK + kq]);
  };
  auto load_slot = [&](int sl) {
    if (sl < NA) {
      const int i = sl;
      if (!STEM) {
        const bool ok = c_kvalid && (unsigned)(r_y[i] + c_oy) < (unsigned)g.SH && (unsigned)(r_x[i] + c_ox) < (unsigned)g.SW;
        rs[sl] = bld4(rsA, ok ? (unsigned)(r_base[i] + c_to + c_kc) * 4u : OOB);
        if (has_pro) lim[sl] = ok ? __builtin_inff() : 0.f;
      } else {
        const bool ok = lcol4 < 7 && (unsigned)(r_y[i] + c_oy) < (unsigned)g.SH &&
                        (unsigned)(r_x[i] + c_ox + lcol4) < (unsigned)g.SW;
        rs[sl] = bld4(rsA, ok ? (unsigned)(r_base[i] + c_to + lcol4 * 4) * 4u : OOB);
      }
    } else {
      const int i = sl - NA;
      rs[sl] = bld4(rsB, (c_kvalid && b_base[i] >= 0) ? (unsigned)(b_base[i] + c_wo + c_kc) * 4u : OOB);
    }
  };
  auto store_slot = [&](int sl, int buf) {
    if (sl < NA) {
      float4 v = rs[sl];
      if (has_pro) v = pro_apply(v, p_sc, p_sh, lim[sl]);
      *reinterpret_cast<float4*>(&As[buf][(lrow + RPP * sl) * LDKT + lcol4 * 4]) = v;
    } else *reinterpret_cast<float4*>(&Bs[buf][(lrow + RPP * (sl - NA)) * LDKT + lcol4 * 4]) = rs[sl];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // Software pipeline (register-staged, write-after-barrier): registers hold chunk ch+1 while chunk ch is
  // computed.  After the first MFMA group of iteration ch each slot is written to the other LDS buffer (last read
  // in iteration ch-1, before the barrier) and immediately re-loaded with chunk ch+2, one slot per pair of MFMAs,
  // so LDS writes, address arithmetic and load issue sit in the shadow of the matrix pipe and every global load
  // has a full iteration (>= 48 MFMAs per wave) to land.  The tail iterations re-load the last chunk (harmless).
  if (has_pro) __syncthreads();                       // s_pro visible before the first activated store
  set_chunk(cbeg < nch ? cbeg : nch - 1);
#pragma unroll
  for (int sl = 0; sl < NSLOT; ++sl) load_slot(sl);
  if (has_pro) load_pro(c_kc);
#pragma unroll
  for (int sl = 0; sl < NSLOT; ++sl) store_slot(sl, 0);
  set_chunk(cbeg + 1 < cend ? cbeg + 1 : (cbeg < nch ? cbeg : nch - 1));
#pragma unroll
  for (int sl = 0; sl < NSLOT; ++sl) load_slot(sl);
  __syncthreads();

  constexpr int G = TM * TN * 4;               // MFMAs per k8 group
  float av[2][TM][4], bv[2][TN][4];
  auto frag_load = [&](int set, const float* Ab, const float* Bb, int kk) {
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      const float4 v = ld4(Ab + a * 32 * LDKT + kk);
      av[set][a][0] = v.x; av[set][a][1] = v.y; av[set][a][2] = v.z; av[set][a][3] = v.w;
    }
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const float4 v = ld4(Bb + b * 32 * LDKT + kk);
      bv[set][b][0] = v.x; bv[set][b][1] = v.y; bv[set][b][2] = v.z; bv[set][b][3] = v.w;
    }
  };
  auto mfma_range = [&](int set, int lo, int hi) {     // MFMAs lo..hi-1 of a group, order j, a, b
#pragma unroll
    for (int q = 0; q < G; ++q) {
      if (q < lo || q >= hi) continue;
      const int j = q / (TM * TN), a = (q / TN) % TM, b = q % TN;
      acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[set][a][j], bv[set][b][j], acc[a][b], 0, 0, 0);
    }
  };

  for (int ch = cbeg; ch < cend; ++ch) {
    const int buf = (ch - cbeg) & 1;
    const float* Ab = &As[buf][(wm * TM * 32 + l31) * LDKT + 4 * h];
    const float* Bb = &Bs[buf][(wn * TN * 32 + l31) * LDKT + 4 * h];
    frag_load(0, Ab, Bb, 0);
    frag_load(1, Ab, Bb, 8);
    mfma_range(0, 0, G);
    if (has_pro) load_pro(c_kc);                          // the registers hold chunk ch + 1: its channels' scale / shift
    set_chunk(ch + 2 < cend ? ch + 2 : cend - 1);
    if (NG == 4) frag_load(0, Ab, Bb, 16);
#pragma unroll
    for (int sl = 0; sl < NSLOT; ++sl) {
      store_slot(sl, buf ^ 1);
      load_slot(sl);
      mfma_range(1, sl * G / NSLOT, (sl + 1) * G / NSLOT);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (NG == 4) {
      frag_load(1, Ab, Bb, 24);
      mfma_range(0, 0, G);
      mfma_range(1, 0, G);
    }
    __syncthreads();
  }

  conv_epilogue<BM, BN, TM, TN, WM, 2 * (BM + BN) * LDKT>(acc, smem, rowoff, bias, dst, g.dst_cstride, g.Cout, co0,
                                                          accumulate, stats, mtile, M, wm, wn, bnb, m0 + BM <= M);
}

// ------------------------------------------------------------------------------------------------
// weight gradient: dW[co][tap][ci] = sum_m dy[m][co] * src[gather(m,tap)][ci]
// Block = one (tap, co tile, ci tile) and one pixel range (split); pixels are the GEMM K dimension,
// staged 32 at a time as [pixel][channel] rows, read column-wise by conflict-free ds_read_b32.
template <int BT, int CH = 32>
__global__ __launch_bounds__(256, (BT == 128 && CH == 16) ? 4 : 2)
void conv_wgrad_kernel(const float* __restrict__ src, const float* __restrict__ dy, float* __restrict__ slab,
                       const DcsConvGeom g, const int dy_cstride, const int split0, const long long mps,
                       const int ciT, const float* __restrict__ pro) {
  constexpr int T = BT / 64;          // 32x32 tiles per wave per dim
  constexpr int C4 = BT / 4;          // float4 per staged row
  constexpr int RP = 256 / C4;        // rows per load pass
  constexpr int NP = CH / RP;         // passes per chunk (CH = pixels per chunk: 32, or 16 -> half the LDS)
  static_assert(NP >= 1, "chunk too small for this tile");
  __shared__ __attribute__((aligned(16))) float Ds[2][CH * BT];
  __shared__ __attribute__((aligned(16))) float Xs[2][CH * BT];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int wm = wid >> 1, wn = wid & 1;
  const int lcol4 = tid % C4, lrow = tid / C4;

  const int bx = blockIdx.x;
  const int t = bx % g.ntaps;
  const int rest = bx / g.ntaps;
  const int ciTile = rest % ciT, coTile = rest / ciT;
  const int co0 = coTile * BT, ci0 = ciTile * BT;
  const int split = blockIdx.y;

  const long long TYX = (long long)g.TY * g.TX;
  const long long M = (long long)g.N * TYX;
  const long long mbeg = (long long)split * mps;
  const long long mend = mbeg + mps < M ? mbeg + mps : M;
  const int Keff = g.stem ? 32 : g.K;
  const int oy = g.offy[t], ox = g.offx[t], wo = g.wofs[t];

  // per-row-slot pixel coordinates, advanced incrementally (no division in the loop)
  int p_n[NP], p_ty[NP], p_tx[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    long long m = mbeg + lrow + RP * i;
    if (m >= M) m = M - 1;
    const int n = (int)(m / TYX);
    const int rem = (int)(m - (long long)n * TYX);
    p_n[i] = n; p_ty[i] = rem / g.TX; p_tx[i] = rem - p_ty[i] * g.TX;
  }

  float4 rd[NP], rx[NP];
  const int kc = ci0 + lcol4 * 4;
  const int cc = co0 + lcol4 * 4;
  // BatchNorm + ReLU prologue on the source operand (see conv_gather_kernel): a thread's channels never change
  const bool has_pro = pro != nullptr && !g.stem;
  float4 p_sc = zero4(), p_sh = zero4();
  if (has_pro && kc < g.K) { p_sc = ld4(pro + kc); p_sh = ld4(pro + g.K + kc); }
  float lim[NP];                     // +inf where rx[i] holds a real element, 0 where it is padding (prologue only)
  // 32-bit buffer addressing relative to this split's first dy row / first source image
  const int n0 = (int)(mbeg / TYX);
  const long long img_elems = (long long)g.SH * g.SW * g.src_cstride;
  const __amdgpu_buffer_rsrc_t rsX = make_rsrc(src + (long long)n0 * img_elems, ((long long)g.N - n0) * img_elems * 4);
  const __amdgpu_buffer_rsrc_t rsD = make_rsrc(dy + mbeg * dy_cstride, (mend - mbeg) * (long long)dy_cstride * 4);
  const bool ccok = cc < g.Cout;

  auto load_chunk = [&](long long mc) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int mr = (int)(mc - mbeg) + lrow + RP * i;          // rows >= mend - mbeg fall outside rsD -> 0
      rd[i] = bld4(rsD, ccok ? (unsigned)(mr * dy_cstride + cc) * 4u : OOB);
      const bool mok = mbeg + mr < mend;
      const int iy = p_ty[i] * g.sy + oy;
      int ix = p_tx[i] * g.sx + ox;
      bool ok = mok && (unsigned)iy < (unsigned)g.SH;
      int off;
      if (!g.stem) {
        ok = ok && kc < g.K && (unsigned)ix < (unsigned)g.SW;
        off = (((p_n[i] - n0) * g.SH + iy) * g.SW + ix) * g.src_cstride + kc;
      } else {
        ix += lcol4;
        ok = ok && lcol4 < 7 && (unsigned)ix < (unsigned)g.SW;
        off = (((p_n[i] - n0) * g.SH + iy) * g.SW + ix) * g.src_cstride;
      }
      rx[i] = bld4(rsX, ok ? (unsigned)off * 4u : OOB);
      if (has_pro) lim[i] = ok ? __builtin_inff() : 0.f;
      // advance this slot by one chunk of pixels
      p_tx[i] += CH;
      while (p_tx[i] >= g.TX) { p_tx[i] -= g.TX; p_ty[i] += 1; }
      while (p_ty[i] >= g.TY) { p_ty[i] -= g.TY; p_n[i] += 1; }
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      *reinterpret_cast<float4*>(&Ds[buf][(lrow + RP * i) * BT + lcol4 * 4]) = rd[i];
      float4 v = rx[i];
      if (has_pro) v = pro_apply(v, p_sc, p_sh, lim[i]);
      *reinterpret_cast<float4*>(&Xs[buf][(lrow + RP * i) * BT + lcol4 * 4]) = v;
    }
  };

  f32x16 acc[T][T];
#pragma unroll
  for (int a = 0; a < T; ++a)
#pragma unroll
    for (int b = 0; b < T; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const long long nch = mend > mbeg ? (mend - mbeg + CH - 1) / CH : 0;
  if (nch > 0) {
    load_chunk(mbeg);
    store_chunk(0);
  }
  __syncthreads();
  for (long long ch = 0; ch < nch; ++ch) {
    const int buf = (int)(ch & 1);
    const bool more = ch + 1 < nch;
    if (more) load_chunk(mbeg + (ch + 1) * CH);
    const float* Db = &Ds[buf][wm * (BT / 2) + l31];
    const float* Xb = &Xs[buf][wn * (BT / 2) + l31];
#pragma unroll
    for (int kk = 0; kk < CH; kk += 8) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = kk + 4 * h + j;
        float av[T], bv[T];
#pragma unroll
        for (int a = 0; a < T; ++a) av[a] = Db[row * BT + a * 32];
#pragma unroll
        for (int b = 0; b < T; ++b) bv[b] = Xb[row * BT + b * 32];
#pragma unroll
        for (int a = 0; a < T; ++a)
#pragma unroll
          for (int b = 0; b < T; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
      }
    }
    if (more) store_chunk(buf ^ 1);
    __syncthreads();
  }

  float* out = slab + (long long)(split0 + split) * g.Cout * g.wstride;
#pragma unroll
  for (int a = 0; a < T; ++a)
#pragma unroll
    for (int b = 0; b < T; ++b) {
      const int ci = ci0 + wn * (BT / 2) + b * 32 + l31;
      if (ci >= Keff) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wm * (BT / 2) + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (co < g.Cout) out[(long long)co * g.wstride + wo + ci] = acc[a][b][r];
      }
    }
}

// ------------------------------------------------------------------------------------------------
// weight gradient of 3x3 / stride 1 / pad 1 convolutions, ALL NINE TAPS per block.
// Block = one 64(co) x 64(ci) tile and one pixel range; a chunk is 32 consecutive output pixels of one image
// row (requires TX % 32 == 0).  Per chunk the block stages dy[32][64] once and the 3 x 34 pixel input halo
// x[3][34][64] once and feeds 9 taps x 16 MFMAs per wave from them (tap (r,s), pixel p reads halo row r,
// column p + s): 34 KB staged per 144 MFMAs per wave, a 4x better ratio than the per-tap kernel.
// Same write-after-barrier register pipeline as conv_gather_kernel.
__global__ __launch_bounds__(256, 2)
void conv_wgrad3x3_kernel(const float* __restrict__ src, const float* __restrict__ dy, float* __restrict__ slab,
                          const DcsConvGeom g, const int dy_cstride, const int split0, const int cps /*chunks per split*/,
                          const int ciT, const float* __restrict__ pro) {
  constexpr int HW_ = 34;                 // halo width in pixels
  constexpr int XROWS = 3 * HW_;          // 102 staged input rows
  constexpr int NSD = 2, NSX = 7, NSLOT = NSD + NSX;
  __shared__ __attribute__((aligned(16))) float Ds[2][32 * 64];
  __shared__ __attribute__((aligned(16))) float Xs[2][XROWS * 64];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int wm = wid >> 1, wn = wid & 1;
  const int lcol4 = tid & 15, lrow = tid >> 4;          // 16 float4 per 64-channel row

  const int ciTile = blockIdx.x % ciT, coTile = blockIdx.x / ciT;
  const int co0 = coTile * 64, ci0 = ciTile * 64;
  const int split = blockIdx.y;
  const int cpr = g.TX >> 5;                            // chunks per image row
  const int nchunks_total = g.N * g.TY * cpr;
  const int cbeg = split * cps;
  const int cend = cbeg + cps < nchunks_total ? cbeg + cps : nchunks_total;
  const int nch = cend > cbeg ? cend - cbeg : 0;

  // chunk coordinates (uniform), advanced incrementally
  int q_n, q_ty, q_tx;                                  // coordinates of the NEXT chunk to be loaded
  {
    const int c = cbeg < nchunks_total ? cbeg : 0;
    const int per_img = g.TY * cpr;
    q_n = c / per_img;
    const int rem = c - q_n * per_img;
    q_ty = rem / cpr;
    q_tx = (rem - q_ty * cpr) << 5;
  }
  const int n0 = q_n;
  const long long img_elems = (long long)g.SH * g.SW * g.src_cstride;
  const __amdgpu_buffer_rsrc_t rsX = make_rsrc(src + (long long)n0 * img_elems, ((long long)g.N - n0) * img_elems * 4);
  const long long mbeg = (long long)cbeg * 32;
  const __amdgpu_buffer_rsrc_t rsD = make_rsrc(dy + mbeg * dy_cstride, (long long)nch * 32 * dy_cstride * 4);
  const int kc = ci0 + lcol4 * 4, cc = co0 + lcol4 * 4;
  const bool kok = kc < g.K, ccok = cc < g.Cout;
  // BatchNorm + ReLU prologue on the source operand (see conv_gather_kernel): a thread's channels never change
  const bool has_pro = pro != nullptr;
  float4 p_sc = zero4(), p_sh = zero4();
  if (has_pro && kok) { p_sc = ld4(pro + kc); p_sh = ld4(pro + g.K + kc); }
  float lim[NSX];                    // +inf where the halo slot holds a real element, 0 where it is padding (prologue only)

  // per-slot constants of the halo rows this thread stages
  int xs_hr[NSX], xs_hx[NSX];
  bool xs_ok[NSX];
#pragma unroll
  for (int k = 0; k < NSX; ++k) {
    const int e = tid + 256 * k;
    const int row = e >> 4;
    xs_ok[k] = row < XROWS;
    xs_hr[k] = row / HW_;
    xs_hx[k] = row - xs_hr[k] * HW_;
  }

  float4 rs[NSLOT];
  int l_chunk = 0;                                      // index (relative to cbeg) of the chunk being loaded
  auto load_slot = [&](int sl) {
    if (sl < NSD) {
      const int mr = l_chunk * 32 + lrow + 16 * sl;     // rows beyond the split fall outside rsD -> 0
      rs[sl] = bld4(rsD, ccok ? (unsigned)(mr * dy_cstride + cc) * 4u : OOB);
    } else {
      const int k = sl - NSD;
      const int iy = q_ty + xs_hr[k] - 1, ix = q_tx + xs_hx[k] - 1;
      const bool ok = xs_ok[k] && kok && l_chunk < nch && (unsigned)iy < (unsigned)g.SH && (unsigned)ix < (unsigned)g.SW;
      const int off = (((q_n - n0) * g.SH + iy) * g.SW + ix) * g.src_cstride + kc;
      rs[sl] = bld4(rsX, ok ? (unsigned)off * 4u : OOB);
      if (has_pro) lim[k] = ok ? __builtin_inff() : 0.f;
    }
  };
  auto advance_chunk = [&]() {
    l_chunk += 1;
    q_tx += 32;
    if (q_tx >= g.TX) { q_tx = 0; q_ty += 1; }
    if (q_ty >= g.TY) { q_ty = 0; q_n += 1; }
  };
  auto store_slot = [&](int sl, int buf) {
    if (sl < NSD) {
      *reinterpret_cast<float4*>(&Ds[buf][(lrow + 16 * sl) * 64 + lcol4 * 4]) = rs[sl];
    } else {
      const int k = sl - NSD;
      float4 v = rs[sl];
      if (has_pro) v = pro_apply(v, p_sc, p_sh, lim[k]);
      if (xs_ok[k]) *reinterpret_cast<float4*>(&Xs[buf][(tid + 256 * k) * 4]) = v;
    }
  };

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

#pragma unroll
  for (int sl = 0; sl < NSLOT; ++sl) load_slot(sl);
  advance_chunk();
#pragma unroll
  for (int sl = 0; sl < NSLOT; ++sl) store_slot(sl, 0);
#pragma unroll
  for (int sl = 0; sl < NSLOT; ++sl) load_slot(sl);
  advance_chunk();
  __syncthreads();

  for (int ch = 0; ch < nch; ++ch) {
    const int buf = ch & 1;
    const float* Db = &Ds[buf][wm * 32 + l31];
    const float* Xb = &Xs[buf][wn * 32 + l31];
#pragma unroll
    for (int kk = 0; kk < 32; kk += 8) {
      float av[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) av[j] = Db[(kk + 4 * h + j) * 64];
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int r = t / 3, sx = t % 3;
        float bv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[j] = Xb[(r * HW_ + kk + 4 * h + j + sx) * 64];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j], acc[t], 0, 0, 0);
        if (kk == 0) {               // stage chunk ch+1 into the other buffer and refill the registers with chunk ch+2
          store_slot(t, buf ^ 1);
          load_slot(t);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (kk == 0) advance_chunk();
    }
    __syncthreads();
  }

  float* out = slab + (long long)(split0 + split) * g.Cout * g.wstride;
  const int ci = ci0 + wn * 32 + l31;
  if (ci < g.K) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (co < g.Cout) out[(long long)co * g.wstride + t * g.K + ci] = acc[t][r];
      }
  }
}

// ------------------------------------------------------------------------------------------------
// weight gradient of the 7x7 / stride 2 / pad 3 stem on the NHWC4 image, all seven filter rows per block.
// A chunk is 32 consecutive output pixels of one output row (requires TX % 32 == 0).  Per chunk the block stages
// dy[32][64] and the 7 x 69 pixel input patch (rows 2*oy-3 .. 2*oy+3, columns 2*ox0-3 .. 2*ox0+65) once.  The
// im2col row of pixel p for filter row r is the contiguous run patch[r][8p .. 8p+31] (7 px x 4 ch + one pad px
// whose products land in the padded weight slots and are dropped by the unpack), so no im2col copy exists.
// Wave w owns filter rows 2w, 2w+1 (row 7 is a dummy) x both 32-channel output tiles.
__global__ __launch_bounds__(256, 2)
void stem_wgrad_kernel(const float* __restrict__ src, const float* __restrict__ dy, float* __restrict__ slab,
                       const DcsConvGeom g, const int dy_cstride, const int split0, const int cps) {
  constexpr int PW = 69;                    // patch width in pixels
  constexpr int PROW = PW * 4 + 4;          // floats per patch row (+4 pad floats so 8p+31 stays in the row)
  constexpr int NSLOT = 4;                  // 2 dy + 2 patch float4 per thread
  __shared__ __attribute__((aligned(16))) float Ds[2][32 * 64];
  __shared__ __attribute__((aligned(16))) float Ps[2][7 * PROW];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int lcol4 = tid & 15, lrow = tid >> 4;
  const int split = blockIdx.x;
  const int cpr = g.TX >> 5;
  const int nchunks_total = g.N * g.TY * cpr;
  const int cbeg = split * cps;
  const int cend = cbeg + cps < nchunks_total ? cbeg + cps : nchunks_total;
  const int nch = cend > cbeg ? cend - cbeg : 0;

  int q_n, q_ty, q_tx;
  {
    const int c = cbeg < nchunks_total ? cbeg : 0;
    const int per_img = g.TY * cpr;
    q_n = c / per_img;
    const int rem = c - q_n * per_img;
    q_ty = rem / cpr;
    q_tx = (rem - q_ty * cpr) << 5;
  }
  const int n0 = q_n;
  const long long img_elems = (long long)g.SH * g.SW * 4;
  const __amdgpu_buffer_rsrc_t rsX = make_rsrc(src + (long long)n0 * img_elems, ((long long)g.N - n0) * img_elems * 4);
  const long long mbeg = (long long)cbeg * 32;
  const __amdgpu_buffer_rsrc_t rsD = make_rsrc(dy + mbeg * dy_cstride, (long long)nch * 32 * dy_cstride * 4);

  // patch staging: element e = tid + 256*k < 7*69 -> (row, pixel)
  int ps_r[2], ps_c[2];
  bool ps_ok[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int e = tid + 256 * k;
    ps_ok[k] = e < 7 * PW;
    ps_r[k] = e / PW;
    ps_c[k] = e - ps_r[k] * PW;
  }

  float4 rs[NSLOT];
  int l_chunk = 0;
  auto load_slot = [&](int sl) {
    if (sl < 2) {
      const int mr = l_chunk * 32 + lrow + 16 * sl;
      rs[sl] = bld4(rsD, (unsigned)(mr * dy_cstride + lcol4 * 4) * 4u);
    } else {
      const int k = sl - 2;
      const int iy = 2 * q_ty - 3 + ps_r[k], ix = 2 * q_tx - 3 + ps_c[k];
      const bool ok = ps_ok[k] && l_chunk < nch && (unsigned)iy < (unsigned)g.SH && (unsigned)ix < (unsigned)g.SW;
      rs[sl] = bld4(rsX, ok ? (unsigned)((((q_n - n0) * g.SH + iy) * g.SW + ix) * 4) * 4u : OOB);
    }
  };
  auto advance_chunk = [&]() {
    l_chunk += 1;
    q_tx += 32;
    if (q_tx >= g.TX) { q_tx = 0; q_ty += 1; }
    if (q_ty >= g.TY) { q_ty = 0; q_n += 1; }
  };
  auto store_slot = [&](int sl, int buf) {
    if (sl < 2) {
      *reinterpret_cast<float4*>(&Ds[buf][(lrow + 16 * sl) * 64 + lcol4 * 4]) = rs[sl];
    } else {
      const int k = sl - 2;
      if (ps_ok[k]) *reinterpret_cast<float4*>(&Ps[buf][ps_r[k] * PROW + ps_c[k] * 4]) = rs[sl];
    }
  };

  if (tid < 14) {                           // the pad floats at the end of each patch row are read, keep them zero
    const int b = tid / 7, r = tid % 7;
    *reinterpret_cast<float4*>(&Ps[b][r * PROW + PW * 4]) = zero4();
  }

  f32x16 acc[2][2];                         // [filter row 2w + a][output-channel tile]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

#pragma unroll
  for (int sl = 0; sl < NSLOT; ++sl) load_slot(sl);
  advance_chunk();
#pragma unroll
  for (int sl = 0; sl < NSLOT; ++sl) store_slot(sl, 0);
#pragma unroll
  for (int sl = 0; sl < NSLOT; ++sl) load_slot(sl);
  advance_chunk();
  __syncthreads();

  const int r0 = 2 * wid;
  const bool two = r0 + 1 < 7;
  for (int ch = 0; ch < nch; ++ch) {
    const int buf = ch & 1;
    const float* Db = &Ds[buf][l31];
    const float* P0 = &Ps[buf][r0 * PROW + l31];
    const float* P1 = &Ps[buf][(two ? r0 + 1 : r0) * PROW + l31];
#pragma unroll
    for (int kk = 0; kk < 32; kk += 8) {
      float a0[4], a1[4], b0[4], b1[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int pix = kk + 4 * h + j;
        a0[j] = Db[pix * 64];
        a1[j] = Db[pix * 64 + 32];
        b0[j] = P0[8 * pix];
        b1[j] = P1[8 * pix];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b0[j], acc[0][1], 0, 0, 0);
        if (two) {
          acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b1[j], acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[j], acc[1][1], 0, 0, 0);
        }
        if (kk == 0) {
          store_slot(j, buf ^ 1);
          load_slot(j);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (kk == 0) advance_chunk();
    }
    __syncthreads();
  }

  float* out = slab + (long long)(split0 + split) * g.Cout * g.wstride;
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int r = r0 + a;
    if (r >= 7) continue;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int co = b * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
        out[(long long)co * g.wstride + r * 32 + l31] = acc[a][b][q];
      }
  }
}

// dw[i] = (accumulate ? dw[i] : 0) + sum_s slab[s][i].  Block = 64 column groups x 4 split lanes: each lane sums splits
// sl, sl+4, ... with independent loads in flight, then the 4 partial sums are added in fixed order (deterministic).
// V = columns per thread: 4 (16-byte loads; n, row_len and dst_stride multiples of 4) or 1.
template <int V>
__global__ __launch_bounds__(256)
void reduce_slab_kernel(const float* __restrict__ slab, float* __restrict__ dw, long long n, int nsplit,
                        int accumulate, int row_len, int dst_stride) {
  __shared__ double sm[4][64][V];
  const int c = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const long long i = ((long long)blockIdx.x * 64 + c) * V;
  double s[V];                         // double accumulators: up to 1024 slabs of mixed sign per element, HBM-bound kernel
#pragma unroll
  for (int v = 0; v < V; ++v) s[v] = 0.0;
  if (i < n) {
    double a[4][V];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int v = 0; v < V; ++v) a[q][v] = 0.0;
    auto add = [&](int q, int k) {
      const float* p = slab + (long long)k * n + i;
      if constexpr (V == 4) {
        const float4 x = *reinterpret_cast<const float4*>(p);
        a[q][0] += (double)x.x; a[q][1] += (double)x.y; a[q][2] += (double)x.z; a[q][3] += (double)x.w;
      } else {
        a[q][0] += (double)p[0];
      }
    };
    int k = sl;
    for (; k + 12 < nsplit; k += 16) { add(0, k); add(1, k + 4); add(2, k + 8); add(3, k + 12); }
    for (; k < nsplit; k += 4) add(0, k);
#pragma unroll
    for (int v = 0; v < V; ++v) s[v] = (a[0][v] + a[1][v]) + (a[2][v] + a[3][v]);
  }
#pragma unroll
  for (int v = 0; v < V; ++v) sm[sl][c][v] = s[v];
  __syncthreads();
  if (sl == 0 && i < n) {
    // row_len > 0: the slab holds compact rows of row_len floats that land at stride dst_stride in dw
    // (a channel slice of a wider weight tensor, used for "virtual concat" convolutions)
    const long long o = row_len > 0 ? (i / row_len) * dst_stride + (i % row_len) : i;
    float r[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const double t = ((sm[0][c][v] + sm[1][c][v]) + (sm[2][c][v] + sm[3][c][v]));
      r[v] = (float)((accumulate ? (double)dw[o + v] : 0.0) + t);
    }
    if constexpr (V == 4) *reinterpret_cast<float4*>(dw + o) = make_float4(r[0], r[1], r[2], r[3]);
    else dw[o] = r[0];
  }
}

__global__ void pack_dgrad_weight_kernel(const float* __restrict__ w, float* __restrict__ o, int Cout, int RS,
                                         int Cin, int cin_total, int cin_off) {
  // o[ci][rs][co] = w[co][rs][cin_off + ci] (w rows have cin_total channels); thread per output element, co fastest
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long n = (long long)Cout * RS * Cin;
  if (i >= n) return;
  const int co = (int)(i % Cout);
  const long long q = i / Cout;
  const int rs = (int)(q % RS);
  const int ci = (int)(q / RS);
  o[i] = w[((long long)co * RS + rs) * cin_total + cin_off + ci];
}

__global__ void pack_stem_weight_kernel(const float* __restrict__ in, float* __restrict__ out, int Cout, int dir) {
  // dir 0: in [Cout][7][7][3] -> out [Cout][7][8][4] zero padded; dir 1: the inverse (drops pads)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (dir == 0) {
    if (i >= Cout * 7 * 32) return;
    const int c = i & 3, s = (i >> 2) & 7, r = (i >> 5) % 7, co = i / 224;
    out[i] = (c < 3 && s < 7) ? in[((co * 7 + r) * 7 + s) * 3 + c] : 0.f;
  } else {
    if (i >= Cout * 147) return;
    const int c = i % 3, s = (i / 3) % 7, r = (i / 21) % 7, co = i / 147;
    out[i] = in[((co * 7 + r) * 8 + s) * 4 + c];
  }
}

__global__ void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int C) {
  __shared__ float tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  for (int j = threadIdx.y; j < 32; j += 8) {
    const int r = by + j, c = bx + threadIdx.x;
    if (r < R && c < C) tile[j][threadIdx.x] = in[(long long)r * C + c];
  }
  __syncthreads();
  for (int j = threadIdx.y; j < 32; j += 8) {
    const int c = bx + j, r = by + threadIdx.x;
    if (r < R && c < C) out[(long long)c * R + r] = tile[threadIdx.x][j];
  }
}

// 3x3 / stride 1 / pad 1 forward geometry whose rows split into whole 32-pixel chunks
bool wgrad3x3_eligible(const DcsConvGeom* g) {
  if (g->stem || g->ntaps != 9 || g->sy != 1 || g->sx != 1 || (g->TX & 31) != 0 || g->wstride != 9 * g->K) return false;
  for (int t = 0; t < 9; ++t)
    if (g->offy[t] != t / 3 - 1 || g->offx[t] != t % 3 - 1 || g->wofs[t] != t * g->K) return false;
  return g->SH == g->TY && g->SW == g->TX;
}


}  // namespace

// K-chunk policy: 64-wide output tiles (64-channel layers, 1x1 skip convs) run the 16-channel chunk variant
// (32.5 KB LDS -> 4 blocks per CU; measured +4 % on 3x3 64->64 and +27 % on the bandwidth-bound 1x1 64->128);
// 128-wide tiles measure the same either way and keep 32-channel chunks.  DCS_CONV_BK16 forces 16 everywhere.
#define g_bk16 (dcs_config().conv_bk16 != 0)
// Measured and dropped: 256-pixel tiles (the BM template parameter; identical TFLOP/s on every C3
// shape, so the loss against peak is not per-tile overhead but MFMA-busy 78-89 % at a 2.0-2.35 GHz DVFS clock) and a
// 3x3 kernel with the 6x34-pixel input halo resident in LDS (3-5 % slower than this per-tap kernel).

static int launch_gather(const float* src, const float* wgt, const float* bias, float* dst, const DcsConvGeom* geom,
                         int accumulate, float* stats, int nsplit, long long slab_stride, void* stream,
                         const BnBwdEpi bnb = BnBwdEpi{nullptr, nullptr, nullptr, 0}, const float* pro = nullptr) {
  int rc = check_geom(geom);
  if (rc != DCS_OK) return rc;
  DCS_CHECK_ARG(src && wgt && dst && dcs_aligned16(src) && dcs_aligned16(wgt));
  DCS_CHECK_ARG(geom->dst_cstride >= geom->Cout && nsplit >= 1 && nsplit <= 64);
  const long long M = (long long)geom->N * geom->TY * geom->TX;
  DCS_CHECK_ARG(M < 0x7FFFFF00ll);
  // 32-bit buffer addressing inside the kernel: source offsets are relative to the first image a block touches and a
  // block's 128 (or, for K splits, the same 128) output pixels may reach into the following image(s); the weight
  // tensor is addressed from its base.  Both windows must stay below 2 GiB (num_records is a 31-bit byte count and
  // offsets >= 2^31 mean "out of range -> 0"), otherwise the kernel would silently read zeros.
  {
    const long long tyx = (long long)geom->TY * geom->TX;
    const long long img_bytes = (long long)geom->SH * geom->SW * geom->src_cstride * 4;
    long long span = 127 / tyx + 2;                       // images a 128-pixel block can touch
    if (span > geom->N) span = geom->N;
    if (span * img_bytes > 0x7FFFFFFFll || (long long)geom->Cout * geom->wstride * 4 > 0x7FFFFFFFll)
      return DCS_E_UNSUPPORTED;
  }
  const int bn = geom->Cout > 64 ? 128 : (geom->Cout > 32 ? 64 : 32);
  const int ntiles = (geom->Cout + bn - 1) / bn;
  DCS_CHECK_ARG(!(stats && accumulate && !bnb.y));
  DCS_CHECK_ARG(!pro || (!geom->stem && geom->K <= DCS_PRO_MAXK && dcs_aligned16(pro)));
  // BatchNorm-backward sums: dense vectorised destination, y (and mask) share its layout
  DCS_CHECK_ARG(!bnb.y || (stats && bnb.bn && nsplit == 1 && (geom->Cout & 3) == 0 && geom->dst_cstride == geom->Cout &&
                           dcs_aligned16(dst) && dcs_aligned16(bnb.y) && (!bnb.mask || dcs_aligned16(bnb.mask))));
  // few K chunks per tile (1x1 convolutions up to 512 channels): 16-channel chunks, 3 blocks per CU, so that the
  // prologue / epilogue of one tile overlaps the main loop of two others (measured +14..18 % on those shapes)
  const bool short_k = (long long)geom->ntaps * ((geom->K + 31) / 32) <= 16;
  const long long mtiles = (M + 127) / 128;
  const long long blocks = mtiles * ntiles;
  DCS_CHECK_ARG(blocks > 0 && blocks < (1ll << 31));
  // K-chunk width of the variant that will run, to cut the chunk range into nsplit equal parts
  const bool stem14 = geom->stem && bn == 64 && geom->ntaps == 14;
  const int bkt = geom->stem ? (stem14 ? 16 : 32) : (bn == 64 ? 16 : (bn == 128 && (g_bk16 || short_k) ? 16 : 32));
  const int nch = geom->ntaps * (geom->stem ? 1 : (geom->K + bkt - 1) / bkt);
  const int cps = (nch + nsplit - 1) / nsplit;
  hipStream_t s = dcs_stream(stream);
#define LAUNCH_K(...)                                                                                                  \
  hipLaunchKernelGGL((conv_gather_kernel<__VA_ARGS__>), dim3((unsigned)blocks, (unsigned)nsplit), dim3(256), 0, s, src, \
                     wgt, bias, dst, *geom, accumulate, ntiles, stats, cps, slab_stride, bnb, pro)
  if (geom->stem) {
    // 14-tap stem geometry = half filter rows of 4 pixels (16 floats): 16-float chunks, 32.5 KB LDS, more blocks per CU
    if (stem14) LAUNCH_K(64, true, 16);
    else if (bn == 128) LAUNCH_K(128, true, 32);
    else if (bn == 64) LAUNCH_K(64, true, 32);
    else LAUNCH_K(32, true, 32);
  } else if (bn == 128) {
    if (bkt == 16) LAUNCH_K(128, false, 16);
    else LAUNCH_K(128, false, 32);
  } else if (bn == 64) {
    LAUNCH_K(64, false, 16);
  } else {
    LAUNCH_K(32, false, 32);
  }
#undef LAUNCH_K
  DCS_LAUNCH_RET();
}

extern "C" int dcs_conv_gather(const float* src, const float* wgt, const float* bias, float* dst,
                               const DcsConvGeom* geom, int accumulate, float* stats, void* stream) {
  return launch_gather(src, wgt, bias, dst, geom, accumulate, stats, 1, 0, stream);
}

extern "C" int dcs_conv_gather_pro(const float* src, const float* wgt, const float* bias, float* dst, const DcsConvGeom* geom,
                                   int accumulate, float* stats, const float* pro, int nsplit, int64_t slab_stride,
                                   void* stream) {
  DCS_CHECK_ARG(pro && nsplit >= 1);
  if (nsplit > 1) {
    DCS_CHECK_ARG(geom && !bias && !stats && !accumulate &&
                  slab_stride >= (int64_t)geom->N * geom->DH * geom->DW * geom->dst_cstride);
    return launch_gather(src, wgt, nullptr, dst, geom, 0, nullptr, nsplit, slab_stride, stream,
                         BnBwdEpi{nullptr, nullptr, nullptr, 0}, pro);
  }
  return launch_gather(src, wgt, bias, dst, geom, accumulate, stats, 1, 0, stream, BnBwdEpi{nullptr, nullptr, nullptr, 0}, pro);
}

extern "C" int dcs_conv_gather_bnbwd(const float* src, const float* wgt, float* dst, const DcsConvGeom* geom, int accumulate,
                                     const float* bn_y, const float* bn_mask, const float* bn, int relu, float* part,
                                     void* stream) {
  DCS_CHECK_ARG(bn_y && bn && part);
  return launch_gather(src, wgt, nullptr, dst, geom, accumulate, part, 1, 0, stream, BnBwdEpi{bn_y, bn_mask, bn, relu});
}

extern "C" int dcs_conv_gather_split(const float* src, const float* wgt, float* slab, const DcsConvGeom* geom, int nsplit,
                                     int64_t slab_stride, void* stream) {
  DCS_CHECK_ARG(geom && slab_stride >= (int64_t)geom->N * geom->DH * geom->DW * geom->dst_cstride);
  return launch_gather(src, wgt, nullptr, slab, geom, 0, nullptr, nsplit, slab_stride, stream);
}

// 128-wide generic weight-gradient tiles stage 16 pixels per chunk (32 KB LDS, 4 blocks per CU): +4..9 % over 32-pixel
// chunks at 2 blocks per CU on the 1x1 layers (the kernel is stall-bound, not MFMA-bound).  DCS_WGRAD_CH32 restores 32.
#define g_wgrad_ch16 (dcs_config().wgrad_ch32 == 0)

static int launch_wgrad(const float* src, const float* dy, float* slab, const DcsConvGeom* geom, int dy_cstride, int split0,
                        int nsplit, void* stream, const float* pro);

extern "C" int dcs_conv_wgrad(const float* src, const float* dy, float* slab, const DcsConvGeom* geom,
                              int dy_cstride, int split0, int nsplit, void* stream) {
  return launch_wgrad(src, dy, slab, geom, dy_cstride, split0, nsplit, stream, nullptr);
}

extern "C" int dcs_conv_wgrad_pro(const float* src, const float* dy, float* slab, const DcsConvGeom* geom, int dy_cstride,
                                  int split0, int nsplit, const float* pro, void* stream) {
  DCS_CHECK_ARG(pro && geom && !geom->stem && dcs_aligned16(pro));
  return launch_wgrad(src, dy, slab, geom, dy_cstride, split0, nsplit, stream, pro);
}

static int launch_wgrad(const float* src, const float* dy, float* slab, const DcsConvGeom* geom, int dy_cstride, int split0,
                        int nsplit, void* stream, const float* pro) {
  int rc = check_geom(geom);
  if (rc != DCS_OK) return rc;
  DCS_CHECK_ARG(src && dy && slab && dcs_aligned16(src) && dcs_aligned16(dy));
  DCS_CHECK_ARG((dy_cstride & 3) == 0 && dy_cstride >= geom->Cout && nsplit > 0 && split0 >= 0);
  // the dy pixel index equals the GEMM row index: destination sub-grid must be the dense tensor
  DCS_CHECK_ARG(geom->dsy == 1 && geom->dsx == 1 && geom->dy0 == 0 && geom->dx0 == 0 &&
                geom->TY == geom->DH && geom->TX == geom->DW);
  const long long M = (long long)geom->N * geom->TY * geom->TX;
  long long mps = (M + nsplit - 1) / nsplit;
  mps = (mps + 31) / 32 * 32;
  // 32-bit buffer offsets inside one split: dy rows and the source pixels they gather from must span < 2 GiB
  const long long span_src = (mps * geom->sy * geom->sx + 4ll * geom->SW) * geom->src_cstride * 4;
  if (mps * (long long)dy_cstride * 4 >= 0x7FFFFFFFll || span_src >= 0x7FFFFFFFll) return DCS_E_UNSUPPORTED;
  if (geom->stem && (geom->TX & 31) == 0 && geom->Cout == 64 && geom->wstride == 224 && dy_cstride == 64) {
    const long long nchunks = (long long)geom->N * geom->TY * (geom->TX / 32);
    const int cps = (int)((nchunks + nsplit - 1) / nsplit);
    const long long span = ((long long)cps * 64 + 8ll * geom->SW) * 16;
    if ((long long)cps * 32 * dy_cstride * 4 < 0x7FFFFFFFll && span < 0x7FFFFFFFll) {
      hipLaunchKernelGGL(stem_wgrad_kernel, dim3((unsigned)nsplit), dim3(256), 0, dcs_stream(stream), src, dy, slab, *geom,
                         dy_cstride, split0, cps);
      DCS_LAUNCH_RET();
    }
  }
  if (wgrad3x3_eligible(geom)) {
    const int cpr = geom->TX / 32;
    const long long nchunks = (long long)geom->N * geom->TY * cpr;
    const int cps = (int)((nchunks + nsplit - 1) / nsplit);
    const int coT = (geom->Cout + 63) / 64, ciT = (geom->K + 63) / 64;
    const long long span = ((long long)cps * 32 + 4ll * geom->SW) * geom->src_cstride * 4;
    if ((long long)cps * 32 * dy_cstride * 4 < 0x7FFFFFFFll && span < 0x7FFFFFFFll) {
      hipLaunchKernelGGL(conv_wgrad3x3_kernel, dim3((unsigned)(coT * ciT), (unsigned)nsplit), dim3(256), 0,
                         dcs_stream(stream), src, dy, slab, *geom, dy_cstride, split0, cps, ciT, pro);
      DCS_LAUNCH_RET();
    }
  }
  const int keff = geom->stem ? 32 : geom->K;
  const int bt = (geom->Cout > 64 && keff > 64) ? 128 : 64;
  const int coT = (geom->Cout + bt - 1) / bt, ciT = (keff + bt - 1) / bt;
  hipStream_t s = dcs_stream(stream);
  dim3 grid((unsigned)(geom->ntaps * coT * ciT), (unsigned)nsplit);
  if (bt == 128)
    if (g_wgrad_ch16)
      hipLaunchKernelGGL((conv_wgrad_kernel<128, 16>), grid, dim3(256), 0, s, src, dy, slab, *geom, dy_cstride, split0, mps, ciT, pro);
    else
      hipLaunchKernelGGL(conv_wgrad_kernel<128>, grid, dim3(256), 0, s, src, dy, slab, *geom, dy_cstride, split0, mps, ciT, pro);
  else
    hipLaunchKernelGGL(conv_wgrad_kernel<64>, grid, dim3(256), 0, s, src, dy, slab, *geom, dy_cstride, split0, mps, ciT, pro);
  DCS_LAUNCH_RET();
}

extern "C" int dcs_reduce_slab(const float* slab, float* dw, int64_t n, int nsplit, int accumulate, int row_len,
                               int dst_stride, void* stream) {
  DCS_CHECK_ARG(slab && dw && n > 0 && nsplit > 0 && row_len >= 0 && (row_len == 0 || (dst_stride >= row_len && n % row_len == 0)));
  const bool v4 = (n & 3) == 0 && (row_len & 3) == 0 && (dst_stride & 3) == 0 && dcs_aligned16(slab) && dcs_aligned16(dw);
  if (v4)
    hipLaunchKernelGGL(reduce_slab_kernel<4>, dim3((unsigned)((n / 4 + 63) / 64)), dim3(256), 0, dcs_stream(stream), slab, dw,
                       (long long)n, nsplit, accumulate, row_len, dst_stride);
  else
    hipLaunchKernelGGL(reduce_slab_kernel<1>, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, dcs_stream(stream), slab, dw,
                       (long long)n, nsplit, accumulate, row_len, dst_stride);
  DCS_LAUNCH_RET();
}

extern "C" int dcs_pack_dgrad_weight(const float* w_krsc, float* w_crsk, int Cout, int R, int S, int Cin,
                                     int cin_total, int cin_off, void* stream) {
  DCS_CHECK_ARG(w_krsc && w_crsk && Cout > 0 && R > 0 && S > 0 && Cin > 0 && cin_off >= 0 && cin_off + Cin <= cin_total);
  const long long n = (long long)Cout * R * S * Cin;
  hipLaunchKernelGGL(pack_dgrad_weight_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, dcs_stream(stream),
                     w_krsc, w_crsk, Cout, R * S, Cin, cin_total, cin_off);
  DCS_LAUNCH_RET();
}

extern "C" int dcs_pack_stem_weight(const float* in, float* out, int Cout, int dir, void* stream) {
  DCS_CHECK_ARG(in && out && Cout > 0 && (dir == 0 || dir == 1));
  const int n = dir == 0 ? Cout * 224 : Cout * 147;
  hipLaunchKernelGGL(pack_stem_weight_kernel, dim3((n + 255) / 256), dim3(256), 0, dcs_stream(stream), in, out, Cout, dir);
  DCS_LAUNCH_RET();
}

extern "C" int dcs_transpose(const float* in, float* out, int R, int C, void* stream) {
  DCS_CHECK_ARG(in && out && R > 0 && C > 0);
  hipLaunchKernelGGL(transpose_kernel, dim3((C + 31) / 32, (R + 31) / 32), dim3(32, 8), 0, dcs_stream(stream), in, out, R, C);
  DCS_LAUNCH_RET();
}
