// fp32 convolution on the bf16 matrix cores of gfx950: every fp32 operand is split EXACTLY into three bf16 pieces
// (x = x1 + x2 + x3: 3 x 8 significand bits = the 24 of an fp32) and a product a*b is accumulated in fp32 from the six
// piece products whose magnitude is >= 2^-16 |a b| (a1b1, a1b2, a2b1, a1b3, a3b1, a2b2); the three dropped ones are
// <= 2^-24 |a b| each, i.e. below the rounding error of one fp32 multiply-add.  Every piece product is exact in fp32
// (8 x 8 bits) and v_mfma_f32_32x32x16_bf16 accumulates in fp32, so the result carries fp32-class error (measured
// against float64 next to the exact-fp32 v_mfma_f32_32x32x2_f32 kernel in tests/test_kernels_gpu.py) at 6/16 of the
// matrix-pipe time: v_mfma_f32_32x32x2_f32 spends 32 cycles per k, six v_mfma_f32_32x32x16_bf16 spend 12.
//
// Same implicit-GEMM gather as conv_igemm.hip (nn.Conv2d forward and its data gradient, network/backbone/
// resnet_pyramid.py:23-25,:110-112,:139, network/utils.py:46-47), same epilogue (conv_shared.h), same prologue.
// Tile: 128 output pixels x BN output channels per block of 4 waves; K in chunks of 16 input channels of one tap.
// LDS row = [piece 1: 16 bf16][piece 2][piece 3][16 B pad] = 112 B (7 x 16 B: odd, ds_read_b128 conflict-free);
// activations are split on their way from the staging registers to LDS (after the optional BatchNorm + ReLU
// prologue), weights are split once per launch sequence by dcs_split_weight into exactly this row image
// [Cout][K chunk][piece][16], so their staging is a straight 96-byte copy.
#include "conv_shared.h"
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

namespace {

// (x0, x1) -> the three bf16 pieces of each, packed pairwise (low half = x0's piece).  Round-to-nearest pieces: the
// remainders are exact in fp32 (Sterbenz), the third remainder has <= 8 significant bits and converts exactly.
__device__ __forceinline__ void split3_pair(const float x0, const float x1, unsigned& q1, unsigned& q2, unsigned& q3) {
  // scalar subtractions on purpose: packed f32 VALU ops cost ~3x their issue slot beside MFMAs (MI355X_MICROARCH.md)
  const f32x2 v = {x0, x1};
  const bf16x2 p1 = __builtin_convertvector(v, bf16x2);
  const f32x2 f1 = __builtin_convertvector(p1, f32x2);
  const f32x2 r1 = {x0 - f1.x, x1 - f1.y};
  const bf16x2 p2 = __builtin_convertvector(r1, bf16x2);
  const f32x2 f2 = __builtin_convertvector(p2, f32x2);
  const f32x2 r2 = {r1.x - f2.x, r1.y - f2.y};
  const bf16x2 p3 = __builtin_convertvector(r2, bf16x2);
  q1 = __builtin_bit_cast(unsigned, p1);
  q2 = __builtin_bit_cast(unsigned, p2);
  q3 = __builtin_bit_cast(unsigned, p3);
}

__device__ __forceinline__ void split3_quad(const float4 v, uint2& p1, uint2& p2, uint2& p3) {
  split3_pair(v.x, v.y, p1.x, p2.x, p3.x);
  split3_pair(v.z, v.w, p1.y, p2.y, p3.y);
}

// ---- fp32 through TWO fp16 pieces (round 3; forward 3x3 convolutions, see conv3x3_x3w_body<.., NP = 2>) ----------------
// x * 2^k = h1 + h2 + r with h1 = fp16(x 2^k), h2 = fp16(x 2^k - h1): two 11-bit significands, |r| <= 2^-24 |x 2^k| -- the
// rounding of ONE fp32 -- as long as h2 stays normal (|x 2^k| >= 2^-2; below that its absolute error is 2^-25: negligible in
// a sum with terms of order one) and nothing overflows (|x 2^k| < 65504).  A product needs h1 h1' + h1 h2' + h2 h1' (the
// dropped h2 h2' is <= 2^-24 |x x'|): THREE v_mfma_f32_32x32x16_f16 where the bf16 split needs six -- fp16 buys its third of
// the significand with the exponent range, which a power-of-two scale per operand (exact) gives back: activations are
// scaled by 2^X2H_KX (|x| < 16384), weights by 2^X2H_KW (|w| < 64), the accumulator by 2^-(KX + KW).
constexpr int X2H_KX = 2, X2H_KW = 10;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2h_pair(const float x0, const float x1, unsigned& q1, unsigned& q2) {
  const f32x2 v = {x0, x1};
  const f16x2 p1 = __builtin_convertvector(v, f16x2);                 // round to nearest even
  const f32x2 f1 = __builtin_convertvector(p1, f32x2);
  const f32x2 r1 = {x0 - f1.x, x1 - f1.y};                            // exact
  const f16x2 p2 = __builtin_convertvector(r1, f16x2);
  q1 = __builtin_bit_cast(unsigned, p1);
  q2 = __builtin_bit_cast(unsigned, p2);
}
__device__ __forceinline__ void split2h_quad(const float4 v, const float sc, uint2& p1, uint2& p2) {
  split2h_pair(v.x * sc, v.y * sc, p1.x, p2.x);
  split2h_pair(v.z * sc, v.w * sc, p1.y, p2.y);
}

// w [rows][wstride] fp32 -> out [rows][wstride / 16][3][16] bf16 (the LDS row image of one 16-channel chunk)
__global__ void split_weight_kernel(const float* __restrict__ w, uint2* __restrict__ out, const long long n4,
                                    const int wstride) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // float4 index
  if (i >= n4) return;
  const int s4 = wstride >> 2;
  const long long row = i / s4;
  const int k4 = (int)(i - row * s4);
  const int c = k4 >> 2, q = k4 & 3;
  uint2 p1, p2, p3;
  split3_quad(ld4(w + i * 4), p1, p2, p3);
  uint2* o = out + ((row * (wstride >> 4) + c) * 3) * 4 + q;
  o[0] = p1; o[4] = p2; o[8] = p3;
}

// the fp16 two-piece form: out [rows][wstride / 16][2][16] fp16 of w * 2^X2H_KW (64 B per 16-channel chunk)
__global__ void split_weight_h2_kernel(const float* __restrict__ w, uint2* __restrict__ out, const long long n4,
                                       const int wstride) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // float4 index
  if (i >= n4) return;
  const int s4 = wstride >> 2;
  const long long row = i / s4;
  const int k4 = (int)(i - row * s4);
  const int c = k4 >> 2, q = k4 & 3;
  uint2 p1, p2;
  split2h_quad(ld4(w + i * 4), (float)(1 << X2H_KW), p1, p2);
  uint2* o = out + ((row * (wstride >> 4) + c) * 2) * 4 + q;
  o[0] = p1; o[4] = p2;
}

// Block coordinates of a kernel body: the hardware block id for single launches, the level-relative id inside a multi
// launch (see the *_multi_kernel wrappers at the end of the kernels).
struct BlkId { int x, nx, y, ny = 0; };      // ny > 0: the kernel may re-map (x, y) over the nx * ny grid (XCD locality)
// Weight-gradient grids are (operand tiles) x (pixel splits): the blocks of ONE split read the same dy / input pixels.  Hardware
// block ids go round-robin over the 8 XCDs, which would hand those blocks to 8 different L2s (each fetching the pixels from
// the fabric again: PMC FETCH_SIZE 2.1x the algorithmic bytes); re-mapped, a split's tiles are neighbours on one XCD.
__device__ __forceinline__ void wgrad_blk(const BlkId bi, int& bx, int& by) {
  bx = bi.x; by = bi.y;
  if (bi.ny > 0 && bi.nx > 1) {          // one tile per split: nothing shared, keep the interleaved order
    const int l = dcs_xcd_remap(bi.y * bi.nx + bi.x, bi.nx * bi.ny);
    bx = l % bi.nx; by = l / bi.nx;
  }
}

constexpr int X3_ROWB = 112;      // bytes per LDS row

// STEM: the 7x7 / stride 2 stem on the NHWC4 image in its 14-tap form (ops.geom_stem_fwd): a tap is a filter half-row of
// 4 pixels x 4 channels = 16 contiguous floats = exactly one chunk; lane quad q of a row is pixel q of the tap and is
// range-checked on its own (network/backbone/resnet_pyramid.py:110-112).
// NP = 2: the fp16 two-piece form (split2h_quad; weights from dcs_split_weight_h2: 64-byte chunk images): LDS rows of 80 B,
// three MFMAs per product; src scaled by 2^X2H_KX or, with src_max, by the power of two that puts the tensor's maximum into
// [2^13, 2^14); the result by the inverse of both scales.
// WF (round 3, fp16 form): the WEIGHT fragments come straight from global memory in the fragment-major split image of
// dcs_split_weight_frag_h2 (as in conv3x3_x3w_kernel) instead of being staged through LDS.  With three MFMAs per product
// the LDS-staged form moves 48 KB through LDS per block and 16-channel step (A + B written, both read back per wave) against
// 384 MFMA cycles: two blocks per CU sit at the 128 B/clk of the LDS -- the 1x1 and strided layers stood at 180-250 TF
// whatever their shape.  Only A (the gathered pixels) still passes through LDS; B costs four 1-KiB loads per wave and step.
template <int BN, int BM = 128, bool STEM = false, int NP = 3, bool WF = false>
__device__ __forceinline__
void conv_gather_x3_body(const float* __restrict__ src, const unsigned char* __restrict__ wsp,
                           const float* __restrict__ bias, float* __restrict__ dst, const DcsConvGeom& g,
                           const int accumulate, const int ntiles, float* __restrict__ stats, const int cps,
                           const long long slab_stride, const BnBwdEpi bnb, const float* __restrict__ pro, const BlkId bi,
                           const unsigned* __restrict__ src_max = nullptr) {
  constexpr int ROWB = NP == 3 ? X3_ROWB : 80;                     // LDS row
  constexpr int WB = 32 * NP;                                       // bytes of one 16-channel chunk image of a weight row
  constexpr int WU = 2 * NP;                                        // ... in 16-byte units
  // 128 x 128 and 128 x 64 tiles: waves 2 x 2; 256 x 64 (64-channel layers of large maps): waves 4 x 1, so that a wave
  // still owns a 64 x 64 sub-tile (24 MFMAs per chunk against 5-6 staging slots instead of 12 against 4)
  constexpr int WN = BM == 256 ? 1 : 2, WM = 4 / WN, TM = BM / (WM * 32), TN = BN / (WN * 32);
  constexpr int A_BYTES = BM * ROWB, B_BYTES = WF ? 0 : BN * ROWB;
  constexpr int EPI_FLOATS_ = 4 * 32 * (TN * 32 + 4) + WM * BN * 2;                 // what conv_epilogue stages
  constexpr int SMEM_FLOATS = 2 * (A_BYTES + B_BYTES) / 4 > EPI_FLOATS_ ? 2 * (A_BYTES + B_BYTES) / 4 : EPI_FLOATS_;
  constexpr int NA = BM / 64;                // A slots per thread: 64 rows x 4 float4 each
  constexpr int NB = WF ? 0 : (BN * WU + 255) / 256;  // B slots per thread: 16-byte pieces of the row images
  constexpr int NBA = NB > 0 ? NB : 1;
  constexpr int NSLOT = NA + NB;
  static_assert(!WF || (NP == 2 && !STEM), "fragment-major weights: fp16 form of the plain per-tap kernel only");
  static_assert((BN == 128 || BN == 64) && (BM == 128 || (BM == 256 && BN == 64)) && TM == 2, "unsupported tile");

  __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
  unsigned char* const sm = reinterpret_cast<unsigned char*>(smem);
  __shared__ long long rowoff[BM];
  __shared__ int s_oy[DCS_MAX_TAPS], s_ox[DCS_MAX_TAPS], s_wo[DCS_MAX_TAPS], s_to[DCS_MAX_TAPS];
  __shared__ __attribute__((aligned(16))) float s_pro[2 * DCS_PRO_MAXK];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int wm = wid / WN, wn = wid % WN;
  const int lrow = tid >> 2, lcol4 = tid & 3;
  const bool has_pro = !STEM && pro != nullptr;
  if (has_pro)
    for (int e = tid; e < 2 * g.K; e += 256) s_pro[(e < g.K ? 0 : DCS_PRO_MAXK - g.K) + e] = pro[e];
  float x2h_in = (float)(1 << X2H_KX), x2h_out = 1.f / (float)(1 << (X2H_KX + X2H_KW));
  if (NP == 2 && src_max != nullptr) {
    const unsigned mbits = __builtin_amdgcn_readfirstlane(*src_max);
    const int e = (int)((mbits >> 23) & 0xffu) - 126;               // max = f 2^e, f in [0.5, 1)
    int k = (mbits >> 23) == 0u ? 0 : 14 - e;
    k = k < -100 ? -100 : (k > 100 ? 100 : k);
    x2h_in = __uint_as_float((unsigned)(127 + k) << 23);
    x2h_out = __uint_as_float((unsigned)(127 - k - X2H_KW) << 23);
  }

  const int bid = dcs_xcd_remap(bi.x, bi.nx);
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

  const int n0 = (int)(m0 / TYX);
  const long long img_elems = (long long)g.SH * g.SW * g.src_cstride;
  const int wchunks = g.wstride >> 4;
  const __amdgpu_buffer_rsrc_t rsA = make_rsrc(src + (long long)n0 * img_elems, ((long long)g.N - n0) * img_elems * 4);
  const int J = (g.Cout + 31) >> 5;
  const __amdgpu_buffer_rsrc_t rsB = make_rsrc(reinterpret_cast<const float*>(wsp), WF ? 2ll * wchunks * J * NP * 1024
                                                                                          : (long long)g.Cout * wchunks * WB);

  if (tid < g.ntaps) {
    s_oy[tid] = g.offy[tid]; s_ox[tid] = g.offx[tid];
    s_wo[tid] = (g.wofs[tid] >> 4) * WB;                       // byte offset of the tap's first chunk in a weight row
    s_to[tid] = (g.offy[tid] * g.SW + g.offx[tid]) * g.src_cstride;
  }

  int r_base[NA], r_y[NA], r_x[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const unsigned m = m0 + lrow + 64 * i;
    const bool ok = m < M;
    const unsigned mm = ok ? m : m0;
    const int n = (int)(mm / TYX);
    const int rem = (int)(mm - (unsigned)n * TYX);
    const int ty = rem / g.TX, tx = rem - ty * g.TX;
    r_y[i] = ok ? ty * g.sy : -(1 << 20);
    r_x[i] = tx * g.sx;
    r_base[i] = (((n - n0) * g.SH + ty * g.sy) * g.SW + tx * g.sx) * g.src_cstride;
  }
  int b_off[NBA], b_lds[NBA];        // byte offset of this thread's 16-byte piece in the split weights (or -1), in LDS
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int e = tid + 256 * j;
    const int brow = e / WU, bs = e - brow * WU;
    const bool ok = e < BN * WU && co0 + brow < g.Cout;
    b_off[j] = ok ? ((co0 + brow) * wchunks) * WB + bs * 16 : -1;
    b_lds[j] = e < BN * WU ? brow * ROWB + bs * 16 : -1;
  }
  __syncthreads();

  const int kch = STEM ? 1 : g.K >> 4;
  const int nch = g.ntaps * kch;
  const int cbeg = (int)bi.y * cps < nch ? (int)bi.y * cps : nch;
  const int cend = cbeg + cps < nch ? cbeg + cps : nch;
  dst += (long long)bi.y * slab_stride;

  float4 rs[NA];
  u32x4 rb[NBA];
  float lim[NA];                     // prologue: +inf where the A slot holds a real element, 0 where it is padding
  int kc_held = 0;                   // first channel of the chunk the registers hold (prologue scale / shift lookup)
  // the next chunk to load, advanced incrementally (tap-major, channel chunks inside a tap)
  int ld_ch = cbeg < nch ? cbeg : nch - 1;
  int ld_t = ld_ch / kch, ld_c0 = (ld_ch - ld_t * kch) << 4;
  int c_oy = 0, c_ox = 0, c_wo = 0, c_to = 0, c_kc = 0;
  auto next_chunk = [&]() {          // chunk parameters of ld_ch into c_*, then advance (stays on the last chunk)
    c_oy = s_oy[ld_t]; c_ox = s_ox[ld_t]; c_to = s_to[ld_t];
    c_wo = s_wo[ld_t] + (ld_c0 >> 4) * WB;
    c_kc = ld_c0 + lcol4 * 4;
    if (ld_ch + 1 < cend) {
      ld_ch += 1; ld_c0 += 16;
      if (STEM || ld_c0 >= g.K) { ld_c0 = 0; ld_t += 1; }
    }
  };
  auto load_slot = [&](int sl) {
    if (sl < NA) {
      const int i = sl;
      const bool ok = (unsigned)(r_y[i] + c_oy) < (unsigned)g.SH &&
                      (unsigned)(r_x[i] + c_ox + (STEM ? lcol4 : 0)) < (unsigned)g.SW;
      rs[i] = bld4(rsA, ok ? (unsigned)(r_base[i] + c_to + c_kc) * 4u : OOB);
      if (has_pro) lim[i] = ok ? __builtin_inff() : 0.f;
    } else {
      const int j = sl - NA;
      rb[j] = __builtin_amdgcn_raw_buffer_load_b128(rsB, b_off[j] >= 0 ? (unsigned)(b_off[j] + c_wo) : OOB, 0, 0);
    }
  };
  // WF: the weight fragments of the chunk computed NEXT, per wave (double-buffered by chunk parity like the accumulators)
  bf16x8 fbq[2][TN][NP];
  int w_t = (cbeg < nch ? cbeg : 0) / kch, w_k = (cbeg < nch ? cbeg : 0) - w_t * kch;
  const unsigned lane16 = (unsigned)lane * 16u;
  const int jt0 = (co0 >> 5) + wn * TN;
  auto load_wf = [&](auto S) {
    constexpr int s_ = decltype(S)::value;
    if constexpr (WF) {
      const int c = (g.wofs[w_t] >> 4) + w_k;
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int p = 0; p < NP; ++p)
          fbq[s_][b][p] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
              rsB, (unsigned)(((c * J + jt0 + b) * NP + p) * 1024) + lane16, 0, 0));
      if (w_t * kch + w_k + 1 < cend) { w_k += 1; if (w_k == kch) { w_k = 0; w_t += 1; } }
    }
  };
  float4 p_sc = zero4(), p_sh = zero4();
  auto store_slot = [&](auto NEG, int sl, int buf) {
    if (sl < NA) {
      float4 v = rs[sl];
      if (has_pro) v = pro_apply(v, p_sc, p_sh, lim[sl]);
      if (decltype(NEG)::value) { v.x = -v.x; v.y = -v.y; v.z = -v.z; v.w = -v.w; }   // odd chunks: see the main loop
      unsigned char* q = sm + buf * A_BYTES + (lrow + 64 * sl) * ROWB + lcol4 * 8;
      if constexpr (NP == 3) {
        uint2 p1, p2, p3;
        split3_quad(v, p1, p2, p3);
        *reinterpret_cast<uint2*>(q) = p1;
        *reinterpret_cast<uint2*>(q + 32) = p2;
        *reinterpret_cast<uint2*>(q + 64) = p3;
      } else {
        uint2 p1, p2;
        split2h_quad(v, x2h_in, p1, p2);
        *reinterpret_cast<uint2*>(q) = p1;
        *reinterpret_cast<uint2*>(q + 32) = p2;
      }
    } else {
      const int j = sl - NA;
      if (NB * 256 == BN * WU || b_lds[j] >= 0)
        *reinterpret_cast<u32x4*>(sm + 2 * A_BYTES + buf * B_BYTES + b_lds[j]) = rb[j];
    }
  };
  auto load_pro = [&](int kc) { p_sc = ld4(&s_pro[kc]); p_sh = ld4(&s_pro[DCS_PRO_MAXK + kc]); };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;

  // Rounding bias: v_mfma_f32_32x32x16_bf16 does not round its sums to nearest -- against float64 its results sit
  // ~0.3 mean-absolute-errors BELOW the exact value whatever their sign (v_mfma_f32_32x32x2_f32: no bias; measured by
  // tools/conv_bias_probe.py).  A sign-independent bias survives sums over millions of pixels that cancel to almost
  // nothing (BatchNorm bias gradients), where zero-mean rounding errors do not.  So there are two accumulator sets:
  // even chunks add +A B to acc[0], odd chunks add (-A) B to acc[1] (the A pieces of odd chunks are negated on their
  // way to LDS), and the tile is acc[0] - acc[1]: the hardware's downward bias enters the two halves of the sum with
  // opposite sign and cancels.
  f32x16 acc[2][TM][TN];
#pragma unroll
  for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[s_][a][b][r] = 0.f;

  load_wf(std::integral_constant<int, 0>{});      // WF: chunk cbeg
  // Software pipeline as in conv_gather_kernel: registers hold chunk i+1 while chunk i is computed; each slot is written
  // to the other LDS buffer and re-loaded with chunk i+2 between groups of MFMAs; one barrier per chunk.
  next_chunk();
#pragma unroll
  for (int sl = 0; sl < NSLOT; ++sl) load_slot(sl);
  if (has_pro) load_pro(c_kc);
#pragma unroll
  for (int sl = 0; sl < NSLOT; ++sl) store_slot(P0{}, sl, 0);
  next_chunk();
  kc_held = c_kc;
#pragma unroll
  for (int sl = 0; sl < NSLOT; ++sl) load_slot(sl);
  __syncthreads();

  // piece products, smallest first: (a1 b3), (a3 b1), (a2 b2), (a1 b2), (a2 b1), (a1 b1)
  constexpr int NTERM = NP == 3 ? 6 : 3;
  constexpr int PA[6] = {NP == 3 ? 0 : 1, NP == 3 ? 2 : 0, NP == 3 ? 1 : 0, 0, 1, 0};
  constexpr int PB[6] = {NP == 3 ? 2 : 0, NP == 3 ? 0 : 1, NP == 3 ? 1 : 0, 1, 0, 0};
  constexpr int G = NTERM * TM * TN;           // MFMAs per chunk
  constexpr int G0 = TM * TN;                  // issued before the staging starts
  bf16x8 fa[TM][NP], fb[TN][NP];
  // iteration i (parity PAR) computes chunk cbeg + i from LDS buffer PAR into acc[PAR], writes chunk cbeg + i + 1 (in
  // the registers; negated when i + 1 is odd) to the other buffer and re-loads the registers with chunk cbeg + i + 2
  auto step = [&](auto PAR) {
    constexpr int par = decltype(PAR)::value;
    auto mfma_range = [&](int lo, int hi) {
#pragma unroll
      for (int q = 0; q < G; ++q) {
        if (q < lo || q >= hi) continue;
        const int term = q / (TM * TN), a = (q / TN) % TM, b = q % TN;
        if constexpr (NP == 3)
          acc[par][a][b] =
              __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][PA[term]], fb[b][PB[term]], acc[par][a][b], 0, 0, 0);
        else
          acc[par][a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[a][PA[term]]),
                                                                  __builtin_bit_cast(f16x8, fb[b][PB[term]]), acc[par][a][b], 0, 0, 0);
      }
    };
    const unsigned char* Ab = sm + par * A_BYTES + (wm * TM * 32 + l31) * ROWB + h * 16;
    const unsigned char* Bb = sm + 2 * A_BYTES + par * B_BYTES + (wn * TN * 32 + l31) * ROWB + h * 16;
#pragma unroll
    for (int p = NP - 1; p >= 0; --p) {
#pragma unroll
      for (int a = 0; a < TM; ++a) fa[a][p] = *reinterpret_cast<const bf16x8*>(Ab + a * 32 * ROWB + p * 32);
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        if constexpr (WF) fb[b][p] = fbq[par][b][p];
        else fb[b][p] = *reinterpret_cast<const bf16x8*>(Bb + b * 32 * ROWB + p * 32);
      }
    }
    load_wf(std::integral_constant<int, 1 - par>{});                   // WF: the next chunk's weight fragments
    mfma_range(0, G0);
    if (has_pro) load_pro(kc_held);
    next_chunk();
#pragma unroll
    for (int sl = 0; sl < NSLOT; ++sl) {
      store_slot(std::integral_constant<int, 1 - par>{}, sl, par ^ 1);
      load_slot(sl);
      mfma_range(G0 + sl * (G - G0) / NSLOT, G0 + (sl + 1) * (G - G0) / NSLOT);
      __builtin_amdgcn_sched_barrier(0);
    }
    kc_held = c_kc;
    __syncthreads();
  };
  for (int ch = cbeg; ch < cend; ch += 2) {
    step(P0{});
    if (ch + 1 < cend) step(P1{});
  }
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[0][a][b][r] -= acc[1][a][b][r];
        if constexpr (NP == 2) acc[0][a][b][r] *= x2h_out;             // a power of two: exact
      }

  conv_epilogue<BM, BN, TM, TN, WM, SMEM_FLOATS>(acc[0], smem, rowoff, bias, dst, g.dst_cstride, g.Cout, co0, accumulate, stats,
                                                 mtile, M, wm, wn, bnb, m0 + BM <= M);
}


// ------------------------------------------------------------------------------------------------
// weight gradient on the bf16 matrix cores: dW[co][tap][ci] = sum_m dy[m][co] * src[gather(m,tap)][ci], pixels are the
// GEMM K dimension.  Block = one (tap, co tile, ci tile) and one pixel range (split), 16 pixels (one MFMA k step) per
// chunk.  Both operands arrive pixel-major and are staged exactly as loaded, [pixel][piece][channel] rows of bf16; the
// MFMA wants 8 consecutive pixels of one channel per lane, which gfx950's transposing LDS read delivers
// (ds_read_b64_tr_b16: a 16-lane group reads 4 rows x 16 columns of 16-bit elements and each lane receives one column,
// cdna_hip_programming.md T10): two of them per fragment.  Row = 3 pieces x BT channels x 2 B + 64 B pad, i.e.
// = 64 or 192 (mod 256): the four rows of a transposed read fall in different bank quarters (conflict-free).
// Register-staged double buffer as in conv_gather_x3_kernel.
// Rounding bias of the bf16 MFMA (see conv_gather_x3_kernel): splits with an odd index accumulate (-dy) x and write the
// NEGATED accumulator, so their downward bias enters the slab sum with the opposite sign of the even splits' -- an
// even nsplit cancels it in dcs_reduce_slab.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

// NP = 2 (dy_max given): the fp16 two-piece form, scaled like conv_wgrad3x3_x3r_body<2> (input by 2^X2H_KX, dy by its maximum).
template <int BT, int NP = 3>
__device__ __forceinline__
void conv_wgrad_x3_body(const float* __restrict__ src, const float* __restrict__ dy, float* __restrict__ slab,
                          const DcsConvGeom& g, const int dy_cstride, const int split0, const long long mps,
                          const int ciT, const float* __restrict__ pro, const BlkId bi,
                          const unsigned* __restrict__ dy_max = nullptr) {
  constexpr int T = BT / 64;             // 32x32 tiles per wave per dim
  constexpr int CHP = 16;                // pixels per chunk
  constexpr int PIECE = BT * 2;          // bytes of one piece of one pixel row
  constexpr int ROW = NP * PIECE + 64;   // 832 / 448 (NP = 3: BT = 128 / 64), 576 / 320 (NP = 2): all = 64 or 192 (mod 256)
  constexpr int Q = BT / 4;              // channel quads per operand tile
  constexpr int NI = CHP * Q / 256;      // float4 per thread per operand and chunk: 2 / 1
  constexpr int NSLOT = 2 * NI;
  constexpr int OPB = CHP * ROW;         // bytes of one operand image
  static_assert(NI >= 1, "tile too narrow");
  __shared__ __attribute__((aligned(16))) unsigned char sm[4 * OPB];   // D buf 0, D buf 1, X buf 0, X buf 1

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int wm = wid >> 1, wn = wid & 1;
  const int sq = tid % Q, spx = tid / Q;     // staging: channel quad, first pixel (items: pixel spx + (256 / Q) * i)
  float sc_dy = 1.f, sc_out = 1.f;
  if (NP == 2) {
    const unsigned mbits = __builtin_amdgcn_readfirstlane(*dy_max);
    const int e = (int)((mbits >> 23) & 0xffu) - 126;               // max = f 2^e, f in [0.5, 1)
    int k = (mbits >> 23) == 0u ? 0 : 14 - e;
    k = k < -100 ? -100 : (k > 100 ? 100 : k);
    sc_dy = __uint_as_float((unsigned)(127 + k) << 23);
    sc_out = __uint_as_float((unsigned)(127 - k - X2H_KX) << 23);
  }

  int bx, split;
  wgrad_blk(bi, bx, split);
  const int t = bx % g.ntaps;
  const int rest = bx / g.ntaps;
  const int ciTile = rest % ciT, coTile = rest / ciT;
  const int co0 = coTile * BT, ci0 = ciTile * BT;
  const bool odd = ((split0 + split) & 1) != 0;

  const long long TYX = (long long)g.TY * g.TX;
  const long long M = (long long)g.N * TYX;
  const long long mbeg = (long long)split * mps;
  const long long mend = mbeg + mps < M ? mbeg + mps : M;
  const int oy = g.offy[t], ox = g.offx[t], wo = g.wofs[t];

  int p_n[NI], p_ty[NI], p_tx[NI];       // source pixel of this thread's X items, advanced incrementally
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    long long m = mbeg + spx + (256 / Q) * i;
    if (m >= M) m = M - 1;
    const int n = (int)(m / TYX);
    const int rem = (int)(m - (long long)n * TYX);
    p_n[i] = n; p_ty[i] = rem / g.TX; p_tx[i] = rem - p_ty[i] * g.TX;
  }
  const int kc = ci0 + sq * 4, cc = co0 + sq * 4;
  const bool kok = kc < g.K, ccok = cc < g.Cout;
  const bool has_pro = pro != nullptr;
  float4 p_sc = zero4(), p_sh = zero4();
  if (has_pro && kok) { p_sc = ld4(pro + kc); p_sh = ld4(pro + g.K + kc); }
  const int n0 = (int)(mbeg / TYX);
  const long long img_elems = (long long)g.SH * g.SW * g.src_cstride;
  const __amdgpu_buffer_rsrc_t rsX = make_rsrc(src + (long long)n0 * img_elems, ((long long)g.N - n0) * img_elems * 4);
  const __amdgpu_buffer_rsrc_t rsD = make_rsrc(dy + mbeg * dy_cstride, (mend - mbeg) * (long long)dy_cstride * 4);

  float4 rs[NSLOT];                       // slots 0..NI-1: dy items, NI..: source items
  float lim[NI];
  int l_row = spx;                        // first row (relative to mbeg) of the chunk being loaded, + this thread's pixel
  auto load_slot = [&](int sl) {
    if (sl < NI) {
      const int mr = l_row + (256 / Q) * sl;                       // rows >= mend - mbeg fall outside rsD -> 0
      rs[sl] = bld4(rsD, ccok ? (unsigned)(mr * dy_cstride + cc) * 4u : OOB);
    } else {
      const int i = sl - NI;
      const long long m = mbeg + l_row + (256 / Q) * i;
      const int iy = p_ty[i] * g.sy + oy, ix = p_tx[i] * g.sx + ox;
      const bool ok = m < mend && kok && (unsigned)iy < (unsigned)g.SH && (unsigned)ix < (unsigned)g.SW;
      const int off = (((p_n[i] - n0) * g.SH + iy) * g.SW + ix) * g.src_cstride + kc;
      rs[sl] = bld4(rsX, ok ? (unsigned)off * 4u : OOB);
      if (has_pro) lim[i] = ok ? __builtin_inff() : 0.f;
      p_tx[i] += CHP;
      while (p_tx[i] >= g.TX) { p_tx[i] -= g.TX; p_ty[i] += 1; }
      while (p_ty[i] >= g.TY) { p_ty[i] -= g.TY; p_n[i] += 1; }
    }
  };
  auto store_slot = [&](int sl, int buf) {
    float4 v = rs[sl];
    const int i = sl < NI ? sl : sl - NI;
    if (sl < NI) {
      if (odd) { v.x = -v.x; v.y = -v.y; v.z = -v.z; v.w = -v.w; }
    } else if (has_pro) {
      v = pro_apply(v, p_sc, p_sh, lim[i]);
    }
    unsigned char* q = sm + (sl < NI ? 0 : 2 * OPB) + buf * OPB + (spx + (256 / Q) * i) * ROW + sq * 8;
    if constexpr (NP == 3) {
      uint2 p1, p2, p3;
      split3_quad(v, p1, p2, p3);
      *reinterpret_cast<uint2*>(q) = p1;
      *reinterpret_cast<uint2*>(q + PIECE) = p2;
      *reinterpret_cast<uint2*>(q + 2 * PIECE) = p3;
    } else {
      uint2 p1, p2;
      split2h_quad(v, sl < NI ? sc_dy : (float)(1 << X2H_KX), p1, p2);
      *reinterpret_cast<uint2*>(q) = p1;
      *reinterpret_cast<uint2*>(q + PIECE) = p2;
    }
  };

  f32x16 acc[T][T];
#pragma unroll
  for (int a = 0; a < T; ++a)
#pragma unroll
    for (int b = 0; b < T; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const long long nch = mend > mbeg ? (mend - mbeg + CHP - 1) / CHP : 0;
#pragma unroll
  for (int sl = 0; sl < NSLOT; ++sl) load_slot(sl);
#pragma unroll
  for (int sl = 0; sl < NSLOT; ++sl) store_slot(sl, 0);
  l_row += CHP;
#pragma unroll
  for (int sl = 0; sl < NSLOT; ++sl) load_slot(sl);
  __syncthreads();

  // transposed-read lane address: lane 4q+p of a 16-lane group supplies row (pixel) q, columns (channels) 4p .. 4p+3;
  // group parity selects channels 0-15 / 16-31 of the 32-channel tile, lane half h the pixels 8h .. 8h+7 (two reads)
  const int tj = lane & 15;
  const int tr_off = (8 * h + (tj >> 2)) * ROW + (16 * ((lane >> 4) & 1) + 4 * (tj & 3)) * 2;
  constexpr int NTERM = NP == 3 ? 6 : 3;
  constexpr int PA[6] = {NP == 3 ? 0 : 1, NP == 3 ? 2 : 0, NP == 3 ? 1 : 0, 0, 1, 0};       // smallest products first
  constexpr int PB[6] = {NP == 3 ? 2 : 0, NP == 3 ? 0 : 1, NP == 3 ? 1 : 0, 1, 0, 0};
  constexpr int G = NTERM * T * T, G0 = T * T;
  bf16x8 fa[T][NP], fb[T][NP];
  auto frag = [&](const unsigned char* base) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 4 * ROW));
    const s16x8 v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    return __builtin_bit_cast(bf16x8, v);
  };
  auto mfma_range = [&](int lo, int hi) {
#pragma unroll
    for (int q = 0; q < G; ++q) {
      if (q < lo || q >= hi) continue;
      const int term = q / (T * T), a = (q / T) % T, b = q % T;
      if constexpr (NP == 3)
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][PA[term]], fb[b][PB[term]], acc[a][b], 0, 0, 0);
      else
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[a][PA[term]]),
                                                           __builtin_bit_cast(f16x8, fb[b][PB[term]]), acc[a][b], 0, 0, 0);
    }
  };
  for (long long ch = 0; ch < nch; ++ch) {
    const int buf = (int)(ch & 1);
    const unsigned char* Db = sm + buf * OPB + tr_off + (wm * T * 32) * 2;
    const unsigned char* Xb = sm + 2 * OPB + buf * OPB + tr_off + (wn * T * 32) * 2;
#pragma unroll
    for (int p = NP - 1; p >= 0; --p) {
#pragma unroll
      for (int a = 0; a < T; ++a) fa[a][p] = frag(Db + p * PIECE + a * 64);
#pragma unroll
      for (int b = 0; b < T; ++b) fb[b][p] = frag(Xb + p * PIECE + b * 64);
    }
    mfma_range(0, G0);
    l_row += CHP;
#pragma unroll
    for (int sl = 0; sl < NSLOT; ++sl) {
      store_slot(sl, buf ^ 1);
      load_slot(sl);
      mfma_range(G0 + sl * (G - G0) / NSLOT, G0 + (sl + 1) * (G - G0) / NSLOT);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }

  float* out = slab + (long long)(split0 + split) * g.Cout * g.wstride;
#pragma unroll
  for (int a = 0; a < T; ++a)
#pragma unroll
    for (int b = 0; b < T; ++b) {
      const int ci = ci0 + (wn * T + b) * 32 + l31;
      if (ci >= g.K) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + (wm * T + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (co < g.Cout) out[(long long)co * g.wstride + wo + ci] = (odd ? -acc[a][b][r] : acc[a][b][r]) * (NP == 2 ? sc_out : 1.f);
      }
    }
}

// ------------------------------------------------------------------------------------------------
// weight gradient of 3x3 / stride 1 / pad 1 convolutions on the bf16 matrix cores, ALL NINE TAPS per block (the split-bf16
// counterpart of conv_wgrad3x3_kernel).  Block = one 64(co) x 64(ci) tile and one range of chunks; a chunk is 16
// consecutive output pixels of one image row (TX % 16 == 0).  Per chunk the block stages dy[16][64] and the 3 x 18 pixel
// input halo once, both as [pixel][piece][channel] bf16 rows (see conv_wgrad_x3_kernel), and feeds 9 taps x 6 MFMAs per
// wave from them: tap (r, s) reads halo rows r*18 + s + pixel -- in the pixel-major image a tap shift is a row offset, so
// the transposing reads stay aligned.  62.7 KB LDS, two blocks per CU, register-staged write-after-barrier pipeline.
// Odd splits: (-dy) x and negated output, as in conv_wgrad_x3_kernel.
__global__ __launch_bounds__(256, 2)
void conv_wgrad3x3_x3_kernel(const float* __restrict__ src, const float* __restrict__ dy, float* __restrict__ slab,
                             const DcsConvGeom g, const int dy_cstride, const int split0, const int cps, const int ciT,
                             const float* __restrict__ pro) {
  constexpr int CHP = 16, HWP = CHP + 2, XROWS = 3 * HWP;      // 54 staged halo rows
  constexpr int PIECE = 128, ROW = 3 * PIECE + 64;             // 448 = 192 (mod 256)
  constexpr int NSD = 1, NSX = (XROWS * 16 + 255) / 256, NSLOT = NSD + NSX;
  constexpr int DB = CHP * ROW, XB = XROWS * ROW;
  __shared__ __attribute__((aligned(16))) unsigned char sm[2 * DB + 2 * XB];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int wm = wid >> 1, wn = wid & 1;
  const int lcol4 = tid & 15, lrow = tid >> 4;          // 16 float4 per 64-channel row

  const int ciTile = blockIdx.x % ciT, coTile = blockIdx.x / ciT;
  const int co0 = coTile * 64, ci0 = ciTile * 64;
  const int split = blockIdx.y;
  const bool odd = ((split0 + split) & 1) != 0;
  const int cpr = g.TX / CHP;                           // chunks per image row
  const int nchunks_total = g.N * g.TY * cpr;
  const int cbeg = split * cps;
  const int cend = cbeg + cps < nchunks_total ? cbeg + cps : nchunks_total;
  const int nch = cend > cbeg ? cend - cbeg : 0;

  int q_n, q_ty, q_tx;                                  // coordinates of the NEXT chunk to be loaded (uniform)
  {
    const int c = cbeg < nchunks_total ? cbeg : 0;
    const int per_img = g.TY * cpr;
    q_n = c / per_img;
    const int rem = c - q_n * per_img;
    q_ty = rem / cpr;
    q_tx = (rem - q_ty * cpr) * CHP;
  }
  const int n0 = q_n;
  const long long img_elems = (long long)g.SH * g.SW * g.src_cstride;
  const __amdgpu_buffer_rsrc_t rsX = make_rsrc(src + (long long)n0 * img_elems, ((long long)g.N - n0) * img_elems * 4);
  const long long mbeg = (long long)cbeg * CHP;
  const __amdgpu_buffer_rsrc_t rsD = make_rsrc(dy + mbeg * dy_cstride, (long long)nch * CHP * dy_cstride * 4);
  const int kc = ci0 + lcol4 * 4, cc = co0 + lcol4 * 4;
  const bool kok = kc < g.K, ccok = cc < g.Cout;
  const bool has_pro = pro != nullptr;
  float4 p_sc = zero4(), p_sh = zero4();
  if (has_pro && kok) { p_sc = ld4(pro + kc); p_sh = ld4(pro + g.K + kc); }
  float lim[NSX];

  int xs_hr[NSX], xs_hx[NSX];
  bool xs_ok[NSX];
#pragma unroll
  for (int k = 0; k < NSX; ++k) {
    const int row = (tid + 256 * k) >> 4;
    xs_ok[k] = row < XROWS;
    xs_hr[k] = row / HWP;
    xs_hx[k] = row - xs_hr[k] * HWP;
  }

  float4 rs[NSLOT];
  int l_chunk = 0;
  auto load_slot = [&](int sl) {
    if (sl < NSD) {
      const int mr = l_chunk * CHP + lrow;              // rows beyond the split fall outside rsD -> 0
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
    q_tx += CHP;
    if (q_tx >= g.TX) { q_tx = 0; q_ty += 1; }
    if (q_ty >= g.TY) { q_ty = 0; q_n += 1; }
  };
  auto store_slot = [&](int sl, int buf) {
    float4 v = rs[sl];
    unsigned char* q;
    if (sl < NSD) {
      if (odd) { v.x = -v.x; v.y = -v.y; v.z = -v.z; v.w = -v.w; }
      q = sm + buf * DB + lrow * ROW + lcol4 * 8;
    } else {
      const int k = sl - NSD;
      if (!xs_ok[k]) return;
      if (has_pro) v = pro_apply(v, p_sc, p_sh, lim[k]);
      q = sm + 2 * DB + buf * XB + ((tid + 256 * k) >> 4) * ROW + lcol4 * 8;
    }
    uint2 p1, p2, p3;
    split3_quad(v, p1, p2, p3);
    *reinterpret_cast<uint2*>(q) = p1;
    *reinterpret_cast<uint2*>(q + PIECE) = p2;
    *reinterpret_cast<uint2*>(q + 2 * PIECE) = p3;
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

  const int tj = lane & 15;
  const int tr_off = (8 * h + (tj >> 2)) * ROW + (16 * ((lane >> 4) & 1) + 4 * (tj & 3)) * 2;
  constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
  auto frag = [&](const unsigned char* base) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 4 * ROW));
    const s16x8 v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    return __builtin_bit_cast(bf16x8, v);
  };
  for (int ch = 0; ch < nch; ++ch) {
    const int buf = ch & 1;
    const unsigned char* Db = sm + buf * DB + tr_off + (wm * 32) * 2;
    const unsigned char* Xb = sm + 2 * DB + buf * XB + tr_off + (wn * 32) * 2;
    bf16x8 fa[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) fa[p] = frag(Db + p * PIECE);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int r = t / 3, sx = t % 3;
      bf16x8 fb[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) fb[p] = frag(Xb + (r * HWP + sx) * ROW + p * PIECE);
#pragma unroll
      for (int term = 0; term < 6; ++term)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[term]], fb[PB[term]], acc[t], 0, 0, 0);
      if (t < NSLOT) {               // stage chunk ch+1 into the other buffer and refill the registers with chunk ch+2
        store_slot(t, buf ^ 1);
        load_slot(t);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    advance_chunk();
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
        if (co < g.Cout) out[(long long)co * g.wstride + t * g.K + ci] = odd ? -acc[t][r] : acc[t][r];
      }
  }
}

// ------------------------------------------------------------------------------------------------
// conv_wgrad3x3_x3_kernel with a ROLLING input window.  Timing experiments on that kernel (loop parts switched off, results
// discarded) put 35 % of its time into staging (global load, split, LDS write of dy[16][64] and the 3 x 18-pixel halo per
// chunk) and the MFMA-only loop at 286-304 TFLOP/s.  Here a block walks DOWN a 16-pixel-wide column strip: two of the
// three halo rows of the next chunk are already resident, so a chunk stages one new input row (18 pixels) and dy instead of
// three rows -- the halo lives in a ring of four row slots (row & 3: the slot of row ty+2 is free while rows ty-1..ty+1 are
// read).  Only the first chunk of a strip (or of a split) loads three rows, behind two barriers.  Work units are
// (image, strip, row) in that order; a split is a range of units.
// NP = 2 (dy_max given): both operands as two fp16 pieces, three MFMAs per product (see split2h_quad): the input is scaled
// by 2^X2H_KX like in the forward kernels, dy by the power of two that puts its maximum (dy_max: the device word
// dcs_bn_bwd_apply leaves) into [2^13, 2^14), the slab by the inverse of both.
template <int NP = 3>
__device__ __forceinline__
void conv_wgrad3x3_x3r_body(const float* __restrict__ src, const float* __restrict__ dy, float* __restrict__ slab,
                              const DcsConvGeom& g, const int dy_cstride, const int split0, const int cps, const int ciT,
                              const float* __restrict__ pro, const BlkId bi, const unsigned* __restrict__ dy_max = nullptr) {
  constexpr int CHP = 16, HWP = CHP + 2;
  constexpr int PIECE = 128, ROW = NP * PIECE + 64;            // 448 = 192, 320 = 64 (mod 256): conflict-free transposing reads
  constexpr int NSX = (3 * HWP * 16 + 255) / 256;              // 256-thread passes over up to three halo rows
  constexpr int DB = CHP * ROW, XB = 4 * HWP * ROW;            // dy double buffer, ring of four halo rows
  __shared__ __attribute__((aligned(16))) unsigned char sm[2 * DB + XB];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int wm = wid >> 1, wn = wid & 1;
  const int lcol4 = tid & 15, lrow = tid >> 4;
  float sc_x = (float)(1 << X2H_KX), sc_dy = 1.f, sc_out = 1.f;
  if (NP == 2) {
    const unsigned mbits = __builtin_amdgcn_readfirstlane(*dy_max);
    const int e = (int)((mbits >> 23) & 0xffu) - 126;               // max = f 2^e, f in [0.5, 1)
    int k = (mbits >> 23) == 0u ? 0 : 14 - e;
    k = k < -100 ? -100 : (k > 100 ? 100 : k);
    sc_dy = __uint_as_float((unsigned)(127 + k) << 23);
    sc_out = __uint_as_float((unsigned)(127 - k - X2H_KX) << 23);
  }

  int bx, split;
  wgrad_blk(bi, bx, split);
  const int ciTile = bx % ciT, coTile = bx / ciT;
  const int co0 = coTile * 64, ci0 = ciTile * 64;
  const bool odd = ((split0 + split) & 1) != 0;
  const int cpr = g.TX / CHP;                           // strips per image
  const int nunits_total = g.N * cpr * g.TY;
  const int ubeg = split * cps;
  const int uend = ubeg + cps < nunits_total ? ubeg + cps : nunits_total;

  const int n0 = (ubeg < nunits_total ? ubeg : 0) / (g.TY * cpr);      // first image this split touches
  const long long img_elems = (long long)g.SH * g.SW * g.src_cstride;
  const long long dimg_elems = (long long)g.TY * g.TX * dy_cstride;
  const __amdgpu_buffer_rsrc_t rsX = make_rsrc(src + (long long)n0 * img_elems, ((long long)g.N - n0) * img_elems * 4);
  const __amdgpu_buffer_rsrc_t rsD = make_rsrc(dy + (long long)n0 * dimg_elems, ((long long)g.N - n0) * dimg_elems * 4);
  const int kc = ci0 + lcol4 * 4, cc = co0 + lcol4 * 4;
  const bool kok = kc < g.K, ccok = cc < g.Cout;
  const bool has_pro = pro != nullptr;
  float4 p_sc = zero4(), p_sh = zero4();
  if (has_pro && kok) { p_sc = ld4(pro + kc); p_sh = ld4(pro + g.K + kc); }

  // halo passes: element e = tid + 256 k -> (row in the list rr, pixel hx); one row = pass 0 + 32 lanes of pass 1
  int xs_rr[NSX], xs_hx[NSX];
#pragma unroll
  for (int k = 0; k < NSX; ++k) {
    const int hrow = (tid + 256 * k) >> 4;
    xs_rr[k] = hrow / HWP;
    xs_hx[k] = hrow - xs_rr[k] * HWP;
  }

  // Two register sets of three slots (dy, 16 halo pixels, 2 halo pixels): a unit's loads are issued TWO iterations before
  // its registers are split and written to LDS (one iteration = 54 MFMAs per wave is shorter than a loaded global round
  // trip: with one set the store waited on the loads for ~15 % of the kernel).  Every iteration issues exactly three
  // loads -- beyond the segment they go out of range -- so that the compiler can count the loads in flight (vmcnt(3))
  // instead of draining them all.
  float4 rs[2][3];
  float lim[2][2];
  int ring[2] = {0, 0};               // ring slot the held halo row goes to
  auto prefetch = [&](auto S, int un, int utx, int uty, bool live) {       // dy of row uty and halo row uty + 1
    constexpr int s_ = decltype(S)::value;
    const int off = (((un - n0) * g.TY + uty) * g.TX + utx + lrow) * dy_cstride + cc;
    rs[s_][0] = bld4(rsD, (ccok && live) ? (unsigned)off * 4u : OOB);
    const int iy = uty + 1;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int ix = utx - 1 + xs_hx[k];
      const bool ok = xs_rr[k] == 0 && kok && live && (unsigned)iy < (unsigned)g.SH && (unsigned)ix < (unsigned)g.SW;
      const int xoff = (((un - n0) * g.SH + iy) * g.SW + ix) * g.src_cstride + kc;
      rs[s_][1 + k] = bld4(rsX, ok ? (unsigned)xoff * 4u : OOB);
      if (has_pro) lim[s_][k] = ok ? __builtin_inff() : 0.f;
    }
    ring[s_] = (iy + 4) & 3;
  };
  auto split_store = [&](float4 v, unsigned char* q, const float sc) {
    if constexpr (NP == 3) {
      uint2 p1, p2, p3;
      split3_quad(v, p1, p2, p3);
      *reinterpret_cast<uint2*>(q) = p1;
      *reinterpret_cast<uint2*>(q + PIECE) = p2;
      *reinterpret_cast<uint2*>(q + 2 * PIECE) = p3;
    } else {
      uint2 p1, p2;
      split2h_quad(v, sc, p1, p2);
      *reinterpret_cast<uint2*>(q) = p1;
      *reinterpret_cast<uint2*>(q + PIECE) = p2;
    }
  };
  auto store_slot = [&](auto S, int sl, int buf) {
    constexpr int s_ = decltype(S)::value;
    float4 v = rs[s_][sl];
    if (sl == 0) {
      if (odd) { v.x = -v.x; v.y = -v.y; v.z = -v.z; v.w = -v.w; }
      split_store(v, sm + buf * DB + lrow * ROW + lcol4 * 8, sc_dy);
    } else {
      const int k = sl - 1;
      if (xs_rr[k] != 0) return;
      if (has_pro) v = pro_apply(v, p_sc, p_sh, lim[s_][k]);
      split_store(v, sm + 2 * DB + (ring[s_] * HWP + xs_hx[k]) * ROW + lcol4 * 8, sc_x);
    }
  };
  // a segment's first unit: dy and rows ty - 1, ty, ty + 1, loaded and stored here and now
  auto cold_stage = [&](int un, int utx, int uty, int buf) {
    {
      const int off = (((un - n0) * g.TY + uty) * g.TX + utx + lrow) * dy_cstride + cc;
      float4 v = bld4(rsD, ccok ? (unsigned)off * 4u : OOB);
      if (odd) { v.x = -v.x; v.y = -v.y; v.z = -v.z; v.w = -v.w; }
      split_store(v, sm + buf * DB + lrow * ROW + lcol4 * 8, sc_dy);
    }
#pragma unroll
    for (int k = 0; k < NSX; ++k) {
      if (xs_rr[k] >= 3) continue;
      const int iy = uty - 1 + xs_rr[k], ix = utx - 1 + xs_hx[k];
      const bool ok = kok && (unsigned)iy < (unsigned)g.SH && (unsigned)ix < (unsigned)g.SW;
      const int xoff = (((un - n0) * g.SH + iy) * g.SW + ix) * g.src_cstride + kc;
      float4 v = bld4(rsX, ok ? (unsigned)xoff * 4u : OOB);
      if (has_pro) v = pro_apply(v, p_sc, p_sh, ok ? __builtin_inff() : 0.f);
      split_store(v, sm + 2 * DB + (((iy + 4) & 3) * HWP + xs_hx[k]) * ROW + lcol4 * 8, sc_x);
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int tj = lane & 15;
  const int tr_off = (8 * h + (tj >> 2)) * ROW + (16 * ((lane >> 4) & 1) + 4 * (tj & 3)) * 2;
  constexpr int NTERM = NP == 3 ? 6 : 3;
  constexpr int PA[6] = {NP == 3 ? 0 : 1, NP == 3 ? 2 : 0, NP == 3 ? 1 : 0, 0, 1, 0};       // smallest products first
  constexpr int PB[6] = {NP == 3 ? 2 : 0, NP == 3 ? 0 : 1, NP == 3 ? 1 : 0, 1, 0, 0};
  auto frag = [&](const unsigned char* base) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 4 * ROW));
    const s16x8 v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    return __builtin_bit_cast(bf16x8, v);
  };

  int dbuf = 0;                         // dy buffer of the unit about to be computed
  for (int u = ubeg; u < uend;) {       // one segment = the rows of ONE column strip that fall into this split
    const int strip = u / g.TY;
    const int ty0 = u - strip * g.TY;
    const int sn = strip / cpr, stx = (strip - sn * cpr) * CHP;
    const int seg_end = (strip + 1) * g.TY < uend ? (strip + 1) * g.TY : uend;
    const int cnt = seg_end - u;
    __syncthreads();                    // the previous segment's readers are done with the ring and the dy buffers
    cold_stage(sn, stx, ty0, dbuf);
    prefetch(S1{}, sn, stx, ty0 + 1, 1 < cnt);
    prefetch(S0{}, sn, stx, ty0 + 2, 2 < cnt);
    __syncthreads();
    // iteration j computes row ty0 + j from dy buffer dbuf ^ (j & 1); set P = (j + 1) & 1 holds row ty0 + j + 1 and is
    // re-loaded with row ty0 + j + 3
    auto step = [&](auto P, const int j) {
      const int buf = dbuf ^ (j & 1);
      const int c_ty = ty0 + j;
      const bool has_next = j + 1 < cnt;
      const unsigned char* Db = sm + buf * DB + tr_off + (wm * 32) * 2;
      const unsigned char* Xb = sm + 2 * DB + tr_off + (wn * 32) * 2;
      bf16x8 fa[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) fa[p] = frag(Db + p * PIECE);
      // the input fragments of tap t + 1 are read while tap t multiplies (software pipeline over the unrolled taps)
      auto load_fb = [&](int t, bf16x8 (&fb)[NP]) {
        const int r = t / 3, sx = t % 3;
        const int slot = (c_ty + r + 3) & 3;                            // ring slot of input row c_ty - 1 + r
#pragma unroll
        for (int p = 0; p < NP; ++p) fb[p] = frag(Xb + (slot * HWP + sx) * ROW + p * PIECE);
      };
      bf16x8 fbq[2][NP];
      load_fb(0, fbq[0]);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        if (t + 1 < 9) load_fb(t + 1, fbq[(t + 1) & 1]);
#pragma unroll
        for (int term = 0; term < NTERM; ++term) {
          if constexpr (NP == 3)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[term]], fbq[t & 1][PB[term]], acc[t], 0, 0, 0);
          else
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[PA[term]]),
                                                            __builtin_bit_cast(f16x8, fbq[t & 1][PB[term]]), acc[t], 0, 0, 0);
        }
        if (t < 3) {                    // row j + 1 (loaded two iterations ago): registers -> LDS in the shadow of the MFMAs
          if (has_next) store_slot(P, t, buf ^ 1);
          if (t == 2) prefetch(P, sn, stx, c_ty + 3, j + 3 < cnt);       // row j + 3 into the freed set
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __syncthreads();
    };
    for (int j = 0; j < cnt; j += 2) {
      step(S1{}, j);
      if (j + 1 < cnt) step(S0{}, j + 1);
    }
    dbuf ^= cnt & 1;
    u = seg_end;
  }

  float* out = slab + (long long)(split0 + split) * g.Cout * g.wstride;
  const int ci = ci0 + wn * 32 + l31;
  if (ci < g.K) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (co < g.Cout) out[(long long)co * g.wstride + t * g.K + ci] = (odd ? -acc[t][r] : acc[t][r]) * (NP == 2 ? sc_out : 1.f);
      }
  }
}

// ------------------------------------------------------------------------------------------------
// weight gradient of the 7x7 / stride 2 / pad 3 stem on the bf16 matrix cores (split-bf16 counterpart of
// stem_wgrad_kernel): all seven filter rows per block, a chunk = 16 consecutive output pixels of one output row.  dy is
// staged as [pixel][piece][64 channels]; the 7 x 38-pixel input patch as one contiguous bf16 run per (filter row, piece).
// The im2col row of output pixel p for filter row r is patch[r][8p .. 8p+31] (7 px x 4 ch + one pad pixel whose products
// land in the padded weight slots): for the transposing read that is a "matrix" whose rows (pixels) are 16 bytes apart --
// every lane of a 16-lane group supplies its own address, so the Toeplitz structure needs no im2col copy here either.
// Wave w owns filter rows 2w, 2w+1 (row 7 does not exist) x both 32-channel output tiles.  Odd splits: (-dy), negated
// output (rounding-bias cancellation, see conv_wgrad_x3_kernel).
// NP = 2 (dy_max given): fp16 two-piece form -- the image by 2^X2H_KX, dy by its maximum (conv_wgrad3x3_x3r_body<2>).
template <int NP = 3>
__device__ __forceinline__
void stem_wgrad_x3_body(const float* __restrict__ src, const float* __restrict__ dy, float* __restrict__ slab,
                          const DcsConvGeom& g, const int dy_cstride, const int split0, const int cps, const BlkId bi,
                          const unsigned* __restrict__ dy_max = nullptr) {
  constexpr int CHP = 16, PWP = 38;                    // pixels per chunk, patch width in pixels
  constexpr int PIECE = 128, ROW = NP * PIECE + 64;    // dy image: 448- / 320-byte rows
  constexpr int PL = PWP * 8;                          // bytes of one (filter row, piece) run: 38 px x 4 ch x 2 B = 304
  constexpr int DB = CHP * ROW, PB = 7 * NP * PL;
  __shared__ __attribute__((aligned(16))) unsigned char sm[2 * DB + 2 * PB];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int lcol4 = tid & 15, lrow = tid >> 4;
  float sc_dy = 1.f, sc_out = 1.f;
  if (NP == 2) {
    const unsigned mbits = __builtin_amdgcn_readfirstlane(*dy_max);
    const int e = (int)((mbits >> 23) & 0xffu) - 126;               // max = f 2^e, f in [0.5, 1)
    int k = (mbits >> 23) == 0u ? 0 : 14 - e;
    k = k < -100 ? -100 : (k > 100 ? 100 : k);
    sc_dy = __uint_as_float((unsigned)(127 + k) << 23);
    sc_out = __uint_as_float((unsigned)(127 - k - X2H_KX) << 23);
  }
  const int split = bi.x;
  const bool odd = ((split0 + split) & 1) != 0;
  const int cpr = g.TX / CHP;
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
    q_tx = (rem - q_ty * cpr) * CHP;
  }
  const int n0 = q_n;
  const long long img_elems = (long long)g.SH * g.SW * 4;
  const __amdgpu_buffer_rsrc_t rsX = make_rsrc(src + (long long)n0 * img_elems, ((long long)g.N - n0) * img_elems * 4);
  const long long mbeg = (long long)cbeg * CHP;
  const __amdgpu_buffer_rsrc_t rsD = make_rsrc(dy + mbeg * dy_cstride, (long long)nch * CHP * dy_cstride * 4);

  int ps_r[2], ps_c[2];
  bool ps_ok[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int e = tid + 256 * k;
    ps_ok[k] = e < 7 * PWP;
    ps_r[k] = e / PWP;
    ps_c[k] = e - ps_r[k] * PWP;
  }

  float4 rs[3];
  int l_chunk = 0;
  auto load_slot = [&](int sl) {
    if (sl == 0) {
      const int mr = l_chunk * CHP + lrow;
      rs[0] = bld4(rsD, (unsigned)(mr * dy_cstride + lcol4 * 4) * 4u);
    } else {
      const int k = sl - 1;
      const int iy = 2 * q_ty - 3 + ps_r[k], ix = 2 * q_tx - 3 + ps_c[k];
      const bool ok = ps_ok[k] && l_chunk < nch && (unsigned)iy < (unsigned)g.SH && (unsigned)ix < (unsigned)g.SW;
      rs[sl] = bld4(rsX, ok ? (unsigned)((((q_n - n0) * g.SH + iy) * g.SW + ix) * 4) * 4u : OOB);
    }
  };
  auto advance_chunk = [&]() {
    l_chunk += 1;
    q_tx += CHP;
    if (q_tx >= g.TX) { q_tx = 0; q_ty += 1; }
    if (q_ty >= g.TY) { q_ty = 0; q_n += 1; }
  };
  auto store_slot = [&](int sl, int buf) {
    float4 v = rs[sl];
    unsigned char* q;
    int pstride;
    if (sl == 0) {
      if (odd) { v.x = -v.x; v.y = -v.y; v.z = -v.z; v.w = -v.w; }
      q = sm + buf * DB + lrow * ROW + lcol4 * 8;
      pstride = PIECE;
    } else {
      const int k = sl - 1;
      if (!ps_ok[k]) return;
      q = sm + 2 * DB + buf * PB + ps_r[k] * NP * PL + ps_c[k] * 8;
      pstride = PL;
    }
    if constexpr (NP == 3) {
      uint2 p1, p2, p3;
      split3_quad(v, p1, p2, p3);
      *reinterpret_cast<uint2*>(q) = p1;
      *reinterpret_cast<uint2*>(q + pstride) = p2;
      *reinterpret_cast<uint2*>(q + 2 * pstride) = p3;
    } else {
      uint2 p1, p2;
      split2h_quad(v, sl == 0 ? sc_dy : (float)(1 << X2H_KX), p1, p2);
      *reinterpret_cast<uint2*>(q) = p1;
      *reinterpret_cast<uint2*>(q + pstride) = p2;
    }
  };

  f32x16 acc[2][2];                         // [filter row 2w + a][output-channel tile]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

#pragma unroll
  for (int sl = 0; sl < 3; ++sl) load_slot(sl);
  advance_chunk();
#pragma unroll
  for (int sl = 0; sl < 3; ++sl) store_slot(sl, 0);
#pragma unroll
  for (int sl = 0; sl < 3; ++sl) load_slot(sl);
  advance_chunk();
  __syncthreads();

  const int tj = lane & 15;
  const int prow = 8 * h + (tj >> 2);                               // pixel (k) row this lane addresses, + 4 for the 2nd read
  const int ccol = 16 * ((lane >> 4) & 1) + 4 * (tj & 3);            // first of its four columns
  const int d_off = prow * ROW + ccol * 2;
  const int p_off = (8 * prow + ccol) * 2;
  constexpr int NTERM = NP == 3 ? 6 : 3;
  constexpr int PA[6] = {NP == 3 ? 0 : 1, NP == 3 ? 2 : 0, NP == 3 ? 1 : 0, 0, 1, 0};       // smallest products first
  constexpr int PBt[6] = {NP == 3 ? 2 : 0, NP == 3 ? 0 : 1, NP == 3 ? 1 : 0, 1, 0, 0};
  auto frag = [&](const unsigned char* base, int second) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + second));
    const s16x8 v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    return __builtin_bit_cast(bf16x8, v);
  };
  const int r0 = 2 * wid;
  const bool two = r0 + 1 < 7;
  for (int ch = 0; ch < nch; ++ch) {
    const int buf = ch & 1;
    const unsigned char* Db = sm + buf * DB + d_off;
    const unsigned char* Pb = sm + 2 * DB + buf * PB + p_off;
    bf16x8 fa[2][NP], fb[2][NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
#pragma unroll
      for (int b = 0; b < 2; ++b) fa[b][p] = frag(Db + p * PIECE + b * 64, 4 * ROW);
      fb[0][p] = frag(Pb + (r0 * NP + p) * PL, 64);                   // 4 pixels further = 4 x 16 bytes
      fb[1][p] = frag(Pb + ((two ? r0 + 1 : r0) * NP + p) * PL, 64);
    }
#pragma unroll
    for (int term = 0; term < NTERM; ++term) {
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        if constexpr (NP == 3) {
          acc[0][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[b][PA[term]], fb[0][PBt[term]], acc[0][b], 0, 0, 0);
          if (two) acc[1][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[b][PA[term]], fb[1][PBt[term]], acc[1][b], 0, 0, 0);
        } else {
          acc[0][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[b][PA[term]]),
                                                             __builtin_bit_cast(f16x8, fb[0][PBt[term]]), acc[0][b], 0, 0, 0);
          if (two) acc[1][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[b][PA[term]]),
                                                                      __builtin_bit_cast(f16x8, fb[1][PBt[term]]), acc[1][b], 0, 0, 0);
        }
      }
      if (term < 3) {                      // three staging slots behind the first three MFMA groups
        store_slot(term, buf ^ 1);
        load_slot(term);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    advance_chunk();
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
        out[(long long)co * g.wstride + r * 32 + l31] = (odd ? -acc[a][b][q] : acc[a][b][q]) * (NP == 2 ? sc_out : 1.f);
      }
  }
}

bool wgrad3x3_x3_eligible(const DcsConvGeom* g) {
  if (g->stem || g->ntaps != 9 || g->sy != 1 || g->sx != 1 || (g->TX & 15) != 0 || g->wstride != 9 * g->K) return false;
  for (int t = 0; t < 9; ++t)
    if (g->offy[t] != t / 3 - 1 || g->offx[t] != t % 3 - 1 || g->wofs[t] != t * g->K) return false;
  return g->SH == g->TY && g->SW == g->TX;
}

// ------------------------------------------------------------------------------------------------
// 3x3 / stride 1 convolutions (forward and data gradient) with the INPUT HALO resident in LDS: a block computes
// TH image rows x 32 pixels x BN channels; per 16-channel chunk it stages the (TH+2) x 34-pixel halo ONCE (split into bf16
// pieces, prologue applied) and all nine taps read it -- in the [pixel][piece][16 channels] image a tap is a row offset
// of the fragment address.  Against conv_gather_x3_kernel that is 6x fewer activation loads, splits and LDS writes per
// MFMA (the weights still stream per tap through a double buffer); on a power-limited chip fewer bytes moved per MFMA
// is also clock.  Rounding-bias cancellation: chunk parity selects the accumulator set as before, but the sign now sits
// on the WEIGHT pieces of odd chunks (sign bits flipped during the copy), because the halo serves both parities.
template <int BN, int TH>
__global__ __launch_bounds__(256, 2)
void conv3x3_x3_kernel(const float* __restrict__ src, const unsigned char* __restrict__ wsp, const float* __restrict__ bias,
                       float* __restrict__ dst, const DcsConvGeom g, const int accumulate, const int ntiles,
                       float* __restrict__ stats, const BnBwdEpi bnb, const float* __restrict__ pro) {
  constexpr int BM = 32 * TH, HWD = 34, HROWS = (TH + 2) * HWD;
  constexpr int WN = BM == 256 ? 1 : 2, WM = 4 / WN, TM = 2, TN = BN / (WN * 32);
  constexpr int A_BYTES = HROWS * X3_ROWB, B_BYTES = BN * X3_ROWB;
  constexpr int SMEM_FLOATS = (A_BYTES + 2 * B_BYTES) / 4;
  constexpr int NH = (HROWS * 4 + 255) / 256;     // halo slots per thread
  constexpr int NB = (BN * 6 + 255) / 256;        // weight slots per thread
  static_assert(TM * WM * 32 == BM && (BN == 64 || BN == 128), "unsupported tile");

  __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
  unsigned char* const sm = reinterpret_cast<unsigned char*>(smem);
  __shared__ long long rowoff[BM];
  __shared__ int s_ho[9], s_wo[9];
  __shared__ __attribute__((aligned(16))) float s_pro[2 * DCS_PRO_MAXK];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int wm = wid / WN, wn = wid % WN;
  const int lcol4 = tid & 3;
  const bool has_pro = pro != nullptr;
  if (has_pro)
    for (int e = tid; e < 2 * g.K; e += 256) s_pro[(e < g.K ? 0 : DCS_PRO_MAXK - g.K) + e] = pro[e];

  const int bid = dcs_xcd_remap(blockIdx.x, gridDim.x);
  const int ntile = bid % ntiles, mtile = bid / ntiles;
  const int co0 = ntile * BN;
  const int tpx = g.TX >> 5, tpy = g.TY / TH;                    // tiles per row / column of one image
  const int n = mtile / (tpx * tpy);
  const int trem = mtile - n * (tpx * tpy);
  const int y0 = (trem / tpx) * TH, x0 = (trem % tpx) << 5;

  if (tid < BM) {
    const int ry = tid >> 5, px = tid & 31;
    rowoff[tid] = (((long long)n * g.DH + (y0 + ry)) * g.DW + (x0 + px)) * g.dst_cstride;
  }
  if (tid < 9) {
    s_ho[tid] = (g.offy[tid] * HWD + g.offx[tid]) * X3_ROWB;
    s_wo[tid] = (g.wofs[tid] >> 4) * 96;
  }
  const long long img_elems = (long long)g.SH * g.SW * g.src_cstride;
  const int wchunks = g.wstride >> 4;
  const __amdgpu_buffer_rsrc_t rsA = make_rsrc(src + (long long)n * img_elems, img_elems * 4);
  const __amdgpu_buffer_rsrc_t rsB = make_rsrc(reinterpret_cast<const float*>(wsp), (long long)g.Cout * wchunks * 96);

  int h_off[NH];                      // element offset of the halo pixel's channel quad (chunk 0) in the image, or -1
  float lim[NH];
#pragma unroll
  for (int j = 0; j < NH; ++j) {
    const int hrow = (tid + 256 * j) >> 2;
    const int hy = hrow / HWD, hx = hrow - hy * HWD;
    const int iy = y0 + hy - 1, ix = x0 + hx - 1;
    const bool ok = hrow < HROWS && (unsigned)iy < (unsigned)g.SH && (unsigned)ix < (unsigned)g.SW;
    h_off[j] = ok ? (iy * g.SW + ix) * g.src_cstride + lcol4 * 4 : -1;
    lim[j] = ok ? __builtin_inff() : 0.f;
  }
  int b_off[NB], b_lds[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int e = tid + 256 * j;
    const int brow = e / 6, bs = e - brow * 6;
    const bool ok = e < BN * 6 && co0 + brow < g.Cout;
    b_off[j] = ok ? ((co0 + brow) * wchunks) * 96 + bs * 16 : -1;
    b_lds[j] = e < BN * 6 ? brow * X3_ROWB + bs * 16 : -1;
  }
  __syncthreads();

  const int kch = g.K >> 4;
  const int nch = 9 * kch;

  float4 rh[NH];
  u32x4 rb[NB];
  auto load_halo = [&](int kc) {
#pragma unroll
    for (int j = 0; j < NH; ++j) rh[j] = bld4(rsA, h_off[j] >= 0 ? (unsigned)(h_off[j] + kc * 16) * 4u : OOB);
  };
  auto store_halo = [&](int kc) {
    float4 p_sc = zero4(), p_sh = zero4();
    if (has_pro) { p_sc = ld4(&s_pro[kc * 16 + lcol4 * 4]); p_sh = ld4(&s_pro[DCS_PRO_MAXK + kc * 16 + lcol4 * 4]); }
#pragma unroll
    for (int j = 0; j < NH; ++j) {
      const int hrow = (tid + 256 * j) >> 2;
      if (NH * 64 != HROWS && hrow >= HROWS) continue;
      float4 v = rh[j];
      if (has_pro) v = pro_apply(v, p_sc, p_sh, lim[j]);
      uint2 p1, p2, p3;
      split3_quad(v, p1, p2, p3);
      unsigned char* q = sm + hrow * X3_ROWB + lcol4 * 8;
      *reinterpret_cast<uint2*>(q) = p1;
      *reinterpret_cast<uint2*>(q + 32) = p2;
      *reinterpret_cast<uint2*>(q + 64) = p3;
    }
  };
  // weights of flattened chunk i = kc * 9 + t
  int lw_kc = 0, lw_t = 0;            // the next weight chunk to load
  auto load_w = [&]() {
    const int wo = s_wo[lw_t] + lw_kc * 96;
#pragma unroll
    for (int j = 0; j < NB; ++j)
      rb[j] = __builtin_amdgcn_raw_buffer_load_b128(rsB, b_off[j] >= 0 ? (unsigned)(b_off[j] + wo) : OOB, 0, 0);
    if (lw_kc * 9 + lw_t + 1 < nch) { lw_t += 1; if (lw_t == 9) { lw_t = 0; lw_kc += 1; } }
  };
  auto store_w = [&](auto NEG, int buf) {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      if (NB * 256 != BN * 6 && b_lds[j] < 0) continue;
      u32x4 v = rb[j];
      if (decltype(NEG)::value) { v.x ^= 0x80008000u; v.y ^= 0x80008000u; v.z ^= 0x80008000u; v.w ^= 0x80008000u; }
      *reinterpret_cast<u32x4*>(sm + A_BYTES + buf * B_BYTES + b_lds[j]) = v;
    }
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;

  f32x16 acc[2][TM][TN];
#pragma unroll
  for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[s_][a][b][r] = 0.f;

  load_halo(0);
  load_w();
  store_halo(0);
  store_w(P0{}, 0);
  load_w();                              // chunk 1
  if (kch > 1) load_halo(1);
  __syncthreads();

  constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
  bf16x8 fa[TM][3], fb[TN][3];
  // fragment base of this lane: halo row of output pixel (tile row wm*2 + a, column l31) at tap offset (0, 0)
  const unsigned char* Abase = sm + ((wm * 2 + 1) * HWD + l31 + 1) * X3_ROWB + h * 16;
  int kc = 0, t = 0;                     // the chunk being computed
  auto step = [&](auto PAR) {
    constexpr int par = decltype(PAR)::value;
    const unsigned char* Ab = Abase + s_ho[t];
    const unsigned char* Bb = sm + A_BYTES + par * B_BYTES + (wn * TN * 32 + l31) * X3_ROWB + h * 16;
#pragma unroll
    for (int p = 2; p >= 0; --p) {
#pragma unroll
      for (int a = 0; a < TM; ++a) fa[a][p] = *reinterpret_cast<const bf16x8*>(Ab + a * HWD * X3_ROWB + p * 32);
#pragma unroll
      for (int b = 0; b < TN; ++b) fb[b][p] = *reinterpret_cast<const bf16x8*>(Bb + b * 32 * X3_ROWB + p * 32);
    }
#pragma unroll
    for (int term = 0; term < 6; ++term) {
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
          acc[par][a][b] =
              __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][PA[term]], fb[b][PB[term]], acc[par][a][b], 0, 0, 0);
      if (term == 1) {                   // the next chunk's weights: registers -> the other buffer, then refill
        store_w(std::integral_constant<int, 1 - par>{}, par ^ 1);
        load_w();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    t += 1;
    if (t == 9) {                        // the halo of the next 16 channels replaces this one
      t = 0; kc += 1;
      if (kc < kch) {
        __syncthreads();
        store_halo(kc);
        if (kc + 1 < kch) load_halo(kc + 1);
      }
    }
    __syncthreads();
  };
  for (int i = 0; i < nch; i += 2) {
    step(P0{});
    if (i + 1 < nch) step(P1{});
  }
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][a][b][r] -= acc[1][a][b][r];

  conv_epilogue<BM, BN, TM, TN, WM, SMEM_FLOATS>(acc[0], smem, rowoff, bias, dst, g.dst_cstride, g.Cout, co0, accumulate, stats,
                                                 mtile, (unsigned long long)g.N * g.TY * g.TX, wm, wn, bnb, true);
}

// ------------------------------------------------------------------------------------------------
// conv3x3_x3_kernel with the WEIGHT fragments straight from global memory (no LDS staging of weights, no per-tap barrier):
// dcs_split_weight_frag lays the split weights out fragment-major -- for K chunk c, 32-row tile j, piece p the 64 lanes'
// 16-byte MFMA B fragments are 1 KiB contiguous -- followed by a second copy with the sign bits flipped (odd chunks read
// that one: rounding-bias cancellation as in conv3x3_x3_kernel, no XOR in the kernel).  A wave loads the fragments of
// the NEXT tap while it multiplies the current one; the halo is the only LDS tenant, so barriers remain only around its
// replacement every nine taps.
// NP = 3: three bf16 pieces, six products (above).  NP = 2: two fp16 pieces, three products (split2h_quad; the weight
// fragments come from dcs_split_weight_frag in its fp16 form): same structure, LDS row = [piece 1: 16 fp16][piece 2][16 B
// pad] = 80 B (5 x 16 B: conflict-free b128 reads), results scaled back by 2^-(X2H_KX + X2H_KW) before the epilogue.
// DCS_X3W_EXP (compile time, tools/x3w_parts.sh; never set in the product build): timing experiments that switch parts of
// the loop off (results discarded) -- 1: no halo replacement, 2: no weight loads, 4: no fragment reads, 8: no epilogue, 32: every weight load reads chunk 0 (L1 hits),
// 64 / 128: MODE.fp_round = +inf / toward zero for the whole kernel (does the MFMA accumulate follow it?  with 16),
// 16: ONE accumulator set, no rounding-bias cancellation (fp16 form only; results kept: for tools/conv_bias_probe.py).
#ifndef DCS_X3W_EXP
#define DCS_X3W_EXP 0
#endif
template <int BN, int TH, int NP = 3>
__device__ __forceinline__
void conv3x3_x3w_body(const float* __restrict__ src, const unsigned char* __restrict__ wfrag, const float* __restrict__ bias,
                        float* __restrict__ dst, const DcsConvGeom& g, const int accumulate, const int ntiles,
                        float* __restrict__ stats, const BnBwdEpi bnb, const float* __restrict__ pro, const int J,
                        const unsigned neg_off, const BlkId bi, const unsigned* __restrict__ src_max = nullptr) {
  constexpr int BM = 32 * TH, HWD = 34, HROWS = (TH + 2) * HWD;
  constexpr int WN = BM == 256 ? 1 : 2, WM = 4 / WN, TM = 2, TN = BN / (WN * 32);
  constexpr int ROWB = NP == 3 ? X3_ROWB : 80;
  constexpr int A_BYTES = HROWS * ROWB;
  constexpr int EPI_FLOATS = 4 * 32 * (TN * 32 + 4) + WM * BN * 2;
  constexpr int SMEM_FLOATS = (A_BYTES / 4) > EPI_FLOATS ? (A_BYTES / 4) : EPI_FLOATS;
  constexpr int NH = (HROWS * 4 + 255) / 256;
  static_assert(TM * WM * 32 == BM && (BN == 64 || BN == 128), "unsupported tile");

  __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
  unsigned char* const sm = reinterpret_cast<unsigned char*>(smem);
  __shared__ long long rowoff[BM];
  __shared__ int s_ho[9], s_wc[9];
  __shared__ __attribute__((aligned(16))) float s_pro[2 * DCS_PRO_MAXK];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int wm = wid / WN, wn = wid % WN;
  const int lcol4 = tid & 3;
  const bool has_pro = pro != nullptr;
  if (has_pro)
    for (int e = tid; e < 2 * g.K; e += 256) s_pro[(e < g.K ? 0 : DCS_PRO_MAXK - g.K) + e] = pro[e];
  // fp16 two-piece form: scale of the source operand (exact powers of two).  Activations: 2^X2H_KX; a tensor whose
  // maximum is known (data gradients): the power of two that puts it into [2^13, 2^14)
  float x2h_in = (float)(1 << X2H_KX), x2h_out = 1.f / (float)(1 << (X2H_KX + X2H_KW));
  if (NP == 2 && src_max != nullptr) {
    const unsigned mbits = __builtin_amdgcn_readfirstlane(*src_max);
    const int e = (int)((mbits >> 23) & 0xffu) - 126;               // max = f 2^e, f in [0.5, 1)
    int k = (mbits >> 23) == 0u ? 0 : 14 - e;
    k = k < -100 ? -100 : (k > 100 ? 100 : k);
    x2h_in = __uint_as_float((unsigned)(127 + k) << 23);
    x2h_out = __uint_as_float((unsigned)(127 - k - X2H_KW) << 23);
  }

  const int bid = dcs_xcd_remap(bi.x, bi.nx);
  const int ntile = bid % ntiles, mtile = bid / ntiles;
  const int co0 = ntile * BN;
  const int tpx = g.TX >> 5, tpy = g.TY / TH;
  const int n = mtile / (tpx * tpy);
  const int trem = mtile - n * (tpx * tpy);
  const int y0 = (trem / tpx) * TH, x0 = (trem % tpx) << 5;

  if (tid < BM) {
    const int ry = tid >> 5, px = tid & 31;
    rowoff[tid] = (((long long)n * g.DH + (y0 + ry)) * g.DW + (x0 + px)) * g.dst_cstride;
  }
  if (tid < 9) {
    s_ho[tid] = (g.offy[tid] * HWD + g.offx[tid]) * ROWB;
    s_wc[tid] = g.wofs[tid] >> 4;                                 // first K chunk of the tap
  }
  const long long img_elems = (long long)g.SH * g.SW * g.src_cstride;
  const int wchunks = g.wstride >> 4;
  const __amdgpu_buffer_rsrc_t rsA = make_rsrc(src + (long long)n * img_elems, img_elems * 4);
  const __amdgpu_buffer_rsrc_t rsB = make_rsrc(reinterpret_cast<const float*>(wfrag), 2ll * wchunks * J * NP * 1024);

  int h_off[NH];
  float lim[NH];
#pragma unroll
  for (int j = 0; j < NH; ++j) {
    const int hrow = (tid + 256 * j) >> 2;
    const int hy = hrow / HWD, hx = hrow - hy * HWD;
    const int iy = y0 + hy - 1, ix = x0 + hx - 1;
    const bool ok = hrow < HROWS && (unsigned)iy < (unsigned)g.SH && (unsigned)ix < (unsigned)g.SW;
    h_off[j] = ok ? (iy * g.SW + ix) * g.src_cstride + lcol4 * 4 : -1;
    lim[j] = ok ? __builtin_inff() : 0.f;
  }
  __syncthreads();

#if DCS_X3W_EXP & 64
  __builtin_amdgcn_s_setreg(2049, 1);      // experiment: fp32 rounding mode +inf -- does the MFMA's accumulate follow MODE?
  __builtin_amdgcn_s_setreg(2177, 1);
#endif
#if DCS_X3W_EXP & 128
  __builtin_amdgcn_s_setreg(2049, 3);      // ... toward zero
  __builtin_amdgcn_s_setreg(2177, 3);
#endif
  const int kch = g.K >> 4;
  const int nch = 9 * kch;
  const int jt0 = (co0 >> 5) + wn * TN;                            // first 32-row weight tile of this wave
  const unsigned lane16 = (unsigned)lane * 16u;

  float4 rh[NH];
  auto load_halo = [&](int kc) {
#pragma unroll
    for (int j = 0; j < NH; ++j) rh[j] = bld4(rsA, h_off[j] >= 0 ? (unsigned)(h_off[j] + kc * 16) * 4u : OOB);
  };
  auto store_halo = [&](int kc) {
    float4 p_sc = zero4(), p_sh = zero4();
    if (has_pro) { p_sc = ld4(&s_pro[kc * 16 + lcol4 * 4]); p_sh = ld4(&s_pro[DCS_PRO_MAXK + kc * 16 + lcol4 * 4]); }
#pragma unroll
    for (int j = 0; j < NH; ++j) {
      const int hrow = (tid + 256 * j) >> 2;
      if (NH * 64 != HROWS && hrow >= HROWS) continue;
      float4 v = rh[j];
      if (has_pro) v = pro_apply(v, p_sc, p_sh, lim[j]);
      unsigned char* q = sm + hrow * ROWB + lcol4 * 8;
      if constexpr (NP == 3) {
        uint2 p1, p2, p3;
        split3_quad(v, p1, p2, p3);
        *reinterpret_cast<uint2*>(q) = p1;
        *reinterpret_cast<uint2*>(q + 32) = p2;
        *reinterpret_cast<uint2*>(q + 64) = p3;
      } else {
        uint2 p1, p2;
        split2h_quad(v, x2h_in, p1, p2);
        *reinterpret_cast<uint2*>(q) = p1;
        *reinterpret_cast<uint2*>(q + 32) = p2;
      }
    }
  };
  bf16x8 fb[2][TN][NP];
  int lw_kc = 0, lw_t = 0;            // the next weight chunk to load
  auto load_w = [&](auto S) {         // chunk parity S: odd chunks come from the sign-flipped copy
    constexpr int s_ = decltype(S)::value;
    const int c = (DCS_X3W_EXP & 32) ? 0 : s_wc[lw_t] + lw_kc;
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const unsigned off = (unsigned)(((c * J + jt0 + b) * NP + p) * 1024) + lane16 + ((s_ && !(DCS_X3W_EXP & 16)) ? neg_off : 0u);
        fb[s_][b][p] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsB, off, 0, 0));
      }
    if (lw_kc * 9 + lw_t + 1 < nch) { lw_t += 1; if (lw_t == 9) { lw_t = 0; lw_kc += 1; } }
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;

  f32x16 acc[2][TM][TN];
#pragma unroll
  for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[s_][a][b][r] = 0.f;

  load_halo(0);
  load_w(P0{});                          // chunk 0
  store_halo(0);
  if (kch > 1) load_halo(1);
  __syncthreads();

  constexpr int NTERM = NP == 3 ? 6 : 3;
  constexpr int PA[6] = {NP == 3 ? 0 : 1, NP == 3 ? 2 : 0, NP == 3 ? 1 : 0, 0, 1, 0};       // smallest products first
  constexpr int PB[6] = {NP == 3 ? 2 : 0, NP == 3 ? 0 : 1, NP == 3 ? 1 : 0, 1, 0, 0};
  bf16x8 fa[TM][NP];
  const unsigned char* Abase = sm + ((wm * 2 + 1) * HWD + l31 + 1) * ROWB + h * 16;
  int kc = 0, t = 0;
  auto step = [&](auto PAR) {
    constexpr int par = decltype(PAR)::value;
    const unsigned char* Ab = Abase + s_ho[t];
#if DCS_X3W_EXP & 4
    if (kc == 0 && t == 0)
#endif
#pragma unroll
    for (int o = 0; o < NP; ++o) {
      const int pa = NP == 3 ? (o == 0 ? 0 : (o == 1 ? 2 : 1)) : (1 - o);
#pragma unroll
      for (int a = 0; a < TM; ++a) fa[a][pa] = *reinterpret_cast<const bf16x8*>(Ab + a * HWD * ROWB + pa * 32);
    }
#if DCS_X3W_EXP & 2
    if (kc == 0 && t < 2)
#endif
    load_w(std::integral_constant<int, 1 - par>{});                 // the next chunk's weight fragments
#pragma unroll
    for (int term = 0; term < NTERM; ++term)
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          if constexpr (NP == 3)
            acc[par][a][b] =
                __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][PA[term]], fb[par][b][PB[term]], acc[par][a][b], 0, 0, 0);
          else
            acc[(DCS_X3W_EXP & 16) ? 0 : par][a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                __builtin_bit_cast(f16x8, fa[a][PA[term]]), __builtin_bit_cast(f16x8, fb[par][b][PB[term]]),
                acc[(DCS_X3W_EXP & 16) ? 0 : par][a][b], 0, 0, 0);
        }
    t += 1;
    if (t == 9) {
      t = 0; kc += 1;
#if !(DCS_X3W_EXP & 1)
      if (kc < kch) {
        __syncthreads();
        store_halo(kc);
        if (kc + 1 < kch) load_halo(kc + 1);
        __syncthreads();
      }
#endif
    }
  };
  for (int i = 0; i < nch; i += 2) {
    step(P0{});
    if (i + 1 < nch) step(P1{});
  }
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[0][a][b][r] -= acc[1][a][b][r];
        if constexpr (NP == 2) acc[0][a][b][r] *= x2h_out;             // a power of two: exact
      }
  __syncthreads();                         // the halo is dead: the epilogue reuses its LDS
#if DCS_X3W_EXP & 8
  {
    float sacc = 0.f;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc += acc[0][a][b][r];
    if (sacc != 12345.678f) return;
  }
#endif

  conv_epilogue<BM, BN, TM, TN, WM, SMEM_FLOATS>(acc[0], smem, rowoff, bias, dst, g.dst_cstride, g.Cout, co0, accumulate, stats,
                                                 mtile, (unsigned long long)g.N * g.TY * g.TX, wm, wn, bnb, true);
}

// ------------------------------------------------------------------------------------------------
// conv3x3_x3w_kernel<.., 2> with the halo DOUBLE-BUFFERED in LDS (round 3; fp16 two-piece form only).  Timing experiments put
// 5-10 % of that kernel into the halo replacement: every nine taps all waves stop at a barrier, split and write the next 16
// channels' halo, stop at a second barrier.  Here a chunk's nine taps are unrolled (two variants for the chunk's starting
// parity), the halo of chunk kc + 1 is written into the OTHER buffer one staging slot per tap behind the MFMAs of chunk kc,
// the loads of chunk kc + 2 go out as soon as the registers are free (unconditionally -- out of range past the end -- so
// that the compiler counts the loads in flight instead of draining them), and ONE barrier per chunk remains.  Same
// products in the same order: bitwise the single-buffered kernel.
// DCS_DB_VAR (compile time): where a tap's staging slot sits -- bit 0: in front of the tap's MFMAs, bit 1: no scheduling barrier
// between taps.  Measured (conv_bench, 64 / 128 / 256 / 512 channels, TF): 0: 258 / 367-375 / 400 / 403, 1: 260 / 367-376 / 405 / 408,
// 2: 258 / 361-367 / 397 / 399, 3: 267 / 368-377 / 402 / 405 (single-buffered kernel on that box: 253 / 352-355 / 375 / 376 + 3 %).
#ifndef DCS_DB_VAR
#define DCS_DB_VAR 3
#endif
template <int BN, int TH>
__device__ __forceinline__
void conv3x3_x3w_db_body(const float* __restrict__ src, const unsigned char* __restrict__ wfrag, const float* __restrict__ bias,
                         float* __restrict__ dst, const DcsConvGeom& g, const int accumulate, const int ntiles,
                         float* __restrict__ stats, const BnBwdEpi bnb, const float* __restrict__ pro, const int J,
                         const unsigned neg_off, const BlkId bi, const unsigned* __restrict__ src_max) {
  constexpr int NP = 2;
  constexpr int BM = 32 * TH, HWD = 34, HROWS = (TH + 2) * HWD;
  constexpr int WN = BM == 256 ? 1 : 2, WM = 4 / WN, TM = 2, TN = BN / (WN * 32);
  constexpr int ROWB = 80;
  constexpr int A_BYTES = HROWS * ROWB;
  constexpr int EPI_FLOATS = 4 * 32 * (TN * 32 + 4) + WM * BN * 2;
  constexpr int SMEM_FLOATS = (2 * A_BYTES / 4) > EPI_FLOATS ? (2 * A_BYTES / 4) : EPI_FLOATS;
  constexpr int NH = (HROWS * 4 + 255) / 256;
  static_assert(TM * WM * 32 == BM && (BN == 64 || BN == 128) && NH < 9, "unsupported tile");

  __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
  unsigned char* const sm = reinterpret_cast<unsigned char*>(smem);
  __shared__ long long rowoff[BM];
  __shared__ int s_ho[9], s_wc[9];
  __shared__ __attribute__((aligned(16))) float s_pro[2 * DCS_PRO_MAXK];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int wm = wid / WN, wn = wid % WN;
  const int lcol4 = tid & 3;
  const bool has_pro = pro != nullptr;
  if (has_pro)
    for (int e = tid; e < 2 * g.K; e += 256) s_pro[(e < g.K ? 0 : DCS_PRO_MAXK - g.K) + e] = pro[e];
  float x2h_in = (float)(1 << X2H_KX), x2h_out = 1.f / (float)(1 << (X2H_KX + X2H_KW));
  if (src_max != nullptr) {
    const unsigned mbits = __builtin_amdgcn_readfirstlane(*src_max);
    const int e = (int)((mbits >> 23) & 0xffu) - 126;               // max = f 2^e, f in [0.5, 1)
    int k = (mbits >> 23) == 0u ? 0 : 14 - e;
    k = k < -100 ? -100 : (k > 100 ? 100 : k);
    x2h_in = __uint_as_float((unsigned)(127 + k) << 23);
    x2h_out = __uint_as_float((unsigned)(127 - k - X2H_KW) << 23);
  }

  const int bid = dcs_xcd_remap(bi.x, bi.nx);
  const int ntile = bid % ntiles, mtile = bid / ntiles;
  const int co0 = ntile * BN;
  const int tpx = g.TX >> 5, tpy = g.TY / TH;
  const int n = mtile / (tpx * tpy);
  const int trem = mtile - n * (tpx * tpy);
  const int y0 = (trem / tpx) * TH, x0 = (trem % tpx) << 5;

  if (tid < BM) {
    const int ry = tid >> 5, px = tid & 31;
    rowoff[tid] = (((long long)n * g.DH + (y0 + ry)) * g.DW + (x0 + px)) * g.dst_cstride;
  }
  if (tid < 9) {
    s_ho[tid] = (g.offy[tid] * HWD + g.offx[tid]) * ROWB;
    s_wc[tid] = g.wofs[tid] >> 4;                                 // first K chunk of the tap
  }
  const long long img_elems = (long long)g.SH * g.SW * g.src_cstride;
  const int wchunks = g.wstride >> 4;
  const __amdgpu_buffer_rsrc_t rsA = make_rsrc(src + (long long)n * img_elems, img_elems * 4);
  const __amdgpu_buffer_rsrc_t rsB = make_rsrc(reinterpret_cast<const float*>(wfrag), 2ll * wchunks * J * NP * 1024);

  int h_off[NH];
  float lim[NH];
#pragma unroll
  for (int j = 0; j < NH; ++j) {
    const int hrow = (tid + 256 * j) >> 2;
    const int hy = hrow / HWD, hx = hrow - hy * HWD;
    const int iy = y0 + hy - 1, ix = x0 + hx - 1;
    const bool ok = hrow < HROWS && (unsigned)iy < (unsigned)g.SH && (unsigned)ix < (unsigned)g.SW;
    h_off[j] = ok ? (iy * g.SW + ix) * g.src_cstride + lcol4 * 4 : -1;
    lim[j] = ok ? __builtin_inff() : 0.f;
  }
  __syncthreads();

  const int kch = g.K >> 4;
  const int nch = 9 * kch;
  const int jt0 = (co0 >> 5) + wn * TN;                            // first 32-row weight tile of this wave
  const unsigned lane16 = (unsigned)lane * 16u;

  float4 rh[NH];
  auto load_halo = [&](int kc) {                                   // unconditional: past the last chunk every lane is out of range
#pragma unroll
    for (int j = 0; j < NH; ++j) rh[j] = bld4(rsA, (h_off[j] >= 0 && kc < kch) ? (unsigned)(h_off[j] + kc * 16) * 4u : OOB);
  };
  auto store_slot = [&](int j, int kc, unsigned char* buf) {
    const int hrow = (tid + 256 * j) >> 2;
    if (NH * 64 != HROWS && hrow >= HROWS) return;
    float4 v = rh[j];
    if (has_pro) {
      const float4 p_sc = ld4(&s_pro[kc * 16 + lcol4 * 4]), p_sh = ld4(&s_pro[DCS_PRO_MAXK + kc * 16 + lcol4 * 4]);
      v = pro_apply(v, p_sc, p_sh, lim[j]);
    }
    unsigned char* q = buf + hrow * ROWB + lcol4 * 8;
    uint2 p1, p2;
    split2h_quad(v, x2h_in, p1, p2);
    *reinterpret_cast<uint2*>(q) = p1;
    *reinterpret_cast<uint2*>(q + 32) = p2;
  };
  bf16x8 fb[2][TN][NP];
  int lw_kc = 0, lw_t = 0;            // the next weight chunk to load
  auto load_w = [&](auto S) {         // chunk parity S: odd chunks come from the sign-flipped copy
    constexpr int s_ = decltype(S)::value;
    const int c = s_wc[lw_t] + lw_kc;
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const unsigned off = (unsigned)(((c * J + jt0 + b) * NP + p) * 1024) + lane16 + (s_ ? neg_off : 0u);
        fb[s_][b][p] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsB, off, 0, 0));
      }
    if (lw_kc * 9 + lw_t + 1 < nch) { lw_t += 1; if (lw_t == 9) { lw_t = 0; lw_kc += 1; } }
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;

  f32x16 acc[2][TM][TN];
#pragma unroll
  for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[s_][a][b][r] = 0.f;

  load_halo(0);
  load_w(P0{});                          // chunk 0
#pragma unroll
  for (int j = 0; j < NH; ++j) store_slot(j, 0, sm);
  load_halo(1);
  __syncthreads();

  constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};               // smallest products first
  bf16x8 fa[TM][NP];
  const int a_off = ((wm * 2 + 1) * HWD + l31 + 1) * ROWB + h * 16;
  // one 16-channel chunk = nine taps, unrolled; START = parity of its first tap (9 is odd: it alternates per chunk)
  auto chunk = [&](auto START, const int kc) {
    constexpr int start = decltype(START)::value;
    const unsigned char* Abase = sm + (kc & 1) * A_BYTES + a_off;
    unsigned char* const other = sm + ((kc + 1) & 1) * A_BYTES;
    const bool more = kc + 1 < kch;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      constexpr int dummy = 0; (void)dummy;
      const int par = (start + t) & 1;
      const unsigned char* Ab = Abase + s_ho[t];
#pragma unroll
      for (int p = NP - 1; p >= 0; --p)
#pragma unroll
        for (int a = 0; a < TM; ++a) fa[a][p] = *reinterpret_cast<const bf16x8*>(Ab + a * HWD * ROWB + p * 32);
      if (par == 0) load_w(P1{}); else load_w(P0{});                // the next tap's weight fragments
#if DCS_DB_VAR & 1
      if (t < NH) { if (more) store_slot(t, kc + 1, other); }
      if (t == NH) load_halo(kc + 2);
#endif
#pragma unroll
      for (int term = 0; term < 3; ++term)
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b)
            acc[par][a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[a][PA[term]]),
                                                                    __builtin_bit_cast(f16x8, fb[par][b][PB[term]]),
                                                                    acc[par][a][b], 0, 0, 0);
#if !(DCS_DB_VAR & 1)
      if (t < NH) { if (more) store_slot(t, kc + 1, other); }       // halo of the next chunk: one staging slot per tap
      if (t == NH) load_halo(kc + 2);                               // registers free again: the chunk after that
#endif
#if !(DCS_DB_VAR & 2)
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
    __syncthreads();
  };
  for (int kc = 0; kc < kch; kc += 2) {
    chunk(P0{}, kc);
    if (kc + 1 < kch) chunk(P1{}, kc + 1);
  }
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][a][b][r] = (acc[0][a][b][r] - acc[1][a][b][r]) * x2h_out;
  // (the last chunk ended with a barrier: the halo is dead, the epilogue reuses its LDS)

  conv_epilogue<BM, BN, TM, TN, WM, SMEM_FLOATS>(acc[0], smem, rowoff, bias, dst, g.dst_cstride, g.Cout, co0, accumulate, stats,
                                                 mtile, (unsigned long long)g.N * g.TY * g.TX, wm, wn, bnb, true);
}

// ------------------------------------------------------------------------------------------------
// The 7x7 / stride 2 / pad 3 stem forward (14-tap form, ops.geom_stem_fwd) with the INPUT PATCH resident in LDS -- the
// counterpart of conv3x3_x3w_kernel for the one layer whose K is a filter row instead of a channel range (round 3; fp16
// two-piece form only).  A block computes 8 rows x 32 pixels x 64 channels.  The (2*8+5) x 70-pixel patch of the NHWC4
// image is loaded ONCE (the per-tap kernel re-read every input pixel ~12x through L1, one LDS staging pass and one barrier
// per tap), split into two fp16 pieces of x * 2^X2H_KX and kept as two planes [row][pixel][4 channels]: the 16 k-values of
// tap (r, c) for output pixel (oy, ox) are the 4 pixels 2 ox - 3 + 4 c .. + 3 of input row 2 oy - 3 + r, i.e. 32
// contiguous bytes of a plane whose address advances by 16 B per output pixel -- a lane's MFMA A fragment (k = 8 h .. 8 h + 7)
// is one aligned ds_read_b128, the 64 lanes of a wave read one contiguous run (conflict-free).  Pixel slot 8 of a filter row
// carries zero weights (ops.pack_stem_weight); its data is real image data or hardware zero fill, never LDS garbage.
// Weight fragments straight from global memory in the fragment-major split image of the packed [64][7][8][4] weight
// (dcs_split_weight_frag in its fp16 form with wstride = 224: chunk = tap), two accumulator sets over the tap parity with the
// sign-flipped copy (rounding-bias cancellation), no barrier inside the loop.
__device__ __forceinline__
void stem7_h2_body(const float* __restrict__ src, const unsigned char* __restrict__ wfrag, float* __restrict__ dst,
                   const DcsConvGeom& g, const int accumulate, float* __restrict__ stats, const unsigned neg_off,
                   const BlkId bi) {
  constexpr int TH = 8, BM = 256, BN = 64, WM = 4, TM = 2, TN = 2, NP = 2, J = 2, NTAP = 14;
  constexpr int PR = 2 * TH + 5, PW = 70;                  // patch rows, pixels per row
  constexpr int PROWB = PW * 8, PLANE = PR * PROWB;        // bytes: 4 fp16 per pixel and piece
  constexpr int A_BYTES = NP * PLANE;
  constexpr int EPI_FLOATS = 4 * 32 * (TN * 32 + 4) + WM * BN * 2;
  constexpr int SMEM_FLOATS = (A_BYTES / 4) > EPI_FLOATS ? (A_BYTES / 4) : EPI_FLOATS;
  constexpr int NH = (PR * PW + 255) / 256;
  static_assert(PROWB % 16 == 0 && PLANE % 16 == 0, "fragment reads must stay 16-byte aligned");

  __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
  unsigned char* const sm = reinterpret_cast<unsigned char*>(smem);
  __shared__ long long rowoff[BM];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wm = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const float x2h_in = (float)(1 << X2H_KX), x2h_out = 1.f / (float)(1 << (X2H_KX + X2H_KW));

  const int mtile = dcs_xcd_remap(bi.x, bi.nx);
  const int tpx = g.TX >> 5, tpy = g.TY / TH;
  const int n = mtile / (tpx * tpy);
  const int trem = mtile - n * (tpx * tpy);
  const int y0 = (trem / tpx) * TH, x0 = (trem % tpx) << 5;
  {
    const int ry = tid >> 5, px = tid & 31;
    rowoff[tid] = (((long long)n * g.DH + (y0 + ry)) * g.DW + (x0 + px)) * g.dst_cstride;
  }
  const long long img_elems = (long long)g.SH * g.SW * 4;
  const __amdgpu_buffer_rsrc_t rsA = make_rsrc(src + (long long)n * img_elems, img_elems * 4);
  const __amdgpu_buffer_rsrc_t rsB = make_rsrc(reinterpret_cast<const float*>(wfrag), 2ll * NTAP * J * NP * 1024);

  // the patch: global -> registers -> two fp16 pieces -> LDS, once
  float4 rh[NH];
#pragma unroll
  for (int j = 0; j < NH; ++j) {
    const int e = tid + 256 * j;
    const int pr = e / PW, pc = e - pr * PW;
    const int iy = 2 * y0 - 3 + pr, ix = 2 * x0 - 3 + pc;
    const bool ok = e < PR * PW && (unsigned)iy < (unsigned)g.SH && (unsigned)ix < (unsigned)g.SW;
    rh[j] = bld4(rsA, ok ? (unsigned)((iy * g.SW + ix) * 4) * 4u : OOB);
  }
  const unsigned lane16 = (unsigned)lane * 16u;
  bf16x8 fb[2][TN][NP];
  auto load_w = [&](auto S, int t) {    // tap t = weight chunk t; odd taps come from the sign-flipped copy
    constexpr int s_ = decltype(S)::value;
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const unsigned off = (unsigned)(((t * J + b) * NP + p) * 1024) + lane16 + (s_ ? neg_off : 0u);
        fb[s_][b][p] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsB, off, 0, 0));
      }
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  load_w(P0{}, 0);
#pragma unroll
  for (int j = 0; j < NH; ++j) {
    const int e = tid + 256 * j;
    if (NH * 256 != PR * PW && e >= PR * PW) continue;
    uint2 p1, p2;
    split2h_quad(rh[j], x2h_in, p1, p2);
    *reinterpret_cast<uint2*>(sm + e * 8) = p1;
    *reinterpret_cast<uint2*>(sm + PLANE + e * 8) = p2;
  }

  f32x16 acc[2][TM][TN];
#pragma unroll
  for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[s_][a][b][r] = 0.f;
  __syncthreads();

  constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};               // smallest products first
  // wave wm: output rows 2 wm, 2 wm + 1 of the tile; lane (l31, h): output pixel l31, k half h = patch pixels +2 h, +2 h + 1
  const unsigned char* Abase = sm + ((4 * wm) * PW + 2 * l31 + 2 * h) * 8;
  bf16x8 fa[TM][NP];
  auto step = [&](auto PAR, int t) {
    constexpr int par = decltype(PAR)::value;
    const unsigned char* Ab = Abase + ((t >> 1) * PW + 4 * (t & 1)) * 8;
#pragma unroll
    for (int p = NP - 1; p >= 0; --p)
#pragma unroll
      for (int a = 0; a < TM; ++a) fa[a][p] = *reinterpret_cast<const bf16x8*>(Ab + a * 2 * PROWB + p * PLANE);
    if (t + 1 < NTAP) load_w(std::integral_constant<int, 1 - par>{}, t + 1);
#pragma unroll
    for (int term = 0; term < 3; ++term)
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
          acc[par][a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[a][PA[term]]),
                                                                  __builtin_bit_cast(f16x8, fb[par][b][PB[term]]),
                                                                  acc[par][a][b], 0, 0, 0);
  };
#pragma unroll
  for (int t = 0; t < NTAP; t += 2) {
    step(P0{}, t);
    step(P1{}, t + 1);
  }
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][a][b][r] = (acc[0][a][b][r] - acc[1][a][b][r]) * x2h_out;
  __syncthreads();                         // the patch is dead: the epilogue reuses its LDS

  conv_epilogue<BM, BN, TM, TN, WM, SMEM_FLOATS>(acc[0], smem, rowoff, nullptr, dst, g.dst_cstride, g.Cout, 0, accumulate, stats,
                                                 mtile, (unsigned long long)g.N * g.TY * g.TX, wm, 0, BnBwdEpi{nullptr, nullptr, nullptr, 0},
                                                 true);
}

// geometries stem7_h2_kernel covers: the 14-tap stem form of ops.geom_stem_fwd on 8 x 32-pixel output tiles
bool stem7_eligible(const DcsConvGeom* g) {
  if (!g->stem || g->ntaps != 14 || g->sy != 2 || g->sx != 2 || g->dsy != 1 || g->dsx != 1 || g->dy0 || g->dx0) return false;
  if (g->Cout != 64 || g->wstride != 224 || g->src_cstride != 4 || (g->TX & 31) || (g->TY & 7)) return false;
  if (g->DH != g->TY || g->DW != g->TX || g->SH < 2 * g->TY - 1 || g->SW < 2 * g->TX - 1) return false;
  for (int t = 0; t < 14; ++t)
    if (g->offy[t] != t / 2 - 3 || g->offx[t] != -3 + 4 * (t & 1) || g->wofs[t] != (t / 2) * 32 + 16 * (t & 1)) return false;
  return (long long)g->SH * g->SW * 16 < 0x7FFFFFFFll;
}

// w [rows][wstride] fp32 -> fragment-major split image (see conv3x3_x3w_kernel): unit (((c*J + j)*3 + p)*2 + h)*32 + r
// (16 bytes) = piece p of row 32j + r, channels 16c + 8h .. +7; rows beyond `rows` are zero; a sign-flipped copy follows.
__global__ void split_weight_frag_kernel(const float* __restrict__ w, u32x4* __restrict__ out, const int rows,
                                         const int wstride, const int J, const long long units) {
  const long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x;       // (c, j, h, r): one thread writes 3 pieces
  const long long nthreads = (long long)(wstride >> 4) * J * 64;
  if (u >= nthreads) return;
  const int r = (int)(u & 31), hh = (int)((u >> 5) & 1);
  const long long cj = u >> 6;
  const int j = (int)(cj % J);
  const long long c = cj / J;
  const int row = 32 * j + r;
  u32x4 q1 = {0, 0, 0, 0}, q2 = q1, q3 = q1;
  if (row < rows) {
    const float* src = w + (long long)row * wstride + c * 16 + hh * 8;
    const float4 v0 = ld4(src), v1 = ld4(src + 4);
    unsigned a1, a2, a3;
    split3_pair(v0.x, v0.y, a1, a2, a3); q1.x = a1; q2.x = a2; q3.x = a3;
    split3_pair(v0.z, v0.w, a1, a2, a3); q1.y = a1; q2.y = a2; q3.y = a3;
    split3_pair(v1.x, v1.y, a1, a2, a3); q1.z = a1; q2.z = a2; q3.z = a3;
    split3_pair(v1.z, v1.w, a1, a2, a3); q1.w = a1; q2.w = a2; q3.w = a3;
  }
  const long long base = ((cj * 3) * 2 + hh) * 32 + r;
  const u32x4 f = {0x80008000u, 0x80008000u, 0x80008000u, 0x80008000u};
  out[base] = q1; out[base + 64] = q2; out[base + 128] = q3;
  out[units + base] = q1 ^ f; out[units + base + 64] = q2 ^ f; out[units + base + 128] = q3 ^ f;
}

// The fp16 two-piece form of the same image (conv3x3_x3w_body<.., NP = 2>): unit (((c*J + j)*2 + p)*2 + h)*32 + r, pieces of
// w * 2^X2H_KW; the sign-flipped copy follows (an fp16 sign bit sits where a bf16 one does).
__global__ void split_weight_frag_h2_kernel(const float* __restrict__ w, u32x4* __restrict__ out, const int rows,
                                            const int wstride, const int J, const long long units) {
  const long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x;       // (c, j, h, r): one thread writes 2 pieces
  const long long nthreads = (long long)(wstride >> 4) * J * 64;
  if (u >= nthreads) return;
  const int r = (int)(u & 31), hh = (int)((u >> 5) & 1);
  const long long cj = u >> 6;
  const int j = (int)(cj % J);
  const long long c = cj / J;
  const int row = 32 * j + r;
  u32x4 q1 = {0, 0, 0, 0}, q2 = q1;
  if (row < rows) {
    const float* src = w + (long long)row * wstride + c * 16 + hh * 8;
    const float4 v0 = ld4(src), v1 = ld4(src + 4);
    const float sc = (float)(1 << X2H_KW);
    uint2 a1, a2, b1, b2;
    split2h_quad(v0, sc, a1, a2);
    split2h_quad(v1, sc, b1, b2);
    q1.x = a1.x; q1.y = a1.y; q1.z = b1.x; q1.w = b1.y;
    q2.x = a2.x; q2.y = a2.y; q2.z = b2.x; q2.w = b2.y;
  }
  const long long base = ((cj * 2) * 2 + hh) * 32 + r;
  const u32x4 f = {0x80008000u, 0x80008000u, 0x80008000u, 0x80008000u};
  out[base] = q1; out[base + 64] = q2;
  out[units + base] = q1 ^ f; out[units + base + 64] = q2 ^ f;
}

// ------------------------------------------------------------------------------------------------------------------
// Kernel entry points.  Every body above runs either as its own launch or as one SUB-LAUNCH of a multi launch: the three
// pyramid levels of a layer share every weight (network/backbone/resnet_pyramid.py:318-341) and run the same kernel on
// maps of 1, 1/4 and 1/16 of the pixels, so one grid covers the blocks of all of them -- sub-launch i owns the hardware
// blocks [blk0_i, blk0_i + nblk_i), blk0_i a multiple of 8 (the XCD round-robin then sees the level-relative id), and a
// block runs exactly the code of the single launch with its level's arguments: results are bitwise those of the
// per-level launches.  Arguments live in the kernel-argument segment; the level index is wave-uniform (scalar loads).
constexpr int MULTI_MAX = DCS_MULTI_MAX;

struct GatherSub {
  const float* src; const unsigned char* w; const float* bias; float* dst; float* stats; const float* pro;
  BnBwdEpi bnb;
  long long slab_stride;
  int accumulate, ntiles, cps, J;
  unsigned neg_off;
  int blk0, nbx, nblk;
  const unsigned* src_max;             // fp16 two-piece kernels: device word with the bits of max |src| (nullable)
};
struct GatherMulti { DcsConvGeom g[MULTI_MAX]; GatherSub s[MULTI_MAX]; };

struct WgradSub {
  const float* src; const float* dy; float* slab; const float* pro;
  long long mps;
  int dy_cstride, split0, cps, ciT;
  int blk0, nbx, nblk;
  const unsigned* dy_max;              // fp16 two-piece rolling kernel: device word with the bits of max |dy| (else null)
};
struct WgradMulti { DcsConvGeom g[MULTI_MAX]; WgradSub s[MULTI_MAX]; };

template <class MP>
__device__ __forceinline__ int multi_level(const MP& P) {
  int lv = 0;
#pragma unroll
  for (int i = 1; i < MULTI_MAX; ++i) lv += (int)blockIdx.x >= P.s[i].blk0 ? 1 : 0;     // unused entries: blk0 = INT_MAX
  return lv;
}

#define DCS_BLK BlkId{(int)blockIdx.x, (int)gridDim.x, (int)blockIdx.y}
#define DCS_BLK2 BlkId{(int)blockIdx.x, (int)gridDim.x, (int)blockIdx.y, (int)gridDim.y}

template <int BN, int BM = 128, bool STEM = false, int NP = 3, bool WF = false>
__global__ __launch_bounds__(256, 2)
void conv_gather_x3_kernel(const float* __restrict__ src, const unsigned char* __restrict__ wsp,
                           const float* __restrict__ bias, float* __restrict__ dst, const DcsConvGeom g,
                           const int accumulate, const int ntiles, float* __restrict__ stats, const int cps,
                           const long long slab_stride, const BnBwdEpi bnb, const float* __restrict__ pro,
                           const unsigned* __restrict__ src_max) {
  conv_gather_x3_body<BN, BM, STEM, NP, WF>(src, wsp, bias, dst, g, accumulate, ntiles, stats, cps, slab_stride, bnb, pro, DCS_BLK,
                                            src_max);
}
template <int BN, int BM = 128, bool STEM = false, int NP = 3, bool WF = false>
__global__ __launch_bounds__(256, 2)
void conv_gather_x3_multi_kernel(const GatherMulti P) {
  const int lv = multi_level(P);
  const GatherSub& s = P.s[lv];
  const int rel = (int)blockIdx.x - s.blk0;
  if (rel >= s.nblk) return;
  conv_gather_x3_body<BN, BM, STEM, NP, WF>(s.src, s.w, s.bias, s.dst, P.g[lv], s.accumulate, s.ntiles, s.stats, s.cps,
                                            s.slab_stride, s.bnb, s.pro, BlkId{rel % s.nbx, s.nbx, rel / s.nbx}, s.src_max);
}

template <int BN, int TH, int NP = 3>
__global__ __launch_bounds__(256, 2)
void conv3x3_x3w_kernel(const float* __restrict__ src, const unsigned char* __restrict__ wfrag, const float* __restrict__ bias,
                        float* __restrict__ dst, const DcsConvGeom g, const int accumulate, const int ntiles,
                        float* __restrict__ stats, const BnBwdEpi bnb, const float* __restrict__ pro, const int J,
                        const unsigned neg_off, const unsigned* __restrict__ src_max) {
  conv3x3_x3w_body<BN, TH, NP>(src, wfrag, bias, dst, g, accumulate, ntiles, stats, bnb, pro, J, neg_off, DCS_BLK, src_max);
}
template <int BN, int TH, int NP = 3>
__global__ __launch_bounds__(256, 2)
void conv3x3_x3w_multi_kernel(const GatherMulti P) {
  const int lv = multi_level(P);
  const GatherSub& s = P.s[lv];
  const int rel = (int)blockIdx.x - s.blk0;
  if (rel >= s.nblk) return;
  conv3x3_x3w_body<BN, TH, NP>(s.src, s.w, s.bias, s.dst, P.g[lv], s.accumulate, s.ntiles, s.stats, s.bnb, s.pro, s.J, s.neg_off,
                               BlkId{rel, s.nbx, 0}, s.src_max);
}

template <int BN, int TH>
__global__ __launch_bounds__(256, 2)
void conv3x3_x3w_db_kernel(const float* __restrict__ src, const unsigned char* __restrict__ wfrag, const float* __restrict__ bias,
                           float* __restrict__ dst, const DcsConvGeom g, const int accumulate, const int ntiles,
                           float* __restrict__ stats, const BnBwdEpi bnb, const float* __restrict__ pro, const int J,
                           const unsigned neg_off, const unsigned* __restrict__ src_max) {
  conv3x3_x3w_db_body<BN, TH>(src, wfrag, bias, dst, g, accumulate, ntiles, stats, bnb, pro, J, neg_off, DCS_BLK, src_max);
}
template <int BN, int TH>
__global__ __launch_bounds__(256, 2)
void conv3x3_x3w_db_multi_kernel(const GatherMulti P) {
  const int lv = multi_level(P);
  const GatherSub& s = P.s[lv];
  const int rel = (int)blockIdx.x - s.blk0;
  if (rel >= s.nblk) return;
  conv3x3_x3w_db_body<BN, TH>(s.src, s.w, s.bias, s.dst, P.g[lv], s.accumulate, s.ntiles, s.stats, s.bnb, s.pro, s.J, s.neg_off,
                              BlkId{rel, s.nbx, 0}, s.src_max);
}

__global__ __launch_bounds__(256, 2)
void stem7_h2_kernel(const float* __restrict__ src, const unsigned char* __restrict__ wfrag, float* __restrict__ dst,
                     const DcsConvGeom g, const int accumulate, float* __restrict__ stats, const unsigned neg_off) {
  stem7_h2_body(src, wfrag, dst, g, accumulate, stats, neg_off, DCS_BLK);
}
__global__ __launch_bounds__(256, 2)
void stem7_h2_multi_kernel(const GatherMulti P) {
  const int lv = multi_level(P);
  const GatherSub& s = P.s[lv];
  const int rel = (int)blockIdx.x - s.blk0;
  if (rel >= s.nblk) return;
  stem7_h2_body(s.src, s.w, s.dst, P.g[lv], s.accumulate, s.stats, s.neg_off, BlkId{rel, s.nbx, 0});
}

template <int BT, int NP = 3>
__global__ __launch_bounds__(256, 2)
void conv_wgrad_x3_kernel(const float* __restrict__ src, const float* __restrict__ dy, float* __restrict__ slab,
                          const DcsConvGeom g, const int dy_cstride, const int split0, const long long mps,
                          const int ciT, const float* __restrict__ pro, const unsigned* __restrict__ dy_max) {
  conv_wgrad_x3_body<BT, NP>(src, dy, slab, g, dy_cstride, split0, mps, ciT, pro, DCS_BLK2, dy_max);
}
template <int BT, int NP = 3>
__global__ __launch_bounds__(256, 2)
void conv_wgrad_x3_multi_kernel(const WgradMulti P) {
  const int lv = multi_level(P);
  const WgradSub& s = P.s[lv];
  const int rel = (int)blockIdx.x - s.blk0;
  if (rel >= s.nblk) return;
  conv_wgrad_x3_body<BT, NP>(s.src, s.dy, s.slab, P.g[lv], s.dy_cstride, s.split0, s.mps, s.ciT, s.pro,
                             BlkId{rel % s.nbx, s.nbx, rel / s.nbx, s.nblk / s.nbx}, s.dy_max);
}

template <int NP = 3>
__global__ __launch_bounds__(256, 2)
void conv_wgrad3x3_x3r_kernel(const float* __restrict__ src, const float* __restrict__ dy, float* __restrict__ slab,
                              const DcsConvGeom g, const int dy_cstride, const int split0, const int cps, const int ciT,
                              const float* __restrict__ pro, const unsigned* __restrict__ dy_max) {
  conv_wgrad3x3_x3r_body<NP>(src, dy, slab, g, dy_cstride, split0, cps, ciT, pro, DCS_BLK2, dy_max);
}
template <int NP = 3>
__global__ __launch_bounds__(256, 2)
void conv_wgrad3x3_x3r_multi_kernel(const WgradMulti P) {
  const int lv = multi_level(P);
  const WgradSub& s = P.s[lv];
  const int rel = (int)blockIdx.x - s.blk0;
  if (rel >= s.nblk) return;
  conv_wgrad3x3_x3r_body<NP>(s.src, s.dy, s.slab, P.g[lv], s.dy_cstride, s.split0, s.cps, s.ciT, s.pro,
                             BlkId{rel % s.nbx, s.nbx, rel / s.nbx, s.nblk / s.nbx}, s.dy_max);
}

template <int NP = 3>
__global__ __launch_bounds__(256, 2)
void stem_wgrad_x3_kernel(const float* __restrict__ src, const float* __restrict__ dy, float* __restrict__ slab,
                          const DcsConvGeom g, const int dy_cstride, const int split0, const int cps,
                          const unsigned* __restrict__ dy_max) {
  stem_wgrad_x3_body<NP>(src, dy, slab, g, dy_cstride, split0, cps, DCS_BLK, dy_max);
}
template <int NP = 3>
__global__ __launch_bounds__(256, 2)
void stem_wgrad_x3_multi_kernel(const WgradMulti P) {
  const int lv = multi_level(P);
  const WgradSub& s = P.s[lv];
  const int rel = (int)blockIdx.x - s.blk0;
  if (rel >= s.nblk) return;
  stem_wgrad_x3_body<NP>(s.src, s.dy, s.slab, P.g[lv], s.dy_cstride, s.split0, s.cps, BlkId{rel, s.nbx, 0}, s.dy_max);
}

// geometries conv3x3_x3_kernel covers: the nine taps of a dense 3x3 / stride 1 / pad 1 window in any order
bool conv3x3_halo_eligible(const DcsConvGeom* g, int th) {
  if (g->stem || g->ntaps != 9 || g->sy != 1 || g->sx != 1 || g->dsy != 1 || g->dsx != 1 || g->dy0 || g->dx0) return false;
  if ((g->TX & 31) || g->TY % th || g->SH != g->TY || g->SW != g->TX || g->DH != g->TY || g->DW != g->TX) return false;
  unsigned seen = 0;
  for (int t = 0; t < 9; ++t) {
    if (g->offy[t] < -1 || g->offy[t] > 1 || g->offx[t] < -1 || g->offx[t] > 1) return false;
    seen |= 1u << ((g->offy[t] + 1) * 3 + g->offx[t] + 1);
  }
  return seen == 0x1FFu;
}

}  // namespace

extern "C" int dcs_split_weight(const float* w, void* out, int64_t rows, int wstride, void* stream) {
  DCS_CHECK_ARG(w && out && rows > 0 && wstride > 0 && (wstride & 15) == 0 && dcs_aligned16(w) && dcs_aligned16(out));
  const long long n4 = (long long)rows * wstride / 4;
  DCS_CHECK_ARG(n4 < (1ll << 31) * 256);
  hipLaunchKernelGGL(split_weight_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, dcs_stream(stream), w,
                     reinterpret_cast<uint2*>(out), n4, wstride);
  DCS_LAUNCH_RET();
}

extern "C" int dcs_split_weight_h2(const float* w, void* out, int64_t rows, int wstride, void* stream) {
  DCS_CHECK_ARG(w && out && rows > 0 && wstride > 0 && (wstride & 15) == 0 && dcs_aligned16(w) && dcs_aligned16(out));
  const long long n4 = (long long)rows * wstride / 4;
  DCS_CHECK_ARG(n4 < (1ll << 31) * 256);
  hipLaunchKernelGGL(split_weight_h2_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, dcs_stream(stream), w,
                     reinterpret_cast<uint2*>(out), n4, wstride);
  DCS_LAUNCH_RET();
}

namespace {

// ---- launch plans: validation + kernel choice of one (sub-)launch, shared by the single and the multi entries -----------
enum GatherKid { GK_X3W_64, GK_X3W_128, GK_X2H_64, GK_X2H_128, GK_HALO_64, GK_HALO_128, GK_STEM_256, GK_STEM_128, GK_128, GK_64_256, GK_64, GK_STEM7,
                 GK_H2 = 32 /* flag: the per-tap kernels in their fp16 two-piece form */,
                 GK_WF = 64 /* flag: ... with the weight fragments straight from global memory (DCS_ACC_WFRAG) */ };
struct GatherPlan { int kid; DcsConvGeom g; GatherSub s; unsigned nbx, nby; };

int plan_gather_x3(const DcsGatherLaunch& a, GatherPlan& P) {
  const DcsConvGeom* geom = a.geom;
  int rc = check_geom(geom);
  if (rc != DCS_OK) return rc;
  const int nsplit = a.nsplit;
  DCS_CHECK_ARG(a.src && a.wgt && a.dst && dcs_aligned16(a.src) && dcs_aligned16(a.wgt));
  DCS_CHECK_ARG(geom->dst_cstride >= geom->Cout && nsplit >= 1 && nsplit <= 64);
  // what this kernel family covers; everything else stays on dcs_conv_gather
  const bool stem14 = geom->stem && geom->ntaps == 14 && geom->Cout == 64 && geom->wstride == 224 && !a.pro && nsplit == 1;
  if ((geom->stem && !stem14) || geom->Cout <= 32 || (!geom->stem && (geom->K & 15)) || (geom->wstride & 15))
    return DCS_E_UNSUPPORTED;
  for (int t = 0; t < geom->ntaps; ++t)
    if (geom->wofs[t] & 15) return DCS_E_UNSUPPORTED;
  const long long M = (long long)geom->N * geom->TY * geom->TX;
  DCS_CHECK_ARG(M < 0x7FFFFF00ll);
  {
    const long long tyx = (long long)geom->TY * geom->TX;
    const long long img_bytes = (long long)geom->SH * geom->SW * geom->src_cstride * 4;
    long long span = 255 / tyx + 2;                       // images a 256-pixel block can touch
    if (span > geom->N) span = geom->N;
    if (span * img_bytes > 0x7FFFFFFFll || (long long)geom->Cout * geom->wstride * 6 > 0x7FFFFFFFll)
      return DCS_E_UNSUPPORTED;
  }
  const BnBwdEpi bnb{a.bn_y, a.bn_mask, a.bn, a.relu};
  const bool accum = (a.accumulate & 1) != 0, h2 = (a.accumulate & DCS_ACC_FP16X2) != 0, wf = (a.accumulate & DCS_ACC_WFRAG) != 0;
  DCS_CHECK_ARG((a.accumulate & ~(1 | DCS_ACC_FP16X2 | DCS_ACC_WFRAG)) == 0 && (!wf || (h2 && !geom->stem)));
  DCS_CHECK_ARG(!(a.stats && accum && !bnb.y));
  DCS_CHECK_ARG(!a.pro || (geom->K <= DCS_PRO_MAXK && dcs_aligned16(a.pro)));
  DCS_CHECK_ARG(!bnb.y || (a.stats && bnb.bn && nsplit == 1 && (geom->Cout & 3) == 0 && geom->dst_cstride == geom->Cout &&
                           dcs_aligned16(a.dst) && dcs_aligned16(bnb.y) && (!bnb.mask || dcs_aligned16(bnb.mask))));
  if (nsplit > 1)
    DCS_CHECK_ARG(!a.bias && !a.stats && !accum && a.slab_stride >= M * geom->dst_cstride);
  const int bn_ = geom->Cout > 64 ? 128 : 64;
  const int ntiles = (geom->Cout + bn_ - 1) / bn_;
  // 64-wide layers of large maps: 256-pixel tiles (enough of them to fill the chip several times over)
  const bool bm256 = bn_ == 64 && nsplit == 1 && M >= 256ll * 2048 && dcs_config().x3_bm128 == 0;
  const long long blocks = (bm256 ? (M + 255) / 256 : (M + 127) / 128) * ntiles;
  DCS_CHECK_ARG(blocks > 0 && blocks * nsplit < (1ll << 30));
  const int nch = geom->ntaps * (geom->stem ? 1 : geom->K >> 4);
  // an even number of chunks per K split: every split then pairs its +A and -A chunks (the bias cancellation of the kernel)
  int cps = (nch + nsplit - 1) / nsplit;
  if (nsplit > 1) cps += cps & 1;
  P.g = *geom;
  P.s = GatherSub{a.src, reinterpret_cast<const unsigned char*>(a.wgt), a.bias, a.dst, a.stats, a.pro, bnb,
                  (long long)a.slab_stride, (accum ? 1 : 0) | (dcs_streams(M * geom->dst_cstride * 4) ? 2 : 0), ntiles, cps,
                  0, 0u, 0, 0, 0, h2 ? a.src_max : nullptr};
  P.nbx = (unsigned)blocks;
  P.nby = (unsigned)nsplit;
  // DCS_X3_HALO=0: never; =2: whenever the geometry allows (tests: small shapes); default: when there are enough tiles
  const bool g_halo = dcs_config().x3_halo != 0 && !h2;              // (the LDS-weights halo kernel has no fp16 form)
  const bool g_halo_force = dcs_config().x3_halo == 2;
  if (g_halo && nsplit == 1 && (long long)geom->SH * geom->SW * geom->src_cstride * 4 <= 0x7FFFFFFFll) {
    // enough tiles to fill the chip twice over, else the per-tap kernel (and its K splits) does better
    if (bn_ == 64 && conv3x3_halo_eligible(geom, 8) && (g_halo_force || M >= 256ll * 1024)) {
      P.kid = GK_HALO_64; P.nbx = (unsigned)((M / 256) * ntiles);
      return DCS_OK;
    }
    if (bn_ == 128 && conv3x3_halo_eligible(geom, 4) && (g_halo_force || (M / 128) * ntiles >= 1024)) {
      P.kid = GK_HALO_128; P.nbx = (unsigned)((M / 128) * ntiles);
      return DCS_OK;
    }
  }
  if (geom->stem) P.kid = bm256 ? GK_STEM_256 : GK_STEM_128;
  else if (bn_ == 128) P.kid = GK_128;
  else P.kid = bm256 ? GK_64_256 : GK_64;
  if (P.kid != GK_128 && P.kid != GK_64) DCS_CHECK_ARG(nsplit == 1);
  if (h2) P.kid |= GK_H2;
  if (wf) P.kid |= GK_WF;
  return DCS_OK;
}

int plan_x3w(const DcsGatherLaunch& a, GatherPlan& P) {
  const DcsConvGeom* geom = a.geom;
  int rc = check_geom(geom);
  if (rc != DCS_OK) return rc;
  DCS_CHECK_ARG(a.src && a.wgt && a.dst && dcs_aligned16(a.src) && dcs_aligned16(a.wgt) && geom->dst_cstride >= geom->Cout);
  DCS_CHECK_ARG(a.nsplit <= 1);
  if (geom->stem) {                       // the 7x7 stem forward with its input patch in LDS (fp16 two-piece form only)
    if (!stem7_eligible(geom) || !(a.accumulate & DCS_ACC_FP16X2) || a.bias || a.pro || a.bn_y || a.src_max)
      return DCS_E_UNSUPPORTED;
    DCS_CHECK_ARG((a.accumulate & ~(1 | DCS_ACC_FP16X2)) == 0 && !(a.stats && (a.accumulate & 1)));
    const long long M = (long long)geom->N * geom->TY * geom->TX;
    DCS_CHECK_ARG(M < 0x7FFFFF00ll && geom->dst_cstride >= 64);
    P.g = *geom;
    P.s = GatherSub{a.src, reinterpret_cast<const unsigned char*>(a.wgt), nullptr, a.dst, a.stats, nullptr,
                    BnBwdEpi{nullptr, nullptr, nullptr, 0}, 0ll,
                    (a.accumulate & 1) | (dcs_streams(M * geom->dst_cstride * 4) ? 2 : 0), 1, 0, 2,
                    (unsigned)(14 * 2 * 2 * 1024), 0, 0, 0, nullptr};
    P.kid = GK_STEM7;
    P.nbx = (unsigned)(M / 256);
    P.nby = 1;
    return DCS_OK;
  }
  const int bn_ = geom->Cout > 64 ? 128 : 64;
  const int th = bn_ == 64 ? 8 : 4;
  if (geom->Cout <= 32 || (geom->wstride & 15) || !conv3x3_halo_eligible(geom, th)) return DCS_E_UNSUPPORTED;
  for (int t = 0; t < 9; ++t)
    if (geom->wofs[t] & 15) return DCS_E_UNSUPPORTED;
  if ((long long)geom->SH * geom->SW * geom->src_cstride * 4 > 0x7FFFFFFFll) return DCS_E_UNSUPPORTED;
  const BnBwdEpi bnb{a.bn_y, a.bn_mask, a.bn, a.relu};
  const bool accum = (a.accumulate & 1) != 0, h2 = (a.accumulate & DCS_ACC_FP16X2) != 0;
  DCS_CHECK_ARG((a.accumulate & ~(1 | DCS_ACC_FP16X2)) == 0);
  DCS_CHECK_ARG(!(a.stats && accum && !bnb.y));
  DCS_CHECK_ARG(!a.pro || (geom->K <= DCS_PRO_MAXK && dcs_aligned16(a.pro)));
  DCS_CHECK_ARG(!bnb.y || (a.stats && bnb.bn && (geom->Cout & 3) == 0 && geom->dst_cstride == geom->Cout &&
                           dcs_aligned16(a.dst) && dcs_aligned16(bnb.y) && (!bnb.mask || dcs_aligned16(bnb.mask))));
  const long long M = (long long)geom->N * geom->TY * geom->TX;
  DCS_CHECK_ARG(M < 0x7FFFFF00ll);
  const int ntiles = (geom->Cout + bn_ - 1) / bn_;
  const int J = (geom->Cout + 31) / 32;
  const long long units = (long long)(geom->wstride >> 4) * J * (h2 ? 2 : 3) * 64;
  P.g = *geom;
  P.s = GatherSub{a.src, reinterpret_cast<const unsigned char*>(a.wgt), a.bias, a.dst, a.stats, a.pro, bnb, 0ll,
                  (accum ? 1 : 0) | (dcs_streams(M * geom->dst_cstride * 4) ? 2 : 0), ntiles, 0, J,
                  (unsigned)(units * 16), 0, 0, 0, h2 ? a.src_max : nullptr};
  P.kid = h2 ? (bn_ == 64 ? GK_X2H_64 : GK_X2H_128) : (bn_ == 64 ? GK_X3W_64 : GK_X3W_128);
  P.nbx = (unsigned)((M / (bn_ == 64 ? 256 : 128)) * ntiles);
  P.nby = 1;
  return DCS_OK;
}

#define DCS_GATHER_ARGS(P) (P).s.src, (P).s.w, (P).s.bias, (P).s.dst, (P).g, (P).s.accumulate, (P).s.ntiles, (P).s.stats
int launch_gather_one(const GatherPlan& P, hipStream_t s) {
  const dim3 grid(P.nbx, P.nby), blk(256);
  switch (P.kid) {
    case GK_X3W_64:
      hipLaunchKernelGGL((conv3x3_x3w_kernel<64, 8>), grid, blk, 0, s, DCS_GATHER_ARGS(P), P.s.bnb, P.s.pro, P.s.J, P.s.neg_off,
                         P.s.src_max);
      break;
    case GK_X3W_128:
      hipLaunchKernelGGL((conv3x3_x3w_kernel<128, 4>), grid, blk, 0, s, DCS_GATHER_ARGS(P), P.s.bnb, P.s.pro, P.s.J, P.s.neg_off,
                         P.s.src_max);
      break;
    case GK_X2H_64:
      if (dcs_config().x3w_db)
        hipLaunchKernelGGL((conv3x3_x3w_db_kernel<64, 8>), grid, blk, 0, s, DCS_GATHER_ARGS(P), P.s.bnb, P.s.pro, P.s.J, P.s.neg_off,
                           P.s.src_max);
      else
        hipLaunchKernelGGL((conv3x3_x3w_kernel<64, 8, 2>), grid, blk, 0, s, DCS_GATHER_ARGS(P), P.s.bnb, P.s.pro, P.s.J, P.s.neg_off,
                           P.s.src_max);
      break;
    case GK_X2H_128:
      if (dcs_config().x3w_db)
        hipLaunchKernelGGL((conv3x3_x3w_db_kernel<128, 4>), grid, blk, 0, s, DCS_GATHER_ARGS(P), P.s.bnb, P.s.pro, P.s.J, P.s.neg_off,
                           P.s.src_max);
      else
        hipLaunchKernelGGL((conv3x3_x3w_kernel<128, 4, 2>), grid, blk, 0, s, DCS_GATHER_ARGS(P), P.s.bnb, P.s.pro, P.s.J, P.s.neg_off,
                           P.s.src_max);
      break;
    case GK_STEM7:
      hipLaunchKernelGGL(stem7_h2_kernel, grid, blk, 0, s, P.s.src, P.s.w, P.s.dst, P.g, P.s.accumulate, P.s.stats, P.s.neg_off);
      break;
    case GK_HALO_64:
      hipLaunchKernelGGL((conv3x3_x3_kernel<64, 8>), grid, blk, 0, s, DCS_GATHER_ARGS(P), P.s.bnb, P.s.pro);
      break;
    case GK_HALO_128:
      hipLaunchKernelGGL((conv3x3_x3_kernel<128, 4>), grid, blk, 0, s, DCS_GATHER_ARGS(P), P.s.bnb, P.s.pro);
      break;
#define DCS_TAP(BN_, BM_, ST_)                                                                                                  \
  do {                                                                                                                        \
    if ((P.kid & GK_WF) && !ST_)                                                                                              \
      hipLaunchKernelGGL((conv_gather_x3_kernel<BN_, BM_, false, 2, true>), grid, blk, 0, s, DCS_GATHER_ARGS(P), P.s.cps,     \
                         P.s.slab_stride, P.s.bnb, P.s.pro, P.s.src_max);                                                    \
    else if (P.kid & GK_H2)                                                                                                   \
      hipLaunchKernelGGL((conv_gather_x3_kernel<BN_, BM_, ST_, 2>), grid, blk, 0, s, DCS_GATHER_ARGS(P), P.s.cps,             \
                         P.s.slab_stride, P.s.bnb, P.s.pro, P.s.src_max);                                                    \
    else                                                                                                                      \
      hipLaunchKernelGGL((conv_gather_x3_kernel<BN_, BM_, ST_, 3>), grid, blk, 0, s, DCS_GATHER_ARGS(P), P.s.cps,             \
                         P.s.slab_stride, P.s.bnb, P.s.pro, nullptr);                                                        \
  } while (0)
    case GK_STEM_256: case GK_STEM_256 | GK_H2: DCS_TAP(64, 256, true); break;
    case GK_STEM_128: case GK_STEM_128 | GK_H2: DCS_TAP(64, 128, true); break;
    case GK_128: case GK_128 | GK_H2: case GK_128 | GK_H2 | GK_WF: DCS_TAP(128, 128, false); break;
    case GK_64_256: case GK_64_256 | GK_H2: case GK_64_256 | GK_H2 | GK_WF: DCS_TAP(64, 256, false); break;
    default: DCS_TAP(64, 128, false);
#undef DCS_TAP
  }
  return hipGetLastError() == hipSuccess ? DCS_OK : DCS_E_LAUNCH;
}

// block ranges of the sub-launches: starts padded to multiples of 8 (one round of the XCD round-robin)
template <class MP, class PL>
long long pack_multi(MP& mp, const PL* const* plans, int n) {
  long long at = 0;
  for (int i = 0; i < MULTI_MAX; ++i) {
    if (i < n) {
      mp.g[i] = plans[i]->g;
      mp.s[i] = plans[i]->s;
      mp.s[i].blk0 = (int)at;
      mp.s[i].nbx = (int)plans[i]->nbx;
      mp.s[i].nblk = (int)(plans[i]->nbx * plans[i]->nby);
      at += ((long long)mp.s[i].nblk + 7) / 8 * 8;
    } else {
      mp.g[i] = mp.g[0];
      mp.s[i] = mp.s[0];
      mp.s[i].blk0 = 0x7FFFFFFF;
      mp.s[i].nblk = 0;
    }
  }
  return at;
}

// same-kernel plans (n >= 2) as one launch
int launch_gather_multi(const GatherPlan* const* plans, int n, hipStream_t s) {
  GatherMulti mp;
  const long long total = pack_multi(mp, plans, n);
  if (total <= 0 || total >= (1ll << 31)) return DCS_E_ARG;
  const dim3 grid((unsigned)total), blk(256);
  switch (plans[0]->kid) {
    case GK_X3W_64: hipLaunchKernelGGL((conv3x3_x3w_multi_kernel<64, 8>), grid, blk, 0, s, mp); break;
    case GK_X3W_128: hipLaunchKernelGGL((conv3x3_x3w_multi_kernel<128, 4>), grid, blk, 0, s, mp); break;
    case GK_X2H_64:
      if (dcs_config().x3w_db) hipLaunchKernelGGL((conv3x3_x3w_db_multi_kernel<64, 8>), grid, blk, 0, s, mp);
      else hipLaunchKernelGGL((conv3x3_x3w_multi_kernel<64, 8, 2>), grid, blk, 0, s, mp);
      break;
    case GK_X2H_128:
      if (dcs_config().x3w_db) hipLaunchKernelGGL((conv3x3_x3w_db_multi_kernel<128, 4>), grid, blk, 0, s, mp);
      else hipLaunchKernelGGL((conv3x3_x3w_multi_kernel<128, 4, 2>), grid, blk, 0, s, mp);
      break;
    case GK_STEM7: hipLaunchKernelGGL(stem7_h2_multi_kernel, grid, blk, 0, s, mp); break;
    case GK_STEM_256: hipLaunchKernelGGL((conv_gather_x3_multi_kernel<64, 256, true>), grid, blk, 0, s, mp); break;
    case GK_STEM_128: hipLaunchKernelGGL((conv_gather_x3_multi_kernel<64, 128, true>), grid, blk, 0, s, mp); break;
    case GK_128: hipLaunchKernelGGL(conv_gather_x3_multi_kernel<128>, grid, blk, 0, s, mp); break;
    case GK_64_256: hipLaunchKernelGGL((conv_gather_x3_multi_kernel<64, 256>), grid, blk, 0, s, mp); break;
    case GK_64: hipLaunchKernelGGL(conv_gather_x3_multi_kernel<64>, grid, blk, 0, s, mp); break;
    case GK_STEM_256 | GK_H2: hipLaunchKernelGGL((conv_gather_x3_multi_kernel<64, 256, true, 2>), grid, blk, 0, s, mp); break;
    case GK_STEM_128 | GK_H2: hipLaunchKernelGGL((conv_gather_x3_multi_kernel<64, 128, true, 2>), grid, blk, 0, s, mp); break;
    case GK_128 | GK_H2: hipLaunchKernelGGL((conv_gather_x3_multi_kernel<128, 128, false, 2>), grid, blk, 0, s, mp); break;
    case GK_64_256 | GK_H2: hipLaunchKernelGGL((conv_gather_x3_multi_kernel<64, 256, false, 2>), grid, blk, 0, s, mp); break;
    case GK_64 | GK_H2: hipLaunchKernelGGL((conv_gather_x3_multi_kernel<64, 128, false, 2>), grid, blk, 0, s, mp); break;
    case GK_128 | GK_H2 | GK_WF: hipLaunchKernelGGL((conv_gather_x3_multi_kernel<128, 128, false, 2, true>), grid, blk, 0, s, mp); break;
    case GK_64_256 | GK_H2 | GK_WF: hipLaunchKernelGGL((conv_gather_x3_multi_kernel<64, 256, false, 2, true>), grid, blk, 0, s, mp); break;
    case GK_64 | GK_H2 | GK_WF: hipLaunchKernelGGL((conv_gather_x3_multi_kernel<64, 128, false, 2, true>), grid, blk, 0, s, mp); break;
    default: return DCS_E_ARG;
  }
  return hipGetLastError() == hipSuccess ? DCS_OK : DCS_E_LAUNCH;
}

bool gather_kid_batches(int kid) { return kid != GK_HALO_64 && kid != GK_HALO_128; }

// n validated plans -> launches: plans that chose the same kernel share one grid, the others go out on their own
int launch_gather_plans(const GatherPlan* plans, int n, hipStream_t s) {
  bool done[DCS_MULTI_MAX] = {};
  for (int i = 0; i < n; ++i) {
    if (done[i]) continue;
    const GatherPlan* grp[DCS_MULTI_MAX];
    int m = 0;
    grp[m++] = &plans[i];
    done[i] = true;
    if (gather_kid_batches(plans[i].kid))
      for (int j = i + 1; j < n; ++j)
        if (!done[j] && plans[j].kid == plans[i].kid) { grp[m++] = &plans[j]; done[j] = true; }
    const int rc = m == 1 ? launch_gather_one(*grp[0], s) : launch_gather_multi(grp, m, s);
    if (rc != DCS_OK) return rc;
  }
  return DCS_OK;
}

enum WgradKid { WK_STEM, WK_ROLL, WK_ROLL_H2, WK_NINE, WK_128, WK_64, WK_128_H2, WK_64_H2, WK_STEM_H2 };
struct WgradPlan { int kid; DcsConvGeom g; WgradSub s; unsigned nbx, nby; };

int plan_wgrad_x3(const DcsWgradLaunch& a, WgradPlan& P) {
  const DcsConvGeom* geom = a.geom;
  const int dy_cstride = a.dy_cstride, nsplit = a.nsplit, split0 = a.split0;
  int rc = check_geom(geom);
  if (rc != DCS_OK) return rc;
  DCS_CHECK_ARG(a.src && a.dy && a.slab && dcs_aligned16(a.src) && dcs_aligned16(a.dy) && (!a.pro || dcs_aligned16(a.pro)));
  DCS_CHECK_ARG((dy_cstride & 3) == 0 && dy_cstride >= geom->Cout && nsplit > 0 && nsplit < 65536 && split0 >= 0);
  DCS_CHECK_ARG(geom->dsy == 1 && geom->dsx == 1 && geom->dy0 == 0 && geom->dx0 == 0 &&
                geom->TY == geom->DH && geom->TX == geom->DW);
  P.g = *geom;
  P.s = WgradSub{a.src, a.dy, a.slab, a.pro, 0ll, dy_cstride, split0, 0, 0, 0, 0, 0, nullptr};
  P.nby = (unsigned)nsplit;
  if (geom->stem) {                    // the seven-tap stem geometry (ops.geom_stem): its own kernel
    if ((geom->TX & 15) || geom->Cout != 64 || geom->wstride != 224 || geom->ntaps != 7 || a.pro) return DCS_E_UNSUPPORTED;
    const long long nchunks = (long long)geom->N * geom->TY * (geom->TX / 16);
    const int cps = (int)((nchunks + nsplit - 1) / nsplit);
    const long long span = ((long long)cps * 32 + 8ll * geom->SW) * 16;
    if ((long long)cps * 16 * dy_cstride * 4 >= 0x7FFFFFFFll || span >= 0x7FFFFFFFll) return DCS_E_UNSUPPORTED;
    P.kid = a.dy_max ? WK_STEM_H2 : WK_STEM; P.s.dy_max = a.dy_max; P.s.cps = cps; P.nbx = (unsigned)nsplit; P.nby = 1;
    return DCS_OK;
  }
  if (geom->Cout & 3) return DCS_E_UNSUPPORTED;
  const long long M = (long long)geom->N * geom->TY * geom->TX;
  long long mps = (M + nsplit - 1) / nsplit;
  mps = (mps + 31) / 32 * 32;
  const long long span_src = (mps * geom->sy * geom->sx + 4ll * geom->SW) * geom->src_cstride * 4;
  if (mps * (long long)dy_cstride * 4 >= 0x7FFFFFFFll || span_src >= 0x7FFFFFFFll) return DCS_E_UNSUPPORTED;
  if (wgrad3x3_x3_eligible(geom)) {
    const long long nchunks = (long long)geom->N * geom->TY * (geom->TX / 16);
    const int cps = (int)((nchunks + nsplit - 1) / nsplit);
    const int coT = (geom->Cout + 63) / 64, ciT = (geom->K + 63) / 64;
    const long long span = ((long long)cps * 16 + 4ll * geom->SW) * geom->src_cstride * 4;
    P.s.cps = cps; P.s.ciT = ciT; P.nbx = (unsigned)(coT * ciT);
    {
      // rolling-window kernel: units are (image, strip, row); a split may reach into the following image(s)
      const long long per_img = (long long)geom->TY * (geom->TX / 16);
      long long imgs = cps / per_img + 2;
      if (imgs > geom->N) imgs = geom->N;
      const long long xbytes = imgs * geom->SH * geom->SW * geom->src_cstride * 4;
      const long long dbytes = imgs * geom->TY * geom->TX * dy_cstride * 4;
      if (dcs_config().wgrad_roll != 0 && xbytes < 0x7FFFFFFFll && dbytes < 0x7FFFFFFFll) {
        P.kid = a.dy_max ? WK_ROLL_H2 : WK_ROLL;                    // dy_max: the fp16 two-piece form (this kernel only)
        P.s.dy_max = a.dy_max;
        return DCS_OK;
      }
    }
    if ((long long)cps * 16 * dy_cstride * 4 < 0x7FFFFFFFll && span < 0x7FFFFFFFll) {
      P.kid = WK_NINE;
      return DCS_OK;
    }
  }
  const int bt = (geom->Cout > 64 && geom->K > 64) ? 128 : 64;
  const int coT = (geom->Cout + bt - 1) / bt, ciT = (geom->K + bt - 1) / bt;
  P.kid = bt == 128 ? (a.dy_max ? WK_128_H2 : WK_128) : (a.dy_max ? WK_64_H2 : WK_64);      // dy_max: the fp16 two-piece form
  P.s.dy_max = a.dy_max;
  P.s.mps = mps; P.s.ciT = ciT; P.s.cps = 0;
  P.nbx = (unsigned)(geom->ntaps * coT * ciT);
  return DCS_OK;
}

#define DCS_WGRAD_ARGS(P) (P).s.src, (P).s.dy, (P).s.slab, (P).g, (P).s.dy_cstride, (P).s.split0
int launch_wgrad_one(const WgradPlan& P, hipStream_t s) {
  const dim3 grid(P.nbx, P.nby), blk(256);
  switch (P.kid) {
    case WK_STEM: hipLaunchKernelGGL(stem_wgrad_x3_kernel<3>, grid, blk, 0, s, DCS_WGRAD_ARGS(P), P.s.cps, nullptr); break;
    case WK_STEM_H2: hipLaunchKernelGGL(stem_wgrad_x3_kernel<2>, grid, blk, 0, s, DCS_WGRAD_ARGS(P), P.s.cps, P.s.dy_max); break;
    case WK_ROLL: hipLaunchKernelGGL(conv_wgrad3x3_x3r_kernel<3>, grid, blk, 0, s, DCS_WGRAD_ARGS(P), P.s.cps, P.s.ciT, P.s.pro, nullptr); break;
    case WK_ROLL_H2: hipLaunchKernelGGL(conv_wgrad3x3_x3r_kernel<2>, grid, blk, 0, s, DCS_WGRAD_ARGS(P), P.s.cps, P.s.ciT, P.s.pro, P.s.dy_max); break;
    case WK_NINE: hipLaunchKernelGGL(conv_wgrad3x3_x3_kernel, grid, blk, 0, s, DCS_WGRAD_ARGS(P), P.s.cps, P.s.ciT, P.s.pro); break;
    case WK_128: hipLaunchKernelGGL((conv_wgrad_x3_kernel<128, 3>), grid, blk, 0, s, DCS_WGRAD_ARGS(P), P.s.mps, P.s.ciT, P.s.pro, nullptr); break;
    case WK_128_H2: hipLaunchKernelGGL((conv_wgrad_x3_kernel<128, 2>), grid, blk, 0, s, DCS_WGRAD_ARGS(P), P.s.mps, P.s.ciT, P.s.pro, P.s.dy_max); break;
    case WK_64_H2: hipLaunchKernelGGL((conv_wgrad_x3_kernel<64, 2>), grid, blk, 0, s, DCS_WGRAD_ARGS(P), P.s.mps, P.s.ciT, P.s.pro, P.s.dy_max); break;
    default: hipLaunchKernelGGL((conv_wgrad_x3_kernel<64, 3>), grid, blk, 0, s, DCS_WGRAD_ARGS(P), P.s.mps, P.s.ciT, P.s.pro, nullptr);
  }
  return hipGetLastError() == hipSuccess ? DCS_OK : DCS_E_LAUNCH;
}

int launch_wgrad_multi(const WgradPlan* const* plans, int n, hipStream_t s) {
  WgradMulti mp;
  const long long total = pack_multi(mp, plans, n);
  if (total <= 0 || total >= (1ll << 31)) return DCS_E_ARG;
  const dim3 grid((unsigned)total), blk(256);
  switch (plans[0]->kid) {
    case WK_STEM: hipLaunchKernelGGL(stem_wgrad_x3_multi_kernel<3>, grid, blk, 0, s, mp); break;
    case WK_STEM_H2: hipLaunchKernelGGL(stem_wgrad_x3_multi_kernel<2>, grid, blk, 0, s, mp); break;
    case WK_ROLL: hipLaunchKernelGGL(conv_wgrad3x3_x3r_multi_kernel<3>, grid, blk, 0, s, mp); break;
    case WK_ROLL_H2: hipLaunchKernelGGL(conv_wgrad3x3_x3r_multi_kernel<2>, grid, blk, 0, s, mp); break;
    case WK_128: hipLaunchKernelGGL((conv_wgrad_x3_multi_kernel<128, 3>), grid, blk, 0, s, mp); break;
    case WK_64: hipLaunchKernelGGL((conv_wgrad_x3_multi_kernel<64, 3>), grid, blk, 0, s, mp); break;
    case WK_128_H2: hipLaunchKernelGGL((conv_wgrad_x3_multi_kernel<128, 2>), grid, blk, 0, s, mp); break;
    case WK_64_H2: hipLaunchKernelGGL((conv_wgrad_x3_multi_kernel<64, 2>), grid, blk, 0, s, mp); break;
    default: return DCS_E_ARG;
  }
  return hipGetLastError() == hipSuccess ? DCS_OK : DCS_E_LAUNCH;
}

int launch_wgrad_plans(const WgradPlan* plans, int n, hipStream_t s) {
  bool done[DCS_MULTI_MAX] = {};
  for (int i = 0; i < n; ++i) {
    if (done[i]) continue;
    const WgradPlan* grp[DCS_MULTI_MAX];
    int m = 0;
    grp[m++] = &plans[i];
    done[i] = true;
    if (plans[i].kid != WK_NINE)
      for (int j = i + 1; j < n; ++j)
        if (!done[j] && plans[j].kid == plans[i].kid) { grp[m++] = &plans[j]; done[j] = true; }
    const int rc = m == 1 ? launch_wgrad_one(*grp[0], s) : launch_wgrad_multi(grp, m, s);
    if (rc != DCS_OK) return rc;
  }
  return DCS_OK;
}

}  // namespace

extern "C" int dcs_conv_gather_x3(const float* src, const void* wsplit, const float* bias, float* dst,
                                  const DcsConvGeom* geom, int accumulate, float* stats, const float* pro,
                                  const float* bn_y, const float* bn_mask, const float* bn, int relu, int nsplit,
                                  int64_t slab_stride, const uint32_t* src_max, void* stream) {
  const DcsGatherLaunch a{src, wsplit, bias, dst, geom, stats, pro, bn_y, bn_mask, bn, slab_stride, accumulate, relu, nsplit, src_max};
  GatherPlan P;
  const int rc = plan_gather_x3(a, P);
  return rc != DCS_OK ? rc : launch_gather_one(P, dcs_stream(stream));
}

extern "C" int dcs_conv_gather_x3_multi(const DcsGatherLaunch* launches, int n, void* stream) {
  DCS_CHECK_ARG(launches && n >= 1 && n <= DCS_MULTI_MAX);
  GatherPlan P[DCS_MULTI_MAX];
  for (int i = 0; i < n; ++i) {                 // nothing is launched unless every sub-launch is valid
    const int rc = plan_gather_x3(launches[i], P[i]);
    if (rc != DCS_OK) return rc;
  }
  return launch_gather_plans(P, n, dcs_stream(stream));
}

extern "C" int dcs_conv_wgrad_x3(const float* src, const float* dy, float* slab, const DcsConvGeom* geom, int dy_cstride,
                                 int split0, int nsplit, const float* pro, const uint32_t* dy_max, void* stream) {
  const DcsWgradLaunch a{src, dy, slab, geom, pro, dy_cstride, split0, nsplit, dy_max};
  WgradPlan P;
  const int rc = plan_wgrad_x3(a, P);
  return rc != DCS_OK ? rc : launch_wgrad_one(P, dcs_stream(stream));
}

extern "C" int dcs_conv_wgrad_x3_multi(const DcsWgradLaunch* launches, int n, void* stream) {
  DCS_CHECK_ARG(launches && n >= 1 && n <= DCS_MULTI_MAX);
  WgradPlan P[DCS_MULTI_MAX];
  for (int i = 0; i < n; ++i) {
    const int rc = plan_wgrad_x3(launches[i], P[i]);
    if (rc != DCS_OK) return rc;
  }
  return launch_wgrad_plans(P, n, dcs_stream(stream));
}

extern "C" int dcs_split_weight_frag(const float* w, void* out, int64_t rows, int wstride, void* stream) {
  DCS_CHECK_ARG(w && out && rows > 0 && wstride > 0 && (wstride & 15) == 0 && dcs_aligned16(w) && dcs_aligned16(out));
  const int J = (int)((rows + 31) / 32);
  const long long units = (long long)(wstride >> 4) * J * 3 * 64;
  const long long nthreads = (long long)(wstride >> 4) * J * 64;
  DCS_CHECK_ARG(units * 32 < 0x7FFFFFFFll);
  hipLaunchKernelGGL(split_weight_frag_kernel, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, dcs_stream(stream), w,
                     reinterpret_cast<u32x4*>(out), (int)rows, wstride, J, units);
  DCS_LAUNCH_RET();
}

extern "C" int dcs_split_weight_frag_h2(const float* w, void* out, int64_t rows, int wstride, void* stream) {
  DCS_CHECK_ARG(w && out && rows > 0 && wstride > 0 && (wstride & 15) == 0 && dcs_aligned16(w) && dcs_aligned16(out));
  const int J = (int)((rows + 31) / 32);
  const long long units = (long long)(wstride >> 4) * J * 2 * 64;
  const long long nthreads = (long long)(wstride >> 4) * J * 64;
  DCS_CHECK_ARG(units * 32 < 0x7FFFFFFFll);
  hipLaunchKernelGGL(split_weight_frag_h2_kernel, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, dcs_stream(stream), w,
                     reinterpret_cast<u32x4*>(out), (int)rows, wstride, J, units);
  DCS_LAUNCH_RET();
}

// dcs_conv_gather_x3 for dense 3x3 / stride 1 geometries with the weights in dcs_split_weight_frag layout
extern "C" int dcs_conv3x3_x3w(const float* src, const void* wfrag, const float* bias, float* dst, const DcsConvGeom* geom,
                               int accumulate, float* stats, const float* pro, const float* bn_y, const float* bn_mask,
                               const float* bn, int relu, const uint32_t* src_max, void* stream) {
  const DcsGatherLaunch a{src, wfrag, bias, dst, geom, stats, pro, bn_y, bn_mask, bn, 0, accumulate, relu, 1, src_max};
  GatherPlan P;
  const int rc = plan_x3w(a, P);
  return rc != DCS_OK ? rc : launch_gather_one(P, dcs_stream(stream));
}

extern "C" int dcs_conv3x3_x3w_multi(const DcsGatherLaunch* launches, int n, void* stream) {
  DCS_CHECK_ARG(launches && n >= 1 && n <= DCS_MULTI_MAX);
  GatherPlan P[DCS_MULTI_MAX];
  for (int i = 0; i < n; ++i) {
    const int rc = plan_x3w(launches[i], P[i]);
    if (rc != DCS_OK) return rc;
  }
  return launch_gather_plans(P, n, dcs_stream(stream));
}
