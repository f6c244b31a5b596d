// Segmentation losses, hard-anchor sampling and the row-wise part of the contrastive losses.
// Replaces utils/loss.py:39-80 (BoundaryAwareFocalLoss), nn.CrossEntropyLoss, utils/loss.py:264-337
// (_hard_anchor_sampling: counting / index selection; the random permutation stays on the host CPU
// generator exactly like the reference) and utils/loss.py:175-204,:361-386 (row reductions of the
// similarity matrix) in the reference.
#include "dcs_common.h"

namespace {

// ---------------------------------------------------------------------------------------------------
// seg loss: one thread per pixel, C logit planes (NCHW -> coalesced across lanes)
template <int MAXC>
__global__ __launch_bounds__(256)
void seg_loss_kernel(const float* __restrict__ logits, int64_t* __restrict__ target, const float* __restrict__ ldw,
                     const float* __restrict__ cw, float* __restrict__ grad, float* __restrict__ partial, int N, int C,
                     long long HW, int mode, float gamma, int ignore) {
  __shared__ double sl[256];
  __shared__ double sc[256];
  const long long total = (long long)N * HW;
  double lsum = 0.0, lcnt = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long n = i / HW, p = i - n * HW;
    const float* lp = logits + n * C * HW + p;
    float* gp = grad + n * C * HW + p;
    long long t = target[i];
    float coef;
    bool counted;
    if (mode == 4) {                       // cross entropy, ignore_index
      counted = t != ignore;
      coef = counted ? 1.f : 0.f;
      if (!counted || t < 0 || t >= C) t = 0;
    } else {
      if (t == ignore) { t = 0; target[i] = 0; }      // utils/loss.py:43 (in place)
      if (t < 0 || t >= C) t = 0;
      const float a = ldw[i];
      counted = a > 0.f;                              // N = (ldw > 0).sum(), loss.py:45
      coef = 1.f;                                     // filled below (needs pt)
    }
    float v[MAXC];
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      v[c] = c < C ? lp[(long long)c * HW] : -INFINITY;
      mx = fmaxf(mx, v[c]);
    }
    float se = 0.f, xt = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      v[c] = c < C ? expf(v[c] - mx) : 0.f;
      se += v[c];
      if (c == (int)t) xt = c < C ? lp[(long long)c * HW] : 0.f;
    }
    const float logpt = xt - mx - logf(se);
    if (mode != 4) {
      const float pt = expf(logpt);
      const float mod = expf(gamma * (1.f - pt));
      const float w = cw ? cw[t] : 1.f;
      const float a = ldw[i];
      coef = mode == 0 ? w * a * mod : (mode == 1 ? mod : (mode == 2 ? a * mod : w * mod));
    }
    lsum += (double)(-coef * logpt);
    lcnt += counted ? 1.0 : 0.0;
    const float inv = 1.f / se;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < C) gp[(long long)c * HW] = -coef * ((c == (int)t ? 1.f : 0.f) - v[c] * inv);
  }
  sl[threadIdx.x] = lsum; sc[threadIdx.x] = lcnt;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) { sl[threadIdx.x] += sl[threadIdx.x + o]; sc[threadIdx.x] += sc[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { partial[2 * blockIdx.x] = (float)sl[0]; partial[2 * blockIdx.x + 1] = (float)sc[0]; }
}

// ---------------------------------------------------------------------------------------------------
// Fused logits upsampling + log-softmax + focal / cross-entropy loss + gradient (network/utils.py:8 `upsample`,
// utils/loss.py:41-42, :60-73): the full-resolution logits [N,C,H,W] (2.55 GB at C3) and their gradient are never
// materialised.  Input = the low-resolution NHWC logits [N,ih,iw,cs]; output = d(sum loss)/d(low-res logits) in the
// same layout, i.e. the adjoint of the bilinear upsampling applied on the fly.  H = F*ih, W = F*iw, F in {2, 4}.
// Interpolation weights: lin_src, the very weights (and association) of upsample_to_nchw_kernel.  Exponentials are
// v_exp_f32 of x * log2(e) (arguments are <= 0 after the max shift, relative error <= 1e-6).
//
// Round 3 -- the CELL form (seg_loss_cells_kernel).  The first fused kernel (round 2: one pass per output pixel for the
// log-sum-exp, then a gather per low-resolution pixel that re-evaluated the 19-class softmax of every output pixel of its
// footprint: 5 softmax evaluations per output pixel, all operands through LDS) was instruction-bound at 2.9 ms for C3 (3 % of
// the HBM roof).  With align_corners=False and an integer factor F the F x F outputs {F k + F/2 .. F k + F/2 + F - 1}^2 share
// their four bilinear taps: the low-resolution pixels (k, k+1) x (k', k'+1) -- a CELL.  One thread owns one cell: per cell
// column it forms the x-interpolated top / bottom rows once, per output pixel it interpolates in y, evaluates ONE softmax
// in registers, and folds the gradient coef (p - onehot) back onto the cell's four corners with the same separable
// weights (rows first, then columns).  A low-resolution pixel is a corner of four cells: the per-corner sums go through
// LDS once and every pixel adds its <= 9 terms in fixed order (bitwise reproducible, no float atomics).  A block owns
// 7 x 31 low-resolution pixels and evaluates the 8 x 32 cells that touch them (cells -1 .. 6 relative to the tile: 15 %
// redundant cell work at tile borders instead of a second kernel or atomics); loss, count and the in-place 255 -> 0 label
// rewrite (utils/loss.py:43) belong to the one block that OWNS the cell.
constexpr int SLC_CH = 8, SLC_CW = 32;                    // cells per block = threads
constexpr int SLC_PH = SLC_CH - 1, SLC_PW = SLC_CW - 1;   // low-resolution pixels owned per block

template <int F, int MAXC>
__global__ __launch_bounds__(SLC_CH * SLC_CW, 2)
void seg_loss_cells_kernel(const float* __restrict__ lr, const int cs, int64_t* __restrict__ target,
                           const float* __restrict__ ldw, const float* __restrict__ cw, float* __restrict__ glr,
                           float* __restrict__ partial, const int C, const int ih, const int iw, const int mode,
                           const float gamma, const int ignore, const int tiles_x, const int tiles_y) {
  constexpr int NT = SLC_CH * SLC_CW;
  constexpr int LH = SLC_CH + 1, LW = SLC_CW + 1;                 // staged low-res pixels: the corners of the block's cells
  constexpr int LP = 21;                                          // odd pixel stride (>= MAXC): conflict-free reads across cells
  constexpr int BS = 43;                                          // odd per-cell stride of the corner-sum exchange buffer
  constexpr int NOPIX = (int)0x80000000;                          // "no such output pixel"
  static_assert(MAXC <= 20, "LP / BS are sized for <= 20 classes");
  constexpr float LOG2E_ = 1.4426950408889634f;
  __shared__ float tile[LH * LW * LP];
  __shared__ float xch[NT * BS];
  __shared__ double s_red[2][NT / 64];
  const int tid = threadIdx.x;
  int b = blockIdx.x;
  const int tx = b % tiles_x; b /= tiles_x;
  const int ty = b % tiles_y;
  const int n = b / tiles_y;
  const int iy0 = ty * SLC_PH, ix0 = tx * SLC_PW;
  const int H = F * ih, W = F * iw;
  const float sc = 1.f / (float)F;
  // ---- stage: tile row ly / column lx = low-res pixel clamp(iy0 - 1 + ly), clamp(ix0 - 1 + lx)
  for (int e = tid; e < LH * LW; e += NT) {
    const int ly = e / LW, lx = e - ly * LW;
    int gy = iy0 - 1 + ly, gx = ix0 - 1 + lx;
    gy = gy < 0 ? 0 : (gy > ih - 1 ? ih - 1 : gy);
    gx = gx < 0 ? 0 : (gx > iw - 1 ? iw - 1 : gx);
    const float* src = lr + (((long long)n * ih + gy) * iw + gx) * cs;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) tile[e * LP + c] = c < C ? src[c] : 0.f;
  }
  __syncthreads();
  // ---- the thread's cell: ky = iy0 - 1 + cy, kx = ix0 - 1 + cx; its corners are tile rows cy, cy + 1 / columns cx, cx + 1
  const int cy = tid / SLC_CW, cx = tid - cy * SLC_CW;
  const int ky = iy0 - 1 + cy, kx = ix0 - 1 + cx;
  const bool cell_ok = ky <= ih - 1 && kx <= iw - 1;
  const bool owned = cell_ok && (cy >= 1 || iy0 == 0) && (cx >= 1 || ix0 == 0);
  const int oyb = F * ky + F / 2, oxb = F * kx + F / 2;           // first output row / column of the cell (may be < 0)
  float acc[2][2][MAXC];                                          // [corner row][corner column][class]
#pragma unroll
  for (int c = 0; c < MAXC; ++c) { acc[0][0][c] = 0.f; acc[0][1][c] = 0.f; acc[1][0][c] = 0.f; acc[1][1][c] = 0.f; }
  double lsum = 0.0, lcnt = 0.0;
  if (cell_ok) {
    // labels / boundary weights of the cell's F x F outputs: all loads in flight before the arithmetic starts
    int tg[F][F];
    float aw[F][F];
#pragma unroll
    for (int ry = 0; ry < F; ++ry) {
      const int oy = oyb + ry;
      const bool rok = oy >= 0 && oy < H;
      const long long rowp = ((long long)n * H + (rok ? oy : 0)) * W;
#pragma unroll
      for (int rx = 0; rx < F; ++rx) {
        const int ox = oxb + rx;
        const bool ok = rok && ox >= 0 && ox < W;
        tg[ry][rx] = ok ? (int)target[rowp + ox] : NOPIX;
        aw[ry][rx] = (ok && mode != 4) ? ldw[rowp + ox] : 0.f;
      }
    }
    const float* t00 = &tile[(cy * LW + cx) * LP];
    const float* t01 = t00 + LP;
    const float* t10 = t00 + LW * LP;
    const float* t11 = t10 + LP;
#pragma unroll
    for (int rx = 0; rx < F; ++rx) {
      const int ox = oxb + rx;
      if (ox < 0 || ox >= W) continue;
      const Lin lx = lin_src(ox, sc, iw);
      float top[MAXC], bot[MAXC], a0[MAXC], a1[MAXC];
#pragma unroll
      for (int c = 0; c < MAXC; ++c) {
        // same association as upsample_to_nchw_kernel: x first, then y
        top[c] = fmaf(lx.w1, t01[c], lx.w0 * t00[c]);
        bot[c] = fmaf(lx.w1, t11[c], lx.w0 * t10[c]);
        a0[c] = 0.f; a1[c] = 0.f;
      }
#pragma unroll
      for (int ry = 0; ry < F; ++ry) {
        int t = tg[ry][rx];
        if (t == NOPIX) continue;
        const int oy = oyb + ry;
        const Lin ly = lin_src(oy, sc, ih);
        const float a = aw[ry][rx];
        bool counted;
        if (mode == 4) {
          counted = t != ignore;
          if (!counted || t < 0 || t >= C) t = 0;
        } else {
          if (t == ignore) { t = 0; if (owned) target[((long long)n * H + oy) * W + ox] = 0; }   // utils/loss.py:43 (in place)
          if (t < 0 || t >= C) t = 0;
          counted = a > 0.f;
        }
        float v[MAXC], mx = -INFINITY, xt = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
          v[c] = c < C ? fmaf(ly.w1, bot[c], ly.w0 * top[c]) : -INFINITY;
          mx = fmaxf(mx, v[c]);
          xt = c == t ? v[c] : xt;
        }
        const float mb = -mx * LOG2E_;
        float se = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) { v[c] = c < C ? __builtin_amdgcn_exp2f(fmaf(v[c], LOG2E_, mb)) : 0.f; se += v[c]; }
        const float logpt = xt - (mx + logf(se));
        float coef;
        if (mode == 4) coef = counted ? 1.f : 0.f;
        else {
          const float pt = __expf(logpt);
          const float mod = __expf(gamma * (1.f - pt));
          const float w = cw ? cw[t] : 1.f;
          coef = mode == 0 ? w * a * mod : (mode == 1 ? mod : (mode == 2 ? a * mod : w * mod));
        }
        if (owned) { lsum += (double)(-coef * logpt); lcnt += counted ? 1.0 : 0.0; }
        // d(-coef logpt)/dv_c = coef (p_c - [c == t])   (pt is detached: loss.py:63); rows first
        const float cp = coef / se;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
          const float g = fmaf(v[c], cp, c == t ? -coef : 0.f);
          a0[c] = fmaf(ly.w0, g, a0[c]);
          a1[c] = fmaf(ly.w1, g, a1[c]);
        }
      }
#pragma unroll
      for (int c = 0; c < MAXC; ++c) {
        acc[0][0][c] = fmaf(lx.w0, a0[c], acc[0][0][c]); acc[0][1][c] = fmaf(lx.w1, a0[c], acc[0][1][c]);
        acc[1][0][c] = fmaf(lx.w0, a1[c], acc[1][0][c]); acc[1][1][c] = fmaf(lx.w1, a1[c], acc[1][1][c]);
      }
    }
  }
  // ---- corner sums -> low-resolution pixels.  Thread = owned pixel (py, px); pixel p is corner 1 of cell p - 1 and corner 0
  // of cell p; clamped borders add corner 0 of cell -1 to pixel 0 and corner 1 of cell in - 1 to pixel in - 1.
  const int py = tid / SLC_PW, px = tid - py * SLC_PW;
  const int iy = iy0 + py, ix = ix0 + px;
  const bool pix_ok = tid < SLC_PH * SLC_PW && iy < ih && ix < iw;
  float gs[MAXC];
#pragma unroll
  for (int c = 0; c < MAXC; ++c) gs[c] = 0.f;
#pragma unroll
  for (int r = 0; r < 2; ++r) {                                   // corner row r of every cell, both corner columns
#pragma unroll
    for (int c = 0; c < MAXC; ++c) { xch[tid * BS + c] = acc[r][0][c]; xch[tid * BS + 21 + c] = acc[r][1][c]; }
    __syncthreads();
    if (pix_ok) {
      // cell rows whose corner row r is this pixel row: r == 1 -> cell py (ky = iy - 1), r == 0 -> cell py + 1 (ky = iy);
      // border: iy == 0 also takes corner row 0 of cell py (ky = -1); iy == ih - 1 also corner row 1 of cell py + 1
      int cys[2], ncy = 0;
      if (r == 1) { cys[ncy++] = py; if (iy == ih - 1) cys[ncy++] = py + 1; }
      else { if (iy == 0) cys[ncy++] = py; cys[ncy++] = py + 1; }
      for (int a = 0; a < ncy; ++a) {
        const int cyy = cys[a];
        if (iy0 - 1 + cyy > ih - 1) continue;                     // no such cell
        // columns: corner column 1 of cell px, corner column 0 of cell px + 1 (+ the clamped borders)
        const float* rowb = &xch[(cyy * SLC_CW) * BS];
        if (ix == 0) { const float* q = rowb + px * BS;
#pragma unroll
          for (int c = 0; c < MAXC; ++c) gs[c] += q[c]; }
        { const float* q = rowb + px * BS + 21;
#pragma unroll
          for (int c = 0; c < MAXC; ++c) gs[c] += q[c]; }
        if (ix0 - 1 + px + 1 <= iw - 1) {
          const float* q = rowb + (px + 1) * BS;
#pragma unroll
          for (int c = 0; c < MAXC; ++c) gs[c] += q[c];
          if (ix == iw - 1) { const float* q1 = q + 21;
#pragma unroll
            for (int c = 0; c < MAXC; ++c) gs[c] += q1[c]; }
        }
      }
    }
    __syncthreads();
  }
  if (pix_ok) {
    float* dst = glr + (((long long)n * ih + iy) * iw + ix) * cs;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) if (c < cs) dst[c] = c < C ? gs[c] : 0.f;
    for (int c = MAXC; c < cs; ++c) dst[c] = 0.f;
  }
  // ---- block reduction of the loss / count
  lsum = dcs_wave_sum_d(lsum); lcnt = dcs_wave_sum_d(lcnt);
  if ((tid & 63) == 0) { s_red[0][tid >> 6] = lsum; s_red[1][tid >> 6] = lcnt; }
  __syncthreads();
  if (tid == 0) {
    partial[2 * blockIdx.x] = (float)((s_red[0][0] + s_red[0][1]) + (s_red[0][2] + s_red[0][3]));
    partial[2 * blockIdx.x + 1] = (float)((s_red[1][0] + s_red[1][1]) + (s_red[1][2] + s_red[1][3]));
  }
}

__global__ void seg_loss_final_kernel(const float* __restrict__ partial, float* __restrict__ out, int blocks) {
  __shared__ double sl[256];
  __shared__ double sc[256];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < blocks; i += 256) { a += (double)partial[2 * i]; b += (double)partial[2 * i + 1]; }
  sl[threadIdx.x] = a; sc[threadIdx.x] = b;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) { sl[threadIdx.x] += sl[threadIdx.x + o]; sc[threadIdx.x] += sc[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double cnt = sc[0];
    out[0] = cnt > 0.0 ? (float)(sl[0] / cnt) : 0.f;
    out[1] = (float)cnt;
    out[2] = cnt > 0.0 ? (float)(1.0 / cnt) : 0.f;
  }
}

// ---------------------------------------------------------------------------------------------------
// hard-anchor sampling, device half
__global__ __launch_bounds__(256)
void anchor_keys_kernel(const float* __restrict__ logits, int cs, int C, const int64_t* __restrict__ labels, int h,
                        int w, int H, int W, int ignore, uint8_t* __restrict__ key, int32_t* __restrict__ hist,
                        int chunk, int nchunks) {
  __shared__ int lh[64];
  const int n = blockIdx.y, ck = blockIdx.x;
  const int HW = h * w;
  if (threadIdx.x < 64) lh[threadIdx.x] = 0;
  __syncthreads();
  const float sy = (float)H / (float)h, sx = (float)W / (float)w;
  for (int q = threadIdx.x; q < chunk; q += 256) {
    const int p = ck * chunk + q;
    if (p >= HW) break;
    const int y = p / w, x = p - y * w;
    const float* lp = logits + ((long long)n * HW + p) * cs;
    float best = lp[0];
    int bi = 0;
    for (int c = 1; c < C; ++c) { const float v = lp[c]; if (v > best) { best = v; bi = c; } }
    // F.interpolate(mode='nearest'): src = min(floor(dst * in/out), in - 1)   (utils/loss.py:400-403)
    int yy = (int)floorf((float)y * sy); if (yy > H - 1) yy = H - 1;
    int xx = (int)floorf((float)x * sx); if (xx > W - 1) xx = W - 1;
    const long long lab = labels[((long long)n * H + yy) * W + xx];
    int k = 255;
    if (lab != ignore && lab >= 0 && lab < C) {
      k = (int)lab * 2 + (bi == (int)lab ? 1 : 0);
      atomicAdd(&lh[k], 1);
    }
    key[(long long)n * HW + p] = (uint8_t)k;
  }
  __syncthreads();
  if (threadIdx.x < 2 * C) hist[((long long)n * nchunks + ck) * 2 * C + threadIdx.x] = lh[threadIdx.x];
}

__global__ __launch_bounds__(256)
void anchor_select_kernel(const uint8_t* __restrict__ key, const int32_t* __restrict__ hist,
                          const int32_t* __restrict__ req, int32_t* __restrict__ out, int HW, int C, int chunk,
                          int nchunks) {
  __shared__ int s_chunk, s_local;
  __shared__ int cnt[256];
  const int q = blockIdx.x;
  const int n = req[3 * q], k = req[3 * q + 1], rank = req[3 * q + 2];
  if (threadIdx.x == 0) {
    int cum = 0, ck = 0;
    s_chunk = -1;
    for (; ck < nchunks; ++ck) {
      const int c = hist[((long long)n * nchunks + ck) * 2 * C + k];
      if (rank < cum + c) { s_chunk = ck; s_local = rank - cum; break; }
      cum += c;
    }
  }
  __syncthreads();
  if (s_chunk < 0) { if (threadIdx.x == 0) out[q] = -1; return; }
  const int per = chunk / 256;
  const int base = s_chunk * chunk + threadIdx.x * per;
  int c = 0;
  for (int j = 0; j < per; ++j) { const int p = base + j; if (p < HW && key[(long long)n * HW + p] == k) ++c; }
  cnt[threadIdx.x] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    int cum = 0;
    for (int t = 0; t < 256; ++t) { const int v = cnt[t]; cnt[t] = cum; cum += v; }
  }
  __syncthreads();
  int seen = cnt[threadIdx.x];
  if (s_local >= seen && s_local < seen + c) {
    for (int j = 0; j < per; ++j) {
      const int p = base + j;
      if (p < HW && key[(long long)n * HW + p] == k) { if (seen == s_local) { out[q] = p; break; } ++seen; }
    }
  }
}

__global__ void gather_rows_kernel(const float* __restrict__ feat, const int32_t* __restrict__ rowidx,
                                   float* __restrict__ X, int A, int C) {
  const int C4 = C >> 2;
  const long long total = (long long)A * C4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int a = (int)(i / C4), c = (int)(i % C4) * 4;
    *reinterpret_cast<float4*>(X + (long long)a * C + c) =
        *reinterpret_cast<const float4*>(feat + (long long)rowidx[a] * C + c);
  }
}

__global__ void scatter_add_rows_kernel(const float* __restrict__ gX, const int32_t* __restrict__ rowidx,
                                        float* __restrict__ gfeat, int A, int C) {
  // rows are unique (a pixel is sampled at most once), so a plain read-modify-write is race free
  const int C4 = C >> 2;
  const long long total = (long long)A * C4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int a = (int)(i / C4), c = (int)(i % C4) * 4;
    float4* d = reinterpret_cast<float4*>(gfeat + (long long)rowidx[a] * C + c);
    const float4 v = *reinterpret_cast<const float4*>(gX + (long long)a * C + c);
    float4 o = *d;
    o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
    *d = o;
  }
}

// Rows of a bilinearly upsampled map WITHOUT the map (network/utils.py:190 upsamples the 2048-channel DeepLab feature
// to the logits' resolution only for utils/loss.py:391-415 to pick <= 608 pixels of it): row a = pixel rowidx[a] of the
// virtual [N,OH,OW] grid, interpolated from feat [N,IH,IW,C] with the arithmetic of upsample_add_kernel.
__global__ void gather_rows_bilinear_kernel(const float* __restrict__ feat, const int32_t* __restrict__ rowidx,
                                            float* __restrict__ X, int A, int C, int IH, int IW, int OH, int OW) {
  const int C4 = C >> 2;
  const long long total = (long long)A * C4;
  const float sy = (float)IH / (float)OH, sx = (float)IW / (float)OW;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int a = (int)(i / C4), c = (int)(i % C4) * 4;
    const int r = rowidx[a];
    const int ox = r % OW, oy = (r / OW) % OH, n = r / (OW * OH);
    const Lin ly = lin_src(oy, sy, IH), lx = lin_src(ox, sx, IW);
    const float* b = feat + (long long)n * IH * IW * C + c;
    const float4 v00 = *reinterpret_cast<const float4*>(b + ((long long)ly.i0 * IW + lx.i0) * C);
    const float4 v01 = *reinterpret_cast<const float4*>(b + ((long long)ly.i0 * IW + lx.i1) * C);
    const float4 v10 = *reinterpret_cast<const float4*>(b + ((long long)ly.i1 * IW + lx.i0) * C);
    const float4 v11 = *reinterpret_cast<const float4*>(b + ((long long)ly.i1 * IW + lx.i1) * C);
    float4 o;
#define DCS_BL(f) fmaf(ly.w1, fmaf(lx.w1, v11.f, lx.w0 * v10.f), ly.w0 * fmaf(lx.w1, v01.f, lx.w0 * v00.f))
    o.x = DCS_BL(x); o.y = DCS_BL(y); o.z = DCS_BL(z); o.w = DCS_BL(w);
#undef DCS_BL
    *reinterpret_cast<float4*>(X + (long long)a * C + c) = o;
  }
}

// Adjoint: gfeat [N,IH,IW,C] += the four weighted taps of every row.  Different rows share taps, so one thread owns
// (image n, channel group c) and walks the rows in order: deterministic, no atomics.
__global__ void scatter_rows_bilinear_kernel(const float* __restrict__ gX, const int32_t* __restrict__ rowidx,
                                             float* __restrict__ gfeat, int A, int C, int IH, int IW, int OH, int OW) {
  const int C4 = C >> 2;
  const int c4 = blockIdx.x * blockDim.x + threadIdx.x;
  if (c4 >= C4) return;
  const int n = blockIdx.y, c = c4 * 4;
  const float sy = (float)IH / (float)OH, sx = (float)IW / (float)OW;
  float* b = gfeat + (long long)n * IH * IW * C + c;
  for (int a = 0; a < A; ++a) {
    const int r = rowidx[a];
    if (r / (OW * OH) != n) continue;
    const int ox = r % OW, oy = (r / OW) % OH;
    const Lin ly = lin_src(oy, sy, IH), lx = lin_src(ox, sx, IW);
    const float4 g = *reinterpret_cast<const float4*>(gX + (long long)a * C + c);
    const int ys[2] = {ly.i0, ly.i1}, xs[2] = {lx.i0, lx.i1};
    const float wy[2] = {ly.w0, ly.w1}, wx[2] = {lx.w0, lx.w1};
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float wgt = wy[j] * wx[k];
        float4* d = reinterpret_cast<float4*>(b + ((long long)ys[j] * IW + xs[k]) * C);
        float4 o = *d;
        o.x = fmaf(wgt, g.x, o.x); o.y = fmaf(wgt, g.y, o.y); o.z = fmaf(wgt, g.z, o.z); o.w = fmaf(wgt, g.w, o.w);
        *d = o;
      }
  }
}

// ---------------------------------------------------------------------------------------------------
// contrastive rows.  One 256-thread block per anchor row i.
__device__ __forceinline__ float block_sum(float v, float* sm) {
  v = dcs_wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  return sm[0] + sm[1] + sm[2] + sm[3];
}
__device__ __forceinline__ float block_max(float v, float* sm) {
  v = dcs_wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
}

__global__ __launch_bounds__(256)
void contrast_rows_kernel(const float* __restrict__ S, const float* __restrict__ labels, float* __restrict__ loss_row,
                          float* __restrict__ G, int A, int ld, int mode, float inv_temp, float inv_rows) {
  __shared__ float sm[4];
  const int i = blockIdx.x;
  const float* s = S + (long long)i * ld;
  float* gr = G + (long long)i * ld;
  const float yi = labels[i];
  // pass 1: row max of S/T (utils/loss.py:363, :179)
  float mx = -INFINITY;
  for (int j = threadIdx.x; j < A; j += 256) mx = fmaxf(mx, s[j] * inv_temp);
  mx = block_max(mx, sm);
  // pass 2: ||S_i - max||_2, F.normalize eps 1e-12 (:366, :194)
  float n2 = 0.f;
  for (int j = threadIdx.x; j < A; j += 256) { const float u = s[j] * inv_temp - mx; n2 = fmaf(u, u, n2); }
  n2 = block_sum(n2, sm);
  const float nraw = sqrtf(n2);
  const float nrm = fmaxf(nraw, 1e-12f);
  const float rn = 1.f / nrm;
  // pass 3: denominators / positive counts
  float den = 0.f, cnt = 0.f;
  for (int j = threadIdx.x; j < A; j += 256) {
    const float L = (s[j] * inv_temp - mx) * rn;
    const bool same = labels[j] == yi;
    if (mode == 0) { if (!same) den += expf(L); }          // neg_logits (:376-377)
    else { if (j != i) den += expf(L); }                   // exp_logits * logits_mask (:196)
    if (same && j != i) cnt += 1.f;
  }
  den = block_sum(den, sm);
  cnt = block_sum(cnt, sm);
  // pass 4: sum over positives of log_prob, and q = sum_pos 1/(E+neg) (pixel mode)
  float lp = 0.f, qv = 0.f;
  for (int j = threadIdx.x; j < A; j += 256) {
    if (j == i || labels[j] != yi) continue;
    const float L = (s[j] * inv_temp - mx) * rn;
    if (mode == 0) { const float d = expf(L) + den; lp += L - logf(d); qv += 1.f / d; }
    else lp += L - logf(den);
  }
  lp = block_sum(lp, sm);
  qv = block_sum(qv, sm);
  if (threadIdx.x == 0) loss_row[i] = -lp / cnt;              // temperature/base_temperature = 1
  // pass 5: dL_j, then through F.normalize: du = (dL - L * <dL, L>) / nrm
  float dot = 0.f;
  for (int j = threadIdx.x; j < A; j += 256) {
    const float L = (s[j] * inv_temp - mx) * rn;
    const float E = expf(L);
    const bool same = labels[j] == yi;
    float dL;
    if (mode == 0) dL = same ? (j != i ? -(den / (E + den)) / cnt : 0.f) : E * qv / cnt;
    else dL = (j != i ? E / den : 0.f) - ((same && j != i) ? 1.f / cnt : 0.f);
    dot = fmaf(dL, L, dot);
  }
  dot = block_sum(dot, sm);
  const bool clamp = nraw <= 1e-12f;
  for (int j = threadIdx.x; j < ld; j += 256) {
    float o = 0.f;
    if (j < A) {
      const float L = (s[j] * inv_temp - mx) * rn;
      const float E = expf(L);
      const bool same = labels[j] == yi;
      float dL;
      if (mode == 0) dL = same ? (j != i ? -(den / (E + den)) / cnt : 0.f) : E * qv / cnt;
      else dL = (j != i ? E / den : 0.f) - ((same && j != i) ? 1.f / cnt : 0.f);
      o = (clamp ? dL : (dL - L * dot)) * rn * inv_rows * inv_temp;
    }
    gr[j] = o;
  }
}

__global__ void symmetrize_kernel(const float* __restrict__ G, float* __restrict__ Gs, int A, int ld) {
  const long long total = (long long)A * ld;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / ld), c = (int)(i % ld);
    Gs[i] = c < A ? G[i] + G[(long long)c * ld + r] : 0.f;
  }
}

// ---------------------------------------------------------------------------------------------------
// validation: per-pixel class id (argmax of the logits, optionally of the bilinearly upsampled low-resolution
// logits so the full-resolution tensor is never materialised) + confusion matrix, fused
// (trainer.py:349 + metrics/stream_metrics.py:330-342).  Integer LDS histogram per block, integer global atomics.
struct LinC { int i0, i1; float w0, w1; };
__device__ __forceinline__ LinC linc(int o, float scale, int in) {
  float s = scale * ((float)o + 0.5f) - 0.5f;
  if (s < 0.f) s = 0.f;
  LinC l;
  l.i0 = (int)s;
  if (l.i0 > in - 1) l.i0 = in - 1;
  l.i1 = l.i0 + (l.i0 < in - 1 ? 1 : 0);
  l.w1 = s - (float)l.i0;
  l.w0 = 1.f - l.w1;
  return l;
}

template <bool LOWRES>
__global__ __launch_bounds__(256)
void confusion_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                      unsigned char* __restrict__ pred_out, unsigned long long* __restrict__ conf, int C, int H, int W,
                      int ih, int iw, int cs) {
  extern __shared__ int hist[];                       // [C*C]
  const int n = blockIdx.y;
  const long long HW = (long long)H * W;
  for (int t = threadIdx.x; t < C * C; t += 256) hist[t] = 0;
  __syncthreads();
  const float sy = (float)ih / (float)H, sx = (float)iw / (float)W;
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < HW; p += (long long)gridDim.x * 256) {
    int best = 0;
    float bv = -INFINITY;
    if (!LOWRES) {
      const float* lp = logits + (long long)n * C * HW + p;
      for (int c = 0; c < C; ++c) { const float v = lp[(long long)c * HW]; if (v > bv) { bv = v; best = c; } }
    } else {
      const int oy = (int)(p / W), ox = (int)(p - (long long)oy * W);
      const LinC ly = linc(oy, sy, ih), lx = linc(ox, sx, iw);
      const float* b = logits + (long long)n * ih * iw * cs;
      const float* p00 = b + ((long long)ly.i0 * iw + lx.i0) * cs;
      const float* p01 = b + ((long long)ly.i0 * iw + lx.i1) * cs;
      const float* p10 = b + ((long long)ly.i1 * iw + lx.i0) * cs;
      const float* p11 = b + ((long long)ly.i1 * iw + lx.i1) * cs;
      for (int c = 0; c < C; ++c) {
        const float top = fmaf(lx.w1, p01[c], lx.w0 * p00[c]);
        const float bot = fmaf(lx.w1, p11[c], lx.w0 * p10[c]);
        const float v = fmaf(ly.w1, bot, ly.w0 * top);           // same arithmetic as upsample_to_nchw_kernel
        if (v > bv) { bv = v; best = c; }
      }
    }
    if (pred_out) pred_out[(long long)n * HW + p] = (unsigned char)best;
    const long long gt = labels[(long long)n * HW + p];
    if (gt >= 0 && gt < C) atomicAdd(&hist[(int)gt * C + best], 1);
  }
  __syncthreads();
  for (int t = threadIdx.x; t < C * C; t += 256)
    if (hist[t]) atomicAdd(&conf[(long long)n * C * C + t], (unsigned long long)hist[t]);
}

inline unsigned grid_for(long long n, unsigned cap = 8192) {
  long long b = (n + 255) / 256;
  if (b < 1) b = 1;
  return (unsigned)(b > cap ? cap : b);
}

}  // namespace

extern "C" int dcs_seg_loss(const float* logits, int64_t* target, const float* ldw, const float* cw, float* grad,
                            float* partial, int N, int C, int H, int W, int mode, float gamma, int ignore, int blocks,
                            void* stream) {
  DCS_CHECK_ARG(logits && target && grad && partial && N > 0 && C > 0 && C <= 32 && H > 0 && W > 0 && blocks > 0);
  DCS_CHECK_ARG(mode >= 0 && mode <= 4 && (mode == 4 || mode == 1 || ldw));
  DCS_CHECK_ARG(mode == 4 || ldw);
  const long long HW = (long long)H * W;
  if (C <= 20)
    hipLaunchKernelGGL(seg_loss_kernel<20>, dim3((unsigned)blocks), dim3(256), 0, dcs_stream(stream), logits, target, ldw, cw,
                       grad, partial, N, C, HW, mode, gamma, ignore);
  else
    hipLaunchKernelGGL(seg_loss_kernel<32>, dim3((unsigned)blocks), dim3(256), 0, dcs_stream(stream), logits, target, ldw, cw,
                       grad, partial, N, C, HW, mode, gamma, ignore);
  DCS_LAUNCH_RET();
}

extern "C" int dcs_seg_loss_fused(const float* logits_lr, int cs, int64_t* target, const float* ldw, const float* cw,
                                  float* grad_lr, float* partial, int N, int C, int ih, int iw, int H, int W, int mode,
                                  float gamma, int ignore, int blocks, void* stream) {
  DCS_CHECK_ARG(logits_lr && target && grad_lr && partial && N > 0 && C > 0 && C <= 20 && cs >= C && ih > 0 && iw > 0);
  DCS_CHECK_ARG(mode >= 0 && mode <= 4 && (mode == 4 || ldw));
  if (H <= 0 || W <= 0 || H % ih != 0 || W % iw != 0 || H / ih != W / iw) return DCS_E_UNSUPPORTED;
  const int F = H / ih;
  const int tiles_x = (iw + SLC_PW - 1) / SLC_PW, tiles_y = (ih + SLC_PH - 1) / SLC_PH;
  DCS_CHECK_ARG((long long)N * tiles_x * tiles_y == blocks);
  hipStream_t s = dcs_stream(stream);
#define DCS_SLC(F_, MC_)                                                                                                  \
  hipLaunchKernelGGL((seg_loss_cells_kernel<F_, MC_>), dim3((unsigned)blocks), dim3(SLC_CH * SLC_CW), 0, s, logits_lr, cs, \
                     target, ldw, cw, grad_lr, partial, C, ih, iw, mode, gamma, ignore, tiles_x, tiles_y)
  if (F == 4) { if (C <= 19) DCS_SLC(4, 19); else DCS_SLC(4, 20); }
  else if (F == 2) { if (C <= 19) DCS_SLC(2, 19); else DCS_SLC(2, 20); }
  else return DCS_E_UNSUPPORTED;
#undef DCS_SLC
  DCS_LAUNCH_RET();
}

extern "C" int dcs_seg_loss_fused_blocks(int N, int ih, int iw) {
  if (N <= 0 || ih <= 0 || iw <= 0) return DCS_E_ARG;
  const long long nb = (long long)N * ((iw + SLC_PW - 1) / SLC_PW) * ((ih + SLC_PH - 1) / SLC_PH);
  return nb < (1ll << 30) ? (int)nb : DCS_E_UNSUPPORTED;
}

extern "C" int dcs_seg_loss_final(const float* partial, float* out, int blocks, void* stream) {
  DCS_CHECK_ARG(partial && out && blocks > 0);
  hipLaunchKernelGGL(seg_loss_final_kernel, dim3(1), dim3(256), 0, dcs_stream(stream), partial, out, blocks);
  DCS_LAUNCH_RET();
}

extern "C" int dcs_anchor_keys(const float* logits, int cs, int C, const int64_t* labels, int N, int h, int w, int H,
                               int W, int ignore, uint8_t* key, int32_t* hist, int chunk, void* stream) {
  DCS_CHECK_ARG(logits && labels && key && hist && N > 0 && h > 0 && w > 0 && H > 0 && W > 0);
  DCS_CHECK_ARG(C > 0 && C <= 32 && cs >= C && chunk > 0 && (chunk & 255) == 0);
  const int nchunks = (h * w + chunk - 1) / chunk;
  hipLaunchKernelGGL(anchor_keys_kernel, dim3((unsigned)nchunks, (unsigned)N), dim3(256), 0, dcs_stream(stream), logits, cs, C,
                     labels, h, w, H, W, ignore, key, hist, chunk, nchunks);
  DCS_LAUNCH_RET();
}

extern "C" int dcs_anchor_select(const uint8_t* key, const int32_t* hist, const int32_t* req, int32_t* out, int Q, int N,
                                 int HW, int C, int chunk, void* stream) {
  DCS_CHECK_ARG(key && hist && req && out && Q > 0 && N > 0 && HW > 0 && C > 0 && C <= 32 && chunk > 0 && (chunk & 255) == 0);
  const int nchunks = (HW + chunk - 1) / chunk;
  hipLaunchKernelGGL(anchor_select_kernel, dim3((unsigned)Q), dim3(256), 0, dcs_stream(stream), key, hist, req, out, HW, C,
                     chunk, nchunks);
  DCS_LAUNCH_RET();
}

extern "C" int dcs_gather_rows(const float* feat, const int32_t* rowidx, float* X, int A, int C, void* stream) {
  DCS_CHECK_ARG(feat && rowidx && X && A > 0 && C > 0 && (C & 3) == 0);
  hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for((long long)A * C / 4)), dim3(256), 0, dcs_stream(stream), feat, rowidx, X, A, C);
  DCS_LAUNCH_RET();
}

extern "C" int dcs_scatter_add_rows(const float* gX, const int32_t* rowidx, float* gfeat, int A, int C, void* stream) {
  DCS_CHECK_ARG(gX && rowidx && gfeat && A > 0 && C > 0 && (C & 3) == 0);
  hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(grid_for((long long)A * C / 4)), dim3(256), 0, dcs_stream(stream), gX, rowidx,
                     gfeat, A, C);
  DCS_LAUNCH_RET();
}

extern "C" int dcs_gather_rows_bilinear(const float* feat, const int32_t* rowidx, float* X, int A, int C, int N, int IH,
                                        int IW, int OH, int OW, void* stream) {
  DCS_CHECK_ARG(feat && rowidx && X && A > 0 && C > 0 && (C & 3) == 0 && N > 0 && IH > 0 && IW > 0 && OH > 0 && OW > 0);
  DCS_CHECK_ARG((long long)N * OH * OW < (1ll << 31));
  hipLaunchKernelGGL(gather_rows_bilinear_kernel, dim3(grid_for((long long)A * C / 4)), dim3(256), 0, dcs_stream(stream),
                     feat, rowidx, X, A, C, IH, IW, OH, OW);
  DCS_LAUNCH_RET();
}

extern "C" int dcs_scatter_rows_bilinear(const float* gX, const int32_t* rowidx, float* gfeat, int A, int C, int N, int IH,
                                         int IW, int OH, int OW, void* stream) {
  DCS_CHECK_ARG(gX && rowidx && gfeat && A > 0 && C > 0 && (C & 3) == 0 && N > 0 && N <= 65535);
  DCS_CHECK_ARG(IH > 0 && IW > 0 && OH > 0 && OW > 0 && (long long)N * OH * OW < (1ll << 31));
  hipLaunchKernelGGL(scatter_rows_bilinear_kernel, dim3((unsigned)((C / 4 + 63) / 64), (unsigned)N), dim3(64), 0,
                     dcs_stream(stream), gX, rowidx, gfeat, A, C, IH, IW, OH, OW);
  DCS_LAUNCH_RET();
}

extern "C" int dcs_contrast_rows(const float* S, const float* labels, float* loss_row, float* G, int A, int ld, int mode,
                                 float inv_temp, void* stream) {
  DCS_CHECK_ARG(S && labels && loss_row && G && A > 0 && ld >= A && (mode == 0 || mode == 1));
  hipLaunchKernelGGL(contrast_rows_kernel, dim3((unsigned)A), dim3(256), 0, dcs_stream(stream), S, labels, loss_row, G, A, ld,
                     mode, inv_temp, 1.f / (float)A);
  DCS_LAUNCH_RET();
}

extern "C" int dcs_symmetrize(const float* G, float* Gs, int A, int ld, void* stream) {
  DCS_CHECK_ARG(G && Gs && A > 0 && ld >= A);
  hipLaunchKernelGGL(symmetrize_kernel, dim3(grid_for((long long)A * ld)), dim3(256), 0, dcs_stream(stream), G, Gs, A, ld);
  DCS_LAUNCH_RET();
}

extern "C" int dcs_confusion(const float* logits, const int64_t* labels, uint8_t* pred_out, uint64_t* conf, int N, int C,
                             int H, int W, int lowres_h, int lowres_w, int cs, void* stream) {
  DCS_CHECK_ARG(logits && labels && conf && N > 0 && C > 0 && C <= 64 && H > 0 && W > 0);
  DCS_CHECK_ARG((lowres_h == 0 && lowres_w == 0) || (lowres_h > 0 && lowres_w > 0 && cs >= C));
  const long long HW = (long long)H * W;
  long long bx = (HW + 255) / 256;
  if (bx > 2048) bx = 2048;
  dim3 grid((unsigned)bx, (unsigned)N);
  const size_t sh = (size_t)C * C * sizeof(int);
  if (lowres_h == 0)
    hipLaunchKernelGGL(confusion_kernel<false>, grid, dim3(256), sh, dcs_stream(stream), logits, labels, pred_out,
                       (unsigned long long*)conf, C, H, W, 0, 0, 0);
  else
    hipLaunchKernelGGL(confusion_kernel<true>, grid, dim3(256), sh, dcs_stream(stream), logits, labels, pred_out,
                       (unsigned long long*)conf, C, H, W, lowres_h, lowres_w, cs);
  DCS_LAUNCH_RET();
}
