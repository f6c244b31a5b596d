// Image pyramid, stem max-pool and bilinear resizes (forward + adjoint), NHWC fp32.
// All HBM-bound; backward passes are written as gathers (no float atomics -> deterministic).
// Replaces F.interpolate / nn.MaxPool2d calls of network/backbone/resnet_pyramid.py:296-325 and
// network/utils.py:8,:92-102 in the reference.
#include "dcs_common.h"

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 f4axpy(float a, float4 x, float4 y) {
  return make_float4(fmaf(a, x.x, y.x), fmaf(a, x.y, y.y), fmaf(a, x.z, y.z), fmaf(a, x.w, y.w));
}
__device__ __forceinline__ float4 f4scale(float a, float4 x) { return make_float4(a * x.x, a * x.y, a * x.z, a * x.w); }
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// ---- level 0: (x - mean) / std, NCHW -> NHWC4 -----------------------------------------------------
__global__ void normalize_kernel(const float* __restrict__ img, float* __restrict__ out, int N, long long HW,
                                 const float* __restrict__ mean3, const float* __restrict__ std3) {
  const long long total = (long long)N * HW;
  const float m0 = mean3[0], m1 = mean3[1], m2 = mean3[2], s0 = std3[0], s1 = std3[1], s2 = std3[2];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long n = i / HW, p = i - n * HW;
    const float* b = img + n * 3 * HW + p;
    st4(out + i * 4, make_float4((b[0] - m0) / s0, (b[HW] - m1) / s1, (b[2 * HW] - m2) / s2, 0.f));
  }
}

// ---- levels 1,2: bicubic (A=-0.75) at exact scale 1/f, f in {2,4}: fractional offset is always 0.5,
// taps f*d + f/2 - 2 .. +1 with weights (-3/32, 19/32, 19/32, -3/32), indices clamped (resnet_pyramid.py:313)
__global__ void bicubic_down_kernel(const float* __restrict__ x0, float* __restrict__ out, int N, int H, int W,
                                    int OH, int OW, int f) {
  const long long total = (long long)N * OH * OW;
  const float wt[4] = {-0.09375f, 0.59375f, 0.59375f, -0.09375f};
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int ox = (int)(i % OW);
    const long long q = i / OW;
    const int oy = (int)(q % OH), n = (int)(q / OH);
    const int by = f * oy + f / 2 - 2, bx = f * ox + f / 2 - 2;
    float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int iy = min(max(by + a, 0), H - 1);
      float4 row = make_float4(0, 0, 0, 0);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int ix = min(max(bx + b, 0), W - 1);
        row = f4axpy(wt[b], ld4(x0 + (((long long)n * H + iy) * W + ix) * 4), row);
      }
      acc = f4axpy(wt[a], row, acc);
    }
    acc.w = 0.f;
    st4(out + i * 4, acc);
  }
}

// ---- stem: maxpool 3x3/2 pad 1 over relu(y*scale+shift), argmax kept as 0..8 -------------------------
__global__ __launch_bounds__(256)
void bn_relu_maxpool_kernel(const float* __restrict__ y, const float* __restrict__ bn, float* __restrict__ out,
                            uint8_t* __restrict__ idx, int N, int H, int W, int OH, int OW, int C, int nt) {
  const int C4 = C >> 2;
  const long long total = (long long)N * OH * OW * C4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    long long q = i / C4;
    const int ox = (int)(q % OW); q /= OW;
    const int oy = (int)(q % OH);
    const int n = (int)(q / OH);
    const float4 sc = ld4(bn + c), sh = ld4(bn + C + c);
    float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    int b0 = 0, b1 = 0, b2 = 0, b3 = 0;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = 2 * oy - 1 + ky;
      if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = 2 * ox - 1 + kx;
        if ((unsigned)ix >= (unsigned)W) continue;
        const float4 v = ld4(y + (((long long)n * H + iy) * W + ix) * C + c);
        const float z0 = fmaxf(fmaf(v.x, sc.x, sh.x), 0.f), z1 = fmaxf(fmaf(v.y, sc.y, sh.y), 0.f);
        const float z2 = fmaxf(fmaf(v.z, sc.z, sh.z), 0.f), z3 = fmaxf(fmaf(v.w, sc.w, sh.w), 0.f);
        const int k = ky * 3 + kx;
        if (z0 > best.x) { best.x = z0; b0 = k; }
        if (z1 > best.y) { best.y = z1; b1 = k; }
        if (z2 > best.z) { best.z = z2; b2 = k; }
        if (z3 > best.w) { best.w = z3; b3 = k; }
      }
    }
    st4s(out + i * 4, best, nt);
    *reinterpret_cast<uchar4*>(idx + i * 4) = make_uchar4((unsigned char)b0, (unsigned char)b1, (unsigned char)b2, (unsigned char)b3);
  }
}

__global__ __launch_bounds__(256)
void maxpool_bwd_kernel(const float* __restrict__ g, const uint8_t* __restrict__ idx, float* __restrict__ gz, int N,
                        int H, int W, int OH, int OW, int C) {
  const int C4 = C >> 2;
  const long long total = (long long)N * H * W * C4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    long long q = i / C4;
    const int ix = (int)(q % W); q /= W;
    const int iy = (int)(q % H);
    const int n = (int)(q / H);
    float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int ty = iy + 1 - ky;
      if (ty < 0 || (ty & 1)) continue;
      const int oy = ty >> 1;
      if (oy >= OH) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int tx = ix + 1 - kx;
        if (tx < 0 || (tx & 1)) continue;
        const int ox = tx >> 1;
        if (ox >= OW) continue;
        const long long o = (((long long)n * OH + oy) * OW + ox) * C + c;
        const uchar4 id = *reinterpret_cast<const uchar4*>(idx + o);
        const float4 gv = ld4(g + o);
        const int k = ky * 3 + kx;
        if (id.x == k) acc.x += gv.x;
        if (id.y == k) acc.y += gv.y;
        if (id.z == k) acc.z += gv.z;
        if (id.w == k) acc.w += gv.w;
      }
    }
    st4(gz + i * 4, acc);
  }
}

// ---- stem backward without the pooled-gradient tensor ------------------------------------------------
// d(loss)/d(relu output) of the 2x2 pixel quad (2qy..2qy+1, 2qx..2qx+1) from the pooled gradient: the quad is touched
// by exactly the four windows (qy..qy+1, qx..qx+1), so one thread gathers 4 x (float4 + uchar4) for 4 pixels.
// Window slot k = ky*3+kx of the forward kernel; additions in the same (ky, kx) order as maxpool_bwd_kernel.
__device__ __forceinline__ void pool_quad_grad(const float* __restrict__ g, const uint8_t* __restrict__ idx, int n, int qy,
                                               int qx, int c, int OH, int OW, int C, float4 out[4]) {
  float4 gv[4];
  uchar4 id[4];
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const int oy = qy + (w >> 1), ox = qx + (w & 1);
    if (oy < OH && ox < OW) {
      const long long o = (((long long)n * OH + oy) * OW + ox) * C + c;
      gv[w] = ld4(g + o);
      id[w] = *reinterpret_cast<const uchar4*>(idx + o);
    } else {
      gv[w] = make_float4(0.f, 0.f, 0.f, 0.f);
      id[w] = make_uchar4(255, 255, 255, 255);
    }
  }
#define DCS_PICK(WI, K) make_float4(id[WI].x == (K) ? gv[WI].x : 0.f, id[WI].y == (K) ? gv[WI].y : 0.f,                \
                                    id[WI].z == (K) ? gv[WI].z : 0.f, id[WI].w == (K) ? gv[WI].w : 0.f)
  out[0] = DCS_PICK(0, 4);
  out[1] = f4add(DCS_PICK(1, 3), DCS_PICK(0, 5));
  out[2] = f4add(DCS_PICK(2, 1), DCS_PICK(0, 7));
  out[3] = f4add(f4add(f4add(DCS_PICK(3, 0), DCS_PICK(2, 2)), DCS_PICK(1, 6)), DCS_PICK(0, 8));
#undef DCS_PICK
}

// partial[group][2][C]: sum of the masked gradient and of gradient * xhat over the group's quads
__global__ __launch_bounds__(256)
void bn_pool_bwd_partial_kernel(const float* __restrict__ g, const uint8_t* __restrict__ idx, const float* __restrict__ y,
                                const float* __restrict__ bn, float* __restrict__ partial, int N, int H, int W, int OH,
                                int OW, int C, int groups, int nt) {
  extern __shared__ __attribute__((aligned(16))) double smd[];   // [2][RL][C]; double: see colsum_partial_kernel
  const int C4 = C >> 2, RL = 256 / C4;
  const int tid = threadIdx.x, col4 = tid % C4, rl = tid / C4;
  const int c = col4 * 4;
  const int QH = (H + 1) >> 1, QW = (W + 1) >> 1;
  const long long Q = (long long)N * QH * QW;
  const long long qpg = (Q + groups - 1) / groups;
  const long long qbeg = (long long)blockIdx.x * qpg;
  const long long qend = qbeg + qpg < Q ? qbeg + qpg : Q;
  const float4 sc = ld4(bn + c), sh = ld4(bn + C + c), mu = ld4(bn + 2 * C + c), is = ld4(bn + 3 * C + c);
  double a0[4] = {0.0, 0.0, 0.0, 0.0}, a1[4] = {0.0, 0.0, 0.0, 0.0};
  for (long long q = qbeg + rl; q < qend; q += RL) {
    const int qx = (int)(q % QW);
    const long long t = q / QW;
    const int qy = (int)(t % QH), n = (int)(t / QH);
    float4 gq[4];
    pool_quad_grad(g, idx, n, qy, qx, c, OH, OW, C, gq);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int iy = 2 * qy + (p >> 1), ix = 2 * qx + (p & 1);
      if (iy >= H || ix >= W) continue;
      const float4 yy = ld4s(y + (((long long)n * H + iy) * W + ix) * C + c, nt);
      float4 v = gq[p];
      v.x = fmaf(yy.x, sc.x, sh.x) > 0.f ? v.x : 0.f; v.y = fmaf(yy.y, sc.y, sh.y) > 0.f ? v.y : 0.f;
      v.z = fmaf(yy.z, sc.z, sh.z) > 0.f ? v.z : 0.f; v.w = fmaf(yy.w, sc.w, sh.w) > 0.f ? v.w : 0.f;
      const float xh[4] = {(yy.x - mu.x) * is.x, (yy.y - mu.y) * is.y, (yy.z - mu.z) * is.z, (yy.w - mu.w) * is.w};
      const double d[4] = {(double)v.x, (double)v.y, (double)v.z, (double)v.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) { a0[e] += d[e]; a1[e] = fma(d[e], (double)xh[e], a1[e]); }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    smd[(0 * RL + rl) * C + c + e] = a0[e];
    smd[(1 * RL + rl) * C + c + e] = a1[e];
  }
  __syncthreads();
  for (int t = tid; t < 2 * C; t += 256) {
    const int which = t / C, cc = t - which * C;
    double s = 0.0;
    for (int k = 0; k < RL; ++k) s += smd[(which * RL + k) * C + cc];
    partial[((long long)blockIdx.x * 2 + which) * C + cc] = (float)s;
  }
}

// The same two sums from the POOLED tensors alone (round 3).  Every pooled gradient lands on exactly one un-pooled position
// (its window's argmax), the ReLU there is open exactly when the pooled output z is positive, and there z = scale * y +
// shift, so y -- and with it xhat = (y - mean) * invstd -- follows from z: sum(gm) = sum_p [z_p > 0] g_p, sum(gm * xhat) =
// sum_p [z_p > 0] g_p * (((z_p - shift) / scale - mean) * invstd).  Reads g and z (1/4 of the un-pooled map each) instead of
// g, idx and y: 5.6 -> 2.1 GB at C3's finest level.  Channels whose |gamma| = |scale / invstd| is below 0.05 would amplify
// the rounding of z by 1 / gamma: those fetch y at the argmax position (idx) like the un-pooled kernel.
__global__ __launch_bounds__(256)
void bn_pool_bwd_partial_pooled_kernel(const float* __restrict__ g, const float* __restrict__ z, const uint8_t* __restrict__ idx,
                                       const float* __restrict__ y, const float* __restrict__ bn, float* __restrict__ partial,
                                       int N, int H, int W, int OH, int OW, int C, int groups) {
  extern __shared__ __attribute__((aligned(16))) double smd[];   // [2][RL][C]
  const int C4 = C >> 2, RL = 256 / C4;
  const int tid = threadIdx.x, col4 = tid % C4, rl = tid / C4;
  const int c = col4 * 4;
  const long long P = (long long)N * OH * OW;
  const long long ppg = (P + groups - 1) / groups;
  const long long pbeg = (long long)blockIdx.x * ppg;
  const long long pend = pbeg + ppg < P ? pbeg + ppg : P;
  const float4 sc4 = ld4(bn + c), sh4 = ld4(bn + C + c), mu4 = ld4(bn + 2 * C + c), is4 = ld4(bn + 3 * C + c);
  const float sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, sh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
  const float mu[4] = {mu4.x, mu4.y, mu4.z, mu4.w}, is[4] = {is4.x, is4.y, is4.z, is4.w};
  float rsc[4];
  bool viaz[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { viaz[e] = fabsf(sc[e]) >= 0.05f * fabsf(is[e]); rsc[e] = viaz[e] ? 1.f / sc[e] : 0.f; }
  double a0[4] = {0.0, 0.0, 0.0, 0.0}, a1[4] = {0.0, 0.0, 0.0, 0.0};
  for (long long q = pbeg + rl; q < pend; q += RL) {
    const long long o = q * C + c;
    const float4 gv4 = ld4(g + o), zv4 = ld4(z + o);
    const float gv[4] = {gv4.x, gv4.y, gv4.z, gv4.w}, zv[4] = {zv4.x, zv4.y, zv4.z, zv4.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (!(zv[e] > 0.f)) continue;
      float yy;
      if (viaz[e]) {
        yy = (zv[e] - sh[e]) * rsc[e];
      } else {                                           // (rare) the argmax position of this window
        const int ox = (int)(q % OW);
        const long long t = q / OW;
        const int oy = (int)(t % OH), n = (int)(t / OH);
        const int k = idx[o + e];
        const int iy = 2 * oy - 1 + k / 3, ix = 2 * ox - 1 + k % 3;
        yy = y[(((long long)n * H + iy) * W + ix) * C + c + e];
      }
      const float xh = (yy - mu[e]) * is[e];
      a0[e] += (double)gv[e];
      a1[e] = fma((double)gv[e], (double)xh, a1[e]);
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    smd[(0 * RL + rl) * C + c + e] = a0[e];
    smd[(1 * RL + rl) * C + c + e] = a1[e];
  }
  __syncthreads();
  for (int t = tid; t < 2 * C; t += 256) {
    const int which = t / C, cc = t - which * C;
    double s = 0.0;
    for (int k = 0; k < RL; ++k) s += smd[(which * RL + k) * C + cc];
    partial[((long long)blockIdx.x * 2 + which) * C + cc] = (float)s;
  }
}

__global__ __launch_bounds__(256)
void bn_pool_bwd_apply_kernel(const float* __restrict__ g, const uint8_t* __restrict__ idx, const float* __restrict__ y,
                              const float* __restrict__ bn, const float* __restrict__ gamma, const float* __restrict__ sums,
                              float* __restrict__ dy, float* __restrict__ dgamma, float* __restrict__ dbeta, int N, int H,
                              int W, int OH, int OW, int C, int acc_param, int training, int nt,
                              unsigned* __restrict__ dy_maxabs) {
  const int C4 = C >> 2;
  const int QH = (H + 1) >> 1, QW = (W + 1) >> 1;
  const long long total = (long long)N * QH * QW * C4;
  float mx = 0.f;
  const float inv = training ? (float)(1.0 / ((double)N * H * W)) : 0.f;
  if (blockIdx.x == 0 && dgamma) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      dbeta[c] = (acc_param ? dbeta[c] : 0.f) + sums[c];
      dgamma[c] = (acc_param ? dgamma[c] : 0.f) + sums[C + c];
    }
  }
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    long long q = i / C4;
    const int qx = (int)(q % QW); q /= QW;
    const int qy = (int)(q % QH);
    const int n = (int)(q / QH);
    const float4 sc = ld4(bn + c), sh = ld4(bn + C + c), mu = ld4(bn + 2 * C + c), is = ld4(bn + 3 * C + c);
    const float4 gw = ld4(gamma + c), s0 = ld4(sums + c), s1 = ld4(sums + C + c);
    float4 gq[4];
    pool_quad_grad(g, idx, n, qy, qx, c, OH, OW, C, gq);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int iy = 2 * qy + (p >> 1), ix = 2 * qx + (p & 1);
      if (iy >= H || ix >= W) continue;
      const long long off = (((long long)n * H + iy) * W + ix) * C + c;
      const float4 yy = ld4s(y + off, nt);
      float4 v = gq[p];
      v.x = fmaf(yy.x, sc.x, sh.x) > 0.f ? v.x : 0.f; v.y = fmaf(yy.y, sc.y, sh.y) > 0.f ? v.y : 0.f;
      v.z = fmaf(yy.z, sc.z, sh.z) > 0.f ? v.z : 0.f; v.w = fmaf(yy.w, sc.w, sh.w) > 0.f ? v.w : 0.f;
      float4 o;
      o.x = gw.x * is.x * (v.x - s0.x * inv - (yy.x - mu.x) * is.x * (s1.x * inv));
      o.y = gw.y * is.y * (v.y - s0.y * inv - (yy.y - mu.y) * is.y * (s1.y * inv));
      o.z = gw.z * is.z * (v.z - s0.z * inv - (yy.z - mu.z) * is.z * (s1.z * inv));
      o.w = gw.w * is.w * (v.w - s0.w * inv - (yy.w - mu.w) * is.w * (s1.w * inv));
      st4s(dy + off, o, nt);
      mx = fmaxf(fmaxf(mx, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
    }
  }
  if (dy_maxabs) {                       // max |dy| as float bits, one filtered atomic per block (see bn_bwd_apply_kernel)
    __shared__ float s_mx[4];
    mx = dcs_wave_max(mx);
    if ((threadIdx.x & 63) == 0) s_mx[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
      mx = fmaxf(fmaxf(s_mx[0], s_mx[1]), fmaxf(s_mx[2], s_mx[3]));
      const unsigned bits = __float_as_uint(mx);
      if (bits > __hip_atomic_load(dy_maxabs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(dy_maxabs, bits);
    }
  }
}

// ---- bilinear, align_corners=False, explicit output size (torch area_pixel_compute_source_index) -----
// Lin / lin_src: dcs_common.h
// total weight with which input index i enters output o
__device__ __forceinline__ float lin_w(int o, float scale, int in, int i) {
  const Lin l = lin_src(o, scale, in);
  float w = 0.f;
  if (l.i0 == i) w += l.w0;
  if (l.i1 == i) w += l.w1;
  return w;
}
__device__ __forceinline__ void out_range(int i, float scale, int out, int& lo, int& hi) {
  // outputs o whose source coordinate lies in (i-1, i+1), widened by one on each side
  const float inv = 1.f / scale;
  lo = (int)floorf(((float)i - 1.f + 0.5f) * inv - 0.5f) - 1;
  hi = (int)ceilf(((float)i + 1.f + 0.5f) * inv - 0.5f) + 1;
  if (lo < 0) lo = 0;
  if (hi > out - 1) hi = out - 1;
}

__global__ __launch_bounds__(256)
void upsample_add_kernel(const float* __restrict__ x, const float* __restrict__ s0, const float* __restrict__ s1,
                         const float* __restrict__ s2, float* __restrict__ t, int N, int IH, int IW, int OH, int OW,
                         int C, int nt, float* __restrict__ partial) {
  // partial (nullable) [gridDim.x][2][C]: per-block sums / sums of squares of the values written -- the batch statistics
  // of the BatchNorm that follows (network/utils.py:36-41 after :89-102) without a pass of their own.  Needs
  // 256 % (C/4) == 0 (a thread keeps its channel group over the grid-stride loop); double accumulators like
  // colsum_partial_kernel, rounded to fp32 once per block.
  extern __shared__ __attribute__((aligned(16))) double smd_ua[];     // [2][RL][C] when partial
  double a0[4] = {0.0, 0.0, 0.0, 0.0}, a1[4] = {0.0, 0.0, 0.0, 0.0};
  const int C4 = C >> 2;
  const long long total = (long long)N * OH * OW * C4;
  const float sy = (float)IH / (float)OH, sx = (float)IW / (float)OW;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    long long q = i / C4;
    const int ox = (int)(q % OW); q /= OW;
    const int oy = (int)(q % OH);
    const int n = (int)(q / OH);
    const Lin ly = lin_src(oy, sy, IH), lx = lin_src(ox, sx, IW);
    const float* b = x + (long long)n * IH * IW * C + c;
    const float4 v00 = ld4(b + ((long long)ly.i0 * IW + lx.i0) * C), v01 = ld4(b + ((long long)ly.i0 * IW + lx.i1) * C);
    const float4 v10 = ld4(b + ((long long)ly.i1 * IW + lx.i0) * C), v11 = ld4(b + ((long long)ly.i1 * IW + lx.i1) * C);
    const float4 top = f4axpy(lx.w1, v01, f4scale(lx.w0, v00));
    const float4 bot = f4axpy(lx.w1, v11, f4scale(lx.w0, v10));
    float4 r = f4axpy(ly.w1, bot, f4scale(ly.w0, top));
    if (s0) {
      float4 sk = ld4s(s0 + i * 4, nt);     // python sum(): ((0 + s0) + s1) + s2, then x + skip
      if (s1) sk = f4add(sk, ld4s(s1 + i * 4, nt));
      if (s2) sk = f4add(sk, ld4s(s2 + i * 4, nt));
      r = f4add(r, sk);
    }
    st4s(t + i * 4, r, nt);
    if (partial) {
      const double d[4] = {(double)r.x, (double)r.y, (double)r.z, (double)r.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) { a0[e] += d[e]; a1[e] = fma(d[e], d[e], a1[e]); }
    }
  }
  if (partial) {
    const int RL = 256 / C4, col4 = threadIdx.x % C4, rl = threadIdx.x / C4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      smd_ua[(0 * RL + rl) * C + col4 * 4 + e] = a0[e];
      smd_ua[(1 * RL + rl) * C + col4 * 4 + e] = a1[e];
    }
    __syncthreads();
    for (int u = threadIdx.x; u < 2 * C; u += 256) {
      const int which = u / C, c = u - which * C;
      double sum = 0.0;
      for (int k = 0; k < RL; ++k) sum += smd_ua[(which * RL + k) * C + c];     // fixed order
      partial[((long long)blockIdx.x * 2 + which) * C + c] = (float)sum;
    }
  }
}

// fy / fx > 0: OH == fy * IH (OW == fx * IW) with a power-of-two factor: an interior input row i then receives exactly
// the 2f output rows f*i - f/2 + k with weights (k + 0.5)/f, k < f, mirrored after (exact in fp32 = what lin_src
// yields); border rows / columns and other ratios take the generic scan.
__global__ __launch_bounds__(256)
void upsample_bwd_kernel(const float* __restrict__ g, float* __restrict__ gx, int N, int IH, int IW, int OH, int OW,
                         int C, int accumulate, int fy, int fx, unsigned* __restrict__ maxabs) {
  const int C4 = C >> 2;
  float mx = 0.f;
  const long long total = (long long)N * IH * IW * C4;
  const float sy = (float)IH / (float)OH, sx = (float)IW / (float)OW;
  const float inv_fy = fy > 0 ? 1.f / (float)fy : 0.f, inv_fx = fx > 0 ? 1.f / (float)fx : 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    long long q = i / C4;
    const int ix = (int)(q % IW); q /= IW;
    const int iy = (int)(q % IH);
    const int n = (int)(q / IH);
    float4 acc = make_float4(0, 0, 0, 0);
    if (fy > 1 && fx > 1 && iy > 0 && iy < IH - 1 && ix > 0 && ix < IW - 1) {
      const float* base = g + (((long long)n * OH + (fy * iy - (fy >> 1))) * OW + (fx * ix - (fx >> 1))) * C + c;
      for (int ky = 0; ky < 2 * fy; ++ky) {
        const float wy = ((float)(ky < fy ? ky : 2 * fy - 1 - ky) + 0.5f) * inv_fy;
        float4 row = make_float4(0, 0, 0, 0);
        const float* rp = base + (long long)ky * OW * C;
        for (int kx = 0; kx < 2 * fx; ++kx) {
          const float wx = ((float)(kx < fx ? kx : 2 * fx - 1 - kx) + 0.5f) * inv_fx;
          row = f4axpy(wx, ld4(rp + (long long)kx * C), row);
        }
        acc = f4axpy(wy, row, acc);
      }
    } else {
      int ylo, yhi, xlo, xhi;
      out_range(iy, sy, OH, ylo, yhi);
      out_range(ix, sx, OW, xlo, xhi);
      for (int oy = ylo; oy <= yhi; ++oy) {
        const float wy = lin_w(oy, sy, IH, iy);
        if (wy == 0.f) continue;
        float4 row = make_float4(0, 0, 0, 0);
        for (int ox = xlo; ox <= xhi; ++ox) {
          const float wx = lin_w(ox, sx, IW, ix);
          if (wx == 0.f) continue;
          row = f4axpy(wx, ld4(g + (((long long)n * OH + oy) * OW + ox) * C + c), row);
        }
        acc = f4axpy(wy, row, acc);
      }
    }
    if (accumulate) acc = f4add(acc, ld4(gx + i * 4));
    st4(gx + i * 4, acc);
    mx = fmaxf(fmaxf(mx, fmaxf(fabsf(acc.x), fabsf(acc.y))), fmaxf(fabsf(acc.z), fabsf(acc.w)));
  }
  if (maxabs) {                          // max |gx| as float bits, one filtered atomic per block (see bn_bwd_apply_kernel)
    __shared__ float s_mx[4];
    mx = dcs_wave_max(mx);
    if ((threadIdx.x & 63) == 0) s_mx[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
      mx = fmaxf(fmaxf(s_mx[0], s_mx[1]), fmaxf(s_mx[2], s_mx[3]));
      const unsigned bits = __float_as_uint(mx);
      if (bits > __hip_atomic_load(maxabs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(maxabs, bits);
    }
  }
}

__global__ __launch_bounds__(256)
void upsample_to_nchw_kernel(const float* __restrict__ x, float* __restrict__ out, int N, int IH, int IW, int cs, int C,
                             int OH, int OW) {
  const long long total = (long long)N * OH * OW;
  const float sy = (float)IH / (float)OH, sx = (float)IW / (float)OW;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int ox = (int)(i % OW);
    const long long q = i / OW;
    const int oy = (int)(q % OH), n = (int)(q / OH);
    const Lin ly = lin_src(oy, sy, IH), lx = lin_src(ox, sx, IW);
    const float* b = x + (long long)n * IH * IW * cs;
    const float* p00 = b + ((long long)ly.i0 * IW + lx.i0) * cs;
    const float* p01 = b + ((long long)ly.i0 * IW + lx.i1) * cs;
    const float* p10 = b + ((long long)ly.i1 * IW + lx.i0) * cs;
    const float* p11 = b + ((long long)ly.i1 * IW + lx.i1) * cs;
    float* o = out + (long long)n * C * OH * OW + (long long)oy * OW + ox;
    for (int c = 0; c < C; ++c) {
      const float top = fmaf(lx.w1, p01[c], lx.w0 * p00[c]);
      const float bot = fmaf(lx.w1, p11[c], lx.w0 * p10[c]);
      o[(long long)c * OH * OW] = fmaf(ly.w1, bot, ly.w0 * top);
    }
  }
}

// thread per (n, c, iy, ix), ix fastest: reads of g stay within a few adjacent rows of one plane
__global__ __launch_bounds__(256)
void upsample_to_nchw_bwd_kernel(const float* __restrict__ g, const float* __restrict__ gscale, float* __restrict__ gx,
                                 int N, int IH, int IW, int cs, int C, int OH, int OW) {
  const long long total = (long long)N * cs * IH * IW;
  const float sy = (float)IH / (float)OH, sx = (float)IW / (float)OW;
  const float gs = gscale ? gscale[0] : 1.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int ix = (int)(i % IW);
    long long q = i / IW;
    const int iy = (int)(q % IH); q /= IH;
    const int c = (int)(q % cs);
    const int n = (int)(q / cs);
    float acc = 0.f;
    if (c < C) {
      int ylo, yhi, xlo, xhi;
      out_range(iy, sy, OH, ylo, yhi);
      out_range(ix, sx, OW, xlo, xhi);
      const float* gp = g + ((long long)n * C + c) * OH * OW;
      for (int oy = ylo; oy <= yhi; ++oy) {
        const float wy = lin_w(oy, sy, IH, iy);
        if (wy == 0.f) continue;
        float row = 0.f;
        for (int ox = xlo; ox <= xhi; ++ox) {
          const float wx = lin_w(ox, sx, IW, ix);
          if (wx != 0.f) row = fmaf(wx, gp[(long long)oy * OW + ox], row);
        }
        acc = fmaf(wy, row, acc);
      }
      acc *= gs;
    }
    gx[(((long long)n * IH + iy) * IW + ix) * cs + c] = acc;
  }
}

// Separable form of the same adjoint (bilinear weights factor into wy * wx): pass 1 folds the X axis of every
// full-resolution row, r[plane][Y][x] = sum_X wx(X,x) g[plane][Y][X]  (reads g once, coalesced along X);
// pass 2 folds Y and transposes to NHWC through LDS so the channel-strided rows are written as whole lines.
// `f` > 0: OW == f * IW with f a power of two.  Then an interior input column i receives exactly the 2f outputs
// f*i - f/2 + k, k = 0..2f-1, with weights (k + 0.5)/f for k < f and (2f - k - 0.5)/f after (exact in fp32, identical
// to what lin_src yields); border columns and other ratios take the generic scan.
__global__ __launch_bounds__(256)
void upsample_fold_x_kernel(const float* __restrict__ g, float* __restrict__ r, long long planes_rows, int IW, int OW,
                            int f) {
  const long long total = planes_rows * IW;
  const float sx = (float)IW / (float)OW;
  const float inv_f = f > 0 ? 1.f / (float)f : 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int ix = (int)(i % IW);
    const long long row = i / IW;
    const float* gp = g + row * OW;
    float acc = 0.f;
    if (f == 4 && ix > 0 && ix < IW - 1) {
      // the 8 outputs 4*ix-2 .. 4*ix+5 out of three ALIGNED 16-byte loads (whole cache lines per wave instead of eight
      // 4-byte loads at a 16-byte lane stride)
      const float4 a0 = ld4(gp + 4 * ix - 4), a1 = ld4(gp + 4 * ix), a2 = ld4(gp + 4 * ix + 4);
      acc = 0.125f * (a0.z + a2.y);
      acc = fmaf(0.375f, a0.w + a2.x, acc);
      acc = fmaf(0.625f, a1.x + a1.w, acc);
      acc = fmaf(0.875f, a1.y + a1.z, acc);
    } else if (f > 1 && ix > 0 && ix < IW - 1) {
      const float* q = gp + f * ix - (f >> 1);
      for (int k = 0; k < f; ++k) {
        const float w = ((float)k + 0.5f) * inv_f;
        acc = fmaf(w, q[k], acc);
        acc = fmaf(w, q[2 * f - 1 - k], acc);
      }
    } else {
      int xlo, xhi;
      out_range(ix, sx, OW, xlo, xhi);
      for (int ox = xlo; ox <= xhi; ++ox) {
        const float wx = lin_w(ox, sx, IW, ix);
        if (wx != 0.f) acc = fmaf(wx, gp[ox], acc);
      }
    }
    r[i] = acc;
  }
}

constexpr int FOLD_MAXC = 32;
__global__ __launch_bounds__(256)
void upsample_fold_y_nhwc_kernel(const float* __restrict__ r, const float* __restrict__ gscale, float* __restrict__ gx,
                                 int IH, int IW, int cs, int C, int OH) {
  extern __shared__ float tile[];            // [256][cs]
  const int n = blockIdx.z, iy = blockIdx.y;
  const int x0 = blockIdx.x * 256;
  const int ix = x0 + threadIdx.x;
  const float sy = (float)IH / (float)OH;
  const float gs = gscale ? gscale[0] : 1.f;
  float acc[FOLD_MAXC];
#pragma unroll
  for (int c = 0; c < FOLD_MAXC; ++c) acc[c] = 0.f;
  if (ix < IW) {
    int ylo, yhi;
    out_range(iy, sy, OH, ylo, yhi);
    for (int oy = ylo; oy <= yhi; ++oy) {
      const float wy = lin_w(oy, sy, IH, iy);
      if (wy == 0.f) continue;
      const float* rp = r + (((long long)n * C) * OH + oy) * IW + ix;
#pragma unroll
      for (int c = 0; c < FOLD_MAXC; ++c)
        if (c < C) acc[c] = fmaf(wy, rp[(long long)c * OH * IW], acc[c]);
    }
  }
#pragma unroll
  for (int c = 0; c < FOLD_MAXC; ++c)
    if (c < cs) tile[threadIdx.x * cs + c] = c < C ? acc[c] * gs : 0.f;
  __syncthreads();
  const int nx = IW - x0 < 256 ? IW - x0 : 256;
  float* o = gx + (((long long)n * IH + iy) * IW + x0) * cs;
  for (int t = threadIdx.x; t < nx * cs; t += 256) o[t] = tile[t];
}

inline unsigned grid_for(long long n, unsigned cap = 16384) {
  long long b = (n + 255) / 256;
  if (b < 1) b = 1;
  return (unsigned)(b > cap ? cap : b);
}

}  // namespace

extern "C" int dcs_normalize_pyramid(const float* img, float* out0, float* out1, float* out2, int N, int H, int W,
                                     const float* mean3, const float* std3, void* stream) {
  DCS_CHECK_ARG(img && out0 && mean3 && std3 && N > 0 && H > 0 && W > 0);
  hipStream_t s = dcs_stream(stream);
  hipLaunchKernelGGL(normalize_kernel, dim3(grid_for((long long)N * H * W)), dim3(256), 0, s, img, out0, N,
                     (long long)H * W, mean3, std3);
  if (out1) {
    DCS_CHECK_ARG(H / 2 > 0 && W / 2 > 0);
    hipLaunchKernelGGL(bicubic_down_kernel, dim3(grid_for((long long)N * (H / 2) * (W / 2))), dim3(256), 0, s, out0, out1,
                       N, H, W, H / 2, W / 2, 2);
  }
  if (out2) {
    DCS_CHECK_ARG(H / 4 > 0 && W / 4 > 0);
    hipLaunchKernelGGL(bicubic_down_kernel, dim3(grid_for((long long)N * (H / 4) * (W / 4))), dim3(256), 0, s, out0, out2,
                       N, H, W, H / 4, W / 4, 4);
  }
  DCS_LAUNCH_RET();
}

extern "C" int dcs_bn_relu_maxpool(const float* y, const float* bn, float* out, uint8_t* idx, int N, int H, int W, int C,
                                   void* stream) {
  DCS_CHECK_ARG(y && bn && out && idx && N > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0);
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  const long long total = (long long)N * OH * OW * (C / 4);
  hipLaunchKernelGGL(bn_relu_maxpool_kernel, dim3(grid_for(total)), dim3(256), 0, dcs_stream(stream), y, bn, out, idx, N, H,
                     W, OH, OW, C, dcs_streams(total * 16) ? 1 : 0);
  DCS_LAUNCH_RET();
}

extern "C" int dcs_maxpool_bwd(const float* g, const uint8_t* idx, float* gz, int N, int H, int W, int C, void* stream) {
  DCS_CHECK_ARG(g && idx && gz && N > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0);
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  const long long total = (long long)N * H * W * (C / 4);
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, dcs_stream(stream), g, idx, gz, N, H, W, OH,
                     OW, C);
  DCS_LAUNCH_RET();
}

extern "C" int dcs_bn_pool_bwd_partial(const float* g, const uint8_t* idx, const float* y, const float* bn, float* partial,
                                       int N, int H, int W, int C, int groups, void* stream) {
  DCS_CHECK_ARG(g && idx && y && bn && partial && N > 0 && H > 0 && W > 0 && groups > 0);
  DCS_CHECK_ARG(C >= 4 && C <= 1024 && (C & 3) == 0 && 256 % (C / 4) == 0);
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  const size_t sh = (size_t)2 * (256 / (C / 4)) * C * sizeof(double);
  hipLaunchKernelGGL(bn_pool_bwd_partial_kernel, dim3((unsigned)groups), dim3(256), sh, dcs_stream(stream), g, idx, y, bn,
                     partial, N, H, W, OH, OW, C, groups, dcs_streams((long long)N * H * W * C * 4) ? 1 : 0);
  DCS_LAUNCH_RET();
}

extern "C" int dcs_bn_pool_bwd_partial_pooled(const float* g, const float* z, const uint8_t* idx, const float* y,
                                              const float* bn, float* partial, int N, int H, int W, int C, int groups,
                                              void* stream) {
  DCS_CHECK_ARG(g && z && idx && y && bn && partial && N > 0 && H > 0 && W > 0 && groups > 0);
  DCS_CHECK_ARG(C >= 4 && C <= 1024 && (C & 3) == 0 && 256 % (C / 4) == 0 && dcs_aligned16(g) && dcs_aligned16(z));
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  const size_t sh = (size_t)2 * (256 / (C / 4)) * C * sizeof(double);
  hipLaunchKernelGGL(bn_pool_bwd_partial_pooled_kernel, dim3((unsigned)groups), dim3(256), sh, dcs_stream(stream), g, z, idx, y,
                     bn, partial, N, H, W, OH, OW, C, groups);
  DCS_LAUNCH_RET();
}

extern "C" int dcs_bn_pool_bwd_apply(const float* g, const uint8_t* idx, const float* y, const float* bn, const float* gamma,
                                     const float* sums, float* dy, float* dgamma, float* dbeta, int N, int H, int W, int C,
                                     int acc_param, int training, uint32_t* dy_maxabs, void* stream) {
  DCS_CHECK_ARG(g && idx && y && bn && gamma && sums && dy && N > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0);
  DCS_CHECK_ARG((dgamma == nullptr) == (dbeta == nullptr));
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  const long long total = (long long)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4);
  hipLaunchKernelGGL(bn_pool_bwd_apply_kernel, dim3(grid_for(total)), dim3(256), 0, dcs_stream(stream), g, idx, y, bn, gamma,
                     sums, dy, dgamma, dbeta, N, H, W, OH, OW, C, acc_param, training,
                     dcs_streams((long long)N * H * W * C * 4) ? 1 : 0, dy_maxabs);
  DCS_LAUNCH_RET();
}

extern "C" int dcs_upsample_add(const float* x, const float* s0, const float* s1, const float* s2, float* t, int N,
                                int IH, int IW, int OH, int OW, int C, void* stream) {
  DCS_CHECK_ARG(x && t && N > 0 && IH > 0 && IW > 0 && OH > 0 && OW > 0 && C > 0 && (C & 3) == 0);
  DCS_CHECK_ARG(s0 || (!s1 && !s2));
  const long long total = (long long)N * OH * OW * (C / 4);
  hipLaunchKernelGGL(upsample_add_kernel, dim3(grid_for(total)), dim3(256), 0, dcs_stream(stream), x, s0, s1, s2, t, N, IH,
                     IW, OH, OW, C, dcs_streams(total * 16) ? 1 : 0, (float*)nullptr);
  DCS_LAUNCH_RET();
}

extern "C" int dcs_upsample_add_stats(const float* x, const float* s0, const float* s1, const float* s2, float* t,
                                      float* partial, int groups, int N, int IH, int IW, int OH, int OW, int C,
                                      void* stream) {
  DCS_CHECK_ARG(x && t && partial && N > 0 && IH > 0 && IW > 0 && OH > 0 && OW > 0 && C > 0 && (C & 3) == 0);
  DCS_CHECK_ARG(s0 || (!s1 && !s2));
  DCS_CHECK_ARG(C <= 1024 && 256 % (C / 4) == 0 && groups > 0 && groups <= 2048);
  const long long total = (long long)N * OH * OW * (C / 4);
  DCS_CHECK_ARG((long long)groups * 256 <= total + 255);          // every block owns at least one element: partial fully written
  const size_t sh = (size_t)2 * (256 / (C / 4)) * C * sizeof(double);
  hipLaunchKernelGGL(upsample_add_kernel, dim3((unsigned)groups), dim3(256), sh, dcs_stream(stream), x, s0, s1, s2, t, N, IH,
                     IW, OH, OW, C, dcs_streams(total * 16) ? 1 : 0, partial);
  DCS_LAUNCH_RET();
}

extern "C" int dcs_upsample_bwd(const float* g, float* gx, int N, int IH, int IW, int OH, int OW, int C, int accumulate,
                                uint32_t* maxabs, void* stream) {
  DCS_CHECK_ARG(g && gx && N > 0 && IH > 0 && IW > 0 && OH > 0 && OW > 0 && C > 0 && (C & 3) == 0);
  const long long total = (long long)N * IH * IW * (C / 4);
  auto pow2_factor = [](int out, int in) { const int f = out / in; return (out % in == 0 && (f & (f - 1)) == 0 && f <= 16) ? f : 0; };
  hipLaunchKernelGGL(upsample_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, dcs_stream(stream), g, gx, N, IH, IW, OH, OW,
                     C, accumulate, pow2_factor(OH, IH), pow2_factor(OW, IW), maxabs);
  DCS_LAUNCH_RET();
}

extern "C" int dcs_upsample_to_nchw(const float* x, float* out, int N, int IH, int IW, int cs, int C, int OH, int OW,
                                    void* stream) {
  DCS_CHECK_ARG(x && out && N > 0 && IH > 0 && IW > 0 && OH > 0 && OW > 0 && C > 0 && cs >= C);
  hipLaunchKernelGGL(upsample_to_nchw_kernel, dim3(grid_for((long long)N * OH * OW)), dim3(256), 0, dcs_stream(stream), x,
                     out, N, IH, IW, cs, C, OH, OW);
  DCS_LAUNCH_RET();
}

extern "C" int dcs_upsample_to_nchw_bwd(const float* g, const float* gscale, float* gx, float* tmp, int N, int IH,
                                        int IW, int cs, int C, int OH, int OW, void* stream) {
  DCS_CHECK_ARG(g && gx && N > 0 && IH > 0 && IW > 0 && OH > 0 && OW > 0 && C > 0 && cs >= C);
  hipStream_t s = dcs_stream(stream);
  if (tmp && C <= FOLD_MAXC && cs <= 2 * FOLD_MAXC && IH <= 65535 && N <= 65535) {
    // tmp: [N*C*OH*IW] floats
    const long long rows = (long long)N * C * OH;
    int fx = (OW % IW == 0 && ((OW / IW) & (OW / IW - 1)) == 0 && OW / IW <= 16) ? OW / IW : 0;
    if (fx == 4 && !dcs_aligned16(g)) fx = 0;            // the x4 path uses aligned 16-byte loads
    hipLaunchKernelGGL(upsample_fold_x_kernel, dim3(grid_for(rows * IW, 1u << 20)), dim3(256), 0, s, g, tmp, rows, IW, OW,
                       fx);
    hipLaunchKernelGGL(upsample_fold_y_nhwc_kernel, dim3((unsigned)((IW + 255) / 256), (unsigned)IH, (unsigned)N), dim3(256),
                       (size_t)256 * cs * sizeof(float), s, tmp, gscale, gx, IH, IW, cs, C, OH);
    DCS_LAUNCH_RET();
  }
  hipLaunchKernelGGL(upsample_to_nchw_bwd_kernel, dim3(grid_for((long long)N * cs * IH * IW)), dim3(256), 0, s, g, gscale,
                     gx, N, IH, IW, cs, C, OH, OW);
  DCS_LAUNCH_RET();
}
