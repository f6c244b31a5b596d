// HBM-bound kernels around the convolutions: per-channel reductions, BatchNorm finalize /
// apply / backward, fused residual + ReLU.  NHWC fp32, 16 B per lane, deterministic reductions
// (fixed-order partial slabs, no float atomics).
// Replaces nn.BatchNorm2d / nn.ReLU / residual add of network/backbone/resnet_pyramid.py:71-89
// and network/utils.py:35-49 in the reference.
#include <cstdlib>
#include "dcs_common.h"

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// partial[b][g][2][C].  Channels are processed in blocks of <= 512 (blockIdx.z) so that any C % 4 == 0 works
// (ResNet-101 has 2048-channel BatchNorms).
template <int MODE>
__global__ __launch_bounds__(256)
void colsum_partial_kernel(const float* __restrict__ x, const float* __restrict__ y,
                           const float* __restrict__ masksrc, const float* __restrict__ bn,
                           float* __restrict__ partial, long long rows, int Ctot, int cstride, int groups, int relu, int nt) {
  // Per-thread and per-block accumulation in DOUBLE: the BatchNorm-backward sums (sum g, sum g*xhat) cancel heavily
  // (their terms have mixed signs) and a sequential fp32 chain over a thread's few hundred rows loses what the
  // reference's CPU batch_norm_backward (double accumulators, at::acc_type<float, false>) keeps: measured 8-33x the
  // reference's own fp32-vs-fp64 error on stem / layer1 BatchNorm gradients with fp32 chains.  The kernel is HBM-bound
  // (two streamed tensors), the fp64 adds are free.  Partials are rounded to fp32 once per group.
  extern __shared__ __attribute__((aligned(16))) double smd[];   // [2][RL][C]
  const int cbase = blockIdx.z * 512;
  const int C = Ctot - cbase < 512 ? Ctot - cbase : 512;
  const int C4 = C >> 2;
  const int RL = 256 / C4;
  const int tid = threadIdx.x;
  const int col4 = tid % C4, rl = tid / C4;
  const int gi = blockIdx.x, b = blockIdx.y;
  const long long rpg = (rows + groups - 1) / groups;
  const long long rbeg = (long long)gi * rpg;
  const long long rend = rbeg + rpg < rows ? rbeg + rpg : rows;
  const float* xb = x + (long long)b * rows * cstride + cbase;
  double a0[4] = {0.0, 0.0, 0.0, 0.0}, a1[4] = {0.0, 0.0, 0.0, 0.0};
  if (rl < RL) {
    float4 sc, sh, mu, is;
    if (MODE == 1) {
      const int cc = cbase + col4 * 4;
      sc = ld4(bn + cc); sh = ld4(bn + Ctot + cc);
      mu = ld4(bn + 2 * Ctot + cc); is = ld4(bn + 3 * Ctot + cc);
    }
    for (long long r = rbeg + rl; r < rend; r += RL) {
      const long long off = r * cstride + col4 * 4;
      float4 v = ld4s(xb + off, nt);
      if (MODE == 0) {
        const double d[4] = {(double)v.x, (double)v.y, (double)v.z, (double)v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { a0[e] += d[e]; a1[e] = fma(d[e], d[e], a1[e]); }
      } else {
        const float4 yy = ld4s(y + cbase + off, nt);
        if (masksrc) {
          const float4 ms = ld4s(masksrc + cbase + off, nt);
          v.x = ms.x > 0.f ? v.x : 0.f; v.y = ms.y > 0.f ? v.y : 0.f;
          v.z = ms.z > 0.f ? v.z : 0.f; v.w = ms.w > 0.f ? v.w : 0.f;
        } else if (relu) {
          v.x = fmaf(yy.x, sc.x, sh.x) > 0.f ? v.x : 0.f; v.y = fmaf(yy.y, sc.y, sh.y) > 0.f ? v.y : 0.f;
          v.z = fmaf(yy.z, sc.z, sh.z) > 0.f ? v.z : 0.f; v.w = fmaf(yy.w, sc.w, sh.w) > 0.f ? v.w : 0.f;
        }
        const float xh[4] = {(yy.x - mu.x) * is.x, (yy.y - mu.y) * is.y, (yy.z - mu.z) * is.z, (yy.w - mu.w) * is.w};
        const double d[4] = {(double)v.x, (double)v.y, (double)v.z, (double)v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { a0[e] += d[e]; a1[e] = fma(d[e], (double)xh[e], a1[e]); }
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      smd[(0 * RL + rl) * C + col4 * 4 + e] = a0[e];
      smd[(1 * RL + rl) * C + col4 * 4 + e] = a1[e];
    }
  }
  __syncthreads();
  for (int t = tid; t < 2 * C; t += 256) {
    const int which = t / C, c = t - which * C;
    double s = 0.0;
    for (int k = 0; k < RL; ++k) s += smd[(which * RL + k) * C + c];
    partial[(((long long)b * groups + gi) * 2 + which) * Ctot + cbase + c] = (float)s;
  }
}

// out[b][2][C] = scale * sum_g partial[b][g][2][C].  Block = 8 float4 columns x 32 group lanes; each lane sums
// groups gl, gl+32, ... in double (independent 16-B loads, unrolled), then the 32 lanes are added in fixed order.
__global__ __launch_bounds__(256)
void colsum_final_kernel(const float* __restrict__ partial, float* __restrict__ out, int groups, int C,
                                    float scale, double moments_count) {
  __shared__ double sm[32][8][4];
  const int b = blockIdx.y;
  const long long ld = 2ll * C;
  if (moments_count > 0.0) {
    // out[0][c] = mean, out[1][c] = biased variance, formed in double from the double totals (E[x^2] - mean^2 cancels
    // when |mean| >> std; rounding the two sums to fp32 first would lose those digits).  32 channels x 8 group lanes.
    if (blockIdx.x * 32 >= C) return;
    const int cl = threadIdx.x & 31, gl2 = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    double s0 = 0.0, s1 = 0.0;
    if (c < C) {
      const float* p = partial + (long long)b * groups * ld;
      // the adds keep their order (bitwise the rolled loop); unrolling only puts 16 independent loads in flight per
      // trip -- with 2048 row tiles this is a 256-trip latency chain on a handful of blocks
#pragma unroll 8
      for (int gi = gl2; gi < groups; gi += 8) { s0 += (double)p[(long long)gi * ld + c]; s1 += (double)p[(long long)gi * ld + C + c]; }
    }
    double (*sm2)[32][2] = reinterpret_cast<double (*)[32][2]>(&sm[0][0][0]);
    sm2[gl2][cl][0] = s0; sm2[gl2][cl][1] = s1;
    __syncthreads();
    if (threadIdx.x < 32 && c < C) {
      double t0 = 0.0, t1 = 0.0;
#pragma unroll
      for (int k = 0; k < 8; ++k) { t0 += sm2[k][cl][0]; t1 += sm2[k][cl][1]; }
      const double m = t0 / moments_count;
      double var = t1 / moments_count - m * m;
      if (var < 0.0) var = 0.0;
      out[(long long)b * ld + c] = (float)m;
      out[(long long)b * ld + C + c] = (float)var;
    }
    return;
  }
  const int cl = threadIdx.x & 7, gl = threadIdx.x >> 3;
  const int col = (blockIdx.x * 8 + cl) * 4;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  if (col < 2 * C) {
    const float* p = partial + (long long)b * groups * ld + col;
#pragma unroll 4
    for (int gi = gl; gi < groups; gi += 32) {
      const float4 v = ld4(p + (long long)gi * ld);
      s0 += (double)v.x; s1 += (double)v.y; s2 += (double)v.z; s3 += (double)v.w;
    }
  }
  sm[gl][cl][0] = s0; sm[gl][cl][1] = s1; sm[gl][cl][2] = s2; sm[gl][cl][3] = s3;
  __syncthreads();
  if (threadIdx.x < 32) {
    const int c8 = threadIdx.x >> 2, j = threadIdx.x & 3;
    const int oc = (blockIdx.x * 8 + c8) * 4 + j;
    if (oc < 2 * C) {
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < 32; ++k) t += sm[k][c8][j];
      out[(long long)b * ld + oc] = (float)(t * (double)scale);
    }
  }
}

__global__ void bn_finalize_kernel(const float* __restrict__ sums, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* __restrict__ rm, float* __restrict__ rv,
                                   float* __restrict__ bn, int C, double count, float eps, float momentum,
                                   int repeats, int training) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float mean, invstd;
  if (training) {
    // training == 2: sums already holds (mean, biased variance) formed in double by dcs_colsum_final
    const double m = training == 2 ? (double)sums[c] : (double)sums[c] / count;
    double var = training == 2 ? (double)sums[C + c] : (double)sums[C + c] / count - m * m;
    if (var < 0.0) var = 0.0;
    mean = (float)m;
    invstd = (float)(1.0 / sqrt(var + (double)eps));
    if (rm) {
      const float vu = (float)(var * (count / (count > 1.0 ? count - 1.0 : 1.0)));
      float a = rm[c], b = rv[c];
      for (int k = 0; k < repeats; ++k) {
        a = (1.f - momentum) * a + momentum * mean;
        b = (1.f - momentum) * b + momentum * vu;
      }
      rm[c] = a; rv[c] = b;
    }
  } else {
    mean = rm[c];
    invstd = 1.f / sqrtf(rv[c] + eps);
  }
  const float sc = gamma[c] * invstd;
  bn[c] = sc;
  bn[C + c] = fmaf(-mean, sc, beta[c]);
  bn[2 * C + c] = mean;
  bn[3 * C + c] = invstd;
}

__global__ void bn_ema_again_kernel(const float* __restrict__ bn, float* __restrict__ rm, float* __restrict__ rv,
                                    int C, double count, float eps, float momentum) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float mean = bn[2 * C + c];
  const double is = (double)bn[3 * C + c];
  double var = 1.0 / (is * is) - (double)eps;
  if (var < 0.0) var = 0.0;
  const float vu = (float)(var * (count / (count > 1.0 ? count - 1.0 : 1.0)));
  rm[c] = (1.f - momentum) * rm[c] + momentum * mean;
  rv[c] = (1.f - momentum) * rv[c] + momentum * vu;
}

__global__ __launch_bounds__(256)
void bn_act_kernel(const float* __restrict__ y, const float* __restrict__ bn, const float* __restrict__ r,
                   const float* __restrict__ bn2, float* __restrict__ z, long long n4, int C, int relu, int nt,
                   unsigned char* __restrict__ mask8) {
  const int C4 = C >> 2;
  // C/4 divides the block size for every BatchNorm width of the networks (64..2048 channels): a thread then keeps ONE
  // channel group for the whole grid-stride loop and its per-channel constants live in registers (7 fewer L1 loads per
  // 16 B streamed in the backward kernel below)
  const bool hoist = (256 % C4) == 0;
  float4 sc, sh, s2, h2;
  auto tab = [&](int c) {
    sc = ld4(bn + c); sh = ld4(bn + C + c);
    if (r && bn2) { s2 = ld4(bn2 + c); h2 = ld4(bn2 + C + c); }
  };
  if (hoist) tab((threadIdx.x % C4) * 4);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    if (!hoist) tab((int)(i % C4) * 4);
    const float4 v = ld4s(y + i * 4, nt);
    float4 o;
    o.x = fmaf(v.x, sc.x, sh.x); o.y = fmaf(v.y, sc.y, sh.y); o.z = fmaf(v.z, sc.z, sh.z); o.w = fmaf(v.w, sc.w, sh.w);
    if (r) {
      float4 q = ld4s(r + i * 4, nt);
      if (bn2) {
        q.x = fmaf(q.x, s2.x, h2.x); q.y = fmaf(q.y, s2.y, h2.y); q.z = fmaf(q.z, s2.z, h2.z); q.w = fmaf(q.w, s2.w, h2.w);
      }
      o.x += q.x; o.y += q.y; o.z += q.z; o.w += q.w;
    }
    if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
    st4s(z + i * 4, o, nt);
    // the ReLU mask of the four values as one byte: the backward (dcs_bn_bwd_apply, the BatchNorm-backward epilogue of the
    // data gradient) then reads 1 byte instead of 16 of z
    if (mask8) mask8[i] = (unsigned char)((o.x > 0.f ? 1 : 0) | (o.y > 0.f ? 2 : 0) | (o.z > 0.f ? 4 : 0) | (o.w > 0.f ? 8 : 0));
  }
}

template <bool NT>
__global__ __launch_bounds__(256)
void bn_bwd_apply_kernel(const float* __restrict__ g, const float* __restrict__ y, const float* __restrict__ masksrc,
                         const float* __restrict__ bn, const float* __restrict__ gamma, const float* __restrict__ sums,
                         float* __restrict__ dy, float* __restrict__ gm_out, float* __restrict__ dgamma,
                         float* __restrict__ dbeta, long long rows, int C, int relu, int acc_dy, int acc_gm,
                         int acc_param, int training, unsigned* __restrict__ dy_maxabs,
                         const unsigned char* __restrict__ mask8) {
  const int C4 = C >> 2;
  float mx = 0.f;
  const long long n4 = rows * C4;
  // eval-mode BatchNorm is a fixed affine map: the batch-statistics terms vanish
  const float inv = training ? (float)(1.0 / (double)rows) : 0.f;
  if (blockIdx.x == 0 && dgamma) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      dbeta[c] = (acc_param ? dbeta[c] : 0.f) + sums[c];
      dgamma[c] = (acc_param ? dgamma[c] : 0.f) + sums[C + c];
    }
  }
  const bool hoist = (256 % C4) == 0;                      // see bn_act_kernel
  float4 sc, sh, mu, is, gi, t0, t1;
  auto tab = [&](int c) {
    sc = ld4(bn + c); sh = ld4(bn + C + c); mu = ld4(bn + 2 * C + c); is = ld4(bn + 3 * C + c);
    if (dy) {
      const float4 gw = ld4(gamma + c), s0 = ld4(sums + c), s1 = ld4(sums + C + c);
      gi = make_float4(gw.x * is.x, gw.y * is.y, gw.z * is.z, gw.w * is.w);
      t0 = make_float4(s0.x * inv, s0.y * inv, s0.z * inv, s0.w * inv);
      t1 = make_float4(s1.x * inv, s1.y * inv, s1.z * inv, s1.w * inv);
    }
  };
  if (hoist) tab((threadIdx.x % C4) * 4);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    if (!hoist) tab((int)(i % C4) * 4);
    float4 v = ldx4<NT>(g + i * 4);
    const float4 yy = ldx4<NT>(y + i * 4);
    if (mask8) {                                          // one byte per four values (dcs_bn_act's mask8)
      const unsigned m = mask8[i];
      v.x = (m & 1u) ? v.x : 0.f; v.y = (m & 2u) ? v.y : 0.f; v.z = (m & 4u) ? v.z : 0.f; v.w = (m & 8u) ? v.w : 0.f;
    } else if (masksrc) {
      const float4 ms = ldx4<NT>(masksrc + i * 4);
      v.x = ms.x > 0.f ? v.x : 0.f; v.y = ms.y > 0.f ? v.y : 0.f; v.z = ms.z > 0.f ? v.z : 0.f; v.w = ms.w > 0.f ? v.w : 0.f;
    } else if (relu) {
      v.x = fmaf(yy.x, sc.x, sh.x) > 0.f ? v.x : 0.f; v.y = fmaf(yy.y, sc.y, sh.y) > 0.f ? v.y : 0.f;
      v.z = fmaf(yy.z, sc.z, sh.z) > 0.f ? v.z : 0.f; v.w = fmaf(yy.w, sc.w, sh.w) > 0.f ? v.w : 0.f;
    }
    if (gm_out) {
      float4 o = v;
      if (acc_gm) { const float4 p = ld4(gm_out + i * 4); o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w; }
      stx4<NT>(gm_out + i * 4, o);
    }
    if (dy) {
      float4 o;                                            // = gamma * invstd * (g - mean(g) - xhat * mean(g * xhat))
      o.x = gi.x * (v.x - t0.x - (yy.x - mu.x) * is.x * t1.x);
      o.y = gi.y * (v.y - t0.y - (yy.y - mu.y) * is.y * t1.y);
      o.z = gi.z * (v.z - t0.z - (yy.z - mu.z) * is.z * t1.z);
      o.w = gi.w * (v.w - t0.w - (yy.w - mu.w) * is.w * t1.w);
      if (acc_dy) { const float4 p = ld4(dy + i * 4); o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w; }
      stx4<NT>(dy + i * 4, o);
      mx = fmaxf(fmaxf(mx, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
    }
  }
  if (dy_maxabs) {                                         // max of non-negative floats = max of their bit patterns
    // one candidate per block; the word only grows, so a (possibly stale) plain read that is already >= the candidate
    // makes the atomic unnecessary: a few dozen same-address atomics per launch instead of thousands (they serialise)
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

// max |x| of a tensor as the bit pattern of a float (see dcs_bn_bwd_apply's dy_maxabs): one filtered atomicMax per block
__global__ __launch_bounds__(256)
void maxabs_kernel(const float* __restrict__ x, const long long n4, unsigned* __restrict__ out, const int nt) {
  float mx = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const float4 v = ld4s(x + i * 4, nt);
    mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  __shared__ float s_mx[4];
  mx = dcs_wave_max(mx);
  if ((threadIdx.x & 63) == 0) s_mx[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    mx = fmaxf(fmaxf(s_mx[0], s_mx[1]), fmaxf(s_mx[2], s_mx[3]));
    const unsigned bits = __float_as_uint(mx);
    if (bits > __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out, bits);
  }
}

__global__ void scale_inplace_kernel(float* __restrict__ x, long long n, const float* __restrict__ a,
                                     const float* __restrict__ b) {
  const float s = a[0] * (b ? b[0] : 1.f);
  const long long n4 = n >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 v = ld4(x + i * 4);
    v.x *= s; v.y *= s; v.z *= s; v.w *= s;
    st4(x + i * 4, v);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) x[n4 * 4 + threadIdx.x] *= s;
}

__global__ void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, long long n, float a) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    y[i] = fmaf(a, x[i], y[i]);
}
// n % 4 == 0, 16-byte aligned operands
__global__ __launch_bounds__(256)
void axpy4_kernel(float* __restrict__ y, const float* __restrict__ x, long long n4, float a, int nt) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const float4 u = ld4s(x + i * 4, nt);
    float4 v = ld4s(y + i * 4, nt);
    v.x = fmaf(a, u.x, v.x); v.y = fmaf(a, u.y, v.y); v.z = fmaf(a, u.z, v.z); v.w = fmaf(a, u.w, v.w);
    st4s(y + i * 4, v, nt);
  }
}

__global__ void add_rowvec_bcast_kernel(float* __restrict__ g, const float* __restrict__ v, long long HW, int C,
                                        float scale, long long n4, int accumulate) {
  const int C4 = C >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    const long long n = i / (HW * C4);
    const float4 a = ld4(v + n * C + c);
    float4 o = accumulate ? ld4(g + i * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    o.x = fmaf(scale, a.x, o.x); o.y = fmaf(scale, a.y, o.y); o.z = fmaf(scale, a.z, o.z); o.w = fmaf(scale, a.w, o.w);
    st4(g + i * 4, o);
  }
}

__global__ void relu_bwd_kernel(const float* __restrict__ g, const float* __restrict__ z, float* __restrict__ out,
                                long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    out[i] = z[i] > 0.f ? g[i] : 0.f;
}

// nn.Dropout: out = x * noise * scale, noise in {0,1}.  noise == null: the keep mask is generated on the fly from a
// counter-based hash of (seed, element index) and written to mask_out (uint8) for the backward pass.
__device__ __forceinline__ unsigned pcg_hash(unsigned v) {
  unsigned s = v * 747796405u + 2891336453u;
  unsigned w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
  return (w >> 22u) ^ w;
}
__global__ void dropout_kernel(const float* __restrict__ x, const float* __restrict__ noise,
                               unsigned char* __restrict__ mask_out, float* __restrict__ out, long long n, float p,
                               float scale, unsigned seed) {
  const unsigned thr = (unsigned)((double)p * 4294967296.0);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float keep;
    if (noise) keep = noise[i];
    else keep = pcg_hash(seed ^ pcg_hash((unsigned)i) ^ (unsigned)(i >> 32)) >= thr ? 1.f : 0.f;
    if (mask_out) mask_out[i] = keep != 0.f;
    out[i] = x[i] * keep * scale;
  }
}
__global__ void dropout_bwd_kernel(const float* __restrict__ g, const unsigned char* __restrict__ mask,
                                   float* __restrict__ out, long long n, float scale) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    out[i] = mask[i] ? g[i] * scale : 0.f;
}

__global__ void sum_scalar_kernel(const float* __restrict__ x, float* __restrict__ out, int n, float scale) {
  __shared__ double sm[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += (double)x[i];
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)(sm[0] * (double)scale);
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long long n, float lr, float b1, float b2, float eps, float wd,
                            float bc1, float bc2s) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float w = p[i];
    const float gr = fmaf(wd, w, g[i]);
    const float mi = b1 * m[i] + (1.f - b1) * gr;
    const float vi = b2 * v[i] + (1.f - b2) * gr * gr;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2s + eps;
    p[i] = w - (lr / bc1) * (mi / denom);
  }
}

inline unsigned grid_for(long long n, int per = 256, unsigned cap = 8192) {
  long long b = (n + per - 1) / per;
  if (b < 1) b = 1;
  return (unsigned)(b > cap ? cap : b);
}

}  // namespace

extern "C" int dcs_colsum_partial(const float* x, const float* y, const float* masksrc, const float* bn,
                                  float* partial, int B, int64_t rows, int C, int cstride, int groups, int mode,
                                  int relu, void* stream) {
  DCS_CHECK_ARG(x && partial && B > 0 && rows > 0 && groups > 0 && C > 0 && (C & 3) == 0 && C <= 8192);
  DCS_CHECK_ARG((cstride & 3) == 0 && cstride >= C && dcs_aligned16(x));
  DCS_CHECK_ARG(mode == 0 || (mode == 1 && y && bn && cstride == C));
  const int Cb = C < 512 ? C : 512;
  const int C4 = Cb / 4, RL = 256 / C4;
  const size_t sh = (size_t)2 * RL * Cb * sizeof(double);
  dim3 grid((unsigned)groups, (unsigned)B, (unsigned)((C + 511) / 512));
  const int nt = dcs_streams((long long)B * rows * cstride * 4) ? 1 : 0;
  if (mode == 0)
    hipLaunchKernelGGL(colsum_partial_kernel<0>, grid, dim3(256), sh, dcs_stream(stream), x, y, masksrc, bn, partial,
                       (long long)rows, C, cstride, groups, relu, nt);
  else
    hipLaunchKernelGGL(colsum_partial_kernel<1>, grid, dim3(256), sh, dcs_stream(stream), x, y, masksrc, bn, partial,
                       (long long)rows, C, cstride, groups, relu, nt);
  DCS_LAUNCH_RET();
}

extern "C" int dcs_colsum_final(const float* partial, float* out, int B, int groups, int C, float scale,
                                double moments_count, void* stream) {
  DCS_CHECK_ARG(partial && out && B > 0 && groups > 0 && C > 0 && (C & 1) == 0 && dcs_aligned16(partial));
  DCS_CHECK_ARG(moments_count == 0.0 || (moments_count > 0.0 && scale == 1.f));
  hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)((2 * C + 31) / 32), (unsigned)B), dim3(256), 0, dcs_stream(stream),
                     partial, out, groups, C, scale, moments_count);
  DCS_LAUNCH_RET();
}

extern "C" int dcs_bn_finalize(const float* sums, const float* gamma, const float* beta, float* running_mean,
                               float* running_var, float* bn, int C, double count, float eps, float momentum,
                               int repeats, int training, void* stream) {
  DCS_CHECK_ARG(gamma && beta && bn && C > 0 && (training ? sums != nullptr : (running_mean && running_var)));
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 127) / 128), dim3(128), 0, dcs_stream(stream), sums, gamma, beta,
                     running_mean, running_var, bn, C, count, eps, momentum, repeats, training);
  DCS_LAUNCH_RET();
}

extern "C" int dcs_bn_ema_again(const float* bn, float* running_mean, float* running_var, int C, double count,
                                float eps, float momentum, void* stream) {
  DCS_CHECK_ARG(bn && running_mean && running_var && C > 0);
  hipLaunchKernelGGL(bn_ema_again_kernel, dim3((C + 127) / 128), dim3(128), 0, dcs_stream(stream), bn, running_mean,
                     running_var, C, count, eps, momentum);
  DCS_LAUNCH_RET();
}

extern "C" int dcs_bn_act(const float* y, const float* bn, const float* r, const float* bn2, float* z, int64_t rows,
                          int C, int relu, uint8_t* mask8, void* stream) {
  DCS_CHECK_ARG(y && bn && z && rows > 0 && C > 0 && (C & 3) == 0 && dcs_aligned16(y) && dcs_aligned16(z));
  const long long n4 = (long long)rows * (C / 4);
  hipLaunchKernelGGL(bn_act_kernel, dim3(grid_for(n4)), dim3(256), 0, dcs_stream(stream), y, bn, r, bn2, z, n4, C, relu,
                     dcs_streams(n4 * 16) ? 1 : 0, mask8);
  DCS_LAUNCH_RET();
}

extern "C" int dcs_bn_bwd_apply(const float* g, const float* y, const float* masksrc, const float* bn,
                                const float* gamma, const float* sums, float* dy, float* gm_out, float* dgamma,
                                float* dbeta, int64_t rows, int C, int relu, int acc_dy, int acc_gm, int acc_param,
                                int training, uint32_t* dy_maxabs, const uint8_t* mask8, void* stream) {
  DCS_CHECK_ARG(g && y && bn && rows > 0 && C > 0 && (C & 3) == 0 && !(mask8 && masksrc));
  DCS_CHECK_ARG(!dy_maxabs || dy);
  DCS_CHECK_ARG(!dy || (gamma && sums));
  DCS_CHECK_ARG((dgamma == nullptr) == (dbeta == nullptr) && (!dgamma || sums));
  const long long n4 = (long long)rows * (C / 4);
  const bool nt = dcs_streams(n4 * 16);               // >= 256 MiB per tensor: nothing to keep in the caches
  if (nt)
    hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, dim3(grid_for(n4)), dim3(256), 0, dcs_stream(stream), g, y, masksrc, bn, gamma,
                       sums, dy, gm_out, dgamma, dbeta, (long long)rows, C, relu, acc_dy, acc_gm, acc_param, training, dy_maxabs, mask8);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, dim3(grid_for(n4)), dim3(256), 0, dcs_stream(stream), g, y, masksrc, bn, gamma,
                       sums, dy, gm_out, dgamma, dbeta, (long long)rows, C, relu, acc_dy, acc_gm, acc_param, training, dy_maxabs, mask8);
  DCS_LAUNCH_RET();
}

extern "C" int dcs_maxabs(const float* x, int64_t n, uint32_t* out, void* stream) {
  DCS_CHECK_ARG(x && out && n > 0 && (n & 3) == 0 && dcs_aligned16(x));
  hipLaunchKernelGGL(maxabs_kernel, dim3(grid_for(n / 4)), dim3(256), 0, dcs_stream(stream), x, (long long)(n / 4), out,
                     dcs_streams((long long)n * 4) ? 1 : 0);
  DCS_LAUNCH_RET();
}

extern "C" int dcs_scale_inplace(float* x, int64_t n, const float* a, const float* b, void* stream) {
  DCS_CHECK_ARG(x && a && n > 0 && dcs_aligned16(x));
  hipLaunchKernelGGL(scale_inplace_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, dcs_stream(stream), x, (long long)n, a, b);
  DCS_LAUNCH_RET();
}

extern "C" int dcs_axpy(float* y, const float* x, int64_t n, float a, void* stream) {
  DCS_CHECK_ARG(x && y && n > 0);
  if ((n & 3) == 0 && dcs_aligned16(x) && dcs_aligned16(y))
    hipLaunchKernelGGL(axpy4_kernel, dim3(grid_for(n / 4)), dim3(256), 0, dcs_stream(stream), y, x, (long long)n / 4, a,
                       dcs_streams((long long)n * 4) ? 1 : 0);
  else
    hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n)), dim3(256), 0, dcs_stream(stream), y, x, (long long)n, a);
  DCS_LAUNCH_RET();
}

extern "C" int dcs_add_rowvec_bcast(float* g, const float* v, int N, int64_t HW, int C, float scale, int accumulate,
                                    void* stream) {
  DCS_CHECK_ARG(g && v && N > 0 && HW > 0 && C > 0 && (C & 3) == 0);
  const long long n4 = (long long)N * HW * (C / 4);
  hipLaunchKernelGGL(add_rowvec_bcast_kernel, dim3(grid_for(n4)), dim3(256), 0, dcs_stream(stream), g, v, (long long)HW, C,
                     scale, n4, accumulate);
  DCS_LAUNCH_RET();
}

extern "C" int dcs_relu_bwd_rows(const float* g, const float* z, float* out, int64_t n, void* stream) {
  DCS_CHECK_ARG(g && z && out && n > 0);
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, dcs_stream(stream), g, z, out, (long long)n);
  DCS_LAUNCH_RET();
}

extern "C" int dcs_dropout(const float* x, const float* noise, uint8_t* mask_out, float* out, int64_t n, float p,
                           uint32_t seed, void* stream) {
  DCS_CHECK_ARG(x && out && n > 0 && p >= 0.f && p < 1.f && (noise || mask_out));
  hipLaunchKernelGGL(dropout_kernel, dim3(grid_for(n)), dim3(256), 0, dcs_stream(stream), x, noise, mask_out, out,
                     (long long)n, p, 1.f / (1.f - p), seed);
  DCS_LAUNCH_RET();
}

extern "C" int dcs_dropout_bwd(const float* g, const uint8_t* mask, float* out, int64_t n, float p, void* stream) {
  DCS_CHECK_ARG(g && mask && out && n > 0 && p >= 0.f && p < 1.f);
  hipLaunchKernelGGL(dropout_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, dcs_stream(stream), g, mask, out, (long long)n,
                     1.f / (1.f - p));
  DCS_LAUNCH_RET();
}

extern "C" int dcs_sum_scalar(const float* x, float* out, int n, float scale, void* stream) {
  DCS_CHECK_ARG(x && out && n > 0);
  hipLaunchKernelGGL(sum_scalar_kernel, dim3(1), dim3(256), 0, dcs_stream(stream), x, out, n, scale);
  DCS_LAUNCH_RET();
}

extern "C" int dcs_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                             float beta2, float eps, float wd, int step, void* stream) {
  DCS_CHECK_ARG(p && g && m && v && n > 0 && step > 0);
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, dcs_stream(stream), p, g, m, v, (long long)n, lr, beta1,
                     beta2, eps, wd, bc1, bc2s);
  DCS_LAUNCH_RET();
}

extern "C" const char* dcs_version(void) { return "dcs_hip 0.1 (gfx950)"; }
