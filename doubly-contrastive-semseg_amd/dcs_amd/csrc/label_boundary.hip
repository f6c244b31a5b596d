// Boundary-aware label weights on the device (the step BEFORE the train step).
// Replaces LabelBoundaryTransform (dataloaders/custom_transforms_acdc.py:656-693 of the reference), which
// runs cv2.distanceTransform(mask, DIST_L2, maskSize=3) once per present class on the CPU.
//
// OpenCV's 3x3 DIST_L2 transform is the two-pass chamfer transform in 16.16 fixed point with step costs
// a = round(0.955 * 65536) = 62587 (edge neighbours) and b = round(1.3693 * 65536) = 89738 (diagonal neighbours),
// "far" = INT_MAX >> 2 outside the image, result (float)t * 2^-16.  The two raster passes compute exactly the
// shortest 8-connected path length to the nearest zero pixel.  Summed over classes, a pixel's value is the path
// length to the nearest pixel whose label differs from its own, which equals
//     d(p) = min over boundary pixels s of  chamfer(p, s) + e(s),
// s = any pixel with a differently-labelled 8-neighbour, e(s) = a if an edge neighbour differs, else b
// (a pixel of another label than p can only give a larger value, so the minimum needs no label test).
// So ONE transform with seeds e(s) serves all classes.  Integer min-plus arithmetic: bit-exact, order-free.
//
// Kernel 1 (parallel): seeds.  Kernel 2 (one 1024-thread block per image): the forward and backward raster
// sweeps, each row = elementwise min over the previous row's three neighbours + a prefix-min scan (the in-row
// recurrence d(x) = min(t(x), d(x-1) + a) is the prefix minimum of t(k) - a*k), then per-image std and
// exp(-d / (2 std)) weights.  Rows are sequential (2*H steps per image), images run side by side.
#include "dcs_common.h"

namespace {

constexpr int HV = 62587;            // cvRound(0.955f  * 65536)
constexpr int DG = 89738;            // cvRound(1.3693f * 65536)
constexpr int FAR = 0x7FFFFFFF >> 2; // OpenCV DIST_MAX
constexpr int NT = 512;
constexpr int MAX_EPT = 8;           // columns per thread -> W <= 4096

__global__ __launch_bounds__(256)
void boundary_seed_kernel(const long long* __restrict__ lab, int* __restrict__ dist, int B, int H, int W) {
  const long long total = (long long)B * H * W;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W);
    const long long q = i / W;
    const int y = (int)(q % H);
    const long long* img = lab + (q / H) * (long long)H * W;
    const long long me = img[(long long)y * W + x];
    bool edge = false, diag = false;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
      const int yy = y + dy;
      if ((unsigned)yy >= (unsigned)H) continue;
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int xx = x + dx;
        if ((dx == 0 && dy == 0) || (unsigned)xx >= (unsigned)W) continue;
        if (img[(long long)yy * W + xx] != me) {
          if (dx == 0 || dy == 0) edge = true; else diag = true;
        }
      }
    }
    dist[i] = edge ? HV : (diag ? DG : FAR);
  }
}

__device__ __forceinline__ int sat_add(int v, int inc) { const int r = v + inc; return r < FAR ? r : FAR; }

// inclusive prefix minimum over the block's NT * ept values (thread-contiguous).  Keys t - a*x fit 32 bits:
// t <= FAR = 2^29, a*x < 62587 * 4096 = 2^28.
template <int EPT>
__device__ __forceinline__ void block_prefix_min(int (&v)[EPT], int* wave_tot) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int e = 1; e < EPT; ++e) v[e] = v[e] < v[e - 1] ? v[e] : v[e - 1];
  int run = v[EPT - 1];
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int other = __shfl_up(run, o, 64);
    if (lane >= o && other < run) run = other;
  }
  if (lane == 63) wave_tot[wid] = run;
  __syncthreads();
  int pre = 0x7FFFFFFF;
  for (int w = 0; w < wid; ++w) pre = wave_tot[w] < pre ? wave_tot[w] : pre;
  const int left = __shfl_up(run, 1, 64);
  if (lane > 0 && left < pre) pre = left;
#pragma unroll
  for (int e = 0; e < EPT; ++e) v[e] = v[e] < pre ? v[e] : pre;
}

// EPT (columns per thread) is a compile-time constant so that the per-row values live in registers and the loads of
// row r+1 stay in flight across the scan of row r (with a run-time trip count hipcc waits for every load at once).
template <int EPT>
__global__ __launch_bounds__(NT)
void boundary_sweep_kernel(const long long* __restrict__ lab, int* __restrict__ dist, float* __restrict__ img_std, int H,
                           int W, int num_classes) {
  extern __shared__ int rowbuf[];            // [2][W + 2], entries 0 and W+1 stay FAR
  __shared__ int wave_tot[2][NT / 64];
  __shared__ double red[2][NT / 64];
  const int tid = threadIdx.x;
  const long long* li = lab + (long long)blockIdx.x * H * W;
  int* di = dist + (long long)blockIdx.x * H * W;
  const int RW = W + 2;
  for (int i = tid; i < 2 * RW; i += NT) rowbuf[i] = FAR;
  __syncthreads();

  double sum = 0.0, sumsq = 0.0;
  for (int pass = 0; pass < 2; ++pass) {
    // pass 0: rows top->bottom, columns left->right.  pass 1: mirrored in both axes (the same code on flipped indices).
    if (pass == 1) {
      for (int i = tid; i < 2 * RW; i += NT) rowbuf[i] = FAR;
      __syncthreads();
    }
    // the row's own values (seeds in pass 0, forward distances in pass 1) and, in pass 1, its labels are fetched one
    // row ahead: the row-to-row dependency goes through LDS only, so no global-memory latency sits on the chain
    int tv[EPT];
    long long lv[EPT];
    auto fetch = [&](int r) {
      const int y = pass ? H - 1 - r : r;
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const int xs = tid * EPT + e;
        if (r < H && xs < W) {
          const int x = pass ? W - 1 - xs : xs;
          tv[e] = di[(long long)y * W + x];
          if (pass == 1) lv[e] = li[(long long)y * W + x];
        }
      }
    };
    fetch(0);
    for (int r = 0; r < H; ++r) {
      const int y = pass ? H - 1 - r : r;
      const int* prev = rowbuf + ((r + 1) & 1) * RW;      // previous row of this sweep (all FAR for r == 0)
      int* cur = rowbuf + (r & 1) * RW;
      int key[EPT];
      long long lab_now[EPT];
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const int xs = tid * EPT + e;                     // sweep column
        int k = 0x7FFFFFFF;
        if (xs < W) {
          int t = tv[e];
          const int up = sat_add(prev[xs + 1], HV), ul = sat_add(prev[xs], DG), ur = sat_add(prev[xs + 2], DG);
          t = t < up ? t : up; t = t < ul ? t : ul; t = t < ur ? t : ur;
          k = t - HV * xs;
        }
        key[e] = k;
        lab_now[e] = lv[e];
      }
      fetch(r + 1);
      block_prefix_min<EPT>(key, wave_tot[r & 1]);
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const int xs = tid * EPT + e;
        if (xs < W) {
          const int dd = key[e] + HV * xs;
          const int d = dd < FAR ? dd : FAR;
          cur[xs + 1] = d;
          const int x = pass ? W - 1 - xs : xs;
          di[(long long)y * W + x] = d;
          if (pass == 1) {
            const long long l = lab_now[e];
            const float f = (l >= 0 && l < num_classes) ? (float)d * (1.f / 65536.f) : 0.f;
            sum += (double)f; sumsq += (double)f * (double)f;
          }
        }
      }
      __syncthreads();
    }
  }
  // per-image population standard deviation of the float distances (np.std)
  sum = dcs_wave_sum_d(sum); sumsq = dcs_wave_sum_d(sumsq);
  if ((tid & 63) == 0) { red[0][tid >> 6] = sum; red[1][tid >> 6] = sumsq; }
  __syncthreads();
  if (tid == 0) {
    double a = 0.0, b = 0.0;
    for (int w = 0; w < NT / 64; ++w) { a += red[0][w]; b += red[1][w]; }
    const double n = (double)H * W, mean = a / n;
    double var = b / n - mean * mean;
    if (var < 0.0) var = 0.0;
    float sd = (float)sqrt(var);
    if (sd == 0.f) sd = 1.f;
    img_std[blockIdx.x] = sd;
  }
}

// weight = exp(-d / (2 std)) over the whole batch (parallel; the sweep kernel only has one block per image)
__global__ __launch_bounds__(256)
void boundary_weight_kernel(const long long* __restrict__ lab, const int* __restrict__ dist,
                            const float* __restrict__ img_std, float* __restrict__ weight, long long hw, long long total,
                            int num_classes, long long ignore_id) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long l = lab[i];
    const float f = (l >= 0 && l < num_classes) ? (float)dist[i] * (1.f / 65536.f) : 0.f;
    weight[i] = l == ignore_id ? 0.f : expf(-(f / (2.f * img_std[i / hw])));
  }
}

}  // namespace

extern "C" int dcs_label_boundary_weights(const int64_t* labels, int32_t* dist, float* img_std, float* weight, int B, int H,
                                          int W, int num_classes, int64_t ignore_id, void* stream) {
  DCS_CHECK_ARG(labels && dist && img_std && weight && B > 0 && H > 0 && W > 0 && num_classes > 0);
  if (W > NT * MAX_EPT) return DCS_E_UNSUPPORTED;
  hipStream_t s = dcs_stream(stream);
  const long long total = (long long)B * H * W;
  long long blocks = (total + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(boundary_seed_kernel, dim3((unsigned)blocks), dim3(256), 0, s,
                     reinterpret_cast<const long long*>(labels), dist, B, H, W);
  const size_t sh = (size_t)2 * (W + 2) * sizeof(int);
  const long long* lp = reinterpret_cast<const long long*>(labels);
  const int ept = (W + NT - 1) / NT;
  if (ept <= 1)      hipLaunchKernelGGL(boundary_sweep_kernel<1>, dim3((unsigned)B), dim3(NT), sh, s, lp, dist, img_std, H, W, num_classes);
  else if (ept <= 2) hipLaunchKernelGGL(boundary_sweep_kernel<2>, dim3((unsigned)B), dim3(NT), sh, s, lp, dist, img_std, H, W, num_classes);
  else if (ept <= 4) hipLaunchKernelGGL(boundary_sweep_kernel<4>, dim3((unsigned)B), dim3(NT), sh, s, lp, dist, img_std, H, W, num_classes);
  else               hipLaunchKernelGGL(boundary_sweep_kernel<8>, dim3((unsigned)B), dim3(NT), sh, s, lp, dist, img_std, H, W, num_classes);
  hipLaunchKernelGGL(boundary_weight_kernel, dim3((unsigned)blocks), dim3(256), 0, s,
                     reinterpret_cast<const long long*>(labels), dist, img_std, weight, (long long)H * W, total, num_classes,
                     (long long)ignore_id);
  DCS_LAUNCH_RET();
}
