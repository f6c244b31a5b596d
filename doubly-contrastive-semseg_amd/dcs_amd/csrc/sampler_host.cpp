// Host half of the hard-anchor sampler (utils/loss.py:264-337 of the reference) in C++: no device code.
//
// The reference draws, per kept (image, class), torch.randperm(num_hard) and torch.randperm(num_easy) from the default CPU
// generator and keeps the first one or two entries of each.  A CPU randperm(n) is a forward Fisher-Yates shuffle driven by
// n - 1 draws of the 32-bit mt19937 (z_i = draw_i % (n - i), swap(r[i], r[i + z_i]), i = 0 .. n-2) -- checked against
// torch.randperm for n up to 2 * 10^6 including the generator state afterwards (tests/test_host_logic_cpu.py).  So the first
// k entries of the permutation need only the first k draws, and the remaining n - 1 - k draws only ADVANCE the generator.
// At BASELINE config 3 a step shuffles 2.1 million elements to keep ~600 of them: 6-7 ms of host time in torch, of which the
// device waits ~4 ms (0.46 ms here).  Here the kept entries come from a sparse Fisher-Yates prefix and the generator is advanced by
// regenerating its state blocks without producing (tempering, reducing, swapping) the skipped outputs.
#include <cstdint>
#include <cstring>

#include "dcs_hip.h"

#define DCS_CHECK_ARG(cond) do { if (!(cond)) return DCS_E_ARG; } while (0)

namespace {

constexpr int MT_N = 624, MT_M = 397;
constexpr uint32_t MATRIX_A = 0x9908b0dfu, UPPER = 0x80000000u, LOWER = 0x7fffffffu;

struct Mt {
  uint32_t s[MT_N];
  int pos;                               // index of the next word; MT_N = block exhausted

  static inline uint32_t twist(uint32_t u, uint32_t v) {
    const uint32_t y = (u & UPPER) | (v & LOWER);
    return (y >> 1) ^ ((v & 1u) ? MATRIX_A : 0u);
  }
  // one state regeneration; three loops without loop-carried dependences (the compiler vectorises them)
  void regen() {
    uint32_t n[MT_N];
    const uint32_t* __restrict o = s;
    uint32_t* __restrict w = n;
    for (int i = 0; i < MT_N - MT_M; ++i) w[i] = o[i + MT_M] ^ twist(o[i], o[i + 1]);                 // 0 .. 226
    for (int i = MT_N - MT_M; i < 2 * (MT_N - MT_M); ++i) w[i] = w[i - (MT_N - MT_M)] ^ twist(o[i], o[i + 1]);   // 227 .. 453
    for (int i = 2 * (MT_N - MT_M); i < MT_N - 1; ++i) w[i] = w[i - (MT_N - MT_M)] ^ twist(o[i], o[i + 1]);      // 454 .. 622
    w[MT_N - 1] = w[MT_M - 1] ^ twist(o[MT_N - 1], w[0]);
    std::memcpy(s, n, sizeof(n));
    pos = 0;
  }
  inline uint32_t next() {
    if (pos >= MT_N) regen();
    uint32_t y = s[pos++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
  }
  void discard(long long m) {
    while (m > 0) {
      if (pos >= MT_N) regen();
      const long long take = m < MT_N - pos ? m : MT_N - pos;
      pos += (int)take;
      m -= take;
    }
  }
};

// first k entries of torch.randperm(n) (0 <= k <= 8), generator advanced by all n - 1 draws
inline int randperm_prefix(Mt& g, long long n, int k, long long* out) {
  long long idx[16], val[16];            // positions touched so far and their current values
  int used = 0;
  auto get = [&](long long p) { for (int e = 0; e < used; ++e) if (idx[e] == p) return val[e]; return p; };
  auto put = [&](long long p, long long v) {
    for (int e = 0; e < used; ++e) if (idx[e] == p) { val[e] = v; return; }
    idx[used] = p; val[used] = v; ++used;
  };
  const long long draws = n > 1 ? n - 1 : 0;
  long long i = 0;
  for (; i < k && i < n; ++i) {
    long long j = i;
    if (i < draws) j = i + (long long)(g.next() % (uint32_t)(n - i));
    const long long vi = get(i), vj = get(j);
    put(i, vj); put(j, vi);
    out[i] = vj;
  }
  const long long consumed = i < draws ? i : draws;
  g.discard(draws - consumed);
  return (int)i;                         // min(k, n) entries, like perm[:k]
}

}  // namespace

extern "C" int dcs_sampler_plan(const int64_t* counts, int B, int C, int max_samples, int max_views, uint32_t* mt_state,
                                int* mt_pos, int32_t* req, int32_t* cls, int32_t* img, int* n_view_out) {
  DCS_CHECK_ARG(counts && mt_state && mt_pos && req && cls && img && n_view_out && B > 0 && C > 0 && max_samples > 0 &&
                max_views > 0 && max_views <= 8 && *mt_pos >= 0 && *mt_pos <= MT_N);
  long long total = 0;
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c) {
      const int64_t nh = counts[((long long)b * C + c) * 2], ne = counts[((long long)b * C + c) * 2 + 1];
      DCS_CHECK_ARG(nh >= 0 && ne >= 0 && nh < 0xFFFFFFFFll && ne < 0xFFFFFFFFll);     // 32-bit draws (n < 2^32)
      if (nh + ne > max_views) ++total;
    }
  *n_view_out = 0;
  if (total == 0) return 0;
  const int n_view = (int)((max_samples / total) < max_views ? (max_samples / total) : max_views);
  if (n_view < 1) return DCS_E_UNSUPPORTED;            // more classes than samples: the caller's general path decides
  Mt g;
  std::memcpy(g.s, mt_state, sizeof(g.s));
  g.pos = *mt_pos;
  int t = 0, r = 0;
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c) {
      const int64_t nh = counts[((long long)b * C + c) * 2], ne = counts[((long long)b * C + c) * 2 + 1];
      if (nh + ne <= max_views) continue;
      int64_t hk, ek;                                  // utils/loss.py:303-316
      if (2 * nh >= n_view && 2 * ne >= n_view) { hk = n_view / 2; ek = n_view - hk; }      // keeps are in [0, n_view]
      else if (2 * nh >= n_view) { ek = ne; hk = n_view - ek; }
      else if (2 * ne >= n_view) { hk = nh; ek = n_view - hk; }
      else return DCS_E_UNSUPPORTED;                   // the reference raises here: the caller's general path does too
      long long pre[8];
      int got = randperm_prefix(g, nh, (int)hk, pre);
      if (got != hk) return DCS_E_UNSUPPORTED;         // fewer pixels than views to keep (max_views > 2 only): general path
      for (int e = 0; e < got; ++e, ++r) { req[3 * r] = b; req[3 * r + 1] = 2 * c; req[3 * r + 2] = (int32_t)pre[e]; }
      got = randperm_prefix(g, ne, (int)ek, pre);
      if (got != ek) return DCS_E_UNSUPPORTED;
      for (int e = 0; e < got; ++e, ++r) { req[3 * r] = b; req[3 * r + 1] = 2 * c + 1; req[3 * r + 2] = (int32_t)pre[e]; }
      cls[t] = c; img[t] = b; ++t;
    }
  std::memcpy(mt_state, g.s, sizeof(g.s));             // the generator moves only when the whole plan succeeded
  *mt_pos = g.pos;
  *n_view_out = n_view;
  return t;
}
