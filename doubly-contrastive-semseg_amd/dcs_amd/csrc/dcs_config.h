// Library-wide switches of libdcs_hip.so (host side; no HIP types).
#pragma once
// Library-wide switches (A/B experiments and test hooks).  Read from the environment ONCE, when the library is loaded
// (sampler_host.cpp), never per launch; tests and tools change them through dcs_set_option (include/dcs_hip.h).
struct DcsConfig {
  int bn_nt;          // DCS_BN_NT       (1)   streaming (non-temporal) accesses for large tensors; 0 = never
  int nt_min_mb;      // DCS_NT_MIN_MB   (256) ... from this many MiB on
  int x3_bm128;       // DCS_X3_BM128    (0)   1 = no 256-pixel tiles for the 64-wide split-bf16 gather
  int x3_halo;        // DCS_X3_HALO     (1)   0 = per-tap kernel only, 2 = halo kernel whenever the geometry allows
  int wgrad_roll;     // DCS_WGRAD_ROLL  (1)   0 = nine-tap weight gradient instead of the rolling-window kernel
  int conv_bk16;      // DCS_CONV_BK16   (0)   1 = 16-channel chunks everywhere in the fp32 gather
  int wgrad_ch32;     // DCS_WGRAD_CH32  (0)   1 = 32-pixel chunks in the generic fp32 weight gradient
  int contrast_fused; // DCS_CONTRAST_FUSED (1) 0 = two launches for the small similarity family (A/B, tests)
  int x3w_db;         // DCS_X3W_DB      (1)   0 = single-buffered halo in the fp16 3x3 kernels (A/B, tests)
};
extern DcsConfig g_dcs_config;
static inline const DcsConfig& dcs_config() { return g_dcs_config; }
