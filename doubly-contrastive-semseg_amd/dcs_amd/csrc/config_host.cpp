// The library's switches: initialised from the environment ONCE at load time (static initialiser), changed afterwards only
// through dcs_set_option (tests, tools).  No kernel launcher reads the environment.
#include <cstdlib>
#include <cstring>

#include "dcs_config.h"
#include "dcs_hip.h"

namespace {
int env_int(const char* name, int dflt) {
  const char* e = std::getenv(name);
  return (e && *e) ? std::atoi(e) : dflt;
}
DcsConfig from_env() {
  DcsConfig c;
  c.bn_nt = env_int("DCS_BN_NT", 1);
  c.nt_min_mb = env_int("DCS_NT_MIN_MB", 256);
  c.x3_bm128 = std::getenv("DCS_X3_BM128") ? 1 : 0;
  c.x3_halo = env_int("DCS_X3_HALO", 1);
  c.wgrad_roll = env_int("DCS_WGRAD_ROLL", 1);
  c.conv_bk16 = std::getenv("DCS_CONV_BK16") ? 1 : 0;
  c.wgrad_ch32 = std::getenv("DCS_WGRAD_CH32") ? 1 : 0;
  c.contrast_fused = env_int("DCS_CONTRAST_FUSED", 1);
  c.x3w_db = env_int("DCS_X3W_DB", 1);
  return c;
}
struct Entry { const char* name; int DcsConfig::*field; };
const Entry kEntries[] = {
    {"bn_nt", &DcsConfig::bn_nt}, {"nt_min_mb", &DcsConfig::nt_min_mb}, {"x3_bm128", &DcsConfig::x3_bm128},
    {"x3_halo", &DcsConfig::x3_halo}, {"wgrad_roll", &DcsConfig::wgrad_roll}, {"conv_bk16", &DcsConfig::conv_bk16},
    {"wgrad_ch32", &DcsConfig::wgrad_ch32}, {"contrast_fused", &DcsConfig::contrast_fused}, {"x3w_db", &DcsConfig::x3w_db},
};
}  // namespace

DcsConfig g_dcs_config = from_env();

extern "C" int dcs_set_option(const char* name, int value) {
  if (!name) return DCS_E_ARG;
  for (const Entry& e : kEntries)
    if (std::strcmp(name, e.name) == 0) { g_dcs_config.*(e.field) = value; return DCS_OK; }
  return DCS_E_ARG;
}

extern "C" int dcs_get_option(const char* name, int* value) {
  if (!name || !value) return DCS_E_ARG;
  for (const Entry& e : kEntries)
    if (std::strcmp(name, e.name) == 0) { *value = g_dcs_config.*(e.field); return DCS_OK; }
  return DCS_E_ARG;
}
