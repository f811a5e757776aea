// truss_emu.cpp -- CPU lane emulator of the HIP step kernel.  *** TEST INFRASTRUCTURE ONLY ***
//
// Builds the same C ABI (include/truss_mi355.h) from the same lane program
// (mop-truss-marl_amd/csrc/truss_body.h) and the same host code (truss_host.h), but runs every
// phase for the 64 lanes of a workgroup one after the other on the CPU.  It exists so that the
// kernel's indexing (band offsets, window rotation, pair/symmetry tables) can be debugged in a
// container without a GPU.  The product (mop-truss-marl_amd/truss_mi355) never loads this library;
// truss_backend() returns "emu" so tests can tell the two apart.
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

struct float4 {
  float x, y, z, w;
};
#define TRUSS_HD inline
#define TRUSS_UNROLL
#define TB_STREAM_STORE(p, v) (*(p) = (v))
static inline void tb_lds_add(double *p, double v) { *p += v; }
static inline double tb_rcp(double d) { return 1.0 / d; }
// DPP row broadcast stand-in: the value lane `src` of this lane's team computed in the previous phase
template <class LN>
static inline double tb_team_bcast(LN &ln, int src) { return ln.peers[ln.lane - ln.gs + src].bx; }
static inline double tb_rsqrt(double x) { return 1.0 / sqrt(x); }

#include "../../mop-truss-marl_amd/csrc/truss_body.h"

#define TRUSS_BACKEND_NAME "emu"
struct truss_topo;
static void *tb_dev_alloc(size_t n) { return malloc(n); }
static void tb_dev_free(void *p) { free(p); }
static bool tb_dev_upload(void *dst, const void *src, size_t n) {
  memcpy(dst, src, n);
  return true;
}
static int tb_launch_step(const truss_topo *t, const StepArgsDev &A, void *stream);
static int tb_launch_obs(const truss_topo *t, const ObsArgsDev &A, void *stream);

#include "../../mop-truss-marl_amd/csrc/truss_host.h"

template <int G, int WL, int RPL, int EPL>
static void emu_run(const truss_topo *t, const StepArgsDev &A) {
  using Lane = StepLane<G, WL, RPL, EPL>;
  constexpr int W_ = Lane::W;
  const TopoDev &T = t->dev;
  const int nblocks = (A.B + Lane::EPB - 1) / Lane::EPB;
  std::vector<char> lds(t->lds_bytes);
  std::vector<Lane> lanes(64);
  for (int b = 0; b < nblocks; ++b) {
    memset(lds.data(), 0xA5, lds.size());  // poison: catches reads of uninitialised LDS
    for (int l = 0; l < 64; ++l) {
      lanes[l].init(l, b, T, A, lds.data());
      lanes[l].peers = lanes.data();
    }
#define PH(call) \
  for (auto &ln : lanes) ln.call
#define PH_NS(call) \
  for (auto &ln : lanes) ln.call
#define BAR() (void)0
    TRUSS_STEP_SCHEDULE(PH, PH_NS, BAR, T, A)
#undef PH
#undef PH_NS
#undef BAR
  }
}

static int tb_launch_step(const truss_topo *t, const StepArgsDev &A, void *) {
  const TbVariant &v = kVariants[t->variant];
#define X(g, wl, r, e) \
  if (v.G == g && v.WL == wl && v.RPL == r && v.EPL == e) { emu_run<g, wl, r, e>(t, A); return TRUSS_OK; }
  TRUSS_VARIANTS(X)
#undef X
  return tb_fail(TRUSS_EUNSUPPORTED, "variant not compiled into the emulator");
}

static int tb_launch_obs(const truss_topo *t, const ObsArgsDev &A, void *) {
  const TopoDev &T = t->dev;
  std::vector<char> lds(tb_obs_lds_bytes(t->N));
  std::vector<ObsLane> lanes(64);
  for (int b = 0; b < A.B; ++b) {
    memset(lds.data(), 0xA5, lds.size());
    for (int l = 0; l < 64; ++l) lanes[l].init(l, b, T, A, lds.data());
#define PH(call) \
  for (auto &ln : lanes) ln.call
    TRUSS_OBS_SCHEDULE(PH, PH, T, A)
#undef PH
  }
  return TRUSS_OK;
}
