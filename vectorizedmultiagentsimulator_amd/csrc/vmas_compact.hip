// vmas_compact.hip - translation unit of the lane-compacted step kernel (vmas_compact.h) and its launcher.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "vmas_step_types.h"
#define VMAS_COMPACT_KERNELS 1
#include "vmas_compact.h"

namespace vmas {
int host_fail(const char* msg);  // vmas_hip.hip: sets vmas_last_error(), returns -1
}

template <int ENV, class EnvArgs, int OWN, int PLAIN>
static int launch_plain(size_t lds, int device, dim3 grid, dim3 block, hipStream_t s, const DevWorld& W, const compact::DevCompact& P,
                        float* state, float* aft, long ld, int batch, int padded, const DevStepArgs& a, const EnvArgs& env) {
  if (lds > 64 * 1024) {  // opt in to the large LDS once per device and instantiation
    static std::atomic<size_t> set_for_dev[64];
    std::atomic<size_t>& set_for = set_for_dev[device & 63];
    if (set_for.load() < lds) {
      if (hipFuncSetAttribute((const void*)compact::step_kernel_compact<ENV, EnvArgs, OWN, PLAIN>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return vmas::host_fail("vmas_world_step: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed for the compacted kernel");
      set_for = lds;
    }
  }
  hipLaunchKernelGGL((compact::step_kernel_compact<ENV, EnvArgs, OWN, PLAIN>), grid, block, lds, s, W, P, state, aft, ld, batch,
                     padded, a, env);
  if (hipGetLastError() != hipSuccess) return vmas::host_fail("vmas_world_step: launch of the compacted step kernel failed");
  return 0;
}

template <int ENV, class EnvArgs, int OWN>
static int launch_one(size_t lds, int device, dim3 grid, dim3 block, hipStream_t s, const DevWorld& W, const compact::DevCompact& P,
                      float* state, float* aft, long ld, int batch, int padded, const DevStepArgs& a, const EnvArgs& env) {
  // PLAIN: none of the optional inputs (their tests and pointers are compiled out of that instantiation)
  const bool plain = !a.pair_mask && !a.sync && !a.entity_gravity && a.first_substep == 0 && a.n_substeps <= 0;
  // (... and the lazy exact broad phase a variant of its own: PLAIN 1 without it, 2 with - vmas_compact.h)
  if (plain && a.lz.slots == nullptr) return launch_plain<ENV, EnvArgs, OWN, 1>(lds, device, grid, block, s, W, P, state, aft, ld, batch, padded, a, env);
  if (plain) return launch_plain<ENV, EnvArgs, OWN, 2>(lds, device, grid, block, s, W, P, state, aft, ld, batch, padded, a, env);
  if (a.lz.slots != nullptr)
    return vmas::host_fail("vmas_world_step: the compacted kernel runs the lazy exact broad phase without other per-call options only");
  return launch_plain<ENV, EnvArgs, OWN, 0>(lds, device, grid, block, s, W, P, state, aft, ld, batch, padded, a, env);
}

int vmas_compact_fill_trig(const compact::DevCompact& P, const float* state, long ld, float4* cache, hipStream_t s) {
  const unsigned long long static_lines = P.line_mask & ~P.dyn_mask;
  if (!static_lines) return 0;
  hipLaunchKernelGGL(compact::compact_trig_kernel, dim3(1), dim3(64), 0, s, static_lines, state, ld, cache);
  if (hipGetLastError() != hipSuccess) return vmas::host_fail("vmas_world_step: launch of compact_trig_kernel failed");
  return 0;
}

template <int ENV, class EnvArgs>
static int launch_own(int own, size_t lds, int device, dim3 grid, dim3 block, hipStream_t s, const DevWorld& W,
                      const compact::DevCompact& P, float* state, float* aft, long ld, int batch, int padded, const DevStepArgs& a,
                      const EnvArgs& env) {
  if (own == 1) return launch_one<ENV, EnvArgs, 1>(lds, device, grid, block, s, W, P, state, aft, ld, batch, padded, a, env);
  if (own == 2) return launch_one<ENV, EnvArgs, 2>(lds, device, grid, block, s, W, P, state, aft, ld, batch, padded, a, env);
  return launch_one<ENV, EnvArgs, 4>(lds, device, grid, block, s, W, P, state, aft, ld, batch, padded, a, env);
}

int vmas_compact_launch(int env_kind, int own, int nw, size_t lds, int device, const DevWorld& W, const compact::DevCompact& P,
                        float* state, float* agent_ft, long ld, int batch, int padded, const DevStepArgs& args, const DevEnv* env,
                        hipStream_t s) {
  const dim3 grid((batch + TILE - 1) / TILE), block(TILE * nw);
  if (env_kind == ENV_NONE) return launch_own<ENV_NONE>(own, lds, device, grid, block, s, W, P, state, agent_ft, ld, batch, padded, args, NoEnv{});
  if (!env) return vmas::host_fail("vmas_compact_launch: null environment arguments");
  if (env_kind == ENV_INGEST) return launch_own<ENV_INGEST>(own, lds, device, grid, block, s, W, P, state, agent_ft, ld, batch, padded, args, *env);
  if (env_kind == ENV_FOOTBALL) return launch_own<ENV_FOOTBALL>(own, lds, device, grid, block, s, W, P, state, agent_ft, ld, batch, padded, args, *env);
  return vmas::host_fail("vmas_compact_launch: this environment stage has no compacted form");
}
