// vmas_hip.hip - MI355X (gfx950 / CDNA4) implementation of the C ABI in include/vmas_hip.h.
//
// One fused kernel advances ALL substeps of World.step() (core.py:1972-2015) for a tile
// of 64 environments:
//
//   HBM (SoA planes, env fastest) --coalesced 256-B rows--> LDS tile [row][64 envs]
//   once      : trig of every Line/Box entity (which ones: masks in the kernel arguments; ONE barrier)
//   per substep, out of LDS, no HBM traffic:
//     B  gather : for every dynamic entity, its prologue force (action/friction/gravity)
//                 plus the force of every incident joint/pair whose OTHER side is static,
//                 accumulated in the reference's order in registers -> a partial-sum row;
//                 every joint/pair of TWO dynamic entities evaluated once -> its own rows
//     C  integrate : partial rows + the entity's side of its shared pairs, in order, then
//                 semi-implicit Euler + clamps, new trig for the next substep
//   LDS tile --coalesced rows (dynamic planes only)--> HBM
//
// Work decomposition ("one wavefront lane per environment, W wavefronts per tile"):
//   * lane l of every wave of a block is environment l of the block's 64-env tile, so
//     EVERYTHING else - which entity, which pair, which shape code, every branch - is
//     wave-uniform: there is no divergence, and LDS rows are read 64 consecutive floats
//     at a time (conflict-free).  Descriptors come from one blob staged into LDS.
//   * the W waves of a block split the tile's work by ITEMS: each dynamic entity's own item
//     list (the reference's accumulation order, core.py:2176-2189) is cut into segments,
//     the shared pairs into runs; both are sorted heaviest first, every wave starts with the
//     one of its own index and pulls further ones from an LDS counter.  Results land in
//     per-segment / per-pair rows, the entity's owner adds them in a fixed order: no
//     atomics, bitwise deterministic whichever wave computed what.
//   * per-environment conservative broad phase (bounding circle, or oriented-box distance
//     for boxes) in front of every narrow phase; a wave skips an item when none of its 64
//     environments needs it; the tests are NaN/inf-aware, so non-finite poses poison their
//     pairs exactly as the reference's arithmetic does.
//   * the kernel is compiled in code LEVELs (which pair types exist), ENV stages (action
//     ingest prologue / scenario epilogue) and PLAIN specialisations (no optional inputs,
//     padded planes, one substep, whole tiles): it is a dependent chain per wave at the
//     benchmark sizes, and every scalar spill, uniform branch and predication block is on it.
//   * waves per tile and which pairs are shared are chosen per world and batch by the
//     number of waves the choice keeps running per CU (select_config).
//
// No MFMA: fp32 elementwise/transcendental work on 2-vectors; nothing is a contraction
// (BASELINE.json north_star).  The roofline that bounds it is HBM (384 B per env-step for
// `balance`), see DESIGN.md.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <map>
#include <numeric>
#include <vector>

#include "../../include/vmas_hip.h"
#include "../../include/vmas_debug_hip.h"
#include "vmas_step_types.h"

#include "vmas_step_device.h"

// ------------------------------------------------------------------------------------
// the fused step kernel: grid = ceil(batch / 64) tiles, block = 64 x W threads
// ------------------------------------------------------------------------------------
// PLAIN: the launch has none of the optional inputs (recorded pair mask, per-environment joint rotations / entity gravity,
// a partial substep range) - their tests, and the scalar registers that would carry the pointers
// through every loop of a kernel that is short of them, are compiled out.
// The same holds for: planes padded to whole tiles (ld >= 64 * tiles: no per-load "is this lane a live environment"
// predication - only the stores of the last tile are masked), an item list that lives in the LDS blob, (PLAIN >= 2) a world with one substep per step (the between-substeps write-back to
// LDS and its trig are dead code) and (PLAIN == 3) a batch of whole tiles (no store predication either).
template <int LEVEL, int ENV, class EnvArgs, int PLAIN>
__global__ __launch_bounds__(TILE*(LEVEL >= 2 ? 8 : MAX_WAVES)) void step_kernel(DevWorld W_in, float* __restrict__ state,
                                                               float* __restrict__ agent_ft, long ld, int batch,
                                                               DevStepArgs args_in, const EnvArgs E) {
  if constexpr (ENV != ENV_NONE) {  // a gated launch behind a validation that raised flags: not a single load or store
    if (env_gate_closed(E)) return;
  }
  DevStepArgs args = args_in;
  DevWorld W = W_in;
  args.lz_x = args.lz_s = nullptr; args.lz_g = nullptr; args.lz_acc = nullptr;  // (the lazy form's LDS words: set per pass below)
  if constexpr (PLAIN != 0) {
    args.pair_mask = nullptr; args.joint_fixed_rot = nullptr; args.entity_gravity = nullptr;
    args.first_substep = 0; args.n_substeps = 0; args.sync = nullptr;
    W.items_in_lds = 1;
    if constexpr (PLAIN >= 2) { W.substeps = 1; args.n_steps = 1; args.ft_stride = 0; }  // (PLAIN == 1 also serves rollouts)
  }
  extern __shared__ float lds[];
  const int lane = threadIdx.x & (TILE - 1);
  const int wv = sgpr(threadIdx.x >> 6);
  const int nw = sgpr(blockDim.x >> 6);
  const int nE = W.nE, nA = W.nA;
  const long env = (long)blockIdx.x * TILE + lane;
  const bool live = PLAIN == 3 || env < batch;  // guards every store and every load from a buffer that is not padded to ld
                                          // (PLAIN == 3: the batch is a whole number of tiles, every lane is an environment)
  const bool lv = PLAIN != 0 || live;     // loads from state / agent_ft: PLAIN launches have ld >= 64 * tiles, so the tail
                                          // lanes of the last tile read the padding columns (harmless) without predication
#ifdef VMAS_TRACE  // profiling build only (scripts/trace_phases.py): per-wave s_memtime stamps
#define STAMP(k)                                                                                     \
  if (args.trace && lane == 0) args.trace[((long)blockIdx.x * 16 + wv) * 16 + (k)] = __builtin_amdgcn_s_memtime()
#define TNOW() (args.trace ? __builtin_amdgcn_s_memtime() : 0ull)
#else
#define STAMP(k)
#define TNOW() 0ull
#endif
  STAMP(0);
  float* tile = lds + lane;  // this lane's column: row r is tile[r * ROWF]
  uint32_t* blob = (uint32_t*)(lds + W.off_blob);
  int* ctr = (int*)(blob + W.blob_words);  // [4] work counters: (substep parity) x (gather, integrate)
  uint32_t* fired_words = (uint32_t*)(ctr + 4);  // [2 parities][2] which shared sphere-sphere pairs fired (bit 4 * record + k)
  constexpr int EW = (int)(sizeof(DevEntity) / 4), SW = (int)(sizeof(DevSegment) / 4), OW = (int)(sizeof(DevOwned) / 4),
                IW = (int)(sizeof(DevItem) / 4);

  // ---- issue this wave's first entity rows and agent-force rows, THEN stage the descriptor
  //      blob: the two HBM latencies overlap instead of adding up
  float v0[6], f0[3];
  if (wv < nE) {
    const float* src = state + (long)wv * 6 * ld + env;
#pragma unroll
    for (int f = 0; f < 6; ++f) v0[f] = lv ? src[f * ld] : 0.f;
  }
  // Environment._set_action + process_action as the prologue: the agent's force rows are computed
  // from its action tensor (and stored to agent_ft, where scenario code reads agent.state.force)
  long act_row0 = 0;  // multi-step rollouts (vmas_world_rollout_env): first row of the running step's actions
  auto load_agent_ft = [&](int a, float* f3) {
    const float* src = agent_ft + (long)a * 3 * ld + env;
    if constexpr (ENV != ENV_NONE) {
      const bool on = E.has_ingest && !(ABLATE(E) & 8);
      const VmasActionSlot& S = E.ingest.agents[a];
      if (on && (S.action != nullptr || S.action_index != nullptr)) {
        uint32_t bad = 0;
        ingest_slot(S, E.ingest.clamp, env, live, agent_ft, ld, f3, bad, act_row0);
        if (S.action_size < 3) f3[2] = lv ? src[2 * ld] : 0.f;  // Holonomic leaves the torque alone
        if (E.err_flags != nullptr && bad != 0) raise_action_error(E.err_flags, bad);
        return;
      }
      if (on && E.ingest.n_scripts > 0 && E.script_of_agent[a] >= 0) {  // scripted: driven by the state about to be stepped
        const VmasAgentScript& SC = E.ingest.scripts[E.script_of_agent[a]];
        run_script(SC, state + (long)SC.entity * 6 * ld + env, ld, env, live, agent_ft, ld, f3);
        f3[2] = lv ? src[2 * ld] : 0.f;
        return;
      }
    }
#pragma unroll
    for (int f = 0; f < 3; ++f) f3[f] = lv ? src[f * ld] : 0.f;
  };
  if (wv < nA) load_agent_ft(wv, f0);
  // the epilogue's HBM inputs are requested now, behind the physics
  [[maybe_unused]] float post_prev = 0.f, post_steps = 0.f;
  if constexpr (ENV == ENV_BALANCE) {
    post_prev = live ? E.balance.o.global_shaping[env] : 0.f;
    if (wv == 0) post_steps = (E.balance.o.limit.steps != nullptr && live) ? E.balance.o.limit.steps[env] : 0.f;
  }
  if constexpr (ENV == ENV_TRANSPORT) {
    float* term = lds + E.scratch_off + lane;
    for (int p = wv; p < E.transport.d.n_packages; p += nw)
      term[p * 64] = live ? E.transport.o.global_shaping[(long)p * batch + env] : 0.f;
    if (wv == 0) post_steps = (E.transport.o.limit.steps != nullptr && live) ? E.transport.o.limit.steps[env] : 0.f;
  }
  if constexpr (ENV == ENV_NAVIGATION) {
    const TileCtx C(batch);
    navigation_prologue_tile(C, E.navigation.d, E.navigation.o, E.navigation.w, batch, lds + E.scratch_off);  // (published by the load barrier)
    if (wv == 0) post_steps = load_steps(E.navigation.o.limit, C);
  }
  {  // the descriptor blob, 16 bytes per thread and up to four requests in flight before the first LDS write (a plain
     // copy loop waits for every load before it issues the next: one full HBM latency per iteration)
    const uint4* src = (const uint4*)W.blob;
    uint4* dst = (uint4*)blob;
    const int n4 = W.blob_words >> 2;  // (the host pads the blob to a multiple of four words)
    const int nt = blockDim.x;
    for (int i0 = threadIdx.x; i0 < n4; i0 += 4 * nt) {
      const int i1 = i0 + nt, i2 = i0 + 2 * nt, i3 = i0 + 3 * nt;
      const uint4 a = src[i0], b = src[i1 < n4 ? i1 : i0], c = src[i2 < n4 ? i2 : i0], d = src[i3 < n4 ? i3 : i0];
      dst[i0] = a;
      if (i1 < n4) dst[i1] = b;
      if (i2 < n4) dst[i2] = c;
      if (i3 < n4) dst[i3] = d;
    }
  }
  const int first_dyn = (ABLATE(args) & 128) ? 0 : nw;  // 128: profiling toggle, fully dynamic
  if (threadIdx.x < 4) ctr[threadIdx.x] = first_dyn;  // the first unit of every wave is static (its own index)
  if (threadIdx.x >= 4 && threadIdx.x < 8) ctr[threadIdx.x] = 0;  // fired bits: set by eval_ssp (atomic or)
  uint32_t* xmask = fired_words + 4;  // [mask_words] this tile's pair bits (in-kernel exact broad phase)
  if constexpr (PLAIN == 0) {
    if (args.sync != nullptr && (int)threadIdx.x < args.mask_words) xmask[threadIdx.x] = 0u;
  }
  // the LAZY exact broad phase (vmas_env_device.h): [2 parities][overlap words | band words] | need | batch | flag, the same
  // LDS as the barrier form's two arrays (a launch runs one form or the other)
  const bool lazy = args.lz.slots != nullptr;
  const int lzp = lazy ? ((args.lz.words + 3) & ~3) : 0;
  uint32_t* lz_words = xmask;
  if (lazy) {
    for (int i = threadIdx.x; i < 6 * lzp + 4; i += blockDim.x) lz_words[i] = 0u;
  }

  // ---- HBM -> LDS, one entity per wave at a time: six 256-byte row reads in flight, then the
  //      entity's trig straight from the registers
  for (int e = wv; e < nE; e += nw) {
    float v[6];
    if (e == wv) {
#pragma unroll
      for (int f = 0; f < 6; ++f) v[f] = v0[f];
    } else {
      const float* src = state + (long)e * 6 * ld + env;
#pragma unroll
      for (int f = 0; f < 6; ++f) v[f] = lv ? src[f * ld] : 0.f;
    }
    float* dst = tile + e * 6 * ROWF;
#pragma unroll
    for (int f = 0; f < 6; ++f) dst[f * ROWF] = v[f];
    // the reference lets a non-finite pose poison every pair it is in, however far apart
    // (cos(inf) = NaN): such environments must not use the distance skip
    if (W.trig_in_args && ((W.trig_mask >> e) & 1ull) && !(ABLATE(args) & 8)) {
      const int tr_row = W.row_tr + 4 * __builtin_popcountll(W.trig_mask & ((1ull << e) - 1ull));
      write_trig(tile + tr_row * ROWF, v[4], ((W.box_mask >> e) & 1ull) ? VMAS_SHAPE_BOX : VMAS_SHAPE_LINE);
    }
  }
  if (wv < nA) {
    float* dst = tile + W.off_af + wv * 3 * ROWF;
#pragma unroll
    for (int f = 0; f < 3; ++f) dst[f * ROWF] = f0[f];
  }
  for (int a = wv + nw; a < nA; a += nw) {
    float f3[3];
    load_agent_ft(a, f3);
    float* dst = tile + W.off_af + a * 3 * ROWF;
#pragma unroll
    for (int f = 0; f < 3; ++f) dst[f * ROWF] = f3[f];
  }
  STAMP(1);
  __syncthreads();
  if (!W.trig_in_args) {  // (more than 64 entities: shapes and offsets from the blob, rotations from the tile)
    for (int e = wv; e < nE && !(ABLATE(args) & 8); e += nw) {
      const int shape = sgpr((int)blob[W.b_ent + e * EW + 1]), tr_off = sgpr((int)blob[W.b_ent + e * EW + 3]);
      if (tr_off >= 0) write_trig(tile + tr_off, tile[(e * 6 + 4) * ROWF], shape);
    }
    __syncthreads();
  }
  STAMP(2);

  const float sub_dt = W.sub_dt;
  const int s_begin = args.first_substep;
  const int s_end = s_begin + (args.n_substeps > 0 ? args.n_substeps : W.substeps - s_begin);
  // dynamic work distribution: waves pull the next segment / entity from an LDS counter
  // (segments are sorted heaviest first).  Results land in per-segment rows, so the force sum
  // stays deterministic no matter which wave computed what.
  auto grab = [&](int* c) {
    int v = 0;
    if (lane == 0) v = atomicAdd(c, 1);
    return sgpr(v);
  };

  const int n_steps = args.n_steps > 1 ? args.n_steps : 1;
#ifdef VMAS_TRACE
  unsigned long long acc_grab = 0, acc_pro = 0, acc_item[4] = {0, 0, 0, 0}, cnt_item[4] = {0, 0, 0, 0};
#endif
  int it = 0;  // running (step, substep) index: parity selects the live pair of work counters
  for (int stp = 0; stp < n_steps; ++stp) {
  float* aft = agent_ft + (long)stp * args.ft_stride;  // this step's agent forces
  if (stp > 0) {  // persistent rollout: only the agent forces (or the actions they are made of) come from HBM, the
                  // state never left LDS
    bool ingested = false;
    if constexpr (ENV != ENV_NONE) {
      if (E.has_ingest) {  // vmas_world_rollout_env: this step's rows of the action tensors through the ingest prologue
        ingested = true;
        act_row0 = (long)stp * batch;
        for (int a = wv; a < nA; a += nw) {
          float f3[3];
          load_agent_ft(a, f3);
          float* dst = tile + W.off_af + a * 3 * ROWF;
#pragma unroll
          for (int f = 0; f < 3; ++f) dst[f * ROWF] = f3[f];
        }
      }
    }
    if (!ingested)
      for (int a = wv; a < nA; a += nw) {
        const float* src = aft + (long)a * 3 * ld + env;
        float* dst = tile + W.off_af + a * 3 * ROWF;
#pragma unroll
        for (int f = 0; f < 3; ++f) dst[f * ROWF] = lv ? src[f * ld] : 0.f;
      }
    __syncthreads();
  }
  for (int substep = s_begin; substep < s_end; ++substep, ++it) {
    const bool last_sub = substep + 1 == s_end;
    const bool last = last_sub && stp + 1 == n_steps;  // write back to HBM instead of LDS
    int* c_gather = ctr + 2 * (it & 1);
    int* c_integrate = c_gather + 1;
    if (threadIdx.x < 2) ctr[2 * ((it + 1) & 1) + threadIdx.x] = first_dyn;  // re-arm the other parity
    if (threadIdx.x >= 2 && threadIdx.x < 4) fired_words[2 * ((it + 1) & 1) + threadIdx.x - 2] = 0u;
    uint32_t* fired = fired_words + 2 * (it & 1);
    if (lazy) {  // this pass's overlap / band words; the other parity's are re-armed (last read behind the previous gather)
      for (int i = threadIdx.x; i < 2 * lzp; i += blockDim.x) lz_words[2 * lzp * ((it + 1) & 1) + i] = 0u;
      args.lz_x = lz_words + 2 * lzp * (it & 1);
      args.lz_s = args.lz_x + lzp;
      args.lz_g = nullptr;
      lazy_overlap_blob(args, blob + W.b_pairs, W.n_pairs, tile, live, wv, nw, it);  // (+ a block barrier; the words go out)
    }
    if constexpr (PLAIN == 0) {
      if (args.sync != nullptr) {
        // ---- World.collides' batch-global bounding-circle test (core.py:2797-2801) for THIS substep, by the whole grid:
        // every tile ORs the pairs some environment of it overlaps into the substep's mask slot, all tiles meet at a
        // grid-wide barrier (the grid is at most one tile per CU, so every tile is resident), then read the mask.
        const uint32_t seq = args.seq0 + (uint32_t)it;
        for (int p = wv; p < args.n_mpairs; p += nw) {
          const DevMaskPair P = args.mpairs[p];
          const float* A = tile + sgpr(P.a) * 6 * ROWF;
          const float* B = tile + sgpr(P.b) * 6 * ROWF;
          const bool hit = live && norm2(A[0] - B[0], A[ROWF] - B[ROWF]) <= P.bound_sum;
          if (__any(hit) && lane == 0) atomicOr(&xmask[p >> 5], 1u << (p & 31));  // (LDS)
        }
        __syncthreads();
        // one atomic per pair word carries this tile's arrival and its bits; wave 0 collects the batch's words into LDS
        // (grid_bits_publish / grid_bits_collect, vmas_env_device.h): the items read the mask from there
        {
          const int groups = ((int)gridDim.x + 31) >> 5;
          unsigned long long* base = (unsigned long long*)(args.sync + 4);
          const size_t stride = (size_t)args.mask_words * groups;
          uint32_t* gmask = xmask + ((args.mask_words + 3) & ~3);
          grid_bits_publish(base + (seq & 3u) * stride, args.mask_words, xmask);
          if ((int)threadIdx.x < args.mask_words) xmask[threadIdx.x] = 0u;  // (re-armed for the next substep by the thread that published it)
          if (wv == 0)
            grid_bits_collect(base + (seq & 3u) * stride, base + ((seq + 2u) & 3u) * stride, args.mask_words, gmask, args.sync + 1,
                              args.gave_up);
          __syncthreads();
          args.pair_mask = gmask;  // (LDS)
        }
      }
    }
    // ================= phase B: gather forces per (entity, segment)
    // (lazy form: the optimistic pass with every pair on; made AGAIN, with the batch's words, by a tile that has an
    //  environment in the band of a pair no environment of the batch overlaps)
    for (;;) {
#ifdef VMAS_TRACE
    unsigned long long tg = TNOW();
#endif
    for (int si = first_dyn ? wv : grab(c_gather); si < W.n_segs; si = grab(c_gather)) {
      const uint4 s0 = ((const uint4*)(blob + W.b_segs + si * SW))[0], s1 = ((const uint4*)(blob + W.b_segs + si * SW))[1];
      const int e = sgpr((int)s0.x);
      const float* Es = tile + (int)s0.y;
      const int i0 = sgpr((int)s0.z), i1s = sgpr((int)s0.w), first = sgpr((int)s1.x);
      // item records: record i + 1 is fetched while record i is evaluated (LDS copy; the global copy of worlds whose item
      // list does not fit the LDS budget holds the same 16 words per item)
      auto item_words = [&](int ii) {
        return W.items_in_lds ? fetch_words(blob + W.b_items + ii * IW) : fetch_words((const uint32_t*)(W.items + ii));
      };
      const int i1 = (ABLATE(args) & 1) ? i0 : i1s;
      // MEASURED AND SWITCHED OFF (profiles/r02_item_prefetch.txt): fetching record i + 1 during record i costs 26 VGPRs
      // (90 -> 116 in the lean kernels: 5 -> 4 waves per SIMD); balance 32768 envs 10.1 -> 10.4 us on one queue and
      // 8.5 -> 10.0 us on two (the second queue's tiles no longer fit beside the first's), 1 M envs 226 -> 234 us.
      constexpr bool PREFETCH = false;
      ItemW cur;
      if (PREFETCH && i0 < i1) cur = item_words(i0);
      if (e < 0) {  // a run of SHARED pairs/joints (both entities dynamic): evaluated once, both owners read the rows in phase C
        for (int ii = i0; ii < i1; ++ii) {
          const ItemW I = PREFETCH ? cur : item_words(ii);
          if (PREFETCH && ii + 1 < i1) cur = item_words(ii + 1);
          if (sgpr((int)I.w0.x) == TASK_SSP) {
            eval_ssp(I, W, args, tile, fired, live);
            continue;
          }
          const ItemV K = load_item(I);
          v2 f = V(0.f, 0.f);
          float ta = 0.f, tb = 0.f;
          if (!(ABLATE(args) & 16)) eval_item<LEVEL>(K, W, args, tile, env, live, ld, f, ta, tb);
          float* R = tile + (K.side >> 2);
          R[0] = f.x; R[ROWF] = f.y; R[2 * ROWF] = ta;
          if (K.type != VMAS_PAIR_LS && K.type != VMAS_PAIR_BS) R[3 * ROWF] = tb;  // (b is a sphere: no torque row)
        }
#ifdef VMAS_TRACE
        tg = TNOW();
#endif
        continue;
      }
      float* P = tile + (int)s1.y;
      const uint32_t efl = (uint32_t)sgpr((int)s1.z);
      v2 F = V(0.f, 0.f);
      float Tq = 0.f;
#ifdef VMAS_TRACE
      { unsigned long long t1 = TNOW(); acc_grab += t1 - tg; tg = t1; }
#endif
      if (first && !(ABLATE(args) & 4)) {  // prologue core.py:1995-2004
        const EntV D = load_ent(blob + W.b_ent + e * EW);
        const uint32_t fl = D.flags;
        if (fl & VMAS_F_AGENT) {
          float* Af = tile + W.off_af + D.agent_index * 3 * ROWF;
          if (fl & VMAS_F_MOVABLE) {  // _apply_action_force core.py:2018-2028
            v2 f = V(Af[0], Af[ROWF]);
            if ((fl & (VMAS_F_MAX_F | VMAS_F_F_RANGE)) && args.lz_g == nullptr) {  // (a pass made again reads the clamped rows)
              if (fl & VMAS_F_MAX_F) f = clamp_with_norm(f, D.max_f);
              if (fl & VMAS_F_F_RANGE) f = V(clamp_t(f.x, D.f_range), clamp_t(f.y, D.f_range));
              Af[0] = f.x; Af[ROWF] = f.y;
              if (last_sub && live) {  // the clamped force is written back (core.py:2021-2027)
                float* gf = aft + (long)D.agent_index * 3 * ld + env;
                gf[0] = f.x; gf[ld] = f.y;
              }
            }
            F = F + f;
          }
          if (fl & VMAS_F_ROTATABLE) {  // _apply_action_torque core.py:2030-2041
            float t = Af[2 * ROWF];
            if ((fl & (VMAS_F_MAX_T | VMAS_F_T_RANGE)) && args.lz_g == nullptr) {
              if (fl & VMAS_F_MAX_T) {
                const float n = fabsf(t);
                const float nt = (t / n) * D.max_t;
                t = n > D.max_t ? nt : t;
              }
              if (fl & VMAS_F_T_RANGE) t = clamp_t(t, D.t_range);
              Af[2 * ROWF] = t;
              if (last_sub && live) aft[((long)D.agent_index * 3 + 2) * ld + env] = t;
            }
            Tq = Tq + t;
          }
        }
        // _apply_friction_force core.py:2054-2102
        if (fl & VMAS_F_LIN_FRICTION) F = F + friction2(V(Es[2 * ROWF], Es[3 * ROWF]), D.lin_friction, D.mass, sub_dt);
        if (fl & VMAS_F_ANG_FRICTION) Tq = Tq + friction1(Es[5 * ROWF], D.ang_friction, D.inertia, sub_dt);
        // _apply_gravity core.py:2043-2052
        if (fl & VMAS_F_MOVABLE) {
          if (W.has_gravity) F = F + V(D.mass * W.gx, D.mass * W.gy);
          if (fl & VMAS_F_GRAVITY) {
            v2 ge = V(D.gx, D.gy);
            if (args.entity_gravity && live) {
              const float* gp = args.entity_gravity + (long)e * 2 * ld + env;
              ge = V(gp[0], gp[ld]);
            }
            F = F + V(D.mass * ge.x, D.mass * ge.y);
          }
        }
      }
      // joints, then pairs, in the reference's accumulation order (core.py:2176-2199)
#ifdef VMAS_TRACE
      { unsigned long long t1 = TNOW(); acc_pro += t1 - tg; tg = t1; }
#endif
      for (int ii = i0; ii < i1; ++ii) {
        const ItemW I = PREFETCH ? cur : item_words(ii);
        if (PREFETCH && ii + 1 < i1) cur = item_words(ii + 1);
        const int packed_type = sgpr((int)I.w0.x);
        if (packed_type >= TASK_SSQ) {  // (packed records exist only in the LDS copy of the item list)
          if (!(ABLATE(args) & 16)) {
            if (LEVEL > 0 || packed_type == TASK_SSQ)  // (line-sphere records are only built for level-0 worlds)
              eval_ssq(I, W, args, tile, efl & VMAS_F_MOVABLE, F, live);
            else
              eval_lsq(I, W, args, tile, efl & VMAS_F_MOVABLE, F, live);
          }
          continue;
        }
        const ItemV K = load_item(I);
        v2 f = V(0.f, 0.f);
        float t = 0.f, t_unused = 0.f;
        if (!(ABLATE(args) & 16)) eval_item<LEVEL>(K, W, args, tile, env, live, ld, f, t, t_unused);
        else f.x = __int_as_float(K.type + K.oa + K.ob + K.index) * 1e-30f;  // descriptor fetch only (profiling)
        if (efl & VMAS_F_MOVABLE) F = F + f;
        if (efl & VMAS_F_ROTATABLE) Tq = Tq + t;
#ifdef VMAS_TRACE
        if (args.trace) {
          unsigned long long t1 = TNOW();
          const int cls = K.type == VMAS_PAIR_SS ? 0 : (K.type == VMAS_PAIR_LS ? 1 : (K.type == VMAS_PAIR_BS ? 2 : 3));
          acc_item[cls] += t1 - tg; cnt_item[cls] += 1; tg = t1;
        }
#endif
      }
      P[0] = F.x; P[ROWF] = F.y; P[2 * ROWF] = Tq;
#ifdef VMAS_TRACE
      tg = TNOW();
#endif
    }
#ifdef VMAS_TRACE
    if (args.trace && lane == 0) {
      unsigned long long* tr = args.trace + ((long)blockIdx.x * 16 + wv) * 16;
      tr[6] = acc_grab; tr[7] = acc_pro;
      for (int c = 0; c < 4; ++c) { tr[8 + c] = acc_item[c]; tr[12 + c] = cnt_item[c]; }
    }
#endif
    STAMP(3);
    __syncthreads();
    if (!lazy || args.lz_g != nullptr) break;
    // ---- lazy form, behind the optimistic pass: publish this tile's overlap words; a band pair that no environment of
    //      the TILE overlaps needs the batch's word (rare: see vmas_env_device.h)
    if (!lazy_after_pass(args, it, lz_words, lzp, c_gather, first_dyn)) break;
    }
    STAMP(4);

    // ================= phase C: _integrate_state core.py:2862-2908 (+ trig for the next substep,
    //                   or, after the last substep, the write-back of the entity's planes)
    const int n_own = (ABLATE(args) & 2) ? 0 : W.n_owned;
    // this substep's fired bytes of the shared sphere-sphere records (one fetch per wave, off the per-entity chain)
    uint32_t fw0 = ~0u, fw1 = ~0u;
    if (W.fired_recs > 0) {
      const uint2 fw = *(const uint2*)fired;
      fw0 = (uint32_t)sgpr((int)fw.x); fw1 = (uint32_t)sgpr((int)fw.y);
    }
    // reference bits 19..25: 1 + index of the pair's fired bit (4 * record + k), 0 = not published (always read)
    auto ref_fired = [&](uint32_t rf) -> bool {
      const uint32_t b1 = (rf >> 19) & 0x7fu;
      if (b1 == 0) return true;
      const uint32_t b = b1 - 1u;
      return (((b < 32u ? fw0 : fw1) >> (b & 31u)) & 1u) != 0u;
    };
    for (int oi = first_dyn ? wv : grab(c_integrate); oi < n_own; oi = grab(c_integrate)) {
      const uint4* op = (const uint4*)(blob + W.b_owned + oi * OW);
      const uint4 o0 = op[0], o1 = op[1], o2 = op[2], o3 = op[3], q0 = op[4], q1 = op[5];
      const int e = sgpr((int)o0.x);
      float* Es = tile + (int)o0.y;
      const float* P = tile + (int)o0.z;
      const int n_parts = sgpr((int)o0.w);
      const int r0 = sgpr((int)o1.x), nr = sgpr((int)o1.y);
      const uint32_t fl = (uint32_t)sgpr((int)o1.z);
      struct { int32_t shape, tr_off; float mass, inertia, one_minus_drag, max_speed, v_range; } D;
      D.shape = sgpr((int)o1.w); D.tr_off = sgpr((int)o2.x);
      D.mass = __uint_as_float(o2.y); D.inertia = __uint_as_float(o2.z); D.one_minus_drag = __uint_as_float(o2.w);
      D.max_speed = __uint_as_float(o3.x); D.v_range = __uint_as_float(o3.y);
      const uint32_t ref[8] = {(uint32_t)sgpr((int)q0.x), (uint32_t)sgpr((int)q0.y), (uint32_t)sgpr((int)q0.z),
                               (uint32_t)sgpr((int)q0.w), (uint32_t)sgpr((int)q1.x), (uint32_t)sgpr((int)q1.y),
                               (uint32_t)sgpr((int)q1.z), (uint32_t)sgpr((int)q1.w)};
      // second round trip: the entity's state, its first partial row and the rows of its first eight shared pairs
      float es[6];
#pragma unroll
      for (int f = 0; f < 6; ++f) es[f] = Es[f * ROWF];
      v2 F = V(P[0], P[ROWF]);
      float Tq = P[2 * ROWF];
      float fx[8], fy[8], tq[8];
#pragma unroll
      for (int c = 0; c < 2; ++c)
        if (nr > 4 * c) {
#pragma unroll
          for (int k = 4 * c; k < 4 * c + 4; ++k) {
            if (!ref_fired(ref[k])) continue;  // (uniform) nobody wrote the rows of a pair that did not fire
            const float* R = tile + (ref[k] & 0xffffu) * ROWF;  // (padding repeats a valid row)
            const int td = (int)((ref[k] >> 17) & 3u);
            fx[k] = R[0];
            fy[k] = R[ROWF];
            tq[k] = R[td * ROWF];
          }
        }
      for (int p = 1; p < n_parts; ++p) {
        F = F + V(P[3 * p * ROWF], P[(3 * p + 1) * ROWF]);
        Tq = Tq + P[(3 * p + 2) * ROWF];
      }
      // the entity's side of the shared pairs/joints, in the reference's order (core.py:2176-2199)
      auto add_ref = [&](uint32_t rf, float x, float y, float t) {
        const uint32_t flip = (rf & 0x10000u) << 15;  // b's side: -f
        if (fl & VMAS_F_MOVABLE) F = F + V(__uint_as_float(__float_as_uint(x) ^ flip), __uint_as_float(__float_as_uint(y) ^ flip));
        if (((rf >> 17) & 3u) && (fl & VMAS_F_ROTATABLE)) Tq = Tq + t;
      };
      // a pair that did not fire contributes +0 (a's side) or -0 (b's side) in every environment: F + (-0) == F, and
      // F + (+0) == F unless F == -0, so one +0 is added at the end if any a-side reference was skipped - the same bits
      // as adding them one by one (once F is +0 only -0 terms could follow without changing it, and they do not)
      bool skipped_a = false;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (k < nr) {
          if (ref_fired(ref[k])) add_ref(ref[k], fx[k], fy[k], tq[k]);
          else if (!(ref[k] & 0x10000u)) skipped_a = true;
        }
      for (int r = 8; r < nr; r += 4) {  // (more than eight: four references per fetch, their rows requested together)
        const uint4 q = *(const uint4*)(blob + W.b_refs + r0 + r);  // (ref_begin is a multiple of 4, the tail is padded)
        const uint32_t rf[4] = {(uint32_t)sgpr((int)q.x), (uint32_t)sgpr((int)q.y), (uint32_t)sgpr((int)q.z),
                                (uint32_t)sgpr((int)q.w)};
        float gx[4], gy[4], gt[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (!ref_fired(rf[k])) continue;
          const float* R = tile + (rf[k] & 0xffffu) * ROWF;
          gx[k] = R[0];
          gy[k] = R[ROWF];
          gt[k] = R[((rf[k] >> 17) & 3u) * ROWF];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (r + k < nr) {
            if (ref_fired(rf[k])) add_ref(rf[k], gx[k], gy[k], gt[k]);
            else if (!(rf[k] & 0x10000u)) skipped_a = true;
          }
      }
      if (skipped_a && (fl & VMAS_F_MOVABLE)) F = F + V(0.f, 0.f);
      float* dst = state + (long)e * 6 * ld + env;
      if (fl & VMAS_F_MOVABLE) {
        v2 vel = V(es[2], es[3]);
        if (substep == 0) vel = V(vel.x * D.one_minus_drag, vel.y * D.one_minus_drag);
        const rcp_t rm = rcp_of(D.mass);
        const v2 acc = V(F.x / rm, F.y / rm);
        vel = V(vel.x + acc.x * sub_dt, vel.y + acc.y * sub_dt);
        if (fl & VMAS_F_MAX_SPEED) vel = clamp_with_norm(vel, D.max_speed);
        if (fl & VMAS_F_V_RANGE) vel = V(clamp_t(vel.x, D.v_range), clamp_t(vel.y, D.v_range));
        v2 np = V(es[0] + vel.x * sub_dt, es[1] + vel.y * sub_dt);
        if (W.xs == W.xs) np.x = clamp_t(np.x, W.xs);
        if (W.ys == W.ys) np.y = clamp_t(np.y, W.ys);
        if (last) {
          if (live) { dst[0] = np.x; dst[ld] = np.y; dst[2 * ld] = vel.x; dst[3 * ld] = vel.y; }
          if constexpr (ENV != ENV_NONE) { Es[0] = np.x; Es[ROWF] = np.y; Es[2 * ROWF] = vel.x; Es[3 * ROWF] = vel.y; }
        } else {
          Es[0] = np.x; Es[ROWF] = np.y; Es[2 * ROWF] = vel.x; Es[3 * ROWF] = vel.y;
        }
      }
      if (fl & VMAS_F_ROTATABLE) {
        float av = es[5];
        if (substep == 0) av = av * D.one_minus_drag;
        av = av + (Tq / D.inertia) * sub_dt;
        const float rot = es[4] + av * sub_dt;
        if (last) {
          if (live) { dst[4 * ld] = rot; dst[5 * ld] = av; }
          if constexpr (ENV != ENV_NONE) { Es[4 * ROWF] = rot; Es[5 * ROWF] = av; }
        } else {
          Es[4 * ROWF] = rot; Es[5 * ROWF] = av;
          if (D.tr_off >= 0) write_trig(tile + D.tr_off, rot, D.shape);
        }
      }
    }
    if (!last) __syncthreads();
  }
  // ---- epilogue of this step: the scenario's reward / observation / done on the tile that is still in LDS.  In a
  //      multi-step rollout (vmas_world_rollout_env) step k writes the k-th slab of every per-step output; the
  //      persistent terms (shaping, step counter) are carried in registers / re-read by the thread that wrote them.
  if constexpr (ENV == ENV_NAVIGATION) {  // (the collision penalties need a reduction over all tiles: a grid barrier per
                                          //  step, or - single steps only - a kernel behind this one; navigation_post_tile)
    __syncthreads();
#ifdef VMAS_TRACE
    unsigned long long* nav_tr = args.trace ? args.trace + ((long)blockIdx.x * 16 + wv) * 16 : nullptr;
#else
    unsigned long long* nav_tr = nullptr;
#endif
    navigation_post_tile(TileCtx(batch), E.navigation.d, E.navigation.o, E.navigation.w, batch, lds, lds + E.scratch_off,
                         post_steps, nav_tr, stp, stp + 1 == n_steps);
    if (stp + 1 < n_steps) __syncthreads();  // the next step's prologue rewrites the agent-force rows
  }
  if constexpr (ENV == ENV_BALANCE || ENV == ENV_TRANSPORT) {
    if (stp + 1 == n_steps) __syncthreads();  // (earlier steps: the substep loop ended with a barrier)
    const TileCtx C(batch);
    if constexpr (ENV == ENV_BALANCE)
      if (!(ABLATE(E) & 4))  // profiling (VMAS_ENV_ABLATE): 1 queries off, 2 observations off, 4 epilogue off, 8 prologue off
      {
        // the floor's cos/sin rows of the tile are current if it cannot rotate (they are not refreshed after the last substep)
        const uint32_t ffl = (uint32_t)sgpr((int)blob[W.b_ent + E.balance.d.floor * EW]);
        const int tr_off = sgpr((int)blob[W.b_ent + E.balance.d.floor * EW + 3]);
        VmasBalanceBuffers o = E.balance.o;
        if (stp > 0) {
          const long nb = (long)E.balance.d.n_agents * batch;
          o.obs += (long)stp * nb * kBalanceObsDim; o.rew += (long)stp * nb;
          o.pos_rew += (long)stp * batch; o.ground_rew += (long)stp * batch; o.done += (long)stp * batch;
        }
        balance_post_tile(C, E.balance.d, o, batch, lds, lds + E.scratch_off, post_prev, post_steps, ABLATE(E),
                          (tr_off >= 0 && !(ffl & VMAS_F_ROTATABLE)) ? tile + tr_off : nullptr);
      }
    if constexpr (ENV == ENV_TRANSPORT) {
      VmasTransportBuffers o = E.transport.o;
      if (stp > 0) {
        const long nb = (long)E.transport.d.n_agents * batch;
        o.obs += (long)stp * nb * transport_obs_dim(E.transport.d.n_packages); o.rew += (long)stp * nb;
        o.done += (long)stp * batch;
      }
      transport_post_tile(C, E.transport.d, o, batch, lds, lds + E.scratch_off, post_steps);
    }
    if (stp + 1 < n_steps) {
      __syncthreads();  // the next step's prologue rewrites the agent-force rows and the epilogue's scratch
      if constexpr (ENV == ENV_TRANSPORT) {  // previous shaping of every package: by the wave that just stored it
        float* term = lds + E.scratch_off + lane;
        for (int p = wv; p < E.transport.d.n_packages; p += nw)
          term[p * 64] = live ? E.transport.o.global_shaping[(long)p * batch + env] : 0.f;
      }
    }
  }
  }
  STAMP(5);
}

// ------------------------------------------------------------------------------------
// world-specialised form of the same kernel (compile-time schedule tables)
// ------------------------------------------------------------------------------------
#ifdef VMAS_PLAN_ONLY  // scripts/plan_lib.sh: the host-only PLANNING library that scripts/gen_spec.py generates vmas_spec_gen.h with -
#define VMAS_SPEC_LIST(X)  // it must not depend on the header it is there to (re)generate
#else
#include "vmas_spec_gen.h"
#endif
#include "vmas_spec_kernel.h"

// ------------------------------------------------------------------------------------
// lane-compacted form for dense sphere worlds (football): broad phase per environment, narrow phase per contact
// ------------------------------------------------------------------------------------
#include "vmas_compact.h"

// ------------------------------------------------------------------------------------
// batch-global broad phase (World.collides core.py:2797-2801)
// ------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void pair_mask_kernel(const DevMaskPair* __restrict__ pairs, int nP, int nE,
                                                        const float* __restrict__ state, long ld, int batch,
                                                        uint32_t* __restrict__ mask) {
  extern __shared__ float pos[];  // [nE * 2][T]: each lane's own column (dynamic indexing, no exchange)
  const int T = blockDim.x;
  const long env = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool ok = env < batch;
  const long e = ok ? env : (long)batch - 1;
  for (int i0 = 0; i0 < nE * 2; i0 += 16) {  // 16 independent coalesced loads in flight, then their LDS stores
    float t[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int i = i0 + k < nE * 2 ? i0 + k : nE * 2 - 1;
      t[k] = state[((long)(i >> 1) * 6 + (i & 1)) * ld + e];
    }
#pragma unroll
    for (int k = 0; k < 16; ++k)
      if (i0 + k < nE * 2) pos[(i0 + k) * T + threadIdx.x] = t[k];
  }
  // one word of 32 pairs at a time: the wave ORs its lanes' hits into `bits` and touches global
  // memory once per word - and only if it would set a bit that is not set yet (the mask only grows,
  // so a stale read is harmless); every wave hammering one word with atomics cost 35 us at 65536 envs
  for (int p0 = 0; p0 < nP; p0 += 32) {
    uint32_t bits = 0;
    const int n = nP - p0 < 32 ? nP - p0 : 32;
    for (int k = 0; k < n; ++k) {
      const DevMaskPair P = pairs[p0 + k];
      const float* sa = pos + P.a * 2 * T + threadIdx.x;
      const float* sb = pos + P.b * 2 * T + threadIdx.x;
      const bool hit = ok && norm2(sa[0] - sb[0], sa[T] - sb[T]) <= P.bound_sum;
      bits |= (__any(hit) ? 1u : 0u) << k;
    }
    if ((threadIdx.x & 63) == 0 && (bits & ~__builtin_nontemporal_load(&mask[p0 >> 5])) != 0u)
      atomicOr(&mask[p0 >> 5], bits);
  }
}

// ------------------------------------------------------------------------------------
// LIDAR (World.cast_rays core.py:1662-1786): one thread per (environment, sensor)
// ------------------------------------------------------------------------------------
// directions of the unrotated rays (see lidar_cast_chunk): the same sincosf on the same input as the general path
__global__ void lidar_table_kernel(const float* __restrict__ angles, int n, float2* __restrict__ cs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s, c;
  sincosf(angles[i] + 0.f, &s, &c);
  cs[i] = make_float2(c, s);
}

template <int RAY_CHUNK>  // rays per thread; blockIdx.z selects the chunk of the sensor's fan
__global__ __launch_bounds__(256) void lidar_kernel(const DevLidar* __restrict__ lidars,
                                                    const DevTarget* __restrict__ targets,
                                                    const float* __restrict__ angles,
                                                    const float2* __restrict__ angles_cs, int max_rays,
                                                    const float* __restrict__ state, long ld, int batch,
                                                    float* __restrict__ out) {
  const long env = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= batch) return;
  const int l = blockIdx.y;
  const DevLidar L = lidars[l];
  const float* sp = state + (long)L.entity * 6 * ld + env;
  const v2 o = V(sp[0], sp[ld]);
  const float arot = sp[4 * ld];
  for (int r0 = blockIdx.z * RAY_CHUNK; r0 < L.n_rays; r0 += gridDim.z * RAY_CHUNK) {
    float best[RAY_CHUNK];
    lidar_cast_chunk<RAY_CHUNK>(L, [&](int ti) { return targets[L.target_off + ti]; }, angles, angles_cs, state + env, ld, o,
                                arot, r0, best);
#pragma unroll
    for (int i = 0; i < RAY_CHUNK; ++i)
      if (r0 + i < L.n_rays) out[((long)l * max_rays + r0 + i) * ld + env] = best[i];
  }
}

// World.cast_rays for sensor sets whose targets are all SPHERES (navigation: every agent's LIDAR sees the other agents),
// lane-compacted like the navigation epilogue's cast (vmas_env_device.h, navigation_post_tile):
//   1. a block owns 64 environments; every (sensor, target) of every environment is tested for reach - the very test of
//      lidar_cast_chunk - and the near ones are queued in LDS (ballot + one LDS atomic per wave and pair);
//   2. a lane per queued (environment, sensor, target): a conservative filter over its rays, then the reference's
//      arithmetic (core.py:1414-1490, expression for expression lidar_cast_chunk's) for the rays that pass; a filtered-out
//      ray measures max_range against this target in the reference too, which never lowers the minimum;
//   3. min over the targets: an LDS atomic min on an order-preserving integer image of the distance (it can be negative:
//      a sensor inside a sphere); the rows start at max_range (core.py:1672-1674).
// A lane walking its own sensor's targets spends ~3 trips where ~1 target is in reach, 12 rays where ~2 can hit:
// 65 536 environments x 8 sensors x 12 rays x 7 targets: see DESIGN.md 3.3.
// LDS: pos[n_ent][2][64] | arot[n_lidars][64] | measured[n_lidars * max_rays][65] | count | queue[n_pairs_total * 64] (u32)
__device__ __forceinline__ int float_order(float f) { const int b = __float_as_int(f); return b ^ ((b >> 31) & 0x7fffffff); }
__device__ __forceinline__ float order_float(int k) { return __int_as_float(k ^ ((k >> 31) & 0x7fffffff)); }
constexpr int kLidarStride = 65;

// tables (host-built, staged into LDS - a descriptor read from global memory per use chained three dependent loads in front of
// every queued item): sensor l = 6 words: slot of its entity | n_rays | first angle | first pair | max_range | half_range;
// pair k = 2 words: sensor | target's slot << 16 ; target's radius
constexpr int kLidarSensorWords = 6, kLidarPairWords = 2;
__global__ __launch_bounds__(1024) void lidar_compact_kernel(const uint32_t* __restrict__ tab, const float* __restrict__ angles,
                                                            const float2* __restrict__ angles_cs,
                                                            const int* __restrict__ slot_ent /* [n_slots] */, int n_slots,
                                                            int n_lidars, int n_pairs_total, int max_rays,
                                                            const float* __restrict__ state, long ld, int batch,
                                                            float* __restrict__ out) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wave = sgpr(threadIdx.x >> 6), nw = sgpr(blockDim.x >> 6);
  const long b0 = (long)blockIdx.x * 64, env = b0 + lane;
  const bool live = env < batch;
  const long e = live ? env : (long)batch - 1;
  float* pos = lds;                                   // [n_slots][2][64]
  float* arot = pos + n_slots * 128;                  // [n_lidars][64]
  int* measured = (int*)(arot + n_lidars * 64);       // [n_lidars * max_rays][65]
  int* count = measured + n_lidars * max_rays * kLidarStride;
  uint32_t* sens = (uint32_t*)(count + 2);            // [n_lidars][6]
  uint32_t* pairs = sens + n_lidars * kLidarSensorWords;  // [n_pairs_total][2]
  uint32_t* queue = pairs + n_pairs_total * kLidarPairWords;  // lane | pair << 6
  // ---- stage: the tables, the positions of every entity that casts or is seen, the sensors' rotations; measured = max_range
  const int n_tab = n_lidars * kLidarSensorWords + n_pairs_total * kLidarPairWords;
  for (int i = threadIdx.x; i < n_tab; i += blockDim.x) sens[i] = tab[i];
  for (int i = wave; i < 2 * n_slots; i += nw) pos[i * 64 + lane] = state[((long)slot_ent[i >> 1] * 6 + (i & 1)) * ld + e];
  for (int l = wave; l < n_lidars; l += nw)
    arot[l * 64 + lane] = state[((long)slot_ent[tab[l * kLidarSensorWords]] * 6 + 4) * ld + e];
  for (int l = 0; l < n_lidars; ++l) {
    const int init = float_order(__uint_as_float(tab[l * kLidarSensorWords + 4]));
    for (int i = threadIdx.x; i < max_rays * kLidarStride; i += blockDim.x) measured[l * max_rays * kLidarStride + i] = init;
  }
  if (threadIdx.x == 0) count[0] = 0;
  __syncthreads();
  // ---- 1. the near (sensor, target) pairs: eight pairs per trip, ONE LDS atomic for the wave's slots
  for (int k0 = wave; k0 < n_pairs_total; k0 += 8 * nw) {
    unsigned long long bal[8];
    int total = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = k0 + j * nw;
      bal[j] = 0ull;
      if (k >= n_pairs_total) continue;
      const uint32_t p0 = (uint32_t)sgpr((int)pairs[k * kLidarPairWords]);
      const float radius = __uint_as_float((uint32_t)sgpr((int)pairs[k * kLidarPairWords + 1]));
      const int l = (int)(p0 & 0xffffu), tslot = (int)(p0 >> 16);
      const int sslot = sgpr((int)sens[l * kLidarSensorWords]);
      const float max_range = __uint_as_float((uint32_t)sgpr((int)sens[l * kLidarSensorWords + 4]));
      const float* ps = pos + sslot * 128 + lane;
      const float* pt = pos + tslot * 128 + lane;
      const float dx = pt[0] - ps[0], dy = pt[64] - ps[64];
      const float lim = max_range + radius + 1e-4f;
      const bool near = live && !(dx * dx + dy * dy > lim * lim);  // NaN counts as near
      bal[j] = __ballot(near);
      total += __popcll(bal[j]);
    }
    if (total == 0) continue;
    int base = 0;
    if (lane == 0) base = atomicAdd(&count[0], total);
    base = sgpr(base);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (bal[j] == 0ull) continue;
      if ((bal[j] >> lane) & 1ull)
        queue[base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal[j] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal[j], 0u))] =
            (uint32_t)lane | (uint32_t)(k0 + j * nw) << 6;
      base += __popcll(bal[j]);
    }
  }
  __syncthreads();
  // ---- 2. a lane per queued item
  const int n_items = count[0];
  for (int i0 = wave * 64; i0 < n_items; i0 += nw * 64) {
    const bool on = i0 + lane < n_items;
    const uint32_t item = on ? queue[i0 + lane] : 0u;
    const int el = item & 63, k = (int)(item >> 6);
    const uint32_t p0 = pairs[k * kLidarPairWords];
    const float radius = __uint_as_float(pairs[k * kLidarPairWords + 1]);
    const int l = (int)(p0 & 0xffffu), tslot = (int)(p0 >> 16);
    const uint32_t* S = sens + l * kLidarSensorWords;
    const int n_rays = (int)S[1], angle_off = (int)S[2];
    const float half_range = __uint_as_float(S[5]);
    const float* ps = pos + (int)S[0] * 128 + el;
    const float* pt = pos + tslot * 128 + el;
    const v2 op = V(ps[0], ps[64]), tpos = V(pt[0], pt[64]);
    const float rot = arot[l * 64 + el];
    const v2 u = tpos - op;
    const float margin = radius + 1e-4f + 1e-5f * (fabsf(op.x) + fabsf(op.y) + fabsf(tpos.x) + fabsf(tpos.y));
    const bool table = __all(!on || rot == 0.f);  // (see lidar_cast_chunk: the same sincosf made the table)
    auto direction = [&](int r, float& c, float& sn) {
      if (table) { const float2 cs = angles_cs[angle_off + r]; c = cs.x; sn = cs.y; }
      else sincosf(angles[angle_off + r] + rot, &sn, &c);  // sensors.py:118
    };
    unsigned long long cand = 0ull;
    for (int r = 0; r < max_rays; ++r) {  // (sensors may differ in n_rays: the surplus is masked)
      float c = 1.f, sn = 0.f;
      const bool in = r < n_rays;
      if (in) direction(r, c, sn);
      const bool maybe = !(fabsf(u.x * sn - u.y * c) > margin) && vdot(u, V(c, sn)) > 0.f;
      if (on && in && maybe) cand |= 1ull << r;
    }
    int* mrow = measured + l * max_rays * kLidarStride + el;
    while (__any(cand != 0ull)) {
      if (cand != 0ull) {
        const int r = __ffsll((long long)cand) - 1;
        cand &= cand - 1ull;
        float c, sn;
        direction(r, c, sn);
        const v2 dir = V(c, sn);  // _cast_rays_to_sphere core.py:1414-1490
        const v2 lp = V(op.x + dir.x * half_range, op.y + dir.y * half_range);
        const v2 cp = closest_point_line<false>(lp, c, sn, 0.f, tpos);
        const float dn = vnorm(tpos - cp);
        const bool ok = (dn < radius) && (vdot(u, dir) > 0.f);
        const float a2 = radius * radius - dn * dn;
        const float m = sqrt_n(a2 > 0.f ? a2 : 1e-8f);
        const float dist = vnorm(cp - op) - m;
        if (ok) atomicMin(mrow + r * kLidarStride, float_order(dist));  // (min(best, dist): best starts at max_range)
      }
    }
  }
  __syncthreads();
  // ---- 3. out[(l * max_rays + r) * ld + env]
  if (live)
    for (int i = wave; i < n_lidars * max_rays; i += nw) {
      const int l = i / max_rays, r = i - l * max_rays;
      if (r < (int)sens[l * kLidarSensorWords + 1]) out[(long)i * ld + env] = order_float(measured[i * kLidarStride + lane]);
    }
}

// ------------------------------------------------------------------------------------
// scenario-side queries: World.get_distance core.py:1822-1905, World.is_overlapping core.py:1907-1969
// one thread per environment, blockIdx.y = query (wave-uniform shape dispatch)
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void query_kernel(const DevQuery* __restrict__ queries, const float* __restrict__ state,
                                                    long ld, int batch, float* __restrict__ out) {
  const long env = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= batch) return;
  const DevQuery Q = queries[blockIdx.y];
  int overlap;
  const float dist = pair_distance(Q, state, ld, env, overlap);
  float r = dist;
  if (Q.kind == VMAS_QUERY_OVERLAP) r = overlap ? 1.f : 0.f;
  out[(long)blockIdx.y * ld + env] = r;
}

// test hook (include/vmas_debug_hip.h, not part of the drop-in ABI): device primitives on arrays, for the accuracy tests
__global__ void math_kernel(int op, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = a[i], y = b ? b[i] : 0.f;
  float r = 0.f, sn, cs;
  switch (op) {
    case VMAS_MATH_SOFTPLUS: r = softplus0(x); break;
    case VMAS_MATH_SQRT: r = sqrt_n(x); break;
    case VMAS_MATH_DIV: r = x / rcp_of(y); break;
    case VMAS_MATH_NORM: r = norm2(x, y); break;
    case VMAS_MATH_COS: sincosf(x, &sn, &cs); r = cs; break;
    case VMAS_MATH_SIN: sincosf(x, &sn, &cs); r = sn; break;
    default: break;
  }
  out[i] = r;
}

// ------------------------------------------------------------------------------------
// host side: the C ABI
// ------------------------------------------------------------------------------------
// A/B and profiling knobs (VMAS_ABLATE, VMAS_TRACE, VMAS_SHARE, VMAS_NO_SSQ ...) are read from the environment only
// in -DVMAS_PROFILE builds (libvmas_hip_profile.so, scripts/gpu_*.sh); the product library has none.
static inline const char* knob(const char* name) {
#ifdef VMAS_PROFILE
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

static thread_local char g_err[512] = "";
static int fail(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return -1;
}
namespace vmas {
int host_fail(const char* msg) { return fail("%s", msg); }  // for vmas_env.hip
// argument validation shared with the stand-alone entry points (vmas_env.hip); n_entities < 0 = unknown
int check_ingest_args(const VmasIngestArgs* args, int32_t batch, const float* agent_ft, int64_t ld);
int check_balance_args(const VmasBalanceDesc* d, const VmasBalanceBuffers* o, int32_t batch, const float* state, int64_t ld,
                       int n_entities);
int check_navigation_args(const VmasNavigationDesc* d, const VmasNavigationBuffers* o, int32_t batch, const float* state,
                          int64_t ld, int fused);
int launch_navigation_collisions(const VmasNavigationDesc* d, const VmasNavigationBuffers* o, int32_t batch,
                                 const float* state, int64_t ld, uint32_t* mask, int mask_words, void* stream,
                                 const uint32_t* gate);
int check_transport_args(const VmasTransportDesc* d, const VmasTransportBuffers* o, int32_t batch, const float* state,
                         int64_t ld, int n_entities);
int launch_football_post(const VmasFootballDesc* d, const VmasFootballBuffers* o, int32_t batch, const float* state, int64_t ld,
                         int stp, void* stream, const uint32_t* gate);
}
#define HIP_TRY(x)                                                                      \
  do {                                                                                  \
    hipError_t e_ = (x);                                                                \
    if (e_ != hipSuccess) return fail("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

template <class T>
static hipError_t upload(T** dst, const std::vector<T>& src) {
  hipError_t e = hipMalloc((void**)dst, src.empty() ? 16 : src.size() * sizeof(T));
  if (e != hipSuccess || src.empty()) return e;
  return hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice);
}

// A schedule = how the tile's items are dealt to `nw` waves (see file header).
struct Sched {
  int nw = 0;
  DevWorld dw{};
  size_t lds_bytes = 0;
  int rows = 0;                  // LDS rows of the tile in front of the blob (state | agent forces | trig | shared | partial)
  std::vector<uint32_t> h_blob;  // host copy of the descriptor blob (planning worlds; the world-specialised kernel's match)
  int spec_id = -1;              // >= 0: this schedule is word for word the one a generated specialisation was built from
  uint32_t* d_blob = nullptr;
  // a specialisation compiled at RUN TIME for exactly this schedule (vmas_world_load_spec): its kernels by launch form
  struct Rt {
    hipModule_t mod = nullptr;
    hipFunction_t lean[2] = {nullptr, nullptr};                 // [tail]
    hipFunction_t multi[5][2][2] = {};                           // [ENV_*][one][tail]
    int post = 0;
    bool ok = false;
    std::map<hipFunction_t, size_t> lds_set;                     // dynamic LDS each function was opted in to (> 64 KB)
  } rt;
  void release() {
    (void)hipFree(d_blob);
    d_blob = nullptr;
    if (rt.mod) (void)hipModuleUnload(rt.mod);
    rt = Rt{};
  }
};

struct VmasWorld {
  int row_tr = 0;  // first trig row of the tile (after state and agent-force rows)
  int device = 0;
  int batch = 0;
  int lanes = 1;  // waves per 64-env tile = lanes cooperating on one environment
  int level = 0;  // kernel code level (which item types exist)
  int n_pairs = 0, n_dyn = 0;
  DevWorld base{};  // schedule-independent part
  std::vector<VmasEntityDesc> ents;
  std::vector<VmasPairDesc> pairs;     // copies of the creation-time description: the item lists can be rebuilt
  std::vector<VmasJointDesc> joints;   // (without shared rows) if a launch needs more LDS than the tile has left
  std::vector<int> tr_row;
  int share_mode = 0;
  int n_cu = 256;
  size_t reserve_fixed = 0, reserve_per_wave = 0;  // LDS (bytes) of a fused epilogue beside the tile (vmas_world_reserve_epilogue)
  std::vector<DevEntity> dev_ents;
  // static item lists
  std::vector<DevItem> items;
  std::vector<int> ent_item_begin;  // [nE+1]
  // pairs/joints of two dynamic entities are evaluated ONCE ("shared"): their items follow the entities' own items,
  // their results live in LDS rows [row_shared, row_shared + n_shared_rows) that both owners read in phase C
  int unit_item_begin = 0, row_shared = 0, n_shared_rows = 0;
  int fired_recs = 0;  // shared sphere-sphere records that publish their fired bits (build_items)
  std::vector<uint32_t> refs;       // per entity, reference order: row | side << 16 | torque row delta << 17
  std::vector<int> ent_ref_begin;   // [nE+1]
  std::vector<int> ent_ref_count;   // [nE]
  std::vector<float> item_cost;
  std::vector<int> trig_ents;
  DevMaskPair* d_mpairs = nullptr;
  unsigned long long* d_trace = nullptr;
  std::map<int, Sched> scheds;
  // vmas_world_step_n over several HIP queues (environments are independent: the launch gap of one part of the batch
  // overlaps the compute of the others).  queues: 0 = library's choice, 1..MAX_QUEUES = that many
  int queues = 0;
  bool host_only = false;  // a PLANNING world (device_id -1): schedules are built on the host, nothing is uploaded or launched
  bool use_spec = true;    // launch the world-specialised kernel when the schedule matches one (vmas_world_set_specialized)
  static constexpr int MAX_QUEUES = 4;
  hipStream_t side[MAX_QUEUES - 1] = {nullptr, nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_join[MAX_QUEUES - 1] = {nullptr, nullptr, nullptr};
  // exact broad phase (VmasStepArgs.exact_broad_phase): grid barrier word + ring of mask slots for the in-kernel form,
  // one mask for the launch-per-substep form used when the grid is larger than the chip
  uint32_t* d_sync = nullptr;
  uint32_t sync_seq = 0;
  // Pinned, device-mapped word the grid barriers set when they give up waiting (exact broad phase, navigation epilogue):
  // every entry point that launches on this world reads it first - no synchronisation - and fails loudly if an EARLIER
  // launch gave up (its step used a partial pair mask); vmas_world_exact_status reports it too.
  uint32_t* h_gave_up = nullptr;
  uint32_t* d_gave_up = nullptr;
  // The compacted kernel against the interpreter, chosen by what the tiles do (compact_mode -1 only).  The compacted kernel
  // wins while contacts are sparse and loses as they get denser (football, 131 072 environments, one queue: random actions
  // - 1.9 contacts per tile and substep - 80 us against 83.5; one action held for 20 / 100 / 600 steps - the bodies drift
  // into the walls, ~50 contacts per tile after 100 steps - 97 / 115 / 178 us against 92 / 98 / 130).  So the kernel adds
  // every (tile, substep)'s contact count to a device counter; every kWindow-th step's worth of API calls that make plain
  // launches copies it out on the caller's stream (read one window later, behind its event); a window with more than
  // kContactsPerTile on average sends the next kBackoff plain launches to the interpreter, after which the compacted kernel
  // is probed again for a window.  Both kernels are within the parity tolerance of the reference but not bit-identical to
  // each other where an entity has three or more contacts (the interpreter adds an entity's items segment by segment, the
  // compacted kernel in the reference's pair order: 1-ulp differences); the choice is a function of the states alone, so
  // reruns are bitwise identical, and vmas_world_set_compact(0 | 1) pins one kernel.
  // the lane-compacted cast of sphere-only sensor sets (lidar_compact_kernel): its entity staging tables and LDS need
  struct LidarCompact { bool ok = false; int mode = -1; uint32_t* d_ent_slot = nullptr /* the tables */; int* d_slot_ent = nullptr; int n_slots = 0,
                        n_pairs_total = 0; size_t lds = 0; } lc;
  struct CompactAdapt {
    static constexpr int kWindow = 16, kBackoff = 512;
    static constexpr double kContactsPerTile = 4.0;
    unsigned long long* d_count = nullptr;   // device counter (DevStepArgs.contacts)
    unsigned long long* h_count = nullptr;   // pinned copy
    hipEvent_t copied = nullptr;   // fires when h_count holds the count as of `tiles_at_copy`
    bool pending = false;
    int in_window = 0, backoff = 0;
    int forced = -1;  // >= 0: the kernel of the step being enqueued was chosen by the caller (vmas_world_step_n over several
                      // queues decides ONCE per step, before its sub-range launches: every part of a step runs the same kernel,
                      // and the backoff counts steps, not launches)
    long tiles = 0, tiles_at_copy = 0, tiles_seen = 0;
    unsigned long long count_seen = 0;
    long switches = 0;
  } adapt;
  uint32_t* d_exact_mask = nullptr;
  // The LAZY form of the exact broad phase (vmas_env_device.h, LazyArgs): [passes][pair words][tiles] 64-bit words, each
  // written by one tile as (launch tag << 32) | pair bits.  Grown on demand (K-step rollouts need K x substeps passes).
  struct Lazy {
    unsigned long long* d = nullptr;
    size_t qwords = 0;              // capacity
    uint32_t tag = 0;               // the last launch's tag (a per-world counter; 0 is never used: fresh memory is zero)
    unsigned long long band_words = 0ull;  // bit w: pair word w holds a band-capable pair
    int n_band = 0;                 // band-capable pairs (anything but sphere-sphere); 0: the per-environment form IS exact
                                    // for every finite state (DESIGN.md 4 on non-finite poses in sphere-only worlds)
    bool ss_bound_differs = false;  // a sphere pair whose fp32 r_a + r_b is not its bounding-circle sum (the double sum
                                    // rounded): the packed sphere records cannot test it - no lazy form for this world
  } lazy;
  uint32_t* d_nav_mask = nullptr;  // navigation epilogue: World.collides' pair bits of the post-step state: two masks that
  int football_form = -1;          // football's Environment.step: -1 the library's choice, 0 one launch, 1 two per step (vmas_debug_football_form)
  int nav_flip = 0;                //   eager launches alternate between (nav_flip: the one the next launch fills; it is zero)
                                   //   and a third for captured launches
  uint32_t* d_nav_sync = nullptr;  // its grid-barrier form: unused | timeout flag | ring of four slots of 64-bit arrival-and-pair-bit words
  uint32_t nav_seq = 0;
  // what the last GATED launch advanced on the host (vmas_world_gated_refused takes it back: a refused launch arrived nowhere)
  uint32_t gated_nav_seq = 0;
  int gated_nav_flip = 0;
  std::vector<DevLidar> h_lidars;  // host copy of the registered sensors (argument checks of the navigation epilogue)
  std::vector<DevTarget> h_targets;
  // the lane-compacted kernel's plan (vmas_compact.h): built at creation when the world qualifies
  struct CompactPlan {
    bool ok = false;
    compact::DevCompact dc{};
    uint32_t* d_blob = nullptr;
    std::vector<uint32_t> h_blob;  // host copy of the tables (vmas_debug_compact_plan: the planner's tests run without a GPU)
    size_t lds_bytes = 0;
    int nw = 8, own = 1;
    float4* d_trig = nullptr;  // cos / sin of the static lines' rotations (environment 0), filled at the first launch
    bool trig_ready = false;
  } cp;
  int compact_mode = -1;  // -1 the library's choice (dense worlds), 0 never, 1 whenever the world qualifies (vmas_world_set_compact)
  // lidars
  DevLidar* d_lidars = nullptr;
  DevTarget* d_targets = nullptr;
  float* d_angles = nullptr;
  float2* d_angles_cs = nullptr;  // cos, sin of every registered ray at rotation 0 (lidar_table_kernel)
  int n_lidars = 0, max_rays = 0;
  // queries
  DevQuery* d_queries = nullptr;
  int n_queries = 0;
};

// circles_overlap's threshold (vmas_env_device.h): the largest fp32 radicand whose correctly rounded root is <= bound, i.e.
// sqrt(fma(dy, dy, dx * dx)) <= bound  <=>  fma(dy, dy, dx * dx) <= overlap_threshold(bound)  (core.py:2797-2799)
static float overlap_threshold(float bound) {
  if (!(bound >= 0.f)) return -1.f;  // (never overlaps; NaN bound: the reference's compare is false too)
  if (bound == kInf) return kInf;
  float t = (float)((double)bound * (double)bound);
  while (t > 0.f && sqrtf(t) > bound) t = nextafterf(t, 0.f);
  while (t < kInf && sqrtf(nextafterf(t, kInf)) <= bound) t = nextafterf(t, kInf);
  return t;
}
static bool band_capable(int pair_type) { return pair_type != VMAS_PAIR_SS; }
constexpr int kLazyMaxPairs = 512;  // worlds the lazy exact broad phase serves: 16 pair words, an 8 KB pair table in the tile

static float type_cost(int type) {
  // relative narrow-phase cost per item, from instruction counts of the compiled kernel
  switch (type) {
    case VMAS_PAIR_SS: return 60.f;   // ~130 in contact, ~30 skipped; sphere contacts are rare
    case VMAS_PAIR_LS: return 160.f;
    case VMAS_PAIR_LL: return 600.f;
    case VMAS_PAIR_BS: return 350.f;
    case VMAS_PAIR_BL: return 600.f;   // ~2000 when executed, but the separating-axis test rejects most
    case VMAS_PAIR_BB: return 15000.f;
    case TASK_JOINT: return 500.f;
  }
  return 100.f;
}

// forces vanish once shapes are farther apart than LINE_MIN_DIST ("per-environment
// broad phase"); the slack absorbs the rounding of the closest-point arithmetic.

// share_mode: 0 every side evaluates its own copy of a pair | 1 pairs/joints of two dynamic entities are evaluated once,
// except sphere-sphere pairs | 2 those too
static void build_items(VmasWorld* w, int share_mode) {
  const std::vector<int>& tr_row = w->tr_row;
  w->share_mode = share_mode;
  const VmasEntityDesc* E = w->ents.data();
  const int nE = (int)w->ents.size();
  const int n_pairs = (int)w->pairs.size(), n_joints = (int)w->joints.size();
  std::vector<std::vector<DevItem>> per(nE);
  auto dyn = [&](int e) { return (E[e].flags & (VMAS_F_MOVABLE | VMAS_F_ROTATABLE)) != 0; };
  // sphere-sphere partners of one entity are consecutive (type-major order): pack them four to
  // a record when the item list will live in LDS (the packed form is read from the blob only)
  const bool pack_ss = (size_t)(2 * n_pairs + 2 * n_joints) * sizeof(DevItem) <= (size_t)ITEMS_LDS_BUDGET &&
                       n_pairs < 65536 && !knob("VMAS_NO_SSQ");
  // ... and a pair/joint of TWO dynamic entities is evaluated once for both (same condition: blob records only)
  const bool share = share_mode > 0 && pack_ss && nE * 6 * ROWF < 65536;
  std::vector<DevItem> units;
  std::vector<std::vector<uint32_t>> ent_refs(nE);
  w->row_shared = w->row_tr + 4 * (int)w->trig_ents.size();
  int shared_rows = 0;
  auto push_sides = [&](DevItem t, int a, int b) {
    if (share && dyn(a) && dyn(b) && (t.type != VMAS_PAIR_SS || share_mode > 1)) {
      const bool no_tq = t.type == VMAS_PAIR_SS;
      const int n_rows = no_tq ? 2 : ((t.type == VMAS_PAIR_LS || t.type == VMAS_PAIR_BS) ? 3 : 4);
      const int row = w->row_shared + shared_rows;
      shared_rows += n_rows;
      t.side = 2 | ((row * ROWF) << 2);
      units.push_back(t);
      ent_refs[a].push_back((uint32_t)row | (no_tq ? 0u : (2u << 17)));
      ent_refs[b].push_back((uint32_t)row | (1u << 16) | (n_rows == 4 ? (3u << 17) : 0u));
      return;
    }
    if (dyn(a)) { t.side = 0; per[a].push_back(t); }
    if (dyn(b)) { t.side = 1; per[b].push_back(t); }
  };
  auto tr_off = [&](int e) { return tr_row[e] >= 0 ? (w->row_tr + tr_row[e]) * ROWF : 0; };
  for (int j = 0; j < n_joints; ++j) {  // joints first (core.py:2176)
    const VmasJointDesc& J = w->joints[j];
    DevItem t{};
    t.type = TASK_JOINT; t.index = j;
    t.oa = J.a * 6 * ROWF; t.ob = J.b * 6 * ROWF;
    t.flags = J.rotate ? 0u : IT_LOCK;
    t.tra = tr_off(J.a); t.trb = tr_off(J.b);
    t.p0 = J.delta_a[0]; t.p1 = J.delta_a[1]; t.q0 = J.delta_b[0]; t.q1 = J.delta_b[1];
    t.p2 = J.dist; t.p3 = J.fixed_rotation;
    t.thr2 = kInf;  // joints are never skipped
    push_sides(t, J.a, J.b);
  }
  for (int p = 0; p < n_pairs; ++p) {  // then pairs, already type-major (core.py:2178-2189)
    const VmasPairDesc& P = w->pairs[p];
    const VmasEntityDesc &A = E[P.a], &B = E[P.b];
    DevItem t{};
    t.type = P.type; t.index = p;
    t.oa = P.a * 6 * ROWF; t.ob = P.b * 6 * ROWF;
    t.flags = ((A.flags & VMAS_F_HOLLOW) ? IT_A_HOLLOW : 0u) | ((B.flags & VMAS_F_HOLLOW) ? IT_B_HOLLOW : 0u);
    t.tra = tr_off(P.a); t.trb = tr_off(P.b);
    float m = P.bound_sum + kLineMinDist + kSkipSlack;
    t.thr2 = m * m;
    t.q0 = overlap_threshold(P.bound_sum);  // (the lazy exact broad phase: this environment's bounding circles overlap)
    switch (P.type) {
      case VMAS_PAIR_SS: {  // the force is exactly 0 for dist > r_a + r_b (core.py:2836)
        t.p0 = A.radius + B.radius;
        m = t.p0 + 1e-4f;
        t.thr2 = m * m;
      } break;
      case VMAS_PAIR_LS:
        t.p0 = A.length / 2.f; t.p1 = B.radius + kLineMinDist;
        t.reach = B.radius + kLineMinDist + kSkipSlack;
        break;
      case VMAS_PAIR_LL: t.p0 = A.length / 2.f; t.p1 = B.length / 2.f; break;
      case VMAS_PAIR_BS:
        t.p0 = A.length; t.p1 = A.width; t.p2 = B.radius + kLineMinDist;
        t.reach = B.radius + kLineMinDist + kSkipSlack;
        break;
      case VMAS_PAIR_BL:
        t.p0 = A.length; t.p1 = A.width; t.p2 = B.length / 2.f;
        t.reach = kLineMinDist + kSkipSlack;  // the separating-axis gap already includes the line's extent
        break;
      case VMAS_PAIR_BB:
        t.p0 = A.length; t.p1 = A.width; t.p2 = B.length; t.p3 = B.width;
        t.reach = B.bound_radius + kLineMinDist + kSkipSlack;
        break;
      default: break;
    }
    push_sides(t, P.a, P.b);
  }
  w->ent_item_begin.assign(nE + 1, 0);
  w->items.clear();
  w->item_cost.clear();
  w->level = 0;
  if (pack_ss) {
    for (int e = 0; e < nE; ++e) {
      std::vector<DevItem> packed;
      size_t i = 0;
      while (i < per[e].size()) {
        if (per[e][i].type != VMAS_PAIR_SS) { packed.push_back(per[e][i++]); continue; }
        size_t j = i;
        while (j < per[e].size() && per[e][j].type == VMAS_PAIR_SS && j - i < 4) ++j;
        uint32_t wds[16] = {0};
        auto fbits = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u; };
        const int n = (int)(j - i);
        int ob[4], idx[4];
        float rs[4];
        for (int k = 0; k < 4; ++k) {
          const DevItem& it = per[e][i + (k < n ? k : 0)];
          ob[k] = it.side ? it.oa : it.ob;  // the OTHER sphere
          rs[k] = it.p0;
          idx[k] = it.index;
        }
        wds[0] = TASK_SSQ; wds[1] = (uint32_t)n;
        wds[4] = (uint32_t)(e * 6 * ROWF); wds[5] = ob[0]; wds[6] = ob[1]; wds[7] = ob[2];
        wds[8] = ob[3]; wds[9] = fbits(rs[0]); wds[10] = fbits(rs[1]); wds[11] = fbits(rs[2]);
        wds[12] = fbits(rs[3]); wds[13] = (uint32_t)idx[0] | ((uint32_t)idx[1] << 16);
        wds[14] = (uint32_t)idx[2] | ((uint32_t)idx[3] << 16);
        DevItem q;
        memcpy(&q, wds, sizeof(q));
        packed.push_back(q);
        i = j;
      }
      per[e].swap(packed);
    }
    // ... and likewise the lines a SPHERE owner meets (its side of line-sphere pairs: force only, no torque) -
    // in worlds of spheres, lines and boxes-vs-spheres only (kernel level 0, e.g. football's 110 wall pairs):
    // the level-1/2 kernels are register-bound and do not carry the extra code
    bool level0 = true;
    for (int e = 0; e < nE; ++e)
      for (const DevItem& t : per[e])
        if (t.type == VMAS_PAIR_LL || t.type == VMAS_PAIR_BL || t.type == VMAS_PAIR_BB || t.type == TASK_JOINT) level0 = false;
    for (const DevItem& t : units)
      if (t.type == VMAS_PAIR_LL || t.type == VMAS_PAIR_BL || t.type == VMAS_PAIR_BB || t.type == TASK_JOINT) level0 = false;
    for (int e = 0; e < nE && level0 && !knob("VMAS_NO_LSQ"); ++e) {
      std::vector<DevItem> packed;
      size_t i = 0;
      auto packable = [](const DevItem& it) {
        return it.type == VMAS_PAIR_LS && it.side == 1 && it.oa >= 0 && it.oa < 65536 && it.tra >= 0 && it.tra < 65536;
      };
      while (i < per[e].size()) {
        if (!packable(per[e][i])) { packed.push_back(per[e][i++]); continue; }
        size_t j = i;
        while (j < per[e].size() && packable(per[e][j]) && j - i < (size_t)LSQ_N) ++j;
        if (j - i < 2) { packed.push_back(per[e][i++]); continue; }  // a lone line is cheaper as a plain item
        uint32_t wds[16] = {0};
        auto fbits = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u; };
        const int n = (int)(j - i);
        uint32_t lo[LSQ_N], to[LSQ_N], idx[LSQ_N];
        float half[LSQ_N], thr[LSQ_N];
        for (int k = 0; k < LSQ_N; ++k) {
          const DevItem& it = per[e][i + (k < n ? k : 0)];
          lo[k] = (uint32_t)it.oa; to[k] = (uint32_t)it.tra; half[k] = it.p0; idx[k] = (uint32_t)it.index; thr[k] = it.q0;
        }
        // (eval_lsq: three lines and their bounding-circle thresholds; four lines until round 6)
        wds[0] = TASK_LSQ; wds[1] = (uint32_t)n; wds[2] = (uint32_t)per[e][i].ob; wds[3] = fbits(per[e][i].p1);
        wds[4] = lo[0] | (lo[1] << 16); wds[5] = lo[2];
        wds[6] = to[0] | (to[1] << 16); wds[7] = to[2];
        for (int k = 0; k < LSQ_N; ++k) wds[8 + k] = fbits(half[k]);
        wds[11] = fbits(thr[0]);
        wds[12] = idx[0] | (idx[1] << 16); wds[13] = idx[2];
        wds[14] = fbits(thr[1]); wds[15] = fbits(thr[2]);
        DevItem q;
        memcpy(&q, wds, sizeof(q));
        packed.push_back(q);
        i = j;
      }
      per[e].swap(packed);
    }
  }
  for (int e = 0; e < nE; ++e) {
    w->ent_item_begin[e] = (int)w->items.size();
    for (const DevItem& t : per[e]) {
      w->items.push_back(t);
      w->item_cost.push_back(t.type == TASK_SSQ   ? 40.f + 40.f * (float)t.side /* side word = n */
                             : t.type == TASK_LSQ ? 40.f + 70.f * (float)t.side
                                                  : type_cost(t.type));
      if (t.type == VMAS_PAIR_BB) w->level = std::max(w->level, 2);
      if (t.type == VMAS_PAIR_LL || t.type == VMAS_PAIR_BL || t.type == TASK_JOINT) w->level = std::max(w->level, 1);
    }
  }
  w->ent_item_begin[nE] = (int)w->items.size();
  // shared items behind the entities' own: consecutive sphere-sphere pairs packed four to a record
  w->unit_item_begin = (int)w->items.size();
  w->n_shared_rows = shared_rows;
  int n_ssp_recs = 0;
  std::map<int, int> fired_bit_of_row;  // first row of a shared sphere-sphere pair -> index of its fired bit
  for (size_t i = 0; i < units.size();) {
    if (units[i].type != VMAS_PAIR_SS) {
      w->items.push_back(units[i]);
      w->item_cost.push_back(type_cost(units[i].type));
      if (units[i].type == VMAS_PAIR_BB) w->level = std::max(w->level, 2);
      if (units[i].type == VMAS_PAIR_LL || units[i].type == VMAS_PAIR_BL || units[i].type == TASK_JOINT)
        w->level = std::max(w->level, 1);
      ++i;
      continue;
    }
    size_t j = i;
    while (j < units.size() && units[j].type == VMAS_PAIR_SS && j - i < 4) ++j;  // (their rows are consecutive: 2 each)
    const int n = (int)(j - i);
    auto fbits = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u; };
    uint32_t oa[4], ob[4], idx[4];
    float rs[4];
    for (int k = 0; k < 4; ++k) {
      const DevItem& it = units[i + (k < n ? k : 0)];
      oa[k] = (uint32_t)it.oa; ob[k] = (uint32_t)it.ob; rs[k] = it.p0; idx[k] = (uint32_t)it.index;
    }
    uint32_t wds[16] = {0};
    wds[0] = TASK_SSP; wds[1] = (uint32_t)n; wds[2] = (uint32_t)(units[i].side >> 2);
    wds[3] = (uint32_t)n_ssp_recs;  // < FIRED_RECS: the record publishes which of its pairs fired
    for (int k = 0; k < n && n_ssp_recs < FIRED_RECS; ++k)
      fired_bit_of_row[(int)((units[i + k].side >> 2) / ROWF)] = 4 * n_ssp_recs + k;
    ++n_ssp_recs;
    wds[4] = oa[0] | (oa[1] << 16); wds[5] = oa[2] | (oa[3] << 16);
    wds[6] = ob[0] | (ob[1] << 16); wds[7] = ob[2] | (ob[3] << 16);
    for (int k = 0; k < 4; ++k) wds[8 + k] = fbits(rs[k]);
    wds[12] = idx[0] | (idx[1] << 16); wds[13] = idx[2] | (idx[3] << 16);
    DevItem q;
    memcpy(&q, wds, sizeof(q));
    w->items.push_back(q);
    w->item_cost.push_back(40.f + 50.f * (float)n);
    i = j;
  }
  w->ent_ref_begin.assign(nE + 1, 0);
  w->refs.clear();
  w->ent_ref_count.assign(nE, 0);
  for (int e = 0; e < nE; ++e) {  // fetched four at a time: every entity's run starts at a multiple of 4, padded with its last reference
    w->ent_ref_begin[e] = (int)w->refs.size();
    w->ent_ref_count[e] = (int)ent_refs[e].size();
    for (uint32_t& rf : ent_refs[e]) {  // bits 19..25: 1 + fired-bit index of a published sphere-sphere pair
      auto it = fired_bit_of_row.find((int)(rf & 0xffffu));
      if (it != fired_bit_of_row.end()) rf |= (uint32_t)(it->second + 1) << 19;
    }
    w->refs.insert(w->refs.end(), ent_refs[e].begin(), ent_refs[e].end());
    while (w->refs.size() % 4) w->refs.push_back(ent_refs[e].back());
  }
  w->ent_ref_begin[nE] = (int)w->refs.size();
  w->fired_recs = std::min(n_ssp_recs, FIRED_RECS);
}

// Cut every dynamic entity's item list into segments of about total/nw cost and deal the
// segments to the waves, heaviest first, always to the least loaded wave.
static int build_sched(VmasWorld* w, int nw, Sched& S) {
  const int nE = w->base.nE;
  std::vector<DevSegment> segs;
  std::vector<float> seg_cost;
  std::vector<DevOwned> owned_all;
  float total = 0.f;
  for (int e = 0; e < nE; ++e)
    if (w->ents[e].flags & (VMAS_F_MOVABLE | VMAS_F_ROTATABLE)) {
      total += 60.f;
      for (int i = w->ent_item_begin[e]; i < w->ent_item_begin[e + 1]; ++i) total += w->item_cost[i];
    }
  for (size_t i = (size_t)w->unit_item_begin; i < w->items.size(); ++i) total += w->item_cost[i];
  const float target = std::max(total / (float)(2 * nw), 150.f);
  for (int e = 0; e < nE; ++e) {
    if (!(w->ents[e].flags & (VMAS_F_MOVABLE | VMAS_F_ROTATABLE))) continue;
    const int b = w->ent_item_begin[e], n = w->ent_item_begin[e + 1];
    DevOwned O{};
    O.entity = e; O.oe = e * 6 * ROWF; O.part_off = 3 * (int)segs.size();  // part_off holds the ROW until rebased below
    O.ref_begin = w->ent_ref_begin[e]; O.n_refs = w->ent_ref_count[e];
    {
      const DevEntity& DE = w->dev_ents[e];
      O.flags = DE.flags; O.shape = DE.shape; O.tr_off = DE.tr_off;
      O.mass = DE.mass; O.inertia = DE.inertia; O.one_minus_drag = DE.one_minus_drag;
      O.max_speed = DE.max_speed; O.v_range = DE.v_range;
      for (int k = 0; k < 8; ++k) {  // (padding repeats the last valid reference; none at all: row 0, never added)
        const int nr = O.n_refs;
        O.refs8[k] = nr > 0 ? w->refs[O.ref_begin + std::min(k, nr - 1)] : 0u;
      }
    }
    int i = b;
    bool first = true;
    do {
      float c = first ? 60.f : 0.f;
      int j = i;
      while (j < n && (j == i || c + w->item_cost[j] <= target)) c += w->item_cost[j++];
      segs.push_back({e, e * 6 * ROWF, i, j, first ? 1 : 0, 3 * (int)segs.size(), w->dev_ents[e].flags, 0});
      seg_cost.push_back(c);
      O.n_parts++;
      first = false;
      i = j;
    } while (i < n);
    owned_all.push_back(O);
  }
  const int n_ent_segs = (int)segs.size();
  for (int i = w->unit_item_begin, n = (int)w->items.size(); i < n;) {  // runs of shared items: no prologue, no partial rows
    float c = 0.f;
    int j = i;
    while (j < n && (j == i || c + w->item_cost[j] <= target)) c += w->item_cost[j++];
    segs.push_back({-1, 0, i, j, 0, 0, 0u, 0});
    seg_cost.push_back(c);
    i = j;
  }
  // waves pull segments from a counter at run time: order them heaviest first (dynamic LPT)
  std::vector<int> order(segs.size());
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return seg_cost[x] > seg_cost[y]; });
  std::vector<DevSegment> segs_sorted;
  for (int si : order) segs_sorted.push_back(segs[si]);
  std::vector<DevOwned> owned = owned_all;
  if (knob("VMAS_DEBUG_SCHED")) {
    fprintf(stderr, "[sched nw=%d] %d own + %d shared items, %d shared rows, %zu refs\n", nw, w->unit_item_begin,
            (int)w->items.size() - w->unit_item_begin, w->n_shared_rows, w->refs.size());
    for (int si : order) {
      fprintf(stderr, "[sched nw=%d] seg cost %.0f {e%d%s", nw, seg_cost[si], segs[si].entity, segs[si].first ? "*" : "");
      for (int i = segs[si].item_begin; i < segs[si].item_end; ++i)
        if (w->items[i].type >= TASK_SSQ)
          fprintf(stderr, " %s(n=%d)", (const char*[]){"SSQ", "LSQ", "SSP"}[w->items[i].type - TASK_SSQ], w->items[i].side);
        else
          fprintf(stderr, " %s%d-%d", (const char*[]){"SS", "LS", "LL", "BS", "BL", "BB", "J"}[w->items[i].type],
                  w->items[i].oa / (6 * ROWF), w->items[i].ob / (6 * ROWF));
      fprintf(stderr, "}\n");
    }
  }
  // rows -> tile offsets: [state | agent forces | trig | partial sums | flag | blob | counters]
  const int row_part = w->row_shared + w->n_shared_rows;
  for (auto& sg : segs_sorted) sg.part_off = sg.entity < 0 ? 0 : (row_part + sg.part_off) * ROWF;
  for (auto& o : owned) o.part_off = (row_part + o.part_off) * ROWF;
  S.nw = nw;
  S.dw = w->base;
  const int row_bad = row_part + 3 * n_ent_segs;
  std::vector<uint32_t> blob;
  auto append = [&](const void* p, size_t bytes) {
    int at = (int)blob.size();
    blob.resize(blob.size() + bytes / 4);
    if (bytes) memcpy(blob.data() + at, p, bytes);
    return at;
  };
  S.dw.b_ent = append(w->dev_ents.data(), w->dev_ents.size() * sizeof(DevEntity));
  while (blob.size() % 4) blob.push_back(0);  // (ds_read_b128 of the segment records)
  S.dw.b_segs = append(segs_sorted.data(), segs_sorted.size() * sizeof(DevSegment));
  while (blob.size() % 4) blob.push_back(0);  // (ds_read_b128 of the owned records)
  S.dw.b_owned = append(owned.data(), owned.size() * sizeof(DevOwned));
  while (blob.size() % 4) blob.push_back(0);  // (ds_read_b128 of four references)
  S.dw.b_refs = append(w->refs.data(), w->refs.size() * sizeof(uint32_t));
  while (blob.size() % 4) blob.push_back(0);  // 16-byte alignment of the pair table and the item records (ds_read_b128)
  // the pair table of the lazy exact broad phase (lazy_overlap_blob; the specialised kernels read it at compile time): per
  // static pair the tile offsets of a's and b's rows and circles_overlap's threshold, then a header of 4 words - right in
  // front of the items, so that its position follows from b_items and the count in the header
  // (worlds of more than kLazyMaxPairs pairs - pollock: 990 - go without: 16 bytes per pair of every resident tile's LDS;
  //  their exact broad phase takes the barrier / launch-per-substep form, exact_form)
  S.dw.n_pairs = (int)w->pairs.size() <= kLazyMaxPairs ? (int)w->pairs.size() : 0;
  S.dw.b_pairs = (int)blob.size();
  {
    auto fbits = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u; };
    for (int p = 0; p < S.dw.n_pairs; ++p) {
      const VmasPairDesc& P = w->pairs[p];
      blob.push_back((uint32_t)(P.a * 6 * ROWF)); blob.push_back((uint32_t)(P.b * 6 * ROWF));
      blob.push_back(fbits(overlap_threshold(P.bound_sum))); blob.push_back(0u);
    }
    blob.push_back((uint32_t)S.dw.n_pairs); blob.push_back(0u); blob.push_back(0u); blob.push_back(0u);
  }
  const size_t item_bytes = w->items.size() * sizeof(DevItem);
  S.dw.items_in_lds = item_bytes <= (size_t)ITEMS_LDS_BUDGET;
  S.dw.b_items = (int)blob.size();
  S.dw.blob_words = (int)blob.size() + (S.dw.items_in_lds ? (int)(item_bytes / 4) : 0);  // (a multiple of 4: 16-byte staging)
  append(w->items.data(), item_bytes);
  S.h_blob = blob;
  S.rows = row_bad;
  // A generated specialisation serves this schedule iff it was generated from the very same words (and geometry).
  S.spec_id = -1;
#define VMAS_MATCH_SPEC(G)                                                                                               \
  if (S.spec_id < 0 && nw == G::NW && w->level == G::LEVEL && (int)blob.size() == G::BLOB_WORDS && S.dw.items_in_lds &&  \
      w->base.trig_in_args && row_bad == G::ROWS && w->base.nE == G::NE && w->base.nA == G::NA &&                        \
      w->base.substeps == G::SUBSTEPS && w->base.off_af == G::OFF_AF && w->base.row_tr == G::ROW_TR &&                   \
      w->base.trig_mask == G::TRIG_MASK && w->base.box_mask == G::BOX_MASK && S.dw.b_ent == G::B_ENT &&                  \
      S.dw.b_segs == G::B_SEGS && S.dw.b_owned == G::B_OWNED && S.dw.b_refs == G::B_REFS && S.dw.b_items == G::B_ITEMS && \
      (int)segs_sorted.size() == G::N_SEGS && (int)owned.size() == G::N_OWNED && w->fired_recs == G::FIRED_RECS &&       \
      memcmp(blob.data(), G::blob, sizeof(uint32_t) * G::BLOB_WORDS) == 0)                                               \
    S.spec_id = G::ID;
  VMAS_SPEC_LIST(VMAS_MATCH_SPEC)
#undef VMAS_MATCH_SPEC
  if (!w->host_only) HIP_TRY(upload(&S.d_blob, blob));
  S.dw.blob = S.d_blob;
  S.dw.items = (const DevItem*)(S.d_blob + S.dw.b_items);
  S.dw.n_segs = (int)segs_sorted.size();
  S.dw.n_owned = (int)owned.size();
  S.dw.off_blob = row_bad * ROWF;  // (row_bad: the first row after the partial sums)
  S.dw.fired_recs = w->fired_recs;
  // + work counters, fired words (2 parities x 2), the tile's pair bits of the in-kernel exact broad phase (whole quads)
  // (... | work counters [4] | fired words [4] | this tile's pair words | the batch's pair words: in-kernel exact broad phase)
  // (the lazy form: [2 parities][overlap | band words] | need | batch | flag [4] - six arrays + 4 words in the same place)
  {
    const size_t mwp = (((size_t)w->n_pairs + 31) / 32 + 3) & ~(size_t)3;
    const size_t exact_words = (int)w->pairs.size() <= kLazyMaxPairs ? 6 * mwp + 4 : 2 * mwp;  // (lazy form | barrier form only)
    S.lds_bytes = ((size_t)row_bad * ROWF + S.dw.blob_words + 4 + 4 + exact_words) * sizeof(float);
  }
  if (knob("VMAS_DEBUG_SCHED")) fprintf(stderr, "[sched nw=%d] %d segments, LDS %zu B per tile\n", nw, (int)segs.size(), S.lds_bytes);
  return 0;
}

// rebuild the item lists without shared rows (all cached schedules are dropped)
static int unshare(VmasWorld* w) {
  for (auto& kv : w->scheds) kv.second.release();
  w->scheds.clear();
  build_items(w, 0);
  return 0;
}

static int get_sched(VmasWorld* w, int nw, Sched** out) {
  auto it = w->scheds.find(nw);
  if (it == w->scheds.end()) {
    Sched S;
    if (build_sched(w, nw, S)) return -1;
    it = w->scheds.emplace(nw, S).first;
  }
  *out = &it->second;
  return 0;
}

// Waves per 64-environment tile ("lanes per env") for the item lists as they are now.  The model, fitted to the sweeps in
// what counts is how many waves of this kernel RUN per CU,
//     running(nw) = min(tiles the CU can hold, tiles the batch gives it) x nw,
// where a CU holds min(160 KB / LDS per tile, max waves per CU / nw) tiles.  Largest running() wins; on a tie, more waves
// per tile if every tile of the batch is resident at once (latency regime: the tile's dependent chain gets shorter -
// football at 16384 envs 23.6 us with 8 waves, 18.7 with 16), fewer if not (throughput regime: less synchronisation per
// tile - balance at 1 M envs 156 us with 4 waves, 191 with 8), and a choice that keeps all tiles resident beats one that
// does not (balance at 32768 envs: 8.2 us with 2 x 8 waves per CU, 11.2 with 1 x 16).
struct LaneChoice { int nw = 0; long running = 0; bool resident = false; size_t lds = 0; bool spec = false; };
static int choose_lanes(VmasWorld* w, int n_cu, LaneChoice* out) {
  const long tiles = ((long)w->batch + TILE - 1) / TILE;
  const long need = (tiles + n_cu - 1) / n_cu;  // tiles per CU the batch asks for
  int work = 0;
  for (float c : w->item_cost) work += (int)c;
  work += w->n_dyn;
  const int max_w = w->level >= 2 ? 8 : MAX_WAVES;     // (the box-box kernel is compiled for 512-thread blocks)
  const int waves_per_cu = w->level >= 2 ? 8 : 16;     // by VGPRs: 2 / 4 waves per SIMD
  LaneChoice best;
  for (int nw = 1; nw <= max_w; nw <<= 1) {
    if (nw > 1 && work < 150 * (nw / 2)) break;  // no more waves than the tile has independent work for
    Sched* S;
    if (get_sched(w, nw, &S)) return -1;
    const size_t lds = S->lds_bytes + w->reserve_fixed + w->reserve_per_wave * nw;
    if (lds > 160 * 1024) continue;
    const long hold = std::min<long>((long)(160 * 1024 / lds), waves_per_cu / nw);
    if (hold < 1) continue;
    LaneChoice c;
    c.nw = nw; c.lds = lds;
    c.running = std::min(hold, need) * nw;
    c.resident = hold >= need;
    c.spec = S->spec_id >= 0;  // a built-in specialisation serves this geometry (1.4-1.8x the interpreter's rate)
    bool better;
    if (best.nw == 0) better = true;
    else if (c.running != best.running) better = c.running > best.running;
    else if (c.resident != best.resident) better = c.resident;
    else if (c.spec != best.spec) better = c.spec;  // (balance at 1 M environments: 4 and 8 waves per tile tie - 179 us on
                                                    //  the interpreter with 4, 99 on the specialised kernel with 8)
    else better = c.resident;  // tie: the larger nw (visited later) in the latency regime, the smaller in the throughput regime
    if (better) best = c;
  }
  *out = best;
  return 0;
}

// Which pairs are evaluated once (shared rows in LDS: mode 2 all of them, 1 all but sphere-sphere, 0 none) and how many
// waves work on a tile are chosen TOGETHER, by the number of waves the choice keeps running per CU (choose_lanes): the
// shared rows cost LDS, i.e. resident tiles.  Ties: co-resident tiles first; then, if the batch is resident at once, more
// waves per tile, else fewer; then the higher mode.  (Measured, DESIGN.md: football at 16384 envs 31.1 -> 28.8 us with
// everything shared, at 131072 envs - two resident tiles per CU without the rows, one with - 224 -> 351 us.)
static int select_config(VmasWorld* w) {
  static const int share_env = knob("VMAS_SHARE") ? atoi(knob("VMAS_SHARE")) : -1;  // (A/B measurements)
  int best_mode = -1;
  LaneChoice best;
  for (int mode = 2; mode >= 0; --mode) {
    if (share_env >= 0 && mode != share_env) continue;
    for (auto& kv : w->scheds) kv.second.release();
    w->scheds.clear();
    build_items(w, mode);
    if (mode > 0 && w->n_shared_rows == 0 && share_env < 0) continue;  // nothing to share at this mode: same world as a lower one
    LaneChoice c;
    if (choose_lanes(w, w->n_cu, &c)) return -1;
    if (c.nw == 0) continue;
    bool better;
    if (best_mode < 0) better = true;
    else if (c.running != best.running) better = c.running > best.running;
    else if (c.resident != best.resident) better = c.resident;
    else if (c.spec != best.spec) better = c.spec;
    else if (c.nw != best.nw) better = c.resident ? c.nw > best.nw : c.nw < best.nw;
    else better = false;  // same geometry: the higher mode (visited first) stays
    if (better) {
      best = c;
      best_mode = mode;
    }
  }
  if (best_mode < 0) return fail("a 64-environment tile of this world does not fit the CU's 160 KiB of LDS");
  if (w->share_mode != best_mode) {
    for (auto& kv : w->scheds) kv.second.release();
    w->scheds.clear();
    build_items(w, best_mode);
  }
  w->lanes = best.nw;
  return 0;
}

static inline int blocks_of(int batch) { return (batch + TILE - 1) / TILE; }
// football's Environment.step (one step per call): one launch (post-step as the epilogue) up to this many tiles per CU (step_env_impl)
constexpr int kFootballOneLaunchTilesPerCu = 1;

// Launch of a generated specialisation G (its schedule is word for word S's): the lean single World.step, the multi-step
// form (several steps and / or substeps per launch) or the fused-environment forms.  0 ok, -1 error.
template <class G, int ENV, class EnvArgs>
static int launch_spec(VmasWorld* w, Sched* S, float* state, float* aft, long ld, const DevStepArgs& a, const EnvArgs& env,
                       size_t extra_lds, hipStream_t s, int batch) {
  size_t lds_spec = ((size_t)G::ROWS * ROWF + SPEC_TAIL_WORDS) * sizeof(float) + extra_lds;
#ifdef VMAS_PROFILE  // occupancy probe (profiling build only): bytes of LDS requested on top of what the kernel uses
  if (const char* pad = getenv("VMAS_DEBUG_LDS_PAD")) lds_spec += (size_t)atol(pad);
#endif
  const bool tail = batch % TILE != 0;
  const int n = a.n_steps > 1 ? a.n_steps : 1;
  const dim3 grid((batch + TILE - 1) / TILE), block(TILE * G::NW);
  if (lds_spec > 64 * 1024) {  // (a fused epilogue's observation staging: opt in to the large LDS once per device)
    static std::atomic<size_t> spec_set_for[64];
    std::atomic<size_t>& set_for = spec_set_for[w->device & 63];
    if (set_for.load() < lds_spec) {
      HIP_TRY(hipFuncSetAttribute((const void*)step_kernel_spec_multi<G, 0, ENV, EnvArgs, false>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_spec));
      HIP_TRY(hipFuncSetAttribute((const void*)step_kernel_spec_multi<G, 1, ENV, EnvArgs, false>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_spec));
      HIP_TRY(hipFuncSetAttribute((const void*)step_kernel_spec_multi<G, 0, ENV, EnvArgs, true>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_spec));
      HIP_TRY(hipFuncSetAttribute((const void*)step_kernel_spec_multi<G, 1, ENV, EnvArgs, true>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_spec));
      set_for = lds_spec;
    }
  }
  if constexpr (ENV == ENV_NONE && G::SUBSTEPS == 1) {
    if (n == 1) {  // the lean form
      if (tail) hipLaunchKernelGGL((step_kernel_spec<G, 1>), grid, block, lds_spec, s, S->dw, state, aft, ld, batch, a.lz);
      else hipLaunchKernelGGL((step_kernel_spec<G, 0>), grid, block, lds_spec, s, S->dw, state, aft, ld, batch, a.lz);
      HIP_TRY(hipGetLastError());
      return 0;
    }
  }
  if (n == 1) {
    if (tail) hipLaunchKernelGGL((step_kernel_spec_multi<G, 1, ENV, EnvArgs, true>), grid, block, lds_spec, s, S->dw, state, aft, ld, batch, 1, 0l, env, a.lz);
    else hipLaunchKernelGGL((step_kernel_spec_multi<G, 0, ENV, EnvArgs, true>), grid, block, lds_spec, s, S->dw, state, aft, ld, batch, 1, 0l, env, a.lz);
  } else {
    if (tail) hipLaunchKernelGGL((step_kernel_spec_multi<G, 1, ENV, EnvArgs, false>), grid, block, lds_spec, s, S->dw, state, aft, ld, batch, n, (long)a.ft_stride, env, a.lz);
    else hipLaunchKernelGGL((step_kernel_spec_multi<G, 0, ENV, EnvArgs, false>), grid, block, lds_spec, s, S->dw, state, aft, ld, batch, n, (long)a.ft_stride, env, a.lz);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

template <int LEVEL, int ENV, class EnvArgs>
static int launch_level(VmasWorld* w, Sched* S, float* state, float* aft, long ld, const DevStepArgs& a,
                        const EnvArgs& env, size_t extra_lds, hipStream_t s, int batch, long pad) {
  // `batch` environments starting at `state` / `aft` (a sub-range of the world's batch when the step is split over two
  // queues); `pad` = columns of the planes that exist from there on (ld minus the range's first environment)
  const size_t lds = S->lds_bytes + extra_lds;
  if (lds > 160 * 1024) return fail("vmas_world_step: %zu bytes of LDS per tile exceed the CU's 160 KB", lds);
  const bool plain = !a.pair_mask && !a.sync && !a.joint_fixed_rot && !a.entity_gravity && a.first_substep == 0 && a.n_substeps <= 0 &&
                     pad >= (long)blocks_of(batch) * TILE && S->dw.items_in_lds;
  const int mode = !plain ? 0 : ((S->dw.substeps == 1 && a.n_steps <= 1) ? (batch % TILE == 0 ? 3 : 2) : 1);
  if (lds > 64 * 1024) {
    static std::atomic<size_t> set_for_dev[64];  // the attribute is per DEVICE (function object of that device's module)
    std::atomic<size_t>& set_for = set_for_dev[w->device & 63];
    if (set_for.load() < lds) {
      HIP_TRY(hipFuncSetAttribute((const void*)step_kernel<LEVEL, ENV, EnvArgs, 0>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      HIP_TRY(hipFuncSetAttribute((const void*)step_kernel<LEVEL, ENV, EnvArgs, 1>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      HIP_TRY(hipFuncSetAttribute((const void*)step_kernel<LEVEL, ENV, EnvArgs, 2>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      HIP_TRY(hipFuncSetAttribute((const void*)step_kernel<LEVEL, ENV, EnvArgs, 3>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      set_for = lds;
    }
  }
  const int blocks = (batch + TILE - 1) / TILE;
  // the world-specialised kernels (csrc/vmas_spec_kernel.h): same results bit for bit, the schedule as compile-time tables
  // (their lazy exact broad phase keeps SPEC_LZP pair words in LDS: worlds of more pairs take the interpreter for it)
  const bool spec_lazy_ok = a.lz.slots == nullptr || a.lz.words <= SPEC_LZP;
  if (plain && S->spec_id >= 0 && w->use_spec && !ABLATE(a) && !a.trace && spec_lazy_ok) {
    bool spec_ok = true;
    if constexpr (ENV != ENV_NONE) spec_ok = env.ingest.n_scripts == 0 && !ABLATE(env);
    if (spec_ok) {
      int rc = 1;  // 1: no specialisation serves this launch
#define VMAS_LAUNCH_SPEC(G)                                                                                              \
      if constexpr (LEVEL == G::LEVEL && (ENV == ENV_NONE || ENV == ENV_INGEST || (ENV == ENV_BALANCE && G::POST == 1) ||  \
                                          (ENV == ENV_TRANSPORT && G::POST == 2) ||                                      \
                                          (ENV == ENV_NAVIGATION && G::POST == 3)))                                      \
        if (rc == 1 && S->spec_id == G::ID)                                                                              \
          rc = launch_spec<G, ENV, EnvArgs>(w, S, state, aft, ld, a, env, extra_lds, s, batch);
      VMAS_SPEC_LIST(VMAS_LAUNCH_SPEC)
#undef VMAS_LAUNCH_SPEC
      if (rc <= 0) return rc;
    }
  }
  // ... or the specialisation compiled at run time for this very schedule (vmas_world_load_spec)
  if (plain && S->rt.ok && w->use_spec && !ABLATE(a) && !a.trace && spec_lazy_ok) {
    bool rt_ok = true;
    if constexpr (ENV != ENV_NONE) rt_ok = env.ingest.n_scripts == 0 && !ABLATE(env);
    if constexpr (ENV == ENV_BALANCE) rt_ok = rt_ok && S->rt.post == 1;
    if constexpr (ENV == ENV_TRANSPORT) rt_ok = rt_ok && S->rt.post == 2;
    if constexpr (ENV == ENV_NAVIGATION) rt_ok = rt_ok && S->rt.post == 3;
    const size_t lds_rt = ((size_t)S->rows * ROWF + SPEC_TAIL_WORDS) * sizeof(float) + extra_lds;
    if (rt_ok && lds_rt <= 160 * 1024) {
      const int tail = batch % TILE != 0 ? 1 : 0;
      const int n = a.n_steps > 1 ? a.n_steps : 1;
      hipFunction_t fn = nullptr;
      DevWorld Wk = S->dw;
      float* st_ = state;
      float* af_ = aft;
      long ld_ = ld, stride_ = (long)a.ft_stride;
      int batch_ = batch, n_ = n;
      EnvArgs env_ = env;
      LazyArgs lz_ = a.lz;
      void* lean_args[] = {&Wk, &st_, &af_, &ld_, &batch_, &lz_};
      void* multi_args[] = {&Wk, &st_, &af_, &ld_, &batch_, &n_, &stride_, &env_, &lz_};
      void** args_ = multi_args;
      if (ENV == ENV_NONE && S->dw.substeps == 1 && n == 1) {
        fn = S->rt.lean[tail];
        args_ = lean_args;
      } else {
        fn = S->rt.multi[ENV][n == 1 ? 1 : 0][tail];
      }
      if (fn != nullptr) {
        if (lds_rt > 64 * 1024 && S->rt.lds_set[fn] < lds_rt) {  // (a fused epilogue's scratch: opt in to the large LDS)
          HIP_TRY(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_rt));
          S->rt.lds_set[fn] = lds_rt;
        }
        HIP_TRY(hipModuleLaunchKernel(fn, (unsigned)blocks, 1, 1, (unsigned)(TILE * S->nw), 1, 1, (unsigned)lds_rt, s, args_, nullptr));
        return 0;
      }
    }
  }
  if (mode == 3)
    hipLaunchKernelGGL((step_kernel<LEVEL, ENV, EnvArgs, 3>), dim3(blocks), dim3(TILE * S->nw), lds, s, S->dw, state, aft,
                       ld, batch, a, env);
  else if (mode == 2)
    hipLaunchKernelGGL((step_kernel<LEVEL, ENV, EnvArgs, 2>), dim3(blocks), dim3(TILE * S->nw), lds, s, S->dw, state, aft,
                       ld, batch, a, env);
  else if (mode == 1)
    hipLaunchKernelGGL((step_kernel<LEVEL, ENV, EnvArgs, 1>), dim3(blocks), dim3(TILE * S->nw), lds, s, S->dw, state, aft,
                       ld, batch, a, env);
  else
    hipLaunchKernelGGL((step_kernel<LEVEL, ENV, EnvArgs, 0>), dim3(blocks), dim3(TILE * S->nw), lds, s, S->dw, state, aft,
                       ld, batch, a, env);
  HIP_TRY(hipGetLastError());
  return 0;
}

template <int ENV, class EnvArgs>
static int launch_any_level(VmasWorld* w, Sched* S, float* state, float* aft, long ld, const DevStepArgs& a,
                            const EnvArgs& env, size_t extra_lds, hipStream_t s, int batch = -1, long pad = -1) {
  if (batch < 0) { batch = w->batch; pad = ld; }
  switch (w->level) {
    case 0: return launch_level<0, ENV>(w, S, state, aft, ld, a, env, extra_lds, s, batch, pad);
    case 1: return launch_level<1, ENV>(w, S, state, aft, ld, a, env, extra_lds, s, batch, pad);
    default: return launch_level<2, ENV>(w, S, state, aft, ld, a, env, extra_lds, s, batch, pad);
  }
}


// ------------------------------------------------------------------------------------
// the lane-compacted kernel's plan (vmas_compact.h): tile layout, broad-phase units dealt to the waves, the owners' pair
// lists, all as one descriptor blob.  Built once at creation; worlds that do not qualify keep cp.ok == false.
// ------------------------------------------------------------------------------------
static int build_compact(VmasWorld* w) {
  using namespace compact;
  VmasWorld::CompactPlan& C = w->cp;
  C.ok = false;
  const int nE = w->base.nE, nA = w->base.nA, nP = (int)w->pairs.size();
  if (nE > 64 || !w->joints.empty() || nP == 0 || nP > 8191) return 0;
  const VmasEntityDesc* E = w->ents.data();
  for (const VmasPairDesc& P : w->pairs) {
    if (P.type != VMAS_PAIR_SS && P.type != VMAS_PAIR_LS) return 0;
    if (E[P.b].shape != VMAS_SHAPE_SPHERE || (P.type == VMAS_PAIR_LS && E[P.a].shape != VMAS_SHAPE_LINE) ||
        (P.type == VMAS_PAIR_SS && E[P.a].shape != VMAS_SHAPE_SPHERE))
      return 0;
  }
  auto dyn = [&](int e) { return (E[e].flags & (VMAS_F_MOVABLE | VMAS_F_ROTATABLE)) != 0; };
  std::vector<int> owned_of(nE, -1), owned;
  for (int e = 0; e < nE; ++e)
    if (dyn(e)) { owned_of[e] = (int)owned.size(); owned.push_back(e); }
  if (owned.empty() || owned.size() > 255) return 0;
  // the owners' pair lists, in the reference's accumulation order = pair order (core.py:2176-2199)
  std::vector<std::vector<uint16_t>> lists(owned.size());
  std::vector<int> pos_a(nP, -1), pos_b(nP, -1);
  std::vector<bool> in_pair(nE, false);
  for (int p = 0; p < nP; ++p) {
    const VmasPairDesc& P = w->pairs[p];
    in_pair[P.a] = in_pair[P.b] = true;
    if (dyn(P.a)) {  // (bit 14: the entry adds a torque - a line-sphere pair seen from a rotatable line)
      const bool tq = P.type == VMAS_PAIR_LS && (E[P.a].flags & VMAS_F_ROTATABLE);
      pos_a[p] = (int)lists[owned_of[P.a]].size();
      lists[owned_of[P.a]].push_back((uint16_t)(p | (tq ? 0x4000 : 0)));
    }
    if (dyn(P.b)) { pos_b[p] = (int)lists[owned_of[P.b]].size(); lists[owned_of[P.b]].push_back((uint16_t)(p | 0x8000)); }
  }
  size_t max_list = 1;
  for (auto& l : lists) max_list = std::max(max_list, l.size());
  if (max_list > (size_t)LIST_MAX) return 0;
  const int hw = (int)((max_list + 31) / 32);
  // tile layout: a dynamic entity keeps all six rows, a static one that is in a pair its position; then the agent
  // forces; then cos / sin of every line that is in a pair.  (The kernel derives the same offsets from the three masks.)
  compact::DevCompact D{};
  std::vector<int> ent_off(nE, -1), tr_off(nE, -1);
  int rows = 0;
  for (int e = 0; e < nE; ++e) {
    if (dyn(e)) { D.dyn_mask |= 1ull << e; ent_off[e] = rows * ROWF; rows += 6; }
    else if (in_pair[e]) { D.static_mask |= 1ull << e; ent_off[e] = rows * ROWF; rows += 2; }
  }
  const int off_af = rows * ROWF;
  rows += nA * 3;
  const int off_tr = rows * ROWF;
  for (int e = 0; e < nE; ++e)
    if (in_pair[e] && E[e].shape == VMAS_SHAPE_LINE) { D.line_mask |= 1ull << e; tr_off[e] = rows * ROWF; rows += 2; }
  if ((long)rows * ROWF >= 65536) return 0;  // (tile offsets are 16-bit fields)
  // broad-phase units: one "row" entity (a line, or the a-sphere of sphere-sphere pairs) against a RUN of partner spheres
  // that are equally spaced in the tile, have consecutive pair indices and share the threshold (vmas_compact.h)
  struct Unit { int row, type, n, stride_rows, first_off, pair0; float thr, cost; };
  std::vector<Unit> units;
  auto fbits = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u; };
  auto pair_thr = [&](const VmasPairDesc& P) {
    if (P.type == VMAS_PAIR_SS) { const float m = (E[P.a].radius + E[P.b].radius) + 1e-4f; return m * m; }
    return (E[P.b].radius + kLineMinDist) + kSkipSlack;
  };
  for (int p = 0; p < nP;) {
    const VmasPairDesc& P0 = w->pairs[p];
    Unit U{P0.a, P0.type, 1, 0, ent_off[P0.b], p, pair_thr(P0), 0.f};
    int q = p + 1;
    while (q < nP && U.n < UNIT_PARTNERS) {
      const VmasPairDesc& Q = w->pairs[q];
      if (Q.a != P0.a || Q.type != P0.type || fbits(pair_thr(Q)) != fbits(U.thr) || fbits(Q.bound_sum) != fbits(P0.bound_sum)) break;
      const int step = (ent_off[Q.b] - ent_off[w->pairs[q - 1].b]) / ROWF;
      if (step <= 0 || step > 255 || (U.n > 1 && step != U.stride_rows)) break;
      U.stride_rows = step;
      ++U.n;
      ++q;
    }
    U.cost = (P0.type == VMAS_PAIR_LS ? 14.f + 14.f * U.n : 10.f + 9.f * U.n);
    units.push_back(U);
    p = q;
  }
  // waves per tile: 8, more if the dynamic entities or the units need them (a wave owns <= OWN_MAX entities and keeps
  // <= WAVE_UNITS units in its registers)
  // (16 when every tile of the batch has a CU of its own: the tile's dependent chain is what the launch takes then)
  int nw = blocks_of(w->batch) <= w->n_cu ? 16 : 8;
  while (nw < MAX_WAVES && ((int)owned.size() > OWN_MAX * nw || (int)units.size() > WAVE_UNITS * nw)) nw <<= 1;
  if ((int)owned.size() > OWN_MAX * nw || (int)units.size() > WAVE_UNITS * nw) return 0;
  const int own_need = ((int)owned.size() + nw - 1) / nw;
  const int own = own_need <= 1 ? 1 : (own_need <= 2 ? 2 : 4);
  // deal the units to the waves, heaviest first, always to the least loaded wave that still has room
  std::vector<std::vector<int>> of_wave(nw);
  {
    std::vector<int> order(units.size());
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return units[x].cost > units[y].cost; });
    std::vector<float> load(nw, 0.f);
    for (int u : order) {
      int best = -1;
      for (int wv = 0; wv < nw; ++wv)
        if ((int)of_wave[wv].size() < WAVE_UNITS && (best < 0 || load[wv] < load[best])) best = wv;
      of_wave[best].push_back(u);
      load[best] += units[u].cost;
    }
  }
  if (knob("VMAS_DEBUG_SCHED"))
    for (int wv = 0; wv < nw; ++wv) {
      float load = 0.f;
      std::string what;
      for (int u : of_wave[wv]) {
        load += units[u].cost;
        what += (units[u].type == VMAS_PAIR_LS ? " LS" : " SS") + std::to_string(units[u].n);
      }
      fprintf(stderr, "[compact nw=%d] wave %2d owns %d, units%s: cost %.0f\n", nw, wv,
              (int)owned.size() > wv ? ((int)owned.size() - wv + nw - 1) / nw : 0, what.c_str(), load);
    }
  std::vector<uint32_t> blob;
  auto align4 = [&]() { while (blob.size() % 4) blob.push_back(0); };
  D.t_owned = (int)blob.size();
  int list_cursor = 0;
  for (size_t k = 0; k < owned.size(); ++k) {
    const int e = owned[k];
    int n_a = 0;
    for (uint16_t en : lists[k]) n_a += (en & 0x8000) ? 0 : 1;
    uint32_t rec[OWNED_W] = {0};
    rec[0] = (uint32_t)e; rec[1] = (uint32_t)ent_off[e]; rec[2] = (uint32_t)list_cursor; rec[3] = (uint32_t)lists[k].size();
    rec[4] = (uint32_t)n_a; rec[5] = (uint32_t)tr_off[e];
    static_assert(sizeof(DevEntity) == 17 * 4 && OWNED_W >= 8 + 17, "owned record");
    memcpy(rec + 8, &w->dev_ents[e], sizeof(DevEntity));
    blob.insert(blob.end(), rec, rec + OWNED_W);
    list_cursor += (int)lists[k].size();
  }
  align4();
  D.t_lists = (int)blob.size();
  {
    std::vector<uint16_t> flat;
    for (auto& l : lists) flat.insert(flat.end(), l.begin(), l.end());
    if (flat.size() % 2) flat.push_back(0);
    for (size_t i = 0; i < flat.size(); i += 2) blob.push_back((uint32_t)flat[i] | ((uint32_t)flat[i + 1] << 16));
  }
  align4();
  std::vector<uint32_t> unit_words;
  std::vector<std::pair<int, int>> wave_range(nw);
  {
    int cursor = 0;
    for (int wv = 0; wv < nw; ++wv) {
      wave_range[wv].first = cursor;
      for (int u : of_wave[wv]) {
        const Unit& U = units[u];
        unit_words.push_back((uint32_t)ent_off[U.row] | ((uint32_t)(tr_off[U.row] >= 0 ? tr_off[U.row] : 0) << 16));
        unit_words.push_back((uint32_t)U.type | ((uint32_t)U.n << 8) | ((uint32_t)U.stride_rows << 16));
        unit_words.push_back((uint32_t)U.first_off | ((uint32_t)U.pair0 << 16));
        unit_words.push_back(fbits(U.type == VMAS_PAIR_LS ? E[U.row].length / 2.f : 0.f));
        unit_words.push_back(fbits(U.thr));
        unit_words.push_back(fbits(overlap_threshold(w->pairs[U.pair0].bound_sum)));  // (the lazy exact broad phase: circles_overlap)
        ++cursor;
      }
      wave_range[wv].second = cursor;
    }
  }
  D.t_units = (int)blob.size();
  blob.insert(blob.end(), unit_words.begin(), unit_words.end());
  align4();
  D.t_pairs = (int)blob.size();
  for (int p = 0; p < nP; ++p) {
    const VmasPairDesc& P = w->pairs[p];
    blob.push_back((uint32_t)ent_off[P.a] | ((uint32_t)ent_off[P.b] << 16));
    blob.push_back((uint32_t)(tr_off[P.a] >= 0 ? tr_off[P.a] : 0) | ((uint32_t)P.type << 16));
    blob.push_back(fbits(P.type == VMAS_PAIR_SS ? E[P.a].radius + E[P.b].radius : E[P.b].radius + kLineMinDist));
    blob.push_back(fbits(P.type == VMAS_PAIR_LS ? E[P.a].length / 2.f : 0.f));
  }
  align4();
  D.t_pairhm = (int)blob.size();
  for (int p = 0; p < nP; ++p) {
    const VmasPairDesc& P = w->pairs[p];
    const uint32_t ha = dyn(P.a) ? (uint32_t)((owned_of[P.a] << 8) | pos_a[p]) : 0xffffu;
    const uint32_t hb = dyn(P.b) ? (uint32_t)((owned_of[P.b] << 8) | pos_b[p]) : 0xffffu;
    blob.push_back(ha | (hb << 16));
  }
  align4();
  D.t_waves = (int)blob.size();
  for (int wv = 0; wv < nw; ++wv) { blob.push_back((uint32_t)wave_range[wv].first); blob.push_back((uint32_t)wave_range[wv].second); }
  align4();
  D.t_entoff = (int)blob.size();
  {  // the load phase's table (vmas_compact.h): per (batch, wave) the four entities wave + (4 * batch + j) * nw, one word each
    // - flags | first row << 3 | cos row << 13 | entity << 23 - 16-byte aligned inside the blob (one scalar load per batch)
    const int batches = (nE + 4 * nw - 1) / (4 * nw);
    for (int b = 0; b < std::max(batches, 1); ++b)
      for (int wv = 0; wv < nw; ++wv)
        for (int j = 0; j < 4; ++j) {
          const int e = wv + (4 * b + j) * nw;
          uint32_t d = 0;
          if (e < nE && ent_off[e] >= 0)
            d = 1u | (dyn(e) ? 2u : 0u) | (tr_off[e] >= 0 ? 4u : 0u) | ((uint32_t)(ent_off[e] / ROWF) << 3) |
                ((uint32_t)(tr_off[e] >= 0 ? tr_off[e] / ROWF : 0) << 13) | ((uint32_t)e << 23);
          blob.push_back(d);
        }
  }
  align4();
  D.t_bounds = (int)blob.size();
  for (int p = 0; p < nP; ++p) blob.push_back(fbits(w->pairs[p].bound_sum));
  align4();
  D.t_band = (int)blob.size();
  for (int p = 0; p < nP; ++p) blob.push_back(fbits(overlap_threshold(w->pairs[p].bound_sum)));
  align4();
  D.blob_words = (int)blob.size();
  D.off_tr = off_tr;
  D.off_af = off_af;
  D.off_tab = rows * ROWF;
  D.n_owned = (int)owned.size(); D.n_pairs = nP; D.hw = hw;
  D.mask_words = (nP + 31) / 32;
  int dyn_at = D.off_tab + D.blob_words;
  dyn_at = (dyn_at + 3) & ~3;
  D.off_dyn = dyn_at;
  // cnt[4] | hit | ballots (u64) | base | keys[CAP] | contacts[CAP] (float2) | torques[CAP] (worlds with rotatable lines) | xmask
  for (const VmasPairDesc& P : w->pairs)
    if (P.type == VMAS_PAIR_LS && (E[P.a].flags & VMAS_F_ROTATABLE)) D.has_torque = 1;
  size_t dyn_words = 4 + (size_t)((D.n_owned * hw + 1) & ~1) + 2 * (size_t)nP + (size_t)((nP + 1) & ~1) + CAP;
  // (the exact broad phase's words: barrier form xmask | gmask; lazy form [2][overlap | band] | need | collected | mask | flag [4])
  dyn_words += 2 * (size_t)CAP + (D.has_torque ? (size_t)CAP : 0) + 7 * (((size_t)D.mask_words + 3) & ~(size_t)3) + 4;
  C.lds_bytes = ((size_t)dyn_at + dyn_words) * sizeof(float);
  if (knob("VMAS_DEBUG_SCHED"))
    fprintf(stderr, "[compact nw=%d] rows %d, tables %d words, per-substep scratch %zu words, LDS %zu B per tile\n", nw, rows,
            D.blob_words, dyn_words, C.lds_bytes);
  if (C.lds_bytes > 160 * 1024) return 0;
  C.h_blob = blob;
  if (!w->host_only) {
    HIP_TRY(upload(&C.d_blob, blob));
    D.blob = C.d_blob;
    if (D.line_mask & ~D.dyn_mask) {
      HIP_TRY(hipMalloc((void**)&C.d_trig, 64 * sizeof(float4)));
      HIP_TRY(hipMemset(C.d_trig, 0, 64 * sizeof(float4)));
    }
  }
  C.dc = D;
  C.nw = nw;
  C.own = own;
  C.ok = true;
  return 0;
}

// a launch's options as the compacted kernel takes them: everything except the lazy exact broad phase TOGETHER with another option
static bool compact_takes(const DevStepArgs& a) {
  const bool plain = !a.pair_mask && !a.sync && !a.entity_gravity && a.first_substep == 0 && a.n_substeps <= 0;
  return a.lz.slots == nullptr || plain;
}
static bool compact_on(const VmasWorld* w) {
  if (!w->cp.ok || w->host_only) return false;
  if (w->compact_mode == 0) return false;
  if (w->compact_mode == 1) return true;
  return w->n_pairs >= 64;  // the library's choice: dense worlds (football: 165 pairs); small ones have their specialisations
}

// one launch of the compacted kernel on `batch` environments starting at state / aft; `pad` as in launch_level
static int launch_compact(VmasWorld* w, int env_kind, float* state, float* aft, long ld, const DevStepArgs& a, const DevEnv* env,
                          size_t extra_lds, hipStream_t s, int batch, long pad) {
  VmasWorld::CompactPlan& C = w->cp;
  const size_t lds = C.lds_bytes + extra_lds;
  if (lds > 160 * 1024) return fail("vmas_world_step: %zu bytes of LDS per tile exceed the CU's 160 KB", lds);
  const int padded = pad >= (long)blocks_of(batch) * TILE ? 1 : 0;
  if (!C.trig_ready && C.d_trig) {
    // cos / sin of the static lines, once, from environment 0 (the kernel uses an entry only where a tile's rotations equal
    // the one it was made from: a value cache, never stale).  Completed before any launch - on whatever queue - can read
    // it; not under graph capture (no synchronisation there: the kernel computes its own until a later eager launch).
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess) cap = hipStreamCaptureStatusNone;
    if (cap == hipStreamCaptureStatusNone) {
      VmasWorld::CompactPlan& M = w->cp;
      compact::DevCompact fill = M.dc;
      if (vmas_compact_fill_trig(fill, state, ld, M.d_trig, s)) return -1;
      HIP_TRY(hipStreamSynchronize(s));
      M.dc.trig_cache = M.d_trig;
      M.trig_ready = true;
    }
  }
  return vmas_compact_launch(env_kind, C.own, C.nw, lds, w->device, w->base, w->cp.dc, state, aft, ld, batch, padded, a, env, s);
}

// plain physics (no environment stages): the compacted kernel where the world has one and the launch's options allow it
// VmasWorld::CompactAdapt: once per API call that makes plain launches, on the caller's stream BEFORE the call's launches
// (every queue of the previous calls has been joined into it: the count it copies is a function of the states alone)
static int compact_adapt_tick(VmasWorld* w, hipStream_t s, int n_steps) {
  VmasWorld::CompactAdapt& A = w->adapt;
  if (w->compact_mode != -1 || A.d_count == nullptr || !compact_on(w)) return 0;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return 0;  // (a graph keeps its kernel)
  A.in_window += n_steps > 1 ? n_steps : 1;
  if (A.in_window < VmasWorld::CompactAdapt::kWindow) return 0;
  A.in_window = 0;
  if (A.pending) {  // the count as of the previous window's end.  WAITED for, not polled: the choice must depend on the
                    // states alone, never on how far the host has run ahead (reruns stay bitwise identical); the host is
                    // at most two windows of calls ahead of the device here, which keeps the queue full
    HIP_TRY(hipEventSynchronize(A.copied));
    const unsigned long long now = __atomic_load_n(A.h_count, __ATOMIC_RELAXED);
    const long tiles = A.tiles_at_copy - A.tiles_seen;
    if (tiles > 0 && (double)(now - A.count_seen) > VmasWorld::CompactAdapt::kContactsPerTile * (double)tiles) {
      A.backoff = VmasWorld::CompactAdapt::kBackoff;
      ++A.switches;
    }
    A.count_seen = now;
    A.tiles_seen = A.tiles_at_copy;
    A.pending = false;
  }
  HIP_TRY(hipMemcpyAsync(A.h_count, A.d_count, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipEventRecord(A.copied, s));
  A.tiles_at_copy = A.tiles;
  A.pending = true;
  return 0;
}

// The adaptive choice for ONE step (VmasWorld::CompactAdapt): 0 = the interpreter's turn (the tiles were overflowing; counts the
// backoff down by one STEP), 1 = the compacted kernel.  Called once per step - by launch_physics for a whole-batch launch, by
// vmas_world_step_n before the sub-range launches of a step split over several queues.
static int compact_pick_step(VmasWorld* w) {
  if (w->adapt.backoff > 0) {
    --w->adapt.backoff;
    return 0;
  }
  return 1;
}

static int launch_physics(VmasWorld* w, Sched* S, float* state, float* aft, long ld, const DevStepArgs& a, hipStream_t s,
                          int batch = -1, long pad = -1) {
  if (batch < 0) { batch = w->batch; pad = ld; }
  if (compact_on(w) && !a.joint_fixed_rot && !(ABLATE(a) & 0xff) && compact_takes(a)) {
    const bool adaptive = w->compact_mode == -1 && w->adapt.d_count != nullptr;
    bool interpreter_turn;
    if (adaptive && w->adapt.forced >= 0) interpreter_turn = w->adapt.forced == 0;  // (chosen for the whole step by the caller)
    else interpreter_turn = adaptive && compact_pick_step(w) == 0;
    if (!interpreter_turn) {
      DevStepArgs ac = a;
      if (adaptive) ac.contacts = w->adapt.d_count;
      if (adaptive) w->adapt.tiles += (long)blocks_of(batch) * w->base.substeps * (a.n_steps > 1 ? a.n_steps : 1);
      return launch_compact(w, ENV_NONE, state, aft, ld, ac, nullptr, 0, s, batch, pad);
    }
  }
  return launch_any_level<ENV_NONE>(w, S, state, aft, ld, a, NoEnv{}, 0, s, batch, pad);
}

// Which form the reference's broad-phase rule (VmasStepArgs.exact_broad_phase) takes for a whole-batch launch of this world:
//   0 none needed - every pair is sphere-sphere, whose force is exactly 0 wherever the circles do not overlap: evaluating a
//     pair per environment IS the reference's result (core.py:2836) - 1 the lazy form inside the step launch (vmas_env_device.h),
//   2 the grid-barrier form inside the launch (at most one tile per CU), 3 a mask launch + a launch per substep.
// VMAS_EXACT_FORM = lazy | barrier | launches pins one where it is possible (A/B measurements, the tests of the older forms).
enum { EXACT_NONE = 0, EXACT_LAZY = 1, EXACT_BARRIER = 2, EXACT_LAUNCHES = 3 };
static int exact_form(const VmasWorld* w, bool capturing) {
  if (w->n_pairs <= 0 || w->lazy.n_band == 0) return EXACT_NONE;
  static const char* pin = knob("VMAS_EXACT_FORM");
  const bool barrier_ok = blocks_of(w->batch) <= w->n_cu && blocks_of(w->batch) <= 512 && !capturing;
  const bool lazy_ok = w->n_pairs <= kLazyMaxPairs && !capturing && !w->lazy.ss_bound_differs;
  if (pin && !strcmp(pin, "launches")) return EXACT_LAUNCHES;
  if (pin && !strcmp(pin, "barrier")) return barrier_ok ? EXACT_BARRIER : EXACT_LAUNCHES;
  if (lazy_ok) return EXACT_LAZY;
  return barrier_ok ? EXACT_BARRIER : EXACT_LAUNCHES;
}
static bool stream_capturing(hipStream_t s) {
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cap) != hipSuccess) cap = hipStreamCaptureStatusNone;
  return cap != hipStreamCaptureStatusNone;
}
// the lazy form's arguments for a launch of `passes` (steps x substeps) passes; every launch gets a tag of its own
static int lazy_prepare(VmasWorld* w, int passes, LazyArgs* out) {
  VmasWorld::Lazy& L = w->lazy;
  const int words = (w->n_pairs + 31) / 32;
  const int tiles_pad = (blocks_of(w->batch) + 63) & ~63;
  const size_t need = (size_t)passes * words * tiles_pad;
  if (need > L.qwords) {  // (first use, or a longer rollout than any before)
    HIP_TRY(hipDeviceSynchronize());
    (void)hipFree(L.d);
    L.d = nullptr; L.qwords = 0;
    HIP_TRY(hipMalloc((void**)&L.d, need * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(L.d, 0, need * sizeof(unsigned long long)));
    L.qwords = need;
  }
  if (++L.tag == 0u) ++L.tag;  // (a launch that is not made after all wastes a tag: harmless)
  LazyArgs Z{};
  Z.slots = L.d;
  Z.flag = w->d_sync + 1;
  Z.gave_up = w->d_gave_up;
  Z.tag = L.tag;
  Z.words = words;
  Z.tiles_pad = tiles_pad;
  Z.band_words = L.band_words;
  *out = Z;
  return 0;
}

static int step_impl(VmasWorld* w, float* state, float* agent_ft, int64_t ld, const VmasStepArgs* args, void* stream,
                     int n_steps, int64_t ft_stride, DevEnv* env = nullptr, int env_kind = ENV_NONE,
                     size_t scratch_fixed = 0, size_t scratch_per_wave = 0, int env_first = 0, int env_count = -1,
                     int scratch_wave_cap = 1 << 20);

// The side streams of vmas_world_step_n and their fork / join events.  Created - and the streams' hardware queues
// brought up by a first operation - when the world is created or the knob is set, never inside a caller's timed loop
// (stream creation and a stream's first submission cost milliseconds).
static int ensure_queues(VmasWorld* w, int nq) {
  if (nq <= 1) return 0;
  if (!w->ev_fork) HIP_TRY(hipEventCreateWithFlags(&w->ev_fork, hipEventDisableTiming));
  for (int q = 0; q < nq - 1 && q < VmasWorld::MAX_QUEUES - 1; ++q)
    if (!w->side[q]) {
      HIP_TRY(hipStreamCreateWithFlags(&w->side[q], hipStreamNonBlocking));
      HIP_TRY(hipEventCreateWithFlags(&w->ev_join[q], hipEventDisableTiming));
      // first submissions (a copy-engine operation and a kernel dispatch): the queue's set-up cost is paid here
      HIP_TRY(hipMemsetAsync(w->d_exact_mask, 0, sizeof(uint32_t), w->side[q]));
      hipLaunchKernelGGL(math_kernel, dim3(1), dim3(64), 0, w->side[q], VMAS_MATH_SQRT, (const float*)w->d_exact_mask,
                         (const float*)nullptr, (float*)w->d_exact_mask, 1);
      HIP_TRY(hipEventRecord(w->ev_join[q], w->side[q]));
      HIP_TRY(hipStreamSynchronize(w->side[q]));
    }
  return 0;
}

// how many HIP queues a vmas_world_step_n of n_steps steps is spread over (see vmas_world_step_n)
static int queues_for(const VmasWorld* w, int n_steps) {
  const int tiles = blocks_of(w->batch);
  int nq = w->queues;
  if (nq == 0) {  // the library's choice: the second queue pays when a launch is long against the 2.9 us the host needs to
                  // enqueue one - four tiles per CU, or two of a world that runs the interpreter (balance 32768 envs:
                  // interpreter 10.0 -> 8.5 us with two queues, specialised kernel 6.7 us on one, 8.5 on two)
    bool spec = false;
    auto it = w->scheds.find(w->lanes);
    if (it != w->scheds.end()) spec = w->use_spec && it->second.spec_id >= 0 && w->base.substeps == 1;
    nq = (n_steps >= 8 && (tiles >= 4 * w->n_cu || (tiles >= 2 * w->n_cu && !spec))) ? 2 : 1;
  }
  while (nq > 1 && tiles < nq) --nq;  // at least one tile per queue
  return nq;
}

extern "C" {

int vmas_abi_version(void) { return VMAS_ABI_VERSION; }
// a digest of the sources this library was built from (csrc/build.sh writes it into its own small translation unit)
extern "C" const char vmas_build_id_string[];
const char* vmas_build_id(void) { return vmas_build_id_string; }
const char* vmas_last_error(void) { return g_err; }

int vmas_world_create(const VmasWorldDesc* d, int32_t batch, int32_t device_id, VmasWorld** out) {
  if (!d || !out) return fail("vmas_world_create: null argument");
  if (d->abi_version != VMAS_ABI_VERSION)
    return fail("vmas_world_create: desc ABI version %d, library %d", d->abi_version, VMAS_ABI_VERSION);
  if (batch <= 0) return fail("vmas_world_create: batch must be > 0, got %d", batch);
  if (d->n_entities <= 0) return fail("vmas_world_create: world has no entities");
  for (int p = 0; p < d->n_pairs; ++p)
    if (d->pairs[p].a < 0 || d->pairs[p].a >= d->n_entities || d->pairs[p].b < 0 || d->pairs[p].b >= d->n_entities ||
        d->pairs[p].type < 0 || d->pairs[p].type > VMAS_PAIR_BB)
      return fail("vmas_world_create: pair %d is malformed", p);
  for (int j = 0; j < d->n_joints; ++j)
    if (d->joints[j].a < 0 || d->joints[j].a >= d->n_entities || d->joints[j].b < 0 || d->joints[j].b >= d->n_entities)
      return fail("vmas_world_create: joint %d is malformed", j);
  const bool host_only = device_id == -1;  // planning world (include/vmas_debug_hip.h): no device is touched
  if (!host_only) {
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device_id < 0 || device_id >= ndev) return fail("vmas_world_create: device %d of %d", device_id, ndev);
    HIP_TRY(hipSetDevice(device_id));
  }

  VmasWorld* w = new VmasWorld();
  struct Guard {  // destroys the half-built world on every early return below
    VmasWorld* w;
    ~Guard() { if (w) vmas_world_destroy(w); }
  } guard{w};
  w->device = device_id;
  w->host_only = host_only;
  w->batch = batch;
  w->ents.assign(d->entities, d->entities + d->n_entities);
  w->pairs.assign(d->pairs, d->pairs + d->n_pairs);
  w->joints.assign(d->joints, d->joints + d->n_joints);
  w->n_pairs = d->n_pairs;

  std::vector<DevEntity> ents(d->n_entities);
  std::vector<int> tr_row(d->n_entities, -1);
  for (int e = 0; e < d->n_entities; ++e) {
    const VmasEntityDesc& s = d->entities[e];
    DevEntity& t = ents[e];
    t.flags = s.flags; t.shape = s.shape; t.agent_index = s.agent_index;
    t.mass = s.mass; t.inertia = s.inertia; t.one_minus_drag = s.one_minus_drag;
    t.max_speed = s.max_speed; t.v_range = s.v_range;
    t.lin_friction = s.lin_friction; t.ang_friction = s.ang_friction;
    t.gx = s.gravity[0]; t.gy = s.gravity[1];
    t.max_f = s.max_f; t.f_range = s.f_range; t.max_t = s.max_t; t.t_range = s.t_range;
    t.tr_off = -1;
    if (s.shape != VMAS_SHAPE_SPHERE) {
      tr_row[e] = 4 * (int)w->trig_ents.size();
      w->trig_ents.push_back(e);
    }
    if (s.flags & (VMAS_F_MOVABLE | VMAS_F_ROTATABLE)) w->n_dyn++;
    if ((s.flags & VMAS_F_AGENT) && (s.agent_index < 0 || s.agent_index >= d->n_agents)) {
      return fail("vmas_world_create: entity %d has agent_index %d outside [0,%d)", e, s.agent_index, d->n_agents);
    }
  }
  DevWorld& W = w->base;
  W.nE = d->n_entities; W.nA = d->n_agents; W.substeps = d->substeps; W.sub_dt = d->sub_dt;
  W.gx = d->gravity[0]; W.gy = d->gravity[1]; W.has_gravity = d->has_gravity;
  W.xs = d->x_semidim; W.ys = d->y_semidim;
  W.k = d->contact_margin; W.tcf = d->torque_constraint_force;
  W.c_coll = d->collision_force;    // sign = +1
  W.c_joint_att = -d->joint_force;  // sign = -1
  W.c_joint_rep = d->joint_force;
  W.off_af = W.nE * 6 * ROWF;
  w->row_tr = W.nE * 6 + W.nA * 3;
  for (int e = 0; e < d->n_entities; ++e)
    if (tr_row[e] >= 0) ents[e].tr_off = (w->row_tr + tr_row[e]) * ROWF;
  W.row_tr = w->row_tr;
  W.trig_in_args = d->n_entities <= 64;
  W.trig_mask = W.box_mask = 0ull;
  for (int e : w->trig_ents)
    if (W.trig_in_args) {
      W.trig_mask |= 1ull << e;
      if (d->entities[e].shape == VMAS_SHAPE_BOX) W.box_mask |= 1ull << e;
    }
  w->tr_row = tr_row;
  std::vector<DevMaskPair> mp(d->n_pairs);
  for (int p = 0; p < d->n_pairs; ++p) mp[p] = {d->pairs[p].a, d->pairs[p].b, d->pairs[p].bound_sum};
  for (int p = 0; p < d->n_pairs; ++p) {
    if (band_capable(d->pairs[p].type)) w->lazy.n_band++;
    // (every pair's word is exchanged: a sphere pair has band events too - a non-finite pose, whose NaN force the reference
    //  only lets through if some environment overlaps; the packed sphere records test them against r_a + r_b, which must
    //  then BE the pair's bounding-circle sum)
    if (p < 64 * 32) w->lazy.band_words |= 1ull << (p >> 5);
    if (d->pairs[p].type == VMAS_PAIR_SS) {
      const float rs = d->entities[d->pairs[p].a].radius + d->entities[d->pairs[p].b].radius;
      if (memcmp(&rs, &d->pairs[p].bound_sum, 4) != 0) w->lazy.ss_bound_differs = true;
    }
  }
  w->dev_ents = ents;
  if (!host_only) HIP_TRY(upload(&w->d_mpairs, mp));
  if (!host_only) {  // exact broad phase: barrier word + four mask slots (in-kernel form), one mask (launch-per-substep form)
    const size_t mw = (size_t)(d->n_pairs + 31) / 32;
    // (unused | gave-up flag | pad, then a ring of four slots of [16 tile groups][pair words] 64-bit words: grid_bits_publish)
    HIP_TRY(hipMalloc((void**)&w->d_sync, (4 + 4 * mw * 16 * 2 + 2) * sizeof(uint32_t)));
    HIP_TRY(hipMemset(w->d_sync, 0, (4 + 4 * mw * 16 * 2 + 2) * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void**)&w->d_exact_mask, (mw ? mw : 1) * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void**)&w->d_nav_mask, (3 * mw + 1) * sizeof(uint32_t)));
    HIP_TRY(hipMemset(w->d_nav_mask, 0, (3 * mw + 1) * sizeof(uint32_t)));
    // navigation epilogue's grid barrier: unused | timeout flag | ring of four slots of [pair words][tile groups of 32] 64-bit
    // words (arrival bits | pair bits: navigation_post_tile); tile groups for the largest grid that uses it (one tile per CU)
    const size_t nav_groups = 16;  // (grids of up to 512 tiles: the barrier form is used at one tile per CU at most)
    HIP_TRY(hipMalloc((void**)&w->d_nav_sync, (2 + 4 * mw * nav_groups * 2 + 2) * sizeof(uint32_t)));
    HIP_TRY(hipMemset(w->d_nav_sync, 0, (2 + 4 * mw * nav_groups * 2 + 2) * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void**)&w->adapt.d_count, sizeof(unsigned long long)));
    HIP_TRY(hipMemset(w->adapt.d_count, 0, sizeof(unsigned long long)));
    HIP_TRY(hipHostMalloc((void**)&w->adapt.h_count, 64, hipHostMallocDefault));
    *w->adapt.h_count = 0ull;
    HIP_TRY(hipEventCreateWithFlags(&w->adapt.copied, hipEventDisableTiming));
    HIP_TRY(hipHostMalloc((void**)&w->h_gave_up, 64, hipHostMallocMapped | hipHostMallocCoherent));
    *w->h_gave_up = 0u;
    HIP_TRY(hipHostGetDevicePointer((void**)&w->d_gave_up, w->h_gave_up, 0));
  }
  if (!host_only) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0) w->n_cu = prop.multiProcessorCount;
  }
  if (select_config(w)) return -1;
  if (build_compact(w)) return -1;
  Sched* S = nullptr;
  if (get_sched(w, w->lanes, &S)) return -1;
  if (S->lds_bytes > 160 * 1024) {
    return fail("vmas_world_create: a 64-environment tile of this world needs %zu B of LDS (> 160 KiB)", S->lds_bytes);
  }
  if (!host_only && ensure_queues(w, queues_for(w, 1 << 20))) return -1;  // the side queue of vmas_world_step_n
  guard.w = nullptr;
  *out = w;
  return 0;
}

void vmas_world_destroy(VmasWorld* w) {
  if (!w) return;
  if (w->host_only) { delete w; return; }
  (void)hipSetDevice(w->device);
  for (auto& kv : w->scheds) kv.second.release();
  for (int q = 0; q < VmasWorld::MAX_QUEUES - 1; ++q) {
    if (w->side[q]) { (void)hipStreamSynchronize(w->side[q]); (void)hipStreamDestroy(w->side[q]); }
    if (w->ev_join[q]) (void)hipEventDestroy(w->ev_join[q]);
  }
  if (w->ev_fork) (void)hipEventDestroy(w->ev_fork);
  (void)hipFree(w->lazy.d);
  (void)hipFree(w->d_sync); (void)hipFree(w->d_exact_mask); (void)hipFree(w->d_nav_mask); (void)hipFree(w->d_nav_sync);
  if (w->h_gave_up) (void)hipHostFree(w->h_gave_up);
  (void)hipFree(w->lc.d_ent_slot); (void)hipFree(w->lc.d_slot_ent);
  (void)hipFree(w->adapt.d_count);
  if (w->adapt.h_count) (void)hipHostFree(w->adapt.h_count);
  if (w->adapt.copied) (void)hipEventDestroy(w->adapt.copied);
  (void)hipFree(w->cp.d_blob); (void)hipFree(w->cp.d_trig);
  (void)hipFree(w->d_mpairs); (void)hipFree(w->d_trace);
  (void)hipFree(w->d_lidars); (void)hipFree(w->d_targets); (void)hipFree(w->d_angles); (void)hipFree(w->d_queries);
  (void)hipFree(w->d_angles_cs);
  delete w;
}

int vmas_world_set_lanes_per_env(VmasWorld* w, int32_t lanes) {
  if (!w) return fail("vmas_world_set_lanes_per_env: null world");
  if (lanes == 0) {
    LaneChoice c;
    if (choose_lanes(w, w->n_cu, &c)) return -1;
    if (c.nw == 0) return fail("vmas_world_set_lanes_per_env: no waves-per-tile setting fits the LDS");
    lanes = c.nw;
  }
  const int max_w = w->level >= 2 ? 8 : MAX_WAVES;
  if (lanes < 1 || lanes > max_w) return fail("lanes_per_env must be in 1..%d for this world, got %d", max_w, lanes);
  if (!w->host_only) HIP_TRY(hipSetDevice(w->device));
  Sched* S;
  if (get_sched(w, lanes, &S)) return -1;
  if (S->lds_bytes > 160 * 1024 && w->n_shared_rows > 0 && (unshare(w) || get_sched(w, lanes, &S))) return -1;
  if (S->lds_bytes > 160 * 1024)
    return fail("lanes_per_env=%d needs %zu B of LDS per tile (> 160 KiB)", lanes, S->lds_bytes);
  w->lanes = lanes;
  return 0;
}
int vmas_world_get_lanes_per_env(const VmasWorld* w) { return w ? w->lanes : -1; }

int vmas_world_reserve_epilogue(VmasWorld* w, int32_t post_kind, int32_t n_packages) {
  if (!w) return fail("vmas_world_reserve_epilogue: null world");
  size_t f0 = 0, f1 = 0;
  if (post_kind == VMAS_POST_BALANCE) {
    f0 = balance_scratch_floats(0); f1 = balance_scratch_floats(1);
  } else if (post_kind == VMAS_POST_TRANSPORT) {
    if (n_packages < 1) return fail("vmas_world_reserve_epilogue: transport needs n_packages >= 1, got %d", n_packages);
    f0 = transport_scratch_floats(0, n_packages); f1 = transport_scratch_floats(1, n_packages);
  } else if (post_kind != VMAS_POST_NONE) {
    return fail("vmas_world_reserve_epilogue: post_kind %d has no fused epilogue", post_kind);
  }
  if (!w->host_only) HIP_TRY(hipSetDevice(w->device));
  w->reserve_fixed = f0 * sizeof(float);
  w->reserve_per_wave = (f1 - f0) * sizeof(float);
  return select_config(w);
}

int64_t vmas_world_step_bytes_per_env(const VmasWorld* w) {
  if (!w) return -1;
  return 24LL * w->base.nE + 12LL * w->base.nA + 24LL * w->n_dyn;
}

int vmas_world_set_specialized(VmasWorld* w, int32_t on) {
  if (!w) return fail("vmas_world_set_specialized: null world");
  w->use_spec = on != 0;
  return 0;
}
int vmas_world_get_specialized(VmasWorld* w) {  // 1: plain World.step launches of this world run a generated specialisation
  if (!w) return 0;
  Sched* S;
  if (get_sched(w, w->lanes, &S)) return 0;
  return (w->use_spec && (S->spec_id >= 0 || S->rt.ok)) ? 1 : 0;
}

int vmas_world_load_spec(VmasWorld* w, const char* code_object_path) {
  if (!w || !code_object_path) return fail("vmas_world_load_spec: null argument");
  if (w->host_only) return fail("vmas_world_load_spec: a planning world (device -1) has no device side");
  HIP_TRY(hipSetDevice(w->device));
  Sched* S;
  if (get_sched(w, w->lanes, &S)) return -1;
  if (S->rt.mod) { (void)hipModuleUnload(S->rt.mod); S->rt = Sched::Rt{}; }
  hipModule_t mod = nullptr;
  const hipError_t load_rc = hipModuleLoad(&mod, code_object_path);
  if (load_rc != hipSuccess) {
    (void)hipGetLastError();  // (the refusal is this call's result: not the next launch check's, whoever makes it)
    // (the caller keeps or drops its cached file by this wording: only the second case says anything about the FILE)
    if (load_rc == hipErrorOutOfMemory || load_rc == hipErrorMemoryAllocation)
      return fail("vmas_world_load_spec: cannot load %s: out of device memory", code_object_path);
    return fail("vmas_world_load_spec: %s is not a loadable gfx950 code object (%s)", code_object_path, hipGetErrorString(load_rc));
  }
  struct Unload { hipModule_t m; ~Unload() { if (m) (void)hipModuleUnload(m); } } guard{mod};
  // the tables the module was generated from must be, word for word, the schedule this world runs
  hipDeviceptr_t dptr = nullptr;
  size_t bytes = 0;
  if (hipModuleGetGlobal(&dptr, &bytes, mod, "vmas_rt_check") != hipSuccess) (void)hipGetLastError();
  if (dptr == nullptr || bytes < 26 * 4) return fail("vmas_world_load_spec: %s has no vmas_rt_check", code_object_path);
  std::vector<uint32_t> got(bytes / 4);
  HIP_TRY(hipMemcpy(got.data(), dptr, bytes, hipMemcpyDeviceToHost));
  const DevWorld& D = S->dw;
  const uint32_t meta[23] = {(uint32_t)S->nw, (uint32_t)w->share_mode, (uint32_t)w->level, (uint32_t)S->h_blob.size(), (uint32_t)D.b_ent,
                             (uint32_t)D.b_segs, (uint32_t)D.b_owned, (uint32_t)D.b_refs, (uint32_t)D.b_items, (uint32_t)D.n_segs,
                             (uint32_t)D.n_owned, (uint32_t)S->rows, (uint32_t)D.fired_recs, (uint32_t)D.items_in_lds, (uint32_t)D.nE,
                             (uint32_t)D.nA, (uint32_t)D.off_af, (uint32_t)D.row_tr, (uint32_t)(D.trig_mask & 0xffffffffu),
                             (uint32_t)(D.trig_mask >> 32), (uint32_t)(D.box_mask & 0xffffffffu), (uint32_t)(D.box_mask >> 32),
                             (uint32_t)D.trig_in_args};
  if (got.size() != 26 + S->h_blob.size() || memcmp(got.data(), meta, sizeof(meta)) != 0 || got[23] != (uint32_t)D.substeps ||
      memcmp(got.data() + 26, S->h_blob.data(), S->h_blob.size() * sizeof(uint32_t)) != 0)
    return fail("vmas_world_load_spec: %s was generated from another schedule (world, batch geometry or library version)",
                code_object_path);
  if (got[25] != kLayoutHash)
    return fail("vmas_world_load_spec: %s was compiled against other kernel-argument layouts than this library (headers on disk "
                "out of step with the built libvmas_hip.so: rebuild it)", code_object_path);
  Sched::Rt rt;
  rt.post = (int)got[24];
  char name[64];
  for (int t = 0; t < 2; ++t) {
    snprintf(name, sizeof(name), "vmas_rt_lean_t%d", t);
    if (hipModuleGetFunction(&rt.lean[t], mod, name) != hipSuccess) rt.lean[t] = nullptr;
    for (int e = 0; e < 5; ++e)
      for (int o = 0; o < 2; ++o) {
        snprintf(name, sizeof(name), "vmas_rt_multi_e%d_o%d_t%d", e, o, t);
        if (hipModuleGetFunction(&rt.multi[e][o][t], mod, name) != hipSuccess) rt.multi[e][o][t] = nullptr;
      }
  }
  (void)hipGetLastError();  // (missing forms are expected: they run the interpreter)
  rt.mod = mod;
  rt.ok = true;
  guard.m = nullptr;
  S->rt = rt;
  return 0;
}

int vmas_world_set_compact(VmasWorld* w, int32_t mode) {
  if (!w) return fail("vmas_world_set_compact: null world");
  if (mode < -1 || mode > 1) return fail("vmas_world_set_compact: mode must be -1 (library's choice), 0 (off) or 1 (on), got %d", mode);
  w->compact_mode = mode;
  return 0;
}
int vmas_world_get_compact(VmasWorld* w) { return (w && compact_on(w)) ? 1 : 0; }

int vmas_world_exact_status(VmasWorld* w) {
  if (!w) return fail("vmas_world_exact_status: null world");
  if (!w->d_sync) return 0;
  uint32_t flag = 0;
  HIP_TRY(hipSetDevice(w->device));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(&flag, w->d_sync + 1, sizeof(flag), hipMemcpyDeviceToHost));
  uint32_t nav_flag = 0;  // (the navigation epilogue's barrier: same rule, same report)
  if (w->d_nav_sync) HIP_TRY(hipMemcpy(&nav_flag, w->d_nav_sync + 1, sizeof(nav_flag), hipMemcpyDeviceToHost));
  const uint32_t host_flag = w->h_gave_up ? __atomic_load_n(w->h_gave_up, __ATOMIC_RELAXED) : 0u;
  return (int)((flag | nav_flag | host_flag) != 0u);
}

int vmas_world_exact_form(VmasWorld* w) {
  if (!w) return fail("vmas_world_exact_form: null world");
  return exact_form(w, false);
}

int vmas_world_set_queues(VmasWorld* w, int32_t queues) {
  if (!w) return fail("vmas_world_set_queues: null world");
  if (queues < 0 || queues > VmasWorld::MAX_QUEUES)
    return fail("vmas_world_set_queues: queues must be 0 (library's choice) .. %d, got %d", VmasWorld::MAX_QUEUES, queues);
  w->queues = queues;
  if (w->host_only) return 0;
  HIP_TRY(hipSetDevice(w->device));
  return ensure_queues(w, queues_for(w, 1 << 20));
}
int vmas_world_get_queues(const VmasWorld* w, int32_t n_steps) { return w ? queues_for(w, n_steps) : 0; }

int vmas_world_step(VmasWorld* w, float* state, float* agent_ft, int64_t ld, const VmasStepArgs* args, void* stream) {
  return step_impl(w, state, agent_ft, ld, args, stream, 1, 0);
}

int vmas_world_rollout(VmasWorld* w, float* state, float* agent_ft, int64_t ld, int64_t ft_step_stride, int32_t n_steps,
                       const VmasStepArgs* args, void* stream) {
  if (n_steps <= 0) return fail("vmas_world_rollout: n_steps must be > 0, got %d", n_steps);
  if (args && (args->first_substep != 0 || args->n_substeps > 0))
    return fail("vmas_world_rollout: partial substep ranges are only valid for single steps");
  return step_impl(w, state, agent_ft, ld, args, stream, n_steps, ft_step_stride);
}

static int step_env_impl(VmasWorld* w, float* state, float* agent_ft, int64_t ld, const VmasStepArgs* args,
                         const VmasIngestArgs* ingest, uint32_t* err_flags, int32_t post_kind, const void* post_desc,
                         const void* post_buffers, int32_t n_steps, void* stream, int gated = 0);

// LDS (bytes) a fused epilogue needs behind the tile: `fixed` + `per_wave` floats for each of min(waves per tile, cap)
// waves.  If the tile plus that does not fit the CU, the shared pair rows are given up once (the schedule is rebuilt).
static int env_extra_lds(VmasWorld* w, Sched** S, size_t fixed, size_t per_wave, int cap, size_t* extra) {
  auto need = [&]() { return (fixed + per_wave * (size_t)std::min((*S)->nw, cap)) * sizeof(float); };
  *extra = need();
  if ((*S)->lds_bytes + *extra > 160 * 1024 && w->n_shared_rows > 0) {
    if (unshare(w) || get_sched(w, w->lanes, S)) return -1;
    *extra = need();
  }
  if ((*S)->lds_bytes + *extra > 160 * 1024)
    return fail("vmas_world_step_env: %zu bytes of LDS per tile with this epilogue exceed the CU's 160 KB",
                (*S)->lds_bytes + *extra);
  return 0;
}

// navigation as an epilogue: argument checks against the world and its registered sensors, LDS need
static int nav_epilogue_plan(VmasWorld* w, const VmasNavigationDesc* d, size_t* fixed, size_t* per_wave) {
  if (d->n_agents < 1 || d->n_agents > VMAS_ENV_MAX_AGENTS) return fail("vmas_world_step_env: navigation n_agents out of range");
  if (d->agent0 < 0 || d->agent0 + d->n_agents > w->base.nE) return fail("vmas_world_step_env: navigation agents out of range");
  for (int a = 0; a < d->n_agents; ++a)
    if (d->goal_of[a] < 0 || d->goal_of[a] >= w->base.nE) return fail("vmas_world_step_env: navigation goal %d out of range", a);
  if (d->collisions) {  // the epilogue casts sensor a = agent a's on the other agents: must be what the world has registered
    if (w->n_lidars != d->n_agents) return fail("vmas_world_step_env: navigation with collisions needs one registered sensor per agent");
    for (int a = 0; a < d->n_agents; ++a) {
      const DevLidar& L = w->h_lidars[a];
      if (L.entity != d->agent0 + a || L.n_rays != d->n_rays || L.angle_off != a * d->n_rays || L.max_range != d->lidar_range ||
          L.n_targets != d->n_agents - 1)
        return fail("vmas_world_step_env: sensor %d is not agent %d's %d-ray LIDAR of range %g on the other agents", a, a,
                    d->n_rays, (double)d->lidar_range);
      for (int t = 0; t < L.n_targets; ++t) {
        const DevTarget& T = w->h_targets[L.target_off + t];
        if (T.entity != d->agent0 + (t < a ? t : t + 1) || T.shape != VMAS_SHAPE_SPHERE || T.radius != d->agent_radius)
          return fail("vmas_world_step_env: target %d of sensor %d is not the %d-th other agent (a sphere of the agents' radius)",
                      t, a, t);
      }
    }
  }
  if (d->collisions && (d->n_rays < 1 || d->n_rays > 64 || (w->n_pairs + 31) / 32 >= VMAS_ENV_MAX_AGENTS))
    return fail("vmas_world_step_env: navigation with collisions needs 1 <= n_rays <= 64 and at most %d collidable pairs",
                32 * (VMAS_ENV_MAX_AGENTS - 1));
  const int D = navigation_obs_dim(*d);
  *fixed = navigation_scratch_floats(0, d->n_agents, D, d->collisions ? d->n_agents * d->n_rays : 0, d->collisions ? w->n_pairs : 0);
  *per_wave = navigation_scratch_floats(1, d->n_agents, D, d->collisions ? d->n_agents * d->n_rays : 0) -
              navigation_scratch_floats(0, d->n_agents, D, d->collisions ? d->n_agents * d->n_rays : 0);
  return 0;
}

int vmas_world_step_env_check(VmasWorld* w, int32_t post_kind, const void* post_desc) {
  if (!w || !post_desc) return fail("vmas_world_step_env_check: null argument");
  if (w->host_only) return fail("vmas_world_step_env_check: a planning world (device -1) cannot be stepped");
  if (post_kind == VMAS_POST_FOOTBALL) {  // the compacted kernel runs this world (its epilogue needs no LDS of its own)
    if (!compact_on(w)) return fail("vmas_world_step_env_check: the football epilogue needs the lane-compacted step kernel");
    return 0;
  }
  if (post_kind != VMAS_POST_NAVIGATION) return fail("vmas_world_step_env_check: post_kind %d", post_kind);
  const auto* d = (const VmasNavigationDesc*)post_desc;
  size_t fixed = 0, per_wave = 0, extra = 0;
  if (nav_epilogue_plan(w, d, &fixed, &per_wave)) return -1;
  Sched* S;
  if (get_sched(w, w->lanes, &S)) return -1;
  if (env_extra_lds(w, &S, fixed, per_wave, 1 << 20, &extra)) return -1;  // (every wave has its columns: see navigation_post_body)
  if (d->n_agents > kNavMaxOwn * S->nw)
    return fail("vmas_world_step_env: %d agents on %d waves per tile (at most %d agents per wave)", d->n_agents, S->nw, kNavMaxOwn);
  return 0;
}

int vmas_world_step_env(VmasWorld* w, float* state, float* agent_ft, int64_t ld, const VmasStepArgs* args,
                        const VmasIngestArgs* ingest, uint32_t* err_flags, int32_t post_kind, const void* post_desc,
                        const void* post_buffers, void* stream) {
  return step_env_impl(w, state, agent_ft, ld, args, ingest, err_flags, post_kind, post_desc, post_buffers, 1, stream);
}

int vmas_world_step_env_gated(VmasWorld* w, float* state, float* agent_ft, int64_t ld, const VmasStepArgs* args,
                              const VmasIngestArgs* ingest, uint32_t* gate, int32_t post_kind, const void* post_desc,
                              const void* post_buffers, void* stream) {
  return step_env_impl(w, state, agent_ft, ld, args, ingest, gate, post_kind, post_desc, post_buffers, 1, stream, 1);
}

int vmas_world_gated_refused(VmasWorld* w) {
  if (!w) return fail("vmas_world_gated_refused: null world");
  w->nav_seq -= w->gated_nav_seq;
  if (w->gated_nav_flip) w->nav_flip ^= 1;
  w->gated_nav_seq = 0; w->gated_nav_flip = 0;
  return 0;
}

int vmas_world_rollout_env(VmasWorld* w, float* state, float* agent_ft, int64_t ld, const VmasStepArgs* args,
                           const VmasIngestArgs* ingest, uint32_t* err_flags, int32_t post_kind, const void* post_desc,
                           const void* post_buffers, int32_t n_steps, void* stream) {
  if (n_steps <= 0) return fail("vmas_world_rollout_env: n_steps must be > 0, got %d", n_steps);
  if (!ingest) return fail("vmas_world_rollout_env: the steps' actions come through `ingest`");
  if (ingest->n_scripts > 0 && n_steps > 1 && !(w && compact_on(w)))  // (the compacted kernel runs the scripts on its tile)
    return fail("vmas_world_rollout_env: scripted agents read the state in HBM, which a multi-step launch does not refresh");
  return step_env_impl(w, state, agent_ft, ld, args, ingest, err_flags, post_kind, post_desc, post_buffers, n_steps, stream);
}

static int step_env_impl(VmasWorld* w, float* state, float* agent_ft, int64_t ld, const VmasStepArgs* args,
                         const VmasIngestArgs* ingest, uint32_t* err_flags, int32_t post_kind, const void* post_desc,
                         const void* post_buffers, int32_t n_steps, void* stream, int gated) {
  if (!w) return fail("vmas_world_step_env: null world");
  if (gated) {
    // A gated launch must be the WHOLE step: kinds whose step can be more than one launch (football beyond one tile per CU,
    // navigation's collision kernel) or carries a grid barrier (its sequence numbers advance on the host whether or not the
    // tiles arrive: the exact broad phase, navigation's collision reduction) are refused.
    if (!err_flags) return fail("vmas_world_step_env_gated: needs the gate word");
    if (n_steps != 1) return fail("vmas_world_step_env_gated: single steps only");
    // (round 6: every launch of a step reads the gate - the step kernel, navigation's collision kernel, football's post-step
    //  kernel - and what the host advances for a launch that arrives at a grid barrier or fills a mask is taken back by
    //  vmas_world_gated_refused when the caller learns that the gate was shut)
    w->gated_nav_seq = 0; w->gated_nav_flip = 0;
    if (args && args->exact_broad_phase && exact_form(w, stream_capturing((hipStream_t)stream)) > EXACT_LAZY)
      return fail("vmas_world_step_env_gated: this world's exact broad phase carries a grid barrier / several launches here and "
                  "cannot be gated (vmas_world_exact_form)");
  }
  if (args && (args->first_substep != 0 || args->n_substeps > 0))
    return fail("vmas_world_step_env: partial substep ranges cannot carry an epilogue");
  if (post_kind != VMAS_POST_NONE && (!post_desc || !post_buffers))
    return fail("vmas_world_step_env: null post-step descriptor");
  if (post_kind == VMAS_POST_NONE && !ingest) return fail("vmas_world_step_env: neither actions nor a post-step given");
  DevEnv env{};
  static const int env_ablate = knob("VMAS_ENV_ABLATE") ? atoi(knob("VMAS_ENV_ABLATE")) : 0;
  env.ablate = env_ablate;
  env.err_flags = err_flags;
  env.gated = gated ? 1 : 0;
  for (int a = 0; a < VMAS_ENV_MAX_AGENTS; ++a) env.script_of_agent[a] = -1;
  if (ingest) {
    if (vmas::check_ingest_args(ingest, w->batch, agent_ft, ld)) return -1;
    env.has_ingest = 1;
    env.ingest.clamp = ingest->clamp;
    env.ingest.n_agents = w->base.nA;
    env.ingest.n_scripts = ingest->n_scripts;
    for (int i = 0; i < ingest->n_agents; ++i) {  // slot of agent a at index a (zero-initialised slots: no action)
      const int a = ingest->agents[i].agent_index;
      if (a >= w->base.nA) return fail("vmas_world_step_env: action slot %d names agent %d of %d", i, a, w->base.nA);
      env.ingest.agents[a] = ingest->agents[i];
    }
    for (int i = 0; i < ingest->n_scripts; ++i) {
      const int a = ingest->scripts[i].agent_index;
      if (a >= w->base.nA || ingest->scripts[i].entity >= w->base.nE)
        return fail("vmas_world_step_env: agent script %d names agent %d / entity %d", i, a, ingest->scripts[i].entity);
      env.ingest.scripts[i] = ingest->scripts[i];
      env.script_of_agent[a] = (int8_t)i;
    }
  }
  if (post_kind == VMAS_POST_BALANCE) {
    const auto* d = (const VmasBalanceDesc*)post_desc;
    const auto* o = (const VmasBalanceBuffers*)post_buffers;
    if (vmas::check_balance_args(d, o, w->batch, state, ld, w->base.nE)) return -1;
    env.balance.d = *d;
    env.balance.o = *o;
    return step_impl(w, state, agent_ft, ld, args, stream, n_steps, 0, &env, ENV_BALANCE, balance_scratch_floats(0),
                     balance_scratch_floats(1) - balance_scratch_floats(0));
  }
  if (post_kind == VMAS_POST_TRANSPORT) {
    const auto* d = (const VmasTransportDesc*)post_desc;
    const auto* o = (const VmasTransportBuffers*)post_buffers;
    if (vmas::check_transport_args(d, o, w->batch, state, ld, w->base.nE)) return -1;
    env.transport.d = *d;
    env.transport.o = *o;
    return step_impl(w, state, agent_ft, ld, args, stream, n_steps, 0, &env, ENV_TRANSPORT,
                     transport_scratch_floats(0, d->n_packages),
                     transport_scratch_floats(1, d->n_packages) - transport_scratch_floats(0, d->n_packages));
  }
  if (post_kind == VMAS_POST_NAVIGATION) {
    const auto* d = (const VmasNavigationDesc*)post_desc;
    const auto* o = (const VmasNavigationBuffers*)post_buffers;
    if (vmas::check_navigation_args(d, o, w->batch, state, ld, 1)) return -1;
    size_t nav_fixed = 0, nav_per_wave = 0;
    if (nav_epilogue_plan(w, d, &nav_fixed, &nav_per_wave)) return -1;
    env.navigation.d = *d;
    env.navigation.o = *o;
    // every tile resident at once (at most one per CU) and no graph capture (a replay would repeat the barrier number
    // baked into the arguments): the reduction is made inside the launch; otherwise by a second kernel behind it
    bool grid_sync = false, capturing = false;
    if (d->collisions) {
      hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
      if (hipStreamIsCapturing((hipStream_t)stream, &cap) != hipSuccess) cap = hipStreamCaptureStatusNone;
      capturing = cap != hipStreamCaptureStatusNone;
      // (one wave polls the barrier's words: lane w loops over the tile groups of pair word w, 64 words per pass -
      //  grid_bits_collect; d_nav_sync holds 16 groups = 512 tiles of every pair word)
      grid_sync = !capturing && blocks_of(w->batch) <= w->n_cu && blocks_of(w->batch) <= 512;
    }
    // the second-kernel form: the tiles OR into one of two masks (zero by now), the collision kernel reads it, and the NEXT
    // eager launch - which fills the other mask - zeroes it.  A captured launch cannot alternate (a replay repeats its
    // arguments): it has a third mask of its own, which a memset node behind the collision kernel zeroes again.
    const int mw = (w->n_pairs + 31) / 32;
    uint32_t* mask = w->d_nav_mask ? w->d_nav_mask + (size_t)(capturing ? 2 : w->nav_flip) * mw : nullptr;
    uint32_t* mask_other = (w->d_nav_mask && !capturing) ? w->d_nav_mask + (size_t)(w->nav_flip ^ 1) * mw : nullptr;
    env.navigation.w = NavWorld{w->d_angles, w->d_angles_cs, w->d_mpairs, mask, grid_sync ? w->d_nav_sync : nullptr,
                                w->nav_seq, d->collisions ? w->n_pairs : 0, w->d_gave_up, env.ablate,
                                grid_sync ? nullptr : mask_other};
    if (n_steps > 1 && d->collisions && !grid_sync)
      return fail("vmas_world_rollout_env: navigation's collision penalties reduce over the whole batch after every step "
                  "(World.collides): several steps per launch need every tile resident at once (%d tiles, %d CUs) and a stream "
                  "that is not being captured", blocks_of(w->batch), w->n_cu);
    if (step_impl(w, state, agent_ft, ld, args, stream, n_steps, 0, &env, ENV_NAVIGATION, nav_fixed, nav_per_wave, 0, -1,
                  1 << 20))  // (the per-wave columns for EVERY wave: with >= 2 waves per agent they share its block's writing)
      return -1;
    if (grid_sync) {
      w->nav_seq += (uint32_t)n_steps;  // (only a launch that was made has arrived at its barriers)
      if (gated) w->gated_nav_seq = (uint32_t)n_steps;  // (... and a gated one that found the gate shut has not: see above)
    }
    if (d->collisions && !grid_sync) {
      if (vmas::launch_navigation_collisions(d, o, w->batch, state, ld, mask, mw, stream, gated ? err_flags : nullptr)) return -1;
      if (capturing) HIP_TRY(hipMemsetAsync(mask, 0, (size_t)mw * sizeof(uint32_t), (hipStream_t)stream));
      else { w->nav_flip ^= 1; if (gated) w->gated_nav_flip = 1; }
    }
    return 0;
  }
  if (post_kind == VMAS_POST_FOOTBALL) {
    const auto* d = (const VmasFootballDesc*)post_desc;
    const auto* o = (const VmasFootballBuffers*)post_buffers;
    const int n = d->n_blue + d->n_red;
    if (d->n_blue < 1 || d->n_red < 1 || n + 1 > VMAS_ENV_MAX_AGENTS || d->agent0 < 0 || d->agent0 + n + 1 > w->base.nE)
      return fail("vmas_world_step_env: football team sizes / agent0 out of range");
    for (int slot = 0; slot <= n; ++slot) {  // the epilogue reads slot k's rows at agent0 + k and its forces at agent index k
      const VmasEntityDesc& e = w->ents[d->agent0 + slot];
      if (!(e.flags & VMAS_F_AGENT) || e.agent_index != slot || !(e.flags & (VMAS_F_MOVABLE | VMAS_F_ROTATABLE)))
        return fail("vmas_world_step_env: football entity %d is not the dynamic agent of index %d", d->agent0 + slot, slot);
    }
    const int adv = d->observe_adversaries ? 1 : 0, mates = d->observe_teammates ? 1 : 0;
    if (d->n_red * adv + (d->n_blue - 1) * mates != d->n_blue * adv + (d->n_red - 1) * mates)
      return fail("vmas_world_step_env: the football epilogue writes one [n_agents][batch][obs_dim] block (equal observation sizes)");
    if (!o->pos_shaping || !o->obs || !o->rew || !o->terms || !o->touching || !o->done)
      return fail("vmas_world_step_env: null football buffer");
    // Two forms, the same device functions in both (the same bits: tests/test_round4_gpu.py).
    // (0) ONE launch: the post-step as the compacted kernel's epilogue, K steps per launch.  Every K-step rollout, and single
    //     steps up to one tile per CU.  Its limits as ONE step per launch beyond that: the epilogue's registers on top of the
    //     physics' (119 against 80) leave the kernel four waves per SIMD = two tiles per CU where the physics alone has three;
    //     every tile is in the same phase at the same time - all in the physics (instruction issue), then all storing
    //     observations, a row per lane (3.6 TB/s of the 5.85 HBM takes contiguous runs at: scripts/micro/store_pattern.hip).
    //     In a K-step launch the tiles drift apart and one tile's stores run beside another's contacts.
    // (1) TWO launches per step: the step kernel with the ingest prologue (three tiles per CU), then the stand-alone post-step
    //     kernel (a tile's rows as the contiguous runs they are).  Single steps beyond one tile per CU: 131 072 environments
    //     245 -> 172-177 us per Environment.step, 65 536: 144 -> 123, 32 768: 71 -> 69 (profiles/r04j_football_forms_by_batch.jsonl;
    //     16 384: 44 one launch, 48 two).  K-step rollouts stay with (0): 181 us per step at 131 072 against 195, 96 against
    //     117 at 65 536.  Measured and dropped: (1) with the post-steps on a second queue beside the NEXT step's physics
    //     (reading a snapshot of the agents' rows): 214 us per step at 131 072, 100 at 65 536 - the two kernels take the
    //     chip from each other.
    static const int env_knob = knob("VMAS_FOOTBALL_SPLIT") ? atoi(knob("VMAS_FOOTBALL_SPLIT")) : -1;  // (A/B, profile builds)
    const int forced = w->football_form >= 0 ? w->football_form : env_knob;
    const bool fused_possible = compact_on(w) && !(args && args->joint_fixed_rot);
    int form = forced >= 0 ? forced : (n_steps == 1 && blocks_of(w->batch) > kFootballOneLaunchTilesPerCu * w->n_cu ? 1 : 0);
    if (form == 0 && !fused_possible) form = 1;
    if (form != 0) {
      if (!o->agent_ft) return fail("vmas_world_step_env: football as two launches per step needs VmasFootballBuffers.agent_ft");
      if (ingest && ingest->n_scripts > 0 && n_steps > 1 && !compact_on(w))
        return fail("vmas_world_rollout_env: scripted agents read the state in HBM, which a multi-step launch does not refresh");
      for (int stp = 0; stp < n_steps; ++stp) {
        if (ingest) {
          DevEnv e1 = env;  // this step's rows of the [n_steps * batch, action_size] action tensors
          for (int a = 0; a < VMAS_ENV_MAX_AGENTS; ++a) {
            VmasActionSlot& S = e1.ingest.agents[a];
            if (S.action) S.action += (size_t)stp * w->batch * S.action_size;
            if (S.action_index) S.action_index += (size_t)stp * w->batch;
          }
          if (step_impl(w, state, agent_ft, ld, args, stream, 1, 0, &e1, ENV_INGEST, 0, 0)) return -1;
        } else if (step_impl(w, state, agent_ft, ld, args, stream, 1, 0, nullptr, ENV_NONE, 0, 0)) {
          return -1;
        }
        if (vmas::launch_football_post(d, o, w->batch, state, ld, stp, stream, gated ? err_flags : nullptr)) return -1;
      }
      return 0;
    }
    env.football.d = *d;
    env.football.o = *o;
    return step_impl(w, state, agent_ft, ld, args, stream, n_steps, 0, &env, ENV_FOOTBALL, 0, 0);
  }
  if (post_kind == VMAS_POST_NONE) return step_impl(w, state, agent_ft, ld, args, stream, n_steps, 0, &env, ENV_INGEST, 0, 0);
  return fail("vmas_world_step_env: post_kind %d has no fused epilogue", post_kind);
}

static int step_impl(VmasWorld* w, float* state, float* agent_ft, int64_t ld, const VmasStepArgs* args, void* stream,
                     int n_steps, int64_t ft_stride, DevEnv* env, int env_kind, size_t scratch_fixed,
                     size_t scratch_per_wave, int env_first, int env_count, int scratch_wave_cap) {
  if (!w || !state) return fail("vmas_world_step: null argument");
  if (w->host_only) return fail("vmas_world_step: a planning world (device -1) cannot be stepped");
  if (w->base.nA > 0 && !agent_ft) return fail("vmas_world_step: world has agents but agent_ft is null");
  if (ld < w->batch) return fail("vmas_world_step: ld (%lld) < batch (%d)", (long long)ld, w->batch);
  if (w->h_gave_up && __atomic_load_n(w->h_gave_up, __ATOMIC_RELAXED) != 0u) {
    // an EARLIER launch on this world flagged it (no synchronisation here: the word is host memory the device wrote)
    __atomic_store_n(w->h_gave_up, 0u, __ATOMIC_RELAXED);
    return fail("vmas_world_step: a grid-wide barrier of an earlier step of this world gave up waiting - the grid was not "
                "co-resident (other work held compute units) and the steps since then used whatever broad-phase bits had "
                "arrived: their results are not the reference's.  Reset the world's state; run exact_broad_phase / the "
                "navigation epilogue on an otherwise idle device, or switch exact_broad_phase off");
  }
  DevStepArgs a{};
  a.n_steps = n_steps;
  a.ft_stride = ft_stride;
  if (args) {
    a.pair_mask = args->pair_mask; a.joint_fixed_rot = args->joint_fixed_rot; a.entity_gravity = args->entity_gravity;
    a.first_substep = args->first_substep; a.n_substeps = args->n_substeps;
    if (a.first_substep < 0 || a.first_substep >= w->base.substeps)
      return fail("vmas_world_step: first_substep %d outside [0,%d)", a.first_substep, w->base.substeps);
  }
  {
    static const int ablate = knob("VMAS_ABLATE") ? atoi(knob("VMAS_ABLATE")) : 0;
    static const int trace = knob("VMAS_TRACE") ? atoi(knob("VMAS_TRACE")) : 0;
    a.ablate = ablate;
    if (trace) {
      const size_t n = (size_t)((w->batch + TILE - 1) / TILE) * 16 * 16;
      if (!w->d_trace) {
        HIP_TRY(hipMalloc((void**)&w->d_trace, n * 8));
        HIP_TRY(hipMemset(w->d_trace, 0, n * 8));
      }
      a.trace = w->d_trace;
      if (trace == 2 && env != nullptr) { env->trace = w->d_trace; a.trace = nullptr; }  // (the specialised kernel's stamps)
    }
  }
  {
    int cur = -1;  // the stream belongs to w->device: make it current if the caller's thread is elsewhere
    if (hipGetDevice(&cur) != hipSuccess || cur != w->device) HIP_TRY(hipSetDevice(w->device));
  }
  Sched* S;
  if (get_sched(w, w->lanes, &S)) return -1;
  hipStream_t s = (hipStream_t)stream;
  // (a whole-batch call on the caller's stream; the sub-range calls of a multi-queue vmas_world_step_n tick there, before the fork)
  if ((env_kind == ENV_NONE || env_kind == ENV_INGEST) && env_count < 0 && compact_adapt_tick(w, s, n_steps)) return -1;
  uint32_t seq_advance = 0;
  int lazy_passes = 0;
  const int form = (args && args->exact_broad_phase) ? exact_form(w, stream_capturing(s)) : EXACT_NONE;
  if (form != EXACT_NONE) {
    // The reference's broad phase: a pair is processed - for ALL environments - iff SOME environment of the batch has the
    // pair's bounding circles overlapping (core.py:2797-2801), re-evaluated at every substep.
    if (args->pair_mask) return fail("vmas_world_step: exact_broad_phase and a recorded pair_mask are mutually exclusive");
    if (env_count >= 0) return fail("vmas_world_step: exact_broad_phase takes the whole batch");
    const int mask_words = (w->n_pairs + 31) / 32;
    const int run = a.n_substeps > 0 ? a.n_substeps : w->base.substeps - a.first_substep;
    // (a launch captured into a HIP graph would replay the barrier sequence number / the slot set baked into its arguments:
    // under capture the launch-per-substep form is used, which keeps no state on the host)
    if (form == EXACT_LAZY) {  // optimistic passes + the batch's words on demand (vmas_env_device.h)
      lazy_passes = run * (n_steps > 1 ? n_steps : 1);
      if (lazy_prepare(w, lazy_passes, &a.lz)) return -1;
    } else if (form == EXACT_BARRIER) {  // every tile resident at once: mask + grid barrier inside the step kernel
      a.sync = w->d_sync; a.mpairs = w->d_mpairs; a.n_mpairs = w->n_pairs; a.mask_words = mask_words;
      a.seq0 = w->sync_seq;
      a.gave_up = w->d_gave_up;
      seq_advance = (uint32_t)(run * (n_steps > 1 ? n_steps : 1));  // (added once the launch has been made)
    } else {  // more tiles than the chip holds: a mask launch + a one-substep launch per substep
      if (env_kind != ENV_NONE || n_steps > 1)
        return fail("vmas_world_step: exact_broad_phase on %d tiles (> %d CUs) runs one launch per substep: no fused "
                    "epilogue, no rollout", blocks_of(w->batch), w->n_cu);
      for (int sub = a.first_substep; sub < a.first_substep + run; ++sub) {
        HIP_TRY(hipMemsetAsync(w->d_exact_mask, 0, (size_t)mask_words * sizeof(uint32_t), s));
        if (vmas_world_pair_mask(w, state, ld, w->d_exact_mask, stream)) return -1;
        DevStepArgs b = a;
        b.pair_mask = w->d_exact_mask; b.first_substep = sub; b.n_substeps = 1;
        if (launch_physics(w, S, state, agent_ft, ld, b, s)) return -1;
      }
      return 0;
    }
  }
  // (the barrier sequence number advances only for a launch that was made: a failed call leaves the arrival counter and
  //  the next launch's target in step)
  auto launch = [&]() -> int {
    if (env_count >= 0) {  // a sub-range of the batch (vmas_world_step_n over two queues): plain physics, no optional inputs
      if (env_kind != ENV_NONE || args) return fail("vmas_world_step: environment sub-ranges take no optional inputs");
      return launch_physics(w, S, state + env_first, agent_ft ? agent_ft + env_first : nullptr, ld, a, s, env_count,
                            (long)ld - env_first);
    }
    if (env_kind == ENV_NONE) return launch_physics(w, S, state, agent_ft, ld, a, s);
    const bool cp = compact_on(w) && !a.joint_fixed_rot && !(ABLATE(a) & 0xff) && !ABLATE(*env) && compact_takes(a);
    if (env_kind == ENV_INGEST && cp) return launch_compact(w, ENV_INGEST, state, agent_ft, ld, a, env, 0, s, w->batch, ld);
    if (env_kind == ENV_FOOTBALL) {
      if (!cp) return fail("vmas_world_step_env: the football epilogue runs behind the compacted step kernel, which this world / "
                           "this launch does not use (vmas_world_set_compact, per-environment joint inputs)");
      // observation staging (football_post_tile): [64][17] floats per wave, the first waves' in the per-substep scratch
      // that is dead by then (ballots, base, keys, contact forces, torques: vmas_compact.h).
      const size_t slab = 64 * (kFootballStageChunk + 1);
      const compact::DevCompact& dc = w->cp.dc;
      const int in_dead = (int)((2 * (size_t)dc.n_pairs + (size_t)((dc.n_pairs + 1) & ~1) + (size_t)compact::CAP * (dc.has_torque ? 4 : 3)) / slab);
      const size_t stage = w->cp.nw > in_dead ? (size_t)(w->cp.nw - in_dead) * slab * sizeof(float) : 0;
      // Only in the latency regime (16 waves per tile: one tile per CU whatever its LDS).  With 8 waves per tile the
      // 79 KB leave the CU two tiles on paper, but measured (131 072 environments, rollout): 657 us per step against 180
      // without; half tiles ([32][17], a chunk in two passes, 61 KB per tile): 191 - 16 384 environments: 23.6 against 28.9.
      static const bool no_stage = knob("VMAS_FOOTBALL_NO_STAGE") != nullptr && knob("VMAS_FOOTBALL_NO_STAGE")[0] == '1';  // (A/B)
      const bool staged = !no_stage && w->cp.nw >= 16 && w->cp.lds_bytes + stage <= 160 * 1024;
      env->scratch_off = staged ? (int32_t)(w->cp.lds_bytes / sizeof(float)) : -1;
      return launch_compact(w, ENV_FOOTBALL, state, agent_ft, ld, a, env, staged ? stage : 0, s, w->batch, ld);
    }
    if (S->nw < 2 && env_kind != ENV_INGEST && env_kind != ENV_NAVIGATION)
      return fail("vmas_world_step_env: the fused epilogue needs at least 2 waves per tile");
    size_t extra = 0;
    if (env_extra_lds(w, &S, scratch_fixed, scratch_per_wave, scratch_wave_cap, &extra)) return -1;
    env->scratch_off = (int32_t)(S->lds_bytes / sizeof(float));
    if (env_kind == ENV_BALANCE) return launch_any_level<ENV_BALANCE>(w, S, state, agent_ft, ld, a, *env, extra, s);
    if (env_kind == ENV_INGEST) return launch_any_level<ENV_INGEST>(w, S, state, agent_ft, ld, a, *env, 0, s);
    if (env_kind == ENV_NAVIGATION) {
      if (env->navigation.d.n_agents > kNavMaxOwn * S->nw)
        return fail("vmas_world_step_env: %d agents on %d waves per tile (at most %d agents per wave)", env->navigation.d.n_agents,
                    S->nw, kNavMaxOwn);
      return launch_any_level<ENV_NAVIGATION>(w, S, state, agent_ft, ld, a, *env, extra, s);
    }
    return launch_any_level<ENV_TRANSPORT>(w, S, state, agent_ft, ld, a, *env, extra, s);
  };
  const int rc = launch();
  if (rc == 0) w->sync_seq += seq_advance;
  return rc;
}

// profiling aid, not part of the ABI: copy out the s_memtime stamps of the last launch
int vmas_debug_trace(VmasWorld* w, unsigned long long* host, int64_t n_words) {
  if (!w || !w->d_trace) return fail("vmas_debug_trace: tracing is off (set VMAS_TRACE=1)");
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(host, w->d_trace, (size_t)n_words * 8, hipMemcpyDeviceToHost));
  return 0;
}

int vmas_debug_schedule(VmasWorld* w, uint32_t* words, int64_t capacity, int32_t* meta) {
  if (!w || !meta) return fail("vmas_debug_schedule: null argument");
  Sched* S;
  if (get_sched(w, w->lanes, &S)) return -1;
  const DevWorld& D = S->dw;
  const int32_t m[24] = {S->nw, w->share_mode, w->level, (int32_t)S->h_blob.size(), D.b_ent, D.b_segs, D.b_owned, D.b_refs,
                         D.b_items, D.n_segs, D.n_owned, S->rows, D.fired_recs, D.items_in_lds, D.nE, D.nA, D.off_af, D.row_tr,
                         (int32_t)(D.trig_mask & 0xffffffffu), (int32_t)(D.trig_mask >> 32), (int32_t)(D.box_mask & 0xffffffffu),
                         (int32_t)(D.box_mask >> 32), D.trig_in_args, S->spec_id};
  memcpy(meta, m, sizeof(m));
  if (words) {
    if (capacity < (int64_t)S->h_blob.size()) return fail("vmas_debug_schedule: %zu words do not fit", S->h_blob.size());
    memcpy(words, S->h_blob.data(), S->h_blob.size() * sizeof(uint32_t));
  }
  return 0;
}

int vmas_debug_compact_stats(VmasWorld* w, int64_t out[4]) {
  if (!w || !out) return fail("vmas_debug_compact_stats: null argument");
  if (!w->adapt.d_count) return fail("vmas_debug_compact_stats: no device side");
  HIP_TRY(hipSetDevice(w->device));
  HIP_TRY(hipDeviceSynchronize());
  unsigned long long c = 0;
  HIP_TRY(hipMemcpy(&c, w->adapt.d_count, sizeof(c), hipMemcpyDeviceToHost));
  out[0] = (int64_t)c; out[1] = (int64_t)w->adapt.tiles; out[2] = (int64_t)w->adapt.switches; out[3] = (int64_t)w->adapt.backoff;
  return 0;
}

int vmas_debug_compact_plan(VmasWorld* w, uint32_t* words, int64_t capacity, int64_t* meta /* [16] */) {
  if (!w || !meta) return fail("vmas_debug_compact_plan: null argument");
  const VmasWorld::CompactPlan& C = w->cp;
  if (!C.ok) return fail("vmas_debug_compact_plan: this world has no plan for the lane-compacted kernel");
  const compact::DevCompact& D = C.dc;
  const int64_t m[16] = {C.nw, C.own, (int64_t)C.lds_bytes, (int64_t)C.h_blob.size(), D.n_pairs, D.n_owned, D.t_entoff, D.t_waves,
                         D.t_units, (int64_t)D.dyn_mask, (int64_t)D.static_mask, (int64_t)D.line_mask, D.off_af, D.off_tr,
                         D.has_torque, w->base.nE};
  memcpy(meta, m, sizeof(m));
  if (words) {
    if (capacity < (int64_t)C.h_blob.size()) return fail("vmas_debug_compact_plan: %zu words, capacity %lld", C.h_blob.size(), (long long)capacity);
    memcpy(words, C.h_blob.data(), C.h_blob.size() * sizeof(uint32_t));
  }
  return 0;
}

int vmas_debug_football_form(VmasWorld* w, int32_t form) {
  if (!w || form < -1 || form > 1) return fail("vmas_debug_football_form: form is -1 (the library's choice), 0 (one launch) or 1 (two per step)");
  w->football_form = form;
  return 0;
}

int vmas_debug_lazy_stats(VmasWorld* w, int64_t out[4]) {  // launches made, tiles that asked the batch, ... that found a pair off, repeated polls
  if (!w || !out || !w->d_sync) return fail("vmas_debug_lazy_stats: no device side");
  HIP_TRY(hipSetDevice(w->device));
  HIP_TRY(hipDeviceSynchronize());
  uint32_t v[4] = {0, 0, 0, 0};
  HIP_TRY(hipMemcpy(v, w->d_sync, sizeof(v), hipMemcpyDeviceToHost));
  out[0] = (int64_t)w->lazy.tag; out[1] = v[2]; out[2] = v[3]; out[3] = 0;
  uint32_t polls = 0;
  HIP_TRY(hipMemcpy(&polls, w->d_sync + 4, sizeof(polls), hipMemcpyDeviceToHost));
  out[3] = polls;
  return 0;
}

int vmas_debug_force_gave_up(VmasWorld* w) {
  if (!w || !w->h_gave_up) return fail("vmas_debug_force_gave_up: no device side");
  __atomic_store_n(w->h_gave_up, 1u, __ATOMIC_RELAXED);
  return 0;
}

int vmas_debug_math(int32_t op, const float* a, const float* b, float* out, int32_t n, void* stream) {
  if (!a || !out || n < 0 || op < 0 || op > VMAS_MATH_SIN) return fail("vmas_debug_math: bad argument");
  if ((op == VMAS_MATH_DIV || op == VMAS_MATH_NORM) && !b) return fail("vmas_debug_math: op %d needs two operands", op);
  if (n == 0) return 0;
  hipLaunchKernelGGL(math_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, op, a, b, out, n);
  HIP_TRY(hipGetLastError());
  return 0;
}

int vmas_world_step_n(VmasWorld* w, float* state, float* agent_ft, int64_t ld, int64_t ft_step_stride, int32_t n_steps,
                      const VmasStepArgs* args, void* stream) {
  if (n_steps < 0) return fail("vmas_world_step_n: n_steps %d < 0", n_steps);
  if (!w) return fail("vmas_world_step_n: null world");
  // Several queues: environments are independent, so the batch is cut at tile boundaries and the parts are stepped by
  // independent launch sequences - one on the caller's stream, the others on side streams forked from it and joined
  // back at the end.  A dependent launch costs ~2.9 us of front-end time during which the chip idles; with two
  // sequences the gap of one part is filled by the kernel of the other, and the load / compute / store phases of the
  // parts' tiles fall out of step and overlap.  Same kernels, same results bit for bit.
  // Library's choice (queues_for): when each half still has a tile for every CU and a launch is long against the host's
  // 2.9 us per enqueue (measured, profiles/r02_queues_sweep.jsonl: interpreter balance 32768 envs -15 %, 131072 -12 %,
  // 1 M -4 %, football 131072 -8.5 %, navigation 65536 -15 %; transport 16384 = one tile per CU in total: +11 %, not
  // split; the specialised balance kernel at 32768 envs is faster on one queue) and the call has >= 8 steps.
  // (a call whose only argument is the exact broad phase on a world that needs none of it - no band-capable pair: navigation -
  //  is a call without arguments: the rule holds per environment there, the parts of the batch need nothing from each other)
  const bool no_args = !args || (!args->pair_mask && !args->joint_fixed_rot && !args->entity_gravity && args->first_substep == 0 &&
                                 args->n_substeps <= 0 && (!args->exact_broad_phase || exact_form(w, false) == EXACT_NONE));
  const int nq = no_args ? queues_for(w, n_steps) : 1;
  if (nq > 1) {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != w->device) HIP_TRY(hipSetDevice(w->device));
    if (ensure_queues(w, nq)) return -1;
    hipStream_t s = (hipStream_t)stream;
    const int tiles = blocks_of(w->batch);
    // queue q steps the tiles [q * tiles / nq, (q + 1) * tiles / nq): queue 0 = the caller's stream, the others are side
    // streams forked from it here and joined back at the end
    auto first_env = [&](int q) { return (int)((long)tiles * q / nq) * TILE; };
    // (in chunks: where the library chooses between two kernels by what the tiles do - VmasWorld::CompactAdapt - it looks
    //  at the joined stream between chunks; a fork / join pair per 64 steps is noise)
    const int chunk = (w->compact_mode == -1 && compact_on(w)) ? 64 : n_steps;
    int rc = 0;
    for (int i0 = 0; i0 < n_steps && !rc; i0 += chunk) {
      const int i1 = std::min(n_steps, i0 + chunk);
      if (compact_adapt_tick(w, s, i1 - i0)) return -1;
      HIP_TRY(hipEventRecord(w->ev_fork, s));
      for (int q = 0; q < nq - 1; ++q) HIP_TRY(hipStreamWaitEvent(w->side[q], w->ev_fork, 0));
      for (int i = i0; i < i1 && !rc; ++i) {
        float* ft = agent_ft ? agent_ft + (int64_t)i * ft_step_stride : nullptr;
        // one kernel for the whole step, whichever queue a part of it runs on (the two football kernels are not bitwise
        // equal in dense contact: halves on different kernels would make the bits depend on the queue split)
        if (w->compact_mode == -1 && w->adapt.d_count != nullptr && compact_on(w)) w->adapt.forced = compact_pick_step(w);
        for (int q = 0; q < nq && !rc; ++q) {
          const int lo = first_env(q), hi = q + 1 == nq ? w->batch : first_env(q + 1);
          rc = step_impl(w, state, ft, ld, nullptr, q == 0 ? (void*)s : (void*)w->side[q - 1], 1, 0, nullptr, ENV_NONE, 0, 0,
                         lo, hi - lo);
        }
        w->adapt.forced = -1;
      }
      for (int q = 0; q < nq - 1; ++q) {  // (joined even after a failed launch: the caller's stream stays ordered)
        HIP_TRY(hipEventRecord(w->ev_join[q], w->side[q]));
        HIP_TRY(hipStreamWaitEvent(s, w->ev_join[q], 0));
      }
    }
    return rc;
  }
  for (int i = 0; i < n_steps; ++i) {
    int rc = vmas_world_step(w, state, agent_ft ? agent_ft + (int64_t)i * ft_step_stride : nullptr, ld, args, stream);
    if (rc) return rc;
  }
  return 0;
}

int vmas_world_pair_mask(VmasWorld* w, const float* state, int64_t ld, uint32_t* mask, void* stream) {
  if (w && w->host_only) return fail("vmas_world_pair_mask: a planning world (device -1) has no device side");
  if (!w || !state || !mask) return fail("vmas_world_pair_mask: null argument");
  hipStream_t s = (hipStream_t)stream;
  const int words = (w->n_pairs + 31) / 32;
  HIP_TRY(hipMemsetAsync(mask, 0, sizeof(uint32_t) * (words ? words : 1), s));
  if (w->n_pairs == 0) return 0;
  const int T = w->base.nE <= 64 ? 256 : 64;  // threads per block: the block's positions must fit in LDS
  const size_t lds = (size_t)w->base.nE * 2 * T * sizeof(float);
  if (lds > 160 * 1024) return fail("vmas_world_pair_mask: %d entities do not fit in LDS", w->base.nE);
  if (lds > 64 * 1024) {
    static std::atomic<size_t> set_for_dev[64];
    std::atomic<size_t>& set_for = set_for_dev[w->device & 63];
    if (set_for.load() < lds) {
      HIP_TRY(hipFuncSetAttribute((const void*)pair_mask_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      set_for = lds;
    }
  }
  hipLaunchKernelGGL(pair_mask_kernel, dim3((w->batch + T - 1) / T), dim3(T), lds, s, w->d_mpairs, w->n_pairs,
                     w->base.nE, state, (long)ld, w->batch, mask);
  HIP_TRY(hipGetLastError());
  return 0;
}

int vmas_world_set_lidars(VmasWorld* w, const VmasLidarDesc* lidars, int32_t n) {
  if (w && w->host_only) return fail("vmas_world_set_lidars: a planning world (device -1) has no device side");
  if (!w || (n > 0 && !lidars)) return fail("vmas_world_set_lidars: null argument");
  HIP_TRY(hipSetDevice(w->device));
  (void)hipFree(w->d_lidars); (void)hipFree(w->d_targets); (void)hipFree(w->d_angles); (void)hipFree(w->d_angles_cs);
  w->d_lidars = nullptr; w->d_targets = nullptr; w->d_angles = nullptr; w->d_angles_cs = nullptr;
  w->n_lidars = 0; w->max_rays = 0;
  w->h_lidars.clear();
  w->h_targets.clear();
  if (n <= 0) return 0;
  std::vector<DevLidar> dl(n);
  std::vector<DevTarget> dt;
  std::vector<float> da;
  for (int i = 0; i < n; ++i) {
    const VmasLidarDesc& L = lidars[i];
    if (L.entity < 0 || L.entity >= w->base.nE || L.n_rays <= 0) return fail("vmas_world_set_lidars: bad sensor %d", i);
    dl[i] = {L.entity, L.n_rays, L.n_targets, (int)dt.size(), (int)da.size(), L.max_range,
             (float)((double)L.max_range / 2.0)};
    for (int t = 0; t < L.n_targets; ++t) {
      int e = L.targets[t];
      if (e < 0 || e >= w->base.nE) return fail("vmas_world_set_lidars: sensor %d target %d out of range", i, e);
      const VmasEntityDesc& E = w->ents[e];
      dt.push_back({e, E.shape, E.length, E.width, E.radius});
    }
    for (int r = 0; r < L.n_rays; ++r) da.push_back(L.angles[r]);
    if (L.n_rays > w->max_rays) w->max_rays = L.n_rays;
  }
  HIP_TRY(upload(&w->d_lidars, dl));
  HIP_TRY(upload(&w->d_targets, dt));
  HIP_TRY(upload(&w->d_angles, da));
  HIP_TRY(hipMalloc((void**)&w->d_angles_cs, da.size() * sizeof(float2)));
  hipLaunchKernelGGL(lidar_table_kernel, dim3(((int)da.size() + 255) / 256), dim3(256), 0, (hipStream_t)0, w->d_angles,
                     (int)da.size(), w->d_angles_cs);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize((hipStream_t)0));
  w->h_lidars = dl;
  w->h_targets = dt;
  w->n_lidars = n;
  // the lane-compacted cast (lidar_compact_kernel) serves sets whose targets are all spheres and whose tables fit in LDS
  {
    VmasWorld::LidarCompact& LC = w->lc;
    (void)hipFree(LC.d_ent_slot); (void)hipFree(LC.d_slot_ent);
    LC.d_ent_slot = nullptr;
    LC.d_slot_ent = nullptr;
    LC.ok = false;
    bool spheres = n <= 1023 && w->max_rays <= 64;
    std::vector<int> ent_slot(w->base.nE, -1), slot_ent;
    auto slot = [&](int e) { if (ent_slot[e] < 0) { ent_slot[e] = (int)slot_ent.size(); slot_ent.push_back(e); } };
    size_t pairs = 0;
    for (int i = 0; i < n && spheres; ++i) {
      if (dl[i].n_targets > 65535) spheres = false;
      slot(dl[i].entity);
      pairs += (size_t)dl[i].n_targets;
      for (int t = 0; t < dl[i].n_targets; ++t) {
        const DevTarget& T = dt[dl[i].target_off + t];
        if (T.shape != VMAS_SHAPE_SPHERE) spheres = false;
        slot(T.entity);
      }
    }
    const size_t words = slot_ent.size() * 128 + (size_t)n * 64 + (size_t)n * w->max_rays * kLidarStride + 2 +
                         (size_t)n * kLidarSensorWords + pairs * kLidarPairWords + pairs * 64;
    if (spheres && n <= 65535 && slot_ent.size() <= 65535 && pairs > 0 && pairs < (1u << 26) && words * sizeof(float) <= 160 * 1024) {
      auto fb = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u; };
      std::vector<uint32_t> tab;
      uint32_t pair0 = 0;
      for (int i = 0; i < n; ++i) {
        tab.push_back((uint32_t)ent_slot[dl[i].entity]); tab.push_back((uint32_t)dl[i].n_rays); tab.push_back((uint32_t)dl[i].angle_off);
        tab.push_back(pair0); tab.push_back(fb(dl[i].max_range)); tab.push_back(fb(dl[i].half_range));
        pair0 += (uint32_t)dl[i].n_targets;
      }
      for (int i = 0; i < n; ++i)
        for (int t = 0; t < dl[i].n_targets; ++t) {
          const DevTarget& T = dt[dl[i].target_off + t];
          tab.push_back((uint32_t)i | (uint32_t)ent_slot[T.entity] << 16);
          tab.push_back(fb(T.radius));
        }
      HIP_TRY(upload(&LC.d_ent_slot, tab));
      HIP_TRY(upload(&LC.d_slot_ent, slot_ent));
      LC.n_slots = (int)slot_ent.size();
      LC.n_pairs_total = (int)pairs;
      LC.lds = words * sizeof(float);
      LC.ok = true;
    }
  }
  return 0;
}

// the lane-compacted cast is taken: when pinned (mode 1), or - the library's choice - once the batch has more tiles than half
// the CUs AND the sensor set's LDS leaves a CU at least two resident tiles (measured on navigation's 8 x 12 rays x 7 targets,
// 46 KB; larger sets keep the plain kernel until they have been measured)
static bool lidar_compact_on(const VmasWorld* w) {
  return w->lc.ok && (w->lc.mode == 1 || (w->lc.mode == -1 && 2 * blocks_of(w->batch) > w->n_cu && w->lc.lds <= 64 * 1024));
}

int vmas_world_set_lidar_compact(VmasWorld* w, int32_t mode) {
  if (!w) return fail("vmas_world_set_lidar_compact: null world");
  if (mode < -1 || mode > 1) return fail("vmas_world_set_lidar_compact: mode %d (-1 library's choice, 0 never, 1 whenever the sensor set qualifies)", mode);
  w->lc.mode = mode;
  return 0;
}

int vmas_world_get_lidar_compact(VmasWorld* w) {
  return (w && lidar_compact_on(w)) ? 1 : 0;
}

int vmas_world_set_queries(VmasWorld* w, const VmasQuery* queries, int32_t n) {
  if (w && w->host_only) return fail("vmas_world_set_queries: a planning world (device -1) has no device side");
  if (!w || (n > 0 && !queries)) return fail("vmas_world_set_queries: null argument");
  HIP_TRY(hipSetDevice(w->device));
  (void)hipFree(w->d_queries);
  w->d_queries = nullptr;
  w->n_queries = 0;
  if (n <= 0) return 0;
  std::vector<DevQuery> dq(n);
  for (int i = 0; i < n; ++i) {
    int a = queries[i].a, b = queries[i].b;
    if (a < 0 || a >= w->base.nE || b < 0 || b >= w->base.nE || queries[i].kind < 0 || queries[i].kind > 1)
      return fail("vmas_world_set_queries: query %d is malformed", i);
    // role order of the reference: (box, sphere), (line, sphere), (box, line)
    auto rank = [&](int e) { return w->ents[e].shape == VMAS_SHAPE_BOX ? 0 : (w->ents[e].shape == VMAS_SHAPE_LINE ? 1 : 2); };
    if (rank(a) > rank(b)) std::swap(a, b);
    const VmasEntityDesc &A = w->ents[a], &B = w->ents[b];
    dq[i] = {queries[i].kind, a, b, A.shape, B.shape, A.length, A.width, A.radius, B.length, B.width, B.radius};
  }
  HIP_TRY(upload(&w->d_queries, dq));
  w->n_queries = n;
  return 0;
}

int vmas_world_run_queries(VmasWorld* w, const float* state, int64_t ld, float* out, void* stream) {
  if (w && w->host_only) return fail("vmas_world_run_queries: a planning world (device -1) has no device side");
  if (!w || !state || !out) return fail("vmas_world_run_queries: null argument");
  if (w->n_queries <= 0) return fail("vmas_world_run_queries: no queries registered (vmas_world_set_queries)");
  hipLaunchKernelGGL(query_kernel, dim3((w->batch + 255) / 256, w->n_queries), dim3(256), 0, (hipStream_t)stream,
                     w->d_queries, state, (long)ld, w->batch, out);
  HIP_TRY(hipGetLastError());
  return 0;
}

int vmas_world_cast_rays(VmasWorld* w, const float* state, int64_t ld, float* out, void* stream) {
  if (w && w->host_only) return fail("vmas_world_cast_rays: a planning world (device -1) has no device side");
  if (!w || !state || !out) return fail("vmas_world_cast_rays: null argument");
  if (w->n_lidars <= 0) return fail("vmas_world_cast_rays: no sensors registered (vmas_world_set_lidars)");
  // sphere-only sensor sets: the lane-compacted cast - the library's choice once the batch has more tiles than half the CUs
  // (at 8 192 environments = 128 tiles the plain kernel's shorter chain still wins: 9.1 us against 10.0)
  if (lidar_compact_on(w)) {
    const VmasWorld::LidarCompact& LC = w->lc;
    if (LC.lds > 64 * 1024) {
      static std::atomic<size_t> set_for_dev[64];
      std::atomic<size_t>& set_for = set_for_dev[w->device & 63];
      if (set_for.load() < LC.lds) {
        HIP_TRY(hipFuncSetAttribute((const void*)lidar_compact_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LC.lds));
        set_for = LC.lds;
      }
    }
    // 16 waves per tile while the chip holds every tile at once or nearly (16 384 environments 10.5 us, plain kernel 14.4), 8
    // beyond (65 536: 23.7 us with 8 waves, 26.3 with 16; plain kernel 37.2)
    const int threads = blocks_of(w->batch) <= 2 * w->n_cu ? 1024 : 512;
    hipLaunchKernelGGL(lidar_compact_kernel, dim3((w->batch + 63) / 64), dim3(threads), LC.lds, (hipStream_t)stream, LC.d_ent_slot,
                       w->d_angles, w->d_angles_cs, LC.d_slot_ent, LC.n_slots, w->n_lidars, LC.n_pairs_total, w->max_rays, state,
                       (long)ld, w->batch, out);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  // rays per thread: few when the batch alone cannot fill the chip (latency-bound), more when it
  // can (each thread then reads its targets once for several rays); VMAS_LIDAR_RPT overrides
  static const int force_rpt = knob("VMAS_LIDAR_RPT") ? atoi(knob("VMAS_LIDAR_RPT")) : 0;
  const long threads = (long)w->batch * w->n_lidars;
  int rpt = force_rpt ? force_rpt : (threads >= (1L << 19) ? 4 : 2);  // measured: navigation 8x12 rays, B = 8192 / 65536
  const dim3 block(256);
  auto grid = [&](int r) { return dim3((w->batch + 255) / 256, w->n_lidars, (w->max_rays + r - 1) / r); };
#define LAUNCH_LIDAR(R)                                                                                            \
  hipLaunchKernelGGL(lidar_kernel<R>, grid(R), block, 0, (hipStream_t)stream, w->d_lidars, w->d_targets, w->d_angles, w->d_angles_cs, \
                     w->max_rays, state, (long)ld, w->batch, out)
  switch (rpt) {
    case 1: LAUNCH_LIDAR(1); break;
    case 2: LAUNCH_LIDAR(2); break;
    case 4: LAUNCH_LIDAR(4); break;
    default: LAUNCH_LIDAR(8); break;
  }
#undef LAUNCH_LIDAR
  HIP_TRY(hipGetLastError());
  return 0;
}

}  // extern "C"
