// vmas_compact.h - the LANE-COMPACTED form of the step kernel for dense sphere worlds (included by vmas_hip.hip).
//
// Why.  step_kernel evaluates a pair's narrow phase for all 64 environments of a tile whenever ONE of them needs it.  In
// `football` (55 sphere-sphere + 110 line-sphere pairs, two substeps) about 50 of a tile's 165 x 64 = 10 560
// (environment, pair) slots are in contact per substep, yet a wave can skip a pair only 4 % of the time: the narrow phase
// ran at ~0.5 % lane utility (profiles/r02e_football131072_physics_pmc_summary.txt: 0.147 of the HBM roof, 0.094 of the
// fp32 roof, VALU busy 21 %).  Here the work is split by what it costs:
//
//   A  broad phase   lane = environment, wave-uniform pair: every lane does useful work (one cheap distance test per
//                    pair: squared centre distance for sphere-sphere, the sphere's gaps in the line's frame for
//                    line-sphere).  The wave's ballot of "this environment needs the narrow phase" is the pair's
//                    64-bit contact mask; the lanes that need it take consecutive slots of the tile's CONTACT LIST
//                    (one LDS counter bump per pair with contacts) and write their key (pair, environment) there.
//   B  narrow phase  lane = CONTACT: slot k of the list - whatever pair, whatever environment - is one lane's job.
//                    ~50 contacts = one pass of one wave instead of ~160 wave-passes of 64 lanes.  Same device
//                    functions (closest_point_line, contact_force) on the same operands: the same bits.
//   C  integrate     lane = environment, wave = owner of an entity: the prologue force (core.py:1995-2004), then the
//                    entity's pairs THAT HAVE CONTACTS in the reference's accumulation order (core.py:2176-2199) -
//                    a per-entity bit mask set in phase A names them, a lane adds contact `base[pair] + rank of its
//                    bit in the pair's mask` if its bit is set - then _integrate_state (core.py:2862-2908).
//
// Exactness.  A pair the broad phase rejects for an environment contributes, in the reference, exactly +0 to a and -0
// to b (core.py:2836-2839: the force is zeroed beyond dist_min; the tests are conservative supersets of that, NaN/inf
// operands always pass).  F + (-0) == F for every F; F + (+0) == F unless F == -0.  So the skipped terms are replaced by
// ONE `+ (+0)` after the sum for lanes that skipped at least one a-side term: once the running sum is +0 only -0 terms
// could follow without changing it, and they do not change it either - the same bits as adding the zeros one by one
// (the argument of step_kernel's "fired" bits, at lane granularity).  Torques likewise.
//
// The contact list has CAP slots per tile.  A tile that needs more (piled-up bodies; non-finite poses, which pass every
// test) is processed in rounds: runs of consecutive pairs with at most CAP contacts between them (a prefix sum over the
// masks stored in phase A; slots re-assigned from the masks, no test is repeated), in pair order, the owners' sums
// carried in registers: same order, same bits.
//
// Scope: worlds whose pairs are all sphere-sphere or line-sphere, no joints, at most 64 entities and OWN_MAX dynamic
// entities per wave; options: recorded / in-kernel exact pair masks, per-environment entity gravity, partial substep
// ranges, multi-step rollouts, the action-ingest prologue and the football epilogue.  Everything else runs step_kernel;
// the two are interchangeable bit for bit (tests/test_compact_gpu.py).
#pragma once

namespace compact {

#ifndef VMAS_COMPACT_CAP
#define VMAS_COMPACT_CAP 256
#endif
constexpr int CAP = VMAS_COMPACT_CAP;     // contact slots per tile and round
constexpr int OWN_MAX = 4;                // dynamic entities one wave can own
constexpr int LIST_MAX = 64;              // pairs one entity can be in (its list lives in one register, one entry per lane)
constexpr int HW_MAX = LIST_MAX / 32;     // 32-bit words of an entity's "pairs with contacts" mask
constexpr int UNIT_PARTNERS = 6;          // partner spheres per broad-phase unit (register arrays of this size)
constexpr int WAVE_UNITS = 64;            // units per wave: a wave keeps its units' records in registers, one lane per unit

// ---- descriptor records (32-bit words in the blob; all offsets are FLOAT offsets into the LDS tile)
//   pair    [4] : oa | ob << 16 ; tra | type << 16 ; p0 (SS: r_a + r_b, LS: r + LINE_MIN_DIST) ; p1 (LS: L / 2)
//   pairhm  [1] : hm_a | hm_b << 16, hm = owned index << 8 | position in that entity's pair list (0xffff: static entity)
//   unit    [6] : o_row | tr_row << 16 ; type | n << 8 | partner stride (rows) << 16 ; first partner's offset | first pair
//                 << 16 ; half length (LS) ; threshold (SS: (r_a + r_b + 1e-4)^2, LS: r + LINE_MIN_DIST + slack) ; the pairs'
//                 bounding-circle threshold (circles_overlap: the lazy exact broad phase).  A unit =
//                 one "row" entity (a line, or the a-sphere of sphere-sphere pairs) against a RUN of n <= 6 partner spheres
//                 that are equally spaced in the tile, have consecutive pair indices and share the threshold - partner i
//                 is at offset + i * stride rows, its pair is first pair + i: nothing per partner is fetched.  Ordered wave
//                 by wave, <= 64 per wave.
//   owned   [32]: entity, oe, list_begin, n_list, n_a, tr_off (-1: none), -, -, DevEntity[17], -
//   list    : 16-bit entries, two per word: pair (13 bits) | adds a torque (bit 14: a line-sphere pair seen from a rotatable
//             line) | side << 15 (1: the entity is b)
//   wave    [2] : first unit, end unit
//   entoff  [nE]: tile offset of an entity's first row (-1: none of its rows is in the tile)
//   bounds  [nP]: the pair's bounding-circle sum R_a + R_b (World.collides core.py:2797-2801, in-kernel exact broad phase)
//   band    [nP]: circles_overlap's threshold of the pair (the lazy exact broad phase)
constexpr int PAIR_W = 4, UNIT_W = 6, OWNED_W = 32, WAVE_W = 2;

struct DevCompact {
  const uint32_t* blob;
  int32_t blob_words;           // multiple of 4
  // which rows of the state the tile holds (bit e): a dynamic entity keeps all six, a static one that is in a pair its
  // position; cos / sin rows for the lines among them.  Offsets follow from popcounts - no descriptor fetch in front of
  // the state loads.
  unsigned long long dyn_mask, static_mask, line_mask;
  int32_t off_af;               // agent force rows
  int32_t off_tr;               // trig rows (2 per line: cos, sin)
  int32_t off_tab;              // the blob copy
  int32_t t_owned, t_lists, t_units, t_pairs, t_pairhm, t_waves, t_entoff, t_bounds, t_band;  // word offsets inside the blob
  int32_t n_owned, n_pairs, hw;
  int32_t off_dyn;              // per-substep scratch: cnt[4] | hit[n_owned][hw] | ballots[n_pairs] (u64) | base[n_pairs] |
                                // keys[CAP] | contacts[CAP] (fx, fy) | [torques[CAP] if has_torque] | xmask[mask_words] ...
                                // (the exact broad phase's words: barrier form xmask | gmask; lazy form [2 parities][overlap |
                                //  band] | need | collected | batch mask | flag [4])
  int32_t has_torque;           // some pair exerts a torque (a rotatable line): the contacts carry a third number
  int32_t mask_words;
  const float4* trig_cache;     // [nE] {rotation, cos, sin, valid} of the static lines, made once from environment 0 (may be NULL)
};

}  // namespace compact

// One launch of the compacted kernel (defined in vmas_compact.hip, its own translation unit): env_kind = ENV_NONE |
// ENV_INGEST | ENV_FOOTBALL (`env` may be NULL for ENV_NONE), own = 1 | 2 | 4 dynamic entities per wave.  0 ok, -1 error
// (vmas_last_error).
int vmas_compact_launch(int env_kind, int own, int nw, size_t lds_bytes, int device, const DevWorld& W,
                        const compact::DevCompact& P, float* state, float* agent_ft, long ld, int batch, int padded,
                        const DevStepArgs& args, const DevEnv* env, hipStream_t stream);
// fills P.trig_cache from environment 0 of `state` (static lines: entities of line_mask that are not in dyn_mask)
int vmas_compact_fill_trig(const compact::DevCompact& P, const float* state, long ld, float4* cache, hipStream_t stream);

#ifdef VMAS_COMPACT_KERNELS
namespace compact {

__device__ __forceinline__ uint32_t lanemask_rank(unsigned long long b) {  // set bits of b below this lane
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
}
__device__ __forceinline__ unsigned long long sgpr64(unsigned long long v) {
  return ((unsigned long long)(uint32_t)sgpr((int)(v >> 32)) << 32) | (uint32_t)sgpr((int)(uint32_t)v);
}
__device__ __forceinline__ uint32_t rdl(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ float rdlf(uint32_t v, int l) { return __uint_as_float(rdl(v, l)); }

// cos / sin of the static lines' rotations in environment 0, with the very sincosf the step kernel uses
__global__ void compact_trig_kernel(unsigned long long lines, const float* __restrict__ state, long ld, float4* __restrict__ cache) {
  const int e = threadIdx.x;
  if (!((lines >> e) & 1ull)) return;
  const float rot = state[((long)e * 6 + 4) * ld];
  float sn, cs;
  sincosf(rot, &sn, &cs);
  cache[e] = make_float4(rot, cs, sn, 1.f);
}

// ENV: ENV_NONE | ENV_INGEST (action ingest as the prologue) | ENV_FOOTBALL (+ football.py's post-step as the epilogue)
// PLAIN: the launch has none of the optional inputs (pair masks / in-kernel exact broad phase, per-environment gravity,
// partial substep ranges): their tests and the scalar registers that carry their pointers are compiled out
// (PLAIN: 0 every option at run time but the lazy exact broad phase | 1 none of them | 2 none but the lazy exact broad phase:
//  its words and the owners' questions cost the kernel registers - three resident tiles per CU need <= 80)
template <int ENV, class EnvArgs, int OWN, int PLAIN>
__global__ __launch_bounds__(TILE * MAX_WAVES)
// (the 8-wave geometry of big grids - two owned entities per wave - holds three tiles per CU at <= 80 registers: six waves
//  per SIMD.  The lazy variant's collect loop would take it to 97; told to stay, the compiler spills a cold value instead)
__attribute__((amdgpu_waves_per_eu((PLAIN == 2 && OWN == 2 && ENV != ENV_FOOTBALL) ? 6 : 1))) void step_kernel_compact(DevWorld W, DevCompact P, float* __restrict__ state,
                                                                        float* __restrict__ agent_ft, long ld, int batch,
                                                                        int padded, DevStepArgs args_in, const EnvArgs E) {
  if constexpr (ENV != ENV_NONE) {  // a gated launch behind a validation that raised flags: not a single load or store
    if (env_gate_closed(E)) return;
  }
  DevStepArgs args = args_in;
  if constexpr (PLAIN != 0) {
    args.pair_mask = nullptr; args.sync = nullptr; args.entity_gravity = nullptr; args.first_substep = 0; args.n_substeps = 0;
#ifndef VMAS_TRACE
    args.trace = nullptr;
#endif
  }
  extern __shared__ float lds[];
  const int lane = threadIdx.x & (TILE - 1);
  const int wv = sgpr(threadIdx.x >> 6);
  // (measured in round 4: the waves per tile as an explicit kernel argument instead of this load from the implicit ones
  //  changes nothing at 16 384 environments - 16.05 us either way - and tips the register allocation of the football
  //  forms, 100+ scalar registers spilled to vector lanes, into a scratch frame: the K-step rollout ran 12 x slower)
  const int nw = sgpr(blockDim.x >> 6);
  const int nA = W.nA;
#ifdef VMAS_TRACE  // profiling build only (scripts/trace_compact.py): per-wave s_memtime stamps and phase sums
  unsigned long long tr_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tr_t = 0;  // (8: lazy overlap phase, 9: lazy resolve)
#define CSTAMP(k) if (args.trace && lane == 0) args.trace[((long)blockIdx.x * 16 + wv) * 16 + (k)] = __builtin_amdgcn_s_memtime()
#define CACC(k) if (args.trace) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tr_acc[k] += t_ - tr_t; tr_t = t_; }
#define CCOUNT(k, v) if (args.trace) tr_acc[k] += (v)
#else
#define CSTAMP(k)
#define CACC(k)
#define CCOUNT(k, v)
#endif
  CSTAMP(0);
  const long env = (long)blockIdx.x * TILE + lane;
  const bool live = env < batch;
  const bool lv = live || padded;  // planes padded to whole tiles: the tail lanes read the padding columns
  const unsigned long long live_mask = __ballot(live);
  float* tile = lds + lane;
  const uint32_t* tab = (const uint32_t*)(lds + P.off_tab);
  uint32_t* dyn = (uint32_t*)(lds + P.off_dyn);
  uint32_t* cnt = dyn;                                       // [0] contacts of this round
  uint32_t* hit = dyn + 4;                                   // [n_owned][hw]
  unsigned long long* ballots = (unsigned long long*)(hit + ((P.n_owned * P.hw + 1) & ~1));
  uint32_t* base = (uint32_t*)(ballots + P.n_pairs);
  uint32_t* keys = base + ((P.n_pairs + 1) & ~1);
  float2* contacts = (float2*)(keys + CAP);   // (8-byte aligned: every part in front of it has an even number of words)
  float* torques = (float*)(contacts + CAP);  // [CAP] only if P.has_torque
  uint32_t* xmask = (uint32_t*)(torques + (P.has_torque ? CAP : 0));
  const int nP = P.n_pairs;
  // the LAZY exact broad phase (vmas_env_device.h): [2 parities][overlap | band words] | need | collected | batch mask | flag
#ifndef VMAS_LZ_CUT  // (compile-time cuts of the lazy form's parts: scripts/variant_lib.sh, measurement only)
#define VMAS_LZ_CUT 0
#endif
#ifdef VMAS_PROFILE  // (A/B of the lazy form's parts, profiling builds only: VMAS_ABLATE bits 8.. - scripts/gpu_run.sh lazy-parts)
#define LZ_ABL(bit) (((args.ablate >> (bit)) & 1) != 0)
#else
#define LZ_ABL(bit) false
#endif
  constexpr bool lazy = PLAIN == 2;  // (the host launches this variant with the lazy form's arguments, and only then; launches
                                     //  with other options AND the lazy form take the interpreter)
  const int lzp = (P.mask_words + 3) & ~3;
  uint32_t* lz_words = xmask;  // [2 parities][lzp]: the pairs some environment of this tile overlaps, per pass

  // tile offset of entity e's first row / of its cos row, from the masks in the kernel arguments
  auto ent_off = [&](int e) {
    const unsigned long long below = (1ull << e) - 1ull;
    return (6 * __builtin_popcountll(P.dyn_mask & below) + 2 * __builtin_popcountll(P.static_mask & below)) * ROWF;
  };

  // ---- the agents' force rows: from agent_ft, or made from the caller's action tensors / the library's agent scripts
  long act_row0 = 0;
  auto load_agent_ft = [&](int a, float* f3, bool from_tile) {
    const float* src = agent_ft + (long)a * 3 * ld + env;
    if constexpr (ENV != ENV_NONE) {
      const bool on = E.has_ingest;
      const VmasActionSlot& S = E.ingest.agents[a];
      if (on && (S.action != nullptr || S.action_index != nullptr)) {
        uint32_t bad = 0;
        ingest_slot(S, E.ingest.clamp, env, live, agent_ft, ld, f3, bad, act_row0);
        if (S.action_size < 3) f3[2] = from_tile ? tile[P.off_af + (a * 3 + 2) * ROWF] : (lv ? src[2 * ld] : 0.f);
        if (E.err_flags != nullptr && bad != 0) raise_action_error(E.err_flags, bad);
        return;
      }
      if (on && E.ingest.n_scripts > 0 && E.script_of_agent[a] >= 0) {  // driven by the state about to be stepped:
        const VmasAgentScript& SC = E.ingest.scripts[E.script_of_agent[a]];
        if (from_tile) run_script(SC, tile + ent_off(SC.entity), ROWF, env, live, agent_ft, ld, f3);
        else run_script(SC, state + (long)SC.entity * 6 * ld + env, ld, env, live, agent_ft, ld, f3);
        f3[2] = from_tile ? tile[P.off_af + (a * 3 + 2) * ROWF] : (lv ? src[2 * ld] : 0.f);
        return;
      }
    }
    if (from_tile) return;  // (no action source: the rows of the previous step stay)
#pragma unroll
    for (int f = 0; f < 3; ++f) f3[f] = lv ? src[f * ld] : 0.f;
  };

  // ---- HBM -> LDS.  Entities with rows the path reads (dynamic: all six; static and in a pair: the position, and a
  //      line's rotation for its cos / sin), the agent forces and the descriptor blob: EVERY global load of the wave is
  //      issued before the first LDS store, so that the phase costs one HBM latency, not one per entity (a load -> store
  //      loop waits for each entity's rows in turn: 9 k of the kernel's 47 k cycles per wave in the first version).
  {
    constexpr int LB = 4;  // entities per wave and batch (4 x 8 waves >= football's 23 entities: one batch)
    static_assert(LB == 4, "the load table holds four entities per (batch, wave): one uint4");
    // the planner's load table (the blob in global memory: its LDS copy is made by this very phase): per (batch, wave) the
    // LB entities the wave loads, one word each - bit 0 valid, 1 dynamic, 2 line | tile row of its first row << 3 | tile row
    // of its cos row << 13 | entity << 23 - as ONE 16-byte scalar load
    const uint4* etab = (const uint4*)(P.blob + P.t_entoff);
    const long env_ld = lv ? env : (long)batch - 1;  // (tail lanes of an unpadded plane read the last environment's column)
    const uint4* bsrc = (const uint4*)P.blob;
    uint4* bdst = (uint4*)(lds + P.off_tab);
    const int n4 = P.blob_words >> 2, nt = blockDim.x;
    // One batch of LB entities per wave.  The first batch is code of its own, outside the loop of the later ones (worlds of
    // more than LB * waves entities): inside a loop the compiler's wait insertion takes every register the previous
    // iteration loaded into for still pending and puts a vmcnt(0) in front of each entity's address arithmetic - the four
    // entities' loads went out one memory round trip after the other.
    auto load_batch = [&](const int e0, const bool first) {
      float v[LB][6];
      float4 tc[LB];
      const uint4 dq = etab[(e0 - wv) / LB + wv];  // (batch * nw + wave)
      const uint32_t desc[LB] = {dq.x, dq.y, dq.z, dq.w};
#pragma unroll
      for (int j = 0; j < LB; ++j) {
        const uint32_t d = desc[j];
        const int e = (int)(d >> 23);
        const bool on = d & 1u, is_dyn = d & 2u, is_line = d & 4u;
        const float* src = state + (long)(on ? e : 0) * 6 * ld + env_ld;
#pragma unroll
        for (int f = 0; f < 6; ++f) v[j][f] = 0.f;
        if (on) {
          v[j][0] = src[0]; v[j][1] = src[ld];
          if (d & 6u) v[j][4] = src[4 * ld];
          if (is_dyn) { v[j][2] = src[2 * ld]; v[j][3] = src[3 * ld]; v[j][5] = src[5 * ld]; }
        }
        // (a scalar load with its own wait per entity - four in a row in front of the tile's first barrier.  Measured in round 4,
        //  same box, the previous build beside it: as vector loads, all in flight together, 16 384 environments 16.38 -> 16.2 us
        //  but 16 more vector registers took the kernel from 6 to 5 waves per SIMD - 131 072 environments 65.3 -> 76.9 us
        //  (profiles/r04h_ab_*.jsonl); as branch-free SCALAR loads in flight with the rows, no register more: 16.33 -> 16.34,
        //  65.3 -> 65.6 (r04k_ab_*.jsonl) - the 7.4 k cycles between a wave's start and its last load request
        //  (r04k_football16384_compact_phase_trace.txt) are not these loads)
        // (branch-free, at an address that is always valid - the entity's entry, or the blob's first bytes: requested with the
        //  rows, no wait of its own; whether it IS a cache entry is asked where it is used)
        tc[j] = *((is_line && !is_dyn && P.trig_cache != nullptr) ? P.trig_cache + e : (const float4*)P.blob);
      }
      // (first batch only) the agent forces that are plain loads, and this thread's share of the blob
      auto plain_row = [&](int a) {
        bool plain = a < nA;
        if constexpr (ENV != ENV_NONE) {
          if (a < nA && E.has_ingest) {
            const VmasActionSlot& S = E.ingest.agents[a];
            plain = !(S.action != nullptr || S.action_index != nullptr) && !(E.ingest.n_scripts > 0 && E.script_of_agent[a] >= 0);
          }
        }
        return plain;
      };
      const bool pa0 = first && plain_row(wv), pa1 = first && plain_row(wv + nw);
      const float* fs0 = agent_ft + (long)(pa0 ? wv : 0) * 3 * ld + env;
      const float* fs1 = agent_ft + (long)(pa1 ? wv + nw : 0) * 3 * ld + env;
      const float g00 = (pa0 && lv) ? fs0[0] : 0.f, g01 = (pa0 && lv) ? fs0[ld] : 0.f, g02 = (pa0 && lv) ? fs0[2 * ld] : 0.f;
      const float g10 = (pa1 && lv) ? fs1[0] : 0.f, g11 = (pa1 && lv) ? fs1[ld] : 0.f, g12 = (pa1 && lv) ? fs1[2 * ld] : 0.f;
      const int bi0 = (int)threadIdx.x, bi1 = (int)threadIdx.x + nt;
      uint4 bq0 = make_uint4(0, 0, 0, 0), bq1 = make_uint4(0, 0, 0, 0);
      if (first) { bq0 = bsrc[bi0 < n4 ? bi0 : 0]; bq1 = bsrc[bi1 < n4 ? bi1 : 0]; }
      if (first) { CSTAMP(12); }  // (trace builds: every load of the first batch is requested)
      // ---- stores (and the lines' cos / sin)
#pragma unroll
      for (int j = 0; j < LB; ++j) {
        const uint32_t d = desc[j];
        if (!(d & 1u)) continue;
        const bool is_dyn = d & 2u, is_line = d & 4u;
        float* dst = tile + (int)((d >> 3) & 1023u) * ROWF;
        if (is_dyn) {
#pragma unroll
          for (int f = 0; f < 6; ++f) dst[f * ROWF] = v[j][f];
        } else {
          dst[0] = v[j][0]; dst[ROWF] = v[j][1];
        }
        if (is_line) {
          // cos / sin of a STATIC line: the ~110-instruction sincosf is skipped when every environment of the tile has the
          // rotation the world's cache entry was made from - by the same sincosf, once (compact_trig_kernel): the same bits
          const float rot = v[j][4];
          float sn, cs;
          bool cached = false;
          if (!is_dyn && P.trig_cache != nullptr && tc[j].w != 0.f && __all(__float_as_uint(rot) == __float_as_uint(tc[j].x) || !live)) {
            cs = tc[j].y; sn = tc[j].z; cached = true;
          }
          if (!cached) sincosf(rot, &sn, &cs);
          const int tr = (int)((d >> 13) & 1023u) * ROWF;
          tile[tr] = cs;
          tile[tr + ROWF] = sn;
        }
      }
      if (first) {
        if (pa0) { float* dst = tile + P.off_af + wv * 3 * ROWF; dst[0] = g00; dst[ROWF] = g01; dst[2 * ROWF] = g02; }
        if (pa1) { float* dst = tile + P.off_af + (wv + nw) * 3 * ROWF; dst[0] = g10; dst[ROWF] = g11; dst[2 * ROWF] = g12; }
        if (bi0 < n4) bdst[bi0] = bq0;
        if (bi1 < n4) bdst[bi1] = bq1;
      }
    };
    load_batch(wv, true);
    for (int e0 = wv + LB * nw; e0 < W.nE; e0 += LB * nw) load_batch(e0, false);
    CSTAMP(13);  // (trace builds: the entity rows are in LDS)
    // agents beyond the first two per wave, and agents whose forces are made from actions / scripts (the ingest prologue)
    for (int a = wv; a < nA; a += nw) {
      bool done = (a - wv) / nw < 2;
      if constexpr (ENV != ENV_NONE) {
        if (done && E.has_ingest) {
          const VmasActionSlot& S = E.ingest.agents[a];
          done = !(S.action != nullptr || S.action_index != nullptr) && !(E.ingest.n_scripts > 0 && E.script_of_agent[a] >= 0);
        }
      }
      if (done) continue;
      float f3[3];
      load_agent_ft(a, f3, false);
      float* dst = tile + P.off_af + a * 3 * ROWF;
#pragma unroll
      for (int f = 0; f < 3; ++f) dst[f * ROWF] = f3[f];
    }
    for (int i = (int)threadIdx.x + 2 * nt; i < n4; i += nt) bdst[i] = bsrc[i];  // (blobs beyond 32 bytes per thread)
    const int n_zero = 4 + P.n_owned * P.hw;
    for (int i = threadIdx.x; i < n_zero; i += blockDim.x) dyn[i] = 0u;
    if (args.sync != nullptr)
      for (int i = threadIdx.x; i < P.mask_words; i += blockDim.x) xmask[i] = 0u;
    if (lazy) {
      for (int i = threadIdx.x; i < 2 * lzp; i += blockDim.x) lz_words[i] = 0u;
    }
  }
  [[maybe_unused]] float fb_prev[4] = {0.f, 0.f, 0.f, 0.f};
  [[maybe_unused]] float post_steps = 0.f;
  if constexpr (ENV == ENV_FOOTBALL) {
#pragma unroll
    for (int k = 0; k < 4; ++k) fb_prev[k] = live ? E.football.o.pos_shaping[(long)k * batch + env] : 0.f;
    if (wv == 0) post_steps = (E.football.o.limit.steps != nullptr && live) ? E.football.o.limit.steps[env] : 0.f;
  }
  CSTAMP(1);
  __syncthreads();
  CSTAMP(2);

  // ---- this wave's records, one LDS round trip for the whole launch: its broad-phase units (unit u in lane u), its owned
  //      entities (word j of the record in lane j) and their pair lists (entry j in lane j).  Phases A and C read them with
  //      v_readlane - no descriptor fetch on any dependent chain.
  const int u0 = sgpr((int)tab[P.t_waves + wv * WAVE_W]), nu = sgpr((int)tab[P.t_waves + wv * WAVE_W + 1]) - u0;
  uint32_t HU0 = 0, HU1 = 0, HU2 = 0, HU3 = 0, HU4 = 0;
  [[maybe_unused]] uint32_t HU5 = 0;
  if (lane < nu) {
    const uint32_t* U = tab + P.t_units + (u0 + lane) * UNIT_W;
    HU0 = U[0]; HU1 = U[1]; HU2 = U[2]; HU3 = U[3]; HU4 = U[4];
    if constexpr (PLAIN == 2) HU5 = U[5];
  }
  uint32_t OWv[OWN], LV[OWN];
#pragma unroll
  for (int s = 0; s < OWN; ++s) {
    const int k = wv + s * nw;
    OWv[s] = 0u; LV[s] = 0u;
    if (k < P.n_owned) {
      if (lane < OWNED_W) OWv[s] = tab[P.t_owned + k * OWNED_W + lane];
      const int list0 = (int)rdl(OWv[s], 2), n_list = (int)rdl(OWv[s], 3);
      if (lane < n_list) LV[s] = ((const uint16_t*)(tab + P.t_lists))[list0 + lane];
    }
  }
#ifdef VMAS_TRACE
  tr_t = __builtin_amdgcn_s_memtime();
#endif

  const float sub_dt = W.sub_dt;
  const int s_begin = args.first_substep;
  const int s_end = s_begin + (args.n_substeps > 0 ? args.n_substeps : W.substeps - s_begin);
  const int n_steps = args.n_steps > 1 ? args.n_steps : 1;
  int it = 0;
  for (int stp = 0; stp < n_steps; ++stp) {
    float* aft = agent_ft + (long)stp * args.ft_stride;
    if (stp > 0) {  // multi-step launch: only this step's agent forces (or the actions they are made of) come from HBM
      bool ingested = false;
      if constexpr (ENV != ENV_NONE) {
        if (E.has_ingest) {
          ingested = true;
          act_row0 = (long)stp * batch;
          for (int a = wv; a < nA; a += nw) {
            float f3[3];
            float* dst = tile + P.off_af + a * 3 * ROWF;
#pragma unroll
            for (int f = 0; f < 3; ++f) f3[f] = dst[f * ROWF];
            load_agent_ft(a, f3, true);
#pragma unroll
            for (int f = 0; f < 3; ++f) dst[f * ROWF] = f3[f];
          }
        }
      }
      if (!ingested)
        for (int a = wv; a < nA; a += nw) {
          const float* src = aft + (long)a * 3 * ld + env;
          float* dst = tile + P.off_af + a * 3 * ROWF;
#pragma unroll
          for (int f = 0; f < 3; ++f) dst[f * ROWF] = lv ? src[f * ld] : 0.f;
        }
      __syncthreads();
    }
    for (int substep = s_begin; substep < s_end; ++substep, ++it) {
      const bool last_sub = substep + 1 == s_end;
      const bool last = last_sub && stp + 1 == n_steps;
      // which pairs the reference processes at all: the caller's recorded mask (global memory) and / or the batch's words the
      // barrier form leaves in LDS.  Two variables on purpose: ONE pointer that is global here and LDS there makes its null
      // test a generic-address-space one, which trips a back-end bug of ROCm 7.2 in some instantiations ("Illegal instruction
      // detected: V_CMP_NE_U32_e32 0, $src_shared_base")
      const uint32_t* pmask = args.pair_mask;
      bool bar_on = false;
      const uint32_t* bar_mask = xmask + ((P.mask_words + 3) & ~3);  // (LDS; valid behind grid_bits_collect)
      auto pair_masked_off = [&](int pr) {
        if (pmask != nullptr && !((mask_word(pmask, pr >> 5) >> (pr & 31)) & 1u)) return true;
        return bar_on && !((bar_mask[pr >> 5] >> (pr & 31)) & 1u);
      };
      // lazy form: this pass's overlap words (the other parity's are re-armed: last read by the previous pass's owners)
      uint32_t* lz_x = lz_words + lzp * (it & 1);
      if (lazy && !(VMAS_LZ_CUT & 16))
        for (int i = threadIdx.x; i < lzp; i += blockDim.x) lz_words[lzp * ((it + 1) & 1) + i] = 0u;
      if (args.sync != nullptr) {
        // World.collides' batch-global rule (core.py:2797-2801) for this substep by the whole grid: see step_kernel
        const uint32_t seq = args.seq0 + (uint32_t)it;
        for (int p = wv; p < nP; p += nw) {  // (descriptors from the LDS tables, not from global memory)
          const uint32_t q = (uint32_t)sgpr((int)tab[P.t_pairs + p * PAIR_W]);
          const float* A = tile + (int)(q & 0xffffu);
          const float* B = tile + (int)(q >> 16);
          const bool h = live && norm2(A[0] - B[0], A[ROWF] - B[ROWF]) <= __uint_as_float(tab[P.t_bounds + p]);
          if (__any(h) && lane == 0) atomicOr(&xmask[p >> 5], 1u << (p & 31));
        }
        __syncthreads();
        {  // one atomic per pair word = this tile's arrival + its bits; wave 0 collects the batch's words into LDS
          const int groups = ((int)gridDim.x + 31) >> 5;
          unsigned long long* base = (unsigned long long*)(args.sync + 4);
          const size_t stride = (size_t)args.mask_words * groups;
          uint32_t* gmask = xmask + ((P.mask_words + 3) & ~3);
          grid_bits_publish(base + (seq & 3u) * stride, args.mask_words, xmask);
          if ((int)threadIdx.x < args.mask_words) xmask[threadIdx.x] = 0u;  // (re-armed for the next substep by the thread that published it)
          if (wv == 0)
            grid_bits_collect(base + (seq & 3u) * stride, base + ((seq + 2u) & 3u) * stride, args.mask_words, gmask, args.sync + 1,
                              args.gave_up);
          __syncthreads();
          bar_on = true;  // (gmask == bar_mask)
        }
      }

      // ---- the owners' running sums (registers: they survive the rounds of an overflowing tile)
      v2 F[OWN];
      float Tq[OWN];
      uint32_t added_a[OWN], added_t[OWN];  // per lane: a-side force terms / computed torque terms actually added
      int n_on[OWN], n_on_a[OWN];           // uniform: the entity's pairs that are on in the pair mask (all / a-side)
#pragma unroll
      for (int s = 0; s < OWN; ++s) { F[s] = V(0.f, 0.f); Tq[s] = 0.f; added_a[s] = 0u; added_t[s] = 0u; n_on[s] = 0; n_on_a[s] = 0; }

      auto prologue = [&]() {
      // ---- prologue of every owned entity core.py:1995-2004 (the statements of step_kernel)
#pragma unroll
      for (int s = 0; s < OWN; ++s) {
        const int k = wv + s * nw;
        if (k >= P.n_owned) break;
        const int e = (int)rdl(OWv[s], 0);
        const float* Es = tile + (int)rdl(OWv[s], 1);
        const uint32_t fl = rdl(OWv[s], 8);
        const float mass = rdlf(OWv[s], 12), inertia = rdlf(OWv[s], 13);
        if (fl & VMAS_F_AGENT) {
          const int agent_index = (int)rdl(OWv[s], 10);
          float* Af = tile + P.off_af + agent_index * 3 * ROWF;
          if (fl & VMAS_F_MOVABLE) {
            v2 f = V(Af[0], Af[ROWF]);
            if (fl & (VMAS_F_MAX_F | VMAS_F_F_RANGE)) {
              if (fl & VMAS_F_MAX_F) f = clamp_with_norm(f, rdlf(OWv[s], 21));
              if (fl & VMAS_F_F_RANGE) { const float fr = rdlf(OWv[s], 22); f = V(clamp_t(f.x, fr), clamp_t(f.y, fr)); }
              Af[0] = f.x; Af[ROWF] = f.y;
              if (last_sub && live) {
                float* gf = aft + (long)agent_index * 3 * ld + env;
                gf[0] = f.x; gf[ld] = f.y;
              }
            }
            F[s] = F[s] + f;
          }
          if (fl & VMAS_F_ROTATABLE) {
            float t = Af[2 * ROWF];
            if (fl & (VMAS_F_MAX_T | VMAS_F_T_RANGE)) {
              if (fl & VMAS_F_MAX_T) {
                const float max_t = rdlf(OWv[s], 23);
                const float n = fabsf(t);
                const float nt = (t / n) * max_t;
                t = n > max_t ? nt : t;
              }
              if (fl & VMAS_F_T_RANGE) t = clamp_t(t, rdlf(OWv[s], 24));
              Af[2 * ROWF] = t;
              if (last_sub && live) aft[((long)agent_index * 3 + 2) * ld + env] = t;
            }
            Tq[s] = Tq[s] + t;
          }
        }
        if (fl & VMAS_F_LIN_FRICTION) F[s] = F[s] + friction2(V(Es[2 * ROWF], Es[3 * ROWF]), rdlf(OWv[s], 17), mass, sub_dt);
        if (fl & VMAS_F_ANG_FRICTION) Tq[s] = Tq[s] + friction1(Es[5 * ROWF], rdlf(OWv[s], 18), inertia, sub_dt);
        if (fl & VMAS_F_MOVABLE) {
          if (W.has_gravity) F[s] = F[s] + V(mass * W.gx, mass * W.gy);
          if (fl & VMAS_F_GRAVITY) {
            v2 ge = V(rdlf(OWv[s], 19), rdlf(OWv[s], 20));
            if (args.entity_gravity && live) {
              const float* gp = args.entity_gravity + (long)e * 2 * ld + env;
              ge = V(gp[0], gp[ld]);
            }
            F[s] = F[s] + V(mass * ge.x, mass * ge.y);
          }
        }
        // how many of the entity's pairs the reference processes at all (pair mask)
        const int n_list = (int)rdl(OWv[s], 3);
        if (pmask == nullptr && !bar_on) {
          n_on[s] = n_list; n_on_a[s] = (int)rdl(OWv[s], 4);
        } else {
          int c_all = 0, c_a = 0;
          for (int j = 0; j < n_list; ++j) {
            const uint32_t en = rdl(LV[s], j);
            const int pr = (int)(en & 0x1fffu);
            if (!pair_masked_off(pr)) { ++c_all; c_a += (en >> 15) ? 0 : 1; }
          }
          n_on[s] = c_all; n_on_a[s] = c_a;
        }
      }
      };
      prologue();
      CACC(0);  // prologue

      // a pair with contacts: its mask, the first slot of its contacts in the list, the lanes' keys, and the "has
      // contacts" bit in the masks of its two entities
      auto take_slots = [&](int pair, unsigned long long b) {
        const uint32_t c = (uint32_t)__builtin_popcountll(b);
        uint32_t s0 = 0;
        if (lane == 0) s0 = atomicAdd(cnt, c);
        s0 = (uint32_t)sgpr((int)s0);
        if ((b >> lane) & 1ull) {
          const uint32_t k = s0 + lanemask_rank(b);
          if (k < (uint32_t)CAP) keys[k] = ((uint32_t)pair << 6) | (uint32_t)lane;
        }
        if (lane == 0) {
          ballots[pair] = b;
          base[pair] = s0;
          const uint32_t hm = tab[P.t_pairhm + pair];
          const uint32_t ha = hm & 0xffffu, hb = hm >> 16;
          if (ha != 0xffffu) atomicOr(&hit[(ha >> 8) * P.hw + ((ha & 255u) >> 5)], 1u << (ha & 31u));
          if (hb != 0xffffu) atomicOr(&hit[(hb >> 8) * P.hw + ((hb & 255u) >> 5)], 1u << (hb & 31u));
        }
      };

      // ================= A: broad phase, lane = environment.  Per unit: the row entity's rows and the partners' positions in
      // ONE LDS round trip (the unit's record comes from registers, partners and pairs are arithmetic), the tests, then -
      // rarely - slots for the pairs with contacts.  `all_masks`: the re-run of an overflowing tile, which stores EVERY
      // pair's mask (zeros too) and takes no slots.
      auto broad_phase = [&](bool all_masks) {
        for (int ul = 0; ul < nu; ++ul) {
          const uint32_t h0 = rdl(HU0, ul), h1 = rdl(HU1, ul), h2 = rdl(HU2, ul);
          const float half = rdlf(HU3, ul);
          const uint32_t key_lo = rdl(HU4, ul) + 1u, key_span = 0x7f800000u - key_lo;  // (threshold, +inf) in bit patterns
          [[maybe_unused]] uint32_t reach_bits = 0u;  // (lazy form) partner i has a contact candidate in some environment
          const int type = (int)(h1 & 0xffu), n = (int)((h1 >> 8) & 0xffu), stride = (int)(h1 >> 16) * ROWF;
          const int pair0 = (int)(h2 >> 16);
          const float* R = tile + (int)(h0 & 0xffffu);
          const float* S0 = tile + (int)(h2 & 0xffffu);
          const v2 pr = V(R[0], R[ROWF]);
          float cs = 0.f, sn = 0.f;
          if (type == VMAS_PAIR_LS) { cs = tile[(int)(h0 >> 16)]; sn = tile[(int)(h0 >> 16) + ROWF]; }
          v2 ps[UNIT_PARTNERS];
#pragma unroll
          for (int i = 0; i < UNIT_PARTNERS; ++i) {  // (slots beyond n repeat partner 0: always valid rows)
            const float* S = S0 + (i < n ? i : 0) * stride;
            ps[i] = V(S[0], S[ROWF]);
          }
#pragma unroll
          for (int i = 0; i < UNIT_PARTNERS; ++i) {
            if (i >= n) break;
            const int pair = pair0 + i;
            const float dx = pr.x - ps[i].x, dy = pr.y - ps[i].y;
            // One number per pair that must NOT lie in (threshold, +inf) for the pair to be in reach: the squared centre
            // distance (sphere-sphere), or the larger of the sphere's two gaps in the line's frame (eval_lsq) - beyond it the
            // force is exactly 0 (core.py:2836).  Only a FINITE excess skips: NaN / inf operands go to the narrow phase,
            // where the reference's own arithmetic decides.  Compared as integers: the bit patterns of non-negative floats,
            // +inf and +NaN included, are ordered like the values, so "threshold < m < inf" is ONE unsigned compare of
            // bits(m) - (bits(threshold) + 1) against a span computed once per unit; a negative gap (sign bit set) loses the
            // signed max against the other, non-negative one.
            uint32_t key;
            if (type == VMAS_PAIR_SS) {
              key = __float_as_uint(dx * dx + dy * dy);
            } else {
              const float along = fabsf(dx * cs + dy * sn) - half;
              const uint32_t perp = __float_as_uint(dy * cs - dx * sn) & 0x7fffffffu;
              const int a_ = (int)__float_as_uint(along), p_ = (int)perp;
              key = (uint32_t)(a_ > p_ ? a_ : p_);
            }
            const bool need = (key - key_lo) >= key_span;
            unsigned long long b = __ballot(need) & live_mask;
            if (b != 0ull && pair_masked_off(pair)) b = 0ull;
            if (all_masks) {
              if (lane == 0) ballots[pair] = b;
            } else if (b != 0ull) {
              take_slots(pair, b);
              if constexpr (PLAIN == 2) reach_bits |= 1u << i;  // (on the rare path: nothing of the lazy form in the loop's hot part)
            }
          }
          // lazy form: World.collides' own test of the pair per environment (core.py:2797-2799) - a line pair always (a wall's
          // bounding circle spans the pitch), a sphere unit only if some pair of it is within reach (overlapping spheres are).
          // A loop of ITS OWN behind the pair loop, on the positions that are still in registers: inside that loop the test's
          // threshold, its result bits and the ballot's scalar pair were live across the unit's twelve partners - and the
          // compiler paid for them with scalar spills in the kernel's hottest code: football 16 384 +2.9 us per step for tests
          // that take 0.8 us to execute (profiles/r06s_lazy_cuts.txt)
          if constexpr (PLAIN == 2) {
            if (lazy && !(VMAS_LZ_CUT & 1) && !LZ_ABL(8) && !all_masks && (type != VMAS_PAIR_SS || reach_bits != 0u)) {
              const float ov_thr = live ? rdlf(HU5, ul) : -1.f;  // (tail lanes never overlap: a squared distance is >= 0 or NaN)
              uint32_t ov_bits = 0u;  // partner i: some environment of the tile has the pair's circles overlapping
#pragma unroll
              for (int i = 0; i < UNIT_PARTNERS; ++i) {
                if (i >= n) break;
                if (__ballot(circles_overlap(pr.x - ps[i].x, pr.y - ps[i].y, ov_thr)) != 0ull) ov_bits |= 1u << i;
              }
              if (ov_bits != 0u && lane == 0) {  // (consecutive pair indices: one word, or two if the run crosses a word boundary)
                atomicOr(&lz_x[pair0 >> 5], ov_bits << (pair0 & 31));
                if ((pair0 & 31) + n > 32) atomicOr(&lz_x[(pair0 >> 5) + 1], ov_bits >> (32 - (pair0 & 31)));
              }
            }
          }
        }
      };
      broad_phase(false);
      // ================= A: broad phase, lane = environment.  Per unit: the row entity's rows and the partners' positions in
      // ONE LDS round trip (the records come from registers), the tests, then - rarely - slots for the pairs with contacts.
      CACC(1);  // A
      __syncthreads();
      CACC(2);  // A barrier
      if constexpr (PLAIN == 2) {
        if (lazy && !(VMAS_LZ_CUT & 8)) lazy_publish(args.lz, it, (int)(lz_x - (uint32_t*)lds));  // this tile's words go out: nobody waits for them
      }

      // ---- rounds: ONE unless the tile has more contacts than the list holds (non-finite poses pass every test); then
      //      the pairs are taken in runs of at most CAP contacts, in pair order, slots re-assigned from the stored masks
      int lo = 0, hi = nP;
      bool rounds = sgpr((int)cnt[0]) > CAP;
      if (threadIdx.x == 0 && args.contacts != nullptr)  // (what the host chooses the kernel by, vmas_hip.hip CompactAdapt)
        atomicAdd(args.contacts, (unsigned long long)cnt[0]);
      if (rounds) {
        __syncthreads();  // (every wave has read the count)
        if (threadIdx.x == 0) cnt[0] = 0u;
        for (int i = threadIdx.x; i < P.n_owned * P.hw; i += blockDim.x) hit[i] = 0u;
        broad_phase(true);  // every pair's mask, zeros included (the first pass stored those with contacts only)
        __syncthreads();
        hi = 0;
      }
      for (;;) {
        if (rounds) {
          // the longest run [lo, hi) with at most CAP contacts: a wave-wide prefix sum over the pairs' counts, 64 at a time
          lo = hi;
          int acc = 0;
          bool full = false;
          while (!full && hi < nP) {
            const int p = hi + lane;
            int c = p < nP ? __builtin_popcountll(ballots[p]) : 0;
            int pre = c;
#pragma unroll
            for (int d = 1; d < TILE; d <<= 1) { const int t = __shfl_up(pre, d); if (lane >= d) pre += t; }
            const unsigned long long over = __ballot(acc + pre > CAP);
            const int fit = over ? __builtin_ctzll(over) : TILE;  // pairs of this chunk that still fit
            const int take = fit < nP - hi ? fit : nP - hi;
            acc += take > 0 ? __builtin_amdgcn_readlane(pre, take - 1) : 0;
            hi += take;
            full = over != 0ull;
          }
          __syncthreads();  // (hit / cnt cleared, the previous round's contacts consumed)
          for (int p = lo + wv; p < hi; p += nw) {
            const unsigned long long b = sgpr64(ballots[p]);
            if (b != 0ull) take_slots(p, b);
          }
          __syncthreads();
        }
        const uint32_t N = (uint32_t)sgpr((int)cnt[0]);
        CCOUNT(6, N);
        CCOUNT(7, 1);
        // ================= B: narrow phase, lane = contact
        for (uint32_t k = (uint32_t)((nw - 1 - wv) * TILE + lane); k < N; k += (uint32_t)(nw * TILE)) {  // (the last wave first:
                                                                                                  //  it owns the fewest entities)
          const uint32_t key = keys[k];
          const int pair = (int)(key >> 6);
          const float* col = lds + (key & 63u);
          const uint4 Q = *(const uint4*)(tab + P.t_pairs + pair * PAIR_W);
          const float* A = col + (Q.x & 0xffffu);
          const float* B = col + (Q.x >> 16);
          const v2 pa = V(A[0], A[ROWF]), pb = V(B[0], B[ROWF]);
          const float p0 = __uint_as_float(Q.z);
          v2 fa;
          float ta = 0.f;
          if ((Q.y >> 16) == (uint32_t)VMAS_PAIR_SS) {  // core.py:2294-2339
            fa = contact_force(pa, pb, p0, W.c_coll, W.k);
          } else {  // a = line, b = sphere  core.py:2341-2392
            const float* T = col + (Q.y & 0xffffu);
            const v2 cp = closest_point_line<true>(pa, T[0], T[ROWF], __uint_as_float(Q.w), pb);
            fa = -contact_force(pb, cp, p0, W.c_coll, W.k);
            ta = vcross(cp - pa, fa);
          }
          // lazy form: this environment is in the pair's band - bounding circles apart, force (or torque) not zero; for a
          // sphere pair that takes a non-finite pose.  Noted in the contact's key: the pair's owners look at it (phase C)
          if constexpr (PLAIN == 2) {
            if (lazy && !(VMAS_LZ_CUT & 2) && !LZ_ABL(10) && (fa.x != 0.f || fa.y != 0.f || ta != 0.f) &&
                !circles_overlap(pa.x - pb.x, pa.y - pb.y, __uint_as_float(tab[P.t_band + pair])))
              keys[k] = key | 0x80000000u;
          }
          contacts[k] = make_float2(fa.x, fa.y);
          if (P.has_torque) torques[k] = ta;
        }
        CACC(3);  // B
        __syncthreads();
        CACC(4);  // B barrier
        // ================= C (first half): every owner adds the contacts of its entity, in the reference's order
        if (threadIdx.x == 0) cnt[0] = 0u;  // (read by every wave before the barrier above; next written in a later phase A)
#pragma unroll
        for (int s = 0; s < OWN; ++s) {
          const int k = wv + s * nw;
          if (k >= P.n_owned) break;
          const uint32_t fl = rdl(OWv[s], 8);
          for (int w_ = 0; w_ < P.hw; ++w_) {
            uint32_t m = (uint32_t)sgpr((int)hit[k * P.hw + w_]);
            if (m == 0u) continue;
            if (lane == 0) hit[k * P.hw + w_] = 0u;  // (owner-private: re-armed for the next round / substep)
            while (m) {
              const int j = __builtin_ctz(m) + 32 * w_;
              m &= m - 1u;
              const uint32_t en = rdl(LV[s], j);
              const int pr = (int)(en & 0x1fffu);
              const bool is_b = (en >> 15) != 0u, has_t = ((en >> 14) & 1u) != 0u;
              const unsigned long long b = sgpr64(ballots[pr]);
              const uint32_t s0 = (uint32_t)sgpr((int)base[pr]);
              const bool mine = (b >> lane) & 1ull;
              if constexpr (PLAIN == 2) {
                // lazy form: one of this entity's contacts of the pair is a BAND contact (circles apart, force not zero) and no
                // environment of the tile has the pair's circles overlapping: the batch decides (World.collides core.py:2797-
                // 2801).  THIS WAVE asks - the other waves go on with their entities - and leaves the pair's contacts out of
                // the sum if no environment of the batch overlaps (the pair's other owner asks too and gets the same answer: a
                // set bit is final, and so is "every tile has arrived and it is clear").  Rare: see vmas_env_device.h.
                // (Measured instead, r06u / r06v_lazy_cuts.txt: the whole tile asking behind phase B with lazy_collect - 7 us
                // per launch where this costs 1.9; wave 0 asking there for everybody - 3.4: code between the phases sits on top of
                // every owner's running sums, and the compiler pays for its registers in the hot loops.)
                if (lazy && !(VMAS_LZ_CUT & 4)) {
                  const bool band = mine && (keys[s0 + lanemask_rank(b)] >> 31) != 0u;
                  if (__any(band) && !((lz_x[pr >> 5] >> (pr & 31)) & 1u) && lazy_ask_wave(args.lz, it, pr) == 0) {
                    n_on[s] -= 1;  // (the reference does not process the pair at all: not one of the zeros it would add)
                    n_on_a[s] -= is_b ? 0 : 1;
                    continue;
                  }
                }
              }
              float2 c = make_float2(0.f, 0.f);
              float ct = 0.f;
              if (mine) {
                c = contacts[s0 + lanemask_rank(b)];
                if (has_t) ct = torques[s0 + lanemask_rank(b)];
              }
              const uint32_t flip = is_b ? 0x80000000u : 0u;  // b's side: -f (an integer xor: see eval_item)
              const v2 f = V(__uint_as_float(__float_as_uint(c.x) ^ flip), __uint_as_float(__float_as_uint(c.y) ^ flip));
              if (fl & VMAS_F_MOVABLE) {
                const v2 Fn = F[s] + f;
                F[s] = mine ? Fn : F[s];
                if (!is_b) added_a[s] += mine ? 1u : 0u;
              }
              if (has_t) {  // the line's torque (the sphere's term is the reference's literal 0: part of the `+ 0` below)
                const float Tn = Tq[s] + ct;
                Tq[s] = mine ? Tn : Tq[s];
                added_t[s] += mine ? 1u : 0u;
              }
            }
          }
        }
        CACC(5);  // C: adding the contacts
        if (!rounds || hi >= nP) break;
      }

      // ================= C (second half): the zeros that were skipped, then _integrate_state core.py:2862-2908
#pragma unroll
      for (int s = 0; s < OWN; ++s) {
        const int k = wv + s * nw;
        if (k >= P.n_owned) break;
        const int e = (int)rdl(OWv[s], 0);
        float* Es = tile + (int)rdl(OWv[s], 1);
        const int tr_off = (int)rdl(OWv[s], 5);
        const uint32_t fl = rdl(OWv[s], 8);
        const float one_minus_drag = rdlf(OWv[s], 14);
        if ((fl & VMAS_F_MOVABLE) && added_a[s] < (uint32_t)n_on_a[s]) F[s] = F[s] + V(0.f, 0.f);
        if ((fl & VMAS_F_ROTATABLE) && added_t[s] < (uint32_t)n_on[s]) Tq[s] = Tq[s] + 0.f;
        float es[6];
#pragma unroll
        for (int f = 0; f < 6; ++f) es[f] = Es[f * ROWF];
        float* dst = state + (long)e * 6 * ld + env;
        if (fl & VMAS_F_MOVABLE) {
          v2 vel = V(es[2], es[3]);
          if (substep == 0) vel = V(vel.x * one_minus_drag, vel.y * one_minus_drag);
          const rcp_t rm = rcp_of(rdlf(OWv[s], 12));
          const v2 acc = V(F[s].x / rm, F[s].y / rm);
          vel = V(vel.x + acc.x * sub_dt, vel.y + acc.y * sub_dt);
          if (fl & VMAS_F_MAX_SPEED) vel = clamp_with_norm(vel, rdlf(OWv[s], 15));
          if (fl & VMAS_F_V_RANGE) { const float vr = rdlf(OWv[s], 16); vel = V(clamp_t(vel.x, vr), clamp_t(vel.y, vr)); }
          v2 np = V(es[0] + vel.x * sub_dt, es[1] + vel.y * sub_dt);
          if (W.xs == W.xs) np.x = clamp_t(np.x, W.xs);
          if (W.ys == W.ys) np.y = clamp_t(np.y, W.ys);
          if (last && live) { dst[0] = np.x; dst[ld] = np.y; dst[2 * ld] = vel.x; dst[3 * ld] = vel.y; }
          if (!last || ENV != ENV_NONE) { Es[0] = np.x; Es[ROWF] = np.y; Es[2 * ROWF] = vel.x; Es[3 * ROWF] = vel.y; }
        }
        if (fl & VMAS_F_ROTATABLE) {
          float av = es[5];
          if (substep == 0) av = av * one_minus_drag;
          av = av + (Tq[s] / rdlf(OWv[s], 13)) * sub_dt;
          const float rot = es[4] + av * sub_dt;
          if (last && live) { dst[4 * ld] = rot; dst[5 * ld] = av; }
          if (!last || ENV != ENV_NONE) { Es[4 * ROWF] = rot; Es[5 * ROWF] = av; }
          if (!last && tr_off >= 0) {
            float sn, cs;
            sincosf(rot, &sn, &cs);
            tile[tr_off] = cs;
            tile[tr_off + ROWF] = sn;
          }
        }
      }
      if (!last) __syncthreads();
      CACC(0);  // (integration + its barrier: counted with the prologue)
    }
    if constexpr (ENV == ENV_FOOTBALL) {
      if (stp + 1 == n_steps) __syncthreads();  // (earlier steps: the substep loop ended with a barrier)
      {
        const float* rows = tile + ent_off(E.football.d.agent0);  // (host-checked: the agents and the ball are consecutive
        const float* af = tile + P.off_af;                        //  dynamic entities, agent index = slot)
        // observation staging (E.scratch_off >= 0: the host found room): a [64][17] tile per wave - the first waves in
        // the per-substep scratch from the ballots to the contact list (ballots | base | keys | forces | torques: written
        // before they are read in every substep, and the next one is a barrier away), the others behind the kernel's own LDS
        int slab = -1;  // (its float offset from the LDS base: football_post_tile)
        if (E.scratch_off >= 0) {
          constexpr int kSlab = 64 * (kFootballStageChunk + 1);
          const int in_dead = (int)((const float*)xmask - (const float*)ballots) / kSlab;
          slab = wv < in_dead ? (int)((const float*)ballots - lds) + wv * kSlab : E.scratch_off + (wv - in_dead) * kSlab;
        }
        football_post_tile(TileCtx(batch), E.football.d, E.football.o, batch,
                           [&](int slot, int k) { return k < 4 ? rows[(slot * 6 + k) * ROWF] : af[(slot * 3 + (k - 4)) * ROWF]; },
                           slab, kFootballStageChunk, fb_prev, post_steps, stp);
      }
      if (stp + 1 < n_steps) __syncthreads();  // the next step's prologue rewrites the agent-force rows
    }
  }
  CSTAMP(3);
#ifdef VMAS_TRACE
  if (args.trace && lane == 0)
    for (int k = 0; k < 8; ++k) args.trace[((long)blockIdx.x * 16 + wv) * 16 + 4 + k] = tr_acc[k];
  if (args.trace && lane == 0) {
    args.trace[((long)blockIdx.x * 16 + wv) * 16 + 14] = tr_acc[8];
    args.trace[((long)blockIdx.x * 16 + wv) * 16 + 15] = tr_acc[9];
  }
#endif
}

}  // namespace compact
#endif  // VMAS_COMPACT_KERNELS
