// wgl_beam.hip -- K5: the WIDE schedule of the Wing-Gong/Lowe search (gfx950).
//
// Same search space, same memoisation and same answers (verdict, failing op) as
// wgl_search.hip / knossos.wgl, but scheduled for 64 lanes instead of one
// thread: per iteration the wavefront takes the K most recent configs off an
// explicit stack and expands ALL their successors at once, ONE (config, open
// call) PAIR PER LANE:
//
//   pop K parents -> LDS                    (entries are read back from the table)
//   pairs = sum of the parents' open calls  (per-front lists built by pack_open.hip)
//   for each round of 64 pairs:
//     lane: open call -> model step -> child (front advance) -> visited-set probe
//     `__ballot` of "model-consistent" / "passed every completion" / "new config"
//     prefix-popcount of the "new" ballot compacts the successors onto the stack
//
// which is the shape BASELINE.json's north_star describes (ballot + prefix-sum
// compaction of linearizable successors into an exact open-addressed visited
// set).  Depth-first in spirit (the newest config's first successor ends up on
// top), so valid histories still finish without enumerating the whole config
// space, while a dead end is abandoned K configs at a time; the sequential
// schedule's heavy tail (one unlucky history taking 10^6 dependent steps)
// disappears.
//
// The schedule is deterministic and specified in oracle/wgl_beam.c, which
// tests/ compare bit-for-bit (verdict, failing op, witness, counters):
// lanes that produce one and the same new config in a round are found by their
// common table slot (equal keys probe in lockstep, so exactly one of them wins
// the CAS and the others lose it on that very slot) and the lowest lane keeps
// the config; everything else follows program order of one wavefront.
//
// Visited-set entry (MW = mask words): k0 = front+1 | state<<32, M[MW],
// {parent entry | op<<32}: 24 B at MW = 1.  The 16 most recent pushes are
// mirrored in an LDS ring so the next iteration's parents come from LDS.  A history is owned by
// one wavefront; entries are read/written with agent-scope (sc1) accesses so
// a lane never sees a stale L1 line of an entry another lane just claimed.
#include <hip/hip_runtime.h>
#include "tbc_internal.h"
#include "device_common.h"

namespace tbc {

namespace {

constexpr uint32_t kNone = 0xFFFFFFFFu;

__device__ __forceinline__ uint64_t ld64(const uint64_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st64(uint64_t* p, uint64_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t ld32(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// LDS words per wave: parents p_k0[16] p_M[16*MW] (u64) p_slot p_off p_nlive p_cnt (u32 x16),
// ring of the 16 most recent pushes r_k0[16] r_M[16*MW] (u64) r_pos r_idx r_off r_nlive r_cnt (u32 x16)
__host__ __device__ constexpr uint32_t beam_lds_words(uint32_t mw) { return 2 * (32 + 32 * mw) + 16 * 9 + 20; }

__device__ __forceinline__ uint32_t key_hash32(uint64_t k0, const uint64_t* M, int mw) {
  uint32_t h = (uint32_t)k0 * 0x9E3779B1u ^ (uint32_t)(k0 >> 32) * 0x85EBCA77u;
  for (int j = 0; j < mw; j++) {
    h = (h << 13) | (h >> 19);
    h ^= (uint32_t)M[j] * 0xC2B2AE3Du ^ (uint32_t)(M[j] >> 32) * 0x27D4EB2Fu;
  }
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
  return h;
}

// Cold path: move a history to a 4x larger visited set (and stack) taken from the batch's growth pool -- re-insert every entry, then
// translate the slot numbers held by parent links, the stack and this iteration's parents.
template <int MW>
__device__ __forceinline__ bool grow_visited_set(const BeamArgs& A, uint64_t* tab, uint32_t* stack, uint32_t cap_log2,
                                                           uint32_t sp, uint32_t np, uint32_t* p_slot, uint32_t* r_pos,
                                                           uint32_t lane, uint64_t** ntab_out, uint32_t** nstack_out) {
  constexpr uint32_t EW = MW + 2;
  const uint64_t old_cap = 1ull << cap_log2, new_cap = old_cap << 2;
  const uint64_t need = new_cap * EW + new_cap / 2 + old_cap / 2;      // table, stack, slot translation
  unsigned long long base = 0;
  if (lane == 0) base = (A.pool && cap_log2 + 2 <= A.max_tab_log2) ? atomicAdd(A.pool_cursor, (unsigned long long)need) : ~0ull;
  base = ru64(base);
  if (!A.pool || cap_log2 + 2 > 31 || cap_log2 + 2 > A.max_tab_log2 || base + need > A.pool_words) return false;
  uint64_t* ntab = A.pool + base;
  uint32_t* nstack = reinterpret_cast<uint32_t*>(ntab + new_cap * EW);
  uint32_t* remap = nstack + new_cap;
  const uint32_t nmask = (uint32_t)(new_cap - 1);
#pragma unroll 1
  for (uint64_t s = lane; s < old_cap; s += 64) {
    const uint64_t* e = tab + s * EW;
    const uint64_t k0 = ld64(e);
    if ((uint32_t)k0 == 0u) continue;
    uint64_t Mx[MW];
#pragma unroll
    for (int j = 0; j < MW; j++) Mx[j] = ld64(e + 1 + j);
    const uint64_t pw = ld64(e + 1 + MW);
    uint32_t idx = key_hash32(k0, Mx, MW) & nmask;
    for (;;) {
      uint64_t* ne = ntab + (uint64_t)idx * EW;
      if (atomicCAS((unsigned long long*)ne, 0ull, (unsigned long long)k0) == 0ull) {
#pragma unroll
        for (int j = 0; j < MW; j++) st64(ne + 1 + j, Mx[j]);
        st64(ne + 1 + MW, pw);
        break;
      }
      idx = (idx + 1u) & nmask;
    }
    __hip_atomic_store(remap + s, idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __threadfence();
#pragma unroll 1
  for (uint64_t s = lane; s < old_cap; s += 64) {       // parent links -> new slot numbers
    if ((uint32_t)ld64(tab + s * EW) == 0u) continue;
    uint64_t* ne = ntab + (uint64_t)ld32(remap + s) * EW + 1 + MW;
    const uint64_t pw = ld64(ne);
    if ((uint32_t)pw != kNone) st64(ne, (uint64_t)ld32(remap + (uint32_t)pw) | (pw & 0xFFFFFFFF00000000ull));
  }
#pragma unroll 1
  for (uint32_t i = lane; i < sp; i += 64)    // sp was already lowered by np: the popped run ...
    __hip_atomic_store(nstack + i, ld32(remap + ld32(stack + i)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (lane < np) p_slot[lane] = ld32(remap + p_slot[lane]);   // ... lives in p_slot
  if (lane < 16) r_pos[lane] = kNone;          // the ring held old slot numbers
  __threadfence();
  *ntab_out = ntab; *nstack_out = nstack;
  return true;
}

template <int MW, bool COMM>
__device__ void beam_one(const BeamArgs& A, const uint32_t hidx, uint32_t* lds, const uint32_t lane) {
  constexpr uint32_t EW = MW + 2;   // u64 words per entry

  const Hist* H = A.hist + hidx;
  const BeamHist* B = A.bh + hidx;
  const uint64_t op_off = ru64(H->op_off);
  const uint64_t ret_off = ru64(H->ret_off);
  const uint64_t off_off = ru64(B->off_off);
  const uint32_t* off = A.off + off_off;
  const uint32_t* ncr = A.ncr + off_off;
  const uint32_t* lst = A.lst + ru64(B->lst_off);
  const uint32_t* crashed = A.crashed + op_off;
  const OpInfo* opinfo = A.opinfo + op_off;
  const uint32_t* ret_slot = A.ret_slot + ret_off;
  uint32_t* stack = A.stack + ru64(B->stack_off);
  uint64_t* tab = A.tab + ru64(B->tab_off) * EW;   // both move when the visited set grows
  const uint32_t R = rfl(H->n_ret), status = rfl(H->status) | rfl(B->status);
  uint32_t cap_log2 = rfl(B->tab_log2);
  uint32_t cap_mask = (uint32_t)((1ull << cap_log2) - 1ull);
  uint32_t full_at = (uint32_t)((1ull << cap_log2) - (1ull << (cap_log2 - 2)));
  // K parents per iteration, G = 64 / K lanes (candidate slots) per parent per round.  A history that has
  // used more than round_budget rounds continues at K = 16: stragglers then need far fewer dependent rounds.
  uint32_t K = A.width, gshift = 6u - (31u - (uint32_t)__builtin_clz(K)), G = 1u << gshift;
  DevResult* out = A.results + hidx;
  Model model{A.model_kind, A.table, A.n_classes, A.pool_vals, (int32_t)rfl((uint32_t)H->aux), A.n_keys};

  uint64_t* p_k0 = reinterpret_cast<uint64_t*>(lds);
  uint64_t* p_M = p_k0 + 16;
  uint64_t* r_k0 = p_M + 16 * MW;
  uint64_t* r_M = r_k0 + 16;
  uint32_t* p_slot = reinterpret_cast<uint32_t*>(r_M + 16 * MW);
  uint32_t* p_off = p_slot + 16;
  uint32_t* p_nlive = p_off + 16;
  uint32_t* p_cnt = p_nlive + 16;
  uint32_t* r_pos = p_cnt + 16;      // stack position mirrored in this ring slot (kNone = empty)
  uint32_t* r_idx = r_pos + 16;
  uint32_t* r_off = r_idx + 16;
  uint32_t* r_nlive = r_off + 16;
  uint32_t* r_cnt = r_nlive + 16;
  uint32_t* p_start = r_cnt + 16;    // 17 entries: pair-number prefix, general mapping only
  if (lane < 16) r_pos[lane] = kNone;

  uint64_t probes = 0, visited = 0, expanded = 0, iterations = 0, rounds = 0;
  uint32_t sp = 0, max_sp = 0, lane_maxf = 0;
  int32_t verdict = -2, cause = TBC_CAUSE_NONE;
  uint32_t win_parent = kNone, win_op = kNone;
  int32_t win_state = A.init_state;
  const uint64_t t0 = A.time_limit_ticks ? wall_clock64() : 0;
#ifdef TBC_SEGPROF   // per-segment cycle counters (scripts/gpu_segprof.py); costs ~20 VGPRs, off in production
  const bool prof = A.dbg != nullptr;
  uint64_t seg[6] = {0, 0, 0, 0, 0, 0}, tlast = prof ? __builtin_readcyclecounter() : 0;
#define SEG(i) do { if (prof) { const uint64_t tn_ = __builtin_readcyclecounter(); seg[i] += tn_ - tlast; tlast = tn_; } } while (0)
#else
  constexpr bool prof = false;
#define SEG(i) do {} while (0)
#endif

  if (status != 0) verdict = TBC_UNKNOWN;
  else if (R == 0) verdict = TBC_VALID;
  else {
    // root config
    const uint64_t k0 = 1ull | ((uint64_t)(uint32_t)A.init_state << 32);
    uint64_t zero[MW];
#pragma unroll
    for (int j = 0; j < MW; j++) zero[j] = 0;
    const uint32_t idx = key_hash32(k0, zero, MW) & cap_mask;
    if (lane == 0) {
      uint64_t* e = tab + (uint64_t)idx * EW;
      st64(e + 0, k0);
#pragma unroll
      for (int j = 0; j < MW; j++) st64(e + 1 + j, 0ull);
      st64(e + 1 + MW, (uint64_t)kNone | ((uint64_t)kNone << 32));
      __hip_atomic_store(stack, idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    sp = 1; visited = 1; max_sp = 1;
  }

  while (verdict == -2) {
    if (sp == 0) { verdict = TBC_INVALID; break; }
    if (A.round_budget && rounds > A.round_budget && K < 16u) { K = 16u; gshift = 2u; G = 4u; }
    const uint32_t np = min(K, sp);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- pop the np most recent configs: lane l < np loads the l-th from the bottom of the popped run
    uint32_t my_cnt = 0;
    if (lane < np) {
      const uint32_t spos = sp - np + lane, rs = spos & 15u;
      if (r_pos[rs] == spos) {                  // pushed recently: config (and its front's list) still in the ring
        p_k0[lane] = r_k0[rs];
#pragma unroll
        for (int j = 0; j < MW; j++) p_M[lane * MW + j] = r_M[rs * MW + j];
        p_slot[lane] = r_idx[rs]; p_off[lane] = r_off[rs]; p_nlive[lane] = r_nlive[rs];
        my_cnt = r_cnt[rs];
      } else {
        const uint32_t idx = ld32(stack + spos);
        const uint64_t* e = tab + (uint64_t)idx * EW;
        const uint64_t k0 = ld64(e);
        p_k0[lane] = k0;
#pragma unroll
        for (int j = 0; j < MW; j++) p_M[lane * MW + j] = ld64(e + 1 + j);
        const uint32_t fi = (uint32_t)k0 - 1u;
        const uint32_t o0 = off[fi], o1 = off[fi + 1], nc = ncr[fi];
        p_slot[lane] = idx; p_off[lane] = o0; p_nlive[lane] = o1 - o0;
        my_cnt = (o1 - o0) + nc;
      }
      p_cnt[lane] = my_cnt;
    }
    sp -= np;
    uint32_t maxcnt = my_cnt;                   // lanes >= np hold 0
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) maxcnt = max(maxcnt, (uint32_t)__shfl_xor(maxcnt, d));
    maxcnt = rl(maxcnt, 0);
    // Pair order (oracle/wgl_beam.c): parents bottom-first, each parent's open calls last-to-first,
    // 64 consecutive pairs per round.  When every parent has at most G = 64/K open calls all pairs
    // fit one round and lane = parent*G + i realises that order directly; otherwise pairs are
    // numbered through a prefix sum over the parents (crash-heavy histories).
    const bool grouped = maxcnt <= G;
    uint32_t T = maxcnt ? 64u : 0u;
    if (!grouped) {
      uint32_t x = my_cnt;
#pragma unroll
      for (int d = 1; d < 16; d <<= 1) {
        const uint32_t y = __shfl_up(x, d);
        if (lane >= (uint32_t)d) x += y;
      }
      if (lane < np) p_start[lane] = x - my_cnt;
      T = rl(x, np - 1);
      if (lane == 0) p_start[np] = T;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- room for every pair of this iteration?  If not, move to a 4x larger visited set (and stack)
    // taken from the batch's growth pool: re-insert every entry, then translate the slot numbers held
    // by parent links, the stack and this iteration's parents.  Results do not depend on the layout.
    {
      const uint32_t worst = grouped ? np * G : T;
      bool failed = false;
      while (__builtin_expect(!failed && (uint64_t)visited + worst > full_at, 0)) {
        uint64_t* ntab = nullptr; uint32_t* nstack = nullptr;
        if (!grow_visited_set<MW>(A, tab, stack, cap_log2, sp, np, p_slot, r_pos, lane, &ntab, &nstack)) { failed = true; break; }
        const uint64_t new_cap = 1ull << (cap_log2 + 2);
        const uint32_t nmask = (uint32_t)(new_cap - 1);
        tab = ntab; stack = nstack; cap_log2 += 2; cap_mask = nmask;
        full_at = (uint32_t)(new_cap - (new_cap >> 2));
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
      if (failed) { sp += np; verdict = TBC_UNKNOWN; cause = TBC_CAUSE_VISITED_FULL; break; }
    }
    iterations++; expanded += np;
    SEG(0);

    for (uint32_t base = 0; base < T && verdict == -2; base += 64) {
      // ---- which (parent, open call) pair this lane handles
      uint32_t q, cd;
      bool has_parent;
      if (grouped) {
        q = lane >> gshift; cd = lane & (G - 1u); has_parent = q < np;
      } else {
        const uint32_t r = base + lane;
        q = 0;
#pragma unroll
        for (uint32_t t = 1; t < 16; t++) q += (t < np && p_start[t] <= r) ? 1u : 0u;
        has_parent = r < T;
        cd = has_parent ? r - p_start[q] : 0u;
      }
      const uint64_t k0p = has_parent ? p_k0[q] : 1ull;
      const uint32_t fi = (uint32_t)k0p - 1u;
      const int32_t st = (int32_t)(uint32_t)(k0p >> 32);
      uint64_t Mp[MW];
#pragma unroll
      for (int j = 0; j < MW; j++) Mp[j] = has_parent ? p_M[q * MW + j] : 0ull;
      const uint32_t pslot = has_parent ? p_slot[q] : 0u, poff = has_parent ? p_off[q] : 0u;
      const uint32_t nlive = has_parent ? p_nlive[q] : 0u, cnt = has_parent ? p_cnt[q] : 0u;
      const bool act = has_parent && cd < cnt;
      const uint32_t next_slot = (act && fi + 1u < R) ? ret_slot[fi + 1u] : 0u;   // first step of a front advance
      const uint32_t c = cnt - 1u - cd;
      uint32_t op = 0;
      OpInfo oi; oi.ret_rank = 0; oi.f_slot = kFNone; oi.a = 0; oi.b = 0;
      if (act) { op = c < nlive ? lst[poff + c] : crashed[c - nlive]; oi = opinfo[op]; }
      const uint32_t p = oi.f_slot >> 8;
      if (prof) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); SEG(1); }
      bool lin = false;
#pragma unroll
      for (int j = 0; j < MW; j++) if ((p >> 6) == (uint32_t)j) lin = (Mp[j] >> (p & 63u)) & 1ull;
      const bool viable = act && !lin && pair_viable<MW, COMM>(model, st, fi, Mp, poff, nlive, cnt, lst, crashed, opinfo, oi);
      int32_t st2 = st;
      uint32_t fi2 = fi;
      uint64_t M2[MW];
      make_child<MW, COMM>(model, viable, st, fi, R, ret_slot, next_slot, oi, Mp, M2, st2, fi2);
      SEG(2);
      rounds++;
      const uint64_t succ = __ballot(viable && fi2 == R);
      if (succ) {   // linearizable: lowest pair wins, nothing of this round is inserted
        const uint32_t wl = (uint32_t)__builtin_ctzll(succ);
        win_parent = rl(pslot, wl); win_op = rl(op, wl); win_state = (int32_t)rl((uint32_t)st2, wl);
        verdict = TBC_VALID;
        break;
      }
      const uint64_t vb = __ballot(viable);
      probes += (uint64_t)__popcll(vb);

      // the child's front: its open-call list (issued now, consumed at push; hidden under the probe)
      uint32_t co0 = 0, co1 = 0, cnc = 0;
      if (viable) { co0 = off[fi2]; co1 = off[fi2 + 1u]; cnc = ncr[fi2]; }

      // ---- visited set: claim-or-find with one CAS per probe step.  Equal keys probe in lockstep.
      const uint64_t k0 = (uint64_t)(fi2 + 1u) | ((uint64_t)(uint32_t)st2 << 32);
      uint32_t idx = key_hash32(k0, M2, MW) & cap_mask;
      bool pending = viable, fresh = false, won = false, lost = false;
      while (__ballot(pending)) {
        if (pending) {
          uint64_t* e = tab + (uint64_t)idx * EW;
          const uint64_t k0e = ld64(e);
          if ((uint32_t)k0e == 0u) {
            const uint64_t old = atomicCAS((unsigned long long*)e, 0ull, (unsigned long long)k0);
            if (old == 0ull) {                       // claimed an empty entry
#pragma unroll
              for (int j = 0; j < MW; j++) st64(e + 1 + j, M2[j]);
              won = true; fresh = true; pending = false;
            } else {
              lost = true;     // claimed by another lane in this very step: look at the same entry again
            }
          } else {
            bool same = k0e == k0;
#pragma unroll
            for (int j = 0; j < MW; j++) same = same && ld64(e + 1 + j) == M2[j];
            if (same) {
              fresh = lost;    // equal keys probe in lockstep: a lost CAS followed by a match on that slot
              pending = false; //   means the config was inserted in THIS round by a sibling lane
            } else {
              lost = false;
              idx = (idx + 1u) & cap_mask;
            }
          }
        }
      }
      SEG(3);
      // the lowest lane among the lanes that produced one and the same new config keeps it
      bool is_new = fresh;
      uint64_t dupl = __ballot(fresh && !won);
      while (dupl) {
        const uint32_t l0 = (uint32_t)__builtin_ctzll(dupl);
        const uint32_t i0 = rl(idx, l0);
        const uint64_t grp = __ballot(fresh && idx == i0);
        const uint32_t winner = (uint32_t)__builtin_ctzll(grp);
        if ((grp >> lane) & 1ull) is_new = lane == winner;
        dupl &= ~grp;
      }
      if (is_new) st64(tab + (uint64_t)idx * EW + 1 + MW, (uint64_t)pslot | ((uint64_t)(op + 1u) << 32));
      const uint64_t nb = __ballot(is_new);
      const uint32_t nn = (uint32_t)__popcll(nb);
      if (is_new) {
        const uint32_t pos = sp + (uint32_t)__popcll(nb & ((1ull << lane) - 1ull));
        __hip_atomic_store(stack + pos, idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (pos + 16u >= sp + nn) {               // one of the 16 topmost: mirror it in the LDS ring
          const uint32_t rs = pos & 15u;
          r_pos[rs] = pos; r_idx[rs] = idx; r_k0[rs] = k0;
#pragma unroll
          for (int j = 0; j < MW; j++) r_M[rs * MW + j] = M2[j];
          r_off[rs] = co0; r_nlive[rs] = co1 - co0; r_cnt[rs] = (co1 - co0) + cnc;
        }
        lane_maxf = max(lane_maxf, fi2);
      }
      sp += nn; visited += nn;
      SEG(4);
    }
    max_sp = max(max_sp, sp);
    if (A.dbg && lane == 0 && (iterations & 255u) == 1u) {
      A.dbg[8] = hidx; A.dbg[9] = (uint32_t)iterations; A.dbg[10] = sp; A.dbg[11] = (uint32_t)probes;
      A.dbg[12] = (uint32_t)visited; A.dbg[13] = maxcnt; A.dbg[14] = np; A.dbg[15] = (uint32_t)rounds;
    }
    if (verdict == -2) {
      if (A.max_steps && probes > A.max_steps) { verdict = TBC_UNKNOWN; cause = TBC_CAUSE_STEP_LIMIT; }
      else if (A.time_limit_ticks && (iterations & 63u) == 0 && (uint64_t)wall_clock64() - t0 > A.time_limit_ticks) {
        verdict = TBC_UNKNOWN; cause = TBC_CAUSE_TIME_LIMIT;
      }
    }
  }

  // ---- results
  uint32_t maxf = lane_maxf;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) maxf = max(maxf, (uint32_t)__shfl_xor(maxf, d));
  maxf = rfl(maxf);
  // ---- invalid: the configs stuck at the failing completion (knossos :configs), by a scan of the visited set
  uint32_t n_cfg = 0;
  if (verdict == TBC_INVALID && A.cfg) {
    uint64_t* cfg = A.cfg + (uint64_t)hidx * kCfgCap * (2 + MW);
    const uint64_t ncap = 1ull << cap_log2;
    for (uint64_t s0 = 0; s0 < ncap; s0 += 64) {
      const uint64_t* e = tab + (s0 + lane) * EW;
      const uint64_t k0 = ld64(e);
      const bool hit = (uint32_t)k0 == maxf + 1u;
      const uint64_t hb = __ballot(hit);
      if (hit) {
        const uint32_t pos = n_cfg + (uint32_t)__popcll(hb & ((1ull << lane) - 1ull));
        if (pos < kCfgCap) {
          uint64_t* o = cfg + (uint64_t)pos * (2 + MW);
          o[0] = k0;
#pragma unroll
          for (int j = 0; j < MW; j++) o[1 + j] = ld64(e + 1 + j);
          const uint64_t pw = ld64(e + 1 + MW);
          o[1 + MW] = (uint32_t)pw == kNone ? (uint64_t)TBC_NO_OP : (pw >> 32) - 1ull;
        }
      }
      n_cfg += (uint32_t)__popcll(hb);
    }
  }
  uint32_t wlen = 0;
  if (verdict == TBC_VALID && R != 0) {
    // witness = ops along the parent chain of the winning config, then the winning op
    wlen = 1;
    uint32_t id = win_parent;
    for (;;) {
      const uint32_t par = (uint32_t)ld64(tab + (uint64_t)id * EW + 1 + MW);
      if (par == kNone) break;
      wlen++; id = par;
    }
    if (A.witness) {
      uint32_t* wit = A.witness + op_off;
      uint32_t w = wlen - 1;
      if (lane == 0) wit[w] = win_op;
      id = win_parent;
      for (;;) {
        const uint64_t po = ld64(tab + (uint64_t)id * EW + 1 + MW);
        const uint32_t par = (uint32_t)po;
        if (par == kNone) break;
        w--;
        if (lane == 0) wit[w] = (uint32_t)(po >> 32) - 1u;
        id = par;
      }
    }
  }
  if (lane == 0) {
    out->valid = verdict; out->cause = cause; out->max_front = maxf; out->depth = wlen;
    out->final_state = win_state; out->n_configs = n_cfg;
    out->fail_op = TBC_NO_OP; out->prev_ok_op = TBC_NO_OP;
    if (verdict == TBC_INVALID) {
      const uint32_t* ret_op = A.ret_op + ret_off;
      out->fail_op = ret_op[maxf];
      if (maxf) out->prev_ok_op = ret_op[maxf - 1];
    }
    out->steps = probes; out->visited = visited; out->probes = probes; out->backtracks = expanded;
    out->max_depth = max_sp; out->bucket_reads = rounds; out->tab_log2 = cap_log2;
  }
  if (A.dbg && lane == 0) {
    A.dbg[4] = 0x300u + hidx; A.dbg[16] = (uint32_t)verdict; A.dbg[17] = (uint32_t)iterations;
#ifdef TBC_SEGPROF
    for (int i = 0; i < 5; i++) { A.dbg[20 + 2 * i] = (uint32_t)seg[i]; A.dbg[21 + 2 * i] = (uint32_t)(seg[i] >> 32); }
#endif
    A.dbg[30] = (uint32_t)rounds;
  }
#undef SEG
}

// Register budget (MW = 1, state-carrying models): 94 VGPRs -> 5 wavefronts per SIMD.  Forcing 6 with a
// min-blocks launch bound spills in the round loop and measures slower; without the growth path the loop
// needs 74, and that build is only 3-5 % faster (profiles/r01_vgpr_ab.txt), so growth stays inline.
template <int MW, bool COMM>
__global__ __launch_bounds__(kBlock) void wgl_beam_kernel(BeamArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const uint32_t lane = threadIdx.x & 63u, wv = rfl(threadIdx.x >> 6);
  const uint32_t w = blockIdx.x * kWavesPerBlock + wv;
  if (w < A.n_work) beam_one<MW, COMM>(A, rfl(A.work[w]), lds + wv * beam_lds_words(MW), lane);
}

template <int MW>
void launch_beam_mw(const BeamArgs& a, uint32_t n_blocks, hipStream_t s) {
  const size_t lds = (size_t)kWavesPerBlock * beam_lds_words(MW) * 4;
  // the commutative (set / bank) models get their own instantiation: their evaluation code would
  // otherwise double the register budget of the register-family kernel
  if (a.model_kind == TBC_MODEL_SET || a.model_kind == TBC_MODEL_BANK)
    hipLaunchKernelGGL((wgl_beam_kernel<MW, true>), dim3(n_blocks), dim3(kBlock), lds, s, a);
  else
    hipLaunchKernelGGL((wgl_beam_kernel<MW, false>), dim3(n_blocks), dim3(kBlock), lds, s, a);
}

}  // namespace

bool launch_beam(const BeamArgs& a, uint32_t mask_words, uint32_t n_blocks, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  switch (mask_words) {
    case 1: launch_beam_mw<1>(a, n_blocks, s); return true;
    case 2: launch_beam_mw<2>(a, n_blocks, s); return true;
    case 4: launch_beam_mw<4>(a, n_blocks, s); return true;
    default: return false;
  }
}

}  // namespace tbc
