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

constexpr uint32_t kMaxWidth = 16;
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

// LDS words per wave: p_k0[16] u64, p_M[16*MW] u64, then 5 x 16 u32 + start[17]
// + ring of the 16 most recent pushes: pos[16], idx[16], k0[16] u64, M[16*MW] u64
__host__ __device__ constexpr uint32_t beam_lds_words(uint32_t mw) { return 2 * 16 + 2 * 16 * mw + 16 * 5 + 20 + 32 + 2 * 16 + 2 * 16 * mw; }

template <int MW>
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
  uint64_t* tab = A.tab + ru64(B->tab_off) * EW;
  const uint32_t R = rfl(H->n_ret), status = rfl(H->status) | rfl(B->status);
  const uint64_t cap = 1ull << rfl(B->tab_log2);
  const uint64_t cap_mask = cap - 1;
  const uint64_t full_at = cap - (cap >> 2);
  const uint32_t K = min(max(A.width, 1u), kMaxWidth);
  DevResult* out = A.results + hidx;
  Model model{A.model_kind, A.table, A.n_classes};

  uint64_t* p_k0 = reinterpret_cast<uint64_t*>(lds);
  uint64_t* p_M = p_k0 + 16;
  uint32_t* p_slot = reinterpret_cast<uint32_t*>(p_M + 16 * MW);
  uint32_t* p_cnt = p_slot + 16;
  uint32_t* p_off = p_cnt + 16;
  uint32_t* p_nlive = p_off + 16;
  uint32_t* p_fi = p_nlive + 16;
  uint32_t* p_start = p_fi + 16;     // 17 entries (20 reserved)
  uint32_t* r_pos = p_start + 20;    // ring: stack position mirrored in this slot (kNone = empty)
  uint32_t* r_idx = r_pos + 16;
  uint64_t* r_k0 = reinterpret_cast<uint64_t*>(r_idx + 16);
  uint64_t* r_M = r_k0 + 16;
  if (lane < 16) r_pos[lane] = kNone;

  uint64_t probes = 0, visited = 0, expanded = 0, iterations = 0, rounds = 0, max_stack = 0;
  uint32_t sp = 0, lane_maxf = 0;
  int32_t verdict = -2, cause = TBC_CAUSE_NONE;
  uint32_t win_parent = kNone, win_op = kNone;
  int32_t win_state = A.init_state;
  const uint64_t t0 = A.time_limit_ticks ? wall_clock64() : 0;

  if (status != 0) verdict = TBC_UNKNOWN;
  else if (R == 0) verdict = TBC_VALID;
  else {
    // root config
    const uint64_t k0 = 1ull | ((uint64_t)(uint32_t)A.init_state << 32);
    uint64_t h = mix64(k0);
#pragma unroll
    for (int j = 0; j < MW; j++) h = mix64(h ^ 0ull) + 0x9E3779B97F4A7C15ull;
    const uint64_t idx = h & cap_mask;
    if (lane == 0) {
      uint64_t* e = tab + idx * EW;
      st64(e + 0, k0);
#pragma unroll
      for (int j = 0; j < MW; j++) st64(e + 1 + j, 0ull);
      st64(e + 1 + MW, (uint64_t)kNone | ((uint64_t)kNone << 32));
      stack[0] = (uint32_t)idx;
    }
    sp = 1; visited = 1; max_stack = 1;
  }

  while (verdict == -2) {
    if (sp == 0) { verdict = TBC_INVALID; break; }
    const uint32_t np = min(K, sp);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- pop the np most recent configs (lane q = q-th from the bottom of the popped run)
    uint32_t my_cnt = 0;
    if (lane < np) {
      const uint32_t spos = sp - np + lane;
      uint32_t idx; uint64_t k0;
      if (r_pos[spos & 15u] == spos) {          // pushed recently: config still in the LDS ring
        idx = r_idx[spos & 15u]; k0 = r_k0[spos & 15u];
#pragma unroll
        for (int j = 0; j < MW; j++) p_M[lane * MW + j] = r_M[(spos & 15u) * MW + j];
      } else {
        idx = ld32(stack + spos);
        const uint64_t* e = tab + (uint64_t)idx * EW;
        k0 = ld64(e);
#pragma unroll
        for (int j = 0; j < MW; j++) p_M[lane * MW + j] = ld64(e + 1 + j);
      }
      p_k0[lane] = k0;
      const uint32_t fi = (uint32_t)k0 - 1u;
      const uint32_t o0 = off[fi], o1 = off[fi + 1], nc = ncr[fi];
      p_slot[lane] = idx; p_fi[lane] = fi; p_off[lane] = o0; p_nlive[lane] = o1 - o0;
      my_cnt = (o1 - o0) + nc;
      p_cnt[lane] = my_cnt;
    }
    sp -= np;
    // inclusive scan of the pair counts over lanes 0..15
    uint32_t x = my_cnt;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
      const uint32_t y = __shfl_up(x, d);
      if (lane >= (uint32_t)d) x += y;
    }
    if (lane < np) p_start[lane] = x - my_cnt;
    const uint32_t T = rl(x, np - 1);
    if (lane == 0) p_start[np] = T;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    iterations++; expanded += np;

    for (uint32_t base = 0; base < T && verdict == -2; base += 64) {
      const uint32_t r = base + lane;
      const bool act = r < T;
      uint32_t q = 0;
#pragma unroll
      for (uint32_t t = 1; t < kMaxWidth; t++) q += (t < np && p_start[t] <= r) ? 1u : 0u;
      const uint32_t fi = p_fi[q];
      const uint64_t k0p = p_k0[q];
      const int32_t st = (int32_t)(uint32_t)(k0p >> 32);
      uint64_t M2[MW];
#pragma unroll
      for (int j = 0; j < MW; j++) M2[j] = p_M[q * MW + j];
      const uint32_t nlive = p_nlive[q];
      const uint32_t c = act ? (p_cnt[q] - 1u - (r - p_start[q])) : 0u;
      uint32_t op = 0;
      if (act) op = c < nlive ? lst[p_off[q] + c] : crashed[c - nlive];
      OpInfo oi; oi.ret_rank = 0; oi.f_slot = kFNone; oi.a = 0; oi.b = 0;
      if (act) oi = opinfo[op];
      const uint32_t next_slot = (act && fi + 1u < R) ? ret_slot[fi + 1u] : 0u;   // speculative: first step of a front advance
      const uint32_t f = oi.f_slot & 0xFFu, p = oi.f_slot >> 8;
      bool lin = false;
#pragma unroll
      for (int j = 0; j < MW; j++) if ((p >> 6) == (uint32_t)j) lin = (M2[j] >> (p & 63u)) & 1ull;
      const bool viable = act && !lin && model.ok(st, f, oi.a);
      int32_t st2 = st;
      uint32_t fi2 = fi;
      if (viable) {
        st2 = model.apply(st, f, oi.a, oi.b);
#pragma unroll
        for (int j = 0; j < MW; j++) if ((p >> 6) == (uint32_t)j) M2[j] |= 1ull << (p & 63u);
        if (oi.ret_rank == fi) {   // the front's own call: the front moves past every completion already linearized
          uint32_t pp = p;
          for (;;) {
#pragma unroll
            for (int j = 0; j < MW; j++) if ((pp >> 6) == (uint32_t)j) M2[j] &= ~(1ull << (pp & 63u));
            fi2++;
            if (fi2 == R) break;
            pp = (fi2 == fi + 1u) ? next_slot : ret_slot[fi2];
            bool bit = false;
#pragma unroll
            for (int j = 0; j < MW; j++) if ((pp >> 6) == (uint32_t)j) bit = (M2[j] >> (pp & 63u)) & 1ull;
            if (!bit) break;
          }
        }
      }
      rounds++;
      const uint64_t succ = __ballot(viable && fi2 == R);
      if (succ) {   // linearizable: lowest pair wins, nothing of this round is inserted
        const uint32_t wl = (uint32_t)__builtin_ctzll(succ);
        win_parent = rl(p_slot[q], wl); win_op = rl(op, wl); win_state = (int32_t)rl((uint32_t)st2, wl);
        verdict = TBC_VALID;
        break;
      }
      const uint64_t vb = __ballot(viable);
      probes += (uint64_t)__popcll(vb);
      if (visited + 64 > full_at) { verdict = TBC_UNKNOWN; cause = TBC_CAUSE_VISITED_FULL; break; }
      // ---- visited set: lookup / claim.  All lanes stay in the loop until every lane is done.
      const uint64_t k0 = (uint64_t)(fi2 + 1u) | ((uint64_t)(uint32_t)st2 << 32);
      uint64_t hsh = mix64(k0);
#pragma unroll
      for (int j = 0; j < MW; j++) hsh = mix64(hsh ^ M2[j]) + 0x9E3779B97F4A7C15ull;
      uint64_t idx = hsh & cap_mask;
      bool pending = viable, fresh = false, lost = false;
      while (__ballot(pending)) {
        if (pending) {
          uint64_t* e = tab + idx * EW;
          const uint64_t k0e = ld64(e);
          if ((uint32_t)k0e == 0u) {
            const uint64_t old = atomicCAS((unsigned long long*)e, 0ull, (unsigned long long)k0);
            if (old == 0ull) {
#pragma unroll
              for (int j = 0; j < MW; j++) st64(e + 1 + j, M2[j]);
              fresh = true; pending = false;
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
              idx = (idx + 1) & cap_mask;
            }
          }
        }
      }
      // the lowest lane among the lanes that produced one and the same new config keeps it
      bool is_new = fresh;
      uint64_t dupl = __ballot(fresh && lost);
      while (dupl) {
        const uint32_t l0 = (uint32_t)__builtin_ctzll(dupl);
        const uint32_t ilo = rl((uint32_t)idx, l0), ihi = rl((uint32_t)(idx >> 32), l0);
        const uint64_t grp = __ballot(fresh && (uint32_t)idx == ilo && (uint32_t)(idx >> 32) == ihi);
        const uint32_t winner = (uint32_t)__builtin_ctzll(grp);
        if ((grp >> lane) & 1ull) is_new = lane == winner;
        dupl &= ~grp;
      }
      if (is_new) st64(tab + idx * EW + 1 + MW, (uint64_t)p_slot[q] | ((uint64_t)op << 32));
      const uint64_t nb = __ballot(is_new);
      const uint32_t nn = (uint32_t)__popcll(nb);
      if (is_new) {
        const uint32_t pos = sp + (uint32_t)__popcll(nb & ((1ull << lane) - 1ull));
        __hip_atomic_store(stack + pos, (uint32_t)idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (pos + 16u >= sp + nn) {               // one of the 16 topmost: mirror it in the LDS ring
          r_pos[pos & 15u] = pos; r_idx[pos & 15u] = (uint32_t)idx; r_k0[pos & 15u] = k0;
#pragma unroll
          for (int j = 0; j < MW; j++) r_M[(pos & 15u) * MW + j] = M2[j];
        }
        lane_maxf = max(lane_maxf, fi2);
      }
      sp += nn; visited += nn;
    }
    max_stack = max(max_stack, (uint64_t)sp);
    if (A.dbg && lane == 0 && (iterations & 255u) == 1u) {
      A.dbg[8] = hidx; A.dbg[9] = (uint32_t)iterations; A.dbg[10] = sp; A.dbg[11] = (uint32_t)probes;
      A.dbg[12] = (uint32_t)visited; A.dbg[13] = T; A.dbg[14] = np; A.dbg[15] = (uint32_t)rounds;
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
        const uint32_t opx = (uint32_t)(po >> 32);
        w--;
        if (lane == 0) wit[w] = opx;
        id = par;
      }
    }
  }
  if (lane == 0) {
    out->valid = verdict; out->cause = cause; out->max_front = maxf; out->depth = wlen;
    out->final_state = win_state; out->n_configs = 0;
    out->fail_op = TBC_NO_OP; out->prev_ok_op = TBC_NO_OP;
    if (verdict == TBC_INVALID) {
      const uint32_t* ret_op = A.ret_op + ret_off;
      out->fail_op = ret_op[maxf];
      if (maxf) out->prev_ok_op = ret_op[maxf - 1];
    }
    out->steps = probes; out->visited = visited; out->probes = probes; out->backtracks = expanded;
    out->max_depth = max_stack; out->bucket_reads = rounds;
  }
  if (A.dbg && lane == 0) { A.dbg[4] = 0x300u + hidx; A.dbg[16] = (uint32_t)verdict; A.dbg[17] = (uint32_t)iterations; }
}

template <int MW>
__global__ __launch_bounds__(kBlock) void wgl_beam_kernel(BeamArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const uint32_t lane = threadIdx.x & 63u, wv = rfl(threadIdx.x >> 6);
  const uint32_t w = blockIdx.x * kWavesPerBlock + wv;
  if (w < A.n_work) beam_one<MW>(A, rfl(A.work[w]), lds + wv * beam_lds_words(MW), lane);
}

template <int MW>
void launch_beam_mw(const BeamArgs& a, uint32_t n_blocks, hipStream_t s) {
  const size_t lds = (size_t)kWavesPerBlock * beam_lds_words(MW) * 4;
  hipLaunchKernelGGL(wgl_beam_kernel<MW>, dim3(n_blocks), dim3(kBlock), lds, s, a);
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
