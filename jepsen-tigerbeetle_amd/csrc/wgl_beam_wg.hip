// wgl_beam_wg.hip -- K5b: the wide schedule with one WORKGROUP (4 wavefronts, 256 lanes) per
// history, for the histories a single wavefront is too slow for (stragglers of a batch, and any
// caller who wants the shortest time-to-verdict): K = 32 or 64 configs come off the stack per
// iteration and 256 (config, open call) pairs are expanded per round, so a hard history needs
// 4-6x fewer dependent rounds than at K = 4 on one wavefront.
//
// Same deterministic schedule as wgl_beam.hip -- oracle/wgl_beam.c with round_pairs = 256 -- but the
// lanes that meet on one table slot may now sit in different wavefronts, so the in-round duplicate
// rule ("the lowest pair number keeps the config") is enforced with memory operations instead of
// lockstep:
//   * an entry is claimed with CAS(k0, 0, key | BUSY); the claimer writes the mask words and its
//     owner tag, releases (agent scope) and only then publishes the key without BUSY; readers that
//     see BUSY re-read;
//   * every lane that produced the config does atomicMax on the entry's owner tag
//     (round number << 8 | 255 - lane-in-round): after the round's barrier the tag names the lowest
//     pair, which records parent and op and pushes the config;
//   * per-wavefront ballots + a 4-entry LDS prefix give the push positions in pair order.
//
// Entry (MW mask words): k0 = front+1 | BUSY<<31 | state<<32, M[MW], {owner tag | parent<<32}, {op+1}.
#include <hip/hip_runtime.h>
#include "tbc_internal.h"
#include "device_common.h"

namespace tbc {

namespace {

constexpr uint32_t kNT = 256;           // lanes (pairs per round) per history
constexpr uint32_t kNWV = kNT / 64;
constexpr uint32_t kMaxK = 64;
constexpr uint32_t kNoneW = 0xFFFFFFFFu;
constexpr uint64_t kBusy = 1ull << 31;

__device__ __forceinline__ uint64_t ld64w(const uint64_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st64w(uint64_t* p, uint64_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t ld32w(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ uint32_t key_hash32w(uint64_t k0, const uint64_t* M, int mw) {
  uint32_t h = (uint32_t)k0 * 0x9E3779B1u ^ (uint32_t)(k0 >> 32) * 0x85EBCA77u;
  for (int j = 0; j < mw; j++) {
    h = (h << 13) | (h >> 19);
    h ^= (uint32_t)M[j] * 0xC2B2AE3Du ^ (uint32_t)(M[j] >> 32) * 0x27D4EB2Fu;
  }
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
  return h;
}

template <int MW, bool COMM>
__global__ __launch_bounds__(kNT) void wgl_beam_wg_kernel(BeamArgs A) {
  constexpr uint32_t EW = MW + 3;
  __shared__ uint64_t p_k0[kMaxK];
  __shared__ uint64_t p_M[kMaxK * MW];
  __shared__ uint32_t p_slot[kMaxK], p_off[kMaxK], p_nlive[kMaxK], p_cnt[kMaxK], p_start[kMaxK + 1];
  __shared__ uint32_t s_T, s_win, s_winfo[3], s_wvia[kNWV], s_wnew[kNWV], s_abort;

  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (blockIdx.x >= A.n_work) return;
  const uint32_t hidx = A.work[blockIdx.x];
  const Hist* H = A.hist + hidx;
  const BeamHist* B = A.bh + hidx;
  const uint64_t op_off = H->op_off, ret_off = H->ret_off, off_off = B->off_off;
  const uint32_t* off = A.off + off_off;
  const uint32_t* ncr = A.ncr + off_off;
  const OpRec* lst = A.lst + B->lst_off;
  const OpRec* crashed = A.crashed + op_off;
  const uint32_t* ret_slot = A.ret_slot + ret_off;
  uint32_t* stack = A.stack + B->stack_off;
  uint64_t* tab = A.tab + B->tab_off * EW;
  const uint32_t R = H->n_ret, status = H->status | B->status;
  const uint32_t cap_log2 = B->tab_log2;
  const uint32_t cap_mask = (uint32_t)((1ull << cap_log2) - 1ull);
  const uint32_t full_at = (uint32_t)((1ull << cap_log2) - (1ull << (cap_log2 - 2)));
  const uint32_t K = min(A.width, kMaxK);
  DevResult* out = A.results + hidx;
  Model model{A.model_kind, A.table, A.n_classes, A.pool_vals, H->aux, A.n_keys};

  uint64_t probes = 0, visited = 0, expanded = 0, iterations = 0, rounds = 0;
  uint32_t sp = 0, max_sp = 0, my_maxf = 0, round_no = 0;
  int32_t verdict = -2, cause = TBC_CAUSE_NONE;
  uint32_t win_parent = kNoneW, win_op = kNoneW;
  int32_t win_state = A.init_state;
  const uint64_t t0 = A.time_limit_ticks ? wall_clock64() : 0;

  if (tid == 0) { s_win = kNoneW; s_abort = 0; }
  if (status != 0) verdict = TBC_UNKNOWN;
  else if (R == 0) verdict = TBC_VALID;
  else {
    const uint64_t k0 = 1ull | ((uint64_t)(uint32_t)A.init_state << 32);
    uint64_t zero[MW];
#pragma unroll
    for (int j = 0; j < MW; j++) zero[j] = 0;
    const uint32_t idx = key_hash32w(k0, zero, MW) & cap_mask;
    if (tid == 0) {
      uint64_t* e = tab + (uint64_t)idx * EW;
      st64w(e, k0);
#pragma unroll
      for (int j = 0; j < MW; j++) st64w(e + 1 + j, 0ull);
      st64w(e + 1 + MW, (uint64_t)0u | ((uint64_t)kNoneW << 32));
      st64w(e + 2 + MW, 0ull);
      __hip_atomic_store(stack, idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    sp = 1; visited = 1; max_sp = 1;
  }
  __threadfence();
  __syncthreads();

  while (verdict == -2) {
    if (sp == 0) { verdict = TBC_INVALID; break; }
    const uint32_t np = min(K, sp);
    // ---- pop: thread l < np loads the l-th config from the bottom of the popped run
    if (tid < np) {
      const uint32_t idx = ld32w(stack + (sp - np + tid));
      const uint64_t* e = tab + (uint64_t)idx * EW;
      const uint64_t k0 = ld64w(e);
      p_k0[tid] = k0;
#pragma unroll
      for (int j = 0; j < MW; j++) p_M[tid * MW + j] = ld64w(e + 1 + j);
      const uint32_t fi = (uint32_t)k0 - 1u;
      const uint32_t o0 = off[fi], o1 = off[fi + 1], nc = ncr[fi];
      p_slot[tid] = idx; p_off[tid] = o0; p_nlive[tid] = o1 - o0; p_cnt[tid] = (o1 - o0) + nc;
    }
    __syncthreads();
    if (wave == 0) {   // pair-number prefix over the <= 64 parents
      uint32_t x = lane < np ? p_cnt[lane] : 0u;
      const uint32_t mine = x;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(x, d);
        if (lane >= (uint32_t)d) x += y;
      }
      if (lane < np) p_start[lane] = x - mine;
      if (lane == np - 1) { p_start[np] = x; s_T = x; }
    }
    __syncthreads();
    const uint32_t T = s_T;
    sp -= np;
    iterations++; expanded += np;
    if (visited + T > full_at) { sp += np; verdict = TBC_UNKNOWN; cause = TBC_CAUSE_VISITED_FULL; break; }

    for (uint32_t base = 0; base < T && verdict == -2; base += kNT) {
      round_no++;
      rounds++;
      const uint32_t r = base + tid;
      const bool has = r < T;
      uint32_t q = 0;
#pragma unroll
      for (uint32_t s = 32; s >= 1; s >>= 1) if (q + s < np && p_start[q + s] <= r) q += s;
      const uint64_t k0p = has ? p_k0[q] : 1ull;
      const uint32_t fi = (uint32_t)k0p - 1u;
      const int32_t st = (int32_t)(uint32_t)(k0p >> 32);
      uint64_t Mp[MW];
#pragma unroll
      for (int j = 0; j < MW; j++) Mp[j] = has ? p_M[q * MW + j] : 0ull;
      const uint32_t pslot = has ? p_slot[q] : 0u, poff = has ? p_off[q] : 0u;
      const uint32_t nlive = has ? p_nlive[q] : 0u, cnt = has ? p_cnt[q] : 0u;
      const uint32_t cd = has ? r - p_start[q] : 0u;
      const bool act = has && cd < cnt;
      const uint32_t next_slot = (act && fi + 1u < R) ? ret_slot[fi + 1u] : 0u;
      const uint32_t c = cnt - 1u - cd;
      OpRec oi; oi.op = 0; oi.f_slot = kFNone; oi.a = 0; oi.b = 0;
      if (act) oi = c < nlive ? lst[poff + c] : crashed[c - nlive];
      const uint32_t op = oi.op;
      const uint32_t p = (oi.f_slot >> 8) & kSlotMask;
      bool lin = false;
#pragma unroll
      for (int j = 0; j < MW; j++) if ((p >> 6) == (uint32_t)j) lin = (Mp[j] >> (p & 63u)) & 1ull;
      const bool viable = act && !lin && pair_viable<MW, COMM>(model, st, fi, Mp, poff, nlive, cnt, lst, crashed, oi);
      int32_t st2; uint32_t fi2; uint64_t M2[MW];
      make_child<MW, COMM>(model, viable, st, fi, R,
                           [=](uint32_t rk) -> uint32_t { return rk == fi + 1u ? next_slot : ret_slot[rk]; },
                           oi, Mp, M2, st2, fi2);

      if (viable && fi2 == R) atomicMin(&s_win, r);
      const uint64_t vb = __ballot(viable);
      if (lane == 0) s_wvia[wave] = (uint32_t)__popcll(vb);
      __syncthreads();
      if (s_win != kNoneW) {   // linearizable: the lowest pair wins, nothing of this round is inserted
        if (r == s_win) { s_winfo[0] = pslot; s_winfo[1] = op; s_winfo[2] = (uint32_t)st2; }
        __syncthreads();
        win_parent = s_winfo[0]; win_op = s_winfo[1]; win_state = (int32_t)s_winfo[2];
        verdict = TBC_VALID;
        break;
      }
#pragma unroll
      for (uint32_t w = 0; w < kNWV; w++) probes += s_wvia[w];

      // ---- visited set: claim-or-find across wavefronts
      const uint64_t k0 = (uint64_t)(fi2 + 1u) | ((uint64_t)(uint32_t)st2 << 32);
      uint32_t idx = key_hash32w(k0, M2, MW) & cap_mask;
      const uint32_t mytag = (round_no << 8) | (255u - tid);
      bool pending = viable, fresh = false;
      uint32_t spins = 0;
      while (pending) {
        uint64_t* e = tab + (uint64_t)idx * EW;
        const uint64_t k0e = ld64w(e);
        if ((uint32_t)k0e == 0u) {
          if (atomicCAS((unsigned long long*)e, 0ull, (unsigned long long)(k0 | kBusy)) == 0ull) {
#pragma unroll
            for (int j = 0; j < MW; j++) st64w(e + 1 + j, M2[j]);
            st64w(e + 1 + MW, (uint64_t)mytag | ((uint64_t)kNoneW << 32));
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");     // mask and tag before the key is readable
            st64w(e, k0);
            fresh = true; pending = false;
          }
        } else if (k0e & kBusy) {
          if (++spins > (1u << 22)) { atomicOr(&s_abort, 1u); pending = false; }
          __builtin_amdgcn_s_sleep(1);
        } else {
          bool same = k0e == k0;
          if (same) {
#pragma unroll
            for (int j = 0; j < MW; j++) same = same && ld64w(e + 1 + j) == M2[j];
          }
          if (same) {
            uint32_t* ow = reinterpret_cast<uint32_t*>(e + 1 + MW);
            if ((ld32w(ow) >> 8) == round_no) { atomicMax(ow, mytag); fresh = true; }
            pending = false;
          } else {
            idx = (idx + 1u) & cap_mask;
          }
        }
      }
      __threadfence();
      __syncthreads();     // every claim and every owner tag of the round is in place
      bool is_new = false;
      if (fresh) {
        uint64_t* e = tab + (uint64_t)idx * EW;
        is_new = ld32w(reinterpret_cast<uint32_t*>(e + 1 + MW)) == mytag;
        if (is_new) {
          st64w(e + 1 + MW, (uint64_t)mytag | ((uint64_t)pslot << 32));
          st64w(e + 2 + MW, (uint64_t)(op + 1u));
          my_maxf = max(my_maxf, fi2);
        }
      }
      const uint64_t nb = __ballot(is_new);
      if (lane == 0) s_wnew[wave] = (uint32_t)__popcll(nb);
      __syncthreads();
      uint32_t before = 0, total = 0;
#pragma unroll
      for (uint32_t w = 0; w < kNWV; w++) { if (w < wave) before += s_wnew[w]; total += s_wnew[w]; }
      if (is_new) {
        const uint32_t pos = sp + before + (uint32_t)__popcll(nb & ((1ull << lane) - 1ull));
        __hip_atomic_store(stack + pos, idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      sp += total; visited += total;
      if (s_abort) { verdict = TBC_UNKNOWN; cause = TBC_CAUSE_STEP_LIMIT; }
      if (round_no >= (1u << 24) - 2u) { verdict = TBC_UNKNOWN; cause = TBC_CAUSE_STEP_LIMIT; }
    }
    max_sp = max(max_sp, sp);
    __threadfence();
    __syncthreads();       // pushes visible to the next iteration's pops; LDS arrays free for reuse
    if (verdict == -2) {
      if (A.max_steps && probes > A.max_steps) { verdict = TBC_UNKNOWN; cause = TBC_CAUSE_STEP_LIMIT; }
      else if (A.time_limit_ticks && (iterations & 63u) == 0) {
        if (tid == 0 && (uint64_t)wall_clock64() - t0 > A.time_limit_ticks) s_abort = 2u;
        __syncthreads();
        if (s_abort == 2u) { verdict = TBC_UNKNOWN; cause = TBC_CAUSE_TIME_LIMIT; }
      }
    }
  }

  // ---- results: wavefront 0 finishes (max front, witness walk, stuck configs)
  __syncthreads();
  uint32_t maxf = my_maxf;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) maxf = max(maxf, (uint32_t)__shfl_xor(maxf, d));
  if (lane == 0) s_wnew[wave] = maxf;
  __syncthreads();
  maxf = max(max(s_wnew[0], s_wnew[1]), max(s_wnew[2], s_wnew[3]));
  if (wave != 0) return;

  uint32_t n_cfg = 0;
  if (verdict == TBC_INVALID && A.cfg) {
    uint64_t* cfg = A.cfg + (uint64_t)hidx * kCfgCap * (2 + MW);
    const uint64_t ncap = 1ull << cap_log2;
    for (uint64_t s0 = 0; s0 < ncap; s0 += 64) {
      const uint64_t* e = tab + (s0 + lane) * EW;
      const uint64_t k0 = ld64w(e);
      const bool hit = (uint32_t)k0 == maxf + 1u;
      const uint64_t hb = __ballot(hit);
      if (hit) {
        const uint32_t pos = n_cfg + (uint32_t)__popcll(hb & ((1ull << lane) - 1ull));
        if (pos < kCfgCap) {
          uint64_t* o = cfg + (uint64_t)pos * (2 + MW);
          o[0] = k0;
#pragma unroll
          for (int j = 0; j < MW; j++) o[1 + j] = ld64w(e + 1 + j);
          const uint64_t opw = ld64w(e + 2 + MW);
          o[1 + MW] = opw == 0ull ? (uint64_t)TBC_NO_OP : opw - 1ull;
        }
      }
      n_cfg += (uint32_t)__popcll(hb);
    }
  }
  uint32_t wlen = 0;
  if (verdict == TBC_VALID && R != 0) {
    wlen = 1;
    uint32_t id = win_parent;
    for (;;) {
      const uint32_t par = (uint32_t)(ld64w(tab + (uint64_t)id * EW + 1 + MW) >> 32);
      if (par == kNoneW) break;
      wlen++; id = par;
    }
    if (A.witness) {
      uint32_t* wit = A.witness + op_off;
      uint32_t w = wlen - 1;
      if (lane == 0) wit[w] = win_op;
      id = win_parent;
      for (;;) {
        const uint32_t par = (uint32_t)(ld64w(tab + (uint64_t)id * EW + 1 + MW) >> 32);
        if (par == kNoneW) break;
        w--;
        if (lane == 0) wit[w] = (uint32_t)ld64w(tab + (uint64_t)id * EW + 2 + MW) - 1u;
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
}

}  // namespace

uint32_t beam_wg_entry_words(uint32_t mask_words) { return mask_words + 3; }

bool launch_beam_wg(const BeamArgs& a, uint32_t mask_words, uint32_t n_hist, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const bool comm = a.model_kind == TBC_MODEL_SET || a.model_kind == TBC_MODEL_BANK;
#define TBC_LAUNCH_WG(MWV)                                                                              \
  do {                                                                                                  \
    if (comm) hipLaunchKernelGGL((wgl_beam_wg_kernel<MWV, true>), dim3(n_hist), dim3(kNT), 0, s, a);   \
    else hipLaunchKernelGGL((wgl_beam_wg_kernel<MWV, false>), dim3(n_hist), dim3(kNT), 0, s, a);       \
  } while (0)
  switch (mask_words) {
    case 1: TBC_LAUNCH_WG(1); return true;
    case 2: TBC_LAUNCH_WG(2); return true;
    case 4: TBC_LAUNCH_WG(4); return true;
    default: return false;
  }
#undef TBC_LAUNCH_WG
}

}  // namespace tbc
