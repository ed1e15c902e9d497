// wgl_search.hip -- K3/K4: the Wing-Gong/Lowe linearizability search (gfx950).
//
// Takes over knossos.wgl/analysis (recalled in SURVEY.md section 8a; the library is
// not in /root/reference -- section 0 F1): a depth-first search over linearization
// prefixes with Lowe's memoisation of (linearized set, model state).
//
// Mapping to CDNA4 (one 64-lane wavefront per history, 4 per workgroup, each
// wave pulling histories off a device-wide queue -- K4 batch mode is the same
// kernel with many histories):
//
//   * A config is (front, mask, state): front = rank of the first completion
//     not yet linearized, mask = one bit per process "its open op is already
//     linearized", state = model state.  That is a bijective re-encoding of
//     Lowe's N-bit set (oracle/wgl_window.c proves it against oracle/wgl_ref.c)
//     and it is the knossos.linear.config layout: pending calls by process.
//   * LANE = PROCESS.  Lane l (and l+64, ... for wide windows) walks process
//     l's op list with a cursor held in registers: the record it has open at
//     the front (`cur`) and the next one (`nxt`, prefetched one step ahead so
//     its latency hides under the visited-set probe).  No LDS scatter, no
//     per-step loop over pending ops: all <= 64*MW candidates of a config are
//     model-stepped by one instruction stream.
//   * `__ballot` turns the per-lane "open, not linearized, model-consistent"
//     predicate into the candidate mask; candidates are taken in invocation
//     order (DPP min-reduce over op index), exactly knossos.wgl's entry order,
//     so the traversal, witness and counters equal the sequential algorithm's.
//   * The DFS stack lives in an LDS ring per wave (top RING frames) and is
//     written through to HBM by one coalesced store per push; pops below the
//     ring re-read HBM.  The HBM copy is also the witness.
//   * The visited set is an exact open-addressed table in HBM: entries are the
//     full key ((front+1 | state<<32), mask words), probed four entries (one
//     64 B line at MW=1) per round by four lanes.  A history is owned by ONE
//     wave, so inserts need no atomics.
//
// Integer/bitset work only -- no MFMA.  Bound: HBM latency/bandwidth of the
// visited-set probes ("roofline").
#include <hip/hip_runtime.h>
#include "tbc_internal.h"
#include "device_common.h"

namespace tbc {

namespace {

__host__ __device__ constexpr uint32_t frame_words(uint32_t mw) { return 4 + 2 * mw; }
__host__ __device__ constexpr uint32_t ring_frames(uint32_t mw) { return mw <= 2 ? 128 : (mw <= 4 ? 64 : 32); }

__device__ __forceinline__ Rec load_rec(const Rec* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  const uint4 x = q[0], y = q[1];
  Rec r;
  r.inv_rank = x.x; r.ret_rank = x.y; r.opidx = x.z; r.f = x.w;
  r.a = (int32_t)y.x; r.b = (int32_t)y.y; r.cls = 0; r.prod = 0;
  return r;
}

template <int MW>
__device__ void search_one(const SearchArgs& A, const uint32_t hidx, uint32_t* ring, const uint32_t lane) {
  constexpr uint32_t FW = frame_words(MW);
  constexpr uint32_t KW = 1 + MW;           // u64 words per visited-set entry
  constexpr uint32_t RING = ring_frames(MW);

  // Every per-history quantity is wave-uniform; pin it in SGPRs explicitly (a value
  // loaded through the vector path is otherwise treated as divergent by the compiler).
  const Hist* H = A.hist + hidx;
  const Rec* rec = A.rec + ru64(H->rec_off);
  const uint32_t* seg = A.seg + ru64(H->seg_off);
  const uint64_t ret_off = ru64(H->ret_off);
  const uint32_t* ret_slot = A.ret_slot + ret_off;
  uint32_t* frames = A.frames + ru64(H->frame_off);
  uint64_t* tab = A.tab + ru64(H->tab_off);
  const uint64_t op_off = ru64(H->op_off);
  const uint32_t W = rfl(H->n_slots), R = rfl(H->n_ret), status = rfl(H->status);
  const uint64_t cap = 1ull << rfl(H->tab_log2);
  const uint64_t cap_mask = cap - 1;
  const uint64_t full_at = cap - (cap >> 2);       // 75 % load => give up (host retries bigger)
  DevResult* out = A.results + hidx;
  Model model{A.model_kind, A.table, A.n_classes, A.pool_vals, 0, 0};

  uint64_t steps = 0, visited = 0, probes = 0, backtracks = 0, max_depth = 0, bucket_reads = 0;
  int32_t verdict = -2, cause = TBC_CAUSE_NONE;

  if (A.dbg && lane == 0) { A.dbg[4] = 0x100u + hidx; A.dbg[5] = R; A.dbg[6] = status; A.dbg[7] = W; }
  if (status != 0) verdict = TBC_UNKNOWN;   // pack rejected it; the host reports the status
  else if (R == 0) verdict = TBC_VALID;

  // ---- per-lane cursors (lane + 64*j = process slot)
  Rec cur[MW], nxt[MW];
  uint32_t kpos[MW];
#pragma unroll
  for (int j = 0; j < MW; j++) {
    const uint32_t slot = lane + 64u * j;
    if (slot < W && verdict == -2) {
      kpos[j] = seg[slot];
      cur[j] = load_rec(rec + kpos[j]);        // head sentinel
      nxt[j] = load_rec(rec + kpos[j] + 1);
    } else {
      kpos[j] = 0;
      cur[j].inv_rank = 0; cur[j].ret_rank = 0; cur[j].opidx = kInf; cur[j].f = kFNone; cur[j].a = 0; cur[j].b = 0;
      nxt[j] = cur[j]; nxt[j].inv_rank = kInf;
    }
  }

  uint32_t fi = 0, depth = 0, ring_lo = 0, maxf = 0, from = 0;
  int32_t st = A.init_state;
  uint64_t M[MW];
#pragma unroll
  for (int j = 0; j < MW; j++) M[j] = 0;
  // chunk of ret_slot held one entry per lane
  uint32_t rs_base = 0;
  uint32_t rs_val = (lane < R && verdict == -2) ? ret_slot[lane] : 0u;

  const uint64_t t0 = A.time_limit_ticks ? wall_clock64() : 0;

  while (verdict == -2) {
    // ---- A: bring every cursor to the front (forward after a push, backward after a pop)
#pragma unroll
    for (int j = 0; j < MW; j++) {
      while (nxt[j].inv_rank <= fi) { cur[j] = nxt[j]; kpos[j]++; nxt[j] = load_rec(rec + kpos[j] + 1); }
      while (cur[j].inv_rank > fi) { nxt[j] = cur[j]; kpos[j]--; cur[j] = load_rec(rec + kpos[j]); }
    }
    maxf = max(maxf, fi);

    // ---- B: candidates = open at the front, not linearized, at/after `from`, model-consistent
    bool cand[MW];
    uint64_t Cm[MW];
    uint64_t any = 0;
#pragma unroll
    for (int j = 0; j < MW; j++) {
      cand[j] = cur[j].ret_rank >= fi && !((M[j] >> lane) & 1ull) && cur[j].opidx >= from &&
                cur[j].f != kFNone && model.ok(st, cur[j].f, cur[j].a, cur[j].b);
      Cm[j] = __ballot(cand[j]);
      any |= Cm[j];
    }

    bool descended = false;
    while (any) {
      // first candidate in invocation order
      uint32_t key = kInf;
#pragma unroll
      for (int j = 0; j < MW; j++) key = min(key, cand[j] ? cur[j].opidx : kInf);
      const uint32_t best = wave_min_u32(key);
      uint32_t bl = 0, bf = 0, bret = 0;
      int32_t ba = 0, bb = 0;
      int bj = 0;
#pragma unroll
      for (int j = 0; j < MW; j++) {
        const uint64_t hit = __ballot(cand[j] && cur[j].opidx == best);
        if (hit) {
          bj = j;
          bl = (uint32_t)__builtin_ctzll(hit);
          bf = rl(cur[j].f, bl); ba = (int32_t)rl((uint32_t)cur[j].a, bl); bb = (int32_t)rl((uint32_t)cur[j].b, bl);
          bret = rl(cur[j].ret_rank, bl);
        }
      }
      steps++;
      if (A.dbg && (steps & 63u) == 1u && lane == 0) {
        A.dbg[8] = hidx; A.dbg[9] = fi; A.dbg[10] = depth; A.dbg[11] = (uint32_t)steps;
        A.dbg[12] = (uint32_t)visited; A.dbg[13] = best; A.dbg[14] = (uint32_t)any; A.dbg[15] = (uint32_t)(any >> 32);
      }
      if (A.max_steps && steps > A.max_steps) { verdict = TBC_UNKNOWN; cause = TBC_CAUSE_STEP_LIMIT; break; }
      if (A.time_limit_ticks && (steps & 255u) == 0 && (uint64_t)wall_clock64() - t0 > A.time_limit_ticks) {
        verdict = TBC_UNKNOWN; cause = TBC_CAUSE_TIME_LIMIT; break;
      }

      // ---- child config
      const int32_t st2 = model.apply(st, bf, ba, bb);
      uint64_t M2[MW];
#pragma unroll
      for (int j = 0; j < MW; j++) M2[j] = M[j] | ((j == bj) ? (1ull << bl) : 0ull);
      uint32_t fi2 = fi;
      if (bret == fi) {   // the front's own op: the front moves past every completion already linearized
        uint32_t slot = bl + 64u * (uint32_t)bj;
        for (;;) {
#pragma unroll
          for (int j = 0; j < MW; j++) if ((slot >> 6) == (uint32_t)j) M2[j] &= ~(1ull << (slot & 63u));
          fi2++;
          if (fi2 == R) break;
          if (fi2 - rs_base >= 64u) {
            rs_base = fi2 & ~63u;
            rs_val = (rs_base + lane < R) ? ret_slot[rs_base + lane] : 0u;
          }
          slot = rl(rs_val, fi2 - rs_base);
          uint64_t bit = 0;
#pragma unroll
          for (int j = 0; j < MW; j++) if ((slot >> 6) == (uint32_t)j) bit = (M2[j] >> (slot & 63u)) & 1ull;
          if (!bit) break;
        }
      }

      // ---- visited set: exact lookup / insert
      uint64_t k0 = (uint64_t)(fi2 + 1u) | ((uint64_t)(uint32_t)st2 << 32);
      uint64_t hsh = mix64(k0);
#pragma unroll
      for (int j = 0; j < MW; j++) hsh = mix64(hsh ^ M2[j]) + 0x9E3779B97F4A7C15ull;
      probes++;
      uint64_t idx = hsh & cap_mask & ~3ull;
      bool is_new = false;
      for (;;) {
        // lanes 0..3 read four consecutive entries
        const uint64_t e = idx + (lane & 3u);
        const uint64_t* ep = tab + e * KW;
        bool match = false, empty = false;
        if (lane < 4) {
          if constexpr (MW == 1) {
            const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(ep);
            empty = (uint32_t)v.x == 0u;
            match = v.x == k0 && v.y == M2[0];
          } else {
            const uint64_t w0 = ep[0];
            empty = (uint32_t)w0 == 0u;
            match = w0 == k0;
#pragma unroll
            for (int j = 0; j < MW; j++) match = match && ep[1 + j] == M2[j];
          }
        }
        bucket_reads++;
        const uint64_t mm = __ballot(match) & 0xFull, em = __ballot(empty) & 0xFull;
        const uint32_t first_empty = em ? (uint32_t)__builtin_ctzll(em) : 4u;
        const uint32_t first_match = mm ? (uint32_t)__builtin_ctzll(mm) : 4u;
        if (first_match < first_empty) { is_new = false; break; }
        if (first_empty < 4u) {
          if (visited >= full_at) { verdict = TBC_UNKNOWN; cause = TBC_CAUSE_VISITED_FULL; break; }
          if (lane == first_empty) {
            uint64_t* wp = tab + (idx + first_empty) * KW;
            if constexpr (MW == 1) {
              ulonglong2 v; v.x = k0; v.y = M2[0];
              *reinterpret_cast<ulonglong2*>(wp) = v;
            } else {
#pragma unroll
              for (int j = 0; j < MW; j++) wp[1 + j] = M2[j];
              wp[0] = k0;
            }
          }
          is_new = true;
          break;
        }
        idx = (idx + 4) & cap_mask;
      }
      if (verdict != -2) break;

      if (is_new) {
        visited++;
        // ---- push the parent frame: LDS ring + coalesced write-through to HBM
        uint32_t w;
        w = lane == 0 ? fi : (lane == 1 ? (uint32_t)st : (lane == 2 ? best : 0u));
#pragma unroll
        for (int j = 0; j < MW; j++) {
          if (lane == 4u + 2u * j) w = (uint32_t)M[j];
          if (lane == 5u + 2u * j) w = (uint32_t)(M[j] >> 32);
        }
        if (lane < FW) {
          ring[(depth % RING) * FW + lane] = w;
          frames[(uint64_t)depth * FW + lane] = w;
        }
        depth++;
        max_depth = max(max_depth, (uint64_t)depth);
        if (depth - ring_lo > RING) ring_lo = depth - RING;
        fi = fi2; st = st2; from = 0;
#pragma unroll
        for (int j = 0; j < MW; j++) M[j] = M2[j];
        if (fi == R) verdict = TBC_VALID;
        descended = true;
        break;
      }
      // already seen: knossos.wgl moves on to the next entry
#pragma unroll
      for (int j = 0; j < MW; j++) {
        if (j == bj) { Cm[j] &= ~(1ull << bl); if (lane == bl) cand[j] = false; }
      }
      any = 0;
#pragma unroll
      for (int j = 0; j < MW; j++) any |= Cm[j];
    }
    if (verdict != -2 || descended) continue;

    // ---- C: the front's completion cannot be passed from here: backtrack
    if (depth == 0) { verdict = TBC_INVALID; break; }
    depth--;
    backtracks++;
    uint32_t w = 0;
    if (depth >= ring_lo) {
      if (lane < FW) w = ring[(depth % RING) * FW + lane];
    } else {
      if (lane < FW) w = frames[(uint64_t)depth * FW + lane];
      ring_lo = depth;
    }
    fi = rl(w, 0);
    st = (int32_t)rl(w, 1);
    from = rl(w, 2) + 1u;
#pragma unroll
    for (int j = 0; j < MW; j++) M[j] = (uint64_t)rl(w, 4 + 2 * j) | ((uint64_t)rl(w, 5 + 2 * j) << 32);
    if (fi - rs_base >= 64u) {   // (also true when fi < rs_base: unsigned wrap)
      rs_base = fi & ~63u;
      rs_val = (rs_base + lane < R) ? ret_slot[rs_base + lane] : 0u;
    }
  }

  // ---- invalid: the configs stuck at the failing completion (knossos :configs), by a scan of the visited set
  uint32_t n_cfg = 0;
  if (verdict == TBC_INVALID && A.cfg) {
    uint64_t* cfg = A.cfg + (uint64_t)hidx * kCfgCap * (2 + MW);
    for (uint64_t s0 = 0; s0 < cap; s0 += 64) {
      const uint64_t* e = tab + (s0 + lane) * KW;
      const uint64_t k0 = e[0];
      const bool hit = (uint32_t)k0 == maxf + 1u;
      const uint64_t hb = __ballot(hit);
      if (hit) {
        const uint32_t pos = n_cfg + (uint32_t)__popcll(hb & ((1ull << lane) - 1ull));
        if (pos < kCfgCap) {
          uint64_t* o = cfg + (uint64_t)pos * (2 + MW);
          o[0] = k0;
#pragma unroll
          for (int j = 0; j < MW; j++) o[1 + j] = e[1 + j];
          o[1 + MW] = (uint64_t)TBC_NO_OP;
        }
      }
      n_cfg += (uint32_t)__popcll(hb);
    }
  }
  // ---- results
  if (A.dbg && lane == 0) { A.dbg[4] = 0x200u + hidx; A.dbg[16] = (uint32_t)verdict; A.dbg[17] = (uint32_t)steps; }
  if (verdict == TBC_VALID && A.witness) {
    uint32_t* wit = A.witness + op_off;
    for (uint32_t d = lane; d < depth; d += 64) wit[d] = frames[(uint64_t)d * FW + 2];
  }
  if (lane == 0) {
    out->valid = verdict; out->cause = cause; out->max_front = maxf; out->depth = depth;
    out->final_state = st;
    out->fail_op = TBC_NO_OP; out->prev_ok_op = TBC_NO_OP;
    if (verdict == TBC_INVALID) {
      const uint32_t* ret_op = A.ret_op + ret_off;
      out->fail_op = ret_op[maxf];
      if (maxf) out->prev_ok_op = ret_op[maxf - 1];
    }
    out->n_configs = n_cfg;
    out->steps = steps; out->visited = visited; out->probes = probes; out->backtracks = backtracks;
    out->max_depth = max_depth; out->bucket_reads = bucket_reads;
    if (verdict != TBC_UNKNOWN && A.progress) {           // tbc_batch_progress
      const uint32_t n = atomicAdd(A.progress_dev, 1u) + 1u;
      __hip_atomic_store(A.progress, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

template <int MW>
__global__ __launch_bounds__(kBlock) void wgl_search_kernel(SearchArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const uint32_t lane = threadIdx.x & 63u, wv = rfl(threadIdx.x >> 6);
  uint32_t* ring = lds + wv * (ring_frames(MW) * frame_words(MW));
  // one wavefront per history; the hardware workgroup dispatcher is the work queue
  const uint32_t w = blockIdx.x * kWavesPerBlock + wv;
  if (w < A.n_work) search_one<MW>(A, rfl(A.work[w]), ring, lane);
}

template <int MW>
void launch_mw(const SearchArgs& a, uint32_t n_blocks, hipStream_t s) {
  const size_t lds = (size_t)kWavesPerBlock * ring_frames(MW) * frame_words(MW) * 4;
  hipLaunchKernelGGL(wgl_search_kernel<MW>, dim3(n_blocks), dim3(kBlock), lds, s, a);
}

}  // namespace

uint32_t search_frame_words(uint32_t mask_words) { return frame_words(mask_words); }

bool launch_search(const SearchArgs& a, uint32_t mask_words, uint32_t n_blocks, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  switch (mask_words) {
    case 1: launch_mw<1>(a, n_blocks, s); return true;
    case 2: launch_mw<2>(a, n_blocks, s); return true;
    case 4: launch_mw<4>(a, n_blocks, s); return true;
    case 8: launch_mw<8>(a, n_blocks, s); return true;
    case 16: launch_mw<16>(a, n_blocks, s); return true;
    default: return false;
  }
}

}  // namespace tbc
