// pack.hip -- K1: history pack kernel (gfx950).
//
// Takes the op-level history as it crosses the C-ABI (SoA columns, invocation
// order; include/tbcheck.h `tbc_ops`) and builds, entirely on the device, the
// layout the search kernel walks:
//
//   * completion ranks: ret_rank(op) = number of completions positioned before
//     its own, inv_rank(op) = number of completions positioned before its
//     invocation.  An op is open at front F iff inv_rank <= F <= ret_rank, so
//     the search never needs history positions again.  Ranks come from a
//     position bitmap (one bit per history row) + a popcount prefix per word.
//   * ret_slot[r] / ret_op[r]: process slot and op index of the completion of
//     rank r (the knossos.wgl return entries, in order).
//   * slot-major records: a stable counting sort of the ops by process -- the
//     per-process op lists knossos.linear.config keeps its pending calls in
//      -- each list bracketed by a head and a tail sentinel so
//     the per-lane cursors of the search kernel need no bounds checks.
//
// One workgroup per history (256 threads in a big batch, 1,024 when a few histories must be quick), grid-strided over the batch.  The
// column reads are coalesced (lane i reads row base+i of each column); the
// record scatter is 32 B per op.  Algorithmic HBM bytes per op: 21 B of
// columns read + 32 B record + 8 B ret_slot/ret_op written (+ 12 B scratch
// written and re-read), about 85 B/op -- the kernel is a small fraction of a
// check.
//
// Also validates what the search relies on: invocations strictly ascending,
// completion after invocation, positions in range and unique, process ids in
// range, one open op per process at a time, ops understood by the model.
#include <hip/hip_runtime.h>
#include "tbc_internal.h"

namespace tbc {

namespace {

__device__ __forceinline__ uint32_t ld_agent(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ bool txn_ok(const PackArgs& A, int32_t a, int32_t b) {
  if (a < 0 || b < 0 || (uint64_t)a + 3ull * (uint64_t)b > A.pool_len) return false;
  for (int32_t i = 0; i < b; i++) {
    const int32_t mf = A.pool_vals[a + 3 * i], k = A.pool_vals[a + 3 * i + 1], v = A.pool_vals[a + 3 * i + 2];
    if ((mf != 0 && mf != 1) || k < 0 || (uint32_t)k >= A.n_keys || k >= 8) return false;
    if (!(v == TBC_NIL && mf == 0) && (v < 0 || v > 13)) return false;
  }
  return true;
}

__device__ __forceinline__ bool op_ok_for_model(uint32_t kind, uint32_t f, int32_t a, uint32_t n_classes) {
  switch (kind) {
    case TBC_MODEL_REGISTER: return f == TBC_F_READ || f == TBC_F_WRITE;
    case TBC_MODEL_CAS_REGISTER: return f == TBC_F_READ || f == TBC_F_WRITE || f == TBC_F_CAS;
    case TBC_MODEL_MUTEX: return f == TBC_F_ACQUIRE || f == TBC_F_RELEASE;
    case TBC_MODEL_TABLE: return f == TBC_F_CLASS && (uint32_t)a < n_classes;
    case TBC_MODEL_SET: return f == TBC_F_ADD || f == TBC_F_READ;
    case TBC_MODEL_BANK: return f == TBC_F_TRANSFER || f == TBC_F_READ;
    default: return false;
  }
}

}  // namespace

__global__ __launch_bounds__(1024, 6) void pack_kernel(PackArgs A) {
  const uint32_t NT = blockDim.x;
  __shared__ uint32_t s_cnt[kMaxSlots];
  __shared__ uint32_t s_seg[kMaxSlots + 1];
  __shared__ uint32_t s_part[1024];
  __shared__ uint32_t s_err, s_total, s_done;
  const uint32_t tid = threadIdx.x;

  for (uint32_t h = A.h0 + blockIdx.x; h < A.n_hist; h += gridDim.x) {
    Hist* H = &A.hist[h];
    const uint32_t n = H->n_ops, W = H->n_slots, E = H->n_events;
    // count form (tbc_internal.h, kRuleCount): the process column holds re-used slots and a crashed call holds none -- it gets its
    // ranks here and nothing else (no place in a slot's record list: the classes of crashed calls are the host's to build)
    const bool cf = (H->flags & kHistCount) != 0u;
    const uint8_t* f = A.f + H->op_off;
    const int32_t* a = A.a + H->op_off;
    const int32_t* b = A.b + H->op_off;
    const int32_t* proc = A.process + H->op_off;
    const uint32_t* inv = A.inv_pos + H->op_off;
    const uint32_t* ret = A.ret_pos + H->op_off;
    uint32_t* bm = A.bitmap + H->bm_off;
    uint32_t* wpre = A.wpre + H->bm_off;
    const uint32_t nw = E / 32 + 1;
    uint32_t* sc_inv = A.scratch + H->frame_off;
    uint32_t* sc_ret = sc_inv + n;
    uint32_t* sc_dst = sc_ret + n;
    Rec* rec = A.rec + H->rec_off;

    if (tid == 0) { s_err = 0; s_done = 0; if (A.dbg) { A.dbg[0] = 0x10u; A.dbg[1] = h; } }
    for (uint32_t p = tid; p < W; p += NT) s_cnt[p] = 0;
    __syncthreads();

    // phase 1: validate rows, set completion bits, count ops per process
    // (the kernel is bound by memory latency, not issue slots: four rows' columns are requested before the first is
    // looked at -- the atomics below would otherwise keep every row's loads behind the previous row's)
    for (uint32_t i0 = tid; i0 < n; i0 += 4u * NT) {
      uint32_t ivs[4], rts[4], pvs[4]; int32_t ps[4], as[4], bs[4]; uint8_t fs[4];
#pragma unroll
      for (uint32_t k = 0; k < 4; k++) {
        const uint32_t i = i0 + k * NT;
        const bool in = i < n;
        ivs[k] = in ? inv[i] : 0u; rts[k] = in ? ret[i] : 0u; ps[k] = in ? proc[i] : 0;
        pvs[k] = (in && i > 0) ? inv[i - 1] : 0u;
        fs[k] = in ? f[i] : (uint8_t)0; as[k] = in ? a[i] : 0; bs[k] = in ? b[i] : 0;
      }
#pragma unroll
      for (uint32_t k = 0; k < 4; k++) {
        const uint32_t i = i0 + k * NT;
        if (i >= n) continue;
        const uint32_t iv = ivs[k], rt = rts[k];
        const int32_t p = ps[k];
        const bool slotless = cf && rt == TBC_POS_CRASHED;
        bool bad = iv >= E || (!slotless && (p < 0 || (uint32_t)p >= W)) || (i > 0 && pvs[k] >= iv);
        if (rt != TBC_POS_CRASHED) bad = bad || rt <= iv || rt >= E;
        if (bad) { atomicOr(&s_err, (uint32_t)TBC_ERR_BAD_HISTORY); continue; }
        if (!(A.model_kind == TBC_MODEL_MULTI_REGISTER ? (fs[k] == TBC_F_TXN && txn_ok(A, as[k], bs[k]))
                                                      : op_ok_for_model(A.model_kind, fs[k], as[k], A.n_classes))) { atomicOr(&s_err, 0x100u | (uint32_t)TBC_ERR_MODEL); continue; }
        if (rt != TBC_POS_CRASHED) { atomicOr(&bm[rt >> 5], 1u << (rt & 31)); atomicAdd(&s_done, 1u); }
        if (!slotless) atomicAdd(&s_cnt[p], 1u);
      }
    }
    __syncthreads();
    // phase 1b: set / bank keep their values in the pool -- every offset the search will dereference
    // (device_common.h pair_viable) is checked here against pool_len, the add count and the account count
    // (the condition is latched before anyone may write s_err again: a fast wavefront's atomicOr below must not
    // make a slow one skip the block and its barriers)
    const bool do1b = !s_err && (A.model_kind == TBC_MODEL_SET || A.model_kind == TBC_MODEL_BANK);
    __syncthreads();
    if (do1b) {
      const uint64_t PL = A.pool_len, Rn = s_done;
      const int64_t aux = H->aux;
      if (A.model_kind == TBC_MODEL_SET) {
        if (tid == 0) s_total = 0;
        __syncthreads();
        uint32_t mine = 0;
        for (uint32_t i = tid; i < n; i += NT) mine += f[i] == TBC_F_ADD;
        if (mine) atomicAdd(&s_total, mine);
        __syncthreads();
        const uint64_t n_adds = s_total, nwords = n_adds ? (n_adds + 31) / 32 : 1;
        bool bad = aux < 0 || (uint64_t)aux + Rn + 1 > PL;
        for (uint32_t i = tid; i < n && !bad; i += NT) {
          if (f[i] == TBC_F_ADD) bad = a[i] < 0 || (uint64_t)a[i] >= n_adds;
          else if (a[i] != TBC_NIL) bad = a[i] < 0 || (uint64_t)a[i] + 2 + nwords > PL;
        }
        if (bad) atomicOr(&s_err, 0x100u | (uint32_t)TBC_ERR_MODEL);
      } else {
        const uint64_t NA = A.n_keys;
        bool bad = NA == 0 || NA > 16 || aux < 0 || (uint64_t)aux + (Rn + 1) * NA > PL;
        for (uint32_t i = tid; i < n && !bad; i += NT) {
          if (f[i] == TBC_F_TRANSFER) {
            bad = a[i] < 0 || (uint64_t)a[i] + 3 > PL;
            if (!bad) { const int32_t d = A.pool_vals[a[i]], c = A.pool_vals[a[i] + 1]; bad = d < 0 || c < 0 || (uint64_t)d >= NA || (uint64_t)c >= NA; }
          } else if (a[i] != TBC_NIL) bad = a[i] < 0 || (uint64_t)a[i] + NA > PL;
        }
        if (bad) atomicOr(&s_err, 0x100u | (uint32_t)TBC_ERR_MODEL);
      }
      __syncthreads();
    }
    if (s_err) {
      if (tid == 0) { H->n_ret = 0; H->status = (s_err & 0x100u) ? (uint32_t)TBC_ERR_MODEL : (uint32_t)TBC_ERR_BAD_HISTORY; }
      __syncthreads();
      continue;
    }

    if (tid == 0 && A.dbg) A.dbg[0] = 0x20u;
    // phase 2: exclusive popcount prefix per bitmap word
    const uint32_t chunk = (nw + NT - 1) / NT;
    const uint32_t lo = min(tid * chunk, nw), hi = min(lo + chunk, nw);
    uint32_t sum = 0;
    for (uint32_t w = lo; w < hi; w++) sum += __popc(ld_agent(&bm[w]));
    s_part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
      uint32_t run = 0;
      for (uint32_t t = 0; t < NT; t++) { uint32_t x = s_part[t]; s_part[t] = run; run += x; }
      s_total = run;
    }
    __syncthreads();
    {
      uint32_t run = s_part[tid];
      for (uint32_t w = lo; w < hi; w++) { wpre[w] = run; run += __popc(ld_agent(&bm[w])); }
    }
    __syncthreads();
    const uint32_t R = s_total;
    if (R != s_done) {   // two completions on one history row
      if (tid == 0) { H->n_ret = 0; H->status = (uint32_t)TBC_ERR_BAD_HISTORY; }
      __syncthreads();
      continue;
    }

    if (tid == 0 && A.dbg) A.dbg[0] = 0x30u;
    // phase 3: ranks; completion order tables
    for (uint32_t i0 = tid; i0 < n; i0 += 2u * NT) {            // two rows per trip: positions, then the words they index
      uint32_t ivs[2], rts[2], wi[2], bi[2], wr[2], br[2]; int32_t ps[2];
#pragma unroll
      for (uint32_t k = 0; k < 2; k++) {
        const uint32_t i = i0 + k * NT;
        const bool in = i < n;
        ivs[k] = in ? inv[i] : 0u; rts[k] = in ? ret[i] : TBC_POS_CRASHED; ps[k] = in ? proc[i] : 0;
      }
#pragma unroll
      for (uint32_t k = 0; k < 2; k++) {
        const bool live = rts[k] != TBC_POS_CRASHED;
        wi[k] = ld_agent(&wpre[ivs[k] >> 5]); bi[k] = ld_agent(&bm[ivs[k] >> 5]);
        wr[k] = live ? ld_agent(&wpre[rts[k] >> 5]) : 0u; br[k] = live ? ld_agent(&bm[rts[k] >> 5]) : 0u;
      }
#pragma unroll
      for (uint32_t k = 0; k < 2; k++) {
        const uint32_t i = i0 + k * NT;
        if (i >= n) continue;
        const uint32_t iv = ivs[k], rt = rts[k];
        const uint32_t ir = wi[k] + __popc(bi[k] & ((1u << (iv & 31)) - 1u));
        uint32_t rr = kInf;
        if (rt != TBC_POS_CRASHED) {
          rr = wr[k] + __popc(br[k] & ((1u << (rt & 31)) - 1u));
          A.ret_slot[H->ret_off + rr] = (uint32_t)ps[k];
          A.ret_op[H->ret_off + rr] = i;
        }
        sc_inv[i] = ir;
        sc_ret[i] = rr;
      }
    }

    // phase 4: segment starts (each list gets a head and a tail sentinel)
    if (tid == 0) {
      uint32_t run = 0;
      for (uint32_t p = 0; p < W; p++) { s_seg[p] = run; run += s_cnt[p] + 2; }
      s_seg[W] = run;
    }
    __syncthreads();
    for (uint32_t p = tid; p <= W; p += NT) A.seg[H->seg_off + p] = s_seg[p];
    for (uint32_t p = tid; p < W; p += NT) s_cnt[p] = 0;   // now: ops placed so far
    __syncthreads();

    if (tid == 0 && A.dbg) A.dbg[0] = 0x40u;
    // phase 5: stable position of every op inside its process list.  One wave walks the ops in invocation order, 64
    // at a time: an op's place = the ops of its process placed by earlier chunks (s_cnt) + the lower lanes of this
    // chunk that hold an op of the same process (64 broadcasts, counted in registers).  The next chunk's column is
    // requested before this one is worked on.
    if (tid < 64) {
      const uint32_t lane = tid;
      const auto slot_of = [&](uint32_t i) -> uint32_t {            // 0xFFFFFFFF past the end / for a slotless call: equal to no process
        return (i < n && !(cf && ret[i] == TBC_POS_CRASHED)) ? (uint32_t)proc[i] : 0xFFFFFFFFu;
      };
      uint32_t p_next = slot_of(lane);
      for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t i = base + lane;
        const uint32_t p = p_next;
        const bool valid = p != 0xFFFFFFFFu;
        p_next = slot_of(i + 64u);
        uint32_t before = 0;
#pragma unroll 8
        for (uint32_t l = 0; l < 64; l++) {
          const uint32_t pl = __builtin_amdgcn_readlane(p, l);
          before += (pl == p && l < lane) ? 1u : 0u;
        }
        if (valid) sc_dst[i] = s_seg[p] + 1u + __hip_atomic_load(&s_cnt[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + before;
        else if (i < n) sc_dst[i] = kInf;                           // slotless: no record
        __builtin_amdgcn_wave_barrier();
        if (valid) atomicAdd(&s_cnt[p], 1u);
        __builtin_amdgcn_wave_barrier();
      }
    }
    __syncthreads();

    if (tid == 0 && A.dbg) A.dbg[0] = 0x50u;
    // phase 6: scatter the records, write the sentinels
    for (uint32_t i0 = tid; i0 < n; i0 += 4u * NT) {            // four records per trip
      Rec r[4]; uint32_t dst[4];
#pragma unroll
      for (uint32_t k = 0; k < 4; k++) {
        const uint32_t i = i0 + k * NT;
        const bool in = i < n;
        r[k].inv_rank = in ? sc_inv[i] : 0u; r[k].ret_rank = in ? sc_ret[i] : 0u; r[k].opidx = i; r[k].f = in ? f[i] : 0u;
        r[k].a = in ? a[i] : 0; r[k].b = in ? b[i] : 0;
        r[k].cls = rec_cls(r[k].f, r[k].a, r[k].ret_rank == kInf); r[k].prod = look_prod(r[k].f, r[k].a, r[k].b);
        dst[k] = in ? ld_agent(&sc_dst[i]) : 0u;
      }
#pragma unroll
      for (uint32_t k = 0; k < 4; k++) if (i0 + k * NT < n && dst[k] != kInf) rec[dst[k]] = r[k];
    }
    for (uint32_t p = tid; p < W; p += NT) {
      Rec hd; hd.inv_rank = 0; hd.ret_rank = 0; hd.opidx = kInf; hd.f = kFNone; hd.a = 0; hd.b = 0; hd.cls = 0; hd.prod = kLookNone;
      Rec tl = hd; tl.inv_rank = kInf; tl.ret_rank = kInf;
      rec[s_seg[p]] = hd;
      rec[s_seg[p + 1] - 1] = tl;
    }
    __syncthreads();

    if (tid == 0 && A.dbg) A.dbg[0] = 0x60u;
    // phase 7: one open op per process: the previous op of the same process
    // must have completed before this one was invoked
    for (uint32_t i0 = tid; i0 < n; i0 += 4u * NT) {            // four ops per trip (two dependent trips each)
      uint32_t d[4], mine[4], prev_ret[4], prev_f[4];
#pragma unroll
      for (uint32_t k = 0; k < 4; k++) {
        const uint32_t i = i0 + k * NT;
        d[k] = i < n ? ld_agent(&sc_dst[i]) : 1u;           // (record 0 of the history is a head sentinel: d - 1 stays in range)
        if (d[k] == kInf) d[k] = 1u;                        // slotless (record 0's successor has no predecessor call: never an error)
        mine[k] = i < n ? sc_inv[i] : 0u;
      }
#pragma unroll
      for (uint32_t k = 0; k < 4; k++) {
        const uint32_t* prev = reinterpret_cast<const uint32_t*>(&rec[d[k] - 1]);
        prev_ret[k] = ld_agent(prev + 1); prev_f[k] = ld_agent(prev + 3);
      }
#pragma unroll
      for (uint32_t k = 0; k < 4; k++)
        if (i0 + k * NT < n && prev_f[k] != kFNone && !(prev_ret[k] < mine[k])) atomicOr(&s_err, (uint32_t)TBC_ERR_BAD_HISTORY);
    }
    __syncthreads();
    if (tid == 0) { H->n_ret = R; H->status = s_err ? (uint32_t)TBC_ERR_BAD_HISTORY : 0u; if (A.dbg) A.dbg[0] = 0x90u; }
    __syncthreads();
  }
}

void launch_pack(const PackArgs& a, void* stream) {
  const uint32_t cnt = a.n_hist - a.h0;
  uint32_t grid = cnt < 4096 ? cnt : 4096;
  hipLaunchKernelGGL(pack_kernel, dim3(grid), dim3(cnt <= 64 ? 1024 : 256), 0, (hipStream_t)stream, a);
}

}  // namespace tbc
