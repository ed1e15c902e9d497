// pack_open.hip -- K1b: open-call lists for the wide search kernel (gfx950).
//
// Runs after pack_kernel (which left every op's inv_rank / ret_rank in the
// scratch arena).  For every front F (= rank of a completion) the wide search
// needs the calls that are open at F.  This kernel builds, per history:
//
//   occ[F]   bit p set  <=>  process p has a LIVE (eventually completed) call open at F
//   off[F]   CSR offsets, off[F+1]-off[F] = popcount(occ[F])
//   lst[]    the live open calls of front F in process-slot order, as whole 16 B records
//            {op, f | slot << 8 | at-front flag, a, b}: a lane of the search reaches its candidate
//            with one load (position of p's call = popcount(occ[F] below p): no sorting needed)
//   crashed[] the :info calls in invocation order, ncr[F] = how many of them were
//            invoked before completion F (they stay open for ever, so they are
//            kept out of the per-front lists).  Crashed READS (value nil) are left
//            out altogether: no effect on the model, no constraint, but each would
//            double the config space (oracle/wgl_beam.c)
//   slot8[]  process slot of the call completing at each rank, one byte each: the search
//            prefetches a 16-rank window of it for the front advance
//   look[]   one lookahead record per completion rank (layout: tbc_internal.h): what the call
//            completing there needs, and which other calls could provide it
//
// This is knossos.linear.config's "pending calls by process" materialised for
// every point of the history.  One workgroup (256 threads in a big batch, 1,024 when a few histories must be quick) per
// history.  off/ncr/occ arenas are zeroed by the host before the launch.
#include <hip/hip_runtime.h>
#include "tbc_internal.h"

namespace tbc {

namespace {

__device__ __forceinline__ uint32_t ld_agent(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint64_t ld_agent64(const uint64_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// exclusive scan of part[0..NT) in place by the first wavefront (NT a multiple of 64): each lane scans its
// NT/64 consecutive entries, a wavefront scan joins the lanes.  All threads call it between two barriers.
__device__ __forceinline__ void scan_parts(uint32_t* part, uint32_t NT, uint32_t* total_slot) {
  if (threadIdx.x < 64) {
    const uint32_t lane = threadIdx.x, per = NT / 64, lo = lane * per;
    uint32_t sum = 0;
    for (uint32_t i = 0; i < per; i++) sum += part[lo + i];
    uint32_t x = sum;
    for (uint32_t d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d); if (lane >= d) x += y; }
    uint32_t run = x - sum;
    for (uint32_t i = 0; i < per; i++) { const uint32_t v = part[lo + i]; part[lo + i] = run; run += v; }
    if (lane == 63 && total_slot) *total_slot = x;
  }
}

// in-place exclusive scan of v[0..m) by the block; returns the total.
// `part` is blockDim.x words of LDS.  Every element is read and written by one thread only.
__device__ uint32_t block_exclusive_scan(uint32_t* v, uint32_t m, uint32_t* part, uint32_t* total_slot) {
  const uint32_t tid = threadIdx.x;
  const uint32_t NT = blockDim.x;
  const uint32_t chunk = (m + NT - 1) / NT;
  const uint32_t lo = min(tid * chunk, m), hi = min(lo + chunk, m);
  uint32_t sum = 0;
  for (uint32_t i = lo; i < hi; i++) sum += ld_agent(&v[i]);
  part[tid] = sum;
  __syncthreads();
  scan_parts(part, NT, total_slot);
  __syncthreads();
  uint32_t run = part[tid];
  for (uint32_t i = lo; i < hi; i++) { uint32_t x = ld_agent(&v[i]); v[i] = run; run += x; }
  __syncthreads();
  return *total_slot;
}

// register value a call must find / leaves behind, as a lookahead byte (0..31, else kLookNone)
__device__ __forceinline__ uint32_t look_val(int32_t v) { return (v >= 0 && v < 32) ? (uint32_t)v : kLookNone; }
__device__ __forceinline__ uint32_t look_need(uint32_t f, int32_t a) {
  return ((f == TBC_F_READ && a != TBC_NIL) || f == TBC_F_CAS) ? look_val(a) : kLookNone;
}
__device__ __forceinline__ uint32_t look_prod(uint32_t f, int32_t a, int32_t b) {
  return f == TBC_F_WRITE ? look_val(a) : (f == TBC_F_CAS ? look_val(b) : kLookNone);
}

}  // namespace

__global__ __launch_bounds__(1024) void pack_open_kernel(PackOpenArgs A) {
  __shared__ uint32_t s_part[1024];
  const uint32_t NT = blockDim.x;
  __shared__ uint32_t s_total;
  const uint32_t tid = threadIdx.x;
  const uint32_t MW = A.mask_words;

  for (uint32_t h = blockIdx.x; h < A.n_hist; h += gridDim.x) {
    const Hist* H = &A.hist[h];
    BeamHist* B = &A.bh[h];
    const uint32_t n = H->n_ops, R = H->n_ret;
    if (H->status != 0 || R == 0) {
      if (tid == 0) { B->status = 0; B->n_crashed = 0; }
      continue;
    }
    const uint8_t* f = A.f + H->op_off;
    const int32_t* a = A.a + H->op_off;
    const int32_t* b = A.b + H->op_off;
    const int32_t* proc = A.process + H->op_off;
    const uint32_t* sc_inv = A.scratch + H->frame_off;
    const uint32_t* sc_ret = sc_inv + n;
    uint32_t* off = A.off + B->off_off;      // R + 1 entries used
    uint32_t* ncr = A.ncr + B->off_off;      // R entries used
    uint64_t* occ = A.occ + B->occ_off;      // R * MW words used
    OpRec* lst = A.lst + B->lst_off;
    OpRec* crashed = A.crashed + H->op_off;
    const uint32_t* ret_slot = A.ret_slot + H->ret_off;
    uint8_t* slot8 = A.slot8 + slot8_off(H->op_off, h);

    // A: occupancy bits of live calls; crashed-call counts by front; completion slots as bytes
    for (uint32_t r = tid; r < R + 16u; r += NT) slot8[r] = r < R ? (uint8_t)ret_slot[r] : (uint8_t)0;
    for (uint32_t i = tid; i < n; i += NT) {
      const uint32_t ir = sc_inv[i], rr = sc_ret[i];
      const uint32_t p = (uint32_t)proc[i];
      if (rr == kInf) {
        if (ir < R && !(f[i] == TBC_F_READ && a[i] == TBC_NIL)) atomicAdd(&ncr[ir], 1u);
      } else {
        const unsigned long long bit = 1ull << (p & 63u);
        for (uint32_t fr = ir; fr <= rr; fr++)
          atomicOr((unsigned long long*)&occ[(uint64_t)fr * MW + (p >> 6)], bit);
      }
    }
    __syncthreads();
    // B: live count per front -> off[F+1]; then exclusive scan -> CSR offsets
    for (uint32_t fr = tid; fr < R; fr += NT) {
      uint32_t c = 0;
      for (uint32_t w = 0; w < MW; w++) c += __popcll(ld_agent64(&occ[(uint64_t)fr * MW + w]));
      off[fr] = c;
    }
    if (tid == 0) off[R] = 0;
    __syncthreads();
    const uint32_t total = block_exclusive_scan(off, R + 1, s_part, &s_total);
    // ncr[F] = crashed calls invoked before completion F (inclusive prefix)
    {
      // exclusive scan then add own count: do it as exclusive scan of a shifted view
      // (ncr[F] currently = #crashed with inv_rank == F)
      const uint32_t chunk = (R + NT - 1) / NT;
      const uint32_t lo = min(tid * chunk, R), hi = min(lo + chunk, R);
      uint32_t sum = 0;
      for (uint32_t i = lo; i < hi; i++) sum += ld_agent(&ncr[i]);
      s_part[tid] = sum;
      __syncthreads();
      scan_parts(s_part, NT, nullptr);
      __syncthreads();
      uint32_t run = s_part[tid];
      for (uint32_t i = lo; i < hi; i++) { run += ld_agent(&ncr[i]); ncr[i] = run; }
      __syncthreads();
    }
    if (total > B->lst_cap) {
      if (tid == 0) { B->status = 1; B->n_crashed = 0; }
      __syncthreads();
      continue;
    }
    // C: fill the per-front lists in process-slot order
    for (uint32_t i = tid; i < n; i += NT) {
      const uint32_t ir = sc_inv[i], rr = sc_ret[i];
      if (rr == kInf) continue;
      const uint32_t p = (uint32_t)proc[i];
      OpRec o; o.op = i; o.f_slot = (uint32_t)f[i] | (p << 8); o.a = a[i]; o.b = b[i];
      for (uint32_t fr = ir; fr <= rr; fr++) {
        uint32_t pos = ld_agent(&off[fr]);
        for (uint32_t w = 0; w < (p >> 6); w++) pos += __popcll(ld_agent64(&occ[(uint64_t)fr * MW + w]));
        pos += __popcll(ld_agent64(&occ[(uint64_t)fr * MW + (p >> 6)]) & ((1ull << (p & 63u)) - 1ull));
        if (fr == rr) o.f_slot |= kAtFront;
        lst[pos] = o;
      }
    }
    // D: crashed calls in invocation order (stable compaction)
    {
      const uint32_t chunk = (n + NT - 1) / NT;
      const uint32_t lo = min(tid * chunk, n), hi = min(lo + chunk, n);
      uint32_t cnt = 0;
      for (uint32_t i = lo; i < hi; i++) cnt += sc_ret[i] == kInf && !(f[i] == TBC_F_READ && a[i] == TBC_NIL);
      s_part[tid] = cnt;
      __syncthreads();
      scan_parts(s_part, NT, &s_total);
      __syncthreads();
      if (tid == 0) { B->n_crashed = s_total; B->status = 0; }
      uint32_t run = s_part[tid];
      for (uint32_t i = lo; i < hi; i++) if (sc_ret[i] == kInf && !(f[i] == TBC_F_READ && a[i] == TBC_NIL)) {
        OpRec o; o.op = i; o.f_slot = (uint32_t)f[i] | ((uint32_t)proc[i] << 8); o.a = a[i]; o.b = b[i];
        crashed[run++] = o;
      }
      __syncthreads();
    }
    // H: dominance tables (register family; tbc_internal.h) -- one thread per front reads that front's list
    //    once: rdm row = slots of its open reads by value, twn = per entry the slots of the calls with the
    //    same effect that complete earlier
    if (A.rdm) {
      __threadfence_block();
      const uint32_t V = A.vpad;
      uint64_t* rdm = A.rdm + H->op_off * V * MW;
      uint64_t* twn = A.twn + B->lst_off * MW;
      for (uint32_t fr = tid; fr < R; fr += NT) {
        const uint32_t o0 = ld_agent(&off[fr]), o1 = ld_agent(&off[fr + 1]);
        uint64_t* row = rdm + (uint64_t)fr * V * MW;
        for (uint32_t e = 0; e < V * MW; e++) row[e] = 0ull;
        // one pass over the front's list: reads go into the row (this thread's own words: plain read-modify-
        // write); writes / cas leave a signature bit per effect -- two calls on one bit MAY be twins
        uint64_t sig = 0ull;
        bool maybe_twins = false;
        for (uint32_t c = o0; c < o1; c++) {
          const OpRec x = lst[c];
          const uint32_t xf = x.f_slot & 0xFFu, xs = (x.f_slot >> 8) & kSlotMask;
          if (xf == TBC_F_READ) {
            const uint32_t vi = rdm_index(x.a, V);
            if (vi != 0 || x.a == TBC_NIL) row[vi * MW + (xs >> 6)] |= 1ull << (xs & 63u);
          } else if (xf == TBC_F_WRITE || xf == TBC_F_CAS) {
            const uint64_t bit = 1ull << (((uint32_t)x.a * 7u + (xf == TBC_F_CAS ? (uint32_t)x.b * 13u + 31u : 0u)) & 63u);
            maybe_twins = maybe_twins || (sig & bit) != 0ull;
            sig |= bit;
          }
        }
        for (uint32_t c = o0; c < o1; c++) {
          uint64_t m[4] = {0, 0, 0, 0};
          if (maybe_twins) {
            const OpRec y = lst[c];
            const uint32_t yf = y.f_slot & 0xFFu;
            if (yf == TBC_F_WRITE || yf == TBC_F_CAS) {
              const uint32_t yr = sc_ret[y.op];
              for (uint32_t d = o0; d < o1; d++) {
                if (d == c) continue;
                const OpRec z = lst[d];
                if ((z.f_slot & 0xFFu) != yf || z.a != y.a || (yf == TBC_F_CAS && z.b != y.b)) continue;
                const uint32_t zr = sc_ret[z.op];
                if (zr < yr || (zr == yr && z.op < y.op)) { const uint32_t zs = (z.f_slot >> 8) & kSlotMask; m[zs >> 6] |= 1ull << (zs & 63u); }
              }
            }
          }
          for (uint32_t w = 0; w < MW; w++) twn[(uint64_t)c * MW + w] = m[w];
        }
      }
      __syncthreads();
    }
    // E-G: lookahead records (register family only; A.look is null otherwise)
    if (A.look) {
      const uint32_t LW = 1 + MW;
      uint64_t* look = A.look + look_off(H->op_off, h, MW);
      const uint32_t* ret_op = A.ret_op + H->ret_off;
      uint32_t* tmp = A.tmp + H->op_off;
      __threadfence_block();
      // E: per rank -- need / prod / dinv, and the producers of `need` open at that front
      for (uint32_t t = tid; t < R + kLookPad; t += NT) {
        uint64_t w0 = (uint64_t)(kLookNone << 16 | kLookNone << 24) | (255ull << 32) | (255ull << 40);
        uint64_t pm[16];
        for (uint32_t w = 0; w < MW; w++) pm[w] = 0;
        if (t < R) {
          const uint32_t op = ret_op[t];
          const uint32_t need = look_need(f[op], a[op]), prod = look_prod(f[op], a[op], b[op]);
          const uint32_t di = min(t - sc_inv[op], 255u);
          w0 = (uint64_t)((uint32_t)proc[op] & 0xFFFFu) | (uint64_t)need << 16 | (uint64_t)prod << 24 |
               (uint64_t)di << 32 | (255ull << 40);
          if (need != kLookNone) {
            const uint32_t o0 = ld_agent(&off[t]), o1 = ld_agent(&off[t + 1]), nc = ld_agent(&ncr[t]);
            for (uint32_t c = 0; c < (o1 - o0) + nc; c++) {
              const OpRec x = c < o1 - o0 ? lst[o0 + c] : crashed[c - (o1 - o0)];
              const uint32_t xf = x.f_slot & 0xFFu, xs = (x.f_slot >> 8) & kSlotMask;
              if (x.op != op && look_prod(xf, x.a, x.b) == need) pm[xs >> 6] |= 1ull << (xs & 63u);
            }
          }
          tmp[t] = 255u;
        }
        look[(uint64_t)t * LW] = w0;
        for (uint32_t w = 0; w < MW; w++) look[(uint64_t)t * LW + 1 + w] = pm[w];
      }
      __syncthreads();
      // F: every producer tells the ranks right after its invocation how recent it is
      for (uint32_t i = tid; i < n; i += NT) {
        const uint32_t pv = look_prod(f[i], a[i], b[i]);
        if (pv == kLookNone || (sc_ret[i] == kInf && f[i] == TBC_F_READ)) continue;
        const uint32_t ir = sc_inv[i];
        for (uint32_t t = ir; t < R && t < ir + kLookahead; t++) {
          const uint32_t need = (uint32_t)(ld_agent64(&look[(uint64_t)t * LW]) >> 16) & 0xFFu;
          if (need == pv && ret_op[t] != i) atomicMin(&tmp[t], t - ir);
        }
      }
      __syncthreads();
      // G: fold dprod into the records
      for (uint32_t t = tid; t < R; t += NT) {
        const uint32_t d = ld_agent(&tmp[t]);
        if (d < 255u) {
          const uint64_t w0 = ld_agent64(&look[(uint64_t)t * LW]);
          look[(uint64_t)t * LW] = (w0 & ~(255ull << 40)) | ((uint64_t)d << 40);
        }
      }
      __syncthreads();
    }
  }
}

void launch_pack_open(const PackOpenArgs& a, void* stream) {
  uint32_t grid = a.n_hist < 4096 ? a.n_hist : 4096;
  // few histories: latency matters (tbc_check), give each the widest workgroup; many: occupancy matters
  hipLaunchKernelGGL(pack_open_kernel, dim3(grid), dim3(a.n_hist <= 64 ? 1024 : 256), 0, (hipStream_t)stream, a);
}

}  // namespace tbc
