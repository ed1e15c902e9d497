// pack_open.hip -- K1b: open-call lists for the wide search kernel and the level sweep (gfx950).
//
// Runs after pack_kernel (which left every op's inv_rank / ret_rank in the scratch arena and the per-process
// record lists rec[] / seg[]).  For every front F (= rank of a completion) the searches need the calls that
// are open at F.  Built per history:
//
//   off[F]   CSR offsets, off[F+1]-off[F] = number of live (eventually completed) calls open at F
//   lst[]    the live open calls of front F in process-slot order, as whole 16 B records
//            {op, f | slot << 8 | at-front flag, a, b}: a lane of the search reaches its candidate with one load
//   crashed[] the :info calls in invocation order, ncr[F] = how many of them were invoked before completion F
//            (they stay open for ever, so they are kept out of the per-front lists).  Crashed READS (value
//            nil) are left out altogether: no effect on the model, no constraint, but each would double the
//            config space (oracle/wgl_beam.c)
//   slot8[]  process slot of the call completing at each rank, one byte each
//   rk8[]    (narrow kernel) per rank: 0xFF = the completing call is not a read, else the rdm index of its value
//   twn[] rdm[]  the dominance tables (tbc_internal.h): twin masks per list entry, open-read masks per front
//   look[]   one lookahead record per completion rank (layout: tbc_internal.h)
//
// This is knossos.linear.config's "pending calls by process" materialised for every point of the history
// (SURVEY.md section 8a).  Three kernels:
//   open_counts_kernel  one workgroup per history: how many calls are open at each front -- a histogram of
//                       invocation ranks and two scans (open(F) = calls invoked by F minus the F completed) --
//                       the crashed-call list, slot8;
//   open_walk_kernel    one WAVEFRONT PER 64 FRONTS, LANE = PROCESS SLOT: each lane keeps the call its process
//                       has open (a cursor into the process's record list, the next record prefetched) and the
//                       wavefront walks its fronts in order; at every front one ballot is the occupancy mask,
//                       a popcount below the lane is the call's position in the front's list, V ballots are the
//                       read masks, the twins are found by broadcasting the (few) open writes / cas, the lookahead
//                       record comes from the lane of the completing call.  Streaming writes only, no atomics,
//                       every chunk of every history independent -- so one history is spread over the GPU and a
//                       batch costs one pass over what it writes;
//   open_dprod_kernel   lookahead only: how recently a producer of the needed value was invoked (scattered by
//                       the producers with atomicMin).
// The off / ncr arenas are zeroed by the host before the launch.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include "tbc_internal.h"
#include "open_walk_impl.h"

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

// The per-history kernels below are bound by memory latency, not by issue slots (3-9 % vector ALU use): a thread that
// waits for every load before it asks for the next spends a pass of 30 elements on 30 trips.  These helpers put up to
// eight loads in flight before the first is consumed; the passes keep their order of operations.
constexpr uint32_t kBatch = 8;
template <class F>
__device__ __forceinline__ void chunk_loads(const uint32_t* p, uint32_t lo, uint32_t hi, F&& each) {
  for (uint32_t i = lo; i < hi; i += kBatch) {
    uint32_t v[kBatch];
#pragma unroll
    for (uint32_t k = 0; k < kBatch; k++) v[k] = i + k < hi ? ld_agent(&p[i + k]) : 0u;
#pragma unroll
    for (uint32_t k = 0; k < kBatch; k++) if (i + k < hi) each(i + k, v[k]);
  }
}

}  // namespace

__global__ __launch_bounds__(1024) void open_counts_kernel(PackOpenArgs A) {
  __shared__ uint32_t s_part[1024];
  const uint32_t NT = blockDim.x;
  __shared__ uint32_t s_total;
  const uint32_t tid = threadIdx.x;
  const uint32_t MW = A.mask_words;

  for (uint32_t h = A.h0 + blockIdx.x; h < A.n_hist; h += gridDim.x) {
    const Hist* H = &A.hist[h];
    BeamHist* B = &A.bh[h];
    const uint32_t n = H->n_ops, R = H->n_ret;
    if (H->status != 0 || R == 0) {
      if (tid == 0) { B->status = 0; B->n_crashed = 0; B->lst_need = 0; }
      continue;
    }
    // count form: crashed calls are not candidates one by one -- their classes (crashed[], cmem[]) are built by the host when the
    // inputs become resident, and count_fronts_kernel below says how many classes each front can draw on
    const bool cf = (H->flags & kHistCount) != 0u;
    const uint8_t* f = A.f + H->op_off;
    const int32_t* a = A.a + H->op_off;
    const int32_t* b = A.b + H->op_off;
    const int32_t* proc = A.process + H->op_off;
    const uint32_t* sc_inv = A.scratch + H->frame_off;
    const uint32_t* sc_ret = sc_inv + n;
    uint32_t* off = A.off + B->off_off;      // R + 1 entries used
    uint32_t* ncr = A.ncr + B->off_off;      // R entries used
    OpRec* crashed = A.crashed + H->op_off;
    const uint32_t* ret_slot = A.ret_slot + H->ret_off;
    uint8_t* slot8 = A.slot8 + slot8_off(H->op_off, h);

    // A: histogram of the live calls' invocation ranks (in off[]), crashed-call counts by front, completion slots as bytes
    for (uint32_t r = tid; r < R + 16u; r += NT) slot8[r] = r < R ? (uint8_t)ret_slot[r] : (uint8_t)0;
    if (A.rk8) {          // what kind of call completes at each rank, for the narrow kernel's register-only front advance
      uint8_t* rk8 = A.rk8 + slot8_off(H->op_off, h);
      const uint32_t* ret_op = A.ret_op + H->ret_off;
      for (uint32_t r = tid; r < R + 16u; r += NT) {
        uint8_t k = 0xFF;
        if (r < R) { const uint32_t x = ret_op[r]; if (f[x] == TBC_F_READ) k = (uint8_t)rdm_index(a[x], A.vpad); }
        rk8[r] = k;
      }
    }
    for (uint32_t i0 = tid; i0 < n; i0 += 4u * NT) {          // four ops per trip
      uint32_t ir[4], rr[4]; bool nilread[4], isread[4];
#pragma unroll
      for (uint32_t k = 0; k < 4; k++) {
        const uint32_t i = i0 + k * NT;
        const bool in = i < n;
        ir[k] = in ? sc_inv[i] : 0u; rr[k] = in ? sc_ret[i] : 0u;
        isread[k] = in && f[i] == TBC_F_READ;
        nilread[k] = isread[k] && a[i] == TBC_NIL;
      }
#pragma unroll
      for (uint32_t k = 0; k < 4; k++) {
        if (i0 + k * NT >= n) continue;
        if (rr[k] == kInf) {
          if (!cf && ir[k] < R && !nilread[k]) atomicAdd(&ncr[ir[k]], 1u);
        } else if (!(A.branch_lists && isread[k])) {
          atomicAdd(&off[ir[k]], 1u);
        }
      }
    }
    __syncthreads();
    // B: open(F) = (live calls invoked at rank <= F) - F, since exactly F calls have completed before front F;
    //    off[] = exclusive scan of open().  Two block scans over per-thread chunks, in place.
    uint32_t total;
    {
      const uint32_t chunk = (R + NT - 1) / NT;
      const uint32_t lo = min(tid * chunk, R), hi = min(lo + chunk, R);
      // branch lists: a completed READ was never in a list, so the calls that have left the lists before front F are the
      // F completions minus the reads among them -- the reads of this thread's ranks as a bit mask (rk8 was written above)
      uint64_t rdbits = 0;
      uint32_t base_rd = 0;
      if (A.branch_lists) {
        const uint8_t* rk8 = A.rk8 + slot8_off(H->op_off, h);
        uint32_t nrd = 0;
        if (hi - lo <= 64u) { for (uint32_t i = lo; i < hi; i++) if (rk8[i] != 0xFFu) rdbits |= 1ull << (i - lo); nrd = (uint32_t)__popcll(rdbits); }
        else for (uint32_t i = lo; i < hi; i++) nrd += rk8[i] != 0xFFu;
        s_part[tid] = nrd;
        __syncthreads();
        scan_parts(s_part, NT, nullptr);
        __syncthreads();
        base_rd = s_part[tid];
        __syncthreads();
      }
      const bool wide_chunk = hi - lo > 64u;
      const uint8_t* rk8w = A.rk8 ? A.rk8 + slot8_off(H->op_off, h) : nullptr;
      const auto rd_at = [&](uint32_t i) -> uint32_t {          // is the call completing at rank i a read (branch lists only)
        if (!A.branch_lists) return 0u;
        return wide_chunk ? (uint32_t)(rk8w[i] != 0xFFu) : (uint32_t)((rdbits >> (i - lo)) & 1ull);
      };
      uint32_t sum = 0;
      chunk_loads(off, lo, hi, [&](uint32_t, uint32_t v) { sum += v; });
      s_part[tid] = sum;
      __syncthreads();
      scan_parts(s_part, NT, nullptr);
      __syncthreads();
      const uint32_t base_inv = s_part[tid];          // listed calls invoked before this chunk's first rank
      __syncthreads();
      uint32_t run = base_inv, open_sum = 0, rd = base_rd;
      chunk_loads(off, lo, hi, [&](uint32_t i, uint32_t v) { run += v; open_sum += run - (i - rd); rd += rd_at(i); });
      s_part[tid] = open_sum;
      __syncthreads();
      scan_parts(s_part, NT, &s_total);
      __syncthreads();
      uint32_t pos = s_part[tid];
      run = base_inv; rd = base_rd;
      chunk_loads(off, lo, hi, [&](uint32_t i, uint32_t v) { run += v; off[i] = pos; pos += run - (i - rd); rd += rd_at(i); });
      total = s_total;
      if (tid == 0) off[R] = total;
      __syncthreads();
    }
    // ncr[F] = crashed calls invoked before completion F (inclusive prefix)
    {
      const uint32_t chunk = (R + NT - 1) / NT;
      const uint32_t lo = min(tid * chunk, R), hi = min(lo + chunk, R);
      uint32_t sum = 0;
      chunk_loads(ncr, lo, hi, [&](uint32_t, uint32_t v) { sum += v; });
      s_part[tid] = sum;
      __syncthreads();
      scan_parts(s_part, NT, nullptr);
      __syncthreads();
      uint32_t run = s_part[tid];
      chunk_loads(ncr, lo, hi, [&](uint32_t i, uint32_t v) { run += v; ncr[i] = run; });
      __syncthreads();
    }
    if (tid == 0) B->lst_need = total;
    if (total > B->lst_cap) {
      if (tid == 0) { B->status = 1; B->n_crashed = 0; }
      __syncthreads();
      continue;
    }
    // D: crashed calls in invocation order (stable compaction)
    if (cf) {
      if (tid == 0) { B->n_crashed = 0; B->status = 0; }
      __syncthreads();
    } else {
      const uint32_t chunk = (n + NT - 1) / NT;
      const uint32_t lo = min(tid * chunk, n), hi = min(lo + chunk, n);
      // which of this thread's ops are crashed candidates: one bit each (a chunk is at most 64 ops, else the slow way)
      const bool small = hi - lo <= 64u;
      uint64_t mine = 0;
      uint32_t cnt = 0;
      if (small) {
        for (uint32_t i = lo; i < hi; i += kBatch) {
          uint32_t rr[kBatch]; uint8_t ff[kBatch]; int32_t aa[kBatch];
#pragma unroll
          for (uint32_t k = 0; k < kBatch; k++) { const bool in = i + k < hi; rr[k] = in ? sc_ret[i + k] : 0u; ff[k] = in ? f[i + k] : (uint8_t)0; aa[k] = in ? a[i + k] : 0; }
#pragma unroll
          for (uint32_t k = 0; k < kBatch; k++)
            if (i + k < hi && rr[k] == kInf && !(ff[k] == TBC_F_READ && aa[k] == TBC_NIL)) mine |= 1ull << (i + k - lo);
        }
        cnt = (uint32_t)__popcll(mine);
      } else {
        for (uint32_t i = lo; i < hi; i++) cnt += sc_ret[i] == kInf && !(f[i] == TBC_F_READ && a[i] == TBC_NIL);
      }
      s_part[tid] = cnt;
      __syncthreads();
      scan_parts(s_part, NT, &s_total);
      __syncthreads();
      if (tid == 0) { B->n_crashed = s_total; B->status = 0; }
      uint32_t run = s_part[tid];
      if (small) {
        while (mine) {
          const uint32_t i = lo + (uint32_t)__builtin_ctzll(mine);
          mine &= mine - 1ull;
          OpRec o; o.op = i; o.f_slot = (uint32_t)f[i] | ((uint32_t)proc[i] << 8); o.a = a[i]; o.b = b[i];
          crashed[run++] = o;
        }
      } else {
        for (uint32_t i = lo; i < hi; i++) if (sc_ret[i] == kInf && !(f[i] == TBC_F_READ && a[i] == TBC_NIL)) {
          OpRec o; o.op = i; o.f_slot = (uint32_t)f[i] | ((uint32_t)proc[i] << 8); o.a = a[i]; o.b = b[i];
          crashed[run++] = o;
        }
      }
      __syncthreads();
    }
    // lookahead records past the last rank: nothing is needed there
    if (A.look) {
      const uint32_t LW = 1 + MW;
      uint64_t* look = A.look + look_off(H->op_off, h, MW);
      for (uint32_t t = R + tid; t < R + kLookPad; t += NT) {
        look[(uint64_t)t * LW] = (uint64_t)(kLookNone << 16 | kLookNone << 24) | (255ull << 32) | (255ull << 40);
        for (uint32_t w = 0; w < MW; w++) look[(uint64_t)t * LW + 1 + w] = 0ull;
      }
    }
  }
}

// ---- the walk: one wavefront per 64 fronts of one history, lane = process slot (+ 64 per mask word)
namespace {
// The call a lane holds, with what the walk asks of it at every front worked out once, when the cursor moves:
// cls bit 0 = a call at all, 1 = live (completes), 2 = crashed and a candidate (not a nil read), 3 = write / cas, 4 = read
struct Cur { uint32_t inv, ret, op, f; int32_t a, b; uint32_t cls, prod; };
__device__ __forceinline__ Cur load_cur(const Rec* r) {
  const Rec x = *r;            // cls / prod come with the record (pack.hip phase 6)
  return Cur{x.inv_rank, x.ret_rank, x.opidx, x.f, x.a, x.b, x.cls, x.prod};
}
}  // namespace

template <int MW>
__global__ __launch_bounds__(256) void open_walk_kernel(PackOpenArgs A) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wid = blockIdx.x * 4u + (threadIdx.x >> 6);
  const uint32_t cph = A.chunks_per_hist;
  const uint32_t hr = wid / cph, c = wid - hr * cph, h = A.h0 + hr;
  if (h >= A.n_hist) return;
  const Hist* H = &A.hist[h];
  const BeamHist* B = &A.bh[h];
  const uint32_t R = H->n_ret;
  const uint32_t F_lo = c * 64u;
  if (H->status != 0 || B->status != 0 || F_lo >= R) return;
  const uint32_t F_hi = min(F_lo + 64u, R);
  const uint32_t W = H->n_slots;
  const Rec* rec = A.rec + H->rec_off;
  const uint32_t* seg = A.seg + H->seg_off;
  const uint32_t* off = A.off + B->off_off;
  OpRec* lst = A.lst + B->lst_off;
  const uint32_t* ret_slot = A.ret_slot + H->ret_off;
  uint64_t* twn = A.twn ? A.twn + B->lst_off * MW : nullptr;
  const uint32_t V = A.vpad;
  const uint32_t FW = A.front_words ? A.front_words : V * MW;            // u64 words per front: a plain row, or a front record
  uint64_t* rdm = A.rdm ? A.rdm + H->op_off * FW : nullptr;
  uint64_t* look = A.look ? A.look + look_off(H->op_off, h, MW) : nullptr;
  uint32_t* tmp = A.tmp ? A.tmp + H->op_off : nullptr;

  // cursors: the first call of each slot that has not completed before F_lo, and the one after it
  Cur cur[MW], nxt[MW];
  uint32_t at[MW], tail[MW];
#pragma unroll
  for (int j = 0; j < MW; j++) {
    const uint32_t p = lane + 64u * (uint32_t)j;
    cur[j] = Cur{kInf, kInf, kInf, kFNone, 0, 0, 0u, kLookNone}; nxt[j] = cur[j]; at[j] = 0; tail[j] = 0;
    if (p < W) {
      uint32_t lo = seg[p] + 1u, hi = seg[p + 1] - 1u;     // [lo, hi): the slot's calls; hi = its tail sentinel
      tail[j] = hi;
      while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (rec[mid].ret_rank >= F_lo) hi = mid; else lo = mid + 1u;
      }
      at[j] = lo;
      cur[j] = load_cur(rec + lo);
      nxt[j] = load_cur(rec + min(lo + 1u, tail[j]));
    }
  }
  const uint32_t w_off = off[min(F_lo + lane, R)];
  const uint32_t w_slot = ret_slot[min(F_lo + lane, R - 1u)];
  const uint64_t below = (1ull << lane) - 1ull;
  uint64_t tw[MW][MW];                  // twin mask of the call each of this lane's registers holds (words over process slots)
  bool known[MW];                       // ... and whether that call has been entered into the masks yet
#pragma unroll
  for (int j = 0; j < MW; j++) {
    known[j] = false;
#pragma unroll
    for (int w = 0; w < MW; w++) tw[j][w] = 0ull;
  }

  for (uint32_t F = F_lo; F < F_hi; F++) {
    const uint32_t base = __builtin_amdgcn_readlane(w_off, F - F_lo);
    const uint32_t px = __builtin_amdgcn_readlane(w_slot, F - F_lo);
    bool live[MW], crashed_open[MW], inl[MW];
    uint64_t occ[MW];
#pragma unroll
    for (int j = 0; j < MW; j++) {
      const bool here = cur[j].inv <= F;            // (a lane without a call holds inv = kInf)
      live[j] = here && (cur[j].cls & 2u);
      crashed_open[j] = here && (cur[j].cls & 4u);
      inl[j] = live[j] && !(A.branch_lists && (cur[j].cls & 16u));      // in the front's list (branch lists: not the reads)
      occ[j] = __ballot(inl[j]);
    }
    // the front's list, in slot order
    uint32_t before = 0, mypos[MW];
#pragma unroll
    for (int j = 0; j < MW; j++) {
      mypos[j] = base + before + (uint32_t)__popcll(occ[j] & below);
      before += (uint32_t)__popcll(occ[j]);
      if (inl[j]) {
        OpRec o; o.op = cur[j].op; o.f_slot = cur[j].f | ((lane + 64u * (uint32_t)j) << 8) | (cur[j].ret == F ? kAtFront : 0u);
        o.a = cur[j].a; o.b = cur[j].b;
        lst[mypos[j]] = o;
      }
    }
    // twin masks, kept from front to front: a call ENTERS the masks once, at the first front it is open at (it is added to
    // the same-effect calls that complete later, and gets the ones that complete earlier as its own mask), and LEAVES them
    // when the front passes its completion (below).  One broadcast per invoked write / cas instead of one per open
    // write / cas per front -- the walk is bound by its vector instructions, and this loop was a third of them.
    if (twn) {
#pragma unroll
      for (int jj = 0; jj < MW; jj++) {
        uint64_t m = __ballot(live[jj] && (cur[jj].cls & 8u) && !known[jj]);
        while (m) {
          const uint32_t l = (uint32_t)__builtin_ctzll(m);
          m &= m - 1ull;
          const uint32_t zf = __builtin_amdgcn_readlane(cur[jj].f, l), zr = __builtin_amdgcn_readlane(cur[jj].ret, l);
          const int32_t za = (int32_t)__builtin_amdgcn_readlane((uint32_t)cur[jj].a, l), zb = (int32_t)__builtin_amdgcn_readlane((uint32_t)cur[jj].b, l);
#pragma unroll
          for (int j = 0; j < MW; j++) {
            const bool same = live[j] && cur[j].f == zf && cur[j].a == za && (zf != TBC_F_CAS || cur[j].b == zb) && !(j == jj && lane == l);
            if (same && zr < cur[j].ret) tw[j][jj] |= 1ull << l;
            const uint64_t earlier = __ballot(same && cur[j].ret < zr);
            if (lane == l) tw[jj][j] = earlier;
          }
        }
      }
#pragma unroll
      for (int j = 0; j < MW; j++) {
        const bool wc = live[j] && (cur[j].cls & 8u);
        if (wc) known[j] = true;
        if (inl[j]) {
#pragma unroll
          for (int w = 0; w < MW; w++) twn[(uint64_t)mypos[j] * MW + w] = wc ? tw[j][w] : 0ull;
        }
      }
    }
    // open-read masks by value: lane vi keeps row entry vi
    if (rdm) {
      // (the open reads are few: each one drops its bit into the lane that keeps its value's entry)
      uint64_t mine[MW];
#pragma unroll
      for (int j = 0; j < MW; j++) mine[j] = 0ull;
#pragma unroll
      for (int j = 0; j < MW; j++) {
        uint64_t rd = __ballot(live[j] && (cur[j].cls & 16u));
        while (rd) {
          const uint32_t l = (uint32_t)__builtin_ctzll(rd);
          rd &= rd - 1ull;
          const int32_t va = (int32_t)__builtin_amdgcn_readlane((uint32_t)cur[j].a, l);
          const uint32_t vi = rdm_index(va, V);
          if (lane == vi && (vi != 0u || va == TBC_NIL)) mine[j] |= 1ull << l;
        }
      }
      if (lane < V) {
#pragma unroll
        for (int j = 0; j < MW; j++) rdm[(uint64_t)F * FW + lane * MW + j] = mine[j];
      }
    }
    // lookahead record of rank F: what the completing call needs / produces, who else open here produces it
    if (look) {
      const uint32_t jx = px >> 6, lx = px & 63u;
      uint32_t xf = kFNone, xinv = 0; int32_t xa = 0, xb = 0;
#pragma unroll
      for (int j = 0; j < MW; j++) if (jx == (uint32_t)j) {
        xf = __builtin_amdgcn_readlane(cur[j].f, lx); xinv = __builtin_amdgcn_readlane(cur[j].inv, lx);
        xa = (int32_t)__builtin_amdgcn_readlane((uint32_t)cur[j].a, lx); xb = (int32_t)__builtin_amdgcn_readlane((uint32_t)cur[j].b, lx);
      }
      const uint32_t need = look_need(xf, xa), prod = look_prod(xf, xa, xb), di = min(F - xinv, 255u);
      const uint64_t w0 = (uint64_t)(px & 0xFFFFu) | (uint64_t)need << 16 | (uint64_t)prod << 24 | (uint64_t)di << 32 | (255ull << 40);
      const uint32_t LW = 1 + MW;
#pragma unroll
      for (int j = 0; j < MW; j++) {
        const bool other = (live[j] || crashed_open[j]) && !(jx == (uint32_t)j && lane == lx);
        const uint64_t pm = need == kLookNone ? 0ull : __ballot(other && cur[j].prod == need);
        if (lane == 0) look[(uint64_t)F * LW + 1 + j] = pm;
      }
      if (lane == 0) { look[(uint64_t)F * LW] = w0; tmp[F] = 255u; }
    }
    // the call completing here leaves: its process's next call takes the lane
#pragma unroll
    for (int j = 0; j < MW; j++) {
      if (twn) {                                    // the completed call leaves every mask
#pragma unroll
        for (int w = 0; w < MW; w++) if ((px >> 6) == (uint32_t)w) tw[j][w] &= ~(1ull << (px & 63u));
      }
      if ((px >> 6) == (uint32_t)j && lane == (px & 63u)) {
        if (twn) {
#pragma unroll
          for (int w = 0; w < MW; w++) tw[j][w] = 0ull;
          known[j] = false;
        }
        cur[j] = nxt[j];
        at[j] = min(at[j] + 1u, tail[j]);
        nxt[j] = load_cur(rec + min(at[j] + 1u, tail[j]));
      }
    }
  }
}

// ---- the same walk with lane = front (open_walk_impl.h): up to 64 process slots, rows of up to VCAP entries
template <int VCAP>
__global__ __launch_bounds__(256, 2) void open_walk_fronts_kernel(PackOpenArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint32_t walk_lds[];
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wv_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  walk::walk_wave<VCAP>(A, blockIdx.x * 4u + wv_, walk_lds + wv_ * walk::walk_lds_words(), lane);
}

// ---- lookahead: how recently another producer of the needed value was invoked
__global__ __launch_bounds__(1024) void open_dprod_kernel(PackOpenArgs A) {
  const uint32_t NT = blockDim.x, tid = threadIdx.x, MW = A.mask_words;
  for (uint32_t h = A.h0 + blockIdx.x; h < A.n_hist; h += gridDim.x) {
    const Hist* H = &A.hist[h];
    const BeamHist* B = &A.bh[h];
    const uint32_t n = H->n_ops, R = H->n_ret;
    if (H->status != 0 || B->status != 0 || R == 0) continue;
    const uint8_t* f = A.f + H->op_off;
    const int32_t* a = A.a + H->op_off;
    const int32_t* b = A.b + H->op_off;
    const uint32_t* sc_inv = A.scratch + H->frame_off;
    const uint32_t* sc_ret = sc_inv + n;
    const uint32_t LW = 1 + MW;
    uint64_t* look = A.look + look_off(H->op_off, h, MW);
    const uint32_t* ret_op = A.ret_op + H->ret_off;
    uint32_t* tmp = A.tmp + H->op_off;
    // F: every producer tells the ranks right after its invocation how recent it is
    for (uint32_t i = tid; i < n; i += NT) {
      const uint32_t pv = look_prod(f[i], a[i], b[i]);
      if (pv == kLookNone || (sc_ret[i] == kInf && f[i] == TBC_F_READ)) continue;
      const uint32_t ir = sc_inv[i];
      uint64_t w[kLookahead]; uint32_t ro[kLookahead];          // the records and completing ops of the next ranks: one trip
#pragma unroll
      for (uint32_t k = 0; k < kLookahead; k++) {
        const uint32_t t = ir + k;
        const bool in = t < R;
        w[k] = in ? ld_agent64(&look[(uint64_t)t * LW]) : 0ull;
        ro[k] = in ? ret_op[t] : 0u;
      }
#pragma unroll
      for (uint32_t k = 0; k < kLookahead; k++) {
        const uint32_t t = ir + k;
        if (t < R && ((uint32_t)(w[k] >> 16) & 0xFFu) == pv && ro[k] != i) atomicMin(&tmp[t], k);
      }
    }
    __syncthreads();
    // G: fold dprod into the records
    for (uint32_t t0 = tid; t0 < R; t0 += 4u * NT) {            // four ranks per trip
      uint32_t d[4]; uint64_t w0[4];
#pragma unroll
      for (uint32_t k = 0; k < 4; k++) {
        const uint32_t t = t0 + k * NT;
        d[k] = t < R ? ld_agent(&tmp[t]) : 255u;
        w0[k] = t < R ? ld_agent64(&look[(uint64_t)t * LW]) : 0ull;
      }
#pragma unroll
      for (uint32_t k = 0; k < 4; k++) {
        const uint32_t t = t0 + k * NT;
        if (t < R && d[k] < 255u) look[(uint64_t)t * LW] = (w0[k] & ~(255ull << 40)) | ((uint64_t)d[k] << 40);
      }
    }
    __syncthreads();
  }
}

// ---- count form (tbc_internal.h, kRuleCount): what the fronts need of the classes of crashed calls.  One workgroup per history:
//   ncr[F]   = classes with a member invoked by front F (the classes are in order of their first invocation)
//   look[t]  bit 48 of word 0: a crashed call that produces the value the call completing at t needs was invoked by then (the
//            lookahead takes such a producer as available whatever the counts: conservative, a dead config is dead)
// phase 0 (before the front walk, whose front records hold the candidate counts): ncr[];  phase 1 (after it): the lookahead bit
__global__ __launch_bounds__(256) void count_fronts_kernel(PackOpenArgs A, uint32_t phase) {
  const uint32_t NT = blockDim.x, tid = threadIdx.x, LW = 1u + A.mask_words;
  for (uint32_t h = A.h0 + blockIdx.x; h < A.n_hist; h += gridDim.x) {
    const Hist* H = &A.hist[h];
    const BeamHist* B = &A.bh[h];
    const uint32_t R = H->n_ret, nc = B->n_classes;
    if (!(H->flags & kHistCount) || H->status != 0 || B->status != 0 || R == 0 || nc == 0) continue;
    const uint64_t* cmem = A.cmem + B->cmem_off;
    const OpRec* cls = reinterpret_cast<const OpRec*>(cmem);          // the class records head the history's block
    uint32_t* ncr = A.ncr + B->off_off;
    uint64_t* look = (A.look && phase == 1u) ? A.look + look_off(H->op_off, h, A.mask_words) : nullptr;
    if (phase == 1u && !look) continue;
    for (uint32_t F = tid; F < R; F += NT) {
      const uint32_t need = look ? (uint32_t)(look[(uint64_t)F * LW] >> 16) & 0xFFu : kLookNone;
      uint32_t avail = 0; bool producer = false;
      for (uint32_t c = 0; c < nc; c++) {
        const OpRec o = cls[c];
        if ((uint32_t)cmem[o.op] > F) break;                          // (first ranks ascend with the class number)
        avail++;
        producer = producer || look_prod(o.f_slot & 0xFFu, o.a, o.b) == need;
      }
      if (phase == 0u) ncr[F] = avail;
      else if (producer && need != kLookNone) look[(uint64_t)F * LW] |= 1ull << 48;
    }
  }
}

// ---- front records (narrow kernel): the part of each record that is not the read masks -- where the front's list is, how
// many calls are open, and the windows of completion slots / read kinds of ranks F .. F + 15.  One thread per front, streaming:
// every input is an array the counts kernel wrote (neighbouring threads read neighbouring words), 48 B written per front.
__global__ __launch_bounds__(256) void front_meta_kernel(PackOpenArgs A) {
  const uint32_t cph = A.chunks_per_hist * 64u;                 // fronts per history at most, rounded up to the walk's chunks
  const uint64_t gid = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  const uint32_t hr = (uint32_t)(gid / cph), F = (uint32_t)(gid - (uint64_t)hr * cph), h = A.h0 + hr;
  if (h >= A.n_hist) return;
  const Hist* H = &A.hist[h];
  const BeamHist* B = &A.bh[h];
  if (H->status != 0 || B->status != 0 || F >= H->n_ret) return;
  const uint32_t* off = A.off + B->off_off;
  const uint32_t o0 = off[F], o1 = off[F + 1u], nc = A.ncr[B->off_off + F];
  const uint8_t* s8 = A.slot8 + slot8_off(H->op_off, h) + F;
  const uint8_t* k8 = A.rk8 + slot8_off(H->op_off, h) + F;
  uint64_t w[4] = {0, 0, 0, 0};
#pragma unroll
  for (uint32_t l = 0; l < 16; l++) {                          // (both arrays are padded 16 past the last rank)
    w[l >> 3] |= (uint64_t)s8[l] << (8u * (l & 7u));
    w[2 + (l >> 3)] |= (uint64_t)k8[l] << (8u * (l & 7u));
  }
  if (A.front_compact) {          // 64 B records: list location in word 6, seven ranks of slot | kind << 6 in word 7
    uint64_t* rec = A.rdm + (H->op_off + F) * kFrontCompactWords;
    uint64_t win = 0;
#pragma unroll
    for (uint32_t l = 0; l < kFrontCompactRanks; l++) {
      const uint32_t sl = (uint32_t)(w[0] >> (8u * l)) & 63u, k = (uint32_t)(w[2] >> (8u * l)) & 0xFFu;
      win |= (uint64_t)(sl | ((k == 0xFFu ? 7u : (k & 7u)) << 6)) << (9u * l);
    }
    rec[6] = (uint64_t)o0 | ((uint64_t)((o1 - o0) & 0xFFu) << 32) | ((uint64_t)(((o1 - o0) + nc) & 0xFFFFFFu) << 40);
    rec[7] = win;
    return;
  }
  uint64_t* rec = A.rdm + (H->op_off + F) * A.front_words + A.vpad * A.mask_words;
  rec[0] = (uint64_t)o0 | ((uint64_t)(o1 - o0) << 32);
  rec[1] = (uint64_t)((o1 - o0) + nc);
  rec[2] = w[0]; rec[3] = w[1]; rec[4] = w[2]; rec[5] = w[3];
}

void launch_open_counts(const PackOpenArgs& a, void* stream) {
  const uint32_t n_here = a.n_hist - a.h0;
  hipLaunchKernelGGL(open_counts_kernel, dim3(n_here < 4096 ? n_here : 4096), dim3(n_here <= 64 ? 1024 : 256), 0, (hipStream_t)stream, a);
}

void launch_pack_open(const PackOpenArgs& a, void* stream, bool skip_counts) {
  hipStream_t s = (hipStream_t)stream;
  const uint32_t n_here = a.n_hist - a.h0;
  const uint32_t grid = n_here < 4096 ? n_here : 4096;
  // few histories: latency matters (tbc_check), give each the widest workgroup; many: occupancy matters
  const uint32_t nt = n_here <= 64 ? 1024 : 256;
  if (!skip_counts) hipLaunchKernelGGL(open_counts_kernel, dim3(grid), dim3(nt), 0, s, a);
  if (a.cmem) hipLaunchKernelGGL(count_fronts_kernel, dim3(grid), dim3(256), 0, s, a, 0u);
  const uint64_t waves = (uint64_t)n_here * a.chunks_per_hist;
  const uint32_t wgrid = (uint32_t)((waves + 3) / 4);
  // one mask word: the walk with lane = front (a fifth of the vector instructions); wider masks take the walk with lane = process slot
  const bool by_front = a.mask_words == 1 && a.vpad <= 32;
  const size_t walk_lds_bytes = 4u * walk::walk_lds_words() * sizeof(uint32_t);
  if (by_front) {
    if (a.vpad <= 8) hipLaunchKernelGGL(open_walk_fronts_kernel<8>, dim3(wgrid), dim3(256), walk_lds_bytes, s, a);
    else hipLaunchKernelGGL(open_walk_fronts_kernel<32>, dim3(wgrid), dim3(256), walk_lds_bytes, s, a);
  } else switch (a.mask_words) {
    case 2: hipLaunchKernelGGL(open_walk_kernel<2>, dim3(wgrid), dim3(256), 0, s, a); break;
    default: hipLaunchKernelGGL(open_walk_kernel<4>, dim3(wgrid), dim3(256), 0, s, a); break;
  }
  // (the walk by front leaves the producer distances in the lookahead records and writes compact front records whole)
  if (a.look && !by_front) hipLaunchKernelGGL(open_dprod_kernel, dim3(grid), dim3(nt), 0, s, a);
  if (a.cmem) hipLaunchKernelGGL(count_fronts_kernel, dim3(grid), dim3(256), 0, s, a, 1u);
  if (a.front_words && !(by_front && a.front_compact)) {
    const uint64_t fronts = (uint64_t)n_here * a.chunks_per_hist * 64u;
    hipLaunchKernelGGL(front_meta_kernel, dim3((uint32_t)((fronts + 255) / 256)), dim3(256), 0, s, a);
  }
}

}  // namespace tbc
