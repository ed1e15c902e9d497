// set_full.hip -- jepsen.checker/set-full's per-element scan on the MI355X (gfx950).
//
// The reference's set-full workload is checked by `(checker/set-full {:linearizable? true})`
// (/root/reference/src/tigerbeetle/workloads/set_full.clj:157; jepsen itself is not in /root/reference --
// the algorithm is recalled and restated in oracle/set_full.py).  Per element three history indices decide
// its outcome: known, last_present, last_absent (include/tbcheck.h).  They are reductions over the
// reads x elements membership matrix:
//   last_present[e] = max read_invoke[r] over reads r that contain e      }  only reads completing after
//   last_absent[e]  = max read_invoke[r] over reads r that do not         }  e's add was invoked count
//   known[e]        = min(add_ok[e], min read_ok[r] over reads containing e)
// This is the one streaming kernel of the path: the matrix is n_reads x n_elements bits (2 GB at 65,536 reads
// of 262,144 elements) and is read about half once.
//
// Layout: rows = reads in invocation order, words_per_row 32-bit words each; a wavefront's 64 lanes take 64
// consecutive words of a row (256 B coalesced).  Two passes, no atomics:
//   setfull_any_kernel      grid = word columns x chunks of rows: a thread ORs "present" and "absent" over FOUR word
//                           columns (128 elements, 16 B loads) of ONE chunk, four rows in flight -- the streaming
//                           pass, coalesced words written per thread (one column, eight rows when a row is not a multiple of 16 B);
//   setfull_resolve_kernel  one thread per word column: the chunk summaries say WHICH chunk holds each element's last
//                           present / last absent / first present read; only those chunks are walked again, bit-parallel
//                           (a 32-bit "still wanted" mask per direction), and the thread writes its own 32 results.
// Elements are numbered by add invocation, so "reads completing after e's add was invoked" is a PREFIX of the elements
// for each row (p[r], one binary search per row in setfull_prefix_kernel) and a chunk whose reads all precede a
// column's adds is skipped without a load: the all-zero triangle below the diagonal is never read.
// known: the first read (in invocation order) containing e need not be the first to complete, so the walk keeps
// offering later rows' read_ok while they were invoked before the latest first-completion seen -- a window bounded by
// the number of concurrent readers.
#include <hip/hip_runtime.h>
#include <vector>
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <new>
#include "tbc_internal.h"

using namespace tbc;

namespace {

constexpr uint32_t kNoneU = 0xFFFFFFFFu;
constexpr uint32_t kSetFullRows = 2048;      // rows per chunk at most (their metadata is staged in LDS)
constexpr uint32_t kWordCounters = 256;      // the words-loaded statistic: a wavefront adds to counter (its workgroup mod 256), 128 B apart -- thousands of
                                             // atomics on ONE address queue up in one L2 channel; the host adds the counters up

__global__ __launch_bounds__(256) void setfull_prefix_kernel(const uint32_t* add_invoke, const uint32_t* read_ok, uint32_t E, uint32_t R,
                                                             uint32_t rows_per_chunk, uint32_t chunks, uint32_t* P, uint32_t* pmax) {
  const uint32_t r = blockIdx.x * 256u + threadIdx.x;
  if (r >= R) return;
  const uint32_t t = read_ok[r];
  uint32_t lo = 0, hi = E;                       // first element whose add was invoked at or after this read's completion
  while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (add_invoke[mid] < t) lo = mid + 1; else hi = mid; }
  P[r] = lo;
  atomicMax(&pmax[r / rows_per_chunk], lo);
  atomicMin(&pmax[chunks + r / rows_per_chunk], lo);         // (the minima lie behind the maxima)
}

__device__ __forceinline__ uint32_t prefix_mask(uint32_t p, uint32_t w) {       // bits of word w below element number p
  return p >= 32u * w + 32u ? 0xFFFFFFFFu : (p <= 32u * w ? 0u : (1u << (p - 32u * w)) - 1u);
}

// ---- the membership matrix from the reads' compact form (tbc_setfull_create_rows): one workgroup per read writes its row --
// ones below top[r], zeros above (a coalesced stream: the matrix is written once, at HBM's write rate) -- and then flips the
// listed exceptions in it.  No row ever exists on the host.
__global__ __launch_bounds__(256) void setfull_rows_kernel(const uint32_t* __restrict__ top, const unsigned long long* __restrict__ exc_off,
                                                           const uint32_t* __restrict__ exc, uint32_t R, uint32_t WPR, uint32_t PITCH, uint32_t* __restrict__ M) {
  for (uint32_t r = blockIdx.x; r < R; r += gridDim.x) {
    uint32_t* row = M + (uint64_t)r * PITCH;
    const uint32_t t = top[r];
    for (uint32_t w = threadIdx.x; w < WPR; w += 256u) row[w] = prefix_mask(t, w);
    __syncthreads();
    const unsigned long long e0 = exc_off[r], e1 = exc_off[r + 1];
    for (unsigned long long i = e0 + threadIdx.x; i < e1; i += 256u) {
      const uint32_t e = exc[i];
      atomicXor(&row[e >> 5], 1u << (e & 31u));
    }
    __syncthreads();
  }
}

// ---- pass 1: per (word column, chunk of rows) -- is any bit of the column present / absent in the chunk?  The streaming
// pass.  VEC = 4: a lane takes FOUR consecutive word columns (16 B loads, a wavefront 1 KB of a row; rows of a multiple of four
// words), VEC = 1: one column; eight rows are requested, then folded.  Nothing a load returns decides whether the next is issued (a
// chunk is at most 2,048 rows: stopping at a saturated column saved nothing on set-full's matrices, where an element is absent before
// its add and present after, and cost a round trip every eight rows); no atomics.  Three kinds of tile (256 VEC columns x a chunk):
// below the diagonal (return), on it (the general path: the chunk's row metadata in LDS, a mask per row and word), above it (the lean
// path: every column counts in every row).
// Round 6 (profiles/r06_setfull_*, scripts/exp/strip_read.hip = this access pattern, bare: 6.1 TB/s a rectangle, 5.3 the triangle):
// 0.16 -> 0.105 ms.  What it was NOT: bytes in flight (4, 8 or 16 rows a lane read alike), the row pitch (a power of two, padded: the
// same), the per-row masks (the lean path alone: 3 %).  What it was: the summaries' scattered 4 B stores (43 us, below) and the XCDs
// (column block j in blockIdx.x was XCD j: 7 %).
// The summaries lie CHUNK-major ([chunk][word column], SP words a chunk, SP a multiple of four): this pass writes a thread's four
// columns as one 16 B store, a wavefront 1 KB.  Rounds 2 - 5 kept them column-major for pass 2's sake (a column's chunks as one line)
// and paid for it here without knowing: 4 B stores 1 KB apart, every one a masked write of its own into a line that sixteen
// workgroups on eight XCDs write -- 43 us of this kernel's 150 (scripts/exp/strip_read.hip: the bare triangle 0.114 ms, with the
// column-major stores 0.157, with these 0.117).  Pass 2 reads a column's chunks as 64 words of 64 lines now, and is given its
// columns so that the workgroups of one XCD share those lines (setfull_resolve_kernel).
template <int VEC>
__device__ __forceinline__ void store_summary(uint32_t* __restrict__ any_p, uint32_t* __restrict__ any_a, uint32_t SP, uint32_t c, uint32_t w0,
                                              const uint32_t (&pa)[VEC], const uint32_t (&aa)[VEC]) {
  const uint64_t at = (uint64_t)c * SP + w0;
  if constexpr (VEC == 4) {
    *reinterpret_cast<uint4*>(any_p + at) = make_uint4(pa[0], pa[1], pa[2], pa[3]);
    *reinterpret_cast<uint4*>(any_a + at) = make_uint4(aa[0], aa[1], aa[2], aa[3]);
  } else {
#pragma unroll
    for (int v = 0; v < VEC; v++) { any_p[at + (uint32_t)v] = pa[v]; any_a[at + (uint32_t)v] = aa[v]; }
  }
}

template <int VEC>
__global__ __launch_bounds__(256) void setfull_any_kernel(const uint32_t* __restrict__ M, const uint32_t* __restrict__ P,
                                                          const uint32_t* __restrict__ pmax, uint32_t E, uint32_t R, uint32_t WPR, uint32_t PITCH,
                                                          uint32_t rows_per_chunk, uint32_t SP, uint32_t* __restrict__ any_p,
                                                          uint32_t* __restrict__ any_a, unsigned long long* words_loaded) {
  constexpr uint32_t U = 8u;                            // rows in flight per lane (16 B each at VEC = 4)
  // grid = (chunks, column blocks), the CHUNK in x.  Workgroup b runs on XCD b % 8 (observed, MI355X_MICROARCH.md "Workgroup dispatch"),
  // and the work is a triangle: column block j counts in the chunks above j / n of the rows only.  With the column block in x (rounds
  // 3 - 5: eight of them for 262,144 elements, i.e. column block j WAS XCD j) XCD 0 streamed 256 chunks and XCD 7 32 -- the kernel
  // lasted as long as XCD 0's 22 % of the matrix through one XCD's fabric port.  With the chunk in x every XCD gets every eighth
  // chunk of every column block.
  // The order the workgroups are handed out in (x fastest, then y): the LAST column block first, and in every column block the chunks
  // from its diagonal on -- the tiles on the diagonal decide per row (the general path below: the longest workgroups) and start first,
  // the tiles below the diagonal, which return at once, come last.
  const uint32_t cb = gridDim.y - 1u - blockIdx.y;
  const uint32_t c = (blockIdx.x + (uint32_t)((uint64_t)cb * gridDim.x / gridDim.y)) % gridDim.x;
  const uint32_t w0 = (cb * 256u + threadIdx.x) * (uint32_t)VEC;
  uint32_t loaded = 0;
  __shared__ uint32_t s_P[kSetFullRows];
  const uint32_t r0 = min(c * rows_per_chunk, R), r1 = min(r0 + rows_per_chunk, R);     // (a trailing chunk may be empty)
  // the whole workgroup lies below the diagonal: nothing to stage, nothing to read, nothing to write (which workgroups these are is
  // fixed with the object -- P is -- and tbc_setfull_create zeroed the summaries)
  if (pmax[c] <= 32u * (cb * 256u * (uint32_t)VEC)) return;
  // A workgroup whose columns ALL count in EVERY row of its chunk (the chunk's least prefix reaches past its last column: four tiles in
  // five of set-full's triangle) has nothing to decide per row: no row metadata, no masks, eight 16 B loads a lane and two instructions
  // a word -- the fold of the general path below costs ~50 vector instructions a row and wavefront, 43 us of a SIMD's time per launch
  // beside 88 us of streaming (scripts/exp/strip_read.hip: this very access pattern, bare, reads at 6.1 TB/s, 6.7 non-temporal).
  const uint32_t tile_hi = 32u * ((cb + 1u) * 256u * (uint32_t)VEC);
  if (pmax[gridDim.x + c] >= (tile_hi < E ? tile_hi : E)) {
    if (w0 < WPR) {
      typedef uint32_t wvec __attribute__((ext_vector_type(VEC)));
      uint32_t po[VEC], na[VEC];
#pragma unroll
      for (int v = 0; v < VEC; v++) { po[v] = 0u; na[v] = 0xFFFFFFFFu; }
      const uint32_t* src = M + (uint64_t)r0 * PITCH + w0;
      uint32_t r = r0;
      constexpr uint32_t UL = 8u;
      for (; r + UL <= r1; r += UL, src += (uint64_t)UL * PITCH) {
        wvec x[UL];
#pragma unroll
        for (uint32_t q = 0; q < UL; q++) x[q] = __builtin_nontemporal_load(reinterpret_cast<const wvec*>(src + (uint64_t)q * PITCH));
#pragma unroll
        for (uint32_t q = 0; q < UL; q++)
#pragma unroll
          for (int v = 0; v < VEC; v++) { po[v] |= x[q][v]; na[v] &= x[q][v]; }
      }
      for (; r < r1; r++, src += PITCH) {
        const wvec x = __builtin_nontemporal_load(reinterpret_cast<const wvec*>(src));
#pragma unroll
        for (int v = 0; v < VEC; v++) { po[v] |= x[v]; na[v] &= x[v]; }
      }
      uint32_t pa[VEC], aa[VEC];
#pragma unroll
      for (int v = 0; v < VEC; v++) {
        const uint32_t lo = 32u * (w0 + (uint32_t)v);
        const uint32_t full = lo >= E ? 0u : (E - lo >= 32u ? 0xFFFFFFFFu : (1u << (E - lo)) - 1u);
        pa[v] = po[v] & full; aa[v] = ~na[v] & full;
        loaded += full ? r1 - r0 : 0u;
      }
      store_summary<VEC>(any_p, any_a, SP, c, w0, pa, aa);
    }
    unsigned long long tot = loaded;
    for (int d = 32; d >= 1; d >>= 1) tot += __shfl_xor(tot, d);
    if ((threadIdx.x & 63u) == 0 && tot) atomicAdd(words_loaded + 16u * ((blockIdx.x + blockIdx.y * gridDim.x) % kWordCounters), tot);
    return;
  }
  for (uint32_t i = threadIdx.x; i < r1 - r0; i += 256u) s_P[i] = P[r0 + i];
  __syncthreads();
  if (w0 < WPR) {
    uint32_t pa[VEC], aa[VEC], full[VEC];
#pragma unroll
    for (int v = 0; v < VEC; v++) {
      pa[v] = 0u; aa[v] = 0u;
      const uint32_t lo = 32u * (w0 + (uint32_t)v);
      full[v] = lo >= E ? 0u : (E - lo >= 32u ? 0xFFFFFFFFu : (1u << (E - lo)) - 1u);
    }
    if (pmax[c] > 32u * w0) {
      // rows [hi - U, hi) of the chunk (latest first; fewer at its start): requested ...
      const auto fetch = [&](uint32_t hi, uint32_t (&wd)[U][VEC], uint32_t (&pr)[U]) {
        const uint32_t lo = hi - r0 >= U ? hi - U : r0;
#pragma unroll
        for (uint32_t q = 0; q < U; q++) {
          const uint32_t r = hi - 1u - q;
          const bool in = hi - lo > q;
          const uint32_t pv = s_P[in ? r - r0 : 0u];                    // (read either way: no branch between the loads)
          pr[q] = in ? pv : 0u;                                         // (0: no element of the row counts)
          const bool need = pr[q] > 32u * w0 && full[0] != 0u;
          // a row that does not count is not fetched -- but by an ADDRESS, not a branch: such a lane reads row 0's words (one hot line
          // for the whole grid; the fold masks them out, vm = 0).  A branch around every load hides from the compiler how many loads
          // are outstanding, and it then waits for ALL of them (s_waitcnt vmcnt(0)) before the fold: the rows requested ahead would
          // be waited for at once, i.e. nothing would be ahead
          const uint32_t* src = M + (need ? (uint64_t)r * PITCH : 0ull) + w0;
          if constexpr (VEC == 4) {
            const uint4 x = *reinterpret_cast<const uint4*>(src);
            wd[q][0] = x.x; wd[q][1] = x.y; wd[q][2] = x.z; wd[q][3] = x.w;
          } else {
            wd[q][0] = *src;
          }
        }
      };
      // ... and folded into the column's two words
      const auto fold = [&](const uint32_t (&wd)[U][VEC], const uint32_t (&pr)[U]) {
#pragma unroll
        for (uint32_t q = 0; q < U; q++) {
#pragma unroll
          for (int v = 0; v < VEC; v++) {
            const uint32_t vm = prefix_mask(pr[q], w0 + (uint32_t)v) & full[v];
            pa[v] |= wd[q][v] & vm; aa[v] |= ~wd[q][v] & vm; loaded += vm ? 1u : 0u;
          }
        }
      };
      // U rows requested, then folded (scripts/exp/strip_read.hip: at this occupancy a second set of registers on its way while the first
      // is folded reads no faster than this, and a workgroup's chain of waits is half as long with eight rows a wait as with four)
      uint32_t wa[U][VEC], pra[U];
      uint32_t hi = r1;
      while (hi > r0) {
        fetch(hi, wa, pra);
        hi = hi - r0 >= U ? hi - U : r0;
        fold(wa, pra);
      }
    }
    store_summary<VEC>(any_p, any_a, SP, c, w0, pa, aa);
  }
  unsigned long long tot = loaded;
  for (int d = 32; d >= 1; d >>= 1) tot += __shfl_xor(tot, d);
  if ((threadIdx.x & 63u) == 0 && tot) atomicAdd(words_loaded + 16u * ((blockIdx.x + blockIdx.y * gridDim.x) % kWordCounters), tot);
}

// ---- pass 2: one WAVEFRONT per word column resolves its 32 elements.  The chunk summaries say WHICH chunk holds an
// element's last present / last absent / first present read: lanes hold the summaries of 64 chunks each, a ballot finds
// the deciding chunk, and that chunk is walked again with lane = row (64 rows loaded at once, one ballot per wanted bit
// finds the row).  No atomics, no serial chain of loads.  Lane b < 32 keeps element b's three results in registers and writes
// them once, in their final form (no index: TBC_NO_OP; known: the earlier of the add's ack and the first read that held it).
__device__ __forceinline__ uint32_t wave_min_u32_all(uint32_t v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, d));
  return v;
}

// 64 rows of one word column requested AHEAD of the walk that uses them (lane l = row base + l < hi): the column's word, the row's
// prefix and read_invoke (read_ok for the walk of `known`).  The three indices of a column are three chains of dependent trips; the
// first trip of each is known as soon as the summaries are, and the three go out together.
struct RowsAhead { uint32_t base, hi, word, pv, third; };
__device__ __forceinline__ RowsAhead rows_ahead(const uint32_t* __restrict__ M, const uint32_t* __restrict__ P, const uint32_t* __restrict__ third,
                                                uint32_t PITCH, uint32_t w, uint32_t base, uint32_t hi, uint32_t lane) {
  RowsAhead a; a.base = base; a.hi = hi;
  const uint32_t r = base + lane;
  const bool in = r < hi;
  a.word = in ? M[(uint64_t)r * PITCH + w] : 0u; a.pv = in ? P[r] : 0u; a.third = in ? third[r] : 0u;
  return a;
}

// walk chunk c from its latest row down, lane = row: for every bit of `want` find the latest row where the bit is present
// (present = true) or absent; lane b keeps read_invoke + 1 of bit b's row in `res`; returns the bits found
__device__ __forceinline__ uint32_t setfull_last_in_chunk(const uint32_t* __restrict__ M, const uint32_t* __restrict__ P,
                                                          const uint32_t* __restrict__ read_invoke, uint32_t PITCH, uint32_t w, uint32_t full,
                                                          uint32_t r0, uint32_t r1, uint32_t want, bool present, uint32_t& res, uint32_t lane,
                                                          uint32_t& loaded, const RowsAhead& ah) {
  uint32_t found = 0;
  if (!(r1 > r0)) return found;
  // the rows of a step are requested while the step before is resolved (the first: with the summaries, if the caller did); word, prefix
  // and read_invoke of a row come in ONE trip (not waiting for P[r] to say whether the row counts)
  RowsAhead cur = ah;
  {
    const uint32_t base = r1 - r0 > 64u ? r1 - 64u : r0;
    if (!(ah.base == base && ah.hi == r1)) cur = rows_ahead(M, P, read_invoke, PITCH, w, base, r1, lane);
  }
  for (uint32_t hi = r1; hi > r0 && (want & ~found); hi = hi - r0 > 64u ? hi - 64u : r0) {
    const uint32_t base = hi - r0 > 64u ? hi - 64u : r0;       // rows [base, hi), lane l = row base + l
    const uint32_t r = base + lane;
    const bool in = r < hi;
    RowsAhead nxt = cur;
    if (base > r0) nxt = rows_ahead(M, P, read_invoke, PITCH, w, base - r0 > 64u ? base - 64u : r0, base, lane);
    const uint32_t word = cur.word;
    const uint32_t valid = in ? prefix_mask(cur.pv, w) & full : 0u;
    const uint32_t inv1 = in ? cur.third + 1u : 0u;
    cur = nxt;
    loaded += valid ? 1u : 0u;
    const uint32_t x = (present ? word : ~word) & valid;
    // row by row, not bit by bit: the latest row that has ANY wanted bit settles all the bits it has (ten instructions), and a group's
    // wanted bits mostly sit in one or two rows -- the last read holds every element but the lost ones; the row before an add lacks all
    // the later elements.  (Rounds 2 - 5 took the bits one at a time, ten instructions each: 2,570 vector instructions a wavefront,
    // 36 us of a SIMD's issue in a 57 us kernel -- profiles/r06_setfull_pmc.txt.)  Never more turns than wanted bits.
    uint32_t todo = want & ~found;
    uint64_t rows = __ballot((x & todo) != 0u);
    while (rows) {
      const uint32_t l = 63u - (uint32_t)__builtin_clzll(rows);
      const uint32_t xl = (uint32_t)__builtin_amdgcn_readlane((int)x, l) & todo;
      const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)inv1, l);
      if (lane < 32u && ((xl >> lane) & 1u)) res = v;
      found |= xl; todo &= ~xl;
      rows = __ballot((x & todo) != 0u) & ((1ull << l) - 1ull);
    }
  }
  return found;
}

#ifndef SF_RESOLVE_MIN_WAVES
#define SF_RESOLVE_MIN_WAVES 8          /* 41 registers: every one of the 8,192 wavefronts of 262,144 elements resident at once */
#endif
__global__ __launch_bounds__(256, SF_RESOLVE_MIN_WAVES) void setfull_resolve_kernel(const uint32_t* __restrict__ M, const uint32_t* __restrict__ P,
                                                              const uint32_t* __restrict__ read_invoke, const uint32_t* __restrict__ read_ok,
                                                              const uint32_t* __restrict__ any_p, const uint32_t* __restrict__ any_a,
                                                              uint32_t E, uint32_t R, uint32_t WPR, uint32_t PITCH, uint32_t rows_per_chunk, uint32_t chunks, uint32_t SP,
                                                              const uint32_t* __restrict__ add_ok, uint32_t* lp, uint32_t* la, uint32_t* known,
                                                              unsigned long long* words_loaded) {
  const uint32_t lane = threadIdx.x & 63u;
  // workgroup b runs on XCD b % 8 (observed): the eight XCDs take eight contiguous ranges of the columns, so that the lines four
  // neighbouring workgroups read 16 B each of -- the summaries' and the matrix rows' -- are fetched into ONE L2 instead of eight
  const uint32_t nb = gridDim.x;
  const uint32_t bb = nb % 8u == 0u ? (blockIdx.x % 8u) * (nb / 8u) + blockIdx.x / 8u : blockIdx.x;
  const uint32_t wv_ = threadIdx.x >> 6;
  const uint32_t w = __builtin_amdgcn_readfirstlane(bb * 4u + wv_);
  uint32_t loaded = 0;
  // the chunk summaries, 64 chunks (one per lane) at a time; any number of chunks
  const uint32_t G = (chunks + 63u) / 64u;
  // up to 256 chunks (what tbc_setfull_create makes of up to 524,288 reads): the summaries of the workgroup's FOUR columns are fetched
  // once, thread t the 16 B of chunk t (chunk-major: its four columns lie side by side), and handed to the four wavefronts through LDS
  // -- a wavefront reading its own column's 256 words asked for 8 x 64 lines, half of all the lines this kernel asks its L1 for
  const bool pre = G <= 4u;
  __shared__ uint4 s_sp[256], s_sa[256];
  if (pre) {
    const uint32_t cl = threadIdx.x;
    const bool in = cl < chunks && bb * 4u < WPR;                     // (SP is a multiple of four: the 16 B lie inside the chunk's row)
    s_sp[cl] = in ? *reinterpret_cast<const uint4*>(any_p + (uint64_t)cl * SP + bb * 4u) : make_uint4(0u, 0u, 0u, 0u);
    s_sa[cl] = in ? *reinterpret_cast<const uint4*>(any_a + (uint64_t)cl * SP + bb * 4u) : make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
  }
  if (w < WPR) {
    const uint32_t full = E - 32u * w >= 32u ? 0xFFFFFFFFu : (1u << (E - 32u * w)) - 1u;
    uint32_t sp[4], sa[4];
#pragma unroll
    for (uint32_t k = 0; k < 4u; k++) {
      const uint32_t cl = lane + 64u * k;
      sp[k] = pre ? reinterpret_cast<const uint32_t*>(&s_sp[cl])[wv_] : 0u;
      sa[k] = pre ? reinterpret_cast<const uint32_t*>(&s_sa[cl])[wv_] : 0u;
    }
    const auto pick = [](const uint32_t (&r)[4], uint32_t gi) -> uint32_t { return gi == 0u ? r[0] : gi == 1u ? r[1] : gi == 2u ? r[2] : r[3]; };
    // the first trip of each of the three walks below, requested now (up to 256 chunks: the summaries are in registers)
    RowsAhead ah_p = {kNoneU, 0u, 0u, 0u, 0u}, ah_a = ah_p, ah_k = ah_p;
    if (pre) {
      uint32_t top_p = kNoneU, top_a = kNoneU, low_p = kNoneU;
#pragma unroll
      for (uint32_t k = 0; k < 4u; k++) {
        const uint64_t bp = __ballot((sp[k] & full) != 0u), ba = __ballot((sa[k] & full) != 0u);
        if (bp) { top_p = 63u - (uint32_t)__builtin_clzll(bp) + 64u * k; if (low_p == kNoneU) low_p = (uint32_t)__builtin_ctzll(bp) + 64u * k; }
        if (ba) top_a = 63u - (uint32_t)__builtin_clzll(ba) + 64u * k;
      }
      const auto top_rows = [&](uint32_t c, uint32_t& base, uint32_t& hi) {
        const uint32_t r0 = min(c * rows_per_chunk, R); hi = min(r0 + rows_per_chunk, R); base = hi - r0 > 64u ? hi - 64u : r0;
      };
      uint32_t b_, h_;
      if (top_p != kNoneU) { top_rows(top_p, b_, h_); ah_p = rows_ahead(M, P, read_invoke, PITCH, w, b_, h_, lane); }
      if (top_a != kNoneU) { top_rows(top_a, b_, h_); ah_a = rows_ahead(M, P, read_invoke, PITCH, w, b_, h_, lane); }
      if (low_p != kNoneU) { b_ = low_p * rows_per_chunk; ah_k = rows_ahead(M, P, read_ok, PITCH, w, b_, min(b_ + 64u, R), lane); }
    }
    // last present / last absent: the latest chunk that has the bit decides; inside it, the latest row
    uint32_t res_p = 0u, res_a = 0u;          // read_invoke + 1 of the element's last present / last absent read, 0 = none
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
      const uint32_t* __restrict__ any = pass == 0 ? any_p : any_a;
      uint32_t& res = pass == 0 ? res_p : res_a;
      uint32_t need = full;
      for (uint32_t gi = G; gi-- > 0 && need;) {
        const uint32_t cl = lane + 64u * gi;
        const uint32_t mine = pre ? pick(pass == 0 ? sp : sa, gi) : (cl < chunks ? any[(uint64_t)cl * SP + w] : 0u);
        uint64_t cand = __ballot((mine & need) != 0u);
        while (cand && need) {
          const uint32_t l = 63u - (uint32_t)__builtin_clzll(cand);
          const uint32_t c = l + 64u * gi;
          const uint32_t mc = (uint32_t)__builtin_amdgcn_readlane((int)mine, l) & need;
          const uint32_t r0 = min(c * rows_per_chunk, R), r1 = min(r0 + rows_per_chunk, R);
          (void)setfull_last_in_chunk(M, P, read_invoke, PITCH, w, full, r0, r1, mc, pass == 0, res, lane, loaded, pass == 0 ? ah_p : ah_a);
          need &= ~mc;
          cand = __ballot((mine & need) != 0u) & ((1ull << l) - 1ull);
        }
      }
    }
    // known: min read_ok over the reads containing the element.  The earliest chunk that has a bit of the column holds
    // the first containing read (in invocation order) of some element; a read invoked before that one completed may still
    // complete earlier, so the walk goes on, 64 rows at a time, while rows were invoked before `until` (the latest first
    // completion seen).
    uint32_t ever = 0, first_c = chunks;
    for (uint32_t gi = 0; gi < G; gi++) {
      const uint32_t cl = lane + 64u * gi;
      uint32_t o = (pre ? pick(sp, gi) : (cl < chunks ? any_p[(uint64_t)cl * SP + w] : 0u)) & full;
      const uint64_t bl = __ballot(o != 0u);
      if (bl && first_c == chunks) first_c = (uint32_t)__builtin_ctzll(bl) + 64u * gi;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) o |= (uint32_t)__shfl_xor((int)o, d);
      ever |= o;
    }
    uint32_t best = 0xFFFFFFFFu;                                // lane b < 32 keeps element b's minimum
    if (ever) {
      uint32_t seen = 0, until = 0;
      // a step's rows (word, prefix, read_ok -- and read_invoke, which says whether the walk goes on) are requested while the step before is
      // resolved: one trip a step where there were two (read_invoke first, the rest only if the walk went on)
      const uint32_t base0 = first_c * rows_per_chunk;
      RowsAhead cur = ah_k.base == base0 ? ah_k : rows_ahead(M, P, read_ok, PITCH, w, base0, min(base0 + 64u, R), lane);
      uint32_t cur_inv = base0 + lane < R ? read_invoke[base0 + lane] : 0xFFFFFFFFu;
      for (uint32_t base = base0; base < R; base += 64u) {
        const uint32_t r = base + lane;
        const bool in = r < R;
        const uint32_t inv_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur_inv);
        if (seen == ever && inv_first > until) break;
        RowsAhead nxt = cur; uint32_t nxt_inv = 0xFFFFFFFFu;
        if (base + 64u < R) {
          nxt = rows_ahead(M, P, read_ok, PITCH, w, base + 64u, min(base + 128u, R), lane);
          nxt_inv = base + 64u + lane < R ? read_invoke[base + 64u + lane] : 0xFFFFFFFFu;
        }
        const uint32_t word = cur.word;
        const uint32_t valid = in ? prefix_mask(cur.pv, w) & full : 0u;
        const uint32_t ok = in ? cur.third : 0xFFFFFFFFu;
        cur = nxt; cur_inv = nxt_inv;
        loaded += valid ? 1u : 0u;
        const uint32_t hits = word & valid;
        uint32_t any_hits = hits;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) any_hits |= (uint32_t)__shfl_xor((int)any_hits, d);
        // row by row here too: the row that completes first among those holding a still-open bit gives its read_ok to every open bit
        // it holds; a turn or two settle nearly every group
        uint32_t open_ = any_hits;
        for (int turn = 0; turn < 8 && open_; turn++) {
          const uint32_t m = wave_min_u32_all((hits & open_) ? ok : 0xFFFFFFFFu);
          const uint64_t who = __ballot((hits & open_) != 0u && ok == m);
          const uint32_t xl = (uint32_t)__builtin_amdgcn_readlane((int)hits, (uint32_t)__builtin_ctzll(who)) & open_;
          if (lane < 32u && ((xl >> lane) & 1u)) best = min(best, m);
          if (xl & ~seen) until = max(until, m);              // (the first batch's minimum bounds the first containing read's completion)
          open_ &= ~xl;
        }
        while (open_) {                                         // (what eight turns leave -- hardly ever anything -- bit by bit)
          const uint32_t b = (uint32_t)__builtin_ctz(open_);
          open_ &= open_ - 1u;
          const uint32_t m = wave_min_u32_all(((hits >> b) & 1u) ? ok : 0xFFFFFFFFu);
          if (lane == b) best = min(best, m);
          if (!((seen >> b) & 1u)) until = max(until, m);
        }
        seen |= any_hits;
      }
    }
    if (lane < 32u) {                                           // (the arrays are padded to whole word columns)
      const uint32_t e = 32u * w + lane;
      lp[e] = res_p ? res_p - 1u : kNoneU;
      la[e] = res_a ? res_a - 1u : kNoneU;
      known[e] = min(best, e < E ? add_ok[e] : kNoneU);
    }
  }
  unsigned long long tot = loaded;
  for (int d = 32; d >= 1; d >>= 1) tot += __shfl_xor(tot, d);
  if ((threadIdx.x & 63u) == 0 && tot) atomicAdd(words_loaded + 16u * (blockIdx.x % kWordCounters), tot);
}

__global__ __launch_bounds__(256) void setfull_finish_kernel(uint32_t* lp1, uint32_t* la1, uint32_t* known, const uint32_t* add_ok, uint32_t E) {
  const uint32_t e = blockIdx.x * 256u + threadIdx.x;
  if (e >= E) return;
  lp1[e] = lp1[e] ? lp1[e] - 1u : kNoneU;
  la1[e] = la1[e] ? la1[e] - 1u : kNoneU;
  known[e] = min(known[e], add_ok[e]);
}

#define SF_TRY(expr)                                                                         \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);  \
      return e_ == hipErrorOutOfMemory ? TBC_ERR_OOM : TBC_ERR_HIP;                          \
    }                                                                                        \
  } while (0)

}  // namespace

struct tbc_setfull {
  int device = 0;
  uint32_t E = 0, R = 0, WPR = 0, PITCH = 0, chunks = 1, rows_per_chunk = 1, chp = 4;       // chp: words per chunk of the summaries (words_per_row rounded up to four)
  uint32_t *d_add_invoke = nullptr, *d_add_ok = nullptr, *d_read_invoke = nullptr, *d_read_ok = nullptr, *d_M = nullptr;
  uint32_t *d_P = nullptr, *d_pmax = nullptr, *d_lp = nullptr, *d_la = nullptr, *d_known = nullptr, *d_anyp = nullptr, *d_anya = nullptr;
  unsigned long long* d_words = nullptr;
  void* arena = nullptr;            // ONE allocation holds every array above (and create_rows' compact reads): one hipMalloc, one hipFree
  unsigned long long h_words[kWordCounters * 16] = {};
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  ~tbc_setfull() {
    if (arena) (void)hipFree(arena);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    if (stream) (void)hipStreamDestroy(stream);
  }
};

static tbc_status setfull_create_impl(const tbc_setfull_in* in, tbc_setfull* S, const tbc_setfull_rows* rows = nullptr) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || (int)in->device >= ndev) {
    set_error("no usable HIP device; libtbcheck has no CPU fallback");
    return TBC_ERR_NO_DEVICE;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, (int)in->device) != hipSuccess || std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    set_error("device %u is not a gfx950 (MI355X) device", in->device);
    return TBC_ERR_NO_DEVICE;
  }
  S->device = (int)in->device; S->E = in->n_elements; S->R = in->n_reads; S->WPR = in->words_per_row;
  if ((uint64_t)S->WPR * 32 < S->E) { set_error("tbc_setfull: words_per_row too small for n_elements"); return TBC_ERR_INVALID_ARG; }
  if (rows) {          // the compact form: every offset and element number is checked here, the kernel trusts them
    if (rows->exc_off[0] != 0) { set_error("tbc_setfull_create_rows: exc_off[0] must be 0"); return TBC_ERR_INVALID_ARG; }
    for (uint32_t r = 0; r < S->R; r++) {
      if (rows->top[r] > S->E || rows->exc_off[r + 1] < rows->exc_off[r]) { set_error("tbc_setfull_create_rows: read %u: bad top / exc_off", r); return TBC_ERR_INVALID_ARG; }
    }
    const uint64_t ne = rows->exc_off[S->R];
    for (uint64_t i = 0; i < ne; i++) if (rows->exc[i] >= S->E) { set_error("tbc_setfull_create_rows: exception %llu names element %u of %u", (unsigned long long)i, rows->exc[i], S->E); return TBC_ERR_INVALID_ARG; }
    // each element at most once per read: the kernel FLIPS the listed bits, a duplicate would flip one back silently.  A strictly ascending
    // list (what the in-repo encoders write) is seen to be duplicate-free in one pass; a list in any other order is sorted aside and looked
    // at again -- round 5 refused every list that was not ascending, which a caller of the earlier header (any order, no duplicates) had
    // no way to know at the same TBC_ABI_VERSION
    std::vector<uint32_t> tmp;
    for (uint32_t r = 0; r < S->R; r++) {
      bool ascending = true;
      for (uint64_t i = rows->exc_off[r] + 1; i < rows->exc_off[r + 1] && ascending; i++) ascending = rows->exc[i] > rows->exc[i - 1];
      if (ascending) continue;
      tmp.assign(rows->exc + rows->exc_off[r], rows->exc + rows->exc_off[r + 1]);
      std::sort(tmp.begin(), tmp.end());
      for (size_t i = 1; i < tmp.size(); i++)
        if (tmp[i] == tmp[i - 1]) { set_error("tbc_setfull_create_rows: read %u lists element %u twice (each element at most once per read)", r, tmp[i]); return TBC_ERR_INVALID_ARG; }
    }
  }
  // the prefix search per row and "the latest row" both rest on the documented orders
  for (uint32_t e = 1; e < S->E; e++) if (in->add_invoke[e] <= in->add_invoke[e - 1]) { set_error("tbc_setfull: add_invoke must be strictly ascending (element %u)", e); return TBC_ERR_INVALID_ARG; }
  for (uint32_t r = 1; r < S->R; r++) if (in->read_invoke[r] <= in->read_invoke[r - 1]) { set_error("tbc_setfull: read_invoke must be strictly ascending (read %u)", r); return TBC_ERR_INVALID_ARG; }
  SF_TRY(hipSetDevice(S->device));
  S->PITCH = std::max(1u, S->WPR);
  // (PITCH: the words between two rows in device memory.  A pitch padded off the power of two was measured -- 16 .. 1,088 words: the
  // same scan within 3 % either way, profiles/r06_setfull_pad_scan.txt -- so it is words_per_row)
  // enough chunks to fill the GPU with wavefronts that each stream a good stretch of rows
  const uint32_t col_blocks = (S->WPR + 255) / 256;
  uint32_t chunks = std::max(1u, std::min(256u, 8192u / std::max(1u, col_blocks)));     // short chunks: what pass 2 walks again is one chunk
  while (chunks > 1 && S->R / chunks < 64) chunks >>= 1;
  while ((S->R + chunks - 1) / chunks > kSetFullRows) chunks <<= 1;
  S->chunks = chunks; S->rows_per_chunk = std::max(1u, (S->R + chunks - 1) / chunks);
  const size_t e4 = (size_t)std::max(1u, S->E) * 4, r4 = (size_t)std::max(1u, S->R) * 4, m4 = std::max<size_t>(4, (size_t)S->R * S->PITCH * 4);
  const size_t ew = (size_t)std::max(1u, S->WPR) * 32 * 4;       // per-element outputs padded to whole words
  S->chp = (std::max(1u, S->WPR) + 3u) / 4u * 4u;
  const size_t any4 = (size_t)S->chp * S->chunks * 4;
  // Everything in ONE allocation (round 6: fourteen hipMalloc + three more for the compact reads, and as many hipFree -- each of which
  // waits for the device -- were most of a caller's 5 ms around a 0.16 ms scan; the reference checks one history per call site,
  // set_full.clj:157, so create + run + destroy IS its time to verdict)
  const uint64_t ne = (rows && S->R) ? rows->exc_off[S->R] : 0;
  size_t cursor = 0;
  const auto take = [&](size_t bytes) { const size_t at = cursor; cursor += (bytes + 255) & ~(size_t)255; return at; };
  const size_t o_M = take(m4), o_ai = take(e4), o_ao = take(e4), o_ri = take(r4), o_ro = take(r4), o_P = take(r4), o_pm = take((size_t)S->chunks * 8),
               o_lp = take(ew), o_la = take(ew), o_kn = take(ew), o_w = take((size_t)kWordCounters * 128), o_ap = take(any4), o_aa = take(any4),
               o_top = take(rows ? r4 : 0), o_off = take(rows ? ((size_t)S->R + 1) * 8 : 0), o_exc = take(rows ? std::max<size_t>(4, ne * 4) : 0);
  SF_TRY(hipMalloc(&S->arena, std::max<size_t>(cursor, 256)));
  char* const A0 = static_cast<char*>(S->arena);
  S->d_M = (uint32_t*)(A0 + o_M); S->d_add_invoke = (uint32_t*)(A0 + o_ai); S->d_add_ok = (uint32_t*)(A0 + o_ao);
  S->d_read_invoke = (uint32_t*)(A0 + o_ri); S->d_read_ok = (uint32_t*)(A0 + o_ro); S->d_P = (uint32_t*)(A0 + o_P);
  S->d_pmax = (uint32_t*)(A0 + o_pm);                              // the chunks' greatest prefixes, then their least
  S->d_lp = (uint32_t*)(A0 + o_lp); S->d_la = (uint32_t*)(A0 + o_la); S->d_known = (uint32_t*)(A0 + o_kn);
  S->d_words = (unsigned long long*)(A0 + o_w); S->d_anyp = (uint32_t*)(A0 + o_ap); S->d_anya = (uint32_t*)(A0 + o_aa);
  SF_TRY(hipStreamCreateWithFlags(&S->stream, hipStreamNonBlocking));
  SF_TRY(hipEventCreate(&S->ev0)); SF_TRY(hipEventCreate(&S->ev1));
  if (S->E) {
    SF_TRY(hipMemcpyAsync(S->d_add_invoke, in->add_invoke, (size_t)S->E * 4, hipMemcpyHostToDevice, S->stream));
    SF_TRY(hipMemcpyAsync(S->d_add_ok, in->add_ok, (size_t)S->E * 4, hipMemcpyHostToDevice, S->stream));
  }
  if (S->R) {
    SF_TRY(hipMemcpyAsync(S->d_read_invoke, in->read_invoke, (size_t)S->R * 4, hipMemcpyHostToDevice, S->stream));
    SF_TRY(hipMemcpyAsync(S->d_read_ok, in->read_ok, (size_t)S->R * 4, hipMemcpyHostToDevice, S->stream));
    if (!rows && S->WPR) SF_TRY(hipMemcpy2DAsync(S->d_M, (size_t)S->PITCH * 4, in->present, (size_t)S->WPR * 4, (size_t)S->WPR * 4, S->R, hipMemcpyHostToDevice, S->stream));
  }
  if (rows && S->R) {
    uint32_t* const d_top = (uint32_t*)(A0 + o_top); uint32_t* const d_exc = (uint32_t*)(A0 + o_exc);
    unsigned long long* const d_off = (unsigned long long*)(A0 + o_off);
    hipError_t e = hipMemcpyAsync(d_top, rows->top, (size_t)S->R * 4, hipMemcpyHostToDevice, S->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_off, rows->exc_off, ((size_t)S->R + 1) * 8, hipMemcpyHostToDevice, S->stream);
    if (e == hipSuccess && ne) e = hipMemcpyAsync(d_exc, rows->exc, ne * 4, hipMemcpyHostToDevice, S->stream);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(setfull_rows_kernel, dim3(std::min<uint32_t>(S->R, 16384u)), dim3(256), 0, S->stream, d_top, d_off, d_exc, S->R, std::max(1u, S->WPR), S->PITCH, S->d_M);
      e = hipGetLastError();
    }
    if (e != hipSuccess) { set_error("tbc_setfull_create_rows: building the matrix failed: %s", hipGetErrorString(e)); return e == hipErrorOutOfMemory ? TBC_ERR_OOM : TBC_ERR_HIP; }
  }
  // p[r] (how many elements had been invoked when read r completed) and the chunks' maxima depend on the inputs only
  SF_TRY(hipMemsetAsync(S->d_pmax, 0, (size_t)S->chunks * 4, S->stream));
  SF_TRY(hipMemsetAsync(S->d_pmax + S->chunks, 0xFF, (size_t)S->chunks * 4, S->stream));
  SF_TRY(hipMemsetAsync(S->d_anyp, 0, any4, S->stream)); SF_TRY(hipMemsetAsync(S->d_anya, 0, any4, S->stream));      // (the workgroups below the diagonal never write theirs)
  if (S->R && S->E)
    hipLaunchKernelGGL(setfull_prefix_kernel, dim3((S->R + 255) / 256), dim3(256), 0, S->stream, S->d_add_invoke, S->d_read_ok, S->E, S->R,
                       S->rows_per_chunk, S->chunks, S->d_P, S->d_pmax);
  SF_TRY(hipGetLastError());
  SF_TRY(hipStreamSynchronize(S->stream));
  return TBC_OK;
}

extern "C" {

tbc_status tbc_setfull_create(const tbc_setfull_in* in, tbc_setfull** handle) {
  if (!in || !handle || (in->n_elements && (!in->add_invoke || !in->add_ok)) ||
      (in->n_reads && (!in->read_invoke || !in->read_ok || !in->present))) {
    set_error("tbc_setfull_create: null argument");
    return TBC_ERR_INVALID_ARG;
  }
  tbc_setfull* S = new (std::nothrow) tbc_setfull();
  if (!S) return TBC_ERR_OOM;
  const tbc_status st = setfull_create_impl(in, S);
  if (st != TBC_OK) { delete S; return st; }
  *handle = S;
  return TBC_OK;
}

tbc_status tbc_setfull_run(tbc_setfull* S, tbc_setfull_out* out) {
  if (!S || !out || (S->E && (!out->known || !out->last_present || !out->last_absent))) { set_error("tbc_setfull_run: null argument"); return TBC_ERR_INVALID_ARG; }
  SF_TRY(hipSetDevice(S->device));
  hipStream_t s = S->stream;
  const size_t ew = (size_t)std::max(1u, S->WPR) * 32 * 4;
  SF_TRY(hipMemsetAsync(S->d_lp, 0, ew, s)); SF_TRY(hipMemsetAsync(S->d_la, 0, ew, s));
  SF_TRY(hipMemsetAsync(S->d_known, 0xFF, ew, s));
  SF_TRY(hipMemsetAsync(S->d_words, 0, (size_t)kWordCounters * 128, s));
  SF_TRY(hipEventRecord(S->ev0, s));
  if (S->R && S->E) {
    if (S->WPR % 4u == 0u)      // (hipMalloc'ed arrays, rows of a multiple of four words: every 16 B load and store is aligned)
      hipLaunchKernelGGL(setfull_any_kernel<4>, dim3(S->chunks, (S->WPR / 4u + 255) / 256), dim3(256), 0, s, S->d_M, S->d_P, S->d_pmax, S->E, S->R, S->WPR, S->PITCH,
                         S->rows_per_chunk, S->chp, S->d_anyp, S->d_anya, S->d_words);
    else
      hipLaunchKernelGGL(setfull_any_kernel<1>, dim3(S->chunks, (S->WPR + 255) / 256), dim3(256), 0, s, S->d_M, S->d_P, S->d_pmax, S->E, S->R, S->WPR, S->PITCH,
                         S->rows_per_chunk, S->chp, S->d_anyp, S->d_anya, S->d_words);
    hipLaunchKernelGGL(setfull_resolve_kernel, dim3((S->WPR + 3) / 4), dim3(256), 0, s, S->d_M, S->d_P, S->d_read_invoke, S->d_read_ok, S->d_anyp,
                       S->d_anya, S->E, S->R, S->WPR, S->PITCH, S->rows_per_chunk, S->chunks, S->chp, S->d_add_ok, S->d_lp, S->d_la, S->d_known, S->d_words);
  } else if (S->E) {        // no read at all: nothing was seen, known = the add's ack
    hipLaunchKernelGGL(setfull_finish_kernel, dim3((S->E + 255) / 256), dim3(256), 0, s, S->d_lp, S->d_la, S->d_known, S->d_add_ok, S->E);
  }
  SF_TRY(hipGetLastError());
  SF_TRY(hipEventRecord(S->ev1, s));
  unsigned long long words = 0;
  if (S->E) {
    SF_TRY(hipMemcpyAsync(out->known, S->d_known, (size_t)S->E * 4, hipMemcpyDeviceToHost, s));
    SF_TRY(hipMemcpyAsync(out->last_present, S->d_lp, (size_t)S->E * 4, hipMemcpyDeviceToHost, s));
    SF_TRY(hipMemcpyAsync(out->last_absent, S->d_la, (size_t)S->E * 4, hipMemcpyDeviceToHost, s));
  }
  unsigned long long* counters = S->h_words;
  SF_TRY(hipMemcpyAsync(counters, S->d_words, (size_t)kWordCounters * 128, hipMemcpyDeviceToHost, s));
  SF_TRY(hipStreamSynchronize(s));
  float ms = 0;
  SF_TRY(hipEventElapsedTime(&ms, S->ev0, S->ev1));
  out->ns_scan = (uint64_t)(ms * 1e6);
  for (uint32_t k = 0; k < kWordCounters; k++) words += counters[16u * k];
  out->bytes_scanned = (uint64_t)words * 4;
  out->bytes_matrix = (uint64_t)S->R * S->WPR * 4;
  return TBC_OK;
}

tbc_status tbc_setfull_create_rows(const tbc_setfull_rows* in, tbc_setfull** handle) {
  if (!in || !handle || (in->n_elements && (!in->add_invoke || !in->add_ok)) ||
      (in->n_reads && (!in->read_invoke || !in->read_ok || !in->top)) || !in->exc_off || (in->exc_off[in->n_reads] && !in->exc) || in->reserved0 != 0) {
    set_error("tbc_setfull_create_rows: null argument");
    return TBC_ERR_INVALID_ARG;
  }
  tbc_setfull_in dense{};
  dense.n_elements = in->n_elements; dense.n_reads = in->n_reads; dense.words_per_row = std::max(1u, (in->n_elements + 31u) / 32u); dense.device = in->device;
  dense.add_invoke = in->add_invoke; dense.add_ok = in->add_ok; dense.read_invoke = in->read_invoke; dense.read_ok = in->read_ok; dense.present = nullptr;
  tbc_setfull* S = new (std::nothrow) tbc_setfull();
  if (!S) return TBC_ERR_OOM;
  const tbc_status st = setfull_create_impl(&dense, S, in);
  if (st != TBC_OK) { delete S; return st; }
  *handle = S;
  return TBC_OK;
}

void tbc_setfull_destroy(tbc_setfull* S) {
  if (!S) return;
  (void)hipSetDevice(S->device);
  delete S;
}

}  // extern "C"
