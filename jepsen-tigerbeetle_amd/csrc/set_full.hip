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
// consecutive words of a row (256 B coalesced), a thread owns ONE word column (32 elements) of ONE chunk of
// rows, grid = columns x chunks.  Bit-parallel: a thread walks its rows from the latest down keeping two
// 32-bit "found" masks; a bit newly seen present (absent) fixes that element's last_present (last_absent)
// for this chunk -- one atomicMax, the latest chunk wins -- and the walk stops when all 32 bits have both.
// Elements are numbered by add invocation, so "reads completing after e's add was invoked" is a PREFIX of the
// elements for each row (p[r], one binary search per row in setfull_prefix_kernel) and a chunk whose reads all
// precede a column's adds is skipped without a load: the all-zero triangle below the diagonal is never read.
// known: the same walk upwards; the first read (in invocation order) containing e need not be the first to
// complete, so the walk keeps offering later rows' read_ok (atomicMin) while they were invoked before the
// latest first-completion seen -- a window bounded by the number of concurrent readers.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstring>
#include <new>
#include "tbc_internal.h"

using namespace tbc;

namespace {

constexpr uint32_t kNoneU = 0xFFFFFFFFu;

__global__ __launch_bounds__(256) void setfull_prefix_kernel(const uint32_t* add_invoke, const uint32_t* read_ok, uint32_t E, uint32_t R,
                                                             uint32_t rows_per_chunk, uint32_t* P, uint32_t* pmax) {
  const uint32_t r = blockIdx.x * 256u + threadIdx.x;
  if (r >= R) return;
  const uint32_t t = read_ok[r];
  uint32_t lo = 0, hi = E;                       // first element whose add was invoked at or after this read's completion
  while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (add_invoke[mid] < t) lo = mid + 1; else hi = mid; }
  P[r] = lo;
  atomicMax(&pmax[r / rows_per_chunk], lo);
}

__device__ __forceinline__ uint32_t prefix_mask(uint32_t p, uint32_t w) {       // bits of word w below element number p
  return p >= 32u * w + 32u ? 0xFFFFFFFFu : (p <= 32u * w ? 0u : (1u << (p - 32u * w)) - 1u);
}

__global__ __launch_bounds__(256) void setfull_scan_kernel(const uint32_t* __restrict__ M, const uint32_t* __restrict__ P,
                                                           const uint32_t* __restrict__ pmax, const uint32_t* __restrict__ read_invoke,
                                                           const uint32_t* __restrict__ read_ok, uint32_t E, uint32_t R, uint32_t WPR,
                                                           uint32_t rows_per_chunk, uint32_t* lp1, uint32_t* la1, uint32_t* known,
                                                           unsigned long long* words_loaded) {
  const uint32_t w = blockIdx.x * 256u + threadIdx.x, c = blockIdx.y;
  uint32_t loaded = 0;
  if (w < WPR && pmax[c] > 32u * w) {
    const uint32_t r0 = c * rows_per_chunk, r1 = min(r0 + rows_per_chunk, R);
    const uint32_t full = E - 32u * w >= 32u ? 0xFFFFFFFFu : (1u << (E - 32u * w)) - 1u;
    // ---- from the latest read down: last_present / last_absent (stored + 1, 0 = none; atomicMax over the chunks)
    uint32_t found_p = 0, found_a = 0;
    for (uint32_t r = r1; r-- > r0;) {
      const uint32_t valid = prefix_mask(P[r], w) & full;
      if (!valid) continue;
      const uint32_t word = M[(uint64_t)r * WPR + w];
      loaded++;
      uint32_t np = word & valid & ~found_p, na = ~word & valid & ~found_a;
      const uint32_t inv1 = read_invoke[r] + 1u;
      found_p |= np; found_a |= na;
      while (np) { const uint32_t b = (uint32_t)__builtin_ctz(np); np &= np - 1u; atomicMax(&lp1[32u * w + b], inv1); }
      while (na) { const uint32_t b = (uint32_t)__builtin_ctz(na); na &= na - 1u; atomicMax(&la1[32u * w + b], inv1); }
      if ((found_p & found_a) == full) break;
    }
    // ---- from the earliest read up: known.  `until` = the latest completion among the reads that were first to
    // contain some element: any read invoked before it may still complete earlier and must be offered too.
    uint32_t seen = 0, until = 0;
    for (uint32_t r = r0; r < r1; r++) {
      if (seen == full && read_invoke[r] > until) break;
      const uint32_t valid = prefix_mask(P[r], w) & full;
      if (!valid) continue;
      const uint32_t word = M[(uint64_t)r * WPR + w];
      loaded++;
      uint32_t hits = word & valid;
      const uint32_t ok = read_ok[r];
      // a bit seen before already holds a completion <= `until`: only a read that completed EARLIER can improve it
      const uint32_t fresh = hits & ~seen;
      if (ok >= until) hits = fresh;
      if (fresh) until = max(until, ok);
      seen |= fresh | hits;
      while (hits) { const uint32_t b = (uint32_t)__builtin_ctz(hits); hits &= hits - 1u; atomicMin(&known[32u * w + b], ok); }
    }
  }
  // words loaded, one atomic per wavefront
  unsigned long long tot = loaded;
  for (int d = 32; d >= 1; d >>= 1) tot += __shfl_xor(tot, d);
  if ((threadIdx.x & 63u) == 0 && tot) atomicAdd(words_loaded, tot);
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
  uint32_t E = 0, R = 0, WPR = 0, chunks = 1, rows_per_chunk = 1;
  uint32_t *d_add_invoke = nullptr, *d_add_ok = nullptr, *d_read_invoke = nullptr, *d_read_ok = nullptr, *d_M = nullptr;
  uint32_t *d_P = nullptr, *d_pmax = nullptr, *d_lp = nullptr, *d_la = nullptr, *d_known = nullptr;
  unsigned long long* d_words = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  ~tbc_setfull() {
    for (void* p : {(void*)d_add_invoke, (void*)d_add_ok, (void*)d_read_invoke, (void*)d_read_ok, (void*)d_M, (void*)d_P, (void*)d_pmax,
                    (void*)d_lp, (void*)d_la, (void*)d_known, (void*)d_words}) if (p) (void)hipFree(p);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    if (stream) (void)hipStreamDestroy(stream);
  }
};

static tbc_status setfull_create_impl(const tbc_setfull_in* in, tbc_setfull* S) {
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
  SF_TRY(hipSetDevice(S->device));
  // enough chunks to fill the GPU with wavefronts that each stream a good stretch of rows
  const uint32_t col_blocks = (S->WPR + 255) / 256;
  uint32_t chunks = std::max(1u, std::min(256u, 4096u / std::max(1u, col_blocks)));
  while (chunks > 1 && S->R / chunks < 64) chunks >>= 1;
  S->chunks = chunks; S->rows_per_chunk = std::max(1u, (S->R + chunks - 1) / chunks);
  const size_t e4 = (size_t)std::max(1u, S->E) * 4, r4 = (size_t)std::max(1u, S->R) * 4, m4 = std::max<size_t>(4, (size_t)S->R * S->WPR * 4);
  SF_TRY(hipMalloc((void**)&S->d_add_invoke, e4)); SF_TRY(hipMalloc((void**)&S->d_add_ok, e4));
  SF_TRY(hipMalloc((void**)&S->d_read_invoke, r4)); SF_TRY(hipMalloc((void**)&S->d_read_ok, r4));
  SF_TRY(hipMalloc((void**)&S->d_M, m4)); SF_TRY(hipMalloc((void**)&S->d_P, r4)); SF_TRY(hipMalloc((void**)&S->d_pmax, (size_t)S->chunks * 4));
  const size_t ew = (size_t)std::max(1u, S->WPR) * 32 * 4;       // per-element outputs padded to whole words
  SF_TRY(hipMalloc((void**)&S->d_lp, ew)); SF_TRY(hipMalloc((void**)&S->d_la, ew)); SF_TRY(hipMalloc((void**)&S->d_known, ew));
  SF_TRY(hipMalloc((void**)&S->d_words, 8));
  SF_TRY(hipStreamCreateWithFlags(&S->stream, hipStreamNonBlocking));
  SF_TRY(hipEventCreate(&S->ev0)); SF_TRY(hipEventCreate(&S->ev1));
  if (S->E) {
    SF_TRY(hipMemcpyAsync(S->d_add_invoke, in->add_invoke, (size_t)S->E * 4, hipMemcpyHostToDevice, S->stream));
    SF_TRY(hipMemcpyAsync(S->d_add_ok, in->add_ok, (size_t)S->E * 4, hipMemcpyHostToDevice, S->stream));
  }
  if (S->R) {
    SF_TRY(hipMemcpyAsync(S->d_read_invoke, in->read_invoke, (size_t)S->R * 4, hipMemcpyHostToDevice, S->stream));
    SF_TRY(hipMemcpyAsync(S->d_read_ok, in->read_ok, (size_t)S->R * 4, hipMemcpyHostToDevice, S->stream));
    SF_TRY(hipMemcpyAsync(S->d_M, in->present, (size_t)S->R * S->WPR * 4, hipMemcpyHostToDevice, S->stream));
  }
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
  SF_TRY(hipMemsetAsync(S->d_pmax, 0, (size_t)S->chunks * 4, s)); SF_TRY(hipMemsetAsync(S->d_words, 0, 8, s));
  SF_TRY(hipEventRecord(S->ev0, s));
  if (S->R && S->E) {
    hipLaunchKernelGGL(setfull_prefix_kernel, dim3((S->R + 255) / 256), dim3(256), 0, s, S->d_add_invoke, S->d_read_ok, S->E, S->R,
                       S->rows_per_chunk, S->d_P, S->d_pmax);
    hipLaunchKernelGGL(setfull_scan_kernel, dim3((S->WPR + 255) / 256, S->chunks), dim3(256), 0, s, S->d_M, S->d_P, S->d_pmax, S->d_read_invoke,
                       S->d_read_ok, S->E, S->R, S->WPR, S->rows_per_chunk, S->d_lp, S->d_la, S->d_known, S->d_words);
  }
  if (S->E) hipLaunchKernelGGL(setfull_finish_kernel, dim3((S->E + 255) / 256), dim3(256), 0, s, S->d_lp, S->d_la, S->d_known, S->d_add_ok, S->E);
  SF_TRY(hipGetLastError());
  SF_TRY(hipEventRecord(S->ev1, s));
  unsigned long long words = 0;
  if (S->E) {
    SF_TRY(hipMemcpyAsync(out->known, S->d_known, (size_t)S->E * 4, hipMemcpyDeviceToHost, s));
    SF_TRY(hipMemcpyAsync(out->last_present, S->d_lp, (size_t)S->E * 4, hipMemcpyDeviceToHost, s));
    SF_TRY(hipMemcpyAsync(out->last_absent, S->d_la, (size_t)S->E * 4, hipMemcpyDeviceToHost, s));
  }
  SF_TRY(hipMemcpyAsync(&words, S->d_words, 8, hipMemcpyDeviceToHost, s));
  SF_TRY(hipStreamSynchronize(s));
  float ms = 0;
  SF_TRY(hipEventElapsedTime(&ms, S->ev0, S->ev1));
  out->ns_scan = (uint64_t)(ms * 1e6);
  out->bytes_scanned = (uint64_t)words * 4;
  out->bytes_matrix = (uint64_t)S->R * S->WPR * 4;
  return TBC_OK;
}

void tbc_setfull_destroy(tbc_setfull* S) {
  if (!S) return;
  (void)hipSetDevice(S->device);
  delete S;
}

}  // extern "C"
