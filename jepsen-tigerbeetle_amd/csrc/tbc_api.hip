// tbc_api.hip -- host orchestration behind the C-ABI (include/tbcheck.h):
// device arenas, H2D of the op columns, pack + search launches on the batch's
// own HIP stream, visited-set overflow retries, verdict marshalling.
//
// There is no CPU path in here by design: every compute entry point needs a
// gfx950 device and says TBC_ERR_NO_DEVICE otherwise.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>
#include <vector>
#include "tbc_internal.h"
#include "reach_table.h"
#include "witness_expand.h"

using namespace tbc;

namespace {

#define HIP_TRY(expr)                                                             \
  do {                                                                            \
    hipError_t e_ = (expr);                                                       \
    if (e_ != hipSuccess) {                                                       \
      set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return e_ == hipErrorOutOfMemory ? TBC_ERR_OOM : TBC_ERR_HIP;               \
    }                                                                             \
  } while (0)

inline uint32_t ceil_log2(uint64_t x) {
  uint32_t l = 0;
  while ((1ull << l) < x) l++;
  return l;
}

// Entries of a history's per-front open-call lists: every live call appears once per front it is open
// at (the completions positioned between its invocation and its completion, plus its own).  Exact when
// the positions are event indices; anything else falls back to the worst case (every slot at every front).
// branch_lists: the lists hold the live :write / :cas calls only (kRuleBranch), so the reads are not counted.
uint64_t open_list_entries(const tbc_ops& c, uint64_t op_off, uint64_t n, uint32_t n_events, uint32_t n_slots,
                           std::vector<uint32_t>& pre, bool branch_lists) {
  const uint64_t worst = std::max<uint64_t>(n, 1) * std::max(1u, n_slots);
  if (n == 0) return 1;
  if ((uint64_t)n_events > 64 * n + 1024) return worst;
  const uint32_t* inv = c.inv_pos + op_off;
  const uint32_t* ret = c.ret_pos + op_off;
  pre.assign((size_t)n_events + 1, 0u);              // pre[x] = completions positioned before x
  for (uint64_t i = 0; i < n; i++) {
    if (ret[i] == TBC_POS_CRASHED) continue;
    if (ret[i] >= n_events || inv[i] > ret[i]) return worst;
    pre[ret[i] + 1] = 1;
  }
  for (uint32_t x = 1; x <= n_events; x++) pre[x] += pre[x - 1];
  uint64_t total = 0;
  const uint8_t* f = c.f + op_off;
  for (uint64_t i = 0; i < n; i++)
    if (ret[i] != TBC_POS_CRASHED && !(branch_lists && f[i] == TBC_F_READ)) total += pre[ret[i]] - pre[inv[i]] + 1;
  return std::min(worst, std::max<uint64_t>(total, 1));
}

// ---- count form (tbc_internal.h, kRuleCount; specified in oracle/wgl_count.c): what the host works out of one history when the
// inputs become resident -- the re-used process slots of the live calls, the classes of the crashed calls with their members'
// invocation ranks, the layout of the count vector.  Returns false when the form does not apply (more than 128 bits of counts,
// a process id out of range: the pack kernel will say what is wrong with such a history).
struct CountHist {
  std::vector<uint64_t> words;       // [2 * n_classes words of OpRec][members of class 0, sentinel, members of class 1, sentinel, ...]
  uint32_t n_classes = 0, n_slots = 1;
  uint64_t top[kCountWords] = {0, 0};
};
bool build_count_form(const tbc_ops& c, uint64_t o0, uint64_t n, uint32_t n_process, bool cas_model, int32_t* slot_col, CountHist& out,
                      std::vector<uint32_t>& rets, std::vector<int32_t>& slot_of, std::vector<uint8_t>& used) {
  const uint8_t* f = c.f + o0; const int32_t* a = c.a + o0; const int32_t* b = c.b + o0; const int32_t* proc = c.process + o0;
  const uint32_t* inv = c.inv_pos + o0; const uint32_t* ret = c.ret_pos + o0;
  out.words.clear(); out.n_classes = 0; out.n_slots = 1; out.top[0] = out.top[1] = 0;
  rets.clear();
  for (uint64_t i = 0; i < n; i++) if (ret[i] != TBC_POS_CRASHED) rets.push_back(ret[i]);
  std::sort(rets.begin(), rets.end());
  slot_of.assign((size_t)n_process + 1, -1);
  used.assign((size_t)n_process + 2, 0);
  struct Cls { uint32_t f; int32_t a, b; std::vector<uint64_t> mem; };
  std::vector<Cls> cls;
  for (uint64_t i = 0; i < n; i++) {
    if (proc[i] < 0 || (uint32_t)proc[i] >= n_process) return false;
    const uint32_t p = (uint32_t)proc[i];
    if (ret[i] == TBC_POS_CRASHED) {
      if (slot_of[p] >= 0) { used[(size_t)slot_of[p]] = 0; slot_of[p] = -1; }     // a process that crashes hands its slot back
      slot_col[i] = 0;                                                            // (slotless: the pack kernel does not look at it)
      if (!(f[i] == TBC_F_WRITE || (f[i] == TBC_F_CAS && cas_model && a[i] != b[i]))) continue;   // no effect: never a candidate
      size_t k = 0;
      while (k < cls.size() && !(cls[k].f == f[i] && cls[k].a == a[i] && (f[i] != TBC_F_CAS || cls[k].b == b[i]))) k++;
      if (k == cls.size()) cls.push_back(Cls{f[i], a[i], f[i] == TBC_F_CAS ? b[i] : 0, {}});
      const uint32_t inv_rank = (uint32_t)(std::lower_bound(rets.begin(), rets.end(), inv[i]) - rets.begin());
      cls[k].mem.push_back((uint64_t)inv_rank | ((uint64_t)i << 32));
      continue;
    }
    if (slot_of[p] < 0) {                                   // the lowest slot that is free when the process first invokes
      uint32_t sl = 0;
      while (used[sl]) sl++;
      used[sl] = 1; slot_of[p] = (int32_t)sl;
      out.n_slots = std::max(out.n_slots, sl + 1);
    }
    slot_col[i] = slot_of[p];
  }
  out.n_classes = (uint32_t)cls.size();
  uint32_t bits = 0;
  out.words.assign(2 * cls.size(), 0ull);
  for (size_t k = 0; k < cls.size(); k++) {
    uint32_t w = 0;
    while ((1ull << w) <= cls[k].mem.size()) w++;
    if ((bits & 63u) + w > 64u) bits = (bits + 63u) & ~63u;     // a field never straddles a word
    if (bits + w > 64u * kCountWords || w > 31u) return false;
    OpRec o; o.op = (uint32_t)out.words.size(); o.f_slot = cls[k].f | (bits << 8) | (w << 16); o.a = cls[k].a; o.b = cls[k].b;
    std::memcpy(&out.words[2 * k], &o, sizeof o);
    const uint32_t t = bits + w - 1;
    out.top[t >> 6] |= 1ull << (t & 63u);
    bits += w;
    out.words.insert(out.words.end(), cls[k].mem.begin(), cls[k].mem.end());
    out.words.push_back(~0ull);                                 // sentinel: no further member is ever invoked
  }
  if (out.words.size() & 1) out.words.push_back(~0ull);         // (the next history's class records stay 16 B aligned)
  return true;
}

// What the layout decisions ask of the op columns, in ONE pass over them (four passes of 3 GB each were 0.6 s of a 32,768-history
// tbc_batch_create), dealt to a few host threads: do all register values fit the rule tables (>= 0), the greatest value, is any
// call crashed, is any crashed call one with an effect (:write, or :cas [a b] with a != b).
struct ColumnScan { bool nonneg = true, any_crashed = false, any_crashed_effect = false; int32_t vmax = -1; };
ColumnScan scan_columns(const tbc_ops& c, uint64_t T) {
  const unsigned nt = T > (1ull << 22) ? 8u : 1u;
  std::vector<ColumnScan> part(nt);
  const auto work = [&](unsigned t) {
    ColumnScan r;
    const uint64_t lo = T * t / nt, hi = T * (t + 1) / nt;
    for (uint64_t i = lo; i < hi; i++) {
      const int32_t a = c.a[i];
      const uint32_t f = c.f[i];
      if (a != TBC_NIL) { r.nonneg = r.nonneg && a >= 0; r.vmax = std::max(r.vmax, a); }
      int32_t b = 0;
      if (f == TBC_F_CAS) { b = c.b[i]; r.nonneg = r.nonneg && b >= 0; r.vmax = std::max(r.vmax, b); }
      if (c.ret_pos[i] == TBC_POS_CRASHED) {
        r.any_crashed = true;
        r.any_crashed_effect = r.any_crashed_effect || f == TBC_F_WRITE || (f == TBC_F_CAS && a != b);
      }
    }
    part[t] = r;
  };
  if (nt == 1) work(0);
  else {
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++) th.emplace_back(work, t);
    for (auto& x : th) x.join();
  }
  ColumnScan out;
  for (const ColumnScan& r : part) {
    out.nonneg = out.nonneg && r.nonneg; out.vmax = std::max(out.vmax, r.vmax);
    out.any_crashed = out.any_crashed || r.any_crashed; out.any_crashed_effect = out.any_crashed_effect || r.any_crashed_effect;
  }
  return out;
}

bool device_is_gfx950(int dev) {
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, dev) != hipSuccess) return false;
  return std::strncmp(p.gcnArchName, "gfx950", 6) == 0;
}

// ---- persistent device contexts for tbc_check.  A single-history call used to pay ~35 hipMalloc / hipFree, a
// stream and six events -- more than its kernels.  A context keeps one device slab, a stream and the events alive
// between calls; a call takes a context from the pool (so concurrent callers -- jepsen.checker/compose runs its
// checkers on several JVM threads -- each get their own), carves its arenas out of the slab with a bump pointer
// and hands the context back.  A call that needs more than the slab holds falls back to hipMalloc for the rest
// and the slab is re-sized for the next call.
struct Ctx {
  int device = 0;
  char* slab = nullptr;
  size_t cap = 0, used = 0, wanted = 0;
  // ... and one PINNED host region (round 5): what a call copies to and from the device -- the op columns staged as one block, the
  // sweep's relation table, the descriptors read back -- goes through it, so a copy is one DMA instead of a staged blit per 64 KB
  // of pageable memory, and nothing waits for a copy before the kernels are queued (tbc_check: 77 + 50 us of its 1.07 ms)
  char* pin = nullptr;
  size_t pin_cap = 0, pin_used = 0, pin_wanted = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev[6] = {};
  hipStream_t stream2 = nullptr;      // the relaxed sweep beside the exact search (tbc_batch::rsweep), created when first wanted
  hipEvent_t ev2[2] = {};
};
thread_local Ctx* t_ctx = nullptr;      // the context the calling thread's DevBufs draw from (tbc_check only)
std::mutex g_ctx_mu;
std::vector<Ctx*> g_ctx_free;

// TBC_GUARD=1 (diagnostic, like TBC_DEBUG): every device arena gets 256 poisoned bytes behind it and tbc_batch_run checks them all when
// it ends -- a kernel that writes past an arena is named (allocation number, address, the first bad byte) instead of corrupting a
// neighbour silently.  Round 4 saw ONE bench run of five die with a GPU memory fault that nothing reproduced; this is how it was hunted
// (profiles/r05_guard_runs.txt).
bool guard_on() { static const bool on = [] { const char* e = std::getenv("TBC_GUARD"); return e && e[0] == '1'; }(); return on; }
// (an arena belongs to the batch being created or run by the allocating thread -- t_guard_owner; a run checks its own batch's arenas
// only: another thread's batch may be poisoning a re-used piece of its context's slab at that very moment -- the first version of this
// check read such bytes and cried wolf, ten times in a two-thread bench run)
struct GuardRec { const char* at; size_t serial; size_t nth; const void* owner; };
std::mutex g_guard_mu;
std::vector<GuardRec> g_guards;
size_t g_guard_serial = 0;
thread_local const void* t_guard_owner = nullptr;
thread_local size_t t_guard_nth = 0;          // the arena's number within its batch (allocation order of batch_create_impl: names it)
constexpr size_t kGuardBytes = 256;
void guard_add(const void* arena_end) {
  static const std::vector<unsigned char> poison(kGuardBytes, 0xA5);
  (void)hipMemcpy(const_cast<void*>(arena_end), poison.data(), kGuardBytes, hipMemcpyHostToDevice);      // (synchronous: the bytes are there before anything is launched)
  std::lock_guard<std::mutex> lk(g_guard_mu);
  g_guards.push_back(GuardRec{(const char*)arena_end, g_guard_serial++, t_guard_nth++, t_guard_owner});
}
void guard_remove(const void* arena_end) {
  std::lock_guard<std::mutex> lk(g_guard_mu);
  for (size_t i = 0; i < g_guards.size(); i++) if (g_guards[i].at == (const char*)arena_end) { g_guards.erase(g_guards.begin() + (long)i); return; }
}
// returns the number of arenas whose guard bytes were overwritten (after the caller's stream is idle)
size_t guard_check(const char* when, const void* owner) {
  std::vector<GuardRec> live;
  { std::lock_guard<std::mutex> lk(g_guard_mu); for (const GuardRec& g : g_guards) if (g.owner == owner) live.push_back(g); }
  size_t bad = 0;
  unsigned char buf[kGuardBytes];
  for (const GuardRec& g : live) {
    if (hipMemcpy(buf, g.at, kGuardBytes, hipMemcpyDeviceToHost) != hipSuccess) continue;
    for (size_t i = 0; i < kGuardBytes; i++) if (buf[i] != 0xA5) {
      std::fprintf(stderr, "[tbc guard] %s: arena #%zu (the batch's %zu-th, ends at %p) overrun: byte +%zu = 0x%02x\n", when, g.serial, g.nth, (const void*)g.at, i, buf[i]);
      bad++;
      break;
    }
  }
  return bad;
}

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  bool owned = false;
  bool guarded = false;
  tbc_status alloc(size_t count) {
    n = count;
    if (count == 0) count = 1;
    const size_t body = (count * sizeof(T) + 255) & ~(size_t)255;
    const size_t bytes = body + (guard_on() ? kGuardBytes : 0);
    guarded = guard_on();
    if (t_ctx) {
      t_ctx->wanted += bytes;
      if (t_ctx->used + bytes <= t_ctx->cap) {
        p = (T*)(t_ctx->slab + t_ctx->used); t_ctx->used += bytes; owned = false;
        if (guarded) guard_add((const char*)p + body);
        return TBC_OK;
      }
    }
    HIP_TRY(hipMalloc((void**)&p, bytes));
    owned = true;
    if (guarded) guard_add((const char*)p + body);
    return TBC_OK;
  }
  void release() {
    if (p && guarded) guard_remove((const char*)p + (((n ? n : 1) * sizeof(T) + 255) & ~(size_t)255));
    if (p && owned) (void)hipFree(p);
    p = nullptr; n = 0; owned = false; guarded = false;
  }
  size_t bytes() const { return (n ? n : 1) * sizeof(T); }
};

// host memory a stream copies into or out of: carved from the calling thread's persistent context's pinned region when there is
// one (tbc_check), a plain vector otherwise (a resident batch reads its results back into pageable memory as before)
template <typename T>
struct HostBuf {
  T* p = nullptr;
  size_t n = 0;
  std::vector<T> own;
  void resize(size_t count) {
    n = count;
    const size_t bytes = (std::max<size_t>(count, 1) * sizeof(T) + 255) & ~(size_t)255;
    if (t_ctx) {
      t_ctx->pin_wanted += bytes;
      if (t_ctx->pin_used + bytes <= t_ctx->pin_cap) { p = (T*)(t_ctx->pin + t_ctx->pin_used); t_ctx->pin_used += bytes; own.clear(); return; }
    }
    own.resize(count);
    p = own.data();
  }
  T* data() { return p; }
  const T* data() const { return p; }
  size_t size() const { return n; }
  T& operator[](size_t i) { return p[i]; }
  const T& operator[](size_t i) const { return p[i]; }
  T* begin() { return p; }
  T* end() { return p + n; }
  const T* begin() const { return p; }
  const T* end() const { return p + n; }
};

Ctx* ctx_acquire(int device) {
  {
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    for (size_t i = 0; i < g_ctx_free.size(); i++) if (g_ctx_free[i]->device == device) {
      Ctx* c = g_ctx_free[i]; g_ctx_free.erase(g_ctx_free.begin() + (long)i); return c;
    }
  }
  Ctx* c = new (std::nothrow) Ctx();
  if (!c) return nullptr;
  c->device = device;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return nullptr; }
  for (auto& e : c->ev) if (hipEventCreate(&e) != hipSuccess) { delete c; return nullptr; }
  return c;
}
void ctx_release(Ctx* c) {
  // size the slab for the next call of this kind (1.25 x what this one asked for), within reason
  if (c->wanted > c->cap && c->wanted < (8ull << 30)) {
    if (c->slab) (void)hipFree(c->slab);
    c->slab = nullptr; c->cap = 0;
    const size_t want = c->wanted + c->wanted / 4;
    if (hipMalloc((void**)&c->slab, want) == hipSuccess) c->cap = want;
  }
  c->used = 0; c->wanted = 0;
  if (c->pin_wanted > c->pin_cap && c->pin_wanted < (1ull << 30)) {
    if (c->pin) { (void)hipHostUnregister(c->pin); std::free(c->pin); }
    c->pin = nullptr; c->pin_cap = 0;
    const size_t want = (c->pin_wanted + c->pin_wanted / 4 + 4095) & ~(size_t)4095;
    // (ordinary cached memory, registered: the composition reads the 0.6 MB relation table right after the copy, and through
    // hipHostMalloc's mapping -- coherent or "non-coherent" alike -- that took 78 us instead of 27; the stream synchronize before it
    // makes the copy visible)
    void* mem = std::aligned_alloc(4096, want);
    if (mem && hipHostRegister(mem, want, hipHostRegisterDefault) == hipSuccess) { c->pin = (char*)mem; c->pin_cap = want; }
    else std::free(mem);
  }
  c->pin_used = 0; c->pin_wanted = 0;
  std::lock_guard<std::mutex> lk(g_ctx_mu);
  g_ctx_free.push_back(c);
}

uint64_t now_ns();
#define SYNC_TRACE(msg) do { if (std::getenv("TBC_SYNC_EACH")) { hipError_t e__ = hipStreamSynchronize(s); std::fprintf(stderr, "[tbc sync] %s -> %s\n", msg, hipGetErrorString(e__)); std::fflush(stderr); } } while (0)
#define TRACE(msg) do { if (std::getenv("TBC_DEBUG")) { std::fprintf(stderr, "[tbc %9.3f ms] %s\n", (double)(now_ns() % 100000000000ull) / 1e6, msg); std::fflush(stderr); } } while (0)

uint64_t now_ns() {
  return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(
             std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

namespace {
// TBC_DEBUG=1: kernels mirror their progress into host-mapped words so a hang can be diagnosed
// from another thread (tbc_debug_peek) while the call is still blocked.
uint32_t* g_dbg = nullptr;
uint32_t* debug_words() {
  static bool tried = false;
  if (!tried) {
    tried = true;
    const char* e = std::getenv("TBC_DEBUG");
    if (e && e[0] == '1') {
      void* p = nullptr;
      if (hipHostMalloc(&p, 64 * sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
        std::memset(p, 0, 64 * sizeof(uint32_t));
        g_dbg = (uint32_t*)p;
      }
    }
  }
  return g_dbg;
}
}  // namespace

extern "C" int tbc_debug_peek(uint32_t* out, uint32_t n) {
  if (!g_dbg || !out) return 0;
  for (uint32_t i = 0; i < n && i < 64; i++) out[i] = ((volatile uint32_t*)g_dbg)[i];
  return 1;
}

struct tbc_batch {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev[6] = {};
  uint32_t n_hist = 0;
  uint64_t total_ops = 0, max_ops = 1;
  uint32_t mask_words = 1;
  uint32_t frame_words = 6;
  tbc_model model{};
  tbc_opts opts{};
  std::vector<Hist> hist;          // host mirror
  std::vector<uint16_t> table_host;
  // device arenas
  DevBuf<uint8_t> d_f;
  DevBuf<int32_t> d_a, d_b, d_proc;
  DevBuf<uint32_t> d_inv, d_ret;
  DevBuf<Hist> d_hist;
  DevBuf<Rec> d_rec;
  DevBuf<uint32_t> d_seg, d_ret_slot, d_ret_op, d_bitmap, d_wpre, d_frames, d_witness, d_work, d_queue;
  DevBuf<uint64_t> d_tab;
  DevBuf<DevResult> d_results;
  DevBuf<uint16_t> d_table;
  DevBuf<int32_t> d_pool_vals;
  uint32_t pool_len = 0;
  DevBuf<uint64_t> d_cfg;           // configs at the failing front, kCfgCap records per history
  // wide schedule (search_width > 1)
  uint32_t width = 1;
  // u64 words per front record (0 = plain rdm rows): the compact 64 B form where one mask word and six row entries do
  uint32_t front_words() const { return !lanes ? 0u : ((rules & kRuleEager) && front_compact_ok(n_dom, mask_words)) ? kFrontCompactWords : front_stride(vpad, mask_words); }
  uint32_t lanes = 0;               // 8 / 16 / 32: several histories per wavefront (wgl_narrow.hip), one config per iteration; 0 = one per wavefront
  // The order of a front's list of open calls (tbc_opts.list_order; PackOpenArgs.list_order).  The search takes a config's candidates last to
  // first and pops the last child first; in order of COMPLETION, a :write placed as if it completed 24 ranks later (16 + 24), the call
  // that completes soonest is tried first and a :cas the state allows now goes before a :write that completes soon after it: on the bench
  // workload 4,513 rounds a history instead of 5,580 in process-slot order, at 19 calls in flight half the rounds of the wide kernel, at 32 a
  // third (oracle counts; measured round 5: search 70.9 -> 43.1 ms per 8,192 x 8 histories).  It is the library's
  // choice wherever nothing depends on slot order: the walk with lane = front (one mask word), the register family under the
  // rules' value range, no level sweep beside the search (its origins are numbered by list position), no count form (its oracle
  // counts say slot order), no round budget.  A witness's absorbed reads are replayed in the same order (witness_expand.h).
  static constexpr uint32_t kDefaultListOrder = 16u + 24u;
  bool list_order_applies() const {
    return width > 1 && mask_words == 1 && vpad <= 32 && !(rules & kRuleCount) && !sweep && opts.round_budget == 0 &&
           (model.kind == TBC_MODEL_REGISTER || model.kind == TBC_MODEL_CAS_REGISTER);
  }
  // PackOpenArgs.list_order: 0 = slot order, 1 = completion, 2 = completion with the :write calls last, 16 + W
  uint32_t list_order() const {
    if (!list_order_applies() || opts.list_order == TBC_ORDER_SLOT) return 0u;
    if (opts.list_order == TBC_ORDER_DEFAULT) return kDefaultListOrder;
    return opts.list_order >= 16u ? opts.list_order : opts.list_order - 1u;        // TBC_ORDER_COMPLETION = 2 -> 1, TBC_ORDER_WRITES_LAST = 3 -> 2
  }
  std::vector<BeamHist> bh;
  DevBuf<BeamHist> d_bh;
  DevBuf<uint32_t> d_off, d_ncr, d_stack;
  DevBuf<OpRec> d_lst, d_crashed;   // per-front open-call lists / crashed calls, whole records
  DevBuf<uint8_t> d_slot8;          // completion slots as bytes
  DevBuf<uint8_t> d_rk8;            // narrow kernel: read kind per rank
  bool lookahead = false;           // wide single-wave schedule, register family, tbc_opts.lookahead != 1
  DevBuf<uint64_t> d_look;          // lookahead records per completion rank
  DevBuf<uint32_t> d_dstack;        // second stack per history: configs the lookahead set aside
  DevBuf<uint32_t> d_looktmp;
  // level sweep (jit_sweep.hip): TBC_ALG_LINEAR, and TBC_ALG_COMPETITION on small batches that want no witness
  bool sweep = false;
  uint32_t max_segs = 1, seg_target = 0, cut_open = 0, n_dom = 1;
  DevBuf<uint32_t> d_cuts, d_seglist;
  DevBuf<SegResult> d_sres;
  HostBuf<SegResult> seg_host;
  // the RELAXED sweep in front of a count-form search (round 5; oracle/sweep_ref.c sweep_set_relaxed, jit_sweep_wg_impl.h RLX): a handful
  // of histories with crashed calls, nobody asking for a witness or a schedule -- every class of crashed calls an unlimited supply, so
  // the sweep's cuts apply and an INVALID history is refuted by hundreds of wavefronts in milliseconds instead of by one wavefront's
  // exhaustion of the relaxed config space (0.7 / 2.0 s on the bench's tiers); the prefix search then pins the failing completion as before
  bool rsweep = false;
  DevBuf<uint32_t> d_zncr, d_reach, d_reach_hdr, d_abort;
  hipStream_t stream2 = nullptr;    // ... on a stream of its own, beside the exact search (the context's when borrowed)
  hipEvent_t ev2[2] = {};
  uint32_t last_segments = 0, last_fallback = 0;
  uint32_t shard_rank = 0, shard_world = 1;      // tbc_batch_set_shard: this rank's share of the sweep's wavefronts
  bool partial_done = false;                     // a tbc_batch_sweep_partial is waiting for its tbc_batch_sweep_finish
  HostBuf<Hist> hist_back_m;                     // descriptors as the pack kernels left them (kept between
  HostBuf<BeamHist> bh_back_m;                   //   tbc_batch_sweep_partial and tbc_batch_sweep_finish)
  HostBuf<char> upload_stage;       // tbc_check: the block create uploads (pinned, the context's)
  bool inputs_fresh = false;        // tbc_check: create has just uploaded hist / bh / work with the columns -- the first run does not again
  uint32_t rules = 0;               // kRuleEager | kRuleTwin: wide single-wave schedule, register family, values 0..kMaxRuleValue
  uint32_t vpad = 0;                // entries per rdm row (nil + values), power of two
  DevBuf<uint64_t> d_twn, d_rdm;    // dominance tables (tbc_internal.h)
  DevBuf<uint64_t> d_occ, d_btab, d_pool;
  // count form (tbc_internal.h, kRuleCount): crashed calls as counts per class; the classes and their members, per history
  bool count_form = false;
  DevBuf<uint64_t> d_cmem;
  std::vector<CountHist> count_hist;       // (kept for the result marshalling: which crashed calls a count vector stands for)
  uint32_t reg_rules() const { return rules & (kRuleEager | kRuleTwin); }     // the register family's rules (their tables: twn, rdm)
  uint32_t epoch = 0;               // narrow kernel: the pass number its visited-set keys are tagged with (1..255; the arena is zeroed when it wraps)
  bool any_crashed = true;          // some op of the batch never completes (else the crashed-call arena is never read: one element)
  // u64 words per entry of the batch's own visited-set arena: the narrow kernel keeps no parent links when nobody wants a witness
  uint32_t tab_stride() const { return entry_words() - ((lanes && !opts.want_witness) ? 1u : 0u); }
  uint32_t entry_words() const { return mask_words + 2u + (count_form ? kCountWords : 0u); }   // u64 words per wide-schedule entry
  DevBuf<unsigned long long> d_pool_cursor;
  // last run
  std::vector<DevResult> res_host;
  std::vector<uint32_t> witness_host;
  uint64_t timing_ns[4] = {0, 0, 0, 0};
  hipEvent_t ev_turn = nullptr;     // narrow kernel: the search's turn on the device has come (SearchTurn), owned by the batch
  uint64_t turn_wait_ns = 0;        // ... and how long the last run waited for it
  tbc_counters sum{};
  uint64_t device_bytes = 0;

  bool borrowed = false;            // stream and events belong to a persistent context (tbc_check)
  ~tbc_batch() {
    d_f.release(); d_a.release(); d_b.release(); d_proc.release(); d_inv.release(); d_ret.release();
    d_hist.release(); d_rec.release(); d_seg.release(); d_ret_slot.release(); d_ret_op.release();
    d_bitmap.release(); d_wpre.release(); d_frames.release(); d_witness.release(); d_work.release();
    d_queue.release(); d_tab.release(); d_results.release(); d_table.release(); d_pool_vals.release(); d_cfg.release();
    d_bh.release(); d_off.release(); d_ncr.release(); d_lst.release(); d_crashed.release(); d_stack.release();
    d_zncr.release(); d_reach.release(); d_reach_hdr.release(); d_abort.release();
    if (!borrowed) { for (auto& e : ev2) if (e) (void)hipEventDestroy(e); if (stream2) (void)hipStreamDestroy(stream2); }
    d_cmem.release(); d_occ.release(); d_btab.release(); d_slot8.release(); d_rk8.release(); d_look.release(); d_looktmp.release(); d_twn.release(); d_rdm.release(); d_cuts.release(); d_seglist.release(); d_sres.release(); d_dstack.release(); d_pool.release(); d_pool_cursor.release();
    if (ev_turn) (void)hipEventDestroy(ev_turn);
    if (!borrowed) {
      for (auto& e : ev) if (e) (void)hipEventDestroy(e);
      if (stream) (void)hipStreamDestroy(stream);
    }
  }
};

extern "C" {

int32_t tbc_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  int k = 0;
  for (int d = 0; d < n; d++) if (device_is_gfx950(d)) k++;
  return k;
}

static PackArgs make_pack_args(tbc_batch* B);
static PackOpenArgs make_pack_open_args(tbc_batch* B);

static tbc_status batch_create_impl(const tbc_batch_desc* desc, const tbc_model* model,
                                    const tbc_opts* opts, tbc_batch* B) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    set_error("no HIP device visible; libtbcheck has no CPU fallback");
    return TBC_ERR_NO_DEVICE;
  }
  B->opts = *opts;
  // several histories per wavefront expand one config per iteration: search_width 1 next to a named lanes_per_history means what
  // 0 means (tbcheck.h says "leave search_width 0 or 1"), not the sequential knossos.wgl kernel
  if (opts->lanes_per_history != 0 && opts->lanes_per_history != 64 && opts->search_width == 1) B->opts.search_width = 0;
  opts = &B->opts;
  B->model = *model;
  B->device = (int)opts->device;
  if (B->device >= ndev || !device_is_gfx950(B->device)) {
    set_error("device %d is not a gfx950 (MI355X) device", B->device);
    return TBC_ERR_NO_DEVICE;
  }
  HIP_TRY(hipSetDevice(B->device));
  switch (model->kind) {
    case TBC_MODEL_REGISTER: case TBC_MODEL_CAS_REGISTER: break;
    case TBC_MODEL_MUTEX:          // tbc_model.init: 0 = free, 1 = held (knossos.model/mutex starts free)
      if (model->init != 0 && model->init != 1) { set_error("mutex: init must be 0 (free) or 1 (held)"); return TBC_ERR_MODEL; }
      break;
    case TBC_MODEL_SET: case TBC_MODEL_BANK:
      if (!desc->cols.pool || desc->cols.pool_len == 0) { set_error("set / bank models need the value pool (see knossos/_analysis.py)"); return TBC_ERR_INVALID_ARG; }
      if (model->kind == TBC_MODEL_BANK && (model->n_keys == 0 || model->n_keys > 16)) { set_error("bank: 1..16 accounts"); return TBC_ERR_MODEL; }
      if (model->kind == TBC_MODEL_BANK && (model->flags & TBC_MODEL_F_NO_NEGATIVE)) { set_error("bank with :negative-balances? false does not commute: use the memo table"); return TBC_ERR_UNSUPPORTED; }
      break;
    case TBC_MODEL_MULTI_REGISTER:
      if (model->n_keys == 0 || model->n_keys > 8) { set_error("multi-register: 1..8 keys on the device (more: use the memo table)"); return TBC_ERR_MODEL; }
      if (desc->cols.pool_len && !desc->cols.pool) { set_error("multi-register needs the value pool"); return TBC_ERR_INVALID_ARG; }
      break;
    case TBC_MODEL_TABLE:
      if (!model->table || model->n_states == 0 || model->n_classes == 0 || model->n_states > 0xFFFEu) {
        set_error("table model needs table, n_states, n_classes");
        return TBC_ERR_MODEL;
      }
      if (model->init < 0 || (uint32_t)model->init >= model->n_states) { set_error("table model: bad init state"); return TBC_ERR_MODEL; }
      break;
    default:
      set_error("model kind %u is not implemented by this build", model->kind);
      return TBC_ERR_UNSUPPORTED;
  }
  if (opts->algorithm > TBC_ALG_LINEAR) { set_error("unknown algorithm %u", opts->algorithm); return TBC_ERR_INVALID_ARG; }

  const uint32_t nh = desc->n_hist;
  B->n_hist = nh;
  TRACE("create: begin");
  B->total_ops = desc->op_off[nh];
  if (B->total_ops != desc->cols.n) { set_error("op_off[n_hist] (%llu) != cols.n (%u)", (unsigned long long)B->total_ops, desc->cols.n); return TBC_ERR_INVALID_ARG; }
  // the offsets index the op columns from here on (width heuristic, value scan, sweep sizing): check them first
  for (uint32_t h = 0; h < nh; h++) {
    if (desc->op_off[h + 1] < desc->op_off[h] || desc->op_off[h + 1] > B->total_ops || desc->op_off[h + 1] - desc->op_off[h] > 0x7FFFFFFFull) {
      set_error("history %u: bad op_off", h);
      return TBC_ERR_INVALID_ARG;
    }
  }
  if (desc->op_off[0] != 0) { set_error("op_off[0] must be 0"); return TBC_ERR_INVALID_ARG; }
  // ---- count form (tbc_internal.h, kRuleCount): a register / cas-register batch with crashed calls that have an effect, under the
  // default rules and knossos.competition (the published orders -- TBC_ALG_WGL, TBC_ALG_LINEAR -- keep a mask bit per crashed call).
  // The process column is re-numbered (re-used slots; a crashed call holds none) and the crashed calls become classes with counts.
  std::vector<int32_t> slot_col;
  const ColumnScan scan = scan_columns(desc->cols, B->total_ops);
  B->any_crashed = scan.any_crashed;
  {
    const bool regfam = model->kind == TBC_MODEL_REGISTER || model->kind == TBC_MODEL_CAS_REGISTER;
    const tbc_ops& c = desc->cols;
    bool want = regfam && opts->algorithm == TBC_ALG_COMPETITION && (opts->dominance & (TBC_DOM_NO_EAGER_READS | TBC_DOM_NO_TWIN_RULE | TBC_DOM_NO_COUNT_FORM)) == 0 &&
                opts->search_width != 1 && opts->lanes_per_history != 4 && opts->lookahead != 1 &&      // (several histories per wavefront: 8 / 16 / 32 lanes in the count form)
                (model->init == TBC_NIL || (model->init >= 0 && model->init <= kMaxRuleValue));
    want = want && scan.nonneg && scan.vmax <= kMaxRuleValue;
    const bool any = scan.any_crashed_effect;
    if (want && any) {
      slot_col.resize((size_t)B->total_ops + 1);
      B->count_hist.resize(nh);
      std::vector<uint32_t> rets; std::vector<int32_t> slot_of; std::vector<uint8_t> used;
      bool ok = true;
      for (uint32_t h = 0; h < nh && ok; h++)
        ok = build_count_form(c, desc->op_off[h], desc->op_off[h + 1] - desc->op_off[h], desc->n_process[h], model->kind == TBC_MODEL_CAS_REGISTER,
                              slot_col.data() + desc->op_off[h], B->count_hist[h], rets, slot_of, used);
      B->count_form = ok;
      if (!ok) { slot_col.clear(); B->count_hist.clear(); }
    }
  }
  const auto slots_of = [&](uint32_t h) -> uint32_t { return B->count_form ? B->count_hist[h].n_slots : desc->n_process[h]; };
  uint32_t maxW = 1;
  for (uint32_t h = 0; h < nh; h++) maxW = std::max(maxW, slots_of(h));
  if (maxW > kMaxSlots) { set_error("%u open processes > %u supported", maxW, kMaxSlots); return TBC_ERR_WINDOW_TOO_WIDE; }
  uint32_t mw = (maxW + 63) / 64;
  B->mask_words = mw <= 1 ? 1 : mw <= 2 ? 2 : mw <= 4 ? 4 : mw <= 8 ? 8 : 16;
  B->frame_words = search_frame_words(B->mask_words);
  if (B->count_form && B->mask_words > 2) { B->count_form = false; slot_col.clear(); B->count_hist.clear(); }   // (the count form's kernel: one or two mask words)
  if (!B->count_form && maxW != 1) {        // (the masks are the mask form's after all)
    maxW = 1;
    for (uint32_t h = 0; h < nh; h++) maxW = std::max(maxW, desc->n_process[h]);
    if (maxW > kMaxSlots) { set_error("%u open processes > %u supported", maxW, kMaxSlots); return TBC_ERR_WINDOW_TOO_WIDE; }
    mw = (maxW + 63) / 64;
    B->mask_words = mw <= 1 ? 1 : mw <= 2 ? 2 : mw <= 4 ? 4 : mw <= 8 ? 8 : 16;
    B->frame_words = search_frame_words(B->mask_words);
  }
  const uint32_t KW = 1 + B->mask_words;
  uint32_t width = opts->search_width ? opts->search_width : (opts->algorithm == TBC_ALG_WGL ? 1u : 4u);   // 4: fewest rounds per history, measured
  if (width > 16) width = 16;           // one wavefront per history: at most 16 configs per round
  while (width & (width - 1)) width &= width - 1;   // the wide kernels take a power of two
  if (B->mask_words > 4) width = 1;          // very wide windows: sequential kernel only
  const bool commutative = model->kind == TBC_MODEL_SET || model->kind == TBC_MODEL_BANK;
  if (commutative) {                          // state-free models exist in the wide kernel only
    if (B->mask_words > 4) { set_error("set / bank: at most 256 processes (incl. crashed) on the device"); return TBC_ERR_WINDOW_TOO_WIDE; }
    if (width < 2) width = 4;
  }
  // knossos.linear = the level sweep; knossos.competition takes it when nobody asked for a witness or a
  // particular schedule and the batch is small enough to be latency-bound (a big batch is throughput-bound:
  // the wide depth-first kernel does less work per history).  It needs a state-carrying model and <= 64 slots;
  // a history it cannot finish (a level outgrows LDS) goes to the wide kernel.
  {
    const char* env = std::getenv("TBC_SWEEP");          // 0 = never, 1 = whenever possible (experiments)
    const bool forced = env && env[0] == '1', never = env && env[0] == '0';
    const bool asked = opts->algorithm == TBC_ALG_LINEAR ||
                       (opts->algorithm == TBC_ALG_COMPETITION && !opts->want_witness && opts->search_width == 0 && nh <= 256 &&
                        (opts->lanes_per_history == 0 || opts->lanes_per_history == 64));      // (a named depth-first schedule is not the sweep)
    B->sweep = !never && (asked || forced) && !commutative && B->mask_words == 1 && width <= 16 && !B->count_form;   // (the sweep's segments cannot start from count vectors)
    if (B->sweep && width < 2) width = 4;                // the fallback's schedule; the per-front lists are the wide kernel's
  }
  if (commutative && !(opts->dominance & TBC_DOM_NO_LAZY_COMMUTING)) B->rules |= kRuleLazyComm;
  B->width = width;
  B->rsweep = B->count_form && !B->sweep && opts->algorithm == TBC_ALG_COMPETITION && !opts->want_witness && opts->search_width == 0 &&
              opts->max_steps == 0 && nh <= 8 && (opts->lanes_per_history == 0 || opts->lanes_per_history == 64) && B->mask_words == 1 &&
              width > 1 && width <= 16 && (model->kind == TBC_MODEL_REGISTER || model->kind == TBC_MODEL_CAS_REGISTER);
  B->lookahead = !B->sweep && width > 1 && width <= 16 && opts->lookahead != 1 &&
                 (model->kind == TBC_MODEL_REGISTER || model->kind == TBC_MODEL_CAS_REGISTER);
  const bool beam = width > 1;
  // dominance rules: same scope as the lookahead, and every register value must index the per-front read table
  if (B->lookahead || (width > 1 && width <= 16 && (model->kind == TBC_MODEL_REGISTER || model->kind == TBC_MODEL_CAS_REGISTER))) {
    const int32_t vmax = std::max(model->init == TBC_NIL ? -1 : model->init, scan.vmax);
    const bool in_range = (model->init == TBC_NIL || model->init >= 0) && scan.nonneg;
    if (in_range && vmax <= kMaxRuleValue) {
      B->n_dom = (uint32_t)(vmax + 2);                   // nil + 0..vmax: the states a register can be in
      B->rules = ((opts->dominance & TBC_DOM_NO_EAGER_READS) ? 0u : kRuleEager) | ((opts->dominance & TBC_DOM_NO_TWIN_RULE) ? 0u : kRuleTwin);
      if (B->count_form) B->rules |= kRuleCount;
      B->vpad = 2; while (B->vpad < (uint32_t)(vmax + 2)) B->vpad <<= 1;
    }
  }
  // Nobody named a width: 4 configs per round, or 2 where that is measured faster -- a register / cas-register batch
  // under both dominance rules at low concurrency, where the depth-first order rarely backtracks and the third and
  // fourth config of a round are mostly expanded in vain (32,768 histories at 6.4 calls in flight: 5.6*10^8 probes and
  // 171 ms against 1.07*10^9 and 198 ms; at 19 in flight 4 is 9 % faster; profiles/r02_k5_width_ab.txt).  Calls in
  // flight are averaged over a sample of the batch's histories: positions from invocation to completion (a crashed
  // call stays open to the end) over the history's length.
  if (B->count_form && !(B->rules & kRuleCount)) { set_error("internal: count form without the rules"); return TBC_ERR_HIP; }
  if (opts->search_width == 0 && B->width == 4 && (B->rules & ~kRuleCount) == (kRuleEager | kRuleTwin)) {
    uint64_t open_sum = 0, events = 0;
    const uint32_t stride = std::max<uint32_t>(1, nh / 64);
    for (uint32_t h = 0; h < nh; h += stride) {
      const uint32_t ne = desc->n_events[h];
      for (uint64_t i = desc->op_off[h]; i < desc->op_off[h + 1]; i++) {
        const uint32_t inv = desc->cols.inv_pos[i], ret = desc->cols.ret_pos[i];
        if (B->count_form && ret == TBC_POS_CRASHED) continue;                               // (count form: a crashed call is no open call)
        open_sum += (ret == TBC_POS_CRASHED || ret > ne ? ne : ret) - std::min(inv, ne);   // malformed rows are the pack kernel's to reject
      }
      events += ne;
    }
    if (events && open_sum <= 10 * events) B->width = 2;
  }
  // Several histories per wavefront (tbc_opts.lanes_per_history).  Asked for by name it must be possible; left to the
  // library it is taken for a big register-family batch at low concurrency under both rules (the batch the width-2 choice
  // above is made for): a wavefront then carries 8 searches instead of one whose rounds fill 4 of its 64 lanes.
  {
    const uint32_t asked = opts->lanes_per_history;
    if (asked != 0 && asked != 4 && asked != 8 && asked != 16 && asked != 32 && asked != 64) { set_error("lanes_per_history must be 0, 4, 8, 16, 32 or 64"); return TBC_ERR_INVALID_ARG; }
    if (opts->list_order > 3 && (opts->list_order < 16 || opts->list_order > 16 + 4096)) { set_error("tbc_opts.list_order must be TBC_ORDER_* or 16 + W, W <= 4096"); return TBC_ERR_INVALID_ARG; }
    const bool regfam3 = model->kind == TBC_MODEL_REGISTER || model->kind == TBC_MODEL_CAS_REGISTER || model->kind == TBC_MODEL_MUTEX;
    // the narrow kernel addresses a history's tables with 32-bit element offsets, and its visited-set keys hold front + 1 in 24 bits
    // (bits 24-31 of the low word are the pass's epoch tag, wgl_narrow_impl.h kFrontMask / entry_empty): a history of 2^24 completions
    // or more would have its fronts truncated -- such a batch keeps a wavefront per history
    uint64_t longest = 0;
    for (uint32_t h = 0; h < nh; h++) longest = std::max<uint64_t>(longest, desc->op_off[h + 1] - desc->op_off[h]);
    const bool can = beam && !B->sweep && longest < kNarrowMaxOps && (!B->count_form || B->mask_words <= 2) && regfam3 && narrow_supported(B->mask_words, 8) && opts->algorithm != TBC_ALG_WGL &&
                     look_words(B->total_ops, nh, B->mask_words) < (1ull << 32);
    if (asked != 0 && asked != 64) {
      if (!can) { set_error("lanes_per_history %u: needs the depth-first search of a register / cas-register / mutex batch with at most 256 process slots and fewer than 2^24 - 16 ops per history (not TBC_ALG_WGL, not the level sweep)", asked); return TBC_ERR_UNSUPPORTED; }
      if (opts->search_width > 1) { set_error("lanes_per_history %u expands one config per iteration: leave search_width 0 or 1", asked); return TBC_ERR_INVALID_ARG; }
      B->lanes = asked;
    } else if (asked == 0 && can && !B->count_form && opts->search_width == 0 && B->width == 2 && nh >= 24576) {      // (count form: by name only until measured)
      // measured (profiles/r03_narrow_batch_sizes.log): 8 lanes per history lose to a wavefront each at 4,096 and 8,192
      // histories (59 / 61 ms against 40 / 47: one round of the narrow kernel is ~10 us of dependent instructions and trips
      // whatever the load, so it needs three or four wavefronts per SIMD to hide it), tie at 16,384, win 97 against 160 ms at 32,768
      B->lanes = 8;
    }
    // under the eager rule the narrow kernel branches over :write / :cas only: lists without reads, root in normal form
    // (the count form's schedule, oracle/wgl_count.c, keeps the full lists and the root as given)
    if (B->lanes && (B->rules & kRuleEager) && !B->count_form) B->rules |= kRuleBranch;
  }
  const uint32_t EW = B->entry_words();   // u64 words per wide-schedule entry
  // the frames arena is the pack kernels' scratch (3 words per op) and the sequential kernel's stack (4 + 2 mask words per op): a
  // wide-schedule batch only needs the former -- the rare history that falls back to the sequential kernel gets frames of its own then
  if (beam) B->frame_words = 3;
  if (B->sweep || B->rsweep) {
    // segments: enough wavefronts to fill the GPU several times over, none shorter than 32 completions; cuts need the
    // register family's value domain (nil + 0..vmax = vpad's range) to enumerate the configs possible at a front
    uint64_t max_n = 1;
    for (uint32_t h = 0; h < nh; h++) max_n = std::max<uint64_t>(max_n, desc->op_off[h + 1] - desc->op_off[h]);
    const char* env = std::getenv("TBC_SWEEP_SEG");
    uint64_t T = env ? std::strtoull(env, nullptr, 10) : std::max<uint64_t>(32, (max_n * nh + 4095) / 4096);
    // one history or a handful -- the workgroup kernel's case (at most 4,096 workgroups): windows of 48 completions.  Measured round 5 with
    // the compact walk (profiles/r05_sweep_segment_length.txt): one 10k-op history 1.07 ms at 32, 0.97 - 1.02 at 40, 0.97 - 0.99 at 48,
    // 1.11 at 56, 1.20 at 64 (fewer workgroups, shorter table to bring back and compose; past 48 the longest segment costs more than that saves)
    if (!env && T < 48 && (uint64_t)nh * ((max_n + 47) / 48) * kSweepSlices <= 4096) T = 48;
    const bool regfam = model->kind == TBC_MODEL_REGISTER || model->kind == TBC_MODEL_CAS_REGISTER;
    if (!regfam || B->vpad == 0 || T == 0 || T >= max_n) { B->seg_target = 0; B->max_segs = 1; }
    else {
      B->seg_target = (uint32_t)T;
      B->max_segs = (uint32_t)std::min<uint64_t>(kSweepMaxSegs, (max_n + T - 1) / T);
      // the last window takes whatever the cap leaves over: raise T if the cap bites
      while ((uint64_t)B->max_segs * B->seg_target < max_n) B->seg_target++;
      B->cut_open = 0;
      while (B->cut_open < 4 && (B->n_dom << (B->cut_open + 1)) <= 32 * kSweepSlices) B->cut_open++;
    }
  }

  const uint64_t default_cap_bytes = 1ull << 30;
  const uint64_t max_bytes = opts->max_visited_bytes ? opts->max_visited_bytes : default_cap_bytes;
  B->hist.resize(nh);
  if (beam) B->bh.resize(nh);
  uint64_t boff_n = 0, bocc_n = 0, blst_n = 0, bstack_n = 0, btab_n = 0;
  std::vector<uint32_t> rank_scratch;
  uint64_t rec_n = 0, seg_n = 0, bm_n = 0, frame_n = 0, tab_n = 0;
  // how many entries each history's per-front lists hold.  A few histories (tbc_check: latency matters): a pass over each
  // history's events on the host.  A big batch: that pass was 1.5 of tbc_batch_create's 1.7 s for 32,768 histories
  // (profiles/r04_cold_batch.log) -- the pack and counts kernels, which every run launches anyway, say the same numbers in 30 ms once the inputs are
  // resident (device_sizing below), and the list arenas are allocated after that.
  const bool device_sizing = beam && nh > 64;
  std::vector<uint32_t> list_caps;
  if (beam && !device_sizing) {
    list_caps.assign(nh, 0u);
    const bool branch = (B->rules & kRuleBranch) != 0;
    for (uint32_t h = 0; h < nh; h++) {
      const uint64_t n = desc->op_off[h + 1] - desc->op_off[h];
      // (tbc_check: the arenas are pieces of a slab that is there already -- where the worst case, every slot at every front, is a few MB,
      // take it and skip the pass over the history's events: 20 us of a 1 ms call)
      const uint64_t worst = std::max<uint64_t>(n, 1) * std::max(1u, slots_of(h));
      if (t_ctx && worst * (sizeof(OpRec) + 8 * B->mask_words) <= (24ull << 20)) list_caps[h] = (uint32_t)worst;
      else list_caps[h] = (uint32_t)std::min<uint64_t>(0xFFFFFFF0ull, open_list_entries(desc->cols, desc->op_off[h], n, desc->n_events[h], std::max(1u, slots_of(h)), rank_scratch, branch));
    }
  }
  TRACE("create: lists sized");
  for (uint32_t h = 0; h < nh; h++) {
    Hist& H = B->hist[h];
    std::memset(&H, 0, sizeof H);
    const uint64_t n = desc->op_off[h + 1] - desc->op_off[h];
    if (desc->op_off[h + 1] < desc->op_off[h] || n > 0x7FFFFFFFull) { set_error("history %u: bad op_off", h); return TBC_ERR_INVALID_ARG; }
    H.op_off = desc->op_off[h];
    H.n_ops = (uint32_t)n;
    B->max_ops = std::max<uint64_t>(B->max_ops, n);
    H.n_events = desc->n_events[h];
    H.n_slots = std::max(1u, slots_of(h));
    H.flags = B->count_form ? kHistCount : 0u;
    H.aux = desc->model_aux ? desc->model_aux[h] : model->init;
    H.rec_off = rec_n; rec_n += n + 2ull * H.n_slots;
    H.seg_off = seg_n; seg_n += H.n_slots + 1;
    H.ret_off = H.op_off;
    H.bm_off = bm_n; bm_n += H.n_events / 32 + 1;
    H.frame_off = frame_n; frame_n += std::max<uint64_t>(n, 1) * B->frame_words;
    const uint64_t per_op = opts->visited_per_op ? opts->visited_per_op : 64;
    uint32_t lg = std::max(10u, ceil_log2(per_op * std::max<uint64_t>(n, 1)));
    while (lg > 10 && (1ull << lg) * KW * 8 > max_bytes) lg--;
    H.tab_log2 = lg;
    H.tab_off = tab_n;
    if (!beam) { tab_n += (1ull << lg) * KW; tab_n = (tab_n + 1) & ~1ull; }   // keep 16 B alignment
    if (beam) {
      BeamHist& Q = B->bh[h];
      std::memset(&Q, 0, sizeof Q);
      uint32_t blg = lg;
      while (blg > 10 && ((1ull << blg) * EW * 8 > max_bytes || blg > kBeamMaxTabLog2)) blg--;
      Q.tab_log2 = blg;
      Q.off_off = boff_n; boff_n += n + 2;
      if (B->count_form) {
        const CountHist& ch = B->count_hist[h];
        Q.cmem_off = bocc_n; bocc_n += ch.words.size();
        Q.n_classes = ch.n_classes; Q.top[0] = ch.top[0]; Q.top[1] = ch.top[1];
      }
      Q.lst_cap = device_sizing ? 0xFFFFFFF0u : list_caps[h];
      Q.lst_off = blst_n; blst_n += device_sizing ? 0u : Q.lst_cap;
      Q.stack_off = bstack_n; bstack_n += (1ull << blg);
      Q.tab_off = btab_n; btab_n += (1ull << blg);
    (void)EW;
    }
  }

  // the narrow kernel addresses lists and fronts with 32-bit element offsets (wgl_narrow_impl.h): a batch past that keeps a wavefront per history
  const auto lists_too_long = [&]() -> bool { return B->lanes && (blst_n >= (1ull << 32) || boff_n >= (1ull << 32)); };
  if (!device_sizing && lists_too_long()) {
    if (opts->lanes_per_history) { set_error("lanes_per_history: the batch's open-call lists exceed 2^32 entries; split the batch"); return TBC_ERR_UNSUPPORTED; }
    const bool had_branch = (B->rules & kRuleBranch) != 0;
    B->lanes = 0; B->rules &= ~kRuleBranch;
    if (had_branch) {                          // the lists hold the reads again: size them for that
      blst_n = 0;
      for (uint32_t h = 0; h < nh; h++) {
        const Hist& H = B->hist[h];
        BeamHist& Q = B->bh[h];
        Q.lst_cap = (uint32_t)std::min<uint64_t>(0xFFFFFFF0ull, open_list_entries(desc->cols, H.op_off, H.n_ops, H.n_events, H.n_slots, rank_scratch, false));
        Q.lst_off = blst_n; blst_n += Q.lst_cap;
      }
    }
  }
  if (device_sizing && B->lanes && boff_n >= (1ull << 32)) {
    if (opts->lanes_per_history) { set_error("lanes_per_history: the batch exceeds 2^32 fronts; split the batch"); return TBC_ERR_UNSUPPORTED; }
    B->lanes = 0; B->rules &= ~kRuleBranch;
  }
  if (B->sweep) { bstack_n = 0; btab_n = 0; }     // the sweep has no visited set; its fallback takes scratch arenas
  TRACE("create: layout done");
  tbc_status s;
  const uint64_t T = B->total_ops;
  // (the order matters to tbc_check, whose arenas are consecutive pieces of its context's slab: what is uploaded -- the six columns,
  // the descriptors, the work list -- first, as one block (upload_block below); then what every run zeroes, as one memset (zero_block))
  if ((s = B->d_f.alloc(T)) || (s = B->d_a.alloc(T)) || (s = B->d_b.alloc(T)) || (s = B->d_proc.alloc(T)) ||
      (s = B->d_inv.alloc(T)) || (s = B->d_ret.alloc(T)) || (s = B->d_hist.alloc(nh)) || (s = B->d_bh.alloc(beam ? nh : 0)) || (s = B->d_work.alloc(nh)) ||
      (s = B->d_bitmap.alloc(bm_n)) || (s = B->d_off.alloc(beam ? boff_n : 0)) || (s = B->d_ncr.alloc(beam ? boff_n : 0)) || (s = B->d_pool_cursor.alloc(1)) ||
      (s = B->d_rec.alloc(rec_n)) ||
      (s = B->d_seg.alloc(seg_n)) || (s = B->d_ret_slot.alloc(T)) || (s = B->d_ret_op.alloc(T)) ||
      (s = B->d_wpre.alloc(bm_n)) || (s = B->d_frames.alloc(frame_n)) ||
      (s = B->d_tab.alloc(tab_n)) || (s = B->d_results.alloc(nh)) ||
      (s = B->d_queue.alloc(4)) || (s = B->d_witness.alloc(opts->want_witness ? T : 0)))
    return s;
  if (beam) {
    if ((!device_sizing && (s = B->d_lst.alloc(blst_n))) || (s = B->d_crashed.alloc((B->count_form || !B->any_crashed) ? 0 : T)) || (s = B->d_cmem.alloc(B->count_form ? bocc_n : 0)) ||
        (s = B->d_slot8.alloc(slot8_bytes(T, nh))) || (s = B->d_stack.alloc(bstack_n)) || (s = B->d_btab.alloc(btab_n * B->tab_stride())))
      return s;
    if ((B->sweep || B->rsweep) && ((s = B->d_cuts.alloc((uint64_t)nh * B->max_segs)) || (s = B->d_sres.alloc((uint64_t)nh * B->max_segs * kSweepSlices)) || (s = B->d_seglist.alloc((uint64_t)nh * B->max_segs * kSweepSlices * 3)))) return s;
    if (B->sweep || B->rsweep) B->seg_host.resize((size_t)nh * B->max_segs * kSweepSlices);
    if (B->rsweep) {          // no crashed call is a candidate of its own (a zero ncr[]); the classes' reach tables (reach_table.h)
      std::vector<uint32_t> reach, hdr;
      for (uint32_t h = 0; h < nh; h++) {
        const CountHist& ch = B->count_hist[h];
        hdr.push_back((uint32_t)reach.size());
        hdr.push_back(build_reach_table(ch.words.data(), ch.n_classes, reach));
      }
      if ((s = B->d_zncr.alloc(boff_n)) || (s = B->d_reach.alloc(reach.size())) || (s = B->d_reach_hdr.alloc(hdr.size())) || (s = B->d_abort.alloc(nh))) return s;
      HIP_TRY(hipMemset(B->d_zncr.p, 0, B->d_zncr.bytes()));
      HIP_TRY(hipMemcpy(B->d_reach.p, reach.data(), reach.size() * 4, hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy(B->d_reach_hdr.p, hdr.data(), hdr.size() * 4, hipMemcpyHostToDevice));
    }
    if (B->lanes && (s = B->d_rk8.alloc(slot8_bytes(T, nh)))) return s;
    if (B->reg_rules() && ((!device_sizing && (s = B->d_twn.alloc(blst_n * B->mask_words))) || (s = B->d_rdm.alloc(B->lanes ? 1 : T * B->vpad * B->mask_words)))) return s;
    // several histories per wavefront: front records (tbc_internal.h) instead of plain rows, with or without the rules
    if (B->lanes) { B->d_rdm.release(); if ((s = B->d_rdm.alloc(T * B->front_words()))) return s; }
    // (d_looktmp: scratch of the walk with lane = process slot only -- launch_pack_open's choice, repeated here)
    const bool by_front = B->mask_words == 1 && B->vpad <= 32;
    if (B->lookahead && ((s = B->d_look.alloc(look_words(T, nh, B->mask_words))) || (s = B->d_looktmp.alloc(by_front ? 0 : T)) ||
                         (s = B->d_dstack.alloc(bstack_n)))) return s;
    // growth pool: 30 % of the visited-set arena -- 10 % for the big quiet batches that run several histories per wavefront, whose sets
    // rarely grow (a history that outgrows its table and finds the pool empty is run again from a scratch arena: at 32 calls in
    // flight a 10 % pool cost 22 s of such retries per 2,048 histories) --, at least room for one history to grow twice (4x, then 16x: keys, parents, two stacks, slot translation), at most 32 GiB
    {
      uint64_t biggest = 0;
      for (uint32_t h = 0; h < nh; h++) biggest = std::max<uint64_t>(biggest, 1ull << B->bh[h].tab_log2);
      uint64_t words = std::max<uint64_t>(btab_n * B->tab_stride() * (B->lanes ? 1u : 3u) / 10, biggest * (4 + 16 + 4) * (EW + 1));
      words = std::min<uint64_t>(words, (32ull << 30) / 8);
      if (B->sweep) words = 1;
      if ((s = B->d_pool.alloc(words))) return s;
    }
  }
  if ((s = B->d_cfg.alloc((uint64_t)nh * kCfgCap * (2 + B->mask_words)))) return s;
  B->pool_len = desc->cols.pool ? desc->cols.pool_len : 0;
  if ((s = B->d_pool_vals.alloc(B->pool_len))) return s;
  if (B->pool_len) HIP_TRY(hipMemcpy(B->d_pool_vals.p, desc->cols.pool, (size_t)B->pool_len * 4, hipMemcpyHostToDevice));
  if (model->kind == TBC_MODEL_TABLE) {
    const size_t tn = (size_t)model->n_states * model->n_classes;
    B->table_host.assign(model->table, model->table + tn);
    if ((s = B->d_table.alloc(tn))) return s;
    HIP_TRY(hipMemcpy(B->d_table.p, B->table_host.data(), tn * 2, hipMemcpyHostToDevice));
    B->model.table = nullptr;
  }
  B->device_bytes = B->d_f.bytes() + B->d_a.bytes() + B->d_b.bytes() + B->d_proc.bytes() + B->d_inv.bytes() +
                    B->d_ret.bytes() + B->d_hist.bytes() + B->d_rec.bytes() + B->d_seg.bytes() + B->d_ret_slot.bytes() +
                    B->d_ret_op.bytes() + B->d_bitmap.bytes() + B->d_wpre.bytes() + B->d_frames.bytes() +
                    B->d_tab.bytes() + B->d_results.bytes() + B->d_work.bytes() + B->d_witness.bytes();
  if (beam) B->device_bytes += B->d_bh.bytes() + B->d_off.bytes() + B->d_ncr.bytes() + B->d_lst.bytes() +
                               B->d_crashed.bytes() + B->d_slot8.bytes() + B->d_rk8.bytes() + B->d_twn.bytes() + B->d_rdm.bytes() + B->d_look.bytes() + B->d_looktmp.bytes() + B->d_dstack.bytes() + B->d_stack.bytes() + B->d_btab.bytes() + B->d_pool.bytes() + B->d_cmem.bytes();

  if (t_ctx) {
    B->borrowed = true; B->stream = t_ctx->stream;
    for (int i = 0; i < 6; i++) B->ev[i] = t_ctx->ev[i];
    if (B->rsweep) {
      if (!t_ctx->stream2) {
        HIP_TRY(hipStreamCreateWithFlags(&t_ctx->stream2, hipStreamNonBlocking));
        for (auto& e : t_ctx->ev2) HIP_TRY(hipEventCreate(&e));
      }
      B->stream2 = t_ctx->stream2; B->ev2[0] = t_ctx->ev2[0]; B->ev2[1] = t_ctx->ev2[1];
    }
  } else {
    HIP_TRY(hipStreamCreateWithFlags(&B->stream, hipStreamNonBlocking));
    for (auto& e : B->ev) HIP_TRY(hipEventCreate(&e));
    if (B->rsweep) {
      HIP_TRY(hipStreamCreateWithFlags(&B->stream2, hipStreamNonBlocking));
      for (auto& e : B->ev2) HIP_TRY(hipEventCreate(&e));
    }
  }

  TRACE("create: arenas allocated");
  // inputs become resident
  std::vector<uint32_t> work(nh);
  for (uint32_t h = 0; h < nh; h++) work[h] = h;
  // tbc_check: columns, descriptors and work list are consecutive pieces of the context's slab -- staged in the context's pinned
  // region and uploaded as ONE copy that nobody waits for (the run's kernels follow it in stream order; the region lives until the
  // call ends).  Eight staged copies of pageable memory and a synchronize were 77 us of a 1.07 ms call.
  bool uploaded = false;
  if (t_ctx && T && !(B->count_form && bocc_n) && !device_sizing) {
    char* const base = (char*)B->d_f.p;
    const auto at = [&](const void* p) { return (size_t)((const char*)p - base); };
    const auto a256 = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const bool consecutive = !B->d_f.owned && !B->d_a.owned && !B->d_b.owned && !B->d_proc.owned && !B->d_inv.owned && !B->d_ret.owned && !B->d_hist.owned &&
                             !B->d_bh.owned && !B->d_work.owned && at(B->d_a.p) == a256(T) && at(B->d_b.p) == at(B->d_a.p) + a256(T * 4) &&
                             at(B->d_proc.p) == at(B->d_b.p) + a256(T * 4) && at(B->d_inv.p) == at(B->d_proc.p) + a256(T * 4) && at(B->d_ret.p) == at(B->d_inv.p) + a256(T * 4) &&
                             at(B->d_hist.p) == at(B->d_ret.p) + a256(T * 4) && at(B->d_bh.p) == at(B->d_hist.p) + a256(nh * sizeof(Hist)) &&
                             at(B->d_work.p) == at(B->d_bh.p) + a256(std::max<size_t>(beam ? nh : 0, 1) * sizeof(BeamHist));
    if (consecutive) {
      const size_t total = at(B->d_work.p) + a256((size_t)nh * 4);
      B->upload_stage.resize(total);
      if (B->upload_stage.own.empty()) {          // (pinned: else the plain copies below)
        char* st = B->upload_stage.data();
        std::memcpy(st, desc->cols.f, T);
        std::memcpy(st + at(B->d_a.p), desc->cols.a, T * 4);
        std::memcpy(st + at(B->d_b.p), desc->cols.b, T * 4);
        std::memcpy(st + at(B->d_proc.p), B->count_form ? slot_col.data() : desc->cols.process, T * 4);
        std::memcpy(st + at(B->d_inv.p), desc->cols.inv_pos, T * 4);
        std::memcpy(st + at(B->d_ret.p), desc->cols.ret_pos, T * 4);
        std::memcpy(st + at(B->d_hist.p), B->hist.data(), nh * sizeof(Hist));
        if (beam) std::memcpy(st + at(B->d_bh.p), B->bh.data(), nh * sizeof(BeamHist));
        std::memcpy(st + at(B->d_work.p), work.data(), (size_t)nh * 4);
        HIP_TRY(hipMemcpyAsync(base, st, total, hipMemcpyHostToDevice, B->stream));
        uploaded = true;
        B->inputs_fresh = true;
      }
    }
  }
  if (uploaded) { B->res_host.resize(nh); TRACE("create: inputs queued as one block"); return TBC_OK; }
  if (T) {
    HIP_TRY(hipMemcpyAsync(B->d_f.p, desc->cols.f, T, hipMemcpyHostToDevice, B->stream));
    HIP_TRY(hipMemcpyAsync(B->d_a.p, desc->cols.a, T * 4, hipMemcpyHostToDevice, B->stream));
    HIP_TRY(hipMemcpyAsync(B->d_b.p, desc->cols.b, T * 4, hipMemcpyHostToDevice, B->stream));
    HIP_TRY(hipMemcpyAsync(B->d_proc.p, B->count_form ? slot_col.data() : desc->cols.process, T * 4, hipMemcpyHostToDevice, B->stream));
    HIP_TRY(hipMemcpyAsync(B->d_inv.p, desc->cols.inv_pos, T * 4, hipMemcpyHostToDevice, B->stream));
    HIP_TRY(hipMemcpyAsync(B->d_ret.p, desc->cols.ret_pos, T * 4, hipMemcpyHostToDevice, B->stream));
  }
  std::vector<uint64_t> cmem_host;
  if (B->count_form && bocc_n) {
    cmem_host.reserve(bocc_n);
    for (uint32_t h = 0; h < nh; h++) cmem_host.insert(cmem_host.end(), B->count_hist[h].words.begin(), B->count_hist[h].words.end());
    HIP_TRY(hipMemcpyAsync(B->d_cmem.p, cmem_host.data(), cmem_host.size() * 8, hipMemcpyHostToDevice, B->stream));
  }
  HIP_TRY(hipMemcpyAsync(B->d_hist.p, B->hist.data(), nh * sizeof(Hist), hipMemcpyHostToDevice, B->stream));
  HIP_TRY(hipMemcpyAsync(B->d_work.p, work.data(), nh * 4, hipMemcpyHostToDevice, B->stream));
  HIP_TRY(hipStreamSynchronize(B->stream));
  TRACE("create: inputs resident");
  if (device_sizing) {
    // the pack and counts kernels over the resident inputs: BeamHist.lst_need = entries each history's per-front lists hold
    hipStream_t st = B->stream;
    for (int attempt = 0; attempt < 2; attempt++) {
      HIP_TRY(hipMemsetAsync(B->d_bitmap.p, 0, B->d_bitmap.bytes(), st));
      HIP_TRY(hipMemsetAsync(B->d_off.p, 0, B->d_off.bytes(), st));
      HIP_TRY(hipMemsetAsync(B->d_ncr.p, 0, B->d_ncr.bytes(), st));
      HIP_TRY(hipMemcpyAsync(B->d_bh.p, B->bh.data(), nh * sizeof(BeamHist), hipMemcpyHostToDevice, st));
      HIP_TRY(hipMemcpyAsync(B->d_hist.p, B->hist.data(), nh * sizeof(Hist), hipMemcpyHostToDevice, st));
      launch_pack(make_pack_args(B), st);
      HIP_TRY(hipGetLastError());
      launch_open_counts(make_pack_open_args(B), st);
      HIP_TRY(hipGetLastError());
      std::vector<BeamHist> back(nh);
      HIP_TRY(hipMemcpyAsync(back.data(), B->d_bh.p, nh * sizeof(BeamHist), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      blst_n = 0;
      for (uint32_t h = 0; h < nh; h++) {
        BeamHist& Q = B->bh[h];
        Q.lst_cap = std::max(1u, back[h].lst_need);
        Q.lst_off = blst_n; blst_n += Q.lst_cap;
      }
      if (!lists_too_long()) break;
      if (opts->lanes_per_history) { set_error("lanes_per_history: the batch's open-call lists exceed 2^32 entries; split the batch"); return TBC_ERR_UNSUPPORTED; }
      const bool had_branch = (B->rules & kRuleBranch) != 0;
      B->lanes = 0; B->rules &= ~kRuleBranch;
      if (!opts->want_witness) {                   // (a wavefront per history keeps parent links whatever the caller wants: the arena grows by them)
        B->device_bytes -= B->d_btab.bytes();
        B->d_btab.release();
        if ((s = B->d_btab.alloc(btab_n * B->tab_stride()))) return s;
        B->device_bytes += B->d_btab.bytes();
      }
      if (!had_branch) break;                      // (else the lists hold the reads again: counted once more)
      for (uint32_t h = 0; h < nh; h++) { B->bh[h].lst_cap = 0xFFFFFFF0u; B->bh[h].lst_off = 0; }
    }
    if ((s = B->d_lst.alloc(blst_n)) || (B->reg_rules() && (s = B->d_twn.alloc(blst_n * B->mask_words)))) return s;
    B->device_bytes += B->d_lst.bytes() + B->d_twn.bytes();
    TRACE("create: lists sized on the device");
  }
  B->res_host.resize(nh);
  return TBC_OK;
}

tbc_status tbc_batch_create(const tbc_batch_desc* desc, const tbc_model* model,
                            const tbc_opts* opts, tbc_batch** out) {
  if (!desc || !model || !opts || !out || !desc->op_off || !desc->n_events || !desc->n_process ||
      desc->n_hist == 0) {
    set_error("tbc_batch_create: null or empty argument");
    return TBC_ERR_INVALID_ARG;
  }
  const tbc_ops& c = desc->cols;
  if (c.n && (!c.f || !c.a || !c.b || !c.process || !c.inv_pos || !c.ret_pos)) {
    set_error("tbc_batch_create: null op column");
    return TBC_ERR_INVALID_ARG;
  }
  tbc_batch* B = new (std::nothrow) tbc_batch();
  if (!B) return TBC_ERR_OOM;
  tbc_status s;
  const void* const guard_prev = t_guard_owner;
  const size_t guard_prev_nth = t_guard_nth;
  t_guard_owner = B; t_guard_nth = 0;
  try {
    s = batch_create_impl(desc, model, opts, B);
  } catch (const std::bad_alloc&) {
    set_error("host allocation failed");
    s = TBC_ERR_OOM;
  } catch (...) {
    set_error("unexpected exception");
    s = TBC_ERR_HIP;
  }
  t_guard_owner = guard_prev; t_guard_nth = guard_prev_nth;
  if (s != TBC_OK) { delete B; return s; }
  *out = B;
  return TBC_OK;
}

static SearchArgs make_search_args(tbc_batch* B, uint64_t* tab, uint32_t n_work) {
  SearchArgs a{};
  a.hist = B->d_hist.p; a.rec = B->d_rec.p; a.seg = B->d_seg.p; a.ret_slot = B->d_ret_slot.p;
  a.ret_op = B->d_ret_op.p;
  a.frames = B->d_frames.p; a.tab = tab; a.results = B->d_results.p;
  a.witness = B->opts.want_witness ? B->d_witness.p : nullptr;
  a.work = B->d_work.p; a.queue = B->d_queue.p;
  a.table = B->d_table.p; a.n_work = n_work; a.model_kind = B->model.kind;
  a.init_state = B->model.init;
  a.n_classes = B->model.n_classes; a.n_states = B->model.n_states;
  a.max_steps = B->opts.max_steps;
  a.time_limit_ticks = B->opts.time_limit_ms * 100000ull;   // wall_clock64 runs at 100 MHz
  a.dbg = debug_words();
  a.pool_vals = B->d_pool_vals.p;
  a.cfg = B->d_cfg.p;
  return a;
}

static PackArgs make_pack_args(tbc_batch* B) {
  PackArgs pa{};
  pa.hist = B->d_hist.p; pa.f = B->d_f.p; pa.a = B->d_a.p; pa.b = B->d_b.p; pa.process = B->d_proc.p;
  pa.inv_pos = B->d_inv.p; pa.ret_pos = B->d_ret.p; pa.rec = B->d_rec.p; pa.seg = B->d_seg.p;
  pa.ret_slot = B->d_ret_slot.p; pa.ret_op = B->d_ret_op.p; pa.bitmap = B->d_bitmap.p; pa.wpre = B->d_wpre.p;
  pa.scratch = B->d_frames.p; pa.frame_words = B->frame_words; pa.n_hist = B->n_hist;
  pa.model_kind = B->model.kind; pa.n_classes = B->model.n_classes; pa.dbg = debug_words();
  pa.pool_vals = B->d_pool_vals.p; pa.pool_len = B->pool_len; pa.n_keys = B->model.n_keys;
  return pa;
}

static PackOpenArgs make_pack_open_args(tbc_batch* B) {
  PackOpenArgs po{};
  po.hist = B->d_hist.p; po.bh = B->d_bh.p; po.f = B->d_f.p; po.a = B->d_a.p; po.b = B->d_b.p; po.process = B->d_proc.p;
  po.scratch = B->d_frames.p; po.off = B->d_off.p; po.ncr = B->d_ncr.p; po.lst = B->d_lst.p;
  po.rec = B->d_rec.p; po.seg = B->d_seg.p; po.chunks_per_hist = (uint32_t)((B->max_ops + 63) / 64);
  po.crashed = B->d_crashed.p; po.ret_slot = B->d_ret_slot.p; po.slot8 = B->d_slot8.p;
  po.ret_op = B->d_ret_op.p; po.look = B->lookahead ? B->d_look.p : nullptr; po.tmp = B->d_looktmp.p; po.n_hist = B->n_hist; po.mask_words = B->mask_words;
  po.branch_lists = (B->rules & kRuleBranch) ? 1u : 0u;
  po.rk8 = B->lanes ? B->d_rk8.p : nullptr; po.front_words = B->front_words(); po.front_compact = B->front_words() == kFrontCompactWords ? 1u : 0u;
  po.twn = B->reg_rules() ? B->d_twn.p : nullptr; po.rdm = (B->reg_rules() || B->lanes) ? B->d_rdm.p : nullptr; po.vpad = B->vpad;
  po.cmem = B->count_form ? B->d_cmem.p : nullptr;
  po.list_order = B->list_order();
  return po;
}

static uint32_t search_blocks(uint32_t n_work) {
  return std::max(1u, (n_work + kWavesPerBlock - 1) / kWavesPerBlock);
}

static BeamArgs make_beam_args(tbc_batch* B, uint64_t* tab, uint32_t* stack, uint32_t* dstack, uint32_t n_work) {
  BeamArgs a{};
  a.hist = B->d_hist.p; a.bh = B->d_bh.p; a.off = B->d_off.p; a.ncr = B->d_ncr.p; a.lst = B->d_lst.p;
  a.crashed = B->d_crashed.p; a.slot8 = B->d_slot8.p; a.look = B->lookahead ? B->d_look.p : nullptr; a.ret_slot = B->d_ret_slot.p; a.ret_op = B->d_ret_op.p;
  a.stack = stack; a.dstack = B->lookahead ? dstack : nullptr; a.tab = tab; a.results = B->d_results.p;
  a.witness = B->opts.want_witness ? B->d_witness.p : nullptr;
  a.work = B->d_work.p; a.table = B->d_table.p; a.n_work = n_work; a.model_kind = B->model.kind;
  const bool comm = B->model.kind == TBC_MODEL_SET || B->model.kind == TBC_MODEL_BANK;
  a.init_state = comm ? 0 : B->model.init;
  a.model_aux = B->model.init; a.n_keys = B->model.n_keys;
  a.n_classes = B->model.n_classes; a.width = B->width;
  a.round_budget = B->opts.round_budget;
  a.cmem = B->d_cmem.p; a.count_mode = kCountExact; a.tab_stride = B->entry_words(); a.epoch = 0;
  a.rules = B->rules; a.twn = B->d_twn.p; a.rdm = B->d_rdm.p; a.vpad = B->vpad; a.rk8 = B->d_rk8.p; a.front_words = B->front_words(); a.next_work = B->d_queue.p;
  a.max_steps = B->opts.max_steps;
  a.time_limit_ticks = B->opts.time_limit_ms * 100000ull;
  a.dbg = debug_words();
  a.pool_vals = B->d_pool_vals.p;
  a.cfg = B->d_cfg.p;
  a.pool = B->d_pool.p; a.pool_cursor = B->d_pool_cursor.p; a.pool_words = B->d_pool.n;
  {
    const uint64_t max_bytes = B->opts.max_visited_bytes ? B->opts.max_visited_bytes : (1ull << 30);
    uint32_t lg = 10;
    while (lg < kBeamMaxTabLog2 && (1ull << (lg + 1)) * B->entry_words() * 8 <= max_bytes) lg++;
    a.max_tab_log2 = lg;
  }
  return a;
}

// One extra pass over the histories in `grp` with per-history visited sets of 2^lg[i] entries in a
// scratch arena (overflow retries, and wide-schedule histories that fall back to the sequential kernel).
// count form: `count_mode` (exact / relaxed), per-history prefix targets and a step limit of the pass's own (steps_override >= 0).
static tbc_status scratch_pass(tbc_batch* B, const std::vector<uint32_t>& grp, const std::vector<uint32_t>& lg,
                               bool beam, const HostBuf<Hist>& hist_back, const HostBuf<BeamHist>& bh_back,
                               uint32_t width_override = 0, uint32_t count_mode = kCountExact, const std::vector<uint32_t>* targets = nullptr,
                               int64_t steps_override = -1) {
  hipStream_t s = B->stream;
  const uint32_t KW = 1 + B->mask_words, EW = B->entry_words();
  const uint64_t words_per_entry = beam ? EW : KW;
  uint64_t entries = 0;
  std::vector<Hist> ph(grp.size());
  std::vector<BeamHist> pb(beam ? grp.size() : 0);
  for (size_t i = 0; i < grp.size(); i++) {
    ph[i] = hist_back[grp[i]];
    if (beam) {
      pb[i] = bh_back[grp[i]];
      pb[i].tab_off = entries; pb[i].stack_off = entries; pb[i].tab_log2 = lg[i];
      pb[i].target = targets ? (*targets)[i] : 0u;
    } else {
      ph[i].tab_off = entries * KW; ph[i].tab_log2 = lg[i];
    }
    entries += 1ull << lg[i];
  }
  DevBuf<uint64_t> big;
  DevBuf<uint32_t> bstack, bdstack, bframes;
  tbc_status st = big.alloc(entries * words_per_entry);
  if (st != TBC_OK) return st;
  const uint32_t seq_fw = search_frame_words(B->mask_words);
  if (!beam && B->frame_words < seq_fw) {          // the batch's frames arena is sized for the pack kernels only: the sequential kernel's stack is taken here
    uint64_t fn = 0;
    for (size_t i = 0; i < grp.size(); i++) { ph[i].frame_off = fn; fn += std::max<uint64_t>(ph[i].n_ops, 1) * seq_fw; }
    if ((st = bframes.alloc(fn)) != TBC_OK) { big.release(); return st; }
  }
  if (beam && (st = bstack.alloc(entries)) != TBC_OK) { big.release(); return st; }
  if (beam && B->lookahead && (st = bdstack.alloc(entries)) != TBC_OK) { big.release(); bstack.release(); return st; }
  hipError_t e = hipMemsetAsync(big.p, 0, entries * words_per_entry * 8, s);
  for (size_t i = 0; i < grp.size() && e == hipSuccess; i++) {
    e = hipMemcpyAsync(B->d_hist.p + grp[i], &ph[i], sizeof(Hist), hipMemcpyHostToDevice, s);
    if (beam && e == hipSuccess) e = hipMemcpyAsync(B->d_bh.p + grp[i], &pb[i], sizeof(BeamHist), hipMemcpyHostToDevice, s);
  }
  if (e == hipSuccess) e = hipMemcpyAsync(B->d_work.p, grp.data(), grp.size() * 4, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) {
    const uint32_t nw = (uint32_t)grp.size();
    if (beam) { BeamArgs ba = make_beam_args(B, big.p, bstack.p, bdstack.p, nw); ba.pool = nullptr; ba.pool_words = 0;
      if (width_override) ba.width = width_override;
      ba.count_mode = count_mode;
      if (steps_override >= 0) ba.max_steps = (uint64_t)steps_override;
      // (a retry runs the schedule of the first pass: several histories per wavefront stay so)
      if (B->lanes && (!width_override || width_override == B->width)) launch_narrow(ba, B->mask_words, B->lanes, s); else launch_beam(ba, B->mask_words, search_blocks(nw), s); }
    else { SearchArgs ra = make_search_args(B, big.p, nw); if (bframes.p) ra.frames = bframes.p; launch_search(ra, B->mask_words, search_blocks(nw), s); }
    e = hipGetLastError();
  }
  for (size_t i = 0; i < grp.size() && e == hipSuccess; i++)
    e = hipMemcpyAsync(&B->res_host[grp[i]], B->d_results.p + grp[i], sizeof(DevResult), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  // put the descriptors back so the next run starts from the resident layout
  for (size_t i = 0; i < grp.size() && e == hipSuccess; i++) {
    e = hipMemcpyAsync(B->d_hist.p + grp[i], &hist_back[grp[i]], sizeof(Hist), hipMemcpyHostToDevice, s);
    if (beam && e == hipSuccess) e = hipMemcpyAsync(B->d_bh.p + grp[i], &bh_back[grp[i]], sizeof(BeamHist), hipMemcpyHostToDevice, s);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  big.release(); bstack.release(); bdstack.release(); bframes.release();
  if (e != hipSuccess) { set_error("scratch pass failed: %s", hipGetErrorString(e)); return TBC_ERR_HIP; }
  return TBC_OK;
}

// :configs of an invalid verdict: the (state, linearized pending calls) pairs stuck at the failing
// completion, sorted, first TBC_MAX_FINAL_CONFIGS.  The pending calls are recomputed from the op
// columns of that one history (copied back on demand -- invalid verdicts are rare).
static tbc_status fill_configs(tbc_batch* B, uint32_t h, const DevResult& d, tbc_result* r) {
  const uint32_t MW = B->mask_words, RW = 2 + MW;
  const uint32_t got = std::min<uint32_t>(d.n_configs, kCfgCap);
  if (got == 0 || d.fail_op == TBC_NO_OP) return TBC_OK;
  std::vector<uint64_t> rec((size_t)got * RW);
  HIP_TRY(hipMemcpy(rec.data(), B->d_cfg.p + (uint64_t)h * kCfgCap * RW, rec.size() * 8, hipMemcpyDeviceToHost));
  const Hist& H = B->hist[h];
  const uint32_t n = H.n_ops;
  std::vector<int32_t> proc(n);
  std::vector<uint32_t> inv(n), ret(n);
  HIP_TRY(hipMemcpy(proc.data(), B->d_proc.p + H.op_off, (size_t)n * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(inv.data(), B->d_inv.p + H.op_off, (size_t)n * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(ret.data(), B->d_ret.p + H.op_off, (size_t)n * 4, hipMemcpyDeviceToHost));
  const uint32_t P = ret[d.fail_op];                 // history position of the failing completion
  std::vector<uint32_t> pending;                     // calls open at that position, invocation order
  for (uint32_t i = 0; i < n && inv[i] < P; i++)
    if (ret[i] == TBC_POS_CRASHED || ret[i] >= P) pending.push_back(i);
  std::vector<uint32_t> order(got);
  for (uint32_t i = 0; i < got; i++) order[i] = i;
  std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
    const uint64_t* a = &rec[(size_t)x * RW]; const uint64_t* b = &rec[(size_t)y * RW];
    const int32_t sa = (int32_t)(a[0] >> 32), sb = (int32_t)(b[0] >> 32);
    if (sa != sb) return sa < sb;
    for (uint32_t j = 0; j < MW; j++) if (a[1 + j] != b[1 + j]) return a[1 + j] < b[1 + j];
    return false;
  });
  (void)0;
  r->n_configs = std::min<uint32_t>(got, TBC_MAX_FINAL_CONFIGS);
  for (uint32_t c = 0; c < r->n_configs; c++) {
    const uint64_t* e = &rec[(size_t)order[c] * RW];
    tbc_config& o = r->configs[c];
    o.state = B->count_form ? (int32_t)((uint32_t)(e[0] >> 32) & ~kHotBit) : (int32_t)(e[0] >> 32);
    o.last_op = (uint32_t)e[1 + MW];
    o.n_pending = (uint32_t)pending.size();
    o.n_linearized = 0; o.linearized_mask = 0;
    for (size_t k = 0; k < pending.size(); k++) {
      const uint32_t p = (uint32_t)proc[pending[k]];
      // (count form: a crashed call holds no slot; which of them a config has linearized is in its count vector, not reported here)
      const bool lin = !(B->count_form && ret[pending[k]] == TBC_POS_CRASHED) && ((e[1 + (p >> 6)] >> (p & 63u)) & 1ull);
      if (k < 16) { o.pending[k] = pending[k]; if (lin) o.linearized_mask |= 1u << k; }
      o.n_linearized += lin;
    }
  }
  return TBC_OK;
}

// Eager reads: the wide search branches over :write / :cas only and its parent chain holds just those calls.
// The full linearization is the chain replayed from the initial state with the rule applied as the search
// applies it: after every chain call the front moves past the completions now linearized, then every open live
// read (process-slot order) whose value is nil or the state is linearized, again after each move of the front.
// wit[0..len) = the chain in, the whole witness out (at most n_ops entries: the caller's slice has that room).
static tbc_status expand_eager_witness(tbc_batch* B, uint32_t h, uint32_t* wit, uint32_t* len) {
  const Hist& H = B->hist[h];
  const uint32_t n = H.n_ops;
  std::vector<uint8_t> f(n);
  std::vector<int32_t> a(n), b(n), proc(n);
  std::vector<uint32_t> inv(n), ret(n);
  HIP_TRY(hipMemcpy(f.data(), B->d_f.p + H.op_off, (size_t)n, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(a.data(), B->d_a.p + H.op_off, (size_t)n * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(b.data(), B->d_b.p + H.op_off, (size_t)n * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(proc.data(), B->d_proc.p + H.op_off, (size_t)n * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(inv.data(), B->d_inv.p + H.op_off, (size_t)n * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(ret.data(), B->d_ret.p + H.op_off, (size_t)n * 4, hipMemcpyDeviceToHost));
  // (the replay itself is plain host code: witness_expand.h -- tests/test_narrow_emu.py runs the same function on the emulator's chains)
  std::vector<uint32_t> out;
  if (!expand_eager_chain(n, f.data(), a.data(), b.data(), proc.data(), inv.data(), ret.data(), H.n_slots, B->model.init,
                          (B->rules & kRuleBranch) != 0u, B->list_order() != 0u, wit, *len, out)) {
    set_error("history %u: malformed witness chain", h);
    return TBC_ERR_HIP;
  }
  std::copy(out.begin(), out.end(), wit);
  *len = (uint32_t)out.size();
  return TBC_OK;
}

// phase 0: the whole run.  phase 1 (tbc_batch_sweep_partial): pack + this rank's share of the sweep, stop before the
// verdicts.  phase 2 (tbc_batch_sweep_finish): verdicts from the merged relation table already placed in seg_host.
// ---- several batches in flight on one device (each on its own stream, from its own host thread).  The narrow kernel is sized
// to the whole GPU and lives on latency, the pack kernels on vector issue: a batch's pack beside ANOTHER batch's search uses
// what the search leaves idle, two searches at once only halve each other.  So the searches of one device are chained through
// an event -- a search starts when the one launched before it, on whatever stream, is done -- and everything else floats.
namespace {
struct SearchTurn {
  static std::mutex& mu() { static std::mutex m; return m; }
  static hipEvent_t& last(int dev) { static hipEvent_t ev[64] = {}; return ev[dev & 63]; }
  std::lock_guard<std::mutex> g;
  int dev; hipStream_t s;
  SearchTurn(int device, hipStream_t stream) : g(mu()), dev(device), s(stream) {
    if (last(dev)) (void)hipStreamWaitEvent(s, last(dev), 0);
  }
  ~SearchTurn() {
    if (!last(dev)) (void)hipEventCreateWithFlags(&last(dev), hipEventDisableTiming);
    if (last(dev)) (void)hipEventRecord(last(dev), s);
  }
};
// wavefronts per SIMD the narrow kernel is launched at (the kernel is built for up to TBC_NARROW_MIN_WAVES = 4; fewer leave
// registers and wave slots for the pack kernels of another batch in flight).  TBC_NARROW_WAVES_PER_SIMD overrides, 0 = the build's.
uint32_t narrow_waves_per_simd() {
  const char* e = std::getenv("TBC_NARROW_WAVES_PER_SIMD");
  return e ? (uint32_t)std::strtoul(e, nullptr, 10) : 0u;
}
}  // namespace

// Histories of a narrow-kernel batch that stopped because they no longer passed completions (BeamArgs.stall_checks) are checked again as
// a small batch of their own -- knossos.competition without a witness, i.e. the level sweep (what it cannot finish: the wide search) --
// from their op columns as they lie in HBM.  A history stalls when it is NOT linearizable (the search is exhausting the configs in front
// of the completion nobody can pass: nine times a valid history's search for a bad read in the middle of a 10k-op history, and a pass is
// as long as its slowest history) or, rarely, in a burst of concurrency; the sweep decides either in milliseconds.
// How long is "no longer"?  A VALID history stalls too, in a burst of concurrency: of 24 bench histories under the emulator 3 stop at 8 looks at
// the clock (512 rounds), 2 at 16, none at 32; on the device, at 48 looks, ~20 of 32,768 -- and a valid history that is stopped has lost
// its search and costs a sweep.  64 looks (4,096 rounds, ~53 ms: a whole valid search) is past nearly every burst; a bad read in the
// middle of a history then holds its pass for one more search's time instead of nine.
static const uint32_t kStallChecks = 64;
static tbc_status hand_over_stalled(tbc_batch* B, const std::vector<uint32_t>& list, std::vector<tbc_result>& out) {
  Ctx* const saved = t_ctx;
  t_ctx = nullptr;                               // (the inner batch owns its arenas, stream and events)
  tbc_status st = TBC_OK;
  const size_t chunk = list.size() <= 32 ? 1 : 256;         // (a few: one at a time through tbc_check's persistent contexts -- no allocation, 1 - 3 ms each)
  for (size_t lo = 0; lo < list.size() && st == TBC_OK; lo += chunk) {
    const size_t hi = std::min(list.size(), lo + chunk);
    const uint32_t k = (uint32_t)(hi - lo);
    std::vector<uint64_t> off(k + 1, 0);
    std::vector<uint32_t> nev(k), npr(k);
    std::vector<int32_t> aux(k);
    for (uint32_t i = 0; i < k; i++) {
      const Hist& H = B->hist[list[lo + i]];
      off[i + 1] = off[i] + H.n_ops; nev[i] = H.n_events; npr[i] = H.n_slots; aux[i] = H.aux;
    }
    const uint64_t T = off[k];
    std::vector<uint8_t> f(T + 1);
    std::vector<int32_t> a(T + 1), b(T + 1), pr(T + 1);
    std::vector<uint32_t> inv(T + 1), ret(T + 1);
    for (uint32_t i = 0; i < k; i++) {
      const Hist& H = B->hist[list[lo + i]];
      const uint64_t n = H.n_ops, o = off[i], s0 = H.op_off;
      if (!n) continue;
      HIP_TRY(hipMemcpy(f.data() + o, B->d_f.p + s0, n, hipMemcpyDeviceToHost));
      HIP_TRY(hipMemcpy(a.data() + o, B->d_a.p + s0, n * 4, hipMemcpyDeviceToHost));
      HIP_TRY(hipMemcpy(b.data() + o, B->d_b.p + s0, n * 4, hipMemcpyDeviceToHost));
      HIP_TRY(hipMemcpy(pr.data() + o, B->d_proc.p + s0, n * 4, hipMemcpyDeviceToHost));
      HIP_TRY(hipMemcpy(inv.data() + o, B->d_inv.p + s0, n * 4, hipMemcpyDeviceToHost));
      HIP_TRY(hipMemcpy(ret.data() + o, B->d_ret.p + s0, n * 4, hipMemcpyDeviceToHost));
    }
    tbc_batch_desc d{};
    d.n_hist = k; d.op_off = off.data(); d.n_events = nev.data(); d.n_process = npr.data(); d.model_aux = aux.data();
    d.cols.n = (uint32_t)T; d.cols.f = f.data(); d.cols.a = a.data(); d.cols.b = b.data(); d.cols.process = pr.data();
    d.cols.inv_pos = inv.data(); d.cols.ret_pos = ret.data();
    tbc_opts o = B->opts;
    o.algorithm = TBC_ALG_COMPETITION; o.want_witness = 0; o.search_width = 0; o.lanes_per_history = 0; o.round_budget = 0; o.max_steps = 0;
    o.visited_per_op = 0; o.list_order = TBC_ORDER_DEFAULT;
    if (chunk == 1) {
      tbc_ops one = d.cols;
      one.n = (uint32_t)T; one.n_events = nev[0]; one.n_process = npr[0];
      tbc_model m1 = B->model;
      m1.init = aux[0];
      tbc_result r1;
      st = tbc_check(&one, &m1, &o, &r1);
      if (st == TBC_OK) { out[list[lo]] = r1; out[list[lo]].witness = nullptr; tbc_result_free(&r1); }
      continue;
    }
    tbc_batch* I = nullptr;
    st = tbc_batch_create(&d, &B->model, &o, &I);
    if (st == TBC_OK) {
      std::vector<tbc_result> res(k);
      st = tbc_batch_run(I, res.data());
      for (uint32_t i = 0; i < k && st == TBC_OK; i++) { out[list[lo + i]] = res[i]; out[list[lo + i]].witness = nullptr; }
      tbc_batch_destroy(I);
    }
  }
  t_ctx = saved;
  return st;
}

static tbc_status batch_run_impl(tbc_batch* B, tbc_result* results, int phase = 0) {
  HIP_TRY(hipSetDevice(B->device));
  const uint64_t t_start = now_ns();
  const uint32_t nh = B->n_hist;
  hipStream_t s = B->stream;
  const bool beam = B->width > 1;
  const uint32_t KW = 1 + B->mask_words, EW = B->entry_words();
  HostBuf<Hist>& hist_back = B->hist_back_m;
  HostBuf<BeamHist>& bh_back = B->bh_back_m;
  // count form: the exact search runs under a budget of probes; what it does not finish goes through the relaxed refutation and
  // the prefix search below (a caller who names max_steps gets one exact pass under that limit instead)
  const uint64_t count_budget = (B->count_form && B->opts.max_steps == 0) ? 32ull * B->max_ops : 0ull;
  SweepArgs swa{};
  if (B->sweep || B->rsweep) {
    swa.hist = B->d_hist.p; swa.bh = B->d_bh.p; swa.off = B->d_off.p; swa.ncr = B->d_ncr.p; swa.lst = B->d_lst.p;
    swa.crashed = B->d_crashed.p; swa.twn = B->reg_rules() ? B->d_twn.p : nullptr; swa.rdm = B->reg_rules() ? B->d_rdm.p : nullptr;
    swa.slot8 = B->d_slot8.p; swa.cuts = B->d_cuts.p; swa.seg = B->d_sres.p; swa.table = B->d_table.p;
    swa.pool_vals = B->d_pool_vals.p; swa.n_hist = nh; swa.max_segs = B->max_segs; swa.seg_target = B->seg_target;
    swa.cut_open = B->cut_open; swa.n_dom = B->n_dom; swa.vpad = B->vpad ? B->vpad : 1; swa.rules = B->rules;
    swa.model_kind = B->model.kind; swa.init_state = B->model.init;
    swa.n_classes = B->model.n_classes; swa.n_keys = B->model.n_keys;
    swa.shard_rank = 0; swa.shard_world = 1;
  }
  // a big quiet batch (several histories per wavefront), IF ASKED (tbc_opts.dominance, TBC_DOM_STALL_HANDOVER: off by default, tbcheck.h says why):
  // a history that stops passing completions is handed to the level sweep (hand_over_stalled) -- where that can answer: register / cas-register, one mask word, nobody asking for a witness or naming a step limit
  const bool stall_on = B->lanes != 0 && (B->opts.dominance & TBC_DOM_STALL_HANDOVER) != 0 && !B->count_form && B->mask_words == 1 && !B->opts.want_witness && B->opts.max_steps == 0 && phase == 0 &&
                        (B->model.kind == TBC_MODEL_REGISTER || B->model.kind == TBC_MODEL_CAS_REGISTER) && B->vpad != 0;
  std::vector<tbc_result> handed;
  std::vector<uint8_t> was_handed(nh, 0);
  // the relaxed sweep's verdicts: the completion rank at which history h is refuted (kInf: not refuted -- or not swept at all)
  std::vector<uint32_t> rs_level(nh, kInf);
  std::vector<uint8_t> rs_valid(nh, 0);          // ... and the histories it could not refute (VALID under the relaxation: the exact search just needs its time)
  if (phase != 2) {

  TRACE("run: begin");
  HIP_TRY(hipEventRecord(B->ev[0], s));
  // (tbc_check: the arenas a run zeroes -- position bitmap, list offsets, crashed-call counts, the pool cursor -- are consecutive
  // pieces of the context's slab: one memset; the descriptors came up with the columns: not again.  Six memsets and two copies were 34 us)
  const bool zero_block = B->borrowed && !guard_on() && !B->d_bitmap.owned && !B->d_off.owned && !B->d_ncr.owned && !B->d_pool_cursor.owned &&
                          (char*)B->d_bitmap.p < (char*)B->d_pool_cursor.p && (size_t)((char*)B->d_pool_cursor.p - (char*)B->d_bitmap.p) < (64u << 20) &&
                          (char*)B->d_off.p > (char*)B->d_bitmap.p && (char*)B->d_off.p < (char*)B->d_pool_cursor.p &&
                          (char*)B->d_ncr.p > (char*)B->d_bitmap.p && (char*)B->d_ncr.p < (char*)B->d_pool_cursor.p;
  const bool fresh = B->inputs_fresh;
  B->inputs_fresh = false;
  if (zero_block) HIP_TRY(hipMemsetAsync(B->d_bitmap.p, 0, (size_t)((char*)B->d_pool_cursor.p - (char*)B->d_bitmap.p) + sizeof(unsigned long long), s));
  else HIP_TRY(hipMemsetAsync(B->d_bitmap.p, 0, B->d_bitmap.bytes(), s));
  // several histories per wavefront: the visited sets are not zeroed before every pass -- the keys carry the pass number and
  // another pass's entries read as empty (wgl_narrow_impl.h, entry_empty); the arena is zeroed when the number wraps (and first of all)
  const bool use_epoch = beam && B->lanes != 0;          // (a batch with lanes has every history below kNarrowMaxOps: batch_create_impl)
  if (beam) {
    if (use_epoch) B->epoch = B->epoch % 255u + 1u;
    // (a sweep batch has no visited sets and no growth pool of its own -- one-element stand-ins nobody reads: what the sweep hands
    // to the depth-first search runs in scratch arenas, scratch_pass)
    if (!B->sweep && (!use_epoch || B->epoch == 1u)) HIP_TRY(hipMemsetAsync(B->d_btab.p, 0, B->d_btab.bytes(), s));
    if (!B->sweep) HIP_TRY(hipMemsetAsync(B->d_pool.p, 0, B->d_pool.bytes(), s));
    if (!zero_block) {
      HIP_TRY(hipMemsetAsync(B->d_pool_cursor.p, 0, sizeof(unsigned long long), s));
      HIP_TRY(hipMemsetAsync(B->d_off.p, 0, B->d_off.bytes(), s));
      HIP_TRY(hipMemsetAsync(B->d_ncr.p, 0, B->d_ncr.bytes(), s));
    }
    if (!fresh) HIP_TRY(hipMemcpyAsync(B->d_bh.p, B->bh.data(), nh * sizeof(BeamHist), hipMemcpyHostToDevice, s));
  } else {
    HIP_TRY(hipMemsetAsync(B->d_tab.p, 0, B->d_tab.bytes(), s));
  }
  if (!fresh) HIP_TRY(hipMemcpyAsync(B->d_hist.p, B->hist.data(), nh * sizeof(Hist), hipMemcpyHostToDevice, s));
  HIP_TRY(hipEventRecord(B->ev[1], s));
  TRACE("run: memsets queued");
  SYNC_TRACE("memsets");

  // a handful of histories are packed by a workgroup's sixteen wavefronts each (pack_one.hip) -- the single-history call's 0.36 ms pack
  // was one wavefront's chain in pack_kernel -- with open_counts_kernel's tables in the same pass where they fit (pack_one_counts_kernel,
  // one launch fewer per call): 0.36 -> 0.10 ms, measured round 5 (profiles/r05_single_history_forms_first_device_run.json)
  bool packed = false, counted = false;
  if (nh <= 8) {
    bool fits = true, fits2 = beam;
    for (uint32_t h = 0; h < nh; h++) {
      fits = fits && pack_one_fits(B->model.kind, B->hist[h].n_ops, B->hist[h].n_events, B->hist[h].n_slots);
      fits2 = fits2 && pack_one_counts_fits(B->model.kind, B->hist[h].n_ops, B->hist[h].n_events, B->hist[h].n_slots);
    }
    if (fits2) packed = counted = launch_pack_one_counts(make_pack_args(B), make_pack_open_args(B), s);
    if (!packed && fits) packed = launch_pack_one(make_pack_args(B), s);
  }
  // a batch of the wide schedule whose histories all fit is packed by four wavefronts per history with the tables in LDS (pack_one.hip,
  // pack_wg_kernel), and the same pass leaves what open_counts_kernel would -- the ranks never leave the registers between the two
  // (round 5, first device run: 14.4 against 16.1 ms per 8,192 bench histories; every batch parity test green under it); the others keep
  // pack_kernel + open_counts_kernel
  if (!packed && beam) {
    bool fits = true, slots64 = true;          // (at most 64 slots everywhere: 19 KB of LDS a history instead of 31, eight workgroups per CU)
    for (uint32_t h = 0; h < nh && fits; h++) {
      fits = pack_wg_fits(B->model.kind, B->hist[h].n_ops, B->hist[h].n_events, B->hist[h].n_slots);
      slots64 = slots64 && pack_wg64_fits(B->model.kind, B->hist[h].n_ops, B->hist[h].n_events, B->hist[h].n_slots);
    }
    if (fits) packed = counted = launch_pack_wg(make_pack_args(B), make_pack_open_args(B), s, slots64);
  }
  if (!packed) launch_pack(make_pack_args(B), s);
  HIP_TRY(hipGetLastError());
  if (beam) {
    PackOpenArgs po = make_pack_open_args(B);
    launch_pack_open(po, s, counted);
    HIP_TRY(hipGetLastError());
  }
  TRACE("run: pack launched");
  SYNC_TRACE("pack");
  HIP_TRY(hipEventRecord(B->ev[2], s));

  // ---- the RELAXED sweep (see tbc_batch::rsweep) on a stream of its own, BESIDE the exact search: INVALID at completion t = invalid, first bad
  // completion at t or earlier -- the history's exact search is told to stop (BeamArgs.abort) and the prefix search below pins the completion;
  // VALID (or a burst that outgrows the sets) proves nothing and the exact search runs on as it always did, the sweep's 17 ms hidden behind it
  const bool rs_on = B->rsweep && phase == 0 && beam && !B->lanes;
  SweepArgs ra = swa;
  if (rs_on) {
    ra.ncr = B->d_zncr.p; ra.crashed = nullptr; ra.reach = B->d_reach.p; ra.reach_hdr = B->d_reach_hdr.p;
    ra.rules = B->rules & (kRuleEager | kRuleTwin);
    HIP_TRY(hipMemsetAsync(B->d_abort.p, 0, B->d_abort.bytes(), s));
    HIP_TRY(hipEventRecord(B->ev2[0], s));                      // the pack's tables are complete, the abort words zero
    HIP_TRY(hipStreamWaitEvent(B->stream2, B->ev2[0], 0));
    hist_back.resize(nh);
    if (launch_sweep(ra, B->stream2)) {
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipMemcpyAsync(B->seg_host.data(), B->d_sres.p, B->seg_host.size() * sizeof(SegResult), hipMemcpyDeviceToHost, B->stream2));
      HIP_TRY(hipMemcpyAsync(hist_back.data(), B->d_hist.p, nh * sizeof(Hist), hipMemcpyDeviceToHost, B->stream2));
    }
  }

  if (B->sweep) {
    SweepArgs mine = swa;
    if (phase == 1 && B->shard_world > 1) {     // another rank's records must read as zero in the exchanged table
      mine.shard_rank = B->shard_rank; mine.shard_world = B->shard_world;
      HIP_TRY(hipMemsetAsync(B->d_sres.p, 0, B->d_sres.bytes(), s));
    }
    if (!launch_sweep(mine, s)) { set_error("level sweep launch failed"); return TBC_ERR_HIP; }
  } else if (beam) {
    BeamArgs ba = make_beam_args(B, B->d_btab.p, B->d_stack.p, B->d_dstack.p, nh);
    if (count_budget) ba.max_steps = count_budget;
    if (rs_on) ba.abort = B->d_abort.p;
    if (B->lanes) { ba.tab_stride = B->tab_stride(); ba.epoch = use_epoch ? B->epoch : 0u; }
    if (stall_on) ba.stall_checks = kStallChecks;
    if (B->lanes) {
      SearchTurn turn(B->device, s);        // one whole-GPU search at a time; another batch's pack runs beside it
      if (!B->ev_turn) HIP_TRY(hipEventCreate(&B->ev_turn));
      HIP_TRY(hipEventRecord(B->ev_turn, s));
      if (!launch_narrow(ba, B->mask_words, B->lanes, s, narrow_waves_per_simd())) { set_error("unsupported mask width"); return TBC_ERR_UNSUPPORTED; }
    } else if (!launch_beam(ba, B->mask_words, search_blocks(nh), s)) { set_error("unsupported mask width"); return TBC_ERR_UNSUPPORTED; }
  } else {
    SearchArgs sa = make_search_args(B, B->d_tab.p, nh);
    if (!launch_search(sa, B->mask_words, search_blocks(nh), s)) { set_error("unsupported mask width"); return TBC_ERR_UNSUPPORTED; }
  }
  HIP_TRY(hipGetLastError());
  TRACE("run: search launched");
  SYNC_TRACE("search");
  if (rs_on) {
    // (the exact search is running; the sweep's relations arrive on the other stream)
    HIP_TRY(hipStreamSynchronize(B->stream2));
    const uint32_t SL = kSweepSlices;
    std::vector<uint32_t> again;
    for (uint32_t h = 0; h < nh; h++) if (hist_back[h].status == 0)
      for (uint32_t k = 0; k < B->max_segs; k++) for (uint32_t j = 0; j < SL; j++)
        if (B->seg_host[((size_t)h * B->max_segs + k) * SL + j].status == kSegOverflow) { again.push_back(h); again.push_back(k); again.push_back(j); }
    if (!again.empty()) {          // the bursts that outgrew the first pass's sets: once more with the big ones
      HIP_TRY(hipMemcpyAsync(B->d_seglist.p, again.data(), again.size() * 4, hipMemcpyHostToDevice, B->stream2));
      SweepArgs r2 = ra;
      r2.seg_list = B->d_seglist.p; r2.n_list = (uint32_t)(again.size() / 3);
      if (launch_sweep(r2, B->stream2)) {
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(B->seg_host.data(), B->d_sres.p, B->seg_host.size() * sizeof(SegResult), hipMemcpyDeviceToHost, B->stream2));
      }
      HIP_TRY(hipStreamSynchronize(B->stream2));
    }
    static const uint32_t kOne = 1u;
    for (uint32_t h = 0; h < nh; h++) {
      if (hist_back[h].status != 0 || hist_back[h].n_ret == 0) continue;
      tbc_sweep_verdict v{};
      (void)tbc_sweep_compose(&B->seg_host[(size_t)h * B->max_segs * SL], B->max_segs, hist_back[h].n_ret, &v);
      if (v.valid == TBC_INVALID) {
        rs_level[h] = v.fail_level;
        HIP_TRY(hipMemcpyAsync(B->d_abort.p + h, &kOne, 4, hipMemcpyHostToDevice, B->stream2));      // its exact search may stop
      }
      rs_valid[h] = v.valid == TBC_VALID;
    }
    TRACE("run: relaxed sweep composed");
  }
  HIP_TRY(hipEventRecord(B->ev[3], s));
  if (!B->sweep) HIP_TRY(hipMemcpyAsync(B->res_host.data(), B->d_results.p, nh * sizeof(DevResult), hipMemcpyDeviceToHost, s));
  else HIP_TRY(hipMemcpyAsync(B->seg_host.data(), B->d_sres.p, B->seg_host.size() * sizeof(SegResult), hipMemcpyDeviceToHost, s));
  hist_back.resize(nh);
  bh_back.resize(beam ? nh : 0);
  HIP_TRY(hipMemcpyAsync(hist_back.data(), B->d_hist.p, nh * sizeof(Hist), hipMemcpyDeviceToHost, s));
  if (beam) HIP_TRY(hipMemcpyAsync(bh_back.data(), B->d_bh.p, nh * sizeof(BeamHist), hipMemcpyDeviceToHost, s));
  if (B->borrowed && B->sweep) {
    // tbc_check through the level sweep: the whole device side of the call is ~1 ms -- waiting for it by asking the stream (a few
    // thousand queries) instead of sleeping until the driver wakes the thread saves the wake-up; anything longer sleeps as before
    const uint64_t t_spin = now_ns();
    hipError_t q;
    while ((q = hipStreamQuery(s)) == hipErrorNotReady && now_ns() - t_spin < 3000000ull) {}
    if (q != hipSuccess && q != hipErrorNotReady) HIP_TRY(q);
  }
  HIP_TRY(hipStreamSynchronize(s));
  TRACE("run: first pass synced");
  for (uint32_t h = 0; h < nh; h++) if (rs_level[h] != kInf) {      // refuted by the relaxed sweep: whatever its exact search got to before it was told to stop is dropped (the passes below, and their counters, are then the same run after run)
    DevResult& d = B->res_host[h];
    std::memset(&d, 0, sizeof d);
    d.valid = TBC_UNKNOWN; d.cause = TBC_CAUSE_STEP_LIMIT; d.fail_op = TBC_NO_OP; d.prev_ok_op = TBC_NO_OP; d.final_state = B->model.init;
    d.tab_log2 = B->bh[h].tab_log2;
  }
  }   // phase != 2
  B->partial_done = phase == 1;
  if (phase == 1) return TBC_OK;
  if (phase == 2) {
    if (!B->sweep || hist_back.size() != nh) { set_error("tbc_batch_sweep_finish without tbc_batch_sweep_partial"); return TBC_ERR_INVALID_ARG; }
    // the device table becomes the merged one, so a second pass over overflowed wavefronts updates it in place
    HIP_TRY(hipMemcpyAsync(B->d_sres.p, B->seg_host.data(), B->seg_host.size() * sizeof(SegResult), hipMemcpyHostToDevice, s));
  }

  const uint64_t max_bytes = B->opts.max_visited_bytes ? B->opts.max_visited_bytes : (1ull << 30);
  std::vector<uint32_t> final_log2(nh);
  std::vector<uint8_t> is_seq(nh, beam ? 0 : 1);       // which kernel owns the history's result
  std::vector<uint8_t> by_sweep(nh, 0);                // answered by the level sweep (analyzer :linear)
  for (uint32_t h = 0; h < nh; h++) final_log2[h] = beam ? B->bh[h].tab_log2 : B->hist[h].tab_log2;
  HIP_TRY(hipEventRecord(B->ev[4], s));
  bool touched_work = false;
  if (B->sweep) {
    // wavefronts whose config sets outgrew the small LDS sets: once more with the big ones (one wavefront per CU)
    const uint32_t SL = kSweepSlices;
    {
      std::vector<uint32_t> again;
      for (uint32_t h = 0; h < nh; h++) if (hist_back[h].status == 0 && bh_back[h].status == 0)
        for (uint32_t k = 0; k < B->max_segs; k++) for (uint32_t j = 0; j < SL; j++)
          if (B->seg_host[((size_t)h * B->max_segs + k) * SL + j].status == kSegOverflow) { again.push_back(h); again.push_back(k); again.push_back(j); }
      if (!again.empty()) {
        HIP_TRY(hipMemcpyAsync(B->d_seglist.p, again.data(), again.size() * 4, hipMemcpyHostToDevice, s));
        SweepArgs sa2 = swa;
        sa2.seg_list = B->d_seglist.p; sa2.n_list = (uint32_t)(again.size() / 3);
        if (launch_sweep(sa2, s)) {
          HIP_TRY(hipGetLastError());
          HIP_TRY(hipMemcpyAsync(B->seg_host.data(), B->d_sres.p, B->seg_host.size() * sizeof(SegResult), hipMemcpyDeviceToHost, s));
        }
        HIP_TRY(hipStreamSynchronize(s));
      }
    }
    // compose the relations in order (tbc_sweep_compose, tbc_host.cpp); what the sweep could not finish goes to the wide kernel
    std::vector<uint32_t> fb, lg;
    for (uint32_t h = 0; h < nh; h++) {
      DevResult& d = B->res_host[h];
      std::memset(&d, 0, sizeof d);
      d.fail_op = TBC_NO_OP; d.prev_ok_op = TBC_NO_OP; d.final_state = B->model.init;
      if (hist_back[h].status != 0) { d.valid = TBC_UNKNOWN; continue; }
      if (hist_back[h].n_ret == 0) { d.valid = TBC_VALID; by_sweep[h] = 1; continue; }
      const SegResult* sg = &B->seg_host[(size_t)h * B->max_segs * SL];
      tbc_sweep_verdict v{};
      (void)tbc_sweep_compose(sg, B->max_segs, hist_back[h].n_ret, &v);
      const bool give_up = bh_back[h].status != 0 || v.valid == TBC_UNKNOWN;
      const uint32_t fail_seg = v.valid == TBC_INVALID ? v.fail_seg : kInf, fail_level = v.fail_level;
      const bool ended = v.valid == TBC_VALID;
      const uint32_t* live_in = v.live_in;
      d.steps = v.probes; d.probes = v.probes; d.visited = v.configs_total; d.backtracks = v.subrounds; d.max_depth = v.max_level;
      if (std::getenv("TBC_DEBUG")) {
        uint32_t longest = 0; uint64_t maxp = 0;
        for (uint32_t q = 0; q < B->max_segs * SL; q++) if (sg[q].status == kSegOk) { longest = std::max(longest, sg[q].F1 - sg[q].F0); maxp = std::max<uint64_t>(maxp, sg[q].probes); }
        std::fprintf(stderr, "[tbc sweep] history %u: %u wavefronts, longest segment %u levels, most probes in one %llu, largest level %llu, verdict %d\n",
                     h, v.n_wavefronts, longest, (unsigned long long)maxp, (unsigned long long)v.max_level, v.valid);
      }
      if (give_up || (fail_seg == kInf && !ended)) { fb.push_back(h); lg.push_back(B->bh[h].tab_log2); continue; }
      by_sweep[h] = 1;
      if (fail_seg == kInf) {
        d.valid = TBC_VALID;
        const bool regfam = B->model.kind == TBC_MODEL_REGISTER || B->model.kind == TBC_MODEL_CAS_REGISTER;
        if (regfam && B->vpad > 1 && v.final_bits) { const uint32_t sb = (uint32_t)__builtin_ctz(v.final_bits); d.final_state = sb == 0 ? TBC_NIL : (int32_t)sb - 1; }
        else d.final_state = (int32_t)v.end_state;
        continue;
      }
      d.valid = TBC_INVALID; d.max_front = fail_level;
      const uint32_t first = fail_level ? fail_level - 1 : 0;
      uint32_t two[2] = {TBC_NO_OP, TBC_NO_OP};
      HIP_TRY(hipMemcpyAsync(two, B->d_ret_op.p + hist_back[h].ret_off + first, (fail_level ? 2 : 1) * 4, hipMemcpyDeviceToHost, s));
      // :configs = the level in front of the failing completion, restricted to what the live origins reach: every
      // slice of the failing segment that holds a live origin appends its part
      HIP_TRY(hipMemsetAsync(&B->d_results.p[h].n_configs, 0, 4, s));
      for (uint32_t j = 0; j < SL; j++) if (live_in[j] && sg[(size_t)fail_seg * SL + j].status == kSegOk) {
        SweepArgs da = swa;
        da.dump_hist = h; da.dump_seg = fail_seg; da.dump_slice = j; da.stop_level = fail_level; da.live_mask = live_in[j];
        da.dump_cfg = B->d_cfg.p + (uint64_t)h * kCfgCap * (2 + B->mask_words);
        da.dump_count = &B->d_results.p[h].n_configs;
        da.seg_list = nullptr;
        (void)launch_sweep(da, s);
        HIP_TRY(hipGetLastError());
      }
      HIP_TRY(hipMemcpyAsync(&d.n_configs, &B->d_results.p[h].n_configs, 4, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      d.fail_op = fail_level ? two[1] : two[0];
      d.prev_ok_op = fail_level ? two[0] : TBC_NO_OP;
    }
    B->last_fallback = (uint32_t)fb.size(); B->last_segments = 0;
    for (const SegResult& g : B->seg_host) B->last_segments += g.status == kSegOk;
    if (!fb.empty()) {
      for (uint32_t h : fb) final_log2[h] = B->bh[h].tab_log2;
      tbc_status st = scratch_pass(B, fb, lg, true, hist_back, bh_back);
      if (st != TBC_OK) return st;
      touched_work = true;
    }
  }
  // wide-schedule histories whose open-call lists did not fit: sequential kernel
  if (beam) {
    std::vector<uint32_t> fb, lg;
    for (uint32_t h = 0; h < nh; h++)
      if (hist_back[h].status == 0 && bh_back[h].status != 0) {
        if (B->model.kind == TBC_MODEL_SET || B->model.kind == TBC_MODEL_BANK) {
          set_error("history %u: open-call lists exceed their arena and set / bank have no sequential kernel", h);
          return TBC_ERR_UNSUPPORTED;
        }
        uint32_t l = B->hist[h].tab_log2;
        fb.push_back(h); lg.push_back(l); final_log2[h] = l; is_seq[h] = 1;
      }
    if (!fb.empty()) {
      // the sequential kernel reads Hist.status only; clear the wide-schedule flag for it
      tbc_status st = scratch_pass(B, fb, lg, false, hist_back, bh_back);
      if (st != TBC_OK) return st;
      touched_work = true;
    }
  }
  std::vector<uint32_t> width_of(nh, B->width);
  // overflow retries: 16x larger visited set each time, up to max_visited_bytes
  const uint64_t arena_budget = 32ull << 30;
  for (;;) {
    std::vector<uint32_t> pend_seq, lg_seq, pend_beam, lg_beam;
    for (uint32_t h = 0; h < nh; h++) {
      if (B->res_host[h].valid == TBC_UNKNOWN && B->res_host[h].cause == TBC_CAUSE_VISITED_FULL) {
        const uint64_t wpe = is_seq[h] ? KW : EW;
        if (!is_seq[h]) final_log2[h] = std::max(final_log2[h], B->res_host[h].tab_log2);   // grown inside the kernel already
        uint32_t lg = final_log2[h] + 4;
        while (lg > final_log2[h] && ((1ull << lg) * wpe * 8 > max_bytes || (!is_seq[h] && lg > kBeamMaxTabLog2))) lg--;
        if (lg > final_log2[h]) {
          if (is_seq[h]) { pend_seq.push_back(h); lg_seq.push_back(lg); }
          else { pend_beam.push_back(h); lg_beam.push_back(lg); }
        }
      }
    }
    if (pend_seq.empty() && pend_beam.empty()) break;
    for (int pass = 0; pass < 2; pass++) {
      const std::vector<uint32_t>& pend = pass ? pend_beam : pend_seq;
      const std::vector<uint32_t>& lgs = pass ? lg_beam : lg_seq;
      const uint64_t wpe = pass ? (uint64_t)EW + 1 : KW;     // + the stack words
      size_t pos = 0;
      while (pos < pend.size()) {
        std::vector<uint32_t> grp, glg;
        uint64_t bytes = 0;
        while (pos < pend.size()) {
          const uint64_t need = (1ull << lgs[pos]) * wpe * 8;
          if (!grp.empty() && (bytes + need > arena_budget || (pass == 1 && width_of[pend[pos]] != width_of[grp[0]]))) break;
          grp.push_back(pend[pos]); glg.push_back(lgs[pos]); final_log2[pend[pos]] = lgs[pos];
          bytes += need; pos++;
        }
        tbc_status st = scratch_pass(B, grp, glg, pass == 1, hist_back, bh_back, pass == 1 ? width_of[grp[0]] : 0, kCountExact, nullptr,
                                     (pass == 1 && count_budget) ? (int64_t)count_budget : -1);
        if (st != TBC_OK) return st;
        touched_work = true;
      }
    }
  }
  if (stall_on) {
    std::vector<uint32_t> stalled;
    for (uint32_t h = 0; h < nh; h++)
      if (!is_seq[h] && hist_back[h].status == 0 && B->res_host[h].valid == TBC_UNKNOWN && B->res_host[h].cause == TBC_CAUSE_STEP_LIMIT) stalled.push_back(h);
    if (!stalled.empty()) {
      handed.resize(nh);
      tbc_status st = hand_over_stalled(B, stalled, handed);
      if (st != TBC_OK) return st;
      for (uint32_t h : stalled) was_handed[h] = 1;
      HIP_TRY(hipSetDevice(B->device));
    }
  }
  // ---- count form: the histories the budgeted exact search left undecided (oracle/wgl_count.c; tests/test_count_form.py states the
  // same pipeline over the oracle).  (1) The RELAXED search -- every class of crashed calls an unlimited supply, counts ignored: a
  // superset of the linearizations over a config space no larger than a crash-free history's -- either finds a linearization (then
  // the exact search simply needs longer: once more, without a budget) or ends INVALID at completion t: the history is invalid and
  // its first bad completion is t or earlier.  (2) The exact search of the PREFIX of t completions: a linearization of it (one
  // depth-first descent, not an exhaustion) pins the failing completion at t; if there is none its own exhaustion names an earlier one.
  if (count_budget) {
    std::vector<uint32_t> pend;
    for (uint32_t h = 0; h < nh; h++)
      if (!is_seq[h] && B->res_host[h].valid == TBC_UNKNOWN && B->res_host[h].cause == TBC_CAUSE_STEP_LIMIT) pend.push_back(h);
    if (!pend.empty()) {
      struct Acc { uint64_t steps, visited, probes, backtracks, max_depth; };
      std::vector<Acc> acc(nh, Acc{0, 0, 0, 0, 0});
      const auto bank = [&](const std::vector<uint32_t>& grp) {
        for (uint32_t h : grp) { const DevResult& d = B->res_host[h]; Acc& a = acc[h]; a.steps += d.steps; a.visited += d.visited; a.probes += d.probes; a.backtracks += d.backtracks; a.max_depth = std::max(a.max_depth, d.max_depth); }
      };
      // one pass over `grp` in a scratch arena; a history whose visited set fills up is taken again with a 16x larger one
      const auto run_pass = [&](const std::vector<uint32_t>& grp, uint32_t mode, const std::vector<uint32_t>* targets) -> tbc_status {
        std::vector<uint32_t> todo = grp, tg, lgs;
        if (targets) tg = *targets;
        for (uint32_t h : todo) {
          uint32_t lg = std::max(final_log2[h], ceil_log2(64ull * std::max<uint64_t>(B->hist[h].n_ops, 1)));
          while (lg > 10 && ((1ull << lg) * EW * 8 > max_bytes || lg > kBeamMaxTabLog2)) lg--;
          lgs.push_back(lg);
        }
        while (!todo.empty()) {
          size_t pos = 0;
          while (pos < todo.size()) {
            std::vector<uint32_t> g, glg, gtg;
            uint64_t bytes = 0;
            while (pos < todo.size()) {
              const uint64_t need = (1ull << lgs[pos]) * ((uint64_t)EW + 2) * 8;
              if (!g.empty() && bytes + need > arena_budget) break;
              g.push_back(todo[pos]); glg.push_back(lgs[pos]); if (targets) gtg.push_back(tg[pos]);
              final_log2[todo[pos]] = lgs[pos]; bytes += need; pos++;
            }
            tbc_status st = scratch_pass(B, g, glg, true, hist_back, bh_back, 0, mode, targets ? &gtg : nullptr, 0);
            if (st != TBC_OK) return st;
          }
          std::vector<uint32_t> again, alg, atg;
          for (size_t i = 0; i < todo.size(); i++) {
            const DevResult& d = B->res_host[todo[i]];
            if (d.valid != TBC_UNKNOWN || d.cause != TBC_CAUSE_VISITED_FULL) continue;
            uint32_t lg = lgs[i] + 4;
            while (lg > lgs[i] && ((1ull << lg) * EW * 8 > max_bytes || lg > kBeamMaxTabLog2)) lg--;
            if (lg > lgs[i]) { again.push_back(todo[i]); alg.push_back(lg); if (targets) atg.push_back(tg[i]); }
          }
          todo.swap(again); lgs.swap(alg); tg.swap(atg);
        }
        return TBC_OK;
      };
      bank(pend);
      // (what the relaxed SWEEP already decided is not searched again: refuted at a completion -> the prefix pass; valid under the
      // relaxation -> the exact search without a budget; only the others -- no sweep, or a burst that outgrew its sets -- take the relaxed search)
      std::vector<uint32_t> pend_dfs;
      for (uint32_t h : pend) if (rs_level[h] == kInf && !rs_valid[h]) pend_dfs.push_back(h);
      tbc_status st = pend_dfs.empty() ? TBC_OK : run_pass(pend_dfs, kCountRelaxed, nullptr);
      if (st != TBC_OK) return st;
      bank(pend_dfs);
      for (uint32_t h : pend) {
        DevResult& r = B->res_host[h];
        if (rs_valid[h]) { r.valid = TBC_VALID; r.cause = TBC_CAUSE_NONE; }
        if (rs_level[h] == kInf) continue;
        const uint32_t t = rs_level[h], first = t ? t - 1 : 0;
        uint32_t two[2] = {TBC_NO_OP, TBC_NO_OP};
        HIP_TRY(hipMemcpy(two, B->d_ret_op.p + hist_back[h].ret_off + first, (t ? 2 : 1) * 4, hipMemcpyDeviceToHost));
        r.valid = TBC_INVALID; r.cause = TBC_CAUSE_NONE; r.max_front = t; r.n_configs = 0;
        r.fail_op = t ? two[1] : two[0]; r.prev_ok_op = t ? two[0] : TBC_NO_OP;
      }
      std::vector<uint32_t> longer, prefix, targets;
      std::vector<DevResult> relaxed(nh);
      for (uint32_t h : pend) {
        const DevResult& r = B->res_host[h];
        relaxed[h] = r;
        if (r.valid == TBC_VALID) longer.push_back(h);
        else if (r.valid == TBC_INVALID && r.max_front != 0) { prefix.push_back(h); targets.push_back(r.max_front); }
        // (INVALID at the very first completion: nothing to pin; UNKNOWN -- a time limit -- stays UNKNOWN)
      }
      if (!longer.empty()) { if ((st = run_pass(longer, kCountExact, nullptr)) != TBC_OK) return st; }
      if (!prefix.empty()) {
        if ((st = run_pass(prefix, kCountExact, &targets)) != TBC_OK) return st;
        for (uint32_t h : prefix) {
          DevResult& d = B->res_host[h];
          if (d.valid != TBC_VALID) continue;             // (its own exhaustion names an earlier completion, or it ran into a limit)
          d.valid = TBC_INVALID; d.cause = TBC_CAUSE_NONE; d.depth = 0; d.n_configs = 0;
          d.max_front = relaxed[h].max_front; d.fail_op = relaxed[h].fail_op; d.prev_ok_op = relaxed[h].prev_ok_op;
        }
      }
      std::vector<uint8_t> third(nh, 0);
      for (uint32_t h : longer) third[h] = 1;
      for (uint32_t h : prefix) third[h] = 1;
      for (uint32_t h : pend) {          // counters: the sum over the passes a history went through
        DevResult& d = B->res_host[h]; const Acc& a = acc[h];
        if (third[h]) { d.steps += a.steps; d.visited += a.visited; d.probes += a.probes; d.backtracks += a.backtracks; d.max_depth = std::max(d.max_depth, a.max_depth); }
        else { d.steps = a.steps; d.visited = a.visited; d.probes = a.probes; d.backtracks = a.backtracks; d.max_depth = a.max_depth; }   // (the relaxed pass is banked already)
      }
      touched_work = true;
    }
  }
  if (touched_work) {   // restore the identity work list for the next run
    std::vector<uint32_t> work(nh);
    for (uint32_t h = 0; h < nh; h++) work[h] = h;
    HIP_TRY(hipMemcpyAsync(B->d_work.p, work.data(), nh * 4, hipMemcpyHostToDevice, s));
  }
  HIP_TRY(hipEventRecord(B->ev[5], s));
  HIP_TRY(hipStreamSynchronize(s));
  TRACE("run: retries done");

  if (B->opts.want_witness) {
    B->witness_host.resize(B->total_ops ? B->total_ops : 1);
    HIP_TRY(hipMemcpy(B->witness_host.data(), B->d_witness.p, B->total_ops * 4, hipMemcpyDeviceToHost));
  }
  TRACE("run: witness copied");

  float ms;
  for (int i = 0; i < 3; i++) {
    HIP_TRY(hipEventElapsedTime(&ms, B->ev[i], B->ev[i + 1]));
    B->timing_ns[i] = (uint64_t)(ms * 1e6);
  }
  HIP_TRY(hipEventElapsedTime(&ms, B->ev[4], B->ev[5]));
  B->timing_ns[3] = (uint64_t)(ms * 1e6);
  B->turn_wait_ns = 0;
  if (B->lanes && B->ev_turn && phase == 0) {      // the search proper: from its turn on the device to its end
    HIP_TRY(hipEventElapsedTime(&ms, B->ev[2], B->ev_turn));
    B->turn_wait_ns = (uint64_t)(ms * 1e6);
    HIP_TRY(hipEventElapsedTime(&ms, B->ev_turn, B->ev[3]));
    B->timing_ns[2] = (uint64_t)(ms * 1e6);
  }
  TRACE("run: timings read");

  std::memset(&B->sum, 0, sizeof B->sum);
  const uint64_t t_end = now_ns();
  tbc_status worst = TBC_OK;
  for (uint32_t h = 0; h < nh; h++) {
    const DevResult& d = B->res_host[h];
    B->sum.steps += d.steps; B->sum.visited += d.visited; B->sum.probes += d.probes;
    B->sum.backtracks += d.backtracks; B->sum.max_depth = std::max(B->sum.max_depth, d.max_depth);
    if (!is_seq[h] && d.tab_log2 > final_log2[h]) final_log2[h] = d.tab_log2;
    B->sum.table_slots += 1ull << final_log2[h];
    if (hist_back[h].status != 0 && worst == TBC_OK) {
      worst = (tbc_status)hist_back[h].status;
      set_error("history %u rejected by the pack kernel: %s", h, tbc_strerror((int)hist_back[h].status));
    }
    if (was_handed[h]) {          // answered by the small batch it was handed to: its result, the first pass's counters added
      const tbc_result& g = handed[h];
      B->sum.steps += g.counters.steps; B->sum.visited += g.counters.visited; B->sum.probes += g.counters.probes; B->sum.backtracks += g.counters.backtracks;
      if (results) {
        tbc_result& r = results[h];
        r = g;
        r.counters.steps += d.steps; r.counters.visited += d.visited; r.counters.probes += d.probes; r.counters.backtracks += d.backtracks;
        r.counters.max_depth = std::max<uint64_t>(r.counters.max_depth, d.max_depth);
        r.counters.ns_pack = B->timing_ns[1]; r.counters.ns_search = B->timing_ns[2] + B->timing_ns[3]; r.counters.ns_total = t_end - t_start;
      }
      continue;
    }
    if (!results) continue;
    tbc_result& r = results[h];
    std::memset(&r, 0, sizeof r);
    r.valid = d.valid; r.cause = d.cause;
    r.analyzer = B->sweep ? (by_sweep[h] ? TBC_ALG_LINEAR : TBC_ALG_WGL)
                          : (B->opts.algorithm == TBC_ALG_LINEAR ? TBC_ALG_LINEAR : TBC_ALG_WGL);
    r.fail_op = TBC_NO_OP; r.prev_ok_op = TBC_NO_OP; r.search_width = (B->lanes && !is_seq[h] && !by_sweep[h]) ? 1u : B->width;
    if (d.valid == TBC_INVALID) {
      r.fail_op = d.fail_op; r.prev_ok_op = d.prev_ok_op;
      tbc_status cs = fill_configs(B, h, d, &r);
      if (cs != TBC_OK) return cs;
    }
    if (d.valid == TBC_VALID) {
      r.final_state = d.final_state; r.n_witness = d.depth;
      if (B->opts.want_witness) {
        r.witness = B->witness_host.data() + B->hist[h].op_off;
        if (by_sweep[h]) { r.witness = nullptr; r.n_witness = 0; }      // knossos.linear returns configs, not a linearization
        // (under branch lists the normalised root may pass every completion by itself -- a history of reads of nil / of the initial
        // value: an empty chain, whose expansion is exactly those reads)
        else if ((B->rules & kRuleEager) && !is_seq[h] && (d.depth || ((B->rules & kRuleBranch) && hist_back[h].n_ret != 0))) {
          tbc_status ws = expand_eager_witness(B, h, r.witness, &r.n_witness);
          if (ws != TBC_OK) return ws;
        }
      }
    }
    r.counters.steps = d.steps; r.counters.visited = d.visited; r.counters.probes = d.probes;
    r.counters.backtracks = d.backtracks; r.counters.max_depth = d.max_depth;
    r.counters.table_slots = 1ull << final_log2[h];
    r.counters.ns_pack = B->timing_ns[1]; r.counters.ns_search = B->timing_ns[2] + B->timing_ns[3];
    r.counters.ns_total = t_end - t_start;
  }
  B->sum.ns_pack = B->timing_ns[1]; B->sum.ns_search = B->timing_ns[2] + B->timing_ns[3];
  B->sum.ns_total = t_end - t_start;
  if (guard_on() && guard_check("tbc_batch_run", B) != 0) { set_error("TBC_GUARD: a kernel wrote past a device arena (see stderr)"); return TBC_ERR_HIP; }
  return worst;
}

tbc_status tbc_batch_run(tbc_batch* b, tbc_result* results) {
  if (!b) { set_error("tbc_batch_run: null batch"); return TBC_ERR_INVALID_ARG; }
  struct OwnerScope { const void* prev; size_t prev_nth; OwnerScope(const void* o) : prev(t_guard_owner), prev_nth(t_guard_nth) { t_guard_owner = o; t_guard_nth = 1000; }
                      ~OwnerScope() { t_guard_owner = prev; t_guard_nth = prev_nth; } } scope(b);      // (scratch arenas of a run: the batch's, numbered from 1000)
  try {
    return batch_run_impl(b, results);
  } catch (const std::bad_alloc&) {
    set_error("host allocation failed");
    return TBC_ERR_OOM;
  } catch (...) {
    set_error("unexpected exception");
    return TBC_ERR_HIP;
  }
}

tbc_status tbc_batch_set_shard(tbc_batch* b, uint32_t rank, uint32_t world) {
  if (!b || world == 0 || rank >= world) { set_error("tbc_batch_set_shard: bad rank / world"); return TBC_ERR_INVALID_ARG; }
  if (!b->sweep) { set_error("tbc_batch_set_shard: this batch does not run the level sweep (TBC_ALG_LINEAR, <= 64 process slots)"); return TBC_ERR_UNSUPPORTED; }
  b->shard_rank = rank; b->shard_world = world;
  return TBC_OK;
}

tbc_status tbc_batch_sweep_partial(tbc_batch* b) {
  if (!b || !b->sweep) { set_error("tbc_batch_sweep_partial: not a sweep batch"); return TBC_ERR_INVALID_ARG; }
  try { return batch_run_impl(b, nullptr, 1); }
  catch (const std::bad_alloc&) { set_error("host allocation failed"); return TBC_ERR_OOM; }
  catch (...) { set_error("unexpected exception"); return TBC_ERR_HIP; }
}

tbc_status tbc_batch_sweep_table(const tbc_batch* b, void** device_ptr, uint64_t* bytes) {
  if (!b || !b->sweep || !device_ptr || !bytes) { set_error("tbc_batch_sweep_table: not a sweep batch"); return TBC_ERR_INVALID_ARG; }
  *device_ptr = b->d_sres.p;
  *bytes = (uint64_t)b->seg_host.size() * sizeof(SegResult);
  return TBC_OK;
}

tbc_status tbc_batch_sweep_finish(tbc_batch* b, const void* merged, uint64_t merged_bytes, tbc_result* results) {
  if (!b || !b->sweep || !merged) { set_error("tbc_batch_sweep_finish: not a sweep batch"); return TBC_ERR_INVALID_ARG; }
  if (merged_bytes != (uint64_t)b->seg_host.size() * sizeof(SegResult)) {
    set_error("tbc_batch_sweep_finish: merged table is %llu bytes, this batch's table is %llu (tbc_batch_sweep_table)",
              (unsigned long long)merged_bytes, (unsigned long long)(b->seg_host.size() * sizeof(SegResult)));
    return TBC_ERR_INVALID_ARG;
  }
  if (!b->partial_done) { set_error("tbc_batch_sweep_finish without tbc_batch_sweep_partial"); return TBC_ERR_INVALID_ARG; }
  try {
    std::memcpy(b->seg_host.data(), merged, b->seg_host.size() * sizeof(SegResult));
    return batch_run_impl(b, results, 2);
  }
  catch (const std::bad_alloc&) { set_error("host allocation failed"); return TBC_ERR_OOM; }
  catch (...) { set_error("unexpected exception"); return TBC_ERR_HIP; }
}

namespace {
// bitwise OR of `world` relation tables lying back to back in device memory into `dst` (every record is written by exactly
// one rank and all zero on the others)
__global__ void sweep_or_kernel(uint64_t* dst, const uint64_t* gathered, uint64_t words, uint32_t world) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= words) return;
  uint64_t v = 0;
  for (uint32_t r = 0; r < world; r++) v |= gathered[(uint64_t)r * words + i];
  dst[i] = v;
}
}  // namespace

tbc_status tbc_batch_sweep_merge(tbc_batch* b, const void* gathered_device, uint64_t gathered_bytes, uint32_t world, tbc_result* results) {
  if (!b || !b->sweep || !gathered_device || world == 0) { set_error("tbc_batch_sweep_merge: not a sweep batch"); return TBC_ERR_INVALID_ARG; }
  const uint64_t bytes = (uint64_t)b->seg_host.size() * sizeof(SegResult);
  if (gathered_bytes != bytes * world) {
    set_error("tbc_batch_sweep_merge: %llu bytes gathered, %u tables of %llu expected", (unsigned long long)gathered_bytes, world, (unsigned long long)bytes);
    return TBC_ERR_INVALID_ARG;
  }
  if (!b->partial_done) { set_error("tbc_batch_sweep_merge without tbc_batch_sweep_partial"); return TBC_ERR_INVALID_ARG; }
  try {
    HIP_TRY(hipSetDevice(b->device));
    const uint64_t words = bytes / 8;
    hipLaunchKernelGGL(sweep_or_kernel, dim3((uint32_t)((words + 255) / 256)), dim3(256), 0, b->stream, (uint64_t*)b->d_sres.p, (const uint64_t*)gathered_device, words, world);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(b->seg_host.data(), b->d_sres.p, bytes, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return batch_run_impl(b, results, 2);
  }
  catch (const std::bad_alloc&) { set_error("host allocation failed"); return TBC_ERR_OOM; }
  catch (...) { set_error("unexpected exception"); return TBC_ERR_HIP; }
}

tbc_status tbc_batch_last_timing(const tbc_batch* b, uint64_t ns[4]) {
  if (!b || !ns) return TBC_ERR_INVALID_ARG;
  for (int i = 0; i < 4; i++) ns[i] = b->timing_ns[i];
  return TBC_OK;
}

tbc_status tbc_batch_last_turn_wait(const tbc_batch* b, uint64_t* ns) {
  if (!b || !ns) { set_error("null argument"); return TBC_ERR_INVALID_ARG; }
  *ns = b->turn_wait_ns;
  return TBC_OK;
}

tbc_status tbc_batch_last_counters(const tbc_batch* b, tbc_counters* out) {
  if (!b || !out) return TBC_ERR_INVALID_ARG;
  *out = b->sum;
  return TBC_OK;
}

uint64_t tbc_batch_device_bytes(const tbc_batch* b) { return b ? b->device_bytes : 0; }
uint32_t tbc_batch_search_width(const tbc_batch* b) { return b ? (b->lanes ? 1u : b->width) : 0; }
uint32_t tbc_batch_lanes_per_history(const tbc_batch* b) { return b ? (b->lanes ? b->lanes : 64u) : 0; }
uint32_t tbc_batch_list_order(const tbc_batch* b) {
  if (!b) return 0;
  const uint32_t lo = b->list_order();          // PackOpenArgs' numbering -> TBC_ORDER_*
  return lo >= 16u ? lo : lo + 1u;
}

tbc_status tbc_batch_sweep_info(const tbc_batch* b, tbc_sweep_info* out) {
  if (!b || !out) return TBC_ERR_INVALID_ARG;
  out->enabled = b->sweep ? 1u : 0u; out->seg_target = b->seg_target; out->max_segs = b->max_segs;
  out->cut_open = b->cut_open; out->n_dom = b->n_dom; out->n_segments = b->last_segments; out->n_fallback = b->last_fallback;
  return TBC_OK;
}

void tbc_batch_destroy(tbc_batch* b) {
  if (!b) return;
  (void)hipSetDevice(b->device);
  delete b;
}

tbc_status tbc_check(const tbc_ops* ops, const tbc_model* model, const tbc_opts* opts, tbc_result* out) {
  if (!ops || !model || !opts || !out) { set_error("tbc_check: null argument"); return TBC_ERR_INVALID_ARG; }
  const uint64_t t0 = now_ns();
  uint64_t op_off[2] = {0, ops->n};
  tbc_batch_desc d{};
  d.n_hist = 1; d.op_off = op_off; d.n_events = &ops->n_events; d.n_process = &ops->n_process; d.cols = *ops;
  tbc_batch* B = nullptr;
  Ctx* ctx = ctx_acquire((int)opts->device);      // null (no device ...): the create below reports why
  t_ctx = ctx;
  tbc_status s = tbc_batch_create(&d, model, opts, &B);
  if (s != TBC_OK) { t_ctx = nullptr; if (ctx) ctx_release(ctx); return s; }
  TRACE("check: batch created");
  s = tbc_batch_run(B, out);
  TRACE("check: run returned");
  if (s == TBC_OK && out->witness) {   // hand the witness over: the batch dies here
    uint32_t* w = (uint32_t*)std::malloc((size_t)std::max(1u, out->n_witness) * 4);
    if (!w) { tbc_batch_destroy(B); t_ctx = nullptr; if (ctx) ctx_release(ctx); return TBC_ERR_OOM; }
    std::memcpy(w, out->witness, (size_t)out->n_witness * 4);
    out->witness = w;
  } else {
    out->witness = nullptr;
  }
  tbc_batch_destroy(B);
  t_ctx = nullptr;
  if (ctx) ctx_release(ctx);
  TRACE("check: destroyed");
  out->counters.ns_total = now_ns() - t0;
  return s;
}

void tbc_result_free(tbc_result* r) {
  if (!r) return;
  std::free(r->witness);
  r->witness = nullptr;
}

}  // extern "C"
