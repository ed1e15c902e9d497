// tbc_batch.h -- the host side of the library, shared by its translation units.  Round 6 cut tbc_api.hip (2,084 lines, two functions
// of 500 and 560 lines carrying every engine) into
//   batch_common.hip   persistent contexts, the arena guard, debug words
//   batch_create.hip   tbc_batch_create: the planners (model checks -> engines -> layout -> arenas -> inputs resident)
//   batch_run.hip      tbc_batch_run: the phases of a pass (first pass, sweep verdicts, fallbacks, retries, count-form pipeline, marshalling)
//   batch_shard.hip    one history over several GPUs (tbc_batch_set_shard ... tbc_batch_sweep_merge)
//   batch_stream.hip   fresh inputs into a batch's arenas (tbc_batch_map_input / tbc_batch_submit_input / tbc_batch_reload)
//   tbc_api.hip        tbc_check and the small getters
// Nothing here crosses the C-ABI (include/tbcheck.h does); nothing here runs on the device (tbc_internal.h is what the kernels share).
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <new>
#include <thread>
#include <vector>
#include "tbc_internal.h"

namespace tbc {

#define HIP_TRY(expr)                                                             \
  do {                                                                            \
    hipError_t e_ = (expr);                                                       \
    if (e_ != hipSuccess) {                                                       \
      ::tbc::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return e_ == hipErrorOutOfMemory ? TBC_ERR_OOM : TBC_ERR_HIP;               \
    }                                                                             \
  } while (0)

inline uint32_t ceil_log2(uint64_t x) {
  uint32_t l = 0;
  while ((1ull << l) < x) l++;
  return l;
}

uint64_t now_ns();
bool device_is_gfx950(int dev);
uint32_t* debug_words();          // TBC_DEBUG=1: host-mapped progress words, else null
#define SYNC_TRACE(msg) do { if (std::getenv("TBC_SYNC_EACH")) { hipError_t e__ = hipStreamSynchronize(s); std::fprintf(stderr, "[tbc sync] %s -> %s\n", msg, hipGetErrorString(e__)); std::fflush(stderr); } } while (0)
#define TRACE(msg) do { if (std::getenv("TBC_DEBUG")) { std::fprintf(stderr, "[tbc %9.3f ms] %s\n", (double)(::tbc::now_ns() % 100000000000ull) / 1e6, msg); std::fflush(stderr); } } while (0)

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
extern thread_local Ctx* t_ctx;         // the context the calling thread's DevBufs draw from (tbc_check only)
Ctx* ctx_acquire(int device);
void ctx_release(Ctx* c);
// a scope in which the calling thread's buffers are NOT drawn from its context (an inner batch that owns its arenas): the context comes
// back whichever way the scope is left (round 5 lost it on an early return: ADVICE.md, hand_over_stalled)
struct CtxSuspend {
  Ctx* const saved;
  CtxSuspend() : saved(t_ctx) { t_ctx = nullptr; }
  ~CtxSuspend() { t_ctx = saved; }
};

// TBC_GUARD=1 (diagnostic, like TBC_DEBUG): every device arena gets 256 poisoned bytes behind it and tbc_batch_run checks them all when
// it ends -- a kernel that writes past an arena is named (allocation number, address, the first bad byte) instead of corrupting a
// neighbour silently.  Round 4 saw ONE bench run of five die with a GPU memory fault that nothing reproduced; this is how it was hunted
// (profiles/r05_guard_runs.txt).
// (an arena belongs to the batch being created or run by the allocating thread -- t_guard_owner; a run checks its own batch's arenas
// only: another thread's batch may be poisoning a re-used piece of its context's slab at that very moment -- the first version of this
// check read such bytes and cried wolf, ten times in a two-thread bench run)
bool guard_on();
constexpr size_t kGuardBytes = 256;
extern thread_local const void* t_guard_owner;
extern thread_local size_t t_guard_nth;          // the arena's number within its batch (allocation order of tbc_batch_create: names it)
void guard_add(const void* arena_end);
void guard_remove(const void* arena_end);
size_t guard_check(const char* when, const void* owner);   // arenas of `owner` whose guard bytes were overwritten (after the caller's stream is idle)

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
  // an arena a fresh input has outgrown (batch_stream.hip): the old one goes, a new one of `count` elements comes (contents undefined)
  tbc_status regrow(size_t count) { release(); return alloc(count); }
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

// ---- count form (tbc_internal.h, kRuleCount; specified in oracle/wgl_count.c): what the host works out of one history when the
// inputs become resident -- the re-used process slots of the live calls, the classes of the crashed calls with their members'
// invocation ranks, the layout of the count vector.
struct CountHist {
  std::vector<uint64_t> words;       // [2 * n_classes words of OpRec][members of class 0, sentinel, members of class 1, sentinel, ...]
  uint32_t n_classes = 0, n_slots = 1;
  uint64_t top[kCountWords] = {0, 0};
};

// Element counts of a batch's per-history arenas: the running sums the layout of the histories leaves behind (batch_create.hip,
// layout_histories) -- what tbc_batch_create allocates, and what a fresh input (batch_stream.hip) must fit into or grow.
struct LayoutTotals {
  uint64_t rec_n = 0, seg_n = 0, bm_n = 0, frame_n = 0, tab_n = 0;             // every batch
  uint64_t boff_n = 0, bocc_n = 0, blst_n = 0, bstack_n = 0, btab_n = 0;      // wide schedule
  uint64_t max_ops = 1, total_ops = 0;
};

}  // namespace tbc

// A fresh input on its way into a batch (batch_stream.hip): the wire columns are in (or on their way to) one of the batch's two device
// stages, the descriptors of its histories wait here until the run that consumes it makes them the batch's own.
struct tbc_pending_input {
  uint32_t stage = 0;                 // which device stage holds its wire columns
  uint32_t slot = 0;                  // ... copied from this pinned host slot
  uint32_t n_hist = 0;
  uint64_t total_ops = 0, max_ops = 1;
  std::vector<tbc::Hist> hist;
  std::vector<tbc::BeamHist> bh;
  tbc::LayoutTotals tot;
  // a count-form batch: the input's classes of crashed calls, planned on the host when it was submitted (batch_stream.hip)
  std::vector<tbc::CountHist> count_hist;
  std::vector<uint64_t> cmem;         // ... their records and members, all histories back to back (what goes to BeamArgs.cmem)
  uint64_t lst_total = 0;             // ... and how many entries its per-front lists hold (a count-form batch has no fallback for lists that do not fit)
};

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
  std::vector<tbc::Hist> hist;          // host mirror
  std::vector<uint16_t> table_host;
  // device arenas
  tbc::DevBuf<uint8_t> d_f;
  tbc::DevBuf<int32_t> d_a, d_b, d_proc;
  tbc::DevBuf<uint32_t> d_inv, d_ret;
  tbc::DevBuf<tbc::Hist> d_hist;
  tbc::DevBuf<tbc::Rec> d_rec;
  tbc::DevBuf<uint32_t> d_seg, d_ret_slot, d_ret_op, d_bitmap, d_wpre, d_frames, d_witness, d_work, d_queue;
  tbc::DevBuf<uint64_t> d_tab;
  tbc::DevBuf<tbc::DevResult> d_results;
  tbc::DevBuf<uint16_t> d_table;
  tbc::DevBuf<int32_t> d_pool_vals;
  uint32_t pool_len = 0;
  tbc::DevBuf<uint64_t> d_cfg;           // configs at the failing front, kCfgCap records per history
  // wide schedule (search_width > 1)
  uint32_t width = 1;
  // u64 words per front record (0 = plain rdm rows): the compact 64 B form where one mask word and six row entries do
  uint32_t front_words() const { return !lanes ? 0u : ((rules & tbc::kRuleEager) && tbc::front_compact_ok(n_dom, mask_words)) ? tbc::kFrontCompactWords : tbc::front_stride(vpad, mask_words); }
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
    return width > 1 && mask_words == 1 && vpad <= 32 && !(rules & tbc::kRuleCount) && !sweep && opts.round_budget == 0 &&
           (model.kind == TBC_MODEL_REGISTER || model.kind == TBC_MODEL_CAS_REGISTER);
  }
  // PackOpenArgs.list_order: 0 = slot order, 1 = completion, 2 = completion with the :write calls last, 16 + W
  uint32_t list_order() const {
    if (order_override != kNoOrderOverride) return order_override;          // a pass of the order restarts (batch_run.hip): its own order
    if (!list_order_applies() || opts.list_order == TBC_ORDER_SLOT) return 0u;
    if (opts.list_order == TBC_ORDER_DEFAULT) return kDefaultListOrder;
    return opts.list_order >= 16u ? opts.list_order : opts.list_order - 1u;        // TBC_ORDER_COMPLETION = 2 -> 1, TBC_ORDER_WRITES_LAST = 3 -> 2
  }
  // ORDER RESTARTS (batch_run.hip, order_restarts): a history of the wide depth-first search that has not ended within a budget of probes
  // is searched again from scratch in the next list order.  While such a pass runs the fronts' lists are walked again in ITS order.
  static constexpr uint32_t kNoOrderOverride = 0xFFFFFFFFu;
  uint32_t order_override = kNoOrderOverride;
  std::vector<uint32_t> order_of_hist;      // last run: the list order (PackOpenArgs numbering) each history was answered in
  // a RACE of list orders (batch_run.hip, race_orders): this batch is one of several that search the same histories, each in its own order;
  // a history one of them decides sets its word, and the others' searches of it stop when they see it (BeamArgs.abort / abort_set)
  const uint32_t* ext_abort = nullptr;
  uint32_t* ext_abort_set = nullptr;
  const uint32_t* ext_abort_map = nullptr;  // which word is history h's (several orders' replicas of one history share a word)
  std::vector<uint32_t> hist_order;         // per history its own list order (PackOpenArgs numbering), or empty: the batch's for all
  tbc::DevBuf<uint32_t> d_order;
  tbc::DevBuf<uint32_t> d_park;             // wide kernel: the search state of every history as the last launch left it (BeamArgs.park)
  uint32_t last_raced = 0;                  // last run: histories that went into a race of orders
  // progress (tbc_batch_progress; the reference's knossos.search reports while it runs): words in HOST memory the search kernels count
  // decided histories into (word 0) and the run writes its phase to (word 1); read by another thread while tbc_batch_run is out
  uint32_t* progress = nullptr;
  tbc::DevBuf<uint32_t> d_progress;         // the count itself, in HBM (wave_env.h count_decided)
  mutable std::atomic<uint32_t> progress_seen{0};     // (two publishers can land out of order: a reader never reports less than it has seen)
  std::atomic<uint64_t> run_t0{0};
  std::atomic<uint32_t> running{0};
  bool order_restarts_apply() const {
    return width > 1 && !lanes && !count_form && list_order_applies() && opts.list_order == TBC_ORDER_DEFAULT && opts.max_steps == 0 &&
           !(opts.dominance & TBC_DOM_NO_ORDER_RESTARTS);
  }
  std::vector<tbc::BeamHist> bh;
  tbc::DevBuf<tbc::BeamHist> d_bh;
  tbc::DevBuf<uint32_t> d_off, d_ncr, d_stack;
  tbc::DevBuf<tbc::OpRec> d_lst, d_crashed;   // per-front open-call lists / crashed calls, whole records
  tbc::DevBuf<uint8_t> d_slot8;          // completion slots as bytes
  tbc::DevBuf<uint8_t> d_rk8;            // narrow kernel: read kind per rank
  bool lookahead = false;           // wide single-wave schedule, register family, tbc_opts.lookahead != 1
  tbc::DevBuf<uint64_t> d_look;          // lookahead records per completion rank
  tbc::DevBuf<uint32_t> d_dstack;        // second stack per history: configs the lookahead set aside
  tbc::DevBuf<uint32_t> d_looktmp;
  // level sweep (jit_sweep.hip): TBC_ALG_LINEAR, and TBC_ALG_COMPETITION on small batches that want no witness
  bool sweep = false;
  uint32_t max_segs = 1, seg_target = 0, cut_open = 0, n_dom = 1;
  tbc::DevBuf<uint32_t> d_cuts, d_seglist;
  tbc::DevBuf<tbc::SegResult> d_sres;
  tbc::HostBuf<tbc::SegResult> seg_host;
  // the RELAXED sweep in front of a count-form search (round 5; oracle/sweep_ref.c sweep_set_relaxed, jit_sweep_wg_impl.h RLX): a handful
  // of histories with crashed calls, nobody asking for a witness or a schedule -- every class of crashed calls an unlimited supply, so
  // the sweep's cuts apply and an INVALID history is refuted by hundreds of wavefronts in milliseconds instead of by one wavefront's
  // exhaustion of the relaxed config space (0.7 / 2.0 s on the bench's tiers); the prefix search then pins the failing completion as before
  bool rsweep = false;
  tbc::DevBuf<uint32_t> d_zncr, d_reach, d_reach_hdr, d_abort;
  hipStream_t stream2 = nullptr;    // ... on a stream of its own, beside the exact search (the context's when borrowed)
  hipEvent_t ev2[2] = {};
  uint32_t* abort_one = nullptr;    // a pinned word holding 1: what the abort words are set from (an async copy wants pinned memory)
  uint32_t last_segments = 0, last_fallback = 0;
  uint32_t shard_rank = 0, shard_world = 1;      // tbc_batch_set_shard: this rank's share of the sweep's wavefronts
  bool partial_done = false;                     // a tbc_batch_sweep_partial is waiting for its tbc_batch_sweep_finish
  tbc::HostBuf<tbc::Hist> hist_back_m;                     // descriptors as the pack kernels left them (kept between
  tbc::HostBuf<tbc::BeamHist> bh_back_m;                   //   tbc_batch_sweep_partial and tbc_batch_sweep_finish)
  tbc::HostBuf<char> upload_stage;       // tbc_check: the block create uploads (pinned, the context's)
  bool inputs_fresh = false;        // tbc_check: create has just uploaded hist / bh / work with the columns -- the first run does not again
  uint32_t rules = 0;               // kRuleEager | kRuleTwin: wide single-wave schedule, register family, values 0..kMaxRuleValue
  uint32_t vpad = 0;                // entries per rdm row (nil + values), power of two
  tbc::DevBuf<uint64_t> d_twn, d_rdm;    // dominance tables (tbc_internal.h)
  tbc::DevBuf<uint64_t> d_occ, d_btab, d_pool;
  // count form (tbc_internal.h, kRuleCount): crashed calls as counts per class; the classes and their members, per history
  bool count_form = false;
  tbc::DevBuf<uint64_t> d_cmem;
  std::vector<tbc::CountHist> count_hist;       // (kept for the result marshalling: which crashed calls a count vector stands for)
  uint32_t reg_rules() const { return rules & (tbc::kRuleEager | tbc::kRuleTwin); }     // the register family's rules (their tables: twn, rdm)
  uint32_t epoch = 0;               // narrow kernel: the pass number its visited-set keys are tagged with (1..255; the arena is zeroed when it wraps)
  bool any_crashed = true;          // some op of the batch never completes (else the crashed-call arena is never read: one element)
  // u64 words per entry of the batch's own visited-set arena: the narrow kernel keeps no parent links when nobody wants a witness
  uint32_t tab_stride() const { return entry_words() - ((lanes && !opts.want_witness) ? 1u : 0u); }
  uint32_t entry_words() const { return mask_words + 2u + (count_form ? tbc::kCountWords : 0u); }   // u64 words per wide-schedule entry
  tbc::DevBuf<unsigned long long> d_pool_cursor;
  // last run
  std::vector<tbc::DevResult> res_host;
  std::vector<uint32_t> witness_host;
  uint64_t timing_ns[4] = {0, 0, 0, 0};
  hipEvent_t ev_turn = nullptr;     // narrow kernel: the search's turn on the device has come (SearchTurn), owned by the batch
  uint64_t turn_wait_ns = 0;        // ... and how long the last run waited for it
  tbc_counters sum{};
  uint64_t device_bytes = 0;
  void count_device_bytes();        // device_bytes = what the arenas hold now (create, and every fresh input that grew one)

  // ---- fresh inputs (batch_stream.hip; include/tbcheck.h "streaming"): the batch's arenas stay, new histories come in as WIRE columns
  // (12 B an op) through pinned host slots the caller fills in place and two device stages, so that the copy of input k + 1 runs
  // under the pass over input k; the run that consumes an input unpacks it into the op columns first (stream_unpack_kernel)
  struct InputSlot { char* mem = nullptr; size_t bytes = 0; hipEvent_t copied = nullptr; bool busy = false; uint32_t* planned = nullptr; };   // planned: a count-form batch's copy of the wire words with the process field re-numbered (pinned; the caller's words stay as written)
  std::vector<InputSlot> in_slots;          // pinned host memory, one block per slot: [op_off][n_events][n_process][word][inv_pos][ret_pos]
  uint32_t in_hist_cap = 0;                 // histories / ops a slot (and a device stage) holds: the first create's, unless tbc_batch_map_input is told more
  uint64_t in_ops_cap = 0;
  tbc::DevBuf<uint32_t> d_stage[2];         // device stages: word[in_ops_cap], inv_pos[in_ops_cap], ret_pos[in_ops_cap]
  hipStream_t stream_copy = nullptr;        // the copies' own stream: they run under whatever `stream` is doing
  hipEvent_t ev_stage[2] = {};              // stage s holds its input completely
  hipEvent_t ev_unpacked[2] = {};           // ... and has been unpacked (the stage may be overwritten)
  bool stage_used[2] = {false, false};
  uint64_t in_seq = 0;                      // inputs submitted so far (input k goes to stage k % 2)
  std::deque<tbc_pending_input> pending;    // submitted, not yet consumed by a run (at most two)
  tbc::DevBuf<uint32_t> d_in_flags;         // what the unpack kernel found that the batch's layout decisions do not allow (bits: kInBad*)
  uint32_t* in_flags_host = nullptr;        // ... read back into pinned memory before the pack is queued
  bool assign_lists = false;                // the resident input came in fresh: its lists' places are dealt on the device (stream_assign_lists_kernel)
  tbc::DevBuf<uint32_t> d_list_over;        // ... and a word that says whether they fit the list arena
  uint64_t in_bytes_copied = 0, in_copy_ns = 0;   // the last consumed input: bytes over PCIe and how long the copy took (events on the copy stream)
  hipEvent_t ev_copy[2][2] = {};            // per stage: copy begins / ends (timing)
  tbc::LayoutTotals cap;                    // the first input's layout totals (what tbc_batch_create allocated for)
  bool lists_headroom = false;              // the list arenas have been given room beyond the first input's exact need
  bool inputs_stale = false;                // the last submitted input was refused: nothing resident to run until the next one
  uint64_t inputs_consumed = 0;
  uint32_t lists_regrown = 0, reload_slot = 0;

  bool borrowed = false;            // stream and events belong to a persistent context (tbc_check)
  ~tbc_batch();
};

namespace tbc {

// ---- batch_create.hip
struct ColumnScan { bool nonneg = true, any_crashed = false, any_crashed_effect = false; int32_t vmax = -1; };
ColumnScan scan_columns(const tbc_ops& c, uint64_t T);
// per-history descriptors (B->hist, B->bh) of `nh` histories and the arenas' running sums; list_caps (null: the lists' places are dealt on
// the device or sized later) = entries of each history's per-front lists
tbc_status layout_histories(tbc_batch* B, uint32_t nh, const uint64_t* op_off, const uint32_t* n_events, const uint32_t* n_slots, const int32_t* aux,
                            const uint32_t* list_caps, bool lists_on_device, std::vector<Hist>& hist, std::vector<BeamHist>& bh, LayoutTotals& tot,
                            const std::vector<CountHist>* count_hist = nullptr);          // (null: the batch's own)
// count form of ONE history (CountHist) from its op columns; slot_col[] = the process column re-numbered.  false: the form does not apply
bool build_count_form(const tbc_ops& c, uint64_t o0, uint64_t n, uint32_t n_process, bool cas_model, int32_t* slot_col, CountHist& out,
                      std::vector<uint32_t>& rets, std::vector<int32_t>& slot_of, std::vector<uint8_t>& used);
// entries of a history's per-front open-call lists, from the op columns (exact when the positions are event indices)
uint64_t open_list_entries(const tbc_ops& c, uint64_t op_off, uint64_t n, uint32_t n_events, uint32_t n_slots, std::vector<uint32_t>& pre, bool branch_lists);
tbc_status batch_create_impl(const tbc_batch_desc* desc, const tbc_model* model, const tbc_opts* opts, tbc_batch* B);

// ---- batch_run.hip
PackArgs make_pack_args(tbc_batch* B);
PackOpenArgs make_pack_open_args(tbc_batch* B);
// phase 0: the whole run.  phase 1 (tbc_batch_sweep_partial): pack + this rank's share of the sweep, stop before the
// verdicts.  phase 2 (tbc_batch_sweep_finish): verdicts from the merged relation table already placed in seg_host.
tbc_status batch_run_impl(tbc_batch* B, tbc_result* results, int phase = 0);

// ---- batch_stream.hip
// the run that is about to start consumes the oldest submitted input, if there is one: wait for its copy, unpack it into the op columns,
// make its descriptors the batch's.  *consumed = whether there was one.
tbc_status stream_consume(tbc_batch* B, bool* consumed);
// after the pack of a fresh input has counted the lists: deal their places (launch on the batch's stream)
tbc_status stream_assign_lists(tbc_batch* B);
// after the run: did the lists fit?  (grows the list arenas for the next input when they did not)
tbc_status stream_after_run(tbc_batch* B, const HostBuf<BeamHist>& bh_back);
void stream_release(tbc_batch* B);

}  // namespace tbc
