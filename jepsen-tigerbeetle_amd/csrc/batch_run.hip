// batch_run.hip -- tbc_batch_run: one pass of the hot path over a resident batch, as phases over a RunState (round 6: this was ONE
// function of 560 lines carrying every engine and every fallback):
//   first_pass            (a fresh input unpacked,) memsets, pack, the search / the sweep launched, the relaxed sweep beside a count-form
//                         search, everything read back
//   sweep_verdicts        level sweep: second pass over overflowed segments, composition, :configs dump, what it hands to the search
//   list_overflow_fallback  wide-schedule histories whose open-call lists did not fit: the sequential kernel
//   overflow_retries      visited sets that filled up: 16x larger ones in scratch arenas
//   stall_handover        (opt-in) histories that stopped passing completions: the level sweep
//   count_form_pipeline   what the budgeted exact search left undecided: relaxed refutation, prefix search
//   finish                timings, counters, verdicts marshalled into tbc_result
// A new engine is a new phase (or a new launcher behind first_pass), not another branch through all of them.
#include "tbc_batch.h"
#include "witness_expand.h"

using namespace tbc;

namespace tbc {

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
  a.progress = B->progress; a.progress_dev = B->d_progress.p;
  return a;
}

PackArgs make_pack_args(tbc_batch* B) {
  PackArgs pa{};
  pa.hist = B->d_hist.p; pa.f = B->d_f.p; pa.a = B->d_a.p; pa.b = B->d_b.p; pa.process = B->d_proc.p;
  pa.inv_pos = B->d_inv.p; pa.ret_pos = B->d_ret.p; pa.rec = B->d_rec.p; pa.seg = B->d_seg.p;
  pa.ret_slot = B->d_ret_slot.p; pa.ret_op = B->d_ret_op.p; pa.bitmap = B->d_bitmap.p; pa.wpre = B->d_wpre.p;
  pa.scratch = B->d_frames.p; pa.frame_words = B->frame_words; pa.n_hist = B->n_hist;
  pa.model_kind = B->model.kind; pa.n_classes = B->model.n_classes; pa.dbg = debug_words();
  pa.pool_vals = B->d_pool_vals.p; pa.pool_len = B->pool_len; pa.n_keys = B->model.n_keys;
  return pa;
}

PackOpenArgs make_pack_open_args(tbc_batch* B) {
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
  po.order_of = (!B->hist_order.empty() && B->d_order.p) ? B->d_order.p : nullptr;
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
  a.abort = B->ext_abort; a.abort_set = B->ext_abort_set; a.abort_map = B->ext_abort_map;
  a.park = B->d_park.p; a.resume = 0;
  a.progress = B->progress; a.progress_dev = B->d_progress.p;
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

// The calls open at the failing completion of history h (invocation order) and the three columns that say so, copied back on demand
// (invalid verdicts are rare).  P = history position of the failing completion.
struct PendingAt {
  std::vector<int32_t> proc;
  std::vector<uint32_t> inv, ret, pending;
};
static tbc_status pending_at(tbc_batch* B, uint32_t h, uint32_t fail_op, PendingAt& out) {
  const Hist& H = B->hist[h];
  const uint32_t n = H.n_ops;
  out.proc.resize(n); out.inv.resize(n); out.ret.resize(n); out.pending.clear();
  HIP_TRY(hipMemcpy(out.proc.data(), B->d_proc.p + H.op_off, (size_t)n * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(out.inv.data(), B->d_inv.p + H.op_off, (size_t)n * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(out.ret.data(), B->d_ret.p + H.op_off, (size_t)n * 4, hipMemcpyDeviceToHost));
  if (fail_op >= n) return TBC_OK;
  const uint32_t P = out.ret[fail_op];               // history position of the failing completion
  for (uint32_t i = 0; i < n && out.inv[i] < P; i++)
    if (out.ret[i] == TBC_POS_CRASHED || out.ret[i] >= P) out.pending.push_back(i);
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
  PendingAt pa;
  tbc_status ps = pending_at(B, h, d.fail_op, pa);
  if (ps != TBC_OK) return ps;
  const std::vector<int32_t>& proc = pa.proc;
  const std::vector<uint32_t>& ret = pa.ret;
  const std::vector<uint32_t>& pending = pa.pending;
  std::vector<uint32_t> order(got);
  for (uint32_t i = 0; i < got; i++) order[i] = i;
  std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
    const uint64_t* a = &rec[(size_t)x * RW]; const uint64_t* b = &rec[(size_t)y * RW];
    const int32_t sa = (int32_t)(a[0] >> 32), sb = (int32_t)(b[0] >> 32);
    if (sa != sb) return sa < sb;
    for (uint32_t j = 0; j < MW; j++) if (a[1 + j] != b[1 + j]) return a[1 + j] < b[1 + j];
    return false;
  });
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
  const uint32_t order = h < B->order_of_hist.size() ? B->order_of_hist[h] : B->list_order();      // (an order restart's pass answers in its own order)
  if (B->rules & kRuleTxnEager) {          // multi-register: the calls the rule absorbed are pure-read txns; the micro-ops are in the value pool
    std::vector<int32_t> pool(B->pool_len);
    if (B->pool_len) HIP_TRY(hipMemcpy(pool.data(), B->d_pool_vals.p, (size_t)B->pool_len * 4, hipMemcpyDeviceToHost));
    if (!expand_eager_txn_chain(n, f.data(), a.data(), b.data(), proc.data(), inv.data(), ret.data(), H.n_slots, B->model.init, order != 0u,
                                pool.data(), wit, *len, out)) {
      set_error("history %u: malformed witness chain", h);
      return TBC_ERR_HIP;
    }
    std::copy(out.begin(), out.end(), wit);
    *len = (uint32_t)out.size();
    return TBC_OK;
  }
  if (!expand_eager_chain(n, f.data(), a.data(), b.data(), proc.data(), inv.data(), ret.data(), H.n_slots, B->model.init,
                          (B->rules & kRuleBranch) != 0u, order != 0u, wit, *len, out)) {
    set_error("history %u: malformed witness chain", h);
    return TBC_ERR_HIP;
  }
  std::copy(out.begin(), out.end(), wit);
  *len = (uint32_t)out.size();
  return TBC_OK;
}

// ---- several batches in flight on one device (each on its own stream, from its own host thread).  The narrow kernel is sized
// to the whole GPU and lives on latency, the pack kernels on vector issue: a batch's pack beside ANOTHER batch's search uses
// what the search leaves idle.  Two searches launched side by side only halve each other -- but a search queued BEHIND a running one
// starts in its tail (below), so by default nothing orders them; SearchTurn can chain the searches of one device through an event
// (a search then starts when the one launched before it, on whatever stream, is done), which is how rounds 3 - 5 ran.
namespace {
struct SearchTurn {
  static std::mutex& mu() { static std::mutex m; return m; }
  static hipEvent_t& last(int dev) { static hipEvent_t ev[64] = {}; return ev[dev & 63]; }
  std::lock_guard<std::mutex> g;
  int dev; hipStream_t s;
  // (round 6: NOT chained by default.  The launch is sized to what the GPU keeps resident, so a second search queued behind a running one
  // gets wavefront slots only as the first one's searches end -- it fills the first one's tail, whose longest history is 1.7 x the mean
  // (profiles/NOTES_r06.md): 366k -> 380k histories/s with two batches in flight, 359k -> 375k with three.  TBC_SEARCH_TURN=1 chains them
  // again -- a search then starts when the one launched before it is done --, which is what rounds 3 - 5 measured with.)
  static bool chained() { static const bool on = std::getenv("TBC_SEARCH_TURN") && std::getenv("TBC_SEARCH_TURN")[0] == '1'; return on; }
  SearchTurn(int device, hipStream_t stream) : g(mu()), dev(device), s(stream) {
    if (chained() && last(dev)) (void)hipStreamWaitEvent(s, last(dev), 0);
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

// The op columns of the histories in `list`, as they lie in HBM, gathered into ONE pinned host block (six columns back to back, the
// histories in list order): a copy into pinned memory is a DMA that is merely queued, one into a std::vector is staged and waited
// for -- ~1.5 ms each, 618 of them were 0.93 s of a 1.2 s pass (round 6, the race of list orders on bench workload_3).
struct FetchedColumns {
  char* pin = nullptr;
  uint64_t T = 0;
  std::vector<uint64_t> off;          // list.size() + 1
  uint8_t* f = nullptr; int32_t* a = nullptr; int32_t* b = nullptr; int32_t* proc = nullptr; uint32_t* inv = nullptr; uint32_t* ret = nullptr;
  ~FetchedColumns() { if (pin) (void)hipHostFree(pin); }
};
static tbc_status fetch_columns(tbc_batch* B, const uint32_t* list, size_t n_list, FetchedColumns& out) {
  out.off.assign(n_list + 1, 0);
  for (size_t i = 0; i < n_list; i++) out.off[i + 1] = out.off[i] + B->hist[list[i]].n_ops;
  const uint64_t T = out.T = out.off[n_list], T4 = (T + 3) & ~3ull;
  HIP_TRY(hipHostMalloc((void**)&out.pin, (size_t)(T4 * 21 + 64), hipHostMallocDefault));
  out.a = (int32_t*)out.pin; out.b = out.a + T4; out.proc = out.b + T4; out.inv = (uint32_t*)(out.proc + T4); out.ret = out.inv + T4; out.f = (uint8_t*)(out.ret + T4);
  for (size_t i = 0; i < n_list; i++) {
    const Hist& H = B->hist[list[i]];
    const uint64_t n = H.n_ops, o = out.off[i], s0 = H.op_off;
    if (!n) continue;
    HIP_TRY(hipMemcpyAsync(out.f + o, B->d_f.p + s0, n, hipMemcpyDeviceToHost, B->stream));
    HIP_TRY(hipMemcpyAsync(out.a + o, B->d_a.p + s0, n * 4, hipMemcpyDeviceToHost, B->stream));
    HIP_TRY(hipMemcpyAsync(out.b + o, B->d_b.p + s0, n * 4, hipMemcpyDeviceToHost, B->stream));
    HIP_TRY(hipMemcpyAsync(out.proc + o, B->d_proc.p + s0, n * 4, hipMemcpyDeviceToHost, B->stream));
    HIP_TRY(hipMemcpyAsync(out.inv + o, B->d_inv.p + s0, n * 4, hipMemcpyDeviceToHost, B->stream));
    HIP_TRY(hipMemcpyAsync(out.ret + o, B->d_ret.p + s0, n * 4, hipMemcpyDeviceToHost, B->stream));
  }
  HIP_TRY(hipStreamSynchronize(B->stream));
  return TBC_OK;
}

static tbc_status hand_over_stalled(tbc_batch* B, const std::vector<uint32_t>& list, std::vector<tbc_result>& out) {
  CtxSuspend own_arenas;                         // (the inner batch owns its arenas, stream and events; the caller's context comes back however this returns)
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
    FetchedColumns fc;
    if ((st = fetch_columns(B, list.data() + lo, k, fc)) != TBC_OK) break;
    const uint64_t T = fc.T;
    uint8_t* const f = fc.f; int32_t* const a = fc.a; int32_t* const b = fc.b; int32_t* const pr = fc.proc; uint32_t* const inv = fc.inv; uint32_t* const ret = fc.ret;
    tbc_batch_desc d{};
    d.n_hist = k; d.op_off = off.data(); d.n_events = nev.data(); d.n_process = npr.data(); d.model_aux = aux.data();
    d.cols.n = (uint32_t)T; d.cols.f = f; d.cols.a = a; d.cols.b = b; d.cols.process = pr;
    d.cols.inv_pos = inv; d.cols.ret_pos = ret;
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
  return st;
}

namespace {

// what the phases of one run hand on to each other
struct RunState {
  tbc_batch* const B;
  tbc_result* const results;
  const int phase;
  const uint64_t t_start;
  const uint32_t nh;
  hipStream_t const s;
  const bool beam;
  const uint32_t KW, EW;
  HostBuf<Hist>& hist_back;
  HostBuf<BeamHist>& bh_back;
  // count form: the exact search runs under a budget of probes; what it does not finish goes through the relaxed refutation and
  // the prefix search (a caller who names max_steps gets one exact pass under that limit instead)
  const uint64_t count_budget;
  // order restarts (tbcheck.h, TBC_DOM_NO_ORDER_RESTARTS): the first pass of a wide depth-first batch runs under the same kind of budget;
  // what it leaves undecided is searched again in the next list order (order_restarts)
  const uint64_t restart_budget;
  uint64_t first_budget() const { return count_budget ? count_budget : restart_budget; }
  SweepArgs swa{};
  // a big quiet batch (several histories per wavefront), IF ASKED (tbc_opts.dominance, TBC_DOM_STALL_HANDOVER: off by default, tbcheck.h says why):
  // a history that stops passing completions is handed to the level sweep (hand_over_stalled) -- where that can answer: register / cas-register, one mask word, nobody asking for a witness or naming a step limit
  const bool stall_on;
  std::vector<tbc_result> handed;
  std::vector<uint8_t> was_handed;
  // the relaxed sweep's verdicts: the completion rank at which history h is refuted (kInf: not refuted -- or not swept at all)
  std::vector<uint32_t> rs_level;
  std::vector<uint8_t> rs_valid;          // ... and the histories it could not refute (VALID under the relaxation: the exact search just needs its time)
  const uint64_t max_bytes;
  std::vector<uint32_t> final_log2;
  std::vector<uint8_t> is_seq;            // which kernel owns the history's result
  std::vector<uint8_t> by_sweep;          // answered by the level sweep (analyzer :linear)
  std::vector<uint32_t> width_of;
  std::vector<uint8_t> scratched;         // the history's last search ran in a scratch arena (freed since: its parked state points nowhere)
  bool touched_work = false;
  static constexpr uint64_t arena_budget = 32ull << 30;

  RunState(tbc_batch* b, tbc_result* res, int ph)
      : B(b), results(res), phase(ph), t_start(now_ns()), nh(b->n_hist), s(b->stream), beam(b->width > 1), KW(1 + b->mask_words), EW(b->entry_words()),
        hist_back(b->hist_back_m), bh_back(b->bh_back_m),
        count_budget((b->count_form && b->opts.max_steps == 0) ? 32ull * b->max_ops : 0ull),
        restart_budget((ph == 0 && !b->sweep && b->order_restarts_apply()) ? 32ull * b->max_ops : 0ull),
        stall_on(b->lanes != 0 && (b->opts.dominance & TBC_DOM_STALL_HANDOVER) != 0 && !b->count_form && b->mask_words == 1 && !b->opts.want_witness && b->opts.max_steps == 0 && ph == 0 &&
                 (b->model.kind == TBC_MODEL_REGISTER || b->model.kind == TBC_MODEL_CAS_REGISTER) && b->vpad != 0),
        was_handed(b->n_hist, 0), rs_level(b->n_hist, kInf), rs_valid(b->n_hist, 0),
        max_bytes(b->opts.max_visited_bytes ? b->opts.max_visited_bytes : (1ull << 30)) {
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
  }
};

// ---- pack: K1 (+ the per-front counts) and K1b, in whichever form takes this batch
tbc_status queue_pack(RunState& R) {
  tbc_batch* B = R.B; hipStream_t s = R.s; const uint32_t nh = R.nh; const bool beam = R.beam;
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
    if (B->assign_lists) {
      // a fresh input (batch_stream.hip): nobody has told the histories where their lists lie -- the counts are on the device now,
      // their places are dealt there (no trip to the host between the pack and the walk)
      if (!counted) { launch_open_counts(po, s); HIP_TRY(hipGetLastError()); counted = true; }
      tbc_status as = stream_assign_lists(B);
      if (as != TBC_OK) return as;
    }
    launch_pack_open(po, s, counted);
    HIP_TRY(hipGetLastError());
  }
  return TBC_OK;
}

// memsets, pack, the search (or the sweep) launched, the relaxed sweep beside a count-form search, everything read back
tbc_status first_pass(RunState& R) {
  tbc_batch* B = R.B; hipStream_t s = R.s; const uint32_t nh = R.nh; const bool beam = R.beam; const int phase = R.phase;
  HostBuf<Hist>& hist_back = R.hist_back; HostBuf<BeamHist>& bh_back = R.bh_back;
  TRACE("run: begin");
  HIP_TRY(hipEventRecord(B->ev[0], s));
  if (B->d_progress.p) HIP_TRY(hipMemsetAsync(B->d_progress.p, 0, sizeof(uint32_t), s));
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
  const bool use_epoch = beam && B->lanes != 0;          // (a batch with lanes has every history below kNarrowMaxOps: batch_create.hip)
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

  if (!B->hist_order.empty()) {          // (a race's batch: every history's own list order goes up before the walk)
    if (B->d_order.n < nh) { B->d_order.release(); const tbc_status os = B->d_order.alloc(nh); if (os != TBC_OK) return os; }
    HIP_TRY(hipMemcpyAsync(B->d_order.p, B->hist_order.data(), (size_t)nh * 4, hipMemcpyHostToDevice, s));
  }
  tbc_status ps = queue_pack(R);
  if (ps != TBC_OK) return ps;
  TRACE("run: pack launched");
  SYNC_TRACE("pack");
  HIP_TRY(hipEventRecord(B->ev[2], s));

  // ---- the RELAXED sweep (see tbc_batch::rsweep) on a stream of its own, BESIDE the exact search: INVALID at completion t = invalid, first bad
  // completion at t or earlier -- the history's exact search is told to stop (BeamArgs.abort) and the prefix search below pins the completion;
  // VALID (or a burst that outgrows the sets) proves nothing and the exact search runs on as it always did, the sweep's 17 ms hidden behind it
  bool rs_on = B->rsweep && phase == 0 && beam && !B->lanes;
  SweepArgs ra = R.swa;
  if (rs_on) {
    ra.ncr = B->d_zncr.p; ra.crashed = nullptr; ra.reach = B->d_reach.p; ra.reach_hdr = B->d_reach_hdr.p;
    ra.rules = B->rules & (kRuleEager | kRuleTwin);
    HIP_TRY(hipMemsetAsync(B->d_abort.p, 0, B->d_abort.bytes(), s));
    HIP_TRY(hipEventRecord(B->ev2[0], s));                      // the pack's tables are complete, the abort words zero
    HIP_TRY(hipStreamWaitEvent(B->stream2, B->ev2[0], 0));
    hist_back.resize(nh);
    // (a sweep that could not be launched -- more workgroups than the workgroup kernel takes, an attribute the runtime refused -- has
    // left nothing to read back: round 5 composed whatever the buffers held then; now there simply is no relaxed sweep this run)
    if (launch_sweep(ra, B->stream2)) {
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipMemcpyAsync(B->seg_host.data(), B->d_sres.p, B->seg_host.size() * sizeof(SegResult), hipMemcpyDeviceToHost, B->stream2));
      HIP_TRY(hipMemcpyAsync(hist_back.data(), B->d_hist.p, nh * sizeof(Hist), hipMemcpyDeviceToHost, B->stream2));
    } else rs_on = false;
  }

  if (B->sweep) {
    SweepArgs mine = R.swa;
    if (phase == 1 && B->shard_world > 1) {     // another rank's records must read as zero in the exchanged table
      mine.shard_rank = B->shard_rank; mine.shard_world = B->shard_world;
      HIP_TRY(hipMemsetAsync(B->d_sres.p, 0, B->d_sres.bytes(), s));
    }
    if (!launch_sweep(mine, s)) { set_error("level sweep launch failed"); return TBC_ERR_HIP; }
  } else if (beam) {
    BeamArgs ba = make_beam_args(B, B->d_btab.p, B->d_stack.p, B->d_dstack.p, nh);
    if (R.first_budget()) ba.max_steps = R.first_budget();
    if (rs_on) ba.abort = B->d_abort.p;
    if (B->lanes) { ba.tab_stride = B->tab_stride(); ba.epoch = use_epoch ? B->epoch : 0u; }
    if (R.stall_on) ba.stall_checks = kStallChecks;
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
    for (uint32_t h = 0; h < nh; h++) {
      if (hist_back[h].status != 0 || hist_back[h].n_ret == 0) continue;
      tbc_sweep_verdict v{};
      (void)tbc_sweep_compose(&B->seg_host[(size_t)h * B->max_segs * SL], B->max_segs, hist_back[h].n_ret, &v);
      if (v.valid == TBC_INVALID) {
        R.rs_level[h] = v.fail_level;
        HIP_TRY(hipMemcpyAsync(B->d_abort.p + h, B->abort_one, 4, hipMemcpyHostToDevice, B->stream2));      // its exact search may stop
      }
      R.rs_valid[h] = v.valid == TBC_VALID;
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
  // (the abort words' copies are on the other stream: nothing of this run may still be on its way when the next run zeroes them -- or
  // when the batch is destroyed; a late copy would stop the next run's exact search for nothing: ADVICE.md)
  if (rs_on) HIP_TRY(hipStreamSynchronize(B->stream2));
  TRACE("run: first pass synced");
  for (uint32_t h = 0; h < nh; h++) if (R.rs_level[h] != kInf) {      // refuted by the relaxed sweep: whatever its exact search got to before it was told to stop is dropped (the passes below, and their counters, are then the same run after run)
    DevResult& d = B->res_host[h];
    std::memset(&d, 0, sizeof d);
    d.valid = TBC_UNKNOWN; d.cause = TBC_CAUSE_STEP_LIMIT; d.fail_op = TBC_NO_OP; d.prev_ok_op = TBC_NO_OP; d.final_state = B->model.init;
    d.tab_log2 = B->bh[h].tab_log2;
  }
  return TBC_OK;
}

// level sweep: second pass over the segments that overflowed, composition of the relations, :configs of an invalid verdict, and
// what the sweep could not finish handed to the wide kernel
tbc_status sweep_verdicts(RunState& R) {
  tbc_batch* B = R.B; hipStream_t s = R.s; const uint32_t nh = R.nh;
  HostBuf<Hist>& hist_back = R.hist_back; HostBuf<BeamHist>& bh_back = R.bh_back;
  const SweepArgs& swa = R.swa;
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
    if (hist_back[h].n_ret == 0) { d.valid = TBC_VALID; R.by_sweep[h] = 1; continue; }
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
    R.by_sweep[h] = 1;
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
    for (uint32_t h : fb) R.final_log2[h] = B->bh[h].tab_log2;
    tbc_status st = scratch_pass(B, fb, lg, true, hist_back, bh_back);
    if (st != TBC_OK) return st;
    R.touched_work = true;
  }
  return TBC_OK;
}

// wide-schedule histories whose open-call lists did not fit: sequential kernel
tbc_status list_overflow_fallback(RunState& R) {
  tbc_batch* B = R.B; const uint32_t nh = R.nh;
  std::vector<uint32_t> fb, lg;
  for (uint32_t h = 0; h < nh; h++)
    if (R.hist_back[h].status == 0 && R.bh_back[h].status != 0) {
      if (B->model.kind == TBC_MODEL_SET || B->model.kind == TBC_MODEL_BANK) {
        set_error("history %u: open-call lists exceed their arena and set / bank have no sequential kernel", h);
        return TBC_ERR_UNSUPPORTED;
      }
      uint32_t l = B->hist[h].tab_log2;
      fb.push_back(h); lg.push_back(l); R.final_log2[h] = l; R.is_seq[h] = 1;
    }
  if (!fb.empty()) {
    // the sequential kernel reads Hist.status only; clear the wide-schedule flag for it
    tbc_status st = scratch_pass(B, fb, lg, false, R.hist_back, R.bh_back);
    if (st != TBC_OK) return st;
    R.touched_work = true;
  }
  return TBC_OK;
}

// overflow retries: 16x larger visited set each time, up to max_visited_bytes
tbc_status overflow_retries(RunState& R) {
  tbc_batch* B = R.B; const uint32_t nh = R.nh; const uint32_t KW = R.KW, EW = R.EW;
  for (;;) {
    std::vector<uint32_t> pend_seq, lg_seq, pend_beam, lg_beam;
    for (uint32_t h = 0; h < nh; h++) {
      if (B->res_host[h].valid == TBC_UNKNOWN && B->res_host[h].cause == TBC_CAUSE_VISITED_FULL) {
        const uint64_t wpe = R.is_seq[h] ? KW : EW;
        if (!R.is_seq[h]) R.final_log2[h] = std::max(R.final_log2[h], B->res_host[h].tab_log2);   // grown inside the kernel already
        uint32_t lg = R.final_log2[h] + 4;
        while (lg > R.final_log2[h] && ((1ull << lg) * wpe * 8 > R.max_bytes || (!R.is_seq[h] && lg > kBeamMaxTabLog2))) lg--;
        if (lg > R.final_log2[h]) {
          if (R.is_seq[h]) { pend_seq.push_back(h); lg_seq.push_back(lg); }
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
          if (!grp.empty() && (bytes + need > R.arena_budget || (pass == 1 && R.width_of[pend[pos]] != R.width_of[grp[0]]))) break;
          grp.push_back(pend[pos]); glg.push_back(lgs[pos]); R.final_log2[pend[pos]] = lgs[pos];
          R.scratched[pend[pos]] = 1;
          bytes += need; pos++;
        }
        tbc_status st = scratch_pass(B, grp, glg, pass == 1, R.hist_back, R.bh_back, pass == 1 ? R.width_of[grp[0]] : 0, kCountExact, nullptr,
                                     (pass == 1 && R.first_budget()) ? (int64_t)R.first_budget() : -1);
        if (st != TBC_OK) return st;
        R.touched_work = true;
      }
    }
  }
  return TBC_OK;
}

tbc_status stall_handover(RunState& R) {
  tbc_batch* B = R.B; const uint32_t nh = R.nh;
  std::vector<uint32_t> stalled;
  for (uint32_t h = 0; h < nh; h++)
    if (!R.is_seq[h] && R.hist_back[h].status == 0 && B->res_host[h].valid == TBC_UNKNOWN && B->res_host[h].cause == TBC_CAUSE_STEP_LIMIT) stalled.push_back(h);
  if (!stalled.empty()) {
    R.handed.resize(nh);
    tbc_status st = hand_over_stalled(B, stalled, R.handed);
    if (st != TBC_OK) return st;
    for (uint32_t h : stalled) R.was_handed[h] = 1;
    HIP_TRY(hipSetDevice(B->device));
  }
  return TBC_OK;
}

// One budgeted pass of the wide / narrow depth-first search over `grp` in a scratch arena (count_mode, per-history prefix targets, `steps`
// probes at most, 0 = no limit); a history whose visited set fills up is taken again with a 16x larger one.
tbc_status budgeted_pass(RunState& R, const std::vector<uint32_t>& grp, uint32_t mode, const std::vector<uint32_t>* targets, int64_t steps) {
  tbc_batch* B = R.B; const uint32_t EW = R.EW;
  std::vector<uint32_t> todo = grp, tg, lgs;
  if (targets) tg = *targets;
  for (uint32_t h : todo) {
    uint32_t lg = std::max(R.final_log2[h], ceil_log2(64ull * std::max<uint64_t>(B->hist[h].n_ops, 1)));
    while (lg > 10 && ((1ull << lg) * EW * 8 > R.max_bytes || lg > kBeamMaxTabLog2)) lg--;
    lgs.push_back(lg);
  }
  while (!todo.empty()) {
    size_t pos = 0;
    while (pos < todo.size()) {
      std::vector<uint32_t> g, glg, gtg;
      uint64_t bytes = 0;
      while (pos < todo.size()) {
        const uint64_t need = (1ull << lgs[pos]) * ((uint64_t)EW + 2) * 8;
        if (!g.empty() && bytes + need > R.arena_budget) break;
        g.push_back(todo[pos]); glg.push_back(lgs[pos]); if (targets) gtg.push_back(tg[pos]);
        R.final_log2[todo[pos]] = lgs[pos]; bytes += need; pos++;
      }
      tbc_status st = scratch_pass(B, g, glg, true, R.hist_back, R.bh_back, 0, mode, targets ? &gtg : nullptr, steps);
      if (st != TBC_OK) return st;
    }
    std::vector<uint32_t> again, alg, atg;
    for (size_t i = 0; i < todo.size(); i++) {
      const DevResult& d = B->res_host[todo[i]];
      if (d.valid != TBC_UNKNOWN || d.cause != TBC_CAUSE_VISITED_FULL) continue;
      uint32_t lg = lgs[i] + 4;
      while (lg > lgs[i] && ((1ull << lg) * EW * 8 > R.max_bytes || lg > kBeamMaxTabLog2)) lg--;
      if (lg > lgs[i]) { again.push_back(todo[i]); alg.push_back(lg); if (targets) atg.push_back(tg[i]); }
    }
    todo.swap(again); lgs.swap(alg); tg.swap(atg);
  }
  return TBC_OK;
}

// ---- ORDER RESTARTS (tbcheck.h, TBC_DOM_NO_ORDER_RESTARTS; oracle/wgl.py check_restart_pipeline states the same passes).  A pass over a
// batch is as long as its slowest history, and at high concurrency the slowest history is slow because of the ORDER its candidates are
// tried in -- another order ends it in a fraction of the probes (the costs are heavy-tailed and nearly independent between orders).  What
// the budgeted first pass left undecided is searched again from scratch, order by order: the fronts' lists are walked again in the
// pass's order (the walk with lane = front has no state to reset: streaming writes over the same places), the undecided histories run in
// a scratch arena under the same budget; whoever no order ended runs in the default order without one.  The counters of a history are
// the sums over its passes.
static const uint32_t kRestartOrders[] = {16u + 48u, 2u, 1u, 16u + 8u, 0u};      // PackOpenArgs.list_order numbers (oracle/wgl.py RESTART_ORDERS)

// ---- THE RACE (the form the library takes when nobody wants a witness).  One order after the other costs a budget per order before the
// lucky one is reached, and a history no order is lucky for pays all of them and then its whole search (measured, bench workload_3:
// 7.5k -> 2.9k histories/s).  Run AT THE SAME TIME the orders cost what the luckiest costs -- and the default order need not start again:
// its search is RESUMED where the budget stopped it (BeamArgs.park / resume) while one small batch holds a replica of every undecided
// history per other order (PackOpenArgs.order_of: each replica's lists walked in its own order), on a stream of its own.  All searches of
// a history share one word: the first to decide it sets the word (BeamArgs.abort_set), the others stop at their next look at the clock
// (BeamArgs.abort).  Six wavefronts a straggler, which is what a GPU whose pass is waiting for its slowest history has idle; a history
// that is simply hard in every order costs what it cost before (its default-order search never stopped), plus the replicas' wavefronts.
// Verdict and failing op are the search's in any order; WHICH order answers (and so the counters) can differ from run to run -- as
// knossos.competition's :analyzer does.
static const uint32_t kRaceOrders[] = {16u + 48u, 2u, 1u, 16u + 8u, 0u};      // PackOpenArgs.list_order numbers: the orders that race the default one
tbc_status race_orders(RunState& R, const std::vector<uint32_t>& pend_all) {
  tbc_batch* B = R.B; const uint32_t nh = R.nh; hipStream_t s = R.s;
  constexpr uint32_t K = sizeof(kRaceOrders) / sizeof(kRaceOrders[0]);
  const uint32_t default_order = B->list_order();
  R.handed.resize(nh);
  B->last_raced = (uint32_t)pend_all.size();
  CtxSuspend own_arenas;                         // (the inner batch owns its arenas, stream and events)
  // groups of histories whose replicas fit three quarters of the device memory that is free now (a replica: a first visited set of
  // 64 entries an op -- a straggler's search needs 10^5 configs and more: a smaller set grows or, worse, fills up and starts again --
  // with its stacks and growth pool, and the per-front tables: ~4 KB an op at 32 calls in flight)
  size_t mem_free = 0, mem_total = 0;
  if (hipMemGetInfo(&mem_free, &mem_total) != hipSuccess) mem_free = 32ull << 30;
  const uint64_t mem_cap = std::max<uint64_t>(8ull << 30, (uint64_t)mem_free / 4 * 3);
  std::vector<DevResult> outer_res(nh);
  size_t lo = 0;
  while (lo < pend_all.size()) {
    uint64_t ops = 0;
    size_t hi = lo;
    while (hi < pend_all.size() && (hi == lo || (ops + B->hist[pend_all[hi]].n_ops) * 4000ull * (K + 1) <= mem_cap)) { ops += B->hist[pend_all[hi]].n_ops; hi++; }
    const uint32_t k = (uint32_t)(hi - lo);
    // ---- the default order goes on where its budgeted pass stopped (BeamArgs.resume: stacks, visited set, counters as they were left),
    // in the batch's own arenas, on the batch's own stream -- launched first, so that nothing of what follows delays it.  (A history
    // whose pass ended in a scratch arena -- its visited set had filled up -- has nothing left to go on from: the default order is one
    // of its replicas below.)
    std::vector<uint32_t> res_list, outer_map(nh, 0u);
    for (uint32_t i = 0; i < k; i++) {
      const uint32_t h = pend_all[lo + i];
      outer_map[h] = i;
      if (!R.scratched[h] && B->d_park.p) res_list.push_back(h);
    }
    DevBuf<uint32_t> decided, d_outer_map, d_inner_map;
    tbc_status st;
    if ((st = decided.alloc(k)) || (st = d_outer_map.alloc(nh))) return st;
    HIP_TRY(hipMemsetAsync(decided.p, 0, (size_t)k * 4, s));
    HIP_TRY(hipMemcpyAsync(d_outer_map.p, outer_map.data(), (size_t)nh * 4, hipMemcpyHostToDevice, s));
    // the replicas' columns come from HBM first (the batch's stream is still idle)
    std::vector<uint32_t> rep_hist, rep_order;          // replica -> (index in the group, its list order)
    for (uint32_t i = 0; i < k; i++) {
      for (uint32_t o : kRaceOrders) if (o != default_order) { rep_hist.push_back(i); rep_order.push_back(o); }
      if (R.scratched[pend_all[lo + i]] || !B->d_park.p) { rep_hist.push_back(i); rep_order.push_back(default_order); }
    }
    const uint32_t nr = (uint32_t)rep_hist.size();
    FetchedColumns fc;
    if ((st = fetch_columns(B, pend_all.data() + lo, k, fc)) != TBC_OK) return st;
    const std::vector<uint64_t>& goff = fc.off;
    const uint8_t* const gf = fc.f; const int32_t* const ga = fc.a; const int32_t* const gb = fc.b; const int32_t* const gp = fc.proc;
    const uint32_t* const gi = fc.inv; const uint32_t* const gr = fc.ret;
    const uint32_t n_res = (uint32_t)res_list.size();
    if (n_res) {
      HIP_TRY(hipMemcpyAsync(B->d_work.p, res_list.data(), (size_t)n_res * 4, hipMemcpyHostToDevice, s));
      BeamArgs ba = make_beam_args(B, B->d_btab.p, B->d_stack.p, B->d_dstack.p, n_res);
      ba.resume = 1; ba.max_steps = 0; ba.abort = decided.p; ba.abort_set = decided.p; ba.abort_map = d_outer_map.p;
      if (!launch_beam(ba, B->mask_words, search_blocks(n_res), s)) { set_error("unsupported mask width"); return TBC_ERR_UNSUPPORTED; }
      HIP_TRY(hipGetLastError());
      // (its results are read back AFTER the replicas have run: a copy into pageable memory queued here would hold this thread until the
      // resumed searches end -- the first version did exactly that, and the race began when it was over: 21.6 s of a 24 s pass)
      R.touched_work = true;
    }
    // ---- the other orders: ONE batch of replicas (a history once per order, each replica walked in its own order: PackOpenArgs.order_of),
    // on a stream of its own beside the resumed default order; every replica of history i and the resumed search share word i
    std::vector<uint64_t> off(nr + 1, 0);
    std::vector<uint32_t> nev(nr), npr(nr), imap(nr);
    for (uint32_t r = 0; r < nr; r++) {
      const Hist& H = B->hist[pend_all[lo + rep_hist[r]]];
      off[r + 1] = off[r] + H.n_ops; nev[r] = H.n_events; npr[r] = H.n_slots; imap[r] = rep_hist[r];
    }
    const uint64_t T = off[nr];
    if (T > 0xFFFFFFFFull) { set_error("the race's replicas exceed 2^32 ops"); return TBC_ERR_OOM; }
    std::vector<uint8_t> f(T + 1);
    std::vector<int32_t> a(T + 1), b(T + 1), pr(T + 1);
    std::vector<uint32_t> inv(T + 1), ret(T + 1);
    for (uint32_t r = 0; r < nr; r++) {
      const uint64_t n = off[r + 1] - off[r], o = off[r], g0 = goff[rep_hist[r]];
      std::memcpy(f.data() + o, gf + g0, n);
      std::memcpy(a.data() + o, ga + g0, n * 4); std::memcpy(b.data() + o, gb + g0, n * 4); std::memcpy(pr.data() + o, gp + g0, n * 4);
      std::memcpy(inv.data() + o, gi + g0, n * 4); std::memcpy(ret.data() + o, gr + g0, n * 4);
    }
    tbc_batch_desc d{};
    d.n_hist = nr; d.op_off = off.data(); d.n_events = nev.data(); d.n_process = npr.data();
    d.cols.n = (uint32_t)T; d.cols.f = f.data(); d.cols.a = a.data(); d.cols.b = b.data(); d.cols.process = pr.data();
    d.cols.inv_pos = inv.data(); d.cols.ret_pos = ret.data();
    tbc_opts o = B->opts;
    o.algorithm = TBC_ALG_COMPETITION; o.want_witness = 0; o.search_width = B->width; o.lanes_per_history = 64; o.round_budget = 0; o.max_steps = 0;
    o.visited_per_op = 64; o.list_order = TBC_ORDER_SLOT; o.dominance = (B->opts.dominance | TBC_DOM_NO_ORDER_RESTARTS) & ~TBC_DOM_STALL_HANDOVER;
    std::vector<tbc_result> res(nr);
    tbc_batch* inner = nullptr;
    TRACE("race: the replicas' batch is being made");
    st = nr ? tbc_batch_create(&d, &B->model, &o, &inner) : TBC_OK;
    if (st == TBC_OK && nr) {
      st = d_inner_map.alloc(nr);
      if (st == TBC_OK && hipMemcpy(d_inner_map.p, imap.data(), (size_t)nr * 4, hipMemcpyHostToDevice) != hipSuccess) st = TBC_ERR_HIP;
      if (st == TBC_OK) {
        inner->ext_abort = decided.p; inner->ext_abort_set = decided.p; inner->ext_abort_map = d_inner_map.p;
        inner->hist_order = rep_order;
        TRACE("race: the replicas run");
        st = tbc_batch_run(inner, res.data());
        TRACE("race: the replicas are done");
      }
    }
    if (inner) tbc_batch_destroy(inner);
    HIP_TRY(hipSetDevice(B->device));
    const bool inner_ok = nr != 0 && (st == TBC_OK || st == TBC_ERR_BAD_HISTORY || st == TBC_ERR_MODEL);
    if (!inner_ok && n_res == 0) { decided.release(); d_outer_map.release(); d_inner_map.release(); return st == TBC_OK ? TBC_ERR_HIP : st; }
    if (n_res) HIP_TRY(hipMemcpyAsync(outer_res.data(), B->d_results.p, (size_t)nh * sizeof(DevResult), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));          // the resumed default order: decided, or told to stop by a replica that was
    TRACE("race: the resumed default order is done");
    decided.release(); d_outer_map.release(); d_inner_map.release();
    // ---- who answered: the default order if it decided (its counters are then the whole single search's -- nothing was done twice);
    // else the first replica, in the orders' sequence, that did.  What the others did before they stopped is work done: it is counted.
    std::vector<uint32_t> first_rep(k, 0xFFFFFFFFu);
    for (uint32_t r = nr; r-- > 0;) first_rep[rep_hist[r]] = r;
    for (uint32_t i = 0; i < k; i++) {
      const uint32_t h = pend_all[lo + i];
      const bool resumed = !R.scratched[h] && B->d_park.p != nullptr && n_res != 0;
      if (resumed) B->res_host[h] = outer_res[h];
      const bool outer_decided = resumed && (outer_res[h].valid == TBC_VALID || outer_res[h].valid == TBC_INVALID);
      tbc_counters extra{};
      int win = -1;
      if (inner_ok) for (uint32_t r = first_rep[i]; r < nr && rep_hist[r] == i; r++) {
        if (win < 0 && (res[r].valid == TBC_VALID || res[r].valid == TBC_INVALID)) win = (int)r;
        const tbc_counters& c = res[r].counters;
        extra.steps += c.steps; extra.visited += c.visited; extra.probes += c.probes; extra.backtracks += c.backtracks;
        extra.max_depth = std::max(extra.max_depth, c.max_depth);
      }
      if (outer_decided) {
        // (the normal marshalling answers from res_host; the replicas' work is added to its counters)
        DevResult& dr = B->res_host[h];
        dr.steps += extra.steps; dr.visited += extra.visited; dr.probes += extra.probes; dr.backtracks += extra.backtracks;
        dr.max_depth = std::max<uint64_t>(dr.max_depth, extra.max_depth);
        continue;
      }
      if (win < 0) {          // nobody decided (a limit): the default order's answer stands, or the first replica's if there is no other
        if (resumed || !inner_ok) continue;
        win = (int)first_rep[i];
      }
      tbc_result r = res[(size_t)win];
      r.witness = nullptr; r.n_witness = 0;
      r.counters.steps = extra.steps; r.counters.visited = extra.visited; r.counters.probes = extra.probes; r.counters.backtracks = extra.backtracks;
      r.counters.max_depth = extra.max_depth;
      R.handed[h] = r;
      R.was_handed[h] = 1;
      B->order_of_hist[h] = rep_order[(size_t)win];
    }
    lo = hi;
  }
  return TBC_OK;
}

tbc_status order_restarts(RunState& R) {
  tbc_batch* B = R.B; const uint32_t nh = R.nh; hipStream_t s = R.s;
  std::vector<uint32_t> pend;
  for (uint32_t h = 0; h < nh; h++)
    if (!R.is_seq[h] && R.hist_back[h].status == 0 && B->res_host[h].valid == TBC_UNKNOWN && B->res_host[h].cause == TBC_CAUSE_STEP_LIMIT) pend.push_back(h);
  if (pend.empty()) return TBC_OK;
  // nobody wants a witness: the orders run at the same time (race_orders); with a witness (whose ops live in the answering batch's
  // memory) one after the other, deterministically, as oracle/wgl.py check_restart_pipeline states
  if (!B->opts.want_witness) {
    const tbc_status rs = race_orders(R, pend);
    if (rs != TBC_ERR_OOM) return rs;
    // (no memory for the race beside this batch: what the budget cut short runs on in the default order, as it did before round 6)
    std::vector<uint32_t> left;
    for (uint32_t h : pend) if (!R.was_handed[h] && B->res_host[h].valid == TBC_UNKNOWN && B->res_host[h].cause == TBC_CAUSE_STEP_LIMIT) left.push_back(h);
    B->last_raced = 0;
    const tbc_status fs = left.empty() ? TBC_OK : budgeted_pass(R, left, kCountExact, nullptr, 0);
    R.touched_work = true;
    return fs;
  }
  struct Acc { uint64_t steps, visited, probes, backtracks, max_depth; };
  std::vector<Acc> acc(nh, Acc{0, 0, 0, 0, 0});
  const auto bank = [&](const std::vector<uint32_t>& grp) {
    for (uint32_t h : grp) { const DevResult& d = B->res_host[h]; Acc& a = acc[h]; a.steps += d.steps; a.visited += d.visited; a.probes += d.probes; a.backtracks += d.backtracks; a.max_depth = std::max(a.max_depth, d.max_depth); }
  };
  const std::vector<uint32_t> all = pend;
  const uint32_t default_order = B->list_order();
  tbc_status st = TBC_OK;
  const auto pass = [&](uint32_t order, int64_t steps) -> tbc_status {
    bank(pend);                                            // what the pass before left behind
    B->order_override = order;
    launch_pack_open(make_pack_open_args(B), s, true);      // the lists again, in this pass's order (counts and places stay)
    HIP_TRY(hipGetLastError());
    tbc_status ps = budgeted_pass(R, pend, kCountExact, nullptr, steps);
    if (ps != TBC_OK) return ps;
    std::vector<uint32_t> left;
    for (uint32_t h : pend) {
      const DevResult& d = B->res_host[h];
      if (d.valid == TBC_UNKNOWN && d.cause == TBC_CAUSE_STEP_LIMIT) left.push_back(h);
      else B->order_of_hist[h] = order;
    }
    pend.swap(left);
    return TBC_OK;
  };
  for (uint32_t order : kRestartOrders) {
    if (pend.empty()) break;
    if ((st = pass(order, (int64_t)R.restart_budget)) != TBC_OK) break;
  }
  if (st == TBC_OK && !pend.empty()) st = pass(default_order, 0);
  B->order_override = tbc_batch::kNoOrderOverride;
  if (st != TBC_OK) return st;
  for (uint32_t h : all) {          // counters: the sum over the passes a history went through (its last pass's are in res_host)
    DevResult& d = B->res_host[h]; const Acc& a = acc[h];
    d.steps += a.steps; d.visited += a.visited; d.probes += a.probes; d.backtracks += a.backtracks; d.max_depth = std::max(d.max_depth, a.max_depth);
  }
  R.touched_work = true;
  return TBC_OK;
}

// ---- count form: the histories the budgeted exact search left undecided (oracle/wgl_count.c; tests/test_count_form.py states the
// same pipeline over the oracle).  (1) The RELAXED search -- every class of crashed calls an unlimited supply, counts ignored: a
// superset of the linearizations over a config space no larger than a crash-free history's -- either finds a linearization (then
// the exact search simply needs longer: once more, without a budget) or ends INVALID at completion t: the history is invalid and
// its first bad completion is t or earlier.  (2) The exact search of the PREFIX of t completions: a linearization of it (one
// depth-first descent, not an exhaustion) pins the failing completion at t; if there is none its own exhaustion names an earlier one.
// :configs of an invalid verdict reached this way: when the prefix search names an earlier completion by its own exhaustion, the configs
// it was stuck with (as any exact search); when the prefix of t completions is linearizable -- the t-th completion is what nobody can
// pass -- the ONE config its linearization ended in (the others would take the exhaustion this pipeline exists to avoid).
tbc_status count_form_pipeline(RunState& R) {
  tbc_batch* B = R.B; const uint32_t nh = R.nh;
  HostBuf<Hist>& hist_back = R.hist_back;
  std::vector<uint32_t> pend;
  for (uint32_t h = 0; h < nh; h++)
    if (!R.is_seq[h] && B->res_host[h].valid == TBC_UNKNOWN && B->res_host[h].cause == TBC_CAUSE_STEP_LIMIT) pend.push_back(h);
  if (pend.empty()) return TBC_OK;
  struct Acc { uint64_t steps, visited, probes, backtracks, max_depth; };
  std::vector<Acc> acc(nh, Acc{0, 0, 0, 0, 0});
  const auto bank = [&](const std::vector<uint32_t>& grp) {
    for (uint32_t h : grp) { const DevResult& d = B->res_host[h]; Acc& a = acc[h]; a.steps += d.steps; a.visited += d.visited; a.probes += d.probes; a.backtracks += d.backtracks; a.max_depth = std::max(a.max_depth, d.max_depth); }
  };
  const auto run_pass = [&](const std::vector<uint32_t>& grp, uint32_t mode, const std::vector<uint32_t>* targets) -> tbc_status { return budgeted_pass(R, grp, mode, targets, 0); };
  bank(pend);
  // (what the relaxed SWEEP already decided is not searched again: refuted at a completion -> the prefix pass; valid under the
  // relaxation -> the exact search without a budget; only the others -- no sweep, or a burst that outgrew its sets -- take the relaxed search)
  std::vector<uint32_t> pend_dfs;
  for (uint32_t h : pend) if (R.rs_level[h] == kInf && !R.rs_valid[h]) pend_dfs.push_back(h);
  tbc_status st = pend_dfs.empty() ? TBC_OK : run_pass(pend_dfs, kCountRelaxed, nullptr);
  if (st != TBC_OK) return st;
  bank(pend_dfs);
  for (uint32_t h : pend) {
    DevResult& r = B->res_host[h];
    if (R.rs_valid[h]) { r.valid = TBC_VALID; r.cause = TBC_CAUSE_NONE; }
    if (R.rs_level[h] == kInf) continue;
    const uint32_t t = R.rs_level[h], first = t ? t - 1 : 0;
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
      // (:configs of this verdict: the config the prefix's linearization ended in -- the wide kernel leaves it as record 0 of the history's
      // :configs arena when a prefix search ends VALID, wgl_beam.hip; the narrow kernel does not: none then)
      d.valid = TBC_INVALID; d.cause = TBC_CAUSE_NONE; d.depth = 0; d.n_configs = std::min(d.n_configs, 1u);
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
  R.touched_work = true;
  return TBC_OK;
}

// timings, counters, and the verdicts marshalled into the caller's tbc_result array
tbc_status finish(RunState& R) {
  tbc_batch* B = R.B; hipStream_t s = R.s; const uint32_t nh = R.nh; tbc_result* results = R.results;
  HostBuf<Hist>& hist_back = R.hist_back;
  if (R.touched_work) {   // restore the identity work list for the next run
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
  if (B->lanes && B->ev_turn && R.phase == 0) {      // the search proper: from its turn on the device to its end
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
    if (!R.is_seq[h] && d.tab_log2 > R.final_log2[h]) R.final_log2[h] = d.tab_log2;
    B->sum.table_slots += 1ull << R.final_log2[h];
    if (hist_back[h].status != 0 && worst == TBC_OK) {
      worst = (tbc_status)hist_back[h].status;
      set_error("history %u rejected by the pack kernel: %s", h, tbc_strerror((int)hist_back[h].status));
    }
    if (R.was_handed[h]) {          // answered by the small batch it was handed to: its result, the first pass's counters added
      const tbc_result& g = R.handed[h];
      B->sum.steps += g.counters.steps; B->sum.visited += g.counters.visited; B->sum.probes += g.counters.probes; B->sum.backtracks += g.counters.backtracks;
      if (results) {
        tbc_result& r = results[h];
        r = g;
        r.counters.steps += d.steps; r.counters.visited += d.visited; r.counters.probes += d.probes; r.counters.backtracks += d.backtracks;
        r.counters.max_depth = std::max<uint64_t>(r.counters.max_depth, d.max_depth);
        r.counters.ns_pack = B->timing_ns[1]; r.counters.ns_search = B->timing_ns[2] + B->timing_ns[3]; r.counters.ns_total = t_end - R.t_start;
      }
      continue;
    }
    if (!results) continue;
    tbc_result& r = results[h];
    std::memset(&r, 0, sizeof r);
    r.valid = d.valid; r.cause = d.cause;
    r.analyzer = B->sweep ? (R.by_sweep[h] ? TBC_ALG_LINEAR : TBC_ALG_WGL)
                          : (B->opts.algorithm == TBC_ALG_LINEAR ? TBC_ALG_LINEAR : TBC_ALG_WGL);
    r.fail_op = TBC_NO_OP; r.prev_ok_op = TBC_NO_OP; r.search_width = (B->lanes && !R.is_seq[h] && !R.by_sweep[h]) ? 1u : B->width;
    if (d.valid == TBC_INVALID) {
      r.fail_op = d.fail_op; r.prev_ok_op = d.prev_ok_op;
      tbc_status cs = fill_configs(B, h, d, &r);
      if (cs != TBC_OK) return cs;
    }
    if (d.valid == TBC_VALID) {
      r.final_state = d.final_state; r.n_witness = d.depth;
      if (B->opts.want_witness) {
        r.witness = B->witness_host.data() + B->hist[h].op_off;
        if (R.by_sweep[h]) { r.witness = nullptr; r.n_witness = 0; }      // knossos.linear returns configs, not a linearization
        // (under branch lists the normalised root may pass every completion by itself -- a history of reads of nil / of the initial
        // value: an empty chain, whose expansion is exactly those reads)
        else if ((B->rules & (kRuleEager | kRuleTxnEager)) && !R.is_seq[h] && (d.depth || ((B->rules & kRuleBranch) && hist_back[h].n_ret != 0))) {
          tbc_status ws = expand_eager_witness(B, h, r.witness, &r.n_witness);
          if (ws != TBC_OK) return ws;
        }
      }
    }
    r.counters.steps = d.steps; r.counters.visited = d.visited; r.counters.probes = d.probes;
    r.counters.backtracks = d.backtracks; r.counters.max_depth = d.max_depth;
    r.counters.table_slots = 1ull << R.final_log2[h];
    r.counters.ns_pack = B->timing_ns[1]; r.counters.ns_search = B->timing_ns[2] + B->timing_ns[3];
    r.counters.ns_total = t_end - R.t_start;
  }
  B->sum.ns_pack = B->timing_ns[1]; B->sum.ns_search = B->timing_ns[2] + B->timing_ns[3];
  B->sum.ns_total = t_end - R.t_start;
  if (guard_on() && guard_check("tbc_batch_run", B) != 0) { set_error("TBC_GUARD: a kernel wrote past a device arena (see stderr)"); return TBC_ERR_HIP; }
  return worst;
}

}  // namespace

// phase 0: the whole run.  phase 1 (tbc_batch_sweep_partial): pack + this rank's share of the sweep, stop before the
// verdicts.  phase 2 (tbc_batch_sweep_finish): verdicts from the merged relation table already placed in seg_host.
namespace {
// tbc_batch_progress: the run says it is out and which phase it is in; the search kernels count into word 0 themselves
struct RunningMark {
  tbc_batch* B;
  explicit RunningMark(tbc_batch* b) : B(b) {
    if (B->progress) { ((volatile uint32_t*)B->progress)[0] = 0u; ((volatile uint32_t*)B->progress)[1] = TBC_PHASE_PACK; }
    B->progress_seen.store(0u, std::memory_order_relaxed);
    B->run_t0.store(now_ns(), std::memory_order_relaxed);
    B->running.store(1u, std::memory_order_release);
  }
  void phase(uint32_t p) const { if (B->progress) ((volatile uint32_t*)B->progress)[1] = p; }
  ~RunningMark() { phase(TBC_PHASE_IDLE); B->running.store(0u, std::memory_order_release); }
};
}  // namespace

tbc_status batch_run_impl(tbc_batch* B, tbc_result* results, int phase) {
  HIP_TRY(hipSetDevice(B->device));
  tbc_status st;
  RunningMark mark(B);
  if (phase == 0) {          // a fresh input waiting (batch_stream.hip)?  It becomes the batch's resident input now
    bool consumed = false;
    if ((st = stream_consume(B, &consumed)) != TBC_OK) return st;
  } else if (!B->pending.empty()) {
    set_error("a submitted input is waiting: tbc_batch_run consumes it (the sharded sweep works on the resident input)");
    return TBC_ERR_INVALID_ARG;
  }
  if (B->inputs_stale) { set_error("the last submitted input was refused: there is nothing resident to run (submit another, or destroy and create)"); return TBC_ERR_INVALID_ARG; }
  RunState R(B, results, phase);
  const uint32_t nh = R.nh;
  B->last_raced = 0;
  if (phase != 2 && (st = first_pass(R)) != TBC_OK) return st;
  B->partial_done = phase == 1;
  if (phase == 1) return TBC_OK;
  if (phase == 2) {
    if (!B->sweep || R.hist_back.size() != nh) { set_error("tbc_batch_sweep_finish without tbc_batch_sweep_partial"); return TBC_ERR_INVALID_ARG; }
    // the device table becomes the merged one, so a second pass over overflowed wavefronts updates it in place
    HIP_TRY(hipMemcpyAsync(B->d_sres.p, B->seg_host.data(), B->seg_host.size() * sizeof(SegResult), hipMemcpyHostToDevice, R.s));
  }
  R.final_log2.resize(nh);
  R.is_seq.assign(nh, R.beam ? 0 : 1);
  R.by_sweep.assign(nh, 0);
  for (uint32_t h = 0; h < nh; h++) R.final_log2[h] = R.beam ? B->bh[h].tab_log2 : B->hist[h].tab_log2;
  HIP_TRY(hipEventRecord(B->ev[4], R.s));
  mark.phase(TBC_PHASE_RETRIES);
  if (B->sweep && (st = sweep_verdicts(R)) != TBC_OK) return st;
  if (R.beam && (st = list_overflow_fallback(R)) != TBC_OK) return st;
  R.width_of.assign(nh, B->width);
  R.scratched.assign(nh, 0);
  if ((st = overflow_retries(R)) != TBC_OK) return st;
  if (R.stall_on && (st = stall_handover(R)) != TBC_OK) return st;
  B->order_of_hist.assign(nh, B->list_order());
  if (R.restart_budget && (st = order_restarts(R)) != TBC_OK) { B->order_override = tbc_batch::kNoOrderOverride; return st; }
  if (R.count_budget && (st = count_form_pipeline(R)) != TBC_OK) return st;
  st = finish(R);
  if (B->progress && results) {          // what the run hands back, whoever decided it (a sweep, a replica of a race, a retry from a scratch arena)
    uint32_t decided = 0;
    for (uint32_t h = 0; h < nh; h++) decided += results[h].valid != TBC_UNKNOWN ? 1u : 0u;
    ((volatile uint32_t*)B->progress)[0] = decided;
  }
  if (phase == 0 && B->assign_lists) {          // a fresh input's lists that did not fit their arena: room for the next one
    const tbc_status gs = stream_after_run(B, R.bh_back);
    if (gs != TBC_OK && st == TBC_OK) st = gs;
  }
  return st;
}

}  // namespace tbc

extern "C" tbc_status tbc_batch_run(tbc_batch* b, tbc_result* results) {
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
