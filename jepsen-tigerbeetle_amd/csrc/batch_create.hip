// batch_create.hip -- tbc_batch_create: host SoA columns in, a batch resident in HBM out.  One planner per decision, in the order the
// decisions depend on each other (round 6: this was ONE function of 500 lines):
//   check_model / check_offsets   what is refused before anything is looked at
//   plan_count_form               crashed calls as counts per class?  (re-numbers the process column) -> mask words
//   plan_engines                  sequential / wide / several histories per wavefront / level sweep / relaxed sweep, rules, lookahead
//   plan_sweep_segments           the sweep's windows and cuts
//   layout_histories              per-history descriptors and the arenas' element counts (shared with batch_stream.hip)
//   alloc_arenas                  every device arena, in the order tbc_check's one-block upload and one-block memset rely on
//   upload_inputs                 columns, descriptors, work list (one pinned block for tbc_check)
//   size_lists_on_device          a big batch: pack + counts once over the resident inputs say how long the per-front lists are
// There is no CPU path in here by design: every compute entry point needs a gfx950 device and says TBC_ERR_NO_DEVICE otherwise.
#include "tbc_batch.h"
#include "reach_table.h"

using namespace tbc;

namespace tbc {

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

// count form of ONE history (tbc_batch.h, CountHist).  Returns false when the form does not apply (more than 128 bits of counts,
// a process id out of range: the pack kernel will say what is wrong with such a history).
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

namespace {

// what the planners hand on to each other
struct CreatePlan {
  const tbc_batch_desc* desc;
  const tbc_model* model;
  const tbc_opts* opts;               // = &B->opts (normalised)
  tbc_batch* B;
  uint32_t nh = 0;
  ColumnScan scan;
  std::vector<int32_t> slot_col;      // count form: the process column re-numbered (re-used slots)
  std::vector<uint32_t> n_slots;      // process slots of each history (count form: the re-used ones)
  bool commutative = false, beam = false, device_sizing = false;
  std::vector<uint32_t> list_caps, rank_scratch;
  LayoutTotals tot;
};

tbc_status check_model(const tbc_batch_desc* desc, const tbc_model* model, const tbc_opts* opts) {
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
  return TBC_OK;
}

// the offsets index the op columns from here on (width heuristic, value scan, sweep sizing): check them first
tbc_status check_offsets(const tbc_batch_desc* desc, tbc_batch* B) {
  const uint32_t nh = desc->n_hist;
  B->total_ops = desc->op_off[nh];
  if (B->total_ops != desc->cols.n) { set_error("op_off[n_hist] (%llu) != cols.n (%u)", (unsigned long long)B->total_ops, desc->cols.n); return TBC_ERR_INVALID_ARG; }
  for (uint32_t h = 0; h < nh; h++) {
    if (desc->op_off[h + 1] < desc->op_off[h] || desc->op_off[h + 1] > B->total_ops || desc->op_off[h + 1] - desc->op_off[h] > 0x7FFFFFFFull) {
      set_error("history %u: bad op_off", h);
      return TBC_ERR_INVALID_ARG;
    }
  }
  if (desc->op_off[0] != 0) { set_error("op_off[0] must be 0"); return TBC_ERR_INVALID_ARG; }
  return TBC_OK;
}

uint32_t mask_words_for(uint32_t max_slots) {
  const uint32_t mw = (max_slots + 63) / 64;
  return mw <= 1 ? 1 : mw <= 2 ? 2 : mw <= 4 ? 4 : mw <= 8 ? 8 : 16;
}

// ---- count form (tbc_internal.h, kRuleCount): a register / cas-register batch with crashed calls that have an effect, under the
// default rules and knossos.competition (the published orders -- TBC_ALG_WGL, TBC_ALG_LINEAR -- keep a mask bit per crashed call).
// The process column is re-numbered (re-used slots; a crashed call holds none) and the crashed calls become classes with counts.
// Leaves the batch's mask words (and the sequential kernel's frame words) behind.
tbc_status plan_count_form(CreatePlan& P) {
  tbc_batch* B = P.B; const tbc_batch_desc* desc = P.desc; const tbc_model* model = P.model; const tbc_opts* opts = P.opts;
  const uint32_t nh = P.nh;
  P.scan = scan_columns(desc->cols, B->total_ops);
  B->any_crashed = P.scan.any_crashed;
  {
    const bool regfam = model->kind == TBC_MODEL_REGISTER || model->kind == TBC_MODEL_CAS_REGISTER;
    const tbc_ops& c = desc->cols;
    bool want = regfam && opts->algorithm == TBC_ALG_COMPETITION && (opts->dominance & (TBC_DOM_NO_EAGER_READS | TBC_DOM_NO_TWIN_RULE | TBC_DOM_NO_COUNT_FORM)) == 0 &&
                opts->search_width != 1 && opts->lanes_per_history != 4 && opts->lookahead != 1 &&      // (several histories per wavefront: 8 / 16 / 32 lanes in the count form)
                (model->init == TBC_NIL || (model->init >= 0 && model->init <= kMaxRuleValue));
    want = want && P.scan.nonneg && P.scan.vmax <= kMaxRuleValue;
    const bool any = P.scan.any_crashed_effect;
    if (want && any) {
      P.slot_col.resize((size_t)B->total_ops + 1);
      B->count_hist.resize(nh);
      std::vector<uint32_t> rets; std::vector<int32_t> slot_of; std::vector<uint8_t> used;
      bool ok = true;
      for (uint32_t h = 0; h < nh && ok; h++)
        ok = build_count_form(c, desc->op_off[h], desc->op_off[h + 1] - desc->op_off[h], desc->n_process[h], model->kind == TBC_MODEL_CAS_REGISTER,
                              P.slot_col.data() + desc->op_off[h], B->count_hist[h], rets, slot_of, used);
      B->count_form = ok;
      if (!ok) { P.slot_col.clear(); B->count_hist.clear(); }
    }
  }
  const auto slots_of = [&](uint32_t h) -> uint32_t { return B->count_form ? B->count_hist[h].n_slots : desc->n_process[h]; };
  uint32_t maxW = 1;
  for (uint32_t h = 0; h < nh; h++) maxW = std::max(maxW, slots_of(h));
  if (maxW > kMaxSlots) { set_error("%u open processes > %u supported", maxW, kMaxSlots); return TBC_ERR_WINDOW_TOO_WIDE; }
  B->mask_words = mask_words_for(maxW);
  B->frame_words = search_frame_words(B->mask_words);
  if (B->count_form && B->mask_words > 2) { B->count_form = false; P.slot_col.clear(); B->count_hist.clear(); }   // (the count form's kernel: one or two mask words)
  if (!B->count_form && maxW != 1) {        // (the masks are the mask form's after all)
    maxW = 1;
    for (uint32_t h = 0; h < nh; h++) maxW = std::max(maxW, desc->n_process[h]);
    if (maxW > kMaxSlots) { set_error("%u open processes > %u supported", maxW, kMaxSlots); return TBC_ERR_WINDOW_TOO_WIDE; }
    B->mask_words = mask_words_for(maxW);
    B->frame_words = search_frame_words(B->mask_words);
  }
  P.n_slots.resize(nh);
  for (uint32_t h = 0; h < nh; h++) P.n_slots[h] = std::max(1u, slots_of(h));
  return TBC_OK;
}

// which engines answer this batch, at which width, under which rules
tbc_status plan_engines(CreatePlan& P) {
  tbc_batch* B = P.B; const tbc_batch_desc* desc = P.desc; const tbc_model* model = P.model; const tbc_opts* opts = P.opts;
  const uint32_t nh = P.nh;
  uint32_t width = opts->search_width ? opts->search_width : (opts->algorithm == TBC_ALG_WGL ? 1u : 4u);   // 4: fewest rounds per history, measured
  if (width > 16) width = 16;           // one wavefront per history: at most 16 configs per round
  while (width & (width - 1)) width &= width - 1;   // the wide kernels take a power of two
  if (B->mask_words > 4) width = 1;          // very wide windows: sequential kernel only
  const bool commutative = P.commutative = model->kind == TBC_MODEL_SET || model->kind == TBC_MODEL_BANK;
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
  const bool beam = P.beam = width > 1;
  // dominance rules: same scope as the lookahead, and every register value must index the per-front read table
  if (B->lookahead || (width > 1 && width <= 16 && (model->kind == TBC_MODEL_REGISTER || model->kind == TBC_MODEL_CAS_REGISTER))) {
    const int32_t vmax = std::max(model->init == TBC_NIL ? -1 : model->init, P.scan.vmax);
    const bool in_range = (model->init == TBC_NIL || model->init >= 0) && P.scan.nonneg;
    if (in_range && vmax <= kMaxRuleValue) {
      B->n_dom = (uint32_t)(vmax + 2);                   // nil + 0..vmax: the states a register can be in
      B->rules = ((opts->dominance & TBC_DOM_NO_EAGER_READS) ? 0u : kRuleEager) | ((opts->dominance & TBC_DOM_NO_TWIN_RULE) ? 0u : kRuleTwin);
      if (B->count_form) B->rules |= kRuleCount;
      B->vpad = 2; while (B->vpad < (uint32_t)(vmax + 2)) B->vpad <<= 1;
    }
  }
  // multi-register under the wide schedule: eager pure-read txns, txn independence (tbc_internal.h kRuleTxnEager / kRuleTxnIndep)
  if (beam && !B->sweep && model->kind == TBC_MODEL_MULTI_REGISTER)
    B->rules |= ((opts->dominance & TBC_DOM_NO_EAGER_TXNS) ? 0u : kRuleTxnEager) | ((opts->dominance & TBC_DOM_NO_TXN_INDEPENDENCE) ? 0u : kRuleTxnIndep);
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
  // the frames arena is the pack kernels' scratch (3 words per op) and the sequential kernel's stack (4 + 2 mask words per op): a
  // wide-schedule batch only needs the former -- the rare history that falls back to the sequential kernel gets frames of its own then
  if (beam) B->frame_words = 3;
  return TBC_OK;
}

// segments: enough wavefronts to fill the GPU several times over, none shorter than 32 completions; cuts need the
// register family's value domain (nil + 0..vmax = vpad's range) to enumerate the configs possible at a front
void plan_sweep_segments(CreatePlan& P) {
  tbc_batch* B = P.B; const tbc_batch_desc* desc = P.desc; const tbc_model* model = P.model;
  const uint32_t nh = P.nh;
  if (!(B->sweep || B->rsweep)) return;
  uint64_t max_n = 1;
  for (uint32_t h = 0; h < nh; h++) max_n = std::max<uint64_t>(max_n, desc->op_off[h + 1] - desc->op_off[h]);
  const char* env = std::getenv("TBC_SWEEP_SEG");
  uint64_t T = env ? std::strtoull(env, nullptr, 10) : std::max<uint64_t>(32, (max_n * nh + 4095) / 4096);
  // one history or a handful -- the workgroup kernel's case (at most 4,096 workgroups): windows of 48 completions.  Measured round 5 with
  // the compact walk (profiles/r05_sweep_segment_length.txt): one 10k-op history 1.07 ms at 32, 0.97 - 1.02 at 40, 0.97 - 0.99 at 48,
  // 1.11 at 56, 1.20 at 64 (fewer workgroups, shorter table to bring back and compose; past 48 the longest segment costs more than that saves)
  if (!env && T < 48 && (uint64_t)nh * ((max_n + 47) / 48) * kSweepSlices <= 4096) T = 48;
  // the relaxed sweep exists in the workgroup kernel only, which takes at most 4,096 workgroups: 3 - 8 count-form histories of ~10k ops
  // cut into windows of 32 were up to 10,016 -- launch_sweep refused, silently, and the relaxed sweep never ran for the very batches
  // it was built for.  Their windows are as long as it takes to stay within what the kernel takes.
  if (!env && B->rsweep) {
    const uint64_t cap_segs = std::max<uint64_t>(1, 4096u / (kSweepSlices * (uint64_t)nh));
    T = std::max<uint64_t>(T, (max_n + cap_segs - 1) / cap_segs);
  }
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

}  // namespace

// Per-history descriptors and the arenas' element counts.  What depends on the batch (mask words, frame words, wide or not, count form,
// sweep, visited-set sizing) is read from B; what depends on the input is passed in -- tbc_batch_create and every fresh input
// (batch_stream.hip) lay their histories out with this one function.
tbc_status layout_histories(tbc_batch* B, uint32_t nh, const uint64_t* op_off, const uint32_t* n_events, const uint32_t* n_slots, const int32_t* aux,
                            const uint32_t* list_caps, bool lists_on_device, std::vector<Hist>& hist, std::vector<BeamHist>& bh, LayoutTotals& tot,
                            const std::vector<CountHist>* count_hist) {
  const bool beam = B->width > 1;
  if (!count_hist) count_hist = &B->count_hist;
  const uint32_t KW = 1 + B->mask_words, EW = B->entry_words();
  const uint64_t default_cap_bytes = 1ull << 30;
  const uint64_t max_bytes = B->opts.max_visited_bytes ? B->opts.max_visited_bytes : default_cap_bytes;
  hist.resize(nh);
  if (beam) bh.resize(nh); else bh.clear();
  tot = LayoutTotals{};
  tot.total_ops = op_off[nh];
  for (uint32_t h = 0; h < nh; h++) {
    Hist& H = hist[h];
    std::memset(&H, 0, sizeof H);
    const uint64_t n = op_off[h + 1] - op_off[h];
    if (op_off[h + 1] < op_off[h] || n > 0x7FFFFFFFull) { set_error("history %u: bad op_off", h); return TBC_ERR_INVALID_ARG; }
    H.op_off = op_off[h];
    H.n_ops = (uint32_t)n;
    tot.max_ops = std::max<uint64_t>(tot.max_ops, n);
    H.n_events = n_events[h];
    H.n_slots = std::max(1u, n_slots[h]);
    H.flags = B->count_form ? kHistCount : 0u;
    H.aux = aux ? aux[h] : B->model.init;
    H.rec_off = tot.rec_n; tot.rec_n += n + 2ull * H.n_slots;
    H.seg_off = tot.seg_n; tot.seg_n += H.n_slots + 1;
    H.ret_off = H.op_off;
    H.bm_off = tot.bm_n; tot.bm_n += H.n_events / 32 + 1;
    H.frame_off = tot.frame_n; tot.frame_n += std::max<uint64_t>(n, 1) * B->frame_words;
    const uint64_t per_op = B->opts.visited_per_op ? B->opts.visited_per_op : 64;
    uint32_t lg = std::max(10u, ceil_log2(per_op * std::max<uint64_t>(n, 1)));
    while (lg > 10 && (1ull << lg) * KW * 8 > max_bytes) lg--;
    H.tab_log2 = lg;
    H.tab_off = tot.tab_n;
    if (!beam) { tot.tab_n += (1ull << lg) * KW; tot.tab_n = (tot.tab_n + 1) & ~1ull; }   // keep 16 B alignment
    if (beam) {
      BeamHist& Q = bh[h];
      std::memset(&Q, 0, sizeof Q);
      uint32_t blg = lg;
      while (blg > 10 && ((1ull << blg) * EW * 8 > max_bytes || blg > kBeamMaxTabLog2)) blg--;
      Q.tab_log2 = blg;
      Q.off_off = tot.boff_n; tot.boff_n += n + 2;
      if (B->count_form) {
        const CountHist& ch = (*count_hist)[h];
        Q.cmem_off = tot.bocc_n; tot.bocc_n += ch.words.size();
        Q.n_classes = ch.n_classes; Q.top[0] = ch.top[0]; Q.top[1] = ch.top[1];
      }
      Q.lst_cap = lists_on_device ? 0xFFFFFFF0u : list_caps[h];
      Q.lst_off = tot.blst_n; tot.blst_n += lists_on_device ? 0u : Q.lst_cap;
      Q.stack_off = tot.bstack_n; tot.bstack_n += (1ull << blg);
      Q.tab_off = tot.btab_n; tot.btab_n += (1ull << blg);
    }
  }
  if (B->sweep) { tot.bstack_n = 0; tot.btab_n = 0; }     // the sweep has no visited set; its fallback takes scratch arenas
  return TBC_OK;
}

}  // namespace tbc

void tbc_batch::count_device_bytes() {
  tbc_batch* const B = this;
  const bool beam = B->width > 1;
  B->device_bytes = B->d_f.bytes() + B->d_a.bytes() + B->d_b.bytes() + B->d_proc.bytes() + B->d_inv.bytes() +
                    B->d_ret.bytes() + B->d_hist.bytes() + B->d_rec.bytes() + B->d_seg.bytes() + B->d_ret_slot.bytes() +
                    B->d_ret_op.bytes() + B->d_bitmap.bytes() + B->d_wpre.bytes() + B->d_frames.bytes() +
                    B->d_tab.bytes() + B->d_results.bytes() + B->d_work.bytes() + B->d_witness.bytes();
  if (beam) B->device_bytes += B->d_bh.bytes() + B->d_off.bytes() + B->d_ncr.bytes() + B->d_lst.bytes() +
                               B->d_crashed.bytes() + B->d_slot8.bytes() + B->d_rk8.bytes() + B->d_twn.bytes() + B->d_rdm.bytes() + B->d_look.bytes() + B->d_looktmp.bytes() + B->d_dstack.bytes() + B->d_stack.bytes() + B->d_btab.bytes() + B->d_pool.bytes() + B->d_cmem.bytes();
  if (B->d_stage[0].p) B->device_bytes += B->d_stage[0].bytes();
  if (B->d_stage[1].p) B->device_bytes += B->d_stage[1].bytes();
}

tbc_batch::~tbc_batch() {
  tbc::stream_release(this);
  d_f.release(); d_a.release(); d_b.release(); d_proc.release(); d_inv.release(); d_ret.release();
  d_hist.release(); d_rec.release(); d_seg.release(); d_ret_slot.release(); d_ret_op.release();
  d_bitmap.release(); d_wpre.release(); d_frames.release(); d_witness.release(); d_work.release();
  d_queue.release(); d_tab.release(); d_results.release(); d_table.release(); d_pool_vals.release(); d_cfg.release();
  d_bh.release(); d_off.release(); d_ncr.release(); d_lst.release(); d_crashed.release(); d_stack.release();
  d_zncr.release(); d_reach.release(); d_reach_hdr.release(); d_abort.release(); d_order.release(); d_park.release();
  if (abort_one) (void)hipHostFree(abort_one);
  if (progress) (void)hipHostFree(progress);
  d_progress.release();
  if (!borrowed) { for (auto& e : ev2) if (e) (void)hipEventDestroy(e); if (stream2) (void)hipStreamDestroy(stream2); }
  d_cmem.release(); d_occ.release(); d_btab.release(); d_slot8.release(); d_rk8.release(); d_look.release(); d_looktmp.release(); d_twn.release(); d_rdm.release(); d_cuts.release(); d_seglist.release(); d_sres.release(); d_dstack.release(); d_pool.release(); d_pool_cursor.release();
  if (ev_turn) (void)hipEventDestroy(ev_turn);
  if (!borrowed) {
    for (auto& e : ev) if (e) (void)hipEventDestroy(e);
    if (stream) (void)hipStreamDestroy(stream);
  }
}

namespace tbc {
namespace {

// Every device arena of the batch, sized from the layout's totals -- in the order tbc_check relies on (its arenas are consecutive pieces
// of its context's slab: what is uploaded -- the six columns, the descriptors, the work list -- first, as one block (upload_inputs);
// then what every run zeroes, as one memset (batch_run.hip, zero_block)).  The lists (d_lst, d_twn) of a big batch come later
// (size_lists_on_device).
tbc_status alloc_arenas(CreatePlan& P) {
  tbc_batch* B = P.B; const tbc_opts* opts = P.opts; const tbc_model* model = P.model;
  const uint32_t nh = P.nh;
  const bool beam = P.beam, device_sizing = P.device_sizing;
  const LayoutTotals& t = P.tot;
  const uint32_t EW = B->entry_words();
  tbc_status s;
  const uint64_t T = B->total_ops;
  // (progress words: host memory the kernels write and another host thread reads, coherent like the debug words of batch_common.hip.
  // tbc_check's one-shot batches, which borrow a context and hand out no handle, have none: nobody could ask)
  if (!t_ctx) {
    HIP_TRY(hipHostMalloc((void**)&B->progress, 16 * sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(B->progress, 0, 16 * sizeof(uint32_t));
    if ((s = B->d_progress.alloc(1))) return s;
  }
  if ((s = B->d_f.alloc(T)) || (s = B->d_a.alloc(T)) || (s = B->d_b.alloc(T)) || (s = B->d_proc.alloc(T)) ||
      (s = B->d_inv.alloc(T)) || (s = B->d_ret.alloc(T)) || (s = B->d_hist.alloc(nh)) || (s = B->d_bh.alloc(beam ? nh : 0)) || (s = B->d_work.alloc(nh)) ||
      (s = B->d_bitmap.alloc(t.bm_n)) || (s = B->d_off.alloc(beam ? t.boff_n : 0)) || (s = B->d_ncr.alloc(beam ? t.boff_n : 0)) || (s = B->d_pool_cursor.alloc(1)) ||
      (s = B->d_rec.alloc(t.rec_n)) ||
      (s = B->d_seg.alloc(t.seg_n)) || (s = B->d_ret_slot.alloc(T)) || (s = B->d_ret_op.alloc(T)) ||
      (s = B->d_wpre.alloc(t.bm_n)) || (s = B->d_frames.alloc(t.frame_n)) ||
      (s = B->d_tab.alloc(t.tab_n)) || (s = B->d_results.alloc(nh)) ||
      (s = B->d_queue.alloc(4)) || (s = B->d_witness.alloc(opts->want_witness ? T : 0)))
    return s;
  if (beam) {
    if ((!device_sizing && (s = B->d_lst.alloc(t.blst_n))) || (s = B->d_crashed.alloc((B->count_form || !B->any_crashed) ? 0 : T)) || (s = B->d_cmem.alloc(B->count_form ? t.bocc_n : 0)) ||
        (s = B->d_slot8.alloc(slot8_bytes(T, nh))) || (s = B->d_stack.alloc(t.bstack_n)) || (s = B->d_btab.alloc(t.btab_n * B->tab_stride())))
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
      if ((s = B->d_zncr.alloc(t.boff_n)) || (s = B->d_reach.alloc(reach.size())) || (s = B->d_reach_hdr.alloc(hdr.size())) || (s = B->d_abort.alloc(nh))) return s;
      HIP_TRY(hipMemset(B->d_zncr.p, 0, B->d_zncr.bytes()));
      HIP_TRY(hipMemcpy(B->d_reach.p, reach.data(), reach.size() * 4, hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy(B->d_reach_hdr.p, hdr.data(), hdr.size() * 4, hipMemcpyHostToDevice));
      // (the abort words are set by an asynchronous copy on the sweep's stream: its source is pinned and lives as long as the batch)
      HIP_TRY(hipHostMalloc((void**)&B->abort_one, sizeof(uint32_t), hipHostMallocDefault));
      *B->abort_one = 1u;
    }
    if (B->lanes && (s = B->d_rk8.alloc(slot8_bytes(T, nh)))) return s;
    if (B->reg_rules() && ((!device_sizing && (s = B->d_twn.alloc(t.blst_n * B->mask_words))) || (s = B->d_rdm.alloc(B->lanes ? 1 : T * B->vpad * B->mask_words)))) return s;
    // several histories per wavefront: front records (tbc_internal.h) instead of plain rows, with or without the rules
    if (B->lanes) { B->d_rdm.release(); if ((s = B->d_rdm.alloc(T * B->front_words()))) return s; }
    // (d_looktmp: scratch of the walk with lane = process slot only -- launch_pack_open's choice, repeated here)
    const bool by_front = B->mask_words == 1 && B->vpad <= 32;
    if (B->lookahead && ((s = B->d_look.alloc(look_words(T, nh, B->mask_words))) || (s = B->d_looktmp.alloc(by_front ? 0 : T)) ||
                         (s = B->d_dstack.alloc(t.bstack_n)))) return s;
    // growth pool: 30 % of the visited-set arena -- 10 % for the big quiet batches that run several histories per wavefront, whose sets
    // rarely grow (a history that outgrows its table and finds the pool empty is run again from a scratch arena: at 32 calls in
    // flight a 10 % pool cost 22 s of such retries per 2,048 histories) --, at least room for one history to grow twice (4x, then 16x: keys, parents, two stacks, slot translation), at most 32 GiB
    {
      uint64_t biggest = 0;
      for (uint32_t h = 0; h < nh; h++) biggest = std::max<uint64_t>(biggest, 1ull << B->bh[h].tab_log2);
      uint64_t words = std::max<uint64_t>(t.btab_n * B->tab_stride() * (B->lanes ? 1u : 3u) / 10, biggest * (4 + 16 + 4) * (EW + 1));
      words = std::min<uint64_t>(words, (32ull << 30) / 8);
      if (B->sweep) words = 1;
      if ((s = B->d_pool.alloc(words))) return s;
    }
  }
  if ((s = B->d_cfg.alloc((uint64_t)nh * kCfgCap * (2 + B->mask_words)))) return s;
  if (beam && (s = B->d_park.alloc((uint64_t)nh * kParkWords))) return s;
  B->pool_len = P.desc->cols.pool ? P.desc->cols.pool_len : 0;
  if ((s = B->d_pool_vals.alloc(B->pool_len))) return s;
  if (B->pool_len) HIP_TRY(hipMemcpy(B->d_pool_vals.p, P.desc->cols.pool, (size_t)B->pool_len * 4, hipMemcpyHostToDevice));
  if (model->kind == TBC_MODEL_TABLE) {
    const size_t tn = (size_t)model->n_states * model->n_classes;
    B->table_host.assign(model->table, model->table + tn);
    if ((s = B->d_table.alloc(tn))) return s;
    HIP_TRY(hipMemcpy(B->d_table.p, B->table_host.data(), tn * 2, hipMemcpyHostToDevice));
    B->model.table = nullptr;
  }
  B->count_device_bytes();
  return TBC_OK;
}

// the batch's stream and events: its own, or its context's (tbc_check)
tbc_status take_streams(tbc_batch* B) {
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
  return TBC_OK;
}

// inputs become resident.  *done = the whole upload went up as one queued block (tbc_check) and nothing else is left to do
tbc_status upload_inputs(CreatePlan& P, bool* done) {
  tbc_batch* B = P.B; const tbc_batch_desc* desc = P.desc;
  const uint32_t nh = P.nh;
  const bool beam = P.beam, device_sizing = P.device_sizing;
  const uint64_t T = B->total_ops, bocc_n = P.tot.bocc_n;
  *done = false;
  std::vector<uint32_t> work(nh);
  for (uint32_t h = 0; h < nh; h++) work[h] = h;
  // tbc_check: columns, descriptors and work list are consecutive pieces of the context's slab -- staged in the context's pinned
  // region and uploaded as ONE copy that nobody waits for (the run's kernels follow it in stream order; the region lives until the
  // call ends).  Eight staged copies of pageable memory and a synchronize were 77 us of a 1.07 ms call.
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
        std::memcpy(st + at(B->d_proc.p), B->count_form ? P.slot_col.data() : desc->cols.process, T * 4);
        std::memcpy(st + at(B->d_inv.p), desc->cols.inv_pos, T * 4);
        std::memcpy(st + at(B->d_ret.p), desc->cols.ret_pos, T * 4);
        std::memcpy(st + at(B->d_hist.p), B->hist.data(), nh * sizeof(Hist));
        if (beam) std::memcpy(st + at(B->d_bh.p), B->bh.data(), nh * sizeof(BeamHist));
        std::memcpy(st + at(B->d_work.p), work.data(), (size_t)nh * 4);
        HIP_TRY(hipMemcpyAsync(base, st, total, hipMemcpyHostToDevice, B->stream));
        B->inputs_fresh = true;
        *done = true;
        TRACE("create: inputs queued as one block");
        return TBC_OK;
      }
    }
  }
  if (T) {
    HIP_TRY(hipMemcpyAsync(B->d_f.p, desc->cols.f, T, hipMemcpyHostToDevice, B->stream));
    HIP_TRY(hipMemcpyAsync(B->d_a.p, desc->cols.a, T * 4, hipMemcpyHostToDevice, B->stream));
    HIP_TRY(hipMemcpyAsync(B->d_b.p, desc->cols.b, T * 4, hipMemcpyHostToDevice, B->stream));
    HIP_TRY(hipMemcpyAsync(B->d_proc.p, B->count_form ? P.slot_col.data() : desc->cols.process, T * 4, hipMemcpyHostToDevice, B->stream));
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
  return TBC_OK;
}

// the narrow kernel addresses lists and fronts with 32-bit element offsets (wgl_narrow_impl.h): a batch past that keeps a wavefront per history
bool lists_too_long(const tbc_batch* B, const LayoutTotals& t) { return B->lanes && (t.blst_n >= (1ull << 32) || t.boff_n >= (1ull << 32)); }

// A big batch: the pack and counts kernels over the resident inputs say how many entries each history's per-front lists hold
// (BeamHist.lst_need); the list arenas are allocated after that.  (A pass over every history's events on the host was 1.5 of
// tbc_batch_create's 1.7 s for 32,768 histories -- profiles/r04_cold_batch.log; the kernels say the same numbers in 30 ms.)
tbc_status size_lists_on_device(CreatePlan& P) {
  tbc_batch* B = P.B; const tbc_opts* opts = P.opts;
  const uint32_t nh = P.nh;
  tbc_status s;
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
    P.tot.blst_n = 0;
    for (uint32_t h = 0; h < nh; h++) {
      BeamHist& Q = B->bh[h];
      Q.lst_cap = std::max(1u, back[h].lst_need);
      Q.lst_off = P.tot.blst_n; P.tot.blst_n += Q.lst_cap;
    }
    if (!lists_too_long(B, P.tot)) break;
    if (opts->lanes_per_history) { set_error("lanes_per_history: the batch's open-call lists exceed 2^32 entries; split the batch"); return TBC_ERR_UNSUPPORTED; }
    const bool had_branch = (B->rules & kRuleBranch) != 0;
    B->lanes = 0; B->rules &= ~kRuleBranch;
    if (!opts->want_witness) {                   // (a wavefront per history keeps parent links whatever the caller wants: the arena grows by them)
      B->d_btab.release();
      if ((s = B->d_btab.alloc(P.tot.btab_n * B->tab_stride()))) return s;
    }
    if (!had_branch) break;                      // (else the lists hold the reads again: counted once more)
    for (uint32_t h = 0; h < nh; h++) { B->bh[h].lst_cap = 0xFFFFFFF0u; B->bh[h].lst_off = 0; }
  }
  if ((s = B->d_lst.alloc(P.tot.blst_n)) || (B->reg_rules() && (s = B->d_twn.alloc(P.tot.blst_n * B->mask_words)))) return s;
  B->count_device_bytes();
  TRACE("create: lists sized on the device");
  return TBC_OK;
}

}  // namespace

tbc_status batch_create_impl(const tbc_batch_desc* desc, const tbc_model* model, const tbc_opts* opts, tbc_batch* B) {
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
  tbc_status s;
  if ((s = check_model(desc, model, opts))) return s;
  CreatePlan P{};
  P.desc = desc; P.model = model; P.opts = opts; P.B = B;
  const uint32_t nh = P.nh = desc->n_hist;
  B->n_hist = nh;
  TRACE("create: begin");
  if ((s = check_offsets(desc, B))) return s;
  if ((s = plan_count_form(P))) return s;
  if ((s = plan_engines(P))) return s;
  plan_sweep_segments(P);
  const bool beam = P.beam;

  // how many entries each history's per-front lists hold.  A few histories (tbc_check: latency matters): a pass over each
  // history's events on the host.  A big batch: the pack and counts kernels, which every run launches anyway, say the same numbers
  // in 30 ms once the inputs are resident (size_lists_on_device), and the list arenas are allocated after that.
  const bool device_sizing = P.device_sizing = beam && nh > 64;
  if (beam && !device_sizing) {
    P.list_caps.assign(nh, 0u);
    const bool branch = (B->rules & kRuleBranch) != 0;
    for (uint32_t h = 0; h < nh; h++) {
      const uint64_t n = desc->op_off[h + 1] - desc->op_off[h];
      // (tbc_check: the arenas are pieces of a slab that is there already -- where the worst case, every slot at every front, is a few MB,
      // take it and skip the pass over the history's events: 20 us of a 1 ms call)
      const uint64_t worst = std::max<uint64_t>(n, 1) * P.n_slots[h];
      if (t_ctx && worst * (sizeof(OpRec) + 8 * B->mask_words) <= (24ull << 20)) P.list_caps[h] = (uint32_t)worst;
      else P.list_caps[h] = (uint32_t)std::min<uint64_t>(0xFFFFFFF0ull, open_list_entries(desc->cols, desc->op_off[h], n, desc->n_events[h], P.n_slots[h], P.rank_scratch, branch));
    }
  }
  TRACE("create: lists sized");
  if ((s = layout_histories(B, nh, desc->op_off, desc->n_events, P.n_slots.data(), desc->model_aux, device_sizing ? nullptr : P.list_caps.data(), device_sizing || !beam,
                            B->hist, B->bh, P.tot))) return s;
  B->max_ops = std::max<uint64_t>(B->max_ops, P.tot.max_ops);

  if (!device_sizing && lists_too_long(B, P.tot)) {
    if (opts->lanes_per_history) { set_error("lanes_per_history: the batch's open-call lists exceed 2^32 entries; split the batch"); return TBC_ERR_UNSUPPORTED; }
    const bool had_branch = (B->rules & kRuleBranch) != 0;
    B->lanes = 0; B->rules &= ~kRuleBranch;
    if (had_branch) {                          // the lists hold the reads again: size them for that
      P.tot.blst_n = 0;
      for (uint32_t h = 0; h < nh; h++) {
        const Hist& H = B->hist[h];
        BeamHist& Q = B->bh[h];
        Q.lst_cap = (uint32_t)std::min<uint64_t>(0xFFFFFFF0ull, open_list_entries(desc->cols, H.op_off, H.n_ops, H.n_events, H.n_slots, P.rank_scratch, false));
        Q.lst_off = P.tot.blst_n; P.tot.blst_n += Q.lst_cap;
      }
    }
  }
  if (device_sizing && B->lanes && P.tot.boff_n >= (1ull << 32)) {
    if (opts->lanes_per_history) { set_error("lanes_per_history: the batch exceeds 2^32 fronts; split the batch"); return TBC_ERR_UNSUPPORTED; }
    B->lanes = 0; B->rules &= ~kRuleBranch;
  }
  TRACE("create: layout done");
  if ((s = alloc_arenas(P))) return s;
  if ((s = take_streams(B))) return s;
  TRACE("create: arenas allocated");
  bool done = false;
  if ((s = upload_inputs(P, &done))) return s;
  if (!done && device_sizing && (s = size_lists_on_device(P))) return s;
  B->cap = P.tot;
  B->cap.total_ops = B->total_ops;
  B->res_host.resize(nh);
  return TBC_OK;
}

}  // namespace tbc

extern "C" tbc_status tbc_batch_create(const tbc_batch_desc* desc, const tbc_model* model,
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
