// tbc_api.hip -- tbc_check (one history: a persistent context, a batch of one, a run) and the small getters of the C-ABI
// (include/tbcheck.h).  The batch itself lives in batch_create.hip / batch_run.hip / batch_shard.hip / batch_stream.hip (tbc_batch.h).
//
// There is no CPU path in here by design: every compute entry point needs a
// gfx950 device and says TBC_ERR_NO_DEVICE otherwise.
#include "tbc_batch.h"

using namespace tbc;

extern "C" {

int32_t tbc_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  int k = 0;
  for (int d = 0; d < n; d++) if (device_is_gfx950(d)) k++;
  return k;
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
uint32_t tbc_batch_last_raced(const tbc_batch* b) { return b ? b->last_raced : 0; }
tbc_status tbc_batch_progress(const tbc_batch* b, tbc_progress* out) {
  if (!b || !out) { set_error("tbc_batch_progress: null argument"); return TBC_ERR_INVALID_ARG; }
  const volatile uint32_t* w = b->progress;
  out->n_histories = b->n_hist;
  out->running = b->running.load(std::memory_order_acquire);
  uint32_t d = w ? w[0] : 0u, seen = b->progress_seen.load(std::memory_order_relaxed);
  while (d > seen && !b->progress_seen.compare_exchange_weak(seen, d, std::memory_order_relaxed)) {}
  if (out->running && d < seen) d = seen;
  out->n_decided = d < b->n_hist ? d : b->n_hist;       // (a history searched twice -- a retry, a race -- may have been counted twice while the run is out)
  out->phase = w ? w[1] : (uint32_t)TBC_PHASE_IDLE;
  const uint64_t t0 = b->run_t0.load(std::memory_order_relaxed);
  out->elapsed_ns = (out->running && t0) ? now_ns() - t0 : 0ull;
  return TBC_OK;
}
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
