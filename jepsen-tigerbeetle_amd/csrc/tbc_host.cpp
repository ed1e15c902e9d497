// tbc_host.cpp -- host-side pieces of the C-ABI that need no device:
//   tbc_pair_events  = knossos.history/complete + /without-failures + pairing
//   tbc_memo_build   = knossos.model.memo/memo
//   error strings
// (Knossos is not in /root/reference -- SURVEY.md section 0 F1; semantics recalled in
// section 8a.  The reference's own uses of the same history helpers:
// tests/ledger.clj:206,239; checker/perf.clj:617,623.)
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>
#include "tbc_internal.h"

namespace tbc {
thread_local std::string g_last_error;
void set_error(const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_last_error = buf;
}
}  // namespace tbc

extern "C" {

uint32_t tbc_version(void) { return TBC_ABI_VERSION; }

const char* tbc_last_error(void) { return tbc::g_last_error.c_str(); }

const char* tbc_strerror(int status) {
  switch (status) {
    case TBC_OK: return "ok";
    case TBC_ERR_INVALID_ARG: return "invalid argument";
    case TBC_ERR_BAD_HISTORY: return "malformed history";
    case TBC_ERR_NO_DEVICE: return "no usable gfx950 device (this library has no CPU fallback)";
    case TBC_ERR_OOM: return "out of memory";
    case TBC_ERR_WINDOW_TOO_WIDE: return "too many open processes for the search window";
    case TBC_ERR_MODEL: return "op not understood by the model";
    case TBC_ERR_HIP: return "HIP runtime error";
    case TBC_ERR_UNSUPPORTED: return "unsupported";
    default: return "unknown status";
  }
}

tbc_status tbc_pair_events(const tbc_events* ev, uint8_t* f, int32_t* a, int32_t* b,
                           int32_t* process, uint32_t* inv_pos, uint32_t* ret_pos,
                           uint32_t* n_ops, uint32_t* n_process) {
  if (!ev || !n_ops || !n_process || (ev->n && (!ev->type || !ev->process || !ev->f || !ev->a || !ev->b)) ||
      (ev->n && (!f || !a || !b || !process || !inv_pos || !ret_pos))) {
    tbc::set_error("tbc_pair_events: null argument");
    return TBC_ERR_INVALID_ARG;
  }
  // pass 1: pair.  open_of[process] = index of its open op in the scratch arrays.
  const uint32_t n = ev->n;
  std::unordered_map<int32_t, uint32_t> open_of;   // process -> op (scratch index)
  std::unordered_map<int32_t, bool> retired;       // processes that crashed
  open_of.reserve(256);
  std::vector<uint8_t> dead;                        // scratch op deleted by :fail
  dead.reserve(n / 2 + 1);
  uint32_t m = 0;                                   // scratch ops, invocation order
  for (uint32_t i = 0; i < n; i++) {
    const int32_t p = ev->process[i];
    switch (ev->type[i]) {
      case TBC_INVOKE: {
        if (open_of.count(p)) {
          tbc::set_error("row %u: process %d invokes while its previous op is still open", i, p);
          return TBC_ERR_BAD_HISTORY;
        }
        if (retired.count(p)) {
          tbc::set_error("row %u: process %d invokes after an :info (crashed) op", i, p);
          return TBC_ERR_BAD_HISTORY;
        }
        open_of[p] = m;
        f[m] = ev->f[i]; a[m] = ev->a[i]; b[m] = ev->b[i]; process[m] = p;
        inv_pos[m] = i; ret_pos[m] = TBC_POS_CRASHED;
        dead.push_back(0);
        m++;
        break;
      }
      case TBC_OK_: case TBC_FAIL: case TBC_INFO: {
        auto it = open_of.find(p);
        if (it == open_of.end()) {
          tbc::set_error("row %u: completion for process %d without an invocation", i, p);
          return TBC_ERR_BAD_HISTORY;
        }
        const uint32_t op = it->second;
        if (ev->type[i] == TBC_OK_) {          // complete: the invocation learns the value
          ret_pos[op] = i; a[op] = ev->a[i]; b[op] = ev->b[i];
        } else if (ev->type[i] == TBC_FAIL) {  // without-failures: it did not happen
          dead[op] = 1;
        } else {                               // :info -- open for ever, process retired
          retired[p] = true;
        }
        open_of.erase(it);
        break;
      }
      default:
        tbc::set_error("row %u: unknown :type code %u", i, (unsigned)ev->type[i]);
        return TBC_ERR_BAD_HISTORY;
    }
  }
  // pass 2: compact, renumber processes densely in order of first appearance
  std::unordered_map<int32_t, int32_t> dense;
  uint32_t k = 0;
  for (uint32_t i = 0; i < m; i++) {
    if (dead[i]) continue;
    auto it = dense.find(process[i]);
    int32_t d;
    if (it == dense.end()) { d = (int32_t)dense.size(); dense.emplace(process[i], d); }
    else d = it->second;
    f[k] = f[i]; a[k] = a[i]; b[k] = b[i]; process[k] = d;
    inv_pos[k] = inv_pos[i]; ret_pos[k] = ret_pos[i];
    k++;
  }
  *n_ops = k;
  *n_process = (uint32_t)dense.size();
  return TBC_OK;
}

// Composition of the level sweep's relations (jit_sweep.hip): the live set is a set of origin ids (<= 128) of the
// current segment; every slice that holds a live origin contributes the next segment's ids its live origins reach.
tbc_status tbc_sweep_compose(const tbc_sweep_rel* rel, uint32_t max_segs, uint32_t n_completions, tbc_sweep_verdict* out) {
  if (!rel || !out || max_segs == 0) { tbc::set_error("tbc_sweep_compose: null argument"); return TBC_ERR_INVALID_ARG; }
  std::memset(out, 0, sizeof *out);
  constexpr uint32_t SL = TBC_SWEEP_SLICES;
  uint32_t live[SL] = {1u, 0u, 0u, 0u};
  bool ended = false;
  out->valid = TBC_UNKNOWN;
  for (uint32_t k = 0; k < max_segs; k++) {
    const tbc_sweep_rel* g = rel + (size_t)k * SL;
    bool any = false;
    for (uint32_t j = 0; j < SL; j++) any = any || g[j].status != 0;
    if (!any) continue;                                  // no cut in this window
    uint32_t next[SL] = {0u, 0u, 0u, 0u}, reached = 0, F1 = 0;
    for (uint32_t j = 0; j < SL; j++) {
      if (g[j].status == 0) { if (live[j]) return TBC_OK; continue; }    // a live origin without its wavefront: unknown
      if (g[j].status != 1) return TBC_OK;                                // overflow: unknown, the caller falls back
      out->n_wavefronts++; out->probes += g[j].probes; out->configs_total += g[j].configs_total; out->subrounds += g[j].subrounds;
      if (g[j].max_level > out->max_level) out->max_level = g[j].max_level;
      F1 = g[j].F1; if (g[j].F0 > reached) reached = g[j].F0;
      if (g[j].n_end) out->end_state = g[j].end_state;
      for (uint32_t o = 0; o < 32; o++) if ((live[j] >> o) & 1u) {
        for (uint32_t w = 0; w < SL; w++) next[w] |= g[j].M[o][w];
        if (g[j].last_level[o] > reached) reached = g[j].last_level[o];
      }
    }
    if (!(next[0] | next[1] | next[2] | next[3])) {
      out->valid = TBC_INVALID; out->fail_level = reached; out->fail_seg = k;
      for (uint32_t w = 0; w < SL; w++) out->live_in[w] = live[w];
      return TBC_OK;
    }
    for (uint32_t w = 0; w < SL; w++) live[w] = next[w];
    ended = F1 == n_completions;
    if (ended) break;
  }
  if (ended) { out->valid = TBC_VALID; out->final_bits = live[0]; }
  return TBC_OK;
}

tbc_status tbc_memo_build(int64_t init_state, uint32_t n_classes, tbc_step_fn step, void* user,
                          uint32_t max_states, uint16_t* table, int64_t* handles, uint32_t* n_states) {
  if (!step || !table || !handles || !n_states || n_classes == 0 || max_states == 0 || max_states > 0xFFFEu) {
    tbc::set_error("tbc_memo_build: bad argument");
    return TBC_ERR_INVALID_ARG;
  }
  std::unordered_map<int64_t, uint32_t> id;
  handles[0] = init_state;
  id.emplace(init_state, 0u);
  uint32_t ns = 1;
  for (uint32_t s = 0; s < ns; s++) {          // breadth-first closure
    for (uint32_t c = 0; c < n_classes; c++) {
      int64_t nx = step(handles[s], c, user);
      uint16_t t = TBC_TABLE_INCONSISTENT;
      if (nx != -1) {
        auto it = id.find(nx);
        if (it == id.end()) {
          if (ns == max_states) {
            tbc::set_error("tbc_memo_build: more than %u reachable states", max_states);
            return TBC_ERR_MODEL;
          }
          handles[ns] = nx;
          id.emplace(nx, ns);
          t = (uint16_t)ns++;
        } else t = (uint16_t)it->second;
      }
      table[(size_t)s * n_classes + c] = t;
    }
  }
  *n_states = ns;
  return TBC_OK;
}

}  // extern "C"
