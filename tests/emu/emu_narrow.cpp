// TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  emu_narrow.cpp: runs the narrow search kernel's body
// (jepsen-tigerbeetle_amd/csrc/wgl_narrow_impl.h, the very file hipcc compiles into libtbcheck.so) on the CPU under
// the wavefront emulator of wave_env_emu.h, on tables built here ON THE HOST FROM THEIR DEFINITIONS
// (csrc/tbc_internal.h, csrc/pack_open.hip's header): a second, independent formulation of what pack_kernel /
// open_counts / open_walk / open_dprod leave in HBM.  tests/test_narrow_emu.py compares the results with the oracle
// (oracle/wgl_beam.c at one config per iteration and L pairs per round).  Built by tests/emu/build.py with g++;
// nothing under jepsen-tigerbeetle_amd/ links or loads it.
#define TBC_EMU 1
#define __HIPCC__ 1          // tbc_internal.h's look_val / look_need / look_prod / rec_cls helpers
#include "wave_env_emu.h"
#include "../../jepsen-tigerbeetle_amd/csrc/wgl_narrow_impl.h"
#include "../../jepsen-tigerbeetle_amd/csrc/witness_expand.h"

using namespace tbc;

#include "host_tables.h"

namespace {

template <int MW, int L>
struct WaveCall { const BeamArgs* A; uint32_t wave; uint32_t* lds; };
template <int MW, int L, bool CF>
void wave_entry(void* p, uint32_t lane) {
  auto* c = (WaveCall<MW, L>*)p;
  narrow::narrow_wave<MW, L, CF>(*c->A, c->wave, c->lds, lane);
}
template <int MW, int L, bool CF>
void wave_entry_cnt(void* p, uint32_t lane) {
  auto* c = (WaveCall<MW, L>*)p;
  narrow::narrow_wave<MW, L, CF, true>(*c->A, c->wave, c->lds, lane);
}
template <int MW, int L>
void run_all(BeamArgs& A, uint32_t max_waves) {
  const uint32_t H = 64 / L;
  uint32_t waves = (A.n_work + H - 1) / H;
  if (max_waves && waves > max_waves) waves = max_waves;      // fewer wavefronts than the batch needs: groups take more work as they finish
  static unsigned int next_work;
  next_work = 0;
  A.first_dynamic = waves * H; A.next_work = &next_work;
  const bool cf = MW == 1 && A.front_words == kFrontCompactWords;
  const bool cnt = (A.rules & kRuleCount) != 0u;
  std::vector<uint32_t> lds(narrow::narrow_lds_words(MW, L, cf, cnt) + 16);
  for (uint32_t w = 0; w < waves; w++) {
    std::fill(lds.begin(), lds.end(), 0xDEADBEEFu);          // LDS is not zeroed on the device either
    WaveCall<MW, L> c{&A, w, lds.data()};
    if constexpr (MW <= 2 && L >= 8) {
      if (cnt) {                                             // the count form's instantiations (wgl_narrow.hip launch_one)
        if constexpr (MW == 1) { if (cf) { wv::run_wave(&wave_entry_cnt<MW, L, true>, &c); continue; } }
        wv::run_wave(&wave_entry_cnt<MW, L, false>, &c);
        continue;
      }
    }
    if constexpr (MW == 1) { if (cf) { wv::run_wave(&wave_entry<MW, L, true>, &c); continue; } }
    wv::run_wave(&wave_entry<MW, L, false>, &c);
  }
}

}  // namespace

extern "C" {

// the product's own replay of a chain of branching calls into the whole witness (csrc/witness_expand.h, what tbc_api.hip's
// expand_eager_witness runs on the columns it copies back); returns the witness's length, -1 for a chain it refuses
int64_t emu_expand_witness(uint32_t n, const uint8_t* f, const int32_t* a, const int32_t* b, const int32_t* slot, const uint32_t* inv_pos,
                           const uint32_t* ret_pos, uint32_t n_slots, int32_t init, uint32_t branch, uint32_t by_completion,
                           const uint32_t* chain, uint32_t chain_len, uint32_t* out) {
  std::vector<uint32_t> w;
  if (!tbc::expand_eager_chain(n, f, a, b, slot, inv_pos, ret_pos, n_slots, init, branch != 0, by_completion != 0, chain, chain_len, w)) return -1;
  std::copy(w.begin(), w.end(), out);
  return (int64_t)w.size();
}

void emu_stats(uint64_t* out, int reset) { for (int i = 0; i < 64; i++) { out[i] = wv::stats()[i]; if (reset) wv::stats()[i] = 0; } }

// tables only (for comparing with what the device kernels wrote): returns sizes through out_n[8] and copies into caller buffers when given
int emu_narrow_run(uint32_t nh, const uint64_t* op_off, const uint32_t* n_process, const uint8_t* f, const int32_t* a, const int32_t* b,
                   const int32_t* process, const uint32_t* inv_pos, const uint32_t* ret_pos, uint32_t model_kind, int32_t init,
                   uint32_t L, uint32_t MW, uint32_t rules, uint32_t vpad, uint32_t lookahead, uint32_t entries_per_op, uint64_t max_steps,
                   uint64_t pool_words, uint32_t want_witness, uint32_t max_waves, uint32_t want_compact, uint32_t epochs, uint32_t count, uint32_t relaxed, const uint32_t* targets,
                   DevResult* results, uint32_t* witness, uint64_t* cfg_out) {
  Tables T;
  // the compact front records where libtbcheck would take them (tbc_api.hip front_words()): eager rule, one mask word, values <= 4
  uint32_t n_dom = 0;
  if (vpad) { int32_t vmax = init == TBC_NIL ? -1 : init; for (uint64_t i = 0; i < op_off[nh]; i++) { if (a[i] != TBC_NIL && a[i] > vmax) vmax = a[i]; if (f[i] == TBC_F_CAS && b[i] > vmax) vmax = b[i]; } n_dom = (uint32_t)(vmax + 2); }
  const bool compact = (want_compact & 1u) && (rules & kRuleEager) && front_compact_ok(n_dom, MW);
  if (want_compact & (2u | 8u)) return 4;          // (bits 1 and 3 were the lean tables and the lazy lookahead: measured slower on the device in round 5, deleted)
  // want_compact bit 2: the fronts' lists in order of completion (tbc_opts.list_order; the kernel reads what it is given)
  if (!build_tables(nh, op_off, n_process, f, a, b, process, inv_pos, ret_pos, MW, vpad, entries_per_op, (rules & kRuleBranch) != 0, compact, T, count != 0, (want_compact >> 16) ? (want_compact >> 16) : (want_compact & 16u) ? 2u : (want_compact & 4u) ? 1u : 0u)) return 1;      // (16: in order of completion with the writes last, list_order 2)
  if (count) { rules |= kRuleCount; for (uint32_t h = 0; h < nh && targets; h++) T.bh[h].target = targets[h]; }
  const uint64_t total = op_off[nh];
  uint64_t entries = 0;
  for (uint32_t h = 0; h < nh; h++) entries += 1ull << T.bh[h].tab_log2;
  // words per entry of the arena: keys and parent links -- or, as in libtbcheck, keys only when nobody wants a witness (tbc_api.hip
  // tab_stride()); guard words behind the arena catch a kernel that writes a link all the same
  const uint32_t EW = (want_witness ? MW + 2 : MW + 1) + (count ? kCountWords : 0u);
  std::vector<uint64_t> tab(entries * EW + 64, 0), pool(pool_words + 1, 0), cfg((uint64_t)nh * kCfgCap * (2 + MW) + 1, 0);
  std::vector<uint32_t> stack(entries + 1, 0), dstack(entries + 1, 0), work(nh), wit(total + 1, 0);
  unsigned long long cursor = 0;
  for (uint32_t h = 0; h < nh; h++) work[h] = h;
  std::vector<DevResult> res(nh);
  memset(res.data(), 0xFF, nh * sizeof(DevResult));
  BeamArgs A{};
  A.hist = T.hist.data(); A.bh = T.bh.data(); A.off = T.off.data(); A.ncr = T.ncr.data(); A.lst = T.lst.data(); A.crashed = T.crashed.data();
  A.look = lookahead ? T.look.data() : nullptr; A.slot8 = T.slot8.data(); A.ret_slot = T.ret_slot.data(); A.ret_op = T.ret_op.data();
  A.stack = stack.data(); A.dstack = lookahead ? dstack.data() : nullptr; A.tab = tab.data(); A.results = res.data();
  A.witness = want_witness ? wit.data() : nullptr; A.work = work.data(); A.table = nullptr; A.n_work = nh; A.model_kind = model_kind;
  A.init_state = init; A.width = 1; A.tab_stride = EW; A.cmem = T.cmem.empty() ? nullptr : T.cmem.data(); A.count_mode = relaxed ? kCountRelaxed : kCountExact; A.max_steps = max_steps; A.time_limit_ticks = 0; A.dbg = nullptr;
  A.pool = pool_words ? pool.data() : nullptr; A.pool_cursor = &cursor; A.pool_words = pool_words; A.max_tab_log2 = 28;
  A.pool_vals = nullptr; A.cfg = cfg.data(); A.rules = rules; A.twn = (rules & kRuleTwin) ? T.twn.data() : nullptr;
  A.stall_checks = (want_compact >> 8) & 0xFFu;          // want_compact bits 8..15: BeamArgs.stall_checks (a history that stops passing completions is stopped)
  A.rdm = T.rdm.data(); A.vpad = vpad; A.rk8 = T.rk8.data(); A.front_words = compact ? kFrontCompactWords : front_stride(vpad, MW);
#define RUN(MWV, LV) if (MW == MWV && L == LV) { run_all<MWV, LV>(A, max_waves); ran = true; }
  bool ran = false;
  // epochs > 0: that many passes over the SAME visited-set arena, never cleared in between, each under its own epoch tag
  // (what libtbcheck does instead of zeroing the arena before every pass); the last pass's results are returned
  for (uint32_t pass = 0; pass < (epochs ? epochs : 1u); pass++) {
    A.epoch = epochs ? pass + 1u : 0u;
    cursor = 0;
    std::fill(pool.begin(), pool.end(), 0ull);            // (the growth pool IS zeroed per pass: a tenth of the arena)
    memset(res.data(), 0xFF, nh * sizeof(DevResult));
    ran = false;
    RUN(1, 4) RUN(1, 8) RUN(1, 16) RUN(1, 32) RUN(2, 8) RUN(2, 16) RUN(4, 8) RUN(4, 16)
    if (!ran) return 2;
  }
#undef RUN
  for (uint64_t i = entries * EW; i < entries * EW + 64; i++) if (tab[i] != 0) return 3;       // something was written past the arena
  memcpy(results, res.data(), nh * sizeof(DevResult));
  if (want_witness && witness) memcpy(witness, wit.data(), total * 4);
  if (cfg_out) memcpy(cfg_out, cfg.data(), (uint64_t)nh * kCfgCap * (2 + MW) * 8);
  return 0;
}

}  // extern "C"
