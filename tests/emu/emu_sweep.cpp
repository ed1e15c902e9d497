// TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  emu_sweep.cpp: runs K6w, the level sweep by workgroups
// (jepsen-tigerbeetle_amd/csrc/jit_sweep_wg_impl.h, the very file hipcc compiles into libtbcheck.so), on the CPU under the workgroup
// emulator of wave_env_wg_emu.h, on tables built here on the host from their definitions (host_tables.h) and cuts placed as
// sweep_cuts_kernel places them.  tests/test_sweep_wg_emu.py compares every record it leaves with oracle/sweep_ref.c.
#define TBC_EMU 1
#define __HIPCC__ 1
#include "wave_env_emu.h"
#include "../../jepsen-tigerbeetle_amd/csrc/jit_sweep_wg_impl.h"

using namespace tbc;

#include "host_tables.h"
#include "reach_table.h"

namespace {

struct Call { const SweepArgs* A; uint32_t* lds; };
template <uint32_t CAP, uint32_t NW, bool COMPACT, bool SOLO, bool RLX>
void entry(void* p, uint32_t) {
  auto* c = (Call*)p;
  sweepwg::segment<CAP, NW, COMPACT, SOLO, RLX>(*c->A, c->lds);
}
template <uint32_t CAP, uint32_t NW, bool COMPACT = false, bool SOLO = false, bool RLX = false>
void run_all(const SweepArgs& A, uint32_t n_wg, uint64_t seed) {
  std::vector<uint32_t> lds(sweepwg::lds_words<CAP, NW, COMPACT>() + 16);
  for (uint32_t w = 0; w < n_wg; w++) {
    std::fill(lds.begin(), lds.end(), 0xDEADBEEFu);          // LDS is not zeroed on the device either
    Call c{&A, lds.data()};
    wv::run_workgroup(&entry<CAP, NW, COMPACT, SOLO, RLX>, &c, (int)NW, w, seed + w);
  }
}

}  // namespace

extern "C" {

void emu_sweep_stats(uint64_t* out, int reset) { for (int i = 0; i < 64; i++) { out[i] = wv::stats()[i]; if (reset) wv::stats()[i] = 0; } }

// ONE history; out: max_segs * 4 tbc_sweep_rel records (zeroed here first) -- or, with seg_list (n_list (0, segment, slice) triples: the
// second pass over segments that overflowed), the first pass's records, of which only the listed ones are swept again
int emu_sweep_wg_run(uint32_t n, uint32_t n_process, const uint8_t* f, const int32_t* a, const int32_t* b, const int32_t* process,
                     const uint32_t* inv_pos, const uint32_t* ret_pos, uint32_t model_kind, int32_t init, uint32_t vpad, uint32_t rules,
                     uint32_t n_dom, uint32_t seg_target, uint32_t max_segs, uint32_t NW, uint32_t CAP, uint32_t queue, uint64_t seed, const uint32_t* seg_list, uint32_t n_list,
                     SegResult* out) {
  Tables T;
  const uint64_t op_off[2] = {0, n};
  // variant bit 4: the RELAXED sweep of a history with crashed calls -- the count form's tables (re-used slots, the crashed calls as classes),
  // no crashed call a candidate of its own (ncr all zero), the reach table from the classes (csrc/reach_table.h, the library's own builder)
  const bool relaxed = (queue & 16u) != 0u;
  if (!build_tables(1, op_off, &n_process, f, a, b, process, inv_pos, ret_pos, 1, vpad, 1, false, false, T, relaxed)) return 1;
  std::vector<uint32_t> reach, reach_hdr{0u, 0u};
  if (relaxed) {
    reach_hdr[1] = build_reach_table(T.cmem.data() + T.bh[0].cmem_off, T.bh[0].n_classes, reach);
    std::fill(T.ncr.begin(), T.ncr.end(), 0u);
  }
  queue &= ~16u;
  const uint32_t R = T.hist[0].n_ret;
  // plain read-mask rows (what K5 / K6 read): vpad words per front
  const uint32_t FS = front_stride(vpad, 1);
  std::vector<uint64_t> rows((uint64_t)(n + 1) * std::max(vpad, 1u) + 1, 0);
  for (uint32_t F = 0; F < R; F++) for (uint32_t v = 0; v < vpad; v++) rows[(uint64_t)F * vpad + v] = T.rdm[(uint64_t)F * FS + v];
  // cuts (jit_sweep.hip sweep_cuts_kernel): in window k the first front with the fewest calls open among those with no crashed call open
  uint32_t cut_open = 0;
  while (cut_open < 4 && (n_dom << (cut_open + 1)) <= 32 * kSweepSlices) cut_open++;
  std::vector<uint32_t> cuts(max_segs, kInf);
  for (uint32_t k = 0; k < max_segs; k++) {
    uint32_t cut = kInf;
    if (k == 0) cut = R != 0 ? 0u : kInf;
    else if (seg_target) {
      const uint64_t lo = (uint64_t)k * seg_target, hi = lo + seg_target < R ? lo + seg_target : R;
      uint32_t best = kInf;
      for (uint64_t F = lo; F < hi; F++) {
        const uint32_t no = T.off[F + 1] - T.off[F];
        if (T.ncr[F] == 0u && no < best) { best = no; cut = (uint32_t)F; }
      }
      if (best > cut_open) cut = kInf;
    }
    cuts[k] = cut;
  }
  if (!seg_list) memset(out, 0, sizeof(SegResult) * (size_t)max_segs * kSweepSlices);
  SweepArgs A{};
  A.hist = T.hist.data(); A.bh = T.bh.data(); A.off = T.off.data(); A.ncr = T.ncr.data(); A.lst = T.lst.data(); A.crashed = T.crashed.data();
  A.twn = (rules & kRuleTwin) ? T.twn.data() : nullptr; A.rdm = (rules & kRuleEager) ? rows.data() : nullptr; A.slot8 = T.slot8.data();
  A.cuts = cuts.data(); A.seg = out; A.table = nullptr; A.pool_vals = nullptr; A.n_hist = 1; A.max_segs = max_segs; A.seg_target = seg_target;
  A.cut_open = cut_open; A.n_dom = n_dom; A.vpad = vpad ? vpad : 1; A.rules = rules; A.model_kind = model_kind; A.init_state = init;
  A.shard_rank = 0; A.shard_world = 1; A.seg_list = seg_list; A.n_list = n_list; A.dump_cfg = nullptr; A.dump_count = nullptr;
  const uint32_t n_wg = seg_list ? n_list : max_segs * kSweepSlices;
  if (relaxed) {
    A.reach = reach.data(); A.reach_hdr = reach_hdr.data(); A.crashed = nullptr;
#define RUNR(C_, W_) if (CAP == C_ && NW == W_ && !queue) { run_all<C_, W_, false, false, true>(A, n_wg, seed); return 0; }
    RUNR(1024, 2) RUNR(1024, 8) RUNR(512, 4) RUNR(2048, 8)
#undef RUNR
    return 2;
  }
#define RUN(C_, W_) if (CAP == C_ && NW == W_ && !queue) { run_all<C_, W_>(A, n_wg, seed); return 0; }
  RUN(1024, 2) RUN(1024, 4) RUN(1024, 8) RUN(512, 4) RUN(2048, 8) RUN(2048, 16)
#undef RUN
  // variant bit 2: the compact walk (COMPACT); bit 3: + narrow passes by wavefront 0 alone (SOLO) -- the library's first pass since round 5
  // (bits 0 and 1 were the ring and the fingerprint forms: measured slower on the device, deleted)
#define RUNC(C_, W_) if (CAP == C_ && NW == W_ && queue == 4) { run_all<C_, W_, true>(A, n_wg, seed); return 0; } \
                     if (CAP == C_ && NW == W_ && queue == 12) { run_all<C_, W_, true, true>(A, n_wg, seed); return 0; }
  RUNC(1024, 2) RUNC(1024, 4) RUNC(1024, 8) RUNC(512, 2) RUNC(512, 4) RUNC(2048, 8) RUNC(2048, 16)
#undef RUNC
  return 2;
}

}  // extern "C"
