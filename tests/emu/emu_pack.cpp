// TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  emu_pack.cpp: runs K1 for one history, pack by a workgroup's sixteen wavefronts
// (jepsen-tigerbeetle_amd/csrc/pack_one_impl.h, the very file hipcc compiles into libtbcheck.so), on the CPU under the workgroup
// emulator of wave_env_wg_emu.h and compares EVERY WORD it leaves behind -- records and sentinels, list starts, completion tables,
// the ranks and places in the scratch arena, n_ret, status -- with a restatement of what pack.hip's header defines, written here
// from those definitions (sorting, no bitmap, no counting sort).  tests/test_pack_one_emu.py drives it.
#define TBC_EMU 1
#define __HIPCC__ 1
#include "wave_env_emu.h"
#include "../../jepsen-tigerbeetle_amd/csrc/pack_one_impl.h"

#include <algorithm>
#include <vector>

using namespace tbc;

#include "host_tables.h"

namespace {

struct Call { const PackArgs* A; uint32_t* lds; const PackOpenArgs* O; };
void entry(void* p, uint32_t) {
  auto* c = (Call*)p;
  const PackOpenArgs none{};
  packone::history<packone::OneGeo>(*c->A, none, c->lds);
}
template <class G>
void entry_counts(void* p, uint32_t) {
  auto* c = (Call*)p;
  packone::history<G>(*c->A, *c->O, c->lds);
}

struct Want {
  uint32_t status = 0, n_ret = 0;
  bool tables = false;                     // records, list starts, completion tables and scratch are defined (status 0, or the one-open-op check failed)
  std::vector<Rec> rec; std::vector<uint32_t> seg, ret_slot, ret_op, scratch;
};

bool op_ok_host(uint32_t kind, uint32_t f, int32_t a, uint32_t n_classes) {
  switch (kind) {
    case TBC_MODEL_REGISTER: return f == TBC_F_READ || f == TBC_F_WRITE;
    case TBC_MODEL_CAS_REGISTER: return f == TBC_F_READ || f == TBC_F_WRITE || f == TBC_F_CAS;
    case TBC_MODEL_MUTEX: return f == TBC_F_ACQUIRE || f == TBC_F_RELEASE;
    case TBC_MODEL_TABLE: return f == TBC_F_CLASS && (uint32_t)a < n_classes;
    default: return false;
  }
}

// pack.hip's header, restated: what a history's pack leaves behind
void want_for(uint32_t n, uint32_t W, uint32_t E, bool cf, const uint8_t* f, const int32_t* a, const int32_t* b, const int32_t* proc,
              const uint32_t* inv, const uint32_t* ret, uint32_t kind, uint32_t n_classes, Want& w) {
  bool bad = false, model_bad = false;
  for (uint32_t i = 0; i < n; i++) {
    const bool slotless = cf && ret[i] == TBC_POS_CRASHED;
    bool rb = inv[i] >= E || (!slotless && (proc[i] < 0 || (uint32_t)proc[i] >= W)) || (i > 0 && inv[i - 1] >= inv[i]);
    if (ret[i] != TBC_POS_CRASHED) rb = rb || ret[i] <= inv[i] || ret[i] >= E;
    if (rb) { bad = true; continue; }
    if (!op_ok_host(kind, f[i], a[i], n_classes)) model_bad = true;
  }
  if (bad || model_bad) { w.status = model_bad ? (uint32_t)TBC_ERR_MODEL : (uint32_t)TBC_ERR_BAD_HISTORY; w.n_ret = 0; return; }
  std::vector<std::pair<uint32_t, uint32_t>> rets;
  for (uint32_t i = 0; i < n; i++) if (ret[i] != TBC_POS_CRASHED) rets.push_back({ret[i], i});
  std::sort(rets.begin(), rets.end());
  for (size_t r = 1; r < rets.size(); r++) if (rets[r].first == rets[r - 1].first) { w.status = (uint32_t)TBC_ERR_BAD_HISTORY; w.n_ret = 0; return; }
  const uint32_t R = (uint32_t)rets.size();
  w.n_ret = R; w.tables = true;
  std::vector<uint32_t> ret_rank(n, kInf), inv_rank(n, 0);
  w.ret_slot.assign(R, 0); w.ret_op.assign(R, 0);
  for (uint32_t r = 0; r < R; r++) { ret_rank[rets[r].second] = r; w.ret_op[r] = rets[r].second; w.ret_slot[r] = (uint32_t)proc[rets[r].second]; }
  for (uint32_t i = 0; i < n; i++) {
    uint32_t c = 0;
    c = (uint32_t)(std::lower_bound(rets.begin(), rets.end(), std::make_pair(inv[i], 0u)) - rets.begin());
    inv_rank[i] = c;
  }
  std::vector<std::vector<uint32_t>> lists(W);
  for (uint32_t i = 0; i < n; i++) if (!(cf && ret[i] == TBC_POS_CRASHED)) lists[(uint32_t)proc[i]].push_back(i);
  w.seg.assign(W + 1, 0);
  for (uint32_t p = 0; p < W; p++) w.seg[p + 1] = w.seg[p] + (uint32_t)lists[p].size() + 2;
  w.rec.assign(w.seg[W], Rec{});
  w.scratch.assign(3 * (size_t)n, 0);
  for (uint32_t i = 0; i < n; i++) { w.scratch[i] = inv_rank[i]; w.scratch[n + i] = ret_rank[i]; w.scratch[2 * (size_t)n + i] = kInf; }
  bool overlap = false;
  for (uint32_t p = 0; p < W; p++) {
    Rec hd{}; hd.inv_rank = 0; hd.ret_rank = 0; hd.opidx = kInf; hd.f = kFNone; hd.a = 0; hd.b = 0; hd.cls = 0; hd.prod = kLookNone;
    Rec tl = hd; tl.inv_rank = kInf; tl.ret_rank = kInf;
    w.rec[w.seg[p]] = hd; w.rec[w.seg[p + 1] - 1] = tl;
    for (size_t k = 0; k < lists[p].size(); k++) {
      const uint32_t i = lists[p][k];
      Rec r{}; r.inv_rank = inv_rank[i]; r.ret_rank = ret_rank[i]; r.opidx = i; r.f = f[i]; r.a = a[i]; r.b = b[i];
      r.cls = rec_cls(f[i], a[i], ret_rank[i] == kInf); r.prod = look_prod(f[i], a[i], b[i]);
      w.rec[w.seg[p] + 1 + k] = r;
      w.scratch[2 * (size_t)n + i] = w.seg[p] + 1 + (uint32_t)k;
      if (k > 0 && !(ret_rank[lists[p][k - 1]] < inv_rank[i])) overlap = true;      // (a crashed predecessor: kInf < x is false)
    }
  }
  w.status = overlap ? (uint32_t)TBC_ERR_BAD_HISTORY : 0u;
}

}  // namespace

extern "C" {

void emu_pack_stats(uint64_t* out, int reset) { for (int i = 0; i < 64; i++) { out[i] = wv::stats()[i]; if (reset) wv::stats()[i] = 0; } }

// nh histories (ops at op_off[h] .. op_off[h + 1]), one workgroup each, launched `per_launch` histories at a time (h0 moves).
// count != 0: the count form's inputs -- live calls on re-used slots, a crashed call slotless (the host's re-numbering, restated as
// in host_tables.h).  Returns 0 when every word agrees, else a code; diag = {history, index, got, want}.
int emu_pack_one_check(uint32_t nh, const uint64_t* op_off, const uint32_t* n_process, const uint32_t* n_events, const uint8_t* f, const int32_t* a,
                       const int32_t* b, const int32_t* process, const uint32_t* inv_pos, const uint32_t* ret_pos, uint32_t model_kind,
                       uint32_t n_classes, uint32_t count, uint32_t per_launch, uint64_t seed, uint64_t* diag) {
  const uint64_t total = op_off[nh];
  std::vector<int32_t> slot(process, process + total);
  std::vector<Hist> hist(nh);
  uint64_t rec_n = 0, seg_n = 0;
  for (uint32_t h = 0; h < nh; h++) {
    const uint64_t o = op_off[h];
    const uint32_t n = (uint32_t)(op_off[h + 1] - o);
    uint32_t W = n_process[h];
    if (count) {
      std::vector<int32_t> slot_of(n_process[h] + 1, -1);
      std::vector<uint8_t> used(n_process[h] + 2, 0);
      W = 1;
      for (uint32_t i = 0; i < n; i++) {
        const int32_t p = process[o + i];
        if (p < 0 || (uint32_t)p >= n_process[h]) continue;                 // (a bad row: left for the kernel to refuse)
        if (ret_pos[o + i] == TBC_POS_CRASHED) { if (slot_of[p] >= 0) { used[slot_of[p]] = 0; slot_of[p] = -1; } slot[o + i] = 0; continue; }
        if (slot_of[p] < 0) { uint32_t sl = 0; while (used[sl]) sl++; used[sl] = 1; slot_of[p] = (int32_t)sl; W = std::max(W, sl + 1); }
        slot[o + i] = slot_of[p];
      }
    }
    Hist& H = hist[h];
    H = Hist{};
    H.op_off = o; H.rec_off = rec_n; H.seg_off = seg_n; H.ret_off = o; H.bm_off = 0; H.frame_off = 3 * o;
    H.n_ops = n; H.n_events = n_events[h]; H.n_slots = W; H.n_ret = 0xABABABABu; H.status = 0xCDCDCDCDu; H.flags = count ? kHistCount : 0u;
    rec_n += (uint64_t)n + 2ull * W; seg_n += W + 1;
    if (!packone::fits(model_kind, n, H.n_events, W)) return 90;
  }
  std::vector<Rec> rec(rec_n);
  memset(rec.data(), 0xEE, rec.size() * sizeof(Rec));
  std::vector<uint32_t> seg(seg_n, 0xEEEEEEEEu), ret_slot(total, 0xEEEEEEEEu), ret_op(total, 0xEEEEEEEEu), scratch(3 * total, 0xEEEEEEEEu);
  PackArgs A{};
  A.hist = hist.data(); A.f = f; A.a = a; A.b = b; A.process = slot.data(); A.inv_pos = inv_pos; A.ret_pos = ret_pos;
  A.rec = rec.data(); A.seg = seg.data(); A.ret_slot = ret_slot.data(); A.ret_op = ret_op.data();
  A.bitmap = nullptr; A.wpre = nullptr;                    // the LDS tables replace them
  A.scratch = scratch.data(); A.frame_words = 3; A.n_hist = nh; A.model_kind = model_kind; A.n_classes = n_classes; A.dbg = nullptr;
  A.pool_vals = nullptr; A.pool_len = 0; A.n_keys = 0;
  std::vector<uint32_t> lds(packone::lds_words());
  if (per_launch == 0) per_launch = nh;
  for (uint32_t h0 = 0; h0 < nh; h0 += per_launch) {
    PackArgs L = A; L.h0 = h0; L.n_hist = std::min(nh, h0 + per_launch);
    for (uint32_t g = 0; g < L.n_hist - h0; g++) {
      std::fill(lds.begin(), lds.end(), 0xDEADBEEFu);        // LDS is not zeroed on the device either
      Call c{&L, lds.data(), nullptr};
      wv::run_workgroup(&entry, &c, (int)packone::kNW, g, seed + h0 + g);
    }
  }
  for (uint32_t h = 0; h < nh; h++) {
    const Hist& H = hist[h];
    const uint64_t o = H.op_off;
    const uint32_t n = H.n_ops, W = H.n_slots;
    Want w;
    want_for(n, W, H.n_events, count != 0, f + o, a + o, b + o, slot.data() + o, inv_pos + o, ret_pos + o, model_kind, n_classes, w);
#define MISMATCH(code, idx, got, want) do { diag[0] = h; diag[1] = (idx); diag[2] = (got); diag[3] = (want); return (code); } while (0)
    if (H.status != w.status) MISMATCH(1, 0, H.status, w.status);
    if (H.n_ret != w.n_ret) MISMATCH(2, 0, H.n_ret, w.n_ret);
    if (!w.tables) continue;
    for (uint32_t p = 0; p <= W; p++) if (seg[H.seg_off + p] != w.seg[p]) MISMATCH(3, p, seg[H.seg_off + p], w.seg[p]);
    for (uint32_t r = 0; r < w.n_ret; r++) {
      if (ret_slot[o + r] != w.ret_slot[r]) MISMATCH(4, r, ret_slot[o + r], w.ret_slot[r]);
      if (ret_op[o + r] != w.ret_op[r]) MISMATCH(5, r, ret_op[o + r], w.ret_op[r]);
    }
    for (uint32_t x = 0; x < 3 * n; x++) if (scratch[3 * o + x] != w.scratch[x]) MISMATCH(6, x, scratch[3 * o + x], w.scratch[x]);
    for (uint32_t x = 0; x < w.seg[W]; x++) {
      const uint32_t* g = reinterpret_cast<const uint32_t*>(&rec[H.rec_off + x]);
      const uint32_t* e = reinterpret_cast<const uint32_t*>(&w.rec[x]);
      for (uint32_t k = 0; k < 8; k++) if (g[k] != e[k]) MISMATCH(7, x * 8 + k, g[k], e[k]);
    }
#undef MISMATCH
  }
  return 0;
}

// THE BATCH FORM (pack_one_impl.h, BatchGeo: four wavefronts, pack + open counts).  As emu_pack_one_check for what pack leaves, and
// what open_counts_kernel leaves -- off[], ncr[], slot8, rk8, the crashed-call list, the lookahead records past the last rank,
// BeamHist.status / n_crashed / lst_need -- against host_tables.h (built from the definitions; valid histories) or, for a history
// pack refuses, against open_counts_kernel's "no lists".  flags: 1 = count form, 2 = branch lists, 4 = lookahead records asked for,
// 8 = rk8 asked for.  lst_cap: the room of every history's list arena (0 = plenty).  Codes 1-7 as above, 10 off, 11 ncr, 12 slot8,
// 13 rk8, 14 crashed, 15 BeamHist.status, 16 n_crashed, 17 lst_need, 18 look.
}  // extern "C"

template <class G>
static int pack_counts_check(uint32_t nh, const uint64_t* op_off, const uint32_t* n_process, const uint32_t* n_events, const uint8_t* f, const int32_t* a,
                      const int32_t* b, const int32_t* process, const uint32_t* inv_pos, const uint32_t* ret_pos, uint32_t model_kind,
                      uint32_t n_classes, uint32_t vpad, uint32_t flags, uint32_t lst_cap, uint32_t per_launch, uint64_t seed, uint64_t* diag) {
  const bool count = flags & 1u, branch = flags & 2u, want_look = flags & 4u, want_rk8 = (flags & 8u) || branch;
  if (flags & 32u) return 9;          // (was: the lean lookahead records -- deleted in round 5)
  const uint64_t total = op_off[nh];
  std::vector<int32_t> slot(process, process + total);
  std::vector<Hist> hist(nh);
  std::vector<BeamHist> bh(nh);
  uint64_t rec_n = 0, seg_n = 0, off_n = 0;
  for (uint32_t h = 0; h < nh; h++) {
    const uint64_t o = op_off[h];
    const uint32_t n = (uint32_t)(op_off[h + 1] - o);
    uint32_t W = n_process[h];
    if (count) {
      std::vector<int32_t> slot_of(n_process[h] + 1, -1);
      std::vector<uint8_t> used(n_process[h] + 2, 0);
      W = 1;
      for (uint32_t i = 0; i < n; i++) {
        const int32_t p = process[o + i];
        if (p < 0 || (uint32_t)p >= n_process[h]) continue;
        if (ret_pos[o + i] == TBC_POS_CRASHED) { if (slot_of[p] >= 0) { used[slot_of[p]] = 0; slot_of[p] = -1; } slot[o + i] = 0; continue; }
        if (slot_of[p] < 0) { uint32_t sl = 0; while (used[sl]) sl++; used[sl] = 1; slot_of[p] = (int32_t)sl; W = std::max(W, sl + 1); }
        slot[o + i] = slot_of[p];
      }
    }
    Hist& H = hist[h];
    H = Hist{};
    H.op_off = o; H.rec_off = rec_n; H.seg_off = seg_n; H.ret_off = o; H.bm_off = 0; H.frame_off = 3 * o;
    H.n_ops = n; H.n_events = n_events[h]; H.n_slots = W; H.n_ret = 0xABABABABu; H.status = 0xCDCDCDCDu; H.flags = count ? kHistCount : 0u;
    BeamHist& B = bh[h];
    B = BeamHist{};
    B.off_off = off_n; B.lst_cap = lst_cap ? lst_cap : 0xFFFFFFFFu; B.status = 0xCDCDCDCDu; B.n_crashed = 0xABABABABu; B.lst_need = 0xEFEFEFEFu;
    rec_n += (uint64_t)n + 2ull * W; seg_n += W + 1; off_n += (uint64_t)n + 2;
    if (!G::fits(model_kind, n, H.n_events, W)) return 90;
  }
  std::vector<Rec> rec(rec_n);
  memset(rec.data(), 0xEE, rec.size() * sizeof(Rec));
  std::vector<uint32_t> seg(seg_n, 0xEEEEEEEEu), ret_slot(total, 0xEEEEEEEEu), ret_op(total, 0xEEEEEEEEu), scratch(3 * total, 0xEEEEEEEEu);
  std::vector<uint32_t> off(off_n + 1, 0u), ncr(off_n + 1, 0u);                 // zeroed by the host before the launch
  std::vector<OpRec> crashed(total + 1);
  memset(crashed.data(), 0xEE, crashed.size() * sizeof(OpRec));
  std::vector<uint8_t> slot8(slot8_bytes(total, nh), 0xEE), rk8(slot8_bytes(total, nh), 0xEE);
  std::vector<uint64_t> look(look_words(total, nh, 1), 0xEEEEEEEEEEEEEEEEull);
  PackArgs A{};
  A.hist = hist.data(); A.f = f; A.a = a; A.b = b; A.process = slot.data(); A.inv_pos = inv_pos; A.ret_pos = ret_pos;
  A.rec = rec.data(); A.seg = seg.data(); A.ret_slot = ret_slot.data(); A.ret_op = ret_op.data();
  A.bitmap = nullptr; A.wpre = nullptr;
  A.scratch = scratch.data(); A.frame_words = 3; A.n_hist = nh; A.model_kind = model_kind; A.n_classes = n_classes; A.dbg = nullptr;
  A.pool_vals = nullptr; A.pool_len = 0; A.n_keys = 0;
  PackOpenArgs O{};
  O.hist = hist.data(); O.bh = bh.data(); O.f = f; O.a = a; O.b = b; O.process = slot.data(); O.scratch = scratch.data(); O.rec = rec.data(); O.seg = seg.data();
  O.branch_lists = branch ? 1u : 0u; O.off = off.data(); O.ncr = ncr.data(); O.crashed = crashed.data(); O.ret_slot = ret_slot.data(); O.ret_op = ret_op.data();
  O.look = want_look ? look.data() : nullptr; O.slot8 = slot8.data(); O.rk8 = want_rk8 ? rk8.data() : nullptr; O.n_hist = nh; O.mask_words = 1; O.vpad = vpad;
  std::vector<uint32_t> lds(G::lds_words());
  if (per_launch == 0) per_launch = nh;
  for (uint32_t h0 = 0; h0 < nh; h0 += per_launch) {
    PackArgs L = A; L.h0 = h0; L.n_hist = std::min(nh, h0 + per_launch);
    PackOpenArgs LO = O; LO.h0 = L.h0; LO.n_hist = L.n_hist;
    for (uint32_t g = 0; g < L.n_hist - h0; g++) {
      std::fill(lds.begin(), lds.end(), 0xDEADBEEFu);
      Call c{&L, lds.data(), &LO};
      wv::run_workgroup(&entry_counts<G>, &c, (int)G::kNW, g, seed + h0 + g);
    }
  }
#define MISMATCH(code, idx, got, want) do { diag[0] = h; diag[1] = (idx); diag[2] = (got); diag[3] = (want); return (code); } while (0)
  for (uint32_t h = 0; h < nh; h++) {
    const Hist& H = hist[h];
    const BeamHist& B = bh[h];
    const uint64_t o = H.op_off;
    const uint32_t n = H.n_ops, W = H.n_slots;
    Want w;
    want_for(n, W, H.n_events, count, f + o, a + o, b + o, slot.data() + o, inv_pos + o, ret_pos + o, model_kind, n_classes, w);
    if (H.status != w.status) MISMATCH(1, 0, H.status, w.status);
    if (H.n_ret != w.n_ret) MISMATCH(2, 0, H.n_ret, w.n_ret);
    if (w.tables) {
      for (uint32_t p = 0; p <= W; p++) if (seg[H.seg_off + p] != w.seg[p]) MISMATCH(3, p, seg[H.seg_off + p], w.seg[p]);
      for (uint32_t r = 0; r < w.n_ret; r++) {
        if (ret_slot[o + r] != w.ret_slot[r]) MISMATCH(4, r, ret_slot[o + r], w.ret_slot[r]);
        if (ret_op[o + r] != w.ret_op[r]) MISMATCH(5, r, ret_op[o + r], w.ret_op[r]);
      }
      for (uint32_t x = 0; x < 3 * n; x++) if (scratch[3 * o + x] != w.scratch[x]) MISMATCH(6, x, scratch[3 * o + x], w.scratch[x]);
      for (uint32_t x = 0; x < w.seg[W]; x++) {
        const uint32_t* g = reinterpret_cast<const uint32_t*>(&rec[H.rec_off + x]);
        const uint32_t* e = reinterpret_cast<const uint32_t*>(&w.rec[x]);
        for (uint32_t k = 0; k < 8; k++) if (g[k] != e[k]) MISMATCH(7, x * 8 + k, g[k], e[k]);
      }
    }
    // ---- what open_counts_kernel leaves
    if (w.status != 0 || w.n_ret == 0) {
      if (B.status != 0) MISMATCH(15, 0, B.status, 0);
      if (B.n_crashed != 0) MISMATCH(16, 0, B.n_crashed, 0);
      if (B.lst_need != 0) MISMATCH(17, 0, B.lst_need, 0);
      continue;
    }
    // the definitions, for this history alone (the ORIGINAL process column: build_tables re-numbers the slots of the count form itself)
    Tables T;
    const uint64_t oo[2] = {0, n};
    const uint32_t np1[1] = {n_process[h]};
    if (!build_tables(1, oo, np1, f + o, a + o, b + o, process + o, inv_pos + o, ret_pos + o, G::kMaxW > 256 ? 16 : 4, vpad, 1, branch, false, T, count)) return 91;
    const uint32_t R = w.n_ret;
    const uint32_t need = T.off[R];
    if (B.lst_need != need) MISMATCH(17, 0, B.lst_need, need);
    if (need > B.lst_cap) {
      if (B.status != 1) MISMATCH(15, 1, B.status, 1);
      if (B.n_crashed != 0) MISMATCH(16, 1, B.n_crashed, 0);
      continue;
    }
    if (B.status != 0) MISMATCH(15, 2, B.status, 0);
    for (uint32_t r = 0; r <= R; r++) if (off[B.off_off + r] != T.off[r]) MISMATCH(10, r, off[B.off_off + r], T.off[r]);
    for (uint32_t r = R + 1; r < n + 2; r++) if (off[B.off_off + r] != 0) MISMATCH(10, r, off[B.off_off + r], 0);
    for (uint32_t r = 0; r < R; r++) {
      const uint32_t want_ncr = count ? 0u : T.ncr[r];                 // (count form: count_fronts_kernel's, not this kernel's)
      if (ncr[B.off_off + r] != want_ncr) MISMATCH(11, r, ncr[B.off_off + r], want_ncr);
    }
    const uint8_t* s8 = slot8.data() + slot8_off(o, h);
    const uint8_t* k8 = rk8.data() + slot8_off(o, h);
    const uint8_t* ts8 = T.slot8.data() + slot8_off(0, 0);
    const uint8_t* tk8 = T.rk8.data() + slot8_off(0, 0);
    for (uint32_t r = 0; r < R + 16; r++) {
      if (s8[r] != ts8[r]) MISMATCH(12, r, s8[r], ts8[r]);
      if (want_rk8 && k8[r] != tk8[r]) MISMATCH(13, r, k8[r], tk8[r]);
    }
    const uint32_t want_crashed = count ? 0u : T.bh[0].n_crashed;
    if (B.n_crashed != want_crashed) MISMATCH(16, 2, B.n_crashed, want_crashed);
    for (uint32_t k = 0; k < want_crashed; k++) {
      const uint32_t* g = reinterpret_cast<const uint32_t*>(&crashed[o + k]);
      const uint32_t* e = reinterpret_cast<const uint32_t*>(&T.crashed[k]);
      for (uint32_t q = 0; q < 4; q++) if (g[q] != e[q]) MISMATCH(14, k * 4 + q, g[q], e[q]);
    }
    if (want_look) {
      const uint64_t* lk = look.data() + look_off(o, h, 1);
      for (uint32_t t = R; t < R + kLookPad; t++) {
        const uint64_t w0 = (uint64_t)(kLookNone << 16 | kLookNone << 24) | (255ull << 32) | (255ull << 40);
        if (lk[2 * (uint64_t)t] != w0) MISMATCH(18, t, lk[2 * (uint64_t)t], w0);
        if (lk[2 * (uint64_t)t + 1] != 0) MISMATCH(18, t, lk[2 * (uint64_t)t + 1], 0);
      }
    }
  }
#undef MISMATCH
  return 0;
}

extern "C" {

// flags bit 16: the one-history geometry with counts (OneCountsGeo, sixteen wavefronts) instead of the batch geometry; bit 64: the batch
// geometry for at most 64 process slots (Batch64Geo: a byte per histogram entry)
int emu_pack_wg_check(uint32_t nh, const uint64_t* op_off, const uint32_t* n_process, const uint32_t* n_events, const uint8_t* f, const int32_t* a,
                      const int32_t* b, const int32_t* process, const uint32_t* inv_pos, const uint32_t* ret_pos, uint32_t model_kind,
                      uint32_t n_classes, uint32_t vpad, uint32_t flags, uint32_t lst_cap, uint32_t per_launch, uint64_t seed, uint64_t* diag) {
  if (flags & 64u) return pack_counts_check<packone::Batch64Geo>(nh, op_off, n_process, n_events, f, a, b, process, inv_pos, ret_pos, model_kind, n_classes, vpad, flags, lst_cap, per_launch, seed, diag);
  if (flags & 16u) return pack_counts_check<packone::OneCountsGeo>(nh, op_off, n_process, n_events, f, a, b, process, inv_pos, ret_pos, model_kind, n_classes, vpad, flags, lst_cap, per_launch, seed, diag);
  return pack_counts_check<packone::BatchGeo>(nh, op_off, n_process, n_events, f, a, b, process, inv_pos, ret_pos, model_kind, n_classes, vpad, flags, lst_cap, per_launch, seed, diag);
}

}  // extern "C"
