// TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  emu_walk.cpp: runs the front walk with lane = front
// (jepsen-tigerbeetle_amd/csrc/open_walk_impl.h, the very file hipcc compiles into libtbcheck.so) on the CPU under the
// wavefront emulator of wave_env_emu.h and compares EVERY WORD it writes -- the fronts' lists, the twin masks, the open-read
// rows / whole compact front records, the lookahead records with their producer distances -- with the tables host_tables.h builds from the definitions.  Its inputs
// (pack_kernel's per-slot record lists with their sentinels, the ranks, off[] / ret_slot / ret_op) are built here the way
// pack.hip and open_counts_kernel leave them.  Built by tests/emu with g++; nothing under jepsen-tigerbeetle_amd/ links it.
#define TBC_EMU 1
#define __HIPCC__ 1          // tbc_internal.h's look_val / look_need / look_prod / rec_cls helpers
#include "wave_env_emu.h"
#include "../../jepsen-tigerbeetle_amd/csrc/open_walk_impl.h"

using namespace tbc;

#include "host_tables.h"

namespace {

struct WalkCall { const PackOpenArgs* A; uint32_t wave; uint32_t* lds; };
template <int VCAP>
void walk_entry(void* p, uint32_t lane) {
  auto* c = (WalkCall*)p;
  walk::walk_wave<VCAP>(*c->A, c->wave, c->lds, lane);
}

}  // namespace

extern "C" {

// flags: 1 = twin masks, 2 = lookahead records, 4 = branch lists, 8 = front records (else plain rows), 16 = the compact form.
// Returns 0 when every word agrees; else a code (1 lst, 2 twn, 3 rdm, 4 look word 0, 5 look mask, 6 tmp, 9 bad input) with
// diag = {history, front, index, got, want}.
int emu_walk_check(uint32_t nh, const uint64_t* op_off, const uint32_t* n_process, const uint8_t* f, const int32_t* a, const int32_t* b,
                   const int32_t* process, const uint32_t* inv_pos, const uint32_t* ret_pos, uint32_t vpad, uint32_t flags, uint64_t* diag) {
  const bool want_twn = flags & 1u, want_look = flags & 2u, branch = flags & 4u, records = flags & 8u, compact = (flags & 16u) && records;
  if (flags & 32u) return 9;          // (was: the lean formats -- deleted in round 5)
  const uint32_t by_ret = (flags >> 16) ? (flags >> 16) : (flags & 128u) ? 2u : (flags & 64u) ? 1u : 0u;      // (bits 16 up: list_order 16 + W)          // a front's list in order of completion (PackOpenArgs.list_order = 1; 2: the writes last)
  Tables T;
  if (!build_tables(nh, op_off, n_process, f, a, b, process, inv_pos, ret_pos, 1, vpad, 1, branch, compact, T, false, by_ret)) return 9;
  const uint64_t total = op_off[nh];
  const uint32_t FS = compact ? kFrontCompactWords : front_stride(vpad, 1);      // the reference's rows (always front records)
  const uint32_t FW = records ? FS : vpad;                                       // the walk's
  // pack_kernel's record lists: per slot a head sentinel, the slot's calls in op order, a tail sentinel
  std::vector<Hist> hist = T.hist;
  std::vector<Rec> rec;
  std::vector<uint32_t> seg, scratch(2 * total + 2, 0);
  uint32_t max_ops = 1;
  for (uint32_t h = 0; h < nh; h++) {
    const uint64_t o = op_off[h];
    const uint32_t n = (uint32_t)(op_off[h + 1] - o), W = n_process[h];
    max_ops = std::max(max_ops, n);
    hist[h].rec_off = rec.size(); hist[h].seg_off = seg.size(); hist[h].frame_off = 2 * o;
    for (uint32_t i = 0; i < n; i++) { scratch[2 * o + i] = T.inv_rank[o + i]; scratch[2 * o + n + i] = T.ret_rank[o + i]; }
    const uint64_t r0 = rec.size();
    for (uint32_t p = 0; p < W; p++) {
      seg.push_back((uint32_t)(rec.size() - r0));
      Rec hd; hd.inv_rank = 0; hd.ret_rank = 0; hd.opidx = kInf; hd.f = kFNone; hd.a = 0; hd.b = 0; hd.cls = 0; hd.prod = kLookNone;
      rec.push_back(hd);
      for (uint32_t i = 0; i < n; i++) if ((uint32_t)process[o + i] == p) {
        Rec r; r.inv_rank = T.inv_rank[o + i]; r.ret_rank = T.ret_rank[o + i]; r.opidx = i; r.f = f[o + i]; r.a = a[o + i]; r.b = b[o + i];
        r.cls = rec_cls(r.f, r.a, r.ret_rank == kInf); r.prod = look_prod(r.f, r.a, r.b);
        rec.push_back(r);
      }
      Rec tl = hd; tl.inv_rank = kInf; tl.ret_rank = kInf;
      rec.push_back(tl);
    }
    seg.push_back((uint32_t)(rec.size() - r0));
  }
  const uint64_t kPoison = 0xA5A5A5A5A5A5A5A5ull;
  std::vector<OpRec> lst(T.lst.size() + 1, OpRec{0xA5A5A5A5u, 0xA5A5A5A5u, (int32_t)0xA5A5A5A5, (int32_t)0xA5A5A5A5});
  std::vector<uint64_t> twn(T.twn.size() + 1, kPoison), rdm(total * (FW ? FW : 1) + 1, kPoison), look(T.look.size() + 1, kPoison);
  std::vector<uint32_t> tmp(total + 1, 0xA5A5A5A5u);
  PackOpenArgs A{};
  A.hist = hist.data(); A.bh = T.bh.data(); A.f = f; A.a = a; A.b = b; A.process = process; A.scratch = scratch.data();
  A.rec = rec.data(); A.seg = seg.data(); A.chunks_per_hist = (max_ops + 63) / 64; A.branch_lists = branch ? 1u : 0u;
  A.off = T.off.data(); A.ncr = T.ncr.data(); A.lst = lst.data(); A.crashed = nullptr; A.ret_slot = T.ret_slot.data(); A.ret_op = T.ret_op.data();
  A.look = want_look ? look.data() : nullptr; A.tmp = want_look ? tmp.data() : nullptr; A.slot8 = nullptr;
  A.front_words = records ? FS : 0u; A.front_compact = compact ? 1u : 0u; A.rk8 = nullptr; A.n_hist = nh; A.mask_words = 1;
  A.twn = want_twn ? twn.data() : nullptr; A.rdm = vpad ? rdm.data() : nullptr; A.vpad = vpad; A.h0 = 0; A.list_order = by_ret;
  std::vector<uint32_t> lds(walk::walk_lds_words() + 16);
  for (uint32_t w = 0; w < nh * A.chunks_per_hist; w++) {
    std::fill(lds.begin(), lds.end(), 0xDEADBEEFu);
    WalkCall c{&A, w, lds.data()};
    if (vpad <= 8) wv::run_wave(&walk_entry<8>, &c); else wv::run_wave(&walk_entry<32>, &c);
  }
  auto fail = [&](int code, uint64_t h, uint64_t F, uint64_t i, uint64_t got, uint64_t want) { diag[0] = h; diag[1] = F; diag[2] = i; diag[3] = got; diag[4] = want; return code; };
  for (uint32_t h = 0; h < nh; h++) {
    const uint64_t o = op_off[h];
    const uint32_t R = T.hist[h].n_ret;
    const uint32_t* off = T.off.data() + T.bh[h].off_off;
    const uint64_t l0 = T.bh[h].lst_off;
    for (uint32_t F = 0; F < R; F++) {
      for (uint32_t i = off[F]; i < off[F + 1]; i++) {
        const OpRec &g = lst[l0 + i], &w = T.lst[l0 + i];
        if (g.op != w.op) return fail(1, h, F, i - off[F], g.op, w.op);
        if (g.f_slot != w.f_slot) return fail(1, h, F, i - off[F], g.f_slot, w.f_slot);
        if (g.a != w.a || g.b != w.b) return fail(1, h, F, i - off[F], (uint32_t)g.a, (uint32_t)w.a);
        if (want_twn && twn[l0 + i] != T.twn[l0 + i]) return fail(2, h, F, i - off[F], twn[l0 + i], T.twn[l0 + i]);
      }
      for (uint32_t v = 0; v < vpad; v++) {
        const uint64_t want = T.rdm[(o + F) * FS + v];      // (a compact record whole: words 6, 7 are the list location and the window of the next ranks)
        if (rdm[(o + F) * FW + v] != want) return fail(3, h, F, v, rdm[(o + F) * FW + v], want);
      }
      if (want_look) {
        const uint64_t lo = look_off(o, h, 1);
        const uint64_t w0 = T.look[lo + (uint64_t)F * 2];                                  // (with the producer distance)
        if (look[lo + (uint64_t)F * 2] != w0) return fail(4, h, F, 0, look[lo + (uint64_t)F * 2], w0);
        if (look[lo + (uint64_t)F * 2 + 1] != T.look[lo + (uint64_t)F * 2 + 1]) return fail(5, h, F, 0, look[lo + (uint64_t)F * 2 + 1], T.look[lo + (uint64_t)F * 2 + 1]);
      }
    }
    // nothing past the history's last list entry
    if (lst[l0 + off[R]].op != 0xA5A5A5A5u && (h + 1 == nh || T.bh[h + 1].lst_off != l0 + off[R])) return fail(1, h, R, 0, lst[l0 + off[R]].op, 0xA5A5A5A5u);
  }
  return 0;
}

}  // extern "C"
