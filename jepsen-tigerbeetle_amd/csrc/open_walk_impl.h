// open_walk_impl.h -- K1b's front walk with LANE = FRONT (gfx950; mask_words == 1, i.e. at most 64 process slots).
//
// What it writes is pack_open.hip's open_walk_kernel<1> word for word (lst / twn / rdm rows / look / tmp; the definitions are
// in that file's header and tbc_internal.h).  How it gets there is turned round.  open_walk_kernel gives a wavefront 64 fronts
// and walks them ONE AFTER THE OTHER with lane = process slot: every front costs ~140 vector instructions, and of the 64 lanes
// each of them drives, the six that hold an open call do something (rocprofv3, round 3: SQ_ACTIVE_INST_VALU 95 % of the
// kernel's 70 ms per 32,768 histories -- the walk is bound by vector issue and nothing else).  Here a wavefront still owns 64
// consecutive fronts, but lane = front, and the loop runs over the CANDIDATES: the calls open at some front of the chunk.
// A process has one call open at a time, so all but the last candidate of a slot complete inside the chunk: at most
// 64 completions + one call per slot = 128 candidates, ~70 at six calls in flight.  The candidates sit in LDS in slot order
// (32 B each); iteration t broadcasts candidate t to the 64 lanes and each lane decides for ITS front whether the call is open
// there (inv <= F <= ret), appends it to its front's list, ors its slot into its read mask / lookahead mask.  Everything that
// was per-front overhead -- list positions, record addresses, the lookahead word -- is now per lane, i.e. done for 64 fronts by
// one instruction: ~20 vector instructions per candidate, ~22 per front instead of ~140.
//
// Twin masks (tbc_internal.h): entry (front F, call c) gets the slots of the calls open at F with c's effect that complete
// before c.  "Same effect, completes earlier, lifetimes overlap" does not depend on the front: with the candidates ALSO held one
// per lane (two register sets), one ballot per write / cas candidate finds its few static twins, and each of those costs the
// lanes one membership test.
//
// The body is written against wave_env.h like the narrow search kernel, so tests/emu runs it on the CPU (lane-accurate
// emulator) and compares every word with tables built on the host from the definitions (tests/test_walk_emu.py).
#pragma once
#include "wave_env.h"
#include "tbc_internal.h"

namespace tbc {
namespace walk {

constexpr uint32_t kCandCap = 128;          // 64 completions in the chunk + one more call per slot (64 slots)
constexpr uint32_t kCandWords = 8;          // inv_rank, ret_rank, opidx, f | a, b, cls | slot << 8 | prod << 16, effect key
constexpr uint32_t kScanWords = 64;
WV_HD constexpr uint32_t walk_lds_words() { return kCandCap * kCandWords + kScanWords; }

// m |= cond ? 1 << slot : 0 with a UNIFORM slot: one half of the mask, two vector instructions.  (wv::opaque keeps the branch
// a branch: left alone, the compiler computes both halves and selects -- and turns the row update below, one of VCAP entries
// chosen by a uniform index, into VCAP x 6 selects per open read: measured, a third of the kernel.)
WV_DEV void or_slot(uint32_t& lo, uint32_t& hi, bool cond, uint32_t slot) {
  if (slot < 32u) lo = wv::opaque(lo | (cond ? (1u << slot) : 0u));
  else hi = wv::opaque(hi | (cond ? (1u << (slot - 32u)) : 0u));
}
// row[vi] |= cond ? 1 << slot : 0 with a UNIFORM vi: a chain of scalar compares, one entry touched
template <int V0, int VCAP>
WV_DEV void or_row(uint32_t (&lo)[VCAP], uint32_t (&hi)[VCAP], uint32_t vi, bool cond, uint32_t slot) {
  if constexpr (V0 < VCAP) {
    if (vi == (uint32_t)V0) or_slot(lo[V0], hi[V0], cond, slot);
    else or_row<V0 + 1, VCAP>(lo, hi, vi, cond, slot);
  }
}
// what "same effect" compares, in one word.  Twin masks are only built under the dominance rules, i.e. for register values
// 0 .. kMaxRuleValue (tbc_api.hip switches the rules off for anything else): f, a and -- for a cas -- b fit 2 + 15 + 15 bits
WV_DEV uint32_t effect_key(uint32_t f, int32_t a, int32_t b) {
  return (f & 3u) | (((uint32_t)a & 0x7FFFu) << 2) | ((f == TBC_F_CAS ? ((uint32_t)b & 0x7FFFu) : 0u) << 17);
}
static_assert(kMaxRuleValue < 0x7FFF && TBC_F_READ < 4 && TBC_F_WRITE < 4 && TBC_F_CAS < 4, "effect_key packs f and two rule values into a word");

// VCAP: row entries kept in registers (vpad <= VCAP)
template <int VCAP>
WV_DEV void walk_wave(const PackOpenArgs& A, uint32_t wid, uint32_t* lds, uint32_t lane) {
  const uint32_t cph = A.chunks_per_hist;
  const uint32_t hr = wid / cph, c = wid - hr * cph, h = A.h0 + hr;
  if (h >= A.n_hist) return;
  const Hist* H = &A.hist[h];
  const BeamHist* B = &A.bh[h];
  const uint32_t R = H->n_ret;
  const uint32_t F_lo = c * 64u;
  if (H->status != 0 || B->status != 0 || F_lo >= R) return;
  const uint32_t F_hi = F_lo + 64u < R ? F_lo + 64u : R;
  const uint32_t W = H->n_slots;
  const Rec* rec = A.rec + H->rec_off;
  const uint32_t* seg = A.seg + H->seg_off;
  const uint32_t* off = A.off + B->off_off;
  OpRec* lst = A.lst + B->lst_off;
  uint64_t* twn = A.twn ? A.twn + B->lst_off : nullptr;
  const uint32_t V = A.vpad;
  const uint32_t FW = A.front_words ? A.front_words : V;             // u64 words per front: a plain row, or a front record
  uint64_t* rdm = (A.rdm && V) ? A.rdm + H->op_off * FW : nullptr;
  uint64_t* look = A.look ? A.look + look_off(H->op_off, h, 1) : nullptr;
  uint32_t* tmp = A.tmp ? A.tmp + H->op_off : nullptr;
  uint32_t* cand = lds;
  uint32_t* scan = lds + kCandCap * kCandWords;

  // ---- A. lane = process slot: the slot's calls that are open at some front of the chunk are consecutive records -- from the
  // first that has not completed before F_lo to the last invoked before the chunk's last front
  uint32_t lo = 0, k = 0;
  if (lane < W) {
    uint32_t l = seg[lane] + 1u, hi = seg[lane + 1] - 1u;          // [l, hi): the slot's calls; hi = its tail sentinel (inv = ret = kInf)
    const uint32_t tail = hi;
    while (l < hi) {
      const uint32_t mid = (l + hi) >> 1;
      if (rec[mid].ret_rank >= F_lo) hi = mid; else l = mid + 1u;
    }
    lo = l;
    uint32_t i = lo;
    for (;;) {                                                      // four invocation ranks per trip (one or two calls is the rule)
      const uint32_t i1 = i + 1u < tail ? i + 1u : tail, i2 = i + 2u < tail ? i + 2u : tail, i3 = i + 3u < tail ? i + 3u : tail;
      const uint32_t v0 = rec[i < tail ? i : tail].inv_rank, v1 = rec[i1].inv_rank, v2 = rec[i2].inv_rank, v3 = rec[i3].inv_rank;
      if (v0 >= F_hi) break;
      i++;
      if (v1 >= F_hi) break;
      i++;
      if (v2 >= F_hi) break;
      i++;
      if (v3 >= F_hi) break;
      i++;
    }
    k = i - lo;
  }
  // where each slot's candidates start: inclusive scan of k over the lanes (through LDS)
  uint32_t incl = k;
  for (uint32_t d = 1; d < 64u; d <<= 1) {
    scan[lane] = incl;
    wv::barrier();
    const uint32_t add = lane >= d ? scan[lane - d] : 0u;
    wv::barrier();
    incl += add;
  }
  const uint32_t base = incl - k;
  const uint32_t NC_all = wv::readlane(incl, 63u);
  const uint32_t NC = NC_all < kCandCap ? NC_all : kCandCap;      // (never more: see the header)
  for (uint32_t j = 0; j < k; j++) {
    const Rec r = rec[lo + j];
    if (base + j < kCandCap) {
      uint32_t* e = cand + (base + j) * kCandWords;
      e[0] = r.inv_rank; e[1] = r.ret_rank; e[2] = r.opidx; e[3] = r.f;
      e[4] = (uint32_t)r.a; e[5] = (uint32_t)r.b; e[6] = r.cls | (lane << 8) | (r.prod << 16);
      e[7] = ((r.cls & 2u) && (r.cls & 8u)) ? effect_key(r.f, r.a, r.b) : 0xFFFFFFFFu;      // a live write / cas (no key has all bits set: f < 3)
    }
  }
  wv::barrier();

  // ---- B. the candidates once more, one per lane (two sets): what the static twin test asks of them
  uint32_t c_key[2] = {0xFFFFFFFFu, 0xFFFFFFFFu}, c_ret[2] = {0u, 0u};
  if (twn) {
    WV_UNROLL
    for (int s = 0; s < 2; s++) {
      const uint32_t idx = lane + 64u * (uint32_t)s;
      const uint32_t* e = cand + (idx < NC ? idx : 0u) * kCandWords;
      c_key[s] = idx < NC ? e[7] : 0xFFFFFFFFu;
      c_ret[s] = e[1];
    }
  }

  // ---- C. lane = front
  const uint32_t F = F_lo + lane;
  const bool active = F < F_hi;
  const uint32_t Fc = active ? F : F_hi - 1u;                       // (loads of the idle lanes of the last chunk stay in range)
  uint32_t pos = off[Fc];
  const uint32_t px = A.ret_slot[H->ret_off + Fc];
  uint32_t need = kLookNone, xprod = kLookNone, di = 0;
  if (look) {            // the call completing at this front, from the op columns (pack_kernel copied the records from them)
    const uint32_t x = A.ret_op[H->ret_off + Fc];
    const uint32_t xf = A.f[H->op_off + x];
    const int32_t xa = A.a[H->op_off + x], xb = A.b[H->op_off + x];
    const uint32_t xinv = A.scratch[H->frame_off + x];
    need = look_need(xf, xa); xprod = look_prod(xf, xa, xb);
    di = F - xinv < 255u ? F - xinv : 255u;
  }
  uint32_t mine_lo[VCAP], mine_hi[VCAP];
  WV_UNROLL
  for (int v = 0; v < VCAP; v++) { mine_lo[v] = 0u; mine_hi[v] = 0u; }
  uint32_t pm_lo = 0u, pm_hi = 0u;

  WV_NOUNROLL
  for (uint32_t t = 0; t < NC; t++) {
    const uint32_t* e = cand + t * kCandWords;
    const uint32_t inv = e[0], ret = e[1];
    const uint32_t cs = wv::readfirstlane(e[6]);
    const uint32_t cls = cs & 0xFFu, slot = (cs >> 8) & 0xFFu;
    const bool member = active && inv <= F && F <= ret;             // open at this lane's front (a crashed call: ret = kInf)
    if (cls & 2u) {                                                 // live
      const bool isread = (cls & 16u) != 0u;
      if (!(A.branch_lists && isread)) {                            // in the fronts' lists (branch lists: not the reads)
        uint32_t tw_lo = 0u, tw_hi = 0u;
        if (twn && (cls & 8u)) {
          const uint32_t key_t = e[7];
          // same effect, completes before t, still open when t is invoked: the same at every front
          uint64_t s0 = wv::ballot(c_key[0] == key_t && c_ret[0] < ret && c_ret[0] >= inv);
          uint64_t s1 = 0ull;
          if (NC > 64u) s1 = wv::ballot(c_key[1] == key_t && c_ret[1] < ret && c_ret[1] >= inv);
          while (s0 | s1) {
            uint32_t u;
            if (s0) { u = (uint32_t)__builtin_ctzll(s0); s0 &= s0 - 1ull; }
            else { u = 64u + (uint32_t)__builtin_ctzll(s1); s1 &= s1 - 1ull; }
            const uint32_t* eu = cand + u * kCandWords;
            const bool mu = eu[0] <= F && F <= eu[1];               // ... and open at THIS front
            or_slot(tw_lo, tw_hi, mu, (wv::readfirstlane(eu[6]) >> 8) & 0xFFu);
          }
        }
        if (member) {
          OpRec o; o.op = e[2]; o.f_slot = e[3] | (slot << 8) | (ret == F ? kAtFront : 0u); o.a = (int32_t)e[4]; o.b = (int32_t)e[5];
          lst[pos] = o;
          if (twn) twn[pos] = (uint64_t)tw_lo | ((uint64_t)tw_hi << 32);
          pos++;
        }
      }
      if (isread && rdm) {                                          // open-read masks by value
        const int32_t va = (int32_t)wv::readfirstlane(e[4]);
        const uint32_t vi = rdm_index(va, V);
        if (vi != 0u || va == TBC_NIL) or_row<0, VCAP>(mine_lo, mine_hi, vi, member, slot);
      }
    }
    if (look && (cls & 6u)) {                                       // who else open here (live, or crashed and a candidate) produces what the completing call needs
      const uint32_t prod = cs >> 16;
      if (prod != kLookNone) or_slot(pm_lo, pm_hi, member && prod == need, slot);
    }
  }

  if (!active) return;
  if (rdm) {
    WV_UNROLL
    for (int v = 0; v < VCAP; v++) if ((uint32_t)v < V) rdm[(uint64_t)F * FW + (uint32_t)v] = (uint64_t)mine_lo[v] | ((uint64_t)mine_hi[v] << 32);
  }
  if (look) {
    uint64_t pm = (uint64_t)pm_lo | ((uint64_t)pm_hi << 32);
    pm &= ~(1ull << (px & 63u));                                    // one call per slot is open at a front: this is the completing call itself
    const uint64_t w0 = (uint64_t)(px & 0xFFFFu) | (uint64_t)need << 16 | (uint64_t)xprod << 24 | (uint64_t)di << 32 | (255ull << 40);
    look[(uint64_t)F * 2u] = w0;
    look[(uint64_t)F * 2u + 1u] = pm;
    tmp[F] = 255u;
  }
}

}  // namespace walk
}  // namespace tbc
