// open_walk_impl.h -- K1b's front walk with LANE = FRONT (gfx950; mask_words == 1, i.e. at most 64 process slots).
//
// What it leaves in HBM is what pack_open.hip's open_walk_kernel<1> + open_dprod_kernel + front_meta_kernel leave, word for
// word (lst / twn / rdm rows or front records / look; the definitions are in that file's header and tbc_internal.h).  How it
// gets there is turned round.  open_walk_kernel gives a wavefront 64 fronts and walks them ONE AFTER THE OTHER with lane =
// process slot: every front costs ~140 vector instructions, and of the 64 lanes each of them drives, the six that hold an open
// call do something (rocprofv3, round 3: SQ_ACTIVE_INST_VALU 95 % of the kernel's 70 ms per 32,768 histories -- the walk is
// bound by vector issue and nothing else).  Here a wavefront still owns 64 consecutive fronts, but lane = front, and the loop
// runs over the CANDIDATES: the calls open at some front of the chunk (plus, for the lookahead's producer distance, those that
// completed within the seven ranks before it).  A process has one call open at a time, so all but the last candidate of a
// slot complete inside ranks F_lo - 7 .. F_hi - 1: at most 71 completions + one call per slot = 135 candidates, ~75 at six
// calls in flight.  The candidates sit in LDS in slot order (32 B each, with what the loop asks of them worked out once, as
// flag bits); iteration t broadcasts candidate t to the 64 lanes and each lane decides for ITS front whether the call is open
// there (inv <= F <= ret), appends it to its front's list, ors its slot into its read row / lookahead mask, keeps the nearest
// producer.  Everything that was per-front overhead -- list positions, record addresses, the lookahead word, the front
// record's windows -- is per lane, i.e. done for 64 fronts by one instruction.
//
// Twin masks (tbc_internal.h): entry (front F, call c) gets the slots of the calls open at F with c's effect that complete
// before c.  "Same effect, completes earlier, lifetimes overlap" does not depend on the front: with the candidates ALSO held one
// per lane (three register sets), one ballot per write / cas candidate finds its few static twins, and each of those costs the
// lanes one membership test.
//
// The body is written against wave_env.h like the narrow search kernel, so tests/emu runs it on the CPU (lane-accurate
// emulator) and compares every word with tables built on the host from the definitions (tests/test_walk_emu.py).
#pragma once
#include "wave_env.h"
#include "tbc_internal.h"

namespace tbc {
namespace walk {

constexpr uint32_t kCandCap = 136;          // 71 completions (ranks F_lo - 7 .. F_hi - 1) + one more call per slot (64 slots), rounded up
constexpr uint32_t kCandWords = 8;          // inv_rank, ret_rank, opidx, f | slot << 8;  a, b, flags (below), effect key
constexpr uint32_t kAuxWords = 80;          // the scan of the per-slot counts; later the rank codes of the chunk + 6
WV_HD constexpr uint32_t walk_lds_words() { return kCandCap * kCandWords + kAuxWords; }

// candidate word 6: what the loop does with the call, decided once when the table is filled
constexpr uint32_t kCList = 1u;             // live and in the fronts' lists (branch lists: not a read)
constexpr uint32_t kCTwin = 2u;             // ... and its entries carry a twin mask (a write / cas, twin masks wanted)
constexpr uint32_t kCRow = 4u;              // a live read whose value has a row entry: bits 4..8 = the entry
constexpr uint32_t kCLook = 8u;             // produces a register value (bits 24..31): lookahead mask, producer distance
//                                             bits 12..17 = the process slot

// m |= cond ? 1 << slot : 0 with a UNIFORM slot: one half of the mask, two vector instructions.  (wv::opaque keeps the branch
// a branch: left alone, the compiler computes both halves and selects.)
WV_DEV void or_slot(uint32_t& lo, uint32_t& hi, bool cond, uint32_t slot) {
  if (slot < 32u) lo = wv::opaque(lo | (cond ? (1u << slot) : 0u));
  else hi = wv::opaque(hi | (cond ? (1u << (slot - 32u)) : 0u));
}
// what "same effect" compares, in one word.  Twin masks are only built under the dominance rules, i.e. for register values
// 0 .. kMaxRuleValue (tbc_api.hip switches the rules off for anything else): f, a and -- for a cas -- b fit 2 + 15 + 15 bits
WV_DEV uint32_t effect_key(uint32_t f, int32_t a, int32_t b) {
  return (f & 3u) | (((uint32_t)a & 0x7FFFu) << 2) | ((f == TBC_F_CAS ? ((uint32_t)b & 0x7FFFu) : 0u) << 17);
}
static_assert(kMaxRuleValue < 0x7FFF && TBC_F_READ < 3 && TBC_F_WRITE < 3 && TBC_F_CAS < 3, "effect_key packs f and two rule values into a word, and no key has all bits set");
static_assert(kLookNone <= 0xFFu, "the produced value shares a word with the flags");

// the nine bits a compact front record keeps per rank: process slot | read kind << 6 (7 = not a read)
WV_DEV uint32_t rank_code(uint32_t slot, uint32_t f, int32_t a, uint32_t vpad) {
  const uint32_t k8 = f == TBC_F_READ ? (rdm_index(a, vpad) & 0xFFu) : 0xFFu;
  return (slot & 63u) | ((k8 == 0xFFu ? 7u : (k8 & 7u)) << 6);
}

// A front's open-read row lives in registers as u32x16 vectors (16 words = 8 entries each: entry v = words 2v, slots 0..31, and
// 2v + 1), the word to touch picked by a UNIFORM index: s_set_gpr_idx, one indexed move each way.  (Plain locals on purpose: as
// members of a struct the compiler moves them to scratch, and a chain of compares over an array made it copy the whole array.)
template <int Q>
WV_DEV void store_entries(const wv::u32x16& r, uint64_t* row, uint32_t n) {       // entries 8Q .. 8Q + 7 below n
  WV_UNROLL
  for (int v = 0; v < 8; v++) if ((uint32_t)(8 * Q + v) < n) row[8 * Q + v] = (uint64_t)r[2 * v] | ((uint64_t)r[2 * v + 1] << 32);
}

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
  const bool compact = A.front_compact != 0u;                        // the 64 B record, written whole (masks, list location, windows)
  uint64_t* rdm = (A.rdm && (V || compact)) ? A.rdm + H->op_off * FW : nullptr;
  const bool want_tw = twn != nullptr;
  uint64_t* look = A.look ? A.look + look_off(H->op_off, h, 1u) : nullptr;
  uint32_t* cand = lds;
  uint32_t* aux = lds + kCandCap * kCandWords;

  // ---- A. lane = process slot: the slot's candidates are consecutive records -- from the first that has not completed before
  // F_lo (lookahead: F_lo - 7, for the producer distance) to the last invoked before the chunk's last front
  const uint32_t ret_from = look ? (F_lo >= kLookahead - 1u ? F_lo - (kLookahead - 1u) : 0u) : F_lo;
  uint32_t lo = 0, k = 0;
  if (lane < W) {
    uint32_t l = seg[lane] + 1u, hi = seg[lane + 1] - 1u;          // [l, hi): the slot's calls; hi = its tail sentinel (inv = ret = kInf)
    const uint32_t tail = hi;
    while (l < hi) {
      const uint32_t mid = (l + hi) >> 1;
      if (rec[mid].ret_rank >= ret_from) hi = mid; else l = mid + 1u;
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
    aux[lane] = incl;
    wv::barrier();
    const uint32_t add = lane >= d ? aux[lane - d] : 0u;
    wv::barrier();
    incl += add;
  }
  const uint32_t base = incl - k;
  const uint32_t NC_all = wv::readlane(incl, 63u);
  const uint32_t NC = NC_all < kCandCap ? NC_all : kCandCap;      // (never more: see the header)
  for (uint32_t j = 0; j < k; j++) {
    const Rec r = rec[lo + j];
    if (base + j < kCandCap) {
      const bool live = (r.cls & 2u) != 0u, isread = (r.cls & 16u) != 0u, wc = (r.cls & 8u) != 0u;
      const bool in_chunk = r.ret_rank >= F_lo;                     // (else: completed just before the chunk, a producer only)
      const bool listed = live && in_chunk && !(A.branch_lists && isread);
      const uint32_t vi = rdm_index(r.a, V);
      uint32_t fl = (lane << 12) | (r.prod << 24);
      if (listed) fl |= kCList;
      if (listed && wc && want_tw) fl |= kCTwin;
      if (live && in_chunk && isread && rdm && V && (vi != 0u || r.a == TBC_NIL)) fl |= kCRow | (vi << 4);
      if (look && (r.cls & 6u) && r.prod != kLookNone) fl |= kCLook;
      uint32_t* e = cand + (base + j) * kCandWords;
      e[0] = r.inv_rank; e[1] = r.ret_rank; e[2] = r.opidx; e[3] = r.f | (lane << 8);
      e[4] = (uint32_t)r.a; e[5] = (uint32_t)r.b; e[6] = fl;
      e[7] = (live && wc && in_chunk) ? effect_key(r.f, r.a, r.b) : 0xFFFFFFFFu;
    }
  }
  wv::barrier();

  // ---- A2. list_order 1: the loop below takes the candidates in order of completion (then every front's list comes out in that
  // order: a front's entries are appended as the loop meets its members).  Candidate i's place = the candidates that complete before it
  // (ties -- crashed calls, ret = kInf, which are in no list -- by table position); the places as 16-bit words over the scan's scratch.
  // list_order 2: in order of completion with the :write calls after everything else -- the search takes a config's candidates last to
  // first and pops the last child first, so a :cas the state allows NOW is tried before a :write, which the state always allows
  // (oracle/wgl_beam.c list order 4: at 19 calls in flight a quarter fewer rounds again than plain completion order).  The key: the
  // completion rank plus 2^30 for a :write (ranks are below 2^30).
  // list_order 16 + W: a :write takes the place of a call completing W ranks later (the soft form of the same preference; keys are doubled
  // ranks, a :write's odd: after the call it ties with) -- W = 16 .. 24 is the best of the scan at 6, 19 and 32 calls in flight (oracle list
  // order 16 + W: at 19 in flight 19 - 22k rounds a history against 28k writes last, 59k plain completion order).  Any key gives a
  // permutation: a place is the count of smaller keys, ties by table position.
  // The keys are made UNIQUE so that a place is one compare per pair: completion ranks are (live calls), and so are ranks + 2^30 for
  // the writes and doubled ranks with the writes' odd; the only ties of the definition are among crashed calls (ret = kInf), which are in
  // no list -- and what the loop does for an unlisted candidate (a bit of a read row, of the lookahead mask, the producer distance)
  // does not depend on when it meets it: they take keys above every live one, by table position.  The keys stay in registers, one
  // candidate per lane (three sets); candidate j's key reaches the lanes by a lane read (j is uniform), and every lane counts the
  // smaller keys for each of its candidates: 1 + 2 vector instructions per set instead of a dozen with two LDS reads.
  const uint32_t list_order = A.order_of ? A.order_of[h] : A.list_order;          // (the history's own order, if the batch says one)
  const bool by_ret = list_order != 0u;
  const uint32_t wr_last = list_order == 2u ? 0x40000000u : list_order >= 16u ? 2u * (list_order - 16u) + 1u : 0u;
  const uint32_t rk_mul = list_order >= 16u ? 2u : 1u;
  uint16_t* const perm = reinterpret_cast<uint16_t*>(aux);
  if (by_ret) {
    uint32_t my[3], place[3] = {0u, 0u, 0u};
    WV_UNROLL
    for (int s = 0; s < 3; s++) {
      const uint32_t idx = lane + 64u * (uint32_t)s;
      const uint32_t* e = cand + (idx < NC ? idx : 0u) * kCandWords;
      const uint32_t rt = e[1];
      my[s] = idx >= NC ? 0xFFFFFFFFu : rt == kInf ? (0xFFFFFF00u | idx) : rt * rk_mul + ((e[3] & 0xFFu) == TBC_F_WRITE ? wr_last : 0u);
    }
    const uint32_t n0 = NC < 64u ? NC : 64u, n1 = NC <= 64u ? 0u : (NC < 128u ? NC - 64u : 64u), n2 = NC <= 128u ? 0u : NC - 128u;
    if (NC <= 64u) {
      for (uint32_t j = 0; j < n0; j++) place[0] += wv::readlane(my[0], j) < my[0] ? 1u : 0u;
    } else if (NC <= 128u) {
      for (uint32_t j = 0; j < n0; j++) { const uint32_t kj = wv::readlane(my[0], j); place[0] += kj < my[0] ? 1u : 0u; place[1] += kj < my[1] ? 1u : 0u; }
      for (uint32_t j = 0; j < n1; j++) { const uint32_t kj = wv::readlane(my[1], j); place[0] += kj < my[0] ? 1u : 0u; place[1] += kj < my[1] ? 1u : 0u; }
    } else {
      WV_UNROLL
      for (int s2 = 0; s2 < 3; s2++) {
        const uint32_t cnt = s2 == 0 ? n0 : s2 == 1 ? n1 : n2;
        for (uint32_t j = 0; j < cnt; j++) {
          const uint32_t kj = wv::readlane(my[s2], j);
          place[0] += kj < my[0] ? 1u : 0u; place[1] += kj < my[1] ? 1u : 0u; place[2] += kj < my[2] ? 1u : 0u;
        }
      }
    }
    WV_UNROLL
    for (int s = 0; s < 3; s++) {
      const uint32_t idx = lane + 64u * (uint32_t)s;
      if (idx < NC) perm[place[s]] = (uint16_t)idx;
    }
    wv::barrier();
  }

  // ---- B. the candidates once more, one per lane (three sets): what the static twin test asks of them
  uint32_t c_key[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, c_ret[3] = {0u, 0u, 0u};
  if (want_tw) {
    WV_UNROLL
    for (int s = 0; s < 3; s++) {
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
  const uint32_t pos0 = off[Fc];
  uint32_t pos = pos0;
  const uint32_t px = A.ret_slot[H->ret_off + Fc];
  uint32_t need = kLookNone, xprod = kLookNone, di = 0, x = 0, xf = kFNone;
  int32_t xa = 0;
  if (look || compact) {     // the call completing at this front, from the op columns (pack_kernel copied the records from them)
    x = A.ret_op[H->ret_off + Fc];
    xf = A.f[H->op_off + x];
    xa = A.a[H->op_off + x];
  }
  if (look) {
    const int32_t xb = A.b[H->op_off + x];
    const uint32_t xinv = A.scratch[H->frame_off + x];
    need = look_need(xf, xa); xprod = look_prod(xf, xa, xb);
    di = F - xinv < 255u ? F - xinv : 255u;
  }
  wv::u32x16 r0, r1, r2, r3;                // the front's row (r1 .. r3: rows of more than 8 entries only)
  WV_UNROLL
  for (int i = 0; i < 16; i++) { r0[i] = 0u; r1[i] = 0u; r2[i] = 0u; r3[i] = 0u; }
  uint32_t pm_lo = 0u, pm_hi = 0u, dmin = 255u;

  WV_NOUNROLL
  for (uint32_t t = 0; t < NC; t++) {
    const uint32_t* e = cand + (by_ret ? (uint32_t)perm[t] : t) * kCandWords;
    const uint32_t inv = e[0], ret = e[1];
    const uint32_t fl = wv::readfirstlane(e[6]);
    const uint32_t slot = (fl >> 12) & 63u;
    const bool member = active && inv <= F && F <= ret;             // open at this lane's front (a crashed call: ret = kInf)
    if (fl & kCList) {
      uint32_t tw_lo = 0u, tw_hi = 0u;
      if (fl & kCTwin) {
        const uint32_t key_t = e[7];
        // same effect, completes before t, still open when t is invoked: the same at every front
        uint64_t s0 = wv::ballot(c_key[0] == key_t && c_ret[0] < ret && c_ret[0] >= inv);
        uint64_t s1 = 0ull, s2 = 0ull;
        if (NC > 64u) s1 = wv::ballot(c_key[1] == key_t && c_ret[1] < ret && c_ret[1] >= inv);
        if (NC > 128u) s2 = wv::ballot(c_key[2] == key_t && c_ret[2] < ret && c_ret[2] >= inv);
        while (s0 | s1 | s2) {
          uint32_t u;
          if (s0) { u = (uint32_t)__builtin_ctzll(s0); s0 &= s0 - 1ull; }
          else if (s1) { u = 64u + (uint32_t)__builtin_ctzll(s1); s1 &= s1 - 1ull; }
          else { u = 128u + (uint32_t)__builtin_ctzll(s2); s2 &= s2 - 1ull; }
          const uint32_t* eu = cand + u * kCandWords;
          const bool mu = eu[0] <= F && F <= eu[1];                 // ... and open at THIS front
          or_slot(tw_lo, tw_hi, mu, (wv::readfirstlane(eu[6]) >> 12) & 63u);
        }
      }
      if (member) {
        OpRec o; o.op = e[2]; o.f_slot = e[3] | (ret == F ? kAtFront : 0u); o.a = (int32_t)e[4]; o.b = (int32_t)e[5];
        lst[pos] = o;
        if (twn) twn[pos] = (uint64_t)tw_lo | ((uint64_t)tw_hi << 32);
        pos++;
      }
    }
    if (fl & kCRow)                                                 // open-read masks by value: one word of one entry
    {
      const uint32_t w = ((fl >> 4) & 31u) * 2u + (slot >> 5), bit = member ? (1u << (slot & 31u)) : 0u;
      if constexpr (VCAP <= 8) r0[w & 15u] |= bit;
      else {
        const uint32_t q = w >> 4;
        if (q == 0u) r0[w & 15u] |= bit; else if (q == 1u) r1[w & 15u] |= bit; else if (q == 2u) r2[w & 15u] |= bit; else r3[w & 15u] |= bit;
      }
    }
    if (fl & kCLook) {                 // who else, open here, produces what the completing call needs; how recently such a call was invoked
      const bool hit = (fl >> 24) == need;
      or_slot(pm_lo, pm_hi, member && hit, slot);
      const uint32_t dist = F - inv;                                // (invoked after this front: wraps past kLookahead)
      if (hit && dist < kLookahead && e[2] != x) dmin = dist < dmin ? dist : dmin;
    }
  }

  // ---- D. the rest of each front's record / row
  uint64_t w6 = 0ull, w7 = 0ull;
  if (compact) {           // list location and the window of the next seven ranks (tbc_internal.h), through LDS: codes of the chunk + 6 after
    if (by_ret) wv::barrier();                                      // (the places above lived in these words: every lane is past the loop)
    aux[lane] = active ? rank_code(px, xf, xa, V) : (7u << 6);
    if (lane < kFrontCompactRanks - 1u) {
      const uint32_t G = F_lo + 64u + lane;
      uint32_t code = 7u << 6;                                      // (past the last rank: slot 0, not a read)
      if (G < R) {
        const uint32_t gx = A.ret_op[H->ret_off + G];
        code = rank_code(A.ret_slot[H->ret_off + G], A.f[H->op_off + gx], A.a[H->op_off + gx], V);
      }
      aux[64u + lane] = code;
    }
    wv::barrier();
    WV_UNROLL
    for (uint32_t l = 0; l < kFrontCompactRanks; l++) w7 |= (uint64_t)aux[lane + l] << (9u * l);
    const uint32_t nl = pos - pos0, nc = A.ncr[B->off_off + Fc];
    w6 = (uint64_t)pos0 | ((uint64_t)(nl & 0xFFu) << 32) | ((uint64_t)((nl + nc) & 0xFFFFFFu) << 40);
  }
  if (!active) return;
  if (rdm) {
    uint64_t* row = rdm + (uint64_t)F * FW;
    const uint32_t n_row = compact ? 6u : V;
    store_entries<0>(r0, row, n_row);
    if constexpr (VCAP > 8) { store_entries<1>(r1, row, n_row); store_entries<2>(r2, row, n_row); store_entries<3>(r3, row, n_row); }
    if (compact) { row[6] = w6; row[7] = w7; }
  }
  if (look) {
    uint64_t pm = (uint64_t)pm_lo | ((uint64_t)pm_hi << 32);
    pm &= ~(1ull << (px & 63u));                                    // one call per slot is open at a front: this is the completing call itself
    const uint64_t w0 = (uint64_t)(px & 0xFFFFu) | (uint64_t)need << 16 | (uint64_t)xprod << 24 | (uint64_t)di << 32 | ((uint64_t)dmin << 40);
    look[(uint64_t)F * 2u] = w0;
    look[(uint64_t)F * 2u + 1u] = pm;
  }
}

}  // namespace walk
}  // namespace tbc
