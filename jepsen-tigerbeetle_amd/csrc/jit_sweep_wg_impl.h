// jit_sweep_wg_impl.h -- K6w: one segment of the level sweep (jit_sweep.hip, K6) swept by a WORKGROUP of NW wavefronts.
//
// Why: one history's sweep takes as long as its most expensive wavefront, and that wavefront is ONE burst of concurrency -- a
// level of 240 .. 1,440 configs whose sub-rounds are thousands of (config, open call) pairs -- that no cut divides and that a
// single origin reaches whole (scripts/sweep_cut_study.py: 21k of a history's 400k probes on the critical wavefront, 50x the
// mean, whatever the segment length, the cut rule or the origins per wavefront).  More segments do not shorten it; more LANES on
// that level do: here a level's passes run 64 x NW pairs at a time over the same LDS-resident sets.
//
// Same algorithm, same sets, same statistics as K6 (verdict, relation, level sizes, probes, sub-rounds: oracle/sweep_ref.c,
// bit for bit) -- only who does what differs:
//   * thread t of the workgroup takes config / pair  base + t  of a pass (K6: lane);
//   * insertion: the probe loop is K6's (claim an empty slot provisionally with the THREAD number, meet a provisional slot ->
//     compare with the claimant's staged key, OR the origin sets of equal keys) run by every wavefront on its own pairs with
//     workgroup-scope LDS atomics and no synchronisation inside; three workgroup barriers around it (staged keys visible ->
//     all claims and ORs done, per-wavefront winner counts visible -> entries committed).  Winners are numbered wavefront by
//     wavefront (K6: lane order) -- a set is a set;
//   * everything a decision hangs on (set sizes, generations, status, the level's numbers) is kept by every thread and
//     derived from LDS words all of them read after a barrier: control flow is uniform across the workgroup by construction.
// Register family only (the models K6 cuts into segments); the dump pass stays K6's.
//
// Written against wave_env.h / wave_env_wg.h: tests/emu runs this very file on NW x 64 host fibers with the wavefronts
// interleaved in seeded orders and compares every record with oracle/sweep_ref.c (tests/test_sweep_wg_emu.py).
#pragma once
#include "tbc_internal.h"
#include "wave_env_wg.h"

namespace tbc {
namespace sweepwg {

constexpr uint32_t kCand = kSweepCandMax;
constexpr uint32_t kProv = 1u << 16;          // slot holds a thread number (this pass's claimant), not an entry
constexpr uint32_t kGenShift = 17;

struct __attribute__((aligned(16))) Ent { uint32_t mlo, mhi, st, org; };

template <uint32_t HS>
WV_DEV uint32_t key_slot(uint32_t mlo, uint32_t mhi, uint32_t st) {
  uint32_t h = mlo * 0x9E3779B1u ^ mhi * 0x85EBCA77u ^ st * 0xC2B2AE3Du;
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 13;
  return h & (HS - 1u);
}
struct Build {
  Ent* e;
  uint32_t* tab;
  uint32_t n;       // the same in every thread
  uint32_t gen;     // the same in every thread, != 0
};

// scratch words the wavefronts exchange counts through (u32 words)
template <uint32_t NW>
struct Scratch {
  static constexpr uint32_t kAny = 0;                    // [2][NW]: does the wavefront have anything to insert (by call parity)
  static constexpr uint32_t kSide = 2 * NW;              // [2][NW]: the wavefront's side count (by call parity)
  static constexpr uint32_t kTot = 4 * NW;               // [NW][2]: winners per target set
  static constexpr uint32_t kOrg = 6 * NW;               // [2]: origins alive at the level just built (by level parity)
  static constexpr uint32_t kProbes = 6 * NW + 2;        // [NW][2]: probes per wavefront, at the end
  static constexpr uint32_t kBad = 8 * NW + 2;           // a final state outside the origin domain was seen
  static constexpr uint32_t kWords = 8 * NW + 8;
};

// COMPACT (see segment): + a block's child offsets (16 bits a config, 64 x NW of them) and 2 x NW scan words
template <uint32_t CAP, uint32_t NW, bool COMPACT = false>
constexpr uint32_t lds_words() {
  return 3 * CAP * 4 + 2 * (2 * CAP) + 64 * NW * 4 + kCand * 4 + kCand * 2 + 2 * 32 * 2 + CAP / 2 + 128 + Scratch<NW>::kWords +
         (COMPACT ? 64 * NW / 2 + 2 * NW : 0u);
}

// exclusive prefix sum of x over the workgroup's threads (thread order), *total = the sum; every thread calls it.  Inside a
// wavefront: rows of 16 by row shifts, the four row sums by lane reads; across wavefronts: NW totals in LDS (sw: 2 x NW words, used
// alternately by consecutive calls -- `flip` -- so that one workgroup barrier per call is enough)
template <uint32_t NW>
WV_DEV uint32_t wg_scan_excl(uint32_t x, uint32_t& total, uint32_t* sw, uint32_t flip, uint32_t lane, uint32_t wave) {
  uint32_t y = x;
  y += wv::row_shr0<1>(y); y += wv::row_shr0<2>(y); y += wv::row_shr0<4>(y); y += wv::row_shr0<8>(y);
  const uint32_t t0 = wv::readlane(y, 15u), t1 = wv::readlane(y, 31u), t2 = wv::readlane(y, 47u), t3 = wv::readlane(y, 63u);
  const uint32_t row = lane >> 4;
  y += (row > 0u ? t0 : 0u) + (row > 1u ? t1 : 0u) + (row > 2u ? t2 : 0u);
  uint32_t* tt = sw + (flip & 1u) * NW;
  if (lane == 0u) tt[wave] = t0 + t1 + t2 + t3;
  wv::wg_barrier();
  uint32_t base = 0u, all = 0u;
  WV_UNROLL
  for (uint32_t w = 0; w < NW; w++) { const uint32_t v = wv::lds_ld32(&tt[w]); base += w < wave ? v : 0u; all += v; }
  total = all;
  return base + y - x;
}

WV_DEV uint64_t mask_of(const Ent& e) { return (uint64_t)e.mlo | ((uint64_t)e.mhi << 32); }
WV_DEV bool reg_ok(int32_t st, uint32_t f, int32_t a) {
  return f == TBC_F_WRITE || (f == TBC_F_READ && (a == TBC_NIL || a == st)) || (f == TBC_F_CAS && a == st);
}
WV_DEV int32_t reg_apply(int32_t st, uint32_t f, int32_t a, int32_t b) { return f == TBC_F_WRITE ? a : (f == TBC_F_CAS ? b : st); }

// What a thread knows about itself and what the workgroup shares
template <uint32_t NW>
struct Ctx {
  uint32_t tid, lane, wave;
  Ent* stage;            // 64 * NW staged keys
  uint32_t* ws;          // Scratch<NW>
  uint32_t parity;       // call parity of insert2 (the same in every thread)
};

// Start a new set in table `tab`: a new generation makes every old slot read as empty.
template <uint32_t HS, uint32_t NW>
WV_DEV void build_begin(Build& b, Ent* e, uint32_t* tab, uint32_t& gen_counter, const Ctx<NW>& X) {
  gen_counter++;
  if (gen_counter >= (1u << (32 - kGenShift))) {       // generation wrapped: really clear
    for (uint32_t i = X.tid; i < HS; i += 64 * NW) tab[i] = 0u;
    gen_counter = 1;
    wv::wg_barrier();
  }
  b.e = e; b.tab = tab; b.n = 0; b.gen = gen_counter;
}

// Insert up to 64 * NW configs, one per thread, each into set A (sel 1) or set B (sel 2), OR-ing the origin sets of equal
// keys.  `side`: a flag per thread the callers want numbered in the same breath (sub-round 0's "still needs X" list):
// side_off = how many threads before this one have it set, side_total = how many in all.  Returns false if a set outgrew CAP.
template <uint32_t CAP, uint32_t NW>
WV_DEV bool insert2(Build& A, Build& B, uint32_t sel, uint32_t mlo, uint32_t mhi, uint32_t st, uint32_t org,
                    bool side, uint32_t& side_off, uint32_t& side_total, Ctx<NW>& X) {
  using S = Scratch<NW>;
  constexpr uint32_t HS = 2 * CAP;
  const uint32_t par = X.parity & 1u;
  X.parity++;
  X.stage[X.tid] = Ent{mlo, mhi, st, sel ? org : 0u};
  const uint64_t sb = wv::ballot(side);
  const uint64_t anyb = wv::ballot(sel != 0u);
  if (X.lane == 0) {
    X.ws[S::kAny + par * NW + X.wave] = anyb ? 1u : 0u;
    X.ws[S::kSide + par * NW + X.wave] = (uint32_t)__builtin_popcountll(sb);
  }
  wv::wg_barrier();                                           // staged keys and the counts are visible
  uint32_t any = 0, soff = 0, stot = 0;
  WV_UNROLL
  for (uint32_t w = 0; w < NW; w++) {
    any |= X.ws[S::kAny + par * NW + w];
    const uint32_t c = X.ws[S::kSide + par * NW + w];
    if (w < X.wave) soff += c;
    stot += c;
  }
  side_off = soff + (uint32_t)__builtin_popcountll(sb & ((1ull << X.lane) - 1ull));
  side_total = stot;
  if (!any) return true;                                      // (the same in every thread: nobody has passed another barrier)
  uint32_t* const tab = sel == 2u ? B.tab : A.tab;
  Ent* const ent = sel == 2u ? B.e : A.e;
  const uint32_t gen = sel == 2u ? B.gen : A.gen;
  const uint32_t gtag = gen << kGenShift;
  uint32_t h = key_slot<HS>(mlo, mhi, st), mine = 0;
  bool pend = sel != 0u, won = false;
  while (wv::ballot(pend)) {
    if (pend) {
      uint32_t s = wv::lds_ld32(&tab[h]);
      if ((s >> kGenShift) != gen) {                      // empty: claim it with the thread number
        const uint32_t old = wv::lds_cas32(&tab[h], s, gtag | kProv | X.tid);
        if (old == s) { won = true; mine = h; pend = false; }
        else s = old;                                     // claimed in this very pass by another thread
      }
      if (pend) {
        const Ent* kp = (s & kProv) ? &X.stage[s & 0x3FFu] : &ent[s & 0xFFFFu];
        if (kp->mlo == mlo && kp->mhi == mhi && kp->st == st) {
          wv::lds_or32(const_cast<uint32_t*>(&kp->org), org);
          pend = false;
        } else {
          h = (h + 1u) & (HS - 1u);
        }
      }
    }
  }
  const uint64_t wa = wv::ballot(won && sel == 1u), wb = wv::ballot(won && sel == 2u);
  if (X.lane == 0) {
    X.ws[S::kTot + 2 * X.wave] = (uint32_t)__builtin_popcountll(wa);
    X.ws[S::kTot + 2 * X.wave + 1] = (uint32_t)__builtin_popcountll(wb);
  }
  wv::wg_barrier();                                           // every claim and every OR is done; the winner counts are visible
  uint32_t offa = A.n, offb = B.n, ta = A.n, tb = B.n;
  WV_UNROLL
  for (uint32_t w = 0; w < NW; w++) {
    const uint32_t ca = X.ws[S::kTot + 2 * w], cb = X.ws[S::kTot + 2 * w + 1];
    if (w < X.wave) { offa += ca; offb += cb; }
    ta += ca; tb += cb;
  }
  const bool fits = ta <= CAP && tb <= CAP;
  if (won && fits) {
    const uint64_t below = (1ull << X.lane) - 1ull;
    const uint32_t idx = sel == 2u ? offb + (uint32_t)__builtin_popcountll(wb & below) : offa + (uint32_t)__builtin_popcountll(wa & below);
    ent[idx] = Ent{mlo, mhi, st, X.stage[X.tid].org};
    tab[mine] = gtag | idx;
  }
  if (fits) { A.n = ta; B.n = tb; }
  wv::wg_barrier();                                           // entries committed; the stage and the counts may be written again
  return fits;
}

// SOLO: insert2 for a pass that fits ONE wavefront, run by wavefront 0 alone (its 64 threads call it, nobody else): the three workgroup
// barriers become wavefront barriers -- the other wavefronts wait at the ONE workgroup barrier behind which the caller publishes the
// new set sizes (solo_publish / solo_collect below).  Same claims, same ORs, same entries: a set is a set.
template <uint32_t CAP, uint32_t NW>
WV_DEV bool insert2_solo(Build& A, Build& B, uint32_t sel, uint32_t mlo, uint32_t mhi, uint32_t st, uint32_t org,
                         bool side, uint32_t& side_off, uint32_t& side_total, Ctx<NW>& X) {
  constexpr uint32_t HS = 2 * CAP;
  X.stage[X.tid] = Ent{mlo, mhi, st, sel ? org : 0u};
  const uint64_t sb = wv::ballot(side);
  const uint64_t anyb = wv::ballot(sel != 0u);
  wv::barrier();                                              // staged keys visible to the wavefront
  side_off = (uint32_t)__builtin_popcountll(sb & ((1ull << X.lane) - 1ull));
  side_total = (uint32_t)__builtin_popcountll(sb);
  if (!anyb) return true;
  uint32_t* const tab = sel == 2u ? B.tab : A.tab;
  Ent* const ent = sel == 2u ? B.e : A.e;
  const uint32_t gen = sel == 2u ? B.gen : A.gen;
  const uint32_t gtag = gen << kGenShift;
  uint32_t h = key_slot<HS>(mlo, mhi, st), mine = 0;
  bool pend = sel != 0u, won = false;
  while (wv::ballot(pend)) {
    if (pend) {
      uint32_t s = wv::lds_ld32(&tab[h]);
      if ((s >> kGenShift) != gen) {
        const uint32_t old = wv::lds_cas32(&tab[h], s, gtag | kProv | X.tid);
        if (old == s) { won = true; mine = h; pend = false; }
        else s = old;
      }
      if (pend) {
        const Ent* kp = (s & kProv) ? &X.stage[s & 0x3FFu] : &ent[s & 0xFFFFu];
        if (kp->mlo == mlo && kp->mhi == mhi && kp->st == st) {
          wv::lds_or32(const_cast<uint32_t*>(&kp->org), org);
          pend = false;
        } else {
          h = (h + 1u) & (HS - 1u);
        }
      }
    }
  }
  const uint64_t wa = wv::ballot(won && sel == 1u), wb = wv::ballot(won && sel == 2u);
  wv::barrier();                                              // every claim and every OR of the wavefront is done
  const uint32_t ta = A.n + (uint32_t)__builtin_popcountll(wa), tb = B.n + (uint32_t)__builtin_popcountll(wb);
  const bool fits = ta <= CAP && tb <= CAP;
  if (won && fits) {
    const uint64_t below = (1ull << X.lane) - 1ull;
    const uint32_t idx = sel == 2u ? B.n + (uint32_t)__builtin_popcountll(wb & below) : A.n + (uint32_t)__builtin_popcountll(wa & below);
    ent[idx] = Ent{mlo, mhi, st, X.stage[X.tid].org};
    tab[mine] = gtag | idx;
  }
  if (fits) { A.n = ta; B.n = tb; }
  wv::barrier();
  return fits;
}
// what wavefront 0 did alone, told to the workgroup: two words by pass parity (the spare words of Scratch), one workgroup barrier
template <uint32_t NW>
WV_DEV void solo_exchange(Build& A, Build& B, bool& fits, uint32_t& side_total, Ctx<NW>& X) {
  using S = Scratch<NW>;
  uint32_t* pub = X.ws + S::kBad + 1u + 2u * (X.parity & 1u);
  X.parity++;
  if (X.tid == 0u) { pub[0] = A.n | (B.n << 16); pub[1] = (fits ? 1u : 0u) | (side_total << 1); }
  wv::wg_barrier();
  const uint32_t w0 = pub[0], w1 = pub[1];
  A.n = w0 & 0xFFFFu; B.n = w0 >> 16; fits = (w1 & 1u) != 0u; side_total = w1 >> 1;
}

template <bool COMPACT> struct CompactState {};
template <> struct CompactState<true> { uint16_t* blk; uint32_t* scanw; uint32_t flip; };

// One workgroup: segment k of history h, origins 32 * sl .. 32 * sl + 31 (as one wavefront of K6).
// COMPACT (round 5: the first pass's form; 93 us of a 10k-op history's 928, profiles/r05_single_history_forms_first_device_run.json): a sub-round no longer walks all 2^gshift (config, open call) slots of
// every config -- of which a burst makes a child of one in three to five -- but, per block of 64 x NW configs: every thread counts
// ITS config's viable calls (a loop over the level's <= 64 open calls in LDS, no barrier), one workgroup scan numbers the children,
// and the insertion passes then take 64 x NW CHILDREN each: thread r finds its config by binary search over the block's 16-bit
// offsets and its call as the j-th viable one.  The children are the same and come in the same (config, call) order, so sets,
// statistics and records are those of the plain form; the barriers of a burst's sub-round drop from 3 per 64 x NW SLOTS to 2 per
// block + 3 per 64 x NW CHILDREN.  1 KB more LDS at NW = 8 (still two workgroups per CU).  Levels with more than 64 open calls keep
// the plain walk.
// SOLO (with COMPACT; likewise the first pass's form): a pass that fits one wavefront -- a level of at most 64
// configs, a sub-round of at most 64 (config, call) slots: 100 of the 187 steps of that critical workgroup, and nearly every step of
// the other 340 -- is run by wavefront 0 alone with wavefront barriers, the others waiting at ONE workgroup barrier for the new set
// sizes (insert2_solo, solo_exchange): one workgroup barrier instead of three, and a probe loop that waits for the longest chain
// of 64 lanes, not 512.
// RLX (round 5): the RELAXED sweep of a history with crashed calls (reach_table.h; oracle/sweep_ref.c, sweep_set_relaxed): a sub-round's
// pairs are COMPOUND steps -- a hop to a state the available classes reach, its open reads absorbed, then an open call (or nothing more)
template <uint32_t CAP, uint32_t NW, bool COMPACT = false, bool SOLO = false, bool RLX = false>
WV_DEV void segment(const SweepArgs& A, uint32_t* lds) {
  static_assert(!SOLO || (COMPACT && CAP < 0x10000u && Scratch<NW>::kBad + 5u <= Scratch<NW>::kWords), "solo passes: with the compact walk; two set sizes share a word");
  static_assert(!RLX || (!COMPACT && !SOLO), "the relaxed sweep: the plain walk");
  static_assert(64 * NW >= kCand, "a level's open calls are parked one per thread");
  static_assert(64 * NW <= 1024 && CAP <= 0x8000u, "thread numbers and entry numbers share a table word");
  static_assert(64 * NW < CAP, "a pass's provisional claims on top of a full set must leave the table (2 x CAP slots) an empty slot");
  constexpr uint32_t HS = 2 * CAP, T = 64 * NW;
  using S = Scratch<NW>;
  Ctx<NW> X;
  X.tid = wv::wg_thread(); X.lane = X.tid & 63u; X.wave = X.tid >> 6; X.parity = 0;
  const uint32_t tid = X.tid, lane = X.lane;
  const uint32_t w = wv::wg_index();
  // second pass (A.seg_list): the listed (history, segment, slice) triples only -- the ones that overflowed the first pass's sets
  uint32_t h, k, sl;
  if (A.seg_list) { h = A.seg_list[3 * w]; k = A.seg_list[3 * w + 1]; sl = A.seg_list[3 * w + 2]; }
  else { const uint32_t per = A.max_segs * kSweepSlices; h = w / per; const uint32_t r_ = w - h * per; k = r_ / kSweepSlices; sl = r_ % kSweepSlices; }
  if (h >= A.n_hist) return;
  if (A.shard_world > 1u && (k * kSweepSlices + sl) % A.shard_world != A.shard_rank) return;   // another rank's (its record stays zero)
  const uint32_t* cuts = A.cuts + (uint64_t)h * A.max_segs;
  SegResult* out = A.seg + ((uint64_t)h * A.max_segs + k) * kSweepSlices + sl;
  const uint32_t F0 = cuts[k];
  if (F0 == kInf) { if (tid == 0) out->status = kSegNone; return; }
  const Hist* H = A.hist + h;
  const BeamHist* B = A.bh + h;
  const uint32_t R = H->n_ret;
  uint32_t F1 = R;
  for (uint32_t k2 = k + 1; k2 < A.max_segs; k2++) { const uint32_t c = cuts[k2]; if (c != kInf) { F1 = c; break; } }
  const uint64_t op_off = H->op_off;
  const uint64_t off_off = B->off_off;
  const uint32_t* off = A.off + off_off;
  const uint32_t* ncr = A.ncr + off_off;
  const OpRec* lst = A.lst + B->lst_off;
  const OpRec* crashed = A.crashed + op_off;
  const uint64_t* twn = A.twn ? A.twn + B->lst_off : nullptr;
  const uint32_t V = A.vpad;
  const uint64_t* rdm = A.rdm ? A.rdm + op_off * V : nullptr;
  const uint8_t* slot8 = A.slot8 + slot8_off(op_off, h);
  const bool eager = (A.rules & kRuleEager) != 0u, twin = (A.rules & kRuleTwin) != 0u && twn != nullptr;

  // LDS carve-up
  Ent* sets = reinterpret_cast<Ent*>(lds);
  uint32_t* tab_nxt = reinterpret_cast<uint32_t*>(sets + 3 * CAP);
  uint32_t* tab_q = tab_nxt + HS;
  X.stage = reinterpret_cast<Ent*>(tab_q + HS);
  OpRec* cand = reinterpret_cast<OpRec*>(X.stage + T);
  uint64_t* cand_tw = reinterpret_cast<uint64_t*>(cand + kCand);
  uint64_t* row_a = cand_tw + kCand;          // read masks of the current level's front
  uint64_t* row_b = row_a + 32;               // ... of the next front
  uint16_t* expl = reinterpret_cast<uint16_t*>(row_b + 32);   // entries of `cur` that still need X
  uint32_t* Mrel = reinterpret_cast<uint32_t*>(expl + CAP);   // 32 x 4 words: origin -> origin ids of the next segment
  X.ws = Mrel + 128;
  CompactState<COMPACT> cs;         // COMPACT: where each config of the block's children start, the scan's words (else: nothing at all)
  if constexpr (COMPACT) { cs.blk = reinterpret_cast<uint16_t*>(X.ws + S::kWords); cs.scanw = reinterpret_cast<uint32_t*>(cs.blk + T); cs.flip = 0; }
  for (uint32_t i = tid; i < 2 * HS; i += T) tab_nxt[i] = 0u;
  for (uint32_t i = tid; i < 128u; i += T) Mrel[i] = 0u;
  for (uint32_t i = tid; i < S::kWords; i += T) X.ws[i] = 0u;
  wv::wg_barrier();
  uint32_t gen_nxt = 0, gen_q = 0;

  uint32_t status = kSegOk;
  uint64_t configs_total = 0, probes = 0;                     // probes: this wavefront's (summed at the end)
  // RLX: this history's reach table and the epoch of the current front
  const uint32_t* rch = nullptr;
  uint32_t n_ep = 0, ep = 0;
  if constexpr (RLX) { const uint32_t* hd = A.reach_hdr + 2u * h; rch = A.reach + hd[0]; n_ep = hd[1]; }
  uint32_t subrounds = 0, max_level = 0, last_level = F0;     // last_level: thread o < 32 keeps origin o's

  // ---- per-front scalars, 64 fronts at a time: lane l of EVERY wavefront holds those of front wbase + l
  uint32_t wbase = F0, w_off = 0, w_ncr = 0, w_px = 0;
  auto load_window = [&](uint32_t base) {
    wbase = base;
    const uint32_t f = base + lane;
    w_off = off[f < R ? f : R];
    w_ncr = ncr[f < R - 1u ? f : R - 1u];
    w_px = (uint32_t)slot8[f < R + 15u ? f : R + 15u];
  };
  load_window(F0);
  auto need_window = [&](uint32_t F) { if (F + 2u - wbase > 63u) load_window(F); };   // F, F+1, F+2 must be inside
  auto off_at = [&](uint32_t F) -> uint32_t { return wv::readlane(w_off, F - wbase); };
  auto ncr_at = [&](uint32_t F) -> uint32_t { return wv::readlane(w_ncr, F - wbase); };
  auto px_at = [&](uint32_t F) -> uint32_t { return wv::readlane(w_px, F - wbase); };

  auto load_row = [&](uint64_t* row, uint32_t F) {
    if (tid < 32) row[tid] = (eager && tid < V && F < R) ? rdm[(uint64_t)F * V + tid] : 0ull;
  };
  // a crashed call's twins: every live call with its effect and the crashed ones before it (the level's records are in LDS)
  auto crashed_twins = [&](uint32_t nlive, uint32_t C) {
    if (!(twin && C > nlive)) return;
    for (uint32_t c = nlive + tid; c < C; c += T) {
      const OpRec y = cand[c];
      const uint32_t yf = y.f_slot & 0xFFu;
      uint64_t m = 0;
      if (yf == TBC_F_WRITE || yf == TBC_F_CAS)
        for (uint32_t d = 0; d < c; d++) {
          const OpRec z = cand[d];
          if ((z.f_slot & 0xFFu) == yf && z.a == y.a && (yf != TBC_F_CAS || z.b == y.b)) m |= 1ull << ((z.f_slot >> 8) & 63u);
        }
      cand_tw[c] = m;
    }
    wv::wg_barrier();
  };
  // the level's open calls into LDS (not prefetched: the first level of a segment, the first front of the next segment)
  auto load_cands = [&](uint32_t F, uint32_t& nlive, uint32_t& C) -> bool {
    const uint32_t o0 = off_at(F), o1 = off_at(F + 1u), nc = ncr_at(F);
    nlive = o1 - o0; C = nlive + nc;
    if (C > kCand) return false;
    if (tid < C) {
      cand[tid] = tid < nlive ? lst[o0 + tid] : crashed[tid - nlive];
      cand_tw[tid] = (twin && tid < nlive) ? twn[o0 + tid] : 0ull;
    }
    wv::wg_barrier();
    return true;
  };

  // ---- origins (K6: lane = id; here thread = id, the first 32 of the workgroup)
  Ent* cur_e = sets; Ent* nxt_e = sets + CAP; Ent* q_e = sets + 2 * CAP;
  Build cur, nxt, q, none;
  none.e = sets; none.tab = tab_q; none.n = 0; none.gen = 0;
  nxt = none; q = none;
  uint32_t n_org = 0, so_ = 0, st_ = 0;
  {
    build_begin<HS, NW>(cur, cur_e, tab_q, gen_q, X);
    load_row(row_a, F0);
    uint32_t nlive = 0, C = 0;
    const bool okc = load_cands(F0, nlive, C);              // (its barrier also publishes row_a)
    if (!okc) wv::wg_barrier();
    if (okc) crashed_twins(nlive, C);
    bool act = false; uint32_t st = (uint32_t)A.init_state; uint64_t m = 0;
    if (!okc) status = kSegOverflow;
    else if (k == 0) {
      act = tid == 0 && sl == 0;
      if (eager) m |= row_a[0] | row_a[rdm_index((int32_t)st, V)];
    } else {
      const uint32_t id = 32u * sl + tid;
      act = tid < 32u && nlive <= 6u && id < (A.n_dom << nlive);
      const uint32_t qd = id >> nlive;
      st = qd == 0 ? (uint32_t)TBC_NIL : qd - 1u;
      for (uint32_t c = 0; c < nlive && c < 6u; c++) if ((id >> c) & 1u) m |= 1ull << ((cand[c].f_slot >> 8) & 63u);
      if (eager) act = act && ((row_a[0] | row_a[rdm_index((int32_t)st, V)]) & ~m) == 0ull;     // in normal form already?
    }
    Build unused = none;
    if (status == kSegOk && !insert2<CAP, NW>(cur, unused, act ? 1u : 0u, (uint32_t)m, (uint32_t)(m >> 32), st, 1u << (tid & 31u), false, so_, st_, X)) status = kSegOverflow;
  }
  n_org = cur.n;
  if (n_org == 0 && status == kSegOk) { if (tid == 0) out->status = kSegNone; return; }   // none of these ids is a config
  if (F0 + 1u < R) load_row(row_b, F0 + 1u); else if (tid < 32) row_b[tid] = 0ull;
  wv::wg_barrier();

  // ---- levels
  for (uint32_t F = F0; F < F1 && status == kSegOk; F++) {
    need_window(F);
    const uint32_t px = px_at(F);
    const uint32_t C = (off_at(F + 1u) - off_at(F)) + ncr_at(F);
    // ---- request level F+1 now (records, twin masks) and the read masks of front F+2: parked in registers until this
    // level's records are dead
    const bool pre = F + 1u < F1;
    OpRec p_rec{0, kFNone, 0, 0};
    uint64_t p_tw = 0ull, p_row = 0ull;
    uint32_t p_nlive = 0, p_C = 0;
    if (pre) {
      const uint32_t o0n = off_at(F + 1u), o1n = off_at(F + 2u);
      p_nlive = o1n - o0n; p_C = p_nlive + ncr_at(F + 1u);
      if (tid < p_C && p_C <= kCand) {
        p_rec = tid < p_nlive ? lst[o0n + tid] : crashed[tid - p_nlive];
        if (twin && tid < p_nlive) p_tw = twn[o0n + tid];
      }
      if (tid < 32 && eager && tid < V && F + 2u < R) p_row = rdm[(uint64_t)(F + 2u) * V + tid];
    }
    const uint64_t xbit = 1ull << (px & 63u);
    build_begin<HS, NW>(nxt, nxt_e, tab_nxt, gen_nxt, X);
    // sub-round 0: a config that has X linearized passes the completion -- X's bit is cleared and the reads open at
    // the next front are absorbed; the others are listed for expansion
    uint32_t n_exp = 0;
    bool solo0 = false;
    if constexpr (SOLO) solo0 = cur.n <= 64u;
    if (solo0) {                       // the whole level is one wavefront's: wavefront 0 passes it alone
      Build unused = none;
      bool fits = true;
      uint32_t stot = 0;
      if (X.wave == 0u) {
        const uint32_t i = tid;
        const bool val = i < cur.n;
        const Ent e = val ? cur.e[i] : Ent{0, 0, 0, 0};
        const uint64_t m = mask_of(e);
        const bool has = val && (m & xbit) != 0ull;
        uint64_t m2 = m & ~xbit;
        if (eager) m2 |= row_b[0] | row_b[rdm_index((int32_t)e.st, V)];
        uint32_t soff = 0;
        fits = insert2_solo<CAP, NW>(nxt, unused, has ? 1u : 0u, (uint32_t)m2, (uint32_t)(m2 >> 32), e.st, e.org, val && !has, soff, stot, X);
        if (val && !has) expl[soff] = (uint16_t)i;
      }
      solo_exchange<NW>(nxt, unused, fits, stot, X);           // (its barrier: the list is complete, too)
      if (!fits) status = kSegOverflow;
      n_exp = stot;
    } else {
    for (uint32_t base = 0; base < cur.n && status == kSegOk; base += T) {
      const uint32_t i = base + tid;
      const bool val = i < cur.n;
      const Ent e = val ? cur.e[i] : Ent{0, 0, 0, 0};
      const uint64_t m = mask_of(e);
      const bool has = val && (m & xbit) != 0ull;
      uint64_t m2 = m & ~xbit;
      if (eager) m2 |= row_b[0] | row_b[rdm_index((int32_t)e.st, V)];
      Build unused = none;
      uint32_t soff = 0, stot = 0;
      if (!insert2<CAP, NW>(nxt, unused, has ? 1u : 0u, (uint32_t)m2, (uint32_t)(m2 >> 32), e.st, e.org, val && !has, soff, stot, X)) status = kSegOverflow;
      if (val && !has) expl[n_exp + soff] = (uint16_t)i;
      n_exp += stot;
    }
    wv::wg_barrier();                                          // the list is complete
    }
    // sub-rounds: expand what still needs X, one (config, open call) pair per thread, 2^gshift pairs per config.  Children
    // that have X go to level F+1, the others to the next sub-round's set -- one probe loop for both.
    uint32_t gshift = 0, ntm = 0;
    if constexpr (RLX) { while (ep + 1u < n_ep && rch[ep + 1u] <= F) ep++; ntm = rch[n_ep + ep]; }
    const uint32_t pairs = RLX ? (1u + ntm) * (C + 1u) : C;         // RLX: (no hop or the t-th reachable state) x (an open call or nothing more)
    while ((1u << gshift) < pairs) gshift++;
    const Ent* src = cur.e;
    uint32_t n_src = n_exp;
    bool via_list = true;
    Ent* dst_e = q_e; Ent* dst_other = cur_e;       // `cur` is dead once its own expansion is done
    while (n_src != 0 && status == kSegOk) {
      subrounds++;
      build_begin<HS, NW>(q, dst_e, tab_q, gen_q, X);
      const uint32_t total = n_src << gshift;
      {
       if constexpr (COMPACT) {
        if (C <= 64u && total > 2u * T) {        // (a sub-round of one or two plain passes keeps them: the block's two barriers would not pay)
          // is open call kc a viable step from config (m, st)?  (the same test as the plain walk's)
          const auto viable_at = [&](uint64_t m, int32_t st, uint32_t kc) -> bool {
            const OpRec y = cand[kc];
            const uint32_t yf = y.f_slot & 0xFFu, ys = (y.f_slot >> 8) & 63u;
            return !((m >> ys) & 1ull) && !(eager && yf == TBC_F_READ) && (cand_tw[kc] & ~m) == 0ull && reg_ok(st, yf, y.a);
          };
          for (uint32_t cb = 0; cb < n_src && status == kSegOk; cb += T) {
            const uint32_t nb = n_src - cb < T ? n_src - cb : T;          // configs of this block
            // every thread counts its config's children
            uint32_t cnt = 0;
            if (tid < nb) {
              const Ent e = src[via_list ? (uint32_t)expl[cb + tid] : cb + tid];
              const uint64_t m = mask_of(e);
              for (uint32_t kc = 0; kc < C; kc++) cnt += viable_at(m, (int32_t)e.st, kc) ? 1u : 0u;
            }
            WV_UNROLL
            for (uint32_t bit = 0; bit < 7u; bit++) probes += (uint64_t)__builtin_popcountll(wv::ballot(((cnt >> bit) & 1u) != 0u)) << bit;
            uint32_t tot = 0;
            const uint32_t my_off = wg_scan_excl<NW>(cnt, tot, cs.scanw, cs.flip++, lane, X.wave);
            cs.blk[tid] = (uint16_t)my_off;                                 // (<= 64 x 512: fits; threads past the block: = tot)
            wv::wg_barrier();                                             // the offsets are visible
            for (uint32_t pb = 0; pb < tot && status == kSegOk; pb += T) {
              const uint32_t r = pb + tid;
              const bool val = r < tot;
              uint32_t lo_ = 0, hi_ = T;                                  // the last config whose children start at or before r
              while (hi_ - lo_ > 1u) { const uint32_t mid = (lo_ + hi_) >> 1; if ((uint32_t)cs.blk[mid] <= r) lo_ = mid; else hi_ = mid; }
              const uint32_t ci = val ? lo_ : 0u;
              const Ent e = val ? src[via_list ? (uint32_t)expl[cb + ci] : cb + ci] : Ent{0, 0, 0, 0};
              const uint64_t m = mask_of(e);
              const int32_t st = (int32_t)e.st;
              uint32_t j = val ? r - (uint32_t)cs.blk[ci] : 0u, kc = 0;
              if (val) for (;; kc++) { if (viable_at(m, st, kc)) { if (j == 0u) break; j--; } }
              const OpRec y = val ? cand[kc] : OpRec{0, kFNone, 0, 0};
              const uint32_t yf = y.f_slot & 0xFFu, ys = (y.f_slot >> 8) & 63u;
              const int32_t st2 = val ? reg_apply(st, yf, y.a, y.b) : st;
              uint64_t m2 = m | (1ull << ys);
              if (eager) m2 |= row_a[0] | row_a[rdm_index(st2, V)];
              const bool has = val && (m2 & xbit) != 0ull;
              if (has) { m2 &= ~xbit; if (eager) m2 |= row_b[0] | row_b[rdm_index(st2, V)]; }
              const uint32_t sel = val ? (has ? 1u : 2u) : 0u;
              if (!insert2<CAP, NW>(nxt, q, sel, (uint32_t)m2, (uint32_t)(m2 >> 32), (uint32_t)st2, e.org, false, so_, st_, X)) status = kSegOverflow;
            }
          }
        } else if (SOLO && total <= 64u) {          // the sub-round is one wavefront's: wavefront 0 walks and inserts it alone
          bool fits = true;
          uint32_t unused_total = 0;
          if (X.wave == 0u) {
            const uint32_t r = tid, ci = r >> gshift, kc = r & ((1u << gshift) - 1u);
            const bool val = r < total && kc < C;
            const Ent e = val ? src[via_list ? (uint32_t)expl[ci] : ci] : Ent{0, 0, 0, 0};
            const OpRec y = val ? cand[kc] : OpRec{0, kFNone, 0, 0};
            const uint64_t tw = val ? cand_tw[kc] : 0ull;
            const uint64_t m = mask_of(e);
            const uint32_t yf = y.f_slot & 0xFFu, ys = (y.f_slot >> 8) & 63u;
            const int32_t st = (int32_t)e.st;
            const bool viable = val && !((m >> ys) & 1ull) && !(eager && yf == TBC_F_READ) && (tw & ~m) == 0ull && reg_ok(st, yf, y.a);
            probes += (uint64_t)__builtin_popcountll(wv::ballot(viable));
            const int32_t st2 = viable ? reg_apply(st, yf, y.a, y.b) : st;
            uint64_t m2 = m | (1ull << ys);
            if (eager) m2 |= row_a[0] | row_a[rdm_index(st2, V)];
            const bool has = viable && (m2 & xbit) != 0ull;
            if (has) { m2 &= ~xbit; if (eager) m2 |= row_b[0] | row_b[rdm_index(st2, V)]; }
            const uint32_t sel = viable ? (has ? 1u : 2u) : 0u;
            fits = insert2_solo<CAP, NW>(nxt, q, sel, (uint32_t)m2, (uint32_t)(m2 >> 32), (uint32_t)st2, e.org, false, so_, st_, X);
          }
          solo_exchange<NW>(nxt, q, fits, unused_total, X);
          if (!fits) status = kSegOverflow;
        } else {
        for (uint32_t base = 0; base < total && status == kSegOk; base += T) {
          const uint32_t r = base + tid, ci = r >> gshift, kc = r & ((1u << gshift) - 1u);
          const bool val = r < total && kc < C;
          const Ent e = val ? src[via_list ? (uint32_t)expl[ci] : ci] : Ent{0, 0, 0, 0};
          const OpRec y = val ? cand[kc] : OpRec{0, kFNone, 0, 0};
          const uint64_t tw = val ? cand_tw[kc] : 0ull;
          const uint64_t m = mask_of(e);
          const uint32_t yf = y.f_slot & 0xFFu, ys = (y.f_slot >> 8) & 63u;
          const int32_t st = (int32_t)e.st;
          const bool viable = val && !((m >> ys) & 1ull) && !(eager && yf == TBC_F_READ) && (tw & ~m) == 0ull && reg_ok(st, yf, y.a);
          probes += (uint64_t)__builtin_popcountll(wv::ballot(viable));
          const int32_t st2 = viable ? reg_apply(st, yf, y.a, y.b) : st;
          uint64_t m2 = m | (1ull << ys);
          if (eager) m2 |= row_a[0] | row_a[rdm_index(st2, V)];
          const bool has = viable && (m2 & xbit) != 0ull;
          if (has) { m2 &= ~xbit; if (eager) m2 |= row_b[0] | row_b[rdm_index(st2, V)]; }
          const uint32_t sel = viable ? (has ? 1u : 2u) : 0u;
          if (!insert2<CAP, NW>(nxt, q, sel, (uint32_t)m2, (uint32_t)(m2 >> 32), (uint32_t)st2, e.org, false, so_, st_, X)) status = kSegOverflow;
        }
        }
       } else {
        for (uint32_t base = 0; base < total && status == kSegOk; base += T) {
          const uint32_t r = base + tid, ci = r >> gshift, kc = r & ((1u << gshift) - 1u);
          bool val = r < total && kc < pairs;
          const Ent e = val ? src[via_list ? (uint32_t)expl[ci] : ci] : Ent{0, 0, 0, 0};
          uint64_t m = mask_of(e);
          int32_t st = (int32_t)e.st;
          uint32_t yc = kc;
          bool alone = false, hop = false;             // RLX: the hop with nothing behind it; a real hop
          if constexpr (RLX) {
            const uint32_t ti = kc / (C + 1u);
            yc = kc - ti * (C + 1u);
            alone = yc == C;
            hop = ti != 0u;
            if (ti != 0u) {                            // the ti-th state this one reaches
              const uint32_t si = rdm_index(st, V);
              uint32_t rr = val ? (rch[2u * n_ep + 32u * ep + si] & ~(1u << si)) : 0u;
              val = val && (uint32_t)__builtin_popcount(rr) >= ti;
              for (uint32_t q2 = 1; q2 < ti; q2++) rr &= rr - 1u;
              const uint32_t t = rr ? (uint32_t)__builtin_ctz(rr) : 0u;
              st = t == 0u ? TBC_NIL : (int32_t)t - 1;
              const uint64_t m1 = eager ? (m | row_a[0] | row_a[t < V ? t : 0u]) : m;
              if (alone) val = val && m1 != m;         // (alone it must have absorbed a read: every child has more calls linearized)
              m = m1;
            } else if (alone) val = false;
          }
          const OpRec y = (val && !alone) ? cand[yc] : OpRec{0, kFNone, 0, 0};
          const uint64_t tw = (val && !alone) ? cand_tw[yc] : 0ull;
          const uint32_t yf = y.f_slot & 0xFFu, ys = (y.f_slot >> 8) & 63u;
          // (after a real hop only a :cas expecting the new state: the lazy rule -- a crashed call is worth a step only for a call that observes it)
          const bool viable = alone ? val : (val && !((m >> ys) & 1ull) && !(eager && yf == TBC_F_READ) && !(hop && yf != TBC_F_CAS) && (tw & ~m) == 0ull && reg_ok(st, yf, y.a));
          probes += (uint64_t)__builtin_popcountll(wv::ballot(viable));
          const int32_t st2 = (viable && !alone) ? reg_apply(st, yf, y.a, y.b) : st;
          uint64_t m2 = alone ? m : (m | (1ull << ys));
          if (eager) m2 |= row_a[0] | row_a[rdm_index(st2, V)];
          const bool has = viable && (m2 & xbit) != 0ull;
          if (has) { m2 &= ~xbit; if (eager) m2 |= row_b[0] | row_b[rdm_index(st2, V)]; }
          const uint32_t sel = viable ? (has ? 1u : 2u) : 0u;
          if (!insert2<CAP, NW>(nxt, q, sel, (uint32_t)m2, (uint32_t)(m2 >> 32), (uint32_t)st2, e.org, false, so_, st_, X)) status = kSegOverflow;
        }
       }
      }
      src = q.e; n_src = q.n; via_list = false;
      { Ent* t = dst_e; dst_e = dst_other; dst_other = t; }
    }
    if (status != kSegOk) break;
    // level F+1 is complete
    configs_total += nxt.n;
    max_level = max_level > nxt.n ? max_level : nxt.n;
    // SOLO: the prefetched level is parked in LDS here already -- nobody reads this level's records or rows any more (every pass of
    // it is behind a workgroup barrier) -- so that the barrier of the origins' word below publishes it too: one barrier a level fewer
    if constexpr (SOLO) {
      if (pre && p_C <= kCand) {
        if (tid < p_C) { cand[tid] = p_rec; cand_tw[tid] = p_tw; }
        { uint64_t* t = row_a; row_a = row_b; row_b = t; }
        if (tid < 32) row_b[tid] = p_row;
      }
    }
    {   // which origins are still alive
      const uint32_t lp = (F & 1u);
      uint32_t any = 0;
      for (uint32_t i = tid; i < nxt.n; i += T) any |= nxt.e[i].org;
      any = wv::wave_or32(any);
      if (lane == 0 && any) wv::lds_or32(&X.ws[S::kOrg + lp], any);
      if (tid == 0) X.ws[S::kOrg + (lp ^ 1u)] = 0u;           // (the other level's word: read before this level's barriers, written again after the next)
      wv::wg_barrier();
      any = X.ws[S::kOrg + lp];
      if (tid < 32 && ((any >> tid) & 1u)) last_level = F + 1;
    }
    {   // the set just built becomes `cur`; the other two are free
      Build t = cur; cur = nxt; nxt = t;
      Ent* old_cur = cur_e; cur_e = nxt_e; nxt_e = old_cur;     // q_e keeps its place
    }
    if (cur.n == 0) break;                           // nobody passes completion F
    // ---- park the prefetched level in LDS
    if (pre) {
      if (p_C > kCand) { status = kSegOverflow; break; }
      if constexpr (!SOLO) {          // (SOLO: parked above, published by the barrier of the origins' word)
      if (tid < p_C) { cand[tid] = p_rec; cand_tw[tid] = p_tw; }
      { uint64_t* t = row_a; row_a = row_b; row_b = t; }
      if (tid < 32) row_b[tid] = p_row;
      wv::wg_barrier();
      }
      crashed_twins(p_nlive, p_C);
    }
  }

  // ---- the relation this workgroup hands on: origin -> ids of the next segment's origin space
  if (status == kSegOk && cur.n != 0) {
    uint32_t no1 = 0;
    if (F1 != R) {
      uint32_t nl = 0, cc = 0;
      need_window(F1);
      const bool okc = load_cands(F1, nl, cc);
      if (!okc) wv::wg_barrier();
      if (!okc || nl > 6u) status = kSegOverflow;
      no1 = nl;
    }
    bool bad = false;
    for (uint32_t base = 0; base < cur.n && status == kSegOk; base += T) {
      const uint32_t i = base + tid;
      const bool val = i < cur.n;
      const Ent e = val ? cur.e[i] : Ent{0, 0, 0, 0};
      uint32_t org = val ? e.org : 0u;
      // last segment: word 0 = the final states reached, as bits (nil = bit 0, value v = bit v + 1)
      uint32_t id2 = V > 1u ? rdm_index((int32_t)e.st, 32u) : 0u;
      if (F1 != R) {
        const uint32_t sidx = e.st == (uint32_t)TBC_NIL ? 0u : e.st + 1u;
        if (val && sidx >= A.n_dom) { bad = true; org = 0u; }        // a state outside the domain: cannot be numbered
        const uint64_t m = mask_of(e);
        id2 = sidx << no1;
        for (uint32_t c = 0; c < no1; c++) if ((m >> ((cand[c].f_slot >> 8) & 63u)) & 1ull) id2 |= 1u << c;
      }
      while (org) { const uint32_t o = (uint32_t)__builtin_ctz(org); wv::lds_or32(&Mrel[o * kSweepSlices + (id2 >> 5)], 1u << (id2 & 31u)); org &= org - 1u; }
    }
    // (a state outside the domain anywhere in the workgroup ends the segment, as in K6)
    const uint64_t bb = wv::ballot(bad);
    if (lane == 0 && bb) wv::lds_or32(&X.ws[S::kBad], 1u);
    wv::wg_barrier();
    if (X.ws[S::kBad]) status = kSegOverflow;
  } else {
    wv::wg_barrier();
  }
  if (lane == 0) { X.ws[S::kProbes + 2 * X.wave] = (uint32_t)probes; X.ws[S::kProbes + 2 * X.wave + 1] = (uint32_t)(probes >> 32); }
  wv::wg_barrier();
  if (tid < 32) {
    WV_UNROLL
    for (uint32_t wd = 0; wd < kSweepSlices; wd++) out->M[tid][wd] = Mrel[tid * kSweepSlices + wd];
    out->last_level[tid] = last_level;
  }
  if (tid == 0) {
    uint64_t pr = 0;
    for (uint32_t w2 = 0; w2 < NW; w2++) pr += (uint64_t)X.ws[S::kProbes + 2 * w2] | ((uint64_t)X.ws[S::kProbes + 2 * w2 + 1] << 32);
    out->status = status; out->F0 = F0; out->F1 = F1; out->n_org = n_org;
    out->max_level = max_level; out->subrounds = subrounds; out->configs_total = configs_total; out->probes = pr;
    out->n_end = cur.n; out->end_state = (status == kSegOk && cur.n) ? cur.e[0].st : 0u;
  }
}

}  // namespace sweepwg
}  // namespace tbc
