// jit_sweep.hip -- K6: knossos.linear/analysis as a SEGMENTED LEVEL SWEEP (gfx950).
//
// Lowe's just-in-time linearization (the algorithm behind knossos.linear, SURVEY.md section 8a rows
// `knossos.linear/analysis + knossos.linear.config`): sweep the completions in order carrying the SET of
// reachable configs (mask of linearized open calls, model state); at the completion of call X every config
// must get X linearized -- configs that have it pass, the others are expanded over their open calls until X
// is in (sub-rounds), configs that cannot are dropped; an empty set = not linearizable at that completion.
// The config set is kept in the normal form of the two dominance rules (tbc_internal.h: eager reads, twin
// rule), which is what keeps it narrow: ~9 configs per level on the 10k-op / 64-process bench histories
// where the plain sweep carries ~60 and the plain depth-first search visits 2*10^5.
//
// What makes it a GPU algorithm is the cut into SEGMENTS.  A depth-first search of one history is a chain
// of >= 10^4 dependent steps whatever the hardware; the sweep's levels are just as
// dependent -- but a level set is small enough to be started from EVERY config that is possible at a front
// at all.  So the history is cut, in every window of T fronts, at the front where the fewest calls are open
// (sweep_cuts_kernel; at most m), the configs possible there -- (state of the domain) x (subset of the open
// calls), nd * 2^m <= 128 "origins", numbered arithmetically -- are dealt to wavefronts 32 at a time, every
// config carries the 32-bit set of its wavefront's origins it is reachable from (duplicates OR their sets;
// sub-rounds go by calls linearized, so a set is final before its config is expanded), and a wavefront hands
// on a 32 x 128 bit relation origin -> origin ids of the next segment.  The host composes the relations in
// order (a few hundred word operations).  One 10k-op history is then checked by ~350 wavefronts at once,
// each walking < 100 levels, instead of by one wavefront walking 10^4 rounds.
//
// Everything a level touches lives in LDS: three config sets (this level, the next, the sub-round set -- the
// sub-rounds alternate between the third set and the first, dead after its own expansion), two open-addressed
// hash tables of entry numbers (generation-tagged, never cleared), the level's open-call records and twin
// masks, the two read-mask rows.  HBM sees one streaming pass over the per-front lists pack_open built -- no
// visited set, no global atomics, no random access -- and that pass is PREFETCHED: the records, twin masks
// and read masks of level F+1 are requested at the top of level F and parked in registers, the per-front
// scalars (list offsets, crashed counts, completion slots) come 64 fronts at a time, so a level waits for LDS
// only.  Exact keys throughout (a hash only picks the slot).  Insertion: a lane claims an empty slot
// provisionally (CAS with its lane number), lanes that meet a provisional slot compare with the claimant's
// staged key; winners are appended in lane order; one probe loop serves both target sets of a round.
// Two instantiations: 512 configs per set (38 KB of LDS, four wavefronts per CU) for every segment, and 2,048
// (134 KB, one per CU) for the segments that overflowed the first.
//
// The schedule is specified in oracle/sweep_ref.c; verdict, failing op, and the sweep's own statistics
// (level sizes summed, largest level, expansions, sub-rounds) are compared bit for bit.  A set that
// outgrows its LDS capacity ends the segment with status OVERFLOW and the host hands that history to the
// wide depth-first kernel (wgl_beam.hip) -- which is what knossos.competition does with its two searches.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "tbc_internal.h"
#include "device_common.h"

namespace tbc {

namespace {

constexpr uint32_t kCand = kSweepCandMax;     // open calls per level
constexpr uint32_t kProv = 1u << 16;          // slot holds a lane number (this round's claimant), not an entry
constexpr uint32_t kGenShift = 17;

struct __attribute__((aligned(16))) Ent { uint32_t mlo, mhi, st, org; };

__device__ __forceinline__ void lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

template <uint32_t HS>
__device__ __forceinline__ uint32_t key_slot(uint32_t mlo, uint32_t mhi, uint32_t st) {
  uint32_t h = mlo * 0x9E3779B1u ^ mhi * 0x85EBCA77u ^ st * 0xC2B2AE3Du;
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 13;
  return h & (HS - 1u);
}

// One set being built: entries + its hash table + the table's generation.
struct Build {
  Ent* e;
  uint32_t* tab;
  uint32_t n;       // wave-uniform
  uint32_t gen;     // wave-uniform, != 0
};

// Start a new set in table `tab`: a new generation makes every old slot read as empty.
template <uint32_t HS>
__device__ __forceinline__ void build_begin(Build& b, Ent* e, uint32_t* tab, uint32_t& gen_counter, uint32_t lane) {
  gen_counter++;
  if (gen_counter >= (1u << (32 - kGenShift))) {       // generation wrapped: really clear
    for (uint32_t i = lane; i < HS; i += 64) tab[i] = 0u;
    gen_counter = 1;
    lds_sync();
  }
  b.e = e; b.tab = tab; b.n = 0; b.gen = gen_counter;
}

// Insert up to 64 configs, one per lane, each into set A (sel 1) or set B (sel 2), OR-ing the origin sets of
// equal keys.  Returns false if a set outgrew CAP.
template <uint32_t CAP>
__device__ __forceinline__ bool build_insert2(Build& A, Build& B, uint32_t sel, uint32_t mlo, uint32_t mhi, uint32_t st,
                                              uint32_t org, Ent* stage, uint32_t lane) {
  constexpr uint32_t HS = 2 * CAP;
  if (!__ballot(sel != 0u)) return true;
  stage[lane] = Ent{mlo, mhi, st, sel ? org : 0u};
  lds_sync();
  uint32_t* const tab = sel == 2u ? B.tab : A.tab;
  Ent* const ent = sel == 2u ? B.e : A.e;
  const uint32_t gen = sel == 2u ? B.gen : A.gen, gtag = gen << kGenShift;
  uint32_t h = key_slot<HS>(mlo, mhi, st), mine = 0;
  bool pend = sel != 0u, won = false;
  while (__ballot(pend)) {
    if (pend) {
      uint32_t s = tab[h];
      if ((s >> kGenShift) != gen) {                      // empty: claim it with the lane number
        const uint32_t old = atomicCAS(&tab[h], s, gtag | kProv | lane);
        if (old == s) { won = true; mine = h; pend = false; }
        else s = old;                                     // claimed in this very step by another lane
      }
      if (pend) {
        const Ent k = (s & kProv) ? stage[s & 63u] : ent[s & 0xFFFFu];
        if (k.mlo == mlo && k.mhi == mhi && k.st == st) {
          if (s & kProv) atomicOr(&stage[s & 63u].org, org); else atomicOr(&ent[s & 0xFFFFu].org, org);
          pend = false;
        } else {
          h = (h + 1u) & (HS - 1u);
        }
      }
    }
  }
  lds_sync();
  const uint64_t wa = __ballot(won && sel == 1u), wb = __ballot(won && sel == 2u);
  const uint64_t below = (1ull << lane) - 1ull;
  const uint32_t idx = sel == 2u ? B.n + (uint32_t)__popcll(wb & below) : A.n + (uint32_t)__popcll(wa & below);
  const uint32_t ta = A.n + (uint32_t)__popcll(wa), tb = B.n + (uint32_t)__popcll(wb);
  if (ta > CAP || tb > CAP) return false;
  if (won) {
    ent[idx] = Ent{mlo, mhi, st, stage[lane].org};
    tab[mine] = gtag | idx;
  }
  A.n = ta; B.n = tb;
  lds_sync();
  return true;
}

__device__ __forceinline__ uint64_t mask_of(const Ent& e) { return (uint64_t)e.mlo | ((uint64_t)e.mhi << 32); }

// LDS words per wavefront: 3 sets, 2 tables, stage, open-call records + twin masks, 2 read-mask rows, expansion list, relation
template <uint32_t CAP>
constexpr uint32_t sweep_lds_words() { return 3 * CAP * 4 + 2 * (2 * CAP) + 64 * 4 + kCand * 4 + kCand * 2 + 2 * 32 * 2 + CAP / 2 + 128 + 128 * 4 + 128; }

}  // namespace

// ---- cut placement: thread k of history h takes, in the window [k*T, (k+1)*T), the front with the fewest calls
// open among those where no crashed call is open (the first such front), if that is <= m (oracle/sweep_ref.c)
__global__ __launch_bounds__(256) void sweep_cuts_kernel(SweepArgs A) {
  const uint32_t h = blockIdx.x;
  if (h >= A.n_hist) return;
  const Hist* H = A.hist + h;
  const BeamHist* B = A.bh + h;
  const uint32_t R = H->n_ret;
  const uint32_t* off = A.off + B->off_off;
  const uint32_t* ncr = A.ncr + B->off_off;
  uint32_t* cuts = A.cuts + (uint64_t)h * A.max_segs;
  for (uint32_t k = threadIdx.x; k < A.max_segs; k += 256) {
    uint32_t cut = kInf;
    if (k == 0) cut = (H->status == 0 && B->status == 0 && R != 0) ? 0u : kInf;
    else if (A.seg_target && H->status == 0 && B->status == 0) {
      const uint64_t lo = (uint64_t)k * A.seg_target;
      const uint64_t hi = lo + A.seg_target < R ? lo + A.seg_target : R;
      uint32_t best = kInf;
      for (uint64_t F = lo; F < hi; F++) {
        const uint32_t no = off[F + 1] - off[F];
        if (ncr[F] == 0u && no < best) { best = no; cut = (uint32_t)F; }
      }
      if (best > A.cut_open) cut = kInf;
    }
    cuts[k] = cut;
  }
}

// ---- the sweep: one wavefront per (history, cut, 32 origins)
template <uint32_t CAP>
__global__ __launch_bounds__(64) void jit_sweep_kernel(SweepArgs A) {
  constexpr uint32_t HS = 2 * CAP;
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const uint32_t lane = threadIdx.x;
  const uint32_t w = blockIdx.x;
  // dump pass (A.dump_cfg): ONE wavefront re-sweeps segment (dump_hist, dump_seg) up to level stop_level and
  // writes that level's configs reachable from the live origins -- knossos.linear's :configs of an invalid verdict.
  // second pass (A.seg_list): the listed (history, segment) pairs only (the ones that overflowed the small sets)
  const bool dump = A.dump_cfg != nullptr;
  uint32_t h, k, sl;
  if (dump) { h = A.dump_hist; k = A.dump_seg; sl = A.dump_slice; }
  else if (A.seg_list) { h = rfl(A.seg_list[3 * w]); k = rfl(A.seg_list[3 * w + 1]); sl = rfl(A.seg_list[3 * w + 2]); }
  else { const uint32_t per = A.max_segs * kSweepSlices; h = w / per; const uint32_t r = w - h * per; k = r / kSweepSlices; sl = r % kSweepSlices; }
  if (h >= A.n_hist) return;
  if (!dump && A.shard_world > 1u && (k * kSweepSlices + sl) % A.shard_world != A.shard_rank) return;   // another rank's (its record stays zero)
  const uint32_t* cuts = A.cuts + (uint64_t)h * A.max_segs;
  SegResult* out = A.seg + ((uint64_t)h * A.max_segs + k) * kSweepSlices + sl;
  const uint32_t F0 = rfl(cuts[k]);
  if (F0 == kInf) { if (lane == 0 && !dump) out->status = kSegNone; return; }
  const Hist* H = A.hist + h;
  const BeamHist* B = A.bh + h;
  const uint32_t R = rfl(H->n_ret);
  uint32_t F1 = R;
  for (uint32_t k2 = k + 1; k2 < A.max_segs; k2++) { const uint32_t c = rfl(cuts[k2]); if (c != kInf) { F1 = c; break; } }
  const uint64_t op_off = ru64(H->op_off);
  const uint64_t off_off = ru64(B->off_off);
  const uint32_t* off = A.off + off_off;
  const uint32_t* ncr = A.ncr + off_off;
  const OpRec* lst = A.lst + ru64(B->lst_off);
  const OpRec* crashed = A.crashed + op_off;
  const uint64_t* twn = A.twn ? A.twn + ru64(B->lst_off) : nullptr;
  const uint32_t V = A.vpad;
  const uint64_t* rdm = A.rdm ? A.rdm + op_off * V : nullptr;
  const uint8_t* slot8 = A.slot8 + slot8_off(op_off, h);
  const bool eager = (A.rules & kRuleEager) != 0u, twin = (A.rules & kRuleTwin) != 0u && twn != nullptr;
  Model model{A.model_kind, A.table, A.n_classes, A.pool_vals, 0, A.n_keys};

  // LDS carve-up
  Ent* sets = reinterpret_cast<Ent*>(lds);
  uint32_t* tab_nxt = reinterpret_cast<uint32_t*>(sets + 3 * CAP);
  uint32_t* tab_q = tab_nxt + HS;
  Ent* stage = reinterpret_cast<Ent*>(tab_q + HS);
  OpRec* cand = reinterpret_cast<OpRec*>(stage + 64);
  uint64_t* cand_tw = reinterpret_cast<uint64_t*>(cand + kCand);
  uint64_t* row_a = cand_tw + kCand;          // read masks of the current level's front
  uint64_t* row_b = row_a + 32;               // ... of the next front
  uint16_t* expl = reinterpret_cast<uint16_t*>(row_b + 32);   // entries of `cur` that still need X
  uint32_t* Mrel = reinterpret_cast<uint32_t*>(expl + CAP);   // 32 x 4 words: origin -> origin ids of the next segment
  Ent* wq = reinterpret_cast<Ent*>(Mrel + 128);               // ring of 128 children waiting for a full insert round (bursts)
  uint32_t* wq_sel = reinterpret_cast<uint32_t*>(wq + 128);
  for (uint32_t i = lane; i < 2 * HS; i += 64) tab_nxt[i] = 0u;
  Mrel[lane] = 0u; Mrel[64 + lane] = 0u;
  lds_sync();
  uint32_t gen_nxt = 0, gen_q = 0;

  uint32_t status = kSegOk;
  uint64_t configs_total = 0, probes = 0;
  uint32_t subrounds = 0, max_level = 0, last_level = F0;   // last_level: lane o < 32 keeps origin o's

  // ---- per-front scalars, 64 fronts at a time: lane l holds those of front wbase + l
  uint32_t wbase = F0, w_off = 0, w_ncr = 0, w_px = 0;
  auto load_window = [&](uint32_t base) {
    wbase = base;
    const uint32_t f = base + lane;
    w_off = off[min(f, R)];
    w_ncr = ncr[min(f, R - 1u)];
    w_px = (uint32_t)slot8[min(f, R + 15u)];
  };
  load_window(F0);
  auto need_window = [&](uint32_t F) { if (F + 2u - wbase > 63u) load_window(F); };   // F, F+1, F+2 must be inside
  auto off_at = [&](uint32_t F) -> uint32_t { return rl(w_off, F - wbase); };
  auto ncr_at = [&](uint32_t F) -> uint32_t { return rl(w_ncr, F - wbase); };
  auto px_at = [&](uint32_t F) -> uint32_t { return rl(w_px, F - wbase); };

  // read-mask row of front F into LDS (lane vi loads entry vi); all zero without the rule
  auto load_row = [&](uint64_t* row, uint32_t F) {
    if (lane < 32) row[lane] = (eager && lane < V && F < R) ? rdm[(uint64_t)F * V + lane] : 0ull;
  };
  // a crashed call's twins: every live call with its effect and the crashed ones before it (the level's records are in LDS)
  auto crashed_twins = [&](uint32_t nlive, uint32_t C) {
    if (!(twin && C > nlive)) return;
    for (uint32_t c = nlive + lane; c < C; c += 64) {
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
    lds_sync();
  };
  // the level's open calls into LDS (not prefetched: first level of a segment, and calls beyond the 64th)
  auto load_cands = [&](uint32_t F, uint32_t from, uint32_t& nlive, uint32_t& C) -> bool {
    const uint32_t o0 = off_at(F), o1 = off_at(F + 1u), nc = ncr_at(F);
    nlive = o1 - o0; C = nlive + nc;
    if (C > kCand) return false;
    for (uint32_t c = from + lane; c < C; c += 64) {
      cand[c] = c < nlive ? lst[o0 + c] : crashed[c - nlive];
      cand_tw[c] = (twin && c < nlive) ? twn[o0 + c] : 0ull;
    }
    lds_sync();
    return true;
  };

  // ---- origins: segment 0 starts from the initial config; every other segment from every config possible at its
  // first front.  Id = state index * 2^no + (bit c = the c-th call open there is linearized); this wavefront takes
  // ids 32 * sl .. 32 * sl + 31, lane = id, and only the ids that are configs -- in normal form -- count.
  Ent* cur_e = sets; Ent* nxt_e = sets + CAP; Ent* q_e = sets + 2 * CAP;
  Build cur, nxt, q, none;
  none.e = sets; none.tab = tab_q; none.n = 0; none.gen = 0;
  uint32_t n_org = 0;
  {
    build_begin<HS>(cur, cur_e, tab_q, gen_q, lane);
    load_row(row_a, F0);
    uint32_t nlive = 0, C = 0;
    bool okc = load_cands(F0, 0u, nlive, C);
    if (okc) crashed_twins(nlive, C);
    bool act = false; uint32_t st = (uint32_t)A.init_state; uint64_t m = 0;
    if (!okc) status = kSegOverflow;
    else if (k == 0) {
      act = lane == 0 && sl == 0;
      if (eager) m |= row_a[0] | row_a[rdm_index((int32_t)st, V)];
    } else {
      const uint32_t id = 32u * sl + lane;
      act = lane < 32u && nlive <= 6u && id < (A.n_dom << nlive);
      const uint32_t qd = id >> nlive;
      st = qd == 0 ? (uint32_t)TBC_NIL : qd - 1u;
      for (uint32_t c = 0; c < nlive && c < 6u; c++) if ((id >> c) & 1u) m |= 1ull << ((cand[c].f_slot >> 8) & 63u);
      if (eager) act = act && ((row_a[0] | row_a[rdm_index((int32_t)st, V)]) & ~m) == 0ull;     // in normal form already?
    }
    Build unused = none;
    if (!build_insert2<CAP>(cur, unused, act ? 1u : 0u, (uint32_t)m, (uint32_t)(m >> 32), st, 1u << (lane & 31u), stage, lane)) status = kSegOverflow;
  }
  n_org = cur.n;
  if (n_org == 0 && status == kSegOk) { if (lane == 0 && !dump) out->status = kSegNone; return; }   // none of these ids is a config
  if (F0 + 1u < R) load_row(row_b, F0 + 1u); else if (lane < 32) row_b[lane] = 0ull;
  lds_sync();

  // ---- levels
  for (uint32_t F = F0; F < F1 && status == kSegOk; F++) {
    if (dump && F == A.stop_level) {
      uint32_t nd = rfl(*A.dump_count);           // the slices of one segment append one after the other (stream order)
      for (uint32_t base = 0; base < cur.n; base += 64) {
        const uint32_t i = base + lane;
        const Ent e = i < cur.n ? cur.e[i] : Ent{0, 0, 0, 0};
        const bool hit = i < cur.n && (e.org & A.live_mask) != 0u;
        const uint64_t hb = __ballot(hit);
        const uint32_t pos = nd + (uint32_t)__popcll(hb & ((1ull << lane) - 1ull));
        if (hit && pos < kCfgCap) {
          A.dump_cfg[3 * pos] = (uint64_t)(F + 1u) | ((uint64_t)e.st << 32);
          A.dump_cfg[3 * pos + 1] = mask_of(e);
          A.dump_cfg[3 * pos + 2] = (uint64_t)TBC_NO_OP;
        }
        nd += (uint32_t)__popcll(hb);
      }
      if (lane == 0) *A.dump_count = nd;
      return;
    }
    need_window(F);
    const uint32_t px = px_at(F);
    const uint32_t nlive = off_at(F + 1u) - off_at(F), C = nlive + ncr_at(F);
    // ---- request level F+1 now (records, twin masks) and the read masks of front F+2: they arrive while this level
    // works in LDS and are parked in registers until its records are dead
    const bool pre = F + 1u < F1;
    OpRec p_rec{0, kFNone, 0, 0};
    uint64_t p_tw = 0ull, p_row = 0ull;
    uint32_t p_nlive = 0, p_C = 0;
    if (pre) {
      const uint32_t o0n = off_at(F + 1u), o1n = off_at(F + 2u);
      p_nlive = o1n - o0n; p_C = p_nlive + ncr_at(F + 1u);
      if (lane < p_C) {
        p_rec = lane < p_nlive ? lst[o0n + lane] : crashed[lane - p_nlive];
        if (twin && lane < p_nlive) p_tw = twn[o0n + lane];
      }
      if (lane < 32 && eager && lane < V && F + 2u < R) p_row = rdm[(uint64_t)(F + 2u) * V + lane];
    }
    const uint64_t xbit = 1ull << (px & 63u);
    build_begin<HS>(nxt, nxt_e, tab_nxt, gen_nxt, lane);
    // sub-round 0: a config that has X linearized passes the completion -- X's bit is cleared and the reads open at
    // the next front are absorbed; the others are remembered for expansion
    uint32_t n_exp = 0;
    for (uint32_t base = 0; base < cur.n && status == kSegOk; base += 64) {
      const uint32_t i = base + lane;
      const bool val = i < cur.n;
      const Ent e = val ? cur.e[i] : Ent{0, 0, 0, 0};
      const uint64_t m = mask_of(e);
      const bool has = val && (m & xbit) != 0ull;
      uint64_t m2 = m & ~xbit;
      if (eager) m2 |= row_b[0] | row_b[rdm_index((int32_t)e.st, V)];
      Build unused = none;
      if (!build_insert2<CAP>(nxt, unused, has ? 1u : 0u, (uint32_t)m2, (uint32_t)(m2 >> 32), e.st, e.org, stage, lane)) status = kSegOverflow;
      const uint64_t nb = __ballot(val && !has);
      if (val && !has) expl[n_exp + (uint32_t)__popcll(nb & ((1ull << lane) - 1ull))] = (uint16_t)i;
      n_exp += (uint32_t)__popcll(nb);
    }
    lds_sync();
    // sub-rounds: expand what still needs X, (config, open call) pair per lane, G lanes per config.  Children that
    // have X go to level F+1, the others to the next sub-round's set -- one probe loop for both.
    uint32_t gshift = 0;
    while ((1u << gshift) < C) gshift++;
    const Ent* src = cur.e;
    uint32_t n_src = n_exp;
    bool via_list = true;
    Ent* dst_e = q_e; Ent* dst_other = cur_e;       // `cur` is dead once its own expansion is done
    while (n_src != 0 && status == kSegOk) {
      subrounds++;
      build_begin<HS>(q, dst_e, tab_q, gen_q, lane);
      const uint32_t total = n_src << gshift;
      // A burst of concurrency makes sub-rounds of thousands of pairs of which a third yield a child.  The insertion
      // (staging, probe loop, commit: three LDS synchronisations) is the expensive half of a round, so a long sub-round
      // is pipelined: children are compacted into a ring and inserted 64 at a time; a short one inserts directly.
      const bool pipelined = total > 64u;
      uint32_t qh = 0, qn = 0;
      for (uint32_t base = 0; base < total && status == kSegOk; base += 64) {
        const uint32_t r = base + lane, ci = r >> gshift, kc = r & ((1u << gshift) - 1u);
        const bool val = r < total && kc < C;
        const Ent e = val ? src[via_list ? (uint32_t)expl[ci] : ci] : Ent{0, 0, 0, 0};
        const OpRec y = val ? cand[kc] : OpRec{0, kFNone, 0, 0};
        const uint64_t tw = val ? cand_tw[kc] : 0ull;
        const uint64_t m = mask_of(e);
        const uint32_t yf = y.f_slot & 0xFFu, ys = (y.f_slot >> 8) & 63u;
        const int32_t st = (int32_t)e.st;
        const bool viable = val && !((m >> ys) & 1ull) && !(eager && yf == TBC_F_READ) && (tw & ~m) == 0ull &&
                            model.ok(st, yf, y.a, y.b);
        probes += (uint64_t)__popcll(__ballot(viable));
        const int32_t st2 = viable ? model.apply(st, yf, y.a, y.b) : st;
        uint64_t m2 = m | (1ull << ys);
        if (eager) m2 |= row_a[0] | row_a[rdm_index(st2, V)];
        const bool has = viable && (m2 & xbit) != 0ull;
        if (has) { m2 &= ~xbit; if (eager) m2 |= row_b[0] | row_b[rdm_index(st2, V)]; }
        const uint32_t sel = viable ? (has ? 1u : 2u) : 0u;
        if (!pipelined) {
          if (!build_insert2<CAP>(nxt, q, sel, (uint32_t)m2, (uint32_t)(m2 >> 32), (uint32_t)st2, e.org, stage, lane)) status = kSegOverflow;
          continue;
        }
        const uint64_t vb = __ballot(viable);
        if (viable) {
          const uint32_t pos = (qh + qn + (uint32_t)__popcll(vb & ((1ull << lane) - 1ull))) & 127u;
          wq[pos] = Ent{(uint32_t)m2, (uint32_t)(m2 >> 32), (uint32_t)st2, e.org};
          wq_sel[pos] = sel;
        }
        qn += (uint32_t)__popcll(vb);
        lds_sync();
        while (qn >= 64u && status == kSegOk) {
          const uint32_t at = (qh + lane) & 127u;
          const Ent c = wq[at];
          const uint32_t cs = wq_sel[at];
          if (!build_insert2<CAP>(nxt, q, cs, c.mlo, c.mhi, c.st, c.org, stage, lane)) status = kSegOverflow;
          qh = (qh + 64u) & 127u; qn -= 64u;
        }
      }
      if (pipelined && qn && status == kSegOk) {        // what is left in the ring
        const uint32_t at = (qh + lane) & 127u;
        const Ent c = wq[at];
        const uint32_t cs = lane < qn ? wq_sel[at] : 0u;
        if (!build_insert2<CAP>(nxt, q, cs, c.mlo, c.mhi, c.st, c.org, stage, lane)) status = kSegOverflow;
      }
      src = q.e; n_src = q.n; via_list = false;
      { Ent* t = dst_e; dst_e = dst_other; dst_other = t; }
    }
    if (status != kSegOk) break;
    // level F+1 is complete
    configs_total += nxt.n;
    max_level = max(max_level, nxt.n);
    {   // which origins are still alive
      uint32_t any = 0;
      for (uint32_t base = 0; base < nxt.n; base += 64) { const uint32_t i = base + lane; any |= i < nxt.n ? nxt.e[i].org : 0u; }
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) any |= (uint32_t)__shfl_xor((int)any, d);
      if (lane < 32 && ((any >> lane) & 1u)) last_level = F + 1;
    }
    {   // the set just built becomes `cur`; the other two are free
      Build t = cur; cur = nxt; nxt = t;
      Ent* old_cur = cur_e; cur_e = nxt_e; nxt_e = old_cur;     // q_e keeps its place
    }
    // (tab_nxt keeps serving the new `cur` for lookups; the next build_begin on it retires it)
    if (cur.n == 0) break;                           // nobody passes completion F
    // ---- park the prefetched level in LDS
    if (pre) {
      if (p_C > kCand) { status = kSegOverflow; break; }
      if (lane < p_C) { cand[lane] = p_rec; cand_tw[lane] = p_tw; }
      { uint64_t* t = row_a; row_a = row_b; row_b = t; }
      if (lane < 32) row_b[lane] = p_row;
      lds_sync();
      if (p_C > 64u) { uint32_t nl, cc; need_window(F + 1u); if (!load_cands(F + 1u, 64u, nl, cc)) { status = kSegOverflow; break; } }
      crashed_twins(p_nlive, p_C);
    }
  }

  // ---- the relation this wavefront hands on: origin -> ids of the next segment's origin space
  if (status == kSegOk && cur.n != 0 && !dump) {
    uint32_t no1 = 0;
    if (F1 != R) {
      uint32_t nl = 0, cc = 0;
      need_window(F1);
      if (!load_cands(F1, 0u, nl, cc) || nl > 6u) status = kSegOverflow;
      no1 = nl;
    }
    for (uint32_t base = 0; base < cur.n && status == kSegOk; base += 64) {
      const uint32_t i = base + lane;
      const bool val = i < cur.n;
      const Ent e = val ? cur.e[i] : Ent{0, 0, 0, 0};
      uint32_t org = val ? e.org : 0u;
      // last segment: word 0 = the final states reached, as bits (register family: nil = bit 0, value v = bit v + 1;
      // other models: bit 0, the state itself goes out as end_state)
      uint32_t id2 = V > 1u ? rdm_index((int32_t)e.st, 32u) : 0u;
      if (F1 != R) {
        const uint32_t sidx = e.st == (uint32_t)TBC_NIL ? 0u : e.st + 1u;
        if (__ballot(val && sidx >= A.n_dom)) { status = kSegOverflow; break; }     // a state outside the domain: cannot be numbered
        const uint64_t m = mask_of(e);
        id2 = sidx << no1;
        for (uint32_t c = 0; c < no1; c++) if ((m >> ((cand[c].f_slot >> 8) & 63u)) & 1ull) id2 |= 1u << c;
      }
      while (org) { const uint32_t o = (uint32_t)__builtin_ctz(org); atomicOr(&Mrel[o * kSweepSlices + (id2 >> 5)], 1u << (id2 & 31u)); org &= org - 1u; }
    }
  }
  lds_sync();
  if (dump) return;
  if (lane < 32) {
#pragma unroll
    for (uint32_t wd = 0; wd < kSweepSlices; wd++) out->M[lane][wd] = Mrel[lane * kSweepSlices + wd];
    out->last_level[lane] = last_level;
  }
  if (lane == 0) {
    out->status = status; out->F0 = F0; out->F1 = F1; out->n_org = n_org;
    out->max_level = max_level; out->subrounds = subrounds; out->configs_total = configs_total; out->probes = probes;
    out->n_end = cur.n; out->end_state = (status == kSegOk && cur.n) ? cur.e[0].st : 0u;
  }
}

bool launch_sweep(const SweepArgs& a, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  constexpr uint32_t kSmall = kSweepCap, kMid = kSweepCapMid, kBig = kSweepCapBig;
  static bool big_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&jit_sweep_kernel<kBig>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, sweep_lds_words<kBig>() * 4) == hipSuccess;
  static bool mid_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&jit_sweep_kernel<kMid>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, sweep_lds_words<kMid>() * 4) == hipSuccess;
  if (a.dump_cfg) {        // cuts are in place from the sweep proper; the big sets hold whatever the sweep held
    if (big_ok) hipLaunchKernelGGL(jit_sweep_kernel<kBig>, dim3(1), dim3(64), sweep_lds_words<kBig>() * 4, s, a);
    else hipLaunchKernelGGL(jit_sweep_kernel<kSmall>, dim3(1), dim3(64), sweep_lds_words<kSmall>() * 4, s, a);
    return true;
  }
  // a workgroup of eight wavefronts per segment (K6w, jit_sweep_wg.hip) where latency is what counts -- a burst of concurrency is
  // one segment whatever the cuts, and only more lanes on its levels shorten it (round 4, one 10k-op history through tbc_check:
  // 2.48 -> 1.56 ms median, profiles/r04_sweep_wg_first_measurement.log)
  if (a.seg_list) {        // second pass over the segments that overflowed the small sets
    if (a.n_list <= 4096u && launch_sweep_wg(a, s)) return true;
    if (a.reach_hdr) return false;          // (the relaxed sweep exists in the workgroup kernel only)
    if (!big_ok) return false;
    hipLaunchKernelGGL(jit_sweep_kernel<kBig>, dim3(a.n_list), dim3(64), sweep_lds_words<kBig>() * 4, s, a);
    return true;
  }
  hipLaunchKernelGGL(sweep_cuts_kernel, dim3(a.n_hist), dim3(256), 0, s, a);
  const uint32_t waves = a.n_hist * a.max_segs * kSweepSlices;
  // few wavefronts (a history or a handful through tbc_check): a workgroup per segment
  if (waves <= 4096u && launch_sweep_wg(a, s)) return true;
  if (a.reach_hdr) return false;
  // a few histories: latency is what counts and the CUs are not full -- take the larger sets, so that a burst of
  // concurrency does not cost a second pass; many: four wavefronts per CU
  if (mid_ok && waves <= 4096u) hipLaunchKernelGGL(jit_sweep_kernel<kMid>, dim3(waves), dim3(64), sweep_lds_words<kMid>() * 4, s, a);
  else hipLaunchKernelGGL(jit_sweep_kernel<kSmall>, dim3(waves), dim3(64), sweep_lds_words<kSmall>() * 4, s, a);
  return true;
}

}  // namespace tbc
