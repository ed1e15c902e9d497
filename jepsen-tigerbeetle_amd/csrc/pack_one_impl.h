// pack_one_impl.h -- K1 for ONE history (or a handful): the body of pack_one_kernel (pack_one.hip), gfx950.
//
// pack_kernel (pack.hip) gives a history to one workgroup and is written for a batch, where 32,768 workgroups fill the GPU and a
// workgroup's own critical path does not matter.  For a single history through tbc_check it IS the critical path -- 0.28 ms of a
// 1.5 ms call (profiles/r04_kernel_stats_tbc_check_k6w.csv) -- and most of that is one chain: the stable place of every op in its
// process's list is found by ONE wavefront walking the ops 64 at a time (157 dependent steps for a 10k-op history), next to five
// passes that fetch the position bitmap and its prefix from global memory through the L2.  This body gives the same history to the
// same 1,024 threads and leaves the same bytes behind (records, list starts, completion tables, the ranks in the scratch arena,
// n_ret, status: tests/test_pack_one_emu.py compares every word with a host restatement of pack.hip's definitions), but
//
//   * the position bitmap and its popcount prefix live in LDS (a history of up to 131,072 events: 2 x 16 KB);
//   * the stable counting sort is two-level: the ops are dealt to the sixteen wavefronts in contiguous blocks, the validation pass
//     counts (process, block) pairs with LDS adds on its way, one scan turns the counts into each block's first place per process,
//     and then every wavefront walks ITS block 64 ops at a time -- ten dependent steps instead of 157;
//   * ranks, destination and record of an op are worked out in that one walk (ranks from LDS; no second and third pass over the
//     columns, nothing re-read from the scratch arena), with the next 64 ops' columns requested before the current ones are used;
//   * the two serial prefix loops (1,024 partial sums, the process slots) are workgroup scans.
//
// Register, cas-register, mutex and table models (what one history through tbc_check is, and what the level sweep takes); set, bank
// and multi-register keep pack_kernel, and so does a history too long for the LDS tables (tbc_api.hip decides, pack_one_fits()).
//
// THE BATCH FORM (geometry BatchGeo; pack_one.hip's pack_wg_kernel).  In a batch the same walk is what a pass pays for
// three times over: pack_kernel (one wavefront of four walking, the bitmap and its prefix in global memory), then open_counts_kernel
// (pack_open.hip), which reads the ranks back from the scratch arena, builds the histogram of invocation ranks with global atomics
// and scans it in global memory three times -- 28 ms of a 60 ms pack per 32,768 histories, at 6 % vector use and 50-70 % of the wave
// cycles waiting (profiles/r03_pmc_final.txt).  With COUNTS the body also does open_counts_kernel's work while the ranks are still in
// registers: completion slots and read kinds as bytes (slot8, rk8) scattered from the walk, the histogram of the listed calls'
// invocation ranks in LDS (16 bits a rank), its scans in LDS, off[] written once; the crashed-call arrays (mask form only) as
// open_counts_kernel builds them.  Four wavefronts and 31 KB of LDS per history: five workgroups per CU.
//
// Batch64Geo is the same body for histories of at most 64 process slots (what the narrow search kernel takes): the (block, process)
// counts shrink with the slots and a histogram entry to a byte -- 19 KB, eight workgroups per CU instead of five; the kernel waits on
// memory for half of its wave cycles (profiles/r05_pmc_final.txt), so residency is what it is short of.
//
// Written against wave_env_wg.h like jit_sweep_wg_impl.h: the same file compiles for the workgroup emulator (tests/emu).
// Both forms are the library's defaults since round 5 (tbc_api.hip: one history -> pack_one_counts_kernel, a batch -> pack_wg_kernel;
// pack_kernel + open_counts_kernel keep what these bodies do not take).
#pragma once
#include "tbc_internal.h"
#include "wave_env_wg.h"

namespace tbc {
namespace packone {

// A geometry: wavefronts per workgroup, the history rows the LDS bitmap holds, process slots, and (COUNTS) completions
// HBITS_: bits of a rank's entry in the histogram of the listed calls' invocation ranks (at most one call per process slot is invoked
// between two completions: an entry holds MAXW_)
template <uint32_t NW_, uint32_t MAXEV_, uint32_t MAXW_, uint32_t MAXR_, uint32_t HBITS_ = 16>
struct Geo {
  static constexpr uint32_t kNW = NW_, kT = 64 * NW_, kMaxEvents = MAXEV_, kMaxW = MAXW_, kMaxR = MAXR_, kHBits = HBITS_, kHPer = 32 / HBITS_;
  static_assert((HBITS_ == 8 || HBITS_ == 16) && MAXW_ < (1u << HBITS_), "a histogram entry counts up to one call per slot");
  static constexpr bool kCounts = MAXR_ != 0;
  static constexpr uint32_t kBmWords = MAXEV_ / 32 + 32;      // (+ the word E / 32 itself and padding)
  // LDS words: bitmap | prefix | (block, process) counts | list starts | per-wavefront scan totals | flags | (COUNTS) histogram of
  // invocation ranks, two (or four) ranks a word | (COUNTS) the ranks at which a read completes, one bit each
  static constexpr uint32_t kOffBm = 0, kOffPre = kBmWords, kOffCnt = 2 * kBmWords, kOffSeg = kOffCnt + NW_ * MAXW_,
                            kOffTot = kOffSeg + MAXW_ + 8, kOffFlag = kOffTot + 2 * NW_, kOffHist = kOffFlag + 8,
                            kHistWords = kCounts ? MAXR_ / kHPer + 4 : 0, kOffRdb = kOffHist + kHistWords,
                            kRdbWords = kCounts ? MAXR_ / 32 + 2 : 0, kLdsWords = kOffRdb + kRdbWords;
  WV_HD static constexpr uint32_t lds_words() { return kLdsWords; }
  // what the body handles (tbc_api.hip asks before it launches it; everything else goes to pack_kernel)
  WV_HD static bool fits(uint32_t model_kind, uint32_t n_ops, uint32_t n_events, uint32_t n_slots) {
    const bool model_ok = model_kind == TBC_MODEL_REGISTER || model_kind == TBC_MODEL_CAS_REGISTER || model_kind == TBC_MODEL_MUTEX ||
                          model_kind == TBC_MODEL_TABLE;
    return model_ok && n_events <= kMaxEvents && n_slots <= kMaxW && (!kCounts || n_ops <= kMaxR);      // (completions <= ops)
  }
};
using OneGeo = Geo<16, 131072, kMaxSlots, 0>;        // one history or a handful through tbc_check: sixteen wavefronts, 103 KB
using OneCountsGeo = Geo<16, 131072, kMaxSlots, 16384>;   // ... and open_counts_kernel's work in the same pass (one launch fewer per call): 138 KB
using BatchGeo = Geo<4, 32768, 256, 8192>;           // a batch: four wavefronts, pack + open counts, 31 KB: five workgroups per CU
using Batch64Geo = Geo<4, 32768, 64, 8192, 8>;       // ... of histories with at most 64 process slots (one mask word: the narrow kernel's batches): 19 KB, eight per CU

constexpr uint32_t kNW = OneGeo::kNW;                // (the one-history form's, for its launcher and the emulator harness)
WV_HD constexpr uint32_t lds_words() { return OneGeo::lds_words(); }
WV_HD inline bool fits(uint32_t model_kind, uint32_t n_ops, uint32_t n_events, uint32_t n_slots) { return OneGeo::fits(model_kind, n_ops, n_events, n_slots); }

WV_DEV bool op_ok(uint32_t kind, uint32_t f, int32_t a, uint32_t n_classes) {
  return kind == TBC_MODEL_REGISTER ? (f == TBC_F_READ || f == TBC_F_WRITE)
       : kind == TBC_MODEL_CAS_REGISTER ? (f == TBC_F_READ || f == TBC_F_WRITE || f == TBC_F_CAS)
       : kind == TBC_MODEL_MUTEX ? (f == TBC_F_ACQUIRE || f == TBC_F_RELEASE)
       : (kind == TBC_MODEL_TABLE && f == TBC_F_CLASS && (uint32_t)a < n_classes);
}

// Exclusive prefix sum of x over the workgroup's 1,024 threads (thread order); *total = the sum.  Every thread calls it.
// Inside a wavefront: rows of 16 by row shifts, the four row sums by lane reads; across wavefronts: sixteen totals in LDS.
// (inside history<G>) `tot` = 2 * G::kNW words of LDS, used alternately by consecutive calls (`flip`), so one workgroup barrier per call is enough.
#define TBC_PACK_ONE_SCAN(out_, total_, x_, flip_)                                                                        \
  do {                                                                                                                   \
    uint32_t y_ = (x_);                                                                                                  \
    y_ += wv::row_shr0<1>(y_); y_ += wv::row_shr0<2>(y_); y_ += wv::row_shr0<4>(y_); y_ += wv::row_shr0<8>(y_);          \
    const uint32_t t0_ = wv::readlane(y_, 15u), t1_ = wv::readlane(y_, 31u), t2_ = wv::readlane(y_, 47u), t3_ = wv::readlane(y_, 63u); \
    const uint32_t row_ = lane >> 4;                                                                                     \
    y_ += (row_ > 0u ? t0_ : 0u) + (row_ > 1u ? t1_ : 0u) + (row_ > 2u ? t2_ : 0u);                                      \
    uint32_t* tt_ = tot + ((flip_) ? G::kNW : 0u);                                                                          \
    if (lane == 0u) tt_[wave] = t0_ + t1_ + t2_ + t3_;                                                                   \
    wv::wg_barrier();                                                                                                    \
    uint32_t base_ = 0u, all_ = 0u;                                                                                      \
    for (uint32_t w_ = 0; w_ < G::kNW; w_++) { const uint32_t v_ = wv::lds_ld32(&tt_[w_]); base_ += w_ < wave ? v_ : 0u; all_ += v_; } \
    (out_) = base_ + y_ - (x_);                                                                                          \
    (total_) = all_;                                                                                                     \
  } while (0)

// One history per workgroup: history A.h0 + wg_index().  O: open_counts_kernel's arguments (G::kCounts only, else unused).
template <class G>
WV_DEV void history(const PackArgs& A, const PackOpenArgs& O, uint32_t* lds) {
  constexpr uint32_t kT = G::kT;
  static_assert(G::kMaxW <= G::kT, "one thread per process slot in the list-start scan");
  const uint32_t tid = wv::wg_thread(), lane = tid & 63u, wave = tid >> 6;
  const uint32_t h = A.h0 + wv::wg_index();
  if (h >= A.n_hist) return;
  Hist* H = &A.hist[h];
  const uint32_t n = H->n_ops, W = H->n_slots, E = H->n_events;
  const bool cf = (H->flags & kHistCount) != 0u;             // count form: a crashed call holds no slot (pack.hip)
  const uint8_t* f = A.f + H->op_off;
  const int32_t* a = A.a + H->op_off;
  const int32_t* b = A.b + H->op_off;
  const int32_t* proc = A.process + H->op_off;
  const uint32_t* inv = A.inv_pos + H->op_off;
  const uint32_t* ret = A.ret_pos + H->op_off;
  uint32_t* sc_inv = A.scratch + H->frame_off;
  uint32_t* sc_ret = sc_inv + n;
  uint32_t* sc_dst = sc_ret + n;
  Rec* rec = A.rec + H->rec_off;
  uint32_t* bm = lds + G::kOffBm;
  uint32_t* pre = lds + G::kOffPre;
  uint32_t* cnt = lds + G::kOffCnt;        // [block][process]: ops of the process in the block; later: placed before the block / so far
  uint32_t* seg = lds + G::kOffSeg;        // first record of each process's list (W + 1 entries)
  uint32_t* tot = lds + G::kOffTot;
  uint32_t* flag = lds + G::kOffFlag;      // 0 = error bits, 1 = completions seen, 2 = (COUNTS) crashed calls that are candidates
  uint32_t* hist = lds + G::kOffHist;      // COUNTS: listed calls invoked at each rank, G::kHBits bits a rank (at most W of them)
  uint32_t* rdb = lds + G::kOffRdb;        // COUNTS, branch lists: a read completes at this rank
  const uint32_t nw = E / 32u + 1u;
  const uint32_t chunks = (n + 63u) / 64u;
  const uint32_t cpw = (chunks + G::kNW - 1u) / G::kNW ? (chunks + G::kNW - 1u) / G::kNW : 1u;       // 64-op chunks per wavefront's block
  // COUNTS: what open_counts_kernel leaves behind (pack_open.hip's header)
  BeamHist* const Bh = G::kCounts ? &O.bh[h] : nullptr;
  const bool branch = G::kCounts && O.branch_lists != 0u;
  uint8_t* const slot8 = G::kCounts ? O.slot8 + slot8_off(H->op_off, h) : nullptr;
  uint8_t* const rk8 = (G::kCounts && O.rk8) ? O.rk8 + slot8_off(H->op_off, h) : nullptr;
  uint32_t* const ncr = G::kCounts ? O.ncr + Bh->off_off : nullptr;
  // a history pack refuses (or one without a completion) leaves no lists: what open_counts_kernel says of it
  const auto no_lists = [&]() { if (G::kCounts && tid == 0u) { Bh->status = 0u; Bh->n_crashed = 0u; Bh->lst_need = 0u; } };

  // TBC_PACK_PROF (a build of its own, scripts/build_variant.sh): where a workgroup's time goes -- its first thread adds the 100 MHz clock's
  // ticks per phase to the debug words 32.. (TBC_DEBUG=1), tbc_debug_peek reads them
#if defined(TBC_PACK_PROF) && !defined(TBC_EMU)
  uint64_t prof_t = wall_clock64();
  uint32_t prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // (the ticks are kept in registers and added to the profile words once, when the workgroup ends)
#define TBC_PROF_PHASE(ph_) do { if (tid == 0u) { const uint64_t t_ = wall_clock64(); prof_acc[ph_] += (uint32_t)(t_ - prof_t); prof_t = t_; \
    if ((ph_) == 7 && A.dbg) { for (int q_ = 1; q_ < 8; q_++) atomicAdd(&A.dbg[32 + q_], prof_acc[q_]); } } } while (0)
#else
#define TBC_PROF_PHASE(ph_) do {} while (0)
#endif
  // ---- phase 0: LDS tables to zero
  for (uint32_t w = tid; w < nw; w += kT) bm[w] = 0u;
  for (uint32_t x = tid; x < G::kNW * W; x += kT) cnt[x] = 0u;
  if (tid < 3u) flag[tid] = 0u;
  if constexpr (G::kCounts) {
    for (uint32_t x = tid; x < G::kHistWords; x += kT) hist[x] = 0u;
    for (uint32_t x = tid; x < G::kRdbWords; x += kT) rdb[x] = 0u;
  }
  wv::wg_barrier();

  // ---- phase 1: validate the rows, set the completion bits, count (block, process) pairs.  Four rows' columns are requested
  // before the first is looked at.  What pack_kernel refuses is refused here, with the same status.
  {
    uint32_t err = 0u, done = 0u;
    for (uint32_t i0 = tid; i0 < n; i0 += 4u * kT) {
      uint32_t ivs[4], rts[4], pvs[4]; int32_t ps[4], as[4]; uint32_t fs[4];
      WV_UNROLL
      for (uint32_t k = 0; k < 4; k++) {
        const uint32_t i = i0 + k * kT;
        const bool in = i < n;
        ivs[k] = in ? inv[i] : 0u; rts[k] = in ? ret[i] : 0u; ps[k] = in ? proc[i] : 0;
        pvs[k] = (in && i > 0u) ? inv[i - 1u] : 0u;
        fs[k] = in ? (uint32_t)f[i] : 0u; as[k] = in ? a[i] : 0;
      }
      WV_UNROLL
      for (uint32_t k = 0; k < 4; k++) {
        const uint32_t i = i0 + k * kT;
        if (i >= n) continue;
        const uint32_t iv = ivs[k], rt = rts[k];
        const int32_t p = ps[k];
        const bool slotless = cf && rt == TBC_POS_CRASHED;
        bool bad = iv >= E || (!slotless && (p < 0 || (uint32_t)p >= W)) || (i > 0u && pvs[k] >= iv);
        if (rt != TBC_POS_CRASHED) bad = bad || rt <= iv || rt >= E;
        if (bad) { err |= (uint32_t)TBC_ERR_BAD_HISTORY; continue; }
        if (!op_ok(A.model_kind, fs[k], as[k], A.n_classes)) { err |= 0x100u | (uint32_t)TBC_ERR_MODEL; continue; }
        if (rt != TBC_POS_CRASHED) { wv::lds_or32(&bm[rt >> 5], 1u << (rt & 31u)); done++; }
        if (!slotless) wv::lds_add32_wg(&cnt[((i >> 6) / cpw) * W + (uint32_t)p], 1u);
      }
    }
    if (err) wv::lds_or32(&flag[0], err);
    if (done) wv::lds_add32_wg(&flag[1], done);
  }
  wv::wg_barrier();
  TBC_PROF_PHASE(1);
  const uint32_t err1 = wv::lds_ld32(&flag[0]), n_done = wv::lds_ld32(&flag[1]);
  if (err1) {
    if (tid == 0u) { H->n_ret = 0u; H->status = (err1 & 0x100u) ? (uint32_t)TBC_ERR_MODEL : (uint32_t)TBC_ERR_BAD_HISTORY; }
    no_lists();
    return;
  }

  // ---- phase 2: exclusive popcount prefix per bitmap word (thread t: words [t * per, (t + 1) * per))
  uint32_t R;
  {
    const uint32_t per = (nw + kT - 1u) / kT;
    const uint32_t lo = tid * per < nw ? tid * per : nw, hi = lo + per < nw ? lo + per : nw;
    uint32_t sum = 0u;
    for (uint32_t w = lo; w < hi; w++) sum += (uint32_t)__builtin_popcount(bm[w]);
    uint32_t run;
    TBC_PACK_ONE_SCAN(run, R, sum, 0);
    for (uint32_t w = lo; w < hi; w++) { pre[w] = run; run += (uint32_t)__builtin_popcount(bm[w]); }
  }
  TBC_PROF_PHASE(2);
  if (R != n_done) {      // two completions on one history row
    if (tid == 0u) { H->n_ret = 0u; H->status = (uint32_t)TBC_ERR_BAD_HISTORY; }
    no_lists();
    return;
  }

  // ---- phase 3: list starts (a head and a tail sentinel per process) and, per block, the ops of each process placed before it
  {
    uint32_t mine = 0u;
    if (tid < W) {
      uint32_t run = 0u;
      for (uint32_t k = 0; k < G::kNW; k++) { const uint32_t c = cnt[k * W + tid]; cnt[k * W + tid] = run; run += c; }
      mine = run + 2u;
    }
    uint32_t start, total;
    TBC_PACK_ONE_SCAN(start, total, mine, 1);
    if (tid < W) { seg[tid] = start; A.seg[H->seg_off + tid] = start; }
    if (tid == 0u) { seg[W] = total; A.seg[H->seg_off + W] = total; }
    // (W can be as many as the threads: the scan's total is the last entry either way)
  }
  wv::wg_barrier();
  TBC_PROF_PHASE(3);

  // ---- phase 4: every wavefront walks its block, 64 ops at a time.  An op's place in its process's list = the list's first
  // record + 1 + the ops of the process placed by earlier blocks and earlier chunks of this block (cnt) + the lower lanes of this
  // chunk holding an op of the same process (64 lane reads, counted in registers).  Ranks come from the LDS bitmap; the record
  // goes out at once.  The next chunk's columns are requested before this one's are used.
  // COUNTS: with the ranks in hand, what open_counts_kernel would fetch them again for -- the completion's slot and read kind as
  // bytes, the call's invocation rank into the histogram of the listed calls (branch lists: a live read is not listed, its
  // completion is marked instead), a crashed candidate into ncr[] (mask form; global, zeroed by the host -- crash-heavy only)
  {
    const uint32_t c0 = wave * cpw, c1 = c0 + cpw < chunks ? c0 + cpw : chunks;
    uint32_t* mycnt = cnt + wave * W;
    uint32_t i = c0 * 64u + lane;
    bool in = c0 < c1 && i < n;
    uint32_t n_iv = in ? inv[i] : 0u, n_rt = in ? ret[i] : TBC_POS_CRASHED, n_f = in ? (uint32_t)f[i] : 0u;
    int32_t n_p = in ? proc[i] : 0, n_a = in ? a[i] : 0, n_b = in ? b[i] : 0;
    uint32_t crashed_cands = 0u;
    for (uint32_t c = c0; c < c1; c++) {
      const uint32_t iv = n_iv, rt = n_rt, ff = n_f; const int32_t pp = n_p, aa = n_a, bb = n_b;
      const bool here = in;
      const uint32_t me = i;
      i += 64u;
      in = c + 1u < c1 && i < n;
      n_iv = in ? inv[i] : 0u; n_rt = in ? ret[i] : TBC_POS_CRASHED; n_f = in ? (uint32_t)f[i] : 0u;
      n_p = in ? proc[i] : 0; n_a = in ? a[i] : 0; n_b = in ? b[i] : 0;
      const bool live = here && rt != TBC_POS_CRASHED;
      const uint32_t p = (here && !(cf && rt == TBC_POS_CRASHED)) ? (uint32_t)pp : 0xFFFFFFFFu;     // 0xFFFFFFFF: equal to no process
      uint32_t before = 0u;
      for (uint32_t l = 0; l < 64u; l++) {
        const uint32_t pl = wv::readlane(p, l);
        before += (pl == p && l < lane) ? 1u : 0u;
      }
      const bool slotted = p != 0xFFFFFFFFu;
      uint32_t dst = kInf;
      if (slotted) dst = seg[p] + 1u + mycnt[p] + before;
      wv::barrier();
      if (slotted) wv::lds_add32(&mycnt[p], 1u);
      wv::barrier();
      if (here) {
        const uint32_t ir = pre[iv >> 5] + (uint32_t)__builtin_popcount(bm[iv >> 5] & ((1u << (iv & 31u)) - 1u));
        uint32_t rr = kInf;
        if (live) {
          rr = pre[rt >> 5] + (uint32_t)__builtin_popcount(bm[rt >> 5] & ((1u << (rt & 31u)) - 1u));
          A.ret_slot[H->ret_off + rr] = (uint32_t)pp;
          A.ret_op[H->ret_off + rr] = me;
        }
        sc_inv[me] = ir; sc_ret[me] = rr; sc_dst[me] = dst;
        if (slotted) {
          Rec r;
          r.inv_rank = ir; r.ret_rank = rr; r.opidx = me; r.f = ff; r.a = aa; r.b = bb;
          r.cls = rec_cls(ff, aa, rr == kInf); r.prod = look_prod(ff, aa, bb);
          rec[dst] = r;
        }
        if constexpr (G::kCounts) {
          const bool isread = ff == TBC_F_READ;
          if (live) {
            slot8[rr] = (uint8_t)pp;
            if (rk8) rk8[rr] = isread ? (uint8_t)rdm_index(aa, O.vpad) : (uint8_t)0xFF;
            if (branch && isread) wv::lds_or32(&rdb[rr >> 5], 1u << (rr & 31u));
            else wv::lds_add32_wg(&hist[ir / G::kHPer], 1u << (G::kHBits * (ir % G::kHPer)));
          } else if (!cf && !(isread && aa == TBC_NIL)) {
            crashed_cands++;
            if (ir < R) atomicAdd(&ncr[ir], 1u);
          }
        }
      }
    }
    if (G::kCounts && crashed_cands) wv::lds_add32_wg(&flag[2], crashed_cands);
  }
  TBC_PROF_PHASE(4);
  // the sentinels
  for (uint32_t p = tid; p < W; p += kT) {
    Rec hd; hd.inv_rank = 0; hd.ret_rank = 0; hd.opidx = kInf; hd.f = kFNone; hd.a = 0; hd.b = 0; hd.cls = 0; hd.prod = kLookNone;
    Rec tl = hd; tl.inv_rank = kInf; tl.ret_rank = kInf;
    rec[seg[p]] = hd;
    rec[seg[p + 1u] - 1u] = tl;
  }
  wv::wg_fence();          // (the records are read back below by other wavefronts of THIS workgroup, through L2: no wider fence -- wave_env.h)
  wv::wg_barrier();
  TBC_PROF_PHASE(5);

  // ---- phase 5: one open op per process: the previous op of the same process must have completed before this one was invoked
  // (the previous record of the list; another wavefront may have written it: agent-scope loads, as pack_kernel's)
  {
    const uint32_t c0 = wave * cpw, c1 = c0 + cpw < chunks ? c0 + cpw : chunks;
    uint32_t err = 0u;
    for (uint32_t c = c0; c < c1; c += 4u) {                  // four chunks' places first, then their eight words of records: two round trips for four chunks
      uint32_t d[4], iv[4], pr[4], pf[4];
      WV_UNROLL
      for (uint32_t k = 0; k < 4u; k++) {
        const uint32_t i = (c + k) * 64u + lane;
        const bool in = c + k < c1 && i < n;
        d[k] = in ? sc_dst[i] : kInf;                         // (this thread's own store; kInf = slotless: no record, no predecessor)
        iv[k] = in ? sc_inv[i] : 0u;
      }
      WV_UNROLL
      for (uint32_t k = 0; k < 4u; k++) {
        const wv::gu32* prev = (const wv::gu32*)(const void*)&rec[d[k] != kInf ? d[k] - 1u : 0u];
        pr[k] = d[k] != kInf ? wv::ld32(prev + 1) : 0u;
        pf[k] = d[k] != kInf ? wv::ld32(prev + 3) : kFNone;
      }
      WV_UNROLL
      for (uint32_t k = 0; k < 4u; k++)
        if (d[k] != kInf && pf[k] != kFNone && !(pr[k] < iv[k])) err = (uint32_t)TBC_ERR_BAD_HISTORY;
    }
    if (err) wv::lds_or32(&flag[0], err);
  }
  wv::wg_barrier();
  TBC_PROF_PHASE(6);
  const uint32_t err5 = wv::lds_ld32(&flag[0]);
  if (tid == 0u) { H->n_ret = R; H->status = err5 ? (uint32_t)TBC_ERR_BAD_HISTORY : 0u; }

  // ---- phase 6 (COUNTS): open_counts_kernel's tables.  open(F) = the listed calls invoked at rank <= F minus those that have left
  // the lists before front F (the F completions; branch lists: minus the reads among them, which were never listed); off[] = the
  // exclusive scan of open().  Thread t scans ranks [t * per, (t + 1) * per) of the LDS histogram; three workgroup scans join them.
  if constexpr (G::kCounts) {
    if (err5 || R == 0u) { no_lists(); return; }
    uint32_t* off = O.off + Bh->off_off;
    if (tid < 16u) { slot8[R + tid] = (uint8_t)0; if (rk8) rk8[R + tid] = (uint8_t)0xFF; }
    const uint32_t per = (R + kT - 1u) / kT;
    const uint32_t lo = tid * per < R ? tid * per : R, hi = lo + per < R ? lo + per : R;
    const auto h16 = [&](uint32_t i) -> uint32_t { return (hist[i / G::kHPer] >> (G::kHBits * (i % G::kHPer))) & ((1u << G::kHBits) - 1u); };
    const auto rd_at = [&](uint32_t i) -> uint32_t { return branch ? (rdb[i >> 5] >> (i & 31u)) & 1u : 0u; };
    uint32_t nrd = 0u, sum = 0u;
    for (uint32_t i = lo; i < hi; i++) { nrd += rd_at(i); sum += h16(i); }
    uint32_t base_rd, base_inv, unused;
    TBC_PACK_ONE_SCAN(base_rd, unused, nrd, 0);
    TBC_PACK_ONE_SCAN(base_inv, unused, sum, 1);
    uint32_t run = base_inv, rd = base_rd, open_sum = 0u;
    for (uint32_t i = lo; i < hi; i++) { run += h16(i); open_sum += run - (i - rd); rd += rd_at(i); }
    uint32_t pos, total;
    TBC_PACK_ONE_SCAN(pos, total, open_sum, 0);
    run = base_inv; rd = base_rd;
    for (uint32_t i = lo; i < hi; i++) { run += h16(i); off[i] = pos; pos += run - (i - rd); rd += rd_at(i); }
    if (tid == 0u) { off[R] = total; Bh->lst_need = total; }
    const uint32_t n_cr = wv::lds_ld32(&flag[2]);          // (every wavefront added its count before the barrier behind the walk)
    if (total > Bh->lst_cap) {           // the lists do not fit their arena: the caller takes the sequential kernel (or sizes the arena from lst_need)
      if (tid == 0u) { Bh->status = 1u; Bh->n_crashed = 0u; }
      return;
    }
    if (cf || n_cr == 0u) {              // (count form: the classes of crashed calls are the host's, count_fronts_kernel counts them per front)
      if (tid == 0u) { Bh->status = 0u; Bh->n_crashed = 0u; }
    } else {
      // mask form with crashed calls: ncr[F] = crashed candidates invoked before completion F (inclusive prefix of what the walk
      // counted into ncr[] -- every wavefront's atomics are behind two workgroup barriers), and the candidates in invocation order
      {
        uint32_t s2 = 0u;
        for (uint32_t i = lo; i < hi; i++) s2 += wv::ld32((const wv::gu32*)(const void*)&ncr[i]);
        uint32_t r2;
        TBC_PACK_ONE_SCAN(r2, unused, s2, 1);
        for (uint32_t i = lo; i < hi; i++) { r2 += wv::ld32((const wv::gu32*)(const void*)&ncr[i]); ncr[i] = r2; }
      }
      OpRec* crashed = O.crashed + H->op_off;
      const uint32_t pern = (n + kT - 1u) / kT;
      const uint32_t l2 = tid * pern < n ? tid * pern : n, h2 = l2 + pern < n ? l2 + pern : n;
      uint32_t mine = 0u;
      for (uint32_t i = l2; i < h2; i++) mine += (ret[i] == TBC_POS_CRASHED && !(f[i] == TBC_F_READ && a[i] == TBC_NIL)) ? 1u : 0u;
      uint32_t at, all;
      TBC_PACK_ONE_SCAN(at, all, mine, 0);
      for (uint32_t i = l2; i < h2; i++) if (ret[i] == TBC_POS_CRASHED && !(f[i] == TBC_F_READ && a[i] == TBC_NIL)) {
        OpRec o; o.op = i; o.f_slot = (uint32_t)f[i] | ((uint32_t)proc[i] << 8); o.a = a[i]; o.b = b[i];
        crashed[at++] = o;
      }
      if (tid == 0u) { Bh->status = 0u; Bh->n_crashed = all; }
    }
    TBC_PROF_PHASE(7);
    if (O.look) {                 // lookahead records past the last rank: nothing is needed there
      const uint32_t MW = O.mask_words, LW = 1u + MW;
      uint64_t* look = O.look + look_off(H->op_off, h, MW);
      for (uint32_t t = R + tid; t < R + kLookPad; t += kT) {
        look[(uint64_t)t * LW] = (uint64_t)(kLookNone << 16 | kLookNone << 24) | (255ull << 32) | (255ull << 40);
        for (uint32_t w = 0; w < MW; w++) look[(uint64_t)t * LW + 1u + w] = 0ull;
      }
    }
  }
}

}  // namespace packone
}  // namespace tbc
