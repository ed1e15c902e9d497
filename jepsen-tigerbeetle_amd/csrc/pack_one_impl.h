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
// Written against wave_env_wg.h like jit_sweep_wg_impl.h: the same file compiles for the workgroup emulator (tests/emu).
// STANDING: verified under the emulator only (TBC_PACK_ONE=1 selects it; the measured default is pack_kernel).
#pragma once
#include "tbc_internal.h"
#include "wave_env_wg.h"

namespace tbc {
namespace packone {

constexpr uint32_t kNW = 16;                         // wavefronts per workgroup
constexpr uint32_t kT = 64 * kNW;
constexpr uint32_t kMaxEvents = 131072;              // history rows the LDS bitmap holds
constexpr uint32_t kBmWords = kMaxEvents / 32 + 32;  // (+ the word E / 32 itself and padding)
// LDS words: bitmap | prefix | (block, process) counts | list starts | per-wavefront scan totals | flags
constexpr uint32_t kOffBm = 0, kOffPre = kBmWords, kOffCnt = 2 * kBmWords, kOffSeg = kOffCnt + kNW * kMaxSlots,
                   kOffTot = kOffSeg + kMaxSlots + 8, kOffFlag = kOffTot + 2 * kNW, kLdsWords = kOffFlag + 8;
WV_HD constexpr uint32_t lds_words() { return kLdsWords; }

// what the body handles (tbc_api.hip asks before it launches it; everything else goes to pack_kernel)
WV_HD inline bool fits(uint32_t model_kind, uint32_t n_ops, uint32_t n_events, uint32_t n_slots) {
  const bool model_ok = model_kind == TBC_MODEL_REGISTER || model_kind == TBC_MODEL_CAS_REGISTER || model_kind == TBC_MODEL_MUTEX ||
                        model_kind == TBC_MODEL_TABLE;
  (void)n_ops;
  return model_ok && n_events <= kMaxEvents && n_slots <= kMaxSlots;
}

WV_DEV bool op_ok(uint32_t kind, uint32_t f, int32_t a, uint32_t n_classes) {
  return kind == TBC_MODEL_REGISTER ? (f == TBC_F_READ || f == TBC_F_WRITE)
       : kind == TBC_MODEL_CAS_REGISTER ? (f == TBC_F_READ || f == TBC_F_WRITE || f == TBC_F_CAS)
       : kind == TBC_MODEL_MUTEX ? (f == TBC_F_ACQUIRE || f == TBC_F_RELEASE)
       : (kind == TBC_MODEL_TABLE && f == TBC_F_CLASS && (uint32_t)a < n_classes);
}

// Exclusive prefix sum of x over the workgroup's 1,024 threads (thread order); *total = the sum.  Every thread calls it.
// Inside a wavefront: rows of 16 by row shifts, the four row sums by lane reads; across wavefronts: sixteen totals in LDS.
// `tot` = 2 * kNW words of LDS, used alternately by consecutive calls (`flip`), so one workgroup barrier per call is enough.
#define TBC_PACK_ONE_SCAN(out_, total_, x_, flip_)                                                                        \
  do {                                                                                                                   \
    uint32_t y_ = (x_);                                                                                                  \
    y_ += wv::row_shr0<1>(y_); y_ += wv::row_shr0<2>(y_); y_ += wv::row_shr0<4>(y_); y_ += wv::row_shr0<8>(y_);          \
    const uint32_t t0_ = wv::readlane(y_, 15u), t1_ = wv::readlane(y_, 31u), t2_ = wv::readlane(y_, 47u), t3_ = wv::readlane(y_, 63u); \
    const uint32_t row_ = lane >> 4;                                                                                     \
    y_ += (row_ > 0u ? t0_ : 0u) + (row_ > 1u ? t1_ : 0u) + (row_ > 2u ? t2_ : 0u);                                      \
    uint32_t* tt_ = tot + ((flip_) ? kNW : 0u);                                                                          \
    if (lane == 0u) tt_[wave] = t0_ + t1_ + t2_ + t3_;                                                                   \
    wv::wg_barrier();                                                                                                    \
    uint32_t base_ = 0u, all_ = 0u;                                                                                      \
    for (uint32_t w_ = 0; w_ < kNW; w_++) { const uint32_t v_ = wv::lds_ld32(&tt_[w_]); base_ += w_ < wave ? v_ : 0u; all_ += v_; } \
    (out_) = base_ + y_ - (x_);                                                                                          \
    (total_) = all_;                                                                                                     \
  } while (0)

// One history per workgroup: history A.h0 + wg_index().
WV_DEV void history(const PackArgs& A, uint32_t* lds) {
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
  uint32_t* bm = lds + kOffBm;
  uint32_t* pre = lds + kOffPre;
  uint32_t* cnt = lds + kOffCnt;           // [block][process]: ops of the process in the block; later: placed before the block / so far
  uint32_t* seg = lds + kOffSeg;           // first record of each process's list (W + 1 entries)
  uint32_t* tot = lds + kOffTot;
  uint32_t* flag = lds + kOffFlag;         // 0 = error bits, 1 = completions seen
  const uint32_t nw = E / 32u + 1u;
  const uint32_t chunks = (n + 63u) / 64u;
  const uint32_t cpw = (chunks + kNW - 1u) / kNW ? (chunks + kNW - 1u) / kNW : 1u;       // 64-op chunks per wavefront's block

  // ---- phase 0: LDS tables to zero
  for (uint32_t w = tid; w < nw; w += kT) bm[w] = 0u;
  for (uint32_t x = tid; x < kNW * W; x += kT) cnt[x] = 0u;
  if (tid < 2u) flag[tid] = 0u;
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
  const uint32_t err1 = wv::lds_ld32(&flag[0]), n_done = wv::lds_ld32(&flag[1]);
  if (err1) {
    if (tid == 0u) { H->n_ret = 0u; H->status = (err1 & 0x100u) ? (uint32_t)TBC_ERR_MODEL : (uint32_t)TBC_ERR_BAD_HISTORY; }
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
  if (R != n_done) {      // two completions on one history row
    if (tid == 0u) { H->n_ret = 0u; H->status = (uint32_t)TBC_ERR_BAD_HISTORY; }
    return;
  }

  // ---- phase 3: list starts (a head and a tail sentinel per process) and, per block, the ops of each process placed before it
  {
    uint32_t mine = 0u;
    if (tid < W) {
      uint32_t run = 0u;
      for (uint32_t k = 0; k < kNW; k++) { const uint32_t c = cnt[k * W + tid]; cnt[k * W + tid] = run; run += c; }
      mine = run + 2u;
    }
    uint32_t start, total;
    TBC_PACK_ONE_SCAN(start, total, mine, 1);
    if (tid < W) { seg[tid] = start; A.seg[H->seg_off + tid] = start; }
    if (tid == 0u) { seg[W] = total; A.seg[H->seg_off + W] = total; }
    // (W can be 1,024 = every thread: the scan's total is the last entry either way)
  }
  wv::wg_barrier();

  // ---- phase 4: every wavefront walks its block, 64 ops at a time.  An op's place in its process's list = the list's first
  // record + 1 + the ops of the process placed by earlier blocks and earlier chunks of this block (cnt) + the lower lanes of this
  // chunk holding an op of the same process (64 lane reads, counted in registers).  Ranks come from the LDS bitmap; the record
  // goes out at once.  The next chunk's columns are requested before this one's are used.
  {
    const uint32_t c0 = wave * cpw, c1 = c0 + cpw < chunks ? c0 + cpw : chunks;
    uint32_t* mycnt = cnt + wave * W;
    uint32_t i = c0 * 64u + lane;
    bool in = c0 < c1 && i < n;
    uint32_t n_iv = in ? inv[i] : 0u, n_rt = in ? ret[i] : TBC_POS_CRASHED, n_f = in ? (uint32_t)f[i] : 0u;
    int32_t n_p = in ? proc[i] : 0, n_a = in ? a[i] : 0, n_b = in ? b[i] : 0;
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
      }
    }
  }
  // the sentinels
  for (uint32_t p = tid; p < W; p += kT) {
    Rec hd; hd.inv_rank = 0; hd.ret_rank = 0; hd.opidx = kInf; hd.f = kFNone; hd.a = 0; hd.b = 0; hd.cls = 0; hd.prod = kLookNone;
    Rec tl = hd; tl.inv_rank = kInf; tl.ret_rank = kInf;
    rec[seg[p]] = hd;
    rec[seg[p + 1u] - 1u] = tl;
  }
  wv::threadfence();
  wv::wg_barrier();

  // ---- phase 5: one open op per process: the previous op of the same process must have completed before this one was invoked
  // (the previous record of the list; another wavefront may have written it: agent-scope loads, as pack_kernel's)
  {
    const uint32_t c0 = wave * cpw, c1 = c0 + cpw < chunks ? c0 + cpw : chunks;
    uint32_t err = 0u;
    for (uint32_t c = c0; c < c1; c++) {
      const uint32_t i = c * 64u + lane;
      if (i >= n) continue;
      uint32_t d = sc_dst[i];                                 // (this thread's own store)
      if (d == kInf) continue;                                // slotless: no record, no predecessor
      const wv::gu32* prev = (const wv::gu32*)(const void*)&rec[d - 1u];
      const uint32_t prev_ret = wv::ld32(prev + 1), prev_f = wv::ld32(prev + 3);
      if (prev_f != kFNone && !(prev_ret < sc_inv[i])) err = (uint32_t)TBC_ERR_BAD_HISTORY;
    }
    if (err) wv::lds_or32(&flag[0], err);
  }
  wv::wg_barrier();
  if (tid == 0u) { H->n_ret = R; H->status = wv::lds_ld32(&flag[0]) ? (uint32_t)TBC_ERR_BAD_HISTORY : 0u; }
}

}  // namespace packone
}  // namespace tbc
