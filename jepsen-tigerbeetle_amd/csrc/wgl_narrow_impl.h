// wgl_narrow_impl.h -- K5n: the Wing-Gong/Lowe search with SEVERAL HISTORIES PER WAVEFRONT (gfx950).
//
// wgl_beam.hip gives a history a whole wavefront; under the dominance rules the search has become nearly greedy
// (about one round per :write / :cas of the history) and at ~6 calls in flight a round keeps ~4 of the 64 lanes
// busy: the kernel is bound by the ~460 vector instructions a round issues for those 4 lanes.  Here a wavefront is
// cut into H = 64 / L groups of L lanes (L = 8 or 16, or 32) and every group searches its own history: the
// instruction stream and the memory trips of a round serve H searches.
//
// A group runs the schedule oracle/wgl_beam.c specifies for ONE config per iteration and L pairs per round
// (wgl_beam_check_rp(K = 1, round_pairs = L)): pop the most recent config; its open calls, last to first, are its
// pairs, L consecutive pairs a round, one (config, open call) pair per lane; a round's new configs are pushed in
// pair order.  Eager reads, twin rule and lookahead as in wgl_beam.hip.  Verdict, failing op, witness and every
// counter are that schedule's, bit for bit (tests/: the emulated build on the CPU, the device build on the GPU).
// With one parent per round no two pairs of a round produce the same config, so the "same new config from two
// lanes" resolution of the wide kernel has nothing to do here.
//
// What used to be wave-uniform scalars (stack depth, visited-set address and size, the parent config, counters) is
// uniform per GROUP and lives in vector registers (each lane holds its group's copy) or, when only one lane needs
// it, in the group's slice of LDS: 64-bit counters are LDS adds without return by the group's first lane.  Groups
// advance independently: each wave iteration is one round for every group that has a parent, a pop for those that
// need one; ballots are read per group (bits gbase .. gbase + L - 1).  Cold paths (growing a visited set, the
// results) are done for one group at a time by the whole wavefront.
//
// The body is plain per-lane C++ over the primitives of wave_env.h, so tests/emu/ can run this very file on the CPU.
#pragma once
#include "tbc_internal.h"
#include "wave_env.h"

namespace tbc {
namespace narrow {

constexpr uint32_t kNone = 0xFFFFFFFFu;

// group flags
enum : uint32_t { F_ACTIVE = 1u, F_NEED_POP = 2u, F_LOOK = 4u, F_NEED_GROW = 8u, F_CAND = 16u };

// LDS words of a group: counters and results, then the ring of the most recent pushes
enum : uint32_t {
  G_DSTACK = 0, G_PROBES = 2, G_EXPANDED = 4, G_ROUNDS = 6, G_VERDICT = 8, G_CAUSE = 9, G_MAXF = 10, G_MAXSP = 11,
  G_WINPAR = 12, G_WINOP = 13, G_WINSTATE = 14,
  G_STALLF = 15,    // BeamArgs.stall_checks: the greatest front at the last look at the clock ...
  G_STALLN = 55,    // ... and for how many looks it has not moved
  G_T0 = 52,        // 2 words: when the history was taken up (time limit)
  G_NEXT = 54,      // the work item a finished group takes next
  G_PF = 16,        // 4 words: front, list offset, live and all open calls of the child whose candidates are fetched ahead
  G_CLAIM = 20,     // 32 words: who takes the empty entry of a bucket (by bucket number mod 32) this round
  G_RING = 56
};
WV_HD constexpr uint32_t ring_size(uint32_t L) { return L <= 8u ? 8u : L; }      // (>= L: a round's pushes never clash.  Round 5: 8 entries for 8 lanes, not 16: 5.9 KB of LDS a wavefront instead of 8.7 -- the same rate on the device, alone and beside another batch's pack: profiles/r05_ring8_ab.txt)
// ring entry: pos idx off nlive cnt (u32), k0 (u64), M[mw] (u64), the window word(s) of the config's front (u64: 4, or 1 compact)
// (count form: + the config's count vector, 2 x u64)
WV_HD constexpr uint32_t group_words(uint32_t mw, uint32_t L, bool cf = false, bool cnt = false) { return G_RING + ring_size(L) * (5u + 2u + 2u * mw + (cf ? 2u : 8u) + (cnt ? 2u * kCountWords : 0u)); }
// + the lookahead staging of a round (wave-wide): c_fi c_st c_lo (u32 x 64), c_M (u64 x 64 x mw); count form: + c_rt (u32 x 64: the prefix target)
WV_HD constexpr uint32_t narrow_lds_words(uint32_t mw, uint32_t L, bool cf = false, bool cnt = false) { return (64u / L) * group_words(mw, L, cf, cnt) + 64u * 3u + 64u * 2u * mw + (cnt ? 64u : 0u); }

WV_DEV uint32_t key_hash32(uint64_t k0, const uint64_t* M, int mw) {      // as wgl_beam.hip (a table a wide run grew is the same table)
  uint32_t h = (uint32_t)k0 * 0x9E3779B1u ^ (uint32_t)(k0 >> 32) * 0x85EBCA77u;
  for (int j = 0; j < mw; j++) {
    h = (h << 13) | (h >> 19);
    h ^= (uint32_t)M[j] * 0xC2B2AE3Du ^ (uint32_t)(M[j] >> 32) * 0x27D4EB2Fu;
  }
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
  return h;
}

// EPOCH TAGS.  A batch's visited sets used to be zeroed before every pass (28 GB of memset for the bench's batch, 2.6x the
// algorithmic bytes of the search itself).  Instead the pass number (BeamArgs.epoch, 1..255) rides in bits 24..31 of every key's
// low word (front + 1 < 2^24), and an entry whose tag is another pass's reads as EMPTY: the arena is zeroed once in 255 passes.
// Zero stays empty too (the growth pool, scratch arenas), and epoch 0 is the untagged form: empty = zero.
WV_DEV bool entry_empty(uint32_t x, uint32_t etag) { return x == 0u || (x & 0xFF000000u) != etag; }
constexpr uint32_t kFrontMask = 0x00FFFFFFu;
static_assert(kNarrowMaxOps + 1 <= kFrontMask, "a history the narrow kernel takes has front + 1 inside the key's 24 bits (tbc_api.hip refuses longer ones)");
// count form: field-wise x >= y over packed count vectors, top = the top bit of every field (oracle/wgl_count.c, counts_ge)
WV_DEV bool counts_ge(const uint64_t (&x)[kCountWords], const uint64_t (&y)[kCountWords], const uint64_t (&top)[kCountWords]) {
  bool ge = true;
  WV_UNROLL
  for (uint32_t w = 0; w < kCountWords; w++) {
    const uint64_t t = (x[w] | top[w]) - (y[w] & ~top[w]);
    ge = ge && ((((x[w] & ~y[w]) | (~(x[w] ^ y[w]) & t)) & top[w]) == top[w]);
  }
  return ge;
}

template <int MW>
WV_DEV bool mask_bit(const uint64_t (&M)[MW], uint32_t p) {
  bool b = false;
  WV_UNROLL
  for (int j = 0; j < MW; j++) if ((p >> 6) == (uint32_t)j) b = (M[j] >> (p & 63u)) & 1ull;
  return b;
}
template <int MW>
WV_DEV void mask_set(uint64_t (&M)[MW], uint32_t p) {
  WV_UNROLL
  for (int j = 0; j < MW; j++) if ((p >> 6) == (uint32_t)j) M[j] |= 1ull << (p & 63u);
}
template <int MW>
WV_DEV void mask_clear(uint64_t (&M)[MW], uint32_t p) {
  WV_UNROLL
  for (int j = 0; j < MW; j++) if ((p >> 6) == (uint32_t)j) M[j] &= ~(1ull << (p & 63u));
}

// register / cas-register / mutex on immediates (device_common.h Model::ok / apply, REGF form)
WV_DEV bool reg_ok(int32_t st, uint32_t f, int32_t a) {
  return f == TBC_F_WRITE || (f == TBC_F_READ && (a == TBC_NIL || a == st)) || (f == TBC_F_CAS && a == st) ||
         (f == TBC_F_ACQUIRE && st == 0) || (f == TBC_F_RELEASE && st == 1);
}
WV_DEV int32_t reg_apply(int32_t st, uint32_t f, int32_t a, int32_t b) {
  return f == TBC_F_WRITE ? a : (f == TBC_F_CAS ? b : (f == TBC_F_ACQUIRE ? 1 : (f == TBC_F_RELEASE ? 0 : st)));
}

// ---- cold path: group `gsel`'s history moves to a 4x larger visited set (and stacks) from the batch's growth pool, done
// by the whole wavefront (as wgl_beam.hip's grow_visited_set).  In / out: the group's table, stack, second stack and
// capacity, wave-uniform.  Slot numbers change: the caller empties the group's ring.
template <int MW, bool CNT, class ColdArgs>
WV_DEV bool grow_group(ColdArgs C, wv::gu64*& tab_u, wv::gu32*& stack_u, wv::gu32*& dstack_u, uint32_t& cap_log2_u,
                       uint32_t sp, uint32_t dsp, uint32_t lane, uint32_t etag, bool links) {
  constexpr uint32_t CWn = CNT ? kCountWords : 0u;        // count form: the count words ride behind the mask words (not hashed)
  constexpr uint32_t KW = MW + 1 + CWn, EW = KW + 1;
  const wv::gu64* tab = tab_u;
  const wv::gu32* stack = stack_u;
  const wv::gu32* dstack = dstack_u;
  const uint32_t cap_log2 = cap_log2_u;
  const uint64_t old_cap = 1ull << cap_log2, new_cap = old_cap << 2;
  const uint64_t need = new_cap * EW + new_cap / 2 + (dstack ? new_cap / 2 : 0) + old_cap / 2;   // keys + parents, stack(s), slot translation
  uint64_t* const pool = C->pool;
  if (!pool || cap_log2 + 2 > 31 || cap_log2 + 2 > C->max_tab_log2) return false;
  unsigned long long base = 0;
  if (lane == 0) base = atomicAdd(C->pool_cursor, (unsigned long long)need);
  base = wv::readlane64(base, 0);
  if (base + need > C->pool_words) return false;
  wv::gu64* ntab = (wv::gu64*)pool + base;
  wv::gu64* npar = ntab + new_cap * KW;
  const wv::gu64* opar = tab + old_cap * KW;
  wv::gu32* nstack = (wv::gu32*)(ntab + new_cap * EW);
  wv::gu32* ndstack = dstack ? nstack + new_cap : nullptr;
  wv::gu32* remap = nstack + new_cap + (dstack ? new_cap : 0);
  const uint32_t nbmask = (uint32_t)((new_cap >> 2) - 1);
  WV_NOUNROLL
  for (uint64_t s = lane; s < old_cap; s += 64) {
    const wv::gu64* e = tab + s * KW;
    const uint64_t k0 = wv::ld64(e);
    if (entry_empty((uint32_t)k0, etag)) continue;
    uint64_t Mx[MW + CWn];
    WV_UNROLL
    for (int j = 0; j < MW + (int)CWn; j++) Mx[j] = wv::ld64(e + 1 + j);
    uint32_t b = key_hash32(k0, Mx, MW) & nbmask, idx = 0;
    for (bool placed = false; !placed; b = (b + 1u) & nbmask) {
      WV_NOUNROLL
      for (uint32_t t = 0; t < 4 && !placed; t++) {
        wv::gu64* ne = ntab + ((uint64_t)b * 4 + t) * KW;
        if (wv::cas64_from_zero(ne, k0) == 0ull) {
          WV_UNROLL
          for (int j = 0; j < MW + (int)CWn; j++) wv::st64(ne + 1 + j, Mx[j]);
          idx = b * 4 + t; placed = true;
        }
      }
    }
    wv::st32(remap + s, idx);
  }
  wv::wg_fence();
  wv::barrier();
  WV_NOUNROLL
  for (uint64_t s = lane; links && s < old_cap; s += 64) {       // parent links -> new slot numbers (none are kept without a witness: the old table may have no room for them at all)
    if (entry_empty((uint32_t)wv::ld64(tab + s * KW), etag)) continue;
    const uint64_t pw = wv::ld64(opar + s);
    const uint64_t npw = (uint32_t)pw != kNone ? ((uint64_t)wv::ld32(remap + (uint32_t)pw) | (pw & 0xFFFFFFFF00000000ull)) : pw;
    wv::st64(npar + wv::ld32(remap + s), npw);
  }
  WV_NOUNROLL
  for (uint32_t i = lane; i < sp; i += 64) {
    const uint32_t x = wv::ld32(stack + i);
    wv::st32(nstack + i, wv::ld32(remap + x));
  }
  WV_NOUNROLL
  for (uint32_t i = lane; i < dsp; i += 64) wv::st32(ndstack + i, wv::ld32(remap + wv::ld32(dstack + i)));
  wv::wg_fence();
  wv::barrier();
  tab_u = ntab; stack_u = nstack; dstack_u = ndstack; cap_log2_u = cap_log2 + 2u;
  return true;
}

// One wavefront: H = 64 / L histories, work items wave_idx * H .. + H - 1 of A.work.
// CF: the front records are the compact 64 B ones (tbc_internal.h; MW = 1)
// CNT: the count form (tbc_internal.h, kRuleCount; the schedule is oracle/wgl_count.c's at one config per iteration and L pairs per round)
template <int MW, int L, bool CF = false, bool CNT = false>
WV_DEV void narrow_wave(const BeamArgs& A, const uint32_t wave_idx, uint32_t* lds, const uint32_t lane) {
  static_assert(!CF || MW == 1, "compact front records have one mask word");
  constexpr uint32_t WN = CF ? 1u : 4u;          // window words per config
  using wv::gu32;
  using wv::gu64;
  constexpr uint32_t CWn = CNT ? kCountWords : 0u;
  constexpr uint32_t H = 64u / L, RS = ring_size(L), KW = MW + 1 + CWn, GW = group_words(MW, L, CF, CNT);
  static_assert(L == 4 || L == 8 || L == 16 || L == 32, "lanes per history");
  const uint32_t li = lane & (L - 1u), gbase = lane & ~(L - 1u), g = lane / L;
  const uint32_t below = (1u << li) - 1u;                       // the group's lower lanes, as group bits
  constexpr uint32_t GMASK = L >= 32 ? 0xFFFFFFFFu : ((1u << (L & 31)) - 1u);
  const auto grp = [gbase](uint64_t bal) -> uint32_t { return (uint32_t)(bal >> gbase) & GMASK; };

  uint32_t* const GS = lds + g * GW;
  uint32_t* const r_pos = GS + G_RING;          // stack position mirrored in this ring slot (kNone = empty)
  uint32_t* const r_idx = r_pos + RS;
  uint32_t* const r_off = r_idx + RS;
  uint32_t* const r_nlive = r_off + RS;
  uint32_t* const r_cnt = r_nlive + RS;
  uint64_t* const r_k0 = reinterpret_cast<uint64_t*>(r_cnt + RS);
  uint64_t* const r_M = r_k0 + RS;
  uint64_t* const r_W = r_M + RS * MW;          // windows of completion slots / read kinds from the config's front on
  uint64_t* const r_C = r_W + RS * WN;          // count form: the config's count vector
  uint32_t* const c_fi = lds + H * GW;          // this round's new configs for the lookahead, wave-wide
  uint32_t* const c_st = c_fi + 64;
  uint32_t* const c_lo = c_st + 64;
  uint64_t* const c_M = reinterpret_cast<uint64_t*>(c_lo + 64);
  uint32_t* const c_rt = reinterpret_cast<uint32_t*>(c_M + 64 * MW);      // count form: the child's history's prefix target

  // ---- the group's history (a group takes a new one whenever it has finished one: A.next_work)
  const uint32_t rules = A.rules, vpad = A.vpad;
  const uint32_t etag = A.epoch << 24;          // (see entry_empty)
  // parent links ({parent entry, op} per config, behind the keys) are what a witness is read from -- and nothing else is: a
  // caller who wants no witness gets none written (an 8 B store to a line of its own per new config: 7 % of the launch), and the
  // batch's arena then has no room for them at all (BeamArgs.tab_stride)
  const bool links = A.witness != nullptr;
  const bool look_avail = A.look != nullptr && A.dstack != nullptr;
  bool has = false;
  uint32_t hidx = 0, op_off = 0, R = 0, lst_off = 0, look_lo = 0, cap_log2 = 10;
  // count form: where the history's block of cmem[] starts (class records, then their members), the top bit of every count
  // field, the completions a config must pass for the search to end VALID (all of them, or a prefix), exact / relaxed
  uint32_t cmem_lo = 0, RT = 0;
  uint64_t top[kCountWords] = {0ull, 0ull};
  const bool relaxed = CNT && A.count_mode == kCountRelaxed;
  uint64_t slot8_lo = 0;
  gu64* tab = (gu64*)A.tab;
  gu32* stack = (gu32*)A.stack;
  uint32_t flags = 0, sp = 0, dsp = 0, visited = 0;
  uint32_t room = 0x7FFFFFFFu;          // step limit: probes left before it, saturated (refreshed from the 64-bit total every 64 iterations)
  const auto init_group = [&](uint32_t wslot) {
    has = wslot < A.n_work;
    hidx = has ? A.work[wslot] : 0u;
    const Hist* const Hd = A.hist + hidx;
    const BeamHist* const Bd = A.bh + hidx;
    op_off = has ? (uint32_t)Hd->op_off : 0u;
    R = has ? Hd->n_ret : 0u;
    const uint32_t status = has ? (Hd->status | Bd->status) : 0u;
    lst_off = has ? (uint32_t)Bd->lst_off : 0u;
    tab = (gu64*)A.tab + (has ? Bd->tab_off : 0ull) * A.tab_stride;
    stack = (gu32*)A.stack + (has ? Bd->stack_off : 0ull);
    cap_log2 = has ? Bd->tab_log2 : 10u;
    look_lo = (uint32_t)look_off(op_off, hidx, (uint32_t)MW);           // u64 units into A.look
    slot8_lo = slot8_off(op_off, hidx);
    RT = R;
    if constexpr (CNT) {
      cmem_lo = has ? (uint32_t)Bd->cmem_off : 0u;
      top[0] = has ? Bd->top[0] : 0ull; top[1] = has ? Bd->top[1] : 0ull;
      const uint32_t tg = has ? Bd->target : 0u;
      if (tg != 0u && tg < R) RT = tg;
    }
    flags = 0; sp = 0; dsp = 0; visited = 0;
    int32_t verdict0 = -2;
    uint32_t f0 = 0;                                // the root's front
    uint64_t M0[MW];
    WV_UNROLL
    for (int j = 0; j < MW; j++) M0[j] = 0;
    if (!has) verdict0 = TBC_UNKNOWN;               // (no history: nothing is written for this group)
    else if (status != 0) verdict0 = TBC_UNKNOWN;
    else if (R == 0) verdict0 = TBC_VALID;
    else {
      if (rules & kRuleBranch) {
        // branch lists: the root in normal form -- the front moves past every completion of a read the initial state allows,
        // the reads of that kind still open at its front are linearized (every lane of the group walks the records alike)
        const uint32_t vi0 = rdm_index(A.init_state, vpad);
        for (bool going = true; going && f0 < R;) {
          const uint64_t* fr = A.rdm + ((uint64_t)op_off + f0) * A.front_words + (CF ? 0u : vpad * MW);
          const uint64_t wk[2] = {CF ? fr[7] : fr[4], CF ? 0ull : fr[5]};
          uint32_t d = 0;
          for (; d < (CF ? kFrontCompactRanks : 16u) && f0 < R; d++, f0++) {
            const uint32_t rk = CF ? (uint32_t)(wk[0] >> (9u * d + 6u)) & 7u : (uint32_t)(wk[d >> 3] >> (8u * (d & 7u))) & 0xFFu;
            if (!(rk == 0u || rk == vi0)) { going = false; break; }
          }
        }
        if (f0 < R) {
          const uint64_t* row = A.rdm + ((uint64_t)op_off + f0) * A.front_words;
          WV_UNROLL
          for (int j = 0; j < MW; j++) M0[j] = row[j] | row[vi0 * MW + j];
        }
      }
      if (f0 == R) verdict0 = TBC_VALID;            // (only reads the initial state allows: nothing to search)
      else {
        const uint64_t k0 = (uint64_t)((f0 + 1u) | etag) | ((uint64_t)(uint32_t)A.init_state << 32);
        const uint32_t idx = (key_hash32(k0, M0, MW) & (uint32_t)((1ull << (cap_log2 - 2)) - 1ull)) * 4u;
        if (li == 0) {                              // root config: first entry of its bucket, on the stack
          gu64* e = tab + (uint64_t)idx * KW;
          wv::own_st64(e, k0);
          WV_UNROLL
          for (int j = 0; j < MW; j++) wv::own_st64(e + 1 + j, M0[j]);
          if constexpr (CNT) { wv::own_st64(e + 1 + MW, 0ull); wv::own_st64(e + 2 + MW, 0ull); }
          if (links) wv::own_st64(tab + ((uint64_t)KW << cap_log2) + idx, (uint64_t)kNone | ((uint64_t)kNone << 32));
          wv::own_st32(stack, idx);
        }
        sp = 1; visited = 1;
        flags = F_ACTIVE | F_NEED_POP | (look_avail ? F_LOOK : 0u);
      }
    }
    {
      const auto C = wv::cold(A);
      const uint64_t ms = C->max_steps;
      room = (ms && ms < 0x7FFFFFFFull) ? (uint32_t)ms : 0x7FFFFFFFu;
      if (li == 0) {
        const uint64_t ds = (look_avail && has) ? (uint64_t)((gu32*)A.dstack + Bd->stack_off) : 0ull;
        const uint64_t t0 = C->time_limit_ticks ? wv::clock100mhz() : 0ull;
        GS[G_DSTACK] = (uint32_t)ds; GS[G_DSTACK + 1] = (uint32_t)(ds >> 32);
        GS[G_PROBES] = 0; GS[G_PROBES + 1] = 0; GS[G_EXPANDED] = 0; GS[G_EXPANDED + 1] = 0; GS[G_ROUNDS] = 0; GS[G_ROUNDS + 1] = 0;
        GS[G_VERDICT] = (uint32_t)verdict0; GS[G_CAUSE] = (uint32_t)TBC_CAUSE_NONE; GS[G_MAXF] = f0 < R ? f0 : 0u; GS[G_MAXSP] = sp;
        GS[G_WINPAR] = kNone; GS[G_WINOP] = kNone; GS[G_WINSTATE] = (uint32_t)A.init_state;
        GS[G_T0] = (uint32_t)t0; GS[G_T0 + 1] = (uint32_t)(t0 >> 32);
        GS[G_STALLF] = f0 < R ? f0 : 0u; GS[G_STALLN] = 0u;
      }
    }
    for (uint32_t i = li; i < RS; i += L) r_pos[i] = kNone;
  };
  init_group(wave_idx * H + g);
  // the parent config of the group (a copy in every lane of the group) and the round it is in
  uint32_t p_fi = 0, pslot = 0, poff = 0, nlive = 0, cnt = 0, base = 0;
  int32_t p_st = 0;
  bool p_hot = false;                    // count form: the parent is hot (bit 30 of its state word)
  uint64_t Cp[kCountWords] = {0ull, 0ull};
  uint64_t Mp[MW];
  WV_UNROLL
  for (int j = 0; j < MW; j++) Mp[j] = 0;
  // this lane's candidate of the group's NEXT round (valid while F_CAND): the open call's record, its twin mask, and
  // They are loaded one iteration ahead --
  // in the probe's trip when the next (parent, round) can be told (same parent; or the highest viable child, if it is
  // kept; or the entry below on the stack when nothing is viable), else by an iteration that does nothing else
  OpRec c_oi; c_oi.op = 0; c_oi.f_slot = kFNone; c_oi.a = 0; c_oi.b = 0;
  uint64_t c_tw[MW];
  WV_UNROLL
  for (int j = 0; j < MW; j++) c_tw[j] = 0;
  // the windows of the parent's front (from its front record): process slots and read kinds of ranks p_fi .. p_fi + 15
  uint64_t p_ws0 = 0, p_ws1 = 0, p_wk0 = ~0ull, p_wk1 = ~0ull;
  const uint32_t FW = A.front_words, FM = vpad * MW;          // u64 words per front record; where its list location starts
  const bool eager = (rules & kRuleEager) != 0u, twin = (rules & kRuleTwin) != 0u;
  // Loads that are in flight across other work are issued UNCONDITIONALLY, from addresses clamped into the history's own
  // arenas, and what they bring is judged where it is used: a load under a divergent `if` with a default on the other path
  // makes the compiler merge the two right behind the load -- a full s_waitcnt there, the trip no longer overlaps anything.
  const uint64_t* const tw_base = twin ? A.twn : (const uint64_t*)A.tab;            // (a valid address when the rule is off)
  const uint64_t* const lk_base = look_avail ? A.look : (const uint64_t*)A.tab;
  // candidate number cnt_ - 1 - (base_ + li) of the config at front F whose list starts at poff_ (ok_ = there is one)
  const auto load_cand = [&](bool ok_, uint32_t poff_, uint32_t nlive_, uint32_t cnt_, uint32_t base_) {
    const uint32_t cd_ = base_ + li;
    const bool have = ok_ && cd_ < cnt_;
    const uint32_t c_ = have ? cnt_ - 1u - cd_ : 0u;
    const bool live_ = have ? c_ < nlive_ : true;
    const uint32_t po = have ? poff_ : 0u;
    // (count form: the candidates past the live calls are the CLASSES of crashed calls, whose records head the history's block of cmem[])
    const OpRec* rp = live_ ? A.lst + ((uint64_t)lst_off + po + c_)
                            : (CNT ? reinterpret_cast<const OpRec*>(A.cmem + cmem_lo) + (c_ - nlive_) : A.crashed + ((uint64_t)op_off + (c_ - nlive_)));
    c_oi = *rp;
    const uint64_t* tw = tw_base + ((twin && live_) ? ((uint64_t)lst_off + po + c_) * MW : 0ull);
    WV_UNROLL
    for (int j = 0; j < MW; j++) c_tw[j] = tw[j];
  };
  // ---- results of the groups that have just finished (sel: mine has).  The witness (only when asked for: the parent chain
  // is thousands of dependent loads) is walked by all of them at once, each lane following its own group's chain; the configs
  // of an invalid verdict are collected for one group at a time by the whole wavefront.
  const auto report = [&](bool sel) {
    wv::wait_stores();
    wv::wg_fence();
    wv::barrier();
    const auto C = wv::cold(A);
    const int32_t verdict = (int32_t)GS[G_VERDICT];
    gu64* const par = tab + ((uint64_t)KW << cap_log2);
    uint32_t wlen = 0;
    if (C->witness != nullptr) {
      const uint32_t win_parent = GS[G_WINPAR], win_op = GS[G_WINOP];
      const bool walk = sel && verdict == TBC_VALID && R != 0u && win_parent != kNone;      // (kNone: the root itself passed everything)
      uint32_t id = win_parent;
      bool more = walk;
      wlen = walk ? 1u : 0u;
      while (wv::ballot(more)) {
        if (more) {
          const uint32_t pr = (uint32_t)wv::ld64(par + id);
          if (pr == kNone) more = false; else { wlen++; id = pr; }
        }
      }
      uint32_t* const wit = C->witness + op_off;
      uint32_t w = wlen ? wlen - 1u : 0u;
      if (walk && li == 0) wit[w] = win_op;
      id = win_parent; more = walk;
      while (wv::ballot(more)) {
        if (more) {
          const uint64_t po = wv::ld64(par + id);
          const uint32_t pr = (uint32_t)po;
          if (pr == kNone) more = false;
          else { w--; if (li == 0) wit[w] = (uint32_t)(po >> 32) - 1u; id = pr; }
        }
      }
    }
    uint32_t n_cfg = 0;
    const uint32_t maxf = GS[G_MAXF];
    {
      const uint64_t ib = wv::ballot(sel && verdict == TBC_INVALID && C->cfg != nullptr);
      for (uint32_t gg = 0; gg < H; gg++) {
        if (!((ib >> (gg * L)) & 1ull)) continue;
        const uint32_t src = gg * L;
        const gu64* t_u = (const gu64*)wv::readlane64((uint64_t)tab, src);
        const uint32_t cap_u = wv::readlane(cap_log2, src), h_u = wv::readlane(hidx, src), mf_u = wv::readlane(maxf, src);
        const gu64* par_u = t_u + ((uint64_t)KW << cap_u);
        uint64_t* cfg = C->cfg + (uint64_t)h_u * kCfgCap * (2 + MW);
        const uint64_t ncap = 1ull << cap_u;
        uint32_t n = 0;
        for (uint64_t s0 = 0; s0 < ncap; s0 += 64) {
          const gu64* e = t_u + (s0 + lane) * KW;
          const uint64_t k0 = wv::ld64(e);
          const bool hit = (uint32_t)k0 == ((mf_u + 1u) | etag);
          const uint64_t hb = wv::ballot(hit);
          if (hit) {
            const uint32_t pos = n + (uint32_t)__builtin_popcountll(hb & ((1ull << lane) - 1ull));
            if (pos < kCfgCap) {
              uint64_t* o = cfg + (uint64_t)pos * (2 + MW);
              o[0] = k0 & ~(uint64_t)0xFF000000u;          // (without the epoch tag)
              WV_UNROLL
              for (int j = 0; j < MW; j++) o[1 + j] = wv::ld64(e + 1 + j);
              const uint64_t pw = links ? wv::ld64(par_u + s0 + lane) : (uint64_t)kNone;
              o[1 + MW] = (uint32_t)pw == kNone ? (uint64_t)TBC_NO_OP : (pw >> 32) - 1ull;
            }
          }
          n += (uint32_t)__builtin_popcountll(hb);
        }
        if (g == gg) n_cfg = n;
      }
    }
    if (sel && li == 0) {
      DevResult* const out = C->results + hidx;
      const uint64_t probes = (uint64_t)GS[G_PROBES] | ((uint64_t)GS[G_PROBES + 1] << 32);
      const uint64_t expanded = (uint64_t)GS[G_EXPANDED] | ((uint64_t)GS[G_EXPANDED + 1] << 32);
      const uint64_t rounds = (uint64_t)GS[G_ROUNDS] | ((uint64_t)GS[G_ROUNDS + 1] << 32);
      out->valid = verdict; out->cause = (int32_t)GS[G_CAUSE]; out->max_front = maxf; out->depth = wlen;
      out->final_state = (int32_t)GS[G_WINSTATE]; out->n_configs = n_cfg;
      out->fail_op = TBC_NO_OP; out->prev_ok_op = TBC_NO_OP;
      if (verdict == TBC_INVALID) {
        const uint32_t* ret_op = C->ret_op + C->hist[hidx].ret_off;
        out->fail_op = ret_op[maxf];
        if (maxf) out->prev_ok_op = ret_op[maxf - 1];
      }
      out->steps = probes; out->visited = (uint64_t)visited; out->probes = probes; out->backtracks = expanded;
      out->max_depth = (uint64_t)GS[G_MAXSP]; out->bucket_reads = rounds; out->tab_log2 = cap_log2; out->pad = 0;
      if (verdict != TBC_UNKNOWN && C->progress) wv::count_decided(C->progress_dev, C->progress, 63u);      // (tbc_batch_progress)
    }
  };

  uint32_t iter = 0, arb = 0;
  for (uint32_t i = li; i < 32u; i += L) GS[G_CLAIM + i] = 0u;

  for (;;) {
    wv::barrier();                                 // ring writes of the last round, LDS results
    // ---- groups that have finished their history report it and take the next one off the batch's queue
    const bool fin = has && !(flags & F_ACTIVE);
    if (wv::ballot(fin)) {
      report(fin);
      if (fin && li == 0) GS[G_NEXT] = A.first_dynamic + atomicAdd(A.next_work, 1u);
      wv::barrier();
      if (fin) init_group(GS[G_NEXT]);
      wv::barrier();
    }
    if (!wv::ballot((flags & F_ACTIVE) != 0u)) break;
    iter++;

    // ---- cold: visited sets that must grow first (the parent that did not fit is still on the stack)
    const uint64_t gb = wv::ballot((flags & F_NEED_GROW) != 0u);
    if (gb) {
      const auto C = wv::cold(A);
      wv::wait_stores();                           // the plain stores of the rounds so far are in L2 before the table is re-read
      wv::wg_fence();
      for (uint32_t gg = 0; gg < H; gg++) {
        if (!((gb >> (gg * L)) & 1ull)) continue;
        const uint32_t src = gg * L;
        gu64* t_u = (gu64*)wv::readlane64((uint64_t)tab, src);
        gu32* s_u = (gu32*)wv::readlane64((uint64_t)stack, src);
        uint32_t* const GSg = lds + gg * GW;
        gu32* d_u = (gu32*)((uint64_t)GSg[G_DSTACK] | ((uint64_t)GSg[G_DSTACK + 1] << 32));
        uint32_t cap_u = wv::readlane(cap_log2, src);
        const uint32_t sp_u = wv::readlane(sp, src), dsp_u = wv::readlane(dsp, src);
        const bool ok = grow_group<MW, CNT, decltype(C)>(C, t_u, s_u, d_u, cap_u, sp_u, dsp_u, lane, etag, links);
        if (g == gg) {
          if (ok) {
            tab = t_u; stack = s_u; cap_log2 = cap_u;
            if (li == 0) { GS[G_DSTACK] = (uint32_t)(uint64_t)d_u; GS[G_DSTACK + 1] = (uint32_t)((uint64_t)d_u >> 32); }
            for (uint32_t i = li; i < RS; i += L) r_pos[i] = kNone;       // the ring held old slot numbers
            flags &= ~F_NEED_GROW;
          } else {
            flags &= ~(F_ACTIVE | F_NEED_GROW);
            if (li == 0) { GS[G_VERDICT] = (uint32_t)TBC_UNKNOWN; GS[G_CAUSE] = (uint32_t)TBC_CAUSE_VISITED_FULL; }
          }
        }
      }
      wv::barrier();
    }

    const uint32_t bmask = (uint32_t)((1ull << (cap_log2 - 2)) - 1ull);       // bucket index mask
    const uint32_t full_at = (uint32_t)((1ull << cap_log2) - (1ull << (cap_log2 - 2)));
    gu64* const par = tab + ((uint64_t)KW << cap_log2);

    // ---- pop: the most recent config becomes the group's parent
    if ((flags & (F_ACTIVE | F_NEED_POP)) == (F_ACTIVE | F_NEED_POP)) {
      if (sp == 0u && dsp != 0u) {
        // no linearization through the live configs: those the lookahead set aside become the stack (in the order
        // they were set aside) and the search goes on without lookahead -- an INVALID verdict has then expanded
        // every reachable config exactly once.  The pop follows in the next iteration (the ring is emptied first).
        // (G_DSTACK keeps naming that array: nothing is set aside any more, so it is not written again)
        gu32* const ds = (gu32*)((uint64_t)GS[G_DSTACK] | ((uint64_t)GS[G_DSTACK + 1] << 32));
        if (li == 0) wv::lds_max32(GS + G_MAXSP, dsp - 1u);                // (the deepest stack counts what is left after a pop)
        stack = ds; sp = dsp; dsp = 0u; flags &= ~(F_LOOK | F_CAND);
        for (uint32_t i = li; i < RS; i += L) r_pos[i] = kNone;           // ring entries are keyed by stack position
      } else if (sp == 0u) {
        flags &= ~F_ACTIVE;
        if (li == 0) GS[G_VERDICT] = (uint32_t)TBC_INVALID;
      } else {
        const uint32_t pos = sp - 1u, rs = pos & (RS - 1u);
        uint64_t k0;
        uint32_t raw_idx;
        if (r_pos[rs] == pos) {                  // pushed recently: config (and its front's list) still in the ring
          k0 = r_k0[rs];
          WV_UNROLL
          for (int j = 0; j < MW; j++) Mp[j] = r_M[rs * MW + j];
          raw_idx = r_idx[rs];
          pslot = raw_idx; poff = r_off[rs]; nlive = r_nlive[rs]; cnt = r_cnt[rs];
          p_ws0 = r_W[rs * WN];
          if constexpr (!CF) { p_ws1 = r_W[rs * 4 + 1]; p_wk0 = r_W[rs * 4 + 2]; p_wk1 = r_W[rs * 4 + 3]; }
          if constexpr (CNT) { Cp[0] = r_C[rs * 2]; Cp[1] = r_C[rs * 2 + 1]; }
        } else {
          raw_idx = wv::own_ld32(stack + pos);
          const uint32_t idx = raw_idx;
          const gu64* e = tab + (uint64_t)idx * KW;
          k0 = wv::own_ld64(e);
          WV_UNROLL
          for (int j = 0; j < MW; j++) Mp[j] = wv::own_ld64(e + 1 + j);
          if constexpr (CNT) { Cp[0] = wv::own_ld64(e + 1 + MW); Cp[1] = wv::own_ld64(e + 2 + MW); }
          const uint64_t* fr = A.rdm + ((uint64_t)op_off + (((uint32_t)k0 & kFrontMask) - 1u)) * FW + (CF ? 6u : FM);      // its front's record
          const uint64_t m0 = fr[0];
          pslot = idx; poff = (uint32_t)m0;
          if constexpr (CF) { nlive = (uint32_t)(m0 >> 32) & 0xFFu; cnt = (uint32_t)(m0 >> 40); p_ws0 = fr[1]; }
          else { nlive = (uint32_t)(m0 >> 32); cnt = (uint32_t)fr[1]; p_ws0 = fr[2]; p_ws1 = fr[3]; p_wk0 = fr[4]; p_wk1 = fr[5]; }
        }
        p_fi = ((uint32_t)k0 & kFrontMask) - 1u; p_st = (int32_t)(uint32_t)(k0 >> 32);
        if constexpr (CNT) { p_hot = ((uint32_t)(k0 >> 32) & kHotBit) != 0u; p_st = (int32_t)((uint32_t)(k0 >> 32) & ~kHotBit); }
        if (visited + cnt > full_at) {
          flags |= F_NEED_GROW;                  // room for every pair of this parent?  If not it stays on the stack
        } else {
          sp = pos; base = 0u;
          if (cnt != 0u) flags &= ~F_NEED_POP;       // (only the root can have no candidate at all: the others are not pushed)
          if (li == 0) wv::lds_add64(GS + G_EXPANDED, 1ull);
        }
      }
    }

    // ---- a group whose candidates were not fetched ahead sits this round out and fetches them with the others' (below)
    const bool wants = (flags & (F_ACTIVE | F_NEED_POP | F_NEED_GROW)) == F_ACTIVE;
    const bool inround = wants && (flags & F_CAND);
    const bool fetch_now = wants && !(flags & F_CAND);
    if (fetch_now && li == 0) wv::stat(22, 1);

    // ---- the round: lane li takes pair base + li of the parent = its open call number cnt - 1 - (base + li)
    const uint32_t cd = base + li;
    const bool act = inround && cd < cnt;
    const uint32_t c = cnt - 1u - cd;
    const uint32_t fi = p_fi;
    const int32_t st = p_st;
    const bool live = c < nlive;
    const OpRec oi = c_oi;
    const uint32_t wbase = fi;
    const uint64_t w0 = p_ws0, w1 = p_ws1, k0w = p_wk0, k1w = p_wk1;
    bool dominated = false;
    if (twin && act && live) {
      WV_UNROLL
      for (int j = 0; j < MW; j++) dominated = dominated || (c_tw[j] & ~Mp[j]) != 0ull;
    }
    // count form: a candidate past the live calls is a CLASS of crashed calls; its next member (the parent's count says which) must
    // be invoked by the parent's front -- one dependent load here (inv_rank | op << 32; the sentinel behind the last member is all ones)
    const bool is_cls = CNT && act && !live;
    uint64_t mem = ~0ull;
    if constexpr (CNT) {
      const uint32_t sh = (oi.f_slot >> 8) & 0xFFu, wd = (oi.f_slot >> 16) & 0xFFu;
      const uint32_t kc = (is_cls && !relaxed) ? (uint32_t)(((sh & 64u) ? Cp[1] : Cp[0]) >> (sh & 63u)) & ((1u << wd) - 1u) : 0u;
      mem = A.cmem[(uint64_t)cmem_lo + (is_cls ? oi.op + kc : 0u)];
      if (!is_cls) mem = ~0ull;
    }
    const uint32_t op = is_cls ? (uint32_t)(mem >> 32) : oi.op;
    const uint32_t f = oi.f_slot & 0xFFu, p = is_cls ? 0u : (oi.f_slot >> 8) & kSlotMask;
    const bool lin = !is_cls && mask_bit<MW>(Mp, p);
    // a crashed call has no per-front entry: its twins are every live open call with its effect (they all complete
    // earlier) and the crashed ones invoked before it -- walk the list (crash-heavy histories only)
    if (!CNT && twin && act && !lin && !live && (f == TBC_F_WRITE || f == TBC_F_CAS)) {
      for (uint32_t cc = 0; cc < c && !dominated; cc++) {
        OpRec y = cc < nlive ? A.lst[(uint64_t)lst_off + poff + cc] : A.crashed[(uint64_t)op_off + (cc - nlive)];
        if ((y.f_slot & 0xFFu) != f || y.a != oi.a || (f == TBC_F_CAS && y.b != oi.b)) continue;
        dominated = !mask_bit<MW>(Mp, (y.f_slot >> 8) & kSlotMask);
      }
    }
    bool viable = act && !lin && !dominated && !is_cls && reg_ok(st, f, oi.a);
    if constexpr (CNT) {
      // a hot config only takes calls whose precondition is its state; a crashed :write is not one, and never writes the state it finds
      if (p_hot) viable = viable && (f == TBC_F_READ || f == TBC_F_CAS) && oi.a == st;
      if (is_cls) viable = (uint32_t)mem <= fi && (f == TBC_F_WRITE ? (!p_hot && oi.a != st) : oi.a == st);
    }
    bool observed = !is_cls;               // (a child reached by a crashed call is hot until a read of exactly its new state takes it)
    uint64_t C2[kCountWords] = {Cp[0], Cp[1]};
    // the child: linearize; the front moves past every completion whose call is linearized -- or, under the eager rule,
    // is a read the child's state allows (value nil or the state: it is open, so the rule takes it).  Slots and read kinds
    // of the next ranks came with the candidate: no memory access here.  The reads the rule takes that are still open at
    // the child's front are one row entry (the trip below).
    int32_t st2 = st;
    uint32_t fi2 = fi;
    uint64_t M2[MW];
    WV_UNROLL
    for (int j = 0; j < MW; j++) M2[j] = Mp[j];
    const auto byte_at = [&](uint64_t x0, uint64_t x1, const uint8_t* arr, uint32_t r) -> uint32_t {
      const uint32_t d = r - wbase;
      if (d < 8u) return (uint32_t)(x0 >> (8u * d)) & 0xFFu;
      if (d < 16u) return (uint32_t)(x1 >> (8u * (d - 8u))) & 0xFFu;
      return (uint32_t)arr[slot8_lo + r];
    };
    if (viable) {
      st2 = reg_apply(st, f, oi.a, oi.b);
      if (!is_cls) mask_set<MW>(M2, p);
      if constexpr (CNT) {
        if (is_cls && !relaxed) {            // the class's count goes up: no mask bit, the front stays where the reads let it
          const uint32_t sh = (oi.f_slot >> 8) & 0xFFu;
          if (sh & 64u) C2[1] += 1ull << (sh & 63u); else C2[0] += 1ull << (sh & 63u);
        }
      }
      const uint32_t vis = eager ? rdm_index(st2, vpad) : 0xFFFFu;
      for (;;) {
        uint32_t pp, rk;
        if constexpr (CF) {
          const uint32_t d = fi2 - wbase;
          if (d < kFrontCompactRanks) { const uint32_t e = (uint32_t)(w0 >> (9u * d)) & 0x1FFu; pp = e & 63u; rk = e >> 6; }
          else { pp = (uint32_t)A.slot8[slot8_lo + fi2]; rk = (uint32_t)A.rk8[slot8_lo + fi2]; }
        } else {
          pp = byte_at(w0, w1, A.slot8, fi2);
          rk = eager ? byte_at(k0w, k1w, A.rk8, fi2) : 0xFFu;
        }
        if (mask_bit<MW>(M2, pp)) mask_clear<MW>(M2, pp);
        else {
          if (!eager) break;
          if (!(rk == 0u || rk == vis)) break;
          if (CNT && rk != 0u) observed = true;        // a read of exactly the state completes here: it observes the crashed call
        }
        fi2++;
        if (fi2 == R) break;
      }
    }
    if (inround && li == 0) wv::lds_add64(GS + G_ROUNDS, 1ull);
    // linearizable: the group's lowest pair wins, nothing of this round is inserted
    const uint32_t gsucc = grp(wv::ballot(viable && fi2 >= RT));        // (RT = R unless the count form checks a prefix)
    if (gsucc) {
      if (li == (uint32_t)__builtin_ctz(gsucc)) {
        GS[G_WINPAR] = pslot; GS[G_WINOP] = op; GS[G_WINSTATE] = (uint32_t)st2; GS[G_VERDICT] = (uint32_t)TBC_VALID;
      }
      flags &= ~F_ACTIVE;
    }
    const bool go = viable && !gsucc;
    const uint32_t gv = grp(wv::ballot(go));
    const uint32_t npr = (uint32_t)__builtin_popcount(gv);
    if (li == 0 && npr) wv::lds_add64(GS + G_PROBES, (uint64_t)npr);
    bool limit_hit = false;
    if (npr > room) limit_hit = true; else room -= npr;

    // ---- trip 1: the child's front's record -- the reads the eager rule takes there, where its open-call list is, and the
    // windows its own children will advance over: one line -- and, in the same trip, the lookahead records of every viable
    // child (8 lanes per child, one rank each, staged wave-wide: the fronts now, states and masks when the rows are in)
    const bool lkme = go && (flags & F_LOOK);
    const uint64_t lk = wv::ballot(lkme);
    const uint32_t ci = (uint32_t)__builtin_popcountll(lk & ((1ull << lane) - 1ull)), nn0 = (uint32_t)__builtin_popcountll(lk);
    if (lkme) { c_fi[ci] = fi2; c_lo[ci] = look_lo; if constexpr (CNT) c_rt[ci] = RT; }
    wv::barrier();
    const uint32_t f3 = go ? fi2 : 0u;
    const uint64_t* const fr = A.rdm + ((uint64_t)op_off + f3) * FW;
    uint32_t co0, cnl, ccnt;
    uint64_t cw[WN];
    uint64_t lw0[2], lpm[2][MW];
    {
      const uint32_t vi = (eager && go) ? rdm_index(st2, vpad) : 0u;
      uint64_t r0[MW], rv[MW];
      WV_UNROLL
      for (int bt = 0; bt < 2; bt++) {
        const uint32_t cc = 8u * bt + (lane >> 3), jr = lane & 7u;
        const bool val = cc < nn0;
        const uint32_t lo_ = val ? c_lo[cc] : (look_avail ? look_lo : 0u), fr_ = val ? c_fi[cc] + jr : 0u;
        const uint64_t* rec = lk_base + (uint64_t)lo_ + (uint64_t)fr_ * (MW + 1);
        lw0[bt] = rec[0];
        WV_UNROLL
        for (int w = 0; w < MW; w++) lpm[bt][w] = rec[1 + w];
      }
      uint64_t m0, m1 = 0;
      if constexpr (CF) { m0 = fr[6]; cw[0] = fr[7]; }
      else {
        m0 = fr[FM]; m1 = fr[FM + 1];
        WV_UNROLL
        for (int t = 0; t < 4; t++) cw[t] = fr[FM + 2 + t];
      }
      WV_UNROLL
      for (int j = 0; j < MW; j++) { r0[j] = eager ? fr[j] : 0ull; rv[j] = eager ? fr[vi * MW + j] : 0ull; }
      if constexpr (CNT) {
        WV_UNROLL
        for (int j = 0; j < MW; j++) observed = observed || (go && (rv[j] & ~M2[j]) != 0ull);
      }
      WV_UNROLL
      for (int j = 0; j < MW; j++) M2[j] |= go ? (r0[j] | rv[j]) : 0ull;
      co0 = (uint32_t)m0;
      if constexpr (CF) { cnl = (uint32_t)(m0 >> 32) & 0xFFu; ccnt = (uint32_t)(m0 >> 40); }
      else { cnl = (uint32_t)(m0 >> 32); ccnt = (uint32_t)m1; }
    }
    // ---- the lookahead's verdicts, before the probe: a child it finds dead is set aside, so it is never the next parent --
    // the candidates fetched below follow the highest child that is ALIVE
    bool dead = false;
    if (lkme) {
      c_st[ci] = (uint32_t)st2;
      WV_UNROLL
      for (int j = 0; j < MW; j++) c_M[ci * MW + j] = M2[j];
    }
    wv::barrier();
    // one batch of 8 children: lane (child cb + lane / 8, rank lane % 8) decides its rank; returns the lanes that found
    // their child's call un-linearizable (wgl_beam.hip states the rule)
    const auto look_batch = [&](uint32_t cb, uint64_t w_0, const uint64_t (&pm)[MW]) -> uint64_t {
      const uint32_t cc = cb + (lane >> 3), j = lane & 7u;
      bool val = cc < nn0;
      if constexpr (CNT) val = val && c_fi[cc] + j < c_rt[cc];       // (completions past a prefix target constrain nothing)
      const int32_t cs = val ? (int32_t)c_st[cc] : 0;
      uint64_t Mc[MW];
      WV_UNROLL
      for (int w = 0; w < MW; w++) Mc[w] = val ? c_M[cc * MW + w] : 0ull;
      uint32_t slot, need, prod, dinv, dprod;
      bool pmhit = false;
      slot = (uint32_t)w_0 & 0xFFFFu; need = (uint32_t)(w_0 >> 16) & 0xFFu; prod = (uint32_t)(w_0 >> 24) & 0xFFu;
      dinv = (uint32_t)(w_0 >> 32) & 0xFFu; dprod = (uint32_t)(w_0 >> 40) & 0xFFu;
      WV_UNROLL
      for (int w = 0; w < MW; w++) pmhit = pmhit || (pm[w] & ~Mc[w]) != 0ull;
      const bool linz = dinv >= j && mask_bit<MW>(Mc, slot);     // open at the config's front and linearized
      // values the calls completing at the ranks before this one can still provide (prefix-OR over the 8 lanes)
      uint32_t acc = (prod != kLookNone && !linz) ? 1u << prod : 0u;
      uint32_t x = wv::row_shr0<1>(acc);
      if (j >= 1u) acc |= x;
      x = wv::row_shr0<2>(acc);
      if (j >= 2u) acc |= x;
      x = wv::row_shr0<4>(acc);
      if (j >= 4u) acc |= x;
      uint32_t before = wv::row_shr0<1>(acc);
      if (j == 0u) before = 0u;
      // (count form: bit 48 = a crashed call producing `need` is invoked by that rank: available whatever the counts)
      const bool ok = need == kLookNone || linz || (int32_t)need == cs || dprod < j || pmhit || ((before >> need) & 1u) || (CNT && ((w_0 >> 48) & 1ull));
      return wv::ballot(val && !ok);
    };
    if (nn0) {
      const uint64_t bad0 = look_batch(0u, lw0[0], lpm[0]);
      if (lkme && ci < 8u) dead = ((bad0 >> (8u * ci)) & 0xFFull) != 0ull;
      if (nn0 > 8u) {
        const uint64_t bad1 = look_batch(8u, lw0[1], lpm[1]);
        if (lkme && ci >= 8u && ci < 16u) dead = ((bad1 >> (8u * (ci - 8u))) & 0xFFull) != 0ull;
      }
      for (uint32_t cb = 16u; cb < nn0; cb += 8u) {          // more than 16 viable children in the wavefront: rare
        const uint32_t cc = cb + (lane >> 3), jr = lane & 7u;
        const bool val = cc < nn0;
        const uint32_t lo_ = val ? c_lo[cc] : (look_avail ? look_lo : 0u), fr_ = val ? c_fi[cc] + jr : 0u;
        const uint64_t* rec = lk_base + (uint64_t)lo_ + (uint64_t)fr_ * (MW + 1);
        uint64_t xpm[MW];
        const uint64_t xw0 = rec[0];
        WV_UNROLL
        for (int w = 0; w < MW; w++) xpm[w] = rec[1 + w];
        const uint64_t badx = look_batch(cb, xw0, xpm);
        if (lkme && ci >= cb && ci < cb + 8u) dead = ((badx >> (8u * (ci - cb))) & 0xFFull) != 0ull;
      }
    }

    // ---- trip 2, issue: the child's bucket of the visited set ...
    const uint64_t k0c = (uint64_t)((fi2 + 1u) | etag) | ((uint64_t)((uint32_t)st2 | ((CNT && !observed) ? kHotBit : 0u)) << 32);
    uint32_t b = key_hash32(k0c, M2, MW) & bmask, idx = 0, full_buckets = 0;
    bool pending = go, fresh = false;
    wv::u32x4 ke[4];                 // MW = 1: the bucket's four 16 B entries
    uint64_t kk[4];                  // MW > 1: their first words
    const auto load_bucket = [&]() {
      const gu64* bp = tab + (uint64_t)b * (4 * KW);
      if constexpr (MW == 1 && !CNT) {
        WV_UNROLL
        for (int t = 0; t < 4; t++) ke[t] = wv::own_ld128(bp + 2 * t);
      } else {
        WV_UNROLL
        for (int t = 0; t < 4; t++) kk[t] = wv::own_ld64(bp + t * KW);
      }
    };
    if (pending) load_bucket();
    // ... and the candidates of the round this group runs next, where that can be told now:
    //   the parent has pairs left -> its next L pairs;  else some child is viable and alive -> the highest such one's (it is on
    //   top of the stack unless it turns out a duplicate);  else nothing will be pushed -> the entry below on the stack
    const uint32_t galive = grp(wv::ballot(go && !dead && ccnt != 0u));
    const bool more = inround && !gsucc && base + L < cnt;
    const bool to_child = inround && !gsucc && !more && galive != 0u;
    const bool to_below = inround && !gsucc && !more && galive == 0u && sp != 0u;
    const uint32_t hv = galive ? 31u - (uint32_t)__builtin_clz(galive) : 0u;
    if (to_child && li == hv) { GS[G_PF] = fi2; GS[G_PF + 1] = co0; GS[G_PF + 2] = cnl; GS[G_PF + 3] = ccnt; }
    wv::barrier();
    bool cand_next = false;
    {
      // ONE candidate fetch per iteration, for everybody: the round a group runs next -- or, for a group that had none
      // ready, the round it wanted to run now
      uint32_t no = poff, nnl = nlive, nct = cnt, nb_ = fetch_now ? base : base + L;
      bool ok_ = more || fetch_now;
      if (to_child) { no = GS[G_PF + 1]; nnl = GS[G_PF + 2]; nct = GS[G_PF + 3]; nb_ = 0u; ok_ = true; }
      if (to_below) {
        const uint32_t pos = sp - 1u, rs = pos & (RS - 1u);
        if (r_pos[rs] == pos) { no = r_off[rs]; nnl = r_nlive[rs]; nct = r_cnt[rs]; nb_ = 0u; ok_ = true; }
      }
      load_cand(ok_, no, nnl, nct, nb_);
      cand_next = ok_;
    }

    // ---- trip 2, consume.  Visited set: find the key in its bucket chain, else take the first empty entry met.  Lanes of
    // different groups never meet (one table per history); two lanes of a group can want the same empty entry (two keys of
    // one bucket, or -- at the root only -- one key twice): they settle it in LDS, the loser looks again.
    for (;;) {
      bool want = false;
      if (pending) {
        uint32_t match = 0, empty = 0;
        if constexpr (MW == 1 && !CNT) {
          const uint32_t k0l = (uint32_t)k0c, k0h = (uint32_t)(k0c >> 32), ml = (uint32_t)M2[0], mh = (uint32_t)(M2[0] >> 32);
          WV_UNROLL
          for (int t = 0; t < 4; t++) {
            if (entry_empty(ke[t].x, etag)) empty |= 1u << t;
            else if (ke[t].x == k0l && ke[t].y == k0h && ke[t].z == ml && ke[t].w == mh) match |= 1u << t;
          }
        } else {
          const gu64* bp = tab + (uint64_t)b * (4 * KW);
          WV_UNROLL
          for (int t = 0; t < 4; t++) {
            if (entry_empty((uint32_t)kk[t], etag)) empty |= 1u << t;
            else if (kk[t] == k0c) {
              bool same = true;
              WV_UNROLL
              for (int j = 0; j < MW; j++) same = same && wv::own_ld64(bp + t * KW + 1 + j) == M2[j];
              if constexpr (CNT) {
                // the Pareto rule: an entry with this key that has used no more of any class dominates the child ("match" = it is dropped);
                // every config of one key lies on this chain, which ends at the first bucket with an empty entry
                if (same) {
                  const uint64_t theirs[kCountWords] = {wv::own_ld64(bp + t * KW + 1 + MW), wv::own_ld64(bp + t * KW + 2 + MW)};
                  same = counts_ge(C2, theirs, top);
                }
              }
              match |= same ? 1u << t : 0u;
            }
          }
        }
        if (match) { idx = b * 4u + (uint32_t)__builtin_ctz(match); pending = false; }
        else if (empty) { idx = b * 4u + (uint32_t)__builtin_ctz(empty); want = true; }
        else {
          b = (b + 1u) & bmask;
          if (++full_buckets > bmask) pending = false;       // every bucket full: cannot happen below the 3/4 fill bound
        }
      }
      if (wv::ballot(want)) {
        // the LOWEST lane that wants an entry of this bucket gets it (so two pairs of a round that produce one and the same
        // config -- the root's nil reads under the eager rule -- resolve as the oracle's pair order does): an LDS max over
        // {arbitration number, 255 - lane}; words of earlier arbitrations hold smaller numbers
        arb++;
        if ((arb & 0xFFFFFFu) == 0u) {                         // the number wraps: clear the claim words (every ~10^7 rounds)
          arb++;
          wv::barrier();
          for (uint32_t i = li; i < 32u; i += L) GS[G_CLAIM + i] = 0u;
          wv::barrier();
        }
        uint32_t* const claim = GS + G_CLAIM + (b & 31u);
        const uint32_t ticket = (arb << 8) | (255u - li);
        if (want) wv::lds_max32(claim, ticket);
        wv::barrier();
        if (want && *claim == ticket) {
          gu64* e = tab + (uint64_t)idx * KW;
          wv::own_st64(e, k0c);
          WV_UNROLL
          for (int j = 0; j < MW; j++) wv::own_st64(e + 1 + j, M2[j]);
          if constexpr (CNT) { wv::own_st64(e + 1 + MW, C2[0]); wv::own_st64(e + 2 + MW, C2[1]); }
          fresh = true; pending = false;
        }
      }
      if (!wv::ballot(pending)) break;
      wv::wait_stores();                                      // a loser must see the winner's entry
      if (pending) load_bucket();
    }
    if (grp(wv::ballot(full_buckets > bmask))) {
      flags &= ~F_ACTIVE;
      if (li == 0) { GS[G_VERDICT] = (uint32_t)TBC_UNKNOWN; GS[G_CAUSE] = (uint32_t)TBC_CAUSE_VISITED_FULL; }
    }
    const bool is_new = fresh;
    if (is_new) {
      if (links) wv::own_st64(par + idx, (uint64_t)pslot | ((uint64_t)(op + 1u) << 32));
      wv::lds_max32(GS + G_MAXF, fi2);
    }
    const uint64_t nb0 = wv::ballot(is_new);
    // ---- push the new configs in pair order: the dead ones aside, the others onto the stack (and into the ring).  A new
    // config whose front has no candidate (branch lists: only reads its state does not allow are open) has no successor:
    // it counts as expanded on the spot and goes nowhere.
    const bool barren = is_new && ccnt == 0u;
    const uint32_t gbar = grp(wv::ballot(barren));
    if (li == 0 && gbar) wv::lds_add64(GS + G_EXPANDED, (uint64_t)__builtin_popcount(gbar));
    dead = dead && !barren;
    const bool keep = is_new && !dead && !barren;
    const uint32_t gdb = grp(wv::ballot(is_new && dead));
    if (is_new && dead) {
      gu32* const ds = (gu32*)((uint64_t)GS[G_DSTACK] | ((uint64_t)GS[G_DSTACK + 1] << 32));
      wv::own_st32(ds + dsp + (uint32_t)__builtin_popcount(gdb & below), idx);
    }
    dsp += (uint32_t)__builtin_popcount(gdb);
    const uint32_t gnb = grp(wv::ballot(keep));
    if (keep) {
      const uint32_t pos = sp + (uint32_t)__builtin_popcount(gnb & below);
      const uint32_t word = idx;
      wv::own_st32(stack + pos, word);
      const uint32_t rs = pos & (RS - 1u);          // at most L <= RS pushes a round: no clash
      r_pos[rs] = pos; r_idx[rs] = word; r_k0[rs] = k0c;
      WV_UNROLL
      for (int j = 0; j < MW; j++) r_M[rs * MW + j] = M2[j];
      r_off[rs] = co0; r_nlive[rs] = cnl; r_cnt[rs] = ccnt;
      WV_UNROLL
      for (uint32_t t = 0; t < WN; t++) r_W[rs * WN + t] = cw[t];
      if constexpr (CNT) { r_C[rs * 2] = C2[0]; r_C[rs * 2 + 1] = C2[1]; }
    }
    sp += (uint32_t)__builtin_popcount(gnb);
    visited += (uint32_t)__builtin_popcount(grp(nb0));
    if (li == 0 && gnb) wv::lds_max32(GS + G_MAXSP, sp);
    {
      // were the candidates fetched above the next round's?  (the highest viable child is the next parent iff it was kept;
      // a group that sat out fetched the round it wanted; anybody else has none)
      // (to_below: every viable child was dead or barren -- none is pushed onto the stack -- unless a dead one... dead ones go aside)
      const bool right = inround ? (cand_next && (more || (to_below && gnb == 0u) || (to_child && ((gnb >> hv) & 1u)))) : fetch_now;
      flags = right ? (flags | F_CAND) : (flags & ~F_CAND);
      if (inround && li == 0) {
        wv::stat(right ? 23 : 24, 1);
        if (!right) wv::stat(to_child ? 25 : (to_below ? 26 : (gsucc ? 28 : 27)), 1);
      }
    }
    if (inround) {
      base += L;
      if (base >= cnt) flags |= F_NEED_POP;
    }
    if (limit_hit && (flags & F_ACTIVE)) {             // the step limit, as after the round that exceeded it
      flags &= ~F_ACTIVE;
      if (li == 0) { GS[G_VERDICT] = (uint32_t)TBC_UNKNOWN; GS[G_CAUSE] = (uint32_t)TBC_CAUSE_STEP_LIMIT; }
    }
    if ((iter & 63u) == 0u) {                          // the clock, and the step room from the 64-bit total
      const auto C = wv::cold(A);
      const uint64_t ms = C->max_steps, limit = C->time_limit_ticks;
      wv::barrier();
      if (ms) {
        const uint64_t done = (uint64_t)GS[G_PROBES] | ((uint64_t)GS[G_PROBES + 1] << 32);
        const uint64_t left = ms - (done < ms ? done : ms);
        room = left < 0x7FFFFFFFull ? (uint32_t)left : 0x7FFFFFFFu;
      }
      if (limit && (flags & F_ACTIVE) && wv::clock100mhz() - ((uint64_t)GS[G_T0] | ((uint64_t)GS[G_T0 + 1] << 32)) > limit) {
        flags &= ~F_ACTIVE;
        if (li == 0) { GS[G_VERDICT] = (uint32_t)TBC_UNKNOWN; GS[G_CAUSE] = (uint32_t)TBC_CAUSE_TIME_LIMIT; }
      }
      const uint32_t stall = C->stall_checks;
      if (stall) {                                     // a history that no longer passes completions (BeamArgs.stall_checks)
        const uint32_t mf = GS[G_MAXF];
        const uint32_t sn = GS[G_STALLF] == mf ? GS[G_STALLN] + 1u : 0u;
        wv::barrier();                                 // (every lane of the group has read the two words)
        if (li == 0) { GS[G_STALLF] = mf; GS[G_STALLN] = sn; }
        if (sn >= stall && (flags & F_ACTIVE)) {
          flags &= ~F_ACTIVE;
          if (li == 0) { GS[G_VERDICT] = (uint32_t)TBC_UNKNOWN; GS[G_CAUSE] = (uint32_t)TBC_CAUSE_STEP_LIMIT; }
        }
      }
    }
  }
}

}  // namespace narrow
}  // namespace tbc
