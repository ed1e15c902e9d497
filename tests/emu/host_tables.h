// TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  host_tables.h: the per-front tables of a batch built ON THE HOST FROM THEIR
// DEFINITIONS (csrc/tbc_internal.h, csrc/pack_open.hip's header): a second, independent formulation of what pack_kernel /
// open_counts / open_walk / open_dprod / front_meta leave in HBM.  emu_narrow.cpp feeds them to the emulated search kernel,
// emu_walk.cpp compares the emulated front walk (csrc/open_walk_impl.h) with them word for word.
#pragma once
#include <algorithm>
#include <vector>

namespace {

struct Tables {
  std::vector<Hist> hist;
  std::vector<BeamHist> bh;
  std::vector<uint32_t> off, ncr, ret_op, ret_slot;
  std::vector<OpRec> lst, crashed;
  std::vector<uint64_t> twn, rdm, look;
  std::vector<uint8_t> slot8, rk8;
  std::vector<uint32_t> inv_rank, ret_rank;      // per op, at op_off (ret_rank kInf = crashed)
  std::vector<uint64_t> cmem;                    // count form: per history the class records, then the classes' members (tbc_internal.h, kRuleCount)
};

// the per-front tables of every history of the batch, from the definitions
bool build_tables(uint32_t nh, const uint64_t* op_off, const uint32_t* n_process, const uint8_t* f, const int32_t* a, const int32_t* b,
                  const int32_t* process, const uint32_t* inv_pos, const uint32_t* ret_pos, uint32_t MW, uint32_t vpad_in,
                  uint32_t tab_log2_per_op, bool branch, bool compact, Tables& T, bool count = false, uint32_t list_by_ret = 0) {
  // list_by_ret: a front's list in order of completion instead of process-slot order (csrc PackOpenArgs.list_order = 1); 2 = in order
  // of completion with the :write calls after everything else (list_order = 2)
  // count = the COUNT FORM, from its definition (oracle/wgl_count.c states it a third time): live calls on re-used slots (the
  // lowest free one when the process first invokes; a process that crashes hands its slot back), crashed calls with an effect
  // grouped into classes by effect in order of first invocation, a count field of bit_length(n) bits each (never across a word)
  const uint32_t vpad = vpad_in ? vpad_in : 1;
  const uint64_t total = op_off[nh];
  T.hist.assign(nh, Hist{}); T.bh.assign(nh, BeamHist{});
  T.ret_op.assign(total + 1, 0); T.ret_slot.assign(total + 1, 0);
  T.crashed.assign(total + 1, OpRec{0, kFNone, 0, 0});
  T.slot8.assign(slot8_bytes(total, nh), 0); T.rk8.assign(slot8_bytes(total, nh), 0xFF);
  const uint32_t FS = compact ? kFrontCompactWords : front_stride(vpad_in, MW), FM = vpad_in * MW;      // front records (tbc_internal.h)
  T.rdm.assign(total * FS + 1, 0);
  T.look.assign(look_words(total, nh, MW), 0);
  T.inv_rank.assign(total + 1, 0); T.ret_rank.assign(total + 1, kInf);
  uint64_t off_n = 0, lst_n = 0, tab_n = 0;
  for (uint32_t h = 0; h < nh; h++) {
    const uint64_t o = op_off[h];
    const uint32_t n = (uint32_t)(op_off[h + 1] - o);
    Hist& H = T.hist[h]; BeamHist& B = T.bh[h];
    H.op_off = o; H.ret_off = o; H.n_ops = n; H.n_slots = n_process[h]; H.status = 0;
    std::vector<uint32_t> slot(n, 0);
    for (uint32_t i = 0; i < n; i++) slot[i] = (uint32_t)process[o + i];
    if (count) {
      std::vector<int32_t> slot_of(n_process[h] + 1, -1);
      std::vector<uint8_t> used(n_process[h] + 2, 0);
      uint32_t W = 1;
      for (uint32_t i = 0; i < n; i++) {
        const uint32_t p = (uint32_t)process[o + i];
        if (p >= n_process[h]) return false;            // (libtbcheck's build_count_form refuses such a history too: it never reaches the kernels in the count form)
        if (ret_pos[o + i] == TBC_POS_CRASHED) { if (slot_of[p] >= 0) { used[slot_of[p]] = 0; slot_of[p] = -1; } slot[i] = 0; continue; }
        if (slot_of[p] < 0) { uint32_t sl = 0; while (used[sl]) sl++; used[sl] = 1; slot_of[p] = (int32_t)sl; W = std::max(W, sl + 1); }
        slot[i] = (uint32_t)slot_of[p];
      }
      H.n_slots = W; H.flags = kHistCount;
    }
    if (H.n_slots > 64 * MW) return false;
    // ranks
    std::vector<std::pair<uint32_t, uint32_t>> rets;
    for (uint32_t i = 0; i < n; i++) if (ret_pos[o + i] != TBC_POS_CRASHED) rets.push_back({ret_pos[o + i], i});
    std::sort(rets.begin(), rets.end());
    const uint32_t R = (uint32_t)rets.size();
    H.n_ret = R;
    std::vector<uint32_t> ret_rank(n, kInf), inv_rank(n, 0);
    for (uint32_t r = 0; r < R; r++) { ret_rank[rets[r].second] = r; T.ret_op[o + r] = rets[r].second; T.ret_slot[o + r] = slot[rets[r].second]; }
    { uint32_t r = 0; for (uint32_t i = 0; i < n; i++) { while (r < R && rets[r].first < inv_pos[o + i]) r++; inv_rank[i] = r; } }
    for (uint32_t i = 0; i < n; i++) { T.inv_rank[o + i] = inv_rank[i]; T.ret_rank[o + i] = ret_rank[i]; }
    // per-front lists
    B.off_off = off_n; B.lst_off = lst_n;
    T.off.resize(off_n + n + 2, 0); T.ncr.resize(off_n + n + 2, 0);
    uint32_t* off = T.off.data() + off_n; uint32_t* ncr = T.ncr.data() + off_n;
    uint32_t ncrash = 0;
    for (uint32_t i = 0; i < n && !count; i++) if (ret_rank[i] == kInf && !(f[o + i] == TBC_F_READ && a[o + i] == TBC_NIL)) {
      T.crashed[o + ncrash++] = OpRec{i, (uint32_t)f[o + i] | ((uint32_t)process[o + i] << 8), a[o + i], b[o + i]};
      if (inv_rank[i] < R) ncr[inv_rank[i]]++;
    }
    struct Cls { uint32_t f; int32_t a, b; std::vector<uint64_t> mem; };
    std::vector<Cls> cls;
    if (count) {
      for (uint32_t i = 0; i < n; i++) {
        if (ret_rank[i] != kInf) continue;
        const uint32_t fi_ = f[o + i];
        if (!(fi_ == TBC_F_WRITE || (fi_ == TBC_F_CAS && a[o + i] != b[o + i]))) continue;
        size_t k = 0;
        while (k < cls.size() && !(cls[k].f == fi_ && cls[k].a == a[o + i] && (fi_ != TBC_F_CAS || cls[k].b == b[o + i]))) k++;
        if (k == cls.size()) cls.push_back(Cls{fi_, a[o + i], fi_ == TBC_F_CAS ? b[o + i] : 0, {}});
        cls[k].mem.push_back((uint64_t)inv_rank[i] | ((uint64_t)i << 32));
      }
      B.cmem_off = T.cmem.size(); B.n_classes = (uint32_t)cls.size(); B.top[0] = B.top[1] = 0;
      std::vector<uint64_t> blk(2 * cls.size(), 0ull);
      uint32_t bits = 0;
      for (size_t k = 0; k < cls.size(); k++) {
        uint32_t w = 0;
        while ((1ull << w) <= cls[k].mem.size()) w++;
        if ((bits & 63u) + w > 64u) bits = (bits + 63u) & ~63u;
        if (bits + w > 64u * kCountWords) return false;
        OpRec rec{(uint32_t)blk.size(), cls[k].f | (bits << 8) | (w << 16), cls[k].a, cls[k].b};
        memcpy(&blk[2 * k], &rec, sizeof rec);
        B.top[(bits + w - 1) >> 6] |= 1ull << ((bits + w - 1) & 63u);
        bits += w;
        blk.insert(blk.end(), cls[k].mem.begin(), cls[k].mem.end());
        blk.push_back(~0ull);
        if ((uint32_t)cls[k].mem[0] < R) ncr[(uint32_t)cls[k].mem[0]]++;       // the class is a candidate from its first member's invocation on
      }
      if (blk.size() & 1) blk.push_back(~0ull);
      T.cmem.insert(T.cmem.end(), blk.begin(), blk.end());
    }
    for (uint32_t r = 1; r < R; r++) ncr[r] += ncr[r - 1];
    B.n_crashed = ncrash;
    std::vector<std::vector<uint32_t>> open(R);
    std::vector<std::vector<uint32_t>> open_reads(R);        // (branch lists: live reads are not candidates, the eager rule finds them through rdm)
    for (uint32_t i = 0; i < n; i++) if (ret_rank[i] != kInf) for (uint32_t F = inv_rank[i]; F <= ret_rank[i]; F++) {
      if (branch && f[o + i] == TBC_F_READ) open_reads[F].push_back(i); else open[F].push_back(i);
    }
    uint32_t run = 0;
    for (uint32_t F = 0; F < R; F++) {
      off[F] = run;
      if (list_by_ret >= 16u) std::sort(open[F].begin(), open[F].end(), [&](uint32_t x, uint32_t y) {      // 16 + W: a :write W ranks later, after its tie
        const uint64_t kx = 2ull * ret_rank[x] + (f[o + x] == TBC_F_WRITE ? 2ull * (list_by_ret - 16u) + 1ull : 0ull);
        const uint64_t ky = 2ull * ret_rank[y] + (f[o + y] == TBC_F_WRITE ? 2ull * (list_by_ret - 16u) + 1ull : 0ull);
        return kx < ky; });
      else if (list_by_ret == 2u) std::sort(open[F].begin(), open[F].end(), [&](uint32_t x, uint32_t y) {
        const bool wx = f[o + x] == TBC_F_WRITE, wy = f[o + y] == TBC_F_WRITE;
        return wx != wy ? wy : ret_rank[x] < ret_rank[y]; });
      else if (list_by_ret) std::sort(open[F].begin(), open[F].end(), [&](uint32_t x, uint32_t y) { return ret_rank[x] < ret_rank[y]; });
      else std::sort(open[F].begin(), open[F].end(), [&](uint32_t x, uint32_t y) { return slot[x] < slot[y]; });
      run += (uint32_t)open[F].size();
    }
    off[R] = run;
    T.lst.resize(lst_n + run + 1); T.twn.resize((lst_n + run + 1) * MW, 0);
    for (uint32_t F = 0; F < R; F++) {
      for (uint32_t k = 0; k < open[F].size(); k++) {
        const uint32_t x = open[F][k];
        const uint32_t px = slot[x];
        T.lst[lst_n + off[F] + k] = OpRec{x, (uint32_t)f[o + x] | (px << 8) | (ret_rank[x] == F ? kAtFront : 0u), a[o + x], b[o + x]};
        // twins: the live calls open here with the same effect that complete earlier
        if (f[o + x] == TBC_F_WRITE || f[o + x] == TBC_F_CAS)
          for (uint32_t y : open[F]) {
            if (y == x || f[o + y] != f[o + x] || a[o + y] != a[o + x] || (f[o + x] == TBC_F_CAS && b[o + y] != b[o + x])) continue;
            if (ret_rank[y] < ret_rank[x]) { const uint32_t py = slot[y]; T.twn[(lst_n + off[F] + k) * MW + (py >> 6)] |= 1ull << (py & 63); }
          }
        // open-read masks by value
        if (f[o + x] == TBC_F_READ && vpad) {
          const uint32_t vi = rdm_index(a[o + x], vpad);
          if (vpad_in && (vi != 0u || a[o + x] == TBC_NIL)) T.rdm[(o + F) * FS + vi * MW + (px >> 6)] |= 1ull << (px & 63);
        }
      }
      for (uint32_t x : open_reads[F]) {
        const uint32_t px = slot[x], vi = rdm_index(a[o + x], vpad);
        if (vpad_in && (vi != 0u || a[o + x] == TBC_NIL)) T.rdm[(o + F) * FS + vi * MW + (px >> 6)] |= 1ull << (px & 63);
      }
    }
    // completion slots as bytes
    uint8_t* s8 = T.slot8.data() + slot8_off(o, h);
    for (uint32_t r = 0; r < R + 16; r++) s8[r] = r < R ? (uint8_t)slot[rets[r].second] : 0;
    uint8_t* k8 = T.rk8.data() + slot8_off(o, h);
    for (uint32_t r = 0; r < R; r++) { const uint32_t x = rets[r].second; k8[r] = f[o + x] == TBC_F_READ ? (uint8_t)rdm_index(a[o + x], vpad_in) : (uint8_t)0xFF; }
    // the rest of each front record: list location, windows of the next 16 ranks
    for (uint32_t F = 0; F < R && compact; F++) {          // the compact form: word 6 = where the list is, word 7 = seven ranks of slot | kind << 6
      uint64_t* rec = T.rdm.data() + (o + F) * FS;
      const uint32_t nl = off[F + 1] - off[F];
      rec[6] = (uint64_t)off[F] | ((uint64_t)(nl & 0xFFu) << 32) | ((uint64_t)((nl + ncr[F]) & 0xFFFFFFu) << 40);
      uint64_t win = 0;
      for (uint32_t l = 0; l < kFrontCompactRanks; l++) {
        const uint32_t k = F + l < R ? k8[F + l] : 0xFFu;
        win |= (uint64_t)((s8[F + l] & 63u) | ((k == 0xFFu ? 7u : (k & 7u)) << 6)) << (9u * l);
      }
      rec[7] = win;
    }
    for (uint32_t F = 0; F < R && !compact; F++) {
      uint64_t* rec = T.rdm.data() + (o + F) * FS + FM;
      rec[0] = (uint64_t)off[F] | ((uint64_t)(off[F + 1] - off[F]) << 32);
      rec[1] = (uint64_t)((off[F + 1] - off[F]) + ncr[F]);
      uint8_t* wb = reinterpret_cast<uint8_t*>(rec + 2);
      for (uint32_t l = 0; l < 16; l++) { wb[l] = s8[F + l]; wb[16 + l] = F + l < R ? k8[F + l] : (uint8_t)0xFF; }
    }
    // lookahead records
    uint64_t* look = T.look.data() + look_off(o, h, MW);
    const uint32_t LW = 1 + MW;
    for (uint32_t t = 0; t < R + kLookPad; t++) {
      if (t >= R) { look[(uint64_t)t * LW] = (uint64_t)(kLookNone << 16 | kLookNone << 24) | (255ull << 32) | (255ull << 40); continue; }
      const uint32_t x = rets[t].second, px = slot[x];
      const uint32_t need = look_need(f[o + x], a[o + x]), prod = look_prod(f[o + x], a[o + x], b[o + x]);
      const uint32_t dinv = std::min(t - inv_rank[x], 255u);
      uint32_t dprod = 255;
      if (need != kLookNone) {
        for (uint32_t i = 0; i < n; i++) {
          if (i == x || look_prod(f[o + i], a[o + i], b[o + i]) != need) continue;
          if (ret_rank[i] == kInf && f[o + i] == TBC_F_READ) continue;
          if (count && ret_rank[i] == kInf) continue;           // (count form: a crashed call holds no slot; its class answers below)
          if (inv_rank[i] <= t && t - inv_rank[i] < kLookahead) dprod = std::min(dprod, t - inv_rank[i]);
          // open at front t (live, or crashed and a candidate) and producing the needed value
          const bool open_here = inv_rank[i] <= t && (ret_rank[i] == kInf || ret_rank[i] >= t);
          if (open_here) { const uint32_t pi = slot[i]; look[(uint64_t)t * LW + 1 + (pi >> 6)] |= 1ull << (pi & 63); }
        }
      }
      uint64_t crashed_producer = 0;            // bit 48: a crashed call of a class producing `need` is invoked by rank t
      for (const Cls& c : cls) if (need != kLookNone && look_prod(c.f, c.a, c.b) == need && (uint32_t)c.mem[0] <= t) crashed_producer = 1ull << 48;
      look[(uint64_t)t * LW] = (uint64_t)(px & 0xFFFFu) | (uint64_t)need << 16 | (uint64_t)prod << 24 | (uint64_t)dinv << 32 | (uint64_t)dprod << 40 | crashed_producer;
    }
    // visited set + stacks
    uint32_t lg = 10;
    while ((1ull << lg) < (uint64_t)tab_log2_per_op * std::max(n, 1u)) lg++;
    B.tab_log2 = lg; B.tab_off = tab_n; B.stack_off = tab_n; B.lst_cap = run; B.status = 0;
    tab_n += 1ull << lg;
    off_n += n + 2; lst_n += run;
  }
  T.lst.resize(lst_n + 1); T.twn.resize((lst_n + 1) * MW);
  return true;
}

}  // namespace
