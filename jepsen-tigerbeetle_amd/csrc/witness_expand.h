// witness_expand.h -- host side of a witness under the eager-read rule: the search kernels return the chain of BRANCHING calls
// only (wgl_narrow_impl.h, wgl_beam.hip: reads the rule absorbs are never pushed); the full linearization order -- what
// knossos.wgl reports as the final configuration's linearized calls (reference: knossos/src/knossos/wgl.clj, the :linearized of
// the last config; jepsen.checker/linearizable hands it on) -- is the chain replayed with the rule applied.  Plain host code, no
// device types: tbc_api.hip calls it on columns copied back from the device, tests/emu/emu_narrow.cpp on the test's own columns.
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>
#include "../../include/tbcheck.h"

namespace tbc {

// The replay both rules share.  After every chain call the front moves past the completions now linearized, then every open live call
// the rule takes in the state reached (`takes(x, state)`) is linearized, again after each move of the front; the calls of one pass come
// out in the order of the front's list: process slot, or -- by_completion (PackOpenArgs.list_order != 0) -- completion.  `apply(op, state)`
// is the model's step.  branch: the root itself starts in normal form (its absorbed calls come first).  proc[] are process slots < n_slots.
// false: the chain names a call twice or one that does not exist.
template <typename State, typename Takes, typename Apply>
inline bool expand_chain_with(uint32_t n, const int32_t* proc, const uint32_t* inv, const uint32_t* ret, uint32_t n_slots, State state, bool branch,
                              bool by_completion, const uint32_t* chain, uint32_t chain_len, Takes takes, Apply apply, std::vector<uint32_t>& out) {
  std::vector<uint32_t> by_ret;                      // completed calls in completion order
  for (uint32_t i = 0; i < n; i++) if (ret[i] != TBC_POS_CRASHED) by_ret.push_back(i);
  std::sort(by_ret.begin(), by_ret.end(), [&](uint32_t x, uint32_t y) { return ret[x] < ret[y]; });
  const uint32_t R = (uint32_t)by_ret.size();
  std::vector<uint32_t> opens_at(n);                 // number of completions before the call's invocation
  { uint32_t r = 0; for (uint32_t i = 0; i < n; i++) { while (r < R && ret[by_ret[r]] < inv[i]) r++; opens_at[i] = r; } }
  std::vector<uint8_t> done(n, 0);
  std::vector<int64_t> open_by_slot(std::max(1u, n_slots), -1);   // live call open on each process slot at the front
  out.clear();
  out.reserve(n);
  uint32_t front = 0, next_inv = 0;
  auto open_calls = [&]() {
    while (next_inv < n && opens_at[next_inv] <= front) {
      if (ret[next_inv] != TBC_POS_CRASHED) open_by_slot[(uint32_t)proc[next_inv]] = next_inv;
      next_inv++;
    }
  };
  auto advance = [&]() -> bool {
    bool moved = false;
    while (front < R && done[by_ret[front]]) { open_by_slot[(uint32_t)proc[by_ret[front]]] = -1; front++; moved = true; open_calls(); }
    return moved;
  };
  std::vector<uint32_t> take;
  auto absorb = [&]() {
    for (bool again = true; again && front < R;) {
      take.clear();
      for (int64_t x : open_by_slot) if (x >= 0 && !done[x] && takes((uint32_t)x, state)) take.push_back((uint32_t)x);
      if (by_completion) std::sort(take.begin(), take.end(), [&](uint32_t x, uint32_t y) { return ret[x] < ret[y]; });
      for (uint32_t x : take) { done[x] = 1; out.push_back(x); }
      again = advance();
    }
  };
  open_calls();
  if (branch) absorb();
  for (uint32_t k = 0; k < chain_len; k++) {
    const uint32_t op = chain[k];
    if (op >= n || done[op]) return false;
    state = apply(op, state);
    done[op] = 1; out.push_back(op);
    advance();
    absorb();
  }
  return out.size() <= n;
}

// register / cas-register under the eager-read rule: what a pass absorbs are the open live reads whose value is nil or the state
inline bool expand_eager_chain(uint32_t n, const uint8_t* f, const int32_t* a, const int32_t* b, const int32_t* proc, const uint32_t* inv,
                               const uint32_t* ret, uint32_t n_slots, int32_t init, bool branch, bool by_completion,
                               const uint32_t* chain, uint32_t chain_len, std::vector<uint32_t>& out) {
  return expand_chain_with(n, proc, inv, ret, n_slots, init, branch, by_completion, chain, chain_len,
                           [&](uint32_t x, int32_t st) { return f[x] == TBC_F_READ && (a[x] == TBC_NIL || a[x] == st); },
                           [&](uint32_t op, int32_t st) { return f[op] == TBC_F_WRITE ? a[op] : (f[op] == TBC_F_CAS ? b[op] : st); }, out);
}

// multi-register under the eager-txn rule (tbc_internal.h kRuleTxnEager; oracle/wgl_beam.c absorb_txns): the state is 4 bits per key
// (0 = nil, v + 1), a call's value its micro-ops {f, key, value} at pool[a .. a + 3 b); what a pass absorbs are the open live txns of
// micro-reads only that the state allows.  The root is not normalised (the kernel's is not either).
inline bool expand_eager_txn_chain(uint32_t n, const uint8_t* f, const int32_t* a, const int32_t* b, const int32_t* proc, const uint32_t* inv,
                                   const uint32_t* ret, uint32_t n_slots, int32_t init, bool by_completion, const int32_t* pool,
                                   const uint32_t* chain, uint32_t chain_len, std::vector<uint32_t>& out) {
  for (uint32_t k = 0; k < chain_len; k++) if (chain[k] < n && f[chain[k]] != TBC_F_TXN) return false;
  return expand_chain_with(n, proc, inv, ret, n_slots, (uint32_t)init, false, by_completion, chain, chain_len,
                           [&](uint32_t x, uint32_t st) {
                             if (f[x] != TBC_F_TXN) return false;
                             for (int32_t i = 0; i < b[x]; i++) {
                               const int32_t mf = pool[a[x] + 3 * i], k = pool[a[x] + 3 * i + 1], v = pool[a[x] + 3 * i + 2];
                               if (mf != 0 || !(v == TBC_NIL || ((st >> (4 * k)) & 15u) == (uint32_t)(v + 1))) return false;
                             }
                             return true;
                           },
                           [&](uint32_t op, uint32_t st) {
                             for (int32_t i = 0; i < b[op]; i++) {
                               const int32_t mf = pool[a[op] + 3 * i], key = pool[a[op] + 3 * i + 1], v = pool[a[op] + 3 * i + 2];
                               if (mf != 0) st = (st & ~(15u << (4 * key))) | ((uint32_t)(v + 1) << (4 * key));
                             }
                             return st;
                           }, out);
}

}  // namespace tbc
