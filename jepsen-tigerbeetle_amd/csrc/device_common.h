// device_common.h -- helpers shared by the search kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include "tbc_internal.h"

namespace tbc {

__device__ __forceinline__ uint32_t rfl(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t rl(uint32_t v, uint32_t lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ uint64_t ru64(uint64_t v) {
  return (uint64_t)rfl((uint32_t)v) | ((uint64_t)rfl((uint32_t)(v >> 32)) << 32);
}

// min over the 64 lanes of a fully active wave; result is wave-uniform
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
  // all-reduce inside each row of 16 lanes with row rotations (gfx9 DPP)
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x128, 0xf, 0xf, false));  // row_ror:8
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x124, 0xf, 0xf, false));  // row_ror:4
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x122, 0xf, 0xf, false));  // row_ror:2
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x121, 0xf, 0xf, false));  // row_ror:1
  const uint32_t r0 = rl(v, 0), r1 = rl(v, 16), r2 = rl(v, 32), r3 = rl(v, 48);
  return min(min(r0, r1), min(r2, r3));
}

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xFF51AFD7ED558CCDull;
  x ^= x >> 33; x *= 0xC4CEB9FE1A85EC53ull;
  x ^= x >> 33;
  return x;
}

struct Model {
  uint32_t kind;
  const uint16_t* table;
  uint32_t n_classes;
  const int32_t* pool;     // wide op values (multi-register micro-ops: {f, key, value} triples)
  int32_t aux;             // commutative models: pool offset of the per-front table
  uint32_t n_keys;         // bank: number of accounts

  // set / bank: the state is a function of WHICH calls are linearized, so configs carry none
  __device__ __forceinline__ bool commutative() const { return kind == TBC_MODEL_SET || kind == TBC_MODEL_BANK; }

  // multi-register: :f :txn, value = [[f k v] ...], atomic.  State = 4 bits per key (0 = nil, v+1).
  __device__ __forceinline__ bool txn(int32_t st, int32_t a, int32_t b, int32_t* out) const {
    uint32_t s = (uint32_t)st;
    bool ok = true;
    for (int32_t i = 0; i < b; i++) {
      const int32_t mf = pool[a + 3 * i], k = pool[a + 3 * i + 1], v = pool[a + 3 * i + 2];
      const uint32_t cur = (s >> (4 * k)) & 15u;
      if (mf == 0) ok = ok && (v == TBC_NIL || cur == (uint32_t)(v + 1));
      else s = (s & ~(15u << (4 * k))) | ((uint32_t)(v + 1) << (4 * k));
    }
    *out = (int32_t)s;
    return ok;
  }
  // the keys a txn reads / writes, as bit sets (kRuleTxnIndep)
  __device__ __forceinline__ void txn_keys(int32_t a, int32_t b, uint32_t& r, uint32_t& w) const {
    r = 0u; w = 0u;
    for (int32_t i = 0; i < b; i++) {
      const int32_t mf = pool[a + 3 * i], k = pool[a + 3 * i + 1];
      if (mf == 0) r |= 1u << k; else w |= 1u << k;
    }
  }
  // a txn of micro-reads only, each nil or what the state holds for its key (kRuleTxnEager)
  __device__ __forceinline__ bool pure_read_ok(int32_t st, uint32_t f, int32_t a, int32_t b) const {
    bool ok = f == TBC_F_TXN;
    for (int32_t i = 0; ok && i < b; i++) {
      const int32_t mf = pool[a + 3 * i], k = pool[a + 3 * i + 1], v = pool[a + 3 * i + 2];
      ok = mf == 0 && (v == TBC_NIL || (((uint32_t)st >> (4 * k)) & 15u) == (uint32_t)(v + 1));
    }
    return ok;
  }
  // knossos.model/step: may op (f,a,b) be applied in state st?  REGF = the caller's kernel is the register-family
  // instantiation (register, cas-register, mutex): the table / multi-register paths are compiled out of it.
  // One formula serves register, cas-register and mutex: pack lets a history carry only its model's own ops
  // (:acquire / :release for a mutex, never :cas for a plain register), and f never matches another model's codes.
  template <bool REGF = false>
  __device__ __forceinline__ bool ok(int32_t st, uint32_t f, int32_t a, int32_t b) const {
    if constexpr (!REGF) {
      if (kind == TBC_MODEL_TABLE) return f == TBC_F_CLASS && table[(uint32_t)st * n_classes + (uint32_t)a] != TBC_TABLE_INCONSISTENT;
      if (kind == TBC_MODEL_MULTI_REGISTER) { int32_t s2; return f == TBC_F_TXN && txn(st, a, b, &s2); }
    }
    return f == TBC_F_WRITE || (f == TBC_F_READ && (a == TBC_NIL || a == st)) || (f == TBC_F_CAS && a == st) ||
           (f == TBC_F_ACQUIRE && st == 0) || (f == TBC_F_RELEASE && st == 1);
  }
  template <bool REGF = false>
  __device__ __forceinline__ int32_t apply(int32_t st, uint32_t f, int32_t a, int32_t b) const {
    if constexpr (!REGF) {
      if (kind == TBC_MODEL_TABLE) return (int32_t)table[(uint32_t)st * n_classes + (uint32_t)a];
      if (kind == TBC_MODEL_MULTI_REGISTER) { int32_t s2 = st; (void)txn(st, a, b, &s2); return s2; }
    }
    return f == TBC_F_WRITE ? a : (f == TBC_F_CAS ? b : (f == TBC_F_ACQUIRE ? 1 : (f == TBC_F_RELEASE ? 0 : st)));
  }
};


// May the open call `oi` be linearized next in the config (fi, Mp, st)?  State-based models ask
// Model::ok; the commutative ones (set, bank) look at the calls completed before the front (per-front
// tables in the pool) and at the open calls already linearized (the parent's open-call list).
template <int MW, bool COMM, bool REGF = false>
__device__ __forceinline__ bool pair_viable(const Model& model, int32_t st, uint32_t fi, const uint64_t (&Mp)[MW],
                                            uint32_t poff, uint32_t nlive, uint32_t cnt, const OpRec* lst,
                                            const OpRec* crashed, const OpRec& oi, bool lazy = false) {
  const uint32_t f = oi.f_slot & 0xFFu;
  if constexpr (!COMM) {
    return model.template ok<REGF>(st, f, oi.a, oi.b);
  } else {
  // the lazy rule (tbcheck.h, TBC_DOM_NO_LAZY_COMMUTING): a mutating call that does not complete at the front is a candidate only
  // while an open, not yet linearized read could take it -- set: one whose value holds the element; bank: one with a value
  if (lazy && (f == TBC_F_ADD || f == TBC_F_TRANSFER) && !(oi.f_slot & kAtFront)) {
    bool wanted = false;
    for (uint32_t cc = 0; !wanted && cc < nlive; cc++) {          // (reads are live calls: a crashed read has no value and is no candidate)
      const OpRec ox = lst[poff + cc];
      if ((ox.f_slot & 0xFFu) != TBC_F_READ || ox.a == TBC_NIL) continue;
      const uint32_t px = (ox.f_slot >> 8) & kSlotMask;
      bool lx = false;
#pragma unroll
      for (int j = 0; j < MW; j++) if ((px >> 6) == (uint32_t)j) lx = (Mp[j] >> (px & 63u)) & 1ull;
      if (lx) continue;
      if (f == TBC_F_TRANSFER) wanted = true;
      else {
        const int32_t* rp = model.pool + ox.a;
        const uint32_t jx = (uint32_t)oi.a;
        wanted = rp[0] >= 0 && ((((uint32_t)rp[2 + (jx >> 5)]) >> (jx & 31u)) & 1u);
      }
    }
    if (!wanted) return false;
  }
  if (model.kind == TBC_MODEL_SET) {
    // knossos.model/set, state-free: a read of R is consistent iff the adds completed before the
    // front plus the open adds already linearized are exactly R (pool layout: include/tbcheck.h)
    if (f != TBC_F_READ || oi.a == TBC_NIL) return f == TBC_F_ADD || f == TBC_F_READ;
    const int32_t* rp = model.pool + oi.a;
    const int32_t nR = rp[0], lead = rp[1];
    int32_t count = model.pool[model.aux + (int32_t)fi];
    bool viable = nR >= 0 && count <= lead;
    for (uint32_t cc = 0; viable && cc < cnt; cc++) {
      const OpRec ox = cc < nlive ? lst[poff + cc] : crashed[cc - nlive];
      const uint32_t px = (ox.f_slot >> 8) & kSlotMask;
      bool lx = false;
#pragma unroll
      for (int j = 0; j < MW; j++) if ((px >> 6) == (uint32_t)j) lx = (Mp[j] >> (px & 63u)) & 1ull;
      if (!lx || (ox.f_slot & 0xFFu) != TBC_F_ADD) continue;
      const uint32_t jx = (uint32_t)ox.a;
      if (!(((uint32_t)rp[2 + (jx >> 5)] >> (jx & 31u)) & 1u)) viable = false;
      count++;
    }
    return viable && count == nR;
  }
  // bank (negative balances allowed => transfers commute): balances = table of the transfers
  // completed before the front + the open transfers already linearized
  if (f != TBC_F_READ || oi.a == TBC_NIL) return f == TBC_F_TRANSFER || f == TBC_F_READ;
  int32_t bal[16];
  const uint32_t NA = model.n_keys;
  for (uint32_t a2 = 0; a2 < NA; a2++) bal[a2] = model.pool[model.aux + (int32_t)(fi * NA + a2)];
  for (uint32_t cc = 0; cc < cnt; cc++) {
    const OpRec ox = cc < nlive ? lst[poff + cc] : crashed[cc - nlive];
    const uint32_t px = (ox.f_slot >> 8) & kSlotMask;
    bool lx = false;
#pragma unroll
    for (int j = 0; j < MW; j++) if ((px >> 6) == (uint32_t)j) lx = (Mp[j] >> (px & 63u)) & 1ull;
    if (!lx || (ox.f_slot & 0xFFu) != TBC_F_TRANSFER) continue;
    const int32_t* tp = model.pool + ox.a;
    const int32_t amt = tp[2];
    for (uint32_t a2 = 0; a2 < NA; a2++) { if ((int32_t)a2 == tp[0]) bal[a2] -= amt; if ((int32_t)a2 == tp[1]) bal[a2] += amt; }
  }
  bool viable = true;
  for (uint32_t a2 = 0; a2 < NA; a2++) viable = viable && bal[a2] == model.pool[oi.a + (int32_t)a2];
  return viable;
  }
}

// The config reached by linearizing `oi` in (fi, Mp, st): set its process bit, step the model, and if it
// was the front's own call move the front past every completion already linearized (clearing their bits).
// slot_at(r) = process slot of the call completing at rank r (the caller decides how it is fetched).
template <int MW, bool COMM, bool REGF = false, class SlotAt>
__device__ __forceinline__ void make_child(const Model& model, bool viable, int32_t st, uint32_t fi, uint32_t R,
                                           SlotAt slot_at, const OpRec& oi,
                                           const uint64_t (&Mp)[MW], uint64_t (&M2)[MW], int32_t& st2, uint32_t& fi2) {
  const uint32_t f = oi.f_slot & 0xFFu, p = (oi.f_slot >> 8) & kSlotMask;
  st2 = st; fi2 = fi;
#pragma unroll
  for (int j = 0; j < MW; j++) M2[j] = Mp[j];
  if (!viable) return;
  if constexpr (COMM) st2 = 0; else st2 = model.template apply<REGF>(st, f, oi.a, oi.b);
#pragma unroll
  for (int j = 0; j < MW; j++) if ((p >> 6) == (uint32_t)j) M2[j] |= 1ull << (p & 63u);
  if (!(oi.f_slot & kAtFront)) return;
  uint32_t pp = p;
  for (;;) {
#pragma unroll
    for (int j = 0; j < MW; j++) if ((pp >> 6) == (uint32_t)j) M2[j] &= ~(1ull << (pp & 63u));
    fi2++;
    if (fi2 == R) break;
    pp = slot_at(fi2);
    bool bit = false;
#pragma unroll
    for (int j = 0; j < MW; j++) if ((pp >> 6) == (uint32_t)j) bit = (M2[j] >> (pp & 63u)) & 1ull;
    if (!bit) break;
  }
}

}  // namespace tbc
