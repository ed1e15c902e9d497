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
  // knossos.model/step: may op (f,a,b) be applied in state st?
  __device__ __forceinline__ bool ok(int32_t st, uint32_t f, int32_t a, int32_t b) const {
    if (kind == TBC_MODEL_MUTEX) return (f == TBC_F_ACQUIRE && st == 0) || (f == TBC_F_RELEASE && st == 1);
    if (kind == TBC_MODEL_TABLE) return f == TBC_F_CLASS && table[(uint32_t)st * n_classes + (uint32_t)a] != TBC_TABLE_INCONSISTENT;
    if (kind == TBC_MODEL_MULTI_REGISTER) { int32_t s2; return f == TBC_F_TXN && txn(st, a, b, &s2); }
    // register / cas-register (pack rejected :cas for plain registers)
    return f == TBC_F_WRITE || (f == TBC_F_READ && (a == TBC_NIL || a == st)) || (f == TBC_F_CAS && a == st);
  }
  __device__ __forceinline__ int32_t apply(int32_t st, uint32_t f, int32_t a, int32_t b) const {
    if (kind == TBC_MODEL_MUTEX) return f == TBC_F_ACQUIRE ? 1 : 0;
    if (kind == TBC_MODEL_TABLE) return (int32_t)table[(uint32_t)st * n_classes + (uint32_t)a];
    if (kind == TBC_MODEL_MULTI_REGISTER) { int32_t s2 = st; (void)txn(st, a, b, &s2); return s2; }
    return f == TBC_F_WRITE ? a : (f == TBC_F_CAS ? b : st);
  }
};


}  // namespace tbc
