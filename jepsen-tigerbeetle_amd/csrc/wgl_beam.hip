// wgl_beam.hip -- K5: the WIDE schedule of the Wing-Gong/Lowe search (gfx950).
//
// Same search space, same memoisation and same answers (verdict, failing op) as
// wgl_search.hip / knossos.wgl, but scheduled for 64 lanes instead of one
// thread: per iteration the wavefront takes the K most recent configs off an
// explicit stack and expands ALL their successors at once, ONE (config, open
// call) PAIR PER LANE:
//
//   pop K parents -> LDS                    (entries are read back from the table)
//   pairs = sum of the parents' open calls  (per-front lists built by pack_open.hip)
//   for each round of 64 pairs:
//     lane: open call -> model step -> child (front advance) -> visited-set probe
//     `__ballot` of "model-consistent" / "passed every completion" / "new config"
//     prefix-popcount of the "new" ballot compacts the successors onto the stack
//
// which is the shape BASELINE.json's north_star describes (ballot + prefix-sum
// compaction of linearizable successors into an exact open-addressed visited
// set).  Depth-first in spirit (the newest config's first successor ends up on
// top), so valid histories still finish without enumerating the whole config
// space, while a dead end is abandoned K configs at a time; the sequential
// schedule's heavy tail (one unlucky history taking 10^6 dependent steps)
// disappears.
//
// The schedule is deterministic and specified in oracle/wgl_beam.c, which
// tests/ compare bit-for-bit (verdict, failing op, witness, counters):
// lanes that produce one and the same new config in a round are found by their
// common table slot (equal keys probe in lockstep, so exactly one of them wins
// the CAS and the others lose it on that very slot) and the lowest lane keeps
// the config; everything else follows program order of one wavefront.
//
// Visited set (MW = mask words): keys k0 = front+1 | state<<32, M[MW] -- 16 B at
// MW = 1, four to a 64 B bucket that one probe reads whole -- and, in a separate
// array, {parent entry | op<<32}.  The 64 most recent pushes are mirrored in an
// LDS ring so the next iterations' parents come from LDS.  Every dependent trip to
// memory is a round's latency, and a round is the unit the whole search is made
// of, so the data is laid out for ONE trip per step: candidate = one 16 B record
// of the front's open-call list (pack_open.hip), front advance = a 16-rank window
// of completion slots fetched with it, probe = one bucket (+ the CAS for a new
// config).  A history is owned by one wavefront; entries are read/written with
// agent-scope (sc1) accesses so a lane never sees a stale L1 line of an entry
// another lane just claimed.
#include <hip/hip_runtime.h>
#include "tbc_internal.h"
#include "device_common.h"

namespace tbc {

namespace {

constexpr uint32_t kNone = 0xFFFFFFFFu;

// The visited set and the stack are addressed through pointers that pass through LDS (parked search
// state) or come out of the growth pool: name their address space, or every access turns into a flat_*.
typedef __attribute__((address_space(1))) uint64_t gu64;
typedef __attribute__((address_space(1))) uint32_t gu32;

__device__ __forceinline__ uint64_t ld64(const gu64* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st64(gu64* p, uint64_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t ld32(const gu32* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st32(gu32* p, uint32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// claim an empty entry: returns what was there (0 = claimed)
__device__ __forceinline__ uint64_t cas64_from_zero(gu64* p, uint64_t desired) {
  uint64_t expected = 0ull;
  __hip_atomic_compare_exchange_strong(p, &expected, desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return expected;
}

constexpr uint32_t kRing = 64;     // most recent pushes mirrored in LDS
constexpr uint32_t kCloByLane = 0xFFFFFFFFu;   // txn independence: no closure worked out for this parent (its key sets are 8 bits each)

// The lane number, opaque to the optimiser.  What an iteration derives from it (LDS addresses, lane
// masks, shuffle indices) is then recomputed per iteration instead of being hoisted out of the search
// loop and kept alive across all of it: a few VALU ops per iteration for a handful of VGPRs.
__device__ __forceinline__ uint32_t opaque_lane(uint32_t lane) {
  asm volatile("" : "+v"(lane));
  return lane;
}

// max over lanes 0..15 (row 0 of the wave) with DPP row rotations; valid in lanes 0..15
__device__ __forceinline__ uint32_t row_max_u32(uint32_t v) {
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x128, 0xf, 0xf, false));  // row_ror:8
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x124, 0xf, 0xf, false));  // row_ror:4
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x122, 0xf, 0xf, false));  // row_ror:2
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x121, 0xf, 0xf, false));  // row_ror:1
  return v;
}

// LDS words per wave: parents p_k0[16] p_M[16*MW] (u64), ring r_k0[kRing] r_M[kRing*MW] (u64),
// p_slot p_off p_nlive p_cnt (u32 x16), r_pos r_idx r_off r_nlive r_cnt (u32 x kRing), p_start (17 -> 20),
// search state (28, S_* words), this round's new configs for the lookahead c_M[64*MW] (u64) c_fi c_st (u32 x64)
// count form (CNT): + the count vectors of the parents and of the ring, p_C[16 * 2] r_C[kRing * 2] (u64)
__host__ __device__ constexpr uint32_t beam_lds_words(uint32_t mw, bool cnt = false) {
  return (16 + kRing) * (2 + 2 * mw) + 16 * 4 + kRing * 5 + 20 + 30 + 64 * (2 + 2 * mw) + (cnt ? (16 + kRing) * 2 * kCountWords : 0u);
}

__device__ __forceinline__ uint32_t key_hash32(uint64_t k0, const uint64_t* M, int mw) {
  uint32_t h = (uint32_t)k0 * 0x9E3779B1u ^ (uint32_t)(k0 >> 32) * 0x85EBCA77u;
  for (int j = 0; j < mw; j++) {
    h = (h << 13) | (h >> 19);
    h ^= (uint32_t)M[j] * 0xC2B2AE3Du ^ (uint32_t)(M[j] >> 32) * 0x27D4EB2Fu;
  }
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
  return h;
}

// ---- visited set -----------------------------------------------------------------------------------
// cap = 2^cap_log2 entries.  keys[cap]: (1 + MW) u64 each (k0, M[]), in BUCKETS of four consecutive
// entries -- 64 B, one cache line, at MW = 1; par[cap]: {parent entry | (op + 1) << 32}, behind the keys.
// A probe reads a whole bucket with four 16 B loads in flight (one trip to L2/HBM), looks for the key,
// else claims the bucket's first empty entry with a CAS on k0; a full bucket sends it to the next one.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// four agent-scope (sc1: never a stale line of this CU's L1) 16 B loads of one 64 B bucket, addressed as
// uniform base (SGPR pair) + 32-bit byte offset (one VGPR): the keys of one history stay below 4 GiB
__device__ __forceinline__ void ld_bucket16(const gu64* keys, uint32_t byte_off, u32x4& e0, u32x4& e1, u32x4& e2, u32x4& e3) {
  asm volatile(
      "global_load_dwordx4 %0, %4, %5 sc1\n\t"
      "global_load_dwordx4 %1, %4, %5 offset:16 sc1\n\t"
      "global_load_dwordx4 %2, %4, %5 offset:32 sc1\n\t"
      "global_load_dwordx4 %3, %4, %5 offset:48 sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(e0), "=&v"(e1), "=&v"(e2), "=&v"(e3)
      : "v"(byte_off), "s"(keys)
      : "memory");
}

// look at bucket `b`: returns match | empty << 4, bit t of `match` = entry t holds (k0, M), bit t of
// `empty` = entry t is free (k0 is never 0: it carries front + 1)
template <int MW>
__device__ __forceinline__ uint32_t scan_bucket(const gu64* tab, uint32_t b, uint64_t k0, const uint64_t (&M)[MW]) {
  constexpr uint32_t KW = MW + 1;
  const gu64* bp = tab + (uint64_t)b * (4 * KW);
  uint32_t match = 0, empty = 0;
  if constexpr (MW == 1) {
    u32x4 e0, e1, e2, e3;
    ld_bucket16(tab, b * 64u, e0, e1, e2, e3);
    const uint32_t k0l = (uint32_t)k0, k0h = (uint32_t)(k0 >> 32), ml = (uint32_t)M[0], mh = (uint32_t)(M[0] >> 32);
    empty = (e0.x == 0u ? 1u : 0u) | (e1.x == 0u ? 2u : 0u) | (e2.x == 0u ? 4u : 0u) | (e3.x == 0u ? 8u : 0u);
    match = ((e0.x == k0l && e0.y == k0h && e0.z == ml && e0.w == mh) ? 1u : 0u) |
            ((e1.x == k0l && e1.y == k0h && e1.z == ml && e1.w == mh) ? 2u : 0u) |
            ((e2.x == k0l && e2.y == k0h && e2.z == ml && e2.w == mh) ? 4u : 0u) |
            ((e3.x == k0l && e3.y == k0h && e3.z == ml && e3.w == mh) ? 8u : 0u);
  } else {
    uint64_t kk[4];
#pragma unroll
    for (int t = 0; t < 4; t++) kk[t] = ld64(bp + t * KW);
#pragma unroll
    for (int t = 0; t < 4; t++) {
      if ((uint32_t)kk[t] == 0u) empty |= 1u << t;
      else if (kk[t] == k0) {
        bool same = true;
#pragma unroll
        for (int j = 0; j < MW; j++) same = same && ld64(bp + t * KW + 1 + j) == M[j];
        match |= same ? 1u << t : 0u;
      }
    }
  }
  return match | empty << 4;
}

// ---- search state ------------------------------------------------------------------------------------
// Everything the round loop carries from one iteration to the next, parked in LDS whenever the loop is
// left (to grow the visited set, or for good).  Inside the loop the visited set's address and size are
// then loop-INVARIANT scalars; growing it in place made them (and, through the compiler's SGPR->VGPR
// demotion of the whole web, every counter next to them) per-lane values: 65 -> 107 VGPRs, 8 -> 4
// wavefronts per SIMD.  The accesses are volatile so nothing is forwarded around the growth code.
enum : uint32_t {
  S_TAB = 0, S_STACK = 2, S_CAP = 4, S_SP = 5, S_MAXSP = 6, S_MAXF = 7, S_K = 8, S_VERDICT = 9, S_CAUSE = 10,
  S_WINPAR = 11, S_WINOP = 12, S_WINSTATE = 13, S_PROBES = 14, S_VISITED = 16, S_EXPANDED = 18, S_ITER = 20,
  S_ROUNDS = 22, S_DSTACK = 24, S_DSP = 26, S_EXACT = 27, S_T0 = 28, S_WORDS = 30
};
typedef __attribute__((address_space(3))) volatile uint32_t* state_ptr;   // LDS, named so: volatile accesses keep the generic (flat) form otherwise
__device__ __forceinline__ uint32_t sld(state_ptr S, uint32_t i) { return rfl(S[i]); }
__device__ __forceinline__ uint64_t sld64(state_ptr S, uint32_t i) {
  return (uint64_t)rfl(S[i]) | ((uint64_t)rfl(S[i + 1]) << 32);
}
__device__ __forceinline__ void sst(state_ptr S, uint32_t i, uint32_t v) { S[i] = v; }
__device__ __forceinline__ void sst64(state_ptr S, uint32_t i, uint64_t v) { S[i] = (uint32_t)v; S[i + 1] = (uint32_t)(v >> 32); }

// Arguments only the cold paths read (limits, growth pool, results, debug words) are fetched from the kernarg
// segment where they are needed.  Read from the by-value struct they are loaded at kernel entry, stay live across
// the round loop and -- the loop needs every scalar register it can get -- are spilled to VGPR lanes and
// reloaded around it; the empty asm keeps the loads where they are written.
typedef const __attribute__((address_space(4))) BeamArgs* cold_args_ptr;
__device__ __forceinline__ cold_args_ptr cold_args() {
  cold_args_ptr p = (cold_args_ptr)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return p;
}

// count form: field-wise x >= y over packed count vectors; top = the top bit of every field (oracle/wgl_count.c, counts_ge)
__device__ __forceinline__ bool counts_ge(const uint64_t (&x)[kCountWords], const uint64_t (&y)[kCountWords], const uint64_t (&top)[kCountWords]) {
  bool ge = true;
#pragma unroll
  for (uint32_t w = 0; w < kCountWords; w++) {
    const uint64_t t = (x[w] | top[w]) - (y[w] & ~top[w]);
    ge = ge && ((((x[w] & ~y[w]) | (~(x[w] ^ y[w]) & t)) & top[w]) == top[w]);
  }
  return ge;
}
__device__ __forceinline__ uint64_t rl64(uint64_t v, uint32_t lane) {
  return (uint64_t)rl((uint32_t)v, lane) | ((uint64_t)rl((uint32_t)(v >> 32), lane) << 32);
}

// Cold path: move a history to a 4x larger visited set (and stack) taken from the batch's growth pool --
// re-insert every entry, then translate the slot numbers held by parent links and the stack.  Called
// between iterations (nothing popped); reads and updates the parked search state.
template <int MW, bool CNT = false>
__device__ __forceinline__ bool grow_visited_set(state_ptr S, uint32_t* r_pos, uint32_t lane) {
  constexpr uint32_t CWn = CNT ? kCountWords : 0u;      // count form: the count words ride behind the mask words (not hashed)
  constexpr uint32_t KW = MW + 1 + CWn, EW = KW + 1;
  const gu64* tab = (const gu64*)sld64(S, S_TAB);
  const gu32* stack = (const gu32*)sld64(S, S_STACK);
  const gu32* dstack = (const gu32*)sld64(S, S_DSTACK);          // configs set aside by the lookahead (may be null)
  const uint32_t cap_log2 = sld(S, S_CAP), sp = sld(S, S_SP), dsp = sld(S, S_DSP);
  const uint64_t old_cap = 1ull << cap_log2, new_cap = old_cap << 2;
  const uint64_t need = new_cap * EW + new_cap / 2 + (dstack ? new_cap / 2 : 0) + old_cap / 2;   // keys + parents, stack(s), slot translation
  unsigned long long base = 0;
  const cold_args_ptr C = cold_args();
  uint64_t* const pool = C->pool;
  if (!pool || cap_log2 + 2 > 31 || cap_log2 + 2 > C->max_tab_log2) return false;   // refused before any pool words are taken
  if (lane == 0) base = atomicAdd(C->pool_cursor, (unsigned long long)need);
  base = ru64(base);
  if (base + need > C->pool_words) return false;
  gu64* ntab = (gu64*)pool + base;
  gu64* npar = ntab + new_cap * KW;
  const gu64* opar = tab + old_cap * KW;
  gu32* nstack = (gu32*)(ntab + new_cap * EW);
  gu32* ndstack = dstack ? nstack + new_cap : nullptr;
  gu32* remap = nstack + new_cap + (dstack ? new_cap : 0);
  const uint32_t nbmask = (uint32_t)((new_cap >> 2) - 1);
#pragma unroll 1
  for (uint64_t s = lane; s < old_cap; s += 64) {
    const gu64* e = tab + s * KW;
    const uint64_t k0 = ld64(e);
    if ((uint32_t)k0 == 0u) continue;
    uint64_t Mx[MW + CWn];
#pragma unroll
    for (int j = 0; j < MW + (int)CWn; j++) Mx[j] = ld64(e + 1 + j);
    uint32_t b = key_hash32(k0, Mx, MW) & nbmask, idx = 0;
    for (bool placed = false; !placed; b = (b + 1u) & nbmask) {
#pragma unroll 1
      for (uint32_t t = 0; t < 4 && !placed; t++) {
        gu64* ne = ntab + ((uint64_t)b * 4 + t) * KW;
        if (cas64_from_zero(ne, k0) == 0ull) {
#pragma unroll
          for (int j = 0; j < MW + (int)CWn; j++) st64(ne + 1 + j, Mx[j]);
          idx = b * 4 + t; placed = true;
        }
      }
    }
    st32(remap + s, idx);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");      // (one wavefront owns the history: its stores are in L2 before it reads them back; an agent-scope fence would write back and drop the XCD's whole L2 -- wave_env.h wg_fence)
#pragma unroll 1
  for (uint64_t s = lane; s < old_cap; s += 64) {       // parent links -> new slot numbers
    if ((uint32_t)ld64(tab + s * KW) == 0u) continue;
    const uint64_t pw = ld64(opar + s);
    const uint64_t npw = (uint32_t)pw != kNone ? ((uint64_t)ld32(remap + (uint32_t)pw) | (pw & 0xFFFFFFFF00000000ull)) : pw;
    st64(npar + ld32(remap + s), npw);
  }
#pragma unroll 1
  for (uint32_t i = lane; i < sp; i += 64)
    st32(nstack + i, ld32(remap + ld32(stack + i)));
#pragma unroll 1
  for (uint32_t i = lane; i < dsp; i += 64)
    st32(ndstack + i, ld32(remap + ld32(dstack + i)));
  if (lane < kRing) r_pos[lane] = kNone;       // the ring held old slot numbers
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");      // (one wavefront owns the history: its stores are in L2 before it reads them back; an agent-scope fence would write back and drop the XCD's whole L2 -- wave_env.h wg_fence)
  if (lane == 0) {
    sst64(S, S_TAB, (uint64_t)ntab); sst64(S, S_STACK, (uint64_t)nstack); sst64(S, S_DSTACK, (uint64_t)ndstack);
    sst(S, S_CAP, cap_log2 + 2u);
  }
  return true;
}

template <int MW, bool COMM, bool REGF, bool CNT = false>
__device__ void beam_one(const BeamArgs& A, const uint32_t hidx, uint32_t* lds, const uint32_t lane) {
  static_assert(!CNT || (REGF && !COMM), "the count form is the register family's");
  constexpr uint32_t CWn = CNT ? kCountWords : 0u;   // count form (tbc_internal.h, kRuleCount): count words behind the mask words
  constexpr uint32_t KW = MW + 1 + CWn;   // u64 words per key

  const Hist* H = A.hist + hidx;
  const BeamHist* B = A.bh + hidx;
  const uint64_t op_off = ru64(H->op_off);
  const uint64_t off_off = ru64(B->off_off);
  const uint32_t* off = A.off + off_off;
  const uint32_t* ncr = A.ncr + off_off;
  const OpRec* lst = A.lst + ru64(B->lst_off);
  // (count form: the candidates past the live calls are the CLASSES of crashed calls, whose records head the history's block of cmem[])
  const OpRec* crashed = CNT ? reinterpret_cast<const OpRec*>(A.cmem + ru64(B->cmem_off)) : A.crashed + op_off;
  const uint8_t* slot8 = A.slot8 + slot8_off(op_off, hidx);
  const uint32_t R = rfl(H->n_ret), status = rfl(H->status) | rfl(B->status);
  // K parents per iteration, G = 64 / K lanes (candidate slots) per parent per round.  A history that has
  // used more than round_budget rounds continues at K = 16: stragglers then need far fewer dependent rounds.
  // the register family (register, cas-register, mutex) steps on immediates: its kernel carries no table / pool pointers
  const Model model = REGF ? Model{A.model_kind, nullptr, 0u, nullptr, 0, 0u}
                           : Model{A.model_kind, A.table, A.n_classes, A.pool_vals, (int32_t)rfl((uint32_t)H->aux), A.n_keys};

  uint64_t* p_k0 = reinterpret_cast<uint64_t*>(lds);
  uint64_t* p_M = p_k0 + 16;
  uint64_t* r_k0 = p_M + 16 * MW;
  uint64_t* r_M = r_k0 + kRing;
  uint32_t* p_slot = reinterpret_cast<uint32_t*>(r_M + kRing * MW);
  uint32_t* p_off = p_slot + 16;
  uint32_t* p_nlive = p_off + 16;
  uint32_t* p_cnt = p_nlive + 16;
  uint32_t* r_pos = p_cnt + 16;      // stack position mirrored in this ring slot (kNone = empty)
  uint32_t* r_idx = r_pos + kRing;
  uint32_t* r_off = r_idx + kRing;
  uint32_t* r_nlive = r_off + kRing;
  uint32_t* r_cnt = r_nlive + kRing;
  uint32_t* p_start = r_cnt + kRing;  // 17 entries: pair-number prefix, general mapping only
  state_ptr S = (state_ptr)(p_start + 20);   // parked search state (S_* words)
  uint64_t* c_M = reinterpret_cast<uint64_t*>(p_start + 20 + S_WORDS);   // lookahead: the round's new configs
  uint32_t* c_fi = reinterpret_cast<uint32_t*>(c_M + 64 * MW);
  uint32_t* c_st = c_fi + 64;
  uint64_t* p_C = reinterpret_cast<uint64_t*>(c_st + 64);                // count form: the parents' / the ring's count vectors
  uint64_t* r_C = p_C + 16 * kCountWords;
  // count form: the classes' members, the top bit of every count field, the prefix target, exact / relaxed
  const uint64_t* cmem = CNT ? A.cmem + ru64(B->cmem_off) : nullptr;
  uint64_t top[kCountWords] = {0ull, 0ull};
  uint32_t RT = R;
  if constexpr (CNT) {
    top[0] = ru64(B->top[0]); top[1] = ru64(B->top[1]);
    const uint32_t tg = rfl(B->target);
    if (tg != 0u && tg < R) RT = tg;
  }
  const bool relaxed = CNT && A.count_mode == kCountRelaxed;
  const uint64_t* look = A.look ? A.look + look_off(op_off, hidx, MW) : nullptr;
  // dominance rules (tbc_internal.h): open-read masks per (front, value), twin masks per list entry
  const uint32_t rules = COMM ? 0u : A.rules, vpad = A.vpad;
  const uint64_t* rdm = A.rdm + op_off * vpad * MW;
  const uint64_t* twn = A.twn + ru64(B->lst_off) * MW;
  if (lane < kRing) r_pos[lane] = kNone;

#ifdef TBC_SEGPROF   // per-segment cycle counters (scripts/gpu_segprof.py); costs ~20 VGPRs, off in production
  const bool prof = A.dbg != nullptr;
  uint64_t seg[6] = {0, 0, 0, 0, 0, 0}, tlast = prof ? __builtin_readcyclecounter() : 0;
#define SEG(i) do { if (prof) { const uint64_t tn_ = __builtin_readcyclecounter(); seg[i] += tn_ - tlast; tlast = tn_; } } while (0)
#else
  constexpr bool prof = false;
#define SEG(i) do {} while (0)
#endif

  {   // initial state -> LDS
    gu32* stack0 = (gu32*)A.stack + ru64(B->stack_off);
    gu64* tab0 = (gu64*)A.tab + ru64(B->tab_off) * (KW + 1);
    const uint32_t cap0 = rfl(B->tab_log2);
    int32_t verdict0 = -2;
    uint32_t sp0 = 0;
    const bool resuming = A.resume != 0u && A.park != nullptr && status == 0 && R != 0;
    if (status != 0) verdict0 = TBC_UNKNOWN;
    else if (R == 0) verdict0 = TBC_VALID;
    else if (!resuming) {
      // root config: first entry of its bucket
      const uint64_t k0 = 1ull | ((uint64_t)(uint32_t)A.init_state << 32);
      uint64_t zero[MW];
#pragma unroll
      for (int j = 0; j < MW; j++) zero[j] = 0;
      const uint32_t idx = (key_hash32(k0, zero, MW) & (uint32_t)((1ull << (cap0 - 2)) - 1ull)) * 4u;
      if (lane == 0) {
        gu64* e = tab0 + (uint64_t)idx * KW;
        st64(e + 0, k0);
#pragma unroll
        for (int j = 0; j < MW + (int)CWn; j++) st64(e + 1 + j, 0ull);
        st64(tab0 + ((uint64_t)KW << cap0) + idx, (uint64_t)kNone | ((uint64_t)kNone << 32));
        st32(stack0, idx);
      }
      sp0 = 1;
    }
    if (resuming) {
      // taken up where a budgeted pass left it: every word of the parked state, the verdict open again, the clock started again
      if (lane < S_WORDS) S[lane] = A.park[(uint64_t)hidx * kParkWords + lane];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (lane == 0) {
        sst(S, S_VERDICT, (uint32_t)-2); sst(S, S_CAUSE, (uint32_t)TBC_CAUSE_NONE);
        sst64(S, S_T0, A.time_limit_ticks ? (uint64_t)wall_clock64() : 0ull);
      }
    } else
    if (lane == 0) {
      sst64(S, S_TAB, (uint64_t)tab0); sst64(S, S_STACK, (uint64_t)stack0);
      sst64(S, S_DSTACK, (look && A.dstack) ? (uint64_t)((gu32*)A.dstack + ru64(B->stack_off)) : 0ull);
      sst(S, S_DSP, 0u); sst(S, S_EXACT, (look && A.dstack) ? 0u : 1u);
      sst(S, S_CAP, cap0); sst(S, S_SP, sp0); sst(S, S_MAXSP, sp0); sst(S, S_MAXF, 0u); sst(S, S_K, A.width);
      sst(S, S_VERDICT, (uint32_t)verdict0); sst(S, S_CAUSE, (uint32_t)TBC_CAUSE_NONE);
      sst(S, S_WINPAR, kNone); sst(S, S_WINOP, kNone); sst(S, S_WINSTATE, (uint32_t)A.init_state);
      sst64(S, S_PROBES, 0ull); sst64(S, S_VISITED, (uint64_t)sp0); sst64(S, S_EXPANDED, 0ull);
      sst64(S, S_ITER, 0ull); sst64(S, S_ROUNDS, 0ull);
      sst64(S, S_T0, A.time_limit_ticks ? (uint64_t)wall_clock64() : 0ull);
    }
  }

  for (;;) {   // search until done; leave the round loop (state parked) whenever the visited set must grow
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  gu64* const tab = (gu64*)sld64(S, S_TAB);
  gu32* const stack = (gu32*)sld64(S, S_STACK);
  const uint32_t cap_log2 = sld(S, S_CAP);
  const uint32_t bmask = (uint32_t)((1ull << (cap_log2 - 2)) - 1ull);       // bucket index mask
  const uint32_t full_at = (uint32_t)((1ull << cap_log2) - (1ull << (cap_log2 - 2)));
  gu64* const par = tab + ((uint64_t)KW << cap_log2);
  uint32_t K = sld(S, S_K), gshift = 6u - (31u - (uint32_t)__builtin_clz(K)), G = 1u << gshift;
  // counters: 32-bit inside the loop (what happened since the state was last parked), 64-bit totals in LDS.  The
  // visited-set fill is an absolute count (it never exceeds the capacity, < 2^28).
  uint32_t probes = 0, expanded = 0, iterations = 0, rounds = 0;
  uint32_t visited = (uint32_t)sld64(S, S_VISITED);
  const uint64_t probes0 = sld64(S, S_PROBES), rounds0 = sld64(S, S_ROUNDS);
  bool need_park = false;
  // how far the 32-bit deltas may go before the loop is left to look at the 64-bit totals: the step limit (or the
  // fold at 2^31), and the round budget after which a straggler continues at K = 16
  uint32_t probe_room = 0x7FFFFFFFu, round_room = 0xFFFFFFFFu;
  {
    const cold_args_ptr C = cold_args();
    const uint64_t ms = C->max_steps, rb = C->round_budget;
    if (ms && ms - min(ms, probes0) < (uint64_t)probe_room) probe_room = (uint32_t)(ms - min(ms, probes0));
    if (rb) round_room = (uint32_t)min(rb - min(rb, rounds0), (uint64_t)0xFFFFFFFEu);
  }
  uint32_t sp = sld(S, S_SP), max_sp = sld(S, S_MAXSP), lane_maxf = sld(S, S_MAXF);
  int32_t verdict = (int32_t)sld(S, S_VERDICT), cause = (int32_t)sld(S, S_CAUSE);
  gu32* const dstack = (gu32*)sld64(S, S_DSTACK);
  uint32_t dsp = sld(S, S_DSP);
  const bool look_on = sld(S, S_EXACT) == 0u;     // false once the set-aside configs are being expanded
  bool need_grow = false, need_switch = false;

  while (verdict == -2) {
    // step limit (as after the ITERATION that exceeded it: every popped config is expanded completely, so the parked state is one the
    // search can be taken up from -- BeamArgs.resume; oracle/wgl_beam.c counts the same way for K > 1), or 2^31 probes to fold into the 64-bit totals
    if (__builtin_expect(probes > probe_room, 0)) { need_park = true; break; }
    if (sp == 0) {
      // no linearization through the live configs.  Those the lookahead set aside become the stack and the
      // search goes on without lookahead: an INVALID verdict has then expanded every reachable config
      // exactly once -- failing op, :configs and visited / probes / expanded are the plain search's
      if (dsp != 0u) { need_switch = true; break; }
      verdict = TBC_INVALID; break;
    }
    if (__builtin_expect(rounds > round_room, 0) && K < 16u) { K = 16u; gshift = 2u; G = 4u; }
    const uint32_t np = min(K, sp);
    const uint32_t ln = opaque_lane(lane);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- pop the np most recent configs: lane l < np loads the l-th from the bottom of the popped run
    uint32_t my_cnt = 0;
    if (ln < np) {
      const uint32_t spos = sp - np + ln, rs = spos & (kRing - 1u);
      if (r_pos[rs] == spos) {                  // pushed recently: config (and its front's list) still in the ring
        p_k0[ln] = r_k0[rs];
#pragma unroll
        for (int j = 0; j < MW; j++) p_M[ln * MW + j] = r_M[rs * MW + j];
        if constexpr (CNT) { p_C[ln * 2] = r_C[rs * 2]; p_C[ln * 2 + 1] = r_C[rs * 2 + 1]; }
        p_slot[ln] = r_idx[rs]; p_off[ln] = r_off[rs]; p_nlive[ln] = r_nlive[rs];
        my_cnt = r_cnt[rs];
      } else {
        const uint32_t idx = ld32(stack + spos);
        const gu64* e = tab + (uint64_t)idx * KW;
        const uint64_t k0 = ld64(e);
        p_k0[ln] = k0;
#pragma unroll
        for (int j = 0; j < MW; j++) p_M[ln * MW + j] = ld64(e + 1 + j);
        if constexpr (CNT) { p_C[ln * 2] = ld64(e + 1 + MW); p_C[ln * 2 + 1] = ld64(e + 2 + MW); }
        const uint32_t fi = (uint32_t)k0 - 1u;
        const uint32_t o0 = off[fi], o1 = off[fi + 1], nc = ncr[fi];
        p_slot[ln] = idx; p_off[ln] = o0; p_nlive[ln] = o1 - o0;
        my_cnt = (o1 - o0) + nc;
      }
      p_cnt[ln] = my_cnt;
    }
    sp -= np;
    const uint32_t maxcnt = rl(row_max_u32(my_cnt), 0);   // lanes >= np hold 0
    // Pair order (oracle/wgl_beam.c): parents bottom-first, each parent's open calls last-to-first,
    // 64 consecutive pairs per round.  When every parent has at most G = 64/K open calls all pairs
    // fit one round and lane = parent*G + i realises that order directly; otherwise pairs are
    // numbered through a prefix sum over the parents (crash-heavy histories).
    const bool grouped = maxcnt <= G;
    uint32_t T = maxcnt ? 64u : 0u;
    if (!grouped) {
      uint32_t x = my_cnt;
#pragma unroll
      for (int d = 1; d < 16; d <<= 1) {
        const uint32_t y = __shfl_up(x, d);
        if (ln >= (uint32_t)d) x += y;
      }
      if (ln < np) p_start[ln] = x - my_cnt;
      T = rl(x, np - 1);
      if (ln >= np && ln <= 16u) p_start[ln] = ln == np ? T : 0xFFFFFFFFu;     // no pair number reaches these
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- room for every pair of this iteration?  If not, put the popped configs back and leave: the
    // visited set moves to a 4x larger one, and the iteration is taken again from its start.
    if (__builtin_expect(visited + (grouped ? np * G : T) > full_at, 0)) { sp += np; need_grow = true; break; }
    const uint32_t lr = opaque_lane(lane);
    iterations++; expanded += np;
    if constexpr (!COMM && !REGF) {
      // txn independence (multi-register, tbc_internal.h kRuleTxnIndep): a parent's closure -- the keys read / written by the calls that
      // conflict, directly or through others, with the call completing at its front -- is worked out ONCE per parent, by the whole
      // wavefront: lane = open call (one trip for the records, one for the micro-ops), the fixed point over two key sets by ballots.
      // (The words are the lookahead's, which the register family alone uses.)
      if (rules & kRuleTxnIndep) {
        for (uint32_t qq = 0; qq < np; qq++) {
          const uint32_t cq = p_cnt[qq], nlq = p_nlive[qq], poq = p_off[qq];
          uint32_t clo = kCloByLane;                        // more than 64 open calls: every lane walks the list itself (below)
          if (cq <= 64u) {
            uint32_t yr = 0u, yw = 0u;
            bool open = false, atf = false;
            if (lane < cq) {
              const OpRec y = lane < nlq ? lst[poq + lane] : crashed[lane - nlq];
              const uint32_t py = (y.f_slot >> 8) & kSlotMask;
              bool ly = false;
#pragma unroll
              for (int j = 0; j < MW; j++) if ((py >> 6) == (uint32_t)j) ly = (p_M[qq * MW + j] >> (py & 63u)) & 1ull;
              open = !ly; atf = lane < nlq && (y.f_slot & kAtFront) != 0u;
              model.txn_keys(y.a, y.b, yr, yw);
            }
            uint32_t cr = 0u, cw = 0u;
            const uint64_t bx = __ballot(atf);               // the call completing at the front starts the closure
            if (bx) { const uint32_t xl = (uint32_t)__builtin_ctzll(bx); cr = rl(yr, xl); cw = rl(yw, xl); }
            for (bool more = true; more;) {
              const bool hit = open && (((yw & (cr | cw)) | (yr & cw)) != 0u);
              uint32_t nr = 0u, nw = 0u;
#pragma unroll
              for (uint32_t kb = 0; kb < 8u; kb++) {
                if (__ballot(hit && ((yr >> kb) & 1u))) nr |= 1u << kb;
                if (__ballot(hit && ((yw >> kb) & 1u))) nw |= 1u << kb;
              }
              more = ((nr & ~cr) | (nw & ~cw)) != 0u;
              cr |= nr; cw |= nw;
            }
            clo = cr | (cw << 16);
          }
          if (lane == 0) c_fi[qq] = clo;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
    SEG(0);

    for (uint32_t base = 0; base < T && verdict == -2; base += 64) {
      // ---- which (parent, open call) pair this lane handles
      uint32_t q, cd;
      bool has_parent;
      if (grouped) {
        q = lr >> gshift; cd = lr & (G - 1u); has_parent = q < np;
      } else {
        const uint32_t r = base + lr;
        q = 0;
#pragma unroll
        for (uint32_t t = 1; t < 16; t++) q += p_start[t] <= r ? 1u : 0u;
        has_parent = r < T;
        q = has_parent ? q : 0u;
        cd = has_parent ? r - p_start[q] : 0u;
      }
      const uint64_t k0p = has_parent ? p_k0[q] : 1ull;
      const uint32_t fi = (uint32_t)k0p - 1u;
      // count form: bit 30 of the state word says the config is hot (reached by a crashed call nobody has observed yet)
      const bool hot = CNT && ((uint32_t)(k0p >> 32) & kHotBit) != 0u;
      const int32_t st = CNT ? (int32_t)((uint32_t)(k0p >> 32) & ~kHotBit) : (int32_t)(uint32_t)(k0p >> 32);
      uint64_t Cp[kCountWords] = {0ull, 0ull};
      if constexpr (CNT) { Cp[0] = has_parent ? p_C[q * 2] : 0ull; Cp[1] = has_parent ? p_C[q * 2 + 1] : 0ull; }
      uint64_t Mp[MW];
#pragma unroll
      for (int j = 0; j < MW; j++) Mp[j] = has_parent ? p_M[q * MW + j] : 0ull;
      const uint32_t pslot = has_parent ? p_slot[q] : 0u, poff = has_parent ? p_off[q] : 0u;
      const uint32_t nlive = has_parent ? p_nlive[q] : 0u, cnt = has_parent ? p_cnt[q] : 0u;
      const bool act = has_parent && cd < cnt;
      const uint32_t c = cnt - 1u - cd;
      // one trip: the candidate's record, and the completion slots of the next 9..16 ranks (front advance)
      OpRec oi; oi.op = 0; oi.f_slot = kFNone; oi.a = 0; oi.b = 0;
      const uint32_t wbase = (fi + 1u) & ~7u;
      uint64_t w0 = 0, w1 = 0;
      if (act) {
        oi = c < nlive ? lst[poff + c] : crashed[c - nlive];
        const uint64_t* wp = reinterpret_cast<const uint64_t*>(slot8 + wbase);
        w0 = wp[0]; w1 = wp[1];
      }
      // twin rule: the slots of the open calls with this call's effect that complete earlier travel with its
      // list entry (same trip); dominated while one of them is not linearized yet
      bool dominated = false;
      if constexpr (!COMM) {
        if ((rules & kRuleTwin) && act && c < nlive) {
#pragma unroll
          for (int j = 0; j < MW; j++) dominated = dominated || (twn[(uint64_t)(poff + c) * MW + j] & ~Mp[j]) != 0ull;
        }
      }
      // count form: a candidate past the live calls is a CLASS of crashed calls; its next member (the count says which) must be invoked
      const bool is_cls = CNT && act && c >= nlive;
      const uint32_t cls_shift = (oi.f_slot >> 8) & 0xFFu, cls_width = (oi.f_slot >> 16) & 0xFFu;
      uint64_t mem = ~0ull;
      if constexpr (CNT) {
        if (is_cls) {
          const uint32_t kc = relaxed ? 0u : (uint32_t)(((cls_shift & 64u) ? Cp[1] : Cp[0]) >> (cls_shift & 63u)) & ((1u << cls_width) - 1u);
          mem = cmem[oi.op + kc];                 // inv_rank | op << 32; the sentinel behind the last member is all ones
        }
      }
      const uint32_t op = is_cls ? (uint32_t)(mem >> 32) : oi.op;
      const uint32_t p = is_cls ? 0u : (oi.f_slot >> 8) & kSlotMask;
      if (prof) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); SEG(1); }
      bool lin = false;
#pragma unroll
      for (int j = 0; j < MW; j++) if (!is_cls && (p >> 6) == (uint32_t)j) lin = (Mp[j] >> (p & 63u)) & 1ull;
      if constexpr (!COMM) {
        // a crashed call has no per-front entry: its twins are every live open call with its effect (they all
        // complete earlier) and the crashed ones invoked before it -- walk the list (crash-heavy histories only)
        const uint32_t cf = oi.f_slot & 0xFFu;
        if (!CNT && (rules & kRuleTwin) && act && !lin && c >= nlive && (cf == TBC_F_WRITE || cf == TBC_F_CAS)) {
          for (uint32_t cc = 0; cc < c && !dominated; cc++) {
            const OpRec y = cc < nlive ? lst[poff + cc] : crashed[cc - nlive];
            if ((y.f_slot & 0xFFu) != cf || y.a != oi.a || (cf == TBC_F_CAS && y.b != oi.b)) continue;
            const uint32_t py = (y.f_slot >> 8) & kSlotMask;
            bool ly = false;
#pragma unroll
            for (int j = 0; j < MW; j++) if ((py >> 6) == (uint32_t)j) ly = (Mp[j] >> (py & 63u)) & 1ull;
            dominated = !ly;
          }
        }
      }
      if constexpr (!COMM && !REGF) {
        // txn independence: a candidate outside its parent's closure (worked out above, once per parent) is not tried
        if ((rules & kRuleTxnIndep) && act && !lin && !(oi.f_slot & kAtFront)) {
          const uint32_t clo = c_fi[q];
          uint32_t cr = clo & 0xFFFFu, cw = clo >> 16, yr, yw;
          if (clo == kCloByLane) {
            cr = 0u; cw = 0u;
            for (uint32_t cc = 0; cc < nlive; cc++) {
              const OpRec y = lst[poff + cc];
              if (y.f_slot & kAtFront) { model.txn_keys(y.a, y.b, cr, cw); break; }
            }
            for (bool grew = true; grew;) {
              grew = false;
              for (uint32_t cc = 0; cc < cnt; cc++) {
                const OpRec y = cc < nlive ? lst[poff + cc] : crashed[cc - nlive];
                const uint32_t py = (y.f_slot >> 8) & kSlotMask;
                bool ly = false;
#pragma unroll
                for (int j = 0; j < MW; j++) if ((py >> 6) == (uint32_t)j) ly = (Mp[j] >> (py & 63u)) & 1ull;
                if (ly) continue;
                model.txn_keys(y.a, y.b, yr, yw);
                if (!((yw & (cr | cw)) | (yr & cw))) continue;
                if ((yr & ~cr) | (yw & ~cw)) { cr |= yr; cw |= yw; grew = true; }
              }
            }
          }
          model.txn_keys(oi.a, oi.b, yr, yw);
          dominated = !((yw & (cr | cw)) | (yr & cw));
        }
      }
      bool viable = act && !lin && !dominated && !is_cls && pair_viable<MW, COMM, REGF>(model, st, fi, Mp, poff, nlive, cnt, lst, crashed, oi, COMM && (A.rules & kRuleLazyComm) != 0u);
      if constexpr (CNT) {
        const uint32_t cf = oi.f_slot & 0xFFu;
        // a hot config only takes calls whose precondition is its state; a crashed :write is not one, and never writes the state it finds
        if (hot) viable = viable && (cf == TBC_F_READ || cf == TBC_F_CAS) && oi.a == st;
        if (is_cls) viable = (uint32_t)mem <= fi && (cf == TBC_F_WRITE ? (!hot && oi.a != st) : oi.a == st);
      }
      int32_t st2 = st;
      uint32_t fi2 = fi;
      uint64_t M2[MW];
      uint64_t C2[kCountWords] = {Cp[0], Cp[1]};
      const auto slot_at = [=](uint32_t r) -> uint32_t {
        const uint32_t d = r - wbase;
        if (d < 8u) return (uint32_t)(w0 >> (8u * d)) & 0xFFu;
        if (d < 16u) return (uint32_t)(w1 >> (8u * (d - 8u))) & 0xFFu;
        return (uint32_t)slot8[r];
      };
      make_child<MW, COMM, REGF>(model, viable && !is_cls, st, fi, R, slot_at, oi, Mp, M2, st2, fi2);
      bool observed = true;
      if constexpr (CNT) {
        if (viable && is_cls) {                     // a crashed call takes effect: no mask bit, the front stays, its class's count goes up
          st2 = (oi.f_slot & 0xFFu) == TBC_F_WRITE ? oi.a : oi.b;
          const uint64_t inc = relaxed ? 0ull : 1ull << (cls_shift & 63u);
          if (cls_shift & 64u) C2[1] += inc; else C2[0] += inc;
          observed = false;
        }
      }
      if constexpr (!COMM) {
        // eager reads: the child takes every open read its state allows (value nil or the state), the front moves
        // past the completions that linearizes, and the calls open at the new front are looked at again
        if ((rules & kRuleEager) && viable && fi2 < R) {
          for (;;) {
            const uint64_t* row = rdm + (uint64_t)fi2 * vpad * MW;
            const uint32_t vi = rdm_index(st2, vpad);
            if constexpr (CNT) {                    // a read of exactly the produced value, absorbed now, observes the crashed call
#pragma unroll
              for (int j = 0; j < MW; j++) observed = observed || (row[vi * MW + j] & ~M2[j]) != 0ull;
            }
#pragma unroll
            for (int j = 0; j < MW; j++) M2[j] |= row[j] | row[vi * MW + j];
            uint32_t pp = slot_at(fi2);
            bool bit = false;
#pragma unroll
            for (int j = 0; j < MW; j++) if ((pp >> 6) == (uint32_t)j) bit = (M2[j] >> (pp & 63u)) & 1ull;
            if (!bit) break;
            do {
#pragma unroll
              for (int j = 0; j < MW; j++) if ((pp >> 6) == (uint32_t)j) M2[j] &= ~(1ull << (pp & 63u));
              fi2++;
              if (fi2 == R) break;
              pp = slot_at(fi2);
              bit = false;
#pragma unroll
              for (int j = 0; j < MW; j++) if ((pp >> 6) == (uint32_t)j) bit = (M2[j] >> (pp & 63u)) & 1ull;
            } while (bit);
            if (fi2 == R) break;
          }
        }
      }
      if constexpr (!COMM && !REGF) {
        // eager txns (multi-register, kRuleTxnEager): the child takes every open pure-read txn its state allows, the front moves past the
        // completions that linearizes, and the calls open at the new front are looked at again
        if ((rules & kRuleTxnEager) && viable && fi2 < R) {
          for (bool again = true; again && fi2 < R;) {
            again = false;
            const uint32_t e1 = off[fi2 + 1u];
            for (uint32_t e = off[fi2]; e < e1; e++) {
              const OpRec y = lst[e];
              const uint32_t py = (y.f_slot >> 8) & kSlotMask;
              bool ly = false;
#pragma unroll
              for (int j = 0; j < MW; j++) if ((py >> 6) == (uint32_t)j) ly = (M2[j] >> (py & 63u)) & 1ull;
              if (ly || !model.pure_read_ok(st2, y.f_slot & 0xFFu, y.a, y.b)) continue;
#pragma unroll
              for (int j = 0; j < MW; j++) if ((py >> 6) == (uint32_t)j) M2[j] |= 1ull << (py & 63u);
            }
            for (;;) {
              const uint32_t pp = slot_at(fi2);
              bool bit = false;
#pragma unroll
              for (int j = 0; j < MW; j++) if ((pp >> 6) == (uint32_t)j) bit = (M2[j] >> (pp & 63u)) & 1ull;
              if (!bit) break;
#pragma unroll
              for (int j = 0; j < MW; j++) if ((pp >> 6) == (uint32_t)j) M2[j] &= ~(1ull << (pp & 63u));
              fi2++; again = true;
              if (fi2 == R) break;
            }
          }
        }
      }
      SEG(2);
      rounds++;
      const uint64_t succ = __ballot(viable && fi2 >= RT);      // (RT = R unless the count form checks a prefix)
      if (succ) {   // linearizable: lowest pair wins, nothing of this round is inserted
        const uint32_t wl = (uint32_t)__builtin_ctzll(succ);
        if (lane == 0) { sst(S, S_WINPAR, rl(pslot, wl)); sst(S, S_WINOP, rl(op, wl)); sst(S, S_WINSTATE, rl((uint32_t)st2, wl)); }
        if constexpr (CNT) {
          // a PREFIX search (count-form pipeline, tbc_api: the relaxed pass refuted completion RT) that ends VALID: the config it ended in
          // stands in front of the completion nobody passes -- record 0 of the history's :configs arena (round 6; the verdict's :configs
          // used to be empty)
          if (RT < R && lane == wl) {
            const cold_args_ptr Cw = cold_args();
            if (Cw->cfg) {
              uint64_t* o = Cw->cfg + (uint64_t)hidx * kCfgCap * (2 + MW);
              o[0] = (uint64_t)(fi2 + 1u) | ((uint64_t)(uint32_t)st2 << 32);
#pragma unroll
              for (int j = 0; j < MW; j++) o[1 + j] = M2[j];
              o[1 + MW] = (uint64_t)op;
            }
          }
        }
        verdict = TBC_VALID;
        break;
      }
      const uint64_t vb = __ballot(viable);
      probes += (uint32_t)__popcll(vb);

      // the child's front: its open-call list (issued now, consumed at push; hidden under the probe)
      uint32_t co0 = 0, co1 = 0, cnc = 0;
      if (viable) { co0 = off[fi2]; co1 = off[fi2 + 1u]; cnc = ncr[fi2]; }

      // ---- visited set: find the key in its bucket chain, else claim the first empty entry met.
      // Equal keys probe in lockstep: they look at the same bucket, go for the same entry, one wins the
      // CAS, and the others find the key there on their next look (`lost` then tells it is new).
      const uint64_t k0 = (uint64_t)(fi2 + 1u) | ((uint64_t)((uint32_t)st2 | ((CNT && !observed) ? kHotBit : 0u)) << 32);
      uint32_t b = key_hash32(k0, M2, MW) & bmask, idx = 0, full_buckets = 0;
      bool is_new = false;
      if constexpr (CNT) {
        // ---- count form: the Pareto rule.  A child is dropped when a visited config with its key (k0, M) has used no more of any
        // class; the pairs of a round count in pair order (oracle/wgl_count.c).  Three steps: (1) every lane walks its key's probe
        // chain -- buckets from the hashed one up to the first with an empty entry -- and compares count vectors; (2) the lanes that
        // survived are compared with each other, lowest lane first; (3) the survivors claim an empty entry each.
        bool pending = viable, cand = false;
        while (__ballot(pending)) {
          if (pending) {
            const gu64* bp = tab + (uint64_t)b * (4 * KW);
            uint64_t kk[4];
#pragma unroll
            for (int t = 0; t < 4; t++) kk[t] = ld64(bp + t * KW);
            bool dom = false, empty = false;
#pragma unroll
            for (int t = 0; t < 4; t++) {
              if ((uint32_t)kk[t] == 0u) empty = true;
              else if (kk[t] == k0) {
                bool same = true;
#pragma unroll
                for (int j = 0; j < MW; j++) same = same && ld64(bp + t * KW + 1 + j) == M2[j];
                if (same) {
                  const uint64_t theirs[kCountWords] = {ld64(bp + t * KW + 1 + MW), ld64(bp + t * KW + 2 + MW)};
                  dom = dom || counts_ge(C2, theirs, top);
                }
              }
            }
            if (dom) pending = false;
            else if (empty) { cand = true; pending = false; }          // the chain ends here: nothing visited dominates it
            else { b = (b + 1u) & bmask; if (++full_buckets > bmask) pending = false; }
          }
        }
        uint64_t rem = __ballot(cand);
        while (rem) {
          const uint32_t l0 = (uint32_t)__builtin_ctzll(rem);
          rem &= rem - 1ull;
          const uint64_t bk0 = rl64(k0, l0);
          bool same = k0 == bk0;
#pragma unroll
          for (int j = 0; j < MW; j++) same = same && M2[j] == rl64(M2[j], l0);
          const uint64_t theirs[kCountWords] = {rl64(C2[0], l0), rl64(C2[1], l0)};
          if (cand && lr > l0 && same && counts_ge(C2, theirs, top)) cand = false;
        }
        bool ins = cand;
        while (__ballot(ins)) {
          if (ins) {
            gu64* bp = tab + (uint64_t)b * (4 * KW);
            uint32_t empty = 0;
#pragma unroll
            for (int t = 0; t < 4; t++) if ((uint32_t)ld64(bp + t * KW) == 0u) empty |= 1u << t;
            if (empty) {
              const uint32_t t = (uint32_t)__builtin_ctz(empty);
              gu64* e = bp + t * KW;
              if (cas64_from_zero(e, k0) == 0ull) {
#pragma unroll
                for (int j = 0; j < MW; j++) st64(e + 1 + j, M2[j]);
                st64(e + 1 + MW, C2[0]); st64(e + 2 + MW, C2[1]);
                idx = b * 4u + t; ins = false;
              }                                      // else another lane took it in this very step: look at the bucket again
            } else { b = (b + 1u) & bmask; if (++full_buckets > bmask) ins = false; }
          }
        }
        is_new = cand;
      } else {
      bool pending = viable, fresh = false, won = false, lost = false;
      while (__ballot(pending)) {
        if (pending) {
          const uint32_t me = scan_bucket<MW>(tab, b, k0, M2), match = me & 15u, empty = me >> 4;
          if (match) {
            idx = b * 4u + (uint32_t)__builtin_ctz(match);
            fresh = lost;      // a lost CAS followed by a match in that bucket: inserted in THIS round
            pending = false;   //   by a sibling lane
          } else if (empty) {
            idx = b * 4u + (uint32_t)__builtin_ctz(empty);
            gu64* e = tab + (uint64_t)idx * KW;
            const uint64_t old = cas64_from_zero(e, k0);
            if (old == 0ull) {                         // claimed an empty entry
#pragma unroll
              for (int j = 0; j < MW; j++) st64(e + 1 + j, M2[j]);
              won = true; fresh = true; pending = false;
            } else {
              lost = true;     // claimed by another lane in this very step: look at the same bucket again
            }
          } else {
            lost = false;
            b = (b + 1u) & bmask;
            if (++full_buckets > bmask) pending = false;   // every bucket full: cannot happen below the 3/4 fill bound
          }
        }
      }
      // the lowest lane among the lanes that produced one and the same new config keeps it
      is_new = fresh;
      uint64_t dupl = __ballot(fresh && !won);
      while (dupl) {
        const uint32_t l0 = (uint32_t)__builtin_ctzll(dupl);
        const uint32_t i0 = rl(idx, l0);
        const uint64_t grp = __ballot(fresh && idx == i0);
        const uint32_t winner = (uint32_t)__builtin_ctzll(grp);
        if ((grp >> lr) & 1ull) is_new = lr == winner;
        dupl &= ~grp;
      }
      }   // !CNT
      if (__ballot(full_buckets > bmask)) { verdict = TBC_UNKNOWN; cause = TBC_CAUSE_VISITED_FULL; break; }
      SEG(3);
      if (is_new) st64(par + idx, (uint64_t)pslot | ((uint64_t)(op + 1u) << 32));
      const uint64_t nb0 = __ballot(is_new);
      // ---- lookahead (register / cas-register): a new config is dead if the call completing at one of the
      // next kLookahead ranks can never be linearized from it: not linearized yet, it needs a value that is
      // neither the state nor produced by any call still to be linearized before that completion (calls
      // invoked later, open calls not yet linearized, the calls completing in between).  Dead configs stay
      // in the visited set and are set aside on a second stack (expanded only if the live ones lead to no
      // linearization).  8 lanes per config, one rank each: one trip, L1-resident.
      bool dead = false;
      if (look_on && nb0) {
        const uint32_t ci = (uint32_t)__popcll(nb0 & ((1ull << lr) - 1ull)), nn0 = (uint32_t)__popcll(nb0);
        if (is_new) {
          c_fi[ci] = fi2; c_st[ci] = (uint32_t)st2;
#pragma unroll
          for (int j = 0; j < MW; j++) c_M[ci * MW + j] = M2[j];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (uint32_t cb = 0; cb < nn0; cb += 8u) {
          const uint32_t c = cb + (lr >> 3), j = lr & 7u;
          const uint32_t cF0 = c < nn0 ? c_fi[c] : 0u;
          const bool val = c < nn0 && (!CNT || cF0 + j < RT);          // (completions past a prefix target constrain nothing; RT = R otherwise: the padding records say so themselves)
          const uint32_t cF = val ? cF0 : 0u;
          const int32_t cs = val ? (int32_t)c_st[c] : 0;
          uint64_t Mc[MW], pm[MW];
#pragma unroll
          for (int w = 0; w < MW; w++) { Mc[w] = val ? c_M[c * MW + w] : 0ull; pm[w] = 0ull; }
          uint64_t w0 = (uint64_t)(kLookNone << 16 | kLookNone << 24);
          if (val) {
            const uint64_t* rec = look + (uint64_t)(cF + j) * (MW + 1);
            w0 = rec[0];
#pragma unroll
            for (int w = 0; w < MW; w++) pm[w] = rec[1 + w];
          }
          const uint32_t slot = (uint32_t)w0 & 0xFFFFu, need = (uint32_t)(w0 >> 16) & 0xFFu, prod = (uint32_t)(w0 >> 24) & 0xFFu;
          const uint32_t dinv = (uint32_t)(w0 >> 32) & 0xFFu, dprod = (uint32_t)(w0 >> 40) & 0xFFu;
          bool bit = false, pmhit = false;
#pragma unroll
          for (int w = 0; w < MW; w++) {
            if ((slot >> 6) == (uint32_t)w) bit = (Mc[w] >> (slot & 63u)) & 1ull;
            pmhit = pmhit || (pm[w] & ~Mc[w]) != 0ull;
          }
          const bool linz = dinv >= j && bit;                       // open at the config's front and linearized
          // values the calls completing at the ranks before this one can still provide (prefix-OR over the group)
          uint32_t acc = (prod != kLookNone && !linz) ? 1u << prod : 0u;
          uint32_t x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)acc, 0x111, 0xf, 0xf, true);   // row_shr:1
          if (j >= 1u) acc |= x;
          x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)acc, 0x112, 0xf, 0xf, true);            // row_shr:2
          if (j >= 2u) acc |= x;
          x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)acc, 0x114, 0xf, 0xf, true);            // row_shr:4
          if (j >= 4u) acc |= x;
          uint32_t before = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)acc, 0x111, 0xf, 0xf, true);
          if (j == 0u) before = 0u;
          // (count form: bit 48 = a crashed call producing `need` is invoked by that rank: available whatever the counts)
          const bool ok = need == kLookNone || linz || (int32_t)need == cs || dprod < j || pmhit || ((before >> need) & 1u) || (CNT && ((w0 >> 48) & 1ull));
          const uint64_t bad = __ballot(val && !ok);
          if (is_new && ci >= cb && ci < cb + 8u) dead = ((bad >> (8u * (ci - cb))) & 0xFFull) != 0ull;
        }
      }
      const bool keep = is_new && !dead;
      const uint64_t db = __ballot(is_new && dead);
      if (db) {                                   // set aside, in pair order
        if (is_new && dead) st32(dstack + dsp + (uint32_t)__popcll(db & ((1ull << lr) - 1ull)), idx);
        dsp += (uint32_t)__popcll(db);
      }
      const uint64_t nb = __ballot(keep);
      const uint32_t nn = (uint32_t)__popcll(nb);
      if (is_new) lane_maxf = max(lane_maxf, fi2);
      if (keep) {
        const uint32_t pos = sp + (uint32_t)__popcll(nb & ((1ull << lr) - 1ull));
        st32(stack + pos, idx);
        {                                           // mirror it in the LDS ring (nn <= 64 = kRing: no clash)
          const uint32_t rs = pos & (kRing - 1u);
          r_pos[rs] = pos; r_idx[rs] = idx; r_k0[rs] = k0;
#pragma unroll
          for (int j = 0; j < MW; j++) r_M[rs * MW + j] = M2[j];
          if constexpr (CNT) { r_C[rs * 2] = C2[0]; r_C[rs * 2 + 1] = C2[1]; }
          r_off[rs] = co0; r_nlive[rs] = co1 - co0; r_cnt[rs] = (co1 - co0) + cnc;
        }
      }
      sp += nn; visited += (uint32_t)__popcll(nb0);
      SEG(4);
    }
    max_sp = max(max_sp, sp);
    if (__builtin_expect((iterations & 63u) == 0u, 0)) {      // the clock, and the progress words of a debug run
      const cold_args_ptr C = cold_args();
      uint32_t* const dbg = C->dbg;
      if (dbg && lr == 0 && (iterations & 255u) == 0u) {
        dbg[8] = hidx; dbg[9] = iterations; dbg[10] = sp; dbg[11] = probes;
        dbg[12] = visited; dbg[13] = maxcnt; dbg[14] = np; dbg[15] = rounds;
      }
      const uint64_t limit = C->time_limit_ticks;
      if (verdict == -2 && limit && probes <= probe_room && (uint64_t)wall_clock64() - sld64(S, S_T0) > limit) {
        verdict = TBC_UNKNOWN; cause = TBC_CAUSE_TIME_LIMIT;
      }
      // (BeamArgs.abort: somebody else has decided this history meanwhile -- read past the caches, it is written while the kernel runs)
      const uint32_t* const ab = C->abort;
      const uint32_t* const abm = C->abort_map;
      if (ab && verdict == -2 && __hip_atomic_load(ab + (abm ? abm[hidx] : hidx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) {
        verdict = TBC_UNKNOWN; cause = TBC_CAUSE_STEP_LIMIT;
      }
    }
  }
  if (need_park && verdict == -2) {       // left to look at the totals: over the step limit?
    const uint64_t ms = cold_args()->max_steps;
    if (ms && probes0 + probes > ms) { verdict = TBC_UNKNOWN; cause = TBC_CAUSE_STEP_LIMIT; }
  }

  // ---- park the state
  {
    uint32_t mf = lane_maxf;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mf = max(mf, (uint32_t)__shfl_xor(mf, d));
    if (lane == 0) {
      sst(S, S_SP, sp); sst(S, S_MAXSP, max_sp); sst(S, S_MAXF, mf); sst(S, S_K, K); sst(S, S_DSP, dsp);
      sst(S, S_VERDICT, (uint32_t)verdict); sst(S, S_CAUSE, (uint32_t)cause);
      sst64(S, S_PROBES, probes0 + probes); sst64(S, S_VISITED, (uint64_t)visited); sst64(S, S_EXPANDED, sld64(S, S_EXPANDED) + expanded);
      sst64(S, S_ITER, sld64(S, S_ITER) + iterations); sst64(S, S_ROUNDS, rounds0 + rounds);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (need_switch) {       // the set-aside configs become the stack; no lookahead from here on
    if (lane == 0) {
      const uint64_t a = S[S_STACK] | ((uint64_t)S[S_STACK + 1] << 32), b = S[S_DSTACK] | ((uint64_t)S[S_DSTACK + 1] << 32);
      sst64(S, S_STACK, b); sst64(S, S_DSTACK, a);
      sst(S, S_SP, S[S_DSP]); sst(S, S_DSP, 0u); sst(S, S_EXACT, 1u);
    }
    if (lane < kRing) r_pos[lane] = kNone;     // ring entries are keyed by stack position
    continue;
  }
  if (need_park) continue;
  if (!need_grow) break;
  if (!grow_visited_set<MW, CNT>(S, r_pos, lane)) {
    if (lane == 0) { sst(S, S_VERDICT, (uint32_t)TBC_UNKNOWN); sst(S, S_CAUSE, (uint32_t)TBC_CAUSE_VISITED_FULL); }
    break;
  }
  }   // for (;;)

  // ---- results
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const gu64* tab = (const gu64*)sld64(S, S_TAB);
  const uint32_t cap_log2 = sld(S, S_CAP);
  const gu64* par = tab + ((uint64_t)KW << cap_log2);
  const int32_t verdict = (int32_t)sld(S, S_VERDICT), cause = (int32_t)sld(S, S_CAUSE);
  const uint32_t maxf = sld(S, S_MAXF), win_parent = sld(S, S_WINPAR), win_op = sld(S, S_WINOP);
  const int32_t win_state = (int32_t)sld(S, S_WINSTATE);
  const uint64_t probes = sld64(S, S_PROBES), visited = sld64(S, S_VISITED), expanded = sld64(S, S_EXPANDED);
  const uint64_t iterations = sld64(S, S_ITER), rounds = sld64(S, S_ROUNDS);
  const uint32_t max_sp = sld(S, S_MAXSP);
  // what the result needs of the arguments and of the history's header is fetched again here rather than kept
  // across the round loop
  const cold_args_ptr C = cold_args();
  const Hist* const Hc = C->hist + hidx;
  const uint64_t op_off_c = ru64(Hc->op_off), ret_off_c = ru64(Hc->ret_off);
  const uint32_t Rc = rfl(Hc->n_ret);
  DevResult* const out = C->results + hidx;
  // ---- invalid: the configs stuck at the failing completion (knossos :configs), by a scan of the visited set
  uint32_t n_cfg = 0;
  if (verdict == TBC_INVALID && C->cfg) {
    uint64_t* cfg = C->cfg + (uint64_t)hidx * kCfgCap * (2 + MW);
    const uint64_t ncap = 1ull << cap_log2;
    for (uint64_t s0 = 0; s0 < ncap; s0 += 64) {
      const gu64* e = tab + (s0 + lane) * KW;
      const uint64_t k0 = ld64(e);
      const bool hit = (uint32_t)k0 == maxf + 1u;
      const uint64_t hb = __ballot(hit);
      if (hit) {
        const uint32_t pos = n_cfg + (uint32_t)__popcll(hb & ((1ull << lane) - 1ull));
        if (pos < kCfgCap) {
          uint64_t* o = cfg + (uint64_t)pos * (2 + MW);
          o[0] = k0;
#pragma unroll
          for (int j = 0; j < MW; j++) o[1 + j] = ld64(e + 1 + j);
          const uint64_t pw = ld64(par + s0 + lane);
          o[1 + MW] = (uint32_t)pw == kNone ? (uint64_t)TBC_NO_OP : (pw >> 32) - 1ull;
        }
      }
      n_cfg += (uint32_t)__popcll(hb);
    }
  }
  if (CNT && verdict == TBC_VALID && RT < Rc && C->cfg) n_cfg = 1;          // (the prefix search's end config, written where the search ended)
  uint32_t wlen = 0;
  if (verdict == TBC_VALID && Rc != 0 && C->witness) {
    // witness = ops along the parent chain of the winning config, then the winning op.  Only when it is wanted: the
    // chain is thousands of dependent loads (a tenth of the whole search of a 10k-op history), and its length means
    // nothing to a caller who does not get the ops
    wlen = 1;
    uint32_t id = win_parent;
    for (;;) {
      const uint32_t pr = (uint32_t)ld64(par + id);
      if (pr == kNone) break;
      wlen++; id = pr;
    }
    {
      uint32_t* wit = C->witness + op_off_c;
      uint32_t w = wlen - 1;
      if (lane == 0) wit[w] = win_op;
      id = win_parent;
      for (;;) {
        const uint64_t po = ld64(par + id);
        const uint32_t pr = (uint32_t)po;
        if (pr == kNone) break;
        w--;
        if (lane == 0) wit[w] = (uint32_t)(po >> 32) - 1u;
        id = pr;
      }
    }
  }
  if (lane == 0) {
    out->valid = verdict; out->cause = cause; out->max_front = maxf; out->depth = wlen;
    out->final_state = win_state; out->n_configs = n_cfg;
    out->fail_op = TBC_NO_OP; out->prev_ok_op = TBC_NO_OP;
    if (verdict == TBC_INVALID) {
      const uint32_t* ret_op = C->ret_op + ret_off_c;
      out->fail_op = ret_op[maxf];
      if (maxf) out->prev_ok_op = ret_op[maxf - 1];
    }
    out->steps = probes; out->visited = visited; out->probes = probes; out->backtracks = expanded;
    out->max_depth = max_sp; out->bucket_reads = rounds; out->tab_log2 = cap_log2;
    if (verdict != TBC_UNKNOWN && C->progress) {          // tbc_batch_progress: a wavefront per history, every count is published
      const uint32_t n = atomicAdd(C->progress_dev, 1u) + 1u;
      __hip_atomic_store(C->progress, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // (a race of list orders: this history is decided -- whoever searches it in another order may stop)
    uint32_t* const done = C->abort_set;
    const uint32_t* const dmap = C->abort_map;
    if (done && (verdict == TBC_VALID || verdict == TBC_INVALID)) __hip_atomic_store(done + (dmap ? dmap[hidx] : hidx), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // the search state as it is left (batch_run.hip, race_orders: a history whose budgeted pass ended at the budget is taken up from here
  // while other list orders race it -- the work done so far is not done again)
  {
    uint32_t* const park = C->park;
    if (park && lane < S_WORDS) park[(uint64_t)hidx * kParkWords + lane] = S[lane];
  }
  uint32_t* const dbg = C->dbg;
  if (dbg && lane == 0) {
    dbg[4] = 0x300u + hidx; dbg[16] = (uint32_t)verdict; dbg[17] = (uint32_t)iterations;
#ifdef TBC_SEGPROF
    for (int i = 0; i < 5; i++) { dbg[20 + 2 * i] = (uint32_t)seg[i]; dbg[21 + 2 * i] = (uint32_t)(seg[i] >> 32); }
#endif
    dbg[30] = (uint32_t)rounds;
  }
#undef SEG
}

// Register budget (MW = 1, state-carrying models): 67 VGPRs on its own, 64 under the min-blocks bound (two
// spilled values) -> 8 wavefronts per SIMD (__launch_bounds__' second argument is wavefronts per SIMD); measured 350 -> 291 ms on the 8192-history probe against 7
// (profiles/r01_vgpr_ab.txt has the whole history: 116 with segment profiling compiled in, 94 without, 107
// with the bucket probe until the search state was parked in LDS around the growth path).  The commutative
// models' kernels need ~90 and are left to the register allocator.
#ifndef TBC_BEAM_MIN_WAVES
#define TBC_BEAM_MIN_WAVES 8
#endif
// Wavefronts per workgroup.  A workgroup's slots are handed back only when its last wavefront ends and
// histories differ 5x in how long they take, so fewer is better for the tail of a batch -- but one-wavefront
// workgroups measured slower (profiles/r01_vgpr_ab.txt), so this is a build-time knob with its A/B on file.
#ifndef TBC_BEAM_WAVES
#define TBC_BEAM_WAVES 4
#endif
constexpr uint32_t kBeamWaves = TBC_BEAM_WAVES;
template <int MW, bool COMM, bool REGF, bool CNT = false>
// (the table / multi-register instance takes the registers it needs: under the register family's bound it spilled 108 scalar registers)
__global__ __launch_bounds__(64 * kBeamWaves, (COMM || CNT || !REGF) ? 1 : TBC_BEAM_MIN_WAVES) void wgl_beam_kernel(BeamArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const uint32_t lane = threadIdx.x & 63u, wv = rfl(threadIdx.x >> 6);
  const uint32_t w = blockIdx.x * kBeamWaves + wv;
  if (w < A.n_work) beam_one<MW, COMM, REGF, CNT>(A, rfl(A.work[w]), lds + wv * beam_lds_words(MW, CNT), lane);
}

template <int MW>
void launch_beam_mw(const BeamArgs& a, hipStream_t s) {
  const bool cnt = (a.rules & kRuleCount) != 0u;       // count form: crashed calls as counts per class (register family only)
  const size_t lds = (size_t)kBeamWaves * beam_lds_words(MW, cnt) * 4;
  const uint32_t n_blocks = (a.n_work + kBeamWaves - 1) / kBeamWaves;
  if (cnt) {
    if constexpr (MW <= 2) hipLaunchKernelGGL((wgl_beam_kernel<MW, false, true, true>), dim3(n_blocks), dim3(64 * kBeamWaves), lds, s, a);
    return;
  }
  // the commutative (set / bank) models get their own instantiation: their evaluation code would
  // otherwise double the register budget of the register-family kernel
  if (a.model_kind == TBC_MODEL_SET || a.model_kind == TBC_MODEL_BANK)
    hipLaunchKernelGGL((wgl_beam_kernel<MW, true, false>), dim3(n_blocks), dim3(64 * kBeamWaves), lds, s, a);
  else if (a.model_kind == TBC_MODEL_REGISTER || a.model_kind == TBC_MODEL_CAS_REGISTER || a.model_kind == TBC_MODEL_MUTEX)
    hipLaunchKernelGGL((wgl_beam_kernel<MW, false, true>), dim3(n_blocks), dim3(64 * kBeamWaves), lds, s, a);
  else
    hipLaunchKernelGGL((wgl_beam_kernel<MW, false, false>), dim3(n_blocks), dim3(64 * kBeamWaves), lds, s, a);
}

}  // namespace

bool launch_beam(const BeamArgs& a, uint32_t mask_words, uint32_t /*n_blocks: derived from n_work*/, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  switch (mask_words) {
    case 1: launch_beam_mw<1>(a, s); return true;
    case 2: launch_beam_mw<2>(a, s); return true;
    case 4: launch_beam_mw<4>(a, s); return true;
    default: return false;
  }
}

}  // namespace tbc
