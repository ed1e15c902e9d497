// wave_env.h -- the handful of wavefront primitives wgl_narrow_impl.h is written against (gfx950).
//
// Everything a kernel body does ACROSS lanes (ballots, lane reads, row shifts, the wavefront barrier) and every
// access whose flavour matters (agent-scope loads / stores / CAS on the visited set, LDS atomics without return)
// goes through this header, so that the body itself is plain per-lane C++.  That buys one thing: the very same
// body can be compiled for a lane-accurate HOST EMULATOR (tests/emu/wave_env_emu.h: 64 fibers per wavefront, every
// cross-lane primitive a rendezvous) and compared with the oracle on the CPU, where a schedule bug costs seconds to
// find instead of a GPU box.  The emulator is test infrastructure: it is compiled only by tests/emu/ (TBC_EMU is
// defined nowhere else), libtbcheck.so contains this device version only and has no CPU path.
#pragma once
#if defined(TBC_EMU)
#include "wave_env_emu.h"
#else
#include <hip/hip_runtime.h>
#include <cstdint>

#define WV_DEV __device__ __forceinline__
#define WV_MEM __device__ __forceinline__          /* the same for member functions */
#define WV_HD __host__ __device__
#define WV_GLOBAL __attribute__((address_space(1)))
#define WV_LDS __attribute__((address_space(3)))
#define WV_UNROLL _Pragma("unroll")
#define WV_NOUNROLL _Pragma("unroll 1")

namespace wv {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// register arrays indexed by a UNIFORM index (s_set_gpr_idx: one indexed move each way, no select chain)
typedef uint32_t u32x16 __attribute__((vector_size(64)));
typedef uint32_t u32x32 __attribute__((vector_size(128)));
typedef WV_GLOBAL uint64_t gu64;
typedef WV_GLOBAL uint32_t gu32;

// ---- cross-lane
WV_DEV uint64_t ballot(bool p) { return __ballot(p); }
WV_DEV uint32_t readlane(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }      // l uniform
WV_DEV uint64_t readlane64(uint64_t v, uint32_t l) { return (uint64_t)readlane((uint32_t)v, l) | ((uint64_t)readlane((uint32_t)(v >> 32), l) << 32); }
WV_DEV uint32_t readfirstlane(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
// the value of lane - N within the lane's row of 16, 0 where that leaves the row (DPP row_shr:N, bound_ctrl)
template <int N>
WV_DEV uint32_t row_shr0(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x110 + N, 0xf, 0xf, true); }
// all lanes have done their LDS / memory accesses before any lane goes on (one wavefront: no s_barrier needed)
WV_DEV void barrier() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
WV_DEV void threadfence() { __threadfence(); }
// Stores of this wavefront (or of the workgroup's other wavefronts, behind a workgroup barrier) are in L2 before what follows reads them
// back through L2 (sc1 loads) -- a WORKGROUP-scope fence: s_waitcnt, nothing else.  __threadfence() is an AGENT-scope fence, and on a
// chip of eight XCDs whose L2s are not coherent with each other that means `buffer_wbl2 sc1` + `buffer_inv sc1`: the XCD's whole L2
// written back and its lines dropped -- for every other wavefront on the XCD too.  Round 6 found one of those in every workgroup of
// pack_wg64_kernel and one per finished history in wgl_narrow_kernel (32,768 a launch); nothing in these kernels is read by another
// compute unit before the kernel ends.
WV_DEV void wg_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); }
// tbc_batch_progress: one more history decided.  The count lives in HBM (an atomic per history straight into HOST memory was measured:
// 32,768 of them on one address serialize at the host bridge, ~0.6 us each, and a wavefront's later loads wait behind its own -- the
// headline search went 57 -> 77 ms); every `every_mask + 1`-th count is published to the host word by a plain store.
WV_DEV void count_decided(uint32_t* dev, uint32_t* host, uint32_t every_mask) {
  const uint32_t n = atomicAdd(dev, 1u) + 1u;
  if ((n & every_mask) == 0u) __hip_atomic_store(host, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// the lane number, opaque to the optimiser: what an iteration derives from it is recomputed instead of kept live
WV_DEV uint32_t opaque(uint32_t v) { asm volatile("" : "+v"(v)); return v; }

// ---- the visited set and the stacks: agent scope (sc1), never a stale line of this CU's L1
WV_DEV uint64_t ld64(const gu64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
WV_DEV void st64(gu64* p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
WV_DEV uint32_t ld32(const gu32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
WV_DEV void st32(gu32* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// the same memory when ONE wavefront owns it for the whole launch (a history's visited set and stacks in wgl_narrow): plain
// accesses.  They stay coherent inside the compute unit (every access of its waves goes through its L1), several loads of one
// line become one request to L2 (an sc1 load is a request of its own: a bucket read as 4-8 of them was 4-8 requests), and a
// plain store leaves the line in L2 to be merged with its neighbours' (an sc1 store writes its sector through to memory).
WV_DEV uint64_t own_ld64(const gu64* p) { return *p; }
WV_DEV void own_st64(gu64* p, uint64_t v) { *p = v; }
WV_DEV uint32_t own_ld32(const gu32* p) { return *p; }
WV_DEV void own_st32(gu32* p, uint32_t v) { *p = v; }
WV_DEV u32x4 own_ld128(const gu64* p) { return *reinterpret_cast<const WV_GLOBAL u32x4*>(p); }
// claim an empty entry: returns what was there (0 = claimed)
WV_DEV uint64_t cas64_from_zero(gu64* p, uint64_t desired) {
  uint64_t expected = 0ull;
  __hip_atomic_compare_exchange_strong(p, &expected, desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return expected;
}
// the four 16 B entries of one 64 B bucket, four loads in flight: one trip
WV_DEV void ld_bucket16(const gu64* bucket, u32x4& e0, u32x4& e1, u32x4& e2, u32x4& e3) {
  asm volatile(
      "global_load_dwordx4 %0, %4, off sc1\n\t"
      "global_load_dwordx4 %1, %4, off offset:16 sc1\n\t"
      "global_load_dwordx4 %2, %4, off offset:32 sc1\n\t"
      "global_load_dwordx4 %3, %4, off offset:48 sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(e0), "=&v"(e1), "=&v"(e2), "=&v"(e3)
      : "v"(bucket)
      : "memory");
}

// every store of this wavefront has reached memory (gfx9: stores count in vmcnt)
WV_DEV void wait_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ---- LDS counters: read-modify-write without a return value (ds_add / ds_max), one lane per word at a time
WV_DEV void lds_add32(uint32_t* p, uint32_t v) { (void)__hip_atomic_fetch_add((WV_LDS uint32_t*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
WV_DEV void lds_add64(uint32_t* p, uint64_t v) { (void)__hip_atomic_fetch_add((WV_LDS uint64_t*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
WV_DEV void lds_max32(uint32_t* p, uint32_t v) { (void)__hip_atomic_fetch_max((WV_LDS uint32_t*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }

WV_DEV uint64_t clock100mhz() { return (uint64_t)wall_clock64(); }

// schedule statistics of an emulated run (tests/emu): nothing on the device
WV_DEV void stat(uint32_t, uint64_t) {}

// arguments only cold paths read come from the kernarg segment where they are used (kept out of the loop's registers)
template <class Args>
WV_DEV const __attribute__((address_space(4))) Args* cold(const Args&) {
  const __attribute__((address_space(4))) Args* p = (const __attribute__((address_space(4))) Args*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return p;
}

}  // namespace wv
#endif
