// wave_env_wg.h -- the WORKGROUP primitives jit_sweep_wg_impl.h needs on top of wave_env.h (gfx950): several wavefronts of one
// workgroup working on one LDS-resident structure.  As with wave_env.h the body stays plain per-lane C++ and the same file
// compiles for the host emulator (tests/emu/wave_env_wg_emu.h: NW x 64 fibers, the workgroup barrier a rendezvous of all of
// them, the wavefronts interleaved in a seeded order between rendezvous).
#pragma once
#include "wave_env.h"
#if defined(TBC_EMU)
#include "wave_env_wg_emu.h"
#else

namespace wv {

// every wavefront of the workgroup has done its LDS accesses before any goes on (s_barrier + workgroup-scope fence)
WV_DEV void wg_barrier() { __syncthreads(); }
WV_DEV uint32_t wg_thread() { return threadIdx.x; }                   // 0 .. 64 * NW - 1; wave = thread / 64, lane = thread % 64
WV_DEV uint32_t wg_index() { return blockIdx.x; }

// LDS words several wavefronts race on: workgroup-scope atomics
WV_DEV uint32_t lds_ld32(const uint32_t* p) { return __hip_atomic_load((const WV_LDS uint32_t*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
WV_DEV uint32_t lds_cas32(uint32_t* p, uint32_t expected, uint32_t desired) {        // returns what was there
  __hip_atomic_compare_exchange_strong((WV_LDS uint32_t*)p, &expected, desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  return expected;
}
WV_DEV void lds_or32(uint32_t* p, uint32_t v) { (void)__hip_atomic_fetch_or((WV_LDS uint32_t*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
WV_DEV uint32_t lds_fetch_add32(uint32_t* p, uint32_t v) { return __hip_atomic_fetch_add((WV_LDS uint32_t*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
WV_DEV void lds_add32_wg(uint32_t* p, uint32_t v) { (void)__hip_atomic_fetch_add((WV_LDS uint32_t*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }   // (ds_add, no return)

// OR over the wavefront's lanes (every lane gets it)
WV_DEV uint32_t wave_or32(uint32_t v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v |= (uint32_t)__shfl_xor((int)v, d);
  return v;
}

}  // namespace wv
#endif
