// jit_sweep_wg.hip -- K6w: the level sweep with a WORKGROUP of wavefronts per (history, segment, 32 origins) (gfx950); the body
// is jit_sweep_wg_impl.h.  Same inputs, same records as jit_sweep_kernel (jit_sweep.hip), which keeps the dump pass and every
// model outside the register family.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "tbc_internal.h"
#include "jit_sweep_wg_impl.h"

namespace tbc {

namespace {

template <uint32_t CAP, uint32_t NW>
__global__ __launch_bounds__(64 * NW) void jit_sweep_wg_kernel(SweepArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  sweepwg::segment<CAP, NW>(A, lds);
}
// the first pass's form since round 5: the compact walk, narrow passes by wavefront 0 alone (jit_sweep_wg_impl.h, COMPACT + SOLO)
template <uint32_t CAP, uint32_t NW>
__global__ __launch_bounds__(64 * NW) void jit_sweep_wg_compact_kernel(SweepArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  sweepwg::segment<CAP, NW, true, true>(A, lds);
}

// the RELAXED sweep of a count-form history (jit_sweep_wg_impl.h, RLX): the plain walk over compound steps
template <uint32_t CAP, uint32_t NW>
__global__ __launch_bounds__(64 * NW) void jit_sweep_wg_relaxed_kernel(SweepArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  sweepwg::segment<CAP, NW, false, false, true>(A, lds);
}

template <uint32_t CAP, uint32_t NW>
bool launch_relaxed(const SweepArgs& a, hipStream_t s) {
  constexpr uint32_t bytes = sweepwg::lds_words<CAP, NW>() * 4;
  static bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&jit_sweep_wg_relaxed_kernel<CAP, NW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
  if (!ok) return false;
  const uint32_t groups = a.seg_list ? a.n_list : a.n_hist * a.max_segs * kSweepSlices;
  hipLaunchKernelGGL((jit_sweep_wg_relaxed_kernel<CAP, NW>), dim3(groups), dim3(64 * NW), bytes, s, a);
  return true;
}

template <uint32_t CAP, uint32_t NW>
bool launch_one(const SweepArgs& a, hipStream_t s) {
  constexpr uint32_t bytes = sweepwg::lds_words<CAP, NW>() * 4;
  static bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&jit_sweep_wg_kernel<CAP, NW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
  if (!ok) return false;
  const uint32_t groups = a.seg_list ? a.n_list : a.n_hist * a.max_segs * kSweepSlices;
  hipLaunchKernelGGL((jit_sweep_wg_kernel<CAP, NW>), dim3(groups), dim3(64 * NW), bytes, s, a);
  return true;
}

template <uint32_t CAP, uint32_t NW>
bool launch_compact(const SweepArgs& a, hipStream_t s) {
  constexpr uint32_t bytes = sweepwg::lds_words<CAP, NW, true>() * 4;
  static bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&jit_sweep_wg_compact_kernel<CAP, NW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
  if (!ok) return false;
  hipLaunchKernelGGL((jit_sweep_wg_compact_kernel<CAP, NW>), dim3(a.n_hist * a.max_segs * kSweepSlices), dim3(64 * NW), bytes, s, a);
  return true;
}

}  // namespace

// the first pass of the sweep (cuts are in place): eight wavefronts per workgroup, sets of kSweepCapMid configs, the compact walk with
// narrow passes by wavefront 0 alone (81 KB of LDS: two workgroups per CU) -- measured round 5 against the plain walk, the ring, the
// fingerprint and sixteen wavefronts on the big sets (834 us of search per 10k-op history against 928 / 1,047 / 947 / 1,203:
// profiles/r05_single_history_forms_first_device_run.json; the losers are deleted); the second pass (a.seg_list: the segments that
// overflowed those sets): sets of kSweepCapBig configs, the plain walk (148 KB: one workgroup per CU)
bool launch_sweep_wg(const SweepArgs& a, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!(a.model_kind == TBC_MODEL_REGISTER || a.model_kind == TBC_MODEL_CAS_REGISTER) || a.dump_cfg) return false;
  if (a.reach_hdr) return a.seg_list ? launch_relaxed<kSweepCapBig, 8>(a, s) : launch_relaxed<kSweepCapMid, 8>(a, s);
  if (a.seg_list) return launch_one<kSweepCapBig, 8>(a, s);
  return launch_compact<kSweepCapMid, 8>(a, s);
}

}  // namespace tbc
