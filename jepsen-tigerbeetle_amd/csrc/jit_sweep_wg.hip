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
// the experimental ring form (jit_sweep_wg_impl.h, QUEUE): a kernel of its own so that the one above stays exactly what was measured
template <uint32_t CAP, uint32_t NW, bool FP = false>
__global__ __launch_bounds__(64 * NW) void jit_sweep_wg_ring_kernel(SweepArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  sweepwg::segment<CAP, NW, true, FP>(A, lds);
}
// ... and the fingerprint form of the plain sweep (FP)
template <uint32_t CAP, uint32_t NW>
__global__ __launch_bounds__(64 * NW) void jit_sweep_wg_fp_kernel(SweepArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  sweepwg::segment<CAP, NW, false, true>(A, lds);
}

// ... and the compact walk (COMPACT), with or without the fingerprint
template <uint32_t CAP, uint32_t NW, bool FP = false, bool SOLO = false>
__global__ __launch_bounds__(64 * NW) void jit_sweep_wg_compact_kernel(SweepArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  sweepwg::segment<CAP, NW, false, FP, true, SOLO>(A, lds);
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

template <uint32_t CAP, uint32_t NW, bool FP>
bool launch_ring(const SweepArgs& a, hipStream_t s) {
  constexpr uint32_t bytes = sweepwg::lds_words<CAP, NW, true>() * 4;
  static bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&jit_sweep_wg_ring_kernel<CAP, NW, FP>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
  if (!ok) return false;
  hipLaunchKernelGGL((jit_sweep_wg_ring_kernel<CAP, NW, FP>), dim3(a.n_hist * a.max_segs * kSweepSlices), dim3(64 * NW), bytes, s, a);
  return true;
}
template <uint32_t CAP, uint32_t NW>
bool launch_fp(const SweepArgs& a, hipStream_t s) {
  constexpr uint32_t bytes = sweepwg::lds_words<CAP, NW>() * 4;
  static bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&jit_sweep_wg_fp_kernel<CAP, NW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
  if (!ok) return false;
  hipLaunchKernelGGL((jit_sweep_wg_fp_kernel<CAP, NW>), dim3(a.n_hist * a.max_segs * kSweepSlices), dim3(64 * NW), bytes, s, a);
  return true;
}

template <uint32_t CAP, uint32_t NW, bool FP, bool SOLO>
bool launch_compact(const SweepArgs& a, hipStream_t s) {
  constexpr uint32_t bytes = sweepwg::lds_words<CAP, NW, false, true>() * 4;
  static bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&jit_sweep_wg_compact_kernel<CAP, NW, FP, SOLO>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
  if (!ok) return false;
  hipLaunchKernelGGL((jit_sweep_wg_compact_kernel<CAP, NW, FP, SOLO>), dim3(a.n_hist * a.max_segs * kSweepSlices), dim3(64 * NW), bytes, s, a);
  return true;
}

}  // namespace

// the first pass of the sweep (cuts are in place): `waves` wavefronts per workgroup, sets of kSweepCapMid configs (78 KB of LDS
// with 8 wavefronts: two workgroups per CU); the second pass (a.seg_list: the segments that overflowed those): sets of
// kSweepCapBig configs, 8 wavefronts (148 KB: one workgroup per CU)
bool launch_sweep_wg(const SweepArgs& a, uint32_t waves, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!(a.model_kind == TBC_MODEL_REGISTER || a.model_kind == TBC_MODEL_CAS_REGISTER) || a.dump_cfg) return false;
  if (a.seg_list) return launch_one<kSweepCapBig, 8>(a, s);
  // TBC_SWEEP_WG_RING=1: the first pass in the ring form (89 KB of LDS: one workgroup per CU).  Verified under the emulator only
  // (tests/test_sweep_wg_emu.py); nothing takes it unless asked
  // TBC_SWEEP_WG_FP=1: a fingerprint of the key in the table word (with or without the ring).  The same standing.
  static const bool ring = [] { const char* e = std::getenv("TBC_SWEEP_WG_RING"); return e && e[0] == '1'; }();
  static const bool fpr = [] { const char* e = std::getenv("TBC_SWEEP_WG_FP"); return e && e[0] == '1'; }();
  // TBC_SWEEP_WG_COMPACT=1: a sub-round of more than two passes takes 512 CHILDREN a pass, not 512 (config, call) slots (81 KB of LDS: still
  // two workgroups per CU; with or without the fingerprint; not with the ring).  The same standing.
  // TBC_SWEEP_WG_COMPACT=2: ... and a pass that fits one wavefront is run by wavefront 0 alone (one workgroup barrier instead of three)
  static const int compact = [] { const char* e = std::getenv("TBC_SWEEP_WG_COMPACT"); return (e && (e[0] == '1' || e[0] == '2')) ? e[0] - '0' : 0; }();
  if (compact == 1 && !ring && waves == 8) return fpr ? launch_compact<kSweepCapMid, 8, true, false>(a, s) : launch_compact<kSweepCapMid, 8, false, false>(a, s);
  if (compact == 2 && !ring && waves == 8) return fpr ? launch_compact<kSweepCapMid, 8, true, true>(a, s) : launch_compact<kSweepCapMid, 8, false, true>(a, s);
  if (ring && waves == 8) return fpr ? launch_ring<kSweepCapMid, 8, true>(a, s) : launch_ring<kSweepCapMid, 8, false>(a, s);
  if (fpr && waves == 8) return launch_fp<kSweepCapMid, 8>(a, s);
  if (waves == 4) return launch_one<kSweepCapMid, 4>(a, s);
  if (waves == 8) return launch_one<kSweepCapMid, 8>(a, s);
  // TBC_SWEEP_WG=16 (experimental, emulator-verified only): sixteen wavefronts on the big sets from the start -- no second pass,
  // 154 KB of LDS, one workgroup per CU
  if (waves == 16) return launch_one<kSweepCapBig, 16>(a, s);
  return false;
}

}  // namespace tbc
