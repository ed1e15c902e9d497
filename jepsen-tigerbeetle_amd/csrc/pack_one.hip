// pack_one.hip -- K1 by a workgroup's wavefronts with its tables in LDS (gfx950); the body is pack_one_impl.h.  Two forms:
//   pack_one_kernel  one history (or a handful): sixteen wavefronts, same inputs and same bytes left behind as pack_kernel (pack.hip);
//   pack_one_counts_kernel  ... and open_counts_kernel's bytes in the same pass (one launch fewer per tbc_check);
//   pack_wg_kernel   a batch: four wavefronts per history, pack_kernel's AND open_counts_kernel's (pack_open.hip) bytes in one pass --
//                    the ranks never leave the registers between the two.
// pack_kernel + open_counts_kernel keep the models and the histories these bodies do not take.
//   pack_wg64_kernel the same for histories of at most 64 process slots: 19 KB of LDS instead of 31, eight workgroups per CU.
// Verified word for word under the workgroup emulator (tests/test_pack_one_emu.py) and on the device through every GPU parity test:
// tbc_api.hip takes pack_one_counts_kernel for one history and pack_wg(64)_kernel for a batch whenever the histories fit.
#include <hip/hip_runtime.h>
#include "tbc_internal.h"
#include "pack_one_impl.h"

namespace tbc {

namespace {
__global__ __launch_bounds__(64 * packone::OneGeo::kNW) void pack_one_kernel(PackArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const PackOpenArgs none{};
  packone::history<packone::OneGeo>(A, none, lds);
}
__global__ __launch_bounds__(64 * packone::OneCountsGeo::kNW) void pack_one_counts_kernel(PackArgs A, PackOpenArgs O) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  packone::history<packone::OneCountsGeo>(A, O, lds);
}
__global__ __launch_bounds__(64 * packone::BatchGeo::kNW) void pack_wg_kernel(PackArgs A, PackOpenArgs O) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[packone::BatchGeo::lds_words()];
  packone::history<packone::BatchGeo>(A, O, lds);
}
__global__ __launch_bounds__(64 * packone::Batch64Geo::kNW) void pack_wg64_kernel(PackArgs A, PackOpenArgs O) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[packone::Batch64Geo::lds_words()];
#ifdef TBC_PACK_LDS_PAD          // (a profiling build: fewer workgroups a CU -- does a workgroup get faster when fewer share the memory system?)
  __shared__ uint32_t lds_pad[TBC_PACK_LDS_PAD];
  if (A.n_hist == 0xFFFFFFFFu) lds_pad[threadIdx.x] = 1u;
  if (A.n_hist == 0xFFFFFFFEu) lds[0] = lds_pad[threadIdx.x ^ 1u];
#endif
  packone::history<packone::Batch64Geo>(A, O, lds);
}
}  // namespace

bool pack_one_fits(uint32_t model_kind, uint32_t n_ops, uint32_t n_events, uint32_t n_slots) {
  return packone::fits(model_kind, n_ops, n_events, n_slots);
}

// histories [a.h0, a.n_hist), one workgroup each (103 KB of LDS: one workgroup per CU); false = not launched (the caller takes pack_kernel)
bool launch_pack_one(const PackArgs& a, void* stream) {
  constexpr uint32_t bytes = packone::lds_words() * 4;
  static bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&pack_one_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
  if (!ok || a.n_hist <= a.h0) return false;
  hipLaunchKernelGGL(pack_one_kernel, dim3(a.n_hist - a.h0), dim3(64 * packone::kNW), bytes, (hipStream_t)stream, a);
  return true;
}

// the same with open_counts_kernel's tables left behind too (138 KB of LDS); o and the walk that follows as for launch_pack_wg below
bool pack_one_counts_fits(uint32_t model_kind, uint32_t n_ops, uint32_t n_events, uint32_t n_slots) {
  return packone::OneCountsGeo::fits(model_kind, n_ops, n_events, n_slots);
}
bool launch_pack_one_counts(const PackArgs& a, const PackOpenArgs& o, void* stream) {
  constexpr uint32_t bytes = packone::OneCountsGeo::lds_words() * 4;
  static bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&pack_one_counts_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
  if (!ok || a.n_hist <= a.h0 || o.h0 != a.h0 || o.n_hist != a.n_hist || !o.bh || !o.off || !o.ncr || !o.slot8 || (o.branch_lists && !o.rk8)) return false;
  hipLaunchKernelGGL(pack_one_counts_kernel, dim3(a.n_hist - a.h0), dim3(64 * packone::OneCountsGeo::kNW), bytes, (hipStream_t)stream, a, o);
  return true;
}

bool pack_wg_fits(uint32_t model_kind, uint32_t n_ops, uint32_t n_events, uint32_t n_slots) {
  return packone::BatchGeo::fits(model_kind, n_ops, n_events, n_slots);
}
bool pack_wg64_fits(uint32_t model_kind, uint32_t n_ops, uint32_t n_events, uint32_t n_slots) {
  return packone::Batch64Geo::fits(model_kind, n_ops, n_events, n_slots);
}

// histories [a.h0, a.n_hist), one workgroup of four wavefronts each (31 KB of LDS, static): pack + open counts.  The caller has
// asked pack_wg_fits() of every history, o is what launch_pack_open() would hand open_counts_kernel (same h0 / n_hist), and the
// walk that follows is launched with skip_counts.  slots64: every history also passed pack_wg64_fits() (19 KB of LDS).  false = not launched.
#ifdef TBC_PACK_PROF
// (a profiling build only, scripts/build_variant.sh: the per-phase ticks go to 64 words of DEVICE memory -- atomics on the host-mapped debug
// words clogged the very thing that was being measured -- and tbc_pack_prof_read brings them back)
static uint32_t* g_pack_prof = nullptr;
extern "C" int tbc_pack_prof_read(uint32_t* out) {
  if (!g_pack_prof) return 0;
  return hipMemcpy(out, g_pack_prof, 64 * 4, hipMemcpyDeviceToHost) == hipSuccess ? 1 : 0;
}
#endif
bool launch_pack_wg(const PackArgs& a_in, const PackOpenArgs& o, void* stream, bool slots64) {
  PackArgs a = a_in;
#ifdef TBC_PACK_PROF
  if (!g_pack_prof && hipMalloc((void**)&g_pack_prof, 64 * 4) == hipSuccess) (void)hipMemset(g_pack_prof, 0, 64 * 4);
  a.dbg = g_pack_prof;
#endif
  if (a.n_hist <= a.h0 || o.h0 != a.h0 || o.n_hist != a.n_hist || !o.bh || !o.off || !o.ncr || !o.slot8 || (o.branch_lists && !o.rk8)) return false;
  if (slots64) hipLaunchKernelGGL(pack_wg64_kernel, dim3(a.n_hist - a.h0), dim3(64 * packone::Batch64Geo::kNW), 0, (hipStream_t)stream, a, o);
  else hipLaunchKernelGGL(pack_wg_kernel, dim3(a.n_hist - a.h0), dim3(64 * packone::BatchGeo::kNW), 0, (hipStream_t)stream, a, o);
  return true;
}

}  // namespace tbc
