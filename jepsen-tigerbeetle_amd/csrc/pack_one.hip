// pack_one.hip -- K1 for one history (or a handful): pack by a workgroup's sixteen wavefronts (gfx950); the body is pack_one_impl.h.
// Same inputs, same bytes left behind as pack_kernel (pack.hip), which keeps the batches and the models this body does not take.
// STANDING: the body is verified under the workgroup emulator (tests/test_pack_one_emu.py) and had not run on the device when it was
// committed; tbc_api.hip takes it only under TBC_PACK_ONE=1 (bench.py's extra.single_history_forms measures it).
#include <hip/hip_runtime.h>
#include "tbc_internal.h"
#include "pack_one_impl.h"

namespace tbc {

namespace {
__global__ __launch_bounds__(64 * packone::kNW) void pack_one_kernel(PackArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  packone::history(A, lds);
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

}  // namespace tbc
