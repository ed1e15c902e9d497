// wgl_narrow.hip -- K5n: the search with several histories per wavefront (gfx950); the body is wgl_narrow_impl.h.
//
// Launch: 4 wavefronts per workgroup, each with its own slice of LDS and its own H = 64 / L histories (work items
// w * H .. w * H + H - 1); wavefronts never talk to each other.  Register family (register, cas-register, mutex) only:
// the models that step on immediates and have the dominance rules that make a narrow round enough.
#include <hip/hip_runtime.h>
#include "tbc_internal.h"
#include "wgl_narrow_impl.h"

namespace tbc {

namespace {

constexpr uint32_t kNarrowWaves = 2;

#ifndef TBC_NARROW_MIN_WAVES
#define TBC_NARROW_MIN_WAVES 4
#endif

template <int MW, int L, bool CF, bool CNT = false>
__global__ __launch_bounds__(64 * kNarrowWaves, CNT ? 2 : TBC_NARROW_MIN_WAVES) void wgl_narrow_kernel(BeamArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wv_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t w = blockIdx.x * kNarrowWaves + wv_;
  narrow::narrow_wave<MW, L, CF, CNT>(A, w, lds + wv_ * narrow::narrow_lds_words(MW, L, CF, CNT), lane);
}
// Wavefronts the GPU keeps resident at once: the launch is sized to that, not to the batch -- a wavefront's groups take more
// histories off the queue as they finish (BeamArgs.next_work), so no wavefront starts late into a half-empty machine and a
// group whose history was short does not idle until its seven neighbours are done.
uint32_t resident_waves(size_t lds_bytes_per_wave, uint32_t waves_per_simd) {
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  uint32_t per_cu = 4u * ((waves_per_simd && waves_per_simd < TBC_NARROW_MIN_WAVES) ? waves_per_simd : TBC_NARROW_MIN_WAVES);
  const uint32_t by_lds = (uint32_t)((160u * 1024u) / (lds_bytes_per_wave ? lds_bytes_per_wave : 1));
  if (by_lds < per_cu) per_cu = by_lds;
  per_cu = per_cu / kNarrowWaves * kNarrowWaves;
  if (per_cu == 0) per_cu = kNarrowWaves;
  return (uint32_t)cus * per_cu;
}

template <int MW, int L, bool CF, bool CNT = false>
void launch_cf(const BeamArgs& a_in, hipStream_t s, uint32_t wps) {
  const uint32_t H = 64u / L;
  const size_t lds_wave = (size_t)narrow::narrow_lds_words(MW, L, CF, CNT) * 4;
  uint32_t waves = (a_in.n_work + H - 1) / H;
  const uint32_t fit = resident_waves(lds_wave, wps);
  if (waves > fit) waves = fit;
  const uint32_t blocks = (waves + kNarrowWaves - 1) / kNarrowWaves;
  BeamArgs a = a_in;
  a.first_dynamic = blocks * kNarrowWaves * H;          // what the launch deals out; the queue hands out the rest
  (void)hipMemsetAsync(a.next_work, 0, sizeof(unsigned int), s);
  hipLaunchKernelGGL((wgl_narrow_kernel<MW, L, CF, CNT>), dim3(blocks), dim3(64 * kNarrowWaves), lds_wave * kNarrowWaves, s, a);
}

template <int MW, int L>
void launch_one(const BeamArgs& a, hipStream_t s, uint32_t wps) {
  if constexpr (MW <= 2 && L >= 8) {
    if (a.rules & kRuleCount) {          // the count form (crashed calls as counts per class): one or two mask words, 8 / 16 / 32 lanes per history
      if constexpr (MW == 1) { if (a.front_words == kFrontCompactWords) { launch_cf<1, L, true, true>(a, s, wps); return; } }
      launch_cf<MW, L, false, true>(a, s, wps);
      return;
    }
  }
  if constexpr (MW == 1) {
    if (a.front_words == kFrontCompactWords) { launch_cf<1, L, true>(a, s, wps); return; }
  }
  launch_cf<MW, L, false>(a, s, wps);
}

}  // namespace

bool narrow_supported(uint32_t mask_words, uint32_t lanes) {
  return (mask_words == 1 || mask_words == 2 || mask_words == 4) && (lanes == 4 || lanes == 8 || lanes == 16 || lanes == 32);
}

bool launch_narrow(const BeamArgs& a, uint32_t mask_words, uint32_t lanes, void* stream, uint32_t waves_per_simd) {
  hipStream_t s = (hipStream_t)stream;
#define NARROW_CASE(MWV, LV) if (mask_words == MWV && lanes == LV) { launch_one<MWV, LV>(a, s, waves_per_simd); return true; }
  NARROW_CASE(1, 4) NARROW_CASE(1, 8) NARROW_CASE(1, 16) NARROW_CASE(1, 32)
  NARROW_CASE(2, 4) NARROW_CASE(2, 8) NARROW_CASE(2, 16) NARROW_CASE(2, 32)
  NARROW_CASE(4, 4) NARROW_CASE(4, 8) NARROW_CASE(4, 16) NARROW_CASE(4, 32)
#undef NARROW_CASE
  return false;
}

}  // namespace tbc
