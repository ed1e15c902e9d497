// batch_shard.hip -- ONE history (or a small batch) over several GPUs: the level sweep's wavefronts dealt to the ranks, one exchange of
// relation tables, every rank composes (include/tbcheck.h, "one history over several GPUs"; jepsen-tigerbeetle_amd/shard.py drives these
// over torch.distributed, csrc/tbc_comm.hip over RCCL from any host language).
#include "tbc_batch.h"

using namespace tbc;

extern "C" {

tbc_status tbc_batch_set_shard(tbc_batch* b, uint32_t rank, uint32_t world) {
  if (!b || world == 0 || rank >= world) { set_error("tbc_batch_set_shard: bad rank / world"); return TBC_ERR_INVALID_ARG; }
  if (!b->sweep) { set_error("tbc_batch_set_shard: this batch does not run the level sweep (TBC_ALG_LINEAR, <= 64 process slots)"); return TBC_ERR_UNSUPPORTED; }
  b->shard_rank = rank; b->shard_world = world;
  return TBC_OK;
}

tbc_status tbc_batch_sweep_partial(tbc_batch* b) {
  if (!b || !b->sweep) { set_error("tbc_batch_sweep_partial: not a sweep batch"); return TBC_ERR_INVALID_ARG; }
  try { return batch_run_impl(b, nullptr, 1); }
  catch (const std::bad_alloc&) { set_error("host allocation failed"); return TBC_ERR_OOM; }
  catch (...) { set_error("unexpected exception"); return TBC_ERR_HIP; }
}

tbc_status tbc_batch_sweep_table(const tbc_batch* b, void** device_ptr, uint64_t* bytes) {
  if (!b || !b->sweep || !device_ptr || !bytes) { set_error("tbc_batch_sweep_table: not a sweep batch"); return TBC_ERR_INVALID_ARG; }
  *device_ptr = b->d_sres.p;
  *bytes = (uint64_t)b->seg_host.size() * sizeof(SegResult);
  return TBC_OK;
}

tbc_status tbc_batch_sweep_finish(tbc_batch* b, const void* merged, uint64_t merged_bytes, tbc_result* results) {
  if (!b || !b->sweep || !merged) { set_error("tbc_batch_sweep_finish: not a sweep batch"); return TBC_ERR_INVALID_ARG; }
  if (merged_bytes != (uint64_t)b->seg_host.size() * sizeof(SegResult)) {
    set_error("tbc_batch_sweep_finish: merged table is %llu bytes, this batch's table is %llu (tbc_batch_sweep_table)",
              (unsigned long long)merged_bytes, (unsigned long long)(b->seg_host.size() * sizeof(SegResult)));
    return TBC_ERR_INVALID_ARG;
  }
  if (!b->partial_done) { set_error("tbc_batch_sweep_finish without tbc_batch_sweep_partial"); return TBC_ERR_INVALID_ARG; }
  try {
    std::memcpy(b->seg_host.data(), merged, b->seg_host.size() * sizeof(SegResult));
    return batch_run_impl(b, results, 2);
  }
  catch (const std::bad_alloc&) { set_error("host allocation failed"); return TBC_ERR_OOM; }
  catch (...) { set_error("unexpected exception"); return TBC_ERR_HIP; }
}

namespace {
// bitwise OR of `world` relation tables lying back to back in device memory into `dst` (every record is written by exactly
// one rank and all zero on the others)
__global__ void sweep_or_kernel(uint64_t* dst, const uint64_t* gathered, uint64_t words, uint32_t world) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= words) return;
  uint64_t v = 0;
  for (uint32_t r = 0; r < world; r++) v |= gathered[(uint64_t)r * words + i];
  dst[i] = v;
}
}  // namespace

tbc_status tbc_batch_sweep_merge(tbc_batch* b, const void* gathered_device, uint64_t gathered_bytes, uint32_t world, tbc_result* results) {
  if (!b || !b->sweep || !gathered_device || world == 0) { set_error("tbc_batch_sweep_merge: not a sweep batch"); return TBC_ERR_INVALID_ARG; }
  const uint64_t bytes = (uint64_t)b->seg_host.size() * sizeof(SegResult);
  if (gathered_bytes != bytes * world) {
    set_error("tbc_batch_sweep_merge: %llu bytes gathered, %u tables of %llu expected", (unsigned long long)gathered_bytes, world, (unsigned long long)bytes);
    return TBC_ERR_INVALID_ARG;
  }
  if (!b->partial_done) { set_error("tbc_batch_sweep_merge without tbc_batch_sweep_partial"); return TBC_ERR_INVALID_ARG; }
  try {
    HIP_TRY(hipSetDevice(b->device));
    const uint64_t words = bytes / 8;
    hipLaunchKernelGGL(sweep_or_kernel, dim3((uint32_t)((words + 255) / 256)), dim3(256), 0, b->stream, (uint64_t*)b->d_sres.p, (const uint64_t*)gathered_device, words, world);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(b->seg_host.data(), b->d_sres.p, bytes, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return batch_run_impl(b, results, 2);
  }
  catch (const std::bad_alloc&) { set_error("host allocation failed"); return TBC_ERR_OOM; }
  catch (...) { set_error("unexpected exception"); return TBC_ERR_HIP; }
}

}  // extern "C"
