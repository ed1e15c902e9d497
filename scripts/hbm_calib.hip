// hbm_calib.hip -- known-byte random-access microbenchmarks to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE for the
// access patterns of the search kernel (MI355X_MICROARCH.md, HBM section: "calibrate for your pattern"):
//   read64   every lane reads one whole random 64 B bucket (four 16 B loads, sc1) -- the visited-set probe
//   write16  every lane writes 16 B at a random 64 B-aligned place               -- a new config's key
//   write8   every lane writes 8 B at a random 8 B-aligned place                 -- a parent link
//   cas8     every lane does a 64-bit CAS at a random 64 B-aligned place         -- the claim of an entry
//   stream   coalesced 16 B per lane copy                                         -- the control
// Known bytes per kernel are printed; run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes)
// and divide.  Build: hipcc --offload-arch=gfx950 -O3 -o hbm_calib scripts/hbm_calib.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x ^= x >> 33; x *= 0xFF51AFD7ED558CCDull; x ^= x >> 33; x *= 0xC4CEB9FE1A85EC53ull; x ^= x >> 33; return x;
}

__global__ void read64(const uint8_t* buf, uint64_t n_lines, uint32_t iters, uint32_t* sink) {
  const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (uint32_t i = 0; i < iters; i++) {
    const uint64_t line = mix(t * 0x9E3779B97F4A7C15ull + i) % n_lines;
    const u32x4* p = reinterpret_cast<const u32x4*>(buf + line * 64);
    u32x4 a = __builtin_nontemporal_load(p), b = __builtin_nontemporal_load(p + 1), c = __builtin_nontemporal_load(p + 2), d = __builtin_nontemporal_load(p + 3);
    acc += a.x ^ b.y ^ c.z ^ d.w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void write16(uint8_t* buf, uint64_t n_lines, uint32_t iters) {
  const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  for (uint32_t i = 0; i < iters; i++) {
    const uint64_t line = mix(t * 0x9E3779B97F4A7C15ull + i + 77) % n_lines;
    *reinterpret_cast<u32x4*>(buf + line * 64) = u32x4{(uint32_t)t, i, 1u, 2u};
  }
}
__global__ void write8(uint8_t* buf, uint64_t n_lines, uint32_t iters) {
  const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  for (uint32_t i = 0; i < iters; i++) {
    const uint64_t w = mix(t * 0x9E3779B97F4A7C15ull + i + 99) % (n_lines * 8);
    *reinterpret_cast<uint64_t*>(buf + w * 8) = t + i;
  }
}
__global__ void cas8(uint8_t* buf, uint64_t n_lines, uint32_t iters) {
  const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  for (uint32_t i = 0; i < iters; i++) {
    const uint64_t line = mix(t * 0x9E3779B97F4A7C15ull + i + 5) % n_lines;
    unsigned long long* p = reinterpret_cast<unsigned long long*>(buf + line * 64);
    atomicCAS(p, 0ull, (unsigned long long)(t + 1));
  }
}
__global__ void stream(const u32x4* __restrict__ src, u32x4* __restrict__ dst, uint64_t n) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

int main(int argc, char** argv) {
  const uint64_t bytes = 8ull << 30, n_lines = bytes / 64;
  const uint32_t blocks = 8192, threads = 256, iters = 64;
  uint8_t *a, *b; uint32_t* sink;
  if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { std::puts("alloc failed"); return 1; }
  hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
  const double lanes = (double)blocks * threads * iters;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms;
#define RUN(name, known, ...) do { hipEventRecord(e0); __VA_ARGS__; hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); \
    std::printf("%-8s known_bytes %.6g  time_ms %.3f  GB/s %.1f\n", name, (double)(known), ms, (double)(known) / ms / 1e6); } while (0)
  for (int rep = 0; rep < 2; rep++) {
    RUN("read64", lanes * 64, hipLaunchKernelGGL(read64, dim3(blocks), dim3(threads), 0, 0, a, n_lines, iters, sink));
    RUN("write16", lanes * 16, hipLaunchKernelGGL(write16, dim3(blocks), dim3(threads), 0, 0, b, n_lines, iters));
    RUN("write8", lanes * 8, hipLaunchKernelGGL(write8, dim3(blocks), dim3(threads), 0, 0, b, n_lines, iters));
    RUN("cas8", lanes * 8, hipLaunchKernelGGL(cas8, dim3(blocks), dim3(threads), 0, 0, b, n_lines, iters));
    RUN("stream", (double)bytes * 2, hipLaunchKernelGGL(stream, dim3(8192), dim3(256), 0, 0, (const u32x4*)a, (u32x4*)b, bytes / 16));
  }
  return 0;
}
