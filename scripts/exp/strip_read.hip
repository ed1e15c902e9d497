// strip_read.hip -- what bounds set-full's streaming pass?  The same bytes read three ways (experiment of round 6, scripts/r06_calls/x.sh):
//   strip : a workgroup reads a 4 KB-wide column strip of ROWS consecutive rows (row pitch 32 KB), 8 x 16 B in flight per lane -- setfull_any_kernel's pattern
//   wide  : a workgroup reads W x 4 KB contiguous of each of its rows (W column groups per thread)
//   linear: a workgroup reads its share of the matrix as one contiguous block
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <time.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <bool NT> __device__ __forceinline__ uint4 ld(const uint4* p) {
  if constexpr (NT) { uint4 v; v.x = __builtin_nontemporal_load(&p->x); v.y = __builtin_nontemporal_load(&p->y); v.z = __builtin_nontemporal_load(&p->z); v.w = __builtin_nontemporal_load(&p->w); return v; }
  else return *p;
}

// grid = (chunks, col blocks of 1024 words); W column groups per thread: the workgroup's piece of a row is W * 4 KB, col blocks = WPR / (1024 W)
template <int W, int U, bool NT>
__global__ __launch_bounds__(256) void strip_kernel(const uint32_t* __restrict__ M, uint32_t WPR, uint32_t rows, uint32_t* out) {
  const uint32_t c = blockIdx.x, j = blockIdx.y;
  const uint32_t r0 = c * rows;
  uint4 acc[W];
#pragma unroll
  for (int w = 0; w < W; w++) acc[w] = make_uint4(0, 0, 0, 0);
  const uint32_t w0 = j * 1024u * W + threadIdx.x * 4u;
  for (uint32_t r = 0; r < rows; r += U) {
    uint4 v[U][W];
#pragma unroll
    for (int q = 0; q < U; q++)
#pragma unroll
      for (int w = 0; w < W; w++) v[q][w] = ld<NT>(reinterpret_cast<const uint4*>(M + (uint64_t)(r0 + r + q) * WPR + w0 + 1024u * w));
#pragma unroll
    for (int q = 0; q < U; q++)
#pragma unroll
      for (int w = 0; w < W; w++) { acc[w].x |= v[q][w].x; acc[w].y |= v[q][w].y; acc[w].z |= v[q][w].z; acc[w].w |= v[q][w].w; }
  }
  uint32_t o = 0;
#pragma unroll
  for (int w = 0; w < W; w++) o |= acc[w].x | acc[w].y | acc[w].z | acc[w].w;
  if (o == 0x12345678u) out[0] = o;
}


// set-full's triangle: column block j of chunk c counts when (c + 1) / chunks > j / 8; the other workgroups return at once
template <int U, bool NT, bool REV>
__global__ __launch_bounds__(256) void tri_kernel(const uint32_t* __restrict__ M, uint32_t WPR, uint32_t rows, uint32_t* out) {
  const uint32_t j = REV ? gridDim.y - 1u - blockIdx.y : blockIdx.y;
  const uint32_t c = REV ? (blockIdx.x + j * gridDim.x / gridDim.y) % gridDim.x : blockIdx.x;
  if ((c + 1u) * gridDim.y <= j * gridDim.x) return;
  const uint32_t r0 = c * rows;
  uint4 acc = make_uint4(0, 0, 0, 0);
  const uint32_t w0 = j * 1024u + threadIdx.x * 4u;
  for (uint32_t r = 0; r < rows; r += U) {
    uint4 v[U];
#pragma unroll
    for (int q = 0; q < U; q++) v[q] = ld<NT>(reinterpret_cast<const uint4*>(M + (uint64_t)(r0 + r + q) * WPR + w0));
#pragma unroll
    for (int q = 0; q < U; q++) { acc.x |= v[q].x; acc.y |= v[q].y; acc.z |= v[q].z; acc.w |= v[q].w; }
  }
  const uint32_t o = acc.x | acc.y | acc.z | acc.w;
  if (o == 0x12345678u) out[0] = o;
}

// the triangle again, with the two summary words per (word column, chunk) written as setfull_any_kernel writes them.  ST: 0 nothing;
// 1 column-major [word column][chunk] (4 B stores 4 * CHP apart), the workgroups below the diagonal write zeros; 2 column-major, those
// write nothing; 3 chunk-major [chunk][word column] (16 B stores, a wavefront 1 KB)
template <int U, bool NT, int ST>
__global__ __launch_bounds__(256) void tri_store_kernel(const uint32_t* __restrict__ M, uint32_t WPR, uint32_t rows, uint32_t* any_p, uint32_t* any_a) {
  const uint32_t j = blockIdx.y, c = blockIdx.x, CHP = gridDim.x;
  const uint32_t w0 = j * 1024u + threadIdx.x * 4u;
  uint4 acc = make_uint4(0, 0, 0, 0), nac = make_uint4(~0u, ~0u, ~0u, ~0u);
  const bool active = !((c + 1u) * gridDim.y <= j * gridDim.x);
  if (!active && ST != 1) return;
  if (active) {
    const uint32_t r0 = c * rows;
    for (uint32_t r = 0; r < rows; r += U) {
      uint4 v[U];
#pragma unroll
      for (int q = 0; q < U; q++) v[q] = ld<NT>(reinterpret_cast<const uint4*>(M + (uint64_t)(r0 + r + q) * WPR + w0));
#pragma unroll
      for (int q = 0; q < U; q++) { acc.x |= v[q].x; acc.y |= v[q].y; acc.z |= v[q].z; acc.w |= v[q].w; nac.x &= v[q].x; nac.y &= v[q].y; nac.z &= v[q].z; nac.w &= v[q].w; }
    }
  }
  if (ST == 1 || ST == 2) {
    any_p[(uint64_t)(w0 + 0) * CHP + c] = acc.x; any_p[(uint64_t)(w0 + 1) * CHP + c] = acc.y; any_p[(uint64_t)(w0 + 2) * CHP + c] = acc.z; any_p[(uint64_t)(w0 + 3) * CHP + c] = acc.w;
    any_a[(uint64_t)(w0 + 0) * CHP + c] = ~nac.x; any_a[(uint64_t)(w0 + 1) * CHP + c] = ~nac.y; any_a[(uint64_t)(w0 + 2) * CHP + c] = ~nac.z; any_a[(uint64_t)(w0 + 3) * CHP + c] = ~nac.w;
  } else if (ST == 3) {
    *reinterpret_cast<uint4*>(any_p + (uint64_t)c * WPR + w0) = acc;
    *reinterpret_cast<uint4*>(any_a + (uint64_t)c * WPR + w0) = make_uint4(~nac.x, ~nac.y, ~nac.z, ~nac.w);
  } else if ((acc.x | acc.y | acc.z | acc.w | nac.x) == 0x12345678u) any_p[0] = 1;
}

template <int U, bool NT>
__global__ __launch_bounds__(256) void linear_kernel(const uint32_t* __restrict__ M, uint64_t words_per_wg, uint32_t* out) {
  const uint4* p = reinterpret_cast<const uint4*>(M + (uint64_t)blockIdx.x * words_per_wg) + threadIdx.x;
  const uint64_t n = words_per_wg / 4 / 256;
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (uint64_t i = 0; i < n; i += U) {
    uint4 v[U];
#pragma unroll
    for (int q = 0; q < U; q++) v[q] = ld<NT>(p + (i + q) * 256);
#pragma unroll
    for (int q = 0; q < U; q++) { acc.x |= v[q].x; acc.y |= v[q].y; acc.z |= v[q].z; acc.w |= v[q].w; }
  }
  const uint32_t o = acc.x | acc.y | acc.z | acc.w;
  if (o == 0x12345678u) out[0] = o;
}

template <class F> static float time_it(F f, int reps = 7) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e9f;
  for (int i = 0; i < reps; i++) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (i && ms < best) best = ms; }
  return best;
}

int main() {
  const uint32_t WPR = 8192, R = 16384;                  // 512 MiB: what set-full's pass reads of its 1 GiB matrix
  const uint64_t words = (uint64_t)WPR * R, bytes = words * 4;
  uint32_t *M, *out; CK(hipMalloc(&M, bytes)); CK(hipMalloc(&out, 4)); CK(hipMemset(M, 0, bytes));
#define REPORT(name, ...) { const float ms = time_it([&] { __VA_ARGS__; }); printf("%-40s %.3f ms  %.2f TB/s\n", name, ms, bytes / ms / 1e9); }
  for (uint32_t rows : {64u, 128u, 256u}) {
    printf("rows per workgroup %u (strip: %u workgroups)\n", rows, R / rows * 8);
    REPORT("strip W=1 U=8", strip_kernel<1, 8, false><<<dim3(R / rows, 8), dim3(256)>>>(M, WPR, rows, out));
    REPORT("strip W=1 U=8 nt", strip_kernel<1, 8, true><<<dim3(R / rows, 8), dim3(256)>>>(M, WPR, rows, out));
    REPORT("strip W=1 U=16", strip_kernel<1, 16, false><<<dim3(R / rows, 8), dim3(256)>>>(M, WPR, rows, out));
    REPORT("strip W=1 U=4", strip_kernel<1, 4, false><<<dim3(R / rows, 8), dim3(256)>>>(M, WPR, rows, out));
    REPORT("wide  W=2 U=4", strip_kernel<2, 4, false><<<dim3(R / rows, 4), dim3(256)>>>(M, WPR, rows, out));
    REPORT("wide  W=2 U=4 nt", strip_kernel<2, 4, true><<<dim3(R / rows, 4), dim3(256)>>>(M, WPR, rows, out));
    REPORT("wide  W=4 U=2", strip_kernel<4, 2, false><<<dim3(R / rows, 2), dim3(256)>>>(M, WPR, rows, out));
    REPORT("wide  W=4 U=4", strip_kernel<4, 4, false><<<dim3(R / rows, 2), dim3(256)>>>(M, WPR, rows, out));
    REPORT("wide  W=8 U=2", strip_kernel<8, 2, false><<<dim3(R / rows, 1), dim3(256)>>>(M, WPR, rows, out));
    REPORT("wide  W=8 U=2 nt", strip_kernel<8, 2, true><<<dim3(R / rows, 1), dim3(256)>>>(M, WPR, rows, out));
  }
  for (uint32_t wgs : {512u, 1024u, 2048u, 4096u}) {
    printf("linear, %u workgroups\n", wgs);
    REPORT("linear U=8", linear_kernel<8, false><<<dim3(wgs), dim3(256)>>>(M, words / wgs, out));
    REPORT("linear U=8 nt", linear_kernel<8, true><<<dim3(wgs), dim3(256)>>>(M, words / wgs, out));
    REPORT("linear U=16", linear_kernel<16, false><<<dim3(wgs), dim3(256)>>>(M, words / wgs, out));
  }
  {
    const uint32_t R2 = 32768; const uint64_t bytes2 = (uint64_t)R2 * WPR * 4;
    uint32_t* M2; CK(hipMalloc(&M2, bytes2));
    for (int fill : {0, 0xA5}) {
      CK(hipMemset(M2, fill, bytes2));
      for (uint32_t chunks : {128u, 256u, 512u}) {
        const uint32_t rows = R2 / chunks;
        uint64_t act = 0; for (uint32_t c = 0; c < chunks; c++) for (uint32_t j = 0; j < 8; j++) if (!((c + 1) * 8 <= j * chunks)) act++;
        const uint64_t bytes = act * rows * 4096ull;
        printf("triangle of a 1 GiB matrix (fill 0x%02X), %u chunks of %u rows: %llu active workgroups, %.0f MB\n", fill, chunks, rows, (unsigned long long)act, bytes / 1e6);
        REPORT("tri U=8", tri_kernel<8, false, false><<<dim3(chunks, 8), dim3(256)>>>(M2, WPR, rows, out));
        REPORT("tri U=8 nt", tri_kernel<8, true, false><<<dim3(chunks, 8), dim3(256)>>>(M2, WPR, rows, out));
        REPORT("tri U=8 nt, diagonal first", tri_kernel<8, true, true><<<dim3(chunks, 8), dim3(256)>>>(M2, WPR, rows, out));
        REPORT("tri U=4 nt", tri_kernel<4, true, false><<<dim3(chunks, 8), dim3(256)>>>(M2, WPR, rows, out));
      }
    }

    {
      uint32_t *ap, *aa; CK(hipMalloc(&ap, 512ull * WPR * 4)); CK(hipMalloc(&aa, 512ull * WPR * 4));
      CK(hipMemset(ap, 0, 512ull * WPR * 4)); CK(hipMemset(aa, 0, 512ull * WPR * 4));
      for (uint32_t chunks : {256u}) {
        const uint32_t rows = R2 / chunks;
        const uint64_t bytes = 604000000ull;
        printf("triangle + summaries, %u chunks\n", chunks);
        REPORT("no stores", tri_store_kernel<8, true, 0><<<dim3(chunks, 8), dim3(256)>>>(M2, WPR, rows, ap, aa));
        REPORT("column-major, zeros below the diagonal", tri_store_kernel<8, true, 1><<<dim3(chunks, 8), dim3(256)>>>(M2, WPR, rows, ap, aa));
        REPORT("column-major, active only", tri_store_kernel<8, true, 2><<<dim3(chunks, 8), dim3(256)>>>(M2, WPR, rows, ap, aa));
        REPORT("chunk-major 16 B stores", tri_store_kernel<8, true, 3><<<dim3(chunks, 8), dim3(256)>>>(M2, WPR, rows, ap, aa));
      }
    }
    // one launch from an idle device (what a caller's single scan sees): sleep, then one timed launch
    for (int i = 0; i < 3; i++) {
      CK(hipDeviceSynchronize()); 
      struct timespec ts = {0, 200000000}; nanosleep(&ts, nullptr);
      hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
      CK(hipEventRecord(a)); tri_kernel<8, true, false><<<dim3(256, 8), dim3(256)>>>(M2, WPR, 128, out); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); printf("tri U=8 nt, 256 chunks, one launch after 200 ms idle: %.3f ms\n", ms);
    }
  }
  return 0;

}
