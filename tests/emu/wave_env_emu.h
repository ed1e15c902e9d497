// TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  wave_env_emu.h: the primitives of csrc/wave_env.h for a lane-accurate HOST
// emulation of one gfx950 wavefront, so that a kernel body written against them (csrc/wgl_narrow_impl.h) can be run and
// compared with the oracle on a machine without a GPU.  Only tests/emu/ defines TBC_EMU; the product never sees this.
//
// One wavefront = 64 fibers (a dozen lines of x86-64 stack switching; ucontext's signal-mask system calls made a
// 10k-op history take minutes), one per lane, run round-robin by a scheduler.  Every cross-lane primitive is a
// RENDEZVOUS: a lane deposits its value and yields; when all live lanes have arrived the scheduler snapshots the
// deposits and resumes them, and each computes its result from the snapshot.  The kernel bodies keep their cross-lane
// operations in wave-uniform control flow (as a GPU kernel should); the emulator CHECKS that: every lane must arrive at
// the same primitive (a site id from __COUNTER__ / line) or the run aborts with both sites named.  Between two
// rendezvous lanes run one after the other (lane 0 first), so code that lets one lane read what another lane writes
// without a barrier in between shows up as a wrong answer here rather than as a GPU heisenbug.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#ifndef __host__
#define __host__
#endif
#ifndef __device__
#define __device__
#endif
#define WV_DEV static inline
#define WV_MEM inline
#define WV_HD
#define WV_GLOBAL
#define WV_LDS
#define WV_UNROLL
#define WV_NOUNROLL

namespace wv {

struct u32x4 { uint32_t x, y, z, w; };
typedef uint32_t u32x16 __attribute__((vector_size(64)));
typedef uint32_t u32x32 __attribute__((vector_size(128)));
typedef uint64_t gu64;
typedef uint32_t gu32;

#if !defined(__x86_64__)
#error "the wavefront emulator's fiber switch is written for x86-64"
#endif
// save the callee-saved registers and the stack pointer of the running fiber in *save_sp, continue on load_sp
extern "C" void tbc_emu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl tbc_emu_switch
.type tbc_emu_switch,@function
tbc_emu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size tbc_emu_switch,.-tbc_emu_switch
)");

struct EmuWave {
  void* sched = nullptr;
  void* ctx[64];
  char* stacks = nullptr;
  uint64_t deposit[64];
  uint64_t snap[64];
  int site[64];
  bool done[64];
  bool present[64];       // deposited at the current rendezvous
  int cur = 0;
  uint64_t rendezvous = 0;
  uint64_t clock = 0;
  int wave_index = 0;     // within its workgroup (wave_env_wg_emu.h)
};
inline EmuWave*& W() { static thread_local EmuWave* w = nullptr; return w; }

inline uint32_t lane_id() { return (uint32_t)W()->cur; }

// deposit v, wait for the wavefront, return the snapshot (64 values; lanes that have ended read as 0)
inline const uint64_t* gather(uint64_t v, int site) {
  EmuWave* w = W();
  w->deposit[w->cur] = v; w->site[w->cur] = site; w->present[w->cur] = true;
  const int me = w->cur;
  tbc_emu_switch(&w->ctx[me], w->sched);
  return W()->snap;
}

#define WV_SITE (__LINE__)

inline uint64_t ballot_at(bool p, int site) {
  const uint64_t* s = gather(p ? 1u : 0u, site);
  uint64_t b = 0;
  for (int l = 0; l < 64; l++) b |= (s[l] & 1ull) << l;
  return b;
}
inline uint32_t readlane_at(uint32_t v, uint32_t l, int site) { return (uint32_t)gather(v, site)[l & 63u]; }
inline uint64_t readlane64_at(uint64_t v, uint32_t l, int site) { return gather(v, site)[l & 63u]; }
// the primitives are macros here so that the site is the CALL site
#define ballot(p) ballot_at((p), WV_SITE)
#define readlane(v, l) readlane_at((v), (l), WV_SITE)
#define readlane64(v, l) readlane64_at((v), (l), WV_SITE)
#define barrier() barrier_at(WV_SITE)
inline uint32_t readfirstlane(uint32_t v) { return v; }
template <int N>
inline uint32_t row_shr0_at(uint32_t v, int site) {
  const int me = W()->cur;
  const uint64_t* s = gather(v, site);
  return (me & 15) >= N ? (uint32_t)s[me - N] : 0u;
}
#define row_shr0 row_shr0_emu
template <int N>
inline uint32_t row_shr0_emu(uint32_t v) { return row_shr0_at<N>(v, 100000 + N); }
inline void barrier_at(int site) { (void)gather(0, site); }
inline void threadfence() {}
inline void wg_fence() {}
inline void count_decided(uint32_t* dev, uint32_t* host, uint32_t every_mask) { const uint32_t n = __atomic_fetch_add(dev, 1u, __ATOMIC_RELAXED) + 1u; if ((n & every_mask) == 0u) *host = n; }
inline uint32_t opaque(uint32_t v) { return v; }

inline uint64_t ld64(const gu64* p) { return *p; }
inline void st64(gu64* p, uint64_t v) { *p = v; }
inline uint32_t ld32(const gu32* p) { return *p; }
inline void st32(gu32* p, uint32_t v) { *p = v; }
inline uint64_t own_ld64(const gu64* p) { return *p; }
inline void own_st64(gu64* p, uint64_t v) { *p = v; }
inline uint32_t own_ld32(const gu32* p) { return *p; }
inline void own_st32(gu32* p, uint32_t v) { *p = v; }
inline u32x4 own_ld128(const gu64* p) { const uint32_t* q = reinterpret_cast<const uint32_t*>(p); return u32x4{q[0], q[1], q[2], q[3]}; }
inline uint64_t cas64_from_zero(gu64* p, uint64_t desired) {
  const uint64_t old = *p;
  if (old == 0ull) *p = desired;
  return old;
}
inline void ld_bucket16(const gu64* bucket, u32x4& e0, u32x4& e1, u32x4& e2, u32x4& e3) {
  const uint32_t* q = reinterpret_cast<const uint32_t*>(bucket);
  e0 = u32x4{q[0], q[1], q[2], q[3]}; e1 = u32x4{q[4], q[5], q[6], q[7]};
  e2 = u32x4{q[8], q[9], q[10], q[11]}; e3 = u32x4{q[12], q[13], q[14], q[15]};
}
inline void wait_stores() {}
inline void lds_add32(uint32_t* p, uint32_t v) { *p += v; }
inline void lds_add64(uint32_t* p, uint64_t v) { uint64_t x; memcpy(&x, p, 8); x += v; memcpy(p, &x, 8); }
inline void lds_max32(uint32_t* p, uint32_t v) { if (v > *p) *p = v; }
inline uint64_t* stats() { static uint64_t s[64]; return s; }
inline void stat(uint32_t id, uint64_t v) { stats()[id & 63u] += v; }
inline uint64_t clock100mhz() { return W()->clock += 1; }
template <class Args>
inline const Args* cold(const Args& a) { return &a; }

// ---- the scheduler: run body(lane) on 64 fibers until all have returned
struct Launch { void (*fn)(void*, uint32_t); void* arg; };
inline Launch& launch_slot() { static thread_local Launch l; return l; }
inline void fiber_main() {
  EmuWave* w = W();
  const int lane = w->cur;
  Launch& L = launch_slot();
  L.fn(L.arg, (uint32_t)lane);
  w = W();
  w->done[lane] = true;
  tbc_emu_switch(&w->ctx[lane], w->sched);
  abort();                                  // a finished fiber is never resumed
}
inline void run_wave(void (*fn)(void*, uint32_t), void* arg) {
  constexpr size_t kStack = 256 * 1024;
  EmuWave* w = new EmuWave();
  w->stacks = (char*)malloc(64 * kStack + 64);
  W() = w;
  launch_slot() = Launch{fn, arg};
  for (int l = 0; l < 64; l++) {
    w->done[l] = false; w->present[l] = false;
    // a fresh fiber: six callee-saved registers to pop, then `ret` into fiber_main with the stack as after a call
    uintptr_t top = ((uintptr_t)(w->stacks + (size_t)(l + 1) * kStack)) & ~(uintptr_t)15;
    void** sp = (void**)top;
    *--sp = nullptr;                         // fake return address of fiber_main
    *--sp = (void*)&fiber_main;
    for (int r = 0; r < 6; r++) *--sp = nullptr;
    w->ctx[l] = (void*)sp;
  }
  for (;;) {
    bool any = false;
    for (int l = 0; l < 64; l++) {
      if (w->done[l]) continue;
      any = true;
      w->cur = l; w->present[l] = false;
      tbc_emu_switch(&w->sched, w->ctx[l]);
    }
    if (!any) break;
    // everybody has either ended or arrived at a primitive: it must be the same one
    int site = -1;
    for (int l = 0; l < 64; l++) {
      if (w->done[l]) { w->snap[l] = 0; continue; }
      if (!w->present[l]) { fprintf(stderr, "emu: lane %d neither ended nor arrived\n", l); abort(); }
      if (site < 0) site = w->site[l];
      else if (site != w->site[l]) { fprintf(stderr, "emu: divergent cross-lane primitive: lane %d at site %d, an earlier lane at site %d (rendezvous %llu)\n", l, w->site[l], site, (unsigned long long)w->rendezvous); abort(); }
      w->snap[l] = w->deposit[l];
    }
    w->rendezvous++;
  }
  free(w->stacks);
  delete w;
  W() = nullptr;
}

}  // namespace wv

// what device code calls unqualified
inline unsigned int atomicAdd(unsigned int* p, unsigned int v) { const unsigned int o = *p; *p += v; return o; }
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { const unsigned long long o = *p; *p += v; return o; }
