// TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  wave_env_wg_emu.h: csrc/wave_env_wg.h for the host emulator -- a WORKGROUP of NW
// wavefronts (64 fibers each, wave_env_emu.h).  Wave-level primitives rendezvous inside a wavefront as before; wg_barrier() is a
// rendezvous of every live fiber of the workgroup.  Between rendezvous the scheduler runs the wavefronts one after the other in an
// order it draws anew every pass from a seed: code in which one wavefront reads LDS another writes without a wg_barrier() between
// them gives different answers for different seeds (the tests run several).  A wavefront that has ended no longer takes part in
// the barrier (as s_barrier counts only the live ones).
#pragma once

namespace wv {

constexpr int kWgSite = 1 << 24;        // site ids of workgroup barriers carry this flag

struct EmuGroup { EmuWave* wave[32]; int nw = 0; uint32_t index = 0; };
inline EmuGroup*& WG() { static thread_local EmuGroup* g = nullptr; return g; }

inline void wg_barrier_at(int site) { (void)gather(0, site | kWgSite); }
#define wg_barrier() wg_barrier_at(WV_SITE)
inline uint32_t wg_thread() { return (uint32_t)(W()->wave_index * 64 + W()->cur); }
inline uint32_t wg_index() { return WG()->index; }
inline uint32_t lds_ld32(const uint32_t* p) { return *p; }
inline uint32_t lds_cas32(uint32_t* p, uint32_t expected, uint32_t desired) { const uint32_t old = *p; if (old == expected) *p = desired; return old; }
inline void lds_or32(uint32_t* p, uint32_t v) { *p |= v; }
inline uint32_t lds_fetch_add32(uint32_t* p, uint32_t v) { const uint32_t old = *p; *p += v; return old; }
inline void lds_add32_wg(uint32_t* p, uint32_t v) { *p += v; }
inline uint32_t wave_or32_at(uint32_t v, int site) {
  const uint64_t* s = gather(v, site);
  uint32_t r = 0;
  for (int l = 0; l < 64; l++) r |= (uint32_t)s[l];
  return r;
}
#define wave_or32(v) wave_or32_at((v), WV_SITE)

// run fn(arg, thread) on NW x 64 fibers
inline void run_workgroup(void (*fn)(void*, uint32_t), void* arg, int nw, uint32_t wg_index, uint64_t seed) {
  constexpr size_t kStack = 256 * 1024;
  EmuGroup g; g.nw = nw; g.index = wg_index;
  WG() = &g;
  launch_slot() = Launch{fn, arg};
  // fiber_main passes the lane as the argument: a workgroup's body wants the thread -- it asks wg_thread() itself
  for (int w = 0; w < nw; w++) {
    EmuWave* e = new EmuWave();
    e->wave_index = w;
    e->stacks = (char*)malloc(64 * kStack + 64);
    for (int l = 0; l < 64; l++) {
      e->done[l] = false; e->present[l] = false;
      uintptr_t top = ((uintptr_t)(e->stacks + (size_t)(l + 1) * kStack)) & ~(uintptr_t)15;
      void** sp = (void**)top;
      *--sp = nullptr;
      *--sp = (void*)&fiber_main;
      for (int r = 0; r < 6; r++) *--sp = nullptr;
      e->ctx[l] = (void*)sp;
    }
    g.wave[w] = e;
  }
  bool blocked[32] = {}, finished[32] = {};
  uint64_t n_barriers = 0;                 // workgroup barriers passed (stats 40: the most of any workgroup so far, 41: their sum)
  int bsite[32] = {};
  uint64_t rng = seed * 0x9E3779B97F4A7C15ull + 0x1234567ull;
  for (;;) {
    int order[32];
    for (int w = 0; w < nw; w++) order[w] = w;
    for (int w = nw - 1; w > 0; w--) {                       // a fresh order every pass
      rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17;
      const int j = (int)(rng % (uint64_t)(w + 1));
      const int t = order[w]; order[w] = order[j]; order[j] = t;
    }
    bool any_unfinished = false, progressed = false;
    for (int oi = 0; oi < nw; oi++) {
      const int wi = order[oi];
      EmuWave* e = g.wave[wi];
      if (finished[wi]) continue;
      any_unfinished = true;
      if (blocked[wi]) continue;
      // (how many passes in a row this wavefront takes before the others move: 1 .. 3, so that a wavefront can run well ahead)
      rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17;
      const int burst = 1 + (int)(rng % 3u);
      for (int b = 0; b < burst && !blocked[wi] && !finished[wi]; b++) {
        bool any = false;
        W() = e;
        for (int l = 0; l < 64; l++) {
          if (e->done[l]) continue;
          any = true;
          e->cur = l; e->present[l] = false;
          tbc_emu_switch(&e->sched, e->ctx[l]);
        }
        if (!any) { finished[wi] = true; break; }
        progressed = true;
        int site = -1;
        bool all_done = true;
        for (int l = 0; l < 64; l++) {
          if (e->done[l]) { e->snap[l] = 0; continue; }
          all_done = false;
          if (!e->present[l]) { fprintf(stderr, "emu: wave %d lane %d neither ended nor arrived\n", wi, l); abort(); }
          if (site < 0) site = e->site[l];
          else if (site != e->site[l]) { fprintf(stderr, "emu: divergent cross-lane primitive in wave %d: lane %d at site %d, an earlier lane at site %d\n", wi, l, e->site[l] & ~kWgSite, site & ~kWgSite); abort(); }
          e->snap[l] = e->deposit[l];
        }
        if (all_done) { finished[wi] = true; break; }
        e->rendezvous++;
        if (site & kWgSite) { blocked[wi] = true; bsite[wi] = site; }
      }
    }
    if (!any_unfinished) break;
    // every live wavefront at the workgroup barrier: it must be the same one; release them
    bool all_blocked = true;
    for (int w = 0; w < nw; w++) if (!finished[w] && !blocked[w]) all_blocked = false;
    if (all_blocked) {
      int site = -1;
      bool any = false;
      for (int w = 0; w < nw; w++) {
        if (finished[w]) continue;
        any = true;
        if (site < 0) site = bsite[w];
        else if (site != bsite[w]) { fprintf(stderr, "emu: wavefronts at different workgroup barriers: wave %d at site %d, an earlier one at site %d\n", w, bsite[w] & ~kWgSite, site & ~kWgSite); abort(); }
      }
      if (!any) break;
      for (int w = 0; w < nw; w++) blocked[w] = false;
      progressed = true;
      n_barriers++;
    }
    if (!progressed) { fprintf(stderr, "emu: workgroup made no progress\n"); abort(); }
  }
  for (int w = 0; w < nw; w++) { free(g.wave[w]->stacks); delete g.wave[w]; }
  if (n_barriers > stats()[40]) stats()[40] = n_barriers;
  stats()[41] += n_barriers;
  W() = nullptr;
  WG() = nullptr;
}

}  // namespace wv
