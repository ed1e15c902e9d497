// reach_table.h -- host side of the RELAXED level sweep (jit_sweep_wg_impl.h, RLX; specified in oracle/sweep_ref.c, sweep_set_relaxed):
// from a history's classes of crashed calls (the count form's block of cmem[]: tbc_internal.h, kRuleCount) the table the sweep reads --
//   [ep_from[n_ep]]   the front from which epoch e holds (ascending, ep_from[0] = 0): the classes whose first member is invoked by a
//                     front are available at it, so the closure below changes only where a class's first member is invoked
//   [nt_max[n_ep]]    the most states any state reaches in epoch e (a sub-round gives a config (1 + nt_max) x (open calls + 1) pairs)
//   [rows[n_ep][32]]  reach[e][si] = bit t: state index t (0 = nil, v + 1 = value v: the read tables' rdm_index) is reachable from
//                     state index si through one or more class steps -- (:write v) from any state, (:cas [a b]) from state a.
// Plain host C++ (no HIP): tbc_api.hip builds it when the inputs become resident; tests/emu builds it for the emulated kernel; the
// oracle has its own, independent statement of the same closure.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>
#include "tbc_internal.h"

namespace tbc {

// cmem: the history's block (n_classes 16 B class records {first member word, f | shift << 8 | width << 16, a, b}, then the members
// inv_rank | op << 32, a sentinel after each class).  Appends the table to `out`, returns the number of epochs.
inline uint32_t build_reach_table(const uint64_t* cmem, uint32_t n_classes, std::vector<uint32_t>& out) {
  struct Cls { uint32_t f, from; int32_t a, b; };
  std::vector<Cls> cls(n_classes);
  std::vector<uint32_t> from{0u};
  for (uint32_t k = 0; k < n_classes; k++) {
    OpRec o;
    std::memcpy(&o, cmem + 2 * (size_t)k, sizeof o);
    cls[k] = Cls{o.f_slot & 0xFFu, (uint32_t)cmem[o.op], o.a, o.b};          // (the first member's invocation rank)
    from.push_back(cls[k].from);
  }
  std::sort(from.begin(), from.end());
  from.erase(std::unique(from.begin(), from.end()), from.end());
  const uint32_t n_ep = (uint32_t)from.size();
  const size_t base = out.size();
  out.resize(base + (size_t)n_ep * 34, 0u);
  for (uint32_t e = 0; e < n_ep; e++) {
    uint32_t* row = out.data() + base + 2 * (size_t)n_ep + 32 * (size_t)e;
    for (bool again = true; again;) {
      again = false;
      for (uint32_t si = 0; si < 32; si++)
        for (const Cls& c : cls) {
          if (c.from > from[e]) continue;
          if (c.f == TBC_F_CAS && (si == 0 ? TBC_NIL : (int32_t)si - 1) != c.a) continue;
          const int32_t tv = c.f == TBC_F_WRITE ? c.a : c.b;
          if (tv < 0 || tv + 1 >= 32) continue;
          const uint32_t add = (1u << (uint32_t)(tv + 1)) | row[tv + 1];
          if ((row[si] | add) != row[si]) { row[si] |= add; again = true; }
        }
    }
    uint32_t ntm = 0;
    for (uint32_t si = 0; si < 32; si++) ntm = std::max(ntm, (uint32_t)__builtin_popcount(row[si] & ~(1u << si)));
    out[base + e] = from[e];
    out[base + n_ep + e] = ntm;
  }
  return n_ep;
}

}  // namespace tbc
