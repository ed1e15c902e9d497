// synth.cpp -- seeded synthetic Jepsen histories (see include/tbsynth.h).
//
// Discrete-event simulation of n_procs clients against one atomic register.
// Each op has three instants: invoke, take-effect (uniform inside the
// interval), complete.  Crashed ops (:info) take effect with probability 1/2
// and retire their process id, exactly as Jepsen replaces a crashed process
// with process + concurrency.  A cas that finds another value completes :fail
// (README.md:41-50 shows the reference's op rows; the cas-register op mix and
// the 0..4 value domain follow SURVEY.md section 8d, config 2).
#include <cstdint>
#include <queue>
#include <vector>
#include <cmath>
#include "../../include/tbsynth.h"
#include "../../include/tbcheck.h"

namespace {

struct Rng {
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 0x1234567ull) {}
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
  uint32_t below(uint32_t n) { return (uint32_t)(next() % n); }
  double expo(double mean) { return -mean * std::log(1.0 - uniform()); }
};

enum Kind : uint8_t { K_INVOKE = 0, K_EFFECT = 1, K_COMPLETE = 2 };

struct Ev {
  double t;
  uint64_t seq;
  uint32_t worker;
  Kind kind;
  bool operator>(const Ev& o) const { return t != o.t ? t > o.t : seq > o.seq; }
};

struct Open {
  int32_t process;
  uint8_t f;
  int32_t a, b;
  bool crashed, takes_effect, failed;
  int32_t observed;
};

}  // namespace

extern "C" int tbs_gen_register(const tbs_params* p, uint8_t* type, int32_t* process,
                                uint8_t* f, int32_t* a, int32_t* b, uint32_t* n_rows) {
  if (!p || !type || !process || !f || !a || !b || !n_rows) return 1;
  if (p->n_procs == 0 || p->n_values == 0) return 1;
  Rng rng(p->seed);
  const double lat = 1.0;
  const double busy = p->busy_permille >= 1000 ? 0.999 : (p->busy_permille ? p->busy_permille / 1000.0 : 0.001);
  const double think = lat * (1.0 - busy) / busy;
  std::priority_queue<Ev, std::vector<Ev>, std::greater<Ev>> q;
  uint64_t seq = 0;
  std::vector<Open> open(p->n_procs);
  std::vector<int32_t> pid(p->n_procs);
  int32_t next_pid = (int32_t)p->n_procs;
  for (uint32_t w = 0; w < p->n_procs; w++) {
    pid[w] = (int32_t)w;
    q.push({rng.expo(think + 0.05 * lat), seq++, w, K_INVOKE});
  }
  int32_t state = TBC_NIL;
  uint32_t issued = 0, rows = 0;
  while (!q.empty()) {
    Ev e = q.top();
    q.pop();
    Open& o = open[e.worker];
    if (e.kind == K_INVOKE) {
      if (issued >= p->n_ops) continue;
      issued++;
      uint32_t r = rng.below(1000);
      o = Open{};
      o.process = pid[e.worker];
      if (r < p->read_permille) { o.f = TBC_F_READ; o.a = TBC_NIL; o.b = 0; }
      else if (r < p->read_permille + p->write_permille) { o.f = TBC_F_WRITE; o.a = (int32_t)rng.below(p->n_values); o.b = 0; }
      else { o.f = TBC_F_CAS; o.a = (int32_t)rng.below(p->n_values); o.b = (int32_t)rng.below(p->n_values); }
      o.crashed = rng.below(1000) < p->info_permille;
      o.takes_effect = o.crashed ? (rng.next() & 1) : true;
      double L = 0.02 * lat + rng.expo(lat);
      type[rows] = TBC_INVOKE; process[rows] = o.process; f[rows] = o.f; a[rows] = o.a; b[rows] = o.b; rows++;
      q.push({e.t + rng.uniform() * L, seq++, e.worker, K_EFFECT});
      q.push({e.t + L, seq++, e.worker, K_COMPLETE});
    } else if (e.kind == K_EFFECT) {
      if (!o.takes_effect) continue;
      if (o.f == TBC_F_READ) o.observed = state;
      else if (o.f == TBC_F_WRITE) state = o.a;
      else { if (state == o.a) state = o.b; else o.failed = true; }
    } else {
      if (o.crashed) {
        type[rows] = TBC_INFO; process[rows] = o.process; f[rows] = o.f; a[rows] = o.a; b[rows] = o.b; rows++;
        pid[e.worker] = next_pid++;
      } else if (o.failed) {
        type[rows] = TBC_FAIL; process[rows] = o.process; f[rows] = o.f; a[rows] = o.a; b[rows] = o.b; rows++;
      } else {
        type[rows] = TBC_OK_; process[rows] = o.process; f[rows] = o.f;
        a[rows] = o.f == TBC_F_READ ? o.observed : o.a; b[rows] = o.b; rows++;
      }
      q.push({e.t + rng.expo(think) + 1e-9, seq++, e.worker, K_INVOKE});
    }
  }
  if (p->corrupt_permille) {
    uint32_t start = (uint32_t)((uint64_t)rows * (p->corrupt_permille > 1000 ? 1000 : p->corrupt_permille) / 1000);
    if (start >= rows) start = rows ? rows - 1 : 0;
    // first :ok read at or after `start`, else the last one before it
    int64_t hit = -1;
    for (uint32_t i = start; i < rows; i++) if (type[i] == TBC_OK_ && f[i] == TBC_F_READ) { hit = i; break; }
    if (hit < 0) for (int64_t i = (int64_t)start - 1; i >= 0; i--) if (type[i] == TBC_OK_ && f[i] == TBC_F_READ) { hit = i; break; }
    if (hit >= 0) {
      // a value no write or cas in this history can produce: never linearizable
      a[hit] = (int32_t)p->n_values + 7;
    }
  }
  *n_rows = rows;
  return 0;
}
