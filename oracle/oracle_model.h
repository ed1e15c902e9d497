/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * build, link, load or call anything under oracle/.
 *
 * PARITY UNPINNED: the algorithm restated here lives in the third-party
 * library `knossos` (Clojars; reached transitively through
 * `jepsen "0.2.8-SNAPSHOT"`, /root/reference/project.clj:8; its version is not
 * pinned anywhere in the reference).  That library is not on
 * this machine, the reference holds no golden vectors for it
 * (/root/reference/test/tigerbeetle/core_test.clj:1-6 asserts `true`), and no
 * JVM exists here to run it.  What follows restates the PUBLISHED algorithms
 * (Wing & Gong 1993; Lowe, "Testing for linearizability", 2017) with the
 * Knossos semantics recalled in SURVEY.md section 8a, and is cross-checked against
 * an independent brute-force definition of linearizability (oracle/brute.py).
 *
 * oracle_model.h -- knossos.model/step for the models on the hot path.
 * Recalled semantics:
 *   register      :write v -> v ; :read v ok iff v nil or v = value
 *   cas-register  + :cas [cur new] -> new iff cur = value, else inconsistent
 *   mutex         :acquire inconsistent if held ; :release inconsistent if free
 *   table         next = table[state*n_classes + class] (knossos.model.memo)
 *   multi-register :txn [[f k v] ...] applied atomically; a micro-read is ok iff v is
 *                 nil or equals the key's value.  State = 4 bits per key (0 nil, v+1).
 *   set / bank    COMMUTATIVE models: the state is a function of WHICH calls are linearized,
 *                 not of their order, so configs carry no state (state word 0) and a read is
 *                 checked against (completions before the front) + (open calls linearized):
 *                 see oracle_cfg_step below.  set = knossos.model/set (:add v, :read s ok iff s
 *                 equals the state exactly; elements unique per history); bank = the model the
 *                 reference's ledger->bank mapping implies (tests/ledger.clj:89-114), negative
 *                 balances allowed (so transfers commute).
 */
#ifndef ORACLE_MODEL_H
#define ORACLE_MODEL_H
#include <stdint.h>

#define O_NIL INT32_MIN
#define O_CRASHED 0xFFFFFFFFu
enum { O_READ = 0, O_WRITE = 1, O_CAS = 2, O_ACQUIRE = 3, O_RELEASE = 4, O_CLASS = 8 };
enum { O_REGISTER = 0, O_CAS_REGISTER = 1, O_MUTEX = 2, O_TABLE = 3, O_MULTI_REGISTER = 4, O_SET = 5, O_BANK = 6 };
enum { O_ADD = 5, O_TXN = 6, O_TRANSFER = 7 };

typedef struct oracle_model {
  uint32_t kind;
  int32_t init;
  const uint16_t* table;
  uint32_t n_states, n_classes;
  const int32_t* pool;   /* multi-register: {f, key, value} triples; op a = offset, b = count */
} oracle_model;

/* returns 1 and sets *next if op (f,a,b) may be applied in `state`, else 0 */
static inline int oracle_step(const oracle_model* m, int32_t state, uint8_t f,
                              int32_t a, int32_t b, int32_t* next) {
  switch (m->kind) {
    case O_REGISTER:
    case O_CAS_REGISTER:
      if (f == O_WRITE) { *next = a; return 1; }
      if (f == O_READ) { *next = state; return a == O_NIL || a == state; }
      if (f == O_CAS && m->kind == O_CAS_REGISTER) { *next = b; return a == state; }
      return 0;
    case O_MUTEX:
      if (f == O_ACQUIRE) { *next = 1; return state == 0; }
      if (f == O_RELEASE) { *next = 0; return state == 1; }
      return 0;
    case O_MULTI_REGISTER: {
      if (f != O_TXN) return 0;
      uint32_t s = (uint32_t)state; int ok = 1;
      for (int32_t i = 0; i < b; i++) {
        int32_t mf = m->pool[a + 3 * i], k = m->pool[a + 3 * i + 1], v = m->pool[a + 3 * i + 2];
        uint32_t cur = (s >> (4 * k)) & 15u;
        if (mf == 0) { if (!(v == O_NIL || cur == (uint32_t)(v + 1))) ok = 0; }
        else s = (s & ~(15u << (4 * k))) | ((uint32_t)(v + 1) << (4 * k));
      }
      *next = (int32_t)s;
      return ok;
    }
    case O_TABLE: {
      if (f != O_CLASS || (uint32_t)a >= m->n_classes || (uint32_t)state >= m->n_states) return 0;
      uint16_t t = m->table[(uint32_t)state * m->n_classes + (uint32_t)a];
      if (t == 0xFFFFu) return 0;
      *next = (int32_t)t;
      return 1;
    }
  }
  return 0;
}

/*
 * Config-dependent step for the commutative models.  `front` = completions already passed,
 * open[0..n_open) = ops open at the front, lin[i] = open[i] already linearized.  Pool layout
 * (written by the host encoder, knossos/_analysis.py):
 *   set : model->init = offset of nadds_before[0..R]; :add a = index j of the add in completion
 *         order; :read a = O_NIL or offset of {nR, lead, words...} (bitset over adds in j order,
 *         nR = |R| or -1 if R holds an element nobody adds, lead = number of leading ones).
 *   bank: model->init = offset of bal_before[(R+1) * n_states] (n_states = #accounts);
 *         :transfer a = offset of {debit idx, credit idx, amount}; :read a = O_NIL or offset of
 *         the balances read.
 */
static inline int oracle_is_cfg_model(const oracle_model* m) { return m->kind == O_SET || m->kind == O_BANK; }
static inline int oracle_cfg_step(const oracle_model* m, uint32_t front, const uint32_t* open, const uint8_t* lin,
                                  uint32_t n_open, const uint8_t* f, const int32_t* a, uint32_t op) {
  const int32_t* pool = m->pool;
  if (m->kind == O_SET) {
    if (f[op] == O_ADD) return 1;
    if (f[op] != O_READ) return 0;
    if (a[op] == O_NIL) return 1;
    const int32_t* rec = pool + a[op];
    const int32_t nR = rec[0], lead = rec[1];
    const uint32_t* words = (const uint32_t*)(rec + 2);
    int32_t count = pool[m->init + (int32_t)front];
    if (nR < 0 || count > lead) return 0;
    for (uint32_t i = 0; i < n_open; i++) {
      if (!lin[i] || f[open[i]] != O_ADD) continue;
      uint32_t j = (uint32_t)a[open[i]];
      if (!(words[j >> 5] >> (j & 31) & 1u)) return 0;
      count++;
    }
    return count == nR;
  }
  if (m->kind == O_BANK) {
    if (f[op] == O_TRANSFER) return 1;
    if (f[op] != O_READ) return 0;
    if (a[op] == O_NIL) return 1;
    const uint32_t A = m->n_states;
    int32_t bal[16];
    for (uint32_t k = 0; k < A; k++) bal[k] = pool[m->init + (int32_t)(front * A + k)];
    for (uint32_t i = 0; i < n_open; i++) {
      if (!lin[i] || f[open[i]] != O_TRANSFER) continue;
      const int32_t* t = pool + a[open[i]];
      bal[t[0]] -= t[2]; bal[t[1]] += t[2];
    }
    for (uint32_t k = 0; k < A; k++) if (bal[k] != pool[a[op] + (int32_t)k]) return 0;
    return 1;
  }
  return 0;
}

/* result block shared by every oracle entry point */
typedef struct oracle_result {
  int32_t valid;          /* 1 / 0 / -1 (step limit) */
  uint32_t fail_op;       /* invalid: op whose completion no config passes */
  uint32_t prev_ok_op;    /* invalid: op completing just before it, or 0xFFFFFFFF */
  int32_t final_state;    /* valid */
  uint32_t n_witness;     /* valid: ops in linearization order */
  uint64_t steps, visited, probes, backtracks, max_depth;
} oracle_result;

#endif
