/*
 * tbsynth.h -- seeded synthetic Jepsen histories (event level), exported by
 * libtbcheck.so next to the checker.  Not part of the reference's surface: the
 * reference holds no recorded histories (/root/reference/.gitignore:7 ignores
 * store/, test/tigerbeetle/core_test.clj:1-6 asserts `true`), so every input
 * used by tests/ and bench.py is synthesised here, in the shapes the reference
 * documents (README.md:41-50,67-74; set_full.clj:29-31,42-45;
 * tests/ledger.clj:89-114) and SURVEY.md section 8d prescribes.
 *
 * A history is produced by simulating ONE atomic object: every op takes effect
 * at a random instant inside its [invoke, complete] interval, so an
 * uncorrupted history is linearizable by construction.
 */
#ifndef TBSYNTH_H
#define TBSYNTH_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct tbs_params {
  uint64_t seed;
  uint32_t n_ops;          /* invocations to issue                               */
  uint32_t n_procs;        /* concurrent worker processes (Jepsen :concurrency)  */
  uint32_t n_values;       /* values are 0..n_values-1 (Jepsen tutorial: 5)      */
  uint32_t busy_permille;  /* fraction of time a process has an op open         */
  uint32_t info_permille;  /* ops that crash (:info), process id then retired    */
  uint32_t read_permille;  /* op mix; remainder after read+write is cas          */
  uint32_t write_permille;
  uint32_t corrupt_permille; /* 0 = none; else corrupt the first :ok read at or  */
                             /* after this fraction of the history (1..1000)     */
} tbs_params;

/* cas-register / register history (register: set read+write = 1000).
 * Output columns are caller-allocated with capacity >= 2*n_ops rows.
 * Returns 0 on success. */
int tbs_gen_register(const tbs_params* p, uint8_t* type, int32_t* process,
                     uint8_t* f, int32_t* a, int32_t* b, uint32_t* n_rows);

#ifdef __cplusplus
}
#endif
#endif
