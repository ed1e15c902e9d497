/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE (see oracle_model.h).
 *
 * many.c -- the CPU restatement on ALL host cores: bench.py's cpu_baseline times wgl_window_check
 * (the sequential knossos.wgl restatement) over a list of histories with a pthread pool, so the
 * "one MI355X vs this host" ratio is not limited by Python's dispatch.  Each worker takes the next
 * history off a shared counter; every history is checked by exactly one thread, as stock Knossos
 * would check independent keys on a thread pool (jepsen.independent/checker).  wgl_beam_check_many does the same
 * with the wide schedule the GPU kernel runs (wgl_beam.c: lookahead, eager reads, twin rule as switched on by the
 * caller beforehand) -- the strongest CPU competitor this repository has.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include "oracle_model.h"

int wgl_window_check(uint32_t n, const uint8_t* f, const int32_t* a, const int32_t* b,
                     const int32_t* process, uint32_t n_process,
                     const uint32_t* inv_pos, const uint32_t* ret_pos,
                     const oracle_model* model, uint64_t max_steps,
                     uint32_t* witness, oracle_result* out);

typedef struct beam_stats { uint64_t iterations, probes, visited, expanded, max_stack, rounds; } beam_stats;
int wgl_beam_check_rp(uint32_t n, const uint8_t* f, const int32_t* a, const int32_t* b,
                      const int32_t* process, uint32_t n_process,
                      const uint32_t* inv_pos, const uint32_t* ret_pos,
                      const oracle_model* model, uint32_t K, uint32_t round_pairs, uint64_t max_probes,
                      uint32_t* witness, oracle_result* out, beam_stats* st);

typedef struct {
  uint32_t n_hist;
  const uint32_t* n; const uint32_t* n_process;
  const uint8_t* const* f; const int32_t* const* a; const int32_t* const* b; const int32_t* const* process;
  const uint32_t* const* inv_pos; const uint32_t* const* ret_pos;
  const oracle_model* model; uint64_t max_steps;
  int32_t* valid;
  volatile uint32_t next;
  uint32_t beam_width;      /* 0 = wgl_window_check, else wgl_beam_check_rp at this many configs per round ... */
  uint32_t round_pairs;     /* ... and this many pairs per round (64 = one history per wavefront; 8 / 16 / 32 = the narrow kernel's schedules) */
} many_job;

static void* many_worker(void* p) {
  many_job* j = (many_job*)p;
  for (;;) {
    const uint32_t i = __atomic_fetch_add(&j->next, 1u, __ATOMIC_RELAXED);
    if (i >= j->n_hist) break;
    oracle_result r;
    beam_stats bs;
    int rc = j->beam_width
      ? wgl_beam_check_rp(j->n[i], j->f[i], j->a[i], j->b[i], j->process[i], j->n_process[i], j->inv_pos[i], j->ret_pos[i],
                          j->model, j->beam_width, j->round_pairs ? j->round_pairs : 64u, j->max_steps, NULL, &r, &bs)
      : wgl_window_check(j->n[i], j->f[i], j->a[i], j->b[i], j->process[i], j->n_process[i], j->inv_pos[i], j->ret_pos[i],
                         j->model, j->max_steps, NULL, &r);
    j->valid[i] = rc ? -2 : r.valid;
  }
  return NULL;
}

int wgl_check_many(uint32_t n_hist, const uint32_t* n, const uint32_t* n_process,
                   const uint8_t* const* f, const int32_t* const* a, const int32_t* const* b, const int32_t* const* process,
                   const uint32_t* const* inv_pos, const uint32_t* const* ret_pos,
                   const oracle_model* model, uint64_t max_steps, uint32_t n_threads, uint32_t beam_width, uint32_t round_pairs, int32_t* valid);

/* valid[i] = verdict of history i (-2 = rejected).  Returns the number of threads actually started. */
int wgl_window_check_many(uint32_t n_hist, const uint32_t* n, const uint32_t* n_process,
                          const uint8_t* const* f, const int32_t* const* a, const int32_t* const* b, const int32_t* const* process,
                          const uint32_t* const* inv_pos, const uint32_t* const* ret_pos,
                          const oracle_model* model, uint64_t max_steps, uint32_t n_threads, int32_t* valid) {
  return wgl_check_many(n_hist, n, n_process, f, a, b, process, inv_pos, ret_pos, model, max_steps, n_threads, 0, 64, valid);
}

int wgl_check_many(uint32_t n_hist, const uint32_t* n, const uint32_t* n_process,
                   const uint8_t* const* f, const int32_t* const* a, const int32_t* const* b, const int32_t* const* process,
                   const uint32_t* const* inv_pos, const uint32_t* const* ret_pos,
                   const oracle_model* model, uint64_t max_steps, uint32_t n_threads, uint32_t beam_width, uint32_t round_pairs, int32_t* valid) {
  many_job j = {n_hist, n, n_process, f, a, b, process, inv_pos, ret_pos, model, max_steps, valid, 0, beam_width, round_pairs};
  if (n_threads == 0) n_threads = 1;
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * n_threads);
  uint32_t started = 0;
  for (uint32_t t = 0; t < n_threads; t++) if (pthread_create(&th[started], NULL, many_worker, &j) == 0) started++;
  if (started == 0) many_worker(&j);
  for (uint32_t t = 0; t < started; t++) pthread_join(th[t], NULL);
  free(th);
  return (int)started;
}
