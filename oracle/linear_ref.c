/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE (see oracle_model.h for the rules
 * and the PARITY UNPINNED statement).
 *
 * linear_ref.c -- plain-C restatement of knossos.linear/analysis as recalled in
 * SURVEY.md section 8a (rows `knossos.linear/analysis + knossos.linear.config`): Lowe's
 * just-in-time linearization.  The history is swept in order while the SET of
 * all reachable configs is carried along:
 *
 *   :invoke  the call becomes pending (in every config);
 *   :ok      every config must have the op linearized by now: a config that
 *            already has it simply returns it; any other config is expanded
 *            over every sequence of its pending calls that ends with this op;
 *            configs that cannot linearize it are dropped;
 *            empty set  =>  not linearizable, :op = this completion,
 *            :previous-ok = the completion before it, :configs = the set as it
 *            stood before this completion;
 *   :info    the call stays pending for ever (it may be linearized inside any
 *            later expansion, or never).
 *
 * A config is (model state, set of pending calls already linearized); calls
 * are indexed by process (one open call per process), as knossos.linear.config
 * does.  The set is exact (full keys compared).
 *
 * Outputs that are properties of (model, history) and therefore comparable
 * bit-for-bit with the HIP sweep: verdict, failing op, previous-ok op, the
 * final config set (sorted), the number of configs after every completion
 * (summed, and the maximum), and the number of model-consistent expansion
 * steps ("probes").
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "oracle_model.h"

typedef struct { uint32_t pos, op; } posop;
static int cmp_posop(const void* x, const void* y) {
  uint32_t a = ((const posop*)x)->pos, b = ((const posop*)y)->pos;
  return a < b ? -1 : a > b;
}
static uint64_t mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xFF51AFD7ED558CCDull;
  x ^= x >> 33; x *= 0xC4CEB9FE1A85EC53ull;
  x ^= x >> 33; return x;
}

/* growable exact set of configs; entry = kw words: [state+1 (never 0)] [mask words] */
typedef struct { uint64_t* tab; size_t cap, n, kw; } cset;
static void cset_init(cset* s, size_t kw) { s->kw = kw; s->cap = 64; s->n = 0; s->tab = (uint64_t*)calloc(s->cap * kw, 8); }
static void cset_clear(cset* s) { memset(s->tab, 0, s->cap * s->kw * 8); s->n = 0; }
static uint64_t key_hash(const uint64_t* k, size_t kw) {
  uint64_t h = mix64(k[0]);
  for (size_t i = 1; i < kw; i++) h = mix64(h ^ k[i]) + 0x9E3779B97F4A7C15ull;
  return h;
}
static int cset_add_nogrow(cset* s, const uint64_t* k) {
  size_t j = key_hash(k, s->kw) & (s->cap - 1);
  for (;;) {
    uint64_t* e = s->tab + j * s->kw;
    if (e[0] == 0) { memcpy(e, k, s->kw * 8); s->n++; return 1; }
    if (memcmp(e, k, s->kw * 8) == 0) return 0;
    j = (j + 1) & (s->cap - 1);
  }
}
static int cset_add(cset* s, const uint64_t* k) {
  if ((s->n + 1) * 2 > s->cap) {
    cset t; t.kw = s->kw; t.cap = s->cap * 2; t.n = 0; t.tab = (uint64_t*)calloc(t.cap * t.kw, 8);
    for (size_t j = 0; j < s->cap; j++) if (s->tab[j * s->kw] != 0) cset_add_nogrow(&t, s->tab + j * s->kw);
    free(s->tab); *s = t;
  }
  return cset_add_nogrow(s, k);
}

typedef struct linear_stats {
  uint64_t configs_total;   /* sum over completions of the config-set size after it */
  uint64_t max_configs;     /* largest config set after a completion */
  uint64_t probes;          /* model-consistent expansion steps (set insert attempts) */
  uint64_t expanded;        /* configs popped from an expansion worklist */
  uint64_t levels;          /* completions processed */
} linear_stats;

static int cmp_key(const void* x, const void* y, void* kwp) {
  size_t kw = *(size_t*)kwp;
  const uint64_t* a = (const uint64_t*)x; const uint64_t* b = (const uint64_t*)y;
  for (size_t i = 0; i < kw; i++) { if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1; }
  return 0;
}
/* qsort_r is not C11: tiny insertion/merge-free sort through a global */
static size_t g_kw;
static int cmp_key_g(const void* x, const void* y) { return cmp_key(x, y, &g_kw); }

/*
 * final_configs: caller buffer of max_final * (1 + mask_words) uint64 (sorted
 * ascending by (state word, mask words)); *n_final = size of the final set
 * (may exceed max_final; only the first max_final are written).
 * max_configs_limit: 0 = none; a config set larger than this => valid = -1.
 */
int linear_ref_check(uint32_t n, const uint8_t* f, const int32_t* a, const int32_t* b,
                     const int32_t* process, uint32_t n_process,
                     const uint32_t* inv_pos, const uint32_t* ret_pos,
                     const oracle_model* model, uint64_t max_configs_limit,
                     uint64_t* final_configs, uint32_t max_final, uint32_t* n_final,
                     oracle_result* out, linear_stats* st) {
  memset(out, 0, sizeof *out); memset(st, 0, sizeof *st);
  out->fail_op = out->prev_ok_op = 0xFFFFFFFFu;
  *n_final = 0;
  uint32_t R = 0;
  for (uint32_t i = 0; i < n; i++) {
    if (i && inv_pos[i] <= inv_pos[i - 1]) return 2;
    if (process[i] < 0 || (uint32_t)process[i] >= n_process) return 2;
    if (ret_pos[i] != O_CRASHED) { if (ret_pos[i] <= inv_pos[i]) return 2; R++; }
  }
  const uint32_t W = n_process ? n_process : 1, MW = (W + 63) / 64, KW = 1 + MW;
  posop* rets = (posop*)malloc(sizeof(posop) * (R ? R : 1));
  uint32_t k = 0;
  for (uint32_t i = 0; i < n; i++) if (ret_pos[i] != O_CRASHED) { rets[k].pos = ret_pos[i]; rets[k].op = i; k++; }
  qsort(rets, R, sizeof(posop), cmp_posop);

  int64_t* slot_op = (int64_t*)malloc(8 * W);      /* op open on each process, or -1 */
  for (uint32_t p = 0; p < W; p++) slot_op[p] = -1;
  cset cur, nxt, clo;
  cset_init(&cur, KW); cset_init(&nxt, KW); cset_init(&clo, KW);
  uint64_t* key = (uint64_t*)calloc(KW, 8);
  uint64_t* key2 = (uint64_t*)calloc(KW, 8);
  size_t wl_cap = 1024, wl_n = 0;
  uint64_t* wl = (uint64_t*)malloc(wl_cap * KW * 8);

  key[0] = (uint64_t)(uint32_t)model->init + 1;   /* state word: value + 1 so it is never 0 */
  cset_add(&cur, key);
  int verdict = 1;
  uint32_t ci = 0;
  for (uint32_t r = 0; r < R && verdict == 1; r++) {
    const uint32_t x = rets[r].op, px = (uint32_t)process[x];
    while (ci < n && inv_pos[ci] < rets[r].pos) {       /* invocations before this completion */
      if (slot_op[process[ci]] != -1) { verdict = -3; break; }
      slot_op[process[ci]] = ci; ci++;
    }
    if (verdict != 1) break;
    cset_clear(&nxt); cset_clear(&clo); wl_n = 0;
    for (size_t j = 0; j < cur.cap; j++) {
      const uint64_t* c = cur.tab + j * KW;
      if (c[0] == 0) continue;
      if (c[1 + (px >> 6)] >> (px & 63) & 1) {           /* already linearized: it returns */
        memcpy(key, c, KW * 8);
        key[1 + (px >> 6)] &= ~(1ull << (px & 63));
        cset_add(&nxt, key);
      } else {
        cset_add(&clo, c);
        if (wl_n == wl_cap) { wl_cap *= 2; wl = (uint64_t*)realloc(wl, wl_cap * KW * 8); }
        memcpy(wl + wl_n * KW, c, KW * 8); wl_n++;
      }
    }
    while (wl_n) {                                        /* just-in-time expansion */
      wl_n--;
      memcpy(key, wl + wl_n * KW, KW * 8);
      st->expanded++;
      const int32_t s = (int32_t)(uint32_t)(key[0] - 1);
      for (uint32_t p = 0; p < W; p++) {
        if (slot_op[p] < 0) continue;
        if (key[1 + (p >> 6)] >> (p & 63) & 1) continue;
        const uint32_t y = (uint32_t)slot_op[p];
        int32_t s2;
        if (!oracle_step(model, s, f[y], a[y], b[y], &s2)) continue;
        st->probes++;
        memcpy(key2, key, KW * 8);
        key2[0] = (uint64_t)(uint32_t)s2 + 1;
        if (y == x) {
          cset_add(&nxt, key2);                           /* linearized and returned at once */
        } else {
          key2[1 + (p >> 6)] |= 1ull << (p & 63);
          if (cset_add(&clo, key2)) {
            if (wl_n == wl_cap) { wl_cap *= 2; wl = (uint64_t*)realloc(wl, wl_cap * KW * 8); }
            memcpy(wl + wl_n * KW, key2, KW * 8); wl_n++;
          }
        }
      }
    }
    st->levels++;
    if (nxt.n == 0) {
      verdict = 0;
      out->fail_op = x;
      out->prev_ok_op = r ? rets[r - 1].op : 0xFFFFFFFFu;
      break;                                              /* `cur` holds :configs */
    }
    slot_op[px] = -1;
    { cset t = cur; cur = nxt; nxt = t; }
    st->configs_total += cur.n;
    if (cur.n > st->max_configs) st->max_configs = cur.n;
    if (max_configs_limit && cur.n > max_configs_limit) { verdict = -1; break; }
  }
  if (verdict == -3) { verdict = 1; free(rets); free(slot_op); free(cur.tab); free(nxt.tab); free(clo.tab); free(key); free(key2); free(wl); return 2; }

  out->valid = verdict;
  /* final config set, sorted */
  size_t m = cur.n, w = 0;
  uint64_t* all = (uint64_t*)malloc((m ? m : 1) * KW * 8);
  for (size_t j = 0; j < cur.cap; j++) if (cur.tab[j * KW] != 0) { memcpy(all + w * KW, cur.tab + j * KW, KW * 8); w++; }
  g_kw = KW;
  qsort(all, m, KW * 8, cmp_key_g);
  *n_final = (uint32_t)m;
  if (final_configs) memcpy(final_configs, all, (m < max_final ? m : max_final) * KW * 8);
  if (verdict == 1 && m) out->final_state = (int32_t)(uint32_t)(all[0] - 1);
  free(all);
  free(rets); free(slot_op); free(cur.tab); free(nxt.tab); free(clo.tab); free(key); free(key2); free(wl);
  return 0;
}
