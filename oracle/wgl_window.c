/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE (see oracle_model.h for the rules
 * and the PARITY UNPINNED statement).
 *
 * wgl_window.c -- the same Wing-Gong/Lowe search as wgl_ref.c (same candidate
 * order, same cache semantics, therefore the same traversal, witness and
 * counters), restated with the WINDOWED config key the HIP kernel uses:
 *
 *     config = (front, mask over process slots, model state)
 *
 *   front  = rank of the first completion not yet linearized; every op that
 *            returned before it is linearized, no op invoked after it is;
 *   mask   = one bit per process: "the op this process has open at the front
 *            is already linearized" (a process has at most one open op, which
 *            is how knossos.linear.config indexes pending calls by process);
 *   state  = model state.
 *
 * (front, mask) is a bijective re-encoding of Lowe's N-bit linearized set, so
 * the cache has exactly the same membership as wgl_ref.c's.  This file is the
 * scalar CPU model of jepsen-tigerbeetle_amd/csrc/wgl_search.hip (per-slot
 * cursors, rank arithmetic, front advance) and the stronger of the two CPU
 * baselines bench.py reports.
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

typedef struct {
  uint64_t* tab;   /* entries of kw words; word0 low 32 bits = front+1 (0 = empty) */
  size_t cap, n, kw;
} wset;

static uint64_t key_hash(const uint64_t* k, size_t kw) {
  uint64_t h = mix64(k[0]);
  for (size_t i = 1; i < kw; i++) h = mix64(h ^ k[i]) + 0x9E3779B97F4A7C15ull;
  return h;
}
static int wset_add_nogrow(wset* s, const uint64_t* k) {
  size_t j = key_hash(k, s->kw) & (s->cap - 1);
  for (;;) {
    uint64_t* e = s->tab + j * s->kw;
    if ((uint32_t)e[0] == 0) { memcpy(e, k, s->kw * 8); s->n++; return 1; }
    if (memcmp(e, k, s->kw * 8) == 0) return 0;
    j = (j + 1) & (s->cap - 1);
  }
}
static int wset_add(wset* s, const uint64_t* k) {
  if ((s->n + 1) * 2 > s->cap) {
    wset t = {(uint64_t*)calloc(s->cap * 2 * s->kw, 8), s->cap * 2, 0, s->kw};
    for (size_t j = 0; j < s->cap; j++)
      if ((uint32_t)s->tab[j * s->kw] != 0) wset_add_nogrow(&t, s->tab + j * s->kw);
    free(s->tab); *s = t;
  }
  return wset_add_nogrow(s, k);
}

/* configs stuck at the failing completion of the LAST invalid run (test infrastructure: not re-entrant) */
/* (thread-local: oracle/many.c runs this restatement on a pthread pool, and an invalid history replaces the buffer) */
static _Thread_local uint64_t* g_cfg = NULL; static _Thread_local uint32_t g_cfg_n = 0, g_cfg_kw = 0;
static _Thread_local size_t g_sort_kw;
static int cmp_cfg(const void* x, const void* y) {
  const uint64_t* a = (const uint64_t*)x; const uint64_t* b = (const uint64_t*)y;
  int32_t sa = (int32_t)(a[0] >> 32), sb = (int32_t)(b[0] >> 32);
  if (sa != sb) return sa < sb ? -1 : 1;
  for (size_t i = 1; i < g_sort_kw; i++) if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
  return 0;
}
/* copies up to max rows of kw words (k0 = front+1 | state<<32, mask words), sorted by (state, mask); returns the total */
uint32_t wgl_window_last_configs(uint64_t* out, uint32_t max, uint32_t* kw) {
  *kw = g_cfg_kw;
  for (uint32_t i = 0; i < g_cfg_n && i < max; i++) memcpy(out + (size_t)i * g_cfg_kw, g_cfg + (size_t)i * g_cfg_kw, g_cfg_kw * 8);
  return g_cfg_n;
}

/*
 * Same contract as wgl_ref_check plus the process column (dense ids) and
 * n_process.  Returns 0 ok; 2 malformed history (unsorted, overlapping ops on
 * one process, op after a crash on the same process); 3 OOM.
 */
int wgl_window_check(uint32_t n, const uint8_t* f, const int32_t* a, const int32_t* b,
                     const int32_t* process, uint32_t n_process,
                     const uint32_t* inv_pos, const uint32_t* ret_pos,
                     const oracle_model* model, uint64_t max_steps,
                     uint32_t* witness, oracle_result* out) {
  memset(out, 0, sizeof *out);
  out->fail_op = out->prev_ok_op = 0xFFFFFFFFu;
  uint32_t R = 0;
  for (uint32_t i = 0; i < n; i++) {
    if (i && inv_pos[i] <= inv_pos[i - 1]) return 2;
    if (process[i] < 0 || (uint32_t)process[i] >= n_process) return 2;
    if (ret_pos[i] != O_CRASHED) { if (ret_pos[i] <= inv_pos[i]) return 2; R++; }
  }
  if (R == 0) { out->valid = 1; out->final_state = oracle_is_cfg_model(model) ? 0 : model->init; return 0; }

  const uint32_t W = n_process, MW = (W + 63) / 64, KW = 1 + MW;
  posop* rets = (posop*)malloc(sizeof(posop) * R);
  uint32_t* ret_rank = (uint32_t*)malloc(4 * n);
  uint32_t* inv_rank = (uint32_t*)malloc(4 * n);
  uint32_t* ret_op = (uint32_t*)malloc(4 * R);
  uint32_t k = 0;
  for (uint32_t i = 0; i < n; i++) if (ret_pos[i] != O_CRASHED) { rets[k].pos = ret_pos[i]; rets[k].op = i; k++; }
  qsort(rets, R, sizeof(posop), cmp_posop);
  for (uint32_t r = 0; r < R; r++) { ret_rank[rets[r].op] = r; ret_op[r] = rets[r].op; }
  { uint32_t r = 0;   /* inv_rank = #returns positioned before the invocation */
    for (uint32_t i = 0; i < n; i++) { while (r < R && rets[r].pos < inv_pos[i]) r++; inv_rank[i] = r; } }
  for (uint32_t i = 0; i < n; i++) if (ret_pos[i] == O_CRASHED) ret_rank[i] = 0xFFFFFFFFu;

  /* slot-major op lists (counting sort by process, stable in time) */
  uint32_t* seg = (uint32_t*)calloc(W + 1, 4);
  for (uint32_t i = 0; i < n; i++) seg[process[i] + 1]++;
  for (uint32_t p = 0; p < W; p++) seg[p + 1] += seg[p];
  uint32_t* fill = (uint32_t*)malloc(4 * W);
  memcpy(fill, seg, 4 * W);
  uint32_t* slot_ops = (uint32_t*)malloc(4 * (n ? n : 1));
  for (uint32_t i = 0; i < n; i++) slot_ops[fill[process[i]]++] = i;
  int bad = 0;
  for (uint32_t p = 0; p < W && !bad; p++)
    for (uint32_t j = seg[p]; j + 1 < seg[p + 1]; j++) {
      uint32_t x = slot_ops[j], y = slot_ops[j + 1];
      if (ret_pos[x] == O_CRASHED || ret_pos[x] >= inv_pos[y]) { bad = 1; break; }
    }
  if (bad) { free(rets); free(ret_rank); free(inv_rank); free(ret_op); free(seg); free(fill); free(slot_ops); return 2; }

  /* cursor[p] = index into slot_ops of the last op of p with inv_rank <= front, or seg[p]-1 */
  int64_t* cursor = (int64_t*)malloc(8 * W);
  for (uint32_t p = 0; p < W; p++) cursor[p] = (int64_t)seg[p] - 1;

  /* frames */
  uint32_t* fr_fi = (uint32_t*)malloc(4 * n);
  int32_t* fr_s = (int32_t*)malloc(4 * n);
  uint32_t* fr_op = (uint32_t*)malloc(4 * n);
  uint64_t* fr_m = (uint64_t*)malloc(8 * (size_t)n * MW);
  uint64_t* M = (uint64_t*)calloc(MW, 8);
  uint64_t* M2 = (uint64_t*)calloc(MW, 8);
  uint64_t* key = (uint64_t*)calloc(KW, 8);
  wset vs = {(uint64_t*)calloc((size_t)4096 * KW, 8), 4096, 0, KW};

  uint32_t fi = 0, depth = 0, maxf = 0;
  const int cfgm = oracle_is_cfg_model(model);
  uint32_t* open_ops = (uint32_t*)malloc(4 * (W + 1));
  uint8_t* open_lin = (uint8_t*)malloc(W + 1);
  int32_t s = cfgm ? 0 : model->init;
  int64_t from = -1;        /* only candidates with op index > from */
  int verdict = -2;

  while (verdict == -2) {
    /* bring cursors to the front (forward or backward) */
    for (uint32_t p = 0; p < W; p++) {
      while (cursor[p] + 1 < (int64_t)seg[p + 1] && inv_rank[slot_ops[cursor[p] + 1]] <= fi) cursor[p]++;
      while (cursor[p] >= (int64_t)seg[p] && inv_rank[slot_ops[cursor[p]]] > fi) cursor[p]--;
    }
    if (fi > maxf) maxf = fi;
    int descended = 0;
    uint32_t n_open = 0xFFFFFFFFu;
    for (;;) {
      /* first candidate in invocation order after `from` */
      int64_t best = -1; uint32_t bestp = 0; int32_t best_s2 = 0;
      for (uint32_t p = 0; p < W; p++) {
        if (cursor[p] < (int64_t)seg[p]) continue;
        uint32_t op = slot_ops[cursor[p]];
        if (ret_rank[op] < fi) continue;                 /* already returned (and linearized) */
        if (M[p >> 6] >> (p & 63) & 1) continue;         /* linearized */
        if ((int64_t)op <= from) continue;
        if (best >= 0 && (int64_t)op > best) continue;
        int32_t s2;
        if (cfgm) {
          if (n_open == 0xFFFFFFFFu) {          /* open calls of this config, gathered once per node */
            n_open = 0;
            for (uint32_t pp = 0; pp < W; pp++) {
              if (cursor[pp] < (int64_t)seg[pp]) continue;
              uint32_t x = slot_ops[cursor[pp]];
              if (ret_rank[x] < fi) continue;
              open_ops[n_open] = x; open_lin[n_open] = (uint8_t)(M[pp >> 6] >> (pp & 63) & 1); n_open++;
            }
          }
          if (!oracle_cfg_step(model, fi, open_ops, open_lin, n_open, f, a, op)) continue;
          s2 = 0;
        } else if (!oracle_step(model, s, f[op], a[op], b[op], &s2)) continue;
        best = op; bestp = p; best_s2 = s2;
      }
      if (best < 0) break;
      uint32_t op = (uint32_t)best;
      out->steps++;
      if (max_steps && out->steps > max_steps) { verdict = -1; break; }
      /* child config */
      memcpy(M2, M, 8 * MW);
      M2[bestp >> 6] |= 1ull << (bestp & 63);
      uint32_t fi2 = fi;
      if (ret_rank[op] == fi) {
        uint32_t p = bestp;
        for (;;) {
          M2[p >> 6] &= ~(1ull << (p & 63));
          fi2++;
          if (fi2 == R) break;
          p = (uint32_t)process[ret_op[fi2]];
          if (!(M2[p >> 6] >> (p & 63) & 1)) break;
        }
      }
      key[0] = (uint64_t)(fi2 + 1) | ((uint64_t)(uint32_t)best_s2 << 32);
      memcpy(key + 1, M2, 8 * MW);
      out->probes++;
      if (wset_add(&vs, key)) {
        out->visited++;
        fr_fi[depth] = fi; fr_s[depth] = s; fr_op[depth] = op;
        memcpy(fr_m + (size_t)depth * MW, M, 8 * MW);
        depth++;
        if (depth > out->max_depth) out->max_depth = depth;
        fi = fi2; s = best_s2; memcpy(M, M2, 8 * MW); from = -1;
        if (fi == R) verdict = 1;
        descended = 1;
        break;
      }
      from = best;   /* seen: next entry */
    }
    if (verdict != -2 || descended) continue;
    /* hit the front's return entry: backtrack */
    if (depth == 0) { verdict = 0; break; }
    depth--; out->backtracks++;
    fi = fr_fi[depth]; s = fr_s[depth]; from = fr_op[depth];
    memcpy(M, fr_m + (size_t)depth * MW, 8 * MW);
  }

  out->valid = verdict;
  if (verdict == 1) {
    out->final_state = s; out->n_witness = depth;
    if (witness) memcpy(witness, fr_op, 4 * (size_t)depth);
  } else if (verdict == 0) {
    out->fail_op = ret_op[maxf];
    out->prev_ok_op = maxf ? ret_op[maxf - 1] : 0xFFFFFFFFu;
    free(g_cfg); g_cfg = NULL; g_cfg_n = 0; g_cfg_kw = KW;
    size_t cnt = 0;
    for (size_t j = 0; j < vs.cap; j++) if ((uint32_t)vs.tab[j * KW] == maxf + 1) cnt++;
    g_cfg = (uint64_t*)malloc((cnt ? cnt : 1) * KW * 8);
    for (size_t j = 0; j < vs.cap; j++)
      if ((uint32_t)vs.tab[j * KW] == maxf + 1) { memcpy(g_cfg + (size_t)g_cfg_n * KW, vs.tab + j * KW, KW * 8); g_cfg_n++; }
    g_sort_kw = KW;
    qsort(g_cfg, g_cfg_n, KW * 8, cmp_cfg);
  }
  free(rets); free(ret_rank); free(inv_rank); free(ret_op); free(seg); free(fill); free(slot_ops);
  free(open_ops); free(open_lin);
  free(cursor); free(fr_fi); free(fr_s); free(fr_op); free(fr_m); free(M); free(M2); free(key); free(vs.tab);
  return 0;
}
