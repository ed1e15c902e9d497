/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE (see oracle_model.h for the rules
 * and the PARITY UNPINNED statement).
 *
 * wgl_ref.c -- plain-C restatement of knossos.wgl/analysis as recalled in
 * SURVEY.md section 8a (rows `knossos.wgl.dll-history` and `knossos.wgl/analysis`):
 * the Wing-Gong search with Lowe's memoisation, exactly in the published
 * shape --
 *
 *   - a doubly-linked list of call and return entries in history order
 *     (crashed calls have no return entry), with lift!/unlift!;
 *   - a cache of (linearized bit set over ALL N ops, model state);
 *   - loop: call entry -> step the model; if consistent and the new
 *     (linearized, state) is not cached, push (entry, state), lift, restart
 *     from the head, else advance; return entry -> pop and unlift, or report
 *     "not linearizable" when the stack is empty.
 *
 * Deliberately NOT the windowed-key formulation the HIP kernel uses: the full
 * N-bit set is kept so this file is an independent check on the window
 * encoding.  (The cache hashes incrementally -- one 64-bit Zobrist word per
 * op -- but membership is decided by comparing the whole bit set, so it is
 * exact.)
 *
 * Two choices where the published algorithm leaves room, applied identically
 * by every implementation in this repo:
 *   1. the search stops as soon as every completed op is linearized (crashed
 *      calls left over are not linearized "for free" at the end);
 *   2. on failure the reported op is the completion with the greatest return
 *      rank that some config reached but none passed -- i.e. the first
 *      completion whose prefix has no linearization, which is what
 *      knossos.linear reports as :op (`Result map`).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "oracle_model.h"

typedef struct node {
  struct node *prev, *next, *match;
  uint32_t op;
  int is_call;
} node;

typedef struct cache {
  uint64_t* arena;     /* entries of (nw + 1) words: bits..., state */
  size_t n, cap_entries;
  uint32_t* slots;     /* index+1 into arena, 0 = empty */
  uint64_t* hashes;
  size_t nslots;       /* power of two */
  size_t nw;
} cache;

static uint64_t splitmix(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

static int cache_init(cache* c, size_t nw) {
  c->nw = nw; c->n = 0; c->cap_entries = 1024; c->nslots = 4096;
  c->arena = (uint64_t*)malloc(c->cap_entries * (nw + 1) * 8);
  c->slots = (uint32_t*)calloc(c->nslots, 4);
  c->hashes = (uint64_t*)malloc(c->cap_entries * 8);
  return c->arena && c->slots && c->hashes;
}
static void cache_free(cache* c) { free(c->arena); free(c->slots); free(c->hashes); }

static void cache_rehash(cache* c) {
  size_t ns = c->nslots * 2;
  uint32_t* s = (uint32_t*)calloc(ns, 4);
  for (size_t i = 0; i < c->n; i++) {
    size_t j = c->hashes[i] & (ns - 1);
    while (s[j]) j = (j + 1) & (ns - 1);
    s[j] = (uint32_t)(i + 1);
  }
  free(c->slots); c->slots = s; c->nslots = ns;
}

/* returns 1 if (bits,state) was newly added, 0 if already present */
static int cache_add(cache* c, const uint64_t* bits, int32_t state, uint64_t h) {
  h = splitmix(h ^ (uint64_t)(uint32_t)state * 0x9E3779B97F4A7C15ull);
  size_t j = h & (c->nslots - 1);
  while (c->slots[j]) {
    size_t e = c->slots[j] - 1;
    if (c->hashes[e] == h) {
      const uint64_t* p = c->arena + e * (c->nw + 1);
      if ((int32_t)p[c->nw] == state && memcmp(p, bits, c->nw * 8) == 0) return 0;
    }
    j = (j + 1) & (c->nslots - 1);
  }
  if (c->n == c->cap_entries) {
    c->cap_entries *= 2;
    c->arena = (uint64_t*)realloc(c->arena, c->cap_entries * (c->nw + 1) * 8);
    c->hashes = (uint64_t*)realloc(c->hashes, c->cap_entries * 8);
  }
  uint64_t* p = c->arena + c->n * (c->nw + 1);
  memcpy(p, bits, c->nw * 8);
  p[c->nw] = (uint64_t)(uint32_t)state;
  c->hashes[c->n] = h;
  c->slots[j] = (uint32_t)(c->n + 1);
  c->n++;
  if (c->n * 2 > c->nslots) cache_rehash(c);
  return 1;
}

typedef struct { uint32_t pos, op; } posop;
static int cmp_posop(const void* x, const void* y) {
  uint32_t a = ((const posop*)x)->pos, b = ((const posop*)y)->pos;
  return a < b ? -1 : a > b;
}

/*
 * ops sorted by inv_pos ascending, all positions distinct.
 * witness: caller buffer of n entries (may be NULL).
 * max_steps: 0 = unlimited.
 * returns 0 ok, nonzero on malformed input / OOM.
 */
int wgl_ref_check(uint32_t n, const uint8_t* f, const int32_t* a, const int32_t* b,
                  const uint32_t* inv_pos, const uint32_t* ret_pos,
                  const oracle_model* model, uint64_t max_steps,
                  uint32_t* witness, oracle_result* out) {
  memset(out, 0, sizeof *out);
  out->fail_op = out->prev_ok_op = 0xFFFFFFFFu;
  uint32_t R = 0;
  for (uint32_t i = 0; i < n; i++) {
    if (i && inv_pos[i] <= inv_pos[i - 1]) return 2;
    if (ret_pos[i] != O_CRASHED) { if (ret_pos[i] <= inv_pos[i]) return 2; R++; }
  }
  if (R == 0) { out->valid = 1; out->final_state = model->init; return 0; }

  /* entries in history order */
  posop* rets = (posop*)malloc(sizeof(posop) * R);
  uint32_t* ret_rank = (uint32_t*)malloc(4 * n);
  uint32_t* rank_op = (uint32_t*)malloc(4 * R);
  node* calls = (node*)calloc(n, sizeof(node));
  node* returns = (node*)calloc(n, sizeof(node));
  uint32_t k = 0;
  for (uint32_t i = 0; i < n; i++) if (ret_pos[i] != O_CRASHED) { rets[k].pos = ret_pos[i]; rets[k].op = i; k++; }
  qsort(rets, R, sizeof(posop), cmp_posop);
  for (uint32_t r = 0; r < R; r++) { ret_rank[rets[r].op] = r; rank_op[r] = rets[r].op; }
  node head; memset(&head, 0, sizeof head);
  node* tail = &head;
  uint32_t ci = 0, ri = 0;
  while (ci < n || ri < R) {
    node* e;
    if (ri >= R || (ci < n && inv_pos[ci] < rets[ri].pos)) {
      e = &calls[ci]; e->is_call = 1; e->op = ci;
      e->match = ret_pos[ci] != O_CRASHED ? &returns[ci] : NULL;
      ci++;
    } else {
      e = &returns[rets[ri].op]; e->is_call = 0; e->op = rets[ri].op; e->match = &calls[e->op];
      ri++;
    }
    e->prev = tail; e->next = NULL; tail->next = e; tail = e;
  }

  size_t nw = (n + 63) / 64;
  uint64_t* lin = (uint64_t*)calloc(nw, 8);
  cache c;
  if (!cache_init(&c, nw)) return 3;
  uint64_t linhash = 0;
  /* stack of (entry, state before) */
  node** stk_e = (node**)malloc(sizeof(node*) * n);
  int32_t* stk_s = (int32_t*)malloc(4 * n);
  uint32_t depth = 0, lifted_returns = 0;
  int32_t s = model->init;
  uint32_t maxf = 0;
  node* entry = head.next;
  int verdict = -2;

  while (verdict == -2) {
    if (entry == NULL) {            /* walked off the end: only crashed calls left */
      verdict = 1; break;
    }
    if (entry->is_call) {
      int32_t s2;
      uint32_t op = entry->op;
      if (oracle_step(model, s, f[op], a[op], b[op], &s2)) {
        out->steps++;
        if (max_steps && out->steps > max_steps) { verdict = -1; break; }
        lin[op >> 6] |= 1ull << (op & 63);
        uint64_t h2 = linhash ^ splitmix(op);
        out->probes++;
        if (cache_add(&c, lin, s2, h2)) {
          out->visited++;
          stk_e[depth] = entry; stk_s[depth] = s; depth++;
          if (depth > out->max_depth) out->max_depth = depth;
          s = s2; linhash = h2;
          /* lift! */
          entry->prev->next = entry->next;
          if (entry->next) entry->next->prev = entry->prev;
          node* m = entry->match;
          if (m) {
            m->prev->next = m->next;
            if (m->next) m->next->prev = m->prev;
            lifted_returns++;
            if (lifted_returns == R) { verdict = 1; break; }
          }
          entry = head.next;
        } else {
          lin[op >> 6] &= ~(1ull << (op & 63));
          entry = entry->next;
        }
      } else {
        entry = entry->next;
      }
    } else {
      uint32_t rk = ret_rank[entry->op];
      if (rk > maxf) maxf = rk;
      if (depth == 0) { verdict = 0; break; }
      depth--;
      out->backtracks++;
      node* e = stk_e[depth];
      s = stk_s[depth];
      uint32_t op = e->op;
      lin[op >> 6] &= ~(1ull << (op & 63));
      linhash ^= splitmix(op);
      /* unlift!: return first, then call */
      node* m = e->match;
      if (m) {
        m->prev->next = m;
        if (m->next) m->next->prev = m;
        lifted_returns--;
      }
      e->prev->next = e;
      if (e->next) e->next->prev = e;
      entry = e->next;
    }
  }

  out->valid = verdict;
  if (verdict == 1) {
    out->final_state = s;
    out->n_witness = depth;
    if (witness) for (uint32_t i = 0; i < depth; i++) witness[i] = stk_e[i]->op;
  } else if (verdict == 0) {
    out->fail_op = rank_op[maxf];
    out->prev_ok_op = maxf ? rank_op[maxf - 1] : 0xFFFFFFFFu;
  }
  cache_free(&c);
  free(lin); free(stk_e); free(stk_s); free(calls); free(returns);
  free(rets); free(ret_rank); free(rank_op);
  return 0;
}
