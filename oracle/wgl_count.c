/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE (see oracle_model.h for the rules
 * and the PARITY UNPINNED statement).
 *
 * wgl_count.c -- scalar CPU restatement of the COUNT FORM of the wide schedule
 * (wgl_beam.c) for register / cas-register histories with crashed (:info) calls;
 * the schedule jepsen-tigerbeetle_amd/csrc/wgl_beam.hip runs when the batch is in
 * count form (tbc_internal.h, kRuleCount).  Same verdict and failing op as the
 * sequential search (wgl_ref.c / wgl_window.c) -- they are properties of
 * (model, history) -- over a config space that does not grow with the NUMBER of
 * crashed calls:
 *
 *   A crashed call never completes: it may take effect at any time after its
 *   invocation, or never (Knossos semantics).  In the mask
 *   form every crashed call holds a process slot for ever (one more mask bit
 *   each) and every subset of them is a different config.  Here:
 *
 *   1. CLASSES.  Crashed calls are grouped by effect: (:write v), (:cas [a b]) with
 *      a != b.  Crashed reads and crashed (:cas [a a]) have no effect on the model
 *      and constrain nothing: they are never candidates (as crashed reads already
 *      are not in wgl_beam.c).  Calls of one class are interchangeable once
 *      invoked, so they are linearized in invocation order (the twin rule among
 *      crashed calls) and a config records, per class, HOW MANY are linearized:
 *      a count vector C instead of a mask bit per call.  The k-th call of a class
 *      (0-based) is available at front F iff it was invoked by then
 *      (inv_rank <= F).  Counts are packed into 128 bits, class c in a field of
 *      bit_length(n_c) bits (a field never straddles a 64-bit word).
 *   2. PROCESS SLOTS are re-used: a live call takes the lowest slot that is free
 *      when its process first invokes; a process that crashes hands its slot
 *      back (its crashed call needs none).  The mask stays as wide as the
 *      number of processes ALIVE at once (the worker threads), whatever the
 *      number of crashes.
 *   3. LAZY RULE ("hot" configs).  A crashed call has no completion, so it is
 *      only ever needed as the PRODUCER of a value some other call observes.  In
 *      any linearization every crashed call whose value is overwritten unobserved
 *      can be deleted, and one whose value is observed can be moved to just
 *      before its first observer (only nil reads lie between, and they commute).
 *      So, without loss of generality, the call linearized right after a crashed
 *      call OBSERVES its value: a read of exactly that value (absorbed at once by
 *      the eager-read rule) or a :cas expecting it (live or crashed).  A config
 *      reached by a crashed call whose value no absorbed read observed is HOT:
 *      only calls whose precondition is the state are candidates in it.  A
 *      crashed :write of the current state is never a candidate (pointless).
 *   4. PARETO RULE.  Crashed calls left over are never an obligation, only a
 *      resource: of two configs with the same (front, mask, state, hot) the one
 *      that has used fewer crashed calls of every class can do whatever the other
 *      can.  A new config is dropped when a config with the same key and a count
 *      vector <= its own (field by field) was visited before (equal included:
 *      that is plain memoisation); pairs of a round are looked at in pair order.
 *
 * Everything else is wgl_beam.c's schedule with eager reads and the twin rule
 * among live calls: pop the K most recent configs, pairs parent-bottom-first and
 * within a parent from its LAST candidate to its first, RP pairs per round,
 * success = lowest pair whose child passed every completion, new configs pushed
 * in pair order, lookahead (dead configs set aside, taken up again only if the
 * search would otherwise end invalid).  Candidates of a config = the live calls
 * open at its front in slot order, then the classes in order of their first
 * invocation, as far as a member is invoked by the front.  The lookahead counts
 * a crashed producer of the needed value as available from its invocation on,
 * whatever the counts (conservative: a dead config is dead).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "oracle_model.h"

typedef struct count_stats {
  uint64_t iterations, probes, visited, expanded, max_stack, rounds;
  uint64_t dominated;      /* viable children dropped by the Pareto rule (equal keys included) */
  uint64_t hot;            /* hot configs inserted */
  uint64_t class_steps;    /* configs inserted that were reached by a crashed call */
  uint32_t n_slots, n_classes, count_bits, n_crashed;
} count_stats;

typedef struct { uint32_t pos, op; } cposop;
static int cmp_cposop(const void* x, const void* y) {
  uint32_t a = ((const cposop*)x)->pos, b = ((const cposop*)y)->pos;
  return a < b ? -1 : a > b;
}
static uint64_t cmix64(uint64_t x) {
  x ^= x >> 33; x *= 0xFF51AFD7ED558CCDull;
  x ^= x >> 33; x *= 0xC4CEB9FE1A85EC53ull;
  x ^= x >> 33; return x;
}

#define HOT_BIT 0x40000000u
#define CW 2u                       /* count words */

typedef struct {
  uint32_t f; int32_t a, b;         /* the effect */
  uint32_t n, shift, width;         /* members, field position in the 128-bit count vector */
  uint32_t first;                   /* index of its first member in mem_rank[] / mem_op[] */
} cclass;

/* configs: KW = 1 + MW + CW words each (k0, M[], C[]), parent, op; hash on (k0, M) -> chain of configs with that key */
typedef struct {
  uint64_t* keys; uint32_t* parent; uint32_t* op; uint32_t* next_same; size_t n, cap, kw, hw;
  uint32_t* slots; size_t nslots;
} carena;

static uint64_t ckey_hash(const uint64_t* k, size_t hw) {
  uint64_t h = cmix64(k[0]);
  for (size_t i = 1; i < hw; i++) h = cmix64(h ^ k[i]) + 0x9E3779B97F4A7C15ull;
  return h;
}
static void carena_rehash(carena* a) {
  size_t ns = a->nslots * 2;
  uint32_t* s = (uint32_t*)calloc(ns, 4);
  for (size_t j0 = 0; j0 < a->nslots; j0++) {
    const uint32_t head = a->slots[j0];
    if (!head) continue;
    size_t j = ckey_hash(a->keys + (size_t)head * a->kw, a->hw) & (ns - 1);
    while (s[j]) j = (j + 1) & (ns - 1);
    s[j] = head;
  }
  free(a->slots); a->slots = s; a->nslots = ns;
}
/* field-wise x >= y over the packed count vectors; H = the top bit of every field */
static int counts_ge(const uint64_t* x, const uint64_t* y, const uint64_t* H) {
  for (uint32_t w = 0; w < CW; w++) {
    const uint64_t t = (x[w] | H[w]) - (y[w] & ~H[w]);
    const uint64_t ge = ((x[w] & ~y[w]) | (~(x[w] ^ y[w]) & t)) & H[w];
    if (ge != H[w]) return 0;
  }
  return 1;
}
/* returns the new index, or 0 if a visited config with this key and counts <= k's dominates it */
static uint32_t carena_add(carena* a, const uint64_t* k, const uint64_t* H, uint32_t parent, uint32_t op) {
  size_t j = ckey_hash(k, a->hw) & (a->nslots - 1);
  uint32_t head = 0;
  while (a->slots[j]) {
    if (memcmp(a->keys + (size_t)a->slots[j] * a->kw, k, a->hw * 8) == 0) { head = a->slots[j]; break; }
    j = (j + 1) & (a->nslots - 1);
  }
  for (uint32_t e = head; e; e = a->next_same[e])
    if (counts_ge(k + a->hw, a->keys + (size_t)e * a->kw + a->hw, H)) return 0;
  if (a->n == a->cap) {
    a->cap *= 2;
    a->keys = (uint64_t*)realloc(a->keys, a->cap * a->kw * 8);
    a->parent = (uint32_t*)realloc(a->parent, a->cap * 4);
    a->op = (uint32_t*)realloc(a->op, a->cap * 4);
    a->next_same = (uint32_t*)realloc(a->next_same, a->cap * 4);
  }
  const uint32_t id = (uint32_t)a->n++;
  memcpy(a->keys + (size_t)id * a->kw, k, a->kw * 8);
  a->parent[id] = parent; a->op[id] = op;
  a->next_same[id] = head;                 /* newest first; the slot always points at the newest */
  a->slots[j] = id;
  if (!head && a->n * 2 > a->nslots) carena_rehash(a);
  return id;
}

/* what the library does with a history (so that tests can compare): slot of every op (0xFFFFFFFF = crashed: none) */
static _Thread_local uint32_t* g_slots_out = NULL;
void wgl_count_set_slots_out(uint32_t* buf) { g_slots_out = buf; }

/* where the step limit is looked at: between iterations (the wide kernel, wgl_beam.hip) or after every round (the narrow kernel,
 * wgl_narrow_impl.h: "as after the round that exceeded it") */
static uint32_t g_limit_per_round = 0;
void wgl_count_set_limit_per_round(uint32_t on) { g_limit_per_round = on; }

static uint32_t get_count(const uint64_t* C, const cclass* c) {
  return (uint32_t)((C[c->shift >> 6] >> (c->shift & 63u)) & ((1ull << c->width) - 1ull));
}

/* eager reads over the live lists (as wgl_beam.c): returns the new front; *observed is set when a read of exactly
 * the state (not nil) was absorbed */
static uint32_t cabsorb(uint64_t* M2, uint32_t fi2, int32_t s, uint32_t R, const uint32_t* off, const uint32_t* lst,
                        const uint32_t* slot, const uint32_t* ret_op, const uint8_t* f, const int32_t* a,
                        int* observed, uint32_t* wit, uint32_t* nw) {
  int again = 1;
  while (again && fi2 < R) {
    again = 0;
    for (uint32_t x0 = off[fi2]; x0 < off[fi2 + 1]; x0++) {
      const uint32_t x = lst[x0], px = slot[x];
      if (M2[px >> 6] >> (px & 63) & 1) continue;
      if (f[x] != O_READ || !(a[x] == O_NIL || a[x] == s)) continue;
      M2[px >> 6] |= 1ull << (px & 63);
      if (a[x] != O_NIL && observed) *observed = 1;
      if (wit) wit[(*nw)++] = x;
    }
    uint32_t pp = slot[ret_op[fi2]];
    while (M2[pp >> 6] >> (pp & 63) & 1) {
      M2[pp >> 6] &= ~(1ull << (pp & 63));
      fi2++; again = 1;
      if (fi2 == R) break;
      pp = slot[ret_op[fi2]];
    }
  }
  return fi2;
}

/* list order (csrc PackOpenArgs.list_order; oracle/wgl_beam.c has the same switch): 0 = a front's live calls in slot order, 1 = in order of completion,
   16 + W = in order of completion with a :write as if it completed W ranks later (a study knob here) */
static uint32_t g_count_list_order = 0;
void wgl_count_set_list_order(uint32_t o) { g_count_list_order = o; }

int wgl_count_check(uint32_t n, const uint8_t* f, const int32_t* a, const int32_t* b,
                    const int32_t* process, uint32_t n_process,
                    const uint32_t* inv_pos, const uint32_t* ret_pos,
                    const oracle_model* model, uint32_t K, uint32_t round_pairs, uint64_t max_probes,
                    uint32_t lookahead, uint32_t relaxed, uint32_t target,
                    uint32_t* witness, oracle_result* out, count_stats* st) {
  memset(out, 0, sizeof *out); memset(st, 0, sizeof *st);
  out->fail_op = out->prev_ok_op = 0xFFFFFFFFu;
  if (K == 0 || K > 64 || round_pairs == 0 || round_pairs > 1024) return 1;
  if (model->kind != O_REGISTER && model->kind != O_CAS_REGISTER) return 3;
  const uint32_t RP = round_pairs;
  uint32_t R = 0;
  for (uint32_t i = 0; i < n; i++) {
    if (i && inv_pos[i] <= inv_pos[i - 1]) return 2;
    if (process[i] < 0 || (uint32_t)process[i] >= n_process) return 2;
    if (ret_pos[i] != O_CRASHED) { if (ret_pos[i] <= inv_pos[i]) return 2; R++; }
    /* the rules need small register values (as the library: 0..30) */
    if (a[i] != O_NIL && (a[i] < 0 || a[i] > 30)) return 3;
    if (f[i] == O_CAS && (b[i] < 0 || b[i] > 30)) return 3;
  }
  if (model->init != O_NIL && (model->init < 0 || model->init > 30)) return 3;
  if (R == 0) { out->valid = 1; out->final_state = model->init; return 0; }
  /* the search ends VALID when a config has passed the first RT completions (5.: RT < R checks a prefix) */
  const uint32_t RT = (target && target < R) ? target : R;

  /* ---- ranks */
  cposop* rets = (cposop*)malloc(sizeof(cposop) * R);
  uint32_t* ret_rank = (uint32_t*)malloc(4 * (size_t)n);
  uint32_t* inv_rank = (uint32_t*)malloc(4 * (size_t)n);
  uint32_t* ret_op = (uint32_t*)malloc(4 * (size_t)R);
  { uint32_t k = 0;
    for (uint32_t i = 0; i < n; i++) if (ret_pos[i] != O_CRASHED) { rets[k].pos = ret_pos[i]; rets[k].op = i; k++; } }
  qsort(rets, R, sizeof(cposop), cmp_cposop);
  for (uint32_t r = 0; r < R; r++) { ret_rank[rets[r].op] = r; ret_op[r] = rets[r].op; }
  { uint32_t r = 0;
    for (uint32_t i = 0; i < n; i++) { while (r < R && rets[r].pos < inv_pos[i]) r++; inv_rank[i] = r; } }
  for (uint32_t i = 0; i < n; i++) if (ret_pos[i] == O_CRASHED) ret_rank[i] = 0xFFFFFFFFu;

  /* ---- process slots, re-used (2.) */
  uint32_t* slot = (uint32_t*)malloc(4 * ((size_t)n + 1));
  uint32_t W = 1;
  { int32_t* slot_of = (int32_t*)malloc(4 * ((size_t)n_process + 1));
    uint8_t* used = (uint8_t*)calloc((size_t)n_process + 1, 1);
    for (uint32_t p = 0; p < n_process; p++) slot_of[p] = -1;
    for (uint32_t i = 0; i < n; i++) {
      const uint32_t p = (uint32_t)process[i];
      if (ret_pos[i] == O_CRASHED) {
        if (slot_of[p] >= 0) { used[slot_of[p]] = 0; slot_of[p] = -1; }
        slot[i] = 0xFFFFFFFFu;
        continue;
      }
      if (slot_of[p] < 0) {
        uint32_t s = 0;
        while (used[s]) s++;
        used[s] = 1; slot_of[p] = (int32_t)s;
        if (s + 1 > W) W = s + 1;
      }
      slot[i] = (uint32_t)slot_of[p];
    }
    free(slot_of); free(used); }
  if (g_slots_out) memcpy(g_slots_out, slot, 4 * (size_t)n);
  const uint32_t MW = (W + 63) / 64, HW = 1 + MW, KW = HW + CW;
  st->n_slots = W;

  /* ---- classes of crashed calls (1.) */
  cclass* cls = (cclass*)calloc((size_t)n + 1, sizeof(cclass));
  uint32_t* cls_of = (uint32_t*)malloc(4 * ((size_t)n + 1));          /* class of a crashed candidate op, else 0xFFFFFFFF */
  uint32_t ncls = 0, n_cr = 0;
  for (uint32_t i = 0; i < n; i++) {
    cls_of[i] = 0xFFFFFFFFu;
    if (ret_pos[i] != O_CRASHED) continue;
    if (!(f[i] == O_WRITE || (f[i] == O_CAS && model->kind == O_CAS_REGISTER && a[i] != b[i]))) continue;
    uint32_t c = 0;
    while (c < ncls && !(cls[c].f == f[i] && cls[c].a == a[i] && (f[i] != O_CAS || cls[c].b == b[i]))) c++;
    if (c == ncls) { cls[c].f = f[i]; cls[c].a = a[i]; cls[c].b = f[i] == O_CAS ? b[i] : 0; cls[c].n = 0; ncls++; }
    cls[c].n++; cls_of[i] = c; n_cr++;
  }
  uint32_t* mem_rank = (uint32_t*)malloc(4 * ((size_t)n_cr + 1));
  uint32_t* mem_op = (uint32_t*)malloc(4 * ((size_t)n_cr + 1));
  { uint32_t run = 0, bits = 0;
    for (uint32_t c = 0; c < ncls; c++) {
      cls[c].first = run; run += cls[c].n;
      uint32_t w = 0; while ((1u << w) <= cls[c].n) w++;
      if ((bits & 63u) + w > 64u) bits = (bits + 63u) & ~63u;      /* a field never straddles a word */
      cls[c].shift = bits; cls[c].width = w; bits += w;
    }
    st->count_bits = bits;
    if (bits > 64u * CW) {
      free(rets); free(ret_rank); free(inv_rank); free(ret_op); free(slot); free(cls); free(cls_of); free(mem_rank); free(mem_op);
      return 3;
    }
    uint32_t* fill = (uint32_t*)calloc((size_t)ncls + 1, 4);
    for (uint32_t i = 0; i < n; i++) if (cls_of[i] != 0xFFFFFFFFu) {
      const uint32_t c = cls_of[i], at = cls[c].first + fill[c]++;
      mem_rank[at] = inv_rank[i]; mem_op[at] = i;
    }
    free(fill); }
  st->n_classes = ncls; st->n_crashed = n_cr;
  uint64_t H[CW] = {0, 0};
  for (uint32_t c = 0; c < ncls; c++) { const uint32_t t = cls[c].shift + cls[c].width - 1; H[t >> 6] |= 1ull << (t & 63u); }
  /* classes available at front F: those whose first member is invoked by then (classes are in that order) */
  uint32_t* ncr = (uint32_t*)calloc((size_t)R + 1, 4);
  for (uint32_t c = 0; c < ncls; c++) { const uint32_t r0 = mem_rank[cls[c].first]; if (r0 < R) ncr[r0]++; }
  for (uint32_t r = 1; r < R; r++) ncr[r] += ncr[r - 1];

  /* ---- per-front lists of the live calls, in slot order */
  uint32_t* off = (uint32_t*)calloc((size_t)R + 1, 4);
  for (uint32_t i = 0; i < n; i++) if (ret_rank[i] != 0xFFFFFFFFu)
    for (uint32_t fr = inv_rank[i]; fr <= ret_rank[i]; fr++) off[fr + 1]++;
  for (uint32_t r = 0; r < R; r++) off[r + 1] += off[r];
  uint32_t* lst = (uint32_t*)malloc(4 * ((size_t)off[R] + 1));
  { uint32_t* fill = (uint32_t*)malloc(4 * ((size_t)R + 1));
    memcpy(fill, off, 4 * ((size_t)R + 1));
    for (uint32_t i = 0; i < n; i++) if (ret_rank[i] != 0xFFFFFFFFu)
      for (uint32_t fr = inv_rank[i]; fr <= ret_rank[i]; fr++) lst[fill[fr]++] = i;
    free(fill);
    for (uint32_t fr = 0; fr < R; fr++)
      for (uint32_t x = off[fr] + 1; x < off[fr + 1]; x++) {
        uint32_t v = lst[x], y = x;
#define COUNT_KEY(o) (g_count_list_order == 1 ? ret_rank[o] : g_count_list_order >= 16 ? \
                      2u * ret_rank[o] + (f[o] == O_WRITE ? 2u * (g_count_list_order - 16u) + 1u : 0u) : (uint32_t)slot[o])      /* 16 + W: as wgl_beam.c */
        while (y > off[fr] && COUNT_KEY(lst[y - 1]) > COUNT_KEY(v)) { lst[y] = lst[y - 1]; y--; }
        lst[y] = v;
      } }

  carena ar; ar.kw = KW; ar.hw = HW; ar.cap = 4096; ar.n = 1; ar.nslots = 8192;
  ar.keys = (uint64_t*)calloc(ar.cap * KW, 8); ar.parent = (uint32_t*)calloc(ar.cap, 4); ar.op = (uint32_t*)calloc(ar.cap, 4);
  ar.next_same = (uint32_t*)calloc(ar.cap, 4);
  ar.slots = (uint32_t*)calloc(ar.nslots, 4);
  size_t scap = 1 << 16, sp = 0, dcap = 1 << 12, dsp = 0;
  uint32_t* stack = (uint32_t*)malloc(scap * 4);
  uint32_t* dstack = (uint32_t*)malloc(dcap * 4);
  int look_on = lookahead != 0;
  uint64_t* key = (uint64_t*)calloc(KW, 8);
  key[0] = 1ull | ((uint64_t)(uint32_t)model->init << 32);
  uint32_t maxf = 0;
  int verdict = -2;
  uint32_t win_parent = 0, win_op = 0; int32_t win_state = 0;
  stack[sp++] = carena_add(&ar, key, H, 0, 0xFFFFFFFFu);
  st->visited = 1; st->max_stack = 1;

  uint32_t par[64], pcnt[64], pstart[65];
  uint64_t* ck = (uint64_t*)malloc((size_t)RP * KW * 8);
  uint32_t cop[1024], cpar[1024]; int cviable[1024]; uint32_t cfront[1024]; int32_t cstate[1024]; int chot[1024], cclassstep[1024];

  while (verdict == -2) {
    if (sp == 0) {
      if (dsp == 0) { verdict = 0; break; }
      { uint32_t* t = stack; stack = dstack; dstack = t; size_t c = scap; scap = dcap; dcap = c; }
      sp = dsp; dsp = 0; look_on = 0;
    }
    uint32_t np = sp < K ? (uint32_t)sp : K;
    for (uint32_t q = 0; q < np; q++) par[q] = stack[sp - np + q];
    sp -= np;
    st->iterations++; st->expanded += np;
    uint32_t T = 0;
    for (uint32_t q = 0; q < np; q++) {
      const uint32_t fi = (uint32_t)ar.keys[(size_t)par[q] * KW] - 1;
      pcnt[q] = (off[fi + 1] - off[fi]) + ncr[fi];
      pstart[q] = T; T += pcnt[q];
    }
    pstart[np] = T;
    for (uint32_t base = 0; base < T && verdict == -2; base += RP) {
      const uint32_t m = T - base < RP ? T - base : RP;
      int success = -1;
      for (uint32_t l = 0; l < m; l++) {
        uint32_t r = base + l, q = 0;
        while (pstart[q + 1] <= r) q++;
        const uint64_t* pk = ar.keys + (size_t)par[q] * KW;
        const uint32_t fi = (uint32_t)pk[0] - 1;
        const uint32_t se = (uint32_t)(pk[0] >> 32);
        const int hot = (se & HOT_BIT) != 0;
        const int32_t s = (int32_t)(se & ~HOT_BIT);
        const uint32_t nlive = off[fi + 1] - off[fi];
        const uint32_t c = pcnt[q] - 1 - (r - pstart[q]);
        cviable[l] = 0; cpar[l] = par[q]; cop[l] = 0xFFFFFFFFu;
        uint64_t* c2 = ck + (size_t)l * KW;
        memcpy(c2, pk, KW * 8);
        int32_t s2; uint32_t fi2 = fi; int observed = 0, class_step = 0;
        if (c < nlive) {
          const uint32_t op = lst[off[fi] + c], p = slot[op];
          cop[l] = op;
          if (pk[1 + (p >> 6)] >> (p & 63) & 1) continue;
          if (f[op] == O_WRITE || f[op] == O_CAS) {                /* twin rule among the live calls */
            int dominated = 0;
            for (uint32_t cc = 0; cc < nlive && !dominated; cc++) {
              const uint32_t y = lst[off[fi] + cc], py = slot[y];
              if (y == op || f[y] != f[op] || a[y] != a[op] || (f[op] == O_CAS && b[y] != b[op])) continue;
              if (pk[1 + (py >> 6)] >> (py & 63) & 1) continue;
              if (ret_rank[y] < ret_rank[op]) dominated = 1;
            }
            if (dominated) continue;
          }
          if (!oracle_step(model, s, f[op], a[op], b[op], &s2)) continue;
          /* a hot config only takes calls whose precondition is its state */
          if (hot && !((f[op] == O_READ && a[op] == s) || (f[op] == O_CAS && a[op] == s))) continue;
          c2[1 + (p >> 6)] |= 1ull << (p & 63);
          if (ret_rank[op] == fi) {
            uint32_t pp = p;
            for (;;) {
              c2[1 + (pp >> 6)] &= ~(1ull << (pp & 63));
              fi2++;
              if (fi2 == R) break;
              pp = slot[ret_op[fi2]];
              if (!(c2[1 + (pp >> 6)] >> (pp & 63) & 1)) break;
            }
          }
          fi2 = cabsorb(c2 + 1, fi2, s2, R, off, lst, slot, ret_op, f, a, NULL, NULL, NULL);
          observed = 1;
        } else {
          const cclass* cc = &cls[c - nlive];
          const uint32_t k = relaxed ? 0u : get_count(pk + HW, cc);
          if (k >= cc->n || mem_rank[cc->first + k] > fi) continue;       /* its next member is not invoked yet (or none is left) */
          if (cc->f == O_WRITE) { if (hot || cc->a == s) continue; s2 = cc->a; }
          else { if (cc->a != s) continue; s2 = cc->b; }
          cop[l] = mem_op[cc->first + k];
          if (!relaxed) c2[HW + (cc->shift >> 6)] += 1ull << (cc->shift & 63u);
          fi2 = cabsorb(c2 + 1, fi2, s2, R, off, lst, slot, ret_op, f, a, &observed, NULL, NULL);
          class_step = 1;
        }
        c2[0] = (uint64_t)(fi2 + 1) | ((uint64_t)((uint32_t)s2 | (observed ? 0u : HOT_BIT)) << 32);
        cviable[l] = 1; cfront[l] = fi2; cstate[l] = s2; chot[l] = !observed; cclassstep[l] = class_step;
        if (fi2 >= RT && success < 0) success = (int)l;
      }
      st->rounds++;
      if (success >= 0) { verdict = 1; win_parent = cpar[success]; win_op = cop[success]; win_state = cstate[success]; break; }
      for (uint32_t l = 0; l < m; l++) {
        if (!cviable[l]) continue;
        st->probes++;
        const uint64_t* c2 = ck + (size_t)l * KW;
        const uint32_t id = carena_add(&ar, c2, H, cpar[l], cop[l]);
        if (!id) { st->dominated++; continue; }
        st->visited++; st->hot += (uint64_t)chot[l]; st->class_steps += (uint64_t)cclassstep[l];
        if (cfront[l] > maxf) maxf = cfront[l];
        if (look_on) {
          /* as wgl_beam.c; a crashed producer of the needed value counts from its invocation on, whatever the counts */
          const uint32_t F = cfront[l]; const int32_t s2 = cstate[l];
          int dead = 0;
          for (uint32_t j = 0; j < 8 && F + j < RT && !dead; j++) {   /* (completions past a prefix target constrain nothing) */
            const uint32_t t = F + j, fop = ret_op[t], pf = slot[fop];
            if (!((f[fop] == O_READ && a[fop] != O_NIL) || f[fop] == O_CAS)) continue;
            const int32_t v = a[fop];
            if (inv_rank[fop] <= F && (c2[1 + (pf >> 6)] >> (pf & 63) & 1)) continue;
            if (v == s2) continue;
            int ok = 0;
            for (uint32_t F2 = F; F2 <= t && !ok; F2++)
              for (uint32_t x0 = off[F2]; x0 < off[F2 + 1] && !ok; x0++) {
                const uint32_t x = lst[x0], px = slot[x];
                if (x == fop) continue;
                if (inv_rank[x] <= F && (c2[1 + (px >> 6)] >> (px & 63) & 1)) continue;
                if ((f[x] == O_WRITE && a[x] == v) || (f[x] == O_CAS && b[x] == v)) ok = 1;
              }
            for (uint32_t cc = 0; cc < ncls && !ok; cc++)
              if (mem_rank[cls[cc].first] <= t && ((cls[cc].f == O_WRITE && cls[cc].a == v) || (cls[cc].f == O_CAS && cls[cc].b == v))) ok = 1;
            if (!ok) dead = 1;
          }
          if (dead) {
            if (dsp == dcap) { dcap *= 2; dstack = (uint32_t*)realloc(dstack, dcap * 4); }
            dstack[dsp++] = id;
            continue;
          }
        }
        if (sp == scap) { scap *= 2; stack = (uint32_t*)realloc(stack, scap * 4); }
        stack[sp++] = id;
      }
      if (g_limit_per_round && max_probes && st->probes > max_probes) { verdict = -1; break; }
    }
    if (sp > st->max_stack) st->max_stack = sp;
    /* the step limit is looked at between iterations (as the wide kernel does) */
    if (verdict == -2 && max_probes && st->probes > max_probes) { verdict = -1; break; }
  }

  out->valid = verdict;
  out->steps = st->probes; out->probes = st->probes; out->visited = st->visited;
  out->backtracks = st->expanded; out->max_depth = st->max_stack;
  if (verdict == 1) {
    uint32_t len = 1, id = win_parent;
    while (ar.parent[id]) { len++; id = ar.parent[id]; }
    out->n_witness = len; out->final_state = win_state;
    if (witness) {
      /* the chain holds the branching calls only: replay it from the root, absorbing reads as the search did */
      uint32_t* chain = (uint32_t*)malloc(4 * (size_t)len);
      uint32_t w = len - 1; id = win_parent;
      chain[w] = win_op;
      while (ar.parent[id]) { chain[--w] = ar.op[id]; id = ar.parent[id]; }
      uint64_t* M2 = (uint64_t*)calloc(MW, 8);
      uint32_t fr = 0, nw = 0; int32_t s = model->init;
      for (uint32_t i = 0; i < len; i++) {
        const uint32_t op = chain[i];
        int32_t s2 = s; (void)oracle_step(model, s, f[op], a[op], b[op], &s2); s = s2;
        witness[nw++] = op;
        if (ret_rank[op] != 0xFFFFFFFFu) {
          const uint32_t p = slot[op];
          M2[p >> 6] |= 1ull << (p & 63);
          if (ret_rank[op] == fr) {
            uint32_t pp = p;
            for (;;) {
              M2[pp >> 6] &= ~(1ull << (pp & 63));
              fr++;
              if (fr == R) break;
              pp = slot[ret_op[fr]];
              if (!(M2[pp >> 6] >> (pp & 63) & 1)) break;
            }
          }
        }
        fr = cabsorb(M2, fr, s, R, off, lst, slot, ret_op, f, a, NULL, witness, &nw);
      }
      out->n_witness = nw;
      free(chain); free(M2);
    }
  } else if (verdict == 0) {
    out->fail_op = ret_op[maxf];
    out->prev_ok_op = maxf ? ret_op[maxf - 1] : 0xFFFFFFFFu;
  }
  free(rets); free(ret_rank); free(inv_rank); free(ret_op); free(slot); free(cls); free(cls_of); free(mem_rank); free(mem_op);
  free(ncr); free(off); free(lst);
  free(ar.keys); free(ar.parent); free(ar.op); free(ar.next_same); free(ar.slots); free(stack); free(dstack); free(key); free(ck);
  return 0;
}
