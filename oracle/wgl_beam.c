/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE (see oracle_model.h for the rules
 * and the PARITY UNPINNED statement).
 *
 * wgl_beam.c -- scalar CPU restatement of the WIDE schedule of the Wing-Gong/
 * Lowe search that jepsen-tigerbeetle_amd/csrc/wgl_beam.hip runs: the same
 * search space and the same memoisation as wgl_ref.c / wgl_window.c
 * (configs = (front, mask, state), exact visited set), but instead of one
 * config per step it takes the K most recent configs off an explicit stack,
 * expands ALL their successors at once and pushes the new ones (so the most
 * recent config's first successor is on top: depth-first in spirit, K wide).
 * The verdict and the failing op are properties of (model, history) and are
 * the same as the sequential search's; the witness and the counters are those
 * of THIS schedule, which is fully deterministic and specified here:
 *
 *   open(F)  = live ops with inv_rank <= F <= ret_rank in order of process slot,
 *              then crashed ops with inv_rank <= F in invocation order -- except crashed
 *              reads (value nil: they never completed), which are never candidates: such
 *              a call has no effect on the model and constrains nothing, so linearizing it
 *              or not changes neither the verdict nor the failing op, but every one of
 *              them would double the number of configs (the sequential restatements keep
 *              them, as Knossos does).
 *   iteration: pop np = min(K, |stack|) configs P_0 (top) .. P_{np-1};
 *     pairs are enumerated parent-bottom-first (P_{np-1} .. P_0), and within a
 *     parent from its LAST open op to its first; pair number r = 0, 1, ...;
 *     pairs are processed in rounds of 64 consecutive r:
 *       - a pair is viable if its op is not yet linearized in the parent and
 *         the model accepts it; its child config is computed as in WGL
 *         (linearize; if it was the front's own op the front moves past every
 *         completion already linearized);
 *       - if any child of the round has passed every completion the search
 *         ends VALID with the lowest such r (nothing of that round is
 *         inserted);
 *       - otherwise, in ascending r, each viable pair probes the visited set;
 *         a child that is new is inserted (remembering parent and op) and
 *         pushed -- unless the lookahead (below, register / cas-register only,
 *         tbc_opts.lookahead) finds it dead: then it is inserted and SET ASIDE on a second stack.
 *   empty stack, configs set aside  =>  they become the stack, lookahead is switched off for the rest
 *         of the search (so an INVALID verdict has expanded every reachable config exactly once: same
 *         failing op, configs and visited / probes / expanded as without lookahead);
 *   empty stack  =>  INVALID; the failing op is the completion of the greatest
 *   front ever inserted (the first completion whose prefix cannot be
 *   linearized), as in wgl_ref.c.
 *
 * Counters: probes = viable pairs of completed rounds, visited = configs
 * inserted (root included), expanded = configs popped, iterations,
 * max_stack.
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

typedef struct beam_stats {
  uint64_t iterations, probes, visited, expanded, max_stack, rounds;
} beam_stats;

/* arena of configs: kw words key + parent + op; index 0 unused */
typedef struct {
  uint64_t* keys; uint32_t* parent; uint32_t* op; size_t n, cap, kw;
  uint32_t* slots; size_t nslots;
} arena;

static uint64_t key_hash(const uint64_t* k, size_t kw) {
  uint64_t h = mix64(k[0]);
  for (size_t i = 1; i < kw; i++) h = mix64(h ^ k[i]) + 0x9E3779B97F4A7C15ull;
  return h;
}
static void arena_rehash(arena* a) {
  size_t ns = a->nslots * 2;
  uint32_t* s = (uint32_t*)calloc(ns, 4);
  for (size_t i = 1; i < a->n; i++) {
    size_t j = key_hash(a->keys + i * a->kw, a->kw) & (ns - 1);
    while (s[j]) j = (j + 1) & (ns - 1);
    s[j] = (uint32_t)i;
  }
  free(a->slots); a->slots = s; a->nslots = ns;
}
/* returns new index if inserted, 0 if present */
static uint32_t arena_add(arena* a, const uint64_t* k, uint32_t parent, uint32_t op) {
  size_t j = key_hash(k, a->kw) & (a->nslots - 1);
  while (a->slots[j]) {
    if (memcmp(a->keys + (size_t)a->slots[j] * a->kw, k, a->kw * 8) == 0) return 0;
    j = (j + 1) & (a->nslots - 1);
  }
  if (a->n == a->cap) {
    a->cap *= 2;
    a->keys = (uint64_t*)realloc(a->keys, a->cap * a->kw * 8);
    a->parent = (uint32_t*)realloc(a->parent, a->cap * 4);
    a->op = (uint32_t*)realloc(a->op, a->cap * 4);
  }
  uint32_t id = (uint32_t)a->n++;
  memcpy(a->keys + (size_t)id * a->kw, k, a->kw * 8);
  a->parent[id] = parent; a->op[id] = op;
  a->slots[j] = id;
  if (a->n * 2 > a->nslots) arena_rehash(a);
  return id;
}

/* what a run writes is thread-local (many.c checks histories on a thread pool); the switches below are process-wide
 * and set before the pool starts */
static _Thread_local uint64_t* g_bcfg = NULL; static _Thread_local uint32_t g_bcfg_n = 0, g_bcfg_kw = 0;
static _Thread_local size_t g_bsort_kw;
static int cmp_bcfg(const void* x, const void* y) {
  const uint64_t* a = (const uint64_t*)x; const uint64_t* b = (const uint64_t*)y;
  int32_t sa = (int32_t)(a[0] >> 32), sb = (int32_t)(b[0] >> 32);
  if (sa != sb) return sa < sb ? -1 : 1;
  for (size_t i = 1; i < g_bsort_kw; i++) if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
  return 0;
}
/* as wgl_window_last_configs, for the last invalid wide-schedule run */
uint32_t wgl_beam_last_configs(uint64_t* out, uint32_t max, uint32_t* kw) {
  *kw = g_bcfg_kw;
  for (uint32_t i = 0; i < g_bcfg_n && i < max; i++) memcpy(out + (size_t)i * g_bcfg_kw, g_bcfg + (size_t)i * g_bcfg_kw, g_bcfg_kw * 8);
  return g_bcfg_n;
}

/* eager reads: mark every open, not yet linearized read whose value is nil or the state `s` as linearized
 * (in list order), move the front past every completion now linearized, and repeat while that opens new
 * calls.  Returns the new front; appends the absorbed ops to wit[*nw...] when wit is given. */
static uint32_t absorb_reads(uint64_t* c2, uint32_t fi2, int32_t s, uint32_t R, const uint32_t* off, const uint32_t* lst,
                             const int32_t* process, const uint32_t* ret_op, const uint8_t* f, const int32_t* a,
                             uint32_t* wit, uint32_t* nw);

/* round_pairs: pairs per round (64 = one wavefront; 256 = the workgroup-cooperative kernel) */
int wgl_beam_check_rp(uint32_t n, const uint8_t* f, const int32_t* a, const int32_t* b,
                      const int32_t* process, uint32_t n_process,
                      const uint32_t* inv_pos, const uint32_t* ret_pos,
                      const oracle_model* model, uint32_t K, uint32_t round_pairs, uint64_t max_probes,
                      uint32_t* witness, oracle_result* out, beam_stats* st);

/* ---- options that are part of the specified schedule (the kernel has the same ones) ------------------
 * widen_after: 0 = never; otherwise a history that has used more than this many rounds continues
 *   with K = 16 (tbc_opts.round_budget).
 * lookahead:   tbc_opts.lookahead; the rule is stated where it is applied.  0 = off. */
static uint32_t g_widen_after = 0;
void wgl_beam_set_widen_after(uint32_t r) { g_widen_after = r; }
static uint32_t g_lookahead = 0, g_lookahead_depth = 8; static _Thread_local uint64_t g_pruned = 0;
void wgl_beam_set_lookahead(uint32_t on) { g_lookahead = on; }
uint64_t wgl_beam_pruned(void) { return g_pruned; }
/* searches that found their linearization only AFTER the set-aside configs were taken up: must stay 0,
 * or the lookahead rule called a live config dead (the verdict would still be right) */
static _Thread_local uint64_t g_late_valid = 0;
uint64_t wgl_beam_late_valid(void) { return g_late_valid; }
/* Self-check of a cheaper twin test for CRASHED candidates (not what the kernel does yet):
 * crashed calls of one effect are linearized in invocation order under the rule, so among the crashed ones only the
 * PREVIOUS call of the same effect has to be looked at -- the walk over the front's list shrinks to its live entries.
 * When switched on, every twin test of a crashed candidate is evaluated both ways; the counters say how often and
 * whether the two ever disagreed. */
static uint32_t g_twin_selfcheck = 0;
static _Thread_local uint64_t g_twin_checked = 0, g_twin_mismatch = 0;
void wgl_beam_set_twin_selfcheck(uint32_t on) { g_twin_selfcheck = on; g_twin_checked = 0; g_twin_mismatch = 0; }
uint64_t wgl_beam_twin_checked(void) { return g_twin_checked; }
uint64_t wgl_beam_twin_mismatch(void) { return g_twin_mismatch; }

/* ---- experiment knobs (all off by default; NOT part of the specified schedule, no kernel counterpart).
 * They produced the negative results recorded in DESIGN.md section 6 (the tail of a batch):
 * list_order:  order of each front's live open calls (the FIRST is tried first): 0 = by process slot
 *              (the specified one), 1 = by completion, 2 = by invocation, 3 = by completion, latest first.
 * stall:       when the greatest front has not moved for `rounds` rounds: mode 0 reverse / mode 1 rotate the
 *              top `width` stack entries (doubling), mode 2 take `width` configs per iteration while stalled.
 * lookahead_depth: completions looked at (the kernel's is 8).
 * trace:       greatest front and stack depth sampled every 256 rounds. */
void wgl_beam_set_lookahead_depth(uint32_t d) { g_lookahead_depth = d; }
/* look_two (the lean lookahead record of the narrow kernel, csrc/tbc_internal.h kLeanLook): the record of a rank names at most TWO of the
 * other calls open at that rank's front that produce the needed value; three or more are read as "one of them is still to be
 * linearized" -- the rule then lets the config live without looking further.  Never a config declared dead that is not. */
static uint32_t g_look_two = 0;
void wgl_beam_set_look_two(uint32_t on) { g_look_two = on; }
/* lazy_look (DESIGN STUDY, no kernel counterpart yet; one config per iteration only): the lookahead is run at once only for the new
 * config of a round that will be popped next (the last one pushed); its siblings are pushed UNCHECKED and looked at if they are ever
 * popped -- a dead one is set aside then.  Same verdicts, and in a nearly greedy search about half the lookahead runs.
 * wgl_beam_look_runs(): lookahead evaluations of the last run. */
static uint32_t g_lazy_look = 0;
static _Thread_local uint64_t g_look_runs = 0;
void wgl_beam_set_lazy_look(uint32_t on) { g_lazy_look = on; }
/* defer (DESIGN STUDY, no kernel counterpart; one config per iteration only): ONE CHILD AT A TIME.  Of a round's viable pairs only the
 * last one -- the child that is popped next -- is probed, inserted and pushed; the others go onto the stack as MARKERS (parent, pair)
 * and become configs only if the search ever pops them (a one-pair round then).  The order of exploration is the eager schedule's;
 * a sibling nobody comes back to costs no probe, no entry, no front record.  Success is still seen at once (a child is computed
 * without memory). */
static uint32_t g_defer = 0;
void wgl_beam_set_defer(uint32_t on) { g_defer = on; }
uint64_t wgl_beam_look_runs(void) { return g_look_runs; }
static uint32_t g_list_order = 0;
void wgl_beam_set_list_order(uint32_t o) { g_list_order = o; }
/* eager reads (experiment for the next round, register family): a read that is viable NOW can be linearized
 * now without loss of generality (it does not change the state, so any later schedule stays possible): every
 * child absorbs all of them, again after each step of the front.  Witness reconstruction is not done here. */
static uint32_t g_eager_reads = 0; static _Thread_local uint64_t g_absorbed = 0;
/* twin writes (experiment, register family): of several open, not yet linearized calls with the same effect
 * (:write v, or :cas [a b] with equal a and b) the one completing first goes first, without loss of generality */
static uint32_t g_twin_rule = 0;
/* branch lists without reads (the narrow kernel's lists under the eager rule, wgl_narrow.hip): every config of the search is
 * in normal form -- every open read its state allows is linearized -- so a read is never a viable candidate; the per-front
 * candidate lists then hold the live :write / :cas calls only (reads are still found by the eager rule itself), a round's
 * pairs are numbered over those, and the ROOT is put into normal form before it is inserted (it is the one config the
 * plain schedule leaves un-normalised).  Needs eager reads; changes pair numbers, hence rounds and counters, not answers. */
static uint32_t g_branch_lists = 0;
void wgl_beam_set_branch_lists(uint32_t on) { g_branch_lists = on; }
void wgl_beam_set_twin_rule(uint32_t on) { g_twin_rule = on; }
void wgl_beam_set_eager_reads(uint32_t on) { g_eager_reads = on; }
/* eager_txns (multi-register; tbc_opts.dominance, TBC_DOM_NO_EAGER_TXNS = off): the eager rule for knossos.model/multi-register.
 * An open :txn made of micro-READS only, each of nil or of its key's current value, changes nothing and can be linearized now
 * without loss of generality -- the argument of the eager reads word for word: everything that must
 * precede it is linearized, the state stays, every later schedule stays possible.  A txn that writes is never absorbed, not
 * even one that writes the values already there (it may be needed later, to put them back). */
static uint32_t g_eager_txns = 0;
void wgl_beam_set_eager_txns(uint32_t on) { g_eager_txns = on; }
uint64_t wgl_beam_absorbed(void) { return g_absorbed; }
/* LAZY RULE for the commutative models (set, bank; tbc_opts.dominance, TBC_DOM_NO_LAZY_COMMUTING off = rule on).  An :add / :transfer
 * changes nothing any call can see except a :read, and the calls of these models commute, so in any linearization a mutating call
 * can be moved later -- up to its own completion, or (a crashed one: for ever) -- past everything but a read.  Without loss of
 * generality it is therefore linearized only when it COMPLETES AT THE FRONT, or when an open, not yet linearized read could take it:
 * set -- a read whose value contains the element (a crashed add no read ever contains is never linearized at all: knossos.model/set
 * reads are exact); bank -- any read with a value.  Config space: no longer 2^(open or crashed mutating calls). */
static uint32_t g_lazy_comm = 0;
void wgl_beam_set_lazy_commuting(uint32_t on) { g_lazy_comm = on; }
static uint32_t g_stall_rounds = 0, g_stall_width = 64, g_stall_mode = 0;
void wgl_beam_set_stall(uint32_t rounds, uint32_t width, uint32_t mode) { g_stall_rounds = rounds; g_stall_width = width; g_stall_mode = mode; }
static uint32_t* g_trace = NULL; static uint32_t g_trace_cap = 0, g_trace_n = 0;
void wgl_beam_set_trace(uint32_t* buf, uint32_t cap) { g_trace = buf; g_trace_cap = cap; g_trace_n = 0; }
uint32_t wgl_beam_trace_len(void) { return g_trace_n; }

int wgl_beam_check(uint32_t n, const uint8_t* f, const int32_t* a, const int32_t* b,
                   const int32_t* process, uint32_t n_process,
                   const uint32_t* inv_pos, const uint32_t* ret_pos,
                   const oracle_model* model, uint32_t K, uint64_t max_probes,
                   uint32_t* witness, oracle_result* out, beam_stats* st) {
  return wgl_beam_check_rp(n, f, a, b, process, n_process, inv_pos, ret_pos, model, K, 64, max_probes, witness, out, st);
}

static uint32_t absorb_reads(uint64_t* c2, uint32_t fi2, int32_t s, uint32_t R, const uint32_t* off, const uint32_t* lst,
                             const int32_t* process, const uint32_t* ret_op, const uint8_t* f, const int32_t* a,
                             uint32_t* wit, uint32_t* nw) {
  int again = 1;
  while (again && fi2 < R) {
    again = 0;
    const uint32_t nl = off[fi2 + 1] - off[fi2];
    for (uint32_t cc = 0; cc < nl; cc++) {
      const uint32_t x = lst[off[fi2] + cc], px = (uint32_t)process[x];
      if (c2[1 + (px >> 6)] >> (px & 63) & 1) continue;
      if (f[x] != O_READ || !(a[x] == O_NIL || a[x] == s)) continue;
      c2[1 + (px >> 6)] |= 1ull << (px & 63); g_absorbed++;
      if (wit) wit[(*nw)++] = x;
    }
    /* the front moves past every completion now linearized; new calls open up: look again */
    uint32_t pp = (uint32_t)process[ret_op[fi2]];
    while (c2[1 + (pp >> 6)] >> (pp & 63) & 1) {
      c2[1 + (pp >> 6)] &= ~(1ull << (pp & 63));
      fi2++; again = 1;
      if (fi2 == R) break;
      pp = (uint32_t)process[ret_op[fi2]];
    }
  }
  return fi2;
}

/* txn_independence (multi-register; tbc_opts.dominance, TBC_DOM_NO_TXN_INDEPENDENCE = off): the persistent-set rule.  Two txns CONFLICT
 * when one writes a key the other reads or writes; txns that do not conflict commute (same states, same viability, either order).  At a
 * config whose front is the completion of call X, every linearization of what is left takes X before the front can move, and everything
 * it takes before X is open at the front (a call invoked later is preceded by X in real time), hence concurrent with X and with each other.
 * Let P = the closure of {X} under "conflicts with" among the open calls not yet linearized (viable now or not).  In any valid
 * continuation the first member of P can be moved to the very front: everything before it is outside P, so it conflicts with no member of
 * P, and all of them overlap in real time.  So the config needs only the candidates in P.  With keys as bit sets the closure is a fixed
 * point over two words: CR / CW = the keys the members read / write; Y joins when Y.w & (CR | CW) or Y.r & CW. */
static uint32_t g_txn_por = 0; static _Thread_local uint64_t g_por_skipped = 0;
void wgl_beam_set_txn_independence(uint32_t on) { g_txn_por = on; }
static void txn_keys(const oracle_model* model, uint8_t f, int32_t a, int32_t b, uint32_t* r, uint32_t* w) {
  *r = *w = 0;
  if (f != O_TXN) return;
  for (int32_t i = 0; i < b; i++) {
    const int32_t mf = model->pool[a + 3 * i], k = model->pool[a + 3 * i + 1];
    if (mf == 0) *r |= 1u << k; else *w |= 1u << k;
  }
}

/* multi-register: is op x a :txn of micro-reads only, each consistent with state s? */
static int pure_read_txn_ok(const oracle_model* model, int32_t s, uint8_t f, int32_t a, int32_t b) {
  if (f != O_TXN) return 0;
  for (int32_t i = 0; i < b; i++) {
    const int32_t mf = model->pool[a + 3 * i], k = model->pool[a + 3 * i + 1], v = model->pool[a + 3 * i + 2];
    if (mf != 0) return 0;
    if (!(v == O_NIL || (((uint32_t)s >> (4 * k)) & 15u) == (uint32_t)(v + 1))) return 0;
  }
  return 1;
}

/* eager_txns: absorb_reads for multi-register -- the config takes every open pure-read txn its state allows, the front moves
 * past the completions that linearizes, the calls open at the new front are looked at again */
static uint32_t absorb_txns(const oracle_model* model, uint64_t* c2, uint32_t fi2, int32_t s, uint32_t R, const uint32_t* off, const uint32_t* lst,
                            const int32_t* process, const uint32_t* ret_op, const uint8_t* f, const int32_t* a, const int32_t* b,
                            uint32_t* wit, uint32_t* nw) {
  int again = 1;
  while (again && fi2 < R) {
    again = 0;
    const uint32_t nl = off[fi2 + 1] - off[fi2];
    for (uint32_t cc = 0; cc < nl; cc++) {
      const uint32_t x = lst[off[fi2] + cc], px = (uint32_t)process[x];
      if (c2[1 + (px >> 6)] >> (px & 63) & 1) continue;
      if (!pure_read_txn_ok(model, s, f[x], a[x], b[x])) continue;
      c2[1 + (px >> 6)] |= 1ull << (px & 63); g_absorbed++;
      if (wit) wit[(*nw)++] = x;
    }
    uint32_t pp = (uint32_t)process[ret_op[fi2]];
    while (c2[1 + (pp >> 6)] >> (pp & 63) & 1) {
      c2[1 + (pp >> 6)] &= ~(1ull << (pp & 63));
      fi2++; again = 1;
      if (fi2 == R) break;
      pp = (uint32_t)process[ret_op[fi2]];
    }
  }
  return fi2;
}

int wgl_beam_check_rp(uint32_t n, const uint8_t* f, const int32_t* a, const int32_t* b,
                      const int32_t* process, uint32_t n_process,
                      const uint32_t* inv_pos, const uint32_t* ret_pos,
                      const oracle_model* model, uint32_t K, uint32_t round_pairs, uint64_t max_probes,
                      uint32_t* witness, oracle_result* out, beam_stats* st) {
  memset(out, 0, sizeof *out); memset(st, 0, sizeof *st);
  out->fail_op = out->prev_ok_op = 0xFFFFFFFFu;
  if (K == 0 || K > 64 || round_pairs == 0 || round_pairs > 1024) return 1;
  const uint32_t RP = round_pairs;
  uint32_t R = 0;
  for (uint32_t i = 0; i < n; i++) {
    if (i && inv_pos[i] <= inv_pos[i - 1]) return 2;
    if (process[i] < 0 || (uint32_t)process[i] >= n_process) return 2;
    if (ret_pos[i] != O_CRASHED) { if (ret_pos[i] <= inv_pos[i]) return 2; R++; }
  }
  const int cfgm = oracle_is_cfg_model(model);
  if (R == 0) { out->valid = 1; out->final_state = cfgm ? 0 : model->init; return 0; }
  const uint32_t W = n_process, MW = (W + 63) / 64, KW = 1 + MW;
  uint32_t* open_ops = (uint32_t*)malloc(4 * ((size_t)n + 1));
  uint8_t* open_lin = (uint8_t*)malloc((size_t)n + 1);

  posop* rets = (posop*)malloc(sizeof(posop) * R);
  uint32_t* ret_rank = (uint32_t*)malloc(4 * n);
  uint32_t* inv_rank = (uint32_t*)malloc(4 * n);
  uint32_t* ret_op = (uint32_t*)malloc(4 * R);
  uint32_t k = 0;
  for (uint32_t i = 0; i < n; i++) if (ret_pos[i] != O_CRASHED) { rets[k].pos = ret_pos[i]; rets[k].op = i; k++; }
  qsort(rets, R, sizeof(posop), cmp_posop);
  for (uint32_t r = 0; r < R; r++) { ret_rank[rets[r].op] = r; ret_op[r] = rets[r].op; }
  { uint32_t r = 0;
    for (uint32_t i = 0; i < n; i++) { while (r < R && rets[r].pos < inv_pos[i]) r++; inv_rank[i] = r; } }
  for (uint32_t i = 0; i < n; i++) if (ret_pos[i] == O_CRASHED) ret_rank[i] = 0xFFFFFFFFu;

  /* open lists: CSR of live ops per front; crashed ops as one list + count per front */
  uint32_t* off = (uint32_t*)calloc((size_t)R + 1, 4);
  uint32_t* ncr = (uint32_t*)calloc((size_t)R + 1, 4);   /* crashed ops with inv_rank <= F */
  uint32_t n_crashed = 0;
  for (uint32_t i = 0; i < n; i++) {
    if (ret_rank[i] == 0xFFFFFFFFu) {
      if (f[i] == O_READ && a[i] == O_NIL) continue;             /* crashed no-op read: never a candidate */
      n_crashed++; if (inv_rank[i] < R) ncr[inv_rank[i]]++; continue;
    }
    for (uint32_t fr = inv_rank[i]; fr <= ret_rank[i]; fr++) off[fr + 1]++;
  }
  for (uint32_t r = 0; r < R; r++) off[r + 1] += off[r];
  for (uint32_t r = 1; r < R; r++) ncr[r] += ncr[r - 1];
  uint32_t* lst = (uint32_t*)malloc(4 * ((size_t)off[R] + 1));
  uint32_t* fill = (uint32_t*)malloc(4 * ((size_t)R + 1));
  memcpy(fill, off, 4 * ((size_t)R + 1));
  uint32_t* crashed = (uint32_t*)malloc(4 * ((size_t)n_crashed + 1));
  { uint32_t c = 0;
    for (uint32_t i = 0; i < n; i++) {
      if (ret_rank[i] == 0xFFFFFFFFu) { if (!(f[i] == O_READ && a[i] == O_NIL)) crashed[c++] = i; continue; }
      for (uint32_t fr = inv_rank[i]; fr <= ret_rank[i]; fr++) lst[fill[fr]++] = i;
    }
#define LIST_KEY(o) (g_list_order == 0 ? (uint32_t)process[o] : g_list_order == 1 ? ret_rank[o] : \
                     g_list_order == 2 ? inv_rank[o] * 1024u + (uint32_t)process[o] : g_list_order == 3 ? 0xFFFFFFFEu - ret_rank[o] : \
                     g_list_order == 4 ? (ret_rank[o] | (f[o] == O_WRITE ? 0x40000000u : 0u)) : \
                     g_list_order == 5 ? (ret_rank[o] | (f[o] == O_CAS ? 0x40000000u : 0u)) : \
                     g_list_order == 6 ? (f[o] == O_WRITE ? 0x7FFFFFFFu - ret_rank[o] : ret_rank[o]) : \
                     g_list_order == 7 ? (f[o] == O_WRITE ? 0x40000000u + inv_rank[o] : ret_rank[o]) : \
                     (ret_rank[o] * 2u + (f[o] == O_WRITE ? 2u * (g_list_order - 16u) + 1u : 0u)))      /* >= 16: a :write as if it completed (order - 16) ranks later */
    for (uint32_t fr = 0; fr < R; fr++)          /* each front's live list in the chosen order */
      for (uint32_t x = off[fr] + 1; x < off[fr + 1]; x++) {
        uint32_t v = lst[x], y = x;
        while (y > off[fr] && LIST_KEY(lst[y - 1]) > LIST_KEY(v)) { lst[y] = lst[y - 1]; y--; }
        lst[y] = v;
      } }

  /* candidate lists: the full lists, or (branch lists) the live calls that are not reads */
  const int regfam = !cfgm && (model->kind == O_REGISTER || model->kind == O_CAS_REGISTER);
  const int branch = g_branch_lists && g_eager_reads && regfam;
  uint32_t* coff = off; uint32_t* clst = lst;
  if (branch) {
    coff = (uint32_t*)calloc((size_t)R + 1, 4);
    clst = (uint32_t*)malloc(4 * ((size_t)off[R] + 1));
    uint32_t run = 0;
    for (uint32_t fr = 0; fr < R; fr++) {
      coff[fr] = run;
      for (uint32_t x = off[fr]; x < off[fr + 1]; x++) if (f[lst[x]] != O_READ) clst[run++] = lst[x];
    }
    coff[R] = run;
  }

  /* previous crashed call of the same effect (index into crashed[], 0xFFFFFFFF = none): the self-check above */
  uint32_t* prev_twin = NULL;
  if (g_twin_selfcheck) {
    prev_twin = (uint32_t*)malloc(4 * ((size_t)n_crashed + 1));
    for (uint32_t k = 0; k < n_crashed; k++) {
      const uint32_t x = crashed[k];
      prev_twin[k] = 0xFFFFFFFFu;
      for (uint32_t j = k; j-- > 0;) {
        const uint32_t y = crashed[j];
        if (f[y] == f[x] && a[y] == a[x] && (f[x] != O_CAS || b[y] == b[x])) { prev_twin[k] = j; break; }
      }
    }
  }

  arena ar; ar.kw = KW; ar.cap = 4096; ar.n = 1; ar.nslots = 8192;
  ar.keys = (uint64_t*)calloc(ar.cap * KW, 8); ar.parent = (uint32_t*)calloc(ar.cap, 4); ar.op = (uint32_t*)calloc(ar.cap, 4);
  ar.slots = (uint32_t*)calloc(ar.nslots, 4);
  size_t scap = 1 << 16, sp = 0;
  uint32_t* stack = (uint32_t*)malloc(scap * 4);
  uint16_t* spair = (uint16_t*)malloc(scap * 2);          /* defer: 0xFFFF = a config; else the stack word is a PARENT and this its pair number */
  size_t spcap = scap;
#define SPAIR_ROOM() do { if (spcap < scap) { spair = (uint16_t*)realloc(spair, scap * 2); spcap = scap; } } while (0)
  /* configs the lookahead found dead: set aside, expanded only if the search would otherwise end INVALID */
  size_t dcap = 1 << 12, dsp = 0;
  uint32_t* dstack = (uint32_t*)malloc(dcap * 4);
  int look_on = g_lookahead && !cfgm && (model->kind == O_REGISTER || model->kind == O_CAS_REGISTER);
  uint64_t* key = (uint64_t*)calloc(KW, 8);
  key[0] = 1ull | ((uint64_t)(uint32_t)(cfgm ? 0 : model->init) << 32);
  uint32_t maxf = 0;
  int verdict = -2;
  uint32_t win_parent = 0, win_op = 0; int32_t win_state = 0;
  int root_wins = 0;
  if (branch) {          /* the root in normal form (it may pass every completion: then there is nothing to search) */
    const uint32_t f0 = absorb_reads(key, 0, model->init, R, off, lst, process, ret_op, f, a, NULL, NULL);
    key[0] = (uint64_t)(f0 + 1) | ((uint64_t)(uint32_t)model->init << 32);
    maxf = f0 < R ? f0 : R - 1;
    if (f0 == R) { root_wins = 1; verdict = 1; win_state = model->init; }
  }
  spair[sp] = 0xFFFFu;
  stack[sp++] = arena_add(&ar, key, 0, 0xFFFFFFFFu);
  st->visited = 1; st->max_stack = 1;

  uint8_t* unchecked = NULL; size_t unchecked_cap = 0;          /* lazy_look: configs pushed without their lookahead run, by arena id */
  g_look_runs = 0;
  /* the lookahead's verdict on a config (key words c2, front F, state s2): 1 = dead (GCC nested function: it reads the search's tables) */
  int look_dead(const uint64_t* c2, uint32_t F, int32_t s2) {
    int dead = 0;
    g_look_runs++;
    for (uint32_t j = 0; j < g_lookahead_depth && F + j < R && !dead; j++) {
      const uint32_t t = F + j, fop = ret_op[t], pf = (uint32_t)process[fop];
      if (!((f[fop] == O_READ && a[fop] != O_NIL) || f[fop] == O_CAS)) continue;
      const int32_t v = a[fop];
      if (v < 0 || v >= 32) continue;
      if (inv_rank[fop] <= F && (c2[1 + (pf >> 6)] >> (pf & 63) & 1)) continue;   /* already linearized */
      if (v == s2) continue;
      int ok = 0;
      if (g_look_two) {              /* three or more producers open at front t: the lean record says no more than that */
        const uint32_t nl = coff[t + 1] - coff[t], tot = nl + ncr[t];
        uint32_t np = 0;
        for (uint32_t cc = 0; cc < tot; cc++) {
          const uint32_t x = cc < nl ? clst[coff[t] + cc] : crashed[cc - nl];
          if (x != fop && ((f[x] == O_WRITE && a[x] == v) || (f[x] == O_CAS && b[x] == v))) np++;
        }
        if (np >= 3) ok = 1;
      }
      for (uint32_t F2 = F; F2 <= t && !ok; F2++) {                /* calls open somewhere in [F, t] */
        const uint32_t nl = coff[F2 + 1] - coff[F2], tot = nl + (F2 == t ? ncr[F2] : 0);
        for (uint32_t cc = 0; cc < tot && !ok; cc++) {
          const uint32_t x = cc < nl ? clst[coff[F2] + cc] : crashed[cc - nl];
          const uint32_t px = (uint32_t)process[x];
          if (x == fop) continue;
          if (inv_rank[x] <= F && (c2[1 + (px >> 6)] >> (px & 63) & 1)) continue;   /* open at F, linearized */
          if ((f[x] == O_WRITE && a[x] == v) || (f[x] == O_CAS && b[x] == v)) ok = 1;
        }
      }
      if (!ok) dead = 1;
    }
    return dead;
  }
  uint32_t par[64], pcnt[64], pstart[65];
  /* per-round scratch */
  uint64_t* ck = (uint64_t*)malloc((size_t)RP * KW * 8);
  uint32_t cop[1024], cpar[1024]; int cviable[1024]; uint32_t cfront[1024]; int32_t cstate[1024]; uint16_t cpair[1024];

  uint64_t last_progress = 0; uint32_t seen_maxf = 0, stall_w = g_stall_width;
  while (verdict == -2) {
    if (sp == 0) {
      if (dsp == 0) { verdict = 0; break; }
      /* no linearization through the live configs: the set-aside ones become the stack (in the order they
       * were set aside) and the search goes on WITHOUT lookahead, so every reachable config is expanded
       * exactly once overall -- failing op, configs, visited / probes / expanded are the plain search's */
      { uint32_t* t = stack; stack = dstack; dstack = t; size_t c = scap; scap = dcap; dcap = c; }
      sp = dsp; dsp = 0; look_on = 0;
      SPAIR_ROOM();
      for (size_t i = 0; i < sp; i++) spair[i] = 0xFFFFu;
    }
    if (g_stall_rounds && g_stall_mode < 2) {
      if (maxf > seen_maxf) { seen_maxf = maxf; last_progress = st->rounds; stall_w = g_stall_width; }
      else if (st->rounds - last_progress > g_stall_rounds) {
        /* stuck below one completion: bring older pending configs to the top */
        size_t w = stall_w < sp ? stall_w : sp;
        if (g_stall_mode == 0) {            /* reverse the top w entries */
          for (size_t i = 0; i < w / 2; i++) { uint32_t t = stack[sp - 1 - i]; stack[sp - 1 - i] = stack[sp - w + i]; stack[sp - w + i] = t; }
        } else {                            /* rotate: the w-th entry from the top becomes the top */
          uint32_t t = stack[sp - w];
          memmove(stack + sp - w, stack + sp - w + 1, (w - 1) * 4);
          stack[sp - 1] = t;
        }
        last_progress = st->rounds;
        if (stall_w < (1u << 20)) stall_w *= 2;
      }
    }
    if (maxf > seen_maxf) { seen_maxf = maxf; last_progress = st->rounds; }
    /* stall widening (tbc_opts.stall_rounds): while the greatest front has not moved for that many rounds the
     * search is exhausting a dead region -- take stall_width configs per iteration until it moves again */
    const uint32_t Ks = (g_stall_rounds && g_stall_mode == 2 && st->rounds - last_progress > g_stall_rounds && K < g_stall_width) ? g_stall_width : K;
    const uint32_t Kc = (g_widen_after && st->rounds > g_widen_after && Ks < 16) ? 16 : Ks;
    uint32_t np = sp < Kc ? (uint32_t)sp : Kc;
    for (uint32_t q = 0; q < np; q++) par[q] = stack[sp - np + q];      /* q = 0 is the bottom-most popped */
    const int single = g_defer && K == 1 && np == 1 && spair[sp - 1] != 0xFFFFu;      /* a marker: its parent, one pair */
    const uint32_t single_c = single ? spair[sp - 1] : 0;
    sp -= np;
    if (g_lazy_look && K == 1 && np == 1 && look_on && par[0] < unchecked_cap && unchecked[par[0]]) {
      /* lazy_look: a sibling that was pushed unchecked is looked at now that it is wanted; dead -> set aside, the next one is popped */
      unchecked[par[0]] = 0;
      const uint64_t* pk0 = ar.keys + (size_t)par[0] * KW;
      if (look_dead(pk0, (uint32_t)pk0[0] - 1, (int32_t)(pk0[0] >> 32))) {
        g_pruned++;
        if (dsp == dcap) { dcap *= 2; dstack = (uint32_t*)realloc(dstack, dcap * 4); }
        dstack[dsp++] = par[0];
        continue;
      }
    }
    st->iterations++; if (!single) st->expanded += np;
    uint32_t T = 0;
    for (uint32_t q = 0; q < np; q++) {
      uint32_t fi = (uint32_t)ar.keys[(size_t)par[q] * KW] - 1;
      pcnt[q] = (coff[fi + 1] - coff[fi]) + ncr[fi];
      pstart[q] = T; T += pcnt[q];
    }
    pstart[np] = T;
    if (single) T = 1;                   /* (pcnt stays the parent's: the twin rule looks at all of its open calls) */
    for (uint32_t base = 0; base < T && verdict == -2; base += RP) {
      uint32_t m = T - base < RP ? T - base : RP;
      int success = -1;
      for (uint32_t l = 0; l < m; l++) {
        uint32_t r = base + l, q = 0;
        while (pstart[q + 1] <= r) q++;
        const uint64_t* pk = ar.keys + (size_t)par[q] * KW;
        uint32_t fi = (uint32_t)pk[0] - 1; int32_t s = (int32_t)(pk[0] >> 32);
        uint32_t nlive = coff[fi + 1] - coff[fi];
        uint32_t c = single ? single_c : pcnt[q] - 1 - (r - pstart[q]);
        cpair[l] = (uint16_t)c;
        uint32_t op = c < nlive ? clst[coff[fi] + c] : crashed[c - nlive];
        uint32_t p = (uint32_t)process[op];
        cviable[l] = 0; cop[l] = op; cpar[l] = par[q];
        if (pk[1 + (p >> 6)] >> (p & 63) & 1) continue;
        if (g_txn_por && model->kind == O_MULTI_REGISTER && op != ret_op[fi]) {
          uint32_t cr, cw, yr, yw;
          txn_keys(model, f[ret_op[fi]], a[ret_op[fi]], b[ret_op[fi]], &cr, &cw);
          for (int grew = 1; grew;) {
            grew = 0;
            for (uint32_t cc = 0; cc < pcnt[q]; cc++) {
              const uint32_t y = cc < nlive ? clst[coff[fi] + cc] : crashed[cc - nlive];
              const uint32_t py = (uint32_t)process[y];
              if (pk[1 + (py >> 6)] >> (py & 63) & 1) continue;
              txn_keys(model, f[y], a[y], b[y], &yr, &yw);
              if (!((yw & (cr | cw)) | (yr & cw))) continue;
              if ((yr & ~cr) | (yw & ~cw)) { cr |= yr; cw |= yw; grew = 1; }
            }
          }
          txn_keys(model, f[op], a[op], b[op], &yr, &yw);
          if (!((yw & (cr | cw)) | (yr & cw))) { g_por_skipped++; continue; }
        }
        if (g_twin_rule && !cfgm && (f[op] == O_WRITE || f[op] == O_CAS)) {
          int dominated = 0;
          for (uint32_t cc = 0; cc < pcnt[q] && !dominated; cc++) {
            uint32_t y = cc < nlive ? clst[coff[fi] + cc] : crashed[cc - nlive];
            uint32_t py = (uint32_t)process[y];
            if (y == op || f[y] != f[op] || a[y] != a[op] || (f[op] == O_CAS && b[y] != b[op])) continue;
            if (pk[1 + (py >> 6)] >> (py & 63) & 1) continue;
            if (ret_rank[y] < ret_rank[op] || (ret_rank[y] == ret_rank[op] && y < op)) dominated = 1;
          }
          if (g_twin_selfcheck && c >= nlive) {
            int fast = 0;
            for (uint32_t cc = 0; cc < nlive && !fast; cc++) {          /* live calls of the same effect: all complete earlier */
              uint32_t y = clst[coff[fi] + cc];
              uint32_t py = (uint32_t)process[y];
              if (f[y] != f[op] || a[y] != a[op] || (f[op] == O_CAS && b[y] != b[op])) continue;
              if (!(pk[1 + (py >> 6)] >> (py & 63) & 1)) fast = 1;
            }
            const uint32_t pt = prev_twin[c - nlive];
            if (!fast && pt != 0xFFFFFFFFu) {
              uint32_t py = (uint32_t)process[crashed[pt]];
              if (!(pk[1 + (py >> 6)] >> (py & 63) & 1)) fast = 1;
            }
            g_twin_checked++;
            if (fast != dominated) g_twin_mismatch++;
          }
          if (dominated) continue;
        }
        int32_t s2;
        if (cfgm) {
          uint32_t no = 0;
          for (uint32_t cc = 0; cc < pcnt[q]; cc++) {
            uint32_t x = cc < nlive ? clst[coff[fi] + cc] : crashed[cc - nlive];
            uint32_t px = (uint32_t)process[x];
            open_ops[no] = x; open_lin[no] = (uint8_t)(pk[1 + (px >> 6)] >> (px & 63) & 1); no++;
          }
          if (g_lazy_comm && (f[op] == O_ADD || f[op] == O_TRANSFER) && ret_rank[op] != fi) {
            int wanted = 0;
            for (uint32_t cc = 0; cc < no && !wanted; cc++) {
              const uint32_t x = open_ops[cc];
              if (open_lin[cc] || f[x] != O_READ || a[x] == O_NIL) continue;
              if (f[op] == O_TRANSFER) wanted = 1;
              else {
                const int32_t* rec = model->pool + a[x];
                const uint32_t j = (uint32_t)a[op];
                wanted = rec[0] >= 0 && ((((const uint32_t*)(rec + 2))[j >> 5] >> (j & 31)) & 1u);
              }
            }
            if (!wanted) continue;
          }
          if (!oracle_cfg_step(model, fi, open_ops, open_lin, no, f, a, op)) continue;
          s2 = 0;
        } else if (!oracle_step(model, s, f[op], a[op], b[op], &s2)) continue;
        uint64_t* c2 = ck + (size_t)l * KW;
        memcpy(c2, pk, KW * 8);
        c2[1 + (p >> 6)] |= 1ull << (p & 63);
        uint32_t fi2 = fi;
        if (ret_rank[op] == fi) {
          uint32_t pp = p;
          for (;;) {
            c2[1 + (pp >> 6)] &= ~(1ull << (pp & 63));
            fi2++;
            if (fi2 == R) break;
            pp = (uint32_t)process[ret_op[fi2]];
            if (!(c2[1 + (pp >> 6)] >> (pp & 63) & 1)) break;
          }
        }
        if (g_eager_reads && !cfgm && (model->kind == O_REGISTER || model->kind == O_CAS_REGISTER))
          fi2 = absorb_reads(c2, fi2, s2, R, off, lst, process, ret_op, f, a, NULL, NULL);
        if (g_eager_txns && model->kind == O_MULTI_REGISTER)
          fi2 = absorb_txns(model, c2, fi2, s2, R, off, lst, process, ret_op, f, a, b, NULL, NULL);
        c2[0] = (uint64_t)(fi2 + 1) | ((uint64_t)(uint32_t)s2 << 32);
        cviable[l] = 1; cfront[l] = fi2; cstate[l] = s2;
        if (fi2 == R && success < 0) success = (int)l;
      }
      st->rounds++;
      if (g_trace && (st->rounds & 255) == 0 && g_trace_n + 2 <= g_trace_cap) { g_trace[g_trace_n++] = maxf; g_trace[g_trace_n++] = (uint32_t)sp; }
      if (success >= 0 && g_lookahead && !look_on && !cfgm && (model->kind == O_REGISTER || model->kind == O_CAS_REGISTER)) g_late_valid++;
      if (success >= 0) { verdict = 1; win_parent = cpar[success]; win_op = cop[success]; win_state = cstate[success]; break; }
      for (uint32_t l = 0; l < m; l++) {
        if (!cviable[l]) continue;
        if (g_defer && K == 1 && !single) {          /* all but the last viable pair of the round: a marker, nothing else */
          int later = 0;
          for (uint32_t l2 = l + 1; l2 < m; l2++) later |= cviable[l2];
          if (later) {
            if (sp == scap) { scap *= 2; stack = (uint32_t*)realloc(stack, scap * 4); }
            SPAIR_ROOM();
            spair[sp] = cpair[l]; stack[sp++] = cpar[l];
            continue;
          }
        }
        st->probes++;
        uint32_t id = arena_add(&ar, ck + (size_t)l * KW, cpar[l], cop[l]);
        if (!id) continue;
        st->visited++;
        if (cfront[l] > maxf) maxf = cfront[l];
        /* branch lists: a config whose front has no candidate at all (only reads its state does not allow are open) has
         * no successor -- it counts as expanded on the spot and is neither pushed nor set aside */
        if (branch && (coff[cfront[l] + 1] - coff[cfront[l]]) + ncr[cfront[l]] == 0) { st->expanded++; continue; }
        if (look_on) {
          /* lookahead: the new config is dead if the call completing at one of the next 8 ranks can never
           * be linearized from it -- it is not linearized yet and needs a register value (0..31) that is
           * neither the state nor produced (:write v / :cas [_ v]) by any OTHER call that can still be
           * linearized before that completion: a call invoked after this front and before the completion,
           * or a call open at this front (crashed ones included) and not linearized.  No linearization goes
           * through a dead config, so it is set aside instead of pushed (see the top of the loop). */
          const int last_new = !g_lazy_look || K != 1 || l + 1 == m || ({ int later = 0; for (uint32_t l2 = l + 1; l2 < m; l2++) later |= cviable[l2]; !later; });
          if (!last_new) { if (id >= unchecked_cap) { const size_t nc = ((size_t)id + 1) * 2; unchecked = (uint8_t*)realloc(unchecked, nc); memset(unchecked + unchecked_cap, 0, nc - unchecked_cap); unchecked_cap = nc; }
                           unchecked[id] = 1; if (sp == scap) { scap *= 2; stack = (uint32_t*)realloc(stack, scap * 4); } SPAIR_ROOM(); spair[sp] = 0xFFFFu; stack[sp++] = id; continue; }
          const int dead = look_dead(ck + (size_t)l * KW, cfront[l], cstate[l]);
          if (dead) {
            g_pruned++;
            if (dsp == dcap) { dcap *= 2; dstack = (uint32_t*)realloc(dstack, dcap * 4); }
            dstack[dsp++] = id;
            continue;
          }
        }
        if (sp == scap) { scap *= 2; stack = (uint32_t*)realloc(stack, scap * 4); }
        SPAIR_ROOM();
        spair[sp] = 0xFFFFu;
        stack[sp++] = id;
      }
      /* the step limit: one config per iteration (the narrow kernel's schedule) ends with the ROUND that exceeded it; K configs per
       * iteration (the wide kernel's) with the ITERATION -- every popped config is then expanded completely, so the state the search
       * is left in (stacks, visited set) is one it can be taken up from (csrc/wgl_beam.hip BeamArgs.resume: the race of list orders) */
      if (K == 1 && max_probes && st->probes > max_probes) { verdict = -1; break; }
    }
    if (K > 1 && verdict == -2 && max_probes && st->probes > max_probes) verdict = -1;
    if (sp > st->max_stack) st->max_stack = sp;
  }

  out->valid = verdict;
  out->steps = st->probes; out->probes = st->probes; out->visited = st->visited;
  out->backtracks = st->expanded; out->max_depth = st->max_stack;
  if (verdict == 1) {
    /* witness: path root -> win_parent, then win_op (nothing at all when the normalised root passed every completion) */
    uint32_t len = root_wins ? 0 : 1, id = win_parent;
    while (!root_wins && ar.parent[id]) { len++; id = ar.parent[id]; }
    out->n_witness = len; out->final_state = win_state;
    if (witness) {
      uint32_t w = len ? len - 1 : 0; id = win_parent;
      if (!root_wins) { witness[w] = win_op; while (ar.parent[id]) { witness[--w] = ar.op[id]; id = ar.parent[id]; } }
      const int etx = g_eager_txns && model->kind == O_MULTI_REGISTER;
#define ABSORB(c2_, fr_, s_) (etx ? absorb_txns(model, c2_, fr_, s_, R, off, lst, process, ret_op, f, a, b, witness, &nw) \
                                  : absorb_reads(c2_, fr_, s_, R, off, lst, process, ret_op, f, a, witness, &nw))
      if (etx || (g_eager_reads && !cfgm && (model->kind == O_REGISTER || model->kind == O_CAS_REGISTER))) {
        /* the chain holds the branching ops only: replay it from the root, absorbing reads as the search did */
        uint32_t* chain = (uint32_t*)malloc(4 * (size_t)len);
        memcpy(chain, witness, 4 * (size_t)len);
        uint64_t* c2 = (uint64_t*)calloc(KW, 8);
        uint32_t fr = 0, nw = 0; int32_t s = model->init;
        if (branch) fr = absorb_reads(c2, 0, s, R, off, lst, process, ret_op, f, a, witness, &nw);      /* the root's own reads first */
        for (uint32_t i = 0; i < len; i++) {
          const uint32_t op = chain[i], p = (uint32_t)process[op];
          int32_t s2 = s; (void)oracle_step(model, s, f[op], a[op], b[op], &s2); s = s2;
          witness[nw++] = op;
          c2[1 + (p >> 6)] |= 1ull << (p & 63);
          if (ret_rank[op] == fr) {
            uint32_t pp = p;
            for (;;) {
              c2[1 + (pp >> 6)] &= ~(1ull << (pp & 63));
              fr++;
              if (fr == R) break;
              pp = (uint32_t)process[ret_op[fr]];
              if (!(c2[1 + (pp >> 6)] >> (pp & 63) & 1)) break;
            }
          }
          fr = ABSORB(c2, fr, s);
        }
#undef ABSORB
        out->n_witness = nw;
        free(chain); free(c2);
      }
    }
  } else if (verdict == 0) {
    out->fail_op = ret_op[maxf];
    out->prev_ok_op = maxf ? ret_op[maxf - 1] : 0xFFFFFFFFu;
    free(g_bcfg); g_bcfg_n = 0; g_bcfg_kw = KW;
    g_bcfg = (uint64_t*)malloc(ar.n * KW * 8);
    for (size_t i = 1; i < ar.n; i++)
      if ((uint32_t)ar.keys[i * KW] == maxf + 1) { memcpy(g_bcfg + (size_t)g_bcfg_n * KW, ar.keys + i * KW, KW * 8); g_bcfg_n++; }
    g_bsort_kw = KW;
    qsort(g_bcfg, g_bcfg_n, KW * 8, cmp_bcfg);
  }
  if (branch) { free(coff); free(clst); }
  free(rets); free(ret_rank); free(inv_rank); free(ret_op); free(off); free(ncr); free(lst); free(fill); free(crashed); free(prev_twin);
  free(open_ops); free(open_lin);
  free(ar.keys); free(ar.parent); free(ar.op); free(ar.slots); free(stack); free(dstack); free(unchecked); free(spair); free(key); free(ck);
  return 0;
}
