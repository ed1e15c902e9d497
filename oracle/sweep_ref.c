/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE (see oracle_model.h for the rules
 * and the PARITY UNPINNED statement).
 *
 * sweep_ref.c -- scalar CPU statement of the LEVEL SWEEP that
 * jepsen-tigerbeetle_amd/csrc/jit_sweep.hip runs: Lowe's just-in-time
 * linearization (the algorithm behind knossos.linear/analysis, SURVEY.md
 * section 8a rows `knossos.linear/analysis + knossos.linear.config`), with the two
 * dominance rules of wgl_beam.c (eager reads, twin rule) applied to the config
 * set, and -- the reason it exists -- cut into SEGMENTS that are swept
 * independently of each other and composed afterwards, so that ONE history is
 * checked by many wavefronts at once.
 *
 * Level F (F = 0..R, R = number of completions) holds the set of configs
 * (mask, state) reachable with exactly the first F completions passed:
 *   mask  = one bit per process slot, "the call this process has open at front F
 *           is already linearized";
 *   state = model state;
 *   normal form: every open live read whose value is nil or the state is linearized
 *           (eager reads: such a read can be linearized now without loss of generality).
 * Transition F -> F+1, X = the call completing at rank F:
 *   sub-round 0: configs that have X linearized return it (bit cleared) and enter level F+1
 *                -- normalised again, because calls invoked between the two completions are
 *                open now; the others form P_0;
 *   sub-round j: every config of P_(j-1) is expanded over its open, not yet linearized calls y
 *                (live ones in slot order, then crashed ones in invocation order; never a read
 *                -- a read that could be linearized is linearized already -- and never a call
 *                dominated under the twin rule): child = normal form of (mask + y, step(state, y));
 *                a child that has X linearized enters level F+1 as above, any other child
 *                joins P_j (exact de-duplication inside P_j and inside level F+1);
 *   P_j empty: the level is done.  Level F+1 empty  =>  NOT linearizable, failing completion = X.
 * Linearizing only up to X is complete: every linearization can be rearranged so that each call
 * takes effect immediately before some completion that needs it (Lowe 2017, section 5).
 * All R levels passed  =>  linearizable.
 *
 * Segments.  The fronts are cut at C_0 = 0 < C_1 < ... < C_S = R.  Segment i sweeps levels
 * C_i .. C_(i+1) starting from EVERY config that is possible at front C_i at all -- each
 * (subset of the calls open at C_i, model state) in normal form, numbered 0..n_origins-1 -- and
 * carries with every config the set of origins it is reachable from (a bit set; duplicates OR
 * their masks -- sub-rounds go by number of calls linearized, so a config's mask is final before it
 * is expanded).  What a segment hands on is its relation {origin -> configs at front C_(i+1)}.
 * Composition walks the segments in order: live set := {initial config}; for each segment the
 * live end configs are those whose origin mask meets the live set; they are translated into
 * origin numbers of the next segment.  An empty live set in segment i => NOT linearizable, and
 * the failing completion is the greatest level of segment i that some live origin still reached
 * (+ C_i): per origin the sweep records the last level at which a config carrying it existed.
 * Cuts (parallel by construction -- the kernel places each one with its own thread): with T = the
 * wanted segment length, cut k = 1, 2, ... is the front F in [k*T, (k+1)*T) with the FEWEST calls open (the
 * first of them) among those where no crashed call is open, provided at most m calls are open there (a crashed call stays open for ever, so there are no
 * cuts after the first crash); a window without such a front has no cut.  The state domain of the
 * origins is {nil, 0 .. vmax} (vmax = the greatest register value in the history or the model),
 * nd = vmax + 2 states; m = the greatest value <= max_cut_open with nd * 2^m <= 128 (the kernel gives every
 * 32 origins of a segment a wavefront of their own); nd > 128 => one segment.  Only the register family is cut.
 *
 * Outputs that are properties of (model, history): verdict, failing op, previous-ok op.  Outputs
 * that are properties of the sweep and compared bit for bit with the kernel: the size of every
 * level of every segment summed (configs_total), the largest level, the number of expansions
 * (probes) and of sub-rounds.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "oracle_model.h"
#include "../include/tbcheck.h"      /* tbc_sweep_rel: the record ranks exchange (the multi-GPU stand-in below writes it) */

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

typedef struct sweep_stats {
  uint64_t configs_total;   /* sum over all levels of all segments of the level's size */
  uint64_t max_level;       /* largest level */
  uint64_t probes;          /* expansions: (config, call) pairs whose model step is consistent */
  uint64_t subrounds;       /* expansion sub-rounds over all levels */
  uint64_t levels;          /* levels swept (all segments) */
  uint64_t n_segments;
  uint64_t max_origins;
  uint64_t max_pending;     /* largest P_j */
  uint64_t longest_segment; /* levels */
  uint64_t n_waves;         /* wavefronts the kernel uses: one per 32 origins of each segment */
  uint64_t max_segment_probes;
  /* design study (sweep_set_model_lanes): lane-steps of the most expensive wavefront if a level's passes ran `lanes` pairs at a time:
   * sum over its levels of ceil(configs / lanes) + sum over sub-rounds of ceil(pending configs * 2^ceil(log2 open calls) / lanes), and its levels */
  uint64_t critical_steps, critical_levels, total_steps;
} sweep_stats;

/* ordered exact set of configs: entries (KW words key: [state+1 | 0][mask...]) + origin mask, insertion order kept */
typedef unsigned __int128 orgset;     /* up to 128 origins per segment (the kernel splits them over wavefronts, 32 each) */
typedef struct { uint64_t* key; orgset* org; uint32_t* slots; size_t n, cap, nslots, kw; } cset;
static void cs_init(cset* s, size_t kw) {
  s->kw = kw; s->cap = 64; s->n = 0; s->nslots = 256;
  s->key = (uint64_t*)malloc(s->cap * kw * 8); s->org = (orgset*)malloc(s->cap * sizeof(orgset));
  s->slots = (uint32_t*)calloc(s->nslots, 4);
}
static void cs_free(cset* s) { free(s->key); free(s->org); free(s->slots); }
static void cs_clear(cset* s) { memset(s->slots, 0, s->nslots * 4); s->n = 0; }
static uint64_t cs_hash(const uint64_t* k, size_t kw) {
  uint64_t h = mix64(k[0]);
  for (size_t i = 1; i < kw; i++) h = mix64(h ^ k[i]) + 0x9E3779B97F4A7C15ull;
  return h;
}
/* returns 1 if new */
static int cs_add(cset* s, const uint64_t* k, orgset org) {
  size_t j = cs_hash(k, s->kw) & (s->nslots - 1);
  while (s->slots[j]) {
    const size_t e = s->slots[j] - 1;
    if (memcmp(s->key + e * s->kw, k, s->kw * 8) == 0) { s->org[e] |= org; return 0; }
    j = (j + 1) & (s->nslots - 1);
  }
  if (s->n == s->cap) {
    s->cap *= 2;
    s->key = (uint64_t*)realloc(s->key, s->cap * s->kw * 8); s->org = (orgset*)realloc(s->org, s->cap * sizeof(orgset));
  }
  memcpy(s->key + s->n * s->kw, k, s->kw * 8); s->org[s->n] = org;
  s->slots[j] = (uint32_t)++s->n;
  if (s->n * 2 > s->nslots) {
    free(s->slots); s->nslots *= 4; s->slots = (uint32_t*)calloc(s->nslots, 4);
    for (size_t e = 0; e < s->n; e++) {
      size_t q = cs_hash(s->key + e * s->kw, s->kw) & (s->nslots - 1);
      while (s->slots[q]) q = (q + 1) & (s->nslots - 1);
      s->slots[q] = (uint32_t)(e + 1);
    }
  }
  return 1;
}

/* knobs of the specified sweep */
static uint32_t g_eager = 1, g_twin = 1, g_seg_target = 0, g_max_cut_open = 3;
void sweep_set_rules(uint32_t eager, uint32_t twin) { g_eager = eager; g_twin = twin; }
/* seg_target: wanted segment length in completions (0 = one segment); a cut is placed at the first front
 * at or after each multiple of it where at most max_cut_open calls are open and none of them is crashed */
void sweep_set_segments(uint32_t seg_target, uint32_t max_cut_open) { g_seg_target = seg_target; g_max_cut_open = max_cut_open; }
/* n_dom: states of the origin domain (nil + 0..n_dom-2); 0 = from the history's own greatest value.  A batch of
 * histories shares one domain in the library (the greatest value of the whole batch). */
static uint32_t g_n_dom = 0;
void sweep_set_domain(uint32_t n_dom) { g_n_dom = n_dom; }
/* ids per segment's origin space: 128 = the kernel's (4 wavefronts of 32 origins per segment, tbc_sweep_rel's four words per origin);
 * up to 512 for design studies (cuts at fronts with more calls open: scripts/sweep_cut_study.py) -- no relation export then */
#define SW_MAX_WORDS 16
#define SW_MAX_KW 20          /* key words of a config: 1 + mask words (<= 1,216 process slots) */
static uint32_t g_max_ids = 128;
/* origins per wavefront ("slice"): 32 = the kernel's; fewer for design studies (no relation export then) */
static uint32_t g_slice = 32, g_model_lanes = 64;
void sweep_set_model_lanes(uint32_t l) { g_model_lanes = l ? l : 64; }
void sweep_set_slice(uint32_t g) { g_slice = (g == 0 || g > 32) ? 32 : g; }
void sweep_set_max_ids(uint32_t n) { g_max_ids = n < 32 ? 32 : (n > 32 * SW_MAX_WORDS ? 32 * SW_MAX_WORDS : n); }
/* Multi-GPU stand-in (tests/test_distributed_gloo.py): instead of composing, write the relation of every wavefront
 * with (segment * 4 + slice) % world == rank into rel[segment * 4 + slice] -- exactly what rank `rank` of `world`
 * GPUs leaves in its table after tbc_batch_sweep_partial -- and sweep nothing else.  rel = NULL switches it off. */
static tbc_sweep_rel* g_rel = NULL; static uint32_t g_rel_segs = 0, g_rel_rank = 0, g_rel_world = 1;
void sweep_set_export(void* rel, uint32_t max_segs, uint32_t rank, uint32_t world) {
  g_rel = (tbc_sweep_rel*)rel; g_rel_segs = max_segs; g_rel_rank = rank; g_rel_world = world ? world : 1;
}

/* lookahead (the wide search's rule, wgl_beam.c, applied to the sweep; 0 = off, else the completions looked at = 8 in the kernels):
 * a config at front F is DEAD when the call completing at one of the next `depth` ranks can never be linearized from it -- it is not
 * linearized yet and needs a register value (0..31) that is neither the state nor produced (:write v / :cas [_ v]) by any OTHER call
 * that can still be linearized before that completion: a call open somewhere in [F, t] (crashed ones invoked before t included) that is
 * not open-at-F-and-linearized.  No linearization goes through a dead config, so the sweep drops it where it would enter a set: an
 * origin, a pending child (at its level's front), a config entering the next level (at that front).  A verdict VALID is therefore
 * unchanged; a set may run EMPTY EARLIER than without the rule (the configs were doomed, not yet refuted), so an INVALID verdict under
 * the rule names a completion at or before the failing one -- the library sweeps such a history once more without the rule for
 * its failing op and :configs.  Counters (levels' sizes, probes, sub-rounds) are the rule's own. */
static uint32_t g_look = 0;
void sweep_set_lookahead(uint32_t depth) { g_look = depth; }

/* RELAXED sweep of a history with crashed (:info) calls (round 5; the refutation pass of the count form, DESIGN.md section 2.4, as a
 * level sweep): every CLASS of crashed calls with the same effect -- (:write v), (:cas [a b]) with a != b on a cas-register -- is an
 * unlimited supply from the moment its first member is invoked: it may take effect any number of times, at any later point.  That is a
 * SUPERSET of the linearizations (each real crashed call takes effect at most once), so INVALID at completion t proves the history
 * invalid with its first bad completion at t or earlier; VALID proves nothing.  In this reading a crashed call holds no process slot
 * and no mask bit (live calls take re-used slots exactly as the count form numbers them: the lowest slot free when the process first
 * invokes, a process that crashes hands its slot back), so a config is (mask over live slots, state) as in a crash-free history, every
 * front with few open calls is a cut, and the history is swept by hundreds of wavefronts at once.
 * A class step changes the state and nothing else.  With reach_F[s] = the states reachable from s through one or more steps of the
 * classes available at front F (a transitive closure over at most 32 states), the expansion of a config (mask, s) in a sub-round is over
 * COMPOUND steps: a hop to a state s' in {s} + reach_F[s] (no hop, or class steps), the open reads of s' absorbed (eager rule), and then
 * either an open live call y the state s' allows -- child = normal form of (mask + reads(s') + y, step(s', y)); after a real hop only
 * a :cas expecting s' (the LAZY rule of the count form: a crashed call is worth linearizing only for a call that observes its value --
 * one whose value is overwritten unobserved can be deleted from any linearization) -- or, after a real hop that absorbed at least
 * one read, nothing more -- child = (mask + reads(s'), s').  Every child has strictly more calls linearized
 * than its parent, so the sub-rounds end as they do without crashed calls, and no closure of whole sets is needed: class steps that
 * nothing follows before a completion can be moved behind it.  A compound step whose parts are all allowed counts as one probe. */
static uint32_t g_relaxed = 0;
void sweep_set_relaxed(uint32_t on) { g_relaxed = on; }
static _Thread_local uint64_t g_look_dropped = 0;
uint64_t sweep_look_dropped(void) { return g_look_dropped; }

typedef struct {
  uint32_t n, R, W, MW, KW;
  const uint8_t* f; const int32_t* a; const int32_t* b; const int32_t* process;
  uint32_t *ret_rank, *inv_rank, *ret_op, *off, *ncr, *lst, *crashed;
  const oracle_model* model;
} hist_t;

static inline int bit(const uint64_t* m, uint32_t p) { return (int)(m[p >> 6] >> (p & 63) & 1); }
static inline void setb(uint64_t* m, uint32_t p) { m[p >> 6] |= 1ull << (p & 63); }
static inline void clrb(uint64_t* m, uint32_t p) { m[p >> 6] &= ~(1ull << (p & 63)); }

/* normal form at front F: linearize every open live read the state allows */
static void normalise(const hist_t* H, uint64_t* key, uint32_t F) {
  if (!g_eager) return;
  const int32_t s = (int32_t)(uint32_t)(key[0] >> 32);
  for (uint32_t c = H->off[F]; c < H->off[F + 1]; c++) {
    const uint32_t x = H->lst[c];
    if (H->f[x] == O_READ && (H->a[x] == O_NIL || H->a[x] == s)) setb(key + 1, (uint32_t)H->process[x]);
  }
}

static int is_dead(const hist_t* H, const uint64_t* key, uint32_t F) {
  if (!g_look || F >= H->R) return 0;
  const int32_t s = (int32_t)(uint32_t)(key[0] >> 32);
  for (uint32_t j = 0; j < g_look && F + j < H->R; j++) {
    const uint32_t t = F + j, fop = H->ret_op[t], pf = (uint32_t)H->process[fop];
    if (!((H->f[fop] == O_READ && H->a[fop] != O_NIL) || H->f[fop] == O_CAS)) continue;
    const int32_t v = H->a[fop];
    if (v < 0 || v >= 32) continue;
    if (H->inv_rank[fop] <= F && bit(key + 1, pf)) continue;                 /* already linearized */
    if (v == s) continue;
    int ok = 0;
    for (uint32_t F2 = F; F2 <= t && !ok; F2++) {                             /* calls open somewhere in [F, t] */
      const uint32_t nl = H->off[F2 + 1] - H->off[F2], tot = nl + (F2 == t ? H->ncr[F2] : 0);
      for (uint32_t cc = 0; cc < tot && !ok; cc++) {
        const uint32_t x = cc < nl ? H->lst[H->off[F2] + cc] : H->crashed[cc - nl];
        if (x == fop) continue;
        if (H->inv_rank[x] <= F && bit(key + 1, (uint32_t)H->process[x])) continue;   /* open at F, linearized */
        if ((H->f[x] == O_WRITE && H->a[x] == v) || (H->f[x] == O_CAS && H->b[x] == v)) ok = 1;
      }
    }
    if (!ok) { g_look_dropped++; return 1; }
  }
  return 0;
}

/* a config that has X (slot px) linearized passes completion F: into level F+1 (front F+1 < R) or the end set */
static void pass_level(const hist_t* H, cset* nxt, const uint64_t* key, orgset org, uint32_t px, uint32_t F, uint64_t* tmp) {
  memcpy(tmp, key, H->KW * 8);
  clrb(tmp + 1, px);
  if (F + 1 < H->R) normalise(H, tmp, F + 1);
  if (is_dead(H, tmp, F + 1)) return;
  cs_add(nxt, tmp, org);
}

int sweep_ref_check(uint32_t n, const uint8_t* f, const int32_t* a, const int32_t* b,
                    const int32_t* process, uint32_t n_process,
                    const uint32_t* inv_pos, const uint32_t* ret_pos,
                    const oracle_model* model, uint64_t max_level_limit,
                    uint32_t* level_sizes /* R entries or NULL */,
                    oracle_result* out, sweep_stats* st) {
  memset(out, 0, sizeof *out); memset(st, 0, sizeof *st);
  out->fail_op = out->prev_ok_op = 0xFFFFFFFFu;
  if (!(model->kind == O_REGISTER || model->kind == O_CAS_REGISTER || model->kind == O_MUTEX || model->kind == O_TABLE ||
        model->kind == O_MULTI_REGISTER)) return 3;
  const int regfam = model->kind == O_REGISTER || model->kind == O_CAS_REGISTER;
  uint32_t R = 0;
  for (uint32_t i = 0; i < n; i++) {
    if (i && inv_pos[i] <= inv_pos[i - 1]) return 2;
    if (process[i] < 0 || (uint32_t)process[i] >= n_process) return 2;
    if (ret_pos[i] != O_CRASHED) { if (ret_pos[i] <= inv_pos[i]) return 2; R++; }
  }
  if (R == 0) { out->valid = 1; out->final_state = model->init; return 0; }
  hist_t H; H.n = n; H.R = R; H.W = n_process; H.MW = (n_process + 63) / 64; H.KW = 1 + H.MW;
  H.f = f; H.a = a; H.b = b; H.process = process; H.model = model;
  posop* rets = (posop*)malloc(sizeof(posop) * R);
  H.ret_rank = (uint32_t*)malloc(4 * (size_t)n); H.inv_rank = (uint32_t*)malloc(4 * (size_t)n); H.ret_op = (uint32_t*)malloc(4 * (size_t)R);
  uint32_t k = 0;
  for (uint32_t i = 0; i < n; i++) if (ret_pos[i] != O_CRASHED) { rets[k].pos = ret_pos[i]; rets[k].op = i; k++; }
  qsort(rets, R, sizeof(posop), cmp_posop);
  for (uint32_t r = 0; r < R; r++) { H.ret_rank[rets[r].op] = r; H.ret_op[r] = rets[r].op; }
  { uint32_t r = 0;
    for (uint32_t i = 0; i < n; i++) { while (r < R && rets[r].pos < inv_pos[i]) r++; H.inv_rank[i] = r; } }
  for (uint32_t i = 0; i < n; i++) if (ret_pos[i] == O_CRASHED) H.ret_rank[i] = 0xFFFFFFFFu;
  /* ---- RELAXED: re-used process slots for the live calls, the crashed calls as classes of effects (see sweep_set_relaxed) */
  const int relaxed = g_relaxed && regfam;
  int32_t* slot_arr = NULL;
  uint32_t ncls = 0;
  uint32_t *cls_f = NULL, *cls_from = NULL; int32_t *cls_a = NULL, *cls_b = NULL;
  if (relaxed) {
    slot_arr = (int32_t*)calloc((size_t)n + 1, 4);
    int32_t* slot_of = (int32_t*)malloc(4 * ((size_t)n_process + 1));
    uint8_t* used = (uint8_t*)calloc((size_t)n_process + 2, 1);
    uint32_t W = 1;
    for (uint32_t p = 0; p < n_process; p++) slot_of[p] = -1;
    cls_f = (uint32_t*)malloc(4 * ((size_t)n + 1)); cls_from = (uint32_t*)malloc(4 * ((size_t)n + 1));
    cls_a = (int32_t*)malloc(4 * ((size_t)n + 1)); cls_b = (int32_t*)malloc(4 * ((size_t)n + 1));
    for (uint32_t i = 0; i < n; i++) {
      const uint32_t p = (uint32_t)process[i];
      if (ret_pos[i] == O_CRASHED) {
        if (slot_of[p] >= 0) { used[slot_of[p]] = 0; slot_of[p] = -1; }
        if (!(f[i] == O_WRITE || (f[i] == O_CAS && model->kind == O_CAS_REGISTER && a[i] != b[i]))) continue;
        uint32_t c = 0;
        while (c < ncls && !(cls_f[c] == f[i] && cls_a[c] == a[i] && (f[i] != O_CAS || cls_b[c] == b[i]))) c++;
        if (c == ncls) { cls_f[c] = f[i]; cls_a[c] = a[i]; cls_b[c] = f[i] == O_CAS ? b[i] : 0; cls_from[c] = H.inv_rank[i]; ncls++; }
        continue;
      }
      if (slot_of[p] < 0) { uint32_t s = 0; while (used[s]) s++; used[s] = 1; slot_of[p] = (int32_t)s; if (s + 1 > W) W = s + 1; }
      slot_arr[i] = slot_of[p];
    }
    free(slot_of); free(used);
    process = slot_arr; n_process = W;
    H.process = process; H.W = W; H.MW = (W + 63) / 64; H.KW = 1 + H.MW;
    if (H.KW > SW_MAX_KW) { free(slot_arr); free(cls_f); free(cls_from); free(cls_a); free(cls_b); free(rets); free(H.ret_rank); free(H.inv_rank); free(H.ret_op); return 3; }
  }
  const uint32_t KW = H.KW;
  H.off = (uint32_t*)calloc((size_t)R + 1, 4); H.ncr = (uint32_t*)calloc((size_t)R + 1, 4);
  uint32_t n_crashed = 0;
  const int eager_on = g_eager && regfam;
  for (uint32_t i = 0; i < n; i++) {
    if (H.ret_rank[i] == 0xFFFFFFFFu) {
      if (relaxed) continue;                                          /* no slot, no list entry: a class */
      if (regfam && f[i] == O_READ && a[i] == O_NIL) continue;       /* crashed read: no effect, no constraint */
      n_crashed++; if (H.inv_rank[i] < R) H.ncr[H.inv_rank[i]]++; continue;
    }
    for (uint32_t fr = H.inv_rank[i]; fr <= H.ret_rank[i]; fr++) H.off[fr + 1]++;
  }
  for (uint32_t r = 0; r < R; r++) H.off[r + 1] += H.off[r];
  for (uint32_t r = 1; r < R; r++) H.ncr[r] += H.ncr[r - 1];
  H.lst = (uint32_t*)malloc(4 * ((size_t)H.off[R] + 1));
  uint32_t* fill = (uint32_t*)malloc(4 * ((size_t)R + 1));
  memcpy(fill, H.off, 4 * ((size_t)R + 1));
  H.crashed = (uint32_t*)malloc(4 * ((size_t)n_crashed + 1));
  { uint32_t c = 0;
    for (uint32_t i = 0; i < n; i++) {
      if (H.ret_rank[i] == 0xFFFFFFFFu) { if (!relaxed && !(regfam && f[i] == O_READ && a[i] == O_NIL)) H.crashed[c++] = i; continue; }
      for (uint32_t fr = H.inv_rank[i]; fr <= H.ret_rank[i]; fr++) H.lst[fill[fr]++] = i;
    }
    for (uint32_t fr = 0; fr < R; fr++)          /* slot order */
      for (uint32_t x = H.off[fr] + 1; x < H.off[fr + 1]; x++) {
        uint32_t v = H.lst[x], y = x;
        while (y > H.off[fr] && process[H.lst[y - 1]] > process[v]) { H.lst[y] = H.lst[y - 1]; y--; }
        H.lst[y] = v;
      } }
  const uint32_t save_eager = g_eager; g_eager = (uint32_t)eager_on;

  /* ---- state domain of the origins: nil + 0..vmax (what the kernel's per-front read table is indexed by) */
  int32_t vmax = model->init == O_NIL ? -1 : model->init;
  if (regfam) for (uint32_t i = 0; i < n; i++) {
    if (a[i] != O_NIL && a[i] > vmax) vmax = a[i];
    if (f[i] == O_CAS && b[i] > vmax) vmax = b[i];
  }
  const uint32_t nd = g_n_dom ? g_n_dom : (uint32_t)(vmax + 2);
  uint32_t m_open = 0;
  int cut_ok = regfam && g_seg_target && nd <= g_max_ids;
  if (cut_ok) while (m_open < g_max_cut_open && (nd << (m_open + 1)) <= g_max_ids) m_open++;
  /* ---- cuts: in every window the front with the FEWEST open calls (the first of them), if that is <= m_open */
  uint32_t* cuts = (uint32_t*)malloc(4 * ((size_t)R + 2));
  uint32_t* cutk = (uint32_t*)malloc(4 * ((size_t)R + 2));     /* window number of each cut = the kernel's segment number */
  uint32_t S = 0;
  cutk[S] = 0; cuts[S++] = 0;
  if (cut_ok) {
    for (uint32_t k = 1; (uint64_t)k * g_seg_target < R; k++) {
      const uint32_t lo = k * g_seg_target, hi = lo + g_seg_target < R ? lo + g_seg_target : R;
      uint32_t best = 0xFFFFFFFFu, bestF = 0;
      for (uint32_t F = lo; F < hi; F++)
        if (H.ncr[F] == 0 && H.off[F + 1] - H.off[F] < best) { best = H.off[F + 1] - H.off[F]; bestF = F; }
      if (best <= m_open) { cutk[S] = k; cuts[S++] = bestF; }
    }
  }
  cuts[S] = R;
  st->n_segments = S;
  int32_t* dom = (int32_t*)malloc(4 * ((size_t)nd + 1));
  dom[0] = O_NIL;
  for (uint32_t q = 1; q < nd; q++) dom[q] = (int32_t)q - 1;

  /* RELAXED: reach[si] at a front = closure over the classes whose first member is invoked by then (they are in that order) */
  uint32_t reach[32]; uint32_t reach_n = 0xFFFFFFFFu;
  memset(reach, 0, sizeof reach);
#define REACH_AT(F_) do { if (relaxed) { uint32_t na_ = 0; while (na_ < ncls && cls_from[na_] <= (F_)) na_++; \
    if (na_ != reach_n) { reach_n = na_; memset(reach, 0, sizeof reach); \
      for (int again_ = 1; again_; ) { again_ = 0; \
        for (uint32_t si_ = 0; si_ < nd && si_ < 32; si_++) for (uint32_t c_ = 0; c_ < na_; c_++) { \
          if (cls_f[c_] == O_CAS && dom[si_] != cls_a[c_]) continue; \
          const int32_t tv_ = cls_f[c_] == O_WRITE ? cls_a[c_] : cls_b[c_]; \
          const uint32_t ti_ = (uint32_t)(tv_ + 1); if (ti_ >= nd || ti_ >= 32) continue; \
          const uint32_t add_ = (1u << ti_) | reach[ti_]; \
          if ((reach[si_] | add_) != reach[si_]) { reach[si_] |= add_; again_ = 1; } } } } } } while (0)
  cset cur, nxt, pa, pb;
  cs_init(&cur, KW); cs_init(&nxt, KW); cs_init(&pa, KW); cs_init(&pb, KW);
  uint64_t* key = (uint64_t*)calloc(KW, 8); uint64_t* tmp = (uint64_t*)calloc(KW, 8);
  int verdict = 1;
  uint32_t fail_level = 0;
  int32_t final_state = model->init;
  /* the composition's live set: ids of the current segment's origin space (<= 128), segment 0: id 0 = the initial config */
  uint32_t live[SW_MAX_WORDS] = {1}, next_live[SW_MAX_WORDS];

  for (uint32_t sg = 0; sg < S && verdict == 1; sg++) {
    const uint32_t F0 = cuts[sg], F1 = cuts[sg + 1];
    /* origin space of this segment: id = state index * 2^no + (bit c = the c-th call open at F0 is linearized); the
     * kernel gives every 32 consecutive ids a wavefront of their own ("slice"), and so does this restatement -- the
     * statistics count a config once per slice it is reachable in.  Ids that are not in normal form are nobody's config. */
    const uint32_t no0 = sg ? H.off[F0 + 1] - H.off[F0] : 0;
    const uint32_t n_ids = sg ? nd << no0 : 1, n_slices = (n_ids + g_slice - 1) / g_slice;
    const uint32_t no1 = F1 < R ? H.off[F1 + 1] - H.off[F1] : 0;
    if (n_ids > st->max_origins) st->max_origins = n_ids;
    if (F1 - F0 > st->longest_segment) st->longest_segment = F1 - F0;
    memset(next_live, 0, sizeof next_live);
    uint32_t reached = F0;
    for (uint32_t sl = 0; sl < n_slices && verdict == 1; sl++) {
      if (g_rel && (cutk[sg] * 4 + sl) % g_rel_world != g_rel_rank) continue;      /* another rank's wavefront */
      if (g_rel && cutk[sg] >= g_rel_segs) { verdict = -3; break; }
      const uint64_t cfg_before = st->configs_total, sub_before = st->subrounds;
      uint32_t slice_max = 0;
      cs_clear(&cur);
      for (uint32_t l = 0; l < g_slice; l++) {
        const uint32_t id = g_slice * sl + l;
        if (id >= n_ids) break;
        memset(key, 0, KW * 8);
        if (sg == 0) key[0] = (uint64_t)(uint32_t)model->init << 32;
        else {
          key[0] = (uint64_t)(uint32_t)dom[id >> no0] << 32;
          for (uint32_t c = 0; c < no0; c++) if (id >> c & 1) setb(key + 1, (uint32_t)process[H.lst[H.off[F0] + c]]);
        }
        memcpy(tmp, key, KW * 8);
        normalise(&H, tmp, F0);
        if (sg == 0) memcpy(key, tmp, KW * 8);                       /* the initial config is taken in normal form */
        else if (memcmp(tmp, key, KW * 8) != 0) continue;             /* not in normal form: no config has this id */
        if (is_dead(&H, key, F0)) continue;
        cs_add(&cur, key, (orgset)1 << l);
      }
      if (cur.n == 0) continue;                                       /* the kernel's wavefront has nothing to sweep */
      st->n_waves++;
      const uint64_t probes_before = st->probes;
      uint64_t steps = 0, levels = 0;
      uint32_t last_level[32]; for (uint32_t q = 0; q < 32; q++) last_level[q] = F0;
      uint32_t M[32][SW_MAX_WORDS]; memset(M, 0, sizeof M);

      for (uint32_t F = F0; F < F1 && verdict == 1; F++) {
        const uint32_t x = H.ret_op[F], px = (uint32_t)process[x];
        cs_clear(&nxt); cs_clear(&pa);
        for (size_t e = 0; e < cur.n; e++) {
          const uint64_t* c = cur.key + e * KW;
          if (bit(c + 1, px)) pass_level(&H, &nxt, c, cur.org[e], px, F, tmp);
          else cs_add(&pa, c, cur.org[e]);
        }
        cset* P = &pa; cset* Q = &pb;
        levels++; steps += (cur.n + g_model_lanes - 1) / g_model_lanes;
        const uint32_t nlive = H.off[F + 1] - H.off[F], tot = nlive + H.ncr[F];
        while (P->n) {
          st->subrounds++;
          { uint32_t gs = 0; while ((1u << gs) < tot) gs++; steps += (((uint64_t)P->n << gs) + g_model_lanes - 1) / g_model_lanes; }
          if (P->n > st->max_pending) st->max_pending = P->n;
          cs_clear(Q);
          if (relaxed) REACH_AT(F);
          for (size_t e = 0; relaxed && e < P->n; e++) {             /* RELAXED: compound steps (see sweep_set_relaxed) */
            const uint64_t* c = P->key + e * KW; const orgset org = P->org[e];
            const int32_t s0 = (int32_t)(uint32_t)(c[0] >> 32);
            const uint32_t si = s0 == O_NIL ? 0u : (uint32_t)(s0 + 1);
            for (uint32_t t = 0; t < nd && t < 32; t++) {
              const int hop = t != si;
              if (hop && !(si < 32 && (reach[si] >> t & 1u))) continue;
              const int32_t s1 = dom[t];
              memcpy(key, c, KW * 8); key[0] = (uint64_t)(uint32_t)s1 << 32;
              normalise(&H, key, F);                                  /* the open reads of s1 */
              uint64_t base[SW_MAX_KW]; memcpy(base, key, KW * 8);
              if (hop && memcmp(base + 1, c + 1, (KW - 1) * 8) != 0) {      /* the hop alone, if it absorbed a read */
                st->probes++;
                if (bit(base + 1, px)) pass_level(&H, &nxt, base, org, px, F, tmp);
                else cs_add(Q, base, org);
              }
              for (uint32_t cc = 0; cc < nlive; cc++) {
                const uint32_t y = H.lst[H.off[F] + cc], py = (uint32_t)process[y];
                if (bit(base + 1, py)) continue;
                if (eager_on && f[y] == O_READ) continue;
                if (hop && f[y] != O_CAS) continue;                  /* the lazy rule: a hop is worth taking only for a call that observes s1 */
                if (g_twin && (f[y] == O_WRITE || f[y] == O_CAS)) {
                  int dominated = 0;
                  for (uint32_t dd = 0; dd < nlive && !dominated; dd++) {
                    const uint32_t z = H.lst[H.off[F] + dd];
                    if (z == y || f[z] != f[y] || a[z] != a[y] || (f[y] == O_CAS && b[z] != b[y])) continue;
                    if (bit(base + 1, (uint32_t)process[z])) continue;
                    if (H.ret_rank[z] < H.ret_rank[y] || (H.ret_rank[z] == H.ret_rank[y] && z < y)) dominated = 1;
                  }
                  if (dominated) continue;
                }
                int32_t s2;
                if (!oracle_step(model, s1, f[y], a[y], b[y], &s2)) continue;
                st->probes++;
                memcpy(key, base, KW * 8);
                setb(key + 1, py);
                key[0] = (uint64_t)(uint32_t)s2 << 32;
                normalise(&H, key, F);
                if (bit(key + 1, px)) pass_level(&H, &nxt, key, org, px, F, tmp);
                else cs_add(Q, key, org);
              }
            }
          }
          for (size_t e = 0; !relaxed && e < P->n; e++) {
            const uint64_t* c = P->key + e * KW; const orgset org = P->org[e];
            const int32_t s0 = (int32_t)(uint32_t)(c[0] >> 32);
            for (uint32_t cc = 0; cc < tot; cc++) {
              const uint32_t y = cc < nlive ? H.lst[H.off[F] + cc] : H.crashed[cc - nlive], py = (uint32_t)process[y];
              if (bit(c + 1, py)) continue;
              if (eager_on && f[y] == O_READ) continue;                 /* could it be linearized it would be already */
              if (g_twin && regfam && (f[y] == O_WRITE || f[y] == O_CAS)) {
                int dominated = 0;
                for (uint32_t dd = 0; dd < tot && !dominated; dd++) {
                  const uint32_t z = dd < nlive ? H.lst[H.off[F] + dd] : H.crashed[dd - nlive];
                  if (z == y || f[z] != f[y] || a[z] != a[y] || (f[y] == O_CAS && b[z] != b[y])) continue;
                  if (bit(c + 1, (uint32_t)process[z])) continue;
                  if (H.ret_rank[z] < H.ret_rank[y] || (H.ret_rank[z] == H.ret_rank[y] && z < y)) dominated = 1;
                }
                if (dominated) continue;
              }
              int32_t s2;
              if (!oracle_step(model, s0, f[y], a[y], b[y], &s2)) continue;
              st->probes++;
              memcpy(key, c, KW * 8);
              setb(key + 1, py);
              key[0] = (uint64_t)(uint32_t)s2 << 32;
              normalise(&H, key, F);
              if (bit(key + 1, px)) pass_level(&H, &nxt, key, org, px, F, tmp);
              else if (!is_dead(&H, key, F)) cs_add(Q, key, org);
            }
          }
          { cset* t = P; P = Q; Q = t; }
        }
        st->levels++;
        st->configs_total += nxt.n;
        if (nxt.n > st->max_level) st->max_level = nxt.n;
        if (nxt.n > slice_max) slice_max = (uint32_t)nxt.n;
        if (level_sizes && S == 1) level_sizes[F] = (uint32_t)nxt.n;
        if (max_level_limit && nxt.n > max_level_limit) { verdict = -1; break; }
        { orgset any = 0; for (size_t e = 0; e < nxt.n; e++) any |= nxt.org[e];
          for (uint32_t q = 0; q < 32; q++) if (any >> q & 1) last_level[q] = F + 1; }
        { cset t = cur; cur = nxt; nxt = t; }
        if (cur.n == 0) break;                             /* nobody of this slice passes completion F */
      }
      if (verdict != 1) break;
      if (st->probes - probes_before > st->max_segment_probes) st->max_segment_probes = st->probes - probes_before;
      st->total_steps += steps;
      if (steps > st->critical_steps) { st->critical_steps = steps; st->critical_levels = levels; }
      /* the relation this slice hands on: origin -> ids of the next segment's origin space (last segment: final states) */
      for (size_t e = 0; e < cur.n; e++) {
        const uint64_t* c = cur.key + e * KW;
        const int32_t sv = (int32_t)(uint32_t)(c[0] >> 32);
        uint32_t sidx = 0xFFFFFFFFu;
        for (uint32_t q = 0; q < nd; q++) if (dom[q] == sv) { sidx = q; break; }
        uint32_t id2;
        if (F1 == R) id2 = regfam && sidx != 0xFFFFFFFFu && sidx < 32 ? sidx : 0;
        else {
          if (sidx == 0xFFFFFFFFu) { verdict = -3; break; }
          id2 = sidx << no1;
          for (uint32_t cc = 0; cc < no1; cc++) if (bit(c + 1, (uint32_t)process[H.lst[H.off[F1] + cc]])) id2 |= 1u << cc;
        }
        for (uint32_t q = 0; q < 32; q++) if (cur.org[e] >> q & 1) M[q][id2 >> 5] |= 1u << (id2 & 31);
      }
      if (g_rel) {
        tbc_sweep_rel* r = g_rel + (size_t)cutk[sg] * 4 + sl;
        memset(r, 0, sizeof *r);
        r->status = 1; r->F0 = F0; r->F1 = F1; r->n_org = 0; r->max_level = slice_max;
        r->subrounds = (uint32_t)(st->subrounds - sub_before); r->n_end = (uint32_t)cur.n;
        r->end_state = cur.n ? (uint32_t)(cur.key[0] >> 32) : 0u;
        r->configs_total = st->configs_total - cfg_before; r->probes = st->probes - probes_before;
        if (g_max_ids > 128 || g_slice != 32) { verdict = -3; break; }                /* (the exported record holds four words per origin) */
        for (uint32_t q = 0; q < 32; q++) memcpy(r->M[q], M[q], sizeof r->M[q]);
        memcpy(r->last_level, last_level, sizeof last_level);
        continue;
      }
      /* composition over this slice's live origins */
      for (uint32_t q = 0; q < g_slice; q++) if (live[(g_slice * sl + q) >> 5] >> ((g_slice * sl + q) & 31) & 1) {
        for (uint32_t w = 0; w < SW_MAX_WORDS; w++) next_live[w] |= M[q][w];
        if (last_level[q] > reached) reached = last_level[q];
      }
    }
    if (verdict != 1) break;
    if (g_rel) continue;                                   /* relations only: the ranks compose after the exchange */
    { uint32_t any_live = 0; for (uint32_t w = 0; w < SW_MAX_WORDS; w++) any_live |= next_live[w];
      if (any_live == 0) { verdict = 0; fail_level = reached; break; } }
    if (F1 == R) { for (uint32_t q = 0; q < 32; q++) if (next_live[0] >> q & 1) { final_state = regfam ? dom[q < nd ? q : 0] : model->init; break; } }
    memcpy(live, next_live, sizeof live);
  }
  if (verdict == 1 && !regfam) final_state = cur.n ? (int32_t)(uint32_t)(cur.key[0] >> 32) : model->init;

  if (g_rel && verdict == 1) verdict = -1;
  out->valid = verdict < -1 ? -1 : verdict;
  if (verdict == 1) out->final_state = final_state;
  if (verdict == 0) {
    out->fail_op = H.ret_op[fail_level];
    out->prev_ok_op = fail_level ? H.ret_op[fail_level - 1] : 0xFFFFFFFFu;
  }
  out->probes = st->probes; out->visited = st->configs_total; out->steps = st->subrounds;
  g_eager = save_eager;
  cs_free(&cur); cs_free(&nxt); cs_free(&pa); cs_free(&pb);
  free(key); free(tmp); free(dom); free(cuts); free(cutk); free(rets); free(fill);
  free(H.ret_rank); free(H.inv_rank); free(H.ret_op); free(H.off); free(H.ncr); free(H.lst); free(H.crashed);
  free(slot_arr); free(cls_f); free(cls_from); free(cls_a); free(cls_b);
#undef REACH_AT
  return verdict == -2 ? 4 : verdict == -3 ? 5 : 0;
}
