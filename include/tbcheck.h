/*
 * tbcheck.h -- C-ABI of libtbcheck.so, the MI355X-native linearizability checker.
 *
 * This is the drop-in boundary for the hot path named by BASELINE.json's
 * north_star: the Wing-Gong/Lowe search behind
 *
 *     (knossos.wgl/analysis model history)
 *     (knossos.linear/analysis model history)
 *     (knossos.competition/analysis model history)
 *     (jepsen.checker/linearizable {:model m :algorithm a})
 *
 * None of those live in /root/reference: the reference reaches Knossos only
 * transitively through `jepsen "0.2.8-SNAPSHOT"` (project.clj:8) and never
 * calls the search.  There is therefore no existing
 * FFI to replace; every entry point below is a new design, and each one cites
 * the Clojure function (recalled from the public jepsen-io/knossos and
 * jepsen-io/jepsen libraries) or the reference call site whose work it takes
 * over.  The binding a maintainer would add (JNA) is in INTEGRATION.md.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types.
 *   - every function returns a tbc_status (0 = OK); no exception or abort ever
 *     crosses the boundary.
 *   - all entry points are re-entrant: jepsen.checker/compose and
 *     jepsen.independent/checker evaluate sub-checkers from several JVM threads
 *     (reference call sites: set_full.clj:155-158, tests/ledger.clj:363-367,
 *     core.clj:139-146), so a call owns its own HIP stream and arena.
 *   - there is NO CPU fallback in this library: without a usable gfx950 device
 *     every compute entry point returns TBC_ERR_NO_DEVICE.
 */
#ifndef TBCHECK_H
#define TBCHECK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: tbc_opts grew to 64 bytes (lanes_per_history, reserved0), tbc_batch_sweep_finish gained merged_bytes, tbc_opts.dominance
 * gained TBC_DOM_NO_COUNT_FORM.  A caller built against another version must refuse the library (tbc_version()).
 * Still 2, additive: the word that was reserved0 (must be 0) is tbc_opts.list_order, 0 = the library's choice; tbc_batch_list_order().
 * Still 2, additive (round 6): tbc_batch_map_input / _submit_input / _reload / _input_info (fresh histories into a batch's arenas),
 * tbc_comm_* (the sharded sweep's exchange behind the C-ABI); tbc_setfull_create_rows takes its exception lists in any order again. */
#define TBC_ABI_VERSION 2u

/* ------------------------------------------------------------------ status */
typedef enum tbc_status {
  TBC_OK = 0,
  TBC_ERR_INVALID_ARG = 1,   /* null pointer, bad enum, inconsistent sizes      */
  TBC_ERR_BAD_HISTORY = 2,   /* unsorted invocations, return before invoke, ... */
  TBC_ERR_NO_DEVICE = 3,     /* no gfx950 device / HIP runtime failure          */
  TBC_ERR_OOM = 4,           /* host or device allocation failed                */
  TBC_ERR_WINDOW_TOO_WIDE = 5, /* more open processes than the kernels support  */
  TBC_ERR_MODEL = 6,         /* op not understood by the model / bad table      */
  TBC_ERR_HIP = 7,           /* a HIP call failed; see tbc_last_error()         */
  TBC_ERR_UNSUPPORTED = 8
} tbc_status;

/* ----------------------------------------------------------------- history
 *
 * Event level: one row per Jepsen history op map, exactly the columns the
 * reference's own checkers read -- :type :f :value :process (:index is the row
 * number).  Shapes: set_full.clj:29-31,42-45,113-116,128-134;
 * tests/ledger.clj:89-114; README.md:41-50,67-74.  Rows whose :process is not
 * an integer (:nemesis ...) must be dropped by the caller, as the reference
 * itself does with (int? process) at tests/ledger.clj:94,204,228.
 */
enum { TBC_INVOKE = 0, TBC_OK_ = 1, TBC_FAIL = 2, TBC_INFO = 3 };

/* op function codes (the :f of an op), shared by every model */
enum {
  TBC_F_READ = 0,     /* register family: a = value read or TBC_NIL             */
  TBC_F_WRITE = 1,    /* register family: a = value written                     */
  TBC_F_CAS = 2,      /* cas-register: a = expected, b = new                    */
  TBC_F_ACQUIRE = 3,  /* mutex                                                  */
  TBC_F_RELEASE = 4,  /* mutex                                                  */
  TBC_F_ADD = 5,      /* set: a = element                                       */
  TBC_F_TXN = 6,      /* multi-register: a = pool offset, b = #micro-ops        */
  TBC_F_TRANSFER = 7, /* bank: a = pool offset of {debit, credit, amount}       */
  TBC_F_CLASS = 8     /* table model: a = op class id (column of the table)     */
  /* TBC_F_READ on set/bank: a = pool offset, b = #values (or TBC_NIL in a)     */
};

#define TBC_NIL INT32_MIN             /* Clojure nil in a value column           */
#define TBC_POS_CRASHED 0xFFFFFFFFu   /* ret_pos of an op that never completed   */

typedef struct tbc_events {
  uint32_t n;               /* number of rows                                   */
  const uint8_t* type;      /* TBC_INVOKE / TBC_OK_ / TBC_FAIL / TBC_INFO       */
  const int32_t* process;   /* integer :process                                 */
  const uint8_t* f;         /* TBC_F_*                                          */
  const int32_t* a;         /* first value column                               */
  const int32_t* b;         /* second value column                              */
} tbc_events;

/*
 * Op level: the history after knossos.history/complete, /without-failures and
 * pairing (recalled: knossos.history complete, without-failures, pair-index;
 * reference uses of the same helpers: tests/ledger.clj:206,239,
 * checker/perf.clj:617,623).  One row per invocation that may have taken
 * effect, sorted by invocation position.  SoA, caller-owned, read-only.
 */
typedef struct tbc_ops {
  uint32_t n;               /* number of ops                                    */
  uint32_t n_events;        /* every position below is < n_events               */
  const uint8_t* f;         /* TBC_F_*                                          */
  const int32_t* a;         /* value column 0 (completed value for reads)       */
  const int32_t* b;         /* value column 1                                   */
  const int32_t* process;   /* dense process id, 0 <= process < n_process       */
  const uint32_t* inv_pos;  /* history position of the invocation, ascending    */
  const uint32_t* ret_pos;  /* position of the :ok completion, TBC_POS_CRASHED  */
  const int32_t* pool;      /* value pool for wide ops (may be NULL)            */
  uint32_t pool_len;
  uint32_t n_process;       /* number of distinct processes (= window slots)    */
} tbc_ops;

/*
 * tbc_pair_events: knossos.history/complete + without-failures + pairing.
 *   - an :ok completion's value columns replace its invocation's (a read learns
 *     what it read), a :fail completion deletes the pair, an :info completion
 *     or a missing one leaves the invocation open for ever (TBC_POS_CRASHED);
 *   - process ids are re-numbered densely in order of first appearance.
 * Output arrays are caller-allocated with capacity ev->n; *n_ops, *n_process
 * receive the counts.  ev_index[i] = row of op i's invocation (its :index).
 * Pure host code, no device needed.
 */
tbc_status tbc_pair_events(const tbc_events* ev,
                           uint8_t* f, int32_t* a, int32_t* b, int32_t* process,
                           uint32_t* inv_pos, uint32_t* ret_pos,
                           uint32_t* n_ops, uint32_t* n_process);

/* ------------------------------------------------------------------ models
 *
 * knossos.model constructors (recalled): register,
 * cas-register, mutex, multi-register, set.  Knossos ships no bank model; the
 * one here follows the reference's own ledger->bank mapping
 * (tests/ledger.clj:89-114; balance = credits - debits, :102-103).
 * TBC_MODEL_TABLE is knossos.model.memo/memo: a dense transition table
 * next = table[state * n_classes + class], 0xFFFF = inconsistent.
 *
 * Value pool (tbc_ops.pool, int32) layouts, written by the caller's encoder
 * (jepsen-tigerbeetle_amd/knossos/_analysis.py is the tested one):
 *   MULTI_REGISTER  :txn op: a = offset, b = #micro-ops of {f (0 read / 1 write),
 *                   key (0..n_keys-1, n_keys <= 8), value (0..13 or TBC_NIL for a
 *                   read)} triples.  State = 4 bits per key; tbc_model.init = packed
 *                   initial state (0 = all nil).
 *   SET             commutative, state-free.  tbc_model.init (or
 *                   tbc_batch_desc.model_aux[h]) = offset of nadds_before[0..R]
 *                   (R = number of :ok completions of the history, in row order).
 *                   :add op: a = index j of the add in completion order (crashed adds
 *                   last); :read op: a = TBC_NIL or offset of {nR, lead, bitset words}
 *                   (bitset over adds in j order; nR = |R| or -1 if R holds an element
 *                   nobody adds; lead = number of leading ones).  Elements unique.
 *   BANK            commutative (negative balances allowed), state-free.  n_keys =
 *                   #accounts (<= 16); init / model_aux = offset of
 *                   bal_before[(R+1) * n_keys]; :transfer a = offset of {debit index,
 *                   credit index, amount}; :read a = TBC_NIL or offset of the balances.
 * SET and BANK exist in the wide schedule only (search_width >= 2 is forced).
 */
enum {
  TBC_MODEL_REGISTER = 0,
  TBC_MODEL_CAS_REGISTER = 1,
  TBC_MODEL_MUTEX = 2,
  TBC_MODEL_TABLE = 3,
  TBC_MODEL_MULTI_REGISTER = 4,
  TBC_MODEL_SET = 5,
  TBC_MODEL_BANK = 6
};

#define TBC_TABLE_INCONSISTENT 0xFFFFu

typedef struct tbc_model {
  uint32_t kind;            /* TBC_MODEL_*                                      */
  int32_t init;             /* initial value (TBC_NIL for (cas-register))       */
  const uint16_t* table;    /* TBC_MODEL_TABLE only                             */
  uint32_t n_states;        /* TBC_MODEL_TABLE only                             */
  uint32_t n_classes;       /* TBC_MODEL_TABLE only                             */
  uint32_t n_keys;          /* multi-register keys / bank accounts              */
  uint32_t flags;           /* TBC_MODEL_F_*                                    */
} tbc_model;

enum { TBC_MODEL_F_NO_NEGATIVE = 1u }; /* bank: balances may not go negative    */

/* ----------------------------------------------------------------- options */
enum { TBC_ALG_COMPETITION = 0, TBC_ALG_WGL = 1, TBC_ALG_LINEAR = 2 };

typedef struct tbc_opts {
  uint32_t algorithm;        /* TBC_ALG_*; jepsen.checker/linearizable :algorithm */
  uint32_t device;           /* HIP device ordinal                               */
  uint64_t time_limit_ms;    /* 0 = none; exceeded => :valid? :unknown           */
  uint64_t max_steps;        /* 0 = none; search-step budget (deterministic)     */
  uint64_t max_visited_bytes;/* 0 = library default; visited-set cap per history */
  uint32_t want_witness;     /* copy the linearization order back                */
  uint32_t visited_per_op;   /* 0 = default (64): first visited-set capacity is   */
                             /* this many entries per op, x16 on each overflow    */
  uint32_t search_width;     /* configs expanded per iteration.  1 = the sequential*/
                             /* knossos.wgl order (witness and counters equal the */
                             /* published algorithm's); 2..16 = the wide schedule */
                             /* (same verdict / failing op, one (config, call)    */
                             /* pair per lane).  0 = default: 1 for TBC_ALG_WGL;  */
                             /* else 4, or 2 when the batch is a register /       */
                             /* cas-register one under the dominance rules with   */
                             /* at most 10 calls in flight on average (fewer      */
                             /* configs expanded in vain; tbc_batch_search_width) */
  uint32_t round_budget;     /* wide schedule, 0 = off: a history that has used     */
                             /* more rounds than this continues at search_width 16  */
                             /* (a batch's stragglers then need far fewer dependent */
                             /* rounds; the schedule stays deterministic)           */
  uint32_t lookahead;        /* wide schedule, register / cas-register: a new config  */
                             /* is SET ASIDE when one of the next 8 completions can   */
                             /* never be linearized from it (its call needs a value   */
                             /* that neither the state nor any call still to be       */
                             /* linearized provides); set-aside configs are expanded  */
                             /* only if the search would otherwise end invalid, so the*/
                             /* failing op and :configs stay exact.  0 = on, 1 = off  */
  uint32_t dominance;        /* wide schedule / level sweep, register / cas-register:  */
                             /* two rules that drop configs without changing a verdict */
                             /* or a failing op (TBC_DOM_*); 0 = both on               */
  uint32_t lanes_per_history;/* depth-first search, register / cas-register / mutex:    */
                             /* 4, 8, 16 or 32 = SEVERAL HISTORIES PER WAVEFRONT (64 / n  */
                             /* them, n lanes each; wgl_narrow.hip): one config per      */
                             /* iteration, n (config, call) pairs per round -- the       */
                             /* schedule for big batches at low concurrency, where a     */
                             /* round has few pairs.  64 = one history per wavefront     */
                             /* (search_width configs per round).  0 = the library       */
                             /* chooses: 8 for a batch of >= 24576 register-family       */
                             /* histories under both dominance rules with at most 10     */
                             /* calls in flight, if search_width is 0 too; else 64.      */
                             /* tbc_batch_lanes_per_history() says what was chosen.      */
  uint32_t list_order;       /* depth-first search, register / cas-register: the order   */
                             /* in which a config's open calls are tried (TBC_ORDER_*).   */
                             /* Verdict, failing op and :previous-ok do not depend on it; */
                             /* counters, witness and :configs are the chosen schedule's. */
                             /* 0 = the library chooses: in order of completion with a    */
                             /* :write as if it completed 24 ranks later (16 + 24) where  */
                             /* it applies (one mask word, no level sweep beside the      */
                             /* search, no count form, no round budget), else slot order. */
                             /* tbc_batch_list_order() says what a batch runs.            */
} tbc_opts;

/* tbc_opts.list_order.  The search takes a config's candidates last to first and pops the last child first, so the order of a
 * front's list of open calls decides which linearization is tried first.  SLOT: by process slot (rounds 1-4; what oracle/wgl_beam.c
 * runs unless told otherwise).  COMPLETION: the call that completes soonest first.  WRITES_LAST: ... and every :write after everything
 * else (a :cas the state allows now before a :write, which every state allows).  16 + W: in order of completion, a :write placed as
 * if it completed W ranks later (W <= 4096) -- the soft form; W = 24 halves the rounds at 19 calls in flight. */
enum { TBC_ORDER_DEFAULT = 0u, TBC_ORDER_SLOT = 1u, TBC_ORDER_COMPLETION = 2u, TBC_ORDER_WRITES_LAST = 3u, TBC_ORDER_WRITE_DELAY = 16u };

/* tbc_opts.dominance bits (set = rule OFF).  Eager reads: an open read whose value is nil or
 * the current state is linearized at once (it changes nothing, so every later schedule stays
 * possible); the search then branches over :write / :cas only.  Twin rule: of several open,
 * not yet linearized calls with the same effect the one completing first goes first.  Both
 * need register values in 0..30; a history outside that range is searched without them. */
enum { TBC_DOM_NO_EAGER_READS = 1u, TBC_DOM_NO_TWIN_RULE = 2u,
       /* COUNT FORM (knossos.competition on a register / cas-register history with crashed calls, both rules above on): crashed
        * calls of one effect are linearized in invocation order, so a config keeps a COUNT per effect class instead of a mask bit
        * per crashed call, process slots are re-used, a crashed call is only linearized right before a call that observes its
        * value, and a config that has used no fewer crashed calls of any class than a visited one with the same (front, mask,
        * state) is dropped.  A history the exact search does not decide within 32 probes per op OF THE BATCH'S LONGEST HISTORY (one
        * budget per launch) -- and only when the caller names no max_steps: a caller's own step limit is one exact pass under that
        * limit, nothing after it -- is first refuted with every
        * class an unlimited supply (a superset of the linearizations), then the prefix before the refuted completion is
        * linearized: verdict and failing op are exact, :configs of such a verdict holds the ONE config the prefix's linearization
        * ended in (the exact search's own exhaustion, when it names an earlier completion, reports its stuck configs as ever).
        * Set = keep one mask bit per crashed call (the published form). */
       TBC_DOM_NO_COUNT_FORM = 4u,
       /* LAZY RULE of the commutative models (set, bank): an :add / :transfer changes nothing a call other than a :read can see and
        * commutes with its kind, so without loss of generality it is linearized only when it completes at the front or when an open,
        * not yet linearized read could take it (set: a read whose value contains the element -- a crashed add no read ever contains
        * is never linearized; bank: a read with a value).  The config space is then no longer 2^(open or crashed mutating calls):
        * a 10k-op set history with 98 crashed adds needs 1.9 * 10^4 probes where the plain search gives up past 2 * 10^7.  Set = off. */
       TBC_DOM_NO_LAZY_COMMUTING = 8u,
       /* STALL HANDOVER -- the one bit of this word that switches something ON (a big quiet batch of register / cas-register histories,
        * several to a wavefront, nobody asking for a witness or naming a step limit): a history whose search has not passed a completion
        * for 4,096 rounds -- it is not linearizable, or in a burst of concurrency -- is stopped and checked again by the level sweep
        * (a few through tbc_check, many as a small batch).  For a batch that holds NOT LINEARIZABLE histories: 342 of 2,048 invalid, search
        * 159 -> 30 ms.  Off by default, measured: valid histories stall too (~10 of 32,768 bench histories at this threshold), they are
        * the ones with the biggest bursts, which the sweep is slowest at, and an all-valid batch pays 54 ms a pass for them while one bad
        * read in the middle of one history costs a pass 104 ms without the handover and 53 with it.  Verdict and failing op are the same
        * either way; the counters of a handed-over history are the stopped search's plus the sweep's, tbc_result.analyzer says who answered. */
       TBC_DOM_STALL_HANDOVER = 16u,
       /* ORDER RESTARTS (round 6; on by default where they apply: the wide depth-first search of a register / cas-register batch, one
        * history per wavefront, the library's own list order, no step limit named).  At high concurrency the cost of a depth-first
        * search is heavy-tailed in the ORDER its candidates are tried and nearly independent between orders (oracle counts, 24 histories
        * at ~32 calls in flight, seven orders: the default order's slowest > 2 * 10^6 probes, its median 192k; the best order per
        * history: slowest 341k, median 91k) -- and a pass over a batch is as long as its slowest history.  So the first pass runs under a
        * budget of 32 probes per op of the batch's longest history; a history that has not ended by then is searched AGAIN FROM SCRATCH
        * in other list orders (16 + 48, writes last, completion, 16 + 8, slot).  Nobody asking for a witness: in all six orders AT THE SAME
        * TIME, as small batches of their own on streams of their own -- the first search to decide a history stops the others' searches
        * of it (a race: knossos.competition's idea, between orders instead of algorithms); verdict and failing op are the search's in
        * any order, which order answers and hence the counters can differ from run to run.  With a witness: one order after the other
        * under the same budget, then the default order without one -- deterministic, the counters the sums over the passes, which
        * oracle/wgl.py check_restart_pipeline states pass by pass.  Set = one pass, no budget. */
       TBC_DOM_NO_ORDER_RESTARTS = 32u,
       /* multi-register (knossos.model/multi-register) under the wide schedule (search_width > 1), round 6; both on by default:
        * EAGER TXNS -- an open :txn of micro-reads only, each nil or what the state holds for its key, is linearized at once (it changes
        * nothing, so every later schedule stays possible; such calls are in the witness, not in the search's chain of branching calls).
        * TXN INDEPENDENCE -- txns conflict when one writes a key the other reads or writes, and txns that do not conflict commute.  At a
        * config whose front is the completion of call X, the candidates are the closure of {X} under "conflicts with" among the open calls
        * not yet linearized (persistent-set reduction: in any valid continuation the first member of that closure can be moved to the
        * front).  Same verdicts and failing ops; fewer configs (oracle/wgl_beam.c: 20k-op histories of 256 processes at 7.7 calls in
        * flight, > 3 * 10^7 probes plain, 4 - 7 * 10^6 under both). */
       TBC_DOM_NO_EAGER_TXNS = 64u, TBC_DOM_NO_TXN_INDEPENDENCE = 128u };

/* ------------------------------------------------------------------ result */
enum { TBC_VALID = 1, TBC_INVALID = 0, TBC_UNKNOWN = -1 };
enum {
  TBC_CAUSE_NONE = 0,
  TBC_CAUSE_TIME_LIMIT = 1,
  TBC_CAUSE_STEP_LIMIT = 2,
  TBC_CAUSE_VISITED_FULL = 3   /* visited set hit max_visited_bytes              */
};

#define TBC_MAX_FINAL_CONFIGS 10   /* jepsen.checker/linearizable truncates to 10 */
#define TBC_NO_OP 0xFFFFFFFFu

typedef struct tbc_config {      /* one knossos config: {:model :pending :last-op} */
  int32_t state;                 /* model state (value / table state id)          */
  uint32_t last_op;              /* op index linearized last, or TBC_NO_OP (always  */
                                 /* TBC_NO_OP from a several-histories-per-wavefront */
                                 /* batch that wants no witness: it keeps no links)  */
  uint32_t n_pending;            /* open ops at the front ...                     */
  uint32_t n_linearized;         /* ... of which this many are already linearized */
  uint32_t pending[16];          /* first 16 open op indices                      */
  uint32_t linearized_mask;      /* bit i: pending[i] is linearized               */
} tbc_config;

typedef struct tbc_counters {
  uint64_t steps;        /* candidate linearizations attempted (model step ok)   */
  uint64_t visited;      /* configs inserted into the visited set                */
  uint64_t probes;       /* visited-set lookups                                  */
  uint64_t backtracks;   /* frames popped                                        */
  uint64_t max_depth;    /* deepest DFS stack                                    */
  uint64_t table_slots;  /* final visited-set capacity (entries)                 */
  uint64_t ns_pack;      /* device time: history pack kernel                     */
  uint64_t ns_search;    /* device time: search kernel(s)                        */
  uint64_t ns_total;     /* wall time inside the call                            */
} tbc_counters;

typedef struct tbc_result {
  int32_t valid;             /* TBC_VALID / TBC_INVALID / TBC_UNKNOWN            */
  int32_t cause;             /* TBC_CAUSE_* when unknown                         */
  uint32_t analyzer;         /* TBC_ALG_WGL or TBC_ALG_LINEAR: who answered      */
  uint32_t fail_op;          /* invalid: op whose completion cannot be passed    */
  uint32_t prev_ok_op;       /* invalid: op completing just before it (:previous-ok) */
  int32_t final_state;       /* valid: model state after the witness             */
  uint32_t n_witness;        /* valid && want_witness: ops in `witness`, else 0  */
  uint32_t search_width;     /* configs per round of the depth-first search of   */
                             /* this call (what tbc_opts.search_width 0 became)  */
  uint32_t* witness;         /* valid && want_witness: op indices, library-owned */
  uint32_t n_configs;        /* final configs (<= TBC_MAX_FINAL_CONFIGS)         */
  tbc_config configs[TBC_MAX_FINAL_CONFIGS];
  tbc_counters counters;
} tbc_result;

/* --------------------------------------------------------- single history
 *
 * tbc_check == (knossos.wgl/analysis model history) et al.: host SoA in,
 * verdict out (H2D + pack kernel + search kernel + D2H).  `out` is caller
 * allocated; out->witness is library-allocated and released by
 * tbc_result_free.
 */
tbc_status tbc_check(const tbc_ops* ops, const tbc_model* model,
                     const tbc_opts* opts, tbc_result* out);
void tbc_result_free(tbc_result* r);

/* ------------------------------------------------------------- batch mode
 *
 * Many independent histories per launch -- what jepsen.independent/checker
 * does per key (set_full.clj:155) and BASELINE.json's batch configs ask for.
 * A batch owns device memory for its inputs, packed layout, stacks and
 * visited sets; inputs stay resident in HBM across tbc_batch_run calls.
 * Histories are concatenated: history h owns ops [op_off[h], op_off[h+1]).
 */
typedef struct tbc_batch tbc_batch;

typedef struct tbc_batch_desc {
  uint32_t n_hist;
  const uint64_t* op_off;     /* n_hist+1 offsets into the op columns            */
  const uint32_t* n_events;   /* per history                                     */
  const uint32_t* n_process;  /* per history                                     */
  tbc_ops cols;               /* concatenated columns; cols.n = total ops        */
  const int32_t* model_aux;   /* optional, per history: overrides tbc_model.init  */
                              /* (set / bank: pool offset of the history's own    */
                              /* per-front table when histories share one pool)   */
} tbc_batch_desc;

tbc_status tbc_batch_create(const tbc_batch_desc* desc, const tbc_model* model,
                            const tbc_opts* opts, tbc_batch** out);
/* one pass of the hot path over the resident batch: pack + search (+ retries
 * of histories whose visited set overflowed).  results: n_hist entries,
 * caller-allocated (witness pointers are library-owned, valid until the next
 * run or tbc_batch_destroy). */
tbc_status tbc_batch_run(tbc_batch* b, tbc_result* results);
/* device-time breakdown of the last run, ns, measured with HIP events on the
 * batch's own stream: [0] memset/init, [1] pack, [2] search, [3] retries */
tbc_status tbc_batch_last_timing(const tbc_batch* b, uint64_t ns[4]);
/* SEVERAL BATCHES IN FLIGHT.  A batch owns its stream and its arenas, so different batches may be run at the
 * same time from different host threads (tbc_batch_run holds no process-wide lock; one batch is still one
 * thread's at a time) -- which is how jepsen.independent/checker drives a checker: one (bounded-)pmap thread
 * per key group (the reference wraps its per-key checkers in jepsen.independent/checker at
 * src/tigerbeetle/workloads/set_full.clj:154-158).  Batches that run several histories per wavefront take the whole GPU for their search and the
 * library gives those searches the device one at a time, in launch order, through a device-side event; the
 * init and pack of the next batch run beside the search of the previous one (measured: 2 x 32,768 histories
 * in flight, 258k histories/s against 204k one batch after the other).  ns = how long the last run's search
 * waited for its turn (not counted in ns[2] of tbc_batch_last_timing). */
tbc_status tbc_batch_last_turn_wait(const tbc_batch* b, uint64_t* ns);
/* sum of tbc_counters over the last run (probes, visited ...) */
tbc_status tbc_batch_last_counters(const tbc_batch* b, tbc_counters* out);
uint64_t tbc_batch_device_bytes(const tbc_batch* b);
/* the search_width this batch runs the depth-first search at (what tbc_opts.search_width = 0 resolved to; 1 when
 * several histories share a wavefront) */
uint32_t tbc_batch_search_width(const tbc_batch* b);
/* lanes per history of the depth-first search: 4 / 8 / 16 / 32 (several histories per wavefront) or 64 */
uint32_t tbc_batch_lanes_per_history(const tbc_batch* b);
/* the order of the fronts' lists this batch's depth-first search runs over: TBC_ORDER_SLOT / _COMPLETION / _WRITES_LAST or 16 + W
 * (what tbc_opts.list_order = 0 resolved to; SLOT wherever another order does not apply) */
uint32_t tbc_batch_list_order(const tbc_batch* b);
/* histories of the last run whose first pass ended at its budget and that were then searched in several list orders at once
 * (tbc_opts.dominance, TBC_DOM_NO_ORDER_RESTARTS) */
uint32_t tbc_batch_last_raced(const tbc_batch* b);
/* PROGRESS of a run that is out (reference: knossos.search reports on a search while it runs -- knossos/src/knossos/search.clj, the
 * reporter its `run` starts; jepsen.checker/linearizable logs it).  Callable from ANOTHER thread while tbc_batch_run is in flight on
 * the batch (the one entry point that is; nothing else may touch a batch that is running): how many of the batch's histories have a
 * verdict so far -- the search kernels count them into host memory as they store their results --, the phase, the time since the
 * run began.  After the run: what it handed back (n_decided = the histories whose verdict is not TBC_UNKNOWN), running = 0. */
enum { TBC_PHASE_IDLE = 0, TBC_PHASE_PACK = 1 /* packing and the first search pass (queued together) */, TBC_PHASE_RETRIES = 2 /* verdicts
       composed; histories that overflowed, stalled or hit a budget are searched again */ };
typedef struct tbc_progress {
  uint32_t n_histories, n_decided, phase, running;
  uint64_t elapsed_ns;
} tbc_progress;
tbc_status tbc_batch_progress(const tbc_batch* b, tbc_progress* out);
/* how the last run was answered when the level sweep (knossos.linear; jit_sweep.hip) is in play:
 * TBC_ALG_LINEAR always asks for it, TBC_ALG_COMPETITION on small batches that want no witness.
 * The sweep cuts a history into segments of about seg_target completions at fronts with at most
 * cut_open open calls and sweeps them concurrently; histories it cannot finish (a config set
 * outgrows on-chip memory, > 64 process slots, set / bank) are answered by the depth-first
 * search instead -- tbc_result.analyzer says which (TBC_ALG_LINEAR = the sweep). */
typedef struct tbc_sweep_info {
  uint32_t enabled;       /* this batch runs the sweep first                              */
  uint32_t seg_target;    /* wanted segment length in completions (0 = one segment)       */
  uint32_t max_segs;      /* segments per history at most                                 */
  uint32_t cut_open;      /* a cut needs at most this many open calls                     */
  uint32_t n_dom;         /* states a segment may start in: nil + 0..greatest value of the batch */
  uint32_t n_segments;    /* last run: segments swept                                     */
  uint32_t n_fallback;    /* last run: histories handed to the depth-first search         */
} tbc_sweep_info;
tbc_status tbc_batch_sweep_info(const tbc_batch* b, tbc_sweep_info* out);

/* ---------------------------------------------- one history over several GPUs
 *
 * BASELINE.json's north_star asks for the search frontier of ONE history to be sharded over the
 * GPUs of a node.  With the level sweep that is a property of the algorithm: a history is a chain
 * of segments, every (segment, 32-origin slice) is swept by its own wavefront and hands on a
 * relation {origin -> origin ids of the next segment}; WHERE a wavefront runs does not matter.
 * So rank r of w sweeps the wavefronts with (segment * 4 + slice) % w == r
 * (tbc_batch_set_shard + tbc_batch_sweep_partial), the ranks exchange their relation tables --
 * ONE all-gather of tbc_batch_sweep_table() bytes over RCCL (jepsen-tigerbeetle_amd/shard.py),
 * entries a rank does not own are all zero, so the merged table is the bitwise OR -- and every rank
 * composes the merged table (tbc_batch_sweep_finish).  The inputs are replicated (a history is a
 * few hundred KB); pack runs on every rank.  A wavefront the sweep cannot finish (status 2) makes
 * the finishing rank fall back to its own depth-first search, as in the single-GPU path.
 */
#define TBC_SWEEP_SLICES 4u
typedef struct tbc_sweep_rel {   /* one per (history, segment, slice) */
  uint32_t status;               /* 0 = no such wavefront (or not this rank's), 1 = swept, 2 = overflow */
  uint32_t F0, F1;               /* levels swept: [F0, F1)                                         */
  uint32_t n_org;                /* origins of this slice that are configs                         */
  uint32_t max_level;            /* largest level                                                  */
  uint32_t subrounds;
  uint32_t n_end;                /* configs at front F1                                            */
  uint32_t end_state;            /* state of the first of them (non-register models)               */
  uint64_t configs_total;        /* sum of the level sizes                                         */
  uint64_t probes;               /* expansions                                                     */
  uint32_t M[32][TBC_SWEEP_SLICES]; /* M[o] = origin ids of the NEXT segment reachable from origin  */
                                 /* 32 * slice + o (128-bit set); last segment: word 0 = final states */
  uint32_t last_level[32];       /* greatest level at which a config reachable from origin o existed */
} tbc_sweep_rel;

typedef struct tbc_sweep_verdict {
  int32_t valid;                 /* TBC_VALID / TBC_INVALID / TBC_UNKNOWN (a wavefront missing or overflowed) */
  uint32_t fail_level;           /* invalid: rank of the completion nobody passes                  */
  uint32_t fail_seg;             /* invalid: the segment it lies in                                */
  uint32_t live_in[TBC_SWEEP_SLICES];  /* invalid: origin ids of that segment reachable from the start */
  uint32_t final_bits;           /* valid: final states reached (register family: nil = bit 0, v = bit v + 1) */
  uint32_t end_state;            /* valid, other models: the state                                 */
  uint32_t n_wavefronts;         /* relations composed                                             */
  uint64_t probes, configs_total, subrounds, max_level;
} tbc_sweep_verdict;

/* compose the relations of ONE history (max_segs * TBC_SWEEP_SLICES records, segment-major) in order.
 * Pure host code, no device needed. */
tbc_status tbc_sweep_compose(const tbc_sweep_rel* rel, uint32_t max_segs, uint32_t n_completions,
                             tbc_sweep_verdict* out);

tbc_status tbc_batch_set_shard(tbc_batch* b, uint32_t rank, uint32_t world);
/* pack + sweep of this rank's wavefronts; no verdicts yet */
tbc_status tbc_batch_sweep_partial(tbc_batch* b);
/* the relation table of the last partial run: DEVICE pointer (for a collective straight out of HBM) and size */
tbc_status tbc_batch_sweep_table(const tbc_batch* b, void** device_ptr, uint64_t* bytes);
/* verdicts from a merged relation table in HOST memory; merged_bytes must equal tbc_batch_sweep_table()'s size
 * (anything else is TBC_ERR_INVALID_ARG, as is a finish that no tbc_batch_sweep_partial precedes) */
tbc_status tbc_batch_sweep_finish(tbc_batch* b, const void* merged, uint64_t merged_bytes, tbc_result* results);
/* the same from the ranks' tables still in DEVICE memory, `world` of them back to back (what an all-gather over RCCL leaves):
 * they are OR-ed on the device into this batch's table, then composed -- the exchanged bytes never pass through the host */
tbc_status tbc_batch_sweep_merge(tbc_batch* b, const void* gathered_device, uint64_t gathered_bytes, uint32_t world, tbc_result* results);
/* ---- the exchange behind the C-ABI (round 6; csrc/tbc_comm.hip).  The host the reference is -- a Clojure process binding this library
 * through JNA (/root/reference/project.clj:6-8) -- has no torch.distributed; with these a rank of ANY host language runs its share
 * of a sharded check:
 *   tbc_comm_unique_id         rank 0: 128 bytes that reach the other ranks however the host likes (they are an ncclUniqueId)
 *   tbc_comm_init              an RCCL communicator over xGMI for this rank (librccl.so is loaded here, at the first call)
 *   tbc_comm_init_host         ... or a transport of the caller's: an all-gather over HOST memory (MPI, a socket, a test's gloo)
 *   tbc_batch_sweep_allgather  tbc_batch_set_shard + tbc_batch_sweep_partial + ONE all-gather of the relation tables (RCCL: straight
 *                              out of HBM, OR-merged on the device; host transport: through tbc_batch_sweep_finish) + composition.
 *                              Every rank calls it with a batch created from the same histories and options; every rank gets the results.
 */
typedef struct tbc_comm tbc_comm;
#define TBC_COMM_ID_BYTES 128
/* gather `bytes` from every rank: recv holds world * bytes, rank r's contribution at r * bytes.  0 = done. */
typedef int (*tbc_allgather_fn)(void* user, const void* send, void* recv, uint64_t bytes);
tbc_status tbc_comm_unique_id(void* id /* TBC_COMM_ID_BYTES */);
tbc_status tbc_comm_init(uint32_t rank, uint32_t world, const void* id, uint32_t device, tbc_comm** out);
tbc_status tbc_comm_init_host(uint32_t rank, uint32_t world, tbc_allgather_fn fn, void* user, tbc_comm** out);
uint32_t tbc_comm_rank(const tbc_comm* c);
uint32_t tbc_comm_world(const tbc_comm* c);
void tbc_comm_destroy(tbc_comm* c);
tbc_status tbc_batch_sweep_allgather(tbc_batch* b, tbc_comm* c, tbc_result* results);
void tbc_batch_destroy(tbc_batch* b);

/* ------------------------------------------------ streaming: the same batch, fresh histories
 *
 * The reference calls a checker ONCE per history: checker/compose over the test's one history
 * (/root/reference/src/tigerbeetle/core.clj:139-146), jepsen.independent/checker once per key
 * (/root/reference/src/tigerbeetle/workloads/set_full.clj:155-158).  A caller that checks many histories
 * never checks one twice, and creating a batch per input pays for 85 GB of allocation, the op columns
 * over PCIe from pageable memory and a sizing pass every time (round 5: 12-22k histories/s against 321k
 * resident).  These entry points keep a batch -- arenas, streams, the decisions tbc_batch_create made --
 * and change its histories:
 *
 *   tbc_batch_map_input     a slot of library-owned PINNED host memory for the caller to fill IN PLACE (a JNA
 *                           Memory / direct ByteBuffer / numpy view over it): no copy inside the library
 *   tbc_batch_submit_input  queues the slot's copy to the device on the batch's copy stream and returns; it runs
 *                           under whatever the batch (or another batch) is computing.  Up to two inputs may wait.
 *   tbc_batch_run           consumes the oldest waiting input, if any (else the resident input once more): waits
 *                           for its copy, unpacks it on the device, packs, searches.  results: n_hist of THAT input.
 *   tbc_batch_reload        convenience: the six columns of a tbc_batch_desc -> wire format -> slot 0 / 1 -> submitted
 *
 * WIRE FORMAT, 12 B an op (21 B in tbc_ops): word = f | a << 4 | b << 12 | process << 20 (f 4 bits; a, b 8 bits:
 * 0..254 or TBC_WIRE_NIL; process 12 bits), inv_pos, ret_pos as in tbc_ops.  Register / cas-register / mutex.
 *
 * What a fresh input must share with the input the batch was created from (else TBC_ERR_UNSUPPORTED from the call
 * that finds out -- tbc_batch_submit_input or the tbc_batch_run that consumes it -- and the caller destroys and
 * creates): at most the first input's number of histories and an eighth more than its ops; process slots within the
 * batch's mask words; register values within the batch's value domain (the greatest value of the first input,
 * when the dominance rules are on); no crashed call if the first input had none.  A COUNT-FORM batch (created from
 * histories with crashed calls that have an effect, under the default rules: what a nemesis makes) takes fresh inputs
 * too: the classes of crashed calls and the re-used process slots are planned on the host inside
 * tbc_batch_submit_input (a few host threads; the caller's words are not modified), and an input whose crashed calls
 * do not fit the form (> 128 bits of counts) is refused.  Batches of the level sweep (incl. a count-form batch of
 * <= 8 histories, which has the relaxed sweep beside it), of set / bank / multi-register / table models and the
 * sequential schedule take no fresh inputs.
 * Verdict, failing op and every counter of a consumed input equal those of a batch created from the same histories
 * (tests/test_stream_gpu.py).
 */
#define TBC_WIRE_NIL 0xFFu
#define TBC_WIRE_WORD(f, a8, b8, process) ((uint32_t)(f) | ((uint32_t)(a8) << 4) | ((uint32_t)(b8) << 12) | ((uint32_t)(process) << 20))

typedef struct tbc_batch_input {   /* pointers into one pinned slot; valid until the batch is destroyed */
  uint32_t n_hist_cap;            /* histories the slot holds at most                              */
  uint32_t reserved0;
  uint64_t ops_cap;               /* ops the slot holds at most                                    */
  uint64_t* op_off;               /* [n_hist_cap + 1] as tbc_batch_desc.op_off                     */
  uint32_t* n_events;             /* [n_hist_cap]                                                  */
  uint32_t* n_process;            /* [n_hist_cap]                                                  */
  uint32_t* word;                 /* [ops_cap] TBC_WIRE_WORD(f, a, b, process)                     */
  uint32_t* inv_pos;              /* [ops_cap]                                                     */
  uint32_t* ret_pos;              /* [ops_cap] TBC_POS_CRASHED = never completed                   */
} tbc_batch_input;

/* slot < 16; allocated when first mapped.  Waits until the slot's previous input has left for the device. */
tbc_status tbc_batch_map_input(tbc_batch* b, uint32_t slot, tbc_batch_input* out);
/* the slot holds n_hist histories (op_off[0..n_hist], n_events, n_process, the wire columns): copy them, asynchronously */
tbc_status tbc_batch_submit_input(tbc_batch* b, uint32_t slot, uint32_t n_hist);
tbc_status tbc_batch_reload(tbc_batch* b, const tbc_batch_desc* desc);

typedef struct tbc_input_info {
  uint32_t n_hist;                /* histories of the resident input (what tbc_batch_run's results hold) */
  uint32_t pending;               /* submitted inputs not yet consumed                                  */
  uint64_t total_ops;             /* ops of the resident input                                          */
  uint64_t bytes_copied;          /* last consumed input: bytes that crossed PCIe ...                   */
  uint64_t ns_copy;               /* ... and how long the copy took (HIP events on the copy stream)     */
  uint64_t inputs_consumed;
  uint32_t lists_regrown;         /* times an input's per-front lists outgrew their arena               */
  uint32_t n_hist_cap;
  uint64_t ops_cap;
} tbc_input_info;
tbc_status tbc_batch_input_info(const tbc_batch* b, tbc_input_info* out);

/* ----------------------------------------------------------------- memo
 *
 * knossos.model.memo/memo for a caller-defined model: given the closure
 * step(state, class) over n_classes op classes, enumerate reachable states
 * breadth-first from state 0 and fill a dense table.  `step` returns the next
 * state id handle or -1 for inconsistent; ids are opaque int64 handles chosen
 * by the caller (for example a packed value).  Host only.
 */
typedef int64_t (*tbc_step_fn)(int64_t state, uint32_t op_class, void* user);
tbc_status tbc_memo_build(int64_t init_state, uint32_t n_classes,
                          tbc_step_fn step, void* user, uint32_t max_states,
                          uint16_t* table /* max_states*n_classes */,
                          int64_t* state_handles /* max_states */,
                          uint32_t* n_states);

/* ------------------------------------------------------- checker/set-full
 *
 * jepsen.checker/set-full -- the checker the reference actually runs for its set-full workload
 * (/root/reference/src/tigerbeetle/workloads/set_full.clj:157, `(checker/set-full {:linearizable? true})`;
 * SURVEY.md section 8f row 3).  Per element of the set three history indices decide everything
 * (recalled from jepsen.checker; restated in oracle/set_full.py):
 *   known         first op that proved the element exists: its add's :ok, or the :ok of a read containing it
 *   last_present  invocation of the latest-invoked :ok read that contained it
 *   last_absent   invocation of the latest-invoked :ok read that did not
 * (only reads completing after the element's add was invoked count).  The device part is the scan of the
 * reads x elements membership matrix that yields those three per element -- a streaming pass, the one
 * kernel on this path with a real HBM roofline; outcomes (:stable / :lost / :never-read, latencies,
 * :valid?) are a few operations per element on the host (jepsen-tigerbeetle_amd/jepsen/set_full.py).
 *
 * Inputs (caller-owned): elements numbered in order of their first :add invocation (add_invoke ascending),
 * reads = the :ok reads in order of invocation (read_invoke ascending); present = n_reads rows of
 * words_per_row 32-bit words, bit e of row r = read r's value contains element e.
 */
typedef struct tbc_setfull_in {
  uint32_t n_elements, n_reads, words_per_row, device;
  const uint32_t* add_invoke;    /* [n_elements] history index of the element's :add invocation  */
  const uint32_t* add_ok;        /* [n_elements] index of that add's :ok, TBC_NO_OP if none      */
  const uint32_t* read_invoke;   /* [n_reads]                                                    */
  const uint32_t* read_ok;       /* [n_reads]                                                    */
  const uint32_t* present;       /* [n_reads * words_per_row]                                    */
} tbc_setfull_in;

typedef struct tbc_setfull_out {   /* arrays caller-allocated, n_elements each; TBC_NO_OP = none */
  uint32_t* known;
  uint32_t* last_present;
  uint32_t* last_absent;
  uint64_t ns_scan;              /* device time of the scan kernel (HIP events)                  */
  uint64_t bytes_scanned;        /* matrix bytes the scan loaded                                 */
  uint64_t bytes_matrix;         /* n_reads * words_per_row * 4                                  */
} tbc_setfull_out;

typedef struct tbc_setfull tbc_setfull;
/* inputs become resident in HBM (H2D here); run scans them; results copied into `out` */
tbc_status tbc_setfull_create(const tbc_setfull_in* in, tbc_setfull** handle);

/* The same with the reads in COMPACT form, the membership matrix built ON THE DEVICE (nothing of size reads x elements exists on
 * the host or crosses PCIe).  A read of a grow-only set is, up to a few exceptions, a PREFIX of the elements in add-invocation
 * order -- what had been added when it ran; so read r is given as top[r] (every element numbered below top[r] is in it) and a
 * list of exceptions exc[exc_off[r] .. exc_off[r + 1]): element numbers, each at most once per read -- a listed element below
 * top[r] is ABSENT from the read, one at or above it PRESENT.  (The reference's reads, workloads/set_full.clj:128-134, return the
 * sorted set: the caller numbers the values by add invocation, takes top = greatest element read + 1 and lists the holes.) */
typedef struct tbc_setfull_rows {
  uint32_t n_elements, n_reads, device, reserved0;
  const uint32_t* add_invoke;    /* as tbc_setfull_in */
  const uint32_t* add_ok;
  const uint32_t* read_invoke;
  const uint32_t* read_ok;
  const uint32_t* top;           /* [n_reads], <= n_elements */
  const uint64_t* exc_off;       /* [n_reads + 1], ascending, exc_off[0] = 0 */
  const uint32_t* exc;           /* [exc_off[n_reads]] element numbers < n_elements, each at most once per read, in any order (a duplicate is TBC_ERR_INVALID_ARG) */
} tbc_setfull_rows;
tbc_status tbc_setfull_create_rows(const tbc_setfull_rows* in, tbc_setfull** handle);
tbc_status tbc_setfull_run(tbc_setfull* handle, tbc_setfull_out* out);
void tbc_setfull_destroy(tbc_setfull* handle);

/* ------------------------------------------------------------------- misc */
uint32_t tbc_version(void);             /* TBC_ABI_VERSION                       */
const char* tbc_strerror(int status);
const char* tbc_last_error(void);       /* thread-local detail of the last error */
int32_t tbc_device_count(void);         /* gfx950 devices visible; 0 if none     */
/* diagnostics: with TBC_DEBUG=1 in the environment the kernels mirror their progress into
 * host-mapped words; this copies up to n (<= 64) of them.  Returns 0 when disabled. */
int tbc_debug_peek(uint32_t* out, uint32_t n);
/* Environment switches, read at launch time -- for experiments and A/B tests, never needed for correct results:
 *   TBC_OPEN_WALK=slots           build the per-front tables with lane = process slot for batches of <= 64
 *                                 slots too (default: lane = front, open_walk_impl.h); same tables either way
 *   TBC_NARROW_WAVES_PER_SIMD=n   wavefronts per SIMD the several-histories-per-wavefront search is launched at
 *                                 (default 4, the build's maximum)
 *   TBC_SWEEP=0|1, TBC_SWEEP_SEG=n  never / whenever possible take the level sweep; its segment length
 *   TBC_SWEEP_WG=0|4|8            wavefronts per segment of a sweep of at most 4,096 wavefronts (default 8: a workgroup per segment,
 *                                 jit_sweep_wg.hip; 0 = one wavefront per segment always)
 *   TBC_SWEEP_WG_RING=1           (experimental) the workgroup sweep gathers a sub-round's children in a ring before inserting them
 *   TBC_SWEEP_WG_FP=1             (experimental) ... keeps 8 bits of a key's hash in its table word (a probe past another key reads no key)
 *   TBC_SWEEP_WG_COMPACT=1|2      (experimental) ... a wide sub-round numbers its children first and inserts 512 CHILDREN a pass (not with the ring);
 *                                 2 = and a pass that fits one wavefront is run by wavefront 0 alone
 *   TBC_NARROW_LEAN=1|2           (experimental; 2 = and the lookahead at once only for the config that is popped next) several histories per wavefront over lean tables: a list entry {call, twin mask} in one array,
 *                                 an 8 B lookahead record (two producer slots; three or more read as "one is still to be linearized")
 *   TBC_NARROW_ORDER=1|2|16+W     (experimental) ... with every front's list in order of completion: the call that completes soonest is tried
 *                                 first (fewer rounds for the same probes); a witness's absorbed reads follow the same order;
 *                                 2 = and the :write calls after everything else (a :cas the state allows now before a :write);
 *                                 16 + W = and a :write as if it completed W ranks later (the soft form of 2; W = 16 .. 24)
 *   TBC_PACK_ONE=1|2              (experimental) a handful of histories are packed by sixteen wavefronts each, tables in LDS (pack_one.hip);
 *                                 2 = and the per-front counts in the same launch
 *   TBC_PACK_WG=1                 (experimental) a batch of the wide schedule is packed by four wavefronts per history, tables in LDS,
 *                                 and the same pass leaves the per-front counts (pack_one.hip, pack_wg_kernel); 2 = a batch that
 *                                 cannot take it is TBC_ERR_UNSUPPORTED instead of silently the default kernels
 *   TBC_DEBUG=1, TBC_SYNC_EACH=1  progress words (above), a traced synchronisation after every launch */

#ifdef __cplusplus
}
#endif
#endif /* TBCHECK_H */
