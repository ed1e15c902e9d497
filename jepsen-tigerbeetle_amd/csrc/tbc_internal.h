// tbc_internal.h -- shared between the host orchestration and the HIP kernels.
#pragma once
#include <cstdarg>
#include <cstdint>
#include "../../include/tbcheck.h"

namespace tbc {

void set_error(const char* fmt, ...);

constexpr uint32_t kInf = 0xFFFFFFFFu;
constexpr uint32_t kMaxSlots = 1024;       // 16 mask words x 64 lanes
constexpr uint32_t kWavesPerBlock = 4;
constexpr uint32_t kBlock = 64 * kWavesPerBlock;
constexpr uint8_t kFNone = 0xFF;           // sentinel record: never a candidate
constexpr uint32_t kBeamMaxTabLog2 = 28;   // wide kernel: the 16 B keys of one history stay below 4 GiB (32-bit byte offsets)
constexpr uint32_t kCfgCap = 256;          // configs at the failing front copied back per invalid history

// One op in the slot-major ("per process, in time order") layout the search
// kernel walks.  32 bytes = two dwordx4 loads per cursor move.
struct __attribute__((aligned(16))) Rec {
  uint32_t inv_rank;   // #completions positioned before this invocation
  uint32_t ret_rank;   // rank of its own completion, kInf if crashed
  uint32_t opidx;      // index in the caller's op columns (= invocation order)
  uint32_t f;          // TBC_F_* or kFNone
  int32_t a, b;
  uint32_t cls;        // what the front walk (pack_open.hip) asks of the call at every front, worked out once: rec_cls()
  uint32_t prod;       // the register value it leaves behind as a lookahead byte (look_prod()), kLookNone if none
};
static_assert(sizeof(Rec) == 32, "Rec must be 32 bytes");

// Per-history descriptor (device resident; written once by the host, n_ret and
// status filled by the pack kernel).
struct __attribute__((aligned(16))) Hist {
  uint64_t op_off;     // first op in the concatenated columns
  uint64_t rec_off;    // Rec units
  uint64_t seg_off;    // u32 units (n_slots + 1 entries)
  uint64_t ret_off;    // u32 units (ret_slot / ret_op, n_ops entries each)
  uint64_t bm_off;     // u32 units (bitmap + word-prefix arenas)
  uint64_t frame_off;  // u32 units (n_ops * frame words)
  uint64_t tab_off;    // u64 units
  uint32_t n_ops;
  uint32_t n_events;
  uint32_t n_slots;
  uint32_t tab_log2;   // visited-set capacity = 1 << tab_log2 entries
  uint32_t n_ret;      // pack: number of completions
  uint32_t status;     // pack: 0 ok, else tbc_status
  int32_t aux;         // commutative models: pool offset of this history's per-front table
  uint32_t flags;      // kHistCount: count form -- the process column holds re-used slots, crashed calls hold none
};
constexpr uint32_t kHistCount = 1u;

struct __attribute__((aligned(8))) DevResult {
  int32_t valid;
  int32_t cause;
  uint32_t max_front;   // greatest front reached (invalid: the completion nobody passes)
  uint32_t depth;       // valid: witness length
  int32_t final_state;
  uint32_t n_configs;
  uint32_t fail_op;     // invalid: op whose completion has rank max_front
  uint32_t prev_ok_op;  // invalid: op completing just before it, or TBC_NO_OP
  uint32_t tab_log2;    // wide schedule: final visited-set capacity (it can grow inside the kernel)
  uint32_t pad;
  uint64_t steps, visited, probes, backtracks, max_depth, bucket_reads;
};

struct PackArgs {
  Hist* hist;
  const uint8_t* f;
  const int32_t* a;
  const int32_t* b;
  const int32_t* process;
  const uint32_t* inv_pos;
  const uint32_t* ret_pos;
  Rec* rec;
  uint32_t* seg;
  uint32_t* ret_slot;
  uint32_t* ret_op;
  uint32_t* bitmap;     // zeroed before launch
  uint32_t* wpre;
  uint32_t* scratch;    // the frames arena doubles as pack scratch
  uint32_t frame_words; // words per op available in scratch (>= 3)
  uint32_t n_hist;      // histories [h0, n_hist) are packed by this launch
  uint32_t model_kind;
  uint32_t n_classes;
  uint32_t h0;
  uint32_t* dbg;
  const int32_t* pool_vals;
  uint32_t pool_len;
  uint32_t n_keys;
};

struct SearchArgs {
  const Hist* hist;
  const Rec* rec;
  const uint32_t* seg;
  const uint32_t* ret_slot;
  const uint32_t* ret_op;
  uint32_t* frames;
  uint64_t* tab;
  DevResult* results;
  uint32_t* witness;        // n_ops entries per history at op_off, may be null
  const uint32_t* work;     // history indices to process
  uint32_t* queue;          // work-queue head, zeroed before launch
  const uint16_t* table;    // TBC_MODEL_TABLE
  uint32_t n_work;
  uint32_t model_kind;
  int32_t init_state;
  uint32_t n_classes;
  uint32_t n_states;
  uint32_t pad;
  uint64_t max_steps;
  uint64_t time_limit_ticks;  // wall_clock64 ticks (100 MHz), 0 = none
  uint32_t* dbg;              // optional host-mapped progress words (TBC_DEBUG=1), else null
  const int32_t* pool_vals;   // wide op values (multi-register micro-ops)
  uint64_t* cfg;              // kCfgCap records of (2 + mask_words) u64 per history: k0, M[], last op
  uint32_t* progress;         // optional, HOST memory: word 0 = histories decided so far, as last published (tbc_batch_progress) ...
  uint32_t* progress_dev;     // ... from this count in HBM
};

// ---- wide ("beam") schedule of the search: extra per-history layout built by pack_open_kernel
// One candidate of one front: the open-call lists hold the whole record, so a lane reaches its
// candidate with ONE load (list entry -> op record would be two dependent trips to HBM).
struct __attribute__((aligned(16))) OpRec {
  uint32_t op;          // index in the caller's op columns
  uint32_t f_slot;      // f | slot << 8 | kAtFront if this front is the call's own completion
  int32_t a, b;
};
static_assert(sizeof(OpRec) == 16, "OpRec must be 16 bytes");
constexpr uint32_t kAtFront = 0x80000000u;

// Lookahead records, one per completion rank t (register / cas-register, wide schedule): what the call
// completing at t needs, what it produces, and who else could produce that value.  (1 + mask_words) u64:
//   word 0: slot (16) | need (8) << 16 | prod (8) << 24 | dinv (8) << 32 | dprod (8) << 40
//           need / prod: register values 0..31, kLookNone otherwise (then the rank constrains nothing here)
//           dinv  = min(t - inv_rank(call), 255): the call is open at front F  <=>  dinv >= t - F
//           dprod = min(t - inv_rank(x), 255) over the other calls x that produce `need` and are invoked
//                   no later than completion t: one of them is invoked after front F  <=>  dprod < t - F
//   words 1..: slots of the other calls open at front t (crashed ones included) that produce `need`
constexpr uint32_t kLookahead = 8;          // completions looked at per new config
constexpr uint32_t kLookNone = 0xFFu;
#ifdef __HIPCC__
// register value a call must find / leaves behind, as a lookahead byte (0..31, else kLookNone)
__host__ __device__ inline uint32_t look_val(int32_t v) { return (v >= 0 && v < 32) ? (uint32_t)v : kLookNone; }
__host__ __device__ inline uint32_t look_need(uint32_t f, int32_t a) {
  return ((f == TBC_F_READ && a != TBC_NIL) || f == TBC_F_CAS) ? look_val(a) : kLookNone;
}
__host__ __device__ inline uint32_t look_prod(uint32_t f, int32_t a, int32_t b) {
  return f == TBC_F_WRITE ? look_val(a) : (f == TBC_F_CAS ? look_val(b) : kLookNone);
}
// Rec.cls: bit 0 = a call at all (not a sentinel), 1 = live (completes), 2 = crashed and a candidate (not a nil read),
// 3 = write / cas, 4 = read
__host__ __device__ inline uint32_t rec_cls(uint32_t f, int32_t a, bool crashed) {
  return 1u | (!crashed ? 2u : 0u) | (crashed && !(f == TBC_F_READ && a == TBC_NIL) ? 4u : 0u) |
         ((f == TBC_F_WRITE || f == TBC_F_CAS) ? 8u : 0u) | (f == TBC_F_READ ? 16u : 0u);
}
#endif
constexpr uint32_t kLookPad = 16;           // records past the last rank (all kLookNone) per history
__host__ __device__ inline uint64_t look_off(uint64_t op_off, uint64_t h, uint32_t mask_words) {
  return (op_off + (uint64_t)kLookPad * h) * (1u + mask_words);
}
__host__ __device__ inline uint64_t look_words(uint64_t total_ops, uint64_t n_hist, uint32_t mask_words) {
  return (total_ops + (uint64_t)kLookPad * (n_hist + 1)) * (1u + mask_words);
}
constexpr uint32_t kSlotMask = 0xFFFFu;     // (f_slot >> 8) & kSlotMask = process slot

// Dominance rules of the wide schedule and of the level sweep (register / cas-register), tbc_opts.dominance:
//   eager reads: an open read whose value is nil or the current state is linearized at once -- it changes
//                nothing, so every later schedule stays possible (configs are kept in that normal form);
//   twin rule:   of several open, not yet linearized calls with the same effect (:write v, or :cas with equal
//                arguments) the one completing first goes first.
// Both are decided from two tables pack_open builds next to the per-front open-call lists:
//   rdm[(F * vpad + vi) * mask_words + w]  slots of the live reads open at front F whose value is nil (vi = 0)
//                                          or vi - 1 (vi = 1 .. vpad - 1); vpad = power of two >= max value + 2
//   twn[e * mask_words + w]                for entry e of lst[] (a live call at one of its fronts): slots of the
//                                          live calls open at that front with the same effect that complete earlier
constexpr uint32_t kRuleEager = 1u, kRuleTwin = 2u;
// narrow kernel under the eager rule: the per-front lists hold the live :write / :cas calls only (a read is never a viable
// candidate of a config in normal form; the rule itself finds the reads through rdm) and the root config starts in normal form
constexpr uint32_t kRuleBranch = 4u;
// COUNT FORM (register / cas-register with crashed calls; specified in oracle/wgl_count.c).  Crashed calls are grouped by effect
// into CLASSES -- (:write v), (:cas [a b]) with a != b; crashed reads and (:cas [a a]) are never candidates -- and a config
// records how many calls of each class are linearized (they go in invocation order): a 128-bit count vector C, class c in a
// field of bit_length(n_c) bits that never straddles a word, instead of one mask bit per crashed call.  Process slots are
// re-used (the host re-numbers the process column: a live call takes the lowest slot free when its process first invokes, a
// process that crashes hands its slot back), so masks stay as wide as the processes alive at once.  Two more rules keep the
// config space small: a crashed call is only linearized right before a call that OBSERVES its value (a config reached by a
// crashed call no absorbed read observed is HOT: bit 30 of the state word; only calls whose precondition is the state are its
// candidates), and a new config is dropped when a visited config with the same key has used no more of any class (Pareto).
//   cmem[]     (at BeamHist.cmem_off) first one 16 B OpRec per CLASS, in order of the class's first invocation:
//              {op = first member's word in this block, f | shift << 8 | width << 16, a, b};  ncr[F] = classes with a member invoked by F;
//              then per class its members in invocation order, inv_rank | op << 32, and a sentinel (all ones)
//   crashed[]  is NOT allocated in the count form (zero elements): nothing may read A.crashed when kRuleCount is set -- the class
//              records are the head of the history's cmem[] block
// Visited-set entries carry the count words behind the mask words; buckets are chosen by (k0, M) alone, so every config with
// one key lies on one probe chain.
constexpr uint32_t kRuleCount = 8u;
// set / bank (wide schedule): the lazy rule of the commutative models (tbcheck.h, TBC_DOM_NO_LAZY_COMMUTING; oracle/wgl_beam.c)
constexpr uint32_t kRuleLazyComm = 16u;
// multi-register (wide schedule, wgl_beam.hip; specified in oracle/wgl_beam.c g_eager_txns / g_txn_por; tbcheck.h TBC_DOM_NO_EAGER_TXNS,
// TBC_DOM_NO_TXN_INDEPENDENCE).  eager txns: an open :txn of micro-reads only, each nil or the state's value of its key, is linearized at
// once (the eager reads' argument word for word: it changes nothing).  txn independence: two txns conflict when one writes a key the other
// reads or writes; at a config whose front is the completion of X only the closure of {X} under "conflicts with" among the open calls
// not yet linearized are candidates -- in any valid continuation the first member of the closure can be moved to the very front (the
// calls before it conflict with no member and overlap all of them in real time).
constexpr uint32_t kRuleTxnEager = 32u, kRuleTxnIndep = 64u;
constexpr uint32_t kHotBit = 0x40000000u;
constexpr uint32_t kCountWords = 2;
// BeamArgs.count_mode
enum : uint32_t { kCountExact = 0, kCountRelaxed = 1 };   // relaxed: every class an unlimited supply (counts stay 0): a superset of the
                                                          // linearizations, so its INVALID verdict bounds the failing completion from above
// FRONT RECORDS (narrow kernel, wgl_narrow.hip): the rdm row of a front, extended to everything else a search step needs of
// that front, so that it costs ONE line of memory instead of a line of each of five arrays (off, ncr, slot8, rk8, rdm) that
// fall out of L2 between the rounds of a history.  front_stride(vpad, mw) u64 words per front at rdm[(op_off + F) * stride]:
//   [0, vpad * mw)        the open-read masks as above
//   + 0   off[F] | nlive << 32        (start and length of the front's list of live open calls)
//   + 1   cnt                          (live + crashed candidates open at F)
//   + 2,3 the process slots of the calls completing at ranks F .. F + 15, one byte each (0 past the end)
//   + 4,5 the read kinds of those ranks (rk8: 0xFF = not a read, else the rdm index of the value read)
__host__ __device__ inline uint32_t front_stride(uint32_t vpad, uint32_t mask_words) { return (vpad * mask_words + 6u + 7u) & ~7u; }
// COMPACT front records: one mask word and at most six row entries in use (nil + values 0..4 -- the reference's registers):
// 64 B, ONE memory transaction instead of two.  Words 0..5 the open-read masks, then
//   6   off[F] | nlive << 32 | cnt << 40
//   7   ranks F .. F + 6, nine bits each: process slot (6) | read kind (3: the rdm index of the value read, 7 = not a read)
constexpr uint32_t kFrontCompactWords = 8, kFrontCompactRanks = 7;
__host__ __device__ inline bool front_compact_ok(uint32_t n_dom, uint32_t mask_words) { return mask_words == 1 && n_dom >= 2 && n_dom <= 6; }
constexpr uint64_t kNarrowMaxOps = 0xFFFFF0ull;          // several histories per wavefront: front + 1 must fit 24 bits of a visited-set key
constexpr int32_t kMaxRuleValue = 30;       // register values 0..30 (vpad <= 32); anything else switches the rules off
__host__ __device__ inline uint32_t rdm_index(int32_t v, uint32_t vpad) {   // row entry of state / read value v
  return (v == TBC_NIL || (uint32_t)(v + 1) >= vpad) ? 0u : (uint32_t)(v + 1);
}

struct __attribute__((aligned(16))) BeamHist {
  uint64_t off_off;     // u32 units: off[] (n_ops + 2), ncr[] at the same offset in its own arena
  uint64_t cmem_off;    // count form: u64 units into cmem[] (the members of the crashed-call classes)
  uint64_t lst_off;     // OpRec units
  uint64_t stack_off;   // u32 units (capacity = table capacity)
  uint64_t tab_off;     // entry units
  uint32_t lst_cap;     // entries available in lst
  uint32_t tab_log2;
  uint32_t n_crashed;   // pack_open: number of crashed ops
  uint32_t status;      // pack_open: 0 ok, 1 = open lists do not fit lst_cap (use the sequential kernel)
  uint32_t n_classes;   // count form: classes of crashed calls (OpRecs at crashed[])
  uint32_t target;      // count form: the search ends VALID when a config has passed this many completions (0 = all of them)
  uint64_t top[kCountWords];   // count form: the top bit of every class's field (field-wise compare of count vectors)
  uint32_t lst_need;    // open_counts_kernel: entries the history's per-front lists hold (tbc_batch_create sizes the list arenas from it)
  uint32_t pad2[3];
};

struct PackOpenArgs {
  const Hist* hist;
  BeamHist* bh;
  const uint8_t* f;
  const int32_t* a;
  const int32_t* b;
  const int32_t* process;
  const uint32_t* scratch;   // pack_kernel's per-op inv_rank / ret_rank (frames arena)
  const Rec* rec;            // pack_kernel's per-process record lists (at Hist.rec_off) ...
  const uint32_t* seg;       // ... and their starts (at Hist.seg_off)
  uint32_t chunks_per_hist;  // ceil(most ops of a history / 64): wavefronts the walk launches per history
  uint32_t branch_lists;     // kRuleBranch: live reads are left out of lst[] / off[] (narrow kernel, eager rule)
  uint32_t* off;
  uint32_t* ncr;
  OpRec* lst;
  OpRec* crashed;            // n_ops entries per history at op_off
  const uint32_t* ret_slot;  // pack_kernel: process slot of the call completing at each rank (at ret_off)
  const uint32_t* ret_op;    // pack_kernel: the call completing at each rank (at ret_off)
  uint64_t* look;            // lookahead records at look_off(), or null (lookahead off / other models)
  uint32_t* tmp;             // n_ops words per history at op_off: scratch for the records
  uint8_t* slot8;            // the same as bytes (mask_words <= 4), at slot8_off(op_off, h): windowed prefetch
  uint32_t front_words;      // 0, or front_stride() / kFrontCompactWords: rdm rows are front records (list location and windows behind the masks)
  uint32_t front_compact;    // 1 = the compact 64 B form
  uint8_t* rk8;              // narrow kernel, eager reads: per rank 0xFF = the completing call is not a read, else the rdm index of the value it read (same offsets as slot8), or null
  uint32_t n_hist;
  uint32_t mask_words;
  uint64_t* twn;             // twin masks, one per lst[] entry (x mask_words), or null
  uint64_t* rdm;             // open-read masks, vpad x mask_words per front at op_off * vpad * mask_words, or null
  uint32_t vpad;
  uint32_t h0;               // histories [h0, n_hist) are worked on by this launch
  const uint64_t* cmem;      // count form: class members (inv_rank | op << 32), or null
  const uint32_t* order_of;  // per history: its own list_order (a race of orders keeps several orders' replicas of a history in one batch), or null
  uint32_t list_order;       // 0 = a front's list in process-slot order; 1 = in order of completion (the walk with lane = front only;
                             // tbc_opts.list_order TBC_ORDER_COMPLETION): the search takes a config's candidates last to first and pops the last child first, so the call
                             // that completes soonest is tried first -- on the bench workload 18 % fewer rounds for the same probes, the longest
                             // history 31 % fewer (oracle/wgl_beam.c, wgl_beam_set_list_order(1))
                             // 2 = in order of completion, the :write calls after everything else (TBC_ORDER_WRITES_LAST; the oracle's list order 4)
                             // 16 + W = in order of completion, a :write as if it completed W ranks later (the oracle's list order 16 + W); 16 + 24 is the library's default
};

// byte offset of history h's slot8[] (n_ret entries + 16 of padding), 8-byte aligned
__host__ __device__ inline uint64_t slot8_off(uint64_t op_off, uint64_t h) { return (op_off + 24ull * h) & ~7ull; }
__host__ __device__ inline uint64_t slot8_bytes(uint64_t total_ops, uint64_t n_hist) { return total_ops + 24ull * n_hist + 32ull; }

struct BeamArgs {
  const Hist* hist;
  const BeamHist* bh;
  const uint32_t* off;
  const uint32_t* ncr;
  const OpRec* lst;
  const OpRec* crashed;
  const uint64_t* look;      // lookahead records, null = lookahead off
  const uint8_t* slot8;
  const uint32_t* ret_slot;
  const uint32_t* ret_op;
  uint32_t* stack;
  uint32_t* dstack;          // second stack (same layout as stack): configs set aside by the lookahead, or null
  uint64_t* tab;             // (2 + mask_words) u64 words per entry; layout is the kernel's own (wgl_beam.hip)
  DevResult* results;
  uint32_t* witness;         // n_ops per history at op_off, may be null
  const uint32_t* work;
  const uint16_t* table;
  uint32_t n_work;
  uint32_t model_kind;
  int32_t init_state;
  uint32_t n_classes;
  uint32_t width;            // K: configs taken off the stack per iteration (1..16)
  uint32_t pad;
  uint64_t max_steps;
  uint64_t time_limit_ticks;
  uint32_t* dbg;
  uint32_t stall_checks;     // narrow kernel, 0 = off: a history whose greatest front has not moved over this many looks at the clock (one every 64
  uint32_t pad_stall;        // rounds) ends UNKNOWN / STEP_LIMIT -- a valid history at this concurrency passes a completion nearly every round, one that
                             // stalls is exhausting the configs in front of a completion nobody can pass: the library hands it to the level
                             // sweep (tbc_api.hip, hand_over_stalled), which refutes it in milliseconds instead of holding the pass for ten passes' time
  const uint32_t* abort;     // wide kernel, optional: one word per history that another stream may set while the search runs -- a history whose
                             // word is set ends UNKNOWN / STEP_LIMIT at its next look at the clock (every 64th round): the relaxed sweep, running
                             // beside the exact search of a count-form history, has refuted it and the prefix search takes over (tbc_api.hip)
  // growth pool: zeroed scratch a wavefront takes a 4x larger visited set + stack from when its own fills up
  uint64_t* pool;
  unsigned long long* pool_cursor;   // words handed out so far (zeroed before the launch)
  uint64_t pool_words;
  uint32_t max_tab_log2;             // growth stops here (tbc_opts.max_visited_bytes)
  uint32_t pad2;
  const int32_t* pool_vals;          // wide op values (multi-register micro-ops)
  uint64_t* cfg;                     // as SearchArgs.cfg
  int32_t model_aux;                 // commutative models: pool offset of the per-front table
  uint32_t round_budget;             // 0 = none; exceeded => TBC_CAUSE_ROUND_BUDGET (host escalates)
  uint32_t n_keys;                   // bank: number of accounts
  uint32_t rules;                    // kRuleEager | kRuleTwin (register family, single-wavefront wide schedule)
  const uint64_t* twn;               // as PackOpenArgs
  const uint64_t* rdm;
  uint32_t vpad;
  uint32_t pad3;
  const uint8_t* rk8;                // as PackOpenArgs (narrow kernel only)
  uint32_t front_words;              // u64 words per front record in rdm (narrow kernel only; kFrontCompactWords = the compact form)
  const uint64_t* cmem;              // count form: class members (tbc_internal.h, kRuleCount)
  uint32_t count_mode;               // kCountExact / kCountRelaxed
  uint32_t tab_stride;               // narrow kernel: u64 words per entry of the batch's visited-set arena: mask_words + 2, or mask_words + 1
                                     // when nobody wants a witness (no parent words: the kernel would not write them anyway)
  uint32_t pad5;
  uint32_t epoch;                    // narrow kernel: this pass's tag in the visited-set keys, 1..255 (wgl_narrow_impl.h, entry_empty); 0 = none
  uint32_t first_dynamic;            // narrow kernel: work items below this are dealt to the wavefronts at launch (wave w, group g: w * H + g) ...
  unsigned int* next_work;           // ... the others are taken from this counter (zeroed before the launch) as groups finish
  uint32_t* abort_set;               // wide kernel, optional: a history that ends VALID or INVALID sets its word here -- the same history searched in
                                     // ANOTHER list order by another batch on another stream (whose BeamArgs.abort this is) may stop (batch_run.hip, race_orders)
  const uint32_t* abort_map;         // ... which word of abort / abort_set is history hidx's (null: word hidx)
  uint32_t* park;                    // wide kernel, optional: kParkWords words per history -- the search state as the kernel leaves it ...
  uint32_t resume;                   // ... and, 1: the search is taken up from there (the visited set, the stacks and the growth pool as they were left) instead of from the root
  uint32_t pad6;
  uint32_t* progress;                // optional, HOST memory: word 0 = histories decided so far, as last published (tbc_batch_progress) ...
  uint32_t* progress_dev;            // ... from this count in HBM (wave_env.h count_decided)
};
constexpr uint32_t kParkWords = 32;

// ---- level sweep (jit_sweep.hip): knossos.linear as segments swept by one wavefront each
constexpr uint32_t kSweepCap = 512;        // configs per LDS set when many wavefronts must share a CU (38 KB, four per CU) ...
constexpr uint32_t kSweepCapMid = 1024;    // ... when a few histories must be quick (70 KB, two per CU): fewer second passes;
constexpr uint32_t kSweepCapBig = 2048;    // a larger level ends the segment with kSegOverflow and it is swept again with sets of this size
constexpr uint32_t kSweepCandMax = 128;    // open calls (live + crashed) per level held in LDS
constexpr uint32_t kSweepMaxSegs = 512;    // cuts per history
enum : uint32_t { kSegNone = 0, kSegOk = 1, kSegOverflow = 2 };
constexpr uint32_t kSweepSlices = TBC_SWEEP_SLICES;   // wavefronts per segment at most: 32 origins each, <= 128 origins per segment
using SegResult = tbc_sweep_rel;           // one per (history, segment, slice); public because ranks exchange it (include/tbcheck.h)
struct SweepArgs {
  const Hist* hist;
  const BeamHist* bh;
  const uint32_t* off;
  const uint32_t* ncr;
  const OpRec* lst;
  const OpRec* crashed;
  const uint64_t* twn;       // may be null (no twin rule)
  const uint64_t* rdm;       // may be null (no eager reads)
  const uint8_t* slot8;
  uint32_t* cuts;            // max_segs per history: first front of segment k, kInf = no such segment
  SegResult* seg;            // max_segs * kSweepSlices per history
  const uint16_t* table;
  const int32_t* pool_vals;
  uint32_t n_hist;
  uint32_t max_segs;
  uint32_t seg_target;       // wanted segment length in completions, 0 = one segment
  uint32_t cut_open;         // m: a cut needs <= m open calls (n_dom << m <= 32 * kSweepSlices)
  uint32_t n_dom;            // states of the origin domain: nil + 0..vmax
  uint32_t vpad;
  uint32_t rules;
  uint32_t model_kind;
  int32_t init_state;
  uint32_t n_classes;
  uint32_t n_keys;
  // dump pass
  uint32_t dump_hist, dump_seg, dump_slice, stop_level, live_mask;
  uint32_t shard_rank, shard_world;   // this rank sweeps the wavefronts with (segment * kSweepSlices + slice) % world == rank
  const uint32_t* seg_list;  // second pass: n_list (history, segment, slice) triples to sweep with the big sets, else null
  uint32_t n_list, pad4;
  uint64_t* dump_cfg;        // records {front + 1 | state << 32, mask, TBC_NO_OP} as SearchArgs.cfg, kCfgCap at most
  uint32_t* dump_count;      // number of configs at that level (may exceed kCfgCap)
  // the RELAXED sweep of a count-form history (reach_table.h; K6w only): crashed calls as classes in unlimited supply.  reach_hdr: two
  // words per history {where its table starts in reach[], epochs}; null = the plain sweep.  (ncr must then be all zero: no crashed call
  // is a candidate of its own.)
  const uint32_t* reach;
  const uint32_t* reach_hdr;
};
bool launch_sweep(const SweepArgs& a, void* stream);
// K6w (jit_sweep_wg.hip): the first pass with `waves` (4 / 8) wavefronts per segment; false = not for this batch (the caller takes K6)
bool launch_sweep_wg(const SweepArgs& a, void* stream);

// skip_counts: open_counts_kernel's tables are already there (launch_pack_wg built them with the pack)
void launch_pack_open(const PackOpenArgs& a, void* stream, bool skip_counts = false);
// open_counts_kernel alone: how many entries each history's per-front lists hold (BeamHist.lst_need), before those arenas exist
void launch_open_counts(const PackOpenArgs& a, void* stream);
bool launch_beam(const BeamArgs& a, uint32_t mask_words, uint32_t n_blocks, void* stream);
// several histories per wavefront (wgl_narrow.hip): `lanes` = 8 / 16 / 32 lanes per history, one config per iteration
bool narrow_supported(uint32_t mask_words, uint32_t lanes);
// waves_per_simd: 0 = as many as fit (4); fewer leaves room for the pack kernels of the next chunk to run beside the search
bool launch_narrow(const BeamArgs& a, uint32_t mask_words, uint32_t lanes, void* stream, uint32_t waves_per_simd = 0);

// kernel launchers (defined in the .hip files)
void launch_pack(const PackArgs& a, void* stream);
// K1 for one history or a handful (pack_one.hip): does the body take such a history; launch it for [a.h0, a.n_hist) (false = not launched)
bool pack_one_fits(uint32_t model_kind, uint32_t n_ops, uint32_t n_events, uint32_t n_slots);
bool launch_pack_one(const PackArgs& a, void* stream);
// K1 + open counts for a batch (pack_one.hip, pack_wg_kernel): four wavefronts per history, tables in LDS; o = what launch_pack_open()
// would give open_counts_kernel.  false = not launched (the caller takes pack_kernel, and launch_pack_open() without skip_counts)
bool pack_one_counts_fits(uint32_t model_kind, uint32_t n_ops, uint32_t n_events, uint32_t n_slots);
bool launch_pack_one_counts(const PackArgs& a, const PackOpenArgs& o, void* stream);      // (pack_one_kernel + open counts, one history or a handful)
bool pack_wg_fits(uint32_t model_kind, uint32_t n_ops, uint32_t n_events, uint32_t n_slots);
bool pack_wg64_fits(uint32_t model_kind, uint32_t n_ops, uint32_t n_events, uint32_t n_slots);      // ... and has at most 64 process slots: the 19 KB geometry
bool launch_pack_wg(const PackArgs& a, const PackOpenArgs& o, void* stream, bool slots64);
// returns false if mw is unsupported
bool launch_search(const SearchArgs& a, uint32_t mask_words, uint32_t n_blocks, void* stream);
uint32_t search_frame_words(uint32_t mask_words);

}  // namespace tbc
