// batch_stream.hip -- FRESH INPUTS into a batch's arenas (include/tbcheck.h, "streaming": tbc_batch_map_input, tbc_batch_submit_input,
// tbc_batch_reload, tbc_batch_input_info).
//
// The reference calls a checker once per history (/root/reference/src/tigerbeetle/core.clj:139-146: checker/compose over the test's ONE
// history; workloads/set_full.clj:155-158: independent/checker, once per key): a caller that checks many histories never checks one
// twice, so a batch whose inputs are resident for many passes is a benchmark's situation, not a caller's.  Round 5's line said what that
// costs: 321k histories/s over two resident batches, 12-22k when every batch is created for its histories (85 GB of hipMalloc, 5 GB of
// pageable columns over PCIe, the list-sizing pass, the destroy).  Here the batch's arenas, streams and layout decisions stay and only
// the histories change:
//   * WIRE FORMAT, 12 B an op instead of 21: f | a | b | process in one word (register family: values 0..254 or nil, 4,096 processes),
//     inv_pos, ret_pos.  5.05 -> 2.88 GB per 32,768 bench histories: 55 ms of PCIe instead of 97.
//   * PINNED SLOTS the caller fills in place (tbc_batch_map_input: a JNA Memory / direct ByteBuffer over it, a numpy view) -- no staging
//     copy inside the library, and the copy to the device is one DMA per column.
//   * TWO DEVICE STAGES: tbc_batch_submit_input queues the copy of input k + 1 on the batch's copy stream and returns; it runs under
//     the pack and the search of input k.  The run that consumes an input waits for its copy, unpacks the wire columns into the op columns
//     (stream_unpack_kernel: 33 B an op of HBM traffic, ~3 ms per 2.4 * 10^8 ops) and checks on the way what the batch's layout
//     decisions cannot take (a value beyond the batch's value domain; a crashed call in a batch created without any).
//   * THE LAYOUT of the new histories (descriptors, arena offsets) is the host's O(histories) loop of tbc_batch_create
//     (layout_histories); the one thing create asks the DEVICE for -- how long each history's per-front lists are -- stays on the
//     device: the pack leaves the lengths, stream_assign_lists_kernel deals the places, the walk fills them.  No trip to the host
//     between the pack and the walk.
//   * ARENAS GROW (6 % at a time) when an input needs more than the batch has; histories whose lists did not fit are answered by the
//     sequential kernel that once (batch_run.hip, list_overflow_fallback) and the list arenas are larger for the next input.
//   * A COUNT-FORM BATCH (crashed calls with an effect under the default rules: what the reference's nemesis makes) takes fresh inputs
//     too: the classes of crashed calls and the re-numbered process slots of every history are planned on the HOST when the input is
//     submitted (plan_count_input: batch_create.hip's build_count_form over the wire columns, on up to 16 threads; the words with
//     the slot numbers go to a pinned copy, which is what travels), the class records go up beside the wire columns, and since such a batch has no sequential
//     fallback for lists that outgrow their arena, the lists' lengths are counted on the host as well and the arena grown before the run.
// What it does not take (TBC_ERR_UNSUPPORTED: destroy and create): batches of the level sweep (a count-form batch of <= 8 histories has
// the relaxed sweep beside it), set / bank / multi-register / table models, batches of tbc_check's persistent contexts.
#include "tbc_batch.h"

using namespace tbc;

namespace tbc {
namespace {

enum : uint32_t { kInBadValue = 1u, kInBadCrashed = 2u };
constexpr uint32_t kMaxInputSlots = 16;

// wire word -> the six op columns, four ops a thread (one 16 B load per wire column, one 4 B + five 16 B stores); the tail by single ops
__global__ __launch_bounds__(256) void stream_unpack_kernel(const uint32_t* __restrict__ word, const uint32_t* __restrict__ inv, const uint32_t* __restrict__ ret, uint64_t T,
                                                            uint8_t* __restrict__ of, int32_t* __restrict__ oa, int32_t* __restrict__ ob, int32_t* __restrict__ op,
                                                            uint32_t* __restrict__ oinv, uint32_t* __restrict__ oret, uint32_t vmax, uint32_t allow_crashed, uint32_t* flags) {
  const uint64_t i4 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4ull;
  if (i4 >= T) return;
  uint32_t bad = 0u;
  const auto one = [&](uint32_t w, uint32_t r, uint32_t& f, int32_t& a, int32_t& b, int32_t& p) {
    f = w & 15u;
    const uint32_t a8 = (w >> 4) & 0xFFu, b8 = (w >> 12) & 0xFFu;
    a = a8 == TBC_WIRE_NIL ? TBC_NIL : (int32_t)a8;
    b = b8 == TBC_WIRE_NIL ? TBC_NIL : (int32_t)b8;
    p = (int32_t)(w >> 20);
    if (f <= TBC_F_CAS && ((a8 != TBC_WIRE_NIL && a8 > vmax) || (f == TBC_F_CAS && b8 != TBC_WIRE_NIL && b8 > vmax))) bad |= kInBadValue;
    if (r == TBC_POS_CRASHED && !allow_crashed) bad |= kInBadCrashed;
  };
  if (i4 + 4ull <= T) {
    const uint4 w = *reinterpret_cast<const uint4*>(word + i4);
    const uint4 iv = *reinterpret_cast<const uint4*>(inv + i4);
    const uint4 rt = *reinterpret_cast<const uint4*>(ret + i4);
    uint32_t f0, f1, f2, f3;
    int4 a, b, p;
    one(w.x, rt.x, f0, a.x, b.x, p.x);
    one(w.y, rt.y, f1, a.y, b.y, p.y);
    one(w.z, rt.z, f2, a.z, b.z, p.z);
    one(w.w, rt.w, f3, a.w, b.w, p.w);
    *reinterpret_cast<uint32_t*>(of + i4) = f0 | (f1 << 8) | (f2 << 16) | (f3 << 24);
    *reinterpret_cast<int4*>(oa + i4) = a;
    *reinterpret_cast<int4*>(ob + i4) = b;
    *reinterpret_cast<int4*>(op + i4) = p;
    *reinterpret_cast<uint4*>(oinv + i4) = iv;
    *reinterpret_cast<uint4*>(oret + i4) = rt;
  } else {
    for (uint64_t i = i4; i < T; i++) {
      uint32_t f; int32_t a, b, p;
      const uint32_t r = ret[i];
      one(word[i], r, f, a, b, p);
      of[i] = (uint8_t)f; oa[i] = a; ob[i] = b; op[i] = p; oinv[i] = inv[i]; oret[i] = r;
    }
  }
  if (bad) atomicOr(flags, bad);
}

// The places of the histories' per-front lists, from the lengths the pack left (BeamHist.lst_need): what tbc_batch_create does on the
// host after its sizing pass -- lst_cap = max(1, need), lst_off = the running sum -- by ONE workgroup (1,024 threads, a contiguous run of
// histories each, one scan).  A history whose lists end beyond the arena is marked (status 1: the walk and the search skip it, the run
// answers it with the sequential kernel) and *over says so.
__global__ __launch_bounds__(1024) void stream_assign_lists_kernel(BeamHist* bh, uint32_t nh, uint64_t cap, uint32_t* over) {
  __shared__ uint64_t part[1024];
  const uint32_t t = threadIdx.x;
  const uint32_t per = (nh + 1023u) / 1024u;
  const uint32_t lo = t * per < nh ? t * per : nh, hi = lo + per < nh ? lo + per : nh;
  uint64_t sum = 0;
  for (uint32_t h = lo; h < hi; h++) { const uint32_t need = bh[h].lst_need; sum += need ? need : 1u; }
  part[t] = sum;
  __syncthreads();
  for (uint32_t d = 1; d < 1024u; d <<= 1) {          // inclusive scan (Hillis-Steele)
    const uint64_t v = t >= d ? part[t - d] : 0ull;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  uint64_t at = part[t] - sum;
  bool any_over = false;
  for (uint32_t h = lo; h < hi; h++) {
    const uint32_t need = bh[h].lst_need, c = need ? need : 1u;
    if (at + c > cap) { bh[h].status = 1u; bh[h].lst_cap = 0u; bh[h].lst_off = 0ull; bh[h].n_crashed = 0u; any_over = true; }
    else { bh[h].lst_cap = c; bh[h].lst_off = at; }
    at += c;
  }
  if (any_over) atomicOr(over, 1u);
}

const char* streamable(const tbc_batch* B) {
  if (B->borrowed) return "a batch of tbc_check's persistent context";
  if (B->width <= 1) return "the sequential knossos.wgl schedule (search_width 1)";
  if (B->sweep || B->rsweep) return "a batch of the level sweep";
  if (!(B->model.kind == TBC_MODEL_REGISTER || B->model.kind == TBC_MODEL_CAS_REGISTER || B->model.kind == TBC_MODEL_MUTEX)) return "a model outside register / cas-register / mutex";
  if (B->pool_len) return "a batch with a value pool";
  return nullptr;
}

// the streams, events and small device words of the streaming path, once per batch
tbc_status ensure_stream(tbc_batch* B) {
  if (B->stream_copy) return TBC_OK;
  if (const char* why = streamable(B)) { set_error("this batch takes no fresh inputs: it is %s -- destroy and create", why); return TBC_ERR_UNSUPPORTED; }
  HIP_TRY(hipSetDevice(B->device));
  HIP_TRY(hipStreamCreateWithFlags(&B->stream_copy, hipStreamNonBlocking));
  for (int i = 0; i < 2; i++) {
    HIP_TRY(hipEventCreateWithFlags(&B->ev_stage[i], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&B->ev_unpacked[i], hipEventDisableTiming));
    HIP_TRY(hipEventCreate(&B->ev_copy[i][0]));
    HIP_TRY(hipEventCreate(&B->ev_copy[i][1]));
  }
  tbc_status s;
  if ((s = B->d_in_flags.alloc(4)) || (s = B->d_list_over.alloc(4))) return s;
  HIP_TRY(hipHostMalloc((void**)&B->in_flags_host, 4 * sizeof(uint32_t), hipHostMallocDefault));
  // what a slot and a stage hold: the histories of the first input, and its ops with an eighth to spare
  B->in_hist_cap = B->n_hist;
  B->in_ops_cap = ((B->total_ops + B->total_ops / 8 + 1023ull) & ~1023ull) + 1024ull;
  return TBC_OK;
}

size_t slot_bytes(const tbc_batch* B) {
  const size_t a = ((size_t)(B->in_hist_cap + 1) * 8 + 255) & ~(size_t)255, b = ((size_t)B->in_hist_cap * 4 + 255) & ~(size_t)255;
  return a + 2 * b + 3 * (size_t)B->in_ops_cap * 4;
}
void slot_pointers(const tbc_batch* B, char* mem, tbc_batch_input* out) {
  const size_t a = ((size_t)(B->in_hist_cap + 1) * 8 + 255) & ~(size_t)255, b = ((size_t)B->in_hist_cap * 4 + 255) & ~(size_t)255;
  out->n_hist_cap = B->in_hist_cap; out->reserved0 = 0; out->ops_cap = B->in_ops_cap;
  out->op_off = (uint64_t*)mem;
  out->n_events = (uint32_t*)(mem + a);
  out->n_process = (uint32_t*)(mem + a + b);
  out->word = (uint32_t*)(mem + a + 2 * b);
  out->inv_pos = out->word + B->in_ops_cap;
  out->ret_pos = out->inv_pos + B->in_ops_cap;
}

template <typename T>
tbc_status ensure(DevBuf<T>& buf, uint64_t need, bool* grew) {
  if (buf.p && buf.n >= need) return TBC_OK;
  *grew = true;
  return buf.regrow((size_t)(need + need / 16));
}

// every per-history arena holds the input laid out as `tot` (the allocation list of batch_create.hip's alloc_arenas, for the batches
// streamable() lets through); an arena that is too small is replaced by one 6 % larger than asked
tbc_status ensure_arenas(tbc_batch* B, const LayoutTotals& t) {
  const uint64_t T = t.total_ops;
  const uint32_t nhc = B->in_hist_cap, MW = B->mask_words;
  bool grew = false, grew_tab = false;
  tbc_status s;
  if ((s = ensure(B->d_f, T, &grew)) || (s = ensure(B->d_a, T, &grew)) || (s = ensure(B->d_b, T, &grew)) || (s = ensure(B->d_proc, T, &grew)) ||
      (s = ensure(B->d_inv, T, &grew)) || (s = ensure(B->d_ret, T, &grew)) || (s = ensure(B->d_ret_slot, T, &grew)) || (s = ensure(B->d_ret_op, T, &grew)) ||
      (s = ensure(B->d_bitmap, t.bm_n, &grew)) || (s = ensure(B->d_wpre, t.bm_n, &grew)) || (s = ensure(B->d_off, t.boff_n, &grew)) || (s = ensure(B->d_ncr, t.boff_n, &grew)) ||
      (s = ensure(B->d_rec, t.rec_n, &grew)) || (s = ensure(B->d_seg, t.seg_n, &grew)) || (s = ensure(B->d_frames, t.frame_n, &grew)) ||
      (s = ensure(B->d_slot8, slot8_bytes(T, nhc), &grew)) || (s = ensure(B->d_stack, t.bstack_n, &grew)))
    return s;
  if (B->opts.want_witness && (s = ensure(B->d_witness, T, &grew))) return s;
  if (B->any_crashed && !B->count_form && (s = ensure(B->d_crashed, T, &grew))) return s;
  if (B->count_form && (s = ensure(B->d_cmem, t.bocc_n, &grew))) return s;
  if (B->lanes && (s = ensure(B->d_rk8, slot8_bytes(T, nhc), &grew))) return s;
  if (B->lanes) { if ((s = ensure(B->d_rdm, T * B->front_words(), &grew))) return s; }
  else if (B->reg_rules() && (s = ensure(B->d_rdm, T * B->vpad * MW, &grew))) return s;
  if (B->lookahead) {
    const bool by_front = MW == 1 && B->vpad <= 32;
    if ((s = ensure(B->d_look, look_words(T, nhc, MW), &grew)) || (!by_front && (s = ensure(B->d_looktmp, T, &grew))) || (s = ensure(B->d_dstack, t.bstack_n, &grew))) return s;
  }
  if ((s = ensure(B->d_btab, t.btab_n * B->tab_stride(), &grew_tab))) return s;
  if (grew_tab) {
    B->epoch = 0;          // (a new visited-set arena: zeroed by the run before its first pass, as a batch's first arena is)
    // ... and the growth pool keeps its share of it (alloc_arenas' rule)
    const uint64_t words = std::min<uint64_t>(t.btab_n * B->tab_stride() * (B->lanes ? 1u : 3u) / 10, (32ull << 30) / 8);
    if (words > B->d_pool.n && (s = B->d_pool.regrow(words))) return s;
  }
  (void)grew;
  return TBC_OK;
}

// A count-form batch's fresh input: every history's classes of crashed calls and its process column re-numbered (slots re-used), as
// tbc_batch_create plans them (build_count_form) -- here from the WIRE columns of the pinned slot, a run of histories per host thread.
// The wire words with the slot numbers in place of the process numbers go to `planned` (pinned, beside the slot: the device unpacks what
// the pack kernel is to see; the caller's own words stay as written, so a slot may be submitted again as it is); n_slots[] and
// P.count_hist / P.cmem / P.lst_total are what the layout, the upload and the list arena ask for.
tbc_status plan_count_input(tbc_batch* B, const tbc_batch_input& in, uint32_t* planned, uint32_t nh, tbc_pending_input& P, std::vector<uint32_t>& n_slots) {
  P.count_hist.assign(nh, CountHist{});
  n_slots.assign(nh, 1u);
  std::vector<uint64_t> lists(nh, 0);
  const bool cas_model = B->model.kind == TBC_MODEL_CAS_REGISTER;
  const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  const unsigned nt = in.op_off[nh] > (1ull << 20) ? std::min(16u, hw) : 1u;
  std::vector<int> bad(nt, -1);
  const auto work = [&](unsigned t) {
    std::vector<uint8_t> f; std::vector<int32_t> a, b, proc, slot; std::vector<uint32_t> rets, pre; std::vector<int32_t> slot_of; std::vector<uint8_t> used;
    for (uint32_t h = (uint32_t)((uint64_t)nh * t / nt); h < (uint32_t)((uint64_t)nh * (t + 1) / nt); h++) {
      const uint64_t o0 = in.op_off[h], n = in.op_off[h + 1] - o0;
      f.resize(n); a.resize(n); b.resize(n); proc.resize(n); slot.assign(n + 1, 0);
      for (uint64_t i = 0; i < n; i++) {
        const uint32_t w = in.word[o0 + i], a8 = (w >> 4) & 0xFFu, b8 = (w >> 12) & 0xFFu;
        f[i] = (uint8_t)(w & 15u); a[i] = a8 == TBC_WIRE_NIL ? TBC_NIL : (int32_t)a8; b[i] = b8 == TBC_WIRE_NIL ? TBC_NIL : (int32_t)b8; proc[i] = (int32_t)(w >> 20);
      }
      tbc_ops c{};
      c.n = (uint32_t)n; c.f = f.data(); c.a = a.data(); c.b = b.data(); c.process = proc.data(); c.inv_pos = in.inv_pos + o0; c.ret_pos = in.ret_pos + o0;
      if (!build_count_form(c, 0, n, in.n_process[h], cas_model, slot.data(), P.count_hist[h], rets, slot_of, used)) { if (bad[t] < 0) bad[t] = (int)h; continue; }
      n_slots[h] = std::max(1u, P.count_hist[h].n_slots);
      for (uint64_t i = 0; i < n; i++) planned[o0 + i] = (in.word[o0 + i] & 0xFFFFFu) | ((uint32_t)slot[i] << 20);
      c.process = slot.data();
      lists[h] = open_list_entries(c, 0, n, in.n_events[h], n_slots[h], pre, false);
    }
  };
  if (nt == 1) work(0);
  else {
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++) th.emplace_back(work, t);
    for (auto& x : th) x.join();
  }
  for (int h : bad) if (h >= 0) {
    set_error("history %d of the input: its crashed calls do not fit the count form (more than 128 bits of counts, or a process number out of range): destroy and create", h);
    return TBC_ERR_UNSUPPORTED;
  }
  size_t words = 0;
  for (uint32_t h = 0; h < nh; h++) { words += P.count_hist[h].words.size(); P.lst_total += std::max<uint64_t>(1, lists[h]); }
  P.cmem.clear(); P.cmem.reserve(words);
  for (uint32_t h = 0; h < nh; h++) P.cmem.insert(P.cmem.end(), P.count_hist[h].words.begin(), P.count_hist[h].words.end());
  return TBC_OK;
}

}  // namespace

void stream_release(tbc_batch* B) {
  for (auto& sl : B->in_slots) {
    if (sl.copied) { (void)hipEventSynchronize(sl.copied); (void)hipEventDestroy(sl.copied); }
    if (sl.mem) (void)hipHostFree(sl.mem);
    if (sl.planned) (void)hipHostFree(sl.planned);
  }
  B->in_slots.clear();
  if (B->stream_copy) { (void)hipStreamSynchronize(B->stream_copy); (void)hipStreamDestroy(B->stream_copy); B->stream_copy = nullptr; }
  for (int i = 0; i < 2; i++) {
    if (B->ev_stage[i]) (void)hipEventDestroy(B->ev_stage[i]);
    if (B->ev_unpacked[i]) (void)hipEventDestroy(B->ev_unpacked[i]);
    if (B->ev_copy[i][0]) (void)hipEventDestroy(B->ev_copy[i][0]);
    if (B->ev_copy[i][1]) (void)hipEventDestroy(B->ev_copy[i][1]);
    B->d_stage[i].release();
  }
  B->d_in_flags.release(); B->d_list_over.release();
  if (B->in_flags_host) { (void)hipHostFree(B->in_flags_host); B->in_flags_host = nullptr; }
}

// The run that is about to start consumes the oldest submitted input, if there is one.
tbc_status stream_consume(tbc_batch* B, bool* consumed) {
  *consumed = false;
  if (B->pending.empty()) return TBC_OK;
  tbc_pending_input& P = B->pending.front();
  hipStream_t s = B->stream;
  struct Drop { tbc_batch* B; ~Drop() { B->pending.pop_front(); } } drop{B};      // (consumed or refused: it leaves the queue either way)
  tbc_status st;
  LayoutTotals want = P.tot;
  if (!B->lists_headroom) {          // the first fresh input: every arena was sized to the element for the first histories -- room for others
    // (asking for a sixty-fourth more than the first input makes ensure() replace each arena by one 6 % larger, once, here -- not in the
    // middle of a stream of inputs the first time one of them is a few ops longer)
    const auto more = [](uint64_t now, uint64_t first) { return std::max(now, first + first / 64 + 1); };
    want.total_ops = more(want.total_ops, B->cap.total_ops); want.rec_n = more(want.rec_n, B->cap.rec_n); want.seg_n = more(want.seg_n, B->cap.seg_n);
    want.bm_n = more(want.bm_n, B->cap.bm_n); want.frame_n = more(want.frame_n, B->cap.frame_n); want.boff_n = more(want.boff_n, B->cap.boff_n);
    want.bstack_n = more(want.bstack_n, B->cap.bstack_n); want.btab_n = more(want.btab_n, B->cap.btab_n);
    B->lists_headroom = true;
    const uint64_t want = B->d_lst.n + B->d_lst.n / 16 + 4096;
    if (!(B->lanes && want >= (1ull << 32))) {
      if ((st = B->d_lst.regrow(want))) return st;
      if (B->reg_rules() && (st = B->d_twn.regrow(want * B->mask_words))) return st;
    }
  }
  if ((st = ensure_arenas(B, want))) return st;
  if (B->count_form && P.lst_total > B->d_lst.n) {          // (no fallback for lists that do not fit: the arena grows BEFORE the run)
    const uint64_t need = P.lst_total + P.lst_total / 16;
    if (B->lanes && need >= (1ull << 32)) { set_error("the input's per-front lists exceed what several histories per wavefront address: destroy and create"); return TBC_ERR_UNSUPPORTED; }
    if ((st = B->d_lst.regrow(need))) return st;
    if (B->reg_rules() && (st = B->d_twn.regrow(need * B->mask_words))) return st;
    B->lists_regrown++;
  }
  B->count_device_bytes();
  const uint64_t T = P.total_ops;
  const uint32_t* stage = B->d_stage[P.stage].p;
  HIP_TRY(hipStreamWaitEvent(s, B->ev_stage[P.stage], 0));
  HIP_TRY(hipMemsetAsync(B->d_in_flags.p, 0, 4, s));
  // a value the batch's tables have no row for (the rules' tables hold nil + 0..n_dom - 2; without the rules any wire value goes)
  const uint32_t vmax = B->vpad ? B->n_dom - 2u : 254u;
  if (T) {
    const uint64_t threads = (T + 3) / 4;
    hipLaunchKernelGGL(stream_unpack_kernel, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, s, stage, stage + B->in_ops_cap, stage + 2 * B->in_ops_cap, T,
                       B->d_f.p, B->d_a.p, B->d_b.p, B->d_proc.p, B->d_inv.p, B->d_ret.p, vmax, B->any_crashed ? 1u : 0u, B->d_in_flags.p);
    HIP_TRY(hipGetLastError());
  }
  if (B->count_form && !P.cmem.empty()) HIP_TRY(hipMemcpyAsync(B->d_cmem.p, P.cmem.data(), P.cmem.size() * 8, hipMemcpyHostToDevice, s));      // (synchronised below, before P goes)
  HIP_TRY(hipMemcpyAsync(B->in_flags_host, B->d_in_flags.p, 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipEventRecord(B->ev_unpacked[P.stage], s));
  HIP_TRY(hipStreamSynchronize(s));          // (the pack must not run over an input the batch cannot take: its crashed-call arena may not exist)
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, B->ev_copy[P.stage][0], B->ev_copy[P.stage][1]) == hipSuccess) B->in_copy_ns = (uint64_t)(ms * 1e6);
  B->in_bytes_copied = T * 12ull;
  const uint32_t flags = B->in_flags_host[0];
  if (flags) {
    set_error("the submitted input does not fit this batch's layout decisions (%s%s%s): destroy and create the batch from it",
              (flags & kInBadValue) ? "a register value beyond the batch's value domain" : "", (flags == (kInBadValue | kInBadCrashed)) ? "; " : "",
              (flags & kInBadCrashed) ? "a crashed call in a batch created from histories without any" : "");
    B->inputs_stale = true;          // (the op columns hold the refused input: there is nothing resident to run until the next one)
    return TBC_ERR_UNSUPPORTED;
  }
  B->inputs_stale = false;
  B->hist.swap(P.hist);
  B->bh.swap(P.bh);
  if (B->count_form) B->count_hist.swap(P.count_hist);
  B->n_hist = P.n_hist;
  B->total_ops = T;
  B->max_ops = P.max_ops;
  B->res_host.resize(B->n_hist);
  B->inputs_fresh = false;
  B->assign_lists = true;
  B->partial_done = false;
  B->inputs_consumed++;
  *consumed = true;
  return TBC_OK;
}

tbc_status stream_assign_lists(tbc_batch* B) {
  HIP_TRY(hipMemsetAsync(B->d_list_over.p, 0, 4, B->stream));
  hipLaunchKernelGGL(stream_assign_lists_kernel, dim3(1), dim3(1024), 0, B->stream, B->d_bh.p, B->n_hist, (uint64_t)B->d_lst.n, B->d_list_over.p);
  HIP_TRY(hipGetLastError());
  return TBC_OK;
}

// after the run: lists that did not fit were answered by the sequential kernel; the next input finds larger arenas
tbc_status stream_after_run(tbc_batch* B, const HostBuf<BeamHist>& bh_back) {
  if (bh_back.size() != B->n_hist) return TBC_OK;
  uint64_t need = 0;
  bool over = false;
  for (uint32_t h = 0; h < B->n_hist; h++) { need += std::max(1u, bh_back[h].lst_need); over = over || bh_back[h].status == 1u; }
  if (!over || need <= B->d_lst.n) return TBC_OK;
  const uint64_t want = need + need / 16;
  if (B->lanes && want >= (1ull << 32)) return TBC_OK;          // (the narrow kernel's 32-bit list offsets: such an input keeps its fallback)
  tbc_status st;
  if ((st = B->d_lst.regrow(want))) return st;
  if (B->reg_rules() && (st = B->d_twn.regrow(want * B->mask_words))) return st;
  B->lists_regrown++;
  B->count_device_bytes();
  return TBC_OK;
}

}  // namespace tbc

extern "C" {

tbc_status tbc_batch_map_input(tbc_batch* b, uint32_t slot, tbc_batch_input* out) {
  if (!b || !out) { set_error("tbc_batch_map_input: null argument"); return TBC_ERR_INVALID_ARG; }
  if (slot >= kMaxInputSlots) { set_error("tbc_batch_map_input: slot %u (at most %u slots)", slot, kMaxInputSlots); return TBC_ERR_INVALID_ARG; }
  try {
    tbc_status s = ensure_stream(b);
    if (s != TBC_OK) return s;
    HIP_TRY(hipSetDevice(b->device));
    if (b->in_slots.size() <= slot) b->in_slots.resize(slot + 1);
    tbc_batch::InputSlot& sl = b->in_slots[slot];
    if (!sl.mem) {
      const size_t bytes = slot_bytes(b);
      // (plain pinned memory, cached on the host side: the caller WRITES it, the DMA engine reads it)
      HIP_TRY(hipHostMalloc((void**)&sl.mem, bytes, hipHostMallocDefault));
      sl.bytes = bytes;
      HIP_TRY(hipEventCreateWithFlags(&sl.copied, hipEventDisableTiming));
    }
    if (sl.busy) { HIP_TRY(hipEventSynchronize(sl.copied)); sl.busy = false; }      // its last input is on the device: the caller may write again
    slot_pointers(b, sl.mem, out);
    return TBC_OK;
  } catch (const std::bad_alloc&) { set_error("host allocation failed"); return TBC_ERR_OOM; }
  catch (...) { set_error("unexpected exception"); return TBC_ERR_HIP; }
}

tbc_status tbc_batch_submit_input(tbc_batch* b, uint32_t slot, uint32_t n_hist) {
  if (!b) { set_error("tbc_batch_submit_input: null batch"); return TBC_ERR_INVALID_ARG; }
  if (!b->stream_copy || slot >= b->in_slots.size() || !b->in_slots[slot].mem) { set_error("tbc_batch_submit_input: slot %u was never mapped (tbc_batch_map_input)", slot); return TBC_ERR_INVALID_ARG; }
  if (n_hist == 0 || n_hist > b->in_hist_cap) { set_error("tbc_batch_submit_input: %u histories, the batch holds 1..%u", n_hist, b->in_hist_cap); return TBC_ERR_INVALID_ARG; }
  if (b->pending.size() >= 2) { set_error("tbc_batch_submit_input: two inputs are waiting already -- tbc_batch_run consumes one"); return TBC_ERR_INVALID_ARG; }
  try {
    HIP_TRY(hipSetDevice(b->device));
    tbc_batch::InputSlot& sl = b->in_slots[slot];
    if (sl.busy) { set_error("tbc_batch_submit_input: slot %u is still being copied (map it again before it is re-filled)", slot); return TBC_ERR_INVALID_ARG; }
    tbc_batch_input in;
    slot_pointers(b, sl.mem, &in);
    if (in.op_off[0] != 0) { set_error("op_off[0] must be 0"); return TBC_ERR_INVALID_ARG; }
    uint32_t max_slots = 1;
    uint64_t longest = 0;
    for (uint32_t h = 0; h < n_hist; h++) {
      if (in.op_off[h + 1] < in.op_off[h] || in.op_off[h + 1] - in.op_off[h] > 0x7FFFFFFFull) { set_error("history %u: bad op_off", h); return TBC_ERR_INVALID_ARG; }
      max_slots = std::max(max_slots, in.n_process[h]);
      longest = std::max<uint64_t>(longest, in.op_off[h + 1] - in.op_off[h]);
    }
    const uint64_t T = in.op_off[n_hist];
    if (T > b->in_ops_cap) { set_error("tbc_batch_submit_input: %llu ops, a slot holds %llu", (unsigned long long)T, (unsigned long long)b->in_ops_cap); return TBC_ERR_INVALID_ARG; }
    if (T > 0xFFFFFFFFull) { set_error("tbc_batch_submit_input: more than 2^32 - 1 ops"); return TBC_ERR_INVALID_ARG; }
    tbc_pending_input P;
    P.slot = slot; P.stage = (uint32_t)(b->in_seq & 1u); P.n_hist = n_hist; P.total_ops = T;
    tbc_status st;
    std::vector<uint32_t> count_slots;
    if (b->count_form) {          // the classes of crashed calls and the re-used process slots, planned here on the host
      if (!sl.planned) HIP_TRY(hipHostMalloc((void**)&sl.planned, (size_t)b->in_ops_cap * 4, hipHostMallocDefault));
      if ((st = plan_count_input(b, in, sl.planned, n_hist, P, count_slots)) != TBC_OK) return st;
      max_slots = 1;
      for (uint32_t h = 0; h < n_hist; h++) max_slots = std::max(max_slots, count_slots[h]);
    }
    if (max_slots > 64u * b->mask_words || max_slots > 4096u) {
      set_error("the input has a history of %u process slots, the batch's kernels were chosen for at most %u: destroy and create", max_slots, 64u * b->mask_words);
      return TBC_ERR_UNSUPPORTED;
    }
    st = layout_histories(b, n_hist, in.op_off, in.n_events, b->count_form ? count_slots.data() : in.n_process, nullptr, nullptr, true, P.hist, P.bh, P.tot,
                          b->count_form ? &P.count_hist : nullptr);
    if (st != TBC_OK) return st;
    P.max_ops = P.tot.max_ops;
    if (b->lanes && (longest >= kNarrowMaxOps || P.tot.boff_n >= (1ull << 32) || look_words(T, b->in_hist_cap, b->mask_words) >= (1ull << 32))) {
      set_error("the input exceeds what several histories per wavefront address (2^24 - 16 ops a history, 2^32 fronts a batch): destroy and create");
      return TBC_ERR_UNSUPPORTED;
    }
    DevBuf<uint32_t>& stage = b->d_stage[P.stage];
    if (!stage.p && (st = stage.alloc(3 * b->in_ops_cap)) != TBC_OK) return st;
    hipStream_t sc = b->stream_copy;
    if (b->stage_used[P.stage]) HIP_TRY(hipStreamWaitEvent(sc, b->ev_unpacked[P.stage], 0));      // (its previous occupant has been unpacked)
    b->stage_used[P.stage] = true;
    HIP_TRY(hipEventRecord(b->ev_copy[P.stage][0], sc));
    if (T) {
      HIP_TRY(hipMemcpyAsync(stage.p, b->count_form ? sl.planned : in.word, T * 4, hipMemcpyHostToDevice, sc));
      HIP_TRY(hipMemcpyAsync(stage.p + b->in_ops_cap, in.inv_pos, T * 4, hipMemcpyHostToDevice, sc));
      HIP_TRY(hipMemcpyAsync(stage.p + 2 * b->in_ops_cap, in.ret_pos, T * 4, hipMemcpyHostToDevice, sc));
    }
    HIP_TRY(hipEventRecord(b->ev_copy[P.stage][1], sc));
    HIP_TRY(hipEventRecord(b->ev_stage[P.stage], sc));
    HIP_TRY(hipEventRecord(sl.copied, sc));
    sl.busy = true;
    b->in_seq++;
    b->pending.push_back(std::move(P));
    return TBC_OK;
  } catch (const std::bad_alloc&) { set_error("host allocation failed"); return TBC_ERR_OOM; }
  catch (...) { set_error("unexpected exception"); return TBC_ERR_HIP; }
}

// the six op columns of a tbc_batch_desc -> the wire columns of the batch's next slot (0 and 1, in turn) -> submitted
tbc_status tbc_batch_reload(tbc_batch* b, const tbc_batch_desc* desc) {
  if (!b || !desc || !desc->op_off || !desc->n_events || !desc->n_process || desc->n_hist == 0) { set_error("tbc_batch_reload: null or empty argument"); return TBC_ERR_INVALID_ARG; }
  const tbc_ops& c = desc->cols;
  if (c.n && (!c.f || !c.a || !c.b || !c.process || !c.inv_pos || !c.ret_pos)) { set_error("tbc_batch_reload: null op column"); return TBC_ERR_INVALID_ARG; }
  try {
    tbc_batch_input in;
    const uint32_t slot = b->reload_slot & 1u;
    tbc_status s = tbc_batch_map_input(b, slot, &in);
    if (s != TBC_OK) return s;
    const uint32_t nh = desc->n_hist;
    if (nh > in.n_hist_cap) { set_error("tbc_batch_reload: %u histories, the batch holds %u", nh, in.n_hist_cap); return TBC_ERR_INVALID_ARG; }
    const uint64_t T = desc->op_off[nh];
    if (T != c.n) { set_error("op_off[n_hist] (%llu) != cols.n (%u)", (unsigned long long)T, c.n); return TBC_ERR_INVALID_ARG; }
    if (T > in.ops_cap) { set_error("tbc_batch_reload: %llu ops, the batch's slots hold %llu (the first input's and an eighth): destroy and create", (unsigned long long)T, (unsigned long long)in.ops_cap); return TBC_ERR_UNSUPPORTED; }
    std::memcpy(in.op_off, desc->op_off, (size_t)(nh + 1) * 8);
    std::memcpy(in.n_events, desc->n_events, (size_t)nh * 4);
    std::memcpy(in.n_process, desc->n_process, (size_t)nh * 4);
    const unsigned nt = T > (1ull << 22) ? 8u : 1u;
    std::vector<uint8_t> bad(nt, 0);
    const auto work = [&](unsigned t) {
      const uint64_t lo = T * t / nt, hi = T * (t + 1) / nt;
      uint8_t bd = 0;
      for (uint64_t i = lo; i < hi; i++) {
        const uint32_t f = c.f[i];
        const int32_t a = c.a[i], bb = c.b[i], p = c.process[i];
        const bool a_ok = a == TBC_NIL || (a >= 0 && a <= 254), b_ok = bb == TBC_NIL || (bb >= 0 && bb <= 254);
        if (f > 15u || !a_ok || (f == TBC_F_CAS && !b_ok) || p < 0 || p > 4095) { bd = 1; continue; }
        const uint32_t a8 = a == TBC_NIL ? TBC_WIRE_NIL : (uint32_t)a, b8 = !b_ok ? 0u : (bb == TBC_NIL ? TBC_WIRE_NIL : (uint32_t)bb);
        in.word[i] = TBC_WIRE_WORD(f, a8, b8, (uint32_t)p);
      }
      bad[t] = bd;
    };
    if (nt == 1) work(0);
    else {
      std::vector<std::thread> th;
      for (unsigned t = 0; t < nt; t++) th.emplace_back(work, t);
      for (auto& x : th) x.join();
    }
    for (uint8_t x : bad) if (x) { set_error("tbc_batch_reload: the wire format holds :f 0..15, values 0..254 or nil and processes 0..4095: destroy and create"); return TBC_ERR_UNSUPPORTED; }
    std::memcpy(in.inv_pos, c.inv_pos, (size_t)T * 4);
    std::memcpy(in.ret_pos, c.ret_pos, (size_t)T * 4);
    s = tbc_batch_submit_input(b, slot, nh);
    if (s == TBC_OK) b->reload_slot++;
    return s;
  } catch (const std::bad_alloc&) { set_error("host allocation failed"); return TBC_ERR_OOM; }
  catch (...) { set_error("unexpected exception"); return TBC_ERR_HIP; }
}

tbc_status tbc_batch_input_info(const tbc_batch* b, tbc_input_info* out) {
  if (!b || !out) { set_error("tbc_batch_input_info: null argument"); return TBC_ERR_INVALID_ARG; }
  std::memset(out, 0, sizeof *out);
  out->n_hist = b->n_hist; out->pending = (uint32_t)b->pending.size();
  out->total_ops = b->total_ops;
  out->bytes_copied = b->in_bytes_copied; out->ns_copy = b->in_copy_ns;
  out->inputs_consumed = b->inputs_consumed; out->lists_regrown = b->lists_regrown;
  out->n_hist_cap = b->in_hist_cap; out->ops_cap = b->in_ops_cap;
  return TBC_OK;
}

}  // extern "C"
