// batch_common.hip -- what every host unit of the library shares at run time (tbc_batch.h): the pool of persistent device contexts
// behind tbc_check, the arena guard (TBC_GUARD=1), the debug words (TBC_DEBUG=1), the clock.
#include "tbc_batch.h"

namespace tbc {

thread_local Ctx* t_ctx = nullptr;
thread_local const void* t_guard_owner = nullptr;
thread_local size_t t_guard_nth = 0;

namespace {
std::mutex g_ctx_mu;
std::vector<Ctx*> g_ctx_free;
struct GuardRec { const char* at; size_t serial; size_t nth; const void* owner; };
std::mutex g_guard_mu;
std::vector<GuardRec> g_guards;
size_t g_guard_serial = 0;
uint32_t* g_dbg = nullptr;
}  // namespace

uint64_t now_ns() {
  return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(
             std::chrono::steady_clock::now().time_since_epoch()).count();
}

bool device_is_gfx950(int dev) {
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, dev) != hipSuccess) return false;
  return std::strncmp(p.gcnArchName, "gfx950", 6) == 0;
}

bool guard_on() { static const bool on = [] { const char* e = std::getenv("TBC_GUARD"); return e && e[0] == '1'; }(); return on; }
void guard_add(const void* arena_end) {
  static const std::vector<unsigned char> poison(kGuardBytes, 0xA5);
  (void)hipMemcpy(const_cast<void*>(arena_end), poison.data(), kGuardBytes, hipMemcpyHostToDevice);      // (synchronous: the bytes are there before anything is launched)
  std::lock_guard<std::mutex> lk(g_guard_mu);
  g_guards.push_back(GuardRec{(const char*)arena_end, g_guard_serial++, t_guard_nth++, t_guard_owner});
}
void guard_remove(const void* arena_end) {
  std::lock_guard<std::mutex> lk(g_guard_mu);
  for (size_t i = 0; i < g_guards.size(); i++) if (g_guards[i].at == (const char*)arena_end) { g_guards.erase(g_guards.begin() + (long)i); return; }
}
// returns the number of arenas whose guard bytes were overwritten (after the caller's stream is idle)
size_t guard_check(const char* when, const void* owner) {
  std::vector<GuardRec> live;
  { std::lock_guard<std::mutex> lk(g_guard_mu); for (const GuardRec& g : g_guards) if (g.owner == owner) live.push_back(g); }
  size_t bad = 0;
  unsigned char buf[kGuardBytes];
  for (const GuardRec& g : live) {
    if (hipMemcpy(buf, g.at, kGuardBytes, hipMemcpyDeviceToHost) != hipSuccess) continue;
    for (size_t i = 0; i < kGuardBytes; i++) if (buf[i] != 0xA5) {
      std::fprintf(stderr, "[tbc guard] %s: arena #%zu (the batch's %zu-th, ends at %p) overrun: byte +%zu = 0x%02x\n", when, g.serial, g.nth, (const void*)g.at, i, buf[i]);
      bad++;
      break;
    }
  }
  return bad;
}

Ctx* ctx_acquire(int device) {
  {
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    for (size_t i = 0; i < g_ctx_free.size(); i++) if (g_ctx_free[i]->device == device) {
      Ctx* c = g_ctx_free[i]; g_ctx_free.erase(g_ctx_free.begin() + (long)i); return c;
    }
  }
  Ctx* c = new (std::nothrow) Ctx();
  if (!c) return nullptr;
  c->device = device;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return nullptr; }
  for (auto& e : c->ev) if (hipEventCreate(&e) != hipSuccess) { delete c; return nullptr; }
  return c;
}
void ctx_release(Ctx* c) {
  // size the slab for the next call of this kind (1.25 x what this one asked for), within reason
  if (c->wanted > c->cap && c->wanted < (8ull << 30)) {
    if (c->slab) (void)hipFree(c->slab);
    c->slab = nullptr; c->cap = 0;
    const size_t want = c->wanted + c->wanted / 4;
    if (hipMalloc((void**)&c->slab, want) == hipSuccess) c->cap = want;
  }
  c->used = 0; c->wanted = 0;
  if (c->pin_wanted > c->pin_cap && c->pin_wanted < (1ull << 30)) {
    if (c->pin) { (void)hipHostUnregister(c->pin); std::free(c->pin); }
    c->pin = nullptr; c->pin_cap = 0;
    const size_t want = (c->pin_wanted + c->pin_wanted / 4 + 4095) & ~(size_t)4095;
    // (ordinary cached memory, registered: the composition reads the 0.6 MB relation table right after the copy, and through
    // hipHostMalloc's mapping -- coherent or "non-coherent" alike -- that took 78 us instead of 27; the stream synchronize before it
    // makes the copy visible)
    void* mem = std::aligned_alloc(4096, want);
    if (mem && hipHostRegister(mem, want, hipHostRegisterDefault) == hipSuccess) { c->pin = (char*)mem; c->pin_cap = want; }
    else std::free(mem);
  }
  c->pin_used = 0; c->pin_wanted = 0;
  std::lock_guard<std::mutex> lk(g_ctx_mu);
  g_ctx_free.push_back(c);
}

// TBC_DEBUG=1: kernels mirror their progress into host-mapped words so a hang can be diagnosed
// from another thread (tbc_debug_peek) while the call is still blocked.
uint32_t* debug_words() {
  static bool tried = false;
  if (!tried) {
    tried = true;
    const char* e = std::getenv("TBC_DEBUG");
    if (e && e[0] == '1') {
      void* p = nullptr;
      if (hipHostMalloc(&p, 64 * sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
        std::memset(p, 0, 64 * sizeof(uint32_t));
        g_dbg = (uint32_t*)p;
      }
    }
  }
  return g_dbg;
}

}  // namespace tbc

extern "C" int tbc_debug_peek(uint32_t* out, uint32_t n) {
  if (!tbc::g_dbg || !out) return 0;
  for (uint32_t i = 0; i < n && i < 64; i++) out[i] = ((volatile uint32_t*)tbc::g_dbg)[i];
  return 1;
}
