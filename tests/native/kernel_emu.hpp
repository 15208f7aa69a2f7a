// TEST-ONLY: runs gatekeeper_amd/csrc/kernel_body.inc -- the HIP source of the dominant kernel, unchanged -- on the CPU, so
// that its STRUCTURE (persistent workgroups, chunk lists, the double-buffered list staging, barriers, wave collectives,
// the LDS result words, multi-pass groups) can be checked in the GPU-less build container against the per-review
// evaluation of hostemu.cpp.  One fiber (ucontext) per GPU thread; a workgroup's fibers run until they reach a barrier
// or a wave collective (__ballot / __shfl / __shfl_xor), which complete when every live lane of the workgroup / wave
// has arrived -- a lane waiting at a barrier while its wave is in a collective is reported as a deadlock, as are
// mismatched collectives.  Never part of the product library.
#pragma once
#include <cstdlib>
#include <ucontext.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <stdexcept>
#include <string>
#include <vector>

namespace gkemu {

struct Dim3 { unsigned x = 0, y = 0, z = 0; };
enum FiberState { F_RUN = 0, F_BARRIER = 1, F_COLL = 2, F_DONE = 3 };
enum CollKind { C_BALLOT = 1, C_SHFL = 2 };

struct Fiber {
  ucontext_t ctx;
  std::vector<char> stack;
  int state = F_DONE, kind = 0;
  unsigned tid = 0;
  uint64_t arg = 0, res = 0;
};

struct State {
  std::vector<Fiber> fibers;
  ucontext_t sched;
  Fiber* cur = nullptr;
  Dim3 block_idx, grid_dim, block_dim;
  std::vector<uint32_t> dyn;
  std::function<void()> body;
  unsigned long long clock = 0;
};
inline State& st() { static State s; return s; }

inline void yield_() { State& s = st(); swapcontext(&s.cur->ctx, &s.sched); }
inline void trampoline() { State& s = st(); s.body(); s.cur->state = F_DONE; }   // returns to uc_link = the scheduler

inline void syncthreads() { st().cur->state = F_BARRIER; yield_(); }
inline uint64_t collective(int kind, uint64_t arg) {
  Fiber* f = st().cur;
  f->state = F_COLL; f->kind = kind; f->arg = arg;
  yield_();
  return f->res;
}
inline unsigned long long ballot(bool p) { return collective(C_BALLOT, p ? 1 : 0); }
inline int shfl(int v, int src) { return (int)(uint32_t)collective(C_SHFL, (uint64_t)(uint32_t)v | ((uint64_t)(uint32_t)(src & 63) << 32)); }
inline int shfl_xor(int v, int m) { return shfl(v, (int)((st().cur->tid & 63u) ^ (unsigned)m)); }
inline uint32_t* dyn_lds() { return st().dyn.data(); }
inline unsigned long long clock_() { return ++st().clock; }

inline void run_block(unsigned block) {
  State& s = st();
  if (s.fibers.size() < block) s.fibers.resize(block);
  for (unsigned t = 0; t < block; t++) {
    Fiber& f = s.fibers[t];
    if (f.stack.empty()) f.stack.resize(256 * 1024);
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack.data();
    f.ctx.uc_stack.ss_size = f.stack.size();
    f.ctx.uc_link = &s.sched;
    makecontext(&f.ctx, (void (*)())trampoline, 0);
    f.state = F_RUN; f.tid = t;
  }
  for (;;) {
    bool progressed = false, any_live = false;
    // (GK_EMU_REVERSE: the threads take their turns from the last to the first -- rows of one review that the device's waves process
    //  concurrently reach the accumulators in the opposite order: an order-dependent pair of phase-1 operations shows up as a difference)
    static const bool reverse = getenv("GK_EMU_REVERSE") != nullptr;
    for (unsigned t0 = 0; t0 < block; t0++) {
      const unsigned t = reverse ? block - 1u - t0 : t0;
      Fiber& f = s.fibers[t];
      if (f.state != F_RUN) continue;
      s.cur = &f;
      swapcontext(&s.sched, &f.ctx);
      progressed = true;
    }
    for (unsigned w = 0; w * 64 < block; w++) {   // wave collectives
      unsigned live = 0, waiting = 0;
      int kind = 0;
      bool same = true;
      for (unsigned l = 0; l < 64 && w * 64 + l < block; l++) {
        Fiber& f = s.fibers[w * 64 + l];
        if (f.state == F_DONE) continue;
        live++;
        if (f.state == F_COLL) { waiting++; if (kind == 0) kind = f.kind; else if (kind != f.kind) same = false; }
      }
      if (!live || waiting != live) continue;
      if (!same) throw std::runtime_error("kernel_emu: the lanes of a wave wait in different collectives");
      unsigned long long mask = 0;
      for (unsigned l = 0; l < 64 && w * 64 + l < block; l++) { Fiber& f = s.fibers[w * 64 + l]; if (f.state == F_COLL && f.arg) mask |= 1ull << l; }
      for (unsigned l = 0; l < 64 && w * 64 + l < block; l++) {
        Fiber& f = s.fibers[w * 64 + l];
        if (f.state != F_COLL) continue;
        if (kind == C_BALLOT) f.res = mask;
        else {
          const unsigned src = (unsigned)(f.arg >> 32);
          const Fiber& g = s.fibers[w * 64 + src];
          f.res = (w * 64 + src < block && g.state == F_COLL) ? (uint32_t)g.arg : (uint32_t)f.arg;
        }
      }
      for (unsigned l = 0; l < 64 && w * 64 + l < block; l++) { Fiber& f = s.fibers[w * 64 + l]; if (f.state == F_COLL) f.state = F_RUN; }
      progressed = true;
    }
    unsigned live = 0, at_barrier = 0;
    for (unsigned t = 0; t < block; t++) { Fiber& f = s.fibers[t]; if (f.state != F_DONE) { live++; any_live = true; if (f.state == F_BARRIER) at_barrier++; } }
    if (live && at_barrier == live) { for (unsigned t = 0; t < block; t++) if (s.fibers[t].state == F_BARRIER) s.fibers[t].state = F_RUN; progressed = true; }
    if (!any_live) break;
    if (!progressed) throw std::runtime_error("kernel_emu: deadlock (threads wait at a barrier / collective the others never reach)");
  }
}

// run `body` as a kernel of grid x block threads with dyn_bytes of dynamic LDS (filled with garbage per workgroup: LDS is
// not zero-initialised on the device either)
inline void launch(unsigned grid, unsigned block, size_t dyn_bytes, std::function<void()> body) {
  State& s = st();
  s.body = std::move(body);
  s.grid_dim.x = grid; s.block_dim.x = block;
  for (unsigned b = 0; b < grid; b++) {
    s.block_idx.x = b;
    s.dyn.assign(dyn_bytes / 4 + 16, 0xCDCDCDCDu);
    run_block(block);
  }
  s.body = nullptr;   // (the state may be shared with another shared object that outlives the caller's)
}

struct Tid { operator Dim3() const { return Dim3{st().cur->tid, 0, 0}; } };

}  // namespace gkemu

// ---- what the device compiler provides, for g++
struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; };
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
struct gk_u32x4 { uint32_t x, y, z, w; };
#define __global__
#define __device__
#define __shared__ static
#define __launch_bounds__(...)
#define threadIdx (gkemu::Dim3{gkemu::st().cur->tid, 0, 0})
#define blockIdx (gkemu::st().block_idx)
#define gridDim (gkemu::st().grid_dim)
#define __syncthreads() gkemu::syncthreads()
#define __ballot(p) gkemu::ballot((p) != 0)
#define __shfl(v, s) gkemu::shfl((v), (s))
#define __shfl_xor(v, m) gkemu::shfl_xor((v), (m))
#define __popcll(x) __builtin_popcountll(x)
#define __builtin_readcyclecounter() gkemu::clock_()
#define __builtin_amdgcn_s_sleep(n) do { } while (0)
#define __builtin_amdgcn_s_setprio(n) do { } while (0)
#define __builtin_amdgcn_mbcnt_lo(m, v) ((v) + ((gkemu::st().cur->tid & 63u) < 32u ? (gkemu::st().cur->tid & 63u) : 32u))
#define __builtin_amdgcn_mbcnt_hi(m, v) ((v) + ((gkemu::st().cur->tid & 63u) < 32u ? 0u : (gkemu::st().cur->tid & 63u) - 32u))
template <class T> static inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
template <class T> static inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
static inline uint32_t min(uint32_t a, uint32_t b) { return a < b ? a : b; }
#define GK_DYN_LDS(name) uint32_t* name = gkemu::dyn_lds()
#define GK_OPAQUE_V2(a, b) do { } while (0)
#define GK_OPAQUE_V1(a) do { } while (0)
#define GK_OPAQUE_S1(a) do { } while (0)
#define GK_OPAQUE() do { } while (0)
#define GK_LDS_ADD(p, v) ((void)(*(p) += (v)))
#define GK_READLANE(v, k) ((uint32_t)gkemu::shfl((int)(v), (int)(k)))
// LDS-DMA of the device build (kernel_body.inc GK_LDS_DMA16): here every lane copies its 16 bytes at once -- a lane only ever
// reads back what it requested itself, behind GK_WAIT_VM
typedef uintptr_t gk_ldsaddr_t;
#define GK_LDS_ADDR(p) ((gk_ldsaddr_t)(uintptr_t)(p))
#define GK_LDS_DMA16(gptr, ldsaddr) memcpy(reinterpret_cast<unsigned char*>(ldsaddr) + (gkemu::st().cur->tid & 63u) * 16u, (gptr), 16)
#define GK_LDS_DMA4(gptr, ldsaddr) memcpy(reinterpret_cast<unsigned char*>(ldsaddr) + (gkemu::st().cur->tid & 63u) * 4u, (gptr), 4)
#define GK_WAIT_VM(n) do { } while (0)
