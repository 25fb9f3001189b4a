// GPU, standalone (hipcc --offload-arch=gfx950 -O3 tools/xcd_barrier_probe.hip -o tools/xcd_barrier_probe.bin): what a PHASE of a persistent decode kernel costs
// when the kernel is confined to ONE XCD -- barrier + a small vector exchanged between all participants -- against the same phase across the whole chip.
// ("C5 persistent decode: measured, not argued".)  256 work-groups of 256 threads with 96 KiB of LDS (one per CU) are launched; every
// work-group takes a census ticket; SCOPE 0: only those whose XCC_ID is 0 stay (32 on MI355X: they share one L2), SCOPE 1: all stay.  Then `phases` times:
// every participant publishes 64 floats, arrives at the barrier (one agent-scope atomic add + a spin on the counter), and reads ALL participants' floats.
// MODE 0: agent-scope release / acquire fences around plain stores / loads (the memory model's way: L2 write-back + invalidate, needed ACROSS XCDs);
// MODE 1: relaxed agent-scope atomic stores / loads (sc1: the per-CU L1 is bypassed, the XCD's own L2 is the meeting point) and NO fences -- sufficient
// inside one XCD only.  Every spin is bounded; the checksum of what was read is verified on the device (stale reads are counted, not assumed away).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

struct Ctl { unsigned total, members, arrive, errors, timeouts, pad[11]; };

template <int MODE, int SCOPE>
__global__ __launch_bounds__(256) void probe(Ctl* ctl, float* buf, int phases, unsigned long long* cycles) {
  extern __shared__ char smem[];
  __shared__ unsigned s_ticket, s_members;
  const int t = threadIdx.x;
  if (t == 0) {
    smem[0] = 0;
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15;  // HW_REG_XCC_ID
    const bool stay = SCOPE == 1 || xcc == 0;
    s_ticket = stay ? __hip_atomic_fetch_add(&ctl->members, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffffffu;
    __hip_atomic_fetch_add(&ctl->total, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    if (stay) {  // census: wait (bounded) until every work-group of the grid has reported, then the member count is final
      unsigned spins = 0;
      while (__hip_atomic_load(&ctl->total, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x && ++spins < (1u << 24)) __builtin_amdgcn_s_sleep(2);
      if (spins >= (1u << 24)) __hip_atomic_fetch_add(&ctl->timeouts, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_members = __hip_atomic_load(&ctl->members, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  const unsigned ticket = s_ticket, P = s_members;
  if (ticket == 0xffffffffu) return;
  const unsigned long long c0 = __builtin_readcyclecounter();
  unsigned bad = 0;
  for (int ph = 0; ph < phases; ++ph) {
    float* cur = buf + (size_t)(ph & 1) * 256 * 64;
    if (t < 64) {
      const float v = (float)((ph * 131 + ticket * 7 + t) & 1023);
      if (MODE == 0) cur[ticket * 64 + t] = v;
      else __hip_atomic_store(&cur[ticket * 64 + t], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (MODE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores have reached L2
    __syncthreads();
    if (t == 0) {
      if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_fetch_add(&ctl->arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = (unsigned)(ph + 1) * P;
      unsigned spins = 0;
      while (__hip_atomic_load(&ctl->arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want && ++spins < (1u << 22)) {}
      if (spins >= (1u << 22)) __hip_atomic_fetch_add(&ctl->timeouts, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    // every thread reads P * 64 / 256 values
    for (unsigned i = t; i < P * 64; i += 256) {
      const float got = MODE == 0 ? cur[i] : __hip_atomic_load(&cur[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const float want = (float)((ph * 131 + (i >> 6) * 7 + (i & 63)) & 1023);
      bad += got != want;
    }
    __syncthreads();  // (the other half of the double buffer is written next phase: everyone has read this one by the barrier after next)
  }
  if (bad) __hip_atomic_fetch_add(&ctl->errors, bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (t == 0 && ticket == 0) cycles[0] = __builtin_readcyclecounter() - c0;
}

template <int MODE, int SCOPE>
static void run(const char* name, int phases) {
  Ctl* ctl; float* buf; unsigned long long* cyc;
  hipMalloc(&ctl, sizeof(Ctl)); hipMalloc(&buf, 2 * 256 * 64 * 4); hipMalloc(&cyc, 8);
  hipFuncSetAttribute((const void*)probe<MODE, SCOPE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f; Ctl h{};
  for (int rep = 0; rep < 3; ++rep) {
    hipMemset(ctl, 0, sizeof(Ctl)); hipMemset(buf, 0, 2 * 256 * 64 * 4);
    hipEventRecord(e0);
    probe<MODE, SCOPE><<<256, 256, 96 * 1024>>>(ctl, buf, phases, cyc);
    hipEventRecord(e1);
    if (hipEventSynchronize(e1) != hipSuccess) { printf("%s: launch failed\n", name); return; }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
    hipMemcpy(&h, ctl, sizeof(Ctl), hipMemcpyDeviceToHost);
  }
  printf("%-86s members %3u  %7.3f us per phase  (stale reads %u, spin time-outs %u)\n", name, h.members, best * 1e3 / phases, h.errors, h.timeouts);
}

int main(int argc, char** argv) {
  const int phases = argc > 1 ? atoi(argv[1]) : 2000;
  printf("phase = publish 64 floats per participant + barrier (agent-scope atomic add + spin) + read all participants' floats; %d phases\n", phases);
  run<1, 0>("ONE XCD, relaxed agent-scope (sc1) stores / loads, no fences", phases);
  run<0, 0>("ONE XCD, plain stores / loads inside agent-scope release / acquire fences", phases);
  run<0, 1>("whole chip (256 work-groups), plain stores / loads inside agent-scope release / acquire fences", phases);
  run<1, 1>("whole chip, relaxed agent-scope (sc1) stores / loads, no fences (NOT sufficient across XCDs: counts the stale reads)", phases);
  return 0;
}
