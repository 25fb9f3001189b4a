// GPU, standalone (hipcc --offload-arch=gfx950 -O3 tools/dma_rate.hip -o /tmp/dma_rate && /tmp/dma_rate): what one CU can move per clock
// from an L2-resident buffer with the instructions the convolution kernel uses -- global_load_lds_dwordx4 (LDS-DMA) with contiguous
// 1 KiB pieces and with the patch pattern (64-byte row segments of 128 / 384-byte rows), plain global_load_dwordx4 into registers, and
// global_store_dwordx4 -- at one and two 512-thread work-groups per CU.  The convolution's operand movement is priced against these.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// mode 0: LDS-DMA, contiguous (lane i reads base + 16 i); 1: LDS-DMA, 64-byte segments of rows `rowb` bytes apart (4 lanes per row);
// 2: global_load_dwordx4 into registers, contiguous; 3: global_store_dwordx4, contiguous; 4: as 2 with the row pattern
template <int MODE, int PIECES>
__global__ __launch_bounds__(512, 2) void rate_kernel(char* buf, long long window, int iters, int rowb, unsigned long long* cycles, unsigned* sink) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* base = buf + (long long)blockIdx.x * window;
  uint4 acc = make_uint4(0, 0, 0, 0);
  char* src[PIECES];  // addresses are computed once: the timed loop is the memory instructions only
  unsigned dst[PIECES];
#pragma unroll
  for (int j = 0; j < PIECES; ++j) {
    const int piece = wave + 8 * j;                             // 1 KiB pieces dealt round-robin to the 8 waves, like the kernel's patch
    long long off;
    if (MODE == 1 || MODE == 4) off = ((long long)(piece * 16 + (lane >> 2)) * rowb + (lane & 3) * 16) % window;
    else off = ((long long)piece * 1024 + lane * 16) % window;
    src[j] = base + off;
    dst[j] = lds0 + (unsigned)((piece & 31) * 1024);
  }
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    u32x4 v[PIECES];
#pragma unroll
    for (int j = 0; j < PIECES; ++j) {
      if (MODE <= 1) dma16(src[j], __builtin_amdgcn_readfirstlane(dst[j]));
      else if (MODE == 3) *reinterpret_cast<uint4*>(src[j]) = make_uint4(it, j, lane, wave);
      else v[j] = *reinterpret_cast<const u32x4*>(src[j]);      // all PIECES loads in flight, one wait
    }
    if (MODE == 2 || MODE == 4) {
#pragma unroll
      for (int j = 0; j < PIECES; ++j) acc.x ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  if (acc.x == 0x12345678u) sink[0] = acc.x;
}

template <int MODE, int PIECES>
static void run(const char* name, char* buf, long long window, int wgs, int rowb, unsigned long long* dcyc, unsigned* sink) {
  const int iters = 200, pieces = PIECES;
  hipFuncSetAttribute(reinterpret_cast<const void*>(rate_kernel<MODE, PIECES>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  for (int rep = 0; rep < 2; ++rep) {  // first launch warms the L2
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    rate_kernel<MODE, PIECES><<<wgs, 512, 32 * 1024>>>(buf, window, iters, rowb, dcyc, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    if (rep == 1) {
      std::vector<unsigned long long> c(wgs);
      hipMemcpy(c.data(), dcyc, wgs * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      double mean = 0;
      for (auto v : c) mean += (double)v;
      mean /= wgs;
      const double bytes_wg = (double)iters * pieces * 8 * 1024;
      const int per_cu = wgs / 256;
      std::printf("%-58s %2d WG/CU  %7.1f B/clk per WG  %7.1f B/clk per CU  (%.2f TB/s chip, %.3f ms)\n", name, per_cu, bytes_wg / mean, per_cu * bytes_wg / mean,
                  (double)wgs * bytes_wg / (ms * 1e-3) / 1e12, ms);
    }
  }
}

int main() {
  const long long window = 96 * 1024;  // per work-group: 512 x 96 KiB = 48 MiB = 6 MiB per XCD-L2 share ... keep below with 256 WGs: 3 MiB
  char* buf = nullptr;
  unsigned long long* dcyc = nullptr;
  unsigned* sink = nullptr;
  hipMalloc(&buf, 512 * window + (1 << 20));
  hipMemset(buf, 1, 512 * window + (1 << 20));
  hipMalloc(&dcyc, 512 * sizeof(unsigned long long));
  hipMalloc(&sink, 64);
  for (int wgs : {256, 512}) {
    run<0, 6>("LDS-DMA dwordx4, contiguous 1 KiB pieces, 6 per wave", buf, window, wgs, 0, dcyc, sink);
    run<1, 6>("LDS-DMA dwordx4, 64 B of 128-B rows (C = 64 patch), 6/wave", buf, window, wgs, 128, dcyc, sink);
    run<1, 6>("LDS-DMA dwordx4, 64 B of 384-B rows (C = 192 patch), 6/wave", buf, window, wgs, 384, dcyc, sink);
    run<2, 6>("global_load_dwordx4 -> VGPR, contiguous, 6 per wave", buf, window, wgs, 0, dcyc, sink);
    run<4, 6>("global_load_dwordx4 -> VGPR, 64 B of 128-B rows, 6 per wave", buf, window, wgs, 128, dcyc, sink);
    run<3, 4>("global_store_dwordx4, contiguous, 4 per wave", buf, window, wgs, 0, dcyc, sink);
    run<0, 2>("LDS-DMA dwordx4, contiguous, 2 per wave (a weight panel)", buf, window, wgs, 0, dcyc, sink);
  }
  return 0;
}
