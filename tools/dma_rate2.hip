// GPU, standalone (hipcc --offload-arch=gfx950 -O3 tools/dma_rate2.hip -o tools/dma_rate2.bin): LDS-DMA rate of one CU from an L2-resident buffer as a
// function of the access pattern -- P bytes (32 / 64 / 128 / 256) taken at byte offset O of rows S bytes apart, 1 KiB per wave-instruction, the rows of
// one instruction consecutive in memory (the W direction of a halo patch).  The convolution's patch / panel layouts are priced against this table.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int PIECES, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void rate_kernel(char* buf, long long window, int iters, int P, int S, int O, unsigned long long* cycles) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* base = buf;  // every work-group reads the SAME span (L2-resident like a weight panel / a neighbour's halo), its pieces rotated by the block index
  char* src[PIECES];
  unsigned dst[PIECES];
  const int lpr = P / 16, rpp = 1024 / P;  // lanes per row, rows per piece
#pragma unroll
  for (int j = 0; j < PIECES; ++j) {
    const int piece = (wave + WAVES * j + (int)blockIdx.x * 5) % (PIECES * WAVES);
    const long long off = (long long)(piece * rpp + lane / lpr) * S + O + (lane % lpr) * 16;
    src[j] = base + off;
    dst[j] = lds0 + (unsigned)((piece & 15) * 1024);
  }
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < PIECES; ++j) dma16(src[j], __builtin_amdgcn_readfirstlane(dst[j]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int PIECES, int WAVES>
static double run(char* buf, long long window, int wgs, int P, int S, int O, unsigned long long* dcyc) {
  const int iters = 200;
  hipFuncSetAttribute(reinterpret_cast<const void*>(rate_kernel<PIECES, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  double out = 0;
  for (int rep = 0; rep < 2; ++rep) {
    rate_kernel<PIECES, WAVES><<<wgs, WAVES * 64, 16 * 1024>>>(buf, window, iters, P, S, O, dcyc);
    hipDeviceSynchronize();
    std::vector<unsigned long long> c(wgs);
    hipMemcpy(c.data(), dcyc, wgs * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto v : c) mean += (double)v;
    mean /= wgs;
    out = (wgs / 256) * (double)iters * PIECES * WAVES * 1024 / mean;
  }
  return out;
}

int main() {
  const long long window = 0;
  char* buf = nullptr;
  unsigned long long* dcyc = nullptr;
  hipMalloc(&buf, 4 << 20);
  hipMemset(buf, 1, 4 << 20);
  hipMalloc(&dcyc, 768 * sizeof(unsigned long long));
  std::printf("B/clk per CU; 4 waves x 6 pieces per work-group; columns: work-groups per CU 1 / 2 / 3\n");
  const int Ps[] = {32, 64, 128, 256};
  const int Ss[] = {64, 128, 192, 256, 320, 384, 512, 768, 1024};
  for (int P : Ps)
    for (int S : Ss) {
      if (S < P) continue;
      for (int O : {0, 64}) {
        if (O + P > S || (O && P > 64)) continue;
        std::printf("P %3d of S %4d at +%2d :", P, S, O);
        for (int w : {256, 512, 768}) std::printf(" %6.1f", run<6, 4>(buf, window, w, P, S, O, dcyc));
        std::printf("\n");
      }
    }
  std::printf("contiguous 1 KiB          :");
  for (int w : {256, 512, 768}) std::printf(" %6.1f", run<6, 4>(buf, window, w, 1024, 1024, 0, dcyc));
  std::printf("\n");
  return 0;
}
