// GPU, standalone: calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on KNOWN byte counts in the access patterns of the convolution kernels
// (MI355X_MICROARCH.md calibrates the x2 correction of FETCH_SIZE only for wide coalesced streaming reads and says
// "other access widths are uncalibrated ... calibrate on a known byte count in your own access pattern").
//   hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o /tmp/fetch_calib
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -o c -- /tmp/fetch_calib      (and a second pass with --pmc WRITE_SIZE)
// Every kernel touches a buffer much larger than the 256 MiB Infinity Cache exactly once, so "bytes requested" = "bytes from HBM":
//   calib_stream_read   16 B per lane, fully coalesced (the guide's calibrated case)                                  reads  BYTES
//   calib_patch64_read  64 B of every 128-B row (4 lanes x 16 B per row: the halo-patch pattern of a 64-channel bf16 tensor, LDS-DMA)  reads  BYTES / 2
//   calib_patch64_of256 64 B of every 256-B row (128-channel tensor)                                                  reads  BYTES / 4
//   calib_row128_read   whole 128-B rows, 8 lanes x 16 B (the epilogue's residual read)                               reads  BYTES
//   calib_row128_write  whole 128-B rows, 8 lanes x 16 B (the epilogue's output store)                                writes BYTES
// tools/fetch_calib.sh prints counter / known-bytes per kernel = the factor to DIVIDE a raw counter by (1 / correction).
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

__global__ __launch_bounds__(256) void calib_stream_read(const uint4* __restrict__ buf, long long n16, unsigned* sink) {
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long long)gridDim.x * blockDim.x) {
    const uint4 v = buf[i];
    acc.x ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc.x == 0x12345678u) sink[0] = acc.x;
}

// rows of ROWB bytes; each wave-instruction moves 16 rows x 64 B (lane -> row lane / 4, 16-byte slot lane % 4) by LDS-DMA, like the patch staging
template <int ROWB>
__global__ __launch_bounds__(256) void calib_patch64_read(const char* __restrict__ buf, long long rows, unsigned* sink) {
  __shared__ __attribute__((aligned(1024))) char smem[4 * 1024];
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long groups = rows / 16;
  for (long long g = (long long)blockIdx.x * 4 + wave; g < groups; g += (long long)gridDim.x * 4) {
    const char* src = buf + (g * 16 + (lane >> 2)) * ROWB + (lane & 3) * 16;
    dma16(src, lds0 + (unsigned)wave * 1024);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if (*reinterpret_cast<volatile unsigned*>(smem + threadIdx.x * 4) == 0x12345678u) sink[0] = 1;
}

__global__ __launch_bounds__(256) void calib_row128_read(const char* __restrict__ buf, long long rows, unsigned* sink) {
  uint4 acc = make_uint4(0, 0, 0, 0);
  const int lane = threadIdx.x & 63;
  const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  for (long long g = wave; g < rows / 8; g += nwaves) {  // 8 rows x 128 B per wave-instruction
    const uint4 v = *reinterpret_cast<const uint4*>(buf + (g * 8 + (lane >> 3)) * 128 + (lane & 7) * 16);
    acc.x ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc.x == 0x12345678u) sink[0] = acc.x;
}

__global__ __launch_bounds__(256) void calib_row128_write(char* __restrict__ buf, long long rows) {
  const int lane = threadIdx.x & 63;
  const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  for (long long g = wave; g < rows / 8; g += nwaves)
    *reinterpret_cast<uint4*>(buf + (g * 8 + (lane >> 3)) * 128 + (lane & 7) * 16) = make_uint4((unsigned)g, lane, 3u, 4u);
}

int main() {
  const long long BYTES = 2LL << 30;  // 2 GiB: eight times the Infinity Cache
  char* buf = nullptr;
  unsigned* sink = nullptr;
  CHECK(hipMalloc(&buf, BYTES));
  CHECK(hipMalloc(&sink, 64));
  CHECK(hipMemset(buf, 1, BYTES));
  CHECK(hipDeviceSynchronize());
  const int grid = 256 * 8;
  for (int rep = 0; rep < 3; ++rep) {
    calib_stream_read<<<grid, 256>>>(reinterpret_cast<const uint4*>(buf), BYTES / 16, sink);
    calib_patch64_read<128><<<grid, 256>>>(buf, BYTES / 128, sink);
    calib_patch64_read<256><<<grid, 256>>>(buf, BYTES / 256, sink);
    calib_row128_read<<<grid, 256>>>(buf, BYTES / 128, sink);
    calib_row128_write<<<grid, 256>>>(buf, BYTES / 128);
    CHECK(hipDeviceSynchronize());
  }
  printf("known bytes per launch: calib_stream_read %lld read; calib_patch64_read<128> %lld read; calib_patch64_read<256> %lld read; "
         "calib_row128_read %lld read; calib_row128_write %lld written\n", BYTES, BYTES / 2, BYTES / 4, BYTES, BYTES);
  return 0;
}
