// GPU, standalone (hipcc --offload-arch=gfx950 -O3 tools/mfma_power.hip -o tools/mfma_power.bin): what the chip SUSTAINS on bf16 MFMA streams as a function
// of the operand DATA (all-zero vs random normal bf16) -- the clock is set by the power budget, and switching power is data dependent -- and of
// the operand source (registers only, or re-read from LDS with one ds_read_b128 per MFMA as the convolution's tap loop does).
// Prints TFLOP/s per variant; the convolution kernels' roofline fractions are read against the random-data rows.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// MODE 0: 32x32x16, operands in registers; 1: 32x32x16, 4 ds_read_b128 per 4 MFMAs; 2: 16x16x32 registers; 3: 16x16x32, 8 reads per 16 MFMAs
template <int MODE>
__global__ __launch_bounds__(256) void mfma_kernel(const uint4* __restrict__ src, float* sink, int iters) {
  __shared__ __attribute__((aligned(16))) uint4 lds[4096];  // 64 KiB of operand data
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = src[(blockIdx.x * 4096 + i) & 65535];
  __syncthreads();
  if (MODE <= 1) {
    f32x16_t acc[2][2] = {};
    uint4 a[2], b[2];
    for (int i = 0; i < 2; ++i) { a[i] = lds[lane + 64 * i]; b[i] = lds[lane + 64 * (2 + i)]; }
    for (int it = 0; it < iters; ++it) {
      if (MODE == 1) {
        const int o = (it & 7) * 256 + ((threadIdx.x >> 6) << 11) / 4;
        for (int i = 0; i < 2; ++i) { a[i] = lds[(o + lane + 64 * i) & 4095]; b[i] = lds[(o + lane + 64 * (2 + i)) & 4095]; }
      }
      for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a[i]), __builtin_bit_cast(bf16x8_t, b[j]), acc[i][j], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 12345.678f) sink[0] = s;
  } else {
    f32x4_t acc[4][4] = {};
    uint4 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = lds[lane + 64 * i]; b[i] = lds[lane + 64 * (4 + i)]; }
    for (int it = 0; it < iters; ++it) {
      if (MODE == 3) {
        const int o = (it & 7) * 512;
        for (int i = 0; i < 4; ++i) { a[i] = lds[(o + lane + 64 * i) & 4095]; b[i] = lds[(o + lane + 64 * (4 + i)) & 4095]; }
      }
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a[i]), __builtin_bit_cast(bf16x8_t, b[j]), acc[i][j], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    if (s == 12345.678f) sink[0] = s;
  }
}

// 32x32x16 with an NA x NB block tile per wave, all NA + NB fragments re-read from LDS per NA * NB MFMAs (reads per MFMA = (NA + NB) / (NA NB))
template <int NA, int NB, int WPS>
__global__ __launch_bounds__(256, WPS) void tile_kernel(const uint4* __restrict__ src, float* sink, int iters) {
  __shared__ __attribute__((aligned(16))) uint4 lds[4096];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = src[(blockIdx.x * 4096 + i) & 65535];
  __syncthreads();
  f32x16_t acc[NA][NB] = {};
  uint4 a[NA], b[NB];
  for (int it = 0; it < iters; ++it) {
    const int o = (it & 3) * 1024 + (threadIdx.x >> 6) * 64;
#pragma unroll
    for (int i = 0; i < NA; ++i) a[i] = lds[(o + lane + 256 * i) & 4095];
#pragma unroll
    for (int j = 0; j < NB; ++j) b[j] = lds[(o + 2048 + lane + 256 * j) & 4095];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a[i]), __builtin_bit_cast(bf16x8_t, b[j]), acc[i][j], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NA; ++i) for (int j = 0; j < NB; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  if (s == 12345.678f) sink[0] = s;
}

template <int NA, int NB, int WPS>
static void run_tile(const uint4* src, float* sink, int wgs_per_cu) {
  const int iters = 40000 / (NA * NB), wgs = 256 * wgs_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f, last = 0;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    tile_kernel<NA, NB, WPS><<<wgs, 256>>>(src, sink, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
    last = ms;
  }
  const double flop = (double)wgs * 4 * iters * NA * NB * 32768.0;
  std::printf("32x32x16 tile %d x %d blocks, %d reads per %2d MFMAs (%.3f/MFMA)     %d WG/CU  best %7.1f  last %7.1f TFLOP/s\n", NA, NB, NA + NB, NA * NB,
              (double)(NA + NB) / (NA * NB), wgs_per_cu, flop / (best * 1e-3) / 1e12, flop / (last * 1e-3) / 1e12);
}

template <int MODE>
static void run(const char* name, const uint4* src, float* sink, int wgs_per_cu) {
  const int iters = 20000, wgs = 256 * wgs_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f, last = 0;
  for (int rep = 0; rep < 4; ++rep) {  // the later repetitions are at the sustained (power-limited) clock
    hipEventRecord(e0);
    mfma_kernel<MODE><<<wgs, 256>>>(src, sink, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
    last = ms;
  }
  const double flop = (double)wgs * 4 * iters * (MODE <= 1 ? 4 * 32768.0 : 16 * 16384.0);
  std::printf("%-58s %d WG/CU  best %7.1f  last %7.1f TFLOP/s\n", name, wgs_per_cu, flop / (best * 1e-3) / 1e12, flop / (last * 1e-3) / 1e12);
}

static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }

int main() {
  const size_t n = 65536;  // uint4
  std::vector<unsigned short> h(n * 8);
  uint4* d; float* sink;
  hipMalloc(&d, n * 16); hipMalloc(&sink, 64);
  for (int data = 0; data < 3; ++data) {
    srand(7);
    for (auto& v : h) {
      if (data == 0) v = 0;
      else {
        float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = (rand() + 1.0f) / (RAND_MAX + 2.0f);
        float g = std::sqrt(-2.0f * std::log(u1)) * std::cos(6.2831853f * u2);
        v = f2bf(data == 1 ? g : 0.05f * g);
      }
    }
    hipMemcpy(d, h.data(), n * 16, hipMemcpyHostToDevice);
    std::printf("== operand data: %s\n", data == 0 ? "all zero" : (data == 1 ? "random normal(0, 1) bf16" : "random normal(0, 0.05) bf16"));
    for (int w : {1, 2}) {
      run<0>("32x32x16, operands in registers", d, sink, w);
      run<1>("32x32x16, 4 ds_read_b128 per 4 MFMAs", d, sink, w);
      run<2>("16x16x32, operands in registers", d, sink, w);
      run<3>("16x16x32, 8 ds_read_b128 per 16 MFMAs", d, sink, w);
    }
    for (int w : {1, 2, 3, 4}) run_tile<2, 2, 4>(d, sink, w);
    for (int w : {1, 2}) run_tile<2, 4, 2>(d, sink, w);
    for (int w : {1, 2}) run_tile<4, 2, 2>(d, sink, w);
    run_tile<4, 4, 1>(d, sink, 1);
    run_tile<2, 8, 1>(d, sink, 1);
  }
  return 0;
}
