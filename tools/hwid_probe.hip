// GPU, standalone (hipcc --offload-arch=gfx950 -O3 tools/hwid_probe.hip -o tools/hwid_probe.bin): where the dispatcher puts the work-groups of a
// launch shaped like the cfg 14 convolution (256 threads, 78.5 KiB of LDS: two work-groups per CU) and WHEN it starts them.  Every work-group
// records HW_ID (CU / SE / work-group slot TG_ID), XCC_ID and the shader clock at its start and end, then burns a fixed number of cycles.
// Prints: the TG_ID values seen, how many CUs host which pair of slots, and the distribution of start-time differences between the two
// work-groups that share a CU, round by round -- the lock step the phase offset of conv_dma.hip removes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <map>
#include <algorithm>

__global__ __launch_bounds__(256, 2) void probe(unsigned long long* out, int spin, int skew_wgs, int skew_sleeps) {
  extern __shared__ char smem[];
  const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);    // HW_REG_HW_ID, all 32 bits
  const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);   // HW_REG_XCC_ID[3:0]
  if ((int)blockIdx.x < skew_wgs && ((hw >> 16) & 1)) for (int k = skew_sleeps; k > 0; --k) __builtin_amdgcn_s_sleep(16);
  const unsigned long long t0 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) smem[0] = 1;
  unsigned long long t1 = t0;
  while (t1 - t0 < (unsigned long long)spin) { __builtin_amdgcn_s_sleep(1); t1 = __builtin_readcyclecounter(); }
  if (threadIdx.x == 0) {
    out[blockIdx.x * 4 + 0] = t0;
    out[blockIdx.x * 4 + 1] = t1;
    out[blockIdx.x * 4 + 2] = hw;
    out[blockIdx.x * 4 + 3] = xcc;
  }
}

int main(int argc, char** argv) {
  const int nwg = argc > 1 ? atoi(argv[1]) : 4096, spin = argc > 2 ? atoi(argv[2]) : 40000;
  const int skew_sleeps = argc > 3 ? atoi(argv[3]) : 0;
  const size_t lds = 80384;
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  unsigned long long* d;
  hipMalloc(&d, (size_t)nwg * 32);
  probe<<<nwg, 256, lds>>>(d, spin, skew_sleeps ? 512 : 0, skew_sleeps);
  hipDeviceSynchronize();
  probe<<<nwg, 256, lds>>>(d, spin, skew_sleeps ? 512 : 0, skew_sleeps);
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
  std::vector<unsigned long long> h((size_t)nwg * 4);
  hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  std::map<unsigned, int> tgs;
  struct W { unsigned long long t0, t1; unsigned tg; int wg; };
  std::map<unsigned, std::vector<W>> cu;  // key: xcc, se, cu
  for (int i = 0; i < nwg; ++i) {
    const unsigned hw = (unsigned)h[i * 4 + 2], xcc = (unsigned)h[i * 4 + 3] & 15;
    const unsigned tg = (hw >> 16) & 15, cuid = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    tgs[tg]++;
    cu[(xcc << 12) | (se << 8) | (sh << 4) | cuid].push_back({h[i * 4], h[i * 4 + 1], tg, i});
  }
  printf("work-groups %d, spin %d cycles, one-time skew of odd TG_ID slots in the first 512: %d x 1024 cycles\n", nwg, spin, skew_sleeps);
  printf("TG_ID histogram:");
  for (auto& kv : tgs) printf(" %u:%d", kv.first, kv.second);
  printf("\nCUs seen: %zu\n", cu.size());
  // per CU: sort by start time; consecutive pairs = co-resident work-groups of a round
  std::vector<long long> dstart;  // start difference inside each round's pair
  std::map<unsigned, int> pairkind;
  unsigned long long tmin = ~0ull, tmax = 0;
  for (auto& kv : cu) {
    auto& v = kv.second;
    std::sort(v.begin(), v.end(), [](const W& a, const W& b) { return a.t0 < b.t0; });
    for (size_t i = 0; i + 1 < v.size(); i += 2) {
      dstart.push_back((long long)(v[i + 1].t0 - v[i].t0));
      pairkind[(v[i].tg << 4) | v[i + 1].tg]++;
    }
    for (auto& w : v) { tmin = std::min(tmin, w.t0); tmax = std::max(tmax, w.t1); }
  }
  std::sort(dstart.begin(), dstart.end());
  if (!dstart.empty())
    printf("start-time difference of the two work-groups sharing a CU (cycles): min %lld  p10 %lld  median %lld  p90 %lld  max %lld  (%zu pairs)\n", dstart.front(),
           dstart[dstart.size() / 10], dstart[dstart.size() / 2], dstart[dstart.size() * 9 / 10], dstart.back(), dstart.size());
  printf("(TG_ID of first, TG_ID of second) per pair:");
  for (auto& kv : pairkind) printf(" (%u,%u):%d", kv.first >> 4, kv.first & 15, kv.second);
  printf("\nlaunch span %llu cycles = %.2f rounds of the spin\n", tmax - tmin, (double)(tmax - tmin) / spin);
  // one CU in detail
  auto& v0 = cu.begin()->second;
  printf("first CU (key %x): ", cu.begin()->first);
  for (size_t i = 0; i < v0.size() && i < 12; ++i) printf("[wg %d tg %u start +%llu len %llu] ", v0[i].wg, v0[i].tg, v0[i].t0 - tmin, v0[i].t1 - v0[i].t0);
  printf("\n");
  return 0;
}
