// Bench-only: the LDS-DMA attention forward at C2's mid-block shape (1 head x 32 768 tokens x 256 channels) with parts of its tile loop removed
// (-DGM_ATTN_ABLATE=mask, see attention_dma.hip) -- what each part costs.  Build: tools/attn_ablate.sh; run on the GPU box.
#include "../generativemodels_amd/csrc/attention_dma.hip"
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
int main(int argc, char** argv) {
  const int L = argc > 1 ? atoi(argv[1]) : 32768, DH = 256;
  std::vector<unsigned short> h((size_t)L * DH);
  unsigned s = 12345u;
  for (auto& x : h) { s = s * 1664525u + 1013904223u; const float f = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; unsigned u; memcpy(&u, &f, 4); x = (unsigned short)(u >> 16); }
  unsigned short *q, *k, *v, *o; char* ws;
  hipMalloc(&q, h.size() * 2); hipMalloc(&k, h.size() * 2); hipMalloc(&v, h.size() * 2); hipMalloc(&o, h.size() * 2);
  hipMemcpy(q, h.data(), h.size() * 2, hipMemcpyHostToDevice); hipMemcpy(k, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(v, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  GmAttnDesc d = {};
  d.q = q; d.k = k; d.v = v; d.o = o; d.q_ld = d.k_ld = d.v_ld = d.o_ld = DH; d.B = 1; d.H = 1; d.Lq = d.Lk = L; d.dh = DH; d.scale = 0.0625f; d.dtype = GM_BF16;
  const long long wb = gm_attention_workspace_bytes(&d);
  hipMalloc(&ws, wb); d.workspace = ws; d.workspace_bytes = wb;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) gm_attention_dma_try(&d, nullptr);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int reps = 10;
  for (int i = 0; i < reps; ++i) gm_attention_dma_try(&d, nullptr);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("ablate mask %2d: %.3f ms per call (L %d, dh %d; includes the V^T pack, ~0.01 ms)  %s\n", GM_ATTN_ABLATE, ms / reps, L, DH, hipGetErrorString(hipGetLastError()));
  return 0;
}
