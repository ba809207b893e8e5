// Where do the waves of a workgroup with a wave count that is not a multiple of 4 go, and does a second such workgroup
// still fit on the CU?  Each wave records (XCC, SE, CU, SIMD) from HW_ID and its start/end clock; the kernel holds
// ~150 VGPRs (3 waves per SIMD).        hipcc --offload-arch=gfx950 -O3 wave_placement.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(unsigned long long* out, int spin) {
  float v[140];
#pragma unroll
  for (int i = 0; i < 140; ++i) v[i] = threadIdx.x * 0.5f + i;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < spin; ++it) {
#pragma unroll
    for (int i = 0; i < 140; ++i) v[i] = v[i] * 1.0001f + v[(i + 1) % 140] * 1e-6f;
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 140; ++i) s += v[i];
  const unsigned long long t1 = __builtin_readcyclecounter();
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if ((threadIdx.x & 63) == 0) {
    unsigned long long* o = out + ((size_t)blockIdx.x * WAVES + (threadIdx.x >> 6)) * 4;
    o[0] = hw; o[1] = xcc; o[2] = t0; o[3] = t1 + (s == 12345.f);
  }
}
template <int WAVES> void run(int blocks) {
  unsigned long long* d; hipMalloc(&d, (size_t)blocks * WAVES * 32);
  k<WAVES><<<blocks, 64 * WAVES>>>(d, 2000); hipDeviceSynchronize();
  std::vector<unsigned long long> h((size_t)blocks * WAVES * 4);
  hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  // per CU: how many workgroups overlap in time (max), and waves per SIMD of the first workgroups
  std::map<unsigned, std::vector<std::pair<unsigned long long, int>>> ev;      // cu key -> (time, +1/-1) of workgroups
  std::map<unsigned, std::map<int, int>> simd_hist;
  for (int b = 0; b < blocks; ++b) {
    const unsigned hw = (unsigned)h[(size_t)b * WAVES * 4], xcc = (unsigned)h[(size_t)b * WAVES * 4 + 1] & 0xf;
    const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    const unsigned key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int w = 0; w < WAVES; ++w) {
      const unsigned long long* o = &h[((size_t)b * WAVES + w) * 4];
      t0 = std::min(t0, o[2]); t1 = std::max(t1, o[3]);
      if (b < 64) simd_hist[b][(int)((o[0] >> 4) & 3)]++;
    }
    ev[key].push_back({t0, +1}); ev[key].push_back({t1, -1});
  }
  // resident workgroups on a CU at the midpoint of every workgroup's life (tails that merely touch do not count)
  int hist[8] = {0};
  for (auto& kv : ev) {
    auto& e = kv.second;
    std::vector<std::pair<unsigned long long, unsigned long long>> iv;
    std::vector<unsigned long long> st, en;
    for (auto& x : e) (x.second > 0 ? st : en).push_back(x.first);
    for (size_t i = 0; i < st.size(); ++i) {
      const unsigned long long mid = st[i] / 2 + en[i] / 2;
      int c = 0;
      for (size_t j = 0; j < st.size(); ++j) c += st[j] <= mid && mid <= en[j];
      hist[std::min(c, 7)]++;
    }
  }
  printf("%d waves per workgroup, %d workgroups on %zu CUs: workgroups by the number resident on their CU at their midpoint:", WAVES, blocks, ev.size());
  for (int i = 1; i < 8; ++i) if (hist[i]) printf("  %d: %d", i, hist[i]);
  printf("\n   waves per SIMD of workgroups 0-3:");
  for (int b = 0; b < 4; ++b) { printf("  ["); for (int s = 0; s < 4; ++s) printf("%d%s", simd_hist[b][s], s < 3 ? "," : "]"); }
  printf("\n");
  hipFree(d);
}
int main() { run<4>(1024); run<5>(1024); run<6>(1024); run<8>(1024); return 0; }
