// Probe (not part of the product): semantics of buffer_load ... lds on gfx950
//  (1) do out-of-range lanes write zeros into LDS?   (2) is the SGPR offset part of the range check?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k(const unsigned* __restrict__ x, unsigned* out, int nbytes, int soff) {
  __shared__ __attribute__((aligned(16))) unsigned smem[256];
  smem[threadIdx.x] = 0xdeadbeefu; smem[threadIdx.x + 64] = 0xdeadbeefu;
  smem[threadIdx.x + 128] = 0xdeadbeefu; smem[threadIdx.x + 192] = 0xdeadbeefu;
  __syncthreads();
  auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, nbytes, 0x00020000);
  unsigned voff = (threadIdx.x & 1) ? 0x80000000u : threadIdx.x * 16;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)smem, 16, voff, soff, 0, 0);
  __syncthreads();
  for (int i = 0; i < 4; ++i) out[threadIdx.x * 4 + i] = smem[threadIdx.x * 4 + i];
}
int main() {
  const int n = 4096;
  std::vector<unsigned> h(n);
  for (int i = 0; i < n; ++i) h[i] = i;
  unsigned *dx, *dout;
  hipMalloc(&dx, n * 4); hipMalloc(&dout, 1024);
  hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice);
  for (int test = 0; test < 2; ++test) {
    const int nbytes = test == 0 ? n * 4 : 512, soff = test == 0 ? 0 : 1024;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, dout, nbytes, soff);
    std::vector<unsigned> o(256);
    hipMemcpy(o.data(), dout, 1024, hipMemcpyDeviceToHost);
    printf("test %d (num_records %d, soffset %d):\n", test, nbytes, soff);
    for (int l = 0; l < 8; ++l) printf("  lane %d -> %08x %08x %08x %08x\n", l, o[l * 4], o[l * 4 + 1], o[l * 4 + 2], o[l * 4 + 3]);
    printf("  lane 40 -> %08x ; lane 41 -> %08x\n", o[160], o[164]);
  }
  return 0;
}
