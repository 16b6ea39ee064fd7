#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned* __restrict__ src, unsigned* dst) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4096 / 4; i += blockDim.x) reinterpret_cast<unsigned*>(smem)[i] = 0xdeadbeef;
  __syncthreads();
  // lane l reads global 16-byte chunk (wave*64 + (l ^ 5)); where does it land?
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (wave * 64 + (lane ^ 5)) * 4),
                                   (__attribute__((address_space(3))) void*)(smem + wave * 1024), 16, 0, 0);
  __builtin_amdgcn_s_waitcnt(0x0f70);
  __syncthreads();
  u32x4_t v = *reinterpret_cast<const u32x4_t*>(smem + threadIdx.x * 16);
  *reinterpret_cast<u32x4_t*>(dst + threadIdx.x * 4) = v;
}
int main() {
  std::vector<unsigned> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = i;   // dword i: chunk i/4
  unsigned *s, *d;
  hipMalloc(&s, 4096); hipMalloc(&d, 4096);
  hipMemcpy(s, h.data(), 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 4096, 0, s, d);
  std::vector<unsigned> o(1024);
  hipMemcpy(o.data(), d, 4096, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int t = 0; t < 256; ++t) {
    int wave = t >> 6, lane = t & 63;
    unsigned expect = (wave * 64 + (lane ^ 5)) * 4;
    for (int e = 0; e < 4; ++e) if (o[t * 4 + e] != expect + e) ++bad;
  }
  printf("ldsdma probe: slot t holds chunk of lane t: bad=%d  (o[0..7]= %u %u %u %u %u %u %u %u)\n", bad, o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7]);
  return bad != 0;
}
