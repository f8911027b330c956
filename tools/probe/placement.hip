// Where does the dispatcher put the blocks of a {D, U} launch?  (round 5 probe, not part of the library)
// A launch of jacobi_fused4_kernel is [n_d pair-problem blocks][n_u update blocks], 256 threads, 38.8 KB of dynamic LDS (four
// blocks fit a CU).  The pair problems are the long blocks (~20 us) and VALU-bound: two of them per CU share its SIMDs
// gracefully, four do not.  This probe launches the same shape with spinning stand-ins and reads HW_ID / XCC_ID per block:
// how many of the first n_d blocks land on each CU?
// build: hipcc --offload-arch=gfx950 -O3 tools/probe/placement.hip -o tools/probe/placement ; run: tools/probe/placement
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <map>
#include <vector>
#include <algorithm>
__global__ __launch_bounds__(256, 4) void spin(unsigned* out, unsigned long long* t, int n_d, long d_cycles, long u_cycles) {
  extern __shared__ float lds[];
  unsigned a, b;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(a));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(b));
  const unsigned long long t0 = wall_clock64();
  const long n = (int)blockIdx.x < n_d ? d_cycles : u_cycles;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  float x = threadIdx.x;
  while ((long)(__builtin_amdgcn_s_memtime() - c0) < n) { x = x * 1.0001f + 1.f; lds[threadIdx.x] = x; }
  if (threadIdx.x == 0) {
    out[blockIdx.x * 2] = a; out[blockIdx.x * 2 + 1] = b;
    t[blockIdx.x * 2] = t0; t[blockIdx.x * 2 + 1] = wall_clock64();
  }
}
int main() {
  const int cases[3][2] = {{512, 2304}, {128, 576}, {16, 72}};
  for (auto& c : cases) {
    const int n_d = c[0], n_u = c[1], nb = n_d + n_u;
    unsigned* out; unsigned long long* t;
    hipMalloc(&out, nb * 8); hipMalloc(&t, nb * 16);
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(spin, dim3(nb), dim3(256), 38800, 0, out, t, n_d, 40000L, 6000L);
      hipDeviceSynchronize();
    }
    std::vector<unsigned> h(nb * 2); std::vector<unsigned long long> ht(nb * 2);
    hipMemcpy(h.data(), out, nb * 8, hipMemcpyDeviceToHost); hipMemcpy(ht.data(), t, nb * 16, hipMemcpyDeviceToHost);
    std::map<unsigned, int> d_per_cu, all_cu;
    unsigned long long tmin = ~0ull, tmax = 0, dmax = 0;
    for (int i = 0; i < nb; ++i) {
      const unsigned key = ((h[i * 2 + 1] & 0xF) << 8) | ((h[i * 2] >> 8) & 0xFF);     // xcc | se, sh, cu
      all_cu[key]++;
      if (i < n_d) d_per_cu[key]++;
      tmin = std::min(tmin, ht[i * 2]); tmax = std::max(tmax, ht[i * 2 + 1]);
      if (i < n_d) dmax = std::max(dmax, ht[i * 2 + 1]);
    }
    int hist[16] = {0};
    for (auto& kv : all_cu) { const int n = d_per_cu.count(kv.first) ? d_per_cu[kv.first] : 0; hist[n < 15 ? n : 15]++; }
    std::vector<unsigned long long> dstart;
    for (int i = 0; i < n_d; ++i) dstart.push_back(ht[i * 2] - tmin);
    std::sort(dstart.begin(), dstart.end());
    printf("n_d %d n_u %d: %zu CUs seen; CUs by number of D blocks: 0:%d 1:%d 2:%d 3:%d 4:%d >4:%d | D starts p50 %.2f us p100 %.2f us | last D ends %.1f us, launch %.1f us (100 MHz ticks)\n",
           n_d, n_u, all_cu.size(), hist[0], hist[1], hist[2], hist[3], hist[4], hist[5] + hist[6] + hist[7] + hist[8],
           dstart[n_d / 2] / 100.0, dstart[n_d - 1] / 100.0, (dmax - tmin) / 100.0, (tmax - tmin) / 100.0);
    // first 24 blocks: xcc, cu key
    printf("  first blocks (xcc:cukey):");
    for (int i = 0; i < 24; ++i) printf(" %u:%02x", h[i * 2 + 1] & 0xF, (h[i * 2] >> 8) & 0xFF);
    printf("\n");
    hipFree(out); hipFree(t);
  }
  return 0;
}
