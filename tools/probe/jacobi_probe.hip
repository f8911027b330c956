// Standalone timing probe for the rotation-set loop (debug aid, not part of the library).
#include "../../wct_tf_amd/csrc/wct.hip"
#include <stdio.h>
void wct_set_error(const char* fmt, ...) {}

template <int MODE, int N, int KB>
__global__ __launch_bounds__((N / 2) * (N / 2) / KB) void probe_kernel(const float* A, long long* out) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  f32x2* SQ = reinterpret_cast<f32x2*>(sm);
  constexpr int PITCH = N + 1, NT = (N / 2) * (N / 2) / KB;
  const int tid = threadIdx.x;
  for (int e = tid; e < N * N; e += NT) {
    const int r = e / N, c = e % N;
    f32x2 v; v[0] = A[r * 64 + c]; v[1] = r == c ? 1.f : 0.f;
    SQ[r * PITCH + c] = v;
  }
  float my_off = 0.f;
  __syncthreads();
  long long t0 = clock64();
  int cur = 0;
  for (int rep = 0; rep < 4; ++rep) cur = jacobi_sets<MODE, N, KB>(SQ, sm + 4 * N * (N + 1), tid, my_off);
  long long t1 = clock64();
  if (tid == 0) { out[0] = t1 - t0; out[2] = (long long)(my_off * 1e6f) + cur; }
}

template <int MODE, int N, int KB> void run(const float* dA, long long* dout, const char* name, int nsets, int nblocks) {
  long long h[4];
  for (int it = 0; it < 2; ++it) {
    hipLaunchKernelGGL((probe_kernel<MODE, N, KB>), dim3(nblocks), dim3((N / 2) * (N / 2) / KB), 2 * N * (N + 1) * 8 + 3 * N * 4, 0, dA, dout);
    hipDeviceSynchronize();
  }
  hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-24s KB=%d blocks=%4d: %6.0f cycles per rotation set, %6.2f us per call (%d sets)\n", name, KB, nblocks, (double)h[0] / (4 * nsets), (double)h[0] / 4 / 2400.0, nsets);
}

int main() {
  float hA[4096];
  for (int i = 0; i < 64; ++i) for (int j = 0; j < 64; ++j) hA[i * 64 + j] = (i == j ? 2.f + i : 0.f) + 0.1f / (1 + abs(i - j));
  float* dA; long long* dout;
  hipMalloc(&dA, sizeof(hA)); hipMalloc(&dout, 64);
  hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice);
  run<SWEEP_CROSS, 32, 1>(dA, dout, "cross N=32", 16, 16);
  run<SWEEP_CROSS, 32, 2>(dA, dout, "cross N=32", 16, 16);
  run<SWEEP_CROSS, 32, 4>(dA, dout, "cross N=32", 16, 16);
  run<SWEEP_CROSS, 64, 1>(dA, dout, "cross N=64", 32, 16);
  run<SWEEP_CROSS, 64, 2>(dA, dout, "cross N=64", 32, 16);
  run<SWEEP_CROSS, 64, 4>(dA, dout, "cross N=64", 32, 16);
  run<SWEEP_CROSS, 64, 8>(dA, dout, "cross N=64", 32, 16);
  run<SWEEP_CROSS, 64, 4>(dA, dout, "cross N=64 (2 WG/CU)", 32, 512);
  run<SWEEP_CROSS, 64, 1>(dA, dout, "cross N=64 (2 WG/CU)", 32, 512);
  run<SWEEP_INTRA, 64, 4>(dA, dout, "intra N=64", 31, 16);
  return 0;
}
