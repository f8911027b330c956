// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns this library uses.
// MI355X_MICROARCH.md: FETCH_SIZE reports 1/2 of the bytes of a 16-B-per-lane streaming read; other widths are
// "uncalibrated: calibrate on a known byte count in your own access pattern".  Every kernel below moves a KNOWN number
// of bytes through a buffer far larger than the 256 MiB Infinity Cache; tools/fetch_calib_summary.py divides the
// counters by those byte counts.
//   read16        16 B per lane, contiguous (conv patch loads, apply, colsum)
//   read4_rows     4 B per lane, a wave reads 256 contiguous bytes of one row, 16 rows per thread, rows 2 KiB apart
//                  (cov_f16x2_kernel's raw_buffer_load_b32: channel-contiguous, pixel-strided, C = 512)
//   read4_lin      4 B per lane, contiguous
//   write16 / write8 / write4   stores of that width per lane, contiguous (conv epilogues: 16 B fp16 / fp32 rows)
// build: hipcc --offload-arch=gfx950 -O3 tools/probe/fetch_calib.hip -o tools/probe/fetch_calib
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void read16(const f4* x, size_t n16, float* sink) {
  f4 a = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) a += x[i];
  if (a[0] + a[1] + a[2] + a[3] == 12345.678f) sink[0] = a[0];
}
__global__ __launch_bounds__(256) void read4_lin(const float* x, size_t n4, float* sink) {
  float a = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) a += x[i];
  if (a == 12345.678f) sink[0] = a;
}
// rows of C floats; block = 128 channels x 2 k-groups (as cov_f16x2_kernel<128>), each thread 16 rows of one channel
__global__ __launch_bounds__(256) void read4_rows(const float* x, int C, size_t nrows, float* sink) {
  const int c = threadIdx.x % 128, kg = threadIdx.x / 128;
  const int ctile = blockIdx.x % (C / 128);
  const size_t r0 = (size_t)(blockIdx.x / (C / 128)) * 32 + kg * 16;
  float a = 0.f;
  if (r0 + 16 <= nrows) {
#pragma unroll
    for (int j = 0; j < 16; ++j) a += x[(r0 + j) * C + ctile * 128 + c];
  }
  if (a == 12345.678f) sink[0] = a;
}
__global__ __launch_bounds__(256) void write16(f4* y, size_t n16) {
  const f4 v = {1.f, 2.f, 3.f, 4.f};
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) y[i] = v;
}
__global__ __launch_bounds__(256) void write8(f2* y, size_t n8) {
  const f2 v = {1.f, 2.f};
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) y[i] = v;
}
__global__ __launch_bounds__(256) void write4(float* y, size_t n4) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) y[i] = 1.f;
}

int main() {
  const size_t bytes = (size_t)1 << 30;              // 1 GiB: four times the Infinity Cache
  float *x, *sink;
  CHECK(hipMalloc(&x, bytes));
  CHECK(hipMalloc(&sink, 256));
  CHECK(hipMemset(x, 0, bytes));
  const int C = 512;
  const size_t nrows = bytes / (C * 4);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(read16, dim3(8192), dim3(256), 0, 0, (const f4*)x, bytes / 16, sink);
    hipLaunchKernelGGL(read4_lin, dim3(8192), dim3(256), 0, 0, x, bytes / 4, sink);
    hipLaunchKernelGGL(read4_rows, dim3((unsigned)((nrows / 32) * (C / 128))), dim3(256), 0, 0, x, C, nrows, sink);
    hipLaunchKernelGGL(write16, dim3(8192), dim3(256), 0, 0, (f4*)x, bytes / 16);
    hipLaunchKernelGGL(write8, dim3(8192), dim3(256), 0, 0, (f2*)x, bytes / 8);
    hipLaunchKernelGGL(write4, dim3(8192), dim3(256), 0, 0, x, bytes / 4);
  }
  CHECK(hipDeviceSynchronize());
  printf("bytes_per_launch %zu\n", bytes);
  return 0;
}
