// Lane-by-lane CPU emulation of the HIP device facilities the eigensolver's pair-problem code (csrc/jacobi_dev.h) uses:
// a workgroup is NT host threads, a wave 64 consecutive ones; __syncthreads is a pthread barrier, cross-lane operations
// (DPP row rotate, __shfl, readfirstlane) and the MFMAs exchange their operands through a per-wave array between two
// wave barriers.  Test infrastructure only (tests/test_jacobi_emul.py): lets the kernel SOURCE run here, where there is no GPU.
#pragma once
#include <pthread.h>
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <cmath>
#include <algorithm>
#include <vector>

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)

namespace emul {
struct Idx { unsigned x, y, z; };
struct Wave {
  pthread_barrier_t bar;
  float f[64];
  half8 a8[64], b8[64];
};
struct Block {
  pthread_barrier_t bar;
  std::vector<Wave> waves;
  int nt;
};
extern thread_local Idx tidx;
extern thread_local Idx bidx;
extern thread_local Block* blk;
inline Wave& wave() { return blk->waves[tidx.x >> 6]; }
inline int lane() { return tidx.x & 63; }
inline void wbar() { pthread_barrier_wait(&wave().bar); }

inline void syncthreads() { pthread_barrier_wait(&blk->bar); }
// every lane of the wave deposits v; returns the value of lane `src`
inline float shfl(float v, int src) {
  Wave& w = wave();
  w.f[lane()] = v;
  wbar();
  const float r = w.f[src & 63];
  wbar();
  return r;
}
inline int shfl_i(int v, int src) { return __builtin_bit_cast(int, shfl(__builtin_bit_cast(float, v), src)); }
inline int readfirstlane(int v) { return shfl_i(v, 0); }
// DPP: the controls the solver uses -- row_ror:n (0x120 + n): lane i of a 16-lane row reads lane (i - n) mod 16 of its row;
// wave_shl:1 (0x130): lane i reads lane i + 1 (lane 63 keeps `old`)
inline int update_dpp(int old, int src, int ctrl, int, int, bool) {
  const int l = lane();
  Wave& w = wave();
  w.f[l] = __builtin_bit_cast(float, src);
  wbar();
  int r = old;
  if (ctrl >= 0x121 && ctrl <= 0x12F) r = __builtin_bit_cast(int, w.f[(l & 48) | ((l - (ctrl - 0x120)) & 15)]);
  else if (ctrl == 0x130) { if (l < 63) r = __builtin_bit_cast(int, w.f[l + 1]); }
  else __builtin_trap();
  wbar();
  return r;
}
// v_mfma_f32_16x16x4_f32: A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15], C/D lane l register r = [4 (l >> 4) + r][l & 15];
// one rounding per product, k ascending (cdna_hip_programming.md 3: bitwise an fmaf chain)
inline f32x4 mfma_16x16x4_f32(float a, float b, f32x4 c) {
  const int l = lane();
  Wave& w = wave();
  w.f[l] = a;
  reinterpret_cast<float*>(w.a8)[l] = b;
  wbar();
  const float* bb = reinterpret_cast<const float*>(w.a8);
  f32x4 d = c;
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * (l >> 4) + r, col = l & 15;
    for (int k = 0; k < 4; ++k) d[r] = fmaf(w.f[k * 16 + row], bb[k * 16 + col], d[r]);
  }
  wbar();
  return d;
}
// v_mfma_f32_16x16x32_f16: A[i = l & 15][k = 8 (l >> 4) + e], B[k = 8 (l >> 4) + e][j = l & 15]; fp32 accumulation
inline f32x4 mfma_16x16x32_f16(half8 a, half8 b, f32x4 c) {
  const int l = lane();
  Wave& w = wave();
  w.a8[l] = a; w.b8[l] = b;
  wbar();
  f32x4 d = c;
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * (l >> 4) + r, col = l & 15;
    float acc = 0.f;
    for (int q = 0; q < 4; ++q)
      for (int e = 0; e < 8; ++e) acc += (float)w.a8[q * 16 + row][e] * (float)w.b8[q * 16 + col][e];
    d[r] += acc;
  }
  wbar();
  return d;
}
inline unsigned atomic_max(unsigned* p, unsigned v) {
  unsigned old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}
// run `fn(tid)` on nt emulated threads of one workgroup with blockIdx.x = bx
template <class F>
void run_block(int nt, int bx, F fn) {
  Block b;
  b.nt = nt;
  b.waves.resize((nt + 63) / 64);
  pthread_barrier_init(&b.bar, nullptr, nt);
  for (size_t w = 0; w < b.waves.size(); ++w) pthread_barrier_init(&b.waves[w].bar, nullptr, std::min(64, nt - (int)w * 64));
  struct Arg { Block* b; int tid, bx; F* fn; };
  std::vector<Arg> args(nt);
  std::vector<pthread_t> th(nt);
  for (int t = 0; t < nt; ++t) {
    args[t] = Arg{&b, t, bx, &fn};
    pthread_attr_t at;
    pthread_attr_init(&at);
    pthread_attr_setstacksize(&at, 1 << 20);
    pthread_create(&th[t], &at, [](void* v) -> void* {
      Arg* a = static_cast<Arg*>(v);
      blk = a->b; tidx = Idx{(unsigned)a->tid, 0, 0}; bidx = Idx{(unsigned)a->bx, 0, 0};
      (*a->fn)(a->tid);
      return nullptr;
    }, &args[t]);
    pthread_attr_destroy(&at);
  }
  for (int t = 0; t < nt; ++t) pthread_join(th[t], nullptr);
  pthread_barrier_destroy(&b.bar);
  for (auto& w : b.waves) pthread_barrier_destroy(&w.bar);
}
}  // namespace emul

#define R4_OPAQUE(x) ((void)0)
#define threadIdx emul::tidx
#define blockIdx emul::bidx
#define __syncthreads() emul::syncthreads()
#define __shfl(v, src, w) emul::shfl((v), (src))
#define __shfl_xor(v, m, w) emul::shfl((v), emul::lane() ^ (m))
#define atomicMax(p, v) emul::atomic_max((p), (v))
#define __float_as_uint(x) __builtin_bit_cast(unsigned, (float)(x))
#define __uint_as_float(x) __builtin_bit_cast(float, (unsigned)(x))
#define __builtin_amdgcn_sqrtf(x) sqrtf(x)
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __builtin_amdgcn_rsqf(x) (1.0f / sqrtf(x))
#define __builtin_amdgcn_readfirstlane(x) emul::readfirstlane(x)
#define __builtin_amdgcn_readlane(x, l) emul::shfl_i((x), (l))
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_update_dpp(o, s, c, rm, bm, bc) emul::update_dpp((o), (s), (c), (rm), (bm), (bc))
#define __builtin_amdgcn_ds_bpermute(addr, v) emul::shfl_i((v), (addr) >> 2)
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) emul::mfma_16x16x4_f32((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, x, y, z) emul::mfma_16x16x32_f16((a), (b), (c))
