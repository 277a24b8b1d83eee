// Developer tool: what a record store costs a wave that is alone on its SIMD (gfx950), by addressing mode.
//   hipcc -O2 --offload-arch=gfx950 -o /tmp/si tools/exp_store_issue.hip && /tmp/si
// Every iteration does WORK dependent VALU ops (stand-in for one ply) and then writes R u64 rows + one u32.
//   mode 0: no stores;  1: per-lane 64-bit pointers (what the rollout kernel does);
//   2: wave-uniform base pointer + 32-bit lane offset (global_store ... s[base:base+1]);  3: mode 2 with nt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE, int R>
__global__ void __launch_bounds__(64) k(uint64_t* rec, uint32_t* meta, int N, int T, int work, uint32_t seed) {
  const uint32_t i = blockIdx.x * 64 + threadIdx.x;
  uint32_t v = seed + i, u = v * 2654435761u;
  uint64_t* p = rec + i;
  uint32_t* q = meta + i;
  uint64_t* base = rec;      // uniform
  uint32_t* mbase = meta;    // uniform
  for (int t = 0; t < T; ++t) {
    for (int w = 0; w < work; ++w) { v = __builtin_amdgcn_alignbit(v, u, 7) ^ (u + w); u += v >> 3; }
    if (MODE == 1) {
#pragma unroll
      for (int r = 0; r < R; ++r) p[(int64_t)r * N] = ((uint64_t)v << 32) | (u + r);
      p += (int64_t)R * N;
      *q = v ^ u;
      q += N;
    } else if (MODE >= 2) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        uint64_t* dst = base + (uint32_t)(r * N) + i;  // uniform + 32-bit per-lane element offset
        const uint64_t val = ((uint64_t)v << 32) | (u + r);
        if (MODE == 3) __builtin_nontemporal_store(val, dst); else *dst = val;
      }
      base += (int64_t)R * N;
      if (MODE == 3) __builtin_nontemporal_store(v ^ u, mbase + i); else mbase[i] = v ^ u;
      mbase += N;
    }
  }
  if (v == 0x12345 && u == 77) rec[0] = v;
}

template <typename F>
static double time_us(F launch) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 200; ++i) launch();
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < 20; ++i) launch();
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3 / 20;
}

int main() {
  const int T = 256, R = 3, work = 56;  // 56 x 3 ops = ~170 VALU per iteration
  for (int N : {65536, 131072}) {
    uint64_t* rec; uint32_t* meta;
    (void)hipMalloc(&rec, (size_t)N * T * R * 8); (void)hipMalloc(&meta, (size_t)N * T * 4);
    const dim3 g(N / 64), b(64);
    const double t0 = time_us([&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(k<0, R>), g, b, 0, 0, rec, meta, N, T, work, 1u); });
    const double t1 = time_us([&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(k<1, R>), g, b, 0, 0, rec, meta, N, T, work, 2u); });
    const double t2 = time_us([&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(k<2, R>), g, b, 0, 0, rec, meta, N, T, work, 3u); });
    const double t3 = time_us([&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(k<3, R>), g, b, 0, 0, rec, meta, N, T, work, 4u); });
    printf("N=%d: no stores %.1f us | per-lane pointers %.1f us | uniform base + lane offset %.1f us | same, nt %.1f us\n", N, t0, t1, t2, t3);
    fflush(stdout);
    (void)hipFree(rec); (void)hipFree(meta);
  }
  return 0;
}
