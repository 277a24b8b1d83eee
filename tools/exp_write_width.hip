// Developer tool: does the width of the per-lane store change the write ceiling of the record pattern?  (gfx950)
//   hipcc -O2 --offload-arch=gfx950 -o ab/wp_width tools/exp_write_width.hip && ab/wp_width
// Same bytes per ply (24 B of planes + 4 B meta at 9x9), one wave = 64 consecutive envs walking t:
//   A  3 x 8-byte row stores + 1 x 4-byte meta store      (today's layout: rows u64[T][3][N], meta u32[T][N])
//   B  1 x 16-byte + 1 x 8-byte + 1 x 4-byte              (rows 0|1 as u128[T][N], row 2 u64[T][N], meta u32[T][N])
//   C  1 x 16-byte + 1 x 16-byte (row 2 | meta | 4 B pad)  (32 B per ply: +14 % bytes, two full-width stores)
//   D  3 x 8-byte rows only, E  1 x 16 + 1 x 8 rows only  (no meta)
// nontemporal stores as in the kernel; `work` dependent integer ops per ply stand in for the game logic.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

template <int V>
__global__ void __launch_bounds__(64) k(char* base, int64_t N, int T, int work, uint64_t seed) {
  const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
  uint64_t v = seed + i;
  uint64_t* r8 = (uint64_t*)base + i;                    // A / D: rows
  uint32_t* m4 = (uint32_t*)(base + (size_t)T * 3 * N * 8) + i;
  ulonglong2* r16 = (ulonglong2*)base + i;               // B / C / E
  uint64_t* r8b = (uint64_t*)(base + (size_t)T * N * 16) + i;
  ulonglong2* r16b = (ulonglong2*)(base + (size_t)T * N * 16) + i;
  for (int t = 0; t < T; ++t) {
    for (int w = 0; w < work; ++w) v = v * 0x9E3779B97F4A7C15ull + (v >> 17);
    if (V == 0 || V == 3) {
      __builtin_nontemporal_store(v, r8); __builtin_nontemporal_store(v + 1, r8 + N); __builtin_nontemporal_store(v + 2, r8 + 2 * N);
      r8 += 3 * N;
      if (V == 0) { __builtin_nontemporal_store((uint32_t)v, m4); m4 += N; }
    } else if (V == 1 || V == 4) {
      ulonglong2 x; x.x = v; x.y = v + 1;
      __builtin_nontemporal_store(x.x, &r16->x); __builtin_nontemporal_store(x.y, &r16->y);  // the compiler merges into dwordx4
      r16 += N;
      __builtin_nontemporal_store(v + 2, r8b); r8b += N;
      if (V == 1) { __builtin_nontemporal_store((uint32_t)v, m4); m4 += N; }
    } else {
      ulonglong2 x; x.x = v; x.y = v + 1;
      __builtin_nontemporal_store(x.x, &r16->x); __builtin_nontemporal_store(x.y, &r16->y);
      r16 += N;
      ulonglong2 y; y.x = v + 2; y.y = v;
      __builtin_nontemporal_store(y.x, &r16b->x); __builtin_nontemporal_store(y.y, &r16b->y);
      r16b += N;
    }
  }
}

template <typename F>
static double time_us(F launch) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 200; ++i) launch();
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < 50; ++i) launch();
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3 / 50;
}

int main() {
  const int T = 256;
  for (int64_t N : {65536, 131072, 262144}) {
    char* buf; (void)hipMalloc(&buf, (size_t)T * N * 32 + 4096);
    for (int work : {0, 60}) {
      const double bytes[5] = {28.0, 28.0, 32.0, 24.0, 24.0};
      const char* names[5] = {"A 3x8+4", "B 16+8+4", "C 16+16 (32 B)", "D 3x8", "E 16+8"};
      double us[5];
      us[0] = time_us([&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(k<0>), dim3(N / 64), dim3(64), 0, 0, buf, N, T, work, 1ull); });
      us[1] = time_us([&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(k<1>), dim3(N / 64), dim3(64), 0, 0, buf, N, T, work, 1ull); });
      us[2] = time_us([&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(k<2>), dim3(N / 64), dim3(64), 0, 0, buf, N, T, work, 1ull); });
      us[3] = time_us([&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(k<3>), dim3(N / 64), dim3(64), 0, 0, buf, N, T, work, 1ull); });
      us[4] = time_us([&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(k<4>), dim3(N / 64), dim3(64), 0, 0, buf, N, T, work, 1ull); });
      for (int v = 0; v < 5; ++v)
        printf("N=%7ld work %2d  %-16s %8.1f us  %6.2f TB/s\n", (long)N, work, names[v], us[v], bytes[v] * N * T / us[v] * 1e-6);
    }
    (void)hipFree(buf);
  }
  return 0;
}
