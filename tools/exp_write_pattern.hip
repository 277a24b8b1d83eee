// Developer tool: HBM write rate of the rollout record's store pattern against alternatives (gfx950).
//   hipcc -O2 --offload-arch=gfx950 -o /tmp/wp tools/exp_write_pattern.hip && /tmp/wp
// A: today's layout  rec[t][row][N] (u64), one wave = 64 consecutive envs, walks t: 512-byte pieces, stride N*8
// B: wave-blocked    rec[block][t][row][64]: every wave writes one contiguous stream
// C: plain grid-stride fill of the same number of bytes (the device's streaming-write rate)
// `work` dependent integer ops per ply stand in for the game logic (0 = pure stores).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int LAYOUT>
__global__ void __launch_bounds__(64) k_rec(uint64_t* rec, int64_t N, int T, int rows, int work, uint64_t seed) {
  const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
  uint64_t v = seed + i;
  uint64_t* p = LAYOUT == 0 ? rec + i : rec + (int64_t)blockIdx.x * T * rows * 64 + threadIdx.x;
  const int64_t row_stride = LAYOUT == 0 ? N : 64;
  for (int t = 0; t < T; ++t) {
    for (int w = 0; w < work; ++w) v = v * 0x9E3779B97F4A7C15ull + (v >> 17);
    for (int r = 0; r < rows; ++r) { p[0] = v + r; p += row_stride; }
  }
}

__global__ void __launch_bounds__(256) k_fill(uint64_t* dst, int64_t n, uint64_t seed) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) dst[j] = seed + j;
}

template <typename F>
static double time_us(F launch) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) launch();
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < 10; ++i) launch();
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3 / 10;
}

int main() {
  const int T = 256, rows = 4;
  for (int64_t N : {65536, 262144}) {
    const int64_t words = N * T * rows;
    uint64_t* rec; (void)hipMalloc(&rec, words * 8);
    const double gb = words * 8 / 1e9;
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, rec, words, 1ull);
    (void)hipDeviceSynchronize();
    const double tc = time_us([&] { hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, rec, words, 1ull); });
    printf("N=%ld  fill            %8.1f us  %6.2f TB/s\n", (long)N, tc, gb / tc * 1e-3 * 1e3);
    for (int work : {0, 40, 80}) {
      const double ta = time_us([&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rec<0>), dim3(N / 64), dim3(64), 0, 0, rec, N, T, rows, work, 2ull); });
      const double tb = time_us([&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rec<1>), dim3(N / 64), dim3(64), 0, 0, rec, N, T, rows, work, 3ull); });
      printf("N=%ld  work %2d  t-major %8.1f us  %6.2f TB/s   wave-blocked %8.1f us  %6.2f TB/s\n", (long)N, work, ta,
             gb / ta * 1e3 * 1e-3, tb, gb / tb * 1e3 * 1e-3);
    }
    (void)hipFree(rec);
  }
  return 0;
}
