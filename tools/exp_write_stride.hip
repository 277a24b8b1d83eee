// Developer tool: does the PITCH of the record rows change the write ceiling of the record pattern?  (gfx950)
//   hipcc -O2 --offload-arch=gfx950 -o ab/wp_stride tools/exp_write_stride.hip && ab/wp_stride
// The records are u64[T][3][N] + u32[T][N] (9x9): at N = 65 536 consecutive rows lie exactly 512 KiB apart, so the four
// stores a wave issues per ply (row 0, 1, 2, meta of the same 64 envs) differ only in address bits 19 and up -- if the
// memory system interleaves channels on lower bits only, they queue on the same channel.  Same bytes, same store
// instructions, row pitch N + pad elements (pad a multiple of 64 envs = 512 B keeps every wave's piece aligned):
// does a pitch that is not a power of two raise the rate?  `work` dependent integer ops per ply stand in for the game.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__global__ void __launch_bounds__(64) k(uint64_t* rows, uint32_t* meta, int64_t pitch, int64_t mpitch, int T, int work, uint64_t seed) {
  const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
  uint64_t v = seed + i;
  uint64_t* r = rows + i;
  uint32_t* m = meta + i;
  for (int t = 0; t < T; ++t) {
    for (int w = 0; w < work; ++w) v = v * 0x9E3779B97F4A7C15ull + (v >> 17);
    __builtin_nontemporal_store(v, r);
    __builtin_nontemporal_store(v + 1, r + pitch);
    __builtin_nontemporal_store(v + 2, r + 2 * pitch);
    __builtin_nontemporal_store((uint32_t)v, m);
    r += 3 * pitch;
    m += mpitch;
  }
}

int main() {
  const int T = 256;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int64_t N : {65536, 131072}) {
    const int64_t pads[] = {0, 64, 192, 1024, 4160, 65536 / 2 + 64};
    for (int work : {0, 60}) {
      for (int64_t pad : pads) {
        const int64_t pitch = N + pad, mpitch = N + pad;
        uint64_t* rows; uint32_t* meta;
        (void)hipMalloc(&rows, (size_t)T * 3 * pitch * 8 + 4096);
        (void)hipMalloc(&meta, (size_t)T * mpitch * 4 + 4096);
        for (int it = 0; it < 200; ++it) hipLaunchKernelGGL(k, dim3(N / 64), dim3(64), 0, 0, rows, meta, pitch, mpitch, T, work, 1ull);
        (void)hipEventRecord(e0, 0);
        for (int it = 0; it < 50; ++it) hipLaunchKernelGGL(k, dim3(N / 64), dim3(64), 0, 0, rows, meta, pitch, mpitch, T, work, 1ull);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / 50;
        printf("N=%7ld work %2d  row pitch N + %-6ld %8.1f us  %6.2f TB/s\n", (long)N, work, (long)pad, us, 28.0 * N * T / us * 1e-6);
        (void)hipFree(rows); (void)hipFree(meta);
      }
    }
  }
  return 0;
}
