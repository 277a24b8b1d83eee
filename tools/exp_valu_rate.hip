// Developer tool: cycles per wave64 instruction of the integer VALU ops the rollout kernel is made of (gfx950).
//   hipcc -O2 --offload-arch=gfx950 -o gpurun_out/valu_rate tools/exp_valu_rate.hip && gpurun_out/valu_rate
// One wave per SIMD (1024 waves) and four waves per SIMD; a dependent chain (latency) and four independent
// chains (issue rate).  Cycles come from s_memtime (100 MHz constant clock is NOT used: wall clock via events
// over a long loop, converted with the measured v_add_u32 rate as the 4-cycle yardstick).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x

#define DEF_KERNEL(NAME, DEP_ASM, IND_ASM)                                                             \
  __global__ void __launch_bounds__(256) dep_##NAME(unsigned* out, int iters, unsigned seed) {         \
    unsigned a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9E3779B9u, c = a + 77u, d = b + 1234567u; \
    unsigned long long q = ((unsigned long long)a << 32) | b;                                          \
    for (int i = 0; i < iters; ++i) { REP16(asm volatile(DEP_ASM : "+v"(a), "+v"(q) : "v"(b), "v"(c), "v"(d) : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) } \
    if (a == 0x12345u && q == 99) out[0] = a;                                                          \
  }                                                                                                    \
  __global__ void __launch_bounds__(256) ind_##NAME(unsigned* out, int iters, unsigned seed) {         \
    unsigned a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9E3779B9u, c = a + 77u, d = b + 1234567u; \
    unsigned a1 = a + 1, a2 = a + 2, a3 = a + 3;                                                       \
    unsigned long long q = ((unsigned long long)a << 32) | b, q1 = q + 1, q2 = q + 2, q3 = q + 3;      \
    for (int i = 0; i < iters; ++i) {                                                                  \
      REP16(asm volatile(IND_ASM : "+v"(a), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(q), "+v"(q1), "+v"(q2), "+v"(q3) \
                         : "v"(b), "v"(c), "v"(d) : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)                                                  \
    }                                                                                                  \
    if ((a ^ a1 ^ a2 ^ a3) == 0x12345u && (q ^ q1 ^ q2 ^ q3) == 99) out[0] = a;                        \
  }

// dep: %0 = a (in/out), %1 = q (64-bit in/out), %2 %3 %4 = b c d
// ind: %0..%3 = a..a3, %4..%7 = q..q3, %8 %9 %10 = b c d
DEF_KERNEL(add, "v_add_u32 %0, %0, %2\n", "v_add_u32 %0, %0, %8\nv_add_u32 %1, %1, %8\nv_add_u32 %2, %2, %8\nv_add_u32 %3, %3, %8\n")
DEF_KERNEL(xor_, "v_xor_b32 %0, %0, %2\n", "v_xor_b32 %0, %0, %8\nv_xor_b32 %1, %1, %8\nv_xor_b32 %2, %2, %8\nv_xor_b32 %3, %3, %8\n")
DEF_KERNEL(mul_lo, "v_mul_lo_u32 %0, %0, %2\n", "v_mul_lo_u32 %0, %0, %8\nv_mul_lo_u32 %1, %1, %8\nv_mul_lo_u32 %2, %2, %8\nv_mul_lo_u32 %3, %3, %8\n")
DEF_KERNEL(mul_hi, "v_mul_hi_u32 %0, %0, %2\n", "v_mul_hi_u32 %0, %0, %8\nv_mul_hi_u32 %1, %1, %8\nv_mul_hi_u32 %2, %2, %8\nv_mul_hi_u32 %3, %3, %8\n")
DEF_KERNEL(mad_u64_u32, "v_mad_u64_u32 %1, vcc, %0, %2, %1\n",
           "v_mad_u64_u32 %4, vcc, %0, %8, %4\nv_mad_u64_u32 %5, vcc, %1, %8, %5\nv_mad_u64_u32 %6, vcc, %2, %8, %6\nv_mad_u64_u32 %7, vcc, %3, %8, %7\n")
DEF_KERNEL(mul_u24, "v_mul_u32_u24 %0, %0, %2\n", "v_mul_u32_u24 %0, %0, %8\nv_mul_u32_u24 %1, %1, %8\nv_mul_u32_u24 %2, %2, %8\nv_mul_u32_u24 %3, %3, %8\n")
DEF_KERNEL(mul_hi_u24, "v_mul_hi_u32_u24 %0, %0, %2\n", "v_mul_hi_u32_u24 %0, %0, %8\nv_mul_hi_u32_u24 %1, %1, %8\nv_mul_hi_u32_u24 %2, %2, %8\nv_mul_hi_u32_u24 %3, %3, %8\n")
DEF_KERNEL(mad_u24, "v_mad_u32_u24 %0, %0, %2, %3\n", "v_mad_u32_u24 %0, %0, %8, %9\nv_mad_u32_u24 %1, %1, %8, %9\nv_mad_u32_u24 %2, %2, %8, %9\nv_mad_u32_u24 %3, %3, %8, %9\n")
DEF_KERNEL(alignbit, "v_alignbit_b32 %0, %0, %2, 7\n", "v_alignbit_b32 %0, %0, %8, 7\nv_alignbit_b32 %1, %1, %8, 7\nv_alignbit_b32 %2, %2, %8, 7\nv_alignbit_b32 %3, %3, %8, 7\n")
DEF_KERNEL(alignbit_v, "v_alignbit_b32 %0, %0, %2, %3\n", "v_alignbit_b32 %0, %0, %8, %9\nv_alignbit_b32 %1, %1, %8, %9\nv_alignbit_b32 %2, %2, %8, %9\nv_alignbit_b32 %3, %3, %8, %9\n")
DEF_KERNEL(bitop3, "v_bitop3_b32 %0, %0, %2, %3 bitop3:0x96\n", "v_bitop3_b32 %0, %0, %8, %9 bitop3:0x96\nv_bitop3_b32 %1, %1, %8, %9 bitop3:0x96\nv_bitop3_b32 %2, %2, %8, %9 bitop3:0x96\nv_bitop3_b32 %3, %3, %8, %9 bitop3:0x96\n")
DEF_KERNEL(and_or, "v_and_or_b32 %0, %0, %2, %3\n", "v_and_or_b32 %0, %0, %8, %9\nv_and_or_b32 %1, %1, %8, %9\nv_and_or_b32 %2, %2, %8, %9\nv_and_or_b32 %3, %3, %8, %9\n")
DEF_KERNEL(cndmask, "v_cndmask_b32 %0, %0, %2, vcc\n", "v_cndmask_b32 %0, %0, %8, vcc\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\nv_cndmask_b32 %3, %3, %8, vcc\n")
DEF_KERNEL(cndmask_s, "v_cndmask_b32 %0, %0, %2, s[20:21]\n", "v_cndmask_b32 %0, %0, %8, s[20:21]\nv_cndmask_b32 %1, %1, %8, s[20:21]\nv_cndmask_b32 %2, %2, %8, s[20:21]\nv_cndmask_b32 %3, %3, %8, s[20:21]\n")
DEF_KERNEL(bcnt, "v_bcnt_u32_b32 %0, %0, %2\n", "v_bcnt_u32_b32 %0, %0, %8\nv_bcnt_u32_b32 %1, %1, %8\nv_bcnt_u32_b32 %2, %2, %8\nv_bcnt_u32_b32 %3, %3, %8\n")
DEF_KERNEL(lshrrev, "v_lshrrev_b32 %0, 3, %0\n", "v_lshrrev_b32 %0, 3, %0\nv_lshrrev_b32 %1, 3, %1\nv_lshrrev_b32 %2, 3, %2\nv_lshrrev_b32 %3, 3, %3\n")
DEF_KERNEL(lshrrev_v, "v_lshrrev_b32 %0, %2, %0\n", "v_lshrrev_b32 %0, %8, %0\nv_lshrrev_b32 %1, %8, %1\nv_lshrrev_b32 %2, %8, %2\nv_lshrrev_b32 %3, %8, %3\n")
DEF_KERNEL(lshr_b64, "v_lshrrev_b64 %1, %2, %1\n", "v_lshrrev_b64 %4, %8, %4\nv_lshrrev_b64 %5, %8, %5\nv_lshrrev_b64 %6, %8, %6\nv_lshrrev_b64 %7, %8, %7\n")
DEF_KERNEL(lshl_add_u64, "v_lshl_add_u64 %1, %1, 1, %1\n", "v_lshl_add_u64 %4, %4, 1, %4\nv_lshl_add_u64 %5, %5, 1, %5\nv_lshl_add_u64 %6, %6, 1, %6\nv_lshl_add_u64 %7, %7, 1, %7\n")
DEF_KERNEL(add3, "v_add3_u32 %0, %0, %2, %3\n", "v_add3_u32 %0, %0, %8, %9\nv_add3_u32 %1, %1, %8, %9\nv_add3_u32 %2, %2, %8, %9\nv_add3_u32 %3, %3, %8, %9\n")
DEF_KERNEL(or3, "v_or3_b32 %0, %0, %2, %3\n", "v_or3_b32 %0, %0, %8, %9\nv_or3_b32 %1, %1, %8, %9\nv_or3_b32 %2, %2, %8, %9\nv_or3_b32 %3, %3, %8, %9\n")
DEF_KERNEL(ffbl, "v_ffbl_b32 %0, %0\n", "v_ffbl_b32 %0, %0\nv_ffbl_b32 %1, %1\nv_ffbl_b32 %2, %2\nv_ffbl_b32 %3, %3\n")
DEF_KERNEL(cmp_cnd, "v_cmp_lt_u32 vcc, %0, %2\nv_cndmask_b32 %0, %0, %3, vcc\n",
           "v_cmp_lt_u32 vcc, %0, %8\nv_cndmask_b32 %0, %0, %9, vcc\nv_cmp_lt_u32 vcc, %1, %8\nv_cndmask_b32 %1, %1, %9, vcc\nv_cmp_lt_u32 vcc, %2, %8\nv_cndmask_b32 %2, %2, %9, vcc\nv_cmp_lt_u32 vcc, %3, %8\nv_cndmask_b32 %3, %3, %9, vcc\n")
DEF_KERNEL(cmp_e64_cnd, "v_cmp_lt_u32 s[20:21], %0, %2\nv_cndmask_b32 %0, %0, %3, s[20:21]\n",
           "v_cmp_lt_u32 s[20:21], %0, %8\nv_cndmask_b32 %0, %0, %9, s[20:21]\nv_cmp_lt_u32 s[22:23], %1, %8\nv_cndmask_b32 %1, %1, %9, s[22:23]\nv_cmp_lt_u32 s[24:25], %2, %8\nv_cndmask_b32 %2, %2, %9, s[24:25]\nv_cmp_lt_u32 s[26:27], %3, %8\nv_cndmask_b32 %3, %3, %9, s[26:27]\n")
DEF_KERNEL(mov_dpp, "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n",
           "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")


// single asm statement per 16 instructions: the compiler cannot put hazard s_nops in between
#define R4(x) x x x x
#define R16S(x) R4(R4(x))
DEF_KERNEL(cnd_vcc1, R16S("v_cndmask_b32 %0, %0, %2, vcc\n"), R4("v_cndmask_b32 %0, %0, %8, vcc\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\nv_cndmask_b32 %3, %3, %8, vcc\n"))
DEF_KERNEL(add_nop, R16S("v_add_u32 %0, %0, %2\ns_nop 0\n"), R4("v_add_u32 %0, %0, %8\ns_nop 0\nv_add_u32 %1, %1, %8\ns_nop 0\nv_add_u32 %2, %2, %8\ns_nop 0\nv_add_u32 %3, %3, %8\ns_nop 0\n"))
DEF_KERNEL(add_sadd, R16S("v_add_u32 %0, %0, %2\ns_add_i32 s20, s20, 1\n"), R4("v_add_u32 %0, %0, %8\ns_add_i32 s20, s20, 1\nv_add_u32 %1, %1, %8\ns_add_i32 s21, s21, 1\nv_add_u32 %2, %2, %8\ns_add_i32 s22, s22, 1\nv_add_u32 %3, %3, %8\ns_add_i32 s23, s23, 1\n"))
DEF_KERNEL(cnd_nop, R16S("v_cndmask_b32 %0, %0, %2, vcc\ns_nop 0\n"), R4("v_cndmask_b32 %0, %0, %8, vcc\ns_nop 0\nv_cndmask_b32 %1, %1, %8, vcc\ns_nop 0\nv_cndmask_b32 %2, %2, %8, vcc\ns_nop 0\nv_cndmask_b32 %3, %3, %8, vcc\ns_nop 0\n"))
DEF_KERNEL(cmpvcc_cnd, R16S("v_cmp_lt_u32 vcc, %0, %2\nv_cndmask_b32 %0, %0, %3, vcc\n"), R4("v_cmp_lt_u32 vcc, %0, %8\nv_cndmask_b32 %0, %0, %9, vcc\nv_cmp_lt_u32 vcc, %1, %8\nv_cndmask_b32 %1, %1, %9, vcc\nv_cmp_lt_u32 vcc, %2, %8\nv_cndmask_b32 %2, %2, %9, vcc\nv_cmp_lt_u32 vcc, %3, %8\nv_cndmask_b32 %3, %3, %9, vcc\n"))
DEF_KERNEL(cmp_sand_cnd, R16S("v_cmp_lt_u32 s[20:21], %0, %2\ns_and_b64 s[22:23], s[20:21], exec\nv_cndmask_b32 %0, %0, %3, s[22:23]\n"), R4("v_cmp_lt_u32 s[20:21], %0, %8\ns_and_b64 s[22:23], s[20:21], exec\nv_cndmask_b32 %0, %0, %9, s[22:23]\nv_cmp_lt_u32 s[20:21], %1, %8\ns_and_b64 s[22:23], s[20:21], exec\nv_cndmask_b32 %1, %1, %9, s[22:23]\nv_cmp_lt_u32 s[20:21], %2, %8\ns_and_b64 s[22:23], s[20:21], exec\nv_cndmask_b32 %2, %2, %9, s[22:23]\nv_cmp_lt_u32 s[20:21], %3, %8\ns_and_b64 s[22:23], s[20:21], exec\nv_cndmask_b32 %3, %3, %9, s[22:23]\n"))
DEF_KERNEL(readlane, R16S("v_readlane_b32 s20, %0, 3\nv_add_u32 %0, s20, %0\n"), R4("v_readlane_b32 s20, %0, 3\nv_add_u32 %0, s20, %0\nv_readlane_b32 s21, %1, 3\nv_add_u32 %1, s21, %1\nv_readlane_b32 s22, %2, 3\nv_add_u32 %2, s22, %2\nv_readlane_b32 s23, %3, 3\nv_add_u32 %3, s23, %3\n"))
DEF_KERNEL(add_sgpr, R16S("v_add_u32 %0, s20, %0\n"), R4("v_add_u32 %0, s20, %0\nv_add_u32 %1, s20, %1\nv_add_u32 %2, s20, %2\nv_add_u32 %3, s20, %3\n"))
DEF_KERNEL(add_lit, R16S("v_add_u32 %0, 0x12345678, %0\n"), R4("v_add_u32 %0, 0x12345678, %0\nv_add_u32 %1, 0x12345678, %1\nv_add_u32 %2, 0x12345678, %2\nv_add_u32 %3, 0x12345678, %3\n"))
DEF_KERNEL(alignbit_lit, R16S("v_alignbit_b32 %0, %0, %2, 10\nv_and_b32 %0, 0x3ff7fdff, %0\n"), R4("v_alignbit_b32 %0, %0, %8, 10\nv_and_b32 %0, 0x3ff7fdff, %0\nv_alignbit_b32 %1, %1, %8, 10\nv_and_b32 %1, 0x3ff7fdff, %1\nv_alignbit_b32 %2, %2, %8, 10\nv_and_b32 %2, 0x3ff7fdff, %2\nv_alignbit_b32 %3, %3, %8, 10\nv_and_b32 %3, 0x3ff7fdff, %3\n"))

struct Case { const char* name; void (*dep)(unsigned*, int, unsigned); void (*ind)(unsigned*, int, unsigned); int per_dep, per_ind; };
#define CASE(N, PD, PI) {#N, dep_##N, ind_##N, PD, PI}

static double time_kernel(void (*k)(unsigned*, int, unsigned), unsigned* out, int blocks, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, 1u);
  (void)hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, 2u + r);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best * 1e-3;
}

int main() {
  unsigned* out; (void)hipMalloc(&out, 4096);
  std::vector<Case> cases = {
      CASE(add, 1, 4), CASE(xor_, 1, 4), CASE(mul_lo, 1, 4), CASE(mul_hi, 1, 4), CASE(mad_u64_u32, 1, 4),
      CASE(mul_u24, 1, 4), CASE(mul_hi_u24, 1, 4), CASE(mad_u24, 1, 4), CASE(alignbit, 1, 4), CASE(alignbit_v, 1, 4),
      CASE(bitop3, 1, 4), CASE(and_or, 1, 4), CASE(cndmask, 1, 4), CASE(cndmask_s, 1, 4), CASE(bcnt, 1, 4),
      CASE(lshrrev, 1, 4), CASE(lshrrev_v, 1, 4), CASE(lshr_b64, 1, 4), CASE(lshl_add_u64, 1, 4), CASE(add3, 1, 4),
      CASE(or3, 1, 4), CASE(ffbl, 1, 4), CASE(cmp_cnd, 2, 8), CASE(cmp_e64_cnd, 2, 8), CASE(mov_dpp, 1, 4),
      CASE(cnd_vcc1, 16, 16), CASE(add_nop, 16, 16), CASE(add_sadd, 16, 16), CASE(cnd_nop, 16, 16), CASE(cmpvcc_cnd, 32, 32),
      CASE(cmp_sand_cnd, 32, 32), CASE(readlane, 32, 32), CASE(add_sgpr, 16, 16), CASE(add_lit, 16, 16), CASE(alignbit_lit, 32, 32)};
  const int iters = 5000;
  // warm the clocks
  for (int i = 0; i < 20; ++i) time_kernel(dep_add, out, 1024, iters);
  // yardstick: wave64 v_add_u32 = 4 cycles issue => clock estimate
  const double t_add = time_kernel(ind_add, out, 256, iters);  // 256 blocks x 4 waves = 1 wave per SIMD
  const double ghz = (double)iters * 16 * 4 * 4 / t_add * 1e-9;
  printf("clock estimate from independent v_add_u32 (4 cycles each): %.2f GHz\n", ghz);
  printf("%-14s %10s %10s %10s %10s   (cycles per instruction per wave; x4 = four waves per SIMD, cycles per SIMD)\n", "op", "dep x1", "ind x1",
         "dep x4", "ind x4");
  for (auto& c : cases) {
    double r[4];
    int k = 0;
    for (int blocks : {256, 1024}) {
      const double waves_per_simd = blocks / 256.0;
      const double td = time_kernel(c.dep, out, blocks, iters), ti = time_kernel(c.ind, out, blocks, iters);
      r[k++] = td * ghz * 1e9 / ((double)iters * 16 * c.per_dep * waves_per_simd);
      r[k++] = ti * ghz * 1e9 / ((double)iters * 16 * c.per_ind * waves_per_simd);
    }
    printf("%-14s %10.2f %10.2f %10.2f %10.2f\n", c.name, r[0], r[1], r[2], r[3]);
    fflush(stdout);
  }
  return 0;
}
