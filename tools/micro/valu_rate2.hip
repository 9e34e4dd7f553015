// issue rate of 32-bit / packed VALU instructions on gfx950 (round 3): which ones run at twice the float64 rate?
// Cycles per wave64 instruction with 1, 4 and 8 waves per SIMD; each kernel runs REPS x 64 independent copies of one instruction.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

#define KINDS(X)                                                         \
  X(0, "v_fma_f64", "v_fma_f64 %0, %3, %4, %4\n\t")                      \
  X(1, "v_fma_f32", "v_fma_f32 %1, %5, %6, %6\n\t")                      \
  X(2, "v_add_f32", "v_add_f32 %1, %5, %6\n\t")                          \
  X(3, "v_mul_f32", "v_mul_f32 %1, %5, %6\n\t")                          \
  X(4, "v_max_f32", "v_max_f32 %1, %5, %6\n\t")                          \
  X(5, "v_floor_f32", "v_floor_f32 %1, %5\n\t")                          \
  X(6, "v_cvt_i32_f32", "v_cvt_i32_f32 %1, %5\n\t")                      \
  X(7, "v_cvt_f32_u32", "v_cvt_f32_u32 %1, %5\n\t")                      \
  X(8, "v_cmp_ge_f32", "v_cmp_ge_f32 vcc, %5, %6\n\t")                   \
  X(9, "v_cmp_ge_u32", "v_cmp_ge_u32 vcc, %5, %6\n\t")                   \
  X(10, "v_sub_u32", "v_sub_u32 %1, %5, %6\n\t")                         \
  X(11, "v_add_u32", "v_add_u32 %1, %5, %6\n\t")                         \
  X(12, "v_and_b32", "v_and_b32 %1, %5, %6\n\t")                         \
  X(13, "v_min_u32", "v_min_u32 %1, %5, %6\n\t")                         \
  X(14, "v_cndmask_b32 vcc", "v_cndmask_b32 %1, %5, %6, vcc\n\t")        \
  X(15, "v_mov_b32", "v_mov_b32 %1, %5\n\t")                             \
  X(16, "v_or3_b32", "v_or3_b32 %1, %5, %6, %6\n\t")                     \
  X(17, "v_min3_i32", "v_min3_i32 %1, %5, %6, %6\n\t")                   \
  X(18, "v_add3_u32", "v_add3_u32 %1, %5, %6, %6\n\t")                   \
  X(19, "v_and_or_b32", "v_and_or_b32 %1, %5, %6, %6\n\t")               \
  X(20, "v_lshl_or_b32", "v_lshl_or_b32 %1, %5, 1, %6\n\t")              \
  X(21, "v_alignbit_b32", "v_alignbit_b32 %1, %5, %6, 31\n\t")           \
  X(22, "v_mad_u32_u24", "v_mad_u32_u24 %1, %5, %6, %6\n\t")             \
  X(23, "v_mul_lo_u32", "v_mul_lo_u32 %1, %5, %6\n\t")                   \
  X(24, "v_pk_fma_f32", "v_pk_fma_f32 %0, %3, %4, %4\n\t")               \
  X(25, "v_pk_add_f32", "v_pk_add_f32 %0, %3, %4\n\t")                   \
  X(26, "v_pk_mul_f32", "v_pk_mul_f32 %0, %3, %4\n\t")                   \
  X(27, "v_addc_co_u32", "v_addc_co_u32 %1, vcc, %5, %5, vcc\n\t")       \
  X(28, "v_sub_u32 clamp", "v_sub_u32 %1, %5, %6 clamp\n\t")             \
  X(29, "v_cmp_le_u32 sgpr dst", "v_cmp_le_u32 s[20:21], %5, %6\n\t")    \
  X(30, "v_bfe_u32", "v_bfe_u32 %1, %5, 3, 8\n\t")                       \
  X(31, "v_lshrrev_b32", "v_lshrrev_b32 %1, 5, %5\n\t")                  \
  X(32, "v_min_f32", "v_min_f32 %1, %5, %6\n\t")                         \
  X(33, "v_sub_f32", "v_sub_f32 %1, %5, %6\n\t")                         \
  X(34, "v_fmac_f32", "v_fmac_f32 %1, %5, %6\n\t")                       \
  X(35, "v_pk_mov_b32", "v_pk_mov_b32 %0, %3, %4\n\t")                   \
  X(36, "v_cmp_ge_f64", "v_cmp_ge_f64 vcc, %3, %4\n\t")                  \
  X(37, "v_cvt_u32_f32", "v_cvt_u32_f32 %1, %5\n\t")                     \
  X(38, "v_rndne_f32", "v_rndne_f32 %1, %5\n\t")                         \
  X(39, "v_fract_f32", "v_fract_f32 %1, %5\n\t")                         \
  X(40, "v_cmp_le_f32 + v_addc", "v_cmp_le_f32 vcc, %5, %6\n\tv_addc_co_u32 %1, vcc, %1, %1, vcc\n\t") \
  X(41, "v_cmp_le_u32 + v_addc", "v_cmp_le_u32 vcc, %5, %6\n\tv_addc_co_u32 %1, vcc, %1, %1, vcc\n\t") \
  X(42, "v_sub_u32 + v_alignbit", "v_sub_u32 %2, %5, %6\n\tv_alignbit_b32 %1, %1, %2, 31\n\t") \
  X(43, "v_sub_f32 + v_min_f32 (pair)", "v_sub_f32 %2, %5, %6\n\tv_min_f32 %1, %1, %2\n\t")

constexpr int kPairFrom = 40;

template <int KIND>
__global__ __launch_bounds__(256) void k_rate(double* out, int reps, double seed) {
  double a = seed + threadIdx.x, b = seed * 3 + threadIdx.x, c = 1.0;
  unsigned f = threadIdx.x, f2 = 3, g = threadIdx.x * 7u, h = threadIdx.x + 77u;
  for (int i = 0; i < reps; ++i) {
#define X(K, NAME, INS) if constexpr (KIND == K) asm volatile(REP64(INS) : "+v"(c), "+v"(f), "+v"(f2) : "v"(a), "v"(b), "v"(g), "v"(h) : "vcc", "s20", "s21");
    KINDS(X)
#undef X
  }
  if (c == 12345.678 || f == 0x12345 || f2 == 99) out[threadIdx.x] = c + f + f2;
}

int main() {
  double* out;
  hipMalloc(&out, 4096);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const double ghz = p.clockRate / 1e6;
  const int cus = p.multiProcessorCount;
  std::printf("%s: %d CUs, %.2f GHz (nominal)\n", p.name, cus, ghz);
  auto run = [&](int kind, const char* name, int waves_per_simd) {
    const int reps = 2000;
    const unsigned blocks = static_cast<unsigned>(cus * waves_per_simd);
    auto launch = [&] {
      switch (kind) {
#define X(K, NAME, INS) case K: hipLaunchKernelGGL((k_rate<K>), dim3(blocks), dim3(256), 0, 0, out, reps, 1.5); break;
        KINDS(X)
#undef X
      }
    };
    launch();
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double per_simd_instr = static_cast<double>(reps) * 64 * waves_per_simd * (kind >= kPairFrom ? 2 : 1);
    std::printf("  %-30s %d wave(s)/SIMD: %.2f cycles per wave instruction\n", name, waves_per_simd, ms * 1e-3 * ghz * 1e9 / per_simd_instr);
  };
#define X(K, NAME, INS) run(K, NAME, 1); run(K, NAME, 4); run(K, NAME, 8);
  KINDS(X)
#undef X
  return 0;
}
