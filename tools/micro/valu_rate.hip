// issue rate of the VALU instructions the NRT / LeastNUMANodes inner loops are made of (gfx950): cycles per wave64
// instruction with one wave per SIMD and with four.  Each kernel runs REPS x 64 independent copies of one instruction.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND>
__global__ __launch_bounds__(256) void k_rate(double* out, int reps, double seed) {
  double a = seed + threadIdx.x, b = seed * 3 + threadIdx.x, c = 1.0;
  unsigned f = threadIdx.x, g = threadIdx.x * 7u;
  unsigned long long q = threadIdx.x * 0x100000001ull, w = 12345;
  for (int i = 0; i < reps; ++i) {
    if constexpr (KIND == 0) asm volatile(REP64("v_cmp_ge_f64 vcc, %0, %1\n\t") : : "v"(a), "v"(b) : "vcc");
    if constexpr (KIND == 1) asm volatile(REP64("v_add_f64 %0, %1, %2\n\t") : "=v"(c) : "v"(a), "v"(b));
    if constexpr (KIND == 2) asm volatile(REP64("v_fma_f64 %0, %1, %2, %2\n\t") : "=v"(c) : "v"(a), "v"(b));
    if constexpr (KIND == 3) asm volatile(REP64("v_cmp_ge_u64 vcc, %0, %1\n\t") : : "v"(q), "v"(w) : "vcc");
    if constexpr (KIND == 4) asm volatile(REP64("v_cmp_ge_u32 vcc, %0, %1\n\t") : : "v"(f), "v"(g) : "vcc");
    if constexpr (KIND == 5) asm volatile(REP64("v_addc_co_u32 %0, vcc, %0, %0, vcc\n\t") : "+v"(f) : : "vcc");
    if constexpr (KIND == 6) asm volatile(REP64("v_alignbit_b32 %0, %0, %1, 31\n\t") : "+v"(f) : "v"(g));
    if constexpr (KIND == 7) asm volatile(REP64("v_sub_co_u32 %0, vcc, %1, %2\n\t") : "=v"(f) : "v"(g), "v"(g) : "vcc");
    if constexpr (KIND == 8) asm volatile(REP64("v_cmp_ge_f32 vcc, %0, %1\n\t") : : "v"(f), "v"(g) : "vcc");
    if constexpr (KIND == 9) asm volatile(REP64("v_cmp_ge_f64 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t") : "+v"(f) : "v"(a), "v"(b) : "vcc");
    if constexpr (KIND == 10) asm volatile(REP64("v_add_f64 v[10:11], %1, -%2\n\tv_alignbit_b32 %0, %0, v11, 31\n\t") : "+v"(f) : "v"(a), "v"(b) : "v10", "v11");
    if constexpr (KIND == 11) asm volatile(REP64("v_cndmask_b32 %0, %1, %2, vcc\n\t") : "=v"(f) : "v"(g), "v"(g) : "vcc");
    if constexpr (KIND == 12) asm volatile(REP64("v_cmp_class_f64 vcc, %0, %1\n\t") : : "v"(a), "v"(f) : "vcc");
    if constexpr (KIND == 13) asm volatile(REP64("v_max_f64 %0, %1, %2\n\t") : "=v"(c) : "v"(a), "v"(b));
    if constexpr (KIND == 14) asm volatile(REP64("v_mul_f64 %0, %1, %2\n\t") : "=v"(c) : "v"(a), "v"(b));
    if constexpr (KIND == 15) asm volatile(REP64("v_lshl_add_u64 %0, %1, 0, %2\n\t") : "=v"(q) : "v"(q), "v"(w));
    if constexpr (KIND == 16) asm volatile("v_cmp_ge_u32 vcc, %1, %2\n\t" REP64("v_cndmask_b32 %0, %1, %2, vcc\n\t") : "=v"(f) : "v"(g), "v"(i) : "vcc");
    if constexpr (KIND == 17) asm volatile("v_cmp_ge_u32 s[20:21], %1, %2\n\t" REP64("v_cndmask_b32 %0, %1, %2, s[20:21]\n\t") : "=v"(f) : "v"(g), "v"(i) : "s20", "s21");
    if constexpr (KIND == 18) asm volatile(REP64("v_cmp_ge_u32 vcc, %1, %2\n\tv_cndmask_b32 %0, %1, %2, vcc\n\t") : "=v"(f) : "v"(g), "v"(i) : "vcc");
    if constexpr (KIND == 19) asm volatile(REP64("v_fma_f32 %0, %1, %2, %2\n\t") : "=v"(f) : "v"(g), "v"(i));
  }
  if (c == 12345.678 || f == 0x12345 || q == 77) out[threadIdx.x] = c + f + q;
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
  std::printf("%s: %d CUs, %.2f GHz\n", p.name, cus, ghz);
  const char* names[] = {"v_cmp_ge_f64", "v_add_f64", "v_fma_f64", "v_cmp_ge_u64", "v_cmp_ge_u32", "v_addc_co_u32", "v_alignbit_b32",
                         "v_sub_co_u32", "v_cmp_ge_f32", "cmp_f64+addc (pair)", "add_f64+alignbit (pair)", "v_cndmask_b32", "v_cmp_class_f64",
                         "v_max_f64", "v_mul_f64", "v_lshl_add_u64", "v_cndmask vcc (cmp once)", "v_cndmask sgpr (cmp once)", "cmp_u32+cndmask (pair)", "v_fma_f32"};
  auto run = [&](int kind, int waves_per_simd) {
    const int reps = 2000;
    const unsigned blocks = static_cast<unsigned>(cus * waves_per_simd);  // 256 threads = 4 waves = one per SIMD of a CU
    auto launch = [&] {
      switch (kind) {
#define C(K) case K: hipLaunchKernelGGL((k_rate<K>), dim3(blocks), dim3(256), 0, 0, out, reps, 1.5); break;
        C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15) C(16) C(17) C(18) C(19)
#undef C
      }
    };
    launch();
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double per_simd_instr = static_cast<double>(reps) * 64 * waves_per_simd * ((kind == 9 || kind == 10 || kind == 18) ? 2 : 1);
    std::printf("  %-26s %d wave(s)/SIMD: %.2f cycles per wave instruction\n", names[kind], waves_per_simd, ms * 1e-3 * ghz * 1e9 / per_simd_instr);
  };
  for (int k = 0; k < 20; ++k) {
    run(k, 1);
    run(k, 4);
  }
  return 0;
}
