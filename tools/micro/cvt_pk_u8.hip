// tools/micro/cvt_pk_u8.hip — what v_cvt_pk_u8_f32 does with a non-integer input (round to nearest even? truncate?), negatives,
// values above 255, NaN.  k_tlp_fast2's streamlined rows feed it the unrounded float32 score.
//   hipcc --offload-arch=gfx950 -O2 -o tools/micro/cvt_pk_u8 tools/micro/cvt_pk_u8.hip && tools/micro/cvt_pk_u8
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
__global__ void k(const float* in, unsigned* out, int n) {
  int i = threadIdx.x;
  if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 0, 0u);
}
int main() {
  const float v[] = {0.0f, 0.4f, 0.5f, 0.50001f, 0.6f, 1.49999f, 1.5f, 2.5f, 3.5f, 39.5f, 40.5f, 99.4999f, 99.5f, 99.7f, 100.0f, -0.3f, -0.5f, -0.7f, -3.0f, 254.5f, 255.4f, 255.5f, 300.0f, NAN, INFINITY, -INFINITY};
  const int n = sizeof v / sizeof v[0];
  float* d;
  unsigned* o;
  hipMalloc(&d, sizeof v);
  hipMalloc(&o, n * 4);
  hipMemcpy(d, v, sizeof v, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, n);
  unsigned h[64];
  hipMemcpy(h, o, n * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) printf("%g -> %u (rint %g)\n", v[i], h[i], std::nearbyint(v[i]));
  return 0;
}
