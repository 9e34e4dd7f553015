// linear-write variants: what makes a plain fill reach 6.85 TB/s on this part (torch's fill_ does)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int V, bool NT>
__global__ void k_block_span(u32x4* out, long n16) {  // each block writes a contiguous span of 256 * V vectors; thread t writes t, t+256, ...
  const long base = static_cast<long>(blockIdx.x) * 256 * V + threadIdx.x;
#pragma unroll
  for (int k = 0; k < V; ++k) {
    const long i = base + k * 256;
    if (i < n16) {
      if (NT) __builtin_nontemporal_store(u32x4{1, 2, 3, 4}, out + i);
      else out[i] = u32x4{1, 2, 3, 4};
    }
  }
}
template <int V>
__global__ void k_thread_span(u32x4* out, long n16) {  // each thread writes V consecutive vectors (64*V bytes)
  const long base = (static_cast<long>(blockIdx.x) * 256 + threadIdx.x) * V;
#pragma unroll
  for (int k = 0; k < V; ++k)
    if (base + k < n16) out[base + k] = u32x4{1, 2, 3, 4};
}
int main() {
  const long bytes = 2000000000L, n16 = bytes / 16;
  u32x4* buf;
  hipMalloc(&buf, bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  auto time = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::printf("%-28s %.3f ms = %.2f TB/s\n", name, ms / 20, bytes / (ms / 20) / 1e9);
  };
  time("hipMemsetAsync", [&] { hipMemsetAsync(buf, 1, bytes, 0); });
#define RUN(K, V, ...) time(#K "<" #V ">", [&] { hipLaunchKernelGGL((K<V __VA_OPT__(,) __VA_ARGS__>), dim3((unsigned)((n16 + 256L * V - 1) / (256L * V))), dim3(256), 0, 0, buf, n16); })
  RUN(k_block_span, 1, false);
  RUN(k_block_span, 4, false);
  RUN(k_block_span, 8, false);
  RUN(k_block_span, 16, false);
  RUN(k_block_span, 4, true);
  RUN(k_block_span, 16, true);
  RUN(k_thread_span, 1);
  RUN(k_thread_span, 4);
  RUN(k_thread_span, 8);
  return 0;
}
