// write-pattern microbenchmark: how fast can the result tables be written, as a function of how the (row, 1 KiB tile) grid is
// dealt to wavefronts and blocks?  No compute; stores only.  usage: wpattern <waves_per_block> <rows_per_chunk> <row_stride> <tables>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void k_pattern(uint8_t* out, int64_t table_bytes, int tables, int64_t stride, int n_tiles, int64_t rows, int rows_per_chunk) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
  const int64_t unit = static_cast<int64_t>(blockIdx.x) * wpb + wave;
  const int tile = static_cast<int>(unit % n_tiles);
  const int64_t chunk = unit / n_tiles;
  if (tables == 3) {  // row-strided assignment: wave (tile, chunk) takes rows chunk, chunk + n_chunks, ... — at every step the
    // concurrently running waves write one compact band of consecutive rows (a linear sweep) instead of one row per chunk band
    const int64_t n_chunks = (rows + rows_per_chunk - 1) / rows_per_chunk;
    const int64_t col = (static_cast<int64_t>(tile) * 64 + lane) * 16;
    if (chunk >= n_chunks || col >= stride) return;
    const uint4 v{1u, 2u, 3u, static_cast<unsigned>(tile)};
    for (int64_t r = chunk; r < rows; r += n_chunks)
      for (int t = 0; t < 2; ++t) *reinterpret_cast<uint4*>(out + t * table_bytes + r * stride + col) = v;
    return;
  }
  const int64_t r0 = chunk * rows_per_chunk;
  if (r0 >= rows) return;
  const int64_t r1 = r0 + rows_per_chunk < rows ? r0 + rows_per_chunk : rows;
  const int64_t col = (static_cast<int64_t>(tile) * 64 + lane) * 16;
  if (col >= stride) return;
  const uint4 v{1u, 2u, 3u, static_cast<unsigned>(tile)};
  if (tables == 4 || tables == 5) {  // drain the wave's stores after every row (4) / every 4th row (5)
    for (int64_t r = r0; r < r1; ++r) {
      for (int t = 0; t < 2; ++t) *reinterpret_cast<uint4*>(out + t * table_bytes + r * stride + col) = v;
      if (tables == 4 || ((r - r0) & 3) == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  } else if (tables > 0) {
    for (int64_t r = r0; r < r1; ++r)
      for (int t = 0; t < tables; ++t) *reinterpret_cast<uint4*>(out + t * table_bytes + r * stride + col) = v;
  } else {  // table-major: all rows of table 0, then all rows of table 1
    for (int t = 0; t < -tables; ++t)
      for (int64_t r = r0; r < r1; ++r) *reinterpret_cast<uint4*>(out + t * table_bytes + r * stride + col) = v;
  }
}

__global__ void k_linear(uint4* out, int64_t n16) {  // the ideal: fully linear, like a fill
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n16; i += static_cast<int64_t>(gridDim.x) * blockDim.x) out[i] = uint4{1, 2, 3, 4};
}

int main(int argc, char** argv) {
  const int64_t rows = 100000, n = 10000;
  uint8_t* buf;
  const int64_t max_stride = 16384;
  hipMalloc(&buf, 2 * rows * max_stride);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  auto time = [&](auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 20;
  };
  {
    const int64_t n16 = 2 * rows * 10112 / 16;
    const float ms = time([&] { hipLaunchKernelGGL(k_linear, dim3(256 * 16), dim3(256), 0, 0, reinterpret_cast<uint4*>(buf), n16); });
    std::printf("linear 2 x %lld B: %.3f ms = %.2f TB/s\n", (long long)(rows * 10112), ms, 2.0 * rows * 10112 / ms / 1e9);
  }
  const int strides[] = {10112, 10240};
  const int wpbs[] = {1, 4};
  const int rpcs[] = {64};
  for (int stride : strides)
    for (int rpc : rpcs)
      for (int wpb : wpbs) {
        const int n_tiles = (stride + 1023) / 1024;
        const int64_t chunks = (rows + rpc - 1) / rpc, units = chunks * n_tiles;
        const unsigned blocks = static_cast<unsigned>((units + wpb - 1) / wpb);
        for (int tables : {2, 4, 5}) {
          const int nt = 2;
          const float ms = time([&] { hipLaunchKernelGGL(k_pattern, dim3(blocks), dim3(64 * wpb), 0, 0, buf, rows * (int64_t)stride, tables, (int64_t)stride, n_tiles, rows, rpc); });
          std::printf("stride %5d rows/chunk %3d waves/block %2d tables %2d: %.3f ms = %.2f TB/s written\n", stride, rpc, wpb, tables, ms, (double)nt * rows * stride / ms / 1e9);
        }
      }
  return 0;
}
