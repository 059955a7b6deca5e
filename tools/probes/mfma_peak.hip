// MFMA issue-rate probe for gfx950: what the int8 matrix pipe can do with no memory traffic at all.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip ; run on the MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef short v8s __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256) void probe(int iters, int* sink, long long* cycles) {
  v4i a = {(int)threadIdx.x * 0x01010101, 0x01020304, (int)blockIdx.x, 0x7f80ff01};
  v4i b = {0x01010101, (int)threadIdx.x, 0x02020202, 0x03030303};
  long long t0 = __builtin_readcyclecounter();
  int out = 0;
  if constexpr (MODE == 0) {  // 32x32x32 i8, 8 independent accumulators
    v16i acc[8];
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) out ^= acc[i][e];
  } else if constexpr (MODE == 1) {  // 16x16x64 i8, 16 independent accumulators
    v4i acc[16];
    for (int i = 0; i < 16; ++i) for (int e = 0; e < 4; ++e) acc[i][e] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) for (int e = 0; e < 4; ++e) out ^= acc[i][e];
  } else {  // 32x32x16 bf16 for calibration against the guide's 2.5 PF
    v16f acc[8];
    v8bf fa, fb;
    for (int e = 0; e < 8; ++e) { fa[e] = (__bf16)(float)(threadIdx.x + e); fb[e] = (__bf16)(float)(e + 1); }
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) out ^= (int)acc[i][e];
  }
  long long t1 = __builtin_readcyclecounter();
  if (out == 0x12345678) sink[0] = out;
  if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

template <int MODE>
void run(const char* name, double ops_per_iter_per_wave, int waves_per_cu) {
  int* sink; long long* cyc;
  hipMalloc(&sink, 4); hipMalloc(&cyc, 8);
  const int iters = 20000;
  const int blocks = 256 * waves_per_cu / 4;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<MODE><<<blocks, 256>>>(100, sink, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<MODE><<<blocks, 256>>>(iters, sink, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  double total = ops_per_iter_per_wave * iters * blocks * 4.0;
  printf("%-22s waves/CU %2d: %8.1f TOP/s  (%.3f ms, %.1f cycles-counter ticks per iter)\n", name, waves_per_cu, total / ms / 1e9, ms, (double)c / iters);
}

int main() {
  for (int w : {4, 8}) {
    run<0>("i8 32x32x32 (x8)", 8 * 2.0 * 32 * 32 * 32, w);
    run<1>("i8 16x16x64 (x16)", 16 * 2.0 * 16 * 16 * 64, w);
    run<2>("bf16 32x32x16 (x8)", 8 * 2.0 * 32 * 32 * 16, w);
  }
  return 0;
}
