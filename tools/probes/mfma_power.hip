// MFMA rate under operand toggling for gfx950: the int8 matrix pipe on pseudo-random operands that change between
// consecutive instructions (the chip is power-limited on real data: tools/probes/mfma_peak.hip uses near-constant operands).
// Compares v_mfma_i32_32x32x32_i8 with v_mfma_i32_16x16x64_i8 at equal MAC counts.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_power mfma_power.hip ; run on the MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__device__ inline unsigned rnd(unsigned& s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }

// MODE 0: 32x32x32, MODE 1: 16x16x64. DATA 0: constant operands, 1: random bytes (full range), 2: random small codes (|v| < 32)
template <int MODE, int DATA>
__global__ __launch_bounds__(256) void probe(int iters, int* sink) {
  unsigned seed = threadIdx.x * 2654435761u + blockIdx.x * 97u + 12345u;
  v4i a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 4; ++e) {
      unsigned ra = rnd(seed), rb = rnd(seed);
      if (DATA == 0) { ra = 0x01010101u; rb = 0x02020202u; }
      if (DATA == 2) { ra = (ra & 0x1f1f1f1fu) ^ ((ra >> 7) & 0x01010101u) * 0xe0u; rb = (rb & 0x1f1f1f1fu) ^ ((rb >> 7) & 0x01010101u) * 0xe0u; }  // sign-extended 6-bit values
      a[i][e] = (int)ra; b[i][e] = (int)rb;
    }
  int out = 0;
  if constexpr (MODE == 0) {
    v16i acc[8];
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) out ^= acc[i][e];
  } else {
    v4i acc[16];
    for (int i = 0; i < 16; ++i) for (int e = 0; e < 4; ++e) acc[i][e] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) for (int e = 0; e < 4; ++e) out ^= acc[i][e];
  }
  if (out == 0x12345678) sink[0] = out;
}

template <int MODE, int DATA>
void run(const char* name, double ops_per_iter_per_wave) {
  int* sink; hipMalloc(&sink, 4);
  const int iters = 40000, blocks = 256 * 2;  // 8 waves per CU
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<MODE, DATA><<<blocks, 256>>>(iters, sink);  // warm (clocks settle under load)
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<MODE, DATA><<<blocks, 256>>>(iters, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s %8.1f TOP/s (%.2f ms)\n", name, ops_per_iter_per_wave * iters * blocks * 4.0 / ms / 1e9, ms);
}

int main() {
  const double o32 = 8 * 2.0 * 32 * 32 * 32, o16 = 16 * 2.0 * 16 * 16 * 64;
  for (int rep = 0; rep < 2; ++rep) {
    run<0, 0>("32x32x32 constant operands", o32);
    run<1, 0>("16x16x64 constant operands", o16);
    run<0, 1>("32x32x32 random bytes, toggling", o32);
    run<1, 1>("16x16x64 random bytes, toggling", o16);
    run<0, 2>("32x32x32 random small codes (|v|<32)", o32);
    run<1, 2>("16x16x64 random small codes (|v|<32)", o16);
  }
  return 0;
}
