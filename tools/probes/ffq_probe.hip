// libffq_probe.so — MEASUREMENT ONLY (bench.py's side measurements; never linked into libffq_hip.so, never on the product path).
// What the int8 matrix pipe of THIS box sustains on toggling operands, so that a bench line can be compared across boxes of a pool
// whose boards run at different clocks under the same 1400 W limit (missing #4): v_mfma_i32_16x16x64_i8 issued back to
// back on pseudo-random bytes, 8 waves per CU, no LDS and no global traffic — the kernel of tools/probes/mfma_power.hip (round 1)
// behind a C entry point that launches on the caller's stream. The bf16 twin (v_mfma_f32_16x16x32_bf16) is the weight-only
// GEMM's reference point.
// Build: make -C tools/probes   (hipcc --offload-arch=gfx950 -O3 -shared -fPIC)
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

__device__ inline unsigned rnd(unsigned& s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }

__global__ __launch_bounds__(256) void probe_i8_kernel(int iters, int* sink) {
  unsigned seed = threadIdx.x * 2654435761u + blockIdx.x * 97u + 12345u;
  v4i a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 4; ++e) { a[i][e] = (int)rnd(seed); b[i][e] = (int)rnd(seed); }
  v4i acc[16];
  for (int i = 0; i < 16; ++i) for (int e = 0; e < 4; ++e) acc[i][e] = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
  }
  int out = 0;
  for (int i = 0; i < 16; ++i) for (int e = 0; e < 4; ++e) out ^= acc[i][e];
  if (out == 0x12345678) sink[0] = out;
}

__global__ __launch_bounds__(256) void probe_bf16_kernel(int iters, int* sink) {
  unsigned seed = threadIdx.x * 2654435761u + blockIdx.x * 97u + 54321u;
  v4i a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 4; ++e) {  // bf16 pairs with exponents near 1.0 (finite, no denormals): the mantissa and sign bits toggle
      a[i][e] = (int)((rnd(seed) & 0x80FF80FFu) | 0x3F003F00u);
      b[i][e] = (int)((rnd(seed) & 0x80FF80FFu) | 0x3F003F00u);
    }
  v4f acc[16];
  for (int i = 0; i < 16; ++i) for (int e = 0; e < 4; ++e) acc[i][e] = 0.0f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i)
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a[i & 3]), __builtin_bit_cast(v8bf, b[(i >> 1) & 3]), acc[i], 0, 0, 0);
  }
  float out = 0.0f;
  for (int i = 0; i < 16; ++i) for (int e = 0; e < 4; ++e) out += acc[i][e];
  if (out == 1234.5678f) sink[0] = 1;
}

// One launch of `blocks` x 4 waves x `iters` x 16 MFMAs on `stream`; `sink`: 4 bytes of device memory (never written in practice).
// Returns the integer multiply-accumulate operations (x2) / flops of the launch through *ops (0 on a launch error).
extern "C" int ffq_probe_mfma_i8(int iters, int blocks, int* sink, double* ops, void* stream) {
  probe_i8_kernel<<<blocks, 256, 0, static_cast<hipStream_t>(stream)>>>(iters, sink);
  if (ops) *ops = 16.0 * 2.0 * 16 * 16 * 64 * (double)iters * (double)blocks * 4.0;
  return hipGetLastError() == hipSuccess ? 0 : 1;
}

extern "C" int ffq_probe_mfma_bf16(int iters, int blocks, int* sink, double* ops, void* stream) {
  probe_bf16_kernel<<<blocks, 256, 0, static_cast<hipStream_t>(stream)>>>(iters, sink);
  if (ops) *ops = 16.0 * 2.0 * 16 * 16 * 32 * (double)iters * (double)blocks * 4.0;
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
