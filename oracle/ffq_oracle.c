/*
 * ffq_oracle.c — CPU restatement of the reference's affine fake-quantization path, in plain C.
 *
 * TEST INFRASTRUCTURE ONLY. This file is the checker the HIP kernels are compared against; it is
 * never shipped, never linked into fastforward_amd, and never a fallback. Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * It exports the ABI of include/ffq.h with HOST pointers (stream is ignored). Each function
 * follows the reference's eager chain step by step, including the intermediate roundings the
 * chain performs because every ATen op materialises its result in the promoted dtype. Citations
 * are file:line under /root/reference/src/fastforward/.
 *
 * Parity pin: tests/test_oracle_golden.py checks this file against the fixtures in tests/golden/,
 * which tests/golden/gen_golden.py produced by importing the reference itself (CPU eager) in the
 * build container, and against the known-answer vectors the reference's own tests hold
 * (tests/nn/test_linear_quantizer.py:20-72,189-219,400-418; tests/quantization/test_tiled_tensor.py;
 * tests/quantization/affine/test_range.py:10-40; tests/range_setting/test_minmax.py:41-87).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off; never -ffast-math).
 */
#include "../include/ffq.h"

#include <float.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static __thread char g_err[512];

static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

int ffq_abi_version(void) { return FFQ_ABI_VERSION; }
const char* ffq_last_error(void) { return g_err; }
const char* ffq_backend_name(void) { return "oracle:c"; }

/* ------------------------------------------------------------------------------------------ */
/* dtype helpers                                                                              */
/* ------------------------------------------------------------------------------------------ */

static int dt_valid(int dt) { return dt >= FFQ_F32 && dt <= FFQ_U8; }
static int dt_is_float(int dt) { return dt == FFQ_F32 || dt == FFQ_BF16 || dt == FFQ_F16 || dt == FFQ_F64; }
static size_t dt_size(int dt) {
  switch (dt) {
    case FFQ_F32: case FFQ_I32: return 4;
    case FFQ_BF16: case FFQ_F16: case FFQ_I16: return 2;
    case FFQ_F64: case FFQ_I64: return 8;
    default: return 1;
  }
}

static float bits_f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static uint32_t f32_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* bf16 <-> f32: round to nearest even, NaN kept quiet (c10::BFloat16 round_to_nearest_even). */
static uint16_t f32_to_bf16(float f) {
  uint32_t u = f32_bits(f);
  if (f != f) return 0x7FC0;
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static float bf16_to_f32(uint16_t h) { return bits_f32((uint32_t)h << 16); }

/* IEEE binary16 <-> f32, round to nearest even, with subnormals and overflow to inf. */
static uint16_t f32_to_f16(float f) {
  uint32_t x = f32_bits(f);
  uint32_t sign = (x >> 16) & 0x8000u;
  uint32_t abs = x & 0x7FFFFFFFu;
  if (abs > 0x7F800000u) return (uint16_t)(sign | 0x7E00u);         /* NaN */
  if (abs >= 0x47800000u) return (uint16_t)(sign | 0x7C00u);        /* >= 65536 -> inf (incl. inf) */
  if (abs < 0x38800000u) {                                          /* subnormal half or zero */
    if (abs < 0x33000000u) return (uint16_t)sign;                   /* < 2^-25 -> 0 */
    /* value = abs * 2^24 rounded to integer, RNE */
    float scaled = bits_f32(abs) * 16777216.0f;                      /* exact: power-of-two scale */
    float r = nearbyintf(scaled);                                   /* RNE in default mode */
    return (uint16_t)(sign | (uint32_t)r);
  }
  uint32_t mant = abs & 0x007FFFFFu;
  uint32_t exp = (abs >> 23) - 112u;                                 /* rebias 127 -> 15 */
  uint32_t h = (exp << 10) | (mant >> 13);
  uint32_t rem = mant & 0x1FFFu;
  if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h += 1u;        /* carries into exponent ok */
  return (uint16_t)(sign | h);
}
static float f16_to_f32(uint16_t h) {
  uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1Fu;
  uint32_t mant = h & 0x3FFu;
  if (exp == 0) {
    float v = (float)mant * (1.0f / 16777216.0f);                    /* mant * 2^-24 */
    return bits_f32(sign | f32_bits(v));
  }
  if (exp == 31) return bits_f32(sign | 0x7F800000u | (mant << 13));
  return bits_f32(sign | ((exp + 112u) << 23) | (mant << 13));
}

/* Load element i of a typed buffer as a double (exact for every supported dtype but I64 > 2^53). */
static double ld(const void* p, int dt, int64_t i) {
  switch (dt) {
    case FFQ_F32: return (double)((const float*)p)[i];
    case FFQ_BF16: return (double)bf16_to_f32(((const uint16_t*)p)[i]);
    case FFQ_F16: return (double)f16_to_f32(((const uint16_t*)p)[i]);
    case FFQ_F64: return ((const double*)p)[i];
    case FFQ_I8: return (double)((const int8_t*)p)[i];
    case FFQ_I16: return (double)((const int16_t*)p)[i];
    case FFQ_I32: return (double)((const int32_t*)p)[i];
    case FFQ_I64: return (double)((const int64_t*)p)[i];
    case FFQ_U8: return (double)((const uint8_t*)p)[i];
  }
  return 0.0;
}

/*
 * Round a value to what a tensor of dtype `dt` can hold. ATen evaluates half/bfloat16 ops in float
 * ("opmath") and rounds the result once; callers therefore pass a value that was computed in
 * float for those dtypes (see op2()).
 */
static double round_to(double v, int dt) {
  switch (dt) {
    case FFQ_F32: return (double)(float)v;
    case FFQ_BF16: return (double)bf16_to_f32(f32_to_bf16((float)v));
    case FFQ_F16: return (double)f16_to_f32(f32_to_f16((float)v));
    default: return v; /* F64, and integer dtypes never appear as a compute stage */
  }
}

/* One binary ATen op in dtype `dt`: operands already hold dt-representable values. */
enum { OP_DIV, OP_SUB, OP_ADD, OP_MUL };
static double op2(int op, double a, double b, int dt) {
  if (dt == FFQ_F64) {
    switch (op) {
      case OP_DIV: return a / b;
      case OP_SUB: return a - b;
      case OP_ADD: return a + b;
      default: return a * b;
    }
  }
  /* f32 / bf16 / f16: evaluate in float, then round to dt. The volatile stops gcc folding the
     float op into a double one. */
  volatile float fa = (float)a, fb = (float)b, r;
  switch (op) {
    case OP_DIV: r = fa / fb; break;
    case OP_SUB: r = fa - fb; break;
    case OP_ADD: r = fa + fb; break;
    default: r = fa * fb; break;
  }
  return round_to((double)r, dt);
}

/* tensor.to(dtype) for a value coming from a float compute stage or an integer tensor. */
static double cast_to(double v, int src_dt, int dst_dt) {
  (void)src_dt;
  if (dt_is_float(dst_dt)) return round_to(v, dst_dt);
  return v;
}

/*
 * Store a float-stage value into an output buffer. For integer outputs this is a C cast of an
 * integer-valued, in-range number; NaN follows the x86 conversion the reference's CPU path
 * executes (cvttss2si gives INT_MIN; narrower types keep its low bits, i.e. 0).
 */
static void st(void* p, int dt, int64_t i, double v) {
  switch (dt) {
    case FFQ_F32: ((float*)p)[i] = (float)v; break;
    case FFQ_BF16: ((uint16_t*)p)[i] = f32_to_bf16((float)v); break;
    case FFQ_F16: ((uint16_t*)p)[i] = f32_to_f16((float)v); break;
    case FFQ_F64: ((double*)p)[i] = v; break;
    case FFQ_I8: ((int8_t*)p)[i] = (v != v) ? 0 : (int8_t)(int64_t)v; break;
    case FFQ_I16: ((int16_t*)p)[i] = (v != v) ? 0 : (int16_t)(int64_t)v; break;
    case FFQ_I32: ((int32_t*)p)[i] = (v != v) ? INT32_MIN : (int32_t)(int64_t)v; break;
    case FFQ_I64: ((int64_t*)p)[i] = (v != v) ? INT64_MIN : (int64_t)v; break;
    case FFQ_U8: ((uint8_t*)p)[i] = (v != v) ? 0 : (uint8_t)(int64_t)v; break;
  }
}

/* torch.result_type for two tensors with dim >= 1 (c10::promoteTypes). */
int ffq_promote_types(int a, int b) {
  if (!dt_valid(a) || !dt_valid(b)) return -FFQ_ERR_ARG;
  if (a == b) return a;
  int fa = dt_is_float(a), fb = dt_is_float(b);
  if (fa && !fb) return a;
  if (fb && !fa) return b;
  if (fa && fb) {
    if (a == FFQ_F64 || b == FFQ_F64) return FFQ_F64;
    if (a == FFQ_F32 || b == FFQ_F32) return FFQ_F32;
    return FFQ_F32; /* bf16 x f16 */
  }
  /* both integer */
  if (a == FFQ_I64 || b == FFQ_I64) return FFQ_I64;
  if (a == FFQ_I32 || b == FFQ_I32) return FFQ_I32;
  if (a == FFQ_I16 || b == FFQ_I16) return FFQ_I16;
  return FFQ_I16; /* i8 x u8 */
}

/* can_support_bitwidth, quantization/_quantizer_impl.py:44-75 */
int ffq_can_support_bitwidth(int dtype, double num_bits) {
  double avail;
  switch (dtype) {
    case FFQ_BF16: avail = 7; break;
    case FFQ_F16: avail = 10; break;
    case FFQ_F32: avail = 23; break;
    case FFQ_F64: avail = 52; break;
    case FFQ_I8: case FFQ_U8: avail = 8; break;
    case FFQ_I16: avail = 16; break;
    case FFQ_I32: avail = 32; break;
    case FFQ_I64: avail = 64; break;
    default: return 0;
  }
  return (avail + 2) >= num_bits;
}

/* ------------------------------------------------------------------------------------------ */
/* tile layout: quantization/tiled_tensor.py                                                  */
/* ------------------------------------------------------------------------------------------ */

/* check_tile_compatibility, tiled_tensor.py:19-42 (rank equality is implied by the struct). */
static int check_tiling(const ffq_tiling* t) {
  if (!t) return fail(FFQ_ERR_ARG, "tiling is NULL");
  if (t->ndim < 0 || t->ndim > FFQ_MAX_DIMS)
    return fail(FFQ_ERR_TILE_RANK, "tiling rank %d outside [0, %d]", t->ndim, FFQ_MAX_DIMS);
  for (int i = 0; i < t->ndim; ++i) {
    if (t->shape[i] < 0) return fail(FFQ_ERR_ARG, "negative extent");
    if (t->tile[i] > 0 && t->shape[i] % t->tile[i] != 0)
      return fail(FFQ_ERR_TILE_DIVIDE,
                  "Each dimension of tile_size must divide the corresponding input dimension. Got "
                  "%lld and %lld for dimension %d.",
                  (long long)t->shape[i], (long long)t->tile[i], i);
    if (t->tile[i] <= 0 && t->shape[i] != 0)
      return fail(FFQ_ERR_TILE_DIVIDE, "tile extent %lld for dimension %d", (long long)t->tile[i], i);
  }
  return FFQ_OK;
}

static int64_t numel_of(const ffq_tiling* t) {
  int64_t n = 1;
  for (int i = 0; i < t->ndim; ++i) n *= t->shape[i];
  return n;
}

int64_t ffq_num_tiles(const ffq_tiling* t) {
  int rc = check_tiling(t);
  if (rc) return -rc;
  int64_t n = 1;
  for (int i = 0; i < t->ndim; ++i) {
    if (t->shape[i] == 0) return 1; /* tiles_to_rows on empty data: reshape(1, 0), :87-88 */
    n *= t->shape[i] / t->tile[i];
  }
  return n;
}

/*
 * Row index of every element under tiles_to_rows (tiled_tensor.py:90-98): the data is reshaped to
 * [n0, t0, n1, t1, ...], permuted to [n0, n1, ..., t0, t1, ...] and flattened to
 * [num_tiles, tile_numel]; so element (i0, i1, ...) lands in row sum_k (i_k / t_k) * stride_k with
 * the strides of the row-major tile grid. An odometer walks the elements in memory order.
 */
typedef struct {
  int nd;
  int64_t idx[FFQ_MAX_DIMS];
  int64_t gstride[FFQ_MAX_DIMS];
  const ffq_tiling* t;
} walker;

static void walker_init(walker* w, const ffq_tiling* t) {
  w->nd = t->ndim;
  w->t = t;
  int64_t s = 1;
  for (int k = t->ndim - 1; k >= 0; --k) {
    w->idx[k] = 0;
    w->gstride[k] = s;
    s *= t->shape[k] / t->tile[k];
  }
}
static int64_t walker_tile(const walker* w) {
  int64_t r = 0;
  for (int k = 0; k < w->nd; ++k) r += (w->idx[k] / w->t->tile[k]) * w->gstride[k];
  return r;
}
static void walker_next(walker* w) {
  for (int k = w->nd - 1; k >= 0; --k) {
    if (++w->idx[k] < w->t->shape[k]) return;
    w->idx[k] = 0;
  }
}

/* Broadcast rule of `scale[:, None]` against the [num_tiles, tile_numel] rows. */
static int check_param_numel(const char* what, int64_t numel, int64_t ntiles) {
  if (numel == ntiles || numel == 1) return FFQ_OK;
  if (ntiles == 1)
    return fail(FFQ_ERR_PARAM_ROWS, "tiled_data is expected to be of size (1, L) but %s has %lld entries",
                what, (long long)numel);
  return fail(FFQ_ERR_PARAM_NUMEL,
              "The size of tensor a (%lld) must match the size of tensor b (%lld) at non-singleton "
              "dimension 0 (%s vs number of tiles)",
              (long long)ntiles, (long long)numel, what);
}

/* torch.round on one element of dtype dt: half-to-even for floats, identity for integers. */
static double round_half_even(double v, int dt) {
  if (!dt_is_float(dt)) return v;
  return nearbyint(v); /* default rounding mode = to nearest even; exact in every float dtype */
}

/* torch.clamp(x, lo, hi) on a float tensor: NaN propagates (ATen clamp_kernel). */
static double clamp_nan(double v, double lo, double hi) {
  if (v != v) return v;
  if (v < lo) v = lo;
  if (v > hi) v = hi;
  return v;
}

/* ------------------------------------------------------------------------------------------ */
/* A1: quantize_by_tile_impl, quantization/_quantizer_impl.py:144-169                          */
/* ------------------------------------------------------------------------------------------ */
int ffq_quantize_by_tile(const void* data, int data_dt, const void* scale, int scale_dt,
                         int64_t scale_numel, const void* offset, int offset_dt,
                         int64_t offset_numel, const ffq_tiling* tiling, double num_bits, void* out,
                         int out_dt, void* stream) {
  (void)stream;
  int rc = check_tiling(tiling);
  if (rc) return rc;
  if (!dt_valid(data_dt) || !dt_valid(scale_dt) || !dt_valid(out_dt) || (offset && !dt_valid(offset_dt)))
    return fail(FFQ_ERR_ARG, "bad dtype tag");
  int64_t n = numel_of(tiling);
  int64_t ntiles = ffq_num_tiles(tiling);
  if (n != 0) { /* an empty tensor broadcasts against anything and comes back empty */
    if ((rc = check_param_numel("scale", scale_numel, ntiles))) return rc;
    if (offset && (rc = check_param_numel("offset", offset_numel, ntiles))) return rc;
  }

  /* offset = round(offset) if given else zeros_like(scale)                      (:140-141,155) */
  int off_dt = offset ? offset_dt : scale_dt;
  /* min_threshold = -(2 ** (num_bits - 1)); max_threshold = -min_threshold - 1   (:158-159) */
  double lo = -pow(2.0, num_bits - 1.0), hi = -lo - 1.0;
  /* row / scale[:, None] is evaluated in result_type(data, scale); "- offset[:, None]" in
     result_type(of that, offset)                                                 (:161) */
  int div_dt = ffq_promote_types(data_dt, scale_dt);
  if (!dt_is_float(div_dt)) div_dt = FFQ_F32; /* true division of integers yields default float */
  int sub_dt = ffq_promote_types(div_dt, off_dt);
  /* output_dtype or result.dtype; the precision guard                            (:164-167) */
  if (!ffq_can_support_bitwidth(out_dt, num_bits))
    return fail(FFQ_ERR_PRECISION, "Provided dtype (%d) is not enough to store %g bits quantized values.",
                out_dt, num_bits);
  if (n == 0) return FFQ_OK;
  if (!data || !scale || !out) return fail(FFQ_ERR_ARG, "NULL buffer");
  /* clamp's scalar bounds are converted to the tensor dtype */
  double lo_c = round_to(lo, sub_dt), hi_c = round_to(hi, sub_dt);

  walker w;
  walker_init(&w, tiling);
  for (int64_t i = 0; i < n; ++i, walker_next(&w)) {
    int64_t t = walker_tile(&w);
    double x = cast_to(ld(data, data_dt, i), data_dt, div_dt);
    double s = cast_to(ld(scale, scale_dt, scale_numel == 1 ? 0 : t), scale_dt, div_dt);
    double q = op2(OP_DIV, x, s, div_dt);
    double o = offset ? round_half_even(ld(offset, offset_dt, offset_numel == 1 ? 0 : t), offset_dt) : 0.0;
    q = op2(OP_SUB, cast_to(q, div_dt, sub_dt), cast_to(o, off_dt, sub_dt), sub_dt);
    q = round_half_even(q, sub_dt);      /* round_ste == torch.round forward, ste.py:96      */
    q = clamp_nan(q, lo_c, hi_c);        /* torch.clamp                                 (:162) */
    st(out, out_dt, i, q);               /* result.to(output_dtype)                     (:168) */
  }
  return FFQ_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* A2: dequantize_by_tile_impl, quantization/_quantizer_impl.py:172-190                        */
/* ------------------------------------------------------------------------------------------ */
int ffq_dequantize_result_dtype(int data_dt, int scale_dt, int offset_dt, int has_offset) {
  int off_dt = has_offset ? offset_dt : scale_dt;
  int add_dt = ffq_promote_types(data_dt, off_dt);
  if (add_dt < 0) return add_dt;
  return ffq_promote_types(add_dt, scale_dt);
}

int ffq_dequantize_by_tile(const void* data, int data_dt, const void* scale, int scale_dt,
                           int64_t scale_numel, const void* offset, int offset_dt,
                           int64_t offset_numel, const ffq_tiling* tiling, void* out, int out_dt,
                           void* stream) {
  (void)stream;
  int rc = check_tiling(tiling);
  if (rc) return rc;
  if (!dt_valid(data_dt) || !dt_valid(scale_dt) || !dt_valid(out_dt) || (offset && !dt_valid(offset_dt)))
    return fail(FFQ_ERR_ARG, "bad dtype tag");
  int64_t n = numel_of(tiling);
  int64_t ntiles = ffq_num_tiles(tiling);
  if (n != 0) {
    if ((rc = check_param_numel("scale", scale_numel, ntiles))) return rc;
    if (offset && (rc = check_param_numel("offset", offset_numel, ntiles))) return rc;
  }
  int off_dt = offset ? offset_dt : scale_dt;
  /* (row + offset[:, None]) * scale[:, None]                                      (:182) */
  int add_dt = ffq_promote_types(data_dt, off_dt);
  int mul_dt = ffq_promote_types(add_dt, scale_dt);
  if (!dt_is_float(mul_dt)) return fail(FFQ_ERR_DTYPE, "integer-only dequantize is not built");
  if (n == 0) return FFQ_OK;
  if (!data || !scale || !out) return fail(FFQ_ERR_ARG, "NULL buffer");

  walker w;
  walker_init(&w, tiling);
  for (int64_t i = 0; i < n; ++i, walker_next(&w)) {
    int64_t t = walker_tile(&w);
    double q = cast_to(ld(data, data_dt, i), data_dt, add_dt);
    double o = offset ? round_half_even(ld(offset, offset_dt, offset_numel == 1 ? 0 : t), offset_dt) : 0.0;
    double v;
    if (dt_is_float(add_dt)) {
      v = op2(OP_ADD, q, cast_to(o, off_dt, add_dt), add_dt);
    } else {
      v = q + o; /* integer add: exact for the code ranges involved */
    }
    double s = cast_to(ld(scale, scale_dt, scale_numel == 1 ? 0 : t), scale_dt, mul_dt);
    v = op2(OP_MUL, cast_to(v, add_dt, mul_dt), s, mul_dt);
    st(out, out_dt, i, cast_to(v, mul_dt, out_dt)); /* .to(output_dtype)               (:184-185) */
  }
  return FFQ_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* A4: RunningMinMaxEstimator.estimate_step, range_setting/minmax.py:227-237                   */
/* ------------------------------------------------------------------------------------------ */
size_t ffq_minmax_workspace_bytes(const ffq_tiling* tiling, int data_dt) {
  (void)tiling; (void)data_dt;
  return 0;
}

/* torch.min / torch.max (reduction and elementwise) propagate NaN. */
static double min_nan(double a, double b) { return (a != a || b != b) ? NAN : (b < a ? b : a); }
static double max_nan(double a, double b) { return (a != a || b != b) ? NAN : (b > a ? b : a); }

static int minmax_core(const void* data, int data_dt, const ffq_tiling* tiling, double* mn, double* mx) {
  int64_t n = numel_of(tiling);
  int64_t ntiles = ffq_num_tiles(tiling);
  for (int64_t t = 0; t < ntiles; ++t) { mn[t] = INFINITY; mx[t] = -INFINITY; }
  walker w;
  walker_init(&w, tiling);
  for (int64_t i = 0; i < n; ++i, walker_next(&w)) {
    int64_t t = walker_tile(&w);
    double v = ld(data, data_dt, i);
    mn[t] = min_nan(mn[t], v);   /* torch.min(reshaped_data, -1).values            (:229) */
    mx[t] = max_nan(mx[t], v);   /* torch.max(reshaped_data, -1).values            (:230) */
  }
  return FFQ_OK;
}

int ffq_minmax_by_tile(const void* data, int data_dt, const ffq_tiling* tiling, void* min_inout,
                       void* max_inout, int accumulate, int32_t* status_flags, void* workspace,
                       size_t workspace_bytes, int32_t* ticket, void* stream) {
  (void)workspace; (void)workspace_bytes; (void)ticket; (void)stream;
  int rc = check_tiling(tiling);
  if (rc) return rc;
  if (!dt_valid(data_dt)) return fail(FFQ_ERR_ARG, "bad dtype tag");
  int64_t n = numel_of(tiling);
  if (n == 0) return fail(FFQ_ERR_EMPTY, "min/max of an empty tensor");
  if (!data || !min_inout || !max_inout) return fail(FFQ_ERR_ARG, "NULL buffer");
  int64_t ntiles = ffq_num_tiles(tiling);
  double* mn = (double*)malloc(sizeof(double) * (size_t)ntiles * 2);
  if (!mn) return fail(FFQ_ERR_ARG, "out of memory");
  double* mx = mn + ntiles;
  minmax_core(data, data_dt, tiling, mn, mx);
  int32_t flags = 0;
  for (int64_t t = 0; t < ntiles; ++t) {
    /* data_min.isinf().any() or data_max.isinf().any()                            (:233) */
    if (isinf(mn[t]) || isinf(mx[t])) flags |= FFQ_FLAG_INF;
    if (mn[t] != mn[t] || mx[t] != mx[t]) flags |= FFQ_FLAG_NAN;
    double a = mn[t], b = mx[t];
    if (accumulate) {
      a = min_nan(ld(min_inout, data_dt, t), a);   /* torch.min(self.min, data_min)  (:236) */
      b = max_nan(ld(max_inout, data_dt, t), b);   /* torch.max(self.max, data_max)  (:237) */
    }
    st(min_inout, data_dt, t, a);
    st(max_inout, data_dt, t, b);
  }
  if (status_flags) *status_flags |= flags;
  free(mn);
  return FFQ_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* A5: parameters_for_range, quantization/affine/range.py:54-122                               */
/* ------------------------------------------------------------------------------------------ */
size_t ffq_parameters_for_range_workspace_bytes(int64_t ntiles, int symmetric, int allow_one_sided) {
  (void)ntiles; (void)symmetric; (void)allow_one_sided;
  return 0;
}

int ffq_parameters_for_range(const void* min_range, const void* max_range, int range_dt,
                             int64_t ntiles, double num_bits, int symmetric, int allow_one_sided,
                             void* scale_out, int scale_dt, void* offset_out, int offset_dt,
                             void* workspace, size_t workspace_bytes, void* stream) {
  (void)stream; (void)workspace; (void)workspace_bytes;
  if (!min_range || !max_range || !scale_out || ntiles <= 0) return fail(FFQ_ERR_ARG, "bad argument");
  if (!dt_valid(range_dt) || !dt_valid(scale_dt) || (offset_out && !dt_valid(offset_dt)))
    return fail(FFQ_ERR_ARG, "bad dtype tag");
  /* min_range.to(float32), max_range.to(float32)                                    (:90) */
  /* one_sided = min_range.min() >= 0 and allow_one_sided                            (:100) */
  double gmin = INFINITY;
  for (int64_t t = 0; t < ntiles; ++t) gmin = min_nan(gmin, (double)(float)ld(min_range, range_dt, t));
  int one_sided = (gmin >= 0.0) && allow_one_sided;
  /* int_min = -(2 ** (num_bits - 1)); integer_maximum = -int_min - 1                 (:9-28) */
  double int_min = -pow(2.0, num_bits - 1.0), int_max = -int_min - 1.0;
  for (int64_t t = 0; t < ntiles; ++t) {
    double lo = (double)(float)ld(min_range, range_dt, t);
    double hi = (double)(float)ld(max_range, range_dt, t);
    if (symmetric && one_sided) lo = 0.0;                      /* zeros_like(min_range) (:104-105) */
    if (symmetric && !one_sided) {
      /* neg_scale = |min| / |int_min|; pos_scale = |max| / |int_max|; max of both    (:107-111) */
      double neg = op2(OP_DIV, fabs(lo), round_to(fabs(int_min), FFQ_F32), FFQ_F32);
      double pos = op2(OP_DIV, fabs(hi), round_to(fabs(int_max), FFQ_F32), FFQ_F32);
      st(scale_out, scale_dt, t, cast_to(max_nan(neg, pos), FFQ_F32, scale_dt));
      /* offset is None -> the setter fills the buffer with 0   (nn/linear_quantizer.py:353-357) */
      if (offset_out) st(offset_out, offset_dt, t, 0.0);
    } else {
      /* num_steps = 2**num_bits - 1; scale = (max - min) / num_steps, clamped at eps   (:117-121) */
      double num_steps = pow(2.0, num_bits) - 1.0;
      double interval = op2(OP_SUB, hi, lo, FFQ_F32);
      double sc = op2(OP_DIV, interval, round_to(num_steps, FFQ_F32), FFQ_F32);
      if (sc == sc && sc < (double)FLT_EPSILON) sc = (double)FLT_EPSILON;
      /* offset = min_range / scale - int_min                                          (:122) */
      double of = op2(OP_SUB, op2(OP_DIV, lo, sc, FFQ_F32), round_to(int_min, FFQ_F32), FFQ_F32);
      st(scale_out, scale_dt, t, cast_to(sc, FFQ_F32, scale_dt));
      if (offset_out) st(offset_out, offset_dt, t, cast_to(of, FFQ_F32, offset_dt));
    }
  }
  return FFQ_OK;
}

/* One RunningMinMaxEstimator.estimate_step (range_setting/minmax.py:215-239): running min / max merged (:236-237), then the range
 * setter nn/linear_quantizer.py:350-357 = parameters_for_range (affine/range.py:54-122) on the merged range — the composition of
 * the two restatements above, as the reference composes the two steps. */
int ffq_running_minmax_step(const void* data, int data_dt, const ffq_tiling* tiling, void* min_inout, void* max_inout,
                            int32_t* status_flags, double num_bits, int symmetric, int allow_one_sided, void* scale_out,
                            int scale_dt, void* offset_out, int offset_dt, void* workspace, size_t workspace_bytes,
                            int32_t* ticket, void* stream) {
  int rc = ffq_minmax_by_tile(data, data_dt, tiling, min_inout, max_inout, 1, status_flags, workspace, workspace_bytes, ticket, stream);
  if (rc) return rc;
  return ffq_parameters_for_range(min_inout, max_inout, data_dt, ffq_num_tiles(tiling), num_bits, symmetric, allow_one_sided, scale_out,
                                  scale_dt, offset_out, offset_dt, NULL, 0, stream);
}

/* range_setting/common.py:218-238 for a RunningMinMax estimator: estimate_step (above), then the quantizer's own forward = A1 with
 * the parameters the step just wrote — the composition of the restatements, as the reference composes the calls */
int ffq_running_minmax_quantize(const void* data, int data_dt, const ffq_tiling* tiling, void* min_inout, void* max_inout,
                                int32_t* status_flags, double num_bits, int symmetric, int allow_one_sided, float* scale_out,
                                float* offset_out, void* out, int out_dt, int32_t* ticket, void* stream) {
  (void)ticket;
  const int64_t ntiles = ffq_num_tiles(tiling);
  if (ntiles < 0) return (int)-ntiles;
  int rc = ffq_running_minmax_step(data, data_dt, tiling, min_inout, max_inout, status_flags, num_bits, symmetric, allow_one_sided, scale_out,
                                   FFQ_F32, offset_out, FFQ_F32, NULL, 0, NULL, stream);
  if (rc) return rc;
  return ffq_quantize_by_tile(data, data_dt, scale_out, FFQ_F32, ntiles, offset_out, FFQ_F32, ntiles, tiling, num_bits, out, out_dt, stream);
}

/* ------------------------------------------------------------------------------------------ */
/* A3: quantize_dynamic_by_tile_impl, quantization/_quantizer_impl.py:243-285                  */
/* ------------------------------------------------------------------------------------------ */
size_t ffq_quantize_dynamic_workspace_bytes(const ffq_tiling* tiling, int data_dt) {
  (void)tiling; (void)data_dt;
  return 0;
}

int ffq_quantize_dynamic_by_tile(const void* data, int data_dt, const ffq_tiling* tiling,
                                 double num_bits, int symmetric, int allow_one_sided, void* out,
                                 int out_dt, float* scale_out, float* offset_out, void* workspace,
                                 size_t workspace_bytes, int32_t* ticket, void* stream) {
  (void)workspace; (void)workspace_bytes; (void)ticket;
  int rc = check_tiling(tiling);
  if (rc) return rc;
  int64_t n = numel_of(tiling);
  /* torch.min over an empty row raises IndexError -> QuantizationError             (:259-264) */
  if (n == 0) return fail(FFQ_ERR_EMPTY, "Cannot dynamically quantize an empty tensor");
  if (!scale_out || !offset_out) return fail(FFQ_ERR_ARG, "NULL parameter output");
  if (!ffq_can_support_bitwidth(out_dt, num_bits))
    return fail(FFQ_ERR_PRECISION, "Provided dtype (%d) is not enough to store %g bits quantized values.",
                out_dt, num_bits);
  int64_t ntiles = ffq_num_tiles(tiling);
  /* min_range / max_range live in the data dtype                                    (:257-258) */
  void* mn = malloc(dt_size(data_dt) * (size_t)ntiles * 2);
  if (!mn) return fail(FFQ_ERR_ARG, "out of memory");
  void* mx = (char*)mn + dt_size(data_dt) * (size_t)ntiles;
  rc = ffq_minmax_by_tile(data, data_dt, tiling, mn, mx, 0, NULL, NULL, 0, NULL, stream);
  /* parameters_for_range(...); offset None -> zeros_like(scale); offset = round(offset) (:266-275) */
  if (!rc)
    rc = ffq_parameters_for_range(mn, mx, data_dt, ntiles, num_bits, symmetric, allow_one_sided,
                                  scale_out, FFQ_F32, offset_out, FFQ_F32, NULL, 0, stream);
  free(mn);
  if (rc) return rc;
  for (int64_t t = 0; t < ntiles; ++t) offset_out[t] = nearbyintf(offset_out[t]);
  /* round(row / scale - offset), clamp, cast                                        (:277-284) */
  return ffq_quantize_by_tile(data, data_dt, scale_out, FFQ_F32, ntiles, offset_out, FFQ_F32, ntiles,
                              tiling, num_bits, out, out_dt, stream);
}

/* ------------------------------------------------------------------------------------------ */
/* A7: GGUF Q4_0 nibble order, export/stages/gguf/_packing.py:44-53                            */
/* ------------------------------------------------------------------------------------------ */
int ffq_pack_int4(const void* codes, int codes_dt, int64_t numel, int64_t block, uint8_t* packed,
                  void* stream) {
  (void)stream;
  if (!dt_valid(codes_dt)) return fail(FFQ_ERR_ARG, "bad dtype tag");
  if (block <= 0 || (block & 1) || numel % block) return fail(FFQ_ERR_ARG, "numel %% block != 0 or odd block");
  if (numel == 0) return FFQ_OK;
  if (!codes || !packed) return fail(FFQ_ERR_ARG, "NULL buffer");
  int64_t half = block / 2;
  for (int64_t b = 0; b < numel / block; ++b) {
    for (int64_t j = 0; j < half; ++j) {
      /* gguf_qs = (int_codes + 8).clamp(0, 15); packed = qs[:, 0, :] | (qs[:, 1, :] << 4) */
      int64_t lo = (int64_t)ld(codes, codes_dt, b * block + j) + 8;
      int64_t hi = (int64_t)ld(codes, codes_dt, b * block + half + j) + 8;
      lo = lo < 0 ? 0 : (lo > 15 ? 15 : lo);
      hi = hi < 0 ? 0 : (hi > 15 ? 15 : hi);
      packed[b * half + j] = (uint8_t)(lo | (hi << 4));
    }
  }
  return FFQ_OK;
}

int ffq_unpack_int4(const uint8_t* packed, int64_t numel, int64_t block, void* codes_out,
                    int codes_dt, void* stream) {
  (void)stream;
  if (!dt_valid(codes_dt)) return fail(FFQ_ERR_ARG, "bad dtype tag");
  if (block <= 0 || (block & 1) || numel % block) return fail(FFQ_ERR_ARG, "numel %% block != 0 or odd block");
  if (numel == 0) return FFQ_OK;
  if (!codes_out || !packed) return fail(FFQ_ERR_ARG, "NULL buffer");
  int64_t half = block / 2;
  for (int64_t b = 0; b < numel / block; ++b) {
    for (int64_t j = 0; j < half; ++j) {
      uint8_t byte = packed[b * half + j];
      st(codes_out, codes_dt, b * block + j, (double)((int)(byte & 15) - 8));
      st(codes_out, codes_dt, b * block + half + j, (double)((int)(byte >> 4) - 8));
    }
  }
  return FFQ_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* A6: fallback.linear, _gen/fallback.py:77-112                                                */
/* ------------------------------------------------------------------------------------------ */
size_t ffq_linear_w8a8_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  (void)M; (void)N; (void)K;
  return 0;
}

/*
 * The reference dequantizes both operands into their dequantize_dtype (A2), runs
 * torch.nn.functional.linear on them and applies the output quantizer. Restated here with the
 * operands dequantized to `out_dt` (the activation dtype of the caller) and the contraction
 * accumulated in double, i.e. the exact value a float GEMM approximates; tests compare with the
 * tolerance stated there.
 */
static int check_rowsum(const char* what, const int8_t* wq, const int32_t* rowsum, int64_t N, int64_t K) {
  if (!rowsum) return FFQ_OK;
  for (int64_t n = 0; n < N; ++n) {
    int32_t sum = 0;
    for (int64_t k = 0; k < K; ++k) sum += wq[n * K + k];
    if (sum != rowsum[n]) return fail(FFQ_ERR_ARG, "%s row sum %lld is %d, the codes sum to %d", what, (long long)n, rowsum[n], sum);
  }
  return FFQ_OK;
}

int ffq_linear_w8a8(const int8_t* xq, const int8_t* wq, const int32_t* w_rowsum, const float* x_scale,
                    const float* x_offset, int x_per_row, const float* w_scale, const float* w_offset, int w_per_row,
                    const void* bias, int bias_dt, void* out, int out_dt, const float* out_scale,
                    const float* out_offset, double out_num_bits, int y_dt, int64_t M, int64_t N, int64_t K,
                    void* workspace, size_t workspace_bytes, void* stream) {
  (void)workspace; (void)workspace_bytes; (void)stream;
  if (M < 0 || N < 0 || K < 0) return fail(FFQ_ERR_ARG, "negative extent");
  if (M == 0 || N == 0) return FFQ_OK;
  if (!xq || !wq || !x_scale || !w_scale || !out) return fail(FFQ_ERR_ARG, "NULL buffer");
  if (K > 0) {  /* sums handed in by the caller must be the sums of the codes */
    int rc = check_rowsum("weight", wq, w_rowsum, N, K);
    if (rc) return rc;
  }
  /* with an output quantizer, y_dt is the dtype F.linear returns (the input's dequantize dtype, nn/linear.py:32-39) */
  int deq_dt = out_scale ? y_dt : out_dt;
  if (!dt_is_float(deq_dt)) return fail(FFQ_ERR_DTYPE, "real-valued output must be a float dtype");
  if (out_scale && !ffq_can_support_bitwidth(out_dt, out_num_bits))
    return fail(FFQ_ERR_PRECISION, "Provided dtype (%d) is not enough to store %g bits quantized values.",
                out_dt, out_num_bits);
  double lo = -pow(2.0, out_num_bits - 1.0), hi = -lo - 1.0;
  double* xr = (double*)malloc(sizeof(double) * (size_t)(K > 0 ? K : 1));
  if (!xr) return fail(FFQ_ERR_ARG, "out of memory");
  for (int64_t m = 0; m < M; ++m) {
    double sx = x_scale[x_per_row ? m : 0];
    double ox = x_offset ? (double)nearbyintf(x_offset[x_per_row ? m : 0]) : 0.0;
    for (int64_t k = 0; k < K; ++k) {
      /* input.dequantize(): (q + o) * s in fp32, cast to the dequantize dtype            (A2) */
      double v = op2(OP_MUL, op2(OP_ADD, (double)xq[m * K + k], ox, FFQ_F32), sx, FFQ_F32);
      xr[k] = cast_to(v, FFQ_F32, deq_dt);
    }
    for (int64_t n = 0; n < N; ++n) {
      double sw = w_scale[w_per_row ? n : 0];
      double ow = w_offset ? (double)nearbyintf(w_offset[w_per_row ? n : 0]) : 0.0;
      double acc = 0.0;
      for (int64_t k = 0; k < K; ++k) {
        double wv = op2(OP_MUL, op2(OP_ADD, (double)wq[n * K + k], ow, FFQ_F32), sw, FFQ_F32);
        acc += xr[k] * cast_to(wv, FFQ_F32, deq_dt);
      }
      if (bias) acc += ld(bias, bias_dt, n);
      double y = round_to(acc, deq_dt);
      if (out_scale) {
        /* output_quantizer(output): per-tensor A1 on the real-valued result (fallback.py:110-111) */
        double o = out_offset ? (double)nearbyintf(out_offset[0]) : 0.0;
        double q = op2(OP_SUB, op2(OP_DIV, y, (double)out_scale[0], FFQ_F32), o, FFQ_F32);
        q = clamp_nan(nearbyint(q), lo, hi);
        st(out, out_dt, m * N + n, q);
      } else {
        st(out, out_dt, m * N + n, y);
      }
    }
  }
  free(xr);
  return FFQ_OK;
}

/*
 * A6, weight-only: fallback.linear with a quantized weight and a plain input (_gen/fallback.py:86-112, strict
 * quantization off): weight.dequantize() — A2 into the weight's dequantize dtype, here the activation dtype — then
 * torch.nn.functional.linear. The contraction is accumulated in double (the exact value of the same operands; a float
 * GEMM's result depends on its summation order), bias added, rounded once to `out_dt`.
 */
int ffq_linear_wq_supported(int x_dt, int w_dt, int out_dt, int64_t M, int64_t N, int64_t K, int64_t group, int64_t pack_block) {
  (void)M; (void)N;
  if (!(x_dt == FFQ_BF16 || x_dt == FFQ_F32 || x_dt == FFQ_F16) || !dt_is_float(out_dt)) return 0;
  if (!(group > 0 && K % group == 0)) return 0;
  if (w_dt == FFQ_I8) return pack_block == 0;
  if (w_dt == FFQ_U8) return pack_block >= 2 && pack_block % 2 == 0 && K % pack_block == 0;
  return 0;
}

size_t ffq_linear_wq_workspace_bytes(int64_t M, int64_t N, int64_t K) { (void)M; (void)N; (void)K; return 0; }
/* the restatement sums every output in double: no tiles, no K slices */
int64_t ffq_linear_wq_split(int64_t M, int64_t N, int64_t K, int mlp) { (void)M; (void)N; (void)K; (void)mlp; return 1; }
int64_t ffq_linear_wq_tickets(int64_t M, int64_t N, int64_t K, int mlp) { (void)M; (void)N; (void)K; (void)mlp; return 0; }
size_t ffq_linear_wq_slab_bytes(int64_t M, int64_t N, int64_t K, int mlp, int64_t split) { (void)M; (void)N; (void)K; (void)mlp; (void)split; return 0; }

/* code (n, k) of a weight stored one code per byte, or packed two per byte as ffq_pack_int4 writes a row of K codes with
 * block `pack_block` (_packing.py:44-53: byte j of a block = code j | code (j + block / 2) << 4, both + 8) */
static double wq_code(const void* w_codes, int64_t pack_block, int64_t n, int64_t k, int64_t K) {
  if (pack_block == 0) return (double)((const int8_t*)w_codes)[n * K + k];
  int64_t half = pack_block / 2, blk = k / pack_block, within = k % pack_block;
  uint8_t byte = ((const uint8_t*)w_codes)[n * (K / 2) + blk * half + within % half];
  return (double)((int)((within / half) ? (byte >> 4) : (byte & 15)) - 8);
}

int ffq_linear_wq(const void* x, int x_dt, const void* w_codes, int w_dt, int64_t pack_block, const float* w_scale,
                  const float* w_offset, int64_t scale_numel, int64_t group, const void* bias, int bias_dt, void* out,
                  int out_dt, int64_t M, int64_t N, int64_t K, void* workspace, size_t workspace_bytes, int32_t* tickets,
                  int64_t split, void* stream) {
  (void)stream; (void)workspace; (void)workspace_bytes; (void)tickets; (void)split;
  if (M < 0 || N < 0 || K < 0) return fail(FFQ_ERR_ARG, "negative extent");
  if (M == 0 || N == 0) return FFQ_OK;
  if (!x || !w_codes || !w_scale || !out) return fail(FFQ_ERR_ARG, "NULL buffer");
  if (!ffq_linear_wq_supported(x_dt, w_dt, out_dt, M, N, K, group, pack_block)) return fail(FFQ_ERR_DTYPE, "weight-only linear: unsupported dtypes / group / packing");
  int64_t groups = K / group;
  if (!(scale_numel == 1 || scale_numel == N * groups))
    return fail(FFQ_ERR_PARAM_NUMEL, "weight-only linear: %lld parameters for %lld x %lld tiles", (long long)scale_numel, (long long)N, (long long)groups);
  double* wr = (double*)malloc(sizeof(double) * (size_t)(K > 0 ? K : 1));
  if (!wr) return fail(FFQ_ERR_ARG, "out of memory");
  for (int64_t n = 0; n < N; ++n) {
    for (int64_t k = 0; k < K; ++k) {
      int64_t t = scale_numel == 1 ? 0 : n * groups + k / group;
      double o = w_offset ? (double)nearbyintf(w_offset[t]) : 0.0;
      /* weight.dequantize(): (q + o) * s in fp32, cast to the dequantize dtype                       (A2) */
      double v = op2(OP_MUL, op2(OP_ADD, wq_code(w_codes, pack_block, n, k, K), o, FFQ_F32), (double)w_scale[t], FFQ_F32);
      wr[k] = cast_to(v, FFQ_F32, x_dt);
    }
    for (int64_t m = 0; m < M; ++m) {
      double acc = 0.0;
      for (int64_t k = 0; k < K; ++k) acc += ld(x, x_dt, m * K + k) * wr[k];
      if (bias) acc += ld(bias, bias_dt, n);
      st(out, out_dt, m * N + n, round_to(acc, out_dt));
    }
  }
  free(wr);
  return FFQ_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* Producer-fused A1 (ABI version 2): the three elementwise producers of the reference's       */
/* quantized Llama helpers, docs/examples/doc_helpers/quantized_llama/, each followed by A1.    */
/* Where the eager chain's result depends on the platform (the order of the fp32 sum in        */
/* mean(), the last bit of rsqrt and exp) this file takes the most accurate reading (double     */
/* sum, 1/sqrtf, expf); tests compare the bf16 results with a one-ulp tolerance and the codes   */
/* through A1 of the values actually produced.                                                  */
/* ------------------------------------------------------------------------------------------ */
static float bf16_round(float v) { return bf16_to_f32(f32_to_bf16(v)); }

static int fan_check(const ffq_fanout* fan, double* lo, double* hi) {
  *lo = *hi = 0.0;
  if (!fan || fan->count == 0) return FFQ_OK;
  if (fan->count < 0 || fan->count > FFQ_MAX_FANOUT) return fail(FFQ_ERR_ARG, "fan-out count must be 0..%d", FFQ_MAX_FANOUT);
  if (!(fan->num_bits >= 1 && fan->num_bits <= 8 && fan->num_bits == floor(fan->num_bits)))
    return fail(FFQ_ERR_PRECISION, "Provided dtype (%d) is not enough to store %g bits quantized values.", FFQ_I8, fan->num_bits);
  for (int j = 0; j < fan->count; ++j)
    if (!fan->scale[j] || !fan->codes[j]) return fail(FFQ_ERR_ARG, "NULL scale / codes in fan-out %d", j);
  *lo = -pow(2.0, fan->num_bits - 1.0);
  *hi = -*lo - 1.0;
  return FFQ_OK;
}

/* A1 on one bf16-valued element for every quantizer of the fan-out (_quantizer_impl.py:154-169,
   bf16 data with fp32 parameters: division and subtraction are fp32 ops). */
static void fan_quantize(const ffq_fanout* fan, double lo, double hi, float z, int64_t i) {
  if (!fan) return;
  for (int j = 0; j < fan->count; ++j) {
    float s = fan->scale[j][0];
    float o = fan->offset[j] ? nearbyintf(fan->offset[j][0]) : 0.0f;
    float q = z / s;
    q = q - o;
    q = nearbyintf(q);
    double c = clamp_nan((double)q, lo, hi);
    st(fan->codes[j], FFQ_I8, i, c);
  }
}

/* rms_norm.py:17-35 (LlamaRMSNorm.forward) behind the residual add of decoder.py:60-90 */
int ffq_add_rmsnorm_quantize(const void* x, const void* delta, void* sum_out, const void* weight,
                             int dt, int64_t rows, int64_t cols, double eps, void* norm_out,
                             const ffq_fanout* fan, void* stream) {
  (void)stream;
  if (rows < 0 || cols < 0) return fail(FFQ_ERR_ARG, "negative extent");
  if (dt != FFQ_BF16) return fail(FFQ_ERR_DTYPE, "fused RMSNorm is built for bf16 activations");
  if (cols == 0) return fail(FFQ_ERR_EMPTY, "RMSNorm over an empty row");
  if (cols % 16 != 0 || cols > 8192) return fail(FFQ_ERR_DTYPE, "fused RMSNorm needs cols %% 16 == 0 and cols <= 8192 (got %lld)", (long long)cols);
  double lo, hi;
  int rc = fan_check(fan, &lo, &hi);
  if (rc) return rc;
  if (rows == 0) return FFQ_OK;
  if (!x || !weight) return fail(FFQ_ERR_ARG, "NULL buffer");
  const uint16_t* xs = (const uint16_t*)x;
  const uint16_t* ds = (const uint16_t*)delta;
  const uint16_t* ws = (const uint16_t*)weight;
  float* h = (float*)malloc((size_t)cols * sizeof(float));
  for (int64_t r = 0; r < rows; ++r) {
    double ss = 0.0;
    for (int64_t c = 0; c < cols; ++c) {
      float v = bf16_to_f32(xs[r * cols + c]);
      if (ds) {                                   /* hidden_states = residual + hidden_states (bf16 add) */
        v = bf16_round(v + bf16_to_f32(ds[r * cols + c]));
        if (sum_out) ((uint16_t*)sum_out)[r * cols + c] = f32_to_bf16(v);
      }
      h[c] = v;                                   /* hidden_states.to(torch.float32)             (:27) */
      float sq = v * v;                           /* .pow(2)                                     (:28) */
      ss += (double)sq;
    }
    float variance = (float)(ss / (double)cols);  /* .mean(-1, keepdim=True)                     (:28) */
    float r_std = 1.0f / sqrtf(variance + (float)eps); /* torch.rsqrt(variance + eps)             (:29) */
    for (int64_t c = 0; c < cols; ++c) {
      float n = bf16_round(h[c] * r_std);         /* (hidden * rsqrt).to(input_dtype)        (:29-30) */
      float z = bf16_round(bf16_to_f32(ws[c]) * n); /* self.weight * hidden (bf16 * bf16)         (:30) */
      if (norm_out) ((uint16_t*)norm_out)[r * cols + c] = f32_to_bf16(z);
      fan_quantize(fan, lo, hi, z, r * cols + c);
    }
  }
  free(h);
  return FFQ_OK;
}

/* mlp.py:30-40: down_proj(act_fn(gate_proj(x)) * up_proj(x)); ATen silu: x / (1 + exp(-x)) in fp32 */
int ffq_silu_mul_quantize(const void* gate, const void* up, int dt, int64_t numel, void* product_out,
                          const ffq_fanout* fan, void* stream) {
  (void)stream;
  if (numel < 0) return fail(FFQ_ERR_ARG, "negative extent");
  if (dt != FFQ_BF16) return fail(FFQ_ERR_DTYPE, "fused SiLU*up is built for bf16 activations");
  if (numel % 16 != 0) return fail(FFQ_ERR_DTYPE, "fused SiLU*up needs numel %% 16 == 0 and numel < 2^35");
  double lo, hi;
  int rc = fan_check(fan, &lo, &hi);
  if (rc) return rc;
  if (numel == 0) return FFQ_OK;
  if (!gate || !up) return fail(FFQ_ERR_ARG, "NULL buffer");
  const uint16_t* g = (const uint16_t*)gate;
  const uint16_t* u = (const uint16_t*)up;
  for (int64_t i = 0; i < numel; ++i) {
    float v = bf16_to_f32(g[i]);
    float act = bf16_round(v / (1.0f + expf(-v)));
    float z = bf16_round(act * bf16_to_f32(u[i]));
    if (product_out) ((uint16_t*)product_out)[i] = f32_to_bf16(z);
    fan_quantize(fan, lo, hi, z, i);
  }
  return FFQ_OK;
}

/* attention.py:20-41: (q * cos) + (rotate_half(q) * sin), rotate_half(x) = cat(-x[D/2:], x[:D/2]) */
int ffq_rope_inplace(void* q, int64_t q_heads, void* k, int64_t k_heads, int dt, int64_t tokens,
                     int64_t seq_len, int64_t head_dim, const void* cos_table, const void* sin_table,
                     void* stream) {
  (void)stream;
  if (tokens < 0 || q_heads < 0 || k_heads < 0 || seq_len <= 0 || head_dim <= 0) return fail(FFQ_ERR_ARG, "bad extent");
  if (dt != FFQ_BF16) return fail(FFQ_ERR_DTYPE, "fused rotary embedding is built for bf16 activations");
  if (head_dim % 16 != 0) return fail(FFQ_ERR_DTYPE, "fused rotary embedding needs head_dim %% 16 == 0");
  if (tokens % seq_len != 0) return fail(FFQ_ERR_ARG, "tokens must be a multiple of seq_len");
  if (tokens * (q_heads + k_heads) == 0) return FFQ_OK;
  if ((q_heads && !q) || (k_heads && !k) || !cos_table || !sin_table) return fail(FFQ_ERR_ARG, "NULL buffer");
  const uint16_t* ct = (const uint16_t*)cos_table;
  const uint16_t* stb = (const uint16_t*)sin_table;
  const int64_t half = head_dim / 2;
  float* tmp = (float*)malloc((size_t)head_dim * sizeof(float));
  for (int which = 0; which < 2; ++which) {
    uint16_t* base = (uint16_t*)(which ? k : q);
    int64_t heads = which ? k_heads : q_heads;
    for (int64_t t = 0; t < tokens; ++t) {
      int64_t pos = t % seq_len;
      for (int64_t hd = 0; hd < heads; ++hd) {
        uint16_t* row = base + (t * heads + hd) * head_dim;
        for (int64_t d = 0; d < head_dim; ++d) {
          float v = bf16_to_f32(row[d]);
          float rot = d < half ? -bf16_to_f32(row[d + half]) : bf16_to_f32(row[d - half]);
          float a = bf16_round(v * bf16_to_f32(ct[pos * head_dim + d]));
          float b = bf16_round(rot * bf16_to_f32(stb[pos * head_dim + d]));
          tmp[d] = bf16_round(a + b);
        }
        for (int64_t d = 0; d < head_dim; ++d) row[d] = f32_to_bf16(tmp[d]);
      }
    }
  }
  free(tmp);
  return FFQ_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* A8: quant_dequant_by_tile_grad_impl, quantization/_quantizer_impl.py:193-237                */
/* The per-tile sums are accumulated in double and rounded to fp32 once (the reference sums in  */
/* fp32 in ATen's order; tests state the tolerance).                                            */
/* ------------------------------------------------------------------------------------------ */
size_t ffq_quantize_backward_workspace_bytes(const ffq_tiling* tiling) { (void)tiling; return 0; }

int ffq_quantize_by_tile_backward(const void* data, const void* output_grad, int dt, const float* scale,
                                  int64_t scale_numel, const float* offset, int64_t offset_numel,
                                  const ffq_tiling* tiling, double num_bits, void* dinput, float* dscale,
                                  float* doffset, void* workspace, size_t workspace_bytes, void* stream) {
  (void)workspace; (void)workspace_bytes; (void)stream;
  int rc = check_tiling(tiling);
  if (rc) return rc;
  if (!(dt == FFQ_F32 || dt == FFQ_BF16 || dt == FFQ_F16)) return fail(FFQ_ERR_DTYPE, "backward is built for f32 / bf16 / f16 data");
  int64_t n = numel_of(tiling);
  int64_t ntiles = ffq_num_tiles(tiling);
  if (n != 0) {
    if ((rc = check_param_numel("scale", scale_numel, ntiles))) return rc;
    if (offset && (rc = check_param_numel("offset", offset_numel, ntiles))) return rc;
  }
  if (n == 0) return FFQ_OK;
  if (!data || !output_grad || !scale || !dinput || !dscale) return fail(FFQ_ERR_ARG, "NULL buffer");
  if (offset && !doffset) return fail(FFQ_ERR_ARG, "doffset is required when offset is given");
  if (scale_numel == 1 && ntiles != 1) return fail(FFQ_ERR_DTYPE, "backward kernel needs one scale per tile");
  float lo = (float)(-pow(2.0, num_bits - 1.0)), hi = -lo - 1.0f;          /* (:208-209) */
  double* acc_s = (double*)calloc((size_t)ntiles, sizeof(double));
  double* acc_o = (double*)calloc((size_t)ntiles, sizeof(double));
  walker w;
  walker_init(&w, tiling);
  for (int64_t i = 0; i < n; ++i, walker_next(&w)) {
    int64_t t = walker_tile(&w);
    float x = (float)ld(data, dt, i), g = (float)ld(output_grad, dt, i);
    float s = scale[t], o = offset ? nearbyintf(offset[t]) : 0.0f;   /* _infer_offset rounds   (:140-141,204) */
    float u = x / s;                                  /* data_as_rows / scale[:, None]          (:214) */
    u = u - o;                                        /*   - round_ste(offset[:, None])               */
    float q = nearbyintf(u);                          /* torch.round(pre_round)                 (:215) */
    int below = q < lo, above = q > hi, clip = below || above;           /* clip_mask  (:216) */
    st(dinput, dt, i, clip ? 0.0 : (double)g);        /* torch.where(clip_mask, 0, grad)        (:218) */
    if (offset) acc_o[t] += clip ? (double)(s * g) : 0.0;                /* (:223-224) */
    float bound = (below ? lo : hi) + o;              /* where(q < min, min, max) + offset  (:226-230) */
    float term = clip ? bound : q - u;                /* where(clip, dscale, q - pre_round)     (:231) */
    acc_s[t] += (double)(term * g);                   /* dscale.mul_(grad)                      (:232) */
  }
  for (int64_t t = 0; t < ntiles; ++t) {
    dscale[t] = (float)acc_s[t];
    if (offset) doffset[t] = (float)acc_o[t];
  }
  free(acc_s); free(acc_o);
  return FFQ_OK;
}

/* A1 + A7 / A7 + A2 composed (the fused kernels must equal the two-step path exactly) */
int ffq_quantize_pack_int4(const void* data, int data_dt, const float* scale, int64_t scale_numel,
                           const float* offset, int64_t offset_numel, const ffq_tiling* tiling, int64_t block,
                           uint8_t* packed, void* stream) {
  int rc = check_tiling(tiling);
  if (rc) return rc;
  int64_t n = numel_of(tiling);
  if (block <= 0 || (block & 1) || n % block) return fail(FFQ_ERR_ARG, "numel %% block != 0 or odd block");
  int8_t* codes = (int8_t*)malloc((size_t)(n ? n : 1));
  rc = ffq_quantize_by_tile(data, data_dt, scale, FFQ_F32, scale_numel, offset, FFQ_F32, offset_numel, tiling, 4.0, codes, FFQ_I8, stream);
  if (!rc) rc = ffq_pack_int4(codes, FFQ_I8, n, block, packed, stream);
  free(codes);
  return rc;
}

int ffq_unpack_dequantize_int4(const uint8_t* packed, const float* scale, int64_t scale_numel, const float* offset,
                               int64_t offset_numel, const ffq_tiling* tiling, int64_t block, void* out, int out_dt,
                               void* stream) {
  int rc = check_tiling(tiling);
  if (rc) return rc;
  int64_t n = numel_of(tiling);
  if (block <= 0 || (block & 1) || n % block) return fail(FFQ_ERR_ARG, "numel %% block != 0 or odd block");
  int8_t* codes = (int8_t*)malloc((size_t)(n ? n : 1));
  rc = ffq_unpack_int4(packed, n, block, codes, FFQ_I8, stream);
  if (!rc) rc = ffq_dequantize_by_tile(codes, FFQ_I8, scale, FFQ_F32, scale_numel, offset, FFQ_F32, offset_numel, tiling, out, out_dt, stream);
  free(codes);
  return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* Min-error grid search, range_setting/min_error.py:218-231 with mse_error (:62-72): per       */
/* candidate, quantize (A1) -> dequantize (A2, cast to the data dtype) -> (qdq - x) ** 2 in the  */
/* data dtype -> summed per tile (double accumulation, rounded to fp32 once).                    */
/* ------------------------------------------------------------------------------------------ */
size_t ffq_grid_sqerror_workspace_bytes(const ffq_tiling* tiling, int64_t ncand) { (void)tiling; (void)ncand; return 0; }

int ffq_grid_sqerror_by_tile(const void* data, int dt, const float* scales, const float* offsets, int64_t ncand,
                             const ffq_tiling* tiling, double num_bits, float* err, int accumulate, void* workspace,
                             size_t workspace_bytes, void* stream) {
  (void)workspace; (void)workspace_bytes; (void)stream;
  int rc = check_tiling(tiling);
  if (rc) return rc;
  if (ncand <= 0 || ncand > 4096) return fail(FFQ_ERR_ARG, "number of candidates must be 1..4096");
  if (!(dt == FFQ_F32 || dt == FFQ_BF16 || dt == FFQ_F16)) return fail(FFQ_ERR_DTYPE, "grid error is built for f32 / bf16 / f16 data");
  int64_t n = numel_of(tiling), ntiles = ffq_num_tiles(tiling);
  if (n == 0) return fail(FFQ_ERR_EMPTY, "grid error over an empty tensor");
  if (!data || !scales || !err) return fail(FFQ_ERR_ARG, "NULL buffer");
  float lo = (float)(-pow(2.0, num_bits - 1.0)), hi = -lo - 1.0f;
  double* acc = (double*)malloc((size_t)ntiles * sizeof(double));
  for (int64_t c = 0; c < ncand; ++c) {
    memset(acc, 0, (size_t)ntiles * sizeof(double));
    walker w;
    walker_init(&w, tiling);
    for (int64_t i = 0; i < n; ++i, walker_next(&w)) {
      int64_t t = walker_tile(&w);
      float x = (float)ld(data, dt, i);
      float s = scales[c * ntiles + t];
      float ro = offsets ? nearbyintf(offsets[c * ntiles + t]) : 0.0f;
      float q = x / s;                         /* A1: _quantizer_impl.py:161 */
      q = q - ro;
      q = (float)clamp_nan((double)nearbyintf(q), lo, hi);
      float y = q + ro;                        /* A2: :186 */
      y = y * s;
      y = (float)round_to((double)y, dt);      /* dequantize() returns the data dtype */
      float d = (float)round_to((double)(y - x), dt);   /* quantized_data - unquantized_data  (min_error.py:72) */
      float e = (float)round_to((double)(d * d), dt);   /* ** 2 */
      acc[t] += (double)e;
    }
    for (int64_t t = 0; t < ntiles; ++t) err[c * ntiles + t] = (accumulate ? err[c * ntiles + t] : 0.0f) + (float)acc[t];
  }
  free(acc);
  return FFQ_OK;
}

/* bmm on codes (_gen/fallback.py:699-798 pattern): `batch` independent products with shared per-tensor parameters — the composition
 * of ffq_linear_w8a8 calls, one per matrix pair */
size_t ffq_bmm_w8a8_workspace_bytes(int64_t batch, int64_t M, int64_t N, int64_t K) { (void)batch; return ffq_linear_w8a8_workspace_bytes(M, N, K); }
int ffq_bmm_w8a8(const int8_t* xq, const int8_t* wq, const float* x_scale, const float* x_offset, const float* w_scale,
                 const float* w_offset, void* out, int out_dt, const float* out_scale, const float* out_offset,
                 double out_num_bits, int y_dt, int64_t batch, int64_t M, int64_t N, int64_t K, void* workspace,
                 size_t workspace_bytes, void* stream) {
  if (batch < 0 || M < 0 || N < 0 || K < 0) return fail(FFQ_ERR_ARG, "negative extent");
  for (int64_t b = 0; b < batch; ++b) {
    int rc = ffq_linear_w8a8(xq + b * M * K, wq + b * N * K, NULL, x_scale, x_offset, 0, w_scale, w_offset, 0, NULL, 0,
                             (char*)out + (size_t)(b * M * N) * (size_t)dt_size(out_dt), out_dt, out_scale, out_offset, out_num_bits, y_dt, M, N, K,
                             workspace, workspace_bytes, stream);
    if (rc) return rc;
  }
  return FFQ_OK;
}

/* mlp.py:30-40 composed from the restatements above: two A6 linears (bf16 outputs), SiLU * up, A1 */
size_t ffq_mlp_gate_up_w8a8_workspace_bytes(int64_t M, int64_t N, int64_t K) { (void)K; return ffq_linear_w8a8_workspace_bytes(M, N, 0); }

int ffq_mlp_gate_up_w8a8(const int8_t* xq, const int8_t* gate_wq, const int8_t* up_wq, const int32_t* gate_rowsum,
                         const int32_t* up_rowsum, const float* x_scale, const float* x_offset,
                         const float* gate_w_scale, const float* up_w_scale, int8_t* codes_out, const float* out_scale,
                         const float* out_offset, double out_num_bits, int64_t M, int64_t N, int64_t K,
                         void* workspace, size_t workspace_bytes, void* stream) {
  if (M < 0 || N < 0 || K < 0) return fail(FFQ_ERR_ARG, "negative extent");
  if (M == 0 || N == 0) return FFQ_OK;
  if (N % 128 != 0 || K % 128 != 0 || K < 256)
    return fail(FFQ_ERR_DTYPE, "fused gate/up kernel needs N %% 128 == 0, K %% 128 == 0, K >= 256 and 16-byte aligned buffers");
  uint16_t* g = (uint16_t*)malloc((size_t)M * N * 2);
  uint16_t* u = (uint16_t*)malloc((size_t)M * N * 2);
  int rc = ffq_linear_w8a8(xq, gate_wq, gate_rowsum, x_scale, x_offset, 0, gate_w_scale, NULL, 1, NULL, 0, g, FFQ_BF16, NULL, NULL, 8.0, FFQ_BF16, M, N, K, workspace, workspace_bytes, stream);
  if (!rc) rc = ffq_linear_w8a8(xq, up_wq, up_rowsum, x_scale, x_offset, 0, up_w_scale, NULL, 1, NULL, 0, u, FFQ_BF16, NULL, NULL, 8.0, FFQ_BF16, M, N, K, workspace, workspace_bytes, stream);
  if (!rc) {
    ffq_fanout fan;
    memset(&fan, 0, sizeof fan);
    fan.count = 1; fan.num_bits = out_num_bits; fan.scale[0] = out_scale; fan.offset[0] = out_offset; fan.codes[0] = codes_out;
    rc = ffq_silu_mul_quantize(g, u, FFQ_BF16, M * N, NULL, &fan, stream);
  }
  free(g); free(u);
  return rc;
}

/* Sibling quantizers while estimating (include/ffq.h): A1's codes depend on the scale's bits and the rounded offset
 * (_quantizer_impl.py:140-141, :158-163), so a later quantizer with the same pair has the earlier one's codes. */
static int same_parameters(const float* scale, const float* offset, const float* scale2, const float* offset2) {
  const float o = offset ? nearbyintf(offset[0]) : 0.0f, o2 = offset2 ? nearbyintf(offset2[0]) : 0.0f;
  return memcmp(scale, scale2, 4) == 0 && o == o2;
}

int ffq_quantize_by_tile_unless_same(const void* data, int data_dt, const float* scale, const float* offset, int64_t numel,
                                     double num_bits, const float* earlier_scale, const float* earlier_offset, int8_t* out,
                                     void* stream) {
  if (numel < 0) return fail(FFQ_ERR_ARG, "negative extent");
  if (!ffq_can_support_bitwidth(FFQ_I8, num_bits))
    return fail(FFQ_ERR_PRECISION, "Provided dtype (%d) is not enough to store %g bits quantized values.", FFQ_I8, num_bits);
  if (numel == 0) return FFQ_OK;
  if (!data || !scale || !earlier_scale || !out) return fail(FFQ_ERR_ARG, "NULL buffer");
  if (num_bits != floor(num_bits) || num_bits < 1 || numel % 16 != 0 || !(data_dt == FFQ_F32 || data_dt == FFQ_BF16 || data_dt == FFQ_F16))
    return fail(FFQ_ERR_DTYPE, "quantize unless same: whole 16-element chunks of f32 / bf16 / f16 data (else ffq_quantize_by_tile)");
  if (same_parameters(scale, offset, earlier_scale, earlier_offset)) return FFQ_OK;
  ffq_tiling t;
  memset(&t, 0, sizeof t);
  t.ndim = 1; t.shape[0] = numel; t.tile[0] = numel;
  return ffq_quantize_by_tile(data, data_dt, scale, FFQ_F32, 1, offset, FFQ_F32, offset ? 1 : 0, &t, num_bits, out, FFQ_I8, stream);
}

int ffq_linear_w8a8_takes_earlier(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  return K % 128 == 0 && K >= 256 && M >= 128 && N >= 128 && ((M + 255) / 256) * ((N + 255) / 256) >= 64;
}

int ffq_linear_w8a8_earlier(const int8_t* xq, const int8_t* earlier_xq, const float* earlier_scale, const float* earlier_offset,
                            const int8_t* wq, const int32_t* w_rowsum, const float* x_scale, const float* x_offset,
                            const float* w_scale, const float* w_offset, int w_per_row, void* out, int out_dt, int64_t M,
                            int64_t N, int64_t K, void* workspace, size_t workspace_bytes, void* stream) {
  if (!earlier_xq || !earlier_scale) return fail(FFQ_ERR_ARG, "NULL earlier codes / scale");
  if (M == 0 || N == 0) return FFQ_OK;
  if (!x_scale) return fail(FFQ_ERR_ARG, "NULL buffer");
  if (!ffq_linear_w8a8_takes_earlier(M, N, K)) return fail(FFQ_ERR_DTYPE, "earlier codes: the persistent kernel's shapes (ffq_linear_w8a8_takes_earlier)");
  const int8_t* x = same_parameters(x_scale, x_offset, earlier_scale, earlier_offset) ? earlier_xq : xq;
  return ffq_linear_w8a8(x, wq, w_rowsum, x_scale, x_offset, 0, w_scale, w_offset, w_per_row, NULL, 0, out, out_dt, NULL, NULL, 8.0, out_dt, M, N, K,
                         workspace, workspace_bytes, stream);
}

/* mlp.py:30-40 up to the product, each linear on its own input quantizer's codes (nn/linear.py:33): two A6 linears (bf16), SiLU(gate) * up */
size_t ffq_mlp_gate_up_w8a8_estimating_workspace_bytes(int64_t M, int64_t N, int64_t K) { return ffq_linear_w8a8_workspace_bytes(M, N, K); }

int ffq_mlp_gate_up_w8a8_estimating(const int8_t* xq_gate, const int8_t* xq_up, const int8_t* gate_wq, const int8_t* up_wq,
                                    const float* x_scale_gate, const float* x_offset_gate, const float* x_scale_up,
                                    const float* x_offset_up, const float* gate_w_scale, const float* gate_w_offset,
                                    const float* up_w_scale, const float* up_w_offset, void* gate_scratch, void* product_out,
                                    int64_t M, int64_t N, int64_t K, void* workspace, size_t workspace_bytes,
                                    uint32_t* extrema_words, void* extrema_pair, void* stream) {
  if (M < 0 || N < 0 || K < 0) return fail(FFQ_ERR_ARG, "negative extent");
  if ((extrema_words == NULL) != (extrema_pair == NULL)) return fail(FFQ_ERR_ARG, "extrema_words and extrema_pair come together");
  if (M == 0 || N == 0) return FFQ_OK;
  if (N % 128 != 0 || K % 128 != 0 || K < 256) return fail(FFQ_ERR_DTYPE, "gate/up while estimating: needs N %% 128 == 0, K %% 128 == 0, K >= 256");
  int rc = ffq_linear_w8a8(xq_gate, gate_wq, NULL, x_scale_gate, x_offset_gate, 0, gate_w_scale, gate_w_offset, 1, NULL, 0, gate_scratch, FFQ_BF16, NULL,
                           NULL, 8.0, FFQ_BF16, M, N, K, workspace, workspace_bytes, stream);
  if (rc) return rc;
  /* up_proj's codes may have been left unwritten where its quantizer holds gate_proj's parameters (ffq_quantize_by_tile_unless_same) */
  if (x_scale_up && x_scale_gate && same_parameters(x_scale_up, x_offset_up, x_scale_gate, x_offset_gate)) xq_up = xq_gate;
  return ffq_linear_w8a8_gated(xq_up, up_wq, NULL, x_scale_up, x_offset_up, 0, up_w_scale, up_w_offset, 1, gate_scratch, product_out, M, N, K, workspace,
                               workspace_bytes, extrema_words, extrema_pair, stream);
}

/* mlp.py:36-38 with gate_proj's result at hand: the second linear (A6, bf16), then SiLU(gate) * up — composed from the restatements above */
int ffq_linear_w8a8_gated(const int8_t* xq, const int8_t* wq, const int32_t* w_rowsum, const float* x_scale,
                          const float* x_offset, int x_per_row, const float* w_scale, const float* w_offset,
                          int w_per_row, const void* gate, void* out, int64_t M, int64_t N, int64_t K,
                          void* workspace, size_t workspace_bytes, uint32_t* extrema_words, void* extrema_pair, void* stream) {
  if (!gate) return fail(FFQ_ERR_ARG, "NULL gate");
  if ((extrema_words == NULL) != (extrema_pair == NULL)) return fail(FFQ_ERR_ARG, "extrema_words and extrema_pair come together");
  if (M <= 0 || N <= 0) return FFQ_OK;
  uint16_t* u = (uint16_t*)malloc((size_t)M * N * 2);
  int rc = ffq_linear_w8a8(xq, wq, w_rowsum, x_scale, x_offset, x_per_row, w_scale, w_offset, w_per_row, NULL, 0, u, FFQ_BF16, NULL, NULL, 8.0,
                           FFQ_BF16, M, N, K, workspace, workspace_bytes, stream);
  if (!rc) rc = ffq_silu_mul_quantize(gate, u, FFQ_BF16, M * N, out, NULL, stream);
  free(u);
  if (!rc && extrema_pair) {  /* A4 of the product as ONE tile (minmax.py:215-232 on the finished tensor) */
    ffq_tiling t;
    memset(&t, 0, sizeof t);
    t.ndim = 1; t.shape[0] = M * N; t.tile[0] = M * N;
    rc = ffq_minmax_by_tile(out, FFQ_BF16, &t, extrema_pair, (uint16_t*)extrema_pair + 1, 0, NULL, NULL, 0, NULL, stream);
  }
  return rc;
}

/* pack_q4_0_blocks / pack_q8_0_blocks, export/stages/gguf/_packing.py:23-72. The module cannot be imported in the
   build container (it pulls in the `gguf` package, which is not installed), so this restatement is pinned by the
   reference's own assertions for it (tests/export/stages/gguf/test_packing.py), re-expressed in tests/. */
int ffq_pack_gguf_blocks(const int8_t* codes, const float* scales, int64_t nblocks, int format, uint8_t* out, void* stream) {
  (void)stream;
  if (format != 4 && format != 8) return fail(FFQ_ERR_ARG, "GGUF block format must be 4 (Q4_0) or 8 (Q8_0)");
  if (nblocks < 0) return fail(FFQ_ERR_ARG, "bad block count");
  if (nblocks == 0) return FFQ_OK;
  if (!codes || !scales || !out) return fail(FFQ_ERR_ARG, "NULL buffer");
  const int rec = format == 4 ? 18 : 34;
  for (int64_t b = 0; b < nblocks; ++b) {
    uint8_t* r = out + b * rec;
    uint16_t d = f32_to_f16(scales[b]);                    /* scales.to(torch.float16).view(uint8)        (:47,:75) */
    r[0] = (uint8_t)(d & 0xFF); r[1] = (uint8_t)(d >> 8);
    const int8_t* q = codes + b * 32;
    if (format == 4) {
      for (int j = 0; j < 16; ++j) {
        int lo = q[j] + 8, hi = q[j + 16] + 8;             /* (int_codes + 8).clamp(0, 15)                     (:49) */
        lo = lo < 0 ? 0 : (lo > 15 ? 15 : lo);
        hi = hi < 0 ? 0 : (hi > 15 ? 15 : hi);
        r[2 + j] = (uint8_t)(lo | (hi << 4));              /* first half low nibble, second half high      (:51-52) */
      }
    } else {
      for (int j = 0; j < 32; ++j) r[2 + j] = (uint8_t)(int8_t)(q[j] < -127 ? -127 : q[j]);  /* clamp(-127, 127)  (:77) */
    }
  }
  return FFQ_OK;
}

/* gptq(), quantization/gptq.py:101-131 for one block of columns (per-row parameters: column_quantizer :149-235) */
int ffq_gptq_block(float* weights, float* quantized, float* errors, int64_t rows, int64_t row_stride,
                   int64_t col0, int64_t block_cols, const float* hinv, int64_t hinv_stride, const float* scale,
                   int64_t scale_numel, const float* offset, int64_t offset_numel, double num_bits, void* stream) {
  (void)stream;
  if (rows < 0 || block_cols < 0 || col0 < 0) return fail(FFQ_ERR_ARG, "negative extent");
  if (block_cols > 128) return fail(FFQ_ERR_DTYPE, "GPTQ block kernel handles at most %d columns per block", 128);
  if (rows == 0 || block_cols == 0) return FFQ_OK;
  if (!weights || !quantized || !errors || !hinv || !scale) return fail(FFQ_ERR_ARG, "NULL buffer");
  if ((scale_numel != 1 && scale_numel != rows) || (offset && offset_numel != 1 && offset_numel != rows))
    return fail(FFQ_ERR_PARAM_NUMEL, "GPTQ block kernel takes one scale / offset per row (or one in total)");
  float lo = (float)(-pow(2.0, num_bits - 1.0)), hi = -lo - 1.0f;
  float* w = (float*)malloc(sizeof(float) * (size_t)block_cols);
  for (int64_t r = 0; r < rows; ++r) {
    float s = scale[scale_numel == 1 ? 0 : r];
    float o = offset ? nearbyintf(offset[offset_numel == 1 ? 0 : r]) : 0.0f;
    for (int64_t k = 0; k < block_cols; ++k) w[k] = weights[r * row_stride + col0 + k];   /* weights_block = ...clone() (:104) */
    for (int64_t j = 0; j < block_cols; ++j) {
      float q = w[j] / s;                                   /* quant_deq(weights_block[:, j])           (:124-125) */
      q = q - o;
      q = (float)clamp_nan((double)nearbyintf(q), lo, hi);
      float dq = q + o;
      dq = dq * s;
      float d = w[j] - dq;
      float e = d / hinv[(col0 + j) * hinv_stride + col0 + j];   /* errors[:, i + j]                    (:127-129) */
      quantized[r * row_stride + col0 + j] = dq;
      errors[r * row_stride + col0 + j] = e;
      for (int64_t k = j + 1; k < block_cols; ++k) {       /* weights_block[:, j + 1:] -= e @ hinv[j, j + 1:]  (:131) */
        float p = e * hinv[(col0 + j) * hinv_stride + col0 + k];
        w[k] = w[k] - p;
      }
    }
  }
  free(w);
  return FFQ_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* Attention of the quantized Llama helper, docs/examples/doc_helpers/quantized_llama/         */
/* attention.py:45-92, op by op with every tensor in bf16 as the eager chain keeps it:         */
/*   repeat_kv (:57-58); weights = bf16(q @ k^T) (:60-64, fp32 accumulation in the matmul);    */
/*   weights = bf16(weights * scaling) (:65); weights = bf16(weights + mask) (:67-68, mask =   */
/*   0 / finfo(bf16).min above the diagonal); softmax in fp32, cast to bf16 (:70-74);          */
/*   out = bf16(weights @ v) (:83-85); transpose / reshape to [batch, seq, heads * head_dim]   */
/*   (:87-88). Then the input quantizer of o_proj (nn/linear.py:33 -> A1).                     */
/* ------------------------------------------------------------------------------------------ */
int ffq_attention(const void* q, const void* k, const void* v, int dt, int64_t batch, int64_t seq_len,
                  int64_t q_heads, int64_t kv_heads, int64_t head_dim, double softmax_scale, int causal,
                  void* ctx_out, int8_t* codes_out, const float* out_scale, const float* out_offset,
                  double out_num_bits, const void* q_cos, const void* q_sin, void* stream) {
  if (batch < 0 || seq_len < 0 || q_heads <= 0 || kv_heads <= 0) return fail(FFQ_ERR_ARG, "bad extent");
  if ((q_cos == NULL) != (q_sin == NULL)) return fail(FFQ_ERR_ARG, "q_cos and q_sin come together");
  if (q_cos && q && batch > 0 && seq_len > 0 && dt == FFQ_BF16 && head_dim == 128) {
    /* attention.py:20-41 then :45-92: rotate a copy of q with the restatement of the rotary embedding, then the attention on it */
    const size_t bytes = (size_t)batch * seq_len * q_heads * head_dim * 2;
    void* rotated = malloc(bytes);
    memcpy(rotated, q, bytes);
    int rc = ffq_rope_inplace(rotated, q_heads, NULL, 0, dt, batch * seq_len, seq_len, head_dim, q_cos, q_sin, stream);
    if (!rc) rc = ffq_attention(rotated, k, v, dt, batch, seq_len, q_heads, kv_heads, head_dim, softmax_scale, causal, ctx_out, codes_out, out_scale,
                                out_offset, out_num_bits, NULL, NULL, stream);
    free(rotated);
    return rc;
  }
  if (dt != FFQ_BF16) return fail(FFQ_ERR_DTYPE, "attention is built for bf16 activations");
  if (head_dim != 128) return fail(FFQ_ERR_DTYPE, "attention is built for head_dim 128");
  if (seq_len % 64 != 0) return fail(FFQ_ERR_DTYPE, "attention needs seq_len %% 64 == 0");
  if (q_heads % kv_heads != 0) return fail(FFQ_ERR_ARG, "q_heads must be a multiple of kv_heads");
  if (batch == 0 || seq_len == 0) return FFQ_OK;
  if (!q || !k || !v || (!ctx_out && !codes_out)) return fail(FFQ_ERR_ARG, "NULL buffer");
  if (codes_out && !out_scale) return fail(FFQ_ERR_ARG, "codes need a scale");
  if (codes_out && !(out_num_bits >= 1.0 && out_num_bits <= 8.0 && out_num_bits == floor(out_num_bits)))
    return fail(FFQ_ERR_ARG, "codes need an integral bit-width in 1..8");
  const uint16_t* qp = (const uint16_t*)q;
  const uint16_t* kp = (const uint16_t*)k;
  const uint16_t* vp = (const uint16_t*)v;
  const int64_t groups = q_heads / kv_heads;
  const float scaling = (float)softmax_scale;
  const float mask_min = bits_f32(0xFF7F0000u); /* torch.finfo(torch.bfloat16).min */
  const double lo = -pow(2.0, out_num_bits - 1.0), hi = -lo - 1.0;
  float* w = (float*)malloc((size_t)seq_len * sizeof(float));
  float* acc = (float*)malloc((size_t)head_dim * sizeof(float));
  for (int64_t b = 0; b < batch; ++b)
    for (int64_t hd = 0; hd < q_heads; ++hd) {
      const int64_t kvh = hd / groups;
      for (int64_t i = 0; i < seq_len; ++i) {
        const uint16_t* qrow = qp + ((b * seq_len + i) * q_heads + hd) * head_dim;
        float mx = -INFINITY;
        for (int64_t j = 0; j < seq_len; ++j) {
          const uint16_t* krow = kp + ((b * seq_len + j) * kv_heads + kvh) * head_dim;
          float dot = 0.0f;
          for (int64_t d = 0; d < head_dim; ++d) dot += bf16_to_f32(qrow[d]) * bf16_to_f32(krow[d]);
          float s = bf16_round(bf16_round(dot) * scaling);
          if (causal) s = bf16_round(s + (j > i ? mask_min : 0.0f));
          w[j] = s;
          if (s > mx) mx = s;
        }
        float sum = 0.0f;
        for (int64_t j = 0; j < seq_len; ++j) { w[j] = expf(w[j] - mx); sum += w[j]; }
        for (int64_t d = 0; d < head_dim; ++d) acc[d] = 0.0f;
        for (int64_t j = 0; j < seq_len; ++j) {
          const float p = bf16_round(w[j] / sum);
          if (p == 0.0f) continue;
          const uint16_t* vrow = vp + ((b * seq_len + j) * kv_heads + kvh) * head_dim;
          for (int64_t d = 0; d < head_dim; ++d) acc[d] += p * bf16_to_f32(vrow[d]);
        }
        const int64_t at = ((b * seq_len + i) * q_heads + hd) * head_dim;
        for (int64_t d = 0; d < head_dim; ++d) {
          const float z = bf16_round(acc[d]);
          if (ctx_out) ((uint16_t*)ctx_out)[at + d] = f32_to_bf16(z);
          if (codes_out) {
            const float s = out_scale[0];
            const float o = out_offset ? nearbyintf(out_offset[0]) : 0.0f;
            float c = z / s;
            c = c - o;
            c = nearbyintf(c);
            st(codes_out, FFQ_I8, at + d, clamp_nan((double)c, lo, hi));
          }
        }
      }
    }
  free(w);
  free(acc);
  return FFQ_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* ABI version 3: weight codes + their row sums in one call, and the GEMM entry points that    */
/* take such sums. The restatement is the composition the fused forms replace: A1              */
/* (_quantizer_impl.py:154-169) per row, then sum_k codes; as the checker it REFUSES row sums   */
/* that are not the sums of the codes it is given.                                              */
/* ------------------------------------------------------------------------------------------ */
int ffq_quantize_rows_rowsum(const void* data, int data_dt, const float* scale, const float* offset, int64_t rows,
                             int64_t cols, double num_bits, int8_t* codes, int32_t* rowsum, void* stream) {
  if (rows < 0 || cols < 0) return fail(FFQ_ERR_ARG, "negative extent");
  if (data_dt != FFQ_BF16) return fail(FFQ_ERR_DTYPE, "fused weight quantize + row sums is built for bf16 weights");
  if (cols % 1024 != 0) return fail(FFQ_ERR_DTYPE, "fused weight quantize + row sums needs cols %% 1024 == 0");
  if (rows == 0 || cols == 0) return FFQ_OK;
  if (!data || !scale || !codes || !rowsum) return fail(FFQ_ERR_ARG, "NULL buffer");
  ffq_tiling t;
  memset(&t, 0, sizeof t);
  t.ndim = 2;
  t.shape[0] = rows; t.shape[1] = cols;
  t.tile[0] = 1; t.tile[1] = cols;
  int rc = ffq_quantize_by_tile(data, data_dt, scale, FFQ_F32, rows, offset, FFQ_F32, offset ? rows : 0, &t, num_bits, codes,
                                FFQ_I8, stream);
  if (rc) return rc;
  for (int64_t r = 0; r < rows; ++r) {
    int32_t sum = 0;
    for (int64_t c = 0; c < cols; ++c) sum += codes[r * cols + c];
    rowsum[r] += sum; /* the caller zeroes the sums */
  }
  return FFQ_OK;
}

/* test hook of the HIP library (kernel-family selection): the oracle has one implementation, nothing to select */
/* ABI 9: q / k / v on the same activation codes in one call = the single-matrix entry once per matrix (fallback.py:77-112 three times) */
int ffq_linear_w8a8_multi(const int8_t* xq, const int8_t* wq, const int32_t* w_rowsum, const float* x_scale, const float* x_offset,
                          int x_per_row, const float* w_scale, int count, void* const* outs, int out_dt, int64_t M, const int64_t* Ns,
                          int64_t K, void* workspace, size_t workspace_bytes, void* stream) {
  if (count < 2 || count > 3 || !outs || !Ns) return fail(FFQ_ERR_ARG, "2 or 3 weight matrices");
  int64_t at = 0;
  for (int i = 0; i < count; ++i) {
    if (Ns[i] <= 0 || !outs[i]) return fail(FFQ_ERR_ARG, "empty weight matrix or NULL output");
    if (i + 1 < count && Ns[i] % 256 != 0) return fail(FFQ_ERR_DTYPE, "every weight matrix but the last needs a multiple of 256 rows");
    int rc = ffq_linear_w8a8(xq, wq + at * K, w_rowsum ? w_rowsum + at : NULL, x_scale, x_offset, x_per_row, w_scale + at, NULL, 1, NULL, 0, outs[i],
                             out_dt, NULL, NULL, 8.0, 0, M, Ns[i], K, workspace, workspace_bytes, stream);
    if (rc) return rc;
    at += Ns[i];
  }
  return FFQ_OK;
}

int ffq_force_generic_kernels(int on) { (void)on; return 0; }

/* A1 of several row-quantized weights: the composition the one-launch form replaces (_quantizer_impl.py:154-169 per member) */
int ffq_quantize_rows_batch(const ffq_rows_batch* batch, int data_dt, void* stream) {
  if (!batch || batch->count < 0 || batch->count > FFQ_MAX_BATCH) return fail(FFQ_ERR_ARG, "batch count must be 0..%d", FFQ_MAX_BATCH);
  for (int i = 0; i < batch->count; ++i) {
    ffq_tiling t;
    memset(&t, 0, sizeof t);
    t.ndim = 2;
    t.shape[0] = batch->rows[i]; t.shape[1] = batch->cols[i];
    t.tile[0] = 1; t.tile[1] = batch->cols[i];
    int rc = ffq_quantize_by_tile(batch->data[i], data_dt, batch->scale[i], FFQ_F32, batch->rows[i], batch->offset[i], FFQ_F32,
                                  batch->offset[i] ? batch->rows[i] : 0, &t, batch->num_bits, batch->codes[i], FFQ_I8, stream);
    if (rc) return rc;
    if (batch->rowsum[i]) {  /* += the sum of each row's codes (the int8 GEMM's zero-point term, fallback.py:94-100 expanded) */
      for (int64_t r = 0; r < batch->rows[i]; ++r) {
        int32_t sum = 0;
        for (int64_t c = 0; c < batch->cols[i]; ++c) sum += batch->codes[i][r * batch->cols[i] + c];
        batch->rowsum[i][r] += sum;
      }
    }
  }
  return FFQ_OK;
}

/*
 * mlp.py:30-40 on a weight-only quantized model: down_proj's argument act_fn(gate_proj(x)) * up_proj(x), the two
 * projections being fallback.linear with a quantized weight (_gen/fallback.py:86-112) in bf16 — restated as exactly that
 * composition of the functions above.
 */
/* several weight matrices on the same activations (q / k / v projections; reference nn/linear.py:32-39 once per module): the
 * composition of ffq_linear_wq calls above — the restatement has no tile walk to share */
int ffq_linear_wq_multi(const void* x, int x_dt, int count, const void* const* w_codes, int w_dt, int64_t pack_block,
                        const float* const* w_scale, const float* const* w_offset, int per_row, int64_t group, void* const* outs,
                        int out_dt, int64_t M, const int64_t* Ns, int64_t K, void* workspace, size_t workspace_bytes, int32_t* tickets,
                        int64_t split, void* stream) {
  (void)workspace; (void)workspace_bytes; (void)tickets; (void)split;
  if (count < 1 || count > 3 || !w_codes || !w_scale || !w_offset || !outs || !Ns) return fail(FFQ_ERR_ARG, "1 to 3 weight matrices");
  for (int i = 0; i < count; ++i) {
    if (i + 1 < count && Ns[i] % 256 != 0) return fail(FFQ_ERR_DTYPE, "every weight matrix but the last needs a multiple of 256 rows");
    if ((w_offset[i] == NULL) != (w_offset[0] == NULL)) return fail(FFQ_ERR_ARG, "offsets for all weight matrices or for none");
    int64_t numel = per_row ? Ns[i] * (K / group) : 1;
    int rc = ffq_linear_wq(x, x_dt, w_codes[i], w_dt, pack_block, w_scale[i], w_offset[i], numel, group, NULL, 0, outs[i], out_dt, M, Ns[i], K, NULL, 0, NULL, 0, stream);
    if (rc != FFQ_OK) return rc;
  }
  return FFQ_OK;
}

size_t ffq_mlp_gate_up_wq_workspace_bytes(int64_t M, int64_t N, int64_t K) { (void)M; (void)N; (void)K; return 0; }

int ffq_mlp_gate_up_wq(const void* x, int x_dt, const void* gate_codes, const void* up_codes, int w_dt, int64_t pack_block,
                       const float* gate_scale, const float* gate_offset, const float* up_scale, const float* up_offset,
                       int64_t scale_numel, int64_t group, void* out, int64_t M, int64_t N, int64_t K, void* workspace,
                       size_t workspace_bytes, int32_t* tickets, int64_t split, void* stream) {
  (void)workspace; (void)workspace_bytes; (void)tickets; (void)split;
  if (M < 0 || N < 0 || K < 0) return fail(FFQ_ERR_ARG, "negative extent");
  if (M == 0 || N == 0) return FFQ_OK;
  if (!x || !gate_codes || !up_codes || !gate_scale || !up_scale || !out) return fail(FFQ_ERR_ARG, "NULL buffer");
  if ((gate_offset == NULL) != (up_offset == NULL)) return fail(FFQ_ERR_ARG, "gate and up need offsets both or neither");
  if (N % 128 != 0) return fail(FFQ_ERR_DTYPE, "fused weight-only gate/up: needs N %% 128 == 0");
  size_t bytes = (size_t)M * (size_t)N * 2u;
  void* g = malloc(bytes);
  void* u = malloc(bytes);
  if (!g || !u) { free(g); free(u); return fail(FFQ_ERR_ARG, "out of memory"); }
  int rc = ffq_linear_wq(x, x_dt, gate_codes, w_dt, pack_block, gate_scale, gate_offset, scale_numel, group, NULL, 0, g, FFQ_BF16, M, N, K, NULL, 0, NULL, 0, stream);
  if (rc == FFQ_OK) rc = ffq_linear_wq(x, x_dt, up_codes, w_dt, pack_block, up_scale, up_offset, scale_numel, group, NULL, 0, u, FFQ_BF16, M, N, K, NULL, 0, NULL, 0, stream);
  if (rc == FFQ_OK) rc = ffq_silu_mul_quantize(g, u, FFQ_BF16, M * N, out, NULL, stream);
  free(g); free(u);
  return rc;
}
