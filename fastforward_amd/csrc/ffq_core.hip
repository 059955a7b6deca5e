// ffq_core.hip — error plumbing, dtype rules and tiling analysis of libffq_hip.so.
#include "ffq_common.h"

#include <math.h>
#include <string.h>

namespace ffq {

static thread_local char g_err[512];

char* err_buf() { return g_err; }

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(FFQ_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return FFQ_OK;
}

// check_tile_compatibility — reference quantization/tiled_tensor.py:19-42
int check_tiling(const ffq_tiling* t) {
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

// Classify the tile grid so the launcher can pick index math that never materialises
// tiles_to_rows (reference quantization/tiled_tensor.py:71-98).
int analyse(const ffq_tiling* t, TileInfo* info) {
  int rc = check_tiling(t);
  if (rc) return rc;
  int64_t numel = 1, ntiles = 1;
  for (int i = 0; i < t->ndim; ++i) numel *= t->shape[i];
  info->numel = numel;
  info->run = info->inner = info->channels = 0;
  if (numel == 0) {
    info->ntiles = 1;
    info->layout = LAYOUT_SCALAR;
    return FFQ_OK;
  }
  for (int i = 0; i < t->ndim; ++i) ntiles *= t->shape[i] / t->tile[i];
  info->ntiles = ntiles;
  if (ntiles == 1) {
    info->layout = LAYOUT_SCALAR;
    return FFQ_OK;
  }
  // Drop extent-1 dimensions: they do not influence the flat -> tile map.
  int64_t shape[FFQ_MAX_DIMS], tile[FFQ_MAX_DIMS];
  int nd = 0;
  for (int i = 0; i < t->ndim; ++i) {
    if (t->shape[i] == 1) continue;
    shape[nd] = t->shape[i];
    tile[nd] = t->tile[i];
    ++nd;
  }
  // ROWS: (1, ..., 1, b, full, ..., full) — every tile is one contiguous run.
  {
    int j = 0;
    while (j < nd && tile[j] == 1) ++j;
    bool ok = true;
    int64_t run = 1;
    if (j < nd) {
      run = tile[j];
      for (int k = j + 1; k < nd; ++k) {
        if (tile[k] != shape[k]) ok = false;
        run *= shape[k];
      }
    }
    if (ok) {
      info->layout = LAYOUT_ROWS;
      info->run = run;
      return FFQ_OK;
    }
  }
  // CHANNEL: exactly one dimension has tile 1, every other dimension is whole.
  {
    int c = -1;
    bool ok = true;
    for (int k = 0; k < nd; ++k) {
      if (tile[k] == shape[k]) continue;
      if (tile[k] == 1 && c < 0) c = k; else ok = false;
    }
    if (ok && c >= 0) {
      int64_t inner = 1;
      for (int k = c + 1; k < nd; ++k) inner *= shape[k];
      info->layout = LAYOUT_CHANNEL;
      info->inner = inner;
      info->channels = shape[c];
      return FFQ_OK;
    }
  }
  info->layout = LAYOUT_GENERIC;
  return FFQ_OK;
}

// Broadcast rule of `scale[:, None]` against rows [ntiles, tile_numel]
// (reference quantization/_quantizer_impl.py:161,182).
int check_param_numel(const char* what, int64_t numel, int64_t ntiles) {
  if (numel == ntiles || numel == 1) return FFQ_OK;
  if (ntiles == 1)
    return fail(FFQ_ERR_PARAM_ROWS, "tiled_data is expected to be of size (1, L) but %s has %lld entries",
                what, (long long)numel);
  return fail(FFQ_ERR_PARAM_NUMEL,
              "The size of tensor a (%lld) must match the size of tensor b (%lld) at non-singleton "
              "dimension 0 (%s vs number of tiles)",
              (long long)ntiles, (long long)numel, what);
}

GenericTiling make_generic(const ffq_tiling* t) {
  GenericTiling g;
  memset(&g, 0, sizeof g);
  g.ndim = t->ndim;
  int64_t s = 1;
  for (int k = t->ndim - 1; k >= 0; --k) {
    g.shape[k] = t->shape[k];
    g.tile[k] = t->tile[k];
    g.gstride[k] = s;
    s *= t->shape[k] / t->tile[k];
  }
  return g;
}

}  // namespace ffq

using namespace ffq;

extern "C" {

int ffq_abi_version(void) { return FFQ_ABI_VERSION; }
const char* ffq_last_error(void) { return err_buf(); }
const char* ffq_backend_name(void) { return "hip:gfx950"; }

int64_t ffq_num_tiles(const ffq_tiling* tiling) {
  TileInfo info;
  int rc = analyse(tiling, &info);
  if (rc) return -rc;
  return info.ntiles;
}

// c10::promoteTypes restricted to the dtypes of this ABI (both operands are dimensioned tensors).
int ffq_promote_types(int a, int b) {
  if (!dt_valid(a) || !dt_valid(b)) return -FFQ_ERR_ARG;
  if (a == b) return a;
  const bool fa = dt_is_float(a), fb = dt_is_float(b);
  if (fa && !fb) return a;
  if (fb && !fa) return b;
  if (fa && fb) {
    if (a == FFQ_F64 || b == FFQ_F64) return FFQ_F64;
    return FFQ_F32;  // f32 x half, or bf16 x f16
  }
  if (a == FFQ_I64 || b == FFQ_I64) return FFQ_I64;
  if (a == FFQ_I32 || b == FFQ_I32) return FFQ_I32;
  return FFQ_I16;  // i16 x {i8,u8}, or i8 x u8
}

// can_support_bitwidth — reference quantization/_quantizer_impl.py:44-75
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

int ffq_dequantize_result_dtype(int data_dt, int scale_dt, int offset_dt, int has_offset) {
  const int off_dt = has_offset ? offset_dt : scale_dt;
  const int add_dt = ffq_promote_types(data_dt, off_dt);
  if (add_dt < 0) return add_dt;
  return ffq_promote_types(add_dt, scale_dt);
}

}  // extern "C"

// ---- test hook: kernel-family selection (no environment variables in the product) ------------------------------------------
namespace ffq {
static int g_force_generic = 0;
bool generic_kernels_forced() { return (__atomic_load_n(&g_force_generic, __ATOMIC_RELAXED) & 1) != 0; }
bool splitk_abandon_forced() { return (__atomic_load_n(&g_force_generic, __ATOMIC_RELAXED) & 2) != 0; }
bool mid_register_form_forced() { return (__atomic_load_n(&g_force_generic, __ATOMIC_RELAXED) & 4) != 0; }
}  // namespace ffq

// bit 0: the generic kernel families; bit 1: every odd K slice of a split-K tile gives up waiting for its peers at once (the
// path a unit takes when its peers cannot become resident: include/ffq.h, ffq_linear_wq); bit 2: the 128-column tiles of the
// weight-only linear take their register-staged kernel where the LDS-DMA kernel would run (bit-equal results: ffq_wmid.hip)
extern "C" int ffq_force_generic_kernels(int on) {
  return __atomic_exchange_n(&ffq::g_force_generic, on & 7, __ATOMIC_RELAXED);
}
