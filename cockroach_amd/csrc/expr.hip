// expr.hip -- the selection and projection operators that sit between the scan and the hash aggregator in the
// reference's Q1 plan (`scan -> selLEInt64Int64ConstOp -> 3 float projections -> hashAggregator`,
// tpch_vec:20537-20546), as whole-column device kernels.
//
// Replaces, for the INT64-vs-constant comparisons and the FLOAT64 +, -, * overloads:
//   colexecsel.sel{EQ,NE,LT,LE,GT,GE}Int64Int64ConstOp.Next   (colexecsel/selection_ops.eg.go:5876,25260,34952 ...)
//   colexecproj.proj{Plus,Minus,Mult}Float64Float64Op.Next     (colexecproj/proj_non_const_ops.eg.go:4922,8126,12196)
//   colexecprojconst.proj{Plus,Minus,Mult}Float64ConstFloat64Op (colexecprojconst/proj_const_left_ops.eg.go:4515,7503,11295)
//   and their constant-on-the-right twins (proj_const_right_ops.eg.go).
//
// Not the reference's shape: a selection there rewrites the batch's selection vector; here it produces a BOOL column
// (1 = row passes; a NULL operand never passes, selection_ops.eg.go:34971) that the aggregator takes as the FILTER of
// every aggregate, so no row is ever moved (a gather of the ~98.6 % passing rows of Q1 would cost more than the
// aggregation).  A projection writes a dense FLOAT64 column; its validity is the AND of its operands' bitmaps
// (`projVec.SetNulls(_outNulls.Or(col1Nulls).Or(col2Nulls))`, proj_non_const_ops.eg.go:12253), values under a NULL
// are unspecified as in the reference.  Outputs are caller-provided device buffers.
//
// Roofline: HBM streaming, 8-24 B read and 1-8 B written per row.
#include <cstring>

#include "device.h"

namespace crgpu {

enum { CMP_EQ = 0, CMP_NE, CMP_LT, CMP_LE, CMP_GT, CMP_GE };
enum { BIN_PLUS = 0, BIN_MINUS, BIN_MULT, BIN_MULT_ONE_MINUS, BIN_MULT_ONE_PLUS };  // the last two: a * (1 -/+ b)

template <int OP>
__global__ __launch_bounds__(256) void select_i64_const_kernel(const long long *__restrict__ col, const uint8_t *__restrict__ valid,
                                                               long long cst, int64_t n, uint8_t *__restrict__ out) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const long long v = col[i];
    bool r;
    switch (OP) {
      case CMP_EQ: r = v == cst; break;
      case CMP_NE: r = v != cst; break;
      case CMP_LT: r = v < cst; break;
      case CMP_LE: r = v <= cst; break;
      case CMP_GT: r = v > cst; break;
      default: r = v >= cst; break;
    }
    if (valid != nullptr && !((valid[i >> 3] >> (i & 7)) & 1)) r = false;
    out[i] = r ? 1 : 0;
  }
}

// a * (1 - b) and a * (1 + b) are two reference operators in a row (projMinusFloat64ConstFloat64Op then
// projMultFloat64Float64Op) done in one pass: the same two roundings (no fused multiply-add applies), half the traffic.
template <int OP>
__device__ __forceinline__ double bin_f64(double x, double y) {
  switch (OP) {
    case BIN_PLUS: return x + y;
    case BIN_MINUS: return x - y;
    case BIN_MULT: return x * y;
    case BIN_MULT_ONE_MINUS: return x * (1.0 - y);
    default: return x * (1.0 + y);
  }
}

template <int OP>
__global__ __launch_bounds__(256) void project_f64_kernel(const double *__restrict__ a, double ac, const double *__restrict__ b,
                                                          double bc, int64_t n, double *__restrict__ out) {
  // two rows per thread: 16-byte loads and stores (n2 pairs, then the odd last row)
  const int64_t n2 = n >> 1;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n2; i += stride) {
    const double2 x = a ? reinterpret_cast<const double2 *>(a)[i] : make_double2(ac, ac);
    const double2 y = b ? reinterpret_cast<const double2 *>(b)[i] : make_double2(bc, bc);
    reinterpret_cast<double2 *>(out)[i] = make_double2(bin_f64<OP>(x.x, y.x), bin_f64<OP>(x.y, y.y));
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) out[n - 1] = bin_f64<OP>(a ? a[n - 1] : ac, b ? b[n - 1] : bc);
}

__global__ __launch_bounds__(256) void and_bitmaps_kernel(const uint8_t *__restrict__ a, const uint8_t *__restrict__ b,
                                                          int64_t nbytes, uint8_t *__restrict__ out) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < nbytes; i += stride)
    out[i] = static_cast<uint8_t>((a ? a[i] : 0xFF) & (b ? b[i] : 0xFF));
}

// BYTES column against a constant: EQ / NE (selEQBytesBytesConstOp, selNEBytesBytesConstOp) and LIKE '%c%'
// (selContainsBytesBytesConstOp, case-sensitive form).  The constant travels in the kernel arguments (<= 64 bytes).
static constexpr int kMaxBytesConst = 64;
struct BytesConst {
  uint8_t b[kMaxBytesConst];
  uint32_t len;
};
enum { BCMP_EQ = 0, BCMP_NE = 1, BCMP_CONTAINS = 6, BCMP_NOT_CONTAINS = 7 };
__global__ __launch_bounds__(256) void select_bytes_const_kernel(int op, const uint8_t *__restrict__ data,
                                                                 const uint32_t *__restrict__ offsets,
                                                                 const uint8_t *__restrict__ valid, BytesConst c, int64_t n,
                                                                 uint8_t *__restrict__ out) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint32_t o = offsets[i], len = offsets[i + 1] - o;
    const uint8_t *p = data + o;
    bool r;
    if (op == BCMP_EQ || op == BCMP_NE) {
      bool eq = len == c.len;
      for (uint32_t t = 0; eq && t < c.len; t++) eq = p[t] == c.b[t];
      r = (op == BCMP_EQ) == eq;
    } else {
      bool found = c.len == 0;  // bytes.Contains(x, "") is true
      for (uint32_t s0 = 0; !found && s0 + c.len <= len; s0++) {
        bool m = true;
        for (uint32_t t = 0; m && t < c.len; t++) m = p[s0 + t] == c.b[t];
        found = m;
      }
      r = (op == BCMP_CONTAINS) == found;
    }
    if (valid != nullptr && !((valid[i >> 3] >> (i & 7)) & 1)) r = false;  // a NULL never passes
    out[i] = r ? 1 : 0;
  }
}

// EXTRACT(year FROM date): the year of a DATE (int64 days since 1970-01-01, coldata canonical family Int) as FLOAT64.
// Proleptic Gregorian calendar, the same as Go's time.Time.Year() that the reference's builtin returns
// (sem/builtins/builtins.go:11946 via colexecbuiltins.defaultBuiltinFuncOperator, builtin_funcs.go:44).
__device__ __forceinline__ long long year_of_days(long long z) {
  z += 719468;  // days from 0000-03-01
  const long long era = (z >= 0 ? z : z - 146096) / 146097;
  const unsigned long long doe = static_cast<unsigned long long>(z - era * 146097);             // [0, 146096]
  const unsigned long long yoe = (doe - doe / 1460 + doe / 36524 - doe / 146096) / 365;         // [0, 399]
  const long long y = static_cast<long long>(yoe) + era * 400;
  const unsigned long long doy = doe - (365 * yoe + yoe / 4 - yoe / 100);                       // [0, 365]
  const unsigned long long mp = (5 * doy + 2) / 153;                                            // [0, 11]
  return y + (mp >= 10 ? 1 : 0);  // January and February belong to the next civil year
}
__global__ __launch_bounds__(256) void extract_year_kernel(const long long *__restrict__ days, int64_t n, double *__restrict__ out) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = static_cast<double>(year_of_days(days[i]));
}

static inline int egrid(int64_t n) {
  int64_t want = (n + 255) / 256;
  const int64_t cap = static_cast<int64_t>(num_cus()) * 16;
  if (want < 1) want = 1;
  return static_cast<int>(want < cap ? want : cap);
}

}  // namespace crgpu

using namespace crgpu;

extern "C" crgpu_status crgpu_dev_select_i64_const(int32_t cmp, const crgpu_dcol *col, int64_t cst, int64_t n, uint8_t *out_bool) {
  CRGPU_ENTER(nullptr);
  if (col == nullptr || (n > 0 && (col->data == nullptr || out_bool == nullptr)))
    return fail(CRGPU_ERR_INVALID, "crgpu_dev_select_i64_const: null argument");
  if (col->type != CRGPU_INT64) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_select_i64_const: INT64 columns only");
  if (cmp < CMP_EQ || cmp > CMP_GE) return fail(CRGPU_ERR_INVALID, "crgpu_dev_select_i64_const: unknown comparison");
  if (n <= 0) return CRGPU_OK;
  hipStream_t st = stream();
  const long long *d = static_cast<const long long *>(col->data);
  ProfScope prof("expr_select");
#define CRGPU_SEL(OP) \
  case OP: hipLaunchKernelGGL(select_i64_const_kernel<OP>, dim3(egrid(n)), dim3(256), 0, st, d, col->valid, static_cast<long long>(cst), n, out_bool); break;
  switch (cmp) {
    CRGPU_SEL(CMP_EQ) CRGPU_SEL(CMP_NE) CRGPU_SEL(CMP_LT) CRGPU_SEL(CMP_LE) CRGPU_SEL(CMP_GT) CRGPU_SEL(CMP_GE)
  }
#undef CRGPU_SEL
  CRGPU_HIP(hipGetLastError());
  mark_default_work();
  return CRGPU_OK;
}

extern "C" crgpu_status crgpu_dev_project_f64(int32_t op, const crgpu_dcol *a, double a_const, const crgpu_dcol *b, double b_const,
                                              int64_t n, double *out_data, uint8_t *out_valid) {
  CRGPU_ENTER(nullptr);
  if (a == nullptr && b == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_project_f64: two constants");
  if (op < BIN_PLUS || op > BIN_MULT_ONE_PLUS) return fail(CRGPU_ERR_INVALID, "crgpu_dev_project_f64: unknown operator");
  for (const crgpu_dcol *c : {a, b}) {
    if (c == nullptr) continue;
    if (c->type != CRGPU_FLOAT64) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_project_f64: FLOAT64 operands only");
    if (n > 0 && c->data == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_project_f64: null data pointer");
    if ((reinterpret_cast<uintptr_t>(c->data) & 15u) != 0) return fail(CRGPU_ERR_INVALID, "crgpu_dev_project_f64: operands must be 16-byte aligned");
  }
  const uint8_t *av = a ? a->valid : nullptr, *bv = b ? b->valid : nullptr;
  if ((av != nullptr || bv != nullptr) && out_valid == nullptr)
    return fail(CRGPU_ERR_INVALID, "crgpu_dev_project_f64: an operand has NULLs but no output bitmap was given");
  if (n <= 0) return CRGPU_OK;
  if (out_data == nullptr || (reinterpret_cast<uintptr_t>(out_data) & 15u) != 0)
    return fail(CRGPU_ERR_INVALID, "crgpu_dev_project_f64: output must be a 16-byte aligned device buffer");
  hipStream_t st = stream();
  const double *ad = a ? static_cast<const double *>(a->data) : nullptr, *bd = b ? static_cast<const double *>(b->data) : nullptr;
  ProfScope prof("expr_project");
  const int grid = egrid((n + 1) / 2);
  switch (op) {
    case BIN_PLUS: hipLaunchKernelGGL(project_f64_kernel<BIN_PLUS>, dim3(grid), dim3(256), 0, st, ad, a_const, bd, b_const, n, out_data); break;
    case BIN_MINUS: hipLaunchKernelGGL(project_f64_kernel<BIN_MINUS>, dim3(grid), dim3(256), 0, st, ad, a_const, bd, b_const, n, out_data); break;
    case BIN_MULT: hipLaunchKernelGGL(project_f64_kernel<BIN_MULT>, dim3(grid), dim3(256), 0, st, ad, a_const, bd, b_const, n, out_data); break;
    case BIN_MULT_ONE_MINUS: hipLaunchKernelGGL(project_f64_kernel<BIN_MULT_ONE_MINUS>, dim3(grid), dim3(256), 0, st, ad, a_const, bd, b_const, n, out_data); break;
    default: hipLaunchKernelGGL(project_f64_kernel<BIN_MULT_ONE_PLUS>, dim3(grid), dim3(256), 0, st, ad, a_const, bd, b_const, n, out_data); break;
  }
  CRGPU_HIP(hipGetLastError());
  if (out_valid != nullptr) {
    hipLaunchKernelGGL(and_bitmaps_kernel, dim3(egrid((n + 7) / 8)), dim3(256), 0, st, av, bv, (n + 7) / 8, out_valid);
    CRGPU_HIP(hipGetLastError());
  }
  mark_default_work();
  return CRGPU_OK;
}

extern "C" crgpu_status crgpu_dev_select_bytes_const(int32_t cmp, const crgpu_dcol *col, const uint8_t *cst, int32_t cst_len,
                                                     int64_t n, uint8_t *out_bool) {
  CRGPU_ENTER(nullptr);
  if (col == nullptr || cst_len < 0 || (cst_len > 0 && cst == nullptr)) return fail(CRGPU_ERR_INVALID, "crgpu_dev_select_bytes_const: null argument");
  if (col->type != CRGPU_BYTES) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_select_bytes_const: BYTES columns only");
  if (!(cmp == BCMP_EQ || cmp == BCMP_NE || cmp == BCMP_CONTAINS || cmp == BCMP_NOT_CONTAINS))
    return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_select_bytes_const: EQ (0), NE (1), CONTAINS (6) and NOT CONTAINS (7) only");
  if (cst_len > kMaxBytesConst) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_select_bytes_const: constants of up to 64 bytes");
  if (n <= 0) return CRGPU_OK;
  if (col->offsets == nullptr || out_bool == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_select_bytes_const: null offsets / output");
  BytesConst c;
  std::memset(&c, 0, sizeof(c));
  c.len = static_cast<uint32_t>(cst_len);
  if (cst_len) std::memcpy(c.b, cst, static_cast<size_t>(cst_len));
  ProfScope prof("expr_select");
  hipLaunchKernelGGL(select_bytes_const_kernel, dim3(egrid(n)), dim3(256), 0, stream(), static_cast<int>(cmp),
                     static_cast<const uint8_t *>(col->data), col->offsets, col->valid, c, n, out_bool);
  CRGPU_HIP(hipGetLastError());
  mark_default_work();
  return CRGPU_OK;
}

extern "C" crgpu_status crgpu_dev_extract_year(const crgpu_dcol *days, int64_t n, double *out_data) {
  CRGPU_ENTER(nullptr);
  if (days == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_extract_year: null argument");
  if (days->type != CRGPU_INT64) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_extract_year: DATE columns are INT64 days since the epoch");
  if (n <= 0) return CRGPU_OK;
  if (days->data == nullptr || out_data == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_extract_year: null data pointer");
  ProfScope prof("expr_project");
  hipLaunchKernelGGL(extract_year_kernel, dim3(egrid(n)), dim3(256), 0, stream(), static_cast<const long long *>(days->data), n, out_data);
  CRGPU_HIP(hipGetLastError());
  mark_default_work();
  return CRGPU_OK;
}
