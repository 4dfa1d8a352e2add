// hash.hip -- columnar hashing: initHash -> rehash per key column -> finalizeHash
// (pkg/sql/colexec/colexechash/hash_utils.go:65-123, hash_utils.eg.go:39-1087),
// one row per lane, all key columns folded in registers (the reference makes one
// pass over memory per column; here the running hash never leaves the lane).
//
// Roofline: pure streaming. Algorithmic bytes per row = sum of key widths read
// + 4 or 8 bytes of hash written; HBM-bound.
#include "device.h"

namespace crgpu {

template <typename T>
__global__ __launch_bounds__(256) void hash_cols_kernel(KeyCols keys, int64_t n, uint64_t init, uint64_t nbuckets,
                                                        T *__restrict__ out) {
  const bool pow2 = (nbuckets & (nbuckets - 1)) == 0;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    T h;
    if (sizeof(T) == 4)
      h = static_cast<T>(hash_row32(keys, i, static_cast<uint32_t>(init)));
    else
      h = static_cast<T>(hash_row64(keys, i, init));
    if (nbuckets != 0) {  // finalizeHash: hash_utils.go:104-123
      h = pow2 ? static_cast<T>(h & static_cast<T>(nbuckets - 1)) : static_cast<T>(h % static_cast<T>(nbuckets));
    }
    out[i] = h;
  }
}

crgpu_status launch_hash(const KeyCols &keys, int64_t n, uint64_t init, int width, uint64_t nbuckets, void *out) {
  if (n <= 0) return CRGPU_OK;
  const int block = 256;
  int64_t want = (n + block - 1) / block;
  int grid = static_cast<int>(want < static_cast<int64_t>(num_cus()) * 8 ? want : static_cast<int64_t>(num_cus()) * 8);
  ProfScope prof("hash");
  if (width == 32)
    hipLaunchKernelGGL(hash_cols_kernel<uint32_t>, dim3(grid), dim3(block), 0, stream(), keys, n, init, nbuckets,
                       static_cast<uint32_t *>(out));
  else
    hipLaunchKernelGGL(hash_cols_kernel<uint64_t>, dim3(grid), dim3(block), 0, stream(), keys, n, init, nbuckets,
                       static_cast<uint64_t *>(out));
  CRGPU_HIP(hipGetLastError());
  return CRGPU_OK;
}

}  // namespace crgpu

using namespace crgpu;

extern "C" crgpu_status crgpu_dev_hash_cols(const crgpu_dcol *cols, int32_t ncols, int64_t n, uint64_t init,
                                            int32_t width, uint64_t nbuckets, void *out_dev) {
  CRGPU_ENTER(nullptr);
  if (ncols < 0 || ncols > kMaxKeyCols) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_hash_cols: more than 8 key columns");
  if (width != 32 && width != 64) return fail(CRGPU_ERR_INVALID, "crgpu_dev_hash_cols: width must be 32 or 64");
  if (n < 0 || (n > 0 && out_dev == nullptr)) return fail(CRGPU_ERR_INVALID, "crgpu_dev_hash_cols: bad n/out");
  if (width == 32 && nbuckets > 0xFFFFFFFFull) return fail(CRGPU_ERR_INVALID, "crgpu_dev_hash_cols: nbuckets exceeds 32 bits");
  KeyCols k;
  k.n = ncols;
  k._pad = 0;
  for (int i = 0; i < ncols; i++) {
    if (!valid_type(cols[i].type)) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_hash_cols: unsupported column type");
    if (n > 0 && cols[i].data == nullptr && !(cols[i].type == CRGPU_BYTES))
      return fail(CRGPU_ERR_INVALID, "crgpu_dev_hash_cols: null data pointer");
    if (cols[i].type == CRGPU_BYTES && cols[i].offsets == nullptr)
      return fail(CRGPU_ERR_INVALID, "crgpu_dev_hash_cols: BYTES column without offsets");
    k.c[i] = to_dcol(cols[i]);
  }
  CRGPU_TRY(launch_hash(k, n, init, width, nbuckets, out_dev));
  CRGPU_HIP(hipStreamSynchronize(stream()));
  return CRGPU_OK;
}
