// gather.hip -- shared materialisation utilities: gathers through row-id vectors
// (Vec.Copy with a selection vector, pkg/col/coldata/vec_tmpl.go:187-269),
// prefix sums, stream compaction and bitmap packing.
#include "device.h"

namespace crgpu {

static constexpr int kScanBlock = 256;
static constexpr int kScanItems = 16;
static constexpr int kScanTile = kScanBlock * kScanItems;  // 4096 elements per workgroup

static inline int grid_for(int64_t n, int block, int per_thread = 1) {
  int64_t want = (n + static_cast<int64_t>(block) * per_thread - 1) / (static_cast<int64_t>(block) * per_thread);
  int64_t cap = static_cast<int64_t>(num_cus()) * 8;
  if (want < 1) want = 1;
  return static_cast<int>(want < cap ? want : cap);
}

// ------------------------------------------------------------------ fixed-width gather
template <typename T>
__global__ __launch_bounds__(256) void gather_fixed_kernel(const T *__restrict__ src, const int32_t *__restrict__ idx,
                                                           int64_t n, T *__restrict__ dst) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    int32_t r = idx[i];
    dst[i] = r >= 0 ? src[r] : T(0);
  }
}

crgpu_status gather_fixed(const DCol &src, const int32_t *idx, int64_t n, void *dst) {
  if (n <= 0) return CRGPU_OK;
  int grid = grid_for(n, 256);
  switch (type_width(src.type)) {
    case 1:
      hipLaunchKernelGGL(gather_fixed_kernel<uint8_t>, dim3(grid), dim3(256), 0, stream(),
                         static_cast<const uint8_t *>(src.data), idx, n, static_cast<uint8_t *>(dst));
      break;
    case 2:
      hipLaunchKernelGGL(gather_fixed_kernel<uint16_t>, dim3(grid), dim3(256), 0, stream(),
                         static_cast<const uint16_t *>(src.data), idx, n, static_cast<uint16_t *>(dst));
      break;
    case 4:
      hipLaunchKernelGGL(gather_fixed_kernel<uint32_t>, dim3(grid), dim3(256), 0, stream(),
                         static_cast<const uint32_t *>(src.data), idx, n, static_cast<uint32_t *>(dst));
      break;
    case 8:
      hipLaunchKernelGGL(gather_fixed_kernel<uint64_t>, dim3(grid), dim3(256), 0, stream(),
                         static_cast<const uint64_t *>(src.data), idx, n, static_cast<uint64_t *>(dst));
      break;
    default: return fail(CRGPU_ERR_INVALID, "gather_fixed: not a fixed-width type");
  }
  CRGPU_HIP(hipGetLastError());
  return CRGPU_OK;
}

// ------------------------------------------------------------------ validity gather (one output byte of 8 rows per lane)
__global__ __launch_bounds__(256) void gather_valid_kernel(DCol src, const int32_t *__restrict__ idx, int64_t n,
                                                           uint8_t *__restrict__ dst) {
  const int64_t nbytes = (n + 7) >> 3;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t b = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; b < nbytes; b += stride) {
    uint32_t bits = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      int64_t i = b * 8 + k;
      uint32_t present = 1;  // trailing bits are set to 1 (col/coldata/nulls.go:35-41)
      if (i < n) {
        int32_t r = idx[i];
        present = r >= 0 && !col_is_null(src, r);
      }
      bits |= present << k;
    }
    dst[b] = static_cast<uint8_t>(bits);
  }
}

crgpu_status gather_valid(const DCol &src, const int32_t *idx, int64_t n, uint8_t *dst_bitmap) {
  if (n <= 0) return CRGPU_OK;
  hipLaunchKernelGGL(gather_valid_kernel, dim3(grid_for((n + 7) / 8, 256)), dim3(256), 0, stream(), src, idx, n,
                     dst_bitmap);
  CRGPU_HIP(hipGetLastError());
  return CRGPU_OK;
}

__global__ __launch_bounds__(256) void pack_bitmap_kernel(const uint8_t *__restrict__ bytes, int64_t n,
                                                          uint8_t *__restrict__ bitmap) {
  const int64_t nbytes = (n + 7) >> 3;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t b = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; b < nbytes; b += stride) {
    uint32_t bits = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      int64_t i = b * 8 + k;
      uint32_t present = i < n ? (bytes[i] != 0) : 1;
      bits |= present << k;
    }
    bitmap[b] = static_cast<uint8_t>(bits);
  }
}

crgpu_status pack_bitmap(const uint8_t *bytes, int64_t n, uint8_t *bitmap) {
  if (n <= 0) return CRGPU_OK;
  hipLaunchKernelGGL(pack_bitmap_kernel, dim3(grid_for((n + 7) / 8, 256)), dim3(256), 0, stream(), bytes, n, bitmap);
  CRGPU_HIP(hipGetLastError());
  return CRGPU_OK;
}

// ------------------------------------------------------------------ scans (reduce-then-scan over 4096-element tiles)
__device__ __forceinline__ uint64_t block_excl_scan_u64(uint64_t v, uint64_t *lds /*[>=5]*/, uint64_t *total) {
  // 256 threads = 4 waves
  uint64_t incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint64_t t = __shfl_up(incl, d, 64);
    if (lane_id() >= static_cast<uint32_t>(d)) incl += t;
  }
  const int w = threadIdx.x >> 6;
  if (lane_id() == 63) lds[w] = incl;
  __syncthreads();
  uint64_t base = 0;
  for (int k = 0; k < w; k++) base += lds[k];
  if (total) *total = lds[0] + lds[1] + lds[2] + lds[3];
  __syncthreads();
  return base + incl - v;
}

template <typename In>
__global__ __launch_bounds__(kScanBlock) void tile_sum_kernel(const In *__restrict__ in, int64_t n,
                                                              uint64_t *__restrict__ tile_sums) {
  __shared__ uint64_t lds[4];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kScanTile;
  uint64_t s = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; k++) {
    int64_t i = base + k * kScanBlock + threadIdx.x;
    if (i < n) s += (sizeof(In) == 1) ? static_cast<uint64_t>(in[i] != 0) : static_cast<uint64_t>(in[i]);
  }
  uint64_t total;
  block_excl_scan_u64(s, lds, &total);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

// single workgroup: in-place exclusive scan of tile sums; writes grand total to tile_sums[ntiles].  Every thread owns a
// contiguous run of the sums (summed, ONE block scan over the runs' totals, then written back): a block scan per 256
// sums took 97 us for the 32768 tiles of 2^27 elements, two barriers apiece.
__global__ __launch_bounds__(kScanBlock) void scan_tile_sums_kernel(uint64_t *__restrict__ tile_sums, int64_t ntiles) {
  __shared__ uint64_t lds[4];
  const int64_t per = (ntiles + kScanBlock - 1) / kScanBlock;
  const int64_t lo = static_cast<int64_t>(threadIdx.x) * per;
  const int64_t hi = lo + per < ntiles ? lo + per : ntiles;
  uint64_t s = 0;
  for (int64_t i = lo; i < hi; i++) s += tile_sums[i];
  uint64_t total;
  uint64_t ex = block_excl_scan_u64(s, lds, &total);
  for (int64_t i = lo; i < hi; i++) {
    const uint64_t v = tile_sums[i];
    tile_sums[i] = ex;
    ex += v;
  }
  if (threadIdx.x == 0) tile_sums[ntiles] = total;
}

__global__ __launch_bounds__(kScanBlock) void scan_apply_u32_kernel(const uint32_t *__restrict__ in, int64_t n,
                                                                    const uint64_t *__restrict__ tile_sums,
                                                                    uint64_t *__restrict__ out) {
  __shared__ uint64_t lds[4];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kScanTile;
  uint64_t carry = tile_sums[blockIdx.x];
  // blocked arrangement: thread t owns elements [t*16, t*16+16) of the tile
  uint32_t v[kScanItems];
  uint64_t s = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; k++) {
    int64_t i = base + static_cast<int64_t>(threadIdx.x) * kScanItems + k;
    v[k] = i < n ? in[i] : 0;
    s += v[k];
  }
  uint64_t ex = block_excl_scan_u64(s, lds, nullptr) + carry;
#pragma unroll
  for (int k = 0; k < kScanItems; k++) {
    int64_t i = base + static_cast<int64_t>(threadIdx.x) * kScanItems + k;
    if (i < n) out[i] = ex;
    ex += v[k];
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) out[n] = tile_sums[gridDim.x];
}

size_t scan_scratch_words(int64_t n) { return static_cast<size_t>((n > 0 ? (n + kScanTile - 1) / kScanTile : 0) + 1); }

crgpu_status exclusive_scan_u32_ws(const uint32_t *counts, int64_t n, uint64_t *offsets, uint64_t *ts, uint64_t *total_host) {
  if (n <= 0) {
    if (total_host) *total_host = 0;
    if (offsets) CRGPU_HIP(hipMemsetAsync(offsets, 0, 8, stream()));
    return CRGPU_OK;
  }
  int64_t ntiles = (n + kScanTile - 1) / kScanTile;
  hipLaunchKernelGGL(tile_sum_kernel<uint32_t>, dim3(ntiles), dim3(kScanBlock), 0, stream(), counts, n, ts);
  hipLaunchKernelGGL(scan_tile_sums_kernel, dim3(1), dim3(kScanBlock), 0, stream(), ts, ntiles);
  hipLaunchKernelGGL(scan_apply_u32_kernel, dim3(ntiles), dim3(kScanBlock), 0, stream(), counts, n, ts, offsets);
  CRGPU_HIP(hipGetLastError());
  if (total_host) {
    CRGPU_HIP(hipMemcpyAsync(total_host, ts + ntiles, 8, hipMemcpyDeviceToHost, stream()));
    CRGPU_HIP(hipStreamSynchronize(stream()));
  }
  return CRGPU_OK;
}

crgpu_status exclusive_scan_u32(const uint32_t *counts, int64_t n, uint64_t *offsets, DevBuf &scratch,
                                uint64_t *total_host) {
  if (n > 0) CRGPU_TRY(scratch.ensure(scan_scratch_words(n) * 8));
  return exclusive_scan_u32_ws(counts, n, offsets, scratch.as<uint64_t>(), total_host);
}

__global__ __launch_bounds__(kScanBlock) void select_apply_kernel(const uint8_t *__restrict__ flags, int64_t n,
                                                                  const uint64_t *__restrict__ tile_sums,
                                                                  int32_t *__restrict__ idx_out) {
  __shared__ uint64_t lds[4];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kScanTile;
  uint64_t carry = tile_sums[blockIdx.x];
  uint32_t f = 0;
  uint64_t s = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; k++) {
    int64_t i = base + static_cast<int64_t>(threadIdx.x) * kScanItems + k;
    uint32_t b = i < n ? (flags[i] != 0) : 0;
    f |= b << k;
    s += b;
  }
  uint64_t ex = block_excl_scan_u64(s, lds, nullptr) + carry;
#pragma unroll
  for (int k = 0; k < kScanItems; k++) {
    if ((f >> k) & 1) {
      idx_out[ex++] = static_cast<int32_t>(base + static_cast<int64_t>(threadIdx.x) * kScanItems + k);
    }
  }
}

// The same two passes for a 16-byte aligned flag array: a thread's 16 flags (blocked arrangement, as in
// select_apply_kernel) arrive in ONE 16-byte load instead of 16 byte loads (measured per 2^27 flags: DESIGN.md 4.5).
// nonzero_bytes: bit 7 of every byte of w that is not zero.
__device__ __forceinline__ uint32_t nonzero_bytes(uint32_t w) { return (w | ((w & 0x7F7F7F7Fu) + 0x7F7F7F7Fu)) & 0x80808080u; }
__device__ __forceinline__ uint32_t flag_bits16(const uint8_t *__restrict__ flags, int64_t first, int64_t n) {
  // bit k = flags[first + k] != 0, k < 16; first is a multiple of 16
  uint32_t f = 0;
  if (first + 16 <= n) {
    const uint4 v = *reinterpret_cast<const uint4 *>(flags + first);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const uint32_t nz = nonzero_bytes(w[q]);  // bits 7, 15, 23, 31 -> bits 4q .. 4q + 3
      f |= (((nz >> 7) & 1u) | ((nz >> 14) & 2u) | ((nz >> 21) & 4u) | ((nz >> 28) & 8u)) << (4 * q);
    }
  } else {
    for (int k = 0; k < 16; k++)
      if (first + k < n && flags[first + k] != 0) f |= 1u << k;
  }
  return f;
}
__global__ __launch_bounds__(kScanBlock) void tile_sum_flags16_kernel(const uint8_t *__restrict__ flags, int64_t n, uint64_t *__restrict__ tile_sums) {
  static_assert(kScanItems == 16, "a thread owns 16 flags");
  __shared__ uint64_t lds[4];
  const int64_t first = static_cast<int64_t>(blockIdx.x) * kScanTile + static_cast<int64_t>(threadIdx.x) * kScanItems;
  const uint64_t s = first < n ? static_cast<uint64_t>(__popc(flag_bits16(flags, first, n))) : 0;
  uint64_t total;
  block_excl_scan_u64(s, lds, &total);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}
__global__ __launch_bounds__(kScanBlock) void select_apply_flags16_kernel(const uint8_t *__restrict__ flags, int64_t n,
                                                                          const uint64_t *__restrict__ tile_sums, int32_t *__restrict__ idx_out) {
  __shared__ uint64_t lds[4];
  const int64_t first = static_cast<int64_t>(blockIdx.x) * kScanTile + static_cast<int64_t>(threadIdx.x) * kScanItems;
  uint32_t f = first < n ? flag_bits16(flags, first, n) : 0;
  uint64_t ex = block_excl_scan_u64(static_cast<uint64_t>(__popc(f)), lds, nullptr) + tile_sums[blockIdx.x];
  while (f) {
    const int k = __ffs(static_cast<int>(f)) - 1;
    f &= f - 1;
    idx_out[ex++] = static_cast<int32_t>(first + k);
  }
}

crgpu_status select_flagged(const uint8_t *flags, int64_t n, int32_t *idx_out, DevBuf &scratch, int64_t *count_host, bool skip_if_all) {
  if (n <= 0) {
    if (count_host) *count_host = 0;
    return CRGPU_OK;
  }
  int64_t ntiles = (n + kScanTile - 1) / kScanTile;
  CRGPU_TRY(scratch.ensure(static_cast<size_t>(ntiles + 1) * 8));
  uint64_t *ts = scratch.as<uint64_t>();
  const bool vec = (reinterpret_cast<uintptr_t>(flags) & 15u) == 0;
  if (vec) hipLaunchKernelGGL(tile_sum_flags16_kernel, dim3(ntiles), dim3(kScanBlock), 0, stream(), flags, n, ts);
  else hipLaunchKernelGGL(tile_sum_kernel<uint8_t>, dim3(ntiles), dim3(kScanBlock), 0, stream(), flags, n, ts);
  hipLaunchKernelGGL(scan_tile_sums_kernel, dim3(1), dim3(kScanBlock), 0, stream(), ts, ntiles);
  uint64_t total = 0;
  if (skip_if_all) {  // the count first: the positions of n flagged rows out of n are not worth writing
    CRGPU_HIP(hipGetLastError());
    CRGPU_HIP(hipMemcpyAsync(&total, ts + ntiles, 8, hipMemcpyDeviceToHost, stream()));
    CRGPU_HIP(hipStreamSynchronize(stream()));
    if (count_host) *count_host = static_cast<int64_t>(total);
    if (total != static_cast<uint64_t>(n)) {
      if (vec) hipLaunchKernelGGL(select_apply_flags16_kernel, dim3(ntiles), dim3(kScanBlock), 0, stream(), flags, n, ts, idx_out);
      else hipLaunchKernelGGL(select_apply_kernel, dim3(ntiles), dim3(kScanBlock), 0, stream(), flags, n, ts, idx_out);
    }
    CRGPU_HIP(hipGetLastError());
    return CRGPU_OK;
  }
  if (vec) hipLaunchKernelGGL(select_apply_flags16_kernel, dim3(ntiles), dim3(kScanBlock), 0, stream(), flags, n, ts, idx_out);
  else hipLaunchKernelGGL(select_apply_kernel, dim3(ntiles), dim3(kScanBlock), 0, stream(), flags, n, ts, idx_out);
  CRGPU_HIP(hipGetLastError());
  CRGPU_HIP(hipMemcpyAsync(&total, ts + ntiles, 8, hipMemcpyDeviceToHost, stream()));
  CRGPU_HIP(hipStreamSynchronize(stream()));
  if (count_host) *count_host = static_cast<int64_t>(total);
  return CRGPU_OK;
}

// ------------------------------------------------------------------ BYTES gather
__global__ __launch_bounds__(256) void bytes_len_kernel(DCol src, const int32_t *__restrict__ idx, int64_t n,
                                                        uint32_t *__restrict__ lens) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    int32_t r = idx[i];
    lens[i] = (r >= 0 && !col_is_null(src, r)) ? src.offsets[r + 1] - src.offsets[r] : 0;
  }
}
__global__ __launch_bounds__(256) void narrow_offsets_kernel(const uint64_t *__restrict__ in, int64_t n1,
                                                             uint32_t *__restrict__ out) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n1; i += stride)
    out[i] = static_cast<uint32_t>(in[i]);
}
// The payload of a gathered BYTES column, written by OUTPUT position: the output bytes of a tile of 1024 rows are one
// contiguous range, so consecutive lanes store consecutive bytes (whole lines leave the CU) and find their row by a
// binary search over the tile's offsets in LDS.  One thread per row copying its bytes one by one wrote a byte per lane
// to addresses a value apart: 2.0 ms for the 32 M nation names of Q9 at SF100 (288 MB).
__global__ __launch_bounds__(256) void bytes_copy_kernel(DCol src, const int32_t *__restrict__ idx, int64_t n,
                                                         const uint32_t *__restrict__ dst_off,
                                                         uint8_t *__restrict__ dst) {
  constexpr int kTile = 1024;
  __shared__ uint32_t s_off[kTile + 1];
  __shared__ uint32_t s_src[kTile];
  const uint8_t *sdata = static_cast<const uint8_t *>(src.data);
  const int64_t ntiles = (n + kTile - 1) / kTile;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t r0 = tile * kTile;
    const int rows = static_cast<int>(n - r0 < kTile ? n - r0 : kTile);
    for (int t = threadIdx.x; t <= rows; t += 256) s_off[t] = dst_off[r0 + t];
    for (int t = threadIdx.x; t < rows; t += 256) {
      const int32_t r = idx[r0 + t];
      s_src[t] = r >= 0 ? src.offsets[r] : 0u;  // (rows without a value have length 0: never looked up)
    }
    __syncthreads();
    const uint32_t lo = s_off[0], hi = s_off[rows];
    // four output bytes per thread and step: their searches and loads are independent, so their latencies overlap
    for (uint32_t p0 = lo + threadIdx.x; p0 < hi; p0 += 4 * 256) {
      uint8_t v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const uint32_t p = p0 + u * 256;
        int a = 0, b = rows;  // the last row whose offset is <= p
#pragma unroll
        for (int step = 0; step < 10; step++) {  // kTile = 2^10
          const int m = (a + b) >> 1;
          const bool right = b - a > 1 && s_off[m] <= p;
          const bool left = b - a > 1 && !right;
          a = right ? m : a;
          b = left ? m : b;
        }
        v[u] = p < hi ? sdata[s_src[a] + (p - s_off[a])] : 0;
      }
#pragma unroll
      for (int u = 0; u < 4; u++)
        if (p0 + u * 256 < hi) dst[p0 + u * 256] = v[u];
    }
    __syncthreads();
  }
}

crgpu_status gather_bytes(const DCol &src, const int32_t *idx, int64_t n, DevBuf &dst_offsets, DevBuf &dst_data,
                          DevBuf &scratch) {
  CRGPU_TRY(dst_offsets.ensure(static_cast<size_t>(n + 1) * 4));
  if (n <= 0) {
    CRGPU_HIP(hipMemsetAsync(dst_offsets.p, 0, 4, stream()));
    return CRGPU_OK;
  }
  // scratch layout: [lens u32 n][offsets u64 n+1][tile sums]
  DevBuf lens, offs64;
  CRGPU_TRY(lens.ensure(static_cast<size_t>(n) * 4));
  CRGPU_TRY(offs64.ensure(static_cast<size_t>(n + 1) * 8));
  hipLaunchKernelGGL(bytes_len_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream(), src, idx, n, lens.as<uint32_t>());
  uint64_t total = 0;
  CRGPU_TRY(exclusive_scan_u32(lens.as<uint32_t>(), n, offs64.as<uint64_t>(), scratch, &total));
  if (total > 0xFFFFFFFFull)
    return fail(CRGPU_ERR_UNSUPPORTED, "gather_bytes: a BYTES output column exceeds 4 GiB; probe in smaller chunks");
  hipLaunchKernelGGL(narrow_offsets_kernel, dim3(grid_for(n + 1, 256)), dim3(256), 0, stream(), offs64.as<uint64_t>(),
                     n + 1, dst_offsets.as<uint32_t>());
  CRGPU_TRY(dst_data.ensure(static_cast<size_t>(total ? total : 1)));
  hipLaunchKernelGGL(bytes_copy_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, stream(), src, idx, n,
                     dst_offsets.as<uint32_t>(), dst_data.as<uint8_t>());
  CRGPU_HIP(hipGetLastError());
  CRGPU_HIP(hipStreamSynchronize(stream()));  // lens/offs64 are freed on return
  return CRGPU_OK;
}

crgpu_status materialize(const DCol &src, const int32_t *idx, int64_t n, bool force_nullable, OutCol &out,
                         DevBuf &scratch) {
  out.type = src.type;
  out.nullable = force_nullable || src.valid != nullptr;
  if (src.type == CRGPU_BYTES) {
    CRGPU_TRY(gather_bytes(src, idx, n, out.offsets, out.data, scratch));
  } else {
    CRGPU_TRY(out.data.ensure(static_cast<size_t>(n > 0 ? n : 1) * type_width(src.type)));
    CRGPU_TRY(gather_fixed(src, idx, n, out.data.p));
  }
  if (out.nullable) {
    CRGPU_TRY(out.valid.ensure(static_cast<size_t>((n + 7) / 8 + 1)));
    CRGPU_TRY(gather_valid(src, idx, n, out.valid.as<uint8_t>()));
  }
  return CRGPU_OK;
}

}  // namespace crgpu
