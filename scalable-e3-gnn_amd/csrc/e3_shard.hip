// Sharded graphs: drop the edges into ghost rows and split the rest by the ownership of their src -- one classification +
// compaction on the device (sharding.GridHalo.split_graph).  Integer / HBM work: no MFMA; rows are whole (a CSR row is kept
// or dropped as a unit), so one thread per row walks its ~24 edges and every output list stays sorted by dst.
#include "e3_common.h"

#include <hipcub/hipcub.hpp>

namespace e3 {

// per row: edges kept (dst owned), of those with owned src (interior) / ghost src (boundary)
__global__ void split_count_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ src,
                                   const uint8_t* __restrict__ ghost, int64_t N, int32_t* __restrict__ ck,
                                   int32_t* __restrict__ ci, int32_t* __restrict__ cb) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i > N) return;
  int k = 0, ni = 0, nb = 0;
  if (i < N && !ghost[i]) {
    const int b = rowptr[i], e = rowptr[i + 1];
    k = e - b;
    for (int p = b; p < e; ++p) nb += ghost[src[p]] ? 1 : 0;
    ni = k - nb;
  }
  ck[i] = k; ci[i] = ni; cb[i] = nb;   // element N = 0: the exclusive scan leaves the totals there
}

__global__ void split_fill_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ src,
                                  const uint8_t* __restrict__ ghost, int64_t N, const int32_t* __restrict__ ok,
                                  const int32_t* __restrict__ oi, const int32_t* __restrict__ ob, int32_t* __restrict__ src_k,
                                  int32_t* __restrict__ dst_k, int32_t* __restrict__ src_i, int32_t* __restrict__ dst_i,
                                  int32_t* __restrict__ src_b, int32_t* __restrict__ dst_b, int32_t* __restrict__ counts) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i == 0) { counts[0] = ok[N]; counts[1] = oi[N]; counts[2] = ob[N]; counts[3] = 0; }
  if (i >= N || ghost[i]) return;
  const int b = rowptr[i], e = rowptr[i + 1];
  int pk = ok[i], pi = oi[i], pb = ob[i];
  for (int p = b; p < e; ++p) {
    const int s = src[p];
    src_k[pk] = s; dst_k[pk] = (int32_t)i; ++pk;
    if (ghost[s]) { src_b[pb] = s; dst_b[pb] = (int32_t)i; ++pb; }
    else { src_i[pi] = s; dst_i[pi] = (int32_t)i; ++pi; }
  }
}

static size_t scan_temp_bytes(int64_t n) {
  size_t t = 0;
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, t, (const int32_t*)nullptr, (int32_t*)nullptr, (int)n);
  return (t + 255) / 256 * 256;
}

}  // namespace e3

using namespace e3;

extern "C" {

int64_t e3_split_edges_workspace_bytes(int64_t N) {
  if (N < 0 || N + 1 > 0x7fffffffLL) return -1;
  const size_t arr = ((size_t)(N + 1) * 4 + 255) / 256 * 256;
  return (int64_t)(5 * arr + scan_temp_bytes(N + 1));
}

int e3_split_edges(const int32_t* rowptr, const int32_t* src, const uint8_t* is_ghost, int64_t N, int64_t E,
                   int32_t* rowptr_kept, int32_t* src_kept, int32_t* dst_kept, int32_t* src_interior, int32_t* dst_interior,
                   int32_t* src_boundary, int32_t* dst_boundary, int32_t* counts, void* workspace, void* stream) {
  if (N < 0 || E < 0 || N + 1 > 0x7fffffffLL || E > 0x7fffffffLL) return E3_ERR_INVALID_ARG;
  if (!rowptr || !is_ghost || !rowptr_kept || !counts || !workspace || (E > 0 && (!src || !src_kept || !dst_kept ||
      !src_interior || !dst_interior || !src_boundary || !dst_boundary))) return E3_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  const size_t arr = ((size_t)(N + 1) * 4 + 255) / 256 * 256;
  char* w = static_cast<char*>(workspace);
  int32_t *ck = (int32_t*)w, *ci = (int32_t*)(w + arr), *cb = (int32_t*)(w + 2 * arr), *oi = (int32_t*)(w + 3 * arr),
          *ob = (int32_t*)(w + 4 * arr);
  void* temp = w + 5 * arr;
  size_t tb = scan_temp_bytes(N + 1);
  const int threads = 256;
  const unsigned blocks = (unsigned)((N + 1 + threads - 1) / threads);
  hipLaunchKernelGGL(split_count_kernel, dim3(blocks), dim3(threads), 0, s, rowptr, src, is_ghost, N, ck, ci, cb);
  E3_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(temp, tb, ck, rowptr_kept, (int)(N + 1), s));
  E3_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(temp, tb, ci, oi, (int)(N + 1), s));
  E3_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(temp, tb, cb, ob, (int)(N + 1), s));
  hipLaunchKernelGGL(split_fill_kernel, dim3(blocks), dim3(threads), 0, s, rowptr, src, is_ghost, N, rowptr_kept, oi, ob,
                     src_kept, dst_kept, src_interior, dst_interior, src_boundary, dst_boundary, counts);
  E3_HIP_CHECK(hipGetLastError());
  return E3_OK;
}

}  // extern "C"
