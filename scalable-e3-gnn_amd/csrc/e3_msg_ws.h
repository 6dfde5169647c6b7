// Weights-stationary form of the fused message kernel (e3_msg_ws.hip); launched by e3_msg_forward (e3_msg_fused.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
namespace e3 {
bool msg_ws_supported(int lmax, int hidden, int dtype);
// Same operands as e3_msg_forward after its checks; `out` already zeroed (or holding the sums to continue).  chunk_edges:
// edges per chunk of the workgroups' round-robin (0 = default 256).  `premix` = what e3_msg_premix wrote: N table rows, then the
// N per-node row maxima of h * in_scale.
int msg_ws_launch(int lmax, int hidden, int dtype, const void* h, int64_t ldh, int64_t N, const float* pos4, const int32_t* src,
                  const int32_t* dst, int64_t E, const void* packed, const float* in_scale, const float* premix, float* out,
                  int64_t ldo, int chunk_edges, hipStream_t stream);
}  // namespace e3
