// Shared host/device declarations for libe3gnn_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

#include <mutex>
#include <string>
#include <vector>

#include "../../include/e3gnn.h"

namespace e3 {

// ---- error plumbing -------------------------------------------------------------------------
void set_hip_error(hipError_t e, const char* what);

#define E3_HIP_CHECK(expr)                                   \
  do {                                                       \
    hipError_t _e = (expr);                                  \
    if (_e != hipSuccess) {                                  \
      ::e3::set_hip_error(_e, #expr);                        \
      return E3_ERR_HIP;                                     \
    }                                                        \
  } while (0)

// ---- element types --------------------------------------------------------------------------
using bf16 = __hip_bfloat16;

template <typename T> struct AccOf { using type = float; };
template <> struct AccOf<double> { using type = double; };

template <typename T> __device__ __forceinline__ typename AccOf<T>::type to_acc(T v) { return v; }
template <> __device__ __forceinline__ float to_acc<bf16>(bf16 v) { return __bfloat162float(v); }

template <typename T, typename A> __device__ __forceinline__ T from_acc(A v) { return static_cast<T>(v); }
template <> __device__ __forceinline__ bf16 from_acc<bf16, float>(float v) { return __float2bfloat16(v); }

// ---- irreps bookkeeping (host) ----------------------------------------------------------------
struct Block {  // one `mul x (l,p)` entry of an Irreps, with its first column
  int l, p, mul, col;
};

// One contiguous run of channels of a class inside a row: channel i of the run starts at column
// col + i*cstride (cstride = 1 for scalars, 3 for vectors whose 3 components are adjacent).
struct Run {
  int col, count, cstride;
};

// Device-visible plan (POD, passed by value as a kernel argument).
struct PlanDev {
  int D1, Dout;
  int n[4];          // in1 channel counts per class: n0e, n0o, n1e, n1o
  int M[4];          // out multiplicities per class
  int cbase[4];      // first canonical position of class c inside a staged in1 row
  int obase[4];      // first canonical position of class c inside a staged out / grad_out row
  int icol_off[4];   // offset of class c inside `icol`
  int ocol_off[4];   // offset of class c inside `ocol`
  const int32_t* cpos;  // [D1]   in1 column -> canonical position ([s0e|s0o|v1e|v1o], xyz adjacent)
  const int32_t* opos;  // [Dout] out column -> canonical position (same class order)
  const int32_t* icol;  // [sum n] first in1 column of channel k of class c
  const int32_t* ocol;  // [sum M] first out column of channel m of class c
};

struct Mfma;

}  // namespace e3

struct e3_l1tp_plan {
  std::vector<e3::Block> in1, out;
  std::vector<e3::Run> irun[4], orun[4];
  e3::PlanDev dev;
  std::vector<int32_t> h_tables;  // [cpos (D1) | opos (Dout) | icol (sum n) | ocol (sum M)]
  int32_t* d_tables = nullptr;    // uploaded on first use
  int device = -1;                // ... to the device current at that moment; calls from another device fail
  std::mutex mu;
  int wrows[4], wcols[4];
  int normlen[4];
  // MFMA path (filled by mfma_plan_init)
  e3::Mfma* mfma = nullptr;
};

namespace e3 {
// mfma path (e3_l1tp_mfma.hip)
int mfma_plan_init(e3_l1tp_plan* plan);    // host only
int mfma_plan_upload(e3_l1tp_plan* plan);  // device tables
void mfma_plan_free(e3_l1tp_plan* plan);
bool mfma_supported(const e3_l1tp_plan* plan, int dtype);
int64_t mfma_packed_bytes(const e3_l1tp_plan* plan);
int mfma_pack(const e3_l1tp_plan* plan, const void* const weights[4], const void* const norms[4], int dtype,
              void* packed, hipStream_t stream);
int mfma_forward(const e3_l1tp_plan* plan, const void* in1, int64_t ld1, const void* in2, int64_t ld2,
                 const void* packed, void* out, int64_t ldo, int64_t B, int dtype, hipStream_t stream);
// shared by e3_l1tp.hip / e3_l1tp_bwd.hip
int ensure_device(const e3_l1tp_plan* plan);
constexpr double kC3 = 0.57735026918962576451;  // 1/sqrt(3)  cg110 = cg011, L1TP.py:92-93
constexpr double kC6 = 0.40824829046386301637;  // 1/sqrt(6)  cg111,         L1TP.py:94
}  // namespace e3
