// Internal declarations shared by e3_tp.hip (plan, generic kernel, C ABI) and the MFMA path (e3_tp_mfma.hip: host plan and
// weight packing; e3_tp_mfma_r16.hip: the kernel).
#pragma once
#include "e3_common.h"

#include <array>
#include <vector>

namespace e3 {

struct TpPath { int c1, l1, l2, wrow; };

struct FChunk {  // one piece (<= 32 channels) of an in1 irreps block, natural parity class of degree l1
  int col, count, l1;
  int wblk[3][3];  // [l2][l3]: first 16-row block of this chunk's weights in class l3's packed matrix, -1 = no coupling
  // staging geometry, [0] fp32 storage (4 elements per 16-byte unit), [1] bf16 storage (8):
  // S = 16-byte units per LDS row (odd), rows_per = rows one 64-lane DMA instruction covers, inv = ceil(2^16 / S)
  // (lane / S == (lane * inv) >> 16 for lane < 64, checked at plan time)
  int S[2], rows_per[2], inv[2];
};
// Packed MFMA section (floats): [normcol (Dout, padded to 4) | header (4) | Whi | Wlo]
//   header: [0] bits of max |w| (scratch of the pack), [1] sw, [2] 1 / sw, [3] unused
//   fp32 storage: Whi / Wlo = fp16 (hi, lo) split of w * sw, sw = 2^k with max |w| * sw in [2^13, 2^14)
//   bf16 storage: Whi = the bf16 weights, Wlo unused, sw = 1
//   layout of Whi / Wlo: [16-row block][k half][channel][8], so that one lane's A operand is one 16-byte load
struct FDev {
  int Dout, Dy, nchunks, ntab, lsh;
  int M[3], NT[3], Mpad[3], ooff[3];  // per output degree l3 (classes 0e, 1o, 2e)
  int bfoff[3];                        // element offset (uint16) of class l3 inside Whi (and inside Wlo)
  int bftotal;                         // uint16 elements of Whi (== Wlo)
};
struct FPack { int l3, orig_row, count, wblk; };

struct TpFast {
  FDev dev;
  std::vector<FChunk> h_chunks;
  std::vector<FPack> h_pack;
  FChunk* d_chunks = nullptr;
  FPack* d_pack = nullptr;
  FDev* d_dev = nullptr;
  bool usable = false;
  bool gate_layout = false;  // out irreps are [0e scalars+gates | 1o | 2e], each class one contiguous run (fused gate epilogue)
};

int fast_plan_init(TpFast* F, const int n[6], const int M[6], int lmax_sh, int Dout, int Dy,
                   const std::vector<std::array<int, 4>>& in_blocks, const std::vector<TpPath>* paths_by_class,
                   const int ocol_off[6]);
int fast_upload(TpFast* F);
void fast_free(TpFast* F);
int64_t fast_packed_bytes(const TpFast* F);
int fast_pack(const TpFast* F, const void* const w[6], const void* const n[6], int dtype, void* packed,
              const int32_t* ocol_tab, hipStream_t s);
int fast_forward(const TpFast* F, const e3_tp_segment* segs, int nseg, int D1, const void* in2, int64_t ld2,
                 const void* packed, void* out, int64_t ldo, int64_t B, int gate, int dtype, const int32_t* ocol_tab,
                 const float* in_scale, hipStream_t s, const int32_t* scatter = nullptr, const void* residual = nullptr,
                 int64_t ldr = 0, uint32_t* amax = nullptr);
// {s, 1/s} from the float bits at out4[2] (e3_scale.hip)
int scale_finalize(float* out4, int target_log2, hipStream_t s);

// name of the kernel family the last fused forward of this thread launched (diagnostics / bench labels)
void fast_note_kernel(const char* name);
const char* fast_last_kernel();

// out irreps are a gated layout the fused gate epilogue handles ([H x0e | (nb H) x0e | H x1o (| H x2e)], H % 16 == 0)
bool fast_gate_shape_ok(const TpFast* F);
// does the 16-row kernel have an instantiation for this plan?
bool r16_supported(const TpFast* F);
// 16-row kernel (e3_tp_mfma_r16.hip): 1 = launched, 0 = no instantiation, < 0 = -status
int fast_forward_r16(const TpFast* F, const void* seg_args, const void* in2, int64_t ld2, const void* packed, void* out,
                     int64_t ldo, int64_t B, int gate, int io16, const int32_t* ocol_tab, const float* in_scale,
                     hipStream_t s);

}  // namespace e3
