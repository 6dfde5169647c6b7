// Internal declarations shared by e3_tp.hip (plan, generic kernel, C ABI) and e3_tp_mfma.hip (MFMA kernel).
#pragma once
#include "e3_common.h"

#include <array>
#include <vector>

namespace e3 {

struct TpPath { int c1, l1, l2, wrow; };

struct FChunk {  // one piece (<= 32 channels) of an in1 irreps block, natural parity class of degree l1
  int col, count, l1;
  int wrow[3][3];  // [l2][l3]: first packed weight row in class l3's matrix, -1 = no coupling
  int wblk[3][3];  // same for the bf16-split layout, in blocks of 16 rows
  // staging geometry of the bf16-pipe kernels, [0] fp32 storage (4 elements per 16-byte unit), [1] bf16 storage (8):
  // S = 16-byte units per LDS row (odd), rows_per = rows one 64-lane DMA instruction covers, inv = ceil(2^16 / S)
  // (lane / S == (lane * inv) >> 16 for lane < 64, checked at plan time)
  int S[2], rows_per[2], inv[2];
};
struct FDev {
  int Dout, Dy, nchunks, nwaves, nbuf, w_in_lds, wtotal, ntab, lsh;
  int M[3], NT[3], Mpad[3], woff[3], ooff[3];  // per output degree l3 (classes 0e, 1o, 2e)
  // bf16-split variant (BF): weights as hi/lo bf16 in [16-row block][k half][channel][8] order
  int bf;            // fp32-storage mode: 1 = bf16x3-split kernel, 0 = exact fp32 MFMA kernel
  int bfoff[3];      // element offset (uint16) of class l3 inside Whi (and inside Wlo)
  int bftotal;       // uint16 elements of Whi (== Wlo)
  unsigned long long* prof;  // per-phase cycle sums (E3_TP_DBG & 8), else nullptr
  int dbg;           // diagnostic build knobs (E3_TP_DBG): 1 = skip output stores, 2 = stage inputs only for the first tile, 4 = skip MFMA runs, 8 = phase timers, 16 = skip the LDS-DMA instructions, 32 = skip the weight preload
};
struct FPack { int l3, orig_row, count, wrow, wblk; };

struct TpFast {
  FDev dev;    // plan for fp32 storage (exact or bf16x3-split operands)
  FDev dev16;  // plan for bf16 storage (smaller chunk buffers, hi-only weights): differs in nwaves/nbuf/w_in_lds
  FDev* d_dev16 = nullptr;
  size_t lds_bytes16 = 0;
  std::vector<FChunk> h_chunks;
  std::vector<FPack> h_pack;
  FChunk* d_chunks = nullptr;
  FPack* d_pack = nullptr;
  FDev* d_dev = nullptr;
  bool usable = false;
  bool gate_layout = false;  // out irreps are [0e scalars+gates | 1o | 2e], each class one contiguous run (fused gate epilogue)
  size_t lds_bytes = 0;
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
                 hipStream_t s, const int32_t* scatter = nullptr);

// name of the kernel family the last fused forward of this thread launched (diagnostics / bench labels)
void fast_note_kernel(const char* name);
const char* fast_last_kernel();

// two-waves-per-tile kernel (e3_tp_mfma_ab.hip): 1 = launched, 0 = no instantiation / disabled, < 0 = -status
int fast_forward_ab(const TpFast* F, const void* seg_args, const void* in2, int64_t ld2, const void* packed, void* out,
                    int64_t ldo, int64_t B, int gate, int mode, const int32_t* ocol_tab, hipStream_t s);

// 16-row kernel (e3_tp_mfma_r16.hip): same contract as fast_forward_ab
int fast_forward_r16(const TpFast* F, const void* seg_args, const void* in2, int64_t ld2, const void* packed, void* out,
                     int64_t ldo, int64_t B, int gate, int mode, const int32_t* ocol_tab, hipStream_t s);

}  // namespace e3
