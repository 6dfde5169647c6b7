// Host side of the MFMA path of the general SH tensor product (l <= 2, natural-parity classes 0e / 1o / 2e): chunk plan,
// weight packing (fp16 hi/lo split with a power-of-two scale for fp32 storage, plain bf16 for bf16 storage) and dispatch
// to the 16-row kernel (e3_tp_mfma_r16.hip).  Shapes without an instantiation run on the generic FMA kernel (e3_tp.hip),
// which is also the exact-fp32 / fp64 path.
#include "e3_common.h"
#include "e3_tp_internal.h"

#include <algorithm>
#include <cstdlib>
#include <vector>

namespace e3 {

#include "e3_tp_mfma_core.h"

// max |w| over the three natural-parity class matrices -> header[0] (float bits, atomicMax; zeroed by the caller)
template <typename T>
__global__ void fast_absmax_kernel(const T* w0, int64_t n0, const T* w1, int64_t n1, const T* w2, int64_t n2,
                                   uint32_t* hdr) {
  const T* w[3] = {w0, w1, w2};
  const int64_t n[3] = {n0, n1, n2};
  float m = 0.f;
  for (int c = 0; c < 3; ++c)
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n[c]; i += (int64_t)gridDim.x * blockDim.x)
      m = fmaxf(m, fabsf(to_acc(w[c][i])));
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0 && m > 0.f && m < INFINITY) atomicMax(hdr, __builtin_bit_cast(uint32_t, m));
}

template <typename T>
__global__ void fast_pack_kernel(const T* w0, const T* w1, const T* w2, const T* n0, const T* n1, const T* n2,
                                 float* packed, FDev d, const FPack* pk, int npk, const int32_t* ocol_tab) {
  constexpr bool IO16 = std::is_same<T, bf16>::value;
  const T* w[3] = {w0, w1, w2};
  const T* nr[3] = {n0, n1, n2};
  float* hdr = packed + ((d.Dout + 3) & ~3);
  const float sw = IO16 ? 1.0f : pow2_scale_from_bits(reinterpret_cast<const uint32_t*>(hdr)[0], 13);
  uint16_t* whi = reinterpret_cast<uint16_t*>(hdr + 4);
  uint16_t* wlo = whi + d.bftotal;
  for (int r = blockIdx.x; r < npk; r += gridDim.x) {
    const FPack q = pk[r];
    const int M = d.M[q.l3], Mpad = d.Mpad[q.l3];
    for (int i = threadIdx.x; i < q.count * M; i += blockDim.x) {
      const int k = i / M, mm = i - k * M;
      const float v = to_acc(w[q.l3][(int64_t)(q.orig_row + k) * M + mm]) * sw;
      // layout [16-row block][k half][channel][8]
      const size_t e = (size_t)d.bfoff[q.l3] + ((size_t)(2 * (q.wblk + (k >> 4)) + ((k >> 3) & 1)) * Mpad + mm) * 8 + (k & 7);
      if (IO16) {
        whi[e] = __builtin_bit_cast(uint16_t, (__bf16)v);
      } else {
        const _Float16 h = (_Float16)v;
        const _Float16 l = (_Float16)(v - (float)h);
        whi[e] = __builtin_bit_cast(uint16_t, h);
        wlo[e] = __builtin_bit_cast(uint16_t, l);
      }
    }
  }
  if (blockIdx.x == 0) {
    for (int l3 = 0; l3 < 3; ++l3) {
      const int width = 2 * l3 + 1;
      for (int i = threadIdx.x; i < d.M[l3] * width; i += blockDim.x) {
        const int mm = i / width, comp = i - mm * width;
        packed[ocol_tab[d.ooff[l3] + mm] + comp] = nr[l3] ? to_acc(nr[l3][i]) : 1.0f;
      }
    }
    if (threadIdx.x == 0) { hdr[1] = sw; hdr[2] = 1.0f / sw; }
  }
}

// gated out irreps [H x0e | (nb H) x0e | H x1o (| H x2e)] with H a multiple of 16 (the fused gate epilogue of the 16-row kernel
// finds the gate of channel c of block b at scalar channel b H + c: same lane, same register)
bool fast_gate_shape_ok(const TpFast* F) {
  const FDev& d = F->dev;
  const int nb = (d.NT[1] > 0) + (d.NT[2] > 0);
  if (!F->gate_layout || nb == 0 || d.M[0] % (1 + nb)) return false;
  const int H = d.M[0] / (1 + nb);
  return H > 0 && H % 16 == 0 && (!d.NT[1] || d.M[1] == H) && (!d.NT[2] || d.M[2] == H);
}

int fast_plan_init(TpFast* F, const int n[6], const int M[6], int lmax_sh, int Dout, int Dy,
                   const std::vector<std::array<int, 4>>& in_blocks /* l,p,mul,col */,
                   const std::vector<TpPath>* paths_by_class /*[6]*/, const int ocol_off[6]) {
  F->usable = false;
  if (n[1] || n[2] || n[5] || M[1] || M[2] || M[5]) return E3_OK;  // natural parity only
  const int cls3[3] = {0, 3, 4};
  FDev& d = F->dev;
  d.Dout = Dout; d.Dy = Dy;
  int ntab = 0;
  for (int c = 0; c < 6; ++c) ntab += M[c];
  d.ntab = ntab;
  for (int l3 = 0; l3 < 3; ++l3) {
    d.M[l3] = M[cls3[l3]];
    d.NT[l3] = (d.M[l3] + 31) / 32;
    d.Mpad[l3] = d.NT[l3] * 32;
    d.ooff[l3] = ocol_off[cls3[l3]];
  }
  d.lsh = lmax_sh;
  int next_blk[3] = {0, 0, 0};
  int chan_seen[3] = {0, 0, 0};  // channels of in class l1 seen so far
  for (auto& b : in_blocks) {
    const int l1 = b[0], mul = b[2];
    for (int c0 = 0; c0 < mul; c0 += 32) {
      FChunk ch;
      ch.col = b[3] + c0 * (2 * l1 + 1);
      ch.count = std::min(32, mul - c0);
      ch.l1 = l1;
      for (int l2 = 0; l2 < 3; ++l2)
        for (int l3 = 0; l3 < 3; ++l3) {
          ch.wblk[l2][l3] = -1;
          if (l2 > lmax_sh || d.M[l3] == 0 || ((l1 + l2 + l3) & 1) || l3 < std::abs(l1 - l2) || l3 > l1 + l2) continue;
          int orig = -1;  // original row offset of path (l1, l2) in class cls3[l3]
          for (auto& p : paths_by_class[cls3[l3]])
            if (p.l1 == l1 && p.l2 == l2) orig = p.wrow;
          if (orig < 0) continue;
          ch.wblk[l2][l3] = next_blk[l3];
          F->h_pack.push_back({l3, orig + chan_seen[l1], ch.count, next_blk[l3]});
          next_blk[l3] += (ch.count + 15) >> 4;
        }
      chan_seen[l1] += ch.count;
      for (int m = 0; m < 2; ++m) {
        const int epu = m ? 8 : 4, cwp = ((ch.count + 15) & ~15) * (2 * l1 + 1);
        ch.S[m] = (cwp / epu) | 1;
        ch.rows_per[m] = 64 / ch.S[m];
        ch.inv[m] = (65536 + ch.S[m] - 1) / ch.S[m];
        for (int ln = 0; ln < 64; ++ln)
          if (((ln * ch.inv[m]) >> 16) != ln / ch.S[m]) return E3_OK;
      }
      F->h_chunks.push_back(ch);
    }
  }
  int bfo = 0;
  for (int l3 = 0; l3 < 3; ++l3) {
    d.bfoff[l3] = bfo;
    bfo += next_blk[l3] * 16 * d.Mpad[l3];
  }
  d.bftotal = bfo;
  d.nchunks = (int)F->h_chunks.size();
  if (d.nchunks == 0 || d.bftotal == 0) return E3_OK;
  F->usable = r16_supported(F);
  return E3_OK;
}

int fast_upload(TpFast* F) {
  if (!F->usable || F->d_dev) return E3_OK;
  E3_HIP_CHECK(hipMalloc((void**)&F->d_chunks, F->h_chunks.size() * sizeof(FChunk)));
  E3_HIP_CHECK(hipMemcpy(F->d_chunks, F->h_chunks.data(), F->h_chunks.size() * sizeof(FChunk), hipMemcpyHostToDevice));
  E3_HIP_CHECK(hipMalloc((void**)&F->d_pack, std::max<size_t>(F->h_pack.size(), 1) * sizeof(FPack)));
  if (!F->h_pack.empty())
    E3_HIP_CHECK(hipMemcpy(F->d_pack, F->h_pack.data(), F->h_pack.size() * sizeof(FPack), hipMemcpyHostToDevice));
  E3_HIP_CHECK(hipMalloc((void**)&F->d_dev, sizeof(FDev)));
  E3_HIP_CHECK(hipMemcpy(F->d_dev, &F->dev, sizeof(FDev), hipMemcpyHostToDevice));
  return E3_OK;
}

void fast_free(TpFast* F) {
  if (F->d_chunks) (void)hipFree(F->d_chunks);
  if (F->d_pack) (void)hipFree(F->d_pack);
  if (F->d_dev) (void)hipFree(F->d_dev);
  F->d_chunks = nullptr; F->d_pack = nullptr; F->d_dev = nullptr;
}

int64_t fast_packed_bytes(const TpFast* F) {
  if (!F->usable) return 0;
  const int64_t words = ((F->dev.Dout + 3) & ~3) + 4 + F->dev.bftotal;  // normcol | header | Whi + Wlo (2 x 2 bytes each)
  return (words * 4 + 255) / 256 * 256;
}

int fast_pack(const TpFast* F, const void* const w[6], const void* const n[6], int dtype, void* packed,
              const int32_t* ocol_tab, hipStream_t s) {
  if (!F->usable) return E3_OK;
  E3_HIP_CHECK(hipMemsetAsync(packed, 0, (size_t)fast_packed_bytes(F), s));
  const int npk = (int)F->h_pack.size();
  const dim3 grid(std::max(1, std::min(npk, 256)));
  const FDev& d = F->dev;
  uint32_t* hdr = reinterpret_cast<uint32_t*>((float*)packed + ((d.Dout + 3) & ~3));
  // element counts of the class matrices: rows = sum over the packed pieces is not the full matrix when a class has
  // unreachable rows, so take them from the pack list's extent
  int64_t rows[3] = {0, 0, 0};
  for (auto& q : F->h_pack) rows[q.l3] = std::max<int64_t>(rows[q.l3], q.orig_row + q.count);
  if (dtype == E3_BF16) {
    hipLaunchKernelGGL(fast_pack_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)w[0], (const bf16*)w[3],
                       (const bf16*)w[4], (const bf16*)(n ? n[0] : nullptr), (const bf16*)(n ? n[3] : nullptr),
                       (const bf16*)(n ? n[4] : nullptr), (float*)packed, d, F->d_pack, npk, ocol_tab);
  } else {
    hipLaunchKernelGGL(fast_absmax_kernel<float>, dim3(64), dim3(256), 0, s, (const float*)w[0], rows[0] * d.M[0],
                       (const float*)w[3], rows[1] * d.M[1], (const float*)w[4], rows[2] * d.M[2], hdr);
    hipLaunchKernelGGL(fast_pack_kernel<float>, grid, dim3(256), 0, s, (const float*)w[0], (const float*)w[3],
                       (const float*)w[4], (const float*)(n ? n[0] : nullptr), (const float*)(n ? n[3] : nullptr),
                       (const float*)(n ? n[4] : nullptr), (float*)packed, d, F->d_pack, npk, ocol_tab);
  }
  E3_HIP_CHECK(hipGetLastError());
  return E3_OK;
}

static thread_local const char* g_last_kernel = "";
void fast_note_kernel(const char* name) { g_last_kernel = name; }
const char* fast_last_kernel() { return g_last_kernel; }

int fast_forward(const TpFast* F, const e3_tp_segment* segs, int nseg, int D1, const void* in2, int64_t ld2,
                 const void* packed, void* out, int64_t ldo, int64_t B, int gate, int dtype, const int32_t* ocol_tab,
                 const float* in_scale, hipStream_t s, const int32_t* scatter, const void* residual, int64_t ldr,
                 uint32_t* amax) {
  if (!F->usable) return E3_ERR_UNSUPPORTED;
  const bool io16 = dtype == E3_BF16;
  const FDev& d = F->dev;
  if (nseg < 1 || nseg > 4) return E3_ERR_INVALID_ARG;
  SegArgs sa;
  int col = 0;
  if (ldo >= ((int64_t)1 << 24)) return E3_ERR_UNSUPPORTED;  // the store loop uses 24-bit multiplies for row * ldo
  for (int i = 0; i < 4; ++i) { sa.base[i] = nullptr; sa.ld[i] = 0; sa.index[i] = nullptr; }
  for (int i = 0; i < nseg; ++i) {
    if (!segs[i].base || segs[i].ncols <= 0 || segs[i].ld < segs[i].ncols) return E3_ERR_INVALID_ARG;
    if (segs[i].ld >= (int64_t)1 << 29) return E3_ERR_UNSUPPORTED;  // the kernel forms 32-bit row strides in bytes
    sa.base[i] = segs[i].base;
    sa.ld[i] = segs[i].ld;
    sa.index[i] = segs[i].row_index;
    sa.col0[i] = col;
    col += segs[i].ncols;
  }
  for (int i = nseg; i < 5; ++i) sa.col0[i] = col;
  sa.nseg = nseg;
  sa.scatter = scatter;
  sa.residual = residual;
  sa.ldr = ldr;
  sa.amax = amax;
  if (residual && ldr >= (int64_t)1 << 29) return E3_ERR_UNSUPPORTED;
  if (col != D1) return E3_ERR_INVALID_ARG;
  for (auto& ch : F->h_chunks) {  // a chunk must not straddle two segments
    int cw = ch.count * (2 * ch.l1 + 1), sidx = 0;
    while (sidx + 1 < nseg && ch.col >= sa.col0[sidx + 1]) ++sidx;
    if (ch.col + cw > sa.col0[sidx + 1]) return E3_ERR_INVALID_ARG;
  }
  if (gate && !fast_gate_shape_ok(F)) return E3_ERR_UNSUPPORTED;
  const int r = fast_forward_r16(F, &sa, in2, ld2, packed, out, ldo, B, gate, io16 ? 1 : 0, ocol_tab, in_scale, s);
  if (r == 1) return E3_OK;
  return r < 0 ? -r : E3_ERR_UNSUPPORTED;
}

}  // namespace e3
