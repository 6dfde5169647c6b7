// fp32 MFMA kernel for the general SH tensor product (l <= 2, natural-parity classes 0e / 1o / 2e),
// with optional fused row gather (in1 = concatenation of row-indexed segments) and fused gate.
//
// Structure ("K-outer, all outputs resident"): one wave owns a tile of 32 rows and keeps EVERY output
// accumulator of those rows in registers (NT0 scalar tiles + 3*NT1 + 5*NT2 accumulators of 16 VGPRs).  It
// then walks the input irreps blocks ("chunks" of <= 32 channels): each chunk is staged once into a small
// LDS buffer by LDS-DMA (gathered per row when a segment carries a row index) and contracted on the
// matrix core (v_mfma_f32_32x32x2_f32) into every output class it couples to:
//     A operand = packed weights W'[k][32 t + (lane&31)]           (LDS when they fit, else L2)
//     B operand = per-row feature  sum_m1 z[m1][m3] x[k][m1]        z = sum_m2 C[m1][m2][m3] Y[m2]  (per lane)
// or, when 2 l1 + 1 < 2 l3 + 1, the raw x[k][m1] into temporaries that are folded with z afterwards.
// So every input element is read from HBM/L2 once per tile and every output is written once; the gather,
// the concat and the gate of the SEGNN message function never materialise in HBM.
#include "e3_common.h"
#include "cg_tables.h"
#include "e3_tp_internal.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace e3 {

#include "e3_tp_mfma_core.h"

#include "e3_tp_mfma_kernel.h"

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ void fast_pack_kernel(const T* w0, const T* w1, const T* w2, const T* n0, const T* n1, const T* n2,
                                 float* packed, FDev d, const FPack* pk, int npk, const int32_t* ocol_tab) {
  const T* w[3] = {w0, w1, w2};
  const T* nr[3] = {n0, n1, n2};
  for (int r = blockIdx.x; r < npk; r += gridDim.x) {
    const FPack q = pk[r];
    const int M = d.M[q.l3], Mpad = d.Mpad[q.l3];
    uint16_t* whi = reinterpret_cast<uint16_t*>(packed + d.wtotal + ((d.Dout + 3) & ~3));
    uint16_t* wlo = whi + d.bftotal;
    for (int i = threadIdx.x; i < q.count * M; i += blockDim.x) {
      int k = i / M, mm = i - k * M;
      const float v = to_acc(w[q.l3][(int64_t)(q.orig_row + k) * M + mm]);
      packed[d.woff[q.l3] + (size_t)(q.wrow + k) * Mpad + mm] = v;
      // bf16 split, layout [16-row block][k half][channel][8]
      const __bf16 h = (__bf16)v;
      const __bf16 l = (__bf16)(v - (float)h);
      const size_t e = (size_t)d.bfoff[q.l3] + ((size_t)(2 * (q.wblk + (k >> 4)) + ((k >> 3) & 1)) * Mpad + mm) * 8 + (k & 7);
      whi[e] = __builtin_bit_cast(uint16_t, h);
      wlo[e] = __builtin_bit_cast(uint16_t, l);
    }
  }
  if (blockIdx.x == 0)
    for (int l3 = 0; l3 < 3; ++l3) {
      const int width = 2 * l3 + 1;
      for (int i = threadIdx.x; i < d.M[l3] * width; i += blockDim.x) {
        int mm = i / width, comp = i - mm * width;
        packed[d.wtotal + ocol_tab[d.ooff[l3] + mm] + comp] = nr[l3] ? to_acc(nr[l3][i]) : 1.0f;
      }
    }
}

// Instantiated signatures = the tensor products of the SEGNN forward (H <= 32 per block):
//   l_max 1: embed (0,1 -> hid), msg1 (0,1,0,1,0 -> gated), msg2 (0,1 -> gated), upd1 (0,1,0,1 -> gated),
//            upd2 (0,1 -> hid), readout (0,1 -> 1o);   l_max 2: the same with (0,1,2) blocks.
std::vector<FastKernelEntry> fast_kernels_part1();  // e3_tp_mfma_p1.hip: l_max 2 message TP #1
std::vector<FastKernelEntry> fast_kernels_part2();  // e3_tp_mfma_p2.hip: l_max 2 message TP #2, update TP #1
static const std::vector<FastKernelEntry>& fast_kernels() {
  static const std::vector<FastKernelEntry> k = [] {
    std::vector<FastKernelEntry> v = {
        E3_FAST(1, 1, 1, 0, 0, 1),       E3_FAST(1, 2, 1, 0, 0, 1, 0, 1, 0), E3_FAST(1, 2, 1, 0, 0, 1),
        E3_FAST(1, 2, 1, 0, 0, 1, 0, 1), E3_FAST(1, 0, 1, 0, 0, 1),
        E3_FAST(2, 1, 1, 1, 0, 1),       E3_FAST(2, 1, 1, 1, 0, 1, 2),       E3_FAST(2, 0, 1, 0, 0, 1, 2),
    };
    for (auto& e : fast_kernels_part1()) v.push_back(e);
    for (auto& e : fast_kernels_part2()) v.push_back(e);
    return v;
  }();
  return k;
}

static const FastKernelEntry* find_fast(int lsh, int a, int b, int c, const std::vector<int>& l1s) {
  for (auto& e : fast_kernels())
    if (e.lsh == lsh && e.nt0 == a && e.nt1 == b && e.nt2 == c && e.l1s == l1s) return &e;
  return nullptr;
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
  d.bf = getenv("E3_TP_EXACT") ? 0 : 1;
  d.dbg = getenv("E3_TP_DBG") ? atoi(getenv("E3_TP_DBG")) : 0;
  d.prof = nullptr;  // timing-only diagnostics, results are wrong when set  // default: bf16-split operands (fp32-grade accuracy)
  int next_row[3] = {0, 0, 0};
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
          ch.wrow[l2][l3] = -1;
          ch.wblk[l2][l3] = -1;
          if (l2 > lmax_sh || d.M[l3] == 0 || ((l1 + l2 + l3) & 1) || l3 < std::abs(l1 - l2) || l3 > l1 + l2) continue;
          // original row offset of path (l1,l2) in class cls3[l3]
          int orig = -1;
          for (auto& p : paths_by_class[cls3[l3]])
            if (p.l1 == l1 && p.l2 == l2) orig = p.wrow;
          if (orig < 0) continue;
          ch.wrow[l2][l3] = next_row[l3];
          ch.wblk[l2][l3] = next_blk[l3];
          F->h_pack.push_back({l3, orig + chan_seen[l1] + 0, ch.count, next_row[l3], next_blk[l3]});
          next_row[l3] += (ch.count + 1) & ~1;
          next_blk[l3] += (ch.count + 15) >> 4;
        }
      chan_seen[l1] += ch.count;
      for (int m = 0; m < 2; ++m) {
        const int epu = m ? 8 : 4, cwp = ((ch.count + 15) & ~15) * (2 * l1 + 1);
        ch.S[m] = (cwp / epu) | 1;
        ch.rows_per[m] = 64 / ch.S[m];
        ch.inv[m] = (65536 + ch.S[m] - 1) / ch.S[m];
        for (int ln = 0; ln < 64; ++ln)
          if (((ln * ch.inv[m]) >> 16) != ln / ch.S[m]) return E3_ERR_UNSUPPORTED;
      }
      F->h_chunks.push_back(ch);
    }
  }
  int woff = 0;
  for (int l3 = 0; l3 < 3; ++l3) {
    d.woff[l3] = woff;
    woff += next_row[l3] * d.Mpad[l3];
  }
  d.wtotal = woff;
  int bfo = 0;
  for (int l3 = 0; l3 < 3; ++l3) {
    d.bfoff[l3] = bfo;
    bfo += next_blk[l3] * 16 * d.Mpad[l3];
  }
  d.bftotal = bfo;
  d.nchunks = (int)F->h_chunks.size();
  if (d.nchunks == 0 || d.wtotal == 0) return E3_OK;
  std::vector<int> l1s;
  for (auto& c : F->h_chunks) l1s.push_back(c.l1);
  if (!find_fast(lmax_sh, d.NT[0], d.NT[1], d.NT[2], l1s)) return E3_OK;
  // LDS plan: [weights?][normcol][ocol][nwaves x (nbuf chunk buffers + Y tile)], once per storage class
  size_t tables = (size_t)((Dout + 4 + 15) & ~15) * 4 + (size_t)((ntab + 15) & ~15) * 4;
  auto lds_plan = [&](FDev& dd, size_t wbytes, int chunk_dwords, size_t* lds_bytes) -> bool {
    auto per_wave = [&](int nbuf) { return (size_t)(nbuf * chunk_dwords + 320) * 4; };
    auto fit = [&](size_t fixed, int nbuf) -> int {
      return fixed + per_wave(nbuf) <= (size_t)kFastLds ? (int)std::min<size_t>(((size_t)kFastLds - fixed) / per_wave(nbuf), 4) : 0;
    };
    // One chunk buffer per wave and as many waves as fit: measured, a second buffer never paid (the wave that
    // issues the copies is the one that consumes them; 3 double-buffered waves lost to 4 single-buffered ones on
    // every kernel of the SEGNN forward).  E3_TP_NBUF=2 keeps the double-buffered variant reachable.
    const int w1 = fit(tables + wbytes, 1);
    if (w1 >= 3) { dd.w_in_lds = 1; dd.nbuf = 1; dd.nwaves = w1; }
    else { dd.w_in_lds = 0; dd.nbuf = 1; dd.nwaves = fit(tables, 1); }
    if (const char* e = getenv("E3_TP_NBUF")) { int v = atoi(e); if (v == 1 || (v == 2 && dd.w_in_lds && fit(tables + wbytes, 2) >= 1)) { dd.nbuf = v; dd.nwaves = fit(tables + (dd.w_in_lds ? wbytes : 0), v); } }
    *lds_bytes = tables + (dd.w_in_lds ? wbytes : 0) + (size_t)dd.nwaves * per_wave(dd.nbuf);
    return dd.nwaves >= 1 && *lds_bytes <= (size_t)kFastLds;
  };
  F->dev16 = d;
  const bool ok32 = lds_plan(d, d.bf ? (size_t)d.bftotal * 4 : (size_t)d.wtotal * 4, kChunkFloats, &F->lds_bytes);
  const bool ok16 = lds_plan(F->dev16, (size_t)d.bftotal * 2, kChunk16, &F->lds_bytes16);
  if (!ok32 || !ok16) return E3_OK;
  F->usable = true;
  return E3_OK;
}

int fast_upload(TpFast* F) {
  if (!F->usable || F->d_dev) return E3_OK;
  E3_HIP_CHECK(hipMalloc((void**)&F->d_chunks, F->h_chunks.size() * sizeof(FChunk)));
  E3_HIP_CHECK(hipMemcpy(F->d_chunks, F->h_chunks.data(), F->h_chunks.size() * sizeof(FChunk), hipMemcpyHostToDevice));
  E3_HIP_CHECK(hipMalloc((void**)&F->d_pack, std::max<size_t>(F->h_pack.size(), 1) * sizeof(FPack)));
  if (!F->h_pack.empty())
    E3_HIP_CHECK(hipMemcpy(F->d_pack, F->h_pack.data(), F->h_pack.size() * sizeof(FPack), hipMemcpyHostToDevice));
  if (F->dev.dbg & 8) {
    unsigned long long* pr = nullptr;
    E3_HIP_CHECK(hipMalloc((void**)&pr, 8 * sizeof(unsigned long long)));
    E3_HIP_CHECK(hipMemset(pr, 0, 8 * sizeof(unsigned long long)));
    F->dev.prof = pr;
    F->dev16.prof = pr;
  }
  E3_HIP_CHECK(hipMalloc((void**)&F->d_dev, sizeof(FDev)));
  E3_HIP_CHECK(hipMemcpy(F->d_dev, &F->dev, sizeof(FDev), hipMemcpyHostToDevice));
  E3_HIP_CHECK(hipMalloc((void**)&F->d_dev16, sizeof(FDev)));
  E3_HIP_CHECK(hipMemcpy(F->d_dev16, &F->dev16, sizeof(FDev), hipMemcpyHostToDevice));
  for (auto& e : fast_kernels())
    for (int f = 0; f < 3; ++f)
      for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b)
          E3_HIP_CHECK(hipFuncSetAttribute(e.fn[f][a][b], hipFuncAttributeMaxDynamicSharedMemorySize, kFastLds));
  return E3_OK;
}

void fast_free(TpFast* F) {
  if (F->d_chunks) (void)hipFree(F->d_chunks);
  if (F->d_pack) (void)hipFree(F->d_pack);
  if (F->d_dev) (void)hipFree(F->d_dev);
  if (F->d_dev16) (void)hipFree(F->d_dev16);
}

int64_t fast_packed_bytes(const TpFast* F) {
  if (!F->usable) return 0;
  int64_t words = (int64_t)F->dev.wtotal + ((F->dev.Dout + 3) & ~3) + F->dev.bftotal;  // fp32 W' | normcol | Whi+Wlo
  return (words * 4 + 255) / 256 * 256;
}

int fast_pack(const TpFast* F, const void* const w[6], const void* const n[6], int dtype, void* packed,
              const int32_t* ocol_tab, hipStream_t s) {
  if (!F->usable) return E3_OK;
  E3_HIP_CHECK(hipMemsetAsync(packed, 0, (size_t)fast_packed_bytes(F), s));
  int npk = (int)F->h_pack.size();
  dim3 grid(std::max(1, std::min(npk, 256)));
  if (dtype == E3_BF16)
    hipLaunchKernelGGL(fast_pack_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)w[0], (const bf16*)w[3],
                       (const bf16*)w[4], (const bf16*)(n ? n[0] : nullptr), (const bf16*)(n ? n[3] : nullptr),
                       (const bf16*)(n ? n[4] : nullptr), (float*)packed, F->dev, F->d_pack, npk, ocol_tab);
  else
    hipLaunchKernelGGL(fast_pack_kernel<float>, grid, dim3(256), 0, s, (const float*)w[0], (const float*)w[3],
                       (const float*)w[4], (const float*)(n ? n[0] : nullptr), (const float*)(n ? n[3] : nullptr),
                       (const float*)(n ? n[4] : nullptr), (float*)packed, F->dev, F->d_pack, npk, ocol_tab);
  E3_HIP_CHECK(hipGetLastError());
  return E3_OK;
}

static thread_local const char* g_last_kernel = "";
void fast_note_kernel(const char* name) { g_last_kernel = name; }
const char* fast_last_kernel() { return g_last_kernel; }

int fast_forward(const TpFast* F, const e3_tp_segment* segs, int nseg, int D1, const void* in2, int64_t ld2,
                 const void* packed, void* out, int64_t ldo, int64_t B, int gate, int dtype, const int32_t* ocol_tab,
                 hipStream_t s, const int32_t* scatter) {
  if (!F->usable) return E3_ERR_UNSUPPORTED;
  const bool io16 = dtype == E3_BF16;
  const FDev& d = io16 ? F->dev16 : F->dev;
  const int mode = io16 ? 2 : (d.bf ? 1 : 0);
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
  if (col != D1) return E3_ERR_INVALID_ARG;
  for (auto& ch : F->h_chunks) {  // a chunk must not straddle two segments
    int cw = ch.count * (2 * ch.l1 + 1), sidx = 0;
    while (sidx + 1 < nseg && ch.col >= sa.col0[sidx + 1]) ++sidx;
    if (ch.col + cw > sa.col0[sidx + 1]) return E3_ERR_INVALID_ARG;
  }
  if (gate) {
    const int nb = (d.NT[1] > 0) + (d.NT[2] > 0);
    if (!F->gate_layout || d.NT[0] != 1 + nb || d.M[0] != 32 * (1 + nb) || (d.NT[1] && d.M[1] != 32) || (d.NT[2] && d.M[2] != 32) ||
        d.NT[1] > 1 || d.NT[2] > 1)
      return E3_ERR_UNSUPPORTED;
  }
  {
    int r = fast_forward_r16(F, &sa, in2, ld2, packed, out, ldo, B, gate, mode, ocol_tab, s);
    if (r == 0) r = fast_forward_ab(F, &sa, in2, ld2, packed, out, ldo, B, gate, mode, ocol_tab, s);
    if (r == 1) return E3_OK;
    if (r < 0) return -r;
    if (scatter) return E3_ERR_UNSUPPORTED;  // only the two-wave kernel has the fused segment-sum
  }
  std::vector<int> l1s;
  for (auto& c : F->h_chunks) l1s.push_back(c.l1);
  const FastKernelEntry* e = find_fast(d.lsh, d.NT[0], d.NT[1], d.NT[2], l1s);
  if (!e) return E3_ERR_UNSUPPORTED;
  int64_t ntiles = (B + 31) / 32;
  int grid = (int)std::min<int64_t>((ntiles + d.nwaves - 1) / d.nwaves, 256);
  const float* in2f = (const float*)in2;
  const float* pk = (const float*)packed;
  void* outf = out;
  const FDev* dd = io16 ? F->d_dev16 : F->d_dev;
  const size_t lds_bytes = io16 ? F->lds_bytes16 : F->lds_bytes;
  const FChunk* dc = F->d_chunks;
  void* args[] = {&sa, &in2f, &ld2, &pk, &outf, &ldo, &B, &dd, &dc, &ocol_tab};
  E3_HIP_CHECK(hipLaunchKernel(e->fn[mode][d.w_in_lds ? 1 : 0][gate ? 1 : 0], dim3(grid), dim3(64 * d.nwaves), args,
                               lds_bytes, s));
  fast_note_kernel("e3::tp_fwd_mfma_kernel");
  return E3_OK;
}

}  // namespace e3
