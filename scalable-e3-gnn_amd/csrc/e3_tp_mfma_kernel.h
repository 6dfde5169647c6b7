// The one-wave-per-tile MFMA tensor-product kernel template and its instantiation table entry type.  Internal header,
// included inside namespace e3 after e3_tp_mfma_core.h.  The instantiations are spread over three translation units
// (e3_tp_mfma.hip, e3_tp_mfma_p1.hip, e3_tp_mfma_p2.hip) so that `make -j` builds them in parallel: the kernels are
// fully unrolled per tensor product and take minutes to compile.
#pragma once

// LSH = SH degree of in2; NT* = number of 32-channel output tiles per degree; L1S... = degrees of the input
// chunks in order (compile-time so that the chunk walk is straight-line code: no control-flow merges of the
// 16-register accumulator tuples, which otherwise explode the register allocation).
// MODE: 0 = exact fp32 MFMA, 1 = fp32 in/out with bf16x3-split operands, 2 = bf16 in/out, single bf16 MFMA.
template <int LSH, int NT0, int NT1, int NT2, bool WLDS, bool GATE, int MODE, int... L1S>
__global__ __launch_bounds__(256) void tp_fwd_mfma_kernel(SegArgs segs, const float* __restrict__ in2, int64_t ld2,
                                                          const float* __restrict__ packed, void* __restrict__ outv,
                                                          int64_t ldo, int64_t B, const FDev* __restrict__ dp,
                                                          const FChunk* __restrict__ chunks,
                                                          const int32_t* __restrict__ ocol_tab) {
  constexpr bool BF = MODE >= 1;     // operands go through the bf16 matrix pipe
  constexpr bool IO16 = MODE == 2;   // bf16 storage
  constexpr int CHUNK = IO16 ? kChunk16 : kChunkFloats;  // dwords per chunk buffer
  extern __shared__ __align__(16) unsigned char smem_raw[];
  float* lds = reinterpret_cast<float*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, half = lane >> 5;
  const int Dout = dp->Dout, Dy = dp->Dy, wtotal = dp->wtotal, nwaves = dp->nwaves, nchunks = dp->nchunks,
            nbuf = dp->nbuf;
  int cM[3], cMpad[3], cWoff[3], cOoff[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) { cM[c] = dp->M[c]; cMpad[c] = dp->Mpad[c]; cWoff[c] = dp->woff[c]; cOoff[c] = dp->ooff[c]; }
  const int ntab = dp->ntab;
  const int bftotal = dp->bftotal;
  const int dbg = dp->dbg;
  int cBfoff[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) cBfoff[c] = dp->bfoff[c];
  // weights section of `packed`: [fp32 W' (wtotal) | normcol (Dout) | Whi (bftotal u16) | Wlo (bftotal u16)]
  const int wwords = IO16 ? (bftotal >> 1) : (BF ? bftotal : wtotal);  // 32-bit words of the weight image kept in LDS

  float* wl = lds;
  float* nrm = lds + (WLDS ? wwords : 0);
  int* ocl = reinterpret_cast<int*>(nrm + ((Dout + 4 + 15) & ~15));  // nrm[Dout .. Dout+3] = 1 ("no norm" entry)
  float* wbase_lds = reinterpret_cast<float*>(ocl + ((ntab + 15) & ~15));
  const int per_wave = nbuf * CHUNK + 320;
  float* cbuf = wbase_lds + (size_t)wave * per_wave;
  float* ybuf = cbuf + nbuf * CHUNK;
  const float* wglob = BF ? packed + wtotal + ((Dout + 3) & ~3) : packed;
  if (WLDS)
    for (int i = tid; i < wwords; i += blockDim.x) wl[i] = wglob[i];
  for (int i = tid; i < Dout; i += blockDim.x) nrm[i] = packed[wtotal + i];
  if (tid < 4) nrm[Dout + tid] = 1.f;
  for (int i = tid; i < ntab; i += blockDim.x) ocl[i] = ocol_tab[i];
  __syncthreads();
  const float* wsrc = WLDS ? wl : wglob;
  const uint4* whi_base = reinterpret_cast<const uint4*>(wsrc);                      // Whi, 8 bf16 per uint4
  const uint4* wlo_base = reinterpret_cast<const uint4*>(wsrc + (bftotal >> 1));     // Wlo follows Whi

  const int64_t ntiles = (B + 31) / 32;
  const int64_t tstride = (int64_t)gridDim.x * nwaves;
  // diagnostic phase timers (E3_TP_DBG & 8): 0 prologue issue, 1 waits for staged data, 2 stage issue, 3 runs, 4 epilogue: gate + transpose into LDS, 5 epilogue: norm + stores
  unsigned long long* const prof = dp->prof;
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
  auto tick = [&](int phase) {
    if (prof) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      tacc[phase] += now - tlast;
      tlast = now;
    }
  };
  if (prof) tlast = __builtin_amdgcn_s_memtime();

  // row ids of the gathered segments, this lane's row (lane & 31), fetched one tile ahead so that no stage call
  // waits on an index load
  // (named scalars, not arrays: an indexed array lands in scratch memory)
  int mc0 = 0, mc1 = 0, mc2 = 0, mc3 = 0, mn0 = 0, mn1 = 0, mn2 = 0, mn3 = 0;
  auto fetch_ids = [&](int64_t t) {
    const int64_t r = t * 32 + j;
    if (r < B) {
      if (segs.nseg > 0 && segs.index[0]) mn0 = segs.index[0][r];
      if (segs.nseg > 1 && segs.index[1]) mn1 = segs.index[1][r];
      if (segs.nseg > 2 && segs.index[2]) mn2 = segs.index[2][r];
      if (segs.nseg > 3 && segs.index[3]) mn3 = segs.index[3][r];
    }
  };
  if ((int64_t)blockIdx.x * nwaves + wave < ntiles) fetch_ids((int64_t)blockIdx.x * nwaves + wave);
  const int inv_dy = (65536 + Dy - 1) / Dy;  // e / Dy == (e * inv_dy) >> 16 for e < 32 * Dy (Dy <= 9)

  for (int64_t tile = (int64_t)blockIdx.x * nwaves + wave; tile < ntiles; tile += tstride) {
    const int64_t row0 = tile * 32;
    const int nrows = (int)((B - row0) < 32 ? (B - row0) : 32);
    mc0 = mn0; mc1 = mn1; mc2 = mn2; mc3 = mn3;
    if (tile + tstride < ntiles) fetch_ids(tile + tstride);

    // stage one chunk of 32 rows (LDS-DMA; per-row gather through the segment's row index).  Row stride (dwords) is
    // odd => the lane=row reads are bank-conflict free.  BF modes zero-pad the chunk to a multiple of 16 channels.
    auto stage = [&](int ci, float* dst) {
      if ((dbg & 2) && tile != (int64_t)blockIdx.x * nwaves + wave) return;
      const FChunk ch = chunks[ci];
      // segment of this chunk; constant subscripts only -- a runtime subscript into the by-value argument struct
      // forces the whole struct into scratch memory and every use becomes a scratch load
      const int s = (segs.nseg > 1 && ch.col >= segs.col0[1]) + (segs.nseg > 2 && ch.col >= segs.col0[2]) +
                    (segs.nseg > 3 && ch.col >= segs.col0[3]);
      auto pick = [&](auto v0, auto v1, auto v2, auto v3) {
        auto v = v0;
        v = s == 1 ? v1 : v;
        v = s == 2 ? v2 : v;
        v = s == 3 ? v3 : v;
        return v;
      };
      const int64_t ld = pick(segs.ld[0], segs.ld[1], segs.ld[2], segs.ld[3]);
      const int32_t* idx = pick(segs.index[0], segs.index[1], segs.index[2], segs.index[3]);
      const void* segbase = pick(segs.base[0], segs.base[1], segs.base[2], segs.base[3]);
      const int segcol = ch.col - pick(segs.col0[0], segs.col0[1], segs.col0[2], segs.col0[3]);
      const int cw = ch.count * (2 * ch.l1 + 1);  // elements per row
      // this lane's row id (lane & 31): prefetched a tile ahead for gathered segments (`mcur`)
      const int mg = pick(mc0, mc1, mc2, mc3);
      const int mr = idx ? mg : (int)row0 + j;  // row ids fit int32 (N, E < 2^31)
      if constexpr (BF) {
        // Layout: rows of S 16-byte units, S odd (=> the lane=row ds_read_b128 of the operand loads are conflict
        // free), the chunk zero-padded to a multiple of 16 channels.  Staging: 16-byte LDS-DMA, lane = (row, unit),
        // several rows per instruction; the per-lane source address does the gather.  (4-byte DMAs run at a quarter
        // of the 16-byte rate and made staging ~45 % of the kernel.)  The wave is alone on its SIMD, so this code is
        // priced in issued instructions: S / rows_per / the lane split come precomputed with the chunk.
        constexpr int ESZ = IO16 ? 2 : 4, EPU = 16 / ESZ, MI = IO16 ? 1 : 0;
        const int cwp = ((ch.count + 15) & ~15) * (2 * ch.l1 + 1);
        const int upr = cw / EPU, S = ch.S[MI];
        const char* base = reinterpret_cast<const char*>(segbase);
        const bool wide = (cw % EPU == 0) && (segcol % EPU == 0) && (ld % EPU == 0) &&
                          ((reinterpret_cast<uintptr_t>(base) & 15) == 0);
        tick(6);
        if (dbg & 64) {  // diagnostic: no copies at all
        } else if (wide) {
          const int rows_per = ch.rows_per[MI];
          const int rl = (lane * ch.inv[MI]) >> 16, u = lane - rl * S;
          const bool lane_ok = rl < rows_per && u < upr;
          const char* lsrc = base + (int64_t)segcol * ESZ + u * 16;
          const uint32_t ldb = (uint32_t)(ld * ESZ);  // row stride in bytes (< 2^32, checked by the host)
          for (int r0 = 0; r0 < 32; r0 += 4 * rows_per) {  // 4 DMA instructions per batch: index fetches first
            int ridx[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) ridx[k] = __shfl(mr, (r0 + k * rows_per + rl) & 31);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int rbase = r0 + k * rows_per;
              if (rbase < 32 && lane_ok && rbase + rl < nrows && !(dbg & 16))
                __builtin_amdgcn_global_load_lds((glb_void_t*)(lsrc + (uint64_t)(uint32_t)ridx[k] * ldb),
                                                 (lds_void_t*)(dst + rbase * S * 4), 16, 0, 0);
            }
          }
          // zero what the copies did not write: padding channels, and every column of rows beyond the batch
          if (cwp > cw || nrows < 32) {
            uint16_t* d16 = reinterpret_cast<uint16_t*>(dst);
            for (int r = 0; r < 32; ++r) {
              const int e0 = (r < nrows) ? cw : 0;
              for (int e = e0 + lane; e < cwp; e += 64) {
                if (IO16) d16[r * S * 8 + e] = 0;
                else dst[r * S * 4 + e] = 0.f;
              }
            }
          }
        } else {
          // narrow or unaligned chunks (the distance scalar, single channels): through registers, lane = (row, parity
          // of the element index), all rows in flight at once; padding and tail rows are written as zeros
          const bool rok = j < nrows;
          if (IO16) {
            uint16_t* drow = reinterpret_cast<uint16_t*>(dst) + j * S * 8;
            const uint16_t* srow = reinterpret_cast<const uint16_t*>(base) + (int64_t)mr * ld + segcol;
            for (int e = half; e < cwp; e += 2) drow[e] = (rok && e < cw) ? srow[e] : (uint16_t)0;
          } else {
            float* drow = dst + j * S * 4;
            const float* srow = reinterpret_cast<const float*>(base) + (int64_t)mr * ld + segcol;
            for (int e = half; e < cwp; e += 2) drow[e] = (rok && e < cw) ? srow[e] : 0.f;
          }
        }
        return;
      } else {
        const int stride = cw | 1;
        const float* base = reinterpret_cast<const float*>(segbase);
        if (cw == 1) {  // one column (the distance scalar): lane r fetches row r
          if (lane < nrows)
            __builtin_amdgcn_global_load_lds((glb_void_t*)(base + (int64_t)mr * ld + segcol), (lds_void_t*)dst, 4, 0, 0);
          else if (lane < 32)
            dst[lane] = 0.f;
          return;
        }
        const int full = cw & ~63;
        float* drow = dst;
        for (int r = 0; r < 32; ++r) {
          if (r < nrows) {
            const int rr = __builtin_amdgcn_readlane(mr, r);
            const float* srow = base + (int64_t)rr * ld + segcol + lane;
            for (int dc = 0; dc < full; dc += 64)
              __builtin_amdgcn_global_load_lds((glb_void_t*)(srow + dc), (lds_void_t*)(drow + dc), 4, 0, 0);
            if (full + lane < cw)
              __builtin_amdgcn_global_load_lds((glb_void_t*)(srow + full), (lds_void_t*)(drow + full), 4, 0, 0);
          } else {
            for (int dc = lane; dc < cw; dc += 64) drow[dc] = 0.f;
          }
          drow += stride;
        }
      }
    };

    // Y tile [32][Dy] fp32 (lane e of piece h fetches element h*64+e of the flattened tile)
    for (int h = 0; h * 64 < 32 * Dy; ++h) {
      const int e = h * 64 + lane;
      const int yr = (e * inv_dy) >> 16, yc = e - yr * Dy;
      if (e < 32 * Dy) {
        if (yr < nrows)
          __builtin_amdgcn_global_load_lds((glb_void_t*)(in2 + (row0 + yr) * ld2 + yc), (lds_void_t*)(ybuf + h * 64), 4,
                                           0, 0);
        else
          ybuf[e] = 0.f;
      }
    }
    stage(0, cbuf);
    tick(0);

    f32x16 a0[NT0 > 0 ? NT0 : 1][1], a1[NT1 > 0 ? NT1 : 1][3], a2[NT2 > 0 ? NT2 : 1][5];
#pragma unroll
    for (int t = 0; t < (NT0 > 0 ? NT0 : 1); ++t) a0[t][0] = f32x16{0};
#pragma unroll
    for (int t = 0; t < (NT1 > 0 ? NT1 : 1); ++t)
#pragma unroll
      for (int c = 0; c < 3; ++c) a1[t][c] = f32x16{0};
#pragma unroll
    for (int t = 0; t < (NT2 > 0 ? NT2 : 1); ++t)
#pragma unroll
      for (int c = 0; c < 5; ++c) a2[t][c] = f32x16{0};

    float y[9];
    int cur = 0;
    int ci = 0;
    // A operands (weights) of the first k block of every path of the NEXT chunk are fetched while the current chunk's
    // copies are being issued: each path otherwise starts with an exposed L2 (or LDS) round trip -- ~40 paths per
    // tile, 40 % of the kernel's cycles were s_waitcnt stalls (SQ_WAIT_ANY) before this
    using Slots = PathSlots<LSH, NT0, NT1, NT2>;
    using Seq = IntSeq<L1S...>;
    constexpr int PWN = BF ? Slots::max_total() : 1;
    uint4 pwh[PWN], pwl[(BF && !IO16) ? PWN : 1];
    auto preload = [&](auto l1tag, int cidx) {
      constexpr int L1n = decltype(l1tag)::value;
      if constexpr (BF && L1n >= 0) {
        const FChunk chn = chunks[cidx];
#define E3_PRE(L2v, L3v, NTv)                                                                                   \
  if constexpr (Slots::valid(L1n, L2v, L3v)) {                                                                  \
    const size_t o = (size_t)(cBfoff[L3v] >> 3) + (size_t)(2 * chn.wblk[L2v][L3v] + half) * cMpad[L3v] + j;     \
    constexpr int s0 = Slots::slot(L1n, L2v, L3v);                                                              \
    _Pragma("unroll") for (int t = 0; t < NTv; ++t) {                                                           \
      pwh[s0 + t] = whi_base[o + 32 * t];                                                                       \
      if constexpr (!IO16) pwl[s0 + t] = wlo_base[o + 32 * t];                                                  \
    }                                                                                                           \
  }
        E3_PRE(0, 0, NT0) E3_PRE(1, 0, NT0) E3_PRE(2, 0, NT0)
        E3_PRE(0, 1, NT1) E3_PRE(1, 1, NT1) E3_PRE(2, 1, NT1)
        E3_PRE(0, 2, NT2) E3_PRE(1, 2, NT2) E3_PRE(2, 2, NT2)
#undef E3_PRE
      }
    };
    preload(std::integral_constant<int, Seq::at(0)>{}, 0);
    auto process = [&](auto itag) {
      constexpr int L1 = Seq::at(decltype(itag)::value);
      constexpr int L1N = Seq::at(decltype(itag)::value + 1);
      wait_vm0();
      wave_sync_lds();
      tick(1);
      if (ci == 0) {
#pragma unroll
        for (int q = 0; q < 9; ++q) y[q] = (q < Dy) ? ybuf[j * Dy + q] : 0.f;
      }
      const float* xt = cbuf + cur * CHUNK;
      if (nbuf == 2 && ci + 1 < nchunks) stage(ci + 1, cbuf + (cur ^ 1) * CHUNK);
      tick(2);
      const FChunk ch = chunks[ci];
      // row stride: exact mode = odd dword count; bf16-pipe modes = S 16-byte units, S odd (see `stage`)
      const int cwp = (BF ? ((ch.count + 15) & ~15) : ch.count) * (2 * L1 + 1);
      const float* xr = BF ? xt + j * (((cwp / (IO16 ? 8 : 4)) | 1) * 4) : xt + j * (cwp | 1);
#define E3_RUN(L2v, L3v, ACC, NTv)                                                                             \
  if constexpr (NTv > 0 && L2v <= LSH && CG<L1, L2v, L3v>::valid && ((L1 + L2v + L3v) % 2 == 0)) {             \
    if (dbg & 4) {                                                                                             \
    } else if constexpr (IO16) {                                                                               \
      const size_t o = (size_t)(cBfoff[L3v] >> 3) + (size_t)(2 * ch.wblk[L2v][L3v] + half) * cMpad[L3v] + j;   \
      static_assert(Slots::valid(L1, L2v, L3v), "path bookkeeping");                                           \
      run_steps_io16<L1, L2v, L3v, NTv>(reinterpret_cast<const uint32_t*>(xr), ch.count, whi_base + o,         \
                                        pwh + Slots::slot(L1, L2v, L3v), cMpad[L3v], half, y, ACC);            \
    } else if constexpr (BF) {                                                                                 \
      const size_t o = (size_t)(cBfoff[L3v] >> 3) + (size_t)(2 * ch.wblk[L2v][L3v] + half) * cMpad[L3v] + j;   \
      static_assert(Slots::valid(L1, L2v, L3v), "path bookkeeping");                                           \
      run_steps_bf<L1, L2v, L3v, NTv>(xr, ch.count, whi_base + o, wlo_base + o, pwh + Slots::slot(L1, L2v, L3v), \
                                      pwl + Slots::slot(L1, L2v, L3v), cMpad[L3v], half, y, ACC);              \
    } else {                                                                                                   \
      const float* wp = wsrc + cWoff[L3v] + (size_t)(ch.wrow[L2v][L3v] + half) * cMpad[L3v] + j;               \
      run_steps<L1, L2v, L3v, NTv>(xr, ch.count, wp, cMpad[L3v], half, y, ACC);                                \
    }                                                                                                          \
  }
      E3_RUN(0, 0, a0, NT0) E3_RUN(1, 0, a0, NT0) E3_RUN(2, 0, a0, NT0)
      E3_RUN(0, 1, a1, NT1) E3_RUN(1, 1, a1, NT1) E3_RUN(2, 1, a1, NT1)
      E3_RUN(0, 2, a2, NT2) E3_RUN(1, 2, a2, NT2) E3_RUN(2, 2, a2, NT2)
#undef E3_RUN
      tick(3);
      if (ci + 1 < nchunks && !(dbg & 32)) preload(std::integral_constant<int, L1N>{}, ci + 1);
      tick(7);
      if (nbuf == 1) {
        wave_sync_lds();
        if (ci + 1 < nchunks) stage(ci + 1, cbuf);
        tick(2);
      } else {
        cur ^= 1;
      }
      ++ci;
    };
    for_each_index(process, std::make_index_sequence<sizeof...(L1S)>{});

    // ---- epilogue: (gate in registers,) transpose through LDS, norm + coalesced 16-byte stores ----
    // Every input chunk is consumed, so the chunk buffer becomes the out tile.  fp32 modes: one pass over the 32
    // channels of a tile; bf16 storage (buffer half as large): two passes of 16 channels, rounded to bf16 on the way
    // out.  The wave is alone on its SIMD, so the epilogue is priced in issued instructions: the norm is applied
    // after the transpose (one table entry per written column instead of two dependent lookups per accumulator
    // register) and each lane moves four consecutive columns per instruction.
    wait_vm0();  // nothing is in flight here; this tells the wait-count pass so (see wait_vm0)
    wave_sync_lds();
    float* ot = cbuf;
    constexpr int NPASS = IO16 ? 2 : 1, NCH = 32 / NPASS, RPP = 16 / NPASS;
    auto chan_of = [&](int r) { return 8 * (r >> 2) + 4 * half + (r & 3); };
    const bool out_vec = !(ldo & 3) && ((reinterpret_cast<uintptr_t>(outv) & 15) == 0);
    // emit one job: this lane's values val(r,c) (reg r, component c) of a tile with D components per channel; the
    // tile-local index lc = D*channel + c is written to column col(lc) after multiplication by nrm[ncol(lc)]
    // (ncol < 0: no norm); `width` = valid lc count; `affine`: col(lc) = col(0) + lc and ncol(lc) = ncol(0) + lc.
    auto emit = [&](auto dtag, auto val, auto col, auto ncol, const int width, const bool affine) {
      constexpr int D = decltype(dtag)::value;
      constexpr int TS = NCH * D + 4;    // row stride, dwords: 4 * odd (16-byte aligned rows, spread over the banks)
      constexpr int UPR = NCH * D / 4;   // 4-column units per row and pass
      const int colb = col(0), ncolb = ncol(0);
      const bool vec = affine && out_vec && !(colb & 3) && !(width & 3);
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {
#pragma unroll
        for (int r = 0; r < RPP; ++r)
#pragma unroll
          for (int c = 0; c < D; ++c) ot[j * TS + D * (chan_of(ps * RPP + r) - ps * NCH) + c] = val(ps * RPP + r, c);
        wave_sync_lds();
        tick(4);
        if (vec) {
          // index math in 24-bit multiplies (full rate; 32-bit integer multiplies run at a quarter of it) and a
          // uniform 64-bit tile base + 32-bit lane offset for the stores
          constexpr uint32_t INV = (65536 + UPR - 1) / UPR;  // u / UPR == (u * INV) >> 16 for u < 32 * UPR
          static_assert(((32u * UPR - 1) * INV >> 16) == 31 && ((31u * UPR + UPR - 1) * INV >> 16) == 31 &&
                        ((31u * UPR) * INV >> 16) == 31 && ((30u * UPR + UPR - 1) * INV >> 16) == 30, "reciprocal");
          const uint32_t ldo32 = (uint32_t)ldo;
          const float* nbase = ncolb >= 0 ? nrm + ncolb + ps * NCH * D : nrm + Dout;
          const uint32_t nstep = ncolb >= 0 ? 4u : 0u;
#pragma unroll 2  // two in flight; full unrolling lets the scheduler hoist every LDS read and spill the accumulators
          for (int it = 0; it < UPR / 2; ++it) {
            const uint32_t u = it * 64 + lane;
            const uint32_t row = __umul24(u, INV) >> 16, un = u - __umul24(row, UPR);
            const uint32_t lc0 = ps * NCH * D + un * 4;
            float4 v = *reinterpret_cast<const float4*>(ot + __umul24(row, TS) + un * 4);
            const float* np = nbase + un * nstep;
            const float n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
            if ((int)row < nrows && (int)lc0 < width && !(dbg & 1)) {
              v.x *= n0; v.y *= n1; v.z *= n2; v.w *= n3;
              const uint32_t o = __umul24(row, ldo32) + (uint32_t)colb + lc0;
              if (IO16) {
                uint2 pk;
                pk.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{v.x, v.y}, bf16x2_t));
                pk.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{v.z, v.w}, bf16x2_t));
                *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(outv) + row0 * ldo + o) = pk;
              } else {
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(outv) + row0 * ldo + o) = v;
              }
            }
          }
        } else {
          for (int lc = lane; lc < NCH * D; lc += 64) {
            const int glc = ps * NCH * D + lc;
            if (glc >= width) continue;
            const int64_t c0 = row0 * ldo + col(glc);
            const int nc = ncol(glc);
            const float nv = nc >= 0 ? nrm[nc] : 1.f;
            const float* src = ot + lc;
            // cold path (unaligned or scattered columns): rolled on purpose -- unrolled, its 32 row offsets were hoisted
            // out of the tile loop and spilled
#pragma unroll 1
            for (int r = 0; r < nrows; ++r) {
              const float v = src[r * TS] * nv;
              if (dbg & 1) continue;
              if (IO16)
                reinterpret_cast<uint16_t*>(outv)[c0 + (int64_t)r * ldo] = __builtin_bit_cast(uint16_t, (__bf16)v);
              else
                reinterpret_cast<float*>(outv)[c0 + (int64_t)r * ldo] = v;
            }
          }
        }
        wave_sync_lds();
        tick(5);
      }
    };
    using I1 = std::integral_constant<int, 1>;
    using I3 = std::integral_constant<int, 3>;
    using I5 = std::integral_constant<int, 5>;
    if (GATE) {
      // TP out irreps = [32 scalars | 32 gates per gated block | 32x1o | 32x2e]: a0[0] scalars, a0[1..] gates;
      // written layout = [silu(s) (32) | sigmoid(g1) v1 (96) | sigmoid(g2) v2 (160)]
      const float* nrm0 = nrm + ocl[cOoff[0]];  // the scalar block is one irreps-contiguous run of 32 * NT0 columns
      emit(I1{}, [&](int r, int) { const float s = a0[0][0][r] * nrm0[chan_of(r)]; return s * sigmoid_(s); },
           [&](int lc) { return lc; }, [&](int) { return -1; }, 32, true);
      int ocol = 32;
      if (NT1 > 0) {
        float g[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) g[r] = sigmoid_(a0[NT0 > 1 ? 1 : 0][0][r] * nrm0[32 + chan_of(r)]);
        const int nb = ocl[cOoff[1]];
        emit(I3{}, [&](int r, int c) { return g[r] * a1[0][c][r]; }, [&](int lc) { return ocol + lc; },
             [&](int lc) { return nb + lc; }, 96, true);
        ocol += 96;
      }
      if (NT2 > 0) {
        constexpr int G2 = (NT1 > 0) ? 2 : 1;  // which scalar tile holds the gates of the 2e block
        float g[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) g[r] = sigmoid_(a0[NT0 > G2 ? G2 : 0][0][r] * nrm0[32 * G2 + chan_of(r)]);
        const int nb = ocl[cOoff[2]];
        emit(I5{}, [&](int r, int c) { return g[r] * a2[0][c][r]; }, [&](int lc) { return ocol + lc; },
             [&](int lc) { return nb + lc; }, 160, true);
      }
    } else {
      // channels of a tile may belong to several irreps blocks: per-channel column lookup unless the tile's columns
      // are contiguous (one block); padded channels (>= M) are never copied (lc >= width)
      auto tile = [&](auto dtag, int l3, int t, auto val) {
        constexpr int D = decltype(dtag)::value;
        const int base = cOoff[l3] + t * 32;
        const int cnt = cM[l3] - t * 32 < 32 ? cM[l3] - t * 32 : 32;
        const bool affine = ocl[base + cnt - 1] == ocl[base] + (cnt - 1) * D;
        auto colf = [&](int lc) { return ocl[base + lc / D] + lc % D; };
        emit(dtag, val, colf, colf, cnt * D, affine);
      };
#pragma unroll
      for (int t = 0; t < NT0; ++t) tile(I1{}, 0, t, [&](int r, int) { return a0[t][0][r]; });
#pragma unroll
      for (int t = 0; t < NT1; ++t) tile(I3{}, 1, t, [&](int r, int c) { return a1[t][c][r]; });
#pragma unroll
      for (int t = 0; t < NT2; ++t) tile(I5{}, 2, t, [&](int r, int c) { return a2[t][c][r]; });
    }
    tick(4);
  }
  if (prof && lane == 0)
    for (int q = 0; q < 8; ++q) atomicAdd(&prof[q], tacc[q]);
}

struct FastKernelEntry {
  int lsh, nt0, nt1, nt2;
  std::vector<int> l1s;
  const void* fn[3][2][2];  // [mode: 0 exact fp32, 1 fp32 + bf16x3 split, 2 bf16 storage][wlds][gate]
};
#define E3_FAST_M(LSH, a, b, c, M, ...)                                                                \
    {{(const void*)tp_fwd_mfma_kernel<LSH, a, b, c, false, false, M, __VA_ARGS__>,                      \
      (const void*)tp_fwd_mfma_kernel<LSH, a, b, c, false, true, M, __VA_ARGS__>},                      \
     {(const void*)tp_fwd_mfma_kernel<LSH, a, b, c, true, false, M, __VA_ARGS__>,                       \
      (const void*)tp_fwd_mfma_kernel<LSH, a, b, c, true, true, M, __VA_ARGS__>}}
#define E3_FAST(LSH, a, b, c, ...)                                                                     \
  {LSH, a, b, c, {__VA_ARGS__},                                                                         \
   {E3_FAST_M(LSH, a, b, c, 0, __VA_ARGS__), E3_FAST_M(LSH, a, b, c, 1, __VA_ARGS__),                   \
    E3_FAST_M(LSH, a, b, c, 2, __VA_ARGS__)}}
