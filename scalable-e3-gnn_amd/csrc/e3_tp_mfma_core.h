// Device-side building blocks shared by the MFMA tensor-product kernels (e3_tp_mfma_r16.hip: one TP per launch;
// e3_msg_fused.hip: the whole message function per launch).  Internal header, included inside namespace e3.
#pragma once
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

constexpr int kFastLds = 160 * 1024;

// Wait for every outstanding vector-memory operation (the LDS-DMA copies).  The asm is the compiler barrier; the builtin
// is the same instruction again in a form the backend's wait-count pass can see -- without it the pass believes the
// copies are still in flight and guards later LDS accesses with its own vmcnt(0), e.g. in every iteration of the
// epilogue's store loop (which then waits for the previous iteration's global store: ~650 cycles each).
__device__ __forceinline__ void wait_vm0() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0x0F70);  // gfx9 encoding: vmcnt 0, expcnt 7, lgkmcnt 15 (= no wait on those)
}
__device__ __forceinline__ void wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));

// v_mfma_f32_16x16x32_{bf16,f16}: A[row = lane & 15][k = 8 (lane >> 4) + i], B[k = 8 (lane >> 4) + i][col = lane & 15],
// D[row = 4 (lane >> 4) + r][col = lane & 15]  (cdna_hip_programming.md §3)
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16h(f16x8 a, f16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// fp32 operand -> fp16 (hi, lo) pair: hi = rne16(v), lo = rne16(v - hi); hi + lo carries 22 significant bits while lo is
// a normal fp16 number, i.e. for |v| >= 2^-3 (callers scale operands by a power of two so that their largest magnitude
// sits near 2^10 .. 2^14; below 2^-3 the absolute error is <= 2^-25).  A product a * b is accumulated as
// ah*bh + ah*bl + al*bh on the f16 MFMA pipe with fp32 accumulation: relative error ~2^-21 (the dropped al*bl term).
__device__ __forceinline__ void split2_f16(float v0, float v1, uint32_t& hi, uint32_t& lo) {
  const f16x2_t h = __builtin_convertvector(f32x2_t{v0, v1}, f16x2_t);
  const f32x2_t hb = __builtin_convertvector(h, f32x2_t);
  const f16x2_t l = __builtin_convertvector(f32x2_t{v0 - hb.x, v1 - hb.y}, f16x2_t);
  hi = __builtin_bit_cast(uint32_t, h);
  lo = __builtin_bit_cast(uint32_t, l);
}

// power-of-two scale s with amax * s in [2^target, 2^(target+1)) from the bits of amax (finite, >= 0); 1 for amax == 0
__host__ __device__ __forceinline__ float pow2_scale_from_bits(uint32_t amax_bits, int target) {
  const int e = (int)((amax_bits >> 23) & 0xff);
  if (e == 0 || e == 0xff) return 1.0f;
  int se = 127 + target - (e - 127);
  se = se < 1 ? 1 : (se > 254 ? 254 : se);
  const uint32_t b = (uint32_t)se << 23;
  float f;
  __builtin_memcpy(&f, &b, 4);
  return f;
}

// Compile-time bookkeeping: which (l1, l2, l3) paths exist for natural-parity irreps with SH degree <= LSH and NT*
// output tiles per degree.
template <int LSH, int NT0, int NT1, int NT2>
struct PathSlots {
  static constexpr int nt(int l3) { return l3 == 0 ? NT0 : l3 == 1 ? NT1 : NT2; }
  static constexpr bool valid(int l1, int l2, int l3) {
    return l1 >= 0 && nt(l3) > 0 && l2 <= LSH && ((l1 + l2 + l3) % 2 == 0) && l3 >= (l1 > l2 ? l1 - l2 : l2 - l1) &&
           l3 <= l1 + l2;
  }
};
template <int... V>
struct IntSeq {
  static constexpr int n = sizeof...(V);
  static constexpr int at(int i) {
    constexpr int a[] = {V...};
    return (i >= 0 && i < n) ? a[i] : -1;
  }
};
template <class F, size_t... I>
__device__ __forceinline__ void for_each_index(F&& f, std::index_sequence<I...>) {
  (f(std::integral_constant<int, (int)I>{}), ...);
}

struct SegArgs {
  const void* base[4];
  int64_t ld[4];
  const int32_t* index[4];
  int col0[5];  // first in1 column of each segment; col0[nseg] = D1
  int nseg;
  const int32_t* scatter;  // fused segment-sum (e3_tp_forward_fused_scatter): node id of every row, ascending; else null
  // epilogue extras (e3_tp_forward_fused_epilogue; plain stores only): out = product + residual, and the running max |out|
  const void* residual;    // [B, out width], storage dtype, same column layout as out; null = none
  int64_t ldr;
  uint32_t* amax;          // float bits, atomicMax of every finite |out| this launch wrote; null = none
};

__device__ __forceinline__ float sigmoid_(float v) { return __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
