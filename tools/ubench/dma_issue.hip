// Microbenchmark: how fast can one wave issue LDS-DMA (global_load_lds) instructions on gfx950?
//   mode 0: 16-byte DMA, M0 rewritten for every instruction (runtime LDS destination)
//   mode 1: 16-byte DMA, constant M0, destination advanced through the immediate offset
//   mode 2: global_load_dwordx4 into VGPRs, then ds_write_b128 (classic staging)
//   mode 3: 4-byte DMA, M0 rewritten
// Prints shader cycles per instruction for 1 wave/CU and 4 waves/CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <utility>
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) void glb_void_t;
constexpr int NI = 32;  // instructions per round

template <int... I>
__device__ __forceinline__ void issue_imm(const float* base, float* dst, int stride_rows, std::integer_sequence<int, I...>) {
  // one M0 value for four instructions: the destination advances through the immediate offset (max 4095)
  (__builtin_amdgcn_global_load_lds((glb_void_t*)(base + (size_t)I * stride_rows), (lds_void_t*)(dst + (I / 4) * 1024), 16,
                                    (I % 4) * 1024 - (I % 4 == 3 ? 0 : 0), 0), ...);
}

template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ src, float* __restrict__ out, unsigned long long* cyc,
                                         int rounds, int stride_rows) {
  extern __shared__ __align__(16) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* dst = lds + wave * (NI * 256 + 64);
  const float* base = src + ((size_t)blockIdx.x * 4 + wave) * 4096 + lane * 4;
  float acc = 0.f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < rounds; ++r) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < NI; ++i)
        __builtin_amdgcn_global_load_lds((glb_void_t*)(base + (size_t)i * stride_rows), (lds_void_t*)(dst + i * 256), 16, 0, 0);
    } else if (MODE == 1) {
      issue_imm(base, dst, stride_rows, std::make_integer_sequence<int, NI>{});
    } else if (MODE == 2) {
      float4 v[NI];
#pragma unroll
      for (int i = 0; i < NI; ++i) v[i] = *reinterpret_cast<const float4*>(base + (size_t)i * stride_rows);
#pragma unroll
      for (int i = 0; i < NI; ++i) *reinterpret_cast<float4*>(dst + i * 256 + lane * 4) = v[i];
    } else {
#pragma unroll
      for (int i = 0; i < NI; ++i)
        __builtin_amdgcn_global_load_lds((glb_void_t*)(base + (size_t)i * stride_rows), (lds_void_t*)(dst + i * 64), 4, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    acc += dst[lane];
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) atomicAdd(cyc, t1 - t0);
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
  const int nblk = 256;
  float *src, *out; unsigned long long* cyc;
  hipMalloc(&src, (size_t)nblk * 4 * 4096 * 4 * 2 + (1 << 22));
  hipMemset(src, 0, (size_t)nblk * 4 * 4096 * 4 * 2 + (1 << 22));
  hipMalloc(&out, nblk * 256 * 4);
  hipMalloc(&cyc, 8);
  const int rounds = 200;
  const size_t shm = 4 * (NI * 256 + 64) * 4;
  auto run = [&](auto kern, const char* name, int threads) {
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    for (int rep = 0; rep < 2; ++rep) {
      hipMemset(cyc, 0, 8);
      hipLaunchKernelGGL(kern, dim3(nblk), dim3(threads), shm, 0, src, out, cyc, rounds, 64);
      hipDeviceSynchronize();
    }
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double waves = (double)nblk * threads / 64;
    printf("%-34s waves/CU %d: %.0f cycles per instruction (incl. final wait), %.1f per round\n", name, threads / 64,
           (double)c / waves / rounds / NI, (double)c / waves / rounds);
  };
  for (int threads : {64, 256}) {
    run(k<0>, "dma16, M0 per instruction", threads);
    run(k<1>, "dma16, constant M0 + imm offset", threads);
    run(k<2>, "global_load x4 -> ds_write_b128", threads);
    run(k<3>, "dma4, M0 per instruction", threads);
  }
  printf("hip status: %s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
