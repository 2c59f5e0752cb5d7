// Does `s_waitcnt vmcnt(N)` order LDS-DMA (global_load_lds_dwordx4) completions on gfx950 the way the conv kernels' counted waits assume?
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_dma_order tools/lds_dma_order.hip && /tmp/lds_dma_order
//
// Every wave of a workgroup (8 waves, like the persistent conv kernel) fills K KiB-slots of LDS with a sentinel, issues K LDS-DMA loads
// (1 KiB each: 16 bytes per lane) into them -- sources "hot" (a 64-KiB region: L2 hits) or "cold" (random KiB of a 4-GiB buffer: HBM) by
// pattern -- then waits with vmcnt(K - J): if completions are reported in issue order, the OLDEST J loads have landed. It then
//   (own)   reads its own J oldest slots straight away,
//   (other) meets the workgroup at a raw s_barrier and reads the J oldest slots of the NEXT wave,
// and counts sentinel words (a load that had not landed) and wrong words. A second stream can run a "thrasher" (random 16-byte gathers and
// 4-byte scattered stores, the access pattern of the one-workgroup proposal NMS) beside it. Patterns:
//   0 all hot   1 all cold   2 oldest J cold, the rest hot   3 oldest J hot, the rest cold   4 alternating
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static constexpr uint32_t SENT = 0xFFFFFFFFu;

__device__ __forceinline__ void glds16(const void* sbase, uint32_t voff, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2" : : "v"(voff), "s"(lds_dst), "s"(sbase) : "memory", "m0");
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ uint32_t rng(uint32_t& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

template <int K, int J>
__global__ __launch_bounds__(512) void order_kernel(const char* hot, const char* cold, uint32_t cold_kib, unsigned long long* out, int iters, uint32_t seed, int pattern) {
  extern __shared__ __attribute__((aligned(16))) char smem[];          // 8 waves x K KiB
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  uint32_t st = seed + blockIdx.x * 7919u + wave * 104729u;
  unsigned long long own_sent = 0, own_bad = 0, oth_sent = 0, oth_bad = 0;
  uint32_t expect[K];
  __shared__ uint32_t s_expect[8][K];
  for (int it = 0; it < iters; ++it) {
    char* mine = smem + wave * (K * 1024);
#pragma unroll
    for (int k = 0; k < K; ++k) *(uint4*)(mine + k * 1024 + lane * 16) = make_uint4(SENT, SENT, SENT, SENT);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int k = 0; k < K; ++k) {
      bool is_cold;
      switch (pattern) {
        case 0: is_cold = false; break;
        case 1: is_cold = true; break;
        case 2: is_cold = k < J; break;
        case 3: is_cold = k >= J; break;
        default: is_cold = (k & 1) == 0; break;
      }
      const uint32_t r = (uint32_t)__builtin_amdgcn_readfirstlane((int)rng(st));
      const uint32_t kib = is_cold ? r % cold_kib : r % 64u;
      const char* base = (is_cold ? cold : hot) + (size_t)kib * 1024;
      const unsigned long long a = (unsigned long long)(uintptr_t)base;
      const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
      const char* sb = (const char*)(uintptr_t)(((unsigned long long)hi << 32) | lo);
      expect[k] = (is_cold ? 0x40000000u : 0x20000000u) | kib;            // the buffers hold (tag | KiB index) in every word
      glds16(sb, (uint32_t)lane * 16u, (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds0 + wave * (K * 1024) + k * 1024)));
    }
    wait_vm<K - J>();
    // (own) the J oldest of my slots
#pragma unroll
    for (int k = 0; k < J; ++k) {
      const uint4 v = *(const uint4*)(mine + k * 1024 + lane * 16);
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) { own_sent += w[e] == SENT; own_bad += (w[e] != SENT && w[e] != expect[k]); }
    }
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < K; ++k) s_expect[wave][k] = expect[k];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // (other) the J oldest slots of the next wave, right behind the barrier
    const int ow = (wave + 1) & 7;
#pragma unroll
    for (int k = 0; k < J; ++k) {
      const uint4 v = *(const uint4*)(smem + ow * (K * 1024) + k * 1024 + lane * 16);
      const uint32_t ex = s_expect[ow][k];
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) { oth_sent += w[e] == SENT; oth_bad += (w[e] != SENT && w[e] != ex); }
    }
    wait_vm<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  if (own_sent) atomicAdd(out + 0, own_sent);
  if (own_bad) atomicAdd(out + 1, own_bad);
  if (oth_sent) atomicAdd(out + 2, oth_sent);
  if (oth_bad) atomicAdd(out + 3, oth_bad);
  if (threadIdx.x == 0) atomicAdd(out + 4, (unsigned long long)iters);
}

// the neighbour: random 16-byte gathers + 4-byte scattered stores for `usec` microseconds
__global__ __launch_bounds__(1024) void thrash_kernel(const uint4* src, uint32_t n16, uint32_t* dst, uint32_t n4, int usec, unsigned* sink) {
  uint32_t st = blockIdx.x * 977u + threadIdx.x * 31u + 1u;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  uint32_t acc = 0;
  while ((__builtin_amdgcn_s_memrealtime() - t0) < (unsigned long long)usec * 100ull) {
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
      const uint4 v = src[(rng(st) * 2654435761u) % n16];
      acc += v.x ^ v.w;
      dst[(rng(st) * 2246822519u) % n4] = acc;
    }
  }
  if (acc == 0x12345u) *sink = acc;
}

__global__ void fill_kernel(uint32_t* p, size_t words, uint32_t tag) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) p[i] = tag | (uint32_t)(i >> 8);
}

int main(int argc, char** argv) {
  int iters = 2000, reps = 3;
  size_t cold_gib = 4;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--iters")) iters = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--reps")) reps = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--cold-gib")) cold_gib = (size_t)atoi(argv[++i]);
  }
  constexpr int K = 8, J = 4;
  char *hot, *cold; uint32_t* scratch; unsigned long long* out; unsigned* sink;
  const size_t cold_bytes = cold_gib << 30, scratch_bytes = (size_t)1 << 30;
  CK(hipMalloc(&hot, 64 * 1024)); CK(hipMalloc(&cold, cold_bytes)); CK(hipMalloc(&scratch, scratch_bytes));
  CK(hipMalloc(&out, 8 * sizeof(unsigned long long))); CK(hipMalloc(&sink, 4));
  fill_kernel<<<1024, 256>>>((uint32_t*)hot, 64 * 1024 / 4, 0x20000000u);
  fill_kernel<<<4096, 256>>>((uint32_t*)cold, cold_bytes / 4, 0x40000000u);
  CK(hipDeviceSynchronize());
  hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
  CK(hipFuncSetAttribute((const void*)order_kernel<K, J>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * K * 1024));
  const char* pname[] = {"all hot", "all cold", "oldest J cold, rest hot", "oldest J hot, rest cold", "alternating"};
  printf("K = %d LDS-DMA loads per wave and round, wait vmcnt(%d), then read the %d oldest slots; 256 workgroups x 8 waves x %d rounds\n", K, K - J, J, iters);
  printf("%-28s %-10s %14s %14s %16s %16s\n", "pattern", "neighbour", "own: not landed", "own: wrong", "other: not landed", "other: wrong");
  for (int with_thrash = 0; with_thrash < 2; ++with_thrash)
    for (int p = 0; p < 5; ++p) {
      unsigned long long tot[5] = {0, 0, 0, 0, 0};
      for (int r = 0; r < reps; ++r) {
        CK(hipMemsetAsync(out, 0, 8 * sizeof(unsigned long long), s1));
        CK(hipStreamSynchronize(s1));
        if (with_thrash) thrash_kernel<<<96, 1024, 0, s2>>>((const uint4*)cold, (uint32_t)(cold_bytes / 16 > 0xFFFFFFF0ull ? 0xFFFFFFF0u : cold_bytes / 16), scratch, (uint32_t)(scratch_bytes / 4), 20000, sink);
        order_kernel<K, J><<<256, 512, 8 * K * 1024, s1>>>(hot, cold, (uint32_t)(cold_bytes / 1024), out, iters, 12345u + r * 17u + p, p);
        CK(hipGetLastError());
        CK(hipStreamSynchronize(s1));
        CK(hipStreamSynchronize(s2));
        unsigned long long h[5];
        CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
        for (int i = 0; i < 5; ++i) tot[i] += h[i];
      }
      printf("%-28s %-10s %14llu %14llu %16llu %16llu   (workgroup rounds: %llu)\n", pname[p], with_thrash ? "thrasher" : "none", tot[0], tot[1], tot[2], tot[3], tot[4]);
      fflush(stdout);
    }
  return 0;
}
