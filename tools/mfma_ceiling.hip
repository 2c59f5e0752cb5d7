// What do the matrix cores of THIS part deliver on benchmark-like data at the package power cap? (VERDICT r5 "next" 3)
//
// DESIGN.md section 4 argues that the conv stack's ~52 % of the NOMINAL bf16 peak (2.5 PFLOP/s at 2.4 GHz) is a power ceiling: the part runs
// the workload at 1.85 - 1.9 GHz, and MFMAs on all-zero operands run in the same cycles at 2.35 GHz. Rounds 3 - 5 leaned on a number quoted
// from the programming guide (its 256^2 8-phase GEMM: 1320 - 1340 TFLOP/s on random operands); the guide's example source is not in this
// image, so that number cannot be re-measured here. This program measures the bound itself, with nothing but v_mfma_f32_32x32x16_bf16 in
// the loop -- no convolution, no halo, no epilogue, no HBM traffic -- and every CU busy with two waves per SIMD (the conv kernels' occupancy):
//
//   reg   operands and accumulators in registers: the matrix pipe alone. An UPPER bound for any kernel built from this instruction.
//   lds   every MFMA's operands come from LDS (`--reads-per-4 N`: N ds_read_b128 per four MFMAs; 4 = one fragment read per MFMA, the
//         rate at which one wave per MFMA saturates the LDS' 128 B/clk; 3 = the persistent conv kernel's mix after fragment reuse).
//   dma   `lds` + LDS-DMA streaming (global_load_lds_dwordx4 out of an L2-resident buffer, one KiB per wave per `--mfma-per-kib` MFMAs:
//         64 = what a 256 x 128-tile K step of the conv kernels moves per wave), i.e. a GEMM main loop without its bookkeeping.
//
// `--data random` fills operands / LDS / the streamed buffer with uniform bf16 in [-1, 1), `--data zero` with zeros. The kernel is launched
// back to back for `--seconds`; TFLOP/s = MFMAs x 32768 / elapsed. tools/mfma_ceiling.py runs the matrix and samples shader clock and
// package power next to it (the sampler bench.py uses).  Build:  hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_ceiling tools/mfma_ceiling.hip
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                          \
  do {                                                                                    \
    hipError_t e_ = (x);                                                                  \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
  } while (0)

__device__ __forceinline__ f32x16 mfma(const u32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

constexpr int ITER_MFMA = 16;      // MFMAs per loop iteration and wave: a 2 x 2 accumulator block x 4 k-slices (the conv kernels' wave tile)

// MODE 0: registers only. 1: operands through LDS (RP4 reads per 4 MFMAs). 2: 1 + LDS-DMA streaming.
template <int MODE, int RP4>
__global__ __launch_bounds__(512) void ceiling_kernel(const u32x4* __restrict__ ops, const char* __restrict__ stream, size_t stream_bytes,
                                                      float* __restrict__ sink, int iters, int mfma_per_kib) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  u32x4 a[2][4], b[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      a[i][q] = ops[((size_t)(i * 4 + q) * 4096 + (blockIdx.x * 512 + tid) % 4096)];
      b[i][q] = ops[((size_t)(8 + i * 4 + q) * 4096 + (blockIdx.x * 512 + tid) % 4096)];
    }
  constexpr int LDS_OPS = 128 * 1024;                // operand region: 128 KiB of fragments (16 KiB per wave); behind it 2 x 8 x 1 KiB DMA landing zones
  if constexpr (MODE >= 1) {
    for (int i = tid; i < LDS_OPS / 16; i += 512) ((u32x4*)smem)[i] = ops[i % (16 * 4096)];
    __syncthreads();
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const char* rd = smem + wave * 16384 + lane * 16;  // a wave walks its own 16 KiB of fragments: conflict-free 16-byte lanes, 16 distinct KiB per iteration
  unsigned kib = 0;
  const size_t nkib = stream_bytes / 1024;
  int since = 0;
  for (int it = 0; it < iters; ++it) {
    const int rot = (it & 7) * 2048;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      u32x4 xa0 = a[0][q], xa1 = a[1][q], xb0 = b[0][q], xb1 = b[1][q];
      if constexpr (MODE >= 1) {
        // RP4 of the four fragments of this k-slice group come from LDS, the rest stay in registers
        if constexpr (RP4 >= 1) xb0 = *(const u32x4*)(rd + ((rot + (q * 4 + 0) * 1024) & 16383));
        if constexpr (RP4 >= 2) xb1 = *(const u32x4*)(rd + ((rot + (q * 4 + 1) * 1024) & 16383));
        if constexpr (RP4 >= 3) xa0 = *(const u32x4*)(rd + ((rot + (q * 4 + 2) * 1024) & 16383));
        if constexpr (RP4 >= 4) xa1 = *(const u32x4*)(rd + ((rot + (q * 4 + 3) * 1024) & 16383));
      }
      acc[0][0] = mfma(xa0, xb0, acc[0][0]);
      acc[0][1] = mfma(xa0, xb1, acc[0][1]);
      acc[1][0] = mfma(xa1, xb0, acc[1][0]);
      acc[1][1] = mfma(xa1, xb1, acc[1][1]);
    }
    if constexpr (MODE == 2) {
      since += ITER_MFMA;
      if (since >= mfma_per_kib) {                    // wave-uniform
        since = 0;
        const size_t k = ((size_t)blockIdx.x * 8 + wave + (size_t)kib * 2048) % nkib;
        const char* src = stream + k * 1024 + lane * 16;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + LDS_OPS + ((kib & 1) * 8 + wave) * 1024);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(uintptr_t)dst, 16, 0, 0);
        ++kib;
      }
    }
  }
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) t += acc[i][j][r];
  if (t == 1234.5678f) sink[blockIdx.x * 512 + tid] = t;       // never true on purpose-made data; keeps the loop alive
}

static uint16_t bf16_of(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

int main(int argc, char** argv) {
  std::string mode = "reg", data = "random";
  double seconds = 3.0;
  int rp4 = 4, per_kib = 64, iters = 4096;
  for (int i = 1; i < argc; ++i) {
    std::string k = argv[i];
    auto val = [&]() -> const char* { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", k.c_str()); exit(2); } return argv[++i]; };
    if (k == "--mode") mode = val();
    else if (k == "--data") data = val();
    else if (k == "--seconds") seconds = atof(val());
    else if (k == "--reads-per-4") rp4 = atoi(val());
    else if (k == "--mfma-per-kib") per_kib = atoi(val());
    else if (k == "--iters") iters = atoi(val());
    else { fprintf(stderr, "unknown argument %s\n", k.c_str()); return 2; }
  }
  const int m = mode == "reg" ? 0 : mode == "lds" ? 1 : mode == "dma" ? 2 : -1;
  if (m < 0 || (data != "random" && data != "zero") || rp4 < 1 || rp4 > 4 || per_kib < ITER_MFMA) { fprintf(stderr, "bad arguments\n"); return 2; }
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  const size_t nops = (size_t)16 * 4096 * 8;         // bf16 elements of the operand table (16 fragment planes x 4096 lanes x 8)
  const size_t stream_bytes = (size_t)64 << 20;      // 64 MiB: streams out of the L2 / Infinity Cache like a layer's weights and windows
  std::vector<uint16_t> h(nops + stream_bytes / 2, 0);
  if (data == "random") {
    std::mt19937 rng(1);
    std::uniform_real_distribution<float> u(-1.f, 1.f);
    for (auto& v : h) v = bf16_of(u(rng));
  }
  uint16_t* d = nullptr;
  float* sink = nullptr;
  CHECK(hipMalloc((void**)&d, h.size() * 2));
  CHECK(hipMalloc((void**)&sink, (size_t)ncu * 8 * 512 * 4));
  CHECK(hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  const u32x4* ops = (const u32x4*)d;
  const char* stream = (const char*)(d + nops);
  const int lds = m == 0 ? 0 : 128 * 1024 + 16 * 1024;
  const int grid = ncu * 4;                          // four workgroups of 8 waves per CU and launch; one resident at a time (registers)
  auto launch = [&]() {
#define L(M, R) hipLaunchKernelGGL((ceiling_kernel<M, R>), dim3(grid), dim3(512), lds, 0, ops, stream, stream_bytes, sink, iters, per_kib)
    if (m == 0) L(0, 4);
    else if (m == 1) { if (rp4 == 1) L(1, 1); else if (rp4 == 2) L(1, 2); else if (rp4 == 3) L(1, 3); else L(1, 4); }
    else { if (rp4 == 1) L(2, 1); else if (rp4 == 2) L(2, 2); else if (rp4 == 3) L(2, 3); else L(2, 4); }
#undef L
  };
  if (m >= 1) {
    CHECK(hipFuncSetAttribute((const void*)ceiling_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CHECK(hipFuncSetAttribute((const void*)ceiling_kernel<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CHECK(hipFuncSetAttribute((const void*)ceiling_kernel<1, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CHECK(hipFuncSetAttribute((const void*)ceiling_kernel<1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CHECK(hipFuncSetAttribute((const void*)ceiling_kernel<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CHECK(hipFuncSetAttribute((const void*)ceiling_kernel<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CHECK(hipFuncSetAttribute((const void*)ceiling_kernel<2, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CHECK(hipFuncSetAttribute((const void*)ceiling_kernel<2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  }
  launch();
  CHECK(hipGetLastError());
  CHECK(hipDeviceSynchronize());
  // back-to-back launches for `seconds`; the first 30 % are a warm-up (the clock settles within a few hundred ms of load)
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const auto t_start = std::chrono::steady_clock::now();
  auto since = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(); };
  long launches = 0;
  bool timing = false;
  while (since() < seconds) {
    if (!timing && since() > 0.3 * seconds) { CHECK(hipDeviceSynchronize()); CHECK(hipEventRecord(e0, 0)); timing = true; launches = 0; }
    for (int k = 0; k < 4; ++k) launch();
    launches += 4;
    CHECK(hipDeviceSynchronize());
  }
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double mfmas = (double)launches * grid * 8.0 * iters * ITER_MFMA;
  const double tf = mfmas * 32768.0 / (ms * 1e-3) / 1e12;
  printf("{\"mode\": \"%s\", \"data\": \"%s\", \"reads_per_4_mfma\": %d, \"mfma_per_kib\": %d, \"cus\": %d, \"waves_per_simd\": 2, \"launches\": %ld, \"ms\": %.3f, "
         "\"tflops\": %.1f, \"frac_of_2500\": %.4f, \"implied_mhz_at_full_issue\": %.0f}\n",
         mode.c_str(), data.c_str(), m == 0 ? 0 : rp4, m == 2 ? per_kib : 0, ncu, launches, ms, tf, tf / 2500.0, tf / 2500.0 * 2400.0);
  return 0;
}
