// Diagnostic: what does the conv GEMM's main loop -- ds_read_b128 fragment reads + v_mfma_f32_32x32x2_f32,
// 8 waves per workgroup, 2 workgroups per CU, one barrier per 128-byte K stage -- sustain when nothing
// else is in the way (LDS filled once, no global loads), and which scheduling of the fragment reads
// gets closest to the bare-MFMA rate (154.6 TFLOP/s on this device, tools/mfma_shapes.hip)?
//   V0  the product's loop as hipcc schedules it (reads of chunk kb just before its MFMAs)
//   V1  all 12 fragment reads of a stage issued up front, counted lgkmcnt waits before each chunk
//   V2  V1 + the next stage's reads issued before the last chunk's MFMAs (no barrier between stages)
//   V3  V0 without the per-stage barrier
//   V4  V0 + the product's LDS-DMA of the next stage (4 x global_load_lds_dwordx4 per wave and stage) from a
//       buffer far larger than L2 / the Infinity Cache (every workgroup streams its own rows)
//   V5  V4 with every workgroup re-reading the same 64 KiB (L2-resident source)
//   V6  V4 with the four DMA pieces spread over the stage (one before each chunk's MFMAs) instead of up front
//   V7  V4 with the second half of the workgroup (waves 4-7) at s_setprio 1
//   V8  V4 with all 32 pieces of a stage issued by waves 0-3 (8 each), none by waves 4-7
//   V9  V4 with half the pieces (2 per wave and stage): timing only
//   V10 V4 with 4-byte pieces (global_load_lds_dword: same instruction count, a quarter of the bytes)
//   V11 V4 with the LDS destinations (M0 values) computed once, in SGPRs
//   hipcc -O3 --offload-arch=gfx950 tools/loop_bench.hip -o build/loop_bench && build/loop_bench
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, ROWB = 128, WM = 2, WN = 4;

__device__ __forceinline__ floatx16 mfma32(float a, float b, floatx16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

template <int V>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void loop_kernel(float *out, int stages,
                                                                                                unsigned seed,
                                                                                                const float *src,
                                                                                                long src_floats,
                                                                                                unsigned long long *stamps) {
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  __shared__ __attribute__((aligned(16))) char lds[2 * (BM + BN) * ROWB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN, r = lane & 31, h = lane >> 5;
  unsigned s = seed ^ (tid * 2654435761u) ^ (blockIdx.x * 40503u);
  for (int i = tid; i < 2 * (BM + BN) * ROWB / 4; i += 512) {
    s = s * 1664525u + 1013904223u;
    reinterpret_cast<float *>(lds)[i] = (float)(int)(s >> 8) * (1.0f / 8388608.0f) - 1.0f;
  }
  __syncthreads();
  char *As = lds, *Bs = lds + 2 * BM * ROWB;
  floatx16 acc0, acc1;
  for (int q = 0; q < 16; ++q) acc0[q] = acc1[q] = 0.f;
  const int sw = (r >> 1) & 7;
  int co[4];
  for (int kb = 0; kb < 4; ++kb) co[kb] = 16 * ((2 * kb + h) ^ sw);
  auto a_ptr = [&](int buf, int mi, int kb) { return reinterpret_cast<const floatx4 *>(As + (buf * BM + wm * 64 + mi * 32 + r) * ROWB + co[kb]); };
  auto b_ptr = [&](int buf, int kb) { return reinterpret_cast<const floatx4 *>(Bs + (buf * BN + wn * 32 + r) * ROWB + co[kb]); };
#define MMA4(A0, A1, B)                                                \
  do {                                                                 \
    acc0 = mfma32(A0[0], B[0], acc0); acc1 = mfma32(A1[0], B[0], acc1); \
    acc0 = mfma32(A0[1], B[1], acc0); acc1 = mfma32(A1[1], B[1], acc1); \
    acc0 = mfma32(A0[2], B[2], acc0); acc1 = mfma32(A1[2], B[2], acc1); \
    acc0 = mfma32(A0[3], B[3], acc0); acc1 = mfma32(A1[3], B[3], acc1); \
  } while (0)
  if (V == 0 || V == 3) {
    for (int kt = 0; kt < stages; ++kt) {
      const int buf = kt & 1;
      if (V == 0) __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const floatx4 a0 = *a_ptr(buf, 0, kb), a1 = *a_ptr(buf, 1, kb), b = *b_ptr(buf, kb);
        MMA4(a0, a1, b);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  } else if (V == 1) {
    for (int kt = 0; kt < stages; ++kt) {
      const int buf = kt & 1;
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      floatx4 a0[4], a1[4], b[4];
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        a0[kb] = *a_ptr(buf, 0, kb);
        a1[kb] = *a_ptr(buf, 1, kb);
        b[kb] = *b_ptr(buf, kb);
      }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(9)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      MMA4(a0[0], a1[0], b[0]);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      MMA4(a0[1], a1[1], b[1]);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      MMA4(a0[2], a1[2], b[2]);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      MMA4(a0[3], a1[3], b[3]);
      __builtin_amdgcn_sched_barrier(0);
    }
  } else if (V == 2) {
    floatx4 a0[4], a1[4], b[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      a0[kb] = *a_ptr(0, 0, kb);
      a1[kb] = *a_ptr(0, 1, kb);
      b[kb] = *b_ptr(0, kb);
    }
    for (int kt = 0; kt < stages; ++kt) {
      const int nbuf = (kt + 1) & 1;
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(9)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      MMA4(a0[0], a1[0], b[0]);
      __builtin_amdgcn_sched_barrier(0);
      a0[0] = *a_ptr(nbuf, 0, 0); a1[0] = *a_ptr(nbuf, 1, 0); b[0] = *b_ptr(nbuf, 0);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(9)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      MMA4(a0[1], a1[1], b[1]);
      __builtin_amdgcn_sched_barrier(0);
      a0[1] = *a_ptr(nbuf, 0, 1); a1[1] = *a_ptr(nbuf, 1, 1); b[1] = *b_ptr(nbuf, 1);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(9)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      MMA4(a0[2], a1[2], b[2]);
      __builtin_amdgcn_sched_barrier(0);
      a0[2] = *a_ptr(nbuf, 0, 2); a1[2] = *a_ptr(nbuf, 1, 2); b[2] = *b_ptr(nbuf, 2);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(9)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      MMA4(a0[3], a1[3], b[3]);
      __builtin_amdgcn_sched_barrier(0);
      a0[3] = *a_ptr(nbuf, 0, 3); a1[3] = *a_ptr(nbuf, 1, 3); b[3] = *b_ptr(nbuf, 3);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (V >= 4) {
    typedef const __attribute__((address_space(1))) void *gptr_t;
    typedef __attribute__((address_space(3))) void *lptr_t;
    // per stage the workgroup fetches 256 rows x 128 B = 32 KiB; lane -> row 8 g + lane / 8, chunk lane % 8 (swizzled)
    const int lrow8 = lane >> 3, lpos = lane & 7;
    long base = (V == 5) ? 0 : (long)blockIdx.x * (src_floats / gridDim.x);
    const long span = (V == 5) ? 0 : (src_floats / gridDim.x) - 8192;
    const float *gsrc[4];
    for (int i = 0; i < 4; ++i) {
      const int row = 8 * (wave + 8 * i) + lrow8;
      gsrc[i] = src + base + row * 32 + 4 * (lpos ^ ((row >> 1) & 7));
    }
    if (V == 7 && wave >= 4) __builtin_amdgcn_s_setprio(1);
    const float *gsrc8[8];
    for (int i = 0; i < 8; ++i) {
      const int row = 8 * ((wave & 3) + 4 * i) + lrow8;
      gsrc8[i] = src + base + row * 32 + 4 * (lpos ^ ((row >> 1) & 7));
    }
    const int swave = __builtin_amdgcn_readfirstlane(wave);
    long off = 0;
    for (int kt = 0; kt < stages; ++kt) {
      const int buf = kt & 1, nbuf = buf ^ 1;
      __syncthreads();
      if (V == 8) {
        if (wave < 4) {
#pragma unroll
          for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(gsrc8[i] + off), (lptr_t)(lds + (nbuf * 256 + 8 * (wave + 4 * i)) * ROWB), 16, 0, 0);
        }
      } else if (V == 9) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
          __builtin_amdgcn_global_load_lds((gptr_t)(gsrc[i] + off), (lptr_t)(lds + (nbuf * 256 + 8 * (wave + 8 * i)) * ROWB), 16, 0, 0);
      } else if (V == 10) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          __builtin_amdgcn_global_load_lds((gptr_t)(gsrc[i] + off), (lptr_t)(lds + (nbuf * 256 + 8 * (wave + 8 * i)) * ROWB), 4, 0, 0);
      } else if (V == 11) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          __builtin_amdgcn_global_load_lds((gptr_t)(gsrc[i] + off), (lptr_t)(lds + (nbuf * 256 + 8 * (swave + 8 * i)) * ROWB), 16, 0, 0);
      } else if (V != 6) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          __builtin_amdgcn_global_load_lds((gptr_t)(gsrc[i] + off), (lptr_t)(lds + (nbuf * 256 + 8 * (wave + 8 * i)) * ROWB), 16, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        if (V == 6) {
          __builtin_amdgcn_global_load_lds((gptr_t)(gsrc[kb] + off), (lptr_t)(lds + (nbuf * 256 + 8 * (wave + 8 * kb)) * ROWB), 16, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        const floatx4 a0 = *a_ptr(buf, 0, kb), a1 = *a_ptr(buf, 1, kb), b = *b_ptr(buf, kb);
        MMA4(a0, a1, b);
      }
      __builtin_amdgcn_sched_barrier(0);
      off += 8192;
      if (off > span) off = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  float sum = 0.f;
  for (int q = 0; q < 16; ++q) sum += acc0[q] + acc1[q];
  out[blockIdx.x * 512 + tid] = sum;
  if (tid == 0) {
    stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - c0;
    stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
}

__global__ void fill_random(float *p, long n, unsigned seed) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    unsigned s = seed ^ (unsigned)(i * 2654435761u) ^ (unsigned)(i >> 32);
    s = s * 1664525u + 1013904223u;
    s ^= s >> 15;
    s *= 2246822519u;
    s ^= s >> 13;
    p[i] = (float)(int)(s >> 8) * (1.0f / 8388608.0f) - 1.0f;
  }
}

int main() {
  float *out, *src;
  unsigned long long *stamps;
  hipMalloc(&stamps, 2 * 4096 * sizeof(unsigned long long));
  hipMalloc(&out, 512 * 4096 * sizeof(float));
  const long src_floats = 1L << 30;   // 4 GiB
  hipMalloc(&src, src_floats * sizeof(float));
  hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, src, src_floats, 7u);   // random operands: DVFS is data dependent
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int stages = 20000;
  for (int rep = 0; rep < 2; ++rep)
    for (int v : {0, 4, 5, 9})
      for (int blocks_per_cu : {1, 2}) {
        const int blocks = 256 * blocks_per_cu;
        float ms = 0;
        for (int r = 0; r < 2; ++r) {
          hipEventRecord(e0);
          if (v == 0) hipLaunchKernelGGL(loop_kernel<0>, dim3(blocks), dim3(512), 0, 0, out, stages, 1u, src, src_floats, stamps);
          if (v == 1) hipLaunchKernelGGL(loop_kernel<1>, dim3(blocks), dim3(512), 0, 0, out, stages, 1u, src, src_floats, stamps);
          if (v == 2) hipLaunchKernelGGL(loop_kernel<2>, dim3(blocks), dim3(512), 0, 0, out, stages, 1u, src, src_floats, stamps);
          if (v == 3) hipLaunchKernelGGL(loop_kernel<3>, dim3(blocks), dim3(512), 0, 0, out, stages, 1u, src, src_floats, stamps);
          if (v == 4) hipLaunchKernelGGL(loop_kernel<4>, dim3(blocks), dim3(512), 0, 0, out, stages, 1u, src, src_floats, stamps);
          if (v == 5) hipLaunchKernelGGL(loop_kernel<5>, dim3(blocks), dim3(512), 0, 0, out, stages, 1u, src, src_floats, stamps);
          if (v == 6) hipLaunchKernelGGL(loop_kernel<6>, dim3(blocks), dim3(512), 0, 0, out, stages, 1u, src, src_floats, stamps);
          if (v == 7) hipLaunchKernelGGL(loop_kernel<7>, dim3(blocks), dim3(512), 0, 0, out, stages, 1u, src, src_floats, stamps);
          if (v == 8) hipLaunchKernelGGL(loop_kernel<8>, dim3(blocks), dim3(512), 0, 0, out, stages, 1u, src, src_floats, stamps);
          if (v == 9) hipLaunchKernelGGL(loop_kernel<9>, dim3(blocks), dim3(512), 0, 0, out, stages, 1u, src, src_floats, stamps);
          if (v == 10) hipLaunchKernelGGL(loop_kernel<10>, dim3(blocks), dim3(512), 0, 0, out, stages, 1u, src, src_floats, stamps);
          if (v == 11) hipLaunchKernelGGL(loop_kernel<11>, dim3(blocks), dim3(512), 0, 0, out, stages, 1u, src, src_floats, stamps);
          hipEventRecord(e1);
          hipEventSynchronize(e1);
          hipEventElapsedTime(&ms, e0, e1);
        }
        const double flops = (double)blocks * 8 * stages * 32.0 * 4096.0;
        unsigned long long h[2 * 512];
        hipMemcpy(h, stamps, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
        double clk = 0;
        for (int i = 0; i < blocks; ++i) clk += (double)h[2 * i] / (double)h[2 * i + 1] * 100.0;
        printf("V%d  workgroups/CU=%d  %.2f ms  %.1f TFLOP/s  clock %.0f MHz  pipe busy %.3f\n", v, blocks_per_cu, ms, flops / ms / 1e9,
               clk / blocks, (double)stages * 32 * 64 * 2 * blocks_per_cu / (double)h[0]);
      }
  return 0;
}
