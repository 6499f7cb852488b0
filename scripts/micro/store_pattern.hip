// store_pattern.hip - what does HBM take football's observation tensor at, depending on how a wave's store instructions
// cover it?  obs = [A = 10][batch][D = 88] floats: for one agent, the rows of a 64-environment tile are ONE contiguous
// 22 528-byte run.  The same bytes, written by the same grid (one block of 8 waves per tile), four ways:
//   P0: every lane stores its own row 16 bytes at a time (stride 352 B between the lanes of an instruction)
//   P1: 16-column chunks - the lanes of an instruction cover 64-byte pieces of 16 rows     (the step kernel's epilogue)
//   P2: 32-column chunks - 128-byte pieces of 8 rows                                        (the stand-alone kernel)
//   P3: the tile's run as it lies - an instruction covers 1 024 contiguous bytes, the block's waves share an agent
//   P4: as P3, but each wave writes whole agents alone (22 instructions in a row on one run)
// build: hipcc --offload-arch=gfx950 -O3 -o store_pattern store_pattern.hip ; run: ./store_pattern [envs] [reps]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int A = 10, D = 88;

template <int P>
__global__ void __launch_bounds__(512) writer(float* __restrict__ obs, long batch, float seed) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const long b0 = (long)blockIdx.x * 64;
  const float4 val = make_float4(seed + lane, seed + wv, seed, (float)blockIdx.x);
  if (P == 3) {
    for (int a = 0; a < A; ++a) {
      float4* run = (float4*)(obs + ((long)a * batch + b0) * D);
      for (int i = threadIdx.x; i < 64 * D / 4; i += blockDim.x) run[i] = val;
    }
    return;
  }
  for (int a = wv; a < A; a += nw) {
    float* out = obs + ((long)a * batch + b0) * D;
    if (P == 0) {
      float* row = out + (long)lane * D;
      for (int c = 0; c < D; c += 4) *(float4*)(row + c) = val;
    } else if (P == 4) {
      float4* run = (float4*)out;
      for (int i = lane; i < 64 * D / 4; i += 64) run[i] = val;
    } else {
      const int chunk = P == 1 ? 16 : 32;
      for (int c0 = 0; c0 < D; c0 += chunk) {
        const int w = D - c0 < chunk ? D - c0 : chunk, w4 = w >> 2, total4 = 64 * w4;
        for (int i = lane; i < total4; i += 64) {
          const int r = i / w4, c4 = i - r * w4;
          *(float4*)(out + r * D + c0 + 4 * c4) = val;
        }
      }
    }
  }
}

template <int P>
static void run(float* obs, long batch, int reps, const char* what) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int blocks = (int)(batch / 64);
  for (int i = 0; i < 3; ++i) writer<P><<<blocks, 512>>>(obs, batch, (float)i);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) writer<P><<<blocks, 512>>>(obs, batch, (float)i);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps, bytes = (double)A * batch * D * 4;
  printf("{\"pattern\": \"%s\", \"envs\": %ld, \"us\": %.2f, \"TBps\": %.3f}\n", what, batch, us, bytes / us * 1e-6);
}

int main(int argc, char** argv) {
  const long batch = argc > 1 ? atol(argv[1]) : 131072;
  const int reps = argc > 2 ? atoi(argv[2]) : 50;
  float* obs;
  CK(hipMalloc(&obs, (size_t)A * batch * D * 4));
  run<0>(obs, batch, reps, "P0 lane rows, 16 B per lane");
  run<1>(obs, batch, reps, "P1 16-column chunks (64 B pieces)");
  run<2>(obs, batch, reps, "P2 32-column chunks (128 B pieces)");
  run<3>(obs, batch, reps, "P3 contiguous run, waves share an agent");
  run<4>(obs, batch, reps, "P4 contiguous run, one wave per agent");
  run<1>(obs, batch, reps, "P1 again");
  return 0;
}
