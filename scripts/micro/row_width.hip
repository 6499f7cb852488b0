// row_width.hip - the bandwidth regime (round-5 review, item 8): does the WIDTH of a lane's global access matter for the step's
// traffic?  The step kernel moves a 64-environment tile row by row, one dword per lane (a wave instruction = one 256-byte
// segment of one row).  Here the same traffic (48 state rows + 12 force rows in, 36 state rows out per tile, rows `ld` floats
// apart) as a tile copy through LDS in three widths: dword (64 lanes x 4 B = one row per instruction), dwordx2 (32 lanes per
// row: two rows per instruction), dwordx4 (16 lanes per row: four rows per instruction; the LDS tile keeps its [row][64] form).
// build: hipcc --offload-arch=gfx950 -O3 -o row_width row_width.hip        run: ./row_width [envs ...]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int W> struct Vec;
template <> struct Vec<1> { using T = float; };
template <> struct Vec<2> { using T = float2; };
template <> struct Vec<4> { using T = float4; };

template <int W>
__global__ __launch_bounds__(512) void copy_tile(float* __restrict__ s, const float* __restrict__ ft, long ld, int n_in, int n_ft, int n_out) {
  using V = typename Vec<W>::T;
  extern __shared__ float lds[];
  constexpr int LPR = 64 / W;           // lanes per row
  constexpr int RPI = W;                // rows per wave instruction
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int sub = lane / LPR, e0 = (lane % LPR) * W;
  const long base = (long)blockIdx.x * 64 + e0;
  V v[8];
  int c = 0;
  for (int r = (wv * RPI) + sub; r < n_in; r += nw * RPI) v[c++] = *(const V*)(s + (long)r * ld + base);
  V f[2];
  int cf = 0;
  for (int r = (wv * RPI) + sub; r < n_ft; r += nw * RPI) f[cf++] = *(const V*)(ft + (long)r * ld + base);
  c = 0;
  for (int r = (wv * RPI) + sub; r < n_in; r += nw * RPI) *(V*)(lds + r * 64 + e0) = v[c++];
  cf = 0;
  for (int r = (wv * RPI) + sub; r < n_ft; r += nw * RPI) *(V*)(lds + (n_in + r) * 64 + e0) = f[cf++];
  __syncthreads();
  for (int r = (wv * RPI) + sub; r < n_out; r += nw * RPI) {
    V o = *(const V*)(lds + (n_in - n_out + r) * 64 + e0);
    float* p = (float*)&o;
    for (int k = 0; k < W; ++k) p[k] += 1e-7f;
    *(V*)(s + (long)(n_in - n_out + r) * ld + base) = o;
  }
}

int main(int argc, char** argv) {
  const int n_in = 48, n_ft = 12, n_out = 36, nw = 8;
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int sizes_default[] = {32768, 262144, 1048576};
  for (int a = 0; a < (argc > 1 ? argc - 1 : 3); ++a) {
    const int B = argc > 1 ? atoi(argv[a + 1]) : sizes_default[a];
    const long ld = B;
    float *s, *ft;
    CK(hipMalloc(&s, sizeof(float) * ld * n_in)); CK(hipMalloc(&ft, sizeof(float) * ld * n_ft));
    CK(hipMemset(s, 0, sizeof(float) * ld * n_in)); CK(hipMemset(ft, 0, sizeof(float) * ld * n_ft));
    const int tiles = B / 64, N = B > 100000 ? 500 : 3000;
    const size_t lds = 35 * 1024;
    const double bytes = (double)B * 4 * (n_in + n_ft + n_out);
    auto time = [&](const char* name, auto launch) {
      for (int rep = 0; rep < 2; ++rep) {
        for (int i = 0; i < 100; ++i) launch();
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < N; ++i) launch();
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / N;
        printf("envs %8d  %-28s %8.2f us/launch  %6.0f GB/s = %.3f of 8 TB/s\n", B, name, us, bytes / us / 1e3, bytes / us / 1e3 / 8000.0);
      }
    };
    time("dword   (one row / instr)", [&] { hipLaunchKernelGGL(copy_tile<1>, dim3(tiles), dim3(64 * nw), lds, st, s, ft, ld, n_in, n_ft, n_out); });
    time("dwordx2 (two rows / instr)", [&] { hipLaunchKernelGGL(copy_tile<2>, dim3(tiles), dim3(64 * nw), lds, st, s, ft, ld, n_in, n_ft, n_out); });
    time("dwordx4 (four rows / instr)", [&] { hipLaunchKernelGGL(copy_tile<4>, dim3(tiles), dim3(64 * nw), lds, st, s, ft, ld, n_in, n_ft, n_out); });
    CK(hipFree(s)); CK(hipFree(ft));
  }
  return 0;
}
