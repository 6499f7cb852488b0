// launch_floor.hip - what does ONE dependent launch cost on this GPU, and what does the step kernel's
// HBM traffic (48 rows in, 36 rows out per 64-environment tile) cost as a plain copy through LDS?
//   K0: empty kernel, same grid/block/LDS as step_kernel (balance @ 32768 envs)
//   K1: K0 + one coalesced row read per wave, no use
//   K2: tile copy: 60 rows HBM -> LDS, barrier, 36 rows LDS -> HBM (the step's algorithmic traffic)
//   K3: K2 with 2 barriers and a dependent LDS round trip between them
// build: hipcc --offload-arch=gfx950 -O3 -o launch_floor launch_floor.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k0(float* s, long ld) { extern __shared__ float lds[]; }
__global__ void k1(float* s, long ld) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long env = (long)blockIdx.x * 64 + lane;
  float v = s[(long)wv * ld + env];
  if (v == 123456.f) lds[0] = v;
}
template <int MODE>
__global__ void k2(float* __restrict__ s, float* __restrict__ ft, long ld, int n_in, int n_ft, int n_out) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const long env = (long)blockIdx.x * 64 + lane;
  float v[8];
  int cnt = 0;
  for (int r = wv; r < n_in; r += nw) v[cnt++] = s[(long)r * ld + env];
  float f[2];
  int cf = 0;
  for (int r = wv; r < n_ft; r += nw) f[cf++] = ft[(long)r * ld + env];
  cnt = 0;
  for (int r = wv; r < n_in; r += nw) lds[r * 64 + lane] = v[cnt++];
  cf = 0;
  for (int r = wv; r < n_ft; r += nw) lds[(n_in + r) * 64 + lane] = f[cf++];
  __syncthreads();
  if (MODE == 3) {
    float a = lds[((wv + 1) % n_in) * 64 + lane] + lds[(n_in + (wv % n_ft)) * 64 + lane];
    lds[(n_in + n_ft + wv) * 64 + lane] = a;
    __syncthreads();
    lds[wv * 64 + lane] += lds[(n_in + n_ft + ((wv + 3) % nw)) * 64 + lane] * 1e-9f;
    __syncthreads();
  }
  for (int r = wv; r < n_out; r += nw) s[(long)(n_in - n_out + r) * ld + env] = lds[(n_in - n_out + r) * 64 + lane] + 1e-7f;
}

// shader clock against the constant 100 MHz counter: what frequency do kernels of this process run at (a profiler may pin
// the device to a stable, lower power state - then its kernel durations cannot agree with a free-running process's)
__global__ void clock_probe(unsigned long long* out, int iters) {
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float x = (float)threadIdx.x;
  for (int i = 0; i < iters; ++i) x = x * 1.000001f + 0.5f;
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; out[2] = (unsigned long long)x; }
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 32768;
  const int nw = argc > 2 ? atoi(argv[2]) : 8;
  const long ld = B;
  const int n_in = 48, n_ft = 12, n_out = 36;
  float *s, *ft;
  CK(hipMalloc(&s, sizeof(float) * ld * n_in));
  CK(hipMalloc(&ft, sizeof(float) * ld * n_ft));
  CK(hipMemset(s, 0, sizeof(float) * ld * n_in));
  CK(hipMemset(ft, 0, sizeof(float) * ld * n_ft));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int tiles = B / 64, N = 5000;
  const size_t lds = 35 * 1024;
  auto time = [&](const char* name, auto launch) {
    for (int i = 0; i < 200; ++i) launch();
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < N; ++i) launch();
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %7.2f us/launch\n", name, ms * 1e3 / N);
  };
  printf("B=%d tiles=%d waves/tile=%d\n", B, tiles, nw);
  time("K0 empty", [&] { hipLaunchKernelGGL(k0, dim3(tiles), dim3(64 * nw), lds, st, s, ld); });
  time("K0 empty, 1 wave/tile", [&] { hipLaunchKernelGGL(k0, dim3(tiles), dim3(64), lds, st, s, ld); });
  time("K0 empty, no LDS", [&] { hipLaunchKernelGGL(k0, dim3(tiles), dim3(64 * nw), 0, st, s, ld); });
  time("K1 one row read per wave", [&] { hipLaunchKernelGGL(k1, dim3(tiles), dim3(64 * nw), lds, st, s, ld); });
  time("K2 tile copy 48+12 rows in, 36 out", [&] { hipLaunchKernelGGL(k2<2>, dim3(tiles), dim3(64 * nw), lds, st, s, ft, ld, n_in, n_ft, n_out); });
  time("K3 = K2 + 2 more barriers/LDS round trips", [&] { hipLaunchKernelGGL(k2<3>, dim3(tiles), dim3(64 * nw), lds, st, s, ft, ld, n_in, n_ft, n_out); });
  // the same through a captured graph of 100 launches
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k2<2>, dim3(tiles), dim3(64 * nw), lds, st, s, ft, ld, n_in, n_ft, n_out);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < N / 100; ++i) CK(hipGraphLaunch(ge, st));
  CK(hipEventRecord(e1, st));
  CK(hipStreamSynchronize(st));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-44s %7.2f us/launch\n", "K2 as a 100-node HIP graph", ms * 1e3 / N);
  // the batch cut in NQ parts on NQ streams, 100 steps each, captured as ONE graph (fork / join by events): what
  // vmas_world_step_n over several queues would cost without the host's NQ enqueues per step; and the same eagerly
  for (int NQ = 2; NQ <= 4; NQ += 2) {
    hipStream_t sq[4]; sq[0] = st;
    hipEvent_t fork, join[4];
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    for (int q = 1; q < NQ; ++q) { CK(hipStreamCreate(&sq[q])); CK(hipEventCreateWithFlags(&join[q], hipEventDisableTiming)); }
    auto first = [&](int q) { return tiles * q / NQ; };
    auto enqueue = [&](int n_steps) {
      CK(hipEventRecord(fork, st));
      for (int q = 1; q < NQ; ++q) CK(hipStreamWaitEvent(sq[q], fork, 0));
      for (int i = 0; i < n_steps; ++i)
        for (int q = 0; q < NQ; ++q)
          hipLaunchKernelGGL(k2<2>, dim3(first(q + 1) - first(q)), dim3(64 * nw), lds, sq[q], s + first(q) * 64, ft + first(q) * 64, ld, n_in, n_ft, n_out);
      for (int q = 1; q < NQ; ++q) { CK(hipEventRecord(join[q], sq[q])); CK(hipStreamWaitEvent(st, join[q], 0)); }
    };
    hipGraph_t g2; hipGraphExec_t ge2;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    enqueue(100);
    CK(hipStreamEndCapture(st, &g2));
    CK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge2, st)); CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < N / 100; ++i) CK(hipGraphLaunch(ge2, st));
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("K2 in %d parts on %d streams, one %d-node graph   %7.2f us/step\n", NQ, NQ, 100 * NQ, ms * 1e3 / N);
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, st));
      enqueue(N);
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      CK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("K2 in %d parts on %d streams, eager                %7.2f us/step\n", NQ, NQ, ms * 1e3 / N);
  }
  // two 64-environment tiles per block (1024 threads): half as many workgroups to dispatch
  if (nw == 8) {
    time("K0 empty, 256 blocks x 1024 threads", [&] { hipLaunchKernelGGL(k0, dim3(tiles / 2), dim3(1024), 2 * lds, st, s, ld); });
    time("K0 empty, 1024 blocks x 256 threads", [&] { hipLaunchKernelGGL(k0, dim3(tiles * 2), dim3(256), lds / 2, st, s, ld); });
  }
  {
    unsigned long long* d; CK(hipMalloc(&d, 64));
    unsigned long long h[3];
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(clock_probe, dim3(1), dim3(64), 0, st, d, 2000000);
      CK(hipStreamSynchronize(st));
      CK(hipMemcpy(h, d, 24, hipMemcpyDeviceToHost));
      printf("clock probe: %llu shader cycles in %llu ticks of the 100 MHz counter -> %.0f MHz\n", h[0], h[1], (double)h[0] / ((double)h[1] / 100.0));
    }
  }
  return 0;
}
