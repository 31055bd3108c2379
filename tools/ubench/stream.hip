// HBM read-rate microbenchmark (round 4, the wgrad_kernel question: what does this box deliver to a kernel that only loads?)
//   A: every workgroup streams a contiguous slice (dwordx4 per lane, 8 loads in flight per thread)
//   B: the wgrad pattern -- W workgroups, each walking its own slice in steps of `stride` bytes, reading `chunk` contiguous bytes
//      per step with two dwordx4 per thread, ONE step in flight (fetch -> use -> barrier), like a k-step of wgrad_kernel
//   C: B with two steps in flight
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/stream tools/ubench/stream.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) stream_a(const f32x4* __restrict__ p, size_t n16, float* out) {
  f32x4 acc = {0, 0, 0, 0};
  const size_t per = n16 / gridDim.x;
  const f32x4* q = p + (size_t)blockIdx.x * per;
  for (size_t i = threadIdx.x; i + 7 * 256 < per; i += 8 * 256) {
    f32x4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = q[i + k * 256];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += v[k];
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = 1.0f;
}
template <int DEPTH, int THREADS>
__global__ void __launch_bounds__(THREADS) stream_b(const f32x4* __restrict__ p, size_t slice16, int steps, size_t stride16, int per_thread, float* out) {
  __shared__ float sink[THREADS];
  f32x4 acc = {0, 0, 0, 0};
  const f32x4* q = p + (size_t)blockIdx.x * slice16;
  f32x4 pre[DEPTH][8];
  for (int d = 0; d < DEPTH; ++d)
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k < per_thread) pre[d][k] = q[(size_t)d * stride16 + threadIdx.x + k * THREADS];
  for (int s = 0; s < steps; s += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (k < per_thread) acc += pre[d][k];
      const int nx = s + d + DEPTH < steps ? s + d + DEPTH : steps - 1;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (k < per_thread) pre[d][k] = q[(size_t)nx * stride16 + threadIdx.x + k * THREADS];
      sink[threadIdx.x] = acc[0];
      __syncthreads();
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] + sink[0] == 12345.678f) out[0] = 1.0f;
}
int main() {
  const size_t bytes = (size_t)1300 << 20;
  void* buf; float* out;
  hipMalloc(&buf, bytes); hipMalloc(&out, 64);
  hipMemset(buf, 0, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto time = [&](const char* name, double moved, auto launch) {
    for (int w = 0; w < 3; ++w) launch();
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-70s %8.1f us  %6.2f TB/s\n", name, ms * 100, moved / (ms * 1e-4) / 1e12);
  };
  for (int g : {256, 512, 1024, 2048, 4096})
    time((std::string("A contiguous slices, grid ") + std::to_string(g)).c_str(), (double)bytes,
         [&] { hipLaunchKernelGGL(stream_a, dim3(g), dim3(256), 0, 0, (const f32x4*)buf, bytes / 16, out); });
  // B/C: W workgroups; per step `per_thread` x THREADS x 16 bytes contiguous; stride = the same (dense walk)
  for (int W : {256, 480, 512, 960}) {
    for (int pt : {4, 8}) {
      const size_t step_bytes = (size_t)pt * 512 * 16;
      const size_t slice = bytes / W / step_bytes * step_bytes;
      const int steps = (int)(slice / step_bytes);
      char nm[128];
      snprintf(nm, sizeof nm, "B 512 thr, W=%d, %zu KB per step, 1 step in flight", W, step_bytes >> 10);
      time(nm, (double)slice * W, [&] { hipLaunchKernelGGL((stream_b<1, 512>), dim3(W), dim3(512), 0, 0, (const f32x4*)buf, slice / 16, steps, step_bytes / 16, pt, out); });
      snprintf(nm, sizeof nm, "C 512 thr, W=%d, %zu KB per step, 2 steps in flight", W, step_bytes >> 10);
      time(nm, (double)slice * W, [&] { hipLaunchKernelGGL((stream_b<2, 512>), dim3(W), dim3(512), 0, 0, (const f32x4*)buf, slice / 16, steps, step_bytes / 16, pt, out); });
    }
  }
  return 0;
}
