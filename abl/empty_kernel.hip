// What a kernel's duration is under rocprofv3 when the kernel does nothing: the dispatch-timestamp floor that every per-kernel average
// of profiles/*_kernel_stats.csv contains (Σ of the metric step's rocprof averages exceeds the unprofiled step).
//   hipcc --offload-arch=gfx950 -O2 abl/empty_kernel.hip -o gpurun_out/empty_kernel
//   rocprofv3 --kernel-trace --stats -d gpurun_out/prof_empty -- gpurun_out/empty_kernel        (abl/empty_kernel.sh)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void k_empty() {}
__global__ void __launch_bounds__(256) k_empty_256x256(int* p) {
  if (p && threadIdx.x == 1024) p[0] = 0;   // never true: 256 workgroups of 256 threads that do nothing
}

int main() {
  hipStream_t s;
  (void)hipStreamCreate(&s);
  for (int i = 0; i < 200; i++) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s);
  (void)hipStreamSynchronize(s);
  const int N = 2000;
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s);
  (void)hipStreamSynchronize(s);
  auto t1 = std::chrono::steady_clock::now();
  for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_empty_256x256, dim3(256), dim3(256), 0, s, (int*)nullptr);
  (void)hipStreamSynchronize(s);
  auto t2 = std::chrono::steady_clock::now();
  std::printf("{\"empty_1x64_us_per_launch_back_to_back\": %.3f, \"empty_256x256_us_per_launch_back_to_back\": %.3f}\n",
              std::chrono::duration<double, std::micro>(t1 - t0).count() / N, std::chrono::duration<double, std::micro>(t2 - t1).count() / N);
  return 0;
}
