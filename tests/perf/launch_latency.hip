// launch_latency.hip -- what one small host-pointer call can cost at best on this box: an (almost) empty kernel on a
// non-blocking stream, completed in four ways.  Build: hipcc --offload-arch=gfx950 -O2 -o /tmp/launch_latency tests/perf/launch_latency.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>

__global__ void k_touch(const uint64_t* in, uint64_t* out) { out[threadIdx.x] = in[threadIdx.x] + 1; }
__global__ void k_touch_flag(const uint64_t* in, uint64_t* out, volatile uint64_t* flag, uint64_t ticket)
{
  out[threadIdx.x] = in[threadIdx.x] + 1;
  __threadfence_system();
  __syncthreads();
  if(threadIdx.x == 0) { *flag = ticket; }
}
__global__ void k_flag(volatile uint64_t* flag, uint64_t ticket) { *flag = ticket; }

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
  hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  uint64_t *z = nullptr, *d = nullptr;
  hipHostMalloc(reinterpret_cast<void**>(&z), 4096, hipHostMallocMapped | hipHostMallocCoherent);
  hipMalloc(reinterpret_cast<void**>(&d), 4096);
  for(int i = 0; i < 512; i++) { z[i] = 0; }
  volatile uint64_t* flag = z + 256;
  const int reps = 20000;
  for(int mode = 0; mode < 5; mode++)
  {
    double t0 = 0;
    for(int r = -1000; r < reps; r++)
    {
      if(r == 0) { t0 = now(); }
      const uint64_t ticket = uint64_t(mode) * 1000000 + uint64_t(r + 2000);
      switch(mode)
      {
        case 0: hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, st, z, z + 64); hipStreamSynchronize(st); break;
        case 1: hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, st, z, z + 64); while(hipStreamQuery(st) == hipErrorNotReady) {} break;
        case 2: hipLaunchKernelGGL(k_touch_flag, dim3(1), dim3(64), 0, st, z, z + 64, flag, ticket); while(*flag != ticket) {} break;
        case 3: hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, st, z, z + 64); hipLaunchKernelGGL(k_flag, dim3(1), dim3(1), 0, st, flag, ticket);
                while(*flag != ticket) {} break;
        case 4: hipMemcpyAsync(d, z, 64, hipMemcpyHostToDevice, st); hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, st, d, d + 64);
                hipMemcpyAsync(z + 64, d + 64, 64, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); break;
      }
    }
    const double us = (now() - t0) / reps * 1e6;
    const char* names[5] = { "zero-copy kernel + hipStreamSynchronize", "zero-copy kernel + hipStreamQuery spin", "kernel writes a flag, host spins on it",
                             "kernel, then flag kernel, host spins", "copy in + kernel + copy out + hipStreamSynchronize" };
    std::printf("%-52s %7.2f us per call\n", names[mode], us);
  }
  hipStreamSynchronize(st);
  return 0;
}
