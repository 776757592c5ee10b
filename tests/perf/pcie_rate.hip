// pcie_rate.hip -- what the host link gives on this box: page-locked host memory to HBM and back, by the copy engines
// (one large copy; 4 MB pieces over 12 streams; both directions at once) and by a kernel reading mapped host memory.
// Build: hipcc --offload-arch=gfx950 -O2 -o tests/perf/pcie_rate tests/perf/pcie_rate.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void k_pull(const ulonglong2* __restrict__ src, ulonglong2* __restrict__ dst, uint64_t n)
{
  for(uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += uint64_t(gridDim.x) * blockDim.x) { dst[i] = src[i]; }
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
  const size_t bytes = size_t(512) << 20, piece = size_t(4) << 20;
  char *h = nullptr, *h2 = nullptr, *d = nullptr, *d2 = nullptr;
  hipHostMalloc(reinterpret_cast<void**>(&h), bytes, hipHostMallocDefault);
  hipHostMalloc(reinterpret_cast<void**>(&h2), bytes, hipHostMallocDefault);
  hipMalloc(reinterpret_cast<void**>(&d), bytes); hipMalloc(reinterpret_cast<void**>(&d2), bytes);
  for(size_t i = 0; i < bytes; i += 4096) { h[i] = char(i); h2[i] = 1; }
  std::vector<hipStream_t> st(12);
  for(hipStream_t& s : st) { hipStreamCreateWithFlags(&s, hipStreamNonBlocking); }
  auto sync_all = [&]() { for(hipStream_t& s : st) { hipStreamSynchronize(s); } };
  for(int mode = 0; mode < 6; mode++)
  {
    double best = 1e9;
    for(int rep = 0; rep < 4; rep++)
    {
      sync_all(); hipDeviceSynchronize();
      const double t0 = now();
      size_t moved = bytes;
      switch(mode)
      {
        case 0: hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st[0]); break;
        case 1: hipMemcpyAsync(h2, d2, bytes, hipMemcpyDeviceToHost, st[0]); break;
        case 2: for(size_t o = 0, k = 0; o < bytes; o += piece, k++) { hipMemcpyAsync(d + o, h + o, piece, hipMemcpyHostToDevice, st[k % 12]); } break;
        case 3: for(size_t o = 0, k = 0; o < bytes; o += piece, k++)
                {
                  hipMemcpyAsync(d + o, h + o, piece, hipMemcpyHostToDevice, st[k % 12]);
                  hipMemcpyAsync(h2 + o, d2 + o, piece, hipMemcpyDeviceToHost, st[k % 12]);
                }
                moved = 2 * bytes; break;
        case 4: hipLaunchKernelGGL(k_pull, dim3(1024), dim3(256), 0, st[0], reinterpret_cast<const ulonglong2*>(h), reinterpret_cast<ulonglong2*>(d), bytes / 16); break;
        case 5: for(size_t o = 0, k = 0; o < bytes; o += piece, k++)
                {
                  hipLaunchKernelGGL(k_pull, dim3(64), dim3(256), 0, st[k % 12], reinterpret_cast<const ulonglong2*>(h + o), reinterpret_cast<ulonglong2*>(d + o), piece / 16);
                  hipLaunchKernelGGL(k_pull, dim3(64), dim3(256), 0, st[k % 12], reinterpret_cast<const ulonglong2*>(d2 + o), reinterpret_cast<ulonglong2*>(h2 + o), piece / 16);
                }
                moved = 2 * bytes; break;
      }
      sync_all();
      const double dt = now() - t0;
      if(dt < best) { best = dt; }
      (void)moved;
      if(rep == 3)
      {
        const char* names[6] = { "H2D, one 512 MB copy", "D2H, one 512 MB copy", "H2D, 4 MB pieces over 12 streams", "H2D + D2H, 4 MB pieces over 12 streams (sum of both)",
                                 "H2D by a kernel reading mapped host memory", "H2D + D2H by kernels, 4 MB pieces over 12 streams (sum of both)" };
        std::printf("%-66s %6.1f GB/s\n", names[mode], double(moved) / best / 1e9);
      }
    }
  }
  // the same two-way traffic by piece size and number of streams
  for(size_t mb : { size_t(1), size_t(4), size_t(16), size_t(64) })
  {
    for(size_t streams : { size_t(2), size_t(4), size_t(12) })
    {
      const size_t part = mb << 20;
      double best = 1e9;
      for(int rep = 0; rep < 3; rep++)
      {
        sync_all(); hipDeviceSynchronize();
        const double t0 = now();
        for(size_t o = 0, k = 0; o < bytes; o += part, k++)
        {
          hipMemcpyAsync(d + o, h + o, part, hipMemcpyHostToDevice, st[k % streams]);
          hipMemcpyAsync(h2 + o, d2 + o, part, hipMemcpyDeviceToHost, st[k % streams]);
        }
        sync_all();
        const double dt = now() - t0;
        if(dt < best) { best = dt; }
      }
      std::printf("H2D + D2H, %2zu MB pieces over %2zu streams (sum of both)            %6.1f GB/s\n", mb, streams, double(2 * bytes) / best / 1e9);
    }
  }
  for(size_t streams : { size_t(1), size_t(2), size_t(3), size_t(4), size_t(6), size_t(8), size_t(12) })
  {
    const size_t part = size_t(4) << 20;
    double best = 1e9;
    for(int rep = 0; rep < 3; rep++)
    {
      sync_all(); hipDeviceSynchronize();
      const double t0 = now();
      for(size_t o = 0, k = 0; o < bytes; o += part, k++) { hipMemcpyAsync(d + o, h + o, part, hipMemcpyHostToDevice, st[k % streams]); }
      sync_all();
      const double dt = now() - t0;
      if(dt < best) { best = dt; }
    }
    std::printf("H2D only, 4 MB pieces over %2zu stream(s)                             %6.1f GB/s\n", streams, double(bytes) / best / 1e9);
  }
  // uploads on their own stream(s), downloads (40 % of the volume, like find(): 40 bytes in, 16 out) on others
  for(size_t up_streams : { size_t(1), size_t(2), size_t(3), size_t(6) })
  {
    const size_t part = size_t(4) << 20, back = part * 2 / 5;
    double best = 1e9;
    for(int rep = 0; rep < 3; rep++)
    {
      sync_all(); hipDeviceSynchronize();
      const double t0 = now();
      for(size_t o = 0, k = 0; o < bytes; o += part, k++)
      {
        hipMemcpyAsync(d + o, h + o, part, hipMemcpyHostToDevice, st[k % up_streams]);
        hipMemcpyAsync(h2 + o, d2 + o, back, hipMemcpyDeviceToHost, st[6 + k % up_streams]);
      }
      sync_all();
      const double dt = now() - t0;
      if(dt < best) { best = dt; }
    }
    std::printf("H2D 4 MB pieces on %zu stream(s) + D2H 1.6 MB pieces on %zu other(s): H2D %6.1f GB/s (both %6.1f)\n", up_streams, up_streams,
                double(bytes) / best / 1e9, double(bytes) * 1.4 / best / 1e9);
  }
  return 0;
}
