// pool_readback_repro.hip -- minimal reproducer for the hazard noted in gcsa2_hip.hip (struct gcsa2_index, d_slots):
// "device-to-host copies of small results out of stream-ordered pool memory (hipMallocAsync) were observed to return
// stale data about once in 5000 calls".  Mimics the locate pipeline's read-back: allocate scratch from the pool, let
// kernel A zero and fill five totals, kernel B add to them, copy the 40 bytes to the host, synchronise, compare with
// the value the iteration must produce, free the scratch.  Variants separate the suspects:
//   pool / pageable   hipMallocAsync scratch, destination on the stack      (the round-1 code path)
//   pool / pinned     hipMallocAsync scratch, destination in hipHostMalloc memory
//   plain / pageable  hipMalloc scratch (allocated once), destination on the stack   (the workaround that was shipped)
// each on the null stream, a blocking stream and a non-blocking stream.
//   hipcc --offload-arch=gfx950 -O2 -o pool_readback_repro tools/hip/pool_readback_repro.hip && ./pool_readback_repro [iterations]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CHECK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(2); } } while(0)

__global__ void k_fill(unsigned long long* totals, unsigned long long* big, unsigned long long n, unsigned long long iter)
{
  unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x;
  if(i == 0) { for(int k = 0; k < 5; k++) { totals[k] = 0; } }
  if(i < n) { big[i] = iter + i; }
}

__global__ void k_add(unsigned long long* totals, const unsigned long long* big, unsigned long long n, unsigned long long iter)
{
  unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x;
  if(i < n && big[i] == iter + i) { atomicAdd(totals + (i % 5), iter + 1); }
}

static unsigned long long run(const char* name, bool pool, bool pinned, hipStream_t stream, unsigned long long iterations)
{
  const unsigned long long n = 4096;
  unsigned long long *plain_totals = nullptr, *plain_big = nullptr, *pinned_dst = nullptr;
  if(!pool) { CHECK(hipMalloc(&plain_totals, 5 * 8)); CHECK(hipMalloc(&plain_big, n * 8)); }
  if(pinned) { CHECK(hipHostMalloc(reinterpret_cast<void**>(&pinned_dst), 5 * 8, hipHostMallocDefault)); }
  unsigned long long bad = 0;
  for(unsigned long long it = 0; it < iterations; it++)
  {
    unsigned long long *totals = plain_totals, *big = plain_big, *other = nullptr;
    if(pool)
    {
      CHECK(hipMallocAsync(reinterpret_cast<void**>(&big), n * 8, stream));
      CHECK(hipMallocAsync(reinterpret_cast<void**>(&other), (1 + it % 7) * 4096, stream));    // varying neighbours, as in the pipeline
      CHECK(hipMallocAsync(reinterpret_cast<void**>(&totals), 5 * 8, stream));
    }
    hipLaunchKernelGGL(k_fill, dim3(n / 256), dim3(256), 0, stream, totals, big, n, it);
    hipLaunchKernelGGL(k_add, dim3(n / 256), dim3(256), 0, stream, totals, big, n, it);
    unsigned long long stack_dst[5] = {~0ull, ~0ull, ~0ull, ~0ull, ~0ull};
    unsigned long long* dst = pinned ? pinned_dst : stack_dst;
    if(pinned) { for(int k = 0; k < 5; k++) { dst[k] = ~0ull; } }
    CHECK(hipMemcpyAsync(dst, totals, 5 * 8, hipMemcpyDeviceToHost, stream));
    CHECK(hipStreamSynchronize(stream));
    for(int k = 0; k < 5; k++)
    {
      const unsigned long long members = n / 5 + (k < int(n % 5) ? 1 : 0);
      if(dst[k] != members * (it + 1))
      {
        if(bad < 5) { std::printf("  %s: iteration %llu total[%d] = %llu, expected %llu\n", name, it, k, dst[k], members * (it + 1)); }
        bad++;
        break;
      }
    }
    if(pool) { CHECK(hipFreeAsync(totals, stream)); CHECK(hipFreeAsync(other, stream)); CHECK(hipFreeAsync(big, stream)); }
  }
  CHECK(hipStreamSynchronize(stream));
  if(!pool) { CHECK(hipFree(plain_totals)); CHECK(hipFree(plain_big)); }
  if(pinned) { CHECK(hipHostFree(pinned_dst)); }
  std::printf("%-46s %llu stale read-backs in %llu iterations\n", name, bad, iterations);
  return bad;
}

int main(int argc, char** argv)
{
  const unsigned long long iterations = (argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 100000);
  hipStream_t blocking = nullptr, nonblocking = nullptr;
  CHECK(hipStreamCreate(&blocking));
  CHECK(hipStreamCreateWithFlags(&nonblocking, hipStreamNonBlocking));
  struct { const char* name; hipStream_t s; } streams[3] = { {"null stream", nullptr}, {"blocking stream", blocking}, {"non-blocking stream", nonblocking} };
  unsigned long long total = 0;
  for(auto& st : streams)
  {
    char name[128];
    std::snprintf(name, sizeof(name), "pool / pageable, %s", st.name);  total += run(name, true, false, st.s, iterations);
    std::snprintf(name, sizeof(name), "pool / pinned, %s", st.name);    total += run(name, true, true, st.s, iterations);
    std::snprintf(name, sizeof(name), "plain / pageable, %s", st.name); total += run(name, false, false, st.s, iterations);
  }
  // the engine's setting: default pool with the release threshold raised to "keep everything"
  hipMemPool_t pool = nullptr;
  int dev = 0;
  CHECK(hipGetDevice(&dev));
  if(hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess && pool != nullptr)
  {
    unsigned long long keep = ~0ull;
    CHECK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep));
    total += run("pool (threshold = max) / pageable, null stream", true, false, nullptr, iterations);
    total += run("pool (threshold = max) / pageable, non-blocking", true, false, nonblocking, iterations);
  }
  std::printf("total stale read-backs: %llu\n", total);
  return total == 0 ? 0 : 1;
}
