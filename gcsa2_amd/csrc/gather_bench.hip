// gather_bench.hip -- microbenchmark: the realistic ceiling for the find() access pattern.
//
// Every query of k_find is a chain of DEPENDENT random fetches of 64-byte rank blocks.  This tool
// measures how many such fetches per second MI355X sustains when the blocks live in L2, in the
// Infinity Cache or in HBM, for several ways of issuing the fetch:
//
//   lane64   one lane fetches its whole 64-B block with 4 x global_load_dwordx4  (= k_find v1)
//   lane16   one lane fetches only 16 B of the block (lower bound on per-lane issue cost)
//   lane8x2  one lane fetches the counter word + one payload word (2 x 8 B in one line)
//   quad     4 lanes share a chain: each fetches 16 B of the same block, DPP-style reduction
//   lds      the wave fetches its 64 blocks as 4 line-coalesced quad instructions, stages them in
//            LDS, and every lane reads its own block back with ds_read_b128
//   lane128  one lane fetches a 128-B block (8 x dwordx4)
//
// The next block index depends on the fetched data, so chains cannot be overlapped by the
// hardware; parallelism comes only from resident lanes, exactly as in k_find.
//
//   gather_bench [log2_bytes ...]                  prints one line per (size, mode)
//   gather_bench --mode lds128 [log2_bytes ...]    that mode only (bench.py: the request-rate ceiling of THIS box)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

typedef uint64_t u64;
typedef uint32_t u32;

#define CHECK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while(0)

__device__ __forceinline__ u64 mix(u64 z)
{
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__global__ void k_fill(u64* buf, u64 words)
{
  u64 i = u64(blockIdx.x) * blockDim.x + threadIdx.x;
  u64 stride = u64(gridDim.x) * blockDim.x;
  for(; i < words; i += stride) { buf[i] = mix(i + 0x9E3779B97F4A7C15ull); }
}

constexpr int TPB = 256;

template<int MODE>
__global__ __launch_bounds__(TPB) void k_chase(const u64* __restrict__ buf, u64 nblocks, int steps, u64* __restrict__ out)
{
  __shared__ ulonglong2 stage[MODE == 4 ? TPB * 4 : 1];    // 64 B per lane (lds mode)
  __shared__ ulonglong2 stage2[MODE == 6 ? TPB * 8 : 1];   // 128 B per lane (lds128 mode)
  u64 tid = u64(blockIdx.x) * TPB + threadIdx.x;
  u32 lane = threadIdx.x & 63;
  u64 chain = (MODE == 3 ? tid >> 2 : tid);
  u64 idx = mix(chain * 2654435761ull + 12345) % nblocks;
  u64 acc = 0;
  for(int s = 0; s < steps; s++)
  {
    u64 v = 0;
    if(MODE == 0)        // lane64
    {
      const ulonglong2* p = reinterpret_cast<const ulonglong2*>(buf + idx * 8);
      ulonglong2 a = p[0], b = p[1], c = p[2], d = p[3];
      v = a.x + __popcll(a.y) + __popcll(b.x) + __popcll(b.y) + __popcll(c.x) + __popcll(c.y) + __popcll(d.x) + __popcll(d.y);
    }
    else if(MODE == 1)   // lane16
    {
      ulonglong2 a = *reinterpret_cast<const ulonglong2*>(buf + idx * 8);
      v = a.x + __popcll(a.y);
    }
    else if(MODE == 2)   // lane8x2
    {
      u64 a = buf[idx * 8];
      u64 b = buf[idx * 8 + 1 + (a % 7)];
      v = a + __popcll(b);
    }
    else if(MODE == 3)   // quad: lanes 4k..4k+3 share idx
    {
      ulonglong2 a = reinterpret_cast<const ulonglong2*>(buf + idx * 8)[lane & 3];
      u64 part = ((lane & 3) == 0 ? a.x : __popcll(a.x)) + __popcll(a.y);
      part += __shfl_xor(part, 1, 64);
      part += __shfl_xor(part, 2, 64);
      v = part;
    }
    else if(MODE == 4)   // lds-staged cooperative fetch
    {
      u32 wave_base = (threadIdx.x & ~63u) * 4;
#pragma unroll
      for(int j = 0; j < 4; j++)
      {
        // instruction j fetches the blocks of lanes 16j .. 16j+15, 4 lanes per block
        u32 owner = 16 * j + (lane >> 2);
        u64 oidx = __shfl(idx, owner, 64);
        ulonglong2 a = reinterpret_cast<const ulonglong2*>(buf + oidx * 8)[lane & 3];
        stage[wave_base + owner * 4 + (lane & 3)] = a;
      }
      __builtin_amdgcn_wave_barrier();
      const ulonglong2* p = stage + wave_base + lane * 4;
      ulonglong2 a = p[0], b = p[1], c = p[2], d = p[3];
      v = a.x + __popcll(a.y) + __popcll(b.x) + __popcll(b.y) + __popcll(c.x) + __popcll(c.y) + __popcll(d.x) + __popcll(d.y);
      __builtin_amdgcn_wave_barrier();
    }
    else if(MODE == 6)   // lds128: 128-B blocks, 8 lanes per block, 8 line-coalesced instructions
    {
      u32 wave_base = (threadIdx.x & ~63u) * 8;
#pragma unroll
      for(int j = 0; j < 8; j++)
      {
        u32 owner = 8 * j + (lane >> 3);
        u64 oidx = __shfl(idx, owner, 64);
        ulonglong2 a = reinterpret_cast<const ulonglong2*>(buf + (oidx >> 1) * 16)[lane & 7];
        stage2[wave_base + owner * 8 + (lane & 7)] = a;
      }
      __builtin_amdgcn_wave_barrier();
      const ulonglong2* p = stage2 + wave_base + lane * 8;
      u64 t = 0;
#pragma unroll
      for(int j = 0; j < 8; j++) { ulonglong2 a = p[j]; t += __popcll(a.x) + __popcll(a.y); }
      v = p[0].x + t;
      __builtin_amdgcn_wave_barrier();
    }
    else                 // lane128
    {
      const ulonglong2* p = reinterpret_cast<const ulonglong2*>(buf + (idx >> 1) * 16);
      u64 t = 0;
#pragma unroll
      for(int j = 0; j < 8; j++) { ulonglong2 a = p[j]; t += __popcll(a.x) + __popcll(a.y); }
      v = p[0].x + t;
    }
    acc += v;
    idx = mix(v + s) % nblocks;
  }
  if(acc == 0x1234567) { out[0] = acc; }   // keep the chain alive
}

template<int MODE>
double run(const u64* buf, u64 nblocks, u64 chains, int steps, u64* out, int reps)
{
  u64 threads = (MODE == 3 ? chains * 4 : chains);
  dim3 grid((threads + TPB - 1) / TPB), block(TPB);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_chase<MODE>, grid, block, 0, 0, buf, nblocks, steps, out);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for(int r = 0; r < reps; r++) { hipLaunchKernelGGL(k_chase<MODE>, grid, block, 0, 0, buf, nblocks, steps, out); }
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return double(chains) * steps * reps / (ms * 1e-3);
}

int main(int argc, char** argv)
{
  std::vector<int> sizes;
  const char* names[7] = {"lane64", "lane16", "lane8x2", "quad", "lds", "lane128", "lds128"};
  int only = -1;
  for(int i = 1; i < argc; i++)
  {
    if(std::string(argv[i]) == "--mode" && i + 1 < argc)
    {
      for(int m = 0; m < 7; m++) { if(std::string(argv[i + 1]) == names[m]) { only = m; } }
      i++;
    }
    else { sizes.push_back(atoi(argv[i])); }
  }
  if(sizes.empty()) { sizes = {21, 27, 33}; }   // 2 MB (L2), 128 MB (Infinity Cache), 8 GB (HBM)
  u64* out; CHECK(hipMalloc(&out, 64));
  printf("%-10s %-8s %12s %12s %10s\n", "bytes", "mode", "Gfetch/s", "GB/s(64B)", "chains");
  for(int lg : sizes)
  {
    u64 bytes = u64(1) << lg, words = bytes / 8, nblocks = bytes / 64;
    u64* buf; CHECK(hipMalloc(&buf, bytes));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, buf, words);
    CHECK(hipDeviceSynchronize());
    const int steps = 64;
    for(u64 chains : {u64(1) << 22})
    {
      double r[7] = {0, 0, 0, 0, 0, 0, 0};
      if(only < 0 || only == 0) { r[0] = run<0>(buf, nblocks, chains, steps, out, 3); }
      if(only < 0 || only == 1) { r[1] = run<1>(buf, nblocks, chains, steps, out, 3); }
      if(only < 0 || only == 2) { r[2] = run<2>(buf, nblocks, chains, steps, out, 3); }
      if(only < 0 || only == 3) { r[3] = run<3>(buf, nblocks, chains, steps, out, 3); }
      if(only < 0 || only == 4) { r[4] = run<4>(buf, nblocks, chains, steps, out, 3); }
      if(only < 0 || only == 5) { r[5] = run<5>(buf, nblocks, chains, steps, out, 3); }
      if(only < 0 || only == 6) { r[6] = run<6>(buf, nblocks, chains, steps, out, only == 6 ? 10 : 3); }
      for(int m = 0; m < 7; m++)
      {
        if(only >= 0 && m != only) { continue; }
        printf("2^%-8d %-8s %12.2f %12.1f %10llu\n", lg, names[m], r[m] / 1e9, r[m] * (m >= 5 ? 128 : 64) / 1e9, (unsigned long long)chains);
      }
    }
    CHECK(hipFree(buf));
  }
  return 0;
}
