// kernels_common.hpp -- launch geometry, LDS tables and the helpers every kernel shares.
// Part of the single translation unit gcsa2_hip.hip (device code, anonymous namespace).
#pragma once

#include "layout.hpp"
#include "../../include/gcsa2_hip.h"

using namespace g2;

namespace {


constexpr int TPB = 256;   // 4 waves per workgroup

struct Tables   // small per-workgroup lookup tables staged into LDS
{
  u64 C[MAX_SIGMA + 1];
  u8 c2c[256];
};

__device__ __forceinline__ void stage_tables(const DevImage& img, Tables& t)
{
  // DevImage lives in the kernarg segment; a lane-indexed read of it is a plain global load.
  if(threadIdx.x <= MAX_SIGMA) { t.C[threadIdx.x] = img.C[threadIdx.x]; }
  t.c2c[threadIdx.x & 255] = img.char2comp[threadIdx.x & 255];
  __syncthreads();
}

__device__ __forceinline__ u64 clampu(u64 x, u64 hi) { return x < hi ? x : hi; }

// pathNodeRange (gcsa.h:253-258)
__device__ __forceinline__ void path_node_range(const DevImage& img, u64& sp, u64& ep)
{
  u64 a, b;
  bv_rank2(img.edges, clampu(sp, img.e), clampu(ep, img.e), a, b);
  sp = a; ep = b;
}

// The per-comp descriptors sit in the kernarg segment; selecting one by a lane-varying comp is a
// global load of the descriptor.  All B_c have the same geometry, so only the base pointer varies.
__device__ __forceinline__ DevBV bwt_of(const DevImage& img, u32 comp)
{
  DevBV bv = img.bwt[0];
  bv.blocks = img.bwt[0].blocks + u64(comp) * (img.bwt[0].nblocks * BLOCK_WORDS);
  return bv;
}


}  // namespace
