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

// Block and in-block offset of a position in the FLP128 (192 positions) / FLB128 (384 positions) arrays.  192 = 3 x 64 and
// 384 = 3 x 128, and positions stay below 2^38 (gcsa2_index_create refuses larger indexes: the blocks of one would not fit
// any GPU's memory), so the division is ONE 32-bit multiply-high instead of the 64-bit magic multiplication the compiler
// emits for `pos / 192` -- four of those per LF step were a tenth of the step's instructions.
constexpr u64 MAX_PATH_NODES = (u64(1) << 38) - 2;
__device__ __forceinline__ u32 div3_u32(u32 x) { return __umulhi(x, 0xAAAAAAABu) >> 1; }
__device__ __forceinline__ void pair_block_of(u64 pos, u32& block, u32& offset)
{
  const u32 hi = u32(pos >> 6);
  block = div3_u32(hi); offset = ((hi - 3 * block) << 6) | (u32(pos) & 63);
}
__device__ __forceinline__ void flb_block_of(u64 pos, u32& block, u32& offset)
{
  const u32 hi = u32(pos >> 7);
  block = div3_u32(hi); offset = ((hi - 3 * block) << 7) | (u32(pos) & 127);
}

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


// completion ticket of a small host-pointer call: written behind the call's kernels into page-locked memory the host polls
__global__ void k_ticket(volatile unsigned long long* flag, unsigned long long ticket)
{
  __threadfence_system();
  *flag = ticket;
}

}  // namespace
