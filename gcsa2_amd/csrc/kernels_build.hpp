// kernels_build.hpp -- the device image is built ON the device: the plain bit arrays of a gcsa2_host_view (in host or device
// memory) are copied to HBM once and turned into RB64 rank blocks, select hints, fused FLB128 blocks and the charRange table by
// the kernels below.  Rounds 1-2 staged the whole image on the host (12 bytes of host memory per path node, ten seconds of a
// 16-thread host for a whole-genome index -- per rank, so eight ranks of one node stood in each other's way).
// Part of the single translation unit gcsa2_hip.hip (device code, anonymous namespace).
#pragma once

#include "kernels_common.hpp"

using namespace g2;

namespace {

// RB64 payload (layout.hpp): block b gets plain words 7 b .. 7 b + 6 (masked past `size`), counts[b] = its ones
__global__ __launch_bounds__(TPB) void k_rb64_fill(const u64* __restrict__ plain, u64 size, u64 nblocks, u64* __restrict__ blocks,
                                                   u64* __restrict__ counts)
{
  const u64 b = u64(blockIdx.x) * TPB + threadIdx.x;
  if(b >= nblocks) { return; }
  const u64 total_words = (size + 63) / 64;
  u64 ones = 0;
#pragma unroll
  for(u64 j = 0; j < PAYLOAD_WORDS; j++)
  {
    const u64 w = b * PAYLOAD_WORDS + j;
    u64 val = 0;
    if(w < total_words)
    {
      val = plain[w];
      if(w == (size >> 6) && (size & 63)) { val &= (u64(1) << (size & 63)) - 1; }
    }
    blocks[b * BLOCK_WORDS + 1 + j] = val;
    ones += u64(__popcll(val));
  }
  counts[b] = ones;
}

// word 0 of every block = ones before it (the exclusive scan of counts)
__global__ __launch_bounds__(TPB) void k_rb64_counts(const u64* __restrict__ before, u64 nblocks, u64* __restrict__ blocks)
{
  const u64 b = u64(blockIdx.x) * TPB + threadIdx.x;
  if(b < nblocks) { blocks[b * BLOCK_WORDS] = before[b]; }
}

// select hint j = the largest block whose counter is below j * SELECT_SAMPLE + 1 (layout.hpp: bv_select)
__global__ __launch_bounds__(TPB) void k_select_hints(const u64* __restrict__ blocks, u64 nblocks, u64 nhints, u32* __restrict__ hints)
{
  const u64 j = u64(blockIdx.x) * TPB + threadIdx.x;
  if(j >= nhints) { return; }
  const u64 target = j * SELECT_SAMPLE + 1;
  u64 lo = 0, hi = nblocks - 1;
  while(lo < hi)
  {
    const u64 mid = (lo + hi + 1) >> 1;
    if(blocks[mid * BLOCK_WORDS] < target) { lo = mid; } else { hi = mid - 1; }
  }
  hints[j] = u32(lo);
}

// rank(edges, min(x, e)) from the RB64 form
__device__ __forceinline__ u64 edges_rank(const DevImage& img, u64 x) { return bv_rank(img.edges, x > img.e ? img.e : x); }

// plain word w of B_c (zero past the vector) from its RB64 blocks
__device__ __forceinline__ u64 bwt_word(const DevBV& bv, u64 w)
{
  const u64 blk = w / PAYLOAD_WORDS;
  return blk < bv.nblocks ? bv.blocks[blk * BLOCK_WORDS + 1 + w % PAYLOAD_WORDS] : 0;
}

// FLB128 blocks (layout.hpp) of every comp: one thread per (comp, block); grid.y = comp
__global__ __launch_bounds__(TPB) void k_build_flb(DevImage img, u64* __restrict__ out)
{
  const u64 b = u64(blockIdx.x) * TPB + threadIdx.x;
  const u32 c = blockIdx.y;
  if(b >= img.flb_nblocks) { return; }
  const DevBV bv = bwt_of(img, c);
  const u64 pos = b * FLB_BITS;                             // <= n
  const u64 ecnt = img.C[c] + bv_rank(bv, pos);
  const u64 prev = (ecnt > 0 && ecnt - 1 < img.e && bv_get(img.edges, ecnt - 1)) ? 1 : 0;
  u64* dst = out + (u64(c) * img.flb_nblocks + b) * FLB_WORDS;
  dst[0] = ecnt;
  dst[1] = edges_rank(img, ecnt) | (prev << 63);
  u64 cum_b = 0, cum_e = 0, run_b = 0, run_e = 0;
#pragma unroll
  for(u64 j = 0; j < FLB_PAYLOAD; j++)
  {
    const u64 pw = bwt_word(bv, b * FLB_PAYLOAD + j);
    const u64 ew = (ecnt + 64 * j < img.e ? bv_bits64(img.edges, ecnt + 64 * j) : 0);
    dst[2 + j] = pw; dst[8 + j] = ew;
    run_b += u64(__popcll(pw)); run_e += u64(__popcll(ew));
    cum_b |= run_b << (10 * j); cum_e |= run_e << (10 * j);
  }
  dst[14] = cum_b; dst[15] = cum_e;
}

// charRange(c) in node space for every comp: pathNodeRange of (C[c], C[c + 1] - 1), gcsa.h:150-153
__global__ void k_crange(DevImage img, u64* __restrict__ out)
{
  const u32 c = threadIdx.x;
  if(c >= img.sigma) { return; }
  out[2 * c] = edges_rank(img, img.C[c]);
  out[2 * c + 1] = edges_rank(img, img.C[c + 1] - 1);
}

}  // namespace
