// layout.hpp -- HBM image of a GCSA2 index and the device-side succinct primitives.
//
// Every bitvector of the reference (fast_bwt / sparse_bwt / edges / sampled_paths / samples /
// the Sadakane vectors; include/gcsa/gcsa.h:214-240) is stored in ONE format, "RB64":
//
//   block b = 8 x u64 = 64 bytes, 64-byte aligned
//     word 0      number of 1-bits in [0, 448 b)
//     words 1..7  payload bits [448 b, 448 (b + 1)), LSB first
//
// so rank(i) is exactly one aligned 64-byte fetch (the "rank probe" of SURVEY.md 8(d)), and
// because 448 = 7 x 64 payload word j of block b is plain word 7 b + j (no bit shifting on
// upload).  A vector of `size` bits gets size / 448 + 1 blocks so that rank(size) is legal
// (it occurs for ep = n - 1, include/gcsa/gcsa.h:272).  The sparse vectors of the reference
// (sd_vector for $, N, # and for the Sadakane counters) are stored densely in the same format:
// rank / select / access are integer functions of the bit sequence, so results are identical
// and the fast/sparse branch of gcsa.h:157-158 disappears.
//
// For find() / LF(range) every B_c is ALSO stored in a fused 128-byte form, "FLB128":
//
//   block b of comp c (16 x u64 = 128 bytes, 128-byte aligned)
//     word 0       ecnt = C[c] + rank(B_c, 384 b)                  edge-space position of the block
//     word 1       ncnt = rank(edges, ecnt), bit 63 = edges[ecnt - 1]
//     words 2..7   B_c payload bits [384 b, 384 (b + 1))
//     words 8..13  edges bits [ecnt, ecnt + 384)                   the slice those ones map into
//     word 14      running popcounts of words 2..7, 10 bits each: w2, w2 + w3, ... (six fields, low to high)
//     word 15      the same for words 8..13
//   (the running popcounts make a rank inside the block ONE masked popcount, of the word that holds the position,
//   instead of one per payload word: round 2 found the matching-statistics kernel bound by instruction issue)
//
// so one LF endpoint, C[c] + rank(B_c, i) followed by rank(edges, .) (gcsa.h:262-274, 253-258), is
// ONE 128-byte fetch instead of two dependent 64-byte fetches: beyond L2 the memory system is
// bound by requests per second, not bytes (profiles/r01_gather_bench.md), and a 128-byte request
// costs the same as a 64-byte one.  Blocks are fetched cooperatively: 8 adjacent lanes load the
// 8 x 16 bytes of one block in a single line-coalesced instruction and stage it in LDS.
//
// Two characters per step, "FLP128": for every ordered pair (c1, c2) of fast characters (c2 is consumed
// first by the backward search, then c1) the composition of two LF steps is itself a rank structure.
// With E_c(i) = C[c] + rank(B_c, i) and N(x) = rank(edges, x):
//
//   H(i)  = E_c1(N(E_c2(i)))                 is monotone with increments of 0 or 1, so H(i) = H(0) + rank(P, i),
//           P[i] = B_c2[i] & edges[x] & B_c1[N(x)],  x = E_c2(i)
//   sp''  = N(H(sp))                                                  (gcsa.h:271, 161 applied twice)
//   ep''  = N(H(ep + 1) - 1 + D[ep + 1]),  D[j] = !edges[x - 1] & B_c1[N(x)],  x = E_c2(j)
//           (D = 1: position j splits the out-edges of one node, which both sides then lead back to)
//
//   block b of pair (c1, c2) (16 x u64 = 128 bytes; 192 positions)
//     word 0        ecnt = H(192 b)
//     word 1        ncnt = rank(edges, ecnt), bit 63 = edges[ecnt - 1]
//     words 2..4    P bits [192 b, 192 (b + 1))
//     words 5..7    D bits of the same positions
//     words 8..10   Q bits of the same positions: Q = B_c2 (is the FIRST of the two steps non-empty?)
//     words 11..14  edges bits [ecnt, ecnt + 256)
//     word 15       running popcounts, one byte each: P0, P0+P1 | Q0, Q0+Q1, Q0+Q1+Q2 | E0, E0+E1, E0+E1+E2 (low to high)
//
// so TWO pattern characters cost ONE 128-byte request per endpoint.  What the block decides, exactly:
//   H(ep + 1) > H(sp)                      both steps are non-empty: the range is (N(H(sp)), N(H(ep + 1) - 1 + D))
//   H(ep + 1) = H(sp), a Q bit in [sp, ep] the first step is non-empty; with D[ep + 1] = 1 the second one is the single edge
//                                          H(sp) (the walk passes a non-last out-edge of a branching node), with D = 0 it is
//                                          empty and (H(sp), H(sp) - 1) are the edge-space integers the reference returns
//                                          (gcsa.h:160)
//   no Q bit in [sp, ep]                   the first step is empty: its edge-space integers are not in this block, the
//                                          step is replayed from the FLB128 blocks (so is a range whose endpoints lie in
//                                          different blocks with no Q bit in either part: undecided)
// Results are therefore unchanged.
//
// select_1 uses one u32 hint per 448 ones (block holding the (448 j + 1)-th one) followed by a
// short binary search over block counters and an in-block scan.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace g2 {

typedef uint64_t u64;
typedef uint32_t u32;
typedef uint8_t  u8;

constexpr u64 BLOCK_WORDS   = 8;
constexpr u64 PAYLOAD_WORDS = 7;
constexpr u64 BLOCK_BITS    = 448;
constexpr u64 BLOCK_BYTES   = 64;
constexpr u64 SELECT_SAMPLE = 448;
constexpr u64 FLB_WORDS     = 16;
constexpr u64 FLB_BITS      = 384;           // positions per FLB128 block (six payload words)
constexpr u64 FLB_PAYLOAD   = 6;
constexpr u64 FLB_BYTES     = 128;
constexpr u64 PREV_BIT      = u64(1) << 63;
constexpr u64 PAIR_BITS     = 192;           // positions per FLP128 block (three payload rows of three words)
constexpr u32 PAIR_WORDS    = 3;
constexpr u32 PAIR_FLAG     = u32(1) << 30;  // block index refers to the FLP128 array
constexpr u32 LCP_FLAG      = u32(1) << 31;  // "block" index refers to the LCP array, in 16-byte units (a 128-byte window of it)
constexpr int MAX_SIGMA     = 16;
constexpr int MAX_LCP_LEVELS = 64;      // a binary tree (branching 2, the smallest the reference allows) over 2^63 values

struct DevBV
{
  const u64* blocks;   // nblocks * 8 words
  const u32* hints;    // select hints or nullptr
  u64 size, nblocks, ones;
};

// Passed to kernels by value (kernarg segment, scalar loads).
struct DevImage
{
  u64 n, e, sigma, fast_chars;
  u64 C[MAX_SIGMA + 1];
  DevBV bwt[MAX_SIGMA];
  DevBV edges;
  const u64* flb;       // fused LF blocks: comp c, block b at flb + (c * flb_nblocks + b) * 16
  u64 flb_nblocks;      // per comp = n / 448 + 1
  const u64* flp;       // fused pair blocks or nullptr: pair (c1, c2), block b at flp + (((c1 - 1) * 4 + c2 - 1) * flp_nblocks + b) * 16
  u64 flp_nblocks;      // per pair = n / 192 + 1
  u64 crange[2 * MAX_SIGMA];   // charRange(c) in node space, precomputed (gcsa.h:150-153)
  const u64* pred4;            // 4 bits per path node: bits 0-2 = comp of the first incoming edge
                               // (gcsa.h:165-183 probe order), bit 3 = sampled(node); nullptr if sigma > 8
  const u64* kmer_table;       // find() of every k-mer over comps 1..4: 4^kmer_k packed entries (kernels_find.hpp, seed_pack)
  u32 kmer_k;                  // 0 = no table
  u32 seed_wide;               // ranges of this many path nodes or more are marked instead of stored (2^24 - 1; lower in tests)
  u32 lcp_shift;               // log2(lcp_branching) when that is a power of two (the reference's default 64), else 0
  const ulonglong2* jump_tab;  // memoised unary LF chains, one entry per path node, or nullptr: x = node reached | steps << 56
                               // (0..8 steps), y = the comps (minus 1) of those steps, 2 bits each, first step lowest
  const u64* locate_tab;       // memoised locateInternal walk (gcsa.cpp:880-896), one u64 per path node, or nullptr:
                               //   bit 63 set: the walk ends in a node with ONE sample; bits 0-62 = sample + steps
                               //   bit 63 clear: bits 0-39 = firstSample(end node), bits 40-62 = steps
  DevBV sampled;        // sampled_paths
  DevBV samples;        // + select
  const u64* stored;    // packed stored_samples
  u64 sample_count, sample_width;
  DevBV xfilter;        // extra_pointers.filter
  DevBV xvalues;        // extra_pointers.values, + select
  DevBV redundant;      // redundant_pointers.data, + select
  const u8* lcp;        // LCP bytes + range-minimum tree levels
  u64 lcp_size, lcp_branching, lcp_levels, lcp_values;
  u64 lcp_offsets[MAX_LCP_LEVELS + 1];
  int has_samples, has_counters, has_lcp;
  u8 char2comp[256];
};

// ------------------------------------------------------------------------------------------
// device primitives

struct Block { u64 w[8]; };

__device__ __forceinline__ Block load_block(const u64* p)
{
  // 64-byte aligned: four 16-byte vector loads
  const ulonglong2* q = reinterpret_cast<const ulonglong2*>(p);
  ulonglong2 a = q[0], b = q[1], c = q[2], d = q[3];
  Block r;
  r.w[0] = a.x; r.w[1] = a.y; r.w[2] = b.x; r.w[3] = b.y;
  r.w[4] = c.x; r.w[5] = c.y; r.w[6] = d.x; r.w[7] = d.y;
  return r;
}

// rank inside a loaded block; r = bit offset in the block, 0 <= r < 448
__device__ __forceinline__ u64 rank_in_block(const Block& b, u32 r)
{
  u64 res = b.w[0];
  u32 wq = r >> 6, rb = r & 63;
  u64 part = (u64(1) << rb) - 1;
#pragma unroll
  for(u32 j = 0; j < 7; j++)
  {
    u64 m = (j < wq ? ~u64(0) : (j == wq ? part : u64(0)));
    res += __popcll(b.w[j + 1] & m);
  }
  return res;
}

__device__ __forceinline__ u64 block_of(u64 i) { return i / BLOCK_BITS; }

// number of 1-bits in [0, i), 0 <= i <= size
__device__ __forceinline__ u64 bv_rank(const DevBV& bv, u64 i)
{
  u64 blk = block_of(i);
  Block b = load_block(bv.blocks + blk * BLOCK_WORDS);
  return rank_in_block(b, u32(i - blk * BLOCK_BITS));
}

// rank at two positions; one fetch when both fall into the same block
__device__ __forceinline__ void bv_rank2(const DevBV& bv, u64 i, u64 j, u64& ri, u64& rj)
{
  u64 bi = block_of(i), bj = block_of(j);
  Block a = load_block(bv.blocks + bi * BLOCK_WORDS);
  ri = rank_in_block(a, u32(i - bi * BLOCK_BITS));
  if(bj == bi) { rj = rank_in_block(a, u32(j - bj * BLOCK_BITS)); }
  else
  {
    Block b = load_block(bv.blocks + bj * BLOCK_WORDS);
    rj = rank_in_block(b, u32(j - bj * BLOCK_BITS));
  }
}

__device__ __forceinline__ bool bv_get(const DevBV& bv, u64 i)
{
  u64 blk = block_of(i);
  u32 r = u32(i - blk * BLOCK_BITS);
  return (bv.blocks[blk * BLOCK_WORDS + 1 + (r >> 6)] >> (r & 63)) & 1;
}

// access + rank with one block fetch
__device__ __forceinline__ bool bv_get_rank(const DevBV& bv, u64 i, u64& rank)
{
  u64 blk = block_of(i);
  u32 r = u32(i - blk * BLOCK_BITS);
  Block b = load_block(bv.blocks + blk * BLOCK_WORDS);
  rank = rank_in_block(b, r);
  return (b.w[1 + (r >> 6)] >> (r & 63)) & 1;
}

// position (0..63) of the k-th (k >= 1) set bit of w
__device__ __forceinline__ u32 select_in_word(u64 w, u32 k)
{
  u32 pos = 0;
#pragma unroll
  for(u32 s = 32; s >= 1; s >>= 1)
  {
    u32 c = __popcll((w >> pos) & ((u64(1) << s) - 1));
    if(c < k) { k -= c; pos += s; }
  }
  return pos;
}

// position of the r-th 1-bit, 1 <= r <= ones
__device__ __forceinline__ u64 bv_select(const DevBV& bv, u64 r)
{
  u64 k = (r - 1) / SELECT_SAMPLE;
  u64 lo = bv.hints[k], hi = bv.hints[k + 1];
  while(lo < hi)   // largest block whose counter is < r
  {
    u64 mid = (lo + hi + 1) >> 1;
    if(bv.blocks[mid * BLOCK_WORDS] < r) { lo = mid; } else { hi = mid - 1; }
  }
  Block b = load_block(bv.blocks + lo * BLOCK_WORDS);
  u32 rem = u32(r - b.w[0]);
  u64 pos = lo * BLOCK_BITS;
  u32 word = 0, found = 0;
#pragma unroll
  for(u32 j = 0; j < 7; j++)
  {
    u32 c = __popcll(b.w[j + 1]);
    if(!found)
    {
      if(c >= rem) { word = j; found = 1; } else { rem -= c; }
    }
  }
  u64 w = b.w[1];
#pragma unroll
  for(u32 j = 1; j < 7; j++) { if(word == j) { w = b.w[j + 1]; } }
  return pos + (u64(word) << 6) + select_in_word(w, rem);
}

// 64 bits of the vector starting at bit `pos` (zero past the last block)
__device__ __forceinline__ u64 bv_bits64(const DevBV& bv, u64 pos)
{
  const u64 w = pos >> 6, s = pos & 63, total = bv.nblocks * PAYLOAD_WORDS;
  const u64 lo = (w < total ? bv.blocks[(w / PAYLOAD_WORDS) * BLOCK_WORDS + 1 + w % PAYLOAD_WORDS] : 0);
  if(s == 0) { return lo; }
  const u64 hi = (w + 1 < total ? bv.blocks[((w + 1) / PAYLOAD_WORDS) * BLOCK_WORDS + 1 + (w + 1) % PAYLOAD_WORDS] : 0);
  return (lo >> s) | (hi << (64 - s));
}

__device__ __forceinline__ u64 packed_get(const u64* words, u64 width, u64 i)
{
  u64 pos = i * width, word = pos >> 6, shift = pos & 63;
  u64 val = words[word] >> shift;
  if(shift + width > 64) { val |= words[word + 1] << (64 - shift); }
  return (width >= 64 ? val : val & ((u64(1) << width) - 1));
}

__device__ __forceinline__ bool range_empty(u64 sp, u64 ep) { return sp + 1 > ep + 1; }

}  // namespace g2
