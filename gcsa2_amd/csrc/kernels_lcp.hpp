// kernels_lcp.hpp -- LCPArray::parent / depth / psv / nsv / rmq over the range-minimum tree, sample and LCP accessors.
// Part of the single translation unit gcsa2_hip.hip (device code, anonymous namespace).
#pragma once

#include "kernels_find.hpp"

using namespace g2;

namespace {

// ---- suffix-tree operations over the LCP range-minimum tree ------------------------------

struct Lcp
{
  const DevImage& img;
  __device__ __forceinline__ u64 at(u64 i) const { return img.lcp[i]; }
  __device__ __forceinline__ u64 root() const { return img.lcp_values - 1; }
  __device__ __forceinline__ u64 parent(u64 node, u64 level) const
  { return img.lcp_offsets[level + 1] + (node - img.lcp_offsets[level]) / img.lcp_branching; }
  __device__ __forceinline__ u64 first_sibling(u64 node, u64 level) const
  { return node - (node - img.lcp_offsets[level]) % img.lcp_branching; }
  __device__ __forceinline__ u64 last_sibling(u64 first_child, u64 level) const
  {
    u64 a = img.lcp_offsets[level + 1], b = first_child + img.lcp_branching;
    return (a < b ? a : b) - 1;
  }
  __device__ __forceinline__ u64 first_child(u64 node, u64 level) const
  { return img.lcp_offsets[level - 1] + (node - img.lcp_offsets[level]) * img.lcp_branching; }
  __device__ __forceinline__ u64 last_child(u64 node, u64 level) const
  { return last_sibling(first_child(node, level), level - 1); }
  __device__ __forceinline__ u64 level_of(u64 node) const
  { u64 level = 0; while(img.lcp_offsets[level + 1] <= node) { level++; } return level; }
};

// ---- byte scans of a sibling group, eight bytes per load ------------------------------------------
// The reference scans a sibling group one value at a time (lcp.cpp:352-366, 408-422); here the group's
// aligned 8-byte words are fetched with independent loads and compared eight bytes at a time: the
// position found is the same (the nearest one in scan direction), the dependent load chain is one
// round trip per level instead of one per byte.

// bit b set iff byte b of w < bound (bound <= 256): bytes in 16-bit lanes, borrow-free subtract
__device__ __forceinline__ u32 bytes_below(u64 w, u64 bound)
{
  const u64 H = 0x8000800080008000ull, M = 0x00FF00FF00FF00FFull;
  const u64 v16 = bound * 0x0001000100010001ull;
  u64 even = ~(((w & M) | H) - v16) & H;             // lane bit 15 set iff the even byte < bound
  u64 odd = ~((((w >> 8) & M) | H) - v16) & H;
  u64 m = (even >> 15) | (odd >> 14);                // bits 16j (byte 2j) and 16j + 1 (byte 2j + 1)
  return u32((m & 3) | ((m >> 14) & 0xC) | ((m >> 28) & 0x30) | ((m >> 42) & 0xC0));
}

constexpr u32 SCAN_WORDS = 8;      // one batch covers a 64-value sibling group (the reference's default branching)

// nearest i in [from, to) below `to` with LCP[i] < bound
__device__ __forceinline__ bool scan_left(const DevImage& img, u64 from, u64 to, u64 bound, u64& rpos, u64& rval)
{
  const u64* words = reinterpret_cast<const u64*>(img.lcp);
  while(to > from)
  {
    const u64 top = (to - 1) >> 3, bottom = from >> 3;
    u64 w[SCAN_WORDS];
#pragma unroll
    for(u32 k = 0; k < SCAN_WORDS; k++) { w[k] = (top >= bottom + k ? words[top - k] : 0); }
#pragma unroll
    for(u32 k = 0; k < SCAN_WORDS; k++)
    {
      if(top < bottom + k) { return false; }
      const u64 base = (top - k) << 3;
      u32 mask = bytes_below(w[k], bound);
      if(to < base + 8) { mask &= (1u << (to - base)) - 1; }
      if(from > base) { mask &= ~((1u << (from - base)) - 1); }
      if(mask != 0)
      {
        u32 byte = 31 - __clz(int(mask));
        rpos = base + byte; rval = (w[k] >> (8 * byte)) & 0xFF;
        return true;
      }
    }
    if(top < bottom + SCAN_WORDS) { return false; }
    to = (top - (SCAN_WORDS - 1)) << 3;
  }
  return false;
}

// first i in [from, last] with LCP[i] < bound
__device__ __forceinline__ bool scan_right(const DevImage& img, u64 from, u64 last, u64 bound, u64& rpos, u64& rval)
{
  const u64* words = reinterpret_cast<const u64*>(img.lcp);
  while(from <= last)
  {
    const u64 bottom = from >> 3, top = last >> 3;
    u64 w[SCAN_WORDS];
#pragma unroll
    for(u32 k = 0; k < SCAN_WORDS; k++) { w[k] = (bottom + k <= top ? words[bottom + k] : 0); }
#pragma unroll
    for(u32 k = 0; k < SCAN_WORDS; k++)
    {
      if(bottom + k > top) { return false; }
      const u64 base = (bottom + k) << 3;
      u32 mask = bytes_below(w[k], bound);
      if(from > base) { mask &= ~((1u << (from - base)) - 1); }
      if(last < base + 7) { mask &= (2u << (last - base)) - 1; }
      if(mask != 0)
      {
        u32 byte = u32(__ffs(int(mask))) - 1;
        rpos = base + byte; rval = (w[k] >> (8 * byte)) & 0xFF;
        return true;
      }
    }
    from = (bottom + SCAN_WORDS) << 3;
  }
  return false;
}

// psv / psev (src/lcp.cpp:345-382)
__device__ void lcp_psv(const DevImage& img, u64 to, bool equal, u64& rpos, u64& rval)
{
  Lcp L{img};
  rpos = rval = img.lcp_values;                     // notFound()
  if(to == 0 || to >= img.lcp_size) { return; }
  u64 level = 0;
  const u64 bound = L.at(to) + (equal ? 1 : 0);     // v < val, or v <= val
  bool found = false;
  while(to != L.root())
  {
    if(scan_left(img, L.first_sibling(to, level), to, bound, rpos, rval)) { found = true; break; }
    to = L.parent(to, level); level++;
  }
  if(!found) { rpos = rval = img.lcp_values; return; }
  while(level > 0)
  {
    u64 from = L.first_child(rpos, level); level--;
    scan_left(img, from, L.last_sibling(from, level) + 1, bound, rpos, rval);
  }
}

// nsv / nsev (src/lcp.cpp:401-438)
__device__ void lcp_nsv(const DevImage& img, u64 from, bool equal, u64& rpos, u64& rval)
{
  Lcp L{img};
  rpos = rval = img.lcp_values;
  if(from + 1 >= img.lcp_size) { return; }
  u64 level = 0;
  const u64 bound = L.at(from) + (equal ? 1 : 0);
  bool found = false;
  while(from != L.root())
  {
    u64 last = L.last_sibling(L.first_sibling(from, level), level);
    if(from + 1 <= last && scan_right(img, from + 1, last, bound, rpos, rval)) { found = true; break; }
    from = L.parent(from, level); level++;
  }
  if(!found) { rpos = rval = img.lcp_values; return; }
  while(level > 0)
  {
    from = L.first_child(rpos, level); level--;
    scan_right(img, from, L.last_sibling(from, level), bound, rpos, rval);
  }
}

// rmq (src/lcp.cpp:448-513): leftmost minimum of LCP[sp..ep].  Same tree walk as the reference;
// its explicit stack of right-hand tails is replaced by one accumulator that prefers the later
// (= more leftward) candidate on ties, which yields the same leftmost minimum.
__device__ void lcp_rmq(const DevImage& img, u64 sp, u64 ep, u64& rpos, u64& rval)
{
  Lcp L{img};
  if(sp > ep || ep >= img.lcp_size) { rpos = rval = img.lcp_values; return; }
  if(sp == ep) { rpos = sp; rval = L.at(sp); return; }
  const u64 INF = ~u64(0);
  u64 lpos = img.lcp_values, lval = INF, tpos = img.lcp_values, tval = INF;
  u64 level = 0, left = sp, right = ep;
  while(true)
  {
    u64 left_par = L.parent(left, level), right_par = L.parent(right, level);
    if(left_par == right_par)
    {
      for(u64 i = left; i <= right; i++) { u64 v = L.at(i); if(v < lval) { lpos = i; lval = v; } }
      break;
    }
    u64 left_child = L.first_child(left_par, level + 1);
    if(left != left_child)
    {
      u64 last = L.last_sibling(left_child, level);
      for(u64 i = left; i <= last; i++) { u64 v = L.at(i); if(v < lval) { lpos = i; lval = v; } }
      left_par++;
    }
    u64 right_child = L.last_child(right_par, level + 1);
    if(right != right_child)
    {
      u64 first = L.first_sibling(right_child, level);
      u64 gpos = img.lcp_values, gval = INF;
      for(u64 i = first; i <= right; i++) { u64 v = L.at(i); if(v < gval) { gpos = i; gval = v; } }
      if(gval <= tval) { tpos = gpos; tval = gval; }      // this group lies left of earlier tails
      right_par--;
    }
    if(left_par >= right_par)
    {
      if(left_par == right_par) { u64 v = L.at(left_par); if(v < lval) { lpos = left_par; lval = v; } }
      break;
    }
    left = left_par; right = right_par; level++;
  }
  if(lval <= tval) { rpos = lpos; rval = lval; } else { rpos = tpos; rval = tval; }
  level = L.level_of(rpos);
  while(level > 0)
  {
    rpos = L.first_child(rpos, level); level--;
    while(L.at(rpos) != rval) { rpos++; }
  }
}

// nodeFor (lcp.h:163-175) + parent (src/lcp.cpp:276-301)
__device__ void lcp_parent(const DevImage& img, u64 sp, u64 ep, gcsa2_stnode& out)
{
  if(sp == 0 && ep == img.lcp_size - 1) { out = gcsa2_stnode{0, img.lcp_size - 1, 0, 0, 0}; return; }
  u64 sp_safe = clampu(sp, img.lcp_size - 1);
  u64 left_lcp = img.lcp[sp_safe];
  u64 right_lcp = (ep + 1 < img.lcp_size ? img.lcp[ep + 1] : 0);
  u64 node_lcp = (left_lcp > right_lcp ? left_lcp : right_lcp);
  u64 lpos = sp, lval = left_lcp, rpos = ep + 1, rval = right_lcp;
  if(left_lcp == node_lcp)
  {
    lcp_psv(img, sp, false, lpos, lval);
    if(lpos == img.lcp_values && lval == img.lcp_values) { lpos = 0; lval = 0; }
  }
  if(right_lcp == node_lcp)
  {
    lcp_nsv(img, ep + 1, false, rpos, rval);
    if(rpos == img.lcp_values && rval == img.lcp_values) { rpos = img.lcp_size; rval = 0; }
  }
  out = gcsa2_stnode{lpos, rpos - 1, lval, rval, node_lcp};
}

__global__ __launch_bounds__(TPB) void k_parent(DevImage img, const u64* __restrict__ ranges, u64 nq,
                                                gcsa2_stnode* __restrict__ out)
{
  u64 q = u64(blockIdx.x) * TPB + threadIdx.x;
  if(q >= nq) { return; }
  ulonglong2 r = reinterpret_cast<const ulonglong2*>(ranges)[q];
  gcsa2_stnode node;
  lcp_parent(img, r.x, r.y, node);
  out[q] = node;
}

// depth(range) (src/lcp.cpp:319-325)
__global__ __launch_bounds__(TPB) void k_depth(DevImage img, const u64* __restrict__ ranges, u64 nq,
                                               u64* __restrict__ out)
{
  u64 q = u64(blockIdx.x) * TPB + threadIdx.x;
  if(q >= nq) { return; }
  ulonglong2 r = reinterpret_cast<const ulonglong2*>(ranges)[q];
  u64 res = GCSA2_UNKNOWN;
  if(r.y + 1 - r.x > 1)
  {
    u64 pos, val;
    lcp_rmq(img, r.x + 1, r.y, pos, val);
    if(!(pos == img.lcp_values && val == img.lcp_values)) { res = val; }
  }
  out[q] = res;
}

__global__ __launch_bounds__(TPB) void k_sv(DevImage img, int op, const u64* __restrict__ positions, u64 nq,
                                            u64* __restrict__ out)
{
  u64 q = u64(blockIdx.x) * TPB + threadIdx.x;
  if(q >= nq) { return; }
  u64 pos, val;
  if(op < 2) { lcp_psv(img, positions[q], op & 1, pos, val); }
  else { lcp_nsv(img, positions[q], op & 1, pos, val); }
  reinterpret_cast<ulonglong2*>(out)[q] = make_ulonglong2(pos, val);
}

__global__ __launch_bounds__(TPB) void k_rmq(DevImage img, const u64* __restrict__ ranges, u64 nq,
                                             u64* __restrict__ out)
{
  u64 q = u64(blockIdx.x) * TPB + threadIdx.x;
  if(q >= nq) { return; }
  ulonglong2 r = reinterpret_cast<const ulonglong2*>(ranges)[q];
  u64 pos, val;
  lcp_rmq(img, r.x, r.y, pos, val);
  reinterpret_cast<ulonglong2*>(out)[q] = make_ulonglong2(pos, val);
}

// sampled / sampleRange / firstSample (gcsa.h:191-206): out[3q] = sampled(node),
// out[3q+1] = sampleRange(node).first = firstSample(node), out[3q+2] = sampleRange(node).second
__global__ __launch_bounds__(TPB) void k_sample_range(DevImage img, const u64* __restrict__ nodes, u64 nq,
                                                      u64* __restrict__ out)
{
  u64 q = u64(blockIdx.x) * TPB + threadIdx.x;
  if(q >= nq) { return; }
  u64 node = clampu(nodes[q], img.n), r;
  bool s = (node < img.n ? bv_get_rank(img.sampled, node, r) : (r = bv_rank(img.sampled, node), false));
  u64 first = (r > 0 ? bv_select(img.samples, r) + 1 : 0);
  u64 second = (r + 1 <= img.samples.ones ? bv_select(img.samples, r + 1) : img.sample_count);
  out[3 * q] = s ? 1 : 0; out[3 * q + 1] = first; out[3 * q + 2] = second;
}

// sample(i), lastSample(i) (gcsa.h:208-210)
__global__ __launch_bounds__(TPB) void k_sample(DevImage img, const u64* __restrict__ idx, u64 nq,
                                                u64* __restrict__ values, u8* __restrict__ last)
{
  u64 q = u64(blockIdx.x) * TPB + threadIdx.x;
  if(q >= nq) { return; }
  u64 i = idx[q];
  bool ok = (i < img.sample_count);
  values[q] = ok ? packed_get(img.stored, img.sample_width, i) : 0;
  last[q] = (ok && bv_get(img.samples, i)) ? 1 : 0;
}

// LCPArray::operator[] (lcp.h:129)
__global__ __launch_bounds__(TPB) void k_lcp_access(DevImage img, const u64* __restrict__ pos, u64 nq, u64* __restrict__ out)
{
  u64 q = u64(blockIdx.x) * TPB + threadIdx.x;
  if(q >= nq) { return; }
  out[q] = (pos[q] < img.lcp_values ? img.lcp[pos[q]] : 0);
}


// ---- matching statistics: LF + parent fused (the interplay vg's MEM finder drives) -------------
// One lane per pattern, right to left: try LF(range, comp) (gcsa.h:155-162); while the result is
// empty, replace the range by its parent (lcp.cpp:276-301) and retry; at the root the character is
// skipped.  ms[offset + i] = length of the longest match starting at i (capped at 65535), the final
// range is the one of position 0.  Paper: paper.tex:344 (after Ohlebusch et al. 2010).
__global__ __launch_bounds__(TPB) void k_match_stats(DevImage img, const u8* __restrict__ patterns,
                                                     const u64* __restrict__ offsets, u64 nq,
                                                     unsigned short* __restrict__ ms, u64* __restrict__ ranges,
                                                     u64* __restrict__ fallbacks)
{
  __shared__ Tables t;
  stage_tables(img, t);
  u64 q = u64(blockIdx.x) * TPB + threadIdx.x;
  if(q >= nq) { return; }
  u64 begin = offsets[q], len = offsets[q + 1] - begin;
  const u8* p = patterns + begin;
  u64 sp = 0, ep = img.n - 1, depth = 0, calls = 0;
  u64 word = 0, word_addr = ~u64(0);        // pattern bytes: back to front from aligned 8-byte words (one load per 8 steps)
  u64 packed = 0; u32 have = 0;             // results: four u16 per aligned 8-byte store
  for(u64 i = len; i-- > 0; )
  {
    u64 addr = reinterpret_cast<u64>(p) + i, aligned = addr & ~u64(7);
    if(aligned != word_addr) { word = *reinterpret_cast<const u64*>(aligned); word_addr = aligned; }
    u32 comp = t.c2c[u32(word >> ((addr & 7) * 8)) & 0xFF];
    while(true)
    {
      u64 a, b, nsp, nep;
      lf_fused_lane(img, comp, sp, ep, a, b, nsp, nep);          // gcsa.h:155-162
      if(!range_empty(a, b))
      {
        sp = nsp; ep = nep; depth++;
        break;
      }
      if(sp == 0 && ep == img.n - 1) { depth = 0; break; }     // at the root: no such character
      gcsa2_stnode node;
      lcp_parent(img, sp, ep, node); calls++;
      sp = node.sp; ep = node.ep; depth = node.node_lcp;
    }
    const u64 idx = begin + i;
    const u32 slot = u32(idx & 3);
    packed |= u64(depth > 65535 ? 65535 : depth) << (16 * slot); have |= 1u << slot;
    if(slot == 0 || i == 0)                   // the group of four is complete, or the pattern ends inside it
    {
      unsigned short* group = ms + (idx & ~u64(3));
      if(have == 15u) { *reinterpret_cast<u64*>(group) = packed; }
      else { for(u32 s = 0; s < 4; s++) { if((have >> s) & 1) { group[s] = (unsigned short)(packed >> (16 * s)); } } }
      packed = 0; have = 0;
    }
  }
  reinterpret_cast<ulonglong2*>(ranges)[q] = make_ulonglong2(sp, ep);
  if(fallbacks != nullptr) { fallbacks[q] = calls; }
}

}  // namespace
