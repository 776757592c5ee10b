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
  // (a 64-bit division by a run-time value is ~150 instructions; the reference's default branching factor is a power of two)
  __device__ __forceinline__ u64 parent(u64 node, u64 level) const
  {
    const u64 d = node - img.lcp_offsets[level];
    return img.lcp_offsets[level + 1] + (img.lcp_shift != 0 ? d >> img.lcp_shift : d / img.lcp_branching);
  }
  __device__ __forceinline__ u64 first_sibling(u64 node, u64 level) const
  {
    const u64 d = node - img.lcp_offsets[level];
    return node - (img.lcp_shift != 0 ? d & (img.lcp_branching - 1) : d % img.lcp_branching);
  }
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
    // the nearest word alone first: the previous smaller value is usually within a few positions, and one
    // load per lane is one memory request instead of eight
    w[0] = words[top];
    {
      u32 mask = bytes_below(w[0], bound);
      if(to < (top << 3) + 8) { mask &= (1u << (to - (top << 3))) - 1; }
      if(from > (top << 3)) { mask &= ~((1u << (from - (top << 3))) - 1); }
      if(mask != 0)
      {
        u32 byte = 31 - __clz(int(mask));
        rpos = (top << 3) + byte; rval = (w[0] >> (8 * byte)) & 0xFF;
        return true;
      }
      if(top == bottom) { return false; }
    }
#pragma unroll
    for(u32 k = 1; k < SCAN_WORDS; k++) { w[k] = (top >= bottom + k ? words[top - k] : 0); }
#pragma unroll
    for(u32 k = 1; k < SCAN_WORDS; k++)
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
    w[0] = words[bottom];                    // the nearest word alone first (see scan_left)
    {
      u32 mask = bytes_below(w[0], bound);
      if(from > (bottom << 3)) { mask &= ~((1u << (from - (bottom << 3))) - 1); }
      if(last < (bottom << 3) + 7) { mask &= (2u << (last - (bottom << 3))) - 1; }
      if(mask != 0)
      {
        u32 byte = u32(__ffs(int(mask))) - 1;
        rpos = (bottom << 3) + byte; rval = (w[0] >> (8 * byte)) & 0xFF;
        return true;
      }
      if(top == bottom) { return false; }
    }
#pragma unroll
    for(u32 k = 1; k < SCAN_WORDS; k++) { w[k] = (bottom + k <= top ? words[bottom + k] : 0); }
#pragma unroll
    for(u32 k = 1; k < SCAN_WORDS; k++)
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

// (round 4's k_pack_patterns wrote the codes and the flags as two arrays; round 5's k_match_stats3 -- 32 LCP-window slots per
// wavefront, parent() and the retry in the round of the failure: half the rounds, twice the instructions per round, 8-17 %
// slower -- was retired in round 6 with variants 6 / 7: profiles/r05_match_stats.md has its A/B series, profiles/r06_match_stats.md
// why the kernel's next step is another one)
// Pre-pass: every pattern as 16-byte records, LAST character first.  Record j of pattern q lives at index
// (offsets[q] >> 5) + q + j (consecutive patterns never overlap, see k_pack_patterns) and holds the characters at distance
// t = 32 j .. 32 j + 31 from the pattern's end: x = comp - 1 of a fast character in bits [2 (t & 31), 2 (t & 31) + 2),
// y = bit (t & 31) set for any other character and for the positions past the pattern's first character.
__global__ __launch_bounds__(TPB) void k_pack_records(DevImage img, const u8* __restrict__ patterns, const u64* __restrict__ offsets,
                                                     u64 nq, ulonglong2* __restrict__ recs)
{
  __shared__ u8 c2c[256];
  c2c[threadIdx.x] = img.char2comp[threadIdx.x];
  __syncthreads();
  const u64 q = u64(blockIdx.x) * TPB + threadIdx.x;
  if(q >= nq) { return; }
  const u64 begin = offsets[q], len = offsets[q + 1] - begin;
  const u64 first_word = (begin >> 5) + q, words = (len + 31) >> 5;
  for(u64 j = 0; j < words; j++)
  {
    const u64 high = len - 32 * j;                              // one past the pattern position of t = 32 j
    const u64 count = (high < 32 ? high : 32), low = reinterpret_cast<u64>(patterns) + begin + high - count;
    const u64 base = low & ~u64(7), last = (low + count - 1) & ~u64(7);
    u64 w[5];
#pragma unroll
    for(u32 k = 0; k < 5; k++) { const u64 a = base + 8 * k; w[k] = *reinterpret_cast<const u64*>(a < last ? a : last); }
    u64 code = 0; u32 flags = (count < 32 ? ~u32(0) << count : 0u);
    for(u32 r = 0; r < count; r++)
    {
      const u64 at = (low - base) + (count - 1 - r);           // byte offset of the character at distance 32 j + r from the end
      u64 word = w[0];
#pragma unroll
      for(u32 k = 1; k < 5; k++) { if((at >> 3) == k) { word = w[k]; } }
      const u32 c = u32(c2c[u32(word >> ((at & 7) * 8)) & 0xFF]) - 1;
      code |= u64(c & 3) << (2 * r);
      flags |= u32(c < 4 ? 0 : 1) << r;
    }
    recs[first_word + j] = make_ulonglong2(code, u64(flags));
  }
  // (the two records behind a pattern's last one are read ahead by the kernel: flags only)
  if(q + 1 == nq) { for(u64 j = words; j < words + 3; j++) { recs[first_word + j] = make_ulonglong2(0, ~u64(0)); } }
}


// parent() of (sp, ep) from a 128-byte window of the LCP array staged in the lane's LDS slot (fetch_blocks with LCP_FLAG): the
// window starts at byte `wstart` (a multiple of 16), 48..63 positions before sp.  node_lcp = max(LCP[sp], LCP[ep + 1]); the
// previous / next smaller value on the side(s) that attain it is the nearest smaller byte of the flat array, which is what
// the tree walks of lcp_psv / lcp_nsv return, so whenever both lie inside the window the answer is complete.  false: not
// decidable from the window (the interval reaches beyond it, or touches the array's ends) -- the caller runs lcp_parent.
// After a failed LF step the climb rarely passes intervals of a few dozen path nodes (a range of w nodes extends with
// probability 1 - 0.73^w on a whole-genome index), so the window decides nearly every call; round 2's two 16-byte chunks left
// 39 % of the calls to the tree walk, which then was 44 % of the kernel's time (profiles/r03_match_stats.md).
// bit 8 k + 7 set iff byte k of w < bound (bound <= 255): the bytes of each 32-bit half in 16-bit lanes, borrow-free subtract;
// the flags stay where the bytes are, so that the nearest one is a count of leading / trailing zeros away (bytes_below above
// compacts them to eight bits: half of its instructions)
__device__ __forceinline__ u64 below_flags(u64 w, u32 bound)
{
  const u32 lo = u32(w), hi = u32(w >> 32), v = bound * 0x00010001u, H = 0x80008000u, M = 0x00FF00FFu;
  const u32 e_lo = ((lo & M) | H) - v, o_lo = (((lo >> 8) & M) | H) - v;          // lane bit 15 CLEAR iff the byte < bound
  const u32 e_hi = ((hi & M) | H) - v, o_hi = (((hi >> 8) & M) | H) - v;
  const u32 f_lo = ((~e_lo & H) >> 8) | (~o_lo & H);                               // even bytes: bit 16 j + 15 -> 16 j + 7
  const u32 f_hi = ((~e_hi & H) >> 8) | (~o_hi & H);
  return u64(f_lo) | (u64(f_hi) << 32);
}

// Where the 128-byte window of the LCP array around (sp, ep) starts (a multiple of 16).  Round 5: when the interval and
// MS_WINDOW_MARGIN bytes on either side lie inside ONE 128-byte line of the array, the window IS that line -- one memory
// request; otherwise it starts 48..63 bytes before sp as in rounds 3-4, which straddles two lines seven times out of eight
// (the PMC pass of round 5 found 1.9 requests per window: 32 M of the kernel's 263 M reads; profiles/r05_match_stats.md).
#ifndef MS_WINDOW_MARGIN
#define MS_WINDOW_MARGIN 24
#endif
__device__ __forceinline__ u64 lcp_window_start(u64 sp, u64 ep)
{
  const u32 off = u32(sp) & 127;
  const u64 width = ep - sp;                                  // (a range wider than the window is not decided from it anyway)
  if(off >= MS_WINDOW_MARGIN && off + width + 1 + MS_WINDOW_MARGIN <= 128) { return sp & ~u64(127); }
  const u64 unit = sp >> 4;
  return (unit >= 3 ? unit - 3 : 0) << 4;
}

__device__ __forceinline__ bool parent_from_window(const ulonglong2* wave_stage, u32 lane, u64 wstart, u64 lcp_size, u64 sp, u64 ep,
                                                   gcsa2_stnode& out)
{
  if(sp == 0 || ep + 2 >= lcp_size || ep + 1 >= wstart + 128) { return false; }
  const u32 lo = u32(sp - wstart), ro = u32(ep + 1 - wstart);              // byte offsets of LCP[sp], LCP[ep + 1] in the window
  const u64 lw = staged_word(wave_stage, lane, lo >> 3), rw = staged_word(wave_stage, lane, ro >> 3);
  const u32 left_lcp = u32(lw >> (8 * (lo & 7))) & 0xFF, right_lcp = u32(rw >> (8 * (ro & 7))) & 0xFF;
  const u32 node_lcp = (left_lcp > right_lcp ? left_lcp : right_lcp);
  u64 lpos = sp, lval = left_lcp, rpos = ep + 1, rval = right_lcp;
  bool decided = true;
  if(left_lcp == node_lcp)
  {
    u32 w = lo >> 3;
    u64 word = lw;
    u64 flags = below_flags(word, left_lcp) & ((u64(1) << (8 * (lo & 7))) - 1);          // the bytes before LCP[sp] in its word
    while(flags == 0 && w > 0) { w--; word = staged_word(wave_stage, lane, w); flags = below_flags(word, left_lcp); }
    if(flags == 0) { decided = false; }
    else { const u32 byte = (63u - u32(__clzll((long long)flags))) >> 3; lpos = wstart + 8 * w + byte; lval = (word >> (8 * byte)) & 0xFF; }
  }
  if(right_lcp == node_lcp)
  {
    // (the positions up to ep + 2 exist; bytes at or past lcp_size belong to the tree and are masked)
    const u32 last = (wstart + 128 <= lcp_size ? 15u : u32((lcp_size - 1 - wstart) >> 3));
    u32 w = ro >> 3;
    u64 word = rw;
    u64 flags = below_flags(word, right_lcp) & ((ro & 7) == 7 ? u64(0) : ~u64(0) << (8 * ((ro & 7) + 1)));   // the bytes behind LCP[ep + 1]
    while(flags == 0 && w < last) { w++; word = staged_word(wave_stage, lane, w); flags = below_flags(word, right_lcp); }
    if(w == last && wstart + 8 * w + 8 > lcp_size) { flags &= (u64(1) << (8 * (lcp_size - wstart - 8 * w))) - 1; }
    if(flags == 0) { decided = false; }
    else { const u32 byte = (u32(__ffsll((long long)flags)) - 1) >> 3; rpos = wstart + 8 * w + byte; rval = (word >> (8 * byte)) & 0xFF; }
  }
  out = gcsa2_stnode{lpos, rpos - 1, lval, rval, node_lcp};
  return decided;
}

// parent() of (sp, ep) from the eight LCP bytes on either side of the interval -- the bytes ending at LCP[sp] and the bytes
// starting at LCP[ep + 1], four independent LDS reads and no loop: after a failed step the interval is a few path nodes wide
// and the nearest smaller values lie one to three positions away (the general scan of the whole window, parent_from_window,
// kernels_lcp.hpp, runs its two word-by-word loops as long as the slowest lane needs: half of this kernel's time when a dozen
// lanes of a wave take it in every round).  false: not decidable from those bytes -- the caller tries the whole window.
__device__ __forceinline__ bool parent_near(const ulonglong2* wave_wstage, u32 slot, u64 wstart, u64 lcp_size, u64 sp, u64 ep, gcsa2_stnode& out)
{
  if(sp < wstart + 8 || ep + 10 >= wstart + 128 || ep + 10 >= lcp_size) { return false; }
  const u32 lo = u32(sp - wstart), ro = u32(ep + 1 - wstart);
  const u32 lw = lo >> 3, ls = lo & 7, rw = ro >> 3, rs = ro & 7;
  const u64 l1 = staged_word(wave_wstage, slot, lw), l0 = staged_word(wave_wstage, slot, lw - 1);
  const u64 r0 = staged_word(wave_wstage, slot, rw), r1 = staged_word(wave_wstage, slot, rw + 1);
  const u64 L = (l1 << (8 * (7 - ls))) | (ls == 7 ? u64(0) : l0 >> (8 * (ls + 1)));      // byte 7 = LCP[sp], byte 6 = LCP[sp - 1], ...
  const u64 R = (r0 >> (8 * rs)) | (rs == 0 ? u64(0) : r1 << (8 * (8 - rs)));            // byte 0 = LCP[ep + 1], byte 1 = LCP[ep + 2], ...
  const u32 left_lcp = u32(L >> 56), right_lcp = u32(R) & 0xFF;
  const u32 node_lcp = (left_lcp > right_lcp ? left_lcp : right_lcp);
  u64 lpos = sp, lval = left_lcp, rpos = ep + 1, rval = right_lcp;
  bool decided = true;
  if(left_lcp == node_lcp)
  {
    const u64 flags = below_flags(L, left_lcp) & 0x0080808080808080ull;
    if(flags == 0) { decided = false; }
    else { const u32 byte = (63u - u32(__clzll((long long)flags))) >> 3; lpos = sp - (7 - byte); lval = (L >> (8 * byte)) & 0xFF; }
  }
  if(right_lcp == node_lcp)
  {
    const u64 flags = below_flags(R, right_lcp) & 0x8080808080808000ull;
    if(flags == 0) { decided = false; }
    else { const u32 byte = (u32(__ffsll((long long)flags)) - 1) >> 3; rpos = ep + 1 + byte; rval = (R >> (8 * byte)) & 0xFF; }
  }
  out = gcsa2_stnode{lpos, rpos - 1, lval, rval, node_lcp};
  return decided;
}

// ---- matching statistics, version 2: wave-cooperative block fetch, two characters per step, batched parent() ----
// Same results as k_match_stats.  One lane = one pattern; the LF steps of the 64 patterns of a wave go through the
// cooperative fetch of k_find2 (one 128-byte request per endpoint, FLP128 pair blocks when the next two
// characters are fast characters and the block proves that neither step empties -- both matching statistics
// then follow at once: depth + 1 and depth + 2).  A lane whose step empties spends its next round on LCPArray::parent
// (lcp.cpp:276-301) -- the 128 bytes of the LCP array around its range travel through the same cooperative fetch as the other
// lanes' blocks and decide the call in all but a few per cent of the cases (parent_from_window) -- and retries the character
// in the round after that.
// Round 4: the blocks and LCP windows arrive through gfx950's direct global -> LDS loads (fetch_blocks_issue / _wait,
// kernels_find.hpp), and the rounds are software-pipelined: the requests of round k + 1 leave as soon as the outcome of round k
// is known; the result stores, the break records, finished and new patterns and the window refill run under them.
// After a step that needed parent() the next COOL_DOWN characters are stepped singly: right after a mismatch the match is
// short and the following characters fail often, so a pair attempt mostly wastes its round (deep suffix tree, 37 parent()
// calls per pattern: 60 -> 68 M patterns/s with 6; 3 / 12 / 24 give 67 / 67 / 66; profiles/r02_config5.md).
constexpr u32 COOL_DOWN = 3;
constexpr u32 MS_REFILL_AT = 8;                         // persistent lanes: idle lanes of a wave that trigger a refill
constexpr u64 MS_REFILL_MIN = u64(1) << 19;             // host-pointer API: smallest ragged batch sent to the persistent lanes

// REFILL = true (variant 5 of gcsa2_match_stats_device_variant): a lane whose pattern is finished draws the next one from
// a global counter (`queue`, zeroed by the host) as soon as `refill_at` lanes of its wave are idle, and the grid is what the
// device holds at once.  For batches of ragged lengths: 2 M patterns of 32..256 bp run at 116 M patterns/s against 85 M/s
// with a lane per pattern.  Batches of equal lengths lose (174 -> 141 M/s on the 256-bp batch of config 5): starting a
// pattern costs its wave three dependent loads, paid once per wave with a lane per pattern and ~8 times here; mixing clean
// and mismatching patterns of one length gains nothing either, a round costs a wave the same whether 32 or 64 lanes take it
// (profiles/r02_config5.md).
// PROF = true (gcsa2_match_stats_profile_device, a diagnostic): shader-clock cycles per phase of the round, summed over the
// waves, and event counts, added to prof[0..15]: 0 statistics stores + loop head (records, finished / new patterns) + window
// refill -- all under the requests in flight since round 4 --, 1 step setup + issue of the next round's requests, 2 the wait
// for the requests, 3 first evaluation, 4 second fetch + evaluation, 5 outcome, 6 parent() from the LCP window, 7 parent() tree
// walk; 8 rounds (per wave),
// 9 rounds with a second fetch, 10 lane steps, 11 lane pair attempts, 12 failed pair attempts, 13 parent() calls, 14 tree walks,
// 15 lane second fetches.
// BREAKS = true (gcsa2_match_breaks_device): instead of one statistic per pattern position the kernel reports the BREAK POINTS
// -- the left-maximal matches, what a MEM finder collects: every position p whose match P[p, p + len) cannot be extended
// by P[p - 1] (LF empties and parent() is taken, lcp.cpp:276-301; the reference's caller shape: src/algorithms.cpp:146-167),
// with its length and its range, and position 0.  Between two break points the statistics rise by one per position to the
// left, so the dense array follows from them (ms[i] = len - (i - p) for p <= i < the break point to the right); the result
// stores, a third of the dense kernel's memory requests, shrink to one 32-byte record per break.  Records are appended to
// `sink.tmp` -- {pattern | ordinal << 32, position | length << 32, sp, ep}; a wavefront reserves blocks of BREAK_BLOCK slots --
// and put into CSR order by k_breaks_scatter once the per-pattern counts have been scanned; slots beyond `sink.cap` are not
// written (the exact number of records is the sum of the per-pattern counts).  
// The record of a round is taken AFTER the step and after plan_and_issue() (the next round's requests are in flight): a failed
// step leaves (i, depth, sp, ep) untouched and sets need_parent, so at that point they still describe the match that starts
// at position i -- the break.  At i == 0 the pattern's first position is the last record.  A character that does not occur
// fails AT THE ROOT and moves on (i--, `pending`): the break is position i + 1, and when that step also reached i == 0 the
// final record and the end of the pattern are deferred by one round (a lane writes one record per round).  The order matters:
// need_parent, pending and last_break are read after the step; moving the record back to the loop head reports the state of
// the NEXT match.  Exactly the positions p with p == 0 or ms[p - 1] != ms[p] + 1 get a record.
struct BreakSink
{
  u64* tmp; u64 cap; unsigned long long* counter; u32* counts; u32 min_length;
};
constexpr u32 BREAK_WORDS = 4;               // {pattern | ordinal << 32, position | length << 32, sp, ep}
constexpr u32 BREAK_BLOCK = 256;             // record slots a wavefront reserves at a time
constexpr u64 BREAK_HOLE = ~u64(0);          // pattern field of an unused slot (the tail of a wave's last block)

template<bool PAIR, bool REFILL, bool PROF = false, bool BREAKS = false>
__global__ __launch_bounds__(TPB2, 4) void k_match_stats2(DevImage img, const u8* __restrict__ patterns,
                                                       const u64* __restrict__ offsets, u64 nq,
                                                       unsigned short* __restrict__ ms, u64* __restrict__ ranges,
                                                       u64* __restrict__ fallbacks, u32 cool_down,
                                                       unsigned long long* __restrict__ queue, u32 refill_at,
                                                       const ulonglong2* __restrict__ recs,
                                                       unsigned long long* __restrict__ prof = nullptr, BreakSink sink = BreakSink{nullptr, 0, nullptr, nullptr, 0})
{
  [[maybe_unused]] u64 prof_t = 0, prof_c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  [[maybe_unused]] u32 prof_n[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if constexpr(PROF) { prof_t = clock64(); }
#define G2_TICK(phase) do { if constexpr(PROF) { const u64 now_ = clock64(); prof_c[phase] += now_ - prof_t; prof_t = now_; } } while(0)
#define G2_COUNT(slot, value) do { if constexpr(PROF) { prof_n[slot] += u32(value); } } while(0)
  __shared__ ulonglong2 stage[TPB2 * 8];
  __shared__ u64 addr_table[TPB2];          // the block address of every lane, for the eight lanes that fetch it (fetch_blocks_issue)
  __shared__ u8 c2c[256];
  c2c[threadIdx.x] = img.char2comp[threadIdx.x];
  c2c[threadIdx.x + TPB2] = img.char2comp[threadIdx.x + TPB2];
  __syncthreads();
  const u32 lane = threadIdx.x & 63;
  ulonglong2* wave_stage = stage + (threadIdx.x & ~63u) * 8;
  u64* wave_addr = addr_table + (threadIdx.x & ~63u);
  u64 q = 0, begin = 0;
  u32 i = 0, total = 0;                     // characters left / in all (a pattern is shorter than 2^32 characters)
  bool has = false;
  [[maybe_unused]] bool exhausted = false;
  u64 sp = 0, ep = img.n - 1;
  u32 depth = 0, calls = 0;
  bool need_parent = false;
  u32 force_single = 0;
  // Round 5: the pattern as 16-byte RECORDS (k_pack_records above: 2-bit codes + "not a fast character" flags of 32
  // characters).  The lane holds the record of position i - 1 (slot (total - i) & 31), the one behind it (a pair step at
  // slot 31 reads its first character there) and the one behind that, REQUESTED when a record is entered and not looked at
  // before the next entry.  Round 4 re-read two code words and two flag words from two arrays every 24 characters and used
  // them at once: ~44 memory requests per 256-bp pattern (a sixth of the kernel's reads) and, with 64 lanes out of step, a
  // memory latency that the whole wave waited for in nearly every round.
  u64 win_code = 0, next_code = 0, pend_code = 0;
  u32 win_bad = 0, next_bad = ~u32(0), pend_bad = ~u32(0);
  // results: sixteen u16 per flush, as two 16-byte stores back to back (`ms` is 8-byte aligned, the hardware takes the 16-byte
  // store at any dword).  The statistics are a third of the kernel's memory requests -- every lane writes into its own
  // pattern's 512 bytes, nothing coalesces across lanes -- so the only lever is fewer, wider writes per lane (8-byte stores: 64
  // per 256-bp pattern; one 16-byte store per eight positions until round 4; a whole 32-byte sector at once: +3 % on a batch
  // without mismatches, profiles/r04_match_stats.md).
  [[maybe_unused]] u64 packed_lo = 0, packed_hi = 0, packed_2 = 0, packed_3 = 0; [[maybe_unused]] u32 have = 0;
  [[maybe_unused]] u32 last_break = ~u32(0), n_breaks = 0;     // BREAKS: position of the latest record, records of this pattern
  [[maybe_unused]] u64 blk_base = 0; [[maybe_unused]] u32 blk_used = BREAK_BLOCK;    // BREAKS: the wave's block of record slots (uniform)
  // (kept in a vector register on purpose: the kernel has no scalar registers to spare, and as a kernel argument the value was
  // re-read from the argument segment -- an s_load and a full s_waitcnt -- in every round of every wave: +10 % on the batch)
  [[maybe_unused]] u32 min_length = sink.min_length;
  if constexpr(BREAKS) { asm volatile("" : "+v"(min_length)); }
  // (The record buffer's address and capacity stay kernel arguments: they are read only where a record is written.  Holding
  // them in vector registers as well -- 127 VGPRs -- cost the persistent form 2.9 ms on config 5's batch; profiles/r04_match_stats.md.)
  [[maybe_unused]] bool pending = false;                       // BREAKS: a character that does not occur: position i is a break of length 0
  auto emit = [&](u32 pos, u32 value)        // ms[begin + pos] = value; positions arrive in descending order
  {
    if constexpr(BREAKS) { return; }
    // sixteen statistics = one 32-byte sector per flush (two 16-byte stores back to back): a 16-byte store is half a
    // sector, and its other half follows ~40 us later -- the memory side then pays for two partial writes
    const u64 idx = begin + pos;
    const u32 slot = u32(idx & 15), w = slot >> 2;
    const u64 field = u64(value > 65535 ? 65535 : value) << (16 * (slot & 3));
    packed_lo |= (w == 0 ? field : 0); packed_hi |= (w == 1 ? field : 0); packed_2 |= (w == 2 ? field : 0); packed_3 |= (w == 3 ? field : 0);
    have |= 1u << slot;
    if(slot == 0 || pos == 0)
    {
      unsigned short* group = ms + (idx & ~u64(15));
      typedef unsigned long long ull2 __attribute__((ext_vector_type(2), aligned(8)));
      if(have == 0xFFFFu)
      {
        *reinterpret_cast<ull2*>(group) = ull2{packed_lo, packed_hi};
        *reinterpret_cast<ull2*>(group + 8) = ull2{packed_2, packed_3};
      }
      else
      {
        const u64 words[4] = {packed_lo, packed_hi, packed_2, packed_3};
#pragma unroll
        for(u32 k = 0; k < 4; k++)
        {
          const u32 h = (have >> (4 * k)) & 0xF;
          if(h == 0xF) { *reinterpret_cast<u64*>(group + 4 * k) = words[k]; }
          else { for(u32 t = 0; t < 4; t++) { if((h >> t) & 1) { group[4 * k + t] = (unsigned short)(words[k] >> (16 * t)); } } }
        }
      }
      packed_lo = 0; packed_hi = 0; packed_2 = 0; packed_3 = 0; have = 0;
    }
  };
  auto start = [&](u64 query)
  {
    q = query; has = true;
    begin = offsets[q]; i = total = u32(offsets[q + 1] - begin);
    sp = 0; ep = img.n - 1; depth = 0; calls = 0; need_parent = false; force_single = 0;
    if constexpr(BREAKS) { last_break = ~u32(0); n_breaks = 0; pending = false; }
    {
      const u64 word = (begin >> 5) + q;
      const ulonglong2 r0 = recs[word], r1 = recs[word + 1], r2 = recs[word + 2];
      win_code = r0.x; win_bad = u32(r0.y); next_code = r1.x; next_bad = u32(r1.y); pend_code = r2.x; pend_bad = u32(r2.y);
    }
    // The k-mer seed table (find() of every k-mer over the fast characters, kernels_find.hpp): when the pattern's last k
    // characters are fast characters and occur, the search starts behind them -- all k suffixes match, so their statistics
    // are 1 .. k -- and skips the steps on the widest ranges, whose endpoints lie in different blocks.  An empty or wide
    // entry starts from scratch.
    const u32 k = img.kmer_k;
    if(k > 0 && total >= k && img.n > 0)
    {
      const u64 tix = win_code & ((u64(1) << (2 * k)) - 1);
      const bool fast = (win_bad & ((1u << k) - 1)) == 0;
      const u64 entry = img.kmer_table[fast ? tix : 0];
      const u64 width = entry >> SEED_SP_BITS;
      if(fast && width != 0 && width != SEED_WIDE)
      {
        sp = entry & SEED_SP_MASK; ep = sp + width - 1;
        for(u32 j = 0; j < k; j++) { emit(total - 1 - j, j + 1); }
        depth = k; i = total - k;            // (k <= 16: still inside record 0)
      }
    }
  };
  // `adv` characters were consumed (i is already lowered): entering the next record makes it the current one and requests
  // the one behind the next -- needed 32 characters from now
  auto consumed = [&](u32 adv)
  {
    const u32 t = total - i;
    if((t & 31) < adv)
    {
      win_code = next_code; win_bad = next_bad; next_code = pend_code; next_bad = pend_bad;
      const ulonglong2 r = recs[(begin >> 5) + q + (t >> 5) + 2];
      pend_code = r.x; pend_bad = u32(r.y);
    }
  };
  if constexpr(!REFILL)
  {
    const u64 gid = u64(blockIdx.x) * TPB2 + threadIdx.x;
    if(gid < nq) { start(gid); }
  }
  // Software-pipelined rounds (round 4; on the direct global -> LDS fetch, fetch_blocks_issue / _wait): as soon as a round's outcome is known the
  // NEXT round's blocks are requested, and everything that does not feed that request -- the result stores, the break records,
  // finished patterns, new patterns for idle lanes, the window refill -- runs while the requests are in flight.  A lane that
  // starts a pattern joins with the following request (one round late).  The plan of a request (what was asked for, and where
  // in the block the answer lies) lives across the loop edge in the registers the direct fetch no longer needs.
  bool planned = false, pair = false;       // planned: this lane has a block (or LCP window) in flight
  u32 comp = 0, r_sp = 0, r_ep = 0, idx_sp = 0, idx_ep = 0, emit_code = 0;
  u64 wstart = 0;
  auto plan_and_issue = [&]()
  {
    const bool active = has && i > 0;
    planned = active;
    const bool stepping = active && !need_parent, parenting = active && need_parent;
    comp = 0; r_sp = 0; r_ep = 0; idx_sp = 0; idx_ep = 0; pair = false; wstart = 0;
    if(parenting)
    {
      // parent(): the 128 bytes of the LCP array around the range, through the same cooperative fetch as the blocks
      wstart = lcp_window_start(sp, ep);
      idx_sp = idx_ep = u32(wstart >> 4) | LCP_FLAG;
    }
    if(stepping)
    {
      const u32 r = (total - i) & 31;                          // slot of position i - 1 in the current record
      const u32 flags = ((win_bad >> r) & 3) | (r == 31 ? (next_bad & 1) << 1 : 0u);
      if constexpr(PAIR)
      {
        if(force_single == 0 && i >= 2)
        {
          pair = (flags == 0);
          if(pair)
          {
            const u32 c2 = u32(win_code >> (2 * r)) & 3, c1 = (r == 31 ? u32(next_code) : u32(win_code >> (2 * r + 2))) & 3;
            u32 b_sp, b_ep;
            pair_block_of(sp, b_sp, r_sp); pair_block_of(ep + 1, b_ep, r_ep);
            const u32 first = (c1 * 4 + c2) * u32(img.flp_nblocks);
            idx_sp = (first + b_sp) | PAIR_FLAG; idx_ep = (first + b_ep) | PAIR_FLAG;
          }
        }
      }
      if(!pair)
      {
        if(flags & 1)
        {
          const u64 addr = reinterpret_cast<u64>(patterns) + begin + i - 1;
          comp = c2c[u32(*reinterpret_cast<const u64*>(addr & ~u64(7)) >> ((addr & 7) * 8)) & 0xFF];
        }
        else { comp = 1 + (u32(win_code >> (2 * r)) & 3); }
        u32 b_sp, b_ep;
        flb_block_of(sp, b_sp, r_sp); flb_block_of(ep + 1, b_ep, r_ep);
        idx_sp = comp * u32(img.flb_nblocks) + b_sp; idx_ep = comp * u32(img.flb_nblocks) + b_ep;
      }
    }

    if(__any(active)) { fetch_blocks_issue<PAIR, true>(img.flb, idx_sp, active, wave_stage, lane, img.flp, img.lcp, wave_addr); }
  };
  plan_and_issue();
  while(true)
  {
    const bool active = planned;
    const bool stepping = active && !need_parent, parenting = active && need_parent;
    PairEnd p_sp = {0, 0, 0}, p_ep = {0, 0, 0};                // a single step keeps (edge, node) in .raw / .node
    const bool need2 = stepping && idx_ep != idx_sp;
    gcsa2_stnode node;
    bool decided = false;
    emit_code = 0;
    G2_COUNT(0, lane == 0); G2_COUNT(2, stepping); G2_COUNT(3, pair); G2_COUNT(7, need2);
    if(__any(active))
    {
      fetch_blocks_wait();
      if constexpr(PROF) { if(active) { asm volatile("" :: "v"(wave_stage[lane * 8 + (lane & 7)].x)); } }     // the fetch has landed
      G2_TICK(2);
      if(stepping)                               // one evaluation for single and pair steps alike (eval_staged)
      {
        p_sp = eval_staged(wave_stage, lane, PAIR && pair, r_sp, false);
        if(idx_ep == idx_sp) { p_ep = eval_staged(wave_stage, lane, PAIR && pair, r_ep, true); }
      }
      G2_TICK(3);
      // (round 5 tried parent_near -- the eight bytes on either side, no loop -- in front of this: 7.18 against 6.81 ms; with ~6
      // parenting lanes per round one of them needs the whole window anyway, and then both forms run.  k_match_stats3 uses it.)
      if(parenting) { decided = parent_from_window(wave_stage, lane, wstart, img.lcp_size, sp, ep, node); G2_COUNT(5, 1); }
      if constexpr(PROF) { if(parenting && decided) { asm volatile("" :: "v"(node.sp)); } }
      G2_TICK(6);
      if(__any(need2))
      {
        G2_COUNT(1, lane == 0);
        __builtin_amdgcn_wave_barrier();
        fetch_blocks_direct<PAIR>(img.flb, idx_ep, need2, wave_stage, lane, img.flp, nullptr, wave_addr);
        if(need2) { p_ep = eval_staged(wave_stage, lane, PAIR && pair, r_ep, true); }
      }
      __builtin_amdgcn_wave_barrier();
      G2_TICK(4);
    }
    if(stepping)
    {
      if(PAIR && pair)
      {
        u64 a = 0, b = 0;
        if(pair_outcome(p_sp, p_ep, idx_ep == idx_sp, a, b) == 2)          // neither step empties (layout.hpp)
        {
          sp = p_sp.node; ep = p_ep.node;
          emit_code = 2;
          depth += 2; i -= 2;
        }
        else { force_single = 2; G2_COUNT(4, 1); }             // an emptying step needs parent(): one character at a time
      }
      else
      {
        const u64 a = p_sp.raw, b = p_ep.raw - 1;              // gcsa.h:155-162
        if(!range_empty(a, b))
        {
          sp = p_sp.node; ep = p_ep.node; depth++;
          emit_code = 1; i--;
          force_single -= (force_single > 0 ? 1 : 0);
        }
        else if(sp == 0 && ep == img.n - 1)                    // at the root: no such character
        {
          depth = 0;
          if constexpr(BREAKS) { pending = true; }
          emit_code = 1; i--;
          force_single -= (force_single > 0 ? 1 : 0);
        }
        else { need_parent = true; force_single = (force_single > cool_down ? force_single : cool_down); }   // parent() in the next round
      }
    }
    G2_TICK(5);
    if(parenting)
    {
      if(!decided) { lcp_parent(img, sp, ep, node); G2_COUNT(6, 1); }      // the interval reaches beyond the window: tree walk (lcp.cpp:276-301)
      calls++;
      sp = node.sp; ep = node.ep; depth = u32(node.node_lcp);
      need_parent = false;
    }
    G2_TICK(7);
    if(emit_code != 0) { consumed(emit_code); }                 // (one site: the characters this round consumed, 0..2)
    plan_and_issue();                                            // the next round's requests leave here
    G2_TICK(1);
    if constexpr(!BREAKS)                                        // the statistics of the step just taken (positions i + 1 / i after the update)
    {
      if(emit_code == 2) { emit(i + 1, depth - 1); }
      if(emit_code != 0) { emit(i, depth); }
    }
    [[maybe_unused]] bool was_pending = false;
    if constexpr(BREAKS)
    {
      // The match that starts at position `pos` cannot be extended to the left: after a failed step (need_parent: pos = i, the
      // state is still the one of that match), at the pattern's start (i == 0), or -- pending -- when the character before a
      // match of length 0 does not occur either (the step failed AT THE ROOT and moved on: the break is the position behind,
      // i + 1; the final record and the end of the pattern then wait one round, a lane writes one record per round).
      was_pending = pending;
      const u32 pos = (was_pending ? i + 1 : i);
      const bool is_break = has && total > 0 && last_break != pos && (was_pending ? pos < total : (i == 0 || need_parent));
      if(is_break) { last_break = pos; }
      const bool record = is_break && depth >= min_length;          // (a MEM finder's minimum length: shorter matches are not reported)
      const u64 writers = __ballot(record);
      if(writers != 0)                                         // uniform
      {
        // slots: the wavefront owns a block of BREAK_BLOCK records of `sink.tmp` at a time (one atomic on the global counter
        // per block; one per round was 9 M same-address atomics for config 5's batch: 45 of the kernel's 53 ms).  The writers
        // of a round that do not fit the current block go to the front of the next one: only a wave's last block has a hole.
        const u32 count = u32(__popcll(writers)), room = BREAK_BLOCK - blk_used;
        u64 next_base = 0;
        if(count > room)
        {
          const u32 leader = u32(__ffsll((long long)writers)) - 1;
          unsigned long long got = 0;
          if(lane == leader) { got = atomicAdd(sink.counter, (unsigned long long)BREAK_BLOCK); }
          next_base = __shfl(got, leader, 64);
        }
        if(record)
        {
          const u32 rank = u32(__popcll(writers & ((u64(1) << lane) - 1)));
          const u64 at = (rank < room ? blk_base + blk_used + rank : next_base + (rank - room));
          if(at < sink.cap)
          {
            ulonglong2* dst = reinterpret_cast<ulonglong2*>(sink.tmp + at * BREAK_WORDS);
            dst[0] = make_ulonglong2(q | (u64(n_breaks) << 32), u64(pos) | (u64(depth) << 32));
            dst[1] = make_ulonglong2(sp, ep);
          }
          n_breaks++;
        }
        if(count > room) { blk_base = next_base; blk_used = count - room; } else { blk_used += count; }
      }
      pending = false;
    }
    if(has && i == 0 && !was_pending)                          // pattern finished (or empty): final range, parent() count
    {
      reinterpret_cast<ulonglong2*>(ranges)[q] = make_ulonglong2(sp, ep);
      if(fallbacks != nullptr) { fallbacks[q] = calls; }
      if constexpr(BREAKS) { sink.counts[q] = n_breaks; }
      has = false;
    }
    if constexpr(REFILL)
    {
      const u64 idle = __ballot(!has);
      if(!exhausted && u32(__popcll(idle)) >= refill_at)
      {
        const u32 want = u32(__popcll(idle)), leader = u32(__ffsll((long long)idle)) - 1;
        unsigned long long base = 0;
        if(lane == leader) { base = atomicAdd(queue, (unsigned long long)want); }
        base = __shfl(base, leader, 64);
        if(!has)
        {
          const u64 mine = base + __popcll(idle & ((u64(1) << lane) - 1));
          if(mine < nq) { start(mine); }
        }
        exhausted = (base + want >= nq);
      }
      if(!__any(has)) { if(exhausted) { break; } continue; }
    }
    else
    {
      if(!__any(has)) { break; }
    }
    G2_TICK(0);
  }
  if constexpr(BREAKS)
  {
    for(u32 j = blk_used + lane; j < BREAK_BLOCK; j += 64)     // the unused tail of the wave's last block (none if it never wrote)
    {
      const u64 at = blk_base + j;
      if(at < sink.cap) { sink.tmp[at * BREAK_WORDS] = BREAK_HOLE; }
    }
  }
  if constexpr(PROF)
  {
#pragma unroll
    for(int k = 0; k < 8; k++)
    {
      u64 events = prof_n[k];
      for(int o = 32; o > 0; o >>= 1) { events += __shfl_down(events, o, 64); }
      if(lane == 0) { atomicAdd(prof + k, (unsigned long long)prof_c[k]); atomicAdd(prof + 8 + k, (unsigned long long)events); }
    }
  }
#undef G2_TICK
#undef G2_COUNT
}

// the appended break records into CSR order: record j of pattern q at offsets[q] + j as {position, length, sp, ep}
__global__ __launch_bounds__(TPB) void k_breaks_scatter(const u64* __restrict__ tmp, u64 stored, const u64* __restrict__ offsets,
                                                        u64* __restrict__ out, u64 capacity)
{
  const u64 r = u64(blockIdx.x) * TPB + threadIdx.x;
  if(r >= stored) { return; }
  const ulonglong2* src = reinterpret_cast<const ulonglong2*>(tmp + r * BREAK_WORDS);
  const ulonglong2 a = src[0], b = src[1];
  if(a.x == BREAK_HOLE) { return; }
  const u64 dest = offsets[a.x & 0xFFFFFFFFull] + (a.x >> 32);
  if(dest >= capacity) { return; }
  ulonglong2* dst = reinterpret_cast<ulonglong2*>(out + 4 * dest);
  dst[0] = make_ulonglong2(a.y & 0xFFFFFFFFull, a.y >> 32);
  dst[1] = b;
}

__global__ void k_copy_word(const u64* __restrict__ src, u64* __restrict__ dst) { *dst = *src; }

__global__ __launch_bounds__(TPB) void k_widen_counts(const u32* __restrict__ counts, u64 nq, u64* __restrict__ wide)
{
  const u64 q = u64(blockIdx.x) * TPB + threadIdx.x;
  if(q <= nq) { wide[q] = (q < nq ? u64(counts[q]) : 0); }
}

}  // namespace
