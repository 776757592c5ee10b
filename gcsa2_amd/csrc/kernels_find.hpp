// kernels_find.hpp -- find() / LF: the k-mer seed table, k_find2 and k_lf2 over fused 128-byte blocks, LF(path_node), LF_fast / LF_all.
// Part of the single translation unit gcsa2_hip.hip (device code, anonymous namespace).
#pragma once

#include "kernels_common.hpp"

using namespace g2;

namespace {

// ---- k-mer seed table ----------------------------------------------------------------------
// table[t] = find() of the k-mer whose j-th character FROM THE END has comp 1 + ((t >> 2j) & 3):
// the exact (sp, ep) the backward search returns, including edge-space empty ranges, so that
// k_find2 can start a pattern whose last k characters are all fast characters at step k.
// Pure memoisation of gcsa.h:96-110; results are unchanged.
//
// One entry = 8 bytes: sp in bits [0, 40), len = ep + 1 - sp in bits [40, 64).  An empty range has
// len = 0 (LF-produced empties and charRange of an absent character are always (x, x - 1), utils.h:93-96);
// len = SEED_WIDE marks a range of `wide` or more path nodes, which is searched from scratch instead (what gcsa.h:96-110
// does for every pattern).  wide = img.seed_wide = 2^24 - 1, the largest length the field holds; tests lower it at create
// time (GCSA2_SEED_WIDE) so that the marked entries -- on a large index only those of the few shortest k-mers -- are met.
constexpr u64 SEED_SP_BITS = 40, SEED_SP_MASK = (u64(1) << SEED_SP_BITS) - 1, SEED_WIDE = (u64(1) << 24) - 1;
__device__ __forceinline__ u64 seed_pack(u64 sp, u64 ep, u64 wide)
{
  const u64 len = ep + 1 - sp;
  return (sp & SEED_SP_MASK) | ((len < wide ? len : SEED_WIDE) << SEED_SP_BITS);
}

// The table is built in place, level by level: the (j + 1)-mer t extends the j-mer t & (4^j - 1) by the character
// 1 + (t >> 2j) in front, i.e. by one LF step (gcsa.h:103-107).  Launch 1 of a level writes the three quarters
// with a non-zero leading code (reading level j, untouched), launch 2 extends the entries of level j themselves.
// A wide entry is recomputed from scratch (only the top levels of a large index have such ranges).
__global__ __launch_bounds__(TPB) void k_seed_level(DevImage img, u32 j, u64 first, u64 last, u64* __restrict__ table)
{
  const u64 tix = first + u64(blockIdx.x) * TPB + threadIdx.x;
  if(tix >= last) { return; }
  u64 sp, ep;
  if(j == 0) { const u32 c = 1 + u32(tix & 3); sp = img.crange[2 * c]; ep = img.crange[2 * c + 1]; }
  else
  {
    const u64 prev = table[tix & ((u64(1) << (2 * j)) - 1)];
    u32 from = j;
    sp = prev & SEED_SP_MASK; ep = sp + (prev >> SEED_SP_BITS) - 1;
    if((prev >> SEED_SP_BITS) == SEED_WIDE)
    {
      const u32 c = 1 + u32(tix & 3);
      sp = img.crange[2 * c]; ep = img.crange[2 * c + 1]; from = 1;
    }
    for(u32 s = from; s <= j && !range_empty(sp, ep); s++)
    {
      const u32 comp = 1 + u32((tix >> (2 * s)) & 3);
      DevBV bv = bwt_of(img, comp);
      u64 ra, rb;
      bv_rank2(bv, sp, ep + 1, ra, rb);
      sp = img.C[comp] + ra; ep = img.C[comp] + rb - 1;
      if(range_empty(sp, ep)) { break; }
      path_node_range(img, sp, ep);
    }
  }
  table[tix] = seed_pack(sp, ep, img.seed_wide);
}

// ---- find, version 2: fused 128-byte LF blocks, wave-cooperative fetch through LDS -----------
//
// One lane = one pattern, 64 patterns per wave walk their LF chains in lockstep.  Per step the
// wave fetches the 64 fused blocks its lanes need with 8 line-coalesced instructions (8 adjacent
// lanes x 16 bytes = one 128-byte block per request), stages them in LDS (XOR-swizzled so that the
// ds_read_b128 read-back is conflict free) and every lane then evaluates its own
// C[c] + rank(B_c, .) and rank(edges, .) from the staged block.  A second fetch round runs only
// for lanes whose sp and ep + 1 fall into different blocks.
constexpr int TPB2 = 128;          // 2 waves: 16 KB of staging + tables -> 9 workgroups / CU

struct Tables2
{
  u64 crange[2 * MAX_SIGMA];
  u8 c2c[256];
};

struct Endpoint { u64 edge; u64 node; u32 ones; };

// A word of the staged block as an opaque register value: selecting among such values compiles to v_cndmask.  Without
// it LLVM folds a select of array elements into a dynamically indexed load, and the block array moves to scratch memory
// (measured: find() 2.1x slower).
__device__ __forceinline__ u64 in_register(u64 x) { asm volatile("" : "+v"(x)); return x; }

// lane-private evaluation of one LF endpoint from a staged fused block (FLB128, layout.hpp)
//   blk = 8 x ulonglong2 (w0..w15), r = bit offset inside the block (< 384)
//   edge = C[c] + rank(B_c, i);  node = rank(edges, edge - back) with back = 0 (sp) or 1 (ep)
__device__ __forceinline__ void eval_endpoint(const ulonglong2 (&blk)[8], u32 r, u32 back, u64& edge, u64& node)
{
  const u32 wq = r >> 6;                                                                   // 0..5
  const u64 part = (u64(1) << (r & 63)) - 1;
  const u64 b0 = in_register(blk[1].x), b1 = in_register(blk[1].y), b2 = in_register(blk[2].x),
            b3 = in_register(blk[2].y), b4 = in_register(blk[3].x), b5 = in_register(blk[3].y);           // w2..w7
  const u64 bword = (wq < 3 ? (wq == 0 ? b0 : (wq == 1 ? b1 : b2)) : (wq == 3 ? b3 : (wq == 4 ? b4 : b5)));
  const u32 ones = (wq == 0 ? 0u : u32(blk[7].x >> (10 * (wq - 1))) & 0x3FF) + u32(__popcll(bword & part));
  edge = blk[0].x + ones;
  const u64 ncnt = blk[0].y & ~PREV_BIT;
  if(back > ones) { node = ncnt - (blk[0].y >> 63); return; }                              // rank(edges, ecnt - 1)
  const u32 k = ones - back, kq = k >> 6;                                                  // 0..6 (6: the whole slice)
  const u64 kpart = (u64(1) << (k & 63)) - 1;
  const u64 e0 = in_register(blk[4].x), e1 = in_register(blk[4].y), e2 = in_register(blk[5].x),
            e3 = in_register(blk[5].y), e4 = in_register(blk[6].x), e5 = in_register(blk[6].y);           // w8..w13
  const u64 eword = (kq < 3 ? (kq == 0 ? e0 : (kq == 1 ? e1 : e2)) : (kq == 3 ? e3 : (kq == 4 ? e4 : (kq == 5 ? e5 : u64(0)))));
  node = ncnt + (kq == 0 ? 0u : u32(blk[7].y >> (10 * (kq - 1))) & 0x3FF) + u32(__popcll(eword & kpart));
}

// lane-private evaluation of one endpoint of a TWO-character step from a staged FLP128 block (layout.hpp):
//   raw = H(i) = ecnt + rank(P, r);  sp side (ep_side = false): node = N(raw)
//   ep side: edge b'' = raw - 1 + D[r], node = N(b''), dbit = D[r]
//   qbefore = rank(Q, r) inside the block, qafter = Q bits at or after r in the block
//   (the three small values share one register: qbefore | qafter << 8 | dbit << 16, each count <= 192)
struct PairEnd
{
  u64 raw, node; u32 bits;
  __device__ __forceinline__ u32 qbefore() const { return bits & 0xFF; }
  __device__ __forceinline__ u32 qafter() const { return (bits >> 8) & 0xFF; }
  __device__ __forceinline__ u32 dbit() const { return bits >> 16; }
};

__device__ __forceinline__ PairEnd eval_pair(const ulonglong2 (&blk)[8], u32 r, bool ep_side)
{
  const u64 w[16] = { blk[0].x, blk[0].y, blk[1].x, blk[1].y, blk[2].x, blk[2].y, blk[3].x, blk[3].y,
                      blk[4].x, blk[4].y, blk[5].x, blk[5].y, blk[6].x, blk[6].y, blk[7].x, blk[7].y };
  const u32 wq = r >> 6;
  const u64 part = (u64(1) << (r & 63)) - 1;
  u32 ones = 0, qb = 0, qt = 0;
  u64 dword = w[5];
#pragma unroll
  for(u32 j = 0; j < PAIR_WORDS; j++)
  {
    const u64 m = (j < wq ? ~u64(0) : (j == wq ? part : u64(0)));
    ones += __popcll(w[2 + j] & m);
    qb += __popcll(w[8 + j] & m); qt += __popcll(w[8 + j]);
    if(j == wq) { dword = w[5 + j]; }
  }
  PairEnd out;
  const u32 dbit = (ep_side ? u32((dword >> (r & 63)) & 1) : 0u);
  out.raw = w[0] + ones; out.bits = qb | ((qt - qb) << 8) | (dbit << 16);
  const u64 ncnt = w[1] & ~PREV_BIT;
  const u32 back = (ep_side ? 1u - dbit : 0u);
  if(back > ones) { out.node = ncnt - (w[1] >> 63); return out; }    // rank(edges, ecnt - 1)
  const u32 k = ones - back, kq = k >> 6;
  const u64 kpart = (u64(1) << (k & 63)) - 1;
  u32 cnt = 0;
#pragma unroll
  for(u32 j = 0; j < 4; j++)
  {
    const u64 m = (j < kq ? ~u64(0) : (j == kq ? kpart : u64(0)));
    cnt += __popcll(w[11 + j] & m);
  }
  out.node = ncnt + cnt;
  return out;
}

// What a fused pair of steps decides (layout.hpp): 2 = both steps non-empty, take (node_sp, node_ep);
// 1 = the second step is empty, (a, b) = its edge-space integers; 0 = replay the two steps singly.
__device__ __forceinline__ u32 pair_outcome(const PairEnd& s, const PairEnd& e, bool same_block, u64& a, u64& b)
{
  if(e.raw > s.raw) { return 2; }
  const bool first_nonempty = (same_block ? e.qbefore() > s.qbefore() : (e.qbefore() > 0 || s.qafter() > 0));
  if(!first_nonempty) { return 0; }
  if(e.dbit()) { return 2; }
  a = s.raw; b = s.raw - 1;
  return 1;
}

// LF_fast / LF_all (src/gcsa.cpp:742-798) of one range for the N comps c0 .. c0 + N - 1 (those <= limit),
// one lane per range.  The N first-block loads are issued unconditionally (an inactive comp reads block 0)
// so that they are all in flight together; children[j] is the node-space range when it is non-empty,
// otherwise what the reference leaves there: Range::empty_range() = (1, 0) for an empty or single-node
// input, the edge-space pair in the general case.
template<int N>
__device__ __forceinline__ void lf_children(const DevImage& img, u32 c0, u32 limit, bool live, u64 sp0, u64 ep0,
                                            u64 (&csp)[N], u64 (&cep)[N])
{
  const bool nonempty = live && !range_empty(sp0, ep0);
  const u64 sp = nonempty ? sp0 : 0, e1 = nonempty ? ep0 + 1 : 0;
  const u64 b_sp = sp / FLB_BITS, b_ep = e1 / FLB_BITS;
  ulonglong2 blk[N][8];
#pragma unroll
  for(int j = 0; j < N; j++)
  {
    const u32 c = c0 + u32(j);
    const u64 idx = (nonempty && c <= limit ? u64(c) * img.flb_nblocks + b_sp : 0);
    const ulonglong2* src = reinterpret_cast<const ulonglong2*>(img.flb + idx * FLB_WORDS);
#pragma unroll
    for(u32 k = 0; k < 8; k++) { blk[j][k] = src[k]; }
  }
#pragma unroll
  for(int j = 0; j < N; j++)
  {
    const u32 c = c0 + u32(j);
    csp[j] = 1; cep[j] = 0;
    if(!(nonempty && c <= limit)) { continue; }
    u64 a, nsp, e_ep, n_ep;
    eval_endpoint(blk[j], u32(sp - b_sp * FLB_BITS), 0, a, nsp);
    if(b_ep != b_sp)
    {
      const ulonglong2* src = reinterpret_cast<const ulonglong2*>(img.flb + (u64(c) * img.flb_nblocks + b_ep) * FLB_WORDS);
#pragma unroll
      for(u32 k = 0; k < 8; k++) { blk[j][k] = src[k]; }
    }
    eval_endpoint(blk[j], u32(e1 - b_ep * FLB_BITS), 1, e_ep, n_ep);
    const u64 b = e_ep - 1;
    if(!range_empty(a, b)) { csp[j] = nsp; cep[j] = n_ep; }
    else if(sp0 != ep0) { csp[j] = a; cep[j] = b; }
  }
}

// wave-cooperative fetch: every lane with need != 0 gets flb block `idx` staged at its slot; with PAIR an
// index carrying PAIR_FLAG selects block idx & ~PAIR_FLAG of the pair array `flp` instead; with LCPW an index carrying
// LCP_FLAG selects the 128 bytes of the LCP array that start at byte 16 * (idx & ~LCP_FLAG)
//
// (Round 3 also tried to let up to eight second blocks -- endpoints in different blocks: rare per lane, 28 % of the rounds per
// wave -- ride along in extra LDS slots of the first round instead of costing the wave a second round trip.  find() did not
// move, it is bound by the request rate, not by rounds (5.03 G queries/s either way); the matching statistics lost a third
// to the planning step in front of every fetch.  Dropped; profiles/r03_match_stats.md.)
template<bool PAIR = false, bool LCPW = false>
__device__ __forceinline__ void fetch_blocks(const u64* __restrict__ flb, u32 idx, bool need, ulonglong2* wave_stage, u32 lane,
                                             const u64* __restrict__ flp = nullptr, const u8* __restrict__ lcp = nullptr)
{
  u32 sub = lane & 7;
#pragma unroll
  for(u32 j = 0; j < 8; j++)
  {
    // Branch-free on purpose: a load inside `if(need)` makes hipcc wait for it (s_waitcnt
    // vmcnt(0)) before the next one is issued, i.e. 8 serialized round trips per step.  Lanes
    // whose owner needs nothing fetch block 0 (one cached line) into a slot nobody reads.
    u32 owner = 8 * j + (lane >> 3);
    u32 oidx = __shfl(need ? idx : 0u, owner, 64);
    const u64* base = flb;
    u32 unit = FLB_WORDS;                                     // u64 words per index step
    if constexpr(PAIR) { base = (oidx & PAIR_FLAG) ? flp : flb; }
    if constexpr(LCPW) { if(oidx & LCP_FLAG) { base = reinterpret_cast<const u64*>(lcp); unit = 2; } }
    oidx &= ~(PAIR_FLAG | LCP_FLAG);
    ulonglong2 a = reinterpret_cast<const ulonglong2*>(base + u64(oidx) * unit)[sub];
    wave_stage[owner * 8 + (sub ^ (owner & 7))] = a;
  }
  __builtin_amdgcn_wave_barrier();
}

// The same fetch with gfx950's direct global -> LDS loads (global_load_lds_dwordx4; the matching statistics use it, round 4): no
// destination registers -- the eight requests of a wave are in flight without the 32 VGPRs the register form keeps for them
// (k_match_stats2: 123 -> 109 VGPRs), and the request can be ISSUED long before its data is needed (fetch_blocks_issue ...
// fetch_blocks_wait).  find() keeps the register form: same speed there (20.2-20.4 ms either way), it is bound by the request rate.  The instruction writes lane L's
// 16 bytes at M0 + 16 L, i.e. linearly; the XOR swizzle of the slots therefore moves to the SOURCE side: the lane at
// position `sub` of an owner's slot loads chunk sub ^ (owner & 7) of the block, which is the same layout as above.
template<bool PAIR = false, bool LCPW = false>
__device__ __forceinline__ void fetch_blocks_issue(const u64* __restrict__ flb, u32 idx, bool need, ulonglong2* wave_stage, u32 lane,
                                                   const u64* __restrict__ flp, const u8* __restrict__ lcp, u64* wave_addr)
{
  // Every lane works out the address of ITS block once and publishes it in a 64-entry table of the wave in LDS (`wave_addr`);
  // the eight lanes that fetch a block read it from there -- one LDS round trip for the whole fetch.  (Round 4's first form
  // sent the 32-bit index through ds_bpermute and rebuilt the address -- array, unit, 64-bit multiply -- in each of the eight
  // iterations, every one waiting for its own permute: 150 instructions and eight LDS latencies per fetch.)
  const u64* base = flb;
  u32 unit = FLB_WORDS;                                     // u64 words per index step
  if constexpr(PAIR) { base = (idx & PAIR_FLAG) ? flp : flb; }
  if constexpr(LCPW) { if(idx & LCP_FLAG) { base = reinterpret_cast<const u64*>(lcp); unit = 2; } }
  const u64 mine = reinterpret_cast<u64>(need ? base + u64(idx & ~(PAIR_FLAG | LCP_FLAG)) * unit : flb);
  __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0): every read of the slots' previous contents has returned
  __builtin_amdgcn_wave_barrier();
  wave_addr[lane] = mine;
  __builtin_amdgcn_wave_barrier();
  const u32 lds_base = __builtin_amdgcn_readfirstlane(u32(reinterpret_cast<size_t>(wave_stage)));     // LDS address of the wave's slots (uniform)
  const u32 group = lane >> 3;
  const u64 chunk = u64(((lane & 7) ^ (group & 7)) * 16);       // the swizzle on the source side: (8 j + group) & 7 = group & 7
  u64 src[8];
#pragma unroll
  for(u32 j = 0; j < 8; j++) { src[j] = wave_addr[8 * j + group]; }
#pragma unroll
  for(u32 j = 0; j < 8; j++)
  {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + chunk),
                                     (__attribute__((address_space(3))) void*)(size_t(lds_base + j * 1024u)), 16, 0, 0);
  }
}
__device__ __forceinline__ void fetch_blocks_wait()
{
  __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0) on gfx9 encodings: vmcnt = 0, expcnt / lgkmcnt untouched
  __builtin_amdgcn_wave_barrier();
}
template<bool PAIR = false, bool LCPW = false>
__device__ __forceinline__ void fetch_blocks_direct(const u64* __restrict__ flb, u32 idx, bool need, ulonglong2* wave_stage, u32 lane,
                                                    const u64* __restrict__ flp, const u8* __restrict__ lcp, u64* wave_addr)
{
  fetch_blocks_issue<PAIR, LCPW>(flb, idx, need, wave_stage, lane, flp, lcp, wave_addr);
  fetch_blocks_wait();
}

__device__ __forceinline__ void read_block(const ulonglong2* wave_stage, u32 lane, ulonglong2 (&blk)[8])
{
#pragma unroll
  for(u32 k = 0; k < 8; k++) { blk[k] = wave_stage[lane * 8 + (k ^ (lane & 7))]; }
}

// ---- one evaluator for both block kinds, straight from the staged block in LDS ------------------------------------
// Both layouts keep the rank vector in words 2.. and running popcounts of it and of the edges slice (layout.hpp), so an
// endpoint is: header, the word that holds the position, one masked popcount, the word of the slice that holds the
// resulting edge, one masked popcount -- the SAME instructions for a single-character step (FLB128) and a two-character
// step (FLP128; its Q and D rows are read on top), with the differences in per-lane values.  A wave whose lanes mix the two
// kinds therefore runs one evaluation per round, not two (profiles/r02_config5.md: the matching statistics are bound by
// instruction issue once the lanes of a wave diverge).
__device__ __forceinline__ u64 staged_word(const ulonglong2* wave_stage, u32 lane, u32 w)
{
  return reinterpret_cast<const u64*>(wave_stage)[(lane * 8 + ((w >> 1) ^ (lane & 7))) * 2 + (w & 1)];
}

__device__ __forceinline__ PairEnd eval_staged(const ulonglong2* wave_stage, u32 lane, bool pair, u32 r, bool ep_side)
{
  const ulonglong2 head = wave_stage[lane * 8 + (lane & 7)];             // w0, w1
  const ulonglong2 tail = wave_stage[lane * 8 + (7 ^ (lane & 7))];       // w14, w15
  const u32 wq = r >> 6, width = (pair ? 8u : 10u), mask = (pair ? 0xFFu : 0x3FFu);
  const u64 part = (u64(1) << (r & 63)) - 1;
  const u64 cum_rank = (pair ? tail.y : tail.x), cum_edges = (pair ? tail.y >> 40 : tail.y);
  const u64 bword = staged_word(wave_stage, lane, 2 + wq);
  const u32 ones = (wq == 0 ? 0u : u32(cum_rank >> (width * (wq - 1))) & mask) + u32(__popcll(bword & part));
  PairEnd out;
  out.raw = head.x + ones; out.bits = 0;
  u32 back = (ep_side ? 1u : 0u);
  if(pair)
  {
    const u64 dword = staged_word(wave_stage, lane, 5 + wq), qword = staged_word(wave_stage, lane, 8 + wq);
    const u32 qb = (wq == 0 ? 0u : u32(tail.y >> (16 + 8 * (wq - 1))) & 0xFF) + u32(__popcll(qword & part));
    const u32 qt = u32(tail.y >> 32) & 0xFF;
    const u32 dbit = (ep_side ? u32((dword >> (r & 63)) & 1) : 0u);
    out.bits = qb | ((qt - qb) << 8) | (dbit << 16);
    back = (ep_side ? 1u - dbit : 0u);
  }
  const u64 ncnt = head.y & ~PREV_BIT;
  if(back > ones) { out.node = ncnt - (head.y >> 63); return out; }      // rank(edges, ecnt - 1)
  const u32 k = ones - back, kq = k >> 6;          // kq may name the word after the slice, only with k & 63 == 0: an empty mask
  const u64 kpart = (u64(1) << (k & 63)) - 1;
  const u64 eword = staged_word(wave_stage, lane, (pair ? 11u : 8u) + kq);
  out.node = ncnt + (kq == 0 ? 0u : u32(cum_edges >> (width * (kq - 1))) & mask) + u32(__popcll(eword & kpart));
  return out;
}

// Jump table entry (16 bytes per path node): the forced chain of up to 8 LF steps out of a node -- every
// node on it has a single incoming label, a fast character -- with the nodes reached after all `len`
// steps, after 4 steps (len >= 4) and after 2 steps (len >= 2), so that a pattern with fewer characters
// left than the chain is long can still take a prefix of it.  Nodes are 36-bit (n < 2^36).
//   x: [0,36) node after len steps   [36,64) low 28 bits of the node after 4 steps
//   y: [0,8) its high 8 bits   [8,44) node after 2 steps   [44,60) labels (comp - 1), 2 bits per step   [60,64) len
constexpr u32 JUMP_MAX = 8;
constexpr u64 JUMP_NODE_BITS = 36, JUMP_NODE_MASK = (u64(1) << JUMP_NODE_BITS) - 1;
__device__ __forceinline__ u64 jt_end(ulonglong2 e) { return e.x & JUMP_NODE_MASK; }
__device__ __forceinline__ u64 jt_after4(ulonglong2 e) { return (e.x >> 36) | ((e.y & 0xFF) << 28); }
__device__ __forceinline__ u64 jt_after2(ulonglong2 e) { return (e.y >> 8) & JUMP_NODE_MASK; }
__device__ __forceinline__ u32 jt_labels(ulonglong2 e) { return u32(e.y >> 44) & 0xFFFF; }
__device__ __forceinline__ u32 jt_len(ulonglong2 e) { return u32(e.y >> 60); }
__device__ __forceinline__ ulonglong2 jt_make(u64 end, u64 after4, u64 after2, u32 labels, u32 len)
{
  return make_ulonglong2(end | (after4 << 36), (after4 >> 28) | (after2 << 8) | (u64(labels) << 44) | (u64(len) << 60));
}

// PAIR = true: two characters per step through the FLP128 pair blocks whenever the next two pattern
// characters are fast characters; a pair that does not prove both steps non-empty is replayed as two
// single steps (`force_single`), so every returned range is the one the single-step search returns.
// Waves per SIMD the plain instantiations (what gcsa2_find_device runs) are compiled for.  4 = 16 waves per CU (107 VGPRs).
// -DGCSA2_FIND_WAVES=5 gives 96 VGPRs with only the pattern and output pointers spilled, outside the step loop, and the 18
// waves per CU that LDS allows -- measured 2-5 % SLOWER (profiles/r02_occupancy.md): at 16 waves the kernel already sits at
// the memory system's request rate, more resident chains only lengthen each one's round trip.
#ifndef GCSA2_FIND_WAVES
#define GCSA2_FIND_WAVES 4
#endif
// PACKED = true (gcsa2_find_packed_device): the patterns arrive as 2-bit codes, all of one length `offsets` (the argument is
// then the LENGTH, not an array), last character first -- word j of pattern q, at patterns + 8 (q W + j) with W = ceil(length /
// 32), holds the characters at distance 32 j .. 32 j + 31 from the pattern's end, comp - 1 of the character at distance t in
// bits [2 (t & 31), 2 (t & 31) + 2) (the layout k_pack_patterns makes for the matching statistics).  A caller that can pack
// sends 8 bytes per 32-mer over the link instead of 32 + 8; the kernel's seed index and pattern window are the words
// themselves.  Only fast characters can be written this way (comps 1..4: a pattern with an N takes the byte interface).
template<bool STATS, bool JUMP = false, bool PAIR = false, bool PACKED = false>
__global__ __launch_bounds__(TPB2, (STATS || JUMP) ? 4 : GCSA2_FIND_WAVES) void k_find2(DevImage img, const u8* __restrict__ patterns,
                                               const u64* __restrict__ offsets, u64 nq,
                                               u64* __restrict__ out, unsigned long long* __restrict__ stats,
                                               const u32* __restrict__ perm)
{
  constexpr bool WINDOW = true;               // the packed pattern window (below); the byte-at-a-time path is kept for reference only
  __shared__ ulonglong2 stage[TPB2 * 8];
  __shared__ Tables2 t;
  if(threadIdx.x < 2 * MAX_SIGMA) { t.crange[threadIdx.x] = img.crange[threadIdx.x]; }
  t.c2c[threadIdx.x] = img.char2comp[threadIdx.x];
  t.c2c[threadIdx.x + TPB2] = img.char2comp[threadIdx.x + TPB2];
  __syncthreads();

  const u32 lane = threadIdx.x & 63;
  ulonglong2* wave_stage = stage + (threadIdx.x & ~63u) * 8;
  u64 blocks = 0, steps = 0, lookups = 0, jumps = 0;
  [[maybe_unused]] u64 fetch_steps = 0, second_fetches = 0, wide_seeds = 0;     // STATS only
  [[maybe_unused]] unsigned int* touched = nullptr;
  if constexpr(STATS) { touched = reinterpret_cast<unsigned int*>(stats[7]); }

  u64 q = ~u64(0), sp = 0, ep = img.n - 1, i = 0;
  const u8* p = patterns;
  bool done = true;
  [[maybe_unused]] bool tried = false;     // JUMP: the entry of the current node was examined and does not apply
  [[maybe_unused]] bool no_jump = false;   // JUMP: no entry can apply for the rest of this pattern
  [[maybe_unused]] u64 win_code = 0;       // packed pattern window (see below)
  [[maybe_unused]] u32 win_used = ~u32(0), win_bad = 0;      // characters consumed since the window was loaded (~0: no window)
  [[maybe_unused]] u32 force_single = 0;   // PAIR: characters that must be consumed by single steps (replay)
  u64 word = 0, word_addr = ~u64(0);        // pattern bytes are consumed back to front from aligned 8-byte words
  auto byte_at = [&](u64 pos) -> u32
  {
    u64 addr = reinterpret_cast<u64>(p) + pos, aligned = addr & ~u64(7);
    if(aligned != word_addr) { word = *reinterpret_cast<const u64*>(aligned); word_addr = aligned; }
    return u32(word >> ((addr & 7) * 8)) & 0xFF;
  };
  auto start = [&](u64 query)               // begin the backward search of `query` (< nq)
  {
    q = query; sp = 0; ep = img.n - 1; i = 0; done = true; word_addr = ~u64(0); tried = false; no_jump = false; win_used = ~u32(0);
    force_single = 0;
    u64 begin = 0, len = 0;
    if constexpr(PACKED)
    {
      len = reinterpret_cast<u64>(offsets);                    // one length for the whole batch
      begin = q * ((len + 31) >> 5) * 8;                       // byte offset of the pattern's first code word
    }
    else { begin = offsets[q]; len = offsets[q + 1] - begin; }
    if(len > 0 && img.n > 0)                                   // gcsa.h:99
    {
      p = patterns + begin;
      const u32 k = img.kmer_k;
      bool seeded = false;
      if(k > 0 && len >= k)
      {
        u64 tix = 0;
        bool fast = true;
        if constexpr(PACKED) { tix = *reinterpret_cast<const u64*>(p) & ((u64(1) << (2 * k)) - 1); }
        else
        {
          for(u32 j = 0; j < k; j++)                           // j-th character from the end
          {
            u32 comp = t.c2c[byte_at(len - 1 - j)];
            fast = fast && (comp - 1 < 4);
            tix |= u64((comp - 1) & 3) << (2 * j);
          }
        }
        if(fast)
        {
          const u64 entry = img.kmer_table[tix];
          sp = entry & SEED_SP_MASK; ep = sp + (entry >> SEED_SP_BITS) - 1;
          fast = (entry >> SEED_SP_BITS) != SEED_WIDE;         // a wide range is not in the table
          if(STATS) { lookups++; wide_seeds += (fast ? 0 : 1); }
        }
        if(fast)
        {
          i = len - k; seeded = true;
        }
      }
      if(!seeded)
      {
        i = len - 1;
        u32 comp = 0;
        if constexpr(PACKED) { comp = 1 + u32(*reinterpret_cast<const u64*>(p) & 3); }
        else { comp = t.c2c[byte_at(i)]; }
        sp = t.crange[2 * comp]; ep = t.crange[2 * comp + 1];  // charRange, gcsa.h:101-102, 150-153
      }
      done = range_empty(sp, ep) || i == 0;                    // gcsa.h:103
    }
  };

  {
    // perm != nullptr: lane g works on query perm[g] (queries ordered by length, so that the 64
    // chains of a wave finish together); results still go to out[query].
    const u64 gid = u64(blockIdx.x) * TPB2 + threadIdx.x;
    if(gid < nq) { start(perm != nullptr ? u64(perm[gid]) : gid); }
  }

  while(true)
  {
    if(!__any(!done)) { break; }
    // JUMP: a range of one path node whose next <= 8 predecessors are forced (a single incoming label
    // each) and spell the next pattern characters moves there with ONE 16-byte lookup.  Same result as
    // stepping: LF of a single node with a matching label is the single node behind that edge.
    bool jumping = false;
    ulonglong2 entry = make_ulonglong2(0, 0);
    if constexpr(JUMP)
    {
      jumping = !done && sp == ep && i >= 2 && !tried && !no_jump;   // the last character is a plain step: same cost, no wasted lookup
      entry = img.jump_tab[jumping ? sp : 0];                  // branch-free: all lanes' loads in flight together
    }
    if constexpr(WINDOW)
    {
      // The next pattern characters as 2-bit codes: window of the 32 positions below win_top = i + win_used,
      // position win_top - 1 - r at bits [2r, 2r + 2) of win_code, bit r of win_bad = "not a fast character".
      // Refilled once per 24 consumed characters (five independent word loads), so that neither the
      // jump test nor a step waits for pattern bytes.
      if constexpr(PACKED)
      {
        if(!done && win_used > 24)                             // two code words and a funnel shift
        {
          const u64 total = reinterpret_cast<u64>(offsets), words = (total + 31) >> 5;
          const u64 t0 = total - i, w = t0 >> 5;
          const u32 s = u32(t0 & 31);
          const u64* code = reinterpret_cast<const u64*>(p);
          const u64 c0 = code[w], c1 = (w + 1 < words ? code[w + 1] : 0);
          win_code = (s == 0 ? c0 : (c0 >> (2 * s)) | (c1 << (64 - 2 * s)));
          win_bad = 0; win_used = 0;
        }
      }
      else if(!done && win_used > 24)
      {
        win_used = 0; win_code = 0; win_bad = 0;
        const u64 count = (i < 32 ? i : 32), low = reinterpret_cast<u64>(p) + i - count, base = low & ~u64(7);
        u64 w[5];
        const u64 last = (low + count - 1) & ~u64(7);           // never read past the word of the last byte needed
#pragma unroll
        for(u32 k = 0; k < 5; k++) { const u64 a = base + 8 * k; w[k] = *reinterpret_cast<const u64*>(a < last ? a : last); }
        for(u32 r = 0; r < count; r++)
        {
          const u64 at = (low - base) + (count - 1 - r);       // byte offset of position win_top - 1 - r
          u64 word = w[0];                                     // (a clamped slot repeats the last word and is never selected)
#pragma unroll
          for(u32 k = 1; k < 5; k++) { if((at >> 3) == k) { word = w[k]; } }
          const u32 c = u32(t.c2c[u32(word >> ((at & 7) * 8)) & 0xFF]) - 1;
          win_code |= u64(c & 3) << (2 * r);
          win_bad |= u32(c < 4 ? 0 : 1) << r;
        }
      }
    }
    const bool stepping = !done && !jumping;
    u32 comp = 0, r_sp = 0, r_ep = 0, idx_sp = 0, idx_ep = 0;
    bool pair = false;
    if(stepping)
    {
      if constexpr(PAIR)
      {
        if(force_single == 0 && i >= 2)
        {
          const u32 r = win_used;                              // window slot of position i - 1; i - 2 is slot r + 1
          pair = ((win_bad >> r) & 3) == 0;                    // both are fast characters
          if(pair)
          {
            const u32 c2 = u32(win_code >> (2 * r)) & 3, c1 = u32(win_code >> (2 * r + 2)) & 3;
            u32 b_sp, b_ep;
            pair_block_of(sp, b_sp, r_sp); pair_block_of(ep + 1, b_ep, r_ep);
            const u32 first = (c1 * 4 + c2) * u32(img.flp_nblocks);
            idx_sp = (first + b_sp) | PAIR_FLAG; idx_ep = (first + b_ep) | PAIR_FLAG;
          }
        }
      }
      if(!pair)
      {
        i--;
        if constexpr(PAIR) { force_single -= (force_single > 0 ? 1 : 0); }
        if constexpr(WINDOW)
        {
          const u32 r = win_used++;
          if((win_bad >> r) & 1)                               // rare: read the byte itself (no cached word kept across steps)
          {
            const u64 addr = reinterpret_cast<u64>(p) + i;
            comp = t.c2c[u32(*reinterpret_cast<const u64*>(addr & ~u64(7)) >> ((addr & 7) * 8)) & 0xFF];
          }
          else { comp = 1 + (u32(win_code >> (2 * r)) & 3); }
        }
        else { comp = t.c2c[byte_at(i)]; }
        u32 b_sp, b_ep;
        flb_block_of(sp, b_sp, r_sp); flb_block_of(ep + 1, b_ep, r_ep);
        idx_sp = comp * u32(img.flb_nblocks) + b_sp; idx_ep = comp * u32(img.flb_nblocks) + b_ep;
      }
    }
    PairEnd p_sp = {0, 0, 0}, p_ep = {0, 0, 0};   // a single step keeps (edge, node) in .raw / .node: one set of registers
    const bool need2 = stepping && idx_ep != idx_sp;
    if(STATS && stepping)
    {
      blocks += 1 + (need2 ? 1 : 0); fetch_steps++; second_fetches += (need2 ? 1 : 0);
      // stats[7], when the caller put a device address there: a bitmap over the blocks of the image (the FLB128 blocks of all
      // comps, then the FLP128 blocks of all pairs) -- the DISTINCT blocks a batch touches are its working set (bench.py)
      if(touched != nullptr)
      {
        const u32 singles = u32(img.sigma) * u32(img.flb_nblocks);
        const u32 a = (idx_sp & PAIR_FLAG) ? singles + (idx_sp & ~PAIR_FLAG) : idx_sp;
        atomicOr(touched + (a >> 5), 1u << (a & 31));
        if(need2) { const u32 b = (idx_ep & PAIR_FLAG) ? singles + (idx_ep & ~PAIR_FLAG) : idx_ep; atomicOr(touched + (b >> 5), 1u << (b & 31)); }
      }
    }
    ulonglong2 blk[8];
    fetch_blocks<PAIR>(img.flb, idx_sp, stepping, wave_stage, lane, img.flp);
    if(stepping)
    {
      read_block(wave_stage, lane, blk);
      if(PAIR && pair)
      {
        p_sp = eval_pair(blk, r_sp, false);
        if(idx_ep == idx_sp) { p_ep = eval_pair(blk, r_ep, true); }
      }
      else
      {
        eval_endpoint(blk, r_sp, 0, p_sp.raw, p_sp.node);      // gcsa.h:271, then rank(edges, sp')
        if(idx_ep == idx_sp) { eval_endpoint(blk, r_ep, 1, p_ep.raw, p_ep.node); }
      }
    }
    if(__any(need2))
    {
      __builtin_amdgcn_wave_barrier();
      fetch_blocks<PAIR>(img.flb, idx_ep, need2, wave_stage, lane, img.flp);
      if(need2)
      {
        read_block(wave_stage, lane, blk);
        if(PAIR && pair) { p_ep = eval_pair(blk, r_ep, true); }
        else { eval_endpoint(blk, r_ep, 1, p_ep.raw, p_ep.node); }   // gcsa.h:272: LF(ep + 1) - 1
      }
    }
    __builtin_amdgcn_wave_barrier();
    if(stepping)
    {
      if(PAIR && pair)
      {
        u64 a = 0, b = 0;
        const u32 outcome = pair_outcome(p_sp, p_ep, idx_ep == idx_sp, a, b);
        if(outcome == 2)                                       // neither step empties
        {
          sp = p_sp.node; ep = p_ep.node; i -= 2; win_used += 2; done = (i == 0);
          if(STATS) { steps += 2; }
          if constexpr(JUMP) { tried = false; }
        }
        else if(outcome == 1)                                  // the second step empties: its edge-space integers, gcsa.h:160
        {
          sp = a; ep = b; i -= 2; win_used += 2; done = true;
          if(STATS) { steps += 2; }
        }
        else { force_single = 2; }                             // replayed as two single steps from the unchanged (sp, ep)
      }
      else
      {
        if(STATS) { steps++; }
        u64 a = p_sp.raw, b = p_ep.raw - 1;                    // edge space
        if(range_empty(a, b)) { sp = a; ep = b; done = true; } // gcsa.h:160
        else { sp = p_sp.node; ep = p_ep.node; done = (i == 0); }   // gcsa.h:161, 103
        if constexpr(JUMP) { tried = false; }
      }
    }
    if constexpr(JUMP)
    {
      if(jumping)
      {
        const u32 len = jt_len(entry), r = win_used;
        u32 bad8 = (win_bad >> r) & 0xFF;                      // one bit per slot -> every second bit
        bad8 = (bad8 | (bad8 << 4)) & 0x0F0F; bad8 = (bad8 | (bad8 << 2)) & 0x3333; bad8 = (bad8 | (bad8 << 1)) & 0x5555;
        // number of leading steps of the chain that the pattern follows
        const u32 diff = (u32(win_code >> (2 * r)) ^ jt_labels(entry)) & 0xFFFF;
        const u32 bad = ((diff | (diff >> 1)) & 0x5555) | bad8;
        u32 usable = (bad != 0 ? u32(__ffs(int(bad)) - 1) >> 1 : 8u);
        usable = (usable < len ? usable : len);
        usable = (usable < i ? usable : u32(i));
        u32 take = (usable == len ? len : (usable >= 4 ? 4u : (usable >= 2 ? 2u : 0u)));
        if(STATS) { jumps++; }
        if(take > 0)
        {
          sp = ep = (take == len ? jt_end(entry) : (take == 4 ? jt_after4(entry) : jt_after2(entry)));
          i -= take; win_used += take; done = (i == 0);
          if(STATS) { steps += take; }
        }
        else
        {
          tried = true;                                        // step normally from this node
          // A chain the pattern leaves (or outlasts by one character): the nodes the steps will visit lie
          // on that same chain, so no later entry can apply either.
          no_jump = (len > 0);
        }
      }
    }
  }
  {
    const u64 gid = u64(blockIdx.x) * TPB2 + threadIdx.x;     // recomputed: the query id is not held across the loop
    if(gid < nq) { reinterpret_cast<ulonglong2*>(out)[perm != nullptr ? u64(perm[gid]) : gid] = make_ulonglong2(sp, ep); }
  }
  if(STATS)
  {
    for(int o = 32; o > 0; o >>= 1)
    {
      blocks += __shfl_down(blocks, o, 64); steps += __shfl_down(steps, o, 64); lookups += __shfl_down(lookups, o, 64);
      jumps += __shfl_down(jumps, o, 64); fetch_steps += __shfl_down(fetch_steps, o, 64);
      second_fetches += __shfl_down(second_fetches, o, 64); wide_seeds += __shfl_down(wide_seeds, o, 64);
    }
    if(lane == 0)
    {
      atomicAdd(stats, (unsigned long long)blocks); atomicAdd(stats + 1, (unsigned long long)steps);
      atomicAdd(stats + 2, (unsigned long long)lookups); atomicAdd(stats + 3, (unsigned long long)jumps);
      atomicAdd(stats + 4, (unsigned long long)fetch_steps); atomicAdd(stats + 5, (unsigned long long)second_fetches);
      atomicAdd(stats + 6, (unsigned long long)wide_seeds);
    }
  }
}

// ---- FLP128 pair blocks (layout.hpp), built on the device from the RB64 vectors -----------------------
// grid (blocks, 4), launched in slices of blocks (a HIP grid holds < 2^32 threads): workgroup (b - first, c2 - 1)
// of 192 threads writes block b of the four pairs (c1, c2), c1 = 1..4; thread r owns position i = 192 b + r.
__global__ __launch_bounds__(192) void k_build_pair_blocks(DevImage img, u64 first, u64* __restrict__ out)
{
  __shared__ u64 s_ecnt[4], s_p[4][3], s_q[3], s_e[4][4];
  const u64 b = first + blockIdx.x, nb = img.flp_nblocks;
  const u32 c2 = 1 + blockIdx.y, r = threadIdx.x;
  const u64 i = b * PAIR_BITS + r, n = img.n, e = img.e;
  const bool valid = i <= n;
  u64 rk = 0;
  const bool q = bv_get_rank(bwt_of(img, c2), valid ? (i < n ? i : n) : 0, rk) && i < n;
  const u64 x = img.C[c2] + rk;                                // E_c2(i)
  u64 u = 0;
  const bool ebit = bv_get_rank(img.edges, clampu(x, e), u) && x < e;      // edges[x], u = N(x)
  const bool split = (x >= 1 && x - 1 < e) && !bv_get(img.edges, x - 1);   // position i splits the out-edges of node u
  const u64 qw = __ballot(valid && q);
#pragma unroll
  for(u32 c1 = 1; c1 <= 4; c1++)
  {
    u64 rc = 0;
    const bool bc = bv_get_rank(bwt_of(img, c1), clampu(u, n), rc) && u < n;
    const u64 pw = __ballot(valid && q && ebit && bc), dw = __ballot(valid && split && bc);
    u64* dst = out + (u64((c1 - 1) * 4 + (c2 - 1)) * nb + b) * FLB_WORDS;
    if((r & 63) == 0) { dst[2 + (r >> 6)] = pw; dst[5 + (r >> 6)] = dw; dst[8 + (r >> 6)] = qw; s_p[c1 - 1][r >> 6] = pw; s_q[r >> 6] = qw; }
    if(r == 0)
    {
      const u64 ecnt = img.C[c1] + rc;                         // H(192 b)
      u64 ncnt = 0;
      (void)bv_get_rank(img.edges, clampu(ecnt, e), ncnt);
      const u64 prev = (ecnt >= 1 && ecnt - 1 < e && bv_get(img.edges, ecnt - 1)) ? PREV_BIT : 0;
      dst[0] = ecnt; dst[1] = ncnt | prev;
      s_ecnt[c1 - 1] = ecnt;
    }
  }
  __syncthreads();
  if(r < 16)
  {
    const u32 c1 = 1 + r / 4, k = r % 4;
    u64* dst = out + (u64((c1 - 1) * 4 + (c2 - 1)) * nb + b) * FLB_WORDS;
    const u64 slice = bv_bits64(img.edges, s_ecnt[c1 - 1] + 64 * k);
    dst[11 + k] = slice; s_e[c1 - 1][k] = slice;
  }
  __syncthreads();
  if(r < 4)                                                    // word 15: running popcounts, one byte each (layout.hpp)
  {
    u64* dst = out + (u64(r * 4 + (c2 - 1)) * nb + b) * FLB_WORDS;
    const u64 p0 = __popcll(s_p[r][0]), p1 = p0 + __popcll(s_p[r][1]);
    const u64 q0 = __popcll(s_q[0]), q1 = q0 + __popcll(s_q[1]), q2 = q1 + __popcll(s_q[2]);
    const u64 e0 = __popcll(s_e[r][0]), e1 = e0 + __popcll(s_e[r][1]), e2 = e1 + __popcll(s_e[r][2]);
    dst[15] = p0 | (p1 << 8) | (q0 << 16) | (q1 << 24) | (q2 << 32) | (e0 << 40) | (e1 << 48) | (e2 << 56);
  }
}

// keys for the length-bucketed launch: len[q] = offsets[q + 1] - offsets[q] (saturated), idx[q] = q
__global__ __launch_bounds__(TPB) void k_pattern_lengths(const u64* __restrict__ offsets, u64 nq,
                                                         u32* __restrict__ len, u32* __restrict__ idx)
{
  u64 q = u64(blockIdx.x) * TPB + threadIdx.x;
  if(q >= nq) { return; }
  u64 l = offsets[q + 1] - offsets[q];
  len[q] = u32(l > 0xFFFFFFFFull ? 0xFFFFFFFFull : l);
  idx[q] = u32(q);
}

// LF(range, comp) (gcsa.h:155-162), one step for the lanes of a wavefront (`live`: this lane has a range): the fused-block
// machinery of k_find2.  Every lane of the wavefront must call it.  This is the primitive vg's MEM loop calls once per character.
__device__ __forceinline__ void lf_step_wave(const DevImage& img, u64 in_sp, u64 in_ep, u32 comp, bool live, ulonglong2* wave_stage, u32 lane,
                                             u64& out_sp, u64& out_ep)
{
  u64 sp = 0, ep = 0;
  u32 idx_sp = 0, idx_ep = 0, r_sp = 0, r_ep = 0;
  if(live)
  {
    if(comp >= img.sigma) { comp = u32(img.sigma - 1); }     // memory safety only
    sp = clampu(in_sp, img.n);
    u64 e1 = clampu(in_ep + 1, img.n);
    u32 b_sp, b_ep;
    flb_block_of(sp, b_sp, r_sp); flb_block_of(e1, b_ep, r_ep);
    idx_sp = comp * u32(img.flb_nblocks) + b_sp; idx_ep = comp * u32(img.flb_nblocks) + b_ep;
  }
  ulonglong2 blk[8];
  u64 e_sp = 0, n_sp = 0, e_ep = 0, n_ep = 0;
  fetch_blocks(img.flb, idx_sp, live, wave_stage, lane);
  if(live)
  {
    read_block(wave_stage, lane, blk);
    eval_endpoint(blk, r_sp, 0, e_sp, n_sp);
    if(idx_ep == idx_sp) { eval_endpoint(blk, r_ep, 1, e_ep, n_ep); }
  }
  bool need2 = live && idx_ep != idx_sp;
  if(__any(need2))
  {
    __builtin_amdgcn_wave_barrier();
    fetch_blocks(img.flb, idx_ep, need2, wave_stage, lane);
    if(need2)
    {
      read_block(wave_stage, lane, blk);
      eval_endpoint(blk, r_ep, 1, e_ep, n_ep);
    }
  }
  if(live)
  {
    u64 a = e_sp, b = e_ep - 1;
    if(range_empty(a, b)) { sp = a; ep = b; } else { sp = n_sp; ep = n_ep; }     // gcsa.h:160-161
  }
  out_sp = sp; out_ep = ep;
}

__global__ __launch_bounds__(TPB2) void k_lf2(DevImage img, const u64* __restrict__ in, const u8* __restrict__ comps,
                                             u64 nq, u64* __restrict__ out)
{
  __shared__ ulonglong2 stage[TPB2 * 8];
  const u32 lane = threadIdx.x & 63;
  ulonglong2* wave_stage = stage + (threadIdx.x & ~63u) * 8;
  const u64 q = u64(blockIdx.x) * TPB2 + threadIdx.x;
  const bool live = q < nq;
  ulonglong2 r = make_ulonglong2(0, 0);
  u32 comp = 0;
  if(live) { r = reinterpret_cast<const ulonglong2*>(in)[q]; comp = comps[q]; }
  u64 sp = 0, ep = 0;
  lf_step_wave(img, r.x, r.y, comp, live, wave_stage, lane, sp, ep);
  if(live) { reinterpret_cast<ulonglong2*>(out)[q] = make_ulonglong2(sp, ep); }
}

// LF(path_node): first incoming edge, comps 1..fast_chars, then fast_chars+1..sigma-1, else 0
__device__ __forceinline__ u64 lf_node(const DevImage& img, const u64* C, u64 node)
{
  u32 sigma = u32(img.sigma);
  u32 comp = 0; u64 rank = 0; bool hit = false;
  for(u32 c = 1; c < sigma && !hit; c++)
  {
    u64 r;
    if(bv_get_rank(bwt_of(img, c), node, r)) { comp = c; rank = r; hit = true; }
  }
  if(!hit) { rank = bv_rank(bwt_of(img, 0), node); }
  return bv_rank(img.edges, clampu(C[comp] + rank, img.e));
}

__global__ __launch_bounds__(TPB) void k_lf_node(DevImage img, const u64* __restrict__ in, u64 nq,
                                                 u64* __restrict__ out)
{
  __shared__ Tables t;
  stage_tables(img, t);
  u64 q = u64(blockIdx.x) * TPB + threadIdx.x;
  if(q >= nq) { return; }
  u64 node = in[q];
  out[q] = (node < img.n ? lf_node(img, t.C, node) : 0);
}

// ---- jump table: memoised unary LF chains ----------------------------------------------------------
// Level 1: a node with exactly one incoming label c, c a fast character, gets (LF(node), 1 step, c - 1).
__global__ __launch_bounds__(TPB) void k_jump_init(DevImage img, u64 first, ulonglong2* __restrict__ table)
{
  __shared__ Tables t;
  stage_tables(img, t);
  u64 v = first + u64(blockIdx.x) * TPB + threadIdx.x;
  if(v >= img.n) { return; }
  u32 labels = 0, comp = 0;
  for(u32 c = 0; c < u32(img.sigma); c++) { if(bv_get(bwt_of(img, c), v)) { labels++; comp = c; } }
  ulonglong2 e = make_ulonglong2(0, 0);
  if(labels == 1 && comp - 1 < 4) { e = jt_make(lf_node(img, t.C, v), 0, 0, comp - 1, 1); }
  table[v] = e;
}

// Doubling: an entry that is full at `have` steps is extended by the entry of the node it reaches; the
// node it reaches is recorded as the 2-step / 4-step landing point when have is 2 / 4.
__global__ __launch_bounds__(TPB) void k_jump_double(const ulonglong2* __restrict__ in, u64 n, u64 first, u32 have,
                                                     ulonglong2* __restrict__ out)
{
  u64 v = first + u64(blockIdx.x) * TPB + threadIdx.x;
  if(v >= n) { return; }
  ulonglong2 e = in[v];
  if(jt_len(e) == have)
  {
    const u64 here = jt_end(e);
    u64 after2 = (have == 2 ? here : jt_after2(e)), after4 = (have == 4 ? here : jt_after4(e));
    u64 end = here; u32 labels = jt_labels(e), len = have;
    ulonglong2 next = in[here];
    if(jt_len(next) > 0)
    {
      labels |= jt_labels(next) << (2 * have);
      end = jt_end(next); len = have + jt_len(next);
      if(have == 1) { after2 = end; }                        // 1 + 1 steps
    }
    e = jt_make(end, after4, after2, labels & 0xFFFF, len);
  }
  out[v] = e;
}

// LF_fast (all = 0, comps 1..fast_chars) / LF_all (all = 1, comps 1..sigma-2); src/gcsa.cpp:742-798
__global__ __launch_bounds__(TPB) void k_lf_all(DevImage img, const u64* __restrict__ in, u64 nq, int all,
                                                u64* __restrict__ out)
{
  __shared__ Tables t;
  stage_tables(img, t);
  u64 q = u64(blockIdx.x) * TPB + threadIdx.x;
  if(q >= nq) { return; }
  ulonglong2 r = reinterpret_cast<const ulonglong2*>(in)[q];
  u32 sigma = u32(img.sigma);
  ulonglong2* dst = reinterpret_cast<ulonglong2*>(out) + q * sigma;
  for(u32 c = 0; c < sigma; c++) { dst[c] = make_ulonglong2(1, 0); }
  if(range_empty(r.x, r.y)) { return; }
  u32 limit = (all ? sigma - 2 : u32(img.fast_chars));
  u64 sp0 = clampu(r.x, img.n), ep0 = clampu(r.y, img.n);
  for(u32 c = 1; c <= limit; c++)
  {
    DevBV bv = bwt_of(img, c);
    if(r.x == r.y)     // single path node: bit probe (gcsa.cpp:748-757)
    {
      u64 rk;
      if(sp0 < img.n && bv_get_rank(bv, sp0, rk))
      {
        u64 v = bv_rank(img.edges, clampu(t.C[c] + rk, img.e));
        dst[c] = make_ulonglong2(v, v);
      }
    }
    else               // general case (gcsa.cpp:758-765)
    {
      u64 ra, rb;
      bv_rank2(bv, sp0, clampu(ep0 + 1, img.n), ra, rb);
      u64 sp = t.C[c] + ra, ep = t.C[c] + rb - 1;
      if(!range_empty(sp, ep)) { path_node_range(img, sp, ep); }
      dst[c] = make_ulonglong2(sp, ep);
    }
  }
}


}  // namespace
