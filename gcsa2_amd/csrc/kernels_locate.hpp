// kernels_locate.hpp -- count(), the locate() pipeline (pred4 nibbles, walks, removeDuplicates) and the countKMers frontier expansion.
// Part of the single translation unit gcsa2_hip.hip (device code, anonymous namespace).
#pragma once

#include "kernels_find.hpp"

using namespace g2;

namespace {

// ---- counting ----------------------------------------------------------------------------

// SadaSparse::count (support.h:329-335)
__device__ __forceinline__ u64 sada_sparse_count(const DevImage& img, u64 sp, u64 ep)
{
  u64 a, b;
  bv_rank2(img.xfilter, sp, ep + 1, a, b);
  if(b <= a) { return 0; }
  return (bv_select(img.xvalues, b) + 1) - (a > 0 ? bv_select(img.xvalues, a) + 1 : 0);
}

// SadaCount::count (support.h:255-258)
__device__ __forceinline__ u64 sada_count(const DevImage& img, u64 sp, u64 ep)
{
  return (bv_select(img.redundant, ep + 1) - ep) - (sp > 0 ? bv_select(img.redundant, sp) + 1 - sp : 0);
}

// Number of values of a sampled node whose first sample has index s: the samples up to and including the next one marked
// last (gcsa.h:208), found 64 marks at a time instead of by one dependent load per sample (tandem repeats put hundreds of
// samples on a node, and the lane that met one held its wavefront).
__device__ __forceinline__ u32 sample_run(const DevImage& img, u64 s)
{
  u64 pos = s;
  while(pos < img.samples.size)
  {
    const u64 w = bv_bits64(img.samples, pos);
    if(w != 0) { return u32(pos + u64(__ffsll((long long)w) - 1) - s + 1); }
    pos += 64;
  }
  return u32(img.samples.size > s ? img.samples.size - s : 1);          // no last mark: a broken index; stay inside the array
}

__device__ __forceinline__ u64 count_range(const DevImage& img, u64 sp, u64 ep)
{
  if(range_empty(sp, ep) || ep >= img.n) { return 0; }                  // gcsa.cpp:805
  u64 res = sada_sparse_count(img, sp, ep) + (ep + 1 - sp);            // gcsa.cpp:806
  if(ep > sp) { res -= sada_count(img, sp, ep - 1); }                  // gcsa.cpp:807
  return res;
}

__global__ __launch_bounds__(TPB) void k_count(DevImage img, const u64* __restrict__ ranges, u64 nq,
                                               u64* __restrict__ out)
{
  u64 q = u64(blockIdx.x) * TPB + threadIdx.x;
  if(q >= nq) { return; }
  ulonglong2 r = reinterpret_cast<const ulonglong2*>(ranges)[q];
  out[q] = count_range(img, r.x, r.y);
}

// ---- pred4: first-predecessor code + sampled flag, 4 bits per path node ------------------------
// One thread per 64 nodes: reads the payload word of every B_c and of sampled_paths and writes
// four u64 (16 nibbles each).  Derived data: LF(path_node) probes comps 1..sigma-1 in order and
// falls back to comp 0 (gcsa.h:165-183); the nibble records which probe hits first.
__global__ __launch_bounds__(TPB) void k_build_pred4(DevImage img, u64 nwords, u64* __restrict__ out)
{
  u64 w = u64(blockIdx.x) * TPB + threadIdx.x;
  if(w >= nwords) { return; }
  u64 blk = w / PAYLOAD_WORDS, j = w - blk * PAYLOAD_WORDS;
  u64 remaining = ~u64(0), p0 = 0, p1 = 0, p2 = 0;
  for(u32 c = 1; c < u32(img.sigma); c++)
  {
    u64 bits = bwt_of(img, c).blocks[blk * BLOCK_WORDS + 1 + j] & remaining;
    if(c & 1) { p0 |= bits; }
    if(c & 2) { p1 |= bits; }
    if(c & 4) { p2 |= bits; }
    remaining &= ~bits;
  }
  u64 smp = (img.has_samples ? img.sampled.blocks[blk * BLOCK_WORDS + 1 + j] : 0);
  for(u32 part = 0; part < 4; part++)
  {
    u64 v = 0;
    for(u32 k = 0; k < 16; k++)
    {
      u32 bit = part * 16 + k;
      u64 nib = ((p0 >> bit) & 1) | (((p1 >> bit) & 1) << 1) | (((p2 >> bit) & 1) << 2) | (((smp >> bit) & 1) << 3);
      v |= nib << (4 * k);
    }
    out[w * 4 + part] = v;
  }
}

__device__ __forceinline__ u32 pred4_get(const u64* pred4, u64 node)
{
  return u32(pred4[node >> 4] >> ((node & 15) * 4)) & 15;
}

// Query owning flattened node g: the last q with node_off[q] <= g, known to lie in [qa, qb]; ga / gb = the first flattened
// node of the bracket and the one behind its last.  Starts from the proportional guess and gallops, so runs of similar-sized
// ranges cost O(1) probes instead of log2 of their number.
__device__ __forceinline__ u64 owner_between(const u64* __restrict__ node_off, u64 qa, u64 qb, u64 ga, u64 gb, u64 g)
{
  if(qa >= qb) { return qa; }
  u64 q = qa + u64(double(g - ga) / double(gb - ga) * double(qb - qa + 1));
  if(q > qb) { q = qb; }
  u64 lo, hi;
  if(node_off[q] <= g)
  {
    lo = q; hi = qb;
    for(u64 step = 1; lo + step <= qb; step <<= 1)
    {
      if(node_off[lo + step] > g) { hi = lo + step - 1; break; }
      lo += step;
    }
  }
  else
  {
    hi = q - 1; lo = qa;                    // node_off[qa] <= g
    for(u64 step = 1; step <= hi - qa; step <<= 1)
    {
      if(node_off[hi - step] <= g) { lo = hi - step; break; }
      hi -= step;
    }
  }
  while(lo < hi)
  {
    u64 mid = (lo + hi + 1) >> 1;
    if(node_off[mid] <= g) { lo = mid; } else { hi = mid - 1; }
  }
  return lo;
}

__device__ __forceinline__ u64 owner_of(const u64* __restrict__ node_off, u64 nq, u64 total_nodes, u64 g)
{
  return owner_between(node_off, 0, nq - 1, 0, total_nodes, g);
}

// The owners of a block of consecutive flattened nodes lie between the owner of its first node and the owner of the next
// block's first node: k_block_owners finds those once, one lane per block (entry `blocks` = the
// owner of the last node), and every lane of the walk kernel searches its bracket only -- a single query when the ranges are
// wide: no probe at all.  (Two full searches by two lanes of every workgroup, the others waiting at a barrier, were most of
// k_locate_tab on the repeat-rich batch.)
__global__ __launch_bounds__(TPB) void k_block_owners(const u64* __restrict__ node_off, u64 nq, u64 total_nodes, u32 threads, u64 blocks,
                                                      u64* __restrict__ owners)
{
  const u64 j = u64(blockIdx.x) * TPB + threadIdx.x;
  if(j > blocks) { return; }
  const u64 g = (j * threads < total_nodes ? j * threads : total_nodes - 1);
  owners[j] = owner_of(node_off, nq, total_nodes, g);
}

// (the brackets are per WAVEFRONT, 64 consecutive nodes: with ranges of a few hundred path nodes most wavefronts lie inside
// one query and skip the search; per workgroup of 256 most did not)
constexpr u32 OWNER_SPAN = 64;

__device__ __forceinline__ u64 owner_in_wave(const u64* __restrict__ owners, const u64* __restrict__ node_off, u64 total_nodes,
                                             u64 g, bool live)
{
  const u64 j = g / OWNER_SPAN, g_first = j * OWNER_SPAN;
  const u64 g_end = (g_first + OWNER_SPAN <= total_nodes ? g_first + OWNER_SPAN : total_nodes);
  return live ? owner_between(node_off, owners[j], owners[j + 1], g_first, g_end, g) : 0;
}

__device__ __forceinline__ void locate_item(const DevImage& img, const u64* __restrict__ ranges, u64 q,
                                            const u64* __restrict__ node_off, const u64* __restrict__ raw_off,
                                            u64 g, u64& node, u64& dest)
{
  u64 sp = ranges[2 * q];
  node = sp + (g - node_off[q]);
  dest = raw_off[q] + (node - sp) + (node > sp ? sada_sparse_count(img, sp, node - 1) : 0);
}

// while(!sampled(node)) { node = LF(node); steps++; } (gcsa.cpp:883-887) for the 64 lanes of a wave:
// the pred4 nibble says which comp's fused block holds the incoming edge (and whether the node is
// sampled), then ONE fused block per step gives C[c] + rank(B_c, node) and rank(edges, .), fetched
// like in k_find2.
__device__ __forceinline__ void walk_to_sample(const DevImage& img, u64& node, u64& steps, bool live,
                                               ulonglong2* wave_stage, u32 lane)
{
  bool walking = live;
  while(__any(walking))
  {
    u32 idx = 0, r = 0;
    if(walking)
    {
      u32 nib = pred4_get(img.pred4, node);
      if(nib & 8) { walking = false; }                       // sampled(node), gcsa.cpp:883
      else
      {
        u64 b = node / FLB_BITS;
        r = u32(node - b * FLB_BITS);
        idx = u32(u64(nib & 7) * img.flb_nblocks + b);
      }
    }
    if(!__any(walking)) { break; }
    fetch_blocks(img.flb, idx, walking, wave_stage, lane);
    if(walking)
    {
      ulonglong2 blk[8];
      read_block(wave_stage, lane, blk);
      u64 edge, next;
      eval_endpoint(blk, r, 0, edge, next);                  // LF(path_node), gcsa.h:165-183
      node = next; steps++;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

__device__ __forceinline__ u64 first_sample(const DevImage& img, u64 node)      // gcsa.h:202-206
{
  u64 srank = bv_rank(img.sampled, node);
  return (srank > 0 ? bv_select(img.samples, srank) + 1 : 0);
}

// one lane per (query, path node): locateInternal (gcsa.cpp:880-896), wave-cooperative walk.
__global__ __launch_bounds__(TPB2) void k_locate_walk2(DevImage img, const u64* __restrict__ ranges, u64 nq,
                                                      const u64* __restrict__ node_off, const u64* __restrict__ raw_off,
                                                      u64 total_nodes, u64* __restrict__ values, const u64* __restrict__ owners)
{
  __shared__ ulonglong2 stage[TPB2 * 8];
  const u32 lane = threadIdx.x & 63;
  ulonglong2* wave_stage = stage + (threadIdx.x & ~63u) * 8;
  u64 g = u64(blockIdx.x) * TPB2 + threadIdx.x;
  bool live = g < total_nodes;
  u64 node = 0, dest = 0, steps = 0;
  const u64 q = owner_in_wave(owners, node_off, total_nodes, g, live);
  if(live) { locate_item(img, ranges, q, node_off, raw_off, g, node, dest); }
  walk_to_sample(img, node, steps, live, wave_stage, lane);
  if(live)
  {
    u64 s = first_sample(img, node);
    do
    {
      values[dest++] = packed_get(img.stored, img.sample_width, s) + steps;   // gcsa.cpp:893
      s++;
    }
    while(!bv_get(img.samples, s - 1));                      // lastSample, gcsa.h:208
  }
}

// ---- memoised walks ------------------------------------------------------------------------------
// The walk of locateInternal depends on the start node only, so it is done once per path node at
// load time (8 bytes per node; HBM is large) and locate() becomes one gather per path node.
constexpr u64 LOCATE_DIRECT = u64(1) << 63;
constexpr u64 LOCATE_INDEX_BITS = 40, LOCATE_STEP_LIMIT = u64(1) << 23;

__global__ __launch_bounds__(TPB2) void k_build_locate_table(DevImage img, u64 first, u64* __restrict__ table, u32* __restrict__ overflow)
{
  __shared__ ulonglong2 stage[TPB2 * 8];
  const u32 lane = threadIdx.x & 63;
  ulonglong2* wave_stage = stage + (threadIdx.x & ~63u) * 8;
  u64 g = first + u64(blockIdx.x) * TPB2 + threadIdx.x;       // launched in slices: a grid holds < 2^32 threads
  bool live = g < img.n;
  u64 node = g, steps = 0;
  walk_to_sample(img, node, steps, live, wave_stage, lane);
  if(!live) { return; }
  u64 s = first_sample(img, node);
  u64 entry;
  if(bv_get(img.samples, s))            // lastSample(s): a single value
  {
    u64 value = packed_get(img.stored, img.sample_width, s) + steps;
    if(value < LOCATE_DIRECT) { table[g] = LOCATE_DIRECT | value; return; }
  }
  if(steps >= LOCATE_STEP_LIMIT || s >= (u64(1) << LOCATE_INDEX_BITS)) { atomicOr(overflow, 1u); entry = 0; }
  else { entry = s | (steps << LOCATE_INDEX_BITS); }
  table[g] = entry;
}

// Where a path node's values go: raw_off[q] + the values of the query's earlier nodes.  Only the FIRST lane of a query's run
// inside the WAVEFRONT asks the counters for that (SadaSparse::count: two ranks and two selects); the other lanes add the
// wave's running sum of the value counts their table entries show.  No barrier, no LDS: the waves of a workgroup share
// nothing but the owner bracket.
// (Path order: locate(sort = false).  When the values are sorted afterwards the unordered two-pass walk below is used.)
__global__ __launch_bounds__(TPB) void k_locate_tab(DevImage img, const u64* __restrict__ ranges, u64 nq,
                                                    const u64* __restrict__ node_off, const u64* __restrict__ raw_off,
                                                    u64 total_nodes, u64* __restrict__ values, const u64* __restrict__ owners)
{
  const u32 lane = threadIdx.x & 63;
  const u64 g = u64(blockIdx.x) * TPB + threadIdx.x;
  const bool live = g < total_nodes;
  const u64 q = owner_in_wave(owners, node_off, total_nodes, g, live);
  u64 sp = 0, node = 0, entry = LOCATE_DIRECT, s = 0, steps = 0;
  u32 count = 0;
  if(live)
  {
    sp = ranges[2 * q];
    node = sp + (g - node_off[q]);
    entry = img.locate_tab[node];
    count = 1;
    if(!(entry & LOCATE_DIRECT))
    {
      s = entry & ((u64(1) << LOCATE_INDEX_BITS) - 1); steps = entry >> LOCATE_INDEX_BITS;
      count = sample_run(img, s);
    }
  }
  const u64 q_left = __shfl_up(q, 1);
  const bool head = live && (lane == 0 || q_left != q);           // (live lanes are a prefix of the wave)
  u64 base = 0;
  if(head)
  {
    // a query whose values are as many as its path nodes has no node with several values: nothing to ask the counters
    const u64 b = raw_off[q];
    const bool extras = (raw_off[q + 1] - b) != (node_off[q + 1] - node_off[q]);
    base = b + (node - sp) + (node > sp && extras ? sada_sparse_count(img, sp, node - 1) : 0);
  }
  // inclusive scans over the wave: values so far, and the latest run head (lane index + 1)
  u32 sum = count, latest = (head ? lane + 1 : 0);
#pragma unroll
  for(u32 d = 1; d < 64; d <<= 1)
  {
    const u32 other_sum = __shfl_up(sum, d), other_head = __shfl_up(latest, d);
    if(lane >= d) { sum += other_sum; latest = (other_head > latest ? other_head : latest); }
  }
  const u32 before = sum - count;                                 // values of the wave's earlier lanes
  const u32 first = (latest > 0 ? latest - 1 : 0);                // the head of this lane's run (lane 0 is one)
  const u64 run_base = __shfl(base, first);
  const u32 run_before = __shfl(before, first);
  if(!live) { return; }
  u64 dest = run_base + (before - run_before);
  const u64 end = raw_off[q + 1];                                 // (see the guard above)
  if(entry & LOCATE_DIRECT) { if(dest < end) { values[dest] = entry & ~LOCATE_DIRECT; } return; }
  for(u32 j = 0; j < count && dest + j < end; j++) { values[dest + j] = packed_get(img.stored, img.sample_width, s + j) + steps; }     // gcsa.cpp:893
}

// The unordered table walk (the values are sorted afterwards), in two passes.  Elimination runs on the repeat-rich batch
// (110 M path nodes): reading the entries and storing them as they are takes 0.55 ms, the whole kernel took 1.7 ms whatever its
// shape (owner search per workgroup or wavefront, scalar loads for what is uniform, two or eight nodes per lane) -- the two
// path nodes in a hundred whose walk ends in a node with several values (sample-by-sample loop, packed samples, a slot
// counter) sat in three waves out of four and held the other 63 lanes for a chain of dependent gathers.  So the first pass
// stores the single values and only MARKS the other nodes, one 64-bit word per wavefront (a list with one atomic per
// wavefront was tried first: 1.2 M atomics on one counter took 8 ms); the second pass gives every word to one lane, which
// works through its one or two marked nodes.  Everything that is uniform over a wavefront is read through the scalar cache.
// (Round 5: TAB_SPANS spans of 64 nodes per wavefront.  With one, a wavefront lived through three dependent memory latencies --
// the owners, the owner's range / offsets, the table entry -- for 64 nodes, and the CU's 32 wavefronts made that 175 G nodes/s:
// 3.1 ms of the 14 ms of the 16-mer batch on the 2^30-base text.  The owner's range and offsets are now read for every span's
// FIRST owner without asking whether the span has one owner -- no branch in front of the loads, the spans' chains overlap --
// and a span with several owners repairs its lanes afterwards.)
constexpr u32 TAB_SPANS = 4;                  // (2: 1.95 ms, 4: 1.69 ms, 8: 1.82 ms for the 16-mer batch's table pass; 1: 3.17 ms)
__global__ __launch_bounds__(TPB) void k_locate_tab_unordered(DevImage img, const u64* __restrict__ ranges, u64 nq,
                                                              const u64* __restrict__ node_off, const u64* __restrict__ raw_off,
                                                              u64 total_nodes, u64* __restrict__ values, const u64* __restrict__ owners,
                                                              u64* __restrict__ later_words, u64 spans)
{
  const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const u32 lane = threadIdx.x & 63;
  const u64 j0 = (u64(blockIdx.x) * (TPB / OWNER_SPAN) + wave) * TAB_SPANS;   // this wave's spans of OWNER_SPAN = 64 nodes
  if(j0 >= spans) { return; }                                          // uniform
  u64 qo[TAB_SPANS + 1];
#pragma unroll
  for(u32 s = 0; s <= TAB_SPANS; s++) { qo[s] = owners[j0 + s <= spans ? j0 + s : spans]; }      // uniform addresses: scalar loads
  u64 sp[TAB_SPANS], first[TAB_SPANS], b[TAB_SPANS];
#pragma unroll
  for(u32 s = 0; s < TAB_SPANS; s++)
  {
    const u64 q = (qo[s] < nq ? qo[s] : nq - 1);                    // (the entry behind the last span names no query)
    sp[s] = ranges[2 * q]; first[s] = node_off[q]; b[s] = raw_off[q];
  }
  u64 entry[TAB_SPANS];
#pragma unroll
  for(u32 s = 0; s < TAB_SPANS; s++)
  {
    const u64 g_first = (j0 + s) * OWNER_SPAN, g = g_first + lane;
    const bool live = (j0 + s < spans && g < total_nodes);
    if(j0 + s < spans && qo[s] != qo[s + 1])                           // uniform branch: several queries meet in this span
    {
      const u64 g_end = (g_first + OWNER_SPAN <= total_nodes ? g_first + OWNER_SPAN : total_nodes);
      const u64 q = owner_between(node_off, qo[s], qo[s + 1], g_first, g_end, live ? g : g_first);
      sp[s] = ranges[2 * q]; first[s] = node_off[q]; b[s] = raw_off[q];
    }
    entry[s] = (live ? img.locate_tab[sp[s] + (g - first[s])] : LOCATE_DIRECT);
  }
#pragma unroll
  for(u32 s = 0; s < TAB_SPANS; s++)
  {
    const u64 g = (j0 + s) * OWNER_SPAN + lane;
    const bool live = (j0 + s < spans && g < total_nodes);
    if(live && (entry[s] & LOCATE_DIRECT)) { values[b[s] + (g - first[s])] = entry[s] & ~LOCATE_DIRECT; }
    const u64 later = __ballot(live && !(entry[s] & LOCATE_DIRECT));
    if(lane == 0 && j0 + s < spans) { later_words[j0 + s] = later; }
  }
}

// second pass: one lane per word of marks; its path nodes have several values each.  A node with more than COOP_RUN values
// (tandem repeats put hundreds of samples on one node) is unpacked by the whole wavefront, 64 samples at a time.
constexpr u32 COOP_RUN = 16;

__global__ __launch_bounds__(TPB) void k_locate_tab_rest(DevImage img, const u64* __restrict__ ranges, const u64* __restrict__ node_off,
                                                         const u64* __restrict__ raw_off, u64 total_nodes, u64* __restrict__ values,
                                                         const u64* __restrict__ owners, const u64* __restrict__ later_words, u64 spans,
                                                         unsigned long long* __restrict__ extra_slots)
{
  const u64 j = u64(blockIdx.x) * TPB + threadIdx.x;
  const u32 lane = threadIdx.x & 63;
  u64 marks = (j < spans ? later_words[j] : 0);
  u64 qa = 0, qb = 0;
  const u64 g_first = j * OWNER_SPAN;
  const u64 g_end = (g_first + OWNER_SPAN <= total_nodes ? g_first + OWNER_SPAN : total_nodes);
  if(marks != 0) { qa = owners[j]; qb = owners[j + 1]; }
  while(__any(marks != 0))                                   // uniform over the wavefront: the lanes with marks take one node each
  {
    u64 s = 0, steps = 0, at = 0, end = 0;
    u32 count = 0;
    if(marks != 0)
    {
      const u64 g = g_first + u64(__ffsll((long long)marks) - 1);
      marks &= marks - 1;
      const u64 q = owner_between(node_off, qa, qb, g_first, g_end, g);
      const u64 b = raw_off[q], first = node_off[q];
      end = raw_off[q + 1];
      const u64 entry = img.locate_tab[ranges[2 * q] + (g - first)];
      s = entry & ((u64(1) << LOCATE_INDEX_BITS) - 1); steps = entry >> LOCATE_INDEX_BITS;
      count = sample_run(img, s);
      values[b + (g - first)] = packed_get(img.stored, img.sample_width, s) + steps;
      // (the guard below only matters for an index whose counters disagree with its samples: nothing is written outside the query's segment)
      if(count > 1) { at = b + (node_off[q + 1] - first) + atomicAdd(extra_slots + q, (unsigned long long)(count - 1)); }
      if(count <= COOP_RUN)
      {
        for(u32 k = 1; k < count && at + k - 1 < end; k++) { values[at + k - 1] = packed_get(img.stored, img.sample_width, s + k) + steps; }
      }
    }
    u64 big = __ballot(count > COOP_RUN);
    while(big != 0)                                          // uniform
    {
      const u32 src = u32(__ffsll((long long)big)) - 1;
      big &= big - 1;
      const u64 run_s = __shfl(s, src), run_steps = __shfl(steps, src), run_at = __shfl(at, src), run_end = __shfl(end, src);
      const u32 run_count = __shfl(count, src);
      for(u32 k = 1 + lane; k < run_count; k += 64)
      {
        if(run_at + k - 1 < run_end) { values[run_at + k - 1] = packed_get(img.stored, img.sample_width, run_s + k) + run_steps; }
      }
    }
  }
}

// Output slots for a whole workgroup with ONE atomic: every wave passes the number of slots it wants and
// gets the index of its first one.  (One atomic per wave and comp on a single counter was what bounded
// these kernels: ~400 K same-address atomics per level.)  All threads of the workgroup must call it.
template<int THREADS> struct WgSlotsOf { u32 count[THREADS / 64]; unsigned long long base; };
typedef WgSlotsOf<TPB> WgSlots;

template<int THREADS>
__device__ __forceinline__ u64 wg_reserve(WgSlotsOf<THREADS>& sh, unsigned long long* counter, u32 wave_count)
{
  const u32 wave = threadIdx.x >> 6;
  if((threadIdx.x & 63) == 0) { sh.count[wave] = wave_count; }
  __syncthreads();
  if(threadIdx.x == 0)
  {
    u32 total = 0;
    for(u32 w = 0; w < THREADS / 64; w++) { total += sh.count[w]; }
    sh.base = (total > 0 ? atomicAdd(counter, (unsigned long long)total) : 0ull);
  }
  __syncthreads();
  u64 first = sh.base;
  for(u32 w = 0; w < wave; w++) { first += sh.count[w]; }
  __syncthreads();                     // sh may be reused
  return first;
}

// removeDuplicates (utils.h:350-357) sorts the values of a query.  Queries with one value need nothing,
// queries with 2..SMALL_SEGMENT values are sorted in registers by one lane each (k_sort_small), queries with up to
// MEDIUM_SEGMENT values by one wavefront each in LDS (k_sort_medium), queries with up to BIG_SEGMENT values by one workgroup
// each in 64 KB of LDS (k_sort_big: the paper's 16-mers average 7129 values, paper.tex:403); longer ones lose their duplicates
// first (k_dedup_huge) and join those lists, and what still has more than BIG_SEGMENT DISTINCT values is sorted by ONE
// device-wide radix sort over (segment, value) keys (k_over_pack / k_over_unpack).  k_collect_multi lists the medium
// segments (from the end of the segment arrays, downwards), the large ones (from the start; MEDIUM_SEGMENT + 1 ..
// BIG_SEGMENT values) and the huge ones (arrays of their own: up to HUGE_SPLIT values from the start, longer ones from the
// end downwards -- the two instantiations of k_dedup_huge) and publishes the totals below.
constexpr u32 SMALL_SEGMENT = 16;
constexpr u32 MEDIUM_SEGMENT = 1024;
constexpr u32 BIG_SEGMENT = 8192;
constexpr u32 HUGE_SPLIT = BIG_SEGMENT / 2;
// totals of one pass of the locate pipeline (a slot of TOTAL_WORDS u64 in device memory, mirrored to page-locked host memory)
enum { T_NODES = 0, T_RAW = 1, T_LARGE = 2, T_UNIQUE = 3, T_MULTI = 4, T_MEDIUM = 5, T_HUGE_A = 6, T_OVER = 7, T_HUGE_B = 8, T_OVER_VALUES = 9,
       T_BUCKETS = 10, T_SKEW = 11, T_SKEW_VALUES = 12, T_BIG_BUCKETS = 13,
       T_DUPS = 14,         // a flag: some sort met a value equal to its left neighbour (round 6: none -> the sorted values are final where they lie)
       T_CAND = 15,         // ranges listed for the look at their table entries (k_classify_fused)
       T_MID_BUCKETS = 16,  // buckets of k_over_split with BUCKET_BY_WAVE + 1 .. MEDIUM_SEGMENT values (k_sort_bucket<MEDIUM_SEGMENT>)
       TOTAL_WORDS = 24 };

// (Workgroups of 1024 lanes since late in round 6: the lists' counters share a cache line, and the L2 takes the atomics on one line
// one after the other -- with 256 lanes per workgroup, 1563 workgroups x 4-6 reservations were the 139 us this kernel took for 400 k ranges.)
constexpr int COLLECT_THREADS = 1024;
__global__ __launch_bounds__(COLLECT_THREADS) void k_collect_multi(const u64* __restrict__ node_off, const u64* __restrict__ raw_off, u64 nq,
                                                       unsigned long long* __restrict__ totals,
                                                       u64* __restrict__ seg_begin, u64* __restrict__ seg_end,
                                                       u64* __restrict__ huge_begin, u64* __restrict__ huge_end, u32 medium_limit,
                                                       u32 big_limit, const u64* __restrict__ ranges, const u64* __restrict__ locate_tab,
                                                       u64* __restrict__ over_begin, u64* __restrict__ over_end, const u64** __restrict__ over_src)
{
  // the six lists' slots are reserved TOGETHER: every wavefront leaves its six counts in LDS, lanes 0..5 of the workgroup add up a
  // list each and ask its counter once, all in one round trip (six reservations one behind the other, each with its barriers and
  // its wait for the atomic, were most of this kernel)
  constexpr u32 LISTS = 6, WAVES = COLLECT_THREADS / 64;
  enum { L_MULTI = 0, L_LARGE = 1, L_MEDIUM = 2, L_HUGE_A = 3, L_HUGE_B = 4, L_OVER = 5 };
  __shared__ u32 wave_count[LISTS][WAVES];
  __shared__ unsigned long long list_base[LISTS], wave_values[WAVES];
  u64 q = u64(blockIdx.x) * COLLECT_THREADS + threadIdx.x;
  const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if(q == 0) { totals[T_NODES] = node_off[nq]; totals[T_RAW] = raw_off[nq]; }
  u64 b = 0, e = 0;
  bool fused = false;
  if(q < nq)
  {
    b = raw_off[q]; e = raw_off[q + 1];
    fused = (e > b && node_off[q + 1] == node_off[q]);        // values and no path node to walk: k_classify_fused took the nodes out
  }
  const bool is_huge = !fused && e - b > medium_limit && e - b > big_limit;
  u64 mask[LISTS];
  mask[L_MULTI] = __ballot(e - b >= 2);
  mask[L_LARGE] = __ballot(!fused && e - b > medium_limit && e - b <= big_limit);
  mask[L_MEDIUM] = __ballot(!fused && e - b > SMALL_SEGMENT && e - b <= medium_limit);
  mask[L_HUGE_A] = __ballot(is_huge && e - b <= HUGE_SPLIT);
  mask[L_HUGE_B] = __ballot(is_huge && e - b > HUGE_SPLIT);
  mask[L_OVER] = __ballot(fused);
  u64 fused_values = (fused ? e - b : 0);
  for(int o = 32; o > 0; o >>= 1) { fused_values += __shfl_down(fused_values, o, 64); }
  if(lane == 0)
  {
#pragma unroll
    for(u32 t = 0; t < LISTS; t++) { wave_count[t][wave] = u32(__popcll(mask[t])); }
    wave_values[wave] = fused_values;
  }
  __syncthreads();
  if(threadIdx.x < LISTS)
  {
    constexpr u32 counter_of[LISTS] = { T_MULTI, T_LARGE, T_MEDIUM, T_HUGE_A, T_HUGE_B, T_OVER };
    u32 total = 0;
    for(u32 w = 0; w < WAVES; w++) { total += wave_count[threadIdx.x][w]; }
    list_base[threadIdx.x] = (total > 0 ? atomicAdd(totals + counter_of[threadIdx.x], (unsigned long long)total) : 0ull);
  }
  else if(threadIdx.x == 64)                                  // (another wavefront: the sum of the fused ranges' values)
  {
    unsigned long long values = 0;
    for(u32 w = 0; w < WAVES; w++) { values += wave_values[w]; }
    if(values != 0) { atomicAdd(totals + T_OVER_VALUES, values); }
  }
  __syncthreads();
  auto my_slot = [&](u32 t) -> u64
  {
    u64 slot = list_base[t];
    for(u32 w = 0; w < wave; w++) { slot += wave_count[t][w]; }
    return slot + u64(__popcll(mask[t] & ((u64(1) << lane) - 1)));
  };
  if((mask[L_LARGE] >> lane) & 1) { const u64 slot = my_slot(L_LARGE); seg_begin[slot] = b; seg_end[slot] = e; }
  if((mask[L_MEDIUM] >> lane) & 1) { const u64 slot = my_slot(L_MEDIUM); seg_begin[nq - 1 - slot] = b; seg_end[nq - 1 - slot] = e; }      // a query is in at most one of the two lists
  if((mask[L_HUGE_A] >> lane) & 1) { const u64 slot = my_slot(L_HUGE_A); huge_begin[slot] = b; huge_end[slot] = e; }
  if((mask[L_HUGE_B] >> lane) & 1) { const u64 slot = my_slot(L_HUGE_B); huge_begin[nq - 1 - slot] = b; huge_end[nq - 1 - slot] = e; }
  if((mask[L_OVER] >> lane) & 1) { const u64 slot = my_slot(L_OVER); over_begin[slot] = b; over_end[slot] = e; over_src[slot] = locate_tab + ranges[2 * q]; }
}

// The value of lane (l ^ STRIDE), for every lane l.  Strides below 16 stay inside a row of sixteen lanes and go through the
// vector unit's own lane crossbar (DPP: quad permutations for 1 and 2, a pair of masked row shifts for 4, a row rotation for 8);
// 16 and 32 cross rows: gfx950's permlane swaps.  (All 21 steps as ds_bpermute -- what __shfl_xor compiles to -- made the run
// sort of k_over_split wait for the LDS unit 42 times per run: profiles/r05_locate.md.)
template<u32 STRIDE>
__device__ __forceinline__ u64 lane_xor(u64 v)
{
  if constexpr(STRIDE >= 16)
  {
    // gfx950's half exchanges: v_permlane32_swap trades lanes 32-63 of its first operand for lanes 0-31 of the second,
    // v_permlane16_swap rows 1 and 3 of the first for rows 0 and 2 of the second; with both operands = v, every lane finds its
    // partner's word in one of the two results
    const u32 lo = u32(v), hi = u32(v >> 32);
    const u32 lane = __lane_id();
    u32 out_lo, out_hi;
    if constexpr(STRIDE == 32)
    {
      const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false), c = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
      out_lo = (lane < 32 ? a[1] : a[0]); out_hi = (lane < 32 ? c[1] : c[0]);
    }
    else
    {
      const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false), c = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
      out_lo = ((lane & 16) ? a[0] : a[1]); out_hi = ((lane & 16) ? c[0] : c[1]);
    }
    return u64(out_lo) | (u64(out_hi) << 32);
  }
  else
  {
    int lo = int(u32(v)), hi = int(u32(v >> 32));
    // (every lane has a source lane in these three: with bound_ctrl the old value of the destination is not an operand)
    if constexpr(STRIDE == 1) { lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, true); }       // quad_perm [1, 0, 3, 2]
    if constexpr(STRIDE == 2) { lo = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xF, 0xF, true); }       // quad_perm [2, 3, 0, 1]
    if constexpr(STRIDE == 4)
    {
      // banks 0 and 2 (lanes with bit 2 clear) read four lanes up (row_shl:4), banks 1 and 3 four lanes down (row_shr:4)
      const int lo_up = __builtin_amdgcn_update_dpp(lo, lo, 0x104, 0xF, 0x5, false), hi_up = __builtin_amdgcn_update_dpp(hi, hi, 0x104, 0xF, 0x5, false);
      lo = __builtin_amdgcn_update_dpp(lo_up, lo, 0x114, 0xF, 0xA, false); hi = __builtin_amdgcn_update_dpp(hi_up, hi, 0x114, 0xF, 0xA, false);
    }
    if constexpr(STRIDE == 8) { lo = __builtin_amdgcn_update_dpp(0, lo, 0x128, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x128, 0xF, 0xF, true); }      // row_ror:8
    return u64(u32(lo)) | (u64(u32(hi)) << 32);
  }
}

template<u32 K, u32 J>
__device__ __forceinline__ void bitonic_steps(u64& v, u32 lane)
{
  const u64 other = lane_xor<J>(v);
  const bool up = ((lane & K) == 0), lower = ((lane & J) == 0);
  const bool take_min = (up == lower);
  v = (take_min ? (other < v ? other : v) : (other > v ? other : v));
  if constexpr(J > 1) { bitonic_steps<K, J / 2>(v, lane); }
}

// The same network on 32-bit keys, for runs whose values lie within 2^32 of a common base (nearly all: a run spans a few
// buckets of the split).  The run sort is bound by the vector ALU -- 8.4 M runs x 21 steps on the 16-mer batch of the 2^30-base
// text -- and a 64-bit step is a 64-bit compare, four selects and the direction test per lane: ~10 instructions.  Here a step
// is min and max of the lane's key and its partner's (the DPP modifier folds into them) and ONE select whose condition is a
// compile-time lane mask in a scalar register pair: three instructions.
constexpr u64 take_min_mask(u32 K, u32 J)
{
  u64 m = 0;
  for(u32 l = 0; l < 64; l++) { if(((l & K) == 0) == ((l & J) == 0)) { m |= u64(1) << l; } }
  return m;
}
// (`mask` has to be uniform in the compiler's eyes -- a constant, a ballot, scalar64() of something -- or the "s" operand
// arrives as a vector register pair and the assembler refuses the instruction)
__device__ __forceinline__ u32 select_by_mask(u32 if_clear, u32 if_set, u64 mask)
{
  u32 r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(mask));
  return r;
}
template<u32 STRIDE>
__device__ __forceinline__ u32 lane_xor32(u32 v, u32 lane)
{
  if constexpr(STRIDE == 32) { const auto a = __builtin_amdgcn_permlane32_swap(v, v, false, false); return (lane < 32 ? a[1] : a[0]); }
  else if constexpr(STRIDE == 16) { const auto a = __builtin_amdgcn_permlane16_swap(v, v, false, false); return ((lane & 16) ? a[0] : a[1]); }
  else if constexpr(STRIDE == 8) { return u32(__builtin_amdgcn_update_dpp(0, int(v), 0x128, 0xF, 0xF, true)); }
  else if constexpr(STRIDE == 4)
  {
    const int up = __builtin_amdgcn_update_dpp(int(v), int(v), 0x104, 0xF, 0x5, false);
    return u32(__builtin_amdgcn_update_dpp(up, int(v), 0x114, 0xF, 0xA, false));
  }
  else if constexpr(STRIDE == 2) { return u32(__builtin_amdgcn_update_dpp(0, int(v), 0x4E, 0xF, 0xF, true)); }
  else { return u32(__builtin_amdgcn_update_dpp(0, int(v), 0xB1, 0xF, 0xF, true)); }
}
template<u32 K, u32 J>
__device__ __forceinline__ void bitonic_steps32(u32& v, u32 lane)
{
  const u32 other = lane_xor32<J>(v, lane);
  const u32 lo = (other < v ? other : v), hi = (other < v ? v : other);
  v = select_by_mask(hi, lo, take_min_mask(K, J));
  if constexpr(J > 1) { bitonic_steps32<K, J / 2>(v, lane); }
}
// ascending; padding = ~0 (a real key is below that)
__device__ __forceinline__ u32 wave_sort32(u32 v, u32 lane)
{
  bitonic_steps32<2, 1>(v, lane); bitonic_steps32<4, 2>(v, lane); bitonic_steps32<8, 4>(v, lane);
  bitonic_steps32<16, 8>(v, lane); bitonic_steps32<32, 16>(v, lane); bitonic_steps32<64, 32>(v, lane);
  return v;
}

// ascending bitonic sort of one value per lane across the wavefront (64 lanes; padding = ~0 sorts to the end)
__device__ __forceinline__ u64 wave_sort(u64 v, u32 lane)
{
  bitonic_steps<2, 1>(v, lane); bitonic_steps<4, 2>(v, lane); bitonic_steps<8, 4>(v, lane);
  bitonic_steps<16, 8>(v, lane); bitonic_steps<32, 16>(v, lane); bitonic_steps<64, 32>(v, lane);
  return v;
}

// A bitonic sort of up to 1024 values held by ONE wavefront in registers: element e = 64 r + lane is register r of the lane.
// Exchange steps with a stride below 64 cross lanes (lane_xor: DPP / permlane, no LDS), steps with a stride of 64 and more pair
// two registers of the same lane; a step is one 64-bit compare whose result -- a lane mask in a scalar pair -- is combined
// with the step's compile-time direction mask by a scalar instruction, and two selects.  (k_sort_medium and k_sort_bucket ran
// the same network over an array in LDS: two reads, two writes and a barrier per step, whose latency a lone wavefront per
// workgroup cannot hide.)
constexpr u64 lanes_with_bit_clear(u32 bit)     // the lanes l of 0 .. 63 with (l & bit) == 0
{
  return bit == 1 ? 0x5555555555555555ull : bit == 2 ? 0x3333333333333333ull : bit == 4 ? 0x0F0F0F0F0F0F0F0Full
       : bit == 8 ? 0x00FF00FF00FF00FFull : bit == 16 ? 0x0000FFFF0000FFFFull : bit == 32 ? 0x00000000FFFFFFFFull : ~0ull;
}
__device__ __forceinline__ u64 scalar64(u64 x)      // a wave-uniform value, said so to the compiler (the "s" operands below)
{
  return u64(u32(__builtin_amdgcn_readfirstlane(int(u32(x))))) | (u64(u32(__builtin_amdgcn_readfirstlane(int(u32(x >> 32))))) << 32);
}
__device__ __forceinline__ u64 select64_by_mask(u64 if_clear, u64 if_set, u64 mask)
{
  return u64(select_by_mask(u32(if_clear), u32(if_set), mask)) | (u64(select_by_mask(u32(if_clear >> 32), u32(if_set >> 32), mask)) << 32);
}
// (`flip`: 0, or all ones for the same network with every comparison inverted -- a descending sort, a descending merge)
template<u32 R, u32 K, u32 J>
__device__ __forceinline__ void regs_steps(u64 (&v)[R], u64 flip_in = 0)
{
  const u64 flip = scalar64(flip_in);
  if constexpr(J >= 64)
  {
    constexpr u32 rj = J / 64;
#pragma unroll
    for(u32 r = 0; r < R; r++)
    {
      if((r & rj) == 0)
      {
        const bool up = (((r * 64) & K) == 0);
        const u64 a = v[r], c = v[r | rj];
        const u64 greater = __ballot(a > c);
        const u64 swap = (up ? greater : ~greater) ^ flip;
        v[r] = select64_by_mask(a, c, swap); v[r | rj] = select64_by_mask(c, a, swap);
      }
    }
  }
  else
  {
#pragma unroll
    for(u32 r = 0; r < R; r++)
    {
      const u64 other = lane_xor<J>(v[r]);
      const u64 smaller = __ballot(other < v[r]);
      const u64 up = (K < 64 ? lanes_with_bit_clear(K) : ((((r * 64) & K) == 0) ? ~0ull : 0ull));
      const u64 take_min = ~(up ^ lanes_with_bit_clear(J));
      v[r] = select64_by_mask(v[r], other, ~(smaller ^ take_min) ^ flip);     // the partner's value where it is the one this lane keeps
    }
  }
  if constexpr(J > 1) { regs_steps<R, K, J / 2>(v, flip); }
}
template<u32 R, u32 K = 2>
__device__ __forceinline__ void wave_sort_regs(u64 (&v)[R], u64 flip = 0)
{
  regs_steps<R, K, K / 2>(v, flip);
  if constexpr(K < 64 * R) { wave_sort_regs<R, 2 * K>(v, flip); }
}
// The same network on 32-bit keys (ascending only): a cross-lane step is min and max with a DPP operand and one select on a
// compile-time lane mask, a cross-register step min and max alone.
template<u32 R, u32 K, u32 J>
__device__ __forceinline__ void regs_steps32(u32 (&v)[R], u32 lane)
{
  if constexpr(J >= 64)
  {
    constexpr u32 rj = J / 64;
#pragma unroll
    for(u32 r = 0; r < R; r++)
    {
      if((r & rj) == 0)
      {
        const bool up = (((r * 64) & K) == 0);
        const u32 a = v[r], c = v[r | rj];
        const u32 lo = (a < c ? a : c), hi = (a < c ? c : a);
        v[r] = (up ? lo : hi); v[r | rj] = (up ? hi : lo);
      }
    }
  }
  else
  {
#pragma unroll
    for(u32 r = 0; r < R; r++)
    {
      const u32 other = lane_xor32<J>(v[r], lane);
      const u32 lo = (other < v[r] ? other : v[r]), hi = (other < v[r] ? v[r] : other);
      const u64 up = (K < 64 ? lanes_with_bit_clear(K) : ((((r * 64) & K) == 0) ? ~0ull : 0ull));
      v[r] = select_by_mask(hi, lo, ~(up ^ lanes_with_bit_clear(J)));
    }
  }
  if constexpr(J > 1) { regs_steps32<R, K, J / 2>(v, lane); }
}
template<u32 R, u32 K = 2>
__device__ __forceinline__ void wave_sort_regs32(u32 (&v)[R], u32 lane)
{
  regs_steps32<R, K, K / 2>(v, lane);
  if constexpr(K < 64 * R) { wave_sort_regs32<R, 2 * K>(v, lane); }
}
// `len` values at src[0, len) sorted into dst[0, len) (dst may be src) by the wavefront, len <= 64 R.  When all of them share
// their upper 32 bits -- the values of a bucket of k_over_split nearly always do, a query's values when they lie in one 4 G
// stretch of the node numbers -- the lower halves are sorted as 32-bit keys.
// Round 6: the 32-bit network in BLOCKED layout -- element e = R lane + r, a lane holds R neighbours -- and in the form whose
// comparators all point the same way (a stage first compares e with e ^ (K - 1), the mirror image inside its block of K, then
// with e ^ J for J = K / 4 ... 1; the smaller value always goes to the smaller index).  The most frequent strides are the
// smallest ones -- stride 1 occurs in every stage --, and in this layout the strides below R stay inside a lane: a compare-
// exchange is v_min_u32 + v_max_u32 on two registers, no lane exchange, no select, no direction mask (with e = 64 r + lane
// every stride below 64 crossed lanes: 33 of the 36 steps of a 256-value sort).  Strides of R and more cross lanes as before
// (DPP / permlane operand, min, max, a select on a compile-time lane mask).  Counted for 256 values, R = 4: 324 vector
// instructions instead of 408; 512 values, R = 8: 720 instead of 984.  Input order is free (the values arrive unsorted), so
// the loads stay coalesced; the sorted registers go through LDS once to return to e = 64 r + lane for the stores.
// A compare-exchange across lanes in ONE instruction behind the exchange (the end of round 6): the lane with bit BIT of its number
// clear keeps the smaller of its value and its partner's, the other lane the larger -- min(a, b) = med3(a, b, 0), max(a, b) =
// med3(a, b, ~0), so both are v_med3_u32 with a third operand that depends on the lane alone (six such masks in a wavefront, each
// computed once).  v_min + v_max + v_cndmask until then: the register sorts are bound by their vector instructions (97 % VALU busy
// in `tools/pmc_split.sh`'s passes), and two in three of them were these.
__device__ __forceinline__ u32 med3_u32(u32 a, u32 b, u32 c) { u32 r; asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
template<u32 BIT>
__device__ __forceinline__ u32 keeps_larger32(u32 lane) { return 0u - ((lane / BIT) & 1u); }      // all ones in the lanes with the bit set
template<u32 M>
__device__ __forceinline__ u32 lane_mirror32(u32 v, u32 lane)      // the value of lane (l ^ M), M = 2^t - 1
{
  static_assert(M == 1 || M == 3 || M == 7 || M == 15 || M == 31 || M == 63, "a mask of low bits");
  if constexpr(M == 1) { return u32(__builtin_amdgcn_update_dpp(0, int(v), 0xB1, 0xF, 0xF, true)); }          // quad_perm [1, 0, 3, 2]
  else if constexpr(M == 3) { return u32(__builtin_amdgcn_update_dpp(0, int(v), 0x1B, 0xF, 0xF, true)); }     // quad_perm [3, 2, 1, 0]
  else if constexpr(M == 7) { return u32(__builtin_amdgcn_update_dpp(0, int(v), 0x141, 0xF, 0xF, true)); }    // row_half_mirror
  else if constexpr(M == 15) { return u32(__builtin_amdgcn_update_dpp(0, int(v), 0x140, 0xF, 0xF, true)); }   // row_mirror
  else if constexpr(M == 31) { return lane_xor32<16>(lane_mirror32<15>(v, lane), lane); }
  else { return lane_xor32<32>(lane_xor32<16>(lane_mirror32<15>(v, lane), lane), lane); }
}
template<u32 R, u32 J>
__device__ __forceinline__ void blocked_steps32(u32 (&v)[R], u32 lane)      // compare e with e ^ J, then J / 2, ..., 1
{
  if constexpr(J >= R)
  {
    constexpr u32 L = J / R;                                  // partner lane = lane ^ L; the lane with the bit clear keeps the smaller value
#pragma unroll
    for(u32 r = 0; r < R; r++)
    {
      const u32 other = lane_xor32<L>(v[r], lane);
      v[r] = med3_u32(v[r], other, keeps_larger32<L>(lane));
    }
  }
  else
  {
#pragma unroll
    for(u32 r = 0; r < R; r++)
    {
      if((r & J) == 0)
      {
        const u32 a = v[r], c = v[r | J];
        v[r] = (a < c ? a : c); v[r | J] = (a < c ? c : a);
      }
    }
  }
  if constexpr(J > 1) { blocked_steps32<R, J / 2>(v, lane); }
}
template<u32 R, u32 K = 2>
__device__ __forceinline__ void wave_sort_blocked32(u32 (&v)[R], u32 lane)
{
  // the mirror step of stage K: e with e ^ (K - 1)
  if constexpr(K <= R)
  {
#pragma unroll
    for(u32 r = 0; r < R; r++)
    {
      const u32 p = r ^ (K - 1);
      if(r < p) { const u32 a = v[r], c = v[p]; v[r] = (a < c ? a : c); v[p] = (a < c ? c : a); }
    }
  }
  else
  {
    constexpr u32 M = K / R - 1;                              // partner: lane ^ M, register R - 1 - r; the lane whose top bit of M + 1 is clear is the lower one
    u32 other[R];
#pragma unroll
    for(u32 r = 0; r < R; r++) { other[r] = lane_mirror32<M>(v[R - 1 - r], lane); }
#pragma unroll
    for(u32 r = 0; r < R; r++) { v[r] = med3_u32(v[r], other[r], keeps_larger32<(M + 1) / 2>(lane)); }
  }
  if constexpr(K >= 4) { blocked_steps32<R, K / 4>(v, lane); }
  if constexpr(K < 64 * R) { wave_sort_blocked32<R, 2 * K>(v, lane); }
}
// The same network for THREE and SIX values per lane (the end of round 6): 192 and 384 values.  On the 32-mer batch of the 2^30-base
// text two buckets in three hold 257..384 values (profiles/r06_locate.md section 11) and were sorted in eight registers per lane
// that they filled to 60 %.  Nothing in the merge needs a power of two: a lane's values are sorted by a sorting network (3 or 12
// comparators), a stage compares element e with its mirror image in the block of twice the size -- lane ^ M, register R - 1 - r --,
// then lane with lane ^ L for L = (M + 1) / 4 ... 1 (every half-cleaner works on an even number of elements), and what a lane holds
// then is a bitonic sequence of R values: for six, compare r with r + 3 and sort the triples.  (tests/test_sort_network_model.py is
// the model this was transcribed from: 2 .. 64 lanes, random inputs and 0-1 inputs.)
__device__ __forceinline__ void compare_exchange32(u32& a, u32& c) { const u32 lo = (a < c ? a : c), hi = (a < c ? c : a); a = lo; c = hi; }
template<u32 R>
__device__ __forceinline__ void lane_sort_odd32(u32 (&v)[R])            // any order -> ascending
{
  static_assert(R == 3 || R == 6, "three or six values per lane");
  if constexpr(R == 3) { compare_exchange32(v[0], v[1]); compare_exchange32(v[1], v[2]); compare_exchange32(v[0], v[1]); }
  else
  {
    compare_exchange32(v[0], v[5]); compare_exchange32(v[1], v[3]); compare_exchange32(v[2], v[4]);
    compare_exchange32(v[1], v[2]); compare_exchange32(v[3], v[4]);
    compare_exchange32(v[0], v[3]); compare_exchange32(v[2], v[5]);
    compare_exchange32(v[0], v[1]); compare_exchange32(v[2], v[3]); compare_exchange32(v[4], v[5]);
    compare_exchange32(v[1], v[2]); compare_exchange32(v[3], v[4]);
  }
}
template<u32 R>
__device__ __forceinline__ void lane_bitonic_odd32(u32 (&v)[R])            // a bitonic sequence of R values -> ascending
{
  if constexpr(R == 3) { lane_sort_odd32<3>(v); }
  else
  {
    compare_exchange32(v[0], v[3]); compare_exchange32(v[1], v[4]); compare_exchange32(v[2], v[5]);
    compare_exchange32(v[0], v[1]); compare_exchange32(v[1], v[2]); compare_exchange32(v[0], v[1]);
    compare_exchange32(v[3], v[4]); compare_exchange32(v[4], v[5]); compare_exchange32(v[3], v[4]);
  }
}
template<u32 R, u32 L>
__device__ __forceinline__ void lane_halves_odd32(u32 (&v)[R], u32 lane)   // lane with lane ^ L, then L / 2, ..., 1: the lane with the bit clear keeps the smaller value
{
  if constexpr(L >= 1)
  {
#pragma unroll
    for(u32 r = 0; r < R; r++)
    {
      const u32 other = lane_xor32<L>(v[r], lane);
      v[r] = med3_u32(v[r], other, keeps_larger32<L>(lane));
    }
    if constexpr(L > 1) { lane_halves_odd32<R, L / 2>(v, lane); }
  }
}
template<u32 R, u32 M = 1>
__device__ __forceinline__ void wave_sort_blocked32_odd(u32 (&v)[R], u32 lane)      // element R lane + r; M + 1 = lanes per block of this stage
{
  if constexpr(M == 1) { lane_sort_odd32<R>(v); }
  u32 other[R];
#pragma unroll
  for(u32 r = 0; r < R; r++) { other[r] = lane_mirror32<M>(v[R - 1 - r], lane); }
#pragma unroll
  for(u32 r = 0; r < R; r++) { v[r] = med3_u32(v[r], other[r], keeps_larger32<(M + 1) / 2>(lane)); }
  if constexpr(M >= 3) { lane_halves_odd32<R, (M + 1) / 4>(v, lane); }
  lane_bitonic_odd32<R>(v);
  if constexpr(M < 63) { wave_sort_blocked32_odd<R, 2 * M + 1>(v, lane); }
}
// sorted registers in blocked layout (element R lane + r) -> the layout of the loads and stores (element 64 r + lane), through
// `stage` (64 R words of LDS that belong to this wavefront)
template<u32 R>
__device__ __forceinline__ void blocked_to_striped32(u32 (&v)[R], u32* stage, u32 lane)
{
  if constexpr(R == 1) { return; }
#pragma unroll
  for(u32 r = 0; r < R; r++) { stage[R * lane + r] = v[r]; }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for(u32 r = 0; r < R; r++) { v[r] = stage[64 * r + lane]; }
  __builtin_amdgcn_wave_barrier();
}

// Values of a wavefront's sorted registers (element e = 64 r + lane, ascending, the first `len` real) that equal their left
// neighbour, summed over the wavefront: what removeDuplicates (utils.h:350-357) will drop.  Round 6: the sorts report them, and
// a batch without any needs no compaction -- its values are sorted in the caller's buffer, at the offsets the size scan gave.
// inclusive prefix sums over the wavefront by DPP: row_shr 1, 2, 4, 8 inside the rows of 16 lanes (a lane without a source adds
// 0), then the last lane of row 0 / 2 into rows 1 / 3 (row_bcast:15) and lane 31 into the upper half (row_bcast:31)
__device__ __forceinline__ u32 wave_scan_dpp(u32 x)
{
  x += u32(__builtin_amdgcn_update_dpp(0, int(x), 0x111, 0xF, 0xF, true));
  x += u32(__builtin_amdgcn_update_dpp(0, int(x), 0x112, 0xF, 0xF, true));
  x += u32(__builtin_amdgcn_update_dpp(0, int(x), 0x114, 0xF, 0xF, true));
  x += u32(__builtin_amdgcn_update_dpp(0, int(x), 0x118, 0xF, 0xF, true));
  x += u32(__builtin_amdgcn_update_dpp(0, int(x), 0x142, 0xA, 0xF, false));
  x += u32(__builtin_amdgcn_update_dpp(0, int(x), 0x143, 0xC, 0xF, false));
  return x;
}

// (the left neighbour by DPP wave_shr:1 -- lane 0 keeps its own value and is overruled --, the last lane of the register before by
// v_readlane, the count by ballots: no ds_bpermute, which is what __shfl_up / __shfl / __shfl_down compile to -- 21 trips through the
// LDS crossbar per 512 sorted values until late in round 6)
__device__ __forceinline__ u32 wave_shr1(u32 v) { return u32(__builtin_amdgcn_update_dpp(int(v), int(v), 0x138, 0xF, 0xF, false)); }
__device__ __forceinline__ u64 wave_shr1(u64 v) { return (u64(wave_shr1(u32(v >> 32))) << 32) | u64(wave_shr1(u32(v))); }
__device__ __forceinline__ u32 last_lane_value(u32 v) { return u32(__builtin_amdgcn_readlane(int(v), 63)); }
__device__ __forceinline__ u64 last_lane_value(u64 v) { return (u64(last_lane_value(u32(v >> 32))) << 32) | u64(last_lane_value(u32(v))); }
template<u32 R, class T>
__device__ __forceinline__ u32 dups_in_regs(const T (&v)[R], u32 len, u32 lane)
{
  u32 total = 0;                                              // (uniform)
#pragma unroll
  for(u32 r = 0; r < R; r++)
  {
    T left = wave_shr1(v[r]);
    if(r > 0) { const T wrap = last_lane_value(v[r - 1]); left = (lane == 0 ? wrap : left); }
    const u32 e = r * 64 + lane;
    total += u32(__popcll(__ballot(e > 0 && e < len && v[r] == left)));
  }
  return total;
}
// `len` values at src[0, len) sorted into dst[0, len) (dst may be src) by the wavefront, len <= 64 R.  When all of them share
// their upper 32 bits -- the values of a bucket of k_over_split nearly always do, a query's values when they lie in one 4 G
// stretch of the node numbers -- the lower halves are sorted as 32-bit keys.  MASK: bits cleared from what is read (the
// locate table's direct flag, when src is the table itself).  Returns the number of duplicates (dups_in_regs).
template<u32 R>
__device__ __forceinline__ u32 sort_segment_regs(const u64* src, u64* dst, u32 len, u32 lane, u64 keep = ~u64(0), u32* stage = nullptr)
{
  u64 v[R];
#pragma unroll
  for(u32 r = 0; r < R; r++) { v[r] = (r * 64 + lane < len ? (src[r * 64 + lane] & keep) : ~u64(0)); }      // padding sorts to the end
  const u32 top = u32(__builtin_amdgcn_readfirstlane(int(u32(v[0] >> 32))));      // (element 0 exists: len >= 1)
  bool differs = false;
#pragma unroll
  for(u32 r = 0; r < R; r++) { differs = differs || (r * 64 + lane < len && u32(v[r] >> 32) != top); }
  if(__ballot(differs) == 0)                                  // (uniform)
  {
    u32 key[R];
#pragma unroll
    for(u32 r = 0; r < R; r++) { key[r] = (r * 64 + lane < len ? u32(v[r]) : ~u32(0)); }           // (a key of all ones ties with the padding: the same value either way)
    if constexpr(R == 3 || R == 6) { wave_sort_blocked32_odd<R>(key, lane); blocked_to_striped32<R>(key, stage, lane); }      // (only called with a transposition buffer)
    else if(stage != nullptr) { wave_sort_blocked32<R>(key, lane); blocked_to_striped32<R>(key, stage, lane); }      // (uniform: the kernel has a transposition buffer)
    else { wave_sort_regs32<R>(key, lane); }
#pragma unroll
    for(u32 r = 0; r < R; r++) { if(r * 64 + lane < len) { dst[r * 64 + lane] = (u64(top) << 32) | key[r]; } }
    return dups_in_regs<R>(key, len, lane);
  }
  if constexpr(R == 3 || R == 6)
  {
    // (values on both sides of a multiple of 2^32: the 64-bit network wants a power of two -- padded; the padding sorts to the end)
    constexpr u32 P = (R == 3 ? 4 : 8);
    u64 w[P];
#pragma unroll
    for(u32 r = 0; r < P; r++) { w[r] = (r < R ? v[r] : ~u64(0)); }
    wave_sort_regs<P>(w);
#pragma unroll
    for(u32 r = 0; r < R; r++) { v[r] = w[r]; }
  }
  else { wave_sort_regs<R>(v); }
#pragma unroll
  for(u32 r = 0; r < R; r++) { if(r * 64 + lane < len) { dst[r * 64 + lane] = v[r]; } }
  return dups_in_regs<R>(v, len, lane);
}
// (MOST: the longest segment the caller passes, 64 R_max -- the register budget of the kernel follows from it)
template<u32 MOST = MEDIUM_SEGMENT>
__device__ __forceinline__ u32 sort_segment_by_wave(const u64* src, u64* dst, u32 len, u32 lane, u64 keep = ~u64(0), u32* stage = nullptr)      // len <= MOST (uniform)
{
  if(len <= 64) { return sort_segment_regs<1>(src, dst, len, lane, keep, stage); }
  else if(len <= 128) { return sort_segment_regs<2>(src, dst, len, lane, keep, stage); }
  else if(len <= 192 && stage != nullptr) { return sort_segment_regs<3>(src, dst, len, lane, keep, stage); }
  else if(len <= 256 || MOST <= 256) { return sort_segment_regs<4>(src, dst, len, lane, keep, stage); }
  else if(len <= 384 && stage != nullptr) { return sort_segment_regs<6>(src, dst, len, lane, keep, stage); }
  else if(len <= 512 || MOST <= 512) { return sort_segment_regs<8>(src, dst, len, lane, keep, stage); }
  else { return sort_segment_regs<16>(src, dst, len, lane, keep, stage); }
}
// totals[T_DUPS] is a FLAG with a count's type: non-zero iff some sort met a duplicate.  A wavefront that has some adds them
// only while the word still reads zero -- on a variation graph most queries have duplicates, and a hundred thousand
// wavefronts adding to one address would cost more than the sorts.
__device__ __forceinline__ void flag_dups(unsigned long long* flag, u32 dups)
{
  if(dups != 0 && __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) { atomicAdd(flag, (unsigned long long)dups); }
}
__device__ __forceinline__ void report_dups(unsigned long long* totals, u32 dups, u32 lane)
{
  if(lane == 0) { flag_dups(totals + T_DUPS, dups); }
}

// one wavefront (= one workgroup) per query with SMALL_SEGMENT + 1 .. MEDIUM_SEGMENT values: bitonic sort in registers
// (wave_sort_regs; in LDS until round 5), in place.  Segment s of the list is the one at seg_begin / seg_end [last - s].
// (The grid is an upper bound -- the duplicate filter appends to the list while the host is not looking --; `count` on the
// device says how many segments there are.)
__global__ __launch_bounds__(64) void k_sort_medium(const u64* __restrict__ seg_begin, const u64* __restrict__ seg_end, u64 last,
                                                    u64* __restrict__ values, unsigned long long* __restrict__ totals)
{
  __shared__ u32 stage[MEDIUM_SEGMENT];                     // (the transposition of the blocked 32-bit network)
  const u32 lane = threadIdx.x;
  if(blockIdx.x >= totals[T_MEDIUM]) { return; }
  const u64 b = seg_begin[last - blockIdx.x];
  const u32 len = u32(seg_end[last - blockIdx.x] - b);
  report_dups(totals, sort_segment_by_wave(values + b, values + b, len, lane, ~u64(0), stage), lane);
}

// one workgroup per LARGE segment (list from the start of the segment arrays) with at most BIG_SEGMENT values: bitonic sort
// in registers and (for the widest strides) LDS, in place (longer segments are on the list of the segmented radix sort).  Round 2 sent everything above 1024 values
// there: on a repeat-rich index that library call was 80 % of locate() (profiles/r03_locate.md).
// Two instantiations share the list: CAPACITY 4096 takes the segments of up to 4096 values in 32 KB of LDS (five workgroups
// per CU), CAPACITY 8192 the rest in 64 KB (two per CU); a workgroup whose segment belongs to the other one exits at once.
// Round 5: every wavefront of the workgroup holds 1024 values in registers (wave_sort_regs) and sorts them there, ascending or
// descending by its parity; the stages above 1024 exchange whole registers between wavefronts through LDS for the strides of
// 1024 and more -- one to three exchanges per stage, six for 8192 values -- and finish in registers again.  (All 78 / 91 steps
// ran over the LDS array with a barrier each until then.)  CAPACITY / 16 threads.
template<u32 CAPACITY> constexpr int big_threads() { return int(CAPACITY / 16); }
// `source`: where the unsorted values are (the same offsets); nullptr = in place.  (The buckets of k_over_split are read from
// its scratch array and land, sorted, in the values array.)
template<u32 CAPACITY, u32 ABOVE>
__global__ __launch_bounds__(CAPACITY / 16) void k_sort_big(const u64* __restrict__ seg_begin, const u64* __restrict__ seg_end,
                                                           u64* values, const unsigned long long* __restrict__ count, unsigned long long* __restrict__ dup_counter,
                                                           const u64* source = nullptr,
                                                           u64 from_end = 0)      // from_end != 0: segment s of the list is at [from_end - s]
{
  __shared__ u64 buf[CAPACITY];
  __shared__ u64 tails[CAPACITY / 1024];                    // the last value of every wavefront's sorted stretch
  const u32 tid = threadIdx.x, lane = tid & 63;
  const u32 wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (a scalar for the compiler too: the direction masks below live in scalar pairs)
  if(blockIdx.x >= *count) { return; }                      // the grid is an upper bound (see k_sort_medium)
  const u64 at = (from_end != 0 ? from_end - blockIdx.x : u64(blockIdx.x));
  const u64 b = seg_begin[at];
  const u32 len = u32(seg_end[at] - b);                    // <= BIG_SEGMENT: k_collect_multi
  if(len > CAPACITY || len <= ABOVE) { return; }            // the other instantiation's segment (uniform per workgroup)
  u32 n2 = 1024;
  while(n2 < len) { n2 <<= 1; }
  const u64* from = (source != nullptr ? source : values);
  const u32 first = wave * 1024;                            // this wavefront's elements: first + 64 r + lane
  const bool active = (first < n2);
  u64 v[16];
#pragma unroll
  for(u32 r = 0; r < 16; r++) { const u32 e = first + r * 64 + lane; v[r] = (active && e < len ? from[b + e] : ~u64(0)); }    // padding sorts to the end
  // Round 6: a segment whose values share their upper 32 bits -- the distinct values the duplicate filter leaves on an index of
  // fewer than 2^32 positions, the long buckets of a split -- is sorted as 32-bit keys by the network of wave_sort_blocked32
  // (element 16 lane + r of a wavefront's 1024, every comparator pointing the same way): the stage that joins blocks of K / 2
  // compares e with its mirror image e ^ (K - 1) -- the partner wavefront's registers in reverse, through LDS --, then e ^ J for
  // J = K / 4 ... 1024 through LDS and J = 512 ... 1 in registers.  Half the registers, half the LDS traffic, and a compare-
  // exchange is two instructions instead of the six of a 64-bit one.
  __shared__ u32 wide;
  if(tid == 0) { wide = 0; }
  __syncthreads();
  const u32 top = u32(from[b] >> 32);                         // (uniform: len >= 1)
  {
    bool differs = false;
#pragma unroll
    for(u32 r = 0; r < 16; r++) { const u32 e = first + r * 64 + lane; differs = differs || (active && e < len && u32(v[r] >> 32) != top); }
    if(__ballot(differs) != 0 && lane == 0) { wide = 1; }
  }
  __syncthreads();
  if(wide == 0)                                               // (uniform)
  {
    u32* buf32 = reinterpret_cast<u32*>(buf);
    u32 key[16];
#pragma unroll
    for(u32 r = 0; r < 16; r++) { const u32 e = first + r * 64 + lane; key[r] = (active && e < len ? u32(v[r]) : ~u32(0)); }      // (a key of all ones ties with the padding: the same value either way)
    if(active) { wave_sort_blocked32<16>(key, lane); }
    for(u32 K = 2048; K <= n2; K <<= 1)
    {
      for(u32 J = K >> 1; J >= 1024; J >>= 1)
      {
        const bool mirror = (J == (K >> 1));                    // the first step of the stage
        if(active)
        {
#pragma unroll
          for(u32 r = 0; r < 16; r++) { buf32[first + 64 * r + lane] = key[r]; }      // (element 16 lane + r lies at word 64 r + lane: no bank is asked twice, forwards or mirrored)
        }
        __syncthreads();
        if(active)
        {
          const bool keep_min = ((first & J) == 0);
          const u32 partner = (mirror ? (first ^ (K - 1024)) + 63 - lane : (first ^ J) + lane);      // (the mirror image of element 16 lane + r: 16 (63 - lane) + 15 - r)
#pragma unroll
          for(u32 r = 0; r < 16; r++)
          {
            const u32 other = (mirror ? buf32[partner + 64 * (15 - r)] : buf32[partner + 64 * r]);
            key[r] = (keep_min ? (other < key[r] ? other : key[r]) : (other > key[r] ? other : key[r]));
          }
        }
        __syncthreads();
      }
      if(active) { blocked_steps32<16, 512>(key, lane); }
    }
    if(active)
    {
      blocked_to_striped32<16>(key, buf32 + first, lane);       // (this wavefront's own 4 KB; every exchange above ended behind a barrier)
#pragma unroll
      for(u32 r = 0; r < 16; r++) { const u32 e = first + r * 64 + lane; if(e < len) { values[b + e] = (u64(top) << 32) | key[r]; } }
      if(lane == 63) { tails[wave] = (u64(top) << 32) | key[15]; }
    }
    __syncthreads();
    if(active)
    {
      u32 dups = dups_in_regs<16>(key, (len > first ? len - first : 0u), lane);
      if(lane == 0 && wave > 0 && first < len && ((u64(top) << 32) | key[0]) == tails[wave - 1]) { dups++; }
      if(lane == 0) { flag_dups(dup_counter, dups); }
    }
    return;
  }
  if(active) { wave_sort_regs<16>(v, (n2 > 1024 && (wave & 1)) ? ~u64(0) : u64(0)); }
  for(u32 K = 2048; K <= n2; K <<= 1)
  {
    const bool descending = (K < n2 && (first & K) != 0);   // (uniform per wavefront)
    for(u32 J = K >> 1; J >= 1024; J >>= 1)
    {
      if(active)
      {
#pragma unroll
        for(u32 r = 0; r < 16; r++) { buf[first + r * 64 + lane] = v[r]; }
      }
      __syncthreads();
      if(active)
      {
        const bool keep_min = (((first & J) == 0) != descending);
#pragma unroll
        for(u32 r = 0; r < 16; r++)
        {
          const u64 other = buf[(first ^ J) + r * 64 + lane];
          v[r] = (keep_min ? (other < v[r] ? other : v[r]) : (other > v[r] ? other : v[r]));
        }
      }
      __syncthreads();
    }
    if(active) { regs_steps<16, 2048, 512>(v, descending ? ~u64(0) : u64(0)); }
  }
  if(active)
  {
#pragma unroll
    for(u32 r = 0; r < 16; r++) { const u32 e = first + r * 64 + lane; if(e < len) { values[b + e] = v[r]; } }
    if(lane == 63) { tails[wave] = v[15]; }
  }
  __syncthreads();
  if(active)                                                  // duplicates: inside the wavefront's stretch, and against the stretch before it
  {
    u32 dups = dups_in_regs<16>(v, (len > first ? len - first : 0u), lane);
    if(lane == 0 && wave > 0 && first < len && v[0] == tails[wave - 1]) { dups++; }
    if(lane == 0) { flag_dups(dup_counter, dups); }
  }
}

// one workgroup per HUGE segment (more than BIG_SEGMENT values before deduplication): on a repeat-rich index such a segment
// holds a few distinct values many times over (five raw values per distinct one in profiles/r03_locate.md), so the duplicates
// are removed BEFORE sorting, through a hash set in LDS.  (With the filter on, k_collect_multi lists every segment beyond the
// medium class here: a segment without duplicates pays one extra pass, a tenth of its sort.)  A segment with at most BIG_SEGMENT distinct values leaves
// as: its distinct values (unsorted) at the front, the rest of the segment filled with its largest value (duplicates that
// the flag pass drops), and the front appended to the medium or large list for the LDS sorts that run next.  A segment with
// more distinct values is left untouched and listed for the device-wide radix sort (over_begin / over_end, counted in totals[T_OVER]).
// Two instantiations, each with its own list (k_collect_multi): 8192 slots (64 KB, two workgroups per CU) take the segments
// of up to HUGE_SPLIT = 4096 values, which cannot overflow (list from the start of huge_begin / huge_end); 16384 slots (128 KB)
// the longer ones (FROM_END: segment s of the list is at [last - s]).  512 threads with 64 KB, 1024 with 128 KB (one per CU).
// A segment that overflows stops reading at once: a range of 200 000 distinct values costs the 16 000 it took to find out.
constexpr u64 HUGE_EMPTY = ~u64(0);

// (Reading the locate table from this kernel instead of the walk's output -- the raw values of these queries never written --
// was measured: 8.7 against 5.9 ms for the repeat-rich batch.  One or two workgroups per CU do not hide the latency of the
// table gathers; the walk kernel with a lane per path node does.)
// WORD (round 6): the table's slot type.  unsigned long long: any value.  u32: an index whose values lie below 2^32 - 1 (the host
// knows: sample_width <= 31) -- the compare-and-swap chains are what this kernel waits for, a 32-bit one is issued at twice the
// rate, and the table of the long segments is 64 KB instead of 128, so that two workgroups share a CU; a value that does not fit
// makes the segment overflow to the split sort (a detour, never a wrong result).
template<u32 HUGE_SLOTS, bool FROM_END, int HUGE_THREADS, class WORD>
__global__ __launch_bounds__(HUGE_THREADS) void k_dedup_huge(const u64* __restrict__ huge_begin, const u64* __restrict__ huge_end, u64 last,
                                                            u64* __restrict__ values, u64 nq, u32 medium_limit,
                                                            unsigned long long* __restrict__ totals,
                                                            u64* __restrict__ seg_begin, u64* __restrict__ seg_end,
                                                            u64* __restrict__ over_begin, u64* __restrict__ over_end, const u64** __restrict__ over_src,
                                                            unsigned long long* __restrict__ dead)
{
  __shared__ WORD table[HUGE_SLOTS];
  constexpr WORD EMPTY = WORD(~WORD(0));
  constexpr bool NARROW = (sizeof(WORD) == 4);
  __shared__ u32 distinct, has_ones, too_wide;
  __shared__ unsigned long long largest;
  const u32 tid = threadIdx.x;
  const u64 at = (FROM_END ? last - blockIdx.x : u64(blockIdx.x));
  const u64 b = huge_begin[at], e = huge_end[at], len = e - b;
  constexpr u32 MOST = HUGE_SLOTS / 2;                      // distinct values a segment may have here
  // No insertion once more than MOST distinct values have been seen: the segment has overflowed, nothing more to learn.  (Lanes
  // that passed the test together still insert: at most MOST + HUGE_THREADS values, well below HUGE_SLOTS, so the probing always
  // ends.  Round 3 filled the table to HUGE_SLOTS - HUGE_THREADS first: linear probing at a load of 0.94 under 1024 lanes of
  // LDS atomics made every all-distinct segment cost a millisecond, 93 ms of a 134 ms batch; profiles/r04_locate.md.)
  constexpr u32 STOP = MOST + 1;
  constexpr u32 AHEAD = 4;                                    // values a lane has loaded before it inserts the first of them
  static_assert(MOST + 1 + AHEAD * HUGE_THREADS < HUGE_SLOTS, "the table must keep free slots");
  // A look at a SAMPLE first (round 5): 512 values at equal distances.  When no two of them are equal the segment almost
  // certainly has more than MOST distinct values -- with D <= 8192 distinct values among 512 random positions ~16 equal pairs
  // are expected, none with probability e^-16 -- and is listed for the split sort at once: a wrong guess costs time there, never
  // a result.  (On the 16-mer batch of the 2^30-base text every one of the 13 899 such segments overflowed after its workgroup
  // had cleared 128 KB of LDS and inserted ~10 000 values: 0.9 of the batch's 12.8 ms, 1.2 GB read for nothing.)
  constexpr u32 SAMPLE = 512, SAMPLE_SLOTS = 2048;
  static_assert(SAMPLE <= HUGE_THREADS && SAMPLE_SLOTS <= HUGE_SLOTS, "the sample uses the front of the table");
  // (the first values of every lane are requested BEFORE the sample is looked at: their round trip passes behind it; a segment
  // the sample sends away has read 32 KB for nothing)
  const u64 items = len;
  u64 got[AHEAD];
#pragma unroll
  for(u32 j = 0; j < AHEAD; j++)
  {
    const u64 i = u64(tid) + u64(j) * HUGE_THREADS;
    got[j] = 0;
    if(i < items) { got[j] = values[b + i]; }
  }
  if(len > MOST)                                              // (uniform; otherwise the segment cannot overflow)
  {
    for(u32 i = tid; i < SAMPLE_SLOTS; i += HUGE_THREADS) { table[i] = EMPTY; }
    if(tid == 0) { distinct = 0; }
    __syncthreads();
    if(tid < SAMPLE)
    {
      const u64 v = values[b + (u64(tid) * len) / SAMPLE];
      u32 slot = u32((v * 0x9E3779B97F4A7C15ull) >> 32) & (SAMPLE_SLOTS - 1);
      while(v != HUGE_EMPTY && !(NARROW && v >= u64(EMPTY)))
      {
        const WORD prev = atomicCAS(&table[slot], EMPTY, WORD(v));
        if(prev == EMPTY) { break; }
        if(prev == WORD(v)) { atomicAdd(&distinct, 1u); break; }    // (here: equal pairs seen)
        slot = (slot + 1) & (SAMPLE_SLOTS - 1);
      }
    }
    __syncthreads();
    const bool all_different = (distinct == 0);               // uniform: read after the barrier
    __syncthreads();
    if(all_different)
    {
      if(tid == 0)
      {
        const u64 slot = atomicAdd(totals + T_OVER, 1ull);
        over_begin[slot] = b; over_end[slot] = e; over_src[slot] = nullptr;
        atomicAdd(totals + T_OVER_VALUES, (unsigned long long)len);
      }
      return;
    }
  }
  for(u32 i = tid; i < HUGE_SLOTS; i += HUGE_THREADS) { table[i] = EMPTY; }
  if(tid == 0) { distinct = 0; has_ones = 0; largest = 0; too_wide = 0; }
  __syncthreads();
  auto insert = [&](u64 v)
  {
    if(v == HUGE_EMPTY) { has_ones = 1; return; }
    if(NARROW && v >= u64(EMPTY)) { too_wide = 1; return; }                    // (not an index this instantiation was chosen for: overflow)
    if(*reinterpret_cast<volatile u32*>(&distinct) >= STOP) { return; }        // the segment has overflowed
    u32 slot = u32((v * 0x9E3779B97F4A7C15ull) >> 32) & (HUGE_SLOTS - 1);
    while(true)
    {
      const WORD prev = atomicCAS(&table[slot], EMPTY, WORD(v));
      if(prev == EMPTY) { atomicAdd(&distinct, 1u); break; }
      if(prev == WORD(v)) { break; }
      slot = (slot + 1) & (HUGE_SLOTS - 1);
    }
  };
  // no barrier inside the loop: insert() stops by itself before the table fills, and whether the segment overflowed is
  // only asked at the end; AHEAD independent loads per lane are in flight before the first insertion.
  // Round 6 timed the phases of this kernel on the 16-mer batch of the 2^23 repeat graph (45 900 segments of ~12 000 values, two
  // workgroups per CU; profiles/r06_locate.md section 6): launch + the list entry 0.53 ms, + the loads 0.9, + the insertions 2.1-2.4,
  // + the sample 0.3, + sweep, tail and listing 0.9-1.7.  What did NOT move the insertions: a 32-bit table (half the LDS, a
  // compare-and-swap of half the width), a plain read of the slot before the compare-and-swap, neither counter operation removed
  // (2.47 -> 2.28 without both), the count of distinct values kept per wavefront (2.09 -> 2.53), the next group requested before
  // this one is inserted (2.09 -> 2.37; again with clamped indices instead of branches around the loads and this group's values
  // pinned, so that the compiler's wait is not for the loads just issued -- what repaired k_over_split's prefetch --: 2.90 -> 3.33 ms
  // for the whole kernel, same box, two runs each).  Every phase of a workgroup waits for the one before it and only two workgroups share a
  // CU: the kernel is a chain of latencies, not a rate.
  for(u64 base = tid; base < items; base += AHEAD * HUGE_THREADS)
  {
    if(*reinterpret_cast<volatile u32*>(&distinct) >= STOP) { break; }           // overflowed: nothing more to learn
    if(base != tid)
    {
#pragma unroll
      for(u32 j = 0; j < AHEAD; j++)
      {
        const u64 i = base + u64(j) * HUGE_THREADS;
        got[j] = 0;
        if(i < items) { got[j] = values[b + i]; }
      }
    }
#pragma unroll
    for(u32 j = 0; j < AHEAD; j++)
    {
      if(base + u64(j) * HUGE_THREADS < items) { insert(got[j]); }
    }
  }
  __syncthreads();
  const bool overflow = (distinct + has_ones > MOST || too_wide != 0);          // uniform: read after the barrier
  if(overflow)
  {
    if(tid == 0)
    {
      const u64 slot = atomicAdd(totals + T_OVER, 1ull);
      over_begin[slot] = b; over_end[slot] = e; over_src[slot] = nullptr;
      atomicAdd(totals + T_OVER_VALUES, (unsigned long long)len);
    }
    return;
  }
  // (the count is final: the segment is listed for its sort NOW, so that the device-wide atomics' round trips pass while the
  // table is swept -- with one or two workgroups on a CU nobody else would hide them)
  if(tid == 0)
  {
    const u32 final_count = distinct + (has_ones ? 1u : 0u);
    if(len > final_count) { flag_dups(totals + T_DUPS, 1u); }      // (the tail: duplicates, or dead slots -- either way the values are not final where they lie)
    if(final_count >= 2)
    {
      if(final_count <= medium_limit && medium_limit > SMALL_SEGMENT)
      {
        const u64 slot = atomicAdd(totals + T_MEDIUM, 1ull);
        seg_begin[nq - 1 - slot] = b; seg_end[nq - 1 - slot] = b + final_count;
      }
      else
      {
        const u64 slot = atomicAdd(totals + T_LARGE, 1ull);
        seg_begin[slot] = b; seg_end[slot] = b + final_count;
      }
    }
  }
  // The table is swept ONCE into registers; a scan over the workgroup's counts places every lane's values (round 6: sixteen
  // rounds of an LDS atomic on the one cursor, each waited for, were a quarter of this kernel).
  unsigned long long mine = 0;
  static_assert(HUGE_SLOTS % HUGE_THREADS == 0, "the sweep over the table has a uniform trip count");
  constexpr u32 PER_LANE = HUGE_SLOTS / HUGE_THREADS, HUGE_WAVES = HUGE_THREADS / 64;
  __shared__ u32 wave_total[HUGE_WAVES];
  WORD held[PER_LANE];
  u32 occupied = 0;
#pragma unroll
  for(u32 k = 0; k < PER_LANE; k++) { held[k] = table[tid + k * HUGE_THREADS]; occupied += u32(held[k] != EMPTY); }
  u32 upto = occupied;                                        // inclusive scan over the wavefront
  upto = wave_scan_dpp(upto);
  if((tid & 63) == 63) { wave_total[tid >> 6] = upto; }
  __syncthreads();
  u32 put = upto - occupied;
  for(u32 w = 0; w < (tid >> 6); w++) { put += wave_total[w]; }
#pragma unroll
  for(u32 k = 0; k < PER_LANE; k++)
  {
    if(held[k] != EMPTY)
    {
      const unsigned long long v = (unsigned long long)held[k];
      values[b + put++] = v;
      mine = (v > mine ? v : mine);
    }
  }
  if(dead == nullptr)                                         // (the largest value fills the tail below; with the bitmap nobody asks)
  {
#pragma unroll
    for(u32 d = 32; d > 0; d >>= 1) { const unsigned long long other = __shfl_xor(mine, int(d)); mine = (other > mine ? other : mine); }
    if((tid & 63) == 0) { atomicMax(&largest, mine); }
  }
  __syncthreads();
  u32 count = distinct;
  unsigned long long top = largest;
  if(has_ones) { if(tid == 0) { values[b + count] = HUGE_EMPTY; } count++; top = HUGE_EMPTY; }
  // The rest of the segment holds nothing: with the bitmap `dead` (round 6; the one-sweep compaction reads it) its slots are
  // marked, a word of 64 slots at a time, and neither written here nor read there -- four raw values in five on a repeat-rich
  // graph; without it (the four-kernel compaction) they are filled with the largest value, duplicates the marks then drop.
  if(dead != nullptr)
  {
    const u64 first = b + count, end = b + len;                 // dead slots: [first, end)
    if(first < end)
    {
      const u64 w0 = first >> 6, w1 = (end - 1) >> 6;
      for(u64 w = w0 + tid; w <= w1; w += HUGE_THREADS)
      {
        unsigned long long bits = ~0ull;
        if(w == w0) { bits &= ~0ull << (first & 63); }
        if(w == w1 && (end & 63) != 0) { bits &= ~(~0ull << (end & 63)); }
        if(w == w0 || w == w1) { atomicOr(dead + w, bits); } else { dead[w] = bits; }      // (the edge words are shared with the neighbours)
      }
    }
  }
  else { for(u64 i = count + tid; i < len; i += HUGE_THREADS) { values[b + i] = top; } }
}

// one lane per query with 2..SMALL_SEGMENT values: bitonic network over registers, in place
__global__ __launch_bounds__(TPB) void k_sort_small(const u64* __restrict__ raw_off, u64 nq, u64* __restrict__ values,
                                                    unsigned long long* __restrict__ totals)
{
  u64 q = u64(blockIdx.x) * TPB + threadIdx.x;
  u64 b = 0, len = 0;
  if(q < nq) { b = raw_off[q]; len = raw_off[q + 1] - b; }
  u32 dups = 0;
  if(len >= 2 && len <= SMALL_SEGMENT)
  {
    u64 v[SMALL_SEGMENT];
#pragma unroll
    for(u32 i = 0; i < SMALL_SEGMENT; i++) { v[i] = (i < len ? values[b + i] : ~u64(0)); }
#pragma unroll
    for(u32 k = 2; k <= SMALL_SEGMENT; k <<= 1)
    {
#pragma unroll
      for(u32 j = k >> 1; j > 0; j >>= 1)
      {
#pragma unroll
        for(u32 i = 0; i < SMALL_SEGMENT; i++)
        {
          const u32 l = i ^ j;
          if(l > i)
          {
            const bool up = ((i & k) == 0);
            const u64 lo = (v[i] < v[l] ? v[i] : v[l]), hi = (v[i] < v[l] ? v[l] : v[i]);
            v[i] = (up ? lo : hi); v[l] = (up ? hi : lo);
          }
        }
      }
    }
    // padding (all ones) sorts to the end; a real value of all ones is then still within the first len slots
#pragma unroll
    for(u32 i = 0; i < SMALL_SEGMENT; i++) { if(i < len) { values[b + i] = v[i]; dups += u32(i > 0 && v[i] == v[i - 1]); } }
  }
  if(__ballot(dups != 0) != 0) { report_dups(totals, 1u, threadIdx.x & 63); }      // (uniform)
}

// ---- countKMers frontier expansion (src/algorithms.cpp:364-421) -------------------------------
// One lane per search state (a non-empty range at depth d): its children are LF_fast / LF_all of
// the range (src/gcsa.cpp:742-798) for comps 1..limit; non-empty children are appended to `out`
// (wave-aggregated atomic slot allocation) or, when out == nullptr, only counted.
constexpr int KMER_CHUNK = 4;       // comps per batch of independent block loads (the fast characters)

__global__ __launch_bounds__(TPB) void k_kmer_expand(DevImage img, const u64* __restrict__ in, u64 n_in, u32 limit,
                                                     u64* __restrict__ out, unsigned long long* __restrict__ counter)
{
  __shared__ WgSlots slots;
  u64 q = u64(blockIdx.x) * TPB + threadIdx.x;
  const u32 lane = threadIdx.x & 63;
  const u64 below = (u64(1) << lane) - 1;
  bool live = q < n_in;
  ulonglong2 r = live ? reinterpret_cast<const ulonglong2*>(in)[q] : make_ulonglong2(1, 0);
  for(u32 c0 = 1; c0 <= limit; c0 += KMER_CHUNK)
  {
    u64 csp[KMER_CHUNK], cep[KMER_CHUNK], mask[KMER_CHUNK];
    lf_children<KMER_CHUNK>(img, c0, limit, live, r.x, r.y, csp, cep);
    u32 wave_count = 0;
#pragma unroll
    for(int j = 0; j < KMER_CHUNK; j++)
    {
      mask[j] = __ballot(live && c0 + u32(j) <= limit && !range_empty(csp[j], cep[j]));
      wave_count += u32(__popcll(mask[j]));
    }
    u64 slot = wg_reserve(slots, counter, wave_count);
#pragma unroll
    for(int j = 0; j < KMER_CHUNK; j++)
    {
      if(out != nullptr && ((mask[j] >> lane) & 1))
      {
        reinterpret_cast<ulonglong2*>(out)[slot + __popcll(mask[j] & below)] = make_ulonglong2(csp[j], cep[j]);
      }
      slot += __popcll(mask[j]);
    }
  }
}

// ---- compareKMers frontier expansion (src/algorithms.cpp:505-616) ------------------------------
// A state is a pair of ranges, one per index; a child survives if it is non-empty in at least one
// index.  The two images are read through pointers (two DevImage values would not fit the 4 KB
// kernel-argument segment).  final != 0: classify the children instead of storing them:
// counters[1] += shared, counters[2] += left only, counters[3] += right only.
// KMerComparisonState::set (algorithms.cpp:451-457): comp of extension step i at bits [3i, 3i + 3).
// (The reference also evaluates `comp >> (64 - bit)` for bit == 0, a shift by the word size; the intended
// "|= 0 unless the comp straddles two words" is what is restated here.)
__device__ __forceinline__ void kmer_set(u64* kmer, u32 i, u64 comp)
{
  u32 offset = (i * 3) >> 6, bit = (i * 3) & 63;
  kmer[offset] |= comp << bit;
  if(bit > 61) { kmer[offset + 1] |= comp >> (64 - bit); }
}

// mode 0: expand (children appended to out / out_keys, counters[0] = number of children)
// mode 1: classify the children: counters[1] += shared, counters[2] += left only, counters[3] += right only
// mode 2: like 1, and the left-only / right-only children are written as 8-u64 KMerComparisonState
//         records (left range, right range, k, kmer[3]) to left_records / right_records
__global__ __launch_bounds__(TPB) void k_kmer_compare(const DevImage* __restrict__ left, const DevImage* __restrict__ right,
                                                      const u64* __restrict__ in, const u64* __restrict__ in_keys, u64 n_in,
                                                      u32 limit, u32 depth, int mode,
                                                      u64* __restrict__ out, u64* __restrict__ out_keys,
                                                      unsigned long long* __restrict__ counters,
                                                      u64* __restrict__ left_records, u64* __restrict__ right_records)
{
  __shared__ WgSlots slots;
  u64 q = u64(blockIdx.x) * TPB + threadIdx.x;
  const u32 lane = threadIdx.x & 63;
  const u64 below = (u64(1) << lane) - 1;
  bool live = q < n_in;
  ulonglong2 l = make_ulonglong2(1, 0), r = make_ulonglong2(1, 0);
  u64 key[3] = {0, 0, 0};
  if(live)
  {
    l = reinterpret_cast<const ulonglong2*>(in)[2 * q]; r = reinterpret_cast<const ulonglong2*>(in)[2 * q + 1];
    if(in_keys != nullptr) { key[0] = in_keys[3 * q]; key[1] = in_keys[3 * q + 1]; key[2] = in_keys[3 * q + 2]; }
  }
  unsigned long long shared_total = 0;          // mode 1 / 2: k-mers in both indexes, per wave
  for(u32 c0 = 1; c0 <= limit; c0 += KMER_CHUNK)
  {
    u64 lcsp[KMER_CHUNK], lcep[KMER_CHUNK], rcsp[KMER_CHUNK], rcep[KMER_CHUNK];
    lf_children<KMER_CHUNK>(*left, c0, limit, live, l.x, l.y, lcsp, lcep);
    lf_children<KMER_CHUNK>(*right, c0, limit, live, r.x, r.y, rcsp, rcep);
    u64 lmask[KMER_CHUNK], rmask[KMER_CHUNK];      // children that are non-empty on the left / right
    u32 any_count = 0, lonly_count = 0, ronly_count = 0;
#pragma unroll
    for(int j = 0; j < KMER_CHUNK; j++)
    {
      const bool active = live && c0 + u32(j) <= limit;
      lmask[j] = __ballot(active && !range_empty(lcsp[j], lcep[j]));
      rmask[j] = __ballot(active && !range_empty(rcsp[j], rcep[j]));
      any_count += u32(__popcll(lmask[j] | rmask[j]));
      lonly_count += u32(__popcll(lmask[j] & ~rmask[j]));
      ronly_count += u32(__popcll(rmask[j] & ~lmask[j]));
      shared_total += (unsigned long long)__popcll(lmask[j] & rmask[j]);
    }
    if(mode == 0)
    {
      u64 slot = wg_reserve(slots, counters, any_count);
#pragma unroll
      for(int j = 0; j < KMER_CHUNK; j++)
      {
        const u64 mask = lmask[j] | rmask[j];
        if(out != nullptr && ((mask >> lane) & 1))
        {
          const u64 at = slot + __popcll(mask & below);
          reinterpret_cast<ulonglong2*>(out)[2 * at] = make_ulonglong2(lcsp[j], lcep[j]);
          reinterpret_cast<ulonglong2*>(out)[2 * at + 1] = make_ulonglong2(rcsp[j], rcep[j]);
          if(out_keys != nullptr)
          {
            u64 child[3] = {key[0], key[1], key[2]};
            kmer_set(child, depth, c0 + u32(j));
            out_keys[3 * at] = child[0]; out_keys[3 * at + 1] = child[1]; out_keys[3 * at + 2] = child[2];
          }
        }
        slot += __popcll(mask);
      }
      continue;
    }
    u64 lslot = wg_reserve(slots, counters + 2, lonly_count);
    u64 rslot = wg_reserve(slots, counters + 3, ronly_count);
    if(mode == 2)
    {
#pragma unroll
      for(int j = 0; j < KMER_CHUNK; j++)
      {
        const u64 lonly = lmask[j] & ~rmask[j], ronly = rmask[j] & ~lmask[j];
        const bool mine_left = (lonly >> lane) & 1, mine_right = (ronly >> lane) & 1;
        if(mine_left || mine_right)
        {
          u64 child[3] = {key[0], key[1], key[2]};
          kmer_set(child, depth, c0 + u32(j));
          u64* rec = (mine_left ? left_records + 8 * (lslot + __popcll(lonly & below)) : right_records + 8 * (rslot + __popcll(ronly & below)));
          rec[0] = lcsp[j]; rec[1] = lcep[j]; rec[2] = rcsp[j]; rec[3] = rcep[j]; rec[4] = u64(depth) + 1;
          rec[5] = child[0]; rec[6] = child[1]; rec[7] = child[2];
        }
        lslot += __popcll(lonly); rslot += __popcll(ronly);
      }
    }
  }
  if(mode != 0) { wg_reserve(slots, counters + 1, u32(shared_total)); }
}

// ---- locate ------------------------------------------------------------------------------

// per query: number of path nodes to walk and number of values before deduplication
// (also clears entry nq of both arrays -- the scans turn it into the totals; the totals slot is cleared by the host)
// fuse_above != 0 (sorted mode with the locate table): a range of more than fuse_above path nodes, each with ONE value, is listed
// as a candidate for the split sort that reads the table itself (k_classify_fused decides).
__global__ __launch_bounds__(TPB) void k_locate_sizes(DevImage img, const u64* __restrict__ ranges, u64 nq,
                                                      u64* __restrict__ node_counts, u64* __restrict__ raw_counts,
                                                      unsigned long long* __restrict__ totals, u64 fuse_above, u64* __restrict__ candidates)
{
  __shared__ WgSlots slots;
  u64 q = u64(blockIdx.x) * TPB + threadIdx.x;
  const u32 lane = threadIdx.x & 63;
  if(q == 0) { node_counts[nq] = 0; raw_counts[nq] = 0; }
  u64 nodes = 0, raw = 0;
  if(q < nq)
  {
    ulonglong2 r = reinterpret_cast<const ulonglong2*>(ranges)[q];
    if(!(range_empty(r.x, r.y) || r.y >= img.n))              // gcsa.cpp:831
    {
      nodes = r.y + 1 - r.x;
      raw = nodes + sada_sparse_count(img, r.x, r.y);         // sum of |values(i)| = sum of (A[i] + 1)
    }
    node_counts[q] = nodes; raw_counts[q] = raw;
  }
  if(fuse_above == 0) { return; }                             // (uniform)
  const u64 listed = __ballot(nodes > fuse_above && raw == nodes);
  if(__syncthreads_or(listed != 0) == 0) { return; }          // (no candidate in the workgroup: the common case)
  u64 slot = wg_reserve(slots, totals + T_CAND, u32(__popcll(listed)));
  if((listed >> lane) & 1) { candidates[slot + __popcll(listed & ((u64(1) << lane) - 1))] = q; }
}

// One wavefront per candidate (a stride loop over the list): CLASSIFY_SAMPLE table entries of the range at equal distances,
// sorted in registers.  All of them direct values and no two equal -> the range almost certainly has more than BIG_SEGMENT
// distinct values (the argument of k_dedup_huge's sample) and is FUSED: its path nodes are taken out of the table pass
// (node_counts[q] = 0; its slots in the raw order stay), k_collect_multi lists it for k_over_split with the table as the source,
// and its values are read once, by the workgroup that splits them -- not written by the table pass and read back (round 5: the
// 4.4 GB written and 4.5 GB read again on the 16-mer batch of the 2^30-base text).  A wrong guess costs time, never a result:
// the split sort takes any segment.
constexpr u32 CLASSIFY_SAMPLE = 256;      // (512: 0.15 ms for the 13 899 candidates of the 16-mer batch on the 2^30-base text -- the sort of the sample)
__global__ __launch_bounds__(64) void k_classify_fused(DevImage img, const u64* __restrict__ ranges, const u64* __restrict__ candidates,
                                                       const unsigned long long* __restrict__ totals, u64* __restrict__ node_counts)
{
  const u32 lane = threadIdx.x;
  const u64 count = totals[T_CAND];
  for(u64 i = blockIdx.x; i < count; i += gridDim.x)          // (uniform)
  {
    const u64 q = candidates[i];
    const u64 sp = ranges[2 * q], nodes = ranges[2 * q + 1] + 1 - sp;
    const u32 sample = u32(nodes < CLASSIFY_SAMPLE ? nodes : CLASSIFY_SAMPLE);
    constexpr u32 R = CLASSIFY_SAMPLE / 64;
    u64 v[R];
    bool direct = true;
#pragma unroll
    for(u32 r = 0; r < R; r++)
    {
      const u32 k = r * 64 + lane;
      v[r] = ~u64(0);
      if(k < sample)
      {
        const u64 entry = img.locate_tab[sp + (u64(k) * nodes) / sample];
        direct = direct && (entry & LOCATE_DIRECT) != 0;
        v[r] = entry & ~LOCATE_DIRECT;
      }
    }
    wave_sort_regs<R>(v);
    const u32 dups = dups_in_regs<R>(v, sample, lane);
    if(__ballot(!direct) == 0 && dups == 0 && lane == 0) { node_counts[q] = 0; }
  }
}

// The whole of locate() for a batch in which EVERY range is one path node with one value that the locate table holds directly
// (config 3's 32-mers on a chr22-like index, the final ranges of config 5's long patterns): value q = the table entry of node
// sp_q, offsets[q] = q -- one kernel, one random 8-byte read per range, instead of the sizes pass, two prefix sums, the list
// builder, the owner search and the walk.  Any other range (empty, wider, a node with several samples) sets `misfit` and the
// caller runs the general pipeline, which rewrites everything written here.
__global__ __launch_bounds__(TPB) void k_locate_single(DevImage img, const u64* __restrict__ ranges, u64 nq, u64* __restrict__ offsets,
                                                       u64* __restrict__ values, u64 capacity, unsigned long long* __restrict__ misfit)
{
  const u64 q = u64(blockIdx.x) * TPB + threadIdx.x;
  if(q > nq) { return; }
  offsets[q] = q;
  bool ok = true;
  if(q < nq)
  {
    const ulonglong2 r = reinterpret_cast<const ulonglong2*>(ranges)[q];
    ok = (r.x == r.y && r.x < img.n);
    const u64 entry = (ok ? img.locate_tab[r.x] : 0);
    ok = ok && (entry & LOCATE_DIRECT) != 0;
    if(ok && q < capacity) { values[q] = entry & ~LOCATE_DIRECT; }
  }
  const u64 bad = __ballot(!ok);
  if(bad != 0 && (threadIdx.x & 63) == u32(__ffsll((long long)bad)) - 1) { atomicOr(misfit, 1ull); }
}

// one lane per (query, path node): locateInternal (gcsa.cpp:880-896)
__global__ __launch_bounds__(TPB) void k_locate_walk(DevImage img, const u64* __restrict__ ranges, u64 nq,
                                                     const u64* __restrict__ node_off, const u64* __restrict__ raw_off,
                                                     u64 total_nodes, u64* __restrict__ values)
{
  __shared__ Tables t;
  stage_tables(img, t);
  u64 g = u64(blockIdx.x) * TPB + threadIdx.x;
  if(g >= total_nodes) { return; }
  // query owning flattened node g: last q with node_off[q] <= g
  u64 lo = 0, hi = nq - 1;
  while(lo < hi)
  {
    u64 mid = (lo + hi + 1) >> 1;
    if(node_off[mid] <= g) { lo = mid; } else { hi = mid - 1; }
  }
  u64 sp = ranges[2 * lo];
  u64 node = sp + (g - node_off[lo]);
  u64 dest = raw_off[lo] + (node - sp) + (node > sp ? sada_sparse_count(img, sp, node - 1) : 0);

  u64 steps = 0, srank;
  while(!bv_get_rank(img.sampled, node, srank))             // gcsa.cpp:883-887
  {
    node = lf_node(img, t.C, node); steps++;
  }
  u64 s = (srank > 0 ? bv_select(img.samples, srank) + 1 : 0);   // firstSample, gcsa.h:202-206
  do
  {
    values[dest++] = packed_get(img.stored, img.sample_width, s) + steps;   // gcsa.cpp:893
    s++;
  }
  while(!bv_get(img.samples, s - 1));                        // lastSample, gcsa.h:208
}

// Segments with more than BIG_SEGMENT distinct values, round 5: ONE WORKGROUP PER SEGMENT sorts it.  It splits the segment on
// the top bits of (value - the segment's smallest value) into up to 4096 buckets of a few dozen values -- minimum and
// maximum, a histogram in LDS, its prefix sums, the scatter into `scratch` at the same offsets -- and then its sixteen
// wavefronts sort the buckets, one bucket of up to 64 values per wavefront at a time, in REGISTERS (a bitonic network over the
// lanes: 21 exchange steps, no LDS, no barrier; 32-bit keys relative to the run's base where they fit, wave_sort32) straight into
// `values`.  A segment is read three times and written twice by the workgroup that owns it, the values are contiguous per segment already, and nothing is sorted across
// segments: round 4 packed (segment rank << 37 | value) keys and gave them to the library's device-wide radix sort, seven
// passes over 51-bit keys of which 14 bits said what the layout already knew (19 of the 35 ms of the 16-mer batch on the
// 2^30-base text, profiles/r04_locate.md; a first form of this kernel that left buckets of ~1500 values to the workgroup
// bitonic sort took 22 ms for them: 66 barriers per bucket, profiles/r05_locate.md).  A bucket of more than 64 values is
// listed for the workgroup sort (k_sort_big reads `scratch`, writes `values`), one of more than `skew_above` values -- values
// crowded into a small part of the segment's span -- goes on the `skew` list, which the host hands to that radix sort as before.
#ifndef GCSA2_SPLIT_THREADS
#define GCSA2_SPLIT_THREADS 1024
#endif
constexpr int SPLIT_THREADS = GCSA2_SPLIT_THREADS;
constexpr u32 SPLIT_BUCKETS_UNTILED = 4096;
// Round 6, TILED: the scatter goes through LDS a tile of SPLIT_TILE values at a time.  Untiled, the 64 lanes of a store
// instruction hit ~50 different buckets: 64 eight-byte write requests where a copy sends a few lines.  The L2 merges them, but
// it takes REQUESTS at a fixed rate: with the stores switched off the kernel took 1.7 of its 4.4 ms on the 16-mer batch of the
// 2^30-base text (profiles/r06_locate.md; the knock-out knob is profiles/r06_locate/split_debug.patch).  Tiled, the workgroup counts the tile's values per bucket (the
// LDS atomics that also rank a value inside its bucket), every wavefront scans the counts for itself (eight per lane, two
// 16-byte reads, twice: the lane's total, then -- after the scan over the lanes -- the offsets; all wavefronts write the same
// offsets, so no barrier), the values are placed bucket by bucket in an LDS buffer and written out in that order: neighbouring
// lanes hold neighbouring values of one bucket, SPLIT_TILE / buckets of them in a row.  Three barriers per tile; the next
// tile's values are requested before the current one is worked on.  At most SPLIT_TILED_BUCKETS = 512 buckets, whatever the
// segment's length: a segment beyond 131 072 values gets longer buckets -- up to 512 values a wavefront sorts in eight
// registers per lane, up to 1024 in sixteen (lists of their own), beyond that the workgroup sort.  (The series, 16-mer /
// 32-mer batch, profiles/r06_locate.md: 256 buckets 6.84 / 6.34 ms, 512 buckets 6.82 / 6.04, 1024 buckets 7.72 / 6.79 -- every
// wavefront reads and writes 8 KB of counts per tile --; counts held in registers across the scan cost the second workgroup
// per CU; workgroups of 512 threads 6.99 / 5.25; tiles only for segments of up to 256 buckets 7.43 / 6.58.)
#ifndef GCSA2_TILED_BUCKETS
#define GCSA2_TILED_BUCKETS 512
#endif
constexpr u32 SPLIT_TILED_BUCKETS = GCSA2_TILED_BUCKETS;
constexpr u32 SPLIT_TILE_PER = 4, SPLIT_TILE = u32(SPLIT_THREADS) * SPLIT_TILE_PER;
constexpr u32 SPLIT_SAMPLE = 8192;             // values whose minimum and maximum stand for the segment's
constexpr u32 SPLIT_AHEAD = 4;                 // independent loads per lane in the streaming passes
constexpr u32 SPLIT_RUNS_AHEAD = 3;           // runs of buckets whose values a wavefront has requested ahead of the one it sorts
constexpr u32 SPLIT_CHUNK = 32;                // buckets a wavefront draws at a time in the run phase
// Values per bucket aimed at.  24 until the sorts of the listed buckets moved into registers (k_sort_bucket, late in round 5):
// buckets a wavefront's RUN could take, because a listed bucket cost an LDS sort.  With a few hundred values per bucket the
// workgroup writes into ~150 streams per segment instead of 4096 -- the partly written lines of all workgroups stay in L2 --
// and nearly every bucket is listed: k_over_split 8.2 -> 4.2 ms, k_sort_bucket 0.4 -> 2.7 ms on the 16-mer batch of the
// 2^30-base text (flat from 192 to 512; profiles/r05_locate.md).
constexpr u32 SPLIT_TARGET = 256;
// The longest bucket one wavefront sorts (k_sort_bucket); a longer one goes to the workgroup sort.  1024 (16 registers of values
// per lane, 102 VGPRs: four wavefronts per SIMD) until round 6; with 512 the kernel needs half the registers and twice the
// wavefronts hide its loads.
constexpr u32 BUCKET_BY_WAVE = 512;

// (waves_per_eu: two workgroups per CU, said to the register allocator -- 4.55 -> 4.43 ms and 3.30 -> 3.03 ms on the two batches
// of the 2^30-base text)
template<bool TILED>
__global__ __launch_bounds__(SPLIT_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_over_split(const u64* __restrict__ over_begin, const u64* __restrict__ over_end,
                                                             u64* values, u64* scratch, u64* __restrict__ bkt_begin, u64* __restrict__ bkt_end,
                                                             u64* __restrict__ skew_begin, u64* __restrict__ skew_end,
                                                             unsigned long long* __restrict__ totals, u32 skew_above, u32 target, u64 bucket_last,
                                                             const u64* const* __restrict__ over_src,
                                                             u64* __restrict__ mid_begin, u64* __restrict__ mid_end)
{
  // (the tiled form never has more than SPLIT_TILED_BUCKETS buckets: its cursor / start arrays are as long as the workgroup)
  constexpr u32 SPLIT_BUCKETS = (TILED ? u32(SPLIT_THREADS) : SPLIT_BUCKETS_UNTILED);
  __shared__ __attribute__((aligned(16))) u32 cursor[SPLIT_BUCKETS];        // histogram, then the buckets' write cursors (= their ends after the scatter)
  __shared__ u32 starts[SPLIT_BUCKETS];
  __shared__ u32 wave_sums[SPLIT_THREADS / 64];
  __shared__ unsigned long long s_lo, s_hi, list_base, big_base, skew_base, mid_base;
  __shared__ u32 wg_listed, wg_big, wg_skewed, wg_skew_values, next_chunk, wg_mid;
  __shared__ __attribute__((aligned(16))) u32 tile_count[2][TILED ? SPLIT_TILED_BUCKETS : 4];      // values of the tile per bucket (two tiles alternate)
  __shared__ __attribute__((aligned(16))) u32 tile_off[TILED ? SPLIT_TILED_BUCKETS : 4];           // their exclusive prefix sums
  __shared__ __attribute__((aligned(16))) u32 tile_delta[TILED ? SPLIT_TILED_BUCKETS : 4];         // where in the segment the bucket's values of this tile go, minus tile_off
  __shared__ u64 tile_value[TILED ? SPLIT_TILE : 1];                  // the tile, bucket by bucket
  constexpr u32 PER_THREAD = SPLIT_BUCKETS / SPLIT_THREADS;
  const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const u64 b = over_begin[blockIdx.x], len = over_end[blockIdx.x] - b;
  // where the segment's unsorted values are: the raw values the table pass wrote, or -- a FUSED range (k_classify_fused) -- the
  // entries of the locate table for its path nodes, which lie side by side there (their flag bit is cleared as they are read)
  // (a pointer that was LOADED -- over_src[] -- is a generic one to the compiler: its loads were flat_load, which count as LDS
  // operations too, so every wait for the LDS behind a prefetch waited for the prefetch.  Said to be global memory here.)
  typedef const __attribute__((address_space(1))) u64* global_values;
  const global_values src = (global_values)(over_src[blockIdx.x] != nullptr ? over_src[blockIdx.x] : values + b);
  constexpr u64 KEEP = ~LOCATE_DIRECT;                        // (a raw value has the bit clear: k_build_locate_table)
  if(tid == 0) { s_lo = ~0ull; s_hi = 0; }
  for(u32 k = tid; k < SPLIT_BUCKETS; k += SPLIT_THREADS) { cursor[k] = 0; }
  __syncthreads();
  unsigned long long lo = ~0ull, hi = 0;
  // (every streaming pass keeps SPLIT_AHEAD independent loads per lane in flight: with one, a workgroup moved ~8 GB/s per CU)
  // (minimum and maximum of a SAMPLE -- the segment's first SPLIT_SAMPLE values, which arrive in path order, i.e. in no order of
  // value -- instead of a pass over the segment: the bucket of a value only has to be a monotone function of it, so what lies
  // outside the sample's span is clamped into the first or the last bucket, ~len / SPLIT_SAMPLE values each)
  for(u64 i0 = tid; i0 < len && i0 < SPLIT_SAMPLE; i0 += SPLIT_AHEAD * SPLIT_THREADS)
  {
    u64 got[SPLIT_AHEAD];
#pragma unroll
    for(u32 j = 0; j < SPLIT_AHEAD; j++) { const u64 i = i0 + u64(j) * SPLIT_THREADS; got[j] = (i < len ? src[i] : src[tid % len]) & KEEP; }
#pragma unroll
    for(u32 j = 0; j < SPLIT_AHEAD; j++) { lo = (got[j] < lo ? got[j] : lo); hi = (got[j] > hi ? got[j] : hi); }
  }
  for(int o = 32; o > 0; o >>= 1)
  {
    const unsigned long long a = __shfl_down(lo, o, 64), c = __shfl_down(hi, o, 64);
    lo = (a < lo ? a : lo); hi = (c > hi ? c : hi);
  }
  if(lane == 0) { atomicMin(&s_lo, lo); atomicMax(&s_hi, hi); }
  __syncthreads();
  lo = s_lo; hi = s_hi;
  // buckets: the smallest power of two with len / buckets <= SPLIT_TARGET, at most SPLIT_BUCKETS; bucket = (v - lo) >> shift
  u32 nb = 2;
  const u32 most_buckets = (TILED ? SPLIT_TILED_BUCKETS : SPLIT_BUCKETS);
  while(nb < most_buckets && u64(nb) * target < len) { nb <<= 1; }          // (target: SPLIT_TARGET; GCSA2_SPLIT_TARGET in tests)
  const u64 span = hi - lo;                                   // largest (v - lo)
  u32 shift = 0;
  while(shift < 63 && (span >> shift) >= nb) { shift++; }
  // (the buckets are a power of two wide, so the sample's span ends somewhere in the upper half of them: the ones behind it stay
  // empty and are dropped -- a value above the sample's largest goes into the last bucket the span reaches.  k_over_split 3.2 -> 3.0 ms
  // on the 16-mer batch of the 2^30-base text: the prefix sums, the lists and the runs walk over fewer buckets.)
  nb = u32(span >> shift) + 1;
  auto bucket_of = [&](u64 v) -> u32
  {
    const u64 raw = (v > lo ? (v - lo) >> shift : 0);
    return u32(raw < nb ? raw : nb - 1);
  };
  // (the histogram pass: the group of SPLIT_AHEAD values behind the one being counted is on its way -- clamped indices instead of
  // branches around the loads and the current values pinned, as in the tile loop below: the compiler's waits stay exact)
  {
    u64 ahead[SPLIT_AHEAD];
#pragma unroll
    for(u32 j = 0; j < SPLIT_AHEAD; j++) { const u64 i = u64(tid) + u64(j) * SPLIT_THREADS; ahead[j] = src[i < len ? i : len - 1]; }
    for(u64 i0 = tid; i0 < len; i0 += SPLIT_AHEAD * SPLIT_THREADS)
    {
      u64 got[SPLIT_AHEAD];
#pragma unroll
      for(u32 j = 0; j < SPLIT_AHEAD; j++) { got[j] = ahead[j] & KEEP; asm volatile("" : "+v"(got[j])); }
#pragma unroll
      for(u32 j = 0; j < SPLIT_AHEAD; j++) { const u64 i = i0 + u64(SPLIT_AHEAD + j) * SPLIT_THREADS; ahead[j] = src[i < len ? i : len - 1]; }
#pragma unroll
      for(u32 j = 0; j < SPLIT_AHEAD; j++) { if(i0 + u64(j) * SPLIT_THREADS < len) { atomicAdd(&cursor[bucket_of(got[j])], 1u); } }
    }
  }
  __syncthreads();
  // exclusive prefix sums of the counts: PER_THREAD consecutive buckets per thread, then across the wavefront and the workgroup
  u32 mine[PER_THREAD], sum = 0;
#pragma unroll
  for(u32 k = 0; k < PER_THREAD; k++) { mine[k] = cursor[tid * PER_THREAD + k]; sum += mine[k]; }
  const u32 incl = wave_scan_dpp(sum);
  if(lane == 63) { wave_sums[wave] = incl; }
  __syncthreads();
  u32 before = incl - sum;
  for(u32 w = 0; w < wave; w++) { before += wave_sums[w]; }
  const u32 small = (skew_above < 64 ? skew_above : 64u);     // a run, and a bucket that needs no list, holds at most this many values
  // the buckets that are too large for a run are listed here, by the threads that own them, in slots the WORKGROUP reserves with
  // one atomic per list (one atomic per bucket on the global counters -- a million of them on one address -- was 5 of the
  // kernel's 14 ms on the clustered segments of the 32-mer batch; profiles/r05_locate.md)
  // (two lists in the same arrays: the buckets one wavefront sorts, k_sort_bucket, from the front; the few beyond MEDIUM_SEGMENT
  // values, k_sort_big, from the back -- as one list, the workgroup sorts launched a million workgroups to find a few hundred)
  u32 my_listed = 0, my_big = 0, my_skewed = 0, my_skew_values = 0, my_mid = 0;
#pragma unroll
  for(u32 k = 0; k < PER_THREAD; k++)
  {
    starts[tid * PER_THREAD + k] = before; cursor[tid * PER_THREAD + k] = before; before += mine[k];
    if(mine[k] > small)
    {
      if(mine[k] > skew_above) { my_skewed++; my_skew_values += mine[k]; }
      else if(mine[k] > MEDIUM_SEGMENT) { my_big++; }
      else if(mine[k] > BUCKET_BY_WAVE) { my_mid++; }
      else { my_listed++; }
    }
  }
  if(tid == 0) { wg_listed = 0; wg_big = 0; wg_skewed = 0; wg_skew_values = 0; next_chunk = 0; wg_mid = 0; }
  __syncthreads();
  u32 listed_at = 0, big_at = 0, skewed_at = 0, mid_at = 0;
  if(my_listed > 0) { listed_at = atomicAdd(&wg_listed, my_listed); }
  if(my_big > 0) { big_at = atomicAdd(&wg_big, my_big); }
  if(my_mid > 0) { mid_at = atomicAdd(&wg_mid, my_mid); }
  if(my_skewed > 0) { skewed_at = atomicAdd(&wg_skewed, my_skewed); atomicAdd(&wg_skew_values, my_skew_values); }
  __syncthreads();
  if(tid == 0)
  {
    list_base = (wg_listed > 0 ? atomicAdd(totals + T_BUCKETS, (unsigned long long)wg_listed) : 0ull);
    big_base = (wg_big > 0 ? atomicAdd(totals + T_BIG_BUCKETS, (unsigned long long)wg_big) : 0ull);
    mid_base = (wg_mid > 0 ? atomicAdd(totals + T_MID_BUCKETS, (unsigned long long)wg_mid) : 0ull);
    skew_base = (wg_skewed > 0 ? atomicAdd(totals + T_SKEW, (unsigned long long)wg_skewed) : 0ull);
    if(wg_skewed > 0) { atomicAdd(totals + T_SKEW_VALUES, (unsigned long long)wg_skew_values); }
  }
  __syncthreads();
  {
    u32 at = starts[tid * PER_THREAD];
#pragma unroll
    for(u32 k = 0; k < PER_THREAD; k++)
    {
      if(mine[k] > small)
      {
        if(mine[k] > skew_above) { const u64 slot = skew_base + skewed_at++; skew_begin[slot] = b + at; skew_end[slot] = b + at + mine[k]; }
        else if(mine[k] > MEDIUM_SEGMENT) { const u64 slot = bucket_last - (big_base + big_at++); bkt_begin[slot] = b + at; bkt_end[slot] = b + at + mine[k]; }
        else if(mine[k] > BUCKET_BY_WAVE) { const u64 slot = mid_base + mid_at++; mid_begin[slot] = b + at; mid_end[slot] = b + at + mine[k]; }
        else { const u64 slot = list_base + listed_at++; bkt_begin[slot] = b + at; bkt_end[slot] = b + at + mine[k]; }
      }
      at += mine[k];
    }
  }
  if constexpr(TILED)
  {
    static_assert(SPLIT_TILED_BUCKETS % 256 == 0 && SPLIT_TILED_BUCKETS <= SPLIT_THREADS, "counts per lane in groups of four; a thread owns a bucket");
    if(tid < SPLIT_TILED_BUCKETS) { tile_count[0][tid] = 0; tile_count[1][tid] = 0; }
    __syncthreads();
    u64 next[SPLIT_TILE_PER];
#pragma unroll
    for(u32 j = 0; j < SPLIT_TILE_PER; j++) { const u64 i = u64(j) * SPLIT_THREADS + tid; next[j] = src[i < len ? i : len - 1]; }      // (no branch around a load: see below)
    u32 which = 0;
    for(u64 t0 = 0; t0 < len; t0 += SPLIT_TILE, which ^= 1u)
    {
      u32* __restrict__ count = tile_count[which];
      u64 got[SPLIT_TILE_PER];
      u32 where[SPLIT_TILE_PER];                              // bucket | rank inside the bucket's values of this tile << 16 (one register: the kernel has 64)
      static_assert(SPLIT_TILED_BUCKETS <= 65536 && SPLIT_TILE <= 65536, "two 16-bit halves");
#pragma unroll
      for(u32 j = 0; j < SPLIT_TILE_PER; j++) { got[j] = next[j] & KEEP; asm volatile("" : "+v"(got[j])); }
      // The next tile's values travel while this one is worked on.  They did not until late in round 6: (i) the loads were
      // flat_load (above); (ii) they sat behind `if(i < len)` branches, and the compiler cannot count memory operations across a
      // branch, so its wait for the PREVIOUS tile's values -- placed at their first use, behind these loads -- was
      // s_waitcnt vmcnt(0): the workgroup waited for the loads it had just issued.  Now the index is clamped instead (every lane
      // loads), and the previous values are pinned in their registers (the empty asm) BEFORE the new loads leave, so that the wait
      // lands there: on loads and stores a tile old.  One uniform base per tile and 32-bit lane offsets keep the addresses in
      // one scalar pair and one register (the kernel lives on 64 registers).
      const u64 next_base = (t0 + SPLIT_TILE < len ? t0 + SPLIT_TILE : len - 1);
      const global_values next_src = src + next_base;
      const u64 left = len - 1 - next_base;
      const u32 last = u32(left < SPLIT_TILE ? left : SPLIT_TILE);      // the last index of that tile that exists
#pragma unroll
      for(u32 j = 0; j < SPLIT_TILE_PER; j++)
      {
        const u32 i = j * SPLIT_THREADS + tid;
        next[j] = next_src[i < last ? i : last];
      }
#pragma unroll
      for(u32 j = 0; j < SPLIT_TILE_PER; j++)
      {
        where[j] = bucket_of(got[j]);
        if(t0 + u64(j) * SPLIT_THREADS + tid < len) { where[j] |= atomicAdd(&count[where[j]], 1u) << 16; }
      }
      __syncthreads();
      {
        // every wavefront scans the counts for itself: lane l has buckets PER l .. PER l + PER - 1 (16-byte reads); wavefront 0
        // also says where the buckets' values of this tile go (tile_delta)
        constexpr u32 PER = SPLIT_TILED_BUCKETS / 64;
        // (two passes over the lane's counts -- its total first, then, after the scan over the lanes, the counts again for the
        // offsets: holding them across the scan cost the kernel its second workgroup per CU once PER was 8)
        u32 sum = 0;
#pragma unroll
        for(u32 k = 0; k < PER; k += 4)
        {
          const uint4 part = *reinterpret_cast<const uint4*>(&count[lane * PER + k]);
          sum += part.x + part.y + part.z + part.w;
        }
        const u32 incl = wave_scan_dpp(sum);                    // (no trip through the LDS crossbar: six __shfl_up were six waits of every wavefront, once per tile)
        u32 at = incl - sum;
        asm volatile("" ::: "memory");                          // (the counts are read again, not kept)
#pragma unroll
        for(u32 k = 0; k < PER; k += 4)
        {
          const uint4 part = *reinterpret_cast<const uint4*>(&count[lane * PER + k]);
          const uint4 off = make_uint4(at, at + part.x, at + part.x + part.y, at + part.x + part.y + part.z);
          *reinterpret_cast<uint4*>(&tile_off[lane * PER + k]) = off;
          if(wave == 0)
          {
            const uint4 cur = *reinterpret_cast<const uint4*>(&cursor[lane * PER + k]);
            *reinterpret_cast<uint4*>(&tile_delta[lane * PER + k]) = make_uint4(cur.x - off.x, cur.y - off.y, cur.z - off.z, cur.w - off.w);
          }
          at += part.x + part.y + part.z + part.w;
        }
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for(u32 j = 0; j < SPLIT_TILE_PER; j++)
      {
        if(t0 + u64(j) * SPLIT_THREADS + tid < len)
        {
          const u32 at = tile_off[where[j] & 0xFFFFu] + (where[j] >> 16);
          tile_value[at] = got[j];
        }
      }
      __syncthreads();
      const u32 here = u32(len - t0 < SPLIT_TILE ? len - t0 : SPLIT_TILE);
#pragma unroll
      for(u32 j = 0; j < SPLIT_TILE_PER; j++)
      {
        const u32 at = j * SPLIT_THREADS + tid;
        if(at < here) { const u64 v = tile_value[at]; scratch[b + u32(tile_delta[bucket_of(v)] + at)] = v; }      // (the bucket again from the value: three instructions instead of a 16-bit LDS array written and read)
      }
      __syncthreads();
      // the owners move the buckets' cursors on and clear this tile's counts (the next tile counts in the other array)
      // (The barrier above is not needed for correctness -- whatever the next tile overwrites, it overwrites behind ITS first
      // barrier, which no wavefront passes before all have stored this tile -- and was taken out once: 2.79 -> 2.93 ms on the 16-mer
      // batch of the 2^30-base text, same box, three runs each.  Wavefronts that run ahead into the next tile's atomics take the
      // LDS from the ones still reading this tile.)
      if(tid < SPLIT_TILED_BUCKETS) { cursor[tid] += count[tid]; count[tid] = 0; }
    }
  }
  else
  for(u64 i0 = tid; i0 < len; i0 += SPLIT_AHEAD * SPLIT_THREADS)
  {
    u64 got[SPLIT_AHEAD];
#pragma unroll
    for(u32 j = 0; j < SPLIT_AHEAD; j++) { const u64 i = i0 + u64(j) * SPLIT_THREADS; got[j] = (i < len ? (src[i] & KEEP) : lo); }
#pragma unroll
    for(u32 j = 0; j < SPLIT_AHEAD; j++)
    {
      if(i0 + u64(j) * SPLIT_THREADS < len)
      {
        scratch[b + atomicAdd(&cursor[bucket_of(got[j])], 1u)] = got[j];
      }
    }
  }
  __syncthreads();                                             // (the workgroup's stores have completed: s_waitcnt vmcnt(0) + barrier)
  // The wavefronts sort the buckets in runs of WHOLE buckets that hold at most 64 values together (the buckets are
  // value-ordered, so a run sorted by value is final): one value per lane, 21 exchange steps.  A wavefront draws CHUNKS of
  // SPLIT_CHUNK consecutive buckets from a counter of the workgroup.
  u32 k = 0, k_end = 0;
  // the next run of this wavefront: [first, first + count) of the segment; false when the segment's buckets are used up
  // (`base`: every value of the run lies in [base, base + 2^32 - 1), or NO_BASE: the first and the last bucket take what lies
  // outside the sample's span, and a wide segment's buckets may be wider than that)
  constexpr u64 NO_BASE = ~u64(0);
  auto next_run = [&](u32& first, u32& count, u64& base) -> bool
  {
    while(true)
    {
      if(k >= k_end)
      {
        u32 got = 0;
        if(lane == 0) { got = atomicAdd(&next_chunk, SPLIT_CHUNK); }
        got = __builtin_amdgcn_readfirstlane(got);
        if(got >= nb) { return false; }
        k = got; k_end = (got + SPLIT_CHUNK < nb ? got + SPLIT_CHUNK : nb);
      }
      first = starts[k];
      // ends of the next 64 buckets (lane j: bucket k + j); the run ends behind the last one that keeps it within `small` values
      const u32 kk = k + lane;
      const u32 end = (kk < k_end ? cursor[kk] : ~u32(0));
      const u64 fits = __ballot(kk < k_end && end - first <= small);        // (a prefix of the lanes: ends grow)
      const u32 take = u32(__popcll(fits));
      if(take == 0)
      {
        // bucket k alone has more than `small` values: listed above for the sorts of the next launches; one that is too large
        // even for those (BIG_SEGMENT; lower in tests, GCSA2_SPLIT_SKEW) goes back to where the radix sort expects it
        const u32 big = cursor[k] - first;
        if(big > skew_above) { for(u32 i = first + lane; i < first + big; i += 64) { values[b + i] = scratch[b + i]; } }
        k++;
        continue;
      }
      count = cursor[k + take - 1] - first;                    // (uniform)
      base = (k > 0 && k + take < nb && shift < 32 && (u64(take) << shift) < 0xFFFFFFFFull ? lo + (u64(k) << shift) : NO_BASE);
      k += take;
      if(count > 0) { return true; }
    }
  };
  // SPLIT_RUNS_AHEAD runs are in flight: a run is sorted while the loads of the ones behind it travel
  static_assert(SPLIT_RUNS_AHEAD >= 2, "the pipeline below shifts at least two stages");
  bool live[SPLIT_RUNS_AHEAD];
  u32 run_first[SPLIT_RUNS_AHEAD], run_count[SPLIT_RUNS_AHEAD];
  u64 run_value[SPLIT_RUNS_AHEAD], run_base[SPLIT_RUNS_AHEAD];
#pragma unroll
  for(u32 d = 0; d < SPLIT_RUNS_AHEAD; d++)
  {
    run_first[d] = 0; run_count[d] = 0; run_base[d] = NO_BASE;
    live[d] = next_run(run_first[d], run_count[d], run_base[d]);
    run_value[d] = (live[d] && lane < run_count[d] ? scratch[b + run_first[d] + lane] : ~u64(0));
  }
  u32 run_dups = 0;
  while(live[0])
  {
    u64 v = run_value[0];
    if(run_count[0] > 1)
    {
      if(run_base[0] != NO_BASE)                               // (uniform)
      {
        const u32 key = wave_sort32(lane < run_count[0] ? u32(v - run_base[0]) : ~u32(0), lane);
        v = run_base[0] + key;
      }
      else { v = wave_sort(v, lane); }
      const u64 left = wave_shr1(v);
      run_dups += u32(__popcll(__ballot(lane > 0 && lane < run_count[0] && v == left)));      // (runs are whole buckets: no value spans two)
    }
    if(lane < run_count[0]) { values[b + run_first[0] + lane] = v; }
#pragma unroll
    for(u32 d = 0; d + 1 < SPLIT_RUNS_AHEAD; d++)
    {
      live[d] = live[d + 1]; run_first[d] = run_first[d + 1]; run_count[d] = run_count[d + 1]; run_value[d] = run_value[d + 1]; run_base[d] = run_base[d + 1];
    }
    constexpr u32 last = SPLIT_RUNS_AHEAD - 1;
    live[last] = live[last - 1] && next_run(run_first[last], run_count[last], run_base[last]);
    run_value[last] = (live[last] && lane < run_count[last] ? scratch[b + run_first[last] + lane] : ~u64(0));
  }
  report_dups(totals, run_dups, lane);
}

// one wavefront (= one workgroup) per listed bucket of up to MEDIUM_SEGMENT values: bitonic sort in registers, read from
// `source`, written to `values` (k_sort_medium's network; the list is k_over_split's: buckets of 65 .. skew_above values, the
// longer ones are left to k_sort_big)
// (MOST = BUCKET_BY_WAVE: the list of the buckets up to 512 values, 57 VGPRs; MOST = MEDIUM_SEGMENT: the list of the ones from 513
// to 1024 values -- sixteen registers of values per lane, 102 VGPRs -- which went to the workgroup sort for a while in round 6:
// 0.6 ms against 0.2 on the 16-mer batch of the 2^30-base text)
// (one wavefront = one workgroup per bucket, 2.7 M workgroups per call of the 16-mer batch of the 2^30-base text: four or eight
// buckets per workgroup, a wavefront each, were measured at the end of round 6 -- 1.94 ms against 1.85 on one box; the dispatcher is not it)
template<u32 MOST>
__global__ __launch_bounds__(64) void k_sort_bucket(const u64* __restrict__ bkt_begin, const u64* __restrict__ bkt_end, u64* values,
                                                    const u64* __restrict__ source, unsigned long long* __restrict__ totals)
{
  if(blockIdx.x >= totals[MOST == BUCKET_BY_WAVE ? T_BUCKETS : T_MID_BUCKETS]) { return; }
  const u32 lane = threadIdx.x;
  const u64 b = bkt_begin[blockIdx.x];
  const u32 len = u32(bkt_end[blockIdx.x] - b);
  if(len > MOST) { return; }                                  // (cannot be: k_over_split fills the lists by length)
  __shared__ u32 stage[MOST];                                 // (the transposition of the blocked 32-bit network)
  report_dups(totals, sort_segment_by_wave<MOST>(source + b, values + b, len, lane, ~u64(0), stage), lane);
}

// Segments with more than BIG_SEGMENT distinct values whose split left a bucket too large (k_over_split's skew list), and every
// such segment with GCSA2_LOCATE_SPLIT_SORT=0 (A/B) (a 16-mer of an interspersed repeat matches 200 000 path nodes on the
// repeat-rich 2^30-base text): ONE device-wide radix sort over keys (rank of the segment among those segments) << value_bits
// | value sorts them all at once, whatever their sizes -- round 3 gave them to the library's SEGMENTED sort, whose work per
// segment made a batch of 16 000 such segments 100 ms.  over_off = exclusive scan of the segment lengths (over + 1 entries).
__global__ __launch_bounds__(TPB) void k_over_lengths(const u64* __restrict__ over_begin, const u64* __restrict__ over_end, u64 over,
                                                      u64* __restrict__ lengths)
{
  const u64 s = u64(blockIdx.x) * TPB + threadIdx.x;
  if(s <= over) { lengths[s] = (s < over ? over_end[s] - over_begin[s] : 0); }
}

// (the segment of a wavefront's first value is searched once, through the scalar cache; its other lanes step on from there --
// a segment has thousands of values, so almost always not at all.  A search per value was 14 dependent loads for each.)
__global__ __launch_bounds__(TPB) void k_over_pack(const u64* __restrict__ over_begin, const u64* __restrict__ over_off, u64 over, u64 total,
                                                   const u64* __restrict__ values, u32 value_bits, u64* __restrict__ keys)
{
  const u64 i = u64(blockIdx.x) * TPB + threadIdx.x;
  const u64 first = __builtin_amdgcn_readfirstlane(u32(i >> 32)) * (u64(1) << 32) + __builtin_amdgcn_readfirstlane(u32(i));
  if(first >= total) { return; }                              // uniform
  u64 lo = 0, hi = over - 1;                                  // last segment with over_off[s] <= first
  while(lo < hi)
  {
    const u64 mid = (lo + hi + 1) >> 1;
    if(over_off[mid] <= first) { lo = mid; } else { hi = mid - 1; }
  }
  if(i >= total) { return; }
  u64 s = lo;
  while(s + 1 < over && over_off[s + 1] <= i) { s++; }
  keys[i] = (s << value_bits) | values[over_begin[s] + (i - over_off[s])];
}

__global__ __launch_bounds__(TPB) void k_over_unpack(const u64* __restrict__ over_begin, const u64* __restrict__ over_off, u64 total,
                                                     const u64* __restrict__ keys, u32 value_bits, u64* __restrict__ values)
{
  const u64 i = u64(blockIdx.x) * TPB + threadIdx.x;
  if(i >= total) { return; }
  const u64 key = keys[i], s = key >> value_bits;
  values[over_begin[s] + (i - over_off[s])] = key & ((u64(1) << value_bits) - 1);
}

// (GCSA2_DEDUP_HUGE=0, an A/B knob: the second huge list becomes the list of the radix sort as it is)
__global__ __launch_bounds__(TPB) void k_huge_to_over(const u64* __restrict__ huge_begin, const u64* __restrict__ huge_end, u64 last, u64 count,
                                                      u64* __restrict__ over_begin, u64* __restrict__ over_end, unsigned long long* __restrict__ totals)
{
  const u64 i = u64(blockIdx.x) * TPB + threadIdx.x;
  if(i >= count) { return; }
  const u64 b = huge_begin[last - i], e = huge_end[last - i];
  over_begin[i] = b; over_end[i] = e;
  atomicAdd(totals + T_OVER_VALUES, (unsigned long long)(e - b));
  if(i == 0) { totals[T_OVER] = count; }
}

// The totals of a pass, copied to page-locked host memory the host polls (h[TOTAL_WORDS - 1] = ticket, written last): a
// read-back for 10 us instead of the 100-400 us of hipMemcpyAsync into pageable memory + hipStreamSynchronize.
__global__ __launch_bounds__(64) void k_publish_totals(const unsigned long long* __restrict__ totals, volatile unsigned long long* h, unsigned long long ticket)
{
  if(threadIdx.x < TOTAL_WORDS - 1) { h[threadIdx.x] = totals[threadIdx.x]; }
  __threadfence_system();
  __syncthreads();
  if(threadIdx.x == 0) { h[TOTAL_WORDS - 1] = ticket; }
}

// First occurrences of every value inside its (sorted) segment -- a value that differs from its predecessor (k_mark_changes, one
// streaming pass) or the first value of a non-empty query (k_mark_starts, one lane per query, afterwards) -- as a BIT MAP, one
// 64-bit word per 64 values (a wavefront's ballot), and their exclusive prefix sums per WORD (k_word_counts + a scan over
// total / 64 counts).  The place of a kept value is its word's prefix + the ones below it in the word.  (Round 2 found the
// owner of every value by binary search over the offsets; the first half of round 3 kept 32-bit flags and their scan per
// VALUE: 1.3 GB of traffic that is now 30 MB.)  words: total / 64 + 1 (bit `total` exists and is 0); word_before: one more.
__global__ __launch_bounds__(TPB) void k_mark_changes(const u64* __restrict__ sorted, u64 total, u64* __restrict__ words)
{
  const u64 g = u64(blockIdx.x) * TPB + threadIdx.x;            // the grid covers whole words up to bit `total`
  const bool first = (g < total && (g == 0 || sorted[g] != sorted[g - 1]));
  const u64 mask = __ballot(first);
  if((threadIdx.x & 63) == 0 && g <= total) { words[g >> 6] = mask; }
}

__global__ __launch_bounds__(TPB) void k_mark_starts(const u64* __restrict__ raw_off, u64 nq, u64* __restrict__ words)
{
  u64 q = u64(blockIdx.x) * TPB + threadIdx.x;
  if(q >= nq) { return; }
  const u64 b = raw_off[q];
  if(raw_off[q + 1] > b) { atomicOr(reinterpret_cast<unsigned long long*>(words + (b >> 6)), 1ull << (b & 63)); }
}

__global__ __launch_bounds__(TPB) void k_word_counts(const u64* __restrict__ words, u64 nwords, u32* __restrict__ counts)
{
  const u64 w = u64(blockIdx.x) * TPB + threadIdx.x;
  if(w <= nwords) { counts[w] = (w < nwords ? u32(__popcll(words[w])) : 0u); }          // entry nwords = 0: the scan's total
}

__device__ __forceinline__ u64 kept_before(const u64* __restrict__ words, const u32* __restrict__ word_before, u64 g)
{
  return u64(word_before[g >> 6]) + u64(__popcll(words[g >> 6] & ((u64(1) << (g & 63)) - 1)));
}

// (capacity: a caller-owned buffer is written while its size is still unchecked on the host -- nothing lands outside it)
__global__ __launch_bounds__(TPB) void k_compact(const u64* __restrict__ sorted, const u64* __restrict__ words,
                                                 const u32* __restrict__ word_before, u64 total, u64* __restrict__ out, u64 capacity)
{
  u64 g = u64(blockIdx.x) * TPB + threadIdx.x;
  if(g >= total) { return; }
  const u64 word = words[g >> 6];
  if((word >> (g & 63)) & 1)
  {
    const u64 dest = u64(word_before[g >> 6]) + u64(__popcll(word & ((u64(1) << (g & 63)) - 1)));
    if(dest < capacity) { out[dest] = sorted[g]; }
  }
}

// k_mark_changes + k_word_counts + the scan + k_compact in ONE sweep over the sorted values, for callers that own the values
// buffer (its size need not be known before the values are written): a workgroup takes the next tile of COMPACT_TILE values
// (a ticket: tiles start in order), marks the first occurrences -- `words` arrives with the segment starts set (k_mark_starts on
// a cleared bitmap) and leaves with all marks --, publishes the tile's count, finds the count of everything in front of it by
// looking back over the tiles before it (their counts, until one that already knows its own prefix: the decoupled look-back of
// single-pass scans), and writes its first occurrences in place.  The 4.3 GB of the 16-mer batch on the 2^30-base text are read
// once instead of twice.  status[tile]: bits 62-63 = 1 count of the tile / 2 count of everything up to and including it.
// (a tile of 8192 values: with 2048 the look-back and the ticket of four times as many tiles cost more than the second read
// they save -- 3.6 ms against 3.3 for the four kernels on the 16-mer batch; 4096: 2.45 ms; 8192: 2.19 ms.  Again at the end of
// round 6, on the 16-mer batch of the 2^23 repeat graph, where two slots in three are dead: 8192 on 124 registers 1.38 ms, 4096
// on 68 registers 1.87 ms; the look-back over 256 tiles per round trip instead of 64: 1.62 ms.  A tile costs its chain of round
// trips -- ticket, bitmap words, values, look-back -- whatever it holds, and the kernel runs as many chains as fit a CU.)
constexpr u32 COMPACT_THREADS = 256, COMPACT_ROWS = 32, COMPACT_TILE = COMPACT_THREADS * COMPACT_ROWS;
constexpr u64 TILE_COUNT = u64(1) << 62, TILE_PREFIX = u64(2) << 62, TILE_VALUE = TILE_COUNT - 1;

// the 64-bit value of one lane (a compile-time lane: v_readlane, a scalar result -- __shfl goes through ds_bpermute whatever the lane)
__device__ __forceinline__ u64 lane_value(u64 v, int from)
{
  return (u64(u32(__builtin_amdgcn_readlane(int(u32(v >> 32)), from))) << 32) | u64(u32(__builtin_amdgcn_readlane(int(u32(v)), from)));
}

__global__ __launch_bounds__(COMPACT_THREADS) void k_mark_compact(const u64* __restrict__ sorted, u64 total, u64 nwords, u64* __restrict__ words,
                                                                  u32* __restrict__ word_before, u64* __restrict__ out, u64 capacity,
                                                                  unsigned long long* __restrict__ status, unsigned int* __restrict__ ticket,
                                                                  unsigned long long* __restrict__ unique_out, const u64* __restrict__ dead)
{
  // IN PLACE (out == sorted, round 6) is safe: a tile writes in front of its own first value, and only once every tile before
  // it has published a count -- which a tile does after ALL its loads have arrived in registers, as this one's have by then.
  constexpr u32 WAVES = COMPACT_THREADS / 64, WORDS = COMPACT_ROWS * WAVES;       // words of the bitmap per tile
  static_assert(WORDS % 64 == 0 && WORDS <= 128, "the scan of the word counts below takes one or two entries per lane");
  __shared__ u32 counts[WORDS];
  __shared__ u32 s_tile;
  __shared__ unsigned long long s_before;
  const u32 tid = threadIdx.x, lane = tid & 63;
  const u32 wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // (a scalar for the compiler too: the words of the bitmap are then scalar loads)
  if(tid == 0) { s_tile = atomicAdd(ticket, 1u); }
  __syncthreads();
  const u64 tile = __builtin_amdgcn_readfirstlane(s_tile), base = tile * COMPACT_TILE;
  // wavefront `wave` takes COMPACT_ROWS consecutive words of the tile: row j = the 64 values of word (base / 64 + wave ROWS + j)
  const u64 mine0 = base + u64(wave) * COMPACT_ROWS * 64 + lane;
  u64 value[COMPACT_ROWS];
  // (`dead`: slots that hold nothing -- the tails behind the distinct values k_dedup_huge left at the front of its segments;
  // they are neither loaded nor marked.  The slot behind a dead stretch starts a segment, so no live value is compared with one.)
  // What a lane knows about its own 32 slots is kept as BITS of three registers (round 6): the rows' words of the bitmaps are
  // uniform, and 32 live words + 32 mark words held across the look-back were 128 scalar registers -- more than a wavefront has;
  // the spills made this kernel 154 vector registers wide, three workgroups on a CU.  The marks of a row are one ballot away.
  static_assert(COMPACT_ROWS <= 32, "a bit per row");
  u32 my_live = 0, my_start = 0, my_keep = 0;
  const u64* __restrict__ wave_values = sorted + (base + u64(wave) * COMPACT_ROWS * 64);      // (uniform: one base, 32-bit lane offsets)
  // (the wavefront's 32 words of either bitmap: ONE coalesced load each, lane j holds the words of row j)
  const u64 w_mine = (base >> 6) + u64(wave) * COMPACT_ROWS + lane;
  const bool w_exists = (lane < COMPACT_ROWS && w_mine < nwords);
  const u64 dead_words = (dead != nullptr && w_exists ? dead[w_mine] : 0);
  const u64 start_words = (w_exists ? words[w_mine] : 0);
#pragma unroll
  for(u32 j = 0; j < COMPACT_ROWS; j++)                       // (all loads of the tile leave before the first is looked at)
  {
    const u64 g = mine0 + j * 64;
    const u64 live = ~lane_value(dead_words, int(j));
    const u64 starts = lane_value(start_words, int(j));        // segment starts
    const bool alive = (g < total && ((live >> lane) & 1) != 0);
    value[j] = (alive ? wave_values[lane + j * 64] : 0);
    my_live |= u32(alive) << j;
    my_start |= u32((starts >> lane) & 1) << j;
  }
  u64 carry = (lane == 0 && mine0 > 0 && mine0 < total ? sorted[mine0 - 1] : 0);       // the value in front of the wavefront's first
#pragma unroll
  for(u32 j = 0; j < COMPACT_ROWS; j++)
  {
    const u64 g = mine0 + j * 64;
    // (the left neighbour: wave_shr:1, a DPP move per half -- no trip through the LDS crossbar; lane 0 keeps its own value and takes the carry)
    const u64 left = (u64(u32(__builtin_amdgcn_update_dpp(int(u32(value[j] >> 32)), int(u32(value[j] >> 32)), 0x138, 0xF, 0xF, false))) << 32)
                     | u64(u32(__builtin_amdgcn_update_dpp(int(u32(value[j])), int(u32(value[j])), 0x138, 0xF, 0xF, false)));
    const bool first = (g == 0 || value[j] != (lane == 0 ? carry : left));
    carry = lane_value(value[j], 63);                          // (lane 0's predecessor in the next row)
    const bool keep = (((my_live >> j) & 1) != 0 && (first || ((my_start >> j) & 1) != 0));
    my_keep |= u32(keep) << j;
    const u64 marks = __ballot(keep);
    if(lane == 0) { counts[wave * COMPACT_ROWS + j] = u32(__popcll(marks)); }
  }
  __syncthreads();
  // exclusive prefix sums of the word counts of the tile (one wavefront, 64 words at a time), the tile's count, and the look-back
  if(wave == 0)
  {
    u32 tile_count = 0;
#pragma unroll
    for(u32 half = 0; half < WORDS; half += 64)
    {
      const u32 mine = counts[half + lane];
      const u32 incl = wave_scan_dpp(mine);
      counts[half + lane] = tile_count + incl - mine;
      tile_count += last_lane_value(incl);
    }
    // the look-back, 64 tiles at a time: lane l reads the status of tile (tile - 1 - l) of the window; the nearest tile that
    // knows its prefix ends the walk, the counts of the tiles in front of it are added up.  (One lane walking back tile by tile
    // met hundreds of resident tiles that had published a count and not yet a prefix: 4.8 ms against 3.3 for the four kernels.)
    u64 before = 0;
    if(tile == 0) { if(lane == 0) { __hip_atomic_store(status, TILE_PREFIX | u64(tile_count), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } }
    else
    {
      if(lane == 0) { __hip_atomic_store(status + tile, TILE_COUNT | u64(tile_count), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      for(u64 top = tile; top > 0; )                            // window: tiles top - 1, top - 2, ..., top - 64 (uniform loop)
      {
        const bool mine_exists = (u64(lane) < top);
        unsigned long long seen = 0;
        do
        {
          seen = (mine_exists ? __hip_atomic_load(status + (top - 1 - lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : TILE_PREFIX);
        }
        while(__ballot((seen >> 62) == 0) != 0);               // (every tile of the window holds a smaller ticket: running or done)
        const u64 knows = __ballot((seen >> 62) == 2);          // lanes whose tile knows its prefix (a lane in front of tile 0 counts as one, with 0)
        const u32 stop = (knows != 0 ? u32(__ffsll((long long)knows)) - 1 : 64u);
        // (the tiles in front of the nearest one that knows its prefix hold COUNTS, at most COMPACT_TILE each: their sum by DPP in 32
        // bits; the prefix itself -- one lane's 64-bit value -- by v_readlane at a scalar lane index: no ds_bpermute in the look-back)
        const u32 small = last_lane_value(wave_scan_dpp(lane < stop ? u32(seen & TILE_VALUE) : 0u));
        u64 known = 0;
        if(stop < 64)
        {
          const u64 value = seen & TILE_VALUE;
          known = (u64(u32(__builtin_amdgcn_readlane(int(u32(value >> 32)), int(stop)))) << 32) | u64(u32(__builtin_amdgcn_readlane(int(u32(value)), int(stop))));
        }
        before += known + small;
        if(knows != 0) { break; }
        top -= 64;
      }
      if(lane == 0) { __hip_atomic_store(status + tile, TILE_PREFIX | (before + tile_count), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    }
    if(lane == 0)
    {
      s_before = before;
      if(base + COMPACT_TILE > total)                          // the last tile (bit `total` lies in it): the number of distinct values
      {
        *unique_out = before + tile_count;
        word_before[nwords] = u32(before + tile_count);
      }
    }
  }
  __syncthreads();
  const u64 before = s_before;
#pragma unroll
  for(u32 j = 0; j < COMPACT_ROWS; j++)
  {
    const u64 g = mine0 + j * 64;
    const u64 w = g >> 6;
    const bool keep = (((my_keep >> j) & 1) != 0);
    const u64 marks = __ballot(keep);
    const u64 word_first = before + counts[wave * COMPACT_ROWS + j];
    if(lane == 0 && w < nwords) { words[w] = marks; word_before[w] = u32(word_first); }
    if(keep)
    {
      const u64 dest = word_first + u64(__popcll(marks & ((u64(1) << lane) - 1)));
      if(dest < capacity) { out[dest] = value[j]; }
    }
  }
}

__global__ void k_publish(const u32* __restrict__ src, unsigned long long* __restrict__ dst) { *dst = *src; }

// in place: offsets[] holds the raw (with duplicates) offsets on entry, the final ones on return; total_unique = word_before[nwords]
__global__ __launch_bounds__(TPB) void k_final_offsets(const u64* __restrict__ words, const u32* __restrict__ word_before, u64 nq, u64 total,
                                                       u64 nwords, u64* offsets)
{
  u64 q = u64(blockIdx.x) * TPB + threadIdx.x;
  if(q > nq) { return; }
  u64 r = (q < nq ? offsets[q] : total);
  offsets[q] = (r < total ? kept_before(words, word_before, r) : u64(word_before[nwords]));
}


}  // namespace
