// kernels_ms3.hpp -- matching statistics, version 3 (round 5): the LF + LCPArray::parent loop of k_match_stats2
// (kernels_lcp.hpp; reference: include/gcsa/gcsa.h:155-162, src/lcp.cpp:276-301, caller shape src/algorithms.cpp:146-167)
// with the two things round 4's counters said it pays for removed (profiles/r05_match_stats.md):
//
//  (a) PATTERN RECORDS.  The pre-pass writes one 16-byte record per 32 pattern characters, {2-bit codes, "not a fast
//      character" flags}, and a lane keeps the current record and the NEXT one in registers; the next-but-one is requested
//      the moment a record is entered (32 characters, >= 16 rounds, before it is needed).  k_match_stats2 re-read two code
//      words and two flag words from two arrays whenever its 32-character window ran low -- every 24 characters, consumed
//      at once: ~44 memory requests per 256-bp pattern (a sixth of the kernel's reads, all of them waited for by the whole
//      wave).  Here it is one request per 32 characters and nobody waits.
//
//  (b) LOCAL RECOVERY.  After a mismatch the search spends ~16 positions on matches of log4(n) characters; nearly every
//      step there fails once, takes parent() and succeeds on the retry: step (block), parent (LCP window), step (the SAME
//      block again, the parent interval is a few path nodes wider) -- three dependent round trips and three requests per
//      position.  A lane now has a second LDS slot for the LCP window (MS3_WINDOWS per wave, handed out by ballot rank), so
//      the block of the failed step stays staged: parent() and the retry run in one round from the two slots, and while a
//      lane is in its cool-down after a parent() the window around its range is requested TOGETHER with the step's block
//      (`speculate`), so that fail -> parent -> retry is one round and two requests.  A lane that asks for no new block reloads
//      the one it holds (an L2 hit), so nothing overwrites a block that a retry may still need.
//
// Results are those of k_match_stats2 bit for bit (tests/test_gpu_parity.py::test_match_stats_kernel_variants).
#pragma once

namespace {

constexpr u32 MS3_WINDOWS = 32;              // LCP-window slots per wavefront (128 bytes each)
constexpr u32 MS3_NONE = ~u32(0);
#ifndef MS3_WAVES
#define MS3_WAVES 3                          // waves per SIMD the kernel is compiled for (168 VGPRs; it needs ~155, and LDS holds six workgroups)
#endif

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

// Both endpoints of a single-character step from ONE staged FLB128 block (eval_staged twice, with the header and the running
// popcounts read once): raw = C[c] + rank(B_c, .), node = rank(edges, raw) for sp and rank(edges, raw - 1) for ep + 1.
__device__ __forceinline__ void eval_both(const ulonglong2* wave_stage, u32 lane, u32 o_sp, u32 o_ep, u64& a_raw, u64& a_node, u64& b_raw, u64& b_node)
{
  const ulonglong2 head = wave_stage[lane * 8 + (lane & 7)];             // w0, w1
  const ulonglong2 tail = wave_stage[lane * 8 + (7 ^ (lane & 7))];       // w14, w15
  const u32 wq_a = o_sp >> 6, wq_b = o_ep >> 6;
  const u64 word_a = staged_word(wave_stage, lane, 2 + wq_a), word_b = staged_word(wave_stage, lane, 2 + wq_b);
  const u32 ones_a = (wq_a == 0 ? 0u : u32(tail.x >> (10 * (wq_a - 1))) & 0x3FF) + u32(__popcll(word_a & ((u64(1) << (o_sp & 63)) - 1)));
  const u32 ones_b = (wq_b == 0 ? 0u : u32(tail.x >> (10 * (wq_b - 1))) & 0x3FF) + u32(__popcll(word_b & ((u64(1) << (o_ep & 63)) - 1)));
  a_raw = head.x + ones_a; b_raw = head.x + ones_b;
  const u64 ncnt = head.y & ~PREV_BIT;
  const u32 k_a = ones_a, k_b = (ones_b > 0 ? ones_b - 1 : 0u);
  const u32 kq_a = k_a >> 6, kq_b = k_b >> 6;
  const u64 eword_a = staged_word(wave_stage, lane, 8 + kq_a), eword_b = staged_word(wave_stage, lane, 8 + kq_b);
  a_node = ncnt + (kq_a == 0 ? 0u : u32(tail.y >> (10 * (kq_a - 1))) & 0x3FF) + u32(__popcll(eword_a & ((u64(1) << (k_a & 63)) - 1)));
  b_node = (ones_b == 0 ? ncnt - (head.y >> 63)
                        : ncnt + (kq_b == 0 ? 0u : u32(tail.y >> (10 * (kq_b - 1))) & 0x3FF) + u32(__popcll(eword_b & ((u64(1) << (k_b & 63)) - 1))));
}

// The wave's fetch of a round: the block at `block_addr` into every lane's slot of `wave_stage`, and the LCP windows of the
// first `nwin` window slots (addresses in wave_waddr[0, nwin)) into `wave_wstage`; gfx950's direct global -> LDS loads, the XOR
// swizzle on the source side (fetch_blocks_issue, kernels_find.hpp).  No predication (round 5 tried one exec-masked load per
// owner: eight branches per fetch, ~60 scalar instructions a round): a lane that asks for nothing new passes the address of the
// block it ALREADY holds -- the reload is served by L2 and leaves the slot as it is, so a retry can still read it -- and a
// lane without a pattern the array's first block.
__device__ __forceinline__ void ms3_issue(u64 block_addr, ulonglong2* wave_stage, u64* wave_addr, u32 lane,
                                          ulonglong2* wave_wstage, const u64* wave_waddr, u32 nwin, u64 dummy_window)
{
  __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0): every read of the slots' previous contents has returned
  __builtin_amdgcn_wave_barrier();
  wave_addr[lane] = block_addr;
  __builtin_amdgcn_wave_barrier();
  const u32 lds_base = __builtin_amdgcn_readfirstlane(u32(reinterpret_cast<size_t>(wave_stage)));
  const u32 group = lane >> 3;
  const u64 chunk = u64(((lane & 7) ^ (group & 7)) * 16);
  u64 src[8];
#pragma unroll
  for(u32 j = 0; j < 8; j++) { src[j] = wave_addr[8 * j + group]; }
#pragma unroll
  for(u32 j = 0; j < 8; j++)
  {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + chunk),
                                     (__attribute__((address_space(3))) void*)(size_t(lds_base + j * 1024u)), 16, 0, 0);
  }
  if(nwin > 0)                                  // uniform
  {
    const u32 wbase = __builtin_amdgcn_readfirstlane(u32(reinterpret_cast<size_t>(wave_wstage)));
#pragma unroll
    for(u32 j = 0; j < MS3_WINDOWS / 8; j++)
    {
      if(8 * j < nwin)                          // uniform: eight window slots per instruction, the unused ones of the last group read `dummy_window`
      {
        const u32 slot = 8 * j + group;
        const u64 a = (slot < nwin ? wave_waddr[slot] : dummy_window);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a + chunk),
                                         (__attribute__((address_space(3))) void*)(size_t(wbase + j * 1024u)), 16, 0, 0);
      }
    }
  }
}

// PAIR / REFILL / PROF / BREAKS as in k_match_stats2.  prof[0..15]: cycles of 0 stores + records + finished / new patterns,
// 1 record advance + plan + issue, 2 wait, 3 first evaluation, 4 second fetch + evaluation, 5 outcome, 6 recovery: parent(), 7 recovery: retry from the staged block;
// events 8 rounds (per wave), 9 lane second fetches, 10 lane steps (= blocks requested first), 11 lane pair attempts, 12 failed pair
// attempts, 13 parent() calls, 14 LCP windows requested, 15 retries answered from the staged block.
template<bool PAIR, bool REFILL, bool PROF = false, bool BREAKS = false>
__global__ __launch_bounds__(TPB2, MS3_WAVES) void k_match_stats3(DevImage img, const u8* __restrict__ patterns,
                                                       const u64* __restrict__ offsets, u64 nq,
                                                       unsigned short* __restrict__ ms, u64* __restrict__ ranges,
                                                       u64* __restrict__ fallbacks, u32 cool_down,
                                                       unsigned long long* __restrict__ queue, u32 refill_at,
                                                       const ulonglong2* __restrict__ recs, u32 speculate,
                                                       unsigned long long* __restrict__ prof = nullptr, BreakSink sink = BreakSink{nullptr, 0, nullptr, nullptr, 0})
{
  [[maybe_unused]] u64 prof_t = 0, prof_c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  [[maybe_unused]] u32 prof_n[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if constexpr(PROF) { prof_t = clock64(); }
#define G3_TICK(phase) do { if constexpr(PROF) { const u64 now_ = clock64(); prof_c[phase] += now_ - prof_t; prof_t = now_; } } while(0)
#define G3_COUNT(slot, value) do { if constexpr(PROF) { prof_n[slot] += u32(value); } } while(0)
  __shared__ ulonglong2 stage[TPB2 * 8];
  __shared__ ulonglong2 wstage[(TPB2 / 64) * MS3_WINDOWS * 8];
  __shared__ u64 addr_table[TPB2];
  __shared__ u64 waddr_table[(TPB2 / 64) * MS3_WINDOWS];
  __shared__ u8 c2c[256];
  c2c[threadIdx.x] = img.char2comp[threadIdx.x];
  c2c[threadIdx.x + TPB2] = img.char2comp[threadIdx.x + TPB2];
  __syncthreads();
  const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  ulonglong2* wave_stage = stage + (threadIdx.x & ~63u) * 8;
  ulonglong2* wave_wstage = wstage + wave * MS3_WINDOWS * 8;
  u64* wave_addr = addr_table + (threadIdx.x & ~63u);
  u64* wave_waddr = waddr_table + wave * MS3_WINDOWS;
  u64 q = 0, begin = 0;
  u32 i = 0, total = 0;                     // characters left / in all
  bool has = false;
  [[maybe_unused]] bool exhausted = false;
  u64 sp = 0, ep = img.n - 1;
  u32 depth = 0, calls = 0;
  bool need_parent = false;
  u32 force_single = 0;
  // the record that holds position i - 1 (slot (total - i) & 31), the one behind it (a pair step at slot 31 reads its first
  // character there), and the one behind that, REQUESTED when a record is entered and not looked at before the next entry: a
  // value that every round's plan reads must not be a load in flight (the compiler waits for it with vmcnt(0) wherever it is
  // read -- one exposed memory latency per round, 107 instead of 145 M patterns/s when `next` was loaded directly)
  u64 win_code = 0, next_code = 0, pend_code = 0;
  u32 win_bad = 0, next_bad = ~u32(0), pend_bad = ~u32(0);
  [[maybe_unused]] u64 packed_lo = 0, packed_hi = 0, packed_2 = 0, packed_3 = 0; [[maybe_unused]] u32 have = 0;
  [[maybe_unused]] u32 last_break = ~u32(0), n_breaks = 0;
  [[maybe_unused]] u64 blk_base = 0; [[maybe_unused]] u32 blk_used = BREAK_BLOCK;
  [[maybe_unused]] u32 min_length = sink.min_length;
  if constexpr(BREAKS) { asm volatile("" : "+v"(min_length)); }
  auto emit = [&](u32 pos, u32 value)        // ms[begin + pos] = value; positions arrive in descending order (k_match_stats2)
  {
    if constexpr(BREAKS) { return; }
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
  // `adv` characters were consumed (i is already lowered): entering the next record makes it the current one and requests
  // the one behind it -- needed 32 characters from now
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
  auto start = [&](u64 query)
  {
    q = query; has = true;
    begin = offsets[q]; i = total = u32(offsets[q + 1] - begin);
    sp = 0; ep = img.n - 1; depth = 0; calls = 0; need_parent = false; force_single = 0;
    if constexpr(BREAKS) { last_break = ~u32(0); n_breaks = 0; }
    const u64 word = (begin >> 5) + q;
    const ulonglong2 r0 = recs[word], r1 = recs[word + 1], r2 = recs[word + 2];
    win_code = r0.x; win_bad = u32(r0.y); next_code = r1.x; next_bad = u32(r1.y); pend_code = r2.x; pend_bad = u32(r2.y);
    // the k-mer seed table, as in k_match_stats2
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
  if constexpr(!REFILL)
  {
    const u64 gid = u64(blockIdx.x) * TPB2 + threadIdx.x;
    if(gid < nq) { start(gid); }
  }
  // the plan of the request in flight, kept across the loop edge
  bool planned = false, pair = false;
  u32 comp = 0, r_sp = 0, r_ep = 0, idx_sp = 0, emit_code = 0;
  bool split = false;                       // the endpoints of the step lie in different blocks (a second fetch: idx_sp + ep_delta)
  u32 ep_delta = 0;
  bool retryable = false;                   // the lane's slot holds single-step block idx_sp with BOTH endpoints in it
  u32 wslot = MS3_NONE;                     // the lane's LCP-window slot of this round
  // (a window starts 48..63 bytes before sp: the same expression where it is requested and where it is read -- sp does not
  // change between the two for a lane that fails)
  auto window_start = [&](u64 at) -> u64 { return lcp_window_start(at, ep); };             // (kernels_lcp.hpp: one line of the array when the interval sits well inside it)
  // (the block address without a select between two pointers: LLVM turns `flag ? img.flp : img.flb` into a per-lane LOAD of the
  // pointer -- from the kernel-argument segment, or from a two-entry array it spills to scratch for the purpose -- with a full
  // s_waitcnt vmcnt(0) in front of every fetch; an opaque register holding the distance between the arrays compiles to one v_cndmask)
  const u64 flb_addr = reinterpret_cast<u64>(img.flb);
  const u64 pair_delta = in_register(reinterpret_cast<u64>(img.flp) - reinterpret_cast<u64>(img.flb));
  auto block_address = [&](u32 idx) -> u64
  {
    u64 base = flb_addr;
    if constexpr(PAIR) { base += (idx & PAIR_FLAG) ? pair_delta : u64(0); }
    return base + u64(idx & ~PAIR_FLAG) * FLB_BYTES;
  };
  auto plan_and_issue = [&]()
  {
    const bool active = has && i > 0;
    const bool stepping = active && !need_parent, parenting = active && need_parent;
    pair = false;
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
            idx_sp = (first + b_sp) | PAIR_FLAG; ep_delta = b_ep - b_sp;
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
        idx_sp = comp * u32(img.flb_nblocks) + b_sp; ep_delta = b_ep - b_sp;
      }
      split = (ep_delta != 0);
      retryable = !pair && !split;
    }
    // LCP windows: a lane that owes a parent(), and -- speculatively -- a lane that steps singly in its cool-down (not at the root:
    // a failure there is "no such character", not a parent())
    const bool wish = parenting || ((speculate & 1u) != 0 && stepping && !pair && force_single > 0 && !(sp == 0 && ep == img.n - 1));
    const u64 wishes = __ballot(wish);
    const u32 rank = u32(__popcll(wishes & ((u64(1) << lane) - 1)));
    wslot = (wish && rank < MS3_WINDOWS ? rank : MS3_NONE);
    const u32 nwin = (u32(__popcll(wishes)) < MS3_WINDOWS ? u32(__popcll(wishes)) : MS3_WINDOWS);
    if(wslot != MS3_NONE)
    {
      wave_waddr[wslot] = reinterpret_cast<u64>(img.lcp) + window_start(sp);
    }
    planned = stepping || (parenting && wslot != MS3_NONE);   // (a parenting lane without a window slot waits a round)
    if(__any(planned)) { ms3_issue(block_address(idx_sp), wave_stage, wave_addr, lane, wave_wstage, wave_waddr, nwin, reinterpret_cast<u64>(img.lcp)); }
  };
  plan_and_issue();
  while(true)
  {
    const bool active = planned;
    const bool stepping = active && !need_parent;
    bool recover = active && need_parent;                      // owes a parent() (and has its window)
    PairEnd p_sp = {0, 0, 0}, p_ep = {0, 0, 0};
    const bool need2 = stepping && split;
    emit_code = 0;
    [[maybe_unused]] bool pending = false;                     // BREAKS: a character that does not occur at all
    [[maybe_unused]] bool failed_here = false;                 // BREAKS: the step of this round failed: (f_pos, f_depth, f_sp, f_ep) is a break
    [[maybe_unused]] u32 f_depth = 0; [[maybe_unused]] u64 f_sp = 0, f_ep = 0;
    G3_COUNT(0, lane == 0); G3_COUNT(2, stepping); G3_COUNT(3, pair); G3_COUNT(1, need2); G3_COUNT(6, active && wslot != MS3_NONE);
    if(__any(active))
    {
      fetch_blocks_wait();
      G3_TICK(2);
      if(stepping)
      {
        p_sp = eval_staged(wave_stage, lane, PAIR && pair, r_sp, false);
        if(!split) { p_ep = eval_staged(wave_stage, lane, PAIR && pair, r_ep, true); }
      }
      G3_TICK(3);
      if(__any(need2))
      {
        ms3_issue(block_address(need2 ? idx_sp + ep_delta : idx_sp), wave_stage, wave_addr, lane, wave_wstage, wave_waddr, 0, 0);
        fetch_blocks_wait();
        if(need2) { p_ep = eval_staged(wave_stage, lane, PAIR && pair, r_ep, true); }
      }
      __builtin_amdgcn_wave_barrier();
      G3_TICK(4);
    }
    if(stepping)
    {
      if(PAIR && pair)
      {
        u64 a = 0, b = 0;
        if(pair_outcome(p_sp, p_ep, !split, a, b) == 2)
        {
          sp = p_sp.node; ep = p_ep.node;
          emit_code = 2;
          depth += 2; i -= 2;
        }
        else { force_single = 2; G3_COUNT(4, 1); }
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
        else
        {
          recover = true;
          if constexpr(BREAKS) { failed_here = true; f_depth = depth; f_sp = sp; f_ep = ep; }
        }
      }
    }
    G3_TICK(5);
    // Recovery: parent() from the lane's LCP window, then the character again from the block that is still staged, as often
    // as it takes (every parent() widens the interval); whatever the two slots cannot answer costs a round trip.
    const u64 wstart = window_start(sp);
    while(recover)
    {
      if(wslot == MS3_NONE) { need_parent = true; break; }     // no window yet: it is requested with the next round
      gcsa2_stnode node;
      bool decided = (speculate & 4u) == 0 && parent_near(wave_wstage, wslot, wstart, img.lcp_size, sp, ep, node);
      if(!decided) { decided = sp >= wstart && parent_from_window(wave_wstage, wslot, wstart, img.lcp_size, sp, ep, node); }
      if(!decided) { lcp_parent(img, sp, ep, node); }
      G3_COUNT(5, 1);
      calls++;
      sp = node.sp; ep = node.ep; depth = u32(node.node_lcp);
      need_parent = false;
      force_single = (force_single > cool_down ? force_single : cool_down);
      G3_TICK(6);
      u32 b_sp, b_ep, o_sp, o_ep;
      flb_block_of(sp, b_sp, o_sp); flb_block_of(ep + 1, b_ep, o_ep);
      const u32 first = comp * u32(img.flb_nblocks);
      if(!retryable || first + b_sp != idx_sp || b_ep != b_sp) { break; }       // the retry needs another block: next round
      G3_COUNT(7, 1);
      u64 a, a_node, b, b_node;
      eval_both(wave_stage, lane, o_sp, o_ep, a, a_node, b, b_node);
      b--;                                                       // gcsa.h:155-162
      if(!range_empty(a, b))
      {
        sp = a_node; ep = b_node; depth++;
        emit_code = 1; i--;
        force_single -= (force_single > 0 ? 1 : 0);
        break;
      }
      if(sp == 0 && ep == img.n - 1)
      {
        depth = 0;
        if constexpr(BREAKS) { pending = true; }
        emit_code = 1; i--;
        force_single -= (force_single > 0 ? 1 : 0);
        break;
      }
      if(speculate & 2u) { need_parent = true; break; }         // (A/B: one parent() per round, the next one with a fresh window)
      G3_TICK(7);
    }
    G3_TICK(7);
    if(emit_code != 0) { consumed(emit_code); }                 // (one site: the characters this round consumed, 0..2)
    plan_and_issue();                                            // the next round's requests leave here
    G3_TICK(1);
    if constexpr(!BREAKS)
    {
      if(emit_code == 2) { emit(i + 1, depth - 1); }
      if(emit_code != 0) { emit(i, depth); }
    }
    if constexpr(BREAKS)
    {
      // One record per lane and round, in descending position order: the failed step's (the state BEFORE parent(): the match
      // that starts at f_pos cannot be extended to the left), else the position behind a character that does not occur, else
      // the pattern's first position.  When two are due -- the step behind a break reached position 0 -- the pattern ends a
      // round later (last_break != 0 keeps it alive; the lane is inactive and only writes its last record).
      const u32 f_pos = i + (emit_code != 0 ? 1u : 0u);         // the recovery consumed at most one character
      u32 pos = i, d = depth; u64 a = sp, b = ep;
      bool is_break = has && total > 0 && i == 0 && last_break != 0;
      if(failed_here) { is_break = last_break != f_pos; pos = f_pos; d = f_depth; a = f_sp; b = f_ep; }
      else if(pending) { is_break = has && last_break != i + 1 && i + 1 < total; pos = i + 1; }
      if(is_break) { last_break = pos; }
      const bool record = is_break && d >= min_length;
      const u64 writers = __ballot(record);
      if(writers != 0)
      {
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
            dst[0] = make_ulonglong2(q | (u64(n_breaks) << 32), u64(pos) | (u64(d) << 32));
            dst[1] = make_ulonglong2(a, b);
          }
          n_breaks++;
        }
        if(count > room) { blk_base = next_base; blk_used = count - room; } else { blk_used += count; }
      }
    }
    bool finished = has && i == 0;
    if constexpr(BREAKS) { finished = finished && (total == 0 || last_break == 0); }
    if(finished)                                               // pattern finished (or empty): final range, parent() count
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
    G3_TICK(0);
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
#undef G3_TICK
#undef G3_COUNT
}

}  // namespace
