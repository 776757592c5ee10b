// kernels_mailbox.hpp -- a resident single-wavefront kernel that answers scalar calls (round 6).
// Part of the single translation unit gcsa2_hip.hip (device code, anonymous namespace).
//
// The reference's caller shape for this path is one call per character: `range = index.LF(range, comp)` in a loop
// (include/gcsa/gcsa.h:155-162; vg's MEM finder adds parent(), src/lcp.cpp:276-301, when a step empties).  Through a kernel
// launch such a call costs 14-17 us -- the launch and the completion signal, not the work -- against 0.35 us for
// a CPU LF step (paper.tex:408).  Here the host posts the request in a page-locked, coherent slot that ONE wavefront polls, and
// polls the same slot for the answer: no launch, no stream synchronisation per call.  The wavefront leaves by itself once it
// has seen no request for `park_ticks` (so that a device-wide synchronisation elsewhere in the process waits that long at
// most), or after `life_ticks` whatever happens; the next call launches it again (one launch, amortised over the loop).
#pragma once

#include "kernels_lcp.hpp"

namespace {

enum : unsigned long long { MAIL_QUIT = 0, MAIL_LF = 1, MAIL_COUNT = 2, MAIL_PARENT = 3, MAIL_LF_NODE = 4, MAIL_NOP = 5 };

// Two 64-byte lines.  The first travels host -> device and is read by ONE load instruction of eight lanes (one 64-byte read
// over the bus: a line is read as a whole, and the host writes the ticket last, so a line that shows a new ticket shows its
// arguments); the second travels device -> host as ONE store instruction of eight lanes, with the ticket in its first AND its
// last word -- should the write ever arrive as two halves, the host sees both tickets only when both halves have arrived.
// (The first version read the ticket, then the arguments, and wrote five results, a fence and the ticket: 5.7 us per LF call
// against 12.4 us through a launch; each of those steps is a trip over the bus.)
struct MailSlot
{
  unsigned long long op;                 // host -> device
  unsigned long long arg[4];
  unsigned long long pad[2];
  unsigned long long request;            // ticket of the request the host has posted: written LAST
  unsigned long long front;              // device -> host: ticket of the answered request (first word of the line) ...
  unsigned long long result[5];
  unsigned long long alive;              // set by the host at launch, cleared by the wavefront when it has decided to leave
  unsigned long long done;               // ... and again as the last word
};
static_assert(sizeof(MailSlot) == 128, "two 64-byte lines");

__device__ __forceinline__ u64 uniform64(u64 x)
{
  return u64(u32(__builtin_amdgcn_readfirstlane(int(u32(x))))) | (u64(u32(__builtin_amdgcn_readfirstlane(int(u32(x >> 32))))) << 32);
}
__device__ __forceinline__ u64 lane_value64(u64 x, u32 from)
{
  return u64(u32(__builtin_amdgcn_readlane(int(u32(x)), int(from)))) | (u64(u32(__builtin_amdgcn_readlane(int(u32(x >> 32)), int(from)))) << 32);
}

// `last`: the ticket answered before this instance started.  Leaving: alive = 0 is published FIRST, then the request line is
// looked at once more -- a request posted while the host still read alive == 1 is answered; one posted later finds alive == 0 and
// the host launches the next instance (same stream: it starts when this one has gone).
__global__ __launch_bounds__(64) void k_mailbox(DevImage img, MailSlot* slot, unsigned long long last, unsigned long long park_ticks,
                                                unsigned long long life_ticks)
{
  __shared__ ulonglong2 stage[64 * 8];
  __shared__ Tables tables;
  stage_tables(img, tables);
  const u32 lane = threadIdx.x;
  unsigned long long* in_line = reinterpret_cast<unsigned long long*>(slot);
  unsigned long long* out_line = in_line + 8;
  const u64 t_start = wall_clock64();
  u64 t_idle = t_start;
  bool leaving = false;
  while(true)
  {
    unsigned long long word = 0;
    if(lane < 8) { word = __hip_atomic_load(in_line + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
    const unsigned long long r = lane_value64(word, 7);
    if(r != last)
    {
      const unsigned long long op = lane_value64(word, 0), a0 = lane_value64(word, 1), a1 = lane_value64(word, 2), a2 = lane_value64(word, 3);
      u64 res[5] = {0, 0, 0, 0, 0};
      if(op == MAIL_LF)                                        // (uniform branches: the request is the wavefront's)
      {
        lf_step_wave(img, a0, a1, u32(a2), lane == 0, stage, lane, res[0], res[1]);
      }
      else if(op == MAIL_COUNT) { if(lane == 0) { res[0] = count_range(img, a0, a1); } }
      else if(op == MAIL_PARENT)
      {
        if(lane == 0)
        {
          gcsa2_stnode node;
          lcp_parent(img, a0, a1, node);
          res[0] = node.sp; res[1] = node.ep; res[2] = node.left_lcp; res[3] = node.right_lcp; res[4] = node.node_lcp;
        }
      }
      else if(op == MAIL_LF_NODE) { if(lane == 0) { res[0] = (a0 < img.n ? lf_node(img, tables.C, a0) : 0); } }
      const bool last_one = (op == MAIL_QUIT || leaving);
      // the answer: lane 0 and lane 7 carry the ticket, lanes 1..5 the results (lane 0 computed them), lane 6 `alive`
      unsigned long long mine = r;
#pragma unroll
      for(u32 k = 0; k < 5; k++) { const u64 v = uniform64(res[k]); mine = (lane == k + 1 ? v : mine); }
      mine = (lane == 6 ? (last_one ? 0ull : 1ull) : mine);
      if(lane < 8) { __hip_atomic_store(out_line + lane, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
      last = r;
      if(last_one) { break; }
      t_idle = wall_clock64();
      continue;
    }
    if(leaving) { break; }
    const u64 now = wall_clock64();
    if(now - t_idle > park_ticks || now - t_start > life_ticks)
    {
      if(lane == 0) { __hip_atomic_store(&slot->alive, 0ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
      __threadfence_system();
      leaving = true;                                          // one more look at the request line, then out
    }
  }
}

}  // namespace
