// gcsa2_hip.hip -- MI355X (gfx950) batched backward-search engine for GCSA2 indexes:
// kernels + the C ABI of include/gcsa2_hip.h.  Written for CDNA4 only (wave64, no CUDA paths).
//
// One translation unit: the device code lives in kernels_*.hpp, this file holds the host side
// (device image staging, handles, launches, the extern "C" entry points).
//
// Kernel <-> reference map (paths relative to the reference tree):
//   kernels_find.hpp    k_find2 / k_find      GCSA::find                 include/gcsa/gcsa.h:96-110
//                       k_lf2                 GCSA::LF(range, comp)      include/gcsa/gcsa.h:155-162, 262-274
//                       k_lf_node             GCSA::LF(path_node)        include/gcsa/gcsa.h:165-183
//                       k_lf_all              GCSA::LF_fast / LF_all     src/gcsa.cpp:742-798
//   kernels_locate.hpp  k_count               GCSA::count, Sada*::count  src/gcsa.cpp:802-809, support.h:255-258,329-335
//                       k_locate_*            GCSA::locate(range), locateInternal, removeDuplicates
//                                                                        src/gcsa.cpp:827-842, 880-896, utils.h:350-357
//                       k_kmer_expand         countKMers                 src/algorithms.cpp:364-421
//   kernels_lcp.hpp     k_parent / k_depth / k_sv / k_rmq   LCPArray     include/gcsa/lcp.h:137-178, src/lcp.cpp:276-519
#include "layout.hpp"
#include "sdsl_reader.hpp"
#include "sdsl_writer.hpp"
#include "../../include/gcsa2_hip.h"

#include <hipcub/hipcub.hpp>
#include <chrono>
#include <condition_variable>
#include <cstdio>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <functional>
#include <atomic>
#include <memory>
#include <mutex>
#include <new>
#include <random>
#include <string>
#include <thread>
#include <vector>

using namespace g2;

#include "kernels_common.hpp"
#include "kernels_find.hpp"
#include "kernels_locate.hpp"
#include "kernels_lcp.hpp"
#include "kernels_mailbox.hpp"
#include "kernels_build.hpp"

// ==========================================================================================
// host code

constexpr unsigned RESULT_SLOTS = 1024;     // concurrent locate calls per handle before two share a slot

// Staging of the host-pointer entry points (`*_batch`): a stream, a grow-only device arena and a pinned host arena.
// A handle keeps a small pool of them, one per concurrent caller, so that a call costs no hipMalloc / hipFree
// and small transfers go through pinned memory asynchronously on the call's own stream.
struct Staging
{
  hipStream_t stream = nullptr;
  char* d = nullptr; size_t d_cap = 0;
  char* h = nullptr; size_t h_cap = 0;       // pinned (hipHostMalloc)
  char* z = nullptr;                         // pinned, coherent, ZERO_COPY_ARENA bytes: kernels of small calls read and write it directly
};

struct gcsa2_index
{
  int device = 0;
  DevImage img;
  void* d_base = nullptr;
  void* d_kmer = nullptr;
  void* d_pred4 = nullptr;
  void* d_locate = nullptr;
  void* d_jump = nullptr;
  void* d_pairs = nullptr;
  hipMemPool_t pool = nullptr;   // stream-ordered scratch of the query pipelines (owned; the default pool is not touched)
  // Small results the host reads back (totals of the locate pipeline) live in per-handle slots, not in per-call
  // scratch: no allocation per call.  (Round 1 moved them here after stale read-backs out of pool memory, about one
  // call in 5000; tools/hip/pool_readback_repro.hip shows the ROCm pool itself delivers 1.1 M such read-backs
  // correctly in every stream / host-memory combination, so that symptom was this engine's then-code, not ROCm.)
  unsigned long long* d_slots = nullptr;
  unsigned long long* h_slots = nullptr;     // the same slots in page-locked host memory the device writes and the host polls (read_totals)
  mutable std::atomic<unsigned> next_slot{0};
  mutable std::atomic<unsigned long long> next_ticket{1};
  mutable std::atomic<int> single_backoff{0};     // locate(): calls that skip the one-kernel attempt after it met a misfit (locate_chunk)
  // the resident wavefront that answers scalar calls (kernels_mailbox.hpp): its slot in page-locked memory, its stream
  struct Mailbox { MailSlot* slot = nullptr; hipStream_t stream = nullptr; std::mutex lock; unsigned long long ticket = 0, launches = 0, calls = 0; bool failed = false; };
  mutable Mailbox mail;
  mutable std::mutex staging_lock;
  mutable std::vector<Staging*> staging_pool;
  int compute_units = 256;
  u64 bytes = 0;
  u64 order = 0;
  // Host pipeline of the large host-pointer batches (gcsa2_find_batch): PIPE_LANES host threads, each with two pinned + device
  // staging sets and a stream of its own, made at the first large batch and kept (one pipelined call at a time per handle).
  struct PipeSet { char* h = nullptr; char* d = nullptr; hipEvent_t computed = nullptr, done = nullptr; bool busy = false; u64 first = 0, count = 0; };
  // uploads + kernel on `stream`, downloads on `down`: a stream that alternates copy directions gets two thirds of the link
  // (tests/perf/pcie_rate.hip: 34 + 34 GB/s mixed against 54 + 22 GB/s with one direction per stream)
  struct PipeLane { hipStream_t stream = nullptr, down = nullptr; PipeSet set[2]; };
  mutable std::mutex pipe_lock;
  mutable std::vector<PipeLane> pipe;
  // scratch arenas of the locate pipeline that are not in use (struct Scratch)
  struct Arena { char* base = nullptr; size_t bytes = 0; };
  mutable std::mutex arena_lock;
  mutable std::vector<Arena> arenas;
  // tuning knobs, read from the environment ONCE, when the index is created (A/B measurements; results never depend on them)
  struct Tuning
  {
    u32 cool_down = COOL_DOWN;         // GCSA2_COOL_DOWN: characters stepped singly after a step that needed parent()
    u32 ms_refill_at = MS_REFILL_AT;   // GCSA2_MS_REFILL_AT: persistent matching statistics, idle lanes of a wave that trigger a refill
    u64 ms_grid = 0;                   // GCSA2_MS_GRID: ... most workgroups launched (0: what the device holds at once)
    u32 sort_medium_limit = 0;         // GCSA2_SORT_MEDIUM=0 sends the 17..1024-value locate segments to the segmented radix sort
    u64 locate_split = (u64(1) << 31) - 1;   // GCSA2_LOCATE_SPLIT: most values (before deduplication) one pass of the locate pipeline handles
    u64 locate_split_queries = u64(1) << 30; // GCSA2_LOCATE_SPLIT_QUERIES: most ranges one pass handles (its lists and grids are 32-bit)
    u64 pipe_chunk = u64(1) << 18;     // GCSA2_PIPE_CHUNK (log2): patterns per chunk of the host pipeline
    u32 pipe_lanes = 6;                // GCSA2_PIPE_LANES: host threads (each with its streams and staging sets) of the large host batches
    u32 ms_threads = 4;                   // GCSA2_MS_THREADS: host threads (one stream each) that send the pieces
    u64 kmer_piece = u64(1) << 27;        // GCSA2_KMER_PIECE (tests): children of one piece of a countKMers / compareKMers frontier
    u64 ms_piece_bytes = u64(32) << 20;   // GCSA2_MS_PIECE_MB: pattern bytes per piece of the large host batches of matching statistics / break points
    bool pipe_blocking = false;        // GCSA2_PIPE_BLOCKING=1: the lanes' events are made with hipEventBlockingSync
    bool pipe_split = false;           // GCSA2_PIPE_SPLIT=1: downloads on a second stream per lane
    bool pipe_wide = false;            // GCSA2_PIPE_WIRE=16: the packed-pattern pipeline brings the ranges home as u64 pairs (A/B)
    bool ms_pieces = true;             // GCSA2_MS_PIECES=0: large host batches of matching statistics go through one copy in, one launch, one copy out
    bool dedup_narrow = true;          // GCSA2_DEDUP_NARROW=0: the duplicate filter's hash table holds 64-bit words even when the index's values fit 32 bits
    bool dedup_huge = true;            // GCSA2_DEDUP_HUGE=0 sends every locate segment of more than 8192 values to the device-wide radix sort, duplicates and all
    bool zero_copy = true;             // GCSA2_ZERO_COPY=0: small host-pointer calls copy through the arenas like large ones
    bool poll_small = true;            // GCSA2_POLL_SMALL=0: zero-copy calls end with hipStreamSynchronize instead of a polled ticket
    u32 seed_wide = (u32(1) << 24) - 1;   // GCSA2_SEED_WIDE: seed-table entries of this many path nodes or more are marked, not stored (tests)
    bool locate_trace = false;         // GCSA2_LOCATE_TRACE=1: host-clock stamps of a locate pass on stderr (profiles/r04_locate.md)
    u32 split_target = SPLIT_TARGET;   // GCSA2_SPLIT_TARGET (tests): values per bucket k_over_split aims at
    u32 split_skew = BIG_SEGMENT;      // GCSA2_SPLIT_SKEW (tests): buckets of more values than this count as skewed and go to the radix sort
    bool locate_split_sort = true;     // GCSA2_LOCATE_SPLIT_SORT=0: segments beyond 8192 distinct values go to the library's device-wide radix sort (round 4; A/B)
    bool locate_fused_compact = true;  // GCSA2_LOCATE_FUSED_COMPACT=0: caller-owned buffers also take the four-kernel compaction of the job interface (A/B)
    bool locate_single = true;         // GCSA2_LOCATE_SINGLE=0: batches of one-value ranges go through the general locate pipeline too (A/B)
    bool locate_fuse = true;           // GCSA2_LOCATE_FUSE=0: wide ranges of one-value path nodes go through the table pass like the others (A/B; round 6)
    u64 fuse_above = BIG_SEGMENT;      // GCSA2_LOCATE_FUSE_ABOVE (tests): path nodes from which such a range is a candidate for the fused split
    bool split_tiled = true;           // GCSA2_SPLIT_TILED=0: k_over_split scatters value by value, as in round 5 (A/B; round 6)
    bool mailbox = true;               // GCSA2_MAILBOX=0: one-query calls of LF / count / parent / LF(node) take the launch path like any batch (A/B; round 6)
    u64 mailbox_park_us = 200;         // GCSA2_MAILBOX_PARK_US: the resident wavefront leaves after this long without a request
    u64 mailbox_life_ms = 20;          // GCSA2_MAILBOX_LIFE_MS: ... and after this long whatever happens (the next call launches it again)
    bool locate_in_place = true;       // GCSA2_LOCATE_IN_PLACE=0: caller-owned buffers that hold the raw values are not used as the sort's target (A/B; round 6)
    size_t arena_cap = size_t(24) << 30;  // GCSA2_ARENA_CAP_MB: most scratch a handle keeps between calls per arena (struct Scratch)
    u64 budget_bytes = 0;              // GCSA2_MEMORY_BUDGET_MB: most device memory the image may take (0: what the device has free)
  } tune;
};

struct gcsa2_locate_job
{
  int device = 0;
  u64 nq = 0, total = 0;
  u64* d_offsets = nullptr;   // nq + 1
  u64* d_values = nullptr;    // total
  hipStream_t stream = nullptr;   // the stream the job ran on
};

namespace {

thread_local std::string g_error;

int fail(int code, const std::string& msg) { g_error = msg; return code; }

#define HIP_TRY(expr) do { hipError_t e_ = (expr); if(e_ != hipSuccess) { \
  return fail(e_ == hipErrorOutOfMemory ? GCSA2_ERR_OUT_OF_MEMORY : GCSA2_ERR_HIP, \
              std::string(#expr) + ": " + hipGetErrorString(e_)); } } while(0)

inline unsigned grid_for(u64 n) { return unsigned((n + TPB - 1) / TPB); }

// Host worker threads of one call.  If starting one of them throws (std::system_error: thread limit), the ones already
// running are joined before the exception travels on -- destroying a joinable std::thread ends the process.
struct Workers
{
  std::vector<std::thread> threads;
  template<class... Args> void emplace_back(Args&&... args) { threads.emplace_back(std::forward<Args>(args)...); }
  void join() { for(std::thread& t : threads) { if(t.joinable()) { t.join(); } } }
  ~Workers() { join(); }
};

// what a memory budget (GCSA2_MEMORY_BUDGET_MB, read at create time) leaves for the next optional table
inline u64 budget_left(const gcsa2_index* ix) { return ix->tune.budget_bytes > ix->bytes ? ix->tune.budget_bytes - ix->bytes : 0; }

inline void launch_walk(const gcsa2_index* ix, const u64* d_ranges, u64 nq, const u64* node_off, const u64* raw_off,
                        u64 total_nodes, u64* values, u64* owners, hipStream_t stream);

// ---- the optional tables of an image (pair blocks, k-mer seed table, locate table): built at create time and again by
// gcsa2_index_set_tables.  Each maker leaves the image without the table when it fails.

inline u64 pair_block_bytes_of(const DevImage& img) { return 16 * (img.n / PAIR_BITS + 1) * FLB_BYTES; }
inline bool pair_blocks_possible(const DevImage& img) { return img.sigma >= 5 && img.n > 0 && 16 * (img.n / PAIR_BITS + 1) < u64(PAIR_FLAG); }
inline bool seed_table_possible(const DevImage& img) { return img.sigma >= 5 && (img.n > img.e ? img.n : img.e) + 2 < (u64(1) << SEED_SP_BITS); }
inline bool locate_table_possible(const DevImage& img) { return img.has_samples && img.pred4 != nullptr && img.n > 0; }

void drop_pair_blocks(gcsa2_index* ix)
{
  if(ix->d_pairs != nullptr) { (void)hipFree(ix->d_pairs); ix->d_pairs = nullptr; ix->bytes -= pair_block_bytes_of(ix->img); }
  ix->img.flp = nullptr; ix->img.flp_nblocks = 0;
}

hipError_t make_pair_blocks(gcsa2_index* ix)
{
  drop_pair_blocks(ix);
  DevImage& img = ix->img;
  const u64 nb = img.n / PAIR_BITS + 1, pair_bytes = pair_block_bytes_of(img);
  hipError_t e = hipMalloc(&ix->d_pairs, pair_bytes);
  if(e == hipSuccess)
  {
    img.flp_nblocks = nb;
    const u64 slice = u64(1) << 20;          // blocks per launch: 2^20 x 4 workgroups of 192 threads
    for(u64 first = 0; first < nb && e == hipSuccess; first += slice)
    {
      const u64 count = (nb - first < slice ? nb - first : slice);
      hipLaunchKernelGGL(k_build_pair_blocks, dim3(unsigned(count), 4), dim3(unsigned(PAIR_BITS)), 0, nullptr, img, first, static_cast<u64*>(ix->d_pairs));
      e = hipGetLastError();
    }
    if(e == hipSuccess) { e = hipDeviceSynchronize(); }
  }
  if(e != hipSuccess)
  {
    if(ix->d_pairs != nullptr) { (void)hipFree(ix->d_pairs); ix->d_pairs = nullptr; }
    img.flp_nblocks = 0; (void)hipGetLastError();
    return e;
  }
  img.flp = static_cast<const u64*>(ix->d_pairs);
  ix->bytes += pair_bytes;
  return hipSuccess;
}

void drop_seed_table(gcsa2_index* ix)
{
  if(ix->d_kmer != nullptr) { (void)hipFree(ix->d_kmer); ix->d_kmer = nullptr; ix->bytes -= u64(8) << (2 * ix->img.kmer_k); }
  ix->img.kmer_table = nullptr; ix->img.kmer_k = 0;
}

// find() of every k-mer over comps 1..4, level by level in place (kernels_find.hpp: k_seed_level); 1 <= k <= 16
hipError_t make_seed_table(gcsa2_index* ix, u32 k)
{
  drop_seed_table(ix);
  DevImage& img = ix->img;
  img.seed_wide = ix->tune.seed_wide;
  const u64 entry_bytes = 8, entries = u64(1) << (2 * k);
  hipError_t e = hipMalloc(&ix->d_kmer, entries * entry_bytes);
  const u64 slice = u64(1) << 30;            // a HIP grid holds < 2^32 threads
  for(u32 j = 0; j < k && e == hipSuccess; j++)
  {
    // level j + 1 from level j: first the quarters with a non-zero leading code, then level j in place
    const u64 have = u64(1) << (2 * j), want = have << 2;
    for(int pass = 0; pass < 2 && e == hipSuccess; pass++)
    {
      const u64 lo = (j == 0 ? 0 : (pass == 0 ? have : 0)), hi = (j == 0 ? (pass == 0 ? want : 0) : (pass == 0 ? want : have));
      for(u64 first = lo; first < hi && e == hipSuccess; first += slice)
      {
        u64 count = (hi - first < slice ? hi - first : slice);
        hipLaunchKernelGGL(k_seed_level, dim3(grid_for(count)), dim3(TPB), 0, nullptr, img, j, first, first + count, static_cast<u64*>(ix->d_kmer));
        e = hipGetLastError();
      }
    }
  }
  if(e == hipSuccess) { e = hipDeviceSynchronize(); }
  if(e != hipSuccess)
  {
    if(ix->d_kmer != nullptr) { (void)hipFree(ix->d_kmer); ix->d_kmer = nullptr; }
    (void)hipGetLastError();
    return e;
  }
  img.kmer_table = static_cast<const u64*>(ix->d_kmer); img.kmer_k = k;
  ix->bytes += entries * entry_bytes;
  return hipSuccess;
}

void drop_locate_table(gcsa2_index* ix)
{
  if(ix->d_locate != nullptr) { (void)hipFree(ix->d_locate); ix->d_locate = nullptr; ix->bytes -= ix->img.n * sizeof(u64); }
  ix->img.locate_tab = nullptr;
}

// the walk of locateInternal (gcsa.cpp:880-896) memoised for every path node; false when an entry does not fit or on any error
bool make_locate_table(gcsa2_index* ix)
{
  drop_locate_table(ix);
  u32* d_overflow = nullptr; u32 overflow = 1;
  hipError_t e = hipMalloc(&ix->d_locate, ix->img.n * sizeof(u64));
  if(e == hipSuccess) { e = hipMalloc(reinterpret_cast<void**>(&d_overflow), sizeof(u32)); }
  if(e == hipSuccess) { e = hipMemset(d_overflow, 0, sizeof(u32)); }
  if(e == hipSuccess)
  {
    const u64 slice = u64(1) << 30;
    for(u64 first = 0; first < ix->img.n && e == hipSuccess; first += slice)
    {
      u64 count = (ix->img.n - first < slice ? ix->img.n - first : slice);
      hipLaunchKernelGGL(k_build_locate_table, dim3(unsigned((count + TPB2 - 1) / TPB2)), dim3(TPB2), 0, nullptr,
                         ix->img, first, static_cast<u64*>(ix->d_locate), d_overflow);
      e = hipGetLastError();
    }
    if(e == hipSuccess) { e = hipMemcpy(&overflow, d_overflow, sizeof(u32), hipMemcpyDeviceToHost); }
  }
  if(d_overflow) { (void)hipFree(d_overflow); }
  if(e == hipSuccess && overflow == 0)
  {
    ix->img.locate_tab = static_cast<const u64*>(ix->d_locate);
    ix->bytes += ix->img.n * sizeof(u64);
    return true;
  }
  if(ix->d_locate) { (void)hipFree(ix->d_locate); ix->d_locate = nullptr; }
  (void)hipGetLastError();
  return false;
}

// host-side staging of the device image --------------------------------------------------

// The image is laid out first (offsets only), allocated once, and filled by the kernels of kernels_build.hpp.
struct Planner
{
  u64 words = 0;               // image size so far, in u64 units; every piece starts on a 64-byte boundary
  u64 reserve(u64 nwords, u64 align_words = 8)
  {
    u64 off = (words + align_words - 1) / align_words * align_words;
    words = off + nwords;
    return off;
  }
};

struct BVPlan { u64 blocks_off = 0, hints_off = 0, size = 0, nblocks = 0, ones = 0, nhints = 0; bool hints = false; };

BVPlan plan_bv(Planner& pl, u64 size, bool with_select)
{
  BVPlan p;
  p.size = size; p.nblocks = size / BLOCK_BITS + 1; p.hints = with_select;
  p.blocks_off = pl.reserve(p.nblocks * BLOCK_WORDS);
  if(with_select)
  {
    p.nhints = size / SELECT_SAMPLE + 2;      // ones / SELECT_SAMPLE + 2 are read (bv_select); ones <= size is known after the build
    p.hints_off = pl.reserve((p.nhints + 1) / 2);
  }
  return p;
}

// Device scratch of one image build: a plain bit array in flight, the per-block popcounts and their prefix sums.
struct BuildScratch
{
  u64 *plain = nullptr, *counts = nullptr, *before = nullptr;
  void* cub = nullptr;
  size_t cub_bytes = 0;
  hipError_t alloc(u64 max_bits)
  {
    const u64 words = (max_bits + 63) / 64 + 1, blocks = max_bits / BLOCK_BITS + 1 + 2 * MAX_SIGMA;   // `before` also receives the charRange table
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&plain), words * sizeof(u64));
    if(e == hipSuccess) { e = hipMalloc(reinterpret_cast<void**>(&counts), blocks * sizeof(u64)); }
    if(e == hipSuccess) { e = hipMalloc(reinterpret_cast<void**>(&before), blocks * sizeof(u64)); }
    if(e == hipSuccess) { e = hipcub::DeviceScan::ExclusiveSum(nullptr, cub_bytes, counts, before, int(blocks)); }
    if(e == hipSuccess) { e = hipMalloc(&cub, cub_bytes > 0 ? cub_bytes : 8); }
    return e;
  }
  ~BuildScratch()
  {
    if(plain) { (void)hipFree(plain); } if(counts) { (void)hipFree(counts); }
    if(before) { (void)hipFree(before); } if(cub) { (void)hipFree(cub); }
  }
};

// RB64 form of one plain bit array (`plain` in host OR device memory) at its planned place in the image; fills p.ones.
hipError_t build_bv(u64* d_base, BVPlan& p, const u64* plain, BuildScratch& s)
{
  const u64 words = (p.size + 63) / 64;
  hipError_t e = hipSuccess;
  if(words > 0) { e = hipMemcpy(s.plain, plain, words * sizeof(u64), hipMemcpyDefault); }
  if(e != hipSuccess) { return e; }
  u64* blocks = d_base + p.blocks_off;
  hipLaunchKernelGGL(k_rb64_fill, dim3(grid_for(p.nblocks)), dim3(TPB), 0, nullptr, s.plain, p.size, p.nblocks, blocks, s.counts);
  size_t bytes = s.cub_bytes;
  e = hipcub::DeviceScan::ExclusiveSum(s.cub, bytes, s.counts, s.before, int(p.nblocks));
  if(e != hipSuccess) { return e; }
  hipLaunchKernelGGL(k_rb64_counts, dim3(grid_for(p.nblocks)), dim3(TPB), 0, nullptr, s.before, p.nblocks, blocks);
  u64 last[2] = { 0, 0 };
  e = hipMemcpy(&last[0], s.before + (p.nblocks - 1), sizeof(u64), hipMemcpyDeviceToHost);
  if(e == hipSuccess) { e = hipMemcpy(&last[1], s.counts + (p.nblocks - 1), sizeof(u64), hipMemcpyDeviceToHost); }
  if(e != hipSuccess) { return e; }
  p.ones = last[0] + last[1];
  if(p.hints)
  {
    hipLaunchKernelGGL(k_select_hints, dim3(grid_for(p.nhints)), dim3(TPB), 0, nullptr, blocks, p.nblocks, p.nhints,
                       reinterpret_cast<u32*>(d_base + p.hints_off));
  }
  return hipGetLastError();
}

DevBV resolve(const BVPlan& p, const u64* d_base)
{
  DevBV bv;
  bv.blocks = d_base + p.blocks_off;
  bv.hints = p.hints ? reinterpret_cast<const u32*>(d_base + p.hints_off) : nullptr;
  bv.size = p.size; bv.nblocks = p.nblocks; bv.ones = p.ones;
  return bv;
}

struct DeviceGuard
{
  int prev = -1;
  bool ok = false;
  explicit DeviceGuard(int dev)
  {
    if(hipGetDevice(&prev) != hipSuccess) { prev = -1; }
    ok = (hipSetDevice(dev) == hipSuccess);
  }
  ~DeviceGuard() { if(prev >= 0) { (void)hipSetDevice(prev); } }
};

// RAII device buffer for the host-pointer entry points
template<class T> struct DBuf
{
  T* p = nullptr;
  hipError_t alloc(u64 count) { return hipMalloc(reinterpret_cast<void**>(&p), (count > 0 ? count : 1) * sizeof(T)); }
  ~DBuf() { if(p) { (void)hipFree(p); } }
};

constexpr size_t PINNED_ARENA = size_t(8) << 20;       // per staging object
constexpr size_t PINNED_MAX_COPY = size_t(2) << 20;    // larger transfers go straight from / to the caller's memory
constexpr size_t DEVICE_ARENA_KEEP = size_t(512) << 20; // a larger arena is released with the call that needed it
// A call that moves at most this much (the scalar find() / LF() / count() / parent() of the facade, batches of a few hundred
// queries) makes no copies at all: its buffers are carved out of pinned host memory that the kernel reads and writes over the
// bus, so the call is one launch and one stream synchronisation instead of two or three copies around them.
constexpr size_t ZERO_COPY_ARENA = size_t(32) << 10;

// One host-pointer call's lease on a Staging object of the handle (see Staging).
class Lease
{
public:
  explicit Lease(const gcsa2_index* ix) : ix(ix), s(nullptr), d_used(0), h_used(0), busy(false), zero(false)
  {
    std::lock_guard<std::mutex> hold(ix->staging_lock);
    if(!ix->staging_pool.empty()) { s = ix->staging_pool.back(); ix->staging_pool.pop_back(); }
  }
  ~Lease()
  {
    if(s == nullptr) { return; }
    if(busy) { (void)hipStreamSynchronize(s->stream); }       // an early error return: nothing of this call may still be in flight
    if(s->d_cap > DEVICE_ARENA_KEEP) { (void)hipFree(s->d); s->d = nullptr; s->d_cap = 0; }
    std::lock_guard<std::mutex> hold(ix->staging_lock);
    ix->staging_pool.push_back(s);
  }
  Lease(const Lease&) = delete;
  Lease& operator=(const Lease&) = delete;

  // device bytes this call will carve out of the arena (sum of its buffers, each rounded up to 256 bytes)
  // zero_copy_ok: every buffer of the call is only ever touched by kernels and by up() / down()
  hipError_t begin(size_t device_bytes, bool zero_copy_ok = false)
  {
    if(s == nullptr)
    {
      s = new(std::nothrow) Staging();
      if(s == nullptr) { return hipErrorOutOfMemory; }
      hipError_t e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
      if(e == hipSuccess) { e = hipHostMalloc(reinterpret_cast<void**>(&s->h), PINNED_ARENA, hipHostMallocDefault); }
      if(e == hipSuccess) { e = hipHostMalloc(reinterpret_cast<void**>(&s->z), ZERO_COPY_ARENA + 64, hipHostMallocMapped | hipHostMallocCoherent); }
      if(e != hipSuccess)
      {
        if(s->stream) { (void)hipStreamDestroy(s->stream); } if(s->h) { (void)hipHostFree(s->h); }
        delete s; s = nullptr; return e;
      }
      s->h_cap = PINNED_ARENA;
    }
    zero = (zero_copy_ok && ix->tune.zero_copy && device_bytes <= ZERO_COPY_ARENA);
    if(zero) { d_used = 0; h_used = 0; pending.clear(); busy = true; return hipSuccess; }
    if(device_bytes > s->d_cap)
    {
      if(s->d != nullptr) { (void)hipFree(s->d); s->d = nullptr; s->d_cap = 0; }
      size_t want = device_bytes + (device_bytes >> 2) + 4096;
      hipError_t e = hipMalloc(reinterpret_cast<void**>(&s->d), want);
      if(e != hipSuccess) { return e; }
      s->d_cap = want;
    }
    d_used = 0; h_used = 0; pending.clear(); busy = true;
    return hipSuccess;
  }
  static size_t need(size_t bytes) { return (bytes + 255) / 256 * 256 + 256; }
  template<class T> T* dev(u64 count)
  {
    char* p = (zero ? s->z : s->d) + d_used;
    d_used += need(count * sizeof(T));
    return reinterpret_cast<T*>(p);
  }
  hipStream_t stream() const { return s->stream; }

  hipError_t up(void* d, const void* h, size_t bytes)
  {
    if(bytes == 0) { return hipSuccess; }
    if(in_zero_arena(d)) { std::memcpy(d, h, bytes); return hipSuccess; }
    char* slot = pinned(bytes);
    if(slot == nullptr) { return hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s->stream); }
    std::memcpy(slot, h, bytes);
    return hipMemcpyAsync(d, slot, bytes, hipMemcpyHostToDevice, s->stream);
  }
  hipError_t down(void* h, const void* d, size_t bytes)
  {
    if(bytes == 0) { return hipSuccess; }
    if(in_zero_arena(d)) { pending.push_back(Pending{h, static_cast<const char*>(d), bytes}); return hipSuccess; }
    char* slot = pinned(bytes);
    if(slot == nullptr) { return hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s->stream); }
    pending.push_back(Pending{h, slot, bytes});
    return hipMemcpyAsync(slot, d, bytes, hipMemcpyDeviceToHost, s->stream);
  }
  // wait for everything enqueued on the call's stream and hand the staged results to the caller.  A zero-copy call (the
  // facade's scalar find() / LF() / count() / parent()) waits for a ticket that a one-thread kernel behind its work writes
  // into the page-locked arena, instead of for hipStreamSynchronize: 8 against 13 us on this path (tests/perf/launch_latency.hip)
  hipError_t finish()
  {
    hipError_t e = hipSuccess;
    if(zero && ix->tune.poll_small)
    {
      volatile unsigned long long* flag = reinterpret_cast<volatile unsigned long long*>(s->z + ZERO_COPY_ARENA);
      const unsigned long long ticket = ix->next_ticket.fetch_add(1);
      hipLaunchKernelGGL(k_ticket, dim3(1), dim3(1), 0, s->stream, flag, ticket);
      e = hipGetLastError();
      for(u64 spins = 1; e == hipSuccess && *flag != ticket; spins++)
      {
        if(spins > 100000) { std::this_thread::yield(); }
        if((spins & 0x3FFF) == 0)
        {
          const hipError_t q = hipStreamQuery(s->stream);
          if(q == hipSuccess) { break; }                        // idle: the ticket is there (or the launch was lost: checked below)
          if(q != hipErrorNotReady) { e = q; }
        }
      }
      if(e == hipSuccess && *flag != ticket) { e = hipStreamSynchronize(s->stream); }
      std::atomic_thread_fence(std::memory_order_acquire);
    }
    else { e = hipStreamSynchronize(s->stream); }
    if(e == hipSuccess) { for(const Pending& p : pending) { std::memcpy(p.user, p.slot, p.bytes); } }
    pending.clear(); busy = false;
    return e;
  }

private:
  bool in_zero_arena(const void* p) const
  {
    const char* c = static_cast<const char*>(p);
    return zero && c >= s->z && c < s->z + ZERO_COPY_ARENA;
  }
  char* pinned(size_t bytes)
  {
    if(bytes > PINNED_MAX_COPY || h_used + bytes > s->h_cap) { return nullptr; }
    char* p = s->h + h_used;
    h_used += (bytes + 63) / 64 * 64;
    return p;
  }
  struct Pending { void* user; const char* slot; size_t bytes; };
  const gcsa2_index* ix; Staging* s; size_t d_used, h_used; bool busy, zero;
  std::vector<Pending> pending;
};

inline hipError_t pool_alloc(const gcsa2_index* ix, void** p, size_t bytes, hipStream_t stream)
{
  return ix->pool != nullptr ? hipMallocFromPoolAsync(p, bytes, ix->pool, stream) : hipMallocAsync(p, bytes, stream);
}

// stream-ordered scratch: no device-wide synchronisation from allocation or release
// Scratch of one locate pass: carved out of an ARENA the handle keeps (one per concurrent call; grown to the largest pass seen,
// released by gcsa2_index_trim).  Round 3 took every buffer from the stream-ordered pool: fifteen hipFreeAsync calls at the end
// of a pass cost 1.0-1.4 ms of host time -- a quarter of a 4.4 ms batch (GCSA2_LOCATE_TRACE, profiles/r04_locate.md).
// A buffer that does not fit the arena is a plain allocation for this pass; the arena is re-made at the new size when the pass
// ends.  The arena goes back to the handle only when the stream is known to be idle (`settled`, or a synchronisation here).
struct Scratch
{
  const gcsa2_index* ix; hipStream_t stream;
  gcsa2_index::Arena arena;
  size_t used = 0, wanted = 0;
  std::vector<void*> extra;
  bool settled = false;
  Scratch(const gcsa2_index* ix, hipStream_t s) : ix(ix), stream(s)
  {
    std::lock_guard<std::mutex> guard(ix->arena_lock);
    size_t best = 0;
    for(size_t i = 1; i < ix->arenas.size(); i++) { if(ix->arenas[i].bytes > ix->arenas[best].bytes) { best = i; } }
    if(!ix->arenas.empty()) { arena = ix->arenas[best]; ix->arenas.erase(ix->arenas.begin() + long(best)); }
  }
  ~Scratch()
  {
    if(!settled) { (void)hipStreamSynchronize(stream); }
    for(void* p : extra) { (void)hipFree(p); }
    // the arena grows to 1.125 x the largest request, up to tune.arena_cap (GCSA2_ARENA_CAP_MB, default 24 GB): what a pass
    // needs beyond it is allocated and freed by that pass, so that one huge locate() does not pin tens of GB on the handle
    // until gcsa2_index_trim
    size_t want = wanted + wanted / 8;
    if(want > ix->tune.arena_cap) { want = ix->tune.arena_cap; }
    if(want > arena.bytes)
    {
      if(arena.base != nullptr) { (void)hipFree(arena.base); }
      arena.base = nullptr; arena.bytes = 0;
      void* fresh = nullptr;
      if(hipMalloc(&fresh, want) == hipSuccess) { arena.base = static_cast<char*>(fresh); arena.bytes = want; }
      else { (void)hipGetLastError(); }
    }
    if(arena.base != nullptr)
    {
      std::lock_guard<std::mutex> guard(ix->arena_lock);
      ix->arenas.push_back(arena);
    }
  }
  template<class T> hipError_t get(T*& p, u64 count)
  {
    const size_t bytes = (size_t(count > 0 ? count : 1) * sizeof(T) + 255) & ~size_t(255);
    wanted += bytes;
    if(arena.base != nullptr && used + bytes <= arena.bytes) { p = reinterpret_cast<T*>(arena.base + used); used += bytes; return hipSuccess; }
    void* raw = nullptr;
    hipError_t e = hipMalloc(&raw, bytes);
    if(e == hipSuccess) { extra.push_back(raw); p = static_cast<T*>(raw); }
    return e;
  }
};

}  // namespace

namespace {

// ---- the mailbox: scalar calls answered by a resident wavefront (kernels_mailbox.hpp) ---------------------------------------
constexpr int MAIL_UNAVAILABLE = 1;          // internal, not a gcsa2_status: take the launch path
constexpr unsigned long long MAIL_TICKS_PER_US = 100;      // wall_clock64(): the constant 100 MHz counter

inline void mailbox_launch(const gcsa2_index* ix, unsigned long long answered)
{
  gcsa2_index::Mailbox& m = ix->mail;
  volatile MailSlot* s = m.slot;
  s->alive = 1;
  std::atomic_thread_fence(std::memory_order_seq_cst);
  hipLaunchKernelGGL(k_mailbox, dim3(1), dim3(64), 0, m.stream, ix->img, m.slot, answered, ix->tune.mailbox_park_us * MAIL_TICKS_PER_US,
                     ix->tune.mailbox_life_ms * 1000 * MAIL_TICKS_PER_US);
  m.launches++;
}

// One request through the slot.  GCSA2_OK with the results, MAIL_UNAVAILABLE when the mailbox is switched off, held by another
// host thread (that thread's loop keeps the wavefront; this call takes the launch path instead of queueing behind it) or could
// not be set up; an error when the device does not answer.
int mailbox_call(const gcsa2_index* ix, unsigned long long op, unsigned long long a0, unsigned long long a1, unsigned long long a2, u64* results, int n_results)
{
  gcsa2_index::Mailbox& m = ix->mail;
  if(!ix->tune.mailbox || m.failed) { return MAIL_UNAVAILABLE; }
  std::unique_lock<std::mutex> hold(m.lock, std::try_to_lock);
  if(!hold.owns_lock()) { return MAIL_UNAVAILABLE; }
  if(m.slot == nullptr)
  {
    void* raw = nullptr;
    if(hipHostMalloc(&raw, sizeof(MailSlot), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) { (void)hipGetLastError(); m.failed = true; return MAIL_UNAVAILABLE; }
    std::memset(raw, 0, sizeof(MailSlot));
    if(hipStreamCreateWithFlags(&m.stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); (void)hipHostFree(raw); m.failed = true; return MAIL_UNAVAILABLE; }
    m.slot = static_cast<MailSlot*>(raw);
  }
  volatile MailSlot* s = m.slot;
  const unsigned long long ticket = ++m.ticket;
  m.calls++;
  s->op = op; s->arg[0] = a0; s->arg[1] = a1; s->arg[2] = a2;
  std::atomic_thread_fence(std::memory_order_release);
  if(s->alive == 0) { mailbox_launch(ix, ticket - 1); if(hipGetLastError() != hipSuccess) { m.failed = true; return MAIL_UNAVAILABLE; } }
  s->request = ticket;
  const auto t0 = std::chrono::steady_clock::now();
  for(u64 spins = 1; s->done != ticket || s->front != ticket; spins++)       // (both ends of the answer's line: kernels_mailbox.hpp)
  {
    __builtin_ia32_pause();
    if(s->alive == 0)
    {
      // the wavefront has left -- perhaps after answering in its last look at the slot; if not, the next instance answers
      std::atomic_thread_fence(std::memory_order_acquire);
      if(s->done == ticket && s->front == ticket) { break; }
      mailbox_launch(ix, ticket - 1);
      if(hipGetLastError() != hipSuccess) { m.failed = true; return fail(GCSA2_ERR_HIP, "mailbox: the resident kernel could not be launched"); }
    }
    if((spins & 0xFFFF) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(10))
    {
      m.failed = true;
      return fail(GCSA2_ERR_HIP, "mailbox: the device did not answer a scalar call within 10 s");
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  for(int k = 0; k < n_results; k++) { results[k] = s->result[k]; }
  return GCSA2_OK;
}

// ends the resident wavefront (before the image changes or goes away) and, with `release`, gives the slot and the stream back
void mailbox_stop(const gcsa2_index* ix, bool release)
{
  gcsa2_index::Mailbox& m = ix->mail;
  std::lock_guard<std::mutex> hold(m.lock);
  if(m.slot == nullptr) { return; }
  volatile MailSlot* s = m.slot;
  if(s->alive != 0 && !m.failed)
  {
    const unsigned long long ticket = ++m.ticket;
    s->op = MAIL_QUIT;
    std::atomic_thread_fence(std::memory_order_release);
    s->request = ticket;
  }
  (void)hipStreamSynchronize(m.stream);                       // (an instance that had left already: nothing to wait for)
  s->alive = 0; s->done = m.ticket; s->front = m.ticket;
  if(release)
  {
    (void)hipStreamDestroy(m.stream); (void)hipHostFree(m.slot);
    m.slot = nullptr; m.stream = nullptr;
  }
}

}  // namespace

namespace {
// owners: scratch of total_nodes / OWNER_SPAN + 3 entries (k_block_owners).  Values in path order (the table walk has an
// unordered two-pass form for the sorted mode: locate_chunk).
inline void launch_walk(const gcsa2_index* ix, const u64* d_ranges, u64 nq, const u64* node_off, const u64* raw_off,
                        u64 total_nodes, u64* values, u64* owners, hipStream_t stream)
{
  if(ix->img.locate_tab != nullptr)
  {
    const u64 blocks = grid_for(total_nodes), spans = (total_nodes + OWNER_SPAN - 1) / OWNER_SPAN;
    hipLaunchKernelGGL(k_block_owners, dim3(grid_for(spans + 1)), dim3(TPB), 0, stream, node_off, nq, total_nodes, OWNER_SPAN, spans, owners);
    hipLaunchKernelGGL(k_locate_tab, dim3(unsigned(blocks)), dim3(TPB), 0, stream,
                       ix->img, d_ranges, nq, node_off, raw_off, total_nodes, values, owners);
  }
  else if(ix->img.pred4 != nullptr)
  {
    const u64 blocks = (total_nodes + TPB2 - 1) / TPB2, spans = (total_nodes + OWNER_SPAN - 1) / OWNER_SPAN;
    hipLaunchKernelGGL(k_block_owners, dim3(grid_for(spans + 1)), dim3(TPB), 0, stream, node_off, nq, total_nodes, OWNER_SPAN, spans, owners);
    hipLaunchKernelGGL(k_locate_walk2, dim3(unsigned(blocks)), dim3(TPB2), 0, stream,
                       ix->img, d_ranges, nq, node_off, raw_off, total_nodes, values, owners);
  }
  else
  {
    hipLaunchKernelGGL(k_locate_walk, dim3(grid_for(total_nodes)), dim3(TPB), 0, stream,
                       ix->img, d_ranges, nq, node_off, raw_off, total_nodes, values);
  }
}
}  // namespace

extern "C" {

const char* gcsa2_last_error(void) { return g_error.c_str(); }

int gcsa2_device_count(void)
{
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if(e != hipSuccess) { return fail(GCSA2_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); }
  return count;
}

int gcsa2_index_create(const gcsa2_host_view* v, int device, gcsa2_index** out)
{
  if(v == nullptr || out == nullptr) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "null argument"); }
  *out = nullptr;
  if(v->sigma == 0 || v->sigma > GCSA2_MAX_SIGMA || v->char2comp == nullptr || v->C == nullptr ||
     v->bwt == nullptr || v->edge_bits == nullptr)
  {
    return fail(GCSA2_ERR_INVALID_ARGUMENT, "host view lacks alphabet / bwt / edges or sigma out of range");
  }
  if(v->lcp_data != nullptr && v->lcp_levels > u64(MAX_LCP_LEVELS)) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "too many LCP levels"); }
  // Consistency of the view itself: the query kernels index device memory with these values and, like the
  // reference's low-level interface (gcsa.h:133-135), do not check them again.
  for(int b = 0; b < 256; b++)
  {
    if(v->char2comp[b] >= v->sigma) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "alpha.char2comp maps a byte to a comp >= sigma"); }
  }
  for(u64 c = 0; c < v->sigma; c++)
  {
    if(v->C[c] > v->C[c + 1]) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "alpha.C is not non-decreasing"); }
  }
  if(v->C[v->sigma] != v->edges) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "alpha.C[sigma] differs from header.edges"); }
  if(v->sampled_path_bits != nullptr)
  {
    if(v->stored_samples == nullptr || v->sample_bits == nullptr || v->sample_width == 0 || v->sample_width > 64)
    {
      return fail(GCSA2_ERR_INVALID_ARGUMENT, "samples: missing array or sample width out of range");
    }
  }
  if(v->extra_filter_bits != nullptr && (v->extra_values_bits == nullptr || v->redundant_bits == nullptr))
  {
    return fail(GCSA2_ERR_INVALID_ARGUMENT, "counters: missing array");
  }
  if(v->lcp_data != nullptr)
  {
    // LCPArray (src/lcp.cpp:224-259): level 0 = one value per path node, every further level one value per
    // `branching` values of the level below, up to a single root.  A stale .lcp beside an index (another graph,
    // another order) would otherwise send parent() outside the index and hang the LF + parent loop.
    if(v->lcp_offsets == nullptr || v->lcp_levels == 0) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "LCP: missing offsets"); }
    if(v->lcp_branching < 2) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "LCP: branching factor below 2"); }
    if(v->path_nodes > 0 && v->lcp_size != v->path_nodes) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "LCP: header.size differs from the number of path nodes (an .lcp file of another index?)"); }
    u64 level_size = v->lcp_size, at = 0, top_size = 0;
    for(u64 l = 0; l < v->lcp_levels; l++)
    {
      if(v->lcp_offsets[l] != at) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "LCP: offsets do not match the level sizes implied by the branching factor"); }
      at += level_size; top_size = level_size;
      if(l + 1 < v->lcp_levels && level_size <= 1) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "LCP: levels above the root"); }
      level_size = (level_size + v->lcp_branching - 1) / v->lcp_branching;
    }
    if(v->lcp_offsets[v->lcp_levels] != at) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "LCP: offsets do not match the level sizes implied by the branching factor"); }
    if(top_size > 1) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "LCP: the top level is not a single root"); }
  }
  int count = gcsa2_device_count();
  if(count <= 0) { return fail(GCSA2_ERR_NO_DEVICE, "no HIP device visible: " + g_error); }
  if(device < 0 || device >= count) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "device index out of range"); }

  if(v->sigma * (v->path_nodes / FLB_BITS + 1) >= u64(PAIR_FLAG) || v->lcp_size / 16 + 16 >= u64(PAIR_FLAG))
  {
    return fail(GCSA2_ERR_INVALID_ARGUMENT, "index too large: the block and LCP-window indices of the fetch keep two flag bits");
  }
  if(v->path_nodes > MAX_PATH_NODES || v->edges > 2 * MAX_PATH_NODES)
  {
    return fail(GCSA2_ERR_INVALID_ARGUMENT, "index of more than 2^38 path nodes: beyond what one device holds and what the block indices of this build address");
  }
  gcsa2_index* ix = new(std::nothrow) gcsa2_index();
  if(ix == nullptr) { return fail(GCSA2_ERR_OUT_OF_MEMORY, "host allocation failed"); }
  ix->device = device; ix->order = v->order;
  {
    auto knob = [](const char* name, long fallback, long lo, long hi) -> long
    {
      const char* e = std::getenv(name);
      long val = (e != nullptr && *e != 0 ? std::atol(e) : fallback);
      return val < lo ? lo : (val > hi ? hi : val);
    };
    ix->tune.cool_down = u32(knob("GCSA2_COOL_DOWN", COOL_DOWN, 0, 1000));
    ix->tune.ms_refill_at = u32(knob("GCSA2_MS_REFILL_AT", MS_REFILL_AT, 1, 64));
    ix->tune.ms_grid = u64(knob("GCSA2_MS_GRID", 0, 0, long(1) << 30));
    ix->tune.sort_medium_limit = (knob("GCSA2_SORT_MEDIUM", 1, 0, 1) == 0 ? SMALL_SEGMENT : MEDIUM_SEGMENT);
    ix->tune.locate_split = u64(knob("GCSA2_LOCATE_SPLIT", (long(1) << 31) - 1, 2, (long(1) << 31) - 1));
    ix->tune.locate_split_queries = u64(knob("GCSA2_LOCATE_SPLIT_QUERIES", long(1) << 30, 1, long(1) << 30));
    ix->tune.zero_copy = (knob("GCSA2_ZERO_COPY", 1, 0, 1) != 0);
    ix->tune.poll_small = (knob("GCSA2_POLL_SMALL", 1, 0, 1) != 0);
    ix->tune.dedup_huge = (knob("GCSA2_DEDUP_HUGE", 1, 0, 1) != 0);
    ix->tune.dedup_narrow = (knob("GCSA2_DEDUP_NARROW", 1, 0, 1) != 0);
    ix->tune.pipe_lanes = u32(knob("GCSA2_PIPE_LANES", 6, 1, 16));
    ix->tune.pipe_chunk = u64(1) << knob("GCSA2_PIPE_CHUNK", 18, 15, 20);
    ix->tune.pipe_split = (knob("GCSA2_PIPE_SPLIT", 0, 0, 1) != 0);
    ix->tune.pipe_blocking = (knob("GCSA2_PIPE_BLOCKING", 0, 0, 1) != 0);
    ix->tune.pipe_wide = (knob("GCSA2_PIPE_WIRE", 0, 0, 16) == 16);
    ix->tune.ms_pieces = (knob("GCSA2_MS_PIECES", 1, 0, 1) != 0);
    ix->tune.ms_piece_bytes = u64(knob("GCSA2_MS_PIECE_MB", 32, 1, 1024)) << 20;
    ix->tune.ms_threads = u32(knob("GCSA2_MS_THREADS", 4, 1, 16));
    ix->tune.kmer_piece = u64(knob("GCSA2_KMER_PIECE", long(1) << 27, 4, long(1) << 27));
    ix->tune.locate_trace = (knob("GCSA2_LOCATE_TRACE", 0, 0, 1) != 0);
    ix->tune.locate_single = (knob("GCSA2_LOCATE_SINGLE", 1, 0, 1) != 0);
    ix->tune.locate_fused_compact = (knob("GCSA2_LOCATE_FUSED_COMPACT", 1, 0, 1) != 0);
    ix->tune.locate_fuse = (knob("GCSA2_LOCATE_FUSE", 1, 0, 1) != 0);
    ix->tune.fuse_above = u64(knob("GCSA2_LOCATE_FUSE_ABOVE", BIG_SEGMENT, 1, long(1) << 40));
    ix->tune.locate_in_place = (knob("GCSA2_LOCATE_IN_PLACE", 1, 0, 1) != 0);
    ix->tune.mailbox = (knob("GCSA2_MAILBOX", 1, 0, 1) != 0);
    ix->tune.mailbox_park_us = u64(knob("GCSA2_MAILBOX_PARK_US", 200, 1, 1000000));
    ix->tune.mailbox_life_ms = u64(knob("GCSA2_MAILBOX_LIFE_MS", 20, 1, 10000));
    ix->tune.split_tiled = (knob("GCSA2_SPLIT_TILED", 1, 0, 1) != 0);
    ix->tune.locate_split_sort = (knob("GCSA2_LOCATE_SPLIT_SORT", 1, 0, 1) != 0);
    ix->tune.split_skew = u32(knob("GCSA2_SPLIT_SKEW", BIG_SEGMENT, 16, BIG_SEGMENT));
    ix->tune.split_target = u32(knob("GCSA2_SPLIT_TARGET", SPLIT_TARGET, 1, 4096));
    ix->tune.arena_cap = size_t(knob("GCSA2_ARENA_CAP_MB", 24576, 0, long(1) << 20)) << 20;    // 24 GB: 1/12 of an MI355X's HBM per arena
    ix->tune.seed_wide = u32(knob("GCSA2_SEED_WIDE", long(SEED_WIDE), 2, long(SEED_WIDE)));     // tests: meet the marked seed entries
    {
      const char* b = std::getenv("GCSA2_MEMORY_BUDGET_MB");           // megabytes, fractions allowed (small test indexes)
      const double mb = (b != nullptr && *b != 0 ? std::atof(b) : 0.0);
      ix->tune.budget_bytes = (mb > 0.0 ? u64(mb * 1048576.0) : 0);
    }
  }
  std::memset(&ix->img, 0, sizeof(DevImage));
  DevImage& img = ix->img;
  img.n = v->path_nodes; img.e = v->edges; img.sigma = v->sigma; img.fast_chars = v->fast_chars;
  for(u64 c = 0; c <= v->sigma; c++) { img.C[c] = v->C[c]; }
  std::memcpy(img.char2comp, v->char2comp, 256);

  try
  {
    // Layout first.  All B_c back to back with identical geometry (k_find selects by base + comp * stride).
    Planner pl;
    std::vector<BVPlan> bwt(v->sigma);
    for(u64 c = 0; c < v->sigma; c++)
    {
      bwt[c] = plan_bv(pl, img.n, false);
      if(c > 0 && bwt[c].blocks_off != bwt[0].blocks_off + c * (bwt[0].nblocks * BLOCK_WORDS))
      {
        delete ix; return fail(GCSA2_ERR_INVALID_ARGUMENT, "internal: B_c stride mismatch");
      }
    }
    BVPlan edges = plan_bv(pl, img.e, false);
    img.flb_nblocks = img.n / FLB_BITS + 1;
    const u64 flb_off = pl.reserve(v->sigma * img.flb_nblocks * FLB_WORDS, FLB_WORDS);
    BVPlan sampled, samples, xfilter, xvalues, redundant;
    u64 stored_off = 0, lcp_off = 0, stored_words = 0, max_bits = (img.n > img.e ? img.n : img.e);
    img.has_samples = (v->sampled_path_bits != nullptr);
    if(img.has_samples)
    {
      sampled = plan_bv(pl, img.n, false);
      samples = plan_bv(pl, v->sample_count, true);
      stored_words = (v->sample_count * v->sample_width + 63) / 64;
      stored_off = pl.reserve(stored_words + 2);
      img.sample_count = v->sample_count; img.sample_width = v->sample_width;
      if(v->sample_count > max_bits) { max_bits = v->sample_count; }
    }
    img.has_counters = (v->extra_filter_bits != nullptr);
    if(img.has_counters)
    {
      xfilter = plan_bv(pl, img.n, false);
      xvalues = plan_bv(pl, v->extra_values_len, true);
      redundant = plan_bv(pl, v->redundant_len, true);
      if(v->extra_values_len > max_bits) { max_bits = v->extra_values_len; }
      if(v->redundant_len > max_bits) { max_bits = v->redundant_len; }
    }
    img.has_lcp = (v->lcp_data != nullptr);
    if(img.has_lcp)
    {
      img.lcp_size = v->lcp_size; img.lcp_branching = v->lcp_branching; img.lcp_levels = v->lcp_levels;
      img.lcp_shift = 0;
      if((v->lcp_branching & (v->lcp_branching - 1)) == 0) { while((u64(1) << img.lcp_shift) < v->lcp_branching) { img.lcp_shift++; } }
      for(u64 l = 0; l <= v->lcp_levels; l++) { img.lcp_offsets[l] = v->lcp_offsets[l]; }
      img.lcp_values = v->lcp_offsets[v->lcp_levels];
      lcp_off = pl.reserve((img.lcp_values + 7) / 8 + 18);  // 128-byte window reads around the last values stay inside (parent_from_window)
    }

    DeviceGuard guard(device);
    if(!guard.ok) { delete ix; return fail(GCSA2_ERR_HIP, "hipSetDevice failed"); }
    {
      int cus = 0;
      if(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) { ix->compute_units = cus; }
      // Query scratch is stream-ordered memory from a pool OWNED BY THIS HANDLE, which keeps what it has grown to
      // instead of returning it to the driver at every synchronisation.  The device's default pool (and with it
      // the caller's own hipMallocAsync behaviour) is left alone.
      hipMemPoolProps props;
      std::memset(&props, 0, sizeof(props));
      props.allocType = hipMemAllocationTypePinned;
      props.handleTypes = hipMemHandleTypeNone;
      props.location.type = hipMemLocationTypeDevice;
      props.location.id = device;
      if(hipMemPoolCreate(&ix->pool, &props) == hipSuccess && ix->pool != nullptr)
      {
        uint64_t keep = ~uint64_t(0);
        (void)hipMemPoolSetAttribute(ix->pool, hipMemPoolAttrReleaseThreshold, &keep);
      }
      else { ix->pool = nullptr; }           // falls back to the default pool with its default threshold
      (void)hipGetLastError();
    }
    ix->bytes = pl.words * sizeof(u64);
    hipError_t e = hipMalloc(&ix->d_base, ix->bytes > 0 ? ix->bytes : 8);
    if(e != hipSuccess) { delete ix; return fail(GCSA2_ERR_OUT_OF_MEMORY, std::string("hipMalloc(image): ") + hipGetErrorString(e)); }
    e = hipMemset(ix->d_base, 0, ix->bytes > 0 ? ix->bytes : 8);       // alignment gaps, the words behind the samples and the LCP values
    if(e == hipSuccess) { e = hipMalloc(reinterpret_cast<void**>(&ix->d_slots), RESULT_SLOTS * TOTAL_WORDS * sizeof(unsigned long long)); }
    if(e == hipSuccess) { e = hipHostMalloc(reinterpret_cast<void**>(&ix->h_slots), RESULT_SLOTS * TOTAL_WORDS * sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocCoherent); }
    if(e == hipSuccess) { std::memset(ix->h_slots, 0, RESULT_SLOTS * TOTAL_WORDS * sizeof(unsigned long long)); }
    if(e != hipSuccess) { gcsa2_index_destroy(ix); return fail(GCSA2_ERR_HIP, std::string("image allocation: ") + hipGetErrorString(e)); }

    // The image is built on the device from the plain arrays of the view, each of them copied once (they may already be in
    // device memory): rank blocks, select hints, the fused FLB128 blocks and the charRange table (kernels_build.hpp).
    u64* wbase = static_cast<u64*>(ix->d_base);
    const u64* base = wbase;
    {
      BuildScratch scratch;
      const char* step = "scratch";
      auto run = [&](const char* name, BVPlan& plan, const u64* plain, DevBV& dst)
      {
        if(e != hipSuccess) { return; }
        step = name;
        e = build_bv(wbase, plan, plain, scratch);
        dst = resolve(plan, base);
      };
      auto copy = [&](const char* name, u64 word_off, const void* src, u64 bytes)
      {
        if(e != hipSuccess || bytes == 0) { return; }
        step = name;
        e = hipMemcpy(wbase + word_off, src, bytes, hipMemcpyDefault);
      };
      e = scratch.alloc(max_bits);
      for(u64 c = 0; c < v->sigma; c++) { run("bwt", bwt[c], v->bwt[c], img.bwt[c]); }
      run("edges", edges, v->edge_bits, img.edges);
      if(e == hipSuccess)
      {
        step = "fused blocks";
        hipLaunchKernelGGL(k_build_flb, dim3(grid_for(img.flb_nblocks), unsigned(v->sigma)), dim3(TPB), 0, nullptr, img, wbase + flb_off);
        hipLaunchKernelGGL(k_crange, dim3(1), dim3(64), 0, nullptr, img, scratch.before);
        e = hipGetLastError();
        if(e == hipSuccess) { e = hipMemcpy(img.crange, scratch.before, 2 * v->sigma * sizeof(u64), hipMemcpyDeviceToHost); }
        img.flb = base + flb_off;
      }
      if(img.has_samples)
      {
        run("sampled_paths", sampled, v->sampled_path_bits, img.sampled);
        run("samples", samples, v->sample_bits, img.samples);
        copy("stored_samples", stored_off, v->stored_samples, stored_words * sizeof(u64));
        img.stored = base + stored_off;
      }
      if(img.has_counters)
      {
        run("extra_pointers.filter", xfilter, v->extra_filter_bits, img.xfilter);
        run("extra_pointers.values", xvalues, v->extra_values_bits, img.xvalues);
        run("redundant_pointers", redundant, v->redundant_bits, img.redundant);
      }
      if(img.has_lcp)
      {
        copy("lcp", lcp_off, v->lcp_data, img.lcp_values);
        img.lcp = reinterpret_cast<const u8*>(base + lcp_off);
      }
      if(e == hipSuccess) { step = "synchronize"; e = hipDeviceSynchronize(); }
      if(e != hipSuccess) { g_error = step; }
    }
    if(e != hipSuccess)
    {
      const std::string step = g_error;
      gcsa2_index_destroy(ix);
      return fail(GCSA2_ERR_HIP, "image build (" + step + "): " + hipGetErrorString(e));
    }

    // pred4 nibbles (first predecessor comp + sampled flag), built on the device
    img.pred4 = nullptr;
    if(img.sigma <= 8 && img.n > 0)
    {
      u64 nwords = (img.n / BLOCK_BITS + 1) * PAYLOAD_WORDS;
      e = hipMalloc(&ix->d_pred4, nwords * 4 * sizeof(u64));
      if(e == hipSuccess)
      {
        hipLaunchKernelGGL(k_build_pred4, dim3(grid_for(nwords)), dim3(TPB), 0, nullptr, img, nwords, static_cast<u64*>(ix->d_pred4));
        e = hipGetLastError();
        if(e == hipSuccess) { e = hipDeviceSynchronize(); }
      }
      if(e != hipSuccess)
      {
        gcsa2_index_destroy(ix);       // releases whatever has been allocated so far
        return fail(GCSA2_ERR_HIP, std::string("pred4: ") + hipGetErrorString(e));
      }
      img.pred4 = static_cast<const u64*>(ix->d_pred4);
      ix->bytes += nwords * 4 * sizeof(u64);
    }

    // FLP128 pair blocks (two characters per step), built on the device: 10.7 bytes per path node.  Default on;
    // skipped when GCSA2_PAIR_BLOCKS=0, when comps 1..4 do not exist, or when they would take more than a third
    // of the free device memory (find() then steps one character at a time, with identical results).
    img.flp = nullptr; img.flp_nblocks = 0;
    {
      const char* penv = std::getenv("GCSA2_PAIR_BLOCKS");
      const u64 pair_bytes = pair_block_bytes_of(img);
      size_t free_bytes = 0, total_bytes = 0;
      bool fits = hipMemGetInfo(&free_bytes, &total_bytes) == hipSuccess && pair_bytes <= free_bytes / 3;
      if(fits && ix->tune.budget_bytes > 0)
      {
        // Under a memory budget (GCSA2_MEMORY_BUDGET_MB) the tables are taken in the order of what they buy find() per byte:
        // a seed table of up to an eighth of what the budget leaves (every character it covers removes one LF step per query
        // for 4x the bytes), then the pair blocks (half the memory requests of the remaining steps for 10.7 bytes per path
        // node), then the seed table grown into what is left, then the locate table (8 bytes per path node; locate() only).
        const u64 left = budget_left(ix);
        u64 reserve = 0;
        for(u32 k0 = 1; k0 <= 16 && (u64(8) << (2 * k0)) <= left / 8; k0++) { reserve = u64(8) << (2 * k0); }
        fits = pair_bytes + reserve <= left;
      }
      if(!(penv != nullptr && std::atoi(penv) == 0) && pair_blocks_possible(img) && fits)
      {
        e = make_pair_blocks(ix);
        if(e != hipSuccess)
        {
          gcsa2_index_destroy(ix);
          return fail(GCSA2_ERR_HIP, std::string("pair blocks: ") + hipGetErrorString(e));
        }
      }
    }

    // k-mer seed table, only if comps 1..4 exist.  Default: the largest k <= 16 whose table (4^k entries of 8 bytes)
    // is at most twice the index image (with the pair blocks) -- each extra character saves one LF step per query
    // and quadruples the table; HBM capacity is what this GPU has to spare.  GCSA2_KMER_TABLE=k asks for exactly
    // k (<= 16; 0 disables).  Either way the table must fit in a quarter of the free device memory.
    u32 k = 0;
    const u64 entry_bytes = 8;
    const char* env = std::getenv("GCSA2_KMER_TABLE");
    {
      size_t free_bytes = 0, total_bytes = 0;
      if(hipMemGetInfo(&free_bytes, &total_bytes) != hipSuccess) { free_bytes = 0; }
      if(env != nullptr)
      {
        k = u32(std::atoi(env));
        if(k > 16) { k = 16; }
        while(k > 0 && (entry_bytes << (2 * k)) > free_bytes / 4) { k--; }
      }
      else
      {
        const u64 image_bytes = 2 * ix->bytes;
        while(k < 16 && (entry_bytes << (2 * (k + 1))) <= image_bytes && (entry_bytes << (2 * (k + 1))) <= free_bytes / 4) { k++; }
      }
      if(ix->tune.budget_bytes > 0) { while(k > 0 && (entry_bytes << (2 * k)) > budget_left(ix)) { k--; } }
    }
    if(!seed_table_possible(img)) { k = 0; }
    img.kmer_k = 0; img.kmer_table = nullptr; img.seed_wide = ix->tune.seed_wide;
    if(k > 0)
    {
      e = make_seed_table(ix, k);
      if(e != hipSuccess)
      {
        gcsa2_index_destroy(ix);
        return fail(GCSA2_ERR_HIP, std::string("k-mer table: ") + hipGetErrorString(e));
      }
    }
  }
  catch(const std::bad_alloc&)
  {
    gcsa2_index_destroy(ix); return fail(GCSA2_ERR_OUT_OF_MEMORY, "host allocation failed");
  }

  DeviceGuard table_guard(device);       // the guard above ended with the try block
  if(!table_guard.ok) { gcsa2_index_destroy(ix); return fail(GCSA2_ERR_HIP, "hipSetDevice failed"); }

  // memoised locate walks: 8 bytes per path node, built with the walk kernel itself.  Optional: skipped
  // when GCSA2_LOCATE_TABLE=0, when it would take more than a third of the free memory, or when an
  // entry does not fit (then locate() walks as before).
  ix->img.locate_tab = nullptr;
  {
    const char* env = std::getenv("GCSA2_LOCATE_TABLE");
    size_t free_bytes = 0, total_bytes = 0;
    bool wanted = locate_table_possible(ix->img) && !(env != nullptr && std::atoi(env) == 0);
    if(wanted && ix->tune.budget_bytes > 0 && ix->img.n * sizeof(u64) > budget_left(ix)) { wanted = false; }
    if(wanted && hipMemGetInfo(&free_bytes, &total_bytes) == hipSuccess && ix->img.n * sizeof(u64) <= free_bytes / 3)
    {
      (void)make_locate_table(ix);
    }
  }
  // memoised unary LF chains for find(): 16 bytes per path node, opt-in (GCSA2_JUMP_TABLE=1).  Built by
  // doubling (1 -> 2 -> 4 -> 8 steps) with a second buffer that is released afterwards.
  ix->img.jump_tab = nullptr;
  {
    const char* env = std::getenv("GCSA2_JUMP_TABLE");
    size_t free_bytes = 0, total_bytes = 0;
    const u64 n = ix->img.n, bytes = n * sizeof(ulonglong2);
    if(env != nullptr && std::atoi(env) != 0 && n > 0 && n <= JUMP_NODE_MASK && ix->img.sigma >= 5 &&
       (ix->tune.budget_bytes == 0 || bytes <= budget_left(ix)) &&
       hipMemGetInfo(&free_bytes, &total_bytes) == hipSuccess && 2 * bytes <= free_bytes / 2)
    {
      void* other = nullptr;
      hipError_t e = hipMalloc(&ix->d_jump, bytes);
      if(e == hipSuccess) { e = hipMalloc(&other, bytes); }
      ulonglong2 *a = static_cast<ulonglong2*>(ix->d_jump), *b = static_cast<ulonglong2*>(other);
      const u64 slice = u64(1) << 30;
      for(u64 first = 0; first < n && e == hipSuccess; first += slice)
      {
        u64 count = (n - first < slice ? n - first : slice);
        hipLaunchKernelGGL(k_jump_init, dim3(grid_for(count)), dim3(TPB), 0, nullptr, ix->img, first, a);
        e = hipGetLastError();
      }
      for(u32 have = 1; have < JUMP_MAX && e == hipSuccess; have *= 2)
      {
        for(u64 first = 0; first < n && e == hipSuccess; first += slice)
        {
          u64 count = (n - first < slice ? n - first : slice);
          hipLaunchKernelGGL(k_jump_double, dim3(grid_for(count)), dim3(TPB), 0, nullptr, a, n, first, have, b);
          e = hipGetLastError();
        }
        std::swap(a, b);
      }
      if(e == hipSuccess) { e = hipDeviceSynchronize(); }
      // three doubling rounds: the result is in the buffer that was `other` at the start
      if(e == hipSuccess)
      {
        (void)hipFree(b); ix->d_jump = a;
        ix->img.jump_tab = a; ix->bytes += bytes;
      }
      else
      {
        if(ix->d_jump) { (void)hipFree(ix->d_jump); } if(other) { (void)hipFree(other); }
        ix->d_jump = nullptr; (void)hipGetLastError();
      }
    }
  }
  *out = ix;
  return GCSA2_OK;
}

int gcsa2_index_set_tables(gcsa2_index* ix, int pair_blocks, int kmer_k, int locate_table)
{
  if(ix == nullptr || pair_blocks < -1 || pair_blocks > 1 || locate_table < -1 || locate_table > 1 || kmer_k < -1 || kmer_k > 16)
  {
    return fail(GCSA2_ERR_INVALID_ARGUMENT, "set_tables: pair_blocks / locate_table in {-1, 0, 1}, kmer_k in -1 .. 16");
  }
  DeviceGuard guard(ix->device);
  if(!guard.ok) { return fail(GCSA2_ERR_HIP, "hipSetDevice failed"); }
  mailbox_stop(ix, false);                   // (so does the resident wavefront of the scalar calls)
  HIP_TRY(hipDeviceSynchronize());           // launches in flight hold the old pointers
  // drops first, so that what is built next finds the memory
  if(pair_blocks == 0) { drop_pair_blocks(ix); }
  if(locate_table == 0) { drop_locate_table(ix); }
  if(kmer_k >= 0 && u32(kmer_k) != ix->img.kmer_k) { drop_seed_table(ix); }
  if(kmer_k > 0 && ix->img.kmer_k == 0)
  {
    if(!seed_table_possible(ix->img)) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "this index cannot have a seed table (no comps 1..4, or positions beyond 40 bits)"); }
    hipError_t e = make_seed_table(ix, u32(kmer_k));
    if(e != hipSuccess) { return fail(e == hipErrorOutOfMemory ? GCSA2_ERR_OUT_OF_MEMORY : GCSA2_ERR_HIP, std::string("k-mer table: ") + hipGetErrorString(e)); }
  }
  if(pair_blocks == 1 && ix->img.flp == nullptr)
  {
    if(!pair_blocks_possible(ix->img)) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "this index cannot have pair blocks"); }
    hipError_t e = make_pair_blocks(ix);
    if(e != hipSuccess) { return fail(e == hipErrorOutOfMemory ? GCSA2_ERR_OUT_OF_MEMORY : GCSA2_ERR_HIP, std::string("pair blocks: ") + hipGetErrorString(e)); }
  }
  if(locate_table == 1 && ix->img.locate_tab == nullptr)
  {
    if(!locate_table_possible(ix->img)) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "this index cannot have a locate table (no samples)"); }
    if(!make_locate_table(ix)) { return fail(GCSA2_ERR_OUT_OF_MEMORY, "locate table: no memory, or an entry that does not fit"); }
  }
  return GCSA2_OK;
}

namespace {
// the host pipeline of the large host-pointer batches and the staging objects of the small ones (pinned + device memory)
void release_host_staging(gcsa2_index* ix)
{
  std::lock_guard<std::mutex> pipe_guard(ix->pipe_lock);
  for(gcsa2_index::PipeLane& lane : ix->pipe)
  {
    for(gcsa2_index::PipeSet& set : lane.set)
    {
      if(set.h) { (void)hipHostFree(set.h); }
      if(set.d) { (void)hipFree(set.d); }
      if(set.done) { (void)hipEventDestroy(set.done); }
      if(set.computed) { (void)hipEventDestroy(set.computed); }
    }
    if(lane.stream) { (void)hipStreamDestroy(lane.stream); }
    if(lane.down) { (void)hipStreamDestroy(lane.down); }
  }
  ix->pipe.clear();
  std::lock_guard<std::mutex> staging_guard(ix->staging_lock);
  for(Staging* st : ix->staging_pool)
  {
    if(st->stream) { (void)hipStreamDestroy(st->stream); }
    if(st->d) { (void)hipFree(st->d); }
    if(st->h) { (void)hipHostFree(st->h); }
    if(st->z) { (void)hipHostFree(st->z); }
    delete st;
  }
  ix->staging_pool.clear();
  std::lock_guard<std::mutex> arena_guard(ix->arena_lock);
  for(gcsa2_index::Arena& a : ix->arenas) { if(a.base) { (void)hipFree(a.base); } }
  ix->arenas.clear();
}
}  // namespace

int gcsa2_index_trim(gcsa2_index* ix)
{
  if(ix == nullptr) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "null index"); }
  DeviceGuard guard(ix->device);
  if(!guard.ok) { return fail(GCSA2_ERR_HIP, "hipSetDevice failed"); }
  mailbox_stop(ix, true);                                      // (the resident wavefront holds a copy of the image's pointers)
  HIP_TRY(hipDeviceSynchronize());
  release_host_staging(ix);
  if(ix->pool != nullptr) { HIP_TRY(hipMemPoolTrimTo(ix->pool, 0)); }
  return GCSA2_OK;
}

// Shape of the host pipeline (gcsa2_find_batch / _packed): lanes = host threads with their streams and staging sets (1..16),
// chunk_log2 = log2 of the patterns per chunk (15..20); 0 / negative leaves a value as it is; blocking: 1 = the lanes sleep in
// hipEventSynchronize (hipEventBlockingSync), 0 = they spin, -1 = as it is.  Gives the current pipeline back.
int gcsa2_index_set_pipeline(gcsa2_index* ix, int lanes, int chunk_log2, int blocking)
{
  if(ix == nullptr) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "null index"); }
  if(lanes > 16 || chunk_log2 > 20 || (chunk_log2 > 0 && chunk_log2 < 15)) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "host pipeline: lanes 1..16, chunk_log2 15..20"); }
  const int rc = gcsa2_index_trim(ix);
  if(rc != GCSA2_OK) { return rc; }
  std::lock_guard<std::mutex> hold(ix->pipe_lock);
  if(lanes > 0) { ix->tune.pipe_lanes = u32(lanes); }
  if(chunk_log2 > 0) { ix->tune.pipe_chunk = u64(1) << chunk_log2; }
  if(blocking >= 0) { ix->tune.pipe_blocking = (blocking != 0); }
  return GCSA2_OK;
}

void gcsa2_index_destroy(gcsa2_index* ix)
{
  if(ix == nullptr) { return; }
  DeviceGuard guard(ix->device);
  mailbox_stop(ix, true);
  if(ix->d_base) { (void)hipFree(ix->d_base); }
  if(ix->d_kmer) { (void)hipFree(ix->d_kmer); }
  if(ix->d_pred4) { (void)hipFree(ix->d_pred4); }
  if(ix->d_locate) { (void)hipFree(ix->d_locate); }
  if(ix->d_jump) { (void)hipFree(ix->d_jump); }
  if(ix->d_pairs) { (void)hipFree(ix->d_pairs); }
  if(ix->d_slots) { (void)hipFree(ix->d_slots); }
  if(ix->h_slots) { (void)hipHostFree(ix->h_slots); }
  if(ix->pool) { (void)hipDeviceSynchronize(); (void)hipMemPoolDestroy(ix->pool); }
  release_host_staging(ix);
  delete ix;
}

// scalar calls answered by the resident wavefront so far, and how often it had to be launched (diagnostics, tests)
int gcsa2_mailbox_stats(const gcsa2_index* ix, uint64_t* calls, uint64_t* launches)
{
  if(ix == nullptr || calls == nullptr || launches == nullptr) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "null argument"); }
  std::lock_guard<std::mutex> hold(ix->mail.lock);
  *calls = ix->mail.calls; *launches = ix->mail.launches;
  return GCSA2_OK;
}

uint64_t gcsa2_size(const gcsa2_index* ix) { return ix->img.n; }
uint64_t gcsa2_edge_count(const gcsa2_index* ix) { return ix->img.e; }
uint64_t gcsa2_order(const gcsa2_index* ix) { return ix->order; }
uint64_t gcsa2_sample_count(const gcsa2_index* ix) { return ix->img.sample_count; }
uint64_t gcsa2_sample_bits(const gcsa2_index* ix) { return ix->img.sample_width; }
int gcsa2_device(const gcsa2_index* ix) { return ix->device; }
uint64_t gcsa2_device_bytes(const gcsa2_index* ix) { return ix->bytes; }
uint64_t gcsa2_block_bits(const gcsa2_index*) { return BLOCK_BITS; }

// ---- device-pointer entry points: enqueue only -------------------------------------------

#define CHECK_INDEX(ix) do { if((ix) == nullptr) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "null index"); } } while(0)
#define LAUNCH_CHECK(name) do { hipError_t e_ = hipGetLastError(); if(e_ != hipSuccess) { \
  return fail(GCSA2_ERR_HIP, std::string(name) + ": " + hipGetErrorString(e_)); } } while(0)

}  // extern "C"

namespace {

// one input array of `in_words` u64 per query, one output array of `out_words` u64 per query, one kernel in between
template<class Launch>
int simple_batch(const gcsa2_index* ix, const uint64_t* in, u64 in_words, uint64_t* out, u64 out_words, u64 nq, const char* name, Launch launch)
{
  DeviceGuard guard(ix->device);
  Lease lease(ix);
  HIP_TRY(lease.begin(Lease::need(in_words * nq * 8) + Lease::need(out_words * nq * 8), true));
  u64* d_in = lease.dev<u64>(in_words * nq); u64* d_out = lease.dev<u64>(out_words * nq);
  HIP_TRY(lease.up(d_in, in, in_words * nq * sizeof(u64)));
  launch(d_in, d_out, lease.stream());
  { hipError_t e_ = hipGetLastError(); if(e_ != hipSuccess) { return fail(GCSA2_ERR_HIP, std::string(name) + ": " + hipGetErrorString(e_)); } }
  HIP_TRY(lease.down(out, d_out, out_words * nq * sizeof(u64)));
  HIP_TRY(lease.finish());
  return GCSA2_OK;
}

// k_find2 instantiation for this image: JUMP with the jump table, PAIR with the pair blocks
template<bool STATS>
void launch_find2(const gcsa2_index* ix, unsigned grid, hipStream_t st, const uint8_t* d_patterns, const uint64_t* d_offsets, u64 nq,
                  uint64_t* d_ranges, unsigned long long* d_stats, const u32* perm)
{
  const bool jump = ix->img.jump_tab != nullptr, pair = ix->img.flp != nullptr;
#define G2_FIND2(J, P) hipLaunchKernelGGL((k_find2<STATS, J, P>), dim3(grid), dim3(TPB2), 0, st, \
                                          ix->img, d_patterns, d_offsets, nq, d_ranges, d_stats, perm)
  if(jump && pair) { G2_FIND2(true, true); }
  else if(jump) { G2_FIND2(true, false); }
  else if(pair) { G2_FIND2(false, true); }
  else { G2_FIND2(false, false); }
#undef G2_FIND2
}

}  // namespace

extern "C" {

int gcsa2_find_device(const gcsa2_index* ix, const uint8_t* d_patterns, const uint64_t* d_offsets,
                      uint64_t nq, uint64_t* d_ranges, void* stream)
{
  CHECK_INDEX(ix);
  DeviceGuard guard(ix->device);          // the launch goes to the index's device whatever the caller's current one is
  if(nq == 0) { return GCSA2_OK; }
  launch_find2<false>(ix, unsigned((nq + TPB2 - 1) / TPB2), static_cast<hipStream_t>(stream), d_patterns, d_offsets, nq, d_ranges,
                      nullptr, nullptr);
  LAUNCH_CHECK("k_find2");
  return GCSA2_OK;
}

// find() of patterns handed over as 2-bit codes, all of one length (kernels_find.hpp: k_find2<.., PACKED>)
int gcsa2_find_packed_device(const gcsa2_index* ix, const uint64_t* d_codes, uint64_t pattern_length, uint64_t nq, uint64_t* d_ranges, void* stream)
{
  CHECK_INDEX(ix);
  DeviceGuard guard(ix->device);
  if(nq == 0) { return GCSA2_OK; }
  if(pattern_length == 0 || pattern_length >= (u64(1) << 32)) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "packed patterns: the common length must be 1 .. 2^32 - 1"); }
  if(ix->img.sigma < 5) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "packed patterns need the comps 1..4"); }
  const unsigned grid = unsigned((nq + TPB2 - 1) / TPB2);
  const u8* codes = reinterpret_cast<const u8*>(d_codes);
  const u64* length = reinterpret_cast<const u64*>(pattern_length);       // PACKED: the offsets argument carries the length
  hipStream_t st = static_cast<hipStream_t>(stream);
  if(ix->img.flp != nullptr)
  {
    hipLaunchKernelGGL((k_find2<false, false, true, true>), dim3(grid), dim3(TPB2), 0, st, ix->img, codes, length, nq, d_ranges, (unsigned long long*)nullptr, (const u32*)nullptr);
  }
  else
  {
    hipLaunchKernelGGL((k_find2<false, false, false, true>), dim3(grid), dim3(TPB2), 0, st, ix->img, codes, length, nq, d_ranges, (unsigned long long*)nullptr, (const u32*)nullptr);
  }
  LAUNCH_CHECK("k_find2<packed>");
  return GCSA2_OK;
}

int gcsa2_find_device_variant(const gcsa2_index* ix, int variant, const uint8_t* d_patterns, const uint64_t* d_offsets,
                              uint64_t nq, uint64_t* d_ranges, void* stream)
{
  CHECK_INDEX(ix);
  if(nq == 0) { return GCSA2_OK; }
  if(variant == 2) { return gcsa2_find_device(ix, d_patterns, d_offsets, nq, d_ranges, stream); }
  if(variant == 4)   // length-bucketed: sort query ids by pattern length, then k_find2 through the permutation
  {
    if(nq >= (u64(1) << 31)) { return fail(GCSA2_ERR_BUFFER_TOO_SMALL, "length-bucketed find is limited to 2^31 queries per launch"); }
    hipStream_t st = static_cast<hipStream_t>(stream);
    DeviceGuard guard(ix->device);
    u32 *len_in = nullptr, *len_out = nullptr, *idx_in = nullptr, *idx_out = nullptr;
    void* tmp = nullptr;
    size_t tmp_bytes = 0;
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, len_in, len_out, idx_in, idx_out, int(nq), 0, 32, st));
    HIP_TRY(pool_alloc(ix, reinterpret_cast<void**>(&len_in), 4 * nq * sizeof(u32), st));     // four u32 arrays
    HIP_TRY(pool_alloc(ix, &tmp, tmp_bytes, st));
    len_out = len_in + nq; idx_in = len_out + nq; idx_out = idx_in + nq;
    hipLaunchKernelGGL(k_pattern_lengths, dim3(grid_for(nq)), dim3(TPB), 0, st, d_offsets, nq, len_in, idx_in);
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, len_in, len_out, idx_in, idx_out, int(nq), 0, 32, st);
    if(e == hipSuccess)
    {
      launch_find2<false>(ix, unsigned((nq + TPB2 - 1) / TPB2), st, d_patterns, d_offsets, nq, d_ranges, nullptr, idx_out);
      e = hipGetLastError();
    }
    (void)hipFreeAsync(tmp, st); (void)hipFreeAsync(len_in, st);    // stream-ordered: freed after the kernel
    if(e != hipSuccess) { return fail(GCSA2_ERR_HIP, std::string("length-bucketed find: ") + hipGetErrorString(e)); }
    return GCSA2_OK;
  }
  return fail(GCSA2_ERR_INVALID_ARGUMENT, "unknown find variant (2: default, 4: queries ordered by pattern length first)");
}

int gcsa2_find_stats_device(const gcsa2_index* ix, const uint8_t* d_patterns, const uint64_t* d_offsets,
                            uint64_t nq, uint64_t* d_ranges, uint64_t* d_stats, void* stream)
{
  CHECK_INDEX(ix);
  DeviceGuard guard(ix->device);          // the launch goes to the index's device whatever the caller's current one is
  if(d_stats == nullptr) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "null stats buffer"); }
  if(nq == 0) { return GCSA2_OK; }
  static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "u64 atomics");
  launch_find2<true>(ix, unsigned((nq + TPB2 - 1) / TPB2), static_cast<hipStream_t>(stream), d_patterns, d_offsets, nq, d_ranges,
                     reinterpret_cast<unsigned long long*>(d_stats), nullptr);
  LAUNCH_CHECK("k_find2<stats>");
  return GCSA2_OK;
}

uint64_t gcsa2_pair_block_bytes(const gcsa2_index* ix) { return ix->img.flp != nullptr ? 16 * ix->img.flp_nblocks * FLB_BYTES : 0; }
uint64_t gcsa2_find_block_bytes(const gcsa2_index*) { return FLB_BYTES; }
uint64_t gcsa2_kmer_table_k(const gcsa2_index* ix) { return ix->img.kmer_k; }
uint64_t gcsa2_jump_table_bytes(const gcsa2_index* ix) { return ix->img.jump_tab != nullptr ? ix->img.n * sizeof(ulonglong2) : 0; }
uint64_t gcsa2_locate_table_bytes(const gcsa2_index* ix) { return ix->img.locate_tab != nullptr ? ix->img.n * sizeof(u64) : 0; }

int gcsa2_lf_device(const gcsa2_index* ix, const uint64_t* d_in, const uint8_t* d_comps, uint64_t nq,
                    uint64_t* d_out, void* stream)
{
  CHECK_INDEX(ix);
  DeviceGuard guard(ix->device);          // the launch goes to the index's device whatever the caller's current one is
  if(nq == 0) { return GCSA2_OK; }
  hipLaunchKernelGGL(k_lf2, dim3(unsigned((nq + TPB2 - 1) / TPB2)), dim3(TPB2), 0, static_cast<hipStream_t>(stream),
                     ix->img, d_in, d_comps, nq, d_out);
  LAUNCH_CHECK("k_lf2");
  return GCSA2_OK;
}

int gcsa2_count_device(const gcsa2_index* ix, const uint64_t* d_ranges, uint64_t nq, uint64_t* d_counts, void* stream)
{
  CHECK_INDEX(ix);
  DeviceGuard guard(ix->device);          // the launch goes to the index's device whatever the caller's current one is
  if(!ix->img.has_counters) { return fail(GCSA2_ERR_MISSING_COMPONENT, "index was created without counters"); }
  if(nq == 0) { return GCSA2_OK; }
  hipLaunchKernelGGL(k_count, dim3(grid_for(nq)), dim3(TPB), 0, static_cast<hipStream_t>(stream),
                     ix->img, d_ranges, nq, d_counts);
  LAUNCH_CHECK("k_count");
  return GCSA2_OK;
}

int gcsa2_parent_device(const gcsa2_index* ix, const uint64_t* d_ranges, uint64_t nq, gcsa2_stnode* d_nodes, void* stream)
{
  CHECK_INDEX(ix);
  DeviceGuard guard(ix->device);          // the launch goes to the index's device whatever the caller's current one is
  if(!ix->img.has_lcp) { return fail(GCSA2_ERR_MISSING_COMPONENT, "index was created without an LCP array"); }
  if(nq == 0) { return GCSA2_OK; }
  hipLaunchKernelGGL(k_parent, dim3(grid_for(nq)), dim3(TPB), 0, static_cast<hipStream_t>(stream),
                     ix->img, d_ranges, nq, d_nodes);
  LAUNCH_CHECK("k_parent");
  return GCSA2_OK;
}

void gcsa2_locate_discard(gcsa2_locate_job* job)
{
  if(job == nullptr) { return; }
  DeviceGuard guard(job->device);
  if(job->d_offsets) { (void)hipFree(job->d_offsets); }
  if(job->d_values) { (void)hipFree(job->d_values); }
  delete job;
}

// The locate pipeline proper.  d_offsets (nq + 1 entries) is written on the device; values_for(count)
// hands out the buffer for `count` values once that is known (nullptr = refuse); *total_out = number of
// values.  Complete (stream synchronised) on return.
typedef std::function<u64*(u64)> ValuesProvider;

} // extern "C"

namespace {

constexpr int LOCATE_NEEDS_SPLIT = 1;      // internal: not a gcsa2_status

// The totals of a pass on the host: k_publish_totals copies the slot to page-locked host memory and writes a ticket behind
// it; the host polls the ticket (tests/perf/launch_latency.hip: 8 us against 13 us for launch + hipStreamSynchronize, and
// against the 100-400 us a hipMemcpyAsync into pageable memory + synchronise took on this path: three of them were 1.2 ms of
// a 5 ms locate() batch, profiles/r03_locate.md).  The stream is asked now and then, so that a failed launch ends the wait.
int read_totals(const gcsa2_index* ix, unsigned slot, unsigned long long (&totals)[TOTAL_WORDS], hipStream_t stream)
{
  const unsigned long long ticket = ix->next_ticket.fetch_add(1);
  volatile unsigned long long* h = ix->h_slots + u64(TOTAL_WORDS) * slot;
  hipLaunchKernelGGL(k_publish_totals, dim3(1), dim3(64), 0, stream, ix->d_slots + u64(TOTAL_WORDS) * slot, h, ticket);
  LAUNCH_CHECK("k_publish_totals");
  for(u64 spins = 1; h[TOTAL_WORDS - 1] != ticket; spins++)
  {
    if(spins > 20000) { std::this_thread::yield(); }          // a long pass (tens of milliseconds of sorting): leave the core to others
    if((spins & 0xFFF) == 0)
    {
      const hipError_t q = hipStreamQuery(stream);
      if(q == hipSuccess) { if(h[TOTAL_WORDS - 1] == ticket) { break; } return fail(GCSA2_ERR_HIP, "locate: the stream went idle without publishing its totals"); }
      if(q != hipErrorNotReady) { return fail(GCSA2_ERR_HIP, std::string("locate: ") + hipGetErrorString(q)); }
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  for(u32 i = 0; i < TOTAL_WORDS; i++) { totals[i] = h[i]; }
  return GCSA2_OK;
}

// One pass of the locate pipeline.  The flag words' prefix sums are 32-bit, so a pass takes fewer than 2^31 values before
// deduplication; a larger batch returns LOCATE_NEEDS_SPLIT (allow_split) with the exclusive scan of the per-query raw counts
// left in d_offsets, and locate_core below cuts it.
// known_out / known_capacity: the caller's own value buffer (gcsa2_locate_into).  The pass then never waits for the number of
// distinct values: the compaction writes into the buffer under a capacity guard and the total is read with the final
// synchronisation (too small a buffer is reported then, as before).  Host round trips of a pass in sorted mode: the totals
// after the size scans (needed for the scratch), one more only if some segment has more than 4096 values, the end.
int locate_chunk(const gcsa2_index* ix, const u64* d_ranges, u64 nq, int sort, u64* d_offsets,
                 const ValuesProvider& values_for, u64* total_out, hipStream_t stream, bool allow_split,
                 u64* known_out = nullptr, u64 known_capacity = 0)
{
  *total_out = 0;
  if(nq == 0)
  {
    HIP_TRY(hipMemsetAsync(d_offsets, 0, sizeof(u64), stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return GCSA2_OK;
  }
  Scratch scratch(ix, stream);
  const auto t_start = std::chrono::steady_clock::now();
  double stamps[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  auto stamp = [&](int k) { if(ix->tune.locate_trace) { stamps[k] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_start).count(); } };

  // [node_counts | raw_counts | node_off] (nq + 1 each), [seg_begin | seg_end] (nq each), the totals; the
  // scan of the raw counts goes straight into the job's offsets (final as they are unless duplicates
  // have to be removed, rewritten in place otherwise)
  u64* sizes = nullptr; u64* segs = nullptr;
  const unsigned slot = ix->next_slot.fetch_add(1) % RESULT_SLOTS;
  unsigned long long* d_totals = ix->d_slots + u64(TOTAL_WORDS) * slot;
  // The one-kernel attempt costs a batch that does not fit it a launch, a host poll and nq words of scratch: after a
  // misfit the next calls on this handle go straight to the pipeline, and every eighth one tries again.
  bool try_single = (ix->img.locate_tab != nullptr && ix->tune.locate_single);
  if(try_single && ix->single_backoff.load(std::memory_order_relaxed) > 0) { ix->single_backoff.fetch_sub(1, std::memory_order_relaxed); try_single = false; }
  if(try_single)
  {
    // every range one path node with one directly stored value?  Then this kernel is the whole answer (k_locate_single); the
    // first misfit sends the batch through the pipeline below.  (The answer does not depend on `sort`: one value per range.)
    // (into scratch, copied out on success: a batch with a misfit must not leave values of this attempt in the caller's buffer
    // beyond what the pipeline then writes)
    u64* out = nullptr;
    HIP_TRY(scratch.get(out, nq));
    HIP_TRY(hipMemsetAsync(d_totals, 0, TOTAL_WORDS * sizeof(unsigned long long), stream));
    hipLaunchKernelGGL(k_locate_single, dim3(grid_for(nq + 1)), dim3(TPB), 0, stream, ix->img, d_ranges, nq, d_offsets, out, nq, d_totals);
    LAUNCH_CHECK("k_locate_single");
    unsigned long long first[TOTAL_WORDS];
    int rc1 = read_totals(ix, slot, first, stream);              // in stream order behind the kernel
    if(rc1 != GCSA2_OK) { return rc1; }
    scratch.settled = true;
    if(first[0] == 0)
    {
      *total_out = nq;
      u64* dest = known_out;
      if(known_out != nullptr) { if(known_capacity < nq) { return fail(GCSA2_ERR_BUFFER_TOO_SMALL, "values buffer too small"); } }
      else
      {
        dest = values_for(nq);
        if(dest == nullptr) { return g_error.empty() ? fail(GCSA2_ERR_BUFFER_TOO_SMALL, "values buffer too small") : GCSA2_ERR_BUFFER_TOO_SMALL; }
      }
      scratch.settled = false;
      HIP_TRY(hipMemcpyAsync(dest, out, nq * sizeof(u64), hipMemcpyDeviceToDevice, stream));
      HIP_TRY(hipStreamSynchronize(stream));
      scratch.settled = true;
      return GCSA2_OK;
    }
    ix->single_backoff.store(7, std::memory_order_relaxed);
    scratch.settled = false;
  }
  // Round 6: a wide range whose path nodes have one value each does not go through the table pass -- the workgroup that splits
  // its values reads them from the locate table itself (k_classify_fused, k_over_split): (sorted mode, table, split sort)
  const bool fuse = (sort && ix->img.locate_tab != nullptr && ix->tune.locate_fuse && ix->tune.dedup_huge && ix->tune.locate_split_sort
                     && ix->img.sample_width < 63);
  const u64 fuse_above = (fuse ? ix->tune.fuse_above : 0);
  HIP_TRY(scratch.get(sizes, 3 * (nq + 1))); HIP_TRY(scratch.get(segs, 7 * nq));
  u64 *node_counts = sizes, *raw_counts = sizes + (nq + 1), *node_off = sizes + 2 * (nq + 1), *raw_off = d_offsets;
  u64 *seg_begin = segs, *seg_end = segs + nq, *huge_begin = segs + 2 * nq, *huge_end = segs + 3 * nq, *over_begin = segs + 4 * nq, *over_end = segs + 5 * nq;
  const u64** over_src = reinterpret_cast<const u64**>(segs + 6 * nq);
  u64* candidates = over_end;                                  // (consumed by k_classify_fused before k_collect_multi writes the list)
  HIP_TRY(hipMemsetAsync(d_totals, 0, TOTAL_WORDS * sizeof(unsigned long long), stream));
  hipLaunchKernelGGL(k_locate_sizes, dim3(grid_for(nq)), dim3(TPB), 0, stream, ix->img, d_ranges, nq, node_counts, raw_counts, d_totals, fuse_above, candidates);
  LAUNCH_CHECK("k_locate_sizes");
  if(fuse_above != 0)
  {
    hipLaunchKernelGGL(k_classify_fused, dim3(unsigned(nq < 8192 ? nq : 8192)), dim3(64), 0, stream, ix->img, d_ranges, candidates, d_totals, node_counts);
    LAUNCH_CHECK("k_classify_fused");
  }

  // exclusive scans over nq + 1 entries: entry nq becomes the total
  size_t tmp_bytes = 0;
  HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, node_counts, node_off, size_t(nq + 1), stream));
  char* tmp = nullptr;
  HIP_TRY(scratch.get(tmp, tmp_bytes));
  HIP_TRY(hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, node_counts, node_off, size_t(nq + 1), stream));
  HIP_TRY(hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, raw_counts, raw_off, size_t(nq + 1), stream));
  // segments with more than one raw value: the only ones removeDuplicates has to touch
  const u32 medium_limit = ix->tune.sort_medium_limit;
  // with the duplicate filter every segment beyond the medium class goes through it first (listed as huge), without it only
  // the ones the workgroup sort cannot hold
  const u32 big_limit = (ix->tune.dedup_huge && sort ? (medium_limit > SMALL_SEGMENT ? medium_limit : SMALL_SEGMENT) : BIG_SEGMENT);
  hipLaunchKernelGGL(k_collect_multi, dim3(unsigned((nq + COLLECT_THREADS - 1) / COLLECT_THREADS)), dim3(COLLECT_THREADS), 0, stream, node_off, raw_off, nq, d_totals, seg_begin, seg_end, huge_begin, huge_end, medium_limit, big_limit,
                     d_ranges, ix->img.locate_tab, over_begin, over_end, over_src);
  LAUNCH_CHECK("k_collect_multi");
  unsigned long long totals[TOTAL_WORDS];
  stamp(0);
  int rc = read_totals(ix, slot, totals, stream);
  if(rc != GCSA2_OK) { return rc; }
  stamp(1);
  scratch.settled = true;            // the totals have arrived: nothing of this pass is in flight (until the next launch)
  const u64 total_nodes = totals[T_NODES], total_raw = totals[T_RAW], multi = totals[T_MULTI], huge_a = totals[T_HUGE_A], huge_b = totals[T_HUGE_B];
  const u64 large = totals[T_LARGE], medium = totals[T_MEDIUM];
  if(total_raw > ix->tune.locate_split || nq > ix->tune.locate_split_queries)
  {
    if(allow_split) { return LOCATE_NEEDS_SPLIT; }
    return fail(GCSA2_ERR_BUFFER_TOO_SMALL, "locate: one range alone has 2^31 or more values before deduplication");
  }

  // the per-workgroup owners of the walk kernel (k_block_owners) fit into one of the count arrays, which the scans have
  // consumed, unless the ranges are wide; the other one serves as the per-query slot counters of the unordered table walk
  u64* owners = raw_counts;
  if(total_nodes / OWNER_SPAN + 3 > nq + 1) { HIP_TRY(scratch.get(owners, total_nodes / OWNER_SPAN + 3)); }
  u64* extra_slots = node_counts;

  if(total_raw == 0)
  {
    HIP_TRY(hipMemsetAsync(d_offsets, 0, (nq + 1) * sizeof(u64), stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return GCSA2_OK;
  }
  if(!sort || multi == 0)
  {
    // sort == false (gcsa.cpp:827-842 without removeDuplicates): values in path order, the values of
    // one path node in sample order, duplicates kept -- exactly the walk's output.  With at most one
    // value per query that output is already sorted and distinct.
    u64* out = (known_out != nullptr ? (total_raw <= known_capacity ? known_out : nullptr) : values_for(total_raw));
    if(out == nullptr) { *total_out = total_raw; return g_error.empty() ? fail(GCSA2_ERR_BUFFER_TOO_SMALL, "values buffer too small") : GCSA2_ERR_BUFFER_TOO_SMALL; }
    scratch.settled = false;
    launch_walk(ix, d_ranges, nq, node_off, raw_off, total_nodes, out, owners, stream);
    LAUNCH_CHECK("k_locate_walk");
    HIP_TRY(hipStreamSynchronize(stream));
    scratch.settled = true;
    *total_out = total_raw;
    return GCSA2_OK;
  }

  u64 *sorted = nullptr, *words = nullptr; u32 *word_counts = nullptr, *word_before = nullptr;
  const u64 nwords = total_raw / 64 + 1;
  // Round 6: a caller's buffer that holds the values BEFORE deduplication is the target of the table pass and of every sort;
  // when no sort meets a duplicate (the flag totals[T_DUPS]) the values are final where they lie, at the offsets of the size
  // scan, and nothing is compacted; otherwise k_mark_compact compacts in place.  (A buffer sized for the distinct values only
  // keeps the scratch array and the out-of-place compaction.)
  // The job interface (the library makes the value buffer) takes the same path when the values before deduplication are at most
  // 2^30 (8 GB): the buffer is then made for THEM, before the sorts instead of behind them -- a job's buffer may be longer than
  // its `total`.
  if(known_out == nullptr && ix->tune.locate_fused_compact && ix->tune.locate_in_place && total_raw <= (u64(1) << 30))
  {
    u64* made = values_for(total_raw);
    if(made == nullptr) { return g_error.empty() ? fail(GCSA2_ERR_BUFFER_TOO_SMALL, "values buffer too small") : GCSA2_ERR_BUFFER_TOO_SMALL; }
    known_out = made; known_capacity = total_raw;
  }
  const bool in_place = (known_out != nullptr && ix->tune.locate_fused_compact && ix->tune.locate_in_place && total_raw <= known_capacity);
  if(in_place) { sorted = known_out; } else { HIP_TRY(scratch.get(sorted, total_raw)); }
  bool force_compact = !in_place;
  HIP_TRY(scratch.get(words, nwords)); HIP_TRY(scratch.get(word_counts, nwords + 1)); HIP_TRY(scratch.get(word_before, nwords + 1));
  scratch.settled = false;
  if(total_nodes == 0) { }                                     // (every range with values is fused: nothing for the table pass)
  else if(ix->img.locate_tab != nullptr)
  {
    // unordered table walk in two passes (kernels_locate.hpp): single values at once, the path nodes with several values
    // marked (one word per 64 nodes) and worked through afterwards.  (The second pass runs whenever the first one may have
    // marked a node: also a node with ONE value is marked when that value does not fit a direct entry.)
    const u64 blocks = grid_for(total_nodes), spans = (total_nodes + OWNER_SPAN - 1) / OWNER_SPAN;
    u64* later_words = nullptr;
    HIP_TRY(scratch.get(later_words, spans));
    HIP_TRY(hipMemsetAsync(extra_slots, 0, nq * sizeof(u64), stream));
    hipLaunchKernelGGL(k_block_owners, dim3(grid_for(spans + 1)), dim3(TPB), 0, stream, node_off, nq, total_nodes, OWNER_SPAN, spans, owners);
    hipLaunchKernelGGL(k_locate_tab_unordered, dim3(unsigned((blocks + TAB_SPANS - 1) / TAB_SPANS)), dim3(TPB), 0, stream, ix->img, d_ranges, nq, node_off, raw_off,
                       total_nodes, sorted, owners, later_words, spans);
    if(total_raw > total_nodes || ix->img.sample_width >= 63)
    {
      hipLaunchKernelGGL(k_locate_tab_rest, dim3(grid_for(spans)), dim3(TPB), 0, stream, ix->img, d_ranges, node_off, raw_off, total_nodes, sorted,
                         owners, later_words, spans, reinterpret_cast<unsigned long long*>(extra_slots));
    }
  }
  else { launch_walk(ix, d_ranges, nq, node_off, raw_off, total_nodes, sorted, owners, stream); }
  LAUNCH_CHECK("k_locate_walk");

  // removeDuplicates: queries with up to SMALL_SEGMENT values are sorted in registers, up to MEDIUM_SEGMENT by a wavefront
  // and up to BIG_SEGMENT by a workgroup in LDS, all in place.  Longer ones first lose their duplicates (k_dedup_huge) and
  // join those lists with their distinct values (the sorts are launched over upper bounds of the list lengths and read the
  // lengths on the device); a segment with more than BIG_SEGMENT distinct values is listed for the device-wide radix sort
  // over (segment, value) keys.  Then flag + scan + compact.
  u64 over = totals[T_OVER], over_values = totals[T_OVER_VALUES];      // (the fused ranges: k_collect_multi listed them)
  const u64 huge = huge_a + huge_b;
  // the slots behind the distinct values of a filtered segment are marked in a bitmap instead of being filled and read again
  // (k_dedup_huge, k_mark_compact) -- when the one-sweep compaction is the one that will run
  // (an index whose values lie below 2^32 -- samples of at most 31 bits + fewer than 2^23 steps -- gets the 32-bit hash table)
  const bool narrow_values = (ix->img.sample_width <= 31 && ix->tune.dedup_narrow);
  unsigned long long* dead = nullptr;
  if(huge > 0 && ix->tune.dedup_huge && known_out != nullptr && ix->tune.locate_fused_compact)
  {
    HIP_TRY(scratch.get(dead, nwords));
    HIP_TRY(hipMemsetAsync(dead, 0, nwords * sizeof(unsigned long long), stream));
  }
  if(huge > 0 && !ix->tune.dedup_huge)
  {
    // (A/B knob: no duplicate filter; every segment of more than BIG_SEGMENT values -- all on the second list -- goes to the radix sort)
    HIP_TRY(hipMemsetAsync(over_src, 0, nq * sizeof(u64), stream));      // (no fused ranges without the filter: every segment is read from the raw values)
    hipLaunchKernelGGL(k_huge_to_over, dim3(grid_for(huge_b)), dim3(TPB), 0, stream, huge_begin, huge_end, nq - 1, huge_b, over_begin, over_end, d_totals);
    LAUNCH_CHECK("k_huge_to_over");
    rc = read_totals(ix, slot, totals, stream);
    if(rc != GCSA2_OK) { return rc; }
    over = totals[T_OVER]; over_values = totals[T_OVER_VALUES];
  }
  else if(huge_a > 0)
  {
    // (64-bit words for the short segments whatever the index: with 32-bit words -- 32 KB, five workgroups on a CU instead of two --
    // this kernel was SLOWER on the 2^23 repeat graph, 0.81 against 0.62 ms for the 32-mer batch; the long segments' kernel gains, 3.6 -> 2.9 ms)
    hipLaunchKernelGGL((k_dedup_huge<BIG_SEGMENT, false, 512, unsigned long long>), dim3(unsigned(huge_a)), dim3(512), 0, stream, huge_begin, huge_end, nq - 1, sorted, nq,
                       medium_limit, d_totals, seg_begin, seg_end, over_begin, over_end, over_src, dead);
    LAUNCH_CHECK("k_dedup_huge");
  }
  if(huge_b > 0 && ix->tune.dedup_huge)
  {
    if(narrow_values)
    {
      hipLaunchKernelGGL((k_dedup_huge<2 * BIG_SEGMENT, true, 1024, u32>), dim3(unsigned(huge_b)), dim3(1024), 0, stream, huge_begin, huge_end, nq - 1, sorted, nq,
                         medium_limit, d_totals, seg_begin, seg_end, over_begin, over_end, over_src, dead);
    }
    else
    {
      hipLaunchKernelGGL((k_dedup_huge<2 * BIG_SEGMENT, true, 1024, unsigned long long>), dim3(unsigned(huge_b)), dim3(1024), 0, stream, huge_begin, huge_end, nq - 1, sorted, nq,
                         medium_limit, d_totals, seg_begin, seg_end, over_begin, over_end, over_src, dead);
    }
    LAUNCH_CHECK("k_dedup_huge");
    stamp(2);
    rc = read_totals(ix, slot, totals, stream);          // only these segments can overflow
    if(rc != GCSA2_OK) { return rc; }
    over = totals[T_OVER]; over_values = totals[T_OVER_VALUES];
    stamp(3);
  }
  hipLaunchKernelGGL(k_sort_small, dim3(grid_for(nq)), dim3(TPB), 0, stream, raw_off, nq, sorted, d_totals);
  LAUNCH_CHECK("k_sort_small");
  if(medium + huge > 0)
  {
    hipLaunchKernelGGL(k_sort_medium, dim3(unsigned(medium + huge)), dim3(64), 0, stream, seg_begin, seg_end, nq - 1, sorted, d_totals);
    LAUNCH_CHECK("k_sort_medium");
  }
  if(large + huge > 0)
  {
    hipLaunchKernelGGL((k_sort_big<4096, 0>), dim3(unsigned(large + huge)), dim3(big_threads<4096>()), 0, stream, seg_begin, seg_end, sorted, d_totals + T_LARGE, d_totals + T_DUPS);
    hipLaunchKernelGGL((k_sort_big<BIG_SEGMENT, 4096>), dim3(unsigned(large + huge)), dim3(big_threads<BIG_SEGMENT>()), 0, stream, seg_begin, seg_end, sorted, d_totals + T_LARGE, d_totals + T_DUPS);
    LAUNCH_CHECK("k_sort_big");
  }
  auto radix_over = [&](u64* over_begin, u64* over_end, u64 over, u64 over_values) -> int
  {
    force_compact = true;                                      // (the library's sort does not say whether it met duplicates)
    // keys = (rank of the segment) << value_bits | value; a value is a sample + fewer than 2^23 steps
    u32 value_bits = u32(ix->img.sample_width > 24 ? ix->img.sample_width : 24) + 1, rank_bits = 1;
    while((u64(1) << rank_bits) < over) { rank_bits++; }
    u64 *over_off = nullptr, *over_len = nullptr;
    HIP_TRY(scratch.get(over_off, over + 1)); HIP_TRY(scratch.get(over_len, over + 1));
    hipLaunchKernelGGL(k_over_lengths, dim3(grid_for(over + 1)), dim3(TPB), 0, stream, over_begin, over_end, over, over_len);
    size_t scan_bytes = 0;
    HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, over_len, over_off, size_t(over + 1), stream));
    char* scan_tmp = nullptr;
    HIP_TRY(scratch.get(scan_tmp, scan_bytes));
    HIP_TRY(hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, over_len, over_off, size_t(over + 1), stream));
    if(value_bits + rank_bits <= 64)
    {
      u64 *keys_a = nullptr, *keys_b = nullptr;
      HIP_TRY(scratch.get(keys_a, over_values)); HIP_TRY(scratch.get(keys_b, over_values));
      hipLaunchKernelGGL(k_over_pack, dim3(grid_for(over_values)), dim3(TPB), 0, stream, over_begin, over_off, over, over_values, sorted, value_bits, keys_a);
      LAUNCH_CHECK("k_over_pack");
      hipcub::DoubleBuffer<u64> keys(keys_a, keys_b);
      size_t sort_bytes = 0;
      HIP_TRY(hipcub::DeviceRadixSort::SortKeys(nullptr, sort_bytes, keys, size_t(over_values), 0, int(value_bits + rank_bits), stream));
      char* sort_tmp = nullptr;
      HIP_TRY(scratch.get(sort_tmp, sort_bytes));
      HIP_TRY(hipcub::DeviceRadixSort::SortKeys(sort_tmp, sort_bytes, keys, size_t(over_values), 0, int(value_bits + rank_bits), stream));
      hipLaunchKernelGGL(k_over_unpack, dim3(grid_for(over_values)), dim3(TPB), 0, stream, over_begin, over_off, over_values, keys.Current(), value_bits, sorted);
      LAUNCH_CHECK("k_over_unpack");
    }
    else
    {
      // (values too wide to share a key with the segment rank: the library's segmented sort, from a copy)
      u64* raw = nullptr;
      HIP_TRY(scratch.get(raw, total_raw));
      HIP_TRY(hipMemcpyAsync(raw, sorted, total_raw * sizeof(u64), hipMemcpyDeviceToDevice, stream));
      size_t sort_bytes = 0;
      HIP_TRY(hipcub::DeviceSegmentedRadixSort::SortKeys(nullptr, sort_bytes, raw, sorted, int(total_raw), int(over), over_begin, over_end, 0, 64, stream));
      char* sort_tmp = nullptr;
      HIP_TRY(scratch.get(sort_tmp, sort_bytes));
      HIP_TRY(hipcub::DeviceSegmentedRadixSort::SortKeys(sort_tmp, sort_bytes, raw, sorted, int(total_raw), int(over), over_begin, over_end, 0, 64, stream));
    }
    return GCSA2_OK;
  };
  if(over > 0 && ix->tune.locate_split_sort)
  {
    // segments of more than BIG_SEGMENT distinct values: one workgroup each splits its segment into buckets that the
    // workgroup sort holds (k_over_split); what a skewed segment leaves over goes to the device-wide radix sort as before
    const u64 bucket_cap = over_values / 64 + over + 16;                  // listed buckets have more than 64 values
    u64 *split_tmp = nullptr, *bkt_begin = nullptr, *bkt_end = nullptr, *skew_begin = nullptr, *skew_end = nullptr;
    HIP_TRY(scratch.get(split_tmp, total_raw));
    HIP_TRY(scratch.get(bkt_begin, bucket_cap)); HIP_TRY(scratch.get(bkt_end, bucket_cap));
    const u32 skew_above = ix->tune.split_skew;                            // BIG_SEGMENT (lower in tests): what the workgroup sort takes
    const u64 skew_cap = over_values / skew_above + over + 16;
    HIP_TRY(scratch.get(skew_begin, skew_cap)); HIP_TRY(scratch.get(skew_end, skew_cap));
    const u64 mid_cap = over_values / BUCKET_BY_WAVE + over + 16;         // buckets of 513 .. 1024 values
    u64 *mid_begin = nullptr, *mid_end = nullptr;
    HIP_TRY(scratch.get(mid_begin, mid_cap)); HIP_TRY(scratch.get(mid_end, mid_cap));
    if(ix->tune.split_tiled)
    {
      hipLaunchKernelGGL(k_over_split<true>, dim3(unsigned(over)), dim3(SPLIT_THREADS), 0, stream, over_begin, over_end, sorted, split_tmp,
                         bkt_begin, bkt_end, skew_begin, skew_end, d_totals, skew_above, ix->tune.split_target, bucket_cap - 1, over_src, mid_begin, mid_end);
    }
    else
    {
      hipLaunchKernelGGL(k_over_split<false>, dim3(unsigned(over)), dim3(SPLIT_THREADS), 0, stream, over_begin, over_end, sorted, split_tmp,
                         bkt_begin, bkt_end, skew_begin, skew_end, d_totals, skew_above, ix->tune.split_target, bucket_cap - 1, over_src, mid_begin, mid_end);
    }
    LAUNCH_CHECK("k_over_split");
    rc = read_totals(ix, slot, totals, stream);
    if(rc != GCSA2_OK) { return rc; }
    const u64 buckets = totals[T_BUCKETS], big_buckets = totals[T_BIG_BUCKETS], skew = totals[T_SKEW], skew_values = totals[T_SKEW_VALUES];
    const u64 mid_buckets = totals[T_MID_BUCKETS];
    if(buckets + big_buckets > bucket_cap || mid_buckets > mid_cap) { return fail(GCSA2_ERR_HIP, "locate: more buckets than the split reserved"); }
    if(buckets > 0)
    {
      hipLaunchKernelGGL(k_sort_bucket<BUCKET_BY_WAVE>, dim3(unsigned(buckets)), dim3(64), 0, stream, bkt_begin, bkt_end, sorted, split_tmp, d_totals);
      LAUNCH_CHECK("k_sort_bucket");
    }
    if(mid_buckets > 0)
    {
      hipLaunchKernelGGL(k_sort_bucket<MEDIUM_SEGMENT>, dim3(unsigned(mid_buckets)), dim3(64), 0, stream, mid_begin, mid_end, sorted, split_tmp, d_totals);
      LAUNCH_CHECK("k_sort_bucket (513 .. 1024 values)");
    }
    if(big_buckets > 0)                                       // (listed from the back of the same arrays)
    {
      hipLaunchKernelGGL((k_sort_big<4096, MEDIUM_SEGMENT>), dim3(unsigned(big_buckets)), dim3(big_threads<4096>()), 0, stream, bkt_begin, bkt_end, sorted, d_totals + T_BIG_BUCKETS, d_totals + T_DUPS, split_tmp, bucket_cap - 1);
      hipLaunchKernelGGL((k_sort_big<BIG_SEGMENT, 4096>), dim3(unsigned(big_buckets)), dim3(big_threads<BIG_SEGMENT>()), 0, stream, bkt_begin, bkt_end, sorted, d_totals + T_BIG_BUCKETS, d_totals + T_DUPS, split_tmp, bucket_cap - 1);
      LAUNCH_CHECK("k_sort_big (buckets)");
    }
    if(skew > 0) { rc = radix_over(skew_begin, skew_end, skew, skew_values); if(rc != GCSA2_OK) { return rc; } }
  }
  else if(over > 0) { rc = radix_over(over_begin, over_end, over, over_values); if(rc != GCSA2_OK) { return rc; } }
  if(in_place && !force_compact)
  {
    // every value is sorted in the caller's buffer, at the offsets of the size scan (d_offsets): if no sort met a duplicate, that
    // is the result.  (One poll of the totals -- the pass is complete behind it; a launch of the compaction that finds out on
    // the device that it has nothing to do cost 0.34 ms of the 8.8 ms of the 16-mer batch on the 2^30-base text.)
    stamp(4);
    rc = read_totals(ix, slot, totals, stream);
    if(rc != GCSA2_OK) { return rc; }
    stamp(5);
    if(totals[T_DUPS] == 0)
    {
      scratch.settled = true;
      *total_out = total_raw;
      HIP_TRY(hipStreamSynchronize(stream));
      stamp(6);
      return GCSA2_OK;
    }
  }
  if(known_out != nullptr && ix->tune.locate_fused_compact)
  {
    // the caller owns the values buffer: marks, counts, prefix sums and compaction in one sweep (k_mark_compact)
    const u64 tiles = (nwords * 64 + COMPACT_TILE - 1) / COMPACT_TILE;
    unsigned long long* tile_status = nullptr;
    HIP_TRY(scratch.get(tile_status, tiles + 1));                         // (+ the ticket counter behind the last tile)
    HIP_TRY(hipMemsetAsync(tile_status, 0, (tiles + 1) * sizeof(unsigned long long), stream));
    HIP_TRY(hipMemsetAsync(words, 0, nwords * sizeof(u64), stream));
    hipLaunchKernelGGL(k_mark_starts, dim3(grid_for(nq)), dim3(TPB), 0, stream, raw_off, nq, words);
    hipLaunchKernelGGL(k_mark_compact, dim3(unsigned(tiles)), dim3(COMPACT_THREADS), 0, stream, sorted, total_raw, nwords, words, word_before,
                       known_out, known_capacity, tile_status, reinterpret_cast<unsigned int*>(tile_status + tiles), d_totals + T_UNIQUE, reinterpret_cast<const u64*>(dead));
    LAUNCH_CHECK("k_mark_starts / k_mark_compact");
    hipLaunchKernelGGL(k_final_offsets, dim3(grid_for(nq + 1)), dim3(TPB), 0, stream, words, word_before, nq, total_raw, nwords, d_offsets);
    LAUNCH_CHECK("k_final_offsets");
    stamp(4);
    rc = read_totals(ix, slot, totals, stream);          // in stream order behind everything above: the pass is complete
    if(rc != GCSA2_OK) { return rc; }
    stamp(5);
    scratch.settled = true;
    *total_out = totals[T_UNIQUE];
    if(totals[T_UNIQUE] > known_capacity) { return fail(GCSA2_ERR_BUFFER_TOO_SMALL, "values buffer too small"); }
    HIP_TRY(hipStreamSynchronize(stream));
    stamp(6);
    return GCSA2_OK;
  }
  hipLaunchKernelGGL(k_mark_changes, dim3(grid_for(nwords * 64)), dim3(TPB), 0, stream, sorted, total_raw, words);
  hipLaunchKernelGGL(k_mark_starts, dim3(grid_for(nq)), dim3(TPB), 0, stream, raw_off, nq, words);
  hipLaunchKernelGGL(k_word_counts, dim3(grid_for(nwords + 1)), dim3(TPB), 0, stream, words, nwords, word_counts);
  LAUNCH_CHECK("k_mark_changes / k_mark_starts / k_word_counts");
  size_t scan_bytes = 0;
  HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, word_counts, word_before, size_t(nwords + 1), stream));
  char* scan_tmp = nullptr;
  HIP_TRY(scratch.get(scan_tmp, scan_bytes));
  HIP_TRY(hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, word_counts, word_before, size_t(nwords + 1), stream));
  hipLaunchKernelGGL(k_publish, dim3(1), dim3(1), 0, stream, word_before + nwords, d_totals + T_UNIQUE);
  LAUNCH_CHECK("k_publish");
  u64 total_unique = 0;
  u64* out = known_out;
  u64 capacity = known_capacity;
  if(out == nullptr)                 // the values buffer is made for the number of distinct values: wait for it
  {
    rc = read_totals(ix, slot, totals, stream);
    if(rc != GCSA2_OK) { return rc; }
    total_unique = totals[T_UNIQUE];
    *total_out = total_unique;
    scratch.settled = true;
    out = values_for(total_unique);
    if(out == nullptr) { return g_error.empty() ? fail(GCSA2_ERR_BUFFER_TOO_SMALL, "values buffer too small") : GCSA2_ERR_BUFFER_TOO_SMALL; }
    capacity = total_unique;
    scratch.settled = false;
  }
  hipLaunchKernelGGL(k_compact, dim3(grid_for(total_raw)), dim3(TPB), 0, stream, sorted, words, word_before, total_raw, out, capacity);
  LAUNCH_CHECK("k_compact");
  hipLaunchKernelGGL(k_final_offsets, dim3(grid_for(nq + 1)), dim3(TPB), 0, stream, words, word_before, nq, total_raw, nwords, d_offsets);
  LAUNCH_CHECK("k_final_offsets");
  stamp(4);
  if(known_out != nullptr)
  {
    rc = read_totals(ix, slot, totals, stream);          // in stream order behind everything above: the pass is complete
    if(rc != GCSA2_OK) { return rc; }
    stamp(5);
    scratch.settled = true;
    total_unique = totals[T_UNIQUE];
    *total_out = total_unique;
    if(total_unique > known_capacity) { return fail(GCSA2_ERR_BUFFER_TOO_SMALL, "values buffer too small"); }
  }
  HIP_TRY(hipStreamSynchronize(stream));
  scratch.settled = true;
  stamp(6);
  if(ix->tune.locate_trace)
  {
    std::fprintf(stderr, "[locate] enqueued sizes %.0f us | totals %.0f | walk + filter enqueued %.0f | totals %.0f | sorts + flags + compact enqueued %.0f | totals %.0f | done %.0f\n",
                 stamps[0], stamps[1], stamps[2], stamps[3], stamps[4], stamps[5], stamps[6]);
  }
  return GCSA2_OK;
}

// largest q1 in (q0, min(nq, q0 + most)] with raw_off[q1] - raw_off[q0] <= limit (q0 if even the first query exceeds it); one thread
__global__ void k_locate_cut(const u64* __restrict__ raw_off, u64 nq, u64 q0, u64 limit, u64 most, unsigned long long* __restrict__ out)
{
  u64 lo = q0, hi = (nq - q0 > most ? q0 + most : nq);
  const u64 base = raw_off[q0];
  while(lo < hi)
  {
    const u64 mid = (lo + hi + 1) >> 1;
    if(raw_off[mid] - base <= limit) { lo = mid; } else { hi = mid - 1; }
  }
  *out = lo;
}

__global__ __launch_bounds__(TPB) void k_uniform_offsets(u64* __restrict__ dst, u64 count, u64 stride)
{
  const u64 i = u64(blockIdx.x) * TPB + threadIdx.x;
  if(i < count) { dst[i] = i * stride; }
}

__global__ __launch_bounds__(TPB) void k_shift_offsets(const u64* __restrict__ src, u64 count, u64 base, u64* __restrict__ dst)
{
  const u64 i = u64(blockIdx.x) * TPB + threadIdx.x;
  if(i < count) { dst[i] = src[i] + base; }
}

// The locate pipeline for batches of any size: one pass when the batch has at most 2^30 ranges and fewer than 2^31 values before deduplication (the
// paper's 16-mer batch has 2.5 G, paper.tex:403), otherwise consecutive sub-batches of queries, each below that, whose value
// arrays are concatenated and whose offsets are shifted -- the same CSR a single pass would give.
int locate_core(const gcsa2_index* ix, const u64* d_ranges, u64 nq, int sort, u64* d_offsets,
                const ValuesProvider& values_for, u64* total_out, hipStream_t stream, u64* known_out = nullptr, u64 known_capacity = 0)
{
  int rc = locate_chunk(ix, d_ranges, nq, sort, d_offsets, values_for, total_out, stream, true, known_out, known_capacity);
  if(rc != LOCATE_NEEDS_SPLIT) { return rc; }
  struct Part { u64 q0 = 0, q1 = 0, total = 0; u64* d_off = nullptr; u64* d_val = nullptr; };
  std::vector<Part> parts;
  struct Release { std::vector<Part>& p; ~Release() { for(Part& x : p) { if(x.d_off) { (void)hipFree(x.d_off); } if(x.d_val) { (void)hipFree(x.d_val); } } } } release{parts};
  unsigned long long* d_cut = ix->d_slots + TOTAL_WORDS * (ix->next_slot.fetch_add(1) % RESULT_SLOTS);
  u64 q0 = 0;
  while(q0 < nq)                                   // d_offsets still holds the scan of the raw counts
  {
    unsigned long long q1 = 0;
    hipLaunchKernelGGL(k_locate_cut, dim3(1), dim3(1), 0, stream, d_offsets, nq, q0, ix->tune.locate_split, ix->tune.locate_split_queries, d_cut);
    HIP_TRY(hipMemcpyAsync(&q1, d_cut, sizeof(q1), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if(q1 <= q0) { return fail(GCSA2_ERR_BUFFER_TOO_SMALL, "locate: one range alone has 2^31 or more values before deduplication"); }
    Part part; part.q0 = q0; part.q1 = q1;
    parts.push_back(part);
    q0 = q1;
  }
  u64 total = 0;
  for(Part& part : parts)
  {
    const u64 count = part.q1 - part.q0;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&part.d_off), (count + 1) * sizeof(u64)));
    Part* self = &part;
    ValuesProvider own = [self](u64 values) -> u64*
    {
      hipError_t e = hipMalloc(reinterpret_cast<void**>(&self->d_val), (values > 0 ? values : 1) * sizeof(u64));
      if(e != hipSuccess) { fail(GCSA2_ERR_OUT_OF_MEMORY, std::string("hipMalloc(values of a sub-batch): ") + hipGetErrorString(e)); return nullptr; }
      return self->d_val;
    };
    g_error.clear();
    rc = locate_chunk(ix, d_ranges + 2 * part.q0, count, sort, part.d_off, own, &part.total, stream, false);
    if(rc != GCSA2_OK) { return rc; }
    total += part.total;
  }
  *total_out = total;
  u64* out = values_for(total);
  if(out == nullptr) { return g_error.empty() ? fail(GCSA2_ERR_BUFFER_TOO_SMALL, "values buffer too small") : GCSA2_ERR_BUFFER_TOO_SMALL; }
  u64 base = 0;
  for(Part& part : parts)
  {
    const u64 count = part.q1 - part.q0;
    if(part.total > 0) { HIP_TRY(hipMemcpyAsync(out + base, part.d_val, part.total * sizeof(u64), hipMemcpyDeviceToDevice, stream)); }
    hipLaunchKernelGGL(k_shift_offsets, dim3(grid_for(count)), dim3(TPB), 0, stream, part.d_off, count, base, d_offsets + part.q0);
    base += part.total;
  }
  HIP_TRY(hipMemcpyAsync(d_offsets + nq, &total, sizeof(u64), hipMemcpyHostToDevice, stream));
  HIP_TRY(hipStreamSynchronize(stream));
  return GCSA2_OK;
}

int locate_checks(const gcsa2_index* ix, u64 nq)
{
  if(!ix->img.has_samples || !ix->img.has_counters)
  {
    return fail(GCSA2_ERR_MISSING_COMPONENT, "locate needs samples and counters (extra_pointers sizes the output)");
  }
  if(nq >= (u64(1) << 38)) { return fail(GCSA2_ERR_BUFFER_TOO_SMALL, "locate batch of >= 2^38 ranges; split the batch"); }      // (batches beyond 2^30 ranges run in sub-batches: locate_core)
  return GCSA2_OK;
}

}  // namespace

extern "C" {

int gcsa2_locate_device(const gcsa2_index* ix, const uint64_t* d_ranges, uint64_t nq, int sort, gcsa2_locate_job** job_out,
                        const uint64_t** d_offsets, const uint64_t** d_values, uint64_t* total_values, void* stream_)
{
  CHECK_INDEX(ix);
  if(job_out == nullptr) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "null job pointer"); }
  *job_out = nullptr;
  int rc = locate_checks(ix, nq);
  if(rc != GCSA2_OK) { return rc; }
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  DeviceGuard guard(ix->device);
  gcsa2_locate_job* job = new(std::nothrow) gcsa2_locate_job();
  if(job == nullptr) { return fail(GCSA2_ERR_OUT_OF_MEMORY, "host allocation failed"); }
  job->device = ix->device; job->nq = nq; job->stream = stream;
  struct Cleanup { gcsa2_locate_job*& j; bool armed = true; ~Cleanup() { if(armed) { gcsa2_locate_discard(j); j = nullptr; } } };
  Cleanup cleanup{job};

  // the job's buffers are read back by the host (locate_run / locate_fetch): plain hipMalloc, not pool memory
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&job->d_offsets), (nq + 1) * sizeof(u64)));
  ValuesProvider provide = [job](u64 count) -> u64*
  {
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&job->d_values), (count > 0 ? count : 1) * sizeof(u64));
    if(e != hipSuccess) { fail(GCSA2_ERR_OUT_OF_MEMORY, std::string("hipMalloc(values): ") + hipGetErrorString(e)); return nullptr; }
    return job->d_values;
  };
  g_error.clear();
  rc = locate_core(ix, d_ranges, nq, sort, job->d_offsets, provide, &job->total, stream);
  if(rc != GCSA2_OK) { return rc; }
  cleanup.armed = false; *job_out = job;
  if(d_offsets) { *d_offsets = job->d_offsets; }
  if(d_values) { *d_values = job->d_values; }
  if(total_values) { *total_values = job->total; }
  return GCSA2_OK;
}

int gcsa2_locate_into(const gcsa2_index* ix, const uint64_t* d_ranges, uint64_t nq, int sort, uint64_t* d_offsets,
                      uint64_t* d_values, uint64_t capacity, uint64_t* total_values, void* stream_)
{
  CHECK_INDEX(ix);
  if(d_offsets == nullptr || total_values == nullptr || (d_values == nullptr && capacity > 0))
  {
    return fail(GCSA2_ERR_INVALID_ARGUMENT, "null buffer");
  }
  int rc = locate_checks(ix, nq);
  if(rc != GCSA2_OK) { return rc; }
  DeviceGuard guard(ix->device);
  ValuesProvider provide = [d_values, capacity](u64 count) -> u64* { return count <= capacity ? d_values : nullptr; };
  g_error.clear();
  const auto t0 = std::chrono::steady_clock::now();
  rc = locate_core(ix, d_ranges, nq, sort, d_offsets, provide, total_values, static_cast<hipStream_t>(stream_), d_values, capacity);
  if(ix->tune.locate_trace)
  {
    std::fprintf(stderr, "[locate] gcsa2_locate_into: %.0f us\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
  }
  return rc;
}

// ---- host-pointer entry points: copy in, run, copy out, synchronise ----------------------

namespace {
// pattern offsets handed over by a host caller: non-decreasing, or the kernels would read outside the pattern buffer
inline bool offsets_ok(const uint64_t* offsets, uint64_t nq, u64* longest = nullptr)
{
  u64 bad = 0, most = 0;
  for(u64 q = 0; q < nq; q++)
  {
    const u64 len = offsets[q + 1] - offsets[q];
    bad |= u64(offsets[q + 1] < offsets[q]);
    most = (len > most ? len : most);
  }
  if(longest != nullptr) { *longest = most; }
  return bad == 0;
}
}

} // extern "C"

namespace {

// ---- large host batches: chunked, double-buffered host -> device -> host pipeline -----------------------------------------
// A batch of millions of patterns in pageable host memory used to go through ONE copy in, one kernel, one copy out: 10 M
// 32-mers took 25 ms, ten times the kernel (19 GB/s end to end).  Here PIPE_LANES host threads each take every
// PIPE_LANES-th chunk of PIPE_CHUNK_QUERIES patterns: copy the chunk into pinned memory (rebasing its offsets), enqueue
// H2D + k_find2 + D2H on the lane's own stream, and while that runs prepare the next chunk in the lane's other staging set;
// a set's results are copied to the caller's array when its event has fired.  Both PCIe directions, the kernel and the host
// copies overlap; what bounds the batch is the host's memcpy rate (56 bytes per 32-mer query through pinned memory).
// (lanes: tune.pipe_lanes, GCSA2_PIPE_LANES, 1..16, default 6; patterns per chunk: tune.pipe_chunk, GCSA2_PIPE_CHUNK = log2, default 18.
// Round 4 re-measured the shape on the headline index, one live image re-shaped with gcsa2_index_set_pipeline: 12 lanes x 2^17
// -- round 3's choice -- 1.4-1.8 G packed 32-mers/s, 6 lanes x 2^18 2.1-2.5 G; 0.9-1.0 -> 1.15-1.3 G for the bytes interface.  The
// boxes give a container 16 CPUs' worth of time: twelve spinning lanes plus the runtime's own threads run into that quota --
// with hipEventBlockingSync twelve lanes gain 20 %, six gain nothing; profiles/r04_host.md.)
constexpr u64 PIPE_CHUNK_BYTES = u64(8) << 20;     // per chunk: at most this many pattern bytes, and tune.pipe_chunk patterns (GCSA2_PIPE_CHUNK = log2, 15..20, default 18)
constexpr u64 PIPE_MIN_QUERIES = u64(1) << 19;                                       // smaller batches take the single-copy path
constexpr int PIPE_PATTERN_TOO_LONG = 1;                                             // internal: not a gcsa2_status

inline u64 pipe_set_bytes(u64 PIPE_CHUNK_QUERIES) { return (PIPE_CHUNK_BYTES + 64) + (PIPE_CHUNK_QUERIES + 8) * 8 + PIPE_CHUNK_QUERIES * 16; }      // patterns (+ 32 bytes of phase, + slack) | offsets | ranges

int pipe_prepare(const gcsa2_index* ix)
{
  if(!ix->pipe.empty()) { return GCSA2_OK; }
  const unsigned PIPE_LANES = ix->tune.pipe_lanes;
  std::vector<gcsa2_index::PipeLane> lanes(PIPE_LANES);
  hipError_t e = hipSuccess;
  for(gcsa2_index::PipeLane& lane : lanes)
  {
    if(e == hipSuccess) { e = hipStreamCreateWithFlags(&lane.stream, hipStreamNonBlocking); }
    if(e == hipSuccess) { e = hipStreamCreateWithFlags(&lane.down, hipStreamNonBlocking); }
    for(gcsa2_index::PipeSet& set : lane.set)
    {
      if(e == hipSuccess) { e = hipHostMalloc(reinterpret_cast<void**>(&set.h), pipe_set_bytes(ix->tune.pipe_chunk), hipHostMallocDefault); }
      if(e == hipSuccess) { e = hipMalloc(reinterpret_cast<void**>(&set.d), pipe_set_bytes(ix->tune.pipe_chunk)); }
      if(e == hipSuccess) { e = hipEventCreateWithFlags(&set.done, hipEventDisableTiming | (ix->tune.pipe_blocking ? hipEventBlockingSync : 0)); }
      if(e == hipSuccess) { e = hipEventCreateWithFlags(&set.computed, hipEventDisableTiming); }
    }
  }
  if(e != hipSuccess)
  {
    for(gcsa2_index::PipeLane& lane : lanes)
    {
      for(gcsa2_index::PipeSet& set : lane.set)
      {
        if(set.h) { (void)hipHostFree(set.h); } if(set.d) { (void)hipFree(set.d); } if(set.done) { (void)hipEventDestroy(set.done); }
        if(set.computed) { (void)hipEventDestroy(set.computed); }
      }
      if(lane.stream) { (void)hipStreamDestroy(lane.stream); }
      if(lane.down) { (void)hipStreamDestroy(lane.down); }
    }
    return fail(e == hipErrorOutOfMemory ? GCSA2_ERR_OUT_OF_MEMORY : GCSA2_ERR_HIP, std::string("host pipeline: ") + hipGetErrorString(e));
  }
  ix->pipe = std::move(lanes);
  return GCSA2_OK;
}

// The ranges of a chunk travel home as (sp, length) pairs in the narrowest exact form (comm.hpp: k_pack_ranges32 / 40): 8 bytes
// per query below 2^32 path nodes and edges, 10 below 2^40, else the 16 bytes of the u64 pairs.  GCSA2_PIPE_WIRE=16 keeps the wide
// form (A/B).  A lane widens them into the caller's array when it retires the chunk.
inline u64 pipe_wire_bytes(const gcsa2_index* ix)
{
  const u64 top = (ix->img.n > ix->img.e ? ix->img.n : ix->img.e);
  return (ix->tune.pipe_wide ? 16 : (top < (u64(1) << 32) ? 8 : (top < (u64(1) << 40) ? 10 : 16)));
}

inline void pipe_widen(const char* h, u64 count, u64 wire, u64* dst)
{
  if(wire == 10)                     // (sp, length) as five u16
  {
    const unsigned short* w = reinterpret_cast<const unsigned short*>(h);
    for(u64 i = 0; i < count; i++, w += 5)
    {
      const u64 sp = u64(w[0]) | (u64(w[1]) << 16) | (u64(w[4] & 0xFF) << 32), len = u64(w[2]) | (u64(w[3]) << 16) | (u64(w[4] >> 8) << 32);
      dst[2 * i] = sp; dst[2 * i + 1] = sp + len - 1;
    }
  }
  else                               // (sp, length) as u32 pairs
  {
    const u32* w = reinterpret_cast<const u32*>(h);
    for(u64 i = 0; i < count; i++) { dst[2 * i] = w[2 * i]; dst[2 * i + 1] = u64(w[2 * i]) + u64(w[2 * i + 1]) - 1; }
  }
}

int find_pipelined(const gcsa2_index* ix, const uint8_t* patterns, const uint64_t* offsets, uint64_t nq, uint64_t* ranges)
{
  std::lock_guard<std::mutex> hold(ix->pipe_lock);
  // patterns per chunk: the handle's setting, less for a batch too small to give every lane two chunks of that size
  u64 PIPE_CHUNK_QUERIES = ix->tune.pipe_chunk;
  { const u64 even = ((nq / (2 * u64(ix->tune.pipe_lanes)) + 63) & ~u64(63)), least = u64(1) << 15;
    if(even < PIPE_CHUNK_QUERIES) { PIPE_CHUNK_QUERIES = (even > least ? even : least); } }
  int rc = pipe_prepare(ix);
  if(rc != GCSA2_OK) { return rc; }
  // chunk boundaries: at most PIPE_CHUNK_QUERIES patterns and PIPE_CHUNK_BYTES pattern bytes each.  The offsets are validated
  // by the lanes, chunk by chunk, while they rebase them (a serial pass over 10 M offsets costs as much as the whole batch);
  // here only the boundaries are looked at, defensively.
  std::vector<u64> cut(1, 0);
  while(cut.back() < nq)
  {
    const u64 b = cut.back();
    u64 e = (nq - b < PIPE_CHUNK_QUERIES ? nq : b + PIPE_CHUNK_QUERIES);
    if(offsets[e] < offsets[b]) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "pattern offsets are not non-decreasing"); }
    if(offsets[e] - offsets[b] > PIPE_CHUNK_BYTES)
    {
      u64 lo = b, hi = e;                                    // largest e' with offsets[e'] - offsets[b] <= PIPE_CHUNK_BYTES
      while(lo < hi) { const u64 mid = (lo + hi + 1) / 2; if(offsets[mid] >= offsets[b] && offsets[mid] - offsets[b] <= PIPE_CHUNK_BYTES) { lo = mid; } else { hi = mid - 1; } }
      e = lo;
      if(e == b) { return PIPE_PATTERN_TOO_LONG; }           // the caller takes the single-copy path
    }
    cut.push_back(e);
  }
  const u64 chunks = cut.size() - 1;
  // A caller's buffer that is already page-locked (hipHostMalloc / hipHostRegister: a pinned torch tensor, the facade's own
  // arena) is handed to the copy engines where it lies; only pageable memory goes through the lanes' pinned staging sets.
  auto page_locked = [](const void* first, u64 bytes) -> bool
  {
    if(bytes == 0) { return false; }
    hipPointerAttribute_t a, b;
    const bool yes = hipPointerGetAttributes(&a, first) == hipSuccess && a.type == hipMemoryTypeHost &&
                     hipPointerGetAttributes(&b, static_cast<const char*>(first) + bytes - 1) == hipSuccess && b.type == hipMemoryTypeHost;
    (void)hipGetLastError();
    return yes;
  };
  const bool direct_pat = page_locked(patterns, offsets[nq] - offsets[0]), direct_off = page_locked(offsets, (nq + 1) * sizeof(u64)),
             direct_out = page_locked(ranges, 2 * nq * sizeof(u64));
  const unsigned PIPE_LANES = ix->tune.pipe_lanes;
  const bool split = ix->tune.pipe_split;
  // pageable result arrays are filled by the lanes anyway: their ranges come home in the narrow form (the pattern bytes of the
  // chunk have been consumed by then: their place in both staging buffers takes it); a page-locked array is written in place
  const u64 wire = (direct_out ? 16 : pipe_wire_bytes(ix));
  std::vector<int> status(PIPE_LANES, GCSA2_OK);
  std::vector<std::string> messages(PIPE_LANES);
  auto work = [&](unsigned t)
  {
    DeviceGuard guard(ix->device);
    gcsa2_index::PipeLane& lane = ix->pipe[t];
    auto fail_lane = [&](const char* what, hipError_t e) { status[t] = GCSA2_ERR_HIP; messages[t] = std::string(what) + ": " + hipGetErrorString(e); };
    auto retire = [&](gcsa2_index::PipeSet& set) -> bool        // wait for the set's chunk and hand its ranges to the caller
    {
      if(!set.busy) { return true; }
      hipError_t e = hipEventSynchronize(set.done);
      if(e != hipSuccess) { fail_lane("hipEventSynchronize", e); return false; }
      if(wire != 16) { pipe_widen(set.h, set.count, wire, ranges + 2 * set.first); }
      else if(!direct_out)
      {
        const char* h_out = set.h + (PIPE_CHUNK_BYTES + 64) + (PIPE_CHUNK_QUERIES + 8) * 8;
        std::memcpy(ranges + 2 * set.first, h_out, set.count * 16);
      }
      set.busy = false;
      return true;
    };
    unsigned turn = 0;
    for(u64 c = t; c < chunks && status[t] == GCSA2_OK; c += PIPE_LANES, turn ^= 1)
    {
      gcsa2_index::PipeSet& set = lane.set[turn];
      if(!retire(set)) { break; }
      const u64 b = cut[c], e = cut[c + 1], count = e - b, base = offsets[b], bytes = offsets[e] - base;
      char* h_pat = set.h; u64* h_off = reinterpret_cast<u64*>(set.h + (PIPE_CHUNK_BYTES + 64));
      char* h_out = reinterpret_cast<char*>(h_off + PIPE_CHUNK_QUERIES + 8);
      // the chunk's pattern bytes land at the same phase within 16 bytes as in the caller's array (the kernel reads aligned
      // words around a pattern's ends); with the caller's absolute offsets the kernel gets the pointer moved back by `base`
      const u64 phase = 16 + (base & 15);
      // patterns of one length (k-mer batches): the chunk's offsets are base + i * length, made on the device instead of sent
      const u64 stride = (offsets[b + 1] - base);
      u64 bad = 0, ragged = 0;
      if(direct_off) { for(u64 i = 1; i <= count; i++) { const u64 o = offsets[b + i]; bad |= u64(o < offsets[b + i - 1]); ragged |= (o - base) ^ (i * stride); } }
      else { for(u64 i = 0; i <= count; i++) { const u64 o = offsets[b + i]; h_off[i] = o - base; bad |= u64(i > 0 && o < offsets[b + i - 1]); ragged |= (o - base) ^ (i * stride); } }
      if(bad != 0 || bytes > PIPE_CHUNK_BYTES) { status[t] = GCSA2_ERR_INVALID_ARGUMENT; messages[t] = "pattern offsets are not non-decreasing"; break; }
      const bool uniform = (ragged == 0);
      char* d_pat = set.d; u64* d_off = reinterpret_cast<u64*>(set.d + (PIPE_CHUNK_BYTES + 64));
      u64* d_out = d_off + PIPE_CHUNK_QUERIES + 8;
      hipError_t err = hipSuccess;
      if(direct_pat) { err = hipMemcpyAsync(d_pat + phase, patterns + base, bytes, hipMemcpyHostToDevice, lane.stream); }
      else
      {
        std::memcpy(h_pat + phase, patterns + base, bytes);
        err = hipMemcpyAsync(d_pat, h_pat, (phase + bytes + 7) / 8 * 8, hipMemcpyHostToDevice, lane.stream);
      }
      if(err == hipSuccess && uniform)
      {
        hipLaunchKernelGGL(k_uniform_offsets, dim3(grid_for(count + 1)), dim3(TPB), 0, lane.stream, d_off, count + 1, stride);
        err = hipGetLastError();
      }
      else if(err == hipSuccess)
      {
        err = hipMemcpyAsync(d_off, direct_off ? offsets + b : h_off, (count + 1) * sizeof(u64), hipMemcpyHostToDevice, lane.stream);
      }
      if(err != hipSuccess) { fail_lane("hipMemcpyAsync", err); break; }
      const uint8_t* d_first = reinterpret_cast<const uint8_t*>(d_pat + phase) - (direct_off && !uniform ? base : 0);
      int rc_find = gcsa2_find_device(ix, d_first, d_off, count, d_out, lane.stream);
      if(rc_find != GCSA2_OK) { status[t] = rc_find; messages[t] = g_error; break; }
      hipStream_t back = lane.stream;
      if(split)
      {
        back = lane.down;
        err = hipEventRecord(set.computed, lane.stream);
        if(err == hipSuccess) { err = hipStreamWaitEvent(lane.down, set.computed, 0); }
      }
      if(err == hipSuccess && wire != 16)
      {
        const int rc_pack = (wire == 10 ? gcsa2_pack_ranges40_device(d_out, count, set.d, back) : gcsa2_pack_ranges32_device(d_out, count, reinterpret_cast<uint32_t*>(set.d), back));
        if(rc_pack != GCSA2_OK) { status[t] = rc_pack; messages[t] = g_error; break; }
        err = hipMemcpyAsync(set.h, set.d, count * wire, hipMemcpyDeviceToHost, back);
      }
      else if(err == hipSuccess) { err = hipMemcpyAsync(direct_out ? reinterpret_cast<char*>(ranges + 2 * b) : h_out, d_out, count * 16, hipMemcpyDeviceToHost, back); }
      if(err == hipSuccess) { err = hipEventRecord(set.done, back); }
      if(err != hipSuccess) { fail_lane("hipMemcpyAsync / hipEventRecord", err); break; }
      set.busy = true; set.first = b; set.count = count;
    }
    for(gcsa2_index::PipeSet& set : lane.set) { if(status[t] == GCSA2_OK) { (void)retire(set); } }
    if(status[t] != GCSA2_OK)           // nothing of this call may still be in flight when it returns
    {
      (void)hipStreamSynchronize(lane.stream); (void)hipStreamSynchronize(lane.down);
      for(gcsa2_index::PipeSet& set : lane.set) { set.busy = false; }
    }
  };
  Workers workers;
  const unsigned lanes = unsigned(chunks < PIPE_LANES ? chunks : PIPE_LANES);
  for(unsigned t = 1; t < lanes; t++) { workers.emplace_back(work, t); }
  work(0);
  workers.join();
  for(unsigned t = 0; t < lanes; t++) { if(status[t] != GCSA2_OK) { return fail(status[t], "pipeline lane " + std::to_string(t) + ": " + messages[t]); } }
  return GCSA2_OK;
}

// The same pipeline for patterns that arrive as 2-bit codes of one length (gcsa2_find_batch_packed): a chunk is its code words
// up the link (8 bytes per 32 characters), one launch, its ranges down; no offsets at all.
int find_packed_pipelined(const gcsa2_index* ix, const uint64_t* codes, u64 length, uint64_t nq, uint64_t* ranges)
{
  std::lock_guard<std::mutex> hold(ix->pipe_lock);
  // patterns per chunk: the handle's setting, less for a batch too small to give every lane two chunks of that size
  u64 PIPE_CHUNK_QUERIES = ix->tune.pipe_chunk;
  { const u64 even = ((nq / (2 * u64(ix->tune.pipe_lanes)) + 63) & ~u64(63)), least = u64(1) << 15;
    if(even < PIPE_CHUNK_QUERIES) { PIPE_CHUNK_QUERIES = (even > least ? even : least); } }
  int rc = pipe_prepare(ix);
  if(rc != GCSA2_OK) { return rc; }
  const u64 words = (length + 31) >> 5;
  u64 per_chunk = PIPE_CHUNK_BYTES / (8 * words);
  if(per_chunk > PIPE_CHUNK_QUERIES) { per_chunk = PIPE_CHUNK_QUERIES; }
  if(per_chunk == 0) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "packed patterns: one pattern exceeds a pipeline chunk"); }
  const u64 chunks = (nq + per_chunk - 1) / per_chunk;
  auto page_locked = [](const void* first, u64 bytes) -> bool
  {
    if(bytes == 0) { return false; }
    hipPointerAttribute_t a, b;
    const bool yes = hipPointerGetAttributes(&a, first) == hipSuccess && a.type == hipMemoryTypeHost &&
                     hipPointerGetAttributes(&b, static_cast<const char*>(first) + bytes - 1) == hipSuccess && b.type == hipMemoryTypeHost;
    (void)hipGetLastError();
    return yes;
  };
  const bool direct_in = page_locked(codes, nq * words * 8), direct_out = page_locked(ranges, 2 * nq * sizeof(u64));
  const unsigned PIPE_LANES = ix->tune.pipe_lanes;
  std::vector<int> status(PIPE_LANES, GCSA2_OK);
  std::vector<std::string> messages(PIPE_LANES);
  const u64 out_at = (PIPE_CHUNK_BYTES + 64) + (PIPE_CHUNK_QUERIES + 8) * 8;      // where a set keeps its ranges (pipe_set_bytes)
  // the ranges travel home as (sp, length) pairs in the narrowest exact form (comm.hpp): 8 bytes below 2^32 path nodes and
  // edges, 10 below 2^40, else the 16 bytes of the u64 pairs; GCSA2_PIPE_WIRE=16 keeps the wide form (A/B)
  const u64 wire = pipe_wire_bytes(ix);
  auto work = [&](unsigned t)
  {
    DeviceGuard guard(ix->device);
    gcsa2_index::PipeLane& lane = ix->pipe[t];
    auto fail_lane = [&](const char* what, hipError_t e) { status[t] = GCSA2_ERR_HIP; messages[t] = std::string(what) + ": " + hipGetErrorString(e); };
    auto retire = [&](gcsa2_index::PipeSet& set) -> bool
    {
      if(!set.busy) { return true; }
      hipError_t e = hipEventSynchronize(set.done);
      if(e != hipSuccess) { fail_lane("hipEventSynchronize", e); return false; }
      u64* dst = ranges + 2 * set.first;
      if(wire != 16) { pipe_widen(set.h, set.count, wire, dst); }
      else if(!direct_out) { std::memcpy(dst, set.h + out_at, set.count * 16); }
      set.busy = false;
      return true;
    };
    unsigned turn = 0;
    for(u64 c = t; c < chunks && status[t] == GCSA2_OK; c += PIPE_LANES, turn ^= 1)
    {
      gcsa2_index::PipeSet& set = lane.set[turn];
      if(!retire(set)) { break; }
      const u64 b = c * per_chunk, count = (nq - b < per_chunk ? nq - b : per_chunk), bytes = count * words * 8;
      hipError_t err = hipSuccess;
      if(direct_in) { err = hipMemcpyAsync(set.d, codes + b * words, bytes, hipMemcpyHostToDevice, lane.stream); }
      else
      {
        std::memcpy(set.h, codes + b * words, bytes);
        err = hipMemcpyAsync(set.d, set.h, bytes, hipMemcpyHostToDevice, lane.stream);
      }
      if(err != hipSuccess) { fail_lane("hipMemcpyAsync", err); break; }
      u64* d_out = reinterpret_cast<u64*>(set.d + out_at);
      int rc_find = gcsa2_find_packed_device(ix, reinterpret_cast<const u64*>(set.d), length, count, d_out, lane.stream);
      if(rc_find != GCSA2_OK) { status[t] = rc_find; messages[t] = g_error; break; }
      if(wire == 16) { err = hipMemcpyAsync(direct_out ? reinterpret_cast<char*>(ranges + 2 * b) : set.h + out_at, d_out, count * 16, hipMemcpyDeviceToHost, lane.stream); }
      else
      {
        // the code words have been consumed: their place takes the narrow form of the ranges, which is what travels
        int rc_pack = (wire == 10 ? gcsa2_pack_ranges40_device(d_out, count, set.d, lane.stream) : gcsa2_pack_ranges32_device(d_out, count, reinterpret_cast<uint32_t*>(set.d), lane.stream));
        if(rc_pack != GCSA2_OK) { status[t] = rc_pack; messages[t] = g_error; break; }
        err = hipMemcpyAsync(set.h, set.d, count * wire, hipMemcpyDeviceToHost, lane.stream);
      }
      if(err == hipSuccess) { err = hipEventRecord(set.done, lane.stream); }
      if(err != hipSuccess) { fail_lane("hipMemcpyAsync / hipEventRecord", err); break; }
      set.busy = true; set.first = b; set.count = count;
    }
    for(gcsa2_index::PipeSet& set : lane.set) { if(status[t] == GCSA2_OK) { (void)retire(set); } }
    if(status[t] != GCSA2_OK)
    {
      (void)hipStreamSynchronize(lane.stream);
      for(gcsa2_index::PipeSet& set : lane.set) { set.busy = false; }
    }
  };
  Workers workers;
  const unsigned lanes = unsigned(chunks < PIPE_LANES ? chunks : PIPE_LANES);
  for(unsigned t = 1; t < lanes; t++) { workers.emplace_back(work, t); }
  work(0);
  workers.join();
  for(unsigned t = 0; t < lanes; t++) { if(status[t] != GCSA2_OK) { return fail(status[t], "pipeline lane " + std::to_string(t) + ": " + messages[t]); } }
  return GCSA2_OK;
}

}  // namespace

extern "C" {

// find() of `nq` patterns of `pattern_length` characters each, given as 2-bit codes in host memory (layout: gcsa2_hip.h)
int gcsa2_find_batch_packed(const gcsa2_index* ix, const uint64_t* codes, uint64_t pattern_length, uint64_t nq, uint64_t* ranges)
{
  CHECK_INDEX(ix);
  if(nq == 0) { return GCSA2_OK; }
  if(codes == nullptr || ranges == nullptr) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "null buffer"); }
  if(pattern_length == 0 || pattern_length >= (u64(1) << 32)) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "packed patterns: the common length must be 1 .. 2^32 - 1"); }
  try
  {
    // (patterns whose codes alone exceed a pipeline chunk -- 32 M characters -- take the single copy below)
    if(nq >= PIPE_MIN_QUERIES / 4 && ((pattern_length + 31) >> 5) * 8 <= PIPE_CHUNK_BYTES) { return find_packed_pipelined(ix, codes, pattern_length, nq, ranges); }
    // small batches: one copy in, one launch, one copy out
    DeviceGuard guard(ix->device);
    const u64 words = (pattern_length + 31) >> 5;
    void* d = nullptr;
    HIP_TRY(hipMalloc(&d, nq * words * 8 + nq * 16));
    u64* d_codes = static_cast<u64*>(d); u64* d_out = d_codes + nq * words;
    hipError_t e = hipMemcpy(d_codes, codes, nq * words * 8, hipMemcpyHostToDevice);
    int rc = (e == hipSuccess ? gcsa2_find_packed_device(ix, d_codes, pattern_length, nq, d_out, nullptr) : fail(GCSA2_ERR_HIP, hipGetErrorString(e)));
    if(rc == GCSA2_OK) { e = hipMemcpy(ranges, d_out, nq * 16, hipMemcpyDeviceToHost); if(e != hipSuccess) { rc = fail(GCSA2_ERR_HIP, hipGetErrorString(e)); } }
    (void)hipFree(d);
    return rc;
  }
  catch(const std::exception& e) { return fail(GCSA2_ERR_OUT_OF_MEMORY, std::string("gcsa2_find_batch_packed: ") + e.what()); }
}

int gcsa2_find_batch(const gcsa2_index* ix, const uint8_t* patterns, const uint64_t* offsets, uint64_t nq, uint64_t* ranges)
{
  CHECK_INDEX(ix);
  if(nq == 0) { return GCSA2_OK; }
  if(offsets == nullptr || ranges == nullptr) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "null buffer"); }
  if(nq >= PIPE_MIN_QUERIES)
  {
    try
    {
      const int rc = find_pipelined(ix, patterns, offsets, nq, ranges);
      if(rc != PIPE_PATTERN_TOO_LONG) { return rc; }
    }
    catch(const std::exception& e) { return fail(GCSA2_ERR_OUT_OF_MEMORY, std::string("gcsa2_find_batch: ") + e.what()); }
  }
  if(!offsets_ok(offsets, nq)) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "pattern offsets are not non-decreasing"); }
  DeviceGuard guard(ix->device);
  const u64 total = offsets[nq];
  Lease lease(ix);
  HIP_TRY(lease.begin(Lease::need(total + 16) + Lease::need((nq + 1) * 8) + Lease::need(2 * nq * 8), true));
  u8* d_pat = lease.dev<u8>(total + 16); u64* d_off = lease.dev<u64>(nq + 1); u64* d_out = lease.dev<u64>(2 * nq);
  HIP_TRY(lease.up(d_pat, patterns, total));
  HIP_TRY(lease.up(d_off, offsets, (nq + 1) * sizeof(u64)));
  int rc = gcsa2_find_device(ix, d_pat, d_off, nq, d_out, lease.stream());
  if(rc != GCSA2_OK) { return rc; }
  HIP_TRY(lease.down(ranges, d_out, 2 * nq * sizeof(u64)));
  HIP_TRY(lease.finish());
  return GCSA2_OK;
}

int gcsa2_lf_batch(const gcsa2_index* ix, const uint64_t* in, const uint8_t* comps, uint64_t nq, uint64_t* out)
{
  CHECK_INDEX(ix);
  if(nq == 0) { return GCSA2_OK; }
  DeviceGuard guard(ix->device);
  if(nq == 1)                                // the per-character caller (gcsa.h:155-162 in a loop): through the resident wavefront
  {
    u64 r[2];
    const int rc = mailbox_call(ix, MAIL_LF, in[0], in[1], comps[0], r, 2);
    if(rc == GCSA2_OK) { out[0] = r[0]; out[1] = r[1]; return GCSA2_OK; }
    if(rc != MAIL_UNAVAILABLE) { return rc; }
  }
  Lease lease(ix);
  HIP_TRY(lease.begin(2 * Lease::need(2 * nq * 8) + Lease::need(nq), true));
  u64* d_in = lease.dev<u64>(2 * nq); u64* d_out = lease.dev<u64>(2 * nq); u8* d_c = lease.dev<u8>(nq);
  HIP_TRY(lease.up(d_in, in, 2 * nq * sizeof(u64)));
  HIP_TRY(lease.up(d_c, comps, nq));
  int rc = gcsa2_lf_device(ix, d_in, d_c, nq, d_out, lease.stream());
  if(rc != GCSA2_OK) { return rc; }
  HIP_TRY(lease.down(out, d_out, 2 * nq * sizeof(u64)));
  HIP_TRY(lease.finish());
  return GCSA2_OK;
}

int gcsa2_lf_node_batch(const gcsa2_index* ix, const uint64_t* in, uint64_t nq, uint64_t* out)
{
  CHECK_INDEX(ix);
  if(nq == 0) { return GCSA2_OK; }
  DeviceGuard guard(ix->device);
  if(nq == 1)
  {
    const int rc = mailbox_call(ix, MAIL_LF_NODE, in[0], 0, 0, out, 1);
    if(rc != MAIL_UNAVAILABLE) { return rc; }
  }
  Lease lease(ix);
  HIP_TRY(lease.begin(2 * Lease::need(nq * 8), true));
  u64* d_in = lease.dev<u64>(nq); u64* d_out = lease.dev<u64>(nq);
  HIP_TRY(lease.up(d_in, in, nq * sizeof(u64)));
  hipLaunchKernelGGL(k_lf_node, dim3(grid_for(nq)), dim3(TPB), 0, lease.stream(), ix->img, d_in, nq, d_out);
  LAUNCH_CHECK("k_lf_node");
  HIP_TRY(lease.down(out, d_out, nq * sizeof(u64)));
  HIP_TRY(lease.finish());
  return GCSA2_OK;
}

int gcsa2_char_range(const gcsa2_index* ix, uint8_t comp, uint64_t* sp, uint64_t* ep)
{
  CHECK_INDEX(ix);
  if(comp >= ix->img.sigma) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "comp >= sigma"); }
  // charRange(comp) = find() of a one-character pattern whose byte maps to comp
  int byte = -1;
  for(int b = 0; b < 256; b++) { if(ix->img.char2comp[b] == comp) { byte = b; break; } }
  if(byte < 0) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "no byte maps to this comp"); }
  uint8_t pat = uint8_t(byte);
  uint64_t off[2] = {0, 1}, rng[2];
  int rc = gcsa2_find_batch(ix, &pat, off, 1, rng);
  if(rc != GCSA2_OK) { return rc; }
  *sp = rng[0]; *ep = rng[1];
  return GCSA2_OK;
}

int gcsa2_lf_all_batch(const gcsa2_index* ix, const uint64_t* in, uint64_t nq, int all, uint64_t* out)
{
  CHECK_INDEX(ix);
  if(nq == 0) { return GCSA2_OK; }
  DeviceGuard guard(ix->device);
  const u64 sigma = ix->img.sigma;
  Lease lease(ix);
  HIP_TRY(lease.begin(Lease::need(2 * nq * 8) + Lease::need(2 * nq * sigma * 8), true));
  u64* d_in = lease.dev<u64>(2 * nq); u64* d_out = lease.dev<u64>(2 * nq * sigma);
  HIP_TRY(lease.up(d_in, in, 2 * nq * sizeof(u64)));
  hipLaunchKernelGGL(k_lf_all, dim3(grid_for(nq)), dim3(TPB), 0, lease.stream(), ix->img, d_in, nq, all, d_out);
  LAUNCH_CHECK("k_lf_all");
  HIP_TRY(lease.down(out, d_out, 2 * nq * sigma * sizeof(u64)));
  HIP_TRY(lease.finish());
  return GCSA2_OK;
}

int gcsa2_count_batch(const gcsa2_index* ix, const uint64_t* ranges, uint64_t nq, uint64_t* counts)
{
  CHECK_INDEX(ix);
  if(nq == 0) { return GCSA2_OK; }
  DeviceGuard guard(ix->device);
  if(nq == 1 && ix->img.has_counters)
  {
    const int rc1 = mailbox_call(ix, MAIL_COUNT, ranges[0], ranges[1], 0, counts, 1);
    if(rc1 != MAIL_UNAVAILABLE) { return rc1; }
  }
  Lease lease(ix);
  HIP_TRY(lease.begin(Lease::need(2 * nq * 8) + Lease::need(nq * 8), true));
  u64* d_in = lease.dev<u64>(2 * nq); u64* d_out = lease.dev<u64>(nq);
  HIP_TRY(lease.up(d_in, ranges, 2 * nq * sizeof(u64)));
  int rc = gcsa2_count_device(ix, d_in, nq, d_out, lease.stream());
  if(rc != GCSA2_OK) { return rc; }
  HIP_TRY(lease.down(counts, d_out, nq * sizeof(u64)));
  HIP_TRY(lease.finish());
  return GCSA2_OK;
}

int gcsa2_locate_run(const gcsa2_index* ix, const uint64_t* ranges, uint64_t nq, int sort, uint64_t* offsets, gcsa2_locate_job** job)
{
  CHECK_INDEX(ix);
  if(offsets == nullptr || job == nullptr) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "null buffer"); }
  DeviceGuard guard(ix->device);
  Lease lease(ix);
  HIP_TRY(lease.begin(Lease::need(2 * nq * 8)));
  u64* d_in = lease.dev<u64>(2 * nq);
  HIP_TRY(lease.up(d_in, ranges, 2 * nq * sizeof(u64)));
  const u64* d_off = nullptr;
  int rc = gcsa2_locate_device(ix, d_in, nq, sort, job, &d_off, nullptr, nullptr, lease.stream());
  if(rc != GCSA2_OK) { return rc; }
  HIP_TRY(lease.down(offsets, d_off, (nq + 1) * sizeof(u64)));      // in the job's stream order
  HIP_TRY(lease.finish());
  (*job)->stream = nullptr;       // the lease's stream belongs to the index; the job is complete and may outlive it
  return GCSA2_OK;
}

int gcsa2_locate_fetch(gcsa2_locate_job* job, uint64_t* values, uint64_t capacity)
{
  if(job == nullptr) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "null job"); }
  if(capacity < job->total) { return fail(GCSA2_ERR_BUFFER_TOO_SMALL, "values buffer smaller than offsets[n_queries]"); }
  {
    DeviceGuard guard(job->device);
    if(job->total > 0)
    {
      HIP_TRY(hipMemcpyAsync(values, job->d_values, job->total * sizeof(u64), hipMemcpyDeviceToHost, job->stream));
      HIP_TRY(hipStreamSynchronize(job->stream));
    }
  }
  gcsa2_locate_discard(job);
  return GCSA2_OK;
}

int gcsa2_parent_batch(const gcsa2_index* ix, const uint64_t* ranges, uint64_t nq, gcsa2_stnode* nodes)
{
  CHECK_INDEX(ix);
  if(!ix->img.has_lcp) { return fail(GCSA2_ERR_MISSING_COMPONENT, "index was created without an LCP array"); }
  if(nq == 0) { return GCSA2_OK; }
  static_assert(sizeof(gcsa2_stnode) == 5 * sizeof(u64), "gcsa2_stnode is five packed u64");
  if(nq == 1)
  {
    DeviceGuard guard(ix->device);
    const int rc = mailbox_call(ix, MAIL_PARENT, ranges[0], ranges[1], 0, reinterpret_cast<uint64_t*>(nodes), 5);
    if(rc != MAIL_UNAVAILABLE) { return rc; }
  }
  return simple_batch(ix, ranges, 2, reinterpret_cast<uint64_t*>(nodes), 5, nq, "k_parent", [&](u64* d_in, u64* d_out, hipStream_t st)
  { hipLaunchKernelGGL(k_parent, dim3(grid_for(nq)), dim3(TPB), 0, st, ix->img, d_in, nq, reinterpret_cast<gcsa2_stnode*>(d_out)); });
}

int gcsa2_depth_batch(const gcsa2_index* ix, const uint64_t* ranges, uint64_t nq, uint64_t* depths)
{
  CHECK_INDEX(ix);
  if(!ix->img.has_lcp) { return fail(GCSA2_ERR_MISSING_COMPONENT, "index was created without an LCP array"); }
  if(nq == 0) { return GCSA2_OK; }
  return simple_batch(ix, ranges, 2, depths, 1, nq, "k_depth", [&](u64* d_in, u64* d_out, hipStream_t st)
  { hipLaunchKernelGGL(k_depth, dim3(grid_for(nq)), dim3(TPB), 0, st, ix->img, d_in, nq, d_out); });
}

int gcsa2_sv_batch(const gcsa2_index* ix, int op, const uint64_t* positions, uint64_t nq, uint64_t* results)
{
  CHECK_INDEX(ix);
  if(!ix->img.has_lcp) { return fail(GCSA2_ERR_MISSING_COMPONENT, "index was created without an LCP array"); }
  if(op < 0 || op > 3) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "op must be 0..3"); }
  if(nq == 0) { return GCSA2_OK; }
  return simple_batch(ix, positions, 1, results, 2, nq, "k_sv", [&](u64* d_in, u64* d_out, hipStream_t st)
  { hipLaunchKernelGGL(k_sv, dim3(grid_for(nq)), dim3(TPB), 0, st, ix->img, op, d_in, nq, d_out); });
}

int gcsa2_rmq_batch(const gcsa2_index* ix, const uint64_t* ranges, uint64_t nq, uint64_t* results)
{
  CHECK_INDEX(ix);
  if(!ix->img.has_lcp) { return fail(GCSA2_ERR_MISSING_COMPONENT, "index was created without an LCP array"); }
  if(nq == 0) { return GCSA2_OK; }
  return simple_batch(ix, ranges, 2, results, 2, nq, "k_rmq", [&](u64* d_in, u64* d_out, hipStream_t st)
  { hipLaunchKernelGGL(k_rmq, dim3(grid_for(nq)), dim3(TPB), 0, st, ix->img, d_in, nq, d_out); });
}

int gcsa2_sample_range_batch(const gcsa2_index* ix, const uint64_t* nodes, uint64_t nq, uint64_t* out)
{
  CHECK_INDEX(ix);
  if(!ix->img.has_samples) { return fail(GCSA2_ERR_MISSING_COMPONENT, "index was created without samples"); }
  if(nq == 0) { return GCSA2_OK; }
  return simple_batch(ix, nodes, 1, out, 3, nq, "k_sample_range", [&](u64* d_in, u64* d_out, hipStream_t st)
  { hipLaunchKernelGGL(k_sample_range, dim3(grid_for(nq)), dim3(TPB), 0, st, ix->img, d_in, nq, d_out); });
}

int gcsa2_sample_batch(const gcsa2_index* ix, const uint64_t* idx, uint64_t nq, uint64_t* values, uint8_t* last)
{
  CHECK_INDEX(ix);
  if(!ix->img.has_samples) { return fail(GCSA2_ERR_MISSING_COMPONENT, "index was created without samples"); }
  if(nq == 0) { return GCSA2_OK; }
  DeviceGuard guard(ix->device);
  Lease lease(ix);
  HIP_TRY(lease.begin(2 * Lease::need(nq * 8) + Lease::need(nq), true));
  u64* d_in = lease.dev<u64>(nq); u64* d_val = lease.dev<u64>(nq); u8* d_last = lease.dev<u8>(nq);
  HIP_TRY(lease.up(d_in, idx, nq * sizeof(u64)));
  hipLaunchKernelGGL(k_sample, dim3(grid_for(nq)), dim3(TPB), 0, lease.stream(), ix->img, d_in, nq, d_val, d_last);
  LAUNCH_CHECK("k_sample");
  HIP_TRY(lease.down(values, d_val, nq * sizeof(u64)));
  HIP_TRY(lease.down(last, d_last, nq));
  HIP_TRY(lease.finish());
  return GCSA2_OK;
}

uint64_t gcsa2_sampled_positions(const gcsa2_index* ix) { return ix->img.has_samples ? ix->img.sampled.ones : 0; }
uint64_t gcsa2_lcp_size(const gcsa2_index* ix) { return ix->img.lcp_size; }
uint64_t gcsa2_lcp_values(const gcsa2_index* ix) { return ix->img.lcp_values; }
uint64_t gcsa2_lcp_levels(const gcsa2_index* ix) { return ix->img.lcp_levels; }
uint64_t gcsa2_lcp_branching(const gcsa2_index* ix) { return ix->img.lcp_branching; }
uint64_t gcsa2_sigma(const gcsa2_index* ix) { return ix->img.sigma; }
uint64_t gcsa2_fast_chars(const gcsa2_index* ix) { return ix->img.fast_chars; }
void gcsa2_derive_comp2char(const uint8_t* char2comp, uint64_t sigma, uint8_t* comp2char)
{
  if(char2comp == nullptr || comp2char == nullptr) { return; }
  for(u64 c = 0; c < sigma; c++)
  {
    int first = -1, upper = -1;
    for(int b = 0; b < 256; b++)
    {
      if(char2comp[b] != c) { continue; }
      if(first < 0) { first = b; }
      if(upper < 0 && !(b >= 'a' && b <= 'z') && b != 0) { upper = b; }
    }
    comp2char[c] = u8(upper >= 0 ? upper : (first >= 0 ? first : 0));
  }
  if(sigma == 7)
  {
    const char* dflt = "$ACGTN#";
    bool is_default = true;
    for(u64 c = 0; c < 7; c++) { is_default = is_default && char2comp[u8(dflt[c])] == c; }
    if(is_default) { for(u64 c = 0; c < 7; c++) { comp2char[c] = u8(dflt[c]); } }
  }
}

void gcsa2_alphabet(const gcsa2_index* ix, uint8_t* char2comp, uint64_t* C)
{
  if(char2comp) { std::memcpy(char2comp, ix->img.char2comp, 256); }
  if(C) { for(u64 c = 0; c <= ix->img.sigma; c++) { C[c] = ix->img.C[c]; } }
}

int gcsa2_lcp_access_batch(const gcsa2_index* ix, const uint64_t* positions, uint64_t nq, uint64_t* out)
{
  CHECK_INDEX(ix);
  if(!ix->img.has_lcp) { return fail(GCSA2_ERR_MISSING_COMPONENT, "index was created without an LCP array"); }
  if(nq == 0) { return GCSA2_OK; }
  return simple_batch(ix, positions, 1, out, 1, nq, "k_lcp_access", [&](u64* d_in, u64* d_out, hipStream_t st)
  { hipLaunchKernelGGL(k_lcp_access, dim3(grid_for(nq)), dim3(TPB), 0, st, ix->img, d_in, nq, d_out); });
}

// GCSA::locate(range, max_positions, results) (src/gcsa.cpp:844-878): host control flow with the
// reference's std::mt19937_64 draws; every locateInternal() runs on the device.  The reference's
// unordered_set is replaced by a sorted unique vector: its iteration order never reaches the
// output, because deterministicShuffle sorts first (utils.h:359-370) and the result is sorted last.
int gcsa2_locate_max(const gcsa2_index* ix, uint64_t sp, uint64_t ep, uint64_t max_positions,
                     uint64_t* values, uint64_t capacity, uint64_t* count_out)
{
  CHECK_INDEX(ix);
  if(count_out == nullptr) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "null count pointer"); }
  *count_out = 0;
  try {   // no C++ exception may cross the C boundary
  uint64_t range[2] = {sp, ep}, total = 0;
  int rc = gcsa2_count_batch(ix, range, 1, &total);
  if(rc != GCSA2_OK) { return rc; }
  if(total == 0) { return GCSA2_OK; }
  if(max_positions > total) { max_positions = total; }
  std::mt19937_64 rng(sp ^ ep);
  std::vector<u64> results;
  auto locate_ranges = [&](const std::vector<u64>& rr, std::vector<u64>& dst) -> int
  {
    u64 nq = rr.size() / 2;
    std::vector<u64> offs(nq + 1);
    gcsa2_locate_job* job = nullptr;
    int r = gcsa2_locate_run(ix, rr.data(), nq, 1, offs.data(), &job);
    if(r != GCSA2_OK) { return r; }
    dst.resize(offs[nq]);
    return gcsa2_locate_fetch(job, dst.data(), dst.size());
  };
  if(max_positions >= total / 2)   // just locate everything (gcsa.cpp:855-858)
  {
    rc = locate_ranges({sp, ep}, results);
    if(rc != GCSA2_OK) { return rc; }
  }
  else                             // random positions until enough distinct values (gcsa.cpp:859-871)
  {
    std::vector<u64> found, tmp;
    while(found.size() < max_positions)
    {
      u64 pos = sp + rng() % (ep + 1 - sp);
      rc = locate_ranges({pos, pos}, tmp);
      if(rc != GCSA2_OK) { return rc; }
      found.insert(found.end(), tmp.begin(), tmp.end());
      std::sort(found.begin(), found.end());
      found.erase(std::unique(found.begin(), found.end()), found.end());
    }
    results.swap(found);
  }
  if(results.size() > max_positions)   // deterministicShuffle + truncate (gcsa.cpp:873-877)
  {
    std::sort(results.begin(), results.end());
    for(u64 i = results.size(); i > 0; i--) { std::swap(results[i - 1], results[rng() % i]); }
    results.resize(max_positions);
  }
  std::sort(results.begin(), results.end());
  if(results.size() > capacity) { *count_out = results.size(); return fail(GCSA2_ERR_BUFFER_TOO_SMALL, "values buffer too small"); }
  std::memcpy(values, results.data(), results.size() * sizeof(u64));
  *count_out = results.size();
  return GCSA2_OK;
  } catch(const std::exception& e) { return fail(GCSA2_ERR_OUT_OF_MEMORY, std::string("gcsa2_locate_max: ") + e.what()); }
}

}  // extern "C"

namespace {
int match_stats_launch(const gcsa2_index* ix, int variant, const uint8_t* d_patterns, const uint64_t* d_offsets, uint64_t nq, u64 total_bytes,
                       uint16_t* d_ms, uint64_t* d_ranges, uint64_t* d_fallbacks, hipStream_t st, const BreakSink* sink = nullptr);
int group_comm_init(gcsa2_group* g);
int match_breaks_pieced(const gcsa2_index* ix, const uint8_t* patterns, const uint64_t* offsets, uint64_t nq, uint64_t min_length,
                        uint64_t* break_offsets, gcsa2_break* breaks, uint64_t capacity, uint64_t* total_breaks, uint64_t* ranges, uint64_t* fallbacks);
// (large host batches of matching statistics / break points go in pieces of tune.ms_piece_bytes -- GCSA2_MS_PIECE_MB, 32 MB -- from two pieces' worth on)

}  // namespace

#include "comm.hpp"

extern "C" {

// ---- single-process multi-GPU: replicated index, contiguous query shards ---------------------
// Queries are independent and the index is read-only (the reference's only data-parallel query
// path is the static split of verifyIndex, src/algorithms.cpp:106-114), so every device gets a
// replica and a contiguous shard.  gcsa2_group_find_batch (host buffers): one host thread per device
// stages its shard, runs k_find2 and copies its ranges straight into the caller's buffer.
// gcsa2_group_find_device (device buffers): every device searches its shard on its own stream and the
// ranges are gathered in the root's HBM with one grouped RCCL send / recv over xGMI (comm.hpp).

struct gcsa2_group
{
  std::vector<gcsa2_index*> replicas;
  std::vector<hipStream_t> streams;        // one per replica, on its device
  std::vector<ncclComm_t> comms;           // ncclCommInitAll over the device list; empty = peer copies instead
  std::vector<u64*> scratch;               // per replica: ranges of its shard before the gather (replicas 1..)
  std::vector<u64> scratch_pairs;
  bool comm_tried = false;
  std::mutex lock;                         // group_find_device calls are serialized (they share streams and scratch)
};

int gcsa2_group_create(const gcsa2_host_view* view, const int* devices, int n_devices, gcsa2_group** out)
{
  if(view == nullptr || devices == nullptr || out == nullptr || n_devices <= 0) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "bad group arguments"); }
  *out = nullptr;
  gcsa2_group* g = new(std::nothrow) gcsa2_group();
  if(g == nullptr) { return fail(GCSA2_ERR_OUT_OF_MEMORY, "host allocation failed"); }
  for(int i = 0; i < n_devices; i++)
  {
    gcsa2_index* ix = nullptr;
    int rc = gcsa2_index_create(view, devices[i], &ix);
    hipStream_t st = nullptr;
    if(rc == GCSA2_OK)
    {
      DeviceGuard guard(devices[i]);
      if(hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { rc = fail(GCSA2_ERR_HIP, "hipStreamCreate failed"); gcsa2_index_destroy(ix); }
    }
    if(rc != GCSA2_OK)
    {
      std::string msg = g_error;
      gcsa2_group_destroy(g);
      return fail(rc, msg);
    }
    g->replicas.push_back(ix); g->streams.push_back(st);
    g->scratch.push_back(nullptr); g->scratch_pairs.push_back(0);
  }
  *out = g;
  return GCSA2_OK;
}

void gcsa2_group_destroy(gcsa2_group* g)
{
  if(g == nullptr) { return; }
  for(size_t i = 0; i < g->comms.size(); i++)
  {
    if(g->comms[i] != nullptr) { DeviceGuard guard(g->replicas[i]->device); (void)rccl().CommDestroy(g->comms[i]); }
  }
  for(size_t i = 0; i < g->replicas.size(); i++)
  {
    DeviceGuard guard(g->replicas[i]->device);
    if(i < g->streams.size() && g->streams[i] != nullptr) { (void)hipStreamDestroy(g->streams[i]); }
    if(i < g->scratch.size() && g->scratch[i] != nullptr) { (void)hipFree(g->scratch[i]); }
  }
  for(gcsa2_index* r : g->replicas) { gcsa2_index_destroy(r); }
  delete g;
}

int gcsa2_group_uses_rccl(const gcsa2_group* g) { return (g != nullptr && !g->comms.empty()) ? 1 : 0; }

int gcsa2_group_find_device(gcsa2_group* g, const uint8_t* const* d_patterns, const uint64_t* const* d_offsets,
                            const uint64_t* counts, uint64_t* d_ranges_root)
{
  if(g == nullptr || g->replicas.empty()) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "null or empty group"); }
  if(d_patterns == nullptr || d_offsets == nullptr || counts == nullptr || d_ranges_root == nullptr) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "null buffer"); }
  try {
  std::lock_guard<std::mutex> hold(g->lock);
  const int G = int(g->replicas.size());
  { int rc = group_comm_init(g); if(rc != GCSA2_OK) { return rc; } }
  // every replica searches its shard on its own stream; replica 0 writes straight into the result buffer
  std::vector<u64> first(size_t(G) + 1, 0);
  for(int r = 0; r < G; r++) { first[size_t(r) + 1] = first[size_t(r)] + counts[r]; }
  for(int r = 0; r < G; r++)
  {
    if(counts[r] == 0) { continue; }
    u64* dst = d_ranges_root;
    if(r > 0)
    {
      DeviceGuard guard(g->replicas[r]->device);
      if(g->scratch_pairs[r] < counts[r])
      {
        if(g->scratch[r] != nullptr) { (void)hipFree(g->scratch[r]); g->scratch[r] = nullptr; g->scratch_pairs[r] = 0; }
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&g->scratch[r]), counts[r] * 2 * sizeof(u64)));
        g->scratch_pairs[r] = counts[r];
      }
      dst = g->scratch[r];
    }
    int rc = gcsa2_find_device(g->replicas[r], d_patterns[r], d_offsets[r], counts[r], dst, g->streams[r]);
    if(rc != GCSA2_OK) { return rc; }
  }
  // the gather: ranges of replica r land at d_ranges_root + 2 * first[r]
  if(!g->comms.empty())
  {
    RcclApi& api = rccl();
    RCCL_TRY(api.GroupStart());
    ncclResult_t res = ncclSuccess;
    for(int r = 1; r < G && res == ncclSuccess; r++)
    {
      if(counts[r] == 0) { continue; }
      res = api.Send(g->scratch[r], counts[r] * 2, ncclUint64, 0, g->comms[r], g->streams[r]);
      if(res == ncclSuccess) { res = api.Recv(d_ranges_root + 2 * first[r], counts[r] * 2, ncclUint64, r, g->comms[0], g->streams[0]); }
    }
    ncclResult_t end = api.GroupEnd();
    if(res != ncclSuccess) { return fail(GCSA2_ERR_HIP, std::string("ncclSend / ncclRecv: ") + api.GetErrorString(res)); }
    if(end != ncclSuccess) { return fail(GCSA2_ERR_HIP, std::string("ncclGroupEnd: ") + api.GetErrorString(end)); }
  }
  else
  {
    for(int r = 1; r < G; r++)
    {
      if(counts[r] == 0) { continue; }
      DeviceGuard guard(g->replicas[r]->device);
      HIP_TRY(hipMemcpyPeerAsync(d_ranges_root + 2 * first[r], g->replicas[0]->device, g->scratch[r], g->replicas[r]->device,
                                 counts[r] * 2 * sizeof(u64), g->streams[r]));
    }
  }
  for(int r = 0; r < G; r++)
  {
    DeviceGuard guard(g->replicas[r]->device);
    HIP_TRY(hipStreamSynchronize(g->streams[r]));
  }
  return GCSA2_OK;
  } catch(const std::exception& e) { return fail(GCSA2_ERR_OUT_OF_MEMORY, std::string("gcsa2_group_find_device: ") + e.what()); }
}

}  // extern "C"

namespace {

// The gather of the group entry points: src[r] (bytes[r] bytes, on the device of replica r, produced on streams[r]) lands at
// dst_root + at[r] on the device of replica 0.  One grouped RCCL send / recv when the group has a communicator, peer copies
// otherwise.  Replica 0's own part is a device copy on its stream (skipped when it is already in place).
int group_gather(gcsa2_group* g, const std::vector<const void*>& src, const std::vector<u64>& bytes, char* dst_root, const std::vector<u64>& at)
{
  const int G = int(g->replicas.size());
  if(bytes[0] > 0 && src[0] != dst_root + at[0])
  {
    DeviceGuard guard(g->replicas[0]->device);
    HIP_TRY(hipMemcpyAsync(dst_root + at[0], src[0], bytes[0], hipMemcpyDeviceToDevice, g->streams[0]));
  }
  if(!g->comms.empty())
  {
    RcclApi& api = rccl();
    RCCL_TRY(api.GroupStart());
    ncclResult_t res = ncclSuccess;
    for(int r = 1; r < G && res == ncclSuccess; r++)
    {
      if(bytes[size_t(r)] == 0) { continue; }
      res = api.Send(src[size_t(r)], bytes[size_t(r)], ncclUint8, 0, g->comms[size_t(r)], g->streams[size_t(r)]);
      if(res == ncclSuccess) { res = api.Recv(dst_root + at[size_t(r)], bytes[size_t(r)], ncclUint8, r, g->comms[0], g->streams[0]); }
    }
    ncclResult_t end = api.GroupEnd();
    if(res != ncclSuccess) { return fail(GCSA2_ERR_HIP, std::string("ncclSend / ncclRecv: ") + api.GetErrorString(res)); }
    if(end != ncclSuccess) { return fail(GCSA2_ERR_HIP, std::string("ncclGroupEnd: ") + api.GetErrorString(end)); }
  }
  else
  {
    for(int r = 1; r < G; r++)
    {
      if(bytes[size_t(r)] == 0) { continue; }
      DeviceGuard guard(g->replicas[size_t(r)]->device);
      HIP_TRY(hipMemcpyPeerAsync(dst_root + at[size_t(r)], g->replicas[0]->device, src[size_t(r)], g->replicas[size_t(r)]->device,
                                 bytes[size_t(r)], g->streams[size_t(r)]));
    }
  }
  return GCSA2_OK;
}

int group_sync(gcsa2_group* g)
{
  for(size_t r = 0; r < g->replicas.size(); r++)
  {
    DeviceGuard guard(g->replicas[r]->device);
    HIP_TRY(hipStreamSynchronize(g->streams[r]));
  }
  return GCSA2_OK;
}

// One communicator per device (ncclCommInitAll), made at the first device-resident group call.  RCCL needs distinct
// devices; a group that lists a device twice (or a host without RCCL) gathers with peer copies instead.
int group_comm_init(gcsa2_group* g)
{
  if(g->comm_tried) { return GCSA2_OK; }
  g->comm_tried = true;
  const int G = int(g->replicas.size());
  std::vector<int> devs;
  bool distinct = true;
  for(int r = 0; r < G; r++)
  {
    for(int d : devs) { distinct = distinct && d != g->replicas[size_t(r)]->device; }
    devs.push_back(g->replicas[size_t(r)]->device);
  }
  if(G > 1 && distinct && rccl().ok)
  {
    g->comms.assign(size_t(G), nullptr);
    ncclResult_t r = rccl().CommInitAll(g->comms.data(), G, devs.data());
    if(r != ncclSuccess) { g->comms.clear(); return fail(GCSA2_ERR_HIP, std::string("ncclCommInitAll: ") + rccl().GetErrorString(r)); }
  }
  return GCSA2_OK;
}

__global__ __launch_bounds__(TPB) void k_rebase_offsets(u64* __restrict__ offsets, u64 count, u64 base)
{
  const u64 i = u64(blockIdx.x) * TPB + threadIdx.x;
  if(i < count) { offsets[i] += base; }
}

}  // namespace

extern "C" {

// Matching statistics of a batch sharded over the group (BASELINE configs[4] on N GPUs): replica r runs its shard
// (d_patterns[r], d_offsets[r] rebased to 0, counts[r] patterns of pattern_bytes[r] bytes in all) and the three result arrays
// are gathered on replica 0's device in query order.  Complete on return.
int gcsa2_group_match_stats_device(gcsa2_group* g, const uint8_t* const* d_patterns, const uint64_t* const* d_offsets, const uint64_t* counts,
                                   const uint64_t* pattern_bytes, uint16_t* d_ms_root, uint64_t* d_ranges_root, uint64_t* d_fallbacks_root)
{
  if(g == nullptr || g->replicas.empty()) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "null or empty group"); }
  if(d_patterns == nullptr || d_offsets == nullptr || counts == nullptr || pattern_bytes == nullptr || d_ms_root == nullptr || d_ranges_root == nullptr)
  {
    return fail(GCSA2_ERR_INVALID_ARGUMENT, "null buffer");
  }
  try {
  std::lock_guard<std::mutex> hold(g->lock);
  const size_t G = g->replicas.size();
  int rc = group_comm_init(g);
  if(rc != GCSA2_OK) { return rc; }
  std::vector<const void*> ms(G, nullptr), rng(G, nullptr), fb(G, nullptr);
  std::vector<u64> ms_bytes(G, 0), rng_bytes(G, 0), fb_bytes(G, 0), ms_at(G, 0), rng_at(G, 0), fb_at(G, 0);
  std::vector<void*> scratch;
  struct Release { gcsa2_group* g; std::vector<void*>& p; std::vector<int> dev; ~Release() { for(size_t i = 0; i < p.size(); i++) { DeviceGuard guard(dev[i]); (void)hipFree(p[i]); } } } release{g, scratch, {}};
  u64 q_before = 0, b_before = 0;
  for(size_t r = 0; r < G; r++)
  {
    ms_at[r] = 2 * b_before; rng_at[r] = 16 * q_before; fb_at[r] = 8 * q_before;
    ms_bytes[r] = 2 * pattern_bytes[r]; rng_bytes[r] = 16 * counts[r]; fb_bytes[r] = 8 * counts[r];
    q_before += counts[r]; b_before += pattern_bytes[r];
    if(counts[r] == 0) { continue; }
    const gcsa2_index* ix = g->replicas[r];
    DeviceGuard guard(ix->device);
    // the kernel wants an 8-byte aligned statistics array with 4 spare entries: a scratch copy per replica (replica 0's
    // slice of the result starts at 0 and is used in place)
    char* buf = nullptr;
    const u64 ms_room = (2 * pattern_bytes[r] + 8 * sizeof(uint16_t) + 15) / 16 * 16;
    uint16_t* my_ms = d_ms_root; u64* my_rng = d_ranges_root; u64* my_fb = d_fallbacks_root;
    if(r > 0)
    {
      HIP_TRY(hipMalloc(reinterpret_cast<void**>(&buf), ms_room + 24 * counts[r]));
      scratch.push_back(buf); release.dev.push_back(ix->device);
      my_ms = reinterpret_cast<uint16_t*>(buf); my_rng = reinterpret_cast<u64*>(buf + ms_room); my_fb = my_rng + 2 * counts[r];
    }
    else if(d_fallbacks_root == nullptr) { fb_bytes[0] = 0; }
    rc = match_stats_launch(ix, 0, d_patterns[r], d_offsets[r], counts[r], pattern_bytes[r], my_ms, my_rng, my_fb, g->streams[r]);
    if(rc != GCSA2_OK) { return rc; }
    ms[r] = my_ms; rng[r] = my_rng; fb[r] = my_fb;
  }
  if(d_fallbacks_root == nullptr) { for(size_t r = 0; r < G; r++) { fb_bytes[r] = 0; } }
  rc = group_gather(g, ms, ms_bytes, reinterpret_cast<char*>(d_ms_root), ms_at);
  if(rc == GCSA2_OK) { rc = group_gather(g, rng, rng_bytes, reinterpret_cast<char*>(d_ranges_root), rng_at); }
  if(rc == GCSA2_OK && d_fallbacks_root != nullptr) { rc = group_gather(g, fb, fb_bytes, reinterpret_cast<char*>(d_fallbacks_root), fb_at); }
  int sc = group_sync(g);
  return rc != GCSA2_OK ? rc : sc;
  } catch(const std::exception& e) { return fail(GCSA2_ERR_OUT_OF_MEMORY, std::string("gcsa2_group_match_stats_device: ") + e.what()); }
}

// locate() of a batch of ranges sharded over the group: replica r locates d_ranges[r] (counts[r] ranges); the root receives the
// CSR of the whole batch in query order -- per-rank totals first, then the offsets (rebased) and the values (SURVEY.md 8(e):
// "all-gather of per-rank counts, then gatherv of the CSR values").  *job owns the values on replica 0's device.
int gcsa2_group_locate_device(gcsa2_group* g, const uint64_t* const* d_ranges, const uint64_t* counts, int sort, uint64_t* d_offsets_root,
                              gcsa2_locate_job** job_out, const uint64_t** d_values_root, uint64_t* total_values)
{
  if(g == nullptr || g->replicas.empty()) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "null or empty group"); }
  if(d_ranges == nullptr || counts == nullptr || d_offsets_root == nullptr || job_out == nullptr) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "null buffer"); }
  *job_out = nullptr;
  try {
  std::lock_guard<std::mutex> hold(g->lock);
  const size_t G = g->replicas.size();
  int rc = group_comm_init(g);
  if(rc != GCSA2_OK) { return rc; }
  // every replica locates its shard from its own host thread (the locate pipeline synchronises its stream)
  std::vector<gcsa2_locate_job*> jobs(G, nullptr);
  std::vector<int> status(G, GCSA2_OK);
  std::vector<std::string> messages(G);
  struct Discard { std::vector<gcsa2_locate_job*>& j; ~Discard() { for(gcsa2_locate_job* x : j) { gcsa2_locate_discard(x); } } } discard{jobs};
  {
    Workers workers;
    for(size_t r = 0; r < G; r++)
    {
      workers.emplace_back([&, r]()
      {
        status[r] = gcsa2_locate_device(g->replicas[r], d_ranges[r], counts[r], sort, &jobs[r], nullptr, nullptr, nullptr, g->streams[r]);
        if(status[r] != GCSA2_OK) { messages[r] = g_error; }
      });
    }
    workers.join();
  }
  for(size_t r = 0; r < G; r++) { if(status[r] != GCSA2_OK) { return fail(status[r], "shard " + std::to_string(r) + ": " + messages[r]); } }
  u64 total = 0, q_before = 0;
  std::vector<const void*> off(G, nullptr), val(G, nullptr);
  std::vector<u64> off_bytes(G, 0), val_bytes(G, 0), off_at(G, 0), val_at(G, 0), base(G, 0);
  for(size_t r = 0; r < G; r++)
  {
    base[r] = total; off_at[r] = 8 * q_before; val_at[r] = 8 * total;
    off[r] = jobs[r]->d_offsets; off_bytes[r] = 8 * counts[r];
    val[r] = jobs[r]->d_values; val_bytes[r] = 8 * jobs[r]->total;
    total += jobs[r]->total; q_before += counts[r];
  }
  const gcsa2_index* root = g->replicas[0];
  gcsa2_locate_job* result = new(std::nothrow) gcsa2_locate_job();
  if(result == nullptr) { return fail(GCSA2_ERR_OUT_OF_MEMORY, "host allocation failed"); }
  result->device = root->device; result->nq = q_before; result->total = total;
  {
    DeviceGuard guard(root->device);
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&result->d_values), (total > 0 ? total : 1) * sizeof(u64));
    if(e != hipSuccess) { delete result; return fail(GCSA2_ERR_OUT_OF_MEMORY, std::string("hipMalloc(values): ") + hipGetErrorString(e)); }
  }
  rc = group_gather(g, off, off_bytes, reinterpret_cast<char*>(d_offsets_root), off_at);
  if(rc == GCSA2_OK) { rc = group_gather(g, val, val_bytes, reinterpret_cast<char*>(result->d_values), val_at); }
  if(rc == GCSA2_OK) { rc = group_sync(g); }
  if(rc == GCSA2_OK)
  {
    DeviceGuard guard(root->device);
    u64 at = 0;
    for(size_t r = 0; r < G; r++)
    {
      if(counts[r] > 0 && base[r] > 0)
      {
        hipLaunchKernelGGL(k_rebase_offsets, dim3(grid_for(counts[r])), dim3(TPB), 0, g->streams[0], d_offsets_root + at, counts[r], base[r]);
      }
      at += counts[r];
    }
    hipError_t e = hipMemcpyAsync(d_offsets_root + q_before, &total, sizeof(u64), hipMemcpyHostToDevice, g->streams[0]);
    if(e == hipSuccess) { e = hipStreamSynchronize(g->streams[0]); }
    if(e != hipSuccess) { rc = fail(GCSA2_ERR_HIP, std::string("offsets of the gathered batch: ") + hipGetErrorString(e)); }
  }
  if(rc != GCSA2_OK) { gcsa2_locate_discard(result); return rc; }
  *job_out = result;
  if(d_values_root) { *d_values_root = result->d_values; }
  if(total_values) { *total_values = total; }
  return GCSA2_OK;
  } catch(const std::exception& e) { return fail(GCSA2_ERR_OUT_OF_MEMORY, std::string("gcsa2_group_locate_device: ") + e.what()); }
}

// ---- the same two queries with one process per GPU (gcsa2_comm): every rank works on its contiguous shard, the results are
// gathered on the root through gcsa2_comm_gather's grouped send / recv ----------------------------------------------------------

// counts[r] / pattern_bytes[r]: patterns and pattern bytes of rank r's shard (the same arrays on every rank).  Enqueue-only on
// `stream` (the scratch of the shard's results is stream-ordered); the root's arrays are complete when the stream is.
int gcsa2_comm_match_stats(gcsa2_comm* c, const gcsa2_index* ix, const uint8_t* d_patterns, const uint64_t* d_offsets, const uint64_t* counts,
                           const uint64_t* pattern_bytes, int root, uint16_t* d_ms_root, uint64_t* d_ranges_root, uint64_t* d_fallbacks_root,
                           void* stream)
{
  CHECK_INDEX(ix);
  if(c == nullptr || counts == nullptr || pattern_bytes == nullptr || root < 0 || root >= c->world) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "bad arguments"); }
  if(c->rank == root && (d_ms_root == nullptr || d_ranges_root == nullptr || d_fallbacks_root == nullptr)) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "the root needs all three result buffers"); }
  if(ix->device != c->device) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "index and communicator are on different devices"); }
  DeviceGuard guard(c->device);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const u64 nq = counts[c->rank], bytes = pattern_bytes[c->rank];
  const u64 ms_room = (2 * bytes + 8 * sizeof(uint16_t) + 15) / 16 * 16;
  char* buf = nullptr;
  HIP_TRY(pool_alloc(ix, reinterpret_cast<void**>(&buf), ms_room + 24 * nq + 16, st));
  uint16_t* my_ms = reinterpret_cast<uint16_t*>(buf); u64* my_rng = reinterpret_cast<u64*>(buf + ms_room); u64* my_fb = my_rng + 2 * nq;
  // A rank whose kernel launch fails still takes part in the three gathers (its buffers exist; the root receives whatever they
  // hold) and reports the error afterwards: leaving now would keep the other ranks waiting in ncclSend / ncclRecv for ever.
  // (The allocation above is the one failure that cannot be carried through: without a buffer there is nothing to send.)
  const int launch_rc = (nq > 0 ? match_stats_launch(ix, 0, d_patterns, d_offsets, nq, bytes, my_ms, my_rng, my_fb, st) : GCSA2_OK);
  const std::string launch_error = (launch_rc != GCSA2_OK ? g_error : std::string());
  std::vector<u64> sizes(size_t(c->world));
  int rc = GCSA2_OK;
  for(int part = 0; part < 3; part++)
  {
    for(int r = 0; r < c->world; r++) { sizes[size_t(r)] = (part == 0 ? 2 * pattern_bytes[r] : (part == 1 ? 16 * counts[r] : 8 * counts[r])); }
    const void* src = (part == 0 ? static_cast<const void*>(my_ms) : (part == 1 ? static_cast<const void*>(my_rng) : static_cast<const void*>(my_fb)));
    void* dst = (part == 0 ? static_cast<void*>(d_ms_root) : (part == 1 ? static_cast<void*>(d_ranges_root) : static_cast<void*>(d_fallbacks_root)));
    const int g_rc = gather_bytes(c, src, sizes.data(), dst, root, st);
    if(rc == GCSA2_OK) { rc = g_rc; }
  }
  (void)hipFreeAsync(buf, st);
  if(launch_rc != GCSA2_OK) { return fail(launch_rc, launch_error); }
  return rc;
}

// locate() of this rank's shard (counts[rank] ranges); on the root: d_offsets_root (sum of counts + 1 entries) and a job that owns
// the values of the whole batch in query order.  Per-rank totals travel first (8 bytes each), then the offsets and the values
// (SURVEY.md 8(e)).  Complete on return (the sizes of the second gather are read on the host).
int gcsa2_comm_locate(gcsa2_comm* c, const gcsa2_index* ix, const uint64_t* d_ranges, const uint64_t* counts, int sort, int root,
                      uint64_t* d_offsets_root, gcsa2_locate_job** job_root, const uint64_t** d_values_root, uint64_t* total_values, void* stream)
{
  CHECK_INDEX(ix);
  if(c == nullptr || counts == nullptr || root < 0 || root >= c->world) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "bad arguments"); }
  if(c->rank == root && (d_offsets_root == nullptr || job_root == nullptr)) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "the root needs result buffers"); }
  if(ix->device != c->device) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "index and communicator are on different devices"); }
  if(job_root != nullptr) { *job_root = nullptr; }
  try {
  DeviceGuard guard(c->device);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t W = size_t(c->world);
  const bool is_root = (c->rank == root);
  // A rank whose own part fails must not leave the others waiting in ncclSend / ncclRecv: it still joins the first gather, with
  // FAILED as its total, and the later gathers with nothing to send; the root, which sees the mark, expects nothing from that
  // rank, lets the healthy ranks finish and reports the failure.  Every rank returns its own status.
  constexpr u64 FAILED = ~u64(0);
  gcsa2_locate_job* mine = nullptr;
  const int local_rc = gcsa2_locate_device(ix, d_ranges, counts[c->rank], sort, &mine, nullptr, nullptr, nullptr, st);
  const std::string local_error = (local_rc != GCSA2_OK ? g_error : std::string());
  struct Discard { gcsa2_locate_job*& j; ~Discard() { gcsa2_locate_discard(j); } } discard{mine};
  // 1. per-rank totals: the last entry of every rank's offsets (a slot of the handle: no allocation that could fail in between)
  std::vector<u64> eight(W, sizeof(u64)), totals(W, 0);
  const unsigned slot = ix->next_slot.fetch_add(1) % RESULT_SLOTS;
  u64* d_mark = reinterpret_cast<u64*>(ix->d_slots + u64(TOTAL_WORDS) * slot);
  u64* d_totals = nullptr;
  struct FreeAsync { u64*& p; hipStream_t st; ~FreeAsync() { if(p != nullptr) { (void)hipFreeAsync(p, st); } } } free_totals{d_totals, st};
  int rc = GCSA2_OK;
  if(is_root)
  {
    hipError_t e = pool_alloc(ix, reinterpret_cast<void**>(&d_totals), W * sizeof(u64), st);
    if(e != hipSuccess) { rc = fail(GCSA2_ERR_OUT_OF_MEMORY, std::string("pool_alloc(totals): ") + hipGetErrorString(e)); }
  }
  if(rc != GCSA2_OK) { return rc; }               // (the root without 8 bytes per rank: nothing can be received at all)
  const u64* my_total = (mine != nullptr ? mine->d_offsets + counts[c->rank] : d_mark);
  if(mine == nullptr)
  {
    const u64 mark = FAILED;
    hipError_t e = hipMemcpyAsync(d_mark, &mark, sizeof(u64), hipMemcpyHostToDevice, st);
    if(e == hipSuccess) { e = hipStreamSynchronize(st); }
    if(e != hipSuccess) { return fail(local_rc, local_error); }    // the device itself is gone
  }
  rc = gather_bytes(c, my_total, eight.data(), d_totals, root, st);
  if(rc == GCSA2_OK && is_root)
  {
    hipError_t e = hipMemcpyAsync(totals.data(), d_totals, W * sizeof(u64), hipMemcpyDeviceToHost, st);
    if(e != hipSuccess) { rc = fail(GCSA2_ERR_HIP, std::string("totals of the shards: ") + hipGetErrorString(e)); }
  }
  if(rc == GCSA2_OK)
  {
    hipError_t e = hipStreamSynchronize(st);
    if(e != hipSuccess) { rc = fail(GCSA2_ERR_HIP, std::string("totals of the shards: ") + hipGetErrorString(e)); }
  }
  if(rc != GCSA2_OK) { return rc; }
  if(!is_root) { totals[size_t(c->rank)] = (mine != nullptr ? mine->total : FAILED); }
  // 2. offsets (counts[r] entries each) and values (totals[r] each); a peer only needs its own sizes
  std::vector<u64> off_bytes(W), val_bytes(W);
  u64 total = 0, queries = 0;
  int failed_rank = -1;
  for(size_t r = 0; r < W; r++)
  {
    const bool bad = (totals[r] == FAILED);
    if(bad) { totals[r] = 0; if(failed_rank < 0) { failed_rank = int(r); } }
    off_bytes[r] = (bad ? 0 : 8 * counts[r]); val_bytes[r] = 8 * totals[r]; total += totals[r]; queries += counts[r];
  }
  gcsa2_locate_job* result = nullptr;
  // a rank whose pass failed sends zero bytes -- from a device address that always exists (a custom transport may look at it)
  const u64* my_offsets = (mine != nullptr ? mine->d_offsets : d_mark);
  const u64* my_values = (mine != nullptr ? mine->d_values : d_mark);
  if(is_root)
  {
    result = new(std::nothrow) gcsa2_locate_job();
    hipError_t e = hipSuccess;
    if(result != nullptr)
    {
      result->device = c->device; result->nq = queries; result->total = total;
      e = hipMalloc(reinterpret_cast<void**>(&result->d_values), (total > 0 ? total : 1) * sizeof(u64));
    }
    if(result == nullptr || e != hipSuccess)
    {
      // the peers are about to send: without a buffer the communicator cannot complete this call on any rank
      delete result;
      return fail(GCSA2_ERR_OUT_OF_MEMORY, "no memory for the values of the batch on the root (the other ranks of this call will not return: destroy the communicator)");
    }
  }
  rc = gather_bytes(c, my_offsets, off_bytes.data(), d_offsets_root, root, st);
  {
    const int g_rc = gather_bytes(c, my_values, val_bytes.data(), is_root ? result->d_values : nullptr, root, st);
    if(rc == GCSA2_OK) { rc = g_rc; }
  }
  if(rc == GCSA2_OK && local_rc != GCSA2_OK) { rc = fail(local_rc, local_error); }
  if(rc == GCSA2_OK && failed_rank >= 0) { rc = fail(GCSA2_ERR_HIP, "locate failed on rank " + std::to_string(failed_rank) + " of the communicator; the batch is incomplete"); }
  if(rc == GCSA2_OK && is_root)
  {
    u64 at = 0, base = 0;
    for(size_t r = 0; r < W; r++)
    {
      if(counts[r] > 0 && base > 0) { hipLaunchKernelGGL(k_rebase_offsets, dim3(grid_for(counts[r])), dim3(TPB), 0, st, d_offsets_root + at, counts[r], base); }
      at += counts[r]; base += totals[r];
    }
    hipError_t e = hipMemcpyAsync(d_offsets_root + queries, &total, sizeof(u64), hipMemcpyHostToDevice, st);
    if(e != hipSuccess) { rc = fail(GCSA2_ERR_HIP, std::string("offsets of the gathered batch: ") + hipGetErrorString(e)); }
  }
  hipError_t se = hipStreamSynchronize(st);
  if(rc == GCSA2_OK && se != hipSuccess) { rc = fail(GCSA2_ERR_HIP, std::string("gather: ") + hipGetErrorString(se)); }
  if(rc != GCSA2_OK) { gcsa2_locate_discard(result); return rc; }
  if(is_root)
  {
    *job_root = result;
    if(d_values_root) { *d_values_root = result->d_values; }
    if(total_values) { *total_values = total; }
  }
  return GCSA2_OK;
  } catch(const std::exception& e) { return fail(GCSA2_ERR_OUT_OF_MEMORY, std::string("gcsa2_comm_locate: ") + e.what()); }
}

int gcsa2_group_size(const gcsa2_group* g) { return g == nullptr ? 0 : int(g->replicas.size()); }

const gcsa2_index* gcsa2_group_index(const gcsa2_group* g, int i)
{
  return (g == nullptr || i < 0 || i >= int(g->replicas.size())) ? nullptr : g->replicas[size_t(i)];
}

int gcsa2_group_find_batch(const gcsa2_group* g, const uint8_t* patterns, const uint64_t* offsets, uint64_t nq, uint64_t* ranges)
{
  if(g == nullptr || g->replicas.empty()) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "null or empty group"); }
  if(nq == 0) { return GCSA2_OK; }
  if(offsets == nullptr || ranges == nullptr) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "null buffer"); }
  if(!offsets_ok(offsets, nq)) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "pattern offsets are not non-decreasing"); }
  try {   // no C++ exception may cross the C boundary
  const u64 G = g->replicas.size(), base = nq / G, rem = nq % G;
  std::vector<int> status(G, GCSA2_OK);
  std::vector<std::string> messages(G);
  Workers workers;
  u64 begin = 0;
  for(u64 r = 0; r < G; r++)
  {
    u64 count = base + (r < rem ? 1 : 0), b = begin;
    begin += count;
    if(count == 0) { continue; }
    workers.emplace_back([&, r, b, count]()
    {
      try
      {
        std::vector<u64> local(count + 1);
        for(u64 i = 0; i <= count; i++) { local[i] = offsets[b + i] - offsets[b]; }
        status[r] = gcsa2_find_batch(g->replicas[r], patterns + offsets[b], local.data(), count, ranges + 2 * b);
        if(status[r] != GCSA2_OK) { messages[r] = g_error; }     // g_error is thread-local
      }
      catch(const std::exception& e) { status[r] = GCSA2_ERR_OUT_OF_MEMORY; messages[r] = e.what(); }
    });
  }
  workers.join();
  for(u64 r = 0; r < G; r++) { if(status[r] != GCSA2_OK) { return fail(status[r], "shard " + std::to_string(r) + ": " + messages[r]); } }
  return GCSA2_OK;
  } catch(const std::exception& e) { return fail(GCSA2_ERR_OUT_OF_MEMORY, std::string("gcsa2_group_find_batch: ") + e.what()); }
}

}  // extern "C"

// countKMers(index, k, parameters) (src/algorithms.cpp:387-421): level-synchronous frontier
// expansion on the device instead of the reference's seed DFS + OpenMP; the count is the number of
// non-empty states at depth k either way.
namespace {
int kmer_level(const gcsa2_index* ix, const u64* d_in, u64 n_in, u32 limit, u64* d_out, unsigned long long* d_counter, u64& produced)
{
  HIP_TRY(hipMemset(d_counter, 0, sizeof(unsigned long long)));
  hipLaunchKernelGGL(k_kmer_expand, dim3(grid_for(n_in)), dim3(TPB), 0, nullptr, ix->img, d_in, n_in, limit, d_out, d_counter);
  LAUNCH_CHECK("k_kmer_expand");
  unsigned long long c = 0;
  HIP_TRY(hipMemcpy(&c, d_counter, sizeof(c), hipMemcpyDeviceToHost));
  produced = c;
  return GCSA2_OK;
}

// One frontier buffer per depth of the search, reused by every piece of that depth (a depth-first walk over pieces: when a
// piece of depth d is taken up, everything below the piece before it is finished) and grown to the largest piece seen.
struct KmerBufs
{
  std::vector<u64*> p; std::vector<size_t> cap;
  ~KmerBufs() { for(u64* q : p) { if(q != nullptr) { (void)hipFree(q); } } }
  hipError_t get(size_t depth, size_t bytes, u64*& out)
  {
    if(p.size() <= depth) { p.resize(depth + 1, nullptr); cap.resize(depth + 1, 0); }
    if(cap[depth] < bytes)
    {
      if(p[depth] != nullptr) { (void)hipFree(p[depth]); p[depth] = nullptr; cap[depth] = 0; }
      hipError_t e = hipMalloc(reinterpret_cast<void**>(&p[depth]), bytes);
      if(e != hipSuccess) { p[depth] = nullptr; return e; }
      cap[depth] = bytes;
    }
    out = p[depth];
    return hipSuccess;
  }
};

// The level-synchronous search tree of countKMers below a frontier of `n` states (pairs sp, ep) at depth `depth`.  A frontier
// whose children might not fit one 2 GB buffer (n x limit > 2^27) is cut in halves that are searched one after the other IN
// PLACE; the children of a piece go into the buffer of their depth.  (Until late round 4 the halves were copied, larger levels
// were expanded twice -- a counting pass, then a filling pass -- and every piece allocated and freed its buffer: k = 17 on the
// 5.73 G-node index, a last frontier of 3.4 G states, took 4.5 s, most of it in hipMalloc / hipFree of gigabyte blocks.)
int kmer_run(const gcsa2_index* ix, const u64* d_frontier, u64 n, u64 depth, u64 k, u32 limit, unsigned long long* d_counter, u64& total, KmerBufs& bufs)
{
  const u64 MAX_CHILDREN = ix->tune.kmer_piece;   // 2^27 states = 2 GB (GCSA2_KMER_PIECE: tests cut finer)
  while(depth < k && n > 0)
  {
    u64 produced = 0;
    int rc = GCSA2_OK;
    if(depth + 1 == k)      // last level: the children only have to be counted (no buffer: any size)
    {
      rc = kmer_level(ix, d_frontier, n, limit, nullptr, d_counter, produced);
      if(rc != GCSA2_OK) { return rc; }
      total += produced;
      return GCSA2_OK;
    }
    if(n * limit > MAX_CHILDREN && n > 1)
    {
      const u64 half = n / 2;
      rc = kmer_run(ix, d_frontier, half, depth, k, limit, d_counter, total, bufs);
      if(rc == GCSA2_OK) { rc = kmer_run(ix, d_frontier + 2 * half, n - half, depth, k, limit, d_counter, total, bufs); }
      return rc;
    }
    u64* next = nullptr;           // one pass into the buffer of the next depth, sized for the worst case (`limit` children per state)
    HIP_TRY(bufs.get(size_t(depth + 1), size_t(n * limit * 2 * sizeof(u64)), next));
    rc = kmer_level(ix, d_frontier, n, limit, next, d_counter, produced);
    if(rc != GCSA2_OK) { return rc; }
    if(produced == 0) { return GCSA2_OK; }
    d_frontier = next; n = produced; depth++;
  }
  if(depth == k) { total += n; }
  return GCSA2_OK;
}
}  // namespace

extern "C" int gcsa2_count_kmers(const gcsa2_index* ix, uint64_t k, int include_ns, int force, uint64_t* result)
{
  CHECK_INDEX(ix);
  if(result == nullptr) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "null result"); }
  *result = 0;
  if(k == 0) { *result = 1; return GCSA2_OK; }                       // algorithms.cpp:390
  if(k > ix->order && !force) { return GCSA2_OK; }                   // algorithms.cpp:391-395 (returns 0)
  if(ix->img.n == 0) { return GCSA2_OK; }
  if(ix->img.sigma < 3) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "alphabet too small for countKMers"); }
  u32 limit = u32(include_ns ? ix->img.sigma - 2 : ix->img.fast_chars);   // comps 1..limit (algorithms.cpp:369, 379)
  DeviceGuard guard(ix->device);
  DBuf<unsigned long long> counter;
  HIP_TRY(counter.alloc(1));
  KmerBufs bufs;
  u64* frontier = nullptr;
  HIP_TRY(bufs.get(0, 2 * sizeof(u64), frontier));
  u64 root[2] = {0, ix->img.n - 1};
  HIP_TRY(hipMemcpy(frontier, root, sizeof(root), hipMemcpyHostToDevice));
  u64 total = 0;
  int rc = kmer_run(ix, frontier, 1, 0, k, limit, counter.p, total, bufs);
  if(rc != GCSA2_OK) { return rc; }
  *result = total;
  return GCSA2_OK;
}

// Matching statistics (LF + parent fused); see k_match_stats2.  variant 2 = one lane per pattern, 5 = persistent lanes that draw
// patterns from a counter, 0 = the library chooses by batch size.  total_bytes = offsets[nq] when the caller knows it (GCSA2_UNKNOWN: read back from the
// device, which waits for the stream once).
namespace {
int match_stats_launch(const gcsa2_index* ix, int variant, const uint8_t* d_patterns, const uint64_t* d_offsets, uint64_t nq, u64 total_bytes,
                       uint16_t* d_ms, uint64_t* d_ranges, uint64_t* d_fallbacks, hipStream_t st, const BreakSink* sink)
{
  CHECK_INDEX(ix);
  DeviceGuard guard(ix->device);
  if(!ix->img.has_lcp) { return fail(GCSA2_ERR_MISSING_COMPONENT, "index was created without an LCP array"); }
  if(variant != 0 && variant != 2 && variant != 5)
  {
    return fail(GCSA2_ERR_INVALID_ARGUMENT, "unknown matching statistics variant (0: the library chooses; 2 / 5: a lane per pattern / persistent lanes)");
  }
  if(nq == 0 || ix->img.n == 0) { return GCSA2_OK; }
  if(total_bytes == GCSA2_UNKNOWN)
  {
    HIP_TRY(hipMemcpyAsync(&total_bytes, d_offsets + nq, sizeof(u64), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
  }
  // pre-pass: the patterns as 16-byte records, last character first (k_pack_records); stream-ordered scratch
  ulonglong2* recs = nullptr;
  HIP_TRY(pool_alloc(ix, reinterpret_cast<void**>(&recs), ((total_bytes >> 5) + nq + 6) * sizeof(ulonglong2), st));
  hipLaunchKernelGGL(k_pack_records, dim3(grid_for(nq)), dim3(TPB), 0, st, ix->img, d_patterns, d_offsets, nq, recs);
  const u32 cool = ix->tune.cool_down;
  unsigned short* out = reinterpret_cast<unsigned short*>(d_ms);
  const u64 lanes_grid = (nq + TPB2 - 1) / TPB2;
  const bool pair = ix->img.flp != nullptr;
  unsigned long long* queue = nullptr;
  const u64 resident = ix->tune.ms_grid != 0 ? ix->tune.ms_grid : u64(ix->compute_units) * 8;     // workgroups the device holds at 4 waves per SIMD
  // variant 0 chooses: a batch of more than two generations of workgroups goes to the persistent lanes -- patterns that need
  // parent() take three times the rounds of those that do not, and a wave whose lanes own fixed patterns waits for its slowest
  // (1 M x 256 bp, half with mismatches: 114 -> 128 M patterns/s; without mismatches 242 -> 236; profiles/r03_match_stats.md)
  if(variant == 0) { variant = (lanes_grid > 2 * resident ? 5 : 2); }
  if(variant == 5)
  {
    hipError_t qe = pool_alloc(ix, reinterpret_cast<void**>(&queue), sizeof(unsigned long long), st);
    if(qe == hipSuccess) { qe = hipMemsetAsync(queue, 0, sizeof(unsigned long long), st); }
    if(qe != hipSuccess) { (void)hipFreeAsync(recs, st); return fail(GCSA2_ERR_HIP, std::string("matching statistics queue: ") + hipGetErrorString(qe)); }
    const unsigned grid = unsigned(lanes_grid < resident ? lanes_grid : resident);
    unsigned long long* none = nullptr;
    if(sink != nullptr && pair)
    {
      hipLaunchKernelGGL((k_match_stats2<true, true, false, true>), dim3(grid), dim3(TPB2), 0, st,
                         ix->img, d_patterns, d_offsets, nq, out, d_ranges, d_fallbacks, cool, queue, ix->tune.ms_refill_at, recs, none, *sink);
    }
    else if(sink != nullptr)
    {
      hipLaunchKernelGGL((k_match_stats2<false, true, false, true>), dim3(grid), dim3(TPB2), 0, st,
                         ix->img, d_patterns, d_offsets, nq, out, d_ranges, d_fallbacks, cool, queue, ix->tune.ms_refill_at, recs, none, *sink);
    }
    else if(pair)
    {
      hipLaunchKernelGGL((k_match_stats2<true, true>), dim3(grid), dim3(TPB2), 0, st,
                         ix->img, d_patterns, d_offsets, nq, out, d_ranges, d_fallbacks, cool, queue, ix->tune.ms_refill_at, recs);
    }
    else
    {
      hipLaunchKernelGGL((k_match_stats2<false, true>), dim3(grid), dim3(TPB2), 0, st,
                         ix->img, d_patterns, d_offsets, nq, out, d_ranges, d_fallbacks, cool, queue, ix->tune.ms_refill_at, recs);
    }
  }
  else if(sink != nullptr)
  {
    unsigned long long* none = nullptr;
    if(pair)
    {
      hipLaunchKernelGGL((k_match_stats2<true, false, false, true>), dim3(unsigned(lanes_grid)), dim3(TPB2), 0, st,
                         ix->img, d_patterns, d_offsets, nq, out, d_ranges, d_fallbacks, cool, none, 64u, recs, none, *sink);
    }
    else
    {
      hipLaunchKernelGGL((k_match_stats2<false, false, false, true>), dim3(unsigned(lanes_grid)), dim3(TPB2), 0, st,
                         ix->img, d_patterns, d_offsets, nq, out, d_ranges, d_fallbacks, cool, none, 64u, recs, none, *sink);
    }
  }
  else if(pair)
  {
    hipLaunchKernelGGL((k_match_stats2<true, false>), dim3(unsigned(lanes_grid)), dim3(TPB2), 0, st,
                       ix->img, d_patterns, d_offsets, nq, out, d_ranges, d_fallbacks, cool, (unsigned long long*)nullptr, 64u, recs);
  }
  else
  {
    hipLaunchKernelGGL((k_match_stats2<false, false>), dim3(unsigned(lanes_grid)), dim3(TPB2), 0, st,
                       ix->img, d_patterns, d_offsets, nq, out, d_ranges, d_fallbacks, cool, (unsigned long long*)nullptr, 64u, recs);
  }
  hipError_t le = hipGetLastError();
  if(queue != nullptr) { (void)hipFreeAsync(queue, st); }
  (void)hipFreeAsync(recs, st);                                        // stream-ordered: released after the kernel
  if(le != hipSuccess) { return fail(GCSA2_ERR_HIP, std::string("k_match_stats2: ") + hipGetErrorString(le)); }
  return GCSA2_OK;
}
}  // namespace

extern "C" int gcsa2_match_stats_device_variant(const gcsa2_index* ix, int variant, const uint8_t* d_patterns, const uint64_t* d_offsets,
                                                uint64_t nq, uint16_t* d_ms, uint64_t* d_ranges, uint64_t* d_fallbacks, void* stream)
{
  return match_stats_launch(ix, variant, d_patterns, d_offsets, nq, GCSA2_UNKNOWN, d_ms, d_ranges, d_fallbacks, static_cast<hipStream_t>(stream));
}

extern "C" int gcsa2_match_stats_device(const gcsa2_index* ix, const uint8_t* d_patterns, const uint64_t* d_offsets, uint64_t nq,
                                        uint16_t* d_ms, uint64_t* d_ranges, uint64_t* d_fallbacks, void* stream)
{
  return match_stats_launch(ix, 0, d_patterns, d_offsets, nq, GCSA2_UNKNOWN, d_ms, d_ranges, d_fallbacks, static_cast<hipStream_t>(stream));
}

// Matching statistics as BREAK POINTS (k_match_stats2<.., BREAKS>): the CSR of the left-maximal matches of every pattern.
// Complete on return: the number of records is read back, and a buffer that is too small is refused with the number needed.
int gcsa2_match_breaks_device(const gcsa2_index* ix, const uint8_t* d_patterns, const uint64_t* d_offsets, uint64_t nq, uint64_t total_bytes,
                              int variant, uint64_t min_length, uint64_t* d_break_offsets, gcsa2_break* d_breaks, uint64_t capacity, uint64_t* total_breaks,
                              uint64_t* d_ranges, uint64_t* d_fallbacks, void* stream)
{
  CHECK_INDEX(ix);
  if(d_break_offsets == nullptr || total_breaks == nullptr || (d_breaks == nullptr && capacity > 0)) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "null buffer"); }
  if(nq >= (u64(1) << 31) - 1) { return fail(GCSA2_ERR_BUFFER_TOO_SMALL, "batch of >= 2^31 patterns; split the batch"); }
  DeviceGuard guard(ix->device);
  hipStream_t st = static_cast<hipStream_t>(stream);
  *total_breaks = 0;
  if(nq == 0 || ix->img.n == 0)
  {
    HIP_TRY(hipMemsetAsync(d_break_offsets, 0, (nq + 1) * sizeof(u64), st));
    HIP_TRY(hipStreamSynchronize(st));
    return GCSA2_OK;
  }
  const auto t_start = std::chrono::steady_clock::now();
  auto since = [&]() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_start).count(); };
  Scratch scratch(ix, st);
  // slots of the temporary records: what the caller's buffer holds + one block per wavefront that may leave a hole
  // (the wavefronts that can leave one: with a lane per pattern every wave of the grid, on the persistent lanes -- what
  // match_stats_launch picks for a batch of more than two generations of workgroups -- only the resident ones; and never more
  // records than one per pattern position and one per pattern, whatever the caller's capacity is: ADVICE r04)
  const u64 lane_waves = (nq + 63) / 64, lanes_grid = (nq + TPB2 - 1) / TPB2;
  const u64 resident = ix->tune.ms_grid != 0 ? ix->tune.ms_grid : u64(ix->compute_units) * 8;
  const bool persistent = (variant == 5 || (variant == 0 && lanes_grid > 2 * resident));
  const u64 waves = (persistent && resident * (TPB2 / 64) < lane_waves ? resident * (TPB2 / 64) : lane_waves);
  const u64 most = (total_bytes != GCSA2_UNKNOWN && total_bytes + nq < capacity ? total_bytes + nq : capacity);
  const u64 tmp_slots = most + (waves + 1) * BREAK_BLOCK;
  BreakSink sink{nullptr, tmp_slots, nullptr, nullptr, u32(min_length > 0xFFFFFFFFull ? 0xFFFFFFFFull : min_length)};
  u64 *wide = nullptr, *own_ranges = nullptr;
  const unsigned slot = ix->next_slot.fetch_add(1) % RESULT_SLOTS;
  unsigned long long* d_totals = ix->d_slots + u64(TOTAL_WORDS) * slot;
  sink.counter = d_totals;
  HIP_TRY(scratch.get(sink.tmp, tmp_slots * BREAK_WORDS));
  HIP_TRY(scratch.get(sink.counts, nq));
  HIP_TRY(scratch.get(wide, nq + 1));
  if(d_ranges == nullptr) { HIP_TRY(scratch.get(own_ranges, 2 * nq)); d_ranges = own_ranges; }
  HIP_TRY(hipMemsetAsync(d_totals, 0, TOTAL_WORDS * sizeof(unsigned long long), st));
  HIP_TRY(hipMemsetAsync(sink.counts, 0, nq * sizeof(u32), st));
  const double t_alloc = since();
  int rc = match_stats_launch(ix, variant, d_patterns, d_offsets, nq, total_bytes, nullptr, d_ranges, d_fallbacks, st, &sink);
  if(rc != GCSA2_OK) { return rc; }
  const double t_launched = since();
  hipLaunchKernelGGL(k_widen_counts, dim3(grid_for(nq + 1)), dim3(TPB), 0, st, sink.counts, nq, wide);
  size_t scan_bytes = 0;
  HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, wide, d_break_offsets, size_t(nq + 1), st));
  char* scan_tmp = nullptr;
  HIP_TRY(scratch.get(scan_tmp, scan_bytes));
  HIP_TRY(hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, wide, d_break_offsets, size_t(nq + 1), st));
  hipLaunchKernelGGL(k_copy_word, dim3(1), dim3(1), 0, st, d_break_offsets + nq, reinterpret_cast<u64*>(d_totals + 1));
  unsigned long long totals[TOTAL_WORDS];
  rc = read_totals(ix, slot, totals, st);
  if(rc != GCSA2_OK) { return rc; }
  scratch.settled = true;
  const double t_totals = since();
  const u64 reserved = totals[0], found = totals[1];          // slots the wavefronts reserved (with holes), records in all
  *total_breaks = found;
  if(found > capacity) { return fail(GCSA2_ERR_BUFFER_TOO_SMALL, "break buffer too small"); }
  //  the record scratch is sized from the caller's `total_bytes`: if that figure was too small the wavefronts
  // reserved slots beyond the scratch and dropped those records -- refused, not returned as a CSR with holes
  if(reserved > tmp_slots)
  {
    return fail(GCSA2_ERR_INVALID_ARGUMENT, "match_breaks: total_bytes is smaller than the patterns' total length (more break records than the scratch sized from it holds)");
  }
  if(found > 0)
  {
    scratch.settled = false;
    const u64 stored = reserved;
    hipLaunchKernelGGL(k_breaks_scatter, dim3(grid_for(stored)), dim3(TPB), 0, st, sink.tmp, stored, d_break_offsets, reinterpret_cast<u64*>(d_breaks), capacity);
    LAUNCH_CHECK("k_breaks_scatter");
  }
  HIP_TRY(hipStreamSynchronize(st));
  scratch.settled = true;
  if(ix->tune.locate_trace)
  {
    std::fprintf(stderr, "[breaks] scratch %.0f us | launched %.0f | totals %.0f | done %.0f (arena %zu of %zu bytes, %zu extra allocations)\n",
                 t_alloc, t_launched, t_totals, since(), scratch.used, scratch.arena.bytes, scratch.extra.size());
  }
  return GCSA2_OK;
}

// The break points of a batch in host memory: one copy in, gcsa2_match_breaks_device, the CSR out.
int gcsa2_match_breaks_batch(const gcsa2_index* ix, const uint8_t* patterns, const uint64_t* offsets, uint64_t nq, uint64_t min_length,
                             uint64_t* break_offsets, gcsa2_break* breaks, uint64_t capacity, uint64_t* total_breaks,
                             uint64_t* ranges, uint64_t* fallbacks)
{
  CHECK_INDEX(ix);
  if(nq == 0)             // an empty batch: no records, whatever the (possibly null) input arrays are
  {
    if(total_breaks != nullptr) { *total_breaks = 0; }
    if(break_offsets != nullptr) { break_offsets[0] = 0; }
    return GCSA2_OK;
  }
  if(offsets == nullptr || break_offsets == nullptr || total_breaks == nullptr || (breaks == nullptr && capacity > 0) || (patterns == nullptr && offsets[nq] > offsets[0]))
  {
    return fail(GCSA2_ERR_INVALID_ARGUMENT, "null buffer");
  }
  *total_breaks = 0;
  u64 longest = 0;
  if(offsets[0] != 0 || !offsets_ok(offsets, nq, &longest)) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "pattern offsets must start at 0 and be non-decreasing"); }
  try
  {
    if(ix->tune.ms_pieces && offsets[nq] >= 2 * ix->tune.ms_piece_bytes && longest <= ix->tune.ms_piece_bytes)
    {
      return match_breaks_pieced(ix, patterns, offsets, nq, min_length, break_offsets, breaks, capacity, total_breaks, ranges, fallbacks);
    }
    DeviceGuard guard(ix->device);
    const u64 total = offsets[nq];
    Lease lease(ix);
    HIP_TRY(lease.begin(Lease::need(total + 16) + Lease::need((nq + 1) * 8) + Lease::need((nq + 1) * 8) + Lease::need((capacity > 0 ? capacity : 1) * 32)
                        + Lease::need(2 * nq * 8) + Lease::need(nq * 8)));
    u8* d_pat = lease.dev<u8>(total + 16); u64* d_off = lease.dev<u64>(nq + 1); u64* d_boff = lease.dev<u64>(nq + 1);
    gcsa2_break* d_brk = reinterpret_cast<gcsa2_break*>(lease.dev<u64>(4 * (capacity > 0 ? capacity : 1)));
    u64* d_rng = lease.dev<u64>(2 * nq); u64* d_fb = lease.dev<u64>(nq);
    HIP_TRY(lease.up(d_pat, patterns, total));
    HIP_TRY(lease.up(d_off, offsets, (nq + 1) * sizeof(u64)));
    int rc = gcsa2_match_breaks_device(ix, d_pat, d_off, nq, total, 0, min_length, d_boff, d_brk, capacity, total_breaks, d_rng, d_fb, lease.stream());
    if(rc != GCSA2_OK) { return rc; }
    HIP_TRY(lease.down(break_offsets, d_boff, (nq + 1) * sizeof(u64)));
    if(*total_breaks > 0) { HIP_TRY(lease.down(breaks, d_brk, *total_breaks * sizeof(gcsa2_break))); }
    if(ranges != nullptr) { HIP_TRY(lease.down(ranges, d_rng, 2 * nq * sizeof(u64))); }
    if(fallbacks != nullptr) { HIP_TRY(lease.down(fallbacks, d_fb, nq * sizeof(u64))); }
    HIP_TRY(lease.finish());
    return GCSA2_OK;
  }
  catch(const std::exception& e) { return fail(GCSA2_ERR_OUT_OF_MEMORY, std::string("gcsa2_match_breaks_batch: ") + e.what()); }
}

// Diagnostic: the default kernel instrumented with shader-clock counters per phase of its round (k_match_stats2<.., PROF>),
// same results; d_prof[0..15] (zeroed by the caller) receives the cycle sums and event counts listed at the kernel.
extern "C" int gcsa2_match_stats_profile_device(const gcsa2_index* ix, const uint8_t* d_patterns, const uint64_t* d_offsets, uint64_t nq,
                                                uint64_t total_bytes, uint16_t* d_ms, uint64_t* d_ranges, uint64_t* d_fallbacks,
                                                uint64_t* d_prof, void* stream)
{
  CHECK_INDEX(ix);
  DeviceGuard guard(ix->device);
  if(!ix->img.has_lcp || ix->img.flp == nullptr) { return fail(GCSA2_ERR_MISSING_COMPONENT, "the profiled kernel needs the LCP array and the pair blocks"); }
  if(d_prof == nullptr) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "null profile buffer"); }
  if(nq == 0 || ix->img.n == 0) { return GCSA2_OK; }
  hipStream_t st = static_cast<hipStream_t>(stream);
  ulonglong2* recs = nullptr;
  HIP_TRY(pool_alloc(ix, reinterpret_cast<void**>(&recs), ((total_bytes >> 5) + nq + 6) * sizeof(ulonglong2), st));
  hipLaunchKernelGGL(k_pack_records, dim3(grid_for(nq)), dim3(TPB), 0, st, ix->img, d_patterns, d_offsets, nq, recs);
  hipLaunchKernelGGL((k_match_stats2<true, false, true>), dim3(unsigned((nq + TPB2 - 1) / TPB2)), dim3(TPB2), 0, st,
                     ix->img, d_patterns, d_offsets, nq, reinterpret_cast<unsigned short*>(d_ms), d_ranges, d_fallbacks, ix->tune.cool_down,
                     (unsigned long long*)nullptr, 64u, recs, reinterpret_cast<unsigned long long*>(d_prof));
  hipError_t le = hipGetLastError();
  (void)hipFreeAsync(recs, st);
  if(le != hipSuccess) { return fail(GCSA2_ERR_HIP, std::string("k_match_stats2<prof>: ") + hipGetErrorString(le)); }
  return GCSA2_OK;
}

extern "C" int gcsa2_match_stats_device_sized(const gcsa2_index* ix, int variant, const uint8_t* d_patterns, const uint64_t* d_offsets, uint64_t nq,
                                              uint64_t total_pattern_bytes, uint16_t* d_ms, uint64_t* d_ranges, uint64_t* d_fallbacks, void* stream)
{
  return match_stats_launch(ix, variant, d_patterns, d_offsets, nq, total_pattern_bytes, d_ms, d_ranges, d_fallbacks, static_cast<hipStream_t>(stream));
}

namespace {

// one copy in, one launch, copies out (offsets[0] = 0)
int match_stats_single(const gcsa2_index* ix, const uint8_t* patterns, const uint64_t* offsets, uint64_t nq, u64 longest,
                       uint16_t* ms, uint64_t* ranges, uint64_t* fallbacks)
{
  DeviceGuard guard(ix->device);
  const u64 total = offsets[nq];
  Lease lease(ix);
  HIP_TRY(lease.begin(Lease::need(total + 16) + Lease::need((nq + 1) * 8) + Lease::need(2 * nq * 8) + Lease::need(nq * 8) + Lease::need((total + 8) * 2)));
  u8* d_pat = lease.dev<u8>(total + 16); u64* d_off = lease.dev<u64>(nq + 1); u64* d_rng = lease.dev<u64>(2 * nq);
  u64* d_fb = lease.dev<u64>(nq); uint16_t* d_ms = lease.dev<uint16_t>(total + 8);
  HIP_TRY(lease.up(d_pat, patterns, total));
  HIP_TRY(lease.up(d_off, offsets, (nq + 1) * sizeof(u64)));
  // ragged batches (longest pattern > 1.25 x the mean) large enough to fill the device go to the persistent lanes
  const bool ragged = nq >= MS_REFILL_MIN && double(longest) * double(nq) > 1.25 * double(total);
  int rc = match_stats_launch(ix, ragged ? 5 : 0, d_pat, d_off, nq, total, d_ms, d_rng, d_fb, lease.stream());
  if(rc != GCSA2_OK) { return rc; }
  HIP_TRY(lease.down(ms, d_ms, total * sizeof(uint16_t)));
  HIP_TRY(lease.down(ranges, d_rng, 2 * nq * sizeof(u64)));
  if(fallbacks != nullptr) { HIP_TRY(lease.down(fallbacks, d_fb, nq * sizeof(u64))); }
  HIP_TRY(lease.finish());
  return GCSA2_OK;
}

// A large host batch moves three times the pattern bytes over the link (one byte in, two out per position) and the kernel
// is faster than that: it is cut into pieces of tune.ms_piece_bytes pattern bytes which MS_PIECE_THREADS host threads send through
// the single-copy path, each on the stream and arenas of its own lease -- one piece's upload, another's kernel and a third's
// download overlap.  (1 M x 256 bp on the chr22-like index: 22.4 ms = 45 M patterns/s in a row, 17.5 ms = 57 M/s in pieces;
// eight threads or 8 MB pieces change nothing: what is left is the rate of copies to and from pageable memory,
// tests/perf/ms_host_batch.py.  Staging those copies ourselves through a ring of pinned 2 MB pieces was measured and is
// twice as slow as the runtime's own path for pageable memory: 47 ms in a row, 24-28 ms in pieces.)
// (pieces of tune.ms_piece_bytes = 32 MB (GCSA2_MS_PIECE_MB) for batches of two pieces' worth (64 MB) and more, on tune.ms_threads = 4 host threads: declared with
// the forward declarations above.  Round 4 re-measured the piece size, 1 M x 256 bp: dense statistics on the chr22-like index 15.9-17.4 ms
// with 16 MB, 15.0 with 32, 16.1 with 64, 18.9 with 128, 22.3 in one copy; break points of at least 20 bp on the 5.73 G-node index 24.5 /
// 16.7 / 16.9 / 19.5 ms and 17.9 in one copy -- a piece must still fill the device: the kernel's time per pattern is latency, not work;
// profiles/r04_host.md)

int match_stats_pieced(const gcsa2_index* ix, const uint8_t* patterns, const uint64_t* offsets, uint64_t nq, u64 longest,
                       uint16_t* ms, uint64_t* ranges, uint64_t* fallbacks)
{
  std::vector<u64> cut(1, 0);
  while(cut.back() < nq)
  {
    const u64 b = cut.back();
    u64 lo = b + 1, hi = nq;                               // largest e with offsets[e] - offsets[b] <= the piece size, at least one pattern
    while(lo < hi) { const u64 mid = (lo + hi + 1) / 2; if(offsets[mid] - offsets[b] <= ix->tune.ms_piece_bytes) { lo = mid; } else { hi = mid - 1; } }
    cut.push_back(lo);
  }
  const u64 pieces = cut.size() - 1;
  const unsigned threads = unsigned(pieces < ix->tune.ms_threads ? pieces : ix->tune.ms_threads);
  std::vector<int> status(threads, GCSA2_OK);
  std::vector<std::string> messages(threads);
  auto work = [&](unsigned t)
  {
    try                                   // (an exception must not leave a worker thread: std::terminate)
    {
      std::vector<u64> local;
      for(u64 c = t; c < pieces && status[t] == GCSA2_OK; c += threads)
      {
        const u64 b = cut[c], count = cut[c + 1] - b, base = offsets[b];
        local.resize(count + 1);
        for(u64 i = 0; i <= count; i++) { local[i] = offsets[b + i] - base; }
        status[t] = match_stats_single(ix, patterns + base, local.data(), count, longest, ms + base, ranges + 2 * b,
                                       fallbacks != nullptr ? fallbacks + b : nullptr);
        if(status[t] != GCSA2_OK) { messages[t] = g_error; }
      }
    }
    catch(const std::exception& e) { status[t] = GCSA2_ERR_OUT_OF_MEMORY; messages[t] = std::string("a piece of the batch: ") + e.what(); }
  };
  Workers workers;
  for(unsigned t = 1; t < threads; t++) { workers.emplace_back(work, t); }
  work(0);
  workers.join();
  for(unsigned t = 0; t < threads; t++) { if(status[t] != GCSA2_OK) { return fail(status[t], messages[t]); } }
  return GCSA2_OK;
}

// The break points of a large host batch, in pieces like the dense statistics above (a MEM finder's reads live in host memory:
// one copy in, one kernel, copies out leaves the device idle two thirds of the time).  A piece's records land in the caller's
// array behind those of the pieces before it, so the pieces COMMIT in order: when its kernel has finished a piece knows its
// number of records, waits for the running total of the pieces before it (they started earlier; the wait is the tail of one
// kernel at most), takes its place and downloads there.  The device-side record buffer of a piece is sized by an estimate
// (one record per 8 pattern bytes; a piece that needs more runs again with the exact size).  Once the caller's capacity is
// exceeded the remaining pieces only count: GCSA2_ERR_BUFFER_TOO_SMALL with the number of records of the whole batch.
int match_breaks_pieced(const gcsa2_index* ix, const uint8_t* patterns, const uint64_t* offsets, uint64_t nq, uint64_t min_length,
                        uint64_t* break_offsets, gcsa2_break* breaks, uint64_t capacity, uint64_t* total_breaks, uint64_t* ranges, uint64_t* fallbacks)
{
  std::vector<u64> cut(1, 0);
  while(cut.back() < nq)
  {
    const u64 b = cut.back();
    u64 lo = b + 1, hi = nq;                               // largest e with offsets[e] - offsets[b] <= the piece size, at least one pattern
    while(lo < hi) { const u64 mid = (lo + hi + 1) / 2; if(offsets[mid] - offsets[b] <= ix->tune.ms_piece_bytes) { lo = mid; } else { hi = mid - 1; } }
    cut.push_back(lo);
  }
  const u64 pieces = cut.size() - 1;
  const unsigned threads = unsigned(pieces < ix->tune.ms_threads ? pieces : ix->tune.ms_threads);
  std::vector<int> status(threads, GCSA2_OK);
  std::vector<std::string> messages(threads);
  struct Order { std::mutex m; std::condition_variable cv; u64 next = 0, base = 0; bool failed = false; } order;
  auto work = [&](unsigned t)
  {
    DeviceGuard guard(ix->device);
    std::vector<u64> local;
    auto give_up = [&](int rc, const std::string& what)
    {
      status[t] = rc; messages[t] = what;
      std::lock_guard<std::mutex> hold(order.m);
      order.failed = true;
      order.cv.notify_all();
    };
    try {
    for(u64 c = t; c < pieces && status[t] == GCSA2_OK; c += threads)
    {
      const u64 b = cut[c], count = cut[c + 1] - b, first = offsets[b], bytes = offsets[b + count] - first;
      local.resize(count + 1);
      for(u64 i = 0; i <= count; i++) { local[i] = offsets[b + i] - first; }
      u64 room = bytes / 8 + count + 1, found = 0;
      bool placed = false;
      for(int attempt = 0; attempt < 2 && !placed && status[t] == GCSA2_OK; attempt++)
      {
        Lease lease(ix);
        hipError_t e = lease.begin(Lease::need(bytes + 16) + 2 * Lease::need((count + 1) * 8) + Lease::need(room * 32) + Lease::need(2 * count * 8) + Lease::need(count * 8));
        if(e != hipSuccess) { give_up(GCSA2_ERR_OUT_OF_MEMORY, std::string("staging of a piece: ") + hipGetErrorString(e)); break; }
        u8* d_pat = lease.dev<u8>(bytes + 16); u64* d_off = lease.dev<u64>(count + 1); u64* d_boff = lease.dev<u64>(count + 1);
        gcsa2_break* d_brk = reinterpret_cast<gcsa2_break*>(lease.dev<u64>(4 * room));
        u64* d_rng = lease.dev<u64>(2 * count); u64* d_fb = lease.dev<u64>(count);
        e = lease.up(d_pat, patterns + first, bytes);
        if(e == hipSuccess) { e = lease.up(d_off, local.data(), (count + 1) * sizeof(u64)); }
        if(e != hipSuccess) { give_up(GCSA2_ERR_HIP, std::string("upload of a piece: ") + hipGetErrorString(e)); break; }
        const int rc = gcsa2_match_breaks_device(ix, d_pat, d_off, count, bytes, 0, min_length, d_boff, d_brk, room, &found, d_rng, d_fb, lease.stream());
        if(rc == GCSA2_ERR_BUFFER_TOO_SMALL && attempt == 0 && found > room) { room = found; (void)lease.finish(); continue; }    // the estimate was short: once more, exactly
        if(rc != GCSA2_OK) { give_up(rc, g_error); break; }
        // commit in piece order: the position of this piece's first record
        u64 at = 0;
        {
          std::unique_lock<std::mutex> hold(order.m);
          order.cv.wait(hold, [&]() { return order.next == c || order.failed; });
          if(order.failed) { status[t] = GCSA2_ERR_HIP; messages[t] = "another piece of the batch failed"; break; }
          at = order.base; order.base += found; order.next = c + 1;
          order.cv.notify_all();
        }
        placed = true;
        if(at + found <= capacity)
        {
          e = lease.down(local.data(), d_boff, (count + 1) * sizeof(u64));
          if(e == hipSuccess && found > 0) { e = lease.down(breaks + at, d_brk, found * sizeof(gcsa2_break)); }
          if(e == hipSuccess && ranges != nullptr) { e = lease.down(ranges + 2 * b, d_rng, 2 * count * sizeof(u64)); }
          if(e == hipSuccess && fallbacks != nullptr) { e = lease.down(fallbacks + b, d_fb, count * sizeof(u64)); }
          if(e == hipSuccess) { e = lease.finish(); }
          if(e != hipSuccess) { give_up(GCSA2_ERR_HIP, std::string("download of a piece: ") + hipGetErrorString(e)); break; }
          for(u64 i = 0; i <= count; i++) { break_offsets[b + i] = local[i] + at; }     // (the last entry is the next piece's first: same value)
        }
        else { (void)lease.finish(); }
      }
    }
    } catch(const std::exception& e) { give_up(GCSA2_ERR_OUT_OF_MEMORY, std::string("a piece of the batch: ") + e.what()); }     // (not out of a worker thread)
  };
  Workers workers;
  for(unsigned t = 1; t < threads; t++) { workers.emplace_back(work, t); }
  work(0);
  workers.join();
  for(unsigned t = 0; t < threads; t++) { if(status[t] != GCSA2_OK && messages[t] != "another piece of the batch failed") { return fail(status[t], messages[t]); } }
  for(unsigned t = 0; t < threads; t++) { if(status[t] != GCSA2_OK) { return fail(status[t], messages[t]); } }
  *total_breaks = order.base;
  if(order.base > capacity) { return fail(GCSA2_ERR_BUFFER_TOO_SMALL, "break buffer too small"); }
  return GCSA2_OK;
}

}  // namespace

extern "C" int gcsa2_match_stats_batch(const gcsa2_index* ix, const uint8_t* patterns, const uint64_t* offsets, uint64_t nq,
                                       uint16_t* ms, uint64_t* ranges, uint64_t* fallbacks)
{
  CHECK_INDEX(ix);
  if(nq == 0) { return GCSA2_OK; }
  if(offsets == nullptr || ms == nullptr || ranges == nullptr) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "null buffer"); }
  u64 longest = 0;
  if(!offsets_ok(offsets, nq, &longest)) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "pattern offsets are not non-decreasing"); }
  try
  {
    if(ix->tune.ms_pieces && offsets[0] == 0 && offsets[nq] >= 2 * ix->tune.ms_piece_bytes && longest <= ix->tune.ms_piece_bytes)
    {
      return match_stats_pieced(ix, patterns, offsets, nq, longest, ms, ranges, fallbacks);
    }
  }
  catch(const std::exception& e) { return fail(GCSA2_ERR_OUT_OF_MEMORY, std::string("gcsa2_match_stats_batch: ") + e.what()); }
  return match_stats_single(ix, patterns, offsets, nq, longest, ms, ranges, fallbacks);
}

// ---- host-view container file ("G2HV"): the interchange format between a process that can read
// .gcsa / .lcp files (the reference + SDSL) and GPU nodes that cannot. --------
// Layout (little endian): char magic[4] = "G2HV"; u32 version = 1; u64 header[12] = path_nodes,
// edges, order, sigma, fast_chars, sample_count, sample_width, extra_values_len, redundant_len,
// lcp_size, lcp_branching, lcp_levels; u64 flags (bit 0 samples, bit 1 counters, bit 2 lcp);
// then the arrays in the order of gcsa2_host_view, each as u64 byte length + bytes padded to 8.

struct gcsa2_view_storage
{
  gcsa2_host_view view;
  std::vector<std::vector<u64>> blobs;    // 8-byte aligned backing store
  std::vector<const u64*> bwt;
  std::vector<uint8_t> comp2char;         // alpha.comp2char as the file had it (GCSA::load keeps it, so serialize() must too)
};

namespace {

constexpr u32 G2HV_VERSION = 1;

u64 words_for_bits(u64 bits) { return (bits + 63) / 64; }

bool write_blob(FILE* f, const void* data, u64 bytes)
{
  u64 padded = (bytes + 7) & ~u64(7);
  static const char zeros[8] = {0};
  if(fwrite(&bytes, 8, 1, f) != 1) { return false; }
  if(bytes > 0 && fwrite(data, 1, bytes, f) != bytes) { return false; }
  if(padded > bytes && fwrite(zeros, 1, padded - bytes, f) != padded - bytes) { return false; }
  return true;
}

bool read_blob(FILE* f, std::vector<u64>& dst, u64& bytes)
{
  if(fread(&bytes, 8, 1, f) != 1) { return false; }
  if(bytes > (u64(1) << 40)) { return false; }
  dst.assign((bytes + 7) / 8 + 2, 0);        // two spare words: word-granular readers may overrun
  u64 padded = (bytes + 7) & ~u64(7);
  return padded == 0 || fread(dst.data(), 1, padded, f) == padded;
}

}  // namespace

extern "C" int gcsa2_host_view_save(const gcsa2_host_view* v, const char* path)
{
  if(v == nullptr || path == nullptr || v->char2comp == nullptr || v->C == nullptr || v->bwt == nullptr || v->edge_bits == nullptr)
  {
    return fail(GCSA2_ERR_INVALID_ARGUMENT, "incomplete host view");
  }
  FILE* f = fopen(path, "wb");
  if(f == nullptr) { return fail(GCSA2_ERR_INVALID_ARGUMENT, std::string("cannot open ") + path); }
  bool ok = true;
  u64 flags = (v->sampled_path_bits ? 1 : 0) | (v->extra_filter_bits ? 2 : 0) | (v->lcp_data ? 4 : 0);
  u64 header[12] = { v->path_nodes, v->edges, v->order, v->sigma, v->fast_chars, v->sample_count, v->sample_width,
                     v->extra_values_len, v->redundant_len, v->lcp_size, v->lcp_branching, v->lcp_levels };
  ok = ok && fwrite("G2HV", 1, 4, f) == 4 && fwrite(&G2HV_VERSION, 4, 1, f) == 1;
  ok = ok && fwrite(header, 8, 12, f) == 12 && fwrite(&flags, 8, 1, f) == 1;
  ok = ok && write_blob(f, v->char2comp, 256) && write_blob(f, v->C, (v->sigma + 1) * 8);
  for(u64 c = 0; ok && c < v->sigma; c++) { ok = write_blob(f, v->bwt[c], words_for_bits(v->path_nodes) * 8); }
  ok = ok && write_blob(f, v->edge_bits, words_for_bits(v->edges) * 8);
  if(flags & 1)
  {
    ok = ok && write_blob(f, v->sampled_path_bits, words_for_bits(v->path_nodes) * 8);
    ok = ok && write_blob(f, v->stored_samples, words_for_bits(v->sample_count * v->sample_width) * 8);
    ok = ok && write_blob(f, v->sample_bits, words_for_bits(v->sample_count) * 8);
  }
  if(flags & 2)
  {
    ok = ok && write_blob(f, v->extra_filter_bits, words_for_bits(v->path_nodes) * 8);
    ok = ok && write_blob(f, v->extra_values_bits, words_for_bits(v->extra_values_len) * 8);
    ok = ok && write_blob(f, v->redundant_bits, words_for_bits(v->redundant_len) * 8);
  }
  if(flags & 4)
  {
    ok = ok && write_blob(f, v->lcp_offsets, (v->lcp_levels + 1) * 8);
    ok = ok && write_blob(f, v->lcp_data, v->lcp_offsets[v->lcp_levels]);
  }
  ok = (fclose(f) == 0) && ok;
  return ok ? GCSA2_OK : fail(GCSA2_ERR_INVALID_ARGUMENT, std::string("write error on ") + path);
}

extern "C" void gcsa2_host_view_free(gcsa2_view_storage* storage) { delete storage; }

extern "C" const gcsa2_host_view* gcsa2_host_view_get(const gcsa2_view_storage* storage) { return storage ? &storage->view : nullptr; }

extern "C" int gcsa2_host_view_load(const char* path, gcsa2_view_storage** out)
{
  if(path == nullptr || out == nullptr) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "null argument"); }
  *out = nullptr;
  FILE* f = fopen(path, "rb");
  if(f == nullptr) { return fail(GCSA2_ERR_INVALID_ARGUMENT, std::string("cannot open ") + path); }
  struct Closer { FILE* f; ~Closer() { fclose(f); } } closer{f};
  try
  {
    char magic[4]; u32 version = 0; u64 header[12], flags = 0;
    if(fread(magic, 1, 4, f) != 4 || std::memcmp(magic, "G2HV", 4) != 0) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "not a G2HV file: invalid tag"); }
    if(fread(&version, 4, 1, f) != 1 || version != G2HV_VERSION) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "G2HV: unsupported version"); }
    if(fread(header, 8, 12, f) != 12 || fread(&flags, 8, 1, f) != 1) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "G2HV: truncated header"); }
    if(header[3] == 0 || header[3] > GCSA2_MAX_SIGMA || header[11] > u64(MAX_LCP_LEVELS)) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "G2HV: header out of range"); }
    std::unique_ptr<gcsa2_view_storage> st(new gcsa2_view_storage());
    gcsa2_host_view& v = st->view;
    std::memset(&v, 0, sizeof(v));
    v.path_nodes = header[0]; v.edges = header[1]; v.order = header[2]; v.sigma = header[3]; v.fast_chars = header[4];
    v.sample_count = header[5]; v.sample_width = header[6]; v.extra_values_len = header[7]; v.redundant_len = header[8];
    v.lcp_size = header[9]; v.lcp_branching = header[10]; v.lcp_levels = header[11];
    u64 nblobs = 2 + v.sigma + 1 + ((flags & 1) ? 3 : 0) + ((flags & 2) ? 3 : 0) + ((flags & 4) ? 2 : 0);
    st->blobs.resize(nblobs);
    std::vector<u64> sizes(nblobs, 0);
    for(u64 b = 0; b < nblobs; b++)
    {
      if(!read_blob(f, st->blobs[b], sizes[b])) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "G2HV: truncated array"); }
    }
    u64 b = 0;
    auto expect = [&](u64 bytes) -> bool { return sizes[b] == bytes; };
    if(!expect(256)) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "G2HV: char2comp size"); }
    v.char2comp = reinterpret_cast<const uint8_t*>(st->blobs[b++].data());
    if(!expect((v.sigma + 1) * 8)) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "G2HV: C size"); }
    v.C = st->blobs[b++].data();
    for(u64 c = 0; c < v.sigma; c++)
    {
      if(!expect(words_for_bits(v.path_nodes) * 8)) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "G2HV: bwt size"); }
      st->bwt.push_back(st->blobs[b++].data());
    }
    v.bwt = st->bwt.data();
    if(!expect(words_for_bits(v.edges) * 8)) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "G2HV: edges size"); }
    v.edge_bits = st->blobs[b++].data();
    if(flags & 1)
    {
      if(v.sample_width == 0 || v.sample_width > 64 || v.sample_count > (u64(1) << 46)) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "G2HV: sample header out of range"); }
      if(!expect(words_for_bits(v.path_nodes) * 8)) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "G2HV: sampled_paths size"); }
      v.sampled_path_bits = st->blobs[b++].data();
      if(!expect(words_for_bits(v.sample_count * v.sample_width) * 8)) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "G2HV: stored_samples size"); }
      v.stored_samples = st->blobs[b++].data();
      if(!expect(words_for_bits(v.sample_count) * 8)) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "G2HV: samples size"); }
      v.sample_bits = st->blobs[b++].data();
    }
    if(flags & 2)
    {
      if(!expect(words_for_bits(v.path_nodes) * 8)) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "G2HV: extra_pointers.filter size"); }
      v.extra_filter_bits = st->blobs[b++].data();
      if(!expect(words_for_bits(v.extra_values_len) * 8)) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "G2HV: extra_pointers.values size"); }
      v.extra_values_bits = st->blobs[b++].data();
      if(!expect(words_for_bits(v.redundant_len) * 8)) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "G2HV: redundant_pointers size"); }
      v.redundant_bits = st->blobs[b++].data();
    }
    if(flags & 4)
    {
      if(!expect((v.lcp_levels + 1) * 8)) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "G2HV: lcp offsets size"); }
      v.lcp_offsets = st->blobs[b++].data();
      if(sizes[b] != v.lcp_offsets[v.lcp_levels]) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "G2HV: lcp data size"); }
      v.lcp_data = reinterpret_cast<const uint8_t*>(st->blobs[b++].data());
    }
    *out = st.release();
    return GCSA2_OK;
  }
  catch(const std::exception& e) { return fail(GCSA2_ERR_OUT_OF_MEMORY, std::string("gcsa2_host_view_load: ") + e.what()); }
}

extern "C" int gcsa2_index_create_from_file(const char* path, int device, gcsa2_index** out)
{
  gcsa2_view_storage* st = nullptr;
  int rc = gcsa2_host_view_load(path, &st);
  if(rc != GCSA2_OK) { return rc; }
  rc = gcsa2_index_create(&st->view, device, out);
  gcsa2_host_view_free(st);
  return rc;
}

// compareKMers(left, right, k, parameters) (src/algorithms.cpp:534-616).  With record buffers the states of
// the unique k-mers are returned as the reference dumps them to output.left / output.right (:606-610).
namespace {

// The search below one piece of the frontier (pairs of ranges, 4 u64 per state; with records also the 3-u64 keys), depth-first
// over pieces like countKMers (kmer_run): a frontier of any size, where round 3 refused more than 2^27 / limit states.
struct CompareRun
{
  const DevImage* images; unsigned long long* counters; u32 limit; u64 k; bool records; u64 piece;
  uint64_t* left_records; uint64_t left_capacity; uint64_t* right_records; uint64_t right_capacity;
  u64 shared = 0, left_only = 0, right_only = 0;       // totals (the unique ones count on even when their records no longer fit)
  KmerBufs states, keys;
};

int compare_run(CompareRun& c, const u64* frontier, const u64* keys, u64 n, u64 depth)
{
  unsigned long long host_counters[4] = {0, 0, 0, 0};
  while(depth < c.k && n > 0)
  {
    HIP_TRY(hipMemset(c.counters, 0, 4 * sizeof(unsigned long long)));
    if(depth + 1 == c.k)        // last level: classify the children; any size
    {
      hipLaunchKernelGGL(k_kmer_compare, dim3(grid_for(n)), dim3(TPB), 0, nullptr, c.images, c.images + 1, frontier, keys, n, c.limit,
                         u32(depth), 1, (u64*)nullptr, (u64*)nullptr, c.counters, (u64*)nullptr, (u64*)nullptr);
      LAUNCH_CHECK("k_kmer_compare");
      HIP_TRY(hipMemcpy(host_counters, c.counters, sizeof(host_counters), hipMemcpyDeviceToHost));
      const u64 l = host_counters[2], r = host_counters[3];
      if(c.records && (l > 0 || r > 0) && c.left_only + l <= c.left_capacity && c.right_only + r <= c.right_capacity)
      {
        DBuf<u64> d_left, d_right;
        HIP_TRY(d_left.alloc(8 * l)); HIP_TRY(d_right.alloc(8 * r));
        HIP_TRY(hipMemset(c.counters, 0, 4 * sizeof(unsigned long long)));
        hipLaunchKernelGGL(k_kmer_compare, dim3(grid_for(n)), dim3(TPB), 0, nullptr, c.images, c.images + 1, frontier, keys, n, c.limit,
                           u32(depth), 2, (u64*)nullptr, (u64*)nullptr, c.counters, d_left.p, d_right.p);
        LAUNCH_CHECK("k_kmer_compare");
        if(l > 0) { HIP_TRY(hipMemcpy(c.left_records + 8 * c.left_only, d_left.p, 8 * l * sizeof(u64), hipMemcpyDeviceToHost)); }
        if(r > 0) { HIP_TRY(hipMemcpy(c.right_records + 8 * c.right_only, d_right.p, 8 * r * sizeof(u64), hipMemcpyDeviceToHost)); }
      }
      c.shared += host_counters[1]; c.left_only += l; c.right_only += r;
      return GCSA2_OK;
    }
    if(n * c.limit > c.piece && n > 1)        // the children might not fit one buffer: halves, one after the other, in place
    {
      const u64 half = n / 2;
      int rc = compare_run(c, frontier, keys, half, depth);
      if(rc == GCSA2_OK) { rc = compare_run(c, frontier + 4 * half, (keys != nullptr ? keys + 3 * half : nullptr), n - half, depth); }
      return rc;
    }
    u64 *next = nullptr, *next_keys = nullptr;
    HIP_TRY(c.states.get(size_t(depth + 1), size_t(n * c.limit * 4 * sizeof(u64)), next));
    if(c.records) { HIP_TRY(c.keys.get(size_t(depth + 1), size_t(n * c.limit * 3 * sizeof(u64)), next_keys)); }
    hipLaunchKernelGGL(k_kmer_compare, dim3(grid_for(n)), dim3(TPB), 0, nullptr, c.images, c.images + 1, frontier, keys, n, c.limit,
                       u32(depth), 0, next, next_keys, c.counters, (u64*)nullptr, (u64*)nullptr);
    LAUNCH_CHECK("k_kmer_compare");
    HIP_TRY(hipMemcpy(host_counters, c.counters, sizeof(host_counters), hipMemcpyDeviceToHost));
    frontier = next; keys = next_keys; n = host_counters[0]; depth++;
  }
  return GCSA2_OK;
}

int compare_kmers_impl(const gcsa2_index* left, const gcsa2_index* right, uint64_t k, int include_ns, int force, uint64_t* result,
                       uint64_t* left_records, uint64_t left_capacity, uint64_t* right_records, uint64_t right_capacity, bool records)
{
  CHECK_INDEX(left); CHECK_INDEX(right);
  if(result == nullptr) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "null result"); }
  result[0] = result[1] = result[2] = 0;
  if(k == 0) { result[0] = 1; return GCSA2_OK; }                                            // :539
  if((k > left->order || k > right->order) && !force) { return GCSA2_OK; }                   // :540-549
  if(k > 64) { return GCSA2_OK; }                                                            // :550-554 (MAX_K)
  if(left->img.sigma != right->img.sigma || left->img.fast_chars != right->img.fast_chars) { return GCSA2_OK; }   // :556-560
  if(left->device != right->device) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "both indexes must live on the same device"); }
  if(left->img.n == 0 || right->img.n == 0 || left->img.sigma < 3) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "empty index or alphabet too small"); }
  try {
  DeviceGuard guard(left->device);
  DBuf<DevImage> images; DBuf<unsigned long long> counters;
  HIP_TRY(images.alloc(2)); HIP_TRY(counters.alloc(4));
  HIP_TRY(hipMemcpy(images.p, &left->img, sizeof(DevImage), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(images.p + 1, &right->img, sizeof(DevImage), hipMemcpyHostToDevice));
  CompareRun run{images.p, counters.p, u32(include_ns ? left->img.sigma - 2 : left->img.fast_chars), k, records, left->tune.kmer_piece,
                 left_records, left_capacity, right_records, right_capacity};
  u64 *frontier = nullptr, *keys = nullptr;
  HIP_TRY(run.states.get(0, 4 * sizeof(u64), frontier));
  const u64 root[4] = {0, left->img.n - 1, 0, right->img.n - 1};
  HIP_TRY(hipMemcpy(frontier, root, sizeof(root), hipMemcpyHostToDevice));
  if(records) { HIP_TRY(run.keys.get(0, 3 * sizeof(u64), keys)); HIP_TRY(hipMemset(keys, 0, 3 * sizeof(u64))); }
  const int rc = compare_run(run, frontier, keys, 1, 0);
  if(rc != GCSA2_OK) { return rc; }
  result[0] = run.shared; result[1] = run.left_only; result[2] = run.right_only;
  if(records && (result[1] > left_capacity || result[2] > right_capacity)) { return fail(GCSA2_ERR_BUFFER_TOO_SMALL, "record buffers smaller than result[1] / result[2]"); }
  return GCSA2_OK;
  } catch(const std::exception& e) { return fail(GCSA2_ERR_OUT_OF_MEMORY, std::string("compareKMers: ") + e.what()); }
}

} // namespace

extern "C" int gcsa2_compare_kmers(const gcsa2_index* left, const gcsa2_index* right, uint64_t k, int include_ns, int force,
                                   uint64_t* result)
{
  return compare_kmers_impl(left, right, k, include_ns, force, result, nullptr, 0, nullptr, 0, false);
}

extern "C" int gcsa2_compare_kmers_records(const gcsa2_index* left, const gcsa2_index* right, uint64_t k, int include_ns, int force,
                                           uint64_t* result, uint64_t* left_records, uint64_t left_capacity,
                                           uint64_t* right_records, uint64_t right_capacity)
{
  if((left_records == nullptr && left_capacity > 0) || (right_records == nullptr && right_capacity > 0)) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "null record buffer"); }
  return compare_kmers_impl(left, right, k, include_ns, force, result, left_records, left_capacity, right_records, right_capacity, true);
}

// ---- `.gcsa` / `.lcp` files as written by GCSA::serialize / LCPArray::serialize ---------------------
// Member order: src/gcsa.cpp:140-216 (GCSA), src/support.cpp:229-250 (Alphabet), :401-418 (SadaCount),
// :493-516 (SadaSparse), src/files.cpp:513-537 / :581-603 (headers), src/lcp.cpp:116-143 (LCPArray).
// SDSL container encodings: sdsl_reader.hpp (format parity unpinned, see its header).

namespace {

// `exact`: the stream must end with the structure (a file); otherwise trailing bytes are left to the caller
void load_gcsa_members(sdsl_file::Cursor& in, gcsa2_view_storage& st, bool exact)
{
  using namespace sdsl_file;
  gcsa2_host_view& v = st.view;

  // GCSAHeader (files.cpp:527-543): tag, version, path_nodes, edges, order, flags
  u32 tag = in.get<u32>("header.tag"), version = in.get<u32>("header.version");
  v.path_nodes = in.get<u64>("header.path_nodes"); v.edges = in.get<u64>("header.edges"); v.order = in.get<u64>("header.order");
  u64 flags = in.get<u64>("header.flags");
  if(tag != 0x6C5A6C5Au || version != 3 || flags != 0)
  {
    in.error("Invalid header: tag " + std::to_string(tag) + ", version " + std::to_string(version) + ", flags " + std::to_string(flags)
             + " (expected GCSA version 3 as written by GCSA::serialize / sdsl::store_to_file; a file wrapped in another"
               " container, e.g. a tagged stream, has to be unwrapped first)");
  }

  // Alphabet (support.cpp:243-250)
  IntVector char2comp = read_int_vector(in, 8, "alpha.char2comp");
  IntVector comp2char = read_int_vector(in, 8, "alpha.comp2char");
  IntVector C = read_int_vector(in, 64, "alpha.C");
  v.sigma = in.get<u64>("alpha.sigma"); v.fast_chars = in.get<u64>("alpha.fast_chars");
  if(char2comp.size() != 256) { in.error("alpha.char2comp must have 256 entries"); }
  if(v.sigma == 0 || v.sigma > GCSA2_MAX_SIGMA || C.size() != v.sigma + 1) { in.error("alphabet size out of range or alpha.C of the wrong length"); }
  if(v.fast_chars >= v.sigma) { in.error("alpha.fast_chars out of range"); }
  if(comp2char.size() == v.sigma)
  {
    std::vector<u64> words;
    comp2char.copy_words(words);
    st.comp2char.assign(reinterpret_cast<const uint8_t*>(words.data()), reinterpret_cast<const uint8_t*>(words.data()) + v.sigma);
    v.comp2char = st.comp2char.data();
  }

  st.blobs.resize(2 + v.sigma + 9);
  u64 b = 0;
  char2comp.copy_words(st.blobs[b]); v.char2comp = reinterpret_cast<const uint8_t*>(st.blobs[b++].data());
  C.copy_words(st.blobs[b]); v.C = st.blobs[b++].data();

  // fast_bwt, fast_rank (empty), sparse_bwt, sparse_rank (empty): gcsa.cpp:196-202.  LF reads fast_bwt for
  // 1 <= comp <= fast_chars and sparse_bwt otherwise (gcsa.h:262-274).
  std::vector<std::vector<u64>> fast(v.sigma), sparse(v.sigma);
  std::vector<u64> fast_size(v.sigma), sparse_size(v.sigma);
  for(u64 c = 0; c < v.sigma; c++) { read_bit_vector_il(in, fast[c], fast_size[c], "fast_bwt"); }
  for(u64 c = 0; c < v.sigma; c++) { read_sd_vector(in, sparse[c], sparse_size[c], "sparse_bwt"); }
  for(u64 c = 0; c < v.sigma; c++)
  {
    const bool is_fast = (c > 0 && c <= v.fast_chars);
    if((is_fast ? fast_size[c] : sparse_size[c]) != v.path_nodes) { in.error("BWT bitvector of comp " + std::to_string(c) + " does not have path_nodes bits"); }
    st.blobs[b].swap(is_fast ? fast[c] : sparse[c]);
    st.bwt.push_back(st.blobs[b++].data());
  }
  v.bwt = st.bwt.data();

  u64 size = 0;
  read_bit_vector_il(in, st.blobs[b], size, "edges");                   // + edge_rank (empty)
  if(size != v.edges) { in.error("edges does not have header.edges bits"); }
  v.edge_bits = st.blobs[b++].data();
  read_bit_vector_il(in, st.blobs[b], size, "sampled_paths");           // + sampled_path_rank (empty)
  if(size != v.path_nodes) { in.error("sampled_paths does not have path_nodes bits"); }
  v.sampled_path_bits = st.blobs[b++].data();

  IntVector stored = read_int_vector(in, 0, "stored_samples");
  v.sample_count = stored.size(); v.sample_width = stored.width;
  stored.copy_words(st.blobs[b]); v.stored_samples = st.blobs[b++].data();
  read_bit_vector(in, st.blobs[b], size, "samples");
  if(size != v.sample_count) { in.error("samples and stored_samples differ in length"); }
  v.sample_bits = st.blobs[b++].data();
  skip_select_mcl(in, "sample_select");

  // extra_pointers (SadaSparse, support.cpp:509-516): filter, filter_rank (empty), values, value_select (empty)
  read_sd_vector(in, st.blobs[b], size, "extra_pointers.filter");
  if(size != v.path_nodes) { in.error("extra_pointers.filter does not have path_nodes bits"); }
  v.extra_filter_bits = st.blobs[b++].data();
  read_sd_vector(in, st.blobs[b], v.extra_values_len, "extra_pointers.values");
  v.extra_values_bits = st.blobs[b++].data();
  // redundant_pointers (SadaCount, support.cpp:414-418): data, select
  read_bit_vector(in, st.blobs[b], v.redundant_len, "redundant_pointers.data");
  v.redundant_bits = st.blobs[b++].data();
  skip_select_mcl(in, "redundant_pointers.select");

  if(exact && !in.at_end()) { in.error(std::to_string(in.remaining()) + " unaccounted bytes after redundant_pointers"); }
}

void load_gcsa_members(const char* path, gcsa2_view_storage& st)
{
  sdsl_file::Mapping map(path);
  sdsl_file::Cursor in(map, std::string("GCSA::load(") + path + ")");
  load_gcsa_members(in, st, true);
}

void load_lcp_members(sdsl_file::Cursor& in, gcsa2_view_storage& st, bool exact)
{
  using namespace sdsl_file;
  gcsa2_host_view& v = st.view;

  // LCPHeader (files.cpp:595-609): tag, version, size, branching, flags
  u32 tag = in.get<u32>("header.tag"), version = in.get<u32>("header.version");
  v.lcp_size = in.get<u64>("header.size"); v.lcp_branching = in.get<u64>("header.branching");
  u64 flags = in.get<u64>("header.flags");
  if(tag != 0x6C5A7C94u || version != 1 || flags != 0)
  {
    in.error("Invalid header: tag " + std::to_string(tag) + ", version " + std::to_string(version) + ", flags " + std::to_string(flags)
             + " (expected LCP version 1)");
  }
  IntVector data = read_int_vector(in, 0, "data");
  IntVector offsets = read_int_vector(in, 64, "offsets");
  if(exact && !in.at_end()) { in.error(std::to_string(in.remaining()) + " unaccounted bytes after offsets"); }
  if(data.width > 8) { in.error("LCP values wider than 8 bits are not supported"); }
  if(offsets.size() < 2 || offsets.size() - 1 > u64(MAX_LCP_LEVELS)) { in.error("offsets: number of levels out of range"); }
  v.lcp_levels = offsets.size() - 1;
  if(offsets.word(0) != 0 || offsets.word(1) != v.lcp_size || offsets.word(v.lcp_levels) != data.size()) { in.error("offsets do not match header.size / data"); }

  st.blobs.emplace_back(); offsets.copy_words(st.blobs.back());
  const u64* offsets_ptr = st.blobs.back().data();
  st.blobs.emplace_back((data.size() + 7) / 8 + 2, 0);
  uint8_t* bytes = reinterpret_cast<uint8_t*>(st.blobs.back().data());
  for(u64 i = 0; i < data.size(); i++) { bytes[i] = uint8_t(data.get(i)); }
  v.lcp_offsets = offsets_ptr; v.lcp_data = bytes;
}

void load_lcp_members(const char* path, gcsa2_view_storage& st)
{
  sdsl_file::Mapping map(path);
  sdsl_file::Cursor in(map, std::string("LCP::load(") + path + ")");
  load_lcp_members(in, st, true);
}

template<class Loader>
int parse_memory(const void* bytes, uint64_t size, uint64_t* consumed, gcsa2_view_storage** out, const char* what, Loader load)
{
  if((bytes == nullptr && size > 0) || out == nullptr) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "null argument"); }
  *out = nullptr;
  try
  {
    std::unique_ptr<gcsa2_view_storage> st(new gcsa2_view_storage());
    std::memset(&st->view, 0, sizeof(st->view));
    st->blobs.reserve(64);
    sdsl_file::Cursor in(bytes, size, what);          // an empty stream fails as "truncated while reading header.tag"
    load(in, *st, consumed == nullptr);
    if(consumed != nullptr) { *consumed = in.consumed(); }
    *out = st.release();
    return GCSA2_OK;
  }
  catch(const sdsl_file::FormatError& e) { return fail(GCSA2_ERR_INVALID_ARGUMENT, e.what()); }
  catch(const std::exception& e) { return fail(GCSA2_ERR_OUT_OF_MEMORY, std::string(what) + ": " + e.what()); }
}

} // namespace

extern "C" int gcsa2_host_view_parse_gcsa(const void* bytes, uint64_t size, uint64_t* consumed, gcsa2_view_storage** out)
{
  return parse_memory(bytes, size, consumed, out, "GCSA::load()",
                      [](sdsl_file::Cursor& in, gcsa2_view_storage& st, bool exact) { load_gcsa_members(in, st, exact); });
}

extern "C" int gcsa2_host_view_parse_lcp(const void* bytes, uint64_t size, uint64_t* consumed, gcsa2_view_storage** out)
{
  return parse_memory(bytes, size, consumed, out, "LCP::load()",
                      [](sdsl_file::Cursor& in, gcsa2_view_storage& st, bool exact) { load_lcp_members(in, st, exact); });
}

// GCSA::serialize (src/gcsa.cpp:140-179): header, alphabet, fast_bwt + fast_rank, sparse_bwt + sparse_rank, edges + edge_rank,
// sampled_paths + rank, stored_samples, samples + select, extra_pointers, redundant_pointers.  The rank supports of
// bit_vector_il and sd_vector, and the select support of sd_vector, serialize to nothing.
extern "C" int gcsa2_host_view_serialize_gcsa(const gcsa2_host_view* v, gcsa2_sink sink, void* ctx, uint64_t* written)
{
  if(v == nullptr || sink == nullptr) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "null argument"); }
  if(v->sampled_path_bits == nullptr || v->extra_filter_bits == nullptr)
  {
    return fail(GCSA2_ERR_MISSING_COMPONENT, "GCSA::serialize() needs the samples and the counters (a .gcsa file always holds them)");
  }
  try
  {
    using namespace sdsl_file;
    Writer out(sink, ctx);
    out.put<u32>(0x6C5A6C5Au); out.put<u32>(3);                       // GCSAHeader (files.cpp:513-525)
    out.put<u64>(v->path_nodes); out.put<u64>(v->edges); out.put<u64>(v->order); out.put<u64>(0);
    // Alphabet (support.cpp:228-241): char2comp, comp2char, C, sigma, fast_chars.  comp2char travels in the view when
    // it came from a file; otherwise it is derived by the one rule the facade's Alphabet uses too.
    out.int_vector8(v->char2comp, 256, false);
    std::vector<u8> comp2char(v->sigma, 0);
    if(v->comp2char != nullptr) { std::memcpy(comp2char.data(), v->comp2char, v->sigma); }      // as loaded (or as the caller's Alphabet has it)
    else { gcsa2_derive_comp2char(v->char2comp, v->sigma, comp2char.data()); }
    out.int_vector8(comp2char.data(), v->sigma, false);
    out.int_vector64(v->C, v->sigma + 1);
    out.put<u64>(v->sigma); out.put<u64>(v->fast_chars);
    for(u64 c = 0; c < v->sigma; c++)
    {
      if(c > 0 && c <= v->fast_chars) { write_bit_vector_il(out, v->bwt[c], v->path_nodes); } else { write_empty_bit_vector_il(out); }
    }
    for(u64 c = 0; c < v->sigma; c++)
    {
      if(c > 0 && c <= v->fast_chars) { write_empty_sd_vector(out); } else { write_sd_vector(out, v->bwt[c], v->path_nodes); }
    }
    write_bit_vector_il(out, v->edge_bits, v->edges);
    write_bit_vector_il(out, v->sampled_path_bits, v->path_nodes);
    out.int_vector0_words(v->stored_samples, v->sample_count, u8(v->sample_width));
    out.bit_vector(v->sample_bits, v->sample_count);
    write_select_mcl(out, v->sample_bits, v->sample_count, true);
    write_sd_vector(out, v->extra_filter_bits, v->path_nodes);        // SadaSparse (support.cpp:493-503)
    write_sd_vector(out, v->extra_values_bits, v->extra_values_len);
    out.bit_vector(v->redundant_bits, v->redundant_len);              // SadaCount (support.cpp:401-408)
    write_select_mcl(out, v->redundant_bits, v->redundant_len, true);
    if(written != nullptr) { *written = out.written(); }
    return GCSA2_OK;
  }
  catch(const std::exception& e) { return fail(GCSA2_ERR_OUT_OF_MEMORY, std::string("GCSA::serialize(): ") + e.what()); }
}

// LCPArray::serialize (src/lcp.cpp:116-128): header, data (bit-compressed int_vector<0>, lcp.cpp:258), offsets.
extern "C" int gcsa2_host_view_serialize_lcp(const gcsa2_host_view* v, gcsa2_sink sink, void* ctx, uint64_t* written)
{
  if(v == nullptr || sink == nullptr) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "null argument"); }
  if(v->lcp_data == nullptr) { return fail(GCSA2_ERR_MISSING_COMPONENT, "the view holds no LCP array"); }
  try
  {
    using namespace sdsl_file;
    Writer out(sink, ctx);
    out.put<u32>(0x6C5A7C94u); out.put<u32>(1);                       // LCPHeader (files.cpp:581-593)
    out.put<u64>(v->lcp_size); out.put<u64>(v->lcp_branching); out.put<u64>(0);
    const u64 values = v->lcp_offsets[v->lcp_levels];
    u8 top = 0;
    for(u64 i = 0; i < values; i++) { top = (v->lcp_data[i] > top ? v->lcp_data[i] : top); }
    const u8 width = u8(bits_hi(top) + 1);                            // sdsl::util::bit_compress
    if(width == 8) { out.int_vector8(v->lcp_data, values, true); }
    else
    {
      std::vector<u64> wide(values);
      for(u64 i = 0; i < values; i++) { wide[i] = v->lcp_data[i]; }
      out.int_vector0(wide, width);
    }
    out.int_vector64(v->lcp_offsets, v->lcp_levels + 1);
    if(written != nullptr) { *written = out.written(); }
    return GCSA2_OK;
  }
  catch(const std::exception& e) { return fail(GCSA2_ERR_OUT_OF_MEMORY, std::string("LCP::serialize(): ") + e.what()); }
}

// An image that holds only the LCP array: what a default-constructed gcsa::LCPArray becomes after load()
// (src/lcp.cpp:130-143).  parent / depth / psv / nsv / rmq work on it; everything that needs the GCSA part fails
// with GCSA2_ERR_MISSING_COMPONENT or returns empty results.
extern "C" int gcsa2_lcp_create(const gcsa2_host_view* v, int device, gcsa2_index** out)
{
  if(v == nullptr || out == nullptr) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "null argument"); }
  *out = nullptr;
  if(v->lcp_data == nullptr || v->lcp_offsets == nullptr) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "the view holds no LCP array"); }
  static const uint8_t c2c[256] = {0};
  static const uint64_t C[2] = {0, 0}, none[2] = {0, 0};
  static const uint64_t* const bwt[1] = { none };
  gcsa2_host_view lcp_only;
  std::memset(&lcp_only, 0, sizeof(lcp_only));
  lcp_only.sigma = 1; lcp_only.char2comp = c2c; lcp_only.C = C; lcp_only.bwt = bwt; lcp_only.edge_bits = none;
  lcp_only.lcp_size = v->lcp_size; lcp_only.lcp_branching = v->lcp_branching; lcp_only.lcp_levels = v->lcp_levels;
  lcp_only.lcp_offsets = v->lcp_offsets; lcp_only.lcp_data = v->lcp_data;
  return gcsa2_index_create(&lcp_only, device, out);
}

extern "C" int gcsa2_host_view_load_gcsa(const char* gcsa_path, const char* lcp_path, gcsa2_view_storage** out)
{
  if(gcsa_path == nullptr || out == nullptr) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "null argument"); }
  *out = nullptr;
  try
  {
    std::unique_ptr<gcsa2_view_storage> st(new gcsa2_view_storage());
    std::memset(&st->view, 0, sizeof(st->view));
    st->blobs.reserve(64);      // pointers into the blobs' heap buffers stay valid either way; avoids reallocation churn
    load_gcsa_members(gcsa_path, *st);
    if(lcp_path != nullptr) { load_lcp_members(lcp_path, *st); }
    *out = st.release();
    return GCSA2_OK;
  }
  catch(const sdsl_file::FormatError& e) { return fail(GCSA2_ERR_INVALID_ARGUMENT, e.what()); }
  catch(const std::exception& e) { return fail(GCSA2_ERR_OUT_OF_MEMORY, std::string("gcsa2_host_view_load_gcsa: ") + e.what()); }
}

extern "C" int gcsa2_index_create_from_gcsa(const char* gcsa_path, const char* lcp_path, int device, gcsa2_index** out)
{
  gcsa2_view_storage* st = nullptr;
  int rc = gcsa2_host_view_load_gcsa(gcsa_path, lcp_path, &st);
  if(rc != GCSA2_OK) { return rc; }
  rc = gcsa2_index_create(&st->view, device, out);
  gcsa2_host_view_free(st);
  return rc;
}
