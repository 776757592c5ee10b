// comm.hpp -- the one collective of the query path: a gather of hit ranges on a root GPU over RCCL / xGMI.
// Part of the single translation unit gcsa2_hip.hip.
//
// The path shards with no data-path exchange (queries are independent, the index is replicated; the
// reference's only data-parallel query path is the static split of verifyIndex, src/algorithms.cpp:106-114),
// so the only inter-GPU traffic is the final gather of (sp, ep) pairs.  It is a true gather, not a ring:
// every peer owns a direct xGMI link into the root, so grouped ncclSend / ncclRecv pairs use all links at
// once (a ring all-gather would be bound by one link and move G times the data).
//
// RCCL is bound at run time (dlopen) so that the library loads, and every single-GPU entry point works,
// on hosts without RCCL; the comm entry points fail loudly with GCSA2_ERR_MISSING_COMPONENT there.
// GCSA2_RCCL_LIB names the library to use (the Python binding points it at the RCCL that torch loaded,
// so a process has one RCCL); default librccl.so.1, then librccl.so.
#pragma once

#include <rccl/rccl.h>
#include <dlfcn.h>

#include <mutex>

namespace {

struct RcclApi
{
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
  bool ok = false;
};

RcclApi& rccl()
{
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, []()
  {
    const char* env = std::getenv("GCSA2_RCCL_LIB");
    const char* names[3] = { env, "librccl.so.1", "librccl.so" };
    for(const char* name : names)
    {
      if(name == nullptr || *name == 0) { continue; }
      api.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if(api.handle != nullptr) { break; }
      api.error = dlerror();
    }
    if(api.handle == nullptr) { api.error = "RCCL not loadable: " + api.error; return; }
    bool all = true;
    auto bind = [&](auto& fn, const char* symbol)
    {
      fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(api.handle, symbol));
      if(fn == nullptr) { all = false; api.error = std::string("RCCL symbol missing: ") + symbol; }
    };
    bind(api.GetUniqueId, "ncclGetUniqueId"); bind(api.CommInitRank, "ncclCommInitRank");
    bind(api.CommInitAll, "ncclCommInitAll"); bind(api.CommDestroy, "ncclCommDestroy");
    bind(api.CommCount, "ncclCommCount");
    bind(api.GroupStart, "ncclGroupStart"); bind(api.GroupEnd, "ncclGroupEnd");
    bind(api.Send, "ncclSend"); bind(api.Recv, "ncclRecv"); bind(api.GetErrorString, "ncclGetErrorString");
    api.ok = all;
  });
  return api;
}

#define RCCL_TRY(expr) do { ncclResult_t r_ = (expr); if(r_ != ncclSuccess) { \
  return fail(GCSA2_ERR_HIP, std::string(#expr) + ": " + rccl().GetErrorString(r_)); } } while(0)

// (sp, ep) u64 pairs <-> (sp, ep + 1 - sp) u32 pairs: exact whenever every path node and edge number of the
// index is below 2^32 (an empty range is (x, x - 1), utils.h:93-96, so its length 0 restores ep = sp - 1 even
// when that wraps).  Halves the bytes the gather moves over xGMI.
__global__ __launch_bounds__(TPB) void k_pack_ranges32(const u64* __restrict__ in, u64 nq, uint2* __restrict__ out)
{
  const u64 q = u64(blockIdx.x) * TPB + threadIdx.x;
  if(q >= nq) { return; }
  const ulonglong2 r = reinterpret_cast<const ulonglong2*>(in)[q];
  out[q] = make_uint2(u32(r.x), u32(r.y + 1 - r.x));
}

__global__ __launch_bounds__(TPB) void k_unpack_ranges32(const uint2* __restrict__ in, u64 nq, u64* __restrict__ out)
{
  const u64 q = u64(blockIdx.x) * TPB + threadIdx.x;
  if(q >= nq) { return; }
  const uint2 r = in[q];
  reinterpret_cast<ulonglong2*>(out)[q] = make_ulonglong2(u64(r.x), u64(r.x) + u64(r.y) - 1);
}

// The same for indexes below 2^40 path nodes and edges (the 5.7 G-node index of BASELINE configs[3]): 10 bytes per range, five
// u16 -- sp bits 0..31, length bits 0..31, then the two high bytes -- instead of 16.
__global__ __launch_bounds__(TPB) void k_pack_ranges40(const u64* __restrict__ in, u64 nq, unsigned short* __restrict__ out)
{
  const u64 q = u64(blockIdx.x) * TPB + threadIdx.x;
  if(q >= nq) { return; }
  const ulonglong2 r = reinterpret_cast<const ulonglong2*>(in)[q];
  const u64 sp = r.x, len = r.y + 1 - r.x;
  unsigned short* o = out + 5 * q;
  o[0] = (unsigned short)sp; o[1] = (unsigned short)(sp >> 16); o[2] = (unsigned short)len; o[3] = (unsigned short)(len >> 16);
  o[4] = (unsigned short)(((sp >> 32) & 0xFF) | (((len >> 32) & 0xFF) << 8));
}

__global__ __launch_bounds__(TPB) void k_unpack_ranges40(const unsigned short* __restrict__ in, u64 nq, u64* __restrict__ out)
{
  const u64 q = u64(blockIdx.x) * TPB + threadIdx.x;
  if(q >= nq) { return; }
  const unsigned short* o = in + 5 * q;
  const u64 sp = u64(o[0]) | (u64(o[1]) << 16) | (u64(o[4] & 0xFF) << 32);
  const u64 len = u64(o[2]) | (u64(o[3]) << 16) | (u64(o[4] >> 8) << 32);
  reinterpret_cast<ulonglong2*>(out)[q] = make_ulonglong2(sp, sp + len - 1);
}

// Six bytes per range for the common case (round 6): sp in 40 bits and the length ep + 1 - sp in ONE byte -- 0 .. 254 as it is,
// 255 = "in the overflow list" -- followed, behind the shard's ranges, by a count and a list of (query, length) pairs for the
// ranges of 255 and more path nodes.  The list has a fixed capacity, so a shard's block has a size every rank knows before
// anything is searched (gcsa2_wire48_bytes): the gather stays one grouped send / recv of fixed sizes.  A 32-mer batch on a
// whole-genome index is nearly all lengths 0 and 1: 75 MB per peer and step at N = 8 instead of the 125 MB of the 40-bit pairs.
// A batch with more long ranges than the list holds is REPORTED (the count behind the ranges exceeds the capacity), never
// truncated silently: the caller falls back to the 40-bit pairs.
// block: [6 n bytes, padded to 16] [u64 count, u64 0] [capacity x (u64 query, u64 length)]
__host__ __device__ inline u64 wire48_list_offset(u64 nq) { return (6 * nq + 15) & ~u64(15); }

__global__ __launch_bounds__(TPB) void k_pack_ranges48(const u64* __restrict__ in, u64 nq, unsigned char* __restrict__ out, u64 capacity)
{
  const u64 q = u64(blockIdx.x) * TPB + threadIdx.x;
  const u32 lane = threadIdx.x & 63;
  bool longer = false;
  u64 len = 0;
  if(q < nq)
  {
    const ulonglong2 r = reinterpret_cast<const ulonglong2*>(in)[q];
    const u64 sp = r.x;
    len = r.y + 1 - r.x;
    longer = (len >= 255);
    unsigned short* o = reinterpret_cast<unsigned short*>(out + 6 * q);
    o[0] = (unsigned short)sp; o[1] = (unsigned short)(sp >> 16);
    o[2] = (unsigned short)(((sp >> 32) & 0xFF) | ((longer ? u64(255) : len) << 8));
  }
  const u64 mask = __ballot(longer);
  if(mask == 0) { return; }                                   // (uniform)
  unsigned long long* list = reinterpret_cast<unsigned long long*>(out + wire48_list_offset(nq));
  unsigned long long base = 0;
  const u32 leader = u32(__ffsll((long long)mask)) - 1;
  if(lane == leader) { base = atomicAdd(list, (unsigned long long)__popcll(mask)); }
  base = __shfl(base, leader, 64);
  if(longer)
  {
    const u64 slot = base + __popcll(mask & ((u64(1) << lane) - 1));
    if(slot < capacity) { list[2 + 2 * slot] = q; list[3 + 2 * slot] = len; }
  }
}

__global__ __launch_bounds__(TPB) void k_unpack_ranges48(const unsigned char* __restrict__ in, u64 nq, u64* __restrict__ out)
{
  const u64 q = u64(blockIdx.x) * TPB + threadIdx.x;
  if(q >= nq) { return; }
  const unsigned short* o = reinterpret_cast<const unsigned short*>(in + 6 * q);
  const u64 sp = u64(o[0]) | (u64(o[1]) << 16) | (u64(o[2] & 0xFF) << 32);
  const u64 len = u64(o[2] >> 8);                              // (255: the overflow pass writes the upper end)
  reinterpret_cast<ulonglong2*>(out)[q] = make_ulonglong2(sp, sp + len - 1);
}

// the ranges of the overflow list: the length is there, the lower end is what k_unpack_ranges48 wrote (stream order)
__global__ __launch_bounds__(TPB) void k_unpack_overflow48(const unsigned char* __restrict__ in, u64 nq, u64 capacity, u64* __restrict__ out,
                                                           u64* __restrict__ count_out)
{
  const unsigned long long* list = reinterpret_cast<const unsigned long long*>(in + wire48_list_offset(nq));
  const u64 count = list[0], i = u64(blockIdx.x) * TPB + threadIdx.x;
  if(i == 0 && count_out != nullptr) { *count_out = count; }
  if(i >= count || i >= capacity) { return; }
  const u64 q = list[2 + 2 * i], len = list[3 + 2 * i];
  if(q < nq) { out[2 * q + 1] = out[2 * q] + len - 1; }
}

}  // namespace

// One rank of the query path's communicator: an RCCL communicator (gcsa2_comm_create), or the application's own transport
// (gcsa2_comm_create_custom: MPI, a host-memory gather, ...) behind the same gather contract.
struct gcsa2_comm
{
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  gcsa2_gather_fn custom = nullptr;
  void* user = nullptr;
};

namespace {

// the gather of the query path: rank r contributes bytes[r] bytes; the root receives them back to back in rank order.
// RCCL: grouped send / recv, enqueued on `st`.  Custom transport: the application's function, same contract.
int gather_bytes(const gcsa2_comm* c, const void* d_send, const u64* bytes, void* d_recv, int root, hipStream_t st)
{
  const int rank = c->rank, world = c->world;
  if(c->custom != nullptr)
  {
    const int rc = c->custom(c->user, d_send, bytes, d_recv, root, static_cast<void*>(st));
    if(rc != 0) { return fail(GCSA2_ERR_HIP, "the communicator's gather function failed with " + std::to_string(rc)); }
    return GCSA2_OK;
  }
  ncclComm_t comm = c->comm;
  RcclApi& api = rccl();
  RCCL_TRY(api.GroupStart());
  ncclResult_t r = ncclSuccess;
  if(rank == root)
  {
    u64 at = 0;
    for(int p = 0; p < world && r == ncclSuccess; p++)
    {
      if(p != root && bytes[p] > 0) { r = api.Recv(static_cast<char*>(d_recv) + at, bytes[p], ncclUint8, p, comm, st); }
      at += bytes[p];
    }
  }
  else if(bytes[rank] > 0) { r = api.Send(d_send, bytes[rank], ncclUint8, root, comm, st); }
  ncclResult_t g = api.GroupEnd();
  if(r != ncclSuccess) { return fail(GCSA2_ERR_HIP, std::string("ncclSend / ncclRecv: ") + api.GetErrorString(r)); }
  if(g != ncclSuccess) { return fail(GCSA2_ERR_HIP, std::string("ncclGroupEnd: ") + api.GetErrorString(g)); }
  if(rank == root && bytes[root] > 0)       // the root's own shard: a device-to-device copy on the same stream
  {
    u64 at = 0;
    for(int p = 0; p < root; p++) { at += bytes[p]; }
    if(static_cast<char*>(d_recv) + at != d_send)
    {
      HIP_TRY(hipMemcpyAsync(static_cast<char*>(d_recv) + at, d_send, bytes[root], hipMemcpyDeviceToDevice, st));
    }
  }
  return GCSA2_OK;
}

}  // namespace

extern "C" {

int gcsa2_comm_unique_id(uint8_t* id)
{
  if(id == nullptr) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "null id buffer"); }
  RcclApi& api = rccl();
  if(!api.ok) { return fail(GCSA2_ERR_MISSING_COMPONENT, api.error); }
  static_assert(GCSA2_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");
  ncclUniqueId uid;
  RCCL_TRY(api.GetUniqueId(&uid));
  std::memcpy(id, uid.internal, NCCL_UNIQUE_ID_BYTES);
  return GCSA2_OK;
}

int gcsa2_comm_create(const uint8_t* id, int rank, int world, int device, gcsa2_comm** out)
{
  if(id == nullptr || out == nullptr || world <= 0 || rank < 0 || rank >= world) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "bad communicator arguments"); }
  *out = nullptr;
  RcclApi& api = rccl();
  if(!api.ok) { return fail(GCSA2_ERR_MISSING_COMPONENT, api.error); }
  DeviceGuard guard(device);
  if(!guard.ok) { return fail(GCSA2_ERR_HIP, "hipSetDevice failed"); }
  gcsa2_comm* c = new(std::nothrow) gcsa2_comm();
  if(c == nullptr) { return fail(GCSA2_ERR_OUT_OF_MEMORY, "host allocation failed"); }
  c->rank = rank; c->world = world; c->device = device;
  ncclUniqueId uid;
  std::memcpy(uid.internal, id, NCCL_UNIQUE_ID_BYTES);
  ncclResult_t r = api.CommInitRank(&c->comm, world, uid, rank);
  if(r != ncclSuccess) { delete c; return fail(GCSA2_ERR_HIP, std::string("ncclCommInitRank: ") + api.GetErrorString(r)); }
  *out = c;
  return GCSA2_OK;
}

int gcsa2_comm_create_custom(int rank, int world, int device, gcsa2_gather_fn gather, void* user, gcsa2_comm** out)
{
  if(gather == nullptr || out == nullptr || world <= 0 || rank < 0 || rank >= world) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "bad communicator arguments"); }
  *out = nullptr;
  DeviceGuard guard(device);
  if(!guard.ok) { return fail(GCSA2_ERR_HIP, "hipSetDevice failed"); }
  gcsa2_comm* c = new(std::nothrow) gcsa2_comm();
  if(c == nullptr) { return fail(GCSA2_ERR_OUT_OF_MEMORY, "host allocation failed"); }
  c->rank = rank; c->world = world; c->device = device; c->custom = gather; c->user = user;
  *out = c;
  return GCSA2_OK;
}

void gcsa2_comm_destroy(gcsa2_comm* c)
{
  if(c == nullptr) { return; }
  if(c->comm != nullptr && rccl().ok) { DeviceGuard guard(c->device); (void)rccl().CommDestroy(c->comm); }
  delete c;
}

int gcsa2_comm_rank(const gcsa2_comm* c) { return c == nullptr ? -1 : c->rank; }
int gcsa2_comm_world(const gcsa2_comm* c) { return c == nullptr ? 0 : c->world; }

int gcsa2_comm_rccl_ranks(const gcsa2_comm* c, int* ranks)
{
  if(c == nullptr || ranks == nullptr) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "bad communicator"); }
  if(c->custom != nullptr) { *ranks = 0; return GCSA2_OK; }            // not an RCCL communicator
  if(c->comm == nullptr) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "bad communicator"); }
  RcclApi& api = rccl();
  if(!api.ok) { return fail(GCSA2_ERR_MISSING_COMPONENT, api.error); }
  RCCL_TRY(api.CommCount(c->comm, ranks));
  return GCSA2_OK;
}

int gcsa2_comm_gather(gcsa2_comm* c, const void* d_send, const uint64_t* bytes, void* d_recv, int root, void* stream)
{
  if(c == nullptr || bytes == nullptr || root < 0 || root >= c->world) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "bad gather arguments"); }
  if(c->rank == root && d_recv == nullptr) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "the root needs a receive buffer"); }
  DeviceGuard guard(c->device);
  return gather_bytes(c, d_send, bytes, d_recv, root, static_cast<hipStream_t>(stream));
}

int gcsa2_pack_ranges32_device(const uint64_t* d_ranges, uint64_t nq, uint32_t* d_packed, void* stream)
{
  if(nq == 0) { return GCSA2_OK; }
  hipLaunchKernelGGL(k_pack_ranges32, dim3(grid_for(nq)), dim3(TPB), 0, static_cast<hipStream_t>(stream), d_ranges, nq, reinterpret_cast<uint2*>(d_packed));
  LAUNCH_CHECK("k_pack_ranges32");
  return GCSA2_OK;
}

int gcsa2_unpack_ranges32_device(const uint32_t* d_packed, uint64_t nq, uint64_t* d_ranges, void* stream)
{
  if(nq == 0) { return GCSA2_OK; }
  hipLaunchKernelGGL(k_unpack_ranges32, dim3(grid_for(nq)), dim3(TPB), 0, static_cast<hipStream_t>(stream), reinterpret_cast<const uint2*>(d_packed), nq, d_ranges);
  LAUNCH_CHECK("k_unpack_ranges32");
  return GCSA2_OK;
}

int gcsa2_pack_ranges40_device(const uint64_t* d_ranges, uint64_t nq, void* d_packed, void* stream)
{
  if(nq == 0) { return GCSA2_OK; }
  hipLaunchKernelGGL(k_pack_ranges40, dim3(grid_for(nq)), dim3(TPB), 0, static_cast<hipStream_t>(stream), d_ranges, nq, static_cast<unsigned short*>(d_packed));
  LAUNCH_CHECK("k_pack_ranges40");
  return GCSA2_OK;
}

int gcsa2_unpack_ranges40_device(const void* d_packed, uint64_t nq, uint64_t* d_ranges, void* stream)
{
  if(nq == 0) { return GCSA2_OK; }
  hipLaunchKernelGGL(k_unpack_ranges40, dim3(grid_for(nq)), dim3(TPB), 0, static_cast<hipStream_t>(stream), static_cast<const unsigned short*>(d_packed), nq, d_ranges);
  LAUNCH_CHECK("k_unpack_ranges40");
  return GCSA2_OK;
}

uint64_t gcsa2_wire48_bytes(uint64_t nq, uint64_t capacity) { return wire48_list_offset(nq) + 16 + 16 * capacity; }

int gcsa2_pack_ranges48_device(const uint64_t* d_ranges, uint64_t nq, void* d_packed, uint64_t capacity, void* stream)
{
  if(d_packed == nullptr) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "null buffer"); }
  hipStream_t st = static_cast<hipStream_t>(stream);
  HIP_TRY(hipMemsetAsync(static_cast<unsigned char*>(d_packed) + wire48_list_offset(nq), 0, 16, st));     // the list's count
  if(nq == 0) { return GCSA2_OK; }
  hipLaunchKernelGGL(k_pack_ranges48, dim3(grid_for(nq)), dim3(TPB), 0, st, d_ranges, nq, static_cast<unsigned char*>(d_packed), capacity);
  LAUNCH_CHECK("k_pack_ranges48");
  return GCSA2_OK;
}

int gcsa2_unpack_ranges48_device(const void* d_packed, uint64_t nq, uint64_t capacity, uint64_t* d_ranges, uint64_t* d_overflow_count, void* stream)
{
  if(d_packed == nullptr) { return fail(GCSA2_ERR_INVALID_ARGUMENT, "null buffer"); }
  hipStream_t st = static_cast<hipStream_t>(stream);
  if(nq > 0)
  {
    hipLaunchKernelGGL(k_unpack_ranges48, dim3(grid_for(nq)), dim3(TPB), 0, st, static_cast<const unsigned char*>(d_packed), nq, d_ranges);
    LAUNCH_CHECK("k_unpack_ranges48");
  }
  hipLaunchKernelGGL(k_unpack_overflow48, dim3(grid_for(capacity > 0 ? capacity : 1)), dim3(TPB), 0, st, static_cast<const unsigned char*>(d_packed), nq, capacity, d_ranges, d_overflow_count);
  LAUNCH_CHECK("k_unpack_overflow48");
  return GCSA2_OK;
}

}  // extern "C"
