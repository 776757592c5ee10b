// <gcsa/utils.h> of the MI355X engine -- the basic types of jltsiren/gcsa2's public interface
// (reference include/gcsa/utils.h), written from scratch over nothing but the standard library so that
// code which includes <gcsa/gcsa.h> / <gcsa/lcp.h> and uses the query interface compiles unchanged
// against this engine.  Only what the query path and its callers use is here: the construction-time
// helpers of the reference's utils.h (TempFile, readRows, parallel sorts, ...) are out of scope.
#ifndef GCSA2_HIP_GCSA_UTILS_H
#define GCSA2_HIP_GCSA_UTILS_H

#include "../gcsa2_hip.h"

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

// serialize() keeps the reference's signature, whose second argument is SDSL's structure tree; only the
// pointer type is needed (the engine ignores it), and this declaration is compatible with the real one.
namespace sdsl { class structure_tree_node; }

namespace gcsa
{

typedef std::uint64_t size_type;                          // utils.h:55
typedef std::uint8_t  char_type;                          // utils.h:59
typedef std::uint8_t  comp_type;                          // utils.h:60
typedef std::uint8_t  byte_type;                          // utils.h:61
typedef std::pair<size_type, size_type> range_type;       // utils.h:84

constexpr size_type WORD_BITS = 64;
constexpr size_type BYTE_BITS = 8;
constexpr size_type KILOBYTE = 1024, MEGABYTE = KILOBYTE * 1024, GIGABYTE = MEGABYTE * 1024;
constexpr double MILLION_DOUBLE = 1000000.0;

inline double inMegabytes(size_type bytes) { return bytes / double(MEGABYTE); }
inline double inGigabytes(size_type bytes) { return bytes / double(GIGABYTE); }

struct Range   // utils.h:86-117
{
  static size_type length(range_type range) { return range.second + 1 - range.first; }
  static bool empty(range_type range) { return (range.first + 1 > range.second + 1); }
  static bool empty(size_type sp, size_type ep) { return (sp + 1 > ep + 1); }
  static size_type bound(size_type value, range_type bounds) { return bound(value, bounds.first, bounds.second); }
  static size_type bound(size_type value, size_type low, size_type high) { return std::max(std::min(value, high), low); }
  static range_type empty_range() { return range_type(1, 0); }
};

struct Version   // utils.h:156-167
{
  constexpr static size_type MAJOR_VERSION = 1, MINOR_VERSION = 3, PATCH_VERSION = 0;
  constexpr static size_type GCSA_VERSION = 3, LCP_VERSION = 1;
  static std::string str(bool verbose = false)
  {
    std::string v = "v" + std::to_string(MAJOR_VERSION) + "." + std::to_string(MINOR_VERSION) + "." + std::to_string(PATCH_VERSION);
    return verbose ? "GCSA2 " + v + " query interface on the MI355X engine (file format GCSA v3, LCP v1)" : v;
  }
};

inline double readTimer()    // utils.cpp:131-135 (wall clock, seconds)
{
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

template<class Element>
void removeDuplicates(std::vector<Element>& vec, bool /* parallel */ = false)   // utils.h:350-357
{
  std::sort(vec.begin(), vec.end());
  vec.resize(size_type(std::unique(vec.begin(), vec.end()) - vec.begin()));
}

// The engine reports failures as status codes; this layer turns them into exceptions, as the reference's
// load() does (src/gcsa.cpp:188-193).  Query methods of the reference never fail.
inline void check(int status, const char* what)
{
  if(status != GCSA2_OK) { throw std::runtime_error(std::string(what) + ": " + gcsa2_last_error()); }
}

// HIP device new indexes are placed on by load() and the default constructors' file helpers (the reference has
// no such notion: its indexes live in host memory).  Initial value: environment variable GCSA2_DEVICE, else 0.
struct Device
{
  static int& current()
  {
    static int device = []() { const char* e = std::getenv("GCSA2_DEVICE"); return e != nullptr ? std::atoi(e) : 0; }();
    return device;
  }
  static void set(int device) { current() = device; }
};

// The rest of a stream as bytes (load(std::istream&) parses from memory and gives unread bytes back).
inline std::vector<char> readRest(std::istream& in)
{
  std::vector<char> data;
  char buffer[1 << 16];
  while(in.read(buffer, sizeof(buffer)) || in.gcount() > 0) { data.insert(data.end(), buffer, buffer + in.gcount()); }
  return data;
}

// sdsl::load_from_file / store_to_file as the reference's tools call them (benchmark/query_gcsa.cpp:55,63): true on success.
template<class Structure>
bool load_from_file(Structure& structure, const std::string& filename)
{
  std::ifstream in(filename.c_str(), std::ios_base::binary);
  if(!in) { return false; }
  structure.load(in);
  return true;
}

template<class Structure>
bool store_to_file(const Structure& structure, const std::string& filename)
{
  std::ofstream out(filename.c_str(), std::ios_base::binary | std::ios_base::trunc);
  if(!out) { return false; }
  structure.serialize(out);
  return bool(out);
}

} // namespace gcsa

// Opt-in: callers that spell the two file helpers with the sdsl:: prefix and do not link SDSL.
#ifdef GCSA2_HIP_SDSL_IO
namespace sdsl
{
template<class Structure> bool load_from_file(Structure& s, const std::string& f) { return gcsa::load_from_file(s, f); }
template<class Structure> bool store_to_file(const Structure& s, const std::string& f) { return gcsa::store_to_file(s, f); }
}
#endif

#endif // GCSA2_HIP_GCSA_UTILS_H
