// <gcsa/files.h> of the MI355X engine: the two file headers callers read through `index.header` /
// `lcp.header` (reference include/gcsa/files.h:135-190, src/files.cpp:513-603).  The k-mer / graph readers of
// the reference's files.h belong to index construction and are out of scope.
#ifndef GCSA2_HIP_GCSA_FILES_H
#define GCSA2_HIP_GCSA_FILES_H

#include "utils.h"

namespace gcsa
{

struct GCSAHeader   // files.h:135-156: 40 bytes on disk
{
  std::uint32_t tag, version;
  std::uint64_t path_nodes, edges, order, flags;

  constexpr static std::uint32_t TAG = 0x6C5A6C5A;
  constexpr static std::uint32_t VERSION = Version::GCSA_VERSION;
  constexpr static std::uint32_t MIN_VERSION = 1;

  GCSAHeader() : tag(TAG), version(VERSION), path_nodes(0), edges(0), order(0), flags(0) {}
  bool check(std::uint32_t expected_version = VERSION) const { return tag == TAG && version == expected_version && flags == 0; }   // files.cpp:527-531
  bool checkNew() const { return tag == TAG && version > VERSION; }
  void swap(GCSAHeader& another) { std::swap(*this, another); }
};

inline std::ostream& operator<<(std::ostream& stream, const GCSAHeader& header)   // files.cpp:545-551
{
  return stream << "GCSA header version " << header.version << ": " << header.path_nodes << " path nodes, " << header.edges
                << " edges, order " << header.order;
}

struct LCPHeader   // files.h:169-190: 32 bytes on disk
{
  std::uint32_t tag, version;
  std::uint64_t size, branching, flags;

  constexpr static std::uint32_t TAG = 0x6C5A7C94;
  constexpr static std::uint32_t VERSION = Version::LCP_VERSION;
  constexpr static std::uint32_t MIN_VERSION = 1;

  LCPHeader() : tag(TAG), version(VERSION), size(0), branching(64), flags(0) {}
  bool check(std::uint32_t expected_version = VERSION) const { return tag == TAG && version == expected_version && flags == 0; }   // files.cpp:595-599
  bool checkNew() const { return tag == TAG && version > VERSION; }
  void swap(LCPHeader& another) { std::swap(*this, another); }
};

inline std::ostream& operator<<(std::ostream& stream, const LCPHeader& header)   // files.cpp:611-616
{
  return stream << "LCP header version " << header.version << ": array size " << header.size << ", branching factor " << header.branching;
}

} // namespace gcsa

#endif // GCSA2_HIP_GCSA_FILES_H
