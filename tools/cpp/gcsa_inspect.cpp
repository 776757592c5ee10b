// gcsa_inspect -- where every member of a .gcsa / .lcp file lies, as this engine's reader sees it.
//
//   gcsa_inspect graph.gcsa [graph.lcp]
//
// Prints, member by member in the order GCSA::serialize / LCPArray::serialize write them (reference src/gcsa.cpp:140-179,
// src/lcp.cpp:116-128; headers src/files.cpp:513-537, 581-603; Alphabet src/support.cpp:228-250; SadaSparse / SadaCount
// src/support.cpp:492-516, 400-418), the byte offset, the byte size and the shape the SDSL container encoding implies
// (lengths, widths, block counts, ones).  The container encodings are restated from sdsl-lite 2.1.1 and are NOT pinned on a
// file written by the real library (none exists in the build environment): run this on such a file first.  Either the walk
// ends exactly at the end of the file -- then diff the listing against the one of `serialize()` of the same index through
// this engine -- or it stops with the member and byte offset at which the restatement and the file disagree.
// Host code only: no GPU, no library of this repository is linked.
#include <cstdio>
#include <cstdlib>
#include <string>

#include "../../gcsa2_amd/csrc/sdsl_reader.hpp"

using namespace sdsl_file;

namespace {

void row(const char* member, u64 begin, u64 end, const std::string& shape)
{
  std::printf("%-34s %14llu %14llu  %s\n", member, (unsigned long long)begin, (unsigned long long)(end - begin), shape.c_str());
}

std::string iv_shape(const IntVector& v, const char* type)
{
  return std::string(type) + ": " + std::to_string(v.bits) + " bits, width " + std::to_string(unsigned(v.width)) + ", " + std::to_string(v.size()) + " elements";
}

void int_vector(Cursor& in, uint8_t fixed, const char* member, const char* type)
{
  const u64 at = in.consumed();
  IntVector v = read_int_vector(in, fixed, member);
  row(member, at, in.consumed(), iv_shape(v, type));
}

void bit_vector_il(Cursor& in, const std::string& member)
{
  const u64 at = in.consumed();
  std::vector<u64> plain; u64 size = 0;
  read_bit_vector_il(in, plain, size, member.c_str());
  u64 ones = 0;
  for(u64 w : plain) { ones += u64(__builtin_popcountll(w)); }
  row(member.c_str(), at, in.consumed(), "bit_vector_il<512>: " + std::to_string(size) + " bits, " + std::to_string(ones) + " ones, "
      + std::to_string(size == 0 ? 0 : (size + 512) / 512) + " superblocks (counts verified)");
}

void select_mcl(Cursor& in, const std::string& member)
{
  const u64 at = in.consumed();
  Cursor probe = in;
  const u64 args = probe.get<u64>(member.c_str());
  skip_select_mcl(in, member.c_str());
  row(member.c_str(), at, in.consumed(), "select_support_mcl: " + std::to_string(args) + " arguments, " + std::to_string((args + 4095) >> 12) + " superblocks");
}

void sd_vector(Cursor& in, const std::string& member)
{
  const u64 at = in.consumed();
  std::vector<u64> plain; u64 size = 0;
  read_sd_vector(in, plain, size, member.c_str());
  u64 ones = 0;
  for(u64 w : plain) { ones += u64(__builtin_popcountll(w)); }
  row(member.c_str(), at, in.consumed(), "sd_vector: universe " + std::to_string(size) + ", " + std::to_string(ones) + " ones (low / high / two select directories)");
}

void plain_bits(Cursor& in, const char* member)
{
  const u64 at = in.consumed();
  std::vector<u64> plain; u64 size = 0;
  read_bit_vector(in, plain, size, member);
  u64 ones = 0;
  for(u64 w : plain) { ones += u64(__builtin_popcountll(w)); }
  row(member, at, in.consumed(), "bit_vector: " + std::to_string(size) + " bits, " + std::to_string(ones) + " ones");
}

int inspect_gcsa(const char* path)
{
  Mapping map(path);
  Cursor in(map, std::string("gcsa_inspect(") + path + ")");
  std::printf("# %s: %llu bytes\n%-34s %14s %14s  %s\n", path, (unsigned long long)map.bytes, "member", "offset", "bytes", "shape");
  u64 at = in.consumed();
  const uint32_t tag = in.get<uint32_t>("header.tag"), version = in.get<uint32_t>("header.version");
  const u64 path_nodes = in.get<u64>("header.path_nodes"), edges = in.get<u64>("header.edges"), order = in.get<u64>("header.order"), flags = in.get<u64>("header.flags");
  char text[200];
  std::snprintf(text, sizeof(text), "GCSAHeader: tag 0x%08X, version %u, path_nodes %llu, edges %llu, order %llu, flags %llu", tag, version,
                (unsigned long long)path_nodes, (unsigned long long)edges, (unsigned long long)order, (unsigned long long)flags);
  row("header", at, in.consumed(), text);
  if(tag != 0x6C5A6C5Au || version != 3) { in.error("not a GCSA version 3 header (a file wrapped in another container has to be unwrapped first)"); }
  int_vector(in, 8, "alpha.char2comp", "int_vector<8>");
  int_vector(in, 8, "alpha.comp2char", "int_vector<8>");
  int_vector(in, 64, "alpha.C", "int_vector<64>");
  at = in.consumed();
  const u64 sigma = in.get<u64>("alpha.sigma"), fast_chars = in.get<u64>("alpha.fast_chars");
  row("alpha.sigma, alpha.fast_chars", at, in.consumed(), "sigma " + std::to_string(sigma) + ", fast_chars " + std::to_string(fast_chars));
  if(sigma == 0 || sigma > 64) { in.error("alphabet size out of range"); }
  for(u64 c = 0; c < sigma; c++) { bit_vector_il(in, "fast_bwt[" + std::to_string(c) + "]"); }
  row("fast_rank[0..sigma)", in.consumed(), in.consumed(), "rank_support_il: nothing on disk");
  for(u64 c = 0; c < sigma; c++) { sd_vector(in, "sparse_bwt[" + std::to_string(c) + "]"); }
  row("sparse_rank[0..sigma)", in.consumed(), in.consumed(), "rank_support_sd: nothing on disk");
  bit_vector_il(in, "edges");
  bit_vector_il(in, "sampled_paths");
  int_vector(in, 0, "stored_samples", "int_vector<0>");
  plain_bits(in, "samples");
  select_mcl(in, "sample_select");
  sd_vector(in, "extra_pointers.filter");
  sd_vector(in, "extra_pointers.values");
  plain_bits(in, "redundant_pointers.data");
  select_mcl(in, "redundant_pointers.select");
  std::printf("# end of the walk at byte %llu of %llu: %s\n", (unsigned long long)in.consumed(), (unsigned long long)map.bytes,
              in.at_end() ? "every byte accounted for" : "TRAILING BYTES NOT ACCOUNTED FOR");
  return in.at_end() ? 0 : 2;
}

int inspect_lcp(const char* path)
{
  Mapping map(path);
  Cursor in(map, std::string("gcsa_inspect(") + path + ")");
  std::printf("# %s: %llu bytes\n%-34s %14s %14s  %s\n", path, (unsigned long long)map.bytes, "member", "offset", "bytes", "shape");
  u64 at = in.consumed();
  const uint32_t tag = in.get<uint32_t>("header.tag"), version = in.get<uint32_t>("header.version");
  const u64 size = in.get<u64>("header.size"), branching = in.get<u64>("header.branching"), flags = in.get<u64>("header.flags");
  char text[200];
  std::snprintf(text, sizeof(text), "LCPHeader: tag 0x%08X, version %u, size %llu, branching %llu, flags %llu", tag, version,
                (unsigned long long)size, (unsigned long long)branching, (unsigned long long)flags);
  row("header", at, in.consumed(), text);
  if(tag != 0x6C5A7C94u || version != 1) { in.error("not an LCP version 1 header"); }
  int_vector(in, 0, "data", "int_vector<0>");
  int_vector(in, 64, "offsets", "int_vector<64>");
  std::printf("# end of the walk at byte %llu of %llu: %s\n", (unsigned long long)in.consumed(), (unsigned long long)map.bytes,
              in.at_end() ? "every byte accounted for" : "TRAILING BYTES NOT ACCOUNTED FOR");
  return in.at_end() ? 0 : 2;
}

}  // namespace

int main(int argc, char** argv)
{
  if(argc < 2 || argc > 3)
  {
    std::fprintf(stderr, "usage: gcsa_inspect graph.gcsa [graph.lcp]\n");
    return EXIT_FAILURE;
  }
  int rc = 0;
  try
  {
    rc = inspect_gcsa(argv[1]);
    if(argc == 3) { const int r2 = inspect_lcp(argv[2]); rc = (rc != 0 ? rc : r2); }
  }
  catch(const FormatError& e)
  {
    std::printf("# STOPPED: %s\n", e.what());
    return 3;
  }
  return rc;
}
