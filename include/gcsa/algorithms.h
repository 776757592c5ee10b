// <gcsa/algorithms.h> of the MI355X engine: the k-mer counting algorithms over the index
// (reference include/gcsa/algorithms.h:59-92, src/algorithms.cpp:364-616) as device frontier expansions.
// verifyIndex and printStatistics need the input graph / SDSL size accounting and are out of scope.
#ifndef GCSA2_HIP_GCSA_ALGORITHMS_H
#define GCSA2_HIP_GCSA_ALGORITHMS_H

#include "gcsa.h"
#include "lcp.h"

#include <array>

namespace gcsa
{

// algorithms.h:59-84 -- k-mer counting over the index.
struct KMerSearchParameters
{
  size_type seed_length;  // kept for source compatibility; the device version needs no seeds
  bool include_Ns;        // also count k-mers containing Ns (comps fast_chars + 1 .. sigma - 2)
  bool force;             // allow k > order()
  std::string output;     // compareKMers: base name of the .left / .right dumps (algorithms.h:63-68)
  constexpr static size_type SEED_LENGTH = 5;
  KMerSearchParameters() : seed_length(SEED_LENGTH), include_Ns(false), force(false), output() {}
};

inline size_type countKMers(const GCSA& index, size_type k, const KMerSearchParameters& parameters = KMerSearchParameters())
{
  size_type result = 0;
  check(gcsa2_count_kmers(index.handle, k, parameters.include_Ns ? 1 : 0, parameters.force ? 1 : 0, &result), "countKMers()");
  return result;
}

// compareKMers(left, right, k, parameters) (include/gcsa/algorithms.h:86-92): {shared, left only, right only}.
// With parameters.output set, the states of the unique k-mers go to output + ".left" / ".right" as in the
// reference (src/algorithms.cpp:562-610; 64 bytes per state, unordered).
inline std::array<size_type, 3> compareKMers(const GCSA& left, const GCSA& right, size_type k,
                                             const KMerSearchParameters& parameters = KMerSearchParameters())
{
  uint64_t result[3] = {0, 0, 0};
  const int ns = parameters.include_Ns ? 1 : 0, force = parameters.force ? 1 : 0;
  check(gcsa2_compare_kmers(left.handle, right.handle, k, ns, force, result), "compareKMers()");
  if(!parameters.output.empty())
  {
    std::ofstream left_output((parameters.output + ".left").c_str(), std::ios_base::binary);
    if(!left_output) { std::cerr << "compareKMers(): Cannot open output file " << parameters.output << ".left" << std::endl; return {0, 0, 0}; }
    std::ofstream right_output((parameters.output + ".right").c_str(), std::ios_base::binary);
    if(!right_output) { std::cerr << "compareKMers(): Cannot open output file " << parameters.output << ".right" << std::endl; return {0, 0, 0}; }
    std::vector<uint64_t> left_states(8 * result[1] + 8), right_states(8 * result[2] + 8);
    check(gcsa2_compare_kmers_records(left.handle, right.handle, k, ns, force, result, left_states.data(), left_states.size() / 8,
                                      right_states.data(), right_states.size() / 8), "compareKMers()");
    left_output.write(reinterpret_cast<const char*>(left_states.data()), std::streamsize(64 * result[1]));
    right_output.write(reinterpret_cast<const char*>(right_states.data()), std::streamsize(64 * result[2]));
  }
  return {size_type(result[0]), size_type(result[1]), size_type(result[2])};
}

} // namespace gcsa

#endif // GCSA2_HIP_GCSA_ALGORITHMS_H
