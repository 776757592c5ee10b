// <gcsa/algorithms.h> of the MI355X engine: the k-mer counting algorithms over the index
// (reference include/gcsa/algorithms.h:59-92, src/algorithms.cpp:364-616) as device frontier expansions, and
// verifyIndex() over a k-mer array (algorithms.h:41-55, src/algorithms.cpp:101-295) as batched queries.
// The InputGraph / NodeMapping overloads read the constructor's k-mer files and are out of scope, like printStatistics.
#ifndef GCSA2_HIP_GCSA_ALGORITHMS_H
#define GCSA2_HIP_GCSA_ALGORITHMS_H

#include "gcsa.h"
#include "lcp.h"

#include <algorithm>
#include <array>
#include <sstream>

namespace gcsa
{

// Index verification (algorithms.h:41-55): the index is queried with every distinct k-mer label of `kmers`, and
// locate() must return exactly the start nodes of the k-mers with that label; count(), and with lcp != 0 parent()
// and depth(), are checked as in src/algorithms.cpp:121-275 -- the same checks in the same order, with the same
// messages, but every query kind runs as ONE batch over all labels instead of a loop of scalar calls.
// Sorts `kmers`.  Returns false if verification fails.
inline bool verifyIndex(const GCSA& index, const LCPArray* lcp, std::vector<KMer>& kmers, size_type kmer_length)
{
  constexpr size_type MAX_ERRORS = 100, RANDOM_LOCATE_SIZE = 10;            // algorithms.cpp:50, 92
  const double start = readTimer();
  size_type fails = 0;
  auto failure = [&fails](const std::string& message)                        // printFailure(), algorithms.cpp:73-82
  {
    if(fails == MAX_ERRORS) { std::cerr << "verifyIndex(): There were further errors" << std::endl; }
    fails++;
    if(fails <= MAX_ERRORS) { std::cerr << "verifyIndex(): " << message << std::endl; }
  };
  auto show = [](range_type r) { return "(" + std::to_string(r.first) + ", " + std::to_string(r.second) + ")"; };
  auto show_occs = [](const std::vector<node_type>& occs)
  {
    std::string out = "{";
    for(size_type i = 0; i < occs.size(); i++) { out += (i > 0 ? ", " : " ") + Node::decode(occs[i]); }
    return out + " }";
  };

  // distinct labels: the pattern (cut after the first endmarker) and the sorted distinct start nodes
  std::sort(kmers.begin(), kmers.end());
  std::vector<std::string> patterns;
  std::vector<std::vector<node_type>> expected;
  for(size_type i = 0; i < kmers.size(); )
  {
    size_type next = i + 1;
    while(next < kmers.size() && Key::label(kmers[next].key) == Key::label(kmers[i].key)) { next++; }
    std::string kmer = Key::decode(kmers[i].key, kmer_length, index.alpha);
    const size_type endmarker = kmer.find('$');
    if(endmarker != std::string::npos) { kmer.resize(endmarker + 1); }
    std::vector<node_type> from;
    for(size_type j = i; j < next; j++) { from.push_back(kmers[j].from); }
    removeDuplicates(from, false);
    patterns.push_back(kmer); expected.push_back(from);
    i = next;
  }
  const size_type unique = patterns.size();
  auto find_prefixes = [&](const std::vector<size_type>& which, const std::vector<size_type>& length)
  {
    std::vector<std::uint8_t> bytes;
    std::vector<size_type> offsets(1, 0);
    for(size_type j = 0; j < which.size(); j++)
    {
      bytes.insert(bytes.end(), patterns[which[j]].begin(), patterns[which[j]].begin() + length[j]);
      offsets.push_back(bytes.size());
    }
    return index.find_batch(bytes, offsets);
  };

  // find()
  std::vector<size_type> all(unique), full(unique);
  for(size_type i = 0; i < unique; i++) { all[i] = i; full[i] = patterns[i].length(); }
  const std::vector<range_type> ranges = (unique > 0 ? find_prefixes(all, full) : std::vector<range_type>());
  std::vector<size_type> alive;
  for(size_type i = 0; i < unique; i++)
  {
    if(Range::empty(ranges[i])) { failure("find(" + patterns[i] + ") returned empty range"); }
    else { alive.push_back(i); }
  }

  // parent() and depth(): parent(range) must be the range of the longest proper prefix whose range differs
  if(lcp != 0 && !alive.empty())
  {
    std::vector<range_type> queried(alive.size());
    for(size_type j = 0; j < alive.size(); j++) { queried[j] = ranges[alive[j]]; }
    const std::vector<LCPArray::node_type> parents = lcp->parent_batch(queried);
    std::vector<size_type> end(alive.size()), open(alive.size());
    std::vector<range_type> shorter(alive.size());
    for(size_type j = 0; j < alive.size(); j++) { end[j] = patterns[alive[j]].length(); open[j] = j; shorter[j] = queried[j]; }
    while(!open.empty())                                    // one batch of find() per prefix length still undecided
    {
      std::vector<size_type> which, length;
      for(size_type j : open) { end[j]--; which.push_back(alive[j]); length.push_back(end[j]); }
      const std::vector<range_type> found = find_prefixes(which, length);
      std::vector<size_type> still;
      for(size_type k = 0; k < open.size(); k++)
      {
        const size_type j = open[k];
        shorter[j] = found[k];
        if(found[k] == queried[j] && end[j] > 0) { still.push_back(j); }
      }
      open.swap(still);
    }
    std::vector<size_type> survivors, parent_of;
    std::vector<range_type> parent_ranges;
    for(size_type j = 0; j < alive.size(); j++)
    {
      if(parents[j].range() != shorter[j] || parents[j].lcp() != end[j])
      {
        std::ostringstream ss;
        ss << "parent" << show(queried[j]) << " returned " << show(parents[j].range()) << " at depth " << parents[j].lcp()
           << ", expected " << show(shorter[j]) << " at depth " << end[j];
        failure(ss.str());
      }
      else { parent_of.push_back(j); parent_ranges.push_back(parents[j].range()); }
    }
    const std::vector<size_type> depths = lcp->depth_batch(parent_ranges);
    for(size_type k = 0; k < parent_of.size(); k++)
    {
      const size_type j = parent_of[k];
      if(depths[k] != parents[j].lcp())
      {
        failure("depth" + show(parent_ranges[k]) + " returned " + std::to_string(depths[k]) + ", expected " + std::to_string(parents[j].lcp()));
      }
      else { survivors.push_back(alive[j]); }
    }
    alive.swap(survivors);
  }

  // count()
  {
    std::vector<range_type> queried(alive.size());
    for(size_type j = 0; j < alive.size(); j++) { queried[j] = ranges[alive[j]]; }
    const std::vector<size_type> counts = index.count_batch(queried);
    std::vector<size_type> survivors;
    for(size_type j = 0; j < alive.size(); j++)
    {
      const size_type i = alive[j];
      if(counts[j] != expected[i].size())
      {
        failure("count" + show(ranges[i]) + " failed: Expected " + std::to_string(expected[i].size()) + " occurrences, got " + std::to_string(counts[j]));
      }
      else { survivors.push_back(i); }
    }
    alive.swap(survivors);
  }

  // locate(), then locate() of at most RANDOM_LOCATE_SIZE random occurrences
  if(!alive.empty())
  {
    std::vector<range_type> queried(alive.size());
    for(size_type j = 0; j < alive.size(); j++) { queried[j] = ranges[alive[j]]; }
    std::vector<size_type> offsets;
    std::vector<node_type> values;
    index.locate_batch(queried, offsets, values);
    for(size_type j = 0; j < alive.size(); j++)
    {
      const size_type i = alive[j];
      const std::vector<node_type> occs(values.begin() + offsets[j], values.begin() + offsets[j + 1]);
      if(occs.size() != expected[i].size())
      {
        failure("locate(" + patterns[i] + ") failed: Expected " + std::to_string(expected[i].size()) + " occurrences, got " + std::to_string(occs.size()));
        continue;
      }
      const auto differ = std::mismatch(occs.begin(), occs.end(), expected[i].begin());
      if(differ.first != occs.end())
      {
        failure("locate(" + patterns[i] + ") failed: Expected " + Node::decode(*differ.second) + ", got " + Node::decode(*differ.first));
        continue;
      }
      std::vector<node_type> random_occs;
      index.locate(ranges[i], RANDOM_LOCATE_SIZE, random_occs);
      const size_type expected_occs = std::min(RANDOM_LOCATE_SIZE, size_type(occs.size()));
      if(random_occs.size() != expected_occs)
      {
        failure("locate(" + patterns[i] + ") failed: Expected " + std::to_string(expected_occs) + " random occurrences, got " + std::to_string(random_occs.size()));
      }
      else if(!std::is_sorted(random_occs.begin(), random_occs.end())
              || std::adjacent_find(random_occs.begin(), random_occs.end()) != random_occs.end()
              || !std::includes(occs.begin(), occs.end(), random_occs.begin(), random_occs.end()))
      {
        failure("locate(" + patterns[i] + ") failed: " + show_occs(random_occs) + " is not a subset of " + show_occs(occs));
      }
    }
  }

  const double seconds = readTimer() - start;
  std::cout << "Queried the index with " << unique << " patterns in " << seconds << " seconds ("
            << (unique / seconds) << " patterns / second)" << std::endl;
  if(fails == 0) { std::cout << "Index verification complete" << std::endl; }
  else { std::cout << "Index verification failed for " << fails << " patterns" << std::endl; }
  std::cout << std::endl;
  return fails == 0;
}

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
