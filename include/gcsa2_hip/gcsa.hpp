// gcsa.hpp -- C++ facade over the C ABI (gcsa2_hip.h) with the reference's class API.
//
// jltsiren/gcsa2 has no plugin ABI: its boundary is the class API of `gcsa::GCSA` and
// `gcsa::LCPArray` (reference include/gcsa/gcsa.h:40-277, include/gcsa/lcp.h:90-194), whose
// query methods are header-inline C++ over SDSL members.  This header keeps the names,
// argument meaning and result types of those methods (citations on each one) so that a caller
// such as vg's MEM finder compiles against it unchanged, and forwards every call to the
// MI355X engine.  Scalar methods are one-element batches; `*_batch` methods are the new,
// throughput-oriented entry points.  Header-only; link with -lgcsa2_hip.
//
// Not provided (outside the query hot path): construction from an
// InputGraph, serialize()/load() of .gcsa/.lcp files, algorithms.h.  An index is created from a
// `gcsa2_host_view`, which a maintainer fills from the SDSL members of a loaded reference
// GCSA.
#ifndef GCSA2_HIP_GCSA_HPP
#define GCSA2_HIP_GCSA_HPP

#include "../gcsa2_hip.h"

#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <array>
#include <fstream>
#include <iostream>
#include <vector>

namespace gcsa
{

typedef std::uint64_t size_type;                          // utils.h:55
typedef std::uint8_t  comp_type;                          // utils.h:60
typedef std::uint64_t node_type;                          // support.h:441
typedef std::pair<size_type, size_type> range_type;       // utils.h:84

struct Range   // utils.h:86-117
{
  static size_type length(range_type range) { return range.second + 1 - range.first; }
  static bool empty(range_type range) { return (range.first + 1 > range.second + 1); }
  static bool empty(size_type sp, size_type ep) { return (sp + 1 > ep + 1); }
  static range_type empty_range() { return range_type(1, 0); }
};

struct Node    // support.h:443-471
{
  constexpr static size_type OFFSET_BITS = 10;
  constexpr static size_type ID_OFFSET = OFFSET_BITS + 1;
  constexpr static size_type ORIENTATION_MASK = static_cast<size_type>(1) << OFFSET_BITS;
  constexpr static size_type OFFSET_MASK = ORIENTATION_MASK - 1;
  static node_type encode(size_type id, size_type offset) { return (id << ID_OFFSET) | offset; }
  static node_type encode(size_type id, size_type offset, bool rc) { return encode(id, offset) | (rc ? ORIENTATION_MASK : 0); }
  static size_type id(node_type node) { return node >> ID_OFFSET; }
  static bool rc(node_type node) { return node & ORIENTATION_MASK; }
  static size_type offset(node_type node) { return node & OFFSET_MASK; }
};

struct STNode  // lcp.h:40-79
{
  size_type sp, ep, left_lcp, right_lcp, node_lcp;
  constexpr static size_type UNKNOWN = ~(size_type)0;
  STNode() : sp(0), ep(0), left_lcp(0), right_lcp(0), node_lcp(0) {}
  STNode(size_type start, size_type end, size_type left, size_type right, size_type depth) :
    sp(start), ep(end), left_lcp(left), right_lcp(right), node_lcp(depth) {}
  range_type range() const { return range_type(sp, ep); }
  size_type lcp() const { return node_lcp; }
  bool operator==(const STNode& n) const { return sp == n.sp && ep == n.ep; }
  bool operator==(range_type r) const { return sp == r.first && ep == r.second; }
  bool operator!=(const STNode& n) const { return !(*this == n); }
  bool operator!=(range_type r) const { return !(*this == r); }
};

// The engine reports failures as status codes; the facade turns them into exceptions, as the
// reference's load() does (src/gcsa.cpp:188-193).  Query methods of the reference never fail.
inline void check(int status, const char* what)
{
  if(status != GCSA2_OK) { throw std::runtime_error(std::string(what) + ": " + gcsa2_last_error()); }
}

struct AlphabetView   // the Alphabet members callers read (support.h:150-151)
{
  std::vector<std::uint8_t> char2comp;
  std::vector<size_type> C;
  size_type sigma = 0, fast_chars = 0;
};

class GCSA
{
public:
  typedef gcsa::size_type size_type;

  GCSA() : handle(nullptr) {}
  GCSA(const gcsa2_host_view& view, int device = 0) : handle(nullptr)
  {
    check(gcsa2_index_create(&view, device, &handle), "GCSA::GCSA()");
    alpha.sigma = gcsa2_sigma(handle); alpha.fast_chars = gcsa2_fast_chars(handle);
    alpha.char2comp.resize(256); alpha.C.resize(alpha.sigma + 1);
    gcsa2_alphabet(handle, alpha.char2comp.data(), alpha.C.data());
  }
  // Opens a G2HV container file (gcsa2_host_view_save).  Like the reference's
  // load() (src/gcsa.cpp:184-216) this throws std::runtime_error on an invalid header.
  explicit GCSA(const std::string& container_file, int device = 0) : handle(nullptr)
  {
    check(gcsa2_index_create_from_file(container_file.c_str(), device, &handle), "GCSA::GCSA()");
    alpha.sigma = gcsa2_sigma(handle); alpha.fast_chars = gcsa2_fast_chars(handle);
    alpha.char2comp.resize(256); alpha.C.resize(alpha.sigma + 1);
    gcsa2_alphabet(handle, alpha.char2comp.data(), alpha.C.data());
  }
  // Opens the reference's own files: base.gcsa + (optionally) its .lcp, as query_gcsa does with
  // sdsl::load_from_file (benchmark/query_gcsa.cpp:53-63).  Pass an empty lcp_file for no LCPArray.
  GCSA(const std::string& gcsa_file, const std::string& lcp_file, int device) : handle(nullptr)
  {
    check(gcsa2_index_create_from_gcsa(gcsa_file.c_str(), lcp_file.empty() ? nullptr : lcp_file.c_str(), device, &handle), "GCSA::load()");
    alpha.sigma = gcsa2_sigma(handle); alpha.fast_chars = gcsa2_fast_chars(handle);
    alpha.char2comp.resize(256); alpha.C.resize(alpha.sigma + 1);
    gcsa2_alphabet(handle, alpha.char2comp.data(), alpha.C.data());
  }
  GCSA(const GCSA&) = delete;
  GCSA& operator=(const GCSA&) = delete;
  GCSA(GCSA&& source) noexcept : alpha(std::move(source.alpha)), handle(source.handle) { source.handle = nullptr; }
  GCSA& operator=(GCSA&& source) noexcept
  {
    if(this != &source) { gcsa2_index_destroy(handle); handle = source.handle; source.handle = nullptr; alpha = std::move(source.alpha); }
    return *this;
  }
  ~GCSA() { gcsa2_index_destroy(handle); }

  // ---- high-level interface (gcsa.h:96-128) ----
  template<class Iterator>
  range_type find(Iterator begin, Iterator end) const                       // gcsa.h:96-110
  {
    std::vector<std::uint8_t> pattern(begin, end);
    size_type offsets[2] = { 0, pattern.size() };
    std::uint8_t dummy = 0;
    size_type range[2];
    check(gcsa2_find_batch(handle, pattern.empty() ? &dummy : pattern.data(), offsets, 1, range), "GCSA::find()");
    return range_type(range[0], range[1]);
  }
  template<class Container>
  range_type find(const Container& pattern) const { return find(pattern.begin(), pattern.end()); }   // gcsa.h:112-116
  template<class Element>
  range_type find(const Element* pattern, size_type length) const { return find(pattern, pattern + length); }  // gcsa.h:118-122

  // Batched find: patterns concatenated, pattern q = [offsets[q], offsets[q + 1]).
  std::vector<range_type> find_batch(const std::vector<std::uint8_t>& patterns, const std::vector<size_type>& offsets) const
  {
    size_type nq = offsets.empty() ? 0 : offsets.size() - 1;
    std::vector<range_type> result(nq);
    static_assert(sizeof(range_type) == 2 * sizeof(size_type), "range_type must be two packed u64");
    std::uint8_t dummy = 0;
    check(gcsa2_find_batch(handle, patterns.empty() ? &dummy : patterns.data(), offsets.data(), nq,
                           reinterpret_cast<size_type*>(result.data())), "GCSA::find_batch()");
    return result;
  }

  size_type count(range_type range) const                                   // src/gcsa.cpp:802-809
  {
    size_type in[2] = { range.first, range.second }, out = 0;
    check(gcsa2_count_batch(handle, in, 1, &out), "GCSA::count()");
    return out;
  }

  std::vector<size_type> count_batch(const std::vector<range_type>& ranges) const
  {
    std::vector<size_type> out(ranges.size());
    size_type dummy_in[2] = { 1, 0 }, dummy_out = 0;
    check(gcsa2_count_batch(handle, ranges.empty() ? dummy_in : reinterpret_cast<const size_type*>(ranges.data()), ranges.size(),
                            ranges.empty() ? &dummy_out : out.data()), "GCSA::count_batch()");
    return out;
  }

  void locate(size_type path, std::vector<node_type>& results, bool append = false, bool sort = true) const  // gcsa.cpp:813-825
  {
    locate(range_type(path, path), results, append, sort);
  }

  // gcsa.cpp:827-842: sort == false keeps path order and duplicates, as the reference does.
  void locate(range_type range, std::vector<node_type>& results, bool append = false, bool sort = true) const
  {
    if(!append) { results.clear(); }
    size_type in[2] = { range.first, range.second }, offsets[2] = { 0, 0 };
    gcsa2_locate_job* job = nullptr;
    check(gcsa2_locate_run(handle, in, 1, sort ? 1 : 0, offsets, &job), "GCSA::locate()");
    size_type old = results.size();
    results.resize(old + offsets[1]);
    node_type dummy = 0;
    check(gcsa2_locate_fetch(job, offsets[1] ? results.data() + old : &dummy, offsets[1] ? offsets[1] : 1), "GCSA::locate()");
    if(append && sort && old > 0) { sort_unique(results); }
  }

  void locate(range_type range, size_type max_positions, std::vector<node_type>& results) const   // gcsa.cpp:844-878
  {
    results.clear();
    size_type total = count(range);
    if(total == 0) { return; }
    results.resize(max_positions < total ? max_positions : total);
    size_type got = 0;
    check(gcsa2_locate_max(handle, range.first, range.second, max_positions, results.data(), results.size(), &got), "GCSA::locate()");
    results.resize(got);
  }

  // CSR batch: offsets[q] .. offsets[q + 1] index the sorted distinct values of ranges[q].
  void locate_batch(const std::vector<range_type>& ranges, std::vector<size_type>& offsets, std::vector<node_type>& values) const
  {
    offsets.assign(ranges.size() + 1, 0);
    gcsa2_locate_job* job = nullptr;
    check(gcsa2_locate_run(handle, reinterpret_cast<const size_type*>(ranges.data()), ranges.size(), 1, offsets.data(), &job), "GCSA::locate_batch()");
    values.resize(offsets.back() ? offsets.back() : 1);
    check(gcsa2_locate_fetch(job, values.data(), values.size()), "GCSA::locate_batch()");
    values.resize(offsets.back());
  }

  // ---- low-level interface (gcsa.h:137-210) ----
  size_type size() const { return gcsa2_size(handle); }
  bool empty() const { return size() == 0; }
  size_type edgeCount() const { return gcsa2_edge_count(handle); }
  size_type order() const { return gcsa2_order(handle); }
  size_type sampleCount() const { return gcsa2_sample_count(handle); }
  size_type sampleBits() const { return gcsa2_sample_bits(handle); }
  size_type sampledPositions() const { return gcsa2_sampled_positions(handle); }

  range_type charRange(comp_type comp) const                                // gcsa.h:150-153
  {
    range_type r;
    check(gcsa2_char_range(handle, comp, &r.first, &r.second), "GCSA::charRange()");
    return r;
  }

  range_type LF(range_type range, comp_type comp) const                     // gcsa.h:155-162
  {
    size_type in[2] = { range.first, range.second }, out[2];
    check(gcsa2_lf_batch(handle, in, &comp, 1, out), "GCSA::LF()");
    return range_type(out[0], out[1]);
  }

  std::vector<range_type> LF_batch(const std::vector<range_type>& ranges, const std::vector<comp_type>& comps) const
  {
    std::vector<range_type> out(ranges.size());
    check(gcsa2_lf_batch(handle, reinterpret_cast<const size_type*>(ranges.data()), comps.data(), ranges.size(),
                         reinterpret_cast<size_type*>(out.data())), "GCSA::LF_batch()");
    return out;
  }

  size_type LF(size_type path_node) const                                   // gcsa.h:165-183
  {
    size_type out = 0;
    check(gcsa2_lf_node_batch(handle, &path_node, 1, &out), "GCSA::LF()");
    return out;
  }

  // results must hold sigma entries, as in the reference (gcsa.cpp:742-798)
  void LF_fast(range_type range, std::vector<range_type>& results) const { lf_all(range, results, 0); }
  void LF_all(range_type range, std::vector<range_type>& results) const { lf_all(range, results, 1); }

  bool sampled(size_type path_node) const { return sample_info(path_node)[0] != 0; }              // gcsa.h:191
  range_type sampleRange(size_type path_node) const                                                // gcsa.h:193-200
  { std::vector<size_type> s = sample_info(path_node); return range_type(s[1], s[2]); }
  size_type firstSample(size_type path_node) const { return sample_info(path_node)[1]; }           // gcsa.h:202-206
  bool lastSample(size_type i) const { size_type v; std::uint8_t l; check(gcsa2_sample_batch(handle, &i, 1, &v, &l), "GCSA::lastSample()"); return l != 0; }  // gcsa.h:208
  node_type sample(size_type i) const { size_type v; std::uint8_t l; check(gcsa2_sample_batch(handle, &i, 1, &v, &l), "GCSA::sample()"); return v; }          // gcsa.h:210

  AlphabetView alpha;        // alpha.char2comp / alpha.C / alpha.sigma / alpha.fast_chars
  gcsa2_index* handle;       // the device image (public, like the reference's data members)

private:
  static void sort_unique(std::vector<node_type>& v)
  {
    // removeDuplicates (utils.h:350-357); small host-side merge used only for append == true
    for(size_type i = 1; i < v.size(); i++) { node_type x = v[i]; size_type j = i; while(j > 0 && v[j - 1] > x) { v[j] = v[j - 1]; j--; } v[j] = x; }
    size_type tail = 0;
    for(size_type i = 0; i < v.size(); i++) { if(i == 0 || v[i] != v[tail - 1]) { v[tail++] = v[i]; } }
    v.resize(tail);
  }
  void lf_all(range_type range, std::vector<range_type>& results, int all) const
  {
    std::vector<size_type> out(2 * alpha.sigma);
    size_type in[2] = { range.first, range.second };
    check(gcsa2_lf_all_batch(handle, in, 1, all, out.data()), "GCSA::LF_all()");
    size_type limit = all ? alpha.sigma - 2 : alpha.fast_chars;
    for(size_type c = 1; c <= limit && c < results.size(); c++) { results[c] = range_type(out[2 * c], out[2 * c + 1]); }
  }
  std::vector<size_type> sample_info(size_type node) const
  {
    std::vector<size_type> out(3);
    check(gcsa2_sample_range_batch(handle, &node, 1, out.data()), "GCSA::sampleRange()");
    return out;
  }
};

// Suffix-tree operations over the LCP part of the same device image (lcp.h:90-194).
class LCPArray
{
public:
  typedef gcsa::size_type size_type;
  typedef STNode node_type;

  explicit LCPArray(const GCSA& index) : handle(index.handle) {}

  size_type size() const { return gcsa2_lcp_size(handle); }
  size_type values() const { return gcsa2_lcp_values(handle); }
  size_type levels() const { return gcsa2_lcp_levels(handle); }
  size_type branching() const { return gcsa2_lcp_branching(handle); }
  size_type operator[](size_type i) const { size_type v; check(gcsa2_lcp_access_batch(handle, &i, 1, &v), "LCPArray::operator[]"); return v; }

  node_type root() const { return node_type(0, size() - 1, 0, 0, 0); }                            // lcp.h:137
  range_type notFound() const { return range_type(values(), values()); }                          // lcp.h:178

  node_type parent(range_type range) const                                                         // lcp.cpp:297-301
  {
    size_type in[2] = { range.first, range.second };
    gcsa2_stnode out;
    check(gcsa2_parent_batch(handle, in, 1, &out), "LCPArray::parent()");
    return node_type(out.sp, out.ep, out.left_lcp, out.right_lcp, out.node_lcp);
  }
  node_type parent(const node_type& node) const { return parent(node.range()); }                  // lcp.cpp:276-295
  std::vector<node_type> parent_batch(const std::vector<range_type>& ranges) const
  {
    std::vector<gcsa2_stnode> raw(ranges.size());
    check(gcsa2_parent_batch(handle, reinterpret_cast<const size_type*>(ranges.data()), ranges.size(), raw.data()), "LCPArray::parent_batch()");
    std::vector<node_type> out;
    out.reserve(raw.size());
    for(const gcsa2_stnode& n : raw) { out.emplace_back(n.sp, n.ep, n.left_lcp, n.right_lcp, n.node_lcp); }
    return out;
  }

  size_type depth(range_type range) const                                                          // lcp.cpp:319-325
  {
    size_type in[2] = { range.first, range.second }, out;
    check(gcsa2_depth_batch(handle, in, 1, &out), "LCPArray::depth()");
    return out;
  }
  std::vector<size_type> depth_batch(const std::vector<range_type>& ranges) const
  {
    std::vector<size_type> out(ranges.size());
    size_type dummy_in[2] = { 0, 0 }, dummy_out = 0;
    check(gcsa2_depth_batch(handle, ranges.empty() ? dummy_in : reinterpret_cast<const size_type*>(ranges.data()), ranges.size(),
                            ranges.empty() ? &dummy_out : out.data()), "LCPArray::depth_batch()");
    return out;
  }
  size_type depth(const node_type& node) const { return node.lcp() != node_type::UNKNOWN ? node.lcp() : depth(node.range()); }  // lcp.cpp:305-309
  size_type depth(node_type& node) const { if(node.lcp() == node_type::UNKNOWN) { node.node_lcp = depth(node.range()); } return node.lcp(); }  // lcp.cpp:311-316

  range_type psv(size_type pos) const { return sv(0, pos); }                                       // lcp.cpp:370-374
  range_type psev(size_type pos) const { return sv(1, pos); }                                      // lcp.cpp:376-380
  range_type nsv(size_type pos) const { return sv(2, pos); }                                       // lcp.cpp:426-430
  range_type nsev(size_type pos) const { return sv(3, pos); }                                      // lcp.cpp:432-436
  range_type rmq(size_type sp, size_type ep) const                                                 // lcp.cpp:448-513
  {
    size_type in[2] = { sp, ep }, out[2];
    check(gcsa2_rmq_batch(handle, in, 1, out), "LCPArray::rmq()");
    return range_type(out[0], out[1]);
  }
  range_type rmq(range_type range) const { return rmq(range.first, range.second); }               // lcp.cpp:515-519

  node_type nodeFor(range_type range) const                                                        // lcp.h:163-175
  {
    size_type right = (range.second + 1 < size() ? (*this)[range.second + 1] : 0);
    return node_type(range.first, range.second, (*this)[range.first], right, node_type::UNKNOWN);
  }

private:
  range_type sv(int op, size_type pos) const
  {
    size_type out[2];
    check(gcsa2_sv_batch(handle, op, &pos, 1, out), "LCPArray::psv/nsv()");
    return range_type(out[0], out[1]);
  }
  gcsa2_index* handle;
};

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

#endif // GCSA2_HIP_GCSA_HPP
