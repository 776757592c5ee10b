// <gcsa/gcsa.h> of the MI355X engine: class gcsa::GCSA with the reference's public query interface
// (reference include/gcsa/gcsa.h:40-277), every call forwarded to the device image behind the C ABI
// (include/gcsa2_hip.h).  Names, argument meaning, result types and error behaviour are the reference's
// (citations on each method), so a caller such as vg's MEM finder or the reference's own query_gcsa /
// count_kmers tools compiles against this header unchanged; `*_batch` methods are the additional,
// throughput-oriented entry points.  Header-only; link with -lgcsa2_hip.
//
// What differs, by design: an index is a device image, not SDSL members in host memory.  It is immutable and
// shared by copies (copying a GCSA copies a reference).  The SDSL-typed data members of the reference
// (gcsa.h:214-240) do not exist; `header` and `alpha` do.  Construction from an InputGraph is out of scope.
#ifndef GCSA2_HIP_GCSA_GCSA_H
#define GCSA2_HIP_GCSA_GCSA_H

#include "files.h"
#include "support.h"

namespace gcsa
{

class GCSA
{
public:
  typedef gcsa::size_type size_type;

  GCSA() : handle(nullptr) {}                                                // gcsa.cpp:56-58: an empty index
  GCSA(const GCSA& source) = default;                                        // copies share the immutable device image
  GCSA(GCSA&& source) noexcept { this->take(source); }
  ~GCSA() = default;
  void swap(GCSA& another)
  {
    std::swap(header, another.header); alpha.swap(another.alpha); std::swap(handle, another.handle);
    owner.swap(another.owner); host.swap(another.host);
  }
  GCSA& operator=(const GCSA& source) = default;
  GCSA& operator=(GCSA&& source) noexcept { if(this != &source) { this->take(source); } return *this; }

  inline static const std::string EXTENSION = ".gcsa";                       // gcsa.cpp:51

  // GCSA::load (gcsa.cpp:184-216): reads one serialized index from the stream, builds the device image on
  // Device::current().  Throws std::runtime_error("GCSA::load(): Invalid header: ...") like the reference on a
  // bad header, and on any other inconsistency.  Bytes of the stream after the index are left unread when the
  // stream is seekable.
  void load(std::istream& in)
  {
    const std::streampos start = in.tellg();
    std::vector<char> data = readRest(in);
    gcsa2_view_storage* storage = nullptr;
    std::uint64_t consumed = 0;
    if(gcsa2_host_view_parse_gcsa(data.data(), data.size(), &consumed, &storage) != GCSA2_OK) { throw std::runtime_error(gcsa2_last_error()); }
    std::shared_ptr<gcsa2_view_storage> keep(storage, gcsa2_host_view_free);
    in.clear();
    if(start != std::streampos(-1)) { in.seekg(start + std::streamoff(consumed)); }
    this->adopt(*gcsa2_host_view_get(storage), Device::current(), keep);
  }

  // GCSA::serialize (gcsa.cpp:140-179): the reference's byte stream.  The device image does not keep the host-side
  // arrays, so this works on an index whose host view was retained (retainHostView(true) before it was loaded or
  // created); otherwise it throws.  Returns the number of bytes written; the structure tree is not filled in.
  // EXPERIMENTAL: the SDSL container encodings inside the stream (bit_vector_il rank samples, sd_vector, the stored
  // select_support_mcl directories) are restated from sdsl-lite and have never met a file written by the real library.
  // The reference LOADS stored select directories without checking them, so a mismatch there would surface as wrong
  // select() / locate() answers in reference tools, not as a load error: do not feed files written here to reference tools
  // before one has been compared with the real library's output for the same index.
  size_type serialize(std::ostream& out, sdsl::structure_tree_node* = nullptr, std::string = "") const
  {
    if(!host) { throw std::runtime_error("GCSA::serialize(): the host view of this index was not retained (GCSA::retainHostView)"); }
    std::uint64_t written = 0;
    check(gcsa2_host_view_serialize_gcsa(gcsa2_host_view_get(host.get()), &GCSA::write_to, &out, &written), "GCSA::serialize()");
    return written;
  }
  const std::shared_ptr<gcsa2_index>& shared() const { return owner; }      // for LCPArray(const GCSA&): keeps the image alive
  static bool& retainHostView() { static bool retain = false; return retain; }
  static void retainHostView(bool retain) { retainHostView() = retain; }

  // ---- engine-specific construction ----
  // From the plain arrays of a loaded reference index, onto HIP device `device`.
  explicit GCSA(const gcsa2_host_view& view, int device = 0) : handle(nullptr) { this->adopt(view, device, nullptr); }
  // From a G2HV container file (gcsa2_host_view_save); throws on an invalid header.
  explicit GCSA(const std::string& container_file, int device = 0) : handle(nullptr)
  {
    gcsa2_index* raw = nullptr;
    check(gcsa2_index_create_from_file(container_file.c_str(), device, &raw), "GCSA::GCSA()");
    this->own(raw);
  }
  // From the reference's own files, base.gcsa + (optionally) its .lcp IN ONE IMAGE, which the fused
  // LF + parent kernels need (matching statistics); as query_gcsa opens them (benchmark/query_gcsa.cpp:53-63).
  GCSA(const std::string& gcsa_file, const std::string& lcp_file, int device) : handle(nullptr)
  {
    gcsa2_index* raw = nullptr;
    check(gcsa2_index_create_from_gcsa(gcsa_file.c_str(), lcp_file.empty() ? nullptr : lcp_file.c_str(), device, &raw), "GCSA::load()");
    this->own(raw);
  }

  // ---- high-level interface (gcsa.h:96-128) ----
  template<class Iterator>
  range_type find(Iterator begin, Iterator end) const                       // gcsa.h:96-110
  {
    if(handle == nullptr) { return range_type(0, size_type(0) - 1); }        // empty index: (0, size() - 1), gcsa.h:99
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

  // Batched find of k-mers handed over as 2-bit codes (gcsa2_find_batch_packed: `length` characters each, last character
  // first, comp - 1 in two bits, ceil(length / 32) words per pattern): 8 instead of 40 bytes per 32-mer over the link.
  std::vector<range_type> find_packed_batch(const std::vector<std::uint64_t>& codes, size_type length) const
  {
    const size_type words = (length + 31) / 32, nq = (words == 0 ? 0 : codes.size() / words);
    std::vector<range_type> result(nq);
    if(nq > 0) { check(gcsa2_find_batch_packed(handle, codes.data(), length, nq, reinterpret_cast<size_type*>(result.data())), "GCSA::find_packed_batch()"); }
    return result;
  }
  // codes of `count` patterns of `length` bytes each, through THIS index's alphabet: comp - 1 of the fast characters
  // (char2comp values 1..4); any other byte throws.  The packing a caller does.
  std::vector<std::uint64_t> packKMers(const std::uint8_t* patterns, size_type count, size_type length) const
  {
    std::uint8_t code_of[256];
    for(size_type c = 0; c < 256; c++) { const size_type comp = alpha.char2comp[c]; code_of[c] = (comp >= 1 && comp <= 4 ? std::uint8_t(comp - 1) : 0xFF); }
    return pack_with(code_of, patterns, count, length);
  }
  // the same for the default alphabet "$ACGTN#" (A C G T in either case): no index needed
  static std::vector<std::uint64_t> pack_kmers(const std::uint8_t* patterns, size_type count, size_type length)
  {
    std::uint8_t code_of[256];
    for(size_type c = 0; c < 256; c++) { code_of[c] = 0xFF; }
    code_of['A'] = code_of['a'] = 0; code_of['C'] = code_of['c'] = 1; code_of['G'] = code_of['g'] = 2; code_of['T'] = code_of['t'] = 3;
    return pack_with(code_of, patterns, count, length);
  }
  static std::vector<std::uint64_t> pack_with(const std::uint8_t* code_of, const std::uint8_t* patterns, size_type count, size_type length)
  {
    const size_type words = (length + 31) / 32;
    std::vector<std::uint64_t> codes(count * words, 0);
    for(size_type q = 0; q < count; q++)
    {
      for(size_type t = 0; t < length; t++)                  // distance t from the pattern's end
      {
        const std::uint64_t code = code_of[patterns[q * length + (length - 1 - t)]];
        if(code > 3) { throw std::invalid_argument("GCSA::pack_kmers(): a pattern holds a character that is not one of the index's four fast characters"); }
        codes[q * words + (t >> 5)] |= code << (2 * (t & 31));
      }
    }
    return codes;
  }

  // The LF + parent() loop of a MEM finder for a whole batch (gcsa2_match_breaks_batch; the reference's caller shape:
  // src/algorithms.cpp:146-167): the left-maximal matches of every pattern, of at least min_length characters, as
  // {position, length, sp, ep} with (sp, ep) = find() of the match.  Needs an index created together with its LCP array
  // (the two-file constructor / load of base.gcsa + base.lcp).  breaks of pattern q: [offsets_out[q], offsets_out[q + 1]).
  void match_breaks_batch(const std::vector<std::uint8_t>& patterns, const std::vector<size_type>& offsets, size_type min_length,
                          std::vector<size_type>& offsets_out, std::vector<gcsa2_break>& breaks) const
  {
    const size_type nq = offsets.empty() ? 0 : offsets.size() - 1;
    offsets_out.assign(nq + 1, 0);
    if(nq == 0) { breaks.clear(); return; }                    // an empty batch: no records
    breaks.assign(4 * nq + 16, gcsa2_break());
    std::uint8_t dummy = 0;
    size_type total = 0;
    for(int attempt = 0; attempt < 2; attempt++)
    {
      const int rc = gcsa2_match_breaks_batch(handle, patterns.empty() ? &dummy : patterns.data(), offsets.data(), nq, min_length, offsets_out.data(),
                                              breaks.data(), breaks.size(), &total, nullptr, nullptr);
      if(rc == GCSA2_ERR_BUFFER_TOO_SMALL && total > breaks.size()) { breaks.assign(total, gcsa2_break()); continue; }
      check(rc, "GCSA::match_breaks_batch()");
      break;
    }
    breaks.resize(total);
  }

  // Memory pressure (gcsa2_index_set_tables / gcsa2_index_trim): drop (0), build (1) or leave (-1) the pair blocks and the
  // locate table, resize the seed table (kmer_k: -1 leaves it, 0 drops it); give back staging and scratch memory.  Results
  // never change.  Not to be called while queries run on this index or on copies of it (copies share the device image).
  void setTables(int pair_blocks, int kmer_k, int locate_table) { check(gcsa2_index_set_tables(owner.get(), pair_blocks, kmer_k, locate_table), "GCSA::setTables()"); }
  void trim() { check(gcsa2_index_trim(owner.get()), "GCSA::trim()"); }
  // Shape of the host pipeline behind find_batch / find_packed_batch (gcsa2_index_set_pipeline): host threads, log2 of the
  // patterns per chunk, spinning (0) or sleeping (1) waits; 0 / 0 / -1 leave a value as it is.
  void setPipeline(int lanes, int chunk_log2, int blocking = -1) { check(gcsa2_index_set_pipeline(owner.get(), lanes, chunk_log2, blocking), "GCSA::setPipeline()"); }
  size_type deviceBytes() const { return gcsa2_device_bytes(handle); }

  size_type count(range_type range) const                                   // src/gcsa.cpp:802-809
  {
    size_type in[2] = { range.first, range.second }, out = 0;
    if(handle == nullptr) { return 0; }
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
    if(handle == nullptr) { return; }
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
  size_type size() const { return header.path_nodes; }                      // gcsa.h:137-148
  bool empty() const { return size() == 0; }
  size_type edgeCount() const { return header.edges; }
  size_type order() const { return header.order; }
  size_type sampleCount() const { return handle != nullptr ? gcsa2_sample_count(handle) : 0; }
  size_type sampleBits() const { return handle != nullptr ? gcsa2_sample_bits(handle) : 0; }
  size_type sampledPositions() const { return handle != nullptr ? gcsa2_sampled_positions(handle) : 0; }

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

  GCSAHeader header;         // gcsa.h:214
  Alphabet   alpha;          // gcsa.h:215: alpha.char2comp / comp2char / C / sigma / fast_chars
  gcsa2_index* handle;       // the device image, for the C ABI (nullptr: empty index)

private:
  static void sort_unique(std::vector<node_type>& v) { removeDuplicates(v); }     // utils.h:350-357, for append == true
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

  void own(gcsa2_index* raw)
  {
    owner.reset(raw, gcsa2_index_destroy);
    handle = raw;
    header = GCSAHeader();
    header.path_nodes = gcsa2_size(raw); header.edges = gcsa2_edge_count(raw); header.order = gcsa2_order(raw);
    alpha.read(raw);
  }
  void adopt(const gcsa2_host_view& view, int device, std::shared_ptr<gcsa2_view_storage> storage)
  {
    gcsa2_index* raw = nullptr;
    check(gcsa2_index_create(&view, device, &raw), "GCSA::GCSA()");
    this->own(raw);
    host = (retainHostView() ? storage : nullptr);
  }
  void take(GCSA& source)
  {
    header = source.header; alpha = std::move(source.alpha); handle = source.handle;
    owner = std::move(source.owner); host = std::move(source.host);
    source.handle = nullptr; source.header = GCSAHeader(); source.alpha = Alphabet();
  }
  static void write_to(void* stream, const void* data, std::uint64_t bytes)
  {
    static_cast<std::ostream*>(stream)->write(static_cast<const char*>(data), std::streamsize(bytes));
  }

  std::shared_ptr<gcsa2_index> owner;
  std::shared_ptr<gcsa2_view_storage> host;
};

} // namespace gcsa

#endif // GCSA2_HIP_GCSA_GCSA_H
